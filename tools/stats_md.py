"""Turn a rocprofv3 *_kernel_stats.csv into a markdown table (top rows).  usage: stats_md.py file.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print("| kernel | calls | total ms | avg µs | % |")
print("|---|---|---|---|---|")
for r in rows[:n]:
    name = r["Name"].replace("|", "/")
    name = name if len(name) <= 110 else name[:107] + "..."
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
