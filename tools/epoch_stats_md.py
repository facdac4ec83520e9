"""rocprofv3 kernel stats of tools/epoch_once.py (40 epochs) -> per-epoch markdown table.  usage: epoch_stats_md.py file.csv [epochs] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ep = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
total = sum(float(r["TotalDurationNs"]) for r in rows) / ep / 1e3
print(f"GPU-busy time per epoch: {total:.0f} µs.\n")
print("| µs / epoch | launches / epoch | kernel |")
print("|---|---|---|")
for r in rows[:n]:
    name = r["Name"].replace("|", "/")
    name = name if len(name) <= 120 else name[:117] + "..."
    print(f"| {float(r['TotalDurationNs']) / ep / 1e3:.1f} | {int(r['Calls']) / ep:.1f} | `{name}` |")
