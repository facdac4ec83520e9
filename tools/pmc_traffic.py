#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) per kernel.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <key> [source note]

`out.json` is a table keyed by workload (`<workload>x<scale>_K<K>_d<d>_<dtype>`, the key bench.py looks up); the entry is
replaced, other entries are kept.  Every entry records the hash of the kernel sources it was collected for
(bench.kernel_source_hash): bench.py reports `traffic` only when that matches the build it is running.

Units and corrections, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950:
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 128-B requests at 64 B, so a wide coalesced
(16 B per lane) read stream reports exactly half its bytes -> doubled here; WRITE_SIZE is exact for
16-B-per-lane stores.  Infinity-Cache hits are included in these fabric-side counters, so "traffic"
is bytes that left the XCD L2s, an upper bound on HBM bytes.
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "dl::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    from bench import kernel_source_hash
    fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out_path, key = sys.argv[3], sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    kernels = {}
    for k in sorted(fetch):
        f_raw = fetch[k] * 1024.0
        w = write.get(k, 0.0) * 1024.0
        kernels[k] = dict(launches=n[k], fetch_bytes_raw=f_raw, fetch_bytes_corrected=2.0 * f_raw, write_bytes=w,
                          traffic_bytes=2.0 * f_raw + w)
    try:
        table = json.load(open(out_path))
        if not all(isinstance(v, dict) and "kernels" in v for v in table.values()):
            table = {}                                      # a summary in the round-1 format: start over
    except (OSError, ValueError):
        table = {}
    table[key] = dict(kernel_source_hash=kernel_source_hash(), source=note, kernels=kernels)
    json.dump(table, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:60s} fetch(x2) {v['fetch_bytes_corrected'] / 1e6:10.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
