#!/usr/bin/env python3
"""Summarise rocprofv3 L2-request passes (tools/pmc_l2_run.sh) per kernel -> profiles/pmc_l2_latest.json format.

    python tools/pmc_l2.py <pass1 counter_collection.csv> <pass2 counter_collection.csv> <out.json> <key> [source note]

Per kernel (mean over its launches): TCC_REQ / TCC_READ / TCC_WRITE / TCC_HIT / TCC_MISS and
    l2_bytes = TCC_READ * 128 + TCC_WRITE * 64
the bytes the compute units requested from the XCD L2s (units from the calibration of round 3,
profiles/r3_pmc_calibration.txt: a 3 GiB 16-B-per-lane read stream = 25.17 M requests of 128 B, a 1.5 GiB store stream =
25.17 M requests of 64 B, random 256-B slices 2 requests each).  Entries carry the hash of the kernel sources.
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
READ_REQ_BYTES, WRITE_REQ_BYTES = 128, 64


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "dl::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, \
        {k: max(len(v) for v in cs.values()) for k, cs in acc.items()}


def main():
    from bench import kernel_source_hash
    a, n = per_kernel(sys.argv[1])
    b, _ = per_kernel(sys.argv[2])
    out_path, key = sys.argv[3], sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    kernels = {}
    for k in sorted(a):
        c = {**a[k], **b.get(k, {})}
        rd, wr = c.get("TCC_READ_sum", 0.0), c.get("TCC_WRITE_sum", 0.0)
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        kernels[k] = dict(launches=n[k], req=c.get("TCC_REQ_sum", 0.0), read=rd, write=wr, hit=hit, miss=miss,
                          hit_rate=hit / (hit + miss) if hit + miss > 0 else None,
                          l2_bytes=rd * READ_REQ_BYTES + wr * WRITE_REQ_BYTES)
    try:
        table = json.load(open(out_path))
    except (OSError, ValueError):
        table = {}
    table[key] = dict(kernel_source_hash=kernel_source_hash(), source=note,
                      units=dict(read_request_bytes=READ_REQ_BYTES, write_request_bytes=WRITE_REQ_BYTES,
                                 calibration="profiles/r3_pmc_calibration.txt"), kernels=kernels)
    json.dump(table, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        hr = "n/a" if v["hit_rate"] is None else f"{v['hit_rate']:.3f}"
        print(f"{k:60s} read {v['read'] / 1e6:9.2f} M  write {v['write'] / 1e6:8.2f} M  L2 bytes {v['l2_bytes'] / 1e6:9.1f} MB  hit rate {hr}")


if __name__ == "__main__":
    main()
