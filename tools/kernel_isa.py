#!/usr/bin/env python3
"""usage: tools/kernel_isa.py <file.hip> <mangled-name substring> [--dump]  -> instruction counts per basic block of one
kernel (VALU / SALU / LDS / VMEM), from hipcc -S for gfx950.  --dump prints the block bodies."""
import collections
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
asm = "/tmp/kernel_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Iinclude", "-Idisenlink_amd/csrc",
                "-S", "--cuda-device-only", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if re.match(r"^\.LBB\S+:", t):
        blocks.append((name, cur))
        name, cur = t.split(":")[0], []
    elif t and not t.startswith((";", ".")):
        cur.append(t)
blocks.append((name, cur))
tot = collections.Counter()
for name, ops in blocks:
    c = collections.Counter("valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_")
                            else "vmem" if o.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other" for o in ops)
    tot += c
    if len(ops) >= 8:
        print(f"{name:12s} ops {len(ops):4d}  valu {c['valu']:4d}  salu {c['salu']:4d}  lds {c['lds']:3d}  vmem {c['vmem']:3d}")
    if "--dump" in sys.argv:
        for o in ops:
            print("      ", o.split(";")[0].rstrip())
print("total", dict(tot))
