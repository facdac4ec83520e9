#!/usr/bin/env python3
"""usage: tools/kernel_regs.py <file.hip> [regex] [--spills]  -> one line per kernel: VGPRs, scratch bytes/lane,
spilled VGPRs, waves/SIMD, LDS (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU)."""
import re
import subprocess
import sys

src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else re.compile(".")
only_spills = "--spills" in sys.argv
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Iinclude",
                      "-Idisenlink_amd/csrc", *__import__("os").environ.get("DL_CXXFLAGS","").split(), "-c", src, "-o", "/tmp/kr.o", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0] if k != "VGPRs Spill" else "Spill"] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("dl::fast::", "").replace("unsigned short", "bf16")
    if not pat.search(n) or (only_spills and r.get("Spill") == "0" and r.get("ScratchSize") == "0"):
        continue
    print(f"{n:58s} vgpr {r.get('VGPRs'):>4} scratch {r.get('ScratchSize'):>4} spill {r.get('Spill'):>3} "
          f"waves {r.get('Occupancy')} lds {r.get('LDS')} sgpr {r.get('TotalSGPRs')}")
