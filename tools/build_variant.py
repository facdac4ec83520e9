#!/usr/bin/env python3
"""Build a VARIANT of libdisenlink_hip.so for same-box A/B runs: tools/build_variant.py <name> <source.hip> "<extra hipcc flags>"
-> variants/libdisenlink_hip_<name>.so (the named source — a file of disenlink_amd/csrc, or a modified copy given with its path,
e.g. variants/src/dl_train.hip — recompiled with the flags, every other object as built).
Use with DL_LIB_PATH=variants/libdisenlink_hip_<name>.so.  Cross-compiles here; the .so travels to the GPU box."""
import os, subprocess, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import build as B
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3].split()
B.build()
out = os.path.join(B.ROOT, "variants")
os.makedirs(out, exist_ok=True)
stem = os.path.basename(src)[:-4]
obj = os.path.join(out, f"{stem}_{name}.o")
# the source: the product file of that name, or (an experiment copy, e.g. variants/src/dl_train.hip) the path as given
path = src if os.path.dirname(src) and os.path.exists(src) else os.path.join(B.CSRC, os.path.basename(src))
subprocess.run([B.HIPCC, *B.FLAGS, *flags, "-c", path, "-o", obj], check=True)
objs = [o for o in sorted(glob.glob(os.path.join(B.OBJ, "*.o"))) if os.path.basename(o) != stem + ".o"] + [obj]
lib = os.path.join(out, f"libdisenlink_hip_{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", lib, *objs], check=True)
print(lib)
