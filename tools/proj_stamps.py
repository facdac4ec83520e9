"""Timeline of one workgroup of project2_fwd_kernel from a DIAGNOSTIC build (tools/build_variant.py stamps dl_project.hip
"-DDL_PROJ_STAMPS=300"): s_memtime at the phase boundaries of waves 0 and 4 (two waves of one SIMD).
usage: DL_LIB_PATH=variants/libdisenlink_hip_stamps.so python tools/proj_stamps.py [N F K nhid d]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import _lib, ops
N, F, K, nhid, d = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (5201, 128, 8, 512, 64)
x = torch.randn(N, F, device="cuda"); W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5; b1 = torch.randn(K, nhid, device="cuda") * 0.1
W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5; b2 = torch.randn(K, d, device="cuda") * 0.1
for _ in range(20):
    ops.project_fwd(x, W1, b1, W2, b2)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 1024)()
lib.dl_debug_read_stamps.restype = C.c_int
assert lib.dl_debug_read_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(2, 512)
names = {1: "start", 2: "prologue loads issued+stashed", 3: "after prologue barrier", 10: "step top", 11: "LDS operands read issued", 12: "12 MFMAs issued (block 0)",
         13: "stash(s+1) done [LDS-DMA build: nothing to stash — bias / W2 requests]", 14: "fetch(s+2) issued [LDS-DMA build: no-op]", 15: "12 MFMAs issued (block 1)", 16: "before barrier", 17: "after barrier", 20: "bias+ReLU done",
         21: "planes split (layer 2 ready)", 22: "layer-2 MFMAs issued", 30: "Z hand-over staged", 31: "end"}
for w in range(2):
    n = int(a[w, 511]); t = a[w, :n] & np.uint64((1 << 56) - 1); code = (a[w, :n] >> np.uint64(56)).astype(int)
    print(f"--- wave {w * 4}: {n} stamps, total {int(t[-1] - t[0])} cycles")
    prev = t[0]; step = -1
    for i in range(n):
        if code[i] == 10: step += 1
        print(f"  step {step:2d} {names.get(code[i], code[i]):34s} +{int(t[i] - prev):6d}   t={int(t[i] - t[0]):7d}")
        prev = t[i]
