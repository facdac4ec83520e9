"""Timeline of one workgroup of nodes_contract_planes_kernel (kernel B of the projection backward: dW1 = dhid^T . x from bf16
planes) from a DIAGNOSTIC build: tools/build_variant.py stampsB dl_project_bwd.hip "-DDL_PROJB_STAMPS=100".
usage: DL_LIB_PATH=variants/libdisenlink_hip_stampsB.so python tools/projb_stamps.py [N F K nhid d]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import _lib, ops
N, F, K, nhid, d = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (5201, 128, 8, 512, 64)
x = torch.randn(N, F, device="cuda"); W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5; b1 = torch.randn(K, nhid, device="cuda") * 0.1
W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5; b2 = torch.randn(K, d, device="cuda") * 0.1
dZ = torch.randn(N, K, d, device="cuda")
Z, hid = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True)
for _ in range(20):
    ops.project_bwd(x, W1, b1, W2, dZ, hid=hid)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 1024)()
lib.dl_debug_read_stamps_b.restype = C.c_int
assert lib.dl_debug_read_stamps_b(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(2, 512)
names = {1: "start", 2: "prologue done (2 fetches, 1 stash, barrier)", 10: "chunk top", 11: "LDS reads + 12 MFMAs issued", 12: "stash(c+1) done (waits its loads)",
         13: "fetch(c+2) + 12 MFMAs issued", 14: "after barrier", 20: "accumulators stored"}
for w in range(2):
    n = int(a[w, 511]); t = a[w, :n] & np.uint64((1 << 56) - 1); code = (a[w, :n] >> np.uint64(56)).astype(int)
    print(f"--- wave {w * 2}: {n} stamps, total {int(t[-1] - t[0])} cycles (s_memtime: 100 MHz ticks x ?)")
    prev = t[0]; c = -1
    for i in range(n):
        if code[i] == 10: c += 1
        print(f"  chunk {c:2d} {names.get(code[i], code[i]):48s} +{int(t[i] - prev):6d}   t={int(t[i] - t[0]):7d}")
        prev = t[i]
