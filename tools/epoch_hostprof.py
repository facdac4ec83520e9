"""cProfile of the eager training loop on a small graph (host time per epoch): python tools/epoch_hostprof.py [dataset]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
name = sys.argv[1] if len(sys.argv) > 1 else "chameleon"
dev = torch.device("cuda:0")
sg = synthetic_graph(name, seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
print("native ops:", os.environ.get("DL_NATIVE_OPS", "1"))
run_link_prediction(model, x, run, epochs=5, lr=1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_link_prediction(model, x, run, epochs=200, lr=1e-4)
torch.cuda.synchronize()
print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per eager epoch")
pr = cProfile.Profile()
pr.enable()
run_link_prediction(model, x, run, epochs=200, lr=1e-4)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
