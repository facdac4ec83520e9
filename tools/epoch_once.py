import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel"
dev = torch.device("cuda:0")
sg = synthetic_graph(name, seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
run_link_prediction(model, x, run, epochs=40, lr=1e-4, use_graph=os.environ.get("DL_EPOCH_GRAPH", "0") == "1")
torch.cuda.synchronize()
