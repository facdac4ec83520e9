"""Command-line driver with the flags of the reference's ``main_disentangled.py`` (:21-50), running the
pair-list pipeline on the MI355X path:

    python -m disenlink_amd.main --dataset chameleon --beta 0.7 --nfactor 5 --nhidden 512 --nembed 32 \\
        --epochs 2000 --lr 0.0001 --m 5 --run 10 [--data-root /path/to/reference/data | --data-file ds.npz]

Differences from the reference script, on purpose: runs are seeded (``--seed`` + run index; the reference
seeds nothing on the CPU path, SURVEY.md §0 finding 5); the split / masks / loss / AUC work on pair lists
(no ``[N,N]`` tensors); flags the reference parses but never uses (``--weight_decay``, ``--nfeat``,
``--loss_weight``, ``--debug``, ``--layer`` other than 1) are accepted and ignored the same way.
Without data files a seeded synthetic stand-in of the named dataset is used (``disenlink_amd.data``).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--debug", action="store_true", default=False)
    p.add_argument("--no-cuda", action="store_true", default=False)
    p.add_argument("--seed", type=int, default=18)
    p.add_argument("--lr", type=float, default=0.0001)
    p.add_argument("--beta", type=float, default=0.9)
    p.add_argument("--nfactor", type=int, default=3)
    p.add_argument("--weight_decay", type=float, default=5e-4)
    p.add_argument("--nfeat", type=int, default=128)
    p.add_argument("--nhidden", type=int, default=512)
    p.add_argument("--nembed", type=int, default=32)
    p.add_argument("--epochs", type=int, default=2000)
    p.add_argument("--temperature", type=int, default=1)
    p.add_argument("--dataset", type=str, default="chameleon")
    p.add_argument("--sub_dataset", type=str, default="Amherst41")
    p.add_argument("--run", type=int, default=10)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--m", type=int, default=5)
    p.add_argument("--save", type=int, default=0)
    p.add_argument("--loss_weight", type=int, default=20)
    p.add_argument("--layer", type=int, default=1)
    p.add_argument("--miniid", type=int, default=9)
    # extensions
    p.add_argument("--data-root", type=str, default=None, help="directory laid out like the reference's data/")
    p.add_argument("--data-file", type=str, default=None, help="binary dataset written by datasets.save_binary")
    p.add_argument("--synthetic", action="store_true", help="seeded synthetic stand-in of --dataset")
    p.add_argument("--table-dtype", choices=["f32", "bf16"], default="f32")
    p.add_argument("--no-graph", action="store_true",
                   help="launch every epoch from Python instead of replaying it from a captured HIP graph")
    p.add_argument("--graph", action="store_true",
                   help="replay every epoch from a captured HIP graph.  Default: replayed when the compiled binding "
                        "(libdisenlink_torch.so) is absent, eager when it is present — with it and the end-of-epoch "
                        "bookkeeping on the device the eager loop is gapless and 3-10 %% faster than the replay")
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--gpus", type=int, default=1,
                   help="row-shard every run over this many GPUs of the node (one process per GPU, RCCL): started "
                        "plainly, the command launches its ranks itself (disenlink_amd/launch.py)")
    return p


def load_dataset(args):
    from . import datasets
    from .data import SPECS, synthetic_graph
    if args.data_file:
        return datasets.load_binary(args.data_file)
    root = args.data_root
    if root and not args.synthetic:
        name = args.dataset
        if name in ("chameleon", "squirrel", "crocodile"):          # main_disentangled.py:73-81, 97-108
            npz = os.path.join(os.path.dirname(root.rstrip("/")), "data_pre_false", name, "raw", f"{name}.npz")
            if not os.path.exists(npz):
                npz = os.path.join(root, name, "raw", f"{name}.npz")
            return datasets.load_npz(npz, name)
        if name in ("cora", "citeseer", "pubmed"):                  # :117-123
            return datasets.load_planetoid(os.path.join(root, name, "raw"), name)
        if name == "fb100":                                         # :62-64, 109-113
            return datasets.load_fb100(os.path.join(root, "facebook100", args.sub_dataset + ".mat"), args.sub_dataset)
        if name == "twitch-e":                                      # :62-64, 109-116
            return datasets.load_twitch(os.path.join(root, "twitch", args.sub_dataset), args.sub_dataset)
        if name in ("texas", "wisconsin", "cornell"):               # :69-71, 91-96
            return datasets.load_webkb(os.path.join(root, name, "raw"), name)
        if name == "photo":                                         # :66-68, 85-90
            return datasets.load_amazon_npz(os.path.join(root, "Photo", "raw", "amazon_electronics_photo.npz"))
        if name == "deezer-europe":                                 # :58-59, 109-113
            return datasets.load_deezer(os.path.join(root, "deezer-europe.mat"))
        if name == "year":                                          # :124-129 (the arxiv-year minis, --miniid 0..9)
            return datasets.load_arxiv_year_mini(os.path.join(os.path.dirname(root.rstrip("/")), "mini", f"year{args.miniid}.pt"),
                                                 f"year{args.miniid}")
        raise SystemExit(f"no loader for --dataset {name} (ogbn-proteins / the full arxiv-year / yelp-chi need the ogb "
                         f"download layout: convert with datasets.save_binary and pass --data-file)")
    key = {"fb100": "penn94", "snap-patents": "snap_patents"}.get(args.dataset, args.dataset)
    if key not in SPECS:
        raise SystemExit(f"no data for --dataset {args.dataset}: give --data-root / --data-file")
    sg = synthetic_graph(key, seed=args.seed)
    return datasets.LinkDataset(f"{key}-synthetic", sg.features(), sg.src, sg.dst)


def save_results(args, result) -> None:
    """--save 1: append the runs' summary line to performance/<dataset>_..._nfactor.csv (main_disentangled.py:225-246)."""
    os.makedirs("performance", exist_ok=True)
    sub = f"{args.sub_dataset}" if args.dataset in ("twitch-e", "fb100") else ""
    with open(f"performance/{args.dataset}_{sub}disentangle_nfactor.csv", "a+") as f:
        f.write(f"{result.mean():.3f} ± {result.std():.3f},{result},beta {args.beta},temperature {args.temperature},"
                f"nfactor {args.nfactor},nhidden {args.nhidden},nembed {args.nembed},dataset {args.dataset},"
                f"run {args.run},epochs {args.epochs},lr {args.lr},m {args.m}\n")


def main_sharded(args):
    """--gpus N: the same runs with the graph's rows, the feature rows and the pair lists sharded over N ranks
    (train.run_link_prediction_sharded).  Every rank loads the dataset and draws the SAME seeded split (host arrays);
    it keeps its own feature rows only.  Rank 0 prints."""
    import torch.distributed as dist
    from .model import Disentangle
    from .splits import make_link_split
    from .train import prepare_run_sharded, run_link_prediction_sharded
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearse = bool(os.environ.get("DL_REHEARSE_ON_ONE_GPU"))      # all ranks on cuda:0, gloo collectives: functional only
    device = torch.device("cuda", 0 if rehearse else local)
    torch.cuda.set_device(device)
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    say = (lambda *a: print(*a, flush=True)) if rank == 0 and not args.quiet else (lambda *a: None)
    try:
        ds = load_dataset(args)
        say(args)
        say(f"dataset {ds.name}: N={ds.n_nodes} F={ds.x.shape[1]} edge rows={ds.src.size}; row-sharded over {world} GPUs")
        tdt = torch.bfloat16 if args.table_dtype == "bf16" else torch.float32
        row_bytes = args.nfactor * args.nembed * (2 if args.table_dtype == "bf16" else 4)
        result = []
        for run in range(args.run):
            say("run:", run)
            split = make_link_split(ds.src, ds.dst, ds.n_nodes, m=args.m, seed=args.seed + run)
            prepared = prepare_run_sharded(split, rank, world, device, row_bytes=row_bytes)
            r0, r1 = prepared.shard.local_real_rows()
            x_loc = torch.from_numpy(np.ascontiguousarray(ds.x[r0:r1])).to(device)
            torch.manual_seed(args.seed + run)                       # identical replicas
            model = Disentangle(ds.x.shape[1], args.nhidden, args.nembed, nfactor=args.nfactor, beta=args.beta,
                                t=args.temperature, table_dtype=tdt).to(device)
            res = run_link_prediction_sharded(model, x_loc, prepared, epochs=args.epochs, lr=args.lr,
                                              log=say if not args.quiet else None)
            say("test auc:", res.test_auc)
            result.append(res.test_auc)
        result = np.array(result)
        if rank == 0:
            print("final", result.mean(), result.std(), flush=True)
            if args.save == 1:
                save_results(args, result)
        return result
    finally:
        dist.destroy_process_group()


def _use_graph(args) -> bool:
    """--graph / --no-graph, else by what is faster: the replayed epoch saves host time per launch, which only matters when
    the launches go through the Python operators (chameleon 0.38 vs 0.76 ms); through the compiled binding, with early
    stopping on the device (early_stop.py), the host runs ahead of the GPU and the eager loop has neither the replay's
    per-node dispatch cost nor a gap between epochs (squirrel 0.837 vs 0.860 ms, chameleon 0.313 vs 0.331, cora-sized 0.665 vs
    0.707: profiles/r5z_*)."""
    if args.graph or args.no_graph:
        return bool(args.graph)
    from . import native
    return not native.available()


def main(argv=None):
    args = build_parser().parse_known_args(argv)[0]                 # unknown tokens ignored, like :50
    if args.layer != 1:
        raise SystemExit("only --layer 1 exists in the reference (main_disentangled.py:147-148)")
    if args.gpus > 1:
        from .launch import launch_ranks, under_launcher
        if not under_launcher():                                    # BEFORE any GPU call: the parent starts and waits
            entry = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_entry.py")
            import json
            raise SystemExit(launch_ranks(args.gpus, [entry], [], env_extra={
                "DL_MAIN_ARGV": json.dumps(list(sys.argv[1:] if argv is None else argv))}))
        if args.no_cuda or not torch.cuda.is_available():
            raise SystemExit("disenlink_amd runs on the GPU only (libdisenlink_hip.so has no CPU fallback)")
        return main_sharded(args)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("disenlink_amd runs on the GPU only (libdisenlink_hip.so has no CPU fallback)")
    from .model import Disentangle
    from .splits import make_link_split
    from .train import prepare_run, run_link_prediction
    device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(device)
    ds = load_dataset(args)
    if not args.quiet:
        print(args)
        print(f"dataset {ds.name}: N={ds.n_nodes} F={ds.x.shape[1]} edge rows={ds.src.size}")
    x = torch.from_numpy(ds.x).to(device)
    tdt = torch.bfloat16 if args.table_dtype == "bf16" else torch.float32
    result = []
    for run in range(args.run):
        if not args.quiet:
            print("run:", run)
        split = make_link_split(ds.src, ds.dst, ds.n_nodes, m=args.m, seed=args.seed + run)
        prepared = prepare_run(split, device, row_bytes=args.nfactor * args.nembed * (2 if args.table_dtype == "bf16" else 4))
        torch.manual_seed(args.seed + run)
        model = Disentangle(x.shape[1], args.nhidden, args.nembed, nfactor=args.nfactor, beta=args.beta,
                            t=args.temperature, table_dtype=tdt).to(device)
        res = run_link_prediction(model, x, prepared, epochs=args.epochs, lr=args.lr,
                                  log=None if args.quiet else print, use_graph=_use_graph(args))
        if not args.quiet:
            print("test auc:", res.test_auc)
        result.append(res.test_auc)
    result = np.array(result)
    print("final", result.mean(), result.std())
    if args.save == 1:                                              # :225-246
        save_results(args, result)
    return result


if __name__ == "__main__":
    main(sys.argv[1:])
