"""Start the ranks of a multi-GPU run as CHILD processes (one per GPU, torch.distributed.run on one node).

Used by ``bench.py --gpus N`` and ``python -m disenlink_amd.main --gpus N`` when they are started plainly, i.e. without
a launcher around them.  The calling (parent) process makes NO GPU / HIP call before or after this — it neither asks
``torch.cuda`` anything nor loads libdisenlink_hip.so — and replaces no running program (no ``os.exec*``): it starts the
children, relays their output and returns their exit code.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def under_launcher() -> bool:
    """Started by torch.distributed.run (or any launcher that sets the rendezvous variables)?"""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n_ranks: int, target: list[str], argv: list[str], cwd: str | None = None, result_marker: str | None = None,
                 env_extra: dict | None = None) -> int:
    """Run ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    <target...> <argv...>`` and wait for it.  `target`: ["script.py"] or ["-m", "package.module"].
    result_marker: when given, stdout lines of the children that start with "{" and contain the marker are taken as THE
    result (the last one wins) and printed on this process's stdout at the end, everything else goes to stderr — a caller
    that promises one JSON line on stdout keeps that promise; without a marker stdout passes through unchanged.
    -> the children's exit code (non-zero if any rank failed, or if a result was promised and none came)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # dmabuf IPC: what the host driver supports
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), *target, *argv]
    print("launching %d ranks: %s" % (n_ranks, " ".join(cmd)), file=sys.stderr, flush=True)
    if result_marker is None:
        return subprocess.call(cmd, env=env, cwd=cwd)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=cwd)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{") and result_marker in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("the ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        print(line, flush=True)
    return rc


class stdout_to_stderr:
    """File-descriptor-level redirect of stdout into stderr for the duration of a block: communication backends print
    connection banners from C++ on fd 1 ("[Gloo] Rank 0 is connected to ...", RCCL's version banner), and a rank that
    promises ONE JSON line on stdout has to keep them off it."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False
