#!/usr/bin/env python3
"""Compile oracle/c/sparse_ref.c (the C restatement, test infrastructure) into oracle/_build/libdl_oracle.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "sparse_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libdl_oracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O2", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", OUT, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
