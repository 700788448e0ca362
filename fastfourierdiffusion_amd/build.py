"""Build libffd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(verbose: bool = False, jobs: int = 8) -> str:
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError(f"building libffd.so failed (exit {res.returncode})")
    out = os.path.join(HERE, "libffd.so")
    if not os.path.exists(out):
        raise RuntimeError("make succeeded but libffd.so is missing")
    return out


if __name__ == "__main__":
    print(build(verbose=True))
