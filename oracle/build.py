"""Build recipe for the C part of the oracle (TEST INFRASTRUCTURE).

``python -m oracle.build`` (also called by ``__graft_entry__.build()``)
compiles ``oracle/rvq_exact.c`` into ``oracle/_build/librvq_exact.so``.

The reference is pure Python with no C/C++ sources, so there is nothing to
compile into ``oracle/_ref/`` -- the reference itself is pinned through the
golden vectors of ``tests/golden/`` instead.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "librvq_exact.so")
SRC = os.path.join(HERE, "rvq_exact.c")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-o", LIB, SRC]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
