"""TEST INFRASTRUCTURE ONLY -- build recipe for the oracle's C restatement.

``python oracle/build.py`` compiles ``oracle/hfem_oracle.c`` with gcc into
``oracle/libhfem_oracle.so`` (git-ignored, travels to the GPU box with gpurun).
``-ffp-contract=off`` keeps gcc from fusing a*b+c so the closed forms round like
the reference's separate ATen mul/add passes.

There is no ``oracle/_ref`` build: the reference is pure Python (no C/C++ sources
to compile); it is run directly in the build container by
``tests/golden/make_golden.py`` instead.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hfem_oracle.c")
OUT = os.path.join(HERE, "libhfem_oracle.so")


def build(force: bool = False) -> str:
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c11",
           "-Wall", "-Wextra", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
