"""Build libttscale.so in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libttscale.so")


def build(force: bool = False) -> str:
    csrc = os.path.join(HERE, "csrc")
    cmd = ["make", "-j4", "-C", csrc] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if not os.path.exists(LIB):
        raise RuntimeError("libttscale.so was not produced by %s" % " ".join(cmd))
    return LIB


if __name__ == "__main__":
    print(build())
