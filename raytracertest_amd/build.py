"""Build librt_mi355x.so in-tree (hipcc --offload-arch=gfx950; cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "librt_mi355x.so")


def build(force=False, verbose=False):
    cmd = ["make", "-C", CSRC, "all", "cli", "e2e"]
    if force:
        cmd.append("-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(cmd, check=True, stdout=out)
    if not os.path.exists(LIB):
        raise RuntimeError("hipcc build did not produce " + LIB)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
