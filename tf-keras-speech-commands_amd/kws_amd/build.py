"""Build libkws_hip.so for gfx950 with hipcc (explicit, in-tree; no JIT cache)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libkws_hip.so")


def build(verbose=False, jobs=None):
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
