"""Build libvispec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vispec_hip.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "kernels.h"), os.path.join(HERE, "csrc", "tree_kernels.h"), os.path.join(HERE, "csrc", "gemm_wide.h"), os.path.join(HERE, "csrc", "gemm_c8.h"),
        os.path.join(HERE, "csrc", "gemm_prefill.h"), os.path.join(HERE, "csrc", "wgclock.h"),
        os.path.join(os.path.dirname(HERE), "include", "vispec_hip.h")]
OUT = os.path.join(HERE, "libvispec_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", SRC, "-o", OUT]
    cmd[1:1] = os.environ.get("VISPEC_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DVISPEC_W_NT=0)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
