#!/usr/bin/env python3
"""Build libevt_hip.so (gfx950 code object + C ABI) in-tree with hipcc.  No cmake, no torch extension
machinery: the library has a plain C ABI and links only against the HIP runtime."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "eventful_transformer", "libevt_hip.so")
ARCH = "gfx950"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(HERE, "..", "include", "evt_abi.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-fvisibility=hidden", "-o", OUT] + os.environ.get("EVT_HIPCC_FLAGS", "").split() + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
