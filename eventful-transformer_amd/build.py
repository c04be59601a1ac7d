#!/usr/bin/env python3
"""Build libevt_hip.so (gfx950 code object + C ABI) in-tree with hipcc.  No cmake, no torch extension
machinery: the library has a plain C ABI and links only against the HIP runtime.  Each csrc/*.hip is
compiled to its own object (in parallel, cached under csrc/.obj by source + header + flags hash) and
the objects are linked into one shared library."""
import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
OUT = os.path.join(HERE, "eventful_transformer", "libevt_hip.so")
ABI_HEADER = os.path.join(HERE, "..", "include", "evt_abi.h")
ARCH = "gfx950"


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [ABI_HEADER]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in _sources() + _headers())


def _flags():
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fvisibility=hidden"] + \
        os.environ.get("EVT_HIPCC_FLAGS", "").split()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    flags = _flags()
    hdr = hashlib.sha256(b"".join(open(h, "rb").read() for h in _headers()) + " ".join(flags).encode()).hexdigest()

    def compile_one(src):
        key = hashlib.sha256(open(src, "rb").read() + hdr.encode()).hexdigest()[:20]
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + "." + key + ".o")
        if force or not os.path.exists(obj):
            for old in glob.glob(os.path.join(OBJ, os.path.basename(src)[:-4] + ".*.o")):
                os.remove(old)
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, _sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
