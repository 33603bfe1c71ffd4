"""Build libccst_hip.so (gfx950 only) in-tree with hipcc.

    python -m ccst_amd.build [--force]

The .so lands next to the sources (ccst_amd/csrc/libccst_hip.so); it is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libccst_hip.so")
STAMP = os.path.join(CSRC, ".build_stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
            [os.path.join(os.path.dirname(HERE), "include", "ccst_hip.h")]:
        with open(f, "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    objs = []
    procs = []
    for src in _sources():
        obj = src + ".o"
        cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    bad = [s for s, p in procs if p.wait() != 0]
    if bad:
        raise RuntimeError("hipcc failed for: %s" % ", ".join(bad))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
