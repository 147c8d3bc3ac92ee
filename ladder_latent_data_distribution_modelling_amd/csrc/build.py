"""Builds libladder_hip.so (gfx950 code object + host launchers) in-tree with hipcc.

    python -m ladder_latent_data_distribution_modelling_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU present; the .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["igemm.hip", "convsplit.hip", "convf32.hip", "convf32s.hip", "upproj.hip", "densef32.hip", "densesplit.hip", "convrgb.hip", "norm.hip", "elbo.hip", "hostutil.hip", "vbgmm.hip"]
LIB = os.path.join(HERE, "libladder_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, h) for h in ("common.h", "split16.h", "filterbank.h", "convf32.h")] + [
                                                       os.path.join(ROOT, "include", "ladder_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
               "-I", os.path.join(ROOT, "include"), "-I", HERE, "-c", os.path.join(HERE, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
