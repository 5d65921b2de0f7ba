"""In-tree build of libhw_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libhw_hip.so")
SOURCES = ["hw_engine.hip", "hw_kernels.hpp", "hw_model.hpp", "hw_stream_guard.hpp", "hw_fleet.hpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wshadow", "-fno-slp-vectorize", "-fPIC", "-shared"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps += [os.path.join(_HERE, "..", "include", h) for h in ("hw_mc.h", "hw_mc_diag.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, "hw_engine.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
