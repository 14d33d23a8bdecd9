"""Build the in-tree shared libraries.

  lib/libindelope_hip.so  hipcc --offload-arch=gfx950 over csrc/*.hip   (the product)
  lib/libihp_synth.so     g++ over csrc/synth.cpp                        (input generator)

`python -m indelope_amd.build [--force]`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INC = os.path.join(HERE, "..", "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".cuh"))]
    hdrs.append(os.path.join(INC, "indelope_hip.h"))
    # generator
    synth = os.path.join(LIBDIR, "libihp_synth.so")
    src = os.path.join(CSRC, "synth.cpp")
    if force or _stale(synth, [src]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", synth, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # product
    lib = os.path.join(LIBDIR, "libindelope_hip.so")
    hips = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if hips and (force or _stale(lib, hips + hdrs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", INC, "-I", CSRC,
               "-Wall", "-Wno-unused-function", "-ffp-contract=off",
               "-o", lib] + hips + ["-ldl"]      # dlopen: librccl.so.1 is opened by the first ihp_dist_* call, not linked
        if os.environ.get("IHP_SAVE_TEMPS"):
            cmd[1:1] = ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib, synth


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
