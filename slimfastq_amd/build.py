"""Build the gfx950 shared library (C ABI + HIP kernels) and the CLI, in-tree.

    python -m slimfastq_amd.build          # or: from slimfastq_amd.build import build; build()

hipcc cross-compiles for gfx950 without a GPU present.  Outputs (git-ignored, but they travel to the
GPU box with the tree):  slimfastq_amd/libslimfastq_amd.so, slimfastq_amd/bin/slimfastq-amd
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libslimfastq_amd.so")
CLI = os.path.join(HERE, "bin", "slimfastq-amd")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

LIB_SOURCES = ["api.cpp", "synth.cpp", "frame.hip", "models_l.hip", "decode_l.hip", "decode_w.hip", "models_w.hip", "models_k.hip", "chains.hip", "gm.hip", "exc.hip", "prior.hip", "container.cpp", "archive_api.cpp"]
CLI_SOURCES = ["cli.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=off"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "slimfastq_amd.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in LIB_SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src + ".o")
        if force or _newer(obj, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd.insert(1, "-x"); cmd.insert(2, "hip")
            _run(cmd)
        objs.append(obj)
    if force or _newer(LIB, objs):
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-lpthread"])
    cli_srcs = [os.path.join(CSRC, s) for s in CLI_SOURCES]
    if all(os.path.exists(s) for s in cli_srcs):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        if force or _newer(CLI, cli_srcs + hdrs + [LIB]):
            _run([HIPCC, "-O2", "-std=c++17", "-Wall", "-o", CLI] + cli_srcs +
                 ["-L" + HERE, "-lslimfastq_amd", "-Wl,-rpath,$ORIGIN/.."])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
