"""Builds libslfp_hip.so (the C-ABI library of include/slfp.h) with hipcc for gfx950.

    python -m cnns_slfp_quantization_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is written IN-TREE next to this file so that
it travels with the source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libslfp_hip.so")
SOURCES = ["codec.hip", "enc_table.hip", "conv_dw.hip", "conv_dw2.hip", "conv_dwpw.hip", "conv_dwc.hip", "conv_pw.hip", "conv_pw_codes.hip", "conv_direct.hip", "conv_dense.hip", "conv_stem_mfma.hip", "conv_stem_small.hip", "pool_codes.hip", "conv_abi.hip"]
# No fast-math: the SLFP encode needs IEEE float32 division and un-contracted rescales.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
         "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "slfp.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    extra = os.environ.get("SLFP_EXTRA_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DSLFP_NT=3)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
