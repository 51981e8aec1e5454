"""Builds ``deepctr_amd/lib/libdctr_hip.so`` — the C-ABI HIP library (include/dctr.h) — for gfx950.

    python -m deepctr_amd.build [--force]

Plain ``hipcc --offload-arch=gfx950`` per translation unit, then one shared link.  No torch
cpp_extension, no hipify, no other targets.  The ``.so`` is built in-tree (git-ignored, but it
travels with the repo snapshot to the GPU box).  The link deliberately carries no rpath to
/opt/rocm: at run time the HIP runtime that PyTorch has already loaded (``libamdhip64.so.7``
under ``torch/lib``) satisfies the library's ``NEEDED`` entry, so kernels and torch streams live in
ONE runtime (see DESIGN.md "HIP runtime duplication").
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "_obj")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libdctr_hip.so")

SOURCES = [
    "abi.cpp",
    "host_pack.cpp",
    "hash_kernels.hip",
    "embed_kernels.hip",
    "mlp_kernels.hip",
    "mlp_kernels_rt1.hip",
    "mlp_kernels_rt2.hip",
    "mlp_kernels_rt4.hip",
    "mlp_kernels_ring.hip", "mlp_kernels_wide.hip", "mlp_kernels_layered.hip",
    "stream_kernels.hip",
    "chain_kernels.hip",
    "chain_kernels_r2w8_m42.hip",
    "chain_kernels_r2w8_m41.hip",
    "chain_kernels_r2w8_m22.hip",
    "chain_kernels_r2w8_m21.hip",
    "chain_kernels_r2w4_m42.hip",
    "chain_kernels_r2w8_m42_x.hip",
    "chain_kernels_r2w8_m42_q.hip",
    "chain_kernels_r2w8_m42_t.hip",
    "chain_kernels_r2w8_m42_w.hip",
    "chain_kernels_r2w8_m42_r.hip",
    "chain_kernels_r2w8_m42_p.hip",
    "interaction_kernels.hip",
    "cin_kernels.hip",
    "cin_bwd_kernels.hip",
    "din_kernels.hip",
    "din_chain_kernels.hip",
    "gemm_kernels.hip",
    "mlp_bwd_kernels.hip",
    "train_kernels.hip",
]

ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include"), "-I", SRC]
if os.environ.get("DCTR_BUILD_LAB") == "1":      # lab build: the DCTR_* A/B environment switches are compiled in (dctr_common.h)
    CFLAGS.append("-DDCTR_LAB")


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (no fallback exists)")
    return exe


def _newer(path, deps):
    if not os.path.exists(path):
        return False
    t = os.path.getmtime(path)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(ROOT, "include", "dctr.h"))
    objs = []
    procs = []
    for name in SOURCES:
        src = os.path.join(SRC, name)
        if not os.path.exists(src):
            raise RuntimeError("missing source " + src)
        obj = os.path.join(OBJ, os.path.splitext(name)[0] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            cmd = [hipcc] + CFLAGS + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print("[dctr build]", " ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd)))
    failed = [n for n, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or not _newer(LIB, objs):
        # plain g++ link (objects are non-RDC, each carries its own fatbin + registration ctor):
        # hipcc's --hip-link would add RUNPATH=/opt/rocm-*/lib, which could pull a second HIP runtime
        # next to the one PyTorch ships; without it NEEDED libamdhip64.so.7 binds to the loaded one.
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib")
        cmd = ["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-L" + rocm_lib, "-lamdhip64", "-pthread"]
        if verbose:
            print("[dctr build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
