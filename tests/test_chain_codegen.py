"""Code-generation guard for the row-chained kernel (csrc/chain_device.h), no GPU needed (hipcc cross-compiles gfx950).

192 of a wave's 256 registers hold accumulators in layer 1; when hipcc runs out it spills accumulator quads to scratch inside
the k-loop, and every reload waits (vmcnt) behind the weight DMA of the step: the kernel still computes the right numbers,
20 % slower (round 2: 666 -> 788 us per 262,144 rows).  So the build is checked: every instantiation of the throughput shape
must need no scratch at all, and must fit two waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CHAIN_UNITS = [("chain_kernels_r2w8_m42.hip", 256), ("chain_kernels_r2w8_m41.hip", 256), ("chain_kernels_r2w8_m22.hip", 256),
               ("chain_kernels_r2w8_m21.hip", 256), ("chain_kernels_r2w4_m42.hip", 512),
               ("chain_kernels_r2w8_m42_x.hip", 256), ("chain_kernels_r2w8_m42_q.hip", 256), ("chain_kernels_r2w8_m42_t.hip", 256),
               ("chain_kernels_r2w8_m42_w.hip", 256), ("chain_kernels_r2w8_m42_p.hip", 256)]


def test_chain_kernel_needs_no_scratch(tmp_path):
    """Every translation unit of the row-chained kernel (compiled concurrently: ~1 minute each)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    procs = []
    for tu, max_vgprs in CHAIN_UNITS:
        src = os.path.join(ROOT, "deepctr_amd", "csrc", tu)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
               "-I", os.path.join(ROOT, "deepctr_amd", "csrc"), "-x", "hip", "--cuda-device-only", "-c", src,
               "-o", str(tmp_path / (tu + ".o")), "-Rpass-analysis=kernel-resource-usage"]
        procs.append((tu, max_vgprs, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)))
    for tu, max_vgprs, proc in procs:
        out = proc.communicate()[0]
        assert proc.returncode == 0, out[-2000:]
        names = re.findall(r"Function Name: (\S+)", out)
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out)]
        vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", out)]
        agprs = [int(x) for x in re.findall(r" AGPRs: (\d+)", out)]
        assert len(names) >= 4 and len(names) == len(scratch) == len(vgprs) == len(agprs), (tu, out[-2000:])
        for n, sc, v, a in zip(names, scratch, vgprs, agprs):
            assert "chain_kernel" in n
            # (the in-pass pooling instantiations keep <= 16 dwords of COLD values in scratch: stored and reloaded once per pass (outside the step loops), some only in the
            # loop over dense k-blocks 1.. — no scratch traffic in the step loops, checked on the ISA: DESIGN.md §4.1)
            assert sc <= (64 if tu.endswith("_p.hip") else 0), "%s: %s spills to scratch (%d B/lane)" % (tu, n, sc)
            assert v + a <= max_vgprs, "%s: %s needs %d registers" % (tu, n, v + a)


def _resource_usage(tmp_path, tu):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "deepctr_amd", "csrc", tu)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "deepctr_amd", "csrc"), "-x", "hip", "--cuda-device-only", "-c", src,
           "-o", str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, check=True).stdout
    names = re.findall(r"Function Name: (\S+)", out)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out)]
    vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", out)]
    assert names and len(names) == len(scratch) == len(vgprs), out[-2000:]
    return dict(zip(names, zip(scratch, vgprs)))


def test_cin_kernels_register_budget(tmp_path):
    """CIN forward (csrc/cin_kernels.hip): the inference instantiation must not pay for the training-mode stores of the layer
    activations (as a run-time branch of one kernel they cost it 580 B of scratch per lane and 25 % of its speed); the z-free
    backward kernels (csrc/cin_bwd_kernels.hip) keep their accumulators and staging registers out of scratch and fit two
    workgroups per CU."""
    fwd = _resource_usage(tmp_path, "cin_kernels.hip")
    for name, (scratch, vgprs) in fwd.items():
        if "cin_kernel" in name:
            assert scratch <= 64, "%s: %d B of scratch per lane" % (name, scratch)
    bwd = _resource_usage(tmp_path, "cin_bwd_kernels.hip")
    seen = 0
    for name, (scratch, vgprs) in bwd.items():
        if "cin_dz_fused_kernel" in name:
            assert scratch == 0 and vgprs <= 256, (name, scratch, vgprs)
            seen += 1
        elif "cin_dw_fused_kernel" in name:
            assert scratch <= 64 and vgprs <= 256, (name, scratch, vgprs)
            seen += 1
    assert seen >= 16


def test_training_step_kernels_register_budget(tmp_path):
    """The kernels the third part of round 3 added to the training step, and the forward kernels they share templates with:
    * csrc/mlp_bwd_kernels.hip — the backward chain runs on mlp_device.h's whole-MLP machinery: no scratch, and the 16- / 32-row
      shapes must keep two 8-wave workgroups per CU (<= 128 VGPRs), as the forward kernels of mlp_kernels_rt{1,2}.hip do — the
      backward epilogue (ACT_BWD) is compiled into its own kernels so that it cannot push the inference kernels over that line;
    * csrc/gemm_kernels.hip — plain and grouped f32 MFMA GEMM: accumulators (128 VGPRs at 128 x 128 tiles) stay out of scratch;
    * csrc/interaction_kernels.hip — cross_matrix_kernel<1 / 2>: one 8-wave workgroup per CU (its LDS tiles), i.e. two waves per
      SIMD: three pipeline stages of (both row tiles') operands in registers without scratch, <= 256 VGPRs."""
    bwd = _resource_usage(tmp_path, "mlp_bwd_kernels.hip")
    seen = 0
    for name, (scratch, vgprs) in bwd.items():
        if "mlp_bwd_kernel" in name:
            seen += 1
            assert scratch == 0, (name, scratch)
            if "ILi4E" not in name:                      # RT = 1 / 2: four waves per SIMD
                assert vgprs <= 128, (name, vgprs)
    assert seen == 3
    for tu in ("mlp_kernels_rt1.hip", "mlp_kernels_rt2.hip"):
        for name, (scratch, vgprs) in _resource_usage(tmp_path, tu).items():
            if "mlp_kernel" in name:
                assert scratch == 0 and vgprs <= 128, (tu, name, scratch, vgprs)
    gemm = _resource_usage(tmp_path, "gemm_kernels.hip")
    seen = 0
    for name, (scratch, vgprs) in gemm.items():
        if "gemm_kernel" in name or "gemm_grouped_kernel" in name:
            seen += 1
            assert scratch == 0 and vgprs <= 256, (name, scratch, vgprs)
    assert seen >= 5
    inter = _resource_usage(tmp_path, "interaction_kernels.hip")
    cross = {n: v for n, v in inter.items() if "cross_matrix_kernel" in n}
    assert len(cross) == 2
    for name, (scratch, vgprs) in cross.items():
        assert scratch == 0 and vgprs <= 256, (name, scratch, vgprs)


def test_din_chain_kernels_register_budget(tmp_path):
    """csrc/din_chain_kernels.hip: sixteen waves per workgroup = four per SIMD need <= 128 VGPRs, and nothing may spill (the compacted
    row walk and the folded lookups keep their per-row state in 32-bit registers for that: round 4 saw 12 B of scratch per lane at
    embedding_dim 64 with 64-bit row offsets)."""
    usage = _resource_usage(tmp_path, "din_chain_kernels.hip")
    seen = 0
    for name, (scratch, vgprs) in usage.items():
        if "din_chain_kernel" in name:
            seen += 1
            assert scratch == 0 and vgprs <= 128, (name, scratch, vgprs)
        elif "din_prep_kernel" in name:
            assert scratch == 0, (name, scratch)
    assert seen == 30
