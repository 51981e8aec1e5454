"""CPU: the C-ABI shared library loads and exports every symbol include/dctr.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dctr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dctr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from deepctr_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        from deepctr_amd import build
        build.build(verbose=False)
    lib = _C.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libdctr_hip.so does not export %s" % s
        assert s in _C.SYMBOLS, "deepctr_amd/_C.py has no binding for %s" % s
    assert set(_C.SYMBOLS) == set(syms)
    assert lib.dctr_abi_version() == _C.ABI_VERSION == 13
    assert lib.dctr_target_arch() == b"gfx950"


def test_struct_sizes_match_the_header_layout():
    import ctypes
    from deepctr_amd import _C
    assert ctypes.sizeof(_C.FieldDesc) == 48
    assert ctypes.sizeof(_C.GatherFmArgs) == 8 * 4 + 4 * 6 + 8 * 3 + 4 * 2 + 8 * 6 + 4 * 6 + 8 * 2 + 4 * 2
    assert ctypes.sizeof(_C.LookupArgs) == 8 * 4 + 4 * 4 + 8 * 4
    assert ctypes.sizeof(_C.PoolArgs) == 8 * 8 + 4 * 6 + 8 * 4


def test_argument_errors_are_reported_without_a_gpu():
    """NULL / bad-size arguments are rejected before any launch, so this runs on the CPU-only container."""
    import ctypes
    from deepctr_amd import _C
    lib = _C.lib()
    assert lib.dctr_hash_bucket_i32(None, 4, 10, 0, None, None) == -1          # DCTR_E_NULL
    assert b"null" in lib.dctr_last_error()
    assert lib.dctr_hash_bucket_i32(None, -1, 10, 0, None, None) == -2         # DCTR_E_DIM
    assert lib.dctr_hash_bucket_i32(None, 0, 10, 0, None, None) == 0           # empty input is a no-op
    assert lib.dctr_fm_fwd(None, 4, 8, 2, 4, None, None) == -1
    assert lib.dctr_embed_gather_fm(None, None) == -1
    a = _C.GatherFmArgs(batch=4, n_fields=0, n_dense=0)
    assert lib.dctr_embed_gather_fm(ctypes.byref(a), None) == -2
    assert lib.dctr_crossnet_fwd(None, 4, 8, 8, None, None, 1, 7, None, 8, None, 0, None) == -4     # DCTR_E_ENUM
    assert lib.dctr_crossnet_workspace_bytes(429, 2, 1, None) == 2 * 429 * 432 * 4 and lib.dctr_crossnet_workspace_bytes(64, 2, 1, None) == 0
    # sibling + training entry points: same contract
    assert lib.dctr_bi_interaction_fwd(None, 4, 8, 2, 4, None, 4, None) == -1
    assert lib.dctr_bi_interaction_fwd(None, 4, 7, 2, 4, None, 4, None) == -2     # row stride smaller than F*E
    assert lib.dctr_bi_interaction_fwd(None, 0, 8, 2, 4, None, 4, None) == 0
    assert lib.dctr_bce_grad(None, None, 4, 0, None, None, None, None) == -1
    assert lib.dctr_bce_grad(None, None, 4, 5, None, None, None, None) == -2
    assert lib.dctr_bce_grad_w(None, None, None, 4, 0, None, None, None, None) == -1
    assert lib.dctr_bce_grad_w(None, None, None, -1, 0, None, None, None, None) == -2
    assert lib.dctr_bce_grad_w(None, None, None, 0, 1, None, None, None, None) == 0
    assert lib.dctr_opt_multi(9, None, 0, 0, 0.001, 0.9, 0.999, 1e-7, 1, None) == -4
    assert b"optimizer kind 9" in lib.dctr_last_error()
    assert lib.dctr_mlp_bwd(None, None) == -1 and lib.dctr_cin_bwd(None, None) == -1 and lib.dctr_crossnet_bwd(None, None) == -1
    assert lib.dctr_embed_gather_fm_bwd(None, None) == -1 and lib.dctr_embed_pool_bwd(None, None) == -1
    # round 3: Hash over an id matrix, the library's own GEMM
    assert lib.dctr_hash_fields(None, 3, None, 8, 1, 0, 8, None, 8, 0, None) == -1
    assert lib.dctr_hash_fields(None, -1, None, 8, 1, 0, 8, None, 8, 0, None) == -2
    assert lib.dctr_hash_fields(None, 0, None, 8, 1, 0, 8, None, 8, 0, None) == 0         # nothing to hash
    assert lib.dctr_sgemm(0, 0, 4, 4, 4, None, 4, 0, None, 4, 0, 0.0, None, 4, 0, 1, None) == -1
    assert lib.dctr_sgemm(0, 0, -4, 4, 4, None, 4, 0, None, 4, 0, 0.0, None, 4, 0, 1, None) == -2
    assert lib.dctr_sgemm(0, 0, 4, 4, 4, None, 4, 0, None, 4, 0, 0.5, None, 4, 0, 1, None) == -5     # beta other than 0 / 1
    assert lib.dctr_sgemm(0, 0, 4, 4, 4, None, 4, 0, None, 4, 0, 0.0, None, 2, 0, 1, None) == -2      # ldc < m
    assert lib.dctr_sgemm(0, 0, 0, 4, 4, None, 4, 0, None, 4, 0, 0.0, None, 4, 0, 1, None) == 0       # empty product
    assert lib.dctr_embed_mlp_fwd_last_kernel() == -1                                                 # no fused launch on this thread yet
    # round 3, third part (ABI 6): CrossNet with the fused head, touched bytes, the chained DNN backward's workspace
    assert lib.dctr_crossnet_head_fwd(None, None) == -1
    fake = ctypes.c_void_p(4096)                                       # a non-NULL pointer the argument checks never dereference
    ca = _C.CrossnetArgs(x=fake, batch=4, x_stride=8, dim=8, layers=1, mode=0, kernels=fake, bias=fake, y=None, head_w=None, logit=None)
    assert lib.dctr_crossnet_head_fwd(ctypes.byref(ca), None) == -1    # neither an output nor a head
    ca.head_w = fake
    assert lib.dctr_crossnet_head_fwd(ctypes.byref(ca), None) == -1 and b"head_w and logit" in lib.dctr_last_error()
    ca.logit, ca.mode = fake, 7
    assert lib.dctr_crossnet_head_fwd(ctypes.byref(ca), None) == -4
    ca.mode, ca.x_stride = 1, 4
    assert lib.dctr_crossnet_head_fwd(ctypes.byref(ca), None) == -2    # row stride smaller than dim
    ca.batch = 0
    assert lib.dctr_crossnet_head_fwd(ctypes.byref(ca), None) == 0
    la = _C.LookupArgs(idx=fake, table=None, vocab=10, n=5, idx_is_i64=0, dim=6, hash_mode=0, out=None, out_stride=0, mask=None, status=None)
    assert lib.dctr_embed_lookup_bwd(ctypes.byref(la), fake, 8, fake, fake, None) == -2 and b"touched" in lib.dctr_last_error()
    la.n = 0
    assert lib.dctr_embed_lookup_bwd(ctypes.byref(la), fake, 8, fake, fake, None) == 0
    units = (ctypes.c_int32 * 3)(256, 128, 64)
    ba = _C.MlpBwdArgs(x=fake, batch=4096, x_stride=432, in_dim=429, n_layers=3, units=ctypes.cast(units, ctypes.c_void_p), activation=1)
    relu_ws = lib.dctr_mlp_bwd_workspace_bytes(ctypes.byref(ba))
    # chained form: dZ_l [B, units] + W_l^T + 16 slices of (K + 1) x N per layer (4-float aligned each)
    want = sum(((4096 * n + 3) // 4 * 4) + ((k * n + 3) // 4 * 4) + ((16 * (k + 1) * n + 3) // 4 * 4) for k, n in ((429, 256), (256, 128), (128, 64))) * 4
    assert relu_ws == max(want, (2 * 4096 * 429 + 8 * 429 * 256) * 4), (relu_ws, want)
    ba.activation = 4                                                  # Dice keeps the layer-by-layer form: three buffers + statistics + slices
    assert lib.dctr_mlp_bwd_workspace_bytes(ctypes.byref(ba)) == ((3 * 4096 * 429 + 2 * 429 + 3) // 4 * 4 + 8 * 429 * 256) * 4
    assert lib.dctr_mlp_bwd_workspace_bytes(None) == 0
    # round 4 (ABI 8): the attention workspace = raw scores + the list of the positions that count (chunks of 4096 rows, doubled until
    # there are <= 1024 of them) + 1024 chunk counts + the workgroups' 160-KiB LDS image; CIN over a gather
    da = _C.DinAttnArgs(batch=2048, maxlen=50, dim=64)
    assert lib.dctr_din_attn_workspace_bytes(ctypes.byref(da)) == 2048 * 50 * 4 + (25 * 4096 + 1024) * 4 + 160 * 1024
    da.batch = 100000                                                   # 5,000,000 rows: 8192-row chunks (611 of them)
    assert lib.dctr_din_attn_workspace_bytes(ctypes.byref(da)) == 5000000 * 4 + (611 * 8192 + 1024) * 4 + 160 * 1024
    da.batch = 0
    assert lib.dctr_din_attn_workspace_bytes(ctypes.byref(da)) == 0
    assert lib.dctr_cin_gather_fwd(None, None, None, None, None) == -1
    ls = (ctypes.c_int32 * 1)(32)
    ptrs = (ctypes.c_void_p * 1)(4096)
    cin = _C.CinArgs(x=None, batch=64, x_stride=0, fields=5, dim=16, n_layers=1, split_half=1, activation=1,
                     layer_size=ctypes.cast(ls, ctypes.c_void_p), filters=ctypes.cast(ptrs, ctypes.c_void_p), bias=ctypes.cast(ptrs, ctypes.c_void_p),
                     out=fake)
    ga = _C.GatherFmArgs(fields=fake, ids=fake, ids_stride_f=64, ids_stride_b=1, n_fields=5, max_dim=16, all_dim4=1, batch=64, uniform_dim=16)
    assert lib.dctr_cin_gather_fwd(ctypes.byref(cin), ctypes.byref(ga), fake, None, None) == -1 and b"come together" in lib.dctr_last_error()
    ga.n_fields = 4
    assert lib.dctr_cin_gather_fwd(ctypes.byref(cin), ctypes.byref(ga), None, None, None) == -2       # a gather of other fields
    ga.n_fields, ga.any_hash = 5, 1
    assert lib.dctr_cin_gather_fwd(ctypes.byref(cin), ctypes.byref(ga), None, None, None) == _C.E_UNSUPPORTED
    ga.any_hash, ga.uniform_dim = 0, 8
    assert lib.dctr_cin_gather_fwd(ctypes.byref(cin), ctypes.byref(ga), None, None, None) == _C.E_UNSUPPORTED
    ga.uniform_dim, cin.batch, ga.batch = 16, 0, 0
    assert lib.dctr_cin_gather_fwd(ctypes.byref(cin), ctypes.byref(ga), None, None, None) == 0       # nothing to do
    # DCN's matrix CrossNet over a gather: matrix form only, the gather's DNN-input width, plain lookups of a power-of-two width
    assert lib.dctr_crossnet_gather_head_fwd(None, None, None) == -1
    ga2 = _C.GatherFmArgs(fields=fake, ids=fake, ids_stride_f=70000, ids_stride_b=1, n_fields=5, max_dim=16, all_dim4=1, batch=70000, uniform_dim=16,
                          n_dense=3, dense=fake, dense_stride=3, dense_out_offset=80, dense_copy_cols=3)
    xa = _C.CrossnetArgs(x=None, batch=70000, x_stride=0, dim=83, layers=2, mode=0, kernels=fake, bias=fake, y=None, head_w=fake, logit=fake)
    assert lib.dctr_crossnet_gather_head_fwd(ctypes.byref(xa), ctypes.byref(ga2), None) == _C.E_UNSUPPORTED      # vector parameterization
    xa.mode, xa.dim = 1, 84
    assert lib.dctr_crossnet_gather_head_fwd(ctypes.byref(xa), ctypes.byref(ga2), None) == -2 and b"DNN-input width" in lib.dctr_last_error()
    xa.dim, ga2.uniform_dim = 83, 12
    assert lib.dctr_crossnet_gather_head_fwd(ctypes.byref(xa), ctypes.byref(ga2), None) == _C.E_UNSUPPORTED      # 12 is no power of two
    ga2.uniform_dim, ga2.any_hash = 16, 1
    assert lib.dctr_crossnet_gather_head_fwd(ctypes.byref(xa), ctypes.byref(ga2), None) == _C.E_UNSUPPORTED


def test_host_pack_columns_converts_like_numpy():
    """dctr_host_pack_columns is host code: row ranges of int32 / int64 / float32 / float64 columns,
    contiguous or strided, into a feature-major matrix of the staging dtype — same values as ndarray.astype."""
    import numpy as np
    from deepctr_amd import _C, engine
    lib = _C.lib()
    rng = np.random.RandomState(4)
    n = 200003                                    # > 3 blocks of 65536 rows
    wide = rng.rand(n, 3) * 50
    cols = [rng.randint(-5, 100000, n).astype(np.int32), rng.randint(0, 2 ** 40, n).astype(np.int64),
            (rng.rand(n) * 100).astype(np.float32), wide[:, 1], rng.randint(0, 9, 2 * n).astype(np.int64)[::2]]
    desc = engine._host_cols(cols)
    assert desc is not None and engine._host_cols([cols[0].astype(np.int16)]) is None
    for kind, dt in (("int32", np.int32), ("int64", np.int64), ("float32", np.float32)):
        for lo, m in ((0, n), (12345, 70001), (n - 3, 3), (5, 0)):
            dst = np.full((len(cols), m + 5), 7, dtype=dt)
            for threads in (1, 4):
                rc = lib.dctr_host_pack_columns(desc, len(cols), lo, m, dst.ctypes.data, m + 5, _C.HOST_KINDS[kind], threads)
                assert rc == 0, lib.dctr_last_error()
                for i, c in enumerate(cols):
                    assert np.array_equal(dst[i, :m], c[lo:lo + m].astype(dt)), (kind, lo, m, i)
                assert (dst[:, m:] == 7).all()
    assert lib.dctr_host_pack_columns(desc, len(cols), 0, 8, None, 8, 0, 1) == -1
    assert lib.dctr_host_pack_columns(desc, len(cols), 0, 8, None, 4, 0, 1) == -2        # column stride < rows
    assert lib.dctr_host_pack_columns(desc, len(cols), 0, 8, None, 8, 3, 1) == -4        # float64 is not a staging dtype


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/dctr.h is the drop-in boundary: it must compile as C99 (no C++, no torch / HIP types) and a C program must link
    against libdctr_hip.so and call it without Python in the process."""
    import os
    import shutil
    import subprocess
    from deepctr_amd import build
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "main.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "dctr.h"\n'
                   "int main(void) {\n"
                   "    dctr_gather_fm_args_t g; memset(&g, 0, sizeof g); g.batch = 4;\n"
                   "    if (dctr_abi_version() != DCTR_ABI_VERSION || DCTR_ABI_VERSION != 13 || strcmp(dctr_target_arch(), \"gfx950\") != 0) return 1;\n"
                   "    if (dctr_fm_fwd(NULL, 4, 8, 2, 4, NULL, NULL) != DCTR_E_NULL) return 2;      /* rejected before any launch */\n"
                   "    if (dctr_embed_gather_fm(&g, NULL) != DCTR_E_DIM) return 3;\n"
                   "    if (strlen(dctr_last_error()) == 0) return 4;\n"
                   "    if (dctr_host_pack_columns(NULL, 0, 0, 0, NULL, 0, DCTR_HOST_I32, 1) != DCTR_OK) return 5;\n"
                   '    puts("ok");\n    return 0;\n}\n')
    exe = tmp_path / "cabi_check"
    lib_dir = os.path.dirname(build.LIB)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                    "-o", str(exe), "-L", lib_dir, "-ldctr_hip", "-Wl,-rpath," + lib_dir], check=True)
    env = dict(os.environ)
    rocm = "/opt/rocm/lib"
    env["LD_LIBRARY_PATH"] = rocm + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_every_ctypes_mirror_has_the_layout_the_c_compiler_gives_the_header(tmp_path):
    """sizeof and the offset of every field of each struct in include/dctr.h, as gcc lays them out, against the ctypes mirrors
    in deepctr_amd/_C.py (same field order by construction: the C side is generated from the ctypes field names)."""
    import ctypes
    import os
    import shutil
    import subprocess
    from deepctr_amd import _C
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"FieldDesc": "dctr_field_t", "GatherFmArgs": "dctr_gather_fm_args_t", "PoolArgs": "dctr_pool_args_t",
             "LookupArgs": "dctr_lookup_args_t", "CinArgs": "dctr_cin_args_t", "MlpArgs": "dctr_mlp_args_t",
             "FieldGrad": "dctr_field_grad_t", "GatherFmBwdArgs": "dctr_gather_fm_bwd_args_t", "PoolBwdArgs": "dctr_pool_bwd_args_t",
             "MlpBwdArgs": "dctr_mlp_bwd_args_t", "DnnTrainLayer": "dctr_dnn_train_layer_t", "CinBwdArgs": "dctr_cin_bwd_args_t", "CrossBwdArgs": "dctr_crossnet_bwd_args_t", "CrossMixBwdArgs": "dctr_crossnet_mix_bwd_args_t",
             "AfmBwdArgs": "dctr_afm_bwd_args_t", "HostCol": "dctr_host_col_t", "AdamSeg": "dctr_adam_seg_t",
             "DinAttnArgs": "dctr_din_attn_args_t", "CrossnetArgs": "dctr_crossnet_args_t", "DinGatherArgs": "dctr_din_gather_t",
             "PoolSeq": "dctr_pool_seq_t"}
    mirrors = [n for n in dir(_C) if isinstance(getattr(_C, n), type) and issubclass(getattr(_C, n), ctypes.Structure)
               and getattr(_C, n) is not ctypes.Structure]
    assert sorted(mirrors) == sorted(pairs), "a ctypes mirror without a header struct in this test: %s" % (set(mirrors) ^ set(pairs))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dctr.h"', "int main(void) {"]
    for py, c in pairs.items():
        lines.append('    printf("%s sizeof %%zu\\n", sizeof(%s));' % (py, c))
        for fname, _ in getattr(_C, py)._fields_:
            lines.append('    printf("%s %s %%zu\\n", offsetof(%s, %s));' % (py, fname, c, fname))
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    for line in filter(None, out):
        py, field, val = line.split()
        cls = getattr(_C, py)
        want = ctypes.sizeof(cls) if field == "sizeof" else getattr(cls, field).offset
        assert int(val) == want, "%s.%s: C says %s, ctypes %d" % (py, field, val, want)


def test_host_packer_is_clean_under_asan_ubsan(tmp_path):
    """host_pack.cpp is the one piece of multi-threaded host code in the library: build it (host side only) with AddressSanitizer
    + UndefinedBehaviorSanitizer together with tests/native/host_pack_san.cpp and run it."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "deepctr_amd", "csrc")
    exe = tmp_path / "host_pack_san"
    cmd = [hipcc, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "--offload-host-only", "-x", "hip",
           "-I", os.path.join(root, "include"), "-I", csrc, os.path.join(csrc, "host_pack.cpp"), os.path.join(csrc, "abi.cpp"),
           os.path.join(root, "tests", "native", "host_pack_san.cpp"), "-o", str(exe), "-pthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", LD_LIBRARY_PATH="/opt/rocm/lib" + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_cin_bwd_workspace_is_z_free_on_the_mfma_shapes():
    """dctr_cin_bwd_workspace_bytes (host arithmetic only): at C3 (B = 4096, F0 = 26, D = 16, CIN[128,128]) the z-free kernels need
    neither the [B*D, F0*Fk] outer-product buffers nor its gradient — the reference's formulation materialises 436 MB for the second
    layer alone (interaction.py:288-295); layer widths outside the kernels' shapes keep the GEMM path and its buffers."""
    import ctypes
    from deepctr_amd import _C
    lib = _C.lib()
    lib.dctr_cin_bwd_workspace_bytes.restype = ctypes.c_size_t

    def need(B, sizes):
        n = len(sizes)
        ls = (ctypes.c_int32 * n)(*sizes)
        fp, bp = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        f = _C.CinArgs(x=None, batch=B, x_stride=432, fields=26, dim=16, n_layers=n, split_half=1, activation=1,
                       layer_size=ctypes.cast(ls, ctypes.c_void_p), filters=ctypes.cast(fp, ctypes.c_void_p),
                       bias=ctypes.cast(bp, ctypes.c_void_p), out=None, workspace=None, workspace_bytes=0)
        a = _C.CinBwdArgs(fwd=ctypes.pointer(f), d_out=None, out_dim=192, dx_accumulate=0, d_filters=None, d_bias=None, dx=None,
                          dx_stride=0)
        return lib.dctr_cin_bwd_workspace_bytes(ctypes.byref(a))

    rows = 4096 * 16
    z_layers = rows * (26 * 26 + 26 * 64) * 4                       # what materialising z costs at C3
    fused, fallback = need(4096, (128, 128)), need(4096, (120, 120))
    assert fused < 0.4 * z_layers, (fused, z_layers)
    assert fallback > z_layers


def test_training_descriptor_arrays_pack_like_the_header():
    """Host logic of the ABI-6 training descriptors (no GPU): ops.make_adam_segments / make_field_grads lay dctr_adam_seg_t /
    dctr_field_grad_t arrays out as the header declares them — pointer, length, l2 and the optional touched bytes per entry —
    and reject touched bytes that do not cover a parameter's 16-B groups."""
    import ctypes
    import torch
    from deepctr_amd import _C, ops
    cpu = torch.device("cpu")
    w = [torch.zeros(40, 16), torch.zeros(7, 4), torch.zeros(5)]
    m, v, g = ([torch.zeros_like(t) for t in w] for _ in range(3))
    tch = [torch.zeros(40 * 16 // 4, dtype=torch.uint8), None, None]
    segs, n, mx = ops.make_adam_segments([(w[i], m[i], v[i], g[i], 0.5 * i, tch[i]) for i in range(3)], cpu)
    assert (n, mx) == (3, 640) and segs.numel() == 3 * ctypes.sizeof(_C.AdamSeg)
    arr = (_C.AdamSeg * 3).from_buffer_copy(bytes(segs.numpy()))
    for i in range(3):
        assert (arr[i].w, arr[i].m, arr[i].v, arr[i].g) == (w[i].data_ptr(), m[i].data_ptr(), v[i].data_ptr(), g[i].data_ptr())
        assert arr[i].n == w[i].numel() and abs(arr[i].l2 - 0.5 * i) < 1e-7
        assert (arr[i].touched or 0) == (0 if tch[i] is None else tch[i].data_ptr())
    # five-tuples (no touched entry) stay valid: the dense step of ABI 5
    segs5, _, _ = ops.make_adam_segments([(w[0], m[0], v[0], g[0], 0.0)], cpu)
    assert (_C.AdamSeg * 1).from_buffer_copy(bytes(segs5.numpy()))[0].touched in (None, 0)
    for bad in (torch.zeros(10, dtype=torch.uint8), torch.zeros(160, dtype=torch.float32)):
        with pytest.raises(ValueError):
            ops.make_adam_segments([(w[0], m[0], v[0], g[0], 0.0, bad)], cpu)
    with pytest.raises(ValueError):                       # a parameter whose length is no multiple of 4 has no 16-B groups to flag
        ops.make_adam_segments([(w[2], m[2], v[2], g[2], 0.0, torch.zeros(1, dtype=torch.uint8))], cpu)
    fg = ops.make_field_grads([(g[0], None, tch[0]), (None, g[2], tch[0]), (g[1], g[2])], cpu)
    fa = (_C.FieldGrad * 3).from_buffer_copy(bytes(fg.numpy()))
    assert (fa[0].g_table, fa[0].g_lin_table or 0, fa[0].touched) == (g[0].data_ptr(), 0, tch[0].data_ptr())
    assert (fa[1].g_table or 0, fa[1].g_lin_table, fa[1].touched or 0) == (0, g[2].data_ptr(), 0)      # no table gradient: no bytes either
    assert (fa[2].g_table, fa[2].g_lin_table, fa[2].touched or 0) == (g[1].data_ptr(), g[2].data_ptr(), 0)
