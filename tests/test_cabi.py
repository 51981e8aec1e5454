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
    assert lib.dctr_abi_version() == _C.ABI_VERSION == 1
    assert lib.dctr_target_arch() == b"gfx950"


def test_struct_sizes_match_the_header_layout():
    import ctypes
    from deepctr_amd import _C
    assert ctypes.sizeof(_C.FieldDesc) == 48
    assert ctypes.sizeof(_C.GatherFmArgs) == 8 * 4 + 4 * 6 + 8 * 3 + 4 * 2 + 8 * 6 + 4 * 2
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
    assert lib.dctr_opt_multi(9, None, 0, 0, 0.001, 0.9, 0.999, 1e-7, 1, None) == -4
    assert b"optimizer kind 9" in lib.dctr_last_error()
    assert lib.dctr_mlp_bwd(None, None) == -1 and lib.dctr_cin_bwd(None, None) == -1 and lib.dctr_crossnet_bwd(None, None) == -1
    assert lib.dctr_embed_gather_fm_bwd(None, None) == -1 and lib.dctr_embed_pool_bwd(None, None) == -1
