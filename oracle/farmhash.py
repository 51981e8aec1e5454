"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Python side of the `Hash` oracle:

* ``fingerprint64_py``  — pure-Python big-int restatement of FarmHash ``farmhashna::Hash64``
  (= TensorFlow ``Fingerprint64``; the arithmetic behind
  ``tf.strings.to_hash_bucket_fast`` called at /root/reference/deepctr/layers/utils.py:103-107).
  Independent of the C file so the two restatements check each other.
* ``fingerprint64`` / ``hash_bucket_int`` / ``hash_bucket_str`` — ctypes wrappers over
  ``oracle/_build/liboracle_farmhash.so`` (built from ``oracle/farmhash64.c`` by
  ``oracle/build_oracle.py``), fast enough for 1e6-element parity runs.
* ``hash_layer`` — the full ``Hash.call`` semantics (reference utils.py:89-112) including the
  ``vocabulary_path`` CSV lookup (``TextFileInitializer(path,'string',1,'int64',0,',')`` at :80-82 —
  key = column 1, value = column 0, miss -> default_value).

Known-answer status is recorded in oracle/farmhash64.c's header.
"""
import ctypes
import os

import numpy as np

M64 = (1 << 64) - 1
K0 = 0xC3A5C85C97CB3127
K1 = 0xB492B66FBE98F273
K2 = 0x9AE16A3B2F90404F


def _f64(s, i):
    return int.from_bytes(s[i:i + 8], "little")


def _f32(s, i):
    return int.from_bytes(s[i:i + 4], "little")


def _rot(v, s):
    return v if s == 0 else ((v >> s) | (v << (64 - s))) & M64


def _smix(v):
    return v ^ (v >> 47)


def _hl16(u, v, mul):
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _weak(s, i, a, b):
    w, x, y, z = _f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24)
    a = (a + w) & M64
    b = _rot((b + a + z) & M64, 21)
    c = a
    a = (a + x + y) & M64
    b = (b + _rot(a, 44)) & M64
    return (a + z) & M64, (b + c) & M64


def fingerprint64_py(s: bytes) -> int:
    n = len(s)
    if n <= 16:
        if n >= 8:
            mul = (K2 + n * 2) & M64
            a = (_f64(s, 0) + K2) & M64
            b = _f64(s, n - 8)
            c = (_rot(b, 37) * mul + a) & M64
            d = ((_rot(a, 25) + b) * mul) & M64
            return _hl16(c, d, mul)
        if n >= 4:
            mul = (K2 + n * 2) & M64
            a = _f32(s, 0)
            return _hl16((n + (a << 3)) & M64, _f32(s, n - 4), mul)
        if n > 0:
            a, b, c = s[0], s[n >> 1], s[n - 1]
            y = (a + (b << 8)) & 0xFFFFFFFF
            z = (n + (c << 2)) & 0xFFFFFFFF
            return (_smix(((y * K2) & M64) ^ ((z * K0) & M64)) * K2) & M64
        return K2
    if n <= 32:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) * K1) & M64
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & M64
        d = (_f64(s, n - 16) * K2) & M64
        return _hl16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64,
                     (a + _rot((b + K2) & M64, 18) + c) & M64, mul)
    if n <= 64:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) * K2) & M64
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & M64
        d = (_f64(s, n - 16) * K2) & M64
        y = (_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64
        z = _hl16(y, (a + _rot((b + K2) & M64, 18) + c) & M64, mul)
        e = (_f64(s, 16) * mul) & M64
        f = _f64(s, 24)
        g = ((y + _f64(s, n - 32)) * mul) & M64
        h = ((z + _f64(s, n - 24)) * mul) & M64
        return _hl16((_rot((e + f) & M64, 43) + _rot(g, 30) + h) & M64,
                     (e + _rot((f + a) & M64, 18) + g) & M64, mul)
    seed = 81
    x = seed
    y = (seed * K1 + 113) & M64
    z = (_smix((y * K2 + 113) & M64) * K2) & M64
    v1 = v2 = w1 = w2 = 0
    x = (x * K2 + _f64(s, 0)) & M64
    end = ((n - 1) // 64) * 64
    last64 = end + ((n - 1) & 63) - 63
    p = 0
    while True:
        x = (_rot((x + y + v1 + _f64(s, p + 8)) & M64, 37) * K1) & M64
        y = (_rot((y + v2 + _f64(s, p + 48)) & M64, 42) * K1) & M64
        x ^= w2
        y = (y + v1 + _f64(s, p + 40)) & M64
        z = (_rot((z + w1) & M64, 33) * K1) & M64
        v1, v2 = _weak(s, p, (v2 * K1) & M64, (x + w1) & M64)
        w1, w2 = _weak(s, p + 32, (z + w2) & M64, (y + _f64(s, p + 16)) & M64)
        z, x = x, z
        p += 64
        if p == end:
            break
    mul = (K1 + ((z & 0xFF) << 1)) & M64
    p = last64
    w1 = (w1 + ((n - 1) & 63)) & M64
    v1 = (v1 + w1) & M64
    w1 = (w1 + v1) & M64
    x = (_rot((x + y + v1 + _f64(s, p + 8)) & M64, 37) * mul) & M64
    y = (_rot((y + v2 + _f64(s, p + 48)) & M64, 42) * mul) & M64
    x ^= (w2 * 9) & M64
    y = (y + v1 * 9 + _f64(s, p + 40)) & M64
    z = (_rot((z + w1) & M64, 33) * mul) & M64
    v1, v2 = _weak(s, p, (v2 * mul) & M64, (x + w1) & M64)
    w1, w2 = _weak(s, p + 32, (z + w2) & M64, (y + _f64(s, p + 16)) & M64)
    z, x = x, z
    return _hl16((_hl16(v1, w1, mul) + ((_smix(y) * K0) & M64) + z) & M64,
                 (_hl16(v2, w2, mul) + x) & M64, mul)


# ---------------------------------------------------------------------------
# ctypes wrappers over the C restatement
# ---------------------------------------------------------------------------
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "liboracle_farmhash.so")
        if not os.path.exists(path):
            from . import build_oracle
            build_oracle.build()
        lib = ctypes.CDLL(path)
        lib.oracle_fingerprint64.restype = ctypes.c_uint64
        lib.oracle_fingerprint64.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        lib.oracle_hash_bucket_i64.restype = None
        lib.oracle_hash_bucket_i64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_void_p]
        lib.oracle_hash_bucket_bytes.restype = None
        lib.oracle_hash_bucket_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_int, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def fingerprint64(s: bytes) -> int:
    return int(_lib().oracle_fingerprint64(s, len(s)))


def hash_bucket_int(x, num_buckets, mask_zero=False):
    """Hash.call on an integer array (any int dtype) -> int64 array, same shape."""
    x = np.ascontiguousarray(np.asarray(x).astype(np.int64))
    out = np.empty_like(x)
    _lib().oracle_hash_bucket_i64(x.ctypes.data, x.size, int(num_buckets), int(bool(mask_zero)), out.ctypes.data)
    return out


def pack_strings(strs):
    """list/array of str|bytes -> (uint8 bytes, int64 offsets[n+1])."""
    bs = [s if isinstance(s, bytes) else str(s).encode("utf-8") for s in strs]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=offsets[1:])
    data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    return data, offsets


def hash_bucket_str(x, num_buckets, mask_zero=False):
    arr = np.asarray(x)
    flat = [v for v in arr.reshape(-1)]
    data, offsets = pack_strings(flat)
    out = np.empty(len(flat), dtype=np.int64)
    _lib().oracle_hash_bucket_bytes(data.ctypes.data, offsets.ctypes.data, len(flat), int(num_buckets),
                                    int(bool(mask_zero)), out.ctypes.data)
    return out.reshape(arr.shape)


def load_vocabulary(path):
    """TextFileInitializer(path, 'string', 1, 'int64', 0, delimiter=',') — reference utils.py:80-82."""
    table = {}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if not line:
                continue
            cols = line.split(",")
            table[cols[1]] = int(cols[0])
    return table


def _as_tf_string(v):
    """tf.as_string semantics for one scalar (reference utils.py:91-93): ints -> %d, floats -> %f."""
    if isinstance(v, (bytes, np.bytes_)):
        return v.decode("utf-8")
    if isinstance(v, (str, np.str_)):
        return str(v)
    if isinstance(v, (float, np.floating)):
        return "%f" % float(v)
    return "%d" % int(v)


def hash_layer(x, num_buckets, mask_zero=False, vocabulary_path=None, default_value=0):
    """Full ``Hash.call`` (reference deepctr/layers/utils.py:89-112)."""
    arr = np.asarray(x)
    if vocabulary_path:
        table = load_vocabulary(vocabulary_path)
        flat = [table.get(_as_tf_string(v), default_value) for v in arr.reshape(-1)]
        return np.asarray(flat, dtype=np.int64).reshape(arr.shape)
    if arr.dtype.kind in "iu":
        return hash_bucket_int(arr, num_buckets, mask_zero)
    strs = np.asarray([_as_tf_string(v) for v in arr.reshape(-1)], dtype=object).reshape(arr.shape)
    return hash_bucket_str(strs, num_buckets, mask_zero)
