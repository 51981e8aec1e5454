// Device-side Fingerprint64 (FarmHash farmhashna::Hash64) for the Hash layer
// (reference deepctr/layers/utils.py:89-112 -> tf.as_string + tf.strings.to_hash_bucket_fast).
//
// Two forms:
//  * dctr_fp64_packed(): the id has already been rendered to decimal ASCII inside three 64-bit
//    registers (little-endian, byte i of the string = byte i of the register file) — no scratch
//    memory, no byte loads; covers every decimal int32 / int64 (<= 20 characters), i.e. the
//    0-16 and 17-32 byte branches of the algorithm.
//  * dctr_fp64_bytes(): arbitrary-length strings read from global memory (string-dtype features).
// All arithmetic is unsigned 64-bit wrap-around, exactly as in the published algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctr {

static constexpr uint64_t FH_K0 = 0xc3a5c85c97cb3127ULL;
static constexpr uint64_t FH_K1 = 0xb492b66fbe98f273ULL;
static constexpr uint64_t FH_K2 = 0x9ae16a3b2f90404fULL;

__device__ __forceinline__ uint64_t fh_rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
__device__ __forceinline__ uint64_t fh_smix(uint64_t v) { return v ^ (v >> 47); }
__device__ __forceinline__ uint64_t fh_len16(uint64_t u, uint64_t v, uint64_t mul) {
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}

// ---- decimal string packed into registers -------------------------------------------------------
struct Packed24 {
    uint64_t w0, w1, w2;  // bytes 0-7, 8-15, 16-23 of the string
    int len;
};

__device__ __forceinline__ void p24_set(Packed24& s, int pos, uint64_t ch) {
    const uint64_t v = ch << ((pos & 7) * 8);
    if (pos < 8) s.w0 |= v;
    else if (pos < 16) s.w1 |= v;
    else s.w2 |= v;
}

// 8 bytes starting at byte `pos` (pos + 8 <= 24)
__device__ __forceinline__ uint64_t p24_fetch64(const Packed24& s, int pos) {
    const int i = pos >> 3, sh = (pos & 7) * 8;
    const uint64_t lo = i == 0 ? s.w0 : (i == 1 ? s.w1 : s.w2);
    const uint64_t hi = i == 0 ? s.w1 : (i == 1 ? s.w2 : 0ull);
    return sh == 0 ? lo : ((lo >> sh) | (hi << (64 - sh)));
}
__device__ __forceinline__ uint64_t p24_fetch32(const Packed24& s, int pos) {  // pos + 4 <= 8 here
    return (s.w0 >> (pos * 8)) & 0xffffffffull;
}
__device__ __forceinline__ uint32_t p24_byte(const Packed24& s, int pos) {      // pos < 8 here
    return (uint32_t)((s.w0 >> (pos * 8)) & 0xffu);
}

// "%lld" rendering of x (tf.as_string on an integer tensor), minus sign included, no padding.
__device__ __forceinline__ Packed24 decimal_ascii(int64_t x) {
    Packed24 s{0, 0, 0, 0};
    const bool neg = x < 0;
    uint64_t u = neg ? (0ull - (uint64_t)x) : (uint64_t)x;
    int nd = 1;
    {
        uint64_t p = 10;
#pragma unroll 1
        while (nd < 20 && u >= p) {
            ++nd;
            p *= 10;   // 10^19 fits in uint64; loop exits before 10^20
        }
    }
    const int sign = neg ? 1 : 0;
    if (neg) p24_set(s, 0, (uint64_t)'-');
#pragma unroll 1
    for (int i = 0; i < nd; ++i) {
        const uint64_t d = u % 10;
        u /= 10;
        p24_set(s, sign + nd - 1 - i, (uint64_t)'0' + d);
    }
    s.len = sign + nd;
    return s;
}

__device__ __forceinline__ Packed24 decimal_ascii_i32(int32_t x) {
    Packed24 s{0, 0, 0, 0};
    const bool neg = x < 0;
    uint32_t u = neg ? (0u - (uint32_t)x) : (uint32_t)x;
    int nd = 1;
    {
        uint32_t p = 10;
#pragma unroll 1
        while (nd < 10 && u >= p) {
            ++nd;
            p = (nd < 10) ? p * 10u : p;  // 10^9 is the last power that fits
        }
    }
    const int sign = neg ? 1 : 0;
    if (neg) p24_set(s, 0, (uint64_t)'-');
#pragma unroll 1
    for (int i = 0; i < nd; ++i) {
        const uint32_t d = u % 10u;
        u /= 10u;
        p24_set(s, sign + nd - 1 - i, (uint64_t)('0' + d));
    }
    s.len = sign + nd;
    return s;
}

// The same rendering of an int32 without loops or variable-position byte stores: all ten digits by constant divisions, packed as a
// ten-byte string with leading zeros, which one 128-bit shift then drops ("%d": no padding); the sign goes in front afterwards.
__device__ __forceinline__ Packed24 decimal_ascii_i32_fast(int32_t x) {
    const bool neg = x < 0;
    const uint32_t u = neg ? (0u - (uint32_t)x) : (uint32_t)x;
    const uint32_t hi = u / 100000u, lo = u - hi * 100000u;           // hi <= 42949, lo <= 99999
    const uint32_t h4 = hi / 10000u, h3 = hi / 1000u - h4 * 10u, h2 = hi / 100u - (hi / 1000u) * 10u, h1 = hi / 10u - (hi / 100u) * 10u,
                   h0 = hi - (hi / 10u) * 10u;
    const uint32_t l4 = lo / 10000u, l3 = lo / 1000u - l4 * 10u, l2 = lo / 100u - (lo / 1000u) * 10u, l1 = lo / 10u - (lo / 100u) * 10u,
                   l0 = lo - (lo / 10u) * 10u;
    // string order = most significant digit first = lowest byte
    const uint32_t b0 = h4 | (h3 << 8) | (h2 << 16) | (h1 << 24);    // bytes 0-3
    const uint32_t b1 = h0 | (l4 << 8) | (l3 << 16) | (l2 << 24);    // bytes 4-7
    const uint32_t b2 = l1 | (l0 << 8);                              // bytes 8-9
    uint64_t w0 = (((uint64_t)b1 << 32) | b0) | 0x3030303030303030ull;
    uint64_t w1 = (uint64_t)(b2 | 0x3030u);
    int nd = 10;                                                      // significant digits: 10 minus the leading zeros
    nd -= (u < 1000000000u) + (u < 100000000u) + (u < 10000000u) + (u < 1000000u) + (u < 100000u) + (u < 10000u) + (u < 1000u) +
          (u < 100u) + (u < 10u);
    const int drop = 8 * (10 - nd);                                   // bits to shift out at the string's front: 0 .. 72
    if (drop >= 64) {
        w0 = w1 >> (drop - 64);
        w1 = 0;
    } else if (drop > 0) {
        w0 = (w0 >> drop) | (w1 << (64 - drop));
        w1 >>= drop;
    }
    Packed24 s{w0, w1, 0, nd};
    if (neg) {                                                        // (at most 11 characters)
        s.w1 = (w1 << 8) | (w0 >> 56);
        s.w0 = (w0 << 8) | (uint64_t)'-';
        s.len = nd + 1;
    }
    return s;
}

// farmhashna::Hash64 restricted to len <= 24 (HashLen0to16 and HashLen17to32)
__device__ __forceinline__ uint64_t dctr_fp64_packed(const Packed24& s) {
    const uint64_t len = (uint64_t)s.len;
    if (s.len <= 16) {
        if (s.len >= 8) {
            const uint64_t mul = FH_K2 + len * 2;
            const uint64_t a = p24_fetch64(s, 0) + FH_K2;
            const uint64_t b = p24_fetch64(s, s.len - 8);
            const uint64_t c = fh_rot(b, 37) * mul + a;
            const uint64_t d = (fh_rot(a, 25) + b) * mul;
            return fh_len16(c, d, mul);
        }
        if (s.len >= 4) {
            const uint64_t mul = FH_K2 + len * 2;
            const uint64_t a = p24_fetch32(s, 0);
            return fh_len16(len + (a << 3), p24_fetch32(s, s.len - 4), mul);
        }
        if (s.len > 0) {
            const uint32_t a = p24_byte(s, 0), b = p24_byte(s, s.len >> 1), c = p24_byte(s, s.len - 1);
            const uint32_t y = a + (b << 8);
            const uint32_t z = (uint32_t)s.len + (c << 2);
            return fh_smix((uint64_t)y * FH_K2 ^ (uint64_t)z * FH_K0) * FH_K2;
        }
        return FH_K2;
    }
    const uint64_t mul = FH_K2 + len * 2;
    const uint64_t a = p24_fetch64(s, 0) * FH_K1;
    const uint64_t b = p24_fetch64(s, 8);
    const uint64_t c = p24_fetch64(s, s.len - 8) * mul;
    const uint64_t d = p24_fetch64(s, s.len - 16) * FH_K2;
    return fh_len16(fh_rot(a + b, 43) + fh_rot(c, 30) + d, a + fh_rot(b + FH_K2, 18) + c, mul);
}

// Hash.call for one integer id: bucket in [0,nb) (or [1,nb] with 0 reserved when mask_zero)
__device__ __forceinline__ int64_t hash_bucket_id(int64_t x, bool is_i32, uint64_t num_buckets, bool mask_zero) {
    const uint64_t nb = mask_zero ? num_buckets - 1 : num_buckets;
    const Packed24 s = is_i32 ? decimal_ascii_i32((int32_t)x) : decimal_ascii(x);
    int64_t h = (int64_t)(dctr_fp64_packed(s) % nb);
    if (mask_zero) h = (x != 0) ? h + 1 : 0;
    return h;
}

// ---- arbitrary byte strings in global memory ------------------------------------------------------
__device__ __forceinline__ uint64_t g_fetch64(const uint8_t* p) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ uint64_t g_fetch32(const uint8_t* p) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ void fh_weak32(const uint8_t* s, uint64_t a, uint64_t b, uint64_t& o1, uint64_t& o2) {
    const uint64_t w = g_fetch64(s), x = g_fetch64(s + 8), y = g_fetch64(s + 16), z = g_fetch64(s + 24);
    a += w;
    b = fh_rot(b + a + z, 21);
    const uint64_t c = a;
    a += x;
    a += y;
    b += fh_rot(a, 44);
    o1 = a + z;
    o2 = b + c;
}

__device__ inline uint64_t dctr_fp64_bytes(const uint8_t* s, uint64_t len) {
    if (len <= 16) {
        if (len >= 8) {
            const uint64_t mul = FH_K2 + len * 2;
            const uint64_t a = g_fetch64(s) + FH_K2;
            const uint64_t b = g_fetch64(s + len - 8);
            const uint64_t c = fh_rot(b, 37) * mul + a;
            const uint64_t d = (fh_rot(a, 25) + b) * mul;
            return fh_len16(c, d, mul);
        }
        if (len >= 4) {
            const uint64_t mul = FH_K2 + len * 2;
            const uint64_t a = g_fetch32(s);
            return fh_len16(len + (a << 3), g_fetch32(s + len - 4), mul);
        }
        if (len > 0) {
            const uint32_t a = s[0], b = s[len >> 1], c = s[len - 1];
            const uint32_t y = a + (b << 8);
            const uint32_t z = (uint32_t)len + (c << 2);
            return fh_smix((uint64_t)y * FH_K2 ^ (uint64_t)z * FH_K0) * FH_K2;
        }
        return FH_K2;
    }
    if (len <= 32) {
        const uint64_t mul = FH_K2 + len * 2;
        const uint64_t a = g_fetch64(s) * FH_K1;
        const uint64_t b = g_fetch64(s + 8);
        const uint64_t c = g_fetch64(s + len - 8) * mul;
        const uint64_t d = g_fetch64(s + len - 16) * FH_K2;
        return fh_len16(fh_rot(a + b, 43) + fh_rot(c, 30) + d, a + fh_rot(b + FH_K2, 18) + c, mul);
    }
    if (len <= 64) {
        const uint64_t mul = FH_K2 + len * 2;
        const uint64_t a = g_fetch64(s) * FH_K2;
        const uint64_t b = g_fetch64(s + 8);
        const uint64_t c = g_fetch64(s + len - 8) * mul;
        const uint64_t d = g_fetch64(s + len - 16) * FH_K2;
        const uint64_t y = fh_rot(a + b, 43) + fh_rot(c, 30) + d;
        const uint64_t z = fh_len16(y, a + fh_rot(b + FH_K2, 18) + c, mul);
        const uint64_t e = g_fetch64(s + 16) * mul;
        const uint64_t f = g_fetch64(s + 24);
        const uint64_t g = (y + g_fetch64(s + len - 32)) * mul;
        const uint64_t h = (z + g_fetch64(s + len - 24)) * mul;
        return fh_len16(fh_rot(e + f, 43) + fh_rot(g, 30) + h, e + fh_rot(f + a, 18) + g, mul);
    }
    const uint64_t seed = 81;
    uint64_t x = seed;
    uint64_t y = seed * FH_K1 + 113;
    uint64_t z = fh_smix(y * FH_K2 + 113) * FH_K2;
    uint64_t v1 = 0, v2 = 0, w1 = 0, w2 = 0;
    x = x * FH_K2 + g_fetch64(s);
    const uint8_t* end = s + ((len - 1) / 64) * 64;
    const uint8_t* last64 = end + ((len - 1) & 63) - 63;
    do {
        x = fh_rot(x + y + v1 + g_fetch64(s + 8), 37) * FH_K1;
        y = fh_rot(y + v2 + g_fetch64(s + 48), 42) * FH_K1;
        x ^= w2;
        y += v1 + g_fetch64(s + 40);
        z = fh_rot(z + w1, 33) * FH_K1;
        fh_weak32(s, v2 * FH_K1, x + w1, v1, v2);
        fh_weak32(s + 32, z + w2, y + g_fetch64(s + 16), w1, w2);
        const uint64_t t = z;
        z = x;
        x = t;
        s += 64;
    } while (s != end);
    const uint64_t mul = FH_K1 + ((z & 0xff) << 1);
    s = last64;
    w1 += ((len - 1) & 63);
    v1 += w1;
    w1 += v1;
    x = fh_rot(x + y + v1 + g_fetch64(s + 8), 37) * mul;
    y = fh_rot(y + v2 + g_fetch64(s + 48), 42) * mul;
    x ^= w2 * 9;
    y += v1 * 9 + g_fetch64(s + 40);
    z = fh_rot(z + w1, 33) * mul;
    fh_weak32(s, v2 * mul, x + w1, v1, v2);
    fh_weak32(s + 32, z + w2, y + g_fetch64(s + 16), w1, w2);
    {
        const uint64_t t = z;
        z = x;
        x = t;
    }
    return fh_len16(fh_len16(v1, w1, mul) + fh_smix(y) * FH_K0 + z, fh_len16(v2, w2, mul) + x, mul);
}

}  // namespace dctr
