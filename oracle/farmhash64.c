/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (deepctr_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * CPU restatement of the string fingerprint that the reference's `Hash` layer
 * delegates to TensorFlow:
 *
 *   /root/reference/deepctr/layers/utils.py:103-107
 *       hash_x = tf.strings.to_hash_bucket_fast(x, num_buckets)
 *
 * The arithmetic is NOT in /root/reference: it lives in TensorFlow (versions the
 * reference's CI pins: 1.15.5, 2.10.0, 2.15.0, 2.20.0 — .github/workflows/ci.yml:46-47,
 * ci2.yml:45-63), whose op `StringToHashBucketFast` is documented as
 * `Fingerprint64(s) mod num_buckets` with Fingerprint64 = FarmHash
 * `farmhashna::Hash64` (google/farmhash, farmhash.cc).  This file restates that
 * published algorithm.  All arithmetic is unsigned 64-bit wrap-around, loads are
 * little-endian and unaligned.
 *
 * Pinned by known-answer vectors in tests/test_oracle_golden.py (KATS, test_to_hash_bucket_fast_doc_examples):
 *   - TensorFlow's own frozen fingerprints (tensorflow/core/platform/fingerprint_test.cc):
 *       Fingerprint64("Hello") = 15404698994557526151, ("World") = 18308117990299812472
 *   - TensorFlow's string_to_hash_bucket_op_test.py comments:
 *       'a' -> 12917804110809363939, 'b' -> 11795596070477164822,
 *       'c' -> 11430444447143000872, 'd' ->  4470636696479570465  (mod 10 -> 9,2,2,5)
 *   - pyfarmhash README: hash64("abc") = 2640714258260161385
 *   - TF API doc: to_hash_bucket_fast(["Hello","TensorFlow","2.x"], 3) = [0,2,2]
 * These cover the 1-3, 4-7 and 8-16 byte branches.  The 17-32, 33-64 and >64 byte
 * branches have NO external known answer available offline: "parity unpinned" for
 * inputs longer than 16 bytes (decimal int64 ids with >= 17 characters, long strings).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static const uint64_t k0 = 0xc3a5c85c97cb3127ULL;
static const uint64_t k1 = 0xb492b66fbe98f273ULL;
static const uint64_t k2 = 0x9ae16a3b2f90404fULL;

static uint64_t fetch64(const unsigned char *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint64_t fetch32(const unsigned char *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}

static uint64_t hash_len0to16(const unsigned char *s, size_t len) {
    if (len >= 8) {
        uint64_t mul = k2 + len * 2;
        uint64_t a = fetch64(s) + k2;
        uint64_t b = fetch64(s + len - 8);
        uint64_t c = rot(b, 37) * mul + a;
        uint64_t d = (rot(a, 25) + b) * mul;
        return hash_len16(c, d, mul);
    }
    if (len >= 4) {
        uint64_t mul = k2 + len * 2;
        uint64_t a = fetch32(s);
        return hash_len16(len + (a << 3), fetch32(s + len - 4), mul);
    }
    if (len > 0) {
        uint8_t a = s[0];
        uint8_t b = s[len >> 1];
        uint8_t c = s[len - 1];
        uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
        uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
        return shift_mix(y * k2 ^ z * k0) * k2;
    }
    return k2;
}

static uint64_t hash_len17to32(const unsigned char *s, size_t len) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch64(s) * k1;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    return hash_len16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + k2, 18) + c, mul);
}

static uint64_t hash_len33to64(const unsigned char *s, size_t len) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch64(s) * k2;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    uint64_t y = rot(a + b, 43) + rot(c, 30) + d;
    uint64_t z = hash_len16(y, a + rot(b + k2, 18) + c, mul);
    uint64_t e = fetch64(s + 16) * mul;
    uint64_t f = fetch64(s + 24);
    uint64_t g = (y + fetch64(s + len - 32)) * mul;
    uint64_t h = (z + fetch64(s + len - 24)) * mul;
    return hash_len16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}

static void weak_hash32_seeds(const unsigned char *s, uint64_t a, uint64_t b, uint64_t *o1, uint64_t *o2) {
    uint64_t w = fetch64(s), x = fetch64(s + 8), y = fetch64(s + 16), z = fetch64(s + 24);
    a += w;
    b = rot(b + a + z, 21);
    uint64_t c = a;
    a += x;
    a += y;
    b += rot(a, 44);
    *o1 = a + z;
    *o2 = b + c;
}

uint64_t oracle_fingerprint64(const unsigned char *s, size_t len) {
    const uint64_t seed = 81;
    if (len <= 32) return len <= 16 ? hash_len0to16(s, len) : hash_len17to32(s, len);
    if (len <= 64) return hash_len33to64(s, len);

    uint64_t x = seed;
    uint64_t y = seed * k1 + 113;
    uint64_t z = shift_mix(y * k2 + 113) * k2;
    uint64_t v1 = 0, v2 = 0, w1 = 0, w2 = 0;
    x = x * k2 + fetch64(s);
    const unsigned char *end = s + ((len - 1) / 64) * 64;
    const unsigned char *last64 = end + ((len - 1) & 63) - 63;
    do {
        x = rot(x + y + v1 + fetch64(s + 8), 37) * k1;
        y = rot(y + v2 + fetch64(s + 48), 42) * k1;
        x ^= w2;
        y += v1 + fetch64(s + 40);
        z = rot(z + w1, 33) * k1;
        weak_hash32_seeds(s, v2 * k1, x + w1, &v1, &v2);
        weak_hash32_seeds(s + 32, z + w2, y + fetch64(s + 16), &w1, &w2);
        uint64_t t = z; z = x; x = t;
        s += 64;
    } while (s != end);
    uint64_t mul = k1 + ((z & 0xff) << 1);
    s = last64;
    w1 += ((len - 1) & 63);
    v1 += w1;
    w1 += v1;
    x = rot(x + y + v1 + fetch64(s + 8), 37) * mul;
    y = rot(y + v2 + fetch64(s + 48), 42) * mul;
    x ^= w2 * 9;
    y += v1 * 9 + fetch64(s + 40);
    z = rot(z + w1, 33) * mul;
    weak_hash32_seeds(s, v2 * mul, x + w1, &v1, &v2);
    weak_hash32_seeds(s + 32, z + w2, y + fetch64(s + 16), &w1, &w2);
    { uint64_t t = z; z = x; x = t; }
    return hash_len16(hash_len16(v1, w1, mul) + shift_mix(y) * k0 + z, hash_len16(v2, w2, mul) + x, mul);
}

/* tf.as_string on an integer tensor == C "%lld" (no padding); the reference calls it at
 * deepctr/layers/utils.py:91-93 before hashing. */
static size_t decimal_ascii(int64_t x, unsigned char *buf) {
    return (size_t)snprintf((char *)buf, 24, "%lld", (long long)x);
}

/* Hash.call for integer inputs, reference deepctr/layers/utils.py:89-112:
 *   nb = num_buckets - (1 if mask_zero else 0)                       (:101)
 *   h  = Fingerprint64(as_string(x)) mod nb   (uint64 modulo -> int64) (:103-107)
 *   if mask_zero: h = (h + 1) * (as_string(x) != "0")                (:108-110)  */
void oracle_hash_bucket_i64(const int64_t *x, int64_t n, int64_t num_buckets, int mask_zero, int64_t *out) {
    uint64_t nb = (uint64_t)(mask_zero ? num_buckets - 1 : num_buckets);
    unsigned char buf[32];
    for (int64_t i = 0; i < n; ++i) {
        size_t len = decimal_ascii(x[i], buf);
        int64_t h = (int64_t)(oracle_fingerprint64(buf, len) % nb);
        if (mask_zero) h = (h + 1) * (int64_t)(x[i] != 0);
        out[i] = h;
    }
}

/* Same for pre-rendered byte strings (string-dtype features): bytes + offsets[n+1]. */
void oracle_hash_bucket_bytes(const unsigned char *bytes, const int64_t *offsets, int64_t n, int64_t num_buckets,
                              int mask_zero, int64_t *out) {
    uint64_t nb = (uint64_t)(mask_zero ? num_buckets - 1 : num_buckets);
    for (int64_t i = 0; i < n; ++i) {
        const unsigned char *s = bytes + offsets[i];
        size_t len = (size_t)(offsets[i + 1] - offsets[i]);
        int64_t h = (int64_t)(oracle_fingerprint64(s, len) % nb);
        if (mask_zero) h = (h + 1) * (int64_t)!(len == 1 && s[0] == '0');
        out[i] = h;
    }
}
