/*
 * s3o_common.h — shared helpers for the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load the library built from these files.
 *
 * Arithmetic conventions pinned here (see DESIGN.md "Arithmetic contract"):
 *   - every file is compiled with -ffp-contract=off; wherever the reference's
 *     nvcc build would contract a*b+c into one fused op we spell the fusion
 *     out with fmaf(), so oracle and HIP kernels agree bit-for-bit by
 *     construction instead of by compiler mood;
 *   - IEEE binary16 is emulated in software with round-to-nearest-even and a
 *     single rounding per operation (products are rounded from f32, sums are
 *     formed exactly in double before the one rounding to half), which is
 *     what v_cvt_f16_f32 / v_add_f16 do on gfx950.
 */
#ifndef S3O_COMMON_H
#define S3O_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define S3O_API __attribute__((visibility("default")))

typedef uint16_t s3o_half;

enum { S3O_F32 = 0, S3O_F16 = 1 };

static inline float s3o_clampf(float x, float lo, float hi) {
    /* fminf(hi, fmaxf(lo, x)) — raymarching.cu:34-36 */
    return fminf(hi, fmaxf(lo, x));
}

static inline uint32_t s3o_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float s3o_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* binary16 -> binary32, exact. */
static inline float s3o_h2f(s3o_half h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return s3o_bits_f32(sign);
        /* subnormal: value = man * 2^-24 */
        float v = (float)man * 5.9604644775390625e-08f;
        return sign ? -v : v;
    }
    if (exp == 31) return s3o_bits_f32(sign | 0x7f800000u | (man << 13));
    return s3o_bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* binary64 -> binary16, round-to-nearest-even, one rounding. */
static inline s3o_half s3o_d2h(double d) {
    uint64_t u; memcpy(&u, &d, 8);
    uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    int64_t exp = (int64_t)((u >> 52) & 0x7ff);
    uint64_t man = u & 0xfffffffffffffull;
    if (exp == 0x7ff) return (s3o_half)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp == 0 && man == 0) return sign;
    int64_t e = exp - 1023;            /* unbiased */
    uint64_t sig = man | (1ull << 52); /* 53-bit significand, value = sig * 2^(e-52) */
    if (exp == 0) { sig = man; e = -1022; }
    /* target: half normal has 11-bit significand with exponent e in [-14, 15];
       subnormal quantum is 2^-24.  shift = number of low bits to drop. */
    int64_t shift;
    int64_t he;
    if (e < -14) { shift = 42 + (-14 - e); he = 0; } else { shift = 42; he = e + 15; }
    if (shift > 63) return sign; /* far below half the smallest subnormal */
    uint64_t q = sig >> shift;
    uint64_t rem = sig & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q++;
    /* q now holds significand incl. hidden bit (for normals) */
    uint32_t out;
    if (he == 0) {
        out = (uint32_t)q; /* may carry into exponent 1: bit 10 set => normal, correct encoding */
    } else {
        out = (uint32_t)(((uint64_t)(he - 1) << 10) + q); /* hidden bit adds 1 to exponent field */
    }
    if (out >= 0x7c00u) out = 0x7c00u; /* overflow -> inf */
    return (s3o_half)(sign | out);
}

static inline s3o_half s3o_f2h(float f) { return s3o_d2h((double)f); }

/* half + half with one rounding (v_add_f16). */
static inline s3o_half s3o_hadd(s3o_half a, s3o_half b) {
    return s3o_d2h((double)s3o_h2f(a) + (double)s3o_h2f(b));
}
static inline s3o_half s3o_hmul(s3o_half a, s3o_half b) {
    return s3o_d2h((double)s3o_h2f(a) * (double)s3o_h2f(b));
}
static inline s3o_half s3o_hsub(s3o_half a, s3o_half b) {
    return s3o_d2h((double)s3o_h2f(a) - (double)s3o_h2f(b));
}

#endif
