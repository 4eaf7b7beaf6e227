// sh_eval.hpp — real spherical-harmonics basis (+ Jacobian) of one direction, shared by encoders.hip (the SH encoder of
// shencoder/src/shencoder.cu:27-382) and ngp_head.hip (the fused colour-network input).  See encoders.hip for the
// formulation: Y_l^{±m} = K_l^m * T_l^m(z) * {Re,Im}(x+iy)^m evaluated by recurrences, fp32.
#pragma once
#include "s3d_common.hpp"
#include <math.h>

namespace s3d {

constexpr uint32_t kMaxDeg = 8;
struct ShNorm { float k[kMaxDeg][kMaxDeg]; };  // k[l][m], m <= l

inline void host_sh_norm(uint32_t degree, ShNorm& K) {
    memset(&K, 0, sizeof(K));
    for (uint32_t l = 0; l < degree; l++)
        for (uint32_t m = 0; m <= l; m++) {
            double ratio = 1.0;
            for (uint32_t k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
            double n = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * ratio);
            if (m) n *= ((m & 1) ? -1.0 : 1.0) * M_SQRT2;
            K.k[l][m] = (float)n;
        }
}

template <uint32_t DEG, bool JAC>
__device__ __forceinline__ void sh_eval(float x, float y, float z, const ShNorm& K, float (&o)[DEG * DEG],
                                        float (&jx)[JAC ? DEG * DEG : 1], float (&jy)[JAC ? DEG * DEG : 1],
                                        float (&jz)[JAC ? DEG * DEG : 1]) {
    float c[DEG + 1], s[DEG + 1];
    c[0] = 1.0f; s[0] = 0.0f;
#pragma unroll
    for (uint32_t m = 1; m <= DEG; m++) {
        c[m] = __builtin_fmaf(x, c[m - 1], -(y * s[m - 1]));
        s[m] = __builtin_fmaf(x, s[m - 1], y * c[m - 1]);
    }
    float T[DEG][DEG + 2];
#pragma unroll
    for (uint32_t l = 0; l < DEG; l++)
#pragma unroll
        for (uint32_t m = 0; m < DEG + 2; m++) T[l][m] = 0.0f;
#pragma unroll
    for (uint32_t m = 0; m < DEG; m++) {
        float dfact = 1.0f;
#pragma unroll
        for (uint32_t k = 1; k <= m; k++) dfact *= (float)(2 * k - 1);
        T[m][m] = dfact;
        if (m + 1 < DEG) T[m + 1][m] = (float)(2 * m + 1) * z * dfact;
#pragma unroll
        for (uint32_t l = m + 2; l < DEG; l++)
            T[l][m] = __builtin_fmaf((float)(2 * l - 1) * z, T[l - 1][m], -((float)(l + m - 1) * T[l - 2][m])) *
                      (1.0f / (float)(l - m));
    }
#pragma unroll
    for (uint32_t l = 0; l < DEG; l++) {
        const uint32_t base = l * l + l;
        o[base] = K.k[l][0] * T[l][0];
        if (JAC) { jx[base] = 0.0f; jy[base] = 0.0f; jz[base] = K.k[l][0] * T[l][1]; }
#pragma unroll
        for (uint32_t m = 1; m <= l; m++) {
            const float kt = K.k[l][m] * T[l][m];
            o[base + m] = kt * c[m];
            o[base - m] = kt * s[m];
            if (JAC) {
                const float kz = K.k[l][m] * T[l][m + 1];
                const float km = kt * (float)m;
                jx[base + m] = km * c[m - 1];
                jx[base - m] = km * s[m - 1];
                jy[base + m] = -km * s[m - 1];
                jy[base - m] = km * c[m - 1];
                jz[base + m] = kz * c[m];
                jz[base - m] = kz * s[m];
            }
        }
    }
}

}  // namespace s3d
