// s3d_common.hpp — shared device/host helpers for libseal3d_hip (gfx950 only).
//
// Arithmetic contract: the library is built with -ffp-contract=off and
// every fused multiply-add is spelled __builtin_fmaf, mirroring the CPU oracle
// expression by expression, so integer/index results (cell coordinates, hash rows,
// per-ray sample counts, span offsets) are bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <string.h>

#include "../../include/seal3d_hip.h"

#define S3D_EXPORT extern "C" __attribute__((visibility("default")))

namespace s3d {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(s3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return S3D_ERR_HIP;
    }
    return S3D_OK;
}

#define S3D_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            s3d::set_error(__VA_ARGS__);  \
            return S3D_ERR_INVALID;       \
        }                                 \
    } while (0)

#define S3D_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) {                                             \
            s3d::set_error("%s: %s", #call, hipGetErrorString(e__));         \
            return S3D_ERR_HIP;                                              \
        }                                                                    \
    } while (0)

// One-time per-device kernel attribute setup (dynamic LDS opt-in): `flags` is a per-call-site static, one bit per device
// ordinal, so a second GPU in the same process gets its own hipFuncSetAttribute calls and concurrent host threads at worst
// repeat an idempotent call.
inline bool device_needs_setup(std::atomic<uint64_t>& flags, int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) { *dev_out = -1; return true; }
    *dev_out = dev;
    return ((flags.load(std::memory_order_acquire) >> dev) & 1ull) == 0;
}
inline void device_setup_done(std::atomic<uint64_t>& flags, int dev) {
    if (dev >= 0) flags.fetch_or(1ull << dev, std::memory_order_release);
}

template <typename T>
__host__ __device__ inline T div_up(T a, T b) { return (a + b - 1) / b; }

// 256 CUs x 8 resident blocks: cap for grid-stride launches of streaming kernels.
constexpr uint32_t kMaxStreamBlocks = 2048;

inline uint32_t stream_grid(uint64_t work, uint32_t block) {
    uint64_t g = div_up<uint64_t>(work, block);
    if (g > kMaxStreamBlocks) g = kMaxStreamBlocks;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// Padded sample batches (`n_valid` of seal3d_hip.h): the ray marcher leaves the number of samples it produced in device
// memory, the buffers behind it have a static extent B.  Rows [round_up(*n_valid, 128), B) are absent for every kernel
// that takes the pointer — not read, not written — so the work follows the samples while launch geometry and strides
// (level-major layouts) stay those of B.  NULL = all B rows.
__device__ __forceinline__ uint32_t valid_rows(uint32_t B, const int32_t* n_valid) {
    if (!n_valid) return B;
    const int32_t n = *n_valid;
    const uint32_t v = n <= 0 ? 0u : (((uint32_t)n + 127u) & ~127u);
    return v < B ? v : B;
}

// lane id inside the 64-wide wavefront
__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// inclusive wave64 prefix sum (shuffle-up ladder; DPP row_shr/bcast is what hipcc lowers it to)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

}  // namespace s3d
