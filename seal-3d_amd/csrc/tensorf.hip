// tensorf.hip — TensoRF vector-matrix features (tensoRF/network.py:112-153 of the reference: get_sigma_feat / get_color_feat).
//
// The reference samples, per point, three plane factors [1,R,H,W] and three line factors [1,R,D,1] with twelve
// F.grid_sample calls (bilinear, zeros padding, align_corners=True; the lines as "fake 2-D" images of width 1), stacks
// and concatenates the [R,N] results, multiplies and (for the density) sums them.  Here one lane does all of it for
// one point: out[N] = sum_i sum_r plane_i[r](u_i, v_i) * line_i[r](w_i)  (reduce = 1), or the products themselves as
// [sum_i R_i, N] (reduce = 0: the operand the reference transposes into basis_mat).  Same interpolation arithmetic as
// torch's grid sampler: index = ((c + 1) / 2) * (size - 1), corner weights as products of the distances to the opposite
// corner, corners accumulated in the order nw, ne, sw, se, out-of-range corners skipped.
#include "s3d_common.hpp"
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>

namespace s3d {
namespace {

struct VmFactors {
    const float* plane[3];
    const float* line[3];
    uint32_t rank[3];
    uint32_t W[3], H[3], Dn[3];   // plane i is [rank, H, W] (W <-> coordinate mat_ids[i][0], H <-> mat_ids[i][1]); line i [rank, Dn]
    uint32_t cu[3], cv[3], cw[3]; // coordinate index of u (-> W), v (-> H), w (-> line)
    uint32_t row0[3];             // first output row of component i (reduce = 0)
    // optional rank-fastest shadows (s3d_vm_transpose_factors): plane_t[i] [H][W][rank], line_t[i] [Dn][rank].  A corner's rank
    // channels are then ONE contiguous run (192 bytes at rank 48) instead of `rank` four-byte words H * W * 4 bytes apart: the
    // colour feature kernel spent 77 of its 117 us in those gathers (timing-only build, profiles/r11_tensorf_vm.md)
    const float* plane_t[3];
    const float* line_t[3];
};

// the four ranks r .. r + 3 of a cell of a rank-fastest shadow (16-byte aligned: rank % 4 == 0), or zeros
__device__ __forceinline__ float4 vm_load4(const float* __restrict__ base, bool ok, size_t cell, uint32_t R, uint32_t r) {
    return ok ? *reinterpret_cast<const float4*>(base + cell * R + r) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__device__ __forceinline__ float unnormalize(float c, uint32_t size) { return ((c + 1.0f) / 2.0f) * (float)(size - 1); }

template <bool REDUCE>
__global__ void __launch_bounds__(256) k_vm_features(const float* __restrict__ x, uint32_t N, VmFactors f, float* __restrict__ out,
                                                     const int32_t* __restrict__ n_valid) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= valid_rows(N, n_valid)) return;  // (a padded sample batch: rows behind the device-side count are absent)
    const float p[3] = {x[(size_t)n * 3], x[(size_t)n * 3 + 1], x[(size_t)n * 3 + 2]};
    float total = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
        const float ix = unnormalize(p[f.cu[i]], f.W[i]), iy = unnormalize(p[f.cv[i]], f.H[i]), iz = unnormalize(p[f.cw[i]], f.Dn[i]);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        // corner weights (grid_sampler: nw = (ix_se - ix)(iy_se - iy), ne = (ix - ix_sw)(iy_sw - iy), ...)
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        const float lz1 = iz - fz, lz0 = (fz + 1.0f) - iz;
        // (non-finite coordinates: every corner is out of range, like the within-bounds tests of the reference kernel)
        const bool okx = fabsf(ix) < 1e9f, oky = fabsf(iy) < 1e9f, okz = fabsf(iz) < 1e9f;
        const int x0 = okx ? (int)fx : -2, y0 = oky ? (int)fy : -2, z0 = okz ? (int)fz : -2;
        const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
        const bool bz0 = z0 >= 0 && z0 < Dn, bz1 = z0 + 1 >= 0 && z0 + 1 < Dn;
        const float* P = f.plane[i];
        const float* Lq = f.line[i];
        const size_t plane_stride = (size_t)H * W;
        const int o_nw = y0 * W + x0;
        float comp = 0.0f;
        if (f.plane_t[i]) {  // rank-fastest shadows: four ranks per 16-byte load, the same arithmetic per rank in the same order
            const float* Pt = f.plane_t[i];
            const float* Ltq = f.line_t[i];
            const uint32_t R = f.rank[i];
            for (uint32_t r0 = 0; r0 < R; r0 += 4) {
                const float4 a_nw = vm_load4(Pt, bx0 && by0, (size_t)o_nw, R, r0), a_ne = vm_load4(Pt, bx1 && by0, (size_t)(o_nw + 1), R, r0);
                const float4 a_sw = vm_load4(Pt, bx0 && by1, (size_t)(o_nw + W), R, r0), a_se = vm_load4(Pt, bx1 && by1, (size_t)(o_nw + W + 1), R, r0);
                const float4 a_l0 = vm_load4(Ltq, bz0, (size_t)z0, R, r0), a_l1 = vm_load4(Ltq, bz1, (size_t)(z0 + 1), R, r0);
                const float vnw[4] = {a_nw.x, a_nw.y, a_nw.z, a_nw.w}, vne[4] = {a_ne.x, a_ne.y, a_ne.z, a_ne.w};
                const float vsw[4] = {a_sw.x, a_sw.y, a_sw.z, a_sw.w}, vse[4] = {a_se.x, a_se.y, a_se.z, a_se.w};
                const float vl0[4] = {a_l0.x, a_l0.y, a_l0.z, a_l0.w}, vl1[4] = {a_l1.x, a_l1.y, a_l1.z, a_l1.w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    float m = 0.0f;
                    if (bx0 && by0) m += vnw[k] * nw;
                    if (bx1 && by0) m += vne[k] * ne;
                    if (bx0 && by1) m += vsw[k] * sw;
                    if (bx1 && by1) m += vse[k] * se;
                    float l = 0.0f;
                    if (bz0) l += vl0[k] * lz0;
                    if (bz1) l += vl1[k] * lz1;
                    const float prod = m * l;
                    if (REDUCE) comp += prod;
                    else out[(size_t)(f.row0[i] + r0 + k) * N + n] = prod;
                }
            }
            total += comp;
            continue;
        }
        for (uint32_t r = 0; r < f.rank[i]; r++) {
            const float* pr = P + r * plane_stride;
            const float* lr = Lq + (size_t)r * Dn;
            // gathers first, then the fixed-order accumulation
            const float v_nw = (bx0 && by0) ? pr[o_nw] : 0.0f, v_ne = (bx1 && by0) ? pr[o_nw + 1] : 0.0f;
            const float v_sw = (bx0 && by1) ? pr[o_nw + W] : 0.0f, v_se = (bx1 && by1) ? pr[o_nw + W + 1] : 0.0f;
            const float l0 = bz0 ? lr[z0] : 0.0f, l1 = bz1 ? lr[z0 + 1] : 0.0f;
            float m = 0.0f;
            if (bx0 && by0) m += v_nw * nw;
            if (bx1 && by0) m += v_ne * ne;
            if (bx0 && by1) m += v_sw * sw;
            if (bx1 && by1) m += v_se * se;
            float l = 0.0f;
            if (bz0) l += l0 * lz0;
            if (bz1) l += l1 * lz1;
            const float prod = m * l;
            if (REDUCE) comp += prod;
            else out[(size_t)(f.row0[i] + r) * N + n] = prod;
        }
        total += comp;
    }
    if (REDUCE) out[n] = total;
}

// ---- the colour features with basis_mat applied on the spot (tensoRF/network.py:149-153: basis_mat((mat * vec).T)) ----
// The products never leave the lane: out[n][c] = sum_row W[c][row] * prod[row], the arithmetic of the fp16 autocast
// nn.Linear the reference runs (operands rounded to binary16, fp32 accumulation, binary16 result).  W sits in LDS as
// fp32 [rows][kVmBasisPad]; every lane reads the same row (broadcast, four columns per ds_read_b128).  Saves the
// [sum R, N] fp32 round trip (2 x 60 MB per step at 1e5 samples), the transposing fp16 cast and the GEMM launch.
constexpr uint32_t kVmBasisPad = 32;  // output channels held per lane (basis_mat: 27)

__global__ void __launch_bounds__(256) k_vm_color_basis(const float* __restrict__ x, uint32_t N, VmFactors f,
                                                        const _Float16* __restrict__ basis, uint32_t Cb, uint32_t rows,
                                                        _Float16* __restrict__ out, const int32_t* __restrict__ n_valid) {
    extern __shared__ float vm_smem[];  // [rows][kVmBasisPad]
    if (blockIdx.x * 256 >= valid_rows(N, n_valid)) return;
    for (uint32_t e = threadIdx.x; e < rows * kVmBasisPad; e += 256) {
        const uint32_t row = e / kVmBasisPad, c = e % kVmBasisPad;
        vm_smem[e] = c < Cb ? (float)basis[(size_t)c * rows + row] : 0.0f;
    }
    __syncthreads();
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= valid_rows(N, n_valid)) return;
    const float p[3] = {x[(size_t)n * 3], x[(size_t)n * 3 + 1], x[(size_t)n * 3 + 2]};
    float acc[kVmBasisPad];
#pragma unroll
    for (uint32_t c = 0; c < kVmBasisPad; c++) acc[c] = 0.0f;
#pragma unroll 1
    for (uint32_t i = 0; i < 3; i++) {
        const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
        const float ix = unnormalize(p[f.cu[i]], f.W[i]), iy = unnormalize(p[f.cv[i]], f.H[i]), iz = unnormalize(p[f.cw[i]], f.Dn[i]);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        const float lz1 = iz - fz, lz0 = (fz + 1.0f) - iz;
        const bool okx = fabsf(ix) < 1e9f, oky = fabsf(iy) < 1e9f, okz = fabsf(iz) < 1e9f;
        const int x0 = okx ? (int)fx : -2, y0 = oky ? (int)fy : -2, z0 = okz ? (int)fz : -2;
        const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const bool by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
        const bool bz0 = z0 >= 0 && z0 < Dn, bz1 = z0 + 1 >= 0 && z0 + 1 < Dn;
        const float* P = f.plane[i];
        const float* Lq = f.line[i];
        const size_t plane_stride = (size_t)H * W;
        const int o_nw = y0 * W + x0;
        auto to_basis = [&](float prod, uint32_t r) {
            const float4* wrow = reinterpret_cast<const float4*>(vm_smem + (size_t)(f.row0[i] + r) * kVmBasisPad);
#pragma unroll
            for (uint32_t q = 0; q < kVmBasisPad / 4; q++) {
                const float4 w = wrow[q];
                acc[4 * q] = __builtin_fmaf(prod, w.x, acc[4 * q]);
                acc[4 * q + 1] = __builtin_fmaf(prod, w.y, acc[4 * q + 1]);
                acc[4 * q + 2] = __builtin_fmaf(prod, w.z, acc[4 * q + 2]);
                acc[4 * q + 3] = __builtin_fmaf(prod, w.w, acc[4 * q + 3]);
            }
        };
        if (f.plane_t[i]) {  // rank-fastest shadows (see VmFactors)
            const float* Pt = f.plane_t[i];
            const float* Ltq = f.line_t[i];
            const uint32_t R = f.rank[i];
            for (uint32_t r0 = 0; r0 < R; r0 += 4) {
                const float4 a_nw = vm_load4(Pt, bx0 && by0, (size_t)o_nw, R, r0), a_ne = vm_load4(Pt, bx1 && by0, (size_t)(o_nw + 1), R, r0);
                const float4 a_sw = vm_load4(Pt, bx0 && by1, (size_t)(o_nw + W), R, r0), a_se = vm_load4(Pt, bx1 && by1, (size_t)(o_nw + W + 1), R, r0);
                const float4 a_l0 = vm_load4(Ltq, bz0, (size_t)z0, R, r0), a_l1 = vm_load4(Ltq, bz1, (size_t)(z0 + 1), R, r0);
                const float vnw[4] = {a_nw.x, a_nw.y, a_nw.z, a_nw.w}, vne[4] = {a_ne.x, a_ne.y, a_ne.z, a_ne.w};
                const float vsw[4] = {a_sw.x, a_sw.y, a_sw.z, a_sw.w}, vse[4] = {a_se.x, a_se.y, a_se.z, a_se.w};
                const float vl0[4] = {a_l0.x, a_l0.y, a_l0.z, a_l0.w}, vl1[4] = {a_l1.x, a_l1.y, a_l1.z, a_l1.w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    float m = 0.0f;
                    if (bx0 && by0) m += vnw[k] * nw;
                    if (bx1 && by0) m += vne[k] * ne;
                    if (bx0 && by1) m += vsw[k] * sw;
                    if (bx1 && by1) m += vse[k] * se;
                    float l = 0.0f;
                    if (bz0) l += vl0[k] * lz0;
                    if (bz1) l += vl1[k] * lz1;
                    to_basis((float)(_Float16)(m * l), r0 + k);
                }
            }
            continue;
        }
        for (uint32_t r = 0; r < f.rank[i]; r++) {
            const float* pr = P + r * plane_stride;
            const float* lr = Lq + (size_t)r * Dn;
            const float v_nw = (bx0 && by0) ? pr[o_nw] : 0.0f, v_ne = (bx1 && by0) ? pr[o_nw + 1] : 0.0f;
            const float v_sw = (bx0 && by1) ? pr[o_nw + W] : 0.0f, v_se = (bx1 && by1) ? pr[o_nw + W + 1] : 0.0f;
            const float l0 = bz0 ? lr[z0] : 0.0f, l1 = bz1 ? lr[z0 + 1] : 0.0f;
            float m = 0.0f;
            if (bx0 && by0) m += v_nw * nw;
            if (bx1 && by0) m += v_ne * ne;
            if (bx0 && by1) m += v_sw * sw;
            if (bx1 && by1) m += v_se * se;
            float l = 0.0f;
            if (bz0) l += l0 * lz0;
            if (bz1) l += l1 * lz1;
            to_basis((float)(_Float16)(m * l), r);  // (the autocast Linear's fp16 input)
        }
    }
    _Float16* o = out + (size_t)n * Cb;
#pragma unroll
    for (uint32_t c = 0; c < kVmBasisPad; c++)
        if (c < Cb) o[c] = (_Float16)acc[c];
}

// ------------------------------------------------------------------ backward (parameter gradients)
// torch's grid_sample backward scatters every corner contribution with a global fp32 atomic: 2.8e8 atomics per step at
// the Lego sample count, ~13 ms at the 21 G/s such atomics retire on MI355X.  Here the points are
// binned first (the caller sorts the keys k_vm_keys produces): per plane by 8x8-cell tile, per line by 64-row chunk.
// One workgroup owns a tile: the plane values of its 9x9 cells and a zeroed accumulator live in LDS, lanes = rank
// channels; per point it re-computes line value l_r and plane value m_r, adds g_r*l_r*w_corner into the LDS accumulator
// and leaves g_r*m_r (what the line gradient needs) in `gm`; the tile is flushed with one global atomic per non-zero
// cell (cells on the +1 border belong to the neighbour as well).  The line kernel does the same over z chunks.
constexpr int kVmTile = 8, kVmTileCells = (kVmTile + 1) * (kVmTile + 1), kVmZChunk = 64;
constexpr uint32_t kVmSkip = 0x7fffffffu;

struct VmPoint {
    float nw, ne, sw, se, lz0, lz1;
    int x0, y0, z0;
    bool valid;
};

struct VmXyz { float u, v, w; };  // a point's coordinates along the plane's two axes and the line's axis
__device__ __forceinline__ VmXyz vm_load_xyz(const float* __restrict__ x, uint32_t n, const VmFactors& f, uint32_t i) {
    return VmXyz{x[(size_t)n * 3 + f.cu[i]], x[(size_t)n * 3 + f.cv[i]], x[(size_t)n * 3 + f.cw[i]]};
}
__device__ __forceinline__ VmPoint vm_locate_xyz(const VmXyz& p, const VmFactors& f, uint32_t i);
__device__ __forceinline__ VmPoint vm_locate(const float* __restrict__ x, uint32_t n, const VmFactors& f, uint32_t i) {
    return vm_locate_xyz(vm_load_xyz(x, n, f, i), f, i);
}
__device__ __forceinline__ VmPoint vm_locate_xyz(const VmXyz& p, const VmFactors& f, uint32_t i) {
    VmPoint q;
    const float px = p.u, py = p.v, pz = p.w;
    const float ix = unnormalize(px, f.W[i]), iy = unnormalize(py, f.H[i]), iz = unnormalize(pz, f.Dn[i]);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    q.nw = wx0 * wy0; q.ne = wx1 * wy0; q.sw = wx0 * wy1; q.se = wx1 * wy1;
    q.lz1 = iz - fz; q.lz0 = (fz + 1.0f) - iz;
    const bool ok = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f && fabsf(iz) < 1e9f;
    q.x0 = ok ? (int)fx : -2; q.y0 = ok ? (int)fy : -2; q.z0 = ok ? (int)fz : -2;
    // at least one corner of the plane cell and one of the line segment in range, else the point contributes nothing
    q.valid = ok && q.x0 >= -1 && q.x0 < (int)f.W[i] && q.y0 >= -1 && q.y0 < (int)f.H[i] && q.z0 >= -1 && q.z0 < (int)f.Dn[i];
    return q;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// keys [6][N]: rows 0-2 plane tile of component i, rows 3-5 line chunk; kVmSkip for points without any contribution
__global__ void __launch_bounds__(256) k_vm_keys(const float* __restrict__ x, uint32_t N, VmFactors f, int32_t* __restrict__ keys) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        const VmPoint q = vm_locate(x, n, f, i);
        const int tiles_x = ((int)f.W[i] + kVmTile - 1) / kVmTile;
        const int tk = (clampi(q.y0, 0, (int)f.H[i] - 1) / kVmTile) * tiles_x + clampi(q.x0, 0, (int)f.W[i] - 1) / kVmTile;
        const int zk = clampi(q.z0, 0, (int)f.Dn[i] - 1) / kVmZChunk;
        keys[(size_t)i * N + n] = q.valid ? tk : (int32_t)kVmSkip;
        keys[(size_t)(3 + i) * N + n] = q.valid ? zk : (int32_t)kVmSkip;
    }
}

// Counting sort of the points by (row, bin): the bins of a row are few (1,444 plane tiles, 5 line chunks at resolution 300)
// and points arrive ray by ray, so the lanes of a wave share a handful of bins — one atomic per distinct bin and wave, in
// both passes.  A point without contribution goes to the row's last bin (n_bounds - 1, behind every bin a kernel asks for).
// The order of the points INSIDE a bin is whatever the atomics make it (the backward sums a bin's points exactly, in fixed
// point: the order does not reach the result).
// the lanes of a wave that share `key`: the group's first lane (leader), the lane's rank inside the group and the group's size —
// a loop over the wave's DISTINCT keys (a handful for ray-ordered points), no memory traffic
struct BinGroup { uint32_t leader, rank, size; };
__device__ __forceinline__ BinGroup wave_bin_group(uint32_t key, uint32_t lane, bool active = true) {
    BinGroup g{0u, 0u, 0u};
    unsigned long long todo = __ballot(active);
    while (todo) {
        const uint32_t first = (uint32_t)__ffsll((long long)todo) - 1u;
        const uint32_t k = (uint32_t)__shfl((int)key, (int)first, 64);
        const unsigned long long same = __ballot(active && key == k) & todo;
        if (active && key == k) {
            g.leader = first;
            g.rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            g.size = (uint32_t)__popcll(same);
        }
        todo &= ~same;
    }
    return g;
}
// The three line rows have a handful of bins (resolution / 64 chunks + the bin of points without contribution): every wave of the
// launch would add to the same ~5 words per row — a thousand same-address atomics each, 25 us of a 30 us kernel.  A workgroup of
// sixteen waves sums them in LDS first (slot = chunk, last slot = no contribution) and sends one atomic per slot.
constexpr uint32_t kVmLineSlots = 33;  // chunks 0..31 + "no contribution"
constexpr uint32_t kVmBinThreads = 1024;
__global__ void __launch_bounds__(256) k_vm_zero_words(uint32_t* __restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
__global__ void __launch_bounds__(kVmBinThreads) k_vm_bin_count(const float* __restrict__ x, uint32_t N, VmFactors f, uint32_t n_bounds,
                                                                uint32_t nlb, uint32_t* __restrict__ keys, uint32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ n_valid) {
    __shared__ uint32_t lc[3][kVmLineSlots];
    if (threadIdx.x < 3 * kVmLineSlots) (&lc[0][0])[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n = blockIdx.x * kVmBinThreads + threadIdx.x;
    const bool live = n < N;
    const bool absent = n >= valid_rows(N, n_valid);  // rows behind the device-side count sort behind every bin, like points without a contribution
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        const VmPoint q = vm_locate(x, live ? n : 0u, f, i);
        const int tiles_x = ((int)f.W[i] + kVmTile - 1) / kVmTile;
        const uint32_t tk = (uint32_t)((clampi(q.y0, 0, (int)f.H[i] - 1) / kVmTile) * tiles_x + clampi(q.x0, 0, (int)f.W[i] - 1) / kVmTile);
        const uint32_t zk = (uint32_t)(clampi(q.z0, 0, (int)f.Dn[i] - 1) / kVmZChunk);
        const bool contributes = q.valid && !absent;
        const uint32_t kp = contributes ? tk : n_bounds - 1u, kl = contributes ? zk : n_bounds - 1u;
        if (live) {
            keys[(size_t)i * N + n] = kp;
            keys[(size_t)(3 + i) * N + n] = kl;
        }
        const BinGroup gp = wave_bin_group(kp, lane, live);
        if (live && lane == gp.leader) atomicAdd(&counts[i * n_bounds + kp], gp.size);  // (no value returned: nothing waits for it)
        const uint32_t slot = contributes ? zk : nlb;
        const BinGroup gl = wave_bin_group(slot, lane, live);
        if (live && lane == gl.leader) atomicAdd(&lc[i][slot], gl.size);
    }
    __syncthreads();
    if (threadIdx.x < 3 * kVmLineSlots) {
        const uint32_t i = threadIdx.x / kVmLineSlots, slot = threadIdx.x % kVmLineSlots;
        const uint32_t c = lc[i][slot];
        if (c && slot <= nlb) atomicAdd(&counts[(3 + i) * n_bounds + (slot == nlb ? n_bounds - 1u : slot)], c);
    }
}
// start[r][t] = points of row r in bins < t: one wave per row, lane l sums bins [l * per, (l + 1) * per) on its own (all its
// loads in flight together), one wave scan over the 64 run totals
__global__ void __launch_bounds__(384) k_vm_bin_scan(const uint32_t* __restrict__ counts, uint32_t n_bounds, int32_t* __restrict__ start) {
    const uint32_t r = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t per = (n_bounds + 63u) / 64u;
    const uint32_t t0 = lane * per;
    constexpr uint32_t kKeep = 32;  // a lane's counts stay in registers up to this run length (n_bounds <= 2,048: resolution <= 360)
    uint32_t c[kKeep];
    uint32_t total = 0;
    if (per <= kKeep) {
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++) c[j] = (j < per && t0 + j < n_bounds) ? counts[r * n_bounds + t0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++) total += c[j];
    } else {
        for (uint32_t j = 0; j < per; j++) total += t0 + j < n_bounds ? counts[r * n_bounds + t0 + j] : 0u;
    }
    uint32_t incl = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
    uint32_t run = incl - total;
    if (per <= kKeep) {
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++) {
            if (j < per && t0 + j < n_bounds) start[r * n_bounds + t0 + j] = (int32_t)run;
            run += c[j];
        }
    } else {
        for (uint32_t j = 0; j < per && t0 + j < n_bounds; j++) {
            start[r * n_bounds + t0 + j] = (int32_t)run;
            run += counts[r * n_bounds + t0 + j];
        }
    }
}
__global__ void __launch_bounds__(kVmBinThreads) k_vm_bin_scatter(const uint32_t* __restrict__ keys, uint32_t N, uint32_t n_bounds,
                                                                  uint32_t nlb, const int32_t* __restrict__ start,
                                                                  uint32_t* __restrict__ cursors, int32_t* __restrict__ perm) {
    __shared__ uint32_t lc[3][kVmLineSlots], lbase[3][kVmLineSlots];
    if (threadIdx.x < 3 * kVmLineSlots) (&lc[0][0])[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n = blockIdx.x * kVmBinThreads + threadIdx.x;
    const bool live = n < N;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t k[6], base[3], first[6], slot[3], woff[3];
    BinGroup g[6];
#pragma unroll
    for (uint32_t r = 0; r < 6; r++) {
        k[r] = live ? keys[(size_t)r * N + n] : 0u;
        first[r] = (uint32_t)start[r * n_bounds + k[r]];
    }
    // plane rows: one reservation per distinct bin of the wave, the three rows' requests in flight together
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) {
        g[r] = wave_bin_group(k[r], lane, live);
        base[r] = (live && lane == g[r].leader) ? atomicAdd(&cursors[r * n_bounds + k[r]], g[r].size) : 0u;
    }
    // line rows: the wave's offset inside the workgroup from LDS, the workgroup's reservation by one thread per slot
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) {
        slot[r] = k[3 + r] == n_bounds - 1u ? nlb : k[3 + r];
        g[3 + r] = wave_bin_group(slot[r], lane, live);
        woff[r] = (live && lane == g[3 + r].leader) ? atomicAdd(&lc[r][slot[r]], g[3 + r].size) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 3 * kVmLineSlots) {
        const uint32_t i = threadIdx.x / kVmLineSlots, sl = threadIdx.x % kVmLineSlots;
        const uint32_t c = lc[i][sl];
        lbase[i][sl] = (c && sl <= nlb) ? atomicAdd(&cursors[(3 + i) * n_bounds + (sl == nlb ? n_bounds - 1u : sl)], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 3; r++) {
        const uint32_t b = (uint32_t)__shfl((int)base[r], (int)g[r].leader, 64);
        const uint32_t w = (uint32_t)__shfl((int)woff[r], (int)g[3 + r].leader, 64);
        if (live) {
            perm[(size_t)r * N + first[r] + b + g[r].rank] = (int32_t)n;
            perm[(size_t)(3 + r) * N + first[3 + r] + lbase[r][slot[r]] + w + g[3 + r].rank] = (int32_t)n;
        }
    }
}

struct VmBackward {
    float* d_plane[3];
    float* d_line[3];
    const float* g;        // REDUCE: [N]; else [N, rows] (rows = sum of ranks), channel-contiguous per point
    float* gm;             // [N, rows], zero-initialised: g_r * m_r
    const int32_t* perm;   // [6][N] point ids sorted by key (rows as in k_vm_keys)
    const int32_t* start;  // [6][n_bounds]: first sorted position with key >= t
    uint32_t n_bounds, rows;
    // basis_mat behind the products (BASIS kernels): g = [N, Cb] fp16 gradient of the Linear's OUTPUT instead of the products'
    const _Float16* basis;   // [Cb][rows]
    const _Float16* g_out;   // [N][kVmBasisPad] (rows padded to 64 bytes: four 16-byte loads per point)
    float* d_basis;          // [Cb][rows] fp32, zero-initialised
    uint32_t Cb;
    // bit patterns of non-negative floats (monotone as uint32; NaN patterns sort above +inf), zero-initialised by the caller:
    // [0] max |g| (BASIS: max |g_out|), [1] max |line parameter|, [2] max |g_r m_r| (written by the plane kernel for the
    // line kernel), [3] BASIS: max_r sum_c |W[c][r]|
    uint32_t* bound;
    // the three line factors TRANSPOSED, [Dn][rank] each (component i at line_t_off[i]): the plane pass has one lane per rank
    // channel read a point's line value — in the parameters' [rank][Dn] layout that is one cache line per lane and load
    // (96 line requests per point, which kept the L1 busy ~4 us per round of trips); transposed it is 192 contiguous bytes
    float* line_t;
    uint32_t line_t_off[3];
    uint32_t pts_plane, pts_line;  // sorted points per workgroup
    float* found_inf;              // optional: raised when a bound is not finite (GradScaler's check made where the gradient is written)
    // staged flushes (optional, s3d_vm_backward_stage_bytes): the cells several workgroups add to — the border of a whole tile's
    // 9 x 9 window, a line chunk's 65 cells — leave as plain stores into per-tile / per-workgroup rows, and k_vm_flush_reduce adds
    // the rows in a fixed order.  Global atomics on these few, heavily shared addresses were 60 - 70 us of the colour plane pass
    // and of the colour line pass (profiles/r11_tensorf_vm.md).  nullptr: atomics as before.
    float* stage_plane;            // [3][stage_tiles][32 border cells][stage_R]
    float* stage_line;             // [3][stage_lslots][65][stage_R]
    uint32_t* plane_flag;          // [3][stage_tiles]: 1 = the tile's border rows are staged (cleared by k_vm_bound)
    uint32_t* line_flag;           // [3][stage_lslots]: chunk + 1 of the staged row block, 0 = unused
    uint32_t stage_tiles, stage_lslots, stage_R;
};
// 9 x 9 window cell -> its place among the window's 32 border cells (row 0: 0..8, row 8: 9..17, column 0 rows 1..7: 18..24,
// column 8 rows 1..7: 25..31), -1 for an inner cell
__device__ __forceinline__ int vm_border_index(int lx, int ly) {
    if (ly == 0) return lx;
    if (ly == kVmTile) return kVmTile + 1 + lx;
    if (lx == 0) return 2 * kVmTile + 1 + ly;
    if (lx == kVmTile) return 3 * kVmTile + ly;
    return -1;
}

// LDS float atomics retire at ~0.2 T/s on MI355X, LDS integer atomics at ~2.3 T/s (tools/ubench, csrc/gridencoder.hip): the
// tile accumulators are 64-bit fixed point.  The scale comes from a bound on one contribution (|g| max x |line| max for the
// plane kernel, max |g m| for the line kernel: the corner weights are <= 1) placed at 2^50: a workgroup adds at most
// kVmMaxPts = 2^12 contributions into a cell before it flushes (one per point of its range), so the 63-bit sums cannot
// overflow, and a contribution is ROUNDED TO NEAREST at 2^-50 of the call-wide bound (no truncation bias; round 5 truncated at
// 2^-40).  A region whose gradients lie 2^40 below the batch maximum still keeps 10 bits per contribution; below 2^-51 of
// the bound a contribution rounds to zero — seven orders of magnitude under what an fp32 sum of the same cell would still
// resolve next to a contribution of the bound's size.  [Measured and rejected, profiles/r11_tensorf.md: an exact bypass for
// contributions under 2^12 quanta (fp32 atomics straight to the gradient) — the cold block costs the colour plane kernel a
// third of its speed whatever the threshold: 339 -> 497 us.]  Integer adds commute: a tile's sum does not depend on the
// order in which its points arrive.  The sums leave as fp32 (plain stores where the segment owns the cell, atomics otherwise).
constexpr int kVmFixBits = 50;
constexpr uint32_t kVmMaxPts = 4096;
__device__ __forceinline__ long long vm_fixed(float v, float scale) {
#ifdef S3D_VM_FIXED_TWO_LIMB  // rounds 5 / 6a: hi = trunc(t / 2^24), lo = t - hi 2^24 (exact), q = hi 2^24 + rn(lo) — the same integer, ~12 vector instructions
    const float t = v * scale;                                  // (power-of-two scale: exact)
    const float hi = truncf(t * 5.9604644775390625e-08f);       // t / 2^24, |hi| < 2^27
    const float lo = __builtin_fmaf(-hi, 16777216.0f, t);       // exact remainder, |lo| < 2^24
    return (long long)(int)hi * 16777216ll + (long long)__float2int_rn(lo);
#else
    // rn(t) for |t| < 2^51 as the low mantissa bits of t + 1.5 x 2^52 (the sum has an ulp of 1: the double add IS the rounding to
    // nearest even): one conversion, one double add, one 64-bit subtract — the same integer as the two-limb form.  Colour plane
    // pass 252 -> 234 us, colour line 38.7 -> 33.9 us (profiles/r11_tensorf_vm.md; an eighth of the LDS adds changes nothing there:
    // the conversions, not the LDS atomics, are what the fixed-point sums cost).
    const double d = (double)(v * scale) + 6755399441055744.0;  // (power-of-two scale: exact; |v * scale| <= 2^50 by the bound)
    return __double_as_longlong(d) - 0x4338000000000000ll;
#endif
}
__device__ __forceinline__ void vm_lds_add(long long* a, long long q) {
#ifdef S3D_VM_EXP_NOLDSADD  // timing experiment only (wrong sums): what the LDS atomics cost
    if (q == 0x7fffffffffffffffll) *a = q;
#else
    atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)q);
#endif
}
__device__ __forceinline__ void vm_flush_add(float* dst, float v) {
#ifdef S3D_VM_EXP_PLAINFLUSH  // timing experiment only (wrong sums where segments share cells): what the global atomics cost
    *dst = v;
#else
    atomicAdd(dst, v);
#endif
}
// scale for a bound given as two (three) factors' bit patterns; returns false when there is nothing to add (a zero factor)
// and sets `poison` when a factor is not finite (the gradient is then non-finite as the float sums would be)
__device__ __forceinline__ bool vm_scale(float bound, float& scale, float& inv, bool& poison) {
    poison = !(bound <= 3.402823466e38f);
    if (poison || !(bound > 0.0f)) return false;
    int e;
    (void)frexpf(bound, &e);  // bound < 2^e
    e = e < -70 ? -70 : e;    // (keeps the scale finite for vanishing bounds: 2^(50 + 70))
    scale = ldexpf(1.0f, kVmFixBits - e);
    inv = ldexpf(1.0f, e - kVmFixBits);
    return true;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t nan_aware_bits(float m, bool bad) { return bad ? 0x7fc00000u : __float_as_uint(m); }

// bounds of one backward call: [0] over the gradient tensor (fp32 `g`, n_g values, or fp16 `g16`, n_g16 values), [1] over the
// three line factors, [3] over the columns of basis_mat
__global__ void __launch_bounds__(256) k_vm_bound(const float* __restrict__ g, size_t n_g, const _Float16* __restrict__ g16, size_t n_g16,
                                                   VmFactors f, const _Float16* __restrict__ basis, uint32_t Cb, uint32_t rows,
                                                   uint32_t* __restrict__ bound, float* __restrict__ line_t,
                                                   uint32_t* __restrict__ stage_flags, uint32_t n_stage_flags,
                                                   uint32_t N, const int32_t* __restrict__ n_valid) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    if (n_valid) {  // (padded sample batch: the gradient rows behind the count are not looked at — they may hold anything)
        const size_t nv = valid_rows(N, n_valid);
        n_g = n_g / N * nv;
        n_g16 = n_g16 / N * nv;
    }
    for (size_t k = tid; k < n_stage_flags; k += nt) stage_flags[k] = 0u;  // (staged flushes: nothing staged yet)
    float m = 0.0f;
    bool bad = false;
    if (g) {
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (size_t k = tid; k < n_g / 4; k += nt) {
            const float4 v = g4[k];
            const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            bad |= !(fabsf(v.x) <= 3.4e38f) || !(fabsf(v.y) <= 3.4e38f) || !(fabsf(v.z) <= 3.4e38f) || !(fabsf(v.w) <= 3.4e38f);
            m = fmaxf(m, a);
        }
        for (size_t k = (n_g / 4) * 4 + tid; k < n_g; k += nt) { bad |= !(fabsf(g[k]) <= 3.4e38f); m = fmaxf(m, fabsf(g[k])); }
    }
    if (g16) {
        typedef _Float16 half8v __attribute__((ext_vector_type(8)));
        const half8v* g8 = reinterpret_cast<const half8v*>(g16);
        for (size_t k = tid; k < n_g16 / 8; k += nt) {
            const half8v v = g8[k];
#pragma unroll
            for (int j = 0; j < 8; j++) { const float a = fabsf((float)v[j]); bad |= !(a <= 65504.0f); m = fmaxf(m, a); }
        }
    }
    // (one atomic per BLOCK: thousands of waves on one word serialise — 46 us per call with one per wave)
    __shared__ uint32_t blk[3];
    if (threadIdx.x < 3) blk[threadIdx.x] = 0u;
    __syncthreads();
    m = wave_max(m);
    bad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63) == 0 && (m > 0.0f || bad)) atomicMax(&blk[0], nan_aware_bits(m, bad));
    // line factors
    float lm = 0.0f;
    bool lbad = false;
    size_t base = 0;
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        const size_t n = (size_t)f.rank[i] * f.Dn[i];
        for (size_t k = tid; k < n; k += nt) {
            const float v = f.line[i][k], a = fabsf(v);
            lbad |= !(a <= 3.4e38f);
            lm = fmaxf(lm, a);
            const size_t r = k / f.Dn[i], z = k - r * f.Dn[i];
            line_t[base + z * f.rank[i] + r] = v;  // (the transposed copy the plane pass reads)
        }
        base += n;
    }
    lm = wave_max(lm);
    lbad = __ballot(lbad) != 0ull;
    if ((threadIdx.x & 63) == 0 && (lm > 0.0f || lbad)) atomicMax(&blk[1], nan_aware_bits(lm, lbad));
    __syncthreads();
    if (threadIdx.x == 0 && blk[0]) atomicMax(bound + 0, blk[0]);
    if (threadIdx.x == 1 && blk[1]) atomicMax(bound + 1, blk[1]);
    if (basis && blockIdx.x == 0) {
        float cm = 0.0f;
        bool cbad = false;
        for (uint32_t r = threadIdx.x; r < rows; r += 256) {
            float s = 0.0f;
            for (uint32_t c = 0; c < Cb; c++) s += fabsf((float)basis[(size_t)c * rows + r]);
            cbad |= !(s <= 3.4e38f);
            cm = fmaxf(cm, s);
        }
        cm = wave_max(cm);
        cbad = __ballot(cbad) != 0ull;
        if ((threadIdx.x & 63) == 0 && (cm > 0.0f || cbad)) atomicMax(bound + 3, nan_aware_bits(cm, cbad));
    }
}

// the range [begin, end) of workgroup `blk` moved to the boundaries of small tiles / chunks (k_vm_plane_backward_mm below: a unit of at most `whole_max` points is processed whole by the workgroup its first position falls to); t = the unit of `begin`
__device__ __forceinline__ bool vm_aligned_range(const int32_t* __restrict__ st, int nunits, uint32_t blk, uint32_t pts, uint32_t whole_max,
                                                 uint32_t valid_end, uint32_t& begin, uint32_t& end, int& t) {
    begin = blk * pts;
    if (begin >= valid_end) return false;
    end = begin + pts < valid_end ? begin + pts : valid_end;
    // the last u with st[u] <= pos (empty units repeat their neighbour's start; st[0] = 0).  A 64-ary search, the lanes of the wave
    // testing 64 positions at once: two rounds of loads for up to 4,096 units instead of a dozen dependent ones (the two searches
    // and the walk over empty tiles were ~20 us of latency per workgroup: profiles/r11_tensorf_vm.md, "skeleton")
    auto unit_of = [&](uint32_t pos) {
        const int lane = (int)(threadIdx.x & 63u);
        int lo = 0, n = nunits;  // the answer lies in [lo, lo + n)
        while (n > 1) {
            const int step = (n + 63) / 64;
            const int u = lo + lane * step;
            const bool le = lane * step < n && (uint32_t)st[u] <= pos;
            const int k = __popcll(__ballot(le));  // st is non-decreasing: the lanes that pass form a prefix (k >= 1: st[lo] <= pos)
            lo += (k - 1) * step;
            n = n - (k - 1) * step < step ? n - (k - 1) * step : step;
        }
        return lo;
    };
    t = unit_of(begin);
    if ((uint32_t)st[t] < begin && (uint32_t)st[t + 1] - (uint32_t)st[t] <= whole_max) begin = (uint32_t)st[t + 1];  // an earlier workgroup's
    if (end < valid_end) {
        const int te = unit_of(end);
        if ((uint32_t)st[te] < end && (uint32_t)st[te + 1] - (uint32_t)st[te] <= whole_max) end = (uint32_t)st[te + 1];  // whole, and mine (or nobody's: begin == end)
    }
    return begin < end;
}
// One workgroup = `pts_plane` consecutive positions of plane i's sorted point order (a hot tile is shared by as many
// workgroups as its points fill, a stretch of sparse tiles is walked by one): per tile segment the 9x9 cells' plane values
// and a cleared accumulator live in LDS, lanes = rank channels, eight waves = eight points (RP = 64) in flight.
// RP = lanes per point (16 or 64 >= rank); a wave handles 64 / RP points per trip
// BASIS (colour features behind basis_mat, REDUCE = false): lane r derives its product gradient from the Linear's output
// gradient, g_r = sum_c W[c][row0 + r] * g_out[n][c] (its column of W lives in registers, the point's Cb gradients are the
// same for the whole 64-lane group), and accumulates basis_mat's own gradient dW[c][row0 + r] += g_out[n][c] * prod_r in
// registers — flushed with one atomic per (c, r) and workgroup.  Every (point, component) is visited by exactly one lane group
// of this kernel, so the three components' launches cover dW once.
constexpr uint32_t kVmBwdThreads = 512;
template <int RP, bool REDUCE, bool BASIS = false>
__global__ void __launch_bounds__(kVmBwdThreads) k_vm_plane_backward(const float* __restrict__ x, uint32_t N, VmFactors f, VmBackward b) {
    static_assert(!BASIS || (RP == 64 && !REDUCE), "basis_mat sits behind the 48-rank colour products");
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem_raw[];
    const uint32_t i = blockIdx.y;
    const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
    const uint32_t R = f.rank[i];
    const int tiles_x = (W + kVmTile - 1) / kVmTile, tiles_y = (H + kVmTile - 1) / kVmTile;
    const int ntiles = tiles_x * tiles_y;
    const int32_t* st = b.start + (size_t)i * b.n_bounds;
    const uint32_t valid_end = (uint32_t)st[ntiles];  // (points without a contribution sort behind every tile)
    float scale = 1.0f, inv = 1.0f;
    bool poison;
    float bound = __uint_as_float(b.bound[0]) * __uint_as_float(b.bound[1]);
    if constexpr (BASIS) bound *= __uint_as_float(b.bound[3]);
    if (!vm_scale(bound, scale, inv, poison)) {
        // (nothing to add: gm stays zero.  A non-finite factor poisons the plane gradient like the float sums would)
        if (poison && blockIdx.x == 0 && threadIdx.x == 0) {
            b.d_plane[i][0] = NAN;
            if (b.found_inf) *b.found_inf = 1.0f;
        }
        return;
    }
    // (ranges moved to tile boundaries, as in k_vm_plane_backward_mm: tiles of up to 2 x pts_plane points are whole)
    uint32_t begin, end;
    int t;
    if (!vm_aligned_range(st, ntiles, blockIdx.x, b.pts_plane, 2 * b.pts_plane < kVmMaxPts ? 2 * b.pts_plane : kVmMaxPts, valid_end, begin, end, t)) return;
    long long* acc = reinterpret_cast<long long*>(vm_smem_raw);                   // [81][R] gradient accumulator (fixed point)
    float* pv = reinterpret_cast<float*>(acc + kVmTileCells * R);                 // [81][R] plane values
    const float* P = f.plane[i];
    const size_t plane_stride = (size_t)H * W;
    constexpr uint32_t PPW = 64 / RP;  // points per wave trip
    constexpr uint32_t NWV = kVmBwdThreads / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sub = lane / RP, r = lane % RP;
    const int32_t* perm = b.perm + (size_t)i * N;
    const float* Lt = b.line_t + b.line_t_off[i];  // [Dn][R]
    float wcol[BASIS ? kVmBasisPad : 1], dw[BASIS ? kVmBasisPad : 1];
    if constexpr (BASIS) {
#pragma unroll
        for (uint32_t c = 0; c < kVmBasisPad; c++) {
            wcol[c] = (c < b.Cb && r < R) ? (float)b.basis[(size_t)c * b.rows + f.row0[i] + r] : 0.0f;
            dw[c] = 0.0f;
        }
    }
    for (uint32_t e = threadIdx.x; e < kVmTileCells * R; e += kVmBwdThreads) acc[e] = 0ll;
    float gm_max = 0.0f;
    uint32_t pos = begin;
    while (pos < end) {
        while ((uint32_t)st[t + 1] <= pos) t++;
        const uint32_t seg_end = (uint32_t)st[t + 1] < end ? (uint32_t)st[t + 1] : end;
        const int cx0 = (t % tiles_x) * kVmTile, cy0 = (t / tiles_x) * kVmTile;
        // (cell fastest: 9-float row segments of one channel.  Four values per lane are requested before the first one is parked:
        //  one element per loop pass waited for its own load each time — 56 of the colour plane pass's 234 us)
        if (const float* Pt = f.plane_t[i]) {
            // rank-fastest shadow: a cell's R values are one contiguous run, and so is the cell's row of `pv` (element e of the
            // window IS pv[e]) — 81 runs of R x 4 bytes instead of 81 x R four-byte words from R planes H * W * 4 bytes apart
            for (uint32_t e0 = threadIdx.x; e0 < kVmTileCells * R; e0 += 4 * kVmBwdThreads) {
                float v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t e = e0 + u * kVmBwdThreads;
                    const uint32_t c = e / R, rr = e % R;
                    const int cy = cy0 + (int)(c / (kVmTile + 1)), cx = cx0 + (int)(c % (kVmTile + 1));
                    v[u] = (e < kVmTileCells * R && cx < W && cy < H) ? Pt[((size_t)cy * W + cx) * R + rr] : 0.0f;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t e = e0 + u * kVmBwdThreads;
                    if (e < kVmTileCells * R) pv[e] = v[u];
                }
            }
        } else
        for (uint32_t e0 = threadIdx.x; e0 < kVmTileCells * R; e0 += 4 * kVmBwdThreads) {
            float v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * kVmBwdThreads;
                const uint32_t rr = e / kVmTileCells, c = e % kVmTileCells;
                const int cy = cy0 + (int)(c / (kVmTile + 1)), cx = cx0 + (int)(c % (kVmTile + 1));
                v[u] = (e < kVmTileCells * R && cx < W && cy < H) ? P[rr * plane_stride + (size_t)cy * W + cx] : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * kVmBwdThreads;
                if (e < kVmTileCells * R) pv[(e % kVmTileCells) * R + e / kVmTileCells] = v[u];
            }
        }
        __syncthreads();  // plane values in place; accumulator clear (start of the kernel / previous flush)
        // A trip is a chain of dependent loads (sorted position -> point id -> coordinates -> line values / gradients): the id
        // of the trip after next and the coordinates of the next trip are requested before this trip's arithmetic — with one
        // chain per trip a workgroup's 64 - 128 trips of ~2.5 us each WERE the kernel's time
        constexpr uint32_t STEP = NWV * PPW;
        const uint32_t k0 = pos + wave * PPW + sub;
        uint32_t n_nx = k0 < seg_end ? (uint32_t)perm[k0] : 0u;
        uint32_t n_nx2 = k0 + STEP < seg_end ? (uint32_t)perm[k0 + STEP] : 0u;
        VmXyz p_nx = k0 < seg_end ? vm_load_xyz(x, n_nx, f, i) : VmXyz{0.0f, 0.0f, 0.0f};
        for (uint32_t k = k0; k < seg_end; k += STEP) {
            uint32_t n = n_nx;
            const VmXyz p_cur = p_nx;
            n_nx = n_nx2;
            if (k + STEP < seg_end) p_nx = vm_load_xyz(x, n_nx, f, i);
            if (k + 2 * STEP < seg_end) n_nx2 = (uint32_t)perm[k + 2 * STEP];
            if constexpr (BASIS) n = __builtin_amdgcn_readfirstlane(n);  // (RP = 64: one point per wave trip)
            const VmPoint q = vm_locate_xyz(p_cur, f, i);
            if (r >= R) continue;
            float g;
            float go[BASIS ? kVmBasisPad : 1];
            if constexpr (BASIS) {
                typedef _Float16 half8v __attribute__((ext_vector_type(8)));
                const half8v* gp = reinterpret_cast<const half8v*>(b.g_out + (size_t)n * kVmBasisPad);
                g = 0.0f;
#pragma unroll
                for (uint32_t qq = 0; qq < kVmBasisPad / 8; qq++) {
                    const half8v h = gp[qq];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t c = 8 * qq + j;
                        go[c] = (float)h[j];  // (columns behind Cb are the caller's zero padding)
                        g = __builtin_fmaf(go[c], wcol[c], g);
                    }
                }
            } else {
                g = REDUCE ? b.g[n] : b.g[(size_t)n * b.rows + f.row0[i] + r];
            }
            const bool bz0 = q.z0 >= 0 && q.z0 < Dn, bz1 = q.z0 + 1 >= 0 && q.z0 + 1 < Dn;
            float l = 0.0f;
            if (bz0) l += Lt[(size_t)q.z0 * R + r] * q.lz0;
            if (bz1) l += Lt[(size_t)(q.z0 + 1) * R + r] * q.lz1;
            const int lx = q.x0 - cx0, ly = q.y0 - cy0;  // nw corner inside the tile's 9x9 window: -1 .. 7
            const bool bx0 = q.x0 >= 0 && q.x0 < W, bx1 = q.x0 + 1 >= 0 && q.x0 + 1 < W;
            const bool by0 = q.y0 >= 0 && q.y0 < H, by1 = q.y0 + 1 >= 0 && q.y0 + 1 < H;
            const int c_nw = ly * (kVmTile + 1) + lx;
            const float gl = g * l;
            float m = 0.0f;
            if (bx0 && by0) { m += pv[c_nw * R + r] * q.nw; vm_lds_add(&acc[c_nw * R + r], vm_fixed(gl * q.nw, scale)); }
            if (bx1 && by0) { m += pv[(c_nw + 1) * R + r] * q.ne; vm_lds_add(&acc[(c_nw + 1) * R + r], vm_fixed(gl * q.ne, scale)); }
            if (bx0 && by1) { m += pv[(c_nw + kVmTile + 1) * R + r] * q.sw; vm_lds_add(&acc[(c_nw + kVmTile + 1) * R + r], vm_fixed(gl * q.sw, scale)); }
            if (bx1 && by1) { m += pv[(c_nw + kVmTile + 2) * R + r] * q.se; vm_lds_add(&acc[(c_nw + kVmTile + 2) * R + r], vm_fixed(gl * q.se, scale)); }
            const float gmv = g * m;
            b.gm[(size_t)n * b.rows + f.row0[i] + r] = gmv;
            gm_max = fmaxf(gm_max, fabsf(gmv));
            if constexpr (BASIS) {
                const float prod = (float)(_Float16)(m * l);  // (the Linear's fp16 input, as in the forward)
#pragma unroll
                for (uint32_t c = 0; c < kVmBasisPad; c++) dw[c] = __builtin_fmaf(go[c], prod, dw[c]);
            }
        }
        __syncthreads();
        float* dP = b.d_plane[i];
        // A segment that holds ALL points of its tile is the only writer of the tile's inner 7 x 7 cells (the window's first
        // row / column are the left / upper neighbours' border, its last row / column the right / lower neighbours' first):
        // those leave as plain stores into the zero-initialised gradient — 1,536 instead of 3,888 global atomics per tile at
        // rank 48, and global atomics (~21 G/s chip-wide) are what this kernel waits for
        const bool whole = pos == (uint32_t)st[t] && seg_end == (uint32_t)st[t + 1];
        const bool staged = whole && b.stage_plane != nullptr;  // border cells of a whole tile: plain stores into the tile's staging rows
        float* stg = staged ? b.stage_plane + ((size_t)i * b.stage_tiles + (size_t)t) * 32 * b.stage_R : nullptr;
        for (uint32_t e = threadIdx.x; e < kVmTileCells * R; e += kVmBwdThreads) {
            const uint32_t rr = e / kVmTileCells, c = e % kVmTileCells;
            const int ly = (int)(c / (kVmTile + 1)), lx = (int)(c % (kVmTile + 1));
            const int cy = cy0 + ly, cx = cx0 + lx;
            const long long qv = acc[c * R + rr];
            const int bi = vm_border_index(lx, ly);
            if (staged && bi >= 0) {  // (every border cell is written, zeros included: the rows are not initialised)
                if (qv != 0ll) acc[c * R + rr] = 0ll;
                stg[(size_t)bi * b.stage_R + rr] = (float)qv * inv;
                continue;
            }
            if (qv != 0ll) {
                acc[c * R + rr] = 0ll;  // cleared behind the read: the next segment starts from zeros
                if (cx < W && cy < H) {
                    float* dst = &dP[rr * plane_stride + (size_t)cy * W + cx];
                    if (whole && bi < 0) *dst = (float)qv * inv;
                    else vm_flush_add(dst, (float)qv * inv);
                }
            }
        }
        if (staged && threadIdx.x == 0) b.plane_flag[(size_t)i * b.stage_tiles + (size_t)t] = 1u;
        pos = seg_end;
        // (the next segment's plane values are written before its barrier; the flush only touches `acc`)
    }
    gm_max = wave_max(gm_max);
    if (lane == 0 && gm_max > 0.0f) atomicMax(b.bound + 2, __float_as_uint(gm_max));
    if constexpr (BASIS) {
        // basis_mat's gradient: the eight waves' register sums are added in LDS first (fixed order), then ONE global atomic per
        // (c, r) and workgroup — per wave it was eight times as many, and global atomics retire at 21 G/s chip-wide
        float* red = reinterpret_cast<float*>(vm_smem_raw);  // [NWV][16][64] (the host sizes the allocation for it)
        constexpr uint32_t HC = kVmBasisPad / 2;
#pragma unroll
        for (uint32_t half = 0; half < 2; half++) {
            __syncthreads();  // accumulator / previous half no longer read
#pragma unroll
            for (uint32_t c = 0; c < HC; c++) red[(wave * HC + c) * 64 + lane] = dw[half * HC + c];
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < HC * 64; e += kVmBwdThreads) {
                const uint32_t c = half * HC + e / 64, rr = e % 64;
                float sum = 0.0f;
#pragma unroll
                for (uint32_t w = 0; w < NWV; w++) sum += red[(w * HC + e / 64) * 64 + rr];
                if (rr < R && c < b.Cb && sum != 0.0f) atomicAdd(&b.d_basis[(size_t)c * b.rows + f.row0[i] + rr], sum);
            }
        }
    }
}

// ---- plane backward, ranks a multiple of 16: 32 points per wave trip, matrix cores for basis_mat ----
// The kernel above gives a wave ONE point (lanes = rank channels): at rank 48 a quarter of the lanes idle and, behind basis_mat,
// every lane walks two 32-term dot products per point on the VALU (277 vector instructions per point and wave,
// profiles/r10_tensorf.md; with every atomic switched off the kernel still takes 265 of its 331 us, profiles/r11_tensorf_vm.md).
// Here a wave takes 32 sorted points of the segment per trip.  Lane (a = lane & 15, q4 = lane >> 4):
//   1. lanes 0..31 locate one point each and leave a 48-byte record (id, corner flags, window cell, line cell, six weights) in
//      the wave's LDS scratch; the ids / coordinates of the NEXT trip are requested before this trip's arithmetic.
//   2. basis_mat (MODE 2): G[p][r] = sum_c g_out[p][c] W[c][r] on v_mfma_f32_16x16x32_f16 — A = the points' g_out rows (one
//      16-byte load per lane, channels 8 q4 ..; row a of block pb is the trip's point 8 (a >> 2) + 4 pb + (a & 3)), B = W^T block
//      rb (registers, loaded once), K = 32 = padded Cb.  The result arrives as D[row 4 q4 + e][col = rank a]: lane (a, q4) owns
//      rank channel 16 rb + a of the points 8 q4 + 4 pb + e — EIGHT CONSECUTIVE sorted positions — so in everything that follows
//      the sixteen lanes of a group work on ONE point and sixteen ADJACENT rank channels: line values, plane values, the
//      fixed-point LDS adds and the g m stores are 64 contiguous bytes (128 for the 64-bit accumulators) per group, four points
//      (eight positions apart) per instruction.  [The first 32-point arrangement had points along the lanes and four rank
//      channels per lane: sixteen different window cells per LDS instruction at a cell stride of 96 words put 64 lanes on 16
//      banks — 422 us, 220 of them in the LDS adds, against 331 us for the kernel above; profiles/r11_tensorf_vm.md.]
//   3. dW[c][r] += sum_p g_out[p][c] prod[p][r]: K = the trip's 32 points in their sorted order, k = 8 q4 + j <-> the lane's own
//      eight points, which makes the B operand exactly the eight fp16 products the lane has just computed; only g_out^T crosses
//      the wave's LDS scratch.  fp32 accumulators stay in registers until the end of the kernel.
// A workgroup's range of the sorted order is moved to tile boundaries: a tile of at most 2 x `pts_plane` points (every tile of
// the Lego batches: ~450 occupied tiles per plane, 240 points on average, the largest ~900) is processed whole by the workgroup
// its first position falls to, so its inner 7 x 7 cells leave as plain stores (2,352 of the 3,888 global atomics of a tile flush
// at rank 48 go away; larger tiles are split at the nominal boundaries as before).
// Same fixed-point accumulation, flush and bound words as the kernel above; MODE 0 / 1 (no basis_mat; g [N] / [N, rows]).
typedef _Float16 vm_half8 __attribute__((ext_vector_type(8)));
typedef float vm_float4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kVmMmThreads = 512, kVmMmWaves = kVmMmThreads / 64;
constexpr uint32_t kVmMmRec = 12;      // words per point record (three 16-byte reads)
constexpr uint32_t kVmMmStride = 40;   // halves per row of the g_out^T scratch (32 points + pad: rows start 16-byte aligned)
constexpr uint32_t kVmFlagX0 = 1, kVmFlagX1 = 2, kVmFlagY0 = 4, kVmFlagY1 = 8, kVmFlagZ0 = 16, kVmFlagZ1 = 32, kVmFlagLive = 64;
__host__ __device__ constexpr size_t vm_mm_smem(uint32_t RB, bool basis) {
    const size_t R = 16 * RB;
    const size_t s = (size_t)kVmTileCells * R * 12 + (size_t)kVmMmWaves * (32 * kVmMmRec * 4 + (basis ? 32 * kVmMmStride * 2 : 0));
    const size_t red = basis ? (size_t)kVmMmWaves * 2 * RB * 256 * 4 : 0;  // the end-of-kernel sum of basis_mat's gradient
    return s > red ? s : red;
}
__device__ __forceinline__ void vm_wave_lds_fence() {  // this wave's LDS writes before its later LDS reads (in order in hardware; this pins the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int RB, int MODE>
__global__ void __launch_bounds__(kVmMmThreads, 4) k_vm_plane_backward_mm(const float* __restrict__ x, uint32_t N, VmFactors f, VmBackward b) {
    constexpr bool BASIS = MODE == 2;
    constexpr uint32_t R = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem_raw[];
    const uint32_t i = blockIdx.y;
    const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
    const int tiles_x = (W + kVmTile - 1) / kVmTile, tiles_y = (H + kVmTile - 1) / kVmTile;
    const int ntiles = tiles_x * tiles_y;
    const int32_t* __restrict__ st = b.start + (size_t)i * b.n_bounds;
    const uint32_t valid_end = (uint32_t)st[ntiles];
    float scale = 1.0f, inv = 1.0f;
    bool poison;
    float bound = __uint_as_float(b.bound[0]) * __uint_as_float(b.bound[1]);
    if constexpr (BASIS) bound *= __uint_as_float(b.bound[3]);
    if (!vm_scale(bound, scale, inv, poison)) {
        if (poison && blockIdx.x == 0 && threadIdx.x == 0) {
            b.d_plane[i][0] = NAN;
            if (b.found_inf) *b.found_inf = 1.0f;
        }
        return;
    }
    uint32_t begin, end;
    int t;
    if (!vm_aligned_range(st, ntiles, blockIdx.x, b.pts_plane, 2 * b.pts_plane < kVmMaxPts ? 2 * b.pts_plane : kVmMaxPts, valid_end, begin, end, t)) return;
    long long* acc = reinterpret_cast<long long*>(vm_smem_raw);                  // [81][R]
    float* pv = reinterpret_cast<float*>(acc + kVmTileCells * R);                // [81][R]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t a = lane & 15, q4 = lane >> 4, pl = lane & 31;
    uint32_t* rec = reinterpret_cast<uint32_t*>(pv + kVmTileCells * R) + (size_t)wave * 32 * kVmMmRec;  // [32][kVmMmRec]
    _Float16* sc = reinterpret_cast<_Float16*>(reinterpret_cast<uint32_t*>(pv + kVmTileCells * R) + (size_t)kVmMmWaves * 32 * kVmMmRec) +
                   (size_t)wave * 32 * kVmMmStride;                                                     // [32 channels][kVmMmStride]
    const float* __restrict__ P = f.plane[i];
    const size_t plane_stride = (size_t)H * W;
    const int32_t* __restrict__ perm = b.perm + (size_t)i * N;
    const float* __restrict__ Lt = b.line_t + b.line_t_off[i];  // [Dn][R]
    float* __restrict__ gm = b.gm;
    const uint32_t row0 = f.row0[i];
    vm_half8 wb[BASIS ? RB : 1];                      // B operand of G: k = c = 8 q4 + j, col = rank 16 rb + a
    vm_float4 dwacc[BASIS ? 2 : 1][BASIS ? RB : 1];   // dW[c = 16 cb + 4 q4 + e][r = 16 rb + a]
    if constexpr (BASIS) {
#pragma unroll
        for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t c = 8 * q4 + j;
                wb[rb][j] = c < b.Cb ? b.basis[(size_t)c * b.rows + row0 + 16 * rb + a] : (_Float16)0.0f;
            }
#pragma unroll
            for (uint32_t cb = 0; cb < 2; cb++) dwacc[cb][rb] = vm_float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    for (uint32_t e = threadIdx.x; e < kVmTileCells * R; e += kVmMmThreads) acc[e] = 0ll;
    float gm_max = 0.0f;
    uint32_t pos = begin;
    while (pos < end) {
        while ((uint32_t)st[t + 1] <= pos) t++;
        const uint32_t seg_end = (uint32_t)st[t + 1] < end ? (uint32_t)st[t + 1] : end;
        const int cx0 = (t % tiles_x) * kVmTile, cy0 = (t / tiles_x) * kVmTile;
        // the first trip's ids and coordinates are requested before the tile's plane values
        constexpr uint32_t STEP = kVmMmWaves * 32;
        const uint32_t kf = pos + wave * 32 + pl;
        uint32_t n_nx = kf < seg_end ? (uint32_t)perm[kf] : 0u;
        uint32_t n_nx2 = kf + STEP < seg_end ? (uint32_t)perm[kf + STEP] : 0u;
        VmXyz p_nx = kf < seg_end ? vm_load_xyz(x, n_nx, f, i) : VmXyz{0.0f, 0.0f, 0.0f};
        // (cell fastest: 9-float row segments of one channel.  Four values per lane are requested before the first one is parked:
        //  one element per loop pass waited for its own load each time — 56 of the colour plane pass's 234 us)
        if (const float* Pt = f.plane_t[i]) {
            // rank-fastest shadow: a cell's R values are one contiguous run, and so is the cell's row of `pv` (element e of the
            // window IS pv[e]) — 81 runs of R x 4 bytes instead of 81 x R four-byte words from R planes H * W * 4 bytes apart
            for (uint32_t e0 = threadIdx.x; e0 < kVmTileCells * R; e0 += 4 * kVmMmThreads) {
                float v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t e = e0 + u * kVmMmThreads;
                    const uint32_t c = e / R, rr = e % R;
                    const int cy = cy0 + (int)(c / (kVmTile + 1)), cx = cx0 + (int)(c % (kVmTile + 1));
                    v[u] = (e < kVmTileCells * R && cx < W && cy < H) ? Pt[((size_t)cy * W + cx) * R + rr] : 0.0f;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t e = e0 + u * kVmMmThreads;
                    if (e < kVmTileCells * R) pv[e] = v[u];
                }
            }
        } else
        for (uint32_t e0 = threadIdx.x; e0 < kVmTileCells * R; e0 += 4 * kVmMmThreads) {
            float v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * kVmMmThreads;
                const uint32_t rr = e / kVmTileCells, c = e % kVmTileCells;
                const int cy = cy0 + (int)(c / (kVmTile + 1)), cx = cx0 + (int)(c % (kVmTile + 1));
                v[u] = (e < kVmTileCells * R && cx < W && cy < H) ? P[rr * plane_stride + (size_t)cy * W + cx] : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * kVmMmThreads;
                if (e < kVmTileCells * R) pv[(e % kVmTileCells) * R + e / kVmTileCells] = v[u];
            }
        }
        __syncthreads();  // plane values in place; accumulator clear (start of the kernel / previous flush)
        for (uint32_t k0 = pos + wave * 32; k0 < seg_end; k0 += STEP) {
            // ---- 1. one point per lane (lanes 32..63 mirror 0..31): record
            const uint32_t k = k0 + pl;
            const bool okp = k < seg_end;
            const uint32_t n = n_nx;
            const VmXyz p_cur = p_nx;
            n_nx = n_nx2;
            if (k + STEP < seg_end) p_nx = vm_load_xyz(x, n_nx, f, i);
            if (k + 2 * STEP < seg_end) n_nx2 = (uint32_t)perm[k + 2 * STEP];
            {
                const VmPoint q = vm_locate_xyz(p_cur, f, i);
                uint32_t fl = kVmFlagLive;
                if (q.x0 >= 0 && q.x0 < W) fl |= kVmFlagX0;
                if (q.x0 + 1 >= 0 && q.x0 + 1 < W) fl |= kVmFlagX1;
                if (q.y0 >= 0 && q.y0 < H) fl |= kVmFlagY0;
                if (q.y0 + 1 >= 0 && q.y0 + 1 < H) fl |= kVmFlagY1;
                if (q.z0 >= 0 && q.z0 < Dn) fl |= kVmFlagZ0;
                if (q.z0 + 1 >= 0 && q.z0 + 1 < Dn) fl |= kVmFlagZ1;
                const int c_nw = (q.y0 - cy0) * (kVmTile + 1) + (q.x0 - cx0);  // nw corner inside the tile's 9x9 window: -1 .. 7 per axis
                if (lane < 32) {
                    uint4* r4 = reinterpret_cast<uint4*>(rec + pl * kVmMmRec);
                    r4[0] = make_uint4(okp ? n : 0u, okp ? fl : 0u, (uint32_t)c_nw, (uint32_t)q.z0);  // (absent rows: no flag set, id 0)
                    r4[1] = make_uint4(__float_as_uint(q.nw), __float_as_uint(q.ne), __float_as_uint(q.sw), __float_as_uint(q.se));
                    r4[2] = make_uint4(__float_as_uint(q.lz0), __float_as_uint(q.lz1), 0u, 0u);
                }
            }
            // ---- 2. basis_mat: the points' output gradients, G = g_out W on the matrix cores, g_out^T for the dW product
            vm_float4 G[2][BASIS ? RB : 1];
            if constexpr (BASIS) {
#pragma unroll
                for (uint32_t pb = 0; pb < 2; pb++) {
                    // row a of block pb <-> the trip's point kk = 8 (a >> 2) + 4 pb + (a & 3): the result rows 4 q4 + e of block pb are
                    // then the points 8 q4 + 4 pb + e — a lane walks EIGHT CONSECUTIVE sorted positions (neighbouring samples of a ray:
                    // every second one in the cell of its predecessor), and the four groups of an instruction are eight positions apart
                    const uint32_t kk = 8 * (a >> 2) + 4 * pb + (a & 3);  // = its column in the K order of the dW product
                    const uint32_t n_p = (uint32_t)__shfl((int)n, (int)kk, 64);
                    const bool ok_p = k0 + kk < seg_end;
                    vm_half8 gA;
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) gA[j] = (_Float16)0.0f;
                    if (ok_p) gA = *reinterpret_cast<const vm_half8*>(b.g_out + (size_t)n_p * kVmBasisPad + 8 * q4);
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) sc[(8 * q4 + j) * kVmMmStride + kk] = gA[j];
#pragma unroll
                    for (uint32_t rb = 0; rb < (uint32_t)RB; rb++)
                        G[pb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gA, wb[rb], vm_float4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                }
            }
            vm_wave_lds_fence();  // records (and g_out^T) written
            // ---- 3. per point: sixteen lanes = sixteen adjacent rank channels.  [Measured and dropped, profiles/r11_tensorf_vm.md: adding
            // the fixed-point contributions of consecutive points in the SAME cell (about half of them) in registers before one LDS add
            // per run — the run logic's branches serialise the points' loads (342 us), with the loads hoisted in front it spills at 128
            // registers (500 us), and the rank-16 kernel, which does neither, gains nothing (142 vs 143 us).]
            vm_half8 prod[BASIS ? RB : 1];  // [rb][4 pb + e]: the B operand of the dW product
#pragma unroll
            for (uint32_t pb = 0; pb < 2; pb++) {
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) {
                    const uint32_t pt = 8 * q4 + 4 * pb + e;
                    const uint4* r4 = reinterpret_cast<const uint4*>(rec + pt * kVmMmRec);
                    const uint4 ra = r4[0], rw = r4[1];
                    const uint2 rl = *reinterpret_cast<const uint2*>(r4 + 2);
                    const uint32_t n_p = ra.x, fl = ra.y;
                    const int c_nw = (int)ra.z, z0 = (int)ra.w;
                    if constexpr (BASIS) {
#pragma unroll
                        for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) prod[rb][4 * pb + e] = (_Float16)0.0f;
                    }
                    if (!(fl & kVmFlagLive)) continue;
                    const float nw = __uint_as_float(rw.x), ne = __uint_as_float(rw.y), sw = __uint_as_float(rw.z), se = __uint_as_float(rw.w);
                    const float lz0 = __uint_as_float(rl.x), lz1 = __uint_as_float(rl.y);
                    const bool c00 = (fl & kVmFlagX0) && (fl & kVmFlagY0), c10 = (fl & kVmFlagX1) && (fl & kVmFlagY0);
                    const bool c01 = (fl & kVmFlagX0) && (fl & kVmFlagY1), c11 = (fl & kVmFlagX1) && (fl & kVmFlagY1);
                    float l0v[RB], l1v[RB], gv[RB];
#pragma unroll
                    for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) {
                        const uint32_t r = 16 * rb + a;
                        l0v[rb] = (fl & kVmFlagZ0) ? Lt[(size_t)z0 * R + r] : 0.0f;
                        l1v[rb] = (fl & kVmFlagZ1) ? Lt[(size_t)(z0 + 1) * R + r] : 0.0f;
                        if constexpr (BASIS) gv[rb] = G[pb][rb][e];
                        else if constexpr (MODE == 1) gv[rb] = b.g[(size_t)n_p * b.rows + row0 + r];
                        else gv[rb] = b.g[n_p];
                    }
#pragma unroll
                    for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) {
                        const uint32_t r = 16 * rb + a;
                        float l = 0.0f;
                        if (fl & kVmFlagZ0) l += l0v[rb] * lz0;
                        if (fl & kVmFlagZ1) l += l1v[rb] * lz1;
                        const float g = gv[rb], gl = g * l;
                        float m = 0.0f;
                        if (c00) { m += pv[c_nw * (int)R + (int)r] * nw; vm_lds_add(&acc[c_nw * (int)R + (int)r], vm_fixed(gl * nw, scale)); }
                        if (c10) { m += pv[(c_nw + 1) * (int)R + (int)r] * ne; vm_lds_add(&acc[(c_nw + 1) * (int)R + (int)r], vm_fixed(gl * ne, scale)); }
                        if (c01) { m += pv[(c_nw + kVmTile + 1) * (int)R + (int)r] * sw; vm_lds_add(&acc[(c_nw + kVmTile + 1) * (int)R + (int)r], vm_fixed(gl * sw, scale)); }
                        if (c11) { m += pv[(c_nw + kVmTile + 2) * (int)R + (int)r] * se; vm_lds_add(&acc[(c_nw + kVmTile + 2) * (int)R + (int)r], vm_fixed(gl * se, scale)); }
                        const float gmv = g * m;
                        gm[(size_t)n_p * b.rows + row0 + r] = gmv;
                        gm_max = fmaxf(gm_max, fabsf(gmv));
                        if constexpr (BASIS) prod[rb][4 * pb + e] = (_Float16)(m * l);  // (the Linear's fp16 input, as in the forward)
                    }
                }
            }
            // ---- 4. basis_mat's gradient on the matrix cores
            if constexpr (BASIS) {
#pragma unroll
                for (uint32_t cb = 0; cb < 2; cb++) {
                    const vm_half8 gT = *reinterpret_cast<const vm_half8*>(sc + (16 * cb + a) * kVmMmStride + 8 * q4);
#pragma unroll
                    for (uint32_t rb = 0; rb < (uint32_t)RB; rb++)
                        dwacc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gT, prod[rb], dwacc[cb][rb], 0, 0, 0);
                }
            }
            vm_wave_lds_fence();  // scratch read before the next trip overwrites it
        }
        __syncthreads();
        float* dP = b.d_plane[i];
        const bool whole = pos == (uint32_t)st[t] && seg_end == (uint32_t)st[t + 1];
        const bool staged = whole && b.stage_plane != nullptr;  // border cells of a whole tile: plain stores into the tile's staging rows
        float* stg = staged ? b.stage_plane + ((size_t)i * b.stage_tiles + (size_t)t) * 32 * b.stage_R : nullptr;
        for (uint32_t e = threadIdx.x; e < kVmTileCells * R; e += kVmMmThreads) {
            const uint32_t rr = e / kVmTileCells, c = e % kVmTileCells;
            const int ly = (int)(c / (kVmTile + 1)), lx = (int)(c % (kVmTile + 1));
            const int cy = cy0 + ly, cx = cx0 + lx;
            const long long qv = acc[c * R + rr];
            const int bi = vm_border_index(lx, ly);
            if (staged && bi >= 0) {  // (every border cell is written, zeros included: the rows are not initialised)
                if (qv != 0ll) acc[c * R + rr] = 0ll;
                stg[(size_t)bi * b.stage_R + rr] = (float)qv * inv;
                continue;
            }
            if (qv != 0ll) {
                acc[c * R + rr] = 0ll;
                if (cx < W && cy < H) {
                    float* dst = &dP[rr * plane_stride + (size_t)cy * W + cx];
                    if (whole && bi < 0) *dst = (float)qv * inv;
                    else vm_flush_add(dst, (float)qv * inv);
                }
            }
        }
        if (staged && threadIdx.x == 0) b.plane_flag[(size_t)i * b.stage_tiles + (size_t)t] = 1u;
        pos = seg_end;
    }
    gm_max = wave_max(gm_max);
    if (lane == 0 && gm_max > 0.0f) atomicMax(b.bound + 2, __float_as_uint(gm_max));
    if constexpr (BASIS) {
        // basis_mat's gradient: the waves' register tiles are added in LDS (fixed order), one global atomic per (c, r) and workgroup
        float* red = reinterpret_cast<float*>(vm_smem_raw);  // [waves][2 RB tiles][64 lanes][4]
        __syncthreads();
#pragma unroll
        for (uint32_t cb = 0; cb < 2; cb++)
#pragma unroll
            for (uint32_t rb = 0; rb < (uint32_t)RB; rb++)
                *reinterpret_cast<vm_float4*>(red + ((size_t)(wave * 2 * RB + cb * RB + rb) * 64 + lane) * 4) = dwacc[cb][rb];
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < 2 * RB * 256; e += kVmMmThreads) {
            const uint32_t blk = e / 256, ln = (e % 256) / 4, ee = e % 4;
            const uint32_t cb = blk / RB, rb = blk % RB;
            float sum = 0.0f;
#pragma unroll
            for (uint32_t w = 0; w < kVmMmWaves; w++) sum += red[((size_t)(w * 2 * RB + blk) * 64 + ln) * 4 + ee];
            const uint32_t c = 16 * cb + 4 * (ln >> 4) + ee, rr = 16 * rb + (ln & 15);
            if (c < b.Cb && sum != 0.0f) atomicAdd(&b.d_basis[(size_t)c * b.rows + row0 + rr], sum);
        }
    }
}

template <int RP>
__global__ void __launch_bounds__(kVmBwdThreads) k_vm_line_backward(const float* __restrict__ x, uint32_t N, VmFactors f, VmBackward b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem_raw[];
    const uint32_t i = blockIdx.y;
    const int Dn = (int)f.Dn[i];
    const uint32_t R = f.rank[i];
    const int nchunks = (Dn + kVmZChunk - 1) / kVmZChunk;
    const int32_t* st = b.start + (size_t)(3 + i) * b.n_bounds;
    const uint32_t valid_end = (uint32_t)st[nchunks];
    const uint32_t begin = blockIdx.x * b.pts_line;
    if (begin >= valid_end) return;
    const uint32_t end = begin + b.pts_line < valid_end ? begin + b.pts_line : valid_end;
    float scale = 1.0f, inv = 1.0f;
    bool poison;
    if (!vm_scale(__uint_as_float(b.bound[2]), scale, inv, poison)) {  // (max |g m| is finite whenever the plane pass ran)
        if (poison && b.found_inf && blockIdx.x == 0 && threadIdx.x == 0) *b.found_inf = 1.0f;
        return;
    }
    long long* acc = reinterpret_cast<long long*>(vm_smem_raw);  // [65][R]
    for (uint32_t e = threadIdx.x; e < (kVmZChunk + 1) * R; e += kVmBwdThreads) acc[e] = 0ll;
    constexpr uint32_t PPW = 64 / RP;
    constexpr uint32_t NWV = kVmBwdThreads / 64;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sub = lane / RP, r = lane % RP;
    const int32_t* perm = b.perm + (size_t)(3 + i) * N;
    int t = 0;
    {
        int lo = 0, hi = nchunks;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)st[mid] <= begin) lo = mid; else hi = mid;
        }
        t = lo;
    }
    uint32_t pos = begin;
    uint32_t nseg = 0;  // segments flushed so far (the first two may be staged)
    while (pos < end) {
        while ((uint32_t)st[t + 1] <= pos) t++;
        const uint32_t seg_end = (uint32_t)st[t + 1] < end ? (uint32_t)st[t + 1] : end;
        const int zb = t * kVmZChunk;
        __syncthreads();  // accumulator clear (start of the kernel / previous flush)
        constexpr uint32_t STEP = NWV * PPW;  // (ids two trips, coordinates and g m one trip ahead: see the plane kernel)
        const uint32_t k0 = pos + wave * PPW + sub;
        uint32_t n_nx = k0 < seg_end ? (uint32_t)perm[k0] : 0u;
        uint32_t n_nx2 = k0 + STEP < seg_end ? (uint32_t)perm[k0 + STEP] : 0u;
        VmXyz p_nx = k0 < seg_end ? vm_load_xyz(x, n_nx, f, i) : VmXyz{0.0f, 0.0f, 0.0f};
        float gm_nx = (k0 < seg_end && r < R) ? b.gm[(size_t)n_nx * b.rows + f.row0[i] + r] : 0.0f;
        for (uint32_t k = k0; k < seg_end; k += STEP) {
            const VmXyz p_cur = p_nx;
            const float gm = gm_nx;
            n_nx = n_nx2;
            if (k + STEP < seg_end) {
                p_nx = vm_load_xyz(x, n_nx, f, i);
                if (r < R) gm_nx = b.gm[(size_t)n_nx * b.rows + f.row0[i] + r];
            }
            if (k + 2 * STEP < seg_end) n_nx2 = (uint32_t)perm[k + 2 * STEP];
            const VmPoint q = vm_locate_xyz(p_cur, f, i);
            if (r >= R) continue;
            const int lz = q.z0 - zb;  // -1 .. 63
            if (q.z0 >= 0 && q.z0 < Dn) vm_lds_add(&acc[lz * R + r], vm_fixed(gm * q.lz0, scale));
            if (q.z0 + 1 >= 0 && q.z0 + 1 < Dn) vm_lds_add(&acc[(lz + 1) * R + r], vm_fixed(gm * q.lz1, scale));
        }
        __syncthreads();
        float* dL = b.d_line[i];
        // the first two segments of a workgroup leave as plain stores into its staging rows (k_vm_flush_reduce adds them per chunk)
        const bool staged = b.stage_line != nullptr && nseg < 2;
        float* stg = staged ? b.stage_line + ((size_t)i * b.stage_lslots + 2 * blockIdx.x + nseg) * (kVmZChunk + 1) * b.stage_R : nullptr;
        for (uint32_t e = threadIdx.x; e < (kVmZChunk + 1) * R; e += kVmBwdThreads) {
            const uint32_t z = e / R, rr = e % R;
            const long long qv = acc[e];
            if (staged) stg[(size_t)z * b.stage_R + rr] = (float)qv * inv;
            if (qv != 0ll) {
                acc[e] = 0ll;
                if (!staged && zb + (int)z < Dn) vm_flush_add(&dL[(size_t)rr * Dn + zb + z], (float)qv * inv);
            }
        }
        if (staged && threadIdx.x == 0) b.line_flag[(size_t)i * b.stage_lslots + 2 * blockIdx.x + nseg] = (uint32_t)t + 1u;
        nseg++;
        pos = seg_end;
    }
}

// line backward, ranks a multiple of 16: the lane assignment of k_vm_plane_backward_mm (32 points per wave trip; the sixteen
// lanes of a group = sixteen adjacent rank channels of one point, four points per instruction): lanes 0..31 leave (id, cell in
// the chunk, the two weights) in the wave's scratch, g m arrives as 64 contiguous bytes per group.  Ranges moved to the
// boundaries of small chunks like the plane pass (a chunk is usually larger than a range: then nothing changes).
template <int RB>
__global__ void __launch_bounds__(kVmMmThreads) k_vm_line_backward_mm(const float* __restrict__ x, uint32_t N, VmFactors f, VmBackward b) {
    constexpr uint32_t R = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char vm_smem_raw[];
    const uint32_t i = blockIdx.y;
    const int Dn = (int)f.Dn[i];
    const int nchunks = (Dn + kVmZChunk - 1) / kVmZChunk;
    const int32_t* __restrict__ st = b.start + (size_t)(3 + i) * b.n_bounds;
    const uint32_t valid_end = (uint32_t)st[nchunks];
    float scale = 1.0f, inv = 1.0f;
    bool poison;
    if (!vm_scale(__uint_as_float(b.bound[2]), scale, inv, poison)) {
        if (poison && b.found_inf && blockIdx.x == 0 && threadIdx.x == 0) *b.found_inf = 1.0f;
        return;
    }
    uint32_t begin, end;
    int t;
    if (!vm_aligned_range(st, nchunks, blockIdx.x, b.pts_line, b.pts_line, valid_end, begin, end, t)) return;
    long long* acc = reinterpret_cast<long long*>(vm_smem_raw);                     // [65][R]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t a = lane & 15, q4 = lane >> 4, pl = lane & 31;
    uint4* rec = reinterpret_cast<uint4*>(acc + (kVmZChunk + 1) * R) + (size_t)wave * 32;  // [32] {id, cell, lz0, lz1 (0 where out of range)}
    for (uint32_t e = threadIdx.x; e < (kVmZChunk + 1) * R; e += kVmMmThreads) acc[e] = 0ll;
    const int32_t* __restrict__ perm = b.perm + (size_t)(3 + i) * N;
    const float* __restrict__ gm = b.gm;
    const uint32_t row0 = f.row0[i], cw = f.cw[i];
    uint32_t pos = begin;
    uint32_t nseg = 0;  // segments flushed so far (the first two may be staged)
    while (pos < end) {
        while ((uint32_t)st[t + 1] <= pos) t++;
        const uint32_t seg_end = (uint32_t)st[t + 1] < end ? (uint32_t)st[t + 1] : end;
        const int zb = t * kVmZChunk;
        constexpr uint32_t STEP = kVmMmWaves * 32;
        const uint32_t kf = pos + wave * 32 + pl;
        uint32_t n_nx = kf < seg_end ? (uint32_t)perm[kf] : 0u;
        uint32_t n_nx2 = kf + STEP < seg_end ? (uint32_t)perm[kf + STEP] : 0u;
        float w_nx = kf < seg_end ? x[(size_t)n_nx * 3 + cw] : 0.0f;
        __syncthreads();  // accumulator clear (start of the kernel / previous flush)
        for (uint32_t k0 = pos + wave * 32; k0 < seg_end; k0 += STEP) {
            const uint32_t k = k0 + pl;
            const uint32_t n = n_nx;
            const float wc = w_nx;
            n_nx = n_nx2;
            if (k + STEP < seg_end) w_nx = x[(size_t)n_nx * 3 + cw];
            if (k + 2 * STEP < seg_end) n_nx2 = (uint32_t)perm[k + 2 * STEP];
            {
                const float iz = unnormalize(wc, f.Dn[i]);
                const float fz = floorf(iz);
                const int z0 = (int)fz;  // (a sorted position below valid_end: finite, at least one end of the segment in range)
                const bool bz0 = z0 >= 0 && z0 < Dn, bz1 = z0 + 1 >= 0 && z0 + 1 < Dn;
                const float lz1 = bz1 ? iz - fz : 0.0f, lz0 = bz0 ? (fz + 1.0f) - iz : 0.0f;
                const uint32_t fl = (k < seg_end ? 4u : 0u) | (bz0 ? 1u : 0u) | (bz1 ? 2u : 0u);
                if (lane < 32) rec[pl] = make_uint4(k < seg_end ? n : 0u, ((uint32_t)(z0 - zb + 1) << 3) | fl, __float_as_uint(lz0), __float_as_uint(lz1));
            }
            vm_wave_lds_fence();
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint4 rc = rec[8 * q4 + j];
                if (!(rc.y & 4u)) continue;
                const int lz = (int)(rc.y >> 3) - 1;  // cell of z0 inside the chunk's 65-cell window: -1 .. 63
                const float lz0 = __uint_as_float(rc.z), lz1 = __uint_as_float(rc.w);
                float gv[RB];
#pragma unroll
                for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) gv[rb] = gm[(size_t)rc.x * b.rows + row0 + 16 * rb + a];
#pragma unroll
                for (uint32_t rb = 0; rb < (uint32_t)RB; rb++) {
                    const int r = (int)(16 * rb + a);
                    if (rc.y & 1u) vm_lds_add(&acc[lz * (int)R + r], vm_fixed(gv[rb] * lz0, scale));
                    if (rc.y & 2u) vm_lds_add(&acc[(lz + 1) * (int)R + r], vm_fixed(gv[rb] * lz1, scale));
                }
            }
            vm_wave_lds_fence();
        }
        __syncthreads();
        float* dL = b.d_line[i];
        // the first two segments of a workgroup leave as plain stores into its staging rows (k_vm_flush_reduce adds them per chunk)
        const bool staged = b.stage_line != nullptr && nseg < 2;
        float* stg = staged ? b.stage_line + ((size_t)i * b.stage_lslots + 2 * blockIdx.x + nseg) * (kVmZChunk + 1) * b.stage_R : nullptr;
        for (uint32_t e = threadIdx.x; e < (kVmZChunk + 1) * R; e += kVmMmThreads) {
            const uint32_t z = e / R, rr = e % R;
            const long long qv = acc[e];
            if (staged) stg[(size_t)z * b.stage_R + rr] = (float)qv * inv;
            if (qv != 0ll) {
                acc[e] = 0ll;
                if (!staged && zb + (int)z < Dn) vm_flush_add(&dL[(size_t)rr * Dn + zb + z], (float)qv * inv);
            }
        }
        if (staged && threadIdx.x == 0) b.line_flag[(size_t)i * b.stage_lslots + 2 * blockIdx.x + nseg] = (uint32_t)t + 1u;
        nseg++;
        pos = seg_end;
    }
}

// Adds the staged rows (VmBackward::stage_*) into the gradients, in a fixed order.  blockIdx.y = component; the first jobs walk the
// tiles: a tile's own row 0 / column 0 cells (15) take its staged border, its left neighbour's column 8, its upper
// neighbour's row 8 and the upper-left neighbour's corner — every plane cell on a tile boundary belongs to exactly one such
// job; behind them kVmLineParts jobs per line chunk: cells 0..63 of the chunk from every staged block of the chunk, plus cell 64
// of the previous chunk's blocks into cell 0.  Atomics of split tiles / third segments are complete by now (kernel boundary): plain `+=`.
constexpr uint32_t kVmLineParts = 16;  // line jobs per chunk
constexpr uint32_t kVmReduceTiles = 16;  // tiles per plane job
constexpr uint32_t kVmStageMaxSlots = 4096;  // staged line blocks per component a call may have (more: the flushes keep their atomics)
__global__ void __launch_bounds__(256) k_vm_flush_reduce(VmFactors f, VmBackward b) {
    const uint32_t i = blockIdx.y;
    const int W = (int)f.W[i], H = (int)f.H[i], Dn = (int)f.Dn[i];
    const uint32_t R = f.rank[i], SR = b.stage_R;
    const uint32_t plane_jobs = (b.stage_tiles + kVmReduceTiles - 1) / kVmReduceTiles;
    if (blockIdx.x < plane_jobs) {
        // plane job = kVmReduceTiles consecutive tiles, one wave per tile at a time (most tiles of a plane are empty: one workgroup
        // per tile spent 15 us of this launch on 4,332 workgroups reading four flags each)
        const int tiles_x = (W + kVmTile - 1) / kVmTile, tiles_y = (H + kVmTile - 1) / kVmTile;
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t* fl = b.plane_flag + (size_t)i * b.stage_tiles;
        const float* S = b.stage_plane + (size_t)i * b.stage_tiles * 32 * SR;
        float* dP = b.d_plane[i];
        const size_t plane_stride = (size_t)H * W;
        for (uint32_t k = wave; k < kVmReduceTiles; k += 4) {
            const int t = (int)(blockIdx.x * kVmReduceTiles + k);
            if (t >= tiles_x * tiles_y) break;
            const int tx = t % tiles_x, ty = t / tiles_x;
            const bool f_own = fl[t] != 0u, f_left = tx > 0 && fl[t - 1] != 0u, f_up = ty > 0 && fl[t - tiles_x] != 0u;
            const bool f_ul = tx > 0 && ty > 0 && fl[t - tiles_x - 1] != 0u;
            if (!(f_own || f_left || f_up || f_ul)) continue;  // (wave-uniform)
            // every load of a lane's cells is issued before the first one is used: unconditional, from valid addresses (a
            // neighbour that does not exist reads tile 0's rows; its value is not added)
            const size_t t_l = t > 0 ? (size_t)t - 1 : 0, t_u = t >= tiles_x ? (size_t)(t - tiles_x) : 0, t_ul = t > tiles_x ? (size_t)(t - tiles_x - 1) : 0;
            constexpr uint32_t kB = 4;  // cells of a lane in flight
            for (uint32_t e0 = lane; e0 < (2 * kVmTile - 1) * R; e0 += 64 * kB) {
                float v0[kB], v1[kB], v2[kB], v3[kB], old[kB];
                size_t dst[kB];
                bool ok[kB];
                int lxs[kB], lys[kB];
#pragma unroll
                for (uint32_t u = 0; u < kB; u++) {
                    const uint32_t e = e0 + 64 * u;
                    const bool in = e < (2 * kVmTile - 1) * R;
                    const uint32_t c15 = in ? e / R : 0u, rr = in ? e % R : 0u;
                    const int lx = c15 < (uint32_t)kVmTile ? (int)c15 : 0, ly = c15 < (uint32_t)kVmTile ? 0 : (int)c15 - (kVmTile - 1);
                    const int cx = tx * kVmTile + lx, cy = ty * kVmTile + ly;
                    ok[u] = in && cx < W && cy < H;
                    lxs[u] = lx; lys[u] = ly;
                    v0[u] = S[((size_t)t * 32 + (size_t)vm_border_index(lx, ly)) * SR + rr];
                    v1[u] = S[(t_l * 32 + (size_t)vm_border_index(kVmTile, ly)) * SR + rr];
                    v2[u] = S[(t_u * 32 + (size_t)vm_border_index(lx, kVmTile)) * SR + rr];
                    v3[u] = S[(t_ul * 32 + (size_t)vm_border_index(kVmTile, kVmTile)) * SR + rr];
                    dst[u] = ok[u] ? rr * plane_stride + (size_t)cy * W + cx : 0;
                    old[u] = dP[dst[u]];
                }
#pragma unroll
                for (uint32_t u = 0; u < kB; u++) {
                    float sum = 0.0f;
                    if (f_own) sum += v0[u];
                    if (lxs[u] == 0 && f_left) sum += v1[u];
                    if (lys[u] == 0 && f_up) sum += v2[u];
                    if (lxs[u] == 0 && lys[u] == 0 && f_ul) sum += v3[u];
                    if (ok[u] && sum != 0.0f) dP[dst[u]] = old[u] + sum;
                }
            }
        }
    } else {
        // line job = (chunk, four of its 64 cells).  The staged blocks of the chunk (and of the one before it, for cell 0) are found
        // once per workgroup — a bit per slot, then the ascending list of the set bits: a fixed summation order — and read eight at
        // a time (independent loads; one load per pass of a scan over the flags was 30 - 45 us for this launch)
        constexpr uint32_t kMaxWords = kVmStageMaxSlots / 32;  // (static LDS of EVERY workgroup of this launch: kept small)
        const uint32_t job = blockIdx.x - plane_jobs;
        const int tc = (int)(job / kVmLineParts);
        const uint32_t part = job % kVmLineParts;
        if (tc >= (Dn + kVmZChunk - 1) / kVmZChunk || !b.stage_line) return;
        __shared__ uint32_t bits[2][kMaxWords];
        __shared__ uint16_t list[2][kMaxWords * 32];
        __shared__ uint32_t count[2];
        const uint32_t nwords = (b.stage_lslots + 31) / 32;
        for (uint32_t w = threadIdx.x; w < nwords; w += 256) bits[0][w] = bits[1][w] = 0u;
        __syncthreads();
        const uint32_t* fl = b.line_flag + (size_t)i * b.stage_lslots;
        for (uint32_t slot = threadIdx.x; slot < b.stage_lslots; slot += 256) {
            const uint32_t v = fl[slot];
            if (v == (uint32_t)tc + 1u) atomicOr(&bits[0][slot >> 5], 1u << (slot & 31));
            else if (tc > 0 && part == 0 && v == (uint32_t)tc) atomicOr(&bits[1][slot >> 5], 1u << (slot & 31));
        }
        __syncthreads();
        if (threadIdx.x < 2) {  // (two lanes walk the two bitmaps: a few hundred bits)
            uint32_t n = 0;
            for (uint32_t w = 0; w < nwords; w++)
                for (uint32_t m = bits[threadIdx.x][w]; m; m &= m - 1u) list[threadIdx.x][n++] = (uint16_t)(32 * w + (uint32_t)__builtin_ctz(m));
            count[threadIdx.x] = n;
        }
        __syncthreads();
        const float* S = b.stage_line + (size_t)i * b.stage_lslots * (kVmZChunk + 1) * SR;
        float* dL = b.d_line[i];
        constexpr uint32_t ZP = kVmZChunk / kVmLineParts;
        for (uint32_t e = threadIdx.x; e < ZP * R; e += 256) {
            const uint32_t z = part * ZP + e / R, rr = e % R;
            if (tc * kVmZChunk + (int)z >= Dn) continue;
            float sum = 0.0f;
            auto add_list = [&](uint32_t which, uint32_t zz) {
                const uint32_t n = count[which];
                for (uint32_t k0 = 0; k0 < n; k0 += 8) {
                    float v[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++)
                        v[u] = k0 + u < n ? S[((size_t)list[which][k0 + u] * (kVmZChunk + 1) + zz) * SR + rr] : 0.0f;
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++) sum += v[u];
                }
            };
            add_list(0, z);
            if (z == 0) add_list(1, kVmZChunk);
            if (sum != 0.0f) dL[(size_t)rr * Dn + tc * kVmZChunk + z] += sum;
        }
    }
}

int fill_factors(VmFactors& f, const float* const* planes, const float* const* lines, const uint32_t* rank,
                 const uint32_t* resolution, uint32_t& rows) {
    static const uint32_t mat_ids[3][2] = {{0, 1}, {0, 2}, {1, 2}};  // tensoRF/network.py:37-38
    static const uint32_t vec_ids[3] = {2, 1, 0};
    rows = 0;
    for (uint32_t i = 0; i < 3; i++) {
        S3D_REQUIRE(planes[i] && lines[i] && rank[i] > 0 && resolution[i] > 0, "vm features: empty factor %u", i);
        f.plane[i] = planes[i];
        f.line[i] = lines[i];
        f.plane_t[i] = nullptr;
        f.line_t[i] = nullptr;
        f.rank[i] = rank[i];
        f.cu[i] = mat_ids[i][0];
        f.cv[i] = mat_ids[i][1];
        f.cw[i] = vec_ids[i];
        f.W[i] = resolution[mat_ids[i][0]];
        f.H[i] = resolution[mat_ids[i][1]];
        f.Dn[i] = resolution[vec_ids[i]];
        f.row0[i] = rows;
        rows += rank[i];
    }
    return S3D_OK;
}

// the optional rank-fastest shadows of a call (s3d_vm_transpose_factors): taken when every rank is a multiple of four (16-byte loads)
static void vm_set_shadows(VmFactors& f, const float* const* planes_t, const float* const* lines_t) {
    if (!planes_t || !lines_t) return;
    for (uint32_t i = 0; i < 3; i++)
        if (!planes_t[i] || !lines_t[i] || f.rank[i] % 4 != 0 || (reinterpret_cast<uintptr_t>(planes_t[i]) | reinterpret_cast<uintptr_t>(lines_t[i])) & 15u) return;
    for (uint32_t i = 0; i < 3; i++) { f.plane_t[i] = planes_t[i]; f.line_t[i] = lines_t[i]; }
}
// [rank][cells] -> [cells][rank] for the three planes (cells = H * W) and the three lines (cells = Dn) of a factor set: a
// workgroup moves 64 cells x all ranks through LDS (reads: 256-byte runs per rank, writes: 64 runs of rank x 4 bytes, contiguous)
__global__ void __launch_bounds__(256) k_vm_transpose_factors(VmFactors f, float* const p0, float* const p1, float* const p2,
                                                              float* const l0, float* const l1, float* const l2, uint32_t plane_blocks) {
    __shared__ float tile[64 * 65];
    const bool is_line = blockIdx.x >= plane_blocks;
    const uint32_t i = blockIdx.y, blk = is_line ? blockIdx.x - plane_blocks : blockIdx.x;
    const uint32_t cells = is_line ? f.Dn[i] : f.W[i] * f.H[i], R = f.rank[i];
    const uint32_t c0 = blk * 64;
    if (c0 >= cells) return;
    const float* src = is_line ? f.line[i] : f.plane[i];
    float* dst = is_line ? (i == 0 ? l0 : i == 1 ? l1 : l2) : (i == 0 ? p0 : i == 1 ? p1 : p2);
    for (uint32_t r0 = 0; r0 < R; r0 += 64) {
        const uint32_t nr = R - r0 < 64 ? R - r0 : 64;
        // (four loads in flight per lane before the first is parked: one element per pass waits for its own load each time)
        for (uint32_t e0 = threadIdx.x; e0 < nr * 64; e0 += 4 * 256) {
            float v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * 256, r = e / 64, c = e % 64;
                v[u] = (e < nr * 64 && c0 + c < cells) ? src[(size_t)(r0 + r) * cells + c0 + c] : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + u * 256;
                if (e < nr * 64) tile[(e / 64) * 65 + e % 64] = v[u];
            }
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < nr * 64; e += 256) {
            const uint32_t c = e / nr, r = e % nr;
            if (c0 + c < cells) dst[(size_t)(c0 + c) * R + r0 + r] = tile[r * 65 + c];
        }
        __syncthreads();
    }
}

// ---- the weights of a bias-free Linear chain in the ffmlp package's flat fp16 layout (and the flat fp16 gradient back into
// fp32 matrices): matrix i is [rows_i, cols_i] fp32 row-major and lands at flat[off_i + r * ld_i + c], the padding (columns
// cols_i..ld_i, rows behind rows_i up to prows_i) is written as zeros
constexpr int kPackMaxMats = 8;
struct PackJobs {
    float* m[kPackMaxMats];
    uint32_t rows[kPackMaxMats], cols[kPackMaxMats], prows[kPackMaxMats], ld[kPackMaxMats], off[kPackMaxMats + 1];
    int32_t count;
};
__global__ void __launch_bounds__(256) k_pack_linear_chain(PackJobs j, _Float16* __restrict__ flat) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= j.off[j.count]) return;
    int i = 0;
    while (i + 1 < j.count && t >= j.off[i + 1]) i++;
    const uint32_t e = t - j.off[i], r = e / j.ld[i], c = e - r * j.ld[i];
    flat[t] = (r < j.rows[i] && c < j.cols[i]) ? (_Float16)j.m[i][(size_t)r * j.cols[i] + c] : (_Float16)0.0f;
}
__global__ void __launch_bounds__(256) k_unpack_linear_chain(PackJobs j, const _Float16* __restrict__ flat) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= j.off[j.count]) return;
    int i = 0;
    while (i + 1 < j.count && t >= j.off[i + 1]) i++;
    const uint32_t e = t - j.off[i], r = e / j.ld[i], c = e - r * j.ld[i];
    if (r < j.rows[i] && c < j.cols[i]) j.m[i][(size_t)r * j.cols[i] + c] = (float)flat[t];
}

// ---- two small pieces of the TensoRF step that were chains of tiny torch launches
// x -> 2 (x - lo) / (hi - lo) - 1 per axis (tensoRF/network.py:155-157 `_normalize`, the reference's operation order)
__global__ void __launch_bounds__(256) k_aabb_normalize(const float* __restrict__ x, const float* __restrict__ aabb, uint32_t n3,
                                                        float* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n3) return;
    const uint32_t a = t % 3u;
    out[t] = (2.0f * (x[t] - aabb[a])) / (aabb[3 + a] - aabb[a]) - 1.0f;
}
// sum_i w_i * sum |t_i| over up to 8 tensors (density_loss(): w_i = 1 / numel_i): block partials, then one block adds them in a
// fixed order
constexpr int kAbsMaxTensors = 8;
constexpr uint32_t kAbsBlocks = 1024;
struct AbsJobs {
    const float* p[kAbsMaxTensors];
    uint64_t n[kAbsMaxTensors];
    float w[kAbsMaxTensors];
    int32_t count;
};
__global__ void __launch_bounds__(256) k_weighted_abs_partial(AbsJobs j, float* __restrict__ partial) {
    __shared__ float part[4];
    float acc = 0.0f;
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (uint64_t)kAbsBlocks * 256;
    for (int i = 0; i < j.count; i++) {
        // 16-byte loads, four independent chains (a single chain of 4-byte loads over 256 workgroups took 28 us for 17 MB)
        const bool vec = ((uintptr_t)j.p[i] & 15) == 0;
        const uint64_t n4 = vec ? j.n[i] / 4 : 0;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (uint64_t q = tid; q < n4; q += nthreads) {
            const float4 v = reinterpret_cast<const float4*>(j.p[i])[q];
            a0 += fabsf(v.x); a1 += fabsf(v.y); a2 += fabsf(v.z); a3 += fabsf(v.w);
        }
        for (uint64_t k = n4 * 4 + tid; k < j.n[i]; k += nthreads) a0 += fabsf(j.p[i][k]);
        acc += j.w[i] * ((a0 + a1) + (a2 + a3));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ void __launch_bounds__(256) k_weighted_abs_final(const float* __restrict__ partial, float* __restrict__ out) {
    __shared__ float part[4];
    float acc = 0.0f;
#pragma unroll
    for (uint32_t k = 0; k < kAbsBlocks / 256; k++) acc += partial[threadIdx.x + 256 * k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (part[0] + part[1]) + (part[2] + part[3]);
}

}  // namespace
}  // namespace s3d

using namespace s3d;

S3D_EXPORT uint32_t s3d_vm_backward_max_bins(const uint32_t* resolution) {
    uint32_t m = 0;
    for (uint32_t a = 0; a < 3; a++)
        for (uint32_t c = 0; c < 3; c++) {
            if (a == c) continue;
            const uint32_t t = div_up<uint32_t>(resolution[a], kVmTile) * div_up<uint32_t>(resolution[c], kVmTile);
            m = t > m ? t : m;
        }
    return m;  // (>= the number of line chunks of any axis as well: ceil(res / 64) <= ceil(res / 8)^2)
}

S3D_EXPORT int s3d_vm_backward_keys(const float* x, uint32_t N, const uint32_t* rank, const uint32_t* resolution, int32_t* keys,
                                    s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && rank && resolution && keys, "vm_backward_keys: null pointer");
    VmFactors f;
    uint32_t rows;
    const float* dummy[3] = {x, x, x};  // (only the geometry is used)
    if (int rc = fill_factors(f, dummy, dummy, rank, resolution, rows)) return rc;
    hipLaunchKernelGGL(k_vm_keys, dim3(div_up<uint32_t>(N, 256)), dim3(256), 0, as_stream(stream), x, N, f, keys);
    return check_launch("vm_backward_keys");
}

S3D_EXPORT size_t s3d_vm_backward_bins_workspace_size(uint32_t N, uint32_t n_bounds) {
    return ((size_t)6 * N + (size_t)12 * n_bounds) * sizeof(uint32_t);  // keys [6,N] | counts [6,n_bounds] | cursors [6,n_bounds]
}
// keys + counts, scan, scatter: four launches (with the clearing of the counters) instead of torch.sort's ~20 merge passes over
// 6N words and a dozen small tensor ops around it
S3D_EXPORT int s3d_vm_backward_bins(const float* x, uint32_t N, const uint32_t* rank, const uint32_t* resolution, int32_t* perm,
                                    int32_t* start, uint32_t n_bounds, void* workspace, size_t workspace_bytes, const int32_t* n_valid, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && rank && resolution && perm && start && workspace, "vm_backward_bins: null pointer");
    S3D_REQUIRE(n_bounds >= s3d_vm_backward_max_bins(resolution) + 2, "vm_backward_bins: n_bounds must exceed max_bins + 1");
    S3D_REQUIRE((uint64_t)6 * N < (1ull << 31), "vm_backward_bins: batch too large");
    S3D_REQUIRE(workspace_bytes >= s3d_vm_backward_bins_workspace_size(N, n_bounds), "vm_backward_bins: workspace too small");
    VmFactors f;
    uint32_t rows;
    const float* dummy[3] = {x, x, x};  // (only the geometry is used)
    if (int rc = fill_factors(f, dummy, dummy, rank, resolution, rows)) return rc;
    uint32_t* keys = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* counts = keys + (size_t)6 * N;
    uint32_t* cursors = counts + (size_t)6 * n_bounds;
    hipStream_t st = as_stream(stream);
    // (a kernel, not hipMemsetAsync: the call is also captured into HIP graphs, where a memset node did not clear the counters on
    //  replay — ROCm 7.2, memory access fault in the scatter of the first replayed step)
    hipLaunchKernelGGL(k_vm_zero_words, dim3(div_up<uint32_t>(12u * n_bounds, 256)), dim3(256), 0, st, counts, 12u * n_bounds);
    uint32_t nlb = 0;  // line chunks of the longest axis
    for (uint32_t i = 0; i < 3; i++) nlb = std::max(nlb, div_up<uint32_t>(resolution[i], (uint32_t)kVmZChunk));
    S3D_REQUIRE(nlb < kVmLineSlots, "vm_backward_bins: resolution %u has more than %u line chunks", nlb * kVmZChunk, kVmLineSlots - 1);
    hipLaunchKernelGGL(k_vm_bin_count, dim3(div_up<uint32_t>(N, kVmBinThreads)), dim3(kVmBinThreads), 0, st, x, N, f, n_bounds, nlb, keys, counts, n_valid);
    hipLaunchKernelGGL(k_vm_bin_scan, dim3(1), dim3(384), 0, st, (const uint32_t*)counts, n_bounds, start);
    hipLaunchKernelGGL(k_vm_bin_scatter, dim3(div_up<uint32_t>(N, kVmBinThreads)), dim3(kVmBinThreads), 0, st, (const uint32_t*)keys, N,
                       n_bounds, nlb, (const int32_t*)start, cursors, perm);
    return check_launch("vm_backward_bins");
}

static const std::array<uint32_t, 4>& vm_lane_pts() {
    static const std::array<uint32_t, 4> pts = [] {
        std::array<uint32_t, 4> v = {256u, 512u, 1024u, 1024u};  // (same-box sweep, tools/vm_pts_sweep.sh)
        if (const char* e = getenv("S3D_VM_PTS")) {
            unsigned a, b2, c, d;
            if (sscanf(e, "%u,%u,%u,%u", &a, &b2, &c, &d) == 4 && a && b2 && c && d) v = {a, b2, c, d};
        }
        return v;
    }();
    return pts;
}
static uint32_t vm_mm_pts(uint32_t which);
// staging rows of the flushes (VmBackward::stage_*): flags | plane rows | line rows
struct VmStage { uint32_t tiles, lslots, R; size_t flag_words, plane_floats, line_floats, bytes; };
static VmStage vm_stage_layout(uint32_t N, const uint32_t* rank, const uint32_t* resolution) {
    VmStage v{};
    for (uint32_t i = 0; i < 3; i++) {
        v.R = rank[i] > v.R ? rank[i] : v.R;
        for (uint32_t j = i + 1; j < 3; j++) {
            const uint32_t t = div_up<uint32_t>(resolution[i], kVmTile) * div_up<uint32_t>(resolution[j], kVmTile);
            v.tiles = t > v.tiles ? t : v.tiles;
        }
    }
    const std::array<uint32_t, 4>& lp = vm_lane_pts();
    const uint32_t min_line = std::min(std::min(std::min(lp[2], lp[3]), kVmMaxPts), vm_mm_pts(2));  // whichever line kernel runs
    v.lslots = 2 * div_up<uint32_t>(N, min_line);
    v.flag_words = ((size_t)3 * v.tiles + (size_t)3 * v.lslots + 63) & ~(size_t)63;
    v.plane_floats = (size_t)3 * v.tiles * 32 * v.R;
    v.line_floats = (size_t)3 * v.lslots * (kVmZChunk + 1) * v.R;
    v.bytes = (v.flag_words + v.plane_floats + v.line_floats) * 4;
    return v;
}
// arms the staged flushes when the caller's buffer holds them; returns the flag words for k_vm_bound to clear
static uint32_t vm_stage_arm(VmBackward& b, uint32_t N, const uint32_t* rank, const uint32_t* resolution, void* stage, size_t stage_bytes,
                             uint32_t*& flags) {
    b.stage_plane = b.stage_line = nullptr; b.plane_flag = b.line_flag = nullptr;
    b.stage_tiles = b.stage_lslots = b.stage_R = 0;
    flags = nullptr;
    static const int mode = [] { const char* e = getenv("S3D_VM_STAGE"); return e ? atoi(e) : 1; }();  // (A/B: 0 off, 1 planes + lines, 2 lines only)
    const bool on = mode != 0;
    const VmStage v = vm_stage_layout(N, rank, resolution);
    if (!on || !stage || stage_bytes < v.bytes || (reinterpret_cast<uintptr_t>(stage) & 15u) || v.lslots > kVmStageMaxSlots) return 0;
    flags = reinterpret_cast<uint32_t*>(stage);
    b.plane_flag = flags;
    b.line_flag = flags + (size_t)3 * v.tiles;
    b.stage_plane = reinterpret_cast<float*>(flags + v.flag_words);
    b.stage_line = b.stage_plane + v.plane_floats;
    b.stage_tiles = v.tiles; b.stage_lslots = v.lslots; b.stage_R = v.R;
    if (mode == 2) { b.stage_plane = nullptr; b.stage_tiles = 0; }
    return (uint32_t)v.flag_words;
}
S3D_EXPORT size_t s3d_vm_backward_stage_bytes(uint32_t N, const uint32_t* rank, const uint32_t* resolution) {
    if (!rank || !resolution || !N) return 0;
    const VmStage v = vm_stage_layout(N, rank, resolution);
    return v.lslots > kVmStageMaxSlots ? 0 : v.bytes;  // (batches too long for the staged flush keep their atomics: nothing to allocate)
}

// launch geometry shared by the two backward entry points: points per workgroup so that the sorted order fills the chip a
// few times over (eight waves x 64 / RP points in flight per workgroup), LDS = fixed-point accumulator + plane values
static void vm_backward_geometry(VmBackward& b, uint32_t N, uint32_t max_rank, bool basis, dim3& gp, dim3& gl, size_t& smem_p,
                                 size_t& smem_l) {
    const uint32_t rp = max_rank <= 16 ? 16u : 64u;
    // Every (range, tile) segment ends with one global atomic per cell of the 9 x 9 window and rank channel (3,888 at rank 48)
    // and global atomics retire at ~21 G/s chip-wide: segments = ranges + occupied tiles, so the ranges are as long as the
    // workgroup count allows (tools/bench_tensorf_step.py with S3D_VM_PTS=plane64,plane16,line64,line16 sweeps them)
    const std::array<uint32_t, 4>& pts = vm_lane_pts();
    b.pts_plane = std::min(rp == 16 ? pts[1] : pts[0], kVmMaxPts);  // (the accumulators' headroom: kVmMaxPts contributions per cell)
    b.pts_line = std::min(rp == 16 ? pts[3] : pts[2], kVmMaxPts);
    gp = dim3(div_up<uint32_t>(N, b.pts_plane), 3);
    gl = dim3(div_up<uint32_t>(N, b.pts_line), 3);
    smem_p = (size_t)kVmTileCells * max_rank * (sizeof(long long) + sizeof(float));
    if (basis && smem_p < (size_t)(kVmBwdThreads / 64) * (kVmBasisPad / 2) * 64 * sizeof(float))
        smem_p = (size_t)(kVmBwdThreads / 64) * (kVmBasisPad / 2) * 64 * sizeof(float);
    smem_l = (size_t)(kVmZChunk + 1) * max_rank * sizeof(long long);
}

// the 32-points-per-trip passes (k_vm_plane_backward_mm / k_vm_line_backward_mm) serve calls whose three components share one
// rank of 32, 48 or 64 (VM-48's colour factors); anything else — rank 16 included: 127 / 30 us on the lane-per-rank kernels
// against 139 / 34 us here, profiles/r11_tensorf_vm.md — takes the lane-per-rank kernels.  S3D_VM_MM=0 switches them off (A/B),
// S3D_VM_MM_PTS=plane,-,line,- sets the nominal sorted positions per workgroup (<= kVmMaxPts).
static uint32_t vm_mm_pts(uint32_t which) {
    static const std::array<uint32_t, 4> pts = [] {
        std::array<uint32_t, 4> v = {512u, 512u, 1024u, 1024u};  // (line: 2,048 without the staged flush, 1,024 with it)
        if (const char* e = getenv("S3D_VM_MM_PTS")) {
            unsigned a, b2, c, d;
            if (sscanf(e, "%u,%u,%u,%u", &a, &b2, &c, &d) == 4 && a >= 32 && b2 >= 32 && c >= 32 && d >= 32)
                v = {std::min(a, kVmMaxPts), std::min(b2, kVmMaxPts), std::min(c, kVmMaxPts), std::min(d, kVmMaxPts)};
        }
        return v;
    }();
    return pts[which];
}
template <int RB, int MODE>
static void launch_plane_mm_t(const float* x, uint32_t N, const VmFactors& f, VmBackward& b, hipStream_t st) {
    constexpr size_t smem = vm_mm_smem(RB, MODE == 2);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vm_plane_backward_mm<RB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        return true;
    }();
    (void)attr;
    b.pts_plane = vm_mm_pts(0);
    hipLaunchKernelGGL((k_vm_plane_backward_mm<RB, MODE>), dim3(div_up<uint32_t>(N, b.pts_plane), 3), dim3(kVmMmThreads), smem, st, x, N, f, b);
}
template <int RB>
static void launch_line_mm_t(const float* x, uint32_t N, const VmFactors& f, VmBackward& b, hipStream_t st) {
    b.pts_line = vm_mm_pts(2);
    const size_t smem = (size_t)(kVmZChunk + 1) * 16 * RB * sizeof(long long) + (size_t)kVmMmWaves * 32 * sizeof(uint4);
    hipLaunchKernelGGL((k_vm_line_backward_mm<RB>), dim3(div_up<uint32_t>(N, b.pts_line), 3), dim3(kVmMmThreads), smem, st, x, N, f, b);
}
static bool launch_line_mm(const float* x, uint32_t N, const VmFactors& f, VmBackward& b, hipStream_t st) {
    static const bool on = [] { const char* e = getenv("S3D_VM_MM"); return !(e && e[0] == '0'); }();
    if (!on || f.rank[0] != f.rank[1] || f.rank[0] != f.rank[2] || f.rank[0] % 16 != 0 || f.rank[0] < 32 || f.rank[0] > 64) return false;
    if (reinterpret_cast<uintptr_t>(b.gm) & 15u) return false;
    switch (f.rank[0] / 16) {
        case 2: launch_line_mm_t<2>(x, N, f, b, st); break;
        case 3: launch_line_mm_t<3>(x, N, f, b, st); break;
        default: launch_line_mm_t<4>(x, N, f, b, st); break;
    }
    return true;
}
template <int MODE>
static bool launch_plane_mm(const float* x, uint32_t N, const VmFactors& f, VmBackward& b, hipStream_t st) {
    static const bool on = [] { const char* e = getenv("S3D_VM_MM"); return !(e && e[0] == '0'); }();
    if (!on || f.rank[0] != f.rank[1] || f.rank[0] != f.rank[2] || f.rank[0] % 16 != 0 || f.rank[0] < 32 || f.rank[0] > 64) return false;
    if ((reinterpret_cast<uintptr_t>(b.gm) | reinterpret_cast<uintptr_t>(b.line_t) | (MODE == 1 ? reinterpret_cast<uintptr_t>(b.g) : 0) |
         (MODE == 2 ? reinterpret_cast<uintptr_t>(b.g_out) : 0)) & 15u) return false;
    switch (f.rank[0] / 16) {
        case 2: launch_plane_mm_t<2, MODE>(x, N, f, b, st); break;
        case 3: launch_plane_mm_t<3, MODE>(x, N, f, b, st); break;
        default: launch_plane_mm_t<4, MODE>(x, N, f, b, st); break;
    }
    return true;
}

S3D_EXPORT int s3d_vm_features_backward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                                        const uint32_t* rank, const uint32_t* resolution, int reduce, const float* grad,
                                        const int32_t* perm, const int32_t* start, uint32_t n_bounds, float* gm,
                                        float* const* grad_planes, float* const* grad_lines, uint32_t* bound_words,
                                        float* line_scratch, void* stage, size_t stage_bytes, float* found_inf, const float* const* planes_t,
                                        const int32_t* n_valid, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && planes && lines && rank && resolution && grad && perm && start && gm && grad_planes && grad_lines && bound_words &&
                line_scratch, "vm_features_backward: null pointer");
    VmFactors f;
    VmBackward b;
    if (int rc = fill_factors(f, planes, lines, rank, resolution, b.rows)) return rc;
    if (planes_t && planes_t[0] && planes_t[1] && planes_t[2] &&
        !((reinterpret_cast<uintptr_t>(planes_t[0]) | reinterpret_cast<uintptr_t>(planes_t[1]) | reinterpret_cast<uintptr_t>(planes_t[2])) & 15u))
        for (uint32_t i = 0; i < 3; i++) f.plane_t[i] = planes_t[i];  // (the tile loads of the plane passes; any rank)
    uint32_t max_rank = 0, max_tiles = 0, max_chunks = 0;
    for (uint32_t i = 0; i < 3; i++) {
        S3D_REQUIRE(grad_planes[i] && grad_lines[i], "vm_features_backward: null gradient buffer %u", i);
        b.d_plane[i] = grad_planes[i];
        b.d_line[i] = grad_lines[i];
        max_rank = rank[i] > max_rank ? rank[i] : max_rank;
        const uint32_t tiles = div_up<uint32_t>(f.W[i], kVmTile) * div_up<uint32_t>(f.H[i], kVmTile);
        max_tiles = tiles > max_tiles ? tiles : max_tiles;
        const uint32_t chunks = div_up<uint32_t>(f.Dn[i], kVmZChunk);
        max_chunks = chunks > max_chunks ? chunks : max_chunks;
    }
    S3D_REQUIRE(max_rank <= 64, "vm_features_backward: rank %u > 64 not supported", max_rank);
    S3D_REQUIRE(n_bounds > max_tiles && n_bounds > max_chunks, "vm_features_backward: `start` needs more than %u columns", max_tiles);
    b.g = grad;
    b.gm = gm;
    b.perm = perm;
    b.start = start;
    b.n_bounds = n_bounds;
    b.basis = nullptr; b.g_out = nullptr; b.d_basis = nullptr; b.Cb = 0;
    b.bound = bound_words;
    b.found_inf = found_inf;
    b.line_t = line_scratch;
    for (uint32_t i = 0, off = 0; i < 3; off += f.rank[i] * f.Dn[i], i++) b.line_t_off[i] = off;
    hipStream_t st = as_stream(stream);
    dim3 gp, gl;
    size_t smem_p, smem_l;
    vm_backward_geometry(b, N, max_rank, false, gp, gl, smem_p, smem_l);
    const dim3 block(kVmBwdThreads);
    const size_t n_g = reduce ? (size_t)N : (size_t)N * b.rows;
    uint32_t* stage_flags;
    const uint32_t n_stage_flags = vm_stage_arm(b, N, rank, resolution, stage, stage_bytes, stage_flags);
    hipLaunchKernelGGL(k_vm_bound, dim3(std::min<uint32_t>(stream_grid(n_g / 4 + 1, 256), 512u)), dim3(256), 0, st, grad, n_g, (const _Float16*)nullptr, (size_t)0, f,
                       (const _Float16*)nullptr, 0u, b.rows, bound_words, line_scratch, stage_flags, n_stage_flags, N, n_valid);
    const bool mm = reduce ? launch_plane_mm<0>(x, N, f, b, st) : launch_plane_mm<1>(x, N, f, b, st);
    if (max_rank <= 16) {
        if (mm) {}
        else if (reduce) hipLaunchKernelGGL((k_vm_plane_backward<16, true>), gp, block, smem_p, st, x, N, f, b);
        else hipLaunchKernelGGL((k_vm_plane_backward<16, false>), gp, block, smem_p, st, x, N, f, b);
        if (!launch_line_mm(x, N, f, b, st)) hipLaunchKernelGGL((k_vm_line_backward<16>), gl, block, smem_l, st, x, N, f, b);
    } else {
        if (mm) {}
        else if (reduce) hipLaunchKernelGGL((k_vm_plane_backward<64, true>), gp, block, smem_p, st, x, N, f, b);
        else hipLaunchKernelGGL((k_vm_plane_backward<64, false>), gp, block, smem_p, st, x, N, f, b);
        if (!launch_line_mm(x, N, f, b, st)) hipLaunchKernelGGL((k_vm_line_backward<64>), gl, block, smem_l, st, x, N, f, b);
    }
    if (b.stage_line) hipLaunchKernelGGL(k_vm_flush_reduce, dim3(div_up<uint32_t>(b.stage_tiles, kVmReduceTiles) + kVmLineParts * max_chunks, 3), dim3(256), 0, st, f, b);
    return check_launch("vm_features_backward");
}

S3D_EXPORT int s3d_vm_color_forward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                                    const uint32_t* rank, const uint32_t* resolution, const uint16_t* basis, uint32_t basis_rows,
                                    uint16_t* out, const float* const* planes_t, const float* const* lines_t, const int32_t* n_valid,
                                    s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && planes && lines && rank && resolution && basis && out, "vm_color_forward: null pointer");
    VmFactors f;
    uint32_t rows;
    if (int rc = fill_factors(f, planes, lines, rank, resolution, rows)) return rc;
    S3D_REQUIRE(basis_rows >= 1 && basis_rows <= kVmBasisPad, "vm_color_forward: basis_mat with %u outputs (1 .. %u supported)",
                basis_rows, kVmBasisPad);
    S3D_REQUIRE(rows * kVmBasisPad * sizeof(float) <= 64 * 1024, "vm_color_forward: %u product rows do not fit the LDS table", rows);
    vm_set_shadows(f, planes_t, lines_t);
    hipLaunchKernelGGL(k_vm_color_basis, dim3(div_up<uint32_t>(N, 256)), dim3(256), rows * kVmBasisPad * sizeof(float), as_stream(stream),
                       x, N, f, (const _Float16*)basis, basis_rows, rows, (_Float16*)out, n_valid);
    return check_launch("vm_color_forward");
}

S3D_EXPORT int s3d_vm_color_backward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                                     const uint32_t* rank, const uint32_t* resolution, const uint16_t* basis, uint32_t basis_rows,
                                     const uint16_t* grad_out, const int32_t* perm, const int32_t* start, uint32_t n_bounds,
                                     float* gm, float* const* grad_planes, float* const* grad_lines, float* grad_basis,
                                     uint32_t* bound_words, float* line_scratch, void* stage, size_t stage_bytes, float* found_inf,
                                     const float* const* planes_t, const int32_t* n_valid, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && planes && lines && rank && resolution && basis && grad_out && perm && start && gm && grad_planes && grad_lines &&
                grad_basis && bound_words && line_scratch, "vm_color_backward: null pointer");
    VmFactors f;
    VmBackward b;
    if (int rc = fill_factors(f, planes, lines, rank, resolution, b.rows)) return rc;
    S3D_REQUIRE(basis_rows >= 1 && basis_rows <= kVmBasisPad, "vm_color_backward: basis_mat with %u outputs (1 .. %u supported)",
                basis_rows, kVmBasisPad);
    if (planes_t && planes_t[0] && planes_t[1] && planes_t[2] &&
        !((reinterpret_cast<uintptr_t>(planes_t[0]) | reinterpret_cast<uintptr_t>(planes_t[1]) | reinterpret_cast<uintptr_t>(planes_t[2])) & 15u))
        for (uint32_t i = 0; i < 3; i++) f.plane_t[i] = planes_t[i];  // (the tile loads of the plane passes; any rank)
    uint32_t max_rank = 0, max_tiles = 0, max_chunks = 0;
    for (uint32_t i = 0; i < 3; i++) {
        S3D_REQUIRE(grad_planes[i] && grad_lines[i], "vm_color_backward: null gradient buffer %u", i);
        b.d_plane[i] = grad_planes[i];
        b.d_line[i] = grad_lines[i];
        max_rank = rank[i] > max_rank ? rank[i] : max_rank;
        const uint32_t tiles = div_up<uint32_t>(f.W[i], kVmTile) * div_up<uint32_t>(f.H[i], kVmTile);
        max_tiles = tiles > max_tiles ? tiles : max_tiles;
        const uint32_t chunks = div_up<uint32_t>(f.Dn[i], kVmZChunk);
        max_chunks = chunks > max_chunks ? chunks : max_chunks;
    }
    S3D_REQUIRE(max_rank <= 64, "vm_color_backward: rank %u > 64 not supported", max_rank);
    S3D_REQUIRE(n_bounds > max_tiles && n_bounds > max_chunks, "vm_color_backward: `start` needs more than %u columns", max_tiles);
    b.g = nullptr;
    b.gm = gm;
    b.perm = perm;
    b.start = start;
    b.n_bounds = n_bounds;
    b.basis = (const _Float16*)basis;
    b.g_out = (const _Float16*)grad_out;
    b.d_basis = grad_basis;
    b.Cb = basis_rows;
    b.bound = bound_words;
    b.found_inf = found_inf;
    b.line_t = line_scratch;
    for (uint32_t i = 0, off = 0; i < 3; off += f.rank[i] * f.Dn[i], i++) b.line_t_off[i] = off;
    hipStream_t st = as_stream(stream);
    dim3 gp, gl;
    size_t smem_p, smem_l;
    vm_backward_geometry(b, N, 64, true, gp, gl, smem_p, smem_l);  // (the BASIS kernel: 64 lanes per point whatever the rank)
    const dim3 block(kVmBwdThreads);
    const size_t n_g16 = (size_t)N * kVmBasisPad;
    uint32_t* stage_flags;
    const uint32_t n_stage_flags = vm_stage_arm(b, N, rank, resolution, stage, stage_bytes, stage_flags);
    hipLaunchKernelGGL(k_vm_bound, dim3(std::min<uint32_t>(stream_grid(n_g16 / 8 + 1, 256), 512u)), dim3(256), 0, st, (const float*)nullptr, (size_t)0,
                       (const _Float16*)grad_out, n_g16, f, (const _Float16*)basis, basis_rows, b.rows, bound_words, line_scratch, stage_flags, n_stage_flags, N, n_valid);
    if (!launch_plane_mm<2>(x, N, f, b, st)) hipLaunchKernelGGL((k_vm_plane_backward<64, false, true>), gp, block, smem_p, st, x, N, f, b);
    if (!launch_line_mm(x, N, f, b, st)) hipLaunchKernelGGL((k_vm_line_backward<64>), gl, block, smem_l, st, x, N, f, b);
    if (b.stage_line) hipLaunchKernelGGL(k_vm_flush_reduce, dim3(div_up<uint32_t>(b.stage_tiles, kVmReduceTiles) + kVmLineParts * max_chunks, 3), dim3(256), 0, st, f, b);
    return check_launch("vm_color_backward");
}

S3D_EXPORT int s3d_vm_transpose_factors(const float* const* planes, const float* const* lines, const uint32_t* rank,
                                        const uint32_t* resolution, float* const* planes_t, float* const* lines_t, s3d_stream_t stream) {
    S3D_REQUIRE(planes && lines && rank && resolution && planes_t && lines_t, "vm_transpose_factors: null pointer");
    VmFactors f;
    uint32_t rows;
    if (int rc = fill_factors(f, planes, lines, rank, resolution, rows)) return rc;
    uint32_t pb = 0, lb = 0;
    for (uint32_t i = 0; i < 3; i++) {
        S3D_REQUIRE(planes_t[i] && lines_t[i], "vm_transpose_factors: null output %u", i);
        pb = std::max(pb, div_up<uint32_t>(f.W[i] * f.H[i], 64));
        lb = std::max(lb, div_up<uint32_t>(f.Dn[i], 64));
    }
    hipLaunchKernelGGL(k_vm_transpose_factors, dim3(pb + lb, 3), dim3(256), 0, as_stream(stream), f, planes_t[0], planes_t[1], planes_t[2],
                       lines_t[0], lines_t[1], lines_t[2], pb);
    return check_launch("vm_transpose_factors");
}

S3D_EXPORT int s3d_vm_features_forward(const float* x, uint32_t N, const float* const* planes, const float* const* lines,
                                       const uint32_t* rank, const uint32_t* resolution, int reduce, float* out,
                                       const float* const* planes_t, const float* const* lines_t, const int32_t* n_valid,
                                       s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && planes && lines && rank && resolution && out, "vm_features_forward: null pointer");
    VmFactors f;
    uint32_t row;
    if (int rc = fill_factors(f, planes, lines, rank, resolution, row)) return rc;
    vm_set_shadows(f, planes_t, lines_t);
    const dim3 grid(div_up<uint32_t>(N, 256)), block(256);
    if (reduce) hipLaunchKernelGGL((k_vm_features<true>), grid, block, 0, as_stream(stream), x, N, f, out, n_valid);
    else hipLaunchKernelGGL((k_vm_features<false>), grid, block, 0, as_stream(stream), x, N, f, out, n_valid);
    return check_launch("vm_features_forward");
}

S3D_EXPORT int s3d_aabb_normalize(const float* x, const float* aabb, uint32_t N, float* out, s3d_stream_t stream) {
    if (N == 0) return S3D_OK;
    S3D_REQUIRE(x && aabb && out && (uint64_t)N * 3 < (1ull << 32), "aabb_normalize: null pointer or too many points");
    hipLaunchKernelGGL(k_aabb_normalize, dim3(div_up<uint32_t>(N * 3, 256)), dim3(256), 0, as_stream(stream), x, aabb, N * 3, out);
    return check_launch("aabb_normalize");
}

S3D_EXPORT size_t s3d_weighted_abs_sum_workspace_size(void) { return kAbsBlocks * sizeof(float); }
S3D_EXPORT int s3d_weighted_abs_sum(const float* const* tensors, const uint64_t* numel, const float* weights, int32_t count, float* out,
                                    float* workspace, s3d_stream_t stream) {
    S3D_REQUIRE(tensors && numel && weights && out && workspace && count >= 1 && count <= kAbsMaxTensors,
                "weighted_abs_sum: null pointer or more than %d tensors", kAbsMaxTensors);
    AbsJobs j;
    memset(&j, 0, sizeof(j));
    j.count = count;
    for (int i = 0; i < count; i++) { j.p[i] = tensors[i]; j.n[i] = numel[i]; j.w[i] = weights[i]; }
    hipLaunchKernelGGL(k_weighted_abs_partial, dim3(kAbsBlocks), dim3(256), 0, as_stream(stream), j, workspace);
    hipLaunchKernelGGL(k_weighted_abs_final, dim3(1), dim3(256), 0, as_stream(stream), (const float*)workspace, out);
    return check_launch("weighted_abs_sum");
}

static int fill_pack_jobs(PackJobs& j, float* const* mats, const uint32_t* rows, const uint32_t* cols, const uint32_t* padded_rows,
                          const uint32_t* ld, int32_t count) {
    S3D_REQUIRE(mats && rows && cols && padded_rows && ld && count >= 1 && count <= kPackMaxMats, "linear_chain pack: null pointer or more than %d matrices",
                kPackMaxMats);
    memset(&j, 0, sizeof(j));
    j.count = count;
    uint32_t off = 0;
    for (int i = 0; i < count; i++) {
        S3D_REQUIRE(mats[i] && ld[i] >= cols[i] && padded_rows[i] >= rows[i], "linear_chain pack: matrix %d: ld >= cols, padded_rows >= rows", i);
        j.m[i] = mats[i]; j.rows[i] = rows[i]; j.cols[i] = cols[i]; j.prows[i] = padded_rows[i]; j.ld[i] = ld[i];
        j.off[i] = off;
        off += padded_rows[i] * ld[i];
    }
    j.off[count] = off;
    return S3D_OK;
}
S3D_EXPORT int s3d_pack_linear_chain(const float* const* mats, const uint32_t* rows, const uint32_t* cols, const uint32_t* padded_rows,
                                     const uint32_t* ld, int32_t count, uint16_t* flat, s3d_stream_t stream) {
    PackJobs j;
    if (int rc = fill_pack_jobs(j, const_cast<float* const*>(mats), rows, cols, padded_rows, ld, count)) return rc;
    S3D_REQUIRE(flat, "pack_linear_chain: null pointer");
    hipLaunchKernelGGL(k_pack_linear_chain, dim3(div_up<uint32_t>(j.off[count], 256)), dim3(256), 0, as_stream(stream), j, (_Float16*)flat);
    return check_launch("pack_linear_chain");
}
S3D_EXPORT int s3d_unpack_linear_chain(const uint16_t* flat, float* const* mats, const uint32_t* rows, const uint32_t* cols,
                                       const uint32_t* padded_rows, const uint32_t* ld, int32_t count, s3d_stream_t stream) {
    PackJobs j;
    if (int rc = fill_pack_jobs(j, mats, rows, cols, padded_rows, ld, count)) return rc;
    S3D_REQUIRE(flat, "unpack_linear_chain: null pointer");
    hipLaunchKernelGGL(k_unpack_linear_chain, dim3(div_up<uint32_t>(j.off[count], 256)), dim3(256), 0, as_stream(stream), j, (const _Float16*)flat);
    return check_launch("unpack_linear_chain");
}
