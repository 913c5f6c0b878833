// Fused projection + voxelisation for gfx950 (HBM-bound, integer/index work bit-exact).
//
// One pass per camera pixel replaces the ~20 ATen launches of
//   SOccDPT.get_semantic_occupancy   /root/reference/SOccDPT/model/SOccDPT.py:264-372
//   rotate_points                    /root/reference/SOccDPT/model/SOccDPT.py:60-130
//   points_to_occupancy_grid         /root/reference/SOccDPT/model/SOccDPT.py:374-463
// bicubic-sample inverse depth, nearest-sample class probabilities, clamp, reciprocal,
// back-project, write inv_up / seg_up / points, rotate in registers, voxel index, bounds
// test, OR into a bit-packed union grid.  A second kernel expands bits -> f32 rows.
//
// Float contract (matches PyTorch's CPU kernels bit-for-bit, see oracle/projection_ref.c):
// this TU is compiled with -ffp-contract=off; every fused multiply-add that the CPU
// build performs is written with fmaf, every other op rounds to f32 separately, all
// divisions are IEEE.
//
// Algorithmic HBM bytes per frame: read (1+C)*h*w*4, write Hc*Wc*4*(1+C+3)
// + one occupancy row gx*gy*gz*C*4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "internal.h"
#include "resample.h"

namespace soccdpt {


__device__ __forceinline__ void rot3(const float p[3], const float* R, float o[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = fmaf(p[2], R[6 + j], fmaf(p[1], R[3 + j], p[0] * R[j]));
}

struct ProjParams {
    const float* inv;   // [B,h,w]
    const float* seg;   // [B,C,h,w]
    float* inv_up;      // [B,Hc,Wc]
    float* seg_up;      // [B,C,Hc,Wc]
    float* points;      // [B,Hc,Wc,3]
    uint32_t* occ_bits; // packed union grid or nullptr
    int B, h, w, Hc, Wc;
    float fx, fy, cx, cy;
    float pc_scale[3], pc_shift[3];
    float rot[27];
    float occ_shape[3];
    int grid[3];
};

// One thread = VEC consecutive pixels of one camera row (VEC = 4: 16-byte stores).
template <int C, int VEC>
__global__ __launch_bounds__(256) void project_kernel(ProjParams P) {
    const int quads_per_row = P.Wc / VEC;
    const long long total = (long long)P.B * P.Hc * quads_per_row;
    const float sy = (float)P.h / (float)P.Hc;
    const float sx = (float)P.w / (float)P.Wc;
    const size_t npix = (size_t)P.Hc * P.Wc;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int qx = (int)(q % quads_per_row);
        const long long ru = q / quads_per_row;
        const int u = (int)(ru % P.Hc);
        const int b = (int)(ru / P.Hc);
        const int v0 = qx * VEC;
        const float* src = P.inv + (size_t)b * P.h * P.w;
        const Taps ty = cubic_taps(u, P.h, sy);
        const int su = nearest_src(u, P.h, sy);
        const float* r0 = src + (size_t)ty.idx[0] * P.w;
        const float* r1 = src + (size_t)ty.idx[1] * P.w;
        const float* r2 = src + (size_t)ty.idx[2] * P.w;
        const float* r3 = src + (size_t)ty.idx[3] * P.w;
        const float yterm = (float)u - P.cy;

        float iv[VEC], sem[C][VEC], pt[VEC][3];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int v = v0 + e;
            const Taps tx = cubic_taps(v, P.w, sx);
            float t0 = dot4(r0[tx.idx[0]], r0[tx.idx[1]], r0[tx.idx[2]], r0[tx.idx[3]], tx.w);
            float t1 = dot4(r1[tx.idx[0]], r1[tx.idx[1]], r1[tx.idx[2]], r1[tx.idx[3]], tx.w);
            float t2 = dot4(r2[tx.idx[0]], r2[tx.idx[1]], r2[tx.idx[2]], r2[tx.idx[3]], tx.w);
            float t3 = dot4(r3[tx.idx[0]], r3[tx.idx[1]], r3[tx.idx[2]], r3[tx.idx[3]], tx.w);
            float val = dot4(t0, t1, t2, t3, ty.w);
            if (val < 1e-8f) val = 1e-8f;  // NaN compares false and stays NaN
            float d = 1.0f / val;
            if (isinf(d) || isnan(d)) d = __builtin_inff();
            iv[e] = val;
            const int sv = nearest_src(v, P.w, sx);
#pragma unroll
            for (int c = 0; c < C; ++c) sem[c][e] = P.seg[(((size_t)b * C + c) * P.h + su) * P.w + sv];
            float p[3];
            p[0] = (((float)v - P.cx) * d) / P.fx;
            p[1] = (yterm * d) / P.fy;
            p[2] = d;
            const size_t n = (size_t)u * P.Wc + v;
            if (n < 3) {  // the reference scales/shifts flat pixels 0,1,2 of each image
#pragma unroll
                for (int k = 0; k < 3; ++k) p[k] = p[k] * P.pc_scale[n] + P.pc_shift[n];
            }
            pt[e][0] = p[0];
            pt[e][1] = p[1];
            pt[e][2] = p[2];

            if (P.occ_bits) {
                float a[3], bq[3], cq[3];
                rot3(p, P.rot, a);
                rot3(a, P.rot + 9, bq);
                rot3(bq, P.rot + 18, cq);
                const bool fin = isfinite(cq[0]) && isfinite(cq[1]) && isfinite(cq[2]);
                const float fi = (cq[0] / P.occ_shape[0]) * (float)P.grid[0];
                const float fj = (cq[1] / P.occ_shape[1]) * (float)P.grid[1];
                const float fk = (cq[2] / P.occ_shape[2]) * (float)P.grid[2];
                // trunc-toward-zero as the reference's .type(int64); window test in float first so the
                // integer conversion is always in range
                const bool inr = fin && fi > -1.0f && fi < 65536.0f && fj > -1.0f && fj < 65536.0f && fk > -1.0f && fk < 65536.0f;
                if (inr) {
                    const int i = (int)fi, j = (int)fj, k = (int)fk;
                    if (0 < i && i < P.grid[0] && 0 < j && j < P.grid[1] && 0 < k && k < P.grid[2]) {
                        const uint32_t base = (uint32_t)(((i * P.grid[1] + j) * P.grid[2] + k) * C);
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            if (sem[c][e] != 0.0f) {
                                const uint32_t bit = base + c;
                                const uint32_t m = 1u << (bit & 31);
                                uint32_t* wp = P.occ_bits + (bit >> 5);
                                // idempotent OR: a stale read only costs a redundant atomic, so the check may be served by the nearest cache
                // (workgroup scope; an agent-scope load bypasses L2 on gfx950 and made scattered scenes 11 % slower)
                                if (!(__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & m)) atomicOr(wp, m);
                            }
                        }
                    }
                }
            }
        }
        // ---- stores (16 B per lane where VEC == 4) ----
        const size_t n0 = (size_t)u * P.Wc + v0;
        if constexpr (VEC == 4) {
            if (P.inv_up) *reinterpret_cast<float4*>(P.inv_up + (size_t)b * npix + n0) = make_float4(iv[0], iv[1], iv[2], iv[3]);
            if (P.seg_up) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    *reinterpret_cast<float4*>(P.seg_up + ((size_t)b * C + c) * npix + n0) =
                        make_float4(sem[c][0], sem[c][1], sem[c][2], sem[c][3]);
            }
            if (P.points) {
                float4* o = reinterpret_cast<float4*>(P.points + ((size_t)b * npix + n0) * 3);
                o[0] = make_float4(pt[0][0], pt[0][1], pt[0][2], pt[1][0]);
                o[1] = make_float4(pt[1][1], pt[1][2], pt[2][0], pt[2][1]);
                o[2] = make_float4(pt[2][2], pt[3][0], pt[3][1], pt[3][2]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (P.inv_up) P.inv_up[(size_t)b * npix + n0 + e] = iv[e];
                if (P.seg_up) {
#pragma unroll
                    for (int c = 0; c < C; ++c) P.seg_up[((size_t)b * C + c) * npix + n0 + e] = sem[c][e];
                }
                if (P.points) {
                    float* o = P.points + ((size_t)b * npix + n0 + e) * 3;
                    o[0] = pt[e][0];
                    o[1] = pt[e][1];
                    o[2] = pt[e][2];
                }
            }
        }
    }
}

// Row-segment variant (the one that runs for the camera sizes in use): one workgroup = one camera row x 1024
// pixels.  The 4 bicubic source rows of inv and the nearest source row of each class are staged once in LDS
// (7 x SW floats, coalesced), so the 16 bicubic taps per pixel are LDS reads instead of per-lane global gathers
// (the gather form was bound by the texture-address path, not by HBM).  Arithmetic is identical to project_kernel.
template <int C, int SW>
__global__ __launch_bounds__(256) void project_rows_kernel(ProjParams P, int nseg) {
    __shared__ float s_inv[4][SW];
    __shared__ float s_seg[C][SW];
    int bid = blockIdx.x;
    const int seg = bid % nseg;
    bid /= nseg;
    const int u = bid % P.Hc;
    const int b = bid / P.Hc;
    const float sy = (float)P.h / (float)P.Hc;
    const float sx = (float)P.w / (float)P.Wc;
    const size_t npix = (size_t)P.Hc * P.Wc;
    const int vlo = seg * 1024;
    const int vhi = (vlo + 1024 < P.Wc) ? vlo + 1024 : P.Wc;
    const Taps ty = cubic_taps(u, P.h, sy);
    const int su = nearest_src(u, P.h, sy);
    // source column window of this segment
    int c_lo = (int)floorf(fmaf(sx, (float)vlo + 0.5f, -0.5f)) - 1;
    c_lo = c_lo < 0 ? 0 : c_lo;
    {
        const float* src = P.inv + (size_t)b * P.h * P.w;
        int col = c_lo + (int)threadIdx.x;
        col = col > P.w - 1 ? P.w - 1 : col;
        if (threadIdx.x < SW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s_inv[i][threadIdx.x] = src[(size_t)ty.idx[i] * P.w + col];
#pragma unroll
            for (int c = 0; c < C; ++c) s_seg[c][threadIdx.x] = P.seg[(((size_t)b * C + c) * P.h + su) * P.w + col];
        }
    }
    __syncthreads();
    const int v0 = vlo + (int)threadIdx.x * 4;
    const bool valid = v0 < vhi;   // (no early return: every thread reaches the barrier of the voxel de-duplication below)
    const float yterm = (float)u - P.cy;
    float iv[4], sem[C][4], pt[4][3];
    int vkey[4];       // first cell index of the pixel's voxel (its C class bits are adjacent), -1 = not in the grid
    uint32_t vcm[4];   // classes with non-zero probability
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vkey[e] = -1;
        vcm[e] = 0;
    }
    if (valid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = v0 + e;
        const Taps tx = cubic_taps(v, P.w, sx);
        const int i0 = tx.idx[0] - c_lo, i1 = tx.idx[1] - c_lo, i2 = tx.idx[2] - c_lo, i3 = tx.idx[3] - c_lo;
        float t0 = dot4(s_inv[0][i0], s_inv[0][i1], s_inv[0][i2], s_inv[0][i3], tx.w);
        float t1 = dot4(s_inv[1][i0], s_inv[1][i1], s_inv[1][i2], s_inv[1][i3], tx.w);
        float t2 = dot4(s_inv[2][i0], s_inv[2][i1], s_inv[2][i2], s_inv[2][i3], tx.w);
        float t3 = dot4(s_inv[3][i0], s_inv[3][i1], s_inv[3][i2], s_inv[3][i3], tx.w);
        float val = dot4(t0, t1, t2, t3, ty.w);
        if (val < 1e-8f) val = 1e-8f;  // NaN compares false and stays NaN
        float d = 1.0f / val;
        if (isinf(d) || isnan(d)) d = __builtin_inff();
        iv[e] = val;
        const int sv = nearest_src(v, P.w, sx) - c_lo;
#pragma unroll
        for (int c = 0; c < C; ++c) sem[c][e] = s_seg[c][sv];
        float p[3];
        p[0] = (((float)v - P.cx) * d) / P.fx;
        p[1] = (yterm * d) / P.fy;
        p[2] = d;
        const size_t n = (size_t)u * P.Wc + v;
        if (n < 3) {  // the reference scales/shifts flat pixels 0,1,2 of each image
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = p[k] * P.pc_scale[n] + P.pc_shift[n];
        }
        pt[e][0] = p[0];
        pt[e][1] = p[1];
        pt[e][2] = p[2];
        if (P.occ_bits) {
            float a[3], bq[3], cq[3];
            rot3(p, P.rot, a);
            rot3(a, P.rot + 9, bq);
            rot3(bq, P.rot + 18, cq);
            const bool fin = isfinite(cq[0]) && isfinite(cq[1]) && isfinite(cq[2]);
            const float fi = (cq[0] / P.occ_shape[0]) * (float)P.grid[0];
            const float fj = (cq[1] / P.occ_shape[1]) * (float)P.grid[1];
            const float fk = (cq[2] / P.occ_shape[2]) * (float)P.grid[2];
            const bool inr = fin && fi > -1.0f && fi < 65536.0f && fj > -1.0f && fj < 65536.0f && fk > -1.0f && fk < 65536.0f;
            if (inr) {
                const int i = (int)fi, j = (int)fj, k = (int)fk;
                if (0 < i && i < P.grid[0] && 0 < j && j < P.grid[1] && 0 < k && k < P.grid[2]) {
                    uint32_t cm = 0;
#pragma unroll
                    for (int c = 0; c < C; ++c) cm |= (sem[c][e] != 0.0f) ? (1u << c) : 0u;
                    if (cm) {
                        vkey[e] = ((i * P.grid[1] + j) * P.grid[2] + k) * C;
                        vcm[e] = cm;
                    }
                }
            }
        }
    }
    }  // valid
    // ---- voxel marking with run-length de-duplication along the camera row ----
    // Neighbouring pixels mostly fall into the same voxel (a 0.5 m cell spans tens of pixels), so a pixel only touches
    // the grid when its (voxel, class set) is not covered by its left neighbour's: by induction every pixel's bits are
    // then set by the nearest acting pixel to its left.  The first pixel of each WORKGROUP always acts; the first lane of the
    // other waves takes its left neighbour from the previous wave through LDS (a forced action per wave is a dependent
    // load -> atomic chain that keeps the wave alive for a memory round trip).
    if (P.occ_bits) {
        __shared__ int s_lk[4];
        __shared__ uint32_t s_lc[4];
        if ((threadIdx.x & 63) == 63) { s_lk[threadIdx.x >> 6] = vkey[3]; s_lc[threadIdx.x >> 6] = vcm[3]; }
        __syncthreads();
        int pk = __shfl_up(vkey[3], 1);
        uint32_t pc = __shfl_up(vcm[3], 1);
        if ((threadIdx.x & 63) == 0) {
            if (threadIdx.x == 0) { pk = -2; pc = 0; }
            else { pk = s_lk[(threadIdx.x >> 6) - 1]; pc = s_lc[(threadIdx.x >> 6) - 1]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (vkey[e] >= 0 && (vkey[e] != pk || (vcm[e] & ~pc))) {
                const uint32_t bit = (uint32_t)vkey[e];
                const unsigned long long m = (unsigned long long)vcm[e] << (bit & 31);
                const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
                uint32_t* wp = P.occ_bits + (bit >> 5);
                // idempotent OR: a stale read only costs a redundant atomic, so the check may be served by the nearest cache
                // (workgroup scope; an agent-scope load bypasses L2 on gfx950 and made scattered scenes 11 % slower)
                if ((__hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & mlo) != mlo) atomicOr(wp, mlo);
                if (mhi && (__hip_atomic_load(wp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & mhi) != mhi) atomicOr(wp + 1, mhi);
            }
            pk = vkey[e];
            pc = vcm[e];
        }
    }
    const size_t n0 = (size_t)u * P.Wc + v0;
    if (valid) {
        typedef __attribute__((ext_vector_type(4))) float v4f;
        if (P.inv_up) __builtin_nontemporal_store(v4f{iv[0], iv[1], iv[2], iv[3]}, reinterpret_cast<v4f*>(P.inv_up + (size_t)b * npix + n0));
        if (P.seg_up) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                __builtin_nontemporal_store(v4f{sem[c][0], sem[c][1], sem[c][2], sem[c][3]}, reinterpret_cast<v4f*>(P.seg_up + ((size_t)b * C + c) * npix + n0));
        }
    }
    if (P.points) {
        // a thread's 4 points are 48 contiguous bytes: written directly, each of the 3 store instructions of a wave touches every third
        // 16-byte slot.  Through LDS the workgroup's 12 KB leave as three fully contiguous 4 KB sweeps instead.
        __shared__ float4 s_pts[3 * 256];
        if (valid) {
            s_pts[3 * threadIdx.x + 0] = make_float4(pt[0][0], pt[0][1], pt[0][2], pt[1][0]);
            s_pts[3 * threadIdx.x + 1] = make_float4(pt[1][1], pt[1][2], pt[2][0], pt[2][1]);
            s_pts[3 * threadIdx.x + 2] = make_float4(pt[2][2], pt[3][0], pt[3][1], pt[3][2]);
        }
        __syncthreads();
        const int nf4 = ((vhi - vlo) / 4) * 3;   // float4 slots of this row segment
        float4* o = reinterpret_cast<float4*>(P.points + ((size_t)b * npix + (size_t)u * P.Wc + vlo) * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = k * 256 + (int)threadIdx.x;
            if (idx < nf4) {
                typedef __attribute__((ext_vector_type(4))) float v4f;
                const float4 q = s_pts[idx];
                __builtin_nontemporal_store(v4f{q.x, q.y, q.z, q.w}, reinterpret_cast<v4f*>(o + idx));
            }
        }
    }
}


// Multi-row form of project_rows_kernel (round 3): one workgroup = R consecutive camera rows x 1024 pixels.  The round-2 kernel was bound by
// VALU issue as much as by HBM (~250 instructions per pixel: the x-direction bicubic taps and weights, six IEEE divisions, three K = 3
// rotations), so this form removes arithmetic that does not depend on the row:
//   * a thread keeps its 4 columns' x taps / weights / nearest column / (v - cx) over the R rows (cubic_taps is ~45 instructions);
//   * the R rows share their bicubic source rows (4 + ceil((R-1) h/Hc) + 1 <= NR rows) and nearest class rows (<= 2), staged once;
//   * rotations Rb, Rc are skipped when they are exactly the identity (the constructor's correction_angle = (7, 0, 0)): for finite points
//     fma(z, 0, fma(y, 0, x * 1)) == x bit for bit, and a non-finite coordinate leaves the voxel out either way (the isfinite test);
//   * one barrier for staging and one for the cross-wave voxel de-duplication per workgroup instead of per row; the points leave through a
//     wave-private LDS transpose (no barrier); the voxel check loads of all R rows are issued together, then the ORs;
//   * the run-length de-duplication of the voxel marks is two-dimensional inside the group: a pixel covered by its left neighbour OR by the
//     pixel above it does not touch the grid (a 0.5 m voxel spans several rows as well as tens of columns).
// Measured in the network (B = 8, profiles/r03_*): 142 -> 111 us (3.33 -> 4.25 TB/s of algorithmic bytes).  Tried on top and dropped: R = 2 / 3 (slower),
// software-pipelined marking without the barrier (133 us: a check load is only consumable after the row's older stores have drained, vmcnt is
// in-order), forcing 5 waves per SIMD (spills: 184 us), and multiply-and-correct division by the five run-time constants (verified bit-equal to
// IEEE division for all 2^23 significands per constant, but no faster: 111 us -- the kernel is not VALU-bound any more).
// Per-pixel arithmetic, its order and therefore every output bit are those of project_rows_kernel / oracle/projection_ref.c.
template <int C, int SW, int R, int NR, int NS>
__global__ __launch_bounds__(256) void project_rowsR_kernel(ProjParams P, int nseg, int rot_bc_identity) {
    __shared__ float s_inv[NR][SW];
    __shared__ float s_seg[C][NS][SW];   // NS nearest-neighbour class rows cover the group (2 for R = 4, 3 for R = 8 at 256 -> 1080)
    __shared__ float4 s_pts[4][3 * 64];   // per wave: 64 threads x 3 float4
    __shared__ int s_lk[R][4];
    __shared__ uint32_t s_lc[R][4];
    int bid = blockIdx.x;
    const int seg = bid % nseg;
    bid /= nseg;
    const int ngrp = (P.Hc + R - 1) / R;
    const int u0 = (bid % ngrp) * R;
    const int b = bid / ngrp;
    const float sy = (float)P.h / (float)P.Hc;
    const float sx = (float)P.w / (float)P.Wc;
    const size_t npix = (size_t)P.Hc * P.Wc;
    const int vlo = seg * 1024;
    const int vhi = (vlo + 1024 < P.Wc) ? vlo + 1024 : P.Wc;
    const int ulast = (u0 + R - 1 < P.Hc) ? u0 + R - 1 : P.Hc - 1;
    // source rows of the group: idx[0] of the first row .. idx[3] of the last (taps are clamped into [0, h-1], monotone in u)
    const int r_lo = cubic_taps(u0, P.h, sy).idx[0];
    const int su0 = nearest_src(u0, P.h, sy);
    int c_lo = (int)floorf(fmaf(sx, (float)vlo + 0.5f, -0.5f)) - 1;
    c_lo = c_lo < 0 ? 0 : c_lo;
    {
        const float* src = P.inv + (size_t)b * P.h * P.w;
        int col = c_lo + (int)threadIdx.x;
        col = col > P.w - 1 ? P.w - 1 : col;
        if (threadIdx.x < SW) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                int rr = r_lo + i;
                rr = rr > P.h - 1 ? P.h - 1 : rr;
                s_inv[i][threadIdx.x] = src[(size_t)rr * P.w + col];
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    int rr = su0 + k;
                    rr = rr > P.h - 1 ? P.h - 1 : rr;
                    s_seg[c][k][threadIdx.x] = P.seg[(((size_t)b * C + c) * P.h + rr) * P.w + col];
                }
        }
    }
    __syncthreads();
    const int v0 = vlo + (int)threadIdx.x * 4;
    const bool valid = v0 < vhi;   // (no early return: every thread reaches the barrier below)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // ---- per-column state, shared by the R rows ----
    int ci[4][4], csv[4];
    float cw[4][4], xt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = v0 + e;
        const Taps tx = cubic_taps(v, P.w, sx);
#pragma unroll
        for (int k = 0; k < 4; ++k) { ci[e][k] = tx.idx[k] - c_lo; cw[e][k] = tx.w[k]; }
        csv[e] = nearest_src(v, P.w, sx) - c_lo;
        xt[e] = (float)v - P.cx;
        if (!valid) {   // keep LDS indices in range for the (unused) lanes past the row end
#pragma unroll
            for (int k = 0; k < 4; ++k) ci[e][k] = 0;
            csv[e] = 0;
        }
    }
    int vkey[R][4];       // first cell index of the pixel's voxel (its C class bits are adjacent), -1 = not in the grid
    uint32_t vcm[R][4];   // classes with non-zero probability
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int u = u0 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vkey[r][e] = -1; vcm[r][e] = 0; }
        if (u > ulast) continue;    // wave-uniform
        const Taps ty = cubic_taps(u, P.h, sy);
        const int sr = nearest_src(u, P.h, sy) - su0;
        const float* q0 = s_inv[ty.idx[0] - r_lo];
        const float* q1 = s_inv[ty.idx[1] - r_lo];
        const float* q2 = s_inv[ty.idx[2] - r_lo];
        const float* q3 = s_inv[ty.idx[3] - r_lo];
        const float yterm = (float)u - P.cy;
        float iv[4], sem[C][4], pt[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = v0 + e;
            const int i0 = ci[e][0], i1 = ci[e][1], i2 = ci[e][2], i3 = ci[e][3];
            float t0 = dot4(q0[i0], q0[i1], q0[i2], q0[i3], cw[e]);
            float t1 = dot4(q1[i0], q1[i1], q1[i2], q1[i3], cw[e]);
            float t2 = dot4(q2[i0], q2[i1], q2[i2], q2[i3], cw[e]);
            float t3 = dot4(q3[i0], q3[i1], q3[i2], q3[i3], cw[e]);
            float val = dot4(t0, t1, t2, t3, ty.w);
            if (val < 1e-8f) val = 1e-8f;  // NaN compares false and stays NaN
            float d = 1.0f / val;
            if (isinf(d) || isnan(d)) d = __builtin_inff();
            iv[e] = val;
#pragma unroll
            for (int c = 0; c < C; ++c) sem[c][e] = s_seg[c][sr][csv[e]];
            float p[3];
            p[0] = (xt[e] * d) / P.fx;
            p[1] = (yterm * d) / P.fy;
            p[2] = d;
            const size_t n = (size_t)u * P.Wc + v;
            if (n < 3) {  // the reference scales/shifts flat pixels 0,1,2 of each image
#pragma unroll
                for (int k = 0; k < 3; ++k) p[k] = p[k] * P.pc_scale[n] + P.pc_shift[n];
            }
            pt[e][0] = p[0];
            pt[e][1] = p[1];
            pt[e][2] = p[2];
            if (P.occ_bits && valid) {
                float a[3], cq[3];
                rot3(p, P.rot, a);
                if (rot_bc_identity) {
                    cq[0] = a[0]; cq[1] = a[1]; cq[2] = a[2];
                } else {
                    float bq[3];
                    rot3(a, P.rot + 9, bq);
                    rot3(bq, P.rot + 18, cq);
                }
                const bool fin = isfinite(cq[0]) && isfinite(cq[1]) && isfinite(cq[2]);
                const float fi = (cq[0] / P.occ_shape[0]) * (float)P.grid[0];
                const float fj = (cq[1] / P.occ_shape[1]) * (float)P.grid[1];
                const float fk = (cq[2] / P.occ_shape[2]) * (float)P.grid[2];
                const bool inr = fin && fi > -1.0f && fi < 65536.0f && fj > -1.0f && fj < 65536.0f && fk > -1.0f && fk < 65536.0f;
                if (inr) {
                    const int i = (int)fi, j = (int)fj, k = (int)fk;
                    if (0 < i && i < P.grid[0] && 0 < j && j < P.grid[1] && 0 < k && k < P.grid[2]) {
                        uint32_t cm = 0;
#pragma unroll
                        for (int c = 0; c < C; ++c) cm |= (sem[c][e] != 0.0f) ? (1u << c) : 0u;
                        if (cm) {
                            vkey[r][e] = ((i * P.grid[1] + j) * P.grid[2] + k) * C;
                            vcm[r][e] = cm;
                        }
                    }
                }
            }
        }
        // ---- this row's stores ----
        const size_t n0 = (size_t)u * P.Wc + v0;
        typedef __attribute__((ext_vector_type(4))) float v4f;
        if (valid) {
            if (P.inv_up) __builtin_nontemporal_store(v4f{iv[0], iv[1], iv[2], iv[3]}, reinterpret_cast<v4f*>(P.inv_up + (size_t)b * npix + n0));
            if (P.seg_up) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    __builtin_nontemporal_store(v4f{sem[c][0], sem[c][1], sem[c][2], sem[c][3]}, reinterpret_cast<v4f*>(P.seg_up + ((size_t)b * C + c) * npix + n0));
            }
        }
        if (P.points) {
            // a thread's 4 points are 48 contiguous bytes; a wave's 64 threads own 3 KB of contiguous output: through the wave's private LDS
            // slab they leave as three fully contiguous 1 KB stores (no workgroup barrier: only this wave touches the slab)
            float4* sp = s_pts[wave];
            sp[3 * lane + 0] = make_float4(pt[0][0], pt[0][1], pt[0][2], pt[1][0]);
            sp[3 * lane + 1] = make_float4(pt[1][1], pt[1][2], pt[2][0], pt[2][1]);
            sp[3 * lane + 2] = make_float4(pt[2][2], pt[3][0], pt[3][1], pt[3][2]);
            __builtin_amdgcn_wave_barrier();
            const int wv0 = vlo + wave * 256;                  // first pixel of this wave
            const int nf4 = wv0 < vhi ? ((vhi - wv0 < 256 ? vhi - wv0 : 256) / 4) * 3 : 0;   // float4 slots the wave owns in this row
            float4* o = reinterpret_cast<float4*>(P.points + ((size_t)b * npix + (size_t)u * P.Wc + wv0) * 3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = k * 64 + lane;
                if (idx < nf4) {
                    const float4 qv = sp[idx];
                    __builtin_nontemporal_store(v4f{qv.x, qv.y, qv.z, qv.w}, reinterpret_cast<v4f*>(o + idx));
                }
            }
            __builtin_amdgcn_wave_barrier();   // the slab is rewritten for the next row
        }
    }
    // ---- voxel marking with run-length de-duplication along each camera row (see project_rows_kernel) ----
    if (P.occ_bits) {
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < R; ++r) { s_lk[r][wave] = vkey[r][3]; s_lc[r][wave] = vcm[r][3]; }
        }
        __syncthreads();
        uint32_t cur_lo[R][4], cur_hi[R][4];
        bool act[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int pk = __shfl_up(vkey[r][3], 1);
            uint32_t pc = __shfl_up(vcm[r][3], 1);
            if (lane == 0) {
                if (wave == 0) { pk = -2; pc = 0; }
                else { pk = s_lk[r][wave - 1]; pc = s_lc[r][wave - 1]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                act[r][e] = vkey[r][e] >= 0 && (vkey[r][e] != pk || (vcm[r][e] & ~pc));
                // ... and not covered by the pixel ABOVE it in this group (same thread, previous row): a 0.5 m voxel spans several camera rows too.
                // By induction along left / up chains every skipped pixel's bits are set by an acting pixel (row 0 chains left to the first
                // pixel of the row, which always acts).
                if (r > 0 && vkey[r][e] == vkey[r - 1][e] && !(vcm[r][e] & ~vcm[r - 1][e])) act[r][e] = false;
                cur_lo[r][e] = ~0u; cur_hi[r][e] = ~0u;
                if (act[r][e]) {   // all R x 4 check loads go out before the first result is needed
                    const uint32_t bit = (uint32_t)vkey[r][e];
                    const uint32_t* wp = P.occ_bits + (bit >> 5);
                    cur_lo[r][e] = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((bit & 31) + C > 32) cur_hi[r][e] = __hip_atomic_load(wp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pk = vkey[r][e];
                pc = vcm[r][e];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (act[r][e]) {
                    const uint32_t bit = (uint32_t)vkey[r][e];
                    const unsigned long long m = (unsigned long long)vcm[r][e] << (bit & 31);
                    const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
                    uint32_t* wp = P.occ_bits + (bit >> 5);
                    // idempotent OR: a stale read only costs a redundant atomic
                    if ((cur_lo[r][e] & mlo) != mlo) atomicOr(wp, mlo);
                    if (mhi && (cur_hi[r][e] & mhi) != mhi) atomicOr(wp + 1, mhi);
                }
            }
    }
}

// bits -> f32, every batch row gets the same union grid.  One thread = 4 consecutive cells.
__global__ __launch_bounds__(256) void occ_expand_kernel(const uint32_t* __restrict__ bits, float* __restrict__ occ, size_t ncell, int B) {
    const size_t nq = ncell / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        const size_t n = q * 4;
        const uint32_t wv = bits[n >> 5] >> (n & 31);
        const float4 val = make_float4((float)(wv & 1u), (float)((wv >> 1) & 1u), (float)((wv >> 2) & 1u), (float)((wv >> 3) & 1u));
        for (int b = 0; b < B; ++b) {   // write-once 25 MB rows: streamed past the caches
            typedef __attribute__((ext_vector_type(4))) float v4f;
            __builtin_nontemporal_store(v4f{val.x, val.y, val.z, val.w}, reinterpret_cast<v4f*>(occ + (size_t)b * ncell + n));
        }
    }
}

// The same expansion in two halves, for the multi-GPU path (soccdpt_amd/dist.py): occ_zero_kernel writes the B x ncell zeros -- independent of every
// rank's voxels, so it runs WHILE the RCCL all-gather of the packed grids is in flight -- and occ_set_kernel, after the union, writes a 1 into every
// row for each set bit (~1e4 of 6.3 M cells).  Together they store exactly what occ_expand_kernel stores.
__global__ __launch_bounds__(256) void occ_zero_kernel(float* __restrict__ occ, size_t n4) {
    typedef __attribute__((ext_vector_type(4))) float v4f;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f*>(occ) + q);
}
__global__ __launch_bounds__(256) void occ_set_kernel(const uint32_t* __restrict__ bits, float* __restrict__ occ, size_t ncell, size_t nwords, int B) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t wv = bits[i];
        while (wv) {
            const int k = __builtin_ctz(wv);
            wv &= wv - 1;
            const size_t n = i * 32 + (size_t)k;
            if (n < ncell)
                for (int b = 0; b < B; ++b) occ[(size_t)b * ncell + n] = 1.0f;
        }
    }
}

__global__ __launch_bounds__(256) void occ_or_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t nwords, int nsets) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = dst[i];
        for (int s = 0; s < nsets; ++s) v |= src[(size_t)s * nwords + i];
        dst[i] = v;
    }
}

int launch_project(const soccdpt_config& cfg, const float* inv, const float* seg, int B, int in_h, int in_w, float* inv_up,
                   float* seg_up, float* points, uint32_t* occ_bits, int clear_bits, hipStream_t stream, std::string& err) {
    if (cfg.num_classes != 3) {
        err = "projection kernel is instantiated for num_classes == 3 (the reference reshapes xyz with num_classes)";
        return 1;
    }
    if (B <= 0 || in_h <= 0 || in_w <= 0) {
        err = "soccdpt_project: empty input";
        return 1;
    }
    const size_t ncell = (size_t)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] * cfg.num_classes;
    if (ncell >= (1ull << 31) || cfg.grid[0] > 65535 || cfg.grid[1] > 65535 || cfg.grid[2] > 65535) {
        err = "occupancy grid too large for 32-bit cell indices";
        return 1;
    }
    ProjParams P;
    P.inv = inv; P.seg = seg; P.inv_up = inv_up; P.seg_up = seg_up; P.points = points; P.occ_bits = occ_bits;
    P.B = B; P.h = in_h; P.w = in_w; P.Hc = cfg.cam_height; P.Wc = cfg.cam_width;
    P.fx = cfg.fx; P.fy = cfg.fy; P.cx = cfg.cx; P.cy = cfg.cy;
    for (int i = 0; i < 3; ++i) {
        P.pc_scale[i] = cfg.pc_scale[i];
        P.pc_shift[i] = cfg.pc_shift[i];
        P.occ_shape[i] = cfg.occupancy_shape[i];
        P.grid[i] = cfg.grid[i];
    }
    for (int i = 0; i < 27; ++i) P.rot[i] = cfg.rot[i];
    if (occ_bits && clear_bits) {
        hipError_t e = hipMemsetAsync(occ_bits, 0, ((ncell + 31) / 32) * sizeof(uint32_t), stream);
        if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    }
    const bool vec4 = (P.Wc % 4 == 0);
    constexpr int SW = 256;
    const float sx_h = (float)P.w / (float)P.Wc;
    if (vec4 && (int)(1024.0f * sx_h) + 8 <= SW) {
        const int nseg = (P.Wc + 1023) / 1024;
        constexpr int R = 4, NR = 7;
        // the R rows of a group need source rows idx0(u0) .. idx3(u0 + R - 1): at most 4 + ceil((R - 1) h / Hc) + 1
        const float sy_h = (float)P.h / (float)P.Hc;
        static const int force_rows1 = getenv("SOCCDPT_PROJECT_ROWS1") ? atoi(getenv("SOCCDPT_PROJECT_ROWS1")) : 0;   // A/B against the one-row kernel
        static const int rows8 = getenv("SOCCDPT_PROJECT_ROWS8") ? atoi(getenv("SOCCDPT_PROJECT_ROWS8")) : 0;         // A/B: 8 camera rows per workgroup
        bool ident = true;   // Rb and Rc exactly the identity? (rotate_points with b = c = 0)
        for (int i = 0; i < 9; ++i) ident = ident && P.rot[9 + i] == ((i % 4 == 0) ? 1.0f : 0.0f) && P.rot[18 + i] == ((i % 4 == 0) ? 1.0f : 0.0f);
        if (rows8 && !force_rows1 && 5 + (int)ceilf(7 * sy_h) <= 7 && 1.0f + 7 * sy_h < 3.0f) {
            const int ngrp = (P.Hc + 7) / 8;
            SOCCDPT_LAUNCH((project_rowsR_kernel<3, SW, 8, 7, 3>), dim3((unsigned)(B * ngrp * nseg)), dim3(256), 0, stream, P, nseg, ident ? 1 : 0);
            hipError_t e3 = hipGetLastError();
            if (e3 != hipSuccess) { err = hipGetErrorString(e3); return 1; }
            return 0;
        }
        if (!force_rows1 && 5 + (int)ceilf((R - 1) * sy_h) <= NR && 1.0f + (R - 1) * sy_h < 2.0f) {
            const int ngrp = (P.Hc + R - 1) / R;
            SOCCDPT_LAUNCH((project_rowsR_kernel<3, SW, R, NR, 2>), dim3((unsigned)(B * ngrp * nseg)), dim3(256), 0, stream, P, nseg, ident ? 1 : 0);
            hipError_t e3 = hipGetLastError();
            if (e3 != hipSuccess) { err = hipGetErrorString(e3); return 1; }
            return 0;
        }
        SOCCDPT_LAUNCH((project_rows_kernel<3, SW>), dim3((unsigned)(B * P.Hc * nseg)), dim3(256), 0, stream, P, nseg);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { err = hipGetErrorString(e2); return 1; }
        return 0;
    }
    const long long total = (long long)B * P.Hc * (vec4 ? P.Wc / 4 : P.Wc);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (vec4)
        SOCCDPT_LAUNCH((project_kernel<3, 4>), dim3((unsigned)blocks), dim3(256), 0, stream, P);
    else
        SOCCDPT_LAUNCH((project_kernel<3, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    return 0;
}

int launch_occ_expand(const soccdpt_config& cfg, const uint32_t* bits, int B, float* occ, hipStream_t stream, std::string& err) {
    const size_t ncell = (size_t)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] * cfg.num_classes;
    if (ncell % 32 != 0) { err = "occupancy cell count must be a multiple of 32"; return 1; }
    size_t blocks = (ncell / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    SOCCDPT_LAUNCH(occ_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bits, occ, ncell, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    return 0;
}

int launch_occ_zero(const soccdpt_config& cfg, int B, float* occ, hipStream_t stream, std::string& err) {
    const size_t ncell = (size_t)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] * cfg.num_classes;
    if (ncell % 4 != 0) { err = "occupancy cell count must be a multiple of 4"; return 1; }
    const size_t n4 = ncell / 4 * (size_t)B;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    SOCCDPT_LAUNCH(occ_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, occ, n4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    return 0;
}

int launch_occ_set(const soccdpt_config& cfg, const uint32_t* bits, int B, float* occ, hipStream_t stream, std::string& err) {
    const size_t ncell = (size_t)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] * cfg.num_classes;
    const size_t nwords = (ncell + 31) / 32;
    SOCCDPT_LAUNCH(occ_set_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, bits, occ, ncell, nwords, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    return 0;
}

int launch_occ_or(const soccdpt_config& cfg, uint32_t* dst, const uint32_t* src, int nsets, hipStream_t stream, std::string& err) {
    const size_t ncell = (size_t)cfg.grid[0] * cfg.grid[1] * cfg.grid[2] * cfg.num_classes;
    const size_t nwords = (ncell + 31) / 32;
    size_t blocks = (nwords + 255) / 256;
    SOCCDPT_LAUNCH(occ_or_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, src, nwords, nsets);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = hipGetErrorString(e); return 1; }
    return 0;
}

}  // namespace soccdpt
