// One row of the Swin-V2 post-norm residual LayerNorm (elementwise.hip: ln_residual_kernel) as a device function: one wave = one row.  A device
// function so that a caller can pass its own row index and lane.
#pragma once
#include <type_traits>

#include "half16.h"
#include "kernels.h"

namespace soccdpt {

// 64-lane sum, result in every lane: 4 DPP steps inside each 16-lane row (VALU, no LDS traffic) + 4 v_readlane.
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    const int vi = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
}

template <int VPL, bool F16>  // values per lane = ceil(C / 64)
__device__ __forceinline__ void ln_residual_row(const float* __restrict__ y, const float* __restrict__ g,
                                                const float* __restrict__ beta, float* __restrict__ xf,
                                                bf16_t* __restrict__ xb, bf16_t* __restrict__ halo, float* __restrict__ halo_f32, int M,
                                                int C, int residual, int res /*spatial size for halo / merge*/, int merge, int x3, int x3h,
                                                const float* __restrict__ row_scale, int rows_per_scale, int row, int lane) {
    if (row >= M) return;
    const float* yr = y + (size_t)row * C;
    float v[VPL], xr[VPL], gg[VPL], bb[VPL];
    float s = 0.f;
    // everything the row needs is requested up front (one memory latency instead of two: the kernel is a few microseconds long)
#pragma unroll
    for (int t = 0; t < VPL; ++t) {
        const int c = lane + 64 * t;
        v[t] = c < C ? yr[c] : 0.f;
        xr[t] = (residual && c < C) ? xf[(size_t)row * C + c] : 0.f;
        gg[t] = c < C ? g[c] : 0.f;
        bb[t] = c < C ? beta[c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < VPL; ++t) s += v[t];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < VPL; ++t) {
        const int c = lane + 64 * t;
        const float d = c < C ? v[t] - mean : 0.f;
        v[t] = d;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
    size_t hoff = 0, boff = (size_t)row * C;
    if (halo || halo_f32 || merge) {
        const int hw = res * res, b = row / hw, r = row - b * hw, yy = r / res, xx = r - yy * res;
        hoff = ((size_t)(b * (res + 2) + yy + 1) * (res + 2) + xx + 1) * C;
        // merge: the operand copy goes straight into the PatchMerging layout [B][res/2][res/2][4C], channel block (yy&1) + 2*(xx&1)
        // (timm order x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]): the gather kernel before the reduction GEMM disappears
        if (merge) boff = (((size_t)(b * (res / 2) + yy / 2) * (res / 2) + xx / 2) * 4 + (yy & 1) + 2 * (xx & 1)) * C;
    }
#pragma unroll
    for (int t = 0; t < VPL; ++t) {
        const int c = lane + 64 * t;
        if (c < C) {
            float o = v[t] * rstd * gg[t] + bb[t];
            if (row_scale) o *= row_scale[row / rows_per_scale];
            if (residual) o += xr[t];
            xf[(size_t)row * C + c] = o;
            const bf16_t ob = f2h<F16>(o);
            if (xb) { if (x3) x3_store1(xb, boff + c, o); else xb[boff + c] = ob; }          // x3 / x3h: that output is an x3 tensor (half16.h)
            if (halo) { if (x3h) x3_store1(halo, hoff + c, o); else halo[hoff + c] = ob; }
            if (halo_f32) halo_f32[hoff + c] = o;
        }
    }
}

// Same, for C % 256 == 0 (C = 256 / 512 / 768 / 1024): a lane owns float4 groups (16-byte loads and stores, a quarter of the memory
// instructions; base_384's 47 launches per forward are mostly these widths).
template <int V4, bool F16>  // float4 groups per lane = C / 256
__device__ __forceinline__ void ln_residual_v4_row(const float* __restrict__ y, const float* __restrict__ g,
                                                   const float* __restrict__ beta, float* __restrict__ xf,
                                                   bf16_t* __restrict__ xb, bf16_t* __restrict__ halo, float* __restrict__ halo_f32, int M,
                                                   int C, int residual, int res, int merge, int x3, int x3h,
                                                   const float* __restrict__ row_scale, int rows_per_scale, int row, int lane) {
    if (row >= M) return;
    const float4* yr = reinterpret_cast<const float4*>(y + (size_t)row * C);
    const float4* xr4 = reinterpret_cast<const float4*>(xf + (size_t)row * C);
    float4 v[V4], xr[V4], gg[V4], bb[V4];
#pragma unroll
    for (int t = 0; t < V4; ++t) {
        const int c4 = lane + 64 * t;
        v[t] = yr[c4];
        xr[t] = residual ? xr4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        gg[t] = reinterpret_cast<const float4*>(g)[c4];
        bb[t] = reinterpret_cast<const float4*>(beta)[c4];
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < V4; ++t) s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < V4; ++t) {
        v[t].x -= mean; v[t].y -= mean; v[t].z -= mean; v[t].w -= mean;
        q += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
    size_t hoff = 0, boff = (size_t)row * C;
    if (halo || halo_f32 || merge) {
        const int hw = res * res, b = row / hw, r = row - b * hw, yy = r / res, xx = r - yy * res;
        hoff = ((size_t)(b * (res + 2) + yy + 1) * (res + 2) + xx + 1) * C;
        if (merge) boff = (((size_t)(b * (res / 2) + yy / 2) * (res / 2) + xx / 2) * 4 + (yy & 1) + 2 * (xx & 1)) * C;
    }
#pragma unroll
    for (int t = 0; t < V4; ++t) {
        const int c = 4 * (lane + 64 * t);
        float4 o;
        o.x = v[t].x * rstd * gg[t].x + bb[t].x;
        o.y = v[t].y * rstd * gg[t].y + bb[t].y;
        o.z = v[t].z * rstd * gg[t].z + bb[t].z;
        o.w = v[t].w * rstd * gg[t].w + bb[t].w;
        if (row_scale) { const float rs = row_scale[row / rows_per_scale]; o.x *= rs; o.y *= rs; o.z *= rs; o.w *= rs; }
        o.x += xr[t].x; o.y += xr[t].y; o.z += xr[t].z; o.w += xr[t].w;
        *reinterpret_cast<float4*>(xf + (size_t)row * C + c) = o;
        uint2 ob;
        ob.x = pack_h2<F16>(o.x, o.y);
        ob.y = pack_h2<F16>(o.z, o.w);
        if (xb) { if (x3) x3_store4(xb, boff + c, o.x, o.y, o.z, o.w); else *reinterpret_cast<uint2*>(xb + boff + c) = ob; }
        if (halo) { if (x3h) x3_store4(halo, hoff + c, o.x, o.y, o.z, o.w); else *reinterpret_cast<uint2*>(halo + hoff + c) = ob; }
        if (halo_f32) *reinterpret_cast<float4*>(halo_f32 + hoff + c) = o;
    }
}

}  // namespace soccdpt
