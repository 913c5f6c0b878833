// HBM-bound glue kernels of the SOccDPT_V3 network on gfx950: patch embedding + LayerNorm,
// post-norm residual LayerNorm, patch-merging gather, align_corners bilinear resampling, the
// 3-channel tail of the seg head, and the one-time weight re-layout / CPB-table kernels.
// Every kernel cites the reference computation it replaces; 64-lane waves, 16-byte accesses
// where the layout allows, NHWC ("token-major") activations throughout.
#include <type_traits>

#include "half16.h"
#include "kernels.h"
#include "ln_body.h"

namespace soccdpt {

// Sum over each 16-lane row, result in every lane of the row: 4 DPP rotations (VALU only).
__device__ __forceinline__ float row_sum(float v) {
    auto ror = [](float x, auto n) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + decltype(n)::value, 0xf, 0xf, true));
    };
    v += ror(v, std::integral_constant<int, 8>{});
    v += ror(v, std::integral_constant<int, 4>{});
    v += ror(v, std::integral_constant<int, 2>{});
    v += ror(v, std::integral_constant<int, 1>{});
    return v;
}

// launch K<false> (bf16) or K<true> (fp16) by the runtime format flag
#define LAUNCH_HF(hf, K, ...)                                   \
    do {                                                        \
        if (hf) SOCCDPT_LAUNCH((K<true>), __VA_ARGS__);     \
        else SOCCDPT_LAUNCH((K<false>), __VA_ARGS__);       \
    } while (0)

// ---------------------------------------------------------------------------------------------
// patch_embed: Conv2d(3, C0, k=4, s=4) + bias, flatten, LayerNorm(C0)   (timm PatchEmbed;
// call site /root/reference/SOccDPT/model/backbones/swin2.py:25-27).  One wave per token.
// x NCHW f32 [B,3,S,S] -> xf [M,C0] f32 residual stream, xb [M,C0] bf16 GEMM operand.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ g,
                                                           const float* __restrict__ beta, float* __restrict__ xf,
                                                           bf16_t* __restrict__ xb, int B, int S, int C0, int x3) {
    // lane owns output channels `lane` and `lane + 64`; their 2 x 48 weights live in registers for the whole kernel,
    // the 48 patch values are broadcast lane -> SGPR with v_readlane (no LDS, no shuffles)
    const int G = S / 4, lane = threadIdx.x & 63;
    const bool a0 = lane < C0, a1 = lane + 64 < C0;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 wv[48];  // (channel lane, channel lane + 64) pairs: one v_pk_fma_f32 per patch value
#pragma unroll
    for (int k = 0; k < 48; ++k) {  // w is the prepared transpose [48][128] (zero-padded): coalesced
        wv[k] = f32x2{w[k * 128 + lane], w[k * 128 + 64 + lane]};
    }
    const float b0 = a0 ? bias[lane] : 0.f, b1 = a1 ? bias[lane + 64] : 0.f;
    const float g0 = a0 ? g[lane] : 0.f, g1 = a1 ? g[lane + 64] : 0.f;
    const float e0 = a0 ? beta[lane] : 0.f, e1 = a1 ? beta[lane + 64] : 0.f;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    // A wave walks STRIPS of 8 horizontally adjacent tokens: the strip's 12 (channel, ky) rows are 12 x 128 contiguous bytes of x,
    // fetched with 6 fully coalesced loads (one token at a time touched 12 lines for 16 bytes each, and the 8 tokens sharing a line
    // sat in 8 different waves).  Element (row, col) of the strip sits in register (row*32+col)/64 of lane (row*32+col)%64.
    const int strips_per_row = G / 8, nstrips = B * G * strips_per_row;
    auto load_strip = [&](int sidx, float (&r)[6]) {
        if (sidx >= nstrips) return;
        const int b = sidx / (G * strips_per_row), rem = sidx - b * G * strips_per_row, py = rem / strips_per_row, px0 = (rem - py * strips_per_row) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = i * 64 + lane, row = idx >> 5, col = idx & 31;  // row = c*4 + ky
            r[i] = x[((size_t)(b * 3 + (row >> 2)) * S + py * 4 + (row & 3)) * S + px0 * 4 + col];
        }
    };
    float cur[6], nxt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    load_strip(wave_global, nxt);
    for (int sidx = wave_global; sidx < nstrips; sidx += nwaves) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cur[i] = nxt[i];
        load_strip(sidx + nwaves, nxt);  // prefetch the next strip under this strip's FMAs
        const int b = sidx / (G * strips_per_row), rem = sidx - b * G * strips_per_row, py = rem / strips_per_row, px0 = (rem - py * strips_per_row) * 8;
        const int tok0 = (b * G + py) * G + px0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x2 oa = f32x2{b0, b1}, ob = f32x2{0.f, 0.f};  // two accumulator chains halve the dependent-FMA latency
#pragma unroll
            for (int k = 0; k < 48; k += 2) {
                const int ia = (k >> 2) * 32 + t * 4 + (k & 3), ib = ia + 1;  // k = (c*4 + ky)*4 + kx; kx and kx+1 share the row
                const float va = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[ia >> 6]), ia & 63));
                const float vb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur[ib >> 6]), ib & 63));
                oa = __builtin_elementwise_fma(f32x2{va, va}, wv[k], oa);
                ob = __builtin_elementwise_fma(f32x2{vb, vb}, wv[k + 1], ob);
            }
            const float o0 = oa[0] + ob[0], o1 = oa[1] + ob[1];
            const int tok = tok0 + t;
            const float mean = wave_sum((a0 ? o0 : 0.f) + (a1 ? o1 : 0.f)) / (float)C0;
            const float d0 = a0 ? o0 - mean : 0.f, d1 = a1 ? o1 - mean : 0.f;
            const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / (float)C0 + 1e-5f);
            if (a0) {
                const float y = d0 * rstd * g0 + e0;
                xf[(size_t)tok * C0 + lane] = y;
                if (xb) { if (x3) x3_store1(xb, (size_t)tok * C0 + lane, y); else xb[(size_t)tok * C0 + lane] = f2h<F16>(y); }
            }
            if (a1) {
                const float y = d1 * rstd * g1 + e1;
                xf[(size_t)tok * C0 + lane + 64] = y;
                if (xb) { if (x3) x3_store1(xb, (size_t)tok * C0 + lane + 64, y); else xb[(size_t)tok * C0 + lane + 64] = f2h<F16>(y); }
            }
        }
    }
}

int launch_patch_embed(const float* x, const float* w, const float* bias, const float* g, const float* beta, float* xf, bf16_t* xb,
                       int hf, int B, int S, int C0, hipStream_t st, std::string& err) {
    if (C0 > 128) { err = "patch_embed: C0 > 128"; return 1; }
    if (S % 32) { err = "patch_embed: image size must be a multiple of 32 (strips of 8 patches)"; return 1; }
    const int nstrips = B * (S / 4) * (S / 32);
    int blocks = (nstrips + 3) / 4;  // one 8-token strip per wave and pass; the FMA chain is latency-bound, so favour waves per SIMD
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    LAUNCH_HF(hf == 1, patch_embed_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, g, beta, xf, xb, B, S, C0, hf == 3 ? 1 : 0);   // hf: 0 bf16, 1 fp16, 3 x3 (half16.h)
    return check_launch("patch_embed", err);
}

// ---------------------------------------------------------------------------------------------
// Swin-V2 residual post-norm: x = x + LN(y) (timm SwinTransformerV2Block; HF modeling_swinv2.py:697-703),
// or x = LN(y) after PatchMerging.reduction.  One wave per token row, C <= 1024.
// Writes the f32 residual stream, its bf16 copy (next GEMM operand) and, for hooked blocks, the
// zero-haloed NHWC bf16 feature map the decoder's 3x3 reassemble conv reads
// (/root/reference/SOccDPT/model/backbones/swin_common.py:38-52 does this as Transpose+Unflatten).
// ---------------------------------------------------------------------------------------------
template <int VPL, bool F16>  // values per lane = ceil(C / 64)
__global__ __launch_bounds__(256) void ln_residual_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                           const float* __restrict__ beta, float* __restrict__ xf,
                                                           bf16_t* __restrict__ xb, bf16_t* __restrict__ halo, float* __restrict__ halo_f32, int M,
                                                           int C, int residual, int res /*spatial size for halo / merge*/, int merge, int x3, int x3h,
                                                           const float* __restrict__ row_scale, int rows_per_scale) {
    ln_residual_row<VPL, F16>(y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale,
                              (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), (int)(threadIdx.x & 63));
}

template <int V4, bool F16>  // float4 groups per lane = C / 256
__global__ __launch_bounds__(256) void ln_residual_v4_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                              const float* __restrict__ beta, float* __restrict__ xf,
                                                              bf16_t* __restrict__ xb, bf16_t* __restrict__ halo, float* __restrict__ halo_f32, int M,
                                                              int C, int residual, int res, int merge, int x3, int x3h,
                                                              const float* __restrict__ row_scale, int rows_per_scale) {
    ln_residual_v4_row<V4, F16>(y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale,
                                (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), (int)(threadIdx.x & 63));
}

int launch_ln_residual(const float* y, const float* g, const float* beta, float* xf, bf16_t* xb, bf16_t* halo, float* halo_f32, int hf, int M,
                       int C, int residual, int res, int merge, hipStream_t st, std::string& err, const float* row_scale, int rows_per_scale, int hf_halo) {
    if (merge && (res <= 0 || (res & 1) || M % (res * res) != 0)) { err = "ln_residual: merged operand layout needs an even token grid"; return 1; }
    if (hf_halo < 0) hf_halo = hf;
    const int x3 = hf == 3 ? 1 : 0, x3h = hf_halo == 3 ? 1 : 0;   // hf / hf_halo: 0 bf16, 1 fp16, 3 x3 (half16.h)
    if ((x3 || x3h) && C % 16) { err = "ln_residual: x3 rows are multiples of 16 elements"; return 1; }
    if ((hf == 0) != (hf_halo == 0) && xb && halo) { err = "ln_residual: bf16 and fp16 / x3 outputs cannot be mixed in one launch"; return 1; }
    hf = (hf == 1 || hf_halo == 1 || x3 || x3h);   // 16-bit outputs beside an x3 one are IEEE fp16 (SOCCDPT_PREC_MIXED)
    const int vpl = (C + 63) / 64;
    dim3 grid((M + 3) / 4), block(256);
    if (C % 256 == 0 && C <= 1024) {
#define LN4_CASE(V)                                                                                                                       \
    do {                                                                                                                                  \
        if (hf) SOCCDPT_LAUNCH((ln_residual_v4_kernel<V, true>), grid, block, 0, st, y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale);  \
        else SOCCDPT_LAUNCH((ln_residual_v4_kernel<V, false>), grid, block, 0, st, y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale);    \
    } while (0)
        switch (C / 256) { case 1: LN4_CASE(1); break; case 2: LN4_CASE(2); break; case 3: LN4_CASE(3); break; default: LN4_CASE(4); break; }
#undef LN4_CASE
        return check_launch("ln_residual", err);
    }
#define LN_CASE(V)                                                                                                                    \
    do {                                                                                                                              \
        if (hf) SOCCDPT_LAUNCH((ln_residual_kernel<V, true>), grid, block, 0, st, y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale);  \
        else SOCCDPT_LAUNCH((ln_residual_kernel<V, false>), grid, block, 0, st, y, g, beta, xf, xb, halo, halo_f32, M, C, residual, res, merge, x3, x3h, row_scale, rows_per_scale);    \
    } while (0)
    if (vpl <= 2) LN_CASE(2);
    else if (vpl <= 4) LN_CASE(4);
    else if (vpl <= 8) LN_CASE(8);
    else if (vpl <= 12) LN_CASE(12);
    else if (vpl <= 16) LN_CASE(16);
    else { err = "ln_residual: C > 1024"; return 1; }
#undef LN_CASE
    return check_launch("ln_residual", err);
}

// ---------------------------------------------------------------------------------------------
// PatchMerging gather (timm PatchMerging; HF modeling_swinv2.py:333-352): [B,R,R,C] bf16 ->
// [B,R/2,R/2,4C] with channel blocks (0,0),(1,0),(0,1),(1,1).  16-byte chunks.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_gather_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int B, int R, int cpt /*16-byte chunks per token*/) {
    const size_t total = (size_t)B * (R / 2) * (R / 2) * 4 * cpt;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpt);
        size_t r = i / cpt;
        const int blk = (int)(r & 3);
        r >>= 2;
        const int R2 = R / 2;
        const int ox = (int)(r % R2);
        r /= R2;
        const int oy = (int)(r % R2);
        const int b = (int)(r / R2);
        const int dy = blk & 1, dx = blk >> 1;  // block order: (0,0),(1,0),(0,1),(1,1) = (dy,dx)
        out[i] = in[((size_t)(b * R + 2 * oy + dy) * R + 2 * ox + dx) * cpt + ch];
    }
}

int launch_merge_gather(const void* in, void* out, int B, int R, int C, int elem_bytes, hipStream_t st, std::string& err) {
    const int cpt = C * elem_bytes / 16;
    const size_t total = (size_t)B * (R / 2) * (R / 2) * 4 * cpt;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(merge_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint4*>(in), static_cast<uint4*>(out), B, R, cpt);
    return check_launch("merge_gather", err);
}

// ---------------------------------------------------------------------------------------------
// Bilinear resampling, align_corners=True, NHWC (F.interpolate in
// /root/reference/SOccDPT/model/blocks.py:488-493 and Interpolate model/blocks.py:239-273).
// src = dst * (in-1)/(out-1); i0 = int(src); l1 = src - i0.  One thread = 4 channels of one
// output pixel.  TIn: float or bf16_t.  Output: f32 plain [M][C] and/or bf16 plain / halo.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <bool F16>
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(h_lo<F16>(u.x), h_hi<F16>(u.x), h_lo<F16>(u.y), h_hi<F16>(u.y));
}

template <typename TIn, bool F16>
__global__ __launch_bounds__(256) void bilinear_kernel(const TIn* __restrict__ in, float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16,
                                                        float* __restrict__ out_f32_halo, int out_halo, int B, int h, int w, int H, int W, int C, int x3) {
    const int c4 = C / 4;
    const size_t total = (size_t)B * H * W * c4;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4) * 4;
        size_t r = i / c4;
        const int ox = (int)(r % W);
        r /= W;
        const int oy = (int)(r % H);
        const int b = (int)(r / H);
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const TIn* base = in + (size_t)b * h * w * C + cc;
        const float4 v00 = load4<F16>(base + ((size_t)y0 * w + x0) * C);
        const float4 v01 = load4<F16>(base + ((size_t)y0 * w + x1) * C);
        const float4 v10 = load4<F16>(base + ((size_t)y1 * w + x0) * C);
        const float4 v11 = load4<F16>(base + ((size_t)y1 * w + x1) * C);
        float4 o;
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        const size_t pix = ((size_t)b * H + oy) * W + ox;
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + pix * C + cc) = o;
        if (out_f32_halo) *reinterpret_cast<float4*>(out_f32_halo + ((size_t)(b * (H + 2) + oy + 1) * (W + 2) + ox + 1) * C + cc) = o;
        if (out_bf16) {
            const size_t off = out_halo ? (((size_t)(b * (H + 2) + oy + 1) * (W + 2) + ox + 1) * C + cc) : (pix * C + cc);
            if (x3) {
                x3_store4(out_bf16, off, o.x, o.y, o.z, o.w);
            } else {
                uint2 p;
                p.x = pack_h2<F16>(o.x, o.y);
                p.y = pack_h2<F16>(o.z, o.w);
                *reinterpret_cast<uint2*>(out_bf16 + off) = p;
            }
        }
    }
}

// Specialisation for the one big launch of the forward (path_1: f32 [B,64,64,256] -> 16-bit halo [B,130,130,256]): 8 channels per
// thread = two 16-byte loads per tap and ONE 16-byte store (the generic kernel stores 8 bytes per thread).
template <bool F16>
__global__ __launch_bounds__(256) void bilinear_f32_to_halo8_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int B, int h, int w,
                                                                     int H, int W, int C) {
    const int c8 = C / 8;
    const size_t total = (size_t)B * H * W * c8;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8) * 8;
        size_t r = i / c8;
        const int ox = (int)(r % W);
        r /= W;
        const int oy = (int)(r % H);
        const int b = (int)(r / H);
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (size_t)b * h * w * C + cc;
        uint32_t pk[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * w + x0) * C + 4 * half);
            const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * w + x1) * C + 4 * half);
            const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * w + x0) * C + 4 * half);
            const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * w + x1) * C + 4 * half);
            // same expression as bilinear_kernel: identical results
            const float ox_ = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            const float oy_ = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            const float oz_ = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            const float ow_ = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
            pk[2 * half] = pack_h2<F16>(ox_, oy_);
            pk[2 * half + 1] = pack_h2<F16>(oz_, ow_);
        }
        *reinterpret_cast<uint4*>(out + ((size_t)(b * (H + 2) + oy + 1) * (W + 2) + ox + 1) * C + cc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

// The same launch in the F16X3 mode: f32 [B,h,w,C] -> x3 halo image, one 8-element unit (two 16-byte stores) per thread
__global__ __launch_bounds__(256) void bilinear_f32_to_halo8_x3_kernel(const float* __restrict__ in, void* __restrict__ out, int B, int h, int w, int H, int W, int C) {
    const int c8 = C / 8;
    const size_t total = (size_t)B * H * W * c8;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8) * 8;
        size_t r = i / c8;
        const int ox = (int)(r % W);
        r /= W;
        const int oy = (int)(r % H);
        const int b = (int)(r / H);
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (size_t)b * h * w * C + cc;
        float o[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * w + x0) * C + 4 * half);
            const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)y0 * w + x1) * C + 4 * half);
            const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * w + x0) * C + 4 * half);
            const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)y1 * w + x1) * C + 4 * half);
            // same expression as bilinear_kernel: identical results
            o[4 * half] = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            o[4 * half + 1] = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            o[4 * half + 2] = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            o[4 * half + 3] = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        }
        x3_store8(out, ((size_t)(b * (H + 2) + oy + 1) * (W + 2) + ox + 1) * C + cc, o);
    }
}

int launch_bilinear(const void* in, int in_is_bf16, float* out_f32, bf16_t* out_bf16, float* out_f32_halo, int out_halo, int hf, int B, int h,
                    int w, int H, int W, int C, hipStream_t st, std::string& err) {
    if (C % 4) { err = "bilinear: C % 4 != 0"; return 1; }
    const int x3 = hf == 3 ? 1 : 0;   // hf: 0 bf16, 1 fp16, 3 x3 output in out_bf16 (half16.h)
    if (x3 && (C % 16 || in_is_bf16)) { err = "bilinear: x3 output needs f32 input and C % 16 == 0"; return 1; }
    hf = hf == 1;
    if (x3 && out_bf16 && out_halo && !out_f32 && !out_f32_halo) {
        const size_t total8 = (size_t)B * H * W * (C / 8);
        size_t blocks8 = (total8 + 255) / 256;
        if (blocks8 > 8192) blocks8 = 8192;
        SOCCDPT_LAUNCH(bilinear_f32_to_halo8_x3_kernel, dim3((unsigned)blocks8), dim3(256), 0, st, (const float*)in, static_cast<void*>(out_bf16), B, h, w, H, W, C);
        return check_launch("bilinear", err);
    }
    if (!x3 && !in_is_bf16 && out_bf16 && out_halo && !out_f32 && !out_f32_halo && C % 8 == 0) {
        const size_t total8 = (size_t)B * H * W * (C / 8);
        size_t blocks8 = (total8 + 255) / 256;
        if (blocks8 > 8192) blocks8 = 8192;
        LAUNCH_HF(hf, bilinear_f32_to_halo8_kernel, dim3((unsigned)blocks8), dim3(256), 0, st, (const float*)in, out_bf16, B, h, w, H, W, C);
        return check_launch("bilinear", err);
    }
    const size_t total = (size_t)B * H * W * (C / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
#define BL_ARGS(T) dim3((unsigned)blocks), dim3(256), 0, st, (const T*)in, out_f32, out_bf16, out_f32_halo, out_halo, B, h, w, H, W, C, x3
    if (in_is_bf16 && hf) SOCCDPT_LAUNCH((bilinear_kernel<bf16_t, true>), BL_ARGS(bf16_t));
    else if (in_is_bf16) SOCCDPT_LAUNCH((bilinear_kernel<bf16_t, false>), BL_ARGS(bf16_t));
    else if (hf) SOCCDPT_LAUNCH((bilinear_kernel<float, true>), BL_ARGS(float));
    else SOCCDPT_LAUNCH((bilinear_kernel<float, false>), BL_ARGS(float));
#undef BL_ARGS
    return check_launch("bilinear", err);
}

// ---------------------------------------------------------------------------------------------
// seg head tail (/root/reference/SOccDPT/model/SOccDPT.py:671-673):
//   (a) Conv2d(256, 3, k=1) + bias on the bf16 [M,256] feature map -> f32 [M,3]
//       16 lanes per pixel, each 16 channels (2 x 16-byte loads), 4-step shuffle reduction.
//   (b) bilinear x2 (align_corners=True) + Sigmoid / ScaledTanh -> NCHW f32 [B,3,2h,2w]
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void conv1x1_c3_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w /*[3][256]*/,
                                                          const float* __restrict__ bias, float* __restrict__ out, int M) {
    // 16 lanes per pixel, 16 channels per lane; a lane's 3 x 16 weights stay in registers over a grid-stride loop of pixels
    const int sub = threadIdx.x & 15;
    float w0[16], w1[16], w2[16];
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(w + sub * 16 + k), b = *reinterpret_cast<const float4*>(w + 256 + sub * 16 + k),
                     c = *reinterpret_cast<const float4*>(w + 512 + sub * 16 + k);
        w0[k] = a.x; w0[k + 1] = a.y; w0[k + 2] = a.z; w0[k + 3] = a.w;
        w1[k] = b.x; w1[k + 1] = b.y; w1[k + 2] = b.z; w1[k + 3] = b.w;
        w2[k] = c.x; w2[k + 1] = c.y; w2[k + 2] = c.z; w2[k + 3] = c.w;
    }
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
    const size_t stride = ((size_t)gridDim.x * blockDim.x) >> 4;
    // M % 16 == 0 pixels per 256-thread pass keeps every 16-lane group entirely inside or outside the loop
    for (size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; pix < (size_t)M; pix += stride) {
        const bf16_t* p = in + pix * 256 + sub * 16;
        const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
        const uint32_t u[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float lo = h_lo<F16>(u[k]), hi = h_hi<F16>(u[k]);
            s0 += lo * w0[2 * k] + hi * w0[2 * k + 1];
            s1 += lo * w1[2 * k] + hi * w1[2 * k + 1];
            s2 += lo * w2[2 * k] + hi * w2[2 * k + 1];
        }
        s0 = row_sum(s0); s1 = row_sum(s1); s2 = row_sum(s2);
        if (sub == 0) {
            out[pix * 3 + 0] = s0 + b0;
            out[pix * 3 + 1] = s1 + b1;
            out[pix * 3 + 2] = s2 + b2;
        }
    }
}

__global__ __launch_bounds__(256) void conv1x1_c3_f32_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int M) {
    const int sub = threadIdx.x & 15;
    const size_t pix = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (pix >= (size_t)M) return;
    const float* p = in + pix * 256 + sub * 16;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + k);
        const int c = sub * 16 + k;
        s0 += v.x * w[c] + v.y * w[c + 1] + v.z * w[c + 2] + v.w * w[c + 3];
        s1 += v.x * w[256 + c] + v.y * w[256 + c + 1] + v.z * w[256 + c + 2] + v.w * w[256 + c + 3];
        s2 += v.x * w[512 + c] + v.y * w[512 + c + 1] + v.z * w[512 + c + 2] + v.w * w[512 + c + 3];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (sub == 0) {
        out[pix * 3 + 0] = s0 + bias[0];
        out[pix * 3 + 1] = s1 + bias[1];
        out[pix * 3 + 2] = s2 + bias[2];
    }
}

__global__ __launch_bounds__(256) void seg_up_act_kernel(const float* __restrict__ in /*[B,h,w,3]*/, float* __restrict__ out /*[B,3,H,W]*/, int B,
                                                          int h, int w, int sigmoid) {
    const int H = 2 * h, W = 2 * w;
    const size_t total = (size_t)B * H * W;
    const float sy = (float)(h - 1) / (float)(H - 1), sx = (float)(w - 1) / (float)(W - 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W);
        size_t r = i / W;
        const int oy = (int)(r % H), b = (int)(r / H);
        const float fy = sy * (float)oy, fx = sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (size_t)b * h * w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = hy * (hx * base[((size_t)y0 * w + x0) * 3 + c] + lx * base[((size_t)y0 * w + x1) * 3 + c]) +
                            ly * (hx * base[((size_t)y1 * w + x0) * 3 + c] + lx * base[((size_t)y1 * w + x1) * 3 + c]);
            const float a = sigmoid ? 1.f / (1.f + expf(-v)) : 0.5f * tanhf(v) + 0.5f;
            out[((size_t)(b * 3 + c) * H + oy) * W + ox] = a;
        }
    }
}

// Finish of the classifier fused into the seg head's convolution (igemm D3 epilogue): logits[m][c] = bias[c] + sum over the n-tiles (in tile order) of
// part[t][m][c]; then the same up-sampling + activation as below.
__global__ __launch_bounds__(256) void seg_logits_finish_kernel(const float* __restrict__ part /*[T][M][4]*/, const float* __restrict__ bias, float* __restrict__ out /*[M][3]*/,
                                                                 size_t M, int T) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        float4 a = *reinterpret_cast<const float4*>(part + m * 4);
        for (int t = 1; t < T; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)t * M + m) * 4);
            a.x += v.x; a.y += v.y; a.z += v.z;
        }
        out[m * 3 + 0] = a.x + bias[0];
        out[m * 3 + 1] = a.y + bias[1];
        out[m * 3 + 2] = a.z + bias[2];
    }
}

int launch_seg_tail_parts(const float* part, int ntiles, const float* bias, float* tmp, float* seg, int B, int h, int wd, int sigmoid, hipStream_t st, std::string& err) {
    const size_t M = (size_t)B * h * wd;
    size_t blocks = (M + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(seg_logits_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, bias, tmp, M, ntiles);
    if (check_launch("seg_logits_finish", err)) return 1;
    const size_t total = (size_t)B * 4 * h * wd;
    blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(seg_up_act_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tmp, seg, B, h, wd, sigmoid);
    return check_launch("seg_up_act", err);
}

int launch_seg_tail(const void* feat, int feat_is_f32, int hf, const float* w, const float* bias, float* tmp, float* seg, int B, int h, int wd,
                    int sigmoid, hipStream_t st, std::string& err) {
    const int M = B * h * wd;
    const dim3 grid((unsigned)(((size_t)M * 16 + 255) / 256));
    unsigned gl = grid.x > 2048 ? 2048 : grid.x;   // grid-stride (16-bit path): weights are loaded once per thread
    if (feat_is_f32) SOCCDPT_LAUNCH(conv1x1_c3_f32_kernel, grid, dim3(256), 0, st, static_cast<const float*>(feat), w, bias, tmp, M);
    else LAUNCH_HF(hf, conv1x1_c3_kernel, dim3(gl), dim3(256), 0, st, static_cast<const bf16_t*>(feat), w, bias, tmp, M);
    if (check_launch("conv1x1_c3", err)) return 1;
    const size_t total = (size_t)B * 4 * h * wd;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(seg_up_act_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tmp, seg, B, h, wd, sigmoid);
    return check_launch("seg_up_act", err);
}

// ---------------------------------------------------------------------------------------------
// One-time weight preparation
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ void cvt_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f2h<F16>(in[i]);
}
__global__ void cvt_f16_ieee_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {   // no clamp: overflow -> inf (training amp, half16.h)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f2h_ieee(in[i]);
}
__global__ void cvt_x3_kernel(const float* __restrict__ in, void* __restrict__ out, size_t n) {   // n % 4 == 0
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(in + i);
        x3_store4(out, i, v.x, v.y, v.z, v.w);
    }
}
// [Cout][Cin][3][3] f32 -> [Cout][3][3][Cin] bf16, optionally scaled per Cout (BatchNorm fold)
template <bool F16>
__global__ void conv_w_kernel(const float* __restrict__ in, const float* __restrict__ scale, bf16_t* __restrict__ out, int Cout, int Cin) {
    const size_t n = (size_t)Cout * Cin * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        size_t r = i / Cin;
        const int tap = (int)(r % 9), co = (int)(r / 9);
        float v = in[((size_t)co * Cin + ci) * 9 + tap];
        if (scale) v *= scale[co];
        out[i] = f2h<F16>(v);
    }
}
template <bool X3>
__global__ void conv_w_f32_kernel(const float* __restrict__ in, const float* __restrict__ scale, float* __restrict__ out, int Cout, int Cin) {
    const size_t n = (size_t)Cout * Cin * 9;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        size_t r = i / Cin;
        const int tap = (int)(r % 9), co = (int)(r / 9);
        float v = in[((size_t)co * Cin + ci) * 9 + tap];
        if (scale) v *= scale[co];
        if constexpr (X3) x3_store1(out, i, v); else out[i] = v;
    }
}
// patch-embed weight [C0][48] -> [48][128] zero-padded (coalesced per-lane loads in patch_embed_kernel)
__global__ void patch_w_kernel(const float* __restrict__ w, float* __restrict__ out, int C0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 48 * 128) {
        const int k = i / 128, n = i % 128;
        out[i] = n < C0 ? w[n * 48 + k] : 0.f;
    }
}
int launch_patch_w(const float* w, float* out, int C0, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(patch_w_kernel, dim3(24), dim3(256), 0, st, w, out, C0);
    return check_launch("patch_w", err);
}
// BatchNorm2d eval fold (model/SOccDPT.py:668): scale = g / sqrt(var + eps), shift = b - mean * scale
__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var, float* scale, float* shift, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        const float s = g[i] / sqrtf(var[i] + 1e-5f);
        scale[i] = s;
        shift[i] = b[i] - mean[i] * s;
    }
}
// qkv bias = cat(q_bias, 0, v_bias)  (timm WindowAttention.forward)
__global__ void qkv_bias_kernel(const float* q, const float* v, float* out, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * C) out[i] = i < C ? q[i] : (i < 2 * C ? 0.f : v[i - 2 * C]);
}
// logit scale: exp(min(logit_scale, ln 100))
__global__ void logit_scale_kernel(const float* ls, float* out, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H) out[i] = expf(fminf(ls[i], 4.605170185988092f));
}
// continuous position bias table: 16*sigmoid(cpb_mlp(coords))  [(2ws-1)^2][heads]
// One thread per (entry, head), 512 hidden units in order (the addition order every earlier round used: same bits).  Round 6: the MLP's weights -- w0 [512][2], b0 [512] and
// w2 [H][512], read by every thread -- are staged in LDS first: the serial loop's loads were L2 round trips (29 us per launch; the training step rebuilds the table of all 12
// blocks every forward: 0.35 ms of a 16 ms step).
__global__ __launch_bounds__(256) void cpb_table_kernel(const float* __restrict__ w0 /*[512][2]*/, const float* __restrict__ b0, const float* __restrict__ w2 /*[H][512]*/,
                                                        float* __restrict__ table, int ws, int pws, int H) {
    extern __shared__ float cpb_s[];
    float* s_w0 = cpb_s;            // 1024
    float* s_b0 = cpb_s + 1024;     // 512
    float* s_w2 = cpb_s + 1536;     // H * 512
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_w0[i] = w0[i];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) s_b0[i] = b0[i];
    for (int i = threadIdx.x; i < H * 512; i += blockDim.x) s_w2[i] = w2[i];
    __syncthreads();
    const int T = 2 * ws - 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * T * H) return;
    const int hd = i % H, e = i / H, dy = e / T - (ws - 1), dx = e % T - (ws - 1);
    const float denom = (float)((pws > 0 ? pws : ws) - 1);
    float cy = (float)dy / denom * 8.f, cx = (float)dx / denom * 8.f;
    cy = (cy > 0.f ? 1.f : (cy < 0.f ? -1.f : 0.f)) * log2f(fabsf(cy) + 1.f) / 3.f;
    cx = (cx > 0.f ? 1.f : (cx < 0.f ? -1.f : 0.f)) * log2f(fabsf(cx) + 1.f) / 3.f;
    float s = 0.f;
    const float* w2h = s_w2 + hd * 512;
#pragma unroll 8
    for (int k = 0; k < 512; ++k) {
        const float hdn = fmaxf(s_w0[2 * k] * cy + s_w0[2 * k + 1] * cx + s_b0[k], 0.f);
        s += hdn * w2h[k];
    }
    table[i] = 16.f / (1.f + expf(-s));
}

int launch_cvt_bf16(const float* in, bf16_t* out, size_t n, int hf, hipStream_t st, std::string& err) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (hf == 3) {   // x3 (half16.h): `out` holds 4 bytes per element
        if (n % 16) { err = "cvt: x3 tensors are multiples of 16 elements"; return 1; }
        SOCCDPT_LAUNCH(cvt_x3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, static_cast<void*>(out), n);
        return check_launch("cvt_x3", err);
    }
    if (hf == 5) {   // IEEE fp16 without saturation: the scaled operands of the training step's fp16 amp mode
        SOCCDPT_LAUNCH(cvt_f16_ieee_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
        return check_launch("cvt_f16_ieee", err);
    }
    LAUNCH_HF(hf, cvt_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
    return check_launch("cvt_bf16", err);
}
int launch_conv_w(const float* in, const float* scale, void* out, int out_is_f32, int hf, int Cout, int Cin, hipStream_t st, std::string& err) {
    size_t n = (size_t)Cout * Cin * 9, blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (hf == 3) SOCCDPT_LAUNCH(conv_w_f32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, in, scale, static_cast<float*>(out), Cout, Cin);   // x3, 4 bytes per element
    else if (out_is_f32) SOCCDPT_LAUNCH(conv_w_f32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, in, scale, static_cast<float*>(out), Cout, Cin);
    else LAUNCH_HF(hf, conv_w_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, scale, static_cast<bf16_t*>(out), Cout, Cin);
    return check_launch("conv_w", err);
}
int launch_bn_fold(const float* g, const float* b, const float* mean, const float* var, float* scale, float* shift, int C, hipStream_t st,
                   std::string& err) {
    SOCCDPT_LAUNCH(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, st, g, b, mean, var, scale, shift, C);
    return check_launch("bn_fold", err);
}
int launch_qkv_bias(const float* q, const float* v, float* out, int C, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(qkv_bias_kernel, dim3((3 * C + 255) / 256), dim3(256), 0, st, q, v, out, C);
    return check_launch("qkv_bias", err);
}
int launch_logit_scale(const float* ls, float* out, int H, hipStream_t st, std::string& err) {
    SOCCDPT_LAUNCH(logit_scale_kernel, dim3(1), dim3(64), 0, st, ls, out, H);
    return check_launch("logit_scale", err);
}
int launch_cpb_table(const float* w0, const float* b0, const float* w2, float* table, int ws, int pws, int H, hipStream_t st,
                     std::string& err) {
    const int n = (2 * ws - 1) * (2 * ws - 1) * H;
    const size_t lds = (size_t)(1536 + H * 512) * sizeof(float);   // 8 - 54 KB for 1 - 24 heads... (32 heads: 70 KB)
    static PerDeviceOnce attr;
    if (attr.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cpb_table_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { err = "cpb_table: hipFuncSetAttribute failed"; return 1; }
        attr.done();
    }
    if (lds > 160 * 1024) { err = "cpb_table: too many heads for the LDS-resident MLP"; return 1; }
    SOCCDPT_LAUNCH(cpb_table_kernel, dim3((n + 255) / 256), dim3(256), lds, st, w0, b0, w2, table, ws, pws, H);
    return check_launch("cpb_table", err);
}

}  // namespace soccdpt
