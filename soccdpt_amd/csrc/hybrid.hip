// Non-GEMM kernels of the ViT-hybrid encoder (dpt_hybrid_384, BASELINE configs[2]) for gfx950.
//
// They stand in for the timm 0.6.12 pieces that /root/reference/SOccDPT/model/backbones/vit.py:244-258 creates and
// forward_flex (vit.py:44-85) runs, restated in oracle/soccdpt_ref.py (resnetv2_backbone, vit_block, hybrid_encoder):
//   * StdConv2dSame weight standardisation (folded once per weight load: soccdpt_prepare),
//   * the 7x7 / stride-2 stem convolution as im2col + igemm (Cin = 3 is no MFMA shape),
//   * GroupNormAct(32 groups, eps 1e-5) (+ ReLU, + shortcut, + ReLU), with the statistics produced by the convolution's own
//     epilogue (igemm.hip, ST instantiation), and the stem's GroupNorm + ReLU + MaxPool2dSame(3, 2) fused,
//   * class token + position embedding + the first pre-norm LayerNorm, and the pre-norm LayerNorm (eps 1e-6) of every ViT block.
// All of them are HBM-bound elementwise / row kernels: 16-byte accesses, NHWC, no LDS staging needed except the row reductions.
// OUT selects the operand format written for the next GEMM: 0 = bf16, 1 = fp16, 2 = f32, 3 = x3 split fp16 (half16.h; SOCCDPT_PREC_F16X3).
#include "half16.h"
#include "kernels.h"

namespace soccdpt {
namespace {

template <int OUT>
struct OutT { typedef uint16_t type; };
template <>
struct OutT<2> { typedef float type; };
template <>
struct OutT<3> { typedef float type; };   // x3: 4 bytes per element

// store 4 consecutive values as the operand type
template <int OUT>
__device__ __forceinline__ void store4(void* base, size_t idx, float a, float b, float c, float d) {
    if constexpr (OUT == 2) {
        *reinterpret_cast<float4*>(static_cast<float*>(base) + idx) = make_float4(a, b, c, d);
    } else if constexpr (OUT == 3) {
        x3_store4(base, idx, a, b, c, d);
    } else {
        uint2 p;
        p.x = pack_h2<OUT == 1>(a, b);
        p.y = pack_h2<OUT == 1>(c, d);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(base) + idx) = p;
    }
}

// ---- weight standardisation: w [Cout][Cin][k][k] f32 -> out [Cout][Kpad] tap-major ((ky*k + kx)*Cin + ci), zero padded ----
// (w - mean) / sqrt(var + eps) per output channel, biased variance over Cin*k*k: timm StdConv2dSame.forward evaluates exactly this
// through F.batch_norm(training=True) on the [1, Cout, fan_in] view.
template <int OUT>
__global__ __launch_bounds__(256) void ws_conv_w_kernel(const float* __restrict__ w, void* __restrict__ out, int Cin, int kk, int Kpad, float eps) {
    __shared__ double red[2][256];
    const int co = blockIdx.x, tid = threadIdx.x;
    const int fan = Cin * kk;
    const float* src = w + (size_t)co * fan;
    double a = 0.0, q = 0.0;
    for (int i = tid; i < fan; i += 256) { const double v = src[i]; a += v; q += v * v; }
    red[0][tid] = a; red[1][tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    const double mean = red[0][0] / fan;
    double var = red[1][0] / fan - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float fm = (float)mean, fr = (float)(1.0 / sqrt(var + (double)eps));
    typename OutT<OUT>::type* dst = static_cast<typename OutT<OUT>::type*>(out) + (size_t)co * Kpad;
    for (int i = tid; i < Kpad; i += 256) {
        float v = 0.f;
        if (i < fan) {
            const int tap = i / Cin, ci = i - tap * Cin;
            v = (src[(size_t)ci * kk + tap] - fm) * fr;
        }
        if constexpr (OUT == 2) dst[i] = v;
        else if constexpr (OUT == 3) x3_store1(out, (size_t)co * Kpad + i, v);
        else dst[i] = f2h<OUT == 1>(v);
    }
}

// ---- stem im2col: x [B][3][S][S] f32 NCHW -> A [B*Ho*Wo][160], k = (ky*7 + kx)*3 + c, zero beyond 147 ----
// Conv2d(3, 64, 7, stride 2) with TF 'SAME' padding: total pad 5 at S = 384 -> 2 left / top, 3 right / bottom.
template <int OUT>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, void* __restrict__ A, int B, int S, int Ho, int pad_lo) {
    const size_t total = (size_t)B * Ho * Ho * 20;   // 20 chunks of 8 k per row
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ch = (int)(gid % 20);
    const size_t m = gid / 20;
    const int ox = (int)(m % Ho), oy = (int)((m / Ho) % Ho), b = (int)(m / ((size_t)Ho * Ho));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ch * 8 + j;
        const int tap = k / 3, c = k - tap * 3;
        const int ky = tap / 7, kx = tap - ky * 7;
        const int iy = oy * 2 + ky - pad_lo, ix = ox * 2 + kx - pad_lo;
        v[j] = (k < 147 && iy >= 0 && iy < S && ix >= 0 && ix < S) ? x[(((size_t)b * 3 + c) * S + iy) * S + ix] : 0.f;
    }
    store4<OUT>(A, m * 160 + ch * 8, v[0], v[1], v[2], v[3]);
    store4<OUT>(A, m * 160 + ch * 8 + 4, v[4], v[5], v[6], v[7]);
}

// GroupNorm of 4 consecutive channels c..c+3 of sample b: ((x - mean) * rstd) * gamma + beta, the order torch.group_norm uses
struct Gn4 {
    float mean, rstd;
    float4 g, be;
    __device__ __forceinline__ void load(const float* stats, const float* gamma, const float* beta, int b, int c, int C, int cpg) {
        const int G = C / cpg;
        const float2 s = *reinterpret_cast<const float2*>(stats + ((size_t)b * G + c / cpg) * 2);
        mean = s.x; rstd = s.y;
        g = *reinterpret_cast<const float4*>(gamma + c);
        be = *reinterpret_cast<const float4*>(beta + c);
    }
    __device__ __forceinline__ float4 apply(float4 v) const {
        return make_float4((v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y, (v.z - mean) * rstd * g.z + be.z,
                           (v.w - mean) * rstd * g.w + be.w);
    }
};
// cpg == 2: a float4 spans two groups
struct Gn4x2 {
    float2 s0, s1;
    float4 g, be;
    __device__ __forceinline__ void load(const float* stats, const float* gamma, const float* beta, int b, int c, int C) {
        const int G = C / 2;
        const float4 s = *reinterpret_cast<const float4*>(stats + ((size_t)b * G + c / 2) * 2);
        s0 = make_float2(s.x, s.y); s1 = make_float2(s.z, s.w);
        g = *reinterpret_cast<const float4*>(gamma + c);
        be = *reinterpret_cast<const float4*>(beta + c);
    }
    __device__ __forceinline__ float4 apply(float4 v) const {
        return make_float4((v.x - s0.x) * s0.y * g.x + be.x, (v.y - s0.x) * s0.y * g.y + be.y, (v.z - s1.x) * s1.y * g.z + be.z,
                           (v.w - s1.x) * s1.y * g.w + be.w);
    }
};

// ---- GroupNorm apply (+ shortcut) (+ ReLU): one thread = 4 channels of one pixel ----
//   y = GN(raw)                                   (+ GN2(raw2): the projection shortcut of a stage's first block)
//                                                 (+ res:       the identity shortcut, f32 residual stream)
//   y = relu(y) when relu != 0
// out_f32 [M][C] (residual stream), out_op [M][C] and out_halo [B][H+2][W+2][C] (operand copies) are each optional.
template <int OUT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ raw, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ raw2, const float* __restrict__ stats2,
                                                       const float* __restrict__ gamma2, const float* __restrict__ beta2, const float* res /* may alias out_f32 */,
                                                       float* out_f32, void* __restrict__ out_op, void* __restrict__ out_halo, int relu,
                                                       size_t M, int HW, int W, int C, int cpg, int halo_mode) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (gid >= M * c4) return;
    const size_t m = gid / c4;
    const int c = (int)(gid - m * c4) * 4;
    const int b = (int)(m / HW);
    float4 v = *reinterpret_cast<const float4*>(raw + m * C + c);
    if (cpg == 2) {
        Gn4x2 n; n.load(stats, gamma, beta, b, c, C);
        v = n.apply(v);
        if (raw2) {
            Gn4x2 n2; n2.load(stats2, gamma2, beta2, b, c, C);
            const float4 s = n2.apply(*reinterpret_cast<const float4*>(raw2 + m * C + c));
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
    } else {
        Gn4 n; n.load(stats, gamma, beta, b, c, C, cpg);
        v = n.apply(v);
        if (raw2) {
            Gn4 n2; n2.load(stats2, gamma2, beta2, b, c, C, cpg);
            const float4 s = n2.apply(*reinterpret_cast<const float4*>(raw2 + m * C + c));
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
    }
    if (res) {
        const float4 s = *reinterpret_cast<const float4*>(res + m * C + c);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + m * C + c) = v;
    if (out_op) store4<OUT>(out_op, m * C + c, v.x, v.y, v.z, v.w);
    if (out_halo) {
        const int rem = (int)(m - (size_t)b * HW);
        const int y = rem / W, x = rem - y * W, H = HW / W;
        const size_t hidx = (((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c;
        // halo_mode: format of the halo image when it differs from out_op's (SOCCDPT_PREC_MIXED: the hooked stage output feeds the decoder's group)
        if (halo_mode == OUT) store4<OUT>(out_halo, hidx, v.x, v.y, v.z, v.w);
        else if (halo_mode == 3) store4<3>(out_halo, hidx, v.x, v.y, v.z, v.w);
        else if (halo_mode == 2) store4<2>(out_halo, hidx, v.x, v.y, v.z, v.w);
        else if (halo_mode == 1) store4<1>(out_halo, hidx, v.x, v.y, v.z, v.w);
        else store4<0>(out_halo, hidx, v.x, v.y, v.z, v.w);
    }
}

// ---- stem: GroupNorm + ReLU + MaxPool2dSame(3, stride 2) fused.  raw [B][Hi][Hi][C] -> out [B][Ho][Ho][C] operand type.
// 'SAME' pooling pads right / bottom with -inf (timm MaxPool2dSame), i.e. out-of-range taps are simply skipped.
template <int OUT>
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const float* __restrict__ raw, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, void* __restrict__ out, int B, int Hi, int Ho, int C, int cpg) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    const size_t total = (size_t)B * Ho * Ho * c4;
    if (gid >= total) return;
    const size_t m = gid / c4;
    const int c = (int)(gid - m * c4) * 4;
    const int ox = (int)(m % Ho), oy = (int)((m / Ho) % Ho), b = (int)(m / ((size_t)Ho * Ho));
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);   // every candidate is >= 0 after the ReLU and the centre tap always exists
    Gn4 n4; Gn4x2 n2;
    if (cpg == 2) n2.load(stats, gamma, beta, b, c, C); else n4.load(stats, gamma, beta, b, c, C, cpg);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 + ky;
        if (iy >= Hi) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 + kx;
            if (ix >= Hi) continue;
            float4 v = *reinterpret_cast<const float4*>(raw + (((size_t)b * Hi + iy) * Hi + ix) * C + c);
            v = cpg == 2 ? n2.apply(v) : n4.apply(v);
            best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
        }
    }
    store4<OUT>(out, m * C + c, best.x, best.y, best.z, best.w);
}

// ---- LayerNorm rows: one wave per row of C = 768 (12 values per lane as three float4) ----
// TOK != 0: the row is first assembled from the patch projection: row 0 of a sample = cls + pos[0], row t = y[b][t-1] + pos[t]
// (forward_flex, /root/reference/SOccDPT/model/backbones/vit.py:58-77), and written to xf.
template <int OUT, int TOK>
__global__ __launch_bounds__(256) void ln768_kernel(const float* __restrict__ y, const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ xf,
                                                    const float* __restrict__ g, const float* __restrict__ be, void* __restrict__ xb, int rows, int ntok, float eps) {
    constexpr int C = 768;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[3];
    if constexpr (TOK) {
        const int b = row / ntok, t = row - b * ntok;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = k * 256 + lane * 4;
            const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)t * C + c);
            const float4 s = t == 0 ? *reinterpret_cast<const float4*>(cls + c) : *reinterpret_cast<const float4*>(y + ((size_t)b * (ntok - 1) + t - 1) * C + c);
            v[k] = make_float4(s.x + p.x, s.y + p.y, s.z + p.z, s.w + p.w);
            *reinterpret_cast<float4*>(xf + (size_t)row * C + c) = v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = *reinterpret_cast<const float4*>(xf + (size_t)row * C + k * 256 + lane * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = v[k].x - mean, b2 = v[k].y - mean, c2 = v[k].z - mean, d2 = v[k].w - mean;
        q += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / C) + eps);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = k * 256 + lane * 4;
        const float4 gg = *reinterpret_cast<const float4*>(g + c), bb = *reinterpret_cast<const float4*>(be + c);
        store4<OUT>(xb, (size_t)row * C + c, (v[k].x - mean) * rstd * gg.x + bb.x, (v[k].y - mean) * rstd * gg.y + bb.y,
                    (v[k].z - mean) * rstd * gg.z + bb.z, (v[k].w - mean) * rstd * gg.w + bb.w);
    }
}

// position-embedding resize at prepare time (_resize_pos_embed, vit.py:23-41): bilinear, align_corners=False, [1 + g0*g0][C] -> [1 + g*g][C]
__global__ void pos_embed_resize_kernel(const float* __restrict__ pos, float* __restrict__ out, int g0, int g, int C) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(1 + g * g) * C;
    if (gid >= total) return;
    const int c = (int)(gid % C), t = (int)(gid / C);
    if (t == 0) { out[gid] = pos[c]; return; }
    const int oy = (t - 1) / g, ox = (t - 1) % g;
    const float sc = (float)g0 / (float)g;
    float fy = ((float)oy + 0.5f) * sc - 0.5f, fx = ((float)ox + 0.5f) * sc - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < g0 - 1), x1 = x0 + (x0 < g0 - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto P = [&](int yy, int xx) { return pos[(size_t)(1 + yy * g0 + xx) * C + c]; };
    out[gid] = (1.f - ly) * ((1.f - lx) * P(y0, x0) + lx * P(y0, x1)) + ly * ((1.f - lx) * P(y1, x0) + lx * P(y1, x1));
}

}  // namespace

#define DISPATCH_OUT(mode, ...)                                              \
    do {                                                                     \
        if ((mode) == 3) { constexpr int OUT = 3; __VA_ARGS__; }             \
        else if ((mode) == 2) { constexpr int OUT = 2; __VA_ARGS__; }        \
        else if ((mode) == 1) { constexpr int OUT = 1; __VA_ARGS__; }        \
        else { constexpr int OUT = 0; __VA_ARGS__; }                         \
    } while (0)

int launch_ws_conv_w(const float* w, void* out, int out_mode, int Cout, int Cin, int k, int Kpad, float eps, hipStream_t st, std::string& err) {
    if (Kpad < Cin * k * k) { err = "ws_conv_w: Kpad too small"; return 1; }
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(ws_conv_w_kernel<OUT>, dim3((unsigned)Cout), dim3(256), 0, st, w, out, Cin, k * k, Kpad, eps));
    return check_launch("ws_conv_w", err);
}

int launch_stem_im2col(const float* x, void* A, int out_mode, int B, int S, hipStream_t st, std::string& err) {
    if (S % 2) { err = "stem_im2col: odd image size"; return 1; }
    const int Ho = S / 2;
    const int total_pad = (Ho - 1) * 2 + 7 - S;   // TF 'SAME'
    const int pad_lo = (total_pad > 0 ? total_pad : 0) / 2;
    const size_t total = (size_t)B * Ho * Ho * 20;
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(stem_im2col_kernel<OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, A, B, S, Ho, pad_lo));
    return check_launch("stem_im2col", err);
}

int launch_gn_apply(const GnApplyArgs& a, int out_mode, hipStream_t st, std::string& err) {
    if (a.C % 4 || a.cpg < 2 || (a.cpg != 2 && a.cpg % 4) || a.C % a.cpg || a.HW <= 0 || a.M % (size_t)a.HW) { err = "gn_apply: bad geometry"; return 1; }
    if (a.raw2 && a.res) { err = "gn_apply: one shortcut kind at a time"; return 1; }
    const size_t total = a.M * (size_t)(a.C / 4);
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(gn_apply_kernel<OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.raw, a.stats, a.gamma, a.beta, a.raw2,
                                               a.stats2, a.gamma2, a.beta2, a.res, a.out_f32, a.out_op, a.out_halo, a.relu, a.M, a.HW, a.W, a.C, a.cpg, a.halo_mode < 0 ? out_mode : a.halo_mode));
    return check_launch("gn_apply", err);
}

int launch_gn_relu_maxpool(const float* raw, const float* stats, const float* gamma, const float* beta, void* out, int out_mode, int B, int Hi, int C, int cpg,
                           hipStream_t st, std::string& err) {
    const int Ho = (Hi + 1) / 2;
    const size_t total = (size_t)B * Ho * Ho * (C / 4);
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH(gn_relu_maxpool_kernel<OUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw, stats, gamma, beta, out, B, Hi,
                                               Ho, C, cpg));
    return check_launch("gn_relu_maxpool", err);
}

int launch_vit_tokens_ln(const float* y, const float* cls, const float* pos, float* xf, const float* g, const float* be, void* xb, int out_mode, int B, int ntok,
                         int C, float eps, hipStream_t st, std::string& err) {
    if (C != 768) { err = "vit_tokens_ln: C must be 768"; return 1; }
    const int rows = B * ntok;
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((ln768_kernel<OUT, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, y, cls, pos, xf, g, be, xb, rows, ntok, eps));
    return check_launch("vit_tokens_ln", err);
}

int launch_ln_rows(float* xf, const float* g, const float* be, void* xb, int out_mode, int rows, int C, float eps, hipStream_t st, std::string& err) {
    if (C != 768) { err = "ln_rows: C must be 768"; return 1; }
    DISPATCH_OUT(out_mode, SOCCDPT_LAUNCH((ln768_kernel<OUT, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, nullptr, nullptr, nullptr, xf, g, be, xb, rows,
                                               0, eps));
    return check_launch("ln_rows", err);
}

int launch_pos_embed_resize(const float* pos, float* out, int g0, int g, int C, hipStream_t st, std::string& err) {
    const size_t total = (size_t)(1 + g * g) * C;
    SOCCDPT_LAUNCH(pos_embed_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pos, out, g0, g, C);
    return check_launch("pos_embed_resize", err);
}

}  // namespace soccdpt
