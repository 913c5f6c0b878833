// Swin-V2 cosine window attention for gfx950 (one workgroup per (batch, window, head)).
//
// Replaces timm WindowAttention.forward + window_partition / roll / window_reverse
// (call site /root/reference/SOccDPT/model/backbones/swin2.py:25-27; maths SURVEY.md §8a a4-E,
// HF modeling_swinv2.py:389-452,652-695):
//   S = normalize(q) normalize(k)^T * exp(min(logit_scale, ln100)) + 16*sigmoid(CPB) [+ shift mask]
//   O = softmax(S) V
// Design (CDNA4):
//  * the cyclic shift, window partition and window reverse are folded into the load/store row index;
//  * q,k rows are L2-normalised in f32 while staging (4 lanes per token, 16-byte loads), the logit
//    scale is folded into q-hat; Q-hat/K-hat live in LDS as [token][32] bf16 with a 16-byte chunk XOR
//    swizzle (conflict-free ds_read_b128), V is stored transposed [d][token] with an 8-byte row pad;
//  * S^T = K-hat Q-hat^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): a lane then owns one
//    query column, so softmax statistics are lane-local plus one cross-half exchange, and the
//    exponentiated accumulators are directly the B operand of O^T = V^T P^T (no LDS round trip);
//  * the CPB bias is precomputed per weight load in MFMA accumulator order and loaded as the
//    initial accumulator (4 x 16-byte loads per tile, coalesced); the shift mask is arithmetic.
#include <stdlib.h>

#include "attention_body.h"

namespace soccdpt {

template <int WS, bool F16, int QS>
__global__ __launch_bounds__(AttnCfg<WS>::THREADS) void window_attention_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                                 const float* __restrict__ scale, bf16_t* __restrict__ out,
                                                                                 int res, int shift, int heads, int out_x3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    window_attention_body<WS, F16, QS>(qkv, bias_acc, scale, out, res, shift, heads, out_x3, (int)blockIdx.x, (int)threadIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Exact-f32 variant (SOCCDPT_PREC_F32, the parity mode): same decomposition with v_mfma_f32_32x32x2_f32.
// q,k,v are f32 [M][3C]; Q-hat/K-hat rows are padded to 33 floats and V^T rows to N+1 floats (conflict-free
// ds_read_b32); S^T accumulators feed O^T = V^T P^T directly: for accumulator register r the two lane halves
// hold keys 32t + (r&3) + 8(r>>2) + 4h, which is exactly one K=2 MFMA step.
// ---------------------------------------------------------------------------------------------
template <int WS>
struct AttnCfgF32 {
    static constexpr int N = WS * WS;
    static constexpr int WAVES = N / 64;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int KT = N / 32, QB = N / 32;
    static constexpr int QK_STRIDE = 33;     // floats
    static constexpr int VT_STRIDE = N + 1;  // floats
    static constexpr int LDS = (2 * N * QK_STRIDE + 32 * VT_STRIDE) * 4;
};

template <int WS>
__global__ __launch_bounds__(AttnCfgF32<WS>::THREADS) void window_attention_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                                       const float* __restrict__ scale, float* __restrict__ out,
                                                                                       int res, int shift, int heads, int x3) {
    using A = AttnCfgF32<WS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Qs = reinterpret_cast<float*>(smem);
    float* Ks = Qs + A::N * A::QK_STRIDE;
    float* Vt = Ks + A::N * A::QK_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const float hscale = scale[head];
    auto token_row = [&](int p) -> size_t {
        const int r = p / WS, c = p % WS;
        int sy = wy * WS + r + shift, sx = wx * WS + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
#pragma unroll
    for (int it = 0; it < (A::N * 4) / A::THREADS; ++it) {
        const int idx = it * A::THREADS + tid;
        const int p = idx >> 2, c = idx & 3;
        const float* src = qkv + token_row(p) * (size_t)(3 * C) + head * 32 + c * 8;
        float qf[8], kf[8], vf[8];
        *reinterpret_cast<float4*>(qf) = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(qf + 4) = *reinterpret_cast<const float4*>(src + 4);
        *reinterpret_cast<float4*>(kf) = *reinterpret_cast<const float4*>(src + C);
        *reinterpret_cast<float4*>(kf + 4) = *reinterpret_cast<const float4*>(src + C + 4);
        *reinterpret_cast<float4*>(vf) = *reinterpret_cast<const float4*>(src + 2 * C);
        *reinterpret_cast<float4*>(vf + 4) = *reinterpret_cast<const float4*>(src + 2 * C + 4);
        float qs = 0.f, ks = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qs += qf[j] * qf[j];
            ks += kf[j] * kf[j];
        }
        qs += __shfl_xor(qs, 1);
        qs += __shfl_xor(qs, 2);
        ks += __shfl_xor(ks, 1);
        ks += __shfl_xor(ks, 2);
        const float qi = hscale / fmaxf(sqrtf(qs), 1e-12f);
        const float ki = 1.0f / fmaxf(sqrtf(ks), 1e-12f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Qs[p * A::QK_STRIDE + c * 8 + j] = qf[j] * qi;
            Ks[p * A::QK_STRIDE + c * 8 + j] = kf[j] * ki;
            Vt[(c * 8 + j) * A::VT_STRIDE + p] = vf[j];
        }
    }
    __syncthreads();
    const int r32 = lane & 31, h = lane >> 5;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
#pragma unroll 1
    for (int qbi = 0; qbi < 2; ++qbi) {
        const int qb = wave * 2 + qbi;
        const int qrow = qb * 32 + r32;
        float qfrag[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) qfrag[st] = Qs[qrow * A::QK_STRIDE + 2 * st + h];
        f32x16 s[A::KT];
        const float* bp = bias_acc + ((size_t)(head * A::QB + qb) * A::KT) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
            const float4* b4 = reinterpret_cast<const float4*>(bp + (size_t)t * 1024);
            const float4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
            f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
            if constexpr (WS == 16) {
                if (lastrow || lastcol) {
                    const bool rowdiff = lastrow && ((t >= 4) != (qb >= 4));
                    const bool qc = (lane >> 3) & 1;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const bool kc = (rg >> 2) & 1;
                        if (rowdiff || (lastcol && (kc != qc))) acc[rg] += -100.0f;
                    }
                }
            }
            const float* krow = Ks + (t * 32 + r32) * A::QK_STRIDE + h;
#pragma unroll
            for (int st = 0; st < 16; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * st], qfrag[st], acc, 0, 0, 0);
            s[t] = acc;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) mx = fmaxf(mx, s[t][rg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const float e = expf(s[t][rg] - mx);
                s[t][rg] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        f32x16 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* vrow = Vt + r32 * A::VT_STRIDE + 4 * h;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg)
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[t * 32 + (rg & 3) + 8 * (rg >> 2)], s[t][rg], o, 0, 0, 0);
        const float inv = 1.0f / sum;
        const size_t e0 = token_row(qrow) * (size_t)C + head * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (x3) x3_store4(out, e0 + 8 * g + 4 * h, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);   // x3 operand of the proj GEMM (half16.h)
            else *reinterpret_cast<float4*>(out + e0 + 8 * g + 4 * h) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
    }
}

template <int WS, bool F16, int QS>
__global__ __launch_bounds__(AttnGenCfg<WS>::THREADS) void window_attention_flash_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                     const float* __restrict__ scale, bf16_t* __restrict__ out, int res,
                                                                     int shift, int heads, int out_x3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    window_attention_flash_body<WS, F16, QS>(qkv, bias_acc, scale, out, res, shift, heads, out_x3, (int)blockIdx.x, (int)threadIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// Exact-f32 windowed cosine attention for the large windows (24 x 24 = 576 and 12 x 12 = 144 tokens of dpt_swin2_base_384) on
// v_mfma_f32_32x32x2_f32: the streaming form of vit_attention_f32_kernel (vit_attention.hip) with head dimension 32, the window / cyclic-shift
// token gather, q-hat / k-hat normalisation at staging time, the CPB bias tile as the initial accumulator (bias_acc is already in MFMA
// accumulator order; padded keys carry -1e30 there) and the shift mask from the token coordinates.  One workgroup = 4 waves = 4 blocks of 32
// queries of one (sample, window, head); key / value tiles of 32 tokens stream through a double-buffered LDS ring, one barrier per tile.
// Replaces the one-thread-per-query VALU kernel below for these window sizes in the F32 and F16X3 modes (round 3: 10.5 -> see DESIGN ms per
// base_384 forward at B = 8).  x3 != 0: the output is written as the x3 operand of the proj GEMM.
// ---------------------------------------------------------------------------------------------
constexpr int WKS = 36;   // floats per K row in LDS (144 B = 9 x 16: conflict-free ds_read_b128 over a half-wave's 32 rows)

template <int WS>
__global__ __launch_bounds__(256) void window_attention_f32_flash_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                         const float* __restrict__ scale, float* __restrict__ out, int res, int shift,
                                                                         int heads, int x3) {
    constexpr int N = WS * WS, NT = (N + 31) / 32, NQB = (NT + 3) / 4, HALF = WS / 2;
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * WKS];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int part = bid % NQB;
    bid /= NQB;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const float hscale = scale[head];
    auto token_row = [&](int p) -> size_t {
        const int r = p / WS, c = p % WS;
        int sy = wy * WS + r + shift, sx = wx * WS + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int r32 = lane & 31, h = lane >> 5;
    const int qb = part * 4 + wave;
    const bool active = qb < NT;             // wave-uniform
    const int qrow = qb * 32 + r32;
    const int qcl = qrow < N ? qrow : N - 1;
    const bool qr_hi = (qcl / WS) >= HALF, qc_hi = (qcl % WS) >= HALF;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);

    // staging: thread -> (token of the tile = tid >> 3, 16-byte chunk c = tid & 7); k-hat = k / max(|k|, 1e-12): the 8 threads of a token share the norm
    float4 kr, vr;
    auto gload = [&](int t) {
        const int p = t * 32 + (tid >> 3), c = tid & 7;
        kr = make_float4(0.f, 0.f, 0.f, 0.f);
        vr = kr;
        if (p < N) {
            const float* src = qkv + token_row(p) * (size_t)(3 * C) + head * 32 + c * 4;
            kr = *reinterpret_cast<const float4*>(src + C);
            vr = *reinterpret_cast<const float4*>(src + 2 * C);
        }
        float ks = kr.x * kr.x + kr.y * kr.y + kr.z * kr.z + kr.w * kr.w;
        ks += __shfl_xor(ks, 1);
        ks += __shfl_xor(ks, 2);
        ks += __shfl_xor(ks, 4);
        const float ki = 1.0f / fmaxf(sqrtf(ks), 1e-12f);
        kr.x *= ki; kr.y *= ki; kr.z *= ki; kr.w *= ki;
    };
    auto lstore = [&](int buf) {
        const int kt = tid >> 3, c = tid & 7;
        *reinterpret_cast<float4*>(&Ks[buf][kt * WKS + c * 4]) = kr;
        *reinterpret_cast<float4*>(&Vs[buf][kt * 32 + c * 4]) = vr;
    };
    // Q-hat fragment (B operand): lane (query r32, half h), step st <-> d = 16 h + st
    float qf[16];
    {
        const float* src = qkv + token_row(qcl) * (size_t)(3 * C) + head * 32 + 16 * h;
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 q4 = *reinterpret_cast<const float4*>(src + 4 * j);
            qf[4 * j] = q4.x; qf[4 * j + 1] = q4.y; qf[4 * j + 2] = q4.z; qf[4 * j + 3] = q4.w;
            qs += q4.x * q4.x + q4.y * q4.y + q4.z * q4.z + q4.w * q4.w;
        }
        qs += __shfl_xor(qs, 32);
        const float qi = hscale / fmaxf(sqrtf(qs), 1e-12f);
#pragma unroll
        for (int j = 0; j < 16; ++j) qf[j] *= qi;
    }
    gload(0);
    lstore(0);
    __syncthreads();
    constexpr float LOG2E = 1.4426950408889634f;
    float m = -3.0e38f, l = 0.f;
    f32x16 o;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) o[rg] = 0.f;
    const float* bp = bias_acc + ((size_t)(head * NT + (active ? qb : 0)) * NT) * 1024 + lane * 16;
    float4 nb0, nb1, nb2, nb3;
    {
        const float4* b4 = reinterpret_cast<const float4*>(bp);
        nb0 = b4[0]; nb1 = b4[1]; nb2 = b4[2]; nb3 = b4[3];
    }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        if (t + 1 < NT) gload(t + 1);
        if (active) {
            const float4 b0 = nb0, b1 = nb1, b2 = nb2, b3 = nb3;
            if (t + 1 < NT) {
                const float4* b4 = reinterpret_cast<const float4*>(bp + (size_t)(t + 1) * 1024);
                nb0 = b4[0]; nb1 = b4[1]; nb2 = b4[2]; nb3 = b4[3];
            }
            f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
            if (lastrow || lastcol) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    int key = t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    key = key < N ? key : N - 1;
                    const bool kr_hi = (key / WS) >= HALF, kc_hi = (key % WS) >= HALF;
                    if ((lastrow && (kr_hi != qr_hi)) || (lastcol && (kc_hi != qc_hi))) acc[rg] += -100.0f;
                }
            }
            const float* krow = &Ks[buf][r32 * WKS + 16 * h];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 k4 = *reinterpret_cast<const float4*>(krow + 4 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[4 * j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[4 * j + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[4 * j + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[4 * j + 3], acc, 0, 0, 0);
            }
            float mt = acc[0];
#pragma unroll
            for (int rg = 1; rg < 16; ++rg) mt = fmaxf(mt, acc[rg]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m, mt);
            const float mnl = mn * LOG2E;
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m, LOG2E, -mnl));
            float psum = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                acc[rg] = __builtin_amdgcn_exp2f(fmaf(acc[rg], LOG2E, -mnl));
                psum += acc[rg];
                o[rg] *= alpha;
            }
            l = l * alpha + psum;
            m = mn;
            const float* vcol = &Vs[buf][(4 * h) * 32 + r32];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vcol[((rg & 3) + 8 * (rg >> 2)) * 32], acc[rg], o, 0, 0, 0);
        }
        if (t + 1 < NT) lstore(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32);
    if (qrow < N) {
        const float inv = 1.0f / l;
        const size_t e0 = token_row(qrow) * (size_t)C + head * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (x3) x3_store4(out, e0 + 8 * g + 4 * h, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            else *reinterpret_cast<float4*>(out + e0 + 8 * g + 4 * h) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exact-f32 attention for ANY window size (parity mode of dpt_swin2_base_384: 24x24 / 12x12 windows).
// Not a throughput kernel: one thread owns one query (q-hat and the output row in registers), keys are staged
// 64 at a time in LDS (normalised K and V rows, read by broadcast), online softmax in f32, CPB bias read from the
// (2ws-1)^2 x heads table by relative position, shift mask from the token coordinates.  One workgroup = 64 queries
// of one (batch, window, head).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void window_attention_f32_any_kernel(const float* __restrict__ qkv, const float* __restrict__ table,
                                                                      const float* __restrict__ scale, float* __restrict__ out, int res, int ws,
                                                                      int shift, int heads, int x3) {
    __shared__ float Ks[64][33];
    __shared__ float Vs[64][33];
    const int N = ws * ws, nqb = (N + 63) / 64;
    const int C = heads * 32, nw = res / ws;
    int bid = blockIdx.x;
    const int qb = bid % nqb;
    bid /= nqb;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const int tid = threadIdx.x;
    auto token_row = [&](int p) -> size_t {
        const int r = p / ws, c = p % ws;
        int sy = wy * ws + r + shift, sx = wx * ws + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int q = qb * 64 + tid;
    const bool qv = q < N;
    const int qc = qv ? q : N - 1;
    const int rq = qc / ws, cq = qc % ws;
    float qh[32], o[32];
    {
        const float* src = qkv + token_row(qc) * (size_t)(3 * C) + head * 32;
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) { qh[d] = src[d]; ss += qh[d] * qh[d]; o[d] = 0.f; }
        const float qi = scale[head] / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < 32; ++d) qh[d] *= qi;
    }
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    const int half = ws / 2;
    float m = -3.0e38f, l = 0.f;
    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();
        {   // stage 64 keys: thread = key
            const int k = k0 + tid;
            const int kc = k < N ? k : N - 1;
            const float* src = qkv + token_row(kc) * (size_t)(3 * C) + head * 32;
            float ss = 0.f;
            float kr[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) { kr[d] = src[C + d]; ss += kr[d] * kr[d]; }
            const float ki = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int d = 0; d < 32; ++d) { Ks[tid][d] = kr[d] * ki; Vs[tid][d] = src[2 * C + d]; }
        }
        __syncthreads();
        const int kn = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < kn; ++kk) {
            const int k = k0 + kk;
            const int rk = k / ws, ck = k % ws;
            float sdot = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) sdot = fmaf(qh[d], Ks[kk][d], sdot);
            sdot += table[(size_t)((rq - rk + ws - 1) * (2 * ws - 1) + (cq - ck + ws - 1)) * heads + head];
            if ((lastrow && ((rk >= half) != (rq >= half))) || (lastcol && ((ck >= half) != (cq >= half)))) sdot += -100.0f;
            const float mn = fmaxf(m, sdot);
            const float alpha = expf(m - mn), p = expf(sdot - mn);
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] = fmaf(p, Vs[kk][d], o[d] * alpha);
            m = mn;
        }
    }
    if (qv) {
        const float inv = 1.0f / l;
        const size_t e0 = token_row(q) * (size_t)C + head * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            if (x3) x3_store4(out, e0 + d, o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
            else *reinterpret_cast<float4*>(out + e0 + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
        }
    }
}

// CPB bias in accumulator order: [head][qb][t][lane][16]; value for query 32qb+(lane&31),
// key 32t + (reg&3) + 8(reg>>2) + 4(lane>>5)
__global__ void attn_bias_kernel(const float* __restrict__ table, float* __restrict__ bias_acc, int ws, int heads) {
    const int N = ws * ws, NT = (N + 31) / 32;
    const size_t total = (size_t)heads * NT * NT * 1024;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int rg = (int)(i & 15), lane = (int)((i >> 4) & 63);
        size_t r = i >> 10;
        const int t = (int)(r % NT);
        r /= NT;
        const int qb = (int)(r % NT);
        const int head = (int)(r / NT);
        const int q = 32 * qb + (lane & 31), k = 32 * t + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        float v = 0.f;
        if (k >= N) v = -1.0e30f;  // padded key: zero probability
        else if (q < N) {
            const int rq = q / ws, cq = q % ws, rk = k / ws, ck = k % ws;
            const int idx = (rq - rk + ws - 1) * (2 * ws - 1) + (cq - ck + ws - 1);
            v = table[(size_t)idx * heads + head];
        }
        bias_acc[i] = v;
    }
}

size_t attn_bias_elems(int ws, int heads) {
    const size_t nt = ((size_t)ws * ws + 31) / 32;
    return (size_t)heads * nt * nt * 1024;
}

int launch_attn_bias(const float* table, float* bias_acc, int ws, int heads, hipStream_t st, std::string& err) {
    const size_t total = attn_bias_elems(ws, heads);
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SOCCDPT_LAUNCH(attn_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, st, table, bias_acc, ws, heads);
    return check_launch("attn_bias", err);
}

int launch_window_attention_f32(const float* qkv, const float* bias_acc, const float* table, const float* scale, float* out, int B, int res,
                                int ws, int shift, int heads, hipStream_t st, std::string& err, int x3) {
    if (res % ws != 0) { err = "window_attention: res % ws != 0"; return 1; }
    const int nw = res / ws;
    static const bool force_any = getenv("SOCCDPT_ATTN_F32_ANY") != nullptr;   // A/B switch: the one-thread-per-query kernel
    // 16 x 16 / 8 x 8 windows too (round 3, late): 0.507 -> 0.296 ms per tiny_256 forward at B = 8 against the whole-score-matrix kernel below
    // (SOCCDPT_ATTN_F32_FLASH16=0 selects that one for A/B)
    static const int flash_small = getenv("SOCCDPT_ATTN_F32_FLASH16") ? atoi(getenv("SOCCDPT_ATTN_F32_FLASH16")) : 1;
    const bool flash = (ws == 24 || ws == 12) || ((ws == 16 || ws == 8) && flash_small == 1);
    if (flash && !force_any) {   // the streaming MFMA-f32 kernel
        const int NT = (ws * ws + 31) / 32, NQB = (NT + 3) / 4;
        const unsigned blocksf = (unsigned)(B * nw * nw * heads * NQB);
        if (ws == 24) SOCCDPT_LAUNCH((window_attention_f32_flash_kernel<24>), dim3(blocksf), dim3(256), 0, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
        else if (ws == 12) SOCCDPT_LAUNCH((window_attention_f32_flash_kernel<12>), dim3(blocksf), dim3(256), 0, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
        else if (ws == 16) SOCCDPT_LAUNCH((window_attention_f32_flash_kernel<16>), dim3(blocksf), dim3(256), 0, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
        else SOCCDPT_LAUNCH((window_attention_f32_flash_kernel<8>), dim3(blocksf), dim3(256), 0, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
        return check_launch("window_attention_f32_flash", err);
    }
    if (ws != 16 && !(ws == 8 && shift == 0)) {  // any other window size: the generic exact kernel (parity mode of base_384)
        const int nqb = (ws * ws + 63) / 64;
        SOCCDPT_LAUNCH(window_attention_f32_any_kernel, dim3((unsigned)(B * nw * nw * heads * nqb)), dim3(64), 0, st, qkv, table, scale, out, res, ws,
                           shift, heads, x3);
        return check_launch("window_attention_f32_any", err);
    }
    const unsigned blocks = (unsigned)(B * nw * nw * heads);
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attention_f32_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, AttnCfgF32<16>::LDS);
        attr_done.done();
    }
    if (ws == 16) {
        using A = AttnCfgF32<16>;
        SOCCDPT_LAUNCH((window_attention_f32_kernel<16>), dim3(blocks), dim3(A::THREADS), A::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
    } else if (ws == 8 && shift == 0) {
        using A = AttnCfgF32<8>;
        SOCCDPT_LAUNCH((window_attention_f32_kernel<8>), dim3(blocks), dim3(A::THREADS), A::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, x3);
    } else {
        err = "window_attention_f32: window size not instantiated (16 and unshifted 8 are)";
        return 1;
    }
    return check_launch("window_attention_f32", err);
}

int launch_window_attention(const bf16_t* qkv, const float* bias_acc, const float* scale, bf16_t* out, int hf, int B, int res, int ws,
                            int shift, int heads, hipStream_t st, std::string& err, int out_x3) {
    if (res % ws != 0) { err = "window_attention: res % ws != 0"; return 1; }
    if (out_x3 && !hf) { err = "window_attention: the x3 output form belongs to the fp16 kernels"; return 1; }
    const int nw = res / ws;
    const unsigned blocks = (unsigned)(B * nw * nw * heads);
    if (ws == 16) {
        // 16x16 windows run the online-softmax kernel too: holding all 8 score tiles of a query block (the kernel above) costs 192-404
        // registers per lane = one or two waves per SIMD; tile-at-a-time softmax needs 127, and with 8 waves per workgroup (one query
        // block each) the CU keeps 16 waves busy: 0.184 -> 0.165 ms of attention per tiny_256 forward.
        static PerDeviceOnce attr16;
        if (attr16.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attention_flash_kernel<16, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, AttnGenCfg<16>::LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attention_flash_kernel<16, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, AttnGenCfg<16>::LDS);
            attr16.done();
        }
        if (hf) SOCCDPT_LAUNCH((window_attention_flash_kernel<16, true, 1>), dim3(blocks), dim3(AttnGenCfg<16>::THREADS), AttnGenCfg<16>::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, out_x3);
        else SOCCDPT_LAUNCH((window_attention_flash_kernel<16, false, 1>), dim3(blocks), dim3(AttnGenCfg<16>::THREADS), AttnGenCfg<16>::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, out_x3);
    } else if (ws == 8) {
        using A = AttnCfg<8>;
        if (shift != 0) { err = "window_attention: shifted 8x8 windows are not instantiated"; return 1; }
        if (hf) SOCCDPT_LAUNCH((window_attention_kernel<8, true, 1>), dim3(blocks), dim3(A::THREADS), A::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, out_x3);
        else SOCCDPT_LAUNCH((window_attention_kernel<8, false, 1>), dim3(blocks), dim3(A::THREADS), A::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, out_x3);
    } else if (ws == 24 || ws == 12) {
        static PerDeviceOnce attr_done;
        if (attr_done.need()) {
#define FLASH_ATTR(H, Q) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attention_flash_kernel<24, H, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, AttnGenCfg<24>::LDS)
            FLASH_ATTR(false, 1); FLASH_ATTR(true, 1); FLASH_ATTR(false, 2); FLASH_ATTR(true, 2);
#undef FLASH_ATTR
            attr_done.done();
        }
#define FLASH(W, H, Q) SOCCDPT_LAUNCH((window_attention_flash_kernel<W, H, Q>), dim3(blocks * Q), dim3(AttnGenCfg<W>::THREADS), AttnGenCfg<W>::LDS, st, qkv, bias_acc, scale, out, res, shift, heads, out_x3)
        if (ws == 24 && blocks < 256) { if (hf) FLASH(24, true, 2); else FLASH(24, false, 2); }   // too few (window, head) pairs: split the queries
        else if (ws == 24) { if (hf) FLASH(24, true, 1); else FLASH(24, false, 1); }
        else { if (hf) FLASH(12, true, 1); else FLASH(12, false, 1); }
#undef FLASH
    } else {
        err = "window_attention: window size not instantiated (16, 8, 24, 12 are)";
        return 1;
    }
    return check_launch("window_attention", err);
}

}  // namespace soccdpt
