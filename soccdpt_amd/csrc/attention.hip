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

#include "half16.h"
#include "kernels.h"

namespace soccdpt {

typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WS>
struct AttnCfg {
    static constexpr int N = WS * WS;
    static constexpr int WAVES = N / 64;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int KT = N / 32;  // 32-key tiles
    static constexpr int QB = N / 32;  // 32-query blocks
    static constexpr int VT_STRIDE = N * 2 + 8;
    static constexpr int QS_OFF = 0, KS_OFF = N * 64, VT_OFF = 2 * N * 64;
    static constexpr int LDS = 2 * N * 64 + 32 * VT_STRIDE;
};


// QS: query split.  QS == 2 gives each (batch, window, head) to two workgroups that stage all of K-hat / V^T but own half of the
// query blocks: stages 1-2 of the B = 8 forward have only 192 / 96 (window, head) pairs for 256 CUs (0.258 -> 0.216 ms per
// forward; a 4-way split with two staging-only waves measured slower again).
template <int WS, bool F16, int QS>
__global__ __launch_bounds__(AttnCfg<WS>::THREADS) void window_attention_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                                 const float* __restrict__ scale, bf16_t* __restrict__ out,
                                                                                 int res, int shift, int heads, int out_x3) {
    using A = AttnCfg<WS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem + A::QS_OFF;
    char* Ks = smem + A::KS_OFF;
    char* Vt = smem + A::VT_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int qh = QS > 1 ? bid % QS : 0;  // which part of the query blocks this workgroup owns
    bid /= QS;
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

    // ---- stage Q-hat (own query rows only), K-hat, V^T: all loads of the thread are requested first (one memory latency) ----
    constexpr int STG = (A::N * 4) / A::THREADS;
    uint4 qld[STG], kld[STG], vld[STG];
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int idx = it * A::THREADS + tid;
        const int p = idx >> 2, c = idx & 3;
        const bf16_t* src = qkv + token_row(p) * (size_t)(3 * C) + head * 32 + c * 8;
        const bool own_q = QS == 1 || (p / (A::N / QS)) == qh;  // uniform over the 4 lanes of a token
        qld[it] = own_q ? *reinterpret_cast<const uint4*>(src) : make_uint4(0u, 0u, 0u, 0u);
        kld[it] = *reinterpret_cast<const uint4*>(src + C);
        vld[it] = *reinterpret_cast<const uint4*>(src + 2 * C);
    }
#pragma unroll
    for (int it = 0; it < STG; ++it) {
        const int idx = it * A::THREADS + tid;
        const int p = idx >> 2, c = idx & 3;
        const bool own_q = QS == 1 || (p / (A::N / QS)) == qh;
        const uint4 qv = qld[it], kv = kld[it], vv = vld[it];
        const uint32_t qu[4] = {qv.x, qv.y, qv.z, qv.w}, ku[4] = {kv.x, kv.y, kv.z, kv.w}, vu[4] = {vv.x, vv.y, vv.z, vv.w};
        float qf[8], kf[8];
        float qs = 0.f, ks = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[2 * j] = h_lo<F16>(qu[j]);
            qf[2 * j + 1] = h_hi<F16>(qu[j]);
            kf[2 * j] = h_lo<F16>(ku[j]);
            kf[2 * j + 1] = h_hi<F16>(ku[j]);
            qs += qf[2 * j] * qf[2 * j] + qf[2 * j + 1] * qf[2 * j + 1];
            ks += kf[2 * j] * kf[2 * j] + kf[2 * j + 1] * kf[2 * j + 1];
        }
        qs += __shfl_xor(qs, 1);
        qs += __shfl_xor(qs, 2);
        ks += __shfl_xor(ks, 1);
        ks += __shfl_xor(ks, 2);
        const float qi = hscale / fmaxf(sqrtf(qs), 1e-12f);  // F.normalize eps
        const float ki = 1.0f / fmaxf(sqrtf(ks), 1e-12f);
        uint4 qo, ko;
        qo.x = pack_h2<F16>(qf[0] * qi, qf[1] * qi); qo.y = pack_h2<F16>(qf[2] * qi, qf[3] * qi);
        qo.z = pack_h2<F16>(qf[4] * qi, qf[5] * qi); qo.w = pack_h2<F16>(qf[6] * qi, qf[7] * qi);
        ko.x = pack_h2<F16>(kf[0] * ki, kf[1] * ki); ko.y = pack_h2<F16>(kf[2] * ki, kf[3] * ki);
        ko.z = pack_h2<F16>(kf[4] * ki, kf[5] * ki); ko.w = pack_h2<F16>(kf[6] * ki, kf[7] * ki);
        const int sw = (c ^ ((p >> 2) & 3)) * 16;
        if (own_q) *reinterpret_cast<uint4*>(Qs + p * 64 + sw) = qo;
        *reinterpret_cast<uint4*>(Ks + p * 64 + sw) = ko;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j) * A::VT_STRIDE + p * 2) = (uint16_t)(vu[j] & 0xffffu);
            *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j + 1) * A::VT_STRIDE + p * 2) = (uint16_t)(vu[j] >> 16);
        }
    }
    __syncthreads();

    const int r32 = lane & 31, h = lane >> 5;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    static_assert(QS == 1 || QS == 2, "query split");
#pragma unroll 1
    for (int qbi = 0; qbi < 2 / QS; ++qbi) {
        const int qb = qh * (A::QB / QS) + wave * (2 / QS) + qbi;
        const int qrow = qb * 32 + r32;
        h16x8 qfrag[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qfrag[ks] = *reinterpret_cast<const h16x8*>(Qs + qrow * 64 + (((ks * 2 + h) ^ ((qrow >> 2) & 3)) * 16));

        f32x16 s[A::KT];
        const float* bp = bias_acc + ((size_t)(head * A::QB + qb) * A::KT) * 1024 + lane * 16;
        // the bias tile of key tile t + 1 is requested before the MFMAs of tile t (the accumulator cannot start without it)
        float4 n0, n1, n2, n3;
        { const float4* b4 = reinterpret_cast<const float4*>(bp); n0 = b4[0]; n1 = b4[1]; n2 = b4[2]; n3 = b4[3]; }
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
            const float4 b0 = n0, b1 = n1, b2 = n2, b3 = n3;
            if (t + 1 < A::KT) {
                const float4* b4 = reinterpret_cast<const float4*>(bp + (size_t)(t + 1) * 1024);
                n0 = b4[0]; n1 = b4[1]; n2 = b4[2]; n3 = b4[3];
            }
            f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
            if constexpr (WS == 16) {
                // shift mask (0 / -100): region differs in the last window row (token rows >= 8) or column (cols >= 8)
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
            const int krow = t * 32 + r32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h16x8 kfrag = *reinterpret_cast<const h16x8*>(Ks + krow * 64 + (((ks * 2 + h) ^ ((krow >> 2) & 3)) * 16));
                acc = mfma_32x32x16<F16>(kfrag, qfrag[ks], acc);
            }
            s[t] = acc;
        }
        // ---- softmax over keys: lane-local over (t, reg) + the other half-wave ----
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) mx = fmaxf(mx, s[t][rg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        const float mxl = mx * 1.4426950408889634f;  // exp(s - mx) = 2^(s*log2e - mx*log2e): one v_fma + one v_exp per logit
#pragma unroll
        for (int t = 0; t < A::KT; ++t)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[t][rg], 1.4426950408889634f, -mxl));
                s[t][rg] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        // ---- O^T = V^T P^T ----
        f32x16 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < A::KT; ++t) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                h16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (short)f2h<F16>(s[t][8 * st + j]);
                // element j of this lane half is key 32t + 16st + 8(j>>2) + 4h + (j&3): V^T must use the same k order
                const char* vrow = Vt + r32 * A::VT_STRIDE + (t * 32 + st * 16 + 4 * h) * 2;
                const h16x4 v0 = *reinterpret_cast<const h16x4*>(vrow);
                const h16x4 v1 = *reinterpret_cast<const h16x4*>(vrow + 16);
                const h16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o = mfma_32x32x16<F16>(vf, pb, o);
            }
        }
        // ---- store: lane owns query column r32, rows d = (rg&3) + 8(rg>>2) + 4h ----
        const float inv = 1.0f / sum;
        const size_t e0 = token_row(qrow) * (size_t)C + head * 32;
        bf16_t* orow = out + e0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (F16 && out_x3) {   // SOCCDPT_PREC_MIXED: the proj GEMM of this block reads x3 operands (half16.h); the f32 accumulators go out unrounded
                x3_store4(out, e0 + 8 * g + 4 * h, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                continue;
            }
            uint2 pkt;
            pkt.x = pack_h2<F16>(o[4 * g] * inv, o[4 * g + 1] * inv);
            pkt.y = pack_h2<F16>(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(orow + 8 * g + 4 * h) = pkt;
        }
    }
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

// ---------------------------------------------------------------------------------------------
// Generic window size (24x24 / 12x12 of dpt_swin2_base_384): same operand layout, but the N x N score matrix
// no longer fits the register file (576 keys = 18 tiles x 16 accumulators), so keys are consumed tile by tile
// with an online softmax (running max m, running sum l, O rescaled by exp(m - m_new) per tile).  N is padded
// to a multiple of 32: padded keys carry a -1e30 bias (zero probability), padded queries are not stored.
// The shift mask is evaluated arithmetically from the token coordinates.
// ---------------------------------------------------------------------------------------------
template <int WS>
struct AttnGenCfg {
    static constexpr int N = WS * WS;
    static constexpr int NT = (N + 31) / 32;
    static constexpr int NPAD = NT * 32;
    // 24x24 windows need 111 KB of LDS (one workgroup per CU): 16 waves, so that every SIMD has four waves to overlap the online
    // softmax (VALU) of one query block with the MFMAs / LDS reads of others (4 waves: 1.43 ms per base_384 forward, 8: 0.98, 16: 0.85);
    // the 12x12 windows (37 KB, four workgroups per CU) keep 4 waves
    static constexpr int THREADS = WS >= 24 ? 1024 : (WS == 16 ? 512 : 256);   // 16x16: 8 query blocks, one per wave
    static constexpr int VT_STRIDE = NPAD * 2 + 8;
    static constexpr int KS_OFF = NPAD * 64, VT_OFF = 2 * NPAD * 64;
    static constexpr int LDS = 2 * NPAD * 64 + 32 * VT_STRIDE;
};

// QS: query split as in window_attention_kernel (each workgroup stages all keys / values, owns 1/QS of the 32-query blocks).
template <int WS, bool F16, int QS>
__global__ __launch_bounds__(AttnGenCfg<WS>::THREADS) void window_attention_flash_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                                     const float* __restrict__ scale, bf16_t* __restrict__ out, int res,
                                                                     int shift, int heads, int out_x3) {
    using A = AttnGenCfg<WS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = smem + A::KS_OFF;
    char* Vt = smem + A::VT_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int qh = QS > 1 ? bid % QS : 0;
    bid /= QS;
    constexpr int QB0 = (AttnGenCfg<WS>::NT + QS - 1) / QS;   // query blocks per workgroup
    const int qb_lo = qh * QB0, qb_hi = (qb_lo + QB0) < AttnGenCfg<WS>::NT ? (qb_lo + QB0) : AttnGenCfg<WS>::NT;
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
    for (int idx = tid; idx < A::NPAD * 4; idx += A::THREADS) {
        const int p = idx >> 2, c = idx & 3;
        uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv;
        const bool own_q = QS == 1 || ((p >> 5) >= qb_lo && (p >> 5) < qb_hi);
        if (p < A::N) {
            const bf16_t* src = qkv + token_row(p) * (size_t)(3 * C) + head * 32 + c * 8;
            if (own_q) qv = *reinterpret_cast<const uint4*>(src);
            kv = *reinterpret_cast<const uint4*>(src + C);
            vv = *reinterpret_cast<const uint4*>(src + 2 * C);
        }
        const uint32_t qu[4] = {qv.x, qv.y, qv.z, qv.w}, ku[4] = {kv.x, kv.y, kv.z, kv.w}, vu[4] = {vv.x, vv.y, vv.z, vv.w};
        float qf[8], kf[8];
        float qs = 0.f, ks = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[2 * j] = h_lo<F16>(qu[j]);
            qf[2 * j + 1] = h_hi<F16>(qu[j]);
            kf[2 * j] = h_lo<F16>(ku[j]);
            kf[2 * j + 1] = h_hi<F16>(ku[j]);
            qs += qf[2 * j] * qf[2 * j] + qf[2 * j + 1] * qf[2 * j + 1];
            ks += kf[2 * j] * kf[2 * j] + kf[2 * j + 1] * kf[2 * j + 1];
        }
        qs += __shfl_xor(qs, 1);
        qs += __shfl_xor(qs, 2);
        ks += __shfl_xor(ks, 1);
        ks += __shfl_xor(ks, 2);
        const float qi = hscale / fmaxf(sqrtf(qs), 1e-12f);
        const float ki = 1.0f / fmaxf(sqrtf(ks), 1e-12f);
        uint4 qo, ko;
        qo.x = pack_h2<F16>(qf[0] * qi, qf[1] * qi); qo.y = pack_h2<F16>(qf[2] * qi, qf[3] * qi);
        qo.z = pack_h2<F16>(qf[4] * qi, qf[5] * qi); qo.w = pack_h2<F16>(qf[6] * qi, qf[7] * qi);
        ko.x = pack_h2<F16>(kf[0] * ki, kf[1] * ki); ko.y = pack_h2<F16>(kf[2] * ki, kf[3] * ki);
        ko.z = pack_h2<F16>(kf[4] * ki, kf[5] * ki); ko.w = pack_h2<F16>(kf[6] * ki, kf[7] * ki);
        const int sw = (c ^ ((p >> 2) & 3)) * 16;
        if (own_q) *reinterpret_cast<uint4*>(Qs + p * 64 + sw) = qo;
        *reinterpret_cast<uint4*>(Ks + p * 64 + sw) = ko;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j) * A::VT_STRIDE + p * 2) = (uint16_t)(vu[j] & 0xffffu);
            *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j + 1) * A::VT_STRIDE + p * 2) = (uint16_t)(vu[j] >> 16);
        }
    }
    __syncthreads();
    const int r32 = lane & 31, h = lane >> 5;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    constexpr int HALF = WS / 2;
    for (int qb = qb_lo + wave; qb < qb_hi; qb += A::THREADS / 64) {
        const int qrow = qb * 32 + r32;
        const int qcl = qrow < A::N ? qrow : A::N - 1;
        const bool qr_hi = (qcl / WS) >= HALF, qc_hi = (qcl % WS) >= HALF;
        h16x8 qfrag[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qfrag[ks] = *reinterpret_cast<const h16x8*>(Qs + qrow * 64 + (((ks * 2 + h) ^ ((qrow >> 2) & 3)) * 16));
        float m = -3.0e38f, l = 0.f;
        f32x16 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* bp = bias_acc + ((size_t)(head * A::NT + qb) * A::NT) * 1024 + lane * 16;
        // the bias tile is the initial accumulator: fetch tile t+1 under tile t's MFMAs / softmax (the loop is not unrolled, and an
        // un-prefetched L2 read per key tile was the critical path of this kernel)
        float4 nb0, nb1, nb2, nb3;
        {
            const float4* b4 = reinterpret_cast<const float4*>(bp);
            nb0 = b4[0]; nb1 = b4[1]; nb2 = b4[2]; nb3 = b4[3];
        }
#pragma unroll 1
        for (int t = 0; t < A::NT; ++t) {
            const float4 b0 = nb0, b1 = nb1, b2 = nb2, b3 = nb3;
            if (t + 1 < A::NT) {
                const float4* b4 = reinterpret_cast<const float4*>(bp + (size_t)(t + 1) * 1024);
                nb0 = b4[0]; nb1 = b4[1]; nb2 = b4[2]; nb3 = b4[3];
            }
            f32x16 acc = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
            if (lastrow || lastcol) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    int key = t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    key = key < A::N ? key : A::N - 1;
                    const bool kr_hi = (key / WS) >= HALF, kc_hi = (key % WS) >= HALF;
                    if ((lastrow && (kr_hi != qr_hi)) || (lastcol && (kc_hi != qc_hi))) acc[rg] += -100.0f;
                }
            }
            const int krow = t * 32 + r32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h16x8 kfrag = *reinterpret_cast<const h16x8*>(Ks + krow * 64 + (((ks * 2 + h) ^ ((krow >> 2) & 3)) * 16));
                acc = mfma_32x32x16<F16>(kfrag, qfrag[ks], acc);
            }
            float mt = acc[0];
#pragma unroll
            for (int rg = 1; rg < 16; ++rg) mt = fmaxf(mt, acc[rg]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m, mt);
            const float mnl = mn * 1.4426950408889634f;   // exp(s - mn) = 2^(s*log2e - mn*log2e): one v_fma + one v_exp per logit
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m, 1.4426950408889634f, -mnl));
            float psum = 0.f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                acc[rg] = __builtin_amdgcn_exp2f(fmaf(acc[rg], 1.4426950408889634f, -mnl));
                psum += acc[rg];
                o[rg] *= alpha;
            }
            l = l * alpha + psum;
            m = mn;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                h16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (short)f2h<F16>(acc[8 * st + j]);
                const char* vrow = Vt + r32 * A::VT_STRIDE + (t * 32 + st * 16 + 4 * h) * 2;
                const h16x4 v0 = *reinterpret_cast<const h16x4*>(vrow);
                const h16x4 v1 = *reinterpret_cast<const h16x4*>(vrow + 16);
                const h16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o = mfma_32x32x16<F16>(vf, pb, o);
            }
        }
        l += __shfl_xor(l, 32);
        if (qrow < A::N) {
            const float inv = 1.0f / l;
            const size_t e0 = token_row(qrow) * (size_t)C + head * 32;
            bf16_t* orow = out + e0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (F16 && out_x3) {   // SOCCDPT_PREC_MIXED: x3 operand of the proj GEMM
                    x3_store4(out, e0 + 8 * g + 4 * h, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                    continue;
                }
                uint2 pkt;
                pkt.x = pack_h2<F16>(o[4 * g] * inv, o[4 * g + 1] * inv);
                pkt.y = pack_h2<F16>(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + 8 * g + 4 * h) = pkt;
            }
        }
    }
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
