// Device bodies of the 16-bit window-attention kernels (attention.hip), kept as device functions so that a caller can pass its own block / thread ids and LDS base.
// Design notes: attention.hip.
#pragma once
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
// The body takes the block index, the thread index within the cooperating group of A::THREADS threads and that group's LDS region explicitly:
// the stand-alone kernel below passes blockIdx.x / threadIdx.x / the dynamic LDS base; a persistent caller (round 4 built one: DESIGN.md section 10.2) runs
// one (window, head) item per wave of a larger workgroup for the single-wave 8 x 8 form.  __syncthreads() inside is workgroup-wide in both uses.
template <int WS, bool F16, int QS>
__device__ __forceinline__ void window_attention_body(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                      const float* __restrict__ scale, bf16_t* __restrict__ out,
                                                      int res, int shift, int heads, int out_x3, int bid, int tid, char* smem) {
    using A = AttnCfg<WS>;
    char* Qs = smem + A::QS_OFF;
    char* Ks = smem + A::KS_OFF;
    char* Vt = smem + A::VT_OFF;
    const int lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
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
                for (int j = 0; j < 8; ++j) pb[j] = (short)f2h_inrange<F16>(s[t][8 * st + j]);
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

// The key loop of the flash form on operands that are already in LDS (Q-hat / K-hat rows of 64 bytes with 16-byte chunks XOR-swizzled by (row >> 2) & 3,
// V^T rows of VT_STRIDE bytes): query blocks qb_lo + wave, + nwaves, ... < qb_hi.  Shared by window_attention_flash_body (operands staged from a qkv tensor)
// and the fused kernel of attention_qkv.hip (operands computed in the workgroup from the block input and the Wqkv slice: round 6).
template <int WS, bool F16>
__device__ __forceinline__ void window_attention_flash_core(const char* Qs, const char* Ks, const char* Vt, const float* __restrict__ bias_acc, bf16_t* __restrict__ out,
                                                            int res, int shift, int heads, int out_x3, int head, int b, int wy, int wx, int qb_lo, int qb_hi,
                                                            int wave, int lane, int nwaves) {
    using A = AttnGenCfg<WS>;
    const int C = heads * 32;
    const int nw = res / WS;
    auto token_row = [&](int p) -> size_t {
        const int r = p / WS, c = p % WS;
        int sy = wy * WS + r + shift, sx = wx * WS + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int r32 = lane & 31, h = lane >> 5;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    constexpr int HALF = WS / 2;
    // (24 x 24 windows unsplit: 18 query blocks on 16 waves, i.e. a second round with 2 waves.  Built and removed in round 5: all 16 waves sharing the two
    //  left-over blocks by key slices that meet in LDS, as vit_attention_kernel's halves do -- 86 -> 98 us per launch: the two waves of the second round have
    //  a SIMD each to themselves and run their 18 tiles in a quarter of the first round's time, less than the slices' merge and the restructured loop cost.)
    for (int qb = qb_lo + wave; qb < qb_hi; qb += nwaves) {
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
            }
            // the key loop is VALU-bound (~180 issue slots per tile against 4 MFMAs, tools/wattn_time.py + the ISA): the running maxima settle after a few
            // tiles, then alpha == 1 in every lane and the 16 multiplications are skipped (x 1.0 is exact: skipping them is too)
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) o[rg] *= alpha;
            }
            l = l * alpha + psum;
            m = mn;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                h16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (short)f2h_inrange<F16>(acc[8 * st + j]);   // probabilities: no clamp needed
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

// QS: query split as in window_attention_kernel (each workgroup stages all keys / values, owns 1/QS of the 32-query blocks).
template <int WS, bool F16, int QS>
__device__ __forceinline__ void window_attention_flash_body(const bf16_t* __restrict__ qkv, const float* __restrict__ bias_acc,
                                                            const float* __restrict__ scale, bf16_t* __restrict__ out, int res,
                                                            int shift, int heads, int out_x3, int bid, int tid, char* smem) {
    using A = AttnGenCfg<WS>;
    char* Qs = smem;
    char* Ks = smem + A::KS_OFF;
    char* Vt = smem + A::VT_OFF;
    const int lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
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
    window_attention_flash_core<WS, F16>(Qs, Ks, Vt, bias_acc, out, res, shift, heads, out_x3, head, b, wy, wx, qb_lo, qb_hi, wave, lane, A::THREADS / 64);
}

}  // namespace soccdpt
