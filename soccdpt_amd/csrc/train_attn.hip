// Cosine window attention BACKWARD on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), training step of the Swin-V2 encoders.
// Replaces the one-thread-per-query / per-key VALU kernels of train.hip (attn_rowstat / attn_bwd_q / attn_bwd_k + the segment sum; those stay as
// the reference form behind SOCCDPT_ATTN_BWD_VALU=1).  Reference: autograd over timm WindowAttention (swin_transformer_v2.py, called from
// /root/reference/SOccDPT/model/backbones/swin_common.py:12-54 through scripts/train_SOccDPT.py:360-393).
//
//   q^ = scale q/|q|, k^ = k/|k|, S = q^ k^T + bias(rel) + mask, P = softmax(S), O = P v
//   dP = dO v^T, delta = dO . O, dS = P (dP - delta), dq^ = dS k^, dk^ = dS^T q^, dv = P^T dO, dscale = sum dS (qn . k^)
//
// Two launches of one template, the streaming shape of the forward kernel (attention.hip window_attention_f32_flash_kernel): a wave OWNS a block of
// 32 tokens (its operand fragments live in registers) and WALKS the other axis in 32-token tiles through a double-buffered LDS ring.  Tiles are
// computed TRANSPOSED -- rows = walked tokens, column = the lane's owned token -- so in
//   PASS 0 (owned = queries, walked = keys) the softmax statistics and delta are per-lane scalars.  Sweep 1: m + ln(l) by online softmax (the
//           forward does not save it); sweep 2: P, dP, dS, dq^ accumulated in an MFMA accumulator, dS stored for the bias-table gradient (train.hip
//           tr_attn_param_grads reduces it over the windows), dscale partial per wave, {m + ln l, delta} stored for pass 1.
//   PASS 1 (owned = keys, walked = queries) P and dS are RECOMPUTED from the same operands (two more MFMA sets, cheaper than reading dS back), the
//           per-query statistics ride along with the walked tile; dk^ and dv accumulate in two MFMA accumulators.
// Every product is an f32 FMA chain inside the MFMA (bitwise an f32 dot product in k order): the gradients are f32-exact like the VALU form, the
// summation order differs.  Deterministic: no atomics, each output element has one owner.
#include <hip/hip_runtime.h>

#include <string>

#include "half16.h"
#include "launch.h"
#include "train.h"

namespace soccdpt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int AST = 36;   // floats per LDS row (144 B): conflict-free for the 16-byte row reads of a half-wave and for the column reads
template <int OP>
__device__ __forceinline__ h16x8 frag8(const float* v) {   // OP 1: bf16 (RNE), OP 2: IEEE fp16 without a clamp (an overflow must reach GradScaler as inf: half16.h f2h_ieee)
    h16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)(OP == 2 ? f2h_ieee(v[j]) : f2h<false>(v[j]));
    return f;
}

// OP: operands of the MFMA products dP, dq^ / dk^ and dv -- 0 exact f32 (v_mfma_f32_32x32x2_f32; f32 / x3 training), 1 bf16, 2 fp16 (v_mfma_f32_32x32x16: the amp modes, where
// autocast runs these matmuls in 16 bits too, /root/reference/SOccDPT/scripts/train_SOccDPT.py:340-343).  The tiles stay f32 in LDS; fragments are rounded in
// registers: the same k <-> (step, lane half, element) mapping on both operands of a product, so any mapping is a valid summation order.  Round 5: the f32 form
// is 48-64 MFMAs x 64 cycles per 32 x 32 tile and wave -- with one wave per SIMD the whole kernel; in the amp modes 16 of them stay (the scores) and the other
// 32-48 become 4-6 16-bit MFMAs x 32 cycles.
template <int WS, int PASS, int OP = 0>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ attn_out,
                                                            const float* __restrict__ table, const float* __restrict__ scale, float* __restrict__ rowstat,
                                                            float* __restrict__ dS_out, float* __restrict__ dscale_part, float* __restrict__ dqkv, int res,
                                                            int shift, int heads) {
    constexpr int N = WS * WS, NT = (N + 31) / 32, NOB = (NT + 3) / 4, HALF = WS / 2, TW = 2 * WS - 1;
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) float X1[2][32 * AST];   // pass 0: k^ tile, pass 1: q^ tile (scaled)
    __shared__ __attribute__((aligned(16))) float X2[2][32 * AST];   // pass 0: v tile,  pass 1: dO tile
    __shared__ float ST[2][32][2];                                    // pass 1: {m + ln l, delta} of the walked queries
    // the head's relative-position bias table, (2 WS - 1)^2 floats.  Round 5: read from global memory inside `cond ? x : raw + table[..]` it was sixteen loads per
    // key tile, each under its own branch with its own s_waitcnt vmcnt(0) (tools/isa_loads.sh): sixteen dependent round trips per tile -- most of the kernel
    __shared__ float TB[TW * TW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * 32;
    const int nw = res / WS;
    int bid = blockIdx.x;
    const int part = bid % NOB;
    bid /= NOB;
    const int head = bid % heads;
    bid /= heads;
    const int wx = bid % nw;
    bid /= nw;
    const int wy = bid % nw;
    const int b = bid / nw;
    const int widx = (b * nw + wy) * nw + wx;
    const float sc = scale[head];
    auto token_row = [&](int p) -> size_t {
        const int r = p / WS, c = p % WS;
        int sy = wy * WS + r + shift, sx = wx * WS + c + shift;
        sy = sy >= res ? sy - res : sy;
        sx = sx >= res ? sx - res : sx;
        return (size_t)(b * res + sy) * res + sx;
    };
    const int r32 = lane & 31, h = lane >> 5;
    const int ob = part * 4 + wave;
    const bool active = ob < NT;              // wave-uniform
    const int orow = ob * 32 + r32;           // the lane's owned token
    const int ocl = orow < N ? orow : N - 1;
    const int ro = ocl / WS, co = ocl % WS;
    const bool or_hi = ro >= HALF, oc_hi = co >= HALF;
    const bool lastrow = (shift > 0) && (wy == nw - 1), lastcol = (shift > 0) && (wx == nw - 1);
    const size_t wh = (size_t)widx * heads + head;

    // ---- staging of a walked tile: thread -> (token tid >> 3, 16-byte chunk tid & 7) ----
    float4 r1, r2;
    float s0 = 0.f, s1 = 0.f;
    auto gload = [&](int t) {
        const int p = t * 32 + (tid >> 3), c = tid & 7;
        r1 = make_float4(0.f, 0.f, 0.f, 0.f);
        r2 = r1;
        s0 = 1.0e30f;   // a padded query contributes P = exp(S - 1e30) = 0
        s1 = 0.f;
        if (p < N) {
            const size_t row = token_row(p);
            const float* src = qkv + row * (size_t)(3 * C) + head * 32 + c * 4;
            if (PASS == 0) {
                r1 = *reinterpret_cast<const float4*>(src + C);
                r2 = *reinterpret_cast<const float4*>(src + 2 * C);
            } else {
                r1 = *reinterpret_cast<const float4*>(src);
                r2 = *reinterpret_cast<const float4*>(dO + row * (size_t)C + head * 32 + c * 4);
                if (c == 0) {
                    const float2 st = *reinterpret_cast<const float2*>(rowstat + (wh * N + p) * 2);
                    s0 = st.x;
                    s1 = st.y;
                }
            }
        }
        float ss = r1.x * r1.x + r1.y * r1.y + r1.z * r1.z + r1.w * r1.w;
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        const float inv = (PASS == 0 ? 1.0f : sc) / fmaxf(sqrtf(ss), 1e-12f);
        r1.x *= inv; r1.y *= inv; r1.z *= inv; r1.w *= inv;
    };
    auto lstore = [&](int buf) {
        const int kt = tid >> 3, c = tid & 7;
        *reinterpret_cast<float4*>(&X1[buf][kt * AST + c * 4]) = r1;
        *reinterpret_cast<float4*>(&X2[buf][kt * AST + c * 4]) = r2;
        if (PASS == 1 && c == 0) { ST[buf][kt][0] = s0; ST[buf][kt][1] = s1; }
    };

    // ---- owned fragments (MFMA B operand): lane (token r32, half h), step j <-> d = 16 h + j ----
    float f1[16], f2[16];
    float m_ln = 0.f, delta = 0.f;   // pass 0: the owned query's m + ln l and delta
    {
        const size_t row = token_row(ocl);
        const float* src = qkv + row * (size_t)(3 * C) + head * 32 + 16 * h + (PASS == 0 ? 0 : C);
        const float* src2 = PASS == 0 ? dO + row * (size_t)C + head * 32 + 16 * h : qkv + row * (size_t)(3 * C) + head * 32 + 16 * h + 2 * C;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(src + 4 * j);
            const float4 c4 = *reinterpret_cast<const float4*>(src2 + 4 * j);
            f1[4 * j] = a.x; f1[4 * j + 1] = a.y; f1[4 * j + 2] = a.z; f1[4 * j + 3] = a.w;
            f2[4 * j] = c4.x; f2[4 * j + 1] = c4.y; f2[4 * j + 2] = c4.z; f2[4 * j + 3] = c4.w;
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
        ss += __shfl_xor(ss, 32);
        const float inv = (PASS == 0 ? sc : 1.0f) / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int j = 0; j < 16; ++j) f1[j] *= inv;
        if (PASS == 0) {
            const float* orow_p = attn_out + row * (size_t)C + head * 32 + 16 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 o4 = *reinterpret_cast<const float4*>(orow_p + 4 * j);
                delta = fmaf(f2[4 * j], o4.x, delta);
                delta = fmaf(f2[4 * j + 1], o4.y, delta);
                delta = fmaf(f2[4 * j + 2], o4.z, delta);
                delta = fmaf(f2[4 * j + 3], o4.w, delta);
            }
            delta += __shfl_xor(delta, 32);
        }
    }

    h16x8 f2h_[2];   // OP != 0: the owned dO (pass 0) / v (pass 1) fragment rounded once; step ks, lane half h, element j <-> d = 16 h + 8 ks + j
    if constexpr (OP != 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f2h_[ks] = frag8<OP>(&f2[8 * ks]);
    }
    // walked index of accumulator register rg in tile t: t * 32 + (rg & 3) + 8 (rg >> 2) + 4 h
    // bias + shift mask of (query, key): table[((rq - rk + WS - 1) TW + (cq - ck + WS - 1)) heads + head]
    auto bias_of = [&](int w) -> float {
        const int wc = w < N ? w : N - 1;
        const int rw = wc / WS, cw = wc % WS;
        const int dr = PASS == 0 ? ro - rw : rw - ro, dc = PASS == 0 ? co - cw : cw - co;
        float v = TB[(dr + WS - 1) * TW + (dc + WS - 1)];
        if ((lastrow && ((rw >= HALF) != or_hi)) || (lastcol && ((cw >= HALF) != oc_hi))) v += -100.0f;
        return v;
    };
    auto raw_scores = [&](int buf) -> f32x16 {   // rows = walked tokens, column = owned token: scale * cos(q, k)
        f32x16 acc;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
        const float* xrow = &X1[buf][r32 * AST + 16 * h];
        // S = scale * cos(q, k) stays on the exact f32 MFMA in every mode: d logit_scale = sum over a row of dS * raw with sum dS = 0 -- rounding q^ / k^ to
        // 8 bits leaves a term that does not cancel (measured: that gradient 137 % off in stage 3 with bf16 raw scores, 5e-2 is the test's bound)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 k4 = *reinterpret_cast<const float4*>(xrow + 4 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, f1[4 * j], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, f1[4 * j + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, f1[4 * j + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, f1[4 * j + 3], acc, 0, 0, 0);
        }
        return acc;
    };

    f32x16 g1, g2;   // pass 0: g1 = dq^ (rows d, column query); pass 1: g1 = dk^, g2 = dv (rows d, column key)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) { g1[rg] = 0.f; g2[rg] = 0.f; }
    float m = -3.0e38f, l = 0.f, dsc = 0.f, dacc = 0.f;
    float* dSrow = dS_out + (wh * N + (size_t)ocl) * N;   // pass 0: the owned query's row of dS

    for (int i = tid; i < TW * TW; i += 256) TB[i] = table[(size_t)i * heads + head];
    gload(0);
    lstore(0);
    __syncthreads();
    constexpr int ITERS = PASS == 0 ? 2 * NT : NT;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        const int buf = it & 1;
        const int t = it < NT ? it : it - NT;
        const int tn = (it + 1) < NT ? it + 1 : it + 1 - NT;
        if (it + 1 < ITERS) gload(tn);
        if (active) {
            f32x16 raw = raw_scores(buf);
            f32x16 s;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int w = t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                const float bw = bias_of(w);   // index clamped inside: read unconditionally, select after
                s[rg] = (PASS == 0 && w >= N) ? -1.0e30f : raw[rg] + bw;
            }
            if (PASS == 0 && it < NT) {   // sweep 1: online softmax statistics of the owned query
                float mt = s[0];
#pragma unroll
                for (int rg = 1; rg < 16; ++rg) mt = fmaxf(mt, s[rg]);
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float mn = fmaxf(m, mt);
                float psum = 0.f;
                if constexpr (OP != 0) {
                    // 16-bit dP: delta = rowsum(P o dP) of THIS arithmetic (how softmax backward is defined, and what autograd of a 16-bit matmul gives),
                    // accumulated online beside l.  With the exact dO . O against a rounded dP the rows of dS no longer sum to zero, and d logit_scale =
                    // sum dS * raw picks up (row sum) * |raw|: that gradient came out 130 % off (tests/test_train_step_gpu.py bounds it at 5 %)
                    f32x16 dp1;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) dp1[rg] = 0.f;
                    const float* vrow = &X2[buf][r32 * AST + 16 * h];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float a8[8];
                        *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(vrow + 8 * ks);
                        *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(vrow + 8 * ks + 4);
                        dp1 = mfma_32x32x16<OP == 2>(frag8<OP>(a8), f2h_[ks], dp1);
                    }
                    float dsum = 0.f;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const float pe = __builtin_amdgcn_exp2f((s[rg] - mn) * LOG2E);
                        psum += pe;
                        dsum = fmaf(pe, dp1[rg], dsum);
                    }
                    const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
                    l = l * alpha + psum;
                    dacc = dacc * alpha + dsum;
                } else {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) psum += __builtin_amdgcn_exp2f((s[rg] - mn) * LOG2E);
                    l = l * __builtin_amdgcn_exp2f((m - mn) * LOG2E) + psum;
                }
                m = mn;
                if (it == NT - 1) {
                    l += __shfl_xor(l, 32);
                    m_ln = m + __logf(l);
                    if constexpr (OP != 0) {
                        dacc += __shfl_xor(dacc, 32);
                        delta = dacc / l;
                    }
                }
            } else {
                // dP^T = X2 rows . f2
                f32x16 dp;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) dp[rg] = 0.f;
                const float* vrow = &X2[buf][r32 * AST + 16 * h];
                if constexpr (OP != 0) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float a8[8];
                        *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(vrow + 8 * ks);
                        *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(vrow + 8 * ks + 4);
                        dp = mfma_32x32x16<OP == 2>(frag8<OP>(a8), f2h_[ks], dp);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 v4 = *reinterpret_cast<const float4*>(vrow + 4 * j);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.x, f2[4 * j], dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.y, f2[4 * j + 1], dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.z, f2[4 * j + 2], dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(v4.w, f2[4 * j + 3], dp, 0, 0, 0);
                    }
                }
                f32x16 p, ds;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int wl = (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    const float ml = PASS == 0 ? m_ln : ST[buf][wl][0];
                    const float dl = PASS == 0 ? delta : ST[buf][wl][1];
                    p[rg] = __builtin_amdgcn_exp2f((s[rg] - ml) * LOG2E);
                    ds[rg] = p[rg] * (dp[rg] - dl);
                }
                // accumulate over the walked tokens: A = column reads of the tiles (token (rg & 3) + 8 (rg >> 2) + 4 h, d = r32)
                const float* c1 = &X1[buf][(4 * h) * AST + r32];
                const float* c2 = &X2[buf][(4 * h) * AST + r32];
                if constexpr (OP != 0) {
                    // step st, lane half h, element j <-> walked token 16 st + 8 (j >> 2) + 4 h + (j & 3) = the token of accumulator register 8 st + j
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        float a8[8], b8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { a8[j] = c1[(16 * st + 8 * (j >> 2) + (j & 3)) * AST]; b8[j] = ds[8 * st + j]; }
                        g1 = mfma_32x32x16<OP == 2>(frag8<OP>(a8), frag8<OP>(b8), g1);
                        if (PASS == 1) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) { a8[j] = c2[(16 * st + 8 * (j >> 2) + (j & 3)) * AST]; b8[j] = p[8 * st + j]; }
                            g2 = mfma_32x32x16<OP == 2>(frag8<OP>(a8), frag8<OP>(b8), g2);
                        }
                    }
                } else {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int wl = (rg & 3) + 8 * (rg >> 2);
                        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[wl * AST], ds[rg], g1, 0, 0, 0);
                        if (PASS == 1) g2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c2[wl * AST], p[rg], g2, 0, 0, 0);
                    }
                }
                if (PASS == 0) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) dsc = fmaf(ds[rg], raw[rg], dsc);   // raw = scale * (qn . k^): divided by scale below
                    if (orow < N) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int w0 = t * 32 + 8 * g + 4 * h;
                            if (N % 4 == 0) {
                                if (w0 < N) *reinterpret_cast<float4*>(dSrow + w0) = make_float4(ds[4 * g], ds[4 * g + 1], ds[4 * g + 2], ds[4 * g + 3]);
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (w0 + i < N) dSrow[w0 + i] = ds[4 * g + i];
                            }
                        }
                    }
                }
            }
        }
        if (it + 1 < ITERS) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: through the L2 normalisation.  Accumulator rows: d = 8 g + 4 h + (0..3) for g = rg >> 2 ----
    if (PASS == 0) {
        // per-wave partial of d scale (tr_attn_param_grads reduces NOB * 4 slots per (window, head) in fixed order); inactive waves write 0
        float v = (active && orow < N) ? dsc / sc : 0.f;
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        if (lane == 0) dscale_part[wh * (NOB * 4) + part * 4 + wave] = v;
    }
    if (!active) return;
    const size_t row = token_row(ocl);
    const float* src = qkv + row * (size_t)(3 * C) + head * 32 + (PASS == 0 ? 0 : C);
    float4 x[4];
    float ss = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        x[g] = *reinterpret_cast<const float4*>(src + 8 * g + 4 * h);
        ss += x[g].x * x[g].x + x[g].y * x[g].y + x[g].z * x[g].z + x[g].w * x[g].w;
    }
    ss += __shfl_xor(ss, 32);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    const float gs = PASS == 0 ? sc : 1.0f;   // pass 0: dqn = scale * dq^ ; pass 1: dk^ was accumulated against the scaled q^ already
    float dot = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        x[g].x /= nrm; x[g].y /= nrm; x[g].z /= nrm; x[g].w /= nrm;
        dot = fmaf(x[g].x, g1[4 * g] * gs, dot);
        dot = fmaf(x[g].y, g1[4 * g + 1] * gs, dot);
        dot = fmaf(x[g].z, g1[4 * g + 2] * gs, dot);
        dot = fmaf(x[g].w, g1[4 * g + 3] * gs, dot);
    }
    dot += __shfl_xor(dot, 32);
    if (orow < N) {
        float* dst = dqkv + row * (size_t)(3 * C) + head * 32 + (PASS == 0 ? 0 : C);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o;
            o.x = (g1[4 * g] * gs - x[g].x * dot) / nrm;
            o.y = (g1[4 * g + 1] * gs - x[g].y * dot) / nrm;
            o.z = (g1[4 * g + 2] * gs - x[g].z * dot) / nrm;
            o.w = (g1[4 * g + 3] * gs - x[g].w * dot) / nrm;
            *reinterpret_cast<float4*>(dst + 8 * g + 4 * h) = o;
            if (PASS == 1) *reinterpret_cast<float4*>(dst + C + 8 * g + 4 * h) = make_float4(g2[4 * g], g2[4 * g + 1], g2[4 * g + 2], g2[4 * g + 3]);
        }
        if (PASS == 0 && h == 0) *reinterpret_cast<float2*>(rowstat + (wh * N + orow) * 2) = make_float2(m_ln, delta);
    }
}


// ---------------------------------------------------------------------------------------------
// Global softmax attention backward of the ViT-hybrid encoder (timm vision_transformer.Attention: P = softmax(q k^T / 8), head dimension 64, 577
// tokens): the same two streaming passes without normalisation, bias or mask.  qkv [B*N][3E] (q | k | v, E = heads * 64), dqkv alike.
// rowstat [B][heads][N][2] = {m + ln l, delta} (pass 0 -> pass 1; the forward's statistics are not needed).
// ---------------------------------------------------------------------------------------------
constexpr int VST = 68;   // floats per LDS row (272 B = 17 x 16)

template <int PASS, int OP = 0>   // OP as in attn_bwd_mfma_kernel: 16-bit operands for dP, dq / dk and dv in the amp modes; the scores stay exact f32
__global__ __launch_bounds__(256) void vit_attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ attn_out,
                                                                float* __restrict__ rowstat, float* __restrict__ dqkv, int N, int heads) {
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) float X1[2][32 * VST];   // pass 0: k tile, pass 1: q / 8 tile
    __shared__ __attribute__((aligned(16))) float X2[2][32 * VST];   // pass 0: v tile, pass 1: dO tile
    __shared__ float ST[2][32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = heads * 64, NT = (N + 31) / 32, NOB = (NT + 3) / 4;
    int bid = blockIdx.x;
    const int part = bid % NOB;
    bid /= NOB;
    const int head = bid % heads;
    const int b = bid / heads;
    const int r32 = lane & 31, h = lane >> 5;
    const int ob = part * 4 + wave;
    const bool active = ob < NT;
    const int orow = ob * 32 + r32;
    const int ocl = orow < N ? orow : N - 1;
    const size_t bh = (size_t)b * heads + head;

    // staging: thread -> (token tid >> 3, two 16-byte chunks c and c + 8 of the 64-float row)
    float4 r1a, r1b, r2a, r2b;
    float s0 = 0.f, s1 = 0.f;
    auto gload = [&](int t) {
        const int p = t * 32 + (tid >> 3), c = tid & 7;
        r1a = make_float4(0.f, 0.f, 0.f, 0.f);
        r1b = r1a; r2a = r1a; r2b = r1a;
        s0 = 1.0e30f;
        s1 = 0.f;
        if (p < N) {
            const size_t row = (size_t)b * N + p;
            const float* src = qkv + row * (size_t)(3 * E) + head * 64 + c * 4;
            if (PASS == 0) {
                r1a = *reinterpret_cast<const float4*>(src + E);
                r1b = *reinterpret_cast<const float4*>(src + E + 32);
                r2a = *reinterpret_cast<const float4*>(src + 2 * E);
                r2b = *reinterpret_cast<const float4*>(src + 2 * E + 32);
            } else {
                r1a = *reinterpret_cast<const float4*>(src);
                r1b = *reinterpret_cast<const float4*>(src + 32);
                r1a.x *= 0.125f; r1a.y *= 0.125f; r1a.z *= 0.125f; r1a.w *= 0.125f;
                r1b.x *= 0.125f; r1b.y *= 0.125f; r1b.z *= 0.125f; r1b.w *= 0.125f;
                const float* dsrc = dO + row * (size_t)E + head * 64 + c * 4;
                r2a = *reinterpret_cast<const float4*>(dsrc);
                r2b = *reinterpret_cast<const float4*>(dsrc + 32);
                if (c == 0) {
                    const float2 st = *reinterpret_cast<const float2*>(rowstat + (bh * N + p) * 2);
                    s0 = st.x;
                    s1 = st.y;
                }
            }
        }
    };
    auto lstore = [&](int buf) {
        const int kt = tid >> 3, c = tid & 7;
        *reinterpret_cast<float4*>(&X1[buf][kt * VST + c * 4]) = r1a;
        *reinterpret_cast<float4*>(&X1[buf][kt * VST + 32 + c * 4]) = r1b;
        *reinterpret_cast<float4*>(&X2[buf][kt * VST + c * 4]) = r2a;
        *reinterpret_cast<float4*>(&X2[buf][kt * VST + 32 + c * 4]) = r2b;
        if (PASS == 1 && c == 0) { ST[buf][kt][0] = s0; ST[buf][kt][1] = s1; }
    };

    // owned fragments (B operand): lane (token r32, half h), step j <-> d = 32 h + j
    float f1[32], f2[32];
    float m_ln = 0.f, delta = 0.f;
    {
        const size_t row = (size_t)b * N + ocl;
        const float* src = qkv + row * (size_t)(3 * E) + head * 64 + 32 * h + (PASS == 0 ? 0 : E);
        const float* src2 = PASS == 0 ? dO + row * (size_t)E + head * 64 + 32 * h : qkv + row * (size_t)(3 * E) + head * 64 + 32 * h + 2 * E;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(src + 4 * j);
            const float4 c4 = *reinterpret_cast<const float4*>(src2 + 4 * j);
            const float sc = PASS == 0 ? 0.125f : 1.0f;
            f1[4 * j] = a.x * sc; f1[4 * j + 1] = a.y * sc; f1[4 * j + 2] = a.z * sc; f1[4 * j + 3] = a.w * sc;
            f2[4 * j] = c4.x; f2[4 * j + 1] = c4.y; f2[4 * j + 2] = c4.z; f2[4 * j + 3] = c4.w;
        }
        if (PASS == 0) {
            const float* op = attn_out + row * (size_t)E + head * 64 + 32 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 o4 = *reinterpret_cast<const float4*>(op + 4 * j);
                delta = fmaf(f2[4 * j], o4.x, delta);
                delta = fmaf(f2[4 * j + 1], o4.y, delta);
                delta = fmaf(f2[4 * j + 2], o4.z, delta);
                delta = fmaf(f2[4 * j + 3], o4.w, delta);
            }
            delta += __shfl_xor(delta, 32);
        }
    }
    h16x8 f2h_[4];   // OP != 0: the owned dO (pass 0) / v (pass 1) fragment rounded once; step ks, lane half h, element j <-> d = 32 h + 8 ks + j
    if constexpr (OP != 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f2h_[ks] = frag8<OP>(&f2[8 * ks]);
    }
    auto rows_dot16 = [&](const float* X) -> f32x16 {   // X rows . the owned f2 fragment on 16-bit MFMAs
        f32x16 acc;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
        const float* xrow = X + r32 * VST + 32 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a8[8];
            *reinterpret_cast<float4*>(a8) = *reinterpret_cast<const float4*>(xrow + 8 * ks);
            *reinterpret_cast<float4*>(a8 + 4) = *reinterpret_cast<const float4*>(xrow + 8 * ks + 4);
            acc = mfma_32x32x16<OP == 2>(frag8<OP == 0 ? 1 : OP>(a8), f2h_[ks], acc);
        }
        return acc;
    };
    auto rows_dot = [&](const float* X, const float* f) -> f32x16 {   // rows = walked tokens of the tile, column = owned token
        f32x16 acc;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
        const float* xrow = X + r32 * VST + 32 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 k4 = *reinterpret_cast<const float4*>(xrow + 4 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, f[4 * j], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, f[4 * j + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, f[4 * j + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, f[4 * j + 3], acc, 0, 0, 0);
        }
        return acc;
    };

    f32x16 g1a, g1b, g2a, g2b;   // rows d (a: 0..31, b: 32..63), column owned token
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) { g1a[rg] = 0.f; g1b[rg] = 0.f; g2a[rg] = 0.f; g2b[rg] = 0.f; }
    float m = -3.0e38f, l = 0.f, dacc = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    const int ITERS = PASS == 0 ? 2 * NT : NT;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        const int buf = it & 1;
        const int t = it < NT ? it : it - NT;
        const int tn = (it + 1) < NT ? it + 1 : it + 1 - NT;
        if (it + 1 < ITERS) gload(tn);
        if (active) {
            f32x16 s = rows_dot(X1[buf], f1);
            if (PASS == 0) {
#pragma unroll
                for (int rg = 0; rg < 16; ++rg)
                    if (t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h >= N) s[rg] = -1.0e30f;
            }
            if (PASS == 0 && it < NT) {
                float mt = s[0];
#pragma unroll
                for (int rg = 1; rg < 16; ++rg) mt = fmaxf(mt, s[rg]);
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float mn = fmaxf(m, mt);
                float psum = 0.f;
                if constexpr (OP != 0) {   // delta = rowsum(P o dP) of the 16-bit dP, accumulated online (see attn_bwd_mfma_kernel)
                    const f32x16 dp1 = rows_dot16(X2[buf]);
                    float dsum = 0.f;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const float pe = __builtin_amdgcn_exp2f((s[rg] - mn) * LOG2E);
                        psum += pe;
                        dsum = fmaf(pe, dp1[rg], dsum);
                    }
                    const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
                    l = l * alpha + psum;
                    dacc = dacc * alpha + dsum;
                } else {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) psum += __builtin_amdgcn_exp2f((s[rg] - mn) * LOG2E);
                    l = l * __builtin_amdgcn_exp2f((m - mn) * LOG2E) + psum;
                }
                m = mn;
                if (it == NT - 1) {
                    l += __shfl_xor(l, 32);
                    m_ln = m + __logf(l);
                    if constexpr (OP != 0) {
                        dacc += __shfl_xor(dacc, 32);
                        delta = dacc / l;
                    }
                }
            } else {
                f32x16 dp;
                if constexpr (OP != 0) dp = rows_dot16(X2[buf]);
                else dp = rows_dot(X2[buf], f2);
                f32x16 p, ds;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int wl = (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    const float ml = PASS == 0 ? m_ln : ST[buf][wl][0];
                    const float dl = PASS == 0 ? delta : ST[buf][wl][1];
                    p[rg] = __builtin_amdgcn_exp2f((s[rg] - ml) * LOG2E);
                    ds[rg] = p[rg] * (dp[rg] - dl);
                }
                const float* c1 = &X1[buf][(4 * h) * VST + r32];
                const float* c2 = &X2[buf][(4 * h) * VST + r32];
                if constexpr (OP != 0) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {   // step st, lane half h, element j <-> walked token 16 st + 8 (j >> 2) + 4 h + (j & 3) = accumulator register 8 st + j
                        float a8[8], b8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) b8[j] = ds[8 * st + j];
                        const h16x8 dsf = frag8<OP>(b8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) a8[j] = c1[(16 * st + 8 * (j >> 2) + (j & 3)) * VST];
                        g1a = mfma_32x32x16<OP == 2>(frag8<OP>(a8), dsf, g1a);
#pragma unroll
                        for (int j = 0; j < 8; ++j) a8[j] = c1[(16 * st + 8 * (j >> 2) + (j & 3)) * VST + 32];
                        g1b = mfma_32x32x16<OP == 2>(frag8<OP>(a8), dsf, g1b);
                        if (PASS == 1) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) b8[j] = p[8 * st + j];
                            const h16x8 pf = frag8<OP>(b8);
#pragma unroll
                            for (int j = 0; j < 8; ++j) a8[j] = c2[(16 * st + 8 * (j >> 2) + (j & 3)) * VST];
                            g2a = mfma_32x32x16<OP == 2>(frag8<OP>(a8), pf, g2a);
#pragma unroll
                            for (int j = 0; j < 8; ++j) a8[j] = c2[(16 * st + 8 * (j >> 2) + (j & 3)) * VST + 32];
                            g2b = mfma_32x32x16<OP == 2>(frag8<OP>(a8), pf, g2b);
                        }
                    }
                } else {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int wl = (rg & 3) + 8 * (rg >> 2);
                        g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[wl * VST], ds[rg], g1a, 0, 0, 0);
                        g1b = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[wl * VST + 32], ds[rg], g1b, 0, 0, 0);
                        if (PASS == 1) {
                            g2a = __builtin_amdgcn_mfma_f32_32x32x2f32(c2[wl * VST], p[rg], g2a, 0, 0, 0);
                            g2b = __builtin_amdgcn_mfma_f32_32x32x2f32(c2[wl * VST + 32], p[rg], g2b, 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (it + 1 < ITERS) lstore(buf ^ 1);
        __syncthreads();
    }
    if (!active || orow >= N) return;
    const size_t row = (size_t)b * N + orow;
    float* dst = dqkv + row * (size_t)(3 * E) + head * 64 + (PASS == 0 ? 0 : E);
    const float gs = PASS == 0 ? 0.125f : 1.0f;   // pass 0: S = (q / 8) . k; pass 1 accumulated against q / 8 already
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dst + 8 * g + 4 * h) = make_float4(g1a[4 * g] * gs, g1a[4 * g + 1] * gs, g1a[4 * g + 2] * gs, g1a[4 * g + 3] * gs);
        *reinterpret_cast<float4*>(dst + 32 + 8 * g + 4 * h) = make_float4(g1b[4 * g] * gs, g1b[4 * g + 1] * gs, g1b[4 * g + 2] * gs, g1b[4 * g + 3] * gs);
        if (PASS == 1) {
            *reinterpret_cast<float4*>(dst + E + 8 * g + 4 * h) = make_float4(g2a[4 * g], g2a[4 * g + 1], g2a[4 * g + 2], g2a[4 * g + 3]);
            *reinterpret_cast<float4*>(dst + E + 32 + 8 * g + 4 * h) = make_float4(g2b[4 * g], g2b[4 * g + 1], g2b[4 * g + 2], g2b[4 * g + 3]);
        }
    }
    if (PASS == 0 && h == 0) *reinterpret_cast<float2*>(rowstat + (bh * N + orow) * 2) = make_float2(m_ln, delta);
}

template <int WS>
int launch_ws(const float* qkv, const float* attn_out, const float* dO, const float* table, const float* scale, float* dS, float* rowstat, float* dscale_part,
              float* dqkv, int B, int res, int shift, int heads, hipStream_t st, int op) {
    constexpr int N = WS * WS, NT = (N + 31) / 32, NOB = (NT + 3) / 4;
    const int nw = res / WS;
    const unsigned blocks = (unsigned)(B * nw * nw * heads * NOB);
#define ATTN_BWD(P, O) SOCCDPT_LAUNCH((attn_bwd_mfma_kernel<WS, P, O>), dim3(blocks), dim3(256), 0, st, qkv, dO, attn_out, table, scale, rowstat, dS, dscale_part, dqkv, res, shift, heads)
    if (op == 1) { ATTN_BWD(0, 1); ATTN_BWD(1, 1); }
    else if (op == 2) { ATTN_BWD(0, 2); ATTN_BWD(1, 2); }
    else { ATTN_BWD(0, 0); ATTN_BWD(1, 0); }
#undef ATTN_BWD
    return 0;
}

}  // namespace

// d scale partial slots per (window, head) written by the MFMA form (0 when the window size has no MFMA instantiation)
int tr_attention_bwd_mfma_slots(int ws) {
    if (ws != 8 && ws != 16 && ws != 12 && ws != 24) return 0;
    const int NT = (ws * ws + 31) / 32;
    return (NT + 3) / 4 * 4;
}

// Same contract as tr_attention_bwd (train.hip) without its `part` scratch: dqkv [B*res*res][3C] receives dq | dk | dv, dS [nwin][heads][N][N],
// rowstat [nwin][heads][N][2] = {m + ln l, delta}, dscale_part [nwin][heads][tr_attention_bwd_mfma_slots(ws)].
int tr_attention_bwd_mfma(const float* qkv, const float* attn_out, const float* dO, const float* table, const float* scale, float* dS, float* rowstat,
                          float* dscale_part, float* dqkv, int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err, int op) {
    if (res % ws) { err = "attention_bwd_mfma: res must be a multiple of the window size"; return 1; }
    switch (ws) {
        case 8: launch_ws<8>(qkv, attn_out, dO, table, scale, dS, rowstat, dscale_part, dqkv, B, res, shift, heads, st, op); break;
        case 16: launch_ws<16>(qkv, attn_out, dO, table, scale, dS, rowstat, dscale_part, dqkv, B, res, shift, heads, st, op); break;
        case 12: launch_ws<12>(qkv, attn_out, dO, table, scale, dS, rowstat, dscale_part, dqkv, B, res, shift, heads, st, op); break;
        case 24: launch_ws<24>(qkv, attn_out, dO, table, scale, dS, rowstat, dscale_part, dqkv, B, res, shift, heads, st, op); break;
        default: err = "attention_bwd_mfma: no instantiation for this window size"; return 1;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("attention_bwd_mfma: ") + hipGetErrorString(e); return 1; }
    return 0;
}


// ViT form: qkv [B*N][3E], O / dO [B*N][E], rowstat B * heads * N * 2 floats of scratch, dqkv [B*N][3E]
int th_vit_attention_bwd_mfma(const float* qkv, const float* O, const float* dO, float* rowstat, float* dqkv, int B, int N, int heads, hipStream_t st, std::string& err, int op) {
    const int NT = (N + 31) / 32, NOB = (NT + 3) / 4;
    const unsigned blocks = (unsigned)(B * heads * NOB);
#define VIT_BWD(P, OPV) SOCCDPT_LAUNCH((vit_attn_bwd_mfma_kernel<P, OPV>), dim3(blocks), dim3(256), 0, st, qkv, dO, O, rowstat, dqkv, N, heads)
    if (op == 1) { VIT_BWD(0, 1); VIT_BWD(1, 1); }
    else if (op == 2) { VIT_BWD(0, 2); VIT_BWD(1, 2); }
    else { VIT_BWD(0, 0); VIT_BWD(1, 0); }
#undef VIT_BWD
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("vit_attention_bwd_mfma: ") + hipGetErrorString(e); return 1; }
    return 0;
}

}  // namespace soccdpt
