// Global multi-head softmax attention of a ViT block for gfx950 (dpt_hybrid_384: 12 heads x 64, 24*24 + 1 = 577 tokens).
//
// Replaces timm 0.6.12 vision_transformer.Attention.forward between the qkv and proj Linear layers (created by
// /root/reference/SOccDPT/model/backbones/vit.py:248, run from forward_flex vit.py:79-80; restated in oracle/soccdpt_ref.py vit_block):
//     attn = softmax(q k^T * d^-0.5);  out = attn v          per (sample, head)
// Design (CDNA4), the same decomposition as the Swin-V2 online-softmax kernel (attention.hip) with d = 64 and no bias:
//  * one workgroup per (sample, head, query part): K (row-major, 16-byte chunks XOR-swizzled -> conflict-free ds_read_b128) and V^T
//    ([d][token], 8-byte row pad) of the whole sequence are staged ONCE in LDS (2 x 78 KB of the CU's 160 KB), each of the 4 waves then
//    walks the key tiles for its own 32-query block: Q stays in registers (read straight from HBM, the 2^-3 softmax scale folded in
//    exactly), nothing is re-staged per tile and no barrier sits in the main loop;
//  * S^T = K Q^T with v_mfma_f32_32x32x16 ("swapped" product): a lane owns one query column, so the running max / sum are lane-local
//    plus one exchange with the other half-wave, and the exponentiated accumulator is directly the B operand of O^T = V^T P^T;
//  * the sequence length is prime (577): the last key tile is masked arithmetically, padded V rows are zero, padded query rows are
//    computed and not stored.
#include "../../include/soccdpt_hip.h"
#include "half16.h"
#include "kernels.h"

namespace soccdpt {
namespace {

typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int D = 64;            // head dimension
constexpr int KROW = D * 2;      // bytes per K row in LDS

constexpr int KH = 2;             // key parts per query block = waves per SIMD (per launch at B = 4: 4 waves 25.8 us, 8 waves 18.5, 16 waves 20.0)
template <bool F16>
__global__ __launch_bounds__(256 * KH) void vit_attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int N, int NPAD, int heads, int QS, int out_x3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VT_STRIDE = NPAD * 2 + 8;
    char* Ks = smem;
    char* Vt = smem + (size_t)NPAD * KROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * D;
    int bid = blockIdx.x;
    const int part = bid % QS;
    bid /= QS;
    const int head = bid % heads;
    const int b = bid / heads;
    const int NT = NPAD / 32;
    const int QB0 = (NT + QS - 1) / QS;
    const int qb_lo = part * QB0, qb_hi = (qb_lo + QB0) < NT ? (qb_lo + QB0) : NT;
    const uint16_t* base = qkv + (size_t)b * N * (3 * C) + head * D;

    // ---- stage K and V^T of the whole sequence: thread = (token, 16-byte chunk of 8 d).  SB items of a thread are loaded before the first is written to
    // LDS: one load-then-store per trip made the 19 trips 19 dependent L2 round trips, (round 5: 25.7 -> 24.3 us per launch) ----
    constexpr int SB = KH == 4 ? 3 : 5;
    for (int idx0 = tid; idx0 < NPAD * 8; idx0 += 256 * KH * SB) {
        uint4 kv[SB], vv[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = idx0 + u * 256 * KH, p = idx >> 3, c = idx & 7;
            kv[u] = make_uint4(0u, 0u, 0u, 0u); vv[u] = kv[u];
            if (idx < NPAD * 8 && p < N) {
                const uint16_t* src = base + (size_t)p * (3 * C) + c * 8;
                kv[u] = *reinterpret_cast<const uint4*>(src + C);
                vv[u] = *reinterpret_cast<const uint4*>(src + 2 * C);
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = idx0 + u * 256 * KH, p = idx >> 3, c = idx & 7;
            if (idx >= NPAD * 8) break;
            *reinterpret_cast<uint4*>(Ks + p * KROW + ((c ^ (p & 7)) * 16)) = kv[u];
            const uint32_t vu[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j) * VT_STRIDE + p * 2) = (uint16_t)(vu[j] & 0xffffu);
                *reinterpret_cast<uint16_t*>(Vt + (c * 8 + 2 * j + 1) * VT_STRIDE + p * 2) = (uint16_t)(vu[j] >> 16);
            }
        }
    }
    __syncthreads();

    const int r32 = lane & 31, h = lane >> 5;
    constexpr float LOG2E = 1.4426950408889634f;
    // 8 waves: wave (qslot, kh) walks key half kh for the 32-query block qb_lo + qslot (host: at most 4 blocks per workgroup); the halves meet in LDS.
    // Round 5: with 4 waves -- one per SIMD -- the softmax VALU work (~850 cycles per key tile) and the MFMAs + LDS reads (~450) of a wave ran strictly
    // one after the other and the CU held nothing else (148 KB of LDS): two waves per SIMD overlap them.
    const int qslot = wave & 3, kh = wave >> 2;
    const int NTP = (NT + KH - 1) / KH, t_lo = kh * NTP, t_hi = (t_lo + NTP) < NT ? (t_lo + NTP) : NT;
    const int qb = qb_lo + qslot;
    const bool active = qb < qb_hi;
    float m = -3.0e38f, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) { o0[rg] = 0.f; o1[rg] = 0.f; }
    const int qrow = qb * 32 + r32;
    if (active) {
        const int qcl = qrow < N ? qrow : N - 1;
        // Q fragment: B operand, lane (query r32, half h) holds d = 16 ks + 8 h + j; scaled by d^-0.5 = 2^-3 (exact in bf16 / fp16)
        h16x8 qfrag[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 q4 = *reinterpret_cast<const uint4*>(base + (size_t)qcl * (3 * C) + ks * 16 + 8 * h);
            const uint32_t qu[4] = {q4.x, q4.y, q4.z, q4.w};
            h16x8 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[2 * j] = (short)f2h<F16>(h_lo<F16>(qu[j]) * 0.125f);
                f[2 * j + 1] = (short)f2h<F16>(h_hi<F16>(qu[j]) * 0.125f);
            }
            qfrag[ks] = f;
        }
#pragma unroll 1
        for (int t = t_lo; t < t_hi; ++t) {
            f32x16 acc;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
            const int krow = t * 32 + r32;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8 kfrag = *reinterpret_cast<const h16x8*>(Ks + krow * KROW + (((ks * 2 + h) ^ (krow & 7)) * 16));
                acc = mfma_32x32x16<F16>(kfrag, qfrag[ks], acc);
            }
            if (t == NT - 1) {   // keys beyond the sequence: excluded from max, sum and P V
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int key = t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    if (key >= N) acc[rg] = -3.0e38f;
                }
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
            }
            if (__any(alpha != 1.0f)) {   // the running maxima settle after a few tiles: most tiles skip the 32 multiplications (x 1.0 is exact, so skipping is too)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) { o0[rg] *= alpha; o1[rg] *= alpha; }
            }
            l = l * alpha + psum;
            m = mn;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                h16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (short)f2h_inrange<F16>(acc[8 * st + j]);   // probabilities in [0, 1]: no clamp
                // element j of this lane half is key 32t + 16st + 8(j>>2) + 4h + (j&3): V^T is read in the same k order
                const int koff = (t * 32 + st * 16 + 4 * h) * 2;
                {
                    const char* vrow = Vt + r32 * VT_STRIDE + koff;
                    const h16x4 v0 = *reinterpret_cast<const h16x4*>(vrow);
                    const h16x4 v1 = *reinterpret_cast<const h16x4*>(vrow + 16);
                    const h16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o0 = mfma_32x32x16<F16>(vf, pb, o0);
                }
                {
                    const char* vrow = Vt + (32 + r32) * VT_STRIDE + koff;
                    const h16x4 v0 = *reinterpret_cast<const h16x4*>(vrow);
                    const h16x4 v1 = *reinterpret_cast<const h16x4*>(vrow + 16);
                    const h16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o1 = mfma_32x32x16<F16>(vf, pb, o1);
                }
            }
        }
        l += __shfl_xor(l, 32);
    }
    // ---- the two key halves meet: kh = 1 parks (O^T, m, l) in LDS (K / V^T are dead: one query block per wave), kh = 0 adds them in with the usual rescale ----
    __syncthreads();
    float* xb = reinterpret_cast<float*>(smem) + (size_t)((kh ? kh - 1 : 0) * 4 + qslot) * (34 * 64);   // [34][64]: register-major, lane-contiguous
    if (kh >= 1 && active) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) { xb[rg * 64 + lane] = o0[rg]; xb[(16 + rg) * 64 + lane] = o1[rg]; }
        xb[32 * 64 + lane] = m;
        xb[33 * 64 + lane] = l;
    }
    __syncthreads();
    if (kh == 0 && active) {
#pragma unroll 1
        for (int pk = 1; pk < KH; ++pk) {   // the other parts in part order (an empty part: m = -3e38, l = 0, O = 0 -> weight 0)
            const float* xp = xb + (size_t)(pk - 1) * 4 * (34 * 64);
            const float m1 = xp[32 * 64 + lane], l1 = xp[33 * 64 + lane];
            const float mn = fmaxf(m, m1), mnl = mn * LOG2E;
            const float a0 = __builtin_amdgcn_exp2f(fmaf(m, LOG2E, -mnl)), a1 = __builtin_amdgcn_exp2f(fmaf(m1, LOG2E, -mnl));
            l = l * a0 + l1 * a1;
            m = mn;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                o0[rg] = o0[rg] * a0 + xp[rg * 64 + lane] * a1;
                o1[rg] = o1[rg] * a0 + xp[(16 + rg) * 64 + lane] * a1;
            }
        }
        if (qrow < N) {   // lane owns query column r32; accumulator register rg is d = (rg&3) + 8(rg>>2) + 4h (+32 for o1)
            const float inv = 1.0f / l;
            const size_t e0 = ((size_t)b * N + qrow) * C + head * D;
            uint16_t* orow = out + e0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (F16 && out_x3) {   // SOCCDPT_PREC_MIXED: the proj GEMM of this block is an x3 launch (half16.h); the f32 accumulators go out unrounded
                    x3_store4(out, e0 + 8 * g + 4 * h, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                    x3_store4(out, e0 + 32 + 8 * g + 4 * h, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
                    continue;
                }
                uint2 p0, p1;
                p0.x = pack_h2<F16>(o0[4 * g] * inv, o0[4 * g + 1] * inv);
                p0.y = pack_h2<F16>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                p1.x = pack_h2<F16>(o1[4 * g] * inv, o1[4 * g + 1] * inv);
                p1.y = pack_h2<F16>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + 8 * g + 4 * h) = p0;
                *reinterpret_cast<uint2*>(orow + 32 + 8 * g + 4 * h) = p1;
            }
        }
    }
}

// Exact-f32 variant (SOCCDPT_PREC_F32, and the attention of SOCCDPT_PREC_F16X3): the same swapped-product decomposition on
// v_mfma_f32_32x32x2_f32 (bitwise an f32 fma chain at the f32 VALU peak rate, all of it on the matrix pipe).
//  * one workgroup = 4 waves = 4 blocks of 32 queries of one (sample, head); key / value tiles of 32 tokens stream through a
//    double-buffered LDS ring (row-major, K rows padded to 68 floats: conflict-free ds_read_b128 of a lane's 32 consecutive d),
//    the next tile's global loads are issued before the current tile's MFMAs and written to LDS after them: one barrier per tile;
//  * S^T = K Q^T: the MFMA's two k-slots (lane halves h = 0 / 1) take d = st and d = 32 + st at step st -- any pairing of the 64
//    head dimensions is a valid summation order as long as both operands use it -- so a lane's Q and K fragments are 32 consecutive
//    floats (Q: eight 16-byte loads straight from HBM, scaled by 2^-3 exactly);
//  * the exponentiated S^T accumulator is directly the B operand of O^T = V^T P^T: accumulator register r of the two lane halves holds
//    keys 32t + (r&3) + 8(r>>2) + 4h, exactly one K = 2 MFMA step, whose A operand V[key][d = lane&31 (+32)] is a conflict-free
//    ds_read_b32 of the row-major tile (all 32 lanes of a half read one row).
// 64 MFMAs x 64 cycles per (32 query x 32 key) tile per wave; 19 x 19 tiles per (sample, head) at N = 577.
// OUT: 2 = plain f32 output, 3 = x3 split-fp16 output (half16.h) for the proj GEMM of the F16X3 mode.
constexpr int KS_STRIDE = 68;   // floats per K row in LDS (272 B = 17 x 16: the 16-byte slot of column c in row r is (r + c) mod 16)

template <int OUT>
__global__ __launch_bounds__(256) void vit_attention_f32_kernel(const float* __restrict__ qkv, void* __restrict__ out, int N, int heads, int NQB) {
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = heads * D;
    int bid = blockIdx.x;
    {   // the NQB query parts of one (sample, head) share K / V: consecutive logical ids -> one XCD (blocks b, b + 8 share an L2)
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int part = bid % NQB;
    bid /= NQB;
    const int head = bid % heads;
    const int b = bid / heads;
    const float* base = qkv + (size_t)b * N * (3 * C) + head * D;
    const int NT = (N + 31) / 32;
    const int r32 = lane & 31, h = lane >> 5;
    const int qb = part * 4 + wave;
    const int qrow = qb * 32 + r32;
    const bool active = qb * 32 < N;           // wave-uniform: this wave owns at least one real query
    const int qcl = qrow < N ? qrow : N - 1;

    // staging map: thread -> (token of the tile = idx >> 4, 16-byte chunk c = idx & 15), idx = tid and tid + 256
    float4 kr[2], vr[2];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int p = t * 32 + (idx >> 4), c = idx & 15;
            kr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            vr[i] = kr[i];
            if (p < N) {
                const float* src = base + (size_t)p * (3 * C) + c * 4;
                kr[i] = *reinterpret_cast<const float4*>(src + C);
                vr[i] = *reinterpret_cast<const float4*>(src + 2 * C);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int kt = idx >> 4, c = idx & 15;
            *reinterpret_cast<float4*>(&Ks[buf][kt * KS_STRIDE + c * 4]) = kr[i];
            *reinterpret_cast<float4*>(&Vs[buf][kt * D + c * 4]) = vr[i];
        }
    };

    // Q fragment (B operand): lane (query r32, half h), step st <-> d = 32 h + st
    float qf[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 q4 = *reinterpret_cast<const float4*>(base + (size_t)qcl * (3 * C) + 32 * h + 4 * j);
        qf[4 * j] = q4.x * 0.125f; qf[4 * j + 1] = q4.y * 0.125f; qf[4 * j + 2] = q4.z * 0.125f; qf[4 * j + 3] = q4.w * 0.125f;
    }
    gload(0);
    lstore(0);
    __syncthreads();

    constexpr float LOG2E = 1.4426950408889634f;
    float m = -3.0e38f, l = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) { o0[rg] = 0.f; o1[rg] = 0.f; }
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        if (t + 1 < NT) gload(t + 1);
        if (active) {
            f32x16 acc;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) acc[rg] = 0.f;
            const float* krow = &Ks[buf][r32 * KS_STRIDE + 32 * h];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 k4 = *reinterpret_cast<const float4*>(krow + 4 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[4 * j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[4 * j + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[4 * j + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[4 * j + 3], acc, 0, 0, 0);
            }
            if (t == NT - 1) {   // keys beyond the sequence: excluded from max, sum and P V
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int key = t * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * h;
                    if (key >= N) acc[rg] = -3.0e38f;
                }
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
            }
            if (__any(alpha != 1.0f)) {   // the running maxima settle after a few tiles: most tiles skip the 32 multiplications (x 1.0 is exact, so skipping is too)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) { o0[rg] *= alpha; o1[rg] *= alpha; }
            }
            l = l * alpha + psum;
            m = mn;
            const float* vcol = &Vs[buf][(4 * h) * D + r32];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int kk = (rg & 3) + 8 * (rg >> 2);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vcol[kk * D], acc[rg], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vcol[kk * D + 32], acc[rg], o1, 0, 0, 0);
            }
        }
        if (t + 1 < NT) lstore(buf ^ 1);   // slot buf^1 was last read in iteration t-1: everybody has passed that iteration's barrier
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32);
    if (qrow < N) {   // lane owns query column r32; accumulator register rg is d = (rg&3) + 8(rg>>2) + 4h (+32 for o1)
        const float inv = 1.0f / l;
        const size_t e0 = ((size_t)b * N + qrow) * C + head * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 8 * g + 4 * h;
            if constexpr (OUT == 3) {
                x3_store4(out, e0 + d0, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                x3_store4(out, e0 + 32 + d0, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            } else {
                float* orow = static_cast<float*>(out) + e0;
                *reinterpret_cast<float4*>(orow + d0) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                *reinterpret_cast<float4*>(orow + 32 + d0) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            }
        }
    }
}

}  // namespace

int launch_vit_attention(const void* qkv, void* out, int prec, int B, int N, int heads, hipStream_t st, std::string& err, int out_x3) {
    if (B <= 0 || N <= 0 || heads <= 0) { err = "vit_attention: bad geometry"; return 1; }
    if (prec == SOCCDPT_PREC_F32 || prec == SOCCDPT_PREC_F16X3) {   // qkv plain f32 in both; the F16X3 mode wants its output in the x3 operand format
        const int NQB = (N + 127) / 128;
        const unsigned blocks = (unsigned)(B * heads * NQB);
        if (prec == SOCCDPT_PREC_F16X3) SOCCDPT_LAUNCH(vit_attention_f32_kernel<3>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(qkv), out, N, heads, NQB);
        else SOCCDPT_LAUNCH(vit_attention_f32_kernel<2>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(qkv), out, N, heads, NQB);
        return check_launch("vit_attention_f32", err);
    }
    if (out_x3 && prec != SOCCDPT_PREC_F16) { err = "vit_attention: the x3 output form belongs to the fp16 kernel"; return 1; }
    const int NPAD = (N + 31) / 32 * 32, NT = NPAD / 32;
    size_t lds = (size_t)NPAD * KROW + (size_t)D * (NPAD * 2 + 8);
    // the key halves meet in the same LDS (dead K / V^T region): up to four query blocks x (KH - 1) parked parts of [34][64] floats each.
    // Short sequences (N <= 128) hold less K / V^T than that exchange needs: size for the larger of the two.
    constexpr size_t XCH = (size_t)(KH - 1) * 4 * 34 * 64 * sizeof(float);
    if (lds < XCH) lds = XCH;
    if (lds > 160 * 1024) { err = "vit_attention: sequence too long for the LDS-resident K / V^T form (max 608 tokens)"; return 1; }
    // query split: enough workgroups to cover the 256 CUs, at most one 32-query block per wave and workgroup
    int QS = (256 + B * heads - 1) / (B * heads);
    if (QS < (NT + 3) / 4) QS = (NT + 3) / 4;   // ... and at most four per workgroup (8 waves = 4 query blocks x 2 key halves, one block per wave: the halves meet in the dead K region)
    QS = QS < 1 ? 1 : (QS > NT ? NT : QS);
    const int per = (NT + QS - 1) / QS;
    QS = (NT + per - 1) / per;    // drop empty parts
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_attention_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_attention_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { err = std::string("vit_attention: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    const unsigned blocks = (unsigned)(B * heads * QS);
    if (prec == SOCCDPT_PREC_F16)
        SOCCDPT_LAUNCH(vit_attention_kernel<true>, dim3(blocks), dim3(256 * KH), lds, st, static_cast<const uint16_t*>(qkv), static_cast<uint16_t*>(out), N, NPAD, heads, QS, out_x3);
    else
        SOCCDPT_LAUNCH(vit_attention_kernel<false>, dim3(blocks), dim3(256 * KH), lds, st, static_cast<const uint16_t*>(qkv), static_cast<uint16_t*>(out), N, NPAD, heads, QS, out_x3);
    return check_launch("vit_attention", err);
}

}  // namespace soccdpt
