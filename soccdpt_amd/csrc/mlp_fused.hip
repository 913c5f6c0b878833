// Fused Swin-V2 MLP half-block for the narrow stages:   x <- x + LayerNorm( fc2( GELU( fc1(x_op) ) ) )
// (timm Mlp + the res-post-norm of SwinTransformerV2Block; HF modeling_swinv2.py:589-706), one launch instead of
// fc1 GEMM (+GELU) -> fc2 GEMM -> LayerNorm+residual.  At C = 96 / 128 those three are pure traffic and latency: K = C gives 3-4
// k-tiles per workgroup and the 4C-wide hidden activation (25 MB at B = 8, 256 x 256) is written and read back.  Here a workgroup
// owns 64 tokens and walks the hidden dimension in chunks of 64 units; the hidden activation only ever exists as a 64 x 64 bf16
// tile in LDS:
//
//   per chunk c:  P1  H_c = GELU(X W1_c^T + b1_c)      64 tokens x 64 hidden,  K = C        (wave w: hidden 16w..16w+15, all 64 tokens)
//                 P2  Y  += H_c W2_c^T                  64 tokens x C,          K = 64       (wave w: tokens 16w..16w+15, all C channels)
//   epilogue:     x_f32 += LN(Y + b2) * g + b ; operand-typed copy (and the hooked stage's zero-halo image)
//
// Same MFMA conventions as igemm.hip: v_mfma_f32_16x16x32_{bf16,f16}, the weight tile is the A operand and the token tile the B
// operand, so a lane's 4 accumulators are 4 consecutive hidden units (P1: one 8-byte LDS store into H_c) or channels (P2: 16-byte
// f32 / 8-byte 16-bit global stores) of one token; and in P2 a wave owns whole token rows, so LayerNorm needs no cross-wave
// exchange.  X, W1_c and W2_c are staged with global_load_lds_dwordx4 (raw s_barrier + explicit waitcnt).  NBUF = 2 keeps
// W1_{c+1} / W2_{c+1} in flight while chunk c is computed; measured, the LDS it costs is worth more as a third resident workgroup
// (C = 96: 59 us per 2 launches either way; C = 128: 167 us single-buffered vs 217 us double-buffered), so NBUF = 1 ships.
// Measured in the network (B = 8): C = 96 29.5 us per half-block against 25 + 27 us for fc1 and fc2+LN; C = 128 (base_384) 84 vs 108 us;
// C = 192 / 256 lose (40 vs 35 us, 107 vs 95 us: the weight stream per 64-token workgroup grows with C^2), so the network fuses C <= 128.
// About 10 us of the 29.5 is the erf-GELU's VALU work itself (12.6 M activations).  16-byte chunks are XOR-swizzled by (row & 7) on the
// source side and on the ds_read_b128 side.  Rows of X / W1 are padded to a multiple of 128 bytes so the swizzle stays in the row.
#include <type_traits>

#include "gelu.h"
#include "half16.h"
#include "kernels.h"

namespace soccdpt {
namespace {

template <int C_, int NBUF_, int BM_ = 64>
struct MlpCfg {
    static constexpr int C = C_, NBUF = NBUF_;
    static constexpr int BM = BM_, HC = 64, THREADS = 256;
    static constexpr int TM1 = BM / 16;                  // token tiles of P1 (every wave covers all BM tokens)
    static constexpr int TM2 = BM / 64;                  // token tiles per wave in P2 (wave w owns tokens w*BM/4 .. +BM/4)
    static constexpr int XROW = ((C + 63) / 64) * 128;   // bytes per X / W1 row in LDS
    static constexpr int XPOS = XROW / 16;               // 16-byte positions per row
    static constexpr int CCH = C / 8;                    // real 16-byte chunks per row
    static constexpr int SWZ = (XPOS % 16 == 0) ? 15 : 7; // swizzle mask: 256-byte row groups sweep all 64 banks with 16 positions
    static constexpr int KS1 = C / 32;                   // MFMA k-steps of P1
    static constexpr int TN2 = C / 16;                   // channel tiles of P2
    static constexpr int X_BYTES = BM * XROW, W1_BYTES = HC * XROW, H_BYTES = BM * HC * 2, W2_BYTES = C * HC * 2;
    static constexpr int X_LOADS = BM * XPOS / THREADS, W1_LOADS = HC * XPOS / THREADS, W2_LOADS = C * 8 / THREADS;
    static constexpr int LDS = X_BYTES + H_BYTES + NBUF * (W1_BYTES + W2_BYTES);
    static_assert(C % 32 == 0 && C <= 256, "C must be a multiple of 32, at most 256");
    static_assert(X_LOADS * THREADS == BM * XPOS && W2_LOADS * THREADS == C * 8, "staging does not divide evenly");
};

template <class K, bool F16>
__global__ __launch_bounds__(K::THREADS) void mlp_ln_kernel(const uint16_t* __restrict__ xop, float* __restrict__ xf, const uint16_t* __restrict__ w1,
                                                            const float* __restrict__ b1, const uint16_t* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ g, const float* __restrict__ be, uint16_t* __restrict__ xop_out,
                                                            uint16_t* __restrict__ halo, int M, int H, int W, int merge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = K::C, HID = 4 * C, NCH = HID / K::HC;
    char* const Xs = smem;
    char* const Hs = Xs + K::X_BYTES;
    char* const W1s = Hs + K::H_BYTES;
    char* const W2s = W1s + K::NBUF * K::W1_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * K::BM;

    auto dma = [](const uint16_t* gsrc, char* ldst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
    };
    // rows x XPOS positions of a [rows][C] matrix (X tile, W1 chunk): position p of row r holds source chunk p ^ (r & SWZ)
    auto stage_rows = [&](const uint16_t* src, int row0, int row_max, char* dst, int loads) {
        for (int i = 0; i < loads; ++i) {
            const int pid = i * K::THREADS + tid;
            const int r = pid / K::XPOS, p = pid % K::XPOS;
            int c = p ^ (r & K::SWZ);
            c = c < K::CCH ? c : 0;   // padding positions of the last 128-byte group: any valid address
            int gr = row0 + r;
            gr = gr < row_max ? gr : row_max - 1;
            dma(src + (size_t)gr * C + c * 8, dst + (i * K::THREADS + wave * 64) * 16);
        }
    };
    auto stage_w2 = [&](int hc, char* dst) {
#pragma unroll
        for (int i = 0; i < K::W2_LOADS; ++i) {
            const int pid = i * K::THREADS + tid;
            const int n = pid >> 3, p = pid & 7;
            dma(w2 + (size_t)n * HID + hc * K::HC + ((p ^ (n & 7)) * 8), dst + (i * K::THREADS + wave * 64) * 16);
        }
    };

    stage_rows(xop, m0, M, Xs, K::X_LOADS);
    stage_rows(w1, 0, HID, W1s, K::W1_LOADS);
    stage_w2(0, W2s);

    f32x4_t acc2[K::TM2][K::TN2];
#pragma unroll
    for (int t = 0; t < K::TM2; ++t)
#pragma unroll
        for (int i = 0; i < K::TN2; ++i) acc2[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int hc = 0; hc < NCH; ++hc) {
        const int buf = K::NBUF == 2 ? (hc & 1) : 0;
        // chunk hc's weights (and, the first time, X) have landed; every wave is done with chunk hc-1 (its H tile and weight buffers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the bias load goes out BEFORE the prefetch: vmcnt retires in order, so a younger load would make its consumer wait for the DMA
        const float4 bb = *reinterpret_cast<const float4*>(b1 + hc * K::HC + wave * 16 + fq * 4);
        asm volatile("" ::: "memory");
        if (K::NBUF == 2 && hc + 1 < NCH) {
            stage_rows(w1, (hc + 1) * K::HC, HID, W1s + (buf ^ 1) * K::W1_BYTES, K::W1_LOADS);
            stage_w2(hc + 1, W2s + (buf ^ 1) * K::W2_BYTES);
        }
        // ---- P1: hidden units 16*wave .. +15 of this chunk for all 64 tokens ----
        const char* w1b = W1s + buf * K::W1_BYTES;
        f32x4_t acc1[K::TM1];
#pragma unroll
        for (int j = 0; j < K::TM1; ++j) acc1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < K::KS1; ++ks) {
            const int wr = wave * 16 + frow;
            const h16x8 wfrag = *reinterpret_cast<const h16x8*>(w1b + wr * K::XROW + (((ks * 4 + fq) ^ (wr & K::SWZ)) * 16));
#pragma unroll
            for (int j = 0; j < K::TM1; ++j) {
                const int xr = j * 16 + frow;
                const h16x8 xfrag = *reinterpret_cast<const h16x8*>(Xs + xr * K::XROW + (((ks * 4 + fq) ^ (xr & K::SWZ)) * 16));
                acc1[j] = mfma_16x16x32<F16>(wfrag, xfrag, acc1[j]);
            }
        }
        {
            const int ch = wave * 2 + (fq >> 1);   // 16-byte chunk of the token's H row that holds hidden 16*wave + 4*fq .. +3
#pragma unroll
            for (int j = 0; j < K::TM1; ++j) {
                const int tr = j * 16 + frow;
                uint2 p;
                p.x = pack_h2<F16>(gelu_fast(acc1[j][0] + bb.x), gelu_fast(acc1[j][1] + bb.y));
                p.y = pack_h2<F16>(gelu_fast(acc1[j][2] + bb.z), gelu_fast(acc1[j][3] + bb.w));
                *reinterpret_cast<uint2*>(Hs + tr * (K::HC * 2) + ((ch ^ (tr & 7)) * 16) + (fq & 1) * 8) = p;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's H stores are in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- P2: tokens 16*wave .. +15, all C channels, K = the 64 hidden units of this chunk ----
        const char* w2b = W2s + buf * K::W2_BYTES;
#pragma unroll
        for (int ks = 0; ks < K::HC / 32; ++ks) {
            h16x8 hfrag[K::TM2];
#pragma unroll
            for (int t = 0; t < K::TM2; ++t) {
                const int tr = (wave * K::TM2 + t) * 16 + frow;
                hfrag[t] = *reinterpret_cast<const h16x8*>(Hs + tr * (K::HC * 2) + (((ks * 4 + fq) ^ (tr & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < K::TN2; ++i) {
                const int n = i * 16 + frow;
                const h16x8 wfrag = *reinterpret_cast<const h16x8*>(w2b + n * (K::HC * 2) + (((ks * 4 + fq) ^ (n & 7)) * 16));
#pragma unroll
                for (int t = 0; t < K::TM2; ++t) acc2[t][i] = mfma_16x16x32<F16>(wfrag, hfrag[t], acc2[t][i]);
            }
        }
        if (K::NBUF == 1 && hc + 1 < NCH) {   // single-buffered weights: everybody is done reading them, then fetch the next chunk
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_rows(w1, (hc + 1) * K::HC, HID, W1s, K::W1_LOADS);
            stage_w2(hc + 1, W2s);
        }
    }

    // ---- epilogue: lane owns channels i*16 + 4*fq .. +3 of token m0 + (wave*TM2 + t)*16 + frow; LayerNorm over the C channels of the token ----
#pragma unroll
    for (int t = 0; t < K::TM2; ++t) {
        const int m = m0 + (wave * K::TM2 + t) * 16 + frow;
        const int mc = m < M ? m : M - 1;
        float4 xres[K::TN2];   // the residual row is requested before the LayerNorm reductions, not after them
#pragma unroll
        for (int i = 0; i < K::TN2; ++i) xres[i] = *reinterpret_cast<const float4*>(xf + (size_t)mc * C + i * 16 + fq * 4);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < K::TN2; ++i) {
            const float4 bb = *reinterpret_cast<const float4*>(b2 + i * 16 + fq * 4);
            acc2[t][i][0] += bb.x; acc2[t][i][1] += bb.y; acc2[t][i][2] += bb.z; acc2[t][i][3] += bb.w;
            sum += acc2[t][i][0] + acc2[t][i][1] + acc2[t][i][2] + acc2[t][i][3];
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < K::TN2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq += (acc2[t][i][r] - mean) * (acc2[t][i][r] - mean);
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)C + 1e-5f);
        if (m >= M) continue;
        const size_t orow = (size_t)m * C;
        size_t hrow = 0, brow = orow;
        if (halo || merge) {
            const int hw = H * W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / W, x = rem - y * W;
            hrow = ((size_t)(b * (H + 2) + y + 1) * (W + 2) + x + 1) * C;
            // merge: operand copy in the PatchMerging layout [B][H/2][W/2][4C], channel block (y&1) + 2*(x&1) (elementwise.hip, ln_residual)
            if (merge) brow = (((size_t)(b * (H / 2) + y / 2) * (W / 2) + x / 2) * 4 + (y & 1) + 2 * (x & 1)) * C;
        }
#pragma unroll
        for (int i = 0; i < K::TN2; ++i) {
            const int n = i * 16 + fq * 4;
            const float4 g4 = *reinterpret_cast<const float4*>(g + n), e4 = *reinterpret_cast<const float4*>(be + n);
            const float4 x4 = xres[i];
            float o[4];
            o[0] = x4.x + ((acc2[t][i][0] - mean) * rstd * g4.x + e4.x);
            o[1] = x4.y + ((acc2[t][i][1] - mean) * rstd * g4.y + e4.y);
            o[2] = x4.z + ((acc2[t][i][2] - mean) * rstd * g4.z + e4.z);
            o[3] = x4.w + ((acc2[t][i][3] - mean) * rstd * g4.w + e4.w);
            *reinterpret_cast<float4*>(xf + orow + n) = make_float4(o[0], o[1], o[2], o[3]);
            uint2 p;
            p.x = pack_h2<F16>(o[0], o[1]);
            p.y = pack_h2<F16>(o[2], o[3]);
            if (xop_out) *reinterpret_cast<uint2*>(xop_out + brow + n) = p;
            if (halo) *reinterpret_cast<uint2*>(halo + hrow + n) = p;
        }
    }
}

template <class K, bool F16>
int launch_one(const uint16_t* xop, float* xf, const uint16_t* w1, const float* b1, const uint16_t* w2, const float* b2, const float* g, const float* be,
               uint16_t* xop_out, uint16_t* halo, int M, int H, int W, int merge, hipStream_t st, std::string& err) {
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_ln_kernel<K, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS);
        if (e != hipSuccess) { err = std::string("mlp_ln: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    SOCCDPT_LAUNCH((mlp_ln_kernel<K, F16>), dim3((unsigned)((M + K::BM - 1) / K::BM)), dim3(K::THREADS), K::LDS, st, xop, xf, w1, b1, w2, b2, g, be,
                       xop_out, halo, M, H, W, merge);
    return check_launch("mlp_ln", err);
}

}  // namespace

bool mlp_ln_supported(int C) { return C == 96 || C == 128 || C == 192 || C == 256; }

int launch_mlp_ln(const bf16_t* xop, float* xf, const bf16_t* w1, const float* b1, const bf16_t* w2, const float* b2, const float* g, const float* be,
                  bf16_t* xop_out, bf16_t* halo, int hf, int M, int C, int H, int W, int merge, hipStream_t st, std::string& err) {
    if (!xop || !xf || !w1 || !b1 || !w2 || !b2 || !g || !be || M < 1) { err = "mlp_ln: bad arguments"; return 1; }
    if ((halo || merge) && (H <= 0 || W <= 0 || M % (H * W) != 0)) { err = "mlp_ln: halo / merged output needs the token grid"; return 1; }
    if (merge && ((H & 1) || (W & 1) || !xop_out || xop_out == xop)) { err = "mlp_ln: merged operand layout needs an even token grid and a separate output"; return 1; }
#define MLP_CASE(CC, NB)                                                                                                              \
    case CC:                                                                                                                          \
        return hf ? launch_one<MlpCfg<CC, NB>, true>(xop, xf, w1, b1, w2, b2, g, be, xop_out, halo, M, H, W, merge, st, err)                 \
                  : launch_one<MlpCfg<CC, NB>, false>(xop, xf, w1, b1, w2, b2, g, be, xop_out, halo, M, H, W, merge, st, err)
    switch (C) {
        MLP_CASE(96, 1);
        MLP_CASE(128, 1);
        MLP_CASE(192, 1);
        MLP_CASE(256, 1);
    }
#undef MLP_CASE
    err = "mlp_ln: C must be 96, 128, 192 or 256";
    return 1;
}

}  // namespace soccdpt
