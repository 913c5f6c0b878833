// bf16 MFMA implicit GEMM for gfx950: Linear layers, 1x1 and 3x3 convolutions of the
// SOccDPT_V3 encoder/decoder with fused epilogues.
//
//   out[m][n] = epilogue( sum_k X[m][k] * Wt[n][k] )            fp32 accumulation
//
// Replaces the ATen conv2d / linear calls of
//   /root/reference/SOccDPT/model/blocks.py:155-191,391-414,488-495 (scratch convs, RCUs, out_conv)
//   /root/reference/SOccDPT/model/dpt.py:199-219 (depth head), model/SOccDPT.py:660-674 (seg head)
//   and timm's qkv / proj / fc1 / fc2 / reduction Linear layers.
//
// Design (CDNA4):
//  * v_mfma_f32_16x16x32_bf16, 64-lane waves; the WEIGHT tile is the MFMA A operand and the
//    activation tile the B operand, so a lane's 4 accumulator registers are 4 consecutive output
//    channels of one pixel -> 8-byte (bf16) / 16-byte (f32) channel-contiguous NHWC stores.
//  * both operand tiles are staged HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round
//    trip), double-buffered; the 3x3 taps are gathered by the per-lane SOURCE address from a
//    zero-haloed NHWC image, so the main loop has no bounds checks.
//  * LDS rows are BK*2 bytes; the 16-byte chunks of a row are XOR-swizzled on the source side and
//    on the ds_read_b128 side (same involution) -> conflict-free fragment reads.
//  * blockIdx is remapped so the N-tiles that share an activation tile run on one XCD (L2 reuse).
#include <type_traits>

#include "gelu.h"
#include "half16.h"
#include "igemm.h"

namespace soccdpt {

typedef __attribute__((ext_vector_type(4))) float f32x4;
struct f16_t { uint16_t v; };  // element tag of the fp16 instantiations (SOCCDPT_PREC_F16); bf16_t tags bf16, float exact f32

// BK_ is the k-tile depth in bf16 elements; a tile row is ROWB = 2*BK_ bytes (128 or 64).  With f32 operands
// (SOCCDPT_PREC_F32) the same byte geometry holds BK_/2 elements per row.
// MF_: MFMA shape of the 16-bit instantiations: 16 = v_mfma_f32_16x16x32 (a wave tile is TM x TN tiles of 16 x 16), 32 = v_mfma_f32_32x32x16 (a wave tile
// is (BM/WM/32) x (BN/WN/32) tiles of 32 x 32; half the MFMA issue slots per FLOP).  Same LDS bytes per FLOP for the same wave tile.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int NS_, int MF_ = 16>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, NS = NS_, MF = MF_;  // NS: LDS stages (tiles in flight + 1)
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int ROWB = BK * 2;    // bytes per LDS tile row
    static constexpr int CPR = ROWB / 16;  // 16-byte chunks per row
    static constexpr int X_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
    static constexpr int STAGE = X_BYTES + W_BYTES;
    static constexpr int X_LOADS = X_BYTES / 16 / THREADS;
    static constexpr int W_LOADS = W_BYTES / 16 / THREADS;
    static constexpr int TM = BM / WM / 16, TN = BN / WN / 16;  // 16x16 tiles per wave
    static constexpr int KS = BK / 32;
    static constexpr int LOADS = X_LOADS + W_LOADS;  // LDS-DMA instructions per thread per k-tile
    static_assert(X_LOADS * THREADS * 16 == X_BYTES && W_LOADS * THREADS * 16 == W_BYTES, "tile/threads mismatch");
    static_assert(NS >= 2 && (NS - 2) * LOADS <= 63, "vmcnt immediate is 6 bits");
};

template <int BK, int MF = 16>
__device__ __forceinline__ int swz_of_row(int row) {
    // 32 x 32 MFMA fragments: the 32 lanes of a half-wave read 32 ROWS at ONE chunk; ds_read_b128 serves lane groups {0-3, 12-15, 20-27} /
    // {4-11, 16-19, 28-31}: rows of equal parity share a 128-byte bank half, (row >> 1) & 7 gives each of a group's 8 such rows its own slot
    if constexpr (MF == 32 && BK == 64) return (row >> 1) & 7;
    if constexpr (BK == 128) return row & 15;        // 256-byte rows (one full bank sweep each): 16 chunks
    else if constexpr (BK == 64) return row & 7;     // 128-byte rows: 8 chunks
    else return (-(row >> 2)) & 3;                   // 64-byte rows: 4 chunks, rows r and r+4 share banks
}

// T = bf16_t (v_mfma_f32_16x16x32_bf16), f16_t (v_mfma_f32_16x16x32_f16), float (v_mfma_f32_16x16x4_f32, exact f32) or x3_t: split-fp16 operands
// (half16.h: every element an fp16 pair hi, lo * 2^11 in the 4-byte-per-element x3 layout; three v_mfma_f32_16x16x32_f16 per product --
// hi hi into `acc`, hi lo + lo hi into a second accumulator set folded in with 2^-11 after the k-loop: SOCCDPT_PREC_F16X3).  The x3 tiles share
// the f32 tiles' byte geometry (128-byte LDS rows = 32 elements = one MFMA k-step), staging code and epilogues.
// LN: the fused post-norm LayerNorm + residual epilogue (d.ln_g) instead of the generic one; a separate instantiation so that
// its registers (row statistics) do not inflate the generic kernels (measured: 110 -> 158 VGPRs, one block per CU less).
// SK: split-K.  gridDim.x = tiles x d.splitk; every workgroup accumulates its slice of the k-tiles, stores the f32 partial
// tile to d.sk_part[split][M][N], and the LAST workgroup to arrive at the tile (a counter in d.sk_count, left at 0 again) sums
// the splitk partials in split order -- deterministic, no float atomics -- and runs the epilogue.  No workgroup ever waits
// for another one.  For long-K problems whose output grid cannot fill the 256 CUs (coarse decoder levels, stage-3 fc2).
// ST: GroupNorm statistics of the raw output (d.gn_stats): per-tile per-group partial sums, finished by the last workgroup of each
// sample (same fence-free sc1 exchange as SK).  ResNetV2 stages of the ViT-hybrid encoder (csrc/hybrid.hip applies the normalisation).
// GEN: the generalised addressing (strided / un-haloed / gathered convolution, row groups, second A segment) and the diagnostics stamps.
// A separate instantiation: carried by every launch they cost the Swin models 1.5 % of the forward (A/B in one GPU call, tools/ab_bench.sh).
template <class C, typename T, bool LN, bool SK = false, bool ST = false, bool GEN = false>
__global__ __launch_bounds__(C::THREADS) void igemm_kernel(IgemmDesc d, int nk, int kpt, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = C::BM, BN = C::BN;
    constexpr int BK = C::ROWB / (int)sizeof(T);   // k-tile depth in elements of T
    constexpr int EPC = 16 / (int)sizeof(T);       // elements per 16-byte chunk
    constexpr bool F16 = std::is_same<T, f16_t>::value;
    constexpr bool X3 = std::is_same<T, x3_t>::value;
    static_assert(!X3 || C::ROWB >= 128, "an x3 k-step (32 elements) is 128 bytes of a tile row");
    constexpr int MF = C::MF;
    static_assert(MF == 16 || (MF == 32 && sizeof(T) == 2 && C::BK == 64 && !LN && !SK && !ST), "32x32x16 tiles: 16-bit operands, 64-deep k-tiles, plain epilogue");
    // epilogue view of a wave's accumulators, common to both MFMA shapes: TME m-tiles x TNE groups of 4 consecutive channels per lane;
    // m_of(j) / n_of(i) = the pixel / first channel a lane owns in group (i, j)
    constexpr int TME = MF == 32 ? BM / C::WM / 32 : C::TM;
    constexpr int TNE = MF == 32 ? (BN / C::WN / 32) * 4 : C::TN;
    const T* const Xp = static_cast<const T*>(d.X);
    const T* const Wtp = static_cast<const T*>(d.Wt);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    if (GEN && d.stamps && tid == 0) d.stamps[4 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    // XCD-aware bijective remap: consecutive logical tiles -> same XCD (blocks b, b+8 share one)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int split = 0, kbase = 0;
    if constexpr (SK) {  // splits of one tile are consecutive logical ids: same XCD, their partials meet in one L2
        split = bid % d.splitk;
        bid /= d.splitk;
        kbase = (int)((long)split * nk / d.splitk);
        nk = (int)((long)(split + 1) * nk / d.splitk) - kbase;
    }
    const int tile_id = bid;
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    auto m_of = [&](int j) { return m0 + wm * (BM / C::WM) + (MF == 32 ? j * 32 + (lane & 31) : j * 16 + (lane & 15)); };
    auto n_of = [&](int i) { return n0 + wn * (BN / C::WN) + (MF == 32 ? (i >> 2) * 32 + (i & 3) * 8 + (lane >> 5) * 4 : i * 16 + (lane >> 4) * 4); };
    const int Ktot = d.taps * d.Cin;
    const int Wp = d.W + 2;                 // OUTPUT halo geometry (out_halo / ln_halo stores)
    const bool conv_addr = d.taps == 9 || (GEN && d.gather1);
    const int Wpi = GEN ? (d.Wi ? d.Wi : d.W) + 2 * d.in_halo : d.W + 2;   // INPUT image geometry
    const int Hpi = GEN ? (d.Hi ? d.Hi : d.H) + 2 * d.in_halo : d.H + 2;

    // ---- per-thread staging sources (element offsets) ----
    uint32_t x_off[C::X_LOADS], x_off2[C::X_LOADS], w_off[C::W_LOADS];
#pragma unroll
    for (int i = 0; i < C::X_LOADS; ++i) {
        const int cid = i * C::THREADS + tid;
        const int row = cid / C::CPR, c = cid % C::CPR;
        int m = m0 + row;
        m = m < d.M ? m : d.M - 1;
        uint32_t base, base2 = 0;
        if (conv_addr) {
            const int hw = d.H * d.W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / d.W, x = rem - y * d.W;
            if constexpr (GEN) base = (uint32_t)(((b * Hpi + y * d.stride + d.in_halo - d.pad) * Wpi + x * d.stride + d.in_halo - d.pad) * d.Cin);
            else base = (uint32_t)(((b * Hpi + y) * Wpi + x) * d.Cin);
        } else if (GEN && d.grp_rows) {
            const int g = m / d.grp_rows, r = m - g * d.grp_rows;
            const uint32_t gb = (uint32_t)((long long)g * d.grp_stride);
            base = gb + (uint32_t)d.grp_off + (uint32_t)r * (uint32_t)d.ldx;
            base2 = gb + (uint32_t)d.seg2_off;
        } else {
            base = (uint32_t)m * (uint32_t)d.ldx;
        }
        const uint32_t sw = (uint32_t)((c ^ swz_of_row<C::BK, C::MF>(row)) * EPC);
        x_off[i] = base + sw;
        x_off2[i] = base2 - base;   // delta to the second-segment row (mod 2^32), added when the k-tile lies in the second segment
    }
#pragma unroll
    for (int i = 0; i < C::W_LOADS; ++i) {
        const int cid = i * C::THREADS + tid;
        const int row = cid / C::CPR, c = cid % C::CPR;
        int n = n0 + row;
        n = n < d.N ? n : d.N - 1;
        if (GEN && d.wt_grp_rows) {   // weight row groups: the tile's rows belong to ONE group (wt_grp_rows % BN == 0): a shifted view of the same matrix
            const int g = n0 / d.wt_grp_rows, ky = g / 3, kx = g - ky * 3;
            const uint32_t shift = d.wt_kx ? (uint32_t)(d.wt_base + (ky - 1) * d.wt_rp + kx * d.wt_kx)
                                           : (uint32_t)(d.wt_base + (ky - 1) * d.wt_rp + (kx - 1) + (kx != 1 ? d.wt_odd : 0));
            w_off[i] = (uint32_t)(n - g * d.wt_grp_rows) * (uint32_t)Ktot + shift + (uint32_t)((c ^ swz_of_row<C::BK, C::MF>(row)) * EPC);
        } else
        w_off[i] = (uint32_t)n * (uint32_t)Ktot + (uint32_t)((c ^ swz_of_row<C::BK, C::MF>(row)) * EPC);
    }

    auto stage = [&](int kt, int buf) {
        kt += kbase;
        uint32_t xk, wk = (uint32_t)kt * BK;  // elements of T
        bool seg2 = false;
        if (d.taps == 9) {
            const int tap = kt / kpt, kc = kt - tap * kpt;
            const int ky = tap / 3, kx = tap - ky * 3;
            xk = (uint32_t)((ky * Wpi + kx) * d.Cin + kc * BK);
        } else if (GEN && d.seg2_k && (int)wk >= d.seg2_k) {   // wave-uniform: the per-group row (ViT readout token)
            seg2 = true;
            xk = wk - (uint32_t)d.seg2_k;
        } else {
            xk = wk;
        }
        char* sb = smem + buf * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::X_LOADS; ++i) {
            const T* g = Xp + (x_off[i] + ((GEN && seg2) ? x_off2[i] : 0u)) + xk;   // a VALUE select: selecting between the two arrays demotes them (and d) to scratch
            char* l = sb + (i * C::THREADS + wave * 64) * 16;  // wave-uniform base; HW adds lane*16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::W_LOADS; ++i) {
            const T* g = Wtp + w_off[i] + wk;
            char* l = sb + C::X_BYTES + (i * C::THREADS + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };

    f32x4 acc[TNE][TME];
#pragma unroll
    for (int i = 0; i < TNE; ++i)
#pragma unroll
        for (int j = 0; j < TME; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accx[X3 ? TNE : 1][X3 ? TME : 1];   // x3: the cross terms hi*lo + lo*hi (scaled by 2^11)
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < TNE; ++i)
#pragma unroll
            for (int j = 0; j < TME; ++j) accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    constexpr int TN32 = MF == 32 ? BN / C::WN / 32 : 1, TM32 = MF == 32 ? BM / C::WM / 32 : 1;
    f32x16_t acc32[TN32][TM32];   // MF == 32: the 32 x 32 accumulators of the main loop (re-viewed as `acc` groups for the epilogue)
    if constexpr (MF == 32) {
#pragma unroll
        for (int i = 0; i < TN32; ++i)
#pragma unroll
            for (int j = 0; j < TM32; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    }

    // Small tiles are latency chains (a handful of k-tiles, then the epilogue): what the epilogue reads -- bias and the f32 residual
    // rows -- is requested here, BEFORE the first LDS-DMA group, so it is older than every counted vmcnt wait and costs no wait of
    // its own.  Big tiles prefetch only the bias (their residual rows would cost 64 registers).
    constexpr bool PRE = !SK && (TNE * TME <= 8);
    constexpr bool PREB = !SK;   // the bias alone is cheap enough (TN x 4 registers) for every tile size: 128x128 convs 338 -> 323 us
    float4 bias_pre[PREB ? TNE : 1];
    float4 res1_pre[PRE ? TNE : 1][PRE ? TME : 1];
    if constexpr (PREB) {
#pragma unroll
        for (int i = 0; i < TNE; ++i) {
            int n = n_of(i);
            n = n < d.N ? n : 0;
            bias_pre[i] = d.bias ? *reinterpret_cast<const float4*>(d.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PRE)
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                int m = m_of(j);
                m = m < d.M ? m : d.M - 1;
                res1_pre[i][j] = d.res1 ? *reinterpret_cast<const float4*>(d.res1 + (size_t)m * d.N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        asm volatile("" ::: "memory");
    }

    // fragment read offsets (bytes within a stage), constant per lane
    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = swz_of_row<C::BK, C::MF>(frow);
    int xr_off[C::KS], wr_off[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        const int q = ((ks * 4 + fq) ^ fswz) * 16;
        xr_off[ks] = (wm * C::TM * 16 + frow) * C::ROWB + q;
        wr_off[ks] = C::X_BYTES + (wn * C::TN * 16 + frow) * C::ROWB + q;
    }
    // x3: k-step s of a row covers units 4s .. 4s+3 (8 elements each); lane quarter fq takes unit u = 4s + fq, whose hi chunk is 2u + (u & 1)
    // and lo chunk 2u + 1 - (u & 1) (half16.h) -- with the row's XOR swizzle the 16-lane groups of ds_read_b128 hit 16 distinct bank slots
    constexpr int KSX = X3 ? C::ROWB / 128 : 1;
    int x3h_off[KSX], x3l_off[KSX];   // byte offsets of the hi / lo chunk inside a tile row (before the row base)
    if constexpr (X3) {
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
            const int u = ks * 4 + fq;
            x3h_off[ks] = ((2 * u + (u & 1)) ^ fswz) * 16;
            x3l_off[ks] = ((2 * u + 1 - (u & 1)) ^ fswz) * 16;
        }
    }
    const int x_row0 = (wm * C::TM * 16 + frow) * C::ROWB, w_row0 = C::X_BYTES + (wn * C::TN * 16 + frow) * C::ROWB;
    // 32 x 32 x 16 fragments: lane (row r32, half h2) reads chunk 2 ks + h2 of k-step ks (16 elements per step)
    constexpr int KS32 = MF == 32 ? C::BK / 16 : 1;
    int x32_off[KS32], w32_off[KS32];
    if constexpr (MF == 32) {
        const int r32 = lane & 31, h2 = lane >> 5, sw32 = swz_of_row<C::BK, C::MF>(r32);
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            const int q = ((ks * 2 + h2) ^ sw32) * 16;
            x32_off[ks] = (wm * (BM / C::WM) + r32) * C::ROWB + q;
            w32_off[ks] = C::X_BYTES + (wn * (BN / C::WN) + r32) * C::ROWB + q;
        }
    }

    // ---- NS-stage LDS ring: up to NS-1 k-tiles of LDS-DMA in flight, ONE raw barrier per k-tile.
    // Tile kt is waited for with a COUNTED vmcnt (the NS-2 younger tiles stay in flight), then the barrier
    // both publishes it to the other waves and retires everybody's reads of ring slot (kt-1)%NS, which the
    // next LDS-DMA group overwrites.  (__syncthreads() would drain vmcnt(0): cdna_hip_programming.md §5.)
    auto wait_tile = [&](int kt) {
        if (kt + C::NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C::NS - 2) * C::LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#pragma unroll
    for (int s0 = 0; s0 < C::NS - 1; ++s0)
        if (s0 < nk) stage(s0, s0);
    // (Tried and removed: running the wm == 1 wave row of the 8-wave tiles half a k-step out of phase, | M1' R0 M0 R1 | against
    //  | R0 M0 R1 M1 |, so that the two waves of a SIMD alternate load and MFMA halves: 5-8 % on the 256x256 / 128x256 convs.
    //  It was dropped when wrong seg pixels showed up in the eager two-stream mode; the same signature was later traced to
    //  packed-f32 math in the seg head's 1x1 kernel under co-residency, so the schedule was probably
    //  innocent -- but the retuned heuristics no longer pick the tiles it applied to.)
    {
    for (int kt = 0; kt < nk; ++kt) {
        wait_tile(kt);
        __builtin_amdgcn_s_barrier();
        if (GEN && kt == 0 && d.stamps && tid == 0) d.stamps[4 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        if (kt + C::NS - 1 < nk) stage(kt + C::NS - 1, (kt + C::NS - 1) % C::NS);
        const char* sb = smem + (kt % C::NS) * C::STAGE;
        if constexpr (X3) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                h16x8 wh[C::TN], wl[C::TN], xh[C::TM], xl[C::TM];
#pragma unroll
                for (int i = 0; i < C::TN; ++i) {
                    wh[i] = *reinterpret_cast<const h16x8*>(sb + w_row0 + x3h_off[ks] + i * 16 * C::ROWB);
                    wl[i] = *reinterpret_cast<const h16x8*>(sb + w_row0 + x3l_off[ks] + i * 16 * C::ROWB);
                }
#pragma unroll
                for (int j = 0; j < C::TM; ++j) {
                    xh[j] = *reinterpret_cast<const h16x8*>(sb + x_row0 + x3h_off[ks] + j * 16 * C::ROWB);
                    xl[j] = *reinterpret_cast<const h16x8*>(sb + x_row0 + x3l_off[ks] + j * 16 * C::ROWB);
                }
#pragma unroll
                for (int i = 0; i < C::TN; ++i)
#pragma unroll
                    for (int j = 0; j < C::TM; ++j) {
                        acc[i][j] = mfma_16x16x32<true>(wh[i], xh[j], acc[i][j]);
                        accx[i][j] = mfma_16x16x32<true>(wh[i], xl[j], accx[i][j]);
                        accx[i][j] = mfma_16x16x32<true>(wl[i], xh[j], accx[i][j]);
                    }
            }
        } else if constexpr (MF == 32) {
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                h16x8 wf[TN32], xf[TM32];
#pragma unroll
                for (int i = 0; i < TN32; ++i) wf[i] = *reinterpret_cast<const h16x8*>(sb + w32_off[ks] + i * 32 * C::ROWB);
#pragma unroll
                for (int j = 0; j < TM32; ++j) xf[j] = *reinterpret_cast<const h16x8*>(sb + x32_off[ks] + j * 32 * C::ROWB);
#pragma unroll
                for (int i = 0; i < TN32; ++i)
#pragma unroll
                    for (int j = 0; j < TM32; ++j) acc32[i][j] = mfma_32x32x16<F16>(wf[i], xf[j], acc32[i][j]);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            if constexpr (sizeof(T) == 2) {
                h16x8 wf[C::TN], xf[C::TM];
#pragma unroll
                for (int i = 0; i < C::TN; ++i) wf[i] = *reinterpret_cast<const h16x8*>(sb + wr_off[ks] + i * 16 * C::ROWB);
#pragma unroll
                for (int j = 0; j < C::TM; ++j) xf[j] = *reinterpret_cast<const h16x8*>(sb + xr_off[ks] + j * 16 * C::ROWB);
#pragma unroll
                for (int i = 0; i < C::TN; ++i)
#pragma unroll
                    for (int j = 0; j < C::TM; ++j)
                        acc[i][j] = mfma_16x16x32<F16>(wf[i], xf[j], acc[i][j]);
            } else {
                // f32: the lane's 16-byte chunk holds 4 consecutive k; element e of every lane forms MFMA k-step e
                // (A and B use the same lane->k map, so any k permutation is a valid dot product order)
                f32x4 wf[C::TN], xf[C::TM];
#pragma unroll
                for (int i = 0; i < C::TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(sb + wr_off[ks] + i * 16 * C::ROWB);
#pragma unroll
                for (int j = 0; j < C::TM; ++j) xf[j] = *reinterpret_cast<const f32x4*>(sb + xr_off[ks] + j * 16 * C::ROWB);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < C::TN; ++i)
#pragma unroll
                        for (int j = 0; j < C::TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], xf[j][e], acc[i][j], 0, 0, 0);
            }
        }
        }
    }
    }
    if constexpr (MF == 32) {   // accumulator register 4 g + r of tile (i, j) = channel 8 g + 4 h2 + r of pixel (lane & 31): epilogue group (4 i + g, j)
#pragma unroll
        for (int i = 0; i < TN32; ++i)
#pragma unroll
            for (int j = 0; j < TM32; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i * 4 + g][j][r] = acc32[i][j][4 * g + r];
    }
    if constexpr (X3) {   // fold the cross terms in: a b = hi hi + 2^-11 (hi lo + lo hi)
#pragma unroll
        for (int i = 0; i < TNE; ++i)
#pragma unroll
            for (int j = 0; j < TME; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(accx[i][j][r], 1.0f / 2048.f, acc[i][j][r]);
    }

    const int N = d.N;
    if (GEN && d.stamps && tid == 0) d.stamps[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    if constexpr (SK) {
        // Cross-workgroup exchange WITHOUT fences: a release/acquire fence at agent scope writes back / invalidates the whole
        // per-XCD L2 on gfx950 (measured: ~30 us per split).  Instead every partial is stored and loaded with agent-scope
        // (sc1, L2-bypassing) relaxed atomics, and "stored before counted" is enforced by s_waitcnt vmcnt(0) + the barrier.
        const size_t MN = (size_t)d.M * N;
        float* mine = d.sk_part + (size_t)split * MN;
        if (d.sk_defer) {   // partial tile out with plain stores; launch_igemm's second launch (sk_reduce_kernel) sums the splits
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                const int m = m_of(j);
#pragma unroll
                for (int i = 0; i < TNE; ++i) {
                    const int n = n_of(i);
                    if (m < d.M && n < N) *reinterpret_cast<float4*>(mine + (size_t)m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TME; ++j) {
            const int m = m_of(j);
#pragma unroll
            for (int i = 0; i < TNE; ++i) {
                const int n = n_of(i);
                if (m < d.M && n < N) {
                    float* q = mine + (size_t)m * N + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) __hip_atomic_store(q + r, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's partial stores are acknowledged at device scope
        __syncthreads();                                   // ... and so are everybody else's in this workgroup
        unsigned* arrival = reinterpret_cast<unsigned*>(smem);  // the staging ring is free after the barrier above
        if (tid == 0) *arrival = __hip_atomic_fetch_add(d.sk_count + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*arrival != (unsigned)d.splitk - 1) return;   // not the last split of this tile: done (nobody waits)
        if (tid == 0) __hip_atomic_store(d.sk_count + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
#pragma unroll
        for (int j = 0; j < TME; ++j) {
            const int m = m_of(j);
#pragma unroll
            for (int i = 0; i < TNE; ++i) {
                const int n = n_of(i);
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                if (m < d.M && n < N) {
                    for (int sp = 0; sp < d.splitk; ++sp) {   // fixed order: bitwise reproducible
                        const float* q = d.sk_part + (size_t)sp * MN + (size_t)m * N + n;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sum[r] += __hip_atomic_load(q + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                acc[i][j] = sum;
            }
        }
    }
    if constexpr (LN) {
        // ---- fused post-norm epilogue: out = (x +) LayerNorm(acc + bias) over the N (<= BN) channels of each row.
        // A row's channels are spread over the TN tiles x 4 lane groups of a wave and over the WN waves: two-pass mean /
        // variance with an in-wave shuffle reduction and a cross-wave exchange through LDS (the staging ring is free now).
        float* red = reinterpret_cast<float*>(smem);  // [BM][WN]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TNE; ++i) {
            const int n = n_of(i);
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (PREB) { if (n < N) b4 = bias_pre[i]; }
            else if (d.bias && n < N) b4 = *reinterpret_cast<const float4*>(d.bias + n);
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w;
            }
        }
        // the residual rows are requested before the two reduction passes (their latency hides behind the barriers)
        float4 xres[TNE][TME];
        if (d.ln_residual) {
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                int m = m_of(j);
                m = m < d.M ? m : d.M - 1;
#pragma unroll
                for (int i = 0; i < TNE; ++i) {
                    int n = n_of(i);
                    n = n < N ? n : 0;
                    xres[i][j] = *reinterpret_cast<const float4*>(d.ln_xf + (size_t)m * N + n);
                }
            }
        }
        float mean[TME], rstd[TME];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < TNE; ++i) {
                    const int n = n_of(i);
                    if (n < N) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = pass == 0 ? acc[i][j][r] : (acc[i][j][r] - mean[j]) * (acc[i][j][r] - mean[j]);
                            sum += t;
                        }
                    }
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                if ((lane >> 4) == 0) red[(wm * TME * 16 + j * 16 + (lane & 15)) * C::WN + wn] = sum;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TME; ++j) {
                float tot = 0.f;
#pragma unroll
                for (int wv = 0; wv < C::WN; ++wv) tot += red[(wm * TME * 16 + j * 16 + (lane & 15)) * C::WN + wv];
                if (pass == 0) mean[j] = tot / (float)N;
                else rstd[j] = rsqrtf(tot / (float)N + 1e-5f);
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < TME; ++j) {
            const int m = m_of(j);
            if (m >= d.M) continue;
            const size_t orow = (size_t)m * N;
            size_t hrow = 0;
            if (d.ln_halo) {
                const int hw = d.H * d.W;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / d.W, x = rem - y * d.W;
                hrow = ((size_t)(b * (d.H + 2) + y + 1) * Wp + x + 1) * N;
            }
#pragma unroll
            for (int i = 0; i < TNE; ++i) {
                const int n = n_of(i);
                if (n >= N) continue;
                const float4 g4 = *reinterpret_cast<const float4*>(d.ln_g + n), e4 = *reinterpret_cast<const float4*>(d.ln_b + n);
                float o[4];
                o[0] = (acc[i][j][0] - mean[j]) * rstd[j] * g4.x + e4.x;
                o[1] = (acc[i][j][1] - mean[j]) * rstd[j] * g4.y + e4.y;
                o[2] = (acc[i][j][2] - mean[j]) * rstd[j] * g4.z + e4.z;
                o[3] = (acc[i][j][3] - mean[j]) * rstd[j] * g4.w + e4.w;
                if (d.ln_residual) {
                    const float4 x4 = xres[i][j];
                    o[0] += x4.x; o[1] += x4.y; o[2] += x4.z; o[3] += x4.w;
                }
                *reinterpret_cast<float4*>(d.ln_xf + orow + n) = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (sizeof(T) == 2) {
                    uint2 p;
                    p.x = pack_h2<F16>(o[0], o[1]);
                    p.y = pack_h2<F16>(o[2], o[3]);
                    if (d.out_op) {
                        if (F16 && d.out_fmt == 3) x3_store4(d.out_op, orow + n, o[0], o[1], o[2], o[3]);
                        else *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + orow + n) = p;
                    }
                    if (d.ln_halo) {
                        if (F16 && d.halo_fmt == 3) x3_store4(d.ln_halo, hrow + n, o[0], o[1], o[2], o[3]);
                        else *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.ln_halo) + hrow + n) = p;
                    }
                } else if constexpr (X3) {
                    uint2 p;
                    p.x = pack_h2<true>(o[0], o[1]);
                    p.y = pack_h2<true>(o[2], o[3]);
                    if (d.out_op) {
                        if (d.out_fmt == 1) *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + orow + n) = p;
                        else x3_store4(d.out_op, orow + n, o[0], o[1], o[2], o[3]);
                    }
                    if (d.ln_halo) {
                        if (d.halo_fmt == 1) *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.ln_halo) + hrow + n) = p;
                        else x3_store4(d.ln_halo, hrow + n, o[0], o[1], o[2], o[3]);
                    }
                } else {
                    if (d.ln_halo) *reinterpret_cast<float4*>(static_cast<float*>(d.ln_halo) + hrow + n) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        return;
    } else {
    // ---- epilogue: lane owns channels n..n+3 of pixel m for each (i, j) ----
    float dot_part[TME];
#pragma unroll
    for (int j = 0; j < TME; ++j) dot_part[j] = 0.f;
    float gsum[ST ? TNE : 1][4], gsq[ST ? TNE : 1][4];   // ST: per-lane sums over this wave's pixel rows of its 4 channels per n-tile
    if constexpr (ST) {
#pragma unroll
        for (int i = 0; i < TNE; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { gsum[i][r] = 0.f; gsq[i][r] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < TME; ++j) {
        const int m = m_of(j);
        const bool mv = m < d.M;
        size_t orow = (size_t)m * N;
        size_t hrow = 0;
        // bilinear source of res2 (4 low-res pixels + weights), computed once per pixel
        size_t up00 = 0, up01 = 0, up10 = 0, up11 = 0;
        float uly = 0.f, ulx = 0.f;
        if ((d.out_halo || d.res2_h) && mv) {
            const int hw = d.H * d.W;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / d.W, x = rem - y * d.W;
            hrow = ((size_t)(b * (d.H + 2) + y + 1) * Wp + x + 1) * N;
            if (d.res2_h) {
                const float sy = d.H > 1 ? (float)(d.res2_h - 1) / (float)(d.H - 1) : 0.f;
                const float sx = d.W > 1 ? (float)(d.res2_w - 1) / (float)(d.W - 1) : 0.f;
                const float fy = sy * (float)y, fx = sx * (float)x;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < d.res2_h - 1), x1 = x0 + (x0 < d.res2_w - 1);
                uly = fy - (float)y0;
                ulx = fx - (float)x0;
                const size_t pb = (size_t)b * d.res2_h * d.res2_w;
                up00 = (pb + (size_t)y0 * d.res2_w + x0) * N;
                up01 = (pb + (size_t)y0 * d.res2_w + x1) * N;
                up10 = (pb + (size_t)y1 * d.res2_w + x0) * N;
                up11 = (pb + (size_t)y1 * d.res2_w + x1) * N;
            }
        }
#pragma unroll
        for (int i = 0; i < TNE; ++i) {
            const int n = n_of(i);
            if (!mv || n >= N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if constexpr (PRE) {
                const float4 b4 = bias_pre[i], r4 = res1_pre[i][j];   // zeros when absent; same (acc + bias) + res1 order as the other branch
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            } else {
                if constexpr (PREB) {
                    const float4 b4 = bias_pre[i];
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                } else if (d.bias) {
                    const float4 b4 = *reinterpret_cast<const float4*>(d.bias + n);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
                if (d.res1) {
                    const float4 r4 = *reinterpret_cast<const float4*>(d.res1 + orow + n);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
            }
            if (d.res2 && d.res2_h) {
                const float4 a00 = *reinterpret_cast<const float4*>(d.res2 + up00 + n), a01 = *reinterpret_cast<const float4*>(d.res2 + up01 + n);
                const float4 a10 = *reinterpret_cast<const float4*>(d.res2 + up10 + n), a11 = *reinterpret_cast<const float4*>(d.res2 + up11 + n);
                const float hy = 1.f - uly, hx = 1.f - ulx;
                v[0] += hy * (hx * a00.x + ulx * a01.x) + uly * (hx * a10.x + ulx * a11.x);
                v[1] += hy * (hx * a00.y + ulx * a01.y) + uly * (hx * a10.y + ulx * a11.y);
                v[2] += hy * (hx * a00.z + ulx * a01.z) + uly * (hx * a10.z + ulx * a11.z);
                v[3] += hy * (hx * a00.w + ulx * a01.w) + uly * (hx * a10.w + ulx * a11.w);
            } else if (d.res2) {
                const float4 r4 = *reinterpret_cast<const float4*>(d.res2 + orow + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if constexpr (ST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { gsum[i][r] += v[r]; gsq[i][r] = fmaf(v[r], v[r], gsq[i][r]); }
            }
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = d.act == ACT_RELU ? fmaxf(v[r], 0.f) : (d.act == ACT_GELU ? (sizeof(T) == 2 ? gelu_fast(v[r]) : gelu_erf(v[r])) : v[r]);
            }
            if (d.out_f32) {
                const float* s = d.act_on_f32 ? a : v;
                *reinterpret_cast<float4*>(d.out_f32 + orow + n) = make_float4(s[0], s[1], s[2], s[3]);
            }
            if (d.out_op) {
                if constexpr (sizeof(T) == 2) {
                    if (F16 && d.out_fmt == 3) x3_store4(d.out_op, (d.out_halo ? hrow : orow) + n, a[0], a[1], a[2], a[3]);   // the next launch reads x3 operands
                    else {
                        uint2 p;
                        p.x = pack_h2<F16>(a[0], a[1]);
                        p.y = pack_h2<F16>(a[2], a[3]);
                        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + (d.out_halo ? hrow : orow) + n) = p;
                    }
                } else if constexpr (X3) {
                    if (d.out_op_f32) *reinterpret_cast<float4*>(static_cast<float*>(d.out_op) + (d.out_halo ? hrow : orow) + n) = make_float4(a[0], a[1], a[2], a[3]);
                    else if (d.out_fmt == 1) {   // the next launch reads fp16 operands
                        uint2 p;
                        p.x = pack_h2<true>(a[0], a[1]);
                        p.y = pack_h2<true>(a[2], a[3]);
                        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(d.out_op) + (d.out_halo ? hrow : orow) + n) = p;
                    } else x3_store4(d.out_op, (d.out_halo ? hrow : orow) + n, a[0], a[1], a[2], a[3]);
                } else {
                    *reinterpret_cast<float4*>(static_cast<float*>(d.out_op) + (d.out_halo ? hrow : orow) + n) = make_float4(a[0], a[1], a[2], a[3]);
                }
            }
            if (d.out_dot) {
                const float4 w4 = *reinterpret_cast<const float4*>(d.dot_w + n);
                dot_part[j] += a[0] * w4.x + a[1] * w4.y + a[2] * w4.z + a[3] * w4.w;
            }
        }
    }
    if constexpr (ST) {
        // ---- GroupNorm statistics.  (1) in-wave: the 16 lanes that share (lane >> 4) hold the same 4 channels of 16 different pixels.
        // (2) per-channel sums of the WM wave rows meet in LDS (the staging ring is free after the barrier), (3) one thread per group adds
        // its gn_cpg channels in a fixed order and publishes the tile's partial with an L2-bypassing store, (4) the last workgroup of the
        // sample to arrive adds the tile partials in tile order in f64 and writes {mean, rstd}.  Fixed orders everywhere: bitwise
        // reproducible; no workgroup waits for another one.
#pragma unroll
        for (int i = 0; i < TNE; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = gsum[i][r], q = gsq[i][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
                gsum[i][r] = a; gsq[i][r] = q;
            }
        float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]
        __syncthreads();
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < TNE; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = wn * TNE * 16 + i * 16 + (lane >> 4) * 4 + r;
                    red[(wm * BN + ch) * 2] = gsum[i][r];
                    red[(wm * BN + ch) * 2 + 1] = gsq[i][r];
                }
        }
        __syncthreads();
        const int cpg = d.gn_cpg, G = N / cpg, gpt = BN / cpg;   // groups in total / per n-tile
        if (tid < gpt && n0 + tid * cpg < N) {
            float a = 0.f, q = 0.f;
            for (int c = 0; c < cpg; ++c)
#pragma unroll
                for (int w = 0; w < C::WM; ++w) { a += red[(w * BN + tid * cpg + c) * 2]; q += red[(w * BN + tid * cpg + c) * 2 + 1]; }
            float* pp = d.gn_part + ((size_t)mt * G + n0 / cpg + tid) * 2;
            __hip_atomic_store(pp, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tps = d.gn_hw / BM;                 // M tiles per sample (host: gn_hw % BM == 0)
        const int sample = mt / tps;
        unsigned* arrival = reinterpret_cast<unsigned*>(smem) + 2 * C::WM * BN;
        if (tid == 0) *arrival = __hip_atomic_fetch_add(d.gn_count + sample, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*arrival == (unsigned)(tps * ntiles) - 1u) {
            if (tid == 0) __hip_atomic_store(d.gn_count + sample, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // All threads share the walk over the sample's tps tile partials: thread (chunk = tid / G, group = tid % G) adds tiles chunk,
            // chunk + nch, ... (four loads in flight at a time), then the chunks are added in chunk order: a fixed order for a given shape,
            // so the statistics stay bitwise reproducible.  (One thread per group walking all tps partials with dependent L2-bypassing
            // loads kept this workgroup alive for ~50 us at 72 tiles per sample -- most of the launch: r03 autotune, ResNetV2 stage 0.)
            double* red64 = reinterpret_cast<double*>(smem + ((2 * C::WM * BN + 2) * 4 + 7) / 8 * 8);
            const int nch = C::THREADS / G > 0 ? C::THREADS / G : 1;
            const int g = tid % G, ch = tid / G;
            if (ch < nch && G <= C::THREADS) {
                double a = 0.0, q = 0.0;
                const float* base = d.gn_part + ((size_t)sample * tps * G + g) * 2;
                int t = ch;
                for (; t + 3 * nch < tps; t += 4 * nch) {
                    float va[4], vq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* pp = base + (size_t)(t + u * nch) * G * 2;
                        va[u] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vq[u] = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a += (double)va[u]; q += (double)vq[u]; }
                }
                for (; t < tps; t += nch) {
                    const float* pp = base + (size_t)t * G * 2;
                    a += (double)__hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q += (double)__hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                red64[(ch * G + g) * 2] = a;
                red64[(ch * G + g) * 2 + 1] = q;
            }
            __syncthreads();
            if (tid < G) {
                double a = 0.0, q = 0.0;
                for (int c2 = 0; c2 < nch; ++c2) { a += red64[(c2 * G + tid) * 2]; q += red64[(c2 * G + tid) * 2 + 1]; }
                const double cnt = (double)d.gn_hw * cpg;
                const double mean = a / cnt;
                double var = q / cnt - mean * mean;
                var = var > 0.0 ? var : 0.0;
                d.gn_stats[((size_t)sample * G + tid) * 2] = (float)mean;
                d.gn_stats[((size_t)sample * G + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)d.gn_eps));
            }
        }
    }
    if (GEN && d.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) d.stamps[4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
    if (MF == 16 && d.out_dot) {  // host guarantees WN == 1 and N <= BN: the whole channel range is in this wave
#pragma unroll
        for (int j = 0; j < TME; ++j) {
            float s = dot_part[j];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const int m = m_of(j);
            if ((lane >> 4) == 0 && m < d.M) d.out_dot[m] = fmaxf(s + d.dot_b, 0.f);
        }
    }
    }  // generic epilogue
}

// does the descriptor use what only the GEN instantiations implement?
static bool need_gen(const IgemmDesc& d) {
    return d.stride != 1 || d.pad != 1 || d.in_halo != 1 || d.Hi || d.Wi || d.gather1 || d.grp_rows || d.seg2_k || d.stamps || d.wt_grp_rows;
}

// deferred split-K: out = sum over the splits (in order) of the partial tiles, 16 bytes per thread and iteration
__global__ __launch_bounds__(256) void sk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = *reinterpret_cast<const float4*>(part + i * 4);
        for (int sp = 1; sp < splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)sp * n4 + i) * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i * 4) = s;
    }
}

template <class C, typename T, bool LN = false, bool SK = false, bool ST = false, bool GEN = false>
static int launch_cfg_t(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (!GEN && need_gen(d)) { err = "igemm: this configuration has no generalised-addressing instantiation (use 2, 20 or a GroupNorm-statistics launch)"; return 1; }
    if (LN != (d.ln_g != nullptr)) { err = "igemm: this configuration has no fused-LayerNorm instantiation"; return 1; }
    if (SK != (d.splitk > 1)) { err = "igemm: this configuration has no split-K instantiation"; return 1; }
    if (ST != (d.gn_stats != nullptr)) { err = "igemm: this configuration has no GroupNorm-statistics instantiation"; return 1; }
    constexpr int BK = C::ROWB / (int)sizeof(T);
    const int nk = d.taps * d.Cin / BK, kpt = d.Cin / BK;
    const int mtiles = (d.M + C::BM - 1) / C::BM, ntiles = (d.N + C::BN - 1) / C::BN;
    const size_t lds = (size_t)C::NS * C::STAGE;
    if constexpr (ST) {
        const int G = d.gn_cpg > 0 ? d.N / d.gn_cpg : 0;
        if (!d.gn_part || !d.gn_count || !d.out_f32 || d.gn_cpg <= 0 || d.N % d.gn_cpg || C::BN % d.gn_cpg || d.gn_hw <= 0 || d.gn_hw % C::BM || d.M % d.gn_hw ||
            (size_t)mtiles * G * 2 > d.gn_part_floats || (size_t)(d.M / d.gn_hw) > d.gn_count_words || G > C::THREADS ||
            (size_t)(2 * C::WM * C::BN + 4) * 4 + (size_t)C::THREADS * 16 > lds) {
            err = "igemm: bad GroupNorm-statistics descriptor (pixels per sample must be a multiple of the M tile)";
            return 1;
        }
    }
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<C, T, LN, SK, ST, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { err = std::string("igemm: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    const int splits = SK ? d.splitk : 1;
    if (SK && (!d.sk_part || !d.sk_count || splits > nk || (size_t)splits * d.M * d.N > d.sk_part_floats ||
               (size_t)mtiles * ntiles > d.sk_count_words)) { err = "igemm: bad split-K descriptor (scratch too small?)"; return 1; }
    if (SK && d.sk_defer && (!d.out_f32 || d.out_op || d.bias || d.res1 || d.res2 || d.act || (d.N & 3))) { err = "igemm: deferred split-K writes out_f32 only"; return 1; }
    SOCCDPT_LAUNCH((igemm_kernel<C, T, LN, SK, ST, GEN>), dim3((unsigned)(mtiles * ntiles * splits)), dim3(C::THREADS), lds, stream, d, nk, kpt, ntiles);
    if (SK && d.sk_defer) {
        const size_t n4 = (size_t)d.M * d.N / 4;
        size_t blocks = (n4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        SOCCDPT_LAUNCH(sk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d.sk_part, d.out_f32, splits, n4);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("igemm launch: ") + hipGetErrorString(e); return 1; }
    return 0;
}

template <class C>
static int launch_cfg(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    return d.f16 ? launch_cfg_t<C, f16_t>(d, stream, err) : launch_cfg_t<C, bf16_t>(d, stream, err);
}
// configurations that also carry the split-K instantiation
template <class C>
static int launch_cfg_sk(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (d.splitk <= 1) return launch_cfg<C>(d, stream, err);
    return d.f16 ? launch_cfg_t<C, f16_t, false, true>(d, stream, err) : launch_cfg_t<C, bf16_t, false, true>(d, stream, err);
}
// configurations that also carry the GroupNorm-statistics epilogue
template <class C>
static int launch_cfg_st(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (!d.gn_stats) return launch_cfg<C>(d, stream, err);
    return d.f16 ? launch_cfg_t<C, f16_t, false, false, true, true>(d, stream, err) : launch_cfg_t<C, bf16_t, false, false, true, true>(d, stream, err);
}
// configurations 2 and 20 also carry the generalised addressing without statistics (and 20 with split-K)
template <class C>
static int launch_cfg_gen(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (d.gn_stats) return launch_cfg_st<C>(d, stream, err);
    if (!need_gen(d)) return launch_cfg<C>(d, stream, err);
    return d.f16 ? launch_cfg_t<C, f16_t, false, false, false, true>(d, stream, err) : launch_cfg_t<C, bf16_t, false, false, false, true>(d, stream, err);
}
// configurations that also carry the fused-LayerNorm epilogue (one n-tile covers the row)
template <class C>
static int launch_cfg_ln(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (!d.ln_g) return launch_cfg<C>(d, stream, err);
    return d.f16 ? launch_cfg_t<C, f16_t, true>(d, stream, err) : launch_cfg_t<C, bf16_t, true>(d, stream, err);
}

// Kernel configurations.  id: name                 tile        ring
//   0  igemm_bf16_128x128x64_s4   big N, K%64==0    4 stages (128 KB LDS, 1 block/CU)
//   1  igemm_bf16_128x128x64_s2                     2 stages ( 64 KB LDS, 2 blocks/CU)
//   2  igemm_bf16_64x64x64_s4     small M*N grids   4 stages ( 64 KB)
//   3  igemm_bf16_128x128x32_s4   K%64!=0 (C=96)    4 stages ( 64 KB)
//   4  igemm_bf16_64x64x32_s4
//   5  igemm_bf16_128x32x64_s4    N<=32 (depth head tail)
static const char* const kCfgNames[] = {"igemm_bf16_128x128x64_s4", "igemm_bf16_128x128x64_s2", "igemm_bf16_64x64x64_s4",
                                        "igemm_bf16_128x128x32_s4", "igemm_bf16_64x64x32_s4", "igemm_bf16_128x32x64_s4",
                                        "igemm_bf16_256x128x64_s2", "igemm_bf16_256x128x64_s3", "igemm_bf16_256x256x64_s2",
                                        "igemm_bf16_128x128x32_s3", "igemm_bf16_128x256x64_s2", "igemm_bf16_64x64x64_s6",
                                        "igemm_bf16_64x64x64_s8", "igemm_bf16_64x128x64_s4", "igemm_bf16_32x64x64_s6", "igemm_bf16_128x256x32_s3",
                                        "igemm_bf16_256x128x32_s3", "igemm_bf16_128x256x32_s4", "igemm_bf16_256x256x32_s3", "igemm_bf16_64x128x32_s4",
                                        "igemm_bf16_32x64x128_s3", "igemm_bf16_128x128x64_s2_w8", "igemm_bf16_32x64x128_s3_w8", "igemm_bf16_64x64x64_s4_w8", "igemm_bf16_128x128x32_s3_w8",
                                        "igemm_bf16_cfg25", "igemm_bf16_cfg26", "igemm_bf16_cfg27", "igemm_bf16_cfg28", "igemm_bf16_cfg29",
                                        "conv8p_bf16_256x256x64", "conv8p_bf16_128x256x64", "conv8p_bf16_256x128x64", "conv8p_var33", "conv8p_var34", "conv8p_var35", "conv8p_var36",
                                        "conv8p_var37", "conv8p_var38", "conv8p_var39",
                                        "igemm_bf16_128x128x64_s2_m32", "igemm_bf16_128x128x64_s2_w8_m32", "igemm_bf16_256x128x64_s2_m32", "igemm_bf16_128x256x64_s2_m32",
                                        "igemm_bf16_256x256x64_s2_m32", "igemm_bf16_128x128x64_s3_m32",
                                        "igemm_bf16_128x128x64_s2_w8_splitk"};

static const char* const kCfgNamesF32[] = {"igemm_f32_128x128x32_s2", "igemm_f32_64x64x32_s4", "igemm_f32_128x32x32_s4", "igemm_f32_128x128x32_s2_w8"};
static const char* const kCfgNamesX3[] = {"igemm_x3_128x128x32_s2", "igemm_x3_64x64x32_s4", "igemm_x3_128x32x32_s4", "igemm_x3_128x128x32_s2_w8",
                                          "igemm_x3_64x64x32_s4_w8", "igemm_x3_128x128x32_s3_w8", "igemm_x3_128x128x32_s4_w8", "igemm_x3_128x64x32_s3",
                                          "igemm_x3_128x128x64_s2_w8", "igemm_x3_64x64x64_s3", "igemm_x3_32x64x64_s3_w8", "igemm_x3_64x128x32_s3"};
constexpr int kNumCfgX3 = 12;
static int pick_cfg_f32(const IgemmDesc& d) {
    const bool tunable = !d.gn_stats && !need_gen(d) && !d.ln_g && d.N > 32 && d.splitk <= 1;
    if (d.x3 && d.tune >= 0 && d.tune < kNumCfgX3 && tunable) {   // in-network tuning (x3 tiles)
        if (d.tune >= 8 && d.tune <= 10 && d.Cin % 64) return 1;   // 64-deep k-tiles
        return d.tune;
    }
    // x3 launches with the GroupNorm-statistics epilogue / generalised addressing: instantiated for configurations 0, 1, 3, 4, 7 (statistics) and 1, 4 (addressing only)
    if (d.x3 && d.tune >= 0 && d.splitk <= 1 && !d.ln_g && (d.gn_stats || need_gen(d))) {
        const int t = d.tune;
        const bool gen = need_gen(d);
        const int bm = (t == 0 || t == 3 || t == 7) ? 128 : 64, bn = (t == 0 || t == 3) ? 128 : 64;
        if (d.gn_stats && (t == 0 || t == 1 || t == 3 || t == 4 || t == 7) && d.gn_hw % bm == 0 && bn % d.gn_cpg == 0 && !(gen && t != 1 && t != 4)) return t;
        if (!d.gn_stats && gen && (t == 1 || t == 4)) return t;
    }
    if (d.x3 && d.gn_stats && d.splitk <= 1 && !need_gen(d)) {
        // in-network timings of the ResNetV2 convolutions of dpt_hybrid_384 (profiles/r03_autotune_x3_hyb_st.txt): the 8-wave 64 x 64 tile
        // wins nearly everywhere (the statistics epilogue is per-wave work, and these launches are short of workgroups)
        const long b64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
        if (d.gn_hw % 128 == 0 && 128 % d.gn_cpg == 0 && d.N >= 256 && d.M >= 32768) return 3;
        if (d.gn_hw % 64 == 0 && 64 % d.gn_cpg == 0 && !(d.taps == 9 && b64 >= 256 && b64 < 512)) return 4;
    }
    if (d.gn_stats) return (d.gn_hw % 128 == 0 && (long)((d.M + 127) / 128) * ((d.N + 127) / 128) >= 256) ? 0 : 1;
    if (need_gen(d)) return 1;   // the generalised addressing is instantiated for the 64 x 64 f32 tile
    if (d.ln_g) return 0;  // 128x128 covers N <= 128 (host only fuses LayerNorm for N <= 128 in f32 mode)
    if (d.N <= 32) return 2;
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long b128 = cdiv(d.M, 128) * cdiv(d.N, 128);
    if (d.x3 && d.splitk <= 1) {
        // Rules read off IN-NETWORK timings of every x3 tile at every launch site of the three models (tools/autotune_network.py ... f16x3,
        // profiles/r03_autotune_x3_*.txt).  The x3 tiles carry 3 MFMAs per 2x the staged bytes of an fp16 tile, so what matters is (1) enough
        // workgroups for the 256 CUs, (2) waves per SIMD to overlap LDS-DMA issue with the other wave's MFMAs: the 8-wave 64 x 64 tile replaces
        // the 4-wave one everywhere except the mid-size 3x3 convolutions, 32 x 64 tiles with 64-deep k-steps take the small grids, and the
        // large convolutions want two resident workgroups (128 x 64, 3 stages) or the 64-deep 128 x 128 tile.
        const bool k64 = d.Cin % 64 == 0;
        const long K = (long)d.taps * d.Cin, b64 = cdiv(d.M, 64) * cdiv(d.N, 64), t128x64 = cdiv(d.M, 128) * cdiv(d.N, 64);
        if (d.taps == 9 && b128 >= 2048 && k64) return 8;
        if (d.taps == 9 && b128 >= 384) return 7;
        if (d.taps == 1 && K >= 768 && b128 >= 256 && k64) return 8;
        if (d.taps == 1 && K >= 1536 && t128x64 >= 256) return 7;
        if (b64 <= 256 && k64) return 10;
        return d.taps == 9 ? 1 : 4;
    }
    // 8 waves (64 x 32 per wave) like the bf16 form: f32 forward 1093 -> 1126 frames/s, training step 45.0 -> 44.2 ms, alternated in one GPU call
    return b128 >= 384 ? 3 : 1;
}

static int pick_cfg(const IgemmDesc& d) {
    const bool k64 = (d.Cin % 64 == 0);
    if (d.tune < 0 && d.ln_g) return k64 ? 13 : 19;  // fused LayerNorm: the whole row (N <= 128) in one 64(M) x 128(N) tile
    // History of this heuristic: first ranked by event timing from Python (host-bound below ~10 us: useless for the small
    // launches), then by rocprofv3 device durations of repeated launches (tools/igemm_tune.py; warm caches flatter big
    // one-workgroup-per-CU tiles and tiles that re-read weights), finally by timing every candidate inside the forward.
    if (d.tune >= 0) return d.tune;
    if (d.gn_stats) {   // GroupNorm-statistics epilogue: instantiated for 128x128x64 (8 waves), 64x64x64 (8 waves) and 64x64x32
        if (!k64) return 4;
        const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
        return (d.gn_hw % 128 == 0 && d.gn_cpg <= 128 && t128 >= 256) ? 21 : 23;
    }
    if (need_gen(d)) {   // generalised addressing without statistics (ViT read-out projection, Conv2d(3, 2, 1) of act_postprocess4): configurations 2 / 20
        if (d.splitk > 1) return 20;
        const long t64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
        return (d.Cin % 128 == 0 && t64 < 256) ? 20 : 2;
    }
    if (d.splitk > 1) return d.Cin % 128 == 0 ? 20 : 14;  // the split-K instantiations: 32(M) x 64(N) tiles
    if (d.N <= 32) return 5;
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long K = (long)d.taps * d.Cin;
    const long b128 = cdiv(d.M, 128) * cdiv(d.N, 128), b64 = cdiv(d.M, 64) * cdiv(d.N, 64);
    if (!k64) return (d.taps == 9 && b128 >= 384 && d.N % 256 == 0) ? 24 : 4;  // C = 96: layer1_rn (128x128x32, 8 waves) / stage-0 Linear layers (64x64x32)
    // The rules below were re-derived from IN-NETWORK timings of every candidate at every launch site of both models
    // (tools/autotune_network.py, profiles/r01j_autotune_in_network_*.txt).  In the real launch sequence weights and activations
    // arrive cold, and tiles that keep TWO workgroups per CU (128x128x64 s2, 256x128x32 s3, 64x64) beat the one-workgroup-per-CU
    // tiles (128x256x64, 256x256x64) that win a warm repeated-launch benchmark: the second workgroup hides the cold misses.
    // Long-K 3x3 convs that land on 128x128 tiles: the 8-wave variant (64x32 per wave, 16 waves per CU instead of 8) hides the
    // LDS-fragment / MFMA-issue stalls better -- 64^2 RCU convs 229 -> 209 us, depth-head conv 104 -> 92 us, base_384 96^2 convs
    // 548 -> 475 us in the network (old / new library alternated inside one GPU call); K <= 1152 keeps the 4-wave tile (78 vs 82 us).
    const int c128 = (d.taps == 9 && K >= 1152) ? 21 : 1;
    if (d.res2_h && b128 >= 384) return c128;  // sampled-residual epilogue (4 gathers per output)
    if (d.N % 256 == 0 && cdiv(d.M, 256) * (d.N / 256) >= 448) return (d.taps == 9 && K >= 1536) ? 21 : 16;   // head-sized problems: 256x128x32, or the 8-wave 128x128x64 for the long-K 3x3 (seg head 159 -> 152 us)   // head-sized convs: 256x128 tiles, 32-deep, 3 stages
    // short K, many output tiles (qkv / fc1 / proj / merge): write-heavy; 32-deep k-tiles halve the LDS footprint -> 5 blocks per CU
    // plain Linear layers / 1x1 convs with many 128x128 tiles and no GELU epilogue (qkv, out_conv): the 8-wave 128x128 tile again
    // (base_384 stage-2 qkv, 18 launches: 412 -> 338 us in the network; with the GELU epilogue of fc1 it loses to 64x64x32)
    if (d.taps == 1 && d.act != ACT_GELU && K >= 128 && K <= 1024 && b128 >= 384) return 21;
    // ViT-B Linear layers of dpt_hybrid_384 (M = B * 577, K = 768): 128x128 tiles already pay from 256 tiles on, GELU epilogue or not
    // (in-network, B = 4: qkv 350 -> 261 us, fc1 452 -> 320 us per 12 launches; profiles/r02e_autotune_in_network_hybrid384.txt)
    if (d.taps == 1 && K >= 768 && K <= 1024 && b128 >= 256) return 21;
    if (K <= 1024 && b64 >= 512) return 4;
    if (b128 >= 256) return c128;
    // small grids: halve the M tile (2x the workgroups) and use 128-deep k-tiles (half the barriers: 8-15 % over a 64-deep
    // 6-stage ring).  A two-stage variant and a wider use of 32-row tiles both win warm and lose in the network (doubled weight
    // re-reads): 3712 -> 3580 frames/s, reverted.
    if (d.Cin % 128 == 0 && ((b64 < 256 && K >= 384) || (b64 <= 128 && K >= 256))) return 22;   // (4-wave form: 20, kept for split-K)
    if ((b64 < 256 && K >= 1536) || (b64 < 128 && K >= 768)) return 14;
    return 23;  // 64x64 (8 waves; the 4-wave form is configuration 2): 4x the blocks of 128x128
}

int igemm_pick_splitk(const IgemmDesc& d, size_t part_floats, size_t count_words) {
    // Measured (tools/igemm_tune.py, rocprofv3 durations): the partial-tile exchange costs ~5 us per split (L2-bypassing stores
    // and loads: the 8 XCD L2s are not coherent with each other inside a kernel), about one kernel floor.  It only pays for the
    // longest K on the smallest grid: layer4_rn (M=512, N=256, K=6912) 30.9 -> 18.3 us at 4 splits; K=2304..3456 on 64..256
    // tiles and the stage-3 Linear layers (K <= 3072) are break-even or slower and stay unsplit.
    if (d.f32 || d.x3) {   // exact-f32 / x3 modes (64 x 64 tiles, 32-deep k-tiles): the same small-grid long-K launches, measured in the f32 forward and the training step
        // alternated with the unsplit build inside one GPU call: tiny_256 f32 forward 1129 -> 1154 frames/s, hybrid_384 f32 136.0 -> 137.0
        if (d.ln_g || d.out_dot || d.gn_stats || need_gen(d) || d.N <= 32 || d.tune >= 0) return 1;
        auto cdiv32 = [](long a, long b) { return (a + b - 1) / b; };
        const long nk32 = (long)d.taps * d.Cin / 32, blocks64 = cdiv32(d.M, 64) * cdiv32(d.N, 64);
        if (blocks64 > 96 || nk32 < 48 || (size_t)blocks64 > count_words) return 1;
        // x3: the 32 x 64 tile with 64-deep k-steps (configuration 10) beats every split of these launches up to K ~ 3500 (r04 per-site timings of the
        // forward: M 512, N 768, K 1536: 34.4 us split vs 13.6; K 3072: 81.5 vs 47.9; 3x3 N 256, K 2304: 59 vs 37 per two launches); only the
        // K = 6912 convolution still wins split (34 vs 49)
        if (d.x3 && nk32 < 160) return 1;
        long S = (256 + blocks64 - 1) / blocks64;
        if (S > nk32 / 8) S = nk32 / 8;
        if (S > 16) S = 16;
        while (S > 1 && (size_t)S * d.M * d.N > part_floats) --S;
        return (int)(S < 1 ? 1 : S);
    }
    if (d.ln_g || d.out_dot || d.gn_stats || d.seg2_k || d.N <= 32 || d.Cin % 64 != 0 || d.tune >= 0) return 1;
    if (need_gen(d) && d.Cin % 128 != 0) return 1;   // split-K with generalised addressing exists for the 128-deep tile only
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long nk = (long)d.taps * d.Cin / 64, blocks = cdiv(d.M, 32) * cdiv(d.N, 64);
    if (blocks > 128 || nk < 96 || (size_t)blocks > count_words) return 1;
    long S = 4;
    while (S > 1 && (size_t)S * d.M * d.N > part_floats) --S;
    return (int)S;
}

int igemm_config_id(const IgemmDesc& d) { return d.x3 ? pick_cfg_f32(d) : (d.f32 ? -1 : pick_cfg(d)); }

const char* igemm_family(const IgemmDesc& d) {
    if (d.x3) return d.splitk > 1 ? (d.tune == 3 || d.tune == 8 ? "igemm_x3_128x128_w8_splitk" : "igemm_x3_64x64x32_s4_splitk") : kCfgNamesX3[pick_cfg_f32(d)];
    if (d.f32) return d.splitk > 1 ? (d.tune == 3 ? "igemm_f32_128x128x32_s2_w8_splitk" : "igemm_f32_64x64x32_s4_splitk") : kCfgNamesF32[pick_cfg_f32(d)];
    const int id = pick_cfg(d);
    if (d.splitk > 1) {
        if (id == 46) return d.f16 ? "igemm_f16_128x128x64_s2_w8_splitk" : "igemm_bf16_128x128x64_s2_w8_splitk";
        if (id == 20) return d.f16 ? "igemm_f16_32x64x128_s3_splitk" : "igemm_bf16_32x64x128_s3_splitk";
        return d.f16 ? "igemm_f16_32x64x64_s6_splitk" : "igemm_bf16_32x64x64_s6_splitk";
    }
    if (!d.f16) return kCfgNames[id];
    static std::string f16_names[sizeof(kCfgNames) / sizeof(kCfgNames[0])];  // "igemm_f16_<tile>": same kernels, fp16 instantiation
    if (f16_names[id].empty()) f16_names[id] = std::string("igemm_f16_") + (kCfgNames[id] + 11);
    return f16_names[id].c_str();
}

int launch_igemm(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (d.M <= 0 || d.N <= 0 || d.Cin <= 0 || !d.X || !d.Wt) { err = "igemm: bad descriptor"; return 1; }
    if (d.N % 4 != 0) { err = "igemm: N must be a multiple of 4"; return 1; }
    if (d.Cin % 32 != 0) { err = "igemm: Cin must be a multiple of 32"; return 1; }
    if (d.taps != 1 && d.taps != 9) { err = "igemm: taps must be 1 or 9"; return 1; }
    if ((d.taps == 9 || d.gather1) && (d.H <= 0 || d.W <= 0 || d.M % (d.H * d.W) != 0)) { err = "igemm: bad conv geometry"; return 1; }
    if (d.taps == 9 || d.gather1) {   // every tap of every output pixel must stay inside the (haloed) input image
        const int Hi = d.Hi ? d.Hi : d.H, Wi = d.Wi ? d.Wi : d.W, k = d.taps == 9 ? 3 : 1;
        const int lo = d.in_halo - d.pad, hiy = (d.H - 1) * d.stride + k - 1 + lo, hix = (d.W - 1) * d.stride + k - 1 + lo;
        if (d.stride < 1 || lo < 0 || hiy >= Hi + 2 * d.in_halo || hix >= Wi + 2 * d.in_halo) { err = "igemm: conv taps leave the input image"; return 1; }
    }
    if (d.seg2_k && (d.taps != 1 || d.gather1 || !d.grp_rows || d.seg2_k % 128 != 0 || d.seg2_k >= d.Cin)) { err = "igemm: bad second-segment descriptor"; return 1; }
    if (d.grp_rows && (d.taps != 1 || d.gather1)) { err = "igemm: row groups are a plain-mode feature"; return 1; }
    if (d.wt_grp_rows && (d.taps != 1 || d.gather1 || d.wt_grp_rows % 64 != 0 || d.N > 9 * d.wt_grp_rows || d.N % d.wt_grp_rows != 0 ||
                          (!d.wt_kx && d.wt_base - d.wt_rp - 1 + (d.wt_odd < 0 ? d.wt_odd : 0) < 0))) { err = "igemm: bad weight row-group descriptor"; return 1; }
    if ((d.out_halo || d.res2_h) && (d.H <= 0 || d.W <= 0)) { err = "igemm: halo output / sampled residual need H, W"; return 1; }
    if (d.out_dot && d.N > 32) { err = "igemm: fused dot tail needs N <= 32"; return 1; }
    if (d.ln_g && (!d.ln_b || !d.ln_xf || d.N > 128 || (d.ln_halo && (d.H <= 0 || d.W <= 0)))) { err = "igemm: bad fused-LayerNorm descriptor"; return 1; }
    if (d.N <= 32 && d.Cin % 64 != 0 && !d.f32 && !d.x3) { err = "igemm: N <= 32 needs Cin % 64 == 0"; return 1; }
    if (d.x3) {   // split-fp16 operands (SOCCDPT_PREC_F16X3): the f32 tile set with T = x3_t
        if (d.wt_grp_rows && (!d.wt_kx || d.wt_kx % 16 || d.wt_base % 16 || d.wt_rp % 16 || d.wt_base - d.wt_rp < 0)) {
            err = "igemm: x3 weight-row views must start at multiples of 16 elements (wt_kx copies)";
            return 1;
        }
        if (d.ldx % 16 || d.Cin % 32 || (d.out_op && !d.out_op_f32 && d.out_fmt != 1 && d.N % 16) || d.grp_off % 16 || d.seg2_off % 16 || d.grp_stride % 16) { err = "igemm: x3 rows must start at multiples of 16 elements"; return 1; }
        if (d.splitk > 1) {
            if (d.ln_g || d.gn_stats || d.out_dot) { err = "igemm: x3 split-K has no LayerNorm, statistics or dot epilogue"; return 1; }
            if (need_gen(d) && d.tune == 3) return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t, false, true, false, true>(d, stream, err);
            if (need_gen(d)) return launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, true, false, true>(d, stream, err);
            // tune 3 / 8: the 8-wave 128 x 128 tiles (32- / 64-deep k-tiles) for the long-K weight-gradient GEMMs of the training step: four times
            // the MFMA work per staged byte of the 64 x 64 tile, which the per-CU L2 -> LDS fill rate bounds (train_step.cpp: gemm_wgrad)
            if (d.tune == 8 && d.Cin % 64 == 0) return launch_cfg_t<Cfg<128, 128, 128, 2, 4, 2>, x3_t, false, true>(d, stream, err);
            if (d.tune == 3) return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t, false, true>(d, stream, err);
            return launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, true>(d, stream, err);
        }
        switch (pick_cfg_f32(d)) {
            case 0: return d.gn_stats ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t, false, false, true, true>(d, stream, err)
                         : d.ln_g ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t, true>(d, stream, err)
                                  : launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t>(d, stream, err);
            case 1: return d.gn_stats ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, false, true, true>(d, stream, err)
                         : need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, false, false, true>(d, stream, err)
                                       : launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t>(d, stream, err);
            case 3: return d.gn_stats ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t, false, false, true, true>(d, stream, err)
                                      : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t>(d, stream, err);
            case 4: return d.gn_stats ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t, false, false, true, true>(d, stream, err)      // 8 waves, 32 x 16 per wave
                         : need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t, false, false, false, true>(d, stream, err)
                                       : launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t>(d, stream, err);
            case 5: return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 3>, x3_t>(d, stream, err);    // 3-stage ring (96 KB)
            case 6: return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 4>, x3_t>(d, stream, err);    // 4-stage ring (128 KB)
            case 7: return d.gn_stats ? launch_cfg_t<Cfg<128, 64, 64, 2, 2, 3>, x3_t, false, false, true, true>(d, stream, err)
                                      : launch_cfg_t<Cfg<128, 64, 64, 2, 2, 3>, x3_t>(d, stream, err);     // 4 waves, 64 x 32 per wave, two workgroups per CU
            case 8: return launch_cfg_t<Cfg<128, 128, 128, 2, 4, 2>, x3_t>(d, stream, err);   // 64-deep k-tiles (256-byte rows): half the barriers
            case 9: return launch_cfg_t<Cfg<64, 64, 128, 2, 2, 3>, x3_t>(d, stream, err);
            case 10: return launch_cfg_t<Cfg<32, 64, 128, 2, 4, 3>, x3_t>(d, stream, err);    // small grids, long K
            case 11: return launch_cfg_t<Cfg<64, 128, 64, 2, 2, 3>, x3_t>(d, stream, err);
            default: return launch_cfg_t<Cfg<128, 32, 64, 4, 1, 4>, x3_t>(d, stream, err);
        }
    }
    if (d.f32) {  // exact-f32 operands (SOCCDPT_PREC_F32): 128-byte rows hold 32 elements, Cin % 32 == 0 suffices
        // split-K in f32: the weight-gradient GEMMs of the training step (K = pixels, a handful of output tiles; train_step.cpp picks the split)
        if (d.splitk > 1) {
            if (d.ln_g || d.gn_stats || d.out_dot) { err = "igemm: f32 split-K has no LayerNorm, statistics or dot epilogue"; return 1; }
            if (d.tune == 3)   // 8-wave 128 x 128 tile (with sk_defer: the last-arriver reduction of many big partial tiles was what made it lose in round 2)
                return need_gen(d) ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, float, false, true, false, true>(d, stream, err)
                                   : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, float, false, true>(d, stream, err);
            if (need_gen(d)) return launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, float, false, true, false, true>(d, stream, err);   // weight row groups (training wgrad)
            return launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, float, false, true>(d, stream, err);
        }
        switch (pick_cfg_f32(d)) {
            case 0: return d.gn_stats ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, float, false, false, true, true>(d, stream, err)
                         : d.ln_g ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, float, true>(d, stream, err)
                                  : launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, float>(d, stream, err);
            case 1: return d.gn_stats ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, float, false, false, true, true>(d, stream, err)
                         : need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, float, false, false, false, true>(d, stream, err)
                                       : launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, float>(d, stream, err);
            case 3: return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, float>(d, stream, err);   // 8 waves, 64 x 32 per wave
            default: return launch_cfg_t<Cfg<128, 32, 64, 4, 1, 4>, float>(d, stream, err);
        }
    }
    const int id = pick_cfg(d);
    const bool k64 = (d.Cin % 64 == 0);
    if (id >= 30 && id <= 39) return launch_conv8p(d, id - 30, stream, err);
    if ((id == 0 || id == 1 || id == 2 || id == 5 || id == 6 || id == 7 || id == 8 || (id >= 10 && id <= 14) || (id >= 21 && id <= 23) || (id >= 40 && id <= 46)) && !k64) { err = "igemm: this configuration needs Cin % 64 == 0"; return 1; }
    if ((id == 20 || id == 22) && d.Cin % 128 != 0) { err = "igemm: this configuration needs Cin % 128 == 0"; return 1; }
    if (d.out_dot && id != 5) { err = "igemm: fused dot tail needs the 128x32 configuration"; return 1; }
    switch (id) {
        case 0: return launch_cfg<Cfg<128, 128, 64, 2, 2, 4>>(d, stream, err);
        case 1: return launch_cfg_st<Cfg<128, 128, 64, 2, 2, 2>>(d, stream, err);
        case 2: return launch_cfg_gen<Cfg<64, 64, 64, 2, 2, 4>>(d, stream, err);
        case 3: return launch_cfg<Cfg<128, 128, 32, 2, 2, 4>>(d, stream, err);
        case 4: return launch_cfg_st<Cfg<64, 64, 32, 2, 2, 4>>(d, stream, err);
        case 5: return launch_cfg<Cfg<128, 32, 64, 4, 1, 4>>(d, stream, err);
        case 6: return launch_cfg<Cfg<256, 128, 64, 4, 2, 2>>(d, stream, err);
        case 7: return launch_cfg<Cfg<256, 128, 64, 4, 2, 3>>(d, stream, err);
        case 8: return launch_cfg<Cfg<256, 256, 64, 2, 4, 2>>(d, stream, err);
        case 9: return launch_cfg<Cfg<128, 128, 32, 2, 2, 3>>(d, stream, err);
        case 10: return launch_cfg<Cfg<128, 256, 64, 2, 4, 2>>(d, stream, err);
        case 11: return launch_cfg<Cfg<64, 64, 64, 2, 2, 6>>(d, stream, err);
        case 12: return launch_cfg<Cfg<64, 64, 64, 2, 2, 8>>(d, stream, err);
        case 13: return launch_cfg_ln<Cfg<64, 128, 64, 2, 2, 4>>(d, stream, err);
        case 14: return launch_cfg_sk<Cfg<32, 64, 64, 2, 2, 6>>(d, stream, err);
        case 15: return launch_cfg<Cfg<128, 256, 32, 2, 4, 3>>(d, stream, err);
        case 16: return launch_cfg<Cfg<256, 128, 32, 4, 2, 3>>(d, stream, err);
        case 17: return launch_cfg<Cfg<128, 256, 32, 2, 4, 4>>(d, stream, err);
        case 18: return launch_cfg<Cfg<256, 256, 32, 2, 4, 3>>(d, stream, err);
        case 19: return launch_cfg_ln<Cfg<64, 128, 32, 2, 2, 4>>(d, stream, err);
        case 21: return launch_cfg_st<Cfg<128, 128, 64, 2, 4, 2>>(d, stream, err);   // 8 waves, 64x32 per wave: twice the resident waves of configuration 1
        case 22: return launch_cfg<Cfg<32, 64, 128, 2, 4, 3>>(d, stream, err);   // 8 waves, 16x16 per wave: the small-grid long-K launches are latency chains,
        case 23: return launch_cfg_st<Cfg<64, 64, 64, 2, 4, 4>>(d, stream, err);    // 8 waves, 32x16 per wave:   twice the waves halve each wave's dependent MFMA chain
        case 24: return launch_cfg<Cfg<128, 128, 32, 2, 4, 3>>(d, stream, err);  // 8 waves, 64x32 per wave, 32-deep k-tiles (C = 96: layer1_rn 43 -> 34 us)
        // v_mfma_f32_32x32x16 forms: same tiles, 32 x 32 MFMA fragments
        case 40: return launch_cfg<Cfg<128, 128, 64, 2, 2, 2, 32>>(d, stream, err);   // 4 waves, 64 x 64 per wave (2 x 2 MFMAs)
        case 41: return launch_cfg<Cfg<128, 128, 64, 2, 4, 2, 32>>(d, stream, err);   // 8 waves, 64 x 32 per wave
        case 42: return launch_cfg<Cfg<256, 128, 64, 4, 2, 2, 32>>(d, stream, err);   // 8 waves, 64 x 64 per wave
        case 43: return launch_cfg<Cfg<128, 256, 64, 2, 4, 2, 32>>(d, stream, err);   // 8 waves, 64 x 64 per wave
        case 44: return launch_cfg<Cfg<256, 256, 64, 2, 4, 2, 32>>(d, stream, err);   // 8 waves, 128 x 64 per wave
        case 45: return launch_cfg<Cfg<128, 128, 64, 2, 2, 3, 32>>(d, stream, err);   // 4 waves, 3-stage ring
        // (tried in round 3 and removed: 128-deep k-tiles on the 128 x 128 tile, 8 and 4 waves -- half the barriers, but 128 KB of LDS = one workgroup per
        //  CU: 251 / 300 us against 198 us on the 64^2 256 -> 256 convolution in the network, 191 / 245 against 153 us on the seg-head convolution)
        case 46:   // 8-wave 128 x 128 x 64 with split-K (training: weight gradients of the wide layers -- long K, few output tiles; tune-selected only)
            if (d.splitk <= 1) { err = "igemm: configuration 46 is the split-K form"; return 1; }
            if (need_gen(d)) return d.f16 ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, f16_t, false, true, false, true>(d, stream, err)
                                          : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, bf16_t, false, true, false, true>(d, stream, err);
            return d.f16 ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, f16_t, false, true>(d, stream, err) : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, bf16_t, false, true>(d, stream, err);
        case 20:
            if (d.gn_stats || (need_gen(d) && d.splitk <= 1)) return launch_cfg_gen<Cfg<32, 64, 128, 2, 2, 3>>(d, stream, err);
            if (need_gen(d)) return d.f16 ? launch_cfg_t<Cfg<32, 64, 128, 2, 2, 3>, f16_t, false, true, false, true>(d, stream, err)
                                          : launch_cfg_t<Cfg<32, 64, 128, 2, 2, 3>, bf16_t, false, true, false, true>(d, stream, err);
            return launch_cfg_sk<Cfg<32, 64, 128, 2, 2, 3>>(d, stream, err);  // 128-deep k-tiles: half the barriers of the long-K small-grid launches
    }
    err = "igemm: unknown configuration id";
    return 1;
}

}  // namespace soccdpt
