// The implicit-GEMM tile of igemm.hip as a device function (igemm.hip: one workgroup = one tile; round 4 also drove it from a persistent
// kernel that walked the tiles of one phase after another: DESIGN.md section 10.2, removed in round 5).  Design notes: igemm.hip.
#pragma once
#include <type_traits>

#include "gelu.h"
#include "half16.h"
#include "igemm.h"

namespace soccdpt {

typedef __attribute__((ext_vector_type(4))) float f32x4;
struct f16_t { uint16_t v; };  // element tag of the fp16 instantiations (SOCCDPT_PREC_F16); bf16_t tags bf16, float exact f32
// x2w (round 5): ONE-SIDED split.  Activations are plain fp16 (2 bytes per element, staged and read exactly like an fp16 launch's), the WEIGHTS are x3
// pairs (hi, lo * 2^11; 4 bytes per element, prepared once), and a product is TWO fp16 MFMAs, hi x + 2^-11 lo x: the weight rounding -- a median 72 % of
// a launch site's fp16 rounding variance (tools/x2_variance_probe.py, profiles/r05_x2_variance_tiny256.json) -- is gone at 1.5x the operand bytes of an
// fp16 launch instead of x3's 2x, and the producers of the site's activations keep writing 2-byte operands.  The element tag is 2 bytes wide: every
// `sizeof(T) == 2` branch (activation addressing, 16-bit epilogue stores) is the fp16 one; the weight tile has its own geometry (rows of 2 * ROWB bytes).
struct x2w_t { uint16_t v; };

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

// bytes of one ring slot: X tile + W tile (the x2w weight tile holds 4-byte x3 pairs beside 2-byte activations)
template <class C, typename T>
constexpr int igemm_stage_bytes() { return C::X_BYTES + C::W_BYTES * (std::is_same<T, x2w_t>::value ? 2 : 1); }

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
// ST: GroupNorm statistics of the raw output (d.gn_stats): per-tile per-group partial sums {sum, sum of squares}; the reader of the output adds them up.
// ResNetV2 stages of the ViT-hybrid encoder (csrc/hybrid.hip applies the normalisation).
// GEN: the generalised addressing (strided / un-haloed / gathered convolution, row groups, second A segment) and the diagnostics stamps.
// A separate instantiation: carried by every launch they cost the Swin models 1.5 % of the forward (A/B in one GPU call, tools/ab_bench.sh).
// One output tile `bid` (logical id: n-tile fastest, split fastest of all under SK) of the launch described by d.  `smem`: C::NS * C::STAGE bytes
// of LDS, free on entry (callers that run several tiles in a row put a workgroup barrier between them).  tid: the thread's index in the workgroup
// (a parameter so that the persistent kernel can keep the per-thread index arithmetic of one phase from being hoisted over all phases).
// D3: three-class classifier tail fused into the epilogue (the seg head: Conv3x3 + BN + ReLU -> Conv1x1(N -> 3), model/SOccDPT.py:660-671): every
// workgroup reduces act(v) . dot_w[c][n0 .. n0 + BN) over its n-tile for its BM pixels and writes the three partial logits to
// out_dot[(n_tile * M + m) * 4 + c]; a finishing pass adds the n-tiles' partials in tile order and the bias.  The N-channel feature map is never stored.
template <class C, typename T, bool LN, bool SK = false, bool ST = false, bool GEN = false, bool D3 = false>
__device__ __forceinline__ void igemm_tile(const IgemmDesc& d, int nk, int kpt, int ntiles, int bid, char* smem, const int tid) {
    constexpr int BM = C::BM, BN = C::BN;
    constexpr int BK = C::ROWB / (int)sizeof(T);   // k-tile depth in elements of T
    constexpr int EPC = 16 / (int)sizeof(T);       // elements per 16-byte chunk
    constexpr bool X2W = std::is_same<T, x2w_t>::value;
    constexpr bool F16 = std::is_same<T, f16_t>::value || X2W;
    constexpr bool X3 = std::is_same<T, x3_t>::value;
    static_assert(!X3 || C::ROWB >= 128, "an x3 k-step (32 elements) is 128 bytes of a tile row");
    static_assert(!X2W || ((C::BK == 32 || C::BK == 64) && C::MF == 16), "x2w tiles: 32- or 64-deep k-tiles (64- / 128-byte activation rows, 128- / 256-byte weight rows)");
    // weight-tile geometry: that of the activations except under x2w (4-byte x3 pairs: rows, chunks per row and LDS-DMA pieces double)
    typedef typename std::conditional<X2W, x3_t, T>::type WT;
    constexpr int WMUL = X2W ? 2 : 1;
    constexpr int W_ROWB = C::ROWB * WMUL, W_CPR = C::CPR * WMUL, W_LOADS = C::W_LOADS * WMUL, W_EPC = 16 / (int)sizeof(WT);
    constexpr int STAGE = C::X_BYTES + C::W_BYTES * WMUL, LOADS = C::X_LOADS + W_LOADS;
    static_assert((C::NS - 2) * LOADS <= 63, "vmcnt immediate is 6 bits");
    constexpr int MF = C::MF;
    static_assert(MF == 16 || (MF == 32 && sizeof(T) == 2 && C::BK == 64 && !LN && !SK && !ST), "32x32x16 tiles: 16-bit operands, 64-deep k-tiles, plain epilogue");
    // epilogue view of a wave's accumulators, common to both MFMA shapes: TME m-tiles x TNE groups of 4 consecutive channels per lane;
    // m_of(j) / n_of(i) = the pixel / first channel a lane owns in group (i, j)
    constexpr int TME = MF == 32 ? BM / C::WM / 32 : C::TM;
    constexpr int TNE = MF == 32 ? (BN / C::WN / 32) * 4 : C::TN;
    const T* const Xp = static_cast<const T*>(d.X);
    const WT* const Wtp = static_cast<const WT*>(d.Wt);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    if (GEN && d.stamps && tid == 0) d.stamps[4 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();

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
    uint32_t x_off[C::X_LOADS], x_off2[C::X_LOADS], w_off[W_LOADS];
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
    for (int i = 0; i < W_LOADS; ++i) {
        const int cid = i * C::THREADS + tid;
        const int row = cid / W_CPR, c = cid % W_CPR;
        int n = n0 + row;
        n = n < d.N ? n : d.N - 1;
        if constexpr (X2W) {   // x3 pairs in rows of W_ROWB bytes: the swizzle of the x3 tiles of that row length
            w_off[i] = (uint32_t)n * (uint32_t)Ktot + (uint32_t)((c ^ swz_of_row<C::BK * 2, 16>(row)) * W_EPC);
        } else
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
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < C::X_LOADS; ++i) {
            const T* g = Xp + (x_off[i] + ((GEN && seg2) ? x_off2[i] : 0u)) + xk;   // a VALUE select: selecting between the two arrays demotes them (and d) to scratch
            char* l = sb + (i * C::THREADS + wave * 64) * 16;  // wave-uniform base; HW adds lane*16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W_LOADS; ++i) {
            const WT* g = Wtp + w_off[i] + wk;
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
    f32x4 accx[(X3 || X2W) ? TNE : 1][(X3 || X2W) ? TME : 1];   // x3: the cross terms hi*lo + lo*hi (scaled by 2^11); x2w: lo_w * x
    if constexpr (X3 || X2W) {
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
    // (round 5: not for the 128 x 128 tiles either -- 64 KB of f32 residual per workgroup queued ahead of the first LDS-DMA group held the first k-tile
    //  back by 6 us on the residual convolutions of the decoder, tools/conv_stamps.py: entry -> first tile 8.2 us against 2.0 us)
    constexpr bool PRE = !SK && (TNE * TME <= 8) && (BM * BN < 128 * 128);
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
    const int x_row0 = (wm * C::TM * 16 + frow) * C::ROWB, w_row0 = C::X_BYTES + (wn * C::TN * 16 + frow) * W_ROWB;
    // x2w: weight fragments of k-step ks = unit 4 ks + fq of the 2 * ROWB-byte x3 row (hi / lo chunk as in the x3 tiles); the activation fragments
    // are the fp16 ones (xr_off)
    int w2h_off[X2W ? C::KS : 1], w2l_off[X2W ? C::KS : 1];
    if constexpr (X2W) {
        const int wswz = swz_of_row<C::BK * 2, 16>(frow);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int u = ks * 4 + fq;
            w2h_off[ks] = ((2 * u + (u & 1)) ^ wswz) * 16;
            w2l_off[ks] = ((2 * u + 1 - (u & 1)) ^ wswz) * 16;
        }
    }
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
        if (kt + C::NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C::NS - 2) * LOADS) : "memory");
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
        const char* sb = smem + (kt % C::NS) * STAGE;
        if constexpr (X2W) {
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                h16x8 wh[C::TN], wl[C::TN], xf[C::TM];
#pragma unroll
                for (int i = 0; i < C::TN; ++i) {
                    wh[i] = *reinterpret_cast<const h16x8*>(sb + w_row0 + w2h_off[ks] + i * 16 * W_ROWB);
                    wl[i] = *reinterpret_cast<const h16x8*>(sb + w_row0 + w2l_off[ks] + i * 16 * W_ROWB);
                }
#pragma unroll
                for (int j = 0; j < C::TM; ++j) xf[j] = *reinterpret_cast<const h16x8*>(sb + xr_off[ks] + j * 16 * C::ROWB);
#pragma unroll
                for (int i = 0; i < C::TN; ++i)
#pragma unroll
                    for (int j = 0; j < C::TM; ++j) {
                        acc[i][j] = mfma_16x16x32<true>(wh[i], xf[j], acc[i][j]);
                        accx[i][j] = mfma_16x16x32<true>(wl[i], xf[j], accx[i][j]);
                    }
            }
        } else if constexpr (X3) {
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
    if constexpr (X3 || X2W) {   // fold the cross terms in: a b = hi hi + 2^-11 (hi lo + lo hi)   (x2w: w x = hi x + 2^-11 lo x)
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
    // (Round 5, built and removed: the 16-bit operand copy staged through LDS and written 16 bytes per lane, a pixel's BN channels contiguous, instead of
    //  16 pixels x 32 bytes per wave-instruction straight from the accumulators.  The stamps showed a 4.5 us store tail per 128 x 128 workgroup and skipping
    //  the stores altogether is worth 133 us of a 2005 us forward, but the coalesced path measured 3982-3987 against 3991-4001 frames/s for the direct one,
    //  alternated in one GPU call (profiles/r05_ab_coalesced_epilogue.txt): what the stores cost is their bytes, not their shape.)
    float dot_part[TME];
#pragma unroll
    for (int j = 0; j < TME; ++j) dot_part[j] = 0.f;
    float d3[D3 ? TME : 1][3];   // D3: this lane's part of the three class logits of its pixels (its 4 x TNE channels)
    if constexpr (D3) {
#pragma unroll
        for (int j = 0; j < TME; ++j) d3[j][0] = d3[j][1] = d3[j][2] = 0.f;
    }
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
#ifdef SOCCDPT_ABLATIONS
            const bool store_op = d.out_op && d.dbg_skip_out_op != 1;
#else
            const bool store_op = d.out_op != nullptr;
#endif
            if (store_op) {
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
            if constexpr (D3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 w4 = *reinterpret_cast<const float4*>(d.dot_w + (size_t)c * N + n);
                    d3[j][c] += a[0] * w4.x + a[1] * w4.y + a[2] * w4.z + a[3] * w4.w;
                }
            } else if (d.out_dot) {
                const float4 w4 = *reinterpret_cast<const float4*>(d.dot_w + n);
                dot_part[j] += a[0] * w4.x + a[1] * w4.y + a[2] * w4.z + a[3] * w4.w;
            }
        }
    }
    if constexpr (ST) {
        // ---- GroupNorm statistics.  (1) in-wave: the 16 lanes that share (lane >> 4) hold the same 4 channels of 16 different pixels.
        // (2) per-channel sums of the WM wave rows meet in LDS (the staging ring is free after the barrier), (3) one thread per group adds
        // its gn_cpg channels in a fixed order and stores the tile's partial.  The READER of out_f32 adds the tile partials of a sample in
        // tile order in f64 (gn_apply / gn_finish, hybrid.hip): fixed orders everywhere, bitwise reproducible, nobody waits.
        // (Rounds 2-4 finished here: the last workgroup of a sample to arrive -- one counter per sample -- walked the partials.  Per-workgroup
        //  stamps priced that at 2-7 us on top of a 1.8 us epilogue for EVERY workgroup: the partial could only be published after vmcnt(0), i.e.
        //  after the whole output tile had landed, then the counter round trip, then the walk -- three dependent memory round trips on the
        //  critical path of launches that are one round of workgroups long.  tools/rn_stamps.py, profiles/r05_rn_stamps_before.txt.)
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
            *reinterpret_cast<float2*>(pp) = make_float2(a, q);   // plain store, nothing to wait for: the reader of out_f32 adds the partials (hybrid.hip)
        }
    }
    if (GEN && d.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) d.stamps[4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (D3) {
        // lanes (lane & 15) + 16 q hold 4-channel groups q of the same pixel: two shuffles; the WN waves of a wave row hold the tile's other
        // channels: they meet in LDS (the staging ring is free once every wave has left the main loop), summed in wave order
        float* red = reinterpret_cast<float*>(smem);   // [BM][WN][4]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TME; ++j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v3 = d3[j][c];
                v3 += __shfl_xor(v3, 16);
                v3 += __shfl_xor(v3, 32);
                d3[j][c] = v3;
            }
            if ((lane >> 4) == 0) {
                float* q = red + ((size_t)(wm * TME * 16 + j * 16 + (lane & 15)) * C::WN + wn) * 4;
                q[0] = d3[j][0]; q[1] = d3[j][1]; q[2] = d3[j][2];
            }
        }
        __syncthreads();
        for (int r = tid; r < BM; r += C::THREADS) {
            const int m = m0 + r;
            if (m >= d.M) continue;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < C::WN; ++w) { const float* q = red + ((size_t)r * C::WN + w) * 4; s0 += q[0]; s1 += q[1]; s2 += q[2]; }
            *reinterpret_cast<float4*>(d.out_dot + ((size_t)nt * d.M + m) * 4) = make_float4(s0, s1, s2, 0.f);
        }
    } else if (MF == 16 && d.out_dot) {  // host guarantees WN == 1 and N <= BN: the whole channel range is in this wave
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

template <class C, typename T, bool LN, bool SK = false, bool ST = false, bool GEN = false, bool D3 = false>
__global__ __launch_bounds__(C::THREADS) void igemm_kernel(IgemmDesc d, int nk, int kpt, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware bijective remap: consecutive logical tiles -> same XCD (blocks b, b+8 share one)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    igemm_tile<C, T, LN, SK, ST, GEN, D3>(d, nk, kpt, ntiles, bid, smem, (int)threadIdx.x);
}

}  // namespace soccdpt
