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
#include "igemm_kernel.h"

namespace soccdpt {

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

template <class C, typename T, bool LN = false, bool SK = false, bool ST = false, bool GEN = false, bool D3 = false>
static int launch_cfg_t(const IgemmDesc& d, hipStream_t stream, std::string& err) {
    if (D3 != (d.dot3 != 0)) { err = "igemm: this configuration has no three-class classifier epilogue"; return 1; }
    if (!GEN && need_gen(d)) { err = "igemm: this configuration has no generalised-addressing instantiation (use 2, 20 or a GroupNorm-statistics launch)"; return 1; }
    if (LN != (d.ln_g != nullptr)) { err = "igemm: this configuration has no fused-LayerNorm instantiation"; return 1; }
    if (SK != (d.splitk > 1)) { err = "igemm: this configuration has no split-K instantiation"; return 1; }
    if (ST != (d.gn_stats != nullptr)) { err = "igemm: this configuration has no GroupNorm-statistics instantiation"; return 1; }
    constexpr int BK = C::ROWB / (int)sizeof(T);
    const int nk = d.taps * d.Cin / BK, kpt = d.Cin / BK;
    const int mtiles = (d.M + C::BM - 1) / C::BM, ntiles = (d.N + C::BN - 1) / C::BN;
    const size_t lds = (size_t)C::NS * igemm_stage_bytes<C, T>();
    if constexpr (ST) {
        const int G = d.gn_cpg > 0 ? d.N / d.gn_cpg : 0;
        if (!d.gn_part || !d.out_f32 || d.gn_cpg <= 0 || d.N % d.gn_cpg || C::BN % d.gn_cpg || d.gn_hw <= 0 || d.gn_hw % C::BM || d.M % d.gn_hw ||
            (size_t)mtiles * G * 2 > d.gn_part_floats || G > C::THREADS ||
            (size_t)(2 * C::WM * C::BN + 4) * 4 + (size_t)C::THREADS * 16 > lds) {
            err = "igemm: bad GroupNorm-statistics descriptor (pixels per sample must be a multiple of the M tile)";
            return 1;
        }
        if (d.gn_bm_out) *d.gn_bm_out = C::BM;
    }
    static PerDeviceOnce attr_done;
    if (attr_done.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<C, T, LN, SK, ST, GEN, D3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { err = std::string("igemm: hipFuncSetAttribute: ") + hipGetErrorString(e); return 1; }
        attr_done.done();
    }
    const int splits = SK ? d.splitk : 1;
    if (SK && (!d.sk_part || !d.sk_count || splits > nk || (size_t)splits * d.M * d.N > d.sk_part_floats ||
               (size_t)mtiles * ntiles > d.sk_count_words)) { err = "igemm: bad split-K descriptor (scratch too small?)"; return 1; }
    if (SK && d.sk_defer && (!d.out_f32 || d.out_op || d.bias || d.res1 || d.res2 || d.act || (d.N & 3))) { err = "igemm: deferred split-K writes out_f32 only"; return 1; }
    SOCCDPT_LAUNCH((igemm_kernel<C, T, LN, SK, ST, GEN, D3>), dim3((unsigned)(mtiles * ntiles * splits)), dim3(C::THREADS), lds, stream, d, nk, kpt, ntiles);
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
    // plain 1x1 launches skip the generalised addressing (round 5: its address set-up is a fifth of a K = 64 .. 256 launch, tools/rn_stamps.py)
    if (!need_gen(d)) return d.f16 ? launch_cfg_t<C, f16_t, false, false, true, false>(d, stream, err) : launch_cfg_t<C, bf16_t, false, false, true, false>(d, stream, err);
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
                                        "igemm_bf16_128x128x64_s2_w8_splitk",
                                        "igemm_bf16_256x256x64_s2_w16", "igemm_bf16_256x256x32_s4_w16", "igemm_bf16_512x128x64_s2_w16",
                                        "igemm_bf16_256x128x64_s2_w16", "igemm_bf16_128x256x64_s2_w16", "igemm_bf16_128x256x64_s2_w16b"};

static const char* const kCfgNamesF32[] = {"igemm_f32_128x128x32_s2", "igemm_f32_64x64x32_s4", "igemm_f32_128x32x32_s4", "igemm_f32_128x128x32_s2_w8"};
static const char* const kCfgNamesX3[] = {"igemm_x3_128x128x32_s2", "igemm_x3_64x64x32_s4", "igemm_x3_128x32x32_s4", "igemm_x3_128x128x32_s2_w8",
                                          "igemm_x3_64x64x32_s4_w8", "igemm_x3_128x128x32_s3_w8", "igemm_x3_128x128x32_s4_w8", "igemm_x3_128x64x32_s3",
                                          "igemm_x3_128x128x64_s2_w8", "igemm_x3_64x64x64_s3", "igemm_x3_32x64x64_s3_w8", "igemm_x3_64x128x32_s3", "igemm_x3_64x64x32_s2_w8"};
constexpr int kNumCfgX3 = 13;
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
        if (d.gn_stats && (t == 0 || t == 1 || t == 3 || t == 4 || t == 7 || t == 12) && d.gn_hw % bm == 0 && bn % d.gn_cpg == 0 && !(gen && t != 1 && t != 4)) return t;
        if (!d.gn_stats && gen && (t == 1 || t == 4)) return t;
    }
    if (d.x3 && d.gn_stats && d.splitk <= 1 && !need_gen(d)) {
        // in-network timings of the ResNetV2 convolutions of dpt_hybrid_384 (profiles/r03_autotune_x3_hyb_st.txt): the 8-wave 64 x 64 tile
        // wins nearly everywhere (the statistics epilogue is per-wave work, and these launches are short of workgroups)
        const long b64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
        // short K on a grid of more than one round: the two-stage ring (two workgroups per CU) -- round 5, tools/rn_stamps.py: 36864 x 256 x 64 24.9 -> 21.4 us,
        // 36864 x 64 x 256 15.2 -> 13.1, 9216 x 512 x 128 17.6 -> 15.9, 2304 x 1024 x 256 15.1 -> 13.6; hybrid_384 forward 892 -> 902 frames/s (alternated in one call)
        if (d.taps == 1 && d.Cin <= 256 && b64 > 256 && d.gn_hw % 64 == 0 && 64 % d.gn_cpg == 0 && d.tune < 0) return 12;
        if (d.gn_hw % 128 == 0 && 128 % d.gn_cpg == 0 && d.N >= 256 && d.M >= 32768) return 3;
        if (d.gn_hw % 64 == 0 && 64 % d.gn_cpg == 0 && !(d.taps == 9 && b64 >= 256 && b64 < 512)) return 4;
    }
    if (d.x3 && d.gn_stats && d.splitk <= 1 && d.gn_hw % 64 == 0 && 64 % d.gn_cpg == 0) {
        // ... with generalised addressing (the 3x3 'SAME' convolutions of the ResNetV2 bottlenecks): 64 x 64 tiles only.  tools/rn_stamps.py (round 5): a
        // 64-channel convolution on the 128 x 128 tile wastes half of it (36864 x 64 x 576: 50.6 us against 24.5 on the 8-wave 64 x 64 tile); short grids
        // take the 8-wave form (2304 x 256 x 2304: 28.8 against 33.0), the mid grid (9216 x 128 x 1152) the 4-wave one (22.7 against 25.2)
        const long b64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
        return (d.N < 128 || b64 < 256) ? 4 : 1;
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
        // (round 5: the two-stage 64 x 64 tile, configuration 12, wins 10-25 % on the wide short-K launches when every site is timed by itself
        //  (tools/autotune_network.py ... f16x3, profiles/r05_autotune_x3_*_c12.txt) and LOSES 4 % over the forward (base_384 x3: 693 -> 667 frames/s, alternated
        //  in one GPU call): it stays with the GroupNorm-statistics launches of the hybrid, where the forward confirmed it (892 -> 902).  Likewise 64 x 64 tiles
        //  instead of 1.1 rounds of 128 x 128 (b128 = 288): +0.7 % on base_384 x3, -0.8 % on hybrid_384 x3: not taken.)
        // ... but between fp16 launches (SOCCDPT_PREC_MIXED: d.x3_among_f16) it does pay over the forward: tiny_256 3979 -> 3995 frames/s, base_384 1265 -> 1275
        // (three alternations each in one GPU call; SOCCDPT_X3_C12=0 switches it off)
        static const int c12 = getenv("SOCCDPT_X3_C12") ? atoi(getenv("SOCCDPT_X3_C12")) : 1;
        if (c12 && d.x3_among_f16 && d.taps == 1 && d.N >= 384 && K <= 512 && b64 >= 512) return 12;
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

// x2w launches (fp16 activations, x3 weight pairs, two MFMAs per product): the tile ids of the x2w family
//   0 64x64x64_s3_w8   1 32x64x64_s4 (small grids, long K)   2 128x128x64_s2_w8 (big grids)   3 64x128x32_s3 (fused LayerNorm, N <= 128)   4 64x64x32_s4 (C = 96)
static const char* const kCfgNamesX2W[] = {"igemm_x2w_64x64x64_s3_w8", "igemm_x2w_32x64x64_s4", "igemm_x2w_128x128x64_s2_w8", "igemm_x2w_64x128x32_s3_ln", "igemm_x2w_64x64x32_s4"};
static int pick_cfg_x2w(const IgemmDesc& d) {
    if (d.ln_g) return 3;
    const bool k64 = d.Cin % 64 == 0;
    if (!k64) return 4;
    if (d.tune >= 0 && d.tune <= 2) return d.tune;
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long b64 = cdiv(d.M, 64) * cdiv(d.N, 64), b128 = cdiv(d.M, 128) * cdiv(d.N, 128);
    if (d.gn_stats) return (d.gn_hw % 128 == 0 && 128 % d.gn_cpg == 0 && b128 >= 256) ? 2 : 0;
    if (need_gen(d)) return 0;
    if (b128 >= 384) return 2;          // the rule of the x3 family (7 / 8): big grids take the 128 x 128 tile
    if (b64 <= 256) return 1;           // the rule of x3 configuration 10
    return 0;
}

int igemm_pick_splitk(const IgemmDesc& d, size_t part_floats, size_t count_words) {
    if (d.x2w) return 1;
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

// channel tile of the fused three-class classifier launch (dot3): 256 when the 16-wave 256 x 256 tile runs it -- M a multiple of 256 whose
// 256-row tiles fill the 256 CUs in whole rounds or at least four of them (fill_probe / conv_tiles A/B of round 5: 928 vs 765 TFLOP/s on the
// seg-head convolution of tiny_256; 4.5 rounds on base_384's), N == 256 -- else 128.  d.tune == 21 / 47 force one or the other.
int igemm_dot3_bn(const IgemmDesc& d) {
    if (d.tune == 21) return 128;
    if (d.N != 256 || d.M % 256 != 0) return 128;
    if (d.tune == 47) return 256;
    static const int mode = getenv("SOCCDPT_CONV16W") ? atoi(getenv("SOCCDPT_CONV16W")) : 1;
    if (!mode) return 128;
    const long tiles = d.M / 256;
    return (tiles % 256 == 0 || tiles >= 1024) ? 256 : 128;
}

int igemm_config_id(const IgemmDesc& d) { return d.x3 ? pick_cfg_f32(d) : (d.f32 ? -1 : pick_cfg(d)); }

const char* igemm_family(const IgemmDesc& d) {
    if (d.x2w) return kCfgNamesX2W[pick_cfg_x2w(d)];
    if (d.dot3 && !d.x3 && !d.f32) {
        if (igemm_dot3_bn(d) == 256) return d.f16 ? "igemm_f16_256x256x64_s2_w16_dot3" : "igemm_bf16_256x256x64_s2_w16_dot3";
        return d.f16 ? "igemm_f16_128x128x64_s2_w8_dot3" : "igemm_bf16_128x128x64_s2_w8_dot3";
    }
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
    if (d.out_dot && !d.dot3 && d.N > 32) { err = "igemm: fused dot tail needs N <= 32"; return 1; }
    if (d.dot3 && (d.x3 || d.f32)) { err = "igemm: the fused three-class classifier exists for 16-bit operands only"; return 1; }
    if (d.ln_g && (!d.ln_b || !d.ln_xf || d.N > 128 || (d.ln_halo && (d.H <= 0 || d.W <= 0)))) { err = "igemm: bad fused-LayerNorm descriptor"; return 1; }
    if (d.N <= 32 && d.Cin % 64 != 0 && !d.f32 && !d.x3) { err = "igemm: N <= 32 needs Cin % 64 == 0"; return 1; }
    if (d.x2w) {   // one-sided split (round 5): fp16 activations, x3 weight pairs
        if (d.f32 || d.x3 || d.dot3 || d.out_dot || d.splitk > 1 || d.wt_grp_rows) { err = "igemm: x2w launches have no fused dot tail, split-K or weight row groups"; return 1; }
        if (d.ln_g && d.Cin % 32) { err = "igemm: x2w needs Cin % 32 == 0"; return 1; }
        const bool gen = need_gen(d);
        switch (pick_cfg_x2w(d)) {
            case 1: if (gen || d.gn_stats) break;
                    return launch_cfg_t<Cfg<32, 64, 64, 2, 2, 4>, x2w_t>(d, stream, err);
            case 2: return d.gn_stats ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x2w_t, false, false, true, true>(d, stream, err)
                         : gen ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x2w_t, false, false, false, true>(d, stream, err)
                               : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x2w_t>(d, stream, err);
            case 3: if (gen || d.gn_stats) break;
                    return launch_cfg_t<Cfg<64, 128, 32, 2, 2, 3>, x2w_t, true>(d, stream, err);
            case 4: if (gen || d.gn_stats) break;
                    return launch_cfg_t<Cfg<64, 64, 32, 2, 2, 4>, x2w_t>(d, stream, err);
            default: break;
        }
        return d.gn_stats ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 3>, x2w_t, false, false, true, true>(d, stream, err)
                   : gen ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 3>, x2w_t, false, false, false, true>(d, stream, err)
                         : launch_cfg_t<Cfg<64, 64, 64, 2, 4, 3>, x2w_t>(d, stream, err);
    }
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
            case 0: return d.gn_stats ? (need_gen(d) ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t, false, false, true, true>(d, stream, err) : launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t, false, false, true, false>(d, stream, err))
                         : d.ln_g ? launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t, true>(d, stream, err)
                                  : launch_cfg_t<Cfg<128, 128, 64, 2, 2, 2>, x3_t>(d, stream, err);
            case 1: return d.gn_stats ? (need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, false, true, true>(d, stream, err) : launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, false, true, false>(d, stream, err))
                         : need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t, false, false, false, true>(d, stream, err)
                                       : launch_cfg_t<Cfg<64, 64, 64, 2, 2, 4>, x3_t>(d, stream, err);
            case 3: return d.gn_stats ? (need_gen(d) ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t, false, false, true, true>(d, stream, err) : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t, false, false, true, false>(d, stream, err))
                                      : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, x3_t>(d, stream, err);
            case 4: return d.gn_stats ? (need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t, false, false, true, true>(d, stream, err) : launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t, false, false, true, false>(d, stream, err))      // 8 waves, 32 x 16 per wave
                         : need_gen(d) ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t, false, false, false, true>(d, stream, err)
                                       : launch_cfg_t<Cfg<64, 64, 64, 2, 4, 4>, x3_t>(d, stream, err);
            case 5: return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 3>, x3_t>(d, stream, err);    // 3-stage ring (96 KB)
            case 6: return launch_cfg_t<Cfg<128, 128, 64, 2, 4, 4>, x3_t>(d, stream, err);    // 4-stage ring (128 KB)
            case 7: return d.gn_stats ? (need_gen(d) ? launch_cfg_t<Cfg<128, 64, 64, 2, 2, 3>, x3_t, false, false, true, true>(d, stream, err) : launch_cfg_t<Cfg<128, 64, 64, 2, 2, 3>, x3_t, false, false, true, false>(d, stream, err))
                                      : launch_cfg_t<Cfg<128, 64, 64, 2, 2, 3>, x3_t>(d, stream, err);     // 4 waves, 64 x 32 per wave, two workgroups per CU
            case 8: return launch_cfg_t<Cfg<128, 128, 128, 2, 4, 2>, x3_t>(d, stream, err);   // 64-deep k-tiles (256-byte rows): half the barriers
            case 9: return launch_cfg_t<Cfg<64, 64, 128, 2, 2, 3>, x3_t>(d, stream, err);
            case 10: return launch_cfg_t<Cfg<32, 64, 128, 2, 4, 3>, x3_t>(d, stream, err);    // small grids, long K
            case 11: return launch_cfg_t<Cfg<64, 128, 64, 2, 2, 3>, x3_t>(d, stream, err);
            case 12:   // 8 waves, TWO stages (64 KB: two workgroups per CU): the short-K 1x1 convolutions (K <= 256 is 2 .. 8 k-tiles; the 4-stage ring only held the CU)
                if (need_gen(d)) { err = "igemm: x3 configuration 12 has no generalised addressing"; return 1; }
                return d.gn_stats ? launch_cfg_t<Cfg<64, 64, 64, 2, 4, 2>, x3_t, false, false, true, false>(d, stream, err)
                                  : launch_cfg_t<Cfg<64, 64, 64, 2, 4, 2>, x3_t>(d, stream, err);
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
    if (d.dot3) {   // Conv3x3 + (folded BN) + ReLU + Conv1x1(N -> 3) of the seg head as one launch: the 8-wave 128 x 128 x 64 tile with the D3 epilogue
        if (d.taps != 9 || d.Cin % 64 || d.N % 128 || !d.dot_w || !d.out_dot || d.out_op || d.out_f32 || d.splitk > 1 || d.ln_g || d.gn_stats || need_gen(d)) {
            err = "igemm: bad three-class classifier descriptor (3x3 convolution, N a multiple of 128, no other outputs)";
            return 1;
        }
        if (igemm_dot3_bn(d) == 256)   // the 16-wave 256 x 256 tile: one n-tile holds all 256 channels, the partial logits need no finishing sum
            return d.f16 ? launch_cfg_t<Cfg<256, 256, 64, 4, 4, 2>, f16_t, false, false, false, false, true>(d, stream, err)
                         : launch_cfg_t<Cfg<256, 256, 64, 4, 4, 2>, bf16_t, false, false, false, false, true>(d, stream, err);
        return d.f16 ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, f16_t, false, false, false, false, true>(d, stream, err)
                     : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, bf16_t, false, false, false, false, true>(d, stream, err);
    }
    const int id = pick_cfg(d);
    const bool k64 = (d.Cin % 64 == 0);
    if (id >= 30 && id <= 39) return launch_conv8p(d, id - 30, stream, err);
    if ((id == 0 || id == 1 || id == 2 || id == 5 || id == 6 || id == 7 || id == 8 || (id >= 10 && id <= 14) || (id >= 21 && id <= 23) || (id >= 40 && id <= 47) || (id >= 49 && id <= 52)) && !k64) { err = "igemm: this configuration needs Cin % 64 == 0"; return 1; }
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
        case 21:   // 8 waves, 64x32 per wave: twice the resident waves of configuration 1
            if (d.stamps && !d.gn_stats)   // diagnostics only (tools/conv_stamps.py): the generalised-addressing instantiation carries the four per-workgroup stamps
                return d.f16 ? launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, f16_t, false, false, false, true>(d, stream, err)
                             : launch_cfg_t<Cfg<128, 128, 64, 2, 4, 2>, bf16_t, false, false, false, true>(d, stream, err);
            return launch_cfg_st<Cfg<128, 128, 64, 2, 4, 2>>(d, stream, err);
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
        // 16-wave tiles (round 5; 1024 threads = 4 waves per SIMD from ONE workgroup: the resident waves of two 8-wave workgroups with the operand
        // bytes per FLOP of a 256 x 256 tile; tools/fill_probe.hip: the L2 -> LDS fill of the 128 x 128 tile is what holds its MFMAs at 0.3-0.4)
        case 47: return launch_cfg<Cfg<256, 256, 64, 4, 4, 2>>(d, stream, err);   // 64 x 64 per wave, 128 KB LDS
        case 48: return launch_cfg<Cfg<256, 256, 32, 4, 4, 4>>(d, stream, err);   // 32-deep k-tiles, three tiles in flight, 128 KB LDS
        case 49: return launch_cfg<Cfg<512, 128, 64, 8, 2, 2>>(d, stream, err);   // N = 128 (depth head): 64 x 64 per wave, all 160 KB of LDS
        case 50: return launch_cfg<Cfg<256, 128, 64, 4, 4, 2>>(d, stream, err);   // N = 128: 64 x 32 per wave, 96 KB
        case 51: return launch_cfg<Cfg<128, 256, 64, 2, 8, 2>>(d, stream, err);   // M = 32768 (64^2 RCU convolutions: 256 tiles): 64 x 32 per wave, 96 KB
        case 52: return launch_cfg<Cfg<128, 256, 64, 4, 4, 2>>(d, stream, err);   // same tile, 32 x 64 per wave
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
