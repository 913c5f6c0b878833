// Descriptor of one implicit-GEMM launch (igemm.hip): out[m][n] = sum_k X[m][k] * Wt[n][k].
// X rows are either plain rows of a [M][K] bf16 matrix (Linear layers, 1x1 convs) or the
// 3x3 neighbourhoods of a zero-haloed NHWC bf16 image (the decoder's 3x3 convolutions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>

#include "launch.h"

namespace soccdpt {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one flag per device and call site (a
// process may hold one handle per GPU).  Setting the attribute twice is harmless, so concurrent first calls need no lock.
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask[4];   // 256 devices
    PerDeviceOnce() { for (auto& m : mask) m.store(0ull); }
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 255; }
    bool need() const { const int d = dev(); return !((mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull); }
    void done() { const int d = dev(); mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release); }
};

typedef uint16_t bf16_t;  // raw 16-bit operand bits: bf16, or fp16 under SOCCDPT_PREC_F16 (half16.h)

// Zero-haloed NHWC activation: pixel (b,y,x) lives at ((b*(H+2)+y+1)*(W+2)+x+1)*C.
// The one-pixel border is never written by any kernel and is zero from workspace init.
struct Halo {
    int H = 0, W = 0, C = 0;
    __host__ __device__ size_t elems(int B) const { return (size_t)B * (H + 2) * (W + 2) * C; }
};

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct IgemmDesc {
    // ---- operands ----
    const void* X = nullptr;   // activations, bf16 (or f32 when f32 != 0)
    const void* Wt = nullptr;  // weights [N][taps*Cin], K contiguous, same element type
    int f32 = 0;               // exact-f32 operands and f32 MFMA (SOCCDPT_PREC_F32)
    int f16 = 0;               // 16-bit operands are IEEE fp16 instead of bf16 (SOCCDPT_PREC_F16); ignored when f32 != 0
    int x3 = 0;                // split-fp16 operands in the x3 layout (half16.h), three fp16 MFMAs per product (SOCCDPT_PREC_F16X3); wins over f32 / f16
    int x2w = 0;               // one-sided split (round 5): X is plain fp16, Wt holds x3 pairs (4 bytes per element), two fp16 MFMAs per product; wins over f16
    int M = 0, N = 0;
    int Cin = 0;      // channels per tap (K of a plain GEMM)
    int taps = 1;     // 1 (GEMM / 1x1) or 9 (3x3, pad 1)
    // plain mode: row m at X + m*ldx.  conv mode (taps == 9): X is a Halo image [B][H+2][W+2][Cin], M = B*H*W
    int ldx = 0;
    int H = 0, W = 0;
    // Generalised conv addressing (taps == 9, or taps == 1 with gather1 != 0: a strided 1x1 convolution).  Output pixel (y, x) of the
    // H x W map reads input pixel (y*stride + ky - pad, x*stride + kx - pad) of an NHWC image [B][Hi + 2*in_halo][Wi + 2*in_halo][Cin]
    // (in_halo = 1: zero-haloed; every tap must stay inside the haloed image -- checked by launch_igemm).  Defaults = the stride-1
    // pad-1 convolution of a halo image.  stride 2 / pad 0 on a halo image is timm's dynamic 'SAME' 3x3 (the extra pixel right / bottom),
    // stride 2 / pad 1 is nn.Conv2d(k=3, s=2, p=1) (/root/reference/SOccDPT/model/backbones/vit.py:196-202).
    int stride = 1, pad = 1, in_halo = 1;
    int Hi = 0, Wi = 0;   // input map size; 0 = H, W
    int gather1 = 0;
    // Plain-mode row groups: row m lives at X + (m / grp_rows) * grp_stride + grp_off + (m % grp_rows) * ldx  (grp_rows == 0: m * ldx).
    // The ViT token matrix [B][577][768] read without its class-token rows is grp_rows = 576, grp_stride = 577*768, grp_off = 768.
    int grp_rows = 0, grp_off = 0;
    long long grp_stride = 0;
    // Second A segment: columns k >= seg2_k come from the PER-GROUP row X + (m / grp_rows) * grp_stride + seg2_off + (k - seg2_k):
    // cat(token, class token) @ W^T of ProjectReadout (/root/reference/SOccDPT/model/backbones/utils.py:27-40) without the cat.
    int seg2_k = 0, seg2_off = 0;
    // Weight row groups (generalised addressing; the 3x3 weight-gradient GEMM of the training step, train_step.cpp: conv3_bwd): row n of Wt
    // lives at Wt + (n % wt_grp_rows) * K + shift(n / wt_grp_rows), shift(g) = wt_base + (g / 3 - 1) * wt_rp + (g % 3 - 1) + (g % 3 != 1 ? wt_odd : 0)
    // -- nine shifted views (one per tap) of ONE transposed halo image instead of an im2col^T.  wt_grp_rows must be a multiple of 64 (a weight
    // tile never straddles two groups); every shift must be >= 0 (the caller offsets Wt).
    int wt_grp_rows = 0, wt_rp = 0, wt_base = 0, wt_odd = 0;
    // wt_kx != 0 (x3 operands): shift(g) = wt_base + (g / 3 - 1) * wt_rp + (g % 3) * wt_kx -- the horizontal tap selects one of three pre-shifted COPIES of the
    // transposed image (copy stride wt_kx) instead of an element offset, so every view starts at a multiple of 16 elements (wt_base, wt_rp, wt_kx % 16 == 0)
    int wt_kx = 0;
    // ---- epilogue: v = acc (+bias[n]) (+res1[m][n]) (+res2[m][n]); act; stores ----
    const float* bias = nullptr;
    const float* res1 = nullptr;  // f32 [M][N]
    const float* res2 = nullptr;  // f32 [M][N]
    // res2 sampled bilinearly (align_corners=True) from a LOW-RES f32 NHWC map [B][res2_h][res2_w][N] instead of read 1:1:
    // fuses F.interpolate of the previous fusion block's output (model/blocks.py:488-493) into this epilogue
    int res2_h = 0, res2_w = 0;
    int act = ACT_NONE;           // applied to every store except out_f32_raw
    float* out_f32 = nullptr;     // [M][N], value BEFORE `act` when act_on_f32 == 0
    int act_on_f32 = 0;
    void* out_op = nullptr;       // operand-typed copy (bf16 / f32): [M][N] plain (ld = N) or Halo image when out_halo != 0
    int out_halo = 0;
    int out_op_f32 = 0;           // x3 launches only: out_op is a plain f32 tensor / halo image (the training tape keeps f32 activations), not an x3 operand
    // Mixed-precision launch sequences (SOCCDPT_PREC_MIXED, model.cpp): the operand copy is written for the NEXT launch's format, which may differ
    // from this launch's own.  -1 = same as this launch's operands; 1 = IEEE fp16 (from an x3 launch); 3 = x3 (from an fp16 launch).  ln_halo follows
    // halo_fmt the same way (the hooked feature map feeds the decoder, the operand copy the next encoder block).
    int out_fmt = -1, halo_fmt = -1;
    // fused 1x1 tail of the depth head: out_dot[m] = relu(sum_n act(v)[n]*dot_w[n] + dot_b)   (N <= 32)
    const float* dot_w = nullptr;
    float dot_b = 0.f;
    float* out_dot = nullptr;
    // fused THREE-class 1x1 tail (the seg head's classifier): dot3 != 0: out_dot[(n_tile * M + m) * 4 + c] = sum over the n-tile's channels of
    // act(v)[n] * dot_w[c * N + n] (c < 3; n-tiles of 128 channels); nothing else is stored.  16-bit operand launches only (igemm.hip)
    int dot3 = 0;
    // fused Swin-V2 residual post-norm (N <= tile width, one n-tile): xf[m][:] = (residual ? xf[m][:] : 0) + LN(v[m][:]) * g + b;
    // also writes the operand-typed copy to out_op (plain) and, when ln_halo != nullptr, to a zero-halo image (hooked stage)
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    float* ln_xf = nullptr;
    void* ln_halo = nullptr;
    int ln_residual = 1;
    // split-K (igemm.hip, SK): splitk > 1 slices the k-tiles over splitk workgroups per output tile; sk_part holds splitk x M x N
    // floats (any contents), sk_count one zero-initialised word per 32 x 64 output tile (left zero again by the kernel)
    int splitk = 1;
    float* sk_part = nullptr;
    unsigned* sk_count = nullptr;
    size_t sk_part_floats = 0, sk_count_words = 0;  // capacities, validated by launch_igemm
    // sk_defer: every split only stores its partial tile (plain 16-byte stores) and a second launch sums the splits in order into out_f32 (nothing
    // else of the epilogue applies).  For MANY splits of BIG tiles: the last-arriver reduction above is one workgroup walking splitk x tile floats
    // with L2-bypassing loads (14 splits of a 128 x 128 tile: ~400 us), the deferred one is a chip-wide elementwise pass (~10 us).
    int sk_defer = 0;
    int tune = -1;  // kernel configuration id (igemm.hip); -1 = heuristic
    int x3_among_f16 = 0;   // hint of the caller: this x3 launch sits between fp16 launches (SOCCDPT_PREC_MIXED) -- the tile heuristic differs (igemm.hip: x3 configuration 12)
    // GroupNorm statistics of the raw output (the ST instantiation): with gn_stats != nullptr the epilogue also reduces sum / sum of squares of v over every
    // (M tile, group of gn_cpg consecutive channels): gn_part[((mt * G) + g) * 2] = {sum, sum of squares} of M tile mt (tile rows: *gn_bm_out), group g,
    // G = N / gn_cpg -- plain stores, no counter.  The reader of out_f32 adds a sample's tiles in tile order in f64 and forms {mean, 1 / sqrt(var + gn_eps)}
    // (biased variance, torch.nn.GroupNorm): launch_gn_apply with GnApplyArgs::part / tps, or launch_gn_finish (hybrid.hip), which also write them to the
    // [B][G][2] array gn_stats names.  gn_stats itself is only the switch here.  gn_hw = pixels per sample (M = B * gn_hw; a multiple of the M tile).
    // Launches with statistics carry no bias / residual (v is the accumulator).
    int* gn_bm_out = nullptr;   // host pointer: launch_igemm stores the M-tile rows of the configuration it launched (tiles per sample = gn_hw / rows)
    float* gn_stats = nullptr;
    float* gn_part = nullptr;
    int gn_cpg = 0, gn_hw = 0;
    float gn_eps = 1e-5f;
    size_t gn_part_floats = 0;   // capacity, validated by launch_igemm
    // diagnostics (tools/igemm_stamps.py): when non-null every workgroup writes 4 s_memrealtime stamps (100 MHz) -- entry, first k-tile
    // landed, main loop done, epilogue done -- to stamps[4 * blockIdx.x ..]; the values are never read by the kernel
    unsigned long long* stamps = nullptr;
#ifdef SOCCDPT_ABLATIONS
    int dbg_skip_out_op = 0;   // SOCCDPT_DBG_SKIP_OUT_OP=1: timing-only ablation (WRONG results): the generic epilogue does not store the operand copy (133 us of 2005 per forward)
#endif
};

int launch_igemm(const IgemmDesc& d, hipStream_t stream, std::string& err);
// conv8p.hip: phase-interleaved big-tile 3x3 convolution (configuration ids 30 / 31 / 32 = 256x256 / 128x256 / 256x128 pixels x channels)
bool conv8p_supported(const IgemmDesc& d, int variant);
int launch_conv8p(const IgemmDesc& d, int variant, hipStream_t stream, std::string& err);
// Split factor the heuristic would use for this problem (1 = none) given scratch for `part_floats` floats / `count_words` tiles.
int igemm_pick_splitk(const IgemmDesc& d, size_t part_floats, size_t count_words);
constexpr size_t kSplitKPartFloats = 2u << 20;  // 8 MB of f32 partials per workspace: splitk * M * N <= this
constexpr size_t kSplitKCountWords = 4096;
const char* igemm_family(const IgemmDesc& d);
int igemm_dot3_bn(const IgemmDesc& d);      // channel tile (128 / 256) of a dot3 launch = N / (number of partial-logit planes it writes)
int igemm_config_id(const IgemmDesc& d);    // tile configuration id launch_igemm picks (bf16 / fp16 path; -1 for f32)  // name of the kernel configuration launch_igemm picks
inline double igemm_flops(const IgemmDesc& d) { return 2.0 * d.M * d.N * (double)d.taps * d.Cin; }

}  // namespace soccdpt
