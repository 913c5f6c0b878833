// Descriptor of one implicit-GEMM launch (igemm.hip): out[m][n] = sum_k X[m][k] * Wt[n][k].
// X rows are either plain rows of a [M][K] bf16 matrix (Linear layers, 1x1 convs) or the
// 3x3 neighbourhoods of a zero-haloed NHWC bf16 image (the decoder's 3x3 convolutions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>

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
    int M = 0, N = 0;
    int Cin = 0;      // channels per tap (K of a plain GEMM)
    int taps = 1;     // 1 (GEMM / 1x1) or 9 (3x3, pad 1)
    // plain mode: row m at X + m*ldx.  conv mode (taps == 9): X is a Halo image [B][H+2][W+2][Cin], M = B*H*W
    int ldx = 0;
    int H = 0, W = 0;
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
    // fused 1x1 tail of the depth head: out_dot[m] = relu(sum_n act(v)[n]*dot_w[n] + dot_b)   (N <= 32)
    const float* dot_w = nullptr;
    float dot_b = 0.f;
    float* out_dot = nullptr;
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
    int tune = -1;  // kernel configuration id (igemm.hip); -1 = heuristic
};

int launch_igemm(const IgemmDesc& d, hipStream_t stream, std::string& err);
// Split factor the heuristic would use for this problem (1 = none) given scratch for `part_floats` floats / `count_words` tiles.
int igemm_pick_splitk(const IgemmDesc& d, size_t part_floats, size_t count_words);
constexpr size_t kSplitKPartFloats = 2u << 20;  // 8 MB of f32 partials per workspace: splitk * M * N <= this
constexpr size_t kSplitKCountWords = 4096;
const char* igemm_family(const IgemmDesc& d);
int igemm_config_id(const IgemmDesc& d);    // tile configuration id launch_igemm picks (bf16 / fp16 path; -1 for f32)  // name of the kernel configuration launch_igemm picks
inline double igemm_flops(const IgemmDesc& d) { return 2.0 * d.M * d.N * (double)d.taps * d.Cin; }

}  // namespace soccdpt
