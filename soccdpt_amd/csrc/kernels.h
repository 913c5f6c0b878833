// Launchers of the non-GEMM kernels (elementwise.hip, attention.hip).
// `hf` selects the operand format of bf16_t buffers: 0 = bf16, 1 = IEEE fp16, 3 = x3 split fp16 (4 bytes per element; half16.h) where a launcher says so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/soccdpt_hip.h"
#include "igemm.h"

namespace soccdpt {

inline int check_launch(const char* what, std::string& err) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return 1;
    }
    return 0;
}

// elementwise.hip
int launch_patch_embed(const float* x, const float* w, const float* bias, const float* g, const float* beta, float* xf, bf16_t* xb,
                       int hf, int B, int S, int C0, hipStream_t st, std::string& err);
// row_scale (optional): the normalised branch of row m is multiplied by row_scale[m / rows_per_scale] before the residual add (stochastic depth
// of the training step: per-sample DropPath scales)
// hf_halo: operand format of `halo` when it differs from xb's (SOCCDPT_PREC_MIXED: the hooked map feeds the decoder, xb the next block); -1 = hf
int launch_ln_residual(const float* y, const float* g, const float* beta, float* xf, bf16_t* xb, bf16_t* halo, float* halo_f32, int hf, int M,
                       int C, int residual, int res, int merge, hipStream_t st, std::string& err, const float* row_scale = nullptr, int rows_per_scale = 1,
                       int hf_halo = -1);
int launch_merge_gather(const void* in, void* out, int B, int R, int C, int elem_bytes, hipStream_t st, std::string& err);
int launch_bilinear(const void* in, int in_is_bf16, float* out_f32, bf16_t* out_bf16, float* out_f32_halo, int out_halo, int hf, int B, int h,
                    int w, int H, int W, int C, hipStream_t st, std::string& err);
int launch_seg_tail(const void* feat, int feat_is_f32, int hf, const float* w, const float* bias, float* tmp, float* seg, int B, int h, int wd,
                    int sigmoid, hipStream_t st, std::string& err);
// the classifier was fused into the seg head's convolution (IgemmDesc::dot3): add the n-tiles' partial logits [ntiles][M][4] + bias -> tmp [M][3], then up-sample + activate
int launch_seg_tail_parts(const float* part, int ntiles, const float* bias, float* tmp, float* seg, int B, int h, int wd, int sigmoid, hipStream_t st, std::string& err);
int launch_patch_w(const float* w, float* out, int C0, hipStream_t st, std::string& err);
int launch_cvt_bf16(const float* in, bf16_t* out, size_t n, int hf, hipStream_t st, std::string& err);
int launch_conv_w(const float* in, const float* scale, void* out, int out_is_f32, int hf, int Cout, int Cin, hipStream_t st, std::string& err);
int launch_bn_fold(const float* g, const float* b, const float* mean, const float* var, float* scale, float* shift, int C, hipStream_t st,
                   std::string& err);
int launch_qkv_bias(const float* q, const float* v, float* out, int C, hipStream_t st, std::string& err);
int launch_logit_scale(const float* ls, float* out, int H, hipStream_t st, std::string& err);
int launch_cpb_table(const float* w0, const float* b0, const float* w2, float* table, int ws, int pws, int H, hipStream_t st,
                     std::string& err);

// depth_tail.hip: bilinear x2 + conv3x3(128->32) + ReLU + conv1x1(32->1) + ReLU fused (bf16 mode)
int launch_depth_tail(const bf16_t* d1, const bf16_t* wt, const float* bias, const float* w4, float b4, float* out, int hf, int B, int h,
                      int w, hipStream_t st, std::string& err);

// metrics.hip
size_t metrics_scratch_bytes(int B, int C);
int launch_depth_metrics(const float* pred, const float* gt, const uint8_t* mask, int B, size_t npix, float* out, void* scratch, hipStream_t st,
                         std::string& err);
int launch_iou_metrics(const float* pred, const float* gt, int B, int C, size_t npix, float* out, void* scratch, hipStream_t st, std::string& err);

// loss.hip: SSI + BCE training criterion at camera resolution, value and gradient w.r.t. the network outputs
size_t loss_scratch_bytes(int B, int H, int W, int h, int w);
int launch_training_loss(int B, int H, int W, int h, int w, int C, int compute_ss, float alpha, float w_d, float w_s, const float* inv,
                         const float* seg, const float* y_disp, const uint8_t* mask_disp, const float* y_seg, const uint8_t* mask_seg,
                         float* out, float* d_inv, float* d_seg, void* scratch, hipStream_t st, std::string& err);

// gt_occ.hip: ground-truth occupancy generator (counting voxelisation of a disparity frame + class map, float64 like numpy)
int launch_gt_occupancy(int B, int H, int W, int C, const double* intr, const double* pc_scale, const double* pc_shift, const double* rot27,
                        const float* occ_shape, const int* grid, float threshold, const float* disparity, const int32_t* seg_class, float* depth,
                        double* points, uint32_t* counts, uint8_t* occ, hipStream_t st, std::string& err);

// upsample_bwd.hip: backward of the projection stage's differentiable outputs (bicubic + clamp, nearest, back-projected points)
size_t upsample_bwd_scratch_bytes(const soccdpt_config& cfg, int B, int h);
int launch_upsample_bwd(const soccdpt_config& cfg, const float* inv_up, const float* d_inv_up, const float* d_seg_up, const float* d_points, int B, int h, int w,
                        float* d_inv, float* d_seg, void* scratch, hipStream_t st, std::string& err);

// input_transform.hip: uint8 HWC frame -> bicubic (OpenCV 8-bit fixed point) resize -> (x - mean) / std -> float32 CHW
int launch_input_transform_u8(const uint8_t* img, int B, int Hs, int Ws, int Hd, int Wd, const double* mean, const double* stdv, float* out,
                              hipStream_t st, std::string& err);

// mlp_fused.hip: x <- x + LN(fc2(GELU(fc1(x_op)))) in one launch for the narrow stages (C = 96 / 128 / 192 / 256), 16-bit operands
bool mlp_ln_supported(int C);
int launch_mlp_ln(const bf16_t* xop, float* xf, const bf16_t* w1, const float* b1, const bf16_t* w2, const float* b2, const float* g, const float* be,
                  bf16_t* xop_out, bf16_t* halo, int hf, int M, int C, int H, int W, int merge, hipStream_t st, std::string& err);

// adam.hip: fused multi-tensor Adam (host arrays of device pointers)
int launch_adam(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                const size_t* sizes, double lr, double beta1, double beta2, double eps, double weight_decay, int step, hipStream_t st,
                std::string& err);

// hybrid.hip: non-GEMM kernels of the ViT-hybrid encoder (dpt_hybrid_384).  out_mode: operand format written, 0 bf16 / 1 fp16 / 2 f32
int launch_ws_conv_w(const float* w, void* out, int out_mode, int Cout, int Cin, int k, int Kpad, float eps, hipStream_t st, std::string& err);
int launch_stem_im2col(const float* x, void* A, int out_mode, int B, int S, hipStream_t st, std::string& err);
struct GnApplyArgs {
    const float* raw = nullptr;      // [M][C] f32 convolution output
    const float* stats = nullptr;    // [B][C/cpg][2] {mean, rstd} (igemm ST epilogue)
    const float *gamma = nullptr, *beta = nullptr;
    const float *raw2 = nullptr, *stats2 = nullptr, *gamma2 = nullptr, *beta2 = nullptr;   // projection shortcut: + GN2(raw2)
    const float* res = nullptr;      // identity shortcut: + res [M][C]
    // deferred statistics (igemm statistics epilogue): per-tile partials of the producing convolution(s), tps = M tiles per sample of that launch; the kernel adds
    // them up itself and workgroup 0 of every sample writes {mean, rstd} to stats / stats2
    const float *part = nullptr, *part2 = nullptr;
    int tps = 0, tps2 = 0;
    float eps = 1e-5f;
    float* out_f32 = nullptr;        // [M][C]
    void* out_op = nullptr;          // [M][C] operand type
    void* out_halo = nullptr;        // [B][H+2][W+2][C] operand type (zero halo untouched)
    int halo_mode = -1;              // operand format of out_halo when it differs from out_op's (SOCCDPT_PREC_MIXED); -1 = the launch's out_mode
    int relu = 1;
    size_t M = 0;
    int HW = 0, W = 0, C = 0, cpg = 0;
};
int launch_gn_apply(const GnApplyArgs& a, int out_mode, hipStream_t st, std::string& err);
int launch_gn_finish(const float* part, float* stats, int B, int tps, int G, int hw, int cpg, float eps, hipStream_t st, std::string& err);   // partials -> {mean, rstd}
int launch_gn_relu_maxpool(const float* raw, const float* stats, const float* gamma, const float* beta, void* out, int out_mode, int B, int Hi, int C, int cpg,
                           hipStream_t st, std::string& err);
int launch_vit_tokens_ln(const float* y, const float* cls, const float* pos, float* xf, const float* g, const float* be, void* xb, int out_mode, int B, int ntok,
                         int C, float eps, hipStream_t st, std::string& err);
int launch_ln_rows(float* xf, const float* g, const float* be, void* xb, int out_mode, int rows, int C, float eps, hipStream_t st, std::string& err);
int launch_pos_embed_resize(const float* pos, float* out, int g0, int g, int C, hipStream_t st, std::string& err);

// vit_attention.hip: global softmax attention of a ViT block (timm Attention.forward): qkv [B*N][3*heads*64] -> out [B*N][heads*64],
// softmax(q k^T / 8) v per (sample, head), N tokens (577 for dpt_hybrid_384).  prec: SOCCDPT_PREC_* of qkv / out.
// out_x3 != 0 (prec == SOCCDPT_PREC_F16 only): fp16 q, k, v in, `out` written in the x3 operand format from the f32 accumulators (SOCCDPT_PREC_MIXED)
int launch_vit_attention(const void* qkv, void* out, int prec, int B, int N, int heads, hipStream_t st, std::string& err, int out_x3 = 0);

// attention.hip
// bias_acc: CPB bias pre-arranged in MFMA accumulator order, see attention.hip
size_t attn_bias_elems(int ws, int heads);
int launch_attn_bias(const float* table, float* bias_acc, int ws, int heads, hipStream_t st, std::string& err);
// out_x3 != 0 (fp16 kernels only): `out` is written in the x3 operand format from the f32 accumulators (SOCCDPT_PREC_MIXED: the proj GEMM is an x3 launch)
int launch_window_attention(const bf16_t* qkv, const float* bias_acc, const float* scale, bf16_t* out, int hf, int B, int res, int ws,
                            int shift, int heads, hipStream_t st, std::string& err, int out_x3 = 0);

// Winograd F(2x2, 3x3) form of a 3x3 stride-1 convolution over a 16-bit zero-halo NHWC image (wino.hip, round 6).  U = launch_wino_weights' output.
struct WinoArgs {
    const void* X = nullptr;       // [B][H + 2][W + 2][C] 16-bit
    const void* U = nullptr;       // [C / 32][16][N][32] 16-bit
    int B = 0, H = 0, W = 0, C = 0, N = 0;
    const float* bias = nullptr;
    const float* res1 = nullptr;   // [M][N] f32
    const float* res2 = nullptr;   // [B][res2_h][res2_w][N] f32, bilinearly sampled
    int res2_h = 0, res2_w = 0;
    int act = 0, act_on_f32 = 0;
    float* out_f32 = nullptr;
    void* out_op = nullptr;
    int out_halo = 0, out_x3 = 0;
    int hf = 1;                    // 1 = fp16 operands, 0 = bf16
    unsigned long long* stamps = nullptr;   // diagnostics
};
bool wino_supported(int H, int W, int C, int N);
size_t wino_weight_elems(int N, int C);
int launch_wino_weights(const float* w /*[N][C][3][3]*/, const float* scale /*[N] or null*/, void* U, int hf, int N, int C, hipStream_t st, std::string& err);
int launch_wino_conv(const WinoArgs& a, hipStream_t st, std::string& err);

// The same with the qkv projection inside the kernel (attention_qkv.hip, round 6): x = the block input [M][C] (16-bit operand copy), wqkv = the prepared
// [3C][C] weight (plain 16-bit, or x3 pairs when x2w != 0), qkv_bias = cat(q_bias, 0, v_bias) f32.  16 x 16 and 8 x 8 windows.
bool window_attention_qkv_supported(int ws, int C, int x2w);
int launch_window_attention_qkv(const bf16_t* x, const void* wqkv, const float* qkv_bias, const float* bias_acc, const float* scale, bf16_t* out, int hf, int x2w,
                                int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err, int out_x3 = 0, unsigned long long* stamps = nullptr);

// x3 != 0: `out` is written in the x3 split-fp16 operand format (half16.h) for the proj GEMM of SOCCDPT_PREC_F16X3
int launch_window_attention_f32(const float* qkv, const float* bias_acc, const float* table, const float* scale, float* out, int B, int res,
                                int ws, int shift, int heads, hipStream_t st, std::string& err, int x3 = 0);

}  // namespace soccdpt
