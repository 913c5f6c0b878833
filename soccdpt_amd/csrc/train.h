// Launchers of train.hip (backward / train-mode forward kernels, exact f32) and the training entry points of train_step.cpp.
#pragma once
#include <vector>
#include "internal.h"
#include "kernels.h"

namespace soccdpt {

// Every dgrad weight operand of one backward pass staged by a handful of launches instead of one per layer (round 4: 81 launches, 0.7 ms per bf16-amp step).
// kind 0: [R][C] -> [C][R] (tr_transpose / tr_transpose16 with Rp = R); kind 1: conv weight [R = N][C][3][3] -> tr_conv_w_dgrad's layout.  Same
// element conversions as the single forms (fmt: -1 f32, 0 bf16, 1 / 2 fp16, 3 x3), so the staged copies are bit-identical to theirs.
struct TrBatchEntry {
    const float* src;
    void* dst;
    int R, C, kind, tile0;   // tile0: first block of this entry in the launch's grid
};
constexpr int kTrBatchMax = 64;
struct TrBatchTable {
    TrBatchEntry e[kTrBatchMax];
    int n;
};
int tr_weight_batch(const TrBatchTable& t, int total_tiles, int fmt, hipStream_t st, std::string& err);
int tr_transpose(const float* in, float* out, int R, int C, int Rp, hipStream_t st, std::string& err);
int tr_transpose16(const float* in, uint16_t* out, int R, int C, int Rp, int f16, hipStream_t st, std::string& err);
int tr_im2colT(const float* halo, float* out, int B, int H, int W, int C, size_t Mp, hipStream_t st, std::string& err);
int tr_im2colT16(const float* halo, uint16_t* out, int B, int H, int W, int C, size_t Mp, int f16, hipStream_t st, std::string& err);
int tr_dy_halo_T(const float* dy, void* out, int out16, int B, int r, int N, int margin, int ld, hipStream_t st, std::string& err, int rpp = 0);
int tr_x_halo_T_x3(const float* halo, void* out, int B, int r, int C, int col0, int ld, int rpp, hipStream_t st, std::string& err);
int tr_wgrad_permute9(const float* in, float* out, int N, int C, hipStream_t st, std::string& err);
int tr_conv_w_dgrad(const float* w, float* out, int N, int C, hipStream_t st, std::string& err);
int tr_conv_w_dgrad16(const float* w, uint16_t* out, int N, int C, int f16, hipStream_t st, std::string& err);
int tr_wgrad_permute(const float* in, float* out, int N, int C, hipStream_t st, std::string& err);
int tr_to_halo(const float* in, float* out, int B, int H, int W, int C, hipStream_t st, std::string& err);
int tr_to_halo_full(const float* in, void* out, int B, int H, int W, int C, int fmt, hipStream_t st, std::string& err);   // whole image incl. a zero border
int tr_to_halo16(const float* in, uint16_t* out, int B, int H, int W, int C, int f16, hipStream_t st, std::string& err);
int tr_from_halo(const float* halo, float* out, int B, int H, int W, int C, int accumulate, hipStream_t st, std::string& err);
int tr_cvt16_pair(const float* in0, uint16_t* out0, size_t n0, const float* in1, uint16_t* out1, size_t n1, int f16, hipStream_t st, std::string& err);   // f32 -> bf16 / IEEE fp16 of two tensors, one launch
int tr_colsum(const float* a, const float* b, float* out, float* scratch, size_t M, int N, int accumulate, hipStream_t st, std::string& err);
int tr_colsum2(const float* a, const float* b, float* out_ab, float* out_a, float* scratch, size_t M, int N, hipStream_t st, std::string& err);   // sum a*b and sum a in one pass
int tr_axpy(float* y, const float* x, size_t n, hipStream_t st, std::string& err);
int tr_relu_bwd(const float* dy, const float* ref, const float* add, float* dx, size_t n, hipStream_t st, std::string& err);
int tr_relu_bwd_halo(const float* dy, const float* ref_halo, const float* add, float* dx, int B, int H, int W, int C, hipStream_t st, std::string& err);
int tr_gelu_bwd(const float* dy, const float* pre, float* dx, size_t n, hipStream_t st, std::string& err);
int tr_ln_bwd(const float* y, const float* g, const float* dout, float* dy, float* xhat, int M, int C, float eps, hipStream_t st, std::string& err);
int tr_bilinear_bwd(const float* dhi, float* dlo, int B, int h, int w, int H, int W, int C, int accumulate, hipStream_t st, std::string& err);
int tr_bn_stats(const float* x, float* stats, float* rmean, float* rvar, void* scratch, int C, size_t M, float eps, float momentum, hipStream_t st, std::string& err);
int tr_bn_relu_dropout_fwd(const float* x, const float* stats, const float* gamma, const float* beta, float* out, uint8_t* keep, size_t M, int C, float p, uint32_t seed, hipStream_t st, std::string& err);
int tr_bn_relu_dropout_bwd_pre(const float* dout, const float* out, const uint8_t* keep, float* dz, size_t n, float p, hipStream_t st, std::string& err);
int tr_bn_bwd(const float* dz, const float* x, const float* stats, const float* gamma, const float* dbeta, const float* dgamma, float* dx, size_t M, int C, hipStream_t st, std::string& err);
int tr_bn_xhat(const float* x, const float* stats, float* xh, size_t M, int C, hipStream_t st, std::string& err);
int tr_smallk_fwd(const float* x, const float* w, const float* bias, float* out, size_t M, int C, int K, hipStream_t st, std::string& err);
int tr_smallk_dgrad(const float* dl, const float* w, float* dx, size_t M, int C, int K, hipStream_t st, std::string& err);
int tr_smallk_wgrad(const float* dl, const float* x, float* dw, float* scratch, size_t M, int C, int K, hipStream_t st, std::string& err);
int tr_seg_act_bwd(const float* dseg, const float* seg, float* dup, int B, int K, int S, int sigmoid, hipStream_t st, std::string& err);
int tr_depth_tail_fwd(const float* e, const float* w4, const float* b4, float* inv, size_t M, int K, hipStream_t st, std::string& err);
int tr_depth_tail_bwd(const float* dinv, const float* inv, const float* e, const float* w4, float* de, float* rowterm, size_t M, int K, hipStream_t st, std::string& err);
int tr_merge_scatter(const float* dg, float* dx, int B, int R, int C, hipStream_t st, std::string& err);
int tr_patch_im2col(const float* x, float* out, int B, int S, hipStream_t st, std::string& err);
int tr_pad_cols(const float* in, float* out, int N, int cin, int cout, hipStream_t st, std::string& err);
int tr_attention_bwd(const float* qkv, const float* attn_out, const float* dO, const float* table, const float* scale, float* dS, float* rowstat, float* dscale_part, float* part, float* dqkv, int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err);
// MFMA form of tr_attention_bwd (train_attn.hip): no `part` scratch; rowstat holds {m + ln l, delta}; dscale_part has tr_attention_bwd_mfma_slots(ws) per (window, head)
int tr_attention_bwd_mfma(const float* qkv, const float* attn_out, const float* dO, const float* table, const float* scale, float* dS, float* rowstat,
                          float* dscale_part, float* dqkv, int B, int res, int ws, int shift, int heads, hipStream_t st, std::string& err, int op = 0);
int tr_attention_bwd_mfma_slots(int ws);
int tr_cvt_x3_pair(const float* in0, void* out0, size_t n0, const float* in1, void* out1, size_t n1, hipStream_t st, std::string& err);   // two f32 -> x3 conversions, one launch
// Weight gradient from operands as stored (train_wgrad_tn.hip): out[Nout][taps * C] = sum_k A[k][n] B[k + shift(tap)][c], 16-bit operands
bool tr_wgrad_tn_ok(size_t K, int Nout, int C, int taps);
// Deferred reduction of the weight-gradient partials (round 6): every tr_wgrad_tn call used to be followed by its own tn_reduce launch -- 74 launches of ~14 us per
// bf16-amp step whose work is a few hundred KB each.  With a TnDefer the partials of successive calls stay in an arena (bump allocation, flushed when full) and ONE
// batched launch per <= 48 gradients sums them -- in split order, like tn_reduce_kernel: the same bits -- at the end of the backward pass (tn_flush).  perm_C > 0: the
// batched launch also writes the 3x3 gradient from the kernel's tap-major [N][9][C] into the parameter layout [N][C][3][3] (tr_wgrad_permute's job).
struct TnPending { const float* part; float* out; const float* bpart; float* bout; unsigned long long n4; int splits, nb, perm_N, perm_C; };
struct TnDefer {
    float* arena = nullptr;
    size_t cap = 0, used = 0;
    std::vector<TnPending> pend;
};
int tn_flush(TnDefer& d, hipStream_t st, std::string& err);
int tr_wgrad_tn(const uint16_t* A, long ldA, const uint16_t* B, long ldB, size_t K, int Nout, int C, int taps, int rp, int f16, float* part, size_t part_floats,
                float* out, hipStream_t st, std::string& err, float* bias_out = nullptr, TnDefer* defer = nullptr, int perm_C = 0);
int tr_attn_param_grads(float* dS, const float* dscale_part, const float* table, const float* ls, const float* w0, const float* b0, const float* w2, float* dtable, float* dt, float* hid, float* dls, float* dw0, float* db0, float* dw2, int nwin, int ws, int pws, int heads, hipStream_t st, std::string& err, int dscale_slots = 0);
int tr_drop_path_fill(float* out, int B, float p, unsigned seed, unsigned stream_id, hipStream_t st, std::string& err);
int tr_scale_rows(const float* in, float* out, const float* scale, size_t M, int C, int rows_per_scale, hipStream_t st, std::string& err);
int tr_unscale_check(float* g, size_t n, float inv_scale, int* found, hipStream_t st, std::string& err);
int tr_qv_bias_grad(const float* dqkv_bias, float* dq, float* dv, int C, hipStream_t st, std::string& err);

// train_step.cpp: SOccDPT_V3 training step (model/SOccDPT.py:660-685 in train mode + autograd), SOCCDPT_PREC_F32 only
size_t train_workspace_bytes(Handle& h, int B);
int train_backward_encoder(Handle& h, int B, const float* const* d_feat, void* ws, size_t ws_bytes, hipStream_t st, std::string& err);
int train_workspace_tensor(Handle& h, int B, const char* name, size_t* byte_offset, size_t* elems);
int train_forward(Handle& h, const float* x, int B, float* inv, float* seg, void* ws, size_t ws_bytes, float dropout_p, unsigned seed, hipStream_t st, std::string& err);
int train_backward(Handle& h, const float* x, int B, const float* d_inv, const float* d_seg, void* ws, size_t ws_bytes, hipStream_t st, std::string& err);

}  // namespace soccdpt
