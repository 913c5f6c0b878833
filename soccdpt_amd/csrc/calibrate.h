// In-product calibration of the SOCCDPT_PREC_MIXED precision map (soccdpt_prec_calibrate): kernels (calibrate_kernels.hip) and the procedure
// (calibrate.cpp).  Replaces the out-of-tree tools/precision_map.py of round 4 for everything but the per-group cost measurement, whose
// results ship as a compiled-in table (prec_cost_table.h).
#pragma once
#include <algorithm>
#include <string>

#include "internal.h"

namespace soccdpt {

constexpr int kCalibQuantities = 7;          // feat0..3, path1, inv, seg_logits (the order of soccdpt_calib_report::err_*)
constexpr int kCalibPartialBlocks = 1024;    // blocks (and partial pairs) of the squared-difference reduction

int launch_calib_decode(const void* src, int kind, int B, int H, int W, int C, float* dst, hipStream_t st, std::string& err);
int launch_calib_sqdiff(const float* a, const float* b, size_t n, double* partial, double* out2, hipStream_t st, std::string& err);
int launch_calib_fingerprint(const float* w, size_t n, unsigned long long mult, unsigned long long* out, hipStream_t st, std::string& err);

// calibrate.cpp
size_t calib_scratch_bytes(Handle& h, int B);
int calib_run(Handle& h, const float* x, int B, const soccdpt_calib_options& opt, void* prepared, size_t prepared_bytes, void* ws, size_t ws_bytes, void* scratch,
              size_t scratch_bytes, soccdpt_calib_report* rep, hipStream_t st, std::string& err);
// 64-bit fingerprint of four bound tensors (decoder, heads, encoder); `tmp` = 8 bytes of device memory.  -> 0 ok, 1 a tensor is not bound, < 0 error
int calib_fingerprint(Handle& h, unsigned long long* tmp, hipStream_t st, unsigned long long* out, std::string& err);
// fingerprint of the bound weights (a few named tensors, 64-bit sums of their bit patterns) against the synthetic draw the shipped map was
// derived from; `tmp` = 8 bytes of device memory.  -> 1 same weights, 0 other weights, < 0 error
int calib_weights_are_the_shipped_draw(Handle& h, unsigned long long* tmp, hipStream_t st, std::string& err);

}  // namespace soccdpt
