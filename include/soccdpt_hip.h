/* libsoccdpt_hip.so — C ABI of the MI355X-native SOccDPT_V3 forward path.
 *
 * The reference has no FFI layer: its "operator API" is nn.Module.forward
 * (SURVEY.md §8b).  This header is the boundary a host binding replaces it
 * with; each entry point cites the reference interface it stands in for
 * (paths relative to /root/reference/SOccDPT).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; `stream` is a hipStream_t passed as void*.
 *  - every pointer named dev_* / documented "device" is a device pointer owned
 *    by the caller (PyTorch allocates; the library borrows for the call and
 *    never frees).  No hidden allocation or synchronisation in *_forward /
 *    stage calls: all launches go to `stream`.
 *  - return value 0 = ok; non-zero = error, text via soccdpt_last_error().
 *  - a handle is bound to the device current at soccdpt_create and is not
 *    thread-safe; one handle per rank.
 */
#ifndef SOCCDPT_HIP_H
#define SOCCDPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a public struct or signature changes (a binding built against another version is refused by soccdpt_create):
 *   1 rounds 1-2; 2 round 3 (soccdpt_igemm_args gained stamps / sk_defer, soccdpt_train_forward changed arity);
 *   3 round 4 (SOCCDPT_PREC_MIXED + the precision map entry points, soccdpt_sizeof, soccdpt_occ_zero / soccdpt_occ_set,
 *     soccdpt_set_stage_xcd / soccdpt_stage_xcd_status);
 *   4 round 5 (the opt-in XCD-local persistent stage kernel and its three entry points are gone: measured 13-15 % slower than the launch
 *     chain in round 4; soccdpt_prec_calibrate and the precision-map source query are new). */
#define SOCCDPT_ABI_VERSION 5

/* backbone ids: model/loader.py:65-77 (model_type switch), model/blocks.py:59-78 */
#define SOCCDPT_BACKBONE_SWIN2T16_256 0 /* dpt_swin2_tiny_256 */
#define SOCCDPT_BACKBONE_SWIN2B24_384 1 /* dpt_swin2_base_384 */
#define SOCCDPT_BACKBONE_VITB_RN50_384 2 /* dpt_hybrid_384: ResNetV2-50 stem + ViT-B/16 (model/loader.py:115-120, backbones/vit.py:147-258) */

/* dtype codes for soccdpt_bind_weight */
#define SOCCDPT_DTYPE_F32 0

/* arithmetic of the encoder/decoder GEMMs and convolutions */
#define SOCCDPT_PREC_BF16 0 /* bf16 MFMA operands, f32 accumulate (the benchmarked configuration) */
#define SOCCDPT_PREC_F32 1  /* f32 operands, exact-f32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32): the parity mode,
                               1/16 of the bf16 MFMA rate */
#define SOCCDPT_PREC_F16 2  /* IEEE fp16 MFMA operands (11-bit significand), f32 accumulate and f32 residual streams: the
                               same kernels and MFMA rate as BF16 with 8x smaller operand rounding; what the reference's
                               optimize=True path (model/loader.py:126-139, .half()) computes in */
#define SOCCDPT_PREC_F16X3 3 /* split-operand fp16: every GEMM / convolution operand is an fp16 pair (hi, lo * 2^11) and every product three
                               fp16 MFMAs (hi hi + hi lo + lo hi) with f32 accumulation: ~22 significand bits at 1/3 of the 16-bit MFMA
                               rate (the f32 MFMA runs at 1/16); attention, normalisations and residual streams in f32.  The fast
                               parity-grade mode for models whose reference arithmetic is fp32 (dpt_hybrid_384: model/loader.py:115-120).
                               Valid operand magnitudes: |v| < 65520 (beyond: +-inf, as an IEEE fp16 conversion gives -- never clipped);
                               22+ significand bits down to ~6e-5, fewer below, 11 bits below ~6e-8 (csrc/half16.h x3_split) */
#define SOCCDPT_PREC_MIXED 4 /* per-site mix of the two above: every GEMM / convolution group runs fp16 operands (one MFMA per product) unless the
                               precision map promotes it to x3 (three).  The default map of each backbone is the cheapest assignment found by
                               tools/precision_map.py that keeps depth, logits, path_1 and every hooked feature map within HALF the north
                               star's 1e-3 of the fp32 reference (model/loader.py:126-139 computes in fp32); soccdpt_prec_map_set edits it. */

#define SOCCDPT_PREC_F16X2W 5 /* a precision-MAP value (and a soccdpt_op_igemm operand mode), not a handle mode: one-sided split -- fp16 activations, the
                               group's weights as x3 pairs, two fp16 MFMAs per product.  Removes the WEIGHT rounding of a launch site (a median 72 % of
                               its fp16 rounding variance on the synthetic weights, profiles/r05_x2_variance_tiny256.json) at 1.5x the operand bytes of an
                               fp16 launch; the producers of the site's activations keep writing 2-byte operands. */

/* Constructor constants of SOccDPT / SOccDPT_V3 (model/SOccDPT.py:134-245,626-679). */
typedef struct soccdpt_config {
    int32_t abi_version;  /* SOCCDPT_ABI_VERSION */
    int32_t backbone;     /* SOCCDPT_BACKBONE_* */
    int32_t num_classes;  /* model/SOccDPT.py:139 (must be 3: :347-349 reshapes xyz with it) */
    int32_t features;     /* 256 (model/SOccDPT.py:183) */
    int32_t sigmoid;      /* 1: nn.Sigmoid, 0: ScaledTanh (model/SOccDPT.py:655-658) */
    int32_t compute_occ;  /* model/SOccDPT.py:152 */
    int32_t precision;    /* SOCCDPT_PREC_* */
    int32_t cam_width;    /* Camera.width  (model/SOccDPT.py:227) */
    int32_t cam_height;   /* Camera.height (model/SOccDPT.py:228) */
    float fx, fy, cx, cy; /* model/SOccDPT.py:222-225, cast to f32 */
    int32_t grid[3];      /* grid_size (model/SOccDPT.py:145) */
    float occupancy_shape[3]; /* grid_size/scale as f32 (model/SOccDPT.py:175-181) */
    float pc_scale[3];    /* model/SOccDPT.py:148 */
    float pc_shift[3];    /* model/SOccDPT.py:149 */
    float rot[27];        /* Ra, Rb, Rc row-major f32, built on the host exactly as
                             rotate_points does (model/SOccDPT.py:74-111) */
} soccdpt_config;

/* ---- lifetime: replaces SOccDPT_V3.__init__ / load_model (model/loader.py:13-138) ---- */
int soccdpt_create(const soccdpt_config* cfg, void** handle);
void soccdpt_destroy(void* handle);
const char* soccdpt_last_error(void* handle); /* handle may be NULL: last create error */
int soccdpt_abi_version(void);
/* sizeof of the public structs as the LIBRARY was compiled (a binding checks its own layout against these): which = 0 soccdpt_config,
 * 1 soccdpt_igemm_args, 2 soccdpt_kernel_stat, 3 soccdpt_calib_report, 4 soccdpt_calib_options; unknown -> 0 */
size_t soccdpt_sizeof(int which);

/* ---- precision map (SOCCDPT_PREC_MIXED handles only) ----
 * Groups are named launch sites of the forward (one operand format per group: its GEMMs / convolutions, their weights and the
 * activation buffers they read):
 *   Swin-V2:  "s<stage>.b<block>.qkv" / ".proj" / ".fc1" / ".fc2" (the four Linear layers of a block; the attention core between qkv and proj
 *             always runs the fp16 kernel and hands proj its operands in proj's format), "merge<stage>" (PatchMerging reduction)
 *   hybrid:   "rn.s<stage>.c1" / ".c2" / ".c3" (per ResNetV2 stage: the 1x1 reduce convolutions + the shortcut projection -- both read the block
 *             input --, the 3x3 convolutions, the 1x1 expand convolutions; one group per stage until round 4), "pe" (patch-embedding
 *             projection), "vit.b<i>.qkv" / ".proj" / ".fc1" / ".fc2", "ro<k>" (ProjectReadout + 1x1 of act_postprocess3 / 4), "pp4" (its 3x3 / 2)
 *   decoder:  "lrn<l>" (scratch.layer<l+1>_rn), "ref<l>" (the four RCU convolutions of refinenet<l+1>), "oc<l>" (its out_conv),
 *             "head" (output_conv.0 and seg_head.0: both read path_1), "head.d2" (output_conv.2 + .4), "head.s1" (format in which
 *             the seg head's conv + BN + ReLU output is kept for the 1x1 classifier: F16 = fp16, F16X3 = plain f32 -- honoured only when that
 *             feature map exists, i.e. when "head" is F16X3 or SOCCDPT_SEG_DOT3_OFF is set: with a 16-bit "head" the classifier rides in the
 *             convolution's epilogue on the f32 accumulators and the entry is inert)
 * F16X2W is refused for "head.d2" and "head.s1" (the fused depth tail reads its filter as plain 16-bit; "head.s1" is a storage format): an exact name is an
 * error, a pattern skips them (they stay F16).  "head" takes it since round 6 (the seg head then runs convolution -> feature map -> classifier, as under F16X3).
 * `group` may end in '*' (prefix match) or be "*".  fmt = SOCCDPT_PREC_F16, SOCCDPT_PREC_F16X2W or SOCCDPT_PREC_F16X3.  Returns the number of groups
 * changed (>= 0) or a negative value on error.  Invalidates the prepared weights and the workspace (call soccdpt_prepare again). */
int soccdpt_prec_map_set(void* handle, const char* group, int fmt);
/* Writes "name=fmt name=fmt ..." (fmt 2 / 5 / 3, launch order) into buf; returns the length needed (excluding the terminator). */
int soccdpt_prec_map_get(void* handle, char* buf, int buf_bytes);

/* ---- calibration of the precision map on the weights bound to the handle (round 5) ----
 * The shipped maps were derived on one synthetic weight draw.  The reference runs whatever checkpoint load_net binds, in fp32
 * (model/SOccDPT.py:29-57,634-636, model/base_model.py:5-37): for any other weights the map is re-derived HERE, inside the library, against the
 * library's own exact-f32 arithmetic on the same weights and the caller's sample frames:
 *   1. an f32 twin of the handle (same bound weight pointers) runs the sample: the reference for the seven checked quantities (hooked feature
 *      maps 0-3, path_1, inverse depth, class logits; relative L2);
 *   2. all groups x3 but ONE in fp16, then that one in x2w -> the variance the group adds to each quantity in either format;
 *   3. greedy selection over single-group upgrades (fp16 -> x2w -> x3) by variance removed per microsecond (compiled-in per-group device-time
 *      costs, csrc/prec_cost_table.h) until every quantity's predicted error is under `budget`, checked by a measured run (tightened and
 *      repeated if the additive model was optimistic);
 *   4. measured prune: demote groups one level at a time, largest saving first, while the acceptance rule below still passes.
 * Acceptance rule (round 6): a map is accepted when its MEASURED worst error over the calibration frames is <= headroom x budget (default 0.85:
 * the frame-to-frame spread of the error is a few per cent, round 5 measured calibrated maps at 1.15 x budget on other frames) AND, when
 * `holdout` > 0, the last `holdout` frames of dev_x -- which take no part in the variance estimates or the selection -- come in under the budget
 * itself.  per_pixel_p999 > 0 adds an eighth constraint: the 99.9th percentile over pixels of |inv - ref| / max(|ref|, 1e-6) (inverse depth) must be
 * <= per_pixel_p999 (scaled into the same selection; the relative-L2 reading of the tolerance lets ~1 pixel in a thousand exceed it otherwise).
 * The handle's map is replaced by the result (soccdpt_prec_map_source() == 1) and its weights are prepared for it.  Synchronises; runs about
 * (2 x groups + promoted groups + 10) forwards of the sample.  All device memory comes from the caller: dev_prepared / dev_workspace as for
 * soccdpt_prepare / soccdpt_network at batch B, dev_scratch of soccdpt_prec_calibrate_scratch_bytes(handle, B) bytes.
 * report (host memory, may be NULL): what was measured. */
typedef struct soccdpt_calib_report {
    int32_t n_groups;          /* launch-site groups of this backbone */
    int32_t n_x3;              /* groups the calibrated map promotes to x3 */
    int32_t n_x2w;             /* ... and to x2w (fp16 activations, x3 weight pairs) */
    int32_t n_x3_shipped;      /* the shipped map's */
    int32_t n_x2w_shipped;
    int32_t forwards;          /* network forwards the calibration ran */
    int32_t met_budget;        /* 1: the calibrated map's measured worst error <= budget */
    int32_t shipped_met_budget;/* 1: the SHIPPED map met the budget on these weights and frames */
    float budget;
    float worst_calibrated, worst_shipped, worst_all_fp16, worst_all_x3;
    float err_calibrated[7], err_shipped[7]; /* feat0, feat1, feat2, feat3, path1, inv, seg_logits */
    float cost_us_calibrated, cost_us_shipped; /* sums of the compiled-in promotion costs of the x3 groups */
    /* ---- ABI 5 (round 6) ---- */
    int32_t calib_frames, holdout_frames;    /* B - holdout, holdout */
    int32_t met_headroom;      /* 1: calibration frames <= headroom x budget under the final map */
    int32_t met_holdout;       /* 1: held-out frames <= budget under the final map; -1: no held-out frames */
    float headroom;            /* the factor used */
    float worst_holdout, worst_holdout_shipped;  /* worst relative L2 of the seven quantities over the held-out frames: final map, shipped map */
    float err_holdout[7];
    float per_pixel_budget;    /* per_pixel_p999 of the options (0 = constraint off) */
    float inv_p999_calibrated, inv_max_calibrated;   /* per-pixel relative error of the inverse depth under the final map, calibration frames */
    float inv_p999_holdout, inv_max_holdout;         /* ... held-out frames */
    float inv_p999_all_fp16, inv_p999_all_x3;        /* the corner maps, calibration frames */
} soccdpt_calib_report;
typedef struct soccdpt_calib_options {
    int32_t struct_bytes;      /* sizeof(soccdpt_calib_options) as the caller compiled it */
    int32_t holdout;           /* last `holdout` frames of dev_x verify only (0 <= holdout < B) */
    float budget;              /* relative-L2 budget of the seven quantities */
    float headroom;            /* calibration frames are held to headroom x budget; 0 -> 0.85 */
    float per_pixel_p999;      /* optional bound on the 99.9th-percentile per-pixel relative error of the inverse depth; 0 = off */
} soccdpt_calib_options;
size_t soccdpt_prec_calibrate_scratch_bytes(void* handle, int B);
int soccdpt_prec_calibrate(void* handle, const float* dev_x, int B, float budget, void* dev_prepared, size_t prepared_bytes, void* dev_workspace,
                           size_t workspace_bytes, void* dev_scratch, size_t scratch_bytes, soccdpt_calib_report* report, void* stream);
/* The same with options (hold-out frames, head-room, per-pixel constraint); soccdpt_prec_calibrate(budget) = {holdout 0, headroom 0.85, no per-pixel bound}.
 * On ANY failure the handle keeps the map and source it had on entry (re-prepared for it).  A calibrated map belongs to the weights it was derived on:
 * soccdpt_prepare re-checks a fingerprint of them and, when other values have been bound or loaded since, falls back to source 3 (all x3). */
int soccdpt_prec_calibrate_ex(void* handle, const float* dev_x, int B, const soccdpt_calib_options* options, void* dev_prepared, size_t prepared_bytes,
                              void* dev_workspace, size_t workspace_bytes, void* dev_scratch, size_t scratch_bytes, soccdpt_calib_report* report, void* stream);
/* Where the handle's current map comes from: 0 = the shipped map, bound weights = the synthetic draw it was derived from (checked by a
 * fingerprint of a few tensors at soccdpt_prepare); 1 = soccdpt_prec_calibrate ran on the bound weights; 2 = edited through
 * soccdpt_prec_map_set; 3 = OTHER weights are bound and no calibration has run: the shipped map's "within tolerance" claim does not carry over
 * (measured: a second synthetic draw leaves the class logits at 1.4e-3 under it), so soccdpt_prepare has put EVERY group on x3 operands -- f32-grade
 * results at about 1.7x the step -- until soccdpt_prec_calibrate (or soccdpt_prec_map_set) says otherwise; the Python mirror prints a notice.
 * -1: not a SOCCDPT_PREC_MIXED handle, or not prepared yet. */
int soccdpt_prec_map_source(void* handle);

/* ---- weights: replaces BaseModel.load_net -> load_state_dict (model/base_model.py:5-37) ----
 * `key` is the reference's state-dict key, e.g.
 * "depth_net.pretrained.model.layers.0.blocks.0.attn.qkv.weight",
 * "depth_net.scratch.refinenet1.resConfUnit1.conv1.weight", "seg_head.1.running_var".
 * Unknown keys return non-zero (the host prints them like load_state_dict(strict=False)). */
int soccdpt_bind_weight(void* handle, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim);
int soccdpt_num_weights(void* handle);                     /* tensors the path consumes */
const char* soccdpt_weight_key(void* handle, int index);   /* their keys, for the host to iterate */

/* Bytes of caller-provided device memory for the re-laid-out (bf16, tap-major,
 * BN-folded, CPB-table) weights, and of per-forward scratch for batch B. */
size_t soccdpt_prepared_bytes(void* handle);
size_t soccdpt_workspace_bytes(void* handle, int B);
/* Workspace contract.  The 3x3 convolutions read one-pixel zero borders ("halos") that no kernel ever writes, and where those
 * borders lie inside the workspace depends on (B, stream count).  THE LIBRARY OWNS THAT INVARIANT: soccdpt_forward /
 * soccdpt_network zero-fill the first soccdpt_workspace_bytes(handle, B) bytes on `stream` whenever they see a
 * (dev_workspace pointer, B, stream count) tuple different from the previous call's (one hipMemsetAsync per layout change,
 * nothing in the steady state).  A caller that lets anything else write into the buffer between calls, or that frees it and
 * is handed the same address again, must call soccdpt_workspace_invalidate() so that the next forward zero-fills again.
 * The buffer may have any contents when first handed over.  A too-small workspace and a handle used while another device
 * is current are errors (non-zero return, soccdpt_last_error). */
int soccdpt_workspace_invalidate(void* handle);
/* Number of zero-fills the handle has issued so far (diagnostics / tests of the contract above). */
int soccdpt_workspace_zero_fills(void* handle);

/* Fold BN, convert/re-lay weights, build CPB bias tables.  Call after all binds and
 * again whenever bound weight values change.  `prepared` stays in use by forward. */
int soccdpt_prepare(void* handle, void* dev_prepared, size_t prepared_bytes, void* stream);

/* ---- the hot path: replaces SOccDPT_V3.forward (model/SOccDPT.py:681-685) ----
 * x        [B,3,S,S] f32 NCHW (S = 256 / 384 by backbone)
 * inv_up   [B,Hc,Wc] f32              clamped camera-resolution inverse depth (:270-290)
 * seg_up   [B,C,Hc,Wc] f32            nearest-upsampled class probabilities (:278-282)
 * points   [B,Hc,Wc,3] f32            un-rotated camera-frame points incl. the 3-pixel quirk (:301-353)
 * occ      [B,gx,gy,gz,C] f32 or NULL union-over-batch binary grid in every row (:374-463)
 * occ_bits (gx*gy*gz*C+31)/32 words   packed union grid of THIS call's frames (always written
 *                                     when compute_occ; the multi-GPU exchange moves only this) */
int soccdpt_forward(void* handle, const float* dev_x, int B, float* dev_inv_up, float* dev_seg_up, float* dev_points,
                    float* dev_occ, uint32_t* dev_occ_bits, void* dev_workspace, size_t workspace_bytes, void* stream);

/* Frames are independent through the network: the batch of a call is dealt to `n` sub-batches (1..8) that run
 * concurrently on the caller's stream plus n-1 library-owned streams, forked from and joined into the caller's
 * stream with events (no host synchronisation; results are stream-ordered on the caller's stream).  Changes the
 * workspace size/layout: query soccdpt_workspace_bytes again (the next forward re-zeroes the halos itself).
 * EXPERIMENTAL for n > 1: the library refuses n > 1 unless the environment variable SOCCDPT_ALLOW_MULTISTREAM=1 is set
 * (a wrong-result hazard of packed-f32 VALU code under kernel co-residency was worked around, not
 * root-caused; one stream is the supported configuration). */
int soccdpt_set_streams(void* handle, int n);
/* on != 0: soccdpt_network / soccdpt_forward capture their launch sequence into a hipGraph the second time they
 * see the same (x, outputs, workspace, B) pointers and replay it afterwards (one host launch per forward).  Callers
 * that pass fresh pointers every call simply keep running eagerly. */
int soccdpt_set_graph(void* handle, int on);

/* ---- stage-level entry points (parity tests, multi-GPU composition) ---- */

/* Encoder + decoder + heads: DPTDepthModel.forward + seg_head (model/dpt.py:142-232,
 * model/SOccDPT.py:660-674,682-683).  inv256 [B,S,S] f32, seg256 [B,C,S,S] f32. */
int soccdpt_network(void* handle, const float* dev_x, int B, float* dev_inv256, float* dev_seg256, void* dev_workspace,
                    size_t workspace_bytes, void* stream);

/* SOccDPT.get_semantic_occupancy + rotate_points + points_to_occupancy_grid index pass
 * (model/SOccDPT.py:264-463) fused into one kernel.  in_h/in_w: network output size.
 * ORs into dev_occ_bits (cleared first when clear_bits != 0; pass NULL to skip voxelisation). */
int soccdpt_project(void* handle, const float* dev_inv, const float* dev_seg, int B, int in_h, int in_w,
                    float* dev_inv_up, float* dev_seg_up, float* dev_points, uint32_t* dev_occ_bits, int clear_bits,
                    void* stream);

/* Backward of soccdpt_project's differentiable outputs: what torch autograd runs through get_semantic_occupancy (model/SOccDPT.py:264-353) when a
 * criterion written in torch ops is applied to the tuple SOccDPT_V3.forward returns in train mode (scripts/train_SOccDPT.py:365-391).
 * dev_inv_up [B,Hc,Wc] is the forward's clamped output (the clamp mask and the points' depth are read from it); dev_d_inv_up [B,Hc,Wc],
 * dev_d_seg_up [B,C,Hc,Wc], dev_d_points [B,Hc,Wc,3] are the upstream gradients (each may be NULL = zero).  Writes dev_d_inv [B,in_h,in_w] and
 * dev_d_seg [B,C,in_h,in_w] -- the arguments of soccdpt_train_backward.  Scratch: soccdpt_project_backward_scratch_bytes.  Deterministic gathers. */
size_t soccdpt_project_backward_scratch_bytes(void* handle, int B, int in_h);
int soccdpt_project_backward(void* handle, const float* dev_inv_up, const float* dev_d_inv_up, const float* dev_d_seg_up, const float* dev_d_points,
                             int B, int in_h, int in_w, float* dev_d_inv, float* dev_d_seg, void* dev_scratch, size_t scratch_bytes, void* stream);

/* dst |= src[0] | ... | src[n_sets-1]  (each set = soccdpt_occ_words() words). */
int soccdpt_occ_or(void* handle, uint32_t* dev_dst_bits, const uint32_t* dev_src_bits, int n_sets, void* stream);
/* packed bits -> dense f32 rows, identical in every batch row (model/SOccDPT.py:449-455). */
int soccdpt_occ_expand(void* handle, const uint32_t* dev_bits, int B, float* dev_occ, void* stream);
/* The same expansion in two halves for the multi-GPU path: soccdpt_occ_zero writes the B dense rows of zeros (no dependence on any rank's voxels:
 * issue it while the RCCL all-gather of the packed grids is in flight), soccdpt_occ_set then writes a 1 into every row for each set bit of the
 * union.  zero + set stores exactly what soccdpt_occ_expand stores. */
int soccdpt_occ_zero(void* handle, int B, float* dev_occ, void* stream);
int soccdpt_occ_set(void* handle, const uint32_t* dev_bits, int B, float* dev_occ, void* stream);
size_t soccdpt_occ_words(void* handle);

/* Number of kernel launches issued by the last soccdpt_network call (diagnostics). */
int soccdpt_last_launch_count(void* handle);
/* Kernels the calling thread has launched from this library since it was loaded (forward, criterion, backward, optimizer: every launch goes
 * through one macro, csrc/launch.h); the difference of two reads brackets a region, e.g. one training step (bench.py --train-step). */
unsigned long long soccdpt_launch_counter(void);

/* ---- evaluation metrics on the device (the step after the hot path; replaces the per-batch .cpu().numpy() round trip of
 * utils/__init__.py:161-332).  No handle: stateless; `scratch` = soccdpt_metrics_scratch_bytes(B, C) bytes of device memory.
 * soccdpt_metrics_depth: pred, gt [B][npix] f32, mask [B][npix] u8 -> out[0..6] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 over
 *   all masked pixels of the batch after per-image least-squares scale/shift alignment (loss/ssi_loss.py:5-32,
 *   utils/__init__.py:109-158,219-234); out[7+2b], out[8+2b] = scale, shift of image b.  out: 7 + 2B floats.
 * soccdpt_metrics_iou: pred, gt [B][C][npix] f32 -> out[b] = mean_c |p>0.5 & g>0.5| / (|p>0.5 | g>0.5| + 1e-7)
 *   (utils/__init__.py:314-330). */
size_t soccdpt_metrics_scratch_bytes(int B, int C);
int soccdpt_metrics_depth(const float* dev_pred, const float* dev_gt, const uint8_t* dev_mask, int B, size_t npix, float* dev_out,
                          void* dev_scratch, void* stream);
int soccdpt_metrics_iou(const float* dev_pred, const float* dev_gt, int B, int C, size_t npix, float* dev_out, void* dev_scratch,
                        void* stream);

/* ---- per-kernel device time (bench.py roofline) ----
 * While enabled, every kernel launched by soccdpt_network / soccdpt_project / soccdpt_occ_expand / soccdpt_forward carries a
 * HIP start / stop event pair bound to its own dispatch (hipExtLaunchKernelGGL): the pair's elapsed time is the kernel's begin -> end on
 * the device, the figure rocprofv3 --kernel-trace reports, and nothing is recorded between kernels.  soccdpt_profile_collect synchronises
 * the events, aggregates per kernel family (sum of kernel ms, launches, algorithmic FLOPs and algorithmic HBM bytes as defined in
 * DESIGN.md) and resets the recording. */
typedef struct soccdpt_kernel_stat {
    char name[48];
    int32_t launches;
    double ms;     /* sum of per-launch durations */
    double flops;  /* algorithmic: 2*M*N*K per GEMM/conv launch, 4*N*N*d per attention (window, head) */
    double bytes;  /* algorithmic HBM bytes (HBM-bound kernels), 0 where not defined */
} soccdpt_kernel_stat;
int soccdpt_profile_enable(void* handle, int on);
int soccdpt_profile_collect(void* handle, soccdpt_kernel_stat* out, int max_entries, int* n_entries);

/* ---- training criterion (first component of the patch-wise training step) ----
 * loss = loss_depth_w * ScaleAndShiftInvariantLoss(clamp(bicubic(inv)), y_disp, mask_disp)      (loss/ssi_loss.py:5-158)
 *      + loss_seg_w * BCELoss(mean)(nearest(seg)[mask_seg], y_seg[mask_seg])                    (scripts/train_SOccDPT.py:323-338,380-386)
 * with the prediction side of model/SOccDPT.py:264-290, and its gradient w.r.t. the network outputs.
 * inv [B,h,w] f32, seg [B,C,h,w] f32 in (0,1): the network outputs (soccdpt_network's inv256 / seg256);
 * y_disp [B,H,W] f32, mask_disp [B,H,W] u8, y_seg [B,C,H,W] f32, mask_seg [B,C,H,W] u8: targets at camera resolution.
 * out[0..2] = loss, loss_disp, loss_seg; out[3+2b], out[4+2b] = per-image scale, shift.  d_inv [B,h,w], d_seg [B,C,h,w].
 * scratch: soccdpt_loss_scratch_bytes(B,H,W,h,w) bytes, any contents. */
size_t soccdpt_loss_scratch_bytes(int B, int H, int W, int h, int w);
int soccdpt_training_loss(int B, int H, int W, int h, int w, int C, int compute_scale_and_shift, float alpha, float loss_depth_w,
                          float loss_seg_w, const float* inv, const float* seg, const float* y_disp, const uint8_t* mask_disp,
                          const float* y_seg, const uint8_t* mask_seg, float* out, float* d_inv, float* d_seg, void* scratch,
                          void* stream);

/* ---- ground-truth occupancy generator (datasets/bdd_helper.py:238-530 OccupancyProcessor) ----
 * disparity [B,H,W] f32, seg_class [B,H,W] i32 (class ids after rgb_seg_to_class) ->
 *   depth  [B,H,W] f32 or NULL        baseline*focal/disparity, upper half hidden, inf/nan -> 0          (:446-455)
 *   points [B,H*W,3] f64 or NULL      scaled, shifted, rotated camera points (what frame["points"] holds)  (:457-489)
 *   occ    [B,g0,g1,g2,C] u8          counts > point_count_threshold                                       (:288-347)
 * counts [B,g0,g1,g2,C] u32 is scratch (zeroed by the call, holds the per-voxel point counts afterwards).
 * intr = {fx, fy, cx, cy, baseline} (float64); rot27 = Ra^T, Rb^T, Rc^T row-major float64 as rotate_points builds them (:604-652);
 * occ_shape = float32(grid_size / scale) (:264-270).  Arithmetic follows numpy 2.x promotion (oracle/gt_occ_ref.c). */
int soccdpt_gt_occupancy(int B, int H, int W, int C, const double* intr, const double* pc_scale, const double* pc_shift,
                         const double* rot27, const float* occ_shape, const int* grid, float threshold, const float* disparity,
                         const int32_t* seg_class, float* depth, double* points, uint32_t* counts, uint8_t* occ, void* stream);

/* Fused MLP half-block of a Swin-V2 block (kernel-level entry for parity tests; the network uses it for the narrow stages):
 *   x_f32[m][:] += LayerNorm(fc2(GELU(fc1(x_op[m][:]))))      fc1: w1 [4C][C] + b1, fc2: w2 [C][4C] + b2, LayerNorm eps 1e-5
 * (timm Mlp + res-post-norm).  x_op [M][C] 16-bit operand copy of x (bf16 / fp16 by precision), w1 / w2 in the
 * same format; x_op_out (may alias x_op) receives the operand copy of the new x, halo (nullable) a zero-halo NHWC image of it
 * ([B][H+2][W+2][C], M = B*H*W).  C in {96, 128, 192, 256}. */
int soccdpt_op_mlp_ln(const void* x_op, float* x_f32, const void* w1, const float* b1, const void* w2, const float* b2, const float* ln_g,
                      const float* ln_b, void* x_op_out, void* halo, int precision, int M, int C, int H, int W, void* stream);

/* ---- input transform (model/loader.py:256-270 Compose([Resize, NormalizeImage, PrepareForNet])) ----
 * img [B,Hs,Ws,3] u8 (RGB frames as the datasets hand them over, datasets/bengaluru_driving_dataset.py:118-130) ->
 * out [B,3,Hd,Wd] f32 = (cv2.resize(img, (Wd,Hd), INTER_CUBIC) - mean) / std, channels first (model/transforms.py:178-251).
 * The resampling is OpenCV's 8-bit bicubic fixed-point path (oracle/input_transform_ref.py states it; cv2 itself is not in the
 * image, so this entry point is parity-unpinned).  Hd, Wd come from Resize.get_size (host side: soccdpt_amd/model/transforms.py).
 * mean, std: 3 host doubles each (numpy evaluates the normalisation in float64). */
int soccdpt_input_transform_u8(const uint8_t* img, int B, int Hs, int Ws, int Hd, int Wd, const double* mean, const double* stdv, float* out,
                               void* stream);

/* Fused multi-tensor Adam step, in place (torch.optim.Adam of scripts/train_SOccDPT.py:311-318; amsgrad=False):
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors DEVICE pointers (f32), sizes[i] elements each; step >= 1 is
 * the step count AFTER this update (bias corrections 1 - beta^step). */
int soccdpt_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const size_t* sizes, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int step, void* stream);

/* ---- in-network tile tuning (tools/autotune_network.py): a warm micro-benchmark mis-ranks tile configurations whose weights and
 * activations arrive cold in the real launch sequence, so candidates are timed inside the forward.
 * soccdpt_profile_sites(h, 1): soccdpt_profile_collect then reports every distinct igemm shape as "site<i>" instead of per tile
 * family; soccdpt_site_count / soccdpt_site_get list those shapes (K = taps * Cin) with the configuration the heuristic picked.
 * soccdpt_tune_set forces configuration `cfg` for one shape (cfg < 0 removes the override); soccdpt_tune_clear removes all. */
int soccdpt_profile_sites(void* handle, int on);
int soccdpt_site_count(void* handle);
int soccdpt_site_get(void* handle, int i, int* M, int* N, int* K, int* taps, int* cfg, int* launches);
int soccdpt_tune_set(void* handle, int M, int N, int K, int taps, int cfg);
int soccdpt_tune_clear(void* handle);

/* ---- kernel-level entry points (parity tests of the individual HIP kernels) ---- */

/* One implicit-GEMM launch: out[m][n] = epilogue(sum_k X[m][k] * Wt[n][k]) with bf16 operands and f32
 * accumulation; stands in for F.conv2d / F.linear of model/blocks.py:155-191,391-414 and timm's Linear layers.
 * taps == 1: X is [M][ldx] bf16.  taps == 9: X is a zero-haloed NHWC bf16 image [B][H+2][W+2][Cin], M = B*H*W,
 * Wt is [N][9*Cin] (tap-major).  v = acc + bias[n] + res1[m][n] + res2[m][n]; act: 0 none, 1 relu, 2 gelu(erf).
 * out_f32 gets v (or act(v) when act_on_f32), out_bf16 gets act(v) ([M][N] or a halo image), out_dot[m] =
 * relu(sum_n act(v)[n]*dot_w[n] + dot_b) (N <= 32).  Unused pointers are NULL. */
typedef struct soccdpt_igemm_args {
    const void* x;
    const void* wt;
    int32_t M, N, Cin, taps, ldx, H, W;
    const float* bias;
    const float* res1;
    const float* res2;
    int32_t act;
    float* out_f32;
    int32_t act_on_f32;
    void* out_bf16;
    int32_t out_halo;
    const float* dot_w;
    float dot_b;
    float* out_dot;
    int32_t tune; /* kernel configuration id, -1 = library heuristic (benchmarking) */
    int32_t precision; /* SOCCDPT_PREC_*: element type of x, wt and out_bf16 (bf16 / f32 / fp16) and the MFMA used */
    int32_t splitk;    /* > 1: slice K over this many workgroups per output tile (deterministic partial-tile exchange);
                          needs sk_part (splitk*M*N floats) and sk_count (ceil(M/32)*ceil(N/64) zero words, left zero) */
    float* sk_part;
    uint32_t* sk_count;
    size_t sk_part_floats; /* capacity of sk_part in floats  */
    size_t sk_count_words; /* capacity of sk_count in words */
    /* generalised addressing (the ViT-hybrid encoder's convolutions and read-out projection; zeros = the plain forms above):
     * conv mode reads input pixel (y*stride + ky - pad, x*stride + kx - pad) of an NHWC image [B][Hi + 2*in_halo][Wi + 2*in_halo][Cin];
     * conv_general == 0 keeps stride 1 / pad 1 / in_halo 1 / Hi = H / Wi = W.  gather1: taps == 1 with conv addressing (strided 1x1).
     * Row groups (plain mode): row m at x + (m / grp_rows) * grp_stride + grp_off + (m % grp_rows) * ldx; columns k >= seg2_k come from the
     * per-group row x + (m / grp_rows) * grp_stride + seg2_off + (k - seg2_k)  (cat(token, class token) of ProjectReadout,
     * model/backbones/utils.py:27-40). */
    int32_t conv_general, stride, pad, in_halo, Hi, Wi, gather1;
    int32_t grp_rows, grp_off, seg2_k, seg2_off;
    int64_t grp_stride;
    /* GroupNorm statistics of the raw output (timm GroupNormAct after every StdConv2dSame of the ResNetV2 stem / stages).  gn_stats NULL = off.  The
     * launch leaves per-tile partials gn_part[(M tile * (N / gn_cpg) + group) * 2] = {sum, sum of squares} (tile rows 32 / 64 / 128 by configuration;
     * size gn_part for 32: (M / 32) * (N / gn_cpg) * 2 floats); no bias / residual on such a launch.  gn_count NULL: that is all -- the form the forward
     * uses since round 5, its GroupNorm-apply kernel adds the partials up (soccdpt_op_gn_apply / soccdpt_op_gn_finish).  gn_count non-NULL (the form of
     * rounds 2-4; the words themselves are no longer read or written): a finish launch follows and gn_stats [M / gn_hw][N / gn_cpg][2] receives
     * {mean, 1/sqrt(var + 1e-5)} per (sample, group). */
    float* gn_stats;
    float* gn_part;
    uint32_t* gn_count;
    int32_t gn_cpg, gn_hw;
    size_t gn_part_floats, gn_count_words;
    uint64_t* stamps; /* diagnostics: 4 s_memrealtime stamps (100 MHz) per workgroup (entry, first k-tile landed, main loop done, stores
                         done); NULL = off (tools/igemm_stamps.py) */
    int32_t sk_defer; /* with splitk > 1: every split only stores its partial tile and a second launch sums the splits in order into out_f32 (no
                         bias / residual / activation / operand output); what the training step's weight-gradient GEMMs use with the 8-wave
                         128 x 128 tiles (tune 46 for bf16 / fp16 operands, 3 for f32 / f16x3) */
} soccdpt_igemm_args;
int soccdpt_op_igemm(const soccdpt_igemm_args* args, void* stream);

/* Kernel-level entries (tests): the reader side of the hybrid's GroupNorm -- timm GroupNormAct after every StdConv2dSame of the ResNetV2 stem / stages
 * (created by _make_pretrained_vitb_rn50_384, /root/reference/SOccDPT/model/backbones/vit.py:147-201; restated in oracle/soccdpt_ref.py rn_bottleneck).
 * soccdpt_op_gn_finish: per-tile partials [B * tps][groups][2] = {sum, sum of squares} (soccdpt_op_igemm with gn_count == NULL) -> stats [B][groups][2] =
 *   {mean, 1 / sqrt(var + eps)} (biased variance), hw pixels per sample, cpg channels per group.
 * soccdpt_op_gn_apply: y = GN(raw) [+ GN2(raw2) | + res], ReLU when relu != 0, for raw [M][C] f32 NHWC (M = B * hw pixels, image width w); outputs, each
 *   optional: out_f32 [M][C], out_op [M][C] and out_halo [B][h+2][w+2][C] (zero halo left untouched) in operand format out_format (SOCCDPT_PREC_BF16 /
 *   _F16 / _F32 / _F16X3).  Statistics: stats / stats2 as above, or -- part / part2 non-NULL, tps / tps2 tiles per sample -- added up from the partials
 *   by the kernel itself (short lists: per thread; up to 64 tiles: per workgroup; longer: a soccdpt_op_gn_finish launch first) and ALSO written to
 *   stats / stats2, which must then be writable.  groups = C / cpg must divide 256, C / 4 must divide 256, hw * C / 4 must be a multiple of 256. */
int soccdpt_op_gn_finish(const float* dev_part, float* dev_stats, int B, int tps, int groups, int hw, int cpg, float eps, void* stream);
int soccdpt_op_gn_apply(const float* dev_raw, float* dev_stats, const float* dev_part, int tps, const float* dev_gamma, const float* dev_beta,
                        const float* dev_raw2, float* dev_stats2, const float* dev_part2, int tps2, const float* dev_gamma2, const float* dev_beta2,
                        const float* dev_res, float* dev_out_f32, void* dev_out_op, void* dev_out_halo, int out_format, int relu, size_t M, int hw, int w,
                        int C, int cpg, float eps, void* stream);

/* Global softmax attention of one ViT block (timm vision_transformer.Attention between the qkv and proj Linear layers; created by
 * model/backbones/vit.py:248): qkv [B*N][3*heads*64] -> out [B*N][heads*64] = softmax(q k^T / 8) v per (sample, head); elements
 * bf16 / f32 / fp16 by `precision`.  N <= 608 (dpt_hybrid_384: N = 577). */
int soccdpt_op_vit_attention(const void* dev_qkv, void* dev_out, int precision, int B, int N, int heads, void* stream);

/* Weight-gradient GEMM from operands as stored (kernel-level entry, tests): out [Nout][taps * C] f32 = sum over k of A[k][n] * B[k + shift(tap)][c] with
 * A [K][ldA] (the output gradient: rows = tokens, or halo pixels of a 3x3 convolution) and B [K][ldB] (the layer input alike), elements bf16 / fp16 / x3 by
 * `precision` (SOCCDPT_PREC_BF16 / _F16 / _F16X3); taps 1 or 9, shift(tap) = (tap / 3 - 1) * rp + (tap % 3 - 1) rows (B must be readable and finite from row
 * -(rp + 1) to K + rp).  K % 64 == 0; taps == 9: Nout, C % 128 == 0; taps == 1: Nout, C % 32 == 0 and both operands readable 127 columns past their rows.
 * scratch: at least 64 * Nout * taps * C floats is always enough (fewer splits are used when less is given).  What the training step's 16-bit and x3 amp
 * modes run instead of transposing both operands (csrc/train_wgrad_tn.hip; autograd of nn.Linear / nn.Conv2d weights, scripts/train_SOccDPT.py:360-393). */
int soccdpt_op_wgrad_tn(const void* dev_a, long lda, const void* dev_b, long ldb, size_t K, int Nout, int C, int taps, int rp, int precision,
                        float* dev_scratch, size_t scratch_floats, float* dev_out, void* stream);

/* Swin-V2 cosine window attention of one block (timm WindowAttention + shift/partition/reverse):
 * qkv [B*res*res][3*heads*32] -> out [B*res*res][heads*32], elements bf16 / f32 / fp16 by `precision` (SOCCDPT_PREC_*).  cpb_table [(2ws-1)^2][heads] f32 is
 * 16*sigmoid(cpb_mlp(coords)); scale[heads] = exp(min(logit_scale, ln 100)); bias_scratch: heads*ceil(ws*ws/32)^2*1024
 * floats.  Window sizes 16 and 8 (single-pass softmax) and 24 and 12 (online softmax over key tiles) are instantiated. */
int soccdpt_op_window_attention(const void* dev_qkv, const float* dev_cpb_table, const float* dev_scale, void* dev_out,
                                float* dev_bias_scratch, int B, int res, int ws, int shift, int heads, int precision, void* stream);

/* The same block with the qkv projection INSIDE the kernel (csrc/attention_qkv.hip, round 6): replaces `qkv = F.linear(x, Wqkv, cat(q_bias, 0, v_bias))` +
 * WindowAttention (timm SwinTransformerV2Block._attn; call site model/backbones/swin2.py:25-27; HF modeling_swinv2.py:389-452) -- the q / k / v tensor never
 * exists.  dev_x [B*res*res][C] 16-bit block input (C = heads*32); dev_wqkv [3C][C]: 16-bit rows (SOCCDPT_PREC_BF16 / _F16) or x3 pairs (SOCCDPT_PREC_F16X2W);
 * dev_qkv_bias [3C] f32; the other arguments as above.  out_x3 != 0 (fp16 kernels): dev_out in the x3 operand format for an x3 proj GEMM.  16 x 16 and 8 x 8 windows.
 * dev_stamps: NULL, or 5 x (B * windows * heads) 64-bit words that receive per-workgroup s_memrealtime stamps (diagnostics: tools/wattn_qkv_stamps.py). */
int soccdpt_op_window_attention_qkv(const void* dev_x, const void* dev_wqkv, const float* dev_qkv_bias, const float* dev_cpb_table, const float* dev_scale,
                                    void* dev_out, float* dev_bias_scratch, int B, int res, int ws, int shift, int heads, int precision, int out_x3, void* dev_stamps, void* stream);

/* Winograd F(2x2, 3x3) form of nn.Conv2d(C, N, 3, padding=1) over a 16-bit zero-halo NHWC image (csrc/wino.hip, round 6; the RCU convolutions of
 * model/blocks.py:391-414).  soccdpt_op_wino_weights: dev_w [N][C][3][3] f32 (x dev_scale[n] when given) -> dev_u, 16 * N * C 16-bit elements laid out [C/32][16][N][32]
 * (U = G g G^T computed in f32, rounded once).  soccdpt_op_wino_conv: dev_x_halo [B][H+2][W+2][C]; H, W multiples of 16, C of 32, N of 64; epilogue in igemm's order:
 * + bias, + res1 [M][N] f32, + res2 [B][res2_h][res2_w][N] f32 sampled bilinearly (align_corners), ReLU on the operand copy (and on the f32 output when act_on_f32),
 * dev_out_f32 [M][N] and / or dev_out_op (16-bit, or x3 when out_x3; plain [M][N] or zero-halo [B][H+2][W+2][N]).  precision = SOCCDPT_PREC_F16 or _BF16. */
int soccdpt_op_wino_weights(const float* dev_w, const float* dev_scale, void* dev_u, int N, int C, int precision, void* stream);
int soccdpt_op_wino_conv(const void* dev_x_halo, const void* dev_u, int B, int H, int W, int C, int N, const float* dev_bias, const float* dev_res1, const float* dev_res2,
                         int res2_h, int res2_w, int relu, int act_on_f32, float* dev_out_f32, void* dev_out_op, int out_halo, int out_x3, int precision, void* dev_stamps /* NULL, or 8 words per workgroup: diagnostics */, void* stream);

/* ---- training step: replaces `masks_pred = net(images)` in train mode + `grad_scaler.scale(loss).backward()`
 * (scripts/train_SOccDPT.py:360-393) for the encoder + decoder + heads (model/SOccDPT.py:660-685, model/dpt.py:142-232).
 * SOCCDPT_PREC_F32 handles only (Swin-V2 and ViT-hybrid backbones); weights are read as bound (no soccdpt_prepare needed: they change every step).
 * soccdpt_bind_grad: `dev_grad` (same shape as the weight, f32) receives d loss / d weight -- WRITTEN, not accumulated -- on every
 * soccdpt_train_backward; NULL unbinds (the weight is frozen and its weight-gradient GEMM is skipped: model/loss.py:110-152).
 * soccdpt_train_forward: train-mode forward (seg head BatchNorm on batch statistics, running buffers bound as "seg_head.1.running_mean/var"
 * updated in place with momentum 0.1; Dropout(dropout_p) with a counter-based mask from `seed`); keeps every activation the backward needs
 * in the workspace.  soccdpt_train_backward: d_inv [B,S,S], d_seg [B,C,S,S] -> bound gradients; must follow a soccdpt_train_forward on the
 * same workspace and B. */
int soccdpt_bind_grad(void* handle, const char* key, float* dev_grad);
/* Mixed precision for the backward (the reference's `amp` sweep parameter, scripts/train_SOccDPT.py:96-121,360-366: fp16 autocast + GradScaler): the
 * gradient GEMMs (dgrad / wgrad of every Linear and 3x3 convolution) run with 16-bit MFMA operands and f32 accumulation; the train-mode forward,
 * every saved activation, weights and gradients stay f32.  mode 0: off (default); 1: bf16 operands (f32-sized exponent: no loss scaling needed);
 * 2: fp16 operands (11-bit significand) -- the caller scales d_inv / d_seg like GradScaler scales the loss (the gradients come out scaled by the
 * same factor; values beyond +-65504 saturate in the operand conversion instead of becoming inf). */
int soccdpt_train_set_amp(void* handle, int mode);
/* Stochastic depth of the Swin-V2 encoder in the train-mode forward: timm's SwinTransformerV2 is created with drop_path_rate = 0.1 by default and
 * model/backbones/swin2.py:15-30 does not override it, so under net.train() the reference drops each residual branch of block i per sample with
 * probability rate * i / (blocks - 1) and scales the kept ones by 1 / (1 - p).  0 (the library default) disables it; the masks come from
 * soccdpt_train_forward's seed (not torch's generator), and soccdpt_train_workspace_tensor("drop_path.<stage>.<block>") reads them back. */
int soccdpt_train_set_drop_path(void* handle, float rate);
/* GradScaler.unscale_ of the fp16 mode over n gradients: dev_grads[i] *= inv_scale; *dev_found_inf |= 1 when a result is not finite (the caller
 * zeroes the flag, skips its optimizer step when it is set and backs the scale off: scripts/train_SOccDPT.py:390-393). */
int soccdpt_train_unscale(float* dev_grads, size_t n, float inv_scale, int* dev_found_inf, void* stream);
size_t soccdpt_train_workspace_bytes(void* handle, int B);
int soccdpt_train_forward(void* handle, const float* dev_x, int B, float* dev_inv, float* dev_seg, void* dev_workspace, size_t workspace_bytes,
                          float dropout_p, uint32_t seed, void* stream);
int soccdpt_train_backward(void* handle, const float* dev_x, int B, const float* dev_d_inv, const float* dev_d_seg, void* dev_workspace,
                           size_t workspace_bytes, void* stream);
/* Test entry: the encoder's backward alone.  dev_d_feat[l] (l = 0..3, finest first): gradient w.r.t. the l-th hooked feature map the encoder
 * hands to scratch.layer<l+1>_rn, f32 [B * r_l^2][C_l] in NHWC order.  Must follow a soccdpt_train_forward on the same workspace.  The Swin-V2
 * encoder has no ReLU, so its gradients can be compared with autograd without the mask-flip floor of the whole network (tests/test_train_step_gpu.py). */
int soccdpt_train_backward_encoder(void* handle, int B, const float* const* dev_d_feat, void* dev_workspace, size_t workspace_bytes, void* stream);
/* Location of a saved activation / gradient inside the training workspace (tests, debugging): f32 [pixels][channels], NHWC order.
 * "seg_conv" (seg_head.0 output), "seg_act" (after BatchNorm + ReLU + Dropout), "seg_logits", "depth_conv0", "depth_conv2", "lrn_raw<l>"
 * (layer<l+1>_rn output), "fused_raw<l>" (RCU2 input of refinenet<l+1>, l < 3), "rcu2_out<l>", "fusion_out<l>" (out_conv output, before
 * the resize), and after a backward "d_path1", "d_feat<l>" (gradient w.r.t. the hooked encoder map l).  Non-zero for unknown names. */
int soccdpt_train_workspace_tensor(void* handle, int B, const char* name, size_t* byte_offset, size_t* elems);

/* Location of a named intermediate inside the workspace handed to soccdpt_network for batch B:
 * "feat0".."feat3" (hooked encoder maps, halo bf16), "path1" (halo bf16), "seg_logits" (seg head before up-sampling/activation, f32 [B,2G,2G,3]), "xf" (final stage tokens f32).
 * kind: 0 = f32 plain, 1 = bf16 plain, 2 = bf16 zero-halo NHWC, 3 = f32 zero-halo NHWC, 4 = fp16 plain, 5 = fp16 zero-halo NHWC, 7 = x3 zero-halo NHWC.
 * Returns non-zero for unknown names, and 2 for "seg_feat" (the seg head's conv + BN + ReLU map) when that map is not materialised: with 16-bit operands in
 * the "head" group the 1x1 classifier is part of the convolution's launch (SOCCDPT_SEG_DOT3_OFF=1 restores the separate map). */
int soccdpt_workspace_tensor(void* handle, int B, const char* name, size_t* byte_offset, size_t* elems, int* kind,
                             int* H, int* W, int* C);

#ifdef __cplusplus
}
#endif
#endif /* SOCCDPT_HIP_H */
