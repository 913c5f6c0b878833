// extern "C" surface of libsoccdpt_hip.so (see include/soccdpt_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <cstdlib>
#include <string>

#include "internal.h"
#include "calibrate.h"
#include "train.h"
#include "kernels.h"

using namespace soccdpt;

static std::string g_create_error;

namespace soccdpt {
LaunchTimer*& launch_timer() {
    static thread_local LaunchTimer* t = nullptr;
    return t;
}
unsigned long long& launch_counter() {
    static thread_local unsigned long long n = 0;
    return n;
}
}  // namespace soccdpt

// algorithmic HBM bytes of the projection kernel: inputs once + every requested output once
static double project_bytes(const soccdpt_config& c, int B, int h, int w, const void* inv_up, const void* seg_up, const void* points) {
    const double px = (double)c.cam_width * c.cam_height * B;
    return (double)B * h * w * 4.0 * (1 + c.num_classes) + px * 4.0 * ((inv_up ? 1 : 0) + (seg_up ? c.num_classes : 0) + (points ? 3 : 0));
}
static double occ_cells(const soccdpt_config& c) { return (double)c.grid[0] * c.grid[1] * c.grid[2] * c.num_classes; }

static int fail(Handle* h, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return 1;
}

extern "C" {

int soccdpt_abi_version(void) { return SOCCDPT_ABI_VERSION; }
size_t soccdpt_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(soccdpt_config);
        case 1: return sizeof(soccdpt_igemm_args);
        case 2: return sizeof(soccdpt_kernel_stat);
        case 3: return sizeof(soccdpt_calib_report);
        case 4: return sizeof(soccdpt_calib_options);
        default: return 0;
    }
}

int soccdpt_prec_map_set(void* handle, const char* group, int fmt) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !group) return -1;
    if (h->cfg.precision != SOCCDPT_PREC_MIXED) { fail(h, "soccdpt_prec_map_set: the handle was not created with SOCCDPT_PREC_MIXED"); return -1; }
    if (fmt != SOCCDPT_PREC_F16 && fmt != SOCCDPT_PREC_F16X3 && fmt != SOCCDPT_PREC_F16X2W) { fail(h, "soccdpt_prec_map_set: fmt must be SOCCDPT_PREC_F16, SOCCDPT_PREC_F16X2W or SOCCDPT_PREC_F16X3"); return -1; }
    const int n = model_prec_set(*h, group, fmt == SOCCDPT_PREC_F16X3 ? 3 : (fmt == SOCCDPT_PREC_F16X2W ? 4 : 1), h->err);
    if (n < 0) return n;
    // formats decide where the zero borders of the 3x3 inputs lie and which weight copies exist: both caches are void
    h->ws_key = Handle::WsKey();
    h->is_prepared = false;
    h->prec_source = 2;   // edited by the caller
    model_drop_graph(*h);
    return n;
}
int soccdpt_prec_map_get(void* handle, char* buf, int buf_bytes) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return -1;
    std::string out;
    for (const auto& g : model_prec_groups(*h)) {
        auto it = h->prec_map.find(g);
        const int f = h->cfg.precision == SOCCDPT_PREC_MIXED ? (it == h->prec_map.end() ? SOCCDPT_PREC_F16 : it->second == 3 ? SOCCDPT_PREC_F16X3 : it->second == 4 ? SOCCDPT_PREC_F16X2W : SOCCDPT_PREC_F16)
                                                             : h->cfg.precision;
        if (!out.empty()) out += ' ';
        out += g + "=" + std::to_string(f);
    }
    if (buf && buf_bytes > 0) {
        const size_t n = out.size() < (size_t)buf_bytes - 1 ? out.size() : (size_t)buf_bytes - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int)out.size();
}

int soccdpt_create(const soccdpt_config* cfg, void** handle) {
    if (!cfg || !handle) return fail(nullptr, "soccdpt_create: null argument");
    if (cfg->abi_version != SOCCDPT_ABI_VERSION) return fail(nullptr, "soccdpt_create: abi_version mismatch");
    if (cfg->num_classes != 3) return fail(nullptr, "soccdpt_create: num_classes must be 3 (model/SOccDPT.py:347-349)");
    if (cfg->features != 256) return fail(nullptr, "soccdpt_create: features must be 256");
    if (cfg->cam_width <= 0 || cfg->cam_height <= 0) return fail(nullptr, "soccdpt_create: bad camera size");
    if (cfg->precision != SOCCDPT_PREC_BF16 && cfg->precision != SOCCDPT_PREC_F32 && cfg->precision != SOCCDPT_PREC_F16 && cfg->precision != SOCCDPT_PREC_F16X3 &&
        cfg->precision != SOCCDPT_PREC_MIXED)
        return fail(nullptr, "soccdpt_create: unknown precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, "soccdpt_create: no HIP device visible (the MI355X path has no CPU fallback)");
    Handle* h = new Handle();
    if (const char* e = getenv("SOCCDPT_MLP_FUSE_MAX")) h->mlp_fuse_max = atoi(e);   // measurement switch (0 = unfused everywhere)
    if (const char* e = getenv("SOCCDPT_FUSE_QKV")) h->fuse_qkv = atoi(e) != 0;         // measurement switch: 0 = qkv igemm + window_attention as in rounds 1-5
    if (const char* e = getenv("SOCCDPT_FUSE_QKV_STAGES")) h->fuse_qkv_mask = atoi(e);
    h->cfg = *cfg;
    (void)hipGetDevice(&h->device);
    std::string err;
    if (model_init(*h, err)) {
        delete h;
        return fail(nullptr, err);
    }
    *handle = h;
    return 0;
}

void soccdpt_destroy(void* handle) { delete static_cast<Handle*>(handle); }

const char* soccdpt_last_error(void* handle) {
    return handle ? static_cast<Handle*>(handle)->err.c_str() : g_create_error.c_str();
}

int soccdpt_bind_weight(void* handle, const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !key || !dev_ptr) return fail(h, "soccdpt_bind_weight: null argument");
    if (dtype != SOCCDPT_DTYPE_F32) return fail(h, std::string("soccdpt_bind_weight: only f32 tensors are accepted: ") + key);
    return model_bind(*h, key, dev_ptr, shape, ndim, h->err);
}

int soccdpt_num_weights(void* handle) { return (int)static_cast<Handle*>(handle)->weights.size(); }
const char* soccdpt_weight_key(void* handle, int index) {
    Handle* h = static_cast<Handle*>(handle);
    if (index < 0 || index >= (int)h->weights.size()) return nullptr;
    return h->weights[index].key.c_str();
}

size_t soccdpt_prepared_bytes(void* handle) { return static_cast<Handle*>(handle)->prepared_bytes; }
size_t soccdpt_workspace_bytes(void* handle, int B) { return model_workspace_bytes(*static_cast<Handle*>(handle), B); }
int soccdpt_workspace_invalidate(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    h->ws_key = Handle::WsKey();
    model_drop_graph(*h);
    return 0;
}
int soccdpt_workspace_zero_fills(void* handle) { return handle ? static_cast<Handle*>(handle)->ws_zero_fills : -1; }

int soccdpt_prepare(void* handle, void* dev_prepared, size_t prepared_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (model_prepare(*h, dev_prepared, prepared_bytes, (hipStream_t)stream, h->err)) return 1;
    if (h->cfg.precision == SOCCDPT_PREC_MIXED && h->prec_source == 1 && h->calib_fp_valid) {
        // A calibrated map is a statement about the weights it was derived on: other values under the same keys -- a rebind, or a
        // load_state_dict / optimizer step into the same storage -- void it.  Same four-tensor fingerprint as below; on a mismatch every group
        // goes back to x3 operands (source 3) until the caller calibrates again.
        unsigned long long* tmp = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(dev_prepared) + h->prepared_bytes - 8) & ~uintptr_t(7));
        unsigned long long fp = 0;
        const int rc = calib_fingerprint(*h, tmp, (hipStream_t)stream, &fp, h->err);
        if (rc < 0) return 1;
        if (rc > 0 || fp != h->calib_fp) {
            h->prec_source = 3;
            h->calib_fp_valid = false;
            for (const auto& g : model_prec_groups(*h)) h->prec_map[g] = 3;
            h->ws_key = Handle::WsKey();
            model_drop_graph(*h);
            if (model_prepare(*h, dev_prepared, prepared_bytes, (hipStream_t)stream, h->err)) return 1;
        }
        return 0;
    }
    if (h->cfg.precision == SOCCDPT_PREC_MIXED && h->prec_source != 1 && h->prec_source != 2) {
        // Is the shipped map running on the weights it was derived from?  (a fingerprint of four tensors; the arena's 256-byte tail is free.)
        // On any other weights its within-tolerance claim is unverified -- measured in round 5: a second synthetic draw leaves the class logits at
        // 1.4e-3 under the shipped map -- so until soccdpt_prec_calibrate has run every group takes x3 operands (f32-grade, about 1.7x the step).
        unsigned long long* tmp = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(dev_prepared) + h->prepared_bytes - 8) & ~uintptr_t(7));
        const int same = calib_weights_are_the_shipped_draw(*h, tmp, (hipStream_t)stream, h->err);
        if (same < 0) return 1;
        const int was = h->prec_source;
        h->prec_source = same ? 0 : 3;
        if (h->prec_source != was && (h->prec_source == 3 || was == 3)) {
            if (same) model_prec_default(*h);
            else for (const auto& g : model_prec_groups(*h)) h->prec_map[g] = 3;
            h->ws_key = Handle::WsKey();
            model_drop_graph(*h);
            if (model_prepare(*h, dev_prepared, prepared_bytes, (hipStream_t)stream, h->err)) return 1;
        }
    }
    return 0;
}

size_t soccdpt_prec_calibrate_scratch_bytes(void* handle, int B) { return handle ? calib_scratch_bytes(*static_cast<Handle*>(handle), B) : 0; }
int soccdpt_prec_calibrate(void* handle, const float* dev_x, int B, float budget, void* dev_prepared, size_t prepared_bytes, void* dev_workspace,
                           size_t workspace_bytes, void* dev_scratch, size_t scratch_bytes, soccdpt_calib_report* report, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    soccdpt_calib_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.struct_bytes = (int32_t)sizeof(opt);
    opt.budget = budget;
    return calib_run(*h, dev_x, B, opt, dev_prepared, prepared_bytes, dev_workspace, workspace_bytes, dev_scratch, scratch_bytes, report, (hipStream_t)stream, h->err);
}
int soccdpt_prec_calibrate_ex(void* handle, const float* dev_x, int B, const soccdpt_calib_options* options, void* dev_prepared, size_t prepared_bytes,
                              void* dev_workspace, size_t workspace_bytes, void* dev_scratch, size_t scratch_bytes, soccdpt_calib_report* report, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (!options || options->struct_bytes != (int32_t)sizeof(soccdpt_calib_options)) return fail(h, "soccdpt_prec_calibrate_ex: options missing or compiled against another header (struct_bytes)");
    return calib_run(*h, dev_x, B, *options, dev_prepared, prepared_bytes, dev_workspace, workspace_bytes, dev_scratch, scratch_bytes, report, (hipStream_t)stream, h->err);
}
int soccdpt_prec_map_source(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    return (h && h->cfg.precision == SOCCDPT_PREC_MIXED) ? h->prec_source : -1;
}

int soccdpt_network(void* handle, const float* dev_x, int B, float* dev_inv256, float* dev_seg256, void* dev_workspace,
                    size_t workspace_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    return model_network(*h, dev_x, B, dev_inv256, dev_seg256, dev_workspace, workspace_bytes, (hipStream_t)stream, h->err);
}

int soccdpt_bind_grad(void* handle, const char* key, float* dev_grad) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !key) return fail(h, "soccdpt_bind_grad: null argument");
    auto it = h->index.find(key);
    if (it == h->index.end()) return fail(h, std::string("soccdpt_bind_grad: unknown key: ") + key);
    h->weights[it->second].grad = dev_grad;
    return 0;
}
int soccdpt_train_set_amp(void* handle, int on) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (on < 0 || on > 3) return fail(h, "soccdpt_train_set_amp: mode must be 0 (off), 1 (bf16), 2 (fp16) or 3 (x3 split fp16)");
    h->train_amp = on;
    return 0;
}
int soccdpt_train_set_drop_path(void* handle, float rate) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (!(rate >= 0.f && rate < 1.f)) return fail(h, "soccdpt_train_set_drop_path: rate must be in [0, 1)");
    h->train_drop_path = rate;
    return 0;
}

int soccdpt_train_unscale(float* dev_grads, size_t n, float inv_scale, int* dev_found_inf, void* stream) {
    if (!dev_grads || !dev_found_inf) return fail(nullptr, "soccdpt_train_unscale: null argument");
    if (n == 0) return 0;
    std::string err;
    if (tr_unscale_check(dev_grads, n, inv_scale, dev_found_inf, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}
size_t soccdpt_train_workspace_bytes(void* handle, int B) { return handle && B > 0 ? train_workspace_bytes(*static_cast<Handle*>(handle), B) : 0; }
int soccdpt_train_backward_encoder(void* handle, int B, const float* const* dev_d_feat, void* dev_workspace, size_t workspace_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (h->train_key.ws != dev_workspace || h->train_key.B != B || !dev_workspace)
        return fail(h, "soccdpt_train_backward_encoder: no soccdpt_train_forward ran on this workspace with this batch size");
    const int amp_now = h->train_amp;
    const float dp_now = h->train_drop_path;
    h->train_amp = h->train_key.amp; h->train_drop_path = h->train_key.drop_path;
    const int rc = train_backward_encoder(*h, B, dev_d_feat, dev_workspace, workspace_bytes, (hipStream_t)stream, h->err);
    h->train_amp = amp_now; h->train_drop_path = dp_now;
    return rc;
}
int soccdpt_train_workspace_tensor(void* handle, int B, const char* name, size_t* byte_offset, size_t* elems) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !byte_offset || !elems) return 1;
    return train_workspace_tensor(*h, B, name, byte_offset, elems);
}
int soccdpt_train_forward(void* handle, const float* dev_x, int B, float* dev_inv, float* dev_seg, void* dev_workspace, size_t workspace_bytes,
                          float dropout_p, uint32_t seed, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (!dev_x || !dev_inv || !dev_seg) return fail(h, "soccdpt_train_forward: null argument");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(h, "soccdpt_train_forward: dropout_p must be in [0, 1)");
    h->train_key = Handle::TrainKey();
    if (train_forward(*h, dev_x, B, dev_inv, dev_seg, dev_workspace, workspace_bytes, dropout_p, seed, (hipStream_t)stream, h->err)) return 1;
    h->train_key.ws = dev_workspace; h->train_key.B = B; h->train_key.dropout_p = dropout_p;
    h->train_key.amp = h->train_amp; h->train_key.drop_path = h->train_drop_path;
    return 0;
}
int soccdpt_train_backward(void* handle, const float* dev_x, int B, const float* dev_d_inv, const float* dev_d_seg, void* dev_workspace,
                           size_t workspace_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (!dev_d_inv || !dev_d_seg) return fail(h, "soccdpt_train_backward: null argument");
    if (h->train_key.ws != dev_workspace || h->train_key.B != B || !dev_workspace)
        return fail(h, "soccdpt_train_backward: no soccdpt_train_forward ran on this workspace with this batch size");
    // the modes the forward ran with (a setter called between forward and backward takes effect at the next forward)
    const int amp_now = h->train_amp;
    const float dp_now = h->train_drop_path;
    h->train_amp = h->train_key.amp; h->train_drop_path = h->train_key.drop_path;
    const int rc = train_backward(*h, dev_x, B, dev_d_inv, dev_d_seg, dev_workspace, workspace_bytes, (hipStream_t)stream, h->err);
    h->train_amp = amp_now; h->train_drop_path = dp_now;
    return rc;
}

int soccdpt_project(void* handle, const float* dev_inv, const float* dev_seg, int B, int in_h, int in_w, float* dev_inv_up,
                    float* dev_seg_up, float* dev_points, uint32_t* dev_occ_bits, int clear_bits, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (!dev_inv || !dev_seg) return fail(h, "soccdpt_project: null input");
    ProfScope ps(h->prof, "project_voxelise", 0.0, project_bytes(h->cfg, B, in_h, in_w, dev_inv_up, dev_seg_up, dev_points), (hipStream_t)stream);
    return launch_project(h->cfg, dev_inv, dev_seg, B, in_h, in_w, dev_inv_up, dev_seg_up, dev_points, dev_occ_bits, clear_bits,
                          (hipStream_t)stream, h->err);
}

size_t soccdpt_occ_words(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    const size_t ncell = (size_t)h->cfg.grid[0] * h->cfg.grid[1] * h->cfg.grid[2] * h->cfg.num_classes;
    return (ncell + 31) / 32;
}

size_t soccdpt_project_backward_scratch_bytes(void* handle, int B, int in_h) {
    Handle* h = static_cast<Handle*>(handle);
    return (h && B > 0 && in_h > 0) ? upsample_bwd_scratch_bytes(h->cfg, B, in_h) : 0;
}

int soccdpt_project_backward(void* handle, const float* dev_inv_up, const float* dev_d_inv_up, const float* dev_d_seg_up, const float* dev_d_points,
                             int B, int in_h, int in_w, float* dev_d_inv, float* dev_d_seg, void* dev_scratch, size_t scratch_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (B <= 0 || in_h <= 0 || scratch_bytes < upsample_bwd_scratch_bytes(h->cfg, B, in_h)) return fail(h, "soccdpt_project_backward: scratch too small");
    if (launch_upsample_bwd(h->cfg, dev_inv_up, dev_d_inv_up, dev_d_seg_up, dev_d_points, B, in_h, in_w, dev_d_inv, dev_d_seg, dev_scratch,
                            (hipStream_t)stream, h->err))
        return 1;
    return 0;
}

int soccdpt_occ_or(void* handle, uint32_t* dev_dst_bits, const uint32_t* dev_src_bits, int n_sets, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (n_sets <= 0) return 0;
    return launch_occ_or(h->cfg, dev_dst_bits, dev_src_bits, n_sets, (hipStream_t)stream, h->err);
}

int soccdpt_occ_expand(void* handle, const uint32_t* dev_bits, int B, float* dev_occ, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (B <= 0) return fail(h, "soccdpt_occ_expand: empty batch");
    ProfScope ps(h->prof, "occ_expand", 0.0, occ_cells(h->cfg) * (4.0 * B + 0.125), (hipStream_t)stream);
    return launch_occ_expand(h->cfg, dev_bits, B, dev_occ, (hipStream_t)stream, h->err);
}

int soccdpt_occ_zero(void* handle, int B, float* dev_occ, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (B <= 0 || !dev_occ) return fail(h, "soccdpt_occ_zero: bad argument");
    ProfScope ps(h->prof, "occ_zero", 0.0, occ_cells(h->cfg) * 4.0 * B, (hipStream_t)stream);
    return launch_occ_zero(h->cfg, B, dev_occ, (hipStream_t)stream, h->err);
}

int soccdpt_occ_set(void* handle, const uint32_t* dev_bits, int B, float* dev_occ, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (B <= 0 || !dev_occ || !dev_bits) return fail(h, "soccdpt_occ_set: bad argument");
    ProfScope ps(h->prof, "occ_set", 0.0, occ_cells(h->cfg) * 0.125, (hipStream_t)stream);
    return launch_occ_set(h->cfg, dev_bits, B, dev_occ, (hipStream_t)stream, h->err);
}

int soccdpt_forward(void* handle, const float* dev_x, int B, float* dev_inv_up, float* dev_seg_up, float* dev_points,
                    float* dev_occ, uint32_t* dev_occ_bits, void* dev_workspace, size_t workspace_bytes, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (B <= 0) return fail(h, "soccdpt_forward: empty batch");
    const int S = h->img;
    // the network outputs live at the head of the workspace
    float* inv = static_cast<float*>(dev_workspace);
    float* seg = inv + (size_t)B * S * S;
    const size_t head = (size_t)B * S * S * (1 + h->cfg.num_classes) * sizeof(float);
    if (workspace_bytes < head) return fail(h, "soccdpt_forward: workspace too small");
    // model_network carves its scratch behind the same head, so the halo images sit at the same place for every entry point
    int rc = model_network(*h, dev_x, B, inv, seg, dev_workspace, workspace_bytes, (hipStream_t)stream, h->err);
    if (rc) return rc;
    const bool occ_on = h->cfg.compute_occ != 0;
    if (occ_on && !dev_occ_bits) return fail(h, "soccdpt_forward: compute_occ needs dev_occ_bits");
    {
        ProfScope ps(h->prof, "project_voxelise", 0.0, project_bytes(h->cfg, B, S, S, dev_inv_up, dev_seg_up, dev_points), (hipStream_t)stream);
        rc = launch_project(h->cfg, inv, seg, B, S, S, dev_inv_up, dev_seg_up, dev_points, occ_on ? dev_occ_bits : nullptr, 1,
                            (hipStream_t)stream, h->err);
    }
    if (rc) return rc;
    if (occ_on && dev_occ) {
        ProfScope ps(h->prof, "occ_expand", 0.0, occ_cells(h->cfg) * (4.0 * B + 0.125), (hipStream_t)stream);
        rc = launch_occ_expand(h->cfg, dev_occ_bits, B, dev_occ, (hipStream_t)stream, h->err);
    }
    return rc;
}

int soccdpt_last_launch_count(void* handle) { return static_cast<Handle*>(handle)->launches; }
unsigned long long soccdpt_launch_counter(void) { return soccdpt::launch_counter(); }

size_t soccdpt_metrics_scratch_bytes(int B, int C) { return metrics_scratch_bytes(B, C); }

int soccdpt_metrics_depth(const float* dev_pred, const float* dev_gt, const uint8_t* dev_mask, int B, size_t npix, float* dev_out,
                          void* dev_scratch, void* stream) {
    std::string err;
    if (!dev_pred || !dev_gt || !dev_mask || !dev_out || !dev_scratch) return fail(nullptr, "soccdpt_metrics_depth: null argument");
    if (launch_depth_metrics(dev_pred, dev_gt, dev_mask, B, npix, dev_out, dev_scratch, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_metrics_iou(const float* dev_pred, const float* dev_gt, int B, int C, size_t npix, float* dev_out, void* dev_scratch,
                        void* stream) {
    std::string err;
    if (!dev_pred || !dev_gt || !dev_out || !dev_scratch) return fail(nullptr, "soccdpt_metrics_iou: null argument");
    if (launch_iou_metrics(dev_pred, dev_gt, B, C, npix, dev_out, dev_scratch, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_set_streams(void* handle, int n) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (n > 1) {   // experimental: see the header
        const char* e = getenv("SOCCDPT_ALLOW_MULTISTREAM");
        if (!e || atoi(e) != 1) return fail(h, "soccdpt_set_streams: n > 1 is experimental (DESIGN.md section 4); set SOCCDPT_ALLOW_MULTISTREAM=1 to opt in");
    }
    h->ws_key = Handle::WsKey();
    return model_set_streams(*h, n, h->err);
}

int soccdpt_set_graph(void* handle, int on) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    if (on && !h->graph_stream) {
        if (hipStreamCreateWithFlags(&h->graph_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->graph_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->graph_out, hipEventDisableTiming) != hipSuccess)
            return fail(h, "soccdpt_set_graph: stream/event creation failed");
    }
    h->use_graph = on != 0;
    if (!h->use_graph) model_drop_graph(*h);
    return 0;
}

int soccdpt_profile_enable(void* handle, int on) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    h->prof.on = on != 0;
    h->prof.recs.clear();
    h->prof.used = 0;
    return 0;
}

int soccdpt_profile_collect(void* handle, soccdpt_kernel_stat* out, int max_entries, int* n_entries) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !out || !n_entries) return 1;
    int n = 0;
    for (const ProfRec& r : h->prof.recs) {
        float ms = 0.f;
        for (size_t i = r.first_pair; i < r.first_pair + r.n_pairs; ++i) {   // each pair is bound to one dispatch: stop - start = the kernel's own duration
            hipEvent_t e0 = h->prof.pool[2 * i], e1 = h->prof.pool[2 * i + 1];
            if (hipEventSynchronize(e1) != hipSuccess) return fail(h, "soccdpt_profile_collect: event sync failed");
            float t = 0.f;
            if (hipEventElapsedTime(&t, e0, e1) != hipSuccess) return fail(h, "soccdpt_profile_collect: elapsed failed");
            ms += t;
        }
        int k = 0;
        for (; k < n; ++k)
            if (!std::strcmp(out[k].name, r.name)) break;
        if (k == n) {
            if (n == max_entries) continue;
            std::memset(&out[n], 0, sizeof(out[n]));
            std::strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
            ++n;
        }
        out[k].launches += (int)r.n_pairs;
        out[k].ms += ms;
        out[k].flops += r.flops;
        out[k].bytes += r.bytes;
    }
    *n_entries = n;
    h->prof.recs.clear();
    h->prof.used = 0;
    return 0;
}

int soccdpt_profile_sites(void* handle, int on) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    h->prof_sites = on != 0;
    h->sites.clear();
    h->sites.reserve(1024);   // the profiler records keep pointers to SiteRec::name: no reallocation while profiling
    return 0;
}
int soccdpt_site_count(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    return h ? (int)h->sites.size() : 0;
}
int soccdpt_site_get(void* handle, int i, int* M, int* N, int* K, int* taps, int* cfg, int* launches) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || i < 0 || i >= (int)h->sites.size()) return 1;
    const SiteRec& r = h->sites[i];
    *M = r.M; *N = r.N; *K = r.K; *taps = r.taps; *cfg = r.cfg; *launches = r.count;
    return 0;
}
int soccdpt_tune_set(void* handle, int M, int N, int K, int taps, int cfg) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    const long long key = (((long long)M * 8192 + N) * 65536 + K) * 16 + taps;
    if (cfg < 0) h->tune_by_shape.erase(key);
    else h->tune_by_shape[key] = cfg;
    model_drop_graph(*h);
    return 0;
}
int soccdpt_tune_clear(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return 1;
    h->tune_by_shape.clear();
    model_drop_graph(*h);
    return 0;
}

int soccdpt_gt_occupancy(int B, int H, int W, int C, const double* intr, const double* pc_scale, const double* pc_shift,
                         const double* rot27, const float* occ_shape, const int* grid, float threshold, const float* disparity,
                         const int32_t* seg_class, float* depth, double* points, uint32_t* counts, uint8_t* occ, void* stream) {
    std::string err;
    if (launch_gt_occupancy(B, H, W, C, intr, pc_scale, pc_shift, rot27, occ_shape, grid, threshold, disparity, seg_class, depth, points, counts, occ,
                            (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_input_transform_u8(const uint8_t* img, int B, int Hs, int Ws, int Hd, int Wd, const double* mean, const double* stdv, float* out,
                               void* stream) {
    std::string err;
    if (launch_input_transform_u8(img, B, Hs, Ws, Hd, Wd, mean, stdv, out, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_op_mlp_ln(const void* x_op, float* x_f32, const void* w1, const float* b1, const void* w2, const float* b2, const float* ln_g,
                      const float* ln_b, void* x_op_out, void* halo, int precision, int M, int C, int H, int W, void* stream) {
    std::string err;
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F16) return fail(nullptr, "soccdpt_op_mlp_ln: 16-bit operand modes only");
    if (launch_mlp_ln(static_cast<const bf16_t*>(x_op), x_f32, static_cast<const bf16_t*>(w1), b1, static_cast<const bf16_t*>(w2), b2, ln_g, ln_b,
                      static_cast<bf16_t*>(x_op_out), static_cast<bf16_t*>(halo), precision == SOCCDPT_PREC_F16, M, C, H, W, 0, (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const size_t* sizes, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int step, void* stream) {
    std::string err;
    if (launch_adam(n_tensors, params, grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

size_t soccdpt_loss_scratch_bytes(int B, int H, int W, int h, int w) { return loss_scratch_bytes(B, H, W, h, w); }

int soccdpt_training_loss(int B, int H, int W, int h, int w, int C, int compute_scale_and_shift, float alpha, float loss_depth_w,
                          float loss_seg_w, const float* inv, const float* seg, const float* y_disp, const uint8_t* mask_disp,
                          const float* y_seg, const uint8_t* mask_seg, float* out, float* d_inv, float* d_seg, void* scratch,
                          void* stream) {
    std::string err;
    if (launch_training_loss(B, H, W, h, w, C, compute_scale_and_shift, alpha, loss_depth_w, loss_seg_w, inv, seg, y_disp, mask_disp, y_seg,
                             mask_seg, out, d_inv, d_seg, scratch, (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_op_igemm(const soccdpt_igemm_args* a, void* stream) {
    if (!a) return fail(nullptr, "soccdpt_op_igemm: null args");
    IgemmDesc d;
    d.X = a->x; d.Wt = a->wt;
    d.M = a->M; d.N = a->N; d.Cin = a->Cin; d.taps = a->taps; d.ldx = a->ldx; d.H = a->H; d.W = a->W;
    d.bias = a->bias; d.res1 = a->res1; d.res2 = a->res2; d.act = a->act; d.out_f32 = a->out_f32; d.act_on_f32 = a->act_on_f32;
    d.out_op = a->out_bf16; d.f32 = a->precision == SOCCDPT_PREC_F32; d.f16 = a->precision == SOCCDPT_PREC_F16; d.x3 = a->precision == SOCCDPT_PREC_F16X3; d.x2w = a->precision == SOCCDPT_PREC_F16X2W; if (d.x2w) d.f16 = 1; d.out_halo = a->out_halo; d.dot_w = a->dot_w; d.dot_b = a->dot_b; d.out_dot = a->out_dot; d.tune = a->tune;
    d.splitk = a->splitk > 1 ? a->splitk : 1; d.sk_part = a->sk_part; d.sk_count = a->sk_count;
    d.sk_part_floats = a->sk_part_floats; d.sk_count_words = a->sk_count_words; d.sk_defer = a->sk_defer;
    if (a->conv_general) { d.stride = a->stride; d.pad = a->pad; d.in_halo = a->in_halo; d.Hi = a->Hi; d.Wi = a->Wi; d.gather1 = a->gather1; }
    d.grp_rows = a->grp_rows; d.grp_off = a->grp_off; d.grp_stride = a->grp_stride; d.seg2_k = a->seg2_k; d.seg2_off = a->seg2_off;
    int gn_bm = 0;
    d.gn_stats = a->gn_stats; d.gn_part = a->gn_part; d.gn_bm_out = &gn_bm; d.gn_cpg = a->gn_cpg; d.gn_hw = a->gn_hw;
    d.gn_part_floats = a->gn_part_floats;
    d.stamps = reinterpret_cast<unsigned long long*>(a->stamps);
    std::string err;
    if (a->gn_stats && a->gn_count && (a->gn_hw <= 0 || a->gn_cpg <= 0 || a->M % a->gn_hw || a->N % a->gn_cpg))
        return fail(nullptr, "soccdpt_op_igemm: GroupNorm statistics requested with gn_hw / gn_cpg that do not divide M / N");
    if (launch_igemm(d, (hipStream_t)stream, err)) return fail(nullptr, err);
    if (a->gn_stats && a->gn_count && (gn_bm <= 0 || a->gn_hw % gn_bm))
        return fail(nullptr, "soccdpt_op_igemm: statistics requested but the chosen tile has no statistics epilogue (tune forces a tile without one)");
    // gn_count != NULL: the caller wants {mean, rstd} in gn_stats (the form of rounds 2-4, where the convolution's last workgroup finished them; the words of
    // gn_count are no longer touched): a finish launch behind the convolution.  gn_count == NULL: partials only, as the forward launches it.
    if (a->gn_stats && a->gn_count && launch_gn_finish(a->gn_part, a->gn_stats, a->M / a->gn_hw, a->gn_hw / gn_bm, a->N / a->gn_cpg, a->gn_hw, a->gn_cpg, 1e-5f, (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_op_wgrad_tn(const void* dev_a, long lda, const void* dev_b, long ldb, size_t K, int Nout, int C, int taps, int rp, int precision, float* dev_scratch,
                        size_t scratch_floats, float* dev_out, void* stream) {
    std::string err;
    if (!dev_a || !dev_b || !dev_scratch || !dev_out) return fail(nullptr, "soccdpt_op_wgrad_tn: null argument");
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F16 && precision != SOCCDPT_PREC_F16X3) return fail(nullptr, "soccdpt_op_wgrad_tn: bf16, fp16 or f16x3 operands");
    const int f16 = precision == SOCCDPT_PREC_F16X3 ? 3 : (precision == SOCCDPT_PREC_F16 ? 1 : 0);
    if (tr_wgrad_tn(static_cast<const uint16_t*>(dev_a), lda, static_cast<const uint16_t*>(dev_b), ldb, K, Nout, C, taps, rp, f16, dev_scratch, scratch_floats, dev_out,
                    (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_op_vit_attention(const void* dev_qkv, void* dev_out, int precision, int B, int N, int heads, void* stream) {
    std::string err;
    if (!dev_qkv || !dev_out) return fail(nullptr, "soccdpt_op_vit_attention: null argument");
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F32 && precision != SOCCDPT_PREC_F16 && precision != SOCCDPT_PREC_F16X3)
        return fail(nullptr, "soccdpt_op_vit_attention: unknown precision");
    if (launch_vit_attention(dev_qkv, dev_out, precision, B, N, heads, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_op_window_attention(const void* dev_qkv, const float* dev_cpb_table, const float* dev_scale, void* dev_out,
                                float* dev_bias_scratch, int B, int res, int ws, int shift, int heads, int precision, void* stream) {
    std::string err;
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F32 && precision != SOCCDPT_PREC_F16 && precision != SOCCDPT_PREC_F16X3)
        return fail(nullptr, "soccdpt_op_window_attention: unknown precision");
    if (launch_attn_bias(dev_cpb_table, dev_bias_scratch, ws, heads, (hipStream_t)stream, err)) return fail(nullptr, err);
    if (precision == SOCCDPT_PREC_F32 || precision == SOCCDPT_PREC_F16X3) {   // F16X3: f32 qkv in, x3 operand out
        if (launch_window_attention_f32(static_cast<const float*>(dev_qkv), dev_bias_scratch, dev_cpb_table, dev_scale, static_cast<float*>(dev_out), B,
                                        res, ws, shift, heads, (hipStream_t)stream, err, precision == SOCCDPT_PREC_F16X3 ? 1 : 0))
            return fail(nullptr, err);
        return 0;
    }
    if (launch_window_attention(static_cast<const bf16_t*>(dev_qkv), dev_bias_scratch, dev_scale, static_cast<bf16_t*>(dev_out),
                                precision == SOCCDPT_PREC_F16 ? 1 : 0, B, res, ws, shift, heads, (hipStream_t)stream, err))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_op_window_attention_qkv(const void* dev_x, const void* dev_wqkv, const float* dev_qkv_bias, const float* dev_cpb_table, const float* dev_scale,
                                    void* dev_out, float* dev_bias_scratch, int B, int res, int ws, int shift, int heads, int precision, int out_x3, void* dev_stamps, void* stream) {
    std::string err;
    if (!dev_x || !dev_wqkv || !dev_qkv_bias || !dev_cpb_table || !dev_scale || !dev_out || !dev_bias_scratch) return fail(nullptr, "soccdpt_op_window_attention_qkv: null argument");
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F16 && precision != SOCCDPT_PREC_F16X2W)
        return fail(nullptr, "soccdpt_op_window_attention_qkv: precision is SOCCDPT_PREC_BF16, _F16 or _F16X2W (16-bit activations)");
    if (!window_attention_qkv_supported(ws, heads * 32, precision == SOCCDPT_PREC_F16X2W)) return fail(nullptr, "soccdpt_op_window_attention_qkv: window size / width not instantiated (16 x 16 and 8 x 8 windows)");
    if (launch_attn_bias(dev_cpb_table, dev_bias_scratch, ws, heads, (hipStream_t)stream, err)) return fail(nullptr, err);
    if (launch_window_attention_qkv(static_cast<const bf16_t*>(dev_x), dev_wqkv, dev_qkv_bias, dev_bias_scratch, dev_scale, static_cast<bf16_t*>(dev_out),
                                    precision != SOCCDPT_PREC_BF16 ? 1 : 0, precision == SOCCDPT_PREC_F16X2W ? 1 : 0, B, res, ws, shift, heads, (hipStream_t)stream, err, out_x3, static_cast<unsigned long long*>(dev_stamps)))
        return fail(nullptr, err);
    return 0;
}

int soccdpt_op_wino_weights(const float* dev_w, const float* dev_scale, void* dev_u, int N, int C, int precision, void* stream) {
    std::string err;
    if (!dev_w || !dev_u || (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F16)) return fail(nullptr, "soccdpt_op_wino_weights: null argument or precision not BF16 / F16");
    if (launch_wino_weights(dev_w, dev_scale, dev_u, precision == SOCCDPT_PREC_F16 ? 1 : 0, N, C, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}
int soccdpt_op_wino_conv(const void* dev_x_halo, const void* dev_u, int B, int H, int W, int C, int N, const float* dev_bias, const float* dev_res1, const float* dev_res2,
                         int res2_h, int res2_w, int relu, int act_on_f32, float* dev_out_f32, void* dev_out_op, int out_halo, int out_x3, int precision, void* dev_stamps, void* stream) {
    std::string err;
    if (precision != SOCCDPT_PREC_BF16 && precision != SOCCDPT_PREC_F16) return fail(nullptr, "soccdpt_op_wino_conv: precision is SOCCDPT_PREC_BF16 or _F16");
    if (dev_res2 && (res2_h <= 0 || res2_w <= 0)) return fail(nullptr, "soccdpt_op_wino_conv: dev_res2 needs its source size (res2_h, res2_w > 0)");
    if (B <= 0) return fail(nullptr, "soccdpt_op_wino_conv: B must be positive");
    WinoArgs a;
    a.X = dev_x_halo; a.U = dev_u; a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.bias = dev_bias; a.res1 = dev_res1; a.res2 = dev_res2; a.res2_h = res2_h; a.res2_w = res2_w;
    a.act = relu ? ACT_RELU : ACT_NONE; a.act_on_f32 = act_on_f32; a.out_f32 = dev_out_f32; a.out_op = dev_out_op; a.out_halo = out_halo; a.out_x3 = out_x3;
    a.hf = precision == SOCCDPT_PREC_F16 ? 1 : 0;
    a.stamps = static_cast<unsigned long long*>(dev_stamps);
    if (launch_wino_conv(a, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_op_gn_finish(const float* dev_part, float* dev_stats, int B, int tps, int groups, int hw, int cpg, float eps, void* stream) {
    std::string err;
    if (launch_gn_finish(dev_part, dev_stats, B, tps, groups, hw, cpg, eps, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_op_gn_apply(const float* dev_raw, float* dev_stats, const float* dev_part, int tps, const float* dev_gamma, const float* dev_beta,
                        const float* dev_raw2, float* dev_stats2, const float* dev_part2, int tps2, const float* dev_gamma2, const float* dev_beta2,
                        const float* dev_res, float* dev_out_f32, void* dev_out_op, void* dev_out_halo, int out_format, int relu, size_t M, int hw, int w,
                        int C, int cpg, float eps, void* stream) {
    std::string err;
    if (!dev_raw || !dev_stats || !dev_gamma || !dev_beta || hw <= 0 || w <= 0 || hw % w) return fail(nullptr, "soccdpt_op_gn_apply: bad arguments");
    const int om = out_format == SOCCDPT_PREC_BF16 ? 0 : (out_format == SOCCDPT_PREC_F16 ? 1 : (out_format == SOCCDPT_PREC_F32 ? 2 : (out_format == SOCCDPT_PREC_F16X3 ? 3 : -1)));
    if (om < 0) return fail(nullptr, "soccdpt_op_gn_apply: out_format is SOCCDPT_PREC_BF16, _F16, _F32 or _F16X3");
    GnApplyArgs g;
    g.raw = dev_raw; g.stats = dev_stats; g.part = dev_part; g.tps = tps; g.gamma = dev_gamma; g.beta = dev_beta;
    g.raw2 = dev_raw2; g.stats2 = dev_stats2; g.part2 = dev_part2; g.tps2 = tps2; g.gamma2 = dev_gamma2; g.beta2 = dev_beta2; g.res = dev_res;
    g.out_f32 = dev_out_f32; g.out_op = dev_out_op; g.out_halo = dev_out_halo; g.relu = relu; g.M = M; g.HW = hw; g.W = w; g.C = C; g.cpg = cpg; g.eps = eps;
    if (launch_gn_apply(g, om, (hipStream_t)stream, err)) return fail(nullptr, err);
    return 0;
}

int soccdpt_workspace_tensor(void* handle, int B, const char* name, size_t* byte_offset, size_t* elems, int* kind, int* H, int* W,
                             int* C) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !name) return 1;
    return model_workspace_tensor(*h, B, name, byte_offset, elems, kind, H, W, C);
}

}  // extern "C"
