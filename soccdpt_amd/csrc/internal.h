// Internal declarations shared by the translation units of libsoccdpt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/soccdpt_hip.h"
#include "launch.h"

namespace soccdpt {

// Swin-V2 geometry (timm swinv2_* as created by /root/reference/SOccDPT/model/backbones/swin2.py:15-30;
// hooks from /root/reference/SOccDPT/model/dpt.py:67-72).
struct Arch {
    int img = 256, patch = 4, embed = 96, window = 16;
    int depths[4] = {2, 2, 6, 2};
    int heads[4] = {3, 6, 12, 24};
    int pretrained_window[4] = {0, 0, 0, 0};
    int hooks[4] = {1, 1, 5, 1};
    // ViT-hybrid (vitb_rn50_384; /root/reference/SOccDPT/model/backbones/vit.py:147-258, model/blocks.py:103-112): ResNetV2 (3, 4, 9) stem +
    // 12 ViT-B blocks; the reassembled pyramid is [256, 512, 768, 768] channels at 1/4, 1/8, 1/16, 1/32 of the input
    bool hybrid = false;
    int vit_depth = 12, vit_heads = 12, vit_dim = 768, stem_ch = 64;
    int rn_layers[3] = {3, 4, 9};
    int vit_hooks[2] = {8, 11};
    int grid() const { return img / patch; }
    int dim(int s) const { return embed << s; }
    int res(int s) const { return grid() >> s; }
    // feature pyramid handed to scratch.layerN_rn (level l = 0 finest)
    int fdim(int l) const { return hybrid ? (l == 0 ? 256 : l == 1 ? 512 : 768) : dim(l); }
    int fres(int l) const { return hybrid ? (img / 4) >> l : res(l); }
    int out_res() const { return img; }   // network output resolution (4 * fres(0))
    int ws(int s) const { return res(s) < window ? res(s) : window; }
    int shift(int s, int j) const { return (j % 2 == 0 || res(s) <= window) ? 0 : window / 2; }
};

struct WeightSlot {
    std::string key;
    std::vector<int64_t> shape;
    const float* ptr = nullptr;
    float* grad = nullptr;   // gradient destination of the training step (soccdpt_bind_grad); nullptr = frozen
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
};

struct Prepared;  // model.cpp

// Per-kernel device times (soccdpt_profile_enable): while a ProfScope is open, every kernel launched on this thread carries a start / stop
// event pair bound to its own dispatch (launch.h); a scope's time is the sum of its kernels' begin -> end times.  Aggregated per family on collect.
struct ProfRec {
    const char* name;
    double flops, bytes;
    size_t first_pair, n_pairs;
};
struct Profiler : LaunchTimer {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;   // pairs: pool[2 i], pool[2 i + 1]
    size_t used = 0;                // events handed out
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pool.push_back(e);
        }
        return pool[used++];
    }
    void next_pair(hipEvent_t* e0, hipEvent_t* e1) override { *e0 = get(); *e1 = get(); }
    ~Profiler() override {
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};
struct ProfScope {
    Profiler* p;
    LaunchTimer* prev = nullptr;
    size_t rec = 0;
    ProfScope(Profiler& prof, const char* name, double flops, double bytes, hipStream_t) : p(prof.on ? &prof : nullptr) {
        if (!p) return;
        rec = p->recs.size();
        p->recs.push_back(ProfRec{name, flops, bytes, p->used / 2, 0});
        prev = launch_timer();
        launch_timer() = p;
    }
    ~ProfScope() {
        if (!p) return;
        p->recs[rec].n_pairs = p->used / 2 - p->recs[rec].first_pair;
        launch_timer() = prev;
    }
};

// One distinct GEMM / conv shape of the launch sequence (in-network tuning: soccdpt_profile_sites, tools/autotune_network.py)
struct SiteRec {
    int M, N, K, taps, cfg, count;
    char name[16];
};

struct Handle {
    soccdpt_config cfg;
    int device = 0;
    int img = 256;
    Arch arch;
    std::string err;
    bool prof_sites = false;                              // profile igemm launches per shape ("site000", ...) instead of per tile family
    std::vector<SiteRec> sites;                           // shapes seen while prof_sites was on, in first-launch order
    std::unordered_map<long long, int> tune_by_shape;     // shape key -> forced tile configuration (in-network tuning)
    bool fuse_qkv = true;                                 // Swin-V2: qkv projection inside the attention kernel where instantiated (attention_qkv.hip); SOCCDPT_FUSE_QKV=0 = the two-launch chain
    int fuse_qkv_mask = 3;                                // ... per stage (bit s); SOCCDPT_FUSE_QKV_STAGES=<mask>: measurement switch
    int mlp_fuse_max = 128;                               // widest stage whose MLP half-block runs as one fused launch (mlp_fused.hip; wider ones lose)
    // SOCCDPT_PREC_MIXED: operand format of every launch-site group (model.cpp: prec_groups), 1 = fp16, 3 = x3; groups absent from the map are fp16
    std::unordered_map<std::string, int> prec_map;
    int prec_source = -1;   // soccdpt_prec_map_source: -1 unknown (not prepared), 0 shipped map on its own weights, 1 calibrated, 2 edited, 3 shipped map on other weights
    unsigned long long calib_fp = 0;   // fingerprint (calibrate.h) of the weights the calibrated map (prec_source 1) was derived on
    bool calib_fp_valid = false;
    std::vector<WeightSlot> weights;
    std::unordered_map<std::string, int> index;
    size_t prepared_bytes = 0;
    Prepared* prep = nullptr;
    bool is_prepared = false;
    int launches = 0;
    Profiler prof;
    int n_streams = 1;                     // sub-batches run concurrently on n_streams streams (caller's + internal)
    std::vector<hipStream_t> sub_streams;  // n_streams - 1 internal non-blocking streams
    std::vector<hipEvent_t> join_events;   // [0] fork, [i] join of sub-stream i
    // hipGraph replay of soccdpt_network (soccdpt_set_graph): captured once per argument tuple
    bool use_graph = false;
    struct GraphKey {
        const void *x = nullptr, *inv = nullptr, *seg = nullptr, *ws = nullptr;
        int B = 0, streams = 0;
        bool operator==(const GraphKey& o) const { return x == o.x && inv == o.inv && seg == o.seg && ws == o.ws && B == o.B && streams == o.streams; }
    } graph_key;
    // zero-halo invariant (model_network): the (buffer, B, streams) tuple the workspace was last zero-filled for
    struct WsKey {
        const void* ws = nullptr;
        int B = 0, streams = 0;
        bool operator==(const WsKey& o) const { return ws == o.ws && B == o.B && streams == o.streams; }
    } ws_key;
    int ws_zero_fills = 0;
    // training step (train.cpp): what the last soccdpt_train_forward ran with, checked by soccdpt_train_backward
    struct TrainKey {
        const void* ws = nullptr;
        int B = 0;
        float dropout_p = 0.f;
        int amp = 0;              // amp mode and stochastic-depth rate LATCHED at the forward: the backward of that forward uses them even if the setters
        float drop_path = 0.f;    // were called in between (the reuse_xt staged operands and the DropPath scales belong to the forward)
    } train_key;
    float train_drop_path = 0.f;   // soccdpt_train_set_drop_path: timm drop_path_rate of the Swin-V2 encoder in train mode (0 = off)
    int train_amp = 0;        // soccdpt_train_set_amp: 16-bit MFMA operands for the gradient GEMMs (f32 accumulate, f32 weights / activations / gradients): 1 bf16, 2 fp16
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;  // capture/replay stream (the caller's may be the legacy null stream, which cannot capture)
    hipEvent_t graph_in = nullptr, graph_out = nullptr;
    int eager_calls = 0;  // calls seen with the current key before capture (the first one warms up lazy state)
    ~Handle();
};

// model.cpp
int model_init(Handle& h, std::string& err);
int model_bind(Handle& h, const char* key, const void* ptr, const int64_t* shape, int ndim, std::string& err);
size_t model_workspace_bytes(Handle& h, int B);
int model_prepare(Handle& h, void* prepared, size_t bytes, hipStream_t stream, std::string& err);
int model_network(Handle& h, const float* x, int B, float* inv256, float* seg256, void* ws, size_t ws_bytes, hipStream_t stream,
                  std::string& err);

std::vector<std::string> model_prec_groups(const Handle& h);   // every group name of this backbone, in launch order
void model_prec_default(Handle& h);                            // the shipped map of the backbone
int model_prec_set(Handle& h, const char* pattern, int fmt, std::string& err);   // -> groups changed, < 0 on error
bool model_prec_x2w_ok(const std::string& group);                                // may the group take x2w (fmt 4)?
int model_set_streams(Handle& h, int n, std::string& err);
void model_drop_graph(Handle& h);
int model_workspace_tensor(Handle& h, int B, const char* name, size_t* byte_offset, size_t* elems, int* kind, int* H, int* W, int* C);

// projection.hip
int launch_project(const soccdpt_config& cfg, const float* inv, const float* seg, int B, int in_h, int in_w, float* inv_up,
                   float* seg_up, float* points, uint32_t* occ_bits, int clear_bits, hipStream_t stream, std::string& err);
int launch_occ_expand(const soccdpt_config& cfg, const uint32_t* bits, int B, float* occ, hipStream_t stream, std::string& err);
int launch_occ_zero(const soccdpt_config& cfg, int B, float* occ, hipStream_t stream, std::string& err);
int launch_occ_set(const soccdpt_config& cfg, const uint32_t* bits, int B, float* occ, hipStream_t stream, std::string& err);
int launch_occ_or(const soccdpt_config& cfg, uint32_t* dst, const uint32_t* src, int nsets, hipStream_t stream, std::string& err);

}  // namespace soccdpt
