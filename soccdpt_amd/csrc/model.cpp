// Host-side orchestration of the SOccDPT_V3 network on gfx950 (weights table, prepare, launch sequence).
#include "internal.h"

namespace soccdpt {

struct Prepared {};
Handle::~Handle() { delete prep; }

static void add_w(Handle& h, const std::string& key, std::vector<int64_t> shape) {
    h.index[key] = (int)h.weights.size();
    h.weights.push_back(WeightSlot{key, std::move(shape), nullptr});
}

int model_init(Handle& h, std::string& err) {
    Arch a;
    if (h.cfg.backbone == SOCCDPT_BACKBONE_SWIN2T16_256) {
        // defaults
    } else if (h.cfg.backbone == SOCCDPT_BACKBONE_SWIN2B24_384) {
        a.img = 384; a.embed = 128; a.window = 24;
        int d[4] = {2, 2, 18, 2}, hd[4] = {4, 8, 16, 32}, pw[4] = {12, 12, 12, 6}, hk[4] = {1, 1, 17, 1};
        for (int i = 0; i < 4; ++i) { a.depths[i] = d[i]; a.heads[i] = hd[i]; a.pretrained_window[i] = pw[i]; a.hooks[i] = hk[i]; }
    } else {
        err = "soccdpt_create: backbone not implemented on the HIP path";
        return 1;
    }
    h.arch = a;
    h.img = a.img;
    const std::string E = "depth_net.pretrained.model.";
    const int64_t C0 = a.embed;
    add_w(h, E + "patch_embed.proj.weight", {C0, 3, a.patch, a.patch});
    add_w(h, E + "patch_embed.proj.bias", {C0});
    add_w(h, E + "patch_embed.norm.weight", {C0});
    add_w(h, E + "patch_embed.norm.bias", {C0});
    for (int s = 0; s < 4; ++s) {
        const int64_t C = a.dim(s), H = a.heads[s];
        for (int j = 0; j < a.depths[s]; ++j) {
            const std::string b = E + "layers." + std::to_string(s) + ".blocks." + std::to_string(j) + ".";
            add_w(h, b + "attn.logit_scale", {H, 1, 1});
            add_w(h, b + "attn.q_bias", {C});
            add_w(h, b + "attn.v_bias", {C});
            add_w(h, b + "attn.cpb_mlp.0.weight", {512, 2});
            add_w(h, b + "attn.cpb_mlp.0.bias", {512});
            add_w(h, b + "attn.cpb_mlp.2.weight", {H, 512});
            add_w(h, b + "attn.qkv.weight", {3 * C, C});
            add_w(h, b + "attn.proj.weight", {C, C});
            add_w(h, b + "attn.proj.bias", {C});
            add_w(h, b + "norm1.weight", {C});
            add_w(h, b + "norm1.bias", {C});
            add_w(h, b + "mlp.fc1.weight", {4 * C, C});
            add_w(h, b + "mlp.fc1.bias", {4 * C});
            add_w(h, b + "mlp.fc2.weight", {C, 4 * C});
            add_w(h, b + "mlp.fc2.bias", {C});
            add_w(h, b + "norm2.weight", {C});
            add_w(h, b + "norm2.bias", {C});
        }
        if (s < 3) {
            const std::string d = E + "layers." + std::to_string(s) + ".downsample.";
            add_w(h, d + "reduction.weight", {2 * C, 4 * C});
            add_w(h, d + "norm.weight", {2 * C});
            add_w(h, d + "norm.bias", {2 * C});
        }
    }
    const std::string S = "depth_net.scratch.";
    const int64_t F = h.cfg.features;
    for (int i = 0; i < 4; ++i) add_w(h, S + "layer" + std::to_string(i + 1) + "_rn.weight", {F, a.dim(i), 3, 3});
    for (int r = 1; r <= 4; ++r) {
        const std::string b = S + "refinenet" + std::to_string(r) + ".";
        add_w(h, b + "out_conv.weight", {F, F, 1, 1});
        add_w(h, b + "out_conv.bias", {F});
        for (int u = 1; u <= 2; ++u) {
            if (r == 4 && u == 1) continue;  // refinenet4 gets one input: its RCU1 never runs (model/dpt.py:163-165)
            for (int c = 1; c <= 2; ++c) {
                add_w(h, b + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(c) + ".weight", {F, F, 3, 3});
                add_w(h, b + "resConfUnit" + std::to_string(u) + ".conv" + std::to_string(c) + ".bias", {F});
            }
        }
    }
    add_w(h, S + "output_conv.0.weight", {F / 2, F, 3, 3});
    add_w(h, S + "output_conv.0.bias", {F / 2});
    add_w(h, S + "output_conv.2.weight", {32, F / 2, 3, 3});
    add_w(h, S + "output_conv.2.bias", {32});
    add_w(h, S + "output_conv.4.weight", {1, 32, 1, 1});
    add_w(h, S + "output_conv.4.bias", {1});
    add_w(h, "seg_head.0.weight", {F, F, 3, 3});
    add_w(h, "seg_head.1.weight", {F});
    add_w(h, "seg_head.1.bias", {F});
    add_w(h, "seg_head.1.running_mean", {F});
    add_w(h, "seg_head.1.running_var", {F});
    add_w(h, "seg_head.4.weight", {h.cfg.num_classes, F, 1, 1});
    add_w(h, "seg_head.4.bias", {h.cfg.num_classes});
    return 0;
}

int model_bind(Handle& h, const char* key, const void* ptr, const int64_t* shape, int ndim, std::string& err) {
    auto it = h.index.find(key);
    if (it == h.index.end()) {
        err = std::string("soccdpt_bind_weight: key not consumed by the HIP path: ") + key;
        return 2;
    }
    WeightSlot& w = h.weights[it->second];
    bool ok = (int)w.shape.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = (w.shape[i] == shape[i]);
    if (!ok) {
        err = std::string("soccdpt_bind_weight: shape mismatch for ") + key;
        return 3;
    }
    w.ptr = static_cast<const float*>(ptr);
    h.is_prepared = false;
    return 0;
}

size_t model_workspace_bytes(Handle& h, int B) { return (size_t)B * h.img * h.img * 4 * sizeof(float); }

int model_prepare(Handle&, void*, size_t, hipStream_t, std::string& err) {
    err = "soccdpt_prepare: network kernels not built yet";
    return 1;
}
int model_network(Handle&, const float*, int, float*, float*, void*, size_t, hipStream_t, std::string& err) {
    err = "soccdpt_network: network kernels not built yet";
    return 1;
}

}  // namespace soccdpt
