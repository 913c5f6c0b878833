// XCD-local persistent stage kernel (stage_xcd.hip): the launch chain of the deep Swin-V2 stages (tiny_256: stages 2 and 3 = 58 launches of 5-15 us,
// 6 % of the forward's FLOPs and 30 % of its time) as ONE launch.  At B = 8 every XCD (32 CUs, one private L2) owns one frame: the phases of that
// frame -- qkv GEMM, window attention, proj GEMM, LayerNorm, fc1, fc2, LayerNorm per block, the PatchMerging reduction between the stages -- run one
// after the other on that XCD's workgroups, separated by a barrier on a per-XCD counter.  No phase ever reads another XCD's data, so nothing has
// to leave the XCD's L2 and no agent-scope write-back / invalidate is needed (the reference's call site: timm SwinTransformerV2 stages behind
// /root/reference/SOccDPT/model/backbones/swin2.py:24-30, hooks swin_common.py:12-54).
//
// A phase is a list of independent work items (GEMM tiles, (head, query-half) attention items, groups of 8 LayerNorm rows) that the XCD's workgroups
// take round-robin by their rank; the item bodies are the device functions of the stand-alone kernels (igemm_kernel.h, attention_body.h, ln_body.h),
// so every output is bit-identical to the launch chain's.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "igemm.h"

namespace soccdpt {

enum XPhaseKind { XP_GEMM = 0, XP_ATTN = 1, XP_LN = 2 };

struct XPhase {
    int kind = 0;
    int cfg = 0;              // XP_GEMM: 0 = 64 x 64 tile (8 waves, 64-deep k-tiles), 1 = 32 x 64 tile (8 waves, 128-deep k-tiles)
    int fmt = 1;              // XP_GEMM: operand format code of the launch (0 bf16, 1 fp16, 3 x3); XP_ATTN / XP_LN: 16-bit flavour (0 bf16, 1 fp16)
    int items_per_frame = 0;  // independent work items of ONE frame
    int rows_per_frame = 0;   // tokens of one frame at this stage (M of the batch launch = B * rows_per_frame)
    // ---- XP_GEMM: the descriptor of the BATCH launch (M = B * rows_per_frame); frame f owns m-tiles [f * mt_per_frame, (f + 1) * mt_per_frame)
    IgemmDesc g;
    int nk = 0, kpt = 0, ntn = 0, mt_per_frame = 0;
    // ---- XP_ATTN: one window per frame (res == window size: stages 2 and 3 of both Swin-V2 models), heads x QS items
    const bf16_t* qkv = nullptr;
    const float* bias_acc = nullptr;
    const float* scale = nullptr;
    bf16_t* attn_out = nullptr;
    int res = 0, ws = 0, heads = 0, out_x3 = 0;
    // ---- XP_LN: x (+)= LN(y) row-wise, 8 rows per item; arguments of launch_ln_residual (elementwise.hip)
    const float* y = nullptr;
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    float* xf = nullptr;
    bf16_t* xb = nullptr;
    bf16_t* halo = nullptr;
    int C = 0, residual = 1, ln_res = 0, merge = 0, x3 = 0, x3h = 0, rows_total = 0;
};

// Per-handle synchronisation words (device memory, zero at rest): registration counters and one arrival counter per XCD, each on a line of its own.
struct XSync {
    unsigned reg_total, pad0[31];
    unsigned done_total, pad1[31];
    unsigned err, pad2[31];          // != 0: a bounded spin gave up (the launch's outputs are invalid); sticky until the host clears it
    struct PerXcd {
        unsigned reg, pad0[31];
        unsigned arrive, pad1[31];
    } x[8];
    // phase timeline (diagnostics): when stamp_on != 0 the rank-0 workgroup of XCD 0 records s_memrealtime (100 MHz) at the start of every phase,
    // after its own last work item and after the phase barrier: t[3 p + 0 .. 2] for the first frame it processes
    unsigned stamp_on, pad3[31];
    unsigned long long t[3 * 96];
};

bool stage_xcd_gemm_supported(const IgemmDesc& d, int fmt, int rows_per_frame, int* cfg_out);
// phases: device copy of `n` XPhase records; sync: device XSync (zeroed by the caller before the FIRST launch; every launch leaves it zero again).
int launch_stage_xcd(const XPhase* dev_phases, int n, XSync* dev_sync, int B, hipStream_t st, std::string& err);
// workgroups the kernel is launched with on the current device (CUs x co-resident workgroups per CU)
int stage_xcd_grid();

}  // namespace soccdpt
