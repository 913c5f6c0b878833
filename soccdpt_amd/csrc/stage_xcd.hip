// XCD-local persistent stage kernel: see stage_xcd.h for what it replaces and why.
//
// Execution model
//  * grid = CUs x (co-resident workgroups per CU: 2 at 72 KB of LDS and <= 128 registers), 512 threads each.  Every workgroup reads the id of the
//    XCD it runs on (HW_REG_XCC_ID) and takes a rank among that XCD's workgroups from a per-XCD ticket counter; one registration counter over the
//    whole grid tells when every workgroup has a rank, after which the per-XCD workgroup count is final.  Nothing assumes a dispatch order.
//  * frame f belongs to XCD f % 8.  The XCD's workgroups walk the phase table; inside a phase they take the frame's work items round-robin by rank.
//  * between phases: every wave drains its stores (s_waitcnt vmcnt(0): they are in the XCD's L2 then), the workgroup meets, one lane adds to the
//    XCD's arrival counter and polls it with L1-bypassing loads until all of the XCD's workgroups have arrived.
//  * visibility without fences: every buffer a phase writes is a buffer of its own (model.cpp gives each (block, tensor) a fresh region), read by
//    the NEXT phase only, so no CU can hold a stale L1 line of it -- a line enters an L1 only after its final content is in L2; the residual stream
//    x is updated in place by LayerNorm phases only, and a row is always handled by the same workgroup (item -> rank is static).  Weights, biases
//    and bias tables are read-only for the whole launch.
//  * every spin is bounded: on a timeout the workgroup sets XSync::err and leaves (wrong results, no hang); the host checks the word.
#include "stage_xcd.h"

#include "attention_body.h"
#include "igemm_kernel.h"
#include "ln_body.h"

namespace soccdpt {
namespace {

constexpr int XT = 512;
using XCfgA = Cfg<64, 64, 64, 2, 4, 4>;     // 8 waves, 32 x 16 per wave, 4-stage ring: 64 KB
using XCfgB = Cfg<32, 64, 128, 2, 4, 3>;    // 8 waves, 16 x 16 per wave, 128-deep k-tiles, 3-stage ring: 72 KB
constexpr int XLDS = XCfgB::NS * XCfgB::STAGE;
static_assert(XCfgA::NS * XCfgA::STAGE <= XLDS && AttnGenCfg<16>::LDS <= XLDS && 4 * AttnCfg<8>::LDS <= XLDS, "LDS budget of the persistent kernel");
static_assert(XCfgA::THREADS == XT && XCfgB::THREADS == XT && AttnGenCfg<16>::THREADS == XT, "one workgroup size for every phase body");
constexpr unsigned kSpinLimit = 400000;     // ~0.3 s of polling before a spin gives up

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <class C, typename T>
__device__ __forceinline__ void gemm_items(const XPhase& ph, int f, int rank, int nranks, char* smem, int tid) {
    for (int it = rank; it < ph.items_per_frame; it += nranks) {
        const int lmt = it / ph.ntn, nt = it - lmt * ph.ntn;
        igemm_tile<C, T, false>(ph.g, ph.nk, ph.kpt, ph.ntn, (f * ph.mt_per_frame + lmt) * ph.ntn + nt, smem, tid);
        __syncthreads();   // the next tile's first LDS-DMA group overwrites ring slots the slowest wave may still be reading
    }
}

template <bool F16>
__device__ __forceinline__ void attn_items(const XPhase& ph, int f, int rank, int nranks, char* smem, int tid) {
    if (ph.ws == 16) {   // (head, query half) items on all 8 waves: the online-softmax kernel of the launch chain, query split 2
        for (int it = rank; it < ph.items_per_frame; it += nranks) {
            window_attention_flash_body<16, F16, 2>(ph.qkv, ph.bias_acc, ph.scale, ph.attn_out, ph.res, 0, ph.heads, ph.out_x3, f * ph.heads * 2 + it, tid, smem);
            __syncthreads();
        }
    } else {             // 8 x 8 windows: the launch chain's one-wave kernel, four heads per item on waves 0-3 (12.5 KB of LDS each)
        const int wave = tid >> 6, lane = tid & 63;
        for (int it = rank; it < ph.items_per_frame; it += nranks) {
            const int head = it * 4 + wave;
            if (wave < 4 && head < ph.heads)
                window_attention_body<8, F16, 1>(ph.qkv, ph.bias_acc, ph.scale, ph.attn_out, ph.res, 0, ph.heads, ph.out_x3, f * ph.heads + head, lane, smem + wave * AttnCfg<8>::LDS);
            else
                __syncthreads();   // the body's one workgroup barrier
            __syncthreads();
        }
    }
}

template <bool F16>
__device__ __forceinline__ void ln_items(const XPhase& ph, int f, int rank, int nranks, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int it = rank; it < ph.items_per_frame; it += nranks) {
        const int lrow = it * 8 + wave;
        if (lrow >= ph.rows_per_frame) continue;
        const int row = f * ph.rows_per_frame + lrow;
#define XLN_ARGS ph.y, ph.ln_g, ph.ln_b, ph.xf, ph.xb, ph.halo, nullptr, ph.rows_total, ph.C, ph.residual, ph.ln_res, ph.merge, ph.x3, ph.x3h, nullptr, 1, row, lane
        switch (ph.C) {   // the instantiations launch_ln_residual picks for these widths (elementwise.hip)
            case 384: ln_residual_row<8, F16>(XLN_ARGS); break;
            case 512: ln_residual_v4_row<2, F16>(XLN_ARGS); break;
            case 768: ln_residual_v4_row<3, F16>(XLN_ARGS); break;
            default: ln_residual_v4_row<4, F16>(XLN_ARGS); break;   // 1024
        }
#undef XLN_ARGS
    }
}

__global__ __launch_bounds__(XT, 4) void stage_xcd_kernel(const XPhase* __restrict__ phases, int nphases, XSync* sy, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_info[4];   // xcc, rank, workgroups on this XCD, alive
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned rank = __hip_atomic_fetch_add(&sy->x[xcc].reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&sy->reg_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0, alive = 1;
        while (ld_sc1(&sy->reg_total) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { alive = 0; break; }
        }
        s_info[0] = xcc; s_info[1] = rank; s_info[2] = ld_sc1(&sy->x[xcc].reg); s_info[3] = alive;
        if (!alive) __hip_atomic_store(&sy->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int xcc = (int)s_info[0], rank = (int)s_info[1], nranks = (int)s_info[2];
    if (!s_info[3]) return;
    unsigned epoch = 0;
    for (int f = xcc; f < B; f += 8) {
        for (int p = 0; p < nphases; ++p) {
            const XPhase& ph = phases[p];
            const bool stamp = sy->stamp_on && xcc == 0 && rank == 0 && f == xcc && tid == 0 && p < 96;
            if (stamp) sy->t[3 * p] = __builtin_amdgcn_s_memrealtime();
            // the thread index of THIS phase, opaque to the optimiser: without it the per-thread index arithmetic of every phase body is hoisted
            // out of the phase loop and kept live across all of them (128 registers + 312 bytes of scratch per lane)
            int ptid = tid;
            asm volatile("" : "+v"(ptid));
            if (ph.kind == XP_GEMM) {
                const int sel = ph.fmt * 2 + ph.cfg;
                switch (sel) {
                    case 0: gemm_items<XCfgA, bf16_t>(ph, f, rank, nranks, smem, ptid); break;
                    case 1: gemm_items<XCfgB, bf16_t>(ph, f, rank, nranks, smem, ptid); break;
                    case 2: gemm_items<XCfgA, f16_t>(ph, f, rank, nranks, smem, ptid); break;
                    case 3: gemm_items<XCfgB, f16_t>(ph, f, rank, nranks, smem, ptid); break;
                    case 6: gemm_items<XCfgA, x3_t>(ph, f, rank, nranks, smem, ptid); break;
                    default: gemm_items<XCfgB, x3_t>(ph, f, rank, nranks, smem, ptid); break;   // 7
                }
            } else if (ph.kind == XP_ATTN) {
                if (ph.fmt) attn_items<true>(ph, f, rank, nranks, smem, ptid); else attn_items<false>(ph, f, rank, nranks, smem, ptid);
            } else {
                if (ph.fmt) ln_items<true>(ph, f, rank, nranks, ptid); else ln_items<false>(ph, f, rank, nranks, ptid);
            }
            if (stamp) sy->t[3 * p + 1] = __builtin_amdgcn_s_memrealtime();
            // ---- XCD-local phase barrier ----
            ++epoch;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the XCD's L2
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&sy->x[xcc].arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = epoch * (unsigned)nranks;
                unsigned spins = 0;
                while (ld_sc1(&sy->x[xcc].arrive) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) {
                        s_info[3] = 0;
                        __hip_atomic_store(&sy->err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
            if (!s_info[3]) return;
            if (stamp) sy->t[3 * p + 2] = __builtin_amdgcn_s_memrealtime();
        }
    }
    // ---- leave the synchronisation words zero for the next launch: the last workgroup of the grid to get here clears them ----
    if (tid == 0) {
        const unsigned d = __hip_atomic_fetch_add(&sy->done_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            for (int x = 0; x < 8; ++x) {
                __hip_atomic_store(&sy->x[x].reg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sy->x[x].arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&sy->reg_total, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->done_total, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

bool stage_xcd_gemm_supported(const IgemmDesc& d, int fmt, int rows_per_frame, int* cfg_out) {
    if (d.taps != 1 || d.gather1 || d.grp_rows || d.seg2_k || d.wt_grp_rows || d.stamps || d.ln_g || d.gn_stats || d.out_dot || d.splitk > 1 || d.res2) return false;
    if (fmt != 0 && fmt != 1 && fmt != 3) return false;
    if (d.Cin % 128 != 0 || rows_per_frame % 64 != 0 || d.M % rows_per_frame != 0 || d.N % 64 != 0) return false;
    // the tile that needs the fewest (rounds over the XCD's ~64 workgroups) x (bytes staged per tile); ties go to the bigger tile
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long ia = (rows_per_frame / 64) * cdiv(d.N, 64), ib = (rows_per_frame / 32) * cdiv(d.N, 64);
    const long ca = cdiv(ia, 64) * (64 + 64), cb = cdiv(ib, 64) * (32 + 64);
    *cfg_out = ca <= cb ? 0 : 1;
    return true;
}

int stage_xcd_grid() {
    static std::atomic<int> cached[256];
    const int dev = PerDeviceOnce::dev();
    int g = cached[dev].load();
    if (g > 0) return g;
    int cus = 0, nb = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_xcd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, XLDS) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&stage_xcd_kernel), XT, XLDS) != hipSuccess || nb <= 0) return 0;
    g = cus * (nb > 2 ? 2 : nb);
    cached[dev].store(g);
    return g;
}

int launch_stage_xcd(const XPhase* dev_phases, int n, XSync* dev_sync, int B, hipStream_t st, std::string& err) {
    if (!dev_phases || n <= 0 || !dev_sync || B <= 0 || B % 8 != 0) { err = "stage_xcd: bad arguments (the batch must be a multiple of 8: one frame per XCD)"; return 1; }
    const int grid = stage_xcd_grid();
    if (grid <= 0) { err = "stage_xcd: the persistent kernel does not fit this device"; return 1; }
    SOCCDPT_LAUNCH(stage_xcd_kernel, dim3((unsigned)grid), dim3(XT), XLDS, st, dev_phases, n, dev_sync, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("stage_xcd launch: ") + hipGetErrorString(e); return 1; }
    return 0;
}

}  // namespace soccdpt
