// Weight-gradient GEMM that takes both operands AS STORED (rows = the reduction index), 16-bit amp modes of the training step.
//
//   dW[n][t * C + c] = sum_k dY[k][n] * X[k + shift(t)][c]        k = tokens (Linear) or halo pixels (3x3 convolution, t = tap, shift = (ky-1) rp + (kx-1))
//
// The igemm (igemm.hip) wants both operands contiguous along the reduction index, so round 2 / early round 3 transposed dY and X (and, for the
// convolutions, wrote shifted copies of the transposed halo image) before every weight-gradient launch: 206 transpose launches, 1.7 ms of an 18.5 ms
// bf16-amp step.  Here the tiles go to LDS in their stored layout -- [64 reduction rows][128 columns], 256-byte rows, by global_load_lds_dwordx4 -- and
// the MFMA fragments are read COLUMN-wise with gfx950's transposing LDS read (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of
// 16-bit elements, delivered column-major; lane 4q+p addresses row q, columns 4p..4p+3, lane i receives column i; semantics pinned by
// tests/tools/tr_read_probe.hip).  Two such reads are one v_mfma_f32_16x16x32 operand (8 consecutive k of one column).  The 16-byte chunks of a
// row are XOR-swizzled by ((row & 3) << 2) | ((row >> 2) & 3) on the DMA's source side (the LDS image of a DMA is lane-linear), which makes the
// transposed reads conflict-free (cdna_hip_programming.md T10, image (b)).
// A 3x3 tap is a ROW offset of the X operand, so the halo image is used as it is (no shifted copies, no alignment games), and dY in halo order is
// the buffer the dgrad launch already staged.
// Tile 128 (n) x 128 (columns) x 64 (k), 8 waves as 2 x 4, each 64 n x 32 columns; two-slot LDS ring (64 KB: two workgroups per CU); split-K with
// the deferred reduction of igemm.h (every split stores its partial tile, sk_reduce sums them in order): deterministic.
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <string>

#include "half16.h"
#include "launch.h"
#include "train.h"

namespace soccdpt {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TnArgs {
    const uint16_t* A;   // dY  [K][ldA]   (16-bit)
    const uint16_t* B;   // X   [K + margins][ldB], B points at row 0 of the un-shifted view
    long ldA, ldB;
    int K, Nout, Ncols, C, taps, rp, splits;
    float* part;         // [splits][Nout][Ncols]
    float* bias_part;    // nullable: [splits][Nout] partial column sums of dY (db = dY^T 1): one more MFMA per fragment against a fragment of ones in the
                         // workgroups of the first column tile, instead of the two colsum launches per layer that re-read dY in f32 (round 5)
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool F16>
__global__ __launch_bounds__(512) void wgrad_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 64 * 256, STAGE = 2 * TILE;   // bytes: A tile + B tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = (a.Ncols + 127) / 128;
    int bid = blockIdx.x;
    const int split = bid % a.splits;
    bid /= a.splits;
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int n0 = mt * 128, col0 = nt * 128;
    const int tap = a.taps > 1 ? col0 / a.C : 0, cc0 = a.taps > 1 ? col0 - tap * a.C : col0;
    const long shift = a.taps > 1 ? (long)(tap / 3 - 1) * a.rp + (tap % 3 - 1) : 0;
    const int nk = a.K / 64;
    const int kt0 = (int)((long)split * nk / a.splits), kt1 = (int)((long)(split + 1) * nk / a.splits);

    // ---- staging: per k-tile 16 wave-DMAs of 4 rows for each operand; wave w issues row blocks 2w and 2w + 1 ----
    const uint16_t* ga[2];
    const uint16_t* gb[2];
    uint32_t lds_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rb = wave * 2 + i, row = 4 * rb + (lane >> 4), pch = lane & 15, ch = pch ^ swz(row);
        ga[i] = a.A + (long)row * a.ldA + n0 + 8 * ch;
        gb[i] = a.B + ((long)row + shift) * a.ldB + cc0 + 8 * ch;
        lds_off[i] = (uint32_t)(rb * 1024);
    }
    auto stage = [&](int kt, int slot) {
        char* sb = smem + slot * STAGE;
        const long ka = (long)kt * 64 * a.ldA, kb = (long)kt * 64 * a.ldB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + ka),
                                             (__attribute__((address_space(3))) void*)(sb + lds_off[i]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + kb),
                                             (__attribute__((address_space(3))) void*)(sb + TILE + lds_off[i]), 16, 0, 0);
        }
    };

    // ---- transposed fragment reads: lane = 16 g + 4 q + p ----
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // byte offset inside a tile of the block (rows 32 ks + 8 g + 4 h2 + q, 16 columns starting at 16-column group cg): row * 256 + 16 * (ch ^ swz(row)) + 8 * (p & 1)
    auto frag = [&](const char* tile, int ks, int cg) -> h16x8 {
        h16x8 f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int row = 32 * ks + 8 * g + 4 * h2 + q;
            const int ch = 2 * cg + (p >> 1);
            const char* ptr = tile + row * 256 + 16 * (ch ^ swz(row)) + 8 * (p & 1);
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
            f[4 * h2] = v[0]; f[4 * h2 + 1] = v[1]; f[4 * h2 + 2] = v[2]; f[4 * h2 + 3] = v[3];
        }
        return f;
    };

    f32x4 acc[2][4];   // [it: 16 columns][jt: 16 n]
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of dY for the bias gradient: D = ones^T dY, every row of D the same sum; the wn == 0 waves of the first column tile carry it
    const bool do_b = a.bias_part != nullptr && nt == 0 && wn == 0;
    f32x4 accb[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) accb[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = F16 ? (short)0x3C00 : (short)0x3F80;

    if (kt0 < kt1) stage(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int slot = (kt - kt0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt has landed
        __builtin_amdgcn_s_barrier();                        // ... everybody's has, and everybody is done reading the other slot
        if (kt + 1 < kt1) stage(kt + 1, slot ^ 1);
        const char* ta = smem + slot * STAGE;
        const char* tb = ta + TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 fa[2], fb[4];
#pragma unroll
            for (int it = 0; it < 2; ++it) fa[it] = frag(tb, ks, wn * 2 + it);       // MFMA A operand: X columns (i = column)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) fb[jt] = frag(ta, ks, wm * 4 + jt);       // MFMA B operand: dY columns (j = n)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[it][jt] = mfma_16x16x32<F16>(fa[it], fb[jt], acc[it][jt]);
            if (do_b) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) accb[jt] = mfma_16x16x32<F16>(ones, fb[jt], accb[jt]);
            }
        }
    }
    if (do_b && lane < 16) {   // row i = 0 of D (lane quarter g = 0, register 0): n = lane & 15
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int n = n0 + wm * 64 + 16 * jt + lane;
            if (n < a.Nout) a.bias_part[(size_t)split * a.Nout + n] = accb[jt][0];
        }
    }
    // ---- partial tile out: lane holds D[i = 4 g + r][j = lane & 15] -> 4 consecutive columns of one n ----
    float* mine = a.part + (size_t)split * a.Nout * a.Ncols;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int n = n0 + wm * 64 + 16 * jt + (lane & 15);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int col = col0 + wn * 32 + 16 * it + 4 * g;
            // partial edge tiles (taps == 1 only): the columns past Nout / Ncols were computed from whatever follows the rows in memory and are dropped here
            // (a column of the product depends on its own operand column only)
            if (n < a.Nout && col < a.Ncols)
                *reinterpret_cast<float4*>(mine + (size_t)n * a.Ncols + col) = make_float4(acc[it][jt][0], acc[it][jt][1], acc[it][jt][2], acc[it][jt][3]);
        }
    }
}


// ---- x3 (split-fp16, half16.h) operands: 4 bytes per element, 8-element units of [hi x 8][lo x 8] (order swapped in odd units).  Tile 128 x 128 x 32:
// 512-byte LDS rows, 32 reduction rows per k-tile (one k-step of the fp16 MFMA), the 16-byte chunks XOR-swizzled in their low four index bits.  A 16-column
// fragment group spans units 2cg (hi at chunk 4cg, lo at 4cg + 1) and 2cg + 1 (lo at 4cg + 2, hi at 4cg + 3); three MFMAs per fragment pair, two
// accumulator sets folded at the end (igemm.hip X3).
__global__ __launch_bounds__(512) void wgrad_tn_x3_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 32 * 512, STAGE = 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = (a.Ncols + 127) / 128;
    int bid = blockIdx.x;
    const int split = bid % a.splits;
    bid /= a.splits;
    const int nt = bid % ntiles, mt = bid / ntiles;
    const int n0 = mt * 128, col0 = nt * 128;
    const int tap = a.taps > 1 ? col0 / a.C : 0, cc0 = a.taps > 1 ? col0 - tap * a.C : col0;
    const long shift = a.taps > 1 ? (long)(tap / 3 - 1) * a.rp + (tap % 3 - 1) : 0;
    const int nk = a.K / 32;
    const int kt0 = (int)((long)split * nk / a.splits), kt1 = (int)((long)(split + 1) * nk / a.splits);
    const char* Ab = reinterpret_cast<const char*>(a.A);
    const char* Bb = reinterpret_cast<const char*>(a.B);

    const char* ga[2];
    const char* gb[2];
    uint32_t lds_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rb = wave * 2 + i, row = 2 * rb + (lane >> 5), pch = lane & 31, ch = (pch & 16) | ((pch & 15) ^ swz(row));
        ga[i] = Ab + ((long)row * a.ldA + n0) * 4 + 16 * ch;
        gb[i] = Bb + (((long)row + shift) * a.ldB + cc0) * 4 + 16 * ch;
        lds_off[i] = (uint32_t)(rb * 1024);
    }
    auto stage = [&](int kt, int slot) {
        char* sb = smem + slot * STAGE;
        const long ka = (long)kt * 32 * a.ldA * 4, kb = (long)kt * 32 * a.ldB * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + ka),
                                             (__attribute__((address_space(3))) void*)(sb + lds_off[i]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + kb),
                                             (__attribute__((address_space(3))) void*)(sb + TILE + lds_off[i]), 16, 0, 0);
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // hi (lo == 0) or lo (lo == 1) fragment of 16-column group cg: rows 8 g + 4 h2 + q
    auto frag = [&](const char* tile, int cg, int lo) -> h16x8 {
        h16x8 f;
        const int odd = p >> 1;                                  // unit 2 cg + odd
        const int ch = 4 * cg + (odd ? (lo ? 2 : 3) : (lo ? 1 : 0));
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int row = 8 * g + 4 * h2 + q;
            const char* ptr = tile + row * 512 + 16 * ((ch & 16) | ((ch & 15) ^ swz(row))) + 8 * (p & 1);
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
            f[4 * h2] = v[0]; f[4 * h2 + 1] = v[1]; f[4 * h2 + 2] = v[2]; f[4 * h2 + 3] = v[3];
        }
        return f;
    };
    f32x4 acc[2][4], accx[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) { acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const bool do_b = a.bias_part != nullptr && nt == 0 && wn == 0;   // bias gradient: ones (hi = 1, lo = 0) against the hi and the lo fragments of dY
    f32x4 accb[4], accbx[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) { accb[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; accbx[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    h16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3C00;

    if (kt0 < kt1) stage(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int slot = (kt - kt0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < kt1) stage(kt + 1, slot ^ 1);
        const char* ta = smem + slot * STAGE;
        const char* tb = ta + TILE;
        h16x8 fah[2], fal[2], fbh[4], fbl[4];
#pragma unroll
        for (int it = 0; it < 2; ++it) { fah[it] = frag(tb, wn * 2 + it, 0); fal[it] = frag(tb, wn * 2 + it, 1); }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) { fbh[jt] = frag(ta, wm * 4 + jt, 0); fbl[jt] = frag(ta, wm * 4 + jt, 1); }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                acc[it][jt] = mfma_16x16x32<true>(fah[it], fbh[jt], acc[it][jt]);
                accx[it][jt] = mfma_16x16x32<true>(fah[it], fbl[jt], accx[it][jt]);
                accx[it][jt] = mfma_16x16x32<true>(fal[it], fbh[jt], accx[it][jt]);
            }
        if (do_b) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                accb[jt] = mfma_16x16x32<true>(ones, fbh[jt], accb[jt]);
                accbx[jt] = mfma_16x16x32<true>(ones, fbl[jt], accbx[jt]);
            }
        }
    }
    if (do_b && lane < 16) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int n = n0 + wm * 64 + 16 * jt + lane;
            if (n < a.Nout) a.bias_part[(size_t)split * a.Nout + n] = fmaf(accbx[jt][0], 1.0f / 2048.f, accb[jt][0]);
        }
    }
    float* mine = a.part + (size_t)split * a.Nout * a.Ncols;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int n = n0 + wm * 64 + 16 * jt + (lane & 15);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int col = col0 + wn * 32 + 16 * it + 4 * g;
            if (n < a.Nout && col < a.Ncols) {
                float4 o;
                o.x = fmaf(accx[it][jt][0], 1.0f / 2048.f, acc[it][jt][0]);
                o.y = fmaf(accx[it][jt][1], 1.0f / 2048.f, acc[it][jt][1]);
                o.z = fmaf(accx[it][jt][2], 1.0f / 2048.f, acc[it][jt][2]);
                o.w = fmaf(accx[it][jt][3], 1.0f / 2048.f, acc[it][jt][3]);
                *reinterpret_cast<float4*>(mine + (size_t)n * a.Ncols + col) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, size_t n4,
                                                        const float* __restrict__ bpart, float* __restrict__ bout, int nb) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = *reinterpret_cast<const float4*>(part + i * 4);
        for (int sp = 1; sp < splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)sp * n4 + i) * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i * 4) = s;
    }
    if (bpart)   // the bias gradient's split partials, summed in split order like the weight gradient's
        for (int n = (int)(blockIdx.x * blockDim.x + threadIdx.x); n < nb; n += (int)(gridDim.x * blockDim.x)) {
            float s = bpart[n];
            for (int sp = 1; sp < splits; ++sp) s += bpart[(size_t)sp * nb + n];
            bout[n] = s;
        }
}

// the deferred form of tn_reduce_kernel (train.h TnDefer): blockIdx.y = gradient, grid-stride over its float4s; the same split-order sums
struct TnBatch { TnPending e[48]; };
__global__ __launch_bounds__(256) void tn_reduce_batch_kernel(TnBatch b) {
    const TnPending& e = b.e[blockIdx.y];
    const size_t n4 = (size_t)e.n4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = *reinterpret_cast<const float4*>(e.part + i * 4);
        for (int sp = 1; sp < e.splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(e.part + ((size_t)sp * n4 + i) * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (e.perm_C) {   // tap-major [N][9][C] element 4 i .. 4 i + 3 (one tap, four consecutive channels) -> [N][C][3][3]
            const size_t el = i * 4, per_n = (size_t)9 * e.perm_C;
            const size_t nn = el / per_n, r = el - nn * per_n;
            const int tap = (int)(r / e.perm_C), c = (int)(r - (size_t)tap * e.perm_C);
            float* o = e.out + (nn * e.perm_C + c) * 9 + tap;
            o[0] = s.x; o[9] = s.y; o[18] = s.z; o[27] = s.w;
        } else {
            *reinterpret_cast<float4*>(e.out + i * 4) = s;
        }
    }
    if (e.bpart)
        for (int n = (int)(blockIdx.x * blockDim.x + threadIdx.x); n < e.nb; n += (int)(gridDim.x * blockDim.x)) {
            float s = e.bpart[n];
            for (int sp = 1; sp < e.splits; ++sp) s += e.bpart[(size_t)sp * e.nb + n];
            e.bout[n] = s;
        }
}

}  // namespace

int tn_flush(TnDefer& d, hipStream_t st, std::string& err) {
    for (size_t i = 0; i < d.pend.size(); i += 48) {
        TnBatch b;
        const size_t n = std::min<size_t>(48, d.pend.size() - i);
        for (size_t k = 0; k < n; ++k) b.e[k] = d.pend[i + k];
        SOCCDPT_LAUNCH(tn_reduce_batch_kernel, dim3(64, (unsigned)n), dim3(256), 0, st, b);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("tn_flush: ") + hipGetErrorString(e); return 1; }
    }
    d.pend.clear();
    d.used = 0;
    return 0;
}

// Shapes the kernel takes (the caller falls back to the transposing path otherwise)
bool tr_wgrad_tn_ok(size_t K, int Nout, int C, int taps) {
    if (K % 64 != 0 || K < 256) return false;
    if (taps == 9) return Nout % 128 == 0 && C % 128 == 0;   // a column tile must not straddle two taps
    return taps == 1 && Nout % 32 == 0 && C % 32 == 0;         // edge tiles are masked (the operands are over-read by up to 127 columns: callers keep them inside scratch)
}

// out [Nout][taps * C] f32 (tap-major for taps == 9).  A = dY [K][ldA] and B = X [K][ldB] are 16-bit (bf16, or fp16 when f16 == 1) or x3 tensors (f16 == 3); for taps == 9 both are
// in halo pixel order with pitch rp and B must be readable (finite) from row -(rp + 1) to row K + rp: zero margins.  part: at least
// splits * Nout * taps * C floats.  Returns the number of splits used through *splits_out.
int tr_wgrad_tn(const uint16_t* A, long ldA, const uint16_t* B, long ldB, size_t K, int Nout, int C, int taps, int rp, int f16, float* part, size_t part_floats,
                float* out, hipStream_t st, std::string& err, float* bias_out, TnDefer* defer, int perm_C) {
    if (!tr_wgrad_tn_ok(K, Nout, C, taps)) { err = "wgrad_tn: unsupported shape"; return 1; }
    const bool x3 = f16 == 3;   // x3 operands: 4 bytes per element, rows start at multiples of 16 elements, 32-row k-tiles
    if (x3 ? ((ldA & 15) || (ldB & 15)) : ((ldA & 7) || (ldB & 7))) { err = "wgrad_tn: row strides must be multiples of 8 (x3: 16) elements"; return 1; }
    TnArgs a;
    a.A = A; a.B = B; a.ldA = ldA; a.ldB = ldB; a.K = (int)K; a.Nout = Nout; a.C = C; a.taps = taps; a.Ncols = taps * C; a.rp = rp; a.part = part;
    const long tiles = (long)((Nout + 127) / 128) * ((a.Ncols + 127) / 128), nk = (long)K / (x3 ? 32 : 64);
    static const long target = getenv("SOCCDPT_TN_TILES") ? atol(getenv("SOCCDPT_TN_TILES")) : 512;
    long S = target / tiles > 0 ? target / tiles : 1;   // one round of two workgroups per CU
    if (S > nk / 2) S = nk / 2 > 0 ? nk / 2 : 1;
    if (S > 64) S = 64;
    const size_t per_split = (size_t)Nout * a.Ncols + (bias_out ? (size_t)(Nout + 3) / 4 * 4 : 0);   // the bias partials sit behind the weight partials
    while (S > 1 && (size_t)S * per_split > part_floats) --S;   // (part_floats bounds the split count in the deferred form too: the same splits, the same bits)
    if ((size_t)S * per_split > part_floats) { err = "wgrad_tn: partial-tile scratch too small"; return 1; }
    if (defer && defer->arena) {   // the partials stay in the arena until tn_flush; a full arena is flushed first (stream order: the sums run before the region is rewritten)
        const size_t need = ((size_t)S * per_split + 63) & ~size_t(63);
        if (need > defer->cap) defer = nullptr;
        else {
            if (defer->used + need > defer->cap && tn_flush(*defer, st, err)) return 1;
            part = defer->arena + defer->used;
            defer->used += need;
            a.part = part;
        }
    } else defer = nullptr;
    if (perm_C && !defer) { err = "wgrad_tn: the permuted output exists in the deferred form only"; return 1; }
    a.splits = (int)S;
    a.bias_part = bias_out ? part + (size_t)S * Nout * a.Ncols : nullptr;
    const dim3 grid((unsigned)(tiles * S)), block(512);
    const size_t lds = 2 * 2 * 64 * 256;
    if (x3) SOCCDPT_LAUNCH(wgrad_tn_x3_kernel, grid, block, lds, st, a);
    else if (f16) SOCCDPT_LAUNCH(wgrad_tn_kernel<true>, grid, block, lds, st, a);
    else SOCCDPT_LAUNCH(wgrad_tn_kernel<false>, grid, block, lds, st, a);
    const size_t n4 = (size_t)Nout * a.Ncols / 4;
    if (defer) {
        defer->pend.push_back(TnPending{part, out, (const float*)a.bias_part, bias_out, (unsigned long long)n4, (int)S, Nout, perm_C ? Nout : 0, perm_C});
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) { err = std::string("wgrad_tn: ") + hipGetErrorString(e0); return 1; }
        return 0;
    }
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    SOCCDPT_LAUNCH(tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, out, (int)S, n4, (const float*)a.bias_part, bias_out, Nout);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("wgrad_tn: ") + hipGetErrorString(e); return 1; }
    return 0;
}

}  // namespace soccdpt
