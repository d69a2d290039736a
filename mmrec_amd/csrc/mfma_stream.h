// Shared fp32-MFMA / LDS-DMA building blocks of the dense kernels (gemm.hip, topk.hip), and the
// streaming 64-deep GEMM  out[n, F] = A[n, 64] B[64, F]  both of them launch:
//   gemm.hip : dX = dY W          (projection backward, freedom.py:58,61 trainable feature tables)
//   topk.hip : S  = Q Ct          (full-sort scores, freedom.py:219 / trainer.py:304)
// Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MMREC_STREAM_PROBE
#define MMREC_STREAM_PROBE 0   // ablation mask of the probes under tools/ (the library uses 0)
#endif
#ifndef MMREC_STREAM_STORE_AUX
#define MMREC_STREAM_STORE_AUX 0   // cache policy bits of the streaming GEMM's output stores (probe: 2 = nt)
#endif

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int d_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ float4 ld4_guard(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const float4*>(p) : f4_zero();
}

#define MMREC_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw (unstrided) buffer descriptor in SGPRs: base, num_records bytes, gfx9 data-format word
__device__ __forceinline__ i32x4 raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const float*)p;
}
// One LDS-DMA piece: 64 lanes x 16 B from rsrc[voff + soff] to LDS[m0 .. +1 KB), lane-linear.
// Issued as asm so that the compiler's waitcnt bookkeeping does not see it (it would drain the
// whole queue, vmcnt(0), at the first LDS read of a loop-carried stage); the pipeline below counts
// its own vmcnt.  M0 is saved and restored inside the statement.
// NT = non-temporal policy for data one CU reads once (the X stream); W stays default (L2 resident).
template <bool NT>
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned lds, int voff, int soff) {
    unsigned keep;
    if (NT)
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff)
            : "memory");
}

// out[n, F] = A[n, 64] B[64, F], streaming form (F % 128 == 0; in gemm.hip A = dY, B = W, out = dX):
// a workgroup owns 128 rows ("items") and walks `ftiles` consecutive
// 128-wide f tiles.  Waves 0-3 compute: each keeps its 32 x 64 dY fragment in registers for the
// whole walk (read once, straight from global: a lane's eight 16-B loads cover its row) and issues
// nothing but MFMAs, LDS reads and its 64 row-segment stores per tile.  Wave 4 is the loader: it
// brings the W tiles by LDS-DMA into a double buffer in their natural [k][f] layout
// (lane-consecutive ds_read_b32) and is the only wave that waits on vmcnt -- on gfx9 stores and loads
// share that counter and complete out of order with respect to each other, so a wave that did both
// could only ever wait for vmcnt(0), i.e. for its own stores.  One barrier per tile.
// Output-write bound: n*F*4 bytes at the HBM store rate.
//
// GMAX (topk.hip): every compute lane also keeps, for each of its 16 rows, the running maximum of the
// outputs it produced over the walk (columns = its residue mod 32 within this workgroup's f range) and
// writes them to gmax[row][32 * blockIdx.y + (lane & 31)]: 32 * gridDim.y "group maxima" per row at
// 16 v_max per 32 MFMAs.  Tiles reaching past column `valid_cols` (zero padding) are left out of the
// maxima (a maximum over a subset is all the caller needs).
template <bool GMAX>
__global__ __launch_bounds__(320, 2) void gemm64_stream_kernel(const float* __restrict__ dY,
                                                                     const float* __restrict__ W,
                                                                     float* __restrict__ dX, int n,
                                                                     int F, int ftiles,
                                                                     float* __restrict__ gmax,
                                                                     int valid_cols) {
    __shared__ __attribute__((aligned(1024))) float Wa[64 * 128], Wb[64 * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 128;
    const int ft0 = blockIdx.y * ftiles, ftn = min(ftiles, F / 128 - ft0);
    if (wave == 4) {  // ------------------------------------------------------------ loader wave
        const i32x4 rw = raw_rsrc(W, 64u * (unsigned)F * 4u);
        const int vw = (lane >> 5) * F * 4 + (lane & 31) * 16;  // a piece = 2 k rows x 512 B
        auto fill = [&](float* ws, int ft) {
            const int so = (ft0 + ft) * 128 * 4;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) lds_dma16<false>(rw, lds_addr(ws + 2 * j * 128), vw, so + 2 * j * F * 4);
            MMREC_WAIT_VM(0);
        };
        if (ftn > 0) fill(Wa, 0);
        for (int ft = 0; ft < ftn;) {
            __builtin_amdgcn_s_barrier();            // tile ft landed; the other buffer is drained
            if (ft + 1 < ftn) fill(Wb, ft + 1);
            if (++ft >= ftn) break;
            __builtin_amdgcn_s_barrier();
            if (ft + 1 < ftn) fill(Wa, ft + 1);
            ++ft;
        }
        return;
    }
    const int i = lane & 31, h = lane >> 5;
    // A fragments: dY[row][k8*8 + 4h .. +3], rows past n read as zero
    const int arow = m0 + wave * 32 + i;
    float4 fa[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) fa[k8] = ld4_guard(dY + (size_t)arow * 64 + k8 * 8 + 4 * h, arow < n);
    const bool full = m0 + 128 <= n;  // uniform: interior workgroups store without row guards
    // stores go through an SRSRC over this workgroup's rows: scalar row/tile offset + one constant
    // per-lane voffset, no per-store address VALU
    const unsigned lane_off = (unsigned)(4 * h * F + i) * 4u;
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dX + (size_t)m0 * F), 0, (unsigned)min(128, n - m0) * (unsigned)F * 4u, 0x00020000);
    // One 128-wide f tile = four 32-wide sub-tiles done one after the other on alternating
    // accumulators: the 16 row-segment stores of sub-tile t-1 are slotted between the MFMAs of
    // sub-tile t (a store issued right behind the MFMA that produced it would stall the wave until
    // that MFMA retires), and the B values of sub-tile t+1 are fetched under the MFMAs of t.
    float rmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rmax[r] = -INFINITY;
    auto tile = [&](const float* ws, int ft, auto guard) {
        constexpr bool GUARD = decltype(guard)::value;
        const bool gm_ok = GMAX && (ft0 + ft + 1) * 128 <= valid_cols;   // wave-uniform
        f32x16 acc[2];
        // B values are fetched half a sub-tile (16 k) ahead of their MFMAs
        float bq[2][16];
        auto fetch = [&](float* b, int j) {   // half j: sub-tile j >> 1, k = 16 (j & 1) .. +15
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * (j & 1) + kk;
                b[kk] = ws[((k >> 2) * 8 + 4 * h + (k & 3)) * 128 + (j >> 1) * 32 + i];
            }
        };
        const int tcol = ((ft0 + ft) * 128) * 4;  // byte offset of this f tile within a row
        fetch(bq[0], 0);
#pragma unroll
        for (int j = 0; j < 10; ++j) {           // halves 0..7 compute, 8..9 drain the last stores
            const int t = j >> 1;
            if (j < 7) fetch(bq[(j + 1) & 1], j + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * (j & 1) + kk;
                if (t < 4 && !((MMREC_STREAM_PROBE & 512) && k >= 4)) {
                    const float4 a4 = fa[k >> 2];
                    const float a = (k & 3) == 0 ? a4.x : (k & 3) == 1 ? a4.y : (k & 3) == 2 ? a4.z : a4.w;
                    if (k == 0) {
                        const f32x16 z = {0};
                        acc[t & 1] = mfma32(a, bq[j & 1][kk], z);
                    } else {
                        acc[t & 1] = mfma32(a, bq[j & 1][kk], acc[t & 1]);
                    }
                }
                if (t > 0 && (k & 1) && !(MMREC_STREAM_PROBE & 256)) {
                    const int r = k >> 1, rr = (r & 3) + 8 * (r >> 2);
                    const float v = acc[(t - 1) & 1][r];  // (bit_cast straight off the vector element picks lane 0 of it)
                    if (GMAX && gm_ok) rmax[r] = fmaxf(rmax[r], v);
                    if (!GUARD || m0 + wave * 32 + rr + 4 * h < n)
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __float_as_uint(v), rdx, (int)lane_off,
                            (wave * 32 + rr) * F * 4 + tcol + (t - 1) * 128, MMREC_STREAM_STORE_AUX);
                }
                if (k & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int ft = 0; ft < ftn;) {
        __builtin_amdgcn_s_barrier();
        if (full) tile(Wa, ft, std::false_type{}); else tile(Wa, ft, std::true_type{});
        if (++ft >= ftn) break;
        __builtin_amdgcn_s_barrier();
        if (full) tile(Wb, ft, std::false_type{}); else tile(Wb, ft, std::true_type{});
        ++ft;
    }
    if (GMAX) {
        const int G = 32 * gridDim.y;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wave * 32 + d_row(r, lane);
            if (row < n) gmax[(size_t)row * G + 32 * blockIdx.y + i] = rmax[r];
        }
    }
}

// ---- C[M, :N] = A[M, K] B[N, K]^T (+ bias[N]),  K % 32 == 0 -----------------------------------
// topk.hip: S = Q C^T for the kNN builds; gemm.hip: nn.Linear with more than 64 outputs.
// Both operands are row-major with the contraction index contiguous, so neither needs a transpose:
// 128 x 128 output tile per workgroup, wave w owns rows 32w..+31 and all 128 columns (4
// accumulators), BK = 32 tiles of both operands by LDS-DMA into a double buffer (source-side bank
// swizzle as in gemm.hip's forward), one barrier per tile, tile t+1 in flight under the 64 MFMAs per
// wave of tile t.  32 FLOP per operand byte; consecutive workgroups share the C tile through L2 and
// the query block is small enough to stay cache resident, so HBM streams C once per query block.
// one 128 x 128 output tile at (m0, n0); As0 .. Bs1: the workgroup's four 16 KB LDS stages
__device__ __forceinline__ void gemm_nt_tile(const float* __restrict__ A, const float* __restrict__ B,
                                             const float* __restrict__ bias, float* __restrict__ S, int M, int N, int K,
                                             int lds_, int ncols, int m0, int n0, float* As0, float* As1, float* Bs0,
                                             float* Bs1) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_a = min(128, M - m0), rows_b = max(0, min(128, N - n0));
    const i32x4 ra = raw_rsrc(A + (size_t)m0 * K, (unsigned)rows_a * (unsigned)K * 4u);
    const i32x4 rb = raw_rsrc(B + (size_t)n0 * K, (unsigned)rows_b * (unsigned)K * 4u);
    int vo[4];   // this wave's 4 pieces (8 rows each) of either operand: rows 32w + 8j + lane/8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        vo[j] = r * K * 4 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](float* as, float* bs, int t) {
        const int so = t * 32 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<false>(ra, lds_addr(as + (4 * wave + j) * 256), vo[j], so);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<false>(rb, lds_addr(bs + (4 * wave + j) * 256), vo[j], so);
    };
    const int i = lane & 31, h = lane >> 5, g = (i >> 1) & 7;
    int ko[4];
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) ko[k8] = ((2 * k8 + h) ^ g) << 2;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    auto compute = [&](const float* as, const float* bs) {
        const float* xa = as + (32 * wave + i) * 32;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const float4 a = *reinterpret_cast<const float4*>(xa + ko[k8]);
            float4 b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const float4*>(bs + (32 * t + i) * 32 + ko[k8]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(a.x, b[t].x, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(a.y, b[t].y, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(a.z, b[t].z, acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(a.w, b[t].w, acc[t]);
        }
    };
    const int T = K / 32;
    issue(As0, Bs0, 0);
    for (int t = 0; t < T;) {
        MMREC_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();              // tile t landed everywhere; the other stage is drained
        if (t + 1 < T) issue(As1, Bs1, t + 1);
        compute(As0, Bs0);
        if (++t >= T) break;
        MMREC_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        if (t + 1 < T) issue(As0, Bs0, t + 1);
        compute(As1, Bs1);
        ++t;
    }
    // row-segment stores through an SRSRC over this workgroup's rows (bounds drop the rows past M)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(S + (size_t)m0 * lds_), 0, (unsigned)rows_a * (unsigned)lds_ * 4u, 0x00020000);
    const int lane_off = (4 * h * lds_ + n0 + i) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = wave * 32 + (r & 3) + 8 * (r >> 2);
        if (rr + 4 * h < rows_a) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = n0 + 32 * t + i;
                const float v = acc[t][r] + (bias && col < N ? bias[col] : 0.f);
                if (col < ncols)   // ncols = N for a plain GEMM, the padded row length for the score blocks
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, lane_off, (rr * lds_ + 32 * t) * 4, 0);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A,
                                                        const float* __restrict__ B,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ S, int M, int N, int K,
                                                        int lds_, int ncols) {
    __shared__ __attribute__((aligned(1024))) float As0[128 * 32], As1[128 * 32], Bs0[128 * 32], Bs1[128 * 32];
    gemm_nt_tile(A, B, bias, S, M, N, K, lds_, ncols, blockIdx.x * 128, blockIdx.y * 128, As0, As1, Bs0, Bs1);
}

// The same GEMM over the first *m_live rows of A only -- a DEVICE-side row count (topk_wide.h: the queries on the rescue
// queue, usually none).  A 1-D grid of resident workgroups walks the live tiles (row tile fastest: the workgroups of a
// round share one 128-row slab of B through L2); with an empty queue every workgroup returns at once.  (The 2-D grid of the
// plain kernel dispatched 125,000 workgroups with 64 KB of LDS each just to have them return: 0.93 ms per 4096 x 500,000
// block, 0.22 s of the config-5 kNN builds -- profiles/r04_c5_plugin_kernels.txt.)
__global__ __launch_bounds__(256, 2) void gemm_nt_live_kernel(const float* __restrict__ A,
                                                             const float* __restrict__ B,
                                                             float* __restrict__ S, int M, int N, int K,
                                                             int lds_, int ncols, const int* __restrict__ m_live) {
    __shared__ __attribute__((aligned(1024))) float As0[128 * 32], As1[128 * 32], Bs0[128 * 32], Bs1[128 * 32];
    const int live = min(*m_live, M);
    if (live <= 0) return;
    const int tm = (live + 127) / 128, total = tm * ((ncols + 127) / 128);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        gemm_nt_tile(A, B, nullptr, S, M, N, K, lds_, ncols, (t % tm) * 128, (t / tm) * 128, As0, As1, Bs0, Bs1);
        __syncthreads();       // the next tile's first LDS-DMA must not land while a wave still reads this tile's last stage
    }
}

// `ncols` columns of every row are written (>= N: the columns past N hold zeros + nothing else).
inline void gemm_nt_launch(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                           int ldc, int ncols, hipStream_t s, const int* m_live = nullptr) {
    if (m_live) {
        const long tiles = (long)((M + 127) / 128) * ((ncols + 127) / 128);
        hipLaunchKernelGGL(gemm_nt_live_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, s, A, B, C, M, N, K,
                           ldc, ncols, m_live);
        return;
    }
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((M + 127) / 128, (ncols + 127) / 128), dim3(256), 0, s, A, B, bias, C,
                       M, N, K, ldc, ncols);
}

// Launch of gemm64_stream_kernel.  f tiles per workgroup: long walks win (measured: 8 tiles beat 2 even
// when that leaves fewer workgroups than CUs); shorten only while the grid would cover under 3/4 of
// the chip.  Requires F % 128 == 0, n > 0.
inline void gemm64_stream_launch(const float* A, const float* B, float* out, int n, int F,
                                 hipStream_t s) {
    const int rt = (n + 127) / 128, nft = F / 128;
    int ftiles = 8;
    while (ftiles > 1 && (long)rt * ((nft + ftiles - 1) / ftiles) < 192) ftiles >>= 1;
#ifdef MMREC_BX_FTILES
    ftiles = MMREC_BX_FTILES;
#endif
    hipLaunchKernelGGL(gemm64_stream_kernel<false>, dim3(rt, (nft + ftiles - 1) / ftiles), dim3(320), 0, s,
                       A, B, out, n, F, ftiles, (float*)nullptr, 0);
}
// Same GEMM, also producing gmax[n][32 * ranges] (see GMAX above).  4 column ranges (128 groups per
// row) when the row tiles alone fill the chip, up to 12 (384 groups) when they do not; returns the
// number of groups per row.
constexpr int GEMM64_MAX_GROUPS = 384;
inline int gemm64_stream_gmax_launch(const float* A, const float* B, float* out, int n, int F,
                                     float* gmax, int valid_cols, hipStream_t s) {
    const int rt = (n + 127) / 128, nft = F / 128;
    int want = (256 + rt - 1) / rt;
    if (want < 4) want = 4;
    if (want > GEMM64_MAX_GROUPS / 32) want = GEMM64_MAX_GROUPS / 32;
    if (want > nft) want = nft;
    const int ftiles = (nft + want - 1) / want;
    const int ranges = (nft + ftiles - 1) / ftiles;
    hipLaunchKernelGGL(gemm64_stream_kernel<true>, dim3(rt, ranges), dim3(320), 0, s, A, B, out, n, F,
                       ftiles, gmax, valid_cols);
    return 32 * ranges;
}

}  // namespace
