// Score + mask + top-K (P5 full-sort evaluation, P6 kNN build).   (a10, a11, a7)
//
// replaces  scores = U_b I^T ; scores[mask] = -1e10 ; topk(scores, K)   (trainer.py:304-309).
// Roofline: fp32 MFMA (2*nq*nc*kd FLOP).  Order: score descending, ties by lower candidate id.
//
// Implementations behind mmrec_score_topk_f32:
//  * kd == 64 with >= 4096 candidates (every full-sort evaluation of the Amazon shapes): fp16 matrix-core FILTER +
//    exact fp32 refinement, topk_filter.hip (0.165 ms on the Baby evaluation against 0.41 ms for the form below);
//  * kd == 64, fewer candidates (or flags & MMREC_TOPK_NO_FILTER): MATERIALISED -- the score block of up to 8 GB worth of
//    queries is written once by the output-bound streaming GEMM of mfma_stream.h (which also emits
//    <= 384 group maxima per query), then `select_topk_kernel` masks, bounds, sweeps and sorts each
//    row on a wave of its own.  See the comment above that kernel.  Baby shape: 0.42 ms against
//    0.65 ms for the fused form below (19445 x 7050, k = 50); kNN-shaped 7050^2: 0.15 vs 0.28 ms.
//  * other kd % 32 == 0 (kNN over 96 / 384 / 4096-d features): MATERIALISED too, with the general-K
//    `gemm_nt_kernel` (mfma_stream.h) (128 x 128 tiles, both operands by LDS-DMA, no transposes) and the bound
//    taken from one extra sweep of the row: 90-124 TFLOP/s whole-call against 16-19 for the fused form.
//  * remaining kd (not a multiple of 32): FUSED -- the [n_query, n_cand] scores are never
//    written; one wave owns 32 queries and streams candidate tiles:
//   D[cand][query] = C_tile Q_tile^T on v_mfma_f32_32x32x2_f32 ("swapped" orientation, so a lane
//   holds ONE query (col = lane&31) and 16 candidates: the running threshold of that query is one
//   register and the common path is 16 compares per tile).  Masked (train-positive) candidates: the
//   query's sorted mask list is walked by a register cursor in step with the candidate stream -> a
//   32-bit "masked" word per tile with no memory access on the common path.
//   Selection: scores above the query's threshold are appended to a 128-slot LDS list; a list that
//   threatens to overflow is bitonic-sorted, cut to k and the threshold raised to its k-th score --
//   nothing that can be in the top-k is ever dropped.  Candidates are split over gridDim.y waves per
//   query block; a rank-counting merge kernel turns the per-split lists into the final sorted top-k.
//   (For kd == 64 the fused form also had a first MFMA pass for per-group maxima -> a lower bound of
//   the k-th score; it is kept, behind MMREC_TOPK_FUSED_ONLY, as the measured alternative.)
#include "mfma_stream.h"
#include "topk_sort.h"
#include "topk_filter.h"
#include <limits.h>

// tools/topk_probe.hip builds this file with an ablation mask (the library only ever uses 0):
// bit0 operand loads only for the first tile, bit1 no selection work, bit2 no MFMAs
#ifndef MMREC_TOPK_PROBE
#define MMREC_TOPK_PROBE 0
#endif
#ifndef MMREC_TOPK_FUSED_ONLY
#define MMREC_TOPK_FUSED_ONLY 0   // probe: never materialise the score block
#endif
#ifndef MMREC_TOPK_S_MB
#define MMREC_TOPK_S_MB 8192    // size limit of the materialised score block
#endif

namespace {

constexpr int TK_Q = 32;        // queries per wave
constexpr int TK_CAPH = 64;     // survivor slots per lane (= per query half) in single-pass mode
constexpr int TK_CAPH2 = 48;    // ... in two-pass mode (few survivors); CAPH - 16 >= k/2 keeps a full tile safe after a compaction
constexpr int TK_MAXGROUPS = 128;
constexpr size_t TK_S_BYTES_MAX = (size_t)MMREC_TOPK_S_MB << 20;  // group maxima per query (2 per lane in the selection sort)

// Everything a wave needs to turn a 32-candidate tile into scores for its 32 queries.
//
// Operand feeding.  Lane l supplies A[cand = l&31][kk = l>>5] of every MFMA step, i.e. ONE candidate
// row per lane.  Reading row-major C that way makes every load instruction touch 32 different
// cache lines and use 32 B of each (measured: the L1 path, not the matrix pipe, then sets the pace,
// 38 % MFMA utilisation).  So the candidates are transposed once per call into Ct[kd_pad][nc_pad]
// (zero padded, tiled LDS transpose): step s / half h needs Ct[kc + 2s + h][c0 + (l&31)], two fully
// used 128-B lines per load instruction, no bounds checks in the loop, and the contraction runs in
// natural k order.  For kd != 64 the queries are transposed the same way (Qt); for kd == 64 the
// query fragment is loaded once from row-major Q and stays in registers.
template <bool KD64>
struct TileScorer {
    const float* Qs;   // KD64: Q [nq][64] row-major ; else Qt [kd_pad][ldq]
    const float* Ct;   // [kd_pad][ldc]
    int nq, nc, kd_pad, ldq, ldc, q, i, h;
    bool q_ok;
    float qf[32];
    // mask cursor
    const int32_t* mask_col;
    int m_cur, m_hi, w0, w1, w2, w3, wn;

    __device__ __forceinline__ void init(const float* Qs_, const float* Ct_, int nq_, int nc_,
                                         int kd_pad_, int ldq_, int ldc_,
                                         const int32_t* mask_rowptr, const int32_t* mask_col_, int q0,
                                         int lane, int c_begin) {
        Qs = Qs_; Ct = Ct_; nq = nq_; nc = nc_; kd_pad = kd_pad_; ldq = ldq_; ldc = ldc_;
        i = lane & 31; h = lane >> 5; q = q0 + i; q_ok = q < nq;
        mask_col = mask_col_;
        m_cur = (mask_rowptr && q_ok) ? mask_rowptr[q] : 0;
        m_hi = (mask_rowptr && q_ok) ? mask_rowptr[q + 1] : 0;
        int lo = m_cur, hi = m_hi;  // first mask entry >= c_begin (lists are sorted)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (mask_col[mid] < c_begin) lo = mid + 1; else hi = mid;
        }
        m_cur = lo;
        refill();
        if (KD64) {
#pragma unroll
            for (int s = 0; s < 32; ++s) qf[s] = q_ok ? Qs[(size_t)q * 64 + 2 * s + h] : 0.f;
        }
    }
    __device__ __forceinline__ void refill() {
        w0 = m_cur + 0 < m_hi ? mask_col[m_cur + 0] : INT_MAX;
        w1 = m_cur + 1 < m_hi ? mask_col[m_cur + 1] : INT_MAX;
        w2 = m_cur + 2 < m_hi ? mask_col[m_cur + 2] : INT_MAX;
        w3 = m_cur + 3 < m_hi ? mask_col[m_cur + 3] : INT_MAX;
        wn = 4;
    }
    __device__ __forceinline__ void load_q(int kc) {   // general kd: query chunk from Qt (coalesced)
        const int qq = blockIdx.x * TK_Q + i;          // < ldq (padded)
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] = Qs[(size_t)(kc + 2 * s + h) * ldq + qq];
    }
    // Tile operand: step s / half h reads Ct[kc + 2s + h][c0 + i].  The row pointer is wave-uniform
    // (scalar registers) and the lane part h*ldc + i never changes, so a load costs no VALU work --
    // which matters: fp32-input MFMA shares the vector pipe, every VALU instruction beside it costs
    // ~5 cycles of matrix time (tools/mfma_valu_probe.hip).
    __device__ __forceinline__ void load_c(float (&a)[32], int c0, int kc) const {
        // SRSRC buffer loads: descriptor + scalar row offset are wave-uniform (SGPRs), the lane part
        // is one constant VGPR -> `buffer_load_dword v, v_lane, s[rsrc], s_row offen`, zero VALU.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(Ct + (size_t)kc * ldc), 0, 64 * ldc * 4, 0x00020000);
        const int lbyte = (h * ldc + i) * 4;
        if ((MMREC_TOPK_PROBE & 1) && c0 >= 64 + (int)blockIdx.y * 0) { if (c0 & 0x40000000) a[0] = 1.f; return; }
#pragma unroll
        for (int s = 0; s < 32; ++s)
            a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                 rsrc, lbyte, (2 * s * ldc + c0) * 4, 0));
    }
    // 32 dependent-free-enough MFMAs: one accumulator chain sustains the issue rate (probe: 145-155 TF)
    __device__ __forceinline__ f32x16 mma(const float (&a)[32], f32x16 acc) const {
        if (MMREC_TOPK_PROBE & 4) { acc[0] = a[0] * qf[0] + a[31] * qf[31]; acc[7] = a[5] * qf[9]; return acc; }
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qf[s], acc, 0, 0, 0);
        return acc;
    }
    // general kd: all chunks of one tile, no cross-tile prefetch
    __device__ __forceinline__ f32x16 tile_general(float (&a)[32], int c0) {
        f32x16 acc = {0};
        for (int kc = 0; kc < kd_pad; kc += 64) {
            load_q(kc);
            load_c(a, c0, kc);
            acc = mma(a, acc);
        }
        return acc;
    }
    // masked candidates score -1e10 (trainer.py:307); candidates >= nc (padding) score -inf
    __device__ __forceinline__ void apply_mask(f32x16& acc, unsigned mbits, int c0) const {
        if (__any(mbits != 0u)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
                acc[r] = ((mbits >> j) & 1u) ? -1e10f : acc[r];
            }
        }
        if (c0 + 32 > nc) {  // last, partial tile only (uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
                acc[r] = (c0 + j < nc) ? acc[r] : -INFINITY;
            }
        }
    }
    static __device__ __forceinline__ float max16(const f32x16& a) {
        const float m0 = fmaxf(fmaxf(a[0], a[1]), a[2]), m1 = fmaxf(fmaxf(a[3], a[4]), a[5]);
        const float m2 = fmaxf(fmaxf(a[6], a[7]), a[8]), m3 = fmaxf(fmaxf(a[9], a[10]), a[11]);
        const float m4 = fmaxf(fmaxf(a[12], a[13]), a[14]);
        return fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, a[15]));
    }
    // bit j set: candidate c0 + j is masked for this lane's query
    __device__ __forceinline__ unsigned mask_bits(int c0) {
        unsigned mbits = 0;
        while (w0 < c0 + 32) {
            mbits |= 1u << (w0 - c0);
            w0 = w1; w1 = w2; w2 = w3; w3 = INT_MAX;
            ++m_cur;
            if (--wn == 0) refill();
        }
        return mbits;
    }
};

// dst[c][r] = src[r][c] for r < rows, c < cols, zero elsewhere; dst is [cols_pad][ld] with ld >= rows
// padded.  32x32 tiles through LDS (+1 pad), both sides coalesced.  grid (ld/32, cols_pad/32).
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src, int rows,
                                                            int cols, float* __restrict__ dst, int ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        dst[(size_t)c * ld + r] = tile[tx][ty + 8 * k];
    }
}

// ---- pass 1 (kd == 64): per query, maximum score of every candidate group --------------------
__global__ __launch_bounds__(64) void score_groupmax_kernel(
    const float* __restrict__ Q, const float* __restrict__ Ct, int nq, int nc, int kd_pad, int ldq,
    int ldc, const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col,
    int tiles_per_group, int groups_per_wave, int n_groups, float* __restrict__ gmax) {
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * TK_Q;
    const int g_begin = blockIdx.y * groups_per_wave;
    const int g_end = min(g_begin + groups_per_wave, n_groups);
    if (g_begin >= g_end) return;
    const int c_begin = g_begin * tiles_per_group * 32;
    const int c_end = min(g_end * tiles_per_group * 32, nc);
    TileScorer<true> ts;
    ts.init(Q, Ct, nq, nc, kd_pad, ldq, ldc, mask_rowptr, mask_col, q0, lane, c_begin);
    float a0[32], a1[32];
    float gm = -INFINITY;
    int g = g_begin, t_in_g = 0;
    auto consume = [&](f32x16 acc, int c0) {
        if (MMREC_TOPK_PROBE & 2) { if (acc[3] == 1234.5f) gm = 1.f; return; }
        ts.apply_mask(acc, ts.mask_bits(c0), c0);
        gm = fmaxf(gm, TileScorer<true>::max16(acc));
        if (++t_in_g == tiles_per_group || c0 + 32 >= c_end) {  // group complete
            gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
            if (ts.q_ok && lane < 32) gmax[(size_t)ts.q * n_groups + g] = gm;
            gm = -INFINITY; t_in_g = 0; ++g;
        }
    };
    // two tiles per trip on ping-pong operand registers: the next tile's loads fly under the MFMAs
    // and no register copies are needed
    ts.load_c(a0, c_begin, 0);
    for (int c0 = c_begin; c0 < c_end; c0 += 64) {
        const bool has1 = c0 + 32 < c_end;
        if (has1) ts.load_c(a1, c0 + 32, 0);
        consume(ts.mma(a0, f32x16{0}), c0);
        if (has1) {
            if (c0 + 64 < c_end) ts.load_c(a0, c0 + 64, 0);
            consume(ts.mma(a1, f32x16{0}), c0 + 32);
        }
    }
}

// thr[q] = k-th largest of gmax[q][0..n_groups) (n_groups <= 128): one wave per query sorts the
// group maxima (2 per lane) with the same bitonic network as the candidate lists.
__global__ __launch_bounds__(64) void kth_largest_kernel(const float* __restrict__ gmax, int n_groups,
                                                         int k, float* __restrict__ thr) {
    const int q = blockIdx.x, lane = threadIdx.x;
    Cand x0, x1;
    x0.v = lane < n_groups ? gmax[(size_t)q * n_groups + lane] : -INFINITY;
    x0.i = lane;
    x1.v = lane + 64 < n_groups ? gmax[(size_t)q * n_groups + lane + 64] : -INFINITY;
    x1.i = lane + 64;
    bitonic128(x0, x1, lane);
    const float kth = k <= 64 ? __shfl(x0.v, (k - 1) & 63, 64) : __shfl(x1.v, (k - 65) & 63, 64);
    if (lane == 0) thr[q] = kth;
}

// ---- scoring + selection pass ---------------------------------------------------------------
// Survivor lists are PRIVATE to a lane (= one query x one half of the candidate rows): CAPH slots
// plus one write-only dump slot.  Appending is branch-free and atomic-free -- per candidate one
// compare, one select (real slot or dump slot), one 8-byte LDS store, one counter add -- because on
// gfx950 every VALU instruction issued beside fp32 MFMAs costs matrix time.  A query's two half
// lists are merged, sorted and cut to k by the whole wave only when one of them could overflow.
template <bool KD64, int CAPH>
__global__ __launch_bounds__(64) void score_topk_kernel(
    const float* __restrict__ Q, const float* __restrict__ Ct, int nq, int nc, int kd_pad, int ldq,
    int ldc, const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col, int k,
    int tiles_per_wave, const float* __restrict__ thr0, int64_t* __restrict__ out_idx,
    float* __restrict__ out_val, int* __restrict__ tmp_idx, float* __restrict__ tmp_val) {
    __shared__ unsigned long long s_list[64][CAPH + 1];
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * TK_Q;
    const int c_begin = blockIdx.y * tiles_per_wave * 32;
    const int c_end = min(c_begin + tiles_per_wave * 32, nc);
    TileScorer<KD64> ts;
    ts.init(Q, Ct, nq, nc, kd_pad, ldq, ldc, mask_rowptr, mask_col, q0, lane, c_begin);
    const int i = ts.i;
    int my_cnt = 0;

    // all entries of query qq (both half lists), element e = lane -> x0, e = lane + 64 -> x1
    auto gather = [&](int qq, Cand& x0, Cand& x1) -> int {
        const int n0 = __shfl(my_cnt, qq, 64), n1 = __shfl(my_cnt, qq + 32, 64);
        const int n = n0 + n1;
        auto fetch = [&](int e) -> Cand {
            if (e < n0) return unpack_cand(s_list[qq][e]);
            if (e < n) return unpack_cand(s_list[qq + 32][e - n0]);
            return Cand{-INFINITY, INT_MAX};
        };
        x0 = fetch(lane);
        x1 = fetch(lane + 64);
        return n;
    };
    // Sort query qq's entries, keep the best min(n,k) (split back over the two half lists); returns
    // the k-th score (a strict threshold from now on) or -inf when fewer than k entries exist.
    auto compact = [&](int qq) -> float {
        Cand x0, x1;
        const int n = gather(qq, x0, x1);
        bitonic128(x0, x1, lane);
        const int keep = min(n, k), h0 = (keep + 1) >> 1;
        __syncthreads();
        if (lane < h0) s_list[qq][lane] = pack_cand(x0.v, x0.i);
        else if (lane < keep) s_list[qq + 32][lane - h0] = pack_cand(x0.v, x0.i);
        if (lane == qq) my_cnt = h0;
        if (lane == qq + 32) my_cnt = keep - h0;
        __syncthreads();
        return n >= k ? __shfl(x0.v, k - 1, 64) : -INFINITY;
    };

    // effective threshold: a score is kept iff s > teff (teff just below the bound while ties with
    // it must still be kept, the bound itself once a compaction made it strict)
    float teff = float_below((thr0 && ts.q_ok) ? thr0[ts.q] : -INFINITY);
    float a0[32], a1[32];
    auto consume = [&](f32x16 acc, int c0) {
        if (MMREC_TOPK_PROBE & 2) { if (acc[3] == 1234.5f) teff = 1.f; return; }
        ts.apply_mask(acc, ts.mask_bits(c0), c0);
        const bool hit = ts.q_ok && TileScorer<KD64>::max16(acc) > teff;
        if (!__any(hit)) return;  // nothing of this tile can enter any list
        const float t = hit ? teff : INFINITY;
        int cnt = my_cnt;
        const int idx0 = c0 + 4 * ts.h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool pass = acc[r] > t;
            s_list[lane][pass ? cnt : CAPH] = pack_cand(acc[r], idx0 + (r & 3) + 8 * (r >> 2));
            cnt += pass ? 1 : 0;
        }
        my_cnt = cnt;
        // a tile adds at most 16 entries per lane: compact the queries that might overflow next time
        unsigned long long mm = __ballot(my_cnt > CAPH - 16);
        mm = (mm | (mm >> 32)) & 0xffffffffull;
        while (mm) {
            const int qq = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const float kth = compact(qq);
            if (i == qq && kth > teff) teff = kth;   // strict from now on (later ties have higher ids)
        }
    };
    if (KD64) {
        if (c_begin < c_end) ts.load_c(a0, c_begin, 0);
        for (int c0 = c_begin; c0 < c_end; c0 += 64) {
            const bool has1 = c0 + 32 < c_end;
            if (has1) ts.load_c(a1, c0 + 32, 0);
            consume(ts.mma(a0, f32x16{0}), c0);
            if (has1) {
                if (c0 + 64 < c_end) ts.load_c(a0, c0 + 64, 0);
                consume(ts.mma(a1, f32x16{0}), c0 + 32);
            }
        }
    } else {
        for (int c0 = c_begin; c0 < c_end; c0 += 32) consume(ts.tile_general(a0, c0), c0);
    }
    __syncthreads();
    // emit: final sorted top-k (single split) or this split's list for the merge kernel
    for (int qq = 0; qq < TK_Q; ++qq) {
        Cand x0, x1;
        const int n = gather(qq, x0, x1);
        if (q0 + qq >= nq) continue;  // uniform
        const size_t o = (size_t)(q0 + qq) * k + lane;
        if (gridDim.y > 1) {
            if (n > k) bitonic128(x0, x1, lane);   // rare: cut to the best k; else hand over unsorted
            if (lane < k) {
                const size_t t = (size_t)blockIdx.y * nq * k + o;
                tmp_idx[t] = lane < n ? x0.i : INT_MAX;
                tmp_val[t] = lane < n ? x0.v : -INFINITY;
            }
        } else {
            bitonic128(x0, x1, lane);
            if (lane < k) {
                out_idx[o] = lane < n ? (int64_t)x0.i : (int64_t)-1;
                if (out_val) out_val[o] = lane < n ? x0.v : -INFINITY;
            }
        }
    }
}

// Final top-k of the S per-split lists (S*k <= 512 slots per query, most of them padding): valid
// entries are packed with a ballot prefix, then rank counting writes every survivor straight to its
// sorted position.  One wave per query.
__global__ __launch_bounds__(64) void merge_topk_kernel(const int* __restrict__ tmp_idx,
                                                        const float* __restrict__ tmp_val, int nq,
                                                        int k, int n_split,
                                                        int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_val) {
    __shared__ float v[512];
    __shared__ int id[512];
    const int q = blockIdx.x, lane = threadIdx.x;
    const int n = n_split * k;
    int n_valid = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        int my_id = INT_MAX;
        float my_v = -INFINITY;
        if (e < n) {
            const int s = e / k, j = e - s * k;
            const size_t t = (size_t)s * nq * k + (size_t)q * k + j;
            my_id = tmp_idx[t];
            my_v = tmp_val[t];
        }
        const unsigned long long ok = __ballot(my_id != INT_MAX);
        if (my_id != INT_MAX) {
            const int pos = n_valid + __popcll(ok & ((1ull << lane) - 1ull));
            v[pos] = my_v;
            id[pos] = my_id;
        }
        n_valid += __popcll(ok);
    }
    __syncthreads();
    for (int e = lane; e < n_valid; e += 64) {
        const Cand me{v[e], id[e]};
        int rank = 0;
        for (int j = 0; j < n_valid; ++j) rank += cand_before(Cand{v[j], id[j]}, me);
        if (rank < k) {
            out_idx[(size_t)q * k + rank] = (int64_t)me.i;
            if (out_val) out_val[(size_t)q * k + rank] = me.v;
        }
    }
}

// ---- materialised path (kd == 64): S = Q Ct by the streaming GEMM, then one wave per query -----
// The fused kernels above pay for selection with VALU instructions issued beside fp32 MFMAs, which
// on gfx950 run on the same pipe (tools/mfma_valu_probe.hip), and run the MFMAs twice.  With 288 GB
// of HBM the other trade is cheaper: write the score block once with the output-bound streaming GEMM
// (mfma_stream.h; its only extra work is 16 v_max per 32 MFMAs for <= 128 group maxima per query),
// then select on waves that do nothing else, ONE sweep of the row per query:
//   bound: with m masked items in the row, the (k+m)-th largest group maximum is a lower bound of the
//      k-th unmasked score (k+m groups reach it and at most m of those maxima are masked items);
//   mask: the wave first writes -1e10 over its row's masked positions (trainer.py:307), in place;
//   sweep: every score >= the bound (~k..2k of them) is appended to an LDS list by ballot prefix; the
//      list is bitonic-sorted (score desc, id asc) and cut to k.
// Exactness does not depend on the bound being tight: should the list fill up (ties, K > #unmasked,
// k + m > #groups) it is compacted at a sweep-step boundary and the threshold becomes the strict k-th
// score so far (all later ids are larger, so ties lose, as in a stable descending sort).
// The kernel is one driver loop with a single sort site (bound / compaction / final all go through
// it): three inlined copies of the sorting network would not fit the instruction cache.
constexpr int SEL_CAP = GEMM64_MAX_GROUPS;   // list slots per query; compaction once > 128 are in use (a step adds <= 256)

__global__ __launch_bounds__(256) void select_topk_kernel(
    float* S, int ld, int rows, int q0, int nc, int k,
    const float* __restrict__ gmax, int n_groups,
    const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col,
    int64_t* __restrict__ out_idx, float* __restrict__ out_val,
    const int* __restrict__ row_map, const int* __restrict__ rows_live) {   // (topk_wide.h: row ql is query row_map[ql], of *rows_live)
    __shared__ unsigned long long s_all[4][SEL_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = blockIdx.x * 4 + wave;
    if (ql >= rows || (rows_live && ql >= *rows_live)) return;   // no workgroup-level synchronisation below: waves are independent
    unsigned long long* list = s_all[wave];
    const int q = row_map ? row_map[ql] : q0 + ql;
    float* row = S + (size_t)ql * ld;
    const float4* row4 = reinterpret_cast<const float4*>(row);
    const int m_lo = mask_rowptr ? mask_rowptr[q] : 0, m_hi = mask_rowptr ? mask_rowptr[q + 1] : 0;
    // scores[mask] = -1e10 (trainer.py:307), in place in this wave's own row, before the row is read
    // (stores are write-through to L2 and this CU has never read the row, so waiting for them to be
    // acknowledged is enough for the wave's own later loads; an agent-scope fence here would write
    // back the whole L2 once per wave -- measured 2.3x slower)
    if (m_hi > m_lo && !(MMREC_TOPK_PROBE & 16)) {
        for (int e = m_lo + lane; e < m_hi; e += 64) {
            const int c = mask_col[e];
            if (c >= 0 && c < nc) row[c] = -1e10f;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    const int steps = (MMREC_TOPK_PROBE & 8) ? 1 : (nc + 255) / 256;          // 256 candidates per step (float4 per lane)
    const int full_steps = nc / 256;             // steps without a candidate >= nc
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto max4 = [](float4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); };
    float teff = -INFINITY;   // keep iff score > teff
    int cnt = 0;
    auto offer = [&](float v, int c) {   // wave-wide: every lane offers one candidate (or -inf)
        bool pass = v > teff;
        if (!__any(pass)) return;
        const unsigned long long b = __ballot(pass);
        if (pass) list[cnt + __popcll(b & lt)] = pack_cand(v, c);
        cnt += __popcll(b);
    };
    // the group maxima go through the same list + sort as everything else
    if (gmax) {
        for (int e = lane; e < max(n_groups, 128); e += 64)   // n_groups <= SEL_CAP
            list[e] = pack_cand(e < n_groups ? gmax[(size_t)ql * n_groups + e] : -INFINITY, e);
        cnt = max(n_groups, 128);
    } else {
        // no maxima from the GEMM (general kd): one extra sweep of the row, 128 lane-strided groups
        float gm0 = -INFINITY, gm1 = -INFINITY;
        for (int st = 0; st < steps; ++st) {
            float4 x = row4[st * 64 + lane];
            if (st >= full_steps) {
                const int c = (st * 64 + lane) * 4;
                x.x = c + 0 < nc ? x.x : -INFINITY; x.y = c + 1 < nc ? x.y : -INFINITY;
                x.z = c + 2 < nc ? x.z : -INFINITY; x.w = c + 3 < nc ? x.w : -INFINITY;
            }
            if (st & 1) gm1 = fmaxf(gm1, max4(x)); else gm0 = fmaxf(gm0, max4(x));
        }
        list[lane] = pack_cand(gm0, lane);
        list[lane + 64] = pack_cand(gm1, lane + 64);
        cnt = 128;
    }
    enum { BOUND, COMPACT, FINAL };
    int why = BOUND, it = 0;
    // four row pieces in flight per wave, rotated through registers (one in flight leaves the sweep
    // bound by loaded-HBM latency: 28 dependent ~2 us steps)
    auto piece = [&](int p) { return p < steps ? row4[p * 64 + lane] : float4{0.f, 0.f, 0.f, 0.f}; };
    float4 v = piece(0), v1 = piece(1), v2 = piece(2), v3 = piece(3);
    for (;;) {
        // ---- sort list[0..cnt): best 64 in y0 (rank = lane), next 64 in y1
        const int n = cnt;
        auto fetch = [&](int e) -> Cand { return e < n ? unpack_cand(list[e]) : Cand{-INFINITY, INT_MAX}; };
        // more than 128 entries (rare): the rest is folded in 64 at a time.  k <= 64 needs the best 64 only: the next chunk
        // simply replaces y1 (whose values then bound ranks 64 .. 127 from below: still a valid BOUND).  k > 64 keeps the best
        // 128: the chunk z first meets y1 (the better half of y1 + z survives), then y0 and y1 are merged again -- through the
        // ONE inlined sorting network (see above).
        const bool deep = k > 64;
        Cand y0 = fetch(lane), y1 = fetch(64 + lane), z{-INFINITY, INT_MAX};
        int pos = 128;
        bool pair_yz = false;
        for (;;) {
            Cand a = pair_yz ? y1 : y0, b = pair_yz ? z : y1;
            bitonic128(a, b, lane);
            if (pair_yz) { y1 = a; pair_yz = false; continue; }
            y0 = a; y1 = b;
            if (pos >= n) break;
            z = fetch(pos + lane);
            pos += 64;
            if (deep) pair_yz = true; else y1 = z;
        }
        if (why == BOUND) {
            const int rank = k + (m_hi - m_lo) - 1;
            const float bound = rank < 64 ? __shfl(y0.v, rank & 63, 64)
                                : rank < 128 ? __shfl(y1.v, (rank - 64) & 63, 64) : -INFINITY;
            teff = float_below(bound);
            cnt = 0;
        } else {
            const int keep = min(n, k);
            if (why == FINAL) {
                if (lane < k) {
                    const size_t o = (size_t)q * k + lane;
                    out_idx[o] = lane < n ? (int64_t)y0.i : (int64_t)-1;
                    if (out_val) out_val[o] = lane < n ? y0.v : -INFINITY;
                }
                if (lane + 64 < k) {       // k = 65 .. 128: ranks 64 .. k - 1 sit in y1
                    const size_t o = (size_t)q * k + lane + 64;
                    out_idx[o] = lane + 64 < n ? (int64_t)y1.i : (int64_t)-1;
                    if (out_val) out_val[o] = lane + 64 < n ? y1.v : -INFINITY;
                }
                return;
            }
            if (lane < keep) list[lane] = pack_cand(y0.v, y0.i);
            if (lane + 64 < keep) list[lane + 64] = pack_cand(y1.v, y1.i);
            cnt = keep;
            if (n >= k) teff = fmaxf(teff, k <= 64 ? __shfl(y0.v, k - 1, 64) : __shfl(y1.v, k - 65, 64));   // strict from now on
        }
        // ---- sweep on until the list needs compacting or the row ends
        why = FINAL;
        while (it < steps) {
            float4 cur = v;
            const int c = (it * 64 + lane) * 4;
            ++it;
            v = v1; v1 = v2; v2 = v3; v3 = piece(it + 3);
            if (it > full_steps) {
                cur.x = c + 0 < nc ? cur.x : -INFINITY; cur.y = c + 1 < nc ? cur.y : -INFINITY;
                cur.z = c + 2 < nc ? cur.z : -INFINITY; cur.w = c + 3 < nc ? cur.w : -INFINITY;
            }
            if (!__any(max4(cur) > teff)) continue;
            offer(cur.x, c); offer(cur.y, c + 1); offer(cur.z, c + 2); offer(cur.w, c + 3);
            if (cnt > SEL_CAP - 256) { why = COMPACT; break; }
        }
    }
}

}  // namespace
#include "topk_wide.h"
namespace {

struct TopkPlan {
    int n_tiles, n_split, tiles_per_wave, two_pass, tiles_per_group, n_groups, groups_per_wave;
    int materialise, qb_rows;   // kd == 64: score block of qb_rows queries in the workspace
};
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline TopkPlan topk_plan(int nq, int nc, int kd, int k) {
    TopkPlan p;
    p.n_tiles = cdiv(nc, 32);
    const int qblocks = cdiv(nq, TK_Q);
    // split count: ~2048 waves are resident at once (2 per SIMD); estimate the makespan as
    // rounds * (tiles per wave + fixed per-wave cost) and take the best split that keeps >= 4k
    // candidates per split and S*k <= 512 merge slots
    int max_s = nc / (4 * k) > 0 ? nc / (4 * k) : 1;
    if (max_s > 16) max_s = 16;
    if (max_s * k > 512) max_s = 512 / k;
    if (max_s < 1) max_s = 1;
    int s = 1;
    long best = -1;
    for (int c = 1; c <= max_s; ++c) {
        const long rounds = cdiv(qblocks * c, 2048);
        const long cost = rounds * (cdiv(p.n_tiles, c) + 4);
        if (best < 0 || cost < best) { best = cost; s = c; }
    }
#ifdef MMREC_TOPK_SPLIT
    s = MMREC_TOPK_SPLIT;
#endif
    p.two_pass = (kd == 64 && p.n_tiles >= 2 * k) ? 1 : 0;
    p.tiles_per_group = p.two_pass ? cdiv(p.n_tiles, TK_MAXGROUPS) : 1;
    p.n_groups = cdiv(p.n_tiles, p.tiles_per_group);
    // splits cover whole groups so that both passes use the same candidate ranges
    p.groups_per_wave = cdiv(p.n_groups, s);
    p.tiles_per_wave = p.groups_per_wave * p.tiles_per_group;
    p.n_split = cdiv(p.n_tiles, p.tiles_per_wave);
    // materialised path: the score block S[qb_rows][pad256(nc)] lives in the workspace, capped
    // kd == 64: streaming GEMM + its group maxima; other kd % 32 == 0: score_gemm_nt_kernel
    p.materialise = ((kd % 32) == 0 && nc <= (2 << 20) && !MMREC_TOPK_FUSED_ONLY) ? 1 : 0;
    p.qb_rows = 0;
    if (p.materialise) {
        // block of queries whose scores are materialised at once: as many as fit TK_S_BYTES_MAX
        // (measured on Baby / Sports / Clothing shapes: one big block beats Infinity-Cache-sized ones,
        // the GEMM grid efficiency matters more than where the select sweep finds its row), in equal
        // blocks (a short last block would run an underfilled grid)
        const size_t row_bytes = (size_t)cdiv(nc, 256) * 256 * 4;
        size_t rows = TK_S_BYTES_MAX / row_bytes / 128 * 128;
        if (rows < 128) rows = 128;
        p.qb_rows = (size_t)nq < rows ? nq : (int)rows;
        const int nblk = cdiv(nq, p.qb_rows);
        p.qb_rows = cdiv(cdiv(nq, nblk), 128) * 128;
        if (p.qb_rows > nq) p.qb_rows = nq;
        p.two_pass = 0; p.n_split = 1;
    }
    return p;
}
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

extern "C" size_t mmrec_topk_workspace_bytes(int32_t nq, int32_t nc, int32_t kd, int32_t k) {
    if (nq <= 0 || nc <= 0 || k <= 0 || kd <= 0) return 0;
    const TopkPlan p = topk_plan(nq, nc, kd, k);
    const int kd_pad = pad_to(kd, 64), ldc = pad_to(nc, 256), ldq = pad_to(nq, 32);
    if (p.materialise) {  // [Ct (kd == 64 only)] [S block] [group maxima]
        const size_t b = (kd == 64 ? al256((size_t)kd_pad * ldc * 4) : 0) + al256((size_t)p.qb_rows * ldc * 4) +
                         al256((size_t)p.qb_rows * GEMM64_MAX_GROUPS * 4);
        // either path may serve the call (`flags` of mmrec_score_topk_f32 decides): size for both
        const size_t f = topk64_filter_applicable(nq, nc, kd, k) ? topk64_filter_workspace_bytes(nq, nc, kd, k) : 0;
        const size_t w = topk_wide_applicable(nq, nc, kd, k) ? topk_wide_extra_bytes(nc, kd, p.qb_rows) : 0;   // wide rows: + fp16 copies, lists
        return (b > f ? b : f) + w;
    }
    size_t b = al256((size_t)kd_pad * ldc * 4);               // Ct
    if (kd != 64) b += al256((size_t)kd_pad * ldq * 4);        // Qt
    if (p.two_pass) b += al256((size_t)nq * p.n_groups * 4) + al256((size_t)nq * 4);
    if (p.n_split > 1) b += 2 * al256((size_t)p.n_split * nq * k * 4);
    return b;
}

extern "C" size_t mmrec_topk_prepared_bytes(int32_t nc, int32_t kd) {
    // the fp16 filter's candidate side; 0 = no shape of this (nc, kd) is served by it
    return (nc > 0 && topk64_filter_applicable(1, nc, kd, 1)) ? topk64_filter_prepared_bytes(nc, kd) : 0;
}

extern "C" int mmrec_topk_prepare_f32(const float* C, int32_t nc, int32_t kd, void* prepared, mmrec_stream_t stream) {
    if (!C || !prepared || nc <= 0) return MMREC_ERR_BAD_ARG;
    if (!topk64_filter_applicable(1, nc, kd, 1)) return MMREC_ERR_UNSUPPORTED;
    return topk64_filter_prepare(C, nc, kd, prepared, mmrec_stream(stream));
}

static int score_topk_impl(const float* Q, const float* C, const void* prepared, int32_t nq, int32_t nc,
                           int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col,
                           int32_t k, int64_t* out_idx, float* out_val, void* workspace,
                           int32_t flags, mmrec_stream_t stream, const FilterHint& hint = FilterHint()) {
    if (nq < 0 || nc < 0 || kd <= 0 || (kd & 3)) return MMREC_ERR_UNSUPPORTED;
    if (flags & ~MMREC_TOPK_NO_FILTER) return MMREC_ERR_BAD_ARG;
    if (k <= 0 || k > MMREC_TOPK_MAX || k > nc) return MMREC_ERR_BAD_ARG;
    if (nq == 0) return 0;
    if (!Q || !C || !out_idx || !workspace) return MMREC_ERR_BAD_ARG;
    if (mask_rowptr == nullptr && mask_col != nullptr) return MMREC_ERR_BAD_ARG;
    const TopkPlan p = topk_plan(nq, nc, kd, k);
    const int kd_pad = pad_to(kd, 64), ldc = pad_to(nc, 256), ldq = pad_to(nq, 32);
    char* ws = static_cast<char*>(workspace);
    hipStream_t s = mmrec_stream(stream);
    if (p.materialise && !(flags & MMREC_TOPK_NO_FILTER) && topk64_filter_applicable(nq, nc, kd, k))
        return topk64_filter_launch(Q, C, nq, nc, kd, mask_rowptr, mask_col, k, out_idx, out_val, workspace, prepared, s, hint);
    if (hint.ids) return MMREC_ERR_UNSUPPORTED;      // a warm call is a call of the fp16 filter
    if (!p.materialise && k > MMREC_TOPK_MAX_OTHER) return MMREC_ERR_UNSUPPORTED;   // 65..128: not on the fused fp32 path (kd % 32 != 0)
    if (p.materialise) {
        float* Ct = nullptr;
        if (kd == 64) {
            Ct = reinterpret_cast<float*>(ws); ws += al256((size_t)kd_pad * ldc * 4);
            hipLaunchKernelGGL(transpose_pad_kernel, dim3(ldc / 32, kd_pad / 32), dim3(256), 0, s, C, nc, kd,
                               Ct, ldc);
        }
        float* S = reinterpret_cast<float*>(ws); ws += al256((size_t)p.qb_rows * ldc * 4);
        float* gm = reinterpret_cast<float*>(ws);
        if (!(flags & MMREC_TOPK_NO_FILTER) && topk_wide_applicable(nq, nc, kd, k))   // wide rows: fp16 pass + exact refinement
            return topk_wide_launch(Q, C, nq, nc, kd, mask_rowptr, mask_col, k, out_idx, out_val, S, ldc, p.qb_rows,
                                    reinterpret_cast<char*>(gm) + al256((size_t)p.qb_rows * GEMM64_MAX_GROUPS * 4), s);
        for (int q0 = 0; q0 < nq; q0 += p.qb_rows) {
            const int rows = nq - q0 < p.qb_rows ? nq - q0 : p.qb_rows;
            if (kd == 64) {
                const int groups = gemm64_stream_gmax_launch(Q + (size_t)q0 * 64, Ct, S, rows, ldc, gm, nc, s);
                hipLaunchKernelGGL(select_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, S, ldc, rows,
                                   q0, nc, k, gm, groups, mask_rowptr, mask_col, out_idx, out_val, (const int*)nullptr,
                                   (const int*)nullptr);
            } else {
                gemm_nt_launch(Q + (size_t)q0 * kd, C, nullptr, S, rows, nc, kd, ldc, ldc, s);
                hipLaunchKernelGGL(select_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, S, ldc, rows,
                                   q0, nc, k, (const float*)nullptr, 0, mask_rowptr, mask_col, out_idx, out_val,
                                   (const int*)nullptr, (const int*)nullptr);
            }
        }
        MMREC_RETURN_LAUNCH_STATUS();
    }
    float* Ct = reinterpret_cast<float*>(ws); ws += al256((size_t)kd_pad * ldc * 4);
    float* Qt = nullptr;
    if (kd != 64) { Qt = reinterpret_cast<float*>(ws); ws += al256((size_t)kd_pad * ldq * 4); }
    float *gmax = nullptr, *thr = nullptr, *tmp_val = nullptr;
    int* tmp_idx = nullptr;
    if (p.two_pass) {
        gmax = reinterpret_cast<float*>(ws); ws += al256((size_t)nq * p.n_groups * 4);
        thr = reinterpret_cast<float*>(ws);  ws += al256((size_t)nq * 4);
    }
    if (p.n_split > 1) {
        tmp_idx = reinterpret_cast<int*>(ws);   ws += al256((size_t)p.n_split * nq * k * 4);
        tmp_val = reinterpret_cast<float*>(ws);
    }
    const int qblocks = (nq + TK_Q - 1) / TK_Q;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(ldc / 32, kd_pad / 32), dim3(256), 0, s, C, nc, kd, Ct,
                       ldc);
    if (Qt)
        hipLaunchKernelGGL(transpose_pad_kernel, dim3(ldq / 32, kd_pad / 32), dim3(256), 0, s, Q, nq, kd,
                           Qt, ldq);
    const dim3 grid(qblocks, p.n_split);
    if (p.two_pass) {
        hipLaunchKernelGGL(score_groupmax_kernel, grid, dim3(64), 0, s, Q, Ct, nq, nc, kd_pad, ldq, ldc,
                           mask_rowptr, mask_col, p.tiles_per_group, p.groups_per_wave, p.n_groups, gmax);
        hipLaunchKernelGGL(kth_largest_kernel, dim3(nq), dim3(64), 0, s, gmax, p.n_groups, k, thr);
    }
    if (p.two_pass)
        hipLaunchKernelGGL((score_topk_kernel<true, TK_CAPH2>), grid, dim3(64), 0, s, Q, Ct, nq, nc, kd_pad,
                           ldq, ldc, mask_rowptr, mask_col, k, p.tiles_per_wave, thr, out_idx, out_val,
                           tmp_idx, tmp_val);
    else if (kd == 64)
        hipLaunchKernelGGL((score_topk_kernel<true, TK_CAPH>), grid, dim3(64), 0, s, Q, Ct, nq, nc, kd_pad,
                           ldq, ldc, mask_rowptr, mask_col, k, p.tiles_per_wave, thr, out_idx, out_val,
                           tmp_idx, tmp_val);
    else
        hipLaunchKernelGGL((score_topk_kernel<false, TK_CAPH>), grid, dim3(64), 0, s, Qt, Ct, nq, nc,
                           kd_pad, ldq, ldc, mask_rowptr, mask_col, k, p.tiles_per_wave, thr, out_idx,
                           out_val, tmp_idx, tmp_val);
    if (p.n_split > 1)
        hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(64), 0, s, tmp_idx, tmp_val, nq, k,
                           p.n_split, out_idx, out_val);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_score_topk_f32(const float* Q, const float* C, int32_t nq, int32_t nc,
                                    int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col,
                                    int32_t k, int64_t* out_idx, float* out_val, void* workspace,
                                    int32_t flags, mmrec_stream_t stream) {
    return score_topk_impl(Q, C, nullptr, nq, nc, kd, mask_rowptr, mask_col, k, out_idx, out_val, workspace, flags, stream);
}

extern "C" int mmrec_score_topk_prepared_f32(const float* Q, const float* C, const void* prepared, int32_t nq, int32_t nc,
                                             int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col,
                                             int32_t k, int64_t* out_idx, float* out_val, void* workspace,
                                             int32_t flags, mmrec_stream_t stream) {
    if (!prepared) return MMREC_ERR_BAD_ARG;
    return score_topk_impl(Q, C, prepared, nq, nc, kd, mask_rowptr, mask_col, k, out_idx, out_val, workspace, flags, stream);
}

extern "C" int mmrec_score_topk_hinted_f32(const float* Q, const float* C, const void* prepared, int32_t nq, int32_t nc,
                                           int32_t kd, const int32_t* mask_rowptr, const int32_t* mask_col, int32_t k,
                                           int32_t* hint, int32_t hint_k, const int64_t* hint_rows,
                                           int64_t* out_idx, float* out_val, void* workspace, int32_t* queue_counts,
                                           int32_t flags, mmrec_stream_t stream) {
    if (!hint || hint_k < k || hint_k > MMREC_TOPK_MAX || (flags & ~(MMREC_TOPK_HINT_COLD | MMREC_TOPK_HINT_KEEP))) return MMREC_ERR_BAD_ARG;
    if (nq > 0 && nc > 0 && !topk64_filter_applicable(nq, nc, kd, k)) return MMREC_ERR_UNSUPPORTED;
    FilterHint h;
    h.ids = hint; h.hk = hint_k; h.rows = hint_rows; h.queue_counts = queue_counts;
    h.cold = (flags & MMREC_TOPK_HINT_COLD) != 0;
    h.update = (flags & MMREC_TOPK_HINT_KEEP) == 0;
    return score_topk_impl(Q, C, prepared, nq, nc, kd, mask_rowptr, mask_col, k, out_idx, out_val, workspace, 0, stream, h);
}
