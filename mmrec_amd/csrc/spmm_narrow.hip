// CSR SpMM on ONE FEATURE SLICE of the embedding tables: rows of d = 8, 16 or 32 floats (DESIGN.md 6
// "feature-sliced layout").
//
// Why: Y = A X is independent per COLUMN of X, so P GPUs that each own 64 / P columns of every table (and the whole CSR)
// propagate any number of layers with NO exchange at all -- the row-sharded layouts move 224-336 MB into every rank per
// layer at config 5 against ~0.1 ms of SpMM.  A rank's launch is this kernel: the whole graph, rows of 64 / P floats.
//
// Bit-exactness: a column's sum is the d = 64 kernel's (spmm.hip), operation for operation -- short rows one sequential
// fma chain in CSR order (padded to 8-nonzero steps with 0 * 0 terms exactly where that kernel pads), long rows cut into
// the same MMREC_SPMM_CHUNK-nonzero chunks, each chunk summed by the same 16 virtual groups (spans of 16 nonzeros, stride
// 256) and the same fixed-order sums over groups and chunks -- so the P slices of a feature-sliced run ARE the columns of
// the single-GPU result (tests: 8 slices == the d = 64 launch bit for bit).
//
// Mapping: a row is LPR = d / 4 lanes x float4 (2 / 4 / 8 lanes).  One LPR-lane sub-group owns one matrix row, so a wave64
// works on 32 / 16 / 8 rows at once and every gather instruction fetches that many 32 / 64 / 128-B row slices; a sub-group
// reads its (col, val) pairs itself, 16 / LPR per lane and step, and passes them round with DPP-width shuffles.  Chunk
// blocks: a 256-thread workgroup takes 16 / LPR chunks at once (16 virtual groups x LPR lanes each).
// Roofline: fabric / Infinity-Cache gathers; algorithmic bytes per launch = (8 + 4 d) per nonzero + (4 + 4 d) per row.
#include "common.h"
#include "spmm_narrow.h"

namespace {

struct NarrowEpilogue {
    const float* Z;
    float* Y;
    const float* acc_in;
    float* acc_out;
    float alpha, beta, acc_scale;
};

template <int LPR>
__device__ __forceinline__ void store_row(const NarrowEpilogue& ep, int row, int t, float4 sum) {
    const size_t off = (size_t)row * LPR + t;      // float4 index
    float4 y = f4_scale(ep.alpha, sum);
    if (ep.Z) y = f4_fma(ep.beta, reinterpret_cast<const float4*>(ep.Z)[off], y);
    if (ep.Y) reinterpret_cast<float4*>(ep.Y)[off] = y;
    if (ep.acc_out) {
        const float4 a = reinterpret_cast<const float4*>(ep.acc_in)[off];
        reinterpret_cast<float4*>(ep.acc_out)[off] = f4_scale(ep.acc_scale, f4_add(a, y));
    }
}

// acc += sum_{k in [s, e)} vals[k] * X[colidx[k]] for one LPR-lane sub-group (t = float4 slot of the lane), in the d = 64
// kernel's order: windows of 16 nonzeros, 8 gathers in flight, the tail of a window padded with 0 * 0 terms up to the next
// multiple of 8 (gather_span of spmm.hip).  All lanes of a sub-group run the same trip counts.
template <int LPR>
__device__ __forceinline__ void gather_span(const int32_t* __restrict__ colidx, const float* __restrict__ vals,
                                            const float4* __restrict__ X4, int s, int e, int t, float4& acc) {
    constexpr int PPL = 16 / LPR;            // pairs per lane and window
    for (int base = s; base < e; base += 16) {
        int c[PPL];
        float v[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) {      // lane t holds pairs t, t + LPR, ... of the window
            const int k = base + t + LPR * i;
            c[i] = 0;
            v[i] = 0.f;
            if (k < e) {
                c[i] = colidx[k];
                v[i] = vals[k];
            }
        }
        const int cnt = min(16, e - base);
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += 8) {
            if (j0 >= cnt) break;            // uniform within the sub-group
            float4 x[8];
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const int cj = __shfl(c[j / LPR], j % LPR, LPR);
                vv[u] = __shfl(v[j / LPR], j % LPR, LPR);
                x[u] = (j < cnt) ? X4[(size_t)cj * LPR + t] : f4_zero();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = f4_fma(vv[u], x[u], acc);
        }
    }
}

// blocks [0, chunk_blocks): 16 / LPR long-row chunks each; blocks [chunk_blocks, ...): 256 / LPR short rows per pass
template <int LPR>
__global__ __launch_bounds__(256) void spmm_narrow_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ X, NarrowEpilogue ep, int n_rows, int long_t, int rows_per_group,
    const int32_t* __restrict__ long_rows, const int32_t* __restrict__ long_chunk_ptr, int n_long, int n_chunks,
    int chunk_blocks, float* __restrict__ partials) {
    constexpr int CPB = 16 / LPR;            // chunks per workgroup
    constexpr int SPB = 256 / LPR;           // sub-groups (rows) per workgroup
    __shared__ float4 red[CPB][16][LPR];
    const int t = threadIdx.x % LPR;
    const int sg = threadIdx.x / LPR;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    if ((int)blockIdx.x < chunk_blocks) {
        const int sb = sg >> 4, g = sg & 15;                 // sub-block (chunk of this workgroup), virtual group
        const int chunk = blockIdx.x * CPB + sb;
        const bool live = chunk < n_chunks;                  // uniform per sub-block
        int lo = 0, row = 0, cs = 0, ce = 0;
        if (live) {
            int hi = n_long;                                 // largest lo with long_chunk_ptr[lo] <= chunk
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (long_chunk_ptr[mid] <= chunk) lo = mid; else hi = mid;
            }
            row = long_rows[lo];
            cs = rowptr[row] + (chunk - long_chunk_ptr[lo]) * MMREC_SPMM_CHUNK;
            ce = min(cs + MMREC_SPMM_CHUNK, rowptr[row + 1]);
        }
        float4 acc = f4_zero();
        for (int base = cs + g * 16; base < ce; base += 256)
            gather_span<LPR>(colidx, vals, X4, base, min(base + 16, ce), t, acc);
        red[sb][g][t] = acc;
        __syncthreads();
        if (live && g == 0) {
            float4 s = red[sb][0][t];
#pragma unroll
            for (int i = 1; i < 16; ++i) s = f4_add(s, red[sb][i][t]);
            if (long_chunk_ptr[lo + 1] - long_chunk_ptr[lo] == 1)
                store_row<LPR>(ep, row, t, s);               // the whole row fitted one chunk: done
            else
                reinterpret_cast<float4*>(partials)[(size_t)chunk * LPR + t] = s;
        }
        return;
    }
    const int row0 = ((int)blockIdx.x - chunk_blocks) * SPB * rows_per_group + sg;
#pragma unroll 1
    for (int i = 0; i < rows_per_group; ++i) {
        const int row = row0 + i * SPB;
        if (row >= n_rows) break;
        const int s = rowptr[row], e = rowptr[row + 1];
        if (e - s > long_t) continue;                        // handled by the chunk blocks
        float4 acc = f4_zero();
        gather_span<LPR>(colidx, vals, X4, s, e, t, acc);
        store_row<LPR>(ep, row, t, acc);
    }
}

// 16 / LPR multi-chunk long rows per workgroup: virtual group g sums chunks g, g + 16, ... in order, then the fixed-order
// sum over the 16 groups (reduce_long_row of spmm.hip)
template <int LPR>
__global__ __launch_bounds__(256) void spmm_narrow_long_reduce_kernel(
    const int32_t* __restrict__ long_rows, const int32_t* __restrict__ long_chunk_ptr, int n_long,
    const float* __restrict__ partials, NarrowEpilogue ep) {
    constexpr int CPB = 16 / LPR;
    __shared__ float4 red[CPB][16][LPR];
    const int t = threadIdx.x % LPR, sg = threadIdx.x / LPR;
    const int sb = sg >> 4, g = sg & 15;
    const int i = blockIdx.x * CPB + sb;
    const int c0 = i < n_long ? long_chunk_ptr[i] : 0, c1 = i < n_long ? long_chunk_ptr[i + 1] : 0;
    const bool live = c1 - c0 > 1;           // single-chunk rows were finished by their chunk block
    float4 s = f4_zero();
    if (live)
        for (int c = c0 + g; c < c1; c += 16) s = f4_add(s, reinterpret_cast<const float4*>(partials)[(size_t)c * LPR + t]);
    red[sb][g][t] = s;
    __syncthreads();
    if (live && g == 0) {
        float4 r = red[sb][0][t];
#pragma unroll
        for (int k = 1; k < 16; ++k) r = f4_add(r, red[sb][k][t]);
        store_row<LPR>(ep, long_rows[i], t, r);
    }
}

// ---- column-sorted rows: all resident rows walk the column space IN STEP ----------------------------------------------
// A slice launch is bound by 128-B fabric lines, about one per nonzero (L2 hit rate 29 % at config 5: the 48 MB slice of X
// is 12 x an XCD's L2).  When every row's nonzeros are sorted by column (the normalised user-item graph: freedom.py:102-126
// builds it row-major sorted), the CSR order of a row IS ascending column order, so a wave may interleave the work of its rows
// freely as long as each row's own terms stay in order.  This kernel makes ALL rows of a launch resident at once -- a wave owns
// R x (64 / LPR) rows, their accumulators and cursors in registers -- and walks the columns in PHASES of `blk_rows` rows of X
// (2 MB: half an XCD's L2): in phase p every row consumes its nonzeros with column < (p + 1) blk_rows.  Waves do the same amount
// of work per phase on average, so without any barrier the whole chip gathers from the same 2-MB window at any time and the
// gathers hit L2.  A row's sum is the same fma chain as ever: same bits.  Long rows stay with the chunk blocks.
__device__ __forceinline__ float w_sel(unsigned m, float a, float b) {       // m all ones: a, all zeros: b
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

template <int LPR, int R>
__global__ __launch_bounds__(256, 4) void spmm_narrow_phased_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ X, NarrowEpilogue ep, int row_lo, int row_hi, int long_t, int n_cols, int blk_rows) {
    constexpr int SPW = 64 / LPR;            // rows a wave handles per register set
    const int lane = threadIdx.x & 63, t = lane % LPR, p = lane / LPR;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = row_lo + wv * (R * SPW) + p;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4 acc[R];
    int cur[R], rem[R], c[R];
    float v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int row = base + i * SPW;
        acc[i] = f4_zero();
        cur[i] = 0;
        rem[i] = -1;                         // no such row / a long row: nothing to do, nothing to store
        c[i] = INT32_MAX;
        v[i] = 0.f;
        if (row < row_hi) {
            const int s = rowptr[row], n = rowptr[row + 1] - s;
            if (n <= long_t) {
                cur[i] = s;
                rem[i] = n;
                if (n > 0) {
                    c[i] = __builtin_nontemporal_load(colidx + s);
                    v[i] = __builtin_nontemporal_load(vals + s);
                }
            }
        }
    }
    for (int lo = 0; lo < n_cols; lo += blk_rows) {
        const int hi = lo + blk_rows >= n_cols ? INT32_MAX : lo + blk_rows;
        for (;;) {
            bool mine = false;
#pragma unroll
            for (int i = 0; i < R; ++i) mine |= c[i] < hi;       // (c == INT32_MAX: row finished / absent)
            if (__ballot(mine) == 0ull) break;                   // the wave has nothing left in this window
#pragma unroll
            for (int i0 = 0; i0 < R; i0 += 4) {                  // four rows' gathers in flight per lane
                // No branches, and no selects the compiler could turn back into branches (it sinks a load whose value is only
                // used under a condition into that condition's block and drains it there, s_waitcnt vmcnt(0) per load): the
                // "taken" decision is a bit mask, every update and / or arithmetic on it.  The sched_barriers keep the three
                // stages apart: left alone the scheduler serialises load -> wait -> use row by row (lowest register pressure).
                float4 x[4];
                unsigned m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    m[u] = (unsigned)-(int)(c[i] < hi);                        // all ones: this row consumes its nonzero now
                    const int col = (c[i] & (int)m[u]) | (lo & (int)~m[u]);    // idle slots re-read the window's first row (an L2 hit)
                    x[u] = X4[(size_t)col * LPR + t];
                }
                __builtin_amdgcn_sched_barrier(0);
                int nc[4];
                float nv[4];
                unsigned more[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    cur[i] += (int)(m[u] & 1u);
                    rem[i] -= (int)(m[u] & 1u);
                    more[u] = (unsigned)-(int)(rem[i] > 0);
                    const int at = cur[i] & (int)more[u];                      // (always a valid address; re-reads are L1 hits)
                    nc[u] = __builtin_nontemporal_load(colidx + at);
                    nv[u] = __builtin_nontemporal_load(vals + at);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    const float4 f = f4_fma(v[i], x[u], acc[i]);
                    acc[i].x = w_sel(m[u], f.x, acc[i].x);
                    acc[i].y = w_sel(m[u], f.y, acc[i].y);
                    acc[i].z = w_sel(m[u], f.z, acc[i].z);
                    acc[i].w = w_sel(m[u], f.w, acc[i].w);
                    const int next_c = (nc[u] & (int)more[u]) | (INT32_MAX & (int)~more[u]);
                    c[i] = (next_c & (int)m[u]) | (c[i] & (int)~m[u]);
                    v[i] = w_sel(m[u], nv[u], v[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (rem[i] >= 0) store_row<LPR>(ep, base + i * SPW, t, acc[i]);
}

template <int LPR>
void launch(hipStream_t s, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
            const NarrowEpilogue& ep, int n_rows, int n_cols, int long_t, const int32_t* long_rows,
            const int32_t* long_chunk_ptr, int n_long, int n_chunks, float* partials, bool phased) {
    constexpr int CPB = 16 / LPR, SPB = 256 / LPR;
    const int chunk_blocks = (n_chunks + CPB - 1) / CPB;
    if (phased) {
        // long rows first (their chunk blocks alone: no row blocks), then the short rows in launches of as many rows as the chip
        // holds resident at once (4 waves per SIMD x R rows per sub-group), every launch walking the column windows in step
        constexpr int R = 8, ROWS_PER_WG = 4 * R * (64 / LPR), WG_RESIDENT = 1024;
        if (chunk_blocks > 0)
            hipLaunchKernelGGL(spmm_narrow_rows_kernel<LPR>, dim3(chunk_blocks), dim3(256), 0, s, rowptr, colidx, vals, X, ep, 0,
                               long_t, 1, long_rows, long_chunk_ptr, n_long, n_chunks, chunk_blocks, partials);
        const int blk_rows = (2 << 20) / (16 * LPR);                 // 2 MB of X rows per window
        const long cap = (long)WG_RESIDENT * ROWS_PER_WG;
        for (long lo = 0; lo < n_rows; lo += cap) {
            const int hi = (int)(lo + cap < n_rows ? lo + cap : n_rows);
            hipLaunchKernelGGL((spmm_narrow_phased_kernel<LPR, R>), dim3((hi - (int)lo + ROWS_PER_WG - 1) / ROWS_PER_WG), dim3(256),
                               0, s, rowptr, colidx, vals, X, ep, (int)lo, hi, long_t, n_cols, blk_rows);
        }
    } else {
        const int rows_per_group = n_rows <= (1 << 18) ? 1 : 4;
        const int blocks = (n_rows + SPB * rows_per_group - 1) / (SPB * rows_per_group);
        hipLaunchKernelGGL(spmm_narrow_rows_kernel<LPR>, dim3(blocks + chunk_blocks), dim3(256), 0, s, rowptr, colidx, vals, X,
                           ep, n_rows, long_t, rows_per_group, long_rows, long_chunk_ptr, n_long, n_chunks, chunk_blocks,
                           partials);
    }
    if (n_long > 0 && n_chunks > n_long)     // at least one row spans several chunks
        hipLaunchKernelGGL(spmm_narrow_long_reduce_kernel<LPR>, dim3((n_long + CPB - 1) / CPB), dim3(256), 0, s, long_rows,
                           long_chunk_ptr, n_long, partials, ep);
}

}  // namespace

int spmm_narrow_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, float* Y,
                       const float* Z, const float* acc_in, float* acc_out, int n_rows, int n_cols, int d, float alpha,
                       float beta, float acc_scale, int long_t, const int32_t* long_rows, const int32_t* long_chunk_ptr,
                       int n_long, int n_chunks, float* partials, bool cols_sorted, hipStream_t s) {
    const NarrowEpilogue ep{Z, Y, acc_in, acc_out, alpha, Z ? beta : 0.f, acc_scale};
    // in-step column windows: for graphs whose rows are column-sorted (the caller says so: locality is all that depends on it,
    // the sums are in CSR order either way) and whose X slice does not fit an XCD's L2 anyway
    const bool phased = cols_sorted && n_cols > 0 && (size_t)n_cols * d * 4 > ((size_t)8 << 20) && n_rows > (1 << 18);
#define MMREC_NARROW_CASE(D, L)                                                                                          \
    case D:                                                                                                              \
        launch<L>(s, rowptr, colidx, vals, X, ep, n_rows, n_cols, long_t, long_rows, long_chunk_ptr, n_long, n_chunks,   \
                  partials, phased);                                                                                     \
        break;
    switch (d) {
        MMREC_NARROW_CASE(8, 2) MMREC_NARROW_CASE(16, 4) MMREC_NARROW_CASE(32, 8)
        default:
            return MMREC_ERR_UNSUPPORTED;
    }
#undef MMREC_NARROW_CASE
    return (int)hipGetLastError();
}
