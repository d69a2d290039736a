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

// ---- listed rows only (the training step of a model that consumes a propagated table at its BATCH rows: FREEDOM's item-item
// layer, freedom.py:173-177 + 197-199 -- 4096 of 500,000 rows at config 5) ---------------------------------------------------
// Y[i] = (A X)[rows[i]] (+ Z[rows[i]], or + Z[i] for a compact Z), one LPR-lane sub-group per listed row, in the order of the full launch: a row of at
// most long_t nonzeros is one sequential chain (gather_span); a longer row that fits ONE chunk is summed as its chunk block
// does -- 16 virtual groups over spans of 16 nonzeros at stride 256, then the fixed-order sum over the groups -- so the listed
// rows carry the full launch's bits (rows spanning several chunks are not served: the caller keeps the full launch).
template <int LPR>
__global__ __launch_bounds__(256) void spmm_pull_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ X, const float* __restrict__ Z, int z_compact, const int64_t* __restrict__ rows, int n_list,
    int long_t, float* __restrict__ Y, int skip_multi_chunk) {
    const int t = threadIdx.x % LPR;
    const int i = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (i >= n_list) return;                                   // uniform within the sub-group
    const int row = (int)rows[i];
    const int s = rowptr[row], e = rowptr[row + 1];
    if (skip_multi_chunk && e - s > long_t && e - s > MMREC_SPMM_CHUNK) return;      // served by spmm_pull_long_rows_kernel
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4 sum = f4_zero();
    if (e - s <= long_t) {
        gather_span<LPR>(colidx, vals, X4, s, e, t, sum);
    } else {
#pragma unroll 1
        for (int g = 0; g < 16; ++g) {
            float4 acc = f4_zero();
            for (int base = s + g * 16; base < e; base += 256) gather_span<LPR>(colidx, vals, X4, base, min(base + 16, e), t, acc);
            sum = g == 0 ? acc : f4_add(sum, acc);
        }
    }
    float4 y = f4_scale(1.f, sum);
    if (Z) y = f4_fma(1.f, reinterpret_cast<const float4*>(Z)[(size_t)(z_compact ? i : row) * LPR + t], y);
    reinterpret_cast<float4*>(Y)[(size_t)i * LPR + t] = y;
}

// Listed rows that span SEVERAL chunks (round 6: the user-item graph of a config-5 training step -- popular items are in every
// batch), d = 64: one workgroup per listed row walks the row's chunks one after the other, each summed as its chunk block of
// the full launch sums it (16 groups over spans of 16 nonzeros at stride 256, then the fixed-order sum over the groups), the
// chunk partials kept in LDS and combined in reduce_long_row's order (group g: chunks g, g + 16, ... from zero; then the
// groups in order) -- the full launch's bits for the row.  Rows of one chunk or less return at once (spmm_pull_rows_kernel
// serves them).  Dynamic LDS: max_chunks float4[16] partials.
__global__ __launch_bounds__(256) void spmm_pull_long_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ X, const float* __restrict__ Z, int z_compact, const int64_t* __restrict__ rows, int n_list,
    int long_t, int max_chunks, float* __restrict__ Y) {
    extern __shared__ float4 s_part[];            // [max_chunks][16]
    __shared__ float4 red[16][16];
    const int i = blockIdx.x;
    const int row = (int)rows[i];
    const int s = rowptr[row], e = rowptr[row + 1];
    if (e - s <= long_t || e - s <= MMREC_SPMM_CHUNK) return;       // uniform
    const int t = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n_ch = (e - s + MMREC_SPMM_CHUNK - 1) / MMREC_SPMM_CHUNK;
    if (n_ch > max_chunks) return;                                  // (the host sized max_chunks from the graph's plan)
    const float4* X4 = reinterpret_cast<const float4*>(X);
    for (int c = 0; c < n_ch; ++c) {
        const int cs = s + c * MMREC_SPMM_CHUNK, ce = min(cs + MMREC_SPMM_CHUNK, e);
        float4 acc = f4_zero();
        for (int base = cs + g * 16; base < ce; base += 256) gather_span<16>(colidx, vals, X4, base, min(base + 16, ce), t, acc);
        red[g][t] = acc;
        __syncthreads();
        if (g == 0) {
            float4 r = red[0][t];
#pragma unroll
            for (int k = 1; k < 16; ++k) r = f4_add(r, red[k][t]);
            s_part[c * 16 + t] = r;
        }
        __syncthreads();
    }
    float4 tsum = f4_zero();
    for (int c = g; c < n_ch; c += 16) tsum = f4_add(tsum, s_part[c * 16 + t]);
    red[g][t] = tsum;
    __syncthreads();
    if (g == 0) {
        float4 r = red[0][t];
#pragma unroll
        for (int k = 1; k < 16; ++k) r = f4_add(r, red[k][t]);
        float4 y = f4_scale(1.f, r);
        if (Z) y = f4_fma(1.f, reinterpret_cast<const float4*>(Z)[(size_t)(z_compact ? i : row) * 16 + t], y);
        reinterpret_cast<float4*>(Y)[(size_t)i * 16 + t] = y;
    }
}

// The transpose-free backward of the above: dX[c] += A[r, c] * G[i] for every nonzero (r, c) of the listed rows r = rows[i]
// (and dZ[r] += G[i]); fp32 atomics -- the order of the sums differs from the full pull launch in the last ulp, like the
// sampled-scoring backward's scatters (callers in `hip_deterministic` mode keep the full launch).
template <int LPR>
__global__ __launch_bounds__(256) void spmm_push_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ G, float g_scale, const int64_t* __restrict__ rows, int n_list, float* __restrict__ dX,
    float* __restrict__ dZ) {
    // one WORKGROUP per listed row, its 256 / LPR sub-groups striding over the row's nonzeros: a batch holds popular items
    // whose rows have thousands of nonzeros (one sub-group alone would take a millisecond on such a row)
    constexpr int SG = 256 / LPR;
    const int t = threadIdx.x % LPR, sg = threadIdx.x / LPR;
    const int i = blockIdx.x;
    const int row = (int)rows[i];
    const float4 g = f4_scale(g_scale, reinterpret_cast<const float4*>(G)[(size_t)i * LPR + t]);
    auto add4 = [&](float* base, float4 v) {
        unsafeAtomicAdd(base + 0, v.x);
        unsafeAtomicAdd(base + 1, v.y);
        unsafeAtomicAdd(base + 2, v.z);
        unsafeAtomicAdd(base + 3, v.w);
    };
    if (dZ && sg == 0) add4(dZ + ((size_t)row * LPR + t) * 4, g);
    if (!dX) return;
    const int s = rowptr[row], e = rowptr[row + 1];
    for (int k = s + sg; k < e; k += SG) add4(dX + ((size_t)colidx[k] * LPR + t) * 4, f4_scale(vals[k], g));
}

template <int LPR>
void launch(hipStream_t s, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
            const NarrowEpilogue& ep, int n_rows, int long_t, const int32_t* long_rows, const int32_t* long_chunk_ptr,
            int n_long, int n_chunks, float* partials) {
    constexpr int CPB = 16 / LPR, SPB = 256 / LPR;
    const int rows_per_group = n_rows <= (1 << 18) ? 1 : 4;
    const int blocks = (n_rows + SPB * rows_per_group - 1) / (SPB * rows_per_group);
    const int chunk_blocks = (n_chunks + CPB - 1) / CPB;
    hipLaunchKernelGGL(spmm_narrow_rows_kernel<LPR>, dim3(blocks + chunk_blocks), dim3(256), 0, s, rowptr, colidx, vals, X,
                       ep, n_rows, long_t, rows_per_group, long_rows, long_chunk_ptr, n_long, n_chunks, chunk_blocks,
                       partials);
    if (n_long > 0 && n_chunks > n_long)     // at least one row spans several chunks
        hipLaunchKernelGGL(spmm_narrow_long_reduce_kernel<LPR>, dim3((n_long + CPB - 1) / CPB), dim3(256), 0, s, long_rows,
                           long_chunk_ptr, n_long, partials, ep);
}

}  // namespace

int spmm_narrow_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, float* Y,
                       const float* Z, const float* acc_in, float* acc_out, int n_rows, int d, float alpha, float beta,
                       float acc_scale, int long_t, const int32_t* long_rows, const int32_t* long_chunk_ptr, int n_long,
                       int n_chunks, float* partials, hipStream_t s) {
    const NarrowEpilogue ep{Z, Y, acc_in, acc_out, alpha, Z ? beta : 0.f, acc_scale};
    switch (d) {
        case 8:
            launch<2>(s, rowptr, colidx, vals, X, ep, n_rows, long_t, long_rows, long_chunk_ptr, n_long, n_chunks, partials);
            break;
        case 16:
            launch<4>(s, rowptr, colidx, vals, X, ep, n_rows, long_t, long_rows, long_chunk_ptr, n_long, n_chunks, partials);
            break;
        case 32:
            launch<8>(s, rowptr, colidx, vals, X, ep, n_rows, long_t, long_rows, long_chunk_ptr, n_long, n_chunks, partials);
            break;
        default:
            return MMREC_ERR_UNSUPPORTED;
    }
    return (int)hipGetLastError();
}

int spmm_pull_rows_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, const float* Z,
                          int z_compact, const int64_t* rows, int n_list, int d, int long_t, float* Y, hipStream_t s,
                          int max_chunks) {
    // max_chunks > 1: the graph has rows of several chunks; the listed ones among them get a workgroup each (d = 64 only)
    const int skip = max_chunks > 1 ? 1 : 0;
    if (skip && d != 64) return MMREC_ERR_UNSUPPORTED;
    if (skip && (size_t)max_chunks * 256 > 120 * 1024) return MMREC_ERR_UNSUPPORTED;      // the row's partials live in LDS
#define MMREC_PULL(L)                                                                                                     \
    hipLaunchKernelGGL(spmm_pull_rows_kernel<L>, dim3((n_list + 256 / L - 1) / (256 / L)), dim3(256), 0, s, rowptr, colidx, vals, \
                       X, Z, z_compact, rows, n_list, long_t, Y, skip)
    switch (d) {
        case 8: MMREC_PULL(2); break;
        case 16: MMREC_PULL(4); break;
        case 32: MMREC_PULL(8); break;
        case 64: MMREC_PULL(16); break;
        default: return MMREC_ERR_UNSUPPORTED;
    }
#undef MMREC_PULL
    if (skip) {
        const size_t lds = (size_t)max_chunks * 256;
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_pull_long_rows_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(spmm_pull_long_rows_kernel, dim3(n_list), dim3(256), lds, s, rowptr, colidx, vals, X, Z, z_compact, rows,
                           n_list, long_t, max_chunks, Y);
    }
    return (int)hipGetLastError();
}

int spmm_push_rows_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* G, float g_scale,
                          const int64_t* rows, int n_list, int d, float* dX, float* dZ, hipStream_t s) {
#define MMREC_PUSH(L)                                                                                                     \
    hipLaunchKernelGGL(spmm_push_rows_kernel<L>, dim3(n_list), dim3(256), 0, s, rowptr, colidx, vals, G, g_scale, rows, n_list, \
                       dX, dZ)
    switch (d) {
        case 8: MMREC_PUSH(2); break;
        case 16: MMREC_PUSH(4); break;
        case 32: MMREC_PUSH(8); break;
        case 64: MMREC_PUSH(16); break;
        default: return MMREC_ERR_UNSUPPORTED;
    }
#undef MMREC_PUSH
    return (int)hipGetLastError();
}
