// CSR SpMM for the normalised user-item / item-item adjacency, d = 64 fp32.   (a5, a6)
//
// Roofline: HBM.  Algorithmic bytes per launch = 264 B per nonzero (4 col + 4 val + 256-B X row)
// + 260 B per output row (4 rowptr + 256-B Y row)   [BASELINE.md section 3].
//
// Mapping: an embedding row is 64 fp32 = 256 B = 16 lanes x float4.  One 16-lane group (a DPP
// "row") owns one matrix row, so a wave64 works on 4 matrix rows at once and every X-row gather is
// one 16-B-per-lane load (1 KiB per wave instruction, fully coalesced per matrix row).  The group
// reads 16 (col,val) pairs with one coalesced 64-B load each and broadcasts them lane by lane
// through the LDS crossbar (ds_bpermute), so index traffic is read exactly once.
//
// Load balance (power-law rows): rows longer than `long_t` are skipped by the row
// blocks and cut into MMREC_SPMM_CHUNK-nonzero chunks, one workgroup each in the same launch (16
// groups x 16-nonzero spans, LDS tree); partial rows go to a workspace and a second tiny launch
// sums them in chunk order.  Short serial chains also keep small cache-resident graphs (Amazon-Baby)
// from being bound by the latency of their longest row.  No float atomics:
// the per-row summation order is fixed and independent of the row partition (multi-GPU == 1 GPU).
#include "common.h"
#include "spmm_narrow.h"

// matrix rows per 16-lane group of a row block (tools/spmm_sweep.py overrides it)
#ifndef MMREC_SPMM_RPG
#define MMREC_SPMM_RPG(n_rows) ((n_rows) <= (1 << 18) ? 1 : 4)
#endif
// Measured and NOT in this file (logs under profiles/, code in the history): nontemporal loads of colidx / vals and nontemporal
// stores of Y change nothing, nontemporal X gathers cost 50 %, a hot / cold column split 60 % (r02_spmm_cache_policy_lab.log,
// r02_spmm_lab_hot_cold_nontemporal.log); walking the rows of a group together with their gathers prefetched is bit-identical
// and slower on every graph (r04_spmm_rows_batched_lab_*.log, last present at commit 3e827a9); LDS staging of the most
// popular X rows: tools/spmm_lds_hot_lab.hip, profiles/r05_spmm_lds_hot_lab.log.

namespace {

// Partial sums that another workgroup of the SAME launch reads (the last-arriver row finish below): written and read with
// agent-scope accesses (sc1: through the XCD's L2 to the device's coherence point), so that no L2 write-back /
// invalidate fence is needed -- an agent-scope release fence writes back the whole L2 (measured 2.3 x on a kernel that did
// one per wave).
__device__ __forceinline__ void st_coherent(float4* p, float4 v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_coherent(const float4* p) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float4*>(p));
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}

// Embedding rows are d = 64 * DCH floats (DCH = 1 for every d = 64 graph model; 4 and 6 for MMGCN's
// 256- and 384-wide modality layers): a 16-lane group walks a row in DCH chunks of 16 x float4.
struct RowEpilogue {
    const float* Z;
    float* Y;
    const float* acc_in;
    float* acc_out;
    float alpha, beta, acc_scale;
    // LayerGCN (d == 64 only): w = cos(y, ego row), scaled = w y, acc_out = acc_in + scaled (layergcn.py:131-135)
    const float* ego;
    float* scaled;
    float* w_out;
};

// LG: the LayerGCN epilogue, its own instantiation -- compiled into the common kernel its registers cost every launch occupancy
template <int DCH, bool LG>
__device__ __forceinline__ void store_row(const RowEpilogue& ep, int row, int lane16, const float4 (&sum)[DCH]) {
    if (DCH == 1 && LG) {   // the 16 lanes of the group hold the whole 64-float row: the row statistics are 4 DPP steps
        const size_t off = (size_t)row * 16 + lane16;
        const float4 y = f4_scale(ep.alpha, sum[0]);
        const float4 g = reinterpret_cast<const float4*>(ep.ego)[off];
        // the arithmetic of cos_scale_fwd_kernel, instruction for instruction (bit-identical results)
        const float dot = row16_sum(f4_dot(y, g));
        const float ne = fmaxf(sqrtf(row16_sum(f4_dot(y, y))), 1e-8f);
        const float ng = fmaxf(sqrtf(row16_sum(f4_dot(g, g))), 1e-8f);
        const float w = dot / (ne * ng);
        const float4 o = f4_scale(w, y);
        if (ep.Y) reinterpret_cast<float4*>(ep.Y)[off] = y;
        reinterpret_cast<float4*>(ep.scaled)[off] = o;
        if (lane16 == 0) ep.w_out[row] = w;
        if (ep.acc_out)
            reinterpret_cast<float4*>(ep.acc_out)[off] =
                ep.acc_in ? f4_add(reinterpret_cast<const float4*>(ep.acc_in)[off], o) : o;
        return;
    }
#pragma unroll
    for (int ch = 0; ch < DCH; ++ch) {
        const size_t off = (size_t)row * (16 * DCH) + ch * 16 + lane16;  // float4 index
        float4 y = f4_scale(ep.alpha, sum[ch]);
        if (ep.Z) y = f4_fma(ep.beta, reinterpret_cast<const float4*>(ep.Z)[off], y);
        if (ep.Y) reinterpret_cast<float4*>(ep.Y)[off] = y;
        if (ep.acc_out) {
            const float4 a = reinterpret_cast<const float4*>(ep.acc_in)[off];
            reinterpret_cast<float4*>(ep.acc_out)[off] = f4_scale(ep.acc_scale, f4_add(a, y));
        }
    }
}

// acc += sum_{k in [s,e)} vals[k] * X[colidx[k]]   for one 16-lane group (lane16 = float4 slot).
// All 16 lanes of a group run the same trip counts, so the shuffles only read active lanes.
template <int DCH>
__device__ __forceinline__ void gather_span(const int32_t* __restrict__ colidx,
                                            const float* __restrict__ vals,
                                            const float4* __restrict__ X4, int s, int e, int lane16,
                                            float4 (&acc)[DCH]) {
    constexpr int NB = DCH == 1 ? 8 : (DCH <= 2 ? 4 : (DCH <= 4 ? 2 : 1));  // nonzeros per step: ~8 loads in flight
    for (int base = s; base < e; base += 16) {
        const int k = base + lane16;
        int c = 0;
        float v = 0.f;
        if (k < e) {
            c = colidx[k];
            v = vals[k];
        }
        const int cnt = min(16, e - base);
        // up to 8 gathers in flight per group per step (latency hiding for long-ish rows on small,
        // cache-resident graphs); the tail predicate is uniform within the group
        for (int j0 = 0; j0 < cnt; j0 += NB) {
            float4 x[NB][DCH];
            float vv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int j = j0 + u;
                const int cj = __shfl(c, j, 16);
                vv[u] = __shfl(v, j, 16);
#pragma unroll
                for (int ch = 0; ch < DCH; ++ch)
                    x[u][ch] = (j < cnt) ? X4[(size_t)cj * (16 * DCH) + ch * 16 + lane16] : f4_zero();
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int ch = 0; ch < DCH; ++ch) acc[ch] = f4_fma(vv[u], x[u][ch], acc[ch]);
        }
    }
}

// Sum of the chunk partials [c0, c1) of one long row by a whole workgroup: group g sums chunks g, g + 16, ... in order, then
// a fixed-order LDS tree over the 16 groups (the heaviest C5 row has 280 chunks; a single sequential chain over them
// would cost ~110 us); group 0 finishes the row.  ONE order for both callers -- the reduce kernel and the last-arriver
// chunk block -- so the two forms give the same bits.
template <int DCH, bool COHERENT, bool LG>
__device__ __forceinline__ void reduce_long_row(const float* __restrict__ partials, int c0, int c1, int row,
                                                const RowEpilogue& ep, float4 (*red)[16 * DCH]) {
    const int lane16 = threadIdx.x & 15, g = threadIdx.x >> 4;
#pragma unroll
    for (int ch = 0; ch < DCH; ++ch) {
        float4 t = f4_zero();
        for (int c = c0 + g; c < c1; c += 16) {
            const float4* src = reinterpret_cast<const float4*>(partials) + (size_t)c * (16 * DCH) + ch * 16 + lane16;
            t = f4_add(t, COHERENT ? ld_coherent(src) : *src);
        }
        red[g][ch * 16 + lane16] = t;
    }
    __syncthreads();
    if (g == 0) {
        float4 r[DCH];
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) {
            r[ch] = red[0][ch * 16 + lane16];
#pragma unroll
            for (int k = 1; k < 16; ++k) r[ch] = f4_add(r[ch], red[k][ch * 16 + lane16]);
        }
        store_row<DCH, LG>(ep, row, lane16, r);
    }
}

// One launch covers both kinds of work: blocks [0, n_chunks) reduce one long-row chunk each (started
// first: they are the longest dependency chains), blocks [n_chunks, ...) process 64 short rows each.
template <int DCH, bool LG>
__global__ __launch_bounds__(256) void spmm_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
    const float* __restrict__ vals, const float* __restrict__ X, RowEpilogue ep, int n_rows,
    int long_t, int rows_per_group, const int32_t* __restrict__ long_rows,
    const int32_t* __restrict__ long_chunk_ptr, int n_long, int n_chunks,
    float* __restrict__ partials, int32_t* __restrict__ tickets) {
    __shared__ float4 red[16][16 * DCH];
    __shared__ int s_last;
    const int lane16 = threadIdx.x & 15;
    const int g = threadIdx.x >> 4;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    if ((int)blockIdx.x < n_chunks) {
        const int chunk = blockIdx.x;
        int lo = 0, hi = n_long;  // largest lo with long_chunk_ptr[lo] <= chunk
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (long_chunk_ptr[mid] <= chunk) lo = mid; else hi = mid;
        }
        const int row = long_rows[lo];
        const int cs = rowptr[row] + (chunk - long_chunk_ptr[lo]) * MMREC_SPMM_CHUNK;
        const int ce = min(cs + MMREC_SPMM_CHUNK, rowptr[row + 1]);
        float4 acc[DCH];
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) acc[ch] = f4_zero();
        for (int base = cs + g * 16; base < ce; base += 256)
            gather_span<DCH>(colidx, vals, X4, base, min(base + 16, ce), lane16, acc);
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) red[g][ch * 16 + lane16] = acc[ch];
        __syncthreads();
        if (g == 0) {
            float4 t[DCH];
#pragma unroll
            for (int ch = 0; ch < DCH; ++ch) {
                t[ch] = red[0][ch * 16 + lane16];
#pragma unroll
                for (int i = 1; i < 16; ++i) t[ch] = f4_add(t[ch], red[i][ch * 16 + lane16]);
            }
            if (long_chunk_ptr[lo + 1] - long_chunk_ptr[lo] == 1) {
                store_row<DCH, LG>(ep, row, lane16, t);  // the whole row fitted one chunk: done
            } else {
#pragma unroll
                for (int ch = 0; ch < DCH; ++ch) {
                    float4* dst = reinterpret_cast<float4*>(partials) + (size_t)chunk * (16 * DCH) + ch * 16 + lane16;
                    if (tickets) st_coherent(dst, t[ch]); else *dst = t[ch];
                }
            }
        }
        // `tickets` (small, latency-bound graphs): the row is finished HERE by the chunk block that arrives last, instead of
        // by a second launch (4.5 us of a 19 us Amazon-Baby layer for the 15 rows that span several chunks).  The partial
        // was written at agent scope; the wave that wrote it waits for the acknowledgement and takes a ticket; the block
        // holding the last ticket reads all partials at agent scope in the reduce kernel's order (same bits) and leaves
        // the ticket at zero for the next launch.
        const int c0 = long_chunk_ptr[lo], c1 = long_chunk_ptr[lo + 1];
        if (!tickets || c1 - c0 == 1) return;     // uniform
        if (threadIdx.x < 64) {                    // the wave of group 0
            // The hand-off is hardware ordering, not the HSA memory model's (which would ask for an agent-scope release:
            // the L2 write-back this scheme exists to avoid): the sc1 stores above are complete at the device's coherence
            // point when vmcnt reaches 0, and only then is the ticket taken.  The two signal fences pin that order for the
            // COMPILER (it must not sink a store below the wait or hoist the atomic above it); hip_ops.CsrGraph.checked()
            // re-zeroes the tickets after a failed launch; tests compare this form with the two-launch form bit for bit.
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);         // my partial is out (vmcnt(0))
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (threadIdx.x == 0)
                s_last = __hip_atomic_fetch_add(tickets + lo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c1 - c0 - 1;
        }
        __syncthreads();
        if (!s_last) return;                       // uniform
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        reduce_long_row<DCH, true, LG>(partials, c0, c1, row, ep, red);
        if (threadIdx.x == 0) __hip_atomic_store(tickets + lo, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int row0 = ((int)blockIdx.x - n_chunks) * 16 * rows_per_group + g;
#pragma unroll 1
    for (int i = 0; i < rows_per_group; ++i) {
        const int row = row0 + i * 16;
        if (row >= n_rows) break;
        const int s = rowptr[row], e = rowptr[row + 1];
        if (e - s > long_t) continue;  // handled by the chunk blocks
        float4 acc[DCH];
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) acc[ch] = f4_zero();
        gather_span<DCH>(colidx, vals, X4, s, e, lane16, acc);
        store_row<DCH, LG>(ep, row, lane16, acc);
    }
}

// One workgroup per long row that spans several chunks (the two-launch form: large graphs).
template <int DCH, bool LG>
__global__ __launch_bounds__(256) void spmm_long_reduce_kernel(
    const int32_t* __restrict__ long_rows, const int32_t* __restrict__ long_chunk_ptr, int n_long,
    const float* __restrict__ partials, RowEpilogue ep) {
    __shared__ float4 red[16][16 * DCH];
    const int i = blockIdx.x;
    const int c0 = long_chunk_ptr[i], c1 = long_chunk_ptr[i + 1];
    if (c1 - c0 <= 1) return;  // uniform for the block: single-chunk rows were finished by their chunk block
    reduce_long_row<DCH, false, LG>(partials, c0, c1, long_rows[i], ep, red);
}

template <int DCH, bool LG = false>
void launch_spmm(hipStream_t s, int blocks, int nch, const int32_t* rowptr, const int32_t* colidx,
                 const float* vals, const float* X, const RowEpilogue& ep, int n_rows, int long_t,
                 int rows_per_group, const int32_t* long_rows, const int32_t* long_chunk_ptr, int n_long,
                 float* partials, int32_t* tickets) {
    hipLaunchKernelGGL((spmm_rows_kernel<DCH, LG>), dim3(blocks + nch), dim3(256), 0, s, rowptr, colidx, vals, X,
                       ep, n_rows, long_t, rows_per_group, long_rows, long_chunk_ptr, n_long, nch, partials, tickets);
    if (!tickets && n_long > 0 && nch > n_long)  // at least one row spans several chunks
        hipLaunchKernelGGL((spmm_long_reduce_kernel<DCH, LG>), dim3(n_long), dim3(256), 0, s, long_rows,
                           long_chunk_ptr, n_long, partials, ep);
}

// ---- LayerGCN per-layer cosine re-weighting (layergcn.py:132-134) -------------------------------
__global__ __launch_bounds__(256) void cos_scale_fwd_kernel(const float* __restrict__ E,
                                                            const float* __restrict__ Ego,
                                                            float* __restrict__ Out,
                                                            float* __restrict__ W,
                                                            float* __restrict__ acc, int n_rows) {
    const int lane16 = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_rows) return;
    const size_t off = (size_t)row * 16 + lane16;
    const float4 e = reinterpret_cast<const float4*>(E)[off];
    const float4 g = reinterpret_cast<const float4*>(Ego)[off];
    const float dot = row16_sum(f4_dot(e, g));
    const float ne = fmaxf(sqrtf(row16_sum(f4_dot(e, e))), 1e-8f);
    const float ng = fmaxf(sqrtf(row16_sum(f4_dot(g, g))), 1e-8f);
    const float w = dot / (ne * ng);
    const float4 o = f4_scale(w, e);
    reinterpret_cast<float4*>(Out)[off] = o;
    if (lane16 == 0) W[row] = w;
    if (acc) reinterpret_cast<float4*>(acc)[off] = f4_add(reinterpret_cast<float4*>(acc)[off], o);
}

__global__ __launch_bounds__(256) void cos_scale_bwd_kernel(
    const float* __restrict__ dOut, const float* __restrict__ E, const float* __restrict__ Ego,
    const float* __restrict__ W, float* __restrict__ dE, float* __restrict__ dEgo, int n_rows) {
    const int lane16 = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_rows) return;
    const size_t off = (size_t)row * 16 + lane16;
    const float4 e = reinterpret_cast<const float4*>(E)[off];
    const float4 g = reinterpret_cast<const float4*>(Ego)[off];
    const float4 d = reinterpret_cast<const float4*>(dOut)[off];
    const float w = W[row];
    const float ne_raw = sqrtf(row16_sum(f4_dot(e, e)));
    const float ng_raw = sqrtf(row16_sum(f4_dot(g, g)));
    const float ne = fmaxf(ne_raw, 1e-8f), ng = fmaxf(ng_raw, 1e-8f);
    const float dw = row16_sum(f4_dot(d, e));
    const float inv = 1.0f / (ne * ng);
    // d w / d e = g/(ne ng) - w e / ne^2 (second term only while the norm is not clamped)
    const float ce = (ne_raw > 1e-8f) ? w / (ne * ne) : 0.f;
    const float cg = (ng_raw > 1e-8f) ? w / (ng * ng) : 0.f;
    float4 de = f4_scale(w, d);
    de = f4_fma(dw * inv, g, de);
    de = f4_fma(-dw * ce, e, de);
    reinterpret_cast<float4*>(dE)[off] = de;
    float4 dg = reinterpret_cast<float4*>(dEgo)[off];
    dg = f4_fma(dw * inv, e, dg);
    dg = f4_fma(-dw * cg, g, dg);
    reinterpret_cast<float4*>(dEgo)[off] = dg;
}

}  // namespace

extern "C" int mmrec_spmm_csr_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals,
                                  const float* X, float* Y, const float* Z, const float* acc_in,
                                  float* acc_out, int32_t n_rows, int32_t d, float alpha, float beta,
                                  float acc_scale, int32_t long_row_threshold,
                                  const int32_t* long_rows, const int32_t* long_chunk_ptr,
                                  int32_t n_long, int32_t n_chunks, float* partials, int32_t* long_tickets,
                                  mmrec_stream_t stream) {
    const bool narrow = d == 8 || d == 16 || d == 32;      // one feature slice of a 64-wide table (spmm_narrow.hip)
    if (!narrow && (d <= 0 || d % MMREC_EMB_DIM || d / MMREC_EMB_DIM > 6)) return MMREC_ERR_UNSUPPORTED;
    if (n_rows < 0 || n_long < 0 || n_chunks < 0 || long_row_threshold < 0) return MMREC_ERR_BAD_ARG;
    if (n_rows == 0) return 0;
    if (!rowptr || !X || (!Y && !acc_out)) return MMREC_ERR_BAD_ARG;
    if (acc_out && !acc_in) return MMREC_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !long_chunk_ptr || !partials || n_chunks <= 0))
        return MMREC_ERR_BAD_ARG;
    if (Y == X) return MMREC_ERR_BAD_ARG;  // other rows still gather from X
    if (narrow)
        return spmm_narrow_launch(rowptr, colidx, vals, X, Y, Z, acc_in, acc_out, n_rows, d, alpha, beta, acc_scale,
                                  n_long > 0 ? long_row_threshold : INT32_MAX, long_rows, long_chunk_ptr, n_long,
                                  n_long > 0 ? n_chunks : 0, partials, mmrec_stream(stream));
    RowEpilogue ep{Z, Y, acc_in, acc_out, alpha, Z ? beta : 0.f, acc_scale, nullptr, nullptr, nullptr};
    hipStream_t s = mmrec_stream(stream);
    // small (cache-resident, latency-bound) graphs: one row per 16-lane group; large graphs: four
    const int rows_per_group = MMREC_SPMM_RPG(n_rows);
    const int blocks = (n_rows + 16 * rows_per_group - 1) / (16 * rows_per_group);
    // without a plan every row goes through the row kernel
    const int long_t = n_long > 0 ? long_row_threshold : INT32_MAX;
    const int nch = n_long > 0 ? n_chunks : 0;
    // rows finished inside the launch (last-arriver) on the small, latency-bound graphs only: a large graph has
    // thousands of multi-chunk rows and is bandwidth bound; its second launch costs nothing measurable
    int32_t* tickets = (n_long > 0 && n_rows <= MMREC_SPMM_FUSED_REDUCE_MAX_ROWS) ? long_tickets : nullptr;
#define MMREC_SPMM_CASE(D)                                                                               \
    case D:                                                                                              \
        launch_spmm<D>(s, blocks, nch, rowptr, colidx, vals, X, ep, n_rows, long_t, rows_per_group,      \
                       long_rows, long_chunk_ptr, n_long, partials, tickets);                            \
        break;
    switch (d / MMREC_EMB_DIM) {
        MMREC_SPMM_CASE(1) MMREC_SPMM_CASE(2) MMREC_SPMM_CASE(3) MMREC_SPMM_CASE(4) MMREC_SPMM_CASE(5)
        MMREC_SPMM_CASE(6)
    }
#undef MMREC_SPMM_CASE
    MMREC_RETURN_LAUNCH_STATUS();
}

// One LayerGCN layer in ONE launch: y = A x, w = cos(y, ego) per row, scaled = w y, acc_out = acc_in + scaled
// (layergcn.py:131-135; SURVEY.md 8b `spmm_csr_f32_layergcn`).  The cosine needs the finished row, which the group that
// stores it holds in registers in all three places a row is finished (row blocks, single-chunk blocks, long-row reduce).
extern "C" int mmrec_spmm_rows_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
                                   const float* Z, int32_t z_compact, const int64_t* rows, int32_t n_list, int32_t d,
                                   int32_t long_row_threshold, float* Y, mmrec_stream_t stream) {
    if (d != 8 && d != 16 && d != 32 && d != 64) return MMREC_ERR_UNSUPPORTED;
    if (n_list < 0 || long_row_threshold < 0) return MMREC_ERR_BAD_ARG;
    if (n_list == 0) return 0;
    if (!rowptr || !X || !rows || !Y || Y == X) return MMREC_ERR_BAD_ARG;
    return spmm_pull_rows_launch(rowptr, colidx, vals, X, Z, z_compact, rows, n_list, d, long_row_threshold, Y, mmrec_stream(stream));
}

// ABI 12: the same for ANY listed row of a d = 64 graph -- rows of several chunks included (max_row_chunks = the largest
// number of MMREC_SPMM_CHUNK-nonzero chunks a row of the graph's plan spans; <= 480).
extern "C" int mmrec_spmm_rows_any_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X,
                                       const float* Z, int32_t z_compact, const int64_t* rows, int32_t n_list, int32_t d,
                                       int32_t long_row_threshold, int32_t max_row_chunks, float* Y, mmrec_stream_t stream) {
    if (d != 8 && d != 16 && d != 32 && d != 64) return MMREC_ERR_UNSUPPORTED;
    if (n_list < 0 || long_row_threshold < 0 || max_row_chunks < 1) return MMREC_ERR_BAD_ARG;
    if (n_list == 0) return 0;
    if (!rowptr || !X || !rows || !Y || Y == X) return MMREC_ERR_BAD_ARG;
    return spmm_pull_rows_launch(rowptr, colidx, vals, X, Z, z_compact, rows, n_list, d, long_row_threshold, Y, mmrec_stream(stream),
                                 max_row_chunks);
}

extern "C" int mmrec_spmm_push_rows_f32(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* G,
                                        float g_scale, const int64_t* rows, int32_t n_list, int32_t d, float* dX, float* dZ,
                                        mmrec_stream_t stream) {
    if (d != 8 && d != 16 && d != 32 && d != 64) return MMREC_ERR_UNSUPPORTED;
    if (n_list < 0) return MMREC_ERR_BAD_ARG;
    if (n_list == 0) return 0;
    if (!rowptr || !G || !rows || (!dX && !dZ)) return MMREC_ERR_BAD_ARG;
    return spmm_push_rows_launch(rowptr, colidx, vals, G, g_scale, rows, n_list, d, dX, dZ, mmrec_stream(stream));
}

extern "C" int mmrec_spmm_csr_f32_layergcn(const int32_t* rowptr, const int32_t* colidx, const float* vals,
                                           const float* X, float* Y, const float* ego, float* scaled, float* w,
                                           const float* acc_in, float* acc_out, int32_t n_rows, int32_t d,
                                           int32_t long_row_threshold, const int32_t* long_rows,
                                           const int32_t* long_chunk_ptr, int32_t n_long, int32_t n_chunks,
                                           float* partials, int32_t* long_tickets, mmrec_stream_t stream) {
    if (d != MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (n_rows < 0 || n_long < 0 || n_chunks < 0 || long_row_threshold < 0) return MMREC_ERR_BAD_ARG;
    if (n_rows == 0) return 0;
    if (!rowptr || !X || !ego || !scaled || !w) return MMREC_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !long_chunk_ptr || !partials || n_chunks <= 0)) return MMREC_ERR_BAD_ARG;
    if (Y == X || scaled == X || acc_out == X) return MMREC_ERR_BAD_ARG;  // other rows still gather from X
    RowEpilogue ep{nullptr, Y, acc_in, acc_out, 1.f, 0.f, 1.f, ego, scaled, w};
    hipStream_t s = mmrec_stream(stream);
    const int rows_per_group = MMREC_SPMM_RPG(n_rows);
    const int blocks = (n_rows + 16 * rows_per_group - 1) / (16 * rows_per_group);
    const int long_t = n_long > 0 ? long_row_threshold : INT32_MAX;
    const int nch = n_long > 0 ? n_chunks : 0;
    int32_t* tickets = (n_long > 0 && n_rows <= MMREC_SPMM_FUSED_REDUCE_MAX_ROWS) ? long_tickets : nullptr;
    launch_spmm<1, true>(s, blocks, nch, rowptr, colidx, vals, X, ep, n_rows, long_t, rows_per_group, long_rows,
                         long_chunk_ptr, n_long, partials, tickets);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_spmm_plan_count(const int32_t* rowptr_host, int32_t n_rows,
                                     int32_t long_row_threshold, int32_t* n_long, int32_t* n_chunks) {
    if (!rowptr_host || !n_long || !n_chunks || n_rows < 0) return MMREC_ERR_BAD_ARG;
    int32_t nl = 0, nc = 0;
    for (int32_t r = 0; r < n_rows; ++r) {
        const int32_t deg = rowptr_host[r + 1] - rowptr_host[r];
        if (deg > long_row_threshold) {
            ++nl;
            nc += (deg + MMREC_SPMM_CHUNK - 1) / MMREC_SPMM_CHUNK;
        }
    }
    *n_long = nl;
    *n_chunks = nc;
    return 0;
}

extern "C" int mmrec_spmm_plan_fill(const int32_t* rowptr_host, int32_t n_rows,
                                    int32_t long_row_threshold, int32_t* long_rows,
                                    int32_t* long_chunk_ptr) {
    if (!rowptr_host || !long_rows || !long_chunk_ptr || n_rows < 0) return MMREC_ERR_BAD_ARG;
    int32_t nl = 0, nc = 0;
    for (int32_t r = 0; r < n_rows; ++r) {
        const int32_t deg = rowptr_host[r + 1] - rowptr_host[r];
        if (deg > long_row_threshold) {
            long_rows[nl] = r;
            long_chunk_ptr[nl] = nc;
            ++nl;
            nc += (deg + MMREC_SPMM_CHUNK - 1) / MMREC_SPMM_CHUNK;
        }
    }
    long_chunk_ptr[nl] = nc;
    return 0;
}

extern "C" int mmrec_cos_scale_fwd_f32(const float* E, const float* Ego, float* Out, float* w,
                                       float* acc, int32_t n_rows, int32_t d, mmrec_stream_t stream) {
    if (d != MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (n_rows < 0 || (n_rows > 0 && (!E || !Ego || !Out || !w))) return MMREC_ERR_BAD_ARG;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(cos_scale_fwd_kernel, dim3((n_rows + 15) / 16), dim3(256), 0,
                       mmrec_stream(stream), E, Ego, Out, w, acc, n_rows);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_cos_scale_bwd_f32(const float* dOut, const float* E, const float* Ego,
                                       const float* w, float* dE, float* dEgo_accum, int32_t n_rows,
                                       int32_t d, mmrec_stream_t stream) {
    if (d != MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (n_rows < 0 || (n_rows > 0 && (!dOut || !E || !Ego || !w || !dE || !dEgo_accum)))
        return MMREC_ERR_BAD_ARG;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(cos_scale_bwd_kernel, dim3((n_rows + 15) / 16), dim3(256), 0,
                       mmrec_stream(stream), dOut, E, Ego, w, dE, dEgo_accum, n_rows);
    MMREC_RETURN_LAUNCH_STATUS();
}
