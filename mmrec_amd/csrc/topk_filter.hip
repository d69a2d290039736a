// Score + mask + top-K for kd == 64 (every full-sort evaluation, trainer.py:304-309) WITHOUT writing the
// [n_query, n_cand] score block: an fp16 FILTER on the matrix cores, then exact fp32 refinement.
//
// Why: the materialised path (topk.hip) is bound twice by the 4*nq*nc-byte score block (written by an
// fp32-MFMA GEMM at 46 % matrix utilisation, swept again by the select kernel): 0.41 ms on the Amazon-Baby
// evaluation.  fp32-input MFMA is 1/16 of the 16-bit rate AND shares the vector pipe, so selection cannot be
// fused into it; fp16 MFMA has neither problem.
//
// How:
//  prep   candidates are CENTRED (c' = c - mean row: a per-query constant shift of all scores, no ranking
//         changes, but the common component LightGCN-smoothed embeddings share no longer inflates the norms the
//         error bound is stated in); every QUERY row is scaled by its own power of two (thresholds are per query,
//         so a per-query scale changes no ranking; one global scale would push a query whose norm is 2^-30 of the
//         largest one into fp16 subnormals, where the relative bound below no longer holds), the centred candidates
//         by one power of two, both into fp16's normal range, and rounded to fp16 (unit roundoff u = 2^-11; an
//         element that still lands below 2^-14 is rounded with ABSOLUTE error <= 2^-25).  Approximate score = fp16 x
//         fp16 products accumulated in fp32 on v_mfma_f32_32x32x16_f16:  |approx - exact| <= (2u + u^2) sum|q_i||c'_i|
//         + 2^-25 (sum|q_i| + sum|c'_i|) + accumulation rounding <= eps_q := 1.0e-3 |q| max|c'| + 2^-22 (|q| + max|c'|)
//         (Cauchy-Schwarz, sum|x_i| <= 8 |x|; a worst-case bound for ANY input, tests/test_host_logic.py).
//  pass 1 approximate scores of every (query, candidate), kept only as 32 running maxima per query and
//         candidate range (lane = query: a group is one accumulator register of one half-wave) ->
//         n_groups = 32 * ranges maxima per query.
//  bound  B = (k + m)-th largest group maximum, m = masked items of the query (k + m groups reach B, at
//         most m of them through a masked item): every true top-k score is >= B - eps, so its approximate
//         score is >= B - 2 eps =: thr.   (k + m)-th largest by bisection on the monotone integer image.
//  pass 2 the same products again; one pass / fail BIT per score (score >= thr), 64 bits per lane and four
//         stages go to memory as one 8-byte store -> nc / 8 bytes per query instead of 4 nc.
//  final  one wave per query: decode the bits (~60-70 ids), drop masked ids (binary search in the LDS-staged
//         mask list), EXACT fp32 scores of the survivors from the original Q and C (16 lanes per candidate row,
//         fixed summation tree), bitonic sort (score desc, id asc), cut to k.
//  slow   queries the filter cannot serve (k + m > n_groups, fewer than k unmasked candidates, more than 256
//         survivors: massive ties, adversarial inputs) are queued on the device and served by persistent waves with
//         a streaming exact top-k over all candidates (same summation tree, masked items at -1e10 exactly like the
//         reference).  Correctness never depends on the filter being selective; no host synchronisation.
// A stage = 64 candidates (two 32-row tiles, 128 B of fp16 per row), staged once per 256-query workgroup in LDS
// (double buffered, 16-B chunks swizzled by (row >> 1) & 7 so both the 256-thread fill and the per-wave
// ds_read_b128 operand reads are bank-conflict free); each wave keeps two 32-query fragments in registers: 16 MFMAs
// (four independent accumulator chains) per stage and wave.
#include "topk_filter.h"
#include "topk_sort.h"
#include <limits.h>
#include <type_traits>

// tools/prof_topk_filter.py builds this file with an ablation mask (the library only ever uses 0):
// bit0 no MFMAs, bit1 operands not read from LDS, bit2 candidate tiles not loaded from memory, bit3 no epilogue,
// bit4 no workgroup barriers, bit5 stop after pass 2 (outputs unwritten)
#ifndef MMREC_TF_PROBE
#define MMREC_TF_PROBE 0
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float acc16;

constexpr int F_QWG = 256;     // queries per workgroup: 4 waves x 2 fragments x 32
constexpr int F_CAPQ = 256;    // survivor slots per query over all ranges (512 where pass 1 looks at every second stage)
#ifndef MMREC_TF_P1S       // pass-1 stage stride: 0 = the default of filter_plan, n = forced (tools/prof_topk_variants.py)
#define MMREC_TF_P1S 0
#endif
#ifndef MMREC_TF_MINR      // tools/prof_topk_ranges.py sweeps the number of candidate ranges
#define MMREC_TF_MINR 8
#define MMREC_TF_MAXR 16
#endif
constexpr int F_MINR = MMREC_TF_MINR, F_MAXR = MMREC_TF_MAXR;   // candidate ranges: 256..512 group maxima per query
constexpr int F_SLOW_WAVES = 8;   // waves sharing one query of the slow queue
constexpr int F_SLOW_MASK_LDS = 4096;   // mask entries of such a query staged in LDS
constexpr int F_SLOW_PARTS = 4096;      // (query, split) partial top-k lists of the split slow path (2 MiB)
constexpr int F_SLOW_SPLIT_NC = 65536;  // from this many candidates on a flagged query is split over 16 workgroups
constexpr int F_MIN_NC = 4096;    // below: the materialised path is as fast (fixed launch costs)
constexpr int F_SPARSE_NC = 32768;   // from this many candidates on pass 2 appends its NON-ZERO 64-bit words to a list
constexpr int F_WCAP = 256;          // ... (word index, word) entries, each >= 1 survivor; the final kernel ranks queries of up to
                                     // this many words (and survivors),
constexpr int F_WCAP2 = 512;         // ... twice as many where pass 1 subsamples (filter_plan): ~2 x the survivors
constexpr int F_P1S2_NC = 131072;    // from this many candidates on pass 1 walks every second stage
constexpr int F_WLIST_X = 2;         // a query's list holds F_WLIST_X x as many words as the final kernel has survivor slots: the
                                     // overflow kernel ranks the queries in between (heavy users, closely packed scores, ties)
constexpr int F_OVER_IDS = 4096;     // survivors of ONE overflowing query the overflow kernel ranks (16 KiB of LDS)
#ifndef MMREC_TF_FINAL_ROWS
#define MMREC_TF_FINAL_ROWS 8
#endif
constexpr int F_FINAL_ROWS = MMREC_TF_FINAL_ROWS;   // candidate rows in flight per lane in the final kernel's exact rescoring
constexpr int F_PF = 4;        // 64-candidate stages in flight per workgroup (register ring)
constexpr int F_MASK_LDS = 512; // mask entries per query staged in LDS by the final kernel (>= F_MAXR * 32)

__device__ __forceinline__ unsigned f2key(float f) {   // monotone: a < b  <=>  key(a) < key(b)
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// The candidate statistics block (1 KiB at the head of the prepared buffer): stats[0 .. kd) = column sums of C, the two
// monotone keys at fixed word offsets: ST_CMAX = key(max |c_ij|), ST_NMAX = key(max norm of a converted, centred row).
// (Queries are scaled row by row: nothing global to collect for them.)  kd = 64 or 128 floats per row.
constexpr int ST_WORDS = 256, ST_CMAX = 252, ST_NMAX = 253;
// Large candidate sets (>= F_P1S2_NC) CLIP the few rows of outlying norm: eps is proportional to the largest stored row norm,
// and ONE low-degree item of 20 x the median norm (config 5 after three propagation layers: 2.5 against 0.11) doubled every
// query's survivors.  The <= F_OUT_CAP rows whose norm lies in the topmost occupied bins of a norm histogram (bins = exponent
// + 3 mantissa bits; tau = lower edge of the lowest such bin) are stored as f c', f = tau / |c'| < 1, and listed at ST_OUT0:
//   * pass 1: a clipped row's approximate score s~ is within eps(tau) of f s.  Where the bound T >= eps (checked per query;
//     otherwise the query goes to the slow queue) every group maximum >= T of a clipped row has f s >= 0, so s >= f s >= s~ -
//     eps: still a lower bound of a real candidate's score;
//   * pass 2 may miss a clipped row (its stored score is too small), so the final kernel ALWAYS rescores the listed rows.
//   * the fp32 rounding of the EXACT scores (the 4e-6 term of eps) keeps the unclipped largest norm (ST_NMAX0).
constexpr int F_OUT_CAP = 32, ST_NOUT = 200, ST_OUT0 = 201, ST_TAU = 240, ST_NMAX0 = 241, F_NHIST = 2048;
// Grid: the 128-row slabs of C.
__global__ __launch_bounds__(256) void filter_stats_kernel(const float* __restrict__ C, int nc, int kd4, float* __restrict__ stats) {
    __shared__ float4 s_sum[16][16];
    __shared__ float s_mx[4];
    const float* X = C;
    const int n = nc, r0 = blockIdx.x * 128;
    const int sub = threadIdx.x & 15, rr = threadIdx.x >> 4;
    float mx = 0.f;
    for (int cc = sub; cc < kd4; cc += 16) {      // 16 float4 columns per sweep (one for kd = 64, two for 128)
        float4 v[8];   // the thread's 8 rows, all loads in flight at once
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + rr + 16 * j;
            v[j] = r < n ? reinterpret_cast<const float4*>(X)[(size_t)r * kd4 + cc] : f4_zero();
        }
        float4 sum = f4_zero();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sum = f4_add(sum, v[j]);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
        }
        __syncthreads();                           // (s_sum of the previous sweep has been read)
        s_sum[rr][sub] = sum;
        __syncthreads();
        if (threadIdx.x < 16) {
            float4 t = f4_zero();
            for (int j = 0; j < 16; ++j) t = f4_add(t, s_sum[j][threadIdx.x]);
            const int c0 = 4 * (cc - sub + threadIdx.x);
            atomicAdd(stats + c0 + 0, t.x);
            atomicAdd(stats + c0 + 1, t.y);
            atomicAdd(stats + c0 + 2, t.z);
            atomicAdd(stats + c0 + 3, t.w);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(stats) + ST_CMAX, f2key(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
}

// power of two that brings values of magnitude <= mx (mx > 0) below 2^13: products of two such numbers summed
// over 64 terms stay far inside fp32, and fp16 (max 65504) holds every element
__device__ __forceinline__ float fp16_scale(float mx) {
    if (!(mx > 0.f)) return 1.f;
    int ex;
    frexpf(mx, &ex);            // mx = f * 2^ex, f in [0.5, 1)
    return ldexpf(1.f, min(13 - ex, 120));   // denormal-sized inputs: the scale itself must stay finite
}

// X [n][kd] fp32 -> fp16(scale * (x - centre)), 8 halves per 16-B chunk, natural k order; rows >= n are zero.  W = kd / 8
// threads per row (8 or 16 consecutive lanes).  Queries: Xs[row][W chunks].  CAND: centre = column mean, *maxnorm_key = max
// row norm of the converted rows, and the TILE layout the pass kernels stage: 64-candidate stage t, 64-column block kb ->
// tile t * KB + kb = 64 rows x 8 chunks contiguous (8 KB), so a pass walks kd = 128 as twice as many 64-wide tiles.
template <bool CAND, int W>
__global__ __launch_bounds__(256) void filter_convert_kernel(const float* __restrict__ X, int n, int n_pad,
                                                            const float* __restrict__ stats, uint4* __restrict__ Xs,
                                                            float* __restrict__ norm, unsigned* __restrict__ maxnorm_key,
                                                            int* __restrict__ zero_one, int* __restrict__ zero_rows) {
    constexpr int KB = W / 8;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t / W, ch = t % W;   // grid covers n_pad rows exactly (n_pad % 64 == 0)
    if (!CAND) {   // the call's counters, zeroed on the way (one int, and one int per real row)
        if (t < 2 && zero_one) zero_one[t] = 0;      // slow-queue length, overflow-queue length
        if (ch == 0 && row < n && zero_rows) zero_rows[row] = 0;
    }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (row < n) {
        const float4 a = reinterpret_cast<const float4*>(X)[(size_t)row * (2 * W) + ch * 2];
        const float4 b = reinterpret_cast<const float4*>(X)[(size_t)row * (2 * W) + ch * 2 + 1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        if (CAND) {
            const float inv = 1.f / (float)n;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] -= stats[ch * 8 + j] * inv;
        }
    }
    float scale;
    if (CAND) {   // one scale for all candidates: |c - mean| <= 2 max|c|
        scale = fp16_scale(2.f * key2f(reinterpret_cast<const unsigned*>(stats)[ST_CMAX]));
    } else {      // a query row's own scale (the W threads of a row are W consecutive lanes)
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(x[j]));
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, W));
        scale = fp16_scale(mx);
    }
    unsigned hb[8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hv = (_Float16)(x[j] * scale);   // round to nearest even
        hb[j] = (unsigned)__builtin_bit_cast(unsigned short, hv);
        const float back = (float)hv;
        ss = fmaf(back, back, ss);
    }
    const size_t dst = CAND ? (((size_t)(row >> 6) * KB + (ch >> 3)) * 64 + (row & 63)) * 8 + (ch & 7) : (size_t)row * W + ch;
    Xs[dst] = make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, W);
    // norms of the ROUNDED rows, inflated by the rounding (1 + 2^-11) and by sqrt's own error
    const float nrm = sqrtf(ss) * 1.0005f;
    if (ch == 0 && (!CAND || norm)) norm[row] = nrm;     // candidates: only where the clipping below wants them
    if (CAND) {   // one atomic per workgroup (n_pad * W is a multiple of 256: no partial workgroups)
        __shared__ float s_mx[4];
        float mx = nrm;
#pragma unroll
        for (int o = 32; o >= W; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(maxnorm_key, f2key(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
    }
}

// histogram of the converted candidate rows' norms (bin = float bits >> 20), 4096 rows per workgroup
__global__ __launch_bounds__(256) void filter_norm_hist_kernel(const float* __restrict__ cnorm, int nc, int* __restrict__ hist) {
    __shared__ int s_h[F_NHIST];
    for (int b = threadIdx.x; b < F_NHIST; b += 256) s_h[b] = 0;
    __syncthreads();
    const int r0 = blockIdx.x * 4096;
    for (int j = threadIdx.x; j < 4096; j += 256)
        if (r0 + j < nc) atomicAdd(&s_h[__float_as_uint(cnorm[r0 + j]) >> 20], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < F_NHIST; b += 256)
        if (s_h[b] != 0) atomicAdd(hist + b, s_h[b]);
}

// tau from the histogram (every workgroup computes it for itself), then the workgroup's 4096 rows: a row of norm >= tau is
// listed and re-converted as f (c - mean) from the fp32 source (ONE rounding, like every other row).  W = kd / 8.
template <int W>
__global__ __launch_bounds__(256) void filter_clip_kernel(const float* __restrict__ C, int nc, const float* __restrict__ cnorm,
                                                         const int* __restrict__ hist, float* __restrict__ stats,
                                                         uint4* __restrict__ Cs) {
    constexpr int KB = W / 8;
    __shared__ int s_part[256];
    __shared__ int s_b;
    int part = 0;
    for (int j = 0; j < 8; ++j) part += hist[threadIdx.x * 8 + j];
    s_part[threadIdx.x] = part;
    __syncthreads();
    if (threadIdx.x == 0) {      // lowest bin b with #{rows in bins >= b} <= F_OUT_CAP (empty bins extend it downwards)
        int cum = 0, b = F_NHIST, p = 255;
        for (; p >= 0 && cum + s_part[p] <= F_OUT_CAP; --p) { cum += s_part[p]; b = 8 * p; }
        if (p >= 0)
            for (int j = 8 * p + 7; j >= 8 * p && cum + hist[j] <= F_OUT_CAP; --j) { cum += hist[j]; b = j; }
        s_b = cum > 0 ? b : F_NHIST;
    }
    __syncthreads();
    const int b = s_b;
    if (b >= F_NHIST) return;                                  // nothing stands out: no clipping
    const float tau = __uint_as_float((unsigned)b << 20);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[ST_TAU] = tau;
        reinterpret_cast<unsigned*>(stats)[ST_NMAX0] = reinterpret_cast<const unsigned*>(stats)[ST_NMAX];
        reinterpret_cast<unsigned*>(stats)[ST_NMAX] = f2key(tau);       // every stored row norm is <= tau now
    }
    const float inv = 1.f / (float)nc;
    const float scale = fp16_scale(2.f * key2f(reinterpret_cast<const unsigned*>(stats)[ST_CMAX]));
    for (int j = threadIdx.x; j < 4096; j += 256) {
        const int row = blockIdx.x * 4096 + j;
        if (row >= nc) break;
        const float nrm = cnorm[row];
        if (!(nrm >= tau)) continue;
        const int slot = atomicAdd(reinterpret_cast<int*>(stats) + ST_NOUT, 1);
        reinterpret_cast<int*>(stats)[ST_OUT0 + slot] = row;             // slot < F_OUT_CAP by the choice of tau
        const float f = scale * (tau * (1.f - 1.f / 512.f) / nrm);       // rounded norm <= tau (nrm already carries 1.0005)
        for (int ch = 0; ch < W; ++ch) {
            const float4 a = reinterpret_cast<const float4*>(C)[(size_t)row * (2 * W) + ch * 2];
            const float4 c = reinterpret_cast<const float4*>(C)[(size_t)row * (2 * W) + ch * 2 + 1];
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            unsigned hb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                hb[e] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)((x[e] - stats[ch * 8 + e] * inv) * f));
            Cs[(((size_t)(row >> 6) * KB + (ch >> 3)) * 64 + (row & 63)) * 8 + (ch & 7)] =
                make_uint4(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16), hb[4] | (hb[5] << 16), hb[6] | (hb[7] << 16));
        }
    }
}

struct PassArgs {
    const uint4* Qs;      // [nq_pad][8]  fp16 rows
    const uint4* Cs;      // [n_stages * 64][8]
    int nq, nc, n_stages, stages_per_range, n_groups;
    int p1_stride;        // pass 1 looks at every p1_stride-th stage of a range only (see filter_plan)
    int wcap;             // SPARSE: entries per query in wlist (F_WLIST_X x the final kernel's survivor slots)
    unsigned* gkeys;      // pass 1 out: [nq][n_groups] monotone keys of the group maxima
    const float* thr;     // pass 2 in:  [nq]
    unsigned long long* bits;   // pass 2 out: [nq][ranges][2][stages_per_range / 2] pass / fail bits
    int* wcnt;            // pass 2 out, SPARSE: [nq] appended words (zeroed by the launcher)
    uint4* wlist;         // pass 2 out, SPARSE: [nq][wcap] (word index in the row above, 0, word lo, word hi)
};

// The MFMAs of a stage are issued back to back (four independent accumulator chains, alternating): on gfx950 ANY
// other instruction between two MFMAs of one chain costs ~43 cycles (the accumulator-forwarding path is lost), so
// the per-score work is NOT interleaved with them -- it runs as one VALU burst per stage and overlaps with the
// MFMA burst of the other wave resident on the SIMD (two workgroups per CU).
//  pass 1: acc -> 16 running maxima per fragment: 1 VALU per score (v_med3 with +inf: a plain fmaxf is 3
//          instructions, two of them canonicalising its inputs).  A partial last stage is simply left out: maxima
//          over a SUBSET of the candidates still bound the k-th score from below.
//  pass 2: acc -> one pass / fail bit per score: 2 VALU per score (thr' - score, then w = (w << 1) | sign as ONE
//          v_alignbit); the 64 bits of two stages go to memory as one 8-byte store per lane and fragment.  No lists,
//          no atomics, no overflow in the hot loop.  Bit 63 - (16 j + r) of word g of row (q, range, h): candidate
//          32 (2 (S_r0 + 2 g) + j) + (r & 3) + 8 (r >> 2) + 4 h   (j = 0..3: the four tiles of two stages).
//  SPARSE (large candidate sets): a query's row of pass / fail bits is nc / 8 bytes, all but ~65 of its words zero.  At
//          500K candidates the rows of one 65,536-query block are 4.1 GB of 8-byte stores scattered over 131,072 rows in
//          flight (pass 2 took 10.2 ms against 3.5 ms for pass 1, the same products) and the final kernel scans them
//          again.  Instead a lane appends its NON-ZERO words to the query's list (one atomic slot counter per query;
//          ~1 % of the words).  The append is software-pipelined: the atomic of one flush is consumed at the next, so no
//          wave waits a memory round trip in the stage loop.  List order is arbitrary; the final kernel sorts anyway.
#ifndef MMREC_TF_NOCLIP    // probe: no clipping of outlying candidate rows
#define MMREC_TF_NOCLIP 0
#endif
#ifndef MMREC_TF_OCC3      // probe: three workgroups per CU for the word-list pass 2 (168 VGPRs: 12 spilled)
#define MMREC_TF_OCC3 0
#endif
// KB = kd / 64 (1 or 2): a row of 128 columns is walked as two 64-wide tiles per 64-candidate stage (the conversion lays the
// candidates out tile by tile), accumulating into the same accumulators; the per-score work runs once per stage.
template <bool FILTER, bool SPARSE, int KB>
__global__ __launch_bounds__(256, (MMREC_TF_OCC3 && FILTER && SPARSE && KB == 1) ? 3 : 2) void filter_pass_kernel(const PassArgs a) {
    __shared__ uint4 s_c[2][2][256];   // [buffer][32-row half of the tile][row * 8 + swizzled chunk]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * F_QWG + wave * 64;
    // stages of this workgroup (stages_per_range is a multiple of 4 stages)
    const int t_r0 = blockIdx.y * a.stages_per_range;
    const int t0 = t_r0;
    const int t1r = min(min(t_r0 + a.stages_per_range, a.n_stages), FILTER ? a.n_stages : a.nc / 64);   // pass 1: whole stages only
    // pass 1 may SUBSAMPLE: maxima over a subset of the candidates still bound the (k + m)-th best score from below (k + m
    // distinct candidates reach the bound), only less tightly -- more survivors for the exact refinement, fewer MFMAs here.
    // Its loop then runs over virtual stages t0 + j that stand for the real stages t0 + j * S.
    const int S = FILTER ? 1 : a.p1_stride;
    const int t1 = FILTER ? t1r : t0 + (max(t1r - t0, 0) + S - 1) / S;
    const int u0 = t0 * KB, u1 = t1 * KB;          // the same in 64 x 64 TILES (micro-steps of the pipeline below)
    float gm[2][16];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[f][r] = -INFINITY;
    if (t0 < t1) {   // uniform
    // query fragments: lane (i, h) holds B[k = 32 h + 8 s + j][n = i], i.e. chunk 4 h + s of column block kb of its query row
    half8 qf[2][4 * KB];
    // C operand of a chain's first MFMA: 0 (pass 1) / -thr of the lane's query (pass 2: acc = score - thr, the pass / fail
    // bit is the accumulator's sign).  The 128-wide rows have no 32 registers to spare for it (256 per wave at two workgroups
    // per CU); the word-list variant has since the prefetch fix (206 VGPRs with it) but measured 0-4 % slower with it than with
    // its quarter-stage skip: both keep thr in one register per fragment and subtract per score.
    constexpr bool CINIT = FILTER && !SPARSE && KB == 1;
    acc16 cinit[2];
    float thr[2];
    unsigned long long* brow[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int q = q0 + f * 32 + i;
#pragma unroll
        for (int c = 0; c < 4 * KB; ++c)
            qf[f][c] = __builtin_bit_cast(half8, a.Qs[(size_t)q * (8 * KB) + (c >> 2) * 8 + h * 4 + (c & 3)]);
        const float nt = CINIT ? ((q < a.nq) ? -a.thr[q] : -INFINITY) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[f][r] = nt;
        thr[f] = (FILTER && q < a.nq) ? float_below(a.thr[q]) : INFINITY;
        brow[f] = (FILTER && !SPARSE) ? a.bits + (((size_t)q * gridDim.y + blockIdx.y) * 2 + h) * (a.stages_per_range >> 1) : nullptr;
    }
    // SPARSE: the append in flight per fragment (slot < 0: none)
    int ps[2] = {-1, -1};
    unsigned pw[2] = {0u, 0u};
    unsigned long long pb[2] = {0ull, 0ull};
    const unsigned wbase = (blockIdx.y * 2 + h) * (a.stages_per_range >> 1);
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (ps[f] >= 0 && ps[f] < a.wcap)
                a.wlist[(size_t)(q0 + f * 32 + i) * a.wcap + ps[f]] =
                    make_uint4(pw[f], 0u, (unsigned)pb[f], (unsigned)(pb[f] >> 32));
            ps[f] = -1;
        }
    };
    // tile fill: thread -> (row rr, chunk cc) of both 32-row halves
    const int rr = tid >> 3, cc = tid & 7;
    const int slot = rr * 8 + (cc ^ ((rr >> 1) & 7));
    const int sw = (i >> 1) & 7;
    // global -> register ring, F_PF tiles ahead (the L2 / HBM latency is several tile times), -> LDS double
    // buffer.  The ring is eight named registers, not an array: an indexed private array went to scratch.
    uint4 ra0, rb0, ra1, rb1, ra2, rb2, ra3, rb3;
    ra0 = rb0 = ra1 = rb1 = ra2 = rb2 = ra3 = rb3 = make_uint4(0, 0, 0, 0);
    // EVERY load below is unconditional (past the range's end the last tile is loaded again): with `if (u + F_PF < u1)` around
    // them the compiler could not count the loads in flight and drained them all (s_waitcnt vmcnt(0)) before every LDS
    // store and every list append -- the four-tile prefetch distance was one stage's MFMAs in practice.
    auto gload = [&](int uu, uint4& xa, uint4& xb) __attribute__((always_inline)) {     // micro-step u -> its tile
        // past the end: the LAST STAGE's tile of the same column block (u0 is a multiple of KB).  Clamping to u1 - 1 alone
        // would hand column block KB - 1 to a kb == 0 micro-step of kd = 128: c[64:128] . q[0:64] is no candidate's score,
        // and pass 1 would take its group maxima -- and the bound -- from it.
        const int u = uu < u1 ? uu : u1 - KB + (uu - u0) % KB;
        const int tv = u / KB, kb = u - tv * KB;
        const size_t o = (((size_t)(t0 + (tv - t0) * S) * KB + kb) * 64 + rr) * 8 + cc;
        xa = a.Cs[o];
        xb = a.Cs[o + 32 * 8];
    };
    gload(u0, ra0, rb0);
    gload(u0 + 1, ra1, rb1);
    gload(u0 + 2, ra2, rb2);
    gload(u0 + 3, ra3, rb3);
    s_c[0][0][slot] = ra0;
    s_c[0][1][slot] = rb0;
    __syncthreads();
    int cur = 0;
    unsigned w[2] = {0u, 0u};   // pass 2: pass / fail bits of the current stage, per fragment
    unsigned long long bw[2] = {0ull, 0ull};
    acc16 a0 = cinit[0], a1 = cinit[1], b0 = cinit[0], b1 = cinit[1];   // a: rows 0..31 of the stage, b: 32..63; 0 / 1: query fragment
    // one micro-step u at position POS of the unrolled sequence (column block kb = POS % KB: a range starts at an even u);
    // (fa, fb) = ring slot that held its tile (free now: refilled with u + F_PF), (na, nb) = slot of u + 1
    auto step = [&](int u, auto POS, uint4& fa, uint4& fb, const uint4& na, const uint4& nb) __attribute__((always_inline)) {
        constexpr int kb = decltype(POS)::value % KB;
        // (micro-steps past u1 -- the range's length rounded up to four -- run on the last tile again: its maxima are real
        // candidates' scores once more, its pass / fail bits are dropped)
        {
        if (!(MMREC_TF_PROBE & 4)) gload(u + F_PF, fa, fb);
        half8 ca[4], cb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MMREC_TF_PROBE & 2) { ca[s] = qf[0][s]; cb[s] = qf[1][s]; continue; }
            ca[s] = __builtin_bit_cast(half8, s_c[cur][0][i * 8 + ((h * 4 + s) ^ sw)]);
            cb[s] = __builtin_bit_cast(half8, s_c[cur][1][i * 8 + ((h * 4 + s) ^ sw)]);
        }
        if (kb == 0) { a0 = cinit[0]; a1 = cinit[1]; b0 = cinit[0]; b1 = cinit[1]; }    // a new stage
        if (MMREC_TF_PROBE & 1) {
            a0[0] = __builtin_bit_cast(float4, ca[0]).x; a1[5] = __builtin_bit_cast(float4, cb[3]).y;
            b0[9] = __builtin_bit_cast(float4, ca[2]).z; b1[2] = __builtin_bit_cast(float4, cb[1]).w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca[s], qf[0][kb * 4 + s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca[s], qf[1][kb * 4 + s], a1, 0, 0, 0);
                b0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cb[s], qf[0][kb * 4 + s], b0, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cb[s], qf[1][kb * 4 + s], b1, 0, 0, 0);
            }
        }
        // the next tile goes to the other LDS buffer while the matrix pipe drains
        s_c[cur ^ 1][0][slot] = na;
        s_c[cur ^ 1][1][slot] = nb;
        // a?[r] / b?[r] = score of candidate 64 t (+ 32) + (r & 3) + 8 (r >> 2) + 4 h for query q0 + 32 f + i
        if (kb != KB - 1) {
            // (more column blocks of this stage to come)
        } else if (MMREC_TF_PROBE & 8) {
            if (a0[3] == 1234.5f) gm[0][0] = a1[7] + b0[1] + b1[2];
        } else if (!FILTER) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gm[0][r] = __builtin_fmaxf(__builtin_fmaxf(gm[0][r], a0[r]), b0[r]);    // ONE v_max3_f32 per two scores
                gm[1][r] = __builtin_fmaxf(__builtin_fmaxf(gm[1][r], a1[r]), b1[r]);
            }
        } else {
            // the accumulators started at -thr: acc = score - thr, FAIL bit = its sign; w = (w << 1) | bit is ONE
            // v_alignbit_b32 per score (no subtraction); the word is inverted once per 32 scores
            w[0] = w[1] = 0u;
            if (CINIT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    w[0] = __builtin_amdgcn_alignbit(w[0], __float_as_uint(a0[r]), 31);
                    w[1] = __builtin_amdgcn_alignbit(w[1], __float_as_uint(a1[r]), 31);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    w[0] = __builtin_amdgcn_alignbit(w[0], __float_as_uint(b0[r]), 31);
                    w[1] = __builtin_amdgcn_alignbit(w[1], __float_as_uint(b1[r]), 31);
                }
                w[0] = ~w[0];
                w[1] = ~w[1];
            } else {
                // Large candidate sets: a query keeps ~65 of 500,000 candidates, so the 32 queries x 64 candidates a fragment
                // scores in a stage hold a survivor in ~1 stage of 4.  The stage's largest score per lane (ONE v_max3 per two
                // scores) decides, wave-uniformly, whether the 2-instruction-per-score bit extraction runs at all for the
                // fragment: pass 2 was VALU co-bound at 8 VALU per MFMA (profiles/r03_topk_pmc.txt: MFMA pipe 53 % busy
                // against 68 % in pass 1, whose per-score work is the max3 alone).
                // bit = sign(thr' - score), thr' just below thr: score >= thr  <=>  score > thr'  <=>  sign bit set
                // The decision is taken per QUARTER of a fragment's stage (32 queries x 16 candidates: 8 accumulators): with the
                // ~160-250 survivors per query of a subsampled pass 1 a whole 32 x 64 block holds one every second stage (the
                // skip then saves nothing: 32 max3 + half of 64 = the 64 it replaces), a 32 x 16 quarter in one stage of 7.
                auto quarter = [&](const acc16& x, auto R0, float t, unsigned& wd) __attribute__((always_inline)) {
                    constexpr int r0 = decltype(R0)::value;
                    float m = __builtin_fmaxf(__builtin_fmaxf(x[r0], x[r0 + 1]), x[r0 + 2]);
                    m = __builtin_fmaxf(__builtin_fmaxf(m, x[r0 + 3]), x[r0 + 4]);
                    m = __builtin_fmaxf(__builtin_fmaxf(m, x[r0 + 5]), x[r0 + 6]);
                    m = __builtin_fmaxf(m, x[r0 + 7]);
                    if (__ballot(m > t) != 0ull) {
#pragma unroll
                        for (int r = r0; r < r0 + 8; ++r) wd = __builtin_amdgcn_alignbit(wd, __float_as_uint(t - x[r]), 31);
                    } else {
                        wd <<= 8;
                    }
                };
                using R0 = std::integral_constant<int, 0>;
                using R8 = std::integral_constant<int, 8>;
                quarter(a0, R0{}, thr[0], w[0]); quarter(a0, R8{}, thr[0], w[0]);
                quarter(b0, R0{}, thr[0], w[0]); quarter(b0, R8{}, thr[0], w[0]);
                quarter(a1, R0{}, thr[1], w[1]); quarter(a1, R8{}, thr[1], w[1]);
                quarter(b1, R0{}, thr[1], w[1]); quarter(b1, R8{}, thr[1], w[1]);
            }
        }
        if (!(MMREC_TF_PROBE & 16)) __syncthreads();
        cur ^= 1;
        }
        if (FILTER && kb == KB - 1) {      // a stage done: its 32 bits per fragment
            if (u >= u1) w[0] = w[1] = 0u;
            bw[0] = (bw[0] << 32) | w[0];
            bw[1] = (bw[1] << 32) | w[1];
        }
    };
    auto flush = [&](int g) __attribute__((always_inline)) {   // two stages done: one word per fragment
        if (FILTER && !SPARSE) {
            if (q0 + i < a.nq) brow[0][g] = bw[0];
            if (q0 + 32 + i < a.nq) brow[1][g] = bw[1];
        }
        if (FILTER && SPARSE) {
            commit();   // the previous flush's atomics have long returned
#pragma unroll
            for (int f = 0; f < 2; ++f)
                if (bw[f] != 0ull && q0 + f * 32 + i < a.nq) {
                    ps[f] = atomicAdd(a.wcnt + q0 + f * 32 + i, 1);
                    pw[f] = wbase + (unsigned)g;
                    pb[f] = bw[f];
                }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;
    for (int ub = u0; ub < u1; ub += F_PF) {       // four micro-steps = four stages (KB = 1) / two stages (KB = 2)
        step(ub, P0{}, ra0, rb0, ra1, rb1);
        step(ub + 1, P1{}, ra1, rb1, ra2, rb2);
        if (KB == 1) flush((ub - u0) >> 1);
        step(ub + 2, P2{}, ra2, rb2, ra3, rb3);
        step(ub + 3, P3{}, ra3, rb3, ra0, rb0);
        flush(KB == 1 ? ((ub - u0) >> 1) + 1 : (ub - u0) >> 2);
    }
    static_assert(F_PF == 4 && (KB == 1 || KB == 2), "the step sequence above is written for a 4-slot ring and 1 or 2 column blocks");
    if (FILTER && SPARSE) commit();
    }
    if (!FILTER) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int q = q0 + f * 32 + i;
            if (q >= a.nq) continue;
            unsigned* dst = a.gkeys + (size_t)q * a.n_groups + blockIdx.y * 32 + h * 16;   // 64-B aligned
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                reinterpret_cast<uint4*>(dst)[r >> 2] = make_uint4(f2key(gm[f][r]), f2key(gm[f][r + 1]),
                                                                   f2key(gm[f][r + 2]), f2key(gm[f][r + 3]));
        }
    }
}


// Shared tail of the two bound kernels (one wave per query): B = a lower bound, in approximate-score units, of the
// query's k-th best unmasked candidate's APPROXIMATE score  ->  thr[q] = B - 2 eps_q, flag[q] = 0; or, where clipped
// candidate rows make B unusable (B < eps), thr = +inf and flag = 1 (the exact slow queue).
// eps in the scaled (by the query's own and the candidates' power of two), centred units of the approximate
// scores: fp16 rounding of both operands (2u + u^2 and
// the accumulation, 1.0e-3 of |q| max|c'|) plus the fp32 rounding of the EXACT scores the final kernel ranks by
// (kd * 2^-24 of |q| max|c|, the uncentred norm: |c| <= |c'| + |mean|)
__device__ __forceinline__ float filter_eps(int q, int nc, int kd, const float* __restrict__ qnorm,
                                            const unsigned* __restrict__ cmax_key, const float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    float mu = 0.f;
    for (int c = lane; c < kd; c += 64) {
        const float mc = stats[c] / (float)nc;
        mu = fmaf(mc, mc, mu);
    }
    mu = wave_sum(mu);
    const float sc = fp16_scale(2.f * key2f(reinterpret_cast<const unsigned*>(stats)[ST_CMAX]));
    const float cmax = key2f(*cmax_key);
    const float kb = (float)kd * (1.f / 64.f);
    // + 2^-25 sqrt(kd) (|q| + max|c'|): elements below fp16's normal range are rounded with absolute error 2^-25
    // (sum |x_i| <= sqrt(kd) |x|; 2.4e-7 = 2^-22 for kd = 64)
    const bool clipped = reinterpret_cast<const int*>(stats)[ST_NOUT] > 0;
    const float cmax0 = clipped ? key2f(reinterpret_cast<const unsigned*>(stats)[ST_NMAX0]) : cmax;   // before clipping
    return qnorm[q] * (1.0e-3f * cmax + 4.0e-6f * kb * (cmax0 + sc * sqrtf(mu))) + 2.4e-7f * sqrtf(kb) * (qnorm[q] + cmax);
}
// need_nonneg: the cold path's bound may be a CLIPPED row's group maximum (stored score ~ f s, f < 1), which bounds that row's
// real score from below only where it is >= eps; the warm path corrects clipped rows one by one and passes false.
__device__ __forceinline__ void filter_eps_store(float B, int q, int lane, int nc, int kd, const float* __restrict__ qnorm,
                                                 const unsigned* __restrict__ cmax_key, const float* __restrict__ stats,
                                                 float* __restrict__ thr, int* __restrict__ flag, bool need_nonneg = true) {
    const float eps = filter_eps(q, nc, kd, qnorm, cmax_key, stats);
    if (lane == 0) {
        const bool clipped = reinterpret_cast<const int*>(stats)[ST_NOUT] > 0;
        // clipped candidate rows are lower bounds of real scores only where the bound is >= eps
        const bool ok = !need_nonneg || !clipped || B >= eps;
        thr[q] = ok ? B - 2.f * eps : INFINITY;
        flag[q] = ok ? 0 : 1;
    }
}

// thr[q] = (k + m)-th largest group maximum - 2 eps_q; queries the filter cannot serve are flagged and
// get thr = +inf (nothing passes).  One wave per query.
__global__ __launch_bounds__(256) void filter_bound_kernel(const unsigned* __restrict__ gkeys, int n_groups, int nq,
                                                           int nc, int k, const int32_t* __restrict__ mask_rowptr,
                                                           const float* __restrict__ qnorm,
                                                           const unsigned* __restrict__ cmax_key,
                                                           const float* __restrict__ stats, int kd,
                                                           float* __restrict__ thr, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int m = mask_rowptr ? mask_rowptr[q + 1] - mask_rowptr[q] : 0;
    const int rank = k + m;
    if (rank > n_groups || nc - m < k) {
        if (lane == 0) { thr[q] = INFINITY; flag[q] = 1; }
        return;
    }
    if (qnorm[q] == 0.f) {       // an all-zero query row (a user without interactions after propagation): every score is exactly 0
        if (lane == 0) { thr[q] = INFINITY; flag[q] = 2; }     // -> the final kernel writes the k lowest unmasked ids itself
        return;
    }
    unsigned key[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = lane + 64 * j;
        key[j] = e < n_groups ? gkeys[(size_t)q * n_groups + e] : 0u;
    }
    // largest T (multiple of 256: the low 8 bits only lower the bound by 2^-15 relative) with #{key >= T} >= rank
    // ... starting below the leading bits all keys share (scores of one query have similar exponents)
    unsigned kmax = 0u, kmin = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane + 64 * j < n_groups) { kmax = max(kmax, key[j]); kmin = min(kmin, key[j]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
    }
    const int top = 31 - __clz((int)((kmax ^ kmin) | 256u));          // highest bit in which two keys differ (>= 8)
    unsigned cur = top < 31 ? kmax & ~((2u << top) - 1u) : 0u;        // the shared prefix
    // `cur` is a lower bound of the (k + m)-th largest key after EVERY step (it only moves up while >= rank keys stay
    // above it), and pass 2 is correct for any lower bound -- a looser one only lets a few more candidates through to
    // the exact refinement.  12 steps resolve 1/4096 of the spread of the query's 256..512 group maxima, far finer than
    // their spacing: the remaining <= 12 steps of a full bisection bought nothing but 40 % of this kernel's time.
    const int last = max(8, top - 11);
    for (int bit = top; bit >= last; --bit) {
        const unsigned trial = cur | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) c += __popcll(__ballot(key[j] >= trial));
        if (c >= rank) cur = trial;
    }
    // eps in the scaled, centred units of the approximate scores: see filter_eps_store
    filter_eps_store(key2f(cur), q, lane, nc, kd, qnorm, cmax_key, stats, thr, flag);
}

// WARM calls (mmrec_score_topk_hinted_f32): the threshold WITHOUT pass 1.  The caller hands over, per query, a list of
// `hk` candidate ids it believes rank high -- the previous evaluation's top-k of that user, or the VALID pass's list when the
// TEST pass follows on the same frozen tables (trainer.py:262,271: same train-positive mask).  ANY k distinct, unmasked,
// in-range candidates bound the k-th best score from below: with a_j their approximate scores (the SAME fp16 operands pass 2
// multiplies -- Qs, Cs -- products exact in fp32, fp32 accumulation: |a_j - s_j| <= eps_q by the bound stated at the head of
// this file, which never depended on the accumulation order), B = min_j a_j gives k unmasked candidates with exact score >=
// B - eps, so every true top-k candidate has exact score >= B - eps and approximate score >= B - 2 eps =: thr -- the cold
// path's argument with the (k + m)-th group maximum replaced by B (no "+ m": the listed ids are checked against the mask).
// More than k usable ids (the final kernel leaves the runners-up it ranked behind the top-k in the list: hk = 64 or 128): B = the
// k-th LARGEST of their scores -- after the tables have moved the new top-k still sits inside the old top-64 long after it has
// left the old top-k.
// Clipped rows (large candidate sets): a listed clipped row is stored as f c', f = tau' / |c'| < 1, so its approximate score a
// bounds f s: s >= (a - eps) / f (a - eps of either sign); it enters with B_j = (a - eps) / f + eps.  (Taking a itself, as the
// cold path's group maxima must, cost every user whose list holds one of the <= 32 outlying items -- most users -- a bound far
// below the k-th score: 16 % of a config-5 block in the overflow / slow queues, 82 ms instead of 6.)
// A STALE list only loosens thr (more survivors for the exact
// refinement, the overflow / slow queues beyond that); a list with fewer than k usable ids sends the query to the exact slow
// queue: results never depend on the hint.  One wave per query; W = kd / 8 lanes per listed row (16-B fp16 chunks).
template <int KB>
__global__ __launch_bounds__(256) void filter_hint_bound_kernel(const uint4* __restrict__ Qs, const uint4* __restrict__ Cs,
                                                                const int32_t* __restrict__ hint, int hk,
                                                                const int64_t* __restrict__ hint_rows, int nq, int nc, int k,
                                                                const int32_t* __restrict__ mask_rowptr,
                                                                const int32_t* __restrict__ mask_col,
                                                                const float* __restrict__ qnorm,
                                                                const unsigned* __restrict__ cmax_key,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ cnorm,     // clipping sets: row norms
                                                                float* __restrict__ thr, int* __restrict__ flag) {
    constexpr int W = 8 * KB, RPS = 64 / W;     // lanes per row, rows per wave step
    constexpr int NB = 8;                       // row steps in flight: 64 rows of 64 columns are ONE batch of gathers
    __shared__ int s_id[4][128];
    __shared__ float s_val[4][128];
    __shared__ int s_mask[4][F_MASK_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = blockIdx.x * 4 + wave;
    if (q >= nq) return;                        // waves are independent below (wave-level fences only)
    // A wave's life is memory round trips and ~500 instructions (x 4 cycles x 19,445 waves on Amazon-Baby: the first form of
    // this kernel, with a 128-element sort and an all-pairs duplicate check, took 59 us -- longer than the pass it replaces):
    // everything that does not depend on something else is requested at once -- the list, the mask range, the query row --
    // then the listed rows AND the mask entries together (the gathers do not wait for the validity checks: an unusable id's
    // score is computed and dropped).
    const int32_t* hrow = hint + (size_t)(hint_rows ? hint_rows[q] : (int64_t)q) * hk;
    const bool two = hk > 64;                   // (uniform) lists of up to 64 ids are one id per lane
    int id[2];
    id[0] = lane < hk ? hrow[lane] : -1;
    id[1] = two && lane + 64 < hk ? hrow[lane + 64] : -1;
    const int m_lo = mask_rowptr ? mask_rowptr[q] : 0, m = mask_rowptr ? mask_rowptr[q + 1] - m_lo : 0;
    const float qn = qnorm[q];
    const int ch = lane % W, sub = lane / W;
    const uint4 qraw = Qs[(size_t)q * W + ch];
    if (nc - m < k) {
        if (lane == 0) { thr[q] = INFINITY; flag[q] = 1; }
        return;
    }
    if (qn == 0.f) {                            // all scores tie at 0: the final kernel writes the k lowest unmasked ids
        if (lane == 0) { thr[q] = INFINITY; flag[q] = 2; }
        return;
    }
    s_id[wave][lane] = id[0];
    if (two) s_id[wave][lane + 64] = id[1];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const bool clips = cnorm != nullptr && reinterpret_cast<const int*>(stats)[ST_NOUT] > 0;
    const float tau = clips ? stats[ST_TAU] : INFINITY;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
    half2_t qp[4];
    qp[0] = __builtin_bit_cast(half2_t, qraw.x); qp[1] = __builtin_bit_cast(half2_t, qraw.y);
    qp[2] = __builtin_bit_cast(half2_t, qraw.z); qp[3] = __builtin_bit_cast(half2_t, qraw.w);
    const bool mask_in_lds = m <= F_MASK_LDS;
    const int32_t* ml = mask_col + m_lo;
    float eps = 0.f;
    bool ok[2] = {false, false};
    for (int e0 = 0; e0 < hk; e0 += RPS * NB) {
        uint4 cv[NB];
        float nrm[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int e = e0 + b * RPS + sub;
            const int rid = e < hk ? s_id[wave][e] : -1;
            const int r = rid >= 0 && rid < nc ? rid : 0;
            cv[b] = Cs[(((size_t)(r >> 6) * KB + (ch >> 3)) * 64 + (r & 63)) * 8 + (ch & 7)];
            nrm[b] = clips ? cnorm[r] : 0.f;
        }
        if (e0 == 0) {
            // under the gathers: the mask entries to LDS, eps, and which ids are in range and unmasked
            if (mask_in_lds)
                for (int e = lane; e < m; e += 64) s_mask[wave][e] = ml[e];
            eps = filter_eps(q, nc, 64 * KB, qnorm, cmax_key, stats);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ok[u] = id[u] >= 0 && id[u] < nc;
                if (ok[u] && m > 0) {               // masked (train-positive) ids bound nothing
                    const int32_t* mm = mask_in_lds ? s_mask[wave] : ml;
                    int lo = 0, hi = m;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (mm[mid] < id[u]) lo = mid + 1; else hi = mid;
                    }
                    ok[u] = !(lo < m && mm[lo] == id[u]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // exact fp16 products, fp32 accumulation (v_dot2_f32_f16: another order than the matrix cores' -- inside eps)
            float a = __builtin_amdgcn_fdot2(qp[0], __builtin_bit_cast(half2_t, cv[b].x), 0.f, false);
            a = __builtin_amdgcn_fdot2(qp[1], __builtin_bit_cast(half2_t, cv[b].y), a, false);
            a = __builtin_amdgcn_fdot2(qp[2], __builtin_bit_cast(half2_t, cv[b].z), a, false);
            a = __builtin_amdgcn_fdot2(qp[3], __builtin_bit_cast(half2_t, cv[b].w), a, false);
            if (W == 16) a += lane_xor_f<8>(a);
            a += lane_xor_f<4>(a);
            a += lane_xor_f<2>(a);
            a += lane_xor_f<1>(a);
            if (nrm[b] >= tau) a = (a - eps) / (tau * (1.f - 1.f / 512.f) / nrm[b]) + eps;     // a clipped row: see above
            const int e = e0 + b * RPS + sub;
            if (e < hk && ch == 0) s_val[wave][e] = a;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // An id counts once.  Sorted by id (ids < 2^24 are exact as floats; unusable ones get distinct negative keys) with the
    // bound riding along, a repeated id is a neighbour: one sort instead of hk^2 / 64 comparisons per lane.
    Cand y0 = Cand{ok[0] ? (float)id[0] : (float)(-1 - lane), __float_as_int(lane < hk ? s_val[wave][lane] : 0.f)};
    Cand y1 = Cand{ok[1] ? (float)id[1] : (float)(-65 - lane), __float_as_int(two && lane + 64 < hk ? s_val[wave][lane + 64] : 0.f)};
    bool use0, use1 = false;
    if (two) {
        bitonic128(y0, y1, lane);
        // rank order: element lane of y0, then element lane of y1; the predecessor of y1's lane 0 is y0's lane 63
        const float p0 = __shfl_up(y0.v, 1, 64), p1 = __shfl_up(y1.v, 1, 64), last0 = __shfl(y0.v, 63, 64);
        use0 = y0.v >= 0.f && (lane == 0 || p0 != y0.v);
        use1 = y1.v >= 0.f && ((lane == 0 ? last0 : p1) != y1.v);
    } else {
        bitonic64(y0, lane);
        const float p0 = __shfl_up(y0.v, 1, 64);
        use0 = y0.v >= 0.f && (lane == 0 || p0 != y0.v);
    }
    const int n_ok = __popcll(__ballot(use0)) + __popcll(__ballot(use1));
    if (n_ok < k) {                             // not enough to bound the k-th score: the exact slow queue
        if (lane == 0) { thr[q] = INFINITY; flag[q] = 1; }
        return;
    }
    // A lower bound of the k-th largest of the usable rows' bounds, by bisection on their monotone integer image (as
    // filter_bound_kernel: `cur` is a lower bound after EVERY step; 16 steps resolve 2^-16 of the bounds' spread).
    const unsigned key0 = use0 ? f2key(__int_as_float(y0.i)) : 0u, key1 = use1 ? f2key(__int_as_float(y1.i)) : 0u;
    unsigned kmax = max(key0, key1), kmin = min(use0 ? key0 : 0xffffffffu, use1 ? key1 : 0xffffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
    }
    const int top = 31 - __clz((int)((kmax ^ kmin) | 256u));
    unsigned cur = top < 31 ? kmax & ~((2u << top) - 1u) : 0u;
    const int last = max(8, top - 15);
    for (int bit = top; bit >= last; --bit) {
        const unsigned trial = cur | (1u << bit);
        const int c = __popcll(__ballot(use0 && key0 >= trial)) + __popcll(__ballot(use1 && key1 >= trial));
        if (c >= k) cur = trial;
    }
    if (lane == 0) {
        const float B = key2f(cur);
        thr[q] = B - 2.f * eps;
        flag[q] = 0;
    }
}

// The lists of the queries the overflow / slow queues ranked (their kernels write out_idx only): top-k ids, -1 beyond.  One wave
// per queue entry; an entry that sits in both queues is written twice with the same ids.
__global__ __launch_bounds__(256) void filter_hint_from_out_kernel(const int* __restrict__ flist, const int* __restrict__ olist,
                                                                   const int* __restrict__ n_flagged,
                                                                   const int64_t* __restrict__ out_idx, int k,
                                                                   int32_t* __restrict__ hint_out, int hk,
                                                                   const int64_t* __restrict__ hint_rows,
                                                                   int* __restrict__ queue_counts) {
    const int lane = threadIdx.x & 63;
    const int n0 = n_flagged[0], n1 = olist ? n_flagged[1] : 0;
    // queue_counts[0] += queries the exact slow queue served, [1] += queries that went through the overflow queue (either is a
    // sign of a loose threshold: a warm caller falls back to the cold path when they grow)
    if (queue_counts && blockIdx.x == 0 && threadIdx.x == 0) {
        queue_counts[0] += n0;
        queue_counts[1] += n_flagged[1];
    }
    if (!hint_out) return;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n0 + n1; e += gridDim.x * 4) {
        const int q = e < n0 ? flist[e] : olist[e - n0];
        int32_t* hr = hint_out + (size_t)(hint_rows ? hint_rows[q] : (int64_t)q) * hk;
        for (int j = lane; j < hk; j += 64) hr[j] = j < k ? (int32_t)out_idx[(size_t)q * k + j] : -1;
    }
}

// exact fp32 score = fixed tree over the 16 float4 chunk products (identical in the final and slow kernels)
__device__ __forceinline__ float tree16(const float (&p)[16]) {
    float l1[8], l2[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) l1[j] = p[j] + p[j + 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) l2[j] = l1[j] + l1[j + 4];
    return (l2[0] + l2[2]) + (l2[1] + l2[3]);
}

// kd = 128: chunk products j and j + 16 are added first (what a lane of the final kernel holds), then the same tree
template <int KB>
__device__ __forceinline__ float exact_score(const float4* __restrict__ q4, const float4* __restrict__ c4) {
    float p[16];
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
        p[ch] = f4_dot(q4[ch], c4[ch]);
        if (KB == 2) p[ch] += f4_dot(q4[ch + 16], c4[ch + 16]);
    }
    return tree16(p);
}

// Sort list[0..n) (n >= 1): on return y0 of lane l is the rank-l entry (l < 64).
__device__ __forceinline__ Cand sort_best64(const unsigned long long* list, int n, int lane) {
    auto fetch = [&](int e) -> Cand { return e < n ? unpack_cand(list[e]) : Cand{-INFINITY, INT_MAX}; };
    Cand y0 = fetch(lane), y1;
    if (n <= 64) {         // (wave-uniform) the whole list is one candidate per lane
        bitonic64(y0, lane);
        return y0;
    }
    int pos = 64;
    do {
        y1 = fetch(pos + lane);
        bitonic128(y0, y1, lane);
        pos += 64;
    } while (pos < n);
    return y0;
}

// Best 128 of list[0..n) (n >= 1) in rank order: y0 of lane l is the rank-l entry, y1 the rank-(64 + l) entry.  The best 128 of
// (y0, y1, z) are y0 and the best 64 of (y1, z): an element of y0 has at most 63 + 64 = 127 entries ahead of it.
__device__ __forceinline__ void sort_best128(const unsigned long long* list, int n, int lane, Cand& y0, Cand& y1) {
    auto fetch = [&](int e) -> Cand { return e < n ? unpack_cand(list[e]) : Cand{-INFINITY, INT_MAX}; };
    y0 = fetch(lane);
    y1 = fetch(64 + lane);
    bitonic128(y0, y1, lane);
    for (int pos = 128; pos < n; pos += 64) {
        Cand z = fetch(pos + lane);
        bitonic128(y1, z, lane);     // y1 = best 64 of (y1, z), in rank order
        bitonic128(y0, y1, lane);    // the 128 kept, in rank order again
    }
}
// rank-ordered best min(k, 128) of a list: ranks 0..63 in y0, 64..127 in y1 (y1 untouched garbage-free when k <= 64)
__device__ __forceinline__ void sort_best_k(const unsigned long long* list, int n, int lane, int k, Cand& y0, Cand& y1) {
    if (k <= 64) {
        y0 = sort_best64(list, n, lane);
        y1 = Cand{-INFINITY, INT_MAX};
    } else {
        sort_best128(list, n, lane, y0, y1);
    }
}

// Exact streaming top-k of ONE query by the F_SLOW_WAVES waves of a workgroup, for the queries the filter cannot
// serve (heavy users whose k + m exceeds the number of groups, massive ties, fewer than k unmasked candidates): the
// waves take the 64-candidate steps round robin, one candidate per lane, scores by the same tree as the fast path,
// masked candidates at -1e10 (trainer.py:307); each wave keeps its own list with threshold = strict k-th best so far
// after every compaction (its later ids are larger: ties lose); wave 0 merges the waves' top-k lists (the global
// top-k under (score desc, id asc) is contained in their union).  One wave per query took 80-200 us per evaluation
// batch for the handful of heavy users it typically holds.
// [st0, st1): the 64-candidate steps this workgroup takes; `part` != nullptr: the workgroup's sorted top-k goes there
// (64 packed entries) instead of the outputs -- large candidate sets are split over several workgroups per query and
// merged by filter_slow_merge_kernel (one workgroup streaming 500K candidates for ONE flagged query took 5 ms, as long
// as the whole 20,000-query block of the fast path).
// LISTED: the candidates are the n_listed ids of `listed` (LDS; ARBITRARY order: the survivors of an overflowing query,
// filter_overflow_kernel) instead of the steps [st0, st1) of all candidates; the running threshold is then the k-th best
// (score, id) PAIR, since a later tie may carry the smaller id.
template <int KB, bool LISTED = false>
__device__ __forceinline__ void slow_topk(const float* __restrict__ Q, const float* __restrict__ C, int nc, int k,
                                          const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col,
                                          int q, unsigned long long (*lists)[F_CAPQ], unsigned long long* merged,
                                          int32_t* mask_lds, int lane, int wave, int64_t* __restrict__ out_idx,
                                          float* __restrict__ out_val, int st0, int st1,
                                          unsigned long long* __restrict__ part, const int* listed = nullptr, int n_listed = 0) {
    unsigned long long* list = lists[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int steps = LISTED ? (n_listed + 63) / 64 : st1;
    q = __builtin_amdgcn_readfirstlane(q);
    const float4* q4 = reinterpret_cast<const float4*>(Q) + (size_t)q * (16 * KB);   // wave-uniform: scalar loads
    // the query's sorted mask list: binary-searched per candidate, from LDS when it fits (heavy users are what this
    // path is for: a per-lane cursor over a list in global memory made them cost ~0.1 ms each)
    const int m_lo = mask_rowptr ? mask_rowptr[q] : 0;
    const int m = mask_rowptr ? mask_rowptr[q + 1] - m_lo : 0;
    const int32_t* ml = mask_col + m_lo;
    if (m > 0 && m <= F_SLOW_MASK_LDS) {
        for (int e = wave * 64 + lane; e < m; e += 64 * F_SLOW_WAVES) mask_lds[e] = ml[e];
        ml = mask_lds;
    }
    __syncthreads();
    float teff = -INFINITY;
    Cand kth = Cand{-INFINITY, INT_MAX};      // LISTED: the k-th best pair so far
    int cnt = 0;
    for (int it = st0 + wave;; it += F_SLOW_WAVES) {
        const bool last = it >= steps;
        if (!last) {
            const int e_l = it * 64 + lane;
            const int c = LISTED ? (e_l < n_listed ? listed[e_l] : INT_MAX) : e_l;
            float v = -INFINITY;
            if (c >= 0 && c < nc) {
                v = exact_score<KB>(q4, reinterpret_cast<const float4*>(C) + (size_t)c * (16 * KB));
                int lo = 0, hi = m;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ml[mid] < c) lo = mid + 1; else hi = mid;
                }
                if (lo < m && ml[lo] == c) v = -1e10f;
            }
            const bool pass = LISTED ? (c >= 0 && c < nc && cand_before(Cand{v, c}, kth)) : v > teff;
            const unsigned long long b = __ballot(pass);
            if (pass) list[cnt + __popcll(b & lt)] = pack_cand(v, c);
            cnt += __popcll(b);
            if (cnt <= F_CAPQ - 64) continue;      // (keeps <= 128 of the 256 slots after a compaction: room for >= 2 more steps)
        }
        // compaction (list nearly full) or this wave's final list: best min(cnt, k) entries, sorted
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int n = cnt;
        Cand y0 = Cand{-INFINITY, INT_MAX}, y1 = y0;
        if (n > 0) sort_best_k(list, n, lane, k, y0, y1);
        const int keep = min(n, k);
        if (last) {
            merged[wave * 128 + lane] = pack_cand(lane < keep ? y0.v : -INFINITY, lane < keep ? y0.i : INT_MAX);
            merged[wave * 128 + 64 + lane] = pack_cand(64 + lane < keep ? y1.v : -INFINITY, 64 + lane < keep ? y1.i : INT_MAX);
            break;
        }
        if (lane < keep) list[lane] = pack_cand(y0.v, y0.i);
        if (64 + lane < keep) list[64 + lane] = pack_cand(y1.v, y1.i);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        cnt = keep;
        if (n >= k) {
            teff = fmaxf(teff, k <= 64 ? __shfl(y0.v, k - 1, 64) : __shfl(y1.v, k - 65, 64));
            if (LISTED) kth = Cand{teff, k <= 64 ? __shfl(y0.i, k - 1, 64) : __shfl(y1.i, k - 65, 64)};   // (the kept list holds the old top-k)
        }
    }
    __syncthreads();
    if (wave == 0) {
        Cand y0, y1;
        sort_best_k(merged, F_SLOW_WAVES * 128, lane, k, y0, y1);
        if (part) {
            part[lane] = pack_cand(lane < k ? y0.v : -INFINITY, lane < k ? y0.i : INT_MAX);
            part[64 + lane] = pack_cand(64 + lane < k ? y1.v : -INFINITY, 64 + lane < k ? y1.i : INT_MAX);
        } else {
            if (lane < k) {
                const bool ok = y0.i != INT_MAX;
                out_idx[(size_t)q * k + lane] = ok ? (int64_t)y0.i : (int64_t)-1;
                if (out_val) out_val[(size_t)q * k + lane] = ok ? y0.v : -INFINITY;
            }
            if (64 + lane < k) {
                const bool ok = y1.i != INT_MAX;
                out_idx[(size_t)q * k + 64 + lane] = ok ? (int64_t)y1.i : (int64_t)-1;
                if (out_val) out_val[(size_t)q * k + 64 + lane] = ok ? y1.v : -INFINITY;
            }
        }
    }
    __syncthreads();   // `merged` / the lists are reused by the workgroup's next query
}

// CAP: survivor slots per query (= entries of its word list when SPARSE)
template <bool SPARSE, int KB, int CAP>
__global__ __launch_bounds__(256) void filter_final_kernel(
    const float* __restrict__ Q, const float* __restrict__ C, int nq, int nc, int k,
    const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col,
    const unsigned long long* __restrict__ bits, const int* __restrict__ wcnt, const uint4* __restrict__ wlist,
    int n_ranges, int tiles_per_range, const int* __restrict__ flag,
    int* __restrict__ flist, int* __restrict__ n_flagged, int64_t* __restrict__ out_idx, float* __restrict__ out_val,
    const int* __restrict__ outl,     // outl: nullptr, or {count, ids ...} of the clipped candidate rows (always rescored)
    int* __restrict__ olist,          // SPARSE: queue of the queries whose survivors do not fit CAP (filter_overflow_kernel);
                                      // its length is n_flagged[1]
    int wcap,                         // SPARSE: entries per query in wlist (>= CAP)
    int32_t* __restrict__ hint_out, int hk, const int64_t* __restrict__ hint_rows) {   // warm calls' lists (nullptr: none kept)
    __shared__ unsigned long long s_l[4][CAP];   // (score, id) of the unmasked survivors
    __shared__ int s_ids[4][CAP];
    __shared__ int s_mask[4][F_MASK_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq) return;   // waves are independent below (wave-level fences only)
    // everything the wave needs first is requested at once: the loads do not depend on each other, and a wave's life
    // is a chain of L2 round trips (flag -> mask range -> mask list -> bit words -> query row -> candidate rows)
    const int gpr = tiles_per_range >> 2;
    // SPARSE: the query's appended (word index, word) entries instead of its row of words
    const int n_app = SPARSE ? wcnt[q] : 0;
    const int n_words = SPARSE ? min(n_app, CAP) : n_ranges * 2 * gpr;
    const unsigned long long* row = SPARSE ? nullptr : bits + (size_t)q * n_words;
    const uint4* ents = SPARSE ? wlist + (size_t)q * wcap : nullptr;
    int wi_next = lane;       // index of the word `x_next` (SPARSE: read from the entry)
    auto word = [&](int e, int& wi) -> unsigned long long {
        if (SPARSE) {
            if (e >= n_words) { wi = 0; return 0ull; }
            const uint4 t = ents[e];
            wi = (int)t.x;
            return (unsigned long long)t.z | ((unsigned long long)t.w << 32);
        }
        wi = e;
        const int rg = e / gpr, g = e - rg * gpr;               // rg = 2 * range + h
        // words of tile groups past the last tile are never written by pass 2
        const bool live = e < n_words && (rg >> 1) * tiles_per_range + 4 * g < (nc + 31) / 32;
        return live ? row[e] : 0ull;
    };
    unsigned long long x_next = word(lane, wi_next);
    float4 qv[KB];      // the lane's chunks (lane & 15) + 16 kb of the query row
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) qv[kb] = reinterpret_cast<const float4*>(Q)[(size_t)q * (16 * KB) + kb * 16 + (lane & 15)];
    const int fl = flag[q];
    const int m_lo = mask_rowptr ? mask_rowptr[q] : 0, m = mask_rowptr ? mask_rowptr[q + 1] - m_lo : 0;
    if (fl == 2 && m <= F_MASK_LDS) {
        // all scores tie at 0 (zero query row; nc - m >= k is the bound kernel's condition): ties go to the lower id, so the
        // answer is the k lowest unmasked ids -- the j-th of them is j + #{i : mask[i] - i <= j} (mask sorted ascending).
        // Through the slow queue each such query cost a 500K-candidate scan (0.4 ms per 65,536-query block at config 5).
        for (int e = lane; e < m; e += 64) s_mask[wave][e] = mask_col[m_lo + e];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        int32_t* hr = hint_out ? hint_out + (size_t)(hint_rows ? hint_rows[q] : (int64_t)q) * hk : nullptr;
        for (int j = lane; j < max(k, hr ? hk : 0); j += 64) {
            if (j >= k) { hr[j] = -1; continue; }
            int lo = 0, hi = m;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_mask[wave][mid] - mid <= j) lo = mid + 1; else hi = mid;
            }
            out_idx[(size_t)q * k + j] = (int64_t)(j + lo);
            if (out_val) out_val[(size_t)q * k + j] = 0.f;
            if (hr) hr[j] = j + lo;
        }
        return;
    }
    bool bad = fl != 0 || m > F_MASK_LDS || n_app > CAP;
    // more survivors than slots (the list itself, or the ids its words decode to): the overflow queue ranks them exactly
    // from its (longer) list -- not by the slow queue's scan of ALL candidates
    bool over = SPARSE && olist && fl == 0 && n_app > CAP;
    if (!bad) {
        // A: decode the pass / fail bits of pass 2 into candidate ids; stage the query's sorted mask list
        for (int e = lane; e < m; e += 64) s_mask[wave][e] = mask_col[m_lo + e];
        const unsigned long long lt = (1ull << lane) - 1ull;
        int n = 0;
        for (int w0 = 0; w0 < n_words; w0 += 64) {
            const int wi = wi_next;
            const int rg = wi / gpr, g = wi - rg * gpr;
            unsigned long long x = x_next;
            if (w0 + 64 < n_words) x_next = word(w0 + lane + 64, wi_next);   // the next word travels while this one is decoded
            const int cbase = ((rg >> 1) * tiles_per_range + 4 * g) * 32 + 4 * (rg & 1);
            for (;;) {
                const unsigned long long b = __ballot(x != 0ull);
                if (b == 0ull) break;
                if (x != 0ull) {
                    const int pz = __clzll((long long)x);          // 16 j + r
                    x &= ~(0x8000000000000000ull >> pz);
                    const int r = pz & 15, dst = n + __popcll(b & lt);
                    if (dst < CAP) s_l[wave][dst] = (unsigned long long)(unsigned)(cbase + (pz >> 4) * 32 + (r & 3) + 8 * (r >> 2));
                }
                n += __popcll(b);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        int c[CAP / 64];
        const int n_out = outl ? outl[0] : 0;
        if (n_out > 0 && n <= CAP) {     // the clipped rows join the survivors unless pass 2 let them through already
#pragma unroll
            for (int u = 0; u < CAP / 64; ++u) c[u] = lane + 64 * u < n ? (int)s_l[wave][lane + 64 * u] : -1;
            const int n0 = n;
            for (int o = 0; o < n_out; ++o) {
                const int id = outl[1 + o];
                bool has = false;
#pragma unroll
                for (int u = 0; u < CAP / 64; ++u) has |= c[u] == id;
                if (__ballot(has) == 0ull) {
                    if (lane == 0 && n < CAP) s_l[wave][n] = (unsigned long long)(unsigned)id;
                    ++n;
                }
            }
            if (n != n0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (n > CAP && SPARSE && olist) over = true;
        if (n > CAP || n < k) bad = true;
#pragma unroll
        for (int u = 0; u < CAP / 64; ++u) c[u] = (!bad && lane + 64 * u < n) ? (int)s_l[wave][lane + 64 * u] : -1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // B: drop padding and masked ids, compact the rest
        int valid = 0;
#pragma unroll
        for (int u = 0; u < CAP / 64; ++u) {
            bool ok = c[u] >= 0 && c[u] < nc;
            if (ok && m > 0) {
                int lo = 0, hi = m;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_mask[wave][mid] < c[u]) lo = mid + 1; else hi = mid;
                }
                ok = !(lo < m && s_mask[wave][lo] == c[u]);
            }
            const unsigned long long b = __ballot(ok);
            if (ok) s_ids[wave][valid + __popcll(b & lt)] = c[u];
            valid += __popcll(b);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (valid < k) bad = true;
        if (!bad) {
            // C: exact scores, 16 lanes per candidate row, F_FINAL_ROWS rows in flight per lane (4 x as many per wave and L2
            // round trip; tools/prof_topk_variants.py `final16`: 16 rows = 2 x fewer round trips, 116 instead of 68 VGPRs)
            const int sub = lane & 15, g = lane >> 4;
            constexpr int NR = F_FINAL_ROWS;
            for (int e0 = 0; e0 < valid; e0 += 4 * NR) {
                int id[NR];
                float4 cv[NR][KB];
#pragma unroll
                for (int u = 0; u < NR; ++u) id[u] = e0 + 4 * u + g < valid ? s_ids[wave][e0 + 4 * u + g] : -1;
#pragma unroll
                for (int u = 0; u < NR; ++u)
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
                        cv[u][kb] = id[u] >= 0 ? reinterpret_cast<const float4*>(C)[(size_t)id[u] * (16 * KB) + kb * 16 + sub] : f4_zero();
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    float part = f4_dot(qv[0], cv[u][0]);
                    if (KB == 2) part += f4_dot(qv[KB - 1], cv[u][KB - 1]);      // chunk + 16: the order of exact_score<2>
                    const float sc = row16_sum(part);
                    if (sub == 0 && id[u] >= 0) s_l[wave][e0 + 4 * u + g] = pack_cand(sc, id[u]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // D: sort, cut to k
            Cand y0, y1;
            sort_best_k(s_l[wave], valid, lane, k, y0, y1);
            if (lane < k) {
                out_idx[(size_t)q * k + lane] = (int64_t)y0.i;
                if (out_val) out_val[(size_t)q * k + lane] = y0.v;
            }
            if (64 + lane < k) {
                out_idx[(size_t)q * k + 64 + lane] = (int64_t)y1.i;
                if (out_val) out_val[(size_t)q * k + 64 + lane] = y1.v;
            }
            if (hint_out) {
                // the next warm call's list: the top-k AND the runners-up this wave ranked anyway (ranks < 64, or < 128 for
                // k > 64: sort_best_k), -1 beyond the survivors
                int32_t* hr = hint_out + (size_t)(hint_rows ? hint_rows[q] : (int64_t)q) * hk;
                if (lane < hk) hr[lane] = lane < valid ? y0.i : -1;
                if (64 + lane < hk) hr[64 + lane] = (k > 64 && 64 + lane < valid) ? y1.i : -1;
            }
        }
    }
    if (over) {
        if (lane == 0) olist[atomicAdd(n_flagged + 1, 1)] = q;       // served by filter_overflow_kernel
    } else if (bad && lane == 0) {
        flist[atomicAdd(n_flagged, 1)] = q;                          // served by filter_slow_kernel
    }
}

// The queue of the final kernel, served by persistent workgroups (inlining slow_topk into the final kernel doubled
// its registers and its time: 53 -> 122 us on the Baby evaluation with an empty queue).
// splits per flagged query: `want` (host: by the candidate count) as long as the (query, split) partial lists fit the
// F_SLOW_PARTS-entry buffer; 1 = the workgroup writes the outputs itself
__device__ __forceinline__ int slow_splits(int nf, int want) {
    return nf > 0 ? max(1, min(want, F_SLOW_PARTS / nf)) : 1;
}

template <int KB>
__global__ __launch_bounds__(64 * F_SLOW_WAVES) void filter_slow_kernel(
    const float* __restrict__ Q, const float* __restrict__ C, int nc, int k,
    const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col, const int* __restrict__ flist,
    const int* __restrict__ n_flagged, int64_t* __restrict__ out_idx, float* __restrict__ out_val, int want_splits,
    unsigned long long* __restrict__ parts) {
    __shared__ unsigned long long s_l[F_SLOW_WAVES][F_CAPQ];
    __shared__ unsigned long long s_m[F_SLOW_WAVES * 128];
    __shared__ int32_t s_mask[F_SLOW_MASK_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nf = *n_flagged;
    const int S = slow_splits(nf, want_splits), steps = (nc + 63) / 64;
    for (int w = blockIdx.x; w < nf * S; w += gridDim.x) {   // uniform per workgroup
        const int j = w / S, sp = w - j * S;
        const int st0 = (int)((long long)steps * sp / S), st1 = (int)((long long)steps * (sp + 1) / S);
        slow_topk<KB>(Q, C, nc, k, mask_rowptr, mask_col, flist[j], s_l, s_m, s_mask, lane, wave, out_idx, out_val, st0, st1,
                  S > 1 ? parts + (size_t)w * 128 : nullptr);
    }
}

// The overflow queue of the final kernel (SPARSE), served by persistent workgroups: a query with more surviving words than
// its list holds (heavy users: ~2 (k + m) survivors; queries whose scores are packed within 2 eps of the bound -- 0.1 % of the
// users of a TRAINED config-5 model, which cost half the evaluation's time in the slow queue's scan of all 500K candidates)
// is ranked exactly from what pass 2 found: its list of up to F_WLIST_X x that many words (+ the clipped rows), decoded into
// an id list in LDS and streamed through slow_topk.  A list or id list that overflows in turn sends the query on to the slow
// queue (this kernel runs before filter_slow_kernel).
template <int KB>
__global__ __launch_bounds__(64 * F_SLOW_WAVES) void filter_overflow_kernel(
    const float* __restrict__ Q, const float* __restrict__ C, int nc, int k,
    const int32_t* __restrict__ mask_rowptr, const int32_t* __restrict__ mask_col, const int* __restrict__ olist,
    int* __restrict__ counters, const int* __restrict__ wcnt, const uint4* __restrict__ wlist, int wcap,
    int tiles_per_range, const int* __restrict__ outl, int* __restrict__ flist,
    int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
    __shared__ unsigned long long s_l[F_SLOW_WAVES][F_CAPQ];
    __shared__ unsigned long long s_m[F_SLOW_WAVES * 128];
    __shared__ int32_t s_mask[F_SLOW_MASK_LDS];
    __shared__ int s_ids[F_OVER_IDS];
    __shared__ int s_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_over = counters[1];
    const int gpr = tiles_per_range >> 2;
    for (int j = blockIdx.x; j < n_over; j += gridDim.x) {       // uniform per workgroup
        const int q = olist[j];
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        auto decode = [&](const uint4 t) {     // the word's set bits -> candidate ids (the final kernel's arithmetic)
            const int wi = (int)t.x, rg = wi / gpr, g = wi - rg * gpr;
            const int cbase = ((rg >> 1) * tiles_per_range + 4 * g) * 32 + 4 * (rg & 1);
            unsigned long long x = (unsigned long long)t.z | ((unsigned long long)t.w << 32);
            while (x != 0ull) {
                const int pz = __clzll((long long)x);
                x &= ~(0x8000000000000000ull >> pz);
                const int r = pz & 15, dst = atomicAdd(&s_n, 1);
                if (dst < F_OVER_IDS) s_ids[dst] = cbase + (pz >> 4) * 32 + (r & 3) + 8 * (r >> 2);
            }
        };
        const int n_app = wcnt[q];
        for (int e = threadIdx.x; e < min(n_app, wcap); e += 64 * F_SLOW_WAVES) decode(wlist[(size_t)q * wcap + e]);
        __syncthreads();
        int n_ids = s_n;
        const int n_out = outl ? outl[0] : 0;
        __syncthreads();
        if (n_app > wcap || n_ids + n_out > F_OVER_IDS) {             // not everything pass 2 found is here: the exact slow queue
            if (threadIdx.x == 0) flist[atomicAdd(counters, 1)] = q;
            continue;
        }
        // the clipped candidate rows, unless pass 2 let them through (a wave per row, round robin)
        for (int o = wave; o < n_out; o += F_SLOW_WAVES) {
            const int id = outl[1 + o];
            bool has = false;
            for (int e = lane; e < n_ids; e += 64) has |= s_ids[e] == id;
            if (__ballot(has) == 0ull && lane == 0) s_ids[atomicAdd(&s_n, 1)] = id;
        }
        __syncthreads();
        n_ids = s_n;
        slow_topk<KB, true>(Q, C, nc, k, mask_rowptr, mask_col, q, s_l, s_m, s_mask, lane, wave, out_idx, out_val, 0, 0, nullptr,
                            s_ids, n_ids);
    }
}

// one wave per flagged query whose candidates were split: the S sorted partial lists -> the outputs
__global__ __launch_bounds__(256) void filter_slow_merge_kernel(const int* __restrict__ flist, const int* __restrict__ n_flagged,
                                                               int want_splits, const unsigned long long* __restrict__ parts,
                                                               int k, int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nf = *n_flagged;
    const int S = slow_splits(nf, want_splits);
    if (S <= 1) return;
    for (int j = blockIdx.x * 4 + wave; j < nf; j += gridDim.x * 4) {
        Cand y0, y1;
        sort_best_k(parts + (size_t)j * S * 128, S * 128, lane, k, y0, y1);
        const int q = flist[j];
        if (lane < k) {
            const bool ok = y0.i != INT_MAX;
            out_idx[(size_t)q * k + lane] = ok ? (int64_t)y0.i : (int64_t)-1;
            if (out_val) out_val[(size_t)q * k + lane] = ok ? y0.v : -INFINITY;
        }
        if (64 + lane < k) {
            const bool ok = y1.i != INT_MAX;
            out_idx[(size_t)q * k + 64 + lane] = ok ? (int64_t)y1.i : (int64_t)-1;
            if (out_val) out_val[(size_t)q * k + 64 + lane] = ok ? y1.v : -INFINITY;
        }
    }
}

struct FilterPlan {
    int n_stages, qblocks, nq_pad, R, spr, n_groups;   // spr: 64-candidate stages per range
    int p1_stride;                                     // pass 1 walks every p1_stride-th stage
    int wcap;                                          // sparse: word-list entries per query
    bool sparse;                                       // pass 2 -> word lists instead of rows of words
    size_t bits_bytes;                                 // the region pass 2 writes (rows, or counters + lists)
};
inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }
inline FilterPlan filter_plan(int nq, int nc) {
    FilterPlan p;
    p.n_stages = cdiv_i(nc, 64);
    p.qblocks = cdiv_i(nq, F_QWG);
    p.nq_pad = p.qblocks * F_QWG;
    // ranges: 8..16 (256..512 group maxima per query), whole groups of 4 stages per range (the register ring; two
    // 64-bit words of pass / fail bits): the most ranges (tightest bound, most head-room for heavily masked queries:
    // k + m must not exceed the number of groups) whose padding stays within 6 % of the least padded split
    long best = -1;
    for (int R = F_MAXR; R >= F_MINR; --R) {
        const int spr = (cdiv_i(p.n_stages, R) + 3) & ~3;
        const long cost = (long)cdiv_i(p.n_stages, spr) * spr;
        if (best < 0 || cost < best) best = cost;
    }
    p.spr = p.R = 0;
    for (int R = F_MAXR; R >= F_MINR && p.R == 0; --R) {
        const int spr = (cdiv_i(p.n_stages, R) + 3) & ~3, reff = cdiv_i(p.n_stages, spr);
        if ((long)reff * spr * 100 <= best * 106) { p.spr = spr; p.R = reff; }
    }
    p.n_groups = 32 * p.R;
    p.sparse = nc >= F_SPARSE_NC;
    // Pass 1 on every second stage of large candidate sets: the maxima of HALF the candidates still bound the (k + m)-th best
    // score from below (k + m distinct candidates reach the bound), about twice as deep in the ranking -- ~2 (k + m) survivors
    // plus what the 2 eps margin lets through -- for half the MFMAs of the pass.  The word lists and the final kernel's
    // survivor slots are twice as long there (F_WCAP2): with 256 slots the SMOOTHED embeddings a propagation produces (one
    // candidate of huge norm sets eps: ~120 survivors per query at stride 1, ~245 at stride 2; tools/topk_survivor_model.py)
    // overflowed for a third of the queries and those fell to the exact slow path -- 30 x slower (profiles/
    // r03_topk_pass1_stride_ab.log, r03_bench_line_p1s2.json).  Small candidate sets keep stride 1: their passes are short and
    // heavy users (k + m near the slot count) are relatively more frequent in real data.
    p.p1_stride = MMREC_TF_P1S > 0 ? MMREC_TF_P1S : (p.sparse && nc >= F_P1S2_NC ? 2 : 1);
    p.wcap = F_WLIST_X * (p.p1_stride >= 2 ? F_WCAP2 : F_WCAP);      // list entries; the final kernel has F_WCAP(2) slots
    p.bits_bytes = p.sparse ? ((((size_t)nq * 4 + 255) & ~(size_t)255) + (size_t)nq * p.wcap * 16)
                            : (size_t)nq * p.R * p.spr * 8;
    return p;
}
inline size_t al256f(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool filter_clips(int nc) { return nc >= F_P1S2_NC && !MMREC_TF_NOCLIP; }

}  // namespace

bool topk64_filter_applicable(int nq, int nc, int kd, int k) {
    return (kd == 64 || kd == 128) && k <= 128 && nc >= F_MIN_NC && nc <= 1000000 && nq >= 1;
}

// The candidate side of a call -- column sums / max |c| (stats), the centred fp16 copy Cs and the largest centred row norm
// -- depends on C only.  A caller that ranks several query blocks against the SAME table (the batches of one evaluation and
// its valid / test pair, trainer.py:298-310: weights are frozen) prepares it once (`prepared`: 1 KiB of statistics + Cs) and
// hands it to every call; without it a call prepares its own copy inside its workspace.
// Layout: stats (1 KiB) | Cs | where rows are clipped (nc >= F_P1S2_NC): the rows' norms (fp32) | their histogram.
size_t topk64_filter_prepared_bytes(int nc, int kd) {
    const size_t n_pad = (size_t)cdiv_i(nc, 64) * 64;
    return ST_WORDS * 4 + al256f(n_pad * 2 * kd) + (filter_clips(nc) ? al256f(n_pad * 4) + F_NHIST * 4 : 0);
}

int topk64_filter_prepare(const float* C, int nc, int kd, void* prepared, hipStream_t s) {
    const int n_stages = cdiv_i(nc, 64);
    float* stats = static_cast<float*>(prepared);
    uint4* Cs = reinterpret_cast<uint4*>(static_cast<char*>(prepared) + ST_WORDS * 4);
    unsigned* nmax = reinterpret_cast<unsigned*>(stats) + ST_NMAX;
    const bool clip = filter_clips(nc);
    const size_t n_pad = (size_t)n_stages * 64;
    float* cnorm = clip ? reinterpret_cast<float*>(reinterpret_cast<char*>(Cs) + al256f(n_pad * 2 * kd)) : nullptr;
    int* hist = clip ? reinterpret_cast<int*>(reinterpret_cast<char*>(cnorm) + al256f(n_pad * 4)) : nullptr;
    hipError_t e = hipMemsetAsync(stats, 0, ST_WORDS * 4, s);
    if (e != hipSuccess) return (int)e;
    if (clip && (e = hipMemsetAsync(hist, 0, F_NHIST * 4, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(filter_stats_kernel, dim3(cdiv_i(nc, 128)), dim3(256), 0, s, C, nc, kd / 4, stats);
    if (kd == 64)
        hipLaunchKernelGGL((filter_convert_kernel<true, 8>), dim3(n_stages * 64 * 8 / 256), dim3(256), 0, s, C, nc, n_stages * 64,
                           stats, Cs, cnorm, nmax, (int*)nullptr, (int*)nullptr);
    else
        hipLaunchKernelGGL((filter_convert_kernel<true, 16>), dim3(n_stages * 64 * 16 / 256), dim3(256), 0, s, C, nc, n_stages * 64,
                           stats, Cs, cnorm, nmax, (int*)nullptr, (int*)nullptr);
    if (clip) {
        hipLaunchKernelGGL(filter_norm_hist_kernel, dim3(cdiv_i(nc, 4096)), dim3(256), 0, s, cnorm, nc, hist);
        if (kd == 64)
            hipLaunchKernelGGL(filter_clip_kernel<8>, dim3(cdiv_i(nc, 4096)), dim3(256), 0, s, C, nc, cnorm, hist, stats, Cs);
        else
            hipLaunchKernelGGL(filter_clip_kernel<16>, dim3(cdiv_i(nc, 4096)), dim3(256), 0, s, C, nc, cnorm, hist, stats, Cs);
    }
    MMREC_RETURN_LAUNCH_STATUS();
}

size_t topk64_filter_workspace_bytes(int nq, int nc, int kd, int k) {
    const FilterPlan p = filter_plan(nq, nc);
    return al256f((size_t)p.nq_pad * 2 * kd) + topk64_filter_prepared_bytes(nc, kd) + al256f((size_t)p.nq_pad * 4) + 256 +
           al256f((size_t)nq * p.n_groups * 4) + 3 * al256f((size_t)nq * 4) + al256f(p.bits_bytes) +
           al256f((size_t)F_SLOW_PARTS * 128 * 8) + (p.sparse ? al256f((size_t)nq * 4) : 0);
}

namespace {
template <int KB>
int filter_launch_kb(const float* Q, const float* C, int nq, int nc, const int32_t* mask_rowptr,
                     const int32_t* mask_col, int k, int64_t* out_idx, float* out_val, void* workspace,
                     const void* prepared, const FilterHint& hint, hipStream_t s) {
    constexpr int kd = 64 * KB;
    const FilterPlan p = filter_plan(nq, nc);
    char* ws = static_cast<char*>(workspace);
    uint4* Qs = reinterpret_cast<uint4*>(ws);          ws += al256f((size_t)p.nq_pad * 2 * kd);
    char* own = ws;                                    ws += topk64_filter_prepared_bytes(nc, kd);   // used when `prepared` is null
    float* qnorm = reinterpret_cast<float*>(ws);       ws += al256f((size_t)p.nq_pad * 4);
    int* n_flagged = reinterpret_cast<int*>(ws);       ws += 256;                                // length of the slow queue
    unsigned* gkeys = reinterpret_cast<unsigned*>(ws); ws += al256f((size_t)nq * p.n_groups * 4);
    float* thr = reinterpret_cast<float*>(ws);         ws += al256f((size_t)nq * 4);
    int* flag = reinterpret_cast<int*>(ws);            ws += al256f((size_t)nq * 4);
    int* flist = reinterpret_cast<int*>(ws);           ws += al256f((size_t)nq * 4);
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(ws);   // [nq][R][2][spr / 2], or:
    int* wcnt = reinterpret_cast<int*>(ws);                                 // sparse: [nq] counters, then
    uint4* wlist = reinterpret_cast<uint4*>(ws + al256f((size_t)nq * 4));   //         [nq][wcap] entries
    ws += al256f(p.bits_bytes);
    unsigned long long* parts = reinterpret_cast<unsigned long long*>(ws);  // [F_SLOW_PARTS][128]
    ws += al256f((size_t)F_SLOW_PARTS * 128 * 8);
    int* olist = p.sparse ? reinterpret_cast<int*>(ws) : nullptr;           // sparse: overflow queue
    if (!prepared) {
        const int rc = topk64_filter_prepare(C, nc, kd, own, s);
        if (rc != 0) return rc;
        prepared = own;
    }
    const float* stats = static_cast<const float*>(prepared);
    const unsigned* cmax = reinterpret_cast<const unsigned*>(prepared) + ST_NMAX;
    const uint4* Cs = reinterpret_cast<const uint4*>(static_cast<const char*>(prepared) + ST_WORDS * 4);
    const int* outl = filter_clips(nc) ? static_cast<const int*>(prepared) + ST_NOUT : nullptr;   // clipped rows (prepare)
    // the query-side conversion also zeroes the call's counters (slow-queue length, word-list lengths): no memset launches
    hipLaunchKernelGGL((filter_convert_kernel<false, 8 * KB>), dim3(p.nq_pad * 8 * KB / 256), dim3(256), 0, s, Q, nq, p.nq_pad,
                       stats, Qs, qnorm, (unsigned*)nullptr, n_flagged, p.sparse ? wcnt : (int*)nullptr);
    PassArgs a{Qs, Cs, nq, nc, p.n_stages, p.spr, p.n_groups, p.p1_stride, p.wcap, gkeys, thr, bits, wcnt, wlist};
    const dim3 grid(p.qblocks, p.R);
    if (hint.ids && !hint.cold) {      // warm: the threshold from the caller's lists, no pass 1
        const size_t n_pad = (size_t)p.n_stages * 64;
        const float* cnorm = filter_clips(nc) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Cs) + al256f(n_pad * 2 * kd))
                                              : (const float*)nullptr;
        hipLaunchKernelGGL(filter_hint_bound_kernel<KB>, dim3(cdiv_i(nq, 4)), dim3(256), 0, s, Qs, Cs, (const int32_t*)hint.ids, hint.hk,
                           hint.rows, nq, nc, k, mask_rowptr, mask_col, qnorm, cmax, stats, cnorm, thr, flag);
    } else {
        hipLaunchKernelGGL((filter_pass_kernel<false, false, KB>), grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(filter_bound_kernel, dim3(cdiv_i(nq, 4)), dim3(256), 0, s, gkeys, p.n_groups, nq, nc, k,
                           mask_rowptr, qnorm, cmax, stats, kd, thr, flag);
    }
    if (p.sparse)
        hipLaunchKernelGGL((filter_pass_kernel<true, true, KB>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_pass_kernel<true, false, KB>), grid, dim3(256), 0, s, a);
    if (MMREC_TF_PROBE & 32) MMREC_RETURN_LAUNCH_STATUS();   // probe: the two passes only
    int32_t* hout = (hint.ids && hint.update) ? hint.ids : (int32_t*)nullptr;      // in place: read by the bound kernel above
    if (p.sparse && p.p1_stride >= 2)
        hipLaunchKernelGGL((filter_final_kernel<true, KB, F_WCAP2>), dim3(cdiv_i(nq, 4)), dim3(256), 0, s, Q, C, nq, nc, k, mask_rowptr,
                           mask_col, bits, wcnt, wlist, p.R, 2 * p.spr, flag, flist, n_flagged, out_idx, out_val, outl, olist, p.wcap, hout, hint.hk, hint.rows);
    else if (p.sparse)
        hipLaunchKernelGGL((filter_final_kernel<true, KB, F_WCAP>), dim3(cdiv_i(nq, 4)), dim3(256), 0, s, Q, C, nq, nc, k, mask_rowptr,
                           mask_col, bits, wcnt, wlist, p.R, 2 * p.spr, flag, flist, n_flagged, out_idx, out_val, outl, olist, p.wcap, hout, hint.hk, hint.rows);
    else
        hipLaunchKernelGGL((filter_final_kernel<false, KB, F_CAPQ>), dim3(cdiv_i(nq, 4)), dim3(256), 0, s, Q, C, nq, nc, k, mask_rowptr,
                           mask_col, bits, wcnt, wlist, p.R, 2 * p.spr, flag, flist, n_flagged, out_idx, out_val, outl, olist, p.wcap, hout, hint.hk, hint.rows);
    if (p.sparse)
        hipLaunchKernelGGL(filter_overflow_kernel<KB>, dim3(256), dim3(64 * F_SLOW_WAVES), 0, s, Q, C, nc, k, mask_rowptr, mask_col,
                           olist, n_flagged, wcnt, wlist, p.wcap, 2 * p.spr, outl, flist, out_idx, out_val);
    const int want = nc >= F_SLOW_SPLIT_NC ? 16 : 1;
    hipLaunchKernelGGL(filter_slow_kernel<KB>, dim3(256), dim3(64 * F_SLOW_WAVES), 0, s, Q, C, nc, k, mask_rowptr, mask_col, flist,
                       n_flagged, out_idx, out_val, want, parts);
    if (want > 1)
        hipLaunchKernelGGL(filter_slow_merge_kernel, dim3(64), dim3(256), 0, s, flist, n_flagged, want, parts, k, out_idx,
                           out_val);
    if (hout || hint.queue_counts)      // the lists of the queue-served queries + the caller's queue counters, one launch
        hipLaunchKernelGGL(filter_hint_from_out_kernel, dim3(hout ? 64 : 1), dim3(256), 0, s, flist, olist, n_flagged, out_idx, k, hout,
                           hint.hk, hint.rows, hint.queue_counts);
    MMREC_RETURN_LAUNCH_STATUS();
}
}  // namespace

int topk64_filter_launch(const float* Q, const float* C, int nq, int nc, int kd, const int32_t* mask_rowptr,
                         const int32_t* mask_col, int k, int64_t* out_idx, float* out_val, void* workspace,
                         const void* prepared, hipStream_t s, const FilterHint& hint) {
    return kd == 64 ? filter_launch_kb<1>(Q, C, nq, nc, mask_rowptr, mask_col, k, out_idx, out_val, workspace, prepared, hint, s)
                    : filter_launch_kb<2>(Q, C, nq, nc, mask_rowptr, mask_col, k, out_idx, out_val, workspace, prepared, hint, s);
}
