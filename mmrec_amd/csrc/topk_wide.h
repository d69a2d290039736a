// Score + mask + top-K for WIDE rows (kd = 192 ... 4096: the kNN builds over raw features, freedom.py:79-91 -- 500,000 x
// 4096 at config 5 -- and GRCN's 192-wide evaluation) by an fp16 matrix-core pass + exact fp32 refinement.
// Included by topk.hip after select_topk_kernel (its own translation unit's anonymous namespace).
//
// Why: fp32-input MFMA is 1/16 of the 16-bit rate; the materialised fp32 path spent 15.9 s (79 % of ITS peak) on the two
// item-item graphs of config 5.  The K-loop form of topk_filter.hip's idea: scores are needed exactly only for the few
// candidates that can be in the top-k.
//
// How, per block of queries:
//   convert   Q block and (once per call) C to fp16, every query row scaled by its own power of two, C by one; norms of the
//             ROUNDED rows ride along.  |approx - exact| <= eps_q := (1.0e-3 + kd 2^-24) |q| max|c| + 2^-25 sqrt(kd) (|q| + max|c|)
//             (fp16 rounding of both operands by Cauchy-Schwarz, fp32 accumulation of kd products, the absolute floor of
//             elements below fp16's normal range) -- a worst-case bound for any input.
//   gemm      S~ = Qh Ch^T on v_mfma_f32_32x32x16_f16, 128 x 128 tiles, both operands by LDS-DMA (the byte layout of
//             gemm_nt_kernel: a 64-half K tile is the same 128 B per row as its 32-float tile), fp32 block in the workspace.
//   select    the K' = 64 best APPROXIMATE scores per query with their ids (select_topk_kernel: masks, ties, sorted).
//   refine    one wave per query.  If v~[K'-1] <= v~[k-1] - 2 eps_q, no candidate outside the 64 can reach the exact top-k
//             (its exact score is below v~[K'-1] + eps <= v~[k-1] - eps <= the exact scores of k listed candidates): the
//             64 are re-scored EXACTLY in fp32 from the original Q and C rows, sorted (score desc, id asc), cut to k.
//             Otherwise (closely packed scores, massive ties) the query is queued ...
//   rescue    ... and the queued queries of the block go through the exact fp32 path (gemm_nt_kernel + select_topk_kernel on
//             the gathered rows): grids sized for the whole block, workgroups past the queue's device-side length return
//             at once.  No host synchronisation; correctness never depends on the margin test passing.
#pragma once

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 w_half8;
constexpr int W_KP = 64;          // approximate candidates kept per query (select_topk_kernel's limit)
constexpr int W_MAX_KD = 4096;    // a query row lives in 16 float4 registers per lane in the refine kernel
constexpr int W_MAX_K = 32;

__device__ __forceinline__ unsigned w_f2key(float f) {   // monotone for f >= 0
    return __float_as_uint(f);
}
// power of two bringing magnitudes <= mx below 2^8: a scaled score is then at most kd 2^16 < 2.7e8 in magnitude for kd <= 4096 --
// well inside the reference's mask sentinel (-1e10; the refine kernel reads "<= -1e9" as masked), and every element down to
// 2^-22 of the row's largest is a normal fp16 number
__device__ __forceinline__ float w_scale_for(float mx) {
    if (!(mx > 0.f)) return 1.f;
    int ex;
    frexpf(mx, &ex);
    return ldexpf(1.f, min(8 - ex, 120));
}

// max |x_ij| over the table -> *amax_key (zeroed by the caller); 256 rows per workgroup
__global__ __launch_bounds__(256) void wide_absmax_kernel(const float* __restrict__ X, size_t n4, unsigned* __restrict__ amax_key) {
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(X)[e];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(amax_key, w_f2key(mx));
}

// X [n][kd] fp32 -> Xh [n][kdp] fp16 (kdp = kd rounded up to 64, zero filled), one wave per row.  ROWSCALE: the row's own
// power-of-two scale (queries), else the table's (from *amax_key).  norm[row] = norm of the ROUNDED scaled row, inflated by
// 1.0005; cands also fold it into *nmax_key.
template <bool ROWSCALE>
__global__ __launch_bounds__(256) void wide_convert_kernel(const float* __restrict__ X, int n, int kd, int kdp,
                                                          const unsigned* __restrict__ amax_key, _Float16* __restrict__ Xh,
                                                          float* __restrict__ norm, unsigned* __restrict__ nmax_key) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + (size_t)row * kd;
    float scale;
    if (ROWSCALE) {
        float mx = 0.f;
        for (int c = lane; c < kd; c += 64) mx = fmaxf(mx, fabsf(x[c]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        scale = w_scale_for(mx);
    } else {
        scale = w_scale_for(__uint_as_float(*amax_key));
    }
    float ss = 0.f;
    for (int c = lane; c < kdp; c += 64) {
        const _Float16 h = (_Float16)(c < kd ? x[c] * scale : 0.f);
        Xh[(size_t)row * kdp + c] = h;
        const float back = (float)h;
        ss = fmaf(back, back, ss);
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss) * 1.0005f;
    if (lane == 0) {
        if (norm) norm[row] = nrm;
        if (nmax_key) atomicMax(nmax_key, w_f2key(nrm));
    }
}

// S[M, :ncols] = Ah[M, K] Bh[N, K]^T, fp16 operands, fp32 accumulation and output; K % 64 == 0.  gemm_nt_kernel's structure
// and byte layout (128 x 128 tile, wave w = rows 32 w ... + 31 x all 128 columns, 128-B row pieces by LDS-DMA into a
// double buffer with the source-side bank swizzle, one barrier per K tile), 16 MFMAs of 32 x 32 x 16 per wave and tile.
__global__ __launch_bounds__(256, 2) void wide_gemm_nt_f16_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                                                 float* __restrict__ S, int M, int N, int K, int lds_,
                                                                 int ncols) {
    __shared__ __attribute__((aligned(1024))) float As0[128 * 32], As1[128 * 32], Bs0[128 * 32], Bs1[128 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int rows_a = min(128, M - m0), rows_b = max(0, min(128, N - n0));
    const i32x4 ra = raw_rsrc(A + (size_t)m0 * K, (unsigned)rows_a * (unsigned)K * 2u);
    const i32x4 rb = raw_rsrc(B + (size_t)n0 * K, (unsigned)rows_b * (unsigned)K * 2u);
    int vo[4];   // this wave's 4 pieces (8 rows each) of either operand: rows 32w + 8j + lane/8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        vo[j] = r * K * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](float* as, float* bs, int t) {
        const int so = t * 128;                     // 64 halves = 128 B per row and K tile
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<false>(ra, lds_addr(as + (4 * wave + j) * 256), vo[j], so);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<false>(rb, lds_addr(bs + (4 * wave + j) * 256), vo[j], so);
    };
    const int i = lane & 31, h = lane >> 5, g = (i >> 1) & 7;
    int ko[4];   // 16-B chunk (8 halves: k = 16 s + 8 h ... + 7) of MFMA step s, swizzled
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) ko[k8] = ((2 * k8 + h) ^ g) << 2;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    auto compute = [&](const float* as, const float* bs) {
        const float* xa = as + (32 * wave + i) * 32;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const w_half8 a = *reinterpret_cast<const w_half8*>(xa + ko[k8]);
            w_half8 b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const w_half8*>(bs + (32 * t + i) * 32 + ko[k8]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[t], 0, 0, 0);
        }
    };
    const int T = K / 64;
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
                if (col < ncols)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), rs, lane_off, (rr * lds_ + 32 * t) * 4, 0);
            }
        }
    }
}

// One wave per query of the block: margin test on the approximate list, exact fp32 re-scoring of its 64 candidates, sort,
// cut to k -- or the query joins the rescue queue (qlist[atomicAdd(n_queued)] = query, block-relative rows in the same slot).
// S~ is in scaled units: approx = sq * sc * exact, so the exact-unit threshold test is done on the approximate values with
// eps in scaled units (qnorm, cmax are norms of the scaled rows).
__global__ __launch_bounds__(256) void wide_refine_kernel(
    const float* __restrict__ Q, const float* __restrict__ C, int q0, int rows, int nc, int kd, int k, int kp,
    const int64_t* __restrict__ idx_a, const float* __restrict__ val_a,          // [rows][kp] approximate lists (block-relative rows)
    const float* __restrict__ qnorm, const unsigned* __restrict__ cmax_key,
    int64_t* __restrict__ out_idx, float* __restrict__ out_val, int* __restrict__ qlist, int* __restrict__ n_queued) {
    const int lane = threadIdx.x & 63, ql = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= rows) return;
    const int q = q0 + ql;
    const float va = lane < kp ? val_a[(size_t)ql * kp + lane] : -INFINITY;
    const int ia = lane < kp ? (int)idx_a[(size_t)ql * kp + lane] : -1;
    const float cmax = __uint_as_float(*cmax_key), qn = qnorm[ql];
    const float kb = (float)kd * (1.f / 64.f);
    const float eps = qn * cmax * (1.0e-3f + (float)kd * 6.0e-8f + 4.0e-6f * kb) + 2.4e-7f * sqrtf(kb) * (qn + cmax);
    const float v_k = __shfl(va, k - 1, 64), v_last = __shfl(va, kp - 1, 64);
    // every candidate is listed (nc <= kp), or the list's tail is masked / clearly below the k-th: the 64 hold the exact top-k
    const bool safe = nc <= kp || v_last <= -1e9f || v_last <= v_k - 2.f * eps;
    if (!safe) {
        if (lane == 0) qlist[atomicAdd(n_queued, 1)] = q;
        return;
    }
    // the query row: float4 slot lane + 64 j, j < 16 (kd <= 4096)
    const int n4 = kd >> 2;
    const float4* q4 = reinterpret_cast<const float4*>(Q + (size_t)q * kd);
    float4 qv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) qv[j] = lane + 64 * j < n4 ? q4[lane + 64 * j] : f4_zero();
    float mine = -INFINITY;
    for (int c = 0; c < kp; ++c) {
        const int id = __shfl(ia, c, 64);
        const float a = __shfl(va, c, 64);
        if (id < 0 || id >= nc) continue;                   // uniform
        if (a <= -1e9f) {                                   // a masked item filling a short list: -1e10 like the reference
            if (lane == c) mine = -1e10f;
            continue;
        }
        const float4* c4 = reinterpret_cast<const float4*>(C + (size_t)id * kd);
        float4 cv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) cv[j] = lane + 64 * j < n4 ? c4[lane + 64 * j] : f4_zero();
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) part += f4_dot(qv[j], cv[j]);
        part = wave_sum(part);
        if (lane == c) mine = part;
    }
    Cand y0{mine, ia >= 0 && ia < nc && lane < kp ? ia : INT_MAX}, y1{-INFINITY, INT_MAX};
    if (y0.i == INT_MAX) y0.v = -INFINITY;
    bitonic128(y0, y1, lane);
    if (lane < k) {
        const bool ok = y0.i != INT_MAX;
        out_idx[(size_t)q * k + lane] = ok ? (int64_t)y0.i : (int64_t)-1;
        if (out_val) out_val[(size_t)q * k + lane] = ok ? y0.v : -INFINITY;
    }
}

// rows of the queued queries, gathered: Qf[j] = Q[qlist[j]] (one wave per row; waves past the queue's length return)
__global__ __launch_bounds__(256) void wide_gather_rows_kernel(const float* __restrict__ Q, int kd, const int* __restrict__ qlist,
                                                              const int* __restrict__ n_queued, float* __restrict__ Qf) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= *n_queued) return;
    const float4* src = reinterpret_cast<const float4*>(Q + (size_t)qlist[j] * kd);
    float4* dst = reinterpret_cast<float4*>(Qf + (size_t)j * kd);
    for (int c = lane; c < (kd >> 2); c += 64) dst[c] = src[c];
}

// where the fp16 pass pays: the fp32 GEMM it replaces grows with nc kd per query, the exact re-scoring of 64 rows with kd only --
// measured on kNN- and evaluation-shaped calls (profiles/r04_topk_wide_crossover.log): 7,050 x 192 / x 384 LOSE (0.51 against
// 0.31 ms, 0.57 against 0.47), 18,357 x 192 ties, 7,050 x 1024 and everything larger wins (x 1.5 ... 3.4) -> nc kd >= 2^22
inline bool topk_wide_applicable(int nq, int nc, int kd, int k) {
    return kd > 128 && kd <= W_MAX_KD && kd % 32 == 0 && k <= W_MAX_K && nc >= 4096 && nq >= 1 &&   // (kd % 32: the rescue's fp32 GEMM)
           (size_t)nc * (size_t)kd >= ((size_t)1 << 22);
}
inline size_t w_al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int w_pad64(int x) { return (x + 63) / 64 * 64; }
// on top of the materialised path's workspace (its S block is reused): fp16 copies, lists, norms, the rescue queue
inline size_t topk_wide_extra_bytes(int nc, int kd, int qb_rows) {
    const size_t kdp = (size_t)w_pad64(kd);
    return 256 + w_al256((size_t)nc * kdp * 2) + w_al256((size_t)qb_rows * kdp * 2) + w_al256((size_t)qb_rows * W_KP * 8) +
           w_al256((size_t)qb_rows * W_KP * 4) + 2 * w_al256((size_t)qb_rows * 4) + w_al256((size_t)qb_rows * kd * 4);
}

// S: the materialised path's [qb_rows][ldc] block; `extra`: topk_wide_extra_bytes
inline int topk_wide_launch(const float* Q, const float* C, int nq, int nc, int kd, const int32_t* mask_rowptr,
                            const int32_t* mask_col, int k, int64_t* out_idx, float* out_val, float* S, int ldc, int qb_rows,
                            char* extra, hipStream_t s) {
    const int kdp = w_pad64(kd), kp = W_KP;
    unsigned* keys = reinterpret_cast<unsigned*>(extra);                extra += 256;      // [0] max |c_ij|, [1] max row norm, [2] queue length
    _Float16* Ch = reinterpret_cast<_Float16*>(extra);                   extra += w_al256((size_t)nc * kdp * 2);
    _Float16* Qh = reinterpret_cast<_Float16*>(extra);                   extra += w_al256((size_t)qb_rows * kdp * 2);
    int64_t* idx_a = reinterpret_cast<int64_t*>(extra);                  extra += w_al256((size_t)qb_rows * kp * 8);
    float* val_a = reinterpret_cast<float*>(extra);                      extra += w_al256((size_t)qb_rows * kp * 4);
    float* qnorm = reinterpret_cast<float*>(extra);                      extra += w_al256((size_t)qb_rows * 4);
    int* qlist = reinterpret_cast<int*>(extra);                          extra += w_al256((size_t)qb_rows * 4);
    float* Qf = reinterpret_cast<float*>(extra);
    int* n_queued = reinterpret_cast<int*>(keys + 2);
    hipError_t e = hipMemsetAsync(keys, 0, 256, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wide_absmax_kernel, dim3(2048), dim3(256), 0, s, C, (size_t)nc * kd / 4, keys);
    hipLaunchKernelGGL(wide_convert_kernel<false>, dim3((nc + 3) / 4), dim3(256), 0, s, C, nc, kd, kdp, keys, Ch, (float*)nullptr,
                       keys + 1);
    for (int q0 = 0; q0 < nq; q0 += qb_rows) {
        const int rows = nq - q0 < qb_rows ? nq - q0 : qb_rows;
        if ((e = hipMemsetAsync(n_queued, 0, 4, s)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(wide_convert_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, Q + (size_t)q0 * kd, rows, kd, kdp,
                           (const unsigned*)nullptr, Qh, qnorm, (unsigned*)nullptr);
        hipLaunchKernelGGL(wide_gemm_nt_f16_kernel, dim3((rows + 127) / 128, (ldc + 127) / 128), dim3(256), 0, s, Qh, Ch, S, rows,
                           nc, kdp, ldc, ldc);
        // the 64 best approximate scores per query (masks applied, sorted); block-relative output rows
        hipLaunchKernelGGL(select_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, S, ldc, rows, q0, nc, kp,
                           (const float*)nullptr, 0, mask_rowptr, mask_col, idx_a - (size_t)q0 * kp, val_a - (size_t)q0 * kp,
                           (const int*)nullptr, (const int*)nullptr);
        hipLaunchKernelGGL(wide_refine_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, Q, C, q0, rows, nc, kd, k, kp, idx_a, val_a,
                           qnorm, keys + 1, out_idx, out_val, qlist, n_queued);
        // rescue: the queued queries through the exact fp32 path (grids for the whole block; idle workgroups return at once)
        hipLaunchKernelGGL(wide_gather_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, Q, kd, qlist, n_queued, Qf);
        gemm_nt_launch(Qf, C, nullptr, S, rows, nc, kd, ldc, ldc, s, n_queued);
        hipLaunchKernelGGL(select_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, S, ldc, rows, 0, nc, k,
                           (const float*)nullptr, 0, mask_rowptr, mask_col, out_idx, out_val, (const int*)qlist,
                           (const int*)n_queued);
    }
    return (int)hipGetLastError();
}

}  // namespace
