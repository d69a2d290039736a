// Fused Adam step: one pass over p, g, m, v instead of torch's multi-kernel foreach
// path.  HBM bound: 16 B read + 12 B written per element.  Dense-Adam semantics are kept (moments
// of rows with zero gradient keep decaying), as the reference's torch.optim.Adam over the full
// 7050 x 4096 feature tables requires.
// replaces: optimizer.step() common/trainer.py:189 (torch.optim.Adam, trainer.py:111-128).
#include "common.h"
#include <limits.h>

#ifndef MMREC_ADAM_NO_SETTLED    // probe build (-DMMREC_ADAM_NO_SETTLED=1; profiles/r04_c5_steady_state_settled_replay_ab.log): the catch-up always replays the full element-step
#define MMREC_ADAM_NO_SETTLED 0
#endif
#ifndef MMREC_ADAM_ILP
#define MMREC_ADAM_ILP 1      // float4 groups per thread and trip of the dense multi-tensor kernel
#endif
#ifndef MMREC_ADAM_NT
#define MMREC_ADAM_NT 0       // (with ILP >= 2) non-temporal loads of g / m / v and stores of m / v
#endif

namespace {

struct AdamArgs {
    float lr_over_bc1, beta1, beta2, eps, weight_decay, inv_bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
    m = fmaf(1.0f - a.beta1, g - m, m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(1.0f - a.beta2, g * g, v * a.beta2);       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    // sqrt and the quotient by the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (EVERY Adam kernel of this file goes through
    // here, so the row-lazy tables stay bit-identical to the dense kernel): the IEEE sequences were 22 of the 34
    // instructions of an element-step, and the row-lazy catch-up -- the same element-steps as dense Adam, replayed from
    // registers -- is bound by exactly that arithmetic (config 5 in steady state: 7.3 ms per training step, 4 of them here).
    // Against torch.optim.Adam the update differs by <= 2 ulp of a quantity bounded by lr (tests: 2e-6 relative).
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), a.inv_bc2_sqrt, a.eps);
    p = fmaf(-a.lr_over_bc1, m * __builtin_amdgcn_rcpf(denom), p);     // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// adam_one(p, 0, m, v) without weight decay, bit for bit (c (0 - m) = -c m; c 0 0 + v b2 = v b2 for v >= 0): what the
// row-lazy catch-up replays
__device__ __forceinline__ void adam_decay_one(float& p, float& m, float& v, const AdamArgs& a) {
    m = fmaf(-(1.0f - a.beta1), m, m);
    v = v * a.beta2;
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), a.inv_bc2_sqrt, a.eps);
    p = fmaf(-a.lr_over_bc1, m * __builtin_amdgcn_rcpf(denom), p);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, AdamArgs a) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
        const size_t i = n4 * 4 + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], a);
    }
}

// Graph-replay-safe form: step count and learning rate live in device memory.  `prepare` (one thread)
// increments the step and derives the two step-dependent scalars; `step_dev` then reads them, so a
// captured hipGraph replays correct bias corrections and picks up learning-rate changes.
__global__ void adam_prepare_kernel(long long* __restrict__ step, const float* __restrict__ lr, float beta1,
                                    float beta2, float* __restrict__ hyper) {
    const long long t = step[0] + 1;
    step[0] = t;
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    hyper[0] = (float)((double)lr[0] / bc1);
    hyper[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       size_t n, const float* __restrict__ hyper, float beta1,
                                                       float beta2, float eps, float weight_decay) {
    const AdamArgs a{hyper[0], beta1, beta2, eps, weight_decay, hyper[1]};
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], a);
    }
}

// Multi-tensor form: every parameter tensor of an optimizer step in ONE launch.  The descriptor table
// travels BY VALUE in the kernel-argument segment (no device table to upload, nothing to keep alive), a
// workgroup finds its tensor by scanning the (<= 24 entry) block prefix with scalar compares and then
// grid-strides inside that tensor's own block range.  Step-dependent scalars come per tensor (host form)
// or from the device pair written by adam_prepare_kernel (graph-replay form).
constexpr int ADAM_MULTI_MAX = 24;

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned long long n;
    float lr_over_bc1, inv_bc2_sqrt;
};

struct AdamTable {
    AdamTensor t[ADAM_MULTI_MAX];
    unsigned block_start[ADAM_MULTI_MAX + 1];
    int n_tensors;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable tab, const float* __restrict__ hyper,
                                                         float beta1, float beta2, float eps,
                                                         float weight_decay) {
    int ti = 0;
    while (ti + 1 < tab.n_tensors && blockIdx.x >= tab.block_start[ti + 1]) ++ti;   // uniform: scalar unit
    const AdamTensor d = tab.t[ti];
    const AdamArgs a{hyper ? hyper[0] : d.lr_over_bc1, beta1, beta2, eps, weight_decay,
                     hyper ? hyper[1] : d.inv_bc2_sqrt};
    const size_t nb = tab.block_start[ti + 1] - tab.block_start[ti];
    const size_t b = blockIdx.x - tab.block_start[ti];
    const size_t n4 = d.n / 4, stride = nb * 256;
    size_t i = b * 256 + threadIdx.x;
#if MMREC_ADAM_ILP >= 2
    // two float4 per thread and trip: eight loads in flight before the first store (round 6 A/B: tools/prof_adam_dense.py)
    for (; i + stride < n4; i += 2 * stride) {
        const size_t j = i + stride;
        float4 p0 = reinterpret_cast<float4*>(d.p)[i], p1 = reinterpret_cast<float4*>(d.p)[j];
#if MMREC_ADAM_NT      // the gradient and the moments are touched once per step: streamed past the caches
        typedef float f4v __attribute__((ext_vector_type(4)));
        auto ntl = [](const float* q, size_t k) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q) + k);
                                                  return make_float4(t.x, t.y, t.z, t.w); };
        const float4 g0 = ntl(d.g, i), g1 = ntl(d.g, j);
        float4 m0 = ntl(d.m, i), m1 = ntl(d.m, j), v0 = ntl(d.v, i), v1 = ntl(d.v, j);
#else
        const float4 g0 = reinterpret_cast<const float4*>(d.g)[i], g1 = reinterpret_cast<const float4*>(d.g)[j];
        float4 m0 = reinterpret_cast<float4*>(d.m)[i], m1 = reinterpret_cast<float4*>(d.m)[j];
        float4 v0 = reinterpret_cast<float4*>(d.v)[i], v1 = reinterpret_cast<float4*>(d.v)[j];
#endif
        adam_one(p0.x, g0.x, m0.x, v0.x, a); adam_one(p0.y, g0.y, m0.y, v0.y, a);
        adam_one(p0.z, g0.z, m0.z, v0.z, a); adam_one(p0.w, g0.w, m0.w, v0.w, a);
        adam_one(p1.x, g1.x, m1.x, v1.x, a); adam_one(p1.y, g1.y, m1.y, v1.y, a);
        adam_one(p1.z, g1.z, m1.z, v1.z, a); adam_one(p1.w, g1.w, m1.w, v1.w, a);
#if MMREC_ADAM_NT
        auto nts = [](float* q, size_t k, float4 x) { f4v t; t.x = x.x; t.y = x.y; t.z = x.z; t.w = x.w;
                                                      __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(q) + k); };
        reinterpret_cast<float4*>(d.p)[i] = p0; nts(d.m, i, m0); nts(d.v, i, v0);
        reinterpret_cast<float4*>(d.p)[j] = p1; nts(d.m, j, m1); nts(d.v, j, v1);
#else
        reinterpret_cast<float4*>(d.p)[i] = p0; reinterpret_cast<float4*>(d.m)[i] = m0; reinterpret_cast<float4*>(d.v)[i] = v0;
        reinterpret_cast<float4*>(d.p)[j] = p1; reinterpret_cast<float4*>(d.m)[j] = m1; reinterpret_cast<float4*>(d.v)[j] = v1;
#endif
    }
#endif
    for (; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(d.p)[i];
        const float4 gg = reinterpret_cast<const float4*>(d.g)[i];
        float4 mm = reinterpret_cast<float4*>(d.m)[i], vv = reinterpret_cast<float4*>(d.v)[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(d.p)[i] = pp;
        reinterpret_cast<float4*>(d.m)[i] = mm;
        reinterpret_cast<float4*>(d.v)[i] = vv;
    }
    if (b == 0 && threadIdx.x < (d.n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        adam_one(d.p[i], d.g[i], d.m[i], d.v[i], a);
    }
}

int adam_multi_launch(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                      int32_t n_tensors, const float* lr, const int64_t* step, const float* hyper_dev, float beta1,
                      float beta2, float eps, float weight_decay, hipStream_t s) {
    if (n_tensors < 0) return MMREC_ERR_BAD_ARG;
    if (n_tensors == 0) return 0;
    if (!p || !g || !m || !v || !n) return MMREC_ERR_BAD_ARG;
    for (int base = 0; base < n_tensors; base += ADAM_MULTI_MAX) {
        AdamTable tab;
        tab.n_tensors = 0;
        unsigned blocks = 0;
        for (int i = base; i < n_tensors && i < base + ADAM_MULTI_MAX; ++i) {
            if (n[i] < 0) return MMREC_ERR_BAD_ARG;
            if (n[i] == 0) continue;
            if (!p[i] || !g[i] || !m[i] || !v[i]) return MMREC_ERR_BAD_ARG;
            if ((reinterpret_cast<uintptr_t>(p[i]) | reinterpret_cast<uintptr_t>(g[i]) |
                 reinterpret_cast<uintptr_t>(m[i]) | reinterpret_cast<uintptr_t>(v[i])) & 15)
                return MMREC_ERR_BAD_ARG;
            AdamTensor& d = tab.t[tab.n_tensors];
            d.p = p[i], d.g = g[i], d.m = m[i], d.v = v[i], d.n = (unsigned long long)n[i];
            d.lr_over_bc1 = d.inv_bc2_sqrt = 0.f;
            if (!hyper_dev) {
                if (!lr || !step || step[i] < 1) return MMREC_ERR_BAD_ARG;
                const double bc1 = 1.0 - pow((double)beta1, (double)step[i]);
                const double bc2 = 1.0 - pow((double)beta2, (double)step[i]);
                d.lr_over_bc1 = (float)((double)lr[i] / bc1);
                d.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
            }
            size_t nb = ((size_t)n[i] / 4 + 255) / 256;
            if (nb > 256 * 16) nb = 256 * 16;
            if (nb < 1) nb = 1;
            tab.block_start[tab.n_tensors] = blocks;
            blocks += (unsigned)nb;
            ++tab.n_tensors;
        }
        if (tab.n_tensors == 0) continue;
        tab.block_start[tab.n_tensors] = blocks;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, s, tab, hyper_dev, beta1, beta2, eps,
                           weight_decay);
    }
    MMREC_RETURN_LAUNCH_STATUS();
}

// ---- row-lazy exact Adam for the trainable raw-feature tables -------------------------------
// A step touches <= 2B rows of an [n_items, F] table (FREEDOM's gathered-rows projection) but dense Adam has
// to stream the whole table and both moments: 63 GB per step for the 500K x 4096 table, ~80 % of the step.  A row
// whose gradient is zero evolves by a recurrence of its OWN state only (moments decay, the parameter keeps moving
// along the decaying first moment), so those updates can be postponed and replayed, in registers, the next time
// the row is needed: same instructions in the same order as the dense kernel (adam_one with g = 0), hence
// bit-identical results, but one read and one write of the row per touch instead of per step.
//   hist[j] = {lr_j / (1 - b1^j), 1 / sqrt(1 - b2^j)}: the step-dependent scalars of optimizer step j (1-based)
//   last_step[row]: optimizer steps already applied to the row
//   owner[row]: position of the row's first occurrence in this step's id list, INT_MAX if absent -- duplicates of a
//               row are served by exactly one workgroup, no sort / unique (those synchronise with the host)
__global__ void adam_hist_set_kernel(float2* __restrict__ hist, int t, float a, float b) { hist[t] = make_float2(a, b); }
// Graph-replay form: the step count and its two scalars are what adam_prepare_kernel left in device memory; a step
// beyond the table's capacity is NOT written and raises the (sticky) overflow flag the host checks per epoch.
__global__ void adam_hist_set_dev_kernel(float2* __restrict__ hist, int capacity, const long long* __restrict__ step,
                                         const float* __restrict__ hyper, int* __restrict__ overflow) {
    const long long t = step[0];
    if (t >= 1 && t < capacity) hist[t] = make_float2(hyper[0], hyper[1]);
    else *overflow = 1;
}

__global__ __launch_bounds__(256) void adam_rows_owner_kernel(const int64_t* __restrict__ ids, int n, int* __restrict__ owner) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && ids[i] >= 0) atomicMin(owner + ids[i], i);    // id < 0: "no row" (a slot another rank serves)
}

// Can the parameter still move?  While a row is skipped its update decays by b1 / sqrt(b2) per step; once
//     max_j |lr_j / (1 - b1^j)| * |m| / (sqrt(v) * b2^128 + eps)   (an upper bound of every update of the next <= 256 steps:
//     |m| and v only shrink, 1 / sqrt(1 - b2^j) >= 1; 1 % slack for the hardware sqrt / rcp and the roundings)
// is below a QUARTER ulp of p, p - update rounds to p in every one of those steps (half the spacing below p even when p is a
// power of two): adam_decay_one leaves p's bits alone and only the two moment decays remain -- 2 instructions per element-step
// instead of 2 + sqrt + rcp + 3.  p = 0 or denormal: never (any update moves it), unless m == 0 (no update at all).
__device__ __forceinline__ bool adam_param_settled(float p, float m, float v, float l_max, float v_fac, float eps) {
    if (m == 0.f) return true;
    const float bound = 1.01f * l_max * fabsf(m) * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_sqrtf(v), v_fac, eps));
    const float quarter_ulp = __int_as_float(__float_as_int(p) & 0x7f800000) * 2.98023224e-8f;     // 2^(e - 25)
    return bound < quarter_ulp;
}

// ---- OPT-IN fast-forward of a skipped row (config `lazy_adam_fast_forward`; NOT bit-identical to the dense kernel) ----
// In real arithmetic a row skipped from step s0 to s0 + n has m_n = b1^n m0, v_n = b2^n v0 and
//     p_n = p0 - m0 * sum_j w_j / (sqrt(v0) d_j + eps),   w_j = (lr_j / (1 - b1^(s0+j))) b1^j,   d_j = b2^(j/2) / sqrt(1 - b2^(s0+j))
// (j = 1 .. n).  w_j and d_j belong to the ROW (they depend on s0 and j only), and d_j hardly varies over the ~100 steps whose
// weight b1^j matters: with dbar = the w-weighted mean of d_j, delta_j = d_j / dbar - 1, y = sqrt(v0), A = y dbar + eps and
// u = y dbar / A (0 <= u <= 1),
//     sum_j w_j / (y d_j + eps) = (1 / A) sum_k (-u)^k M_k,        M_k = sum_j w_j delta_j^k   (M_0 = W, M_1 = 0)
// for EVERY ratio of eps to sqrt(v) -- a geometric series in u delta_j, cut after k = 6 with a relative remainder
// <= R = sum_j |w_j| |delta_j|^7 / |W| (tests/test_host_logic.py restates this in float64 against the direct sum).  The row's
// workgroup forms W, dbar, M_2 .. M_6 and R once from the per-step scalar table (one step per thread: the first 256 skipped
// steps; what follows weighs < b1^256), and every element then costs ~16 instructions whatever the gap.  The row is replayed
// EXACTLY instead when R > 1e-7 (the first ~100 optimizer steps, where the bias correction of v still moves by per cents per
// step: a row last visited then is replayed up to step FAST_FROM_STEP and advanced in closed form from there), when b1^256 >
// 1e-9, under weight decay, or when the gap is short (<= FAST_MIN_GAP steps: the replay is as cheap).
// Against the exact replay: p within 2e-6 of the distance it moved + half an ulp, m and v within 1e-6 sqrt(n) (the replay
// rounds them n times; the closed form once) -- inside north_star's 1e-4, outside "lazy == dense bit for bit".
constexpr int FAST_MIN_GAP = 12;
constexpr int FAST_FROM_STEP = 128;
struct FastRow {
    float W, dbar, M2, M3, M4, M5, M6, B1, B2;
    int ok;
};

__device__ __forceinline__ float block256_sum(float x, float* s_red /* [4] */) {
    x = wave_sum(x);
    __syncthreads();                                    // s_red free again
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__device__ __forceinline__ void adam_fast_row_scalars(FastRow& fr, const float2* __restrict__ hist, int s0, int t_now,
                                                      float beta1, float beta2, float* s_red) {
    const int n = t_now - s0, jp = (int)threadIdx.x + 1;
    const float l2b1 = log2f(beta1), l2b2 = log2f(beta2);
    float w = 0.f, d = 0.f;
    if (jp <= n) {
        const float2 h = hist[s0 + jp];
        w = h.x * exp2f((float)jp * l2b1);
        d = h.y * exp2f(0.5f * (float)jp * l2b2);
    }
    const float W = block256_sum(w, s_red);
    const float dbar = block256_sum(w * d, s_red) / W;
    const float de = jp <= n ? d / dbar - 1.f : 0.f;
    const float de2 = de * de, w2 = w * de2, w4 = w2 * de2;
    fr.W = W, fr.dbar = dbar;
    fr.M2 = block256_sum(w2, s_red);
    fr.M3 = block256_sum(w2 * de, s_red);
    fr.M4 = block256_sum(w4, s_red);
    fr.M5 = block256_sum(w4 * de, s_red);
    fr.M6 = block256_sum(w4 * de2, s_red);
    const float R = block256_sum(fabsf(w4 * de2 * de), s_red);
    __syncthreads();
    if (threadIdx.x < 2) s_red[threadIdx.x] = (float)pow((double)(threadIdx.x ? beta2 : beta1), (double)n);   // two lanes, once per row
    __syncthreads();
    fr.B1 = s_red[0], fr.B2 = s_red[1];
    const bool tail_ok = n <= 256 || exp2f(256.f * l2b1) < 1e-9f;
    fr.ok = tail_ok && fabsf(W) > 0.f && dbar > 0.f && R <= 1e-7f * fabsf(W) && isfinite(W) && isfinite(dbar);
}

__device__ __forceinline__ void adam_fast_one(float& p, float& m, float& v, const FastRow& fr, float eps) {
    const float y = __builtin_amdgcn_sqrtf(v) * fr.dbar;
    const float r = __builtin_amdgcn_rcpf(y + eps);
    const float u = y * r;
    float s = fmaf(-u, fr.M6, fr.M5);
    s = fmaf(-u, s, fr.M4);
    s = fmaf(-u, s, fr.M3);
    s = fmaf(-u, s, fr.M2);
    s = fmaf(u * u, s, fr.W);
    p = fmaf(-(m * r), s, p);
    m *= fr.B1;
    v *= fr.B2;
}

// ids == nullptr: every row (flush).  One workgroup per listed row.  The row stays in registers while the steps
// are replayed (columns in tiles of 4096 floats: 4 float4 per thread); the per-step scalars come through LDS in
// tiles of 256 steps (one global load per step and thread made the first version wait on memory 135 us per call).
// FAST: the opt-in closed form above for rows it serves; everything else takes the exact replay below.
template <bool FAST>
__global__ __launch_bounds__(256) void adam_rows_catchup_kernel(
    float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const int64_t* __restrict__ ids,
    int* __restrict__ owner, int F, int* __restrict__ last_step, const float2* __restrict__ hist, int t_now,
    float beta1, float beta2, float eps, float weight_decay, const long long* __restrict__ step_dev, int capacity) {
    __shared__ float2 s_h[256];
    __shared__ float s_lmax[4];
    if (step_dev) {                                     // graph-replay form: optimizer steps taken so far, from the device
        t_now = (int)step_dev[0];
        // a step beyond the scalar table was refused by adam_hist_set_dev_kernel (sticky overflow flag, reported by the host
        // at the next epoch): there is nothing valid to replay from -- leave the rows as they are, the error stays recoverable
        if (t_now >= capacity) return;
    }
    const int64_t row = ids ? ids[blockIdx.x] : (int64_t)blockIdx.x;
    if (row < 0) return;                                // "no row"
    if (ids && owner[row] != (int)blockIdx.x) return;   // a duplicate: the first occurrence does the work
    const int s0 = last_step[row];
    __syncthreads();                                    // everybody has read last_step / owner before they change
    if (ids && threadIdx.x == 0) owner[row] = INT_MAX;  // idle again (late duplicates see INT_MAX != their position)
    if (s0 >= t_now) return;
    const int f4 = F / 4;
    float4* p4 = reinterpret_cast<float4*>(p + (size_t)row * F);
    float4* m4 = reinterpret_cast<float4*>(m + (size_t)row * F);
    float4* v4 = reinterpret_cast<float4*>(v + (size_t)row * F);
    FastRow fr;
    fr.ok = 0;
    // a row last visited in the first FAST_FROM_STEP steps is replayed exactly up to that step and advanced in closed form from
    // there (early in a run the series is refused: the bias correction of v moves too fast)
    const int s_fast = max(s0, min(t_now, FAST_FROM_STEP));
    if (FAST && weight_decay == 0.f && t_now - s_fast > FAST_MIN_GAP)
        adam_fast_row_scalars(fr, hist, s_fast, t_now, beta1, beta2, s_lmax);
    const int t_exact = (FAST && fr.ok) ? s_fast : t_now;      // the exact replay covers steps s0 + 1 .. t_exact
    for (int c0 = 0; c0 < f4; c0 += 1024) {
        float4 pp[4], mm[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + threadIdx.x + 256 * u;
            if (c < f4) { pp[u] = p4[c]; mm[u] = m4[c]; vv[u] = v4[c]; }
            else pp[u] = mm[u] = vv[u] = f4_zero();
        }
        // A row that has never had a gradient (both moments exactly 0) does not move while it is skipped: adam_one(p, 0, 0, 0)
        // leaves m = v = 0 and subtracts lr * 0 / eps = 0 from p.  Without weight decay the replay of such a tile is the
        // identity -- most touches of the first epoch (500K rows, 4096 per step at config 5) -- and is not run.
        bool untouched = weight_decay == 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            untouched = untouched && mm[u].x == 0.f && mm[u].y == 0.f && mm[u].z == 0.f && mm[u].w == 0.f &&
                        vv[u].x == 0.f && vv[u].y == 0.f && vv[u].z == 0.f && vv[u].w == 0.f;
        if (__syncthreads_and(untouched)) continue;
        const float v_fac = 0.99f * powf(beta2, 128.f);         // sqrt(v) after <= 256 more decays, from below
        for (int jb = s0 + 1; jb <= t_exact; jb += 256) {
            __syncthreads();
            float lx = 0.f;
            if (jb + (int)threadIdx.x <= t_exact) {
                s_h[threadIdx.x] = hist[jb + threadIdx.x];
                lx = fabsf(s_h[threadIdx.x].x);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) lx = fmaxf(lx, __shfl_xor(lx, o, 64));
            if ((threadIdx.x & 63) == 0) s_lmax[threadIdx.x >> 6] = lx;
            __syncthreads();
            const float l_max = fmaxf(fmaxf(s_lmax[0], s_lmax[1]), fmaxf(s_lmax[2], s_lmax[3]));   // largest step size of the tile
            const int nj = min(256, t_exact - jb + 1);
            for (int j0 = 0; j0 < nj; j0 += 32) {              // 32 steps at a time: can p still move?  (per wave)
                const int j1 = min(j0 + 32, nj);
                bool settled = weight_decay == 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    settled = settled && adam_param_settled(pp[u].x, mm[u].x, vv[u].x, l_max, v_fac, eps) &&
                              adam_param_settled(pp[u].y, mm[u].y, vv[u].y, l_max, v_fac, eps) &&
                              adam_param_settled(pp[u].z, mm[u].z, vv[u].z, l_max, v_fac, eps) &&
                              adam_param_settled(pp[u].w, mm[u].w, vv[u].w, l_max, v_fac, eps);
                if (!MMREC_ADAM_NO_SETTLED && __all(settled)) { // the moments keep decaying, bit for bit as adam_decay_one does
                    const float nb1 = -(1.0f - beta1);
                    for (int j = j0; j < j1; ++j) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            mm[u].x = fmaf(nb1, mm[u].x, mm[u].x); vv[u].x = vv[u].x * beta2;
                            mm[u].y = fmaf(nb1, mm[u].y, mm[u].y); vv[u].y = vv[u].y * beta2;
                            mm[u].z = fmaf(nb1, mm[u].z, mm[u].z); vv[u].z = vv[u].z * beta2;
                            mm[u].w = fmaf(nb1, mm[u].w, mm[u].w); vv[u].w = vv[u].w * beta2;
                        }
                    }
                    continue;
                }
                for (int j = j0; j < j1; ++j) {
                    const float2 h = s_h[j];
                    const AdamArgs a{h.x, beta1, beta2, eps, weight_decay, h.y};
                    if (weight_decay == 0.f) {          // uniform
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            adam_decay_one(pp[u].x, mm[u].x, vv[u].x, a);
                            adam_decay_one(pp[u].y, mm[u].y, vv[u].y, a);
                            adam_decay_one(pp[u].z, mm[u].z, vv[u].z, a);
                            adam_decay_one(pp[u].w, mm[u].w, vv[u].w, a);
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        adam_one(pp[u].x, 0.f, mm[u].x, vv[u].x, a);
                        adam_one(pp[u].y, 0.f, mm[u].y, vv[u].y, a);
                        adam_one(pp[u].z, 0.f, mm[u].z, vv[u].z, a);
                        adam_one(pp[u].w, 0.f, mm[u].w, vv[u].w, a);
                    }
                }
            }
        }
        if (FAST && fr.ok) {                                    // (uniform over the workgroup) steps t_exact + 1 .. t_now
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                adam_fast_one(pp[u].x, mm[u].x, vv[u].x, fr, eps);
                adam_fast_one(pp[u].y, mm[u].y, vv[u].y, fr, eps);
                adam_fast_one(pp[u].z, mm[u].z, vv[u].z, fr, eps);
                adam_fast_one(pp[u].w, mm[u].w, vv[u].w, fr, eps);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + threadIdx.x + 256 * u;
            if (c < f4) { p4[c] = pp[u]; m4[c] = mm[u]; v4[c] = vv[u]; }
        }
    }
    if (threadIdx.x == 0) last_step[row] = t_now;
}

// Optimizer step t on the listed rows (all caught up to t - 1).  g[i] = gradient of OCCURRENCE i of the id list; the
// workgroup of a row's first occurrence (its owner) sums the row's occurrences itself, in position order: it scans the
// id list once (n ids from L2; duplicates are rare) into an LDS position list, then adds the gradient rows while the
// parameter row is in registers.  Deterministic (the former zero-fill + index_add_ pre-pass summed with float atomics)
// and two passes over the [n, F] gradient cheaper.  Dynamic LDS: n ints.
__global__ __launch_bounds__(256) void adam_rows_step_kernel(
    float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const int64_t* __restrict__ ids,
    int* __restrict__ owner, const float* __restrict__ g, int n, int F, int* __restrict__ last_step, int t, AdamArgs a,
    const long long* __restrict__ step_dev, const float* __restrict__ hyper_dev, int capacity) {
    // n == 0: g is already summed per owner slot (id lists too long for the LDS position list): only the own position
    extern __shared__ int s_pos[];
    if (step_dev) {                                     // graph-replay form (after adam_prepare_kernel of this step)
        t = (int)step_dev[0];
        a.lr_over_bc1 = hyper_dev[0];
        a.inv_bc2_sqrt = hyper_dev[1];
        if (t >= capacity) return;                      // no scalar-table entry for this step (overflow flag is up): no half-applied state
    }
    __shared__ int s_wave[4];
    __shared__ int s_total;
    const int64_t row = ids[blockIdx.x];
    if (row < 0 || owner[row] != (int)blockIdx.x) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    if (threadIdx.x == 0) owner[row] = INT_MAX;         // idle again
    for (int base = 0; base < n; base += 256) {         // ascending positions, compacted in order
        const int i = base + threadIdx.x;
        const bool hit = i < n && ids[i] == row;
        const unsigned long long b = __ballot(hit);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_total;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (hit) s_pos[off + __popcll(b & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) s_total += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (n == 0 && threadIdx.x == 0) { s_pos[0] = blockIdx.x; s_total = 1; }
    __syncthreads();
    const int cnt = s_total;
    const int f4 = F / 4;
    float4* p4 = reinterpret_cast<float4*>(p + (size_t)row * F);
    float4* m4 = reinterpret_cast<float4*>(m + (size_t)row * F);
    float4* v4 = reinterpret_cast<float4*>(v + (size_t)row * F);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int c = threadIdx.x; c < f4; c += 256) {
        float4 pp = p4[c], mm = m4[c], vv = v4[c];
        float4 gg = g4[(size_t)s_pos[0] * f4 + c];      // s_pos[0] == blockIdx.x
        for (int q = 1; q < cnt; ++q) gg = f4_add(gg, g4[(size_t)s_pos[q] * f4 + c]);
        adam_one(pp.x, gg.x, mm.x, vv.x, a);
        adam_one(pp.y, gg.y, mm.y, vv.y, a);
        adam_one(pp.z, gg.z, mm.z, vv.z, a);
        adam_one(pp.w, gg.w, mm.w, vv.w, a);
        p4[c] = pp; m4[c] = mm; v4[c] = vv;
    }
    if (threadIdx.x == 0) last_step[row] = t;
}

}  // namespace

extern "C" int mmrec_adam_hist_set(float* hist, int32_t t, float lr, float beta1, float beta2, mmrec_stream_t stream) {
    if (!hist || t < 1) return MMREC_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    hipLaunchKernelGGL(adam_hist_set_kernel, dim3(1), dim3(1), 0, mmrec_stream(stream), reinterpret_cast<float2*>(hist),
                       t, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_rows_owner(const int64_t* ids, int32_t n, int32_t* owner, mmrec_stream_t stream) {
    if (n < 0 || (n > 0 && (!ids || !owner))) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_rows_owner_kernel, dim3((n + 255) / 256), dim3(256), 0, mmrec_stream(stream), ids, n, owner);
    MMREC_RETURN_LAUNCH_STATUS();
}

static int rows_catchup(bool fast, float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                        int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                        const float* hist, int32_t t_now, float beta1, float beta2, float eps,
                        float weight_decay, mmrec_stream_t stream) {
    if (F <= 0 || (F & 3) || t_now < 0 || !p || !m || !v || !last_step || !hist) return MMREC_ERR_BAD_ARG;
    if (ids && !owner) return MMREC_ERR_BAD_ARG;
    const int blocks = ids ? n_ids : n_rows;
    if (blocks <= 0) return blocks < 0 ? MMREC_ERR_BAD_ARG : 0;
    if (fast)
        hipLaunchKernelGGL(adam_rows_catchup_kernel<true>, dim3(blocks), dim3(256), 0, mmrec_stream(stream), p, m, v, ids, owner,
                           F, last_step, reinterpret_cast<const float2*>(hist), t_now, beta1, beta2, eps, weight_decay,
                           (const long long*)nullptr, 0);
    else
        hipLaunchKernelGGL(adam_rows_catchup_kernel<false>, dim3(blocks), dim3(256), 0, mmrec_stream(stream), p, m, v, ids,
                           owner, F, last_step, reinterpret_cast<const float2*>(hist), t_now, beta1, beta2, eps, weight_decay,
                           (const long long*)nullptr, 0);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_hist_set_dev(float* hist, int32_t capacity, const int64_t* step_dev, const float* hyper_dev,
                                       int32_t* overflow, mmrec_stream_t stream) {
    if (!hist || capacity < 2 || !step_dev || !hyper_dev || !overflow) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(adam_hist_set_dev_kernel, dim3(1), dim3(1), 0, mmrec_stream(stream), reinterpret_cast<float2*>(hist),
                       capacity, reinterpret_cast<const long long*>(step_dev), hyper_dev, overflow);
    MMREC_RETURN_LAUNCH_STATUS();
}

static int rows_catchup_dev(bool fast, float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                            int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                            const float* hist, int32_t capacity, const int64_t* step_dev, float beta1,
                            float beta2, float eps, float weight_decay, mmrec_stream_t stream) {
    if (F <= 0 || (F & 3) || !p || !m || !v || !last_step || !hist || !step_dev || capacity < 2) return MMREC_ERR_BAD_ARG;
    if (ids && !owner) return MMREC_ERR_BAD_ARG;
    const int blocks = ids ? n_ids : n_rows;
    if (blocks <= 0) return blocks < 0 ? MMREC_ERR_BAD_ARG : 0;
    if (fast)
        hipLaunchKernelGGL(adam_rows_catchup_kernel<true>, dim3(blocks), dim3(256), 0, mmrec_stream(stream), p, m, v, ids, owner,
                           F, last_step, reinterpret_cast<const float2*>(hist), 0, beta1, beta2, eps, weight_decay,
                           reinterpret_cast<const long long*>(step_dev), capacity);
    else
        hipLaunchKernelGGL(adam_rows_catchup_kernel<false>, dim3(blocks), dim3(256), 0, mmrec_stream(stream), p, m, v, ids,
                           owner, F, last_step, reinterpret_cast<const float2*>(hist), 0, beta1, beta2, eps, weight_decay,
                           reinterpret_cast<const long long*>(step_dev), capacity);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_rows_catchup_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                           int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                                           const float* hist, int32_t t_now, float beta1, float beta2, float eps,
                                           float weight_decay, mmrec_stream_t stream) {
    return rows_catchup(false, p, m, v, ids, owner, n_ids, n_rows, F, last_step, hist, t_now, beta1, beta2, eps, weight_decay,
                        stream);
}
extern "C" int mmrec_adam_rows_fastforward_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                               int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                                               const float* hist, int32_t t_now, float beta1, float beta2, float eps,
                                               float weight_decay, mmrec_stream_t stream) {
    return rows_catchup(true, p, m, v, ids, owner, n_ids, n_rows, F, last_step, hist, t_now, beta1, beta2, eps, weight_decay,
                        stream);
}
extern "C" int mmrec_adam_rows_catchup_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                               int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                                               const float* hist, int32_t capacity, const int64_t* step_dev, float beta1,
                                               float beta2, float eps, float weight_decay, mmrec_stream_t stream) {
    return rows_catchup_dev(false, p, m, v, ids, owner, n_ids, n_rows, F, last_step, hist, capacity, step_dev, beta1, beta2,
                            eps, weight_decay, stream);
}
extern "C" int mmrec_adam_rows_fastforward_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                                   int32_t n_ids, int32_t n_rows, int32_t F, int32_t* last_step,
                                                   const float* hist, int32_t capacity, const int64_t* step_dev, float beta1,
                                                   float beta2, float eps, float weight_decay, mmrec_stream_t stream) {
    return rows_catchup_dev(true, p, m, v, ids, owner, n_ids, n_rows, F, last_step, hist, capacity, step_dev, beta1, beta2,
                            eps, weight_decay, stream);
}

extern "C" int mmrec_adam_rows_step_dev_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                            const float* g, int32_t n_ids, int32_t F, int32_t* last_step,
                                            int32_t capacity, const int64_t* step_dev, const float* hyper_dev, float beta1,
                                            float beta2, float eps, float weight_decay, int32_t presummed,
                                            mmrec_stream_t stream) {
    if (F <= 0 || (F & 3) || n_ids < 0 || capacity < 2) return MMREC_ERR_BAD_ARG;
    if (n_ids == 0) return 0;
    if (!p || !m || !v || !ids || !owner || !g || !last_step || !step_dev || !hyper_dev) return MMREC_ERR_BAD_ARG;
    if (!presummed && n_ids > MMREC_ADAM_ROWS_MAX_IDS) return MMREC_ERR_UNSUPPORTED;
    const AdamArgs a{0.f, beta1, beta2, eps, weight_decay, 0.f};
    hipLaunchKernelGGL(adam_rows_step_kernel, dim3(n_ids), dim3(256), presummed ? sizeof(int) : (size_t)n_ids * sizeof(int),
                       mmrec_stream(stream), p, m, v, ids, owner, g, presummed ? 0 : n_ids, F, last_step, 0, a,
                       reinterpret_cast<const long long*>(step_dev), hyper_dev, capacity);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_rows_step_f32(float* p, float* m, float* v, const int64_t* ids, int32_t* owner,
                                        const float* g, int32_t n_ids, int32_t F, int32_t* last_step, int32_t t,
                                        float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int32_t presummed, mmrec_stream_t stream) {
    if (F <= 0 || (F & 3) || t < 1 || n_ids < 0) return MMREC_ERR_BAD_ARG;
    if (n_ids == 0) return 0;
    if (!p || !m || !v || !ids || !owner || !g || !last_step) return MMREC_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    const AdamArgs a{(float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2))};
    if (!presummed && n_ids > MMREC_ADAM_ROWS_MAX_IDS) return MMREC_ERR_UNSUPPORTED;     // the position list lives in LDS
    hipLaunchKernelGGL(adam_rows_step_kernel, dim3(n_ids), dim3(256), presummed ? sizeof(int) : (size_t)n_ids * sizeof(int),
                       mmrec_stream(stream), p, m, v, ids, owner, g, presummed ? 0 : n_ids, F, last_step, t, a,
                       (const long long*)nullptr, (const float*)nullptr, 0);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_multi_step_f32(float* const* p, const float* const* g, float* const* m, float* const* v,
                                         const int64_t* n, int32_t n_tensors, const float* lr,
                                         const int64_t* step, float beta1, float beta2, float eps,
                                         float weight_decay, mmrec_stream_t stream) {
    if (!lr || !step) return MMREC_ERR_BAD_ARG;
    return adam_multi_launch(p, g, m, v, n, n_tensors, lr, step, nullptr, beta1, beta2, eps, weight_decay,
                             mmrec_stream(stream));
}

extern "C" int mmrec_adam_multi_step_dev_f32(float* const* p, const float* const* g, float* const* m,
                                             float* const* v, const int64_t* n, int32_t n_tensors,
                                             const float* hyper_dev, float beta1, float beta2, float eps,
                                             float weight_decay, mmrec_stream_t stream) {
    if (!hyper_dev) return MMREC_ERR_BAD_ARG;
    return adam_multi_launch(p, g, m, v, n, n_tensors, nullptr, nullptr, hyper_dev, beta1, beta2, eps,
                             weight_decay, mmrec_stream(stream));
}

extern "C" int mmrec_adam_prepare(int64_t* step_dev, const float* lr_dev, float beta1, float beta2,
                                  float* hyper_dev, mmrec_stream_t stream) {
    if (!step_dev || !lr_dev || !hyper_dev) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, mmrec_stream(stream),
                       reinterpret_cast<long long*>(step_dev), lr_dev, beta1, beta2, hyper_dev);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n,
                                       const float* hyper_dev, float beta1, float beta2, float eps,
                                       float weight_decay, mmrec_stream_t stream) {
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!p || !g || !m || !v || !hyper_dev) return MMREC_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return MMREC_ERR_BAD_ARG;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, mmrec_stream(stream), p, g, m, v,
                       (size_t)n, hyper_dev, beta1, beta2, eps, weight_decay);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay,
                                   int64_t step, mmrec_stream_t stream) {
    if (n < 0 || step < 1) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!p || !g || !m || !v) return MMREC_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return MMREC_ERR_BAD_ARG;  // float4 path needs 16-byte aligned tensors
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamArgs a{(float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2))};
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, mmrec_stream(stream), p, g, m, v,
                       (size_t)n, a);
    MMREC_RETURN_LAUNCH_STATUS();
}
