// Fused sampled scoring (P4): gather u/p/n rows, <u,p> - <u,n>, (log)sigmoid loss, and the
// scatter-add backward.   (a9)
// One 16-lane group (float4 per lane) per sample; 3 x 256-B gathers per sample forward, plus
// 3 x 256-B atomic read-modify-writes backward.  B = 2048 => launch-bound; report us/batch.
#include "common.h"

namespace {

__device__ __forceinline__ float neg_logsigmoid(float x) {  // -logsigmoid(x) = softplus(-x)
    return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void bpr_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ P, const float* __restrict__ N,
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
    const int64_t* __restrict__ neg, int batch, int d4, int variant, float* __restrict__ loss_i,
    float* __restrict__ coef) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    // d = 4*d4 floats per row (a multiple of 64): each 16-lane group walks the row in 64-float steps
    const size_t iu = (size_t)users[b] * d4, ip = (size_t)pos[b] * d4, in = (size_t)neg[b] * d4;
    float pp = 0.f, nn = 0.f;
    for (int c = lane16; c < d4; c += 16) {
        const float4 u = reinterpret_cast<const float4*>(U)[iu + c];
        pp += f4_dot(u, reinterpret_cast<const float4*>(P)[ip + c]);
        nn += f4_dot(u, reinterpret_cast<const float4*>(N)[in + c]);
    }
    // pos and neg scores are reduced separately, then subtracted (as the reference does)
    const float ps = row16_sum(pp);
    const float ns = row16_sum(nn);
    if (lane16 == 0) {
        const float x = ps - ns;
        float l, c;
        if (variant == MMREC_BPR_LOGSIG) {
            l = neg_logsigmoid(x);
            c = -1.0f / (1.0f + expf(x));  // -sigmoid(-x)
        } else {
            const float s = 1.0f / (1.0f + expf(-x));
            l = -logf(1e-10f + s);
            c = -(s * (1.0f - s)) / (1e-10f + s);
        }
        loss_i[b] = l;
        coef[b] = c;
    }
}

// The two halves of bpr_fwd_kernel for a COLUMN SLICE of the tables (feature-sliced multi-GPU layout): a rank
// holds d / P columns, <u, p> and <u, n> are sums over the ranks of these partial dot products (one small all-reduce by the
// caller), then the loss and d loss / d score are computed from the summed scores, replicated.
__global__ __launch_bounds__(256) void bpr_dots_kernel(
    const float* __restrict__ U, const float* __restrict__ P, const float* __restrict__ N,
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int batch, int d4,
    float* __restrict__ dots) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    const size_t iu = (size_t)users[b] * d4, ip = (size_t)pos[b] * d4, in = (size_t)neg[b] * d4;
    float pp = 0.f, nn = 0.f;
    for (int c = lane16; c < d4; c += 16) {      // slices of 8 / 16 / 32 columns: 2 / 4 / 8 live lanes, the rest add zeros
        const float4 u = reinterpret_cast<const float4*>(U)[iu + c];
        pp += f4_dot(u, reinterpret_cast<const float4*>(P)[ip + c]);
        nn += f4_dot(u, reinterpret_cast<const float4*>(N)[in + c]);
    }
    const float ps = row16_sum(pp);
    const float ns = row16_sum(nn);
    if (lane16 == 0) {
        dots[b] = ps;
        dots[batch + b] = ns;
    }
}

__global__ __launch_bounds__(256) void bpr_from_dots_kernel(const float* __restrict__ dots, int batch, int variant,
                                                            float* __restrict__ loss_i, float* __restrict__ coef) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float x = dots[b] - dots[batch + b];
    float l, c;
    if (variant == MMREC_BPR_LOGSIG) {
        l = neg_logsigmoid(x);
        c = -1.0f / (1.0f + expf(x));
    } else {
        const float s = 1.0f / (1.0f + expf(-x));
        l = -logf(1e-10f + s);
        c = -(s * (1.0f - s)) / (1e-10f + s);
    }
    loss_i[b] = l;
    coef[b] = c;
}

// out[0] = scale * sum(v[0..n)) ; single block, fixed-order (strided partial sums + LDS tree).
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ v, int n,
                                                         float scale, float* __restrict__ out) {
    __shared__ float red[256];
    float t = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) t += v[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = scale * red[0];
}

__device__ __forceinline__ void atomic_add_f4(float* base, float4 v) {
    unsafeAtomicAdd(base + 0, v.x);
    unsafeAtomicAdd(base + 1, v.y);
    unsafeAtomicAdd(base + 2, v.z);
    unsafeAtomicAdd(base + 3, v.w);
}

__global__ __launch_bounds__(256) void bpr_bwd_kernel(
    const float* __restrict__ U, const float* __restrict__ P, const float* __restrict__ N,
    const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
    const int64_t* __restrict__ neg, int batch, int d4, const float* __restrict__ coef,
    const float* __restrict__ grad_scalar, float scale, float* __restrict__ dU,
    float* __restrict__ dP, float* __restrict__ dN) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    const size_t ru = (size_t)users[b] * d4, rp = (size_t)pos[b] * d4, rn = (size_t)neg[b] * d4;
    const float c = grad_scalar[0] * scale * coef[b];
    for (int k = lane16; k < d4; k += 16) {
        const size_t iu = ru + k, ip = rp + k, in = rn + k;
        if (dU) {
            const float4 p = reinterpret_cast<const float4*>(P)[ip];
            const float4 n = reinterpret_cast<const float4*>(N)[in];
            atomic_add_f4(dU + iu * 4, make_float4(c * (p.x - n.x), c * (p.y - n.y), c * (p.z - n.z),
                                                   c * (p.w - n.w)));
        }
        const float4 u = reinterpret_cast<const float4*>(U)[iu];
        if (dP) atomic_add_f4(dP + ip * 4, f4_scale(c, u));
        if (dN) atomic_add_f4(dN + in * 4, f4_scale(-c, u));
    }
}

__global__ __launch_bounds__(256) void gather_sqnorm_kernel(const float* __restrict__ E,
                                                            const int64_t* __restrict__ ids,
                                                            int batch, int d4,
                                                            float* __restrict__ sq_i) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    float t = 0.f;
    for (int k = lane16; k < d4; k += 16) {
        const float4 e = reinterpret_cast<const float4*>(E)[(size_t)ids[b] * d4 + k];
        t += f4_dot(e, e);
    }
    const float s = row16_sum(t);
    if (lane16 == 0) sq_i[b] = s;
}

__global__ __launch_bounds__(256) void gather_scale_add_kernel(const float* __restrict__ E,
                                                               const int64_t* __restrict__ ids,
                                                               int batch, int d4,
                                                               const float* __restrict__ coef_scalar,
                                                               float* __restrict__ dE) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    for (int k = lane16; k < d4; k += 16) {
        const size_t i = (size_t)ids[b] * d4 + k;
        atomic_add_f4(dE + i * 4, f4_scale(coef_scalar[0], reinterpret_cast<const float4*>(E)[i]));
    }
}

// ---- several BPR terms over the SAME user rows in one launch pair (ABI 14): FREEDOM's id term and its two modality terms,
// total = sum_t w_t scale sum_b loss(<u, p_t> - <u, n_t>) (freedom.py:197-211: bpr + reg_weight (mf_t + mf_v)).  Per term the step
// ran bpr_fwd_kernel + reduce_sum_kernel, then three scalar launches to weight and add the losses, and on the way back two
// scalar launches and one bpr_bwd_kernel per term: 9 + 5 launches for what is 2 + 1 here (grid.y = term).  Per-sample arithmetic
// is bpr_fwd_kernel's / bpr_bwd_kernel's; losses[t] = scale sum_b loss_t as the per-term call leaves it.
struct BprTerms {
    const float* I[MMREC_BPR_MAX_TERMS];
    const int64_t* pos[MMREC_BPR_MAX_TERMS];
    const int64_t* neg[MMREC_BPR_MAX_TERMS];
    float* dI[MMREC_BPR_MAX_TERMS];
    float w[MMREC_BPR_MAX_TERMS];
    int n_terms;
};

__global__ __launch_bounds__(256) void bpr_multi_fwd_kernel(const float* __restrict__ U, const int64_t* __restrict__ users,
                                                            const BprTerms a, int batch, int d4, int variant,
                                                            float* __restrict__ part, float* __restrict__ coef) {
    __shared__ float s_row[16];
    const int t = blockIdx.y, lane16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int b = blockIdx.x * 16 + g;
    float l = 0.f;
    if (b < batch) {
        const size_t iu = (size_t)users[b] * d4, ip = (size_t)a.pos[t][b] * d4, in = (size_t)a.neg[t][b] * d4;
        float pp = 0.f, nn = 0.f;
        for (int c = lane16; c < d4; c += 16) {
            const float4 u = reinterpret_cast<const float4*>(U)[iu + c];
            pp += f4_dot(u, reinterpret_cast<const float4*>(a.I[t])[ip + c]);
            nn += f4_dot(u, reinterpret_cast<const float4*>(a.I[t])[in + c]);
        }
        const float x = row16_sum(pp) - row16_sum(nn);
        float c;
        if (variant == MMREC_BPR_LOGSIG) {
            l = neg_logsigmoid(x);
            c = -1.0f / (1.0f + expf(x));
        } else {
            const float sg = 1.0f / (1.0f + expf(-x));
            l = -logf(1e-10f + sg);
            c = -(sg * (1.0f - sg)) / (1e-10f + sg);
        }
        if (lane16 == 0) coef[(size_t)t * batch + b] = c;
    }
    if (lane16 == 0) s_row[g] = l;
    __syncthreads();
    if (threadIdx.x == 0) {                              // the block's 16 losses in sample order
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += s_row[i];
        part[(size_t)t * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void bpr_multi_finish_kernel(const BprTerms a, const float* __restrict__ part, int n_blocks,
                                                               float scale, float* __restrict__ total, float* __restrict__ losses) {
    __shared__ float red[MMREC_BPR_MAX_TERMS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[MMREC_BPR_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < MMREC_BPR_MAX_TERMS; ++t) {
        x[t] = 0.f;
        if (t < a.n_terms)
            for (int i = threadIdx.x; i < n_blocks; i += 256) x[t] += part[(size_t)t * n_blocks + i];
    }
#pragma unroll
    for (int t = 0; t < MMREC_BPR_MAX_TERMS; ++t) {
        if (t < a.n_terms) {
            const float w = wave_sum(x[t]);
            if (lane == 0) red[t][wave] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int t = 0; t < a.n_terms; ++t) {
            const float lt = scale * ((red[t][0] + red[t][1]) + (red[t][2] + red[t][3]));
            if (losses) losses[t] = lt;
            tot += a.w[t] * lt;
        }
        total[0] = tot;
    }
}

__global__ __launch_bounds__(256) void bpr_multi_bwd_kernel(const float* __restrict__ U, const int64_t* __restrict__ users,
                                                            const BprTerms a, int batch, int d4, const float* __restrict__ coef,
                                                            const float* __restrict__ grad_scalar, float scale,
                                                            float* __restrict__ dU) {
    const int t = blockIdx.y, lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    const size_t ru = (size_t)users[b] * d4, rp = (size_t)a.pos[t][b] * d4, rn = (size_t)a.neg[t][b] * d4;
    const float c = grad_scalar[0] * scale * a.w[t] * coef[(size_t)t * batch + b];
    float* dI = a.dI[t];
    for (int k = lane16; k < d4; k += 16) {
        const size_t iu = ru + k, ip = rp + k, in = rn + k;
        if (dU) {
            const float4 p = reinterpret_cast<const float4*>(a.I[t])[ip];
            const float4 n = reinterpret_cast<const float4*>(a.I[t])[in];
            atomic_add_f4(dU + iu * 4, make_float4(c * (p.x - n.x), c * (p.y - n.y), c * (p.z - n.z), c * (p.w - n.w)));
        }
        if (dI) {
            const float4 u = reinterpret_cast<const float4*>(U)[iu];
            atomic_add_f4(dI + ip * 4, f4_scale(c, u));
            atomic_add_f4(dI + in * 4, f4_scale(-c, u));
        }
    }
}

// ---- several mean-cosine terms in one launch pair (ABI 14): BM3's six BYOL terms (bm3.py:129-144) ---------------------------------
// out = sum_t w_t mean_b cos(X_t[ix_t[b]], Y_t[iy_t[b]]) with cosine_fwd_kernel's arithmetic per row; the per-term calls were
// 6 x (rows, sum) launches forward and 6 backward plus ~25 elementwise launches for the `1 - .` / weights / sums around them,
// a sixth of a BM3 step at Amazon-Clothing size.  Backward: dX_t[ix_t[b]] += g w_t / B_t (coef.x y - coef.y x), all terms in
// one launch; terms that share X share its gradient buffer.
struct CosTerms {
    const float* X[MMREC_COSINE_MAX_TERMS];
    const float* Y[MMREC_COSINE_MAX_TERMS];
    const int64_t* ix[MMREC_COSINE_MAX_TERMS];
    const int64_t* iy[MMREC_COSINE_MAX_TERMS];
    float* dX[MMREC_COSINE_MAX_TERMS];
    float w[MMREC_COSINE_MAX_TERMS];
    int batch[MMREC_COSINE_MAX_TERMS];
    int n_terms, max_batch;
};

__global__ __launch_bounds__(256) void cosine_multi_fwd_kernel(const CosTerms a, int d4, float* __restrict__ cos_i,
                                                               float2* __restrict__ coef) {
    __shared__ float s_row[16];
    const int t = blockIdx.y, lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if ((int)blockIdx.x * 16 >= a.batch[t]) return;     // (uniform)
    float xy = 0.f, xx = 0.f, yy = 0.f;
    if (b < a.batch[t]) {
        const size_t rx = (size_t)(a.ix[t] ? a.ix[t][b] : b) * d4, ry = (size_t)(a.iy[t] ? a.iy[t][b] : b) * d4;
        for (int c = lane16; c < d4; c += 16) {
            const float4 x = reinterpret_cast<const float4*>(a.X[t])[rx + c];
            const float4 y = reinterpret_cast<const float4*>(a.Y[t])[ry + c];
            xy += f4_dot(x, y);
            xx += f4_dot(x, x);
            yy += f4_dot(y, y);
        }
    }
    xy = row16_sum(xy);
    xx = row16_sum(xx);
    yy = row16_sum(yy);
    if (lane16 == 0 && b < a.batch[t]) {
        const float nx = sqrtf(xx), ny = sqrtf(yy);
        const float cx = fmaxf(nx, 1e-8f), cy = fmaxf(ny, 1e-8f);
        const float inv = 1.0f / (cx * cy), cs = xy * inv;
        coef[(size_t)t * a.max_batch + b] = make_float2(inv, nx > 1e-8f ? cs / (cx * cx) : 0.f);
        s_row[threadIdx.x >> 4] = cs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                              // the block's 16 cosines in row order: one value per workgroup
        float tot = 0.f;
        const int live = min(16, a.batch[t] - (int)blockIdx.x * 16);
        for (int i = 0; i < live; ++i) tot += s_row[i];
        cos_i[(size_t)t * ((a.max_batch + 15) / 16) + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void cosine_multi_finish_kernel(const CosTerms a, const float* __restrict__ cos_i,
                                                                  float* __restrict__ out) {
    __shared__ float red[MMREC_COSINE_MAX_TERMS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[MMREC_COSINE_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < MMREC_COSINE_MAX_TERMS; ++t) {
        x[t] = 0.f;
        if (t < a.n_terms)
            for (int i = threadIdx.x; i < (a.batch[t] + 15) / 16; i += 256) x[t] += cos_i[(size_t)t * ((a.max_batch + 15) / 16) + i];
    }
#pragma unroll
    for (int t = 0; t < MMREC_COSINE_MAX_TERMS; ++t) {
        if (t < a.n_terms) {
            const float w = wave_sum(x[t]);
            if (lane == 0) red[t][wave] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int t = 0; t < a.n_terms; ++t)
            total += a.w[t] * (((red[t][0] + red[t][1]) + (red[t][2] + red[t][3])) / (float)max(a.batch[t], 1));
        out[0] = total;
    }
}

__global__ __launch_bounds__(256) void cosine_multi_bwd_kernel(const CosTerms a, int d4, const float2* __restrict__ coef,
                                                               const float* __restrict__ grad_scalar) {
    const int t = blockIdx.y, lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= a.batch[t] || !a.dX[t]) return;
    const size_t rx = (size_t)(a.ix[t] ? a.ix[t][b] : b) * d4, ry = (size_t)(a.iy[t] ? a.iy[t][b] : b) * d4;
    const float g = grad_scalar[0] * a.w[t] / (float)a.batch[t];
    const float2 cf = coef[(size_t)t * a.max_batch + b];
    const float ca = g * cf.x, cc = g * cf.y;
    for (int k = lane16; k < d4; k += 16) {
        const float4 x = reinterpret_cast<const float4*>(a.X[t])[rx + k];
        const float4 y = reinterpret_cast<const float4*>(a.Y[t])[ry + k];
        atomic_add_f4(a.dX[t] + (rx + k) * 4, make_float4(ca * y.x - cc * x.x, ca * y.y - cc * x.y, ca * y.z - cc * x.z,
                                                          ca * y.w - cc * x.w));
    }
}

// ---- the regulariser of a training step in ONE forward and ONE backward launch pair (ABI 14) --------------------------------
// reg = scale * sum_t f(S_t),  S_t = sum_b ||E_t[ids_t[b]]||^2,  f = identity (mode 0: the L2 regulariser on batch rows,
// layergcn.py:154-161, lattice.py:214-216) or sqrt (mode 1: EmbLoss, common/loss.py:46-51 as used at vbpr.py:95,
// lightgcn.py:145-149, bpr.py).  A step used to spend ~11 launches forward (per term: row norms, their sum, a sqrt, the adds
// and scalings around them) and ~13 backward for its three terms -- a quarter of a LayerGCN / VBPR step at Amazon-Baby size,
// where every launch is ~4.5 us of a 0.35 ms step.  Here: row norms of ALL terms (grid.y = term), then one workgroup sums each
// term in fixed order (reduce_sum_kernel's), forms reg and leaves coef[t] = d reg / d row scale (2 scale, or scale / sqrt(S_t),
// 0 where S_t = 0 as torch.norm's backward does); backward: dE_t[ids_t[b]] += g coef[t] E_t[ids_t[b]] for all terms at once.
struct RegTerms {
    const float* E[MMREC_ROWS_REG_MAX_TERMS];
    const int64_t* ids[MMREC_ROWS_REG_MAX_TERMS];
    float* dE[MMREC_ROWS_REG_MAX_TERMS];
    int batch[MMREC_ROWS_REG_MAX_TERMS];
    int exclusive[MMREC_ROWS_REG_MAX_TERMS];   // no other term writes this term's dE
    int n_terms, max_batch, max_blocks;
};

__global__ __launch_bounds__(256) void rows_reg_sq_kernel(const RegTerms a, int d4, float* __restrict__ sq) {
    // sq[t][block] = the squared norms of the block's 16 rows, summed in row order (one value per workgroup: a whole table as a
    // term -- 39,387 rows at Amazon-Clothing size -- left the one-workgroup finish 57 us of sums)
    __shared__ float s_row[16];
    const int t = blockIdx.y, lane16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int b = blockIdx.x * 16 + g;
    if (blockIdx.x * 16 >= a.batch[t]) return;          // (uniform)
    float x = 0.f;
    if (b < a.batch[t]) {
        const float4* row = reinterpret_cast<const float4*>(a.E[t]) + (size_t)(a.ids[t] ? a.ids[t][b] : b) * d4;     // ids NULL: every row
        for (int k = lane16; k < d4; k += 16) {
            const float4 e = row[k];
            x += f4_dot(e, e);
        }
    }
    const float s = row16_sum(x);
    if (lane16 == 0) s_row[g] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += s_row[i];
        sq[(size_t)t * a.max_blocks + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void rows_reg_finish_kernel(const RegTerms a, const float* __restrict__ sq, int mode, float scale,
                                                              float* __restrict__ out, float* __restrict__ coef) {
    // every term summed by the whole workgroup in a fixed order (strided partial sums, wave butterflies, the four waves in
    // order); the terms' reductions are independent, so their loads are all issued before the first barrier
    __shared__ float red[MMREC_ROWS_REG_MAX_TERMS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[MMREC_ROWS_REG_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < MMREC_ROWS_REG_MAX_TERMS; ++t) {
        x[t] = 0.f;
        if (t < a.n_terms)
            for (int i = threadIdx.x; i < (a.batch[t] + 15) / 16; i += 256) x[t] += sq[(size_t)t * a.max_blocks + i];
    }
#pragma unroll
    for (int t = 0; t < MMREC_ROWS_REG_MAX_TERMS; ++t) {
        if (t < a.n_terms) {
            const float w = wave_sum(x[t]);
            if (lane == 0) red[t][wave] = w;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int t = 0; t < a.n_terms; ++t) {
            const float S = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
            if (mode == 0) {
                total += S;
                coef[t] = 2.f * scale;
            } else {
                const float nrm = sqrtf(S);
                total += nrm;
                coef[t] = S > 0.f ? scale / nrm : 0.f;
            }
        }
        out[0] = scale * total;
    }
}

__global__ __launch_bounds__(256) void rows_reg_bwd_kernel(const RegTerms a, int d4, const float* __restrict__ coef,
                                                           const float* __restrict__ g) {
    const int t = blockIdx.y, lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= a.batch[t]) return;
    const float c = g[0] * coef[t];
    const bool store = a.ids[t] == nullptr && a.exclusive[t];      // a whole table named by this term only: every element once
    for (int k = lane16; k < d4; k += 16) {
        const size_t i = (size_t)(a.ids[t] ? a.ids[t][b] : b) * d4 + k;
        const float4 v = f4_scale(c, reinterpret_cast<const float4*>(a.E[t])[i]);
        if (store) reinterpret_cast<float4*>(a.dE[t])[i] = v;
        else atomic_add_f4(a.dE[t] + i * 4, v);
    }
}

// mean_b cos(X[ix[b]], Y[iy[b]]) with F.cosine_similarity's clamp (each norm at least 1e-8) -- BM3's six BYOL terms
// (bm3.py:129-144; the targets Y are detached there: gradient w.r.t. X only).  ix / iy == nullptr: row b itself.
// coef[b] = {1 / (nx ny), cos / nx^2 (0 where the clamp is active)}: dcos/dx = coef.x * y - coef.y * x.
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ ix,
                                                         const float* __restrict__ Y, const int64_t* __restrict__ iy,
                                                         int batch, int d4, float* __restrict__ cos_i,
                                                         float2* __restrict__ coef) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    const size_t rx = (size_t)(ix ? ix[b] : b) * d4, ry = (size_t)(iy ? iy[b] : b) * d4;
    float xy = 0.f, xx = 0.f, yy = 0.f;
    for (int c = lane16; c < d4; c += 16) {
        const float4 x = reinterpret_cast<const float4*>(X)[rx + c];
        const float4 y = reinterpret_cast<const float4*>(Y)[ry + c];
        xy += f4_dot(x, y);
        xx += f4_dot(x, x);
        yy += f4_dot(y, y);
    }
    xy = row16_sum(xy);
    xx = row16_sum(xx);
    yy = row16_sum(yy);
    if (lane16 == 0) {
        const float nx = sqrtf(xx), ny = sqrtf(yy);
        const float cx = fmaxf(nx, 1e-8f), cy = fmaxf(ny, 1e-8f);
        const float inv = 1.0f / (cx * cy), cs = xy * inv;
        cos_i[b] = cs;
        coef[b] = make_float2(inv, nx > 1e-8f ? cs / (cx * cx) : 0.f);
    }
}

__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ ix,
                                                         const float* __restrict__ Y, const int64_t* __restrict__ iy,
                                                         int batch, int d4, const float2* __restrict__ coef,
                                                         const float* __restrict__ grad_scalar, float scale,
                                                         float* __restrict__ dX) {
    const int lane16 = threadIdx.x & 15;
    const int b = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (b >= batch) return;
    const size_t rx = (size_t)(ix ? ix[b] : b) * d4, ry = (size_t)(iy ? iy[b] : b) * d4;
    const float g = grad_scalar[0] * scale;
    const float a = g * coef[b].x, c = g * coef[b].y;
    for (int k = lane16; k < d4; k += 16) {
        const float4 x = reinterpret_cast<const float4*>(X)[rx + k];
        const float4 y = reinterpret_cast<const float4*>(Y)[ry + k];
        atomic_add_f4(dX + (rx + k) * 4, make_float4(a * y.x - c * x.x, a * y.y - c * x.y, a * y.z - c * x.z,
                                                      a * y.w - c * x.w));
    }
}

}  // namespace

extern "C" size_t mmrec_cosine_workspace_bytes(int32_t batch) {
    return batch > 0 ? (size_t)batch * sizeof(float) : 0;
}

extern "C" int mmrec_cosine_fwd_f32(const float* X, const int64_t* ix, const float* Y, const int64_t* iy,
                                    int32_t batch, int32_t d, float scale, float* out, float* coef, void* workspace,
                                    mmrec_stream_t stream) {
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats
    if (!out || (batch > 0 && (!X || !Y || !coef || !workspace))) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    if (batch > 0)
        hipLaunchKernelGGL(cosine_fwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, s, X, ix, Y, iy, batch, d / 4,
                           static_cast<float*>(workspace), reinterpret_cast<float2*>(coef));
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), batch, scale,
                       out);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_cosine_bwd_f32(const float* X, const int64_t* ix, const float* Y, const int64_t* iy,
                                    int32_t batch, int32_t d, const float* coef, const float* grad_scalar, float scale,
                                    float* dX, mmrec_stream_t stream) {
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (batch == 0) return 0;
    if (!X || !Y || !coef || !grad_scalar || !dX) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, mmrec_stream(stream), X, ix, Y, iy,
                       batch, d / 4, reinterpret_cast<const float2*>(coef), grad_scalar, scale, dX);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" size_t mmrec_bpr_workspace_bytes(int32_t batch) {
    return batch > 0 ? (size_t)batch * sizeof(float) : 0;
}

extern "C" int mmrec_bpr_fwd_f32(const float* U, const float* P, const float* N,
                                 const int64_t* users, const int64_t* pos, const int64_t* neg,
                                 int32_t batch, int32_t d, int32_t variant, float scale,
                                 float* loss_out, float* coef, void* workspace,
                                 mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats
    if (variant != MMREC_BPR_LOGSIG && variant != MMREC_BPR_GAMMA) return MMREC_ERR_BAD_ARG;
    if (batch < 0 || !loss_out) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    if (batch > 0) {
        if (!U || !P || !N || !users || !pos || !neg || !coef || !workspace) return MMREC_ERR_BAD_ARG;
        hipLaunchKernelGGL(bpr_fwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, s, U, P, N, users,
                           pos, neg, batch, d / 4, variant, static_cast<float*>(workspace), coef);
    }
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s,
                       static_cast<const float*>(workspace), batch, scale, loss_out);
    MMREC_RETURN_LAUNCH_STATUS();
}

// rows of a feature slice (8 / 16 / 32 floats) or of whole tables (64 k floats)
static bool bpr_slice_width(int d) { return d == 8 || d == 16 || d == 32 || (d > 0 && d % MMREC_EMB_DIM == 0); }

extern "C" int mmrec_bpr_dots_f32(const float* U, const float* P, const float* N, const int64_t* users, const int64_t* pos,
                                  const int64_t* neg, int32_t batch, int32_t d, float* dots, mmrec_stream_t stream) {
    if (!bpr_slice_width(d)) return MMREC_ERR_UNSUPPORTED;
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    if (batch == 0) return 0;
    if (!U || !P || !N || !users || !pos || !neg || !dots) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bpr_dots_kernel, dim3((batch + 15) / 16), dim3(256), 0, mmrec_stream(stream), U, P, N, users, pos, neg,
                       batch, d / 4, dots);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_bpr_loss_from_dots_f32(const float* dots, int32_t batch, int32_t variant, float scale, float* loss_out,
                                            float* coef, void* workspace, mmrec_stream_t stream) {
    if (variant != MMREC_BPR_LOGSIG && variant != MMREC_BPR_GAMMA) return MMREC_ERR_BAD_ARG;
    if (batch < 0 || !loss_out) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    if (batch > 0) {
        if (!dots || !coef || !workspace) return MMREC_ERR_BAD_ARG;
        hipLaunchKernelGGL(bpr_from_dots_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, dots, batch, variant,
                           static_cast<float*>(workspace), coef);
    }
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, static_cast<const float*>(workspace), batch, scale,
                       loss_out);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_bpr_bwd_f32(const float* U, const float* P, const float* N,
                                 const int64_t* users, const int64_t* pos, const int64_t* neg,
                                 int32_t batch, int32_t d, const float* coef,
                                 const float* grad_scalar, float scale, float* dU, float* dP,
                                 float* dN, mmrec_stream_t stream) {
    if (!bpr_slice_width(d)) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats, or a slice of 8 / 16 / 32
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    if (batch == 0) return 0;
    if (!U || !P || !N || !users || !pos || !neg || !coef || !grad_scalar) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bpr_bwd_kernel, dim3((batch + 15) / 16), dim3(256), 0, mmrec_stream(stream), U,
                       P, N, users, pos, neg, batch, d / 4, coef, grad_scalar, scale, dU, dP, dN);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_gather_sqnorm_fwd_f32(const float* E, const int64_t* ids, int32_t batch,
                                           int32_t d, float* out, void* workspace,
                                           mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats
    if (batch < 0 || !out) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    if (batch > 0) {
        if (!E || !ids || !workspace) return MMREC_ERR_BAD_ARG;
        hipLaunchKernelGGL(gather_sqnorm_kernel, dim3((batch + 15) / 16), dim3(256), 0, s, E, ids,
                           batch, d / 4, static_cast<float*>(workspace));
    }
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s,
                       static_cast<const float*>(workspace), batch, 1.0f, out);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_gather_scale_add_bwd_f32(const float* E, const int64_t* ids, int32_t batch,
                                              int32_t d, const float* coef_scalar, float* dE,
                                              mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    if (batch == 0) return 0;
    if (!E || !ids || !coef_scalar || !dE) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(gather_scale_add_kernel, dim3((batch + 15) / 16), dim3(256), 0,
                       mmrec_stream(stream), E, ids, batch, d / 4, coef_scalar, dE);
    MMREC_RETURN_LAUNCH_STATUS();
}

namespace {
int bpr_terms(BprTerms& a, const float* const* I, const int64_t* const* pos, const int64_t* const* neg, const float* w,
              float* const* dI, int32_t n_terms, int32_t batch) {
    if (n_terms < 1 || n_terms > MMREC_BPR_MAX_TERMS || !I || !pos || !neg || !w) return MMREC_ERR_BAD_ARG;
    a.n_terms = n_terms;
    for (int t = 0; t < n_terms; ++t) {
        if (batch > 0 && (!I[t] || !pos[t] || !neg[t])) return MMREC_ERR_BAD_ARG;
        a.I[t] = I[t], a.pos[t] = pos[t], a.neg[t] = neg[t], a.w[t] = w[t], a.dI[t] = dI ? dI[t] : nullptr;
    }
    return 0;
}
}  // namespace

extern "C" size_t mmrec_bpr_multi_workspace_bytes(int32_t n_terms, int32_t batch) {
    return (size_t)(n_terms > 0 ? n_terms : 0) * (size_t)(((batch > 0 ? batch : 0) + 15) / 16) * sizeof(float) + 64;
}

extern "C" int mmrec_bpr_multi_fwd_f32(const float* U, const int64_t* users, const float* const* I, const int64_t* const* pos,
                                       const int64_t* const* neg, const float* w, int32_t n_terms, int32_t batch, int32_t d,
                                       int32_t variant, float scale, float* total, float* losses, float* coef, void* workspace,
                                       mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (variant != MMREC_BPR_LOGSIG && variant != MMREC_BPR_GAMMA) return MMREC_ERR_BAD_ARG;
    if (batch < 0 || !total) return MMREC_ERR_BAD_ARG;
    BprTerms a;
    if (int err = bpr_terms(a, I, pos, neg, w, nullptr, n_terms, batch)) return err;
    if (batch > 0 && (!U || !users || !coef || !workspace)) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    const int n_blocks = (batch + 15) / 16;
    if (batch > 0)
        hipLaunchKernelGGL(bpr_multi_fwd_kernel, dim3(n_blocks, n_terms), dim3(256), 0, s, U, users, a, batch, d / 4, variant,
                           static_cast<float*>(workspace), coef);
    hipLaunchKernelGGL(bpr_multi_finish_kernel, dim3(1), dim3(256), 0, s, a, static_cast<const float*>(workspace), n_blocks, scale,
                       total, losses);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_bpr_multi_bwd_f32(const float* U, const int64_t* users, const float* const* I, const int64_t* const* pos,
                                       const int64_t* const* neg, const float* w, int32_t n_terms, int32_t batch, int32_t d,
                                       const float* coef, const float* grad_scalar, float scale, float* dU, float* const* dI,
                                       mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    if (batch < 0) return MMREC_ERR_BAD_ARG;
    BprTerms a;
    if (int err = bpr_terms(a, I, pos, neg, w, dI, n_terms, batch)) return err;
    if (batch == 0) return 0;
    if (!U || !users || !coef || !grad_scalar) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(bpr_multi_bwd_kernel, dim3((batch + 15) / 16, n_terms), dim3(256), 0, mmrec_stream(stream), U, users, a, batch,
                       d / 4, coef, grad_scalar, scale, dU);
    MMREC_RETURN_LAUNCH_STATUS();
}

namespace {
int cos_terms(CosTerms& a, const float* const* X, const int64_t* const* ix, const float* const* Y, const int64_t* const* iy,
              const float* w, float* const* dX, const int32_t* batch, int32_t n_terms) {
    if (n_terms < 1 || n_terms > MMREC_COSINE_MAX_TERMS || !X || !Y || !ix || !iy || !w || !batch) return MMREC_ERR_BAD_ARG;
    a.n_terms = n_terms, a.max_batch = 0;
    for (int t = 0; t < n_terms; ++t) {
        if (batch[t] < 0 || (batch[t] > 0 && (!X[t] || !Y[t]))) return MMREC_ERR_BAD_ARG;
        a.X[t] = X[t], a.Y[t] = Y[t], a.ix[t] = ix[t], a.iy[t] = iy[t], a.w[t] = w[t], a.dX[t] = dX ? dX[t] : nullptr;
        a.batch[t] = batch[t];
        if (batch[t] > a.max_batch) a.max_batch = batch[t];
    }
    return 0;
}
}  // namespace

extern "C" size_t mmrec_cosine_multi_workspace_bytes(int32_t n_terms, int32_t max_batch) {
    return (size_t)(n_terms > 0 ? n_terms : 0) * (size_t)(max_batch > 0 ? max_batch : 0) * sizeof(float) + 16;
}

extern "C" int mmrec_cosine_multi_fwd_f32(const float* const* X, const int64_t* const* ix, const float* const* Y,
                                          const int64_t* const* iy, const float* w, const int32_t* batch, int32_t n_terms, int32_t d,
                                          float* out, float* coef, void* workspace, mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    CosTerms a;
    if (int err = cos_terms(a, X, ix, Y, iy, w, nullptr, batch, n_terms)) return err;
    if (!out || !coef || !workspace) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    float* cos_i = static_cast<float*>(workspace);
    if (a.max_batch > 0)
        hipLaunchKernelGGL(cosine_multi_fwd_kernel, dim3((a.max_batch + 15) / 16, n_terms), dim3(256), 0, s, a, d / 4, cos_i,
                           reinterpret_cast<float2*>(coef));
    hipLaunchKernelGGL(cosine_multi_finish_kernel, dim3(1), dim3(256), 0, s, a, (const float*)cos_i, out);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_cosine_multi_bwd_f32(const float* const* X, const int64_t* const* ix, const float* const* Y,
                                          const int64_t* const* iy, const float* w, const int32_t* batch, int32_t n_terms, int32_t d,
                                          const float* coef, const float* grad_scalar, float* const* dX, mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    CosTerms a;
    if (!dX) return MMREC_ERR_BAD_ARG;
    if (int err = cos_terms(a, X, ix, Y, iy, w, dX, batch, n_terms)) return err;
    if (!coef || !grad_scalar) return MMREC_ERR_BAD_ARG;
    if (a.max_batch == 0) return 0;
    hipLaunchKernelGGL(cosine_multi_bwd_kernel, dim3((a.max_batch + 15) / 16, n_terms), dim3(256), 0, mmrec_stream(stream), a, d / 4,
                       reinterpret_cast<const float2*>(coef), grad_scalar);
    MMREC_RETURN_LAUNCH_STATUS();
}

namespace {
int reg_terms(RegTerms& a, const float* const* E, const int64_t* const* ids, float* const* dE, const int32_t* batch,
              int32_t n_terms) {
    if (n_terms < 1 || n_terms > MMREC_ROWS_REG_MAX_TERMS || !E || !ids || !batch) return MMREC_ERR_BAD_ARG;
    a.n_terms = n_terms, a.max_batch = 0;
    for (int t = 0; t < n_terms; ++t) {
        if (batch[t] < 0 || (batch[t] > 0 && (!E[t] || (dE && !dE[t])))) return MMREC_ERR_BAD_ARG;      // ids[t] NULL: rows 0 .. batch[t] - 1
        a.E[t] = E[t], a.ids[t] = ids[t], a.dE[t] = dE ? dE[t] : nullptr, a.batch[t] = batch[t];
        if (batch[t] > a.max_batch) a.max_batch = batch[t];
    }
    a.max_blocks = (a.max_batch + 15) / 16;
    for (int t = 0; t < n_terms; ++t) {
        a.exclusive[t] = 1;
        for (int u = 0; u < n_terms; ++u)
            if (u != t && dE && dE[u] == dE[t]) a.exclusive[t] = 0;
    }
    return 0;
}
}  // namespace

extern "C" size_t mmrec_rows_reg_workspace_bytes(int32_t n_terms, int32_t max_batch) {
    return (size_t)(n_terms > 0 ? n_terms : 0) * (size_t)((max_batch > 0 ? max_batch : 0) + 15) / 16 * sizeof(float) + 64;
}

extern "C" int mmrec_rows_reg_fwd_f32(const float* const* E, const int64_t* const* ids, const int32_t* batch, int32_t n_terms,
                                      int32_t d, int32_t mode, float scale, float* out, float* coef, void* workspace,
                                      mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;  // rows of 64, 128, ... floats
    if (mode != 0 && mode != 1) return MMREC_ERR_BAD_ARG;
    RegTerms a;
    if (int err = reg_terms(a, E, ids, nullptr, batch, n_terms)) return err;
    if (!out || !coef || !workspace) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    float* sq = static_cast<float*>(workspace);
    if (a.max_batch > 0)
        hipLaunchKernelGGL(rows_reg_sq_kernel, dim3(a.max_blocks, n_terms), dim3(256), 0, s, a, d / 4, sq);
    hipLaunchKernelGGL(rows_reg_finish_kernel, dim3(1), dim3(256), 0, s, a, (const float*)sq, mode, scale, out, coef);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_rows_reg_bwd_f32(const float* const* E, const int64_t* const* ids, const int32_t* batch, int32_t n_terms,
                                      int32_t d, const float* coef, const float* g, float* const* dE, mmrec_stream_t stream) {
    if (d <= 0 || d % MMREC_EMB_DIM) return MMREC_ERR_UNSUPPORTED;
    RegTerms a;
    if (!dE) return MMREC_ERR_BAD_ARG;
    if (int err = reg_terms(a, E, ids, dE, batch, n_terms)) return err;
    if (!coef || !g) return MMREC_ERR_BAD_ARG;
    if (a.max_batch == 0) return 0;
    hipLaunchKernelGGL(rows_reg_bwd_kernel, dim3((a.max_batch + 15) / 16, n_terms), dim3(256), 0, mmrec_stream(stream), a, d / 4,
                       coef, g);
    MMREC_RETURN_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------------------------
// Deterministic scatter-add of per-sample rows (`hip_deterministic`): out[ids[b]] += rows[b] with the occurrences of an id
// summed in POSITION order by the one 16-lane group that owns the id's segment of the (stable) sorted order -- no float
// atomics, so duplicated ids (the same user / item several times in a batch) give the same bits run after run, like the
// reference's CPU scatter (its CPU path is deterministic).  `order` = stable argsort of ids (the caller sorts:
// plumbing); ids < 0 are skipped.  Rows of d = 64 k floats.
namespace {
__global__ __launch_bounds__(256) void scatter_rows_sorted_kernel(const int64_t* __restrict__ order,
                                                                  const int64_t* __restrict__ ids,
                                                                  const float* __restrict__ rows, int n, int d4,
                                                                  float* __restrict__ out) {
    const int s = blockIdx.x * 16 + (threadIdx.x >> 4), lane16 = threadIdx.x & 15;
    if (s >= n) return;
    const int64_t id = ids[order[s]];
    if (id < 0 || (s > 0 && ids[order[s - 1]] == id)) return;     // not the head of a segment
    for (int c = lane16; c < d4; c += 16) {
        float4 acc = f4_zero();
        for (int j = s; j < n && ids[order[j]] == id; ++j)
            acc = f4_add(acc, reinterpret_cast<const float4*>(rows)[(size_t)order[j] * d4 + c]);
        float4* dst = reinterpret_cast<float4*>(out) + (size_t)id * d4 + c;
        *dst = f4_add(*dst, acc);
    }
}
}  // namespace

extern "C" int mmrec_scatter_add_rows_sorted_f32(const int64_t* order, const int64_t* ids, const float* rows, int32_t n,
                                                 int32_t d, float* out, mmrec_stream_t stream) {
    if (!bpr_slice_width(d)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!order || !ids || !rows || !out) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(scatter_rows_sorted_kernel, dim3((n + 15) / 16), dim3(256), 0, mmrec_stream(stream), order, ids, rows,
                       n, d / 4, out);
    MMREC_RETURN_LAUNCH_STATUS();
}
