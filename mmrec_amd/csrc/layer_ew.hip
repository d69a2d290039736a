// Elementwise tails of an MMGCN layer in one launch each way (a5''; mmgcn.py:170-173 and its two repeats):
//     h = leaky_relu(conv(x));  x_hat = leaky_relu(linear(x)) + id_embedding;  cat((h, x_hat), dim=1)
// is four launches forward (two activations, an add, the cat) and four backward (the cat's two slices, two activation
// gradients) on [26,495 x 64 ... 384] tensors -- 6-8 us each in a step that is ~300 such launches.  Here:
//     out[:, :wa] = leaky_relu(A),  out[:, wa:] = leaky_relu(B) + R          (one launch; R may be absent)
//     dA = dOut[:, :wa] * (A > 0 ? 1 : slope),  dB = dOut[:, wa:] * (B > 0 ? 1 : slope),  dR = dOut[:, wa:]   (one launch)
// the same values as the torch ops (leaky_relu's gradient at 0 is `slope`, as at::leaky_relu_backward has it).
#include "common.h"

namespace {

__device__ __forceinline__ float4 leaky4(float4 a, float s) {
    return make_float4(a.x > 0.f ? a.x : a.x * s, a.y > 0.f ? a.y : a.y * s, a.z > 0.f ? a.z : a.z * s, a.w > 0.f ? a.w : a.w * s);
}
__device__ __forceinline__ float4 leaky4_grad(float4 a, float4 g, float s) {
    return make_float4(a.x > 0.f ? g.x : g.x * s, a.y > 0.f ? g.y : g.y * s, a.z > 0.f ? g.z : g.z * s, a.w > 0.f ? g.w : g.w * s);
}

// one thread per float4 of the output row-major [n, (wa + wb) / 4]
__global__ __launch_bounds__(256) void cat_leaky_fwd_kernel(const float4* __restrict__ A, const float4* __restrict__ B,
                                                            const float4* __restrict__ R, size_t n, int wa4, int wb4, float slope,
                                                            float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int w4 = wa4 + wb4;
    if (i >= n * w4) return;
    const size_t row = i / w4;
    const int c = (int)(i - row * w4);
    if (c < wa4) {
        out[i] = leaky4(A[row * wa4 + c], slope);
    } else {
        float4 y = leaky4(B[row * wb4 + (c - wa4)], slope);
        if (R) y = f4_add(y, R[row * wb4 + (c - wa4)]);
        out[i] = y;
    }
}

__global__ __launch_bounds__(256) void cat_leaky_bwd_kernel(const float4* __restrict__ A, const float4* __restrict__ B,
                                                            const float4* __restrict__ dOut, size_t n, int wa4, int wb4, float slope,
                                                            float4* __restrict__ dA, float4* __restrict__ dB, float4* __restrict__ dR) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int w4 = wa4 + wb4;
    if (i >= n * w4) return;
    const size_t row = i / w4;
    const int c = (int)(i - row * w4);
    const float4 g = dOut[i];
    if (c < wa4) {
        if (dA) dA[row * wa4 + c] = leaky4_grad(A[row * wa4 + c], g, slope);
    } else {
        const size_t j = row * wb4 + (c - wa4);
        if (dB) dB[j] = leaky4_grad(B[j], g, slope);
        if (dR) dR[j] = g;
    }
}

// F.normalize(x, p=2, dim=1) (lattice.py:165, mmgcn.py:167): y = x / max(||x||, 1e-12) -- four launches forward (norm, clamp,
// expand, div) and half a dozen backward in torch; here one each way.  A 16-lane group per row (d = 4 w4 floats, any w4).
// inv[row] = 1 / max(||x||, eps) is kept for the backward: dx = inv (g - y (y . g)) where the clamp is inactive, inv g where it is
// (the norm is then a constant), as autograd derives it from the four ops.
__global__ __launch_bounds__(256) void row_normalize_fwd_kernel(const float4* __restrict__ X, size_t n, int w4, float eps,
                                                                float4* __restrict__ Y, float* __restrict__ inv) {
    const int lane16 = threadIdx.x & 15;
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n) return;
    float t = 0.f;
    for (int k = lane16; k < w4; k += 16) {
        const float4 x = X[row * w4 + k];
        t += f4_dot(x, x);
    }
    const float nrm = sqrtf(row16_sum(t));
    const float r = 1.0f / fmaxf(nrm, eps);
    for (int k = lane16; k < w4; k += 16) Y[row * w4 + k] = f4_scale(r, X[row * w4 + k]);
    if (lane16 == 0) inv[row] = nrm > eps ? r : -r;      // sign bit: the clamp was active
}

__global__ __launch_bounds__(256) void row_normalize_bwd_kernel(const float4* __restrict__ Y, const float4* __restrict__ G,
                                                                const float* __restrict__ inv, size_t n, int w4,
                                                                float4* __restrict__ dX) {
    const int lane16 = threadIdx.x & 15;
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n) return;
    const float r = inv[row];
    float t = 0.f;
    if (r > 0.f)
        for (int k = lane16; k < w4; k += 16) t += f4_dot(Y[row * w4 + k], G[row * w4 + k]);
    const float yg = row16_sum(t);
    const float a = fabsf(r);
    for (int k = lane16; k < w4; k += 16) {
        const float4 y = Y[row * w4 + k], g = G[row * w4 + k];
        dX[row * w4 + k] = make_float4(a * (g.x - y.x * yg), a * (g.y - y.y * yg), a * (g.z - y.z * yg), a * (g.w - y.w * yg));
    }
}

}  // namespace

extern "C" int mmrec_row_normalize_fwd_f32(const float* X, int64_t n, int32_t d, float eps, float* Y, float* inv,
                                           mmrec_stream_t stream) {
    if (d <= 0 || (d & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!X || !Y || !inv) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(row_normalize_fwd_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, mmrec_stream(stream),
                       reinterpret_cast<const float4*>(X), (size_t)n, d / 4, eps, reinterpret_cast<float4*>(Y), inv);
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_row_normalize_bwd_f32(const float* Y, const float* G, const float* inv, int64_t n, int32_t d, float* dX,
                                           mmrec_stream_t stream) {
    if (d <= 0 || (d & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!Y || !G || !inv || !dX) return MMREC_ERR_BAD_ARG;
    hipLaunchKernelGGL(row_normalize_bwd_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, mmrec_stream(stream),
                       reinterpret_cast<const float4*>(Y), reinterpret_cast<const float4*>(G), inv, (size_t)n, d / 4,
                       reinterpret_cast<float4*>(dX));
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_cat_leaky_fwd_f32(const float* A, const float* B, const float* R, int64_t n, int32_t wa, int32_t wb, float slope,
                                       float* out, mmrec_stream_t stream) {
    if (wa <= 0 || wb <= 0 || (wa & 3) || (wb & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!A || !B || !out) return MMREC_ERR_BAD_ARG;
    const size_t total = (size_t)n * ((wa + wb) / 4);
    hipLaunchKernelGGL(cat_leaky_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mmrec_stream(stream),
                       reinterpret_cast<const float4*>(A), reinterpret_cast<const float4*>(B), reinterpret_cast<const float4*>(R), (size_t)n,
                       wa / 4, wb / 4, slope, reinterpret_cast<float4*>(out));
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_cat_leaky_bwd_f32(const float* A, const float* B, const float* dOut, int64_t n, int32_t wa, int32_t wb, float slope,
                                       float* dA, float* dB, float* dR, mmrec_stream_t stream) {
    if (wa <= 0 || wb <= 0 || (wa & 3) || (wb & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!A || !B || !dOut) return MMREC_ERR_BAD_ARG;
    const size_t total = (size_t)n * ((wa + wb) / 4);
    hipLaunchKernelGGL(cat_leaky_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mmrec_stream(stream),
                       reinterpret_cast<const float4*>(A), reinterpret_cast<const float4*>(B), reinterpret_cast<const float4*>(dOut), (size_t)n,
                       wa / 4, wb / 4, slope, reinterpret_cast<float4*>(dA), reinterpret_cast<float4*>(dB), reinterpret_cast<float4*>(dR));
    MMREC_RETURN_LAUNCH_STATUS();
}
