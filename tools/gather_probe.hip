// Row-gather ceiling of the chip: what a CSR SpMM's X-row fetches can reach at best.
// Every 16-lane group fetches whole 256-B rows (float4 per lane) T[idx[i]] with 8 loads in flight and sums them; one
// 256-B row per 64 indices is written back.  Tables from L2-sized to 1 GiB, indices uniform / zipf(0.8) / sequential.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_libs/gather_probe tools/gather_probe.hip && tools/probe_libs/gather_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void gather_rows(const v4f* __restrict__ T, const int* __restrict__ idx, long n_idx,
                                                   v4f* __restrict__ out) {
    const int lane16 = threadIdx.x & 15;
    const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long s = g * 64;
    if (s >= n_idx) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < 64; b += 16) {
        const int c = idx[s + b + lane16];
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += 8) {
            v4f x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cj = __shfl(c, j0 + u, 16);
                const v4f* p = T + (size_t)cj * 16 + lane16;
                x[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += x[u];
        }
    }
    out[g * 16 + lane16] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const long n_idx = 20'000'000 / 64 * 64;
    std::mt19937_64 rng(1);
    int* d_idx; v4f* d_out; v4f* d_T;
    const long max_rows = 4L << 20;
    CK(hipMalloc(&d_idx, n_idx * 4));
    CK(hipMalloc(&d_out, n_idx / 64 * 256));
    CK(hipMalloc(&d_T, max_rows * 256));
    CK(hipMemset(d_T, 0, max_rows * 256));
    std::vector<int> h(n_idx);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long sizes[] = {16384, 65536, 131072, 500000, 1000000, 2097152, 4194304};
    printf("%-10s %-12s %-6s %10s %10s\n", "rows", "table", "dist", "GB/s", "GB/s(nt)");
    for (long rows : sizes) {
        for (int dist = 0; dist < 3; ++dist) {
            if (dist == 1 && rows != 500000 && rows != 1000000) continue;   // host-side zipf sampling is slow
            if (dist == 2 && rows != 1000000 && rows != 4194304) continue;
            if (dist == 0) {
                std::uniform_int_distribution<long> u(0, rows - 1);
                for (long i = 0; i < n_idx; ++i) h[i] = (int)u(rng);
            } else if (dist == 1) {   // zipf(0.8) over a random permutation of the rows
                std::vector<double> cdf(rows);
                double s = 0; for (long r = 0; r < rows; ++r) { s += std::pow((double)(r + 1), -0.8); cdf[r] = s; }
                std::vector<int> perm(rows); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), rng);
                std::uniform_real_distribution<double> u(0.0, s);
                for (long i = 0; i < n_idx; ++i) h[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), u(rng)) - cdf.begin()];
            } else {
                for (long i = 0; i < n_idx; ++i) h[i] = (int)(i % rows);
            }
            CK(hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
            float ms[2];
            for (int nt = 0; nt < 2; ++nt) {
                const int blocks = (int)(n_idx / 64 / 16);
                for (int rep = 0; rep < 2; ++rep) {
                    if (nt) hipLaunchKernelGGL(gather_rows<1>, dim3(blocks), dim3(256), 0, 0, d_T, d_idx, n_idx, d_out);
                    else hipLaunchKernelGGL(gather_rows<0>, dim3(blocks), dim3(256), 0, 0, d_T, d_idx, n_idx, d_out);
                }
                CK(hipEventRecord(e0));
                for (int rep = 0; rep < 5; ++rep) {
                    if (nt) hipLaunchKernelGGL(gather_rows<1>, dim3(blocks), dim3(256), 0, 0, d_T, d_idx, n_idx, d_out);
                    else hipLaunchKernelGGL(gather_rows<0>, dim3(blocks), dim3(256), 0, 0, d_T, d_idx, n_idx, d_out);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[nt], e0, e1)); ms[nt] /= 5;
            }
            printf("%-10ld %-9.0f MB %-6s %10.0f %10.0f\n", rows, rows * 256 / 1e6, dist == 0 ? "unif" : dist == 1 ? "zipf" : "seq",
                   n_idx * 256.0 / ms[0] / 1e6, n_idx * 256.0 / ms[1] / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
