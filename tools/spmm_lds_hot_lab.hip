// LAB (not in the library): "LDS-staged embedding tiles" for the CSR SpMM (north_star's wording), measured.
//
// Question: does keeping the H most popular X rows in LDS pay for the d = 64 gather SpMM on the config-5 graph?  The counters
// (profiles/r04_spmm_pmc.txt) priced it: the hot rows are ~10 % of the gathers and already L2 hits, every L2 MISS is a 128-B
// fabric request that LDS cannot remove -- but a price is not a measurement, so here is the A/B.
//
// Both variants are the SAME persistent kernel (1024-thread workgroups = 64 sixteen-lane row groups, grid = 2 per CU, each group
// walks rows r, r + 64 * grid, ...; the row kernel's gather: 16 (col, val) pairs per coalesced load, ds_bpermute broadcast,
// 8 X-row loads in flight); they differ only in where a popular column's row comes from:
//   H = 0   every X row from global memory (what the library's spmm_rows_kernel does)
//   H > 0   the H most popular columns are renumbered to -(slot + 1) in a preprocessed colidx; their rows are copied to LDS once
//           per workgroup (H x 256 B: 32 / 64 / 128 KB) and read from there.  A wave's four groups diverge on the sign.
// Rows longer than `long_t` are skipped by both (the library gives them to chunk blocks; they gather USER rows, which have no
// popular columns).  Same summation order in both: outputs are compared bit for bit by the driver (tools/spmm_lds_hot_lab.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe_libs/libspmm_lds_hot_lab.so tools/spmm_lds_hot_lab.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc) {
    acc.x = fmaf(a, x.x, acc.x);
    acc.y = fmaf(a, x.y, acc.y);
    acc.z = fmaf(a, x.z, acc.z);
    acc.w = fmaf(a, x.w, acc.w);
    return acc;
}

template <int H>
__global__ __launch_bounds__(1024) void lab_rows_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colenc,
                                                        const float* __restrict__ vals, const float* __restrict__ X,
                                                        const int32_t* __restrict__ hot_ids, float* __restrict__ Y, int n_rows,
                                                        int long_t) {
    extern __shared__ float4 hot[];               // [H][16]
    const int lane16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    if (H > 0) {
        for (int e = threadIdx.x; e < H * 16; e += 1024) hot[e] = X4[(size_t)hot_ids[e >> 4] * 16 + (e & 15)];
        __syncthreads();
    }
    for (int row = blockIdx.x * 64 + g; row < n_rows; row += gridDim.x * 64) {
        const int s = rowptr[row], e = rowptr[row + 1];
        if (e - s > long_t) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = s; base < e; base += 16) {
            const int k = base + lane16;
            int c = 0;
            float v = 0.f;
            if (k < e) {
                c = colenc[k];
                v = vals[k];
            }
            const int cnt = min(16, e - base);
            for (int j0 = 0; j0 < cnt; j0 += 8) {
                float4 x[8];
                float vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    const int cj = __shfl(c, j, 16);
                    vv[u] = __shfl(v, j, 16);
                    if (j >= cnt) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else if (H > 0 && cj < 0) x[u] = hot[(-cj - 1) * 16 + lane16];
                    else x[u] = X4[(size_t)cj * 16 + lane16];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = f4_fma(vv[u], x[u], acc);
            }
        }
        reinterpret_cast<float4*>(Y)[(size_t)row * 16 + lane16] = acc;
    }
}

template <int H>
int launch(const int32_t* rowptr, const int32_t* colenc, const float* vals, const float* X, const int32_t* hot_ids, float* Y,
           int n_rows, int long_t, int grid, hipStream_t s) {
    const size_t lds = (size_t)H * 256;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_rows_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(lab_rows_kernel<H>, dim3(grid), dim3(1024), lds, s, rowptr, colenc, vals, X, hot_ids, Y, n_rows, long_t);
    return (int)hipGetLastError();
}

}  // namespace

// H in {0, 128, 256, 512}; grid: persistent workgroups (2 per CU: 512; H = 512 fits one per CU: 256).
extern "C" int lab_spmm_rows(int H, const int32_t* rowptr, const int32_t* colenc, const float* vals, const float* X,
                             const int32_t* hot_ids, float* Y, int n_rows, int long_t, int grid, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (H) {
        case 0: return launch<0>(rowptr, colenc, vals, X, hot_ids, Y, n_rows, long_t, grid, s);
        case 128: return launch<128>(rowptr, colenc, vals, X, hot_ids, Y, n_rows, long_t, grid, s);
        case 256: return launch<256>(rowptr, colenc, vals, X, hot_ids, Y, n_rows, long_t, grid, s);
        case 512: return launch<512>(rowptr, colenc, vals, X, hot_ids, Y, n_rows, long_t, grid, s);
    }
    return -1;
}
