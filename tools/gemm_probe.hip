// Ablation probe for linear_fwd_kernel (gemm.hip), compile-time variants:
//   for m in 0 1 2 4 5 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMMREC_GEMM_PROBE_MODE=$m \
//       tools/gemm_probe.hip -o tools/gemm_probe_$m.bin; done
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../mmrec_amd/csrc/gemm.hip"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 7050, F = 4096;
    float *X, *W, *b, *Y, *ws;
    hipMalloc(&X, (size_t)n * F * 4); hipMalloc(&W, 64 * F * 4); hipMalloc(&b, 256); hipMalloc(&Y, n * 64 * 4);
    hipMalloc(&ws, mmrec_linear_workspace_bytes(n, F, 64));
    hipMemset(X, 0x3c, (size_t)n * F * 4); hipMemset(W, 0x3c, 64 * F * 4); hipMemset(b, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) mmrec_linear_fwd_f32(X, W, b, Y, n, F, 64, ws, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) mmrec_linear_fwd_f32(X, W, b, Y, n, F, 64, ws, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("n %d ablation mask %2d : %.1f us / call (incl. split-K reduce)  %.1f TF\n", n, MMREC_GEMM_PROBE_MODE, ms / reps * 1e3, 2.0 * n * F * 64 / (ms / reps * 1e-3) / 1e12);
    return 0;
}
