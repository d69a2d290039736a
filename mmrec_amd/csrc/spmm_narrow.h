// spmm_narrow.hip: CSR SpMM on one feature slice (rows of 8 / 16 / 32 floats); called by mmrec_spmm_csr_f32 (spmm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int spmm_narrow_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, float* Y,
                       const float* Z, const float* acc_in, float* acc_out, int n_rows, int d, float alpha, float beta,
                       float acc_scale, int long_t, const int32_t* long_rows, const int32_t* long_chunk_ptr, int n_long,
                       int n_chunks, float* partials, hipStream_t s);

// listed rows only (pull, in the full launch's order) and its transpose-free atomic backward (push); d = 8 / 16 / 32 / 64
int spmm_pull_rows_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, const float* Z,
                          int z_compact, const int64_t* rows, int n_list, int d, int long_t, float* Y, hipStream_t s,
                          int max_chunks = 1);
int spmm_push_rows_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* G, float g_scale,
                          const int64_t* rows, int n_list, int d, float* dX, float* dZ, hipStream_t s);
