// spmm_narrow.hip: CSR SpMM on one feature slice (rows of 8 / 16 / 32 floats); called by mmrec_spmm_csr_f32 (spmm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// window-major lists of the short rows' nonzeros (see spmm_narrow_windows_kernel); null: the row-per-sub-group kernel
struct NarrowWindowLists {
    const int32_t* col;        // [n_groups * (64 / LPR)] column ids (padding: 0)
    const float* val;          //                          values     (padding: 0)
    const int32_t* row;        //                          row - wave's first row (padding: -1)
    const int32_t* wave_ptr;   // [n_waves + 1] group offsets
    int n_waves, rows_per_wave;
};

int spmm_narrow_launch(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* X, float* Y,
                       const float* Z, const float* acc_in, float* acc_out, int n_rows, int d, float alpha, float beta,
                       float acc_scale, int long_t, const int32_t* long_rows, const int32_t* long_chunk_ptr, int n_long,
                       int n_chunks, float* partials, const NarrowWindowLists* wl, hipStream_t s);
