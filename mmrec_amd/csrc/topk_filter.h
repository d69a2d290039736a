// Full-sort top-K for kd == 64 / 128 by a low-precision FILTER + exact refinement (topk_filter.hip).
#pragma once
#include "common.h"

// true when the filter path can serve this shape (mmrec_score_topk_f32 takes it unless the caller passes
// MMREC_TOPK_NO_FILTER in `flags`, which keeps the materialised path for A/B measurements)
bool topk64_filter_applicable(int nq, int nc, int kd, int k);
size_t topk64_filter_workspace_bytes(int nq, int nc, int kd, int k);
// candidate-side preparation (column statistics + centred fp16 copy of C), reusable across calls on the same C
size_t topk64_filter_prepared_bytes(int nc, int kd);
int topk64_filter_prepare(const float* C, int nc, int kd, void* prepared, hipStream_t s);
// A WARM call's extras (mmrec_score_topk_hinted_f32): per query a list of `hk` candidate ids expected to rank high (row
// rows[q] of `ids`, or row q when rows is null) from which the threshold is taken without pass 1; queue_counts (nullable):
// [2] device counters the call ADDS its slow-queue / overflow-queue lengths to.  ids == nullptr: the cold two-pass call.
struct FilterHint {
    const int32_t* ids = nullptr;
    int hk = 0;
    const int64_t* rows = nullptr;
    int* queue_counts = nullptr;
};
// same contract as mmrec_score_topk_f32 (kd == 64 or 128): enqueues on `s`, never synchronises; `prepared` may be null
int topk64_filter_launch(const float* Q, const float* C, int nq, int nc, int kd, const int32_t* mask_rowptr,
                         const int32_t* mask_col, int k, int64_t* out_idx, float* out_val, void* workspace,
                         const void* prepared, hipStream_t s, const FilterHint& hint = FilterHint());
