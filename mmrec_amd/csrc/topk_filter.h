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
// A call with per-query LISTS (mmrec_score_topk_hinted_f32): row rows[q] of `ids` (row q when rows is null), `hk` ids wide.
// !cold: the threshold is taken from the list instead of pass 1 (a warm call); update: the call leaves its ranking (top-k and the
// runners-up it ranked, -1 padded) in the row for the next one.  queue_counts (nullable): [2] device counters the call ADDS its
// slow-queue / overflow-queue lengths to.  ids == nullptr: a plain call.
struct FilterHint {
    int32_t* ids = nullptr;
    int hk = 0;
    const int64_t* rows = nullptr;
    int* queue_counts = nullptr;
    bool cold = false, update = true;
};
// same contract as mmrec_score_topk_f32 (kd == 64 or 128): enqueues on `s`, never synchronises; `prepared` may be null
int topk64_filter_launch(const float* Q, const float* C, int nq, int nc, int kd, const int32_t* mask_rowptr,
                         const int32_t* mask_col, int k, int64_t* out_idx, float* out_val, void* workspace,
                         const void* prepared, hipStream_t s, const FilterHint& hint = FilterHint());
