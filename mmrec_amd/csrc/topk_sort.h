// Candidate record, its order (score descending, ties by lower id), the 128-element wave bitonic sort and
// the packed LDS form -- shared by topk.hip and topk_filter.hip.
#pragma once
#include "common.h"
#include <limits.h>

namespace {

struct Cand {
    float v;
    int i;
};
__device__ __forceinline__ bool cand_before(Cand a, Cand b) {  // a ranks ahead of b
    return a.v > b.v || (a.v == b.v && a.i < b.i);
}
__device__ __forceinline__ Cand cand_shfl_xor(Cand c, int m) {      // (m: a power of two, a constant after unrolling)
    Cand o;
    o.v = lane_xor_f(c.v, m);
    o.i = lane_xor_i(c.i, m);
    return o;
}

// Sort 128 candidates (element e = lane -> x0, e = lane + 64 -> x1) into rank order.
__device__ __forceinline__ void bitonic128(Cand& x0, Cand& x1, int lane) {
#pragma unroll
    for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride == 64) {  // partner of e = lane is e + 64: in-lane exchange, ranked order
                if (cand_before(x1, x0)) { const Cand t = x0; x0 = x1; x1 = t; }
            } else {
                const bool lower = (lane & stride) == 0;
                const bool desc0 = (lane & size) == 0;                          // e = lane
                const bool desc1 = size == 128 ? true : (size == 64 ? false : desc0);  // e = lane+64
                const Cand o0 = cand_shfl_xor(x0, stride), o1 = cand_shfl_xor(x1, stride);
                const bool first0 = (lower == desc0), first1 = (lower == desc1);
                if (cand_before(x0, o0) != first0) x0 = o0;
                if (cand_before(x1, o1) != first1) x1 = o1;
            }
        }
    }
}

// Sort 64 candidates (element e = lane) into rank order: 21 exchange steps of ONE candidate per lane, against 28 steps of two
// for bitonic128 -- a third of the instructions for the lists that fit (a wave's ~1,000-instruction sort of <= 64 survivors was
// most of the Amazon-Baby evaluation's final kernel: 19,445 waves x 4 cycles per instruction).
__device__ __forceinline__ void bitonic64(Cand& x, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool lower = (lane & stride) == 0;
            const bool desc = (lane & size) == 0;          // size == 64: every lane -> rank order
            const Cand o = cand_shfl_xor(x, stride);
            if (cand_before(x, o) != (lower == desc)) x = o;
        }
    }
}

// largest float strictly below x (x finite or -inf): `s > below(x)`  <=>  `s >= x`
__device__ __forceinline__ float float_below(float x) {
    if (x == -INFINITY) return x;
    if (x == 0.f) return -1.4e-45f;
    const int b = __float_as_int(x);
    return __int_as_float(x > 0.f ? b - 1 : b + 1);
}
__device__ __forceinline__ unsigned long long pack_cand(float v, int idx) {
    return ((unsigned long long)(unsigned)idx << 32) | (unsigned)__float_as_int(v);
}
__device__ __forceinline__ Cand unpack_cand(unsigned long long e) {
    Cand c;
    c.v = __int_as_float((int)(unsigned)e);
    c.i = (int)(e >> 32);
    return c;
}

}  // namespace
