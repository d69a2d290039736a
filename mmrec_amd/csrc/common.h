// Shared device/host helpers for libmmrec_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmrec_hip.h"

#define MMREC_WAVE 64

// Every launch is followed by this: report launch-configuration errors without synchronising.
#define MMREC_RETURN_LAUNCH_STATUS() return (int)hipGetLastError()

static inline hipStream_t mmrec_stream(mmrec_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc) {
    acc.x = fmaf(a, x.x, acc.x);
    acc.y = fmaf(a, x.y, acc.y);
    acc.z = fmaf(a, x.z, acc.z);
    acc.w = fmaf(a, x.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_scale(float s, float4 a) {
    return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// x of lane (l ^ M), M a power of two, WITHOUT the LDS address path of __shfl_xor (which is a ds_bpermute_b32 plus the VALU
// instructions that form its byte address -- every wave-level sort, reduction and broadcast of this library was made of them:
// 196 in the warm bound kernel, 482 in the top-K final kernel): M = 1, 2 are DPP quad permutations (one VALU modifier), M = 4,
// 8, 16 the LDS crossbar's fixed-pattern swizzle (no address registers), M = 32 the generic form.  Same value as
// __shfl_xor(x, M, 64) for every lane.
template <int M>
__device__ __forceinline__ int lane_xor_i(int x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "a power of two below 64");
    if (M == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1, 0, 3, 2]
    if (M == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);        // quad_perm [2, 3, 0, 1]
    if (M == 32) return __shfl_xor(x, 32, 64);
    return __builtin_amdgcn_ds_swizzle(x, (M << 10) | 0x1f);                             // bit mode: lane ^ M within 32 lanes
}
template <int M>
__device__ __forceinline__ float lane_xor_f(float x) { return __int_as_float(lane_xor_i<M>(__float_as_int(x))); }
// the same with the distance as a (compile-time foldable) value: the unrolled loops of the sorting networks
__device__ __forceinline__ int lane_xor_i(int x, int m) {
    switch (m) {
        case 1: return lane_xor_i<1>(x);
        case 2: return lane_xor_i<2>(x);
        case 4: return lane_xor_i<4>(x);
        case 8: return lane_xor_i<8>(x);
        case 16: return lane_xor_i<16>(x);
        default: return lane_xor_i<32>(x);
    }
}
__device__ __forceinline__ float lane_xor_f(float x, int m) { return __int_as_float(lane_xor_i(__float_as_int(x), m)); }

// Sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane gets the total.  The butterfly's pairs and order
// are those of the __shfl_xor form this replaces: the same bits.
__device__ __forceinline__ float row16_sum(float v) {
    v += lane_xor_f<8>(v);
    v += lane_xor_f<4>(v);
    v += lane_xor_f<2>(v);
    v += lane_xor_f<1>(v);
    return v;
}
// Sum over all 64 lanes of the wave.
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor_f<32>(v);
    v += lane_xor_f<16>(v);
    return row16_sum(v);
}
