// Modal feature projection on the fp32 matrix cores (P3, SURVEY.md 8a: a8).
//   fwd : Y[n,64]  = X[n,F] W[64,F]^T + b          (nn.Linear, freedom.py:205,208)
//   bwdW: dW[64,F] = dY[n,64]^T X[n,F] ; db = colsum(dY)
//   bwdX: dX[n,F]  = dY[n,64] W[64,F]              (X is a trainable table, freedom.py:58,61)
//
// Roofline: fp32 MFMA (157.3 TFLOP/s).  2*n*F*64 FLOP against n*F*4 bytes of X gives 32 FLOP/B,
// above the 19.7 FLOP/B ridge, so X is streamed from HBM exactly once and the matrix pipe is the
// bound.  Instruction: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles / SIMD).  Lane l supplies
// A[i = l&31][kk = l>>5] and B[kk = l>>5][j = l&31]; D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// The contraction index is permuted so that lane half h owns 4 consecutive k (k8*8 + 4h .. +3): one
// 16-B LDS read feeds 4 MFMAs, and A/B use the same permutation, so the product is unchanged.
// All split partial sums are combined in a fixed order (no float atomics): deterministic.


// tools/gemm_probe.hip builds the forward kernel with an ablation bit mask (see linear_fwd_kernel);
// the library only ever uses 0.
#ifndef MMREC_GEMM_PROBE_MODE
#define MMREC_GEMM_PROBE_MODE 0
#endif
#define MMREC_STREAM_PROBE MMREC_GEMM_PROBE_MODE
#include "mfma_stream.h"
#ifndef MMREC_GEMM_LEGACY_FWD
#define MMREC_GEMM_LEGACY_FWD 0   // probe: force the register-staged forward for every F
#endif
#ifndef MMREC_FWD_ROT
#define MMREC_FWD_ROT 11         // split-operand forward over an X streamed from HBM: row block b starts its k walk at tile (b * ROT) % T of its chunk (0 = every block at tile 0)
#endif
#ifndef MMREC_GEMM_DYN_LDS
#define MMREC_GEMM_DYN_LDS 0   // probe: extra dynamic LDS per workgroup, caps workgroups per CU
#endif

namespace {

constexpr int LIN_BM = 128, LIN_BK = 64, LIN_LD = LIN_BK + 4;  // LD/4 odd -> conflict-free b128 reads
constexpr int LIN_Q4 = LIN_BK / 4;          // float4 per staged row
constexpr int LIN_RPP = 256 / LIN_Q4;       // rows staged per pass of the 256 threads
constexpr int LIN_XP = LIN_BM / LIN_RPP, LIN_WP = 64 / LIN_RPP;

// ---------------------------------------------------------------------------------------- forward
// grid (ceil(n/128), ksplit); wave w owns rows w*32..+31 and all 64 outputs (2 accumulators).
// ABL (tools/gemm_probe.hip only; the library uses 0): bit0 no streaming global loads, bit1 no LDS
// stores after the first tile, bit2 no barriers after the first tile, bit3 LDS fragments read once.
template <int ABL>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int n, int F,
                                                         int k_chunk) {
    __shared__ __attribute__((aligned(16))) float Xs[LIN_BM][LIN_LD];
    __shared__ __attribute__((aligned(16))) float Ws[64][LIN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * LIN_BM;
    const int kb = blockIdx.y * k_chunk, ke = min(kb + k_chunk, F);
    const int lr = tid / LIN_Q4, lq = (tid % LIN_Q4) * 4;  // staging: row within a pass, k offset
    f32x16 acc0 = {0}, acc1 = {0};
    float4 xr[LIN_XP], wr[LIN_WP];
    auto gload = [&](int k0) {
        const int k = k0 + lq;
#pragma unroll
        for (int p = 0; p < LIN_XP; ++p) {
            const int row = m0 + lr + LIN_RPP * p;
            xr[p] = ld4_guard(X + (size_t)row * F + k, row < n && k < ke);
        }
#pragma unroll
        for (int p = 0; p < LIN_WP; ++p) wr[p] = ld4_guard(W + (size_t)(lr + LIN_RPP * p) * F + k, k < ke);
    };
    gload(kb);
    const int i = lane & 31, h = lane >> 5;
    float4 fa[LIN_BK / 8], fb0[LIN_BK / 8], fb1[LIN_BK / 8];
    auto frags = [&]() {
#pragma unroll
        for (int k8 = 0; k8 < LIN_BK / 8; ++k8) {
            fa[k8] = *reinterpret_cast<const float4*>(&Xs[wave * 32 + i][k8 * 8 + 4 * h]);
            fb0[k8] = *reinterpret_cast<const float4*>(&Ws[i][k8 * 8 + 4 * h]);
            fb1[k8] = *reinterpret_cast<const float4*>(&Ws[32 + i][k8 * 8 + 4 * h]);
        }
    };
    for (int k0 = kb; k0 < ke; k0 += LIN_BK) {
        const bool first = k0 == kb;
        if (!(ABL & 2) || first) {
#pragma unroll
            for (int p = 0; p < LIN_XP; ++p) *reinterpret_cast<float4*>(&Xs[lr + LIN_RPP * p][lq]) = xr[p];
#pragma unroll
            for (int p = 0; p < LIN_WP; ++p) *reinterpret_cast<float4*>(&Ws[lr + LIN_RPP * p][lq]) = wr[p];
        }
        if (!(ABL & 4) || first) __syncthreads();
        if (k0 + LIN_BK < ke && !(ABL & 1)) gload(k0 + LIN_BK);  // next tile in flight under the MFMAs
        if (!(ABL & 8) || first) frags();
#pragma unroll
        for (int k8 = 0; k8 < LIN_BK / 8; ++k8) {
            const float4 a = fa[k8], b0 = fb0[k8], b1 = fb1[k8];
            acc0 = mfma32(a.x, b0.x, acc0); acc1 = mfma32(a.x, b1.x, acc1);
            acc0 = mfma32(a.y, b0.y, acc0); acc1 = mfma32(a.y, b1.y, acc1);
            acc0 = mfma32(a.z, b0.z, acc0); acc1 = mfma32(a.z, b1.z, acc1);
            acc0 = mfma32(a.w, b0.w, acc0); acc1 = mfma32(a.w, b1.w, acc1);
        }
        if (!(ABL & 4)) __syncthreads();
    }
    // out = Y (+bias) when gridDim.y == 1, else partial slab blockIdx.y of the workspace
    float* dst = out + (size_t)blockIdx.y * n * 64;
    const float b0 = (bias && gridDim.y == 1) ? bias[i] : 0.f;
    const float b1 = (bias && gridDim.y == 1) ? bias[32 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + d_row(r, lane);
        if (row < n) {
            dst[(size_t)row * 64 + i] = acc0[r] + b0;
            dst[(size_t)row * 64 + 32 + i] = acc1[r] + b1;
        }
    }
}

// ------------------------------------------------------------------- forward, LDS-DMA pipeline
// Same tile (128 x 64 per workgroup, wave w owns rows 32w..+31) but the operands go HBM -> LDS with
// `buffer_load_dwordx4 ... lds` (no staging VGPRs, no address VALU: scalar k offset + per-lane
// constant voffset through an SRSRC whose bounds zero-fill the rows past n), through a 3-stage ring
// of BK = 32 tiles with ONE barrier per tile and the loads of tile t+2 in flight under the MFMAs of
// tile t.  An LDS-DMA wave-instruction writes 1 KB linearly (8 rows x 128 B), so the bank swizzle is
// applied on the source side: 16-B chunk c of row r lives at position c ^ ((r >> 1) & 7), which makes
// every 16-lane service group of ds_read_b128 ({0-3,12-15,20-27}, ...) cover all 64 banks.
// Requires F % 32 == 0 (4096, 384, 4480 all are); other F take linear_fwd_kernel.
constexpr int DM_BK = 32, DM_STAGES = 3;
template <bool NT>
__global__ __launch_bounds__(256, 2) void linear_fwd_dma_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out, int n, int F,
                                                                int k_chunk, const int* __restrict__ redo) {
    __shared__ __attribute__((aligned(1024))) float Xs0[LIN_BM * DM_BK], Xs1[LIN_BM * DM_BK], Xs2[LIN_BM * DM_BK];
    __shared__ __attribute__((aligned(1024))) float Ws0[64 * DM_BK], Ws1[64 * DM_BK], Ws2[64 * DM_BK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fix-up launch of mmrec_linear_fwd_split_f32: only the 128-row blocks the split kernel flagged as outside its domain
    // (redo[block]) -- or every block when a row of W is (redo[gridDim.x + row]) -- are recomputed here, in exact fp32
    if (redo && !redo[blockIdx.x] && !__builtin_amdgcn_ballot_w64(redo[gridDim.x + lane] != 0)) return;
    const int m0 = blockIdx.x * LIN_BM;
    const int kb = blockIdx.y * k_chunk, ke = min(kb + k_chunk, F);
    const int T = (ke - kb) / DM_BK;
    const int rows_left = min(LIN_BM, n - m0);
    // probe bit 4: every workgroup streams the same 128 rows (L2 resident) instead of its own
    const i32x4 rx = raw_rsrc(X + ((MMREC_GEMM_PROBE_MODE & 16) ? 0 : (size_t)m0 * F), (unsigned)rows_left * (unsigned)F * 4u);
    const i32x4 rw = raw_rsrc(W, 64u * (unsigned)F * 4u);
    // per-lane source offsets of this wave's pieces (4 of X, 2 of W), swizzle on the source side
    int vx[4], vw[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        vx[j] = r * F * 4 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * wave + 8 * j + (lane >> 3);
        vw[j] = r * F * 4 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](float* xs, float* ws, int t) {
        const int so = (kb + t * DM_BK) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<NT>(rx, lds_addr(xs + (4 * wave + j) * 256), vx[j], so);
#pragma unroll
        for (int j = 0; j < 2; ++j)   // probe bit 5: W only for the first two tiles
            if (!(MMREC_GEMM_PROBE_MODE & 32) || t < 2) lds_dma16<false>(rw, lds_addr(ws + (2 * wave + j) * 256), vw[j], so);
    };
    const int i = lane & 31, h = lane >> 5, g = (i >> 1) & 7;
    int ko[4];  // float offset of this lane's 16-B chunk within a row, per k8
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) ko[k8] = ((2 * k8 + h) ^ g) << 2;
    f32x16 acc0 = {0}, acc1 = {0};
    auto compute = [&](const float* xs, const float* ws) {
        const float* xa = xs + (32 * wave + i) * DM_BK;
        const float* wb = ws + i * DM_BK;
        float4 fa[4], fb0[4], fb1[4];
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            fa[k8] = *reinterpret_cast<const float4*>(xa + ko[k8]);
            fb0[k8] = *reinterpret_cast<const float4*>(wb + ko[k8]);
            fb1[k8] = *reinterpret_cast<const float4*>(wb + 32 * DM_BK + ko[k8]);
        }
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const float4 a = fa[k8], b0 = fb0[k8], b1 = fb1[k8];
            acc0 = mfma32(a.x, b0.x, acc0); acc1 = mfma32(a.x, b1.x, acc1);
            acc0 = mfma32(a.y, b0.y, acc0); acc1 = mfma32(a.y, b1.y, acc1);
            acc0 = mfma32(a.z, b0.z, acc0); acc1 = mfma32(a.z, b1.z, acc1);
            acc0 = mfma32(a.w, b0.w, acc0); acc1 = mfma32(a.w, b1.w, acc1);
        }
    };
    // one pipeline step: tile t sits in (xc, wc); (xn, wn) is the stage tile t-1 used, now free
    auto step = [&](const float* xc, const float* wc, float* xn, float* wn, int t) {
        if (t + 1 < T && !(MMREC_GEMM_PROBE_MODE & 32)) MMREC_WAIT_VM(6); else MMREC_WAIT_VM(0);  // my pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                            // everyone's have; tile t-1 is consumed
        if (t + 2 < T && !((MMREC_GEMM_PROBE_MODE & 64) && t >= 2)) issue(xn, wn, t + 2);  // probe bit 6: no loads
        compute(xc, wc);
    };
    if (T > 0) issue(Xs0, Ws0, 0);
    if (T > 1) issue(Xs1, Ws1, 1);
    for (int t = 0; t < T;) {
        step(Xs0, Ws0, Xs2, Ws2, t); if (++t >= T) break;
        step(Xs1, Ws1, Xs0, Ws0, t); if (++t >= T) break;
        step(Xs2, Ws2, Xs1, Ws1, t); ++t;
    }
    float* dst = out + (size_t)blockIdx.y * n * 64;
    const float b0 = (bias && gridDim.y == 1) ? bias[i] : 0.f;
    const float b1 = (bias && gridDim.y == 1) ? bias[32 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + d_row(r, lane);
        if (row < n) {
            dst[(size_t)row * 64 + i] = acc0[r] + b0;
            dst[(size_t)row * 64 + 32 + i] = acc1[r] + b1;
        }
    }
}

// ------------------------------------------------- forward on the 16-bit matrix cores, fp32-accurate (split operands)
// fp32-input MFMA runs at 1/16 of the 16-bit rate, and at K = 4096 it -- not the 115 MB ... 8.2 GB stream of X -- bounds the
// projection (0.55 of ITS peak at Amazon-Baby).  Every fp32 number is hi + lo with hi = fp16(x) and lo = x - hi, |lo| <= 2^-11 |x|;
// with lo' = fp16(2^11 lo) both parts are normal fp16 numbers and
//     x w = hi_x hi_w + 2^-11 (hi_x lo'_w + lo'_x hi_w) + lo_x lo_w,        |lo_x lo_w| <= 2^-22 |x w|,
// three v_mfma_f32_32x32x16_f16 products (exact in the fp32 accumulators) instead of eight fp32 MFMAs per 16 k, two accumulator
// sets (hi hi / cross terms) combined once at the end.  Error per product <= 2^-21 |x w| (the dropped lo lo term + the rounding
// of lo' to 11 bits) -- the size of fp32's own accumulation error over K = 4096 terms; measured against float64 it is as
// accurate as the fp32 kernel (tests).
// DOMAIN, and what happens outside it (nothing silent): the split is exact to 2^-22 |x| only while fp16 holds both parts.
//   * |x| or |w| >= 65520 (or inf / NaN): hi is inf, the products are inf / NaN -> the OUTPUT is non-finite.
//   * 0 < |x| < 2^-14: hi is an fp16 subnormal and lo' cannot hold what it drops: absolute error 2^-36 per element, which is
//     harmless next to elements of ordinary size in the same row (error 2^-36 |w_k| against a sum of size |x_k'| |w_k'|) and NOT
//     harmless in a row whose every element is tiny (relative error 1e-4 at |x| ~ 1e-7, 1e-2 at 1e-8).
//   The kernel therefore tracks max |x| per row (one v_max3 per two elements) and looks at its own outputs; a 128-row block
//   with a non-finite output or a row with 0 < max |x| < 2^-10 sets redo[block], linear_w_split_kernel sets redo[last] for a
//   W row with 0 < max |w| < 2^-10, and a second launch -- the fp32 kernel, which returns at once for blocks that are not
//   flagged -- recomputes those blocks in exact fp32: inf / NaN propagate as F.linear's do, tiny rows get fp32's relative
//   accuracy.  No host synchronisation, capture-safe; the guarantee for unflagged rows is
//   |err| <= 2^-21 sum |x w| + 2^-25 max|x_row| sum |w_row|  (the second term: <= 2^-36 per tiny element, row max >= 2^-10).
// X streams through the same LDS-DMA ring as linear_fwd_dma_kernel (fp32 tiles, split in registers: ~6 VALU per element, about
// the time of the 12 MFMAs they feed); W is split ONCE per call into the workspace in the tile layout the DMA wants (per 32-k
// block and row: 32 hi halves | 32 lo' halves = the same 128 B as the fp32 tile row).  Bound: the HBM stream of X.
typedef __attribute__((ext_vector_type(8))) _Float16 g_half8;

constexpr float SPLIT_ROW_MIN = 0x1p-10f;    // a row (of X or W) whose largest |element| is below this, and not 0, is recomputed in fp32

// grid 64 (one workgroup per W row): a thread splits 8 consecutive floats at a time (two 16-B loads, coalesced across the
// workgroup) into 8 hi halves + 8 lo' halves of the row's tile layout; redo[0 .. nblocks) = 0 (workgroup 0),
// redo[nblocks + row] = 1 if W row `row` leaves the split's small-magnitude domain.
__global__ __launch_bounds__(256) void linear_w_split_kernel(const float* __restrict__ W, int F, float* __restrict__ Wsp,
                                                             int* __restrict__ redo, int nblocks) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    if (row == 0)
        for (int j = threadIdx.x; j < nblocks; j += 256) redo[j] = 0;
    float mx = 0.f;
    for (int e8 = threadIdx.x; e8 < F / 8; e8 += 256) {            // F % 32 == 0
        const int kb = e8 >> 2, sub = (e8 & 3) * 8;
        const float4* src = reinterpret_cast<const float4*>(W + (size_t)row * F + kb * 32 + sub);
        const float4 a = src[0], b = src[1];
        const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        g_half8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hi[k] = (_Float16)w[k];
            lo[k] = (_Float16)((w[k] - (float)hi[k]) * 2048.f);
            mx = fmaxf(mx, fabsf(w[k]));
        }
        _Float16* dst = reinterpret_cast<_Float16*>(Wsp + (size_t)row * F + kb * 32);
        *reinterpret_cast<g_half8*>(dst + sub) = hi;
        *reinterpret_cast<g_half8*>(dst + 32 + sub) = lo;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        redo[nblocks + row] = (mx > 0.f && mx < SPLIT_ROW_MIN) ? 1 : 0;
    }
}

template <bool NT>
__global__ __launch_bounds__(256, 2) void linear_fwd_dma_f16x3_kernel(const float* __restrict__ X,
                                                                      const float* __restrict__ Wsp,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ out, int n, int F,
                                                                      int k_chunk, float* __restrict__ rowmax_part,
                                                                      int* __restrict__ redo) {
    __shared__ __attribute__((aligned(1024))) float Xs0[LIN_BM * DM_BK], Xs1[LIN_BM * DM_BK], Xs2[LIN_BM * DM_BK];
    __shared__ __attribute__((aligned(1024))) float Ws0[64 * DM_BK], Ws1[64 * DM_BK], Ws2[64 * DM_BK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * LIN_BM;
    const int kb = blockIdx.y * k_chunk, ke = min(kb + k_chunk, F);
    const int T = (ke - kb) / DM_BK;
    const int rows_left = min(LIN_BM, n - m0);
    const i32x4 rx = raw_rsrc(X + (size_t)m0 * F, (unsigned)rows_left * (unsigned)F * 4u);
    const i32x4 rw = raw_rsrc(Wsp, 64u * (unsigned)F * 4u);
    int vx[4], vw[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        vx[j] = r * F * 4 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * wave + 8 * j + (lane >> 3);
        vw[j] = r * F * 4 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    // Row blocks walk their k chunk from DIFFERENT starting tiles (and wrap): with every workgroup at the same column window of
    // rows 16 KB apart the launch ran 10-13 % slower at one chunk per row block (Sports 92.7 -> 81 us, Clothing 90 -> 83 us
    // forward, profiles/r06_linear_fwd_rotation_and_splits.log); a row's summation order then depends on its block, which the
    // split count already made a function of n -- fp32-accurate either way (the tests' float64 comparisons).
    // Only for an X that is streamed from HBM (NT: larger than the Infinity Cache): against a cache-resident X the rotation
    // measured 1-2 us SLOWER (7,050 rows: 96.8 against 94.5 us forward + backward as a replay; 4,096 rows: +1.5 us).
#if MMREC_FWD_ROT
    const int rot = (NT && T > 0) ? (int)((blockIdx.x * (unsigned)MMREC_FWD_ROT) % (unsigned)T) : 0;
#else
    const int rot = 0;
#endif
    auto issue = [&](float* xs, float* ws, int t) {
        int tt = t + rot;
        if (tt >= T) tt -= T;
        const int so = (kb + tt * DM_BK) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<NT>(rx, lds_addr(xs + (4 * wave + j) * 256), vx[j], so);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16<false>(rw, lds_addr(ws + (2 * wave + j) * 256), vw[j], so);
    };
    const int i = lane & 31, h = lane >> 5, g = (i >> 1) & 7;
    f32x16 hh0 = {0}, hh1 = {0}, cx0 = {0}, cx1 = {0};      // hi x hi and cross-term accumulators of the two 32-output tiles
    float xmax = 0.f;                                       // max |x| over this lane's half of row 32 wave + i (domain check)
    auto compute = [&](const float* xs, const float* ws) {
        const float* xa = xs + (32 * wave + i) * DM_BK;
        const float* wb = ws + i * DM_BK;
#pragma unroll
        for (int st = 0; st < 2; ++st) {                     // MFMA step st: k = 16 st + 8 h ... + 7 of the 32-k tile
            const int c0 = 4 * st + 2 * h;
            const float4 xa0 = *reinterpret_cast<const float4*>(xa + ((c0 ^ g) << 2));
            const float4 xa1 = *reinterpret_cast<const float4*>(xa + (((c0 + 1) ^ g) << 2));
            const float xv[8] = {xa0.x, xa0.y, xa0.z, xa0.w, xa1.x, xa1.y, xa1.z, xa1.w};
            g_half8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 hi = (_Float16)xv[e];
                ah[e] = hi;
                al[e] = (_Float16)((xv[e] - (float)hi) * 2048.f);
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) xmax = fmaxf(fmaxf(xmax, fabsf(xv[e])), fabsf(xv[e + 1]));     // v_max3_f32 |.|, |.|
            const int ch = ((2 * st + h) ^ g) << 2, cl = ((4 + 2 * st + h) ^ g) << 2;     // hi / lo' chunks of the W tile row
            const g_half8 bh0 = *reinterpret_cast<const g_half8*>(wb + ch), bl0 = *reinterpret_cast<const g_half8*>(wb + cl);
            const g_half8 bh1 = *reinterpret_cast<const g_half8*>(wb + 32 * DM_BK + ch);
            const g_half8 bl1 = *reinterpret_cast<const g_half8*>(wb + 32 * DM_BK + cl);
            hh0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, hh0, 0, 0, 0);
            hh1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, hh1, 0, 0, 0);
            cx0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, cx0, 0, 0, 0);
            cx1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, cx1, 0, 0, 0);
            cx0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, cx0, 0, 0, 0);
            cx1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, cx1, 0, 0, 0);
        }
    };
    auto step = [&](const float* xc, const float* wc, float* xn, float* wn, int t) {
        if (t + 1 < T) MMREC_WAIT_VM(6); else MMREC_WAIT_VM(0);  // my pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                            // everyone's have; tile t-1 is consumed
        if (t + 2 < T) issue(xn, wn, t + 2);
        compute(xc, wc);
    };
    if (T > 0) issue(Xs0, Ws0, 0);
    if (T > 1) issue(Xs1, Ws1, 1);
    for (int t = 0; t < T;) {
        step(Xs0, Ws0, Xs2, Ws2, t); if (++t >= T) break;
        step(Xs1, Ws1, Xs0, Ws0, t); if (++t >= T) break;
        step(Xs2, Ws2, Xs1, Ws1, t); ++t;
    }
    float* dst = out + (size_t)blockIdx.y * n * 64;
    const float b0 = (bias && gridDim.y == 1) ? bias[i] : 0.f;
    const float b1 = (bias && gridDim.y == 1) ? bias[32 + i] : 0.f;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wave * 32 + d_row(r, lane);
        if (row < n) {
            const float y0 = fmaf(cx0[r], 1.f / 2048.f, hh0[r]) + b0, y1 = fmaf(cx1[r], 1.f / 2048.f, hh1[r]) + b1;
            dst[(size_t)row * 64 + i] = y0;
            dst[(size_t)row * 64 + 32 + i] = y1;
            bad |= !(fabsf(y0) < __builtin_inff()) | !(fabsf(y1) < __builtin_inff());      // inf or NaN
        }
    }
    // domain check (see the comment above the kernel): the two lane halves hold the two k halves of row 32 wave + i
    xmax = fmaxf(xmax, __shfl_xor(xmax, 32));
    if (gridDim.y == 1) {
        bad |= xmax > 0.f && xmax < SPLIT_ROW_MIN;
        if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) redo[blockIdx.x] = 1;       // every writer stores the same 1
    } else {                                            // the row's maximum over the slabs is taken by slab_reduce_kernel
        const int row = m0 + wave * 32 + i;
        if (h == 0 && row < n) rowmax_part[(size_t)blockIdx.y * n + row] = xmax;
        if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) redo[blockIdx.x] = 1;
    }
}

// out[idx] = sum_s part[s][idx] (+ bias[idx % 64]) in slab order; total = n*64 (multiple of 4).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, int nslab,
                                                          size_t slab_elems,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out,
                                                          const float* __restrict__ rowmax_part = nullptr,
                                                          int* __restrict__ redo = nullptr) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= slab_elems) return;
    part += (size_t)blockIdx.y * nslab * slab_elems;      // (grid.y > 1: the 64-output groups of a wide dW, each with its own slabs)
    out += (size_t)blockIdx.y * slab_elems;
    // four independent chains keep the slab loads in flight (a single chain of up to 64 dependent
    // adds was latency bound: 17 us for a few MB); the combination order is still fixed
    float4 t0 = f4_zero(), t1 = f4_zero(), t2 = f4_zero(), t3 = f4_zero();
    int s = 0;
    for (; s + 3 < nslab; s += 4) {
        t0 = f4_add(t0, reinterpret_cast<const float4*>(part + (size_t)(s + 0) * slab_elems)[i4]);
        t1 = f4_add(t1, reinterpret_cast<const float4*>(part + (size_t)(s + 1) * slab_elems)[i4]);
        t2 = f4_add(t2, reinterpret_cast<const float4*>(part + (size_t)(s + 2) * slab_elems)[i4]);
        t3 = f4_add(t3, reinterpret_cast<const float4*>(part + (size_t)(s + 3) * slab_elems)[i4]);
    }
    for (; s < nslab; ++s) t0 = f4_add(t0, reinterpret_cast<const float4*>(part + (size_t)s * slab_elems)[i4]);
    float4 t = f4_add(f4_add(t0, t1), f4_add(t2, t3));
    if (bias) t = f4_add(t, reinterpret_cast<const float4*>(bias)[i4 & 15]);
    reinterpret_cast<float4*>(out)[i4] = t;
    if (redo) {     // split-operand forward (64 columns = 16 threads per row): non-finite sums, rows of tiny magnitude -> redo in fp32
        const float inf = __builtin_inff();
        bool bad = !(fabsf(t.x) < inf) | !(fabsf(t.y) < inf) | !(fabsf(t.z) < inf) | !(fabsf(t.w) < inf);
        const size_t row = i4 >> 4, nrows = slab_elems >> 6;
        if ((i4 & 15) == 0) {
            float mx = 0.f;
            for (int q = 0; q < nslab; ++q) mx = fmaxf(mx, rowmax_part[(size_t)q * nrows + row]);
            bad |= mx > 0.f && mx < SPLIT_ROW_MIN;
        }
        if (bad) redo[row / LIN_BM] = 1;
    }
}

// ------------------------------------------------------------------------------------- backward W
// dW partial[o][f] over an item chunk.  grid (ceil(F/128), nsplit).  D[i=o][j=f]; A[i][kk=item] =
// dY[item][o]; B[kk=item][j] = X[item][f]: both operands are item-major in memory, so the LDS tiles
// keep the natural layout and fragment reads are lane-consecutive (conflict-free ds_read_b32).
constexpr int BW_BK = 32, BW_BF = 128;
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ dY,
                                                           const float* __restrict__ X,
                                                           float* __restrict__ part,
                                                           float* __restrict__ dbpart, int n, int F,
                                                           int n_chunk, int ldg, size_t zs_part, size_t zs_db) {
    // blockIdx.z = the 64-output group (out > 64: MMGCN's 256 / 384-wide layers): its columns of dY, its slabs, its db partials
    dY += 64 * blockIdx.z;
    part += blockIdx.z * zs_part;
    if (dbpart) dbpart += blockIdx.z * zs_db;
    __shared__ __attribute__((aligned(16))) float Gs[BW_BK][64];
    __shared__ __attribute__((aligned(16))) float Xs[BW_BK][BW_BF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = blockIdx.x * BW_BF;
    const int nb = blockIdx.y * n_chunk, ne = min(nb + n_chunk, n);
    f32x16 acc0 = {0}, acc1 = {0};
    float dbacc = 0.f;
    float4 gr[2], xr[4];
    // staging: Gs 32x64 floats = 512 float4 (2/thread); Xs 32x128 = 1024 float4 (4/thread)
    auto gload = [&](int it0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = tid + 256 * p, r = e >> 4, c = (e & 15) * 4;
            gr[p] = ld4_guard(dY + (size_t)(it0 + r) * ldg + c, it0 + r < ne);   // ldg: row stride of dY (64-column block of a wider dY)
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = tid + 256 * p, r = e >> 5, c = (e & 31) * 4;
            xr[p] = ld4_guard(X + (size_t)(it0 + r) * F + f0 + c, it0 + r < ne && f0 + c < F);
        }
    };
    gload(nb);
    const int i = lane & 31, h = lane >> 5;
    for (int it0 = nb; it0 < ne; it0 += BW_BK) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = tid + 256 * p;
            *reinterpret_cast<float4*>(&Gs[e >> 4][(e & 15) * 4]) = gr[p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = tid + 256 * p;
            *reinterpret_cast<float4*>(&Xs[e >> 5][(e & 31) * 4]) = xr[p];
        }
        __syncthreads();
        if (it0 + BW_BK < ne) gload(it0 + BW_BK);
        if (dbpart && blockIdx.x == 0 && tid < 64) {  // bias gradient: column sums of this chunk
#pragma unroll
            for (int k = 0; k < BW_BK; ++k) dbacc += Gs[k][tid];
        }
#pragma unroll
        for (int s = 0; s < BW_BK / 2; ++s) {
            const int k = 2 * s + h;
            const float b = Xs[k][wave * 32 + i];
            acc0 = mfma32(Gs[k][i], b, acc0);
            acc1 = mfma32(Gs[k][32 + i], b, acc1);
        }
        __syncthreads();
    }
    if (dbpart && blockIdx.x == 0 && tid < 64) dbpart[blockIdx.y * 64 + tid] = dbacc;
    float* dst = part + (size_t)blockIdx.y * 64 * F;
    const int f = f0 + wave * 32 + i;
    if (f < F) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            dst[(size_t)o * F + f] = acc0[r];
            dst[(size_t)(32 + o) * F + f] = acc1[r];
        }
    }
}

// dW on the LDS-DMA pipeline (same contract as linear_bwd_w_kernel): both operand tiles keep their
// natural item-major layout, which is exactly what an LDS-DMA piece writes (Xs: 2 items x 512 B per
// piece, Gs: 4 items x 256 B), so no swizzle is needed and the fragment reads stay lane-consecutive
// ds_read_b32.  3-stage ring of 32-item tiles, one barrier per tile, tile t+2 in flight under the 32
// MFMAs per wave of tile t.  SRSRC bounds over the workgroup's item chunk zero-fill the items past it.
// Needs n_chunk * F * 4 < 2^31 (scalar byte offsets); the caller falls back otherwise.
__global__ __launch_bounds__(256, 2) void linear_bwd_w_dma_kernel(const float* __restrict__ dY,
                                                                  const float* __restrict__ X,
                                                                  float* __restrict__ part,
                                                                  float* __restrict__ dbpart, int n, int F,
                                                                  int n_chunk, int ldg, size_t zs_part, size_t zs_db) {
    // blockIdx.z = the 64-output group (out > 64: MMGCN's 256 / 384-wide layers): its columns of dY, its slabs, its db partials
    dY += 64 * blockIdx.z;
    part += blockIdx.z * zs_part;
    if (dbpart) dbpart += blockIdx.z * zs_db;
    __shared__ __attribute__((aligned(1024))) float G0[BW_BK * 64], G1[BW_BK * 64], G2[BW_BK * 64];
    __shared__ __attribute__((aligned(1024))) float X0[BW_BK * BW_BF], X1[BW_BK * BW_BF], X2[BW_BK * BW_BF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = blockIdx.x * BW_BF;
    const int nb = blockIdx.y * n_chunk, ne = min(nb + n_chunk, n);
    const int rows = ne - nb;
    const i32x4 rx = raw_rsrc(X + (size_t)nb * F + f0, (unsigned)rows * (unsigned)F * 4u - (unsigned)f0 * 4u);
    const i32x4 rg = raw_rsrc(dY + (size_t)nb * ldg, (unsigned)(rows - 1) * (unsigned)ldg * 4u + 256u);
    int vx[4], vg[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) vx[j] = (2 * (4 * wave + j) + (lane >> 5)) * F * 4 + (lane & 31) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) vg[j] = (4 * (2 * wave + j) + (lane >> 4)) * ldg * 4 + (lane & 15) * 16;
    auto issue = [&](float* gs, float* xs, int t) {
        const int sx = t * BW_BK * F * 4, sg = t * BW_BK * ldg * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16<false>(rx, lds_addr(xs + (4 * wave + j) * 256), vx[j], sx);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16<false>(rg, lds_addr(gs + (2 * wave + j) * 256), vg[j], sg);
    };
    const int i = lane & 31, h = lane >> 5;
    const bool do_db = dbpart && blockIdx.x == 0 && tid < 64;
    f32x16 acc0 = {0}, acc1 = {0};
    float dbacc = 0.f;
    auto compute = [&](const float* gs, const float* xs) {
        if (do_db) {
#pragma unroll
            for (int k = 0; k < BW_BK; ++k) dbacc += gs[k * 64 + tid];
        }
#pragma unroll
        for (int s = 0; s < BW_BK / 2; ++s) {
            const int k = 2 * s + h;
            const float b = xs[k * BW_BF + wave * 32 + i];
            acc0 = mfma32(gs[k * 64 + i], b, acc0);
            acc1 = mfma32(gs[k * 64 + 32 + i], b, acc1);
        }
    };
    const int T = (rows + BW_BK - 1) / BW_BK;
    auto step = [&](const float* gc, const float* xc, float* gn, float* xn, int t) {
        if (t + 1 < T) MMREC_WAIT_VM(6); else MMREC_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        if (t + 2 < T) issue(gn, xn, t + 2);
        compute(gc, xc);
    };
    if (T > 0) issue(G0, X0, 0);
    if (T > 1) issue(G1, X1, 1);
    for (int t = 0; t < T;) {
        step(G0, X0, G2, X2, t); if (++t >= T) break;
        step(G1, X1, G0, X0, t); if (++t >= T) break;
        step(G2, X2, G1, X1, t); ++t;
    }
    if (do_db) dbpart[blockIdx.y * 64 + tid] = dbacc;
    float* dst = part + (size_t)blockIdx.y * 64 * F;
    const int f = f0 + wave * 32 + i;
    if (f < F) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            dst[(size_t)o * F + f] = acc0[r];
            dst[(size_t)(32 + o) * F + f] = acc1[r];
        }
    }
}

// db[c] = sum_s dbpart[s][c]: 16 slices of the splits in parallel, combined in slice order.
__global__ __launch_bounds__(1024) void db_reduce_kernel(const float* __restrict__ dbpart, int nsplit,
                                                         float* __restrict__ db) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    dbpart += (size_t)blockIdx.x * nsplit * 64;           // (one workgroup per 64-output group)
    db += 64 * blockIdx.x;
    float t = 0.f;
    for (int s = sl; s < nsplit; s += 16) t += dbpart[s * 64 + c];
    red[sl][c] = t;
    __syncthreads();
    if (sl == 0) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += red[k][c];
        db[c] = r;
    }
}

// ------------------------------------------------------------------------------------- backward X
// dX[item][f] = sum_o dY[item][o] W[o][f].  grid (ceil(n/128), ceil(F/128)); wave w owns items
// w*32..+31 and four 32-wide f sub-tiles (4 accumulators) that share its A fragment.  K = 64 fits one
// LDS tile: dY transposed-read (padded rows), W natural.
constexpr int BX_LD = 64 + 4;
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ dY,
                                                           const float* __restrict__ W,
                                                           float* __restrict__ dX, int n, int F) {
    __shared__ __attribute__((aligned(16))) float Gs[128][BX_LD];
    __shared__ __attribute__((aligned(16))) float Ws[64][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 128, f0 = blockIdx.y * 128;
    // Gs: 128x64 floats = 2048 float4 (8/thread); Ws: 64x128 = 2048 float4 (8/thread)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int e = tid + 256 * p;
        const int r = e >> 4, c = (e & 15) * 4;
        *reinterpret_cast<float4*>(&Gs[r][c]) = ld4_guard(dY + (size_t)(m0 + r) * 64 + c, m0 + r < n);
        const int o = e >> 5, fc = (e & 31) * 4;
        *reinterpret_cast<float4*>(&Ws[o][fc]) = ld4_guard(W + (size_t)o * F + f0 + fc, f0 + fc < F);
    }
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
        const float4 a = *reinterpret_cast<const float4*>(&Gs[wave * 32 + i][k8 * 8 + 4 * h]);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k8 * 8 + 4 * h + u;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma32(av[u], Ws[k][t * 32 + i], acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int f = f0 + t * 32 + i;
        if (f < F) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 32 + d_row(r, lane);
                if (row < n) dX[(size_t)row * F + f] = acc[t][r];
            }
        }
    }
}

// =====================================================================================================================
// Projection BACKWARD on the 16-bit matrix cores with split operands (ABI 11: mmrec_linear_bwd_split_f32).
// dW = dY^T X and dX = dY W are fp32-MFMA bound in the kernels above (2 n F 64 FLOP each at the 1/16-rate fp32 matrix pipe:
// 23.5 us at Amazon-Baby size at the nominal clock, 36 us measured) although each only has to stream X once (dW) or write dX once.
//   * dW (contraction over the ITEMS: gradients of one batch span 40 binades along it, features are what they are): three-way
//     bf16 split of both operands, six products per 16 k, no scales, no guards -- linear_bwd_w_bf16x3_kernel;
//   * dX (contraction over the 64 outputs): the forward's two fp16 halves, three products, with the operands brought into fp16's
//     range by exact power-of-two scales that are undone in the epilogue: one per ROW of dY, in registers (K = 64: a lane pair
//     holds the whole row), and one per COLUMN f of W, applied when W is written transposed + split (bwd_wt_split_kernel).
// Inf / NaN anywhere give non-finite results as F.linear's backward does.
// =====================================================================================================================
__device__ __forceinline__ void split8(const float (&x)[8], g_half8& hi, g_half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)((x[e] - (float)h) * 2048.f);
    }
}
// exact power-of-two scale that brings a maximum magnitude `mx` into [2^target, 2^(target + 1)), and its inverse; 1 for
// mx == 0 and for inf / NaN (which then propagate); clamped for maxima below 2^-100
__device__ __forceinline__ void pow2_scale(float mx, int target, float& sc, float& inv) {
    const unsigned bits = __float_as_uint(mx) & 0x7fffffffu;
    const int ex = (int)(bits >> 23);
    if (bits == 0u || ex == 255) { sc = 1.f; inv = 1.f; return; }
    int e = target + 127 - ex;                       // sc = 2^e
    e = max(-120, min(120, e));
    sc = __uint_as_float((unsigned)(e + 127) << 23);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
}

// W [64][F] -> Wt_sp: row f = 256 B = 16 chunks of 16 B: chunks 0..7 the fp16 hi halves of cs_f w[8c .. 8c + 7][f], chunks 8..15
// the lo' halves; chunk c is stored at position c ^ (f & 15) (the LDS-DMA copies a 128-row tile linearly; the swizzle makes the
// 16-B fragment reads of 32 consecutive rows conflict free).  wcs_inv[f] = 1 / cs_f.
__device__ __forceinline__ void bwd_wt_split_body(int block, const float* __restrict__ W, int F, float* __restrict__ Wt_sp,
                                                  float* __restrict__ wcs_inv) {
    const int f = block * 256 + threadIdx.x;
    if (f >= F) return;
    float w[64], mx = 0.f;
#pragma unroll
    for (int o = 0; o < 64; ++o) {
        w[o] = W[(size_t)o * F + f];
        mx = fmaxf(mx, fabsf(w[o]));       // (NaN: fmaxf drops it; the products below still carry it)
    }
    float cs, inv;
    pow2_scale(mx, 0, cs, inv);
    wcs_inv[f] = inv;
    float* row = Wt_sp + (size_t)f * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w[8 * c + e] * cs;
        g_half8 hi, lo;
        split8(x, hi, lo);
        *reinterpret_cast<g_half8*>(row + ((c ^ (f & 15)) << 2)) = hi;
        *reinterpret_cast<g_half8*>(row + (((8 + c) ^ (f & 15)) << 2)) = lo;
    }
}
__global__ __launch_bounds__(256) void bwd_wt_split_kernel(const float* __restrict__ W, int F, float* __restrict__ Wt_sp,
                                                           float* __restrict__ wcs_inv) {
    bwd_wt_split_body(blockIdx.x, W, F, Wt_sp, wcs_inv);
}

// dW partial[o][f] over an item chunk on v_mfma_f32_32x32x16_bf16 with THREE-WAY split operands: every fp32 number is
// b1 + b2 + b3 with b1 = bf16(x), b2 = bf16(x - b1), b3 = bf16(x - b1 - b2) -- 24 significand bits in three pieces that each
// carry fp32's 8-bit exponent, so there is nothing to scale and nothing to guard: gradients of 1e-30 next to gradients of 1e-3
// (a batch's BPR coefficients span 40 binades once pairs separate: measured on the Amazon-Sports-shaped FREEDOM step), features
// of any magnitude, inf / NaN (non-finite in, non-finite out; an inf may come out as NaN: inf - bf16(inf) is NaN) all take the
// same path.  x y = b1 c1 + (b1 c2 + b2 c1) + (b2 c2 + b1 c3 + b3 c1) + R, |R| <= 2^-22 |x y| (the three dropped products b2 c3,
// b3 c2, b3 c3 are each below 2^-24 |x y|; the bound every statement of this file and of mmrec_hip.h uses): six products per 16 k (the fp16 form of
// the forward takes three, eight fp32 MFMAs take 16 x the time), one fp32 accumulator set, smallest products first.
// The FIRST form of this kernel used the forward's two fp16 halves with one power-of-two scale per column of dY and a range
// guard; real gradient columns leave any single scale's range within an epoch (guard at 2^26: fired on most steps, at 2^30: on
// 16 % of the steps of epoch 0 and rising), each time sending dW to a slow fix-up -- profiles/r05_run_configs.log.
// Structure: linear_bwd_w_dma_kernel's (both operand tiles by LDS-DMA in their natural item-major layout, 3-stage ring, fused
// bias-gradient partials); X is split in registers by the wave that multiplies it, dY -- wanted by every wave -- once per
// workgroup into LDS fragments one tile ahead (the split is ~6 VALU per element).
typedef __attribute__((ext_vector_type(8))) __bf16 g_bf8;
__device__ __forceinline__ void split3(const float (&x)[8], g_bf8& p1, g_bf8& p2, g_bf8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 b1 = (__bf16)x[e];
        const float r1 = x[e] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        p1[e] = b1;
        p2[e] = b2;
        p3[e] = (__bf16)(r1 - (float)b2);
    }
}

// Eight waves, two per SIMD: wave w multiplies k-step (w >> 2) of every 32-item tile into its own accumulators for the 32 columns
// (w & 3); the two k-step halves are added through LDS at the end.  (With four waves -- one per SIMD running its LDS reads, ~170
// VALU and 24 MFMAs per tile back to back -- the kernel took 39.9 us at Amazon-Baby size; with two per SIMD one wave's MFMAs run
// under the other's split: -3.4 us forward + backward at Baby size, -15 us at Sports size, profiles/r05_linear_bwd_w_bf16x3_ab.log.)
#ifndef MMREC_BWD_W_STAGES
#define MMREC_BWD_W_STAGES 3     // X tiles in the LDS ring (S - 1 in flight per workgroup); 4, 5, 6 measured 1 ... 2 us slower at Amazon-Baby size: profiles/r05_linear_bwd_w_bf16x3_ab.log
#endif
typedef __attribute__((ext_vector_type(4))) __bf16 g_bf4;
template <bool NT>     // NT: X does not fit the Infinity Cache and is read once per call: streamed non-temporal (as in the forward)
__global__ __launch_bounds__(512) void linear_bwd_w_bf16x3_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                  float* __restrict__ part, float* __restrict__ dbpart, int n, int F,
                                                                  int n_chunk) {
    constexpr int S = MMREC_BWD_W_STAGES;
    __shared__ __attribute__((aligned(1024))) float Xr[S][BW_BK * BW_BF];     // X tiles [item][f], S-stage ring
    __shared__ __attribute__((aligned(1024))) float Gq[S - 1][BW_BK * 64];    // dY tiles [item][o] as they arrive
    __shared__ __attribute__((aligned(1024))) g_bf8 Asp[2][12 * 64];          // their split MFMA fragments [2 kstep + otile][part][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ft = wave & 3, ks = wave >> 2;
    const int f0 = blockIdx.x * BW_BF;
    const int nb = blockIdx.y * n_chunk, ne = min(nb + n_chunk, n);
    const int rows = ne - nb;
    const i32x4 rx = raw_rsrc(X + (size_t)nb * F + f0, (unsigned)rows * (unsigned)F * 4u - (unsigned)f0 * 4u);
    const i32x4 rg = raw_rsrc(dY + (size_t)nb * 64, (unsigned)rows * 256u);
    int vx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) vx[j] = (2 * (2 * wave + j) + (lane >> 5)) * F * 4 + (lane & 31) * 16;
    const int vg = (4 * wave + (lane >> 4)) * 256 + (lane & 15) * 16;
    auto issue = [&](int t, int gb, int xb) {      // the dY tile FIRST: it is wanted one step before its X tile
        lds_dma16<false>(rg, lds_addr(&Gq[gb][wave * 256]), vg, t * BW_BK * 256);
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16<NT>(rx, lds_addr(&Xr[xb][(2 * wave + j) * 256]), vx[j], t * BW_BK * F * 4);
    };
    const int i = lane & 31, h = lane >> 5;
    const bool do_db = dbpart && blockIdx.x == 0 && tid < 64;
    f32x16 acc0 = {0}, acc1 = {0};
    float dbacc = 0.f;
    // each dY element is split ONCE per workgroup: wave w owns half (w & 1) of fragment (kstep = w >> 2, otile = (w >> 1) & 1)
    auto split_g = [&](int gb, int ab) {
        const float* gq = Gq[gb];
        if (do_db) {
#pragma unroll
            for (int k = 0; k < BW_BK; ++k) dbacc += gq[k * 64 + tid];
        }
        const int frag = wave >> 1, half = wave & 1;
        const float* gr = gq + (16 * (frag >> 1) + 8 * h + 4 * half) * 64 + 32 * (frag & 1) + i;
        g_bf4 a1, a2, a3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = gr[e * 64];
            const __bf16 b1 = (__bf16)x;
            const float r1 = x - (float)b1;
            const __bf16 b2 = (__bf16)r1;
            a1[e] = b1;
            a2[e] = b2;
            a3[e] = (__bf16)(r1 - (float)b2);
        }
        g_bf4* dst = reinterpret_cast<g_bf4*>(&Asp[ab][frag * 3 * 64 + lane]) + half;
        dst[0] = a1;
        dst[128] = a2;
        dst[256] = a3;
    };
    auto compute = [&](int ab, int xb) {
        const float* xr = Xr[xb] + (16 * ks + 8 * h) * BW_BF + 32 * ft + i;
        g_bf8 b1, b2, b3;
        {
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = xr[e * BW_BF];
            split3(xv, b1, b2, b3);
        }
        {   // output tile 0 (o = i)
            const g_bf8* ap = &Asp[ab][(2 * ks) * 3 * 64 + lane];
            const g_bf8 a1 = ap[0], a2 = ap[64], a3 = ap[128];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc0, 0, 0, 0);
        }
        {   // output tile 1 (o = 32 + i)
            const g_bf8* ap = &Asp[ab][(2 * ks + 1) * 3 * 64 + lane];
            const g_bf8 a1 = ap[0], a2 = ap[64], a3 = ap[128];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
        }
    };
    // vmcnt queue: [G(0) X(0)] [G(1) X(1)] ... (1 + 2 copies per tile and wave), S - 1 tiles ahead, tiles past the end included
    // (zero fill): step t wants X(t) and G(t + 1), i.e. at most X(t + 1) and the S - 3 groups behind it still in flight.
    const int T = (rows + BW_BK - 1) / BW_BK;
    if (T > 0) {
#pragma unroll
        for (int u = 0; u < S - 1; ++u) issue(u, u, u);
        MMREC_WAIT_VM(2 + 3 * (S - 2));
        __builtin_amdgcn_s_barrier();
        split_g(0, 0);
    }
    int xb = 0, gb = 0;                                          // t % S, t % (S - 1)
    for (int t = 0; t < T; ++t) {
        MMREC_WAIT_VM(2 + 3 * (S - 3));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's part of tile t's fragments is in Asp
        __builtin_amdgcn_s_barrier();
        const int gn = gb == S - 2 ? 0 : gb + 1;
        // (the two waves of a SIMD running these phases in opposite order -- one's MFMAs under the other's copies and splits --
        // measured 2 us slower at Baby size, 10 us at Sports size: the later copies cost more than the overlap gives)
        issue(t + S - 1, gb, xb == 0 ? S - 1 : xb - 1);
        if (t + 1 < T) split_g(gn, (t + 1) & 1);
        compute(t & 1, xb);
        xb = xb == S - 1 ? 0 : xb + 1;
        gb = gn;
    }
    MMREC_WAIT_VM(0);      // (the zero-fill copies of the tiles past the end)
    if (do_db) dbpart[blockIdx.y * 64 + tid] = dbacc;
    // the two k-step halves: waves 4-7 hand theirs over through LDS (the X ring is free now: 64 x 128 floats)
    float* red = &Xr[0][0];
    static_assert(S * BW_BK * BW_BF >= 64 * BW_BF, "the X ring holds the 64 x 128 hand-over tile");
    __builtin_amdgcn_s_barrier();
    if (ks == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            red[o * BW_BF + ft * 32 + i] = acc0[r];
            red[(32 + o) * BW_BF + ft * 32 + i] = acc1[r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks == 0) {
        float* dst = part + (size_t)blockIdx.y * 64 * F;
        const int f = f0 + ft * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            dst[(size_t)o * F + f] = acc0[r] + red[o * BW_BF + ft * 32 + i];
            dst[(size_t)(32 + o) * F + f] = acc1[r] + red[(32 + o) * BW_BF + ft * 32 + i];
        }
    }
}

// ---- Round 6: the same product with the dY side split ONCE PER CALL instead of once per workgroup ---------------------------
// In the kernel above every one of the F / 128 column-block workgroups splits the same 32 x 64 dY tile again (a third of its VALU
// work), through an LDS round trip (Gq -> registers -> Asp) that sits between the tile's barrier and its MFMAs, and the workgroups
// with blockIdx.x == 0 also carry the bias gradient.  At 39 us for Amazon-Baby (X at 2.9 TB/s, MFMA pipe busy 30 %) nothing was
// near a hardware limit: per 32-item tile a CU spent ~3,400 cycles where its MFMAs take 768 and its VALU work ~600 -- latency
// chains between one barrier and the next, with a single 8-wave workgroup per CU to hide them.
//   bwd_dy_split_kernel   dY -> the three bf16 parts of every tile, ALREADY in MFMA A-fragment layout (12 KB per 32-item tile,
//                         [2 kstep + otile][part][lane] x 16 B: the old Asp image), + the bias gradient's partial sums;
//   linear_bwd_w_v2       X tiles by LDS-DMA (3-stage ring) AND the ready-made A fragments by LDS-DMA (linear 12-KB copies: no
//                         VALU, no LDS write, nothing between barrier and MFMAs but the X split); 72 KB of LDS, so TWO
//                         workgroups share a CU and one's barrier wait runs under the other's MFMAs.
// Same six products in the same order into the same accumulators as the kernel above: bit-identical dW; db's partial sums are
// regrouped (per dy-split workgroup instead of per item chunk), fixed order, still free of atomics.
#ifndef MMREC_BWD_W_V2
#define MMREC_BWD_W_V2 1        // 0: the round-5 kernel (A/B: tools/prof_linear.py build-variants v1=-DMMREC_BWD_W_V2=0)
#endif
#ifndef MMREC_BWD_W2_AS
#define MMREC_BWD_W2_AS 2       // A-fragment tiles in the LDS ring (2: issued one tile ahead; 3: two ahead, 84 KB = one workgroup per CU)
#endif
#ifndef MMREC_BWD_W2_OCC
#define MMREC_BWD_W2_OCC 2      // workgroups per CU the kernel is compiled for (waves per SIMD = 2 x this)
#endif
constexpr int BW2_ATILE = 12 * 64 * 16;      // bytes of a tile's split dY fragments

// grid: min(#tiles, 256) workgroups of 4 waves; workgroup g takes tiles g, g + G, ...; wave w = fragment (kstep = w >> 1, otile = w & 1).
__device__ __forceinline__ void bwd_dy_split_body(int block, int n_blocks, const float* __restrict__ dY, int n,
                                                  g_bf8* __restrict__ Asp_g, float* __restrict__ dbpart) {
    __shared__ float s_db[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int n_tiles = (n + BW_BK - 1) / BW_BK;
    float dbacc = 0.f;
    for (int t = block; t < n_tiles; t += n_blocks) {
        const int item0 = t * BW_BK + 16 * (wave >> 1) + 8 * h;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = item0 + e < n ? dY[(size_t)(item0 + e) * 64 + 32 * (wave & 1) + i] : 0.f;
        g_bf8 p1, p2, p3;
        split3(x, p1, p2, p3);
        g_bf8* dst = Asp_g + (size_t)t * (12 * 64) + wave * 3 * 64 + lane;
        dst[0] = p1;
        dst[64] = p2;
        dst[128] = p3;
#pragma unroll
        for (int e = 0; e < 8; ++e) dbacc += x[e];
    }
    if (dbpart) {       // column o = 32 (wave & 1) + i: the partial sums of (kstep, h) in fixed order
        s_db[wave][lane] = dbacc;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int ot = threadIdx.x >> 5, c = threadIdx.x & 31;
            dbpart[block * 64 + threadIdx.x] = (s_db[ot][c] + s_db[ot][32 + c]) + (s_db[2 + ot][c] + s_db[2 + ot][32 + c]);
        }
    }
}
__global__ __launch_bounds__(256) void bwd_dy_split_kernel(const float* __restrict__ dY, int n, g_bf8* __restrict__ Asp_g,
                                                          float* __restrict__ dbpart) {
    bwd_dy_split_body(blockIdx.x, gridDim.x, dY, n, Asp_g, dbpart);
}
// Both operand preparations of a backward call that wants dW AND dX in ONE launch (as launches of their own they take ~5 us
// each at Amazon-Baby size -- a sixth of the call was spent in six such kernels): workgroups [0, dy_wgs) split dY, the rest W^T.
__global__ __launch_bounds__(256) void bwd_prep_kernel(const float* __restrict__ dY, int n, g_bf8* __restrict__ Asp_g,
                                                      float* __restrict__ dbpart, int dy_wgs, const float* __restrict__ W, int F,
                                                      float* __restrict__ Wt_sp, float* __restrict__ wcs_inv) {
    if ((int)blockIdx.x < dy_wgs) bwd_dy_split_body(blockIdx.x, dy_wgs, dY, n, Asp_g, dbpart);
    else bwd_wt_split_body(blockIdx.x - dy_wgs, W, F, Wt_sp, wcs_inv);
}
// dW = sum of the item chunks' slabs (slab order) and, in the LAST workgroup, db = sum of the bias-gradient partials (four
// slices of the partials in parallel, combined in slice order): one launch instead of slab_reduce_kernel + db_reduce_kernel.
__global__ __launch_bounds__(256) void bwd_w_finish_kernel(const float* __restrict__ part, int nslab, size_t slab_elems,
                                                          float* __restrict__ dW, const float* __restrict__ dbpart, int n_dbpart,
                                                          float* __restrict__ db) {
    __shared__ float red[4][64];
    if (blockIdx.x + 1 == gridDim.x) {
        if (!db) return;
        const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
        float t = 0.f, t2 = 0.f;
        int s = sl;
        for (; s + 4 < n_dbpart; s += 8) {          // two independent chains: the loads travel together
            t += dbpart[s * 64 + c];
            t2 += dbpart[(s + 4) * 64 + c];
        }
        if (s < n_dbpart) t += dbpart[s * 64 + c];
        t += t2;
        red[sl][c] = t;
        __syncthreads();
        if (sl == 0) db[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        return;
    }
    if (nslab <= 1) return;            // (a single chunk wrote dW itself)
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 * 4 >= slab_elems) return;
    float4 t0 = f4_zero(), t1 = f4_zero(), t2 = f4_zero(), t3 = f4_zero();
    int s = 0;
    for (; s + 3 < nslab; s += 4) {      // four independent chains, fixed combination order (as slab_reduce_kernel)
        t0 = f4_add(t0, reinterpret_cast<const float4*>(part + (size_t)(s + 0) * slab_elems)[i4]);
        t1 = f4_add(t1, reinterpret_cast<const float4*>(part + (size_t)(s + 1) * slab_elems)[i4]);
        t2 = f4_add(t2, reinterpret_cast<const float4*>(part + (size_t)(s + 2) * slab_elems)[i4]);
        t3 = f4_add(t3, reinterpret_cast<const float4*>(part + (size_t)(s + 3) * slab_elems)[i4]);
    }
    for (; s < nslab; ++s) t0 = f4_add(t0, reinterpret_cast<const float4*>(part + (size_t)s * slab_elems)[i4]);
    reinterpret_cast<float4*>(dW)[i4] = f4_add(f4_add(t0, t1), f4_add(t2, t3));
}

template <bool NT>
__global__ __launch_bounds__(512, 2 * MMREC_BWD_W2_OCC) void linear_bwd_w_v2_kernel(const g_bf8* __restrict__ Asp_g,
                                                                                    const float* __restrict__ X,
                                                                                    float* __restrict__ part, int n, int F, int n_chunk) {
    constexpr int S = 3, AS = MMREC_BWD_W2_AS;
    __shared__ __attribute__((aligned(1024))) float Xr[S][BW_BK * BW_BF];      // X tiles [item][f], 3-stage ring (48 KB)
    __shared__ __attribute__((aligned(1024))) g_bf8 Ar[AS][12 * 64];           // split dY fragments of a tile (12 KB each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ft = wave & 3, ks = wave >> 2;
    const int f0 = blockIdx.x * BW_BF;
    const int nb = blockIdx.y * n_chunk, ne = min(nb + n_chunk, n);
    const int rows = ne - nb;
    const int T = (rows + BW_BK - 1) / BW_BK;
    const i32x4 rx = raw_rsrc(X + (size_t)nb * F + f0, (unsigned)rows * (unsigned)F * 4u - (unsigned)f0 * 4u);
    const i32x4 ra = raw_rsrc(reinterpret_cast<const char*>(Asp_g) + (size_t)(nb / BW_BK) * BW2_ATILE, (unsigned)T * (unsigned)BW2_ATILE);
    int vx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) vx[j] = (2 * (2 * wave + j) + (lane >> 5)) * F * 4 + (lane & 31) * 16;
    const int va = lane * 16;
    // a tile's fragments are 12 linear 1-KB pieces: wave w brings piece w, waves 0-3 also piece 8 + w
    auto issue_a = [&](int t) {
        const unsigned dst = lds_addr(reinterpret_cast<const float*>(&Ar[t % AS][0]));
        lds_dma16<false>(ra, dst + wave * 1024, va, t * BW2_ATILE + wave * 1024);
        if (wave < 4) lds_dma16<false>(ra, dst + (8 + wave) * 1024, va, t * BW2_ATILE + (8 + wave) * 1024);
    };
    auto issue_x = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16<NT>(rx, lds_addr(&Xr[t % S][(2 * wave + j) * 256]), vx[j], t * BW_BK * F * 4);
    };
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc0 = {0}, acc1 = {0};
    auto compute = [&](int ab, int xb) {
        const float* xr = Xr[xb] + (16 * ks + 8 * h) * BW_BF + 32 * ft + i;
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = xr[e * BW_BF];
        const g_bf8* ap = &Ar[ab][(2 * ks) * 3 * 64 + lane];
        const g_bf8 a1 = ap[0], a2 = ap[64], a3 = ap[128], c1 = ap[192], c2 = ap[256], c3 = ap[320];
        g_bf8 b1, b2, b3;
        split3(xv, b1, b2, b3);
        // smallest products first, exactly as in the round-5 kernel
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c3, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b3, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b1, acc1, 0, 0, 0);
    };
    // vmcnt queue of a wave (AS = 2):  A(0) X(0) X(1) | A(1) X(2) | A(2) X(3) | ...   -- step t issues A(t + 1) BEFORE X(t + 2), so
    // that "all but the two newest copies have landed" means X(t) and A(t) whatever a wave's number of A pieces.
    // (AS = 3: A(0) A(1) X(0) X(1) | A(2) X(2) | ...: the newest 2 + nA copies may be in flight.)  Tiles past the end are issued
    // too: out of the descriptors' range, they fill with zeros and are never multiplied.
    if (T > 0) {
        issue_a(0);
        if (AS == 3) issue_a(1);
        issue_x(0);
        issue_x(1);
    }
    for (int t = 0; t < T; ++t) {
        if (AS == 3) {
            if (wave < 4) MMREC_WAIT_VM(4); else MMREC_WAIT_VM(3);
        } else {
            MMREC_WAIT_VM(2);
        }
        __builtin_amdgcn_s_barrier();       // everyone's pieces of tile t are in LDS; everyone has read tile t - 1
        issue_a(t + AS - 1);
        issue_x(t + 2);
        compute(t % AS, t % S);
    }
    MMREC_WAIT_VM(0);      // (the zero-fill copies of the tiles past the end)
    // the two k-step halves: waves 4-7 hand theirs over through LDS (the X ring is free now: 64 x 128 floats)
    float* red = &Xr[0][0];
    static_assert(S * BW_BK * BW_BF >= 64 * BW_BF, "the X ring holds the 64 x 128 hand-over tile");
    __builtin_amdgcn_s_barrier();
    if (ks == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            red[o * BW_BF + ft * 32 + i] = acc0[r];
            red[(32 + o) * BW_BF + ft * 32 + i] = acc1[r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks == 0) {
        float* dst = part + (size_t)blockIdx.y * 64 * F;
        const int f = f0 + ft * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = d_row(r, lane);
            dst[(size_t)o * F + f] = acc0[r] + red[o * BW_BF + ft * 32 + i];
            dst[(size_t)(32 + o) * F + f] = acc1[r] + red[(32 + o) * BW_BF + ft * 32 + i];
        }
    }
}

// dX[n, F] = dY[n, 64] W[64, F] on v_mfma_f32_32x32x16_f16, the streaming form of gemm64_stream_kernel (mfma_stream.h): a
// workgroup owns 128 rows and walks `ftiles` 128-column tiles; waves 0-3 keep their 32 x 64 dY fragment -- scaled per row,
// split -- in registers for the whole walk and issue LDS reads, 12 MFMAs and 16 row-segment stores per 32-column sub-tile; wave 4
// brings the pre-split W^T tiles (32 KB, linear) and their 128 column scales by LDS-DMA into a double buffer and is the only wave
// that waits on vmcnt.  Output-write bound: n F 4 bytes.
template <int STORE_AUX>     // cache policy bits of the dX stores (2 = nt: a dX larger than the Infinity Cache is written once and not re-read here)
__global__ __launch_bounds__(320, 2) void bwd_x_f16x3_kernel(const float* __restrict__ dY, const float* __restrict__ Wt_sp,
                                                             const float* __restrict__ wcs_inv, float* __restrict__ dX, int n,
                                                             int F, int ftiles) {
    __shared__ __attribute__((aligned(1024))) float Wa[8192 + 256], Wb[8192 + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 128;
    const int ft0 = blockIdx.y * ftiles, ftn = min(ftiles, F / 128 - ft0);
    if (wave == 4) {  // ------------------------------------------------------------ loader wave
        const i32x4 rw = raw_rsrc(Wt_sp, (unsigned)F * 256u);
        const i32x4 rs = raw_rsrc(wcs_inv, (unsigned)F * 4u);
        auto fill = [&](float* ws, int ft) {
            const int so = (ft0 + ft) * 32768;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) lds_dma16<false>(rw, lds_addr(ws + j * 256), lane * 16, so + j * 1024);
            lds_dma16<false>(rs, lds_addr(ws + 8192), lane * 16, (ft0 + ft) * 512);     // (lanes 32-63: the next tile's, unused)
            MMREC_WAIT_VM(0);
        };
        if (ftn > 0) fill(Wa, 0);
        for (int ft = 0; ft < ftn;) {
            __builtin_amdgcn_s_barrier();            // tile ft landed; the other buffer is drained
            if (ft + 1 < ftn) fill(Wb, ft + 1);
            if (++ft >= ftn) break;
            __builtin_amdgcn_s_barrier();
            if (ft + 1 < ftn) fill(Wa, ft + 1);
            ++ft;
        }
        return;
    }
    const int i = lane & 31, h = lane >> 5;
    // A fragments: row m0 + 32 wave + i of dY, k = 16 s + 8 h .. + 7 per MFMA step s; rows past n read as zero
    const int arow = m0 + wave * 32 + i;
    float a[4][8];
    float rmax = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const float4 u = ld4_guard(dY + (size_t)arow * 64 + 16 * st + 8 * h, arow < n);
        const float4 v = ld4_guard(dY + (size_t)arow * 64 + 16 * st + 8 * h + 4, arow < n);
        a[st][0] = u.x; a[st][1] = u.y; a[st][2] = u.z; a[st][3] = u.w;
        a[st][4] = v.x; a[st][5] = v.y; a[st][6] = v.z; a[st][7] = v.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) {     // (bit patterns: a NaN or inf element must surface in the maximum)
            const float ax = fabsf(a[st][e]);
            rmax = __uint_as_float(max(__float_as_uint(rmax), __float_as_uint(ax)));
        }
    }
    rmax = __uint_as_float(max(__float_as_uint(rmax), __float_as_uint(__shfl_xor(rmax, 32))));
    float rsc, rinv;
    pow2_scale(rmax, 0, rsc, rinv);
    g_half8 ah[4], al[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a[st][e] * rsc;
        split8(x, ah[st], al[st]);
    }
    float rs16[16];     // inverse row scale of the 16 rows this lane's accumulator registers belong to
#pragma unroll
    for (int r = 0; r < 16; ++r) rs16[r] = __shfl(rinv, d_row(r, lane));
    const unsigned lane_off = (unsigned)(4 * h * F + i) * 4u;
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dX + (size_t)m0 * F), 0, (unsigned)min(128, n - m0) * (unsigned)F * 4u, 0x00020000);   // rows past n: dropped
    auto tile = [&](const float* ws, int ft) {
        const int tcol = ((ft0 + ft) * 128) * 4;  // byte offset of this f tile within a row
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = 32 * t + i, sw = col & 15;
            const float* wr = ws + col * 64;
            f32x16 hh = {0}, cx = {0};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const g_half8 bh = *reinterpret_cast<const g_half8*>(wr + (((2 * st + h) ^ sw) << 2));
                const g_half8 bl = *reinterpret_cast<const g_half8*>(wr + (((8 + 2 * st + h) ^ sw) << 2));
                hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], bh, hh, 0, 0, 0);
                cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], bl, cx, 0, 0, 0);
                cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], bh, cx, 0, 0, 0);
            }
            const float cinv = ws[8192 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                // the two inverse scales are exact powers of two, each up to 2^+-120: where their PRODUCT leaves fp32's range
                // (inf: an exact 0 accumulator would become NaN; 0: a finite result would vanish -- round-5 advice) they are
                // applied one after the other instead
                const float x = fmaf(cx[r], 1.f / 2048.f, hh[r]), sc = rs16[r] * cinv;
                const float v = (sc != 0.f && fabsf(sc) < __builtin_inff()) ? x * sc : x * rs16[r] * cinv;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rdx, (int)lane_off,
                                                      (wave * 32 + rr) * F * 4 + tcol + t * 128, STORE_AUX);
            }
        }
    };
    for (int ft = 0; ft < ftn;) {
        __builtin_amdgcn_s_barrier();
        tile(Wa, ft);
        if (++ft >= ftn) break;
        __builtin_amdgcn_s_barrier();
        tile(Wb, ft);
        ++ft;
    }
}


inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Split of the contraction extent over workgroups.  With `tiles` output tiles and 256 CUs the
// makespan of one launch is ~ max(ceil(tiles*s / 256), 2) * (chunk(s) + slab) : every CU runs its
// workgroups' MFMA streams back to back, fewer than 2 workgroups per CU leave load latency and
// barriers exposed, and each split costs one partial slab (~128 columns' worth of traffic per tile).
// Pick the split count that minimises it; chunk is a multiple of `gran`.
inline void pick_split(int tiles, int extent, int gran, int* nsplit, int* chunk) {
    const int max_s = ceil_div(extent, gran);
    long best_cost = -1;
    int best_s = 1, best_c = ceil_div(extent, gran) * gran;
    for (int s = 1; s <= max_s && s <= 64; ++s) {
        const int c = ceil_div(ceil_div(extent, s), gran) * gran;
        const int real_s = ceil_div(extent, c);
        long rounds = ceil_div(tiles * real_s, 256);
        if (rounds < 2) rounds = 2;
        const long cost = rounds * (c + (real_s > 1 ? 128 : 0));
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_s = real_s;
            best_c = c;
        }
    }
    *nsplit = best_s;
    *chunk = best_c;
}

// The split-operand kernels stream X at 4-5 TB/s and are not bound per CU: measured (tools/prof_linear_replay.py, same lease,
// forced split counts: profiles/r05_linear_split_counts_ab.log) they want about ONE workgroup per CU and the longest chunks that
// still give that -- 8 / 4 / 2 k-splits of the forward at 4,096 / 7,050 / 23,033 rows (the fp32 cost model above picks 16 / 8 / 4)
// and 8 item-splits of dW at every size (16): -7.6 % forward + backward at 4,096 rows, -2.7 % at 7,050, -3.5 % at 23,033.
inline void pick_split_stream(int tiles, int extent, int gran, int* nsplit, int* chunk) {
    int s = 256 / tiles;                                   // at most one workgroup per CU ...
    // ... and between 129 and 255 row blocks: a CU runs its workgroups at about the rate of one (the kernel is bound per CU:
    // ~0.6 us per 128 x 32 tile whether one or two workgroups are resident), so the launch ends with the most loaded CU.  Two
    // chunks per block put two half-length workgroups on tiles - 128 ... CUs: the same makespan as one chunk plus a slab pass;
    // THREE chunks (<= 512 workgroups: all resident) cut it to 2/3 (Sports, 144 blocks: 95-97 us with two, 77-78 with three,
    // 81-83 with one; Clothing, 180 blocks: 540 workgroups would not be resident together -- one chunk, 86-97 against 95-101).
    if (s < 2) s = tiles <= 170 ? 3 : 1;
    const int max_s = ceil_div(extent, 2 * gran);          // a chunk is at least two tiles of the ring
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    *chunk = ceil_div(ceil_div(extent, s), gran) * gran;
    *nsplit = ceil_div(extent, *chunk);
}

}  // namespace

extern "C" size_t mmrec_linear_workspace_bytes(int32_t n, int32_t F, int32_t out) {
    if (n <= 0 || F <= 0 || out <= 0 || (out & 63)) return 0;   // out > 64: only the dW call uses it
    int s1, c1, s2, c2, s1s, c1s;
    pick_split(ceil_div(n, LIN_BM), F, LIN_BK, &s1, &c1);
    pick_split_stream(ceil_div(n, LIN_BM), F, LIN_BK, &s1s, &c1s);      // (the split-operand forward's rule)
    if (s1s > s1) s1 = s1s;
#ifdef MMREC_FWD_SPLIT      // probe builds force the forward's split count: room for their slabs
    if (s1 < MMREC_FWD_SPLIT + 1) s1 = MMREC_FWD_SPLIT + 1;
#endif
    pick_split(ceil_div(F, BW_BF), n, BW_BK, &s2, &c2);
    // forward: partial slabs, then (mmrec_linear_fwd_split_f32) the 64 x F split copy of W, the per-slab row maxima of |X| and
    // the redo flags of its domain check (one per 128-row block + one per W row)
    const size_t fwd = ((s1 > 1 ? (size_t)s1 * n * 64 * sizeof(float) : 0) + 255) / 256 * 256 + (size_t)64 * F * sizeof(float) +
                       (s1 > 1 ? (size_t)s1 * n * sizeof(float) : 0) + ((size_t)ceil_div(n, LIN_BM) + 64) * sizeof(int);
    const size_t bww = ((s2 > 1 ? (size_t)s2 * 64 * F : 0) + (size_t)s2 * 64) * sizeof(float) * (size_t)(out / 64);   // every 64-output group its own slabs
    return fwd > bww ? fwd : bww;
}

extern "C" int mmrec_linear_fwd_f32(const float* X, const float* W, const float* b, float* Y,
                                    int32_t n, int32_t F, int32_t out, void* workspace,
                                    mmrec_stream_t stream) {
    if (out != 64 || F <= 0 || (F & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!X || !W || !Y) return MMREC_ERR_BAD_ARG;
    int nsplit, chunk;
    pick_split(ceil_div(n, LIN_BM), F, LIN_BK, &nsplit, &chunk);
#ifdef MMREC_FWD_SPLIT
    chunk = ceil_div(ceil_div(F, MMREC_FWD_SPLIT), LIN_BK) * LIN_BK; nsplit = ceil_div(F, chunk);
#endif
    hipStream_t s = mmrec_stream(stream);
    const bool dma = (F % DM_BK) == 0 && !MMREC_GEMM_LEGACY_FWD;
    if (nsplit > 1 && !workspace) return MMREC_ERR_BAD_ARG;
    float* dst = nsplit == 1 ? Y : static_cast<float*>(workspace);
    const dim3 grid(ceil_div(n, LIN_BM), nsplit);
    // X larger than the 256 MB Infinity Cache is read once per call: stream it non-temporal
    const bool nt = (size_t)n * F * sizeof(float) > ((size_t)192 << 20);
    if (dma && nt)
        hipLaunchKernelGGL(linear_fwd_dma_kernel<true>, grid, dim3(256), 0, s, X, W, b, dst, n, F, chunk, (const int*)nullptr);
    else if (dma)
        hipLaunchKernelGGL(linear_fwd_dma_kernel<false>, grid, dim3(256), 0, s, X, W, b, dst, n, F, chunk, (const int*)nullptr);
    else
        hipLaunchKernelGGL(linear_fwd_kernel<MMREC_GEMM_PROBE_MODE>, grid, dim3(256),
                           MMREC_GEMM_DYN_LDS, s, X, W, b, dst, n, F, chunk);
    if (nsplit > 1) {
        float* part = dst;
        const size_t elems = (size_t)n * 64;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)), dim3(256), 0,
                           s, part, nsplit, elems, b, Y);
    }
    MMREC_RETURN_LAUNCH_STATUS();
}

// mmrec_linear_fwd_f32 on the 16-bit matrix cores with split operands (fp32-accurate: see linear_fwd_dma_f16x3_kernel).  F % 32 != 0
// takes the fp32 kernels.  Rows / weights outside the split's domain (|.| >= 65520, inf, NaN, or a whole row below 2^-10) are
// detected on the device and recomputed by the fp32 kernel in the same call (no host synchronisation): see the kernel's comment.
// Workspace: mmrec_linear_workspace_bytes (slabs | split W | per-slab row maxima | redo flags).
extern "C" int mmrec_linear_fwd_split_f32(const float* X, const float* W, const float* b, float* Y, int32_t n, int32_t F,
                                          int32_t out, void* workspace, mmrec_stream_t stream) {
    if (out != 64 || F <= 0 || (F & 3)) return MMREC_ERR_UNSUPPORTED;
    if ((F % DM_BK) != 0) return mmrec_linear_fwd_f32(X, W, b, Y, n, F, out, workspace, stream);
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!X || !W || !Y || !workspace) return MMREC_ERR_BAD_ARG;
    int nsplit, chunk;
    pick_split_stream(ceil_div(n, LIN_BM), F, LIN_BK, &nsplit, &chunk);
#ifdef MMREC_FWD_SPLIT      // probe (tools/prof_linear_replay.py ab): force the split-K count
    chunk = ceil_div(ceil_div(F, MMREC_FWD_SPLIT), LIN_BK) * LIN_BK; nsplit = ceil_div(F, chunk);
#endif
    hipStream_t s = mmrec_stream(stream);
    const int nblocks = ceil_div(n, LIN_BM);
    float* part = static_cast<float*>(workspace);
    float* Wsp = reinterpret_cast<float*>(static_cast<char*>(workspace) +
                                          ((nsplit > 1 ? (size_t)nsplit * n * 64 * sizeof(float) : 0) + 255) / 256 * 256);
    float* rowmax_part = Wsp + (size_t)64 * F;
    int* redo = reinterpret_cast<int*>(rowmax_part + (nsplit > 1 ? (size_t)nsplit * n : 0));
    hipLaunchKernelGGL(linear_w_split_kernel, dim3(64), dim3(256), 0, s, W, F, Wsp, redo, nblocks);
    float* dst = nsplit == 1 ? Y : part;
    const dim3 grid(nblocks, nsplit);
    const bool nt = (size_t)n * F * sizeof(float) > ((size_t)192 << 20);
    if (nt)
        hipLaunchKernelGGL(linear_fwd_dma_f16x3_kernel<true>, grid, dim3(256), 0, s, X, Wsp, b, dst, n, F, chunk, rowmax_part, redo);
    else
        hipLaunchKernelGGL(linear_fwd_dma_f16x3_kernel<false>, grid, dim3(256), 0, s, X, Wsp, b, dst, n, F, chunk, rowmax_part, redo);
    if (nsplit > 1) {
        const size_t elems = (size_t)n * 64;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)), dim3(256), 0, s, part, nsplit, elems,
                           b, Y, (const float*)rowmax_part, redo);
    }
    // fix-up: the fp32 kernel over the flagged 128-row blocks (the others return at once), whole K per workgroup
    if (nt)
        hipLaunchKernelGGL(linear_fwd_dma_kernel<true>, dim3(nblocks, 1), dim3(256), 0, s, X, W, b, Y, n, F, F, (const int*)redo);
    else
        hipLaunchKernelGGL(linear_fwd_dma_kernel<false>, dim3(nblocks, 1), dim3(256), 0, s, X, W, b, Y, n, F, F, (const int*)redo);
    MMREC_RETURN_LAUNCH_STATUS();
}

// ---- ABI 11: the projection's backward on the 16-bit matrix cores (see the kernels' comment block) ----------------------------
// Workspace layout (mmrec_linear_bwd_split_workspace_bytes): [W^T split: F x 256 B][wcs_inv: F + 256 floats][db partials:
// nsplit x 64 floats][dW slabs: nsplit > 1 ? nsplit x 64 x F floats]; never smaller than what the fp32 entry points need.
namespace {
struct BwdSplitWs {
    size_t wt, wcs, dbp, slabs, asp, total;
    int nsplit, chunk, dy_wgs;
};
inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }
inline BwdSplitWs bwd_split_ws(int n, int F) {
    BwdSplitWs w;
    pick_split_stream(ceil_div(F, BW_BF), n, BW_BK, &w.nsplit, &w.chunk);
    // (the v2 kernel fits two workgroups per CU, but twice the chunks measured 3 % SLOWER forward + backward at 7,050 rows and the
    // same at 18,357 -- twice the slabs to sum; a quarter of the chunks 10 % slower: profiles/r06_linear_bwd_w_v2_ab.log)
#ifdef MMREC_BWD_W_SPLIT    // probe: force the split-over-items count of the dW kernel
    w.chunk = ceil_div(ceil_div(n, MMREC_BWD_W_SPLIT), BW_BK) * BW_BK; w.nsplit = ceil_div(n, w.chunk);
#endif
    // LDS-DMA scalar offsets of the dW kernel: a workgroup's item chunk must stay below 2^31 bytes of X
    while ((size_t)(w.chunk + 8 * BW_BK) * F * 4 >= ((size_t)1 << 31)) {      // (+ the tiles issued past the end)
        w.chunk = ceil_div(w.chunk / 2, BW_BK) * BW_BK;
        w.nsplit = ceil_div(n, w.chunk);
    }
    size_t off = 0;
    w.wt = off; off += al256((size_t)F * 256);
    w.wcs = off; off += al256(((size_t)F + 256) * 4);
    const int n_tiles = ceil_div(n, BW_BK);
    // bwd_dy_split_kernel's grid = its bias-gradient partials, which ONE workgroup sums afterwards (64: 16 loads per thread; with
    // 256 partials that chain of loads made the finishing launch 16 us instead of 5)
    w.dy_wgs = n_tiles < 64 ? n_tiles : 64;
    w.dbp = off; off += al256((size_t)(w.nsplit > w.dy_wgs ? w.nsplit : w.dy_wgs) * 64 * 4);
    w.slabs = off; off += al256(w.nsplit > 1 ? (size_t)w.nsplit * 64 * F * 4 : 0);
    w.asp = off; off += MMREC_BWD_W_V2 ? al256((size_t)n_tiles * BW2_ATILE) : 0;      // dY split once per call: 384 B per item
    w.total = off;
    return w;
}
#ifndef MMREC_BWD_NT
#define MMREC_BWD_NT 1     // bit 0 = dW reads an X larger than the Infinity Cache non-temporal (measured: -14 % forward + backward at Sports / Clothing size), bit 1 = dX written non-temporal (measured: +4 ... +12 % at 62,500 / 500,000 rows: off); profiles/r05_linear_bwd_nt_ab.log, tools/prof_linear.py run-variants
#endif
constexpr bool BWD_X_NT_STORES = (MMREC_BWD_NT & 2) != 0;
inline bool bwd_split_serves(int n, int F, int out) { return out == 64 && n > 0 && F > 0 && (F % BW_BF) == 0; }
}  // namespace

extern "C" size_t mmrec_linear_bwd_split_workspace_bytes(int32_t n, int32_t F, int32_t out) {
    const size_t plain = mmrec_linear_workspace_bytes(n, F, out);
    if (!bwd_split_serves(n, F, out)) return plain;
    const size_t mine = bwd_split_ws(n, F).total;
    return mine > plain ? mine : plain;
}

// dW [64, F] = dY^T X, db [64] = column sums of dY, dX [n, F] = dY W in ONE call (any of dW+db / dX may be NULL: not wanted).
// out == 64 and F % 128 == 0 run the split-operand kernels; other shapes are handed to mmrec_linear_bwd_w_f32 /
// mmrec_linear_bwd_x_f32 (same workspace).  Results: fp32-accurate (dW: error <= 2^-22 of sum |a b| per output + fp32
// accumulation, any magnitudes; dX: 2^-21 with the operands scaled into fp16's range by exact powers of two).
extern "C" int mmrec_linear_bwd_split_f32(const float* dY, const float* X, const float* W, float* dW, float* db, float* dX,
                                          int32_t n, int32_t F, int32_t out, void* workspace, mmrec_stream_t stream) {
    if (n < 0 || F <= 0 || out <= 0) return MMREC_ERR_BAD_ARG;
    if (!bwd_split_serves(n, F, out)) {
        if (dW) {
            const int rc = mmrec_linear_bwd_w_f32(dY, X, dW, db, n, F, out, workspace, stream);
            if (rc) return rc;
        }
        return dX ? mmrec_linear_bwd_x_f32(dY, W, dX, n, F, out, stream) : 0;
    }
    if (!dY || !workspace || (dW && !X) || (dX && !W) || (db && !dW)) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    const BwdSplitWs w = bwd_split_ws(n, F);
    char* base = static_cast<char*>(workspace);
    float* Wt_sp = reinterpret_cast<float*>(base + w.wt);
    float* wcs_inv = reinterpret_cast<float*>(base + w.wcs);
    float* dbp = reinterpret_cast<float*>(base + w.dbp);
    float* slabs = reinterpret_cast<float*>(base + w.slabs);
    const int ncb = F / BW_BF;
    const bool big = (size_t)n * F * sizeof(float) > ((size_t)192 << 20);      // X / dX larger than the Infinity Cache
    bool wt_done = false;
#ifdef MMREC_BWD_W_FP32      // probe: dW by the fp32-MFMA kernel inside the split entry (A/B against the bf16 x 3 kernel)
    if (dW) {
        const int rc = mmrec_linear_bwd_w_f32(dY, X, dW, db, n, F, out, workspace, stream);
        if (rc) return rc;
        dW = nullptr;
    }
#endif
    if (dW) {
        float* part = w.nsplit == 1 ? dW : slabs;
        float* dbpart = db ? dbp : (float*)nullptr;
        int n_dbpart = w.nsplit;
#if MMREC_BWD_W_V2
        g_bf8* Asp_g = reinterpret_cast<g_bf8*>(base + w.asp);
        if (dX) {       // dY and W^T are split in one launch
            hipLaunchKernelGGL(bwd_prep_kernel, dim3(w.dy_wgs + ceil_div(F, 256)), dim3(256), 0, s, dY, n, Asp_g, dbpart, w.dy_wgs,
                               W, F, Wt_sp, wcs_inv);
            wt_done = true;
        } else {
            hipLaunchKernelGGL(bwd_dy_split_kernel, dim3(w.dy_wgs), dim3(256), 0, s, dY, n, Asp_g, dbpart);
        }
        n_dbpart = w.dy_wgs;
        if (big && (MMREC_BWD_NT & 1))
            hipLaunchKernelGGL(linear_bwd_w_v2_kernel<true>, dim3(ncb, w.nsplit), dim3(512), 0, s, (const g_bf8*)Asp_g, X, part, n, F, w.chunk);
        else
            hipLaunchKernelGGL(linear_bwd_w_v2_kernel<false>, dim3(ncb, w.nsplit), dim3(512), 0, s, (const g_bf8*)Asp_g, X, part, n, F, w.chunk);
#else
        if (big && (MMREC_BWD_NT & 1))
            hipLaunchKernelGGL(linear_bwd_w_bf16x3_kernel<true>, dim3(ncb, w.nsplit), dim3(512), 0, s, dY, X, part, dbpart, n, F, w.chunk);
        else
            hipLaunchKernelGGL(linear_bwd_w_bf16x3_kernel<false>, dim3(ncb, w.nsplit), dim3(512), 0, s, dY, X, part, dbpart, n, F, w.chunk);
#endif
        if (w.nsplit > 1 || db) {      // slabs -> dW and partials -> db, one launch
            const size_t elems = (size_t)64 * F;
            hipLaunchKernelGGL(bwd_w_finish_kernel, dim3((unsigned)((elems / 4 + 255) / 256) + 1), dim3(256), 0, s, (const float*)slabs,
                               w.nsplit, elems, dW, (const float*)dbp, n_dbpart, db);
        }
    }
    if (dX) {
        if (!wt_done) hipLaunchKernelGGL(bwd_wt_split_kernel, dim3(ceil_div(F, 256)), dim3(256), 0, s, W, F, Wt_sp, wcs_inv);
        const int rt = ceil_div(n, 128), nft = F / 128;
        int ftiles = 8;
        while (ftiles > 1 && (long)rt * ceil_div(nft, ftiles) < 192) ftiles >>= 1;
        if (big && BWD_X_NT_STORES)
            hipLaunchKernelGGL(bwd_x_f16x3_kernel<2>, dim3(rt, ceil_div(nft, ftiles)), dim3(320), 0, s, dY, (const float*)Wt_sp,
                               (const float*)wcs_inv, dX, n, F, ftiles);
        else
            hipLaunchKernelGGL(bwd_x_f16x3_kernel<0>, dim3(rt, ceil_div(nft, ftiles)), dim3(320), 0, s, dY, (const float*)Wt_sp,
                               (const float*)wcs_inv, dX, n, F, ftiles);
    }
    MMREC_RETURN_LAUNCH_STATUS();
}

// out = 64 j: the 64-column blocks of dY are handled one after the other (same workspace), each by
// the out = 64 kernel with dY's row stride = out.
extern "C" int mmrec_linear_bwd_w_f32(const float* dY, const float* X, float* dW, float* db,
                                      int32_t n, int32_t F, int32_t out, void* workspace,
                                      mmrec_stream_t stream) {
    if (out <= 0 || (out & 63) || F <= 0 || (F & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (!dW) return MMREC_ERR_BAD_ARG;
    hipStream_t s = mmrec_stream(stream);
    if (n == 0) {
        (void)hipMemsetAsync(dW, 0, (size_t)out * F * sizeof(float), s);
        if (db) (void)hipMemsetAsync(db, 0, out * sizeof(float), s);
        MMREC_RETURN_LAUNCH_STATUS();
    }
    if (!dY || !X) return MMREC_ERR_BAD_ARG;
    int nsplit, chunk;
    pick_split(ceil_div(F, BW_BF), n, BW_BK, &nsplit, &chunk);
    if (!workspace) return MMREC_ERR_BAD_ARG;
    float* part = static_cast<float*>(workspace);
    // workspace layout: per 64-output group z [nsplit > 1 ? nsplit*64*F : 0] dW partial slabs, then per z [nsplit*64] db partials.
    // ALL groups in one launch (grid.z; round 6: MMGCN's 256 / 384-wide layers sent 4 / 6 launches of 128-192 workgroups each)
    const int nz = out / 64;
    const size_t slab_z = nsplit > 1 ? (size_t)nsplit * 64 * F : 0;
    float* dbpart = db ? part + slab_z * nz : nullptr;
    const bool dma = (size_t)chunk * F * 4 < ((size_t)1 << 31) && (size_t)chunk * out * 4 < ((size_t)1 << 31) &&
                     !MMREC_GEMM_LEGACY_FWD;
    auto kern = dma ? linear_bwd_w_dma_kernel : linear_bwd_w_kernel;
    if (nsplit == 1) {
        hipLaunchKernelGGL(kern, dim3(ceil_div(F, BW_BF), 1, nz), dim3(256), 0, s, dY, X, dW, dbpart, n, F, chunk, out,
                           (size_t)64 * F, (size_t)nsplit * 64);
    } else {
        hipLaunchKernelGGL(kern, dim3(ceil_div(F, BW_BF), nsplit, nz), dim3(256), 0, s, dY, X, part, dbpart, n, F, chunk, out, slab_z,
                           (size_t)nsplit * 64);
        const size_t elems = (size_t)64 * F;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256), nz), dim3(256), 0,
                           s, part, nsplit, elems, (const float*)nullptr, dW);
    }
    if (db) hipLaunchKernelGGL(db_reduce_kernel, dim3(nz), dim3(1024), 0, s, dbpart, nsplit, db);
    MMREC_RETURN_LAUNCH_STATUS();
}

// C[M, :N] = A[M, K] B[N, K]^T (+ bias[N]): nn.Linear with any number of outputs (and, with B = W^T,
// its dX).  K % 32 == 0, ldc >= N; the 128 x 128 LDS-DMA GEMM of mfma_stream.h.
extern "C" int mmrec_gemm_nt_f32(const float* A, const float* B, const float* bias, float* C, int32_t M,
                                 int32_t N, int32_t K, int32_t ldc, mmrec_stream_t stream) {
    if (K <= 0 || (K & 31) || N <= 0 || ldc < N || ldc > (2 << 20)) return MMREC_ERR_UNSUPPORTED;
    if (M < 0) return MMREC_ERR_BAD_ARG;
    if (M == 0) return 0;
    if (!A || !B || !C) return MMREC_ERR_BAD_ARG;
    gemm_nt_launch(A, B, bias, C, M, N, K, ldc, N, mmrec_stream(stream));
    MMREC_RETURN_LAUNCH_STATUS();
}

extern "C" int mmrec_linear_bwd_x_f32(const float* dY, const float* W, float* dX, int32_t n,
                                      int32_t F, int32_t out, mmrec_stream_t stream) {
    if (out != 64 || F <= 0 || (F & 3)) return MMREC_ERR_UNSUPPORTED;
    if (n < 0) return MMREC_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!dY || !W || !dX) return MMREC_ERR_BAD_ARG;
    if (F % 128 == 0 && !MMREC_GEMM_LEGACY_FWD) {
        gemm64_stream_launch(dY, W, dX, n, F, mmrec_stream(stream));
    } else {
        hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(ceil_div(n, 128), ceil_div(F, 128)), dim3(256), 0,
                           mmrec_stream(stream), dY, W, dX, n, F);
    }
    MMREC_RETURN_LAUNCH_STATUS();
}
