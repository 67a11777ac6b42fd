// Wavefront-level primitives for the one-wave-per-chain NUTS kernels (gfx950, wave64).
//
// Layout rule used everywhere: a D-vector of a chain is spread over the 64 lanes of the
// chain's wavefront, lane l owning coordinates l, l+64, l+128, ... (slot k <-> coordinate
// l + 64 k).  A global load/store of slot k is therefore one fully coalesced 512-byte
// access per wave instruction, and a dot product is NPL lane-local fma's followed by a
// 64-lane butterfly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

constexpr int WAVE = 64;

// Make a wave-uniform value provably uniform (SGPR) so branches on it are scalar branches.
__device__ __forceinline__ uint32_t uni_u32(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int uni_i32(int x) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)x); }
__device__ __forceinline__ double uni_f64(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// One DPP-moved copy of a double (two 32-bit DPP movs).  The permutations used are total (every lane reads a valid
// lane), so the destination needs no initial value: mov_dpp, not update_dpp with a zeroed `old` (two v_mov per copy).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// DPP move restricted to the rows of ROWMASK; the other rows are left UNDEFINED (wave_allreduce reads lane 63 only,
// and no value of a masked-out row reaches it).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_rows_f64(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, ROWMASK, 0xF, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)b, lane);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// All-reduce sums of N independent values over the 64 lanes, every lane (and the returned
// value, made scalar) getting the same bits.  The pairing is the xor butterfly 1,2,4,8,16,32:
// the ABI's summation order (include/dhmc.h; oracle/mathops.hpp wave_tree).  Steps 1-8 are
// DPP (quad_perm, row_half_mirror, row_mirror: equal to xor 1,2,4,8 because the lanes of
// each already-reduced group hold identical values); 16 and 32 are row_bcast15 / row_bcast31,
// which leave (r3 + r2) + (r1 + r0) in lane 63 — the same bits as (r0 + r1) + (r2 + r3) since
// IEEE addition commutes — read back as a scalar.  (After row_bcast15 only rows 1 and 3 are meaningful, after
// row_bcast31 only row 3: lane 63 = (v₃ + v₂) + lane 31, lane 31 = v₁ + v₀.)
// `nl` (16, 32 or 64; wave-uniform): lanes nl .. 63 are known to hold ±0 in every value — a chain of at most nl coordinates at
// one slot per lane, whose pad lanes contribute exact zeros to every partial sum.  Then whole rows of 16 lanes add nothing
// (x + ±0 = x bit for bit), the last one or two butterfly steps are skipped and the total is read from lane nl - 1: the same
// bits with 2 × 3 dependent instructions per value less on the critical path of a short chain (BASELINE config 4's 30-dim funnel).
__host__ __device__ constexpr int reduce_lanes(int NPL, int D) { return NPL > 1 ? 64 : (D <= 16 ? 16 : (D <= 32 ? 32 : 64)); }
template <int N>
__device__ __forceinline__ void wave_allreduce(double (&v)[N], int nl = 64) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0xB1>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x4E>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x141>(v[i]);  // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_f64<0x140>(v[i]);  // row_mirror
    if (nl > 16) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_rows_f64<0x142, 0xA>(v[i]);  // row_bcast15 -> rows 1,3
    }
    if (nl > 32) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = v[i] + dpp_rows_f64<0x143, 0xC>(v[i]);  // row_bcast31 -> rows 2,3
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = readlane_f64(v[i], nl - 1);  // (r3 + r2) + (r1 + r0), now scalar
}
__device__ __forceinline__ double wave_allreduce1(double x, int nl = 64) {
    double v[1] = {x};
    wave_allreduce<1>(v, nl);
    return v[0];
}

// ---- The same tree on a PERMUTED wave (round 6; the wide per-draw kernel of coordinate-wise targets) ------------------------
// The ABI pins the pairing over COORDINATES (partial sum l of a row pairs with l^1, then l^2, … l^32), not over hardware lanes.
// If hardware lane p holds partial sum  l(p) = p[5] | p[4] << 1 | p[3:0] << 2  (a row of 512 bytes is still one coalesced
// segment, read in another lane order), the tree's first two levels pair p with p^32 and p^16 — exactly what gfx950's
// v_permlane32_swap / v_permlane16_swap exchange between TWO registers.  So two values are reduced for the price of one at
// those levels (swap the halves / rows of x and y, add: the low half now holds x's pair sums, the high half y's), six values
// arrive at level 3 in two registers instead of six, and levels 3–6 stay inside a row of 16 lanes (quad_perm, quad_perm,
// row_half_mirror, row_mirror; no row_bcast).  Same pairs, same operands per addition (IEEE addition commutes): same bits as
// wave_allreduce on the unpermuted wave.  6 values: 39 vector instructions instead of 108; 2 values: 18 instead of 36.
__device__ __forceinline__ int xl_logical_lane(int p) { return ((p >> 5) & 1) | (((p >> 4) & 1) << 1) | ((p & 15) << 2); }

__device__ __forceinline__ void xl_words(double x, uint32_t& lo, uint32_t& hi) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    lo = (uint32_t)b; hi = (uint32_t)(b >> 32);
}
__device__ __forceinline__ double xl_f64(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); }
// x := [x lanes 0-31 | y lanes 0-31],  y := [x lanes 32-63 | y lanes 32-63]
__device__ __forceinline__ void xl_swap32(double& x, double& y) {
    uint32_t xl, xh, yl, yh;
    xl_words(x, xl, xh); xl_words(y, yl, yh);
    auto a = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
    x = xl_f64(a[0], b[0]); y = xl_f64(a[1], b[1]);
}
// rows of 16 lanes: x := [x0, y0, x2, y2],  y := [x1, y1, x3, y3]
__device__ __forceinline__ void xl_swap16(double& x, double& y) {
    uint32_t xl, xh, yl, yh;
    xl_words(x, xl, xh); xl_words(y, yl, yh);
    auto a = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
    x = xl_f64(a[0], b[0]); y = xl_f64(a[1], b[1]);
}
__device__ __forceinline__ double xl_any() {      // a register whose contents do not matter (the partner of an odd value out)
    uint32_t lo, hi;
    asm volatile("" : "=v"(lo));
    asm volatile("" : "=v"(hi));
    return xl_f64(lo, hi);
}
// N partial sums per lane -> (N + 3) / 4 registers; the total of value i = 4 j + t stands in EVERY lane of row XL_ROW[t] of
// register j (rows that belong to no value hold garbage)
__device__ constexpr int xl_row(int t) { return t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : 3; }
template <int N>
__device__ __forceinline__ void xl_reduce(const double (&v)[N], double (&u)[(N + 3) / 4]) {
    constexpr int NW = (N + 1) / 2, NU = (N + 3) / 4;
    double w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {                 // level 1: p ^ 32
        double x = v[2 * i], y = (2 * i + 1 < N) ? v[2 * i + 1] : xl_any();
        xl_swap32(x, y);
        w[i] = x + y;                              // [v_2i | v_2i+1]
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {                 // level 2: p ^ 16
        double x = w[2 * j], y = (2 * j + 1 < NW) ? w[2 * j + 1] : xl_any();
        xl_swap16(x, y);
        u[j] = x + y;                              // rows [v_4j, v_4j+2, v_4j+1, v_4j+3]
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = u[j] + dpp_f64<0xB1>(u[j]);    // levels 3-6 inside the rows
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = u[j] + dpp_f64<0x4E>(u[j]);
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = u[j] + dpp_f64<0x141>(u[j]);
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = u[j] + dpp_f64<0x140>(u[j]);
}
template <int N>
__device__ __forceinline__ double xl_value(const double (&u)[(N + 3) / 4], int i) { return readlane_f64(u[i / 4], 16 * xl_row(i % 4)); }
// any of the N totals < 0 (strict; false for NaN and -0, like the scalar comparison): two compares and a mask instead of 2 N
// v_readlane and N compares
template <int N>
__device__ __forceinline__ bool xl_any_negative(const double (&u)[(N + 3) / 4]) {
    constexpr int NU = (N + 3) / 4;
    uint64_t any = 0;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        uint64_t valid = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * j + t < N) valid |= 0xffffull << (16 * xl_row(t));
        any |= __ballot(u[j] < 0.0) & valid;
    }
    return any != 0;
}
template <int N>
__device__ __forceinline__ void wave_allreduce_xl(double (&v)[N]) {
    double u[(N + 3) / 4];
    xl_reduce<N>(v, u);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = xl_value<N>(u, i);
}

// the totals as scalars, by either reduction
template <int N, bool XL>
__device__ __forceinline__ void wave_totals(double (&v)[N], int nl) {
    if constexpr (XL) wave_allreduce_xl<N>(v);
    else wave_allreduce<N>(v, nl);
}
template <bool XL>
__device__ __forceinline__ double wave_total1(double x, int nl) {
    double v[1] = {x};
    wave_totals<1, XL>(v, nl);
    return v[0];
}

__device__ __forceinline__ bool wave_all(bool pred) { return __ballot(pred) == ~0ull; }

// The ABI's per-lane part of a dot product (include/dhmc.h "Summation order"): a padded row is cut into BLOCKS of
// 256 coordinates (4 slots per lane); inside a block lane l accumulates its slots k ascending with fma; the blocks'
// partial sums are then combined PER LANE by an adjacent-pairs binary tree, and the 64 lane values by the xor
// butterfly (wave_allreduce).  One block (D <= 256) is the plain fma chain.  The block structure is what lets a
// chain of 512+ coordinates be spread over several waves (one block per wave: tools/experiments/mw/nuts_mw_kernel.hpp,
// dense_rounds_k3b.hpp) without serialising the chain across them; a single wave holding the whole row keeps one
// accumulator per block instead (more independent fma chains, same bits).
template <int N, int NPL>
struct LaneAcc {
    static constexpr int NB = NPL >= 4 ? NPL / 4 : 1;
    static constexpr int LG = NB >= 16 ? 4 : NB >= 8 ? 3 : NB >= 4 ? 2 : NB >= 2 ? 1 : 0;
    // cur: the block being accumulated; t[l]: the folded sum of a completed, left-aligned group of 2^l blocks that still
    // waits for its right neighbour (a binary counter over the blocks: the adjacent-pairs tree, built as the blocks
    // complete, so that at most 1 + log2(NB) partial sums per dot are alive instead of NB)
    double cur[N];
    double t[N][LG + 1];
    __device__ __forceinline__ LaneAcc() {
#pragma unroll
        for (int n = 0; n < N; ++n) cur[n] = 0.0;
    }
    // slot k (compile-time after unrolling, ascending) of dot n
    __device__ __forceinline__ void add(int n, int k, double x, double y) {
        cur[n] = __builtin_fma(x, y, cur[n]);
        if (NB > 1 && (k % 4) == 3) {              // block k / 4 is complete
            const int b = k / 4;
            double v = cur[n];
            int l = 0;
#pragma unroll
            for (; l < LG; ++l) {
                if (((b >> l) & 1) == 0) break;
                v = t[n][l] + v;                    // left group + right group
            }
            t[n][l] = v;
            cur[n] = 0.0;
        }
    }
    __device__ __forceinline__ double fold(int n) const { return NB > 1 ? t[n][LG] : cur[n]; }
    __device__ __forceinline__ void fold_all(double (&out)[N]) const {
#pragma unroll
        for (int n = 0; n < N; ++n) out[n] = fold(n);
    }
};

// Coalesced vector access: slot k of lane l <-> element l + 64 k of a padded [Dpad] row.
template <int NPL>
__device__ __forceinline__ void ldv(const double* __restrict__ row, int lane, double (&v)[NPL]) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) v[k] = row[lane + WAVE * k];
}
template <int NPL>
__device__ __forceinline__ void stv(double* __restrict__ row, int lane, const double (&v)[NPL]) {
#pragma unroll
    for (int k = 0; k < NPL; ++k) row[lane + WAVE * k] = v[k];
}

// Slot k of a chain's D-vector as a functor: from an LDS / HBM row (lane-strided) or from the lane's registers.
struct LdsRow {
    const double* row;
    int lane;
    __device__ __forceinline__ double operator()(int k) const { return row[lane + WAVE * k]; }
};
template <int NPL>
struct RegRow {
    const double (&v)[NPL];
    __device__ __forceinline__ double operator()(int k) const { return v[k]; }
};

// A small array of wave-uniform scalars held in the LANES of one register: element i lives in lane i (i < 64).
// v_readlane / v_cndmask instead of an LDS round trip, and no LDS bytes.
struct LaneArrF64 {
    double v = 0.0;
    __device__ __forceinline__ double get(int i) const { return readlane_f64(v, i); }
    __device__ __forceinline__ void set(int i, double x, int lane) { v = (lane == i) ? x : v; }
};
struct LaneArrI64 {      // exact for any count (a float64 lane array would cost two conversions per access)
    uint32_t lo = 0, hi = 0;
    __device__ __forceinline__ int64_t get(int i) const {
        return (int64_t)(((uint64_t)__builtin_amdgcn_readlane(hi, i) << 32) | __builtin_amdgcn_readlane(lo, i));
    }
    __device__ __forceinline__ void set(int i, int64_t x, int lane) {
        lo = (lane == i) ? (uint32_t)(uint64_t)x : lo;
        hi = (lane == i) ? (uint32_t)((uint64_t)x >> 32) : hi;
    }
};
struct LaneArrI32 {
    int v = 0;
    __device__ __forceinline__ int get(int i) const { return (int)__builtin_amdgcn_readlane((uint32_t)v, i); }
    __device__ __forceinline__ void set(int i, int x, int lane) { v = (lane == i) ? x : v; }
};

// out = S v for symmetric S (rows streamed, coalesced): out_i = Σ_k fma(S[k][i], v_k, ·), k ascending.
template <int NPL>
__device__ __forceinline__ void sym_matvec(const double* __restrict__ S, int Dpad, int D, int lane,
                                           const double (&v)[NPL], double (&out)[NPL]) {
#pragma unroll
    for (int s = 0; s < NPL; ++s) out[s] = 0.0;
    // rows of S are fetched UB at a time (a lone wave otherwise waits out one L2 round trip per row); the fma chain of every
    // output stays k ascending
    constexpr int UB = NPL <= 2 ? 8 : (NPL <= 4 ? 4 : (NPL <= 8 ? 2 : 1));
#pragma unroll
    for (int s2 = 0; s2 < NPL; ++s2) {
        const int kcount = (D - WAVE * s2) < WAVE ? (D - WAVE * s2) : WAVE;
        int l2 = 0;
        if constexpr (UB > 1) {
            for (; l2 + UB <= kcount; l2 += UB) {
                double r[UB][NPL], vk[UB];
#pragma unroll
                for (int b = 0; b < UB; ++b) {
                    const double* __restrict__ rowk = S + (size_t)(WAVE * s2 + l2 + b) * Dpad;
#pragma unroll
                    for (int s = 0; s < NPL; ++s) r[b][s] = rowk[lane + WAVE * s];
                    vk[b] = readlane_f64(v[s2], l2 + b);
                }
#pragma unroll
                for (int b = 0; b < UB; ++b)
#pragma unroll
                    for (int s = 0; s < NPL; ++s) out[s] = __builtin_fma(r[b][s], vk[b], out[s]);
            }
        }
        for (; l2 < kcount; ++l2) {
            const double vk = readlane_f64(v[s2], l2);
            const double* __restrict__ rowk = S + (size_t)(WAVE * s2 + l2) * Dpad;
#pragma unroll
            for (int s = 0; s < NPL; ++s) out[s] = __builtin_fma(rowk[lane + WAVE * s], vk, out[s]);
        }
    }
}

}  // namespace dhmc
