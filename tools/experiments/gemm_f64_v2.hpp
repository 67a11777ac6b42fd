// Experiment (round 3): the 64×64-per-wave shape of the vendor's fp64 GEMM (DESIGN §8 yardstick) with this library's contract —
// one k-ordered fma chain per output (v_mfma_f64_16x16x4_f64, K walked in ascending order, no split inside a block), gathered rows,
// optional split-K blocks.  Against gemm_rows_f64_kernel: FM×FN MFMA tiles per wave (4×4: 16 independent accumulators, 8 operand
// reads per 16 MFMAs), LDS double buffer with ONE barrier per K-tile, global loads two tiles ahead.
#pragma once
#include "../../dynamichmc.jl_amd/csrc/gemm_f64_mfma.hpp"

namespace dhmc {

template <int WM, int WN, int FM, int FN, int TK>
__global__ __launch_bounds__(64 * WM * WN) void gemm_rows_f64_v2_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                                       double* __restrict__ OUT, int ldo, int K, int nrows,
                                                                       const int* __restrict__ row_list, const int* __restrict__ row_count,
                                                                       int kblk = 0, size_t zstride = 0) {
    constexpr int TM = 16 * FM * WM, TN = 16 * FN * WN, NT = 64 * WM * WN;
    constexpr int LSA = TM + 16, LSB = TN + 16;
    const int count = row_list ? *row_count : nrows;
    const int row0 = blockIdx.y * TM;
    if (row0 >= count) return;
    const int col0 = blockIdx.x * TN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w / WN, wc = w % WN;

    extern __shared__ double lds[];
    double* As = lds;                        // [2][TK][LSA]
    double* Bs = lds + 2 * TK * LSA;         // [2][TK][LSB]

    // global -> LDS assignment.  A: TM rows × TK k, each thread PA consecutive k of one row; B: TK k × TN cols, PB consecutive cols
    constexpr int PA = (TM * TK) / NT, PB = (TK * TN) / NT;
    static_assert(PA >= 2 && PA % 2 == 0 && PB >= 2 && PB % 2 == 0, "16-byte loads");
    constexpr int A_TPR = TK / PA;           // threads per A row
    constexpr int B_TPR = TN / PB;           // threads per B row
    const int a_row = t / A_TPR, a_k = (t % A_TPR) * PA;
    int a_grow = row0 + a_row;
    a_grow = a_grow < count ? a_grow : count - 1;
    if (row_list) a_grow = row_list[a_grow];
    const int kbeg = kblk > 0 ? (int)blockIdx.z * kblk : 0;
    if (kblk > 0) {
        K = (K - kbeg) < kblk ? (K - kbeg) : kblk;
        OUT += (size_t)blockIdx.z * zstride;
    }
    const double* a_src = A + (size_t)a_grow * lda + kbeg + a_k;
    const int b_k = t / B_TPR, b_c = (t % B_TPR) * PB;
    const double* b_src = B + (size_t)(kbeg + b_k) * ldb + col0 + b_c;

    mfma_d4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};

    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 av[PA / 2], bv[PB / 2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PA / 2; ++i) av[i] = *reinterpret_cast<const d2*>(a_src + k0 + 2 * i);
#pragma unroll
        for (int i = 0; i < PB / 2; ++i) bv[i] = *reinterpret_cast<const d2*>(b_src + (size_t)k0 * ldb + 2 * i);
    };
    auto lstore = [&](int buf) {
        double* as = As + buf * TK * LSA;
        double* bs = Bs + buf * TK * LSB;
#pragma unroll
        for (int i = 0; i < PA / 2; ++i) {
            as[(a_k + 2 * i) * LSA + a_row] = av[i][0];
            as[(a_k + 2 * i + 1) * LSA + a_row] = av[i][1];
        }
#pragma unroll
        for (int i = 0; i < PB / 2; ++i) *reinterpret_cast<d2*>(bs + b_k * LSB + b_c + 2 * i) = bv[i];
    };

    const int ntiles = K / TK;
    gload(0);
    lstore(0);
    if (ntiles > 1) gload(TK);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) lstore(buf ^ 1);                  // tile+1 arrived during the previous iteration's MFMAs
        if (tile + 2 < ntiles) gload((tile + 2) * TK);
        const double* as = As + buf * TK * LSA + wr * (16 * FM) + (lane & 15);
        const double* bs = Bs + buf * TK * LSB + wc * (16 * FN) + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = as[kr * LSA + 16 * i];
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = bs[kr * LSB + 16 * j];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = row0 + wr * (16 * FM) + i * 16 + (lane >> 4) + 4 * r;
            if (lrow < count) {
                const int grow = row_list ? row_list[lrow] : lrow;
                double* o = OUT + (size_t)grow * ldo + col0 + wc * (16 * FN) + (lane & 15);
#pragma unroll
                for (int j = 0; j < FN; ++j) o[16 * j] = acc[i][j][r];
            }
        }
}

template <int WM, int WN, int FM, int FN, int TK>
constexpr size_t gemm_v2_lds_bytes() { return sizeof(double) * 2 * TK * ((16 * FM * WM + 16) + (16 * FN * WN + 16)); }


// To run the library's square products through a variant (config 3 under contention with K3b — measured: base 1.27e7, 64×64 1.18e7,
// 128×64 1.21e7, 128×128 0.93e7 with two parts / 1.15e7 with one): forward-declare launch_gemm_rows_v2 in gemm_f64_mfma.hpp, call it first
// thing in launch_gemm_rows, and build dhmc_capi.hip with -DDHMC_GEMM_ROWS_V2=1|2|3 -include this file (build_variant_capi.sh).
#ifdef DHMC_GEMM_ROWS_V2      // 1 = 64×64 (2×2 per wave), 2 = 128×64 (4×2), 3 = 128×128 (4×4)
template <int FM, int FN>
inline void launch_gemm_rows_v2_t(const double* A, const double* B, double* OUT, int ld, int nrows, const int* row_list, const int* row_count,
                                  hipStream_t s) {
    constexpr int TM = 32 * FM, TN = 32 * FN;
    constexpr size_t lds = gemm_v2_lds_bytes<2, 2, FM, FN, 16>();
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)gemm_rows_f64_v2_kernel<2, 2, FM, FN, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
    }();
    (void)once;
    dim3 grid(ld / TN, (nrows + TM - 1) / TM);
    hipLaunchKernelGGL((gemm_rows_f64_v2_kernel<2, 2, FM, FN, 16>), grid, dim3(256), lds, s, A, ld, B, ld, OUT, ld, ld, nrows, row_list, row_count, 0, (size_t)0);
}
inline void launch_gemm_rows_v2(const double* A, const double* B, double* OUT, int ld, int nrows, const int* row_list, const int* row_count,
                                hipStream_t s) {
#if DHMC_GEMM_ROWS_V2 == 1
    launch_gemm_rows_v2_t<2, 2>(A, B, OUT, ld, nrows, row_list, row_count, s);
#elif DHMC_GEMM_ROWS_V2 == 2
    if (ld % 64 == 0) launch_gemm_rows_v2_t<4, 2>(A, B, OUT, ld, nrows, row_list, row_count, s);
    else launch_gemm_rows_v2_t<2, 2>(A, B, OUT, ld, nrows, row_list, row_count, s);
#else
    if (ld % 128 == 0) launch_gemm_rows_v2_t<4, 4>(A, B, OUT, ld, nrows, row_list, row_count, s);
    else launch_gemm_rows_v2_t<2, 2>(A, B, OUT, ld, nrows, row_list, row_count, s);
#endif
}
#endif

}  // namespace dhmc
