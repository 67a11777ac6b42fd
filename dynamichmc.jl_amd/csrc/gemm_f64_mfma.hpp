// fp64 MFMA GEMM for the dense kinetic energy: OUT[r][:] = A[r][:] · B for a (possibly gathered) set
// of chain rows r, with A, OUT row-major [C][ld] and B row-major [ld][ld] (ld = Dpad).
//
// This is the p·M⁻¹ contraction of the reference's dense GaussianKineticEnergy
// (calculate_p♯ / ∇kinetic_energy, src/hamiltonian.jl:110,117, and rand_p's W·z, :124) for ALL chains
// at once: one chain is one row of A, so M⁻¹ (B) is streamed once per 64-row tile instead of once per
// chain.  v_mfma_f64_16x16x4_f64 is an exact k-ordered chain of fused multiply-adds (checked on gfx950,
// tools/experiments/mfma_f64_numerics.hip), and this kernel walks k in ascending order without split-K,
// so OUT[r][i] = fma(A[r][K-1], B[K-1][i], … fma(A[r][0], B[0][i], 0)) — bit for bit the oracle's chain.
//
// Tiling: 256-thread workgroup (4 waves) -> 64 rows × 64 columns; each wave a 32×32 block as 2×2 MFMA
// tiles (16 accumulator VGPR pairs); K advances 16 at a time through LDS (A tile stored k-major so both
// operand fragments are conflict-free ds_read_b64: row stride 80 doubles puts the four k-rows of one
// fragment on disjoint bank halves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_TM = 64, GEMM_TN = 64, GEMM_TK = 16, GEMM_LDS_STRIDE = 80;

// row_list == nullptr: rows 0..nrows-1.  Otherwise rows row_list[0..*row_count-1] (device-side count).
__global__ __launch_bounds__(256) void gemm_rows_f64_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            double* __restrict__ OUT, int ld, int K, int nrows,
                                                            const int* __restrict__ row_list,
                                                            const int* __restrict__ row_count) {
    const int count = row_list ? *row_count : nrows;
    const int row0 = blockIdx.y * GEMM_TM;
    if (row0 >= count) return;
    const int col0 = blockIdx.x * GEMM_TN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;

    __shared__ double As[GEMM_TK * GEMM_LDS_STRIDE];   // As[k][row]
    __shared__ double Bs[GEMM_TK * GEMM_LDS_STRIDE];   // Bs[k][col]

    // global -> LDS assignment
    const int a_row = t >> 2, a_k = (t & 3) * 4;        // 4 consecutive k of one row
    int a_grow = row0 + a_row;
    a_grow = a_grow < count ? a_grow : count - 1;       // clamp (results of clamped rows are not stored)
    if (row_list) a_grow = row_list[a_grow];
    const double* a_src = A + (size_t)a_grow * ld + a_k;
    const int b_k = t >> 4, b_c = (t & 15) * 4;         // 4 consecutive columns of one k
    const double* b_src = B + (size_t)b_k * ld + col0 + b_c;

    mfma_d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};

    // software pipeline: the global loads of K-tile t+1 are in flight while tile t is multiplied
    double2 a01 = *reinterpret_cast<const double2*>(a_src);
    double2 a23 = *reinterpret_cast<const double2*>(a_src + 2);
    double2 b01 = *reinterpret_cast<const double2*>(b_src);
    double2 b23 = *reinterpret_cast<const double2*>(b_src + 2);
    for (int k0 = 0; k0 < K; k0 += GEMM_TK) {
        __syncthreads();   // previous tile fully consumed
        As[(a_k + 0) * GEMM_LDS_STRIDE + a_row] = a01.x;
        As[(a_k + 1) * GEMM_LDS_STRIDE + a_row] = a01.y;
        As[(a_k + 2) * GEMM_LDS_STRIDE + a_row] = a23.x;
        As[(a_k + 3) * GEMM_LDS_STRIDE + a_row] = a23.y;
        *reinterpret_cast<double2*>(&Bs[b_k * GEMM_LDS_STRIDE + b_c]) = b01;
        *reinterpret_cast<double2*>(&Bs[b_k * GEMM_LDS_STRIDE + b_c + 2]) = b23;
        __syncthreads();
        if (k0 + GEMM_TK < K) {
            a01 = *reinterpret_cast<const double2*>(a_src + k0 + GEMM_TK);
            a23 = *reinterpret_cast<const double2*>(a_src + k0 + GEMM_TK + 2);
            b01 = *reinterpret_cast<const double2*>(b_src + (size_t)(k0 + GEMM_TK) * ld);
            b23 = *reinterpret_cast<const double2*>(b_src + (size_t)(k0 + GEMM_TK) * ld + 2);
        }
#pragma unroll
        for (int kk = 0; kk < GEMM_TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * GEMM_LDS_STRIDE;
            const double a0 = As[kr + wr * 32 + (lane & 15)];
            const double a1 = As[kr + wr * 32 + 16 + (lane & 15)];
            const double b0 = Bs[kr + wc * 32 + (lane & 15)];
            const double b1 = Bs[kr + wc * 32 + 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lrow = row0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r;
            if (lrow < count) {
                const int grow = row_list ? row_list[lrow] : lrow;
                double* o = OUT + (size_t)grow * ld + col0 + wc * 32 + (lane & 15);
                o[0] = acc[i][0][r];
                o[16] = acc[i][1][r];
            }
        }
}

// host launcher: OUT rows <- A rows · B  (K = ld = Dpad, a multiple of 64)
inline void launch_gemm_rows(const double* A, const double* B, double* OUT, int ld, int nrows, const int* row_list,
                             const int* row_count, hipStream_t s) {
    dim3 grid(ld / GEMM_TN, (nrows + GEMM_TM - 1) / GEMM_TM);
    hipLaunchKernelGGL(gemm_rows_f64_kernel, grid, dim3(256), 0, s, A, B, OUT, ld, ld, nrows, row_list, row_count);
}

}  // namespace dhmc
