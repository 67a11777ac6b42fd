// fp64 GEMM on v_mfma_f64_4x4x4_f64, producer/consumer form, 64×64 outputs per workgroup.
//
// v_mfma_f64_16x16x4_f64 tops out at ≈48 TFLOP/s on gfx950 whatever the tiling (profiles/r02_gemm_tile_sweep.txt); the
// 4×4×4 form (four independent 4×4×4 products per instruction) reaches 73 with its operands in registers (DESIGN.md
// §8).  gemm_skinny_pc_f64_kernel (gemm_f64_mfma.hpp) built a kernel around it for ONE 16×16 tile per wave and ended
// LDS-bound: four operand registers per four MFMAs.  Here a consumer wave owns 2×2 tiles, so eight operand registers
// feed sixteen MFMAs, half the LDS bytes per flop.
//
//   workgroup = 512 threads: waves 0-3 CONSUMERS (wave (wr, wc) the 32×32 block at rows 32 wr, columns 32 wc of the
//   64×64 tile: LDS fragment reads and MFMAs, nothing else), waves 4-7 PRODUCERS (global loads through a register
//   ring, LDS tile writes).  K in steps of 32; three LDS buffers (tile t multiplied, t+1 being read by the consumers, t+2
//   being written); one s_barrier per step.
//   LDS tile: entry [kp][x] is the PAIR (k, k+4) of one row (A) / column (B) x, kp = 4 h + kk for k = 8 h + kk,
//   kk = 0..3: one ds_read_b128 gives a lane its operands of the two MFMA steps 2h, 2h+1.
//   4×4×4 operand layout (measured, tools/experiments/mfma_f64_4x4x4_layout.hip): lane 16 g + 4 b + c holds A[k = g]
//   [row 4 b + c] resp. B[k = g][column 4 b + c] of block b; D[lane 16 i + 4 b + j] is row i, column j of block b's 4×4
//   product, a k-ascending fma chain from C.  A 16×16×4 step = four instructions on four accumulators with operand
//   variants u (A rows of block b^u) and v (B columns of block b^2v): the pairs (b^u, b^2v) are the tile's 16 blocks.
//
// Every output element is fma(A[r][K-1], B[K-1][c], … fma(A[r][0], B[0][c], 0)): the chain of gemm_rows_f64_kernel, bit
// for bit (tools/experiments/gemm_tile_sweep.hip checks it).  Same row-list and split-K conventions as that kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

typedef double pc64_d2 __attribute__((ext_vector_type(2)));
constexpr int PC64_TK = 32;          // k per step
constexpr int PC64_RING = 4;         // steps of global loads in flight per producer thread
constexpr size_t PC64_LDS_BYTES = 2 * 3 * 16 * 64 * sizeof(pc64_d2);   // As + Bs, three buffers: 96 KB

template <int RING>
__global__ __launch_bounds__(512, 1) void gemm_pc64_f64_kernel(const double* __restrict__ A, int lda,
                                                               const double* __restrict__ B, int ldb,
                                                               double* __restrict__ OUT, int ldo, int K, int nrows,
                                                               const int* __restrict__ row_list,
                                                               const int* __restrict__ row_count,
                                                               int kblk, size_t zstride) {
    const int count = row_list ? *row_count : nrows;
    const int row0 = blockIdx.y * 64;
    if (row0 >= count) return;
    const int col0 = blockIdx.x * 64;
    const int kbeg = kblk > 0 ? (int)blockIdx.z * kblk : 0;
    if (kblk > 0) {
        K = (K - kbeg) < kblk ? (K - kbeg) : kblk;
        OUT += (size_t)blockIdx.z * zstride;
    }
    const int nt = K / PC64_TK;
    extern __shared__ pc64_d2 pc64_lds[];
    pc64_d2* const As = pc64_lds;                    // [3][16][64]: As[buf][kp][row ^ 2 (kp >> 2)]
    pc64_d2* const Bs = pc64_lds + 3 * 16 * 64;      // [3][16][64]: Bs[buf][kp][col]

    if (threadIdx.x >= 256) {
        // ---------------- producers ----------------
        const int t = threadIdx.x - 256;
        // A: row a_row of the tile, the 8 consecutive k's of step pair a_h: four pairs (k, k+4), kk = 0..3
        const int a_row = t >> 2, a_h = t & 3;
        int a_grow = row0 + a_row;
        a_grow = a_grow < count ? a_grow : count - 1;          // clamped rows are computed and not stored
        if (row_list) a_grow = row_list[a_grow];
        const double* a_src = A + (size_t)a_grow * lda + kbeg + 8 * a_h;
        const int a_dst = (4 * a_h) * 64 + (a_row ^ (2 * a_h));
        // B: pair entry b_kp (rows k and k+4, k = 8 (b_kp >> 2) + (b_kp & 3)), columns b_c, b_c+16, b_c+32, b_c+48
        const int b_kp = t >> 4, b_c = t & 15;
        const int b_k = 8 * (b_kp >> 2) + (b_kp & 3);
        const double* b_src = B + (size_t)(kbeg + b_k) * ldb + col0 + b_c;
        const int b_dst = b_kp * 64 + b_c;
        double av[RING][8], bv[RING][8];
        auto issue = [&](int tile, double (&a)[8], double (&b)[8]) {
            const int tt = tile < nt ? tile : nt - 1;
            const double* ap = a_src + (size_t)tt * PC64_TK;
            const double* bp = b_src + (size_t)tt * PC64_TK * ldb;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = ap[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b[j] = bp[16 * j];
                b[4 + j] = bp[(size_t)4 * ldb + 16 * j];
            }
        };
        auto stage = [&](int buf, const double (&a)[8], const double (&b)[8]) {
            pc64_d2* as = As + buf * (16 * 64);
            pc64_d2* bs = Bs + buf * (16 * 64);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) as[a_dst + 64 * kk] = pc64_d2{a[kk], a[kk + 4]};
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[b_dst + 16 * j] = pc64_d2{b[j], b[4 + j]};
        };
        // steps 0 .. RING-1 requested; barrier k is passed once step k is staged (k < nt), plus the closing barrier nt
#pragma unroll
        for (int s = 0; s < RING; ++s) issue(s, av[s], bv[s]);
        int buf = 0;
        const int nfull = nt - nt % RING;
        for (int t0 = 0; t0 < nfull; t0 += RING) {
#pragma unroll
            for (int s = 0; s < RING; ++s) {
                stage(buf, av[s], bv[s]);
                buf = buf == 2 ? 0 : buf + 1;
#ifndef PC64_NO_LOADS
                issue(t0 + s + RING, av[s], bv[s]);
#endif
                __syncthreads();
            }
        }
#pragma unroll
        for (int s = 0; s < RING; ++s)
            if (nfull + s < nt) {                            // uniform
                stage(buf, av[s], bv[s]);
                buf = buf == 2 ? 0 : buf + 1;
                __syncthreads();
            }
        __syncthreads();                                     // closing barrier nt
        return;
    }
    // ---------------- consumers ----------------
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r16 = lane & 15, kk = lane >> 4;
    double acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.0;
    // fragment addresses: lane base (per variant) + compile-time offsets per (step pair, tile) — no address arithmetic in the loop
    const int a_base[2] = {kk * 64 + 32 * wr + r16, kk * 64 + 32 * wr + (r16 ^ 4)};
    const int b_base[2] = {kk * 64 + 32 * wc + r16, kk * 64 + 32 * wc + (r16 ^ 8)};
    pc64_d2 fa[2][2][2], fb[2][2][2];                        // [register set][tile i / j][variant u / v]
    auto fetch = [&](int set, int buf, int h) {              // step pair h of the step in LDS buffer buf
#ifdef PC64_NO_FETCH
        if (buf >= 0 && h >= 0) return;
#endif
        const pc64_d2* as = As + buf * (16 * 64) + (4 * h) * 64;
        const pc64_d2* bs = Bs + buf * (16 * 64) + (4 * h) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                fa[set][i][u] = as[(a_base[u] + 16 * i) ^ (2 * h)];
                fb[set][i][u] = bs[b_base[u] + 16 * i];
            }
    };
    auto mult = [&](int set) {
#ifdef PC64_NO_MFMA
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j][0] += fa[set][i][0][0] * fb[set][j][1][1] + fa[set][i][1][1] * fb[set][j][0][0];
        return;
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int uv = 0; uv < 4; ++uv)
                        acc[i][j][uv] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[set][i][uv >> 1][e], fb[set][j][uv & 1][e], acc[i][j][uv], 0, 0, 0);
    };
    // The consumer's barriers are bare s_barrier's: it never writes LDS, and its outstanding fragment reads need not
    // drain there.  The compiler-level fences keep the reads of a step behind the barrier that publishes it.
    auto handover = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#ifdef PC64_NO_CONSUMER_BARRIER
        return;
#endif
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // Rolling pipeline over the step pairs: the fragments of step pair s+1 are requested (eight ds_read_b128) before the
    // 32 MFMAs of step pair s are issued, into the other register set.  Never more than 16 reads in flight, so every
    // wait is an exact count (the LDS counter has 4 bits).
    handover();                                              // barrier 0: step 0 staged
    fetch(0, 0, 0);
    int buf = 0;
    for (int tile = 0; tile < nt; ++tile) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
        fetch(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mult(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, buf, 2);
        __builtin_amdgcn_sched_barrier(0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, buf, 3);
        __builtin_amdgcn_sched_barrier(0);
        mult(0);
        handover();                                          // barrier tile+1: the next step is staged (or the closing one)
        if (tile + 1 < nt) fetch(0, nbuf, 0);
        __builtin_amdgcn_sched_barrier(0);
        mult(1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
    // D layout: lane 16 g + 4 bb + bj holds row g, column bj of block bb; acc[i][j][2u+v] is block (bb^u, bb^2v) of tile (i, j)
    const int bb = (lane >> 2) & 3, bj = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lrow = row0 + 32 * wr + 16 * i + 4 * (bb ^ u) + kk;
            if (lrow < count) {
                const int grow = row_list ? row_list[lrow] : lrow;
                double* o = OUT + (size_t)grow * ldo + col0 + 32 * wc + bj;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int v = 0; v < 2; ++v) o[16 * j + 4 * (bb ^ (2 * v))] = acc[i][j][2 * u + v];
            }
        }
}

// OUT rows (all, or those of row_list) <- A rows · B; N a multiple of 64, K (and kblk) of 32.  kblk > 0: split-K into
// OUT + z zstride (see gemm_rows_f64_kernel).
inline void launch_gemm_pc64(const double* A, int lda, const double* B, int ldb, double* OUT, int ldo, int M, int K, int N,
                             const int* row_list, const int* row_count, int kblk, size_t zstride, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_pc64_f64_kernel<PC64_RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PC64_LDS_BYTES);
        attr_set = true;
    }
    const int nz = kblk > 0 ? (K + kblk - 1) / kblk : 1;
    dim3 grid(N / 64, (M + 63) / 64, nz);
    hipLaunchKernelGGL((gemm_pc64_f64_kernel<PC64_RING>), grid, dim3(512), PC64_LDS_BYTES, s, A, lda, B, ldb, OUT, ldo, K, M, row_list,
                       row_count, kblk, zstride);
}

}  // namespace dhmc
