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
// Tiling: 256-thread workgroup (4 waves) -> 64 rows × 64 columns (or one wave -> 32×32 for skinny
// products); each wave a 32×32 block as 2×2 MFMA
// tiles (16 accumulator VGPR pairs); K advances 16 at a time through LDS (A tile stored k-major so both
// operand fragments are conflict-free ds_read_b64: row stride 80 doubles puts the four k-rows of one
// fragment on disjoint bank halves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef DHMC_GEMM_BLK
#define DHMC_GEMM_BLK false
#endif

namespace dhmc {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));
typedef double gemm_d2s __attribute__((ext_vector_type(2)));

// one DPP-rotated copy of a double inside each row of 16 lanes (two 32-bit DPP movs; every lane has a source)
template <int CTRL>
__device__ __forceinline__ double gemm_dpp_f64(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

constexpr int GEMM_TK = 16, GEMM_LDS_STRIDE = 80;   // metric_dense_adapt.hpp's covariance kernel uses these

// OUT[r][0..N) = Σ_k A[r][k] · B[k][0..N), k = 0..K-1 ascending, for rows r = 0..nrows-1 or the gathered rows
// row_list[0..*row_count-1].  A is [*][lda], B is [K][ldb], OUT is [*][ldo]; K a multiple of TK, N a multiple of
// the column tile.  WT = waves per side, FR = 16×16 MFMA tiles per wave side: <2,2> -> 64×64 tile (4 waves,
// each 32×32); <2,1> -> 32×32 tile (4 waves, each one MFMA tile) for skinny products.
template <int WT, int FR, int TK, bool BLK = false>
__global__ __launch_bounds__(64 * WT * WT) void gemm_rows_f64_kernel(const double* __restrict__ A, int lda,
                                                                   const double* __restrict__ B, int ldb,
                                                                   double* __restrict__ OUT, int ldo, int K, int nrows,
                                                                   const int* __restrict__ row_list,
                                                                   const int* __restrict__ row_count,
                                                                   int kblk = 0, size_t zstride = 0) {
    constexpr int WS = 16 * FR;                       // rows/cols per wave
    constexpr int TM = WS * WT, TN = WS * WT, NT = 64 * WT * WT;
    constexpr int LS = TN + 16;                       // LDS row stride: the 4 k-rows of a fragment land on disjoint bank halves
    const int count = row_list ? *row_count : nrows;
    const int row0 = blockIdx.y * TM;
    if (row0 >= count) return;
    const int col0 = blockIdx.x * TN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w / WT, wc = w % WT;

    __shared__ __attribute__((aligned(16))) double As[TK * LS];   // As[k][row] (ROT: column (row + 8 (k >> 2)) mod TM)
    __shared__ __attribute__((aligned(16))) double Bs[TK * LS];   // Bs[k][col]

    // global -> LDS assignment: A tile TM×TK and B tile TK×TN, (TM*TK)/NT doubles per thread each
    constexpr int PER = (TM * TK) / NT;
    constexpr int A_TPR = TK / PER;                     // threads per A row
    const int a_row = t / A_TPR, a_k = (t % A_TPR) * PER;
    int a_grow = row0 + a_row;
    a_grow = a_grow < count ? a_grow : count - 1;       // clamp (clamped rows are not stored)
    if (row_list) a_grow = row_list[a_grow];
    // split-K (kblk > 0): workgroup z owns k in [z kblk, min(K, (z+1) kblk)) and writes its partial products to
    // OUT + z zstride; the caller adds the partial results in ascending z (an ABI order: include/dhmc.h, logistic Σ_n)
    const int kbeg = kblk > 0 ? (int)blockIdx.z * kblk : 0;
    if (kblk > 0) {
        K = (K - kbeg) < kblk ? (K - kbeg) : kblk;
        OUT += (size_t)blockIdx.z * zstride;
    }
    const double* a_src = A + (size_t)a_grow * lda + kbeg + a_k;
    // LDS column of the thread's A row.  A write instruction covers, per group of 16 lanes, 4 rows × the 4 k-groups a_k = 0, 4, 8, 12 — with
    // As[k][row] as it stands all four k-groups fall on the same banks (the k stride is a multiple of 64 dwords): a 4-way conflict on every
    // store of the transposition.  ROT rotates the columns of k-group g by 8 g (mod TM): the four groups land on four bank ranges, the
    // fragment reads of a step (one k-group) see the same uniform rotation.  (round 6: +4-6 % on the engines' shapes, same bits)
    constexpr bool ROT = !BLK && PER == 4 && TM == 64;
    const int a_col = ROT ? ((a_row + 2 * a_k) & (TM - 1)) : a_row;
    constexpr int B_TPR = TN / PER;                     // threads per B row (one k)
    const int b_k = t / B_TPR, b_c = (t % B_TPR) * PER;
    const double* b_src = B + (size_t)(kbeg + b_k) * ldb + col0 + b_c;

    // BLK: every 16×16×4 step is issued as four v_mfma_f64_4x4x4_f64 on four accumulators (operand pairing described at
    // gemm_skinny_f64_kernel below) — same bits, higher issue rate than v_mfma_f64_16x16x4_f64 on gfx950.
    mfma_d4 acc[FR][FR];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = mfma_d4{0.0, 0.0, 0.0, 0.0};

    // software pipeline: the global loads of K-tiles t+1 … t+PD are in flight while tile t is multiplied (a ring of PD register sets, the
    // k-loop unrolled by PD with no branch inside the unrolled body, so that the compiler's s_waitcnt placement keeps them in flight).
    // PD = 2 for the 64×64 tile: +6-8 % over one tile on the engines' shapes, most of it at the launch's ramp (profiles/r06_gemm_lds_rotation.txt §5)
#ifndef DHMC_GEMM_PD
#define DHMC_GEMM_PD 2
#endif
    constexpr int PD = (PER <= 4) ? DHMC_GEMM_PD : 1;
    const int nt = K / TK;
    double av[PD][PER], bv[PD][PER];
    auto issue = [&](int tile, double (&a)[PER], double (&b)[PER]) {
        const int tt = tile < nt ? tile : (nt > 0 ? nt - 1 : 0);   // past the end: the last tile once more (never staged)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            a[i] = a_src[tt * TK + i];
            b[i] = b_src[(size_t)(tt * TK) * ldb + i];
        }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) issue(s, av[s], bv[s]);
    auto step = [&](int tile, int s) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < PER; ++i) As[(a_k + i) * LS + a_col] = av[s][i];
        if constexpr (PER % 2 == 0) {                      // the B row's PER consecutive doubles as 16-byte stores
#pragma unroll
            for (int i = 0; i < PER; i += 2) *reinterpret_cast<gemm_d2s*>(&Bs[b_k * LS + b_c + i]) = gemm_d2s{bv[s][i], bv[s][i + 1]};
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) Bs[b_k * LS + b_c + i] = bv[s][i];
        }
        __syncthreads();
        issue(tile + PD, av[s], bv[s]);                    // (issued before the barrier instead: config 3 level, config 5 -0.7 % on one box)
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            const int kr = (kk + (lane >> 4)) * LS;
            if constexpr (!BLK) {
                double a[FR], b[FR];
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    const int ac = wr * WS + 16 * i + (lane & 15);
                    a[i] = As[kr + (ROT ? ((ac + 2 * kk) & (TM - 1)) : ac)];          // 8 (k >> 2) = 2 kk for the step's four k
                    b[i] = Bs[kr + wc * WS + 16 * i + (lane & 15)];
                }
#pragma unroll
                for (int i = 0; i < FR; ++i)
#pragma unroll
                    for (int j = 0; j < FR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
                // operands: u = 0 as read; A u = 1 holds row block b+2 (lanes rotated by 8 inside each row of 16),
                // B v = 1 holds column block b+1 (rotated by 4): the pairs (b+2u, b+v) are the 16 blocks of the tile.
                // One DPP rotation serves FR tiles, so it costs far less than a second LDS read.
                double a[FR][2], b[FR][2];
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    a[i][0] = As[kr + wr * WS + 16 * i + (lane & 15)];
                    b[i][0] = Bs[kr + wc * WS + 16 * i + (lane & 15)];
                }
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    a[i][1] = gemm_dpp_f64<0x128>(a[i][0]);   // row_ror:8
                    b[i][1] = gemm_dpp_f64<0x12C>(b[i][0]);   // row_ror:12: lane c reads lane (c + 4) & 15
                }
#pragma unroll
                for (int i = 0; i < FR; ++i)
#pragma unroll
                    for (int j = 0; j < FR; ++j)
#pragma unroll
                        for (int uv = 0; uv < 4; ++uv)
                            acc[i][j][uv] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][uv >> 1], b[j][uv & 1], acc[i][j][uv], 0, 0, 0);
            }
        }
    };
    const int nfull = nt - nt % PD;
    for (int t0 = 0; t0 < nfull; t0 += PD) {
#pragma unroll
        for (int s = 0; s < PD; ++s) step(t0 + s, s);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (nfull + s < nt) step(nfull + s, s);            // uniform
    if constexpr (!BLK) {
        // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = row0 + wr * WS + i * 16 + (lane >> 4) + 4 * r;
                if (lrow < count) {
                    const int grow = row_list ? row_list[lrow] : lrow;
                    double* o = OUT + (size_t)grow * ldo + col0 + wc * WS + (lane & 15);
#pragma unroll
                    for (int j = 0; j < FR; ++j) o[16 * j] = acc[i][j][r];
                }
            }
    } else {
        // acc[i][j][2u+v], lane 16 g + 4 bb + bj: row 4 ((bb+2u)&3) + g, column 4 ((bb+v)&3) + bj of the 16×16 tile (i, j)
        const int g = lane >> 4, bb = (lane >> 2) & 3, bj = lane & 3;
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int lrow = row0 + wr * WS + i * 16 + 4 * ((bb + 2 * u) & 3) + g;
                if (lrow < count) {
                    const int grow = row_list ? row_list[lrow] : lrow;
                    double* o = OUT + (size_t)grow * ldo + col0 + wc * WS + bj;
#pragma unroll
                    for (int j = 0; j < FR; ++j)
#pragma unroll
                        for (int v = 0; v < 2; ++v) o[16 * j + 4 * ((bb + v) & 3)] = acc[i][j][2 * u + v];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Skinny product with a very long K (the logistic gradient G = R·X: 1024 × 10⁵ × 256): only M·N/256 ≈ one 16×16
// MFMA tile per SIMD exists and each is one 10⁵-long k-ordered chain, so the kernel has to keep a single wave
// per SIMD fed.  One workgroup per CU = 32×32 outputs (4 waves × one MFMA tile), K in steps of 64:
//   * A (R[row][k], contiguous along k: transposed on the way) and B (X[k][col]) tiles go through double-buffered
//     LDS with XOR swizzles (conflict-free fragment reads, at most 2-way conflicts on the writes);
//   * three K-steps of 16-byte global loads are in flight ahead of the one being multiplied (register ring, loop
//     unrolled by 4, no branch inside the unrolled body so the s_waitcnt placement keeps them in flight);
//   * each 16×16×4 step is issued as four independent v_mfma_f64_4x4x4_f64 chains (below);
//   * workgroup id -> tile is XCD-aware: the N/32 column tiles of one row block run on the same XCD back to back,
//     so a row block of A is fetched from HBM once and re-read from that XCD's L2.
// ------------------------------------------------------------------------------------------------------------
constexpr int SK_TK = 64;

// v_mfma_f64_4x4x4_f64 (layout and order measured on gfx950, tools/experiments/mfma_f64_4x4x4_layout.hip): with
// lane L = 16 g + 4 b + c, the instruction computes four independent 4×4×4 products, block b = 0..3:
//     D[lane 16 i + 4 b + j] = fma chain over k = 0,1,2,3 (ascending, from C) of A[lane 16 k + 4 b + i] · B[lane 16 k + 4 b + j].
// A 16×16×4 step of a wave's tile (4×4 blocks of 4×4) is issued as FOUR such instructions on four independent
// accumulators: A operands u = 0,1 hold row block b^u in position b, B operands v = 0,1 hold column block
// b^(2v); the 16 (row block, column block) pairs (b^u, b^2v) are all distinct, so acc[u][v] owns block pair
// (b^u, b^2v) for the whole K loop — the same ascending k chain per output element as v_mfma_f64_16x16x4_f64,
// but a single wave issues the four independent chains ~2.8× faster than the one dependent 16x16x4 chain
// (tools/experiments/mfma_f64_chain.hip, mfma_f64_blocks_rate.hip: 73 vs 26 TFLOP/s at one wave per SIMD).
// The operand "permutations" are just LDS read addresses (no DPP: lane rotations cost more than the MFMAs).
template <int P>
__global__ __launch_bounds__(256, 1) void gemm_skinny_f64_kernel(const double* __restrict__ A, int lda,
                                                                 const double* __restrict__ B, int ldb,
                                                                 double* __restrict__ OUT, int ldo, int K, int M, int N) {
    static_assert(P % 2 == 0, "the LDS double buffer is indexed by the ring slot");
    const int ncol = N / 32, nrb = (M + 31) / 32;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rb = xcd + 8 * (j / ncol), cb = j % ncol;
    if (rb >= nrb) return;
    const int row0 = rb * 32, col0 = cb * 32;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r16 = lane & 15, kk = lane >> 4;

    __shared__ double As[2][SK_TK * 32];   // As[buf][k][row ^ 2 (k >> 3)]
    __shared__ double Bs[2][SK_TK * 32];   // Bs[buf][k][col ^ 2 (k & 3)]

    // A: thread -> row a_row, the 8 consecutive k starting at 8 a_c (64 contiguous bytes)
    const int a_row = t >> 3, a_c = t & 7;
    int a_grow = row0 + a_row;
    a_grow = a_grow < M ? a_grow : M - 1;
    const double* a_src = A + (size_t)a_grow * lda + 8 * a_c;
    const int a_dst = (8 * a_c) * 32 + (a_row ^ (2 * a_c));
    // B: thread -> k row b_k, the 8 consecutive columns starting at 8 b_c
    const int b_k = t >> 2, b_c = t & 3;
    const double* b_src = B + (size_t)b_k * ldb + col0 + 8 * b_c;
    const int b_dst = b_k * 32 + 8 * b_c, b_x = 2 * (b_k & 3);
    // fragment reads
    const int a_rd = 16 * wr + r16;                     // row; swizzle 2 (i >> 1) applied per step
    const int b_rd = kk * 32 + ((16 * wc + r16) ^ (2 * kk));
    const int nt = K / SK_TK;

    double av[P][8], bv[P][8];
    auto issue = [&](int tile, double (&a)[8], double (&b)[8]) {
        const int tt = tile < nt ? tile : nt - 1;
        const double* ap = a_src + (size_t)tt * SK_TK;
        const double* bp = b_src + (size_t)tt * SK_TK * ldb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    };
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < P - 1; ++s) issue(s, av[s], bv[s]);

    // one K-step: refill the ring slot consumed in the previous step, stage tile `tile` through LDS, multiply
    auto step = [&](int tile, int s) {
#ifndef DHMC_SK_NO_LOADS
        issue(tile + P - 1, av[(s + P - 1) % P], bv[(s + P - 1) % P]);
#endif
        double* as = As[s & 1];
        double* bs = Bs[s & 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            as[a_dst + i * 32] = av[s][i];
            bs[b_dst + (i ^ b_x)] = bv[s][i];
        }
        __syncthreads();
        double af[2][16], bf[2][16];                   // all operands of the tile first: one LDS latency per tile
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ar = (4 * i + kk) * 32, sw = 2 * (i >> 1);
            af[0][i] = as[ar + (a_rd ^ sw)];
            af[1][i] = as[ar + (a_rd ^ sw ^ 4)];
            bf[0][i] = bs[(4 * i) * 32 + b_rd];
            bf[1][i] = bs[(4 * i) * 32 + (b_rd ^ 8)];
        }
#ifndef DHMC_SK_NO_MFMA
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0][i], bf[0][i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0][i], bf[1][i], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1][i], bf[0][i], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1][i], bf[1][i], acc[3], 0, 0, 0);
        }
#else
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] += af[0][i] * bf[0][i] + af[1][i] * bf[1][i];
#endif
    };
    // Full groups of P steps run without any branch inside the unrolled body, so the compiler's s_waitcnt
    // placement keeps P-1 K-steps of loads in flight (a conditional body made it drain the queue every P steps).
    const int nfull = nt - nt % P;
    for (int t0 = 0; t0 < nfull; t0 += P) {
#pragma unroll
        for (int s = 0; s < P; ++s) step(t0 + s, s);
    }
#pragma unroll
    for (int s = 0; s < P; ++s)
        if (nfull + s < nt) step(nfull + s, s);        // uniform
    // acc[2u+v], lane 16 i + 4 b + j: row 4 (b^u) + i, column 4 (b^2v) + j of the wave's 16×16 tile
    const int bb = (lane >> 2) & 3, bj = lane & 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int grow = row0 + 16 * wr + 4 * (bb ^ u) + kk;
        if (grow < M) {
#pragma unroll
            for (int v = 0; v < 2; ++v) OUT[(size_t)grow * ldo + col0 + 16 * wc + 4 * (bb ^ (2 * v)) + bj] = acc[2 * u + v];
        }
    }
}

// Producer/consumer form of the same kernel: 512 threads = per SIMD one CONSUMER wave (LDS fragment reads + the
// four MFMA chains of its 16×16 tile, nothing else in its instruction stream) and one PRODUCER wave (the global
// loads, the register ring and the LDS tile writes).  One s_barrier per 64-k step hands tile t+1 over while tile t
// is being multiplied, so the load/transposition phase of the single-role kernel no longer sits in front of every
// MFMA phase.  Same tiles, same LDS layout, same bits.
typedef double gemm_d2 __attribute__((ext_vector_type(2)));
constexpr int PC_TK = 32;      // k per step of the producer/consumer kernel

template <int P>
__global__ __launch_bounds__(512, 1) void gemm_skinny_pc_f64_kernel(const double* __restrict__ A, int lda,
                                                                    const double* __restrict__ B, int ldb,
                                                                    double* __restrict__ OUT, int ldo, int K, int M, int N) {
    const int ncol = N / 32, nrb = (M + 31) / 32;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rb = xcd + 8 * (j / ncol), cb = j % ncol;
    if (rb >= nrb) return;
    const int row0 = rb * 32, col0 = cb * 32;
    const int nt = K / PC_TK;
    // LDS tiles hold PAIRS (k, k+4): entry kp = 4 h + kk (h = pair of MFMA steps 2h, 2h+1; kk = k & 3) carries the
    // operands of both steps of one lane, so a fragment read is one ds_read_b128.  Three buffers: tile t is being
    // multiplied from registers, t+1 is being read into the other register set, t+2 is being written.
    __shared__ gemm_d2 As[3][16 * 32];   // As[buf][kp][row ^ 2 h]
    __shared__ gemm_d2 Bs[3][16 * 32];   // Bs[buf][kp][col ^ 2 kk]

    if (threadIdx.x >= 256) {
        // ---------------- producer ----------------
        const int t = threadIdx.x - 256;
        // A: row a_row; group of 8 k's a_h (= step pair), elements m = 2 a_m, 2 a_m + 1 and their partners m + 4
        const int a_row = t >> 3, a_h = (t >> 1) & 3, a_m = t & 1;
        int a_grow = row0 + a_row;
        a_grow = a_grow < M ? a_grow : M - 1;
        const double* a_src = A + (size_t)a_grow * lda + 8 * a_h + 2 * a_m;
        const int a_dst = (4 * a_h + 2 * a_m) * 32 + (a_row ^ (2 * a_h));
        // B: pair entry b_kp (rows k and k+4, k = 8 (b_kp >> 2) + (b_kp & 3)), the 2 consecutive columns from 2 b_c
        const int b_kp = t >> 4, b_c = t & 15;
        const int b_k = 8 * (b_kp >> 2) + (b_kp & 3);
        const double* b_src = B + (size_t)b_k * ldb + col0 + 2 * b_c;
        const int b_dst = b_kp * 32, b_x = 2 * (b_kp & 3);
        double av[P][4], bv[P][4];
        auto issue = [&](int tile, double (&a)[4], double (&b)[4]) {
            const int tt = tile < nt ? tile : nt - 1;
            const double* ap = a_src + (size_t)tt * PC_TK;
            const double* bp = b_src + (size_t)tt * PC_TK * ldb;
            a[0] = ap[0]; a[1] = ap[1]; a[2] = ap[4]; a[3] = ap[5];
            b[0] = bp[0]; b[1] = bp[1]; b[2] = bp[(size_t)4 * ldb]; b[3] = bp[(size_t)4 * ldb + 1];
        };
        auto stage = [&](int buf, const double (&a)[4], const double (&b)[4]) {
            gemm_d2* as = As[buf];
            gemm_d2* bs = Bs[buf];
            as[a_dst] = gemm_d2{a[0], a[2]};
            as[a_dst + 32] = gemm_d2{a[1], a[3]};
            bs[b_dst + ((2 * b_c) ^ b_x)] = gemm_d2{b[0], b[2]};
            bs[b_dst + ((2 * b_c + 1) ^ b_x)] = gemm_d2{b[1], b[3]};
        };
        // tiles 0 .. P-1 requested; barrier k is passed once tile k is staged (k < nt), plus the closing barrier nt
#pragma unroll
        for (int s = 0; s < P; ++s) issue(s, av[s], bv[s]);
        int buf = 0;
        const int nfull = nt - nt % P;
        for (int t0 = 0; t0 < nfull; t0 += P) {
#pragma unroll
            for (int s = 0; s < P; ++s) {
                stage(buf, av[s], bv[s]);
                buf = buf == 2 ? 0 : buf + 1;
#ifndef DHMC_SK_NO_LOADS
                issue(t0 + s + P, av[s], bv[s]);
#endif
                __syncthreads();
            }
        }
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (nfull + s < nt) {                            // uniform
                stage(buf, av[s], bv[s]);
                buf = buf == 2 ? 0 : buf + 1;
                __syncthreads();
            }
        __syncthreads();                                     // closing barrier nt
        return;
    }
    // ---------------- consumer ----------------
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int r16 = lane & 15, kk = lane >> 4;
    const int a_rd = kk * 32 + 16 * wr + r16;               // + 128 per step pair h, row swizzle 2 h applied there
    const int b_rd = kk * 32 + ((16 * wc + r16) ^ (2 * kk));
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    gemm_d2 fa[2][2][4], fb[2][2][4];                        // [register set][variant][step pair]
    auto fetch = [&](int buf, gemm_d2 (&af)[2][4], gemm_d2 (&bf)[2][4], int h0, int h1) {
        const gemm_d2* as = As[buf];
        const gemm_d2* bs = Bs[buf];
#pragma unroll
        for (int h = h0; h < h1; ++h) {
            af[0][h] = as[(a_rd + 128 * h) ^ (2 * h)];
            af[1][h] = as[(a_rd + 128 * h) ^ (2 * h) ^ 4];
            bf[0][h] = bs[128 * h + b_rd];
            bf[1][h] = bs[128 * h + (b_rd ^ 8)];
        }
    };
    auto mult = [&](const gemm_d2 (&af)[2][4], const gemm_d2 (&bf)[2][4], int h0, int h1) {
#ifdef DHMC_SK_NO_MFMA
        for (int h = h0; h < h1; ++h) acc[h & 3] += af[0][h][0] * bf[0][h][1] + af[1][h][1] * bf[1][h][0];
        return;
#endif
#pragma unroll
        for (int h = h0; h < h1; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0][h][e], bf[0][h][e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[0][h][e], bf[1][h][e], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1][h][e], bf[0][h][e], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[1][h][e], bf[1][h][e], acc[3], 0, 0, 0);
            }
    };
    // one tile: multiply the fragments read during the previous tile while reading the next tile's, in two halves of
    // 8 reads (never more than 15 LDS reads outstanding: the lgkmcnt counter has 4 bits, and waiting for "the old
    // reads" must not mean waiting for new ones)
    auto tile_step = [&](int nbuf, bool more, gemm_d2 (&caf)[2][4], gemm_d2 (&cbf)[2][4], gemm_d2 (&naf)[2][4], gemm_d2 (&nbf)[2][4]) {
        if (more) fetch(nbuf, naf, nbf, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mult(caf, cbf, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (more) fetch(nbuf, naf, nbf, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
        mult(caf, cbf, 2, 4);
    };
    // The consumer's barriers are bare s_barrier's: it never writes LDS, and its own outstanding fragment reads need
    // not drain there (a __syncthreads would wait for them and serialise the read-ahead with the MFMAs).  The
    // compiler-level fences keep the reads of a tile behind the barrier that publishes it, and the phases in order.
    auto handover = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    handover();                                              // barrier 0: tile 0 staged
    fetch(0, fa[0], fb[0], 0, 4);
    int buf = 1;
    int tile = 0;
    for (; tile + 1 < nt; tile += 2) {
        handover();                                          // barrier tile+1: the next tile is staged
        tile_step(buf, true, fa[0], fb[0], fa[1], fb[1]);
        buf = buf == 2 ? 0 : buf + 1;
        handover();
        tile_step(buf, tile + 2 < nt, fa[1], fb[1], fa[0], fb[0]);
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (tile < nt) {                                         // odd tile count: the last tile is in set 0
        handover();
        tile_step(buf, false, fa[0], fb[0], fa[1], fb[1]);
    }
    const int bb = (lane >> 2) & 3, bj = lane & 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int grow = row0 + 16 * wr + 4 * (bb ^ u) + kk;
        if (grow < M) {
#pragma unroll
            for (int v = 0; v < 2; ++v) OUT[(size_t)grow * ldo + col0 + 16 * wc + 4 * (bb ^ (2 * v)) + bj] = acc[2 * u + v];
        }
    }
}

// host launchers.  Square metric products: OUT rows <- A rows · B with K = N = ld = Dpad.
// k per LDS stage of the 64×64-tile kernel (K must be a multiple; Dpad is one of 64): DHMC_GEMM_TK at build time for experiments
#ifndef DHMC_GEMM_TK
#define DHMC_GEMM_TK 16
#endif
constexpr int GEMM_ROWS_TK = DHMC_GEMM_TK;
inline void launch_gemm_rows(const double* A, const double* B, double* OUT, int ld, int nrows, const int* row_list,
                             const int* row_count, hipStream_t s) {
    dim3 grid(ld / 64, (nrows + 63) / 64);
    hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, GEMM_ROWS_TK, DHMC_GEMM_BLK>), grid, dim3(256), 0, s, A, ld, B, ld, OUT, ld, ld, nrows, row_list, row_count);
}
// Split-K product over gathered rows: P[z][r][0..N) = Σ_{k in block z} A[r][k] · B[k][0..N) for the rows r of row_list
// (all rows 0..M-1 when row_list is null), z = 0 .. ceil(K / kblk) - 1; P is [nz][zstride] with rows of ldo doubles.
inline void launch_gemm_splitk(const double* A, int lda, const double* B, int ldb, double* P, int ldo, size_t zstride, int M,
                               int K, int N, int kblk, const int* row_list, const int* row_count, hipStream_t s) {
    const int nz = (K + kblk - 1) / kblk;
    dim3 grid(N / 64, (M + 63) / 64, nz);
    hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, GEMM_ROWS_TK, DHMC_GEMM_BLK>), grid, dim3(256), 0, s, A, lda, B, ldb, P, ldo, K, M, row_list,
                       row_count, kblk, zstride);
}
// OUT[r][0..N) = A[r][0..K) · B for the rows r of row_list (64×64 tiles)
inline void launch_gemm_list(const double* A, int lda, const double* B, int ldb, double* OUT, int ldo, int M, int K, int N,
                             const int* row_list, const int* row_count, hipStream_t s) {
    dim3 grid(N / 64, (M + 63) / 64);
    hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, GEMM_ROWS_TK, DHMC_GEMM_BLK>), grid, dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, row_list,
                       row_count, 0, (size_t)0);
}
// General product OUT[M][N] = A[M][K] · B[K][N] (N a multiple of 32, K of 16).  64×64 tiles (4 waves × 32×32)
// when that grid fills the chip; otherwise 32×32 tiles worked by 4 waves of one 16×16 MFMA tile each, so a
// skinny product (e.g. R·X: 1024 × 10⁵ × 256) still puts a wave on every SIMD.
inline void launch_gemm(const double* A, int lda, const double* B, int ldb, double* OUT, int ldo, int M, int K, int N,
                        hipStream_t s) {
    const long tiles64 = (long)((M + 63) / 64) * (N / 64);
    if (N % 64 == 0 && tiles64 >= 512) {
        dim3 grid(N / 64, (M + 63) / 64);
        hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 2, GEMM_ROWS_TK, DHMC_GEMM_BLK>), grid, dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, nullptr, nullptr);
    } else if (K % SK_TK == 0 && K >= 64 * SK_TK) {   // (SK_TK is a multiple of PC_TK)
        const int ncol = N / 32, nrb = (M + 31) / 32;
#ifdef DHMC_SK_SINGLE_ROLE
        hipLaunchKernelGGL((gemm_skinny_f64_kernel<4>), dim3(8 * ncol * ((nrb + 7) / 8)), dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, N);
#else
        hipLaunchKernelGGL((gemm_skinny_pc_f64_kernel<8>), dim3(8 * ncol * ((nrb + 7) / 8)), dim3(512), 0, s, A, lda, B, ldb, OUT, ldo, K, M, N);
#endif
    } else {
        dim3 grid(N / 32, (M + 31) / 32);
        hipLaunchKernelGGL((gemm_rows_f64_kernel<2, 1, 64>), grid, dim3(256), 0, s, A, lda, B, ldb, OUT, ldo, K, M, nullptr, nullptr);
    }
}

}  // namespace dhmc
