// Plain-data parameter blocks of the per-draw kernels and the layout of a chain's HBM workspace: no HIP types, so that the
// packed small-D engine's body (packed_body.inc) can also be compiled by g++ for the CPU simulation the tests run it through
// (tests/hostsim/ — test infrastructure; the product library has no host execution path).
#pragma once
#include <stdint.h>
#include "../../include/dhmc.h"
#include "../../include/dhmc_detmath.h"

namespace dhmc {

struct TargetParams {
    const double* a;  // DIAG_NORMAL: mu      TRIDIAG: diag      LOGISTIC: X  [n][Dpad]     DENSE_NORMAL: mu        (padded, device)
    const double* b;  // DIAG_NORMAL: prec    TRIDIAG: off       LOGISTIC: Xᵀ [D][npad]    DENSE_NORMAL: P [Dpad][Dpad]
    const double* c;  //                                          LOGISTIC: y  [npad]
    int64_t n;        //                                          LOGISTIC: observations
    int64_t npad;     //                                          LOGISTIC: n rounded up to 64
    int32_t Dpad;
    int32_t pad_;
};

// DualAveragingState (stepsize.jl:121-127)
struct DAState {
    double mu;
    double Hbar;
    double logeps;
    double logeps_bar;
    int64_t m;
};

struct DeviceOutputs {
    double* draws;
    double* logdensities;
    double* eps;
    double* pi;
    double* acceptance_rate;
    int64_t* steps;
    int64_t* term_left;
    int64_t* term_right;
    int32_t* depth;
    uint32_t* directions;
};

struct ChainArrays {
    double* q;       // [C][Dpad]
    double* g;       // [C][Dpad]
    double* lq;      // [C]
    double* minv;    // [C][Dpad]  pads = 1
    double* W;       // [C][Dpad]  sqrt(1/minv), pads = 0
    double* eps;     // [C]
    DAState* da;     // [C]
    uint32_t* transition;  // [C]
    uint32_t* status;      // [C]
    double* ws;      // [C][nvec][Dpad] workspace
};

struct RunParams {
    int D, Dpad, C, chain_offset, max_depth, nvec;
    int l1_in_lds, chain_base;   // chain_base: first chain of this launch (round engines run half-batches)
    int k3_block;                // round engines: K3 as a workgroup per chain (dense_rounds_k3b.hpp) where it applies
    int one_product;             // dense round engine: one M⁻¹ product per leapfrog (dense_rounds.hpp; include/dhmc.h dhmc_set_dense_products)
    int fuse_k2;                 // … and K3b also takes the chain's next position update and density evaluation (K2's work) where the
                                 // family can be evaluated a block per wave (targets.hpp BlockEval)
    double min_delta;
    uint64_t seed;
    int64_t N;
    int64_t out_stride;          // record (chain, n) of the outputs is at chain * out_stride + n (0: out_stride = N): a call whose
                                 // host outputs leave in chunks writes every chunk into a staging buffer of the chunk's length
    ChainArrays st;
    int adapt, da_init, da_finalize, t0;
    double delta, gamma, kappa;
    DeviceOutputs out;
    TargetParams tp;
    unsigned long long* leapfrog_counter;  // total leapfrog steps of the launch (may be null)
    // an open metric window (include/dhmc.h dhmc_metric_window_begin): running mean and sum of squared deviations of every chain's
    // draws, [C][Dpad] each (null: no window), and the number of draws the window held before this launch
    double* win_mean;
    double* win_m2;
    int64_t win_n0;
    // The per-draw kernels walk all N transitions of a chain in one wave, so a launch ends with its slowest chain.  chain_work[chain]
    // := the leapfrog steps this launch spent on the chain (may be null); launch_order (may be null): workgroup b takes chain
    // launch_order[b] — the host sorts the chains by the previous launch's work, longest first (dhmc_capi.hip run_call), so that a
    // chain with persistently deeper trees starts in the first wave of workgroups instead of holding the last one open.
    unsigned* chain_work;
    const int* launch_order;
    // the packed small-D engine (packed_core.hpp): suspended levels 1 .. pk_lds_levels of a chain live in LDS (the rest in the HBM
    // workspace), and a transition starts only on trips of the wave's main loop that are multiples of pk_align (a power of two)
    int pk_lds_levels, pk_align;
    int pk_cpl;                  // coordinates per lane of the packed layout (2 or 4)
    int pk_order_base;           // the packed launch takes places pk_order_base .. C-1 of the launch order
    unsigned* pk_queue;          // null: lane group j of the launch runs place pk_order_base + j.  Else a device counter, initialised by the
                                 // host to the first place no group starts on: a group whose chain is done takes the next place from it
    int pk_max_waves;            // host side of the queue: the most waves a packed launch starts (0: one lane group per place, no queue)
    // A call continued by a later launch (the END GAME of a packed launch, below): prog[chain] = the transitions of this call the chain has
    // behind it (null: none, and N transitions per launch).  A launch takes every chain of its list from prog[chain] to N — N counted
    // from the call's start, like the records (outputs, window counts) — and writes prog back.  A chain the packed kernel gives up
    // goes to pk_evicted[(*pk_evict_count)++].  With prog, chain_work is added to instead of set.
    int* prog;
    int* pk_evicted;
    unsigned* pk_evict_count;
    // The end game of a packed launch: pk_live counts the lane groups that still have a chain (the host sets it to the groups the
    // launch starts; a group that finds no place left takes itself off).  Once it is at or below pk_handover_below (0: never),
    // every group gives its chain up at the next transition boundary (through pk_evicted), and the host finishes
    // those chains — the launch's deepest, by then — with the pipeline kernel at a third of the latency per leapfrog.
    unsigned* pk_live;
    int pk_handover_below;
};

// the pipeline kernel's LDS (nuts_pipeline_kernel.hpp; the host sizes launches and engine choices with it)
constexpr int PAIR_RING = 16;     // leaf records in flight (heavy cascades of B1 and B2 fall on the same leaves: a deeper ring lets the others run
                                  // on while one works through a long cascade)
constexpr int PIPE_NXL = 6;       // suspended levels 2 … 7 in LDS (first, last, ρ), deeper ones in the HBM workspace
// a chain's rows are 64·NPL doubles wide (NPL = 1, 2, 4: up to 256 coordinates)
DHMC_HD constexpr size_t pipeline_lds_bytes(int npl) {
    return sizeof(double) * ((size_t)64 * npl * (3 + 3 * PIPE_NXL) + (size_t)PAIR_RING * (2 * 64 * npl + 4) + PAIR_RING + (size_t)64 * npl + 4 + 8);   // mb_s[2..3]: B3's result
}

// workspace vector indices (units of Dpad doubles inside one chain's block)
DHMC_HD int ws_p0() { return 0; }
DHMC_HD int ws_edge(int dir, int which) { return 1 + 3 * dir + which; }  // which: 0 q, 1 p, 2 g
DHMC_HD int ws_rho_top() { return 7; }
DHMC_HD int ws_stack(int level, int which) { return 8 + 3 * level + which; }  // 0 first, 1 last, 2 rho
DHMC_HD int ws_slot(int max_depth, int s, int which) { return 8 + 3 * max_depth + 2 * s + which; }  // 0 q, 1 g
DHMC_HD int ws_nslots(int max_depth) { return max_depth + 3; }
DHMC_HD int ws_nvec(int max_depth) { return 8 + 3 * max_depth + 2 * ws_nslots(max_depth); }

}  // namespace dhmc
