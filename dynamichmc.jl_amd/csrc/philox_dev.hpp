// Philox4x32-10 on the device and the dhmc stream convention (include/dhmc.h): key =
// (seed lo, global chain index), counter = (index, purpose, transition, seed hi).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dhmc {

enum : uint32_t { PURPOSE_MOMENTUM = 0, PURPOSE_DIRECTIONS = 1, PURPOSE_TREE = 2,
                  PURPOSE_SEARCH_MOMENTUM = 3, PURPOSE_INIT_POSITION = 4, PURPOSE_PROBE_MOMENTUM = 5 };

struct ChainKey {
    uint32_t k0, k1, seed_hi;
};

// The two 32 x 32 -> 64 bit products of a Philox round: the compiler emits ONE v_mad_u64_u32 for each (ROCm 7.2; older
// compilers chose v_mul_hi_u32 + v_mul_lo_u32 and rounds 2-3 wrote the instruction in asm — see csrc/detmath_dev.hpp for why no
// instruction is written in asm any more).
__device__ __forceinline__ void philox_round_products(uint32_t m0, uint32_t m1, uint32_t c0, uint32_t c2,
                                                      uint32_t& hi0, uint32_t& lo0, uint32_t& hi1, uint32_t& lo1) {
    const uint64_t e0 = (uint64_t)m0 * c0, e1 = (uint64_t)m1 * c2;
    hi0 = (uint32_t)(e0 >> 32); lo0 = (uint32_t)e0;
    hi1 = (uint32_t)(e1 >> 32); lo1 = (uint32_t)e1;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_round_products(0xD2511F53u, 0xCD9E8D57u, c0, c2, hi0, lo0, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void stream_raw64(const ChainKey& key, uint32_t index, uint32_t purpose,
                                              uint32_t transition, uint64_t& r1, uint64_t& r2) {
    uint32_t w[4];
    philox4x32_10(index, purpose, transition, key.seed_hi, key.k0, key.k1, w);
    r1 = ((uint64_t)w[1] << 32) | w[0];
    r2 = ((uint64_t)w[3] << 32) | w[2];
}

}  // namespace dhmc
