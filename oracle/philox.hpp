// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link or call
// anything under oracle/.  See oracle/README.md.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11), restated independently of the device implementation, plus the dhmc stream
// convention of include/dhmc.h.  The reference draws from a caller-supplied AbstractRNG
// (src/NUTS.jl:232-233, src/trees.jl:23, src/NUTS.jl:44, src/hamiltonian.jl:124) and pins no
// stream; "parity unpinned" against Julia's Xoshiro/ziggurat streams, by construction.
#pragma once
#include <cstdint>
#include <array>

namespace oracle {

inline std::array<uint32_t, 4> philox4x32_10(std::array<uint32_t, 4> ctr,
                                             std::array<uint32_t, 2> key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)M0 * ctr[0];
        uint64_t p1 = (uint64_t)M1 * ctr[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        ctr = {hi1 ^ ctr[1] ^ key[0], lo1, hi0 ^ ctr[3] ^ key[1], lo0};
        key[0] += W0;
        key[1] += W1;
    }
    return ctr;
}

enum Purpose : uint32_t {
    PURPOSE_MOMENTUM = 0,
    PURPOSE_DIRECTIONS = 1,
    PURPOSE_TREE = 2,
    PURPOSE_SEARCH_MOMENTUM = 3,
    PURPOSE_INIT_POSITION = 4,
    PURPOSE_PROBE_MOMENTUM = 5,
};

// One chain's view of the stream: key = (seed lo, global chain index),
// counter = (index, purpose, transition, seed hi).
struct ChainStream {
    uint64_t seed;
    uint32_t chain;
    std::array<uint32_t, 4> raw(uint32_t index, uint32_t purpose, uint32_t transition) const {
        return philox4x32_10({index, purpose, transition, (uint32_t)(seed >> 32)},
                             {(uint32_t)seed, chain});
    }
    void raw64(uint32_t index, uint32_t purpose, uint32_t transition, uint64_t& r1,
               uint64_t& r2) const {
        auto w = raw(index, purpose, transition);
        r1 = ((uint64_t)w[1] << 32) | w[0];
        r2 = ((uint64_t)w[3] << 32) | w[2];
    }
};

}  // namespace oracle
