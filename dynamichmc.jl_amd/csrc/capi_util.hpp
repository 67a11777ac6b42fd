// Shared by every translation unit of the C ABI (not part of the ABI: include/dhmc.h is): the standard headers, a device
// temporary with scope lifetime, and the interface of capi_rtc.hip (the caller's device functor compiled at run time).
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>


#include "../../include/dhmc.h"

// The value of `key` in the "key=value,key=value" list of an environment variable: DHMC_PK (the packed engine's launch layout) and
// DHMC_DENSE (the dense engines) — tuning and test switches, every one bit-neutral (include/dhmc.h "Environment").
inline bool env_list_value(const char* var, const char* key, long long* out) {
    const char* s = std::getenv(var);
    const size_t kl = std::strlen(key);
    while (s && *s) {
        const char* end = std::strchr(s, ',');
        const size_t len = end ? (size_t)(end - s) : std::strlen(s);
        if (len > kl + 1 && std::strncmp(s, key, kl) == 0 && s[kl] == '=') { *out = std::atoll(s + kl + 1); return true; }
        s = end ? end + 1 : nullptr;
    }
    return false;
}

// a device temporary that is released on every return path
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// ---- the caller's device functor, compiled at run time (capi_rtc.hip) ----------------------------------------------------
struct UserKernels {
    hipModule_t mod = nullptr;
    hipFunction_t run_lds = nullptr, run = nullptr, init = nullptr, search = nullptr, probe_traj = nullptr, probe_ratio = nullptr;
    hipFunction_t pipeline = nullptr;      // nuts_run_pipeline_kernel (chain widths of one or two slots per lane)
    int traits = 0;                        // dhmc_user_traits of the module: 1 kRecomputeGrad, 2 kBigDims, 4 PackedFunctor::kEligible
    hipFunction_t packed = nullptr;        // nuts_run_packed_kernel<PackedFunctor<T>, L, cpl>: a module of its own per (L, cpl), key (device, -(100 L + cpl))
    // DHMC_METRIC_DENSE: a second module, compiled when the first dense context of this functor is created
    hipModule_t dense_mod = nullptr;
    hipFunction_t k0 = nullptr, k2 = nullptr, k3 = nullptr, run_dense = nullptr, search_dense = nullptr, probe_traj_dense = nullptr,
                  probe_ratio_dense = nullptr;
    hipFunction_t eval = nullptr;      // 32 / 64 slots per lane (1024 < D <= 4096): functor_eval_kernel, the only kernel of `mod` then
};
struct UserTarget {
    std::string source, name;
    std::map<std::pair<int, int>, UserKernels> built;   // (device, slots per lane) -> module
};
extern std::vector<UserTarget> g_user_targets;
extern std::mutex g_user_mutex;
extern std::string g_rtc_log;
int rtc_compile(const std::string& source, const std::string& name, int npl, bool dense, std::vector<char>* code, std::vector<std::string>* lowered, bool fresh = false);
int rtc_load(const std::vector<char>& code, const std::vector<std::string>& low, hipModule_t* mod, std::initializer_list<hipFunction_t*> fns);
int npl_for_user_dim(int D);

