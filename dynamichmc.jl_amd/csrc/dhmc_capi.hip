// libdhmc_amd.so — host side of the C ABI declared in include/dhmc.h.
//
// Owns the opaque context (device-resident chain state + workspace), validates arguments where
// the reference uses @argcheck, launches the HIP kernels of nuts_kernels.hpp on the caller's
// stream, and maps per-chain failure words onto return codes.  There is NO CPU path in this
// library: without a HIP device every entry point that computes returns DHMC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <hipcub/hipcub.hpp>
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
#include "dense_factor.hpp"
#include "dense_rounds_k3b.hpp"
#include "external_rounds.hpp"
#include "ess_kernels.hpp"
#include "logistic_rounds.hpp"
#include "metric_dense_adapt.hpp"
#include "treestat_kernels.hpp"
#include "launch.hpp"
#include "util_kernels.hpp"

using namespace dhmc;

// ---- the caller's device functor, compiled at run time (include/dhmc.h dhmc_register_target_source) -----------------------
#include "gen/rtc_headers.inc"     // const char dhmc_rtc_headers[]: the kernel headers as one string (make_rtc_source.py)
namespace {
struct UserKernels {
    hipModule_t mod = nullptr;
    hipFunction_t run_lds = nullptr, run = nullptr, init = nullptr, search = nullptr, probe_traj = nullptr, probe_ratio = nullptr;
    // DHMC_METRIC_DENSE: a second module, compiled when the first dense context of this functor is created
    hipModule_t dense_mod = nullptr;
    hipFunction_t k0 = nullptr, k2 = nullptr, k3 = nullptr, run_dense = nullptr, search_dense = nullptr, probe_traj_dense = nullptr,
                  probe_ratio_dense = nullptr;
};
struct UserTarget {
    std::string source, name;
    std::map<std::pair<int, int>, UserKernels> built;   // (device, slots per lane) -> module
};
std::vector<UserTarget> g_user_targets;
std::mutex g_user_mutex;
std::string g_rtc_log;

// hiprtc has the HIP device runtime built in but no system headers: the fixed-width integer names the headers use
const char* rtc_prelude() {
    return "typedef unsigned char uint8_t; typedef unsigned int uint32_t; typedef int int32_t;\n"
           "typedef unsigned long long uint64_t; typedef long long int64_t;\n";
}
// the kernels a functor needs, as name expressions: the wave-per-chain set of the diagonal metric, or the dense metric's
// (round engine K0/K2/K3, wave-per-chain run and search, the two probes)
uint64_t rtc_checksum(const char* p, uint64_t n) {      // FNV-1a over the code object
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}
std::vector<std::string> rtc_kernel_names(const std::string& name, int npl, bool dense) {
    const std::string T = "dhmc::" + name, N = std::to_string(npl);
    if (!dense)
        return {"dhmc::nuts_run_kernel<" + T + ", " + N + ", true>", "dhmc::nuts_run_kernel<" + T + ", " + N + ", false>",
                "dhmc::init_kernel<" + T + ", " + N + ">", "dhmc::stepsize_search_kernel<" + T + ", " + N + ">",
                "dhmc::probe_kernel<" + T + ", " + N + ", false, 0>", "dhmc::probe_kernel<" + T + ", " + N + ", false, 1>"};
    return {"dhmc::rounds_k0_kernel<" + T + ", " + N + ">", "dhmc::rounds_k2_kernel<" + T + ", " + N + ">",
            (npl >= 8 ? "dhmc::rounds_k3b_kernel<" : "dhmc::rounds_k3_kernel<") + T + ", " + N + ">",
            "dhmc::nuts_run_dense_kernel<" + T + ", " + N + ">", "dhmc::stepsize_search_dense_kernel<" + T + ", " + N + ">",
            "dhmc::probe_kernel<" + T + ", " + N + ", true, 0>", "dhmc::probe_kernel<" + T + ", " + N + ", true, 1>"};
}
// compile `source` (which defines dhmc::`name`) with the kernel templates for one chain width; *code receives the code object
int rtc_compile(const std::string& source, const std::string& name, int npl, bool dense, std::vector<char>* code, std::vector<std::string>* lowered) {
    std::string src = rtc_prelude();
    src += dhmc_rtc_headers;
    src += "\n// ---- the caller's functor -------------------------------------------------------------\n";
    src += source;
    src += "\n";
    const std::vector<std::string> exprs = rtc_kernel_names(name, npl, dense);
    // DHMC_RTC_CACHE=<directory>: code objects are kept there, keyed by everything that went into them, so that the next
    // process (a new Julia session) loads instead of compiling (≈ 2 s diagonal, ≈ 15 s dense per functor and chain width)
    std::string cache_file;
    if (const char* dir = std::getenv("DHMC_RTC_CACHE"); dir && *dir && code && lowered) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const std::string& t) { for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        mix(src); mix(dhmc_version()); mix(std::to_string(major) + "." + std::to_string(minor));
        for (const auto& e : exprs) mix(e);
        char hex[17];
        std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)h);
        cache_file = std::string(dir) + "/dhmc_rtc_" + hex + ".co";
        if (FILE* f = std::fopen(cache_file.c_str(), "rb")) {
            bool ok = false;
            char magic[8];
            uint32_t n = 0;
            std::vector<std::string> names;
            if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "DHMCRTC2", 8) == 0 && std::fread(&n, 4, 1, f) == 1 && n == exprs.size()) {
                ok = true;
                for (uint32_t i = 0; i < n && ok; ++i) {
                    uint32_t len = 0;
                    ok = std::fread(&len, 4, 1, f) == 1 && len < 4096;
                    std::string t(ok ? len : 0, '\0');
                    ok = ok && (len == 0 || std::fread(&t[0], 1, len, f) == len);
                    names.push_back(t);
                }
                uint64_t cs = 0, sum = 0;
                ok = ok && std::fread(&cs, 8, 1, f) == 1 && cs > 0 && cs < ((uint64_t)1 << 31);
                if (ok) { code->resize(cs); ok = std::fread(code->data(), 1, cs, f) == cs; }
                ok = ok && std::fread(&sum, 8, 1, f) == 1 && sum == rtc_checksum(code->data(), cs);      // a damaged payload is not trusted
            }
            std::fclose(f);
            if (ok) { *lowered = names; g_rtc_log = "(loaded from " + cache_file + ")"; return DHMC_OK; }
            code->clear();
        }
    }
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "dhmc_user_target.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return DHMC_ERR_HIP;
    for (const auto& e : exprs) (void)hiprtcAddNameExpression(prog, e.c_str());
    // the architecture of the device the context lives on (this library's own kernels are built for gfx950; a functor follows
    // whatever device it will run beside them on)
    std::string arch = "--offload-arch=gfx950";
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0])
            arch = std::string("--offload-arch=") + prop.gcnArchName;
    }
    const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
    size_t ls = 0;
    (void)hiprtcGetProgramLogSize(prog, &ls);
    g_rtc_log.assign(ls, '\0');
    if (ls) (void)hiprtcGetProgramLog(prog, &g_rtc_log[0]);
    int rc = DHMC_OK;
    if (r != HIPRTC_SUCCESS) {
        rc = DHMC_ERR_INVALID_ARGUMENT;
    } else {
        if (lowered)
            for (const auto& e : exprs) {
                const char* low = nullptr;
                if (hiprtcGetLoweredName(prog, e.c_str(), &low) != HIPRTC_SUCCESS || !low) { rc = DHMC_ERR_HIP; break; }
                lowered->push_back(low);
            }
        if (rc == DHMC_OK && code) {
            size_t cs = 0;
            if (hiprtcGetCodeSize(prog, &cs) != HIPRTC_SUCCESS) rc = DHMC_ERR_HIP;
            else { code->resize(cs); if (hiprtcGetCode(prog, code->data()) != HIPRTC_SUCCESS) rc = DHMC_ERR_HIP; }
        }
    }
    (void)hiprtcDestroyProgram(&prog);
    if (rc == DHMC_OK && !cache_file.empty()) {           // written under another name first: a concurrent reader never sees half a file
        static std::atomic<unsigned> serial{0};              // unique per process (pid) and per call: concurrent ranks never share a tmp file
        const std::string tmp = cache_file + ".tmp" + std::to_string((long long)getpid()) + "_" + std::to_string(serial.fetch_add(1));
        if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
            const uint32_t n = (uint32_t)lowered->size();
            bool ok = std::fwrite("DHMCRTC2", 1, 8, f) == 8 && std::fwrite(&n, 4, 1, f) == 1;
            for (const auto& t : *lowered) {
                const uint32_t len = (uint32_t)t.size();
                ok = ok && std::fwrite(&len, 4, 1, f) == 1 && std::fwrite(t.data(), 1, len, f) == len;
            }
            const uint64_t cs = code->size(), sum = rtc_checksum(code->data(), cs);
            ok = ok && std::fwrite(&cs, 8, 1, f) == 1 && std::fwrite(code->data(), 1, cs, f) == cs && std::fwrite(&sum, 8, 1, f) == 1;
            ok = (std::fclose(f) == 0) && ok;
            if (!ok || std::rename(tmp.c_str(), cache_file.c_str()) != 0) (void)std::remove(tmp.c_str());
        }
    }
    return rc;
}
// load a compiled module and look its kernels up in the order of rtc_kernel_names
// (the module handle and the functions are published only when every kernel resolved; otherwise the module is unloaded and
// *mod stays null, so that a later context compiles again instead of finding a module without kernels)
int rtc_load(const std::vector<char>& code, const std::vector<std::string>& low, hipModule_t* mod, std::initializer_list<hipFunction_t*> fns) {
    hipModule_t m = nullptr;
    if (hipModuleLoadData(&m, code.data()) != hipSuccess) return DHMC_ERR_HIP;
    std::vector<hipFunction_t> got;
    size_t i = 0;
    for (size_t k = 0; k < fns.size(); ++k) {
        hipFunction_t f = nullptr;
        if (i >= low.size() || hipModuleGetFunction(&f, m, low[i++].c_str()) != hipSuccess || !f) {
            (void)hipModuleUnload(m);
            return DHMC_ERR_HIP;
        }
        got.push_back(f);
    }
    i = 0;
    for (hipFunction_t* f : fns) *f = got[i++];
    *mod = m;
    return DHMC_OK;
}
int npl_for_user_dim(int D) { return D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : D <= 512 ? 8 : D <= 1024 ? 16 : 0; }
}  // namespace

struct dhmc_ctx {
    dhmc_config cfg{};
    int Dpad = 0, NPL = 0, nvec = 0;
    hipStream_t stream = nullptr;
    ChainArrays st{};
    TargetParams tp{};
    void* d_tp_a = nullptr;
    void* d_tp_b = nullptr;
    unsigned long long* d_counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    unsigned long long last_leapfrogs = 0;
    int l1_in_lds = 1;
    int k3_block = 1;
    DenseMetric dm{};          // DHMC_METRIC_DENSE only
    double* d_Minv = nullptr;
    double* d_WT = nullptr;
    double* d_fwork = nullptr;   // 4 × Dpad² doubles: work space of the device factorisation (dense_factor.hpp)
    int* d_fflags = nullptr;     // [2]: non-finite input, not positive definite
    RoundBuffers rb{};         // round-based dense engine (dense_rounds.hpp)
    RoundBuffers rbp[4]{};     // dense round engine: the batch is run as up to 4 parts on as many streams; every part has its own
                               // list and counters, the vectors are shared
    hipStream_t streams[4] = {};
    int dense_parts = 2;       // DHMC_DENSE_PARTS
    int dense_row_lists = 1;   // DHMC_DENSE_ROW_LISTS: products over the running chains only once some have finished
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_joins[4] = {};
    int dense_rounds = 1;
    int fuse_k2 = 1;           // DHMC_FUSE_K2=0: K3b leaves the next position update / density evaluation to K2 (dense_rounds_k3b.hpp)
    int dense_products = 2;    // dhmc_set_dense_products: 2 = the reference's recurrence; 1 = one M⁻¹ product per leapfrog (either dense engine)
    int per_chain_dense = 0;   // cfg.dense_per_chain: every chain has its own M⁻¹ / Wᵀ ([C][Dpad][Dpad]); wave-per-chain kernels only
    int use_graph = 0;         // dense round engine: capture four rounds into a hipGraph (DHMC_GRAPH=1; measured slower, see dhmc_run)
    int logistic_rounds = 0;   // GEMM-gradient round engine for DHMC_TARGET_LOGISTIC with a diagonal metric
    int logistic_batched = 0;  // … and with it (or beyond 1024 coefficients) ℓ, ∇ℓ of all chains by the same GEMMs wherever they are needed
                               // outside a round: initialisation, step-size search, the Diagnostics probes (external_eval)
    LogisticRound lr{};
    int external = 0;          // DHMC_TARGET_EXTERNAL: density from the host's callback, round engine always
    int builtin_big = 0;       // a built-in family whose density the LIBRARY evaluates for all chains between kernels, where an external model's
                               // callback stands (more than 1024 coordinates; the logistic regression with a dense metric): the same engine
    int* d_all_rows = nullptr; // logistic_batched: the row list 0..C-1 and its length, for the GEMMs of the batched gradient
    double* d_big[2] = {};     // DHMC_TARGET_DENSE_NORMAL beyond 1024 coordinates: q − μ and P(q − μ) of all chains ([C][Dpad] each)
    dhmc_logdensity_fn ext_fn = nullptr;
    void* ext_user = nullptr;
    ExtSearchState* d_ss = nullptr;
    uint32_t* d_sflags = nullptr;   // [C][4]: ℓ(q′) (a double) and the position flag between two search kernels (dense)
    unsigned long long last_rounds = 0;
    uint64_t ws_bytes = 0;
    // host outputs of dhmc_run: persistent device staging (two buffers per field, grown on demand — no hipMalloc per call),
    // and a copy stream
    struct StageBuf { void* p = nullptr; size_t cap = 0; };
    StageBuf stage[2][10];
    hipStream_t copy_stream = nullptr;
    dhmc_allreduce_fn metric_allreduce = nullptr;   // dhmc_set_metric_allreduce: the shared dense metric adapted from the draws of all ranks
    void* metric_allreduce_user = nullptr;
    hipEvent_t ev_k0[2] = {}, ev_k1[2] = {}, ev_copy[2] = {};
    int* h_done = nullptr;     // page-locked [2][8]: the dense round engine's done-counters, read without draining the streams
    hipEvent_t ev_done[2] = {};
    int64_t host_chunk = 0;    // DHMC_HOST_CHUNK: transitions per chunk of a call with host outputs (0: ≈1 GiB of draws per chunk)
    const UserKernels* user = nullptr;   // target >= DHMC_TARGET_USER_BASE: the run-time compiled kernels of the caller's functor
    void* d_user_params = nullptr;
    bool poisoned = false;     // an external callback failed in the middle of dhmc_run: (q, ℓq, ∇ℓ) are inconsistent until dhmc_init / dhmc_import_state
    std::string err;
    std::vector<void*> allocs;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return DHMC_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define DHMC_CHECK_USABLE(ctx)                                                                                     \
    do {                                                                                                          \
        if ((ctx)->poisoned) {                                                                                    \
            (ctx)->err = "the context's chain state is inconsistent after a failed log-density callback: call dhmc_init or dhmc_import_state"; \
            return DHMC_ERR_CALLBACK;                                                                             \
        }                                                                                                         \
    } while (0)

template <class Tp>
int dev_alloc(dhmc_ctx* c, Tp** p, size_t count) {
    void* v = nullptr;
    HIP_TRY(c, hipMalloc(&v, count * sizeof(Tp)));
    c->allocs.push_back(v);
    c->ws_bytes += count * sizeof(Tp);
    *p = (Tp*)v;
    return DHMC_OK;
}

// slots per lane: the register/LDS-resident kernels go up to 16 (D <= 1024); the streaming round-engine kernels
// of an external model up to 64 (D <= 4096)
int npl_for_dim(int D, bool big) {
    int npl = (D + WAVE - 1) / WAVE;
    for (int cand : {1, 2, 4, 8, 16, 32, 64})
        if (npl <= cand) return (cand <= 16 || big) ? cand : 0;
    return 0;
}


void launch_logistic_op(int which, int npl, const RoundArgs& a, const LogisticRound& L, hipStream_t s) {
    const dim3 g(a.P.C), b(WAVE);
#define DHMC_NPL_SWITCH(KERNEL, ...)                                                              \
    switch (npl) {                                                                                \
    case 1: hipLaunchKernelGGL((KERNEL<1>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 2: hipLaunchKernelGGL((KERNEL<2>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 4: hipLaunchKernelGGL((KERNEL<4>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 8: hipLaunchKernelGGL((KERNEL<8>), g, b, 0, s, __VA_ARGS__); break;                      \
    case 16: hipLaunchKernelGGL((KERNEL<16>), g, b, 0, s, __VA_ARGS__); break;                    \
    case 32: hipLaunchKernelGGL((KERNEL<32>), g, b, 0, s, __VA_ARGS__); break;                    \
    default: hipLaunchKernelGGL((KERNEL<64>), g, b, 0, s, __VA_ARGS__); break;                    \
    }
    switch (which) {
    case 0: DHMC_NPL_SWITCH(rounds_momentum_diag_kernel, a.P, a.R) break;
    case 1: DHMC_NPL_SWITCH(rounds_k1_diag_kernel, a.P, a.R) break;
    case 2:
        hipLaunchKernelGGL(logistic_link_kernel, dim3((unsigned)L.nz, a.P.C), b, 0, s, a.P, a.R, L);
        break;
    case 4:
        (void)hipMemsetAsync(L.act_count, 0, sizeof(int), s);
        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((a.P.C + 255) / 256), dim3(256), 0, s, a.P, a.R, L);
        break;
    default: DHMC_NPL_SWITCH(rounds_k2_logistic_kernel, a.P, a.R, L) break;
    }
#undef DHMC_NPL_SWITCH
}

int dispatch(const dhmc_ctx* c, Op op, const void* P, hipStream_t stream_override = nullptr, bool use_override = false) {
    hipStream_t cs = use_override ? stream_override : c->stream;
    const DenseMetric* M = c->cfg.metric == DHMC_METRIC_DENSE ? &c->dm : nullptr;
    if (c->user) {     // the caller's functor: the same kernels, from the run-time compiled modules
        const UserKernels& U = *c->user;
        // the grid is the op's own chain count: the dense round engine runs half-batches (RoundArgs::P.C chains from P.chain_base)
        // on two streams, and the kernels index chain_base + blockIdx.x without a bounds guard
        unsigned grid = (unsigned)c->cfg.chains;
        auto launch = [&](hipFunction_t f, unsigned block, unsigned lds, void** args) {
            return f && grid > 0 && hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, cs, args, nullptr) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
        };
        DenseMetric dm = M ? *M : DenseMetric{};
        switch (op) {
        case Op::Run: {
            const RunParams& R = *(const RunParams*)P;
            grid = (unsigned)R.C;
            void* args[] = {const_cast<void*>(P), &dm};
            if (M) return launch(U.run_dense, WAVE, (unsigned)lds_bytes_dense(), args);
            return launch(R.l1_in_lds ? U.run_lds : U.run, WAVE,
                          (unsigned)(R.l1_in_lds ? lds_bytes(R.Dpad, true, lds_extra_levels(c->NPL)) : lds_bytes(R.Dpad, false, 0)), args);
        }
        case Op::Init: { grid = (unsigned)((const InitParams*)P)->C; void* args[] = {const_cast<void*>(P)}; return launch(U.init, WAVE, 0, args); }
        case Op::Search: {
            grid = (unsigned)((const SearchParams*)P)->C;
            void* args[] = {const_cast<void*>(P), &dm};
            return M ? launch(U.search_dense, WAVE, 0, args) : launch(U.search, WAVE, (unsigned)(sizeof(double) * c->Dpad), args);
        }
        case Op::ProbeTrajectory: case Op::ProbeRatios: {     // Diagnostics.leapfrog_trajectory / explore_log_acceptance_ratios
            grid = (unsigned)((const ProbeParams*)P)->C;
            void* args[] = {const_cast<void*>(P), &dm};
            const bool traj = op == Op::ProbeTrajectory;
            if (M) return launch(traj ? U.probe_traj_dense : U.probe_ratio_dense, WAVE, 0, args);
            return launch(traj ? U.probe_traj : U.probe_ratio, WAVE, (unsigned)(sizeof(double) * c->Dpad), args);
        }
        case Op::RoundStart: return dispatch_family<StdNormalT>(c->NPL, op, P, cs, M);      // no density in this kernel
        case Op::RoundK0: case Op::RoundK2: case Op::RoundK3: {
            RoundArgs a = *(const RoundArgs*)P;
            grid = (unsigned)a.P.C;
            void* args[] = {&a.P, &a.R};
            if (op == Op::RoundK0) return launch(U.k0, WAVE, 0, args);
            if (op == Op::RoundK2) return launch(U.k2, WAVE, 0, args);
            return launch(U.k3, c->NPL >= 8 ? WAVE * k3b_waves(c->NPL) : WAVE, 0, args);   // a workgroup per chain from 512 coordinates
        }
        default: return DHMC_ERR_UNSUPPORTED;
        }
    }
    if (c->builtin_big) return dispatch_family<ExternalT>(c->NPL, op, P, cs, M);
    switch (c->cfg.target) {
    case DHMC_TARGET_STD_NORMAL: return dispatch_family<StdNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_DIAG_NORMAL: return dispatch_family<DiagNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_TRIDIAG_NORMAL: return dispatch_family<TridiagNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_FUNNEL: return dispatch_family<FunnelT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_LOGISTIC: return dispatch_family<LogisticT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_DENSE_NORMAL: return dispatch_family<DenseNormalT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_ALWAYS_DIVERGENT: return dispatch_family<AlwaysDivergentT>(c->NPL, op, P, cs, M);
    case DHMC_TARGET_EXTERNAL: return dispatch_family<ExternalT>(c->NPL, op, P, cs, M);
    default: return DHMC_ERR_UNSUPPORTED;
    }
}

// upload S (symmetric M⁻¹) and Wᵀ padded to [Dpad][Dpad]
int upload_dense_metric(dhmc_ctx* c, const std::vector<double>& S, const std::vector<double>& W) {
    const int D = c->cfg.dim;
    const size_t Dp = c->Dpad;
    std::vector<double> a(Dp * Dp, 0.0), b(Dp * Dp, 0.0);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            a[(size_t)i * Dp + j] = S[(size_t)i * D + j];
            b[(size_t)j * Dp + i] = W[(size_t)i * D + j];   // transpose: WT[k][i] = W[i][k]
        }
    const size_t nmat = c->per_chain_dense ? (size_t)c->cfg.chains : 1;
    for (size_t m = 0; m < nmat; ++m) {
        HIP_TRY(c, hipMemcpyAsync(c->d_Minv + m * Dp * Dp, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_WT + m * Dp * Dp, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
// κ := GaussianKineticEnergy(Symmetric(src)) (hamiltonian.jl:73) entirely on the device (dense_factor.hpp): src is a
// device matrix with row stride lsrc whose upper triangle is read.  The context's metric is replaced only if src is
// finite and positive definite; otherwise DHMC_ERR_INVALID_ARGUMENT (the reference's cholesky throws).
int device_dense_metric(dhmc_ctx* c, const double* src, int lsrc, int slot = -1) {   // slot: a chain of a per-chain dense context, -1: all
    const int D = c->cfg.dim, ld = c->Dpad;
    const size_t n = (size_t)ld * ld;
    double* Stmp = c->d_fwork + 3 * n;
    double* WTtmp = c->d_fwork + n;              // the X buffer: free again once M = XᵀX exists
    int flags[2] = {0, 0};
    HIP_TRY(c, hipMemsetAsync(c->d_fflags, 0, 2 * sizeof(int), c->stream));
    hipLaunchKernelGGL(df_symmetrize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, lsrc, D, Stmp, ld, c->d_fflags);
    df_dense_metric(Stmp, Stmp, WTtmp, D, ld, c->d_fwork, c->d_fflags, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(flags, c->d_fflags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (flags[0] || flags[1]) return DHMC_ERR_INVALID_ARGUMENT;
    const size_t nmat = c->per_chain_dense ? (size_t)c->cfg.chains : 1;
    for (size_t m = 0; m < nmat; ++m) {
        if (slot >= 0 && (size_t)slot != m) continue;
        HIP_TRY(c, hipMemcpyAsync(c->d_Minv + m * n, Stmp, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_WT + m * n, WTtmp, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
void launch_metric(const dhmc_ctx* c, const double* draws, int64_t N) {
    int D = c->cfg.dim, Dp = c->Dpad, C = c->cfg.chains;
    switch (c->NPL) {
    case 1: hipLaunchKernelGGL((metric_diag_kernel<1>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 2: hipLaunchKernelGGL((metric_diag_kernel<2>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 4: hipLaunchKernelGGL((metric_diag_kernel<4>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 8: hipLaunchKernelGGL((metric_diag_kernel<8>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 16: hipLaunchKernelGGL((metric_diag_kernel<16>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    case 32: hipLaunchKernelGGL((metric_diag_kernel<32>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    default: hipLaunchKernelGGL((metric_diag_kernel<64>), dim3(C), dim3(WAVE), 0, c->stream, D, Dp, N, draws, c->st.minv, c->st.W); break;
    }
}

// a device temporary that is released on every return path
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// Stage a host array onto the device (returns a temp the caller frees), or pass through.
struct Staged {
    const void* dev = nullptr;
    void* temp = nullptr;
    ~Staged() { if (temp) { (void)hipDeviceSynchronize(); (void)hipFree(temp); } }   // error paths; stage_free is the normal one
};
int stage_in(dhmc_ctx* c, const void* p, size_t bytes, int on_device, Staged* s) {
    if (on_device) { s->dev = p; return DHMC_OK; }
    HIP_TRY(c, hipMalloc(&s->temp, bytes));
    HIP_TRY(c, hipMemcpyAsync(s->temp, p, bytes, hipMemcpyHostToDevice, c->stream));
    s->dev = s->temp;
    return DHMC_OK;
}
void stage_free(dhmc_ctx* c, Staged* s) {
    if (s->temp) { (void)hipStreamSynchronize(c->stream); (void)hipFree(s->temp); s->temp = nullptr; }
}

int read_status(dhmc_ctx* c, std::vector<uint32_t>& st) {
    st.resize(c->cfg.chains);
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->st.status, sizeof(uint32_t) * st.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}
int status_code(dhmc_ctx* c) {
    std::vector<uint32_t> st;
    int rc = read_status(c, st);
    if (rc != DHMC_OK) return rc;
    for (uint32_t s : st)
        if (s) return DHMC_ERR_CHAIN_FAILURE;
    return DHMC_OK;
}

int copy_out_padded(dhmc_ctx* c, const double* padded, double* dst, int on_device) {
    int D = c->cfg.dim, C = c->cfg.chains;
    size_t n = (size_t)C * D;
    double* d = dst;
    DevBuf temp;
    if (!on_device) {
        HIP_TRY(c, hipMalloc(&temp.p, n * sizeof(double)));
        d = (double*)temp.p;
    }
    hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, D, c->Dpad, C, padded, d);
    HIP_TRY(c, hipGetLastError());
    if (!on_device) {
        HIP_TRY(c, hipMemcpyAsync(dst, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return DHMC_OK;
}
int copy_out_scalar(dhmc_ctx* c, const void* src, void* dst, size_t bytes, int on_device) {
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}

}  // namespace

extern "C" {

const char* dhmc_version(void) { return "dhmc_amd 0.1.0 (gfx950)"; }

const char* dhmc_last_error(const dhmc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dhmc_create(const dhmc_config* cfg, dhmc_ctx** out) {
    if (!cfg || !out) return DHMC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (cfg->dim <= 0 || cfg->chains <= 0 || cfg->chain_offset < 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (!(0 < cfg->max_depth && cfg->max_depth <= 32)) return DHMC_ERR_INVALID_ARGUMENT;  // NUTS.jl:190
    if (!(cfg->min_delta < 0)) return DHMC_ERR_INVALID_ARGUMENT;                           // NUTS.jl:191
    if (cfg->metric != DHMC_METRIC_DIAG && cfg->metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    const int D = cfg->dim;
    switch (cfg->target) {
    case DHMC_TARGET_STD_NORMAL: case DHMC_TARGET_ALWAYS_DIVERGENT: break;
    case DHMC_TARGET_FUNNEL: if (D < 2) return DHMC_ERR_INVALID_ARGUMENT; break;
    case DHMC_TARGET_DIAG_NORMAL: case DHMC_TARGET_TRIDIAG_NORMAL:
        if (!cfg->target_params || cfg->target_params_bytes != sizeof(double) * 2 * (size_t)D) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    case DHMC_TARGET_DENSE_NORMAL:
        if (!cfg->target_params || cfg->target_params_bytes != sizeof(double) * ((size_t)D + (size_t)D * D)) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    case DHMC_TARGET_LOGISTIC: {
        if (!cfg->target_params || cfg->target_params_bytes < 8) return DHMC_ERR_INVALID_ARGUMENT;
        int64_t n;
        std::memcpy(&n, cfg->target_params, 8);
        if (n <= 0 || cfg->target_params_bytes != 8 + sizeof(double) * (uint64_t)n * (D + 1)) return DHMC_ERR_INVALID_ARGUMENT;
        break;
    }
    case DHMC_TARGET_EXTERNAL:
        break;
    default:
        if (cfg->target >= DHMC_TARGET_USER_BASE) {
            std::lock_guard<std::mutex> lock(g_user_mutex);
            if ((size_t)(cfg->target - DHMC_TARGET_USER_BASE) >= g_user_targets.size()) return DHMC_ERR_INVALID_ARGUMENT;
            if (D > 1024 || (cfg->metric == DHMC_METRIC_DENSE && cfg->dense_per_chain)) return DHMC_ERR_UNSUPPORTED;
            if (cfg->target_params_bytes % sizeof(double) != 0 || (cfg->target_params_bytes && !cfg->target_params)) return DHMC_ERR_INVALID_ARGUMENT;
            break;
        }
        return DHMC_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return DHMC_ERR_NO_DEVICE;
    dhmc_ctx* c = new (std::nothrow) dhmc_ctx();
    if (!c) return DHMC_ERR_HIP;
    c->cfg = *cfg;
    c->cfg.target_params = nullptr;
    // beyond 1024 coordinates the streaming round-engine kernels serve external models and the built-in normal families
    c->builtin_big = D > 1024 && (cfg->target == DHMC_TARGET_STD_NORMAL || cfg->target == DHMC_TARGET_DIAG_NORMAL ||
                                  cfg->target == DHMC_TARGET_TRIDIAG_NORMAL || cfg->target == DHMC_TARGET_FUNNEL ||
                                  cfg->target == DHMC_TARGET_DENSE_NORMAL || cfg->target == DHMC_TARGET_LOGISTIC ||
                                  cfg->target == DHMC_TARGET_ALWAYS_DIVERGENT);
    // … and the logistic regression with a shared dense metric at ANY width: its functor re-reads X twice per gradient and chain
    // (hundreds of ms per leapfrog at N = 10⁵), the batched evaluation is two GEMMs over all chains
    if (cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DENSE && !cfg->dense_per_chain) c->builtin_big = 1;
    c->NPL = npl_for_dim(D, cfg->target == DHMC_TARGET_EXTERNAL || c->builtin_big);
    if (cfg->target == DHMC_TARGET_EXTERNAL)
        if (const char* e = std::getenv("DHMC_FORCE_NPL")) {       // tests: run a narrow chain through the wide kernels
            const int f = std::atoi(e);
            if ((f == 32 || f == 64) && f >= c->NPL) c->NPL = f;
        }
    if (c->NPL == 0) { delete c; return DHMC_ERR_UNSUPPORTED; }
    c->Dpad = c->NPL * WAVE;
    // Round engines pay ≈8 launches per leapfrog round; they win once a round carries enough chains to fill the
    // chip (GEMM rows), otherwise the one-wave-per-chain kernels are faster.  DHMC_*_ROUNDS=0/1 overrides.
    // logistic regression, diagonal metric: the GEMM engine whatever the chain count — the wave-per-chain functor re-reads X twice per
    // gradient and chain (4 chains, N = 10⁵, p = 256: 306 ms per leapfrog against 0.27; N = 10³, p = 16: 140 µs against 105)
    c->logistic_rounds = cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DIAG && !c->builtin_big;
    if (const char* e = std::getenv("DHMC_LOGISTIC_ROUNDS"))
        c->logistic_rounds = cfg->target == DHMC_TARGET_LOGISTIC && cfg->metric == DHMC_METRIC_DIAG && std::atoi(e) != 0 && !c->builtin_big;
    // dense metric: the wave-per-chain kernel (a matvec per chain from L2, several rows in flight, no launches per round) up to 128
    // coordinates at any chain count and up to 256 below 2048 chains — measured, D = 64 / 128 / 256: 346 / 110 / 29 M leapfrogs/s at
    // 4096 chains against 51 / 76 / 52 for the GEMM rounds, 58 / 33 / 23 against 3 / 3 / 10 at 256–512 chains — the rounds otherwise;
    // the same bits either way
    c->dense_rounds = D > 256 || (D > 128 && cfg->chains >= 2048);
    c->logistic_batched = cfg->target == DHMC_TARGET_LOGISTIC && (c->builtin_big || c->logistic_rounds);
    c->external = cfg->target == DHMC_TARGET_EXTERNAL || c->builtin_big;
    c->nvec = (cfg->metric == DHMC_METRIC_DENSE || c->logistic_rounds || c->external) ? wd_nvec(cfg->max_depth) : ws_nvec(cfg->max_depth);
    if (const char* e = std::getenv("DHMC_L1_LDS")) c->l1_in_lds = std::atoi(e) != 0;  // tuning knob
    if (const char* e = std::getenv("DHMC_K3_BLOCK")) c->k3_block = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_GRAPH")) c->use_graph = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_HOST_CHUNK")) c->host_chunk = std::atoll(e);
    if (const char* e = std::getenv("DHMC_FUSE_K2")) c->fuse_k2 = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_DENSE_ROW_LISTS")) c->dense_row_lists = std::atoi(e) != 0;
    if (const char* e = std::getenv("DHMC_DENSE_PARTS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4) c->dense_parts = v; }
    auto fail = [&](int rc) { dhmc_destroy(c); return rc; };
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(DHMC_ERR_NO_DEVICE);
    const size_t C = cfg->chains, Dp = c->Dpad;
    int rc;
    if ((rc = dev_alloc(c, &c->st.q, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.g, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.lq, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.minv, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.W, C * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.eps, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.da, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.transition, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.status, C))) return fail(rc);
    if ((rc = dev_alloc(c, &c->st.ws, C * (size_t)c->nvec * Dp))) return fail(rc);
    if ((rc = dev_alloc(c, &c->d_counter, 1))) return fail(rc);
    if (hipMemset(c->st.q, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.g, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.da, 0, C * sizeof(DAState)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.status, 0, C * sizeof(uint32_t)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (hipMemset(c->st.transition, 0, C * sizeof(uint32_t)) != hipSuccess) return fail(DHMC_ERR_HIP);
    if (cfg->target >= DHMC_TARGET_USER_BASE) {
        // the caller's functor: its parameters to the device, its kernels from the cache or from hiprtc
        const size_t nb = (size_t)cfg->target_params_bytes;
        if (nb) {
            if (hipMalloc(&c->d_user_params, nb) != hipSuccess) return fail(DHMC_ERR_HIP);
            c->allocs.push_back(c->d_user_params);
            if (hipMemcpy(c->d_user_params, cfg->target_params, nb, hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        c->tp.a = (const double*)c->d_user_params;
        c->tp.n = (int64_t)(nb / sizeof(double));
        std::lock_guard<std::mutex> lock(g_user_mutex);
        UserTarget& U = g_user_targets[cfg->target - DHMC_TARGET_USER_BASE];
        const auto key = std::make_pair((int)cfg->device, c->NPL);
        auto it = U.built.find(key);
        if (it == U.built.end()) {
            std::vector<char> code;
            std::vector<std::string> low;
            if ((rc = rtc_compile(U.source, U.name, c->NPL, false, &code, &low))) return fail(rc);
            UserKernels K;
            if ((rc = rtc_load(code, low, &K.mod, {&K.run_lds, &K.run, &K.init, &K.search, &K.probe_traj, &K.probe_ratio}))) return fail(rc);
            it = U.built.emplace(key, K).first;
        }
        if (cfg->metric == DHMC_METRIC_DENSE && !it->second.dense_mod) {
            std::vector<char> code;
            std::vector<std::string> low;
            if ((rc = rtc_compile(U.source, U.name, c->NPL, true, &code, &low))) return fail(rc);
            UserKernels& K = it->second;
            if ((rc = rtc_load(code, low, &K.dense_mod, {&K.k0, &K.k2, &K.k3, &K.run_dense, &K.search_dense, &K.probe_traj_dense, &K.probe_ratio_dense})))
                return fail(rc);
        }
        c->user = &it->second;
    }
    if (cfg->target == DHMC_TARGET_DIAG_NORMAL || cfg->target == DHMC_TARGET_TRIDIAG_NORMAL) {
        std::vector<double> a(Dp, 0.0), b(Dp, 0.0);
        const double* src = (const double*)cfg->target_params;
        std::memcpy(a.data(), src, sizeof(double) * D);
        std::memcpy(b.data(), src + D, sizeof(double) * D);
        double *da = nullptr, *db = nullptr;
        if ((rc = dev_alloc(c, &da, Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &db, Dp))) return fail(rc);
        if (hipMemcpy(da, a.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(db, b.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = da;
        c->tp.b = db;
    }
    if (cfg->metric == DHMC_METRIC_DENSE) {   // GaussianKineticEnergy(N) as a dense identity
        c->per_chain_dense = cfg->dense_per_chain != 0;
        if (c->per_chain_dense && c->external) return fail(DHMC_ERR_UNSUPPORTED);
        const size_t nmat = c->per_chain_dense ? C : 1;
        if ((rc = dev_alloc(c, &c->d_Minv, nmat * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_WT, nmat * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_fwork, 4 * Dp * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_fflags, 2))) return fail(rc);
        c->dm = DenseMetric{c->d_Minv, c->d_WT, c->per_chain_dense ? Dp * Dp : (size_t)0};
        if (const char* e = std::getenv("DHMC_DENSE_ROUNDS")) c->dense_rounds = std::atoi(e) != 0;  // 0: wave-per-chain matvec kernel
        if (c->per_chain_dense) c->dense_rounds = 0;   // the GEMM engine shares one M⁻¹ across the rows of a product
        c->dense_products = c->per_chain_dense ? 2 : 1;
        if (const char* e = std::getenv("DHMC_DENSE_PRODUCTS")) { const int v = std::atoi(e); if (v == 1 || v == 2) c->dense_products = v; }
        std::vector<double> I((size_t)D * D, 0.0);
        for (int i = 0; i < D; ++i) I[(size_t)i * D + i] = 1.0;
        if ((rc = upload_dense_metric(c, I, I))) return fail(rc);
    }
    if (cfg->metric == DHMC_METRIC_DENSE || c->logistic_rounds || c->external) {   // buffers of the round engines
        if ((rc = dev_alloc(c, &c->rb.cp, C * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.cps, C * Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.tbuf, C * Dp))) return fail(rc);
        if (cfg->metric == DHMC_METRIC_DENSE) {
            if ((rc = dev_alloc(c, &c->rb.cu, C * Dp))) return fail(rc);
            if (hipMemset(c->rb.cu, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        if ((rc = dev_alloc(c, &c->rb.ts, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.list, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->rb.list_count, 8))) return fail(rc);
        if (cfg->metric == DHMC_METRIC_DENSE && !c->lr.act) {       // row lists of the dense round engine: [C] + a counter per part
            if ((rc = dev_alloc(c, &c->lr.act, C + 4))) return fail(rc);
            c->lr.act_count = c->lr.act + C;
        }
        c->rb.done_count = c->rb.list_count + 1;
        if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) return fail(DHMC_ERR_HIP);
        for (int i = 2; i < 4; ++i)
            if (hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        for (int i = 1; i < 4; ++i)
            if (hipEventCreateWithFlags(&c->ev_joins[i], hipEventDisableTiming) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.ts, 0, C * sizeof(TreeState)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.cp, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.cps, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemset(c->rb.tbuf, 0, C * Dp * sizeof(double)) != hipSuccess) return fail(DHMC_ERR_HIP);
    }
    c->tp.Dpad = (int32_t)Dp;
    if (cfg->target == DHMC_TARGET_DENSE_NORMAL) {
        const double* src = (const double*)cfg->target_params;
        std::vector<double> mu(Dp, 0.0), Pm(Dp * Dp, 0.0);
        std::memcpy(mu.data(), src, sizeof(double) * D);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) Pm[(size_t)i * Dp + j] = (i <= j) ? src[D + (size_t)i * D + j] : src[D + (size_t)j * D + i];
        double *dmu = nullptr, *dP = nullptr;
        if ((rc = dev_alloc(c, &dmu, Dp))) return fail(rc);
        if ((rc = dev_alloc(c, &dP, Dp * Dp))) return fail(rc);
        if (hipMemcpy(dmu, mu.data(), Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dP, Pm.data(), Dp * Dp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = dmu; c->tp.b = dP;
    }
    if (cfg->target == DHMC_TARGET_LOGISTIC) {
        // X [npad][Dpad] row-major and Xᵀ [Dpad][npad], zero padded (GEMM operands of the round engine; the
        // wave-per-chain functor reads the same arrays), y [npad]
        int64_t n;
        std::memcpy(&n, cfg->target_params, 8);
        const double* X = (const double*)((const char*)cfg->target_params + 8);
        const double* y = X + (size_t)n * D;
        const size_t npad = ((size_t)n + WAVE - 1) / WAVE * WAVE;
        std::vector<double> xp(npad * Dp, 0.0), xt(Dp * npad, 0.0), yp(npad, 0.0);
        for (int64_t i = 0; i < n; ++i)
            for (int d = 0; d < D; ++d) {
                xp[(size_t)i * Dp + d] = X[(size_t)i * D + d];
                xt[(size_t)d * npad + i] = X[(size_t)i * D + d];
            }
        for (int64_t i = 0; i < n; ++i) yp[i] = y[i];
        double *dx = nullptr, *dxt = nullptr, *dy = nullptr;
        if ((rc = dev_alloc(c, &dx, xp.size()))) return fail(rc);
        if ((rc = dev_alloc(c, &dxt, xt.size()))) return fail(rc);
        if ((rc = dev_alloc(c, &dy, yp.size()))) return fail(rc);
        if (hipMemcpy(dx, xp.data(), xp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dxt, xt.data(), xt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(dy, yp.data(), yp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        c->tp.a = dx; c->tp.b = dxt; c->tp.c = dy; c->tp.n = n; c->tp.npad = (int64_t)npad; c->tp.Dpad = (int32_t)Dp;
        if (c->logistic_batched) {        // the gradient of all chains by GEMMs, in the rounds and between other kernels (external_eval)
            if ((rc = dev_alloc(c, &c->lr.H, C * npad))) return fail(rc);
            c->lr.nz = (int)((npad + DHMC_LOGISTIC_BLOCK - 1) / DHMC_LOGISTIC_BLOCK);
            if ((rc = dev_alloc(c, &c->lr.P, (size_t)c->lr.nz * C * Dp))) return fail(rc);
            if ((rc = dev_alloc(c, &c->lr.S1P, (size_t)c->lr.nz * C))) return fail(rc);
            if ((rc = dev_alloc(c, &c->d_all_rows, C + 1))) return fail(rc);
            if (c->builtin_big && !c->lr.act) {                          // row list of a round (the rounds engine below has its own)
                if ((rc = dev_alloc(c, &c->lr.act, C + 4))) return fail(rc);
                c->lr.act_count = c->lr.act + C;
            }
            hipLaunchKernelGGL(builtin_all_rows_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, nullptr, (int)C, c->d_all_rows, c->d_all_rows + C);
            if (hipDeviceSynchronize() != hipSuccess) return fail(DHMC_ERR_HIP);
        }
        if (c->logistic_rounds) {         // (H, P, S1P: above)
            if ((rc = dev_alloc(c, &c->lr.S1, C))) return fail(rc);
            if ((rc = dev_alloc(c, &c->lr.act, C + 1))) return fail(rc);
            c->lr.act_count = c->lr.act + C;
            if ((rc = dev_alloc(c, &c->d_ss, C))) return fail(rc);      // the batched step-size search's per-chain state
        }
    }
    if (c->builtin_big && cfg->target == DHMC_TARGET_DENSE_NORMAL)
        for (int i = 0; i < 2; ++i)
            if ((rc = dev_alloc(c, &c->d_big[i], C * Dp))) return fail(rc);
    if (c->external) {
        if ((rc = dev_alloc(c, &c->lr.S1, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_ss, C))) return fail(rc);
        if ((rc = dev_alloc(c, &c->d_sflags, 4 * C))) return fail(rc);
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return fail(DHMC_ERR_HIP);
    // unit metric, ε unspecified
    {
        std::vector<double> ones(C * Dp, 1.0), w(C * Dp, 0.0), nanv(C, std::nan(""));
        for (size_t ch = 0; ch < C; ++ch)
            for (int e = 0; e < D; ++e) w[ch * Dp + e] = 1.0;
        if (hipMemcpy(c->st.minv, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(c->st.W, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
        if (hipMemcpy(c->st.eps, nanv.data(), C * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(DHMC_ERR_HIP);
    }
    *out = c;
    return DHMC_OK;
}

int dhmc_register_target_source(const char* hip_source, const char* functor_name, int32_t* target_handle) {
    if (!hip_source || !functor_name || !*functor_name || !target_handle) return DHMC_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(g_user_mutex);
    g_user_targets.push_back(UserTarget{hip_source, functor_name, {}});
    *target_handle = (int32_t)g_user_targets.size() - 1;
    return DHMC_OK;
}
int dhmc_check_target_source(const char* hip_source, const char* functor_name, int32_t dim, int32_t metric, char* log, uint64_t log_bytes) {
    if (!hip_source || !functor_name || (metric != DHMC_METRIC_DIAG && metric != DHMC_METRIC_DENSE)) return DHMC_ERR_INVALID_ARGUMENT;
    const int npl = npl_for_user_dim(dim);
    if (npl == 0) return DHMC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_user_mutex);
    const int rc = rtc_compile(hip_source, functor_name, npl, metric == DHMC_METRIC_DENSE, nullptr, nullptr);
    if (log && log_bytes) {
        const size_t n = std::min<size_t>(g_rtc_log.size(), (size_t)log_bytes - 1);
        std::memcpy(log, g_rtc_log.data(), n);
        log[n] = '\0';
    }
    return rc;
}
const char* dhmc_target_source_log(void) { return g_rtc_log.c_str(); }

int dhmc_host_alloc(void** out, uint64_t nbytes) {
    if (!out) return DHMC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (nbytes == 0) return DHMC_OK;
    return hipHostMalloc(out, (size_t)nbytes, hipHostMallocDefault) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
}
int dhmc_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? DHMC_OK : DHMC_ERR_HIP; }

int dhmc_destroy(dhmc_ctx* c) {
    if (!c) return DHMC_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    for (int i = 2; i < 4; ++i)
        if (c->streams[i]) { (void)hipStreamSynchronize(c->streams[i]); (void)hipStreamDestroy(c->streams[i]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (int i = 1; i < 4; ++i) if (c->ev_joins[i]) (void)hipEventDestroy(c->ev_joins[i]);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_done) (void)hipHostFree(c->h_done);
    for (int b = 0; b < 2; ++b) if (c->ev_done[b]) (void)hipEventDestroy(c->ev_done[b]);
    for (int b = 0; b < 2; ++b) {
        for (auto& sb : c->stage[b]) if (sb.p) (void)hipFree(sb.p);
        if (c->ev_k0[b]) (void)hipEventDestroy(c->ev_k0[b]);
        if (c->ev_k1[b]) (void)hipEventDestroy(c->ev_k1[b]);
        if (c->ev_copy[b]) (void)hipEventDestroy(c->ev_copy[b]);
    }
    delete c;
    return DHMC_OK;
}

// ---- DHMC_TARGET_EXTERNAL (external_rounds.hpp) -------------------------------------------------
namespace {
#define DHMC_EXT_NPL_FWD(KERNEL, GRID, ...)                                                                    \
    switch (c->NPL) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    case 32: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    default: hipLaunchKernelGGL((KERNEL<64>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    }
// ℓ and ∇ℓ of `q` ([C][Dpad], device) for all chains through the host's callback: lq -> c->lr.S1, grad -> c->rb.tbuf
// `active` (the logistic family's GEMM evaluation only): the rows of the chains in a leaf phase, listed in c->lr.act by the caller —
// the round loops pass it, so that chains which have finished their transitions are not multiplied
int external_eval(dhmc_ctx* c, const double* q, bool active = false) {
    if (c->logistic_batched) {
        // the GEMM gradient of the logistic round engine over all chains (logistic_rounds.hpp), folded by builtin_logistic_fold_kernel
        const int C = c->cfg.chains, ld = c->Dpad, npad = (int)c->tp.npad;
        RunParams P{};
        P.D = c->cfg.dim; P.Dpad = ld; P.C = C; P.tp = c->tp;
        RoundBuffers R{};
        LogisticRound L = c->lr;
        if (!(active && c->lr.act)) { L.act = c->d_all_rows; L.act_count = c->d_all_rows + C; }                          // else: every chain
        launch_gemm_list(q, ld, c->tp.b, npad, L.H, npad, C, ld, npad, L.act, L.act_count, c->stream);                    // η = Q·Xᵀ
        hipLaunchKernelGGL(logistic_link_kernel, dim3((unsigned)L.nz, C), dim3(WAVE), 0, c->stream, P, R, L);            // r, the blocks' sums
        launch_gemm_splitk(L.H, npad, c->tp.a, ld, L.P, ld, (size_t)C * ld, C, npad, ld, DHMC_LOGISTIC_BLOCK, L.act, L.act_count,
                           c->stream);                                                                                   // Xᵀr, block by block
        DHMC_EXT_NPL_FWD(builtin_logistic_fold_kernel, dim3(C), C, ld, q, L, c->lr.S1, c->rb.tbuf)                         // (listed chains only)
        return DHMC_OK;
    }
    if (c->builtin_big && c->cfg.target == DHMC_TARGET_DENSE_NORMAL) {
        const dim3 g(c->cfg.chains), b(WAVE);
        const int ld = c->Dpad;
        if (c->NPL == 32) hipLaunchKernelGGL((builtin_dense_normal_pre_kernel<32>), g, b, 0, c->stream, ld, q, c->tp.a, c->d_big[0]);
        else hipLaunchKernelGGL((builtin_dense_normal_pre_kernel<64>), g, b, 0, c->stream, ld, q, c->tp.a, c->d_big[0]);
        launch_gemm_rows(c->d_big[0], c->tp.b, c->d_big[1], ld, c->cfg.chains, nullptr, nullptr, c->stream);      // P·d for every chain
        if (c->NPL == 32) hipLaunchKernelGGL((builtin_dense_normal_post_kernel<32>), g, b, 0, c->stream, ld, (const double*)c->d_big[0], (const double*)c->d_big[1], c->lr.S1, c->rb.tbuf);
        else hipLaunchKernelGGL((builtin_dense_normal_post_kernel<64>), g, b, 0, c->stream, ld, (const double*)c->d_big[0], (const double*)c->d_big[1], c->lr.S1, c->rb.tbuf);
        return DHMC_OK;
    }
    if (c->builtin_big) {
        const int kind = c->cfg.target == DHMC_TARGET_STD_NORMAL ? 0 : c->cfg.target == DHMC_TARGET_DIAG_NORMAL ? 1 :
                         c->cfg.target == DHMC_TARGET_TRIDIAG_NORMAL ? 2 : c->cfg.target == DHMC_TARGET_FUNNEL ? 3 : 4;
        const dim3 g(c->cfg.chains), b(WAVE);
        if (c->NPL == 32)
            hipLaunchKernelGGL((builtin_normal_eval_kernel<32>), g, b, 0, c->stream, kind, c->cfg.dim, c->Dpad, q, c->tp.a, c->tp.b, c->lr.S1, c->rb.tbuf);
        else
            hipLaunchKernelGGL((builtin_normal_eval_kernel<64>), g, b, 0, c->stream, kind, c->cfg.dim, c->Dpad, q, c->tp.a, c->tp.b, c->lr.S1, c->rb.tbuf);
        return DHMC_OK;
    }
    if (!c->ext_fn) { c->err = "DHMC_TARGET_EXTERNAL: no callback set (dhmc_set_logdensity_callback)"; return DHMC_ERR_CALLBACK; }
    const int rc = c->ext_fn(c->ext_user, q, c->cfg.chains, c->Dpad, c->cfg.dim, c->lr.S1, c->rb.tbuf, (void*)c->stream);
    if (rc != 0) { c->err = "DHMC_TARGET_EXTERNAL: the callback returned " + std::to_string(rc); return DHMC_ERR_CALLBACK; }
    return DHMC_OK;
}
#define DHMC_EXT_NPL(KERNEL, GRID, ...)                                                                        \
    switch (c->NPL) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;               \
    case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    case 32: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    default: hipLaunchKernelGGL((KERNEL<64>), GRID, dim3(WAVE), 0, c->stream, __VA_ARGS__); break;             \
    }
}  // namespace

int dhmc_set_logdensity_callback(dhmc_ctx* c, dhmc_logdensity_fn fn, void* user) {
    if (!c || !c->external || c->builtin_big) return DHMC_ERR_INVALID_ARGUMENT;
    c->ext_fn = fn;
    c->ext_user = user;
    return DHMC_OK;
}

int dhmc_set_stream(dhmc_ctx* c, void* s) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    c->stream = (hipStream_t)s;
    return DHMC_OK;
}

int dhmc_init(dhmc_ctx* c, const double* q0, int q0_on_device) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // (the poisoned mark is lifted only once the new (q, ℓq, ∇ℓ) are complete; a failure after the positions were
    // overwritten sets it)
    Staged s;
    if (q0) {
        int rc = stage_in(c, q0, sizeof(double) * (size_t)c->cfg.chains * c->cfg.dim, q0_on_device, &s);
        if (rc) return rc;
    }
    InitParams P{c->cfg.dim, c->Dpad, c->cfg.chains, c->cfg.chain_offset, c->cfg.seed, (const double*)s.dev, c->st, c->tp};
    int rc = DHMC_OK;
    const bool batched = c->external || c->logistic_batched;     // ℓ, ∇ℓ of all chains between kernels (callback / library GEMMs)
    if (batched) { DHMC_EXT_NPL(external_init_positions_kernel, dim3(c->cfg.chains), P) }
    else rc = dispatch(c, Op::Init, &P);
    if (rc) { stage_free(c, &s); return rc; }
    c->poisoned = true;    // st.q is being overwritten: inconsistent with ℓq, ∇ℓ until the evaluation below has succeeded
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    if (batched) {         // the positions are set; ℓ and ∇ℓ come from the callback, then evaluate_ℓ(strict) (mcmc.jl:131)
        if ((rc = external_eval(c, c->st.q))) return rc;
        DHMC_EXT_NPL(external_init_finish_kernel, dim3(c->cfg.chains), c->cfg.dim, c->Dpad, c->st, c->lr.S1, c->rb.tbuf)
        HIP_TRY(c, hipGetLastError());
    }
    c->poisoned = false;
    return status_code(c);
}

int dhmc_set_position(dhmc_ctx* c, const double* q, int on_device) {
    if (!c || !q) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // dhmc_init does the evaluation (and resets κ, ϵ, counters): keep those aside and put them back
    const size_t C = c->cfg.chains, Dp = c->Dpad;
    struct Keep { void* live; size_t bytes; DevBuf copy; };
    Keep keep[7] = {{c->st.minv, C * Dp * sizeof(double), {}}, {c->st.W, C * Dp * sizeof(double), {}}, {c->st.eps, C * sizeof(double), {}},
                    {c->st.da, C * sizeof(DAState), {}}, {c->st.transition, C * sizeof(uint32_t), {}},
                    {nullptr, 0, {}}, {nullptr, 0, {}}};   // (dhmc_init leaves a dense metric alone)
    for (auto& k : keep) {
        if (!k.live) continue;
        HIP_TRY(c, hipMalloc(&k.copy.p, k.bytes));
        HIP_TRY(c, hipMemcpyAsync(k.copy.p, k.live, k.bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    const int rc = dhmc_init(c, q, on_device);    // (a failed evaluation leaves the context poisoned, as in dhmc_init)
    for (auto& k : keep)
        if (k.live) HIP_TRY(c, hipMemcpyAsync(k.live, k.copy.p, k.bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return rc;
}

int dhmc_get_position(dhmc_ctx* c, double* q, double* lq, double* grad, int on_device) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if (q && (rc = copy_out_padded(c, c->st.q, q, on_device))) return rc;
    if (grad && (rc = copy_out_padded(c, c->st.g, grad, on_device))) return rc;
    if (lq && (rc = copy_out_scalar(c, c->st.lq, lq, sizeof(double) * c->cfg.chains, on_device))) return rc;
    return DHMC_OK;
}

int dhmc_set_metric_diag(dhmc_ctx* c, const double* minv, int per_chain, int on_device) {
    if (!c || !minv || c->cfg.metric != DHMC_METRIC_DIAG) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim, C = c->cfg.chains;
    const size_t n = per_chain ? (size_t)C * D : (size_t)D;
    if (!on_device) {
        for (size_t i = 0; i < n; ++i)
            if (!(minv[i] > 0) || !std::isfinite(minv[i])) return DHMC_ERR_INVALID_ARGUMENT;
    } else {                                   // the same @argcheck (hamiltonian.jl:63) for a device array
        DevBuf flag;
        int bad = 0;
        HIP_TRY(c, hipMalloc(&flag.p, sizeof(int)));
        HIP_TRY(c, hipMemsetAsync(flag.p, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(check_positive_finite_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, minv, n, (int*)flag.p);
        HIP_TRY(c, hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (bad) return DHMC_ERR_INVALID_ARGUMENT;
    }
    Staged s;
    int rc = stage_in(c, minv, n * sizeof(double), on_device, &s);
    if (rc) return rc;
    size_t tot = (size_t)C * c->Dpad;
    hipLaunchKernelGGL(set_metric_diag_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, D, c->Dpad, C,
                       (const double*)s.dev, per_chain, c->st.minv, c->st.W);
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    return DHMC_OK;
}

int dhmc_get_metric_diag(dhmc_ctx* c, double* minv, int on_device) {
    if (!c || !minv) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_padded(c, c->st.minv, minv, on_device);
}

int dhmc_set_metric_dense(dhmc_ctx* c, const double* minv, int on_device) {
    if (!c || !minv || c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim;
    Staged s;
    int rc = stage_in(c, minv, sizeof(double) * (size_t)D * D, on_device, &s);
    if (rc) return rc;
    rc = device_dense_metric(c, (const double*)s.dev, D);      // symmetrise, check, factorise: all on the device
    stage_free(c, &s);
    return rc;
}

int dhmc_set_dense_products(dhmc_ctx* c, int32_t products) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || (products != 1 && products != 2)) return DHMC_ERR_INVALID_ARGUMENT;
    c->dense_products = products;
    return DHMC_OK;
}
int dhmc_get_dense_products(const dhmc_ctx* c) { return (c && c->cfg.metric == DHMC_METRIC_DENSE) ? c->dense_products : 0; }

int dhmc_get_metric_dense_chain(dhmc_ctx* c, int32_t chain, double* minv, double* W) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || chain < 0 || chain >= c->cfg.chains) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim;
    const size_t Dp = c->Dpad;
    const size_t off = c->per_chain_dense ? (size_t)chain * Dp * Dp : 0;
    std::vector<double> a(Dp * Dp), b(Dp * Dp);
    HIP_TRY(c, hipMemcpyAsync(a.data(), c->d_Minv + off, a.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(b.data(), c->d_WT + off, b.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            if (minv) minv[(size_t)i * D + j] = a[(size_t)i * Dp + j];
            if (W) W[(size_t)i * D + j] = b[(size_t)j * Dp + i];
        }
    return DHMC_OK;
}

int dhmc_get_metric_dense(dhmc_ctx* c, double* minv, double* W) { return dhmc_get_metric_dense_chain(c, 0, minv, W); }

int dhmc_set_stepsize(dhmc_ctx* c, const double* eps, int per_chain, int on_device) {
    if (!c || !eps) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains;
    std::vector<double> h(C);
    if (on_device) {
        if (!per_chain) return DHMC_ERR_INVALID_ARGUMENT;
        HIP_TRY(c, hipMemcpyAsync(h.data(), eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    } else {
        for (int i = 0; i < C; ++i) h[i] = eps[per_chain ? i : 0];
    }
    for (int i = 0; i < C; ++i)
        if (!(h[i] > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    HIP_TRY(c, hipMemcpyAsync(c->st.eps, h.data(), sizeof(double) * C, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DHMC_OK;
}

int dhmc_get_stepsize(dhmc_ctx* c, double* eps, int on_device) {
    if (!c || !eps) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_scalar(c, c->st.eps, eps, sizeof(double) * c->cfg.chains, on_device);
}

int dhmc_get_status(dhmc_ctx* c, uint32_t* status) {
    if (!c || !status) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return copy_out_scalar(c, c->st.status, status, sizeof(uint32_t) * c->cfg.chains, 0);
}

int dhmc_find_initial_stepsize(dhmc_ctx* c, const dhmc_stepsize_search* p) {
    if (!c) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    dhmc_stepsize_search d{0.1, std::log(0.8), 400, 0};
    if (p) d = *p;
    if (!(std::isfinite(d.log_threshold) && d.log_threshold < 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:31
    if (!(std::isfinite(d.initial_eps) && 0 < d.initial_eps)) return DHMC_ERR_INVALID_ARGUMENT;      // :32
    if (!(d.maxiter_crossing >= 50)) return DHMC_ERR_INVALID_ARGUMENT;                                // :33
    const int C = c->cfg.chains;
    std::vector<double> h(C);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->st.eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (double e : h)
        if (!std::isnan(e)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:137 "stepsize ϵ manually specified"
    if (c->external && c->cfg.metric == DHMC_METRIC_DENSE) {
        // dense metric: p₀ = z·Wᵀ, M⁻¹pₘ and p′♯ are GEMMs over all chains between the search kernels
        ExtSearchParams E{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold, d.maxiter_crossing,
                          c->st, c->d_ss, c->rb.cps, c->rb.cp, c->lr.S1, c->rb.tbuf, c->rb.list_count};
        const int ld = c->Dpad;
        const size_t p1_stride = (size_t)c->nvec * c->Dpad;
        int remaining = 0;
        HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
        DHMC_EXT_NPL(ext_search_dense_z_kernel, dim3(C), E, c->rb.cp)
        launch_gemm_rows(c->rb.cp, c->d_WT, c->rb.cps, ld, C, nullptr, nullptr, c->stream);                 // p₀ = z·Wᵀ
        launch_gemm_rows(c->rb.cps, c->d_Minv, c->rb.tbuf, ld, C, nullptr, nullptr, c->stream);             // p₀♯
        DHMC_EXT_NPL(ext_search_dense_begin_kernel, dim3(C), E, (const double*)c->rb.tbuf, c->rb.cp)
        HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        while (remaining > 0) {
            launch_gemm_rows(c->rb.cp, c->d_Minv, c->rb.tbuf, ld, C, nullptr, nullptr, c->stream);          // M⁻¹pₘ
            DHMC_EXT_NPL(ext_search_dense_trial_kernel, dim3(C), E, (const double*)c->rb.tbuf)
            int rc = external_eval(c, c->rb.cp);
            if (rc) return rc;
            DHMC_EXT_NPL(ext_search_dense_p1_kernel, dim3(C), E, c->d_sflags, c->st.ws, p1_stride)
            launch_gemm(c->st.ws, (int)p1_stride, c->d_Minv, ld, c->rb.tbuf, ld, C, ld, ld, c->stream);     // p′♯
            HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
            DHMC_EXT_NPL(ext_search_dense_decide_kernel, dim3(C), E, (const double*)c->st.ws, p1_stride, (const double*)c->rb.tbuf,
                         (const uint32_t*)c->d_sflags, c->rb.cp)
            HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        HIP_TRY(c, hipGetLastError());
        return status_code(c);
    }
    if (c->external || c->logistic_batched) {
        // the same bracketing search, all chains per callback: trial positions -> callback -> one decision per chain
        ExtSearchParams E{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold, d.maxiter_crossing,
                          c->st, c->d_ss, c->rb.cps, c->rb.cp, c->lr.S1, c->rb.tbuf, c->rb.list_count};
        int remaining = 0;
        HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
        DHMC_EXT_NPL(ext_search_begin_kernel, dim3(C), E)
        HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        while (remaining > 0) {
            int rc = external_eval(c, c->rb.cp);
            if (rc) return rc;
            HIP_TRY(c, hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream));
            DHMC_EXT_NPL(ext_search_step_kernel, dim3(C), E)
            HIP_TRY(c, hipMemcpyAsync(&remaining, c->rb.list_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        HIP_TRY(c, hipGetLastError());
        return status_code(c);
    }
    SearchParams P{c->cfg.dim, c->Dpad, C, c->cfg.chain_offset, c->cfg.seed, d.initial_eps, d.log_threshold,
                   d.maxiter_crossing, c->st, c->tp};
    int rc = dispatch(c, Op::Search, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    return status_code(c);
}

namespace {
int run_call(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (da) {
        if (!(0 < da->delta && da->delta < 1)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:108
        if (!(da->gamma > 0)) return DHMC_ERR_INVALID_ARGUMENT;                   // :109
        if (!(0.5 < da->kappa && da->kappa <= 1)) return DHMC_ERR_INVALID_ARGUMENT;  // :110
        if (!(da->t0 >= 0)) return DHMC_ERR_INVALID_ARGUMENT;                     // :111
    }
    const int C = c->cfg.chains, D = c->cfg.dim;
    if (!da || da->init) {
        std::vector<double> h(C);
        HIP_TRY(c, hipMemcpyAsync(h.data(), c->st.eps, sizeof(double) * C, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (double e : h)
            if (!(e > 0)) return DHMC_ERR_INVALID_ARGUMENT;  // stepsize.jl:135
    }
    c->last_ms = 0.0;
    c->last_leapfrogs = 0;
    if (N == 0) return status_code(c);

    RunParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset;
    P.max_depth = c->cfg.max_depth; P.nvec = c->nvec; P.min_delta = c->cfg.min_delta; P.seed = c->cfg.seed;
    P.N = N; P.st = c->st; P.tp = c->tp; P.leapfrog_counter = c->d_counter;
    P.l1_in_lds = c->l1_in_lds;
    P.k3_block = c->k3_block;
    P.one_product = c->cfg.metric == DHMC_METRIC_DENSE && c->dense_products == 1;
    if (da) {
        P.adapt = 1; P.da_init = da->init; P.da_finalize = da->finalize; P.t0 = da->t0;
        P.delta = da->delta; P.gamma = da->gamma; P.kappa = da->kappa;
    }
    // Outputs: device pointers pass through.  Host pointers are served from the context's persistent staging buffers
    // (grown on demand: no hipMalloc / hipFree per call).  The one-kernel engine (diagonal metric) runs a call with host
    // outputs in CHUNKS of L transitions, two staging buffers deep: chunk k leaves over the copy stream (strided 2-D
    // copies into the caller's [C][N][…] arrays; truly asynchronous when those are page-locked — dhmc_host_alloc) while
    // chunk k+1 computes.  The chunks are the same transitions of the same kernel as one launch would run: same bits.
    struct Field { void** dev; void* host; size_t elem; int idx; };   // elem: bytes of one (chain, transition) record
    std::vector<Field> staged;
    const bool one_kernel = c->cfg.metric == DHMC_METRIC_DIAG && !c->logistic_rounds && !c->external;
    const bool host_out = out && !out->on_device &&
                          (out->draws || out->logdensities || out->eps || out->pi || out->acceptance_rate || out->steps ||
                           out->term_left || out->term_right || out->depth || out->directions);
    int64_t L = N;
    if (host_out && one_kernel) {
        const int64_t per_transition = (int64_t)C * D * (int64_t)sizeof(double);
        // default: ≈ 1 GiB of draws per chunk, but at least four chunks per call so that most of the copy runs under a kernel
        L = c->host_chunk > 0 ? c->host_chunk : std::min(((int64_t)1 << 30) / (per_transition > 0 ? per_transition : 1), (N + 3) / 4);
        if (L < 1) L = 1;
        if (L > N) L = N;
    }
    const int nbuf = L < N ? 2 : 1;
    auto bind = [&](void* user, void** slot, size_t elem, int idx) -> int {
        *slot = nullptr;
        if (!user) return DHMC_OK;
        if (out->on_device) { *slot = user; return DHMC_OK; }
        const size_t need = (size_t)C * (size_t)L * elem;
        for (int b = 0; b < nbuf; ++b) {
            auto& sb = c->stage[b][idx];
            if (sb.cap < need) {
                if (sb.p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(sb.p)); sb.p = nullptr; sb.cap = 0; }
                HIP_TRY(c, hipMalloc(&sb.p, need));
                sb.cap = need;
            }
        }
        *slot = c->stage[0][idx].p;
        staged.push_back({slot, user, elem, idx});
        return DHMC_OK;
    };
    int rc = DHMC_OK;
    if (out) {
        if (!rc) rc = bind(out->draws, (void**)&P.out.draws, D * sizeof(double), 0);
        if (!rc) rc = bind(out->logdensities, (void**)&P.out.logdensities, sizeof(double), 1);
        if (!rc) rc = bind(out->eps, (void**)&P.out.eps, sizeof(double), 2);
        if (!rc) rc = bind(out->pi, (void**)&P.out.pi, sizeof(double), 3);
        if (!rc) rc = bind(out->acceptance_rate, (void**)&P.out.acceptance_rate, sizeof(double), 4);
        if (!rc) rc = bind(out->steps, (void**)&P.out.steps, sizeof(int64_t), 5);
        if (!rc) rc = bind(out->term_left, (void**)&P.out.term_left, sizeof(int64_t), 6);
        if (!rc) rc = bind(out->term_right, (void**)&P.out.term_right, sizeof(int64_t), 7);
        if (!rc) rc = bind(out->depth, (void**)&P.out.depth, sizeof(int32_t), 8);
        if (!rc) rc = bind(out->directions, (void**)&P.out.directions, sizeof(uint32_t), 9);
    }
    auto cleanup = [&]() {};   // (the staging buffers belong to the context)
    if (rc) { cleanup(); return rc; }
    if (!staged.empty() && !c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(c, hipEventCreate(&c->ev_k0[b]));
            HIP_TRY(c, hipEventCreate(&c->ev_k1[b]));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copy[b], hipEventDisableTiming));
        }
    }
    // transitions [n0, n0 + len) of every staged field: staging buffer b (record stride L) -> the caller's arrays (stride N)
    auto d2h = [&](int b, int64_t n0, int64_t len, hipStream_t s) -> hipError_t {
        for (auto& f : staged) {
            char* dst = (char*)f.host + (size_t)n0 * f.elem;
            hipError_t ce = hipMemcpy2DAsync(dst, (size_t)N * f.elem, c->stage[b][f.idx].p, (size_t)L * f.elem, (size_t)len * f.elem,
                                             (size_t)C, hipMemcpyDeviceToHost, s);
            if (ce != hipSuccess) return ce;
        }
        return hipSuccess;
    };
    double chunk_ms = 0.0;

    hipError_t e = hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
    if (e == hipSuccess && c->external && c->cfg.metric == DHMC_METRIC_DENSE) {
        // dense round engine (dense_rounds.hpp) with the host's callback as the density, one batch on one stream
        P.one_product = c->dense_products == 1;
        RoundArgs ra{P, c->rb};
        const int ld = c->Dpad;
        const RoundBuffers& R = c->rb;
        e = hipMemsetAsync(R.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_gemm_rows(R.cp, c->d_WT, R.tbuf, ld, C, R.list, R.list_count, c->stream);       // p₀ = z·Wᵀ
                launch_gemm_rows(R.tbuf, c->d_Minv, R.cps, ld, C, R.list, R.list_count, c->stream);    // p♯₀
                if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, R.list, R.list_count, c->stream);   // u₀ = ∇ℓq₀·M⁻¹
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(R.list_count, 0, sizeof(int), c->stream);
                if (!P.one_product) launch_gemm_rows(R.cp, c->d_Minv, R.tbuf, ld, C, nullptr, nullptr, c->stream);   // M⁻¹pₘ
                DHMC_EXT_NPL(rounds_k2a_dense_external_kernel, dim3(C), ra.P, ra.R)                    // q′ (one product: M⁻¹pₘ = p♯ + (ϵ/2)u)
                if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
                rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
                if (rc) { c->poisoned = true; cleanup(); return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
                DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′
                if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, C, nullptr, nullptr, c->stream);   // u′ = ∇ℓq′·M⁻¹
                else launch_gemm_rows(R.cp, c->d_Minv, R.cps, ld, C, nullptr, nullptr, c->stream);     // p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&done, R.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->external) {
        // round engine with the host's callback as the gradient (external_rounds.hpp)
        RoundArgs ra{P, c->rb};
        e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_logistic_op(0, c->NPL, ra, c->lr, c->stream);                                   // p = W∘z, p♯
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
                launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
                if (c->logistic_batched && c->lr.act) launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);   // the rows of this round
                rc = external_eval(c, c->st.q, true);                                                  // ℓ(q′), ∇ℓ(q′)
                if (rc) { c->poisoned = true; cleanup(); return rc; }   // st.q holds trial positions: see DHMC_CHECK_USABLE
                DHMC_EXT_NPL(rounds_k2_external_kernel, dim3(C), ra.P, ra.R, c->lr)                    // evaluate_ℓ, p′, p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&done, c->rb.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->logistic_rounds) {
        // GEMM-gradient round engine (logistic_rounds.hpp)
        RoundArgs ra{P, c->rb};
        const int ld = c->Dpad;
        const int npad = (int)c->tp.npad;
        e = hipMemsetAsync(c->rb.list_count, 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess) { rc = dispatch(c, Op::RoundStart, &ra); if (rc) { cleanup(); return rc; } }
        unsigned long long rounds = 0;
        int done = 0;
        // the done-counter is read through page-locked memory one batch of four rounds behind (as in the dense engine below): the
        // host never drains the stream inside the loop; the rounds enqueued after the last chain finished find no chain in a leaf
        // phase and an empty row list
        if (!c->h_done && e == hipSuccess) {
            e = hipHostMalloc((void**)&c->h_done, 2 * 8 * sizeof(int), hipHostMallocDefault);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming);
        }
        long long batch = 0;
        while (e == hipSuccess && done < C) {
            for (int rep = 0; rep < 4 && e == hipSuccess; ++rep, ++rounds) {
                launch_logistic_op(0, c->NPL, ra, c->lr, c->stream);                                   // p = W∘z, p♯
                if ((rc = dispatch(c, Op::RoundK0, &ra))) { cleanup(); return rc; }
                e = hipMemsetAsync(c->rb.list_count, 0, sizeof(int), c->stream);
                launch_logistic_op(1, c->NPL, ra, c->lr, c->stream);                                   // q′
                launch_logistic_op(4, c->NPL, ra, c->lr, c->stream);                                   // the rows of this round
                launch_gemm_list(c->st.q, ld, c->tp.b, npad, c->lr.H, npad, C, ld, npad, c->lr.act, c->lr.act_count, c->stream);   // η = Q′·Xᵀ
                launch_logistic_op(2, c->NPL, ra, c->lr, c->stream);                                   // r, S₁
                launch_gemm_splitk(c->lr.H, npad, c->tp.a, ld, c->lr.P, ld, (size_t)C * ld, C, npad, ld, DHMC_LOGISTIC_BLOCK,
                                   c->lr.act, c->lr.act_count, c->stream);                             // Xᵀr = R·X, block by block
                launch_logistic_op(3, c->NPL, ra, c->lr, c->stream);                                   // ∇ℓ, ℓ, p′, p♯
                if ((rc = dispatch(c, Op::RoundK3, &ra))) { cleanup(); return rc; }
            }
            if (e == hipSuccess) e = hipGetLastError();
            int* slot = c->h_done + 8 * (batch & 1);
            if (e == hipSuccess) e = hipMemcpyAsync(slot, c->rb.done_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_done[batch & 1], c->stream);
            if (batch >= 1 && e == hipSuccess) {
                e = hipEventSynchronize(c->ev_done[(batch - 1) & 1]);
                done = c->h_done[8 * ((batch - 1) & 1)];
            }
            batch += 1;
        }
        c->last_rounds = rounds;
    } else if (e == hipSuccess && c->cfg.metric == DHMC_METRIC_DENSE && c->dense_rounds) {
        // (both dense engines run either recurrence with the same bits, so which one serves a context is a matter of speed only)
        P.one_product = c->dense_products == 1;
        P.fuse_k2 = c->fuse_k2;
        // Round-based dense engine (dense_rounds.hpp): every round is one leapfrog for every chain.  The chains
        // run as two half-batches on two streams so that one half's HBM-bound tree kernel overlaps the other
        // half's MFMA-bound contractions.
        const int ld = c->Dpad;
        const int nh = (C >= 256 && C % c->dense_parts == 0) ? c->dense_parts : 1;
        struct Half { RoundArgs ra; hipStream_t s; int base, count; } H[4];
        c->streams[0] = c->stream; c->streams[1] = c->stream2;
        for (int h = 0; h < nh; ++h) {
            H[h].base = h * (C / nh);
            H[h].count = (h == nh - 1) ? C - H[h].base : C / nh;
            c->rbp[h] = c->rb;
            c->rbp[h].list = c->rb.list + H[h].base;
            c->rbp[h].list_count = c->rb.list_count + 2 * h;
            c->rbp[h].done_count = c->rb.list_count + 2 * h + 1;
            H[h].ra = RoundArgs{P, c->rbp[h]};
            H[h].ra.P.chain_base = H[h].base;
            H[h].ra.P.C = H[h].count;
            H[h].s = c->streams[h];
        }
        e = hipMemsetAsync(c->rb.list_count, 0, 8 * sizeof(int), c->stream);
        if (nh >= 2 && e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
        for (int h = 1; h < nh && e == hipSuccess; ++h) e = hipStreamWaitEvent(c->streams[h], c->ev_fork, 0);
        for (int h = 0; h < nh && e == hipSuccess; ++h)
            if ((rc = dispatch(c, Op::RoundStart, &H[h].ra, H[h].s, true))) { cleanup(); return rc; }
        unsigned long long rounds = 0;
        int done[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // Four rounds of both half-batches = 64 launches on two streams, the same for the whole call.  DHMC_GRAPH=1
        // captures them ONCE into a graph (fork to stream2, join back) and launches the graph until every chain is done:
        // one host call per four rounds instead of 64.  Measured on config 3: 8.05e6 leapfrog-steps/s against 8.84e6 with
        // plain launches — the host was never the limit, and the graph's branches overlap less than two free-running
        // streams do — so it is off by default.
        constexpr int REPS = 4;
        // Once the first chains have finished their transitions (the host sees the done-counters every REPS rounds), the
        // two products of a round are taken over the rows of the chains still running only (a row list per part, rebuilt
        // every round): a call ends when its slowest chain does, and until then every round multiplied all rows.  While
        // every chain is running — BASELINE config 3's equal trees from start to end — nothing changes.
        bool row_lists = false;
        auto enqueue_reps = [&]() -> int {
            for (int rep = 0; rep < REPS && e == hipSuccess; ++rep) {
                for (int h = 0; h < nh && e == hipSuccess; ++h) {
                    const RoundBuffers& R = H[h].ra.R;
                    const size_t off = (size_t)H[h].base * ld;
                    hipStream_t s = H[h].s;
                    launch_gemm_rows(R.cp, c->d_WT, R.tbuf, ld, H[h].count, R.list, R.list_count, s);               // p₀ = z·Wᵀ
                    launch_gemm_rows(R.tbuf, c->d_Minv, R.cps, ld, H[h].count, R.list, R.list_count, s);            // p♯₀
                    if (P.one_product) launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, H[h].count, R.list, R.list_count, s);   // u₀ = ∇ℓq₀·M⁻¹
                    if (int r = dispatch(c, Op::RoundK0, &H[h].ra, s, true)) return r;
                    e = hipMemsetAsync(R.list_count, 0, sizeof(int), s);
                    if (P.one_product && !row_lists) {
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;                          // M⁻¹pₘ = p♯ + (ϵ/2)u
                        launch_gemm_rows(c->st.g + off, c->d_Minv, R.cu + off, ld, H[h].count, nullptr, nullptr, s); // u′ = ∇ℓq′·M⁻¹
                    } else if (P.one_product) {
                        LogisticRound L = c->lr;
                        L.act = c->lr.act + H[h].base;
                        L.act_count = c->lr.act + C + h;
                        if (e == hipSuccess) e = hipMemsetAsync(L.act_count, 0, sizeof(int), s);
                        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((H[h].count + 255) / 256), dim3(256), 0, s, H[h].ra.P, R, L);
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(c->st.g, c->d_Minv, R.cu, ld, H[h].count, L.act, L.act_count, s);          // u′
                    } else if (!row_lists) {
                        launch_gemm_rows(R.cp + off, c->d_Minv, R.tbuf + off, ld, H[h].count, nullptr, nullptr, s); // M⁻¹pₘ
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(R.cp + off, c->d_Minv, R.cps + off, ld, H[h].count, nullptr, nullptr, s);  // p♯
                    } else {
                        LogisticRound L = c->lr;
                        L.act = c->lr.act + H[h].base;
                        L.act_count = c->lr.act + C + h;
                        if (e == hipSuccess) e = hipMemsetAsync(L.act_count, 0, sizeof(int), s);
                        hipLaunchKernelGGL(rounds_active_list_kernel, dim3((H[h].count + 255) / 256), dim3(256), 0, s, H[h].ra.P, R, L);
                        launch_gemm_rows(R.cp, c->d_Minv, R.tbuf, ld, H[h].count, L.act, L.act_count, s);           // M⁻¹pₘ
                        if (int r = dispatch(c, Op::RoundK2, &H[h].ra, s, true)) return r;
                        launch_gemm_rows(R.cp, c->d_Minv, R.cps, ld, H[h].count, L.act, L.act_count, s);            // p♯
                    }
                    if (int r = dispatch(c, Op::RoundK3, &H[h].ra, s, true)) return r;
                }
            }
            return DHMC_OK;
        };
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        // (not on the legacy default stream, which cannot be captured: hosts that hand over stream 0 get plain launches)
        if (c->use_graph && c->stream != nullptr && e == hipSuccess &&
            hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            if (nh >= 2 && e == hipSuccess) e = hipEventRecord(c->ev_fork, c->stream);
            for (int h = 1; h < nh && e == hipSuccess; ++h) e = hipStreamWaitEvent(c->streams[h], c->ev_fork, 0);
            if (e == hipSuccess) rc = enqueue_reps();
            for (int h = 1; h < nh && e == hipSuccess; ++h) {
                e = hipEventRecord(c->ev_joins[h], c->streams[h]);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_joins[h], 0);
            }
            hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess && !rc) e = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
            if (rc || e != hipSuccess) {
                if (gexec) (void)hipGraphExecDestroy(gexec);
                if (graph) (void)hipGraphDestroy(graph);
                cleanup();
                if (rc) return rc;
                c->err = std::string("dhmc_run (graph capture): ") + hipGetErrorString(e);
                return DHMC_ERR_HIP;
            }
        }
        // The host never drains the streams to look at the done-counters: after every batch of REPS rounds they are copied
        // into page-locked memory behind an event, and the host reads the PREVIOUS batch's copy once the next batch is
        // enqueued.  So it runs one batch ahead; the (at most REPS) rounds enqueued after the last chain finished find no
        // chain in a leaf phase and, with the row lists, no rows to multiply.
        if (!c->h_done && e == hipSuccess) {
            e = hipHostMalloc((void**)&c->h_done, 2 * 8 * sizeof(int), hipHostMallocDefault);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming);
        }
        long long batch = 0;
        while (e == hipSuccess && done[1] + done[3] + done[5] + done[7] < C) {
            if (gexec) {
                e = hipGraphLaunch(gexec, c->stream);
            } else {
                if ((rc = enqueue_reps())) { cleanup(); return rc; }
                if (e == hipSuccess) e = hipGetLastError();
                for (int h = 1; h < nh && e == hipSuccess; ++h) {
                    e = hipEventRecord(c->ev_joins[h], c->streams[h]);
                    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_joins[h], 0);
                }
            }
            rounds += REPS;
            int* slot = c->h_done + 8 * (batch & 1);
            if (e == hipSuccess) e = hipMemcpyAsync(slot, c->rb.list_count, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_done[batch & 1], c->stream);
            if (batch >= 1 && e == hipSuccess) {
                e = hipEventSynchronize(c->ev_done[(batch - 1) & 1]);
                std::memcpy(done, c->h_done + 8 * ((batch - 1) & 1), 8 * sizeof(int));
            }
            batch += 1;
            row_lists = c->dense_row_lists && !gexec && done[1] + done[3] + done[5] + done[7] > 0;
        }
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        c->last_rounds = rounds;
    } else if (e == hipSuccess && nbuf == 2) {
        // the one-kernel engine, host outputs, in chunks: kernel of chunk k ‖ copy of chunk k-1
        const int64_t nchunks = (N + L - 1) / L;
        bool used[2] = {false, false};
        for (int64_t k = 0; k < nchunks && e == hipSuccess; ++k) {
            const int b = (int)(k & 1);
            const int64_t n0 = k * L, len = (n0 + L <= N) ? L : N - n0;
            RunParams Q = P;
            Q.N = len;
            Q.out_stride = L;
            if (da) { Q.da_init = (k == 0) ? da->init : 0; Q.da_finalize = (k == nchunks - 1) ? da->finalize : 0; }
            for (auto& f : staged) *f.dev = c->stage[b][f.idx].p;          // (the slots are fields of P.out: copy them again)
            Q.out = P.out;
            if (used[b]) {                                                   // buffer b: its previous copy has left, and its kernel time is known
                e = hipStreamWaitEvent(c->stream, c->ev_copy[b], 0);
                if (e == hipSuccess) e = hipEventSynchronize(c->ev_k1[b]);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_k0[b], c->ev_k1[b]);
                chunk_ms += ms;
            }
            if (e == hipSuccess) e = hipEventRecord(c->ev_k0[b], c->stream);
            if (e == hipSuccess && (rc = dispatch(c, Op::Run, &Q))) { (void)hipDeviceSynchronize(); cleanup(); return rc; }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(c->ev_k1[b], c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->ev_k1[b], 0);
            if (e == hipSuccess) e = d2h(b, n0, len, c->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(c->ev_copy[b], c->copy_stream);
            used[b] = true;
        }
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            if (used[b]) {
                e = hipEventSynchronize(c->ev_k1[b]);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev_k0[b], c->ev_k1[b]);
                chunk_ms += ms;
            }
        if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
    } else if (e == hipSuccess) {
        rc = dispatch(c, Op::Run, &P);
        if (rc) { cleanup(); return rc; }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess && nbuf == 1 && !staged.empty()) e = d2h(0, 0, N, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&c->last_leapfrogs, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->last_ms = nbuf == 2 ? chunk_ms : ms;     // kernel time only: not the waits for copies between the chunks
    }
    cleanup();
    if (e != hipSuccess) {
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);     // no copy into the caller's arrays may outlive the call
        c->err = std::string("dhmc_run: ") + hipGetErrorString(e);
        return DHMC_ERR_HIP;
    }
    // staging of a call that could not be chunked (round engines: [C][N] records at once) is given back when it is large;
    // the chunked engine's two buffers (≤ ≈1 GiB of draws each) stay with the context
    if (nbuf == 1 && !staged.empty()) {
        size_t held = 0;
        for (auto& sb : c->stage[0]) held += sb.cap;
        if (held > ((size_t)1 << 30))
            for (auto& sb : c->stage[0])
                if (sb.p) { (void)hipFree(sb.p); sb.p = nullptr; sb.cap = 0; }
    }
    return status_code(c);
}
}  // namespace

// The round engines (dense metric, GEMM-shaped gradients, external models) advance their chains round by round, so a call cannot
// hand out its first transitions while it computes the last ones.  With HOST outputs of more than ≈ 2 GiB of draws the call is
// therefore run as several calls of L transitions (the chains resume where they stand: the same transitions, the same bits) into
// two device staging buffers of ≈ 1 GiB, and chunk k leaves over the copy stream while chunk k + 1 computes — what the diagonal
// engine does inside one call (run_call).  Dual averaging: initialised by the first chunk, finalised by the last.
int dhmc_run(dhmc_ctx* c, int64_t N, const dhmc_dual_averaging* da, const dhmc_outputs* out) {
    if (!c || N < 0) return DHMC_ERR_INVALID_ARGUMENT;
    const bool one_kernel = c->cfg.metric == DHMC_METRIC_DIAG && !c->logistic_rounds && !c->external;
    const bool host_draws = out && !out->on_device && out->draws;
    const int64_t per_transition = (int64_t)c->cfg.chains * c->cfg.dim * (int64_t)sizeof(double);
    int64_t L = N;
    if (host_draws && !one_kernel && N > 1) {
        if (c->host_chunk > 0) L = std::min<int64_t>(c->host_chunk, N);
        else if (per_transition * N > ((int64_t)2 << 30)) L = std::max<int64_t>(1, ((int64_t)1 << 30) / per_transition);
    }
    if (L >= N) return run_call(c, N, da, out);

    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    struct F { void* host; size_t elem; int idx; };
    const F fields[10] = {{out->draws, (size_t)c->cfg.dim * sizeof(double), 0}, {out->logdensities, sizeof(double), 1}, {out->eps, sizeof(double), 2},
                          {out->pi, sizeof(double), 3}, {out->acceptance_rate, sizeof(double), 4}, {out->steps, sizeof(int64_t), 5},
                          {out->term_left, sizeof(int64_t), 6}, {out->term_right, sizeof(int64_t), 7}, {out->depth, sizeof(int32_t), 8},
                          {out->directions, sizeof(uint32_t), 9}};
    const size_t C = (size_t)c->cfg.chains;
    for (const F& f : fields) {
        if (!f.host) continue;
        for (int b = 0; b < 2; ++b) {
            auto& sb = c->stage[b][f.idx];
            const size_t need = C * (size_t)L * f.elem;
            if (sb.cap < need) {
                if (sb.p) { HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipFree(sb.p)); sb.p = nullptr; sb.cap = 0; }
                HIP_TRY(c, hipMalloc(&sb.p, need));
                sb.cap = need;
            }
        }
    }
    if (!c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(c, hipEventCreate(&c->ev_k0[b]));
            HIP_TRY(c, hipEventCreate(&c->ev_k1[b]));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copy[b], hipEventDisableTiming));
        }
    }
    const int64_t nchunks = (N + L - 1) / L;
    double ms = 0.0;
    unsigned long long leapfrogs = 0, rounds = 0;
    int rc = DHMC_OK;
    bool used[2] = {false, false};
    // Every exit drains the copy stream: asynchronous copies into the caller's arrays must not outlive the call (the caller —
    // numpy, Julia's GC — may free them the moment it sees an error code).
    auto chunks = [&]() -> int {
        for (int64_t k = 0; k < nchunks; ++k) {
            const int b = (int)(k & 1);
            const int64_t n0 = k * L, len = std::min(L, N - n0);
            if (used[b]) HIP_TRY(c, hipEventSynchronize(c->ev_copy[b]));            // staging buffer b is free again
            dhmc_outputs dev{};
            dev.on_device = 1;
            void** slots[10] = {(void**)&dev.draws, (void**)&dev.logdensities, (void**)&dev.eps, (void**)&dev.pi, (void**)&dev.acceptance_rate,
                                (void**)&dev.steps, (void**)&dev.term_left, (void**)&dev.term_right, (void**)&dev.depth, (void**)&dev.directions};
            for (const F& f : fields)
                if (f.host) *slots[f.idx] = c->stage[b][f.idx].p;
            dhmc_dual_averaging dk{};
            if (da) { dk = *da; dk.init = (k == 0) ? da->init : 0; dk.finalize = (k == nchunks - 1) ? da->finalize : 0; }
            const int r = run_call(c, len, da ? &dk : nullptr, &dev);              // returns with the stream drained
            ms += c->last_ms; leapfrogs += c->last_leapfrogs; rounds += c->last_rounds;
            if (r != DHMC_OK && r != DHMC_ERR_CHAIN_FAILURE) return r;
            if (r != DHMC_OK) rc = r;                                              // (a failed chain: the call goes on, as one call would)
            for (const F& f : fields) {
                if (!f.host) continue;
                HIP_TRY(c, hipMemcpy2DAsync((char*)f.host + (size_t)n0 * f.elem, (size_t)N * f.elem, c->stage[b][f.idx].p, (size_t)len * f.elem,
                                            (size_t)len * f.elem, C, hipMemcpyDeviceToHost, c->copy_stream));
            }
            HIP_TRY(c, hipEventRecord(c->ev_copy[b], c->copy_stream));
            used[b] = true;
        }
        return DHMC_OK;
    };
    const int lr = chunks();
    const hipError_t se = hipStreamSynchronize(c->copy_stream);
    if (lr != DHMC_OK) return lr;
    HIP_TRY(c, se);
    c->last_ms = ms; c->last_leapfrogs = leapfrogs; c->last_rounds = rounds;
    return rc;
}

int dhmc_update_metric_diag(dhmc_ctx* c, const double* draws, int64_t n, double lambda, int on_device) {
    if (!c || !draws || c->cfg.metric != DHMC_METRIC_DIAG) return DHMC_ERR_INVALID_ARGUMENT;
    if (n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:191-192 (N >= 20 is the host wrapper's check)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    Staged s;
    int rc = stage_in(c, draws, sizeof(double) * (size_t)c->cfg.chains * n * c->cfg.dim, on_device, &s);
    if (rc) return rc;
    launch_metric(c, (const double*)s.dev, n);
    HIP_TRY(c, hipGetLastError());
    stage_free(c, &s);
    return DHMC_OK;
}

int dhmc_update_metric_dense(dhmc_ctx* c, const double* draws, int64_t n, double lambda, int on_device) {
    if (!c || !draws || c->cfg.metric != DHMC_METRIC_DENSE) return DHMC_ERR_INVALID_ARGUMENT;
    if (n < 2 || !(lambda >= 0)) return DHMC_ERR_INVALID_ARGUMENT;  // mcmc.jl:191-192
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int D = c->cfg.dim, ld = c->Dpad;
    // shared M⁻¹: one estimate from the pooled draws of all chains; per-chain: every chain from its own n draws (mcmc.jl:281-285)
    const int nest = c->per_chain_dense ? c->cfg.chains : 1;
    const int64_t J = c->per_chain_dense ? n : (int64_t)c->cfg.chains * n;
    Staged s;
    int rc = stage_in(c, draws, sizeof(double) * (size_t)c->cfg.chains * n * D, on_device, &s);
    if (rc) return rc;
    int refused = 0, first_refused = -1;
    if (!c->per_chain_dense) {
        DevBuf bmean, bS;
        HIP_TRY(c, hipMalloc(&bmean.p, sizeof(double) * ld));
        HIP_TRY(c, hipMalloc(&bS.p, sizeof(double) * (size_t)ld * ld));
        double* const mean = (double*)bmean.p;
        double* const S = (double*)bS.p;
        const double* x = (const double*)s.dev;
        double Jtot = (double)J;
        if (!c->metric_allreduce) {
            hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, J, x, mean, (size_t)0, (size_t)0, 0);
            hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)0, (size_t)0, (size_t)0);
        } else {
            // job-wide estimate (include/dhmc.h dhmc_set_metric_allreduce): column sums + row count over the ranks, then the
            // scatter about the job's mean over the ranks; `mean` has Dpad >= D + 1 slots except when D is a multiple of 64
            DevBuf bsum;
            HIP_TRY(c, hipMalloc(&bsum.p, sizeof(double) * (size_t)(D + 1)));
            double* const sums = (double*)bsum.p;
            hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, J, x, sums, (size_t)0, (size_t)0, 1);
            HIP_TRY(c, hipMemcpyAsync(sums + D, &Jtot, sizeof(double), hipMemcpyHostToDevice, c->stream));
            if (c->metric_allreduce(c->metric_allreduce_user, sums, (int64_t)D + 1, (void*)c->stream) != 0) {
                c->err = "dhmc_update_metric_dense: the all-reduce callback failed (column sums)"; stage_free(c, &s); return DHMC_ERR_CALLBACK;
            }
            HIP_TRY(c, hipMemcpyAsync(&Jtot, sums + D, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            hipLaunchKernelGGL(pooled_mean_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, c->stream, D, sums);
            HIP_TRY(c, hipMemcpyAsync(mean, sums, sizeof(double) * (size_t)D, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)0, (size_t)0, (size_t)0);
            if (c->metric_allreduce(c->metric_allreduce_user, S, (int64_t)ld * ld, (void*)c->stream) != 0) {
                c->err = "dhmc_update_metric_dense: the all-reduce callback failed (scatter matrix)"; stage_free(c, &s); return DHMC_ERR_CALLBACK;
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));       // Jtot is on the host now
            if (!(Jtot >= 2.0)) { stage_free(c, &s); return DHMC_ERR_INVALID_ARGUMENT; }
        }
        hipLaunchKernelGGL(cov_regularize_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256)), dim3(256), 0, c->stream, D, ld, (int64_t)Jtot, lambda, S, (size_t)0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { c->err = std::string("dhmc_update_metric_dense: ") + hipGetErrorString(e); rc = DHMC_ERR_HIP; }
        else rc = device_dense_metric(c, S, ld, -1);   // DHMC_ERR_INVALID_ARGUMENT: the estimate is not positive definite
    } else {
        // per-chain metrics (mcmc.jl:281-285 runs per chain): estimate, regularise and factorise a BATCH of chains per launch
        // (blockIdx.z = chain; the same kernels, so the same bits as chain by chain), ≈ 1 GiB of work space at a time.  Every chain
        // stands for itself: one whose estimate is refused keeps its metric, the others are updated all the same.
        const size_t n = (size_t)ld * ld;
        const int Bmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)nest, ((size_t)1 << 30) / (5 * n * sizeof(double))));
        DevBuf bmean, bS, bwork, bflags;
        HIP_TRY(c, hipMalloc(&bmean.p, sizeof(double) * (size_t)Bmax * ld));
        HIP_TRY(c, hipMalloc(&bS.p, sizeof(double) * (size_t)Bmax * n * 2));          // the estimates, and Symmetric(estimate)
        HIP_TRY(c, hipMalloc(&bwork.p, sizeof(double) * (size_t)Bmax * n * 3));
        HIP_TRY(c, hipMalloc(&bflags.p, sizeof(int) * 2 * (size_t)Bmax));
        double* const mean = (double*)bmean.p;
        double* const S = (double*)bS.p;
        double* const Ssym = S + (size_t)Bmax * n;
        double* const work = (double*)bwork.p;
        int* const flags = (int*)bflags.p;
        std::vector<int> hflags(2 * (size_t)Bmax);
        for (int k0 = 0; k0 < nest && rc == DHMC_OK; k0 += Bmax) {
            const int B = std::min(Bmax, nest - k0);
            const unsigned Bz = (unsigned)B;
            const double* x = (const double*)s.dev + (size_t)k0 * J * D;
            HIP_TRY(c, hipMemsetAsync(flags, 0, sizeof(int) * 2 * (size_t)B, c->stream));
            hipLaunchKernelGGL(pooled_mean_kernel, dim3((D + 255) / 256, 1, Bz), dim3(256), 0, c->stream, D, J, x, mean, (size_t)J * D, (size_t)ld, 0);
            hipLaunchKernelGGL(pooled_cov_kernel, dim3(ld / 64, ld / 64, Bz), dim3(256), 0, c->stream, D, J, x, mean, S, ld, (size_t)J * D, (size_t)ld, n);
            hipLaunchKernelGGL(cov_regularize_kernel, dim3((unsigned)(((size_t)D * D + 255) / 256), 1, Bz), dim3(256), 0, c->stream, D, ld, J, lambda, S, n);
            hipLaunchKernelGGL(df_symmetrize_kernel, dim3((unsigned)((n + 255) / 256), 1, Bz), dim3(256), 0, c->stream, (const double*)S, ld, D, Ssym, ld,
                               flags, n, n);
            double* const WTb = work + (size_t)B * n;                                  // the X buffers: free again once M = XᵀX exists
            df_dense_metric(Ssym, Ssym, WTb, D, ld, work, flags, c->stream, B);
            hipLaunchKernelGGL(df_commit_kernel, dim3((unsigned)((n + 255) / 256), 1, Bz), dim3(256), 0, c->stream, (const double*)Ssym, (const double*)WTb,
                               (const int*)flags, c->d_Minv + (size_t)k0 * n, c->d_WT + (size_t)k0 * n, n);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { c->err = std::string("dhmc_update_metric_dense: ") + hipGetErrorString(e); rc = DHMC_ERR_HIP; break; }
            HIP_TRY(c, hipMemcpyAsync(hflags.data(), flags, sizeof(int) * 2 * (size_t)B, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (int b2 = 0; b2 < B; ++b2)
                if (hflags[2 * b2] || hflags[2 * b2 + 1]) { if (refused++ == 0) first_refused = k0 + b2; }
        }
    }
    stage_free(c, &s);
    if (rc == DHMC_OK && refused) {
        c->err = "dhmc_update_metric_dense: the covariance estimate of " + std::to_string(refused) + " chain(s) (first: chain " +
                 std::to_string(first_refused) + ") is not finite / positive definite; those chains keep their metric";
        return DHMC_ERR_INVALID_ARGUMENT;
    }
    return rc;
}

int dhmc_set_metric_allreduce(dhmc_ctx* c, dhmc_allreduce_fn fn, void* user) {
    if (!c || c->cfg.metric != DHMC_METRIC_DENSE || c->per_chain_dense) return DHMC_ERR_INVALID_ARGUMENT;   // a shared dense metric is what is pooled
    c->metric_allreduce = fn;
    c->metric_allreduce_user = fn ? user : nullptr;
    return DHMC_OK;
}

// ---- resume blob: header + raw images of the per-chain arrays -------------------------------
struct BlobHeader {
    uint64_t magic;
    int32_t dim, chains, Dpad, reserved;
};
static const uint64_t BLOB_MAGIC = 0x31434d4844ull;  // "DHMC1"

// ---- Diagnostics probes (probe_kernels.hpp) ---------------------------------------------------
namespace {
int probe_finish(dhmc_ctx* c, const DevBuf& dst, uint32_t* status) {
    const int C = c->cfg.chains;
    std::vector<uint32_t> st(C);
    HIP_TRY(c, hipMemcpyAsync(st.data(), dst.p, sizeof(uint32_t) * C, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int rc = DHMC_OK;
    for (int i = 0; i < C; ++i) {
        if (status) status[i] = st[i];
        if (st[i]) rc = DHMC_ERR_CHAIN_FAILURE;
    }
    return rc;
}

// the same two probes for a model evaluated by the host's callback (external_rounds.hpp): lock-step leapfrogs of all chains
struct ExtProbe {
    DevBuf rows[8], pi0, lq, alive;
    ExtProbeParams E{};
    dhmc_ctx* c = nullptr;
    int init(dhmc_ctx* ctx, uint32_t* d_status) {
        c = ctx;
        const size_t C = c->cfg.chains, n = C * c->Dpad;
        for (DevBuf& b : rows) {
            HIP_TRY(c, hipMalloc(&b.p, sizeof(double) * n));
            HIP_TRY(c, hipMemsetAsync(b.p, 0, sizeof(double) * n, c->stream));
        }
        HIP_TRY(c, hipMalloc(&pi0.p, sizeof(double) * C));
        HIP_TRY(c, hipMalloc(&lq.p, sizeof(double) * C));
        HIP_TRY(c, hipMalloc(&alive.p, sizeof(int32_t) * C));
        HIP_TRY(c, hipMemsetAsync(d_status, 0, sizeof(uint32_t) * C, c->stream));
        E.D = c->cfg.dim; E.Dpad = c->Dpad; E.C = (int)C; E.chain_offset = c->cfg.chain_offset; E.seed = c->cfg.seed; E.st = c->st;
        E.q = (double*)rows[0].p; E.p = (double*)rows[1].p; E.g = (double*)rows[2].p; E.pm = (double*)rows[3].p;
        E.trial = (double*)rows[4].p; E.ps = (double*)rows[5].p; E.p0 = (double*)rows[6].p; E.ps0 = (double*)rows[7].p;
        E.pi0 = (double*)pi0.p; E.lq_cur = (double*)lq.p; E.alive = (int32_t*)alive.p; E.status = d_status;
        E.lq_in = c->lr.S1; E.grad_in = c->rb.tbuf; E.dense = c->cfg.metric == DHMC_METRIC_DENSE;
        return DHMC_OK;
    }
    // p₀ (the caller's m-th momentum, or rand_p from the chains' streams), p₀♯, π₀
    int momentum(const double* d_p_in, int n_mom, int m, uint32_t momentum_index, bool ratios) {
        const int C = E.C, ld = E.Dpad;
        if (E.dense && !d_p_in) {                      // z into pₘ's row (free here), p₀ = z·Wᵀ
            ExtProbeParams Z = E;
            Z.p0 = E.pm;
            DHMC_EXT_NPL(ext_probe_momentum_kernel, dim3(C), Z, d_p_in, n_mom, m, momentum_index)
            launch_gemm_rows(E.pm, c->d_WT, E.p0, ld, C, nullptr, nullptr, c->stream);
        } else {
            DHMC_EXT_NPL(ext_probe_momentum_kernel, dim3(C), E, d_p_in, n_mom, m, momentum_index)
        }
        if (E.dense) launch_gemm_rows(E.p0, c->d_Minv, E.ps0, ld, C, nullptr, nullptr, c->stream);
        DHMC_EXT_NPL(ext_probe_start_kernel, dim3(C), E, (int)ratios)
        return DHMC_OK;
    }
    void restart(bool ratios) { DHMC_EXT_NPL(ext_probe_restart_kernel, dim3(E.C), E, (int)ratios) }
    int step(double eps) {                             // one leapfrog of every chain that still steps
        const int C = E.C, ld = E.Dpad;
        DHMC_EXT_NPL(ext_probe_half_kernel, dim3(C), E, eps)
        if (E.dense) {
            launch_gemm_rows(E.pm, c->d_Minv, E.ps, ld, C, nullptr, nullptr, c->stream);
            DHMC_EXT_NPL(ext_probe_pos_kernel, dim3(C), E, eps)
        }
        if (int rc = external_eval(c, E.trial)) return rc;
        DHMC_EXT_NPL(ext_probe_finish_kernel, dim3(C), E, eps)
        if (E.dense) launch_gemm_rows(E.p, c->d_Minv, E.ps, ld, C, nullptr, nullptr, c->stream);
        return DHMC_OK;
    }
    void record(int idx, int npos, int pos, bool start, double* od, double* ol, double* oq, double* op, int32_t* orange) {
        DHMC_EXT_NPL(ext_probe_record_kernel, dim3(E.C), E, idx, npos, pos, (int)start, od, ol, oq, op, orange)
    }
};
}  // namespace

int dhmc_leapfrog_trajectory(dhmc_ctx* c, double eps, int32_t first, int32_t last, uint32_t momentum_index,
                             const double* p, double* delta, double* logdensity, double* q_out, double* p_out,
                             int32_t* range, uint32_t* status) {
    if (!c || !delta || !logdensity) return DHMC_ERR_INVALID_ARGUMENT;
    if (!(first <= 0 && 0 <= last)) return DHMC_ERR_INVALID_ARGUMENT;   // diagnostics.jl:218
    if (c->external) DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains, D = c->cfg.dim;
    const size_t npos = (size_t)last - first + 1;
    DevBuf dp, dd, dl, dq, dpo, dr, dst;
    HIP_TRY(c, hipMalloc(&dd.p, sizeof(double) * C * npos));
    HIP_TRY(c, hipMalloc(&dl.p, sizeof(double) * C * npos));
    HIP_TRY(c, hipMalloc(&dr.p, sizeof(int32_t) * 2 * C));
    HIP_TRY(c, hipMalloc(&dst.p, sizeof(uint32_t) * C));
    // positions that are not visited stay NaN (all-ones bit pattern)
    HIP_TRY(c, hipMemsetAsync(dd.p, 0xFF, sizeof(double) * C * npos, c->stream));
    HIP_TRY(c, hipMemsetAsync(dl.p, 0xFF, sizeof(double) * C * npos, c->stream));
    if (q_out) {
        HIP_TRY(c, hipMalloc(&dq.p, sizeof(double) * C * npos * D));
        HIP_TRY(c, hipMemsetAsync(dq.p, 0xFF, sizeof(double) * C * npos * D, c->stream));
    }
    if (p_out) {
        HIP_TRY(c, hipMalloc(&dpo.p, sizeof(double) * C * npos * D));
        HIP_TRY(c, hipMemsetAsync(dpo.p, 0xFF, sizeof(double) * C * npos * D, c->stream));
    }
    if (p) {
        HIP_TRY(c, hipMalloc(&dp.p, sizeof(double) * C * D));
        HIP_TRY(c, hipMemcpyAsync(dp.p, p, sizeof(double) * C * D, hipMemcpyHostToDevice, c->stream));
    }
    if (c->external || c->logistic_batched) {   // the density is evaluated for all chains between kernels: one batched evaluation per step
        ExtProbe X;
        int rc;
        if ((rc = X.init(c, (uint32_t*)dst.p))) return rc;
        HIP_TRY(c, hipMemsetAsync(dr.p, 0, sizeof(int32_t) * 2 * C, c->stream));
        if ((rc = X.momentum((const double*)dp.p, 1, 0, momentum_index, false))) return rc;
        X.restart(false);
        X.record(-first, (int)npos, 0, true, (double*)dd.p, (double*)dl.p, (double*)dq.p, (double*)dpo.p, (int32_t*)dr.p);
        for (int dir = 0; dir < 2; ++dir) {
            const double e = dir == 0 ? eps : -eps;                          // diagnostics.jl:223,225
            const int count = dir == 0 ? last : -first;
            X.restart(false);
            for (int i = 1; i <= count; ++i) {
                if ((rc = X.step(e))) return rc;
                const int pos = dir == 0 ? i : -i;
                X.record(pos - first, (int)npos, pos, false, (double*)dd.p, (double*)dl.p, (double*)dq.p, (double*)dpo.p, (int32_t*)dr.p);
            }
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(delta, dd.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(logdensity, dl.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
        if (q_out) HIP_TRY(c, hipMemcpyAsync(q_out, dq.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
        if (p_out) HIP_TRY(c, hipMemcpyAsync(p_out, dpo.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
        if (range) HIP_TRY(c, hipMemcpyAsync(range, dr.p, sizeof(int32_t) * 2 * C, hipMemcpyDeviceToHost, c->stream));
        return probe_finish(c, dst, status);
    }
    ProbeParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset; P.seed = c->cfg.seed;
    P.st = c->st; P.tp = c->tp; P.momentum_index = momentum_index; P.p_in = (const double*)dp.p; P.n_mom = 1;
    P.eps = eps; P.first = first; P.last = last;
    P.out_delta = (double*)dd.p; P.out_lq = (double*)dl.p; P.out_q = (double*)dq.p; P.out_p = (double*)dpo.p;
    P.out_range = (int32_t*)dr.p; P.out_status = (uint32_t*)dst.p;
    int rc = dispatch(c, Op::ProbeTrajectory, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(delta, dd.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(logdensity, dl.p, sizeof(double) * C * npos, hipMemcpyDeviceToHost, c->stream));
    if (q_out) HIP_TRY(c, hipMemcpyAsync(q_out, dq.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
    if (p_out) HIP_TRY(c, hipMemcpyAsync(p_out, dpo.p, sizeof(double) * C * npos * D, hipMemcpyDeviceToHost, c->stream));
    if (range) HIP_TRY(c, hipMemcpyAsync(range, dr.p, sizeof(int32_t) * 2 * C, hipMemcpyDeviceToHost, c->stream));
    return probe_finish(c, dst, status);
}

int dhmc_explore_log_acceptance_ratios(dhmc_ctx* c, const double* eps, int32_t n_eps, int32_t n_momenta,
                                       uint32_t momentum_index, const double* ps, double* out, uint32_t* status) {
    if (!c || !eps || !out || n_eps <= 0 || n_momenta <= 0) return DHMC_ERR_INVALID_ARGUMENT;
    if (c->external) DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int C = c->cfg.chains, D = c->cfg.dim;
    const size_t nout = (size_t)C * n_momenta * n_eps;
    DevBuf de, dp, dout, dst;
    HIP_TRY(c, hipMalloc(&de.p, sizeof(double) * n_eps));
    HIP_TRY(c, hipMalloc(&dout.p, sizeof(double) * nout));
    HIP_TRY(c, hipMalloc(&dst.p, sizeof(uint32_t) * C));
    HIP_TRY(c, hipMemcpyAsync(de.p, eps, sizeof(double) * n_eps, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dout.p, 0xFF, sizeof(double) * nout, c->stream));
    if (ps) {
        HIP_TRY(c, hipMalloc(&dp.p, sizeof(double) * C * n_momenta * D));
        HIP_TRY(c, hipMemcpyAsync(dp.p, ps, sizeof(double) * C * n_momenta * D, hipMemcpyHostToDevice, c->stream));
    }
    if (c->external || c->logistic_batched) {
        ExtProbe X;
        int rc;
        if ((rc = X.init(c, (uint32_t*)dst.p))) return rc;
        for (int m = 0; m < n_momenta; ++m) {
            if ((rc = X.momentum((const double*)dp.p, n_momenta, m, momentum_index, true))) return rc;
            for (int e = 0; e < n_eps; ++e) {
                X.restart(true);
                if ((rc = X.step(eps[e]))) return rc;
                X.record(m * n_eps + e, n_momenta * n_eps, 0, false, (double*)dout.p, nullptr, nullptr, nullptr, nullptr);
            }
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(out, dout.p, sizeof(double) * nout, hipMemcpyDeviceToHost, c->stream));
        return probe_finish(c, dst, status);
    }
    ProbeParams P{};
    P.D = D; P.Dpad = c->Dpad; P.C = C; P.chain_offset = c->cfg.chain_offset; P.seed = c->cfg.seed;
    P.st = c->st; P.tp = c->tp; P.momentum_index = momentum_index; P.p_in = (const double*)dp.p; P.n_mom = n_momenta;
    P.eps_list = (const double*)de.p; P.n_eps = n_eps; P.out_delta = (double*)dout.p; P.out_status = (uint32_t*)dst.p;
    int rc = dispatch(c, Op::ProbeRatios, &P);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, sizeof(double) * nout, hipMemcpyDeviceToHost, c->stream));
    return probe_finish(c, dst, status);
}

// ESS and R-hat of `ncoords` series sets laid out as draws[C][n][dim] (ess_kernels.hpp): n <= ESS_LDS_MAX_N with the whole series in LDS,
// longer series from HBM a chunk of lags at a time.  de / dr: device [ncoords].  DHMC_ESS_LONG=1 forces the long path (tests).
namespace {
struct EssWork {
    DevBuf da, dm, dx, dst;
    bool long_series = false;
    int prepare(int64_t chains, int64_t n, int ncoords) {
        const char* e = std::getenv("DHMC_ESS_LONG");
        const bool force_long = e && std::atoi(e) != 0;
        long_series = n > ESS_LDS_MAX_N || force_long;
        const size_t nseries = (size_t)ncoords * chains;
        if (nseries > 0x7fffffffull) return DHMC_ERR_UNSUPPORTED;
        if (hipMalloc(&dm.p, sizeof(double) * nseries) != hipSuccess) return DHMC_ERR_HIP;
        if (!long_series) return hipMalloc(&da.p, sizeof(double) * nseries * n) == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
        if (hipMalloc(&dx.p, sizeof(double) * nseries * n) != hipSuccess || hipMalloc(&da.p, sizeof(double) * nseries * ESS_LAG_CHUNK) != hipSuccess ||
            hipMalloc(&dst.p, sizeof(EssState) * ncoords) != hipSuccess)
            return DHMC_ERR_HIP;
        return DHMC_OK;
    }
};
// the short path only enqueues work on s; the long path returns with the stream drained
int ess_estimate(EssWork& w, hipStream_t s, const double* draws, int64_t chains, int64_t n, int64_t dim, const int32_t* d_coords, int ncoords,
                 double* de, double* dr) {
    if (!w.long_series) {
        hipLaunchKernelGGL(ess_acov_kernel, dim3(ncoords, (unsigned)chains), dim3(ESS_THREADS), sizeof(double) * n, s, draws, n, dim,
                           d_coords, chains, (double*)w.da.p, (double*)w.dm.p);
        hipLaunchKernelGGL(ess_finish_kernel, dim3(ncoords), dim3(ESS_THREADS), sizeof(double) * n, s, (const double*)w.da.p,
                           (const double*)w.dm.p, n, chains, de, dr);
        return hipGetLastError() == hipSuccess ? DHMC_OK : DHMC_ERR_HIP;
    }
    const size_t nseries = (size_t)ncoords * chains;
    hipLaunchKernelGGL(ess_center_kernel, dim3(ncoords, (unsigned)chains), dim3(ESS_THREADS), 0, s, draws, n, dim, d_coords, chains,
                       (double*)w.dx.p, (double*)w.dm.p);
    std::vector<EssState> st(ncoords);
    for (int64_t t0 = 0; t0 < n; t0 += ESS_LAG_CHUNK) {
        hipLaunchKernelGGL(ess_acov_lags_kernel, dim3((unsigned)nseries, ESS_LAG_CHUNK / ESS_THREADS), dim3(ESS_THREADS), 0, s,
                           (const double*)w.dx.p, n, t0, (double*)w.da.p);
        hipLaunchKernelGGL(ess_finish_chunk_kernel, dim3(ncoords), dim3(ESS_THREADS), 0, s, (const double*)w.da.p, (const double*)w.dm.p, n,
                           chains, t0, (EssState*)w.dst.p, de, dr);
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(st.data(), w.dst.p, sizeof(EssState) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
        bool all = true;
        for (const EssState& e : st) all = all && e.done;
        if (all) break;
    }
    return DHMC_OK;
}
}  // namespace

int dhmc_ess_rhat(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat) {
    if (!draws || !coords || !ess || !rhat || chains < 1 || n < 4 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dc, de, dr;
    EssWork work;
    if (int rc = work.prepare(chains, n, ncoords)) return rc;
    if (hipMalloc(&dc.p, sizeof(int32_t) * ncoords) != hipSuccess || hipMalloc(&de.p, sizeof(double) * ncoords) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double) * ncoords) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc.p, coords, sizeof(int32_t) * ncoords, hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    if (int rc = ess_estimate(work, s, draws, chains, n, dim, (const int32_t*)dc.p, ncoords, (double*)de.p, (double*)dr.p)) return rc;
    if (hipMemcpyAsync(ess, de.p, sizeof(double) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipMemcpyAsync(rhat, dr.p, sizeof(double) * ncoords, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    return DHMC_OK;
}

int dhmc_ess_bulk(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess, double* rhat) {
    if (!draws || !coords || !ess || !rhat || chains < 1 || n < 8 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t half = n / 2, N2 = 2 * half, S = chains * N2, C2 = 2 * chains;
    if (S > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dk, dk2, di, di2, dz, de, dr, dtmp, dc0;
    EssWork work;
    if (int rc = work.prepare(C2, half, 1)) return rc;
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
    const int32_t zero = 0;
    if (hipMalloc(&dk.p, sizeof(double) * S) != hipSuccess || hipMalloc(&dk2.p, sizeof(double) * S) != hipSuccess ||
        hipMalloc(&di.p, sizeof(int32_t) * S) != hipSuccess || hipMalloc(&di2.p, sizeof(int32_t) * S) != hipSuccess ||
        hipMalloc(&dz.p, sizeof(double) * S) != hipSuccess || hipMalloc(&de.p, sizeof(double)) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double)) != hipSuccess || hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess ||
        hipMalloc(&dc0.p, sizeof(int32_t)) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc0.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    const unsigned nb = (unsigned)((S + 255) / 256);
    for (int j = 0; j < ncoords; ++j) {
        hipLaunchKernelGGL(ess_gather_kernel, dim3(nb), dim3(256), 0, s, draws, n, dim, coords[j], chains, N2, (double*)dk.p, (int32_t*)di.p);
        if (hipcub::DeviceRadixSort::SortPairs(dtmp.p, tmp_bytes, (const double*)dk.p, (double*)dk2.p, (const int32_t*)di.p,
                                               (int32_t*)di2.p, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
        hipLaunchKernelGGL(ess_rank_kernel, dim3(nb), dim3(256), 0, s, (const double*)dk2.p, (const int32_t*)di2.p, S, (double*)dz.p);
        // z is [2C][N'][1]: the estimator of dhmc_ess_rhat on one "coordinate"
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (int rc = ess_estimate(work, s, (const double*)dz.p, C2, half, 1, (const int32_t*)dc0.p, 1, (double*)de.p, (double*)dr.p)) return rc;
        if (hipMemcpyAsync(ess + j, de.p, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(rhat + j, dr.p, sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    return DHMC_OK;
}

int dhmc_ess_tail(int32_t device, void* stream, const double* draws, int64_t chains, int64_t n, int64_t dim,
                  const int32_t* coords, int32_t ncoords, double* ess) {
    if (!draws || !coords || !ess || chains < 1 || n < 8 || dim < 1 || ncoords < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t half = n / 2, N2 = 2 * half, S = chains * N2, C2 = 2 * chains;
    if (S > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    for (int i = 0; i < ncoords; ++i)
        if (coords[i] < 0 || coords[i] >= dim) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf dk, dk2, di, di2, dz, de, dr, dtmp, dc0;
    EssWork work;
    if (int rc = work.prepare(C2, half, 1)) return rc;
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
    const int32_t zero = 0;
    if (hipMalloc(&dk.p, sizeof(double) * S) != hipSuccess || hipMalloc(&dk2.p, sizeof(double) * S) != hipSuccess ||
        hipMalloc(&di.p, sizeof(int32_t) * S) != hipSuccess || hipMalloc(&di2.p, sizeof(int32_t) * S) != hipSuccess ||
        hipMalloc(&dz.p, sizeof(double) * S) != hipSuccess || hipMalloc(&de.p, sizeof(double) * 2) != hipSuccess ||
        hipMalloc(&dr.p, sizeof(double)) != hipSuccess || hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess ||
        hipMalloc(&dc0.p, sizeof(int32_t)) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemcpyAsync(dc0.p, &zero, sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
    const unsigned nb = (unsigned)((S + 255) / 256);
    std::vector<double> both((size_t)ncoords * 2);
    for (int j = 0; j < ncoords; ++j) {
        hipLaunchKernelGGL(ess_gather_kernel, dim3(nb), dim3(256), 0, s, draws, n, dim, coords[j], chains, N2, (double*)dk.p, (int32_t*)di.p);
        if (hipcub::DeviceRadixSort::SortPairs(dtmp.p, tmp_bytes, (const double*)dk.p, (double*)dk2.p, (const int32_t*)di.p,
                                               (int32_t*)di2.p, (int)S, 0, 64, s) != hipSuccess) return DHMC_ERR_HIP;
        for (int upper = 0; upper < 2; ++upper) {
            hipLaunchKernelGGL(ess_tail_indicator_kernel, dim3(nb), dim3(256), 0, s, (const double*)dk2.p, (const int32_t*)di2.p, S, upper, (double*)dz.p);
            if (int rc = ess_estimate(work, s, (const double*)dz.p, C2, half, 1, (const int32_t*)dc0.p, 1, (double*)de.p + upper, (double*)dr.p)) return rc;
        }
        if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
        if (hipMemcpyAsync(both.data() + 2 * j, de.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;     // (de is reused by the next coordinate)
    }
    for (int j = 0; j < ncoords; ++j) ess[j] = std::min(both[2 * j], both[2 * j + 1]);
    return DHMC_OK;
}

int dhmc_summarize_tree_statistics(int32_t device, void* stream, const double* pi, const double* acceptance_rate,
                                   const int64_t* term_left, const int64_t* term_right, const int32_t* depth,
                                   int64_t chains, int64_t n, int on_device, dhmc_tree_statistics_summary* summary,
                                   double* ebfmi) {
    if (!pi || !acceptance_rate || !term_left || !term_right || !depth || !summary || chains < 1 || n < 1) return DHMC_ERR_INVALID_ARGUMENT;
    const int64_t total = chains * n;
    if (total > 0x7fffffffll) return DHMC_ERR_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    DevBuf in[5], dsum, deb, dcnt, dsorted, dtmp, dout;
    const void* src[5] = {pi, acceptance_rate, term_left, term_right, depth};
    const size_t esz[5] = {8, 8, 8, 8, 4};
    const void* dev[5];
    for (int i = 0; i < 5; ++i) {
        dev[i] = src[i];
        if (!on_device) {
            if (hipMalloc(&in[i].p, esz[i] * total) != hipSuccess) return DHMC_ERR_HIP;
            if (hipMemcpyAsync(in[i].p, src[i], esz[i] * total, hipMemcpyHostToDevice, s) != hipSuccess) return DHMC_ERR_HIP;
            dev[i] = in[i].p;
        }
    }
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, (const double*)nullptr, (double*)nullptr, (int)total, 0, 64, s) != hipSuccess)
        return DHMC_ERR_HIP;
    const size_t ncnt = 3 + TS_DEPTH_BINS;
    if (hipMalloc(&dsum.p, sizeof(double) * chains) != hipSuccess || hipMalloc(&deb.p, sizeof(double) * chains) != hipSuccess ||
        hipMalloc(&dcnt.p, sizeof(unsigned long long) * ncnt) != hipSuccess || hipMalloc(&dsorted.p, sizeof(double) * total) != hipSuccess ||
        hipMalloc(&dtmp.p, tmp_bytes ? tmp_bytes : 8) != hipSuccess || hipMalloc(&dout.p, sizeof(double) * 6) != hipSuccess)
        return DHMC_ERR_HIP;
    if (hipMemsetAsync(dcnt.p, 0, sizeof(unsigned long long) * ncnt, s) != hipSuccess) return DHMC_ERR_HIP;
    hipLaunchKernelGGL(treestat_chain_kernel, dim3((unsigned)chains), dim3(WAVE), 0, s, (const double*)dev[0], (const double*)dev[1],
                       (const int64_t*)dev[2], (const int64_t*)dev[3], (const int32_t*)dev[4], n, (double*)deb.p, (double*)dsum.p,
                       (unsigned long long*)dcnt.p);
    if (hipcub::DeviceRadixSort::SortKeys(dtmp.p, tmp_bytes, (const double*)dev[1], (double*)dsorted.p, (int)total, 0, 64, s) != hipSuccess)
        return DHMC_ERR_HIP;
    hipLaunchKernelGGL(treestat_finish_kernel, dim3(1), dim3(WAVE), 0, s, (const double*)dsum.p, chains, total, (const double*)dsorted.p,
                       (double*)dout.p);
    if (hipGetLastError() != hipSuccess) return DHMC_ERR_HIP;
    double out6[6];
    unsigned long long cnt[3 + TS_DEPTH_BINS];
    if (hipMemcpyAsync(out6, dout.p, sizeof(out6), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipMemcpyAsync(cnt, dcnt.p, sizeof(cnt), hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (ebfmi && hipMemcpyAsync(ebfmi, deb.p, sizeof(double) * chains, hipMemcpyDeviceToHost, s) != hipSuccess) return DHMC_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return DHMC_ERR_HIP;
    summary->n = total;
    summary->a_mean = out6[0];
    for (int i = 0; i < 5; ++i) summary->a_quantiles[i] = out6[1 + i];
    summary->max_depth = (int64_t)cnt[0];
    summary->divergence = (int64_t)cnt[1];
    summary->turning = (int64_t)cnt[2];
    for (int d = 0; d < TS_DEPTH_BINS; ++d) summary->depth_counts[d] = (int64_t)cnt[3 + d];
    return DHMC_OK;
}

int dhmc_state_bytes(dhmc_ctx* c, uint64_t* nbytes) {
    if (!c || !nbytes) return DHMC_ERR_INVALID_ARGUMENT;
    const uint64_t C = c->cfg.chains, Dp = c->Dpad;
    *nbytes = sizeof(BlobHeader) + 4 * C * Dp * sizeof(double) + 2 * C * sizeof(double) + C * sizeof(DAState) + 2 * C * sizeof(uint32_t);
    if (c->cfg.metric == DHMC_METRIC_DENSE) *nbytes += (c->per_chain_dense ? C : 1) * 2 * Dp * Dp * sizeof(double);   // dense M⁻¹ and Wᵀ (shared, or one pair per chain)
    return DHMC_OK;
}

static int blob_io(dhmc_ctx* c, char* blob, bool exporting) {
    const size_t C = c->cfg.chains, Dp = c->Dpad;
    char* p = blob + sizeof(BlobHeader);
    auto io = [&](void* dev, size_t bytes) -> hipError_t {
        hipError_t e = exporting ? hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost) : hipMemcpy(dev, p, bytes, hipMemcpyHostToDevice);
        p += bytes;
        return e;
    };
    HIP_TRY(c, io(c->st.q, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.g, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.minv, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.W, C * Dp * sizeof(double)));
    HIP_TRY(c, io(c->st.lq, C * sizeof(double)));
    HIP_TRY(c, io(c->st.eps, C * sizeof(double)));
    HIP_TRY(c, io(c->st.da, C * sizeof(DAState)));
    HIP_TRY(c, io(c->st.transition, C * sizeof(uint32_t)));
    HIP_TRY(c, io(c->st.status, C * sizeof(uint32_t)));
    if (c->cfg.metric == DHMC_METRIC_DENSE) {
        const size_t nmat = c->per_chain_dense ? C : 1;
        HIP_TRY(c, io(c->d_Minv, nmat * Dp * Dp * sizeof(double)));
        HIP_TRY(c, io(c->d_WT, nmat * Dp * Dp * sizeof(double)));
    }
    return DHMC_OK;
}

int dhmc_export_state(dhmc_ctx* c, void* host_blob, uint64_t nbytes) {
    uint64_t need;
    if (!c || !host_blob || dhmc_state_bytes(c, &need) || nbytes < need) return DHMC_ERR_INVALID_ARGUMENT;
    DHMC_CHECK_USABLE(c);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    BlobHeader h{BLOB_MAGIC, c->cfg.dim, c->cfg.chains, c->Dpad, 0};
    std::memcpy(host_blob, &h, sizeof(h));
    return blob_io(c, (char*)host_blob, true);
}

int dhmc_import_state(dhmc_ctx* c, const void* host_blob, uint64_t nbytes) {
    uint64_t need;
    if (!c || !host_blob || dhmc_state_bytes(c, &need) || nbytes < need) return DHMC_ERR_INVALID_ARGUMENT;
    BlobHeader h;
    std::memcpy(&h, host_blob, sizeof(h));
    if (h.magic != BLOB_MAGIC || h.dim != c->cfg.dim || h.chains != c->cfg.chains || h.Dpad != c->Dpad) return DHMC_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->poisoned = true;    // a partially copied blob is no state
    const int rc = blob_io(c, const_cast<char*>((const char*)host_blob), false);
    if (rc == DHMC_OK) c->poisoned = false;
    return rc;
}

double dhmc_last_run_kernel_ms(const dhmc_ctx* c) { return c ? c->last_ms : 0.0; }
uint64_t dhmc_last_run_rounds(const dhmc_ctx* c) { return c ? c->last_rounds : 0; }
uint64_t dhmc_last_run_leapfrogs(const dhmc_ctx* c) { return c ? c->last_leapfrogs : 0; }
uint64_t dhmc_workspace_bytes(const dhmc_ctx* c) { return c ? c->ws_bytes : 0; }

}  // extern "C"
