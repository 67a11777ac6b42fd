// The caller's own log density as a DEVICE FUNCTOR (include/dhmc.h dhmc_register_target_source): its HIP source is compiled at
// run time (hiprtc) into the library's own kernel templates; code objects are kept in a checksummed disk cache (DHMC_RTC_CACHE).
#include <hip/hiprtc.h>
#include "capi_util.hpp"

// ---- the caller's device functor, compiled at run time (include/dhmc.h dhmc_register_target_source) -----------------------
#include "gen/rtc_headers.inc"     // const char dhmc_rtc_headers[]: the kernel headers as one string (make_rtc_source.py)
std::vector<UserTarget> g_user_targets;
std::mutex g_user_mutex;
std::string g_rtc_log;

namespace {
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
// npl < 0: the PACKED per-draw kernel alone (several chains per wavefront, packed_kernels.hpp PackedFunctor), -npl = 100 L + cpl
std::vector<std::string> rtc_kernel_names(const std::string& name, int npl, bool dense) {
    const std::string T = "dhmc::" + name, N = std::to_string(npl);
    if (npl < 0) return {"dhmc::nuts_run_packed_kernel<dhmc::PackedFunctor<" + T + ">, " + std::to_string((-npl) / 100) + ", " + std::to_string((-npl) % 100) + ">"};
    if (npl >= 32) return {"dhmc::functor_eval_kernel<" + T + ", " + N + ">"};      // beyond 1024 coordinates: the batched evaluation only
    if (!dense) {
        std::vector<std::string> names = {"dhmc::nuts_run_kernel<" + T + ", " + N + ", true>", "dhmc::nuts_run_kernel<" + T + ", " + N + ", false>",
                                          "dhmc::init_kernel<" + T + ", " + N + ">", "dhmc::stepsize_search_kernel<" + T + ", " + N + ">",
                                          "dhmc::probe_kernel<" + T + ", " + N + ", false, 0>", "dhmc::probe_kernel<" + T + ", " + N + ", false, 1>"};
        if (npl <= 2) names.push_back("dhmc::nuts_run_pipeline_kernel<" + T + ", " + N + ">");     // four wavefronts per chain (D <= 128)
        return names;
    }
    return {"dhmc::rounds_k0_kernel<" + T + ", " + N + ">", "dhmc::rounds_k2_kernel<" + T + ", " + N + ">",
            (npl >= 8 ? "dhmc::rounds_k3b_kernel<" : "dhmc::rounds_k3_kernel<") + T + ", " + N + ">",
            "dhmc::nuts_run_dense_kernel<" + T + ", " + N + ">", "dhmc::stepsize_search_dense_kernel<" + T + ", " + N + ">",
            "dhmc::probe_kernel<" + T + ", " + N + ", true, 0>", "dhmc::probe_kernel<" + T + ", " + N + ", true, 1>"};
}
}  // namespace
// compile `source` (which defines dhmc::`name`) with the kernel templates for one chain width; *code receives the code object
int rtc_compile(const std::string& source, const std::string& name, int npl, bool dense, std::vector<char>* code, std::vector<std::string>* lowered, bool fresh) {
    std::string src = rtc_prelude();
    src += dhmc_rtc_headers;
    src += "\n// ---- the caller's functor -------------------------------------------------------------\n";
    src += source;
    // the functor's traits for the host (dhmc_create reads the symbol: which kernels may run it)
    src += "\nextern \"C\" __device__ __attribute__((used)) int dhmc_user_traits = (dhmc::" + name + "::kRecomputeGrad ? 1 : 0) | (dhmc::" + name + "::kBigDims ? 2 : 0)"
           " | (dhmc::PackedFunctor<dhmc::" + name + ">::kEligible ? 4 : 0);\n";       // 4: may run packed (a sum of per-coordinate terms)
    const std::vector<std::string> exprs = rtc_kernel_names(name, npl, dense);
    // the architecture of the device the context lives on (this library's own kernels are built for gfx950; a functor follows
    // whatever device it will run beside them on)
    std::string arch = "--offload-arch=gfx950";
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0])
            arch = std::string("--offload-arch=") + prop.gcnArchName;
    }
    // DHMC_RTC_CACHE=<directory>: code objects are kept there, keyed by everything that went into them, so that the next
    // process (a new Julia session) loads instead of compiling (≈ 2 s diagonal, ≈ 15 s dense per functor and chain width)
    std::string cache_file;
    if (const char* dir = std::getenv("DHMC_RTC_CACHE"); dir && *dir && code && lowered) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const std::string& t) { for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; } h ^= 0xff; h *= 1099511628211ull; };
        int major = 0, minor = 0;
        (void)hiprtcVersion(&major, &minor);
        mix(src); mix(dhmc_version()); mix(std::to_string(major) + "." + std::to_string(minor));
        mix(arch);                                  // a cache directory shared by machines with different GPUs holds one object per architecture
        for (const auto& e : exprs) mix(e);
        char hex[17];
        std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)h);
        cache_file = std::string(dir) + "/dhmc_rtc_" + hex + ".co";
        if (FILE* f = fresh ? nullptr : std::fopen(cache_file.c_str(), "rb")) {     // fresh: the cached object did not load — compile and replace it
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
int npl_for_user_dim(int D) { return D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : D <= 512 ? 8 : D <= 1024 ? 16 : D <= 2048 ? 32 : D <= 4096 ? 64 : 0; }

extern "C" {
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
    const int rc = rtc_compile(hip_source, functor_name, npl, metric == DHMC_METRIC_DENSE, nullptr, nullptr, false);
    if (log && log_bytes) {
        const size_t n = std::min<size_t>(g_rtc_log.size(), (size_t)log_bytes - 1);
        std::memcpy(log, g_rtc_log.data(), n);
        log[n] = '\0';
    }
    return rc;
}
const char* dhmc_target_source_log(void) { return g_rtc_log.c_str(); }
}  // extern "C"
