// Where do the waves of a 5-wave (or 4-wave) workgroup land, and how many such workgroups share a CU?
// Each wave records HW_ID / XCC_ID and start/end timestamps while spinning ~30 us; the host counts, per CU, the
// workgroups whose lifetimes overlap and the SIMD of every wave.   hipcc --offload-arch=gfx950 -O2 wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };
template <int WAVES, int WPE>
__global__ __launch_bounds__(64 * WAVES, WPE) void probe(Rec* out, int spin) {
    extern __shared__ double lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    double x = threadIdx.x;
    if (WPE == 4) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (WPE == 5) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    lds[threadIdx.x] = x;
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = Rec{hwid, xcc, t0, t1};
}
template <int WAVES, int WPE>
void run(const char* name, int nwg, size_t ldsb) {
    Rec* d; hipMalloc(&d, sizeof(Rec) * nwg * WAVES);
    hipFuncSetAttribute((const void*)probe<WAVES, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL((probe<WAVES, WPE>), dim3(nwg), dim3(64 * WAVES), ldsb, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<Rec> h(nwg * WAVES);
    hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    // CU key: xcc, se, sh, cu
    std::map<unsigned, std::vector<int>> by_cu;
    for (int g = 0; g < nwg; ++g) {
        const Rec& r = h[g * WAVES];
        unsigned key = ((r.xcc & 0xf) << 16) | (r.hwid & 0xff00);   // cu_id[11:8], sh[12], se[15:13]
        by_cu[key].push_back(g);
    }
    // workgroups alive at the start of the LAST-started workgroup's first wave is not robust; instead: max overlap per CU
    std::map<int, int> hist;
    for (auto& kv : by_cu) {
        std::vector<std::pair<unsigned long long, int>> ev;
        for (int g : kv.second) { ev.push_back({h[g * WAVES].t0, +1}); ev.push_back({h[g * WAVES].t1, -1}); }
        std::sort(ev.begin(), ev.end());
        int cur = 0, mx = 0;
        for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
        hist[mx]++;
    }
    printf("%s: %d workgroups of %d waves, LDS %zu B -> %zu CUs seen; max co-resident workgroups per CU:", name, nwg, WAVES, ldsb, by_cu.size());
    for (auto& kv : hist) printf("  %d WGs on %d CUs;", kv.first, kv.second);
    printf("\n   SIMD of waves 0..%d of the first 6 workgroups:", WAVES - 1);
    for (int g = 0; g < 6 && g < nwg; ++g) {
        printf("  [");
        for (int w = 0; w < WAVES; ++w) printf("%u", (h[g * WAVES + w].hwid >> 4) & 3);
        printf(" cu%u]", (h[g * WAVES].hwid >> 8) & 0xff);
    }
    printf("\n");
    hipFree(d);
}
int main() {
    run<5, 4>("5 waves, 128 VGPR, 53280 B", 2048, 53280);
    run<5, 4>("5 waves, 128 VGPR, 53248 B", 2048, 53248);
    run<5, 4>("5 waves, 128 VGPR, 40000 B", 2048, 40000);
    run<5, 5>("5 waves,  96 VGPR, 40000 B", 2048, 40000);
    run<5, 5>("5 waves,  96 VGPR, 53248 B", 2048, 53248);
    run<4, 4>("4 waves, 128 VGPR, 40000 B", 2048, 40000);
    run<4, 4>("4 waves, 128 VGPR, 53248 B", 2048, 53248);
    run<8, 4>("8 waves, 128 VGPR, 53248 B", 2048, 53248);
    run<6, 4>("6 waves, 128 VGPR, 53248 B", 2048, 53248);
    return 0;
}
