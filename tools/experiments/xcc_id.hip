// Which XCD does workgroup b of a launch run on (gfx942 / gfx950: s_getreg_b32 HW_REG_XCC_ID), and is it b % 8?
//   hipcc --offload-arch=gfx950 -O3 -o bin/xcc_id xcc_id.hip && bin/xcc_id
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)); }   // id 20 = XCC_ID, bits [3:0]
__global__ void k(unsigned* out, int spin) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
int main() {
    for (int n : {8, 64, 1024, 4096}) {
        unsigned* d; hipMalloc(&d, n * 4);
        hipLaunchKernelGGL(k, dim3(n), dim3(64), 40 * 1024, 0, d, 100000);
        std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int match = 0, cnt[16] = {0};
        for (int i = 0; i < n; ++i) { match += (int)(h[i] & 15) == i % 8; cnt[h[i] & 15]++; }
        printf("grid %5d: xcc == block %% 8 for %d of them; per XCD:", n, match);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("   first 16:");
        for (int i = 0; i < 16 && i < n; ++i) printf(" %u", h[i] & 15);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
