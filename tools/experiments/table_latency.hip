// What a table lookup costs a LONE wave per SIMD on gfx950 (the headline kernel's situation): round-trip clocks of
//   (a) s_load_dwordx16 of one 64-byte row of a 16 KB constant table (the softplus cell of det_logaddexp, csrc/detmath_dev.hpp),
//       rows picked at random from the first `span` rows (span = 1: always the same row; 32: the hot 2 KB; 256: all 16 KB);
//   (b) the same row fetched a second time right away (scalar-cache hit);
//   (c) a per-lane global_load_dwordx4 gather from a 4 KB table (the log / sincos tables of the momentum refresh);
//   (d) a ds_read_b64 for scale.
// 1024 blocks of one wave (one per SIMD, like the kernel), each timing `iters` lookups with s_memtime; the s_memtime pair's own
// cost is measured and subtracted.      hipcc --offload-arch=gfx950 -O3 -o table_latency table_latency.hip && ./table_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double v8d __attribute__((ext_vector_type(8)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

__global__ __launch_bounds__(64) void lat_kernel(const double* __restrict__ tbl, const double* __restrict__ vtbl, int span, int iters,
                                                 unsigned long long* out) {
    __shared__ double lds[512];
    const int lane = threadIdx.x;
    lds[lane] = lane; lds[lane + 64] = 2.0 * lane;
    __syncthreads();
    uint32_t state = 12345u + 977u * blockIdx.x;
    unsigned long long t_empty = 0, t_smem = 0, t_smem2 = 0, t_gather = 0, t_lds = 0;
    double sink = 0.0;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const int row = (int)__builtin_amdgcn_readfirstlane((state >> 8) % (uint32_t)span);
        const double* p = tbl + (size_t)row * 8;
        uint64_t a = memtime();
        uint64_t b = memtime();
        t_empty += b - a;
        v8d r;
        a = memtime();
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
        b = memtime();
        t_smem += b - a;
        sink += r[0] + r[7];
        a = memtime();
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
        b = memtime();
        t_smem2 += b - a;
        sink += r[1];
        const uint32_t off = (((state >> 4) + 2654435761u * (uint32_t)lane) & 255u) * 16u;   // 256 rows of 16 bytes, per lane
        v2d g;
        a = memtime();
        asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(off), "s"(vtbl) : "memory");
        b = memtime();
        t_gather += b - a;
        sink += g[0];
        double l;
        const uint32_t laddr = (uint32_t)(uintptr_t)lds + 8u * (uint32_t)((lane * 7 + it) & 127);
        a = memtime();
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(l) : "v"(laddr) : "memory");
        b = memtime();
        t_lds += b - a;
        sink += l;
    }
    if (lane == 0) {
        atomicAdd(&out[0], t_empty); atomicAdd(&out[1], t_smem); atomicAdd(&out[2], t_smem2);
        atomicAdd(&out[3], t_gather); atomicAdd(&out[4], t_lds);
        if (sink == 1.2345) out[7] = 1;
    }
}

int main() {
    const int rows = 256, iters = 2000, blocks = 1024;
    std::vector<double> h(rows * 8), hv(512);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i;
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = (double)i;
    double *d, *dv; unsigned long long* o;
    hipMalloc(&d, h.size() * 8); hipMalloc(&dv, hv.size() * 8); hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
    for (int span : {1, 32, 256}) {
        hipMemset(o, 0, 64);
        hipLaunchKernelGGL(lat_kernel, dim3(blocks), dim3(64), 0, 0, d, dv, span, iters, o);
        hipDeviceSynchronize();
        unsigned long long r[8];
        hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
        const double n = (double)blocks * iters, e = r[0] / n;
        printf("rows in play %3d: s_memtime pair %.0f clocks | s_load_dwordx16 %.0f | again (cache hit) %.0f | global_load_dwordx4 gather %.0f | ds_read_b64 %.0f  (net of the pair)\n",
               span, e, r[1] / n - e, r[2] / n - e, r[3] / n - e, r[4] / n - e);
    }
    return 0;
}
