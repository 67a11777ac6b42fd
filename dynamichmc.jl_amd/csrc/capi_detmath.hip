// dhmc_detmath_selftest: the ABI's scalar math (include/dhmc_detmath.h) evaluated ON THE DEVICE, one value per call slot, with
// the operand placement the kernels use (csrc/detmath_dev.hpp): policy 0 = dm_generic (the header as compiled by hipcc),
// 1 = dm_vector (per-lane arguments, 64 values per wavefront), 2 = dm_uniform (wave-uniform arguments: one value per
// wavefront, table rows by scalar loads).  The three must return the bits of the CPU side's dm_generic instantiation — that
// is the whole contract — and tests/test_gpu_detmath.py checks exactly that against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dhmc.h"
#include "detmath_dev.hpp"
#include "wave.hpp"

namespace dhmc {
namespace {

template <class P>
__device__ __forceinline__ double detmath_eval(int kind, double x, double y) {
    double s, c;
    const uint64_t r1 = dm_bits(x), r2 = dm_bits(y);
    switch (kind) {
    case 0: return det_exp_t<P>(x);
    case 1: return det_log_t<P>(x);
    case 2: return det_log1p_nonneg_t<P>(x);
    case 3: det_sincos2pi_t<P>(x, &s, &c); return s;
    case 4: det_sincos2pi_t<P>(x, &s, &c); return c;
    case 5: return det_randexp_t<P>(r1);
    case 6: det_randn2_t<P>(r1, r2, &s, &c); return s;
    case 7: det_randn2_t<P>(r1, r2, &s, &c); return c;
    case 8: return det_logaddexp_t<P>(x, y);
    case 9: return det_pow_pos_t<P>(x, y);
    case 10: return det_logistic_sigma_t<P>(x);
    case 11: return det_log1pexp_t<P>(x);
    default: return dm_nan();
    }
}

template <class P>
__global__ __launch_bounds__(64) void detmath_lane_kernel(int kind, int64_t n, const double* __restrict__ x,
                                                         const double* __restrict__ y, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = detmath_eval<P>(kind, x[i], y ? y[i] : 0.0);
}

// kinds 10 / 11 with policy 1: the logistic round engine's own evaluation — logistic_link_batch (detmath_dev.hpp), two arguments
// of a lane at once (lane i: x[2i] and x[2i + 1]; the second stands in for the first when n is odd), common path and whole-wave
// rare path as the data decide
__global__ __launch_bounds__(64) void detmath_link_batch_kernel(int kind, int64_t n, const double* __restrict__ x, double* __restrict__ out) {
    const int64_t i0 = 2 * ((int64_t)blockIdx.x * 64 + threadIdx.x), i1 = i0 + 1;
    double eta[2], sig[2], l1pe[2];
    eta[0] = i0 < n ? x[i0] : 0.0;
    eta[1] = i1 < n ? x[i1] : eta[0];
    logistic_link_batch<2>(eta, sig, l1pe);
    if (i0 < n) out[i0] = kind == 10 ? sig[0] : l1pe[0];
    if (i1 < n) out[i1] = kind == 10 ? sig[1] : l1pe[1];
}

// one value per wavefront: every lane holds the same argument, as the tree logic's scalars do
__global__ __launch_bounds__(64) void detmath_uniform_kernel(int kind, int64_t n, const double* __restrict__ x,
                                                             const double* __restrict__ y, double* __restrict__ out) {
    const int64_t i = blockIdx.x;
    const double xv = uni_f64(x[i]), yv = uni_f64(y ? y[i] : 0.0);
    const double r = detmath_eval<dm_uniform>(uni_i32(kind), xv, yv);
    if (threadIdx.x == 0) out[i] = r;
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace
}  // namespace dhmc

extern "C" int dhmc_detmath_selftest(int32_t device, int32_t kind, int32_t policy, int64_t n, const double* x, const double* y, double* out) {
    using namespace dhmc;
    if (!x || !out || n < 1 || n > (1ll << 24) || kind < 0 || kind > 11 || policy < 0 || policy > 2) return DHMC_ERR_INVALID_ARGUMENT;
    if (kind >= 10 && policy == 2) return DHMC_ERR_INVALID_ARGUMENT;        // the link's arguments are never wave-uniform
    if (kind >= 6 && kind <= 9 && !y) return DHMC_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device) != hipSuccess) return DHMC_ERR_NO_DEVICE;
    Buf dx, dy, dout;
    const size_t bytes = sizeof(double) * (size_t)n;
    if (hipMalloc(&dx.p, bytes) != hipSuccess || hipMalloc(&dout.p, bytes) != hipSuccess || (y && hipMalloc(&dy.p, bytes) != hipSuccess)) return DHMC_ERR_HIP;
    if (hipMemcpy(dx.p, x, bytes, hipMemcpyHostToDevice) != hipSuccess) return DHMC_ERR_HIP;
    if (y && hipMemcpy(dy.p, y, bytes, hipMemcpyHostToDevice) != hipSuccess) return DHMC_ERR_HIP;
    const double* px = (const double*)dx.p;
    const double* py = (const double*)dy.p;
    double* po = (double*)dout.p;
    const unsigned blocks = (unsigned)((n + 63) / 64);
    if (policy == 0) hipLaunchKernelGGL(detmath_lane_kernel<dm_generic>, dim3(blocks), dim3(64), 0, 0, kind, n, px, py, po);
    else if (policy == 1 && kind >= 10) hipLaunchKernelGGL(detmath_link_batch_kernel, dim3((unsigned)((n + 127) / 128)), dim3(64), 0, 0, kind, n, px, po);
    else if (policy == 1) hipLaunchKernelGGL(detmath_lane_kernel<dm_vector>, dim3(blocks), dim3(64), 0, 0, kind, n, px, py, po);
    else hipLaunchKernelGGL(detmath_uniform_kernel, dim3((unsigned)n), dim3(64), 0, 0, kind, n, px, py, po);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return DHMC_ERR_HIP;
    if (hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return DHMC_ERR_HIP;
    return DHMC_OK;
}
