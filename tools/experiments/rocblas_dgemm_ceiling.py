"""What does the vendor's fp64 GEMM reach on the dense path's shapes?  (A yardstick for csrc/gemm_f64_mfma.hpp's ≈ 50 TFLOP/s:
torch.mm on float64 CUDA tensors is rocBLAS / hipBLASLt.)   python tools/experiments/rocblas_dgemm_ceiling.py"""
import torch

def rate(M, N, K, reps=30):
    a = torch.randn(M, K, dtype=torch.float64, device="cuda")
    b = torch.randn(K, N, dtype=torch.float64, device="cuda")
    for _ in range(5):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9

for shape in [(4096, 1024, 1024), (2048, 1024, 1024), (1024, 1024, 1024), (512, 1024, 1024), (128, 1024, 1024),
              (8192, 1024, 1024), (4096, 4096, 4096), (8192, 8192, 8192), (1024, 100352, 256), (1024, 256, 100352)]:
    ms, tf = rate(*shape)
    print("M=%6d N=%6d K=%6d  %8.3f ms  %6.1f TFLOP/s" % (*shape, ms, tf), flush=True)
