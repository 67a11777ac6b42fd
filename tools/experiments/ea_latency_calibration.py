"""Calibration kernels for profiles/r05_traffic_destination.txt: the latency of the L2's requests to the fabric (TCC_EA0_RDREQ_LEVEL /
TCC_EA0_RDREQ, TCC_EA0_WRREQ_LEVEL / TCC_EA0_WRREQ) when the lines come from (a) HBM — a copy through 8 GiB, every line touched once —
and (b) the Infinity Cache — a copy inside 96 MiB (above the 32 MiB of L2, below the 256 MiB of Infinity Cache) repeated 200 times.
Run under rocprofv3 --pmc; the kernels are torch's elementwise copy kernels, told apart by their order and grid sizes."""
import torch
dev = torch.device("cuda:0")
big = torch.empty(1 << 29, dtype=torch.float64, device=dev)           # 4 GiB source, 4 GiB destination
big2 = torch.empty_like(big)
big.fill_(1.0); torch.cuda.synchronize()
for _ in range(3):
    big2.copy_(big)                                                    # marker: grid of a 4 GiB copy
torch.cuda.synchronize()
del big, big2
small = torch.empty(6 << 20, dtype=torch.float64, device=dev)          # 48 MiB source, 48 MiB destination
small2 = torch.empty_like(small)
small.fill_(1.0); torch.cuda.synchronize()
for _ in range(200):
    small2.copy_(small)
torch.cuda.synchronize()
print("done")
