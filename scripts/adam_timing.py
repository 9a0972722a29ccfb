#!/usr/bin/env python3
"""Dev tool (GPU box): FusedAdam.step alone on T1M-sized parameters (two fp64 tensors of ~500 k rows x 2), K steps per
hipGraph -> us per step and the achieved bandwidth over its 7 arrays per tensor (p, g, m, v read; p, m, v written)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hidenn_fem_amd.optim import FusedAdam

dev = torch.device("cuda:0")
n = [(499000, 2), (500000, 2)]
ps = [torch.nn.Parameter(torch.randn(s, dtype=torch.float64, device=dev)) for s in n]
for p in ps:
    p.grad = torch.randn_like(p)
opt = FusedAdam(ps, lr=1e-9, capturable=True).init_state()
K = 100
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(K):
        opt.step()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    g.replay(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K)
us = sorted(ts)[2] * 1e6
byt = sum(p.numel() for p in ps) * 8 * 7
print(f"FusedAdam.step: {us:.2f} us, {byt / 1e6:.1f} MB -> {byt / us / 1e6:.2f} TB/s")
