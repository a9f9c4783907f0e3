"""A few launches of the split-bf16 GEMM on a post-net-shaped problem at four batches per chain (M = 38400), for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
M = int(os.environ.get("M", 38400))
A = torch.randn(M, 2560, device="cuda"); Wt = torch.randn(512, 2560, device="cuda")
for _ in range(5): native.op_gemm(A, Wt, x3=True)
torch.cuda.synchronize()
