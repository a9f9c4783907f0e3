"""A few launches of the f32 and of the split-bf16 GEMM on the post-net shape at four batches per chain, for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
M = int(os.environ.get("M", 38400))
A = torch.randn(M, 2560, device="cuda"); Wt = torch.randn(512, 2560, device="cuda")
for _ in range(3): native.op_gemm(A, Wt, x3=True)
for _ in range(3): native.op_gemm(A, Wt, x3=False)
torch.cuda.synchronize()
