"""Times one post-net-shaped split-bf16 GEMM with the library named by L2S_LIB.  Used with temporary builds of gemm_x3.hip in which one
ingredient was compiled out (-DX3_VAR=1 no split VALU, 2 no global loads in the loop, 3 a third of the MFMAs, 4 no staging, 5 no MFMAs; the
switch was removed again) to find what bounds the kernel: every variant ran 500-590 us against 600-690 us - the operand fetch, DESIGN.md §6."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
M, N, K = 38400, 512, 2560
A = torch.randn(M, K, device="cuda"); Wt = torch.randn(N, K, device="cuda")
for _ in range(3): native.op_gemm(A, Wt, x3=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): native.op_gemm(A, Wt, x3=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"{os.path.basename(native.LIB_PATH)}: {dt*1e6:.1f} us")
