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
