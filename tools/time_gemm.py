"""fp32 MFMA GEMM kernel efficiency on post-net-shaped problems (plain and implicit-conv addressing)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K) in ((9600, 512, 2560), (9600, 512, 512), (928, 512, 5632), (928, 4096, 1024), (19200, 512, 2560), (9600, 1024, 2560)):
    A = torch.randn(M, K, device="cuda"); Wt = torch.randn(N, K, device="cuda")
    dt = timeit(lambda: native.op_gemm(A, Wt))
    print(f"plain  M={M:6d} N={N:5d} K={K:5d}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:6.1f} TFLOP/s")
B, S = 32, 300
X = torch.randn(B, S, 512, device="cuda"); Wp = torch.randn(512, 5 * 512, device="cuda")
dt = timeit(lambda: native.op_conv1d(X, Wp, taps=5, pad=2))
print(f"conv1d B*S=9600 Cin=512 k=5 Cout=512: {dt*1e6:8.1f} us  {2*9600*512*2560/dt/1e12:6.1f} TFLOP/s")
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
dt = timeit(lambda: torch.mm(a, b), 5)
print(f"torch.mm fp32 8192^3 (hipBLASLt): {dt*1e3:.2f} ms {2*8192**3/dt/1e12:.1f} TFLOP/s")
a = torch.randn(9600, 2560, device="cuda"); b = torch.randn(2560, 512, device="cuda")
dt = timeit(lambda: torch.mm(a, b))
print(f"torch.mm fp32 9600x2560x512: {dt*1e6:.1f} us {2*9600*512*2560/dt/1e12:.1f} TFLOP/s")
