"""Split-bf16 GEMM (gemm_x3.hip) next to the f32 MFMA GEMM: time and error against an fp64 product."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
for (M, N, K) in ((9600, 512, 2560), (38400, 512, 2560), (9600, 512, 400), (928, 512, 5632), (3712, 512, 5632), (3712, 4096, 1024), (300, 200, 64), (130, 129, 36)):
    A = torch.randn(M, K, device="cuda"); Wt = torch.randn(N, K, device="cuda") / K ** 0.5
    ref = A.double() @ Wt.double().t()
    c32 = native.op_gemm(A, Wt); c3 = native.op_gemm(A, Wt, x3=True)
    e32 = (c32.double() - ref).abs().max().item(); e3 = (c3.double() - ref).abs().max().item()
    r32 = (c32.double() - ref).pow(2).mean().sqrt().item(); r3 = (c3.double() - ref).pow(2).mean().sqrt().item()
    d32 = timeit(lambda: native.op_gemm(A, Wt)); d3 = timeit(lambda: native.op_gemm(A, Wt, x3=True))
    print(f"M={M:6d} N={N:5d} K={K:5d}: f32 {d32*1e6:8.1f} us {2*M*N*K/d32/1e12:6.1f} TF err max {e32:.2e} rms {r32:.2e} | x3 {d3*1e6:8.1f} us {2*M*N*K/d3/1e12:6.1f} TF err max {e3:.2e} rms {r3:.2e} | {d32/d3:.2f}x", flush=True)
B, S = 32, 300
X = torch.randn(B, S, 512, device="cuda"); Wp = torch.randn(512, 5 * 512, device="cuda") / 2560 ** 0.5
ref = torch.nn.functional.conv1d(X.double().permute(0, 2, 1), Wp.double().view(512, 5, 512).permute(0, 2, 1), padding=2).permute(0, 2, 1)
for x3 in (False, True):
    out = native.op_conv1d(X, Wp, taps=5, pad=2, x3=x3)
    dt = timeit(lambda: native.op_conv1d(X, Wp, taps=5, pad=2, x3=x3))
    print(f"conv1d B*S=9600 Cin=512 k=5 Cout=512 x3={x3}: {dt*1e6:8.1f} us  {2*9600*512*2560/dt/1e12:6.1f} TFLOP/s  err max {(out.double()-ref).abs().max().item():.2e}")
