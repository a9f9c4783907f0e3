"""bf16-operand GEMM (gemm_nt_kernel<.,true>) next to the f32 and split-bf16 kernels: time and error against an fp64 product."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
for (M, N, K) in ((9600, 512, 2560), (76800, 512, 2560), (76800, 512, 400), (76800, 80, 2560), (7424, 512, 5632), (7424, 4096, 1024), (66816, 768, 464)):
    A = torch.randn(M, K, device="cuda"); Wt = torch.randn(N, K, device="cuda") / K ** 0.5
    ref = A.double() @ Wt.double().t()
    out = []
    for kw in ({}, {"x3": True}, {"bf16": True}):
        c = native.op_gemm(A, Wt, **kw); e = (c.double() - ref).abs().max().item()
        d = timeit(lambda: native.op_gemm(A, Wt, **kw)); out.append(f"{d*1e6:8.1f} us {2*M*N*K/d/1e12:6.1f} TF err {e:.1e}")
    print(f"M={M:6d} N={N:5d} K={K:5d}: f32 {out[0]} | x3 {out[1]} | bf16 {out[2]}", flush=True)
