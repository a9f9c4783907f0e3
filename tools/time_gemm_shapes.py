import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from lip2speech_amd import native
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, (M, N, K) in {"postnet last layer, 1 batch": (9600, 80, 2560), "postnet last layer, 8 batches": (76800, 80, 2560),
                        "postnet first layer, 1 batch": (9600, 512, 400), "postnet first layer, 8 batches": (76800, 512, 400),
                        "conv_last, 1 batch": (8352, 1024, 464), "conv_last, 8 batches": (66816, 1024, 464),
                        "bottleneck, 1 batch": (928, 512, 2560), "bottleneck, 8 batches": (7424, 512, 2560),
                        "enc_proj, 1 batch": (928, 512, 1024), "enc_proj, 8": (7424, 512, 1024),
                        "k=1 multihop, 1 batch": (928, 512, 512), "k=1 multihop, 8": (7424, 512, 512)}.items():
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    f = timeit(lambda: native.op_gemm(A, W))
    w = timeit(lambda: native.op_gemm(A, W, x3=True))
    n = timeit(lambda: native.op_gemm(A, W, x3=True, x3_narrow=True))
    print(f"{name:32s} {M:6d}x{N}x{K}: f32 {f:7.1f} us | x3 wide {w:7.1f} | x3 narrow {n:7.1f}")
