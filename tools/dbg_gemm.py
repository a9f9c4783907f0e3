import sys, os, torch
sys.path.insert(0, "/root/repo")
from lip2speech_amd import native
torch.manual_seed(0)
for (M, N, K) in ((64, 64, 32), (64, 64, 64), (128, 64, 32), (100, 70, 36)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    C = native.op_gemm(A, W)
    ref = A.double() @ W.double().t()
    err = (C.double() - ref).abs()
    print(M, N, K, "max err", err.max().item(), "bad rows", (err.max(dim=1).values > 1e-3).sum().item(), "bad cols", (err.max(dim=0).values > 1e-3).sum().item())
    if err.max() > 1e-3:
        # which k contributions are missing? test with one-hot A columns
        for kk in (0, 3, 4, 31 if K > 31 else K - 1):
            A2 = torch.zeros(M, K, device="cuda"); A2[:, kk] = 1
            C2 = native.op_gemm(A2, W); r2 = W[:, kk].unsqueeze(0).expand(M, N)
            print("   k", kk, "err", (C2 - r2).abs().max().item())
