"""Conv1d stacks on the split-bf16 kernel at the grouped sizes (post-net layer: 256 x 300 rows, 512 -> 512, k 5; MultiHop branches: 256 x 29
rows, k 3 / 7 / 11): time and error against an fp64 convolution, f32 kernel beside it; both tiles of the split-bf16 kernel (forced: in the product the launch picks one by its
number of rounds, `x3_wide` in gemm_x3.hip)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
for (B, S, Ci, Co, k) in ((256, 300, 512, 512, 5), (32, 300, 512, 512, 5), (256, 29, 512, 512, 11), (256, 29, 512, 512, 7), (256, 29, 512, 512, 3), (256, 300, 80, 512, 5)):
    X = torch.randn(B, S, Ci, device="cuda"); Wt = torch.randn(Co, Ci, k, device="cuda") / (Ci * k) ** 0.5
    Wp = Wt.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    ref = torch.nn.functional.conv1d(X[:8].double().permute(0, 2, 1), Wt.double(), padding=k // 2).permute(0, 2, 1)
    out = []
    for x3, narrow in ((False, False), (True, False), (True, True)):
        o = native.op_conv1d(X, Wp, taps=k, pad=k // 2, x3=x3, x3_narrow=narrow)
        dt = timeit(lambda: native.op_conv1d(X, Wp, taps=k, pad=k // 2, x3=x3, x3_narrow=narrow))
        out.append(f"{dt*1e6:8.1f} us {2*B*S*Co*Ci*k/dt/1e12:6.1f} TF err {(o[:8].double()-ref).abs().max().item():.1e}")
    print(f"B*S={B*S:6d} Cin={Ci:4d} k={k:2d} Cout={Co}: f32 {out[0]} | x3 128x256x16 tile {out[1]} | x3 128x128x32 tile {out[2]}", flush=True)
