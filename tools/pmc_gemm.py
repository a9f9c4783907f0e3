"""One post-net-shaped conv GEMM, a few launches, for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
X = torch.randn(32, 300, 512, device="cuda"); Wp = torch.randn(512, 5 * 512, device="cuda")
for _ in range(5): native.op_conv1d(X, Wp, taps=5, pad=2)
torch.cuda.synchronize()
