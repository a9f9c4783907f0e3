"""K-tile timeline of the split-bf16 GEMM from the stamped measurement build (`l2s_op_gemm_x3_timeline`): lane 0 of each of the eight waves of ONE block
stamps the shader clock per K tile - consumers (waves 0-3): tile start / first 24 MFMAs issued / past the barrier / second 24 issued; producers (waves 4-7):
tile start / older register set landed / split + LDS writes done / past the barrier.  Usage: python tools/gemm_x3_timeline.py [M N K] [block]
-> profiles/rNN_gemm_x3_timeline.txt (`dma`: rNN_gemm_x3_timeline_dma.txt)"""
import os, sys
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lip2speech_amd import native

NARROW = "narrow" in sys.argv                            # the 128x128x32 tile (4 + 4 waves, stamps per K tile of 32); default: 128x256x16 (8 + 4 waves, per K step of 16)
DMA = "dma" in sys.argv                                  # the wide tile with its weight operand by LDS-DMA (flags bit 8)
args = [a for a in sys.argv[1:] if a not in ("narrow", "dma")]
M, N, K = (int(a) for a in args[0:3]) if len(args) >= 3 else (38400, 512, 2560)
block = int(args[3]) if len(args) > 3 else 0
NC = 4 if NARROW else 8                                  # consumer waves
KT = 32 if NARROW else 16
L = native.lib()
A = torch.randn(M, K, device="cuda"); Wt = torch.randn(N, K, device="cuda")
for _ in range(3): native.op_gemm(A, Wt, x3=True, x3_narrow=NARROW, x3_dma=DMA)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): native.op_gemm(A, Wt, x3=True, x3_narrow=NARROW, x3_dma=DMA)
e1.record(); torch.cuda.synchronize()
print(f"{M}x{N}x{K} ({'128x128x32 tile' if NARROW else '128x256x16 tile'}): production kernel {e0.elapsed_time(e1) * 100:.1f} us per launch")
ts = torch.zeros(12 * 96 * 8, dtype=torch.int64, device="cuda")
native.check(L.l2s_op_gemm_x3_timeline(ts.data_ptr(), block))
native.op_gemm(A, Wt, x3=True, x3_narrow=NARROW, x3_dma=DMA); torch.cuda.synchronize()
ts.zero_()
e0.record(); native.op_gemm(A, Wt, x3=True, x3_narrow=NARROW, x3_dma=DMA); e1.record(); torch.cuda.synchronize()
native.check(L.l2s_op_gemm_x3_timeline(None, 0))
print(f"stamped kernel {e0.elapsed_time(e1) * 1e3:.1f} us")
t = ts.cpu().numpy().reshape(12, 96, 8)[:NC + 4].astype(np.float64)
nkt = min(96, (K + KT - 1) // KT)
t0 = t[:, :nkt, :4][t[:, :nkt, :4] > 0].min()
t = t - t0
c, p = t[0:NC, :nkt], t[NC:NC + 4, :nkt]
# shader-clock ticks: report in ticks and, against the event time of the whole kernel, nothing else (the counter's rate is printed from the span)
span = t[:, :nkt, :4].max()
print(f"block {block}: stamped span of the K loop {span:.0f} ticks over {nkt} K tiles = {span / nkt:.0f} ticks per K tile")
per = np.diff(c[:, :, 0], axis=1)                       # consumer tile period
print(f"consumer K-tile period (ticks) median {np.median(per):.0f}, 10/90 % {np.percentile(per, 10):.0f} / {np.percentile(per, 90):.0f}")
NM = 24 if NARROW else 12
print(f"MFMA waves (median ticks per tile):  first {NM} MFMAs issued {{:.0f}} | barrier wait {{:.0f}} | next {NM} issued {{:.0f}}".format(
    np.median(c[:, 2:, 1] - c[:, 2:, 0]), np.median(c[:, 2:, 2] - c[:, 2:, 1]), np.median(c[:, 2:, 3] - c[:, 2:, 2])))
pv = p[:, 2:nkt - 3]
print("staging waves (median ticks per tile):  fetch issue + wait for the older set {:.0f} | split + LDS writes {:.0f} | barrier wait {:.0f}".format(
    np.median(pv[:, :, 1] - pv[:, :, 0]), np.median(pv[:, :, 2] - pv[:, :, 1]), np.median(pv[:, :, 3] - pv[:, :, 2])))
if not NARROW: print("producers: address arithmetic + 6 load requests {:.0f} | wait for the set requested two steps ago {:.0f}".format(
    np.median(pv[:, :, 4] - pv[:, :, 0]), np.median(pv[:, :, 1] - pv[:, :, 4])))
if NARROW: print("producers: split VALU (older set landed -> all 16 splits in registers) {:.0f} | 24 ds_write_b64 + lgkmcnt(0) {:.0f}".format(
    np.median(pv[:, :, 4] - pv[:, :, 1]), np.median(pv[:, :, 2] - pv[:, :, 4])))
print("first 12 K tiles, MFMA wave 0 (tile start / first half issued / past the barrier / second half issued) and the first staging wave (tile start / older set landed / stores done / past the barrier), ticks since the block's first stamp:")
for kt in range(min(12, nkt)):
    print(f"  kt {kt:2d}  MFMA wave " + " ".join(f"{v:7.0f}" for v in c[0, kt, :4]) + "   staging wave " + " ".join(f"{v:7.0f}" for v in p[0, kt, :4]))
