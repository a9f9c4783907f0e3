"""Per-kernel time of the device vocoder + metric (vocoder.hip) at evaluate.py's sizes: N clips of L = 77 mel frames, 256 + 256 iterations,
HIP events around each entry point; next to the torch-op restatement of the same algorithms.  -> profiles/r04_vocoder_kernels.txt"""
import os
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lip2speech_amd import metrics, native
from lip2speech_amd.datasets.spectrograms import MelSpec2Audio

N = int(os.environ.get("N", 256))
L, ITERS = 77, int(os.environ.get("ITERS", 256))
g = torch.Generator(device="cuda").manual_seed(0)
mel = torch.randn(N, 80, L, device="cuda", generator=g) * 2.0 - 5.0
voc = MelSpec2Audio(max_iters=ITERS, backend="hip").cuda()
clean = 0.1 * torch.randn(N, 256 * (L - 1), device="cuda", generator=g)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, out


ms_all, wave = timed(lambda: voc(mel, rows_per_call=32))
ms_inv, spec = timed(lambda: voc.inverse_mel(torch.exp(mel), rows_per_call=32))
ms_gl, _ = timed(lambda: voc.griffin_lim(spec))
ms_es, score = timed(lambda: metrics.estoi_device(clean, wave, 16000))
print(f"N = {N} clips, L = {L}, {ITERS} + {ITERS} iterations (wall ms, median of 3, one stream):")
print(f"  MelSpec2Audio (hip)      {ms_all:8.2f} ms   = inverse_mel {ms_inv:.2f} + griffin_lim {ms_gl:.2f}")
print(f"  estoi_device             {ms_es:8.2f} ms")
native.profile_enable(True)
native.profile_reset()
voc(mel, rows_per_call=32)
metrics.estoi_device(clean, wave, 16000)
torch.cuda.synchronize()
for name, launches, ms in sorted(native.profile_read(), key=lambda r: -r[2]):
    print(f"  {name:24s} {launches:3d} scope(s)  {ms:8.3f} ms (HIP events)")
native.profile_enable(False)
# FFT work of Griffin-Lim: (ITERS + 1) inverse + ITERS forward 1024-point real transforms per frame
ffts = N * L * (2 * ITERS + 1)
print(f"  griffin_lim: {ffts / 1e6:.2f} M real 1024-point FFTs in {ms_gl:.2f} ms = {ffts / ms_gl / 1e3:.1f} M FFT/s "
      f"(~{ffts * 5 * 512 * 9 / ms_gl / 1e9:.1f} TFLOP/s at 5 N log2 N per complex 512-point transform)")
if N <= 64:
    vt = MelSpec2Audio(max_iters=ITERS, backend="torch").cuda()
    ms_t, _ = timed(lambda: vt(mel, rows_per_call=min(32, N)), reps=1)
    print(f"  MelSpec2Audio (torch ops) {ms_t:8.2f} ms")
