"""Three post-net passes at 256 clips (S = 300) for rocprofv3 --pmc passes; DMA=0 switches the LDS-DMA weight operand off (option gemm_x3_dma)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith("decoder.")}
nm = native.NativeModel(); nm.set_option("gemm_x3_dma", int(os.environ.get("DMA", 1))); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
mel = torch.randn(256, 300, 80, device="cuda")
for _ in range(3): nm.postnet(mel)
torch.cuda.synchronize()
