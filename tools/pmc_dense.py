"""The encoder of one grouped pass at 256 clips (front-end conv, ShuffleNet trunk units, conv_last), for
rocprofv3 --pmc / --kernel-trace passes (tools/pmc_dense_kernels.sh).
-> profiles/rNN_pmc_dense_kernels.txt (through tools/pmc_dense_kernels.sh)"""
import os, sys, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
ROWS = int(os.environ.get("ROWS", 256))
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
video = torch.cat([synth.synth_video(32, 29, tag=f"b{i}") for i in range(ROWS // 32)]).cuda()
nm.encoder_fwd(video)
torch.cuda.synchronize()
print(f"encoder forward of {ROWS} clips done (run under rocprofv3: tools/pmc_dense_kernels.sh)")
