"""Which host-side calls of one training step end up as device-to-device memcpy kernels (__amd_rocclr_copyBuffer)?  torch.profiler with stacks."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from model.model import get_network
from lip2speech_amd import native, synth
from lip2speech_amd.training import AdamWAmsgrad, GradAllReducer, draw_dropout, model_forward_backward
Bt, T, St = 8, 29, 77
net = get_network("train").cuda()
net.load_state_dict({k: v for k, v in synth.synth_state_dict().items() if k.startswith(("encoder.", "decoder."))}, strict=False)
flat = net._train_state(); nm = net.native_model(); opt = AdamWAmsgrad(flat, lr=1e-4, weight_decay=1e-6); reducer = GradAllReducer(flat.grad)
video = synth.synth_video(Bt, T, tag="t").cuda(); emb = synth.synth_speaker_embedding(Bt, tag="t").cuda(); gum = synth.synth_gumbel(Bt * 4, tag="t").cuda()
mels = synth.synth_mels(Bt, St, tag="t").cuda(); gate = torch.zeros(Bt, St, device="cuda"); gate[:, -1] = 1
bos = dict(net.decoder.named_parameters())["BOS"]; mask = torch.zeros(St, dtype=torch.bool); mask[1::2] = True
nm.train_set_bn(True, 0.1)
def step():
    drop = draw_dropout(Bt, T, St, video.device)
    out = model_forward_backward(nm, video, emb, gum, mels, gate, teacher_mask=mask, bos=bos.detach(), drop=drop)
    reducer.start(); mul = reducer.wait(); opt.step(max_norm=1.0, grad_mul=mul); nm.train_refresh_weights(); return out
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
kinds = collections.Counter()
for e in ev:
    n = e.name
    if "Memcpy" in n or "copyBuffer" in n or "memcpy" in n.lower():
        kinds[(n, e.device_type)] += 1
for k, c in kinds.most_common(20): print(c, k)
# aggregate by python stack for ops that are copies
agg = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::to", "aten::_to_copy") and e.stack:
        site = next((s for s in e.stack if "lip2speech_amd" in s or "tools/" in s), e.stack[0])
        agg[(e.name, site)] += 1
for k, c in agg.most_common(25): print(c, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12))
