"""SHA-256 of Lip2Speech.inference outputs (mel_post, lengths, attention) at B = 32 and as a group of 8 batches: a change that claims the same bits
(data movement, instruction scheduling) must not change the hashes.  L2S_LIB=<other build> python tools/hash_inference.py for the A/B.
-> profiles/rNN_inference_hash.txt"""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
h = lambda *ts: hashlib.sha256(b"".join(t.detach().cpu().numpy().tobytes() for t in ts)).hexdigest()[:16]
print(os.path.basename(native.LIB_PATH))
batches = [(synth.synth_video(32, 29, tag=f"b{i}").cuda(), synth.synth_speaker_embedding(32, tag=f"b{i}").cuda(), synth.synth_gumbel(32 * 4, tag=f"b{i}").cuda()) for i in range(8)]
mel, lengths, attn = nm.inference(*batches[0], S=300, want_attn=True)
print(f"  B=32 single: sha256 {h(mel, lengths, attn)}")
outs = nm.inference_multi(batches, S=300)
print(f"  8 batches, one chain: sha256 {h(*[o[0] for o in outs])}  (first batch equals the single call: {torch.equal(outs[0][0], mel)})")
