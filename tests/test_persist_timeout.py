"""The guard rails of the persistent decode loop (pdecode.hip): a launch whose workgroups are not all resident must give up after ~2 s, hand back
NaN (mel, stop logits AND attention), count in l2s_persist_timeouts(), FAIL THE NEXT PERSISTENT-ELIGIBLE CALL ONCE with that error (reported by the gate
in front of the persistent forms - a caller that only uses the C entry points learns of it too), leave the launch path in charge afterwards, and come
back when the option is set again (re-arm, per device).  Runs in its own process on the diagnostic build (the starvation hook L2S_TEST_PDECODE_STARVE is
compiled into libl2s_diag.so only).  `-m gpu`."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import time, torch, sys
    sys.path.insert(0, %r)
    from lip2speech_amd import native, synth
    sd = synth.synth_state_dict()
    nm = native.NativeModel(); nm.set_option("persist_decode", 4); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
    ref = native.NativeModel(); ref.set_option("persist_decode", 0); ref.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
    if not native.persist_available():
        print("SKIP: persistent forms not available on this device"); sys.exit(0)
    B, T, S = 1, 8, 5
    v = synth.synth_video(B, T, tag="pt").cuda(); e = synth.synth_speaker_embedding(B, tag="pt").cuda(); g = synth.synth_gumbel(B * native.min_T(T), tag="pt").cuda()
    feat = ref.encoder_fwd(v); vis = native.build_visual(feat, e); state, _ = ref.decoder_prologue(vis, e, g)
    want = ref.decode_steps(state.clone(), B, T, S, want_attn=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mel, stop, attn = nm.decode_steps(state.clone(), B, T, S, want_attn=True)      # one workgroup short (L2S_TEST_PDECODE_STARVE): nobody makes progress
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 1.5 < dt < 20.0, dt                                   # the 2 s wall-clock budget (per poll loop a wave is in), not minutes
    assert torch.isnan(mel).all() and torch.isnan(stop).all() and torch.isnan(attn).all()
    assert native.persist_timeouts() == 1 and not native.persist_available()
    try:                                                         # the library itself reports it: the next persistent-eligible call fails, once
        nm.decode_steps(state.clone(), B, T, S, want_attn=True); raise SystemExit("the call after a timed-out launch did not fail")
    except RuntimeError as e:
        assert "gave up" in str(e), e
    try:
        native.check_persist_timeouts(); raise SystemExit("check_persist_timeouts did not raise")
    except RuntimeError:
        pass
    native.check_persist_timeouts()                              # raised once
    got = nm.decode_steps(state.clone(), B, T, S, want_attn=True)     # persist_available() is 0 now: the launch path, bit for bit
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    nm.set_option("persist_decode", 4)                           # asking again re-arms the device ...
    assert native.persist_available() and native.persist_timeouts() == 1
    nm.set_option("persist_decode", 0)
    print("TIMEOUT PATH OK %%.2f s" %% dt)
""") % ROOT


@pytest.mark.gpu
def test_starved_persistent_launch_gives_up_and_is_reported():
    env = dict(os.environ, L2S_TEST_PDECODE_STARVE="1", L2S_LIB="diag")
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "TIMEOUT PATH OK" in r.stdout or "SKIP" in r.stdout, r.stdout[-2000:]
