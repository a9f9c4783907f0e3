"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/l2s.h declares, the host
mirror has the reference's interface, and the sizing helpers are consistent.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from lip2speech_amd import native
    if not (os.path.exists(native.LIB_PATH) and os.path.exists(native.DIAG_LIB_PATH)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lip2speech_amd", "csrc"), "-j", "8"], check=True)
    return ctypes.CDLL(native.LIB_PATH)


def _declared(header_path):
    """names declared as functions in a header: `l2s_name(` at the start of a declaration (comments mention other names)"""
    text = re.sub(r"/\*.*?\*/", "", open(header_path).read(), flags=re.S)
    return set(re.findall(r"\b(l2s_[a-z_0-9A-Z]+)\s*\(", text))


def test_header_symbols_exported(built_lib):
    """include/l2s.h is the PRODUCT boundary: libl2s_hip.so exports exactly what it declares - what INTEGRATION.md's stub and the package's callers
    bind - and include/l2s_diag.h the diagnostic surface: libl2s_diag.so exports the product ABI plus exactly those symbols."""
    from lip2speech_amd import native
    declared = _declared(os.path.join(ROOT, "include", "l2s.h"))
    assert declared == set(native.ABI_SYMBOLS), declared ^ set(native.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(built_lib, sym), f"{sym} not exported"
    assert built_lib.l2s_abi_version() == 2
    diag_declared = _declared(os.path.join(ROOT, "include", "l2s_diag.h"))
    assert diag_declared == set(native.DIAG_SYMBOLS), diag_declared ^ set(native.DIAG_SYMBOLS)
    for sym in diag_declared:
        assert not hasattr(built_lib, sym), f"{sym} is a diagnostic entry point: it must not be in the product library"
    exported = set(re.findall(r" T (l2s_[a-z_0-9A-Z]+)\n", subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout))
    assert exported == declared, exported ^ declared
    dlib = ctypes.CDLL(native.DIAG_LIB_PATH)
    for sym in declared | diag_declared:
        assert hasattr(dlib, sym), f"libl2s_diag.so lacks {sym}"
    # the A/B switches of the block forms are diagnostic too: the product library does not know them
    h = ctypes.c_void_p()
    assert built_lib.l2s_model_create(ctypes.byref(h)) == 0
    assert built_lib.l2s_model_set_option(h, b"persist_decode", 0) == 0
    for name in sorted(native.DIAG_OPTIONS):
        assert built_lib.l2s_model_set_option(h, name.encode(), 0) != 0, name
    built_lib.l2s_model_destroy(h)
    dlib.l2s_model_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    dlib.l2s_model_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    dlib.l2s_model_destroy.argtypes = [ctypes.c_void_p]
    assert dlib.l2s_model_create(ctypes.byref(h)) == 0
    for name in sorted(native.DIAG_OPTIONS):
        assert dlib.l2s_model_set_option(h, name.encode(), 1) == 0, name
    dlib.l2s_model_destroy(h)


def test_sizes_and_errors(built_lib):
    from lip2speech_amd import native
    L = native.lib()
    assert [native.min_T(t) for t in (29, 75, 50, 25, 7)] == [4, 10, 7, 3, 1]
    assert L.l2s_workspace_bytes(32, 29, 96, 96, 300) > L.l2s_workspace_bytes(2, 29, 96, 96, 300) > 0
    offs = [L.l2s_state_offset(2, 29, f) for f in range(9)]
    assert offs == sorted(offs) and offs[0] == 0 and L.l2s_state_floats(2, 29) > offs[-1]
    # error path: finalize without tensors -> non-zero + message
    h = ctypes.c_void_p()
    assert L.l2s_model_create(ctypes.byref(h)) == 0
    assert L.l2s_model_finalize(h, None) != 0
    assert b"no encoder" in L.l2s_last_error()
    L.l2s_model_destroy(h)


def test_boundary_interface_matches_reference():
    """Names/signatures of the reference's model.model / hparams API."""
    import inspect

    import hparams
    from model.model import Lip2Speech, get_network
    from model.modules import Decoder, FaceRecognizer, SpeakerEncoder, VideoExtractor  # noqa: F401
    net = get_network("test")
    assert not net.training and get_network("train").training
    assert list(inspect.signature(Lip2Speech.forward).parameters)[:9] == [
        "self", "video_frames", "face_frames", "audio_frames", "melspecs", "video_lengths", "audio_lengths",
        "melspec_lengths", "tf_ratio"]
    assert list(inspect.signature(Lip2Speech.inference).parameters)[:4] == ["self", "video_frames", "face_frames", "speaker_embedding"]
    for attr in ("encoder", "decoder", "vgg_face"):
        assert isinstance(getattr(net, attr), torch.nn.Module)
    hp = hparams.create_hparams()
    assert (hp.max_decoder_steps, hp.n_mel_channels, hp.batch_size, hp.seed, hp.learning_rate) == (300, 80, 64, 1234, 1e-4)
    # optimizer groups as train.py:102-104 builds them
    n = sum(p.numel() for p in net.decoder.parameters()) + sum(p.numel() for p in net.encoder.parameters())
    assert n == 38_436_836
    # strict checkpoint round trip with the reference's key names
    from lip2speech_amd import synth
    sd = synth.synth_state_dict(seed=7)
    net.load_state_dict(sd, strict=True)
    assert torch.equal(net.state_dict()["encoder.trunk.0.4.banch2.3.weight"], sd["encoder.trunk.0.4.banch2.3.weight"])


def test_fresh_model_starts_from_the_reference_initialisers():
    """A new model is initialised like the reference's (video.py:27-43, decoder.py:43-49,78-80,99-100,206,237,289,302; torch defaults) from
    torch's RNG - not from the randomised test fixture - so `torch.manual_seed` decides it."""
    import math

    from model.model import get_network
    torch.manual_seed(1)
    sd = get_network("train").state_dict()
    for k, v in sd.items():
        if k.endswith(("running_mean",)) or (".1.bias" in k and k.startswith("encoder.")):
            assert float(v.abs().max()) == 0.0, k
        if k.endswith("running_var"):
            assert torch.equal(v, torch.ones_like(v)), k
    assert torch.equal(sd["encoder.frontend3D.1.weight"], torch.ones(24)) and torch.equal(sd["encoder.frontend3D.2.weight"], torch.full((24,), 0.25))
    assert torch.equal(sd["decoder.postnet.sin_activation.0.w"], torch.ones(512))
    assert abs(float(sd["decoder.temperature"]) - 512 ** 0.5) < 1e-5 and float(sd["decoder.content.temperature"]) == 16.0
    w = sd["encoder.frontend3D.0.weight"]
    assert abs(float(w.std()) / math.sqrt(2.0 / (5 * 7 * 7 * 24)) - 1) < 0.03
    q = sd["decoder.Q.0.linear_layer.weight"]                      # xavier_uniform, gain 1: bound sqrt(6 / (fan_in + fan_out))
    assert float(q.abs().max()) <= math.sqrt(6.0 / (1024 + 512)) and float(q.abs().max()) > 0.99 * math.sqrt(6.0 / (1024 + 512))
    assert float(sd["decoder.decoder_rnn.weight_hh_l1"].abs().max()) <= 1 / math.sqrt(512)
    assert 0.0 <= float(sd["decoder.content.word_embeddings"].min()) and float(sd["decoder.content.word_embeddings"].max()) < 1.0
    torch.manual_seed(1)
    again = get_network("train").state_dict()
    torch.manual_seed(2)
    other = get_network("train").state_dict()
    assert torch.equal(again["decoder.BOS"], sd["decoder.BOS"]) and not torch.equal(other["decoder.BOS"], sd["decoder.BOS"])


def test_reference_checkpoint_with_face_tower_keys_round_trips_strictly():
    """demo.py:30-38: the checkpoint carries `vgg_face.resnet.*` / `vgg_face.projection_layer.*` (third-party tower) and is loaded with
    strict=True after the `speaker_encoder.*` keys are popped; the keys come back from state_dict() unchanged."""
    from lip2speech_amd import synth
    from model.model import get_network
    ck = dict(synth.synth_state_dict(seed=3))
    ck["vgg_face.resnet.conv2d_1a.conv.weight"] = torch.randn(32, 3, 3, 3)
    ck["vgg_face.resnet.conv2d_1a.bn.num_batches_tracked"] = torch.tensor(7)
    ck["vgg_face.resnet.last_linear.weight"] = torch.randn(512, 1792)
    ck["vgg_face.projection_layer.0.weight"] = torch.randn(512, 512)
    ck["vgg_face.projection_layer.2.bias"] = torch.randn(256)
    net = get_network("test")
    res = net.load_state_dict(ck, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    out = net.state_dict()
    assert set(out) == set(ck) and all(torch.equal(out[k], ck[k]) for k in ck)
    net2 = get_network("test")                                       # and what this model saves loads strictly again
    net2.load_state_dict(out, strict=True)
    with pytest.raises(RuntimeError):
        net.load_state_dict({**ck, "decoder.not_a_key": torch.zeros(1)}, strict=True)


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must raise, not silently compute elsewhere; and the product never imports the oracle."""
    from model.model import get_network
    net = get_network("test")
    with pytest.raises(RuntimeError):
        net.inference(torch.zeros(1, 3, 8, 96, 96), None, speaker_embedding=torch.zeros(1, 256))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lip2speech_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"


def test_bench_refuses_to_report_more_gpus_than_it_sees():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself - and must fail loudly when fewer than N devices are visible
    (this container has none) instead of measuring one GPU and printing `n_gpus: 1`."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing to report" in (r.stdout + r.stderr), r.stdout + r.stderr
    # a launcher / --gpus mismatch is an error too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stdout + r.stderr)
