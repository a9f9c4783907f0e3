"""The import lines of the reference's caller scripts resolve against this repository (`model.model`, `hparams`,
`datasets`, `datasets.{lrw,grid,avspeech,wild}`, `train_utils.losses`), with the repository root first on sys.path as INTEGRATION.md
prescribes (a HuggingFace `datasets` wheel is installed in this image).  The lines are quoted from /root/reference/train.py:21-30,
evaluate.py:5-14 and demo.py:8-18; what is NOT in them are the reference's own caller-side files (`evaluate`, `arg_parser`, `logger`,
`train_utils.tensorboard_logger`) and third-party packages (apex, pystoi, cv2, sounddevice, soundfile) - out of scope, SURVEY.md §2."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRAIN_PY = """
from datasets import train_collate_fn_pad, FaceAugmentation
from datasets.grid import GRID
from datasets.wild import WILD
from datasets.lrw import LRW
from datasets.avspeech import AVSpeech
from train_utils.losses import *
from model import model
from hparams import create_hparams
"""
EVALUATE_PY = """
from datasets import MelSpec2Audio
from hparams import create_hparams
from torch.utils.data import DataLoader
from model import model
from datasets.lrw import LRW
from datasets.grid import GRID
from datasets.avspeech import AVSpeech
from datasets.wild import WILD
from datasets import train_collate_fn_pad
"""
DEMO_PY = """
from torch.utils.data import DataLoader
from model import model
from model.modules import SpeakerEncoder
from datasets import MelSpec2Audio
from hparams import create_hparams
from datasets.grid import GRID
from datasets.avspeech import AVSpeech
from datasets.lrw import LRW
from datasets.wild import WILD
from datasets import train_collate_fn_pad, test_collate_fn_pad
"""
CHECKS = """
import torch
assert callable(model.get_network) and callable(create_hparams)
hp = create_hparams()
assert hp.sampling_rate == 16000 and hp.max_decoder_steps == 300
aug = FaceAugmentation() if 'FaceAugmentation' in dir() else None
if aug is not None:
    torch.manual_seed(0)
    frames = [torch.arange(12.).view(1, 3, 4), torch.ones(1, 3, 4)]
    outs = [aug(frames) for _ in range(16)]
    flipped = [o for o in outs if not torch.equal(torch.as_tensor(o[0]), frames[0])]
    assert flipped and all(torch.equal(torch.as_tensor(o[0]), frames[0].flip(-1)) for o in flipped)     # hflip, with probability 1 - p
    assert len(flipped) < 16
    assert 'Loss' in dir() and isinstance(Loss(), torch.nn.Module) and Loss().attention_mask.shape == (1, 77)
for cls in (GRID, AVSpeech, WILD):
    try:
        cls('/nonexistent')
    except NotImplementedError as e:
        assert 'out of scope' in str(e)
    else:
        raise AssertionError('file-I/O loaders must say they are out of scope')
assert LRW.__init__.__code__.co_varnames[:7] == ('self', 'rootpth', 'face_size', 'mode', 'demo', 'duration', 'face_augmentation')
print('imports ok')
"""


def _run(src):
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", src + CHECKS], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "imports ok" in r.stdout, r.stdout + r.stderr


def test_train_py_import_lines():
    _run(TRAIN_PY)


def test_evaluate_py_import_lines():
    _run(EVALUATE_PY)


def test_demo_py_import_lines():
    _run(DEMO_PY)
