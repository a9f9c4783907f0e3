"""Drop-in name ``datasets.wild`` (reference: datasets/wild/__init__.py); the loader itself is out of scope.  The collate is the
per-corpus one: mel targets padded with zeros (datasets/wild/dataset.py), unlike the top-level ``train_collate_fn_pad``."""
from lip2speech_amd.datasets.unported import WILD  # noqa: F401
from lip2speech_amd.datasets import av_speech_collate_fn_pad  # noqa: F401
