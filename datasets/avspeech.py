"""Drop-in name ``datasets.avspeech`` (reference: datasets/avspeech/__init__.py); the loader itself is out of scope.  The collate is the
per-corpus one: mel targets padded with zeros (datasets/avspeech/dataset.py), unlike the top-level ``train_collate_fn_pad``."""
from lip2speech_amd.datasets.unported import AVSpeech  # noqa: F401
from lip2speech_amd.datasets import av_speech_collate_fn_pad, av_speech_collate_fn_trim  # noqa: F401
