"""Drop-in name ``datasets.avspeech`` (reference: datasets/avspeech/__init__.py); the loader itself is out of scope."""
from lip2speech_amd.datasets.unported import AVSpeech  # noqa: F401
from lip2speech_amd.datasets import train_collate_fn_pad as av_speech_collate_fn_pad  # noqa: F401
