"""Drop-in name ``datasets.wild`` (reference: datasets/wild/__init__.py); the loader itself is out of scope."""
from lip2speech_amd.datasets.unported import WILD  # noqa: F401
from lip2speech_amd.datasets import train_collate_fn_pad as av_speech_collate_fn_pad  # noqa: F401
