"""Drop-in name ``datasets.lrw`` (reference: datasets/lrw/__init__.py)."""
from lip2speech_amd.datasets.lrw import LRW  # noqa: F401
from lip2speech_amd.datasets import train_collate_fn_pad  # noqa: F401
