"""Drop-in name for the reference's top-level ``hparams`` module (hparams.py:1-101)."""
from lip2speech_amd.hparams import create_hparams  # noqa: F401
