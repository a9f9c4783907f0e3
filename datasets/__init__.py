"""Drop-in name for the reference's top-level ``datasets`` package (collates, MelSpectrogram, LRW).
NOTE: a HuggingFace ``datasets`` wheel is installed in this image - the repository root must come first on sys.path."""
from lip2speech_amd.datasets import LRW, MelSpec2Audio, MelSpectrogram, test_collate_fn_pad, train_collate_fn_pad  # noqa: F401
