"""Drop-in name for the reference's top-level ``datasets`` package (collates, MelSpectrogram, MelSpec2Audio, FaceAugmentation; the
corpus loaders live in ``datasets.lrw`` / ``.grid`` / ``.avspeech`` / ``.wild`` like the reference's sub-packages).
NOTE: a HuggingFace ``datasets`` wheel is installed in this image - the repository root must come first on sys.path."""
from lip2speech_amd.datasets import (FaceAugmentation, LRW, MelSpec2Audio, MelSpectrogram,  # noqa: F401
                                     test_collate_fn_pad, train_collate_fn_pad)
