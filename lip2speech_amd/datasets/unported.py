"""Names of the reference's other corpus loaders (reference: /root/reference/datasets/grid/dataset.py:74,
datasets/avspeech/dataset.py:99-100, datasets/wild/dataset.py:79-80).

Their bodies are video/audio file I/O, face detection and alignment over cv2 / torchvision / face_alignment - outside
the hot path this repository builds (SURVEY.md §2 "OUT OF SCOPE").  The classes exist so that the
reference's caller scripts import unchanged; constructing one says what is missing instead of failing later.  Anything
that yields the `(mouth (T,3,96,96), speech (1,N), melspec (80,M), face_crop (2,3,160,160))` items of
`datasets.lrw.LRW` feeds the same collates and the same model path (variable T is covered by the GRID / AVSpeech shaped
parity cases in tests/).
"""
from __future__ import annotations

from torch.utils.data import Dataset


class _FileIODataset(Dataset):
    corpus = "?"

    def __init__(self, rootpth, face_size=(96, 96), mode="train", demo=False, duration=1, face_augmentation=None, *args, **kwargs):
        raise NotImplementedError(
            f"datasets.{type(self).__name__}: the {self.corpus} loader is file I/O + face alignment (cv2, torchvision, face_alignment) "
            "and is out of scope of the MI355X hot path (SURVEY.md §2). Feed the model through datasets.lrw.LRW or any Dataset "
            "yielding (mouth (T,3,96,96), speech (1,N), melspec (80,M), face_crop (2,3,160,160)) items with train_collate_fn_pad.")


class GRID(_FileIODataset):
    corpus = "GRID"


class AVSpeech(_FileIODataset):
    corpus = "AVSpeech"


class WILD(_FileIODataset):
    corpus = "in-the-wild"
