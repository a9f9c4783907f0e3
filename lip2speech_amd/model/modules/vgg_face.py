"""Face speaker tower placeholder (reference: /root/reference/model/modules/vgg_face.py:12-60).

The reference wraps the third-party ``facenet_pytorch.InceptionResnetV1`` whose architecture and
weights are not part of the reference tree and whose constructor downloads a checkpoint; it is
outside the measured path ("parity unpinned") and is bypassed whenever a
``speaker_embedding`` is supplied (model.py:47-50).  The boundary keeps the attribute and the
``inference`` method so callers keep working; using it without the third-party package raises.
"""
from torch import nn


class FaceRecognizer(nn.Module):
    def __init__(self):
        super().__init__()

    def inference(self, face_frames):
        raise RuntimeError("FaceRecognizer needs the third-party facenet_pytorch tower, which is outside this "
                           "path; pass speaker_embedding=... (the --encoding voice route, demo.py:84-86)")

    forward = inference
