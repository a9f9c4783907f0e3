"""Same export list as the reference's model/modules/__init__.py:1-4 (minus the dead wav2vec pieces)."""
from .video import VideoExtractor  # noqa: F401
from .decoder import Decoder  # noqa: F401
from .vgg_face import FaceRecognizer  # noqa: F401
from .audio import SpeakerEncoder  # noqa: F401
