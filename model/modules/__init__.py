from lip2speech_amd.model.modules import VideoExtractor, FaceRecognizer, Decoder, SpeakerEncoder  # noqa: F401
