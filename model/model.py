from lip2speech_amd.model.model import Lip2Speech, get_network, device  # noqa: F401
