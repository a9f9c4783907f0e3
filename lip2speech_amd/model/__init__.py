from .model import Lip2Speech, get_network  # noqa: F401
