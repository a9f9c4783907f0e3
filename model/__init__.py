"""Drop-in name for the reference's top-level ``model`` package: ``from model.model import get_network``."""
