from .builder import build_loader
