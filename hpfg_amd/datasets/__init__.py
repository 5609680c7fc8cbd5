"""Data sources behind the reference's ``build_loader(args)`` contract (datasets/builder.py): synthetic ACDC-shaped batches on the
host and the HBM-resident slice pool with on-device augmentation."""
from .builder import build_loader

__all__ = ["build_loader"]
