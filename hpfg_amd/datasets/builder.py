"""``build_loader(args)``: the reference's dataset factory (datasets/builder.py:9-77) for the hot path.

Adds the keys "synthetic" -> (label_loader, unlabel_loader, test_loader) and "sup_synthetic" -> (train_loader, test_loader),
which honour the same batch contract as "acdc"/"sup_acdc".  The reference's real-data keys need h5py / albumentations and the
ACDC/LIDC files, none of which exist on the build or GPU boxes; they raise NotImplementedError like an unknown key
(builder.py:76-77) with a pointer to the synthetic equivalents.
"""
from .synthetic import get_ssl_synthetic_loader, get_synthetic_loader

_REAL = {"acdc", "lidc", "synapse", "isic", "sup_lidc", "sup_acdc", "sup_synapse", "sup_isic", "sup_building"}


def build_loader(args, rank: int = 0):
    if args.datasets == "synthetic":
        return get_ssl_synthetic_loader(args, rank)
    if args.datasets == "sup_synthetic":
        return get_synthetic_loader(args, rank)
    if args.datasets in _REAL:
        raise NotImplementedError(f"dataset '{args.datasets}' (real-data I/O) is outside the MI355X hot-path build; "
                                  f"use 'synthetic' / 'sup_synthetic' (same batch contract)")
    raise NotImplementedError
