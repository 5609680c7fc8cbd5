"""``build_loader(args)``: the reference's dataset factory (datasets/builder.py:9-77) for the hot path.

Adds the keys "synthetic" -> (label_loader, unlabel_loader, test_loader) and "sup_synthetic" -> (train_loader, test_loader),
which honour the same batch contract as "acdc"/"sup_acdc", and "device_synthetic": the slices live in HBM and every batch is
augmented (the reference's RandomGenerator) and assembled by one kernel.  "acdc" / "sup_acdc" read an ACDC-format tree (h5 slices, the reference's list files) through the dependency-free reader
``h5lite`` into the same HBM pool.  The reference's other real-data keys (LIDC, Synapse, ISIC, Building: png / npz trees with albumentations
pipelines) are outside the hot-path build and raise NotImplementedError like an unknown key (builder.py:76-77).
"""
from .synthetic import get_ssl_synthetic_loader, get_synthetic_loader, synth_batch

_REAL = {"lidc", "synapse", "isic", "sup_lidc", "sup_synapse", "sup_isic", "sup_building"}


def _device_synthetic(args, rank):
    """(label_loader, unlabel_loader, None): synthetic slices of ACDC-like varying size resident in HBM, augmented on the device
    (device_pool.RandomGeneratorDevice = the reference's RandomGenerator, datasets/utils.py:99-117)."""
    import numpy as np
    import torch
    from .device_pool import DevicePoolLoader, DeviceSlicePool
    g = np.random.default_rng(1234 + rank)
    n_lab, n_unl = int(getattr(args, "num_labeled", 64)), int(getattr(args, "num_unlabeled", 256))
    slices = []
    for i in range(n_lab + n_unl):
        h, w = 32 * int(g.integers(6, 10)), 32 * int(g.integers(6, 10))      # 192..288, ACDC-like varying slice sizes
        x, y = synth_batch(int(g.integers(1 << 30)), 1, h, w, args.in_channels, args.num_classes, 32)
        slices.append((x[0, 0].numpy(), y[0].numpy()))
    pool = DeviceSlicePool(slices, torch.device(getattr(args, "device", "cuda")))
    size = getattr(args, "train_crop_size", (224, 224))
    lab = DevicePoolLoader(pool, args.batch_size, size, indices=range(n_lab))
    unl = DevicePoolLoader(pool, getattr(args, "unlabel_batch_size", args.batch_size), size, indices=range(n_lab, n_lab + n_unl))
    return lab, unl, None


def build_loader(args, rank: int = 0):
    if args.datasets == "synthetic":
        return get_ssl_synthetic_loader(args, rank)
    if args.datasets == "sup_synthetic":
        return get_synthetic_loader(args, rank)
    if args.datasets == "device_synthetic":
        return _device_synthetic(args, rank)
    if args.datasets == "acdc":          # builder.py:10-17: ACDC h5 slices from disk, resident in HBM, augmented on the device
        from .acdc import get_ssl_acdc_loader
        return get_ssl_acdc_loader(root=args.data_path, train_crop_size=args.train_crop_size, batch_size=args.batch_size,
                                   unlabel_batch_size=args.unlabel_batch_size, label_num=args.label_num, device=getattr(args, "device", "cuda"))
    if args.datasets == "sup_acdc":      # builder.py:45-50
        from .acdc import get_acdc_loader
        return get_acdc_loader(root=args.data_path, train_crop_size=args.train_crop_size, batch_size=args.batch_size, device=getattr(args, "device", "cuda"))
    if args.datasets in _REAL:
        raise NotImplementedError(f"dataset '{args.datasets}' (real-data I/O) is outside the MI355X hot-path build; "
                                  f"use 'synthetic' / 'sup_synthetic' (same batch contract)")
    raise NotImplementedError
