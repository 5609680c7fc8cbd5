"""``build_model(args)``: the reference's string-keyed model factory (model/builder.py:14-62) for the hot-path architectures.

Keys handled here: "unet" -> UNet, "unet_plus" -> UNet_Plus ("unet_lidc" is the same graph in the reference: model/unet_LIDC.py
differs from model/unet.py by whitespace only), "segformer" -> SegFormer-B0 (first version, see model/segformer.py).  Every other key of the reference's factory is outside this build's scope and
raises NotImplementedError exactly like an unknown key does there (builder.py:59-60).
"""
from .segformer import SegFormer
from .unet import UNet, UNet_Plus


def build_model(args):
    if args.model in ("unet", "unet_lidc"):
        return UNet(in_channels=args.in_channels, num_classes=args.num_classes)
    if args.model == "unet_plus":
        return UNet_Plus(in_channels=args.in_channels, num_classes=args.num_classes)
    if args.model == "segformer":
        return SegFormer(image_size=args.train_crop_size, in_channels=args.in_channels, num_classes=args.num_classes)
    raise NotImplementedError(f"model '{args.model}' is outside the MI355X hot-path build (see DESIGN.md, scope)")
