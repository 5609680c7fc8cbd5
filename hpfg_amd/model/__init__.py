from .builder import build_model
from .segformer import SegFormer
from .unet import UNet, UNet_Plus, reset_dropout_streams
