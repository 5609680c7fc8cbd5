from .builder import build_model
from .unet import UNet, UNet_Plus
