from .unet_model import UNet, UNet_S, UNet_T, UNetDepth  # noqa: F401
from .unet_parts import DoubleConv, Down, OutConv, Up  # noqa: F401
