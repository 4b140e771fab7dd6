"""MI355X-native volume-rendering hot path with the EvenNICER-SLAM Renderer / decoder interface."""
from . import _lib, common, decoder, functional, losses, mapper, renderer, scene, tracker, event  # noqa: F401
from .decoder import NICE, get_model  # noqa: F401
from .renderer import Renderer  # noqa: F401
from ._lib import EnslamError, LIB_PATH  # noqa: F401
