"""`from muse.modeling_ema import EMAModel` of the reference package keeps working: the class lives in muse/ema.py"""
from .ema import EMAModel  # noqa: F401
