"""cartslam -- Python plumbing around the MI355X dense-stereo engine (C ABI: include/cart_engine.h).

Only device-memory/stream plumbing, the synthetic scene generator and the frame-sharded batch driver
live here; all arithmetic is in the HIP library (cart-slam_amd/csrc).  Nothing here imports oracle/.
"""
from . import _lib, synth  # noqa: F401
from ._lib import EngineParams, PlaneParams, SuperpixelParams  # noqa: F401
from .engine import INVALID, DevicePlaneSchedule, Engine, EngineError, Superpixels, find_peaks, find_plane_params  # noqa: F401
