"""tribe_hip: ctypes binding + torch front end of libtribe_hip.so (gfx950 HIP kernels)."""
from . import _lib, ops  # noqa: F401
from ._lib import TribeHipError, lib  # noqa: F401
