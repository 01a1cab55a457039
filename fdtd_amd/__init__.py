"""MI355X-native FDTD time-stepping engine: ctypes front end over the gfx950 HIP library."""
from .engine import Engine2D, Engine3D, EngineError, LIB_PATH, lib  # noqa: F401
