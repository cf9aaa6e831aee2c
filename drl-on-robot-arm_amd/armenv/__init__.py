"""armenv -- MI355X-native batched robot-arm environments behind the gym-style Env API of
Shimly-2/DRL-on-robot-arm.  Host-side mirror of the reference's interface for the env hot path;
the arithmetic lives in libarmenv.so (HIP, gfx950) behind include/armenv.h."""
from . import _lib
from ._lib import ArmEnvError
from .config import opt
from .spaces import Box
from . import urdf

__all__ = ["ArmEnvError", "opt", "Box", "urdf", "envs"]


def __getattr__(name):
    if name == "envs":           # lazy: envs imports torch
        import importlib
        return importlib.import_module(".envs", __name__)
    raise AttributeError(name)
