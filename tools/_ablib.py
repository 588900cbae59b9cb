"""FV_AB_LIB=<path to a library build> for the tuning tools (never for the product): load that build instead of the
tree's, tolerating entry points an older build lacks."""
import ctypes
import os


def use_lib_from_env(_native):
    path = os.environ.get("FV_AB_LIB")
    if not path:
        return

    class Tolerant(ctypes.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if not name.startswith("fv_"):
                    raise
                stub = ctypes.CFUNCTYPE(ctypes.c_int)(lambda: -1)
                setattr(self, name, stub)
                return stub
    _native.LIB_PATH = os.path.abspath(path)
    _native.ctypes.CDLL = Tolerant
