"""kws_amd -- ctypes access to libkws_hip.so (the C ABI declared in include/kws.h).

There is no CPU fallback: importing `kws_amd.lib` without the built library raises, and every
compute entry point raises `KwsError` when no HIP device is present.
"""
from .lib import KwsError, check, get_lib, device_count, version  # noqa: F401
