"""rust-raytracer_amd — MI355X-native drop-in for the ray_color hot path of dps/rust-raytracer.

    host  librt_host.so  scene JSON / camera / JPEG / PNG (C++; stays on the CPU)
    hip   librt_hip.so   the gfx950 megakernel behind the C ABI of include/rt_abi.h

The directory name carries a hyphen, so load it with `__graft_entry__.load_package()`
(importlib, module name `rust_raytracer_amd`).  There is NO CPU fallback: `hip` raises
ImportError if the HIP library is missing and RtError if no GPU is visible.
"""
from . import abi, host  # noqa: F401

__all__ = ["abi", "host", "hip"]


def __getattr__(name):
    if name == "hip":
        import importlib
        return importlib.import_module(__name__ + ".hip")
    raise AttributeError(name)
