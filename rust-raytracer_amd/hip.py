"""librt_hip.so — the gfx950 megakernel behind the C ABI (include/rt_abi.h), via ctypes.

This module is the product's only compute path.  It raises ImportError when the HIP library
has not been built and RtError(RT_ERR_NO_DEVICE) when no GPU is visible: there is no CPU
fallback, by design.  torch is used by callers only to own device buffers and streams; this
module passes raw device pointers.
"""
import ctypes as C
import os

from . import abi
from .host import RtError

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_PROBE = None
LIB_PATH = os.path.join(_HERE, "librt_hip.so")
PROBE_LIB_PATH = os.path.join(_HERE, "librt_hip_probe.so")   # the same sources + the device probes / debug calls of include/rt_abi_test.h (tests, tools/diag.py)


def _bind(path, probes):
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing — build it with __graft_entry__.build(); "
                          "there is no CPU fallback for the hot path")
    L = C.CDLL(path)
    if os.environ.get("RT_SKIP_LAYOUT_CHECK"):   # (development: an older library for A/B runs — symbols it lacks bind to a stub)
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    class _Stub:
                        argtypes = restype = None
                    return _Stub()
        L = _Tolerant(L)
    L.rt_hip_device_count.restype = C.c_int
    L.rt_hip_device_warm.argtypes = [C.c_int]
    L.rt_hip_last_error.restype = C.c_char_p
    L.rt_hip_setup_profile.restype = C.c_char_p
    L.rt_strerror.argtypes = [C.c_int]
    L.rt_strerror.restype = C.c_char_p
    L.rt_hip_scene_create.argtypes = [C.POINTER(abi.RtScene), C.c_int, C.POINTER(C.c_void_p)]
    L.rt_hip_scene_destroy.argtypes = [C.c_void_p]
    L.rt_hip_scene_destroy.restype = None
    L.rt_hip_render.argtypes = [C.c_void_p, C.POINTER(abi.RtRowTiles), C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_hip_wait.argtypes = [C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rt_hip_scene_query.argtypes = [C.c_void_p, C.c_char_p]
    L.rt_hip_scene_query.restype = C.c_int64
    L.rt_render_rgb8.argtypes = [C.POINTER(abi.RtScene), C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_set_camera.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4
    L.rt_hip_render_to_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_abi_sizeof.argtypes = [C.c_char_p]
    L.rt_abi_sizeof.restype = C.c_size_t
    L.rt_abi_version.restype = C.c_uint32
    L.rt_hip_group_create.argtypes = [C.POINTER(abi.RtScene), C.c_uint32, C.POINTER(C.c_void_p)]
    L.rt_hip_group_destroy.argtypes = [C.c_void_p]
    L.rt_hip_group_destroy.restype = None
    L.rt_hip_group_size.argtypes = [C.c_void_p]
    L.rt_hip_group_size.restype = C.c_uint32
    L.rt_hip_group_set_camera.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4
    L.rt_hip_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rt_hip_group_render_to_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_group_render.argtypes = [C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_group_submit.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_hip_group_collect.argtypes = [C.c_void_p, C.POINTER(abi.RtStats)]
    L.rt_hip_group_info.argtypes = [C.c_void_p, C.POINTER(abi.RtGroupInfo)]
    L.rt_hip_group_ranks.argtypes = [C.c_void_p, C.POINTER(abi.RtGroupRank), C.c_uint32]
    L.rt_hip_group_ranks.restype = C.c_uint32
    L.rt_hip_group_fallback_reason.argtypes = [C.c_void_p]
    L.rt_hip_group_fallback_reason.restype = C.c_char_p
    L.rt_hip_group_frame.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.rt_hip_group_frame.restype = C.c_void_p
    L.rt_hip_group_stacked_row.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.rt_hip_group_stacked_row.restype = C.c_uint32
    if probes:   # include/rt_abi_test.h
        L.rt_hip_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.rt_hip_debug_tile_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.rt_hip_math_probe.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p]
        L.rt_hip_hit_probe.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p]
        L.rt_hip_atan2_probe.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p]
        L.rt_hip_texel_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.rt_hip_quot_probe.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p]
    for name in (() if os.environ.get("RT_SKIP_LAYOUT_CHECK") else ("RtSphere", "RtTexture", "RtScene", "RtRowTiles", "RtStats", "RtGroupInfo", "RtGroupRank")):   # the binding's own layout check
        if L.rt_abi_sizeof(name.encode()) != C.sizeof(getattr(abi, name)):
            raise ImportError(f"{path}: sizeof({name}) = {L.rt_abi_sizeof(name.encode())} but abi.py has "
                              f"{C.sizeof(getattr(abi, name))} — rebuild with __graft_entry__.build()")
    if L.rt_abi_version() != abi.RT_ABI_VERSION and not os.environ.get("RT_SKIP_LAYOUT_CHECK"):
        raise ImportError(f"{path}: ABI version {L.rt_abi_version()} != {abi.RT_ABI_VERSION}")
    return L


def lib():
    """the PRODUCT library (include/rt_abi.h)"""
    global _LIB
    if _LIB is None:
        _LIB = _bind(LIB_PATH, probes=False)
    return _LIB


def probe_lib():
    """librt_hip_probe.so: the product's sources + the device probes and debug calls of include/rt_abi_test.h — test
    infrastructure.  A scene the debug calls look into must have been created through THIS library: HipScene(..., library=probe_lib())."""
    global _PROBE
    if _PROBE is None:
        _PROBE = _bind(PROBE_LIB_PATH, probes=True)
    return _PROBE


def _check(rc, L=None):
    if rc != abi.RT_OK:
        L = L or lib()
        raise RtError(rc, f"{L.rt_strerror(rc).decode()}: {L.rt_hip_last_error().decode('utf-8', 'replace')}")


def setup_profile():
    """rt_hip_setup_profile: {stage: ms} of the most recent group creation / one-shot render of this process"""
    import json
    return json.loads(lib().rt_hip_setup_profile().decode())


def device_count():
    return lib().rt_hip_device_count()


class HipScene:
    """Scene tables + textures resident in HBM of one GPU (rt_hip_scene_create).  `library`: probe_lib() for the tests and
    tools that use the debug calls; default: the product library."""

    def __init__(self, scene_ptr, device=0, library=None):
        self._L = library or lib()
        self._h = C.c_void_p()
        _check(self._L.rt_hip_scene_create(scene_ptr, device, C.byref(self._h)), self._L)
        sc = scene_ptr.contents
        self.width, self.height = sc.width, sc.height
        self.device = device

    def set_option(self, key, value):
        _check(self._L.rt_hip_set_option(self._h, key.encode(), int(value)), self._L)

    def query(self, key):
        """rt_hip_scene_query: what the resident scene was built into ("grid_cells", "table_bytes", ...); -1 = unknown key"""
        return int(self._L.rt_hip_scene_query(self._h, key.encode()))

    def render(self, d_rgb8, d_linear=0, tiles=None, stream=0):
        """enqueue the megakernel; d_* are raw device pointers (ints), stream a hipStream_t"""
        _check(self._L.rt_hip_render(self._h, C.byref(tiles) if tiles is not None else None,
                                     C.c_void_p(d_rgb8), C.c_void_p(d_linear or None), C.c_void_p(stream or None)), self._L)

    def set_camera(self, origin, lower_left, horizontal, vertical):
        """move the camera of the resident scene (the four vectors of camera.rs:52-63)"""
        v = [(C.c_double * 3)(*x) for x in (origin, lower_left, horizontal, vertical)]
        _check(self._L.rt_hip_set_camera(self._h, *v), self._L)

    def render_to_host(self):
        """whole frame into a numpy [h,w,3] array (blocking) + stats"""
        import numpy as np
        out = np.zeros((self.height, self.width, 3), np.uint8)
        st = abi.RtStats()
        _check(self._L.rt_hip_render_to_host(self._h, out.ctypes.data, C.byref(st)), self._L)
        return out, st.as_dict()

    def wait(self):
        st = abi.RtStats()
        _check(self._L.rt_hip_wait(self._h, C.byref(st)), self._L)
        return st.as_dict()

    def debug_tile_depth(self, cap=1 << 22):
        """(probe library only) deepest camera path per pixel tile of the last measuring frame, as a 2-D array [tile rows, tile columns]"""
        import numpy as np
        buf = np.zeros(cap, np.uint32)
        tx = C.c_uint32(0)
        n = self._L.rt_hip_debug_tile_depth(self._h, buf.ctypes.data, cap, C.byref(tx))
        if n < 0:
            _check(n, self._L)
        return buf[:n].reshape(-1, tx.value) if tx.value and n % tx.value == 0 else buf[:n]

    def debug_timeline(self, max_waves=8192):
        """(probe library only) {start, end, queue-empty time, tail iterations | lane-iterations << 32} of every wave of the last launch, 100 MHz ticks (RT_PROFILE builds of the library only)."""
        import numpy as np
        buf = np.zeros(32 + 4 * max_waves, np.uint64)
        n = self._L.rt_hip_debug_timeline(self._h, buf.ctypes.data, max_waves)
        if n < 0:
            _check(n, self._L)
        return buf[32:32 + 4 * n].reshape(n, 4), buf[:32]

    def close(self):
        if self._h:
            self._L.rt_hip_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipGroup:
    """The scene resident on n_gpus devices of this node, frames sharded by interleaved scanline tiles
    inside librt_hip.so (rt_hip_group_*): host threads + streams + ONE gather per frame, no torch."""

    def __init__(self, scene_ptr, n_gpus=0, library=None):
        self._L = library or lib()     # (library: probe_lib() for the tests that inject transport faults)
        self._h = C.c_void_p()
        _check(self._L.rt_hip_group_create(scene_ptr, n_gpus, C.byref(self._h)), self._L)
        sc = scene_ptr.contents
        self.width, self.height = sc.width, sc.height
        self.size = self._L.rt_hip_group_size(self._h)

    def set_option(self, key, value):
        _check(self._L.rt_hip_group_set_option(self._h, key.encode(), int(value)), self._L)

    def set_camera(self, origin, lower_left, horizontal, vertical):
        v = [(C.c_double * 3)(*x) for x in (origin, lower_left, horizontal, vertical)]
        _check(self._L.rt_hip_group_set_camera(self._h, *v), self._L)

    def render_to_host(self, out=None):
        import numpy as np
        if out is None:
            out = np.zeros((self.height, self.width, 3), np.uint8)
        st = abi.RtStats()
        _check(self._L.rt_hip_group_render_to_host(self._h, out.ctypes.data, C.byref(st)), self._L)
        return out, st.as_dict()

    def render(self):
        """one frame, left in HBM of the group's first device (frame_ptr()); blocking; returns the stats"""
        st = abi.RtStats()
        _check(self._L.rt_hip_group_render(self._h, C.byref(st)), self._L)
        return st.as_dict()

    def submit(self, out=None):
        """enqueue one frame (rt_hip_group_submit); `out`: a numpy uint8 [h,w,3] array the frame is copied into (it must stay
        alive until the frame is collected), or None to leave it in HBM.  Two frames may be in flight."""
        _check(self._L.rt_hip_group_submit(self._h, out.ctypes.data if out is not None else None), self._L)

    def collect(self):
        """wait for the oldest submitted frame (rt_hip_group_collect); returns its stats"""
        st = abi.RtStats()
        _check(self._L.rt_hip_group_collect(self._h, C.byref(st)), self._L)
        return st.as_dict()

    def frame_ptr(self):
        """(device pointer of the assembled RGB8 frame collected last, device ordinal)"""
        dev = C.c_int(0)
        return self._L.rt_hip_group_frame(self._h, C.byref(dev)), dev.value

    def info(self):
        """what the group runs on: ranks, distinct devices, gather transport, RCCL communicators (rt_hip_group_info)"""
        gi = abi.RtGroupInfo()
        _check(self._L.rt_hip_group_info(self._h, C.byref(gi)), self._L)
        return {"n_ranks": gi.n_ranks, "n_devices": gi.n_devices, "transport": ("none", "rccl", "peer")[gi.transport],
                "rccl_comms": gi.rccl_comms, "tile_rows": gi.tile_rows, "pad_rows": gi.pad_rows, "emulated": bool(gi.emulated),
                "transport_fallback": bool(gi.transport_fallback),
                "fallback_reason": (self._L.rt_hip_group_fallback_reason(self._h) or b"").decode("utf-8", "replace"),
                "rank_devices": [gi.device[r] for r in range(min(gi.n_ranks, abi.RT_GROUP_INFO_MAX_RANKS))]}

    def ranks(self):
        """per rank: where it runs (device, PCI id, NUMA node, CPUs its host thread is pinned to, peer access to the first rank's
        device) and its share of the last frame (kernel_ms of the frame collected last; t_wake_us / t_enq_us of the frame submitted last)"""
        n = self._L.rt_hip_group_ranks(self._h, None, 0)
        arr = (abi.RtGroupRank * n)()
        self._L.rt_hip_group_ranks(self._h, arr, n)
        return [{"device": a.device, "pci_bus_id": a.pci_bus_id.decode("ascii", "replace"), "numa_node": a.numa_node, "pinned_cpus": a.pinned_cpus,
                 "peer_to_root": a.peer_to_root, "kernel_ms": a.kernel_ms, "t_wake_us": a.t_wake_us, "t_enq_us": a.t_enq_us} for a in arr]

    def close(self):
        if self._h:
            self._L.rt_hip_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_rgb8(scene_ptr):
    """rt_render_rgb8: host scene in, host RGB8 out (numpy [h,w,3]) + stats"""
    import numpy as np
    sc = scene_ptr.contents
    out = np.zeros((sc.height, sc.width, 3), np.uint8)
    st = abi.RtStats()
    _check(lib().rt_render_rgb8(scene_ptr, out.ctypes.data, C.byref(st)))
    return out, st.as_dict()
