"""librt_host.so — scene JSON / camera / JPEG / PNG plumbing (C++), via ctypes.

Mirrors the reference's host side: `serde_json::from_slice::<Config>` (main.rs:15),
`Camera::new` (camera.rs:45-77), `find_lights` (raytracer.rs:220-229), `write_image`
(raytracer.rs:33-42).  No path tracing happens here.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[rt error {code}] {msg}")
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.rt_scene_load_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.rt_scene_load_string.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.rt_scene_get.argtypes = [C.c_void_p]
        L.rt_scene_get.restype = C.POINTER(abi.RtScene)
        L.rt_scene_get_mut.argtypes = [C.c_void_p]
        L.rt_scene_get_mut.restype = C.POINTER(abi.RtScene)
        L.rt_scene_free.argtypes = [C.c_void_p]
        L.rt_scene_free.restype = None
        L.rt_scene_to_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.rt_host_last_error.restype = C.c_char_p
        L.rt_camera_derive.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.rt_camera_derive.restype = None
        L.rt_find_lights.argtypes = [C.POINTER(abi.RtSphere), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
        L.rt_find_lights.restype = C.c_uint32
        L.rt_jpeg_decode_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rt_png_write_rgb8.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rt_jpeg_last_error.restype = C.c_char_p
        L.rt_free.argtypes = [C.c_void_p]
        L.rt_free.restype = None
        _LIB = L
    return _LIB


def _check(rc):
    if rc != abi.RT_OK:
        raise RtError(rc, lib().rt_host_last_error().decode("utf-8", "replace"))


class Scene:
    """An owned `Config` (config.rs:66-75) with its derived camera and decoded textures."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(lib().rt_scene_load_file(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def loads(cls, text):
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(lib().rt_scene_load_string(data, len(data), C.byref(h)))
        return cls(h)

    @property
    def c(self):
        """the RtScene struct (mutable: tests override width/height like raytracer.rs:272-273)"""
        return lib().rt_scene_get_mut(self._h).contents

    @property
    def ptr(self):
        return lib().rt_scene_get_mut(self._h)

    def to_json(self):
        need = C.c_size_t()
        lib().rt_scene_to_json(self._h, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        _check(lib().rt_scene_to_json(self._h, buf, need.value, None))
        return buf.value.decode("utf-8")

    def lights(self):
        sc = self.c
        out = (C.c_uint32 * max(1, sc.n_spheres))()
        n = lib().rt_find_lights(sc.spheres, sc.n_spheres, out, sc.n_spheres)
        return list(out[:n])

    def close(self):
        if self._h:
            lib().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_derive(look_from, look_at, vup, vfov, aspect):
    arr = lambda v: (C.c_double * 3)(*v)
    out = (C.c_double * 13)()
    lib().rt_camera_derive(arr(look_from), arr(look_at), arr(vup), vfov, aspect, out)
    o = list(out)
    return {"origin": o[0:3], "lower_left_corner": o[3:6], "horizontal": o[6:9], "vertical": o[9:12], "focal_length": o[12]}


def jpeg_decode(path):
    import numpy as np
    px = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().rt_jpeg_decode_file(os.fsencode(path), C.byref(px), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().rt_free(px)


def jpeg_decode_mem(data):
    """rt_jpeg_decode_mem: bytes -> numpy [h,w,3]; raises RtError(RT_ERR_TEXTURE) on anything malformed"""
    import numpy as np
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    px = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    rc = lib().rt_jpeg_decode_mem(buf, len(data), C.byref(px), C.byref(w), C.byref(h))
    if rc != abi.RT_OK:
        raise RtError(rc, lib().rt_jpeg_last_error().decode("utf-8", "replace"))
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().rt_free(px)


def png_write(path, rgb8):
    import numpy as np
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    _check(lib().rt_png_write_rgb8(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0]))
