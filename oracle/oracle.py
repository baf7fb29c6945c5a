"""ctypes binding of oracle/librt_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_INDEP = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib(abi):
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.rt_oracle_render.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtRowTiles), C.c_void_p, C.c_void_p,
                                       C.POINTER(abi.RtStats), C.c_int]
        L.rt_oracle_render_window.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtRowTiles), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.POINTER(abi.RtStats), C.c_int]
        L.rt_oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.rt_oracle_philox4x32_10.restype = None
        L.rt_oracle_sphere_hit.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_refract.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_refract.restype = None
        L.rt_oracle_reflect.argtypes = [C.POINTER(C.c_double)] * 3
        L.rt_oracle_reflect.restype = None
        L.rt_oracle_reflectance.argtypes = [C.c_double, C.c_double]
        L.rt_oracle_reflectance.restype = C.c_double
        L.rt_oracle_camera_new.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_camera_new.restype = None
        L.rt_oracle_get_ray.argtypes = [C.POINTER(abi.RtScene), C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_get_ray.restype = None
        L.rt_oracle_ray_color.argtypes = [C.POINTER(abi.RtScene), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.rt_oracle_ray_color.restype = None
        L.rt_oracle_texture_albedo.argtypes = [C.POINTER(abi.RtSphere), C.POINTER(abi.RtTexture), C.c_double, C.c_double, C.POINTER(C.c_float)]
        L.rt_oracle_texture_albedo.restype = None
        L.rt_oracle_f32_to_u8.argtypes = [C.c_float]
        L.rt_oracle_f32_to_u8.restype = C.c_uint8
        L.rt_oracle_find_lights.argtypes = [C.POINTER(abi.RtSphere), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
        L.rt_oracle_find_lights.restype = C.c_uint32
        L.rt_oracle_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_oracle_draws.restype = None
        L.rt_oracle_threads.restype = C.c_int
        L.rt_oracle_p3_op.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_ray_at.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.rt_oracle_ray_at.restype = None
        L.rt_oracle_atan2_v.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.rt_oracle_atan2_v.restype = None
        L.rt_oracle_atan2.argtypes = [C.c_double, C.c_double]
        L.rt_oracle_atan2.restype = C.c_double
        _LIB = L
    return _LIB


def indep_lib(abi):
    """librt_oracle_indep.so: the restatement with the light-sampling draw on a Philox block of its own (round 3's addressing);
    an independent reference for STATISTICS of the shipped addressing (tests/test_light_draw_statistics.py), never for parity"""
    global _INDEP
    if _INDEP is None:
        path = os.path.join(_HERE, "librt_oracle_indep.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.rt_oracle_render_window.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtRowTiles), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.POINTER(abi.RtStats), C.c_int]
        _INDEP = L
    return _INDEP


def render(abi, scene_ptr, tiles=None, n_threads=0, want_linear=True, x_range=None, independent_light_draw=False):
    """-> (rgb8 [rows,w,3] u8, linear [rows,w,3] f32 | None, stats dict); x_range = (x0, x1): only those pixels
    of the rows are rendered (the rest of the arrays stays 0)"""
    sc = scene_ptr.contents
    rows = abi.tiles_local_rows(sc.height, tiles)
    rgb = np.zeros((rows, sc.width, 3), np.uint8)
    lin = np.zeros((rows, sc.width, 3), np.float32) if want_linear else None
    st = abi.RtStats()
    x0, x1 = x_range if x_range is not None else (0, sc.width)
    rc = (indep_lib(abi) if independent_light_draw else lib(abi)).rt_oracle_render_window(scene_ptr, C.byref(tiles) if tiles is not None else None, x0, x1, rgb.ctypes.data,
                                          lin.ctypes.data if lin is not None else None, C.byref(st), n_threads)
    if rc != 0:
        raise RuntimeError(f"rt_oracle_render failed: {rc}")
    return rgb, lin, st.as_dict()
