"""ctypes mirror of include/rt_abi.h (plain C ABI; no torch types cross this boundary)."""
import ctypes as C

RT_ABI_VERSION = 5
RT_MAX_LIGHT_NEST = 8
RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP = 0, -1, -2, -3
RT_ERR_IO, RT_ERR_PARSE, RT_ERR_TEXTURE, RT_ERR_PNG, RT_ERR_UNSUPPORTED = -4, -5, -6, -7, -8
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_GLASS, RT_MAT_TEXTURE, RT_MAT_LIGHT = range(5)
RT_SKY_NONE, RT_SKY_GRADIENT, RT_SKY_TEXTURE = range(3)


class RtSphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("fuzz_or_ior", C.c_double),
                ("h_offset", C.c_double), ("tex_w", C.c_uint64), ("tex_h", C.c_uint64),
                ("albedo", C.c_float * 3), ("kind", C.c_uint32), ("tex_id", C.c_uint32),
                ("reserved", C.c_uint32)]


class RtTexture(C.Structure):
    _fields_ = [("rgb8", C.POINTER(C.c_uint8)), ("nbytes", C.c_uint64),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class RtScene(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32), ("sky_mode", C.c_uint32),
                ("cam_origin", C.c_double * 3), ("cam_lower_left", C.c_double * 3),
                ("cam_horizontal", C.c_double * 3), ("cam_vertical", C.c_double * 3),
                ("sky_rgb8", C.POINTER(C.c_uint8)), ("sky_w", C.c_uint64), ("sky_h", C.c_uint64),
                ("spheres", C.POINTER(RtSphere)), ("n_spheres", C.c_uint32), ("n_textures", C.c_uint32),
                ("textures", C.POINTER(RtTexture)), ("seed", C.c_uint64),
                ("n_gpus", C.c_uint32), ("reserved0", C.c_uint32)]


class RtRowTiles(C.Structure):
    _fields_ = [("tile_rows", C.c_uint32), ("first_tile", C.c_uint32), ("tile_stride", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("exact_tests", C.c_uint64), ("tex_oob", C.c_uint64),
                ("kernel_ms", C.c_double), ("frame_ms", C.c_double), ("grid_steps", C.c_uint64), ("wave_iters", C.c_uint64 * 4), ("prof_cycles", C.c_uint64 * 12),
                ("segments_discarded", C.c_uint64), ("n_gpus_used", C.c_uint32), ("segments_repeated", C.c_uint32),
                ("gather_ms", C.c_double), ("setup_ms", C.c_double), ("group_us", C.c_double * 8)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k in ("wave_iters", "prof_cycles", "group_us") else getattr(self, k)) for k, _ in self._fields_}


RT_GROUP_INFO_MAX_RANKS = 64
RT_GATHER_NONE, RT_GATHER_RCCL, RT_GATHER_PEER = range(3)


class RtGroupInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_uint32), ("n_devices", C.c_uint32), ("transport", C.c_uint32), ("rccl_comms", C.c_uint32),
                ("tile_rows", C.c_uint32), ("pad_rows", C.c_uint32), ("emulated", C.c_uint32), ("transport_fallback", C.c_uint32),
                ("device", C.c_int32 * RT_GROUP_INFO_MAX_RANKS)]


class RtGroupRank(C.Structure):
    _fields_ = [("device", C.c_int32), ("numa_node", C.c_int32), ("pinned_cpus", C.c_int32), ("peer_to_root", C.c_int32),
                ("pci_bus_id", C.c_char * 16), ("kernel_ms", C.c_double), ("t_wake_us", C.c_double), ("t_enq_us", C.c_double)]


def tiles_local_rows(height, tiles):
    """rt_tiles_local_rows() of the header."""
    if tiles is None or tiles.tile_rows == 0 or tiles.tile_stride == 0:
        return height
    n_tiles = (height + tiles.tile_rows - 1) // tiles.tile_rows
    rows = 0
    for k in range(tiles.first_tile, n_tiles, tiles.tile_stride):
        rows += min((k + 1) * tiles.tile_rows, height) - k * tiles.tile_rows
    return rows


def tiles_global_rows(height, tiles):
    """global scanline of every local packed row (rt_tiles_global_row for lr in range(local_rows))."""
    if tiles is None or tiles.tile_rows == 0 or tiles.tile_stride == 0:
        return list(range(height))
    out = []
    n_tiles = (height + tiles.tile_rows - 1) // tiles.tile_rows
    for k in range(tiles.first_tile, n_tiles, tiles.tile_stride):
        out.extend(range(k * tiles.tile_rows, min((k + 1) * tiles.tile_rows, height)))
    return out
