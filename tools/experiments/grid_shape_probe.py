import ctypes as C, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/scenes'); sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/tools')
import __graft_entry__ as graft
import numpy as np, torch, procedural, ab_bench
from fuzz_worlds import big_flat_world_json
os.chdir(ROOT)
pkg = graft.load_package(); abi = pkg.abi
L = ab_bench.bind(os.path.join(ROOT, "build", "ab", "librt_hip_default.so"), abi)
L.rt_abi_version.restype = C.c_uint32
L.rt_hip_scene_query.argtypes = [C.c_void_p, C.c_char_p]; L.rt_hip_scene_query.restype = C.c_int64
worlds = [("flat 2e5", pkg.host.Scene.loads(big_flat_world_json(200000, np.random.default_rng(5), width=640, height=360, spp=16, depth=50, half=224.0))),
          ("flat 6e4", pkg.host.Scene.loads(big_flat_world_json(60000, np.random.default_rng(5), width=640, height=360, spp=16, depth=50, half=122.0))),
          ("cfg5 uniform", pkg.host.Scene.loads(procedural.make_json(width=960, height=540, spp=32, half=50, seed=0)))]
stream = torch.cuda.current_stream().cuda_stream
st = abi.RtStats()
for gn in ("256,1,256", "256,2,256", "255,2,255", "200,2,200", "128,2,128", "256,3,256", "256,4,256", "250,1,250", "128,1,128"):
    os.environ["RT_GRID_N"] = gn
    for name, sc in worlds:
        sc.c.abi_version = L.rt_abi_version()
        hs = C.c_void_p()
        assert L.rt_hip_scene_create(sc.ptr, 0, C.byref(hs)) == 0, L.rt_hip_last_error()
        rgb = torch.zeros((sc.c.height, sc.c.width, 3), dtype=torch.uint8, device="cuda:0")
        ks = []
        for _ in range(3):
            assert L.rt_hip_render(hs, None, rgb.data_ptr(), None, stream) == 0
            assert L.rt_hip_wait(hs, C.byref(st)) == 0
            ks.append(st.kernel_ms)
        q = {k: L.rt_hip_scene_query(hs, k.encode()) for k in ("grid_cells", "grid_items", "grid_large", "grid_wide")}
        print(f"grid {gn:10s} | {name:13s} kernel {min(ks[1:]):8.3f} ms  tests/seg {st.exact_tests / max(1, st.segments):6.2f}  steps/seg {st.grid_steps / max(1, st.segments):5.2f} {q}", flush=True)
        L.rt_hip_scene_destroy(hs)
