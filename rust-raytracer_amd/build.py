"""In-tree build of the native pieces (no cmake; plain g++ / hipcc command lines)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra"]
# RT_WAVES_PER_EU=4: cap the megakernel at 128 VGPRs (4 waves/SIMD); measured best of 2/3/4/5 (profiles/r01_run2_ab.log)
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DRT_WAVES_PER_EU=4"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)


def _srcs(*rel):
    return [os.path.join(HERE, r) for r in rel]


def build_host(force=False):
    out = os.path.join(HERE, "librt_host.so")
    src = _srcs("csrc/host/scene.cpp", "csrc/host/jpeg.cpp")
    deps = src + _srcs("csrc/host/json.hpp") + [os.path.join(ROOT, "include/rt_abi.h")]
    if force or _newer(out, deps):
        _run(["g++", *CXXFLAGS, "-shared", *src, "-o", out, "-lz", "-lpthread"])
    return out


def build_hip(force=False):
    """librt_hip.so = the product (include/rt_abi.h, nothing else exported); librt_hip_probe.so = the same sources with
    -DRT_TEST_PROBES: + the device probes and debug calls of include/rt_abi_test.h, for tests/ and tools/diag.py only"""
    out = os.path.join(HERE, "librt_hip.so")
    probe = os.path.join(HERE, "librt_hip_probe.so")
    src = _srcs("csrc/hip/rt_hip_api.hip")
    deps = _srcs(*HIP_DEPS) + [os.path.join(ROOT, "include/rt_abi.h"), os.path.join(ROOT, "include/rt_abi_test.h")]
    # (beside the file times: the source hash the libraries on disk were built from — a checkout or a clock that does not move
    #  forward leaves times that say nothing)
    try:
        import json
        built_from = json.load(open(os.path.join(HERE, "BUILD_INFO.json"))).get("kernel_src_hash")
    except Exception:
        built_from = None
    if os.path.exists(os.path.join(ROOT, ".git")) and built_from != kernel_src_hash():
        force = True
    jobs = []
    if force or _newer(out, deps):
        jobs.append(["hipcc", *HIPFLAGS, "-shared", *src, "-o", out])
    if force or _newer(probe, deps):
        jobs.append(["hipcc", *HIPFLAGS, "-DRT_TEST_PROBES", "-shared", *src, "-o", probe])
    procs = []
    for cmd in jobs:   # (side by side: each is ~20 s of one core)
        print("+", " ".join(cmd), file=sys.stderr, flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return out


def build_cli(force=False):
    """the `raytracer <config_file> <output_file>` binary (reference main.rs)"""
    out = os.path.join(HERE, "raytracer")
    src = _srcs("csrc/host/main.cpp")
    deps = src + [os.path.join(HERE, "librt_host.so"), os.path.join(HERE, "librt_hip.so")]
    if os.path.exists(src[0]) and (force or _newer(out, deps)):
        _run(["g++", *CXXFLAGS, *src, "-o", out, "-L" + HERE, "-lrt_host", "-lrt_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    return out


HIP_DEPS = ("csrc/hip/rt_hip_api.hip", "csrc/hip/rt_kernel.hip", "csrc/hip/rt_hip_group.hip", "csrc/hip/rt_core.h", "csrc/hip/rt_tables.h",
            "csrc/common/rt_atan2.h")


def kernel_src_hash():
    """sha1 over the sources librt_hip.so is compiled from: what a counter file and a bench run must share to describe the
    same kernel (a commit id changes with every documentation commit; this does not)"""
    import hashlib
    h = hashlib.sha1()
    for f in _srcs(*HIP_DEPS) + [os.path.join(ROOT, "include/rt_abi.h")]:
        h.update(open(f, "rb").read())
    h.update(" ".join(HIPFLAGS).encode())
    return h.hexdigest()[:12]


def write_build_info():
    """BUILD_INFO.json next to the libraries: the commit they were built from.  The GPU box gets a snapshot without .git,
    so bench.py and the PMC tools read the head from here (git-ignored, travels with the .so files)."""
    import json
    import time

    def git(*a):
        try:
            return subprocess.run(["git", *a], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip()
        except Exception:
            return ""
    head = git("rev-parse", "--short", "HEAD")
    if not head:
        return  # (not a git checkout: keep whatever file travelled with the snapshot)
    dirty = bool(git("status", "--porcelain", "--untracked-files=no"))
    info = {"git_head": head + ("+dirty" if dirty else ""), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
            "kernel_src_hash": kernel_src_hash()}
    with open(os.path.join(HERE, "BUILD_INFO.json"), "w") as f:
        json.dump(info, f)


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_cli(force)
    write_build_info()


if __name__ == "__main__":
    build_all("--force" in sys.argv)
