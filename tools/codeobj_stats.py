#!/usr/bin/env python3
"""Register / spill / scratch figures of every kernel in librt_hip.so, from the compiler's own metadata.

    python tools/codeobj_stats.py [extra hipcc flags ...]  > profiles/rNN_codeobj.txt

Compiles rt_hip_api.hip for gfx950 with the product's flags (+ extras) and -save-temps into a scratch directory
and prints one line per kernel: VGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, code size, and a static
instruction census of the megakernel instantiations (v_mov share: the copies at control-flow joins).
No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rust-raytracer_amd", "csrc", "hip", "rt_hip_api.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DRT_WAVES_PER_EU=4"]


def demangle_mk(name):
    m = re.match(r"_ZN3rtk13rt_megakernelILb(\d)ELb(\d)ELb(\d)ELb(\d)EEEvNS_5KArgsE", name)
    if m:
        hl, simple, lds, wide = (int(x) for x in m.groups())
        return f"rt_megakernel<lights={hl}, simple_colour={simple}, lds_tables={lds}" + (", wide_tables=1>" if wide else ">")
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return out.split("(")[0] if out else name


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["hipcc", *FLAGS, *extra, "-shared", SRC, "-o", os.path.join(td, "x.so"), "-save-temps"], check=True, cwd=td,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "rt_hip_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    print("# hipcc " + " ".join(FLAGS + extra))
    # metadata block: one YAML record per kernel
    meta = {}
    for rec in re.split(r"\n  - \.", asm[asm.index("amdhsa.kernels:"):]):
        f = dict(re.findall(r"\.?(\w+):\s+(\S+)", rec))
        if "name" in f and "vgpr_count" in f:
            meta[f["name"]] = f
    # static census per function body
    bodies = {}
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\s*s_endpgm", asm, flags=re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    for name, f in meta.items():
        line = (f"{demangle_mk(name):70s} vgpr {int(f['vgpr_count']):3d}  sgpr {int(f['sgpr_count']):3d}  vgpr_spill {int(f['vgpr_spill_count']):3d}  "
                f"sgpr_spill {int(f['sgpr_spill_count']):3d}  scratch {int(f['private_segment_fixed_size']):4d} B/lane")
        b = bodies.get(name)
        if b and "megakernel" in name:
            ins = re.findall(r"^\s+([vsd][a-z0-9_]+|scratch_\w+|global_\w+|flat_\w+|buffer_\w+)\b", b, flags=re.M)
            v = [i for i in ins if i.startswith("v_")]
            line += (f"  | static: {len(ins)} instr, {len(v)} VALU, v_mov {sum(i.startswith('v_mov') for i in v)} ({100.0 * sum(i.startswith('v_mov') for i in v) / max(1, len(v)):.1f} % of VALU), "
                     f"v_cndmask {sum(i.startswith('v_cndmask') for i in v)}, scratch ld/st {sum(i.startswith('scratch_') for i in ins)}")
        print(line)


if __name__ == "__main__":
    main()
