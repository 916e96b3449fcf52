#!/usr/bin/env python3
"""Build a profiling VARIANT of the HIP library next to (never over) the product build:

    python tools/build_variant.py <name> [extra hipcc flags for bsx_kernels.hip ...]
        -> deep-rl-battlespace_amd/csrc/variants/lib_<name>.so

    BSX_LIB_PATH=deep-rl-battlespace_amd/csrc/variants/lib_<name>.so [BSX_ALLOW_DIAG=1] python bench.py ...

Same sources and base flags as deep-rl-battlespace_amd/build.py; the extra flags select what differs (-DBSX_DIAG=<bits>,
-DBSX_STAMPS -- see csrc/bsx_diag.h, which only these builds include --, -mllvm ...).  A variant whose results are not the reference's reports that through
bsx_build_flags() and the binding refuses it without BSX_ALLOW_DIAG=1.  hipcc cross-compiles without a GPU, so variants
are built in the build container and travel to the GPU box with the snapshot (csrc/variants/*.so is git-ignored).
A variant is ONE translation unit (bsx_kernels.hip carries every step-kernel instance) compiled with build.py's STEP_FLAGS: it does not
have the per-call unit's `-amdgpu-sched-strategy=max-ilp`.  Compare a variant with another variant (`build_variant.py base` = no extra
flag), or pass `-mllvm -amdgpu-sched-strategy=max-ilp` to both; against the product only where that flag is neutral (C2, 2v2 ... 4v4)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_bsx_build", os.path.join(ROOT, "deep-rl-battlespace_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def build_variant(name, extra, drop=()):
    out_dir = os.path.join(B.CSRC, "variants")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src, flags in B.SOURCES:
        obj = os.path.join(out_dir, f"{name}_{os.path.basename(src)}.o")
        fl = [f for f in flags if f not in drop]
        # -DBSX_VARIANT: bsx_kernels.hip takes its diagnostic switches from csrc/bsx_diag.h instead of the product constants
        # bsx_kernels.hip carries every step-kernel instance in a variant build; the three instance units compile to nothing
        step_tu = os.path.basename(src).startswith(("bsx_kernels", "bsx_step_"))
        more = ["-DBSX_VARIANT", *extra] if step_tu else []
        subprocess.run([hipcc, *B.COMMON, *fl, *more, "-I", B.INCLUDE, "-c", src, "-o", obj], check=True)
        objs.append(obj)
    lib = os.path.join(out_dir, f"lib_{name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    for o in objs:
        os.remove(o)
    return lib


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    print(build_variant(sys.argv[1], sys.argv[2:]))
