#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of a gfx950 code object, from the assembly hipcc emits (runs in the build
container: hipcc cross-compiles without a GPU).

    python tools/isa_meta.py                 # compiles csrc/bsx_kernels.hip and csrc/bsx_actor.hip with the product flags
    python tools/isa_meta.py --json out.json # ... and writes the table as JSON (profiles/rNN_isa_meta.json)
    python tools/isa_meta.py file.s ...      # reads existing assembly files (hipcc --save-temps)
    python tools/isa_meta.py --diff a.s b.s  # are two builds the same kernel bits? (instruction streams compared per kernel; a.s / b.s may be comma-separated lists)

Columns: VGPRs, spilled VGPRs, SGPRs, spilled SGPRs, scratch bytes per lane, LDS bytes per workgroup.  The step kernel's
template arguments are shown as <N, CONT, MULTI, ACTOR, LG, OFF32> with T/F for the booleans."""
import importlib.util
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = (".vgpr_count", ".vgpr_spill_count", ".sgpr_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size")


def short_name(mangled):
    m = re.match(r"_ZN(?:12_GLOBAL__N_1|4bsxk)15bsx_step_kernelILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])ELb([01])E", mangled)
    if m:
        return "bsx_step_kernel<%s,%s>" % (m.group(1), ",".join("FT"[int(b)] for b in m.groups()[1:]))
    m = re.match(r"_ZN(?:12_GLOBAL__N_1|4bsxk)21bsx_step_split_kernelILb([01])ELb([01])ELi(\d)ELb([01])ELb([01])E", mangled)
    if m:
        g = m.groups()
        return "bsx_step_split_kernel<%s,%s,%s,%s,%s>" % ("FT"[int(g[0])], "FT"[int(g[1])], g[2], "FT"[int(g[3])], "FT"[int(g[4])])     # <LG, OFF32, MANY (0 per call, 1 / 2 multi-tick forms), CONT, DRAW>
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", mangled)
    if m:
        n = int(m.group(1))
        rest = mangled[len(m.group(0)):]
        name, tail = rest[:n], rest[n:]
        t = re.match(r"ILi(\d+)E", tail)
        return name + (f"<{t.group(1)}>" if t else "")
    return mangled


def parse_meta(path):
    """-> {kernel: {key: int}} from the amdhsa.kernels metadata block of an assembly file."""
    out, cur = {}, {}
    for line in open(path, errors="replace"):
        s = line.strip()
        if s.startswith(".name:"):
            cur["name"] = s.split(":", 1)[1].strip()
        for k in KEYS:
            if s.startswith(k + ":"):
                cur[k] = int(s.split(":", 1)[1])
        if s.startswith(".wavefront_size:") or s.startswith("- .agpr_count") or s.startswith("- .args"):
            if "name" in cur and all(k in cur for k in KEYS):
                out[cur["name"]] = {k: cur[k] for k in KEYS}
                cur = {}
    if "name" in cur and all(k in cur for k in KEYS):
        out[cur["name"]] = {k: cur[k] for k in KEYS}
    return out


def parse_streams(path):
    """-> {kernel: [instruction lines]} (labels, comments and directives dropped)."""
    out, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            out[cur] = []
            continue
        s = line.split(";", 1)[0].strip()
        if cur is None or not s or s.startswith(".") or s.endswith(":"):
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
                cur = None if s.startswith(".section") else cur
            continue
        out[cur].append(re.sub(r"\s+", " ", s))
    return {k: v for k, v in out.items() if v}


def compile_product(tmp):
    spec = importlib.util.spec_from_file_location("_bsx_build", os.path.join(ROOT, "deep-rl-battlespace_amd", "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        src, extra = job
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *B.COMMON, *extra, "-I", B.INCLUDE, "--cuda-device-only", "-S", src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(max_workers=4) as ex:            # the translation units compile side by side, as in build.py
        return list(ex.map(one, B.SOURCES))


def main(argv):
    if argv and argv[0] == "--diff":
        def norm(streams):                                   # kernels by readable name; symbol spellings of the two namespaces made equal
            def line(i):                                     # (block labels carry the function's index in ITS translation unit; a static's name an L)
                i = re.sub(r"_ZN(?:12_GLOBAL__N_1|4bsxk)L?", "_ZN#", i)
                return re.sub(r"\.LBB\d+_", ".LBB#_", i)
            return {short_name(k): [line(i) for i in v] for k, v in streams.items()}
        a, b = {}, {}
        for f in argv[1].split(","):
            a.update(norm(parse_streams(f)))
        for f in argv[2].split(","):
            b.update(norm(parse_streams(f)))
        bad = 0
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                bad += 1
                print("DIFFERS" if k in a and k in b else "ONLY IN " + ("A" if k in a else "B"), k, len(a.get(k, [])), len(b.get(k, [])))
        print(f"{len(set(a) & set(b))} kernels in both, {bad} differ")
        return 1 if bad else 0
    js = None
    if argv and argv[0] == "--json":
        js, argv = argv[1], argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        files = argv or compile_product(tmp)
        meta = {}
        for f in files:
            meta.update(parse_meta(f))
    rows = sorted((short_name(k), v) for k, v in meta.items())
    print(f"{'kernel':44s} {'VGPR':>5s} {'spill':>5s} {'SGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>6s}")
    for name, v in rows:
        print(f"{name:44s} " + " ".join(f"{v[k]:>{w}d}" for k, w in zip(KEYS, (5, 5, 5, 5, 7, 6))))
    if js:
        json.dump({name: {k.lstrip("."): v[k] for k in KEYS} for name, v in rows}, open(js, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
