"""Build recipe for the HIP extension: csrc/*.hip -> csrc/libbattlespace_hip.so (in-tree, gfx950 only).

    python deep-rl-battlespace_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: plane and bullet positions are float64
add-then-truncate in the reference (envs/sprites.py:130-131,332-333) and a fused multiply-add would round differently.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libbattlespace_hip.so")
COMMON = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# (source, extra flags).  The step path forbids FMA contraction (bit-exact float64 add-then-truncate).  The actor MLP has no
# such contract, but its arithmetic is shared with the fused rollout kernel inside bsx_kernels.hip and must give the same
# bits in both files: one module-wide setting (library code like tanhf is fused or not by it), contraction where wanted
# through `#pragma clang fp contract(fast)` in bsx_actor_core.h.
# -disable-machine-licm: in the multi-tick kernels (a tick loop around the whole step body) machine LICM would hoist every
# fp64 literal of sincos / atan2 out of the loop into ~110 VGPRs held for the whole tick (256 VGPRs, 1-2 waves per SIMD);
# the one-call kernels compile to the same code either way.
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary VGPRs (every kernel here fits 256 of them at two waves per SIMD);
# the default puts them in AGPRs and pays a v_accvgpr_read per value the LayerNorm / head then touches (11 % of the actor).
# -amdgpu-kernarg-preload-count=15: the step kernels' eight leading arguments (game count and the pointers of a wave's first
# loads) arrive in SGPRs with the dispatch instead of through a cold scalar-cache fetch at the start of every wave.
STEP_FLAGS = ["-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-kernarg-preload-count=15"]
# -amdgpu-sched-strategy=max-ilp, the per-call kernels only (round 5): the machine scheduler orders for instruction-level parallelism
# instead of minimal register pressure -- 1v1 66 -> 76 VGPRs (six resident waves instead of seven), a few instructions fewer; same
# results (scheduling only).  Measured as five ALTERNATING runs of product and base variant in one gpurun call (medians; the 1 M figure
# moves by +-5 % from process to process, which a single pair does not survive -- profiles/r05_experiments.json): 65 536 x 4v4 20.42 ->
# 20.23 us, 1 M x 1v1 44.2 -> 43.7, C2 6.09 -> 6.07, 262 144 games unchanged; 2v2 and the bullet-heavy workload +0.6 %.  Small, and free.
# The two-wave 1v1 kernels (bsx_step_two_wave.hip) gain more: per call 5.93 -> 5.66 us at C2, multi-tick 2.57 -> 2.54 us per tick (1.80 -> 1.75
# at 32 768 games).  NOT the one-wave multi-tick kernels (neutral) and NOT the fused rollouts (1v1 18.9 -> 19.8 us per tick, 4v4 84.5 -> 94: their register budget
# is what the default strategy protects).
PER_CALL_FLAGS = STEP_FLAGS + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# bsx_kernels.hip: reset / export / scripted-opponent kernels, launchers, C ABI; the 84 step-kernel instances are four more translation
# units (bsx_step_instances.h), compiled side by side: the build is as long as the largest of them instead of their sum
SOURCES = [(os.path.join(CSRC, "bsx_kernels.hip"), STEP_FLAGS),
           (os.path.join(CSRC, "bsx_step_per_call.hip"), PER_CALL_FLAGS),
           (os.path.join(CSRC, "bsx_step_multi_tick.hip"), STEP_FLAGS),
           (os.path.join(CSRC, "bsx_step_two_wave.hip"), PER_CALL_FLAGS),
           (os.path.join(CSRC, "bsx_step_rollout.hip"), STEP_FLAGS),
           (os.path.join(CSRC, "bsx_actor.hip"), ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"])]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    import glob
    deps = [src for src, _ in SOURCES] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl")) + \
        [os.path.join(INCLUDE, "battlespace_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the extension if it is missing or older than its sources.  Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(item):                                  # the translation units compile side by side (two hipcc processes)
        src, extra = item
        obj = os.path.splitext(src)[0] + ".o"
        cmd = [hipcc, *COMMON, *extra, "-I", INCLUDE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
