#!/usr/bin/env python3
"""GPU box: does the MEMORY TYPE of the state block and the outputs change what a kernel boundary costs?  The step path's stores leave
6.5 MB dirty in the L2 at the end of every launch (written back at the boundary) and its first loads miss the L2 that the boundary
invalidated.  Here the same job (65 536 x 1v1, raw C-ABI launches, HIP events over 2 000 steps) runs with the state and the outputs in
ordinary device memory (hipMalloc), fine-grained (hipDeviceMallocFinegrained) and uncached (hipDeviceMallocUncached) device memory."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from deep_rl_battlespace_amd import _lib

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
L = _lib.load()
E, n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 1
A, D = 2 * n, 3 * n + 2
torch.cuda.init(); torch.zeros(1, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
nbytes = ctypes.c_size_t()
assert L.bsx_state_bytes(E, n, ctypes.byref(nbytes)) == 0
acts = torch.randint(0, 4, (100, E, A), dtype=torch.int32, device="cuda")
cfg = _lib.BsxRewards(100, 10, -1, -5, -20)


def alloc(size, flag):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), (size + 255) // 256 * 256, flag)
    assert rc == 0, rc
    return p.value


def run(flag_state, flag_out, steps=2000):
    st = alloc(nbytes.value, flag_state)
    obs, rew, done, ed, wn = alloc(E * A * D * 4, flag_out), alloc(E * A * 4, flag_out), alloc(E * A, flag_out), alloc(E, flag_out), alloc(E, flag_out)
    assert L.bsx_state_init(st, E, n, stream) == 0
    assert L.bsx_reset(st, E, n, None, None, 1234, 1, 0, obs, stream) == 0

    def step(t):
        rc = L.bsx_step_discrete(st, E, n, acts[t % 100].data_ptr(), 0, None, obs, rew, done, ed, wn, ctypes.byref(cfg), _lib.F_AUTO_RESET, 1234, 0, stream)
        assert rc == 0, rc
    for t in range(400):
        step(t)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for t in range(200):
            step(t)
        e0.record()
        for t in range(steps):
            step(t)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / steps * 1e3)
    for p in (st, obs, rew, done, ed, wn):
        hip.hipFree(p)
    return sorted(best)[1]


names = {0: "ordinary", 1: "fine-grained", 3: "uncached"}
for fs, fo in ((0, 0), (1, 0), (0, 1), (1, 1), (3, 0), (0, 3), (3, 3), (0, 0)):
    print(f"state {names[fs]:12s} outputs {names[fo]:12s}: {run(fs, fo):7.3f} us per step", flush=True)
