"""Diagnostic (GPU box): build a stamped library copy, run C2 steps, print per-phase shares of a wave's lifetime."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
# the stamped build is a VARIANT next to the product library (tools/build_variant.py), selected through BSX_LIB_PATH for this
# process only -- the product file is never renamed or overwritten, so a kill or timeout here cannot leave a wrong library behind
V = os.path.join(ROOT, "deep-rl-battlespace_amd/csrc/variants", os.environ.get("BSX_STAMPS_LIB", "lib_stamps.so"))   # another stamped variant: BSX_STAMPS_LIB=lib_<name>.so
if not os.path.exists(V):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools/build_variant.py"), "stamps", "-DBSX_STAMPS"], check=True)
os.environ["BSX_LIB_PATH"], os.environ["BSX_ALLOW_DIAG"] = V, "1"
FINE = "fine" in os.path.basename(V)                                   # a -DBSX_STAMPS_FINE variant: stamps 3..6 sit inside the shot phase
if True:
    import deep_rl_battlespace_amd as bsx
    from deep_rl_battlespace_amd import _lib
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cont = len(sys.argv) > 3 and sys.argv[3] == 'cont'
    mode = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] in ("many", "rollout") else "step"
    env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1, continuous_actions=cont, one_wave=bool(os.environ.get("BSX_STAMPS_ONE_WAVE")))
    env.reset()
    L = _lib.load()
    G = 2 if n == 1 else (4 if n == 2 else 8)
    waves = (E * G + 63) // 64
    split = mode == "step" and n == 1 and not cont and E <= 114688 and not os.environ.get("BSX_STAMPS_ONE_WAVE")   # the wave-specialised kernel: two rows per workgroup
    if split:
        waves *= 2
    if mode == "rollout":
        waves = (E + 31) // 32                      # the fused kernel's workgroup = 32 games; stamps are indexed by workgroup (its waves share a row)
    buf = torch.zeros(waves * 10, dtype=torch.int64, device="cuda")
    L.bsx_debug_set_stamps.argtypes = [ctypes.c_void_p]
    acts = (torch.rand((64, E, 2 * n, 3), device='cuda') * 2 - 1) if cont else torch.randint(0, 4, (64, E, 2 * n), dtype=torch.int32, device="cuda")
    for t in range(60):
        env.step_batch(acts[t])
    torch.cuda.synchronize()
    assert L.bsx_debug_set_stamps(buf.data_ptr()) == 0
    tot = np.zeros(8); span = []; pre = 0.0; split_acc = {}
    reps = 0
    if mode == "rollout":
        from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor
        actor = StackedActor(2 * n, 3 * n + 2, 4, device="cuda")
        with torch.no_grad():
            actor.w3.mul_(100.0)
        opp = None
        if len(sys.argv) > 4 and sys.argv[4] == "scripted":            # the evaluation workload's shape: red = actors, blue = the scripted team in-kernel
            from deep_rl_battlespace_amd import instinct
            opp = instinct.Team(env.possible_blue, env.possible_red, env)
        ro = PolicyRollout(env, actor, 16, noise_std=0.1, one_launch=True, opponent=opp); ro.start()
    fence = 0.0
    for t in range(60, 64):
        if mode == "rollout":
            ro.run()                      # stamps of the LAST of the 16 ticks survive
        elif mode == "many":
            env.step_many(acts[:16].contiguous(), store=True)
        else:
            env.step_batch(acts[t])
        torch.cuda.synchronize()
        s10 = buf.cpu().numpy().reshape(waves, 10).astype(np.float64)
        if split:                                  # even rows: the first wave of each workgroup (everything but the geometry), odd rows: the second (geometry) -- printed one after the other
            roles = {"first": s10[0::2], "second": s10[1::2]}
            for rn, r in roles.items():
                seg = np.diff(np.concatenate([r[:, 8:9], r[:, :8]], axis=1), axis=1)
                acc = split_acc.setdefault(rn, np.zeros(8)); acc += seg.mean(0)
            split_acc["n"] = split_acc.get("n", 0) + 1
            if os.environ.get("BSX_STAMPS_PLACEMENT") and "placement" not in split_acc:
                # slot 9 = HW_ID (gfx9: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 [gfx950: se 15:13 + xcc elsewhere]); print what the bits say
                hw = s10[:, 9].astype(np.int64) if False else buf.cpu().numpy().reshape(waves, 10)[:, 9]
                role = np.arange(waves) & 1
                simd = (hw >> 4) & 3
                cu_key = ((hw >> 8) & 0xFF) | (((hw >> 32) & 0xF) << 8)  # cu_id 11:8, sh_id 12, se_id 15:13 of HW_ID + XCC_ID 3:0: one key per CU
                import collections
                per = collections.Counter()
                for k, sd, r in zip(cu_key.tolist(), simd.tolist(), role.tolist()):
                    per[(k, sd, r)] += 1
                hist = collections.Counter()
                for (k, sd) in {(k, sd) for (k, sd, r) in per}:
                    hist[(per.get((k, sd, 0), 0), per.get((k, sd, 1), 0))] += 1
                wg_pair = collections.Counter(((simd[0::2] - simd[1::2]) & 3).tolist())
                # does a first-role wave live longer on a SIMD that holds more of them?  lifetime = entry -> last stamp, by (first-role, all) waves on its SIMD
                raw = buf.cpu().numpy().reshape(waves, 10).astype(np.float64)
                life_w = raw[:, 7] - raw[:, 8]
                by = collections.defaultdict(list)
                for w in range(0, waves, 2):
                    k, sd = int(cu_key[w]), int(simd[w])
                    by[(per.get((k, sd, 0), 0), per.get((k, sd, 0), 0) + per.get((k, sd, 1), 0))].append(life_w[w])
                split_acc["life_by_load"] = sorted((key, len(v), float(np.mean(v)) * 10, float(np.percentile(v, 95)) * 10) for key, v in by.items())
                split_acc["placement"] = (sorted(hist.items()), sorted(wg_pair.items()), len({k for (k, sd, r) in per}))
            s10 = s10[0::2]
        s = s10[:, :8]
        if FINE:
            s = s[:, [0, 1, 3, 4, 5, 6, 2, 7]]
        pre += (s10[:, 0] - s10[:, 8]).mean(); fence += (s10[:, 9] - s10[:, 7]).mean()
        d = np.diff(s, axis=1)
        tot[:7] += d.mean(0); reps += 1
        span.append(((s[:, 7].max() - s[:, 0].min()), (s[:, 7] - s[:, 0]).mean(), (s[:, 0].max() - s[:, 0].min())))
        lf = s10[:, 7] - s10[:, 8]                        # entry -> last stamp, per wave (s_memtime bases differ between XCDs: no cross-wave differences)
        life = np.sort(lf)
        lifes = life if reps == 1 else np.concatenate([lifes, life])
        order = np.argsort(lf)
        seg = np.diff(np.concatenate([s10[:, 8:9], s], axis=1), axis=1)      # entry -> T0 issue, then the seven phases
        fast, slow = seg[order[:len(order) // 2]].mean(0), seg[order[-len(order) // 10:]].mean(0)
        fs = np.stack([fast, slow]) if reps == 1 else fs + np.stack([fast, slow])
    names = ["T0 loads -> plane record, heading-table request", "classify, shot (ballot, philox, table step / sincos), staging", "move (or re-spawn), pose hand-off, staging", "obs geometry", "bullet rounds (packed pass, part 2)", "resolve + rewards / game end", "stores"]
    if FINE:
        names = ["T0 loads -> plane record, heading-table request", "group ballot, call mode", "shot ballot, owner flags staged", "LDS init, wave barrier",
                 "shot (Philox, sincos, ring store), first slot fetch", "move (or re-spawn) -> hand-off [phase stamps 3..6 are off]", "everything else up to the stores' end"]
    tot /= reps
    print(f"E={E} n={n} waves={waves}; s_memtime ticks, mean over waves (divide the printed value by 10 for shader cycles)")
    print(f"  {'kernel entry -> first kernarg use (p.E)':40s} {pre / reps * 10:9.1f} ns")
    for nme, v in zip(names, tot[:7]):
        print(f"  {nme:40s} {v*10:9.1f} ns")
    if mode != "step":
        print(f"  {'end-of-tick fence (stores acknowledged)':40s} {fence / reps * 10:9.1f} ns   [mode {mode}: last tick of 16; the first phase includes the actor in rollout mode]")
    q = np.percentile(lifes, [5, 25, 50, 75, 95, 99, 100]) * 10
    print("  wave lifetime entry -> end, percentiles 5 / 25 / 50 / 75 / 95 / 99 / max: " + " / ".join(f"{v:.0f}" for v in q) + " ns")
    if not FINE:
        print("  per phase, ticks x 10: the faster half of the waves / the slowest tenth")
        for nme, a_, b_ in zip(["entry -> T0 issue"] + names, fs[0] / reps * 10, fs[1] / reps * 10):
            print(f"    {nme:40s} {a_:9.0f} {b_:9.0f}")
    if split:
        print("  wave-specialised kernel, shader cycles x 10 per segment (entry->T0 | T0->STAMP1 | ->move | ->geometry | ->bullets | ->resolve | ->stores | ->end):")
        for rn in ("first", "second"):
            v = split_acc[rn] / split_acc["n"] * 10
            print(f"    {rn:8s} " + " ".join(f"{x:8.0f}" for x in v) + f"   total {v.sum():8.0f}")
    if split and "placement" in split_acc:
        hist, pair, ncu = split_acc["placement"]
        print(f"  placement of the two roles (HW_ID): {ncu} distinct CU keys; SIMDs by (first-role waves, second-role waves) resident in one launch: {hist}")
        print(f"  (SIMD of a workgroup's first wave - SIMD of its second wave) mod 4: {pair}")
        print("  first-role wave lifetime (ns label = shader cycles x 10) by (first-role waves, all waves) on its SIMD: count, mean, p95")
        for key, cnt, mean, p95 in split_acc["life_by_load"]:
            print(f"    {key}: {cnt:5d} {mean:9.0f} {p95:9.0f}")
    sp = np.asarray(span).mean(0)
    print(f"  wave lifetime mean {sp[1]*10:.0f} ns; first-start to last-end {sp[0]*10:.0f} ns; start skew {sp[2]*10:.0f} ns")
