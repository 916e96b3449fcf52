// bsx_step_phase_shot.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): heading-table gather, call classification (inert / auto-reset / tie / physics), the shot (Bullet.__init__, sprites.py:293-318) and part 1 of
// the wave-packed bullet pass (shots queued in LDS behind the pool).  Reads: the T0 records (x, y, hp, dir, er, games, act / a0..a2, pc).  Writes: mode,
// tick, spawn, phys, ks, dl, dir_rot, slots, pool_pass, shot_exact, nd / nbdir / ncode / nexact, s_agg, s_eb, s_fl, s_new, bdir ring.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads -
// @writes  a2 rin_next
// @exports CHEAP_SHOT alive0 cnt_delta d0 dir_rot dl ks mode nbdir ncode nd nexact pool_pass rw shot_exact slots spawn srank sx0 sy0 tick
// @lds     s_agg s_eb s_fl s_new
    // ================= T1: the heading-table entry -- the one dependent load of the common path =====================
    // The heading table (361 x 16 B, read by every wave of every launch) stays hot in each CU's L1: the entry for the
    // post-rotation heading is gathered as soon as the action is known, so the plane can move while the shot is prepared.
    double dir_rot = dir;
    if (!CONT) dir_rot = rotate_dir(dir, act == 2 ? 15.0 : (act == 3 ? -15.0 : 0.0));   // one straight-line rotate (+0 leaves any heading in [0, 360] as it is)
    double2 dl = make_double2(0.0, 0.0);
    if (!CONT && !(DIAG & 32u)) dl = p.st.lut[min(max(int(dir_rot), 0), 360)];   // 21.5*cos(-radians(d)), 21.5*sin(-radians(d)) from host libm
    if (!CONT && (DIAG & 32u)) dl = make_double2(dir_rot * 0.05, 21.5 - dir_rot * 0.05);   // (timing-only variant: no dependent gather; wrong moves)
    if (MULTI && !ACTOR && tk + 1 < p.T) load_inputs(tk + 1, rin_next);   // behind this tick's own loads: nothing waits for it before the tick ends
    const bool alive0 = valid && hp > 0;
    STAMP(1);

    // "no agents left" (battle_env.py:309): group ballot over the alive flags
    const unsigned long long bal = ballot64(alive0);
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (lane & ~(G - 1)));
    const bool any_alive = (bal & gmask) != 0ull;

    // ---- what kind of call is this for my env (battle_env.py:303-323)
    int mode;
    int tick = er.tick;
    if (er.done) mode = (p.flags & BSX_F_AUTO_RESET) ? M_RESET : M_INERT;
    else if ((p.flags & BSX_F_EMPTY_CALL) || !any_alive) mode = M_TIE;
    else {
        tick += 1;
        mode = (tick >= tie_tick) ? M_TIE : M_PHYS;
    }
    if (!env_ok) mode = M_INERT;

    int4 cnt_delta = make_int4(0, 0, 0, 0);              // games, ties, red wins, blue wins
    const double d0 = dir;
    const int64_t genv = env_offset_t + ec;
    // does this call fire? (battle_env.py:404-406 / :423; the shot leaves from the PRE-move pose, so it is prepared first:
    // its Philox draw and sincos run while the heading-table entry of the move below is still on its way from the L2)
    if (CONT) a2 = fmin(fmax(a2, -1.0), 1.0);
    bool spawn = (mode == M_PHYS) && alive0 && !(DIAG & 2u) && (CONT ? (a2 > 0.0) : (act == 1));
    // ---- Bullet.__init__ (sprites.py:293-318) for this call's shot: heading = pre-move heading + (u*8 - 4)
    const bool phys = (mode == M_PHYS) && valid && !(DIAG & 2u);
    const int ks = tick % K;                             // birth-tick ring slot of this call's shot (heading, export only)
    // ---- wave-packed bullet pass, part 1.  The wave's bullets ARE a packed array -- its pool, pc entries in memory -- and this call's
    // shots queue up behind them in LDS by shot rank: slot w < pc is pool entry w, slot pc + r the r-th shooter's new bullet; slot w is
    // served by lane w % 64 in round w / 64.  Under uniform play a plane holds 0.6 bullets and fires every fourth call: ~37 + 16 slots,
    // ONE round.  What a slot needs from its bullet's owner (named by the entry) is staged per owner lane in LDS.
    FSTAMP(3);
    const unsigned long long shb = ballot64(spawn);
    const int srank = int(__builtin_amdgcn_mbcnt_hi(uint32_t(shb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(shb), 0u)));
    const int slots = int(pc) + __popcll(shb);           // wave-uniform
    if constexpr (R_BULLETS) {                           // (split kernels: only the wave that runs the bullets stages for the work slots)
        s_agg[tid] = 0u;
        s_eb[tid] = make_rect(pack_xy(team == 0 ? er.bbx : er.brx, team == 0 ? er.bby : er.bry), true, 33, 32, 33, 31);
        s_fl[tid] = uint32_t(ks) | (phys ? OWN_PHYS : 0u) | ((mode == M_RESET && valid) ? OWN_DROP : 0u);
    }
    // does this call touch the pool at all?  (not if no game of the wave is in its physics call or being re-spawned: entries stay as they are)
    const bool pool_pass = any64(phys || (mode == M_RESET && valid));
    FSTAMP(4);
    FSTAMP(5);
    // Discrete actions, compile-time team sizes: the shot's step from the heading table by angle addition instead of a float64 sincos
    // (below).  Round 3 had it at 1v1 only -- for larger teams the shorter shot measured SLOWER then (the table entry arrived later than
    // the sincos took).  With layout v2 every load of the step leaves in the first batch and the table shot pays at 2v2 and 3v3 (-3.6 % /
    // -4.1 %; with the corner-form tests -5 %); at 4v4 it pays only together with ordinary observation-row stores (bsx_step_phase_stores.inl:
    // 20.5 us, stable; with non-temporal rows a 4v4 kernel that reaches its stores sooner runs SLOWER: 24.5).
    constexpr bool CHEAP_SHOT = !CONT && N >= 1 && (N <= 3 || (N == 4 && (!MULTI || ACTOR)));   // (not the multi-tick 4v4 kernel: 15.8 -> 17.3 us per tick with it)
    double2 nd = make_double2(0.0, 0.0);                 // this call's shot: float64 step (CHEAP_SHOT: to ~1e-8 unless flagged exact), step code, heading
    double nbdir = 0.0;
    uint32_t ncode = 0u;
    bool nexact = false;
    // ONE Philox block per lane and call serves whichever of the two a lane needs -- they exclude each other: the jitter of this call's shot
    // (stream JITTER, counter (episode, tick | plane)), or the pose of a plane whose game this call re-spawns (stream AUTORESET, counter
    // (episode, plane); the first plane of each team also draws its base)
    const bool respawn = mode == M_RESET;
    uint4 rw = make_uint4(0u, 0u, 0u, 0u);
    // (split kernels: the wave that runs the bullets draws the jitter, a wave that moves planes the re-spawn -- the same block, whoever computes it.
    //  R_DRAW_LDS == 2, the per-call two-wave kernel's first wave: NEITHER -- its geometry wave, idle until the planes have moved, computes the
    //  block from the game's record and leaves it in LDS (bsx_step_split_geom_body.inl); this wave takes it after the pose hand-over and
    //  makes the shot there: bsx_step_phase_move.inl)
    if (R_DRAW_LDS != 2 && ((R_BULLETS && spawn && !u_t && !(DIAG & 8u)) || (R_MOVE && respawn)))
        rw = draw4(seed_t, genv, respawn ? STREAM_AUTORESET : STREAM_JITTER, games, respawn ? uint32_t(a < A ? a : A - 1) : ((uint32_t(tick) << 8) | uint32_t(a)));
    const int sx0 = x, sy0 = y;                          // the PRE-move position: the shot's origin wherever the block below runs
    if constexpr (R_DRAW_LDS != 2) {
#include "bsx_step_phase_shot_entry.inl"
    }
    // rare (step_code): a shot that moves by the float64 sum.  Asked once per wave, here, long before anything branches on it
    bool shot_exact = any64(spawn && nexact);
    FSTAMP(6);

