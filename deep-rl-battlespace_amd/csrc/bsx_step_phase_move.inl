// bsx_step_phase_move.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): in-kernel re-spawn (M_RESET) or process_action's move (battle_env.py:383-424), then the hand-off of post-move poses to the game's other planes
// (1v1: DPP; larger teams: wave-private LDS) and the planes' sprites as rectangles for the work slots.  Writes: x, y, dir, hp, er (re-spawn),
// nx_, ny_, nhp_ (1v1), s_x, s_y, s_hp, s_bhit, s_pq.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   alive0 dir_rot dl mode rw
// @writes  a0 a1 dir er hp tick x y
// @exports nhp_ nx_ ny_
// @lds     s_bhit s_hp s_pq s_t0 s_t1 s_x s_y
    STAMP(2);
    if constexpr (R_MOVE) {                              // (split kernels: a wave that only runs bullets does not move planes)
    if (mode == M_RESET) {
        // re-spawn in place of the inert call; episode id = games played so far.  My block holds my pose and (first plane of a team) my
        // team's base; the two bases travel to every lane of the game (all lanes of a game are here together)
        const SpawnDraw sd = spawn_from_words(rw, a < A ? a : A - 1, n);
        if constexpr (N == 1) {
            const int ox = lane_xor1(sd.bx), oy = lane_xor1(sd.by);
            er.brx = team == 0 ? sd.bx : ox; er.bry = team == 0 ? sd.by : oy;
            er.bbx = team == 0 ? ox : sd.bx; er.bby = team == 0 ? oy : sd.by;
        } else {
            er.brx = __shfl(sd.bx, gl); er.bry = __shfl(sd.by, gl);
            er.bbx = __shfl(sd.bx, gl + n); er.bby = __shfl(sd.by, gl + n);
        }
        er.bhp_r = er.bhp_b = 5 * n;
        er.tick = 0; er.done = 0; er.winner = BSX_WINNER_NONE;
        tick = 0;
        x = sd.x; y = sd.y; dir = double(sd.dir);
        hp = PLANE_HP;
    } else if (mode == M_PHYS && alive0) {
        // ---- process_action (battle_env.py:383-424)
        if (!CONT) {
            dir = dir_rot;
            if (act >= 0 && act <= 3) {
                x = int(double(x) + dl.x);               // Rect.center store truncates toward zero
                y = int(double(y) + dl.y);
                clamp_plane(x, y);
            }
        } else {
            a0 = fmin(fmax(a0, -1.0), 1.0); a1 = fmin(fmax(a1, -1.0), 1.0);
            const double speed = ((a0 + 1.0) / 2.0) * 75.0 + 200.0;    // battle_env.py:419
            double sn, cs;
            sincos(-(dir * DEG2RAD), &sn, &cs);
            const double st = speed * TIME_STEP;
            x = int(double(x) + (st * cs));
            y = int(double(y) + (st * sn));
            clamp_plane(x, y);
            dir = rotate_dir(dir, a1 * 35.0);                          // :421-422
        }
    }
    }

    // ---- hand the post-move pose and hit points to the other planes of the game.  1v1: the only other plane is the lane
    //      next door, three cross-lane moves (DPP) instead of LDS round trips; larger teams stage the block in LDS.
    int nx_ = 0, ny_ = 0, nhp_ = 0;                      // 1v1: the enemy's x, y, hit points
    if constexpr (R_STAGE) s_pq[tid] = make_rect(pack_xy(x, y), valid && hp > 0, 27, 25, 27, 24);
    if constexpr (N == 1) {
        nx_ = lane_xor1(x); ny_ = lane_xor1(y); nhp_ = lane_xor1(valid ? hp : 0);
    } else {
        s_x[tid] = x; s_y[tid] = y; s_hp[tid] = valid ? hp : 0;
        s_bhit[tid] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // two-wave per-call kernel: this wave leaves what the observation geometry needs -- my pose, the enemy's position, the enemy base --
    // for the geometry wave, which has waited for it (32 bytes per lane; the geometry wave loads nothing and repeats nothing)
    if constexpr (R_POSE_LDS) {
        const double dq = dir;
        s_t0[tid] = v4u_t{uint32_t(x), uint32_t(y), uint32_t(nx_), uint32_t(ny_)};
        s_t1[tid] = v4u_t{uint32_t(__double2loint(dq)), uint32_t(__double2hiint(dq)), uint32_t(team == 0 ? er.bbx : er.brx), uint32_t(team == 0 ? er.bby : er.bry)};
        split_rendezvous();
    }
