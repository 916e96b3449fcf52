// bsx_step_split_geom_body.inl -- the GEOMETRY wave of bsx_step_split_kernel's per-call form (bsx_step_split.h includes it): no loads, no game
// logic -- it waits until the first wave has moved the planes, takes the poses from LDS, works out the observation geometry (the SAME phase
// file, bsx_step_phase_geometry.inl) and hands the four observation values per agent back before the stores.
    {
        uint4 rw = make_uint4(0u, 0u, 0u, 0u);
        bool respawn = false;
        if constexpr (DRAW) {
        // While the first wave waits for its loads, classifies and moves, this wave has nothing to do -- so it computes the call's ONE Philox
        // block per lane (bsx_step_phase_shot.inl: the jitter of this call's shot, or the pose of a plane whose game this call re-spawns; they
        // exclude each other) from the game's own record: its key is (seed, global game, stream, episode, tick | plane), all of it in the
        // record or the kernel's arguments.  The block is in LDS before this wave arrives at the pose rendezvous; the first wave takes it
        // there.  Which of the two a lane needs follows from the record alone: a finished game under auto-reset is re-spawned (M_RESET),
        // every other lane gets the jitter block of tick + 1 -- the first wave uses it only if the call turns out to be a physics call
        // in which the lane fires, and then tick + 1 is the call's tick.
        // (a launch of more than 65 536 games puts more than two workgroups on a SIMD: there this wave's block would queue behind every first
        //  wave's classify and move, and ITS first wave would wait for it at the rendezvous -- so it is computed at the first waves' priority
        //  then: 81 920 games 6.50 -> 6.14 us, 98 304 6.75 -> 6.40; up to 65 536 games the ports have room and the raise only costs: 5.42 -> 5.44.
        //  At 114 688 games -- seven waves on a SIMD -- the draw does not pay in this wave at either priority (7.13 -> 7.5 ... 8.1): the launcher
        //  takes the kernel without it, DRAW = false, above 98 304 games)
        const bool crowded = E_ > 65536;
        if (crowded) __builtin_amdgcn_s_setprio(1);
        const uint2 edw = *elem(envd_, ix_t(ec));
        respawn = env_ok && ((edw.x >> 27) & 1u) != 0u && (p.flags & BSX_F_AUTO_RESET) != 0u;
        rw = draw4(p.seed, p.env_offset + int64_t(ec), respawn ? STREAM_AUTORESET : STREAM_JITTER, edw.y,
                               respawn ? uint32_t(a) : (((((edw.x >> 18) & 511u) + 1u) << 8) | uint32_t(a)));
        s_rw[tid] = v4u_t{rw.x, rw.y, rw.z, rw.w};
        if (crowded) __builtin_amdgcn_s_setprio(0);
        }
        split_rendezvous();                              // the first wave has moved its planes (bsx_step_phase_move.inl, R_POSE_LDS)
        STAMP(0); STAMP(1); STAMP(2);                    // (diagnostic builds: this wave's row of stamps -- waited for the poses | geometry | waited for the stores)
        const v4u_t h0 = s_t0[tid];
        int x = sx16(h0.x), y = sy16(h0.x);
        double dir = __hiloint2double(int(h0.z), int(h0.y));
        int ebx_ = sx16(h0.w), eby_ = sy16(h0.w);
        if (DRAW && any64(respawn)) {                    // wave-uniform: a game of this wave is re-spawned by this call -- its planes' new poses and
            const SpawnDraw sd = spawn_from_words(rw, a, 1);     // bases from the block, exactly as the first wave applies them after the rendezvous
            const int obx_ = lane_xor1(sd.bx), oby_ = lane_xor1(sd.by);      // (the enemy's base is the draw of the lane next door)
            if (respawn) { x = sd.x; y = sd.y; dir = double(sd.dir); ebx_ = obx_; eby_ = oby_; }
        }
        const int nx_ = lane_xor1(x), ny_ = lane_xor1(y);        // the enemy's position (for a re-spawned game: its new one)
        EnvU er = {};
        er.bbx = er.brx = ebx_; er.bby = er.bry = eby_;             // (the phase reads the enemy base through my team: both are it)
        // what the phase file names besides: the exact-shot store is the first wave's (R_BULLETS), this wave takes no part in it
        constexpr bool R_BULLETS = false, R_GEOM = true, CHEAP_SHOT = false;
        constexpr int R_GEOM_LDS = 1;
        const int team = 0, lane = tid, ks = 0, gl = tid & ~(G - 1), eb = gl;
        const bool shot_exact = false, spawn = false, nexact = false;
        double nbdir = 0.0;
        double2 nd = make_double2(0.0, 0.0);
        const ix_t gt = g, EAt = 0;
        (void)lane; (void)ks; (void)shot_exact; (void)spawn; (void)nexact; (void)nbdir; (void)nd; (void)gt; (void)EAt; (void)team; (void)gl; (void)eb;
#include "bsx_step_phase_geometry.inl"
        STAMP(4); STAMP(5); STAMP(6);
        split_rendezvous();                              // the hand-over before the stores (bsx_step_phase_stores.inl, R_GEOM_LDS)
        STAMP(7);
    }
