// bsx_step_phase_geometry.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): observation geometry (battle_env.py:202-244): range and angle-off to the enemy base and every enemy plane in binary64 (bsx_geometry.h), each
// red-blue pair once for n >= 2; also stores a flagged shot's float64 step (exact path).  Reads: post-move poses.  Writes: ob_d, ob_a, oe_d[], oe_a[], ex[], ey[].
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   CHEAP_SHOT ks nbdir nexact nx_ ny_ shot_exact spawn
// @writes  nd
// @exports ob_a ob_d oe_a oe_d
// @lds     s_gm s_nd s_pd s_pr
    PSTAMP(3);
    // ---- observation geometry (battle_env.py:202-244) from the staged block, BEFORE the bullets: poses are final after
    //      the move, only the alive flags can still change; this fp64 math runs while the bullet-step loads are in flight.
    if (R_BULLETS && shot_exact) {                       // wave-uniform and rare: a flagged shot leaves its float64 step in the ring (and in LDS for its first update)
        if constexpr (N == 1) asm volatile("");          // (keeps this a scalar branch; see the bullet rounds)
        if (spawn && nexact) {
            if constexpr (CHEAP_SHOT) {                  // the exact float64 step, as Bullet.update evaluates it (sprites.py:35-42,330-333)
                double sn, cs;
                sincos(-(nbdir * DEG2RAD), &sn, &cs);
                nd = make_double2(BULLET_STEP * cs, BULLET_STEP * sn);
            }
            *elem(p.st.bd, ix_t(ks) * EAt + gt) = nd;
            s_nd[2 * tid] = nd.x; s_nd[2 * tid + 1] = nd.y;
        }
    }
    const int obx = team == 0 ? er.bbx : er.brx, oby = team == 0 ? er.bby : er.bry;   // enemy base
    float ob_d = -1.0f, ob_a = -1.0f;
    float oe_d[NE], oe_a[NE];
    int ex[NE], ey[NE];
    if constexpr (!R_GEOM) {                                         // (split kernels: a wave without the observation geometry)
    } else if constexpr (N == 0) {                                   // runtime-n build: the enemy planes' pairs are worked out at row assembly
        if (!(DIAG & 1u)) obs_pair(x, y, dir, obx, oby, ob_d, ob_a);
    } else if constexpr (N == 1) {
        ex[0] = nx_; ey[0] = ny_;
        oe_d[0] = -1.0f; oe_a[0] = -1.0f;
        if (!(DIAG & 1u)) {                                          // enemy base and enemy plane, the two evaluations in lockstep
            const int tx[2] = {obx, nx_}, ty[2] = {oby, ny_};
            float d[2]; double rd[2];
            geometry_n<2>(x, y, tx, ty, d, rd);
            ob_d = d[0]; ob_a = float(rel_from_rads(rd[0], dir) * (1.0 / 360.0));
            oe_d[0] = d[1]; oe_a[0] = float(rel_from_rads(rd[1], dir) * (1.0 / 360.0));
        }
    } else if constexpr (N >= 2) {
        // The range of a pair is symmetric and its bearing differs by pi between the two ends, so each red-blue pair is
        // worked out once -- by red plane i for blue j when i + j is even, by blue j otherwise -- in (N + 1) / 2 rounds of
        // one sqrt + atan2 per lane instead of N, and the other end derives its bearing: rads +- pi (coincident planes: 0,
        // as atan2(+0, +0) gives both ends).  The derived value can differ from a direct atan2 in its last bits (<= ~4 ulp
        // of float64), which survives the single rounding to float32 with probability ~1e-8, like the libm difference.
        constexpr double PI_D = 3.14159265358979323846;
        const int mi = min(team == 0 ? a : a - N, N - 1);            // my index inside my team (lanes beyond A: clamped, never write)
#pragma unroll
        for (int j = 0; j < NE; ++j) { ex[j] = s_x[eb + j]; ey[j] = s_y[eb + j]; oe_d[j] = -1.0f; oe_a[j] = -1.0f; }
        if (!(DIAG & 1u)) {
            // the enemy base and the (N + 1) / 2 pairs this lane owns: all evaluations in lockstep (geometry_n)
            constexpr int R = (N + 1) / 2;
            int tx[R + 1], ty[R + 1], ojc[R];
            float d[R + 1]; double rd[R + 1];
            tx[0] = obx; ty[0] = oby;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oj = (team == 0 ? (mi & 1) : ((mi + 1) & 1)) + 2 * r;   // the enemy I own in this round
                ojc[r] = min(oj, N - 1);
                tx[r + 1] = s_x[eb + ojc[r]]; ty[r + 1] = s_y[eb + ojc[r]];
            }
            geometry_n<R + 1>(x, y, tx, ty, d, rd);
            ob_d = d[0]; ob_a = float(rel_from_rads(rd[0], dir) * (1.0 / 360.0));
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oj = (team == 0 ? (mi & 1) : ((mi + 1) & 1)) + 2 * r;
                const bool own = oj < N && a < A;
                const int slot = (gl + (team == 0 ? mi : ojc[r])) * N + (team == 0 ? ojc[r] : mi);   // [red plane of my game][blue index]
                if (own) { s_pd[slot] = d[r + 1]; s_pr[slot] = rd[r + 1]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int ri = team == 0 ? mi : j, bi = team == 0 ? j : mi;
                const int slot = (gl + ri) * N + bi;
                const bool mine = (team == 0) == (((ri + bi) & 1) == 0);
                const double r0 = s_pr[slot];
                const bool same = ex[j] == x && ey[j] == y;
                const double rads = mine ? r0 : (same ? 0.0 : (r0 < PI_D ? r0 + PI_D : r0 - PI_D));
                oe_d[j] = s_pd[slot];
                oe_a[j] = float(rel_from_rads(rads, dir) * (1.0 / 360.0));
            }
        }
    }
    // two-wave per-call kernel: the geometry wave hands its four observation values to the wave that stores the rows, and is done
    if constexpr (R_GEOM_LDS == 1) s_gm[tid] = v4f_t{ob_d, ob_a, oe_d[0], oe_a[0]};
