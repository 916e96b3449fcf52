// bsx_step_phase_shot_entry.inl -- part of the shot PHASE (bsx_step_phase_shot.inl includes it in place; the per-call two-wave kernel's first
// wave includes it after the pose hand-over instead, bsx_step_phase_move.inl, once the geometry wave's Philox block is there): Bullet.__init__
// (sprites.py:293-318) for this call's shot from the block `rw` -- heading = pre-move heading + (u*8 - 4), the per-update step, the step
// code, the heading in the export ring, the shot as a pool entry queued in LDS by shot rank.  Reads rw, spawn, d0, dl, sx0, sy0, uu_in;
// writes nbdir, nd, ncode, nexact, s_new.
    if (R_BULLETS && spawn) {
        double uu = uu_in;
        if (!u_t && !(DIAG & 8u)) uu = uniform53(rw.x, rw.y);
        const double jit = uu * 8.0 - 4.0;
        nbdir = d0 + jit;
        if constexpr (CHEAP_SHOT) {
            // Discrete headings are whole degrees and a shooter does not turn, so (21.5 cos d0, -21.5 sin d0) is the heading-table
            // entry `dl` this lane gathered for its move; the jitter is at most 4 degrees.  The integer step code only needs the
            // step to ~2^-18 (step_code's guard is wider than any error here), so the common path takes it from the angle-addition
            // formulas with two-term series for the jitter -- |error| < 1e-8 on 45 cos -- instead of a float64 sincos of ~110
            // instructions.  A shot the code flags as not provably exact (one in ~30 000) gets the library sincos below, behind the
            // wave-uniform branch of the exact path; every other shot's integer moves are those of the exact step (same floor, the
            // fraction far from 0 and 1), so the results do not change.
            const double jr = jit * DEG2RAD, t = jr * jr;
            const double cj = __builtin_fma(t, __builtin_fma(t, 1.0 / 24.0, -0.5), 1.0);
            const double sj = jr * __builtin_fma(t, __builtin_fma(t, 1.0 / 120.0, -1.0 / 6.0), 1.0);
            constexpr double K45 = BULLET_STEP / 21.5;
            nd = make_double2(K45 * __builtin_fma(dl.x, cj, dl.y * sj), K45 * __builtin_fma(dl.y, cj, -(dl.x * sj)));
        } else {
            double sn, cs;
            sincos(-(nbdir * DEG2RAD), &sn, &cs);
            nd = make_double2(BULLET_STEP * cs, BULLET_STEP * sn);
        }
        ncode = step_code(nd.x, nd.y, nexact);
        st_store<NT_STATE>(elem(p.st.bdir, ix_t(ks) * EAt + gt), nbdir);      // ring by birth tick: never moves, read only by bsx_export_state
        // the shot as a pool entry, queued by shot rank: age 0, the PRE-move pose, my lane as its owner
        s_new[srank] = u32x2{pack_bullet(sx0, sy0, 0) | (nexact ? ENT_EXACT : 0u) | (uint32_t(lane) << ENT_OWNER_SHIFT), ncode};
    }
