// bsx_step_phase_stores.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): write-back: plane and game records, reward, done flag, the observation row (straight from registers, or through LDS in the fused rollout and
// the runtime-n kernel), counters by atomics at a game's end, the pool count of a multi-tick launch.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   alive cnt_delta mode nhp_ nplane rew
// @writes  din_next ob_a ob_d oe_a oe_d
// @exports -
// @lds     s_pub
    PSTAMP(6);
    if (MULTI && !ACTOR && tk + 1 < p.T) din_next = decode(rin_next);   // the prefetch has long arrived; no store of this tick is out yet
    // ---- write back (MULTI: plane and game records travel in registers; memory gets them once, after the last tick)
    const bool last_tick = !MULTI || tk == p.T - 1;
    // two-wave per-call kernel: the two waves meet; the storing wave takes the observation values the geometry wave left in LDS
    if constexpr (R_GEOM_LDS != 0) {
        split_rendezvous();
        if constexpr (R_GEOM_LDS == 2) { const v4f_t gm = s_gm[tid]; ob_d = gm.x; ob_a = gm.y; oe_d[0] = gm.z; oe_a[0] = gm.w; }
    }
    // two-wave multi-tick kernel, form 2: the game wave publishes what the outputs wave needs of this tick (one 16-byte word per agent, a buffer
    // per tick parity): the post-move position, heading, flags, the enemy base, the reward -- and the two waves meet
    if constexpr (R_PUB == 1) {
        const bool on_pub = alive && (mode == M_PHYS ? nhp_ - nplane : nhp_) > 0;
        s_pub[(tk & 1) * SPB + tid] = v4u_t{pack_xy(x, y),
                                            uint32_t(int(dir)) | (alive ? 512u : 0u) | (on_pub ? 1024u : 0u) | (er.done ? 2048u : 0u),
                                            __float_as_uint(float(rew)), pack_xy(team == 0 ? er.bbx : er.brx, team == 0 ? er.bby : er.bry)};
        split_rendezvous();
    }
    if constexpr (R_ST_STATE || R_ST_OUT) {              // (split kernels: which wave stores what of the step's results)
    if (valid) {
        if constexpr (R_ST_STATE)
        if (MULTI ? last_tick : (mode == M_PHYS || mode == M_RESET)) {
            const uint2 pw = pack_plane(x, y, hp, dir, CONT);
            st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.plane, gt)), u32x2{pw.x, pw.y});
            if constexpr (CONT) st_store<NT_STATE>(elem(p.st.pdirf, gt), dir);   // continuous headings are fractional: the float64 beside the record
        }
        if constexpr (R_ST_OUT) {
        out_store(elem(rew_t, gt), float(rew));
        out_store(elem(done_t, gt), er.done ? uint8_t(1) : uint8_t(alive ? 0 : 1));
        }
    }
    // observation row: a dead observer sees all -1, a dead enemy is [-1,-1,-1] (battle_env.py:215-218,235-242).
    // Compile-time team sizes outside the fused rollout: the row leaves straight from registers, 16 bytes at a time plus a tail
    // (rows are 4 (3n + 2) bytes apart, so the stores are only dword-aligned -- fine for global_store_dwordx4).  Round 1 staged
    // rows in LDS to emit fully coalesced 16-byte stores; with non-temporal stores that transpose only costs: C2 8.21 -> 7.92 us,
    // 4v4 28.0 -> 24.9.  The fused rollout keeps its rows in LDS (the actor reads them there), the runtime-n kernel stages them too.
    constexpr bool DIRECT_OBS = !ACTOR && N > 0;
    if constexpr (!R_ST_OUT) {
    } else if constexpr (DIRECT_OBS) {
        constexpr int D = 3 * N + 2;
        float row[D];
        row[0] = alive ? ob_d : -1.0f;
        row[1] = alive ? ob_a : -1.0f;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const bool on = alive && ((N == 1) ? (mode == M_PHYS ? nhp_ - nplane : nhp_) : s_hp[eb + j]) > 0;
            row[2 + 3 * j] = on ? 1.0f : -1.0f;
            row[3 + 3 * j] = on ? oe_d[j] : -1.0f;
            row[4 + 3 * j] = on ? oe_a[j] : -1.0f;
        }
        if (valid) {
            float* out = elem(obs_t, gt * ix_t(D));
            // (rows leave non-temporal, except in the 4v4 discrete kernels: there ordinary stores -- the L2 merges a row's pieces.  With the
            //  hint the per-step 4v4 kernel has two modes by process, ~20.3 and ~22.4 us, and any cut of its instruction count makes it SLOWER
            //  (the table shot: 24.5); with ordinary stores + the table shot it runs 20.5, stable; the multi-tick form 15.7 against 19 ... 23.
            //  Every other kernel is faster or equal with the hint; profiles/r04_experiments.json)
            constexpr bool ROW_PLAIN = N == 4 && !CONT;
            auto row_store = [](auto* q, auto v) { if constexpr (ROW_PLAIN) *q = v; else out_store(q, v); };
#pragma unroll
            for (int i = 0; i + 4 <= D; i += 4) row_store(reinterpret_cast<v4f_t*>(out + i), v4f_t{row[i], row[i + 1], row[i + 2], row[i + 3]});
            typedef float v2f_t __attribute__((ext_vector_type(2)));
            if constexpr ((D & 3) >= 2) row_store(reinterpret_cast<v2f_t*>(out + (D & ~3)), v2f_t{row[D & ~3], row[(D & ~3) + 1]});
            if constexpr ((D & 1) != 0) row_store(out + D - 1, row[D - 1]);
        }
    } else
    {
        // Rows are staged in LDS ([lane][D], D odd -> conflict-free) and leave as coalesced 16-byte stores: the wave's rows
        // are one contiguous block of global memory when every lane is an agent (G == A).
        const int D = 3 * n + 2;
        float* srow = &s_obs[tid * D];
        srow[0] = alive ? ob_d : -1.0f;
        srow[1] = alive ? ob_a : -1.0f;
        if (N > 0) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const bool on = alive && ((N == 1) ? (mode == M_PHYS ? nhp_ - nplane : nhp_) : s_hp[eb + j]) > 0;
                srow[2 + 3 * j] = on ? 1.0f : -1.0f;
                srow[3 + 3 * j] = on ? oe_d[j] : -1.0f;
                srow[4 + 3 * j] = on ? oe_a[j] : -1.0f;
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const bool on = alive && s_hp[eb + j] > 0;
                float od = -1.0f, oa = -1.0f;
                if (on && !(DIAG & 1u)) obs_pair(x, y, dir, s_x[eb + j], s_y[eb + j], od, oa);
                srow[2 + 3 * j] = on ? 1.0f : -1.0f; srow[3 + 3 * j] = od; srow[4 + 3 * j] = oa;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (G == A && (reinterpret_cast<uintptr_t>(obs_t) & 15u) == 0) {
            // rows of this wave: global floats [base, base + rows*D); the wave's offset SPB*D*4 bytes is a multiple of 16
            const int64_t e_first = wblk * EPB;
            const int64_t rows = min(int64_t(SPB), (E_ - e_first) * A);
            const int64_t nfl = rows * D;                                   // floats to write
            float* gbase = obs_t + size_t(e_first) * A * D;
            for (int i = tid * 4; i < nfl; i += SPB * 4) {
                if (i + 4 <= nfl) {
                    out_store(reinterpret_cast<v4f_t*>(gbase + i), *reinterpret_cast<const v4f_t*>(&s_obs[i]));   // ds_read_b128
                } else {
                    for (int t = i; t < nfl; ++t) gbase[t] = s_obs[t];
                }
            }
        } else if (valid) {
            float* out = obs_t + gt * size_t(D);
            for (int i = 0; i < D; ++i) out[i] = srow[i];
        }
    }
    if constexpr (R_ST_STATE)
    if (valid) {
        if (a == 0) {
            if (MULTI ? last_tick : (mode != M_INERT)) {
                // the game record: hit points, clock, flags and the episode number; the base positions only when the game was re-spawned
                st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.envd, ix_t(e))), u32x2{pack_envd(er), games + uint32_t(cnt_delta.x)});
                if (MULTI || mode == M_RESET) {
                    const uint2 cw = pack_envc(er);
                    st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.envc, ix_t(e))), u32x2{cw.x, cw.y});
                }
            }
            if (cnt_delta.x) {                           // game over: the win / tie counters (nothing on the step path reads them: fire-and-forget atomics)
                int* const c4 = elem(p.st.cnt, ix_t(e) * 4);
                __hip_atomic_fetch_add(c4, cnt_delta.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.y) __hip_atomic_fetch_add(c4 + 1, cnt_delta.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.z) __hip_atomic_fetch_add(c4 + 2, cnt_delta.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.w) __hip_atomic_fetch_add(c4 + 3, cnt_delta.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last_tick) {
                if (p.env_done) *elem(p.env_done, ix_t(e)) = uint8_t(er.done);
                if (p.winner) *elem(p.winner, ix_t(e)) = uint8_t(er.winner);
            }
            if (MULTI && p.env_done_t) p.env_done_t[int64_t(tk) * E_ + e] = uint8_t(er.done);
        }
    }
    }
    if (R_BULLETS && MULTI && last_tick && lane == 0) *elem(p.st.bcnt, ix_t(wblk)) = pc;   // the pool's length travelled in a register
