// bsx_step_phase_bullets.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): Bullet.update (sprites.py:321-351) for every work slot of the wave (part 2 of the wave-packed pass: move, miss / base / plane-overlap tests,
// compaction of the pool) and the ordered plane-hit resolve (battle_env.py:332-360).  Reads: pool_first / the pool, s_new, s_eb, s_pq, s_fl, slots, pc.
// Writes: the pool and its count, nmiss, nbase, nplane (+ the enemy's: nplane_other, nbase_other), s_hp / s_bhit (n >= 2), tombstones.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   nhp_ pool_pass slots
// @writes  pc
// @exports nbase nbase_other nmiss nplane nplane_other
// @lds     s_agg s_bhit s_hp s_npl s_ov s_pp
    PSTAMP(4);
    // ---- Bullet.update (sprites.py:321-351) per work slot, predicates as integer sign masks (0 / -1).
    uint64_t ovl[OW];
#pragma unroll
    for (int q = 0; q < OW; ++q) ovl[q] = 0;
    int nmiss = 0, nbase = 0, nplane = 0;
    // the float64 move of the rare entries that carry the exact-path flag (exact_step() fetches the step the shot left in the ring)
    auto move_exact = [&](uint32_t ew, auto exact_step) {
        const double2 dd = exact_step();
        const int ebx = int(double(bullet_x(ew)) + dd.x);    // truncation toward zero
        const int eby = int(double(bullet_y(ew)) + dd.y);
        return (uint32_t(ebx) & 0xFFFFu) | (uint32_t(eby) << 16);
    };
    bool any_hit = false;
    if (R_BULLETS && pool_pass) {
        // ---- wave-packed bullet pass, part 2: Bullet.update per work slot.  A slot reads what its bullet's OWNER would have had
        // in registers -- the enemy base, the enemy planes' post-move poses and alive flags -- from the wave's LDS block, moves the
        // bullet, and hands the outcome back: one LDS add per bullet that ended (miss and base-hit counts), the survivor straight to
        // its place in the compacted pool (a wave-wide prefix count of the survivors: one ballot), the rare plane-overlap candidates
        // as the by-age bit fields the ordered resolve below walks.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the pool's first entries arrived long ago, behind the shot and the observation geometry
                                                         // (and no later wait is held up by the stores issued since: vmcnt is in-order)
        // a round's survivor stores are issued at the START of the next round (after the loop for the last one), i.e. BEFORE the
        // loads of the round after: loads and stores share the in-order vmcnt, so a wait for loads issued ahead of stores would
        // also sit out the stores' acknowledgement; issued behind them, both are long done when the entries are needed.  A round's
        // survivors land below that round's first slot, i.e. never on an entry that is still to be read.
        bool st_on = false; int st_ps = 0; uint2 st_w = make_uint2(0u, 0u);
        auto flush_stores = [&]() {
            if (st_on) st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.bent, pool0 + ix_t(st_ps))), u32x2{st_w.x, st_w.y});
        };
        const ix_t gb0 = ix_t(wblk * EPB) * ix_t(A);     // row of lane 0's agent; owner lane o sits (o / G) * A + (o % G) rows on
        // what a slot needs from its owner's side of the game, read from the wave's LDS block: enemy base, enemy planes (post-move
        // pose + alive flag), the owner's flags
        struct Ctx { rect_t ebw, pq[NE]; uint32_t fl; int ebl; };
        auto fetch_ctx = [&](int o) {
            Ctx c;
            c.ebw = s_eb[o];
            c.ebl = (o & ~(G - 1)) + (((o & (G - 1)) < n) ? n : 0);          // first lane of the owner's enemy team
            if constexpr (N > 0) {
#pragma unroll
                for (int q = 0; q < NE; ++q) c.pq[q] = s_pq[c.ebl + q];
            }
            c.fl = s_fl[o];
            return c;
        };
        int wpos = 0;                                    // survivors written so far = the pool's new length (wave-uniform)
        uint2 nxt = pool_first;                          // slot rd * 64 + lane of the round about to run, as loaded from the pool
        auto do_round = [&](const int rd) {
            const int w = rd * SPB + lane;
            const bool on = w < slots;
            uint2 en = nxt;
            if (w >= int(pc)) { const u32x2 sh = s_new[on ? w - int(pc) : 0]; en = make_uint2(sh.x, sh.y); }                 // one of this call's shots (LDS, by shot rank)
            const int o = on ? int(en.x >> ENT_OWNER_SHIFT) : lane;
            const Ctx c = fetch_ctx(o);
            if (rd > 0) flush_stores();
            // more than 64 slots in the wave: the next round's pool entries are fetched while this one is worked on
            if ((rd + 1) * SPB < int(pc)) nxt = *elem(p.st.bent, pool0 + ix_t((rd + 1) * SPB + lane));
            const uint32_t age0f = en.x & ENT_AGE;                              // updates so far, << 11
            const bool ophys = on && (c.fl & OWN_PHYS) != 0u;                   // the owner's game is in its physics call
            const int lvm = (ophys && age0f != (TOMBSTONE_AGE << 11)) ? -1 : 0;  // a tombstone (plane hit last call) is dropped
            // Everything from here to the outcome works on (x, y) PAIRS in the two 16-bit halves of a register: the move, and every
            // rectangle test as "some lower or upper margin is negative" = a sign bit in either half.
            uint32_t bpk = step_pk(en.x & ENT_XY, en.y);
            if (any64(lvm != 0 && (en.x & ENT_EXACT) != 0u)) {                  // wave-uniform and rare: the float64 move of flagged entries
                if constexpr (N == 1) asm volatile("");   // (1v1: keeps this a scalar branch on the common path)
                if (lvm != 0 && (en.x & ENT_EXACT) != 0u) {                     // (this call's shot left its step in LDS, older ones in the ring by birth tick)
                    const ix_t go = gb0 + ix_t(o / G) * ix_t(A) + ix_t(o & (G - 1));
                    bpk = move_exact(en.x, [&]() {
                        return age0f == 0u ? make_double2(s_nd[2 * o], s_nd[2 * o + 1])
                                           : *elem(p.st.bd, ix_t(ring_pos(int(c.fl & 15u), int(age0f >> 11))) * EAt + go);
                    });
                }
            }
            // miss: off the field (x > 1200 | x < 0 | y > 800 | y < 0), or dist_travelled >= 500 <=> this is the 12th update.  Plain
            // 32-bit arithmetic with literals on the packed pair: a half that borrows from (or carries into) its neighbour does so
            // only when a coordinate is negative or beyond the limit -- the bullet is a miss then, whatever the other half says, and
            // the base / plane results below are discarded for a miss.
            const uint32_t over = CORNERS ? pk_const(FIELD_W, FIELD_H) - bpk : pk_bits(as_pk(pk_const(FIELD_W, FIELD_H)) - as_pk(bpk));
            const int missm = pk_any_negative(bpk | over | (0x5000u - age0f));
            const s16x2 b2 = as_pk(bpk + (CORNERS ? pk_const(PK_BIAS, PK_BIAS) : 0u));
            // base: 6x3 bullet rect vs 62x62 base rect, strict overlap <=> dx in [-33, 33] and dy in [-32, 31]
            const int basem = hits_rect(b2, c.ebw, 33, 32, 33, 31) & ~missm;
            // planes: vs the un-rotated 50x48 rect at the post-move pose <=> dx in [-27, 27] and dy in [-25, 24]
            uint32_t m = 0;
            if constexpr (N > 0) {
#pragma unroll
                for (int q = 0; q < NE; ++q) m |= uint32_t(hits_rect(b2, c.pq[q], 27, 25, 27, 24)) & (1u << q);
            } else {
                for (int q = 0; q < n; ++q) m |= uint32_t(hits_rect(b2, s_pq[c.ebl + q], 27, 25, 27, 24)) & (1u << q);
            }
            const int age = int(age0f >> 11) + 1;
            const int gonem = (missm | basem) & lvm;
            const int keepm = lvm & ~gonem;
            m &= uint32_t(keepm);
            // One LDS add hands a bullet that ended to its owner: misses << 16 | base hits << 24.
            const uint32_t add = (uint32_t(missm & lvm & 1) << 16) | (uint32_t(basem & lvm & 1) << 24);
            if (add) __hip_atomic_fetch_add(&s_agg[o], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // What stays in the pool: a bullet that flies on (new position, age + 1), and -- untouched -- the entries of games that are
            // not in their physics call (finished and waiting, or tied by this call); the entries of a game this call re-spawns go.
            const bool asis = on && !ophys && (c.fl & OWN_DROP) == 0u;
            const bool stay = keepm != 0 || asis;
            const unsigned long long kb = ballot64(stay);
            const int ps = wpos + int(__builtin_amdgcn_mbcnt_hi(uint32_t(kb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(kb), 0u)));
            wpos += __popcll(kb);
            st_on = stay; st_ps = ps;
            st_w = make_uint2(asis ? en.x : (((en.x & ~ENT_XY) | bpk) + 0x800u), en.y);   // the new position, age + 1; flag and owner as they were
            if (any64(m != 0u)) {                        // wave-uniform and rare: a bullet overlaps a live enemy plane
                any_hit = true;
                if (m != 0u) {
                    if constexpr (OW == 1) __hip_atomic_fetch_or(&s_ov[o], (unsigned long long)(m) << (age * FW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_or(&s_ov[o * OW + (age >> 2)], (unsigned long long)(m) << ((age & 3) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    s_pp[o * K + age] = uint16_t(ps);    // (age 1 .. 11 here: a 12th update is always a range miss)
                }
            }
        };
        // the first round stands alone (under sparse play it is the only one in 85 % of the waves): straight-line code, no loop-carried
        // copies of the prefetch registers
        if (slots > 0) do_round(0);
        for (int rd = 1; rd * SPB < slots; ++rd) do_round(rd);
        if (slots > 0) flush_stores();
        if (wpos != int(pc) || MULTI) {                  // the pool's new length (one word per wave)
            if (lane == 0 && !MULTI) *elem(p.st.bcnt, ix_t(wblk)) = uint32_t(wpos);
            pc = uint32_t(wpos);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t agg = s_agg[tid];
        nmiss = int((agg >> 16) & 0xFFu); nbase = int((agg >> 24) & 0xFFu);
        if (any_hit) {
#pragma unroll
            for (int q = 0; q < OW; ++q) { ovl[q] = s_ov[tid * OW + q]; s_ov[tid * OW + q] = 0ull; }
        }
        if (N != 1 && nbase) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(&s_bhit[gl + team]), nbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PSTAMP(5);
    // ---- ordered plane-hit resolve (battle_env.py:332-360 with sprites.py:348-350): creation order = oldest age
    //      first, then shooter id; a plane killed earlier in the walk no longer stops later bullets.
    uint64_t any_ovl = 0;
#pragma unroll
    for (int q = 0; q < OW; ++q) any_ovl |= ovl[q];
    if (R_BULLETS && ballot64(any_ovl != 0ull) != 0ull && !(DIAG & 4u)) {   // wave-uniform: most waves have no candidate at all
        uint32_t consumed = 0;                                 // by age
        if constexpr (N == 1) {
            // one shooter per target: my candidates, oldest first, hit until the enemy's hit points run out; the rest fly on
            int left = nhp_;
            for (int ag = K - 1; ag >= 1; --ag) {
                const bool hit = ((ovl[0] >> (ag * FW)) & 1ull) != 0ull && left > 0;
                if (hit) { left -= 1; nplane += 1; consumed |= 1u << ag; }
            }
        } else
        for (int ag = K - 1; ag >= 1; --ag) {                  // oldest first; an age-12 bullet is always a range miss
            uint64_t wsel = ovl[0];
            if (OW == 3) wsel = ((ag >> 2) == 0) ? ovl[0] : (((ag >> 2) == 1) ? ovl[OW > 1 ? 1 : 0] : ovl[OW > 2 ? 2 : 0]);
            const uint32_t m = uint32_t(wsel >> (OW == 1 ? ag * FW : (ag & 3) * 16)) & ((1u << FW) - 1u);
            if (ballot64(m != 0) == 0ull) continue;            // wave-uniform: nobody has a candidate of this age
            for (int i = 0; i < n; ++i) {
                if (m != 0 && (a - (team ? n : 0)) == i) {
                    for (int j = 0; j < n; ++j) {
                        if (((m >> j) & 1u) && s_hp[eb + j] > 0) {
                            s_hp[eb + j] = s_hp[eb + j] - 1;                                 // Plane.hit
                            nplane += 1;
                            consumed |= 1u << ag;
                            break;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // a bullet that hit a plane is gone: its pool entry becomes a tombstone, dropped by the next call's compaction
        // (the survivor entry was stored by ANOTHER lane of this wave; let it land before its first word is overwritten)
        if (any64(consumed != 0u)) __builtin_amdgcn_s_waitcnt(0x0F70);
        while (consumed) {
            const int ag = __builtin_ctz(consumed);
            consumed &= consumed - 1u;
            elem(p.st.bent, pool0 + ix_t(s_pp[tid * K + ag]))->x = pack_bullet(0, 0, int(TOMBSTONE_AGE)) | (uint32_t(lane) << ENT_OWNER_SHIFT);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (R_RDV_COUNTS != 0) {
        // two-wave multi-tick kernel, up to 32 768 games: the game wave hands its bullets' counts to the outputs wave -- misses and base hits
        // sit in s_agg (the work slots' adds), the plane hits of the ordered resolve go beside them -- and the two waves meet
        // (one packed word per shooter in a buffer of the tick's parity: the game wave runs up to a tick ahead)
        if constexpr (R_RDV_COUNTS == 1) s_npl[(tk & 1) * SPB + tid] = uint32_t(nmiss) | (uint32_t(nbase) << 8) | (uint32_t(nplane) << 16);
        split_rendezvous();
        if constexpr (R_RDV_COUNTS == 2) {
            const uint32_t c3 = s_npl[(tk & 1) * SPB + tid];
            nmiss = int(c3 & 0xFFu); nbase = int((c3 >> 8) & 0xFFu); nplane = int((c3 >> 16) & 0xFFu);
        }
    }
    int nplane_other = 0, nbase_other = 0;               // 1v1: what the enemy's bullets did to me / to my base
    if constexpr (N == 1) { nplane_other = lane_xor1(nplane); nbase_other = lane_xor1(nbase); }

