// bsx_step_phase_outcome.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): rewards (battle_env.py:337-359), deaths, base hit points, win / tie (:363-372, :469-496).  Reads: nmiss, nbase, nplane, *_other, mode, tick.
// Writes: rew, hp, alive, er (hit points, tick, done, winner), cnt_delta.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   alive0 mode nbase nbase_other nmiss nplane nplane_other tick
// @writes  cnt_delta er hp
// @exports alive rew
// @lds -
    // ---- rewards (battle_env.py:337-359), deaths, bases, win / tie (:363-372, :469-496)
    double rew = double(nmiss) * p.cfg.miss_punishment + double(nbase) * p.cfg.hit_base_reward +
                 double(nplane) * p.cfg.hit_plane_reward;
    bool alive = valid && hp > 0;
    if constexpr (R_OUTCOME) {                           // (split kernels: a wave that neither stores the results nor carries the state on works nothing out)
    if (mode == M_PHYS) {
        const int hp_new = (N == 1) ? (valid ? hp : 0) - nplane_other : s_hp[tid];
        if (alive0 && hp_new <= 0) rew += p.cfg.die_punishment;                              // :359
        hp = valid ? hp_new : hp;
        alive = valid && hp > 0;
        er.tick = tick;
        if constexpr (N == 1) {
            er.bhp_b -= team == 0 ? nbase : nbase_other;     // red shooters damage the blue base
            er.bhp_r -= team == 0 ? nbase_other : nbase;
        } else {
            er.bhp_b -= s_bhit[gl + 0];                      // red shooters damage the blue base
            er.bhp_r -= s_bhit[gl + 1];
        }
        if (er.bhp_b <= 0) {                             // blue base dead: every red plane gets lose_punishment; red wins
            if (team == 0) rew += p.cfg.lose_punishment;
            er.winner = BSX_WINNER_RED; er.done = 1; cnt_delta.x += 1; cnt_delta.z += 1;
        }
        if (er.bhp_r <= 0) {
            if (team == 1) rew += p.cfg.lose_punishment;
            er.winner = BSX_WINNER_BLUE; er.done = 1; cnt_delta.x += 1; cnt_delta.w += 1;
        }
    } else if (mode == M_TIE) {
        er.tick = tick;
        er.winner = BSX_WINNER_TIE; er.done = 1; cnt_delta.x += 1; cnt_delta.y += 1;
    }
    }
