// bsx_step_phase_respawn.inl -- part of the move PHASE (bsx_step_phase_move.inl includes it under `mode == M_RESET`: in place, or -- the per-call
// two-wave kernel's first wave -- after the pose hand-over, once the geometry wave's Philox block is there): the in-kernel re-spawn of a
// finished game (Plane.reset sprites.py:74-91, Base.reset :238-252, parallel_env.reset battle_env.py:246-279) from the block `rw`.
// Writes x, y, dir, hp, er, tick.
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
