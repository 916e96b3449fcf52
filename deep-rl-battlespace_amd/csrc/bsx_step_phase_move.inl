// bsx_step_phase_move.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): in-kernel re-spawn (M_RESET) or process_action's move (battle_env.py:383-424), then the hand-off of post-move poses to the game's other planes
// (1v1: DPP; larger teams: wave-private LDS) and the planes' sprites as rectangles for the work slots.  Writes: x, y, dir, hp, er (re-spawn),
// nx_, ny_, nhp_ (1v1), s_x, s_y, s_hp, s_bhit, s_pq.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads   CHEAP_SHOT alive0 d0 dir_rot dl ks mode nexact spawn srank sx0 sy0
// @writes  a0 a1 dir er hp nbdir ncode nd rw shot_exact tick x y
// @exports nhp_ nx_ ny_
// @lds     s_bhit s_hp s_new s_pq s_t0 s_x s_y
    STAMP(2);
    // re-spawn in place of the inert call; episode id = games played so far.  My block holds my pose and (first plane of a team) my
    // team's base; the two bases travel to every lane of the game (all lanes of a game are here together)
    if constexpr (R_MOVE) {                              // (split kernels: a wave that only runs bullets does not move planes)
    if (mode == M_RESET) {
        if constexpr (R_DRAW_LDS != 2) {                 // (R_DRAW_LDS == 2: after the pose hand-over, below -- the block comes from the geometry wave)
#include "bsx_step_phase_respawn.inl"
        }
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
    // for the geometry wave, which has waited for it (16 bytes per lane; the geometry wave repeats no game logic)
    if constexpr (R_POSE_LDS) {
        const double dq = dir;
        // (one 16-byte word per lane: position, heading, enemy base; the enemy's position is the lane next door's -- the geometry wave takes it by DPP)
        s_t0[tid] = v4u_t{pack_xy(x, y), uint32_t(__double2loint(dq)), uint32_t(__double2hiint(dq)), pack_xy(team == 0 ? er.bbx : er.brx, team == 0 ? er.bby : er.bry)};
        split_rendezvous();
        if constexpr (R_DRAW_LDS == 2) {
            // The call's Philox block -- the shot's jitter, or the pose of a plane whose game this call re-spawns -- was computed by the geometry
            // wave while this wave classified and moved (the same block from the same key: bit-identical), and is in LDS since before that wave
            // arrived at the rendezvous above.  A re-spawned game's lanes take their new pose HERE: nothing between the move and this point reads
            // it (their sprites and enemy base are staged for work slots that a re-spawn drops; the geometry wave works the new pose out itself).
            const v4u_t w4 = s_rw[tid];
            rw = make_uint4(w4.x, w4.y, w4.z, w4.w);
            if (mode == M_RESET) {
#include "bsx_step_phase_respawn.inl"
            }
            nhp_ = lane_xor1(valid ? hp : 0);
#include "bsx_step_phase_shot_entry.inl"
            shot_exact = any64(spawn && nexact);
        }
    }
