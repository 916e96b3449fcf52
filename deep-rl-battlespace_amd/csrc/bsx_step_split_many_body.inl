// bsx_step_split_many_body.inl -- one wave's T ticks of bsx_step_split_kernel<.., MANY = true> (bsx_step_split.h includes it twice, with the
// R_* constants of the wave's role): bsx_step_kernel's multi-tick loop -- state in registers, next tick's inputs fetched a tick ahead, the
// kernel's arguments read through the kernarg segment per tick -- around the seven phase files.
    {
        int x = 0, y = 0, hp = 0;
        uint32_t games = 0;
        double dir = 0.0;
        EnvU er = {};
        uint32_t pc = 0;
        uint2 pool_first = make_uint2(0u, 0u);
        RawIn rin_next = {};
        DecIn din_next = {-1, 0.0};
        load_inputs(0, rin_next); din_next = decode(rin_next);
        for (int tk = 0; tk < p_.T; ++tk) {
            ix_t gt = g, EAt = EA;
            uint64_t seed_t = p_.seed;
            int64_t env_offset_t = p_.env_offset;
            constexpr int tie_tick = tie_tick_const(1);
            asm volatile("" : "+v"(gt));
            asm volatile("" : "+s"(EAt));
            asm volatile("" : "+s"(seed_t));
            typedef const StepArgs __attribute__((address_space(4))) StepArgsK;
            const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            auto& p = *reinterpret_cast<StepArgsK*>(ka + 64);
            const int lane = tid;
            const int gl = tid & ~(G - 1);
            const int team = (a < n) ? 0 : 1;
            const int eb = gl + (team == 0 ? n : 0);
            const double* const u_t = p.u ? p.u + int64_t(tk) * p.u_ts : p.u;
            float* const obs_t = p.obs + int64_t(tk) * p.obs_ts;
            float* const rew_t = p.rew + int64_t(tk) * p.rew_ts;
            uint8_t* const done_t = p.done + int64_t(tk) * p.done_ts;
            (void)eb; (void)u_t; (void)obs_t; (void)rew_t; (void)done_t; (void)EAt; (void)env_offset_t;
            if (tk == 0) {
                const uint2 ecw = *elem(envc_, ix_t(ec));
                const uint2 edw = *elem(envd_, ix_t(ec));
                const uint2 prw = *elem(plane_, gt);
                if constexpr (R_BULLETS) {
                    pool_first = *elem(bent_, pool0 + ix_t(lane));
                    pc = __builtin_amdgcn_readfirstlane(*elem(bcnt_, ix_t(wblk)));
                }
                unpack_plane(prw, x, y, hp, dir);
                er = unpack_env(ecw, edw.x);
                games = edw.y;
            } else if constexpr (R_BULLETS) {
                pool_first = *elem(bent_, pool0 + ix_t(lane));   // this tick's first 64 entries (the last tick's stores precede this load in program order)
            }
            const DecIn din = din_next;
            int act = din.act;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, uu_in = din.uu;
            (void)a0; (void)a1; (void)a2;
#include "bsx_step_phase_actor.inl"
#include "bsx_step_phase_shot.inl"
#include "bsx_step_phase_move.inl"
#include "bsx_step_phase_geometry.inl"
#include "bsx_step_phase_bullets.inl"
#include "bsx_step_phase_outcome.inl"
#include "bsx_step_phase_stores.inl"
            games += uint32_t(cnt_delta.x);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
