// bsx_step_split_out_body.inl -- the OUTPUTS wave of the two-wave multi-tick kernel, form 2 (bsx_step_split.h; launches of more than 32 768 games): per tick it meets the game
// wave, takes the 16 bytes per agent that wave published (post-move position, heading, flags, enemy base, reward), and runs the
// observation geometry and the output stores of bsx_step_kernel's phase files on them -- none of the game logic, no loads.
    {
        RawIn rin_next = {}; DecIn din_next = {-1, 0.0};
        (void)rin_next; (void)din_next;
        for (int tk = 0; tk < p_.T; ++tk) {
            ix_t gt = g, EAt = EA;
            asm volatile("" : "+v"(gt));
            asm volatile("" : "+s"(EAt));
            typedef const StepArgs __attribute__((address_space(4))) StepArgsK;
            const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            auto& p = *reinterpret_cast<StepArgsK*>(ka + 64);
            const int lane = tid;
            const int gl = tid & ~(G - 1);
            const int team = (a < n) ? 0 : 1;
            const int eb = gl + (team == 0 ? n : 0);
            float* const obs_t = p.obs + int64_t(tk) * p.obs_ts;
            float* const rew_t = p.rew + int64_t(tk) * p.rew_ts;
            uint8_t* const done_t = p.done + int64_t(tk) * p.done_ts;
            (void)eb; (void)EAt; (void)gl;
            split_rendezvous();                          // the game wave has published this tick
            const v4u_t pb = s_pub[(tk & 1) * SPB + tid];
            int x = sx16(pb.x), y = sy16(pb.x), hp = 0;
            double dir = double(int(pb.y & 511u));
            const bool alive = (pb.y & 512u) != 0u;
            const double rew = double(__uint_as_float(pb.z));
            EnvU er = {};
            er.bbx = er.brx = sx16(pb.w); er.bby = er.bry = sy16(pb.w);
            er.done = (pb.y & 2048u) ? 1 : 0;
            const int mode = M_INERT, nhp_ = (pb.y & 1024u) ? 1 : 0, nplane = 0;      // (the row's "enemy is alive" flag arrives worked out)
            const int nx_ = lane_xor1(x), ny_ = lane_xor1(y);
            // names of the game logic that the two phase files mention behind guards that are off here
            constexpr bool CHEAP_SHOT = true;
            const bool shot_exact = false, spawn = false, nexact = false;
            const double nbdir = 0.0; double2 nd = make_double2(0.0, 0.0); const int ks = 0;
            const int4 cnt_delta = make_int4(0, 0, 0, 0); const uint32_t games = 0u; const uint32_t pc = 0u;
            (void)hp; (void)nbdir; (void)nd; (void)ks; (void)cnt_delta; (void)games; (void)pc; (void)shot_exact; (void)spawn; (void)nexact; (void)mode;
#include "bsx_step_phase_geometry.inl"
#include "bsx_step_phase_stores.inl"
        }
    }
