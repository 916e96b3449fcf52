// bsx_step_split_body.inl -- the FIRST wave's call in bsx_step_split_kernel's per-call form (bsx_step_split.h includes it with that wave's R_*
// constants): the first loads, the decode, and the seven phase files of bsx_step_kernel in tick order -- everything but the observation
// geometry, which the second wave works out from the post-move poses this wave leaves in LDS (bsx_step_split_geom_body.inl).
    {
        ix_t gt = g, EAt = ix_t(p.E) * ix_t(A);
        uint64_t seed_t = p.seed;
        int64_t env_offset_t = p.env_offset;
        constexpr int tie_tick = tie_tick_const(1);
        const int lane = tid;
        const int gl = tid & ~(G - 1);
        const int team = (a < n) ? 0 : 1;
        const int eb = gl + (team == 0 ? n : 0);
        const double* const u_t = p.u;
        float* const obs_t = p.obs;
        float* const rew_t = p.rew;
        uint8_t* const done_t = p.done;
        (void)eb; (void)u_t; (void)obs_t; (void)rew_t; (void)done_t; (void)EAt;
        RawIn rin_next = {}; DecIn din_next = {-1, 0.0};             // (named by the phases behind `if (MULTI ...)`: never reached in this form)
        (void)rin_next; (void)din_next;
        // ---- T0: every load of the call, back to back
        int x = 0, y = 0, hp = 0;
        uint32_t games = 0;
        double dir = 0.0;
        EnvU er = {};
        uint32_t pc = 0;
        uint2 pool_first = make_uint2(0u, 0u);
        int act = -1;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, uu_in = 0.0;
        (void)a0; (void)a1; (void)a2;
        STAMP(0);
        {
            uint2 ecw, edw, prw;
            double dirf = 0.0; (void)dirf;
            ecw = *elem(envc_, ix_t(ec));
            edw = *elem(envd_, ix_t(ec));
            prw = *elem(plane_, gt);
            const char* const abase = static_cast<const char*>(act_);
            float4 lg = make_float4(0.f, 0.f, 0.f, 0.f);
            int ai = -1;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0;
            if constexpr (CONT) {                        // (as bsx_step_kernel's load_inputs: the triple by encoding, uniform branches)
                dirf = *elem(p.st.pdirf, gt);
                if (has_act) {
                    if (kind_ == BSX_ACT_F32) {
                        const float* ap = static_cast<const float*>(act_) + 3 * g;
                        f0 = ap[0]; f1 = ap[1]; f2 = ap[2];
                    } else if (kind_ == BSX_ACT_F32X4) {
                        const float4 v = static_cast<const float4*>(act_)[g];
                        f0 = v.x; f1 = v.y; f2 = v.z;
                    } else {
                        const double* ap = static_cast<const double*>(act_) + 3 * g;
                        c0 = ap[0]; c1 = ap[1]; c2 = ap[2];
                    }
                }
            } else if constexpr (LG) lg = *reinterpret_cast<const float4*>(elem(abase, has_act ? g * 16 : ix_t(0)));
            else ai = *reinterpret_cast<const int32_t*>(elem(abase, has_act ? g * 4 : ix_t(0)));
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+s"(seed_t), "+s"(env_offset_t));
            if constexpr (CONT) {
                if (has_act) {
                    if (kind_ == BSX_ACT_F32 || kind_ == BSX_ACT_F32X4) { a0 = double(f0); a1 = double(f1); a2 = double(f2); }
                    else { a0 = c0; a1 = c1; a2 = c2; }
                }
            } else if (has_act) act = LG ? argmax4(lg.x, lg.y, lg.z, lg.w) : ai;
            pool_first = *elem(bent_, pool0 + ix_t(lane));
            pc = __builtin_amdgcn_readfirstlane(*elem(bcnt_, ix_t(wblk)));
            if (p.u) uu_in = p.u[g];                     // (injected jitter: uniform branch)
            unpack_plane(prw, x, y, hp, dir);
            if constexpr (CONT) dir = (prw.y & PLANE_FRAC) ? dirf : dir;
            er = unpack_env(ecw, edw.x);
            games = edw.y;
        }
#include "bsx_step_phase_actor.inl"
#include "bsx_step_phase_shot.inl"
#include "bsx_step_phase_move.inl"
#include "bsx_step_phase_geometry.inl"
#include "bsx_step_phase_bullets.inl"
#include "bsx_step_phase_outcome.inl"
#include "bsx_step_phase_stores.inl"
        STAMP(7);
    }
