// bsx_step_split_geom_body.inl -- the GEOMETRY wave of bsx_step_split_kernel's per-call form (bsx_step_split.h includes it): no loads, no game
// logic -- it waits until the first wave has moved the planes, takes the poses from LDS, works out the observation geometry (the SAME phase
// file, bsx_step_phase_geometry.inl) and hands the four observation values per agent back before the stores.
    {
        split_rendezvous();                              // the first wave has moved its planes (bsx_step_phase_move.inl, R_POSE_LDS)
        STAMP(0); STAMP(1); STAMP(2);                    // (diagnostic builds: this wave's row of stamps -- waited for the poses | geometry | waited for the stores)
        const v4u_t h0 = s_t0[tid], h1 = s_t1[tid];
        const int x = int(h0.x), y = int(h0.y), nx_ = int(h0.z), ny_ = int(h0.w);
        const double dir = __hiloint2double(int(h1.y), int(h1.x));
        EnvU er = {};
        er.bbx = er.brx = int(h1.z); er.bby = er.bry = int(h1.w);   // (the phase reads the enemy base through my team: both are it)
        // what the phase file names besides: the exact-shot store is the first wave's (R_BULLETS), this wave takes no part in it
        constexpr bool R_BULLETS = false, R_GEOM = true, CHEAP_SHOT = false;
        constexpr int R_GEOM_LDS = 1;
        const int team = 0, lane = tid, ks = 0, gl = tid & ~(G - 1), eb = gl;
        const bool shot_exact = false, spawn = false, nexact = false;
        double nbdir = 0.0;
        double2 nd = make_double2(0.0, 0.0);
        const ix_t gt = g, EAt = 0;
        (void)lane; (void)ks; (void)shot_exact; (void)spawn; (void)nexact; (void)nbdir; (void)nd; (void)gt; (void)EAt; (void)team; (void)gl; (void)eb;
#include "bsx_step_phase_geometry.inl"
        STAMP(4); STAMP(5); STAMP(6);
        split_rendezvous();                              // the hand-over before the stores (bsx_step_phase_stores.inl, R_GEOM_LDS)
        STAMP(7);
    }
