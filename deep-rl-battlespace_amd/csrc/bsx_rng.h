// bsx_rng.h -- the in-kernel random draws: Philox4x32-10 keyed by (seed, global game, stream, episode, tick | agent); spawn draws with the reference's ranges
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h); everything lives in the translation unit's anonymous namespace.
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------- Philox4x32-10
using bsx_actor::philox4x32_10;   // one definition, shared with the actor's exploration noise (bsx_actor_core.h)
enum : uint32_t { STREAM_RESET = 0, STREAM_AUTORESET = 1, STREAM_JITTER = 2 };
__device__ inline uint4 draw4(uint64_t seed, int64_t genv, uint32_t stream, uint32_t seq, uint32_t who) {
    return philox4x32_10(make_uint4(uint32_t(genv), uint32_t(uint64_t(genv) >> 32) ^ (stream << 28), seq, who),
                         make_uint2(uint32_t(seed), uint32_t(seed >> 32)));
}
// inclusive integer range, multiply-shift
__device__ inline int randint(uint32_t r, int lo, int hi) { return lo + int(__umulhi(r, uint32_t(hi - lo + 1))); }
// 53-bit uniform in [0,1), the construction CPython's random.random() uses on two 32-bit words
__device__ inline double uniform53(uint32_t a, uint32_t b) {
    return (double(a >> 5) * 67108864.0 + double(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Spawn draws (sprites.py:74-91,238-252).  Every lane of an env computes the same base draws.
template <class ENV>
__device__ inline void spawn_bases(uint64_t seed, int64_t genv, uint32_t stream, uint32_t seq, ENV& er) {
    const uint4 r = draw4(seed, genv, stream, seq, 0xFFFFu);
    er.brx = randint(r.x, 62, 379);     // randint(w, (W-w)//3)
    er.bry = randint(r.y, 62, 738);
    er.bbx = randint(r.z, 758, 1138);   // randint((W-w)//3*2, W-w)
    er.bby = randint(r.w, 62, 738);
}
__device__ inline void spawn_plane(uint64_t seed, int64_t genv, uint32_t stream, uint32_t seq, int a, int n,
                                   int& x, int& y, double& dir) {
    const uint4 r = draw4(seed, genv, stream, seq, uint32_t(a));
    if (a < n) {
        x = randint(r.x, 50, 383); y = randint(r.y, 48, 752);
        int d = randint(r.z, 270, 450);
        if (d >= 360) d -= 360;
        dir = double(d);
    } else {
        x = randint(r.x, 766, 1150); y = randint(r.y, 48, 752);
        dir = double(randint(r.z, 90, 270));
    }
}

}  // namespace
