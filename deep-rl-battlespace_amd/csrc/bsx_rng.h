// bsx_rng.h -- the in-kernel random draws: Philox4x32-10 keyed by (seed, global game, stream, episode, tick | agent); spawn draws with the reference's ranges
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h) and by the three translation units that instantiate the step kernels; namespace bsxk.
#pragma once

namespace bsxk {

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

// Spawn draws (sprites.py:74-91,238-252): ONE Philox block per plane -- counter (stream, episode, who = plane id) -- yields the plane's pose
// and, for the first plane of each team, its team's base; the game's lanes then exchange the two base positions.  x and the heading come
// from one word as a mixed-radix pair (the product's high word, then the low word scaled again: together one multiply-shift onto range x 181
// cells, bias < 2e-5), y and the base coordinates from a word each, all inclusive ranges as the reference draws them:
//   red plane   x randint(50, 383)   y randint(48, 752)   heading randint(270, 450) - 360 if >= 360      (sprites.py:82-87)
//   blue plane  x randint(766, 1150) y randint(48, 752)   heading randint(90, 270)                       (sprites.py:88-91)
//   red base    x randint(62, 379)   y randint(62, 738);   blue base  x randint(758, 1138)  y randint(62, 738)   (sprites.py:246-252)
struct SpawnDraw { int x, y, dir, bx, by; };
__device__ inline SpawnDraw spawn_from_words(const uint4 r, int a, int n) {
    const bool red = a < n;
    const uint64_t p = uint64_t(r.x) * uint32_t(red ? 334 : 385);
    SpawnDraw s;
    s.x = (red ? 50 : 766) + int(uint32_t(p >> 32));
    int d = (red ? 270 : 90) + int(__umulhi(uint32_t(p), 181u));
    s.dir = d >= 360 ? d - 360 : d;
    s.y = 48 + int(__umulhi(r.y, 705u));
    s.bx = (red ? 62 : 758) + int(__umulhi(r.z, red ? 318u : 381u));
    s.by = 62 + int(__umulhi(r.w, 677u));
    return s;
}

}  // namespace bsxk
