// bsx_state.h -- constants of the game, the state block's layout (ABI 14), record and pool-entry formats, the integer bullet step
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h) and by the three translation units that instantiate the step kernels; namespace bsxk.
#pragma once

namespace bsxk {


constexpr int K = BSX_BULLET_SLOTS;
constexpr int TPB = 256;   // reset / export kernels
constexpr int SPB = 64;    // step kernel: a game never spans a wavefront, so the waves of a workgroup share nothing and no block barrier is needed
constexpr int WPB = 1;     // wavefronts per workgroup of the per-step / multi-tick kernels (the fused rollout has its own: 32 games per workgroup;
                           // 2 / 4 waves measured slower: 8.54 / 8.39 us against 8.20)

// obs / rew / done leave with the non-temporal hint: nothing on the step path reads them back, so they need not sit dirty in
// the L2 until the end-of-kernel write-back (C2: 8.21 -> 8.02 us per step against ordinary stores).
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <class T> __device__ inline void out_store(T* p, T v) { __builtin_nontemporal_store(v, p); }

#define BSX_LDS(T, arr) ((__attribute__((address_space(3))) T*)(uintptr_t)(arr))

// The per-step kernel's STATE stores (plane / game records, bullet entries) can leave non-temporal as well -- the next launch
// finds the L2 invalidated anyway.  Measured (same box, A/B): 65 536 x 4v4 28.6 -> 27.8 us, but 65 536 x 1v1 8.22 -> 8.44 and
// 1 M x 1v1 73.4 -> 74.5: used for team sizes >= 2 only (NT_STATE below), never inside a multi-tick launch (the same wave reads
// its bullet rows back one tick later).
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
template <bool NT, class T> __device__ inline void st_store(T* p, T v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ inline v4u_t as_v4u(uint4 v) { return v4u_t{v.x, v.y, v.z, v.w}; }

constexpr int FIELD_W = 1200, FIELD_H = 800;      // sprites.py:9-10
constexpr int PLANE_HW = 25, PLANE_HH = 24;        // 50x48 sprite, half sizes (w>>1, h>>1)
constexpr int PLANE_HP = 4;                        // battle_env.py:92
constexpr double DEG2RAD = 3.141592653589793238462643383279502884 / 180.0;  // CPython math.radians
constexpr double RAD2DEG = 180.0 / 3.141592653589793238462643383279502884;  // CPython math.degrees
constexpr double TWO_PI = 2.0 * 3.141592653589793;                            // 2*math.pi
constexpr double FIELD_DIAG = 1442.2205101855957;  // sqrt(1200^2 + 800^2), battle_env.py:230
constexpr double BULLET_STEP = 45.0;               // 450 * 0.1 in binary64
constexpr double TIME_STEP = 0.1;

// ---------------------------------------------------------------------------------------------- state layout
// Records are sized by what a call MOVES: every field a step() rewrites sits in an 8-byte record of its own array, what a game never
// changes (its base positions) in another, and what only a game's end touches (the win / tie counters) is updated there by atomics.
//   plane  uint2 [E*A]   .x = x | y << 16 (sprite centre, pygame Rect ints)
//                        .y = heading in whole degrees (9 bits, 0..360) | hp << 9 (3 bits; alive <=> hp > 0, sprites.py:143-153)
//                             | 1 << 12: the heading is fractional and lives in `pdirf` (continuous actions only)
//   pdirf  double [E*A]  heading in degrees, [0, 360]: read and written by the continuous kernels only
//   envc   uint2 [E]     base centres: .x = red x | y << 16, .y = blue x | y << 16; written by reset / auto-reset only
//   envd   uint2 [E]     .x = red base hp (9 bits, signed: may go negative within a step, sprites.py:260-262) | blue base hp << 9
//                             | tick << 18 (9 bits: total_time == tick * 0.1 accumulated) | done << 27 | winner << 28
//                        .y = games this slot has finished = the episode number the random streams are keyed by
//   cnt    int4 [E]      games, ties, red wins, blue wins: touched at a game's end only (atomic adds; export reads them)
//   bullets: one POOL per wave block (the 64 lanes = 64 / G games a wavefront of the step kernel owns): `bcnt[block]` entries, dense,
//            in no particular order, at `bent[block * POOL_CAP ...]`; an entry names its owner lane.  The wave reads its pool with
//            fully coalesced loads whatever the bullets' distribution over the planes (the first 64 entries unconditionally, in the
//            first batch of loads: no dependent round trip) and writes the survivors back compacted.
constexpr int POOL_CAP = 64 * BSX_BULLET_SLOTS;   // every lane of a wave block holding 11 older bullets + this call's shot
struct Layout { size_t lut, envc, envd, cnt, plane, pdirf, bcnt, bent, bdir, bd, total; };

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
__host__ __device__ constexpr int group_width(int n) {
    int g = 2;
    while (g < 2 * n) g <<= 1;
    return g;
}
__host__ __device__ inline int64_t wave_blocks(int64_t E, int n) { const int epb = 64 / group_width(n); return (E + epb - 1) / epb; }

__host__ __device__ inline Layout make_layout(int64_t E, int n) {
    Layout L;
    const size_t EA = size_t(E) * size_t(2 * n);
    const size_t NB = size_t(wave_blocks(E, n));
    size_t o = 0;
    L.lut = o;   o = align256(o + 361 * sizeof(double2));
    L.envc = o;  o = align256(o + size_t(E) * sizeof(uint2));
    L.envd = o;  o = align256(o + size_t(E) * sizeof(uint2));
    L.cnt = o;   o = align256(o + size_t(E) * sizeof(int4));
    L.plane = o; o = align256(o + EA * sizeof(uint2));
    L.pdirf = o; o = align256(o + EA * sizeof(double));
    L.bcnt = o;  o = align256(o + NB * sizeof(uint32_t));
    // A pool entry = two words: .x = x (11 bits) | age << 11 | exact-path flag << 15 | y << 16 (10 bits) | owner lane << 26, rewritten by
    // every update; .y = the step code (step_code below), written by the shot.
    L.bent = o;  o = align256(o + NB * size_t(POOL_CAP) * sizeof(uint2));
    L.bdir = o;  o = align256(o + size_t(K) * EA * sizeof(double));   // [K][EA]: heading, RING by birth tick % 12 (export only)
    L.bd = o;    o = align256(o + size_t(K) * EA * sizeof(double2));  // [K][EA]: float64 step (45cos, 45sin), RING by birth tick % 12, of the RARE shots whose
                                                                      //          step code carries the exact-path flag; never read or written otherwise
    L.total = o;
    return L;
}

struct StatePtrs {
    const double2* lut; uint2* envc; uint2* envd; int* cnt; uint2* plane; double* pdirf; uint32_t* bcnt; uint2* bent; double* bdir; double2* bd;
};
inline StatePtrs state_ptrs(void* base, int64_t E, int n) {
    Layout L = make_layout(E, n);
    char* b = static_cast<char*>(base);
    return StatePtrs{reinterpret_cast<const double2*>(b + L.lut), reinterpret_cast<uint2*>(b + L.envc), reinterpret_cast<uint2*>(b + L.envd),
                     reinterpret_cast<int*>(b + L.cnt), reinterpret_cast<uint2*>(b + L.plane), reinterpret_cast<double*>(b + L.pdirf),
                     reinterpret_cast<uint32_t*>(b + L.bcnt), reinterpret_cast<uint2*>(b + L.bent), reinterpret_cast<double*>(b + L.bdir),
                     reinterpret_cast<double2*>(b + L.bd)};
}

// Wave votes on a BOOL: the builtin takes the lane mask as it stands.  HIP's __ballot / __any take an int -- the compiler materialises the
// predicate as 0 / 1 in a VGPR and compares it with zero again: two vector instructions per vote, ~10 votes on the step's hot path.
__device__ inline unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// the value of the lane next door (lane ^ 1): one DPP move (quad_perm [1,0,3,2]); __shfl_xor compiles to an LDS permute with its index math
__device__ inline int lane_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ inline uint32_t rotl12(uint32_t v, int s) {  // rotate a 12-bit mask left by s in [0, 12)
    return ((v << s) | (v >> (12 - s))) & 0xFFFu;
}
__device__ inline int sx16(uint32_t w) { return int(int16_t(w & 0xFFFFu)); }
__device__ inline int sy16(uint32_t w) { return int(int16_t(w >> 16)); }
__device__ inline uint32_t pack_xy(int x, int y) { return (uint32_t(x) & 0xFFFFu) | (uint32_t(y) << 16); }
// Pool entry, two words.  .x = x (11 bits) | age << 11 (4 bits: updates so far, 15 = tombstone) | exact-path flag << 15 |
// y << 16 (10 bits) | owner lane << 26: a stored bullet is inside the field (0..1200, 0..800), and the two coordinates sit in the two
// 16-bit halves so that the move and every rectangle test below work on both at once (v_pk_*_i16).  .y = the per-update step as two
// signed 16-bit halves.
constexpr uint32_t TOMBSTONE_AGE = 15;
constexpr uint32_t ENT_XY = 0x03FF07FFu, ENT_AGE = 0x7800u, ENT_EXACT = 0x8000u;
constexpr int ENT_OWNER_SHIFT = 26;                     // bits 26..31: the owner's lane in its wave block
__device__ inline uint32_t pack_bullet(int x, int y, int age) { return uint32_t(x) | (uint32_t(age) << 11) | (uint32_t(y) << 16); }
__device__ inline int bullet_x(uint32_t w) { return int(w & 0x7FFu); }
__device__ inline int bullet_y(uint32_t w) { return int((w >> 16) & 0x3FFu); }
__device__ inline int bullet_age(uint32_t w) { return int((w >> 11) & 0xFu); }
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ inline s16x2 as_pk(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ inline uint32_t pk_bits(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
constexpr uint32_t pk_const(int lo, int hi) { return (uint32_t(lo) & 0xFFFFu) | (uint32_t(hi) << 16); }
constexpr int PK_BIAS = 64;                        // > 46: a moved bullet's biased coordinates are positive
// 0 / -1: is the sign bit of either half set?
__device__ inline int pk_any_negative(uint32_t t) { return int(t | (t << 16)) >> 31; }
// The step.  Bullet.update's move (sprites.py:330-333) is rect.center = (x + 45 cos, y + 45 sin) in binary64 from the INTEGER centre,
// the store truncating toward zero.  For d = 45 cos with f = floor(d) and r = d - f: the exact sum s = x + d lies at distance
// min(r, 1 - r) from an integer and the binary64 sum fl(x + d) is at most 2^-43 away from s (|s| < 2048), so whenever r stays
// 2^-40 away from 0 and 1 the rounded sum lies strictly between the same two integers n = x + f and n + 1 as s, and int() of it
// is n for n >= 0 and n + 1 for n < 0 (truncation toward zero: x = 3, d = -3.5 -> 0) -- integer arithmetic, exactly the
// reference's result.  So a pool entry carries f for both axes (|d| <= 45) and, for the other case, a flag: such a shot also stores its
// float64 step (ring `bd`) and its updates take the float64 sum, as every bullet did before round 3.
// The shot decides in float32, with a wider guard: |float(d) - d| <= 2^-19 for |d| < 64, so a float32 fraction in
// [2^-17, 1 - 2^-17] puts d itself at least 2^-18 from every integer -- floor(float(d)) is floor(d) and r is far inside the band.
// The flag is then set for one shot in ~30 000 (and for headings on an axis: scripted tests); the float64 path it selects is a
// 16-byte load and two adds behind a branch that a wave takes only if one of the entries it is about to update carries the flag.
constexpr float STEP_GUARD = 0x1p-17f;
__device__ inline uint32_t step_code(double dx, double dy, bool& exact) {
    const float dxf = float(dx), dyf = float(dy);
    const float fx = floorf(dxf), fy = floorf(dyf);
    const float rx = dxf - fx, ry = dyf - fy;                       // exact
    exact = !(fminf(rx, ry) >= STEP_GUARD && fmaxf(rx, ry) <= 1.0f - STEP_GUARD);
    return pk_bits(__builtin_amdgcn_cvt_pk_i16(int(fx), int(fy)));
}
// (x, y) + (fx, fy) with the truncation toward zero of a negative sum, both halves at once
__device__ inline uint32_t step_pk(uint32_t xy, uint32_t code) {
    const s16x2 b = as_pk(xy) + as_pk(code);
    return pk_bits(b - (b >> 15));
}
__device__ inline int ring_pos(int ks, int back) { const int q = ks - back; return q + ((q >> 31) & BSX_BULLET_SLOTS); }   // (ks - back) mod 12, 0 <= back < 12

// Record (un)packing on raw words (layout: see make_layout).
constexpr uint32_t PLANE_FRAC = 1u << 12;               // plane word 1: the heading is fractional and lives in pdirf
__device__ inline void unpack_plane(const uint2 w, int& x, int& y, int& hp, double& dir) {
    x = sx16(w.x); y = sy16(w.x); hp = int((w.y >> 9) & 7u);
    dir = double(int(w.y & 511u));                      // whole degrees; a continuous kernel replaces it by pdirf when PLANE_FRAC is set
}
__device__ inline uint2 pack_plane(int x, int y, int hp, double dir, bool frac) {
    return make_uint2(pack_xy(x, y), (uint32_t(int(dir)) & 511u) | (uint32_t(hp) << 9) | (frac ? PLANE_FRAC : 0u));
}
struct EnvU {   // a game's record in registers
    int brx, bry, bbx, bby, bhp_r, bhp_b, tick, done, winner;
};
__device__ inline EnvU unpack_env(const uint2 c, const uint32_t d) {
    EnvU e;
    e.brx = sx16(c.x); e.bry = sy16(c.x); e.bbx = sx16(c.y); e.bby = sy16(c.y);
    e.bhp_r = int(d << 23) >> 23; e.bhp_b = int(d << 14) >> 23; e.tick = int((d >> 18) & 511u); e.done = int((d >> 27) & 1u); e.winner = int((d >> 28) & 3u);
    return e;
}
__device__ inline uint2 pack_envc(const EnvU& e) { return make_uint2(pack_xy(e.brx, e.bry), pack_xy(e.bbx, e.bby)); }
__device__ inline uint32_t pack_envd(const EnvU& e) {
    return (uint32_t(e.bhp_r) & 511u) | ((uint32_t(e.bhp_b) & 511u) << 9) | (uint32_t(e.tick) << 18) | (uint32_t(e.done) << 27) | (uint32_t(e.winner) << 28);
}
static_assert(5 * BSX_MAX_N < 256 && 12 * BSX_MAX_N < 256, "base hit points (start 5n, at most 12n hits in one call) fit 9 signed bits");

}  // namespace bsxk
