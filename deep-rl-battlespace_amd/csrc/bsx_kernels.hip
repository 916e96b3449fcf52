// bsx_kernels.hip -- the batched Battlespace step() path for MI355X (gfx950 / CDNA4), behind include/battlespace_hip.h.
//
// One thread per agent, one fused launch per step(): plane kinematics -> bullet spawn -> bullet flight / miss / base
// hit / plane-overlap classification -> ordered plane-hit resolve -> win / tie -> rewards, dones -> observations.
// Reference behaviour followed (paths relative to the reference repo): envs/battle_env.py:281-381 (step),
// :383-424 (process_action), :202-244 (observe), :38-58 (rel_angle, dist), :246-279 (reset), :469-496 (tie/win);
// envs/sprites.py:35-42 (calc_new_xy), :74-153 (Plane), :238-263 (Base), :293-351 (Bullet).
//
// Mapping to the hardware
//   * lane = agent, lanes of one env are adjacent (group width G = next pow2 >= 2n, G <= 32), so an env never
//     straddles a 64-wide wavefront and a workgroup is ONE wavefront: LDS hand-offs are wave-private, no s_barrier.
//     Every per-agent array is struct-of-arrays indexed e*A + a: consecutive lanes touch consecutive 16-byte records.
//   * the kernel is issue- and boundary-bound at 65 536 games (two waves per SIMD) and issue-bound at a million, so: all
//     independent loads go out in one batch as raw 16-byte words (clamped indices, no per-lane branches); the dependent loads
//     (heading-table entry, bullet entries) are covered by the shot's Philox + sincos and the fp64 observation math; predicates
//     are integer sign masks, not SGPR lane masks.
//   * bullets live in one POOL per wave block (the 64 lanes a wavefront owns): a dense, unordered array of 8-byte entries -- position,
//     age and owner lane in one word, the per-update step as an INTEGER code in the other (written once by the shot; exactly the
//     reference's float64 add-then-truncate, see step_code).  The wave reads its pool with fully coalesced loads, the first 64 entries
//     in the first batch of loads (nothing on the common path waits for a dependent load but the heading-table entry), updates the
//     bullets of ALL its lanes in WORK SLOTS (slot = pool entry; this call's shots queue up behind them) -- under sparse play one round
//     of 64 slots -- and writes the survivors back compacted by a wave-wide prefix count.  What a slot needs from its bullet's owner is
//     staged per lane in LDS; the outcome returns to the owner through one LDS add per bullet that ended.  The order in which the
//     reference resolves plane hits (creation order) is the bullets' AGE, which the entries carry.
//   * post-move plane poses and hit points are handed to the other planes of the game by cross-lane moves (1v1) or wave-private
//     LDS (larger teams); the all-pairs range / angle-off geometry of a team pair is computed once per pair; observation rows
//     leave straight from registers with the non-temporal hint (the fused rollout keeps them in LDS for the actor's MFMAs).
//   * the ordered plane-hit resolve walks bullet ages oldest-first; a wavefront ballot skips ages at which no lane of
//     the wave has a candidate (almost all of them), and group ballots give the "nobody left alive" test.
//   * HBM-bound integer/fp64 work, no dense contraction: no MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off   (no FMA contraction: positions are float64 add-then-
// truncate in the reference, sprites.py:130-131,332-333, and must round exactly as CPython rounds them).

#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "battlespace_hip.h"
#include "bsx_actor_core.h"

// Diagnostic builds (tools/build_variant.py compiles this file with -DBSX_VARIANT): timing-only ablations (DIAG bits; results are
// WRONG with any bit set) and in-kernel phase stamps live in bsx_diag.h.  The product build sees the constants below: no
// ablation, stamps compile to nothing, bsx_build_flags() == 0.
#ifdef BSX_VARIANT
#include "bsx_diag.h"
#else
constexpr unsigned DIAG = 0;
constexpr int BUILD_FLAGS = 0;
constexpr int OBS_FORM = 0;
constexpr bool X_CHEAP_ALL = false, X_CORNERS_ALL = false;
constexpr int X_DEPHASE = 0;
constexpr int X_MIN_WAVES = 1;
#define STAMP(i) do { } while (0)
#define FSTAMP(i) do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif

namespace {

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
constexpr int POOL_CAP = 64 * BSX_BULLET_SLOTS;   // every lane of a wave block with a full list (11 older bullets + this call's shot)
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

// ---------------------------------------------------------------------------------------------- game arithmetic
// Plane.forward clamp on the un-rotated 50x48 rect (sprites.py:134-141)
__device__ inline void clamp_plane(int& x, int& y) {
    if (x - PLANE_HW < 0) x = PLANE_HW;
    if (x + PLANE_HW > FIELD_W) x = FIELD_W - PLANE_HW;
    if (y - PLANE_HH <= 0) y = PLANE_HH;
    if (y + PLANE_HH >= FIELD_H) y = FIELD_H - PLANE_HH;
}
// Plane.rotate (sprites.py:99-103): [0, 360] inclusive.  The reference's two `while` loops run at most once each for
// |ang| <= 360 (the discrete turn is 15 degrees, the continuous one at most 35), so they are single selects here.
__device__ inline double rotate_dir(double d, double ang) {
    d += ang;
    d = d > 360.0 ? d - 360.0 : d;
    d = d < 0.0 ? d + 360.0 : d;
    return d;
}
// a * b + c with c a compile-time constant held in an SGPR pair.  gfx950's VOP3 encoding takes no 64-bit literal, and for a Horner
// step the compiler's choice is v_fmac into a VGPR pair it first fills with two v_mov: three vector instructions per coefficient
// where one vector and two scalar ones do -- the scalar unit is otherwise idle here, the vector unit is what the step is bound by.
__device__ inline double fma_k(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
// math.atan2(iy, ix) (battle_env.py:39) for integer pixel differences (|.| < 2^11): the device library's atan2, operation for
// operation -- q = min / max of the magnitudes (correctly rounded quotient: reciprocal estimate, two Newton steps, one residual
// correction), q + q * t * P(t) with t = q * q and its 20-coefficient odd minimax polynomial, then the octant / quadrant selects
// and the sign of y -- minus what integers in this range never need (the quotient's range scaling and fix-up, infinities, NaNs),
// and with the polynomial's coefficients in SGPRs (fma_k): 47 vector instructions instead of 88, the same bits (a device test
// compares it with the library on every argument pair).
// K independent evaluations in lockstep: with two waves per SIMD nothing else fills the ~8 cycles a dependent float64 operation
// waits for its predecessor, so K chains advance together, stage by stage, and every coefficient is materialised once for all K.
__constant__ double ATAN2_COEF[20] = {
    0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13, 0x1.b2bb069efb384p-11, -0x1.7952daf56de9bp-9, 0x1.d6d43a595c56fp-8,
    -0x1.c6ea4a57d9582p-7, 0x1.67e295f08b19fp-6, -0x1.e9ae6fc27006ap-6, 0x1.2c15b5711927ap-5, -0x1.59976e82d3ff0p-5,
    0x1.82d5d6ef28734p-5, -0x1.ae5ce6a214619p-5, 0x1.e1bb48427b883p-5, -0x1.110e48b207f05p-4, 0x1.3b13657b87036p-4,
    -0x1.745d119378e4fp-4, 0x1.c71c717e1913cp-4, -0x1.2492492376b7dp-3, 0x1.99999999952ccp-3, -0x1.5555555555523p-2};
template <int K>
__device__ inline void atan2_pixels_n(const int (&iy)[K], const int (&ix)[K], double (&out)[K]) {
    double ax[K], ay[K], u[K], v[K], y[K], e[K], q[K], r[K], t[K], p[K];
#define BSX_EACH _Pragma("unroll") for (int k = 0; k < K; ++k)
    BSX_EACH { ax[k] = fabs(double(ix[k])); ay[k] = fabs(double(iy[k])); }
    BSX_EACH { u[k] = fmax(fmax(ax[k], ay[k]), 1.0); v[k] = fmin(ax[k], ay[k]); }   // (1.0 only for ix = iy = 0: quotient 0, result 0, as the library's y == 0 case)
    BSX_EACH y[k] = __builtin_amdgcn_rcp(u[k]);
    BSX_EACH e[k] = __builtin_fma(-u[k], y[k], 1.0);
    BSX_EACH y[k] = __builtin_fma(y[k], e[k], y[k]);
    BSX_EACH e[k] = __builtin_fma(-u[k], y[k], 1.0);
    BSX_EACH y[k] = __builtin_fma(y[k], e[k], y[k]);
    BSX_EACH q[k] = v[k] * y[k];
    BSX_EACH r[k] = __builtin_fma(-u[k], q[k], v[k]);
    BSX_EACH q[k] = __builtin_fma(r[k], y[k], q[k]);
    BSX_EACH t[k] = q[k] * q[k];
    constexpr bool TABLE = K <= 2;                       // measured: 1v1 (K = 2) 7.40 -> 7.34 us; 4v4 (K = 3) 24.0 -> 24.4, so literals there
    if constexpr (TABLE) {
        // The 20 coefficients come from constant memory: three scalar loads (8 + 8 + 4 doubles) instead of forty s_mov.  With two
        // waves per SIMD the step is bound by the SIMD's issue port -- one instruction of ANY class per ~4 cycles
        // (tools/micro/issue_rates.hip) -- so what counts is the number of instructions, not which unit runs them.
        typedef const double __attribute__((address_space(4))) * const_f64_ptr;   // constant address space: uniform reads become s_load
        const_f64_ptr C = (const_f64_ptr)(unsigned long long)(&ATAN2_COEF[0]);
        asm("" : "+s"(C));                                   // (an opaque address: otherwise the table is folded back into 40 literal moves)
        BSX_EACH p[k] = __builtin_fma(t[k], C[0], C[1]);
#pragma unroll
        for (int i = 2; i < 20; ++i) { BSX_EACH p[k] = __builtin_fma(t[k], p[k], C[i]); }
    } else {
        BSX_EACH p[k] = fma_k(t[k], 0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13);
#define BSX_HORNER(c) BSX_EACH p[k] = fma_k(t[k], p[k], c);
        BSX_HORNER(0x1.b2bb069efb384p-11) BSX_HORNER(-0x1.7952daf56de9bp-9) BSX_HORNER(0x1.d6d43a595c56fp-8) BSX_HORNER(-0x1.c6ea4a57d9582p-7)
        BSX_HORNER(0x1.67e295f08b19fp-6) BSX_HORNER(-0x1.e9ae6fc27006ap-6) BSX_HORNER(0x1.2c15b5711927ap-5) BSX_HORNER(-0x1.59976e82d3ff0p-5)
        BSX_HORNER(0x1.82d5d6ef28734p-5) BSX_HORNER(-0x1.ae5ce6a214619p-5) BSX_HORNER(0x1.e1bb48427b883p-5) BSX_HORNER(-0x1.110e48b207f05p-4)
        BSX_HORNER(0x1.3b13657b87036p-4) BSX_HORNER(-0x1.745d119378e4fp-4) BSX_HORNER(0x1.c71c717e1913cp-4) BSX_HORNER(-0x1.2492492376b7dp-3)
        BSX_HORNER(0x1.99999999952ccp-3) BSX_HORNER(-0x1.5555555555523p-2)
#undef BSX_HORNER
    }
    constexpr double PI_ = 0x1.921fb54442d18p+1, PI_2 = 0x1.921fb54442d18p+0;
    BSX_EACH {
        double a = __builtin_fma(q[k], t[k] * p[k], q[k]);
        a = ay[k] > ax[k] ? PI_2 - a : a;
        a = ix[k] < 0 ? PI_ - a : a;                                 // (the library's separate y == 0 case -- pi or 0 by the sign of x -- is what
        out[k] = iy[k] < 0 ? -a : a;                                 //  q = 0 gives here anyway); copysign(a, y): a >= 0, and iy = 0 keeps +a
    }
#undef BSX_EACH
}
__device__ inline double atan2_pixels(int iy, int ix) {
    const int ys[1] = {iy}, xs[1] = {ix};
    double o[1];
    atan2_pixels_n<1>(ys, xs, o);
    return o[0];
}
// rel_angle (battle_env.py:38-52), p0 = observer, p1 = target
__device__ inline double rel_angle(int x0, int y0, double a0, int x1, int y1) {
    double rads = atan2_pixels(y0 - y1, x0 - x1);
    rads = rads < 0.0 ? rads + TWO_PI : (rads == 0.0 ? 0.0 : rads);   // Python float %: fmod is exact for |rads| <= pi; -0.0 -> +0.0
    const double degs = rads * RAD2DEG;
    double r = (180.0 + a0) - (360.0 - degs);
    r = r < -180.0 ? r + 360.0 : r;
    r = r > 180.0 ? r - 360.0 : r;
    return r;
}
// rel_angle's second half: from rads (already reduced to [0, 2 pi)) to the wrapped difference with the observer's heading
__device__ inline double rel_from_rads(double rads, double a0) {
    const double degs = rads * RAD2DEG;
    double r = (180.0 + a0) - (360.0 - degs);
    if (r < -180.0) r += 360.0;
    if (r > 180.0) r -= 360.0;
    return r;
}
__device__ inline double pair_rads(int x0, int y0, int x1, int y1) {   // rel_angle's first half: atan2 % 2 pi, observer p0
    const double rads = atan2_pixels(y0 - y1, x0 - x1);
    return rads < 0.0 ? rads + TWO_PI : (rads == 0.0 ? 0.0 : rads);
}
// The two divisions by constants (battle_env.py:230-231) are multiplications by the float64 reciprocal here: the
// float64 result can differ in its last bit, which survives the single rounding to float32 with probability ~2^-29.
// sqrt of a squared pixel distance q = dx*dx + dy*dy (an integer below 2^22): the correctly rounded binary64 root, as math.sqrt
// gives it (battle_env.py:57).  Same iteration as the library sqrt -- reciprocal-root estimate, two coupled Newton steps on
// (g ~ sqrt x, h ~ 1 / (2 sqrt x)), two residual corrections with exact fma residuals -- without its range scaling and class tests,
// which an integer in [0, 2^22) never needs; q = 0 is returned as is (-0.9 % of the step against the library call).
__device__ inline double sqrt_pixels(int q) {
    const double x = double(q);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return q == 0 ? 0.0 : g;
}
__device__ inline float obs_dist(int x0, int y0, int x1, int y1) {
    const int dx = x0 - x1, dy = y0 - y1;
    return float(sqrt_pixels(dx * dx + dy * dy) * (2.0 / FIELD_DIAG) - 1.0);
}
// Range (obs_dist) and bearing (pair_rads) from (x, y) to K targets, the K evaluations in lockstep (see atan2_pixels_n).
template <int K>
__device__ inline void geometry_n(int x, int y, const int (&tx)[K], const int (&ty)[K], float (&d)[K], double (&rads)[K]) {
    int dx[K], dy[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { dx[k] = x - tx[k]; dy[k] = y - ty[k]; }
    double q[K], w[K], g[K], h[K], r[K], c[K];
#define BSX_EACH _Pragma("unroll") for (int k = 0; k < K; ++k)
    BSX_EACH q[k] = double(__mul24(dx[k], dx[k]) + __mul24(dy[k], dy[k]));           // |dx|, |dy| < 2^11
    BSX_EACH w[k] = __builtin_amdgcn_rsq(q[k]);
    BSX_EACH { g[k] = q[k] * w[k]; h[k] = w[k] * 0.5; }
    BSX_EACH r[k] = __builtin_fma(-h[k], g[k], 0.5);
    BSX_EACH { g[k] = __builtin_fma(g[k], r[k], g[k]); h[k] = __builtin_fma(h[k], r[k], h[k]); }
    BSX_EACH c[k] = __builtin_fma(-g[k], g[k], q[k]);
    BSX_EACH g[k] = __builtin_fma(c[k], h[k], g[k]);
    BSX_EACH c[k] = __builtin_fma(-g[k], g[k], q[k]);
    BSX_EACH g[k] = __builtin_fma(c[k], h[k], g[k]);
    BSX_EACH d[k] = float((q[k] == 0.0 ? 0.0 : g[k]) * (2.0 / FIELD_DIAG) - 1.0);
#undef BSX_EACH
    double a[K];
    atan2_pixels_n<K>(dy, dx, a);
#pragma unroll
    for (int k = 0; k < K; ++k) rads[k] = a[k] < 0.0 ? a[k] + TWO_PI : (a[k] == 0.0 ? 0.0 : a[k]);
}
__device__ inline float obs_angle(int x0, int y0, double a0, int x1, int y1) {
    return float(rel_angle(x0, y0, a0, x1, y1) * (1.0 / 360.0));
}
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
// reference's result.  So the list carries f for both axes (|d| <= 45) and, for the other case, a flag: such a shot also stores its
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

// np.argmax over four scores (battle_env.py:327-328): the first maximum; a NaN compares as the maximum.  The running maximum is a
// register, not v[arg]: a dynamically indexed local array lives in scratch memory.
__device__ inline int argmax4(float a, float b, float c, float d) {
    int am = 0;
    float best = a;
    const float v[3] = {b, c, d};
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (!(best != best) && (v[i] > best || v[i] != v[i])) { am = i + 1; best = v[i]; }
    return am;
}

struct StepArgs {
    StatePtrs st;
    int64_t E; int n;
    const void* actions; int action_kind;
    const double* u;
    float* obs; float* rew; uint8_t* done; uint8_t* env_done; uint8_t* winner;
    uint8_t* env_done_t;                                 // MULTI: nullable [T][E], env_done after every tick
    BsxRewards cfg;
    uint32_t flags; uint64_t seed; int64_t env_offset; int tie_tick;
    // multi-tick launches (bsx_step_many_*): T ticks, per-tick strides of the action / output arrays (0 = same array every tick)
    int T; int64_t act_tb /* bytes */, u_ts, obs_ts, rew_ts, done_ts /* elements */;
    // fused rollout (bsx_rollout_discrete): the actor in front of every tick
    const float* aw; int aprec; int scripted_team /* -1 none, 0 red, 1 blue */; const float* obs0; float* scores; int64_t scores_ts; BsxActorNoise nz; uint64_t aseed, aseq; const uint64_t* aseq_base;
    uint64_t iseed;                                      // continuous scripted opponent in the fused rollout: its Philox seed (bsx_instinct_continuous's `seed`)
};

// Observation row for one agent from the LDS-staged block (battle_env.py:202-244).
// s_* are indexed by thread id; `gl` = first thread of this env's group.
template <int N>
__device__ inline void write_obs(float* __restrict__ out, int n, bool alive, int x, int y, double dir, int a,
                                 int ebx, int eby, int gl, const volatile int* s_x, const volatile int* s_y,
                                 const volatile int* s_hp) {
    const int D = 3 * n + 2;
    if (!alive || (DIAG & 1u)) {
        for (int i = 0; i < D; ++i) out[i] = -1.0f;
        return;
    }
    out[0] = obs_dist(x, y, ebx, eby);
    out[1] = obs_angle(x, y, dir, ebx, eby);
    const int eb = gl + (a < n ? n : 0);
    for (int j = 0; j < n; ++j) {
        if (s_hp[eb + j] > 0) {
            const int qx = s_x[eb + j], qy = s_y[eb + j];
            out[2 + 3 * j] = 1.0f;
            out[3 + 3 * j] = obs_dist(x, y, qx, qy);
            out[4 + 3 * j] = obs_angle(x, y, dir, qx, qy);
        } else {
            out[2 + 3 * j] = -1.0f; out[3 + 3 * j] = -1.0f; out[4 + 3 * j] = -1.0f;
        }
    }
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

enum Mode : int { M_INERT = 0, M_TIE = 1, M_PHYS = 2, M_RESET = 3 };

// bsx_tie_tick(n) at compile time (battle_env.py:168,316-319: total_time += 0.1 in binary64 until >= 10 + 2n), for the kernels
// templated on n: one kernel argument fewer to fetch -- it was the one scalar load the compiler issued inside the branch that
// needs it, a fully exposed round trip of ~900 cycles (in-kernel stamps, r02am).
constexpr int tie_tick_const(int n) {
    const double max_time = double(10 + n * 2);
    double t = 0.0;
    int k = 0;
    do { t += 0.1; ++k; } while (!(t >= max_time));
    return k;
}
static_assert(tie_tick_const(1) == 121 && tie_tick_const(2) == 141 && tie_tick_const(3) == 161 && tie_tick_const(4) == 181 &&
              tie_tick_const(5) == 200, "time-limit tick");
static_assert(tie_tick_const(BSX_MAX_N) < 512, "the game clock fits the 9 bits of the game record");

// The scripted opponent's target choice and discrete action (instinct/agent.py:10-39,56-62) from one observation row,
// ob(k) = value k of the row: score every target by dist * |angle| (base first, strict '<' keeps the first minimum, a dead
// enemy scores 1e6), shoot inside 250 px and 20 degrees, else turn toward it.  binary64 on the float32 values, as the
// reference computes under its pinned numpy.  Also returns the chosen target's distance / angle (continuous branch).
template <class OB>
__device__ inline int instinct_choose(OB ob, int n, double& td, double& ta) {
    td = (double(ob(0)) + 1.0) / 2.0 * FIELD_DIAG;               // agent.py:15-16
    ta = double(ob(1)) * 360.0;
    double best = td * fabs(ta);
    for (int j = 0; j < n; ++j) {                                // agent.py:20-39
        const double d = (double(ob(3 + 3 * j)) + 1.0) / 2.0 * FIELD_DIAG, an = double(ob(4 + 3 * j)) * 360.0;
        const double sc = (ob(2 + 3 * j) == 1.0f) ? d * fabs(an) : 1000000.0;
        if (sc < best) { best = sc; td = d; ta = an; }
    }
    return (td < 250.0 && fabs(ta) < 20.0) ? 1 : (ta > 0.0 ? 3 : 2);   // agent.py:56-62
}
// The scripted opponent's continuous action (instinct/agent.py:41-54) for the chosen target at distance td / angle ta: shoot with
// probability 0.6 inside 2/3 of the shot distance and 20 degrees, speed from the distance, turn toward the target, uniform(-0.15,
// 0.15) noise on all three, clip.  Draws: row g of the launch, sequence number seq (bsx_instinct_continuous's keying).
__device__ inline void instinct_continuous_draws(uint64_t seed, uint64_t seq, uint64_t g, double& r0, double& n0, double& n1, double& n2) {
    const uint4 r = philox4x32_10(make_uint4(uint32_t(g), uint32_t(g >> 32) ^ 0x10000000u, uint32_t(seq), uint32_t(seq >> 32)),
                                  make_uint2(uint32_t(seed), uint32_t(seed >> 32)));
    r0 = double(r.x) * (1.0 / 4294967296.0);
    n0 = -0.15 + 0.3 * (double(r.y) * (1.0 / 4294967296.0));
    n1 = -0.15 + 0.3 * (double(r.z) * (1.0 / 4294967296.0));
    n2 = -0.15 + 0.3 * (double(r.w) * (1.0 / 4294967296.0));
}
__device__ inline void instinct_continuous_action(double td, double ta, double r0, double n0, double n1, double n2, double& o0, double& o1, double& o2) {
    double a2 = 0.0;
    if (td < 500.0 / 3.0 * 2.0 && fabs(ta) < 20.0) a2 = r0 < 0.6 ? 1.0 : -1.0;
    const double a0 = td / FIELD_DIAG * 2.0 - 1.0;
    const double a1 = ta > 0.0 ? fmax(-ta / 35.0, -1.0) : fmin(-ta / 35.0, 1.0);
    o0 = fmin(fmax(a0 + n0, -1.0), 1.0);
    o1 = fmin(fmax(a1 + n1, -1.0), 1.0);
    o2 = fmin(fmax(a2 + n2, -1.0), 1.0);
}
__device__ inline float4 one_hot_scores(int act) {               // what the score-vector step path arg-maxes back to `act`
    return make_float4(act == 0 ? 1.f : -1.f, act == 1 ? 1.f : -1.f, act == 2 ? 1.f : -1.f, act == 3 ? 1.f : -1.f);
}

// ---------------------------------------------------------------------------------------------- the step kernel
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

// Range / angle-off pair of one observer->target (battle_env.py:230-231,240-241)
__device__ inline void obs_pair(int x, int y, double dir, int tx, int ty, float& od, float& oa) {
    od = obs_dist(x, y, tx, ty);
    oa = obs_angle(x, y, dir, tx, ty);
}

// MULTI: the wave walks its games through p.T consecutive calls in one launch.  Games never leave their wave, so the
// only ordering needed between ticks is a lane's own stores before its own loads (program order through one L1: a
// wavefront-scope fence, no wait, no cache maintenance); the state stays in the L2 instead of crossing a kernel boundary
// (write-back + invalidate + a cold first round trip) every tick.
// ACTOR (discrete, MULTI, n <= 4): the caller's whole rollout loop `for t: actions = actor(obs); obs, rew, done = step(actions)`
// (main.py:177-181) in one launch.  The observation rows never leave the CU: the step leaves them in LDS, the actor
// (bsx_actor_core.h, MFMA) reads them there as its B operands.  An MFMA tile is 32 rows of ONE actor, so a workgroup is
// 32 games = G/2 waves (1v1: one wave, 2v2: two, 3v3 / 4v4: four) and holds one tile per plane id; wave w runs the tiles of
// planes w and w + G/2 (tile 0 in its lower lane half's name, tile 1 in the upper's), lane l finishes row (game l & 31 of the
// workgroup, that plane), and the arg-max travels back to the plane's own lane through LDS (one cross-lane move for 1v1).
// Everything else stays private to a wave exactly as in the other variants: a wave still only touches its own games.
// LG (discrete only): the actions are float32 [4] score vectors (arg-maxed here) instead of int32 indices -- a compile-time switch, so
// that each encoding's kernel issues exactly its own action load in the first batch (an unconditional load of the unused encoding's
// dummy line cost 1.6 % of the step; a load under a branch costs a second round trip, see load_inputs).
// Row and byte offsets of the step kernel come in two widths (template parameter OFF32).  A job whose largest array stays below
// 4 GB -- every measured configuration; 200 bytes per agent (the widest observation rows) are the bound, so up to 21 M agents -- addresses
// memory as SGPR base + 32-bit VGPR byte offset: one shift or 24-bit multiply-add per dependent access where 64-bit offsets take two
// 64 x 32 multiply-adds, two moves and a 64-bit shift-add (C2 7.33 -> 7.22 us, bullet-heavy 14.96 -> 14.73).  Larger jobs (2^30 games
// are allowed) and BSX_F_WIDE_OFFSETS take the 64-bit kernels.
template <class T> __device__ inline T* elem(T* base, uint32_t i) {
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    return reinterpret_cast<T*>(reinterpret_cast<byte_t*>(base) + uint32_t(i * uint32_t(sizeof(T))));
}
template <class T> __device__ inline T* elem(T* base, size_t i) { return base + i; }
template <int N, bool CONT, bool MULTI, bool ACTOR = false, bool LG = false, bool OFF32 = false>
__global__ __launch_bounds__(SPB * (ACTOR ? group_width(N > 0 ? N : 1) / 2 : WPB)) __attribute__((amdgpu_waves_per_eu((ACTOR && N > 1) ? 2 : ((!ACTOR && !MULTI && N >= 2) ? X_MIN_WAVES : 1))))
void bsx_step_kernel(const int64_t E_, const uint2* const envc_, const uint2* const envd_, const uint2* const plane_, const void* const act_,
                     const uint2* const bent_, const uint32_t* const bcnt_, const int kind_, const StepArgs p_) {
    const StepArgs& p = p_;                              // (the tick loop of the multi-tick forms shadows this name: see there)
    // The eight leading arguments repeat p.E, p.st.envc, p.st.envd, p.st.plane, p.actions, p.st.bent, p.st.bcnt, p.action_kind: fifteen
    // dwords that the dispatcher preloads into SGPRs (-amdgpu-kernarg-preload-count), so that a wave's first loads need nothing
    // from the kernarg segment and do not queue behind its cold scalar-cache fetch.
    STAMP(8);                                            // diagnostic builds: kernel entry, before any kernarg load
    typedef typename std::conditional<OFF32, uint32_t, size_t>::type ix_t;     // row / element offsets
    typedef typename std::conditional<OFF32, int32_t, int64_t>::type ixs_t;    // game indices
    constexpr bool NT_STATE = !MULTI && N >= 2;
    constexpr int WAVES = ACTOR ? group_width(N > 0 ? N : 1) / 2 : WPB;
    const int n = (N > 0) ? N : p.n;
    const int A = 2 * n;
    const int G = group_width(n);
    const int EPB = SPB / G;
    const int wave = (WAVES > 1) ? int(threadIdx.x >> 6) : 0;
    const int tid = (WAVES > 1) ? int(threadIdx.x & 63) : int(threadIdx.x);   // position in my wave = LDS index in its private arrays
    const ixs_t wblk = (WAVES > 1) ? ixs_t(blockIdx.x) * WAVES + wave : ixs_t(blockIdx.x);   // which 64 lanes of the job I am
    const int a = tid & (G - 1);
    const ixs_t e = wblk * EPB + (tid / G);
    const bool env_ok = e < ixs_t(E_);
    const bool valid = env_ok && a < A;
    // E_ = the games THIS launch steps (rows 0 .. E_ - 1 of every array it was handed); p.E = the games the state was laid out for, i.e.
    // the row stride of the entry-major bullet arrays.  They differ only for a launch over a sub-range of the games (bsx_step_*_range:
    // every [E]-major pointer arrives advanced to the range's first game, the entry-major ones by the same rows within their first entry).
    const ix_t EA = ix_t((MULTI || ACTOR) ? E_ : p.E) * ix_t(A);
    // out-of-range lanes read a valid row (the last one) and never store: loads stay unconditional
    const ixs_t ec = env_ok ? e : ixs_t(E_ - 1);
    const ix_t g = ix_t(ec) * A + (a < A ? a : A - 1);
    constexpr int NE = (N > 0) ? N : 1;                  // compile-time enemy count (runtime-n build reads LDS in loops)

    // LDS is private to this wavefront: accesses are volatile (program order) and the hardware runs one wave's LDS
    // operations in order, so cross-lane hand-offs need no s_barrier -- only a compiler scheduling fence.
    constexpr int DROW = (N > 0) ? 3 * N + 2 : 3 * BSX_MAX_N + 2;
    __shared__ volatile int s_x_all[WAVES * SPB], s_y_all[WAVES * SPB], s_hp_all[WAVES * SPB];
    __shared__ volatile int s_bhit_all[WAVES * SPB];     // base hits, index gl + shooter team
    __shared__ __attribute__((aligned(16))) float s_obs_all[WAVES * SPB * DROW];   // observation rows, [wave][lane][D]
    __shared__ __attribute__((aligned(16))) float s_small[ACTOR ? (N == 1 ? 4 : 2 * N) * bsx_actor::SMALL : 4];   // per-neuron vectors + heads of the actors (1v1: + the two value heads')
    __shared__ int s_act_all[(ACTOR && !CONT && WAVES > 1) ? WAVES * SPB : 1];       // arg-max per row, ACTOR with several waves
    __shared__ float s_actf_all[(ACTOR && CONT && WAVES > 1) ? WAVES * SPB * 3 : 1];  // continuous: [speed, turn, shoot] per row
    __shared__ double s_actd_all[(ACTOR && CONT && WAVES > 1) ? WAVES * SPB * 3 : 1]; // ... of a scripted team's rows, binary64
    __shared__ int s_gdone_all[(ACTOR && WAVES > 1) ? 32 : 1];                       // game-over flag per game of the workgroup
    // n >= 2: every plane-to-plane pair is computed ONCE, by one of its two planes, and handed to the other through these
    __shared__ float s_pd_all[(N >= 2) ? WAVES * SPB * N : 1];      // range (symmetric)
    __shared__ double s_pr_all[(N >= 2) ? WAVES * SPB * N : 1];     // the owner's bearing in radians, [0, 2 pi)
    // wave-packed bullet pass: the wave's pool entries (and this call's shots behind them) are WORK SLOTS, one per lane and round
    __shared__ __attribute__((aligned(8))) u32x2 s_new_all[WAVES * SPB];   // this call's shots as pool entries (age 0, the PRE-move pose), by shot rank
    __shared__ uint32_t s_agg_all[WAVES * SPB];          // per owner: misses << 16 | base hits << 24
    // The rectangles a bullet is tested against (enemy base, enemy planes' sprites), staged per owner / per plane for the work slots.
    // 1v1: as (lower corner, upper corner) pairs of packed (x, y) halves BIASED by +64, so that no half is ever negative and the
    // corners are plain 32-bit adds of packed literals: a bullet at b overlaps <=> no half of (b - lower) | (upper - b) is negative
    // (C2 7.33 -> 7.20 us against the centre form: ~15 instructions fewer per round where the instruction count is the bound).
    // Larger teams: the centre (x | alive << 15 | y << 16) and the margins as constants in the slot -- measured faster there
    // (4v4 23.1 us against 24.7 with corners; the same launches, four runs each).
    constexpr bool CORNERS = N == 1 || X_CORNERS_ALL;
    typedef typename std::conditional<CORNERS, u32x2, uint32_t>::type rect_t;
    __shared__ __attribute__((aligned(8))) rect_t s_eb_all[WAVES * SPB];        // per owner: the enemy base, dx in [-33, 33], dy in [-32, 31]
    __shared__ __attribute__((aligned(8))) rect_t s_pq_all[WAVES * SPB];        // per plane: its post-move sprite, dx in [-27, 27], dy in [-25, 24]; dead: never hit
    // rect(centre, alive, margins below / above): what the owner side stages
    auto make_rect = [](uint32_t c, bool alive, int xl, int yl, int xh, int yh) {
        if constexpr (CORNERS) return alive ? u32x2{c + pk_const(PK_BIAS - xl, PK_BIAS - yl), c + pk_const(PK_BIAS + xh, PK_BIAS + yh)} : u32x2{0x7F007F00u, 0u};
        else return c | (alive ? 0x8000u : 0u);
    };
    // 0 / -1: does the bullet at b (CORNERS: biased) overlap the rectangle?
    auto hits_rect = [](s16x2 b, rect_t r, int xl, int yl, int xh, int yh) {
        if constexpr (CORNERS) return ~pk_any_negative(pk_bits(b - as_pk(r.x)) | pk_bits(as_pk(r.y) - b));
        else {
            const s16x2 d = b - as_pk(r & ENT_XY);
            return ~pk_any_negative(pk_bits(d + as_pk(pk_const(xl, yl))) | pk_bits(as_pk(pk_const(xh, yh)) - d)) & (int(r << 16) >> 31);
        }
    };
    // per owner: what a work slot must know about its bullet's owner: the owner's tick % 12 (the exact-path ring) | 16: the owner's game
    // is in its physics call (its bullets fly) | 32: the game is being re-spawned by this call (its bullets are dropped)
    constexpr uint32_t OWN_PHYS = 16u, OWN_DROP = 32u;
    __shared__ uint32_t s_fl_all[WAVES * SPB];
    __shared__ __attribute__((aligned(16))) double s_nd_all[WAVES * SPB * 2];   // per owner: this call's shot's float64 step (written and read on the exact path only)
    // plane-overlap candidates per owner, by AGE (rare): FW bits per age (which enemy planes the bullet of that age overlaps), and where
    // that bullet's entry now sits in the pool (for the tombstone of a consumed bullet)
    constexpr int FW = (N > 0 && N <= 4) ? 4 : 16;       // bits per overlap field
    constexpr int OW = (FW == 4) ? 1 : 3;                // 64-bit words holding the 12 fields
    __shared__ unsigned long long s_ov_all[WAVES * SPB * OW];
    __shared__ uint16_t s_pp_all[WAVES * SPB * K];
    auto* const s_new = BSX_LDS(u32x2, s_new_all) + wave * SPB;
    auto* const s_agg = BSX_LDS(uint32_t, s_agg_all) + wave * SPB;
    auto* const s_eb = BSX_LDS(rect_t, s_eb_all) + wave * SPB;
    auto* const s_pq = BSX_LDS(rect_t, s_pq_all) + wave * SPB;
    auto* const s_fl = BSX_LDS(uint32_t, s_fl_all) + wave * SPB;
    auto* const s_nd = BSX_LDS(double, s_nd_all) + wave * SPB * 2;
    auto* const s_ov = BSX_LDS(unsigned long long, s_ov_all) + wave * SPB * OW;
    auto* const s_pp = BSX_LDS(uint16_t, s_pp_all) + wave * SPB * K;
#pragma unroll
    for (int q = 0; q < OW; ++q) s_ov[tid * OW + q] = 0ull;   // cleared again by whoever finds them set
    float* const s_pd = s_pd_all + ((N >= 2) ? wave * SPB * N : 0);
    double* const s_pr = s_pr_all + ((N >= 2) ? wave * SPB * N : 0);
    // (explicit LDS address space: a volatile access through a generic pointer compiles to flat_load / flat_store)
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint* const s_x = (lds_vint*)(uintptr_t)(s_x_all) + wave * SPB;
    lds_vint* const s_y = (lds_vint*)(uintptr_t)(s_y_all) + wave * SPB;
    lds_vint* const s_hp = (lds_vint*)(uintptr_t)(s_hp_all) + wave * SPB;
    lds_vint* const s_bhit = (lds_vint*)(uintptr_t)(s_bhit_all) + wave * SPB;
    float* const s_obs = s_obs_all + wave * SPB * DROW;

    const bool has_act = kind_ >= 0;                     // an empty call (step({})) comes with action_kind -1 and a dummy, mapped action pointer
    // Raw inputs of one call (decoded at the top of the tick that uses them).
    struct RawIn { int ai; float4 lg; float f0, f1, f2; double c0, c1, c2, uu; };
    auto load_inputs = [&](int t, RawIn& r) {
        const void* const at = MULTI ? static_cast<const void*>(static_cast<const char*>(act_) + int64_t(t) * p.act_tb) : act_;
        const double* const ut = (MULTI && p.u) ? p.u + int64_t(t) * p.u_ts : p.u;
        if (!CONT) {
            // one unconditional load (a load under a branch makes the compiler's wait-count pass drain EVERYTHING in flight before the
            // other branch's load: the action then cost a second full round trip); an empty call reads a mapped dummy line
            // (the host passes a mapped address for it -- the state block -- and action_kind -1: no pointer select in front of the first loads)
            const char* const abase = static_cast<const char*>(at);
            if constexpr (LG) r.lg = *reinterpret_cast<const float4*>(elem(abase, has_act ? g * 16 : ix_t(0)));
            else r.ai = *reinterpret_cast<const int32_t*>(elem(abase, has_act ? g * 4 : ix_t(0)));
        } else if (has_act) {                            // uniform branch
            if (kind_ == BSX_ACT_F32) {
                const float* ap = static_cast<const float*>(at) + 3 * g;
                r.f0 = ap[0]; r.f1 = ap[1]; r.f2 = ap[2];
            } else if (kind_ == BSX_ACT_F32X4) {
                const float4 v = static_cast<const float4*>(at)[g];
                r.f0 = v.x; r.f1 = v.y; r.f2 = v.z;
            } else {
                const double* ap = static_cast<const double*>(at) + 3 * g;
                r.c0 = ap[0]; r.c1 = ap[1]; r.c2 = ap[2];
            }
        }
        // (the loads above need nothing but preloaded kernel arguments: they must be in flight BEFORE anything waits for the
        //  kernarg segment's scalar fetch -- p.u is the first thing that does)
        __builtin_amdgcn_sched_barrier(0);
        if (ut) r.uu = ut[g];                            // uniform branch
    };
    // MULTI: what one call hands to the next stays in REGISTERS -- my plane, my game's record and episode count (every
    // lane of a game computes the same record) -- so a later tick starts with its bullet loads instead of a state round
    // trip, and the inputs of tick t+1 are fetched while tick t computes.  Bullet lists and counters go to memory every
    // tick, the plane and game records once, after the last one.
    int x = 0, y = 0, hp = 0;
    uint32_t games = 0;                                  // games my slot has finished = episode number of the random streams (travels in the game record)
    double dir = 0.0;
    EnvU er = {};
    // my wave block's bullet pool: `pc` entries at bent[pool0 ...] (wave-uniform); the first 64 entries are requested with the first
    // batch of loads, whatever pc is (a mapped, aligned 512-byte row: fully coalesced, and no load of the step depends on another)
    const ix_t pool0 = ix_t(wblk) * ix_t(POOL_CAP);
    uint32_t pc = 0;
    uint2 pool_first = make_uint2(0u, 0u);
    RawIn rin = {}, rin_next = {};
    struct DecIn { int act; double a0, a1, a2, uu; };    // a call's inputs, decoded
    auto decode = [&](const RawIn& r) {
        DecIn d = {-1, 0.0, 0.0, 0.0, 0.0};
        if (has_act) {                                   // uniform branch
            if (!CONT) {
                if constexpr (!LG) d.act = r.ai;
                else d.act = argmax4(r.lg.x, r.lg.y, r.lg.z, r.lg.w);
            } else if (kind_ == BSX_ACT_F32 || kind_ == BSX_ACT_F32X4) {
                d.a0 = double(r.f0); d.a1 = double(r.f1); d.a2 = double(r.f2);
            } else {
                d.a0 = r.c0; d.a1 = r.c1; d.a2 = r.c2;
            }
        }
        if (p.u) d.uu = r.uu;                            // uniform branch
        return d;
    };
    // MULTI: the inputs of the NEXT tick are fetched while this one computes and decoded BEFORE this tick's stores go out:
    // vmcnt is in-order and shared by loads and stores, so decoding at the top of the next tick would wait for all of them.
    DecIn din_next = {-1, 0.0, 0.0, 0.0, 0.0};
    if (MULTI && !ACTOR) { load_inputs(0, rin_next); din_next = decode(rin_next); }
    if constexpr (ACTOR) {
        constexpr int D = 3 * N + 2;
        for (int i = int(threadIdx.x); i < 2 * N * bsx_actor::SMALL / 4; i += SPB * WAVES) {
            const int ag = i / (bsx_actor::SMALL / 4), j = i - ag * (bsx_actor::SMALL / 4);
            reinterpret_cast<float4*>(s_small)[i] =
                reinterpret_cast<const float4*>(p.aw + size_t(ag) * bsx_actor::blob_floats(D) + bsx_actor::off_small(D))[j];
        }
        if constexpr (N == 1) {
            if (p.nz.value_weights)
                for (int i = int(threadIdx.x); i < 2 * bsx_actor::SMALL / 4; i += SPB * WAVES) {
                    const int ag = i / (bsx_actor::SMALL / 4), j = i - ag * (bsx_actor::SMALL / 4);
                    reinterpret_cast<float4*>(s_small + 2 * bsx_actor::SMALL)[i] =
                        reinterpret_cast<const float4*>(p.nz.value_weights + size_t(ag) * bsx_actor::blob_floats(D) + bsx_actor::off_small(D))[j];
                }
        }
        // the observations the rollout starts from (obs[0]): this wave's rows are one contiguous block
        const int64_t e_first = wblk * EPB;
        const int64_t nfl = min(int64_t(SPB), (E_ - e_first) * A) * D;
        if (G == A) {
            for (int i = tid; i < SPB * D; i += SPB) s_obs[i] = i < nfl ? p.obs0[size_t(e_first) * A * D + i] : -1.0f;
        } else {                                         // 3v3: lanes 6, 7 of a group own no row
            for (int k = 0; k < D; ++k) s_obs[tid * D + k] = valid ? p.obs0[g * D + k] : -1.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (WAVES > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    if constexpr ((ACTOR || (X_DEPHASE & 512)) && X_DEPHASE != 0) {   // variant builds only: half of the workgroups start late (do the waves of a SIMD fall into anti-phase?)
        if ((X_DEPHASE & 256) ? (blockIdx.x & 1u) : (blockIdx.x >= gridDim.x / 2))
            for (int i = 0; i < (X_DEPHASE & 63); ++i) __builtin_amdgcn_s_sleep((X_DEPHASE & 1024) ? 31 : 127);
    }
    for (int tk = 0; tk < (MULTI ? p.T : 1); ++tk) {
    // In the tick loop the compiler would hoist everything loop-invariant -- 36 row addresses, the Philox key schedule,
    // every fp64 constant -- and run out of registers (256 VGPRs, 1-2 waves per SIMD, SGPR spills).  Passing the three
    // values all of that hangs on through an empty asm makes it per-tick work again, as in the one-call kernel.
    ix_t gt = g, EAt = EA;
    uint64_t seed_t = p.seed;
    int64_t env_offset_t = p.env_offset;
    constexpr int TIE_C = tie_tick_const(N > 0 ? N : 1);
    int tie_tick = (N > 0) ? TIE_C : p.tie_tick;
    if (MULTI) {
        asm volatile("" : "+v"(gt));
        asm volatile("" : "+s"(EAt));
        asm volatile("" : "+s"(seed_t));
    }
    // The fused rollout of teams >= 2 runs at 256 registers: there the lane's indices pass through an empty asm per tick as well, so
    // that the ~35 LDS / row addresses derived from them (pair slots, enemy lanes, staging rows) are per-tick work next to their use
    // instead of registers held across the actor's matrix products -- with them hoisted the kernels spilled to scratch memory.
    int tid_k = tid;
    if constexpr (ACTOR && N > 1) asm volatile("" : "+v"(tid_k));
    // The kernel's arguments likewise: ~60 scalar registers of pointers, strides and reward constants were held across the tick
    // loop, ~40 of them spilled to VGPR lanes before it and read back one v_readlane at a time in every tick (78 of them at
    // 1v1).  Inside a tick the arguments are read through the kernarg segment's own address, made opaque per tick: scalar loads
    // of 4 ... 16 dwords next to their use, nothing carried.  (StepArgs follows eight leading arguments: 7 x 8 + 4 bytes, padded to 64.)
    typedef const StepArgs __attribute__((address_space(4))) StepArgsK;
    static_assert(alignof(StepArgs) == 8, "kernarg offset of StepArgs");
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    if (MULTI) asm volatile("" : "+s"(ka));
    // (the one-call kernels keep the parameter itself: their argument fetch is placed by hand in the shadow of the first loads, and
    //  through the segment pointer it measured slower -- C2 7.21 -> 7.37 us, 4v4 23.4 -> 27.3)
    auto& p = [&]() -> decltype(auto) { if constexpr (MULTI) return (*reinterpret_cast<StepArgsK*>(ka + 64)); else return (p_); }();
    const int tid = tid_k, lane = tid, a = tid & (G - 1);
    const int gl = tid & ~(G - 1);                       // first thread of my env's group
    const int team = (a < n) ? 0 : 1;                    // 0 red, 1 blue
    const int eb = gl + (team == 0 ? n : 0);             // first enemy lane (thread index)
    const double* const u_t = (MULTI && p.u) ? p.u + int64_t(tk) * p.u_ts : p.u;
    float* const obs_t = MULTI ? p.obs + int64_t(tk) * p.obs_ts : p.obs;
    float* const rew_t = MULTI ? p.rew + int64_t(tk) * p.rew_ts : p.rew;
    uint8_t* const done_t = MULTI ? p.done + int64_t(tk) * p.done_ts : p.done;
    STAMP(0);
    // ================= T0: every load of the step, issued back to back as raw words: none depends on another ========
    // (the kernel is latency-bound at 65 536 games -- 2 waves per SIMD -- so memory-level parallelism is what pays)
    if (!MULTI || tk == 0) {
        const uint2 ecw = *elem(envc_, ix_t(ec));
        const uint2 edw = *elem(envd_, ix_t(ec));         // .y = games finished so far = episode id of the RNG streams
        const uint2 prw = *elem(plane_, gt);
        double dirf = 0.0;
        if constexpr (CONT) dirf = *elem(p.st.pdirf, gt);
        if (!MULTI) load_inputs(0, rin);
        pool_first = *elem(bent_, pool0 + ix_t(lane));
        if (!(DIAG & 2u)) pc = __builtin_amdgcn_readfirstlane(*elem(bcnt_, ix_t(wblk)));
        if (!MULTI) {
            // every kernel argument the step needs later is fetched HERE, in the shadow of the first vector loads: left to
            // the compiler, the ones first used inside a branch are loaded there -- a cold scalar fetch with nothing to hide it
            asm volatile("" : "+s"(seed_t), "+s"(env_offset_t));
            if (N == 0) asm volatile("" : "+s"(tie_tick));
        }
        unpack_plane(prw, x, y, hp, dir);
        if constexpr (CONT) dir = (prw.y & PLANE_FRAC) ? dirf : dir;
        er = unpack_env(ecw, edw.x);
        games = edw.y;
    } else {
        pool_first = *elem(bent_, pool0 + ix_t(lane));   // this tick's first 64 entries (the last tick's stores precede this load in program order)
    }
    const DecIn din = MULTI ? din_next : decode(rin);
    int act = din.act;
    double a0 = din.a0, a1 = din.a1, a2 = din.a2, uu_in = din.uu;
    if constexpr (ACTOR) {
        // ---- actions = argmax(actor(obs)) (maddpg/agent.py:25-33, battle_env.py:327-328), rows straight from LDS
        constexpr int D = 3 * N + 2, A_ = 2 * N, G_ = group_width(N);
        if (WAVES > 1) {                                 // every wave's rows (written at the end of the last tick) and game flags
            if (a == 0) s_gdone_all[wave * (SPB / G_) + tid / G_] = er.done;
            __syncthreads();
        }
        const int hh = lane >> 5, c = lane & 31;         // I finish row (game c of the workgroup, plane id `mine`)
        const int mine = wave + hh * WAVES;
        const bool has_row = mine < A_;                  // 3v3: waves 2 and 3 have one tile only
        const int mine_c = has_row ? mine : A_ - 1;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma nounroll
        for (int ti = 0; ti < 2; ++ti) {                 // one tile at a time: its 64 weight registers are reused by the next
            const int ag = wave + ti * WAVES;            // wave-uniform
            if (ag >= A_ || p.scripted_team == (ag >= N ? 1 : 0)) continue;   // no such plane / played by the scripted opponent
            const float* const Wn = p.aw + size_t(ag) * bsx_actor::blob_floats(D);
            const float* const smn = s_small + ag * bsx_actor::SMALL;
            auto xb = [&](int k) { return k < D ? s_obs_all[(c * G_ + ag) * D + k] : 0.f; };
            float4 o;                                    // uniform branches
            constexpr bool ROLL = N > 1;             // teams >= 2 carry more state across the actor: the 64 x 64 layer's weights as a rolling window
            if (p.aprec == BSX_ACTOR_BF16X3) o = bsx_actor::tile_forward<BSX_ACTOR_BF16X3, ROLL>(Wn, smn, D, lane, xb);
            else if (p.aprec == BSX_ACTOR_BF16X6) o = bsx_actor::tile_forward<BSX_ACTOR_BF16X6, ROLL>(Wn, smn, D, lane, xb);   // (teams >= 2: its 96 weight registers fit as a rolling window of 72)
            else o = bsx_actor::tile_forward<BSX_ACTOR_F32, ROLL>(Wn, smn, D, lane, xb);
            if (hh == ti) r4 = o;                        // lower half finishes the wave's first tile, upper half the second
        }
        const float4 b3 = *reinterpret_cast<const float4*>(s_small + mine_c * bsx_actor::SMALL + 6 * bsx_actor::H + bsx_actor::H * bsx_actor::NA);
        const int64_t er_ = int64_t(blockIdx.x) * 32 + c;
        const bool row_ok = has_row && er_ < E_;
        const size_t row = size_t(er_ < E_ ? er_ : E_ - 1) * A + mine_c;
        const uint64_t aseq = p.aseq + (p.aseq_base ? *p.aseq_base : 0ull) + uint64_t(tk);
        bool game_over;
        if (WAVES > 1) game_over = s_gdone_all[c] != 0;
        else game_over = __shfl(er.done, 2 * c) != 0;
        if (p.nz.ou_keep) game_over = false;             // the evaluation loop never restarts its noise process (evaluate.py:52-76)
        const bool scripted_row = p.scripted_team == (mine_c >= N ? 1 : 0);
        double sd0 = 0.0, sd1 = 0.0, sd2 = 0.0;          // continuous: the scripted row's binary64 actions, as bsx_instinct_continuous writes them
        if (scripted_row) {                              // instinct/team.py:13-15 for this team's rows
            double td_, ta_;
            const int sact = instinct_choose([&](int k) { return s_obs_all[(c * G_ + mine_c) * D + k]; }, N, td_, ta_);
            if constexpr (!CONT) r4 = one_hot_scores(sact);                     // ... as one-hot score rows
            else {
                double r0, n0, n1, n2;
                instinct_continuous_draws(p.iseed, aseq, uint64_t(row), r0, n0, n1, n2);
                instinct_continuous_action(td_, ta_, r0, n0, n1, n2, sd0, sd1, sd2);
                r4 = make_float4(float(sd0), float(sd1), float(sd2), 0.f);      // the record holds them rounded to float32; the step takes the binary64 values
            }
        } else {
            r4 = bsx_actor::finish_row(r4, b3, p.nz, p.aseed, aseq, row, uint64_t(p.env_offset) * uint64_t(A) + row, game_over, row_ok,
                                       size_t(tk) * size_t(E_) * size_t(A) + row);
        }
        if constexpr (N == 1) {
            if (p.nz.value_weights) {
                // ---- the value head (1v1): a second MLP of the actor's shape on the same LDS rows, in the actors' precision mode; its
                //      per-neuron vectors and head sit behind the actors' in LDS
                float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma nounroll
                for (int ti = 0; ti < 2; ++ti) {
                    const float* const Wn = p.nz.value_weights + size_t(ti) * bsx_actor::blob_floats(D);
                    const float* const smn = s_small + (2 + ti) * bsx_actor::SMALL;
                    auto xb = [&](int k) { return k < D ? s_obs_all[(c * G_ + ti) * D + k] : 0.f; };
                    float4 o;                            // the 64 x 64 layer in the actors' precision mode (uniform branches)
                    if (p.aprec == BSX_ACTOR_BF16X3) o = bsx_actor::tile_forward<BSX_ACTOR_BF16X3>(Wn, smn, D, lane, xb);
                    else if (p.aprec == BSX_ACTOR_BF16X6) o = bsx_actor::tile_forward<BSX_ACTOR_BF16X6>(Wn, smn, D, lane, xb);
                    else o = bsx_actor::tile_forward<BSX_ACTOR_F32>(Wn, smn, D, lane, xb);
                    if (hh == ti) v4 = o;
                }
                if (row_ok) p.nz.value[size_t(tk) * size_t(E_) * size_t(A) + row] = v4.x + s_small[(2 + mine_c) * bsx_actor::SMALL + 6 * bsx_actor::H + bsx_actor::H * bsx_actor::NA];
            }
        }
        if (row_ok) reinterpret_cast<float4*>(p.scores + int64_t(tk) * p.scores_ts)[row] = r4;
        if constexpr (!CONT) {
            const int am = argmax4(r4.x, r4.y, r4.z, r4.w);
            if (WAVES > 1) {                             // plane (game, id) sits in lane game*G + id of the workgroup
                if (has_row) s_act_all[c * G_ + mine] = am;
                __syncthreads();
                act = s_act_all[wave * SPB + tid];
            } else {
                act = __shfl(am, ((lane & 1) << 5) | (lane >> 1));      // plane (game L>>1, agent L&1) <- lane 32*(L&1) + (L>>1)
            }
        } else {
            // [speed, turn, shoot] of the row, as bsx_step_continuous reads a BSX_ACT_F32X4 row: float32 -> binary64
            float f0, f1, f2;
            if (WAVES > 1) {
                if (has_row) { float* q = &s_actf_all[(c * G_ + mine) * 3]; q[0] = r4.x; q[1] = r4.y; q[2] = r4.z; }
                __syncthreads();
                const float* q = &s_actf_all[(wave * SPB + tid) * 3];
                f0 = q[0]; f1 = q[1]; f2 = q[2];
            } else {
                const int src = ((lane & 1) << 5) | (lane >> 1);
                f0 = __shfl(r4.x, src); f1 = __shfl(r4.y, src); f2 = __shfl(r4.z, src);
            }
            a0 = double(f0); a1 = double(f1); a2 = double(f2);
            if (p.scripted_team >= 0) {                  // uniform: the scripted planes' binary64 actions travel the same way, unrounded
                double d0_, d1_, d2_;
                if (WAVES > 1) {
                    __syncthreads();
                    if (has_row) { double* q = &s_actd_all[(c * G_ + mine) * 3]; q[0] = sd0; q[1] = sd1; q[2] = sd2; }
                    __syncthreads();
                    const double* q = &s_actd_all[(wave * SPB + tid) * 3];
                    d0_ = q[0]; d1_ = q[1]; d2_ = q[2];
                } else {
                    const int src = ((lane & 1) << 5) | (lane >> 1);
                    d0_ = __shfl(sd0, src); d1_ = __shfl(sd1, src); d2_ = __shfl(sd2, src);
                }
                if (team == p.scripted_team) { a0 = d0_; a1 = d1_; a2 = d2_; }
            }
        }
    }

    // ================= T1: the heading-table entry -- the one dependent load of the common path =====================
    // The heading table (361 x 16 B, read by every wave of every launch) stays hot in each CU's L1: the entry for the
    // post-rotation heading is gathered as soon as the action is known, so the plane can move while the shot is prepared.
    double dir_rot = dir;
    if (!CONT) dir_rot = rotate_dir(dir, act == 2 ? 15.0 : (act == 3 ? -15.0 : 0.0));   // one straight-line rotate (+0 leaves any heading in [0, 360] as it is)
    double2 dl = make_double2(0.0, 0.0);
    if (!CONT) dl = p.st.lut[min(max(int(dir_rot), 0), 360)];   // 21.5*cos(-radians(d)), 21.5*sin(-radians(d)) from host libm
    if (MULTI && !ACTOR && tk + 1 < p.T) load_inputs(tk + 1, rin_next);   // behind this tick's own loads: nothing waits for it before the tick ends
    const bool alive0 = valid && hp > 0;
    STAMP(1);

    // "no agents left" (battle_env.py:309): group ballot over the alive flags
    const unsigned long long bal = __ballot(alive0);
    const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (lane & ~(G - 1)));
    const bool any_alive = (bal & gmask) != 0ull;

    // ---- what kind of call is this for my env (battle_env.py:303-323)
    int mode;
    int tick = er.tick;
    if (er.done) mode = (p.flags & BSX_F_AUTO_RESET) ? M_RESET : M_INERT;
    else if ((p.flags & BSX_F_EMPTY_CALL) || !any_alive) mode = M_TIE;
    else {
        tick += 1;
        mode = (tick >= tie_tick) ? M_TIE : M_PHYS;
    }
    if (!env_ok) mode = M_INERT;

    int4 cnt_delta = make_int4(0, 0, 0, 0);              // games, ties, red wins, blue wins
    const double d0 = dir;
    const int64_t genv = env_offset_t + ec;
    // does this call fire? (battle_env.py:404-406 / :423; the shot leaves from the PRE-move pose, so it is prepared first:
    // its Philox draw and sincos run while the heading-table entry of the move below is still on its way from the L2)
    if (CONT) a2 = fmin(fmax(a2, -1.0), 1.0);
    bool spawn = (mode == M_PHYS) && alive0 && !(DIAG & 2u) && (CONT ? (a2 > 0.0) : (act == 1));
    // ---- Bullet.__init__ (sprites.py:293-318) for this call's shot: heading = pre-move heading + (u*8 - 4)
    const bool phys = (mode == M_PHYS) && valid && !(DIAG & 2u);
    const int ks = tick % K;                             // birth-tick ring slot of this call's shot (heading, export only)
    // ---- wave-packed bullet pass, part 1.  The wave's bullets ARE a packed array -- its pool, pc entries in memory -- and this call's
    // shots queue up behind them in LDS by shot rank: slot w < pc is pool entry w, slot pc + r the r-th shooter's new bullet; slot w is
    // served by lane w % 64 in round w / 64.  Under uniform play a plane holds 0.6 bullets and fires every fourth call: ~37 + 16 slots,
    // ONE round.  What a slot needs from its bullet's owner (named by the entry) is staged per owner lane in LDS.
    FSTAMP(3);
    const unsigned long long shb = __ballot(spawn);
    const int srank = int(__builtin_amdgcn_mbcnt_hi(uint32_t(shb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(shb), 0u)));
    const int slots = int(pc) + __popcll(shb);           // wave-uniform
    s_agg[tid] = 0u;
    s_eb[tid] = make_rect(pack_xy(team == 0 ? er.bbx : er.brx, team == 0 ? er.bby : er.bry), true, 33, 32, 33, 31);
    s_fl[tid] = uint32_t(ks) | (phys ? OWN_PHYS : 0u) | ((mode == M_RESET && valid) ? OWN_DROP : 0u);
    // does this call touch the pool at all?  (not if no game of the wave is in its physics call or being re-spawned: entries stay as they are)
    const bool pool_pass = __any(phys || (mode == M_RESET && valid));
    FSTAMP(4);
    FSTAMP(5);
    // 1v1 discrete: the shot's step from the heading table by angle addition instead of a float64 sincos (below).  Larger teams keep
    // the sincos: there the shorter shot measured SLOWER (4v4 23.3 -> 25.3 us, two runs each) -- the table entry it needs arrives
    // later than the ~110 instructions of the sincos take, and nothing else is left to cover it.
    constexpr bool CHEAP_SHOT = !CONT && (N == 1 || (X_CHEAP_ALL && N > 0));
    double2 nd = make_double2(0.0, 0.0);                 // this call's shot: float64 step (CHEAP_SHOT: to ~1e-8 unless flagged exact), step code, heading
    double nbdir = 0.0;
    uint32_t ncode = 0u;
    bool nexact = false;
    if (spawn) {
        double uu = uu_in;
        if (!u_t && !(DIAG & 8u)) {
            const uint4 r = draw4(seed_t, genv, STREAM_JITTER, games, (uint32_t(tick) << 8) | uint32_t(a));
            uu = uniform53(r.x, r.y);
        }
        const double jit = uu * 8.0 - 4.0;
        nbdir = d0 + jit;
        if constexpr (CHEAP_SHOT) {
            // Discrete headings are whole degrees and a shooter does not turn, so (21.5 cos d0, -21.5 sin d0) is the heading-table
            // entry `dl` this lane gathered for its move; the jitter is at most 4 degrees.  The integer step code only needs the
            // step to ~2^-18 (step_code's guard is wider than any error here), so the common path takes it from the angle-addition
            // formulas with two-term series for the jitter -- |error| < 1e-8 on 45 cos -- instead of a float64 sincos of ~110
            // instructions.  A shot the code flags as not provably exact (one in ~30 000) gets the library sincos below, behind the
            // wave-uniform branch of the exact path; every other shot's integer moves are those of the exact step (same floor, the
            // fraction far from 0 and 1), so the results do not change.
            const double jr = jit * DEG2RAD, t = jr * jr;
            const double cj = __builtin_fma(t, __builtin_fma(t, 1.0 / 24.0, -0.5), 1.0);
            const double sj = jr * __builtin_fma(t, __builtin_fma(t, 1.0 / 120.0, -1.0 / 6.0), 1.0);
            constexpr double K45 = BULLET_STEP / 21.5;
            nd = make_double2(K45 * __builtin_fma(dl.x, cj, dl.y * sj), K45 * __builtin_fma(dl.y, cj, -(dl.x * sj)));
        } else {
            double sn, cs;
            sincos(-(nbdir * DEG2RAD), &sn, &cs);
            nd = make_double2(BULLET_STEP * cs, BULLET_STEP * sn);
        }
        ncode = step_code(nd.x, nd.y, nexact);
        st_store<NT_STATE>(elem(p.st.bdir, ix_t(ks) * EAt + gt), nbdir);      // ring by birth tick: never moves, read only by bsx_export_state
        // the shot as a pool entry, queued by shot rank: age 0, the PRE-move pose, my lane as its owner
        s_new[srank] = u32x2{pack_bullet(x, y, 0) | (nexact ? ENT_EXACT : 0u) | (uint32_t(lane) << ENT_OWNER_SHIFT), ncode};
    }
    // rare (step_code): a shot that moves by the float64 sum.  Asked once per wave, here, long before anything branches on it
    const bool shot_exact = __any(spawn && nexact);
    FSTAMP(6);

    STAMP(2);
    if (mode == M_RESET) {
        // re-spawn in place of the inert call; episode id = games played so far
        spawn_bases(seed_t, genv, STREAM_AUTORESET, games, er);
        er.bhp_r = er.bhp_b = 5 * n;
        er.tick = 0; er.done = 0; er.winner = BSX_WINNER_NONE;
        tick = 0;
        spawn_plane(seed_t, genv, STREAM_AUTORESET, games, a < A ? a : A - 1, n, x, y, dir);
        hp = PLANE_HP;
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

    // ---- hand the post-move pose and hit points to the other planes of the game.  1v1: the only other plane is the lane
    //      next door, three cross-lane moves (DPP) instead of LDS round trips; larger teams stage the block in LDS.
    int nx_ = 0, ny_ = 0, nhp_ = 0;                      // 1v1: the enemy's x, y, hit points
    s_pq[tid] = make_rect(pack_xy(x, y), valid && hp > 0, 27, 25, 27, 24);
    if constexpr (N == 1) {
        nx_ = __shfl_xor(x, 1); ny_ = __shfl_xor(y, 1); nhp_ = __shfl_xor(valid ? hp : 0, 1);
    } else {
        s_x[tid] = x; s_y[tid] = y; s_hp[tid] = valid ? hp : 0;
        s_bhit[tid] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    PSTAMP(3);
    // ---- observation geometry (battle_env.py:202-244) from the staged block, BEFORE the bullets: poses are final after
    //      the move, only the alive flags can still change; this fp64 math runs while the bullet-step loads are in flight.
    if (shot_exact) {                                    // wave-uniform and rare: a flagged shot leaves its float64 step in the ring (and in LDS for its first update)
        if constexpr (N == 1) asm volatile("");          // (keeps this a scalar branch; see the bullet rounds)
        if (spawn && nexact) {
            if constexpr (CHEAP_SHOT) {                  // the exact float64 step, as Bullet.update evaluates it (sprites.py:35-42,330-333)
                double sn, cs;
                sincos(-(nbdir * DEG2RAD), &sn, &cs);
                nd = make_double2(BULLET_STEP * cs, BULLET_STEP * sn);
            }
            *elem(p.st.bd, ix_t(ks) * EAt + gt) = nd;
            s_nd[2 * tid] = nd.x; s_nd[2 * tid + 1] = nd.y;
        }
    }
    const int obx = team == 0 ? er.bbx : er.brx, oby = team == 0 ? er.bby : er.bry;   // enemy base
    float ob_d = -1.0f, ob_a = -1.0f;
    float oe_d[NE], oe_a[NE];
    int ex[NE], ey[NE];
    if constexpr (N == 0) {                                          // runtime-n build: the enemy planes' pairs are worked out at row assembly
        if (!(DIAG & 1u)) obs_pair(x, y, dir, obx, oby, ob_d, ob_a);
    } else if constexpr (N == 1) {
        ex[0] = nx_; ey[0] = ny_;
        oe_d[0] = -1.0f; oe_a[0] = -1.0f;
        if (!(DIAG & 1u)) {                                          // enemy base and enemy plane, the two evaluations in lockstep
            const int tx[2] = {obx, nx_}, ty[2] = {oby, ny_};
            float d[2]; double rd[2];
            geometry_n<2>(x, y, tx, ty, d, rd);
            ob_d = d[0]; ob_a = float(rel_from_rads(rd[0], dir) * (1.0 / 360.0));
            oe_d[0] = d[1]; oe_a[0] = float(rel_from_rads(rd[1], dir) * (1.0 / 360.0));
        }
    } else if constexpr (N >= 2) {
        // The range of a pair is symmetric and its bearing differs by pi between the two ends, so each red-blue pair is
        // worked out once -- by red plane i for blue j when i + j is even, by blue j otherwise -- in (N + 1) / 2 rounds of
        // one sqrt + atan2 per lane instead of N, and the other end derives its bearing: rads +- pi (coincident planes: 0,
        // as atan2(+0, +0) gives both ends).  The derived value can differ from a direct atan2 in its last bits (<= ~4 ulp
        // of float64), which survives the single rounding to float32 with probability ~1e-8, like the libm difference.
        constexpr double PI_D = 3.14159265358979323846;
        const int mi = min(team == 0 ? a : a - N, N - 1);            // my index inside my team (lanes beyond A: clamped, never write)
#pragma unroll
        for (int j = 0; j < NE; ++j) { ex[j] = s_x[eb + j]; ey[j] = s_y[eb + j]; oe_d[j] = -1.0f; oe_a[j] = -1.0f; }
        if (!(DIAG & 1u)) {
            // the enemy base and the (N + 1) / 2 pairs this lane owns: all evaluations in lockstep (geometry_n)
            constexpr int R = (N + 1) / 2;
            int tx[R + 1], ty[R + 1], ojc[R];
            float d[R + 1]; double rd[R + 1];
            tx[0] = obx; ty[0] = oby;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oj = (team == 0 ? (mi & 1) : ((mi + 1) & 1)) + 2 * r;   // the enemy I own in this round
                ojc[r] = min(oj, N - 1);
                tx[r + 1] = s_x[eb + ojc[r]]; ty[r + 1] = s_y[eb + ojc[r]];
            }
            geometry_n<R + 1>(x, y, tx, ty, d, rd);
            ob_d = d[0]; ob_a = float(rel_from_rads(rd[0], dir) * (1.0 / 360.0));
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int oj = (team == 0 ? (mi & 1) : ((mi + 1) & 1)) + 2 * r;
                const bool own = oj < N && a < A;
                const int slot = (gl + (team == 0 ? mi : ojc[r])) * N + (team == 0 ? ojc[r] : mi);   // [red plane of my game][blue index]
                if (own) { s_pd[slot] = d[r + 1]; s_pr[slot] = rd[r + 1]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int ri = team == 0 ? mi : j, bi = team == 0 ? j : mi;
                const int slot = (gl + ri) * N + bi;
                const bool mine = (team == 0) == (((ri + bi) & 1) == 0);
                const double r0 = s_pr[slot];
                const bool same = ex[j] == x && ey[j] == y;
                const double rads = mine ? r0 : (same ? 0.0 : (r0 < PI_D ? r0 + PI_D : r0 - PI_D));
                oe_d[j] = s_pd[slot];
                oe_a[j] = float(rel_from_rads(rads, dir) * (1.0 / 360.0));
            }
        }
    }

    PSTAMP(4);
    // ---- Bullet.update (sprites.py:321-351) per work slot, predicates as integer sign masks (0 / -1).
    uint64_t ovl[OW];
#pragma unroll
    for (int q = 0; q < OW; ++q) ovl[q] = 0;
    int nmiss = 0, nbase = 0, nplane = 0;
    // the float64 move of the rare entries that carry the exact-path flag (exact_step() fetches the step the shot left in the ring)
    auto move_exact = [&](uint32_t ew, auto exact_step) {
        const double2 dd = exact_step();
        const int ebx = int(double(bullet_x(ew)) + dd.x);    // truncation toward zero
        const int eby = int(double(bullet_y(ew)) + dd.y);
        return (uint32_t(ebx) & 0xFFFFu) | (uint32_t(eby) << 16);
    };
    bool any_hit = false;
    if (pool_pass) {
        // ---- wave-packed bullet pass, part 2: Bullet.update per work slot.  A slot reads what its bullet's OWNER would have had
        // in registers -- the enemy base, the enemy planes' post-move poses and alive flags -- from the wave's LDS block, moves the
        // bullet, and hands the outcome back: one LDS add per bullet that ended (miss and base-hit counts), the survivor straight to
        // its place in the compacted pool (a wave-wide prefix count of the survivors: one ballot), the rare plane-overlap candidates
        // as the by-age bit fields the ordered resolve below walks.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the pool's first entries arrived long ago, behind the shot and the observation geometry
                                                         // (and no later wait is held up by the stores issued since: vmcnt is in-order)
        // a round's survivor stores are issued at the START of the next round (after the loop for the last one), i.e. BEFORE the
        // loads of the round after: loads and stores share the in-order vmcnt, so a wait for loads issued ahead of stores would
        // also sit out the stores' acknowledgement; issued behind them, both are long done when the entries are needed.  A round's
        // survivors land below that round's first slot, i.e. never on an entry that is still to be read.
        bool st_on = false; int st_ps = 0; uint2 st_w = make_uint2(0u, 0u);
        auto flush_stores = [&]() {
            if (st_on) st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.bent, pool0 + ix_t(st_ps))), u32x2{st_w.x, st_w.y});
        };
        const ix_t gb0 = ix_t(wblk * EPB) * ix_t(A);     // row of lane 0's agent; owner lane o sits (o / G) * A + (o % G) rows on
        // what a slot needs from its owner's side of the game, read from the wave's LDS block: enemy base, enemy planes (post-move
        // pose + alive flag), the owner's flags
        struct Ctx { rect_t ebw, pq[NE]; uint32_t fl; int ebl; };
        auto fetch_ctx = [&](int o) {
            Ctx c;
            c.ebw = s_eb[o];
            c.ebl = (o & ~(G - 1)) + (((o & (G - 1)) < n) ? n : 0);          // first lane of the owner's enemy team
            if constexpr (N > 0) {
#pragma unroll
                for (int q = 0; q < NE; ++q) c.pq[q] = s_pq[c.ebl + q];
            }
            c.fl = s_fl[o];
            return c;
        };
        int wpos = 0;                                    // survivors written so far = the pool's new length (wave-uniform)
        uint2 nxt = pool_first;                          // slot rd * 64 + lane of the round about to run, as loaded from the pool
        auto do_round = [&](const int rd) {
            const int w = rd * SPB + lane;
            const bool on = w < slots;
            uint2 en = nxt;
            if (w >= int(pc)) { const u32x2 sh = s_new[on ? w - int(pc) : 0]; en = make_uint2(sh.x, sh.y); }                 // one of this call's shots (LDS, by shot rank)
            const int o = on ? int(en.x >> ENT_OWNER_SHIFT) : lane;
            const Ctx c = fetch_ctx(o);
            if (rd > 0) flush_stores();
            // more than 64 slots in the wave: the next round's pool entries are fetched while this one is worked on
            if ((rd + 1) * SPB < int(pc)) nxt = *elem(p.st.bent, pool0 + ix_t((rd + 1) * SPB + lane));
            const uint32_t age0f = en.x & ENT_AGE;                              // updates so far, << 11
            const bool ophys = on && (c.fl & OWN_PHYS) != 0u;                   // the owner's game is in its physics call
            const int lvm = (ophys && age0f != (TOMBSTONE_AGE << 11)) ? -1 : 0;  // a tombstone (plane hit last call) is dropped
            // Everything from here to the outcome works on (x, y) PAIRS in the two 16-bit halves of a register: the move, and every
            // rectangle test as "some lower or upper margin is negative" = a sign bit in either half.
            uint32_t bpk = step_pk(en.x & ENT_XY, en.y);
            if (__any(lvm != 0 && (en.x & ENT_EXACT) != 0u)) {                  // wave-uniform and rare: the float64 move of flagged entries
                if constexpr (N == 1) asm volatile("");   // (1v1: keeps this a scalar branch on the common path)
                if (lvm != 0 && (en.x & ENT_EXACT) != 0u) {                     // (this call's shot left its step in LDS, older ones in the ring by birth tick)
                    const ix_t go = gb0 + ix_t(o / G) * ix_t(A) + ix_t(o & (G - 1));
                    bpk = move_exact(en.x, [&]() {
                        return age0f == 0u ? make_double2(s_nd[2 * o], s_nd[2 * o + 1])
                                           : *elem(p.st.bd, ix_t(ring_pos(int(c.fl & 15u), int(age0f >> 11))) * EAt + go);
                    });
                }
            }
            // miss: off the field (x > 1200 | x < 0 | y > 800 | y < 0), or dist_travelled >= 500 <=> this is the 12th update.  Plain
            // 32-bit arithmetic with literals on the packed pair: a half that borrows from (or carries into) its neighbour does so
            // only when a coordinate is negative or beyond the limit -- the bullet is a miss then, whatever the other half says, and
            // the base / plane results below are discarded for a miss.
            const uint32_t over = CORNERS ? pk_const(FIELD_W, FIELD_H) - bpk : pk_bits(as_pk(pk_const(FIELD_W, FIELD_H)) - as_pk(bpk));
            const int missm = pk_any_negative(bpk | over | (0x5000u - age0f));
            const s16x2 b2 = as_pk(bpk + (CORNERS ? pk_const(PK_BIAS, PK_BIAS) : 0u));
            // base: 6x3 bullet rect vs 62x62 base rect, strict overlap <=> dx in [-33, 33] and dy in [-32, 31]
            const int basem = hits_rect(b2, c.ebw, 33, 32, 33, 31) & ~missm;
            // planes: vs the un-rotated 50x48 rect at the post-move pose <=> dx in [-27, 27] and dy in [-25, 24]
            uint32_t m = 0;
            if constexpr (N > 0) {
#pragma unroll
                for (int q = 0; q < NE; ++q) m |= uint32_t(hits_rect(b2, c.pq[q], 27, 25, 27, 24)) & (1u << q);
            } else {
                for (int q = 0; q < n; ++q) m |= uint32_t(hits_rect(b2, s_pq[c.ebl + q], 27, 25, 27, 24)) & (1u << q);
            }
            const int age = int(age0f >> 11) + 1;
            const int gonem = (missm | basem) & lvm;
            const int keepm = lvm & ~gonem;
            m &= uint32_t(keepm);
            // One LDS add hands a bullet that ended to its owner: misses << 16 | base hits << 24.
            const uint32_t add = (uint32_t(missm & lvm & 1) << 16) | (uint32_t(basem & lvm & 1) << 24);
            if (add) __hip_atomic_fetch_add(&s_agg[o], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // What stays in the pool: a bullet that flies on (new position, age + 1), and -- untouched -- the entries of games that are
            // not in their physics call (finished and waiting, or tied by this call); the entries of a game this call re-spawns go.
            const bool asis = on && !ophys && (c.fl & OWN_DROP) == 0u;
            const bool stay = keepm != 0 || asis;
            const unsigned long long kb = __ballot(stay);
            const int ps = wpos + int(__builtin_amdgcn_mbcnt_hi(uint32_t(kb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(kb), 0u)));
            wpos += __popcll(kb);
            st_on = stay; st_ps = ps;
            st_w = make_uint2(asis ? en.x : (((en.x & ~ENT_XY) | bpk) + 0x800u), en.y);   // the new position, age + 1; flag and owner as they were
            if (__any(m != 0u)) {                        // wave-uniform and rare: a bullet overlaps a live enemy plane
                any_hit = true;
                if (m != 0u) {
                    if constexpr (OW == 1) __hip_atomic_fetch_or(&s_ov[o], (unsigned long long)(m) << (age * FW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_fetch_or(&s_ov[o * OW + (age >> 2)], (unsigned long long)(m) << ((age & 3) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    s_pp[o * K + age] = uint16_t(ps);    // (age 1 .. 11 here: a 12th update is always a range miss)
                }
            }
        };
        // the first round stands alone (under sparse play it is the only one in 85 % of the waves): straight-line code, no loop-carried
        // copies of the prefetch registers
        if (slots > 0) do_round(0);
        for (int rd = 1; rd * SPB < slots; ++rd) do_round(rd);
        if (slots > 0) flush_stores();
        if (wpos != int(pc) || MULTI) {                  // the pool's new length (one word per wave)
            if (lane == 0 && !MULTI) *elem(p.st.bcnt, ix_t(wblk)) = uint32_t(wpos);
            pc = uint32_t(wpos);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t agg = s_agg[tid];
        nmiss = int((agg >> 16) & 0xFFu); nbase = int((agg >> 24) & 0xFFu);
        if (any_hit) {
#pragma unroll
            for (int q = 0; q < OW; ++q) { ovl[q] = s_ov[tid * OW + q]; s_ov[tid * OW + q] = 0ull; }
        }
        if (N != 1 && nbase) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(&s_bhit[gl + team]), nbase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PSTAMP(5);
    // ---- ordered plane-hit resolve (battle_env.py:332-360 with sprites.py:348-350): creation order = oldest age
    //      first, then shooter id; a plane killed earlier in the walk no longer stops later bullets.
    uint64_t any_ovl = 0;
#pragma unroll
    for (int q = 0; q < OW; ++q) any_ovl |= ovl[q];
    if (__ballot(any_ovl != 0ull) != 0ull && !(DIAG & 4u)) {   // wave-uniform: most waves have no candidate at all
        uint32_t consumed = 0;                                 // by age
        if constexpr (N == 1) {
            // one shooter per target: my candidates, oldest first, hit until the enemy's hit points run out; the rest fly on
            int left = nhp_;
            for (int ag = K - 1; ag >= 1; --ag) {
                const bool hit = ((ovl[0] >> (ag * FW)) & 1ull) != 0ull && left > 0;
                if (hit) { left -= 1; nplane += 1; consumed |= 1u << ag; }
            }
        } else
        for (int ag = K - 1; ag >= 1; --ag) {                  // oldest first; an age-12 bullet is always a range miss
            uint64_t wsel = ovl[0];
            if (OW == 3) wsel = ((ag >> 2) == 0) ? ovl[0] : (((ag >> 2) == 1) ? ovl[OW > 1 ? 1 : 0] : ovl[OW > 2 ? 2 : 0]);
            const uint32_t m = uint32_t(wsel >> (OW == 1 ? ag * FW : (ag & 3) * 16)) & ((1u << FW) - 1u);
            if (__ballot(m != 0) == 0ull) continue;            // wave-uniform: nobody has a candidate of this age
            for (int i = 0; i < n; ++i) {
                if (m != 0 && (a - (team ? n : 0)) == i) {
                    for (int j = 0; j < n; ++j) {
                        if (((m >> j) & 1u) && s_hp[eb + j] > 0) {
                            s_hp[eb + j] = s_hp[eb + j] - 1;                                 // Plane.hit
                            nplane += 1;
                            consumed |= 1u << ag;
                            break;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // a bullet that hit a plane is gone: its pool entry becomes a tombstone, dropped by the next call's compaction
        // (the survivor entry was stored by ANOTHER lane of this wave; let it land before its first word is overwritten)
        if (__any(consumed != 0u)) __builtin_amdgcn_s_waitcnt(0x0F70);
        while (consumed) {
            const int ag = __builtin_ctz(consumed);
            consumed &= consumed - 1u;
            elem(p.st.bent, pool0 + ix_t(s_pp[tid * K + ag]))->x = pack_bullet(0, 0, int(TOMBSTONE_AGE)) | (uint32_t(lane) << ENT_OWNER_SHIFT);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int nplane_other = 0, nbase_other = 0;               // 1v1: what the enemy's bullets did to me / to my base
    if constexpr (N == 1) { nplane_other = __shfl_xor(nplane, 1); nbase_other = __shfl_xor(nbase, 1); }

    // ---- rewards (battle_env.py:337-359), deaths, bases, win / tie (:363-372, :469-496)
    double rew = double(nmiss) * p.cfg.miss_punishment + double(nbase) * p.cfg.hit_base_reward +
                 double(nplane) * p.cfg.hit_plane_reward;
    bool alive = valid && hp > 0;
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

    PSTAMP(6);
    if (MULTI && !ACTOR && tk + 1 < p.T) din_next = decode(rin_next);   // the prefetch has long arrived; no store of this tick is out yet
    // ---- write back (MULTI: plane and game records travel in registers; memory gets them once, after the last tick)
    const bool last_tick = !MULTI || tk == p.T - 1;
    if (valid) {
        if (MULTI ? last_tick : (mode == M_PHYS || mode == M_RESET)) {
            const uint2 pw = pack_plane(x, y, hp, dir, CONT);
            st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.plane, gt)), u32x2{pw.x, pw.y});
            if constexpr (CONT) st_store<NT_STATE>(elem(p.st.pdirf, gt), dir);   // continuous headings are fractional: the float64 beside the record
        }
        out_store(elem(rew_t, gt), float(rew));
        out_store(elem(done_t, gt), er.done ? uint8_t(1) : uint8_t(alive ? 0 : 1));
    }
    // observation row: a dead observer sees all -1, a dead enemy is [-1,-1,-1] (battle_env.py:215-218,235-242).
    // Compile-time team sizes outside the fused rollout: the row leaves straight from registers, 16 bytes at a time plus a tail
    // (rows are 4 (3n + 2) bytes apart, so the stores are only dword-aligned -- fine for global_store_dwordx4).  Round 1 staged
    // rows in LDS to emit fully coalesced 16-byte stores; with non-temporal stores that transpose only costs: C2 8.21 -> 7.92 us,
    // 4v4 28.0 -> 24.9 (-DBSX_X_LDSOBS builds it for A/B).  The fused rollout keeps its rows in LDS (the actor reads them there).
    constexpr bool DIRECT_OBS = !ACTOR && N > 0 && OBS_FORM == 0;
    if constexpr (DIRECT_OBS) {
        constexpr int D = 3 * N + 2;
        float row[D];
        row[0] = alive ? ob_d : -1.0f;
        row[1] = alive ? ob_a : -1.0f;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const bool on = alive && ((N == 1) ? (mode == M_PHYS ? nhp_ - nplane : nhp_) : s_hp[eb + j]) > 0;
            row[2 + 3 * j] = on ? 1.0f : -1.0f;
            row[3 + 3 * j] = on ? oe_d[j] : -1.0f;
            row[4 + 3 * j] = on ? oe_a[j] : -1.0f;
        }
        if (valid) {
            float* out = elem(obs_t, gt * ix_t(D));
#pragma unroll
            for (int i = 0; i + 4 <= D; i += 4) out_store(reinterpret_cast<v4f_t*>(out + i), v4f_t{row[i], row[i + 1], row[i + 2], row[i + 3]});
            typedef float v2f_t __attribute__((ext_vector_type(2)));
            if constexpr ((D & 3) >= 2) out_store(reinterpret_cast<v2f_t*>(out + (D & ~3)), v2f_t{row[D & ~3], row[(D & ~3) + 1]});
            if constexpr ((D & 1) != 0) out_store(out + D - 1, row[D - 1]);
        }
    } else
    {
        // Rows are staged in LDS ([lane][D], D odd -> conflict-free) and leave as coalesced 16-byte stores: the wave's rows
        // are one contiguous block of global memory when every lane is an agent (G == A).
        const int D = 3 * n + 2;
        float* srow = &s_obs[tid * D];
        srow[0] = alive ? ob_d : -1.0f;
        srow[1] = alive ? ob_a : -1.0f;
        if (N > 0) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const bool on = alive && ((N == 1) ? (mode == M_PHYS ? nhp_ - nplane : nhp_) : s_hp[eb + j]) > 0;
                srow[2 + 3 * j] = on ? 1.0f : -1.0f;
                srow[3 + 3 * j] = on ? oe_d[j] : -1.0f;
                srow[4 + 3 * j] = on ? oe_a[j] : -1.0f;
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const bool on = alive && s_hp[eb + j] > 0;
                float od = -1.0f, oa = -1.0f;
                if (on && !(DIAG & 1u)) obs_pair(x, y, dir, s_x[eb + j], s_y[eb + j], od, oa);
                srow[2 + 3 * j] = on ? 1.0f : -1.0f; srow[3 + 3 * j] = od; srow[4 + 3 * j] = oa;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (OBS_FORM == 2 && N == 4 && !ACTOR && (reinterpret_cast<uintptr_t>(obs_t) & 63u) == 0) {
            // (variant builds, 4v4: a game's 8 rows are 448 contiguous bytes = seven 64-byte segments; lane j < 7 of the game writes
            //  segment j whole -- four 16-byte stores into ONE aligned 64-byte sector instead of rows that straddle sectors)
            if (valid && a < 7) {
                float* gseg = obs_t + (gt - ix_t(a)) * ix_t(D) + 16 * a;
                const float* sseg = &s_obs[gl * D + 16 * a];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) out_store(reinterpret_cast<v4f_t*>(gseg + 4 * k4), *reinterpret_cast<const v4f_t*>(sseg + 4 * k4));
            }
        } else if (G == A && (reinterpret_cast<uintptr_t>(obs_t) & 15u) == 0) {
            // rows of this wave: global floats [base, base + rows*D); the wave's offset SPB*D*4 bytes is a multiple of 16
            const int64_t e_first = wblk * EPB;
            const int64_t rows = min(int64_t(SPB), (E_ - e_first) * A);
            const int64_t nfl = rows * D;                                   // floats to write
            float* gbase = obs_t + size_t(e_first) * A * D;
            for (int i = tid * 4; i < nfl; i += SPB * 4) {
                if (i + 4 <= nfl) {
                    out_store(reinterpret_cast<v4f_t*>(gbase + i), *reinterpret_cast<const v4f_t*>(&s_obs[i]));   // ds_read_b128
                } else {
                    for (int t = i; t < nfl; ++t) gbase[t] = s_obs[t];
                }
            }
        } else if (valid) {
            float* out = obs_t + gt * size_t(D);
            for (int i = 0; i < D; ++i) out[i] = srow[i];
        }
    }
    if (valid) {
        if (a == 0) {
            if (MULTI ? last_tick : (mode != M_INERT)) {
                // the game record: hit points, clock, flags and the episode number; the base positions only when the game was re-spawned
                st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.envd, ix_t(e))), u32x2{pack_envd(er), games + uint32_t(cnt_delta.x)});
                if (MULTI || mode == M_RESET) {
                    const uint2 cw = pack_envc(er);
                    st_store<NT_STATE>(reinterpret_cast<u32x2*>(elem(p.st.envc, ix_t(e))), u32x2{cw.x, cw.y});
                }
            }
            if (cnt_delta.x) {                           // game over: the win / tie counters (nothing on the step path reads them: fire-and-forget atomics)
                int* const c4 = elem(p.st.cnt, ix_t(e) * 4);
                __hip_atomic_fetch_add(c4, cnt_delta.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.y) __hip_atomic_fetch_add(c4 + 1, cnt_delta.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.z) __hip_atomic_fetch_add(c4 + 2, cnt_delta.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cnt_delta.w) __hip_atomic_fetch_add(c4 + 3, cnt_delta.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last_tick) {
                if (p.env_done) *elem(p.env_done, ix_t(e)) = uint8_t(er.done);
                if (p.winner) *elem(p.winner, ix_t(e)) = uint8_t(er.winner);
            }
            if (MULTI && p.env_done_t) p.env_done_t[int64_t(tk) * E_ + e] = uint8_t(er.done);
        }
    }
    if (MULTI && last_tick && lane == 0) *elem(p.st.bcnt, ix_t(wblk)) = pc;   // the pool's length travelled in a register
    STAMP(7);
    if (MULTI) {
        games += uint32_t(cnt_delta.x);
        // The only memory one tick hands to the next is what a LANE stored itself and reloads itself (its bullet rows; the
        // game counters' read-modify-write) plus this wave's LDS rows.  A wavefront's vector memory operations are performed
        // in order through the one L1 of its CU, so wavefront scope is enough: a compiler ordering point, no s_waitcnt -- this
        // tick's stores (observation rows included) drain while the next tick computes.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(9);
    }
    }   // tick loop
}

// ---------------------------------------------------------------------------------------------- reset / observe
struct ResetArgs {
    StatePtrs st; int64_t E; int n; const uint8_t* mask; const int32_t* spawn; uint64_t seed; uint64_t nonce;
    int64_t env_offset; float* obs; int observe_only;
};

__global__ __launch_bounds__(TPB) void bsx_reset_kernel(const ResetArgs p) {
    const int n = p.n, A = 2 * n, G = group_width(n), EPB = TPB / G;
    const int tid = threadIdx.x, a = tid & (G - 1), gl = tid & ~(G - 1);
    const int64_t e = int64_t(blockIdx.x) * EPB + (tid / G);
    const bool env_ok = e < p.E, valid = env_ok && a < A;
    const size_t g = valid ? size_t(e) * A + a : 0;
    __shared__ volatile int s_x[TPB], s_y[TPB], s_hp[TPB];

    EnvU er = {};
    uint32_t games = 0;
    int x = 0, y = 0, hp = 0;
    double dir = 0.0;
    bool frac = false;
    if (env_ok) {
        const uint2 dw = p.st.envd[e];
        er = unpack_env(p.st.envc[e], dw.x);
        games = dw.y;
    }
    if (valid) {
        const uint2 pw = p.st.plane[g];
        unpack_plane(pw, x, y, hp, dir);
        frac = (pw.y & PLANE_FRAC) != 0u;
        if (frac) dir = p.st.pdirf[g];
    }
    const bool doit = env_ok && !p.observe_only && (!p.mask || p.mask[e]);
    if (doit) {
        const int64_t genv = p.env_offset + e;
        if (p.spawn) {
            const int32_t* s = p.spawn + size_t(e) * (4 + 3 * A);
            er.brx = s[0]; er.bry = s[1]; er.bbx = s[2]; er.bby = s[3];
            if (valid) { x = s[4 + 3 * a]; y = s[5 + 3 * a]; dir = double(s[6 + 3 * a]); }
        } else {
            spawn_bases(p.seed, genv, STREAM_RESET, uint32_t(p.nonce), er);
            if (valid) spawn_plane(p.seed, genv, STREAM_RESET, uint32_t(p.nonce), a, n, x, y, dir);
        }
        er.bhp_r = er.bhp_b = 5 * n;
        er.tick = 0; er.done = 0; er.winner = BSX_WINNER_NONE;
        hp = PLANE_HP; frac = false;                         // spawn headings are whole degrees (sprites.py:85,91; injected spawns are int32)
        if (valid) p.st.plane[g] = pack_plane(x, y, hp, dir, false);
        if (valid && a == 0) { p.st.envc[e] = pack_envc(er); p.st.envd[e] = make_uint2(pack_envd(er), games); }   // the episode number stays
    }
    // the bullets of a game that is reset go (battle_env.py:268): every wavefront of this kernel covers exactly one wave block of the
    // step kernels (64 lanes, the same lane <-> plane mapping), so it filters that block's pool: entries whose owner's game stays, stay
    if (!p.observe_only) {
        const int lane = tid & 63;
        const int64_t wb = int64_t(blockIdx.x) * (TPB / 64) + (tid >> 6);
        const unsigned long long resetting = __ballot(doit);
        if (resetting != 0ull && wb < wave_blocks(p.E, n)) {     // wave-uniform
            uint2* const pool = p.st.bent + size_t(wb) * POOL_CAP;
            const int pc = int(p.st.bcnt[wb]);
            int wpos = 0;
            for (int base = 0; base < pc; base += 64) {
                const int w = base + lane;
                const uint2 en = pool[w < pc ? w : 0];
                const bool stay = w < pc && ((resetting >> (en.x >> ENT_OWNER_SHIFT)) & 1ull) == 0ull;
                const unsigned long long kb = __ballot(stay);
                const int ps = wpos + __popcll(kb & ((1ull << lane) - 1ull));
                if (stay) pool[ps] = en;                     // ps <= w: lands on an entry this or an earlier round has already read
                wpos += __popcll(kb);
            }
            if (lane == 0) p.st.bcnt[wb] = uint32_t(wpos);
        }
    }
    s_x[tid] = x; s_y[tid] = y; s_hp[tid] = valid ? hp : 0;
    __syncthreads();
    if (valid && p.obs) {
        const int team = a < n ? 0 : 1;
        write_obs<0>(p.obs + g * size_t(3 * n + 2), n, hp > 0, x, y, dir, a,
                     team == 0 ? er.bbx : er.brx, team == 0 ? er.bby : er.bry, gl, s_x, s_y, s_hp);
    }
}

__global__ void bsx_mark_done_kernel(uint2* envd, int64_t E) {
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e < E) envd[e] = make_uint2(1u << 27, 0u);          // done = 1, everything else 0
}

struct ExportArgs { StatePtrs st; int64_t E; int n; BsxExport out; int tie_tick; };

__global__ __launch_bounds__(TPB) void bsx_export_kernel(const ExportArgs p) {
    const int A = 2 * p.n;
    const size_t EA = size_t(p.E) * A;
    const size_t g = size_t(blockIdx.x) * TPB + threadIdx.x;
    if (g >= EA) return;
    const int64_t e = int64_t(g / A);
    const int a = int(g % A);
    const uint2 pw = p.st.plane[g];
    int x, y, hp;
    double dir;
    unpack_plane(pw, x, y, hp, dir);
    if (pw.y & PLANE_FRAC) dir = p.st.pdirf[g];
    const BsxExport& o = p.out;
    if (o.px) o.px[g] = x;
    if (o.py) o.py[g] = y;
    if (o.pdir) o.pdir[g] = dir;
    if (o.php) o.php[g] = hp;
    if (o.palive) o.palive[g] = hp > 0;
    // pool entries -> the slot view of the export schema: slot = birth tick % 12, birth tick = (ticks on which bullets
    // were updated) - age + 1; the time-limit tie call advances the clock but not the bullets (battle_env.py:316-323)
    const uint2 dw = p.st.envd[e];
    const EnvU ev = unpack_env(p.st.envc[e], dw.x);
    const int ptick = ev.tick - ((ev.done && ev.winner == BSX_WINNER_TIE && ev.tick >= p.tie_tick) ? 1 : 0);
    for (int k = 0; k < K; ++k) {
        const size_t i = g * K + k;
        if (o.bl_live) o.bl_live[i] = 0;
        if (o.bl_x) o.bl_x[i] = 0;
        if (o.bl_y) o.bl_y[i] = 0;
        if (o.bl_dir) o.bl_dir[i] = 0.0;
    }
    // my bullets are the entries of my wave block's pool that name my lane (debug / test path: a plain scan)
    const int G = group_width(p.n), EPB = 64 / G;
    const int64_t wb = e / EPB;
    const uint32_t me = uint32_t(int(e % EPB) * G + a);
    const uint2* const pool = p.st.bent + size_t(wb) * POOL_CAP;
    const int cnt = int(p.st.bcnt[wb]);
    for (int j2 = 0; j2 < cnt; ++j2) {
        const uint32_t w = pool[j2].x;
        const int age = bullet_age(w);
        if ((w >> ENT_OWNER_SHIFT) != me || age == int(TOMBSTONE_AGE)) continue;
        int slot = (ptick - age + 1) % K;
        if (slot < 0) slot += K;
        const size_t i = g * K + slot;
        if (o.bl_live) o.bl_live[i] = 1;
        if (o.bl_x) o.bl_x[i] = bullet_x(w);
        if (o.bl_y) o.bl_y[i] = bullet_y(w);
        if (o.bl_dir) o.bl_dir[i] = p.st.bdir[size_t(slot) * EA + g];
    }
    if (a == 0) {
        if (o.base_xy) { o.base_xy[4 * e] = ev.brx; o.base_xy[4 * e + 1] = ev.bry; o.base_xy[4 * e + 2] = ev.bbx; o.base_xy[4 * e + 3] = ev.bby; }
        if (o.bhp) { o.bhp[2 * e] = ev.bhp_r; o.bhp[2 * e + 1] = ev.bhp_b; }
        if (o.tick) o.tick[e] = ev.tick;
        if (o.env_done) o.env_done[e] = ev.done;
        if (o.winner) o.winner[e] = ev.winner;
        if (o.counters) {
            const int* c4 = p.st.cnt + 4 * e;
            o.counters[4 * e] = c4[0]; o.counters[4 * e + 1] = c4[1]; o.counters[4 * e + 2] = c4[2]; o.counters[4 * e + 3] = c4[3];
        }
    }
}

// ---------------------------------------------------------------------------------------------- scripted opponent
// instinct/agent.py:10-62: decode the observation row, score every target by dist * |angle| (base wins ties, a dead
// enemy scores 1e6), then shoot / turn toward the chosen target.  binary64 on the float32 values, as the reference
// computes under its pinned numpy.
struct InstinctArgs {
    const float* obs; void* actions; const double* rnd; int64_t E; int n; int team; int out_kind; int continuous;
    uint64_t seed; uint64_t seq; const uint64_t* seq_base;
};

__global__ __launch_bounds__(TPB) void bsx_instinct_kernel(const InstinctArgs p) {
    const int A = 2 * p.n, D = 3 * p.n + 2;
    const size_t g = size_t(blockIdx.x) * TPB + threadIdx.x;
    if (g >= size_t(p.E) * A) return;
    const int a = int(g % A);
    const int tm = a < p.n ? 0 : 1;
    if (p.team != 2 && p.team != tm) return;
    const float* o = p.obs + g * D;
    double td, ta;
    const int act = instinct_choose([&](int k) { return o[k]; }, p.n, td, ta);
    if (!p.continuous) {
        if (p.out_kind == BSX_ACT_I32) static_cast<int32_t*>(p.actions)[g] = act;
        else static_cast<float4*>(p.actions)[g] = one_hot_scores(act);
        return;
    }
    double r0, n0, n1, n2;                                        // agent.py:41-54
    if (p.rnd) { r0 = p.rnd[4 * g]; n0 = p.rnd[4 * g + 1]; n1 = p.rnd[4 * g + 2]; n2 = p.rnd[4 * g + 3]; }
    else instinct_continuous_draws(p.seed, p.seq + (p.seq_base ? *p.seq_base : 0ull), uint64_t(g), r0, n0, n1, n2);
    double* out = static_cast<double*>(p.actions) + 3 * g;
    instinct_continuous_action(td, ta, r0, n0, n1, n2, out[0], out[1], out[2]);
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
inline int grid_for(int64_t E, int n, int tpb = TPB) {
    const int epb = tpb / group_width(n);
    return int((E + epb - 1) / epb);
}

template <bool CONT, bool MULTI, bool LG, bool OFF32>
void launch_for_n_w(int n, dim3 grid, dim3 block, hipStream_t s, const StepArgs& a, int64_t bound) {
    switch (n) {
        case 1: hipLaunchKernelGGL((bsx_step_kernel<1, CONT, MULTI, false, LG, OFF32>), grid, block, 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        case 2: hipLaunchKernelGGL((bsx_step_kernel<2, CONT, MULTI, false, LG, OFF32>), grid, block, 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        case 3: hipLaunchKernelGGL((bsx_step_kernel<3, CONT, MULTI, false, LG, OFF32>), grid, block, 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        case 4: hipLaunchKernelGGL((bsx_step_kernel<4, CONT, MULTI, false, LG, OFF32>), grid, block, 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        default: hipLaunchKernelGGL((bsx_step_kernel<0, CONT, MULTI, false, LG, OFF32>), grid, block, 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
    }
}
// 32-bit offsets when every array of the job stays below 4 GB: an observation row is at most 4 (3 * 16 + 2) = 200 bytes per agent,
// the widest state rows are the exact-path step ring's (12 x 16 bytes per agent, allocated but all but never touched).
inline bool narrow_offsets_ok(int64_t E, int n, uint32_t flags) {
    return !(flags & BSX_F_WIDE_OFFSETS) && uint64_t(E) * uint64_t(2 * n) * 200ull <= 0xFFFFFFFFull;
}
template <bool CONT, bool MULTI, bool LG>
void launch_for_n(int n, dim3 grid, dim3 block, hipStream_t s, const StepArgs& a, int64_t bound) {
    if (narrow_offsets_ok(a.E, n, a.flags)) launch_for_n_w<CONT, MULTI, LG, true>(n, grid, block, s, a, bound);
    else launch_for_n_w<CONT, MULTI, LG, false>(n, grid, block, s, a, bound);
}

// T == 0: one call (bsx_step_*);  T >= 1: bsx_step_many_* -- T calls in one launch, arrays with a leading T axis
template <bool CONT>
int launch_step(void* state, int64_t E, int n, const void* actions, int action_kind, const double* u, float* obs,
                float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, const BsxRewards* cfg, uint32_t flags,
                uint64_t seed, int64_t env_offset, void* stream, int T = 0, int store_all = 0, uint8_t* env_done_t = nullptr,
                int64_t first = 0, int64_t count = -1) {
    if (T < 0 || T > BSX_MAX_T) return BSX_E_ARG;
    if (count < 0) count = E - first;
    // a sub-range of the games (bsx_step_*_range): whole 256-game blocks, so that every array the launch is handed stays aligned as the
    // full arrays are; the one-call form only (a multi-tick launch strides its per-tick arrays by the games it steps)
    if (first < 0 || count <= 0 || first > E - count || (first & 255) || ((first || count != E) && T != 0)) return BSX_E_ARG;
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || !obs || !rew || !done || !cfg) return BSX_E_ARG;
    if (!actions && !(flags & BSX_F_EMPTY_CALL)) return BSX_E_ARG;
    if (!aligned(state, 256) || !aligned(obs, 4) || !aligned(rew, 4) || (u && !aligned(u, 8))) return BSX_E_ALIGN;
    if (!CONT && action_kind == BSX_ACT_LOGITS_F32 && !aligned(actions, 16)) return BSX_E_ALIGN;
    if (!CONT && action_kind != BSX_ACT_I32 && action_kind != BSX_ACT_LOGITS_F32) return BSX_E_ARG;
    if (CONT && action_kind != BSX_ACT_F32 && action_kind != BSX_ACT_F64 && action_kind != BSX_ACT_F32X4) return BSX_E_ARG;
    if (CONT && action_kind == BSX_ACT_F32X4 && !aligned(actions, 16)) return BSX_E_ALIGN;
    StepArgs a;
    a.st = state_ptrs(state, E, n);
    a.E = E; a.n = n; a.actions = actions; a.action_kind = action_kind; a.u = u;
    if (!actions) { a.actions = state; a.action_kind = -1; }   // empty call: the kernel reads (and ignores) one mapped word instead of selecting a pointer
    a.obs = obs; a.rew = rew; a.done = done; a.env_done = env_done; a.winner = winner; a.env_done_t = env_done_t;
    a.cfg = *cfg; a.flags = flags; a.seed = seed; a.env_offset = env_offset; a.tie_tick = bsx_tie_tick(n);
    const int64_t EA = E * 2 * n;
    a.aw = nullptr; a.aprec = 0; a.scripted_team = -1; a.iseed = 0; a.obs0 = nullptr; a.scores = nullptr; a.scores_ts = 0; a.nz = BsxActorNoise{};
    a.aseed = 0; a.aseq = 0; a.aseq_base = nullptr;
    a.T = T;
    a.act_tb = EA * (CONT ? (action_kind == BSX_ACT_F32 ? 12 : (action_kind == BSX_ACT_F64 ? 24 : 16)) : (action_kind == BSX_ACT_I32 ? 4 : 16));
    a.u_ts = EA;
    a.obs_ts = store_all ? EA * (3 * n + 2) : 0; a.rew_ts = store_all ? EA : 0; a.done_ts = store_all ? EA : 0;
    if (first) {
        // Every array is indexed by game, by agent row (game * 2n + plane) or by wave block (the bullet pools), the two birth-tick rings by
        // slot * (E * 2n) + agent row with the stride taken from a.E: advancing each pointer to the range's first row turns the kernel's
        // row r into row first + r of the full arrays.
        const int64_t fa = first * 2 * n;
        const int64_t fb = first / (64 / group_width(n));     // the range's first wave block (first is a multiple of 256 games)
        a.st.envc += first; a.st.envd += first; a.st.cnt += 4 * first; a.st.plane += fa; a.st.pdirf += fa;
        a.st.bcnt += fb; a.st.bent += fb * POOL_CAP; a.st.bdir += fa; a.st.bd += fa;
        if (actions) a.actions = static_cast<const char*>(actions) + fa * (a.act_tb / EA);
        if (u) a.u += fa;
        a.obs += fa * (3 * n + 2); a.rew += fa; a.done += fa;
        if (env_done) a.env_done += first;
        if (winner) a.winner += first;
        a.env_offset += first;                             // the draws are keyed by the global game index
    }
    const dim3 grid(grid_for(count, n, SPB * WPB)), block(SPB * WPB);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lg = !CONT && action_kind == BSX_ACT_LOGITS_F32;
    if (T == 0) {
        if (lg) launch_for_n<CONT, false, !CONT>(n, grid, block, s, a, count);
        else launch_for_n<CONT, false, false>(n, grid, block, s, a, count);
    } else {
        if (lg) launch_for_n<CONT, true, !CONT>(n, grid, block, s, a, count);
        else launch_for_n<CONT, true, false>(n, grid, block, s, a, count);
    }
    return int(hipGetLastError());
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int bsx_abi_version(void) { return BSX_ABI_VERSION; }

int bsx_build_flags(void) { return BUILD_FLAGS; }

int bsx_stream_synchronize(void* stream) { return int(hipStreamSynchronize(static_cast<hipStream_t>(stream))); }

int bsx_host_device_pointer(void* host, void** device) {
    if (!host || !device) return BSX_E_ARG;
    return int(hipHostGetDevicePointer(device, host, 0));
}

int bsx_tie_tick(int n) {
    // battle_env.py:168,316-319: total_time += 0.1 (binary64) until >= 10 + 2n
    const double max_time = double(10 + n * 2);
    volatile double t = 0.0;
    int k = 0;
    for (;;) {
        t = t + 0.1;
        ++k;
        if (t >= max_time) return k;
    }
}

int bsx_state_bytes(int64_t E, int n, size_t* bytes) {
    if (E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || !bytes) return BSX_E_ARG;
    *bytes = make_layout(E, n).total;
    return 0;
}

int bsx_state_init(void* state, int64_t E, int n, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N) return BSX_E_ARG;
    if (!aligned(state, 256)) return BSX_E_ALIGN;
    const Layout L = make_layout(E, n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t err = hipMemsetAsync(state, 0, L.total, s);
    if (err != hipSuccess) return int(err);
    // heading table through the host libm, exactly as calc_new_xy evaluates it (sprites.py:40-41) with speed*time = 215*0.1
    static double2 lut[361];
    const double st = 215 * 0.1;
    for (int d = 0; d <= 360; ++d) {
        const double ang = -(double(d) * DEG2RAD);
        lut[d].x = st * cos(ang);
        lut[d].y = st * sin(ang);
    }
    err = hipMemcpyAsync(static_cast<char*>(state) + L.lut, lut, sizeof(lut), hipMemcpyHostToDevice, s);
    if (err != hipSuccess) return int(err);
    // every env starts finished (done = 1), so a step before the first reset is the inert call of battle_env.py:303-306
    hipLaunchKernelGGL(bsx_mark_done_kernel, dim3(unsigned((E + TPB - 1) / TPB)), dim3(TPB), 0, s,
                       reinterpret_cast<uint2*>(static_cast<char*>(state) + L.envd), E);
    return int(hipGetLastError());
}

int bsx_reset(void* state, int64_t E, int n, const uint8_t* reset_mask, const int32_t* spawn, uint64_t seed,
              uint64_t nonce, int64_t env_offset, float* obs, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N) return BSX_E_ARG;
    if (!aligned(state, 256) || (spawn && !aligned(spawn, 4)) || (obs && !aligned(obs, 4))) return BSX_E_ALIGN;
    ResetArgs a{state_ptrs(state, E, n), E, n, reset_mask, spawn, seed, nonce, env_offset, obs, 0};
    hipLaunchKernelGGL(bsx_reset_kernel, dim3(grid_for(E, n)), dim3(TPB), 0, static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

int bsx_step_discrete(void* state, int64_t E, int n, const void* actions, int action_kind, const double* u, float* obs,
                      float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, const BsxRewards* cfg,
                      uint32_t flags, uint64_t seed, int64_t env_offset, void* stream) {
    return launch_step<false>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                              env_offset, stream);
}

int bsx_step_continuous(void* state, int64_t E, int n, const void* actions, int action_kind, const double* u,
                        float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner,
                        const BsxRewards* cfg, uint32_t flags, uint64_t seed, int64_t env_offset, void* stream) {
    return launch_step<true>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                             env_offset, stream);
}

int bsx_step_discrete_range(void* state, int64_t E, int n, int64_t first, int64_t count, const void* actions, int action_kind, const double* u,
                            float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, const BsxRewards* cfg,
                            uint32_t flags, uint64_t seed, int64_t env_offset, void* stream) {
    return launch_step<false>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                              env_offset, stream, 0, 0, nullptr, first, count);
}

int bsx_step_continuous_range(void* state, int64_t E, int n, int64_t first, int64_t count, const void* actions, int action_kind, const double* u,
                              float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, const BsxRewards* cfg,
                              uint32_t flags, uint64_t seed, int64_t env_offset, void* stream) {
    return launch_step<true>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                             env_offset, stream, 0, 0, nullptr, first, count);
}

int bsx_step_many_discrete(void* state, int64_t E, int n, int T, const void* actions, int action_kind, const double* u,
                           float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg,
                           uint32_t flags, int store_all, uint64_t seed, int64_t env_offset, void* stream) {
    if (T < 1 || !actions) return BSX_E_ARG;
    return launch_step<false>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                              env_offset, stream, T, store_all, env_done_t);
}

int bsx_step_many_continuous(void* state, int64_t E, int n, int T, const void* actions, int action_kind, const double* u,
                             float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg,
                             uint32_t flags, int store_all, uint64_t seed, int64_t env_offset, void* stream) {
    if (T < 1 || !actions) return BSX_E_ARG;
    return launch_step<true>(state, E, n, actions, action_kind, u, obs, rew, done, env_done, winner, cfg, flags, seed,
                              env_offset, stream, T, store_all, env_done_t);
}

}  // extern "C"

namespace {
// T x (actor -> step) in one launch; CONT = continuous actions (three actor outputs, BSX_ACT_F32X4 rows)
template <bool CONT, bool OFF32>
void launch_rollout_w(int n, dim3 grid, hipStream_t s, const StepArgs& a) {
    switch (n) {
        case 1: hipLaunchKernelGGL((bsx_step_kernel<1, CONT, true, true, false, OFF32>), grid, dim3(SPB * 1), 0, s, a.E, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        case 2: hipLaunchKernelGGL((bsx_step_kernel<2, CONT, true, true, false, OFF32>), grid, dim3(SPB * 2), 0, s, a.E, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        case 3: hipLaunchKernelGGL((bsx_step_kernel<3, CONT, true, true, false, OFF32>), grid, dim3(SPB * 4), 0, s, a.E, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
        default: hipLaunchKernelGGL((bsx_step_kernel<4, CONT, true, true, false, OFF32>), grid, dim3(SPB * 4), 0, s, a.E, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a); break;
    }
}
template <bool CONT>
int launch_rollout(void* state, int64_t E, int n, int T, const float* weights, int precision, int scripted_team, uint64_t scripted_seed, float* obs, float* scores, float* rew,
                   uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg, uint32_t flags,
                   const BsxActorNoise* noise, uint64_t actor_seed, uint64_t seq, const uint64_t* seq_base, uint64_t seed,
                   int64_t env_offset, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > 4 || T < 1 || T > BSX_MAX_T || !weights || !obs || !scores || !rew || !done || !cfg)
        return BSX_E_ARG;
    if ((flags & BSX_F_EMPTY_CALL) || (precision != BSX_ACTOR_F32 && precision != BSX_ACTOR_BF16X3 && precision != BSX_ACTOR_BF16X6) ||
        scripted_team < -1 || scripted_team > 1) return BSX_E_ARG;
    if (!aligned(state, 256) || !aligned(weights, 16) || !aligned(scores, 16) || !aligned(obs, 4) || !aligned(rew, 4)) return BSX_E_ALIGN;
    BsxActorNoise nz = {};
    if (noise) nz = *noise;
    if (nz.ou_scale > 0.f && (!nz.ou_state || !aligned(nz.ou_state, 16))) return nz.ou_state ? BSX_E_ALIGN : BSX_E_ARG;
    if (nz.z_inject || nz.u_inject) return BSX_E_ARG;    // injected draws are per call: bsx_actor_forward only
    if (nz.sample_mode != 0 && (nz.sample_mode != 1 || !(nz.temperature > 0.f) || CONT)) return BSX_E_ARG;   // a categorical head belongs to discrete actions
    if ((nz.logp && !aligned(nz.logp, 4)) || (nz.value_weights && !aligned(nz.value_weights, 16))) return BSX_E_ALIGN;
    if (nz.value_weights && (n != 1 || !nz.value || !aligned(nz.value, 4))) return BSX_E_ARG;          // the value head rides in the 1v1 kernel only
    const int64_t EA = E * 2 * n, D = 3 * n + 2;
    StepArgs a;
    a.st = state_ptrs(state, E, n);
    a.E = E; a.n = n; a.actions = nullptr; a.action_kind = CONT ? BSX_ACT_F32X4 : BSX_ACT_LOGITS_F32; a.u = nullptr;
    a.obs = obs + EA * D; a.rew = rew; a.done = done; a.env_done = env_done; a.winner = winner; a.env_done_t = env_done_t;
    a.cfg = *cfg; a.flags = flags; a.seed = seed; a.env_offset = env_offset; a.tie_tick = bsx_tie_tick(n);
    a.T = T; a.act_tb = 0; a.u_ts = 0; a.obs_ts = EA * D; a.rew_ts = EA; a.done_ts = EA;
    a.aw = weights; a.aprec = precision; a.scripted_team = scripted_team; a.obs0 = obs; a.scores = scores; a.scores_ts = EA * 4; a.nz = nz; a.aseed = actor_seed; a.aseq = seq;
    a.aseq_base = seq_base; a.iseed = scripted_seed;
    const dim3 grid(unsigned((E + 31) / 32));            // a workgroup = 32 games = G/2 waves
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (narrow_offsets_ok(E, n, flags)) launch_rollout_w<CONT, true>(n, grid, s, a);
    else launch_rollout_w<CONT, false>(n, grid, s, a);
    return int(hipGetLastError());
}
}  // namespace

extern "C" {

int bsx_rollout_discrete(void* state, int64_t E, int n, int T, const float* weights, int precision, int scripted_team, float* obs, float* scores, float* rew,
                         uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg, uint32_t flags,
                         const BsxActorNoise* noise, uint64_t actor_seed, uint64_t seq, const uint64_t* seq_base, uint64_t seed,
                         int64_t env_offset, void* stream) {
    return launch_rollout<false>(state, E, n, T, weights, precision, scripted_team, 0, obs, scores, rew, done, env_done, winner, env_done_t, cfg, flags,
                                 noise, actor_seed, seq, seq_base, seed, env_offset, stream);
}

int bsx_rollout_continuous(void* state, int64_t E, int n, int T, const float* weights, int precision, int scripted_team, uint64_t scripted_seed, float* obs, float* scores, float* rew,
                           uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg, uint32_t flags,
                           const BsxActorNoise* noise, uint64_t actor_seed, uint64_t seq, const uint64_t* seq_base, uint64_t seed,
                           int64_t env_offset, void* stream) {
    return launch_rollout<true>(state, E, n, T, weights, precision, scripted_team, scripted_seed, obs, scores, rew, done, env_done, winner, env_done_t, cfg, flags,
                                noise, actor_seed, seq, seq_base, seed, env_offset, stream);
}

// Self-test of atan2_pixels against the device library on the square [-R, R]^2 of argument pairs: out[0] = pairs whose 64-bit
// results differ, out[1] = pairs tested.
__global__ void bsx_selftest_atan2_kernel(int R, unsigned long long* out) {
    const int W = 2 * R + 1;
    const long long total = (long long)W * W;
    unsigned long long bad = 0, seen = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int iy = int(i / W) - R, ix = int(i % W) - R;
        const double lib = atan2(double(iy), double(ix));
        const double one = atan2_pixels(iy, ix);                       // K = 1: coefficients from the constant table
        const int ys[3] = {iy, ix, -iy}, xs[3] = {ix, iy, ix};        // K = 3: coefficients as literals (the 3v3 / 4v4 kernels)
        double three[3];
        atan2_pixels_n<3>(ys, xs, three);
        const bool ok = __double_as_longlong(one) == __double_as_longlong(lib) && __double_as_longlong(three[0]) == __double_as_longlong(lib) &&
                        __double_as_longlong(three[1]) == __double_as_longlong(atan2(double(ix), double(iy))) &&
                        __double_as_longlong(three[2]) == __double_as_longlong(atan2(double(-iy), double(ix)));
        bad += ok ? 0ull : 1ull;
        seen += 1ull;
    }
    atomicAdd(&out[0], bad); atomicAdd(&out[1], seen);
}
int bsx_selftest_atan2(int R, uint64_t* out, void* stream) {
    if (!out || R < 0 || R > 4096) return BSX_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t err = hipMemsetAsync(out, 0, 2 * sizeof(uint64_t), st);
    if (err != hipSuccess) return int(err);
    bsx_selftest_atan2_kernel<<<1024, 256, 0, st>>>(R, reinterpret_cast<unsigned long long*>(out));
    return int(hipGetLastError());
}

int bsx_observe(void* state, int64_t E, int n, float* obs, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || !obs) return BSX_E_ARG;
    if (!aligned(state, 256) || !aligned(obs, 4)) return BSX_E_ALIGN;
    // same kernel as reset with nothing selected: it only stages the poses and writes the rows (battle_env.py:202-244)
    ResetArgs a{state_ptrs(state, E, n), E, n, nullptr, nullptr, 0, 0, 0, obs, 1};
    hipLaunchKernelGGL(bsx_reset_kernel, dim3(grid_for(E, n)), dim3(TPB), 0, static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

int bsx_export_state(const void* state, int64_t E, int n, const BsxExport* out, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || !out) return BSX_E_ARG;
    if (!aligned(state, 256)) return BSX_E_ALIGN;
    ExportArgs a{state_ptrs(const_cast<void*>(state), E, n), E, n, *out, bsx_tie_tick(n)};
    const size_t EA = size_t(E) * 2 * n;
    hipLaunchKernelGGL(bsx_export_kernel, dim3(unsigned((EA + TPB - 1) / TPB)), dim3(TPB), 0,
                       static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

int bsx_instinct_discrete(const float* obs, void* actions, int out_kind, int64_t E, int n, int team, void* stream) {
    if (!obs || !actions || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || team < 0 || team > 2) return BSX_E_ARG;
    if (out_kind != BSX_ACT_I32 && out_kind != BSX_ACT_LOGITS_F32) return BSX_E_ARG;
    if (!aligned(obs, 4) || !aligned(actions, out_kind == BSX_ACT_I32 ? 4 : 16)) return BSX_E_ALIGN;
    InstinctArgs a{obs, actions, nullptr, E, n, team, out_kind, 0, 0, 0, nullptr};
    const size_t EA = size_t(E) * 2 * n;
    hipLaunchKernelGGL(bsx_instinct_kernel, dim3(unsigned((EA + TPB - 1) / TPB)), dim3(TPB), 0, static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

int bsx_instinct_continuous(const float* obs, double* actions, const double* rnd, int64_t E, int n, int team,
                            uint64_t seed, uint64_t seq, const uint64_t* seq_base, void* stream) {
    if (!obs || !actions || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N || team < 0 || team > 2) return BSX_E_ARG;
    if (!aligned(obs, 4) || !aligned(actions, 8) || (rnd && !aligned(rnd, 8))) return BSX_E_ALIGN;
    InstinctArgs a{obs, actions, rnd, E, n, team, 0, 1, seed, seq, seq_base};
    const size_t EA = size_t(E) * 2 * n;
    hipLaunchKernelGGL(bsx_instinct_kernel, dim3(unsigned((EA + TPB - 1) / TPB)), dim3(TPB), 0, static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

}  // extern "C"
