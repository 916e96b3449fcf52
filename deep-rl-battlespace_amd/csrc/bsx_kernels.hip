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

#include <mutex>
#include <unordered_map>

#include "bsx_config.h"
#include "bsx_state.h"
#include "bsx_rng.h"
#include "bsx_geometry.h"
#include "bsx_instinct.h"
#include "bsx_step_kernel.h"
#define BSX_INST_PER_CALL
#define BSX_INST_MULTI_TICK
#define BSX_INST_ROLLOUT
#include "bsx_step_split.h"                              // the two-wave 1v1 kernels
#define BSX_INST_SPLIT_MANY
#define BSX_INST_SPLIT
#define BSX_INST_SPLIT_CONT
#ifdef BSX_VARIANT
#define BSX_INST_KW
#else
#define BSX_INST_KW extern
#endif
#include "bsx_step_instances.h"
using namespace bsxk;

namespace {

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
    if (p.spawn) {                                           // injected spawn states
        if (doit) {
            const int32_t* s = p.spawn + size_t(e) * (4 + 3 * A);
            er.brx = s[0]; er.bry = s[1]; er.bbx = s[2]; er.bby = s[3];
            if (valid) { x = s[4 + 3 * a]; y = s[5 + 3 * a]; dir = double(s[6 + 3 * a]); }
        }
    } else if (!p.observe_only) {                            // Philox spawns: one block per plane, the two bases handed round the game's lanes
        const int64_t genv = p.env_offset + (env_ok ? e : p.E - 1);
        const int ac = a < A ? a : A - 1;
        const SpawnDraw sd = spawn_from_words(draw4(p.seed, genv, STREAM_RESET, uint32_t(p.nonce), uint32_t(ac)), ac, n);
        const int lane0 = (tid & 63) & ~(G - 1);             // (every lane takes part in the exchange; only the games being reset use it)
        const int rbx = __shfl(sd.bx, lane0), rby = __shfl(sd.by, lane0), bbx = __shfl(sd.bx, lane0 + n), bby = __shfl(sd.by, lane0 + n);
        if (doit) {
            er.brx = rbx; er.bry = rby; er.bbx = bbx; er.bby = bby;
            if (valid) { x = sd.x; y = sd.y; dir = double(sd.dir); }
        }
    }
    if (doit) {
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

// Which action family has advanced a state block (ABI 14: the discrete kernels keep headings as whole degrees in the plane record, the
// continuous ones as float64 beside it -- a parallel_env has one mode for its life, battle_env.py:73).  Kept on the HOST, per state
// address: nothing on the step path reads or writes device memory for it.  bsx_state_init and a bsx_reset of ALL games forget the
// family (every heading is a whole degree again); the first step / rollout call after that claims it; a call of the other family is
// refused with BSX_E_FAMILY instead of silently reading headings truncated to whole degrees.
std::mutex g_family_mu;
std::unordered_map<const void*, int> g_family;                // state -> 1 discrete, 2 continuous
inline void family_forget(const void* state) {
    std::lock_guard<std::mutex> lk(g_family_mu);
    g_family.erase(state);
}
inline bool family_claim(const void* state, bool cont) {
    std::lock_guard<std::mutex> lk(g_family_mu);
    int& f = g_family[state];
    if (f == 0) f = cont ? 2 : 1;
    return f == (cont ? 2 : 1);
}
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
// The two-wave 1v1 kernels (bsx_step_split.h); in both the first wave runs at s_setprio 1.  BSX_F_ONE_WAVE keeps the one-wave kernel.
// Multi-tick launches (bsx_step_many_discrete) of up to 65 536 games: a GAME wave and an OUTPUTS wave per 64 agents, four waves per SIMD at
// 65 536 games -- 3.05 -> 2.22 us per tick there (form 2: the outputs wave takes 16 bytes per agent and tick and repeats no game logic; launches
// of more than 32 768 games), 2.65 -> 1.75 at 32 768 (form 1: it carries the state too); beyond 65 536 the SIMDs are full of waves anyway and the
// one-wave kernel's fewer instructions win (81 920 games: 3.72 against 3.94).
// Per-call launches (bsx_step_discrete, *_range) of up to 114 688 games: a wave for everything but the observation geometry and a GEOMETRY
// wave that takes the post-move poses from it -- 6.08 -> 5.60 us per call at 65 536 games, 4.34 -> 3.99 at 4 096, 6.99 -> 6.33 at 81 920,
// 8.11 -> 7.11 at 114 688 (131 072: 8.31 -> 8.85, so not there).  Continuous actions (bsx_step_continuous) take the same form up to 81 920
// games (84 ... 88 registers: six waves per SIMD at most): 8.07 -> 7.67 us at 65 536 games, 6.26 -> 5.85 at 16 384, 9.49 -> 8.74 at 81 920 (98 304: 9.92 -> 11.95).
// Round 6: in per-call launches of up to 98 304 games (and every continuous one) the geometry wave, idle until the planes have moved, also
// computes the call's Philox block for the first wave (template parameter DRAW): 5.60 -> 5.36 us at 65 536 games, 4.01 -> 3.86 at 4 096,
// 6.55 -> 6.40 at 98 304; at seven waves per SIMD it finds no room (114 688: 7.13 -> 7.5 ... 8.1), so above 98 304 games DRAW = false.
constexpr int64_t SPLIT_MAX_GAMES = 114688, SPLIT_MANY_MAX_GAMES = 65536, SPLIT_CONT_MAX_GAMES = 81920;
constexpr int64_t SPLIT_DRAW_MAX_GAMES = 98304;            // per-call discrete launches of up to this many games: the geometry wave also computes the call's Philox block (round 6)
constexpr int64_t SPLIT_MANY_FORM2_FROM = 32768;           // multi-tick launches of MORE games than this (two workgroups on some SIMD): the outputs wave that repeats no game logic
template <bool CONT, bool MULTI>
inline bool split_applies(int n, const StepArgs& a, int64_t bound) {
    if (n != 1 || (a.flags & BSX_F_ONE_WAVE)) return false;
    if (CONT) return !MULTI && bound <= SPLIT_CONT_MAX_GAMES;   // (continuous actions: the per-call form only)
    return bound <= (MULTI ? SPLIT_MANY_MAX_GAMES : SPLIT_MAX_GAMES);
}
template <bool LG, bool OFF32, int MANY, bool CONT = false, bool DRAW = false>
void launch_split(dim3 grid, hipStream_t s, const StepArgs& a, int64_t bound) {
    hipLaunchKernelGGL((bsx_step_split_kernel<LG, OFF32, MANY, CONT, DRAW>), grid, dim3(2 * SPB), 0, s, bound, a.st.envc, a.st.envd, a.st.plane, a.actions, a.st.bent, a.st.bcnt, a.action_kind, a);
}
template <bool CONT, bool MULTI, bool LG>
void launch_for_n(int n, dim3 grid, dim3 block, hipStream_t s, const StepArgs& a, int64_t bound) {
    if constexpr (!CONT) {
        if (split_applies<CONT, MULTI>(n, a, bound)) {   // (the grid is the same: one workgroup per 64 agents, of two waves instead of one)
            const bool narrow = narrow_offsets_ok(a.E, n, a.flags);
            if constexpr (!MULTI) {
                if (bound <= SPLIT_DRAW_MAX_GAMES) {
                    if (narrow) launch_split<LG, true, 0, false, true>(grid, s, a, bound); else launch_split<LG, false, 0, false, true>(grid, s, a, bound);
                } else {                                     // seven waves per SIMD: no room beside the first waves for the geometry wave's draw
                    if (narrow) launch_split<LG, true, 0>(grid, s, a, bound); else launch_split<LG, false, 0>(grid, s, a, bound);
                }
            } else if (bound > SPLIT_MANY_FORM2_FROM) {    // two workgroups on some SIMD: the outputs wave that repeats no game logic (form 2)
                if (narrow) launch_split<LG, true, 2>(grid, s, a, bound); else launch_split<LG, false, 2>(grid, s, a, bound);
            } else {                                         // one workgroup per SIMD at most: the outputs wave that carries the state too (form 1)
                if (narrow) launch_split<LG, true, 1>(grid, s, a, bound); else launch_split<LG, false, 1>(grid, s, a, bound);
            }
            return;
        }
    } else if constexpr (!MULTI) {
        if (split_applies<CONT, MULTI>(n, a, bound)) {
            if (narrow_offsets_ok(a.E, n, a.flags)) launch_split<false, true, 0, true, true>(grid, s, a, bound);
            else launch_split<false, false, 0, true, true>(grid, s, a, bound);
            return;
        }
    }
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
    if (!family_claim(state, CONT)) return BSX_E_FAMILY;
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
    family_forget(state);
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

int bsx_state_release(void* state) {
    if (!state) return BSX_E_ARG;
    family_forget(state);
    return 0;
}

int bsx_reset(void* state, int64_t E, int n, const uint8_t* reset_mask, const int32_t* spawn, uint64_t seed,
              uint64_t nonce, int64_t env_offset, float* obs, void* stream) {
    if (!state || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N) return BSX_E_ARG;
    if (!aligned(state, 256) || (spawn && !aligned(spawn, 4)) || (obs && !aligned(obs, 4))) return BSX_E_ALIGN;
    if (!reset_mask) family_forget(state);                  // every game re-spawned: whole-degree headings, either family may follow
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
    if (!family_claim(state, CONT)) return BSX_E_FAMILY;
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
