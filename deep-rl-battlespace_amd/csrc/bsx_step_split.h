// bsx_step_split.h -- bsx_step_split_kernel: the 1v1 step() as TWO co-operating wavefronts per 64 agents; namespace bsxk.
// Part of the step() path of libbattlespace_hip.so (included after bsx_step_kernel.h; instantiated in bsx_step_two_wave.hip).  Two uses:
//   * MANY = 1 / 2: multi-tick launches (bsx_step_many_discrete) of up to 65 536 games -- 3.05 -> 2.22 us per tick at 65 536 games
//     (59 G agent-steps/s; MANY = 2), 2.65 -> 1.75 at 32 768 (MANY = 1);
//   * MANY = 0: one call per launch (bsx_step_discrete, *_range) of up to 114 688 games -- C2 6.08 -> 5.60 us; with continuous actions
//     (CONT; bsx_step_continuous, *_range) of up to 81 920 games -- 8.07 -> 7.67 us at 65 536 games.
// In both the wave that carries the game's chain runs at s_setprio 1: a SIMD's arbiter serves its resident waves oldest-first, and
// without the priority the chain's wave queues behind the wave that has slack -- the per-call form then LOSES to the one-wave kernel
// (6.70 us), the multi-tick form gains less (2.85).  DESIGN.md section 4; the forms that were built and measured against these (a planes
// wave + a bullets wave; a geometry wave that repeats classify and move; own loads per wave; other priorities): profiles/HISTORY_r05.md.
//
// The idea.  At 65 536 x 1v1 the one-wave kernel (bsx_step_kernel<1, ...>) puts two waves on a SIMD, and its tick is one long chain of
// dependent latencies: first loads -> classify -> shot (Philox, step code) -> move -> geometry -> bullet round -> resolve -> outcome ->
// stores; two lock-stepped waves hide little of each other.  But the observation geometry (the largest block of vector work) and the
// output stores need nothing from the shot or the bullets except three small counts, and the bullets need nothing from them.
//
// The multi-tick form (MANY).  The waves are persistent, so nothing is launched per tick, and ONE rendezvous per tick is enough:
//   wave 0, GAME     the whole state machine except observations and outputs: classify -> shot -> move -> pool pass -> resolve -> outcome
//                    (in registers), tick after tick; before each tick's outcome it leaves the bullets' counts in LDS (a buffer per tick parity)
//   wave 1, OUTPUTS  classify -> move -> geometry -> [rendezvous: the counts] -> outcome -> stores (rows, rewards, flags; the state after the last tick)
// Both waves carry the planes' and the game's records in registers and advance them by the same arithmetic on the same counts, so they
// never exchange state; the outputs wave runs up to a tick behind, its geometry and stores beside the game wave's next shot.  That is
// MANY = 1, for launches of up to 32 768 games: one workgroup per SIMD, a tick is a latency chain, and an outputs wave that works
// beside the game wave from the start of the tick is what shortens it.
// MANY = 2, for launches of more than 32 768 games -- two workgroups on some SIMD, the vector port is the bound: the game wave owns
// the state alone (and stores it after the last tick); per tick it PUBLISHES what the tick's outputs need (post-move position, heading,
// flags, enemy base, reward: 16 bytes per agent, a buffer per tick parity) and the outputs wave (bsx_step_split_out_body.inl) does the
// geometry, the row and the output stores from that, with none of the game logic: half the instructions in the wave that fills the
// gaps.  65 536 games: 2.55 -> 2.21 us per tick; 32 768: 1.74 -> 1.96 (hence MANY = 1 there).
//
// The per-call form (MANY = 0).  A single call has no next tick to run ahead into; the split is by what the call's chain can shed.
// Wave 0 does everything but the observation geometry; after its move it leaves the post-move poses (position, heading, enemy base:
// 16 bytes per lane) in LDS.  Wave 1, GEOMETRY, repeats no game logic: it waits for the poses, works out the geometry
// (bsx_step_split_geom_body.inl: the same phase file) and hands the four observation values per agent back; the waves meet a second time
// before the stores, wave 0 stores everything.  By size, one-wave / two-wave (round 5): 16 384 games 4.84 / 4.34 us, 32 768 5.22 / 4.80,
// 65 536 6.09 / 5.60, 81 920 6.99 / 6.33, 114 688 8.11 / 7.11; beyond that (131 072: 8.31 against 8.85) the one-wave kernel wins again.
// DRAW (round 6; launches of up to 98 304 games, and every continuous launch): until the planes have moved wave 1 has nothing to do --
// so it loads the game's record itself and computes the call's ONE Philox block per lane (the shot's jitter, or the pose of a plane whose
// game the call re-spawns: the key is in the record and the kernel's arguments), ~65 of the ~100 vector instructions of wave 0's shot,
// and leaves it in LDS before the pose rendezvous.  Wave 0 takes it there: the shot's entry and a re-spawned game's new pose are made
// AFTER the hand-over (bsx_step_phase_move.inl; nothing between the move and that point reads them), the geometry wave works a
// re-spawned plane's pose out of the block itself.  The same block from the same key: bit-identical.  65 536 games 5.60 -> 5.39 us,
// 16 384 4.36 -> 4.22, 32 768 4.84 -> 4.65, 98 304 6.55 -> 6.40, bullet-heavy play 9.64 -> 9.45, continuous 7.65 -> 7.60
// (profiles/r06_experiments.json).  At 114 688 games -- seven waves on a SIMD -- there is no room for the draw beside the first waves
// (7.13 -> 7.5 ... 8.1): above 98 304 games the launcher takes DRAW = false, the first wave draws itself.
//
// How.  No second copy of the game logic: the kernel includes the SAME phase files as bsx_step_kernel, once per wave, with the R_*
// constants of the wave's role.  The phases guard their side effects (LDS staging, stores, the pool pass, the rendezvous) by them;
// everything that only feeds a guarded-off side effect is dead code to the compiler.  Results are those of the one-wave kernels bit
// for bit: every size the launcher gives to one of these kernels is run against the C oracle (tests/test_hip_fullsize.py), and
// tests/test_hip_split.py runs them against the one-wave kernels.
#pragma once

namespace bsxk {

// <LG, OFF32, MANY, CONT>: action encoding (score rows), 32-bit offsets, the launch form -- MANY = 0 one call per launch, MANY = 1 / 2 the
// multi-tick forms -- and continuous actions (per call only); all described above.
// DRAW (per call only): the geometry wave also computes the call's Philox block (below); without it the first wave draws itself, as in round 5.
template <bool LG, bool OFF32, int MANY = 0, bool CONT_ = false, bool DRAW = false>
__global__ __launch_bounds__(2 * SPB)
void bsx_step_split_kernel(const int64_t E_, const uint2* const envc_, const uint2* const envd_, const uint2* const plane_, const void* const act_,
                           const uint2* const bent_, const uint32_t* const bcnt_, const int kind_, const StepArgs p_) {
    constexpr int N = 1;
    constexpr bool CONT = CONT_, MULTI = MANY != 0, ACTOR = false;
    // continuous actions (bsx_step_continuous): the per-call form only -- its geometry wave needs nothing but the poses, so only the
    // first wave's loads differ (the action triple by encoding, the float64 heading beside the plane record)
    static_assert(!CONT || (MANY == 0 && !LG), "continuous actions: the per-call form only");
    static_assert(!DRAW || MANY == 0, "the geometry wave's draw: the per-call form only");
    const StepArgs& p = p_;
    typedef typename std::conditional<OFF32, uint32_t, size_t>::type ix_t;     // row / element offsets
    typedef typename std::conditional<OFF32, int32_t, int64_t>::type ixs_t;    // game indices
    constexpr bool NT_STATE = false;
    constexpr int WAVES = 1;                             // LDS is sized for ONE set of 64 agents: the two waves share it
    const int n = 1, A = 2, G = 2, EPB = SPB / 2;
    const int wave = 0;                                  // (the phases' LDS offsets: both waves use the set's only slice)
    const int role_wave = int(threadIdx.x >> 6);         // the wave's role: see the forms above
    const unsigned stamp_row = blockIdx.x * 2u + unsigned(role_wave); (void)stamp_row;   // (diagnostic builds: a row of stamps per wave)
    STAMP(8);
    STAMP_HWID();
    const int tid = int(threadIdx.x & 63);
    const ixs_t wblk = ixs_t(blockIdx.x);
    const int a = tid & (G - 1);
    const ixs_t e = wblk * EPB + (tid / G);
    const bool env_ok = e < ixs_t(E_);
    const bool valid = env_ok && a < A;
    const ix_t EA = ix_t(MULTI ? E_ : p.E) * ix_t(A);
    const ixs_t ec = env_ok ? e : ixs_t(E_ - 1);
    const ix_t g = ix_t(ec) * A + (a < A ? a : A - 1);
    constexpr int NE = 1;
    constexpr int DROW = 3 * N + 2;
    __shared__ volatile int s_x_all[SPB], s_y_all[SPB], s_hp_all[SPB];
    __shared__ volatile int s_bhit_all[SPB];
    __shared__ __attribute__((aligned(16))) float s_obs_all[4];
    __shared__ __attribute__((aligned(16))) float s_small[4];
    __shared__ int s_act_all[1];
    __shared__ float s_actf_all[1];
    __shared__ double s_actd_all[1];
    __shared__ int s_gdone_all[1];
    __shared__ float s_pd_all[1];
    __shared__ double s_pr_all[1];
    __shared__ __attribute__((aligned(8))) u32x2 s_new_all[SPB];
    __shared__ uint32_t s_agg_all[SPB];
    __shared__ uint32_t s_npl_all[2 * SPB];              // the bullets' counts per shooter (misses | base hits << 8 | plane hits << 16), by tick parity
    __shared__ __attribute__((aligned(16))) v4u_t s_t0_all[MANY == 0 ? SPB : 1];  // per call: the post-move poses (position, heading, enemy base), wave 0 -> geometry wave
    __shared__ __attribute__((aligned(16))) v4u_t s_pub_all[MANY == 2 ? 2 * SPB : 1];   // MANY = 2: what a tick's outputs need, game wave -> outputs wave, by tick parity
    __shared__ __attribute__((aligned(16))) v4f_t s_gm_all[SPB];                  // per call: the four observation values, geometry wave -> storing wave
    __shared__ __attribute__((aligned(16))) v4u_t s_rw_all[DRAW ? SPB : 1];  // per call: the call's Philox block per lane (jitter or re-spawn), geometry wave -> first wave
    constexpr bool CORNERS = true;
    typedef u32x2 rect_t;
    __shared__ __attribute__((aligned(8))) rect_t s_eb_all[SPB];
    __shared__ __attribute__((aligned(8))) rect_t s_pq_all[SPB];
    auto make_rect = [](uint32_t c, bool alive, int xl, int yl, int xh, int yh) {
        return alive ? u32x2{c + pk_const(PK_BIAS - xl, PK_BIAS - yl), c + pk_const(PK_BIAS + xh, PK_BIAS + yh)} : u32x2{0x7F007F00u, 0u};
    };
    auto hits_rect = [](s16x2 b, rect_t r, int xl, int yl, int xh, int yh) {
        return ~pk_any_negative(pk_bits(b - as_pk(r.x)) | pk_bits(as_pk(r.y) - b));
    };
    constexpr uint32_t OWN_PHYS = 16u, OWN_DROP = 32u;
    __shared__ uint32_t s_fl_all[SPB];
    __shared__ __attribute__((aligned(16))) double s_nd_all[SPB * 2];
    constexpr int FW = 4, OW = 1;
    __shared__ unsigned long long s_ov_all[SPB * OW];
    __shared__ uint16_t s_pp_all[SPB * K];
    auto* const s_new = BSX_LDS(u32x2, s_new_all);
    auto* const s_agg = BSX_LDS(uint32_t, s_agg_all);
    auto* const s_npl = BSX_LDS(uint32_t, s_npl_all);
    auto* const s_gm = (__attribute__((address_space(3))) volatile v4f_t*)(uintptr_t)(s_gm_all);
    auto* const s_pub = (__attribute__((address_space(3))) volatile v4u_t*)(uintptr_t)(s_pub_all);
    auto* const s_t0 = (__attribute__((address_space(3))) volatile v4u_t*)(uintptr_t)(s_t0_all);
    auto* const s_rw = (__attribute__((address_space(3))) volatile v4u_t*)(uintptr_t)(s_rw_all);
    auto* const s_eb = BSX_LDS(rect_t, s_eb_all);
    auto* const s_pq = BSX_LDS(rect_t, s_pq_all);
    auto* const s_fl = BSX_LDS(uint32_t, s_fl_all);
    auto* const s_nd = BSX_LDS(double, s_nd_all);
    auto* const s_ov = BSX_LDS(unsigned long long, s_ov_all);
    auto* const s_pp = BSX_LDS(uint16_t, s_pp_all);
    float* const s_pd = s_pd_all;
    double* const s_pr = s_pr_all;
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint* const s_x = (lds_vint*)(uintptr_t)(s_x_all);
    lds_vint* const s_y = (lds_vint*)(uintptr_t)(s_y_all);
    lds_vint* const s_hp = (lds_vint*)(uintptr_t)(s_hp_all);
    lds_vint* const s_bhit = (lds_vint*)(uintptr_t)(s_bhit_all);
    float* const s_obs = s_obs_all;
    (void)s_small; (void)s_act_all; (void)s_actf_all; (void)s_actd_all; (void)s_gdone_all; (void)s_obs; (void)s_pd; (void)s_pr;
    (void)s_x; (void)s_y; (void)s_hp; (void)s_bhit; (void)DROW; (void)wave; (void)EA;
    // the two waves of a workgroup meet: every LDS operation of this wave has landed (lgkmcnt) before it arrives, nothing of it is
    // reordered across (the LDS arrays are volatile; the asm is a compiler fence).  Global stores in flight (the shot's heading in the
    // export ring, pool entries) are NOT waited for: nothing the other wave reads travels through global memory.
    auto split_rendezvous = [] {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    const bool has_act = kind_ >= 0;
    // a call's inputs: the action (int32 or four scores) and, when the caller injects it, the shot's random() value -- fetched a tick ahead in
    // the multi-tick form (as bsx_step_kernel does)
    struct RawIn { int ai; float4 lg; double uu; };
    struct DecIn { int act; double uu; };
    auto load_inputs = [&](int t, RawIn& r) {
        const char* const abase = static_cast<const char*>(act_) + (MULTI ? int64_t(t) * p.act_tb : int64_t(0));
        if constexpr (LG) r.lg = *reinterpret_cast<const float4*>(elem(abase, has_act ? g * 16 : ix_t(0)));
        else r.ai = *reinterpret_cast<const int32_t*>(elem(abase, has_act ? g * 4 : ix_t(0)));
        __builtin_amdgcn_sched_barrier(0);
        const double* const ut = (MULTI && p.u) ? p.u + int64_t(t) * p.u_ts : p.u;
        if (ut) r.uu = ut[g];
    };
    auto decode = [&](const RawIn& r) {
        DecIn d = {-1, 0.0};
        if (has_act) d.act = LG ? argmax4(r.lg.x, r.lg.y, r.lg.z, r.lg.w) : r.ai;
        if (p.u) d.uu = r.uu;
        return d;
    };
    const ix_t pool0 = ix_t(wblk) * ix_t(POOL_CAP);
    // (the roles: the header of this file)
    if constexpr (MANY != 0) {
        if (role_wave == 0) {
            __builtin_amdgcn_s_setprio(1);               // the game wave's tick sets the pace: it goes first at the SIMD's ports
            constexpr bool R_BULLETS = true, R_MOVE = true, R_STAGE = true, R_GEOM = false, R_OUTCOME = true, R_POSE_LDS = false;
            constexpr bool R_ST_STATE = MANY == 2, R_ST_OUT = false;
            constexpr int R_RDV_COUNTS = MANY == 2 ? 0 : 1, R_GEOM_LDS = 0, R_PUB = MANY == 2 ? 1 : 0, R_DRAW_LDS = 0;
            s_ov[tid] = 0ull;                            // (cleared again by whoever finds it set)
#include "bsx_step_split_many_body.inl"
        } else if constexpr (MANY == 2) {                // (the outputs wave repeats none of the game logic, the game wave publishes 16 bytes per agent and tick)
            constexpr bool R_BULLETS = false, R_GEOM = true, R_ST_STATE = false, R_ST_OUT = true;
            constexpr int R_GEOM_LDS = 0, R_PUB = 0;
#include "bsx_step_split_out_body.inl"
        } else {
            constexpr bool R_BULLETS = false, R_MOVE = true, R_STAGE = false, R_GEOM = true, R_OUTCOME = true, R_POSE_LDS = false;
            constexpr bool R_ST_STATE = true, R_ST_OUT = true;
            constexpr int R_RDV_COUNTS = 2, R_GEOM_LDS = 0, R_PUB = 0, R_DRAW_LDS = 0;
#include "bsx_step_split_many_body.inl"
        }
    } else {
        constexpr int R_PUB = 0;
        const int tk = 0;
        if (role_wave == 0) {
            __builtin_amdgcn_s_setprio(1);               // this wave's chain is the call's: it goes first at the SIMD's ports
            constexpr bool R_BULLETS = true, R_MOVE = true, R_STAGE = true, R_GEOM = false, R_OUTCOME = true, R_ST_STATE = true, R_ST_OUT = true;
            constexpr bool R_POSE_LDS = true;
            constexpr int R_RDV_COUNTS = 0, R_GEOM_LDS = 2, R_DRAW_LDS = DRAW ? 2 : 0;
            s_ov[tid] = 0ull;                            // (cleared again by whoever finds it set)
#include "bsx_step_split_body.inl"
        } else {
#include "bsx_step_split_geom_body.inl"
        }
    }
}

}  // namespace bsxk
