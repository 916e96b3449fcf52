// bsx_step_kernel.h -- bsx_step_kernel: ONE fused launch per step() (and its multi-tick / fused-rollout forms)
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h) and by the three translation units that instantiate the step kernels; namespace bsxk.
#pragma once

namespace bsxk {

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
// (the multi-tick 2v2 kernels are compiled for four waves per SIMD: 65 536 games are 4 096 of their waves = four per SIMD, and at 129 ... 135
//  registers -- three resident waves -- the launch ran as two rounds: 6.9 us per tick against 6.2 for round 3's 125-register kernel)
template <int N, bool CONT, bool MULTI, bool ACTOR = false, bool LG = false, bool OFF32 = false>
__global__ __launch_bounds__(SPB * (ACTOR ? group_width(N > 0 ? N : 1) / 2 : WPB)) __attribute__((amdgpu_waves_per_eu((ACTOR && N > 1) ? 2 : ((!ACTOR && MULTI && N == 2) ? 4 : 1))))
void bsx_step_kernel(const int64_t E_, const uint2* const envc_, const uint2* const envd_, const uint2* const plane_, const void* const act_,
                     const uint2* const bent_, const uint32_t* const bcnt_, const int kind_, const StepArgs p_) {
    const StepArgs& p = p_;                              // (the tick loop of the multi-tick forms shadows this name: see there)
    // The eight leading arguments repeat p.E, p.st.envc, p.st.envd, p.st.plane, p.actions, p.st.bent, p.st.bcnt, p.action_kind: fifteen
    // dwords that the dispatcher preloads into SGPRs (-amdgpu-kernarg-preload-count), so that a wave's first loads need nothing
    // from the kernarg segment and do not queue behind its cold scalar-cache fetch.
    const unsigned stamp_row = blockIdx.x; (void)stamp_row;   // (diagnostic builds: where this wave's stamps go)
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
    // 4v4 and the runtime-n kernel: the centre (x | alive << 15 | y << 16) and the margins as constants in the slot -- measured faster there
    // (round 3: 4v4 23.1 us against 24.7 with corners; round 4, with the table shot: 20.5 against 20.8).  2v2 / 3v3 take the corner form
    // since round 4 (with the table shot 10.22 -> 9.67 and 17.97 -> 17.05 us; profiles/r04_experiments.json).
    constexpr bool CORNERS = N >= 1 && N <= 3;
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
    // lane of a game computes the same record), the pool's length -- so a later tick starts with its pool loads instead of a
    // state round trip, and the inputs of tick t+1 are fetched while tick t computes.  Pool entries go to memory every tick
    // (and the win / tie counters at a game's end), the plane and game records and the pool's length once, after the last one.
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
    // The multi-tick kernels of larger teams likewise (round 5): with ~20 ... 35 hoisted LDS addresses the 2v2 kernels that take score rows
    // or continuous actions (compiled for four waves per SIMD = 128 registers) kept 2 ... 7 registers in scratch memory, and 3v3 / 4v4
    // stood at 135 ... 161 registers = three resident waves; recomputed per tick: 2v2 93 ... 106 without scratch, 3v3 / 4v4 104 ... 122 =
    // four waves (per tick, same box: 3v3 12.19 -> 11.56 us, 4v4 16.9 -> 16.8, 2v2 continuous 8.80 -> 8.53, 4v4 continuous 21.5 -> 19.7).
    // Not 1v1 (3.06 -> 3.10) and not 2v2 with int32 actions (5.03 -> 5.31: that kernel fitted as it was).  OPAQUE_MULTI_MASK (bsx_config.h): bit n =
    // the n-v-n kernels with int32 actions, bit 8 + n = those with score rows or continuous actions.
    if constexpr ((ACTOR && N > 1) || (MULTI && !ACTOR && N > 0 && ((OPAQUE_MULTI_MASK >> (((LG || CONT) ? 8 : 0) + N)) & 1))) asm volatile("" : "+v"(tid_k));
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
        pool_first = *elem(bent_, pool0 + ix_t(lane));
        if (!(DIAG & 2u)) pc = __builtin_amdgcn_readfirstlane(*elem(bcnt_, ix_t(wblk)));
        // (the loads above need nothing but preloaded kernel arguments; load_inputs ends in a branch -- the injected jitter -- behind which
        //  the compiler's wait-count pass drains everything in flight, so it comes last)
        if constexpr (CONT) dirf = *elem(p.st.pdirf, gt);
        if (!MULTI) load_inputs(0, rin);
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
    // ---- the tick, phase by phase (each file opens with what it reads and writes; they share the locals above and each other's):
    //   actor     fused rollout only: this tick's actions from the actors' MFMA pass over the wave's LDS observation rows
    //   shot      heading-table gather; call mode per game (inert / re-spawn / tie / physics); ONE Philox block (jitter or re-spawn); the shot
    //             as a pool entry queued in LDS; owner flags and the enemy base staged for the work slots
    //   move      re-spawn or process_action's move; post-move poses to the game's other planes (DPP / LDS); sprites staged as rectangles
    //   geometry  range / angle-off to the enemy base and planes in binary64 (the observation row's values)
    //   bullets   Bullet.update for every work slot (pool entries + queued shots), pool compaction, ordered plane-hit resolve
    //   outcome   rewards, deaths, base hit points, win / tie
    //   stores    plane / game records, reward, done, observation row, counters, pool length
    // R_*: which parts of the tick this wave runs -- here all of them.  The two-wave 1v1 kernels (bsx_step_split.h: the product's kernels for
    // discrete 1v1 launches up to 114 688 games per call / 65 536 per multi-tick launch, continuous up to 81 920 per call) include the same
    // phase files once per wave with different parts switched on; the phases guard their side effects by these constants and the rest falls
    // to dead-code elimination.  With every part on, every guard is a compile-time `true` (the ISA of all 76 kernels is unchanged by the guards).
    constexpr bool R_BULLETS = true, R_MOVE = true, R_STAGE = true, R_GEOM = true, R_OUTCOME = true, R_ST_STATE = true, R_ST_OUT = true, R_POSE_LDS = false;
    constexpr int R_RDV_COUNTS = 0, R_GEOM_LDS = 0, R_PUB = 0, R_DRAW_LDS = 0;
    auto split_rendezvous = [] {};                       // (names of the split kernel's role code: never reached here)
    uint32_t* const s_npl = nullptr;
    v4f_t* const s_gm = nullptr;
    v4u_t* const s_t0 = nullptr; v4u_t* const s_pub = nullptr; v4u_t* const s_rw = nullptr;
#include "bsx_step_phase_actor.inl"
#include "bsx_step_phase_shot.inl"
#include "bsx_step_phase_move.inl"
#include "bsx_step_phase_geometry.inl"
#include "bsx_step_phase_bullets.inl"
#include "bsx_step_phase_outcome.inl"
#include "bsx_step_phase_stores.inl"
    STAMP(7);
    if (MULTI) {
        games += uint32_t(cnt_delta.x);
        // The only memory one tick hands to the next is this WAVE's own: its pool entries (stored by one lane, loaded by another of the
        // same wave next tick) plus its LDS rows.  A wavefront's vector memory operations are performed in order through the one L1 of
        // its CU, whatever the lane, so wavefront scope is enough: a compiler ordering point, no s_waitcnt -- this tick's stores
        // (observation rows included) drain while the next tick computes.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(9);
    }
    }   // tick loop
}

}  // namespace bsxk
