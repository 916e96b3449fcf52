/* battlespace_hip.h -- C ABI of libbattlespace_hip.so: the batched Battlespace step() path on MI355X (gfx950).
 *
 * The reference (WilliamFlinchbaugh/Deep-RL-Battlespace) has no FFI: its hot path is the Python class
 * `parallel_env` (envs/battle_env.py:61).  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md); each entry point names the reference code it replaces.  Everything is plain C: device pointers,
 * sizes, a hipStream_t passed as void*.  No torch types.
 *
 * Conventions
 *   - E = number of independent games (envs), n = planes per team, A = 2n agents per env, D = 3n+2 observation floats.
 *   - Agent a of env e is row e*A + a of every per-agent array; a < n is red ("plane{a}"), a >= n blue -- the order
 *     of `possible_agents` (battle_env.py:106-108).
 *   - All pointers are DEVICE-ACCESSIBLE memory owned by the caller: device memory for batches; for a single game
 *     (the drop-in surface: a few dozen bytes per call) pinned host memory (hipHostMalloc, mapped at the same address)
 *     works as well and saves both copies.  Calls only enqueue work on `stream` and return; they never synchronise
 *     (bsx_stream_synchronize is the one exception, by name) and never allocate.
 *     Return value: 0 = ok, <0 = argument error (BSX_E_*), >0 = a hipError_t.
 *   - Outputs are overwritten by the next call; inputs are read-only for the duration of the kernel.
 *   - Supported n: 1..16.
 */
#ifndef BATTLESPACE_HIP_H
#define BATTLESPACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSX_ABI_VERSION 14
#define BSX_BULLET_SLOTS 12 /* a bullet is removed at the latest on its 12th update (sprites.py:334-337: 12*45 >= 500) */
#define BSX_MAX_N 16
#define BSX_MAX_E (INT64_C(1) << 30)   /* games per call: keeps every grid below 2^31 workgroups; 2^30 games of 1v1 is ~0.8 TB of state */

#define BSX_E_ARG (-1)     /* null pointer / bad size / unsupported n */
#define BSX_E_ALIGN (-2)   /* a pointer is not aligned as documented */
#define BSX_E_FAMILY (-3)  /* a discrete call on a state the continuous kernels have advanced, or the reverse (see bsx_step_continuous) */

/* winner codes (battle_env.py:254,474,490: 'none' | 'red' | 'blue' | 'tie') */
#define BSX_WINNER_NONE 0
#define BSX_WINNER_RED 1
#define BSX_WINNER_BLUE 2
#define BSX_WINNER_TIE 3

/* Constructor kwargs of parallel_env that reach step() (battle_env.py:73, :178-182). */
typedef struct BsxRewards {
    double hit_base_reward;   /* default 100 */
    double hit_plane_reward;  /* default 10  */
    double miss_punishment;   /* default -1  */
    double die_punishment;    /* default -5  */
    double lose_punishment;   /* default -20 */
} BsxRewards;

/* Flags for bsx_step_* */
#define BSX_F_AUTO_RESET 1u  /* a call on a finished env re-spawns it (Philox) instead of the inert step of battle_env.py:303-306 */
#define BSX_F_EMPTY_CALL 2u  /* step({}) : every running env ties (battle_env.py:309-313) */
#define BSX_F_ONE_WAVE 8u     /* keep the one-wave step kernel where the library would take a two-wave form of it (csrc/bsx_step_split.h): discrete 1v1 launches -- multi-tick ones of up to 65 536 games (bsx_step_many_discrete: a game wave + an outputs wave per 64 agents), per-call ones of up to 114 688 games (bsx_step_discrete, _range: a wave for everything but the observation geometry + a geometry wave) -- and continuous 1v1 per-call launches of up to 81 920 games (bsx_step_continuous, _range: the same form).  Same results either way -- for the tests that run the two against each other, and for A/B runs */
#define BSX_F_WIDE_OFFSETS 4u /* take the 64-bit-offset kernels although the job's arrays stay below 4 GB (they are chosen automatically above that; same results -- for tests) */

/* Action encodings for bsx_step_discrete */
#define BSX_ACT_I32 0        /* int32 [E*A]: 0 fwd, 1 shoot, 2 left, 3 right, anything else = no movement (battle_env.py:399-417) */
#define BSX_ACT_LOGITS_F32 1 /* float32 [E*A*4]: flat argmax, first maximum wins (battle_env.py:327-328) */
/* Action encodings for bsx_step_continuous: [E*A*3] = speed, turn, shoot in [-1,1], clipped in-kernel (battle_env.py:295-297) */
#define BSX_ACT_F32 0
#define BSX_ACT_F64 1
#define BSX_ACT_F32X4 2      /* float32 [E*A*4], 16-byte aligned: speed, turn, shoot + one ignored float -- the 4-wide rows bsx_actor_forward writes */

int bsx_abi_version(void);

/* 0 for the product build.  Non-zero = a diagnostic build whose RESULTS ARE NOT THE REFERENCE'S (timing ablations compiled
 * with -DBSX_DIAG=<bits>: bits 0-3; in-kernel phase stamps -DBSX_STAMPS: bit 8).  A binding must refuse such a library
 * unless the caller asked for it explicitly (deep-rl-battlespace_amd/_lib.py: BSX_ALLOW_DIAG=1). */
int bsx_build_flags(void);

/* Size in bytes of the opaque per-job state block for E envs of n-per-team (256-byte aligned base required).
 * Holds what parallel_env holds between calls (battle_env.py:165-184,254-276): planes, bases, bullets, time, flags,
 * win/tie counters, plus the 361-entry discrete-heading displacement table.  Layout (ABI 14): 8-byte plane and game records, the
 * base positions in an array of their own, and one dense bullet pool per block of 64 lanes (= 64 / G games, G = the next power of two
 * >= 2n): what a step() moves is what the game holds, not the 12 bullet slots per plane the reference's objects reserve. */
int bsx_state_bytes(int64_t E, int n, size_t* bytes);

/* One-time initialisation of a state block: zeroes it and uploads the heading table
 * (21.5*cos(-radians(d)), 21.5*sin(-radians(d)), d = 0..360, host libm: sprites.py:35-42 with speed*time = 215*0.1).
 * Counters start at 0.  Every env starts "finished"; call bsx_reset before stepping. */
int bsx_state_init(void* state, int64_t E, int n, void* stream);

/* parallel_env.close (battle_env.py:457): the caller is about to free (or re-use for something else) a state block.  Forgets what the
 * library keeps on the HOST per state address -- which action family has advanced the block (see bsx_step_continuous) -- so that a
 * long-lived process that cycles allocations does not accumulate entries, and a later allocation at the same address does not inherit
 * the claim.  bsx_state_init is mandatory on every (re)allocated block anyway and forgets the claim too; this call is for the
 * address that is NOT initialised again.  Touches no device memory; 0, or BSX_E_ARG for a null pointer. */
int bsx_state_release(void* state);

/* parallel_env.reset (battle_env.py:246-279; Plane.reset sprites.py:74-91; Base.reset sprites.py:238-252).
 *   reset_mask  nullable uint8[E]: only envs with a non-zero byte are reset (null = all).
 *   spawn       nullable int32[E][4+3A]: base_red x,y, base_blue x,y, then x,y,dir per plane in id order (parity runs
 *               inject what the reference drew); null = draw in-kernel from Philox4x32-10 keyed by
 *               (seed, env_offset+e, nonce) with the reference's inclusive ranges.
 *   obs         nullable float32[E*A*D]: reset observations (written for every env, reset or not).
 * Counters (games, ties, wins) persist across resets as in the reference. */
int bsx_reset(void* state, int64_t E, int n, const uint8_t* reset_mask, const int32_t* spawn,
              uint64_t seed, uint64_t nonce, int64_t env_offset, float* obs, void* stream);

/* parallel_env.step, discrete actions (battle_env.py:281-381, process_action :383-417).
 *   actions     per `action_kind`
 *   u           nullable float64[E*A]: the random.random() value for agent a's shot this call (sprites.py:314);
 *               null = Philox keyed by (seed, env_offset+e, games played, tick, agent).
 *   obs, rew, done  float32[E*A*D], float32[E*A], uint8[E*A]
 *   env_done, winner nullable uint8[E] copies of the per-env flags (battle_env.py:276,254)
 */
int bsx_step_discrete(void* state, int64_t E, int n, const void* actions, int action_kind, const double* u,
                      float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner,
                      const BsxRewards* cfg, uint32_t flags, uint64_t seed, int64_t env_offset, void* stream);

/* parallel_env.step, continuous actions (battle_env.py:295-297, :418-424).
 * A state block belongs to ONE action mode for its life, as a parallel_env does (battle_env.py:73 `continuous_actions`): the discrete
 * kernels keep headings as whole degrees inside the 8-byte plane record (reset spawns and the 15-degree turns are whole degrees), the
 * continuous kernels keep them as float64 beside it.  bsx_state_init and a bsx_reset of ALL games (reset_mask == NULL) may be followed
 * by either family; the first step / rollout call after that claims the block for its family, and a call of the OTHER family on it is
 * refused with BSX_E_FAMILY (it would read headings truncated to whole degrees).  The claim is kept on the host per state address --
 * nothing on the step path touches device memory for it; a block whose bytes the caller copied elsewhere is unclaimed there, a
 * snapshot loaded into a block takes the block's claim (the Python surface checks a snapshot's action mode itself: state_dict meta);
 * bsx_state_release drops the claim when the block is freed. */
int bsx_step_continuous(void* state, int64_t E, int n, const void* actions, int action_kind, const double* u,
                        float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner,
                        const BsxRewards* cfg, uint32_t flags, uint64_t seed, int64_t env_offset, void* stream);

/* The same call for the games [first, first + count) of the state only (battle_env.py:281-381: the reference's games share nothing,
 * so any partition of the batch steps to the same results).  Every array argument is the FULL array, exactly as bsx_step_discrete /
 * bsx_step_continuous take it; rows outside the range are neither read nor written.  What it is for: a caller with all T calls'
 * actions at hand enqueues the batch as a few independent chains of launches (one per range, each on its own stream, or as parallel
 * branches of one HIP graph), so that one chain's kernel boundary -- launch, first loads, store drain -- runs under another chain's
 * arithmetic (`parallel_env.capture_steps(..., chains=)`).  `first` must be a multiple of 256 (BSX_E_ARG otherwise: sub-arrays keep the
 * alignment of the full ones); count >= 1, first + count <= E.  Launches over disjoint ranges may run concurrently. */
int bsx_step_discrete_range(void* state, int64_t E, int n, int64_t first, int64_t count, const void* actions, int action_kind,
                            const double* u, float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner,
                            const BsxRewards* cfg, uint32_t flags, uint64_t seed, int64_t env_offset, void* stream);
int bsx_step_continuous_range(void* state, int64_t E, int n, int64_t first, int64_t count, const void* actions, int action_kind,
                              const double* u, float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner,
                              const BsxRewards* cfg, uint32_t flags, uint64_t seed, int64_t env_offset, void* stream);

/* T consecutive parallel_env.step calls in ONE launch (the caller's `for t: step(actions[t])` loop, battle_env.py:281, when
 * the actions of all T calls are known up front: scripted or random play, replays).  Same arguments and results as T calls
 * of bsx_step_discrete / bsx_step_continuous, bit for bit; arrays gain a leading T axis: actions [T][E*A] (or [T][E*A*4],
 * [T][E*A*3]), u [T][E*A] (nullable).  store_all != 0: obs [T][E*A*D], rew [T][E*A], done [T][E*A] hold every call's
 * results; store_all == 0: obs / rew / done are [E*A...] and hold the LAST call's (each call still writes them).
 * env_done / winner (nullable, [E]) are the state after the last call; env_done_t (nullable, uint8 [T][E]) receives env_done
 * after EVERY call, so a consumer can tell which of the T rows of a game belong to a running game (a call on a finished game is
 * the reference's inert call, battle_env.py:303-306, or -- with BSX_F_AUTO_RESET -- the re-spawn: neither is a transition).
 * 1 <= T <= BSX_MAX_T.  A wavefront walks its games through the T calls, so between calls the state stays in the L2. */
#define BSX_MAX_T 65535
int bsx_step_many_discrete(void* state, int64_t E, int n, int T, const void* actions, int action_kind, const double* u,
                           float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t,
                           const BsxRewards* cfg, uint32_t flags, int store_all, uint64_t seed, int64_t env_offset,
                           void* stream);
int bsx_step_many_continuous(void* state, int64_t E, int n, int T, const void* actions, int action_kind, const double* u,
                             float* obs, float* rew, uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t,
                             const BsxRewards* cfg, uint32_t flags, int store_all, uint64_t seed, int64_t env_offset,
                             void* stream);

/* parallel_env.observe for every agent (battle_env.py:202-244): obs float32[E*A*D].  No state change. */
int bsx_observe(void* state, int64_t E, int n, float* obs, void* stream);

/* Unpacked copy of the game state, for tests / checkpointing / rendering one env on the host.
 * Any pointer may be null.  Shapes: px,py,php int32[E*A]; pdir float64[E*A]; palive uint8[E*A];
 * base_xy int32[E*4] (red x,y, blue x,y); bhp int32[E*2]; tick int32[E]; env_done,winner uint8[E];
 * bl_live uint8[E*A*12]; bl_x,bl_y int32[E*A*12]; bl_dir float64[E*A*12] (slot = birth tick % 12);
 * counters int32[E*4] = games, ties, red wins, blue wins (battle_env.py:169-170,102-103). */
typedef struct BsxExport {
    int32_t* px; int32_t* py; double* pdir; int32_t* php; uint8_t* palive;
    int32_t* base_xy; int32_t* bhp; int32_t* tick; uint8_t* env_done; uint8_t* winner;
    uint8_t* bl_live; int32_t* bl_x; int32_t* bl_y; double* bl_dir; int32_t* counters;
} BsxExport;
int bsx_export_state(const void* state, int64_t E, int n, const BsxExport* out, void* stream);

/* ---- on-device policy rollout (BASELINE.json configs[4]): the fused per-agent actor in front of bsx_step_discrete.
 * Replaces, on the rollout path, ActorNetwork.forward + noise + clamp of the reference's callers
 * (maddpg/networks.py:81-85, maddpg/agent.py:25-33): obs[3n+2] -> 64 -> LayerNorm -> ReLU -> 64 -> LayerNorm -> ReLU ->
 * 4 -> tanh [-> + N(0, noise_std) -> clamp(-1,1)], one independent weight set per agent a = 0..2n-1.
 *   weights  float32, 16-byte aligned, 2n blobs of bsx_actor_blob_floats(3n+2) floats each, packed for the MFMA fragments
 *            of the transposed product H^T = W^T X^T (f32 MFMA 32x32x2: exact f32).  With Dp = D rounded up to even and
 *            nid(m, v, hh) = 32m + (v&3) + 8(v>>2) + 4hh (the neuron held by accumulator register v of tile m, lane half hh):
 *              W1A[mo 2][s Dp/2][lane 64]          = W1[2s + (lane>>5)][32mo + (lane&31)], 0 beyond D
 *              W2A[mo 2][mt 2][vq 4][lane 64][t 4] = W2[nid(mt, 4vq + t, lane>>5)][32mo + (lane&31)]
 *              b1 ln1_gain ln1_bias b2 ln2_gain ln2_bias, each [hh 2][mo 2][v 16] = vec[nid(mo, v, hh)]
 *              W3P[hh 2][mt 2][v 16][4] = W3[nid(mt, v, hh)][0..3];  b3[4]
 *              W2B[mo 2][s 4][term 3][lane 64][i 8] bfloat16 (6144 floats' worth): term 0 = bf16(W2), term 1 = bf16(W2 - term 0),
 *                  term 2 = bf16(W2 - term 0 - term 1), of W2[nid(s>>1, 8(s&1) + i, lane>>5)][32mo + (lane&31)]  -- read only with
 *                  BSX_ACTOR_BF16X3 (terms 0-1) / BSX_ACTOR_BF16X6 (all three)
 *            (W[k][j] multiplies input k into output j, i.e. the transpose of torch's Linear.weight.)
 *   obs      float32[E*A*D] (what bsx_step_* / bsx_reset wrote);  scores float32[E*A*4], 16-byte aligned: feed it to
 *            bsx_step_discrete with BSX_ACT_LOGITS_F32.
 *   noise    nullable.  Exploration noise added to the tanh outputs, then clamp(-1, 1) (maddpg/agent.py:30-31): Gaussian
 *            and/or the reference's Ornstein-Uhlenbeck process (utils/noise.py:4-21).  Normal draws are Philox-keyed by
 *            (seed, seq + *seq_base, GLOBAL row (env_offset + e)*A + a), so a job's noise does not depend on how its games are
 *            sharded; the Gaussian term takes a second, independent draw when both processes are on.  Pass a new seq per call,
 *            or -- inside a captured HIP graph, whose arguments are frozen -- a device word seq_base (nullable) that the graph
 *            itself advances once per replay.  z_inject replaces the draws by the caller's normals (parity runs against
 *            utils/noise.py with the reference's own np.random.randn values). */
typedef struct BsxActorNoise {
    float gaussian_std;       /* > 0: scores += N(0, gaussian_std) */
    float ou_scale;           /* > 0: x += ou_theta*(ou_mu - x) + ou_sigma*N(0,1);  scores += ou_scale * x   (utils/noise.py:17-21) */
    float ou_theta, ou_sigma, ou_mu;   /* reference defaults 0.15, 0.2, 0 */
    float* ou_state;          /* float32[E*A*4] process state x, required when ou_scale > 0 */
    const uint8_t* env_done;  /* nullable uint8[E]: rows of finished games restart from ou_mu (main.py:155 reset_noise per game) */
    const float* z_inject;    /* nullable float32[E*A*4], 16-byte aligned: standard normals to use instead of the Philox draws
                                 (bsx_actor_forward only; one set per call, used by the OU and the Gaussian term alike) */
    int ou_keep;              /* bsx_rollout_*: 0 = the process restarts from ou_mu at every game start, as the training loop does
                                 (main.py:155 reset_noise per game); 1 = it never restarts -- the reference's evaluation loop
                                 (evaluate.py:52-76) never calls reset_noise.  bsx_actor_forward restarts exactly where env_done says. */
    /* ---- policy-gradient rollouts (BASELINE.json configs[4] words C5 as a "PPO policy rollout"; README.md:13 mentions the PPO runs
     * that preceded MADDPG, of which the reference keeps no code): a stochastic policy head and a value head next to the
     * deterministic-plus-noise one of maddpg/agent.py:25-33.  All fields may be zero / null. */
    int sample_mode;          /* 0: the score rows go out as they are and the step arg-maxes them (the reference);
                                 1: categorical policy -- the action is DRAWN from softmax(scores / temperature) by the Gumbel-max rule:
                                    the row written to `scores` is scores / temperature + g, g_i = -log(-log u_i), u_i uniform in (0, 1)
                                    from Philox keyed like the noise draws (own stream), so the step's arg-max of that row IS the draw */
    float temperature;        /* > 0 when sample_mode = 1 */
    float* logp;              /* nullable float32 [E*A] (bsx_rollout_*: [T][E*A]): log softmax(scores / temperature)[drawn action] */
    const float* u_inject;    /* nullable float32 [E*A*4], 16-byte aligned: the uniforms to use instead of the Philox draws (bsx_actor_forward only: tests) */
    const float* value_weights; /* nullable: a second MLP per agent of the actor's shape (obs -> 64 -> LayerNorm -> ReLU -> 64 -> LayerNorm -> ReLU -> 1,
                                 the widths of maddpg/networks.py:14-52's critic on the agent's own observation), packed like `weights`
                                 (head column 0 = the value, columns 1-3 zero); evaluated in the call's `precision` mode on the rows the actor reads */
    float* value;             /* float32 [E*A] (bsx_rollout_*: [T][E*A]), required with value_weights: V(obs) = head + bias, no tanh.
                                 bsx_rollout_* take a value head for n = 1 only (the per-tick form for any n). */
} BsxActorNoise;
/* precision of the 64 x 64 layer: exact float32 (an fmaf chain, bit for bit), or both operands split in two bf16 terms and
 * three bf16 matrix products accumulated in float32 (about 1e-5 on a score; 16x the matrix rate).  All else is float32. */
#define BSX_ACTOR_F32 0
#define BSX_ACTOR_BF16X3 1
#define BSX_ACTOR_BF16X6 2   /* three bf16 terms per operand, six products: float32-class accuracy (~1e-7), 48 matrix instructions of 32 cycles
                               instead of 64 of 64; every actor entry point, every team size */
int bsx_actor_blob_floats(int obs_len, int* floats_per_agent);
int bsx_actor_forward(const float* weights, const float* obs, float* scores, int64_t E, int n, int precision,
                      const BsxActorNoise* noise, uint64_t seed, uint64_t seq, const uint64_t* seq_base, int64_t env_offset,
                      void* stream);

/* The caller's rollout loop -- `for t in range(T): actions = actor(obs) (+ noise, clamp); obs, rew, done = step(actions)`
 * (main.py:177-181 with maddpg/agent.py:25-33) -- in ONE launch: T x (bsx_actor_forward -> bsx_step_discrete with
 * BSX_ACT_LOGITS_F32), same results bit for bit.  A wavefront keeps its games' observation rows in LDS (the step writes
 * them, the actor's MFMA reads them), their state in registers / L2, and walks them through all T ticks.
 *   obs     float32 [T+1][E*A*D]: obs[0] = the observations to start from (in), obs[t+1] = after tick t (out)
 *   scores  float32 [T][E*A*4] (out, 16-byte aligned): what the actors produced = the actions taken
 *   rew     float32 [T][E*A], done uint8 [T][E*A] (out); env_done / winner (nullable, [E]): state after the last tick;
 *           env_done_t (nullable, uint8 [T][E]): env_done after every tick (which rows are transitions: see bsx_step_many_*)
 *   scripted_team  -1: both teams act by their actors; 0 / 1: the red / blue planes are played by the scripted opponent
 *            (bsx_instinct_discrete, instinct/agent.py:10-62) -- the reference's training setup, main.py:119-122 -- their actor
 *            is not evaluated, their score rows are the one-hot rows bsx_instinct_discrete writes, no noise
 *   weights, precision, noise, actor_seed, seq, seq_base: as bsx_actor_forward; tick t uses sequence number seq + *seq_base + t
 *   cfg, flags, seed, env_offset: as bsx_step_discrete (BSX_F_EMPTY_CALL is refused)
 * n = 1 .. 4.  An MFMA tile is 32 rows of one actor, so a workgroup is 32 games = G/2 wavefronts (1v1: one wave whose rows are
 * exactly two tiles; 2v2: two waves; 3v3 / 4v4: four) that exchange observation rows and arg-maxes through LDS.  Larger teams
 * return BSX_E_ARG -- use the per-tick form. */
int bsx_rollout_discrete(void* state, int64_t E, int n, int T, const float* weights, int precision, int scripted_team, float* obs, float* scores, float* rew,
                         uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg, uint32_t flags,
                         const BsxActorNoise* noise, uint64_t actor_seed, uint64_t seq, const uint64_t* seq_base,
                         uint64_t seed, int64_t env_offset, void* stream);
/* The same loop for a continuous-action env (battle_env.py:295-297,418-424; the reference's own driver test_env.py:22-43 is
 * continuous): T x (bsx_actor_forward -> bsx_step_continuous with BSX_ACT_F32X4), bit for bit.  The actors have three outputs
 * [speed, turn, shoot] padded to the 4-wide rows (`scores` [T][E*A*4]: what was fed to the step; the fourth value is ignored).
 *   scripted_team  -1 / 0 / 1 as above: that team's planes are played by the continuous scripted opponent (instinct/agent.py:41-54),
 *            exactly as bsx_instinct_continuous computes them with seed = scripted_seed and sequence number seq + *seq_base + t for
 *            tick t (row = e*A + a of this launch): binary64 actions that go to the step unrounded, as the reference's float64 arrays
 *            do; their `scores` rows record them rounded to float32.  The per-tick equivalent: the actors' rows widened to binary64
 *            next to bsx_instinct_continuous' rows in one BSX_ACT_F64 array. */
int bsx_rollout_continuous(void* state, int64_t E, int n, int T, const float* weights, int precision, int scripted_team, uint64_t scripted_seed,
                           float* obs, float* scores, float* rew,
                           uint8_t* done, uint8_t* env_done, uint8_t* winner, uint8_t* env_done_t, const BsxRewards* cfg, uint32_t flags,
                           const BsxActorNoise* noise, uint64_t actor_seed, uint64_t seq, const uint64_t* seq_base,
                           uint64_t seed, int64_t env_offset, void* stream);

/* ---- the reference's scripted opponent on device (instinct/agent.py:10-62, instinct/team.py:3-15): a pure function of
 * the observation rows.  Writes ONLY the rows of `team` (0 red, 1 blue, 2 both) so a learned policy can fill the others
 * in the same tensor.  binary64 arithmetic on the float32 observation values, as the reference computes.
 *   discrete:   actions int32[E*A] (BSX_ACT_I32) or one-hot +-1 float32[E*A*4] (BSX_ACT_LOGITS_F32, 16-byte aligned)
 *   continuous: actions float64[E*A*3] (speed, turn, shoot; noise added and clipped as agent.py:51-52);
 *               rnd nullable float64[E*A*4] = the np.random.rand() value and the three uniform(-0.15, 0.15) values to
 *               use (parity runs); null = Philox keyed by (seed, seq + *seq_base, row). */
int bsx_instinct_discrete(const float* obs, void* actions, int out_kind, int64_t E, int n, int team, void* stream);
int bsx_instinct_continuous(const float* obs, double* actions, const double* rnd, int64_t E, int n, int team,
                            uint64_t seed, uint64_t seq, const uint64_t* seq_base, void* stream);

/* Host helper: the call number on which the time-limit tie fires for n-per-team -- the reference accumulates
 * total_time += 0.1 in binary64 and compares >= 10+2n (battle_env.py:168,316-319): 121, 141, 161, 181, 200 ... */
int bsx_tie_tick(int n);

/* Self-test: the step path computes math.atan2 of pixel differences (battle_env.py:39) with its own instruction sequence --
 * the device library's algorithm with its constants held in scalar registers.  Counts the argument pairs (iy, ix) of
 * [-R, R]^2, R <= 4096 (the field is 1200 x 800), whose binary64 result differs from the library's: out[0] = differing,
 * out[1] = tested (device uint64[2]). */
int bsx_selftest_atan2(int R, uint64_t* out, void* stream);

/* hipHostGetDevicePointer for a binding that links no HIP runtime of its own: the device address of PINNED host memory
 * (hipHostMalloc / hipHostRegister) -- what the entry points above must be given for host-resident buffers.  With hipHostMalloc the
 * two addresses are equal; with registered memory they need not be.  Returns the hipError_t (memory that is not pinned: an error). */
int bsx_host_device_pointer(void* host, void** device);

/* hipStreamSynchronize(stream) for a binding that links no HIP runtime of its own: the single-game drop-in surface
 * returns host values from step() (battle_env.py:374-381), so it has to wait for the launch it just enqueued. */
int bsx_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BATTLESPACE_HIP_H */
