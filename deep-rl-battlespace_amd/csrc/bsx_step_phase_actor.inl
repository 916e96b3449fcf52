// bsx_step_phase_actor.inl -- a PHASE of bsx_step_kernel's tick (bsx_step_kernel.h includes it inside the kernel body, in tick order; it shares the
// kernel's locals, so this is a textual unit for reading and review, not a function): fused rollout only (ACTOR): this tick's actions = arg-max / squash of actor(obs) on the wave's LDS observation rows (MFMA, bsx_actor_core.h), or the
// scripted opponent's.  Reads: s_obs_all, s_small, er.done, p (weights, noise, seeds).  Writes: act (discrete) or a0, a1, a2 (continuous), p.scores,
// p.nz.logp / value.  Compiles to nothing in the per-step and multi-tick kernels.
// The contract (tools/check_phase_contract.py checks it against this file's text in the CPU suite; names are the kernel's locals):
// @reads -
// @writes  a0 a1 a2 act
// @exports -
// @lds     s_act_all s_gdone_all
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

