// bsx_actor_core.h -- device pieces of the per-agent actor MLP shared by the stand-alone actor kernel (bsx_actor.hip) and
// the fused rollout kernel (actor -> step in one launch, bsx_kernels.hip).  Reference: maddpg/networks.py:54-85,
// maddpg/agent.py:25-33, utils/noise.py:4-21.
//
// Both translation units must produce the SAME bits (the fused rollout is tested bit-for-bit against the two-kernel
// one), and bsx_kernels.hip is built with -ffp-contract=off: every function here pins its own contraction mode.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "battlespace_hip.h"

namespace bsx_actor {

constexpr int H = 64;            // fc1_dims = fc2_dims = 64 (main.py:15-16)
constexpr int NA = 4;            // discrete action scores
constexpr int SMALL = 6 * H + H * NA + NA;   // per-agent LDS block: b1 g1 be1 b2 g2 be2 | W3P | b3

typedef float f32x16 __attribute__((ext_vector_type(16)));

// packed blob per agent (floats); Dp = obs_len rounded up to even
//   W1A[mo 2][s Dp/2][lane 64]                 W1[k = 2s + (lane>>5)][32*mo + (lane&31)], 0 for k >= D
//   W2A[mo 2][mt 2][vq 4][lane 64][t 4]        W2[k = nid(mt, 4*vq+t, lane>>5)][32*mo + (lane&31)]
//   small: b1p g1p be1p b2p g2p be2p, each [hh 2][mo 2][v 16] = value[nid(mo, v, hh)]
//   W3P[hh 2][mt 2][v 16][4]                   W3[k = nid(mt, v, hh)][0..3]
//   b3[4]
//   W2B[mo 2][s 4][term 3][lane 64][i 8] bf16  the 64 x 64 layer once more, split in bf16 terms: term 0 = bf16(W2), term 1 =
//                                              bf16(W2 - term 0), term 2 = bf16(W2 - term 0 - term 1), of
//                                              W2[k = nid(s>>1, 8*(s&1) + i, lane>>5)][32*mo + (lane&31)]
//                                              (BSX_ACTOR_BF16X3 reads terms 0-1, BSX_ACTOR_BF16X6 all three)
// nid(m, v, hh) = 32*m + (v&3) + 8*(v>>2) + 4*hh  -- the neuron held by accumulator register v of tile m in lane half hh
__host__ __device__ constexpr int dpad(int D) { return (D + 1) & ~1; }
__host__ __device__ constexpr int off_w2(int D) { return H * dpad(D); }
__host__ __device__ constexpr int off_small(int D) { return off_w2(D) + H * H; }
__host__ __device__ constexpr int off_w3(int D) { return off_small(D) + 6 * H; }
__host__ __device__ constexpr int off_b3(int D) { return off_w3(D) + H * NA; }
__host__ __device__ constexpr int off_w2b(int D) { return off_b3(D) + NA; }            // 16-byte aligned: every section is a multiple of 4 floats
__host__ __device__ constexpr int blob_floats(int D) { return off_w2b(D) + H * H * 3 / 2; }     // 3 bf16 terms = 6 bytes per weight

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: both are quarter rate
        const uint64_t p0 = uint64_t(0xD2511F53u) * ctr.x, p1 = uint64_t(0xCD9E8D57u) * ctr.z;
        const uint32_t hi0 = uint32_t(p0 >> 32), lo0 = uint32_t(p0), hi1 = uint32_t(p1 >> 32), lo1 = uint32_t(p1);
        // (the three-input XORs as ONE v_bitop3_b32 each: left to itself the compiler issues two v_xor_b32 -- 20 vector instructions
        //  per block where the instruction count is what the step is bound by)
        ctr = make_uint4(__builtin_amdgcn_bitop3_b32(hi1, ctr.y, key.x, 0x96), lo1, __builtin_amdgcn_bitop3_b32(hi0, ctr.w, key.y, 0x96), lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}

// LayerNorm over the 64 neurons of each row + ReLU, in place on ONE 32-row tile held as acc[mo] (torch semantics: biased
// variance, eps 1e-5).  A row's 64 neurons = 32 registers of the lane + the partner lane l^32.  gp / bp: this lane half's
// gain / bias vectors, [mo 2][v 16] floats in LDS (8-byte aligned).  Written on register PAIRS so that the adds, multiplies
// and FMAs issue as packed-f32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: two values per issue slot).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline void ln_relu_tile(f32x16& a0, f32x16& a1, const float* __restrict__ gp, const float* __restrict__ bp) {
#pragma clang fp contract(fast)
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 16; v += 2) s2 += f32x2{a0[v], a0[v + 1]};
#pragma unroll
    for (int v = 0; v < 16; v += 2) s2 += f32x2{a1[v], a1[v + 1]};
    float s = s2.x + s2.y;
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / H);
    const f32x2 m2 = {mean, mean};
    f32x2 d0[8], d1[8], q2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { d0[i] = f32x2{a0[2 * i], a0[2 * i + 1]} - m2; q2 += d0[i] * d0[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { d1[i] = f32x2{a1[2 * i], a1[2 * i + 1]} - m2; q2 += d1[i] * d1[i]; }
    float q = q2.x + q2.y;
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q * (1.0f / H) + 1e-5f);
    const f32x2 r2 = {rstd, rstd};
    const f32x2* g2 = reinterpret_cast<const f32x2*>(gp);
    const f32x2* b2 = reinterpret_cast<const f32x2*>(bp);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2 y = d0[i] * (r2 * g2[i]) + b2[i];
        a0[2 * i] = fmaxf(y.x, 0.f); a0[2 * i + 1] = fmaxf(y.y, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2 y = d1[i] * (r2 * g2[8 + i]) + b2[8 + i];
        a1[2 * i] = fmaxf(y.x, 0.f); a1[2 * i + 1] = fmaxf(y.y, 0.f);
    }
}

// Four standard normals from one Philox4x32-10 block (Box-Muller), keyed by (seed, seq ^ salt, global row).
__device__ inline float4 normals4(uint64_t seed, uint64_t seq, uint64_t grow, uint32_t salt) {
#pragma clang fp contract(fast)
    const uint4 r = philox4x32_10(make_uint4(uint32_t(grow), uint32_t(grow >> 32), uint32_t(seq), uint32_t(seq >> 32) ^ salt),
                                  make_uint2(uint32_t(seed), uint32_t(seed >> 32) ^ 0xA5A5A5A5u));
    const float u0 = (float(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = float(r.y >> 8) * (1.0f / 16777216.0f);
    const float u2 = (float(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = float(r.w >> 8) * (1.0f / 16777216.0f);
    const float m0 = sqrtf(-2.0f * __logf(u0)), m1 = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.2831853071795864f * u1, &s0, &c0);
    __sincosf(6.2831853071795864f * u3, &s1, &c1);
    return make_float4(m0 * c0, m0 * s0, m1 * c1, m1 * s1);
}

// Four uniforms in (0, 1) from one Philox4x32-10 block, keyed like normals4 (own salt): the categorical policy's Gumbel draws.
__device__ inline float4 uniforms4(uint64_t seed, uint64_t seq, uint64_t grow, uint32_t salt) {
#pragma clang fp contract(fast)
    const uint4 r = philox4x32_10(make_uint4(uint32_t(grow), uint32_t(grow >> 32), uint32_t(seq), uint32_t(seq >> 32) ^ salt),
                                  make_uint2(uint32_t(seed), uint32_t(seed >> 32) ^ 0xA5A5A5A5u));
    const float k = 1.0f / 16777216.0f;
    return make_float4((float(r.x >> 8) + 0.5f) * k, (float(r.y >> 8) + 0.5f) * k, (float(r.z >> 8) + 0.5f) * k, (float(r.w >> 8) + 0.5f) * k);
}

// tanh of the head sums + bias, exploration noise, clamp (maddpg/networks.py:85, maddpg/agent.py:30-31) for ONE row.
// `row` indexes this launch's arrays (OU state, injected normals); `grow` = the job-wide row (env_offset + e)*A + a keys the
// draws, so the noise a game sees does not depend on the sharding.  Optional Ornstein-Uhlenbeck state at ou_state[row]
// (utils/noise.py:17-21), restarted from mu when `game_over` (main.py:155); with both processes on, the Gaussian term takes its
// own draw (salt) -- the OU increment and the white term are independent.  `store`: this lane owns a real row.
// sample_mode 1 (categorical policy): the row becomes scores / temperature + Gumbel noise -- its arg-max is a draw from
// softmax(scores / temperature) -- and the drawn action's log-probability goes to nz.logp[orow] (orow: the row in a [T][E*A] record).
template <class NZ>   // BsxActorNoise, in whatever address space the caller holds it (the fused kernels read it from the kernarg segment)
__device__ inline float4 finish_row(float4 r4, const float4 b3, const NZ& nz, uint64_t seed, uint64_t seq,
                                    size_t row, uint64_t grow, bool game_over, bool store, size_t orow) {
    // No implicit contraction in here: the stand-alone actor kernel and the fused rollout must produce the same bits, and which
    // multiply the back end would fuse into which add depends on the code around the call.  Every fused operation is spelled out.
#pragma clang fp contract(off)
    r4.x = tanhf(r4.x + b3.x); r4.y = tanhf(r4.y + b3.y); r4.z = tanhf(r4.z + b3.z); r4.w = tanhf(r4.w + b3.w);
    if (nz.gaussian_std > 0.f || nz.ou_scale > 0.f) {
        const float4 z = nz.z_inject ? reinterpret_cast<const float4*>(nz.z_inject)[row] : normals4(seed, seq, grow, 0u);
        if (nz.ou_scale > 0.f) {
            float4* xs = reinterpret_cast<float4*>(nz.ou_state) + row;
            float4 x = *xs;
            if (game_over) x = make_float4(nz.ou_mu, nz.ou_mu, nz.ou_mu, nz.ou_mu);
            x.x += fmaf(nz.ou_theta, nz.ou_mu - x.x, nz.ou_sigma * z.x);
            x.y += fmaf(nz.ou_theta, nz.ou_mu - x.y, nz.ou_sigma * z.y);
            x.z += fmaf(nz.ou_theta, nz.ou_mu - x.z, nz.ou_sigma * z.z);
            x.w += fmaf(nz.ou_theta, nz.ou_mu - x.w, nz.ou_sigma * z.w);
            if (store) *xs = x;
            r4.x = fmaf(nz.ou_scale, x.x, r4.x); r4.y = fmaf(nz.ou_scale, x.y, r4.y);
            r4.z = fmaf(nz.ou_scale, x.z, r4.z); r4.w = fmaf(nz.ou_scale, x.w, r4.w);
        }
        if (nz.gaussian_std > 0.f) {
            const float4 zg = (nz.ou_scale > 0.f && !nz.z_inject) ? normals4(seed, seq, grow, 0x80000000u) : z;
            r4.x = fmaf(nz.gaussian_std, zg.x, r4.x); r4.y = fmaf(nz.gaussian_std, zg.y, r4.y);
            r4.z = fmaf(nz.gaussian_std, zg.z, r4.z); r4.w = fmaf(nz.gaussian_std, zg.w, r4.w);
        }
        r4.x = fminf(fmaxf(r4.x, -1.f), 1.f); r4.y = fminf(fmaxf(r4.y, -1.f), 1.f);
        r4.z = fminf(fmaxf(r4.z, -1.f), 1.f); r4.w = fminf(fmaxf(r4.w, -1.f), 1.f);
    }
    if (nz.sample_mode == 1) {
        const float it = 1.0f / nz.temperature;
        const float4 z = make_float4(r4.x * it, r4.y * it, r4.z * it, r4.w * it);
        const float4 u = nz.u_inject ? reinterpret_cast<const float4*>(nz.u_inject)[row] : uniforms4(seed, seq, grow, 0x40000000u);
        const float4 pr = make_float4(z.x - __logf(-__logf(u.x)), z.y - __logf(-__logf(u.y)), z.z - __logf(-__logf(u.z)), z.w - __logf(-__logf(u.w)));
        // the draw = the first maximum of the perturbed row, as the step's arg-max will find it; its log-probability under softmax(z)
        int am = 0; float best = pr.x, za = z.x;
        if (pr.y > best) { am = 1; best = pr.y; za = z.y; }
        if (pr.z > best) { am = 2; best = pr.z; za = z.z; }
        if (pr.w > best) { am = 3; best = pr.w; za = z.w; }
        const float m = fmaxf(fmaxf(z.x, z.y), fmaxf(z.z, z.w));
        const float lse = m + __logf(__expf(z.x - m) + __expf(z.y - m) + __expf(z.z - m) + __expf(z.w - m));
        if (store && nz.logp) nz.logp[orow] = za - lse;
        (void)am;
        r4 = pr;
    }
    return r4;
}

// ONE 32-row tile of ONE agent through the three layers, everything transposed (see bsx_actor.hip): returns the four head
// sums of row (lane & 31), already added across the two lane halves (both halves hold them).
//   W   the agent's packed blob (global; layer-2 operands are fetched here, 16 x 16-byte loads per lane)
//   sm  the agent's SMALL block in LDS
//   xb  xb(k) = this lane's layer-1 B operand: observation value k of row (lane & 31), 0 for k >= D
//
// PREC = BSX_ACTOR_BF16X3 (1): the 64 x 64 layer on the bf16 matrix path with both operands split in two bf16 terms, x = xh + xl,
// w = wh + wl, and three products wh*xh + wh*xl + wl*xh accumulated in f32 (the dropped wl*xl and the split residue are ~2^-16
// relative: about 1e-5 on a score, against 4e-3 for plain bf16).  v_mfma_f32_32x32x16_bf16 retires 16x the multiply-adds per cycle of
// the f32 MFMA -- whose rate equals packed-f32 VALU and which does not overlap with vector instructions -- so the layer costs 24
// matrix instructions of 32 cycles instead of 64 of 64.
// PREC = BSX_ACTOR_BF16X6 (2): THREE terms per operand (8 + 8 + 8 significant bits = float32's 24) and the six products of order up to
// 2^-16 -- wh*xh, wh*xm, wm*xh, wh*xl, wm*xm, wl*xh, added smallest first; what is dropped is below 2^-24 relative, the size of a
// float32 rounding -- 48 matrix instructions of 32 cycles: float32-class accuracy (not the fmaf chain's bits) at 2.7x the f32 MFMA rate.
// Layer 1 (K = 6) and everything else stay f32 in every mode.
// ROLL: the 64 x 64 layer's operands as a ROLLING window instead of all at once -- the first groups are requested behind layer 1's
// loads, group g + AHEAD right before group g is multiplied (f32: a group = the eight registers of four K steps, 8 MFMAs = 512
// cycles, AHEAD = 3; bf16: a group = one K step's terms for both output tiles, AHEAD = 2), so that at most AHEAD + 1 groups are
// live: 32 / 48 weight registers instead of 64.  Same loads in the same (K) order, same arithmetic; it is what lets the fused
// 2v2 ... 4v4 rollout kernels (bsx_kernels.hip), which carry a game's state across the actor, fit 256 registers without scratch.
template <int PREC, bool ROLL = false, class XB>
__device__ inline float4 tile_forward(const float* __restrict__ W, const float* __restrict__ sm_agent, int D, int lane, XB xb) {
#pragma clang fp contract(fast)
    constexpr bool BF = PREC != BSX_ACTOR_F32;
    constexpr int NT = PREC == BSX_ACTOR_BF16X6 ? 3 : 2;                        // bf16 terms read per weight
    const int Dp = dpad(D), hh = lane >> 5;
    const float* sm = sm_agent + hh * 32;      // this lane half's [mo][v] slice of each 64-float vector ([hh][mo][v])
    f32x16 acc1[2];
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1[mo][v] = sm[0 * H + mo * 16 + v];
    for (int s = 0; s < Dp / 2; ++s) {
        const int k = 2 * s + hh;
        const float a0 = W[(0 * (Dp / 2) + s) * 64 + lane], a1 = W[(1 * (Dp / 2) + s) * 64 + lane];
        const float b = xb(k);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc1[0], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1[1], 0, 0, 0);
    }
    // The 64 x 64 layer's operands are fetched HERE, behind layer 1's own loads: vmcnt counts in order, so any load issued
    // after these would have to wait for all 16 KB of them; the LayerNorm below (~1 k cycles of VALU) covers their latency.
    constexpr int W2AHEAD = ROLL ? 3 : 8, WBAHEAD = ROLL ? 2 : 4;
    float4 w2[8][2];                           // f32: [group g = mt * 4 + vq][mo] x 4 k-steps
    float4 wb[4][2][NT];                       // bf16 modes: [s][mo][term] x 8 bf16
    auto load_w2 = [&](int g) {
        w2[g][0] = reinterpret_cast<const float4*>(W + off_w2(D))[((0 * 2 + (g >> 2)) * 4 + (g & 3)) * 64 + lane];
        w2[g][1] = reinterpret_cast<const float4*>(W + off_w2(D))[((1 * 2 + (g >> 2)) * 4 + (g & 3)) * 64 + lane];
    };
    auto load_wb = [&](int s) {
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wb[s][mo][t] = reinterpret_cast<const float4*>(W + off_w2b(D))[((mo * 4 + s) * 3 + t) * 64 + lane];
    };
    if constexpr (!BF) {
#pragma unroll
        for (int g = 0; g < (W2AHEAD < 8 ? W2AHEAD : 8); ++g) load_w2(g);
    } else {
#pragma unroll
        for (int s = 0; s < (WBAHEAD < 4 ? WBAHEAD : 4); ++s) load_wb(s);
    }
    ln_relu_tile(acc1[0], acc1[1], sm + 1 * H, sm + 2 * H);
    f32x16 acc2[2];
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2[mo][v] = sm[3 * H + mo * 16 + v];
    if constexpr (BF) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {          // K step s = accumulator registers 8*(s&1) .. +7 of layer-1 tile s>>1
            if constexpr (ROLL) {
                if (s + WBAHEAD < 4) load_wb(s + WBAHEAD);
            }
            bf16x8 xh, xm, xl;                 // two-term mode: xm is the low term, xl unused
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x = acc1[s >> 1][8 * (s & 1) + i];
                const __bf16 h = static_cast<__bf16>(x);                       // round to nearest even
                const float r1 = x - static_cast<float>(h);                    // exact
                const __bf16 m = static_cast<__bf16>(r1);
                xh[i] = h; xm[i] = m;
                xl[i] = static_cast<__bf16>(r1 - static_cast<float>(m));
            }
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
                const bf16x8 wh = __builtin_bit_cast(bf16x8, wb[s][mo][0]);
                const bf16x8 wm = __builtin_bit_cast(bf16x8, wb[s][mo][1]);
                if constexpr (PREC == BSX_ACTOR_BF16X6) {
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, wb[s][mo][NT - 1]);
                    acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc2[mo], 0, 0, 0);
                    acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, acc2[mo], 0, 0, 0);
                    acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc2[mo], 0, 0, 0);
                }
                acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc2[mo], 0, 0, 0);
                acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, acc2[mo], 0, 0, 0);
                acc2[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc2[mo], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) {          // K order: mt = g >> 2, accumulator registers 4 (g & 3) .. + 3 of layer 1's tile mt
            if constexpr (ROLL) {
                if (g + W2AHEAD < 8) load_w2(g + W2AHEAD);
                __builtin_amdgcn_sched_barrier(0);   // (the window is the point: the scheduler must not pull the later loads up to the first)
            }
            const float4 q0 = w2[g][0], q1 = w2[g][1];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mt = g >> 2, v = 4 * (g & 3) + t;
                const float wa0 = t == 0 ? q0.x : (t == 1 ? q0.y : (t == 2 ? q0.z : q0.w));
                const float wa1 = t == 0 ? q1.x : (t == 1 ? q1.y : (t == 2 ? q1.z : q1.w));
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, acc1[mt][v], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, acc1[mt][v], acc2[1], 0, 0, 0);
            }
        }
    }
    ln_relu_tile(acc2[0], acc2[1], sm + 4 * H, sm + 5 * H);
    const float4* w3 = reinterpret_cast<const float4*>(sm_agent + 6 * H) + hh * 32;   // [hh][mt][v] float4
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float4 ww = w3[mt * 16 + v];
            const float hv = acc2[mt][v];
            o.x = fmaf(hv, ww.x, o.x); o.y = fmaf(hv, ww.y, o.y); o.z = fmaf(hv, ww.z, o.z); o.w = fmaf(hv, ww.w, o.w);
        }
    o.x += __shfl_xor(o.x, 32); o.y += __shfl_xor(o.y, 32); o.z += __shfl_xor(o.z, 32); o.w += __shfl_xor(o.w, 32);
    return o;
}

}  // namespace bsx_actor
