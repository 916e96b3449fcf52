// bsx_actor.hip -- fused per-agent actor MLP for the on-device policy rollout (BASELINE.json configs[4]), gfx950.
//
// Replaces, for the rollout path, the reference's per-agent `ActorNetwork.forward` + noise + clamp
// (maddpg/networks.py:81-85, maddpg/agent.py:25-33): obs[D] -> Linear 64 -> LayerNorm -> ReLU -> Linear 64 ->
// LayerNorm -> ReLU -> Linear 4 -> tanh (-> + N(0, std) -> clamp(-1, 1)), one independent weight set per agent.
// Composed from torch ops this is ~20 memory-bound passes over [A, E, 64] activations (420 us per tick at
// 65 536 x 1v1); fused, a row never leaves the register file: 4*D bytes in, 16 bytes out.
//
// This is the one GEMM-shaped op of the project, so it runs on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate -- bit-for-bit an fmaf chain, no reduced precision).  Everything is computed TRANSPOSED,
//     H^T [64 neurons x 64 rows] = W^T [64 x K] * X^T [K x 64 rows],
// so that a wave's 64 observation rows sit on the N (lane) axis and neurons on the M (register) axis:
//   * A operand = weights, one VGPR per k-step and 32-neuron tile, loaded once per wave (the 64x64 layer is 64 VGPRs);
//   * B operand of layer 1 = observation values straight from memory;
//   * B operand of layer 2 = the ACCUMULATOR REGISTERS of layer 1 as they stand: register v of tile mt in lane l holds
//     neuron 32*mt + (v&3) + 8*(v>>2) + 4*(l>>5) of row (l&31) -- exactly a (k_lo, k_hi) pair of a K=2 step.  The K order
//     is therefore permuted; the host packs W2 in the same order, and a sum does not care.  No LDS, no shuffles.
//   * LayerNorm needs a row's 64 neurons: 32 registers of the lane plus the partner lane l^32 -> one xor-32 shuffle.
//   * the 64 -> 4 head is 128 VALU FMAs per lane on the accumulator registers plus the same xor-32 add.
// A workgroup = 4 waves x 64 rows of ONE agent (blockIdx.y); the small per-neuron vectors and the head weights are
// staged once in LDS.  Checked against a torch fp32 reference (2e-5), not bit-exact with it; bit-exact with the fused
// rollout kernel, which runs the same per-tile arithmetic (bsx_actor_core.h): both files are built with -ffp-contract=off
// (so that library code such as tanhf compiles the same in both) and the core functions pin contraction themselves.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "battlespace_hip.h"
#include "bsx_actor_core.h"

namespace {

using namespace bsx_actor;       // H, NA, SMALL, blob layout, ln_relu_tile, finish_row (shared with the fused rollout kernel)

constexpr int TPB = 256;
constexpr int ROWS_PER_WAVE = 64;

struct ActorArgs {
    const float* weights; const float* obs; float* scores;
    int64_t E; int A; int D; BsxActorNoise nz; uint64_t seed; uint64_t seq; const uint64_t* seq_base; int64_t env_offset;
};

// PREC = BSX_ACTOR_F32: exact f32, both 32-row tiles of the wave interleaved (four independent accumulators).
// PREC = BSX_ACTOR_BF16X3 / _BF16X6: the 64 x 64 layer as split-bf16 MFMAs (bsx_actor_core.h), one tile after the other.
template <int PREC>
__global__ __launch_bounds__(TPB) void bsx_actor_kernel(const ActorArgs p) {
#pragma clang fp contract(fast)
    const int D = p.D, Dp = dpad(D);
    const int a = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, c = lane & 31;
    const float* __restrict__ W = p.weights + size_t(a) * blob_floats(D);

    // per-neuron vectors (6 x 64), head weights (256) and head bias (4): once per workgroup into LDS
    __shared__ __attribute__((aligned(16))) float s_small[SMALL];
    for (int i = tid; i < SMALL / 4; i += TPB)
        reinterpret_cast<float4*>(s_small)[i] = reinterpret_cast<const float4*>(W + off_small(D))[i];

    const int64_t row0 = (int64_t(blockIdx.x) * (TPB / 64) + wave) * ROWS_PER_WAVE;   // first env of this wave
    // the replay counter is a device word behind a kernel argument: requested here, with the first weight loads, not where the noise
    // needs it (the compiler's choice -- a scalar miss with nothing left to hide it at the end of the kernel)
    uint64_t seq_add = p.seq_base ? *p.seq_base : 0ull;
    float4 o[2];
    if constexpr (PREC != BSX_ACTOR_F32) {
        __syncthreads();
        o[0] = make_float4(0.f, 0.f, 0.f, 0.f); o[1] = o[0];
#pragma nounroll
        for (int nt = 0; nt < 2; ++nt) {                 // rolled: one tile's registers at a time
            const int64_t en = row0 + 32 * nt + c;
            const float* xr = p.obs + (size_t(en < p.E ? en : p.E - 1) * p.A + a) * D;
            const float* Wn = W;
            asm volatile("" : "+s"(Wn));                 // keeps the weight loads inside the loop (hoisted, they double the registers)
            const float4 t = tile_forward<PREC>(Wn, s_small, D, lane, [&](int k) { return k < D ? xr[k] : 0.f; });
            if (nt == 0) o[0] = t; else o[1] = t;
        }
    } else {
        __syncthreads();
        const float* sm = s_small + hh * 32;      // this lane half's [mo][v] slice of each 64-float vector ([hh][mo][v])

        // ---- layer 1: acc1[mo][nt] = b1 + W1^T * X^T
        f32x16 acc1[2][2];
    #pragma unroll
        for (int mo = 0; mo < 2; ++mo)
    #pragma unroll
            for (int v = 0; v < 16; ++v) { acc1[mo][0][v] = sm[0 * H + mo * 16 + v]; acc1[mo][1][v] = acc1[mo][0][v]; }
        const float* xrow[2];
    #pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t en = row0 + 32 * nt + c;
            xrow[nt] = p.obs + (size_t(en < p.E ? en : p.E - 1) * p.A + a) * D;
        }
        for (int s = 0; s < Dp / 2; ++s) {
            const int k = 2 * s + hh;
            const float a0 = W[(0 * (Dp / 2) + s) * 64 + lane], a1 = W[(1 * (Dp / 2) + s) * 64 + lane];
            const float b0 = k < D ? xrow[0][k] : 0.f, b1 = k < D ? xrow[1][k] : 0.f;
            acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc1[0][0], 0, 0, 0);
            acc1[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc1[0][1], 0, 0, 0);
            acc1[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc1[1][0], 0, 0, 0);
            acc1[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1[1][1], 0, 0, 0);
        }
        // the 64 x 64 layer's A operands: 64 VGPRs, coalesced 16-byte loads, issued behind layer 1's own loads (vmcnt is in-order), in flight under the LayerNorm
        float4 w2[2][2][4];
    #pragma unroll
        for (int mo = 0; mo < 2; ++mo)
    #pragma unroll
            for (int mt = 0; mt < 2; ++mt)
    #pragma unroll
                for (int vq = 0; vq < 4; ++vq)
                    w2[mo][mt][vq] = reinterpret_cast<const float4*>(W + off_w2(D))[((mo * 2 + mt) * 4 + vq) * 64 + lane];

        asm volatile("" : "+s"(seq_add));                // (waited for here, under the 64 x 64 layer's weight loads)
        ln_relu_tile(acc1[0][0], acc1[1][0], sm + 1 * H, sm + 2 * H);
        ln_relu_tile(acc1[0][1], acc1[1][1], sm + 1 * H, sm + 2 * H);

        // ---- layer 2: acc2[mo][nt] = b2 + W2^T * H1^T, the K index running over layer 1's accumulator registers
        f32x16 acc2[2][2];
    #pragma unroll
        for (int mo = 0; mo < 2; ++mo)
    #pragma unroll
            for (int v = 0; v < 16; ++v) { acc2[mo][0][v] = sm[3 * H + mo * 16 + v]; acc2[mo][1][v] = acc2[mo][0][v]; }
    #pragma unroll
        for (int mt = 0; mt < 2; ++mt)
    #pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float4 q0 = w2[0][mt][v >> 2], q1 = w2[1][mt][v >> 2];
                const float wa0 = (v & 3) == 0 ? q0.x : ((v & 3) == 1 ? q0.y : ((v & 3) == 2 ? q0.z : q0.w));
                const float wa1 = (v & 3) == 0 ? q1.x : ((v & 3) == 1 ? q1.y : ((v & 3) == 2 ? q1.z : q1.w));
                acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, acc1[mt][0][v], acc2[0][0], 0, 0, 0);
                acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, acc1[mt][1][v], acc2[0][1], 0, 0, 0);
                acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, acc1[mt][0][v], acc2[1][0], 0, 0, 0);
                acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, acc1[mt][1][v], acc2[1][1], 0, 0, 0);
            }
        ln_relu_tile(acc2[0][0], acc2[1][0], sm + 4 * H, sm + 5 * H);
        ln_relu_tile(acc2[0][1], acc2[1][1], sm + 4 * H, sm + 5 * H);

        // ---- head: 64 -> 4 on the vector pipe; each lane sums its 32 neurons, the partner lane l^32 has the other 32
        const float4* w3 = reinterpret_cast<const float4*>(s_small + 6 * H) + hh * 32;   // [hh][mt][v] float4
        o[0] = make_float4(0.f, 0.f, 0.f, 0.f); o[1] = o[0];
    #pragma unroll
        for (int mt = 0; mt < 2; ++mt)
    #pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float4 ww = w3[mt * 16 + v];
    #pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float hv = acc2[mt][nt][v];
                    o[nt].x = fmaf(hv, ww.x, o[nt].x); o[nt].y = fmaf(hv, ww.y, o[nt].y);
                    o[nt].z = fmaf(hv, ww.z, o[nt].z); o[nt].w = fmaf(hv, ww.w, o[nt].w);
                }
            }
    #pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            o[nt].x += __shfl_xor(o[nt].x, 32); o[nt].y += __shfl_xor(o[nt].y, 32);
            o[nt].z += __shfl_xor(o[nt].z, 32); o[nt].w += __shfl_xor(o[nt].w, 32);
        }
}
    // lower half finishes the rows of tile 0, upper half those of tile 1: every lane writes one row
    float4 r4 = hh ? o[1] : o[0];
    const float4 b3 = *reinterpret_cast<const float4*>(s_small + 6 * H + H * NA);
    const int64_t e = row0 + 32 * hh + c;
    const size_t row = size_t(e < p.E ? e : p.E - 1) * p.A + a;
    const uint64_t seq = p.seq + seq_add;                // seq_base: device word, so graph replays re-key
    const bool game_over = p.nz.ou_scale > 0.f && p.nz.env_done && p.nz.env_done[e < p.E ? e : p.E - 1];
    r4 = finish_row(r4, b3, p.nz, p.seed, seq, row, uint64_t(p.env_offset) * uint64_t(p.A) + row, game_over, e < p.E, row);
    if (e < p.E) reinterpret_cast<float4*>(p.scores)[row] = r4;
    if (p.nz.value_weights) {
        // ---- the value head: a second MLP of the same shape on the same rows, in the same precision mode (per-tile code of
        //      bsx_actor_core.h, which the fused rollout runs as well: same bits), one 32-row tile at a time; its
        //      per-neuron vectors and head take the place of the actor's in LDS (every lane is done with them)
        const float* __restrict__ Wv = p.nz.value_weights + size_t(a) * blob_floats(D);
        __syncthreads();
        for (int i = tid; i < SMALL / 4; i += TPB)
            reinterpret_cast<float4*>(s_small)[i] = reinterpret_cast<const float4*>(Wv + off_small(D))[i];
        __syncthreads();
        float4 v[2];
#pragma nounroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t en = row0 + 32 * nt + c;
            const float* xr = p.obs + (size_t(en < p.E ? en : p.E - 1) * p.A + a) * D;
            const float* Wn = Wv;
            asm volatile("" : "+s"(Wn));
            const float4 t = tile_forward<PREC>(Wn, s_small, D, lane, [&](int k) { return k < D ? xr[k] : 0.f; });
            if (nt == 0) v[0] = t; else v[1] = t;
        }
        const float vh = hh ? v[1].x : v[0].x;
        if (e < p.E) p.nz.value[row] = vh + s_small[6 * H + H * NA];
    }
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

extern "C" {

int bsx_actor_blob_floats(int obs_len, int* floats_per_agent) {
    if (obs_len < 1 || obs_len > 3 * BSX_MAX_N + 2 || !floats_per_agent) return BSX_E_ARG;
    *floats_per_agent = blob_floats(obs_len);
    return 0;
}

int bsx_actor_forward(const float* weights, const float* obs, float* scores, int64_t E, int n, int precision,
                      const BsxActorNoise* noise, uint64_t seed, uint64_t seq, const uint64_t* seq_base, int64_t env_offset,
                      void* stream) {
    if (!weights || !obs || !scores || E <= 0 || E > BSX_MAX_E || n < 1 || n > BSX_MAX_N) return BSX_E_ARG;
    if (precision != BSX_ACTOR_F32 && precision != BSX_ACTOR_BF16X3 && precision != BSX_ACTOR_BF16X6) return BSX_E_ARG;
    if (!aligned(weights, 16) || !aligned(scores, 16) || !aligned(obs, 4)) return BSX_E_ALIGN;
    BsxActorNoise nz = {};
    if (noise) nz = *noise;
    if (nz.ou_scale > 0.f && (!nz.ou_state || !aligned(nz.ou_state, 16))) return nz.ou_state ? BSX_E_ALIGN : BSX_E_ARG;
    if (nz.z_inject && !aligned(nz.z_inject, 16)) return BSX_E_ALIGN;
    if (nz.sample_mode != 0 && (nz.sample_mode != 1 || !(nz.temperature > 0.f))) return BSX_E_ARG;
    if ((nz.u_inject && !aligned(nz.u_inject, 16)) || (nz.logp && !aligned(nz.logp, 4)) || (nz.value_weights && !aligned(nz.value_weights, 16))) return BSX_E_ALIGN;
    if (nz.value_weights && (!nz.value || !aligned(nz.value, 4))) return nz.value ? BSX_E_ALIGN : BSX_E_ARG;
    if (env_offset < 0) return BSX_E_ARG;
    const int A = 2 * n, D = 3 * n + 2;
    ActorArgs a{weights, obs, scores, E, A, D, nz, seed, seq, seq_base, env_offset};
    const int rows_per_block = (TPB / 64) * ROWS_PER_WAVE;
    const dim3 grid(unsigned((E + rows_per_block - 1) / rows_per_block), unsigned(A)), block(TPB);
    if (precision == BSX_ACTOR_BF16X6) hipLaunchKernelGGL(bsx_actor_kernel<BSX_ACTOR_BF16X6>, grid, block, 0, static_cast<hipStream_t>(stream), a);
    else if (precision == BSX_ACTOR_BF16X3) hipLaunchKernelGGL(bsx_actor_kernel<BSX_ACTOR_BF16X3>, grid, block, 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(bsx_actor_kernel<BSX_ACTOR_F32>, grid, block, 0, static_cast<hipStream_t>(stream), a);
    return int(hipGetLastError());
}

}  // extern "C"
