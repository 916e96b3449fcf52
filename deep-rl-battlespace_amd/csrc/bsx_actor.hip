// bsx_actor.hip -- fused per-agent actor MLP for the on-device policy rollout (BASELINE.json configs[4]), gfx950.
//
// Replaces, for the rollout path, the reference's per-agent `ActorNetwork.forward` + noise + clamp
// (maddpg/networks.py:81-85, maddpg/agent.py:25-33): obs[D] -> Linear 64 -> LayerNorm -> ReLU -> Linear 64 ->
// LayerNorm -> ReLU -> Linear n_actions(4) -> tanh (-> + N(0, std) -> clamp(-1, 1)), one independent weight set per
// agent.  Composed from torch ops this is ~20 memory-bound passes over [A, E, 64] activations (420 us per tick at
// 65 536 x 1v1); fused, a row never leaves registers: 20 B in, 16 B out.
//
// Mapping: one lane = one observation row; a workgroup handles 256 rows of ONE agent index, so its weights are
// workgroup-uniform: they arrive through the SCALAR path (s_load -> SGPR operands of the FMAs), not LDS or VGPRs.
// The 64x64 layer is 4096 FMAs per row issued as packed-f32 FMAs (v_pk_fma_f32) with both activation vectors in
// VGPRs.  f32 MFMA would run at the same rate as packed VALU on gfx950 (64 FLOP/clk/SIMD) and the per-lane row layout
// needs no fragment shuffles, so this stays on the vector pipe.
// Built WITHOUT -ffp-contract=off (no bit-exactness contract here; checked against a torch fp32 reference).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "battlespace_hip.h"

namespace {

constexpr int H = 64;            // fc1_dims = fc2_dims = 64 (main.py:15-16)
constexpr int NA = 4;            // discrete action scores
constexpr int TPB = 256;

typedef float float2v __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int blob_floats(int D) {
    // W1[D][H] b1 g1 be1 | W2[H][H] b2 g2 be2 | W3[H][NA] b3[NA]
    return D * H + 3 * H + H * H + 3 * H + H * NA + NA;
}

__device__ inline uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}

// y = relu(layernorm(h) * g + b), in place, h in registers (torch semantics: biased variance, eps 1e-5)
__device__ inline void ln_relu(float (&h)[H], const float* __restrict__ g, const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < H; ++j) s += h[j];
    const float mean = s * (1.0f / H);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < H; ++j) { const float d = h[j] - mean; v = fmaf(d, d, v); }
    const float rstd = rsqrtf(v * (1.0f / H) + 1e-5f);
#pragma unroll
    for (int j = 0; j < H; j += 4) {
        const float4 gg = *reinterpret_cast<const float4*>(g + j), bb = *reinterpret_cast<const float4*>(b + j);
        h[j + 0] = fmaxf(fmaf((h[j + 0] - mean) * rstd, gg.x, bb.x), 0.f);
        h[j + 1] = fmaxf(fmaf((h[j + 1] - mean) * rstd, gg.y, bb.y), 0.f);
        h[j + 2] = fmaxf(fmaf((h[j + 2] - mean) * rstd, gg.z, bb.z), 0.f);
        h[j + 3] = fmaxf(fmaf((h[j + 3] - mean) * rstd, gg.w, bb.w), 0.f);
    }
}

// out[j] += x * w[j], j = 0..H-1, as packed FMAs; w = 64 contiguous floats at a wave-uniform address (scalar loads)
__device__ inline void axpy64(float2v (&acc)[H / 2], float x, const float* __restrict__ w) {
    const float2v xx = {x, x};
#pragma unroll
    for (int j = 0; j < H; j += 4) {
        const float4 ww = *reinterpret_cast<const float4*>(w + j);
        const float2v w0 = {ww.x, ww.y}, w1 = {ww.z, ww.w};
        acc[j / 2] = __builtin_elementwise_fma(xx, w0, acc[j / 2]);
        acc[j / 2 + 1] = __builtin_elementwise_fma(xx, w1, acc[j / 2 + 1]);
    }
}

struct ActorArgs {
    const float* weights; const float* obs; float* scores;
    int64_t E; int A; int D; float noise_std; uint64_t seed; uint64_t seq; const uint64_t* seq_base;
};

template <int DT>   // DT = compile-time obs length (5, 8, 11, 14) or 0 = runtime D (obs row re-read from memory)
__global__ __launch_bounds__(TPB) void bsx_actor_kernel(const float* __restrict__ weights, const ActorArgs p) {
    const int D = DT > 0 ? DT : p.D;
    const int a = blockIdx.y;
    const int64_t e = int64_t(blockIdx.x) * TPB + threadIdx.x;
    const int P = blob_floats(D);
    // This workgroup's weights are wave-uniform addresses of read-only memory: the compiler fetches them with scalar
    // loads (s_load_dwordx8/x16 through the scalar cache) and feeds them to the packed FMAs as SGPR operands -- no LDS
    // traffic and no per-lane weight registers.
    const float* __restrict__ sw = weights + size_t(a) * P;
    const int64_t ec = e < p.E ? e : p.E - 1;
    const size_t row = size_t(ec) * p.A + a;
    float x[DT > 0 ? DT : 1];
    if (DT > 0) {
#pragma unroll
        for (int k = 0; k < DT; ++k) x[k] = p.obs[row * DT + k];
    }
    const float* W1 = sw;                 const float* b1 = W1 + D * H;   const float* g1 = b1 + H; const float* be1 = g1 + H;
    const float* W2 = be1 + H;            const float* b2 = W2 + H * H;   const float* g2 = b2 + H; const float* be2 = g2 + H;
    const float* W3 = be2 + H;            const float* b3 = W3 + H * NA;

    // ---- layer 1
    float2v acc[H / 2];
#pragma unroll
    for (int j = 0; j < H; j += 4) {
        const float4 bb = *reinterpret_cast<const float4*>(b1 + j);
        acc[j / 2] = float2v{bb.x, bb.y}; acc[j / 2 + 1] = float2v{bb.z, bb.w};
    }
    if (DT > 0) {
#pragma unroll
        for (int k = 0; k < DT; ++k) axpy64(acc, x[k], W1 + k * H);
    } else {
        for (int k = 0; k < D; ++k) axpy64(acc, p.obs[row * D + k], W1 + k * H);
    }
    float h[H];
#pragma unroll
    for (int j = 0; j < H / 2; ++j) { h[2 * j] = acc[j].x; h[2 * j + 1] = acc[j].y; }
    ln_relu(h, g1, be1);

    // ---- layer 2: 64 x 64, fully unrolled so that h[k] is a static register index
#pragma unroll
    for (int j = 0; j < H; j += 4) {
        const float4 bb = *reinterpret_cast<const float4*>(b2 + j);
        acc[j / 2] = float2v{bb.x, bb.y}; acc[j / 2 + 1] = float2v{bb.z, bb.w};
    }
#pragma unroll
    for (int k = 0; k < H; ++k) axpy64(acc, h[k], W2 + k * H);
#pragma unroll
    for (int j = 0; j < H / 2; ++j) { h[2 * j] = acc[j].x; h[2 * j + 1] = acc[j].y; }
    ln_relu(h, g2, be2);

    // ---- head: 64 -> 4, tanh
    float4 o = *reinterpret_cast<const float4*>(b3);
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const float4 ww = *reinterpret_cast<const float4*>(W3 + k * NA);
        o.x = fmaf(h[k], ww.x, o.x); o.y = fmaf(h[k], ww.y, o.y); o.z = fmaf(h[k], ww.z, o.z); o.w = fmaf(h[k], ww.w, o.w);
    }
    o.x = tanhf(o.x); o.y = tanhf(o.y); o.z = tanhf(o.z); o.w = tanhf(o.w);

    // ---- exploration noise + clamp (maddpg/agent.py:30-31), Gaussian via Philox + Box-Muller, keyed by (seed, seq, row)
    if (p.noise_std > 0.f) {
        const uint64_t seq = p.seq + (p.seq_base ? *p.seq_base : 0ull);   // seq_base: device word, so graph replays re-key
        const uint4 r = philox4x32_10(make_uint4(uint32_t(row), uint32_t(uint64_t(row) >> 32), uint32_t(seq), uint32_t(seq >> 32)),
                                      make_uint2(uint32_t(p.seed), uint32_t(p.seed >> 32) ^ 0xA5A5A5A5u));
        const float u0 = (float(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = float(r.y >> 8) * (1.0f / 16777216.0f);
        const float u2 = (float(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = float(r.w >> 8) * (1.0f / 16777216.0f);
        const float m0 = sqrtf(-2.0f * __logf(u0)), m1 = sqrtf(-2.0f * __logf(u2));
        float s0, c0, s1, c1;
        __sincosf(6.2831853071795864f * u1, &s0, &c0);
        __sincosf(6.2831853071795864f * u3, &s1, &c1);
        o.x = fminf(fmaxf(fmaf(p.noise_std, m0 * c0, o.x), -1.f), 1.f);
        o.y = fminf(fmaxf(fmaf(p.noise_std, m0 * s0, o.y), -1.f), 1.f);
        o.z = fminf(fmaxf(fmaf(p.noise_std, m1 * c1, o.z), -1.f), 1.f);
        o.w = fminf(fmaxf(fmaf(p.noise_std, m1 * s1, o.w), -1.f), 1.f);
    }
    if (e < p.E) reinterpret_cast<float4*>(p.scores)[row] = o;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

extern "C" {

int bsx_actor_blob_floats(int obs_len, int* floats_per_agent) {
    if (obs_len < 1 || obs_len > 3 * BSX_MAX_N + 2 || !floats_per_agent) return BSX_E_ARG;
    *floats_per_agent = blob_floats(obs_len);
    return 0;
}

int bsx_actor_forward(const float* weights, const float* obs, float* scores, int64_t E, int n, float noise_std,
                      uint64_t seed, uint64_t seq, const uint64_t* seq_base, void* stream) {
    if (!weights || !obs || !scores || E <= 0 || n < 1 || n > BSX_MAX_N) return BSX_E_ARG;
    if (!aligned(weights, 16) || !aligned(scores, 16) || !aligned(obs, 4)) return BSX_E_ALIGN;
    const int A = 2 * n, D = 3 * n + 2;
    ActorArgs a{weights, obs, scores, E, A, D, noise_std, seed, seq, seq_base};
    const dim3 grid(unsigned((E + TPB - 1) / TPB), unsigned(A)), block(TPB);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (n) {
        case 1: hipLaunchKernelGGL(bsx_actor_kernel<5>, grid, block, 0, s, weights, a); break;
        case 2: hipLaunchKernelGGL(bsx_actor_kernel<8>, grid, block, 0, s, weights, a); break;
        case 3: hipLaunchKernelGGL(bsx_actor_kernel<11>, grid, block, 0, s, weights, a); break;
        case 4: hipLaunchKernelGGL(bsx_actor_kernel<14>, grid, block, 0, s, weights, a); break;
        default: hipLaunchKernelGGL(bsx_actor_kernel<0>, grid, block, 0, s, weights, a); break;
    }
    return int(hipGetLastError());
}

}  // extern "C"
