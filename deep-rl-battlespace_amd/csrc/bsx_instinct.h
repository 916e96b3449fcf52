// bsx_instinct.h -- the scripted 'instinct' opponent's action from one observation row (instinct/agent.py:10-62)
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h) and by the three translation units that instantiate the step kernels; namespace bsxk.
#pragma once

namespace bsxk {

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

}  // namespace bsxk
