// bsx_geometry.h -- plane kinematics helpers and the observation geometry (range, angle-off) in binary64: atan2 / sqrt on integer pixel differences
// Part of the step() path of libbattlespace_hip.so (included by bsx_kernels.hip, in this order: bsx_state.h, bsx_rng.h, bsx_geometry.h,
// bsx_instinct.h, bsx_step_kernel.h) and by the three translation units that instantiate the step kernels; namespace bsxk.
#pragma once

namespace bsxk {

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
static __constant__ double ATAN2_COEF[20] = {   // (one copy per translation unit: no device linking)
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
    constexpr bool TABLE = K <= ATAN_TABLE_MAX_K;      // measured: 1v1 (K = 2) 7.40 -> 7.34 us; 4v4 (K = 3) 24.0 -> 24.4, so literals there
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

// Range / angle-off pair of one observer->target (battle_env.py:230-231,240-241)
__device__ inline void obs_pair(int x, int y, double dir, int tx, int ty, float& od, float& oa) {
    od = obs_dist(x, y, tx, ty);
    oa = obs_angle(x, y, dir, tx, ty);
}

}  // namespace bsxk
