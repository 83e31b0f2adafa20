// flatten_fast.h -- a DECISION procedure for flatten's subdivision test (flatten.wgsl:94-133 + :401), shared by
// k_flatten_items (device) and tools/flatten_fast_check.cpp (host, the same IEEE binary32 operations).
//
// The subdivision test of flatten_euler accepts an interval iff  v = fl(fl(err * chord_len) * scale) <= tol,  where err
// comes out of cubic_from_points_derivs: two atan2, two cos, two sin (pinned: binary64 sequences rounded once to
// binary32, dmath.h) and ~45 binary32 operations.  The VALUE of err is not kept -- only the comparison is -- and th0 / th1
// are only needed for ACCEPTED intervals.  ff_decide computes an estimate v~ of v without transcendentals together with a
// bound delta that is PROVEN to hold |v~ - v| <= delta (derivation: DESIGN.md section 5.1), and answers
//     FF_ACCEPT   v~ + delta + 1e-6 < tol   =>  v <= tol      FF_REJECT   v~ - delta - 1e-6 > tol   =>  v > tol
//     FF_UNSURE   anything else (the band around tol, obtuse or near-right angles, degenerate lengths, NaN / Inf):
//                 the caller runs the pinned sequence.
// The pinned numerics do not change: the oracle keeps evaluating the exact sequence for every node, and the parity suite
// compares the lines.  `make VARIANT=ffcheck EXTRA=-DFL_FAST_CHECK` builds a library that evaluates BOTH and counts
// contradictions (tools/soak_flatten_fast.sh).
//
// Domain of the estimate: cosines of the same sign and clear of zero (|c| >= 1e-3) and, for obtuse pairs, 1 + c >= 1/256
// (the quotients e = (2/3) / (1 + c) stay below 171); when the cosines have opposite signs (|c| >= 1e-3 each) the WGSL
// takes err = 2 and v = 2 chord_len scale is known to a few u (chord_len's 1-ulp root).
#pragma once

#ifndef FF_INLINE
#define FF_INLINE static inline
#endif
// Reciprocal and square root of the ESTIMATE: nothing here is kept, so the 1-ulp hardware instructions do (v_rcp_f32 /
// v_sqrt_f32: 1 instruction instead of the 12 / 18 of the IEEE sequences -- seven divisions and four roots per node).  The
// bound below charges every one of them 2 u (1 ulp).  The host tool defines them as the IEEE result moved by -1 / 0 / +1
// ulp at random: the bound has to hold for whatever a 1-ulp implementation returns.
#ifndef FF_RCP
#define FF_RCP(x) __builtin_amdgcn_rcpf(x)
#define FF_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif

namespace ffast {

enum { FF_UNSURE = 0, FF_ACCEPT = 1, FF_REJECT = 2 };

// atan2(y, x) from a = min / max of the magnitudes: odd polynomial of degree 13 in a (7 coefficients, minimax on
// [0, 1]: |a P(a^2) - atan a| <= 3.7e-7 for every binary32 a in [0, 1] incl. the evaluation's roundings -- checked
// exhaustively by the host tool), pi/2 - r for the steep half, pi - r for the left half plane.
// |result - RN(atan2(y, x))| <= FF_ET (measured 6e-7: polynomial 3.7e-7, the 1-ulp reciprocal, three roundings of values <= pi,
// the two constants).
#define FF_ET 2.0e-6f
FF_INLINE float ff_atan2_est(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const bool steep = ay > ax;
    const float num = steep ? ax : ay, den = steep ? ay : ax;
    const float a = num * FF_RCP(den);
    const float z = a * a;
    float p = 0.006811792496591806f;
    p = p * z + -0.0336042195558548f;
    p = p * z + 0.07962366938591003f;
    p = p * z + -0.1323334276676178f;
    p = p * z + 0.19807815551757812f;
    p = p * z + -0.3331736922264099f;
    p = p * z + 0.9999961256980896f;
    float r = a * p;
    if (steep) r = 1.5707963705062866f - r;
    if (x < 0.0f) r = 3.1415927410125732f - r;
    return __builtin_copysignf(r, y);
}

// Inputs: what cubic_from_points_derivs computes up to d0 / d1 -- h0, h1 with the very same operations, the rest through
// FF_RCP / FF_SQRT (relative errors against the pinned path's values: chord_len 3 u, len 4 u, d 12 u; section 5.1):
//   h0, h1        the end tangents rotated into the chord's frame (flatten.wgsl:111,114)
//   len0, len1    length(h0), length(h1)
//   d0, d1        len * (dt / chord_squared)
//   chord_len, scale, tol
// Output: the decision; *v_est / *delta for the check build and the host tool.
FF_INLINE int ff_decide(float h0x, float h0y, float len0, float h1x, float h1y, float len1, float d0, float d1, float chord_len, float scale,
                        float tol, float* v_est, float* delta) {
    *v_est = 0.0f;
    *delta = 0.0f;
    // lengths whose squares stay inside binary32's normal range (also refuses NaN / Inf)
    if (!(len0 >= 1e-18f && len0 <= 1e18f && len1 >= 1e-18f && len1 <= 1e18f)) return FF_UNSURE;
    const float r0 = FF_RCP(len0), r1 = FF_RCP(len1);
    const float c0 = h0x * r0, s0 = h0y * r0, c1 = h1x * r1, s1 = h1y * r1;  // |c~ - cos_(th)| <= 11 u, u = 2^-24 (section 5.1)
    const float CMIN = 1e-3f;
    const float m0 = 1.0f + c0, m1 = 1.0f + c1;
    const bool acute = c0 >= CMIN && c1 >= CMIN;
    const bool obtuse = c0 <= -CMIN && c1 <= -CMIN && m0 >= 0.00390625f && m1 >= 0.00390625f;
    if (!(acute || obtuse)) {
        // opposite signs, both clear of zero: fl(cth0 * cth1) < 0, the WGSL takes err = 2: v = fl(fl(2 chord_len) scale), known
        // up to chord_len's 1-ulp root (3 u) and two roundings on either side
        if ((c0 >= CMIN && c1 <= -CMIN) || (c0 <= -CMIN && c1 >= CMIN)) {
            float err = 2.0f;
            err *= chord_len;
            const float v = err * scale;
            const float dl = 6.0e-7f * v;  // 10 u v
            *v_est = v;
            *delta = dl;
            const float guard = dl + 1e-6f;
            if (v + guard < tol) return FF_ACCEPT;
            if (v - guard > tol) return FF_REJECT;
        }
        return FF_UNSURE;
    }
    const float t0 = ff_atan2_est(h0y, h0x), t1 = ff_atan2_est(h1y, h1x);
    // the WGSL's operations on the estimates (same expression tree as cubic_from_points_derivs; max(1 + c, 1e-9) is 1 + c here)
    const float TWO_THIRDS = (float)(2.0 / 3.0);
    const float e0 = TWO_THIRDS * FF_RCP(m0);
    const float e1 = TWO_THIRDS * FF_RCP(m1);
    const float s01 = c0 * s1 + c1 * s0;
    const float amin = 0.15f * (2.0f * e0 * s0 + 2.0f * e1 * s1 - e0 * e1 * s01);
    const float a = 0.15f * (2.0f * d0 * s0 + 2.0f * d1 * s1 - d0 * d1 * s01);
    const float aerr = __builtin_fabsf(a - amin);
    const float symm = __builtin_fabsf(t0 + t1);
    const float asymm = __builtin_fabsf(t0 - t1);
    const float dx = d0 - e0, dy = d1 - e1;
    const float dist = FF_SQRT(dx * dx + dy * dy);
    const float symm2 = symm * symm;
    const float ctr = (4.625e-6f * symm * symm2 + 7.5e-3f * asymm) * symm2;
    const float halo = (5e-3f * symm + 7e-2f * asymm) * dist;
    float err = ctr + 1.55f * aerr + halo;
    err *= chord_len;
    const float v = err * scale;
    // |v~ - v| <= u [(94 + 39 D + S (26 + S (26 + 1.4 S))) L + 13 v~],  D = d0 + d1 + d0 d1,  S = e0 + e1,  L = chord_len * scale
    // (section 5.1); 7.5e-8 = 1.25 u
    const float D = d0 + d1 + d0 * d1;
    const float S = e0 + e1;
    const float L = chord_len * scale;
    const float dl = 7.5e-8f * ((94.0f + 39.0f * D + S * (26.0f + S * (26.0f + 1.4f * S))) * L + 13.0f * v);
    *v_est = v;
    *delta = dl;
    const float guard = dl + 1e-6f;
    if (v + guard < tol) return FF_ACCEPT;
    if (v - guard > tol) return FF_REJECT;
    return FF_UNSURE;
}

}  // namespace ffast
