// tests/native/flatten_fast_check.cpp -- TEST INFRASTRUCTURE.  Holds ffast::ff_decide (jello_amd/csrc/flatten_fast.h: the
// transcendental-free decision of flatten's subdivision test) to the oracle's pinned sequence (oracle/oracle.cpp,
// fl::cubic_from_points_derivs) with the same IEEE binary32 operations the device executes:
//   1. every binary32 a in [0, 1]: |ff_atan_acute - atan| (the polynomial incl. its evaluation's roundings);
//   2. subdivision trees of random cubics (C3-like control points, and wide / tiny / nearly straight / cusped ones,
//      scales 0.02 ... 40): at every node |v~ - v| <= delta and no decision contradicts v <= tol; rates of the outcomes;
//   3. adversarial operands: angles uniformly over the acute domain incl. its borders, d0 / d1, chord_len and scale
//      log-uniform over 12 decades.
// usage: flatten_fast_check [n_cubics] [n_adversarial] [exhaustive 0/1]   (exit code 1 on any violation)
#include "../../oracle/oracle.cpp"
#define FF_INLINE static inline
// The device uses v_rcp_f32 / v_sqrt_f32 (1 ulp).  Here: the IEEE result moved by -1, 0 or +1 ulp at random, so that the bound
// is held to ANY 1-ulp implementation, not to one.
static thread_local uint64_t ff_rng_state = 0x9E3779B97F4A7C15ull;
static inline float ff_wobble(float x) {
    ff_rng_state = ff_rng_state * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t r = (uint32_t)(ff_rng_state >> 33) % 3u;
    if (!(x == x) || std::isinf(x) || x == 0.0f) return x;
    return r == 0u ? x : (r == 1u ? std::nextafterf(x, INFINITY) : std::nextafterf(x, -INFINITY));
}
#define FF_RCP(x) ff_wobble(1.0f / (x))
#define FF_SQRT(x) ff_wobble(std::sqrt(x))
#include "../../jello_amd/csrc/flatten_fast.h"

#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <random>

using fl::PointDeriv;

struct Stats {
    uint64_t nodes = 0, acc = 0, rej = 0, unsure = 0, exact_sign = 0, contradictions = 0, bound_violations = 0, tiny_chord = 0;
    double max_ratio = 0.0, max_rel_band = 0.0;
};

// one node: the values both paths share, then both decisions.  Returns the exact decision.
static bool test_node(V2 lp, V2 p1, V2 q0, V2 q1, float dt, float scale, Stats& st) {
    const float tol = 0.25f;
    fl::CubicParams cp = fl::cubic_from_points_derivs(lp, p1, q0, q1, dt);
    const float v = cp.err * scale;
    const bool accept = v <= tol;
    st.nodes++;
    V2 chord = p1 - lp;
    float chord_squared = dot(chord, chord);
    if (chord_squared < fl::DERIV_THRESH_SQUARED) { st.tiny_chord++; return accept; }  // (this branch has no transcendentals)
    // (as k_flatten_items' node_test_fast computes them: the roots and the quotient through the 1-ulp operations)
    float chord_len = FF_SQRT(chord_squared);
    float sc = dt * FF_RCP(chord_squared);
    V2 h0 = v2(q0.x * chord.x + q0.y * chord.y, q0.y * chord.x - q0.x * chord.y);
    V2 h1 = v2(q1.x * chord.x + q1.y * chord.y, q1.x * chord.y - q1.y * chord.x);
    float len0 = FF_SQRT(h0.x * h0.x + h0.y * h0.y), len1 = FF_SQRT(h1.x * h1.x + h1.y * h1.y);
    float d0 = len0 * sc, d1 = len1 * sc;
    float ve, dl;
    int k = ffast::ff_decide(h0.x, h0.y, len0, h1.x, h1.y, len1, d0, d1, chord_len, scale, tol, &ve, &dl);
    if (k == ffast::FF_UNSURE) st.unsure++;
    if (k == ffast::FF_ACCEPT) { st.acc++; if (!accept) st.contradictions++; }
    if (k == ffast::FF_REJECT) { st.rej++; if (accept) st.contradictions++; }
    if (dl > 0.0f) {
        double diff = std::fabs((double)ve - (double)v);
        if (!(diff <= (double)dl)) st.bound_violations++;
        double r = diff / (double)dl;
        if (r > st.max_ratio) st.max_ratio = r;
        if (v > 0.05f && v < 1.25f && dl / v > st.max_rel_band) st.max_rel_band = dl / v;
        if (ve == 2.0f * chord_len * scale) st.exact_sign++;  // (the err = 2 arm)
    } else if (k != ffast::FF_UNSURE && ve != 0.0f) {
        st.bound_violations++;  // a decision without a bound
    }
    return accept;
}

// the sequential walk of flatten.wgsl:362-403 over one cubic (decisions by the exact path), testing every node
static void walk(V2 p0, V2 p1, V2 p2, V2 p3, float scale, Stats& st) {
    if (veq(p0, p1) && veq(p0, p2) && veq(p0, p3)) return;
    uint32_t t0_u = 0u;
    float dt = 1.0f;
    V2 last_p = p0, last_q = p1 - p0;
    if (dot(last_q, last_q) < fl::DERIV_THRESH_SQUARED) last_q = fl::eval_cubic_and_deriv(p0, p1, p2, p3, fl::DERIV_EPS).deriv;
    float last_t = 0.0f;
    for (;;) {
        float t0 = (float)t0_u * dt;
        if (t0 == 1.0f) break;
        float t1 = t0 + dt;
        PointDeriv pq1 = fl::eval_cubic_and_deriv(p0, p1, p2, p3, t1);
        if (dot(pq1.deriv, pq1.deriv) < fl::DERIV_THRESH_SQUARED) {
            PointDeriv n = fl::eval_cubic_and_deriv(p0, p1, p2, p3, t1 - fl::DERIV_EPS);
            pq1.deriv = n.deriv;
            if (t1 < 1.0f) { pq1.point = n.point; t1 = t1 - fl::DERIV_EPS; }
        }
        float actual_dt = t1 - last_t;
        bool accept = test_node(last_p, pq1.point, last_q, pq1.deriv, actual_dt, scale, st);
        if (accept || dt <= fl::SUBDIV_LIMIT) {
            last_p = pq1.point; last_q = pq1.deriv; last_t = t1;
            t0_u += 1u;
            uint32_t shift = t0_u == 0u ? 32u : (uint32_t)__builtin_ctz(t0_u);
            t0_u = shift >= 32u ? 0u : (t0_u >> shift);
            dt *= (float)(1u << (shift & 31u));
        } else {
            t0_u *= 2u;
            dt *= 0.5f;
        }
    }
}

static void report(const char* what, const Stats& s) {
    std::printf("%-26s nodes %10llu  accept %6.2f%%  reject %6.2f%%  unsure %6.3f%%  (err=2 exact %5.2f%%, tiny chord %llu)  "
                "max |dv|/delta %.3f  widest band delta/v %.2e  contradictions %llu  bound violations %llu\n",
                what, (unsigned long long)s.nodes, 100.0 * s.acc / (double)s.nodes, 100.0 * s.rej / (double)s.nodes,
                100.0 * s.unsure / (double)s.nodes, 100.0 * s.exact_sign / (double)s.nodes, (unsigned long long)s.tiny_chord, s.max_ratio,
                s.max_rel_band, (unsigned long long)s.contradictions, (unsigned long long)s.bound_violations);
}

int main(int argc, char** argv) {
    const uint64_t n_cubics = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 200000ull;
    const uint64_t n_adv = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 2000000ull;
    const bool exhaustive = argc > 3 ? std::atoi(argv[3]) != 0 : false;
    int bad = 0;
    // ---- 1. the arctangent ----
    {
        double worst = 0.0, worst_vs_pinned = 0.0;
        uint64_t step = exhaustive ? 1u : 4099u;
        const uint32_t one = f2u(1.0f);
#pragma omp parallel for reduction(max : worst, worst_vs_pinned) schedule(static)
        for (int64_t u = 0; u <= (int64_t)one; u += (int64_t)step) {
            float a = u2f((uint32_t)u);
            // all eight octants: (y, x) = (+-a, +-1) and (+-1, +-a)
            double e1 = 0.0, p1 = 0.0;
            for (int o = 0; o < 8; o++) {
                float y = (o & 4) ? 1.0f : a, x = (o & 4) ? a : 1.0f;
                if (o & 1) y = -y;
                if (o & 2) x = -x;
                if (a == 0.0f && (o & 4)) continue;  // (x = +-0: the pinned sequence looks at the sign bit, the estimate at x < 0 -- outside ff_decide's domain)
                const float est = ffast::ff_atan2_est(y, x);
                e1 = std::fmax(e1, std::fabs((double)est - std::atan2((double)y, (double)x)));
                p1 = std::fmax(p1, std::fabs((double)est - (double)atan2_(y, x)));
            }
            double e2 = 0.0, e3 = 0.0, p2 = 0.0;
            double e = std::fmax(e1, std::fmax(e2, e3));
            if (e > worst) worst = e;
            if (std::fmax(p1, p2) > worst_vs_pinned) worst_vs_pinned = std::fmax(p1, p2);
        }
        std::printf("atan: %s binary32 ratios in [0,1]: max |ff_atan2_est - atan2| over the 8 octants = %.3e, vs the pinned atan2_ = %.3e (FF_ET = %.1e)\n",
                    exhaustive ? "ALL" : "every 4099th of the", worst, worst_vs_pinned, (double)FF_ET);
        if (!(worst_vs_pinned <= (double)FF_ET)) bad = 1;
    }
    // ---- 2. subdivision trees ----
    struct Family { const char* name; float span, handle; float smin, smax; };
    const Family fam[] = {
        {"C3-like (+-32 px)", 4096.0f, 32.0f, 1.0f, 1.0f},
        {"wide (+-1000 px)", 4096.0f, 1000.0f, 0.5f, 4.0f},
        {"tiny (+-0.5 px)", 64.0f, 0.5f, 0.02f, 40.0f},
        {"nearly straight", 4096.0f, 200.0f, 1.0f, 1.0f},
        {"cusps / loops", 1024.0f, 100.0f, 0.1f, 10.0f},
    };
    for (int f = 0; f < 5; f++) {
        Stats tot;
#pragma omp parallel
        {
            Stats st;
            std::mt19937_64 rng(0x6A656C6C6Full + 977ull * (uint64_t)f + 131ull * (uint64_t)omp_get_thread_num());
            std::uniform_real_distribution<float> U(0.0f, 1.0f);
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n_cubics; i++) {
                const Family& F = fam[f];
                V2 a = v2(U(rng) * F.span, U(rng) * F.span);
                auto off = [&]() { return v2((U(rng) * 2.0f - 1.0f) * F.handle, (U(rng) * 2.0f - 1.0f) * F.handle); };
                V2 p0 = a, p1 = a + off(), p2 = a + off(), p3 = a + off();
                if (f == 3) {  // control points within 1e-3 ... 1 px of a straight line
                    V2 d = off();
                    float w = std::exp(U(rng) * 7.0f - 7.0f);
                    p1 = a + d * 0.33f + v2(-d.y, d.x) * (w * (U(rng) - 0.5f) / (1.0f + length(d)));
                    p2 = a + d * 0.66f + v2(-d.y, d.x) * (w * (U(rng) - 0.5f) / (1.0f + length(d)));
                    p3 = a + d;
                }
                if (f == 4) {  // handles crossing over: loops, cusps, end tangents pointing backwards
                    V2 d = off();
                    p3 = a + d * 0.1f;
                    p1 = a + off();
                    p2 = p3 + off();
                    if ((i & 7) == 0) p1 = p0;
                    if ((i & 15) == 1) p2 = p3;
                }
                float scale = F.smin * std::exp(U(rng) * std::log(F.smax / F.smin));
                walk(p0, p1, p2, p3, scale, st);
            }
#pragma omp critical
            {
                tot.nodes += st.nodes; tot.acc += st.acc; tot.rej += st.rej; tot.unsure += st.unsure; tot.exact_sign += st.exact_sign;
                tot.contradictions += st.contradictions; tot.bound_violations += st.bound_violations; tot.tiny_chord += st.tiny_chord;
                tot.max_ratio = std::fmax(tot.max_ratio, st.max_ratio); tot.max_rel_band = std::fmax(tot.max_rel_band, st.max_rel_band);
            }
        }
        report(fam[f].name, tot);
        if (tot.contradictions || tot.bound_violations) bad = 1;
    }
    // ---- 3. adversarial operands ----
    {
        Stats tot;
#pragma omp parallel
        {
            Stats st;
            std::mt19937_64 rng(0xADull + 7919ull * (uint64_t)omp_get_thread_num());
            std::uniform_real_distribution<double> U(0.0, 1.0);
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n_adv; i++) {
                // construct p, q vectors that give chosen angles / magnitudes: chord along a random direction
                double chord_len = std::exp(U(rng) * 27.6 - 13.8), ang = U(rng) * 6.283185307179586;
                auto pick_angle = [&]() {
                    double r = U(rng);
                    const double H = 1.5707963267948966;
                    if (r < 0.25) return (U(rng) * 2.0 - 1.0) * H;                                  // anywhere acute
                    if (r < 0.45) return (U(rng) < 0.5 ? -1.0 : 1.0) * (H - std::exp(-U(rng) * 16.0));  // towards +-pi/2
                    if (r < 0.65) return (U(rng) * 2.0 - 1.0) * std::exp(-U(rng) * 16.0);              // towards 0
                    if (r < 0.8) return (U(rng) < 0.5 ? -1.0 : 1.0) * (0.7853981633974483 + (U(rng) - 0.5) * std::exp(-U(rng) * 16.0));  // the fix-up seam
                    if (r < 0.9) return (U(rng) < 0.5 ? -1.0 : 1.0) * (3.141592653589793 - std::exp(-U(rng) * 12.0) * 1.5);  // obtuse, towards pi
                    return (U(rng) * 2.0 - 1.0) * 3.141592653589793;                                // anything
                };
                double th0 = pick_angle(), th1 = pick_angle();
                double m0 = std::exp(U(rng) * 27.6 - 13.8), m1 = std::exp(U(rng) * 27.6 - 13.8);
                if (i % 3 == 0) { m0 = chord_len * std::exp(U(rng) * 4.0 - 2.0); m1 = chord_len * std::exp(U(rng) * 4.0 - 2.0); }  // d0, d1 ~ 1
                V2 lp = v2((float)(U(rng) * 4096.0), (float)(U(rng) * 4096.0));
                if (i % 5 == 0) lp = v2(0.0f, 0.0f);
                V2 p1 = lp + v2((float)(chord_len * std::cos(ang)), (float)(chord_len * std::sin(ang)));
                V2 q0 = v2((float)(m0 * std::cos(ang + th0)), (float)(m0 * std::sin(ang + th0)));
                V2 q1 = v2((float)(m1 * std::cos(ang - th1)), (float)(m1 * std::sin(ang - th1)));
                float dt = (float)std::exp2(-(double)(rng() % 10));
                float scale = (float)std::exp(U(rng) * 7.6 - 3.9);
                test_node(lp, p1, q0, q1, dt, scale, st);
            }
#pragma omp critical
            {
                tot.nodes += st.nodes; tot.acc += st.acc; tot.rej += st.rej; tot.unsure += st.unsure; tot.exact_sign += st.exact_sign;
                tot.contradictions += st.contradictions; tot.bound_violations += st.bound_violations; tot.tiny_chord += st.tiny_chord;
                tot.max_ratio = std::fmax(tot.max_ratio, st.max_ratio); tot.max_rel_band = std::fmax(tot.max_rel_band, st.max_rel_band);
            }
        }
        report("adversarial operands", tot);
        if (tot.contradictions || tot.bound_violations) bad = 1;
    }
    std::printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
