// Device-side manifold maps on register-resident row tiles.
//
// Each function restates one method of the reference (file:line in the comment)
// with the same clamp order and the same dtype rules.  A row is rescaled by ONE factor
// computed per row (x * (a/b) instead of the reference's elementwise a*x/b: <= 2 ulp
// apart, no per-element division).  Arithmetic in the row's dtype T, artanh/arcosh evaluated in fp64 and cast back
// (utils/math_utils.py:30-36,57-63), eps chosen by dtype
// (manifolds/poincare.py:23, manifolds/hyperboloid.py:21).
#pragma once
#include "hd_common.cuh"

namespace hd {

template <typename T> struct Lim;
template <> struct Lim<float> {
    static constexpr double pb_eps = 4e-3, hb_eps = 1e-7;
};
template <> struct Lim<double> {
    static constexpr double pb_eps = 1e-5, hb_eps = 1e-15;
};

constexpr double kMinNorm = 1e-15;

template <typename T> __device__ __forceinline__ T clamp_min(T x, T lo) { return x < lo ? lo : x; }
template <typename T> __device__ __forceinline__ T clamp_max(T x, T hi) { return x > hi ? hi : x; }
template <typename T> __device__ __forceinline__ T clamp15(T x) {
    return x < T(-15) ? T(-15) : (x > T(15) ? T(15) : x);
}
__device__ __forceinline__ float hd_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double hd_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float hd_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double hd_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float hd_cosh(float x) { return coshf(x); }
__device__ __forceinline__ double hd_cosh(double x) { return cosh(x); }
__device__ __forceinline__ float hd_sinh(float x) { return sinhf(x); }
__device__ __forceinline__ double hd_sinh(double x) { return sinh(x); }

// 1-ulp hardware reciprocal / square root for per-row factors that feed no cancellation
// (a relative error of 1e-7 in a scale factor is 100x below the parity bar); everything that
// enters artanh / arcosh keeps the correctly rounded forms.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double fast_div(double a, double b) { return a / b; }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }

// utils/math_utils.py:14-15
template <typename T> __device__ __forceinline__ T tanh15(T x) { return hd_tanh(clamp15(x)); }
template <typename T> __device__ __forceinline__ T cosh15(T x) { return hd_cosh(clamp15(x)); }
template <typename T> __device__ __forceinline__ T sinh15(T x) { return hd_sinh(clamp15(x)); }

// utils/math_utils.py:30-36: clamp to [-1+1e-15, 1-1e-15] in T (a no-op bound in fp32),
// then 0.5*(log(1+z) - log(1-z)) in fp64, cast back to T.
//
// fp32 rows: the reference's fp64 detour exists to dodge the cancellation of log(1+z) -
// log(1-z) / z*z-1 in fp32.  The same values (to <= 4 ulp of fp32, measured against the
// golden vectors) come from cancellation-free fp32 forms -- 1-z and z-1 are exact for
// z in [0.5, 2] -- at a fraction of the cost of two software fp64 logs per row:
//     artanh(z) = 0.5 * log1p(2z / (1 - z)),   arcosh(z) = log1p((z-1) + sqrt((z-1)(z+1))).
// fp64 rows keep the literal fp64 formula.
template <typename T> __device__ __forceinline__ T artanh_ref(T x) {
    const T lo = (T)(-1.0 + 1e-15), hi = (T)(1.0 - 1e-15);
    x = x < lo ? lo : (x > hi ? hi : x);
    const double z = (double)x;
    return (T)(0.5 * (log(1.0 + z) - log(1.0 - z)));
}
#ifndef HD_LITERAL_FP64_INVERSE
template <> __device__ __forceinline__ float artanh_ref<float>(float x) {
    x = x < -1.f ? -1.f : (x > 1.f ? 1.f : x);
    const float a = fabsf(x);
    const float r = 0.5f * log1pf(2.f * a / (1.f - a));
    return x < 0.f ? -r : r;
}
#endif
// utils/math_utils.py:57-63
template <typename T> __device__ __forceinline__ T arcosh_ref(T x) {
    const T lo = (T)(1.0 + 1e-15);
    x = x < lo ? lo : x;
    const double z = (double)x;
    double a = z + sqrt(z * z - 1.0);
    a = a < 1e-15 ? 1e-15 : a;
    return (T)log(a);
}
#ifndef HD_LITERAL_FP64_INVERSE
template <> __device__ __forceinline__ float arcosh_ref<float>(float x) {
    x = x < 1.f ? 1.f : x;
    const float t = x - 1.f;
    return log1pf(t + sqrtf(t * (x + 1.f)));
}
#endif

// sinh and cosh of a clamped argument (|x| <= 15) from one exp; sinh by its series below
// 0.25 where e^x - e^-x cancels
__device__ __forceinline__ void sinhcosh15(float x, float& sh, float& ch) {
    x = clamp15(x);
    const float a = fabsf(x), e = expf(a), ei = 1.f / e;
    ch = 0.5f * (e + ei);
    const float a2 = a * a;
    const float small = a * (1.f + a2 * (1.f / 6.f) * (1.f + a2 * (1.f / 20.f) * (1.f + a2 * (1.f / 42.f))));
    const float s = a < 0.25f ? small : 0.5f * (e - ei);
    sh = x < 0.f ? -s : s;
}
__device__ __forceinline__ void sinhcosh15(double x, double& sh, double& ch) {
    x = clamp15(x);
    sh = sinh(x);
    ch = cosh(x);
}

// ================================ Poincare ball ================================
template <typename T> struct PB {
    // manifolds/poincare.py:42-47
    template <class TL> static __device__ __forceinline__ void proj(TL& r, const Curv& cv) {
        const T n = clamp_min(hd_sqrt(r.sumsq()), (T)kMinNorm);
        const T maxnorm = (T)((1.0 - Lim<T>::pb_eps) / cv.sc);
        if (n > maxnorm) r.scale(maxnorm / n);
    }
    // manifolds/poincare.py:73-77
    template <class TL> static __device__ __forceinline__ void expmap0(TL& r, const Curv& cv) {
        const T sc = (T)cv.sc;
        const T n = clamp_min(hd_sqrt(r.sumsq()), (T)kMinNorm);
        r.scale(tanh15(sc * n) / (sc * n));
    }
    // manifolds/poincare.py:79-83
    template <class TL> static __device__ __forceinline__ void logmap0(TL& r, const Curv& cv) {
        const T sc = (T)cv.sc;
        const T n = clamp_min(hd_sqrt(r.sumsq()), (T)kMinNorm);
        const T f = T(1) / sc * artanh_ref(sc * n) / n;
        r.scale(f);
    }
    // manifolds/poincare.py:85-91 : x <- x (+) y
    template <class TL> static __device__ __forceinline__ void mobius_add(TL& x, const TL& y, const Curv& cv) {
        const T c = (T)cv.c;
        const T x2 = x.sumsq(), y2 = y.sumsq(), xy = x.dot(y);
        const T a = T(1) + T(2) * c * xy + c * y2;
        const T b = T(1) - c * x2;
        const T den = clamp_min(T(1) + T(2) * c * xy + c * c * x2 * y2, (T)kMinNorm);
        const T ad = a / den, bd = b / den;
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) x.v[i] = ad * x.v[i] + bd * y.v[i];
    }
    // manifolds/poincare.py:33-35
    template <class TL> static __device__ __forceinline__ T lambda_x(const TL& x, const Curv& cv) {
        return T(2) / clamp_min(T(1) - (T)cv.c * x.sumsq(), (T)kMinNorm);
    }
    // manifolds/poincare.py:55-64 : u <- expmap(u, p)
    template <class TL> static __device__ __forceinline__ void expmap(TL& u, const TL& p, const Curv& cv) {
        const T sc = (T)cv.sc;
        const T un = clamp_min(hd_sqrt(u.sumsq()), (T)kMinNorm);
        u.scale(tanh15(sc / T(2) * lambda_x(p, cv) * un) / (sc * un));
        TL q = p;
        mobius_add(q, u, cv);
        u = q;
    }
    // manifolds/poincare.py:66-71 : p2 <- logmap(p1, p2)
    template <class TL> static __device__ __forceinline__ void logmap(const TL& p1, TL& p2, const Curv& cv) {
        const T sc = (T)cv.sc;
        TL s = p1;
        s.map([](T x) { return -x; });
        mobius_add(s, p2, cv);
        const T sn = clamp_min(hd_sqrt(s.sumsq()), (T)kMinNorm);
        const T lam = lambda_x(p1, cv);
        s.scale(T(2) / sc / lam * artanh_ref(sc * sn) / sn);
        p2 = s;
    }
    // manifolds/poincare.py:25-31
    template <class TL> static __device__ __forceinline__ T sqdist(const TL& p1, const TL& p2, const Curv& cv) {
        const T sc = (T)cv.sc;
        TL s = p1;
        s.map([](T x) { return -x; });
        mobius_add(s, p2, cv);
        const T d = artanh_ref(sc * hd_sqrt(s.sumsq())) * T(2) / sc;
        return d * d;
    }
    // manifolds/poincare.py:93-102 after mx = x @ W^T: r holds mx, xn = clamp_min(|x|)
    template <class TL> static __device__ __forceinline__ void matvec_epilogue(TL& r, T xn, const Curv& cv) {
        const T sc = (T)cv.sc;
        const bool nz = r.any_nonzero();
        const T mxn = clamp_min(hd_sqrt(r.sumsq()), (T)kMinNorm);
        if (nz) r.scale(tanh15(mxn / xn * artanh_ref(sc * xn)) / (mxn * sc));
        else r.zero();
    }
    template <class TL> static __device__ __forceinline__ void proj_tan0(TL&, const Curv&) {}
};

// ================================= Hyperboloid =================================
// Column 0 of every row is the time coordinate.
template <typename T> struct HB {
    // manifolds/hyperboloid.py:43-52
    template <class TL> static __device__ __forceinline__ void proj(TL& r, const Curv& cv) {
        const T n = hd_sqrt(r.sumsq(1));
        const T x0 = hd_sqrt(clamp_min((T)cv.K + n * n, (T)Lim<T>::hb_eps));
        r.set_first(x0);
    }
    // manifolds/hyperboloid.py:64-68
    template <class TL> static __device__ __forceinline__ void proj_tan0(TL& r, const Curv&) {
        if (r.j == 0) r.v[0] = r.v[0] - r.v[0];
    }
    // manifolds/hyperboloid.py:90-101 (ends with proj)
    template <class TL> static __device__ __forceinline__ void expmap0(TL& r, const Curv& cv) {
        const T sK = (T)cv.sK;
        const T n = clamp_min(hd_sqrt(r.sumsq(1)), (T)kMinNorm);
        const T theta = n / sK;
        T shv, chv;
        sinhcosh15(theta, shv, chv);
        r.scale(sK * shv / n);
        r.set_first(sK * chv);
        proj(r, cv);
    }
    // manifolds/hyperboloid.py:103-113
    template <class TL> static __device__ __forceinline__ void logmap0(TL& r, const Curv& cv) {
        const T sK = (T)cv.sK;
        const T n = clamp_min(hd_sqrt(r.sumsq(1)), (T)kMinNorm);
        const T theta = clamp_min(r.first() / sK, (T)(1.0 + Lim<T>::hb_eps));
        r.scale(sK * arcosh_ref(theta) / n);
        r.set_first(T(0));
    }
    // manifolds/hyperboloid.py:35-41
    template <class TL> static __device__ __forceinline__ T sqdist(const TL& x, const TL& y, const Curv& cv) {
        const T K = (T)cv.K;
        const T x0 = x.first(), y0 = y.first();
        // sum over all columns minus twice the time product (hyperboloid.py:26)
        const T prod = x.dot(y) - T(2) * x0 * y0;
        const T theta = clamp_min(-prod / K, (T)(1.0 + Lim<T>::hb_eps));
        const T a = arcosh_ref(theta);
        return clamp_max(K * (a * a), T(50));
    }
};

// ---- activations applied in the tangent space (layers/hyp_layers.py:171) -------
enum Act { ACT_IDENTITY = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_ELU = 4, ACT_LEAKY_RELU = 5 };

template <typename T> __device__ __forceinline__ T apply_act(T x, int act) {
    switch (act) {
        case ACT_RELU: return x > T(0) ? x : (x != x ? x : T(0));
        case ACT_TANH: return hd_tanh(x);
        case ACT_SIGMOID: return T(1) / (T(1) + (T)exp((double)-x));
        case ACT_ELU: return x > T(0) ? x : (T)expm1((double)x);
        case ACT_LEAKY_RELU: return x > T(0) ? x : T(0.01) * x;
        default: return x;
    }
}

// HypAct for a positively homogeneous activation (identity / relu / leaky_relu).
// act(f*x) = f*act(x) for the positive per-row factor f of logmap0, so every norm of the
// chain logmap0 -> act -> expmap0 -> proj follows from TWO sums over the input row,
// s1 = |x|^2 and s2 = |act(x)|^2 (|f*u| = f*|u|: the literal recomputation differs by
// rounding only), and the row is rescaled once: one sweep of reductions instead of four
// dependent ones, which is what kept this map VALU-bound instead of HBM-bound.
template <typename T, class TL>
__device__ __forceinline__ void hyp_act_homog(PB<T>*, TL& r, const Curv& cin, const Curv& cout, T slope) {
    T s1, s2;
    r.sumsq_pair(0, slope, s1, s2);
    const T sci = (T)cin.sc, sco = (T)cout.sc;
    const T n1 = clamp_min(hd_sqrt(s1), (T)kMinNorm);
    const T f1 = fast_div((T)(1.0 / cin.sc) * artanh_ref(sci * n1), n1);  // logmap0 (poincare.py:79-83)
    const T m = f1 * fast_sqrt(s2);                                       // |act(logmap0(x))|
    const T n2 = clamp_min(m, (T)kMinNorm);
    const T f2 = fast_div(tanh15(sco * n2), sco * n2);                    // expmap0 (:73-77)
    const T n3 = clamp_min(f2 * m, (T)kMinNorm);
    const T maxnorm = (T)((1.0 - Lim<T>::pb_eps) / cout.sc);
    const T f3 = n3 > maxnorm ? fast_div(maxnorm, n3) : T(1);             // proj (:42-47)
    const T f = f1 * f2 * f3;
    r.map([=](T x) { return f * (x > T(0) ? x : (x != x ? x : slope * x)); });
}
template <typename T, class TL>
__device__ __forceinline__ void hyp_act_homog(HB<T>*, TL& r, const Curv& cin, const Curv& cout, T slope) {
    T s1, s2;
    const T x0 = r.first();
    r.sumsq_pair(1, slope, s1, s2);
    const T sKi = (T)cin.sK, sKo = (T)cout.sK;
    const T n1 = clamp_min(hd_sqrt(s1), (T)kMinNorm);
    const T theta = clamp_min(x0 / sKi, (T)(1.0 + Lim<T>::hb_eps));
    const T f1 = fast_div(sKi * arcosh_ref(theta), n1);                   // logmap0 (hyperboloid.py:103-113)
    const T m = f1 * fast_sqrt(s2);                                       // |act(logmap0(x))| (spatial)
    const T n2 = clamp_min(m, (T)kMinNorm);
    T shv, chv;
    sinhcosh15(n2 * (T)(1.0 / cout.sK), shv, chv);                        // expmap0 (:90-101)
    const T f2 = fast_div(sKo * shv, n2);
    const T ns = f2 * m;                                                  // spatial norm after expmap0
    const T t0 = hd_sqrt(clamp_min((T)cout.K + ns * ns, (T)Lim<T>::hb_eps));   // proj (:43-52), idempotent
    const T f = f1 * f2;
    r.map([=](T x) { return f * (x > T(0) ? x : (x != x ? x : slope * x)); });
    r.set_first(t0);
}

// HypAgg's whole tail for a positively homogeneous activation, as ONE factor: expmap0(c_in) -> proj ->
// HypAct(c_in, c_out) of a tangent row u with s = |u|^2 and sr = |act(u)|^2 (spatial part on the hyperboloid)
// is f * act(u), time coordinate t0 (hyperboloid).  Every intermediate norm follows analytically from s and sr
// (|g u| = g |u| for the positive factors g: rounding-level difference against re-measuring the row, as in
// hyp_act_homog); same clamp order as expmap0 / proj / hyp_act_homog above.  Scalars only, so a wavefront can
// evaluate it for several rows at once, one row per lane.
template <typename T>
__device__ __forceinline__ void agg_act_scalars(PB<T>*, T s, T sr, const Curv& cin, const Curv& cout, T& f, T& t0) {
    const T sci = (T)cin.sc, sco = (T)cout.sc;
    const T n0 = hd_sqrt(s);
    const T n = clamp_min(n0, (T)kMinNorm);
    const T fe = tanh15(sci * n) / (sci * n);                              // expmap0 (poincare.py:73-77)
    const T np = clamp_min(fe * n0, (T)kMinNorm);
    const T maxi = (T)((1.0 - Lim<T>::pb_eps) / cin.sc);
    const T g = np > maxi ? fe * (maxi / np) : fe;                         // proj (:42-47)
    const T n1 = clamp_min(g * n0, (T)kMinNorm);
    const T f1 = fast_div((T)(1.0 / cin.sc) * artanh_ref(sci * n1), n1);   // logmap0 (:79-83)
    const T m = f1 * g * fast_sqrt(sr);                                    // |act(logmap0(.))|
    const T n2 = clamp_min(m, (T)kMinNorm);
    const T f2 = fast_div(tanh15(sco * n2), sco * n2);                     // expmap0
    const T n3 = clamp_min(f2 * m, (T)kMinNorm);
    const T maxo = (T)((1.0 - Lim<T>::pb_eps) / cout.sc);
    const T f3 = n3 > maxo ? fast_div(maxo, n3) : T(1);                    // proj
    f = g * f1 * f2 * f3;
    t0 = T(0);
}
template <typename T>
__device__ __forceinline__ void agg_act_scalars(HB<T>*, T s, T sr, const Curv& cin, const Curv& cout, T& f, T& t0) {
    const T sKi = (T)cin.sK, sKo = (T)cout.sK;
    const T n0 = hd_sqrt(s);
    const T n = clamp_min(n0, (T)kMinNorm);
    T shv, chv;
    sinhcosh15(n / sKi, shv, chv);                                         // expmap0 (hyperboloid.py:90-101)
    const T fe = sKi * shv / n;
    const T m1 = fe * n0;
    const T x0 = hd_sqrt(clamp_min((T)cin.K + m1 * m1, (T)Lim<T>::hb_eps)); // proj (:43-52)
    const T n1 = clamp_min(m1, (T)kMinNorm);
    const T theta = clamp_min(x0 / sKi, (T)(1.0 + Lim<T>::hb_eps));
    const T f1 = fast_div(sKi * arcosh_ref(theta), n1);                    // logmap0 (:103-113)
    const T m = f1 * fe * fast_sqrt(sr);
    const T n2 = clamp_min(m, (T)kMinNorm);
    sinhcosh15(n2 * (T)(1.0 / cout.sK), shv, chv);                         // expmap0
    const T f2 = fast_div(sKo * shv, n2);
    const T ns = f2 * m;
    t0 = hd_sqrt(clamp_min((T)cout.K + ns * ns, (T)Lim<T>::hb_eps));       // proj
    f = fe * f1 * f2;
}
// hyp_act_homog's chain on scalars: s1 = |x|^2, s2 = |act(x)|^2 (spatial part on the hyperboloid, x0 its time
// coordinate): HypAct(x) = f * act(x), time coordinate t0
template <typename T>
__device__ __forceinline__ void hyp_act_scalars(PB<T>*, T s1, T s2, T, const Curv& cin, const Curv& cout, T& f, T& t0) {
    const T sci = (T)cin.sc, sco = (T)cout.sc;
    const T n1 = clamp_min(hd_sqrt(s1), (T)kMinNorm);
    const T f1 = fast_div((T)(1.0 / cin.sc) * artanh_ref(sci * n1), n1);
    const T m = f1 * fast_sqrt(s2);
    const T n2 = clamp_min(m, (T)kMinNorm);
    const T f2 = fast_div(tanh15(sco * n2), sco * n2);
    const T n3 = clamp_min(f2 * m, (T)kMinNorm);
    const T maxnorm = (T)((1.0 - Lim<T>::pb_eps) / cout.sc);
    const T f3 = n3 > maxnorm ? fast_div(maxnorm, n3) : T(1);
    f = f1 * f2 * f3;
    t0 = T(0);
}
template <typename T>
__device__ __forceinline__ void hyp_act_scalars(HB<T>*, T s1, T s2, T x0, const Curv& cin, const Curv& cout, T& f, T& t0) {
    const T sKi = (T)cin.sK, sKo = (T)cout.sK;
    const T n1 = clamp_min(hd_sqrt(s1), (T)kMinNorm);
    const T theta = clamp_min(x0 / sKi, (T)(1.0 + Lim<T>::hb_eps));
    const T f1 = fast_div(sKi * arcosh_ref(theta), n1);
    const T m = f1 * fast_sqrt(s2);
    const T n2 = clamp_min(m, (T)kMinNorm);
    T shv, chv;
    sinhcosh15(n2 * (T)(1.0 / cout.sK), shv, chv);
    const T f2 = fast_div(sKo * shv, n2);
    const T ns = f2 * m;
    t0 = hd_sqrt(clamp_min((T)cout.K + ns * ns, (T)Lim<T>::hb_eps));
    f = f1 * f2;
}
template <typename T> struct IsHB { static constexpr bool value = false; };
template <typename T> struct IsHB<HB<T>> { static constexpr bool value = true; };

// layers/hyp_layers.py:170-173: logmap0(c_in) -> act -> proj_tan0 -> expmap0(c_out) -> proj(c_out)
template <class M, typename T, class TL>
__device__ __forceinline__ void hyp_act(TL& r, const Curv& cin, const Curv& cout, int act) {
    if (act == ACT_IDENTITY || act == ACT_RELU || act == ACT_LEAKY_RELU) {
        hyp_act_homog<T>((M*)nullptr, r, cin, cout, act == ACT_IDENTITY ? T(1) : (act == ACT_RELU ? T(0) : T(0.01)));
        return;
    }
    M::logmap0(r, cin);
    r.map([=](T x) { return apply_act(x, act); });
    M::proj_tan0(r, cout);
    M::expmap0(r, cout);
    M::proj(r, cout);
}

}  // namespace hd
