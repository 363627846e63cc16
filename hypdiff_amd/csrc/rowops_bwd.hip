// C-ABI entry points: backward (vector-Jacobian products) of the fused per-row manifold maps.
//
// Every forward row kernel computes  out = factor(norms) * row  (plus, on the hyperboloid, a
// time coordinate that is a function of the spatial norm).  Its VJP is
//     d_row = factor * d_out + (d factor / d norm) * <row, d_out> / norm * row  (+ time terms)
// so a backward kernel is one read of the saved row, one read of the incoming gradient and
// one write -- the same HBM traffic class as the forward map.  The factor and its partial
// derivatives come from evaluating the forward's scalar chain on dual numbers (hd_dual.cuh).
// Reference: autograd of manifolds/poincare.py:42-47,73-83,93-102, manifolds/hyperboloid.py:
// 43-52,90-113,120-123, layers/hyp_layers.py:98-108,133-152,170-173 as lp_train.py:66-82 runs it.
#include "hd_rowkernels.cuh"
#include "hd_dual.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

inline Curv mk(const hd_curv& c) { return Curv{c.c, c.sqrt_c, c.K, c.sqrt_K}; }

// ---- Poincare-ball scalar chains (S = Dual<T,NV>) ----------------------------------------
template <typename T, class S> __device__ __forceinline__ S pb_proj_f(const S& rn, const Curv& cv) {
    const S n = d_clamp_min(rn, (T)kMinNorm);
    const T maxnorm = (T)((1.0 - Lim<T>::pb_eps) / cv.sc);
    if (n.v > maxnorm) return S(maxnorm) / n;
    return S(T(1));
}
template <typename T, class S> __device__ __forceinline__ S pb_expmap0_f(const S& rn, const Curv& cv) {
    const T sc = (T)cv.sc;
    const S n = d_clamp_min(rn, (T)kMinNorm);
    return d_tanh15(n * sc) / (n * sc);
}
template <typename T, class S> __device__ __forceinline__ S pb_logmap0_f(const S& rn, const Curv& cv) {
    const T sc = (T)cv.sc;
    const S n = d_clamp_min(rn, (T)kMinNorm);
    return d_artanh(n * sc) * (T(1) / sc) / n;
}

// g <- f*g + (k/norm)*s  on all columns (k = derivative term already multiplied by <s,g>)
template <class TL, typename T>
__device__ __forceinline__ void axpby(TL& g, const TL& s, T f, T k) {
#pragma unroll
    for (int i = 0; i < TL::kRegs; ++i) g.v[i] = f * g.v[i] + k * s.v[i];
}

// ---------------------------------------------------------------------------------------
// standalone maps: which = 0 proj, 1 expmap0, 2 logmap0, 3 proj(expmap0(.))
template <typename T> struct PbMapBwd {
    Curv cv;
    int which;
    template <class TL> __device__ __forceinline__ void operator()(TL& x, TL& g, int64_t) const {
        using S = Dual<T, 1>;
        const T nr = hd_sqrt(x.sumsq());
        const S n = S::var(nr, 0);
        S f;
        if (which == 0) f = pb_proj_f<T>(n, cv);
        else if (which == 1) f = pb_expmap0_f<T>(n, cv);
        else if (which == 2) f = pb_logmap0_f<T>(n, cv);
        else { f = pb_expmap0_f<T>(n, cv); f = f * pb_proj_f<T>(f * n, cv); }
        const T t = x.dot(g);
        axpby(g, x, f.v, nr > T(0) ? f.d[0] * t / nr : T(0));
    }
};

// hyperboloid: spatial factor f(ns) and time coordinate tau(ns) of expmap0 (+proj)
template <typename T, class S>
__device__ __forceinline__ void hb_expmap0_f(const S& ns, const Curv& cv, S& f, S& tau) {
    const T sK = (T)cv.sK;
    const S n = d_clamp_min(ns, (T)kMinNorm);
    S sh, ch;
    d_sinhcosh15(n / sK, sh, ch);
    f = sh * sK / n;
    const S yn = f * ns;
    tau = d_sqrt(d_clamp_min(yn * yn + (T)cv.K, (T)Lim<T>::hb_eps));
}
// logmap0 factor from the time coordinate y0 and the raw spatial norm yn
template <typename T, class S> __device__ __forceinline__ S hb_logmap0_f(const S& y0, const S& yn, const Curv& cv) {
    const T sK = (T)cv.sK;
    const S n = d_clamp_min(yn, (T)kMinNorm);
    const S theta = d_clamp_min(y0 / sK, (T)(1.0 + Lim<T>::hb_eps));
    return d_arcosh(theta) * sK / n;
}

// which = 0 proj, 1 expmap0 (incl. proj), 2 logmap0, 3 input lift (= expmap0 of the spatial part)
template <typename T> struct HbMapBwd {
    Curv cv;
    int which;
    template <class TL> __device__ __forceinline__ void operator()(TL& x, TL& g, int64_t) const {
        const T nr = hd_sqrt(x.sumsq(1));
        const T t = x.dot(g, 1);             // <x_sp, g_sp>
        const T g0 = g.first();
        if (which == 2) {
            using S = Dual<T, 2>;
            const S f = hb_logmap0_f<T>(S::var(x.first(), 0), S::var(nr, 1), cv);
            axpby(g, x, f.v, nr > T(0) ? f.d[1] * t / nr : T(0));
            g.set_first(f.d[0] * t);
            return;
        }
        using S = Dual<T, 1>;
        const S n = S::var(nr, 0);
        S f, tau;
        if (which == 0) {
            f = S(T(1));
            tau = d_sqrt(d_clamp_min(n * n + (T)cv.K, (T)Lim<T>::hb_eps));
        } else {
            hb_expmap0_f<T>(n, cv, f, tau);
        }
        axpby(g, x, f.v, nr > T(0) ? (f.d[0] * t + tau.d[0] * g0) / nr : T(0));
        g.set_first(T(0));
    }
};

// ---- HypLinear tails --------------------------------------------------------------------
// forward: xt = Phi(|mx|, |x|) * mx  (matvec_epilogue [+proj] [+logmap0]);  returns d_mx in g and
// d|x| per row (poincare.py:93-102, hyp_layers.py:100-101, :134)
template <typename T> struct PbLinearTailBwd {
    Curv cv;
    const T* xnorm;
    T* dxnorm;
    int flags;
    template <class TL> __device__ __forceinline__ void operator()(TL& mx, TL& g, int64_t row) const {
        using S = Dual<T, 2>;
        const T sc = (T)cv.sc;
        const bool nz = mx.any_nonzero();
        const T mr = hd_sqrt(mx.sumsq());
        const T xr = xnorm[row];
        const S m = S::var(mr, 0);
        const S xn = xr >= (T)kMinNorm ? S::var(xr, 1) : S((T)kMinNorm);
        const S mc = d_clamp_min(m, (T)kMinNorm);
        S f = d_tanh15(mc / xn * d_artanh(xn * sc)) / (mc * sc);
        if (flags & 1) f = f * pb_proj_f<T>(f * m, cv);
        if (flags & 2) f = f * pb_logmap0_f<T>(f * m, cv);
        const T t = mx.dot(g);
        if (nz) {
            axpby(g, mx, f.v, mr > T(0) ? f.d[0] * t / mr : T(0));
            if (g.j == 0) dxnorm[row] = f.d[1] * t;
        } else {
            g.zero();
            if (g.j == 0) dxnorm[row] = T(0);
        }
    }
};

// forward: expmap0 (+proj) (+logmap0) of mu (hyperboloid.py:120-123 tail); column 0 of mu is unused
template <typename T> struct HbLinearTailBwd {
    Curv cv;
    int flags;
    template <class TL> __device__ __forceinline__ void operator()(TL& mu, TL& g, int64_t) const {
        using S = Dual<T, 1>;
        const T nr = hd_sqrt(mu.sumsq(1));
        const S n = S::var(nr, 0);
        S f, tau;
        hb_expmap0_f<T>(n, cv, f, tau);
        const T t = mu.dot(g, 1);
        const T g0 = g.first();
        T k;
        if (flags & 2) {
            f = f * hb_logmap0_f<T>(tau, f * n, cv);
            k = f.d[0] * t;                       // tangent output: time column is identically 0
        } else {
            k = f.d[0] * t + tau.d[0] * g0;
        }
        axpby(g, mu, f.v, nr > T(0) ? k / nr : T(0));
        g.set_first(T(0));
    }
};

// ---- HypAgg tails (+ fused HypAct) --------------------------------------------------------
// forward: out = [HypAct](proj(expmap0(s)));  s = spmm(adj, xt) saved by the forward pass
template <typename T> struct PbAggTailBwd {
    Curv cin, cout;
    int act, fuse;
    template <class TL> __device__ __forceinline__ void operator()(TL& s, TL& g, int64_t) const {
        using S = Dual<T, 1>;
        const T sr = hd_sqrt(s.sumsq());
        const S n = S::var(sr, 0);
        S f1 = pb_expmap0_f<T>(n, cin);
        f1 = f1 * pb_proj_f<T>(f1 * n, cin);
        if (fuse) {
            f1 = f1 * pb_logmap0_f<T>(f1 * n, cin);
            // a = act(u), u = f1 * s; second radial chain on |a|
            TL a = s;
            const T f1v = f1.v;
            const int ac = act;
            a.map([=](T x) { return apply_act(f1v * x, ac); });
            const T ar = hd_sqrt(a.sumsq());
            const S na = S::var(ar, 0);
            S f2 = pb_expmap0_f<T>(na, cout);
            f2 = f2 * pb_proj_f<T>(f2 * na, cout);
            const T t2 = a.dot(g);
            axpby(g, a, f2.v, ar > T(0) ? f2.d[0] * t2 / ar : T(0));      // g = dL/da
#pragma unroll
            for (int i = 0; i < TL::kRegs; ++i) g.v[i] *= act_grad(f1v * s.v[i], ac);   // dL/du
        }
        const T t1 = s.dot(g);
        axpby(g, s, f1.v, sr > T(0) ? f1.d[0] * t1 / sr : T(0));
    }
};

template <typename T> struct HbAggTailBwd {
    Curv cin, cout;
    int act, fuse;
    template <class TL> __device__ __forceinline__ void operator()(TL& s, TL& g, int64_t) const {
        using S = Dual<T, 1>;
        const T sr = hd_sqrt(s.sumsq(1));
        const S n = S::var(sr, 0);
        S f1, tau1;
        hb_expmap0_f<T>(n, cin, f1, tau1);
        if (!fuse) {
            const T t = s.dot(g, 1), g0 = g.first();
            axpby(g, s, f1.v, sr > T(0) ? (f1.d[0] * t + tau1.d[0] * g0) / sr : T(0));
            g.set_first(T(0));
            return;
        }
        f1 = f1 * hb_logmap0_f<T>(tau1, f1 * n, cin);
        TL a = s;
        const T f1v = f1.v;
        const int ac = act;
        a.map([=](T x) { return apply_act(f1v * x, ac); });
        a.set_first(T(0));                                   // proj_tan0 (hyperboloid.py:64-68)
        const T ar = hd_sqrt(a.sumsq(1));
        const S na = S::var(ar, 0);
        S f2, tau2;
        hb_expmap0_f<T>(na, cout, f2, tau2);
        const T t2 = a.dot(g, 1), g0 = g.first();
        axpby(g, a, f2.v, ar > T(0) ? (f2.d[0] * t2 + tau2.d[0] * g0) / ar : T(0));
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) g.v[i] *= act_grad(f1v * s.v[i], ac);
        g.set_first(T(0));
        const T t1 = s.dot(g, 1);
        axpby(g, s, f1.v, sr > T(0) ? f1.d[0] * t1 / sr : T(0));
        g.set_first(T(0));
    }
};

// forward tails on a stored aggregate (training forward = hd_spmm_csr + this, so that s is saved)
template <class M, typename T> struct AggTailFwd {
    Curv cin, cout;
    int act, fuse;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t) const {
        M::expmap0(r, cin);
        M::proj(r, cin);
        if (fuse) hyp_act<M, T>(r, cin, cout, act);
    }
};

// g <- g + coef[row] * x / |x|   (d|x| flowing back into the row; 0 for a zero row)
template <typename T> struct NormBwdAdd {
    const T* coef;
    template <class TL> __device__ __forceinline__ void operator()(TL& x, TL& g, int64_t row) const {
        const T nr = hd_sqrt(x.sumsq());
        const T k = nr >= (T)kMinNorm ? coef[row] / nr : T(0);
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) g.v[i] += k * x.v[i];
    }
};

// ---- LPModel.decode backward (models/base_models.py:152-159, layers/layers.py:85-87) --------
// per pair e: probs = 1/(exp((sq - r)/t) + 1), sq = sqdist(h[i], h[j]);  given d probs, writes
// d h[i] into ge[e] and d h[j] into ge[n_pairs + e]; the caller sums the rows per node with a
// selection-matrix SpMM (deterministic).  G lanes per pair, rows in registers.
template <typename T, int G, int VEC, bool HYPERBOLOID>
__global__ __launch_bounds__(kBlock) void edge_decode_bwd(const T* __restrict__ h, const int64_t* __restrict__ pairs,
                                                          int64_t n_rows, const T* __restrict__ gprob,
                                                          T* __restrict__ ge, int64_t n, int d, Curv cv, int fd, T fr,
                                                          T ft) {
    constexpr int RPB = kBlock / G;
    Tile<T, G, VEC> x, y;
    x.init(d);
    y.init(d);
    for (int64_t e = (int64_t)blockIdx.x * RPB + threadIdx.x / G; e < n; e += (int64_t)gridDim.x * RPB) {
        int64_t ia = pairs[2 * e], ib = pairs[2 * e + 1];
        ia = ia < 0 ? 0 : (ia >= n_rows ? n_rows - 1 : ia);
        ib = ib < 0 ? 0 : (ib >= n_rows ? n_rows - 1 : ib);
        x.load(h + ia * d);
        y.load(h + ib * d);
        T gs = gprob[e];                                   // -> d loss / d sqdist
        if constexpr (HYPERBOLOID) {
            const T K = (T)cv.K;
            const T x0 = x.first(), y0 = y.first();
            const T prod = x.dot(y) - T(2) * x0 * y0;
            const T raw = -prod / K, lo = (T)(1.0 + Lim<T>::hb_eps);
            const T theta = raw < lo ? lo : raw;
            const T a = arcosh_ref(theta);
            const T sq_raw = K * (a * a);
            const T sq = sq_raw > T(50) ? T(50) : sq_raw;
            if (fd) { const T p = T(1) / ((T)exp((double)((sq - fr) / ft)) + T(1)); gs *= -p * (T(1) - p) / ft; }
            // clamp(max=50) and clamp(min=1+eps) pass gradient only inside the closed range
            T gt = (sq_raw <= T(50) && raw >= lo) ? gs * T(2) * K * a / hd_sqrt(theta * theta - T(1)) : T(0);
            gt = -gt / K;                                 // d theta / d<x,y>_L
            // d<x,y>_L/dx = y with the time component negated
#pragma unroll
            for (int i = 0; i < Tile<T, G, VEC>::kRegs; ++i) {
                const T xv = x.v[i], yv = y.v[i];
                x.v[i] = gt * yv;
                y.v[i] = gt * xv;
            }
            if (x.j == 0) { x.v[0] = -x.v[0]; y.v[0] = -y.v[0]; }
        } else {
            const T c = (T)cv.c, sc = (T)cv.sc;
            // m = mobius_add(a, y), a = -x (poincare.py:85-91)
            const T x2 = x.sumsq(), y2 = y.sumsq(), xy = -x.dot(y);
            const T A = T(1) + T(2) * c * xy + c * y2, B = T(1) - c * x2;
            const T Draw = T(1) + T(2) * c * xy + c * c * x2 * y2;
            const bool dlive = Draw >= (T)kMinNorm;
            const T D = dlive ? Draw : (T)kMinNorm;
            Tile<T, G, VEC> m = x;
#pragma unroll
            for (int i = 0; i < Tile<T, G, VEC>::kRegs; ++i) m.v[i] = (A * (-x.v[i]) + B * y.v[i]) / D;
            const T mn = hd_sqrt(m.sumsq());
            const T z = sc * mn;
            const T dist = artanh_ref(z) * T(2) / sc;
            const T sq = dist * dist;
            if (fd) { const T p = T(1) / ((T)exp((double)((sq - fr) / ft)) + T(1)); gs *= -p * (T(1) - p) / ft; }
            const T zc = z > (T)(1.0 - 1e-15) ? (T)(1.0 - 1e-15) : z;
            // d sq / d|m| = 2 dist * 2 / (1 - z^2);  d|m|/dm = m/|m| (0 at m = 0, like torch.norm)
            const T gm_scale = mn > T(0) ? gs * T(2) * dist * T(2) / (T(1) - zc * zc) / mn : T(0);
            // gm = gm_scale * m ; s_a = <gm,a>, s_p = <gm,y>, s_m = <gm,m>
            const T ma = -m.dot(x), mp = m.dot(y), mm = mn * mn;
            const T s_a = gm_scale * ma, s_p = gm_scale * mp, s_m = dlive ? gm_scale * mm : T(0);
            const T c2 = T(2) * c;
#pragma unroll
            for (int i = 0; i < Tile<T, G, VEC>::kRegs; ++i) {
                const T av = -x.v[i], pv = y.v[i], gmv = gm_scale * m.v[i];
                const T ga = (A * gmv + c2 * s_a * pv - c2 * s_p * av - s_m * (c2 * pv + c2 * c * y2 * av)) / D;
                const T gp = (B * gmv + c2 * s_a * (av + pv) - s_m * (c2 * av + c2 * c * x2 * pv)) / D;
                x.v[i] = -ga;
                y.v[i] = gp;
            }
        }
        x.store(ge + e * d);
        y.store(ge + (n + e) * d);
    }
}

template <typename T, bool HYPERBOLOID>
int launch_edge_decode_bwd(const char* name, const void* h, const int64_t* pairs, int64_t n_rows, const void* gprob,
                           void* ge, int64_t n, int64_t d, hd_curv c, int fd, double r, double t, void* st) {
    HD_REQUIRE(n >= 0 && d >= 1 && n_rows >= 1, "%s: bad shape", name);
    if (n == 0) return HD_OK;
    HD_REQUIRE(h && pairs && gprob && ge, "%s: null pointer", name);
    const RowCfg cfg = pick_cfg<T>(d, is_aligned16(h) && is_aligned16(ge));
    HD_REQUIRE(cfg.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    return dispatch_cfg<T>(cfg, [&](auto g, auto v) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value;
        edge_decode_bwd<T, G, V, HYPERBOLOID><<<grid_for_rows(n, G), kBlock, 0, (hipStream_t)st>>>(
            (const T*)h, pairs, n_rows, (const T*)gprob, (T*)ge, n, (int)d, mk(c), fd, (T)r, (T)t);
        return check_launch(name);
    });
}


// ---- two-point maps of the Poincare ball called on their own (poincare.py:25-31, 55-71, 85-91) ------------------
// Every one of them is  out = alpha * a + beta * b  with alpha, beta functions of the three row sums
// (a.a, b.b, a.b): the chain is evaluated on Dual<T, 3> seeded on those sums, and with q_i = d alpha/d s_i <g, a> +
// d beta/d s_i <g, b>
//     d a = alpha g + 2 q_0 a + q_2 b,      d b = beta g + 2 q_1 b + q_2 a.
// _lambda_x is computed on .data in the reference (poincare.py:34): a constant here as well.
template <typename T, class S>
__device__ __forceinline__ void pb_madd_coef(const S& x2, const S& y2, const S& xy, const Curv& cv, S& A, S& B) {
    const T c = (T)cv.c;
    const S den = d_clamp_min(xy * (T(2) * c) + x2 * y2 * (c * c) + T(1), (T)kMinNorm);
    A = (xy * (T(2) * c) + y2 * c + T(1)) / den;
    B = (S(T(1)) - x2 * c) / den;
}
// which = 0 mobius_add(x = a, y = b), 1 expmap(u = b, p = a), 2 logmap(p1 = a, p2 = b)
template <typename T, class S>
__device__ __forceinline__ void pb_map2_coef(int which, const S& a2, const S& b2, const S& ab, const Curv& cv, S& alpha, S& beta) {
    const T c = (T)cv.c, sc = (T)cv.sc;
    if (which == 0) {
        pb_madd_coef<T>(a2, b2, ab, cv, alpha, beta);
    } else if (which == 1) {
        const T lam = T(2) / clamp_min(T(1) - c * a2.v, (T)kMinNorm);
        const S un = d_clamp_min(d_sqrt(b2), (T)kMinNorm);
        const S F = d_tanh15(un * (sc / T(2) * lam)) / (un * sc);          // second = F * u
        S A, B;
        pb_madd_coef<T>(a2, F * F * b2, F * ab, cv, A, B);
        alpha = A;
        beta = B * F;
    } else {
        const T lam = T(2) / clamp_min(T(1) - c * a2.v, (T)kMinNorm);
        S A, B;
        pb_madd_coef<T>(a2, b2, -ab, cv, A, B);                            // sub = (-p1) (+) p2 = -A a + B b
        const S sn2 = A * A * a2 - A * B * ab * T(2) + B * B * b2;
        const S sn = d_clamp_min(d_sqrt(sn2), (T)kMinNorm);
        const S C = d_artanh(sn * sc) * (T(2) / sc / lam) / sn;
        alpha = -(C * A);
        beta = C * B;
    }
}

template <typename T> struct PbMap2Bwd {
    Curv cv;
    int which;
    const T* b;
    T* db;
    // a = saved first operand, g = incoming gradient (left holding d a); the second operand is loaded here, d b stored here
    template <class TL> __device__ __forceinline__ void operator()(TL& a, TL& g, int64_t row) const {
        using S = Dual<T, 3>;
        TL bt;
        bt.init(a.d);
        bt.load(b + row * a.d);
        const T a2 = a.sumsq(), b2 = bt.sumsq(), ab = a.dot(bt), ga = g.dot(a), gb = g.dot(bt);
        S alpha, beta;
        // d_sqrt of an exactly zero sum has an infinite slope that the clamp behind it discards; keep it finite
        pb_map2_coef<T>(which, S::var(a2, 0), S::var(b2 > T(0) ? b2 : T(0), 1), S::var(ab, 2), cv, alpha, beta);
        T q[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T da = alpha.d[i], dbt = beta.d[i];
            q[i] = (da == da && dbt == dbt) ? da * ga + dbt * gb : T(0);
        }
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) {
            const T gi = g.v[i], ai = a.v[i], bi = bt.v[i];
            g.v[i] = alpha.v * gi + T(2) * q[0] * ai + q[2] * bi;
            bt.v[i] = beta.v * gi + T(2) * q[1] * bi + q[2] * ai;
        }
        bt.store(db + row * a.d);
    }
};

// sqdist (ball, :25-31): out = (2 / sc * artanh(sc |(-p1) (+) p2|))^2, one scalar per row; g = its incoming gradient
template <typename T> struct PbSqdistBwd {
    Curv cv;
    const T* g;
    T* da;
    template <class TL> __device__ __forceinline__ void operator()(TL& a, TL& bt, int64_t row) const {
        using S = Dual<T, 3>;
        const T sc = (T)cv.sc;
        const T a2 = a.sumsq(), b2 = bt.sumsq(), ab = a.dot(bt);
        S A, B;
        pb_madd_coef<T>(S::var(a2, 0), S::var(b2, 1), -S::var(ab, 2), cv, A, B);
        const S sn2 = A * A * S::var(a2, 0) - A * B * S::var(ab, 2) * T(2) + B * B * S::var(b2, 1);
        T f[3] = {T(0), T(0), T(0)};
        if (sn2.v > T(0)) {                                  // torch.norm's gradient at the zero vector is zero
            const S dist = d_artanh(d_sqrt(sn2) * sc) * (T(2) / sc);
            const S out = dist * dist;
#pragma unroll
            for (int i = 0; i < 3; ++i) f[i] = out.d[i];
        }
        const T gs = g[row];
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) {
            const T ai = a.v[i], bi = bt.v[i];
            a.v[i] = gs * (T(2) * f[0] * ai + f[2] * bi);
            bt.v[i] = gs * (T(2) * f[1] * bi + f[2] * ai);
        }
        a.store(da + row * a.d);
    }
};
// sqdist (hyperboloid, :35-41): min(K arcosh(max(-<x, y>_L / K, 1 + eps))^2, 50)
template <typename T> struct HbSqdistBwd {
    Curv cv;
    const T* g;
    T* da;
    template <class TL> __device__ __forceinline__ void operator()(TL& x, TL& y, int64_t row) const {
        using S = Dual<T, 1>;
        const T K = (T)cv.K;
        const T x0 = x.first(), y0 = y.first();
        const S prod = S::var(x.dot(y) - T(2) * x0 * y0, 0);
        const S theta = d_clamp_min(-prod / K, (T)(1.0 + Lim<T>::hb_eps));
        const S ac = d_arcosh(theta);
        const S out = ac * ac * K;
        const T dprod = (out.v <= T(50) ? out.d[0] : T(0)) * g[row];   // torch.clamp(max=50) passes the gradient up to 50
        // d prod / d x = y with the time coordinate negated, and vice versa
#pragma unroll
        for (int i = 0; i < TL::kRegs; ++i) {
            const T xi = x.v[i], yi = y.v[i];
            x.v[i] = dprod * yi;
            y.v[i] = dprod * xi;
        }
        x.set_first(-dprod * y0);
        y.set_first(-dprod * x0);
        x.store(da + row * x.d);
    }
};

template <typename T, class Op>
int bwd2(const char* name, const void* saved, const void* grad, void* out, int64_t n, int64_t d, Op op, void* st) {
    return launch_map2<T, int32_t>(name, (const T*)saved, nullptr, (const T*)grad, (T*)out, n, d, op, (hipStream_t)st);
}

}  // namespace

#define HD_BWD(T, sfx)                                                                                        \
    extern "C" int hd_pb_map_bwd_##sfx(int which, const void* x, const void* dy, void* dx, int64_t n,        \
                                       int64_t d, hd_curv c, void* st) {                                      \
        HD_REQUIRE(which >= 0 && which <= 3, "hd_pb_map_bwd: bad map code");                                  \
        return bwd2<T>("hd_pb_map_bwd_" #sfx, x, dy, dx, n, d, PbMapBwd<T>{mk(c), which}, st);               \
    }                                                                                                         \
    extern "C" int hd_hb_map_bwd_##sfx(int which, const void* x, const void* dy, void* dx, int64_t n,        \
                                       int64_t d, hd_curv c, void* st) {                                      \
        HD_REQUIRE(which >= 0 && which <= 3, "hd_hb_map_bwd: bad map code");                                  \
        return bwd2<T>("hd_hb_map_bwd_" #sfx, x, dy, dx, n, d, HbMapBwd<T>{mk(c), which}, st);               \
    }                                                                                                         \
    extern "C" int hd_pb_linear_tail_bwd_##sfx(const void* mx, const void* xnorm, const void* dxt, void* dmx, \
                                               void* dxnorm, int64_t n, int64_t d, hd_curv c, int flags,      \
                                               void* st) {                                                    \
        HD_REQUIRE((xnorm && dxnorm) || n == 0, "hd_pb_linear_tail_bwd: xnorm / dxnorm missing");             \
        return bwd2<T>("hd_pb_linear_tail_bwd_" #sfx, mx, dxt, dmx, n, d,                                     \
                       PbLinearTailBwd<T>{mk(c), (const T*)xnorm, (T*)dxnorm, flags}, st);                    \
    }                                                                                                         \
    extern "C" int hd_hb_linear_tail_bwd_##sfx(const void* mu, const void* dxt, void* dmu, int64_t n,        \
                                               int64_t d, hd_curv c, int flags, void* st) {                   \
        return bwd2<T>("hd_hb_linear_tail_bwd_" #sfx, mu, dxt, dmu, n, d, HbLinearTailBwd<T>{mk(c), flags}, st); \
    }                                                                                                         \
    extern "C" int hd_pb_agg_tail_bwd_##sfx(const void* s, const void* dout, void* ds, int64_t n, int64_t d, \
                                            hd_curv ci, hd_curv co, int act, int fuse, void* st) {            \
        return bwd2<T>("hd_pb_agg_tail_bwd_" #sfx, s, dout, ds, n, d, PbAggTailBwd<T>{mk(ci), mk(co), act, fuse}, st); \
    }                                                                                                         \
    extern "C" int hd_hb_agg_tail_bwd_##sfx(const void* s, const void* dout, void* ds, int64_t n, int64_t d, \
                                            hd_curv ci, hd_curv co, int act, int fuse, void* st) {            \
        return bwd2<T>("hd_hb_agg_tail_bwd_" #sfx, s, dout, ds, n, d, HbAggTailBwd<T>{mk(ci), mk(co), act, fuse}, st); \
    }                                                                                                         \
    extern "C" int hd_pb_agg_tail_##sfx(const void* s, void* out, int64_t n, int64_t d, hd_curv ci,          \
                                        hd_curv co, int act, int fuse, void* st) {                            \
        return launch_map1<T>("hd_pb_agg_tail_" #sfx, (const T*)s, (T*)out, n, d,                            \
                              AggTailFwd<PB<T>, T>{mk(ci), mk(co), act, fuse}, (hipStream_t)st);              \
    }                                                                                                         \
    extern "C" int hd_hb_agg_tail_##sfx(const void* s, void* out, int64_t n, int64_t d, hd_curv ci,          \
                                        hd_curv co, int act, int fuse, void* st) {                            \
        return launch_map1<T>("hd_hb_agg_tail_" #sfx, (const T*)s, (T*)out, n, d,                            \
                              AggTailFwd<HB<T>, T>{mk(ci), mk(co), act, fuse}, (hipStream_t)st);              \
    }                                                                                                         \
    extern "C" int hd_pb_edge_decode_bwd_##sfx(const void* h, const int64_t* pairs, int64_t n_rows,          \
                                               const void* gprob, void* ge, int64_t n, int64_t d, hd_curv c,  \
                                               int fermi_dirac, double r, double t, void* st) {               \
        return launch_edge_decode_bwd<T, false>("hd_pb_edge_decode_bwd_" #sfx, h, pairs, n_rows, gprob, ge, n, d, c, \
                                                fermi_dirac, r, t, st);                                       \
    }                                                                                                         \
    extern "C" int hd_hb_edge_decode_bwd_##sfx(const void* h, const int64_t* pairs, int64_t n_rows,          \
                                               const void* gprob, void* ge, int64_t n, int64_t d, hd_curv c,  \
                                               int fermi_dirac, double r, double t, void* st) {               \
        return launch_edge_decode_bwd<T, true>("hd_hb_edge_decode_bwd_" #sfx, h, pairs, n_rows, gprob, ge, n, d, c, \
                                               fermi_dirac, r, t, st);                                        \
    }                                                                                                         \
    extern "C" int hd_pb_map2_bwd_##sfx(int which, const void* a, const void* b, const void* g, void* da,     \
                                        void* db, int64_t n, int64_t d, hd_curv c, void* st) {                \
        HD_REQUIRE(which >= 0 && which <= 2, "hd_pb_map2_bwd: map code must be 0 (mobius_add), 1 (expmap) or 2 (logmap)"); \
        HD_REQUIRE((b && db && db != da) || n == 0, "hd_pb_map2_bwd: second operand / its gradient missing");  \
        return bwd2<T>("hd_pb_map2_bwd_" #sfx, a, g, da, n, d, PbMap2Bwd<T>{mk(c), which, (const T*)b, (T*)db}, st); \
    }                                                                                                         \
    extern "C" int hd_pb_sqdist_bwd_##sfx(const void* a, const void* b, const void* g, void* da, void* db,    \
                                          int64_t n, int64_t d, hd_curv c, void* st) {                        \
        HD_REQUIRE((g && da && da != db) || n == 0, "hd_pb_sqdist_bwd: gradient pointers missing");            \
        return bwd2<T>("hd_pb_sqdist_bwd_" #sfx, a, b, db, n, d, PbSqdistBwd<T>{mk(c), (const T*)g, (T*)da}, st); \
    }                                                                                                         \
    extern "C" int hd_hb_sqdist_bwd_##sfx(const void* a, const void* b, const void* g, void* da, void* db,    \
                                          int64_t n, int64_t d, hd_curv c, void* st) {                        \
        HD_REQUIRE((g && da && da != db) || n == 0, "hd_hb_sqdist_bwd: gradient pointers missing");            \
        return bwd2<T>("hd_hb_sqdist_bwd_" #sfx, a, b, db, n, d, HbSqdistBwd<T>{mk(c), (const T*)g, (T*)da}, st); \
    }                                                                                                         \
    extern "C" int hd_rownorm_bwd_add_##sfx(const void* x, const void* coef, const void* g, void* out,       \
                                            int64_t n, int64_t d, void* st) {                                 \
        HD_REQUIRE(coef || n == 0, "hd_rownorm_bwd_add: coef missing");                                       \
        return bwd2<T>("hd_rownorm_bwd_add_" #sfx, x, g, out, n, d, NormBwdAdd<T>{(const T*)coef}, st);       \
    }
HD_BWD(float, f32)
HD_BWD(double, f64)
