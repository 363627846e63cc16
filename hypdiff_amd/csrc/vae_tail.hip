// Graph-VAE hyperbolic tail: the row maps of HNNLayer = HypLinear -> HypNorm ->
// HypAct of the reference's models/hyp_layers.py:45-61,172-206,280-319 on the Poincare ball (c = 1 there),
// fused into ONE pass over the rows of mx = x @ W^T (the Euclidean product comes from the GEMM kernels).
//   flags bit0  HypLinear tail :198-205  h = expmap0(mx);  b = ptransp0(h, bias) = 2 bias / lambda_h;
//                                         h = expmap(u = h, p = b)   (the reference's argument order: the point
//                                         sits in the tangent slot, the transported bias in the base-point slot)
//         bit1  HypNorm :313-319          h = expmap0(LayerNorm(logmap0(h)))
//         bit2  HypAct  :292-296          h = expmap0(act(logmap0(h)))        (no proj, unlike layers/hyp_layers.py)
// HBM-bound: 8 d bytes per row (fp32), the three parameter vectors stay in L2.
#include "hd_rowkernels.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

template <typename T> struct OpVaeTail {
    Curv cv;
    const T* bias;
    const T* lnw;
    const T* lnb;
    T eps;
    int flags, act;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t) const {
        using P = PB<T>;
        if (flags & 1) {
            P::expmap0(r, cv);
            TL b;
            b.init(r.d);
            b.load(bias);
            const T lam = clamp_min(P::lambda_x(r, cv), (T)kMinNorm);     // manifolds/poincare.py:136-138
            b.scale(T(2) / lam);
            P::expmap(r, b, cv);                                           // r <- expmap(u = r, p = b)
        }
        if (flags & 2) {
            P::logmap0(r, cv);
            const T inv_d = T(1) / (T)r.d;
            const T mean = r.sum() * inv_d;
            r.map_cols([=](T x, int) { return x - mean; });
            const T rstd = T(1) / hd_sqrt(r.sumsq() * inv_d + eps);        // biased variance, as nn.LayerNorm
            const T* w = lnw;
            const T* bb = lnb;
            r.map_cols([=](T x, int c) { return x * rstd * w[c] + bb[c]; });
            P::expmap0(r, cv);
        }
        if (flags & 4) {
            P::logmap0(r, cv);
            const int a = act;
            r.map([=](T x) { return apply_act(x, a); });
            P::expmap0(r, cv);
        }
    }
};

template <typename T>
int run(const char* name, const void* mx, const void* bias, const void* lnw, const void* lnb, void* out, int64_t n,
        int64_t d, hd_curv c, double eps, int flags, int act, void* st) {
    HD_REQUIRE(flags >= 1 && flags <= 7, "%s: flags must select at least one of linear tail (1), norm (2), act (4)", name);
    HD_REQUIRE(!(flags & 1) || (bias && is_aligned16(bias)), "%s: bias missing or not 16-byte aligned", name);
    HD_REQUIRE(!(flags & 2) || (lnw && lnb), "%s: LayerNorm weight / bias missing", name);
    HD_REQUIRE(act >= 0 && act <= 5, "%s: unknown activation code %d", name, act);
    return launch_map1<T>(name, (const T*)mx, (T*)out, n, d,
                          OpVaeTail<T>{Curv{c.c, c.sqrt_c, c.K, c.sqrt_K}, (const T*)bias, (const T*)lnw, (const T*)lnb,
                                       (T)eps, flags, act},
                          (hipStream_t)st);
}

}  // namespace

extern "C" int hd_pb_vae_hnn_tail_f32(const void* mx, const void* bias, const void* ln_w, const void* ln_b, void* out,
                                      int64_t rows, int64_t dim, hd_curv c, double eps, int flags, int act, void* st) {
    return run<float>("hd_pb_vae_hnn_tail_f32", mx, bias, ln_w, ln_b, out, rows, dim, c, eps, flags, act, st);
}
extern "C" int hd_pb_vae_hnn_tail_f64(const void* mx, const void* bias, const void* ln_w, const void* ln_b, void* out,
                                      int64_t rows, int64_t dim, hd_curv c, double eps, int flags, int act, void* st) {
    return run<double>("hd_pb_vae_hnn_tail_f64", mx, bias, ln_w, ln_b, out, rows, dim, c, eps, flags, act, st);
}
