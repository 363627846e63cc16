// Exact-erf GELU (F.gelu default, models/direction_diffusion.py:56-69, ddpm.py:17-27) and its
// derivative, shared by the GEMM write-outs and the elementwise training kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace hd {

// erf to < 1 ulp (max relative error 8e-8 against fp64 erf over [-6, 6]) in ~20 VALU ops:
// two minimax polynomials (|x| <= 0.9277: odd polynomial; beyond: 1 - exp(poly)), both
// evaluated and selected, no branches.  The library erff costs ~5x more, and 64 of them per
// lane made the write-out of a 128x128 tile ~15 % of a wavefront's lifetime.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - __expf(r);
    r = copysignf(r, a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
// Two at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of
// FMAs per issue): the same operations in the same order as erf_fast / gelu_erf, so the results are bit
// identical; only exp and the selects stay scalar.  The GEMM write-outs are VALU-bound on this.
typedef float hd_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hd_f32x2 gelu_erf2(hd_f32x2 x) {
    const hd_f32x2 a = x * 0.70710678118654752440f;
    const hd_f32x2 t = {fabsf(a.x), fabsf(a.y)}, s = a * a;
    auto fma2 = [](hd_f32x2 p, hd_f32x2 q, hd_f32x2 c) { return __builtin_elementwise_fma(p, q, c); };
    auto k = [](float c) { return hd_f32x2{c, c}; };
    hd_f32x2 r = fma2(k(-1.72853470e-5f), t, k(3.83197126e-4f));
    const hd_f32x2 u = fma2(k(-3.88396438e-3f), t, k(2.42546219e-2f));
    r = fma2(r, s, u);
    r = fma2(r, t, k(-1.06777877e-1f));
    r = fma2(r, t, k(-6.34846687e-1f));
    r = fma2(r, t, k(-1.28717512e-1f));
    r = fma2(r, t, -t);
    r = hd_f32x2{1.0f - __expf(r.x), 1.0f - __expf(r.y)};
    r = hd_f32x2{copysignf(r.x, a.x), copysignf(r.y, a.y)};
    hd_f32x2 q = k(-5.96761703e-4f);
    q = fma2(q, s, k(4.99119423e-3f));
    q = fma2(q, s, k(-2.67681349e-2f));
    q = fma2(q, s, k(1.12819925e-1f));
    q = fma2(q, s, k(-3.76125336e-1f));
    q = fma2(q, s, k(1.28379166e-1f));
    q = fma2(q, a, a);
    const hd_f32x2 e = {t.x > 0.927734375f ? r.x : q.x, t.y > 0.927734375f ? r.y : q.y};
    return (x * 0.5f) * (e + 1.0f);
}
// The forward GELU of the GEMM write-outs, in about half the instructions of gelu_erf: GELU only needs
// 1 + erf(a) = erfc(-a), i.e. erfc(|a|) for x < 0 and 2 - erfc(|a|) for x >= 0, and erfc(t) = 2^(t P(t)) with one
// degree-7 P on all of t >= 0 (weighted minimax fit of log2 erfc on [0, 4.3], |erfc error| < 2e-8 before
// rounding; the leading coefficient is negative, so the tail runs to 2^-inf = 0).  No second polynomial and no
// cancellation: against fp64, max relative error 3.3e-6 on |x| < 3 (F.gelu in fp32: 2.2e-4, it cancels in
// 1 + erf), 3.4e-7 for x > -1, absolute error < 1e-7 |x| everywhere.
__device__ __forceinline__ float gelu_erfc(float x) {
    const float t = fabsf(x * 0.70710678118654752440f);
    float r = -4.535861479e-05f;
    r = fmaf(r, t, 4.455077501e-04f);
    r = fmaf(r, t, -1.489441880e-03f);
    r = fmaf(r, t, -7.746293751e-04f);
    r = fmaf(r, t, 2.825368093e-02f);
    r = fmaf(r, t, -1.484816150e-01f);
    r = fmaf(r, t, -9.184163933e-01f);
    r = fmaf(r, t, -1.627908593e+00f);
    const float e = __builtin_amdgcn_exp2f(r * t);                       // erfc(|a|)
    return x * (x >= 0.f ? fmaf(-0.5f, e, 1.0f) : 0.5f * e);
}
// gelu_erfc on two values at a time (v_pk_mul_f32 / v_pk_fma_f32: the same operations per element in the same order, so
// the same bits; exp2 and the select stay per element)
__device__ __forceinline__ hd_f32x2 gelu_erfc2(hd_f32x2 x) {
    const hd_f32x2 a = x * 0.70710678118654752440f;
    const hd_f32x2 t = {fabsf(a.x), fabsf(a.y)};
    auto fma2 = [](hd_f32x2 p, hd_f32x2 q, float c) { return __builtin_elementwise_fma(p, q, hd_f32x2{c, c}); };
    hd_f32x2 r = {-4.535861479e-05f, -4.535861479e-05f};
    r = fma2(r, t, 4.455077501e-04f);
    r = fma2(r, t, -1.489441880e-03f);
    r = fma2(r, t, -7.746293751e-04f);
    r = fma2(r, t, 2.825368093e-02f);
    r = fma2(r, t, -1.484816150e-01f);
    r = fma2(r, t, -9.184163933e-01f);
    r = fma2(r, t, -1.627908593e+00f);
    const hd_f32x2 rt = r * t;
    const float e0 = __builtin_amdgcn_exp2f(rt.x), e1 = __builtin_amdgcn_exp2f(rt.y);
    return hd_f32x2{x.x * (x.x >= 0.f ? fmaf(-0.5f, e0, 1.0f) : 0.5f * e0), x.y * (x.y >= 0.f ? fmaf(-0.5f, e1, 1.0f) : 0.5f * e1)};
}
// d/dx of the exact-erf GELU: Phi(x) + x * phi(x).  Phi from the same single-polynomial erfc as gelu_erfc (Phi(x) =
// erfc(-x / sqrt 2) / 2: no 1 + erf cancellation on the negative side), phi as one exp2: ~18 VALU ops instead of ~40 --
// the backward write-outs (epilogue 3) are VALU-bound on it.  Max abs error against fp64 over [-9, 9]: 1.3e-7 (the
// 0.5 (1 + erf) form with a correctly rounded erf: 1.3e-7).
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float t = fabsf(x * 0.70710678118654752440f);
    float r = -4.535861479e-05f;
    r = fmaf(r, t, 4.455077501e-04f);
    r = fmaf(r, t, -1.489441880e-03f);
    r = fmaf(r, t, -7.746293751e-04f);
    r = fmaf(r, t, 2.825368093e-02f);
    r = fmaf(r, t, -1.484816150e-01f);
    r = fmaf(r, t, -9.184163933e-01f);
    r = fmaf(r, t, -1.627908593e+00f);
    const float e = __builtin_amdgcn_exp2f(r * t);                       // erfc(|x| / sqrt 2)
    const float Phi = x >= 0.f ? fmaf(-0.5f, e, 1.0f) : 0.5f * e;
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return fmaf(x, pdf, Phi);
}

}  // namespace hd
