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
// d/dx of the exact-erf GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

}  // namespace hd
