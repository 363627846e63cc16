// Forward-mode dual numbers for the per-row SCALAR factors of the manifold maps.
//
// Every fused row kernel of the encoder is "row * factor(norms)": the factor is a chain of
// tanh / artanh / arcosh / clamps / branches on one or two per-row scalars.  The backward
// kernels evaluate the same chain on Dual<T,NV> values seeded on those scalars and read the
// partial derivatives off the result, so each clamp and branch differentiates exactly the way
// torch autograd treats the reference's expression (clamp: gradient inside the closed range
// only; torch.where: the taken branch; Artanh / Arcosh: their hand-written backward,
// utils/math_utils.py:38-41,65-68).
#pragma once
#include "hd_manifold.cuh"

namespace hd {

template <typename T, int NV>
struct Dual {
    T v;
    T d[NV];
    __device__ __forceinline__ Dual() {}
    __device__ __forceinline__ Dual(T x) : v(x) {
#pragma unroll
        for (int i = 0; i < NV; ++i) d[i] = T(0);
    }
    static __device__ __forceinline__ Dual var(T x, int which) {
        Dual r(x);
        r.d[which] = T(1);
        return r;
    }
};

#define HD_DUAL_T template <typename T, int NV> __device__ __forceinline__ Dual<T, NV>

HD_DUAL_T operator+(const Dual<T, NV>& a, const Dual<T, NV>& b) {
    Dual<T, NV> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
HD_DUAL_T operator-(const Dual<T, NV>& a, const Dual<T, NV>& b) {
    Dual<T, NV> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
HD_DUAL_T operator-(const Dual<T, NV>& a) {
    Dual<T, NV> r;
    r.v = -a.v;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = -a.d[i];
    return r;
}
HD_DUAL_T operator*(const Dual<T, NV>& a, const Dual<T, NV>& b) {
    Dual<T, NV> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
HD_DUAL_T operator/(const Dual<T, NV>& a, const Dual<T, NV>& b) {
    Dual<T, NV> r;
    const T inv = T(1) / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
HD_DUAL_T operator*(const Dual<T, NV>& a, T s) {
    Dual<T, NV> r;
    r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] * s;
    return r;
}
HD_DUAL_T operator*(T s, const Dual<T, NV>& a) { return a * s; }
HD_DUAL_T operator/(const Dual<T, NV>& a, T s) { return a * (T(1) / s); }
HD_DUAL_T operator+(const Dual<T, NV>& a, T s) {
    Dual<T, NV> r = a;
    r.v += s;
    return r;
}
HD_DUAL_T d_chain(const Dual<T, NV>& a, T value, T deriv) {     // f(a) with known f(a.v), f'(a.v)
    Dual<T, NV> r;
    r.v = value;
#pragma unroll
    for (int i = 0; i < NV; ++i) r.d[i] = a.d[i] * deriv;
    return r;
}
HD_DUAL_T d_sqrt(const Dual<T, NV>& a) {
    const T s = hd_sqrt(a.v);
    return d_chain(a, s, T(0.5) / s);
}
// torch.clamp(min=lo): gradient passes where x >= lo
HD_DUAL_T d_clamp_min(const Dual<T, NV>& a, T lo) { return a.v >= lo ? a : Dual<T, NV>(lo); }
// utils/math_utils.py:6-15: x.clamp(-15, 15) then the function
HD_DUAL_T d_tanh15(const Dual<T, NV>& a) {
    const bool in = a.v >= T(-15) && a.v <= T(15);
    const T t = tanh15(a.v);
    return d_chain(a, t, in ? T(1) - t * t : T(0));
}
template <typename T, int NV>
__device__ __forceinline__ void d_sinhcosh15(const Dual<T, NV>& a, Dual<T, NV>& sh, Dual<T, NV>& ch) {
    const bool in = a.v >= T(-15) && a.v <= T(15);
    T s, c;
    sinhcosh15(a.v, s, c);
    sh = d_chain(a, s, in ? c : T(0));
    ch = d_chain(a, c, in ? s : T(0));
}
// Artanh: forward artanh_ref, backward g / (1 - x^2) on the clamped input (math_utils.py:38-41)
HD_DUAL_T d_artanh(const Dual<T, NV>& a) {
    const T lo = (T)(-1.0 + 1e-15), hi = (T)(1.0 - 1e-15);
    const T xc = a.v < lo ? lo : (a.v > hi ? hi : a.v);
    return d_chain(a, artanh_ref(a.v), T(1) / (T(1) - xc * xc));
}
// Arcosh: backward g / sqrt(x^2 - 1) on the clamped input (math_utils.py:65-68)
HD_DUAL_T d_arcosh(const Dual<T, NV>& a) {
    const T lo = (T)(1.0 + 1e-15);
    const T xc = a.v < lo ? lo : a.v;
    return d_chain(a, arcosh_ref(a.v), T(1) / hd_sqrt(xc * xc - T(1)));
}

// derivative of the tangent-space activations (layers/hyp_layers.py:171), given input x
template <typename T> __device__ __forceinline__ T act_grad(T x, int act) {
    switch (act) {
        case ACT_RELU: return x > T(0) ? T(1) : T(0);
        case ACT_TANH: { const T t = hd_tanh(x); return T(1) - t * t; }
        case ACT_SIGMOID: { const T s = T(1) / (T(1) + (T)exp((double)-x)); return s * (T(1) - s); }
        case ACT_ELU: return x > T(0) ? T(1) : (T)exp((double)x);
        case ACT_LEAKY_RELU: return x > T(0) ? T(1) : T(0.01);
        default: return T(1);
    }
}

}  // namespace hd
