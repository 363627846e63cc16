// C-ABI entry points: fused per-row manifold maps (Poincare ball and hyperboloid).
// Memory-bound kernels: every row is read once and written once; all norms, clamps,
// artanh/arcosh and rescales of a composite map happen in registers in between.
#include "hd_rowkernels.cuh"
#include "hd_limbs.cuh"
#include <stdlib.h>
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

inline Curv mk(const hd_curv& c) { return Curv{c.c, c.sqrt_c, c.K, c.sqrt_K}; }

enum Map1 { M_PROJ, M_EXPMAP0, M_LOGMAP0, M_EXPMAP0_PROJ, M_PROJ_TAN0 };

template <class M, typename T, int WHICH> struct Op1 {
    Curv cv;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t) const {
        if constexpr (WHICH == M_PROJ) M::proj(r, cv);
        else if constexpr (WHICH == M_EXPMAP0) M::expmap0(r, cv);
        else if constexpr (WHICH == M_LOGMAP0) M::logmap0(r, cv);
        else if constexpr (WHICH == M_PROJ_TAN0) M::proj_tan0(r, cv);
        else { M::proj_tan0(r, cv); M::expmap0(r, cv); M::proj(r, cv); }
    }
};

template <class M, typename T> struct OpAct {
    Curv cin, cout;
    int act;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t) const {
        hyp_act<M, T>(r, cin, cout, act);
    }
};

enum Map2 { B_MOBIUS_ADD, B_EXPMAP, B_LOGMAP };
// y <- op(x, y) with the reference's argument roles:
//   mobius_add(x, y); expmap(u=y, p=x); logmap(p1=x, p2=y)
template <typename T, int WHICH> struct Op2 {
    Curv cv;
    template <class TL> __device__ __forceinline__ void operator()(TL& x, TL& y, int64_t) const {
        if constexpr (WHICH == B_MOBIUS_ADD) { PB<T>::mobius_add(x, y, cv); y = x; }
        else if constexpr (WHICH == B_EXPMAP) PB<T>::expmap(y, x, cv);
        else PB<T>::logmap(x, y, cv);
    }
};

template <class M, typename T> struct OpSqdist {
    Curv cv;
    template <class TL> __device__ __forceinline__ T operator()(const TL& x, const TL& y) const {
        return M::sqdist(x, y, cv);
    }
};

// mobius_matvec tail (+ optional proj, + optional logmap0 for the fused HGCN layer)
template <typename T> struct OpMatvecEpi {
    Curv cv;
    const T* xnorm;
    int flags;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t row) const {
        const T xn = clamp_min(xnorm[row], (T)kMinNorm);
        PB<T>::matvec_epilogue(r, xn, cv);
        if (flags & 1) PB<T>::proj(r, cv);
        if (flags & 2) PB<T>::logmap0(r, cv);
    }
};

// hyperboloid mobius_matvec tail: expmap0 (+proj of HypLinear) (+logmap0)
template <typename T> struct OpHbLinearEpi {
    Curv cv;
    int flags;
    template <class TL> __device__ __forceinline__ void operator()(TL& r, int64_t) const {
        HB<T>::expmap0(r, cv);
        if (flags & 1) HB<T>::proj(r, cv);
        if (flags & 2) HB<T>::logmap0(r, cv);
    }
};

template <class M, typename T, int WHICH>
int map1(const char* name, const void* in, void* out, int64_t n, int64_t d, hd_curv c, void* st, int sop) {
    const RowCfg cfg = pick_cfg<T>(d, is_aligned16(in) && is_aligned16(out));
    if (!cfg.ok && sop >= 0 && n > 0 && d >= 1)
        return launch_stream<T>(name, (const T*)in, (T*)out, n, d, sop, mk(c), (hipStream_t)st);
    return launch_map1<T>(name, (const T*)in, (T*)out, n, d, Op1<M, T, WHICH>{mk(c)}, (hipStream_t)st);
}

}  // namespace

#define HD_MAP1(fn, M, WHICH, SOP)                                                                   \
    extern "C" int fn##_f32(const void* in, void* out, int64_t n, int64_t d, hd_curv c, void* st) { \
        return map1<M<float>, float, WHICH>(#fn "_f32", in, out, n, d, c, st, SOP);                 \
    }                                                                                                \
    extern "C" int fn##_f64(const void* in, void* out, int64_t n, int64_t d, hd_curv c, void* st) { \
        return map1<M<double>, double, WHICH>(#fn "_f64", in, out, n, d, c, st, SOP);               \
    }

HD_MAP1(hd_pb_proj, PB, M_PROJ, S_PB_PROJ)
HD_MAP1(hd_pb_expmap0, PB, M_EXPMAP0, S_PB_EXPMAP0)
HD_MAP1(hd_pb_logmap0, PB, M_LOGMAP0, S_PB_LOGMAP0)
HD_MAP1(hd_pb_expmap0_proj, PB, M_EXPMAP0_PROJ, S_PB_EXPMAP0_PROJ)
HD_MAP1(hd_hb_proj, HB, M_PROJ, S_HB_PROJ)
HD_MAP1(hd_hb_proj_tan0, HB, M_PROJ_TAN0, -1)
HD_MAP1(hd_hb_expmap0, HB, M_EXPMAP0, S_HB_EXPMAP0)
HD_MAP1(hd_hb_logmap0, HB, M_LOGMAP0, S_HB_LOGMAP0)
HD_MAP1(hd_hb_expmap0_proj, HB, M_EXPMAP0_PROJ, S_HB_EXPMAP0)

__device__ __forceinline__ float lane_get(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// HypAct for fp32 rows of 256..512 columns and a positively homogeneous activation: four rows per wavefront, the
// scalar chain (artanh / tanh or arcosh / sinh / cosh) evaluated once per wavefront, row r in lane r (the generic
// row kernel evaluates it wavefront-uniformly, i.e. 64 times over: 0.46-0.54 of the HBM peak in round 1)
template <class M>
__global__ __launch_bounds__(256) void hyp_act_rows4(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                     int d, Curv cin, Curv cout, float slope) {
    constexpr int R = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using TL = Tile<float, 64, 4, 8>;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * R; base < n; base += (int64_t)gridDim.x * 4 * R) {
        TL r[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            r[i].init(d);
            r[i].load(in + (base + i < n ? base + i : n - 1) * d);
        }
        float s1_l = 0.f, s2_l = 0.f, x0_l = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float s1, s2;
            r[i].sumsq_pair(IsHB<M>::value ? 1 : 0, slope, s1, s2);
            const float x0 = lane_get(r[i].v[0], 0);
            if (lane == i) { s1_l = s1; s2_l = s2; x0_l = x0; }
        }
        float f_l, t0_l;
        hyp_act_scalars<float>((M*)nullptr, s1_l, s2_l, x0_l, cin, cout, f_l, t0_l);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float f = lane_get(f_l, i);
            r[i].map([=](float v) { return f * (v > 0.f ? v : (v != v ? v : slope * v)); });
            if (IsHB<M>::value) r[i].set_first(lane_get(t0_l, i));
            if (base + i < n) r[i].store(out + (base + i) * d);
        }
    }
}

// HypAct for fp32 rows of at most 64 columns: 16 lanes per row, 32 rows per wavefront and pass (tile k of a lane's
// 16-lane group g is row base + 4 k + g, so each of the eight loads of a wavefront is 4 consecutive rows).  The scalar
// chain runs once per pass with row (k, g) in lane 16 g + k (lanes 8..15 of a group repeat rows 0..7), and its results
// go back to the row's 16 lanes with one DPP row broadcast each (the generic row kernel evaluates the chain in every lane
// of every row, 16 times over: 0.46-0.49 of the HBM peak at d = 64).
template <int K> __device__ __forceinline__ float row_bcast(float v) { return dpp_get<0x150 + K>(v); }   // row_newbcast:K
template <class M>
__global__ __launch_bounds__(256) void hyp_act_rows32(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                      int d, Curv cin, Curv cout, float slope) {
    constexpr int R = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    using TL = Tile<float, 16, 4, 4>;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 32; base < n; base += (int64_t)gridDim.x * 4 * 32) {
        TL r[R];
        float s1[R], s2[R], x0[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t row = base + 4 * k + g;
            r[k].init(d);
            r[k].load(in + (row < n ? row : n - 1) * d);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            r[k].sumsq_pair(IsHB<M>::value ? 1 : 0, slope, s1[k], s2[k]);
            x0[k] = IsHB<M>::value ? row_bcast<0>(r[k].v[0]) : 0.f;
        }
        float s1_l = s1[0], s2_l = s2[0], x0_l = x0[0];
#pragma unroll
        for (int k = 1; k < R; ++k) {
            const bool mine = (j & 7) == k;
            s1_l = mine ? s1[k] : s1_l;
            s2_l = mine ? s2[k] : s2_l;
            x0_l = mine ? x0[k] : x0_l;
        }
        float f_l, t0_l;
        hyp_act_scalars<float>((M*)nullptr, s1_l, s2_l, x0_l, cin, cout, f_l, t0_l);
#define HD_ROW(K)                                                                                     \
        {                                                                                             \
            const float f = row_bcast<K>(f_l);                                                        \
            r[K].map([=](float v) { return f * (v > 0.f ? v : (v != v ? v : slope * v)); });          \
            if (IsHB<M>::value) r[K].set_first(row_bcast<K>(t0_l));                                   \
            const int64_t row = base + 4 * K + g;                                                     \
            if (row < n) r[K].store(out + row * d);                                                   \
        }
        HD_ROW(0) HD_ROW(1) HD_ROW(2) HD_ROW(3) HD_ROW(4) HD_ROW(5) HD_ROW(6) HD_ROW(7)
#undef HD_ROW
    }
}

#define HD_ACT(fn, M)                                                                                  \
    extern "C" int fn##_f32(const void* in, void* out, int64_t n, int64_t d, hd_curv ci, hd_curv co,  \
                            int act, void* st) {                                                       \
        if (n > 0 && d >= 256 && d <= 512 && d % 4 == 0 && in && out && is_aligned16(in) && is_aligned16(out) && \
            (act == ACT_IDENTITY || act == ACT_RELU || act == ACT_LEAKY_RELU)) {                       \
            int64_t blocks = (n + 15) / 16;                                                            \
            blocks = blocks < 8192 ? blocks : 8192;                                                    \
            hyp_act_rows4<M<float>><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>(        \
                (const float*)in, (float*)out, n, (int)d, mk(ci), mk(co),                              \
                act == ACT_IDENTITY ? 1.f : (act == ACT_RELU ? 0.f : 0.01f));                          \
            return check_launch(#fn "_f32");                                                           \
        }                                                                                              \
        if (n >= 4096 && d >= 16 && d <= 64 && d % 4 == 0 && in && out && is_aligned16(in) && is_aligned16(out) && \
            (act == ACT_IDENTITY || act == ACT_RELU || act == ACT_LEAKY_RELU)) {                       \
            int64_t blocks = (n + 127) / 128;                                                          \
            blocks = blocks < 4096 ? blocks : 4096;                                                    \
            hyp_act_rows32<M<float>><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>(       \
                (const float*)in, (float*)out, n, (int)d, mk(ci), mk(co),                              \
                act == ACT_IDENTITY ? 1.f : (act == ACT_RELU ? 0.f : 0.01f));                          \
            return check_launch(#fn "_f32");                                                           \
        }                                                                                              \
        return launch_map1<float>(#fn "_f32", (const float*)in, (float*)out, n, d,                    \
                                  OpAct<M<float>, float>{mk(ci), mk(co), act}, (hipStream_t)st);      \
    }                                                                                                  \
    extern "C" int fn##_f64(const void* in, void* out, int64_t n, int64_t d, hd_curv ci, hd_curv co,  \
                            int act, void* st) {                                                       \
        return launch_map1<double>(#fn "_f64", (const double*)in, (double*)out, n, d,                 \
                                   OpAct<M<double>, double>{mk(ci), mk(co), act}, (hipStream_t)st);   \
    }
HD_ACT(hd_pb_hyp_act, PB)
HD_ACT(hd_hb_hyp_act, HB)

#define HD_MAP2(fn, WHICH)                                                                             \
    extern "C" int fn##_f32(const void* a, const void* b, void* out, int64_t n, int64_t d, hd_curv c, \
                            void* st) {                                                                \
        return launch_map2<float, int32_t>(#fn "_f32", (const float*)a, nullptr, (const float*)b,     \
                                           (float*)out, n, d, Op2<float, WHICH>{mk(c)}, (hipStream_t)st); \
    }                                                                                                  \
    extern "C" int fn##_f64(const void* a, const void* b, void* out, int64_t n, int64_t d, hd_curv c, \
                            void* st) {                                                                \
        return launch_map2<double, int32_t>(#fn "_f64", (const double*)a, nullptr, (const double*)b,  \
                                            (double*)out, n, d, Op2<double, WHICH>{mk(c)}, (hipStream_t)st); \
    }
HD_MAP2(hd_pb_mobius_add, B_MOBIUS_ADD)
HD_MAP2(hd_pb_expmap, B_EXPMAP)      // a = p (base point), b = u (tangent)
HD_MAP2(hd_pb_logmap, B_LOGMAP)      // a = p1 (base point), b = p2

// base point gathered per row from a small table of centres: a = centres (K, d), labels (n,)
#define HD_MAP2_AT(fn, WHICH)                                                                          \
    extern "C" int fn##_f32(const void* centres, const int32_t* labels, int64_t k, const void* b,     \
                            void* out, int64_t n, int64_t d, hd_curv c, void* st) {                    \
        HD_REQUIRE(labels && k >= 1, #fn ": labels/centres missing");                                  \
        return launch_map2<float, int32_t>(#fn "_f32", (const float*)centres, labels, (const float*)b, \
                                           (float*)out, n, d, Op2<float, WHICH>{mk(c)}, (hipStream_t)st, k); \
    }                                                                                                  \
    extern "C" int fn##_f64(const void* centres, const int32_t* labels, int64_t k, const void* b,     \
                            void* out, int64_t n, int64_t d, hd_curv c, void* st) {                    \
        HD_REQUIRE(labels && k >= 1, #fn ": labels/centres missing");                                  \
        return launch_map2<double, int32_t>(#fn "_f64", (const double*)centres, labels, (const double*)b, \
                                            (double*)out, n, d, Op2<double, WHICH>{mk(c)}, (hipStream_t)st, k); \
    }
HD_MAP2_AT(hd_pb_expmap_at, B_EXPMAP)
HD_MAP2_AT(hd_pb_logmap_at, B_LOGMAP)
// logmap at the centres of fp32 rows in fp64 (diffusion.forward :350-352 on encoder output: x.double() first): the rows
// are widened on load -- the same bits as a cast pass followed by hd_pb_logmap_at_f64, one (N, d) fp64 pass fewer
extern "C" int hd_pb_logmap_at_f32_f64(const void* centres, const int32_t* labels, int64_t k, const void* b, void* out,
                                       int64_t n, int64_t d, hd_curv c, void* st) {
    HD_REQUIRE(labels && k >= 1, "hd_pb_logmap_at_f32_f64: labels/centres missing");
    return launch_map2<double, int32_t, Op2<double, B_LOGMAP>, float>("hd_pb_logmap_at_f32_f64", (const double*)centres, labels,
                                                                     (const float*)b, (double*)out, n, d,
                                                                     Op2<double, B_LOGMAP>{mk(c)}, (hipStream_t)st, k);
}

#define HD_SQDIST(fn, M)                                                                               \
    extern "C" int fn##_f32(const void* a, const void* b, void* out, int64_t n, int64_t d, hd_curv c, \
                            void* st) {                                                                \
        return launch_reduce2<float>(#fn "_f32", (const float*)a, (const float*)b, (float*)out, n, d, \
                                     OpSqdist<M<float>, float>{mk(c)}, (hipStream_t)st);              \
    }                                                                                                  \
    extern "C" int fn##_f64(const void* a, const void* b, void* out, int64_t n, int64_t d, hd_curv c, \
                            void* st) {                                                                \
        return launch_reduce2<double>(#fn "_f64", (const double*)a, (const double*)b, (double*)out, n, d, \
                                      OpSqdist<M<double>, double>{mk(c)}, (hipStream_t)st);           \
    }
HD_SQDIST(hd_pb_sqdist, PB)
HD_SQDIST(hd_hb_sqdist, HB)

#define HD_DECODE(fn, M)                                                                                \
    extern "C" int fn##_f32(const void* h, const int64_t* pairs, int64_t n_rows, void* out, int64_t n, \
                            int64_t d, hd_curv c, int fermi_dirac, double r, double t, void* st) {      \
        return launch_reduce2_gather<float>(#fn "_f32", (const float*)h, pairs, n_rows, (float*)out, n, d, \
                                            OpSqdist<M<float>, float>{mk(c)}, fermi_dirac, r, t, (hipStream_t)st); \
    }                                                                                                   \
    extern "C" int fn##_f64(const void* h, const int64_t* pairs, int64_t n_rows, void* out, int64_t n, \
                            int64_t d, hd_curv c, int fermi_dirac, double r, double t, void* st) {      \
        return launch_reduce2_gather<double>(#fn "_f64", (const double*)h, pairs, n_rows, (double*)out, n, d, \
                                             OpSqdist<M<double>, double>{mk(c)}, fermi_dirac, r, t, (hipStream_t)st); \
    }
HD_DECODE(hd_pb_edge_decode, PB)
HD_DECODE(hd_hb_edge_decode, HB)

#define HD_EPI(T, sfx)                                                                                   \
    extern "C" int hd_pb_matvec_epilogue_##sfx(const void* mx, const void* xnorm, void* out, int64_t n, \
                                               int64_t d, hd_curv c, int flags, void* st) {              \
        HD_REQUIRE(xnorm || n == 0, "hd_pb_matvec_epilogue: xnorm missing");                             \
        return launch_map1<T>("hd_pb_matvec_epilogue_" #sfx, (const T*)mx, (T*)out, n, d,               \
                              OpMatvecEpi<T>{mk(c), (const T*)xnorm, flags}, (hipStream_t)st);           \
    }                                                                                                    \
    extern "C" int hd_hb_linear_epilogue_##sfx(const void* mu, void* out, int64_t n, int64_t d,         \
                                               hd_curv c, int flags, void* st) {                         \
        return launch_map1<T>("hd_hb_linear_epilogue_" #sfx, (const T*)mu, (T*)out, n, d,               \
                              OpHbLinearEpi<T>{mk(c), flags}, (hipStream_t)st);                          \
    }                                                                                                    \
    extern "C" int hd_rownorm_##sfx(const void* x, void* out, int64_t n, int64_t d, void* st) {         \
        return launch_stream<T>("hd_rownorm_" #sfx, (const T*)x, (T*)out, n, d, S_ROWNORM, Curv{1, 1, 1, 1}, \
                                (hipStream_t)st);                                                        \
    }
HD_EPI(float, f32)
HD_EPI(double, f64)

// ---- HypLinear on the hyperboloid with a narrow input (d_in <= 64), one kernel --------------------
// layers/hyp_layers.py:98-103 + manifolds/hyperboloid.py:103-123: logmap0(x) -> u W^T -> expmap0 (-> proj
// -> logmap0).  With a few dozen input columns the contraction is a handful of FMAs per output (the first
// HGCN layer of the graph-set configs: 21 features -> 512), so the three launches of the wide path (operand
// producer, matrix-core GEMM, row tail) spend their time on launch ramps and on writing / re-reading mu.
// Here a wavefront owns R rows: it reads x, takes logmap0, accumulates mu = u W^T as an fp32 fma chain over
// k against W^T ((d_in, d_out) row-major, a few tens of KB: copied to LDS once per 16-wavefront workgroup, or
// streamed from L2 when it does not fit; 16-byte reads, each reused for the R rows), applies the row tail to the accumulators (they already sit in the
// row-tile layout of the row kernels) and writes the result: x is read once, the output written once,
// nothing in between touches memory.
namespace {

// expmap0 (-> proj) (-> logmap0) of a row whose spatial sum of squares is s2, as ONE factor f on the spatial
// part and the new time coordinate t0 (same clamp order as HB<T>::expmap0 / proj / logmap0; the norm of an
// already scaled row is taken analytically, |f x| = f |x|, as hyp_act_homog does: rounding-level difference).
__device__ __forceinline__ void hb_linear_tail_scalars(float s2, int flags, const Curv& cv, float& f, float& t0) {
    const float sK = (float)cv.sK;
    const float n0 = hd_sqrt(s2);
    const float n = clamp_min(n0, (float)kMinNorm);
    float shv, chv;
    sinhcosh15(n / sK, shv, chv);
    const float f1 = sK * shv / n;
    const float m1 = f1 * n0;
    const float x0 = hd_sqrt(clamp_min((float)cv.K + m1 * m1, (float)Lim<float>::hb_eps));
    f = f1;
    t0 = x0;
    if (flags & 2) {
        const float n2 = clamp_min(m1, (float)kMinNorm);
        const float th = clamp_min(x0 / sK, (float)(1.0 + Lim<float>::hb_eps));
        f = f1 * (sK * arcosh_ref(th) / n2);
        t0 = 0.f;
    }
}

// The per-row scalar chains (logmap0's arcosh, the tail's sinh / cosh / arcosh: ~50 quarter-rate
// transcendental issues per row) are wavefront-uniform in the row kernels, i.e. computed 64 times over.
// Here a wavefront owns R rows and evaluates each chain ONCE for all of them, row r in lane r, then hands
// the factors back with v_readlane: the kernel is bound by exactly this VALU work otherwise.
template <int R, int WAVES, bool LDSW>
__global__ __launch_bounds__(WAVES * 64) void hb_linear_smallk(const float* __restrict__ x, const float* __restrict__ Wt_g,
                                                        float* __restrict__ out, int64_t n, int K, int N,
                                                        Curv cv, int flags) {
    extern __shared__ float4 Ws4[];                          // LDSW: W^T copied once per workgroup (K * N floats)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Wt = Wt_g;
    if constexpr (LDSW) {
        const float4* src = reinterpret_cast<const float4*>(Wt_g);
        for (int i = tid; i < K * N / 4; i += WAVES * 64) Ws4[i] = src[i];
        __syncthreads();
        Wt = reinterpret_cast<const float*>(Ws4);
    }
    using TL = Tile<float, 64, 4, 8>;
    const float sK = (float)cv.sK;
    const int c0 = 4 * lane, c1 = 256 + 4 * lane;
    for (int64_t base = ((int64_t)blockIdx.x * WAVES + wave) * R; base < n; base += (int64_t)gridDim.x * WAVES * R) {
        float xv[R], u[R];
        float ss_l = 0.f, x0_l = 0.f;                        // lane r: row r's spatial |x|^2 and time coordinate
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = base + r < n ? base + r : n - 1;
            xv[r] = lane < K ? x[row * K + lane] : 0.f;
            const float ss = group_sum<64>(lane >= 1 ? xv[r] * xv[r] : 0.f);     // hyperboloid.py:105-106
            const float x0 = lane_bcast0(xv[r]);
            if (lane == r) { ss_l = ss; x0_l = x0; }
        }
        {                                                   // logmap0 (hyperboloid.py:103-113), R rows at once
            const float nrm = clamp_min(hd_sqrt(ss_l), (float)kMinNorm);
            const float theta = clamp_min(x0_l / sK, (float)(1.0 + Lim<float>::hb_eps));
            const float f_l = sK * arcosh_ref(theta) / nrm;
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = lane == 0 ? 0.f : xv[r] * lane_get(f_l, r);
        }
        TL acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r].init(N);
            acc[r].zero();
        }
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float* wk = Wt + (int64_t)k * N;
            float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
            if (c0 < N) w0 = *reinterpret_cast<const float4*>(wk + c0);
            if (c1 < N) w1 = *reinterpret_cast<const float4*>(wk + c1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float uk = lane_get(u[r], k);
                acc[r].v[0] = fmaf(uk, w0.x, acc[r].v[0]);
                acc[r].v[1] = fmaf(uk, w0.y, acc[r].v[1]);
                acc[r].v[2] = fmaf(uk, w0.z, acc[r].v[2]);
                acc[r].v[3] = fmaf(uk, w0.w, acc[r].v[3]);
                acc[r].v[4] = fmaf(uk, w1.x, acc[r].v[4]);
                acc[r].v[5] = fmaf(uk, w1.y, acc[r].v[5]);
                acc[r].v[6] = fmaf(uk, w1.z, acc[r].v[6]);
                acc[r].v[7] = fmaf(uk, w1.w, acc[r].v[7]);
            }
        }
        float s2_l = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s2 = acc[r].sumsq(1);
            if (lane == r) s2_l = s2;
        }
        float f_l, t0_l;
        hb_linear_tail_scalars(s2_l, flags, cv, f_l, t0_l);  // expmap0 (-> proj) (-> logmap0), R rows at once
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = base + r;
            acc[r].scale(lane_get(f_l, r), 1);
            acc[r].set_first(lane_get(t0_l, r));
            if (row < n) acc[r].store(out + row * N);
        }
    }
}

// The same fused layer on the matrix cores (GEMM_MODE bf16x3 / f16x2): with K <= 32 the contraction is ONE
// k-step of v_mfma_f32_16x16x32_bf16 per 16x16 output tile and limb pair (6 per tile: fp32-faithful, dropped
// terms < 2^-26 |u||w|), ~1.3 us of matrix time for 15 777 x 21 -> 512 against ~5 us of fp32 FMAs.  A wavefront
// owns 16 rows: lane (r16, kq) reads x[row r16][8 kq .. 8 kq + 7], takes logmap0 (4 lanes per row, two
// shuffles), splits u into three bf16 limbs in registers -- the A operand never touches memory -- and multiplies
// it with the pre-split W image (hd_smallk_pack_w_f32: [n-tile][limb][lane] uint4, staged once per workgroup
// into LDS by LDS-DMA).  The accumulators hold row r16, columns 16 jt + 4 kq .. + 3: the row tail reduces over
// them with two shuffles and the result is written as float4s.
template <int NT>
__global__ __launch_bounds__(256) void hb_linear_smallk_mfma(const float* __restrict__ x, const uint4* __restrict__ Wimg,
                                                             float* __restrict__ out, int64_t n, int K, int N,
                                                             Curv cv, int flags) {
    extern __shared__ uint4 wl[];                            // [jt][limb][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int nt = N >> 4;
    for (int p = wave; p < nt * 3; p += 4) limb_glds16(Wimg + p * 64 + lane, wl + p * 64);
    const int64_t row_raw = ((int64_t)blockIdx.x * 4 + wave) * 16 + r16;
    const int64_t row = row_raw < n ? row_raw : n - 1;
    const float sK = (float)cv.sK;
    float xv[8], ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kq + e;
        xv[e] = k < K ? x[row * K + k] : 0.f;
        if (k >= 1) ss = fmaf(xv[e], xv[e], ss);             // hyperboloid.py:105-106
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float x0 = __shfl(xv[0], r16);                     // time coordinate: lane (r16, kq = 0)
    uint4 ua[3];
    {                                                        // logmap0 (hyperboloid.py:103-113)
        const float nrm = clamp_min(hd_sqrt(ss), (float)kMinNorm);
        const float theta = clamp_min(x0 / sK, (float)(1.0 + Lim<float>::hb_eps));
        const float f = sK * arcosh_ref(theta) / nrm;
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = (8 * kq + e == 0) ? 0.f : xv[e] * f;
        limb_split8(u, ua[0], ua[1], ua[2]);
    }
    lf32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = lf32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // operand maps of v_mfma_f32_16x16x32: first operand = W rows (n = 16 jt + r16), second = u rows (m = r16),
    // k = 8 kq .. 8 kq + 7; D column = lane & 15 = m, D rows 4 kq + reg = n
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j < nt) {
            const uint4 w0 = wl[(j * 3 + 0) * 64 + lane], w1 = wl[(j * 3 + 1) * 64 + lane], w2 = wl[(j * 3 + 2) * 64 + lane];
            lf32x4 c = acc[j];
#define HD_MM(wv, av) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, wv), \
                                                          __builtin_bit_cast(lbf16x8, av), c, 0, 0, 0);
            HD_MM(w0, ua[2]) HD_MM(w2, ua[0]) HD_MM(w1, ua[1]) HD_MM(w0, ua[1]) HD_MM(w1, ua[0]) HD_MM(w0, ua[0])
#undef HD_MM
            acc[j] = c;
        }
    }
    float s2 = 0.f;                                          // spatial |mu|^2 of row r16 (column 0 excluded)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (!(j == 0 && r == 0)) s2 = fmaf(acc[j][r], acc[j][r], s2);
    if (kq != 0) s2 = fmaf(acc[0][0], acc[0][0], s2);
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    float f2, t0;
    hb_linear_tail_scalars(s2, flags, cv, f2, t0);           // expmap0 (-> proj) (-> logmap0)
    if (row_raw >= n) return;
    float* o = out + row * N + 4 * kq;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j < nt) {
            float4 v = make_float4(acc[j][0] * f2, acc[j][1] * f2, acc[j][2] * f2, acc[j][3] * f2);
            if (j == 0 && kq == 0) v.x = t0;
            *reinterpret_cast<float4*>(o + 16 * j) = v;
        }
    }
}

// W (N, K) fp32 row-major (row stride ld) -> the operand image of hb_linear_smallk_mfma: uint4 index
// (jt * 3 + limb) * 64 + lane holds limb `limb` of W[16 jt + (lane & 15)][8 (lane >> 4) .. + 7], zero for k >= K
__global__ __launch_bounds__(64) void smallk_pack_w(const float* __restrict__ W, int64_t ld, int N, int K,
                                                    uint4* __restrict__ img) {
    const int jt = blockIdx.x, lane = threadIdx.x;
    const int nrow = 16 * jt + (lane & 15), k0 = 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (nrow < N && k0 + e < K) ? W[(int64_t)nrow * ld + k0 + e] : 0.f;
    uint4 h, m, l;
    limb_split8(v, h, m, l);
    img[(jt * 3 + 0) * 64 + lane] = h;
    img[(jt * 3 + 1) * 64 + lane] = m;
    img[(jt * 3 + 2) * 64 + lane] = l;
}

// hd_hb_linear_epilogue for fp32 rows of up to 512 columns: four rows per wavefront, scalar chain once
template <int R>
__global__ __launch_bounds__(256) void hb_linear_tail_rows(const float* __restrict__ mu, float* __restrict__ out,
                                                           int64_t n, int d, Curv cv, int flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using TL = Tile<float, 64, 4, 8>;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * R; base < n; base += (int64_t)gridDim.x * 4 * R) {
        TL r[R];
        float s2_l = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            r[i].init(d);
            r[i].load(mu + (base + i < n ? base + i : n - 1) * d);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float s2 = r[i].sumsq(1);
            if (lane == i) s2_l = s2;
        }
        float f_l, t0_l;
        hb_linear_tail_scalars(s2_l, flags, cv, f_l, t0_l);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            r[i].scale(lane_get(f_l, i), 1);
            r[i].set_first(lane_get(t0_l, i));
            if (base + i < n) r[i].store(out + (base + i) * d);
        }
    }
}

}  // namespace

// fp32 rows short enough for one 64-lane tile: the four-rows-per-wavefront kernel above
extern "C" int hd_hb_linear_tail_rows_f32(const void* mu, void* out, int64_t n, int64_t d, hd_curv c, int flags,
                                          void* st) {
    HD_REQUIRE(n >= 0 && d >= 4 && d <= 512 && d % 4 == 0, "hd_hb_linear_tail_rows_f32: dim must be a multiple of 4, <= 512");
    if (n == 0) return HD_OK;
    HD_REQUIRE(mu && out && is_aligned16(mu) && is_aligned16(out), "hd_hb_linear_tail_rows_f32: null or misaligned pointer");
    int64_t blocks = (n + 15) / 16;
    blocks = blocks < 4096 ? blocks : 4096;
    hb_linear_tail_rows<4><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>((const float*)mu, (float*)out, n,
                                                                                      (int)d, mk(c), flags);
    return check_launch("hd_hb_linear_tail_rows_f32");
}

extern "C" int64_t hd_smallk_image_bytes(int64_t d_out) { return d_out < 1 ? -1 : (d_out + 15) / 16 * 3 * 64 * 16; }

extern "C" int hd_smallk_pack_w_f32(const void* W, int64_t ld, int64_t d_out, int64_t d_in, void* img, void* st) {
    HD_REQUIRE(d_out >= 1 && d_in >= 1 && d_in <= 32 && ld >= d_in, "hd_smallk_pack_w_f32: d_in must be <= 32");
    HD_REQUIRE(W && img && is_aligned16(img), "hd_smallk_pack_w_f32: null or misaligned pointer");
    HD_REQUIRE(d_out < (1LL << 24), "hd_smallk_pack_w_f32: d_out too large");
    smallk_pack_w<<<dim3((unsigned)((d_out + 15) / 16)), dim3(64), 0, (hipStream_t)st>>>(
        (const float*)W, ld, (int)d_out, (int)d_in, (uint4*)img);
    return check_launch("hd_smallk_pack_w_f32");
}

extern "C" int hd_hb_linear_smallk_bf16x3_f32(const void* x, const void* Wimg, void* out, int64_t rows, int64_t d_in,
                                              int64_t d_out, hd_curv c, int flags, void* st) {
    HD_REQUIRE(rows >= 0 && d_in >= 1 && d_in <= 32 && d_out >= 16 && d_out <= 512 && d_out % 16 == 0,
               "hd_hb_linear_smallk_bf16x3_f32: shape outside the fused path (d_in <= 32, d_out <= 512 and a multiple of 16)");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && Wimg && out && is_aligned16(out) && is_aligned16(Wimg),
               "hd_hb_linear_smallk_bf16x3_f32: null or misaligned pointer");
    HD_REQUIRE(rows < (1LL << 31) * 64, "hd_hb_linear_smallk_bf16x3_f32: too many rows");
    const size_t lds = (size_t)(d_out / 16) * 3 * 64 * 16;
    const dim3 grid((unsigned)((rows + 63) / 64));
#define HD_SKM(NT_)                                                                                            \
    {                                                                                                          \
        if (lds > 64 * 1024) {                                                                                 \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hb_linear_smallk_mfma<NT_>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) {                                                                             \
                set_error("hd_hb_linear_smallk_bf16x3_f32: hipFuncSetAttribute(%zu B of LDS): %s", lds,        \
                          hipGetErrorString(e));                                                               \
                return HD_EHIP;                                                                                \
            }                                                                                                  \
        }                                                                                                      \
        hb_linear_smallk_mfma<NT_><<<grid, dim3(256), lds, (hipStream_t)st>>>(                                 \
            (const float*)x, (const uint4*)Wimg, (float*)out, rows, (int)d_in, (int)d_out, mk(c), flags);      \
    }
    if (d_out <= 128) HD_SKM(8) else if (d_out <= 256) HD_SKM(16) else HD_SKM(32)
#undef HD_SKM
    return check_launch("hd_hb_linear_smallk_bf16x3_f32");
}

extern "C" int hd_hb_linear_smallk_f32(const void* x, const void* Wt, void* out, int64_t rows, int64_t d_in,
                                       int64_t d_out, hd_curv c, int flags, void* st) {
    HD_REQUIRE(rows >= 0 && d_in >= 1 && d_in <= 64 && d_out >= 4 && d_out <= 512 && d_out % 4 == 0,
               "hd_hb_linear_smallk_f32: shape outside the fused path (d_in <= 64, d_out <= 512 and a multiple of 4)");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && Wt && out && is_aligned16(out) && is_aligned16(Wt), "hd_hb_linear_smallk_f32: null or misaligned pointer");
    int R = 4;                                           // measured: 2 -> 25.6 us, 4 -> 23.0, 8 -> 26.4 (15 777 x 24 -> 512)
    if (const char* e = getenv("HD_SMALLK_R")) R = atoi(e);
#define HD_SK(R_)                                                                                              \
    {                                                                                                          \
        int64_t blocks = (rows + 4 * R_ - 1) / (4 * R_);                                                       \
        blocks = blocks < 4096 ? blocks : 4096;                                                                \
        hb_linear_smallk<R_, 4, false><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>(             \
            (const float*)x, (const float*)Wt, (float*)out, rows, (int)d_in, (int)d_out, mk(c), flags);        \
    }
    int lds_mode = 2;                                    // measured in the bench step: 0 (W^T from L2) 33 us, 1-3: 21 us
    if (const char* e = getenv("HD_SMALLK_LDS")) lds_mode = atoi(e);
    const size_t wbytes = (size_t)d_in * d_out * sizeof(float);
    if (lds_mode && wbytes <= 64 * 1024) {
        // W^T in LDS, 16 wavefronts per workgroup: lds_mode = workgroups per CU (256 CUs assumed for the grid)
        int64_t blocks = (rows + 16 * 4 - 1) / (16 * 4);
        const int64_t cap = 256LL * lds_mode;
        blocks = blocks < cap ? blocks : cap;
        hb_linear_smallk<4, 16, true><<<dim3((unsigned)blocks), dim3(1024), wbytes, (hipStream_t)st>>>(
            (const float*)x, (const float*)Wt, (float*)out, rows, (int)d_in, (int)d_out, mk(c), flags);
    } else if (R == 2) HD_SK(2) else if (R == 8) HD_SK(8) else HD_SK(4)
#undef HD_SK
    return check_launch("hd_hb_linear_smallk_f32");
}
