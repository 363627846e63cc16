// C-ABI entry points: fused per-row manifold maps (Poincare ball and hyperboloid).
// Memory-bound kernels: every row is read once and written once; all norms, clamps,
// artanh/arcosh and rescales of a composite map happen in registers in between.
#include "hd_rowkernels.cuh"
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

#define HD_ACT(fn, M)                                                                                  \
    extern "C" int fn##_f32(const void* in, void* out, int64_t n, int64_t d, hd_curv ci, hd_curv co,  \
                            int act, void* st) {                                                       \
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
