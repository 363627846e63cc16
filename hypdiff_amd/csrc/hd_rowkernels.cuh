// Generic row kernels: one group of G lanes per row, row in registers.
#pragma once
#include <type_traits>
#include "hd_manifold.cuh"

namespace hd {

template <int N> using IC = std::integral_constant<int, N>;

// calls f(IC<G>, IC<VEC>) for the configuration picked for (T, d, alignment)
template <typename T, class F>
inline int dispatch_cfg(const RowCfg& c, F&& f) {
    constexpr int V = VecOf<T>::N;
    if (c.vec == 1) {
        if (c.g == 4) return f(IC<4>{}, IC<1>{});
        if (c.g == 16) return f(IC<16>{}, IC<1>{});
        return f(IC<64>{}, IC<1>{});
    }
    if (c.g == 4) return f(IC<4>{}, IC<V>{});
    if (c.g == 16) return f(IC<16>{}, IC<V>{});
    return f(IC<64>{}, IC<V>{});
}

// Rows per lane group and loop iteration.  A row that needs only 4 (8) of the 16 tile
// registers leaves room for 4 (2) independent rows per group: their loads are issued back
// to back and their norm -> transcendental -> rescale chains interleave, which is what a
// latency-bound composite map (HypAct: three dependent reductions per row) needs to keep
// HBM busy.  Batching is used when the launch still has >= 1024 workgroups; fp64 rows and
// short launches run one row per group (REGS = 16).
template <typename T>
inline int pick_regs(const RowCfg& c, int64_t n, int64_t d) {
    if (sizeof(T) != 4) return kRegs;
    const int64_t span = (int64_t)c.g * c.vec;
    const int need = (int)((d + span - 1) / span) * c.vec;
    int regs = need <= 4 ? 4 : (need <= 8 ? 8 : kRegs);
    const int64_t rpb = kBlock / c.g;
    while (regs < kRegs && n / (rpb * (kRegs / regs)) < 1024) regs *= 2;
    return regs;
}

// calls f(IC<G>, IC<VEC>, IC<REGS>)
template <typename T, class F>
inline int dispatch_cfg_regs(const RowCfg& c, int regs, F&& f) {
    return dispatch_cfg<T>(c, [&](auto g, auto v) {
        constexpr int V = decltype(v)::value;
        if constexpr (sizeof(T) == 4) {
            if (regs == 4 && V <= 4) return f(g, v, IC<4>{});
            if (regs == 8) return f(g, v, IC<8>{});
        }
        return f(g, v, IC<kRegs>{});
    });
}

inline int grid_for_rows_batched(int64_t rows, int g, int r) {
    const int64_t rows_per_block = (int64_t)(kBlock / g) * r;
    int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks < 1) blocks = 1;
    const int64_t cap = 256 * 16;
    return (int)(blocks < cap ? blocks : cap);
}

// out[row] = op(in[row]).  R = kRegs / REGS rows per group and iteration; a slot past the end
// recomputes the last row and skips the store, so loads and reductions stay unconditional.
template <class Op, typename T, int G, int VEC, int REGS>
__global__ __launch_bounds__(kBlock) void row_map1(const T* __restrict__ in, T* __restrict__ out,
                                                   int64_t n, int d, Op op) {
    constexpr int RPB = kBlock / G, R = kRegs / REGS;
    Tile<T, G, VEC, REGS> r[R];
#pragma unroll
    for (int i = 0; i < R; ++i) r[i].init(d);
    const int g = threadIdx.x / G;
    for (int64_t base = (int64_t)blockIdx.x * (RPB * R); base < n; base += (int64_t)gridDim.x * (RPB * R)) {
        int64_t row[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            row[i] = base + i * RPB + g;
            r[i].load(in + (row[i] < n ? row[i] : n - 1) * d);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) op(r[i], row[i] < n ? row[i] : n - 1);
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (row[i] < n) r[i].store(out + row[i] * d);
    }
}

// out[row] = op(a[row or gathered], b[row]); ia (optional) gathers rows of a
// (TB: the type b is stored in -- T, or float for an fp64 map of fp32 rows, widened on load)
template <class Op, typename T, typename I, int G, int VEC, int REGS, typename TB = T>
__global__ __launch_bounds__(kBlock) void row_map2(const T* __restrict__ a, const I* __restrict__ ia,
                                                   const TB* __restrict__ b, T* __restrict__ out,
                                                   int64_t n, int d, Op op, int64_t na) {
    constexpr int RPB = kBlock / G, R = kRegs / REGS;
    Tile<T, G, VEC, REGS> x[R], y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { x[i].init(d); y[i].init(d); }
    const int g = threadIdx.x / G;
    for (int64_t base = (int64_t)blockIdx.x * (RPB * R); base < n; base += (int64_t)gridDim.x * (RPB * R)) {
        int64_t row[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            row[i] = base + i * RPB + g;
            const int64_t rl = row[i] < n ? row[i] : n - 1;
            int64_t ra = ia ? (int64_t)ia[rl] : rl;
            if (ia) ra = ra < 0 ? 0 : (ra >= na ? na - 1 : ra);      // a label outside the table must not fault
            x[i].load(a + ra * d);
            if constexpr (sizeof(TB) == sizeof(T)) y[i].load(reinterpret_cast<const T*>(b) + rl * d);
            else y[i].load_cvt(b + rl * d);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) op(x[i], y[i], row[i] < n ? row[i] : n - 1);      // result left in y
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (row[i] < n) y[i].store(out + row[i] * d);
    }
}

// out[row] = scalar op(a[row], b[row])
template <class Op, typename T, int G, int VEC, int REGS>
__global__ __launch_bounds__(kBlock) void row_reduce2(const T* __restrict__ a, const T* __restrict__ b,
                                                      T* __restrict__ out, int64_t n, int d, Op op) {
    constexpr int RPB = kBlock / G, R = kRegs / REGS;
    Tile<T, G, VEC, REGS> x[R], y[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { x[i].init(d); y[i].init(d); }
    const int g = threadIdx.x / G;
    for (int64_t base = (int64_t)blockIdx.x * (RPB * R); base < n; base += (int64_t)gridDim.x * (RPB * R)) {
        int64_t row[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            row[i] = base + i * RPB + g;
            const int64_t rl = row[i] < n ? row[i] : n - 1;
            x[i].load(a + rl * d);
            y[i].load(b + rl * d);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const T s = op(x[i], y[i]);
            if (x[i].j == 0 && row[i] < n) out[row[i]] = s;
        }
    }
}

// out[e] = post(op(h[ia[e]], h[ib[e]])): edge decoder, rows gathered by int64 index pairs
// (models/base_models.py:152-159); fd != 0 applies the Fermi-Dirac map 1/(exp((s-r)/t)+1)
template <class Op, typename T, int G, int VEC>
__global__ __launch_bounds__(kBlock) void row_reduce2_gather(const T* __restrict__ h, const int64_t* __restrict__ pairs,
                                                             int64_t pair_stride, int64_t n_rows, T* __restrict__ out,
                                                             int64_t n, int d, Op op, int fd, T fr, T ft) {
    constexpr int RPB = kBlock / G;
    Tile<T, G, VEC> x, y;
    x.init(d);
    y.init(d);
    for (int64_t e = (int64_t)blockIdx.x * RPB + threadIdx.x / G; e < n; e += (int64_t)gridDim.x * RPB) {
        int64_t ia = pairs[e * pair_stride], ib = pairs[e * pair_stride + 1];
        ia = ia < 0 ? 0 : (ia >= n_rows ? n_rows - 1 : ia);
        ib = ib < 0 ? 0 : (ib >= n_rows ? n_rows - 1 : ib);
        x.load(h + ia * d);
        y.load(h + ib * d);
        T s = op(x, y);
        if (fd) s = T(1) / ((T)exp((double)((s - fr) / ft)) + T(1));
        if (x.j == 0) out[e] = s;
    }
}

template <typename T, class Op>
inline int launch_reduce2_gather(const char* name, const T* h, const int64_t* pairs, int64_t n_rows, T* out, int64_t n,
                                 int64_t d, Op op, int fd, double fr, double ft, hipStream_t st) {
    HD_REQUIRE(n >= 0 && d >= 1 && n_rows >= 1, "%s: bad shape", name);
    if (n == 0) return HD_OK;
    HD_REQUIRE(h && pairs && out, "%s: null pointer", name);
    const RowCfg c = pick_cfg<T>(d, is_aligned16(h));
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    return dispatch_cfg<T>(c, [&](auto g, auto v) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value;
        row_reduce2_gather<Op, T, G, V><<<grid_for_rows(n, G), kBlock, 0, st>>>(h, pairs, 2, n_rows, out, n, (int)d, op,
                                                                               fd, (T)fr, (T)ft);
        return check_launch(name);
    });
}

template <typename T, class Op>
inline int launch_map1(const char* name, const T* in, T* out, int64_t n, int64_t d, Op op, hipStream_t st) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (n == 0) return HD_OK;
    HD_REQUIRE(in && out, "%s: null pointer", name);
    const RowCfg c = pick_cfg<T>(d, is_aligned16(in) && is_aligned16(out));
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit (use the streaming entry point)", name, (long long)d);
    return dispatch_cfg_regs<T>(c, pick_regs<T>(c, n, d), [&](auto g, auto v, auto rg) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value, RG = decltype(rg)::value;
        row_map1<Op, T, G, V, RG><<<grid_for_rows_batched(n, G, kRegs / RG), kBlock, 0, st>>>(in, out, n, (int)d, op);
        return check_launch(name);
    });
}

template <typename T, typename I, class Op, typename TB = T>
inline int launch_map2(const char* name, const T* a, const I* ia, const TB* b, T* out, int64_t n, int64_t d,
                       Op op, hipStream_t st, int64_t na = 0) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (n == 0) return HD_OK;
    HD_REQUIRE(a && b && out, "%s: null pointer", name);
    const RowCfg c = pick_cfg<T>(d, is_aligned16(a) && is_aligned16(b) && is_aligned16(out));
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    return dispatch_cfg_regs<T>(c, pick_regs<T>(c, n, d), [&](auto g, auto v, auto rg) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value, RG = decltype(rg)::value;
        row_map2<Op, T, I, G, V, RG, TB><<<grid_for_rows_batched(n, G, kRegs / RG), kBlock, 0, st>>>(a, ia, b, out, n,
                                                                                                   (int)d, op, na);
        return check_launch(name);
    });
}

template <typename T, class Op>
inline int launch_reduce2(const char* name, const T* a, const T* b, T* out, int64_t n, int64_t d, Op op,
                          hipStream_t st) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (n == 0) return HD_OK;
    HD_REQUIRE(a && b && out, "%s: null pointer", name);
    const RowCfg c = pick_cfg<T>(d, is_aligned16(a) && is_aligned16(b));
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    return dispatch_cfg_regs<T>(c, pick_regs<T>(c, n, d), [&](auto g, auto v, auto rg) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value, RG = decltype(rg)::value;
        row_reduce2<Op, T, G, V, RG><<<grid_for_rows_batched(n, G, kRegs / RG), kBlock, 0, st>>>(a, b, out, n, (int)d,
                                                                                               op);
        return check_launch(name);
    });
}

// ---- streaming path for rows longer than the register tile ------------------------
// One wavefront per row, two sweeps: sum of squares, then scale.  The second sweep
// re-reads the row from L2.  Norms of already scaled rows are taken analytically
// (|f*x| = |f||x|), which differs from the literal recomputation by rounding only.
enum StreamOp { S_PB_PROJ = 0, S_PB_EXPMAP0, S_PB_LOGMAP0, S_PB_EXPMAP0_PROJ,
                S_HB_PROJ, S_HB_EXPMAP0, S_HB_LOGMAP0, S_ROWNORM };

template <typename T>
__global__ __launch_bounds__(kBlock) void row_stream(const T* __restrict__ in, T* __restrict__ out,
                                                     int64_t n, int64_t d, int op, Curv cv) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const bool hb = (op == S_HB_PROJ || op == S_HB_EXPMAP0 || op == S_HB_LOGMAP0);
    // fp32 rows of 16-byte multiples (config 4's 2708-wide identity features): 16-byte accesses in both sweeps
    const bool vec4 = sizeof(T) == 4 && !hb && (d & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    for (int64_t row = wave; row < n; row += nwaves) {
        const T* x = in + row * d;
        T ss = T(0);
        if (vec4) {
            const float4* xv = reinterpret_cast<const float4*>(x);
            for (int64_t i = lane; i < (d >> 2); i += 64) {
                const float4 q = xv[i];
                ss += (T)(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            }
        } else {
            for (int64_t i = lane + (hb ? 64 : 0); i < d; i += 64) ss += x[i] * x[i];
            if (hb) {       // first 64 columns without the time coordinate
                const int64_t i = lane;
                if (i >= 1 && i < d) ss += x[i] * x[i];
            }
        }
        ss = group_sum<64>(ss);
        const T nrm = hd_sqrt(ss);
        if (op == S_ROWNORM) {
            if (lane == 0) out[row] = nrm;
            continue;
        }
        const T n1 = clamp_min(nrm, (T)kMinNorm);
        T f = T(1), t0 = T(0);
        bool set_t0 = false;
        const T maxnorm = (T)((1.0 - Lim<T>::pb_eps) / cv.sc);
        const T sc = (T)cv.sc, sK = (T)cv.sK;
        switch (op) {
            case S_PB_PROJ: f = n1 > maxnorm ? maxnorm / n1 : T(1); break;
            case S_PB_EXPMAP0: f = tanh15(sc * n1) / (sc * n1); break;
            case S_PB_LOGMAP0: f = T(1) / sc * artanh_ref(sc * n1) / n1; break;
            case S_PB_EXPMAP0_PROJ: {
                f = tanh15(sc * n1) / (sc * n1);
                const T n2 = clamp_min(f * nrm, (T)kMinNorm);
                if (n2 > maxnorm) f = f / n2 * maxnorm;
            } break;
            case S_HB_PROJ:
                t0 = hd_sqrt(clamp_min((T)cv.K + nrm * nrm, (T)Lim<T>::hb_eps)); set_t0 = true; break;
            case S_HB_EXPMAP0: {
                const T theta = n1 / sK;
                T shv, chv;
                sinhcosh15(theta, shv, chv);
                f = sK * shv / n1;
                const T n2 = f * nrm;
                t0 = hd_sqrt(clamp_min((T)cv.K + n2 * n2, (T)Lim<T>::hb_eps)); set_t0 = true;
            } break;
            case S_HB_LOGMAP0: {
                const T theta = clamp_min(x[0] / sK, (T)(1.0 + Lim<T>::hb_eps));
                f = sK * arcosh_ref(theta) / n1; t0 = T(0); set_t0 = true;
            } break;
        }
        T* y = out + row * d;
        if (vec4) {
            const float4* xv = reinterpret_cast<const float4*>(x);
            float4* yv = reinterpret_cast<float4*>(y);
            const float ff = (float)f;
            for (int64_t i = lane; i < (d >> 2); i += 64) {
                const float4 q = xv[i];
                yv[i] = make_float4(q.x * ff, q.y * ff, q.z * ff, q.w * ff);
            }
            continue;
        }
        for (int64_t i = lane; i < d; i += 64) y[i] = (set_t0 && i == 0) ? t0 : x[i] * f;
    }
}

template <typename T>
inline int launch_stream(const char* name, const T* in, T* out, int64_t n, int64_t d, int op, const Curv& cv,
                         hipStream_t st) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (n == 0) return HD_OK;
    HD_REQUIRE(in && out, "%s: null pointer", name);
    row_stream<T><<<grid_for_rows(n, 64), kBlock, 0, st>>>(in, out, n, d, op, cv);
    return check_launch(name);
}

}  // namespace hd
