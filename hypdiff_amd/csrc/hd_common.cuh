// Shared device/host helpers for the hypdiff HIP library (gfx950 / CDNA4 only).
//
// Row tiles: a row of `d` elements is held in registers by a group of G lanes
// (G in {4,16,64}, groups are aligned sub-ranges of one 64-lane wavefront).  Lane j
// of the group owns elements (k*G + j)*VEC + e, k < ch, e < VEC, so every wave-level
// load instruction of a group touches G*VEC consecutive elements (coalesced), and
// consecutive groups of a wave own consecutive rows.  Reductions over a row are
// butterfly __shfl_xor exchanges that never leave the group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

namespace hd {

// ---- error reporting through the C ABI ------------------------------------------
enum { HD_OK = 0, HD_EINVAL = 1, HD_EHIP = 2, HD_EUNSUPPORTED = 3 };
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define HD_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            hd::set_error(__VA_ARGS__);            \
            return hd::HD_EINVAL;                  \
        }                                          \
    } while (0)

// Curvature bundle computed on the host with the reference's precision rules
// (python-float c: double; 1-element tensor c: in the tensor's dtype).
struct Curv {
    double c, sc, K, sK;
};

constexpr int kRegs = 16;           // register budget (elements) of one tile per lane
constexpr int kBlock = 256;         // threads per block for row kernels

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int N = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int N = 2; };

template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// value held by the first lane of the group
template <int G, typename T>
__device__ __forceinline__ T group_first(T v) {
    const int lane = threadIdx.x & 63;
    return __shfl(v, lane & ~(G - 1), 64);
}

template <typename T, int G, int VEC>
struct Tile {
    static constexpr int CHMAX = kRegs / VEC;
    T v[kRegs];
    int ch;      // chunks in use (uniform)
    int d;       // row length
    int j;       // lane within group

    __device__ __forceinline__ void init(int d_) {
        d = d_;
        j = threadIdx.x & (G - 1);
        ch = (d + G * VEC - 1) / (G * VEC);
    }
    // element index of register slot (k, e)
    __device__ __forceinline__ int col(int k, int e) const { return (k * G + j) * VEC + e; }

    __device__ __forceinline__ void load(const T* __restrict__ row) {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            const int c0 = (k * G + j) * VEC;
            if (k < ch && c0 < d) {
                if constexpr (VEC == 1) {
                    v[k] = row[c0];
                } else {
                    using V = typename VecOf<T>::type;
                    const V t = *reinterpret_cast<const V*>(row + c0);
                    const T* tp = reinterpret_cast<const T*>(&t);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[k * VEC + e] = tp[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k * VEC + e] = T(0);
            }
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < kRegs; ++i) v[i] = T(0);
    }
    __device__ __forceinline__ void store(T* __restrict__ row) const {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            const int c0 = (k * G + j) * VEC;
            if (k < ch && c0 < d) {
                if constexpr (VEC == 1) {
                    row[c0] = v[k];
                } else {
                    using V = typename VecOf<T>::type;
                    V t;
                    T* tp = reinterpret_cast<T*>(&t);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) tp[e] = v[k * VEC + e];
                    *reinterpret_cast<V*>(row + c0) = t;
                }
            }
        }
    }
    // sum of squares over the row, skipping the first `skip` columns (0 or 1)
    __device__ __forceinline__ T sumsq(int skip = 0) const {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const T x = v[k * VEC + e];
                    if (k == 0 && e == 0) s += (skip && j == 0) ? T(0) : x * x;
                    else s += x * x;
                }
            }
        }
        return group_sum<G>(s);
    }
    __device__ __forceinline__ T dot(const Tile& o, int skip = 0) const {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const T x = v[k * VEC + e] * o.v[k * VEC + e];
                    if (k == 0 && e == 0) s += (skip && j == 0) ? T(0) : x;
                    else s += x;
                }
            }
        }
        return group_sum<G>(s);
    }
    __device__ __forceinline__ void scale(T f, int skip = 0) {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (k == 0 && e == 0) { if (!(skip && j == 0)) v[0] *= f; }
                    else v[k * VEC + e] *= f;
                }
            }
        }
    }
    // column 0 of the row, broadcast to the group
    __device__ __forceinline__ T first() const { return group_first<G>(v[0]); }
    __device__ __forceinline__ void set_first(T x) {
        if (j == 0) v[0] = x;
    }
    // true if any element of the row is non-zero (NaN counts as non-zero)
    __device__ __forceinline__ bool any_nonzero() const {
        float f = 0.f;
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) f += (v[k * VEC + e] != T(0)) ? 1.f : 0.f;
            }
        }
        return group_sum<G>(f) > 0.f;
    }
    template <class F>
    __device__ __forceinline__ void map(F f) {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k * VEC + e] = f(v[k * VEC + e]);
            }
        }
    }
};

// ---- launch geometry ----------------------------------------------------------
struct RowCfg {
    int g;       // lanes per row: 4, 16 or 64
    int vec;     // 1 or the 16-byte vector width
    bool ok;     // fits the register tile
};

template <typename T>
inline RowCfg pick_cfg(int64_t d, bool aligned16) {
    const int V = VecOf<T>::N;
    RowCfg c;
    c.vec = (aligned16 && d % V == 0) ? V : 1;
    const int64_t lanes = (d + c.vec - 1) / c.vec;       // lanes wanted at one chunk
    const int chmax = kRegs / c.vec;
    if (lanes <= 4) c.g = 4;
    else if (lanes <= 16) c.g = 16;
    else if (lanes <= 32 && chmax >= 2) c.g = 16;          // two chunks, no idle lanes
    else c.g = 64;
    c.ok = lanes <= (int64_t)c.g * chmax;
    return c;
}

inline bool is_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int grid_for_rows(int64_t rows, int g) {
    const int64_t rows_per_block = kBlock / g;
    int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks < 1) blocks = 1;
    const int64_t cap = 256 * 16;       // 256 CUs x 16 resident blocks, grid-stride beyond
    return (int)(blocks < cap ? blocks : cap);
}

}  // namespace hd
