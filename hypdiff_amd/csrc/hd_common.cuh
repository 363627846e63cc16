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

// ---- cross-lane sums on the DPP data path (no LDS crossbar, ~1 VALU issue per step) ----
// dpp_ctrl encodings: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141,
// row_mirror=0x140, row_bcast:15=0x142, row_bcast:31=0x143 (a DPP row = 16 lanes).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_get(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_bcast0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
}
__device__ __forceinline__ double lane_bcast0(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 0);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 0);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_bcast63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double lane_bcast63(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// sum over an aligned group of G lanes (G = 4, 16 or 64), result in every lane of the group
template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
    static_assert(G == 4 || G == 16 || G == 64, "lane groups are 4, 16 or 64 wide");
    v += dpp_get<0xB1>(v);                  // lane ^ 1
    v += dpp_get<0x4E>(v);                  // lane ^ 2   -> quad sums
    if constexpr (G >= 16) {
        v += dpp_get<0x141>(v);             // mirror inside 8  -> sums of 8
        v += dpp_get<0x140>(v);             // mirror inside 16 -> row sums
    }
    if constexpr (G == 64) {
        v += dpp_get<0x142, 0xA>(v);        // rows 1,3 += lane 15 of rows 0,2
        v += dpp_get<0x143, 0xC>(v);        // rows 2,3 += lane 31 -> lane 63 holds the total
        v = lane_bcast63(v);
    }
    return v;
}

template <typename T, int G, int VEC, int REGS = kRegs>
struct Tile {
    static constexpr int CHMAX = REGS / VEC;
    static constexpr int kRegs = REGS;      // shadows hd::kRegs inside the tile
    T v[REGS];
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
    // load a row kept in memory as a NARROWER type (fp32 rows into an fp64 tile): the same values as a cast pass
    // followed by load(), without the (N, d) intermediate
    template <typename S>
    __device__ __forceinline__ void load_cvt(const S* __restrict__ row) {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            const int c0 = (k * G + j) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[k * VEC + e] = T(0);
            if (k < ch && c0 < d) {
                if constexpr (VEC == 2 && sizeof(S) == 4) {
                    const float2 t = *reinterpret_cast<const float2*>(row + c0);
                    v[k * VEC] = (T)t.x;
                    v[k * VEC + 1] = (T)t.y;
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[k * VEC + e] = (T)row[c0 + e];
                }
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
    // column 0 of the row, broadcast to the group (lane 0 of the group holds it)
    __device__ __forceinline__ T first() const {
        if constexpr (G == 64) return lane_bcast0(v[0]);
        else return __shfl(v[0], (int)(threadIdx.x & 63 & ~(G - 1)), 64);
    }
    // sum of squares of the row (s1) and of act(row) (s2) for a positively homogeneous
    // activation (identity: slope 1, relu: 0, leaky_relu: 0.01 on the negative side), one
    // sweep over the registers and two interleaved butterflies; `skip` drops column 0
    __device__ __forceinline__ void sumsq_pair(int skip, T neg_slope, T& s1, T& s2) const {
        T a = T(0), b = T(0);
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    T x = v[k * VEC + e];
                    if (k == 0 && e == 0) x = (skip && j == 0) ? T(0) : x;
                    const T y = x > T(0) ? x : (x != x ? x : neg_slope * x);
                    a += x * x;
                    b += y * y;
                }
            }
        }
        s1 = group_sum<G>(a);
        s2 = group_sum<G>(b);
    }
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
    // plain sum over the row (pad slots hold zeros)
    __device__ __forceinline__ T sum() const {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[k * VEC + e];
            }
        }
        return group_sum<G>(s);
    }
    // v <- f(v, column) on the columns that exist; pad slots (column >= d) are reset to zero
    template <class F>
    __device__ __forceinline__ void map_cols(F f) {
#pragma unroll
        for (int k = 0; k < CHMAX; ++k) {
            if (k < ch) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int c = col(k, e);
                    v[k * VEC + e] = c < d ? f(v[k * VEC + e], c) : T(0);
                }
            }
        }
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
