// HypAgg as a tangent-space CSR SpMM with the manifold tail fused in.
//
//   support[i] = sum_k val[k] * xt[col[k]]   (k over row i of the CSR, in stored order)
//   out[i]     = epilogue(support[i])
//
// spmm_rows: one group of G lanes owns one output row and keeps its accumulator in
// registers (lanes across features, neighbours visited in CSR order, which is the COO
// order torch.spmm walks -- layers/hyp_layers.py:150).  The rows of a block are
// consecutive, so their (col,val) entries form ONE contiguous CSR segment: the block
// stages that segment into LDS with coalesced loads and every group reads its neighbour
// list from LDS (same-address broadcast inside a group).  Neighbour rows are gathered
// with 16-byte loads, eight in flight per lane (predicated at the row's end).
//
// spmm_long: power-law hubs.  A row with more than kLongRow neighbours would keep one
// wavefront busy for milliseconds, so such rows are skipped by spmm_rows and handled by
// one 512-thread workgroup each: its 8 wavefronts take 8 contiguous slices of the
// neighbour list, and the 8 partial rows are added in slice order through LDS (fixed
// shape -> bit-reproducible; differs from the strictly sequential sum by rounding only).
//
// The epilogue (expmap0 -> proj [-> HypAct]) runs on the register row, so the aggregate
// costs one write of (N,d) and no intermediate (N,d) passes.
#pragma once
#include "hd_rowkernels.cuh"
#include <stdlib.h>
#include "../../include/hypdiff.h"

namespace hd {

constexpr int kStage = 2048;     // CSR entries staged per block iteration (16 KiB of LDS)
constexpr int kLongRow = 512;    // rows with more neighbours go to spmm_long
constexpr int kDepth = 8;        // neighbour rows in flight per lane, at most
// ... such that a lane keeps 128 bytes of gathered rows in flight: wider register tiles take fewer
// rows (8 x 4 regs, 4 x 8 regs, 2 x 16 regs), which keeps the kernel at 3-4 wavefronts per SIMD
template <typename T, int REGS>
constexpr int depth_for() {
    constexpr int d = 128 / (REGS * (int)sizeof(T));
    return d > kDepth ? kDepth : (d < 2 ? 2 : d);
}

enum Epi { E_NONE = 0, E_AGG = 1, E_AGG_ACT = 2 };

template <class M, typename T, class TL>
__device__ __forceinline__ void agg_epilogue(TL& acc, int epi, const Curv& cin, const Curv& cout, int act) {
    if (epi >= E_AGG) {                    // layers/hyp_layers.py:151
        M::expmap0(acc, cin);
        M::proj(acc, cin);
    }
    if (epi == E_AGG_ACT) hyp_act<M, T>(acc, cin, cout, act);   // :170-173
}

// lane k's value of v, k wavefront-uniform
__device__ __forceinline__ float wave_get(float v, int k) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
__device__ __forceinline__ double wave_get(double v, int k) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// acc += sum over entries [k0, k1) read through (cols, vals) with index offset `off`
template <typename T, int DEPTH, class TL, class CI, class VI>
__device__ __forceinline__ void gather_accumulate(TL& acc, const T* __restrict__ x, int d, int64_t ldx, int k0, int k1,
                                                  const CI& cols, const VI& vals) {
    for (; k0 < k1; k0 += DEPTH) {
        TL t[DEPTH];
        T w[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            t[u].init(d);
            if (k0 + u < k1) {
                t[u].load(x + (int64_t)cols(k0 + u) * ldx);
                w[u] = vals(k0 + u);
            } else {
                t[u].zero();
                w[u] = T(0);
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int i = 0; i < TL::kRegs; ++i) acc.v[i] += w[u] * t[u].v[i];
    }
}

template <class M, typename T, int G, int VEC, int REGS>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 ? 4 : 1)) void spmm_rows(const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col,
                                                    const T* __restrict__ val, const T* __restrict__ x,
                                                    T* __restrict__ out, int64_t n, int d, int64_t ldx, int64_t ldo,
                                                    int epi, Curv cin, Curv cout, int act, int skip_long) {
    constexpr int RPB = kBlock / G;
    constexpr int DEPTH = depth_for<T, REGS>();
    __shared__ int32_t s_col[kStage];
    __shared__ T s_val[kStage];
    using TL = Tile<T, G, VEC, REGS>;
    TL acc;
    acc.init(d);
    const int64_t nblk = (n + RPB - 1) / RPB;
    if constexpr (G == 64) {
        // one wavefront per row: its neighbour list goes through the wavefront's own registers (lane k holds entry
        // k of a 64-entry window, handed out with v_readlane) -- no LDS stage, no workgroup barrier, so the four
        // wavefronts of a block run their rows independently (the barrier pair cost the short rows of the graph-set
        // workloads more than the gather itself)
        const int lane = threadIdx.x & 63;
        if (sizeof(T) == 4 && epi == E_AGG_ACT && (skip_long & 2) &&
            (act == ACT_IDENTITY || act == ACT_RELU || act == ACT_LEAKY_RELU)) {
            // fused aggregate + HypAct, four rows per wavefront: the tail's scalar chain (sinh / cosh / arcosh or
            // tanh / artanh, ~150 quarter-rate issues) is wavefront-uniform in the one-row form, i.e. computed 64
            // times over; here lane r evaluates it for row r from the two row sums and hands the factor back
            constexpr int R = 4;
            const T slope = act == ACT_IDENTITY ? T(1) : (act == ACT_RELU ? T(0) : T(0.01));
            for (int64_t row0 = ((int64_t)blockIdx.x * RPB + (threadIdx.x >> 6)) * R; row0 < n;
                 row0 += (int64_t)gridDim.x * RPB * R) {
                TL a4[R];
                bool ok4[R];
                T s_l = T(0), sr_l = T(0);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    a4[r].init(d);
                    a4[r].zero();
                    const int64_t row = row0 + r;
                    ok4[r] = row < n;
                    if (ok4[r]) {
                        const int rs = rowptr[row], re = rowptr[row + 1];
                        if ((skip_long & 1) && re - rs > (skip_long >> 8)) ok4[r] = false;
                        else
                            for (int base = rs; base < re; base += 64) {
                                const int cnt = re - base < 64 ? re - base : 64;
                                const int32_t c_l = lane < cnt ? col[base + lane] : 0;
                                const T v_l = lane < cnt ? val[base + lane] : T(0);
                                gather_accumulate<T, DEPTH>(a4[r], x, d, ldx, 0, cnt,
                                                            [&](int k) { return __builtin_amdgcn_readlane(c_l, k); },
                                                            [&](int k) { return wave_get(v_l, k); });
                            }
                    }
                    T s1, s2;
                    a4[r].sumsq_pair(IsHB<M>::value ? 1 : 0, slope, s1, s2);
                    if (lane == r) { s_l = s1; sr_l = s2; }
                }
                T f_l, t0_l;
                agg_act_scalars<T>((M*)nullptr, s_l, sr_l, cin, cout, f_l, t0_l);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T f = wave_get(f_l, r);
                    a4[r].map([=](T v) { return f * (v > T(0) ? v : (v != v ? v : slope * v)); });
                    if (IsHB<M>::value) a4[r].set_first(wave_get(t0_l, r));
                    if (ok4[r]) a4[r].store(out + (row0 + r) * ldo);
                }
            }
            return;
        }
        for (int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * RPB) {
            const int rs = rowptr[row], re = rowptr[row + 1];
            if ((skip_long & 1) && re - rs > (skip_long >> 8)) continue;
            acc.zero();
            for (int base = rs; base < re; base += 64) {
                const int cnt = re - base < 64 ? re - base : 64;
                const int32_t c_l = lane < cnt ? col[base + lane] : 0;
                const T v_l = lane < cnt ? val[base + lane] : T(0);
                gather_accumulate<T, DEPTH>(acc, x, d, ldx, 0, cnt, [&](int k) { return __builtin_amdgcn_readlane(c_l, k); },
                                            [&](int k) { return wave_get(v_l, k); });
            }
            agg_epilogue<M, T>(acc, epi, cin, cout, act);
            acc.store(out + row * ldo);
        }
        return;
    }
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = blk * RPB;
        const int64_t r1 = r0 + RPB < n ? r0 + RPB : n;
        const int64_t row = r0 + threadIdx.x / G;
        bool valid = row < n;
        int rs = 0, re = 0;
        if (valid) { rs = rowptr[row]; re = rowptr[row + 1]; }
        if ((skip_long & 1) && re - rs > (skip_long >> 8)) { valid = false; re = rs; }
        const int seg0 = rowptr[r0], seg1 = rowptr[r1];
        acc.zero();
        if (seg1 - seg0 <= kStage) {
            // common case: the whole segment of the block's rows fits one LDS stage
            __syncthreads();
            for (int i = threadIdx.x; i < seg1 - seg0; i += kBlock) {
                s_col[i] = col[seg0 + i];
                s_val[i] = val[seg0 + i];
            }
            __syncthreads();
            gather_accumulate<T, DEPTH>(acc, x, d, ldx, rs, re, [&](int k) { return s_col[k - seg0]; },
                                 [&](int k) { return s_val[k - seg0]; });
        } else {
            // a long (unsplit) row in this block: read the lists straight from global memory
            gather_accumulate<T, DEPTH>(acc, x, d, ldx, rs, re, [&](int k) { return col[k]; }, [&](int k) { return val[k]; });
        }
        if (valid) {
            agg_epilogue<M, T>(acc, epi, cin, cout, act);
            acc.store(out + row * ldo);
        }
    }
}

constexpr int kLongWaves = 8;

// one workgroup of kLongWaves wavefronts per long row (G = 64 lanes across features)
template <class M, typename T, int VEC, int REGS>
__global__ __launch_bounds__(kLongWaves * 64) void spmm_long(const int32_t* __restrict__ long_rows,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const T* __restrict__ val, const T* __restrict__ x,
                                                             T* __restrict__ out, int d, int64_t ldx, int64_t ldo,
                                                             int epi, Curv cin, Curv cout, int act) {
    using TL = Tile<T, 64, VEC, REGS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* part = reinterpret_cast<T*>(smem);                 // [kLongWaves][d]
    const int row = long_rows[blockIdx.x];
    const int rs = rowptr[row], re = rowptr[row + 1];
    const int wave = threadIdx.x >> 6;
    const int len = re - rs, per = (len + kLongWaves - 1) / kLongWaves;
    int k0 = rs + wave * per, k1 = k0 + per;
    k0 = k0 < re ? k0 : re;
    k1 = k1 < re ? k1 : re;
    TL acc;
    acc.init(d);
    acc.zero();
    gather_accumulate<T, depth_for<T, REGS>()>(acc, x, d, ldx, k0, k1, [&](int k) { return col[k]; }, [&](int k) { return val[k]; });
    acc.store(part + (size_t)wave * d);
    __syncthreads();
    if (wave == 0) {
        TL tmp;
        tmp.init(d);
        for (int w = 1; w < kLongWaves; ++w) {
            tmp.load(part + (size_t)w * d);
#pragma unroll
            for (int i = 0; i < TL::kRegs; ++i) acc.v[i] += tmp.v[i];
        }
        agg_epilogue<M, T>(acc, epi, cin, cout, act);
        acc.store(out + (int64_t)row * ldo);
    }
}

template <class M, typename T>
int launch_spmm(const char* name, const int32_t* rowptr, const int32_t* col, const void* val, const void* x,
                void* out, int64_t n, int64_t d, int epi, hd_curv ci, hd_curv co, int act,
                const int32_t* long_rows, int64_t n_long, void* st, int64_t ldx = 0, int64_t ldo = 0, int form = 0,
                int long_thr = 0) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (ldx == 0) ldx = d;                              // row pitches (elements); 0 = contiguous rows
    if (ldo == 0) ldo = d;
    HD_REQUIRE(ldx >= d && ldo >= d, "%s: row pitch smaller than dim", name);
    if (n == 0) return HD_OK;
    HD_REQUIRE(rowptr && col && val && x && out, "%s: null pointer", name);
    HD_REQUIRE(x != out, "%s: out must not alias x (rows are gathered)", name);
    HD_REQUIRE(n_long >= 0 && (n_long == 0 || long_rows), "%s: long-row list missing", name);
    const bool al16 = is_aligned16(x) && is_aligned16(out) && ldx % VecOf<T>::N == 0 && ldo % VecOf<T>::N == 0;
    RowCfg c = pick_cfg<T>(d, al16);
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    const Curv cin{ci.c, ci.sqrt_c, ci.K, ci.sqrt_K}, cout{co.c, co.sqrt_c, co.K, co.sqrt_K};
    const int ch = (int)((d + (int64_t)c.g * c.vec - 1) / ((int64_t)c.g * c.vec));
    const int need = ch * c.vec;                       // registers the row needs per lane
    hipStream_t s = (hipStream_t)st;
    // bit 0: rows longer than kLongRow go to spmm_long; bit 1: four rows per wavefront with the fused analytic tail
    static const int rows4 = [] { const char* e = getenv("HD_SPMM_ROWS4"); return e ? atoi(e) : 1; }();
    // (on the hyperboloid the four-row form measured slower on hub-heavy graphs -- 2.79 -> 3.29 ms on the power-law
    // graph at 10^6 x 256, cause not found; faster on the graph sets: 31 -> 26 us -- so there it is taken only for
    // adjacencies without long rows; on the ball it won on both, 2.72 -> 2.69 ms and 29.5 -> 20.5 us)
    // (and only when there are rows enough to keep every SIMD busy with four-row tasks: a 2708-row graph ran its
    // aggregate 45 us in the four-row form against 20 us one row per wavefront -- latency-bound, not chain-bound)
    // The two forms of the fused tail differ by rounding, so WHICH one runs is a property of the adjacency, not of the
    // launch: `form` 1 (one row per wavefront) / 2 (four rows) pins it -- a row shard passes the form the unsharded
    // graph takes (hypdiff_amd/dist.py) -- and 0 decides from this launch's own rows.
    HD_REQUIRE(form >= 0 && form <= 2, "%s: form must be 0 (auto), 1 or 2", name);
    const bool four = form == 2 || (form == 0 && rows4 && n >= 8192 && (!IsHB<M>::value || n_long == 0));
    // long_thr: rows with MORE entries are in `long_rows` and taken by spmm_long (0 = kLongRow).  A small graph passes a
    // lower threshold (hypdiff_amd/ops.py): there one wavefront walking a row of a few hundred neighbours is the whole
    // launch's critical path (config 4: 42 us for 2708 rows), and eight wavefronts with the fixed-shape sum are not.
    if (long_thr <= 0) long_thr = kLongRow;
    HD_REQUIRE(long_thr >= 8 && long_thr < (1 << 20), "%s: long-row threshold out of range", name);
    const int row_flags = (n_long > 0 ? 1 : 0) | (four ? 2 : 0) | (long_thr << 8);
    int rc = dispatch_cfg<T>(c, [&](auto g, auto v) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value;
        constexpr int R0 = V < 4 ? 4 : V;              // smallest specialisation
        if (sizeof(T) == 4 && need <= R0)
            spmm_rows<M, T, G, V, R0><<<grid_for_rows(n, G), kBlock, 0, s>>>(
                rowptr, col, (const T*)val, (const T*)x, (T*)out, n, (int)d, ldx, ldo, epi, cin, cout, act, row_flags);
        else if (sizeof(T) == 4 && need <= 8)
            spmm_rows<M, T, G, V, 8><<<grid_for_rows(n, G), kBlock, 0, s>>>(
                rowptr, col, (const T*)val, (const T*)x, (T*)out, n, (int)d, ldx, ldo, epi, cin, cout, act, row_flags);
        else
            spmm_rows<M, T, G, V, kRegs><<<grid_for_rows(n, G), kBlock, 0, s>>>(
                rowptr, col, (const T*)val, (const T*)x, (T*)out, n, (int)d, ldx, ldo, epi, cin, cout, act, row_flags);
        return check_launch(name);
    });
    if (rc || n_long == 0) return rc;
    // long rows: G = 64 always; scalar loads unless 16-byte rows
    const int V = VecOf<T>::N;
    const bool vec = al16 && d % V == 0;
    const int64_t lanes = vec ? d / V : d;
    HD_REQUIRE(lanes <= 64LL * (kRegs / (vec ? V : 1)), "%s: dim too large for the long-row kernel", name);
    const size_t lds = (size_t)kLongWaves * d * sizeof(T);
    HD_REQUIRE(lds <= 64 * 1024, "%s: dim too large for the long-row LDS reduce", name);
    const int need64 = (int)((lanes + 63) / 64) * (vec ? V : 1);
#define HD_LONG(VV, RR)                                                                                   \
    spmm_long<M, T, VV, RR><<<(unsigned)n_long, kLongWaves * 64, lds, s>>>(                               \
        long_rows, rowptr, col, (const T*)val, (const T*)x, (T*)out, (int)d, ldx, ldo, epi, cin, cout, act)
    if (vec) {
        if (sizeof(T) == 4 && need64 <= 4) HD_LONG(VecOf<T>::N, 4);
        else if (sizeof(T) == 4 && need64 <= 8) HD_LONG(VecOf<T>::N, 8);
        else HD_LONG(VecOf<T>::N, kRegs);
    } else {
        if (sizeof(T) == 4 && need64 <= 4) HD_LONG(1, 4);
        else HD_LONG(1, kRegs);
    }
#undef HD_LONG
    return check_launch(name);
}

}  // namespace hd

#define HD_SPMM(T, sfx)                                                                                    \
    extern "C" int hd_spmm_csr_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,          \
                                     const void* x, void* out, int64_t n, int64_t d,                       \
                                     const int32_t* long_rows, int64_t n_long, void* st) {                 \
        hd_curv one{1, 1, 1, 1};                                                                           \
        return hd::launch_spmm<hd::PB<T>, T>("hd_spmm_csr_" #sfx, rowptr, col, val, x, out, n, d, hd::E_NONE, one, \
                                             one, 0, long_rows, n_long, st);                               \
    }                                                                                                      \
    extern "C" int hd_spmm_csr_strided_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,  \
                                             const void* x, int64_t ldx, void* out, int64_t ldo, int64_t n,  \
                                             int64_t d, const int32_t* long_rows, int64_t n_long,          \
                                             int64_t long_threshold, void* st) {                           \
        hd_curv one{1, 1, 1, 1};                                                                           \
        HD_REQUIRE(long_threshold >= 0 && long_threshold < (1 << 20), "hd_spmm_csr_strided: bad long-row threshold"); \
        return hd::launch_spmm<hd::PB<T>, T>("hd_spmm_csr_strided_" #sfx, rowptr, col, val, x, out, n, d, hd::E_NONE, \
                                             one, one, 0, long_rows, n_long, st, ldx, ldo, 0, (int)long_threshold); \
    }                                                                                                      \
    extern "C" int hd_pb_agg_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,            \
                                   const void* xt, void* out, int64_t n, int64_t d, hd_curv c_in,          \
                                   hd_curv c_out, int act, int fuse_act, const int32_t* long_rows,         \
                                   int64_t n_long, void* st) {                                             \
        return hd::launch_spmm<hd::PB<T>, T>("hd_pb_agg_" #sfx, rowptr, col, val, xt, out, n, d,          \
                                             (fuse_act & 1) ? hd::E_AGG_ACT : hd::E_AGG, c_in, c_out, act, \
                                             long_rows, n_long, st, 0, 0, (fuse_act >> 8) & 3, (fuse_act >> 10) & 0xfffff); \
    }                                                                                                      \
    extern "C" int hd_hb_agg_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,            \
                                   const void* xt, void* out, int64_t n, int64_t d, hd_curv c_in,          \
                                   hd_curv c_out, int act, int fuse_act, const int32_t* long_rows,         \
                                   int64_t n_long, void* st) {                                             \
        return hd::launch_spmm<hd::HB<T>, T>("hd_hb_agg_" #sfx, rowptr, col, val, xt, out, n, d,          \
                                             (fuse_act & 1) ? hd::E_AGG_ACT : hd::E_AGG, c_in, c_out, act, \
                                             long_rows, n_long, st, 0, 0, (fuse_act >> 8) & 3, (fuse_act >> 10) & 0xfffff); \
    }
