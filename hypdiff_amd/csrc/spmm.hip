// HypAgg as a tangent-space CSR SpMM with the manifold tail fused in.
//
//   support[i] = sum_k val[k] * xt[col[k]]   (k over row i of the CSR, in stored order)
//   out[i]     = epilogue(support[i])
//
// One group of G lanes owns one output row and keeps its accumulator in registers
// (lanes across features, neighbours visited sequentially in CSR order, which is the
// COO order torch.spmm walks -- layers/hyp_layers.py:150).  The rows of a block are
// consecutive, so their (col,val) entries form ONE contiguous CSR segment: the block
// stages that segment into LDS with coalesced loads and every group then reads its
// neighbour list from LDS (same-address broadcast inside a group).  Neighbour rows are
// gathered with 16-byte loads, several in flight per lane.  The epilogue
// (expmap0 -> proj [-> HypAct]) runs on the register row, so the aggregate costs one
// write of (N,d) and no intermediate (N,d) passes.
#include "hd_rowkernels.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

constexpr int kStage = 2048;     // CSR entries staged per block iteration (16 KiB of LDS)

enum Epi { E_NONE = 0, E_AGG = 1, E_AGG_ACT = 2 };

template <class M, typename T, int G, int VEC>
__global__ __launch_bounds__(kBlock) void spmm_csr(const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col,
                                                   const T* __restrict__ val, const T* __restrict__ x,
                                                   T* __restrict__ out, int64_t n, int d, int epi, Curv cin,
                                                   Curv cout, int act) {
    constexpr int RPB = kBlock / G;
    __shared__ int32_t s_col[kStage];
    __shared__ T s_val[kStage];
    Tile<T, G, VEC> acc, nb;
    acc.init(d);
    nb.init(d);
    const int64_t nblk = (n + RPB - 1) / RPB;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = blk * RPB;
        const int64_t r1 = r0 + RPB < n ? r0 + RPB : n;
        const int64_t row = r0 + threadIdx.x / G;
        const bool valid = row < n;
        const int seg0 = rowptr[r0], seg1 = rowptr[r1];
        int rs = 0, re = 0;
        if (valid) { rs = rowptr[row]; re = rowptr[row + 1]; }
        acc.zero();
        for (int base = seg0; base < seg1; base += kStage) {
            const int cnt = seg1 - base < kStage ? seg1 - base : kStage;
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += kBlock) {
                s_col[i] = col[base + i];
                s_val[i] = val[base + i];
            }
            __syncthreads();
            int k0 = rs > base ? rs : base;
            const int k1 = re < base + cnt ? re : base + cnt;
            // neighbours in stored order; four gathers in flight per lane
            for (; k0 + 4 <= k1; k0 += 4) {
                Tile<T, G, VEC> t0, t1, t2, t3;
                t0.init(d); t1.init(d); t2.init(d); t3.init(d);
                const int o = k0 - base;
                t0.load(x + (int64_t)s_col[o] * d);
                t1.load(x + (int64_t)s_col[o + 1] * d);
                t2.load(x + (int64_t)s_col[o + 2] * d);
                t3.load(x + (int64_t)s_col[o + 3] * d);
                const T w0 = s_val[o], w1 = s_val[o + 1], w2 = s_val[o + 2], w3 = s_val[o + 3];
#pragma unroll
                for (int i = 0; i < kRegs; ++i) acc.v[i] += w0 * t0.v[i];
#pragma unroll
                for (int i = 0; i < kRegs; ++i) acc.v[i] += w1 * t1.v[i];
#pragma unroll
                for (int i = 0; i < kRegs; ++i) acc.v[i] += w2 * t2.v[i];
#pragma unroll
                for (int i = 0; i < kRegs; ++i) acc.v[i] += w3 * t3.v[i];
            }
            for (; k0 < k1; ++k0) {
                const int o = k0 - base;
                nb.load(x + (int64_t)s_col[o] * d);
                const T w = s_val[o];
#pragma unroll
                for (int i = 0; i < kRegs; ++i) acc.v[i] += w * nb.v[i];
            }
        }
        if (valid) {
            if (epi >= E_AGG) {                    // layers/hyp_layers.py:151
                M::expmap0(acc, cin);
                M::proj(acc, cin);
            }
            if (epi == E_AGG_ACT) hyp_act<M, T>(acc, cin, cout, act);   // :170-173
            acc.store(out + row * d);
        }
    }
}

template <class M, typename T>
int launch_spmm(const char* name, const int32_t* rowptr, const int32_t* col, const void* val, const void* x,
                void* out, int64_t n, int64_t d, int epi, hd_curv ci, hd_curv co, int act, void* st) {
    HD_REQUIRE(n >= 0 && d >= 1, "%s: bad shape rows=%lld dim=%lld", name, (long long)n, (long long)d);
    if (n == 0) return HD_OK;
    HD_REQUIRE(rowptr && col && val && x && out, "%s: null pointer", name);
    HD_REQUIRE(x != out, "%s: out must not alias x (rows are gathered)", name);
    const RowCfg c = pick_cfg<T>(d, is_aligned16(x) && is_aligned16(out));
    HD_REQUIRE(c.ok, "%s: dim=%lld exceeds the register-tile limit", name, (long long)d);
    const Curv cin{ci.c, ci.sqrt_c, ci.K, ci.sqrt_K}, cout{co.c, co.sqrt_c, co.K, co.sqrt_K};
    return dispatch_cfg<T>(c, [&](auto g, auto v) {
        constexpr int G = decltype(g)::value, V = decltype(v)::value;
        spmm_csr<M, T, G, V><<<grid_for_rows(n, G), kBlock, 0, (hipStream_t)st>>>(
            rowptr, col, (const T*)val, (const T*)x, (T*)out, n, (int)d, epi, cin, cout, act);
        return check_launch(name);
    });
}

}  // namespace

#define HD_SPMM(T, sfx)                                                                                    \
    extern "C" int hd_spmm_csr_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,          \
                                     const void* x, void* out, int64_t n, int64_t d, void* st) {           \
        hd_curv one{1, 1, 1, 1};                                                                           \
        return launch_spmm<PB<T>, T>("hd_spmm_csr_" #sfx, rowptr, col, val, x, out, n, d, E_NONE, one, one, 0, st); \
    }                                                                                                      \
    extern "C" int hd_pb_agg_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,            \
                                   const void* xt, void* out, int64_t n, int64_t d, hd_curv c_in,          \
                                   hd_curv c_out, int act, int fuse_act, void* st) {                       \
        return launch_spmm<PB<T>, T>("hd_pb_agg_" #sfx, rowptr, col, val, xt, out, n, d,                  \
                                     fuse_act ? E_AGG_ACT : E_AGG, c_in, c_out, act, st);                  \
    }                                                                                                      \
    extern "C" int hd_hb_agg_##sfx(const int32_t* rowptr, const int32_t* col, const void* val,            \
                                   const void* xt, void* out, int64_t n, int64_t d, hd_curv c_in,          \
                                   hd_curv c_out, int act, int fuse_act, void* st) {                       \
        return launch_spmm<HB<T>, T>("hd_hb_agg_" #sfx, rowptr, col, val, xt, out, n, d,                  \
                                     fuse_act ? E_AGG_ACT : E_AGG, c_in, c_out, act, st);                  \
    }
HD_SPMM(float, f32)
HD_SPMM(double, f64)
