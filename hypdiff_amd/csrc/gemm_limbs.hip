// fp32-accurate dense contraction on the bf16 matrix cores, operands pre-split into bf16 limbs.
//
//     C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N])
//
// Arithmetic (same as csrc/gemm_split.hip): a = a_h + a_m + a_l exactly up to 2^-26 |a| with
// a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m); a product is the six limb
// products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh), each exact, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The dropped terms are below one fp32 rounding of the product.
//
// What is different here: the split is done ONCE per element by whoever produces the operand
// (hd_limb_pack_f32 for inputs and weights, the write-out of the previous layer's GEMM for
// activations), not once per tile that consumes it, and the operands live in HBM in the exact
// image the kernel wants in LDS, so staging is a linear LDS-DMA copy with no vector-ALU work.
//
// Limb layout ("limb image") of a (rows, cols) fp32 matrix, in 16-byte chunks of 8 bf16:
//     chunk(rb, s, l, kc, r) = (((rb * nslab + s) * 3 + l) * 4 + kc) * 128 + r
// rb = row / 128, r = row % 128, s = col / 32 (K slab), kc = (col % 32) / 8, l = limb (0 h, 1 m,
// 2 l); element e = col % 8 inside the chunk.  nslab = ceil(cols / 32); columns past `cols` in the
// last slab hold zeros, rows past `rows` in the last block are unspecified (never consumed into
// a stored result).  One (rb, s) tile = 1536 chunks = 24 KiB contiguous.
//
// Kernel: 256 threads = 2x2 wavefronts over a 128 x BNT tile (BNT = 64: two workgroups per CU),
// persistent over tiles, K slabs of 32 in two LDS buffers filled by global_load_lds (16 B per
// lane, 1 KiB per wave-instruction, source and destination both linear) two slabs ahead of the
// multiply.  Fragment reads are conflict-free by construction (16 lanes = 16 consecutive rows =
// 256 contiguous bytes).  The MFMA takes W as its first operand and A as its second, so the
// accumulator holds C^T: a lane owns one row m and 4 consecutive columns per accumulator ->
// 16-byte fp32 stores and 8-byte limb stores (a lane pair completes a chunk).
// Replaces x @ W^T of manifolds/poincare.py:96, manifolds/hyperboloid.py:122 and nn.Linear +
// F.gelu of models/direction_diffusion.py:56-69, ddpm.py:9-41.
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "../../include/hypdiff.h"
#include <stdlib.h>

using namespace hd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTile = 12 * 128;            // chunks of one (row block, slab) tile

__device__ __forceinline__ unsigned pk_bf16(float x, float y) {      // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// four fp32 -> three limbs of four bf16 (8 bytes each)
__device__ __forceinline__ void split4(const float4 v, uint2& h, uint2& m, uint2& l) {
    h.x = pk_bf16(v.x, v.y);
    h.y = pk_bf16(v.z, v.w);
    const float r0 = v.x - bf_lo(h.x), r1 = v.y - bf_hi(h.x), r2 = v.z - bf_lo(h.y), r3 = v.w - bf_hi(h.y);
    m.x = pk_bf16(r0, r1);
    m.y = pk_bf16(r2, r3);
    l.x = pk_bf16(r0 - bf_lo(m.x), r1 - bf_hi(m.x));
    l.y = pk_bf16(r2 - bf_lo(m.y), r3 - bf_hi(m.y));
}

__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// fp32 row-major -> limb image.  One thread per (row, 8-column chunk); consecutive threads take
// consecutive rows, so each limb store of a wavefront is 1 KiB contiguous.
__global__ __launch_bounds__(256) void limb_pack(const float* __restrict__ in, int64_t ld, int rows, int cols,
                                                 int nslab, uint4* __restrict__ out, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;      // ((rb * nslab + s) * 4 + kc) * 128 + r
    if (id >= total) return;
    const int r = (int)(id & 127);
    const int64_t q = id >> 7;
    const int kc = (int)(q & 3);
    const int64_t tile = q >> 2;                                     // rb * nslab + s
    const int s = (int)(tile % nslab);
    const int64_t rb = tile / nslab;
    const int64_t row = rb * 128 + r;
    const int c0 = s * 32 + kc * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < rows) {
        const float* p = in + row * ld + c0;
        if (c0 + 8 <= cols && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < cols) v[e] = p[e];
        }
    }
    uint2 h0, m0, l0, h1, m1, l1;
    split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
    split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
    uint4* o = out + tile * kTile + kc * 128 + r;
    o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    o[4 * 128] = make_uint4(m0.x, m0.y, m1.x, m1.y);
    o[8 * 128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// limb image -> fp32 row-major (h + m + l in fp32; tests and debugging)
__global__ __launch_bounds__(256) void limb_unpack(const uint4* __restrict__ in, int rows, int cols, int nslab,
                                                   float* __restrict__ out, int64_t ld, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int r = (int)(id & 127);
    const int64_t q = id >> 7;
    const int kc = (int)(q & 3);
    const int64_t tile = q >> 2;
    const int s = (int)(tile % nslab);
    const int64_t row = (tile / nslab) * 128 + r;
    if (row >= rows) return;
    const uint4* p = in + tile * kTile + kc * 128 + r;
    const uint4 h = p[0], m = p[4 * 128], l = p[8 * 128];
    const unsigned hh[4] = {h.x, h.y, h.z, h.w}, mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
    const int c0 = s * 32 + kc * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = (bf_lo(ll[e]) + bf_lo(mm[e])) + bf_lo(hh[e]);
        const float b = (bf_hi(ll[e]) + bf_hi(mm[e])) + bf_hi(hh[e]);
        if (c0 + 2 * e < cols) out[row * ld + c0 + 2 * e] = a;
        if (c0 + 2 * e + 1 < cols) out[row * ld + c0 + 2 * e + 1] = b;
    }
}

// epi: 0 none, 1 gelu, 2 resid + gelu (ddpm.py:20-22).  C (fp32 row-major, ldc) and CL (limb image
// with nslab_out = ceil(N/32) slabs) are both optional outputs of the same values.
// write-out of 4 consecutive columns nb..nb+3 of row m (tile-local row mloc): bias, epilogue, fp32 and limb stores
__device__ __forceinline__ void store_quad(float v[4], int m, int mloc, int nb, int half, int N, int nout, int ldc,
                                           int epi, const float* __restrict__ bias, const float* __restrict__ resid,
                                           float* __restrict__ C, uint4* __restrict__ CL, size_t cl_tile_base) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = nb + e;
        float x = 0.f;
        if (n < N) {
            x = v[e] + (bias ? bias[n] : 0.f);
            if (epi == 1) x = gelu_erf(x);
            else if (epi == 2) x = resid[(size_t)m * ldc + n] + gelu_erf(x);
        }
        v[e] = x;
    }
    if (C) {
        float* o = C + (size_t)m * ldc + nb;
        if (nb + 3 < N && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nb + e < N) o[e] = v[e];
        }
    }
    if (CL && nb < nout) {
        uint2 h, mm, l;
        split4(make_float4(v[0], v[1], v[2], v[3]), h, mm, l);
        const int so = nb >> 5, kc = (nb & 31) >> 3;
        uint2* o = reinterpret_cast<uint2*>(CL + cl_tile_base + (size_t)so * kTile + kc * 128 + mloc) + half;
        o[0] = h;
        o[2 * 4 * 128] = mm;
        o[2 * 8 * 128] = l;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Persistent kernel: workgroup w takes tiles w, w + grid, w + 2 grid, ...; the first slabs of its
// next tile stream into LDS while the current tile's last slabs are multiplied and written out.
// v_mfma_f32_16x16x32_bf16 (one k-step per 32-deep slab): same cycles and LDS reads as 32x32x16,
// but the chip holds a higher clock on this shape under load (measured +13 % on the denoiser chain).
template <int BNT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_limbs(const uint4* __restrict__ Ag, const uint4* __restrict__ Wg,
                                                  const float* __restrict__ bias, const float* __restrict__ resid,
                                                  float* __restrict__ C, uint4* __restrict__ CL, int M, int N,
                                                  int nslab, int ldc, int epi, int tiles_m, int tiles_n,
                                                  int nslab_out) {
    constexpr int WCN = WAVES / 2;               // wavefronts are laid out 2 (rows) x WCN (columns)
    constexpr int WCOLS = BNT / WCN;             // columns per wavefront: 32 or 64
    constexpr int MT = 4, NT16 = WCOLS / 16;     // 16x16 MFMA tiles per wavefront
    constexpr int QA = 24 / WAVES, QW = (12 * BNT / 64) / WAVES;      // 1 KiB pieces per wavefront and slab
    constexpr int CH_A = 12 * 128, CH_W = 12 * BNT, CH_BUF = CH_A + CH_W;
    __shared__ uint4 lds[2 * CH_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCN, wc = wave % WCN;
    const int ntiles = tiles_m * tiles_n;
    // XCD-aware tile order (see gemm.hip): every XCD walks a contiguous run of tiles, N fastest.
    // The grid is a multiple of 8 (or covers every tile), so a workgroup's tiles stay on its XCD.
    auto decode = [&](int v, int& tm_, int& tn_) {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        tm_ = tile / tiles_n;
        tn_ = tile - tm_ * tiles_n;
    };
    auto a_ptr = [&](int tm_) { return Ag + (size_t)tm_ * nslab * kTile + lane; };
    auto w_ptr = [&](int tn_) {
        return Wg + (size_t)(BNT == 128 ? tn_ : (tn_ >> 1)) * nslab * kTile + (BNT == 64 ? (tn_ & 1) * 64 : 0) + lane;
    };
    // slab t of the tile with sources (as_, ws_) -> LDS buffer b: 1 KiB pieces, QA (A) + QW (W) per wavefront
#define HD_STAGE(as_, ws_, t, b)                                                            \
    {                                                                                       \
        const uint4* as = (as_) + (size_t)(t) * kTile;                                      \
        const uint4* ws = (ws_) + (size_t)(t) * kTile;                                      \
        uint4* la = &lds[(b) * CH_BUF];                                                     \
        _Pragma("unroll") for (int q = 0; q < QA; ++q) {                                    \
            const int piece = q * WAVES + wave;                                             \
            glds16(as + piece * 64, la + piece * 64);                                       \
        }                                                                                   \
        _Pragma("unroll") for (int q = 0; q < QW; ++q) {                                    \
            const int piece = q * WAVES + wave;                                             \
            glds16(ws + piece * (BNT == 128 ? 64 : 128), la + CH_A + piece * 64);           \
        }                                                                                   \
    }
    const int nout = nslab_out * 32;                      // limb image width: zeros from N up to here
    // operand maps of v_mfma_f32_16x16x32_bf16: lane (r16 = lane&15, kq = lane>>4) supplies row r16,
    // k = 8 kq .. 8 kq + 7 of the slab = chunk kc = kq.  First operand = W rows (n), second = A rows
    // (m): the accumulator is the C^T tile, D column = lane&15 = m, D rows 4 kq + reg = n.
    const int r16 = lane & 15, kq = lane >> 4;
    const int a_off = kq * 128 + wr * 64 + r16, w_off = CH_A + kq * BNT + wc * WCOLS + r16;

    int vt = blockIdx.x, tm, tn, tm2 = 0, tn2 = 0;
    decode(vt, tm, tn);
    // a next tile is prefetched across the tile boundary only when every tile has >= 2 slabs
    bool has_next = nslab >= 2 && vt + (int)gridDim.x < ntiles;
    if (has_next) decode(vt + gridDim.x, tm2, tn2);
    const uint4 *a_cur = a_ptr(tm), *w_cur = w_ptr(tn), *a_nxt = a_ptr(tm2), *w_nxt = w_ptr(tn2);
    int pb = 0;                                            // slab t of the current tile lives in buffer (pb + t) & 1

    // Pipeline: a slab's fragments are pulled into registers in one go, so its LDS buffer is free
    // again before the multiply starts and the slab two ahead (of this tile or the next) streams
    // into it meanwhile: two buffers, loads in flight for two slab times.  LDS-DMA completion is
    // tracked with a counted vmcnt -- the youngest group (the slab after the one needed) stays in
    // flight; stores of a write-out issued in between only make the wait stricter -- and raw
    // barriers: __syncthreads would drain vmcnt to 0 (an LDS-DMA is a pending LDS write to the
    // compiler's fence).
    HD_STAGE(a_cur, w_cur, 0, 0)
    if (nslab > 1) HD_STAGE(a_cur, w_cur, 1, 1)
    for (;;) {
        f32x4 acc[MT][NT16];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT16; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int t = 0; t < nslab; ++t) {
            if (t + 1 < nslab || has_next) {
                static_assert(QA + QW == 12 || QA + QW == 9 || QA + QW == 6, "vmcnt literal below");
                if constexpr (QA + QW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (QA + QW == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                       // slab t has landed, for every wave
            bf16x8 fa[MT][3], fw[NT16][3];
            const uint4* la = &lds[((pb + t) & 1) * CH_BUF];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    fa[i][l] = __builtin_bit_cast(bf16x8, la[a_off + l * 4 * 128 + i * 16]);
#pragma unroll
                for (int j = 0; j < NT16; ++j)
                    fw[j][l] = __builtin_bit_cast(bf16x8, la[w_off + l * 4 * BNT + j * 16]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // every wave holds its fragments: buffer free
            if (t + 2 < nslab) HD_STAGE(a_cur, w_cur, t + 2, (pb + t) & 1)
            else if (has_next) HD_STAGE(a_nxt, w_nxt, t + 2 - nslab, (pb + t) & 1)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT16; ++j) {
                    f32x4 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][0], fa[i][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][2], fa[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][1], fa[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][0], fa[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][1], fa[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][0], fa[i][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        // write-out of the C^T tile: a lane owns row m and 4 consecutive columns per accumulator
        {
            const int m0 = tm * 128, n0 = tn * BNT;
            const size_t cl_base = (size_t)tm * nslab_out * kTile;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mloc = wr * 64 + i * 16 + r16;
                if (m0 + mloc >= M) continue;
#pragma unroll
                for (int j = 0; j < NT16; ++j) {
                    const int nb = n0 + wc * WCOLS + j * 16 + 4 * kq;
                    if (nb >= nout && nb >= N) continue;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store_quad(v, m0 + mloc, mloc, nb, kq & 1, N, nout, ldc, epi, bias, resid, C, CL, cl_base);
                }
            }
        }
        if (!has_next) break;
        pb = (pb + nslab) & 1;
        vt += gridDim.x;
        tm = tm2; tn = tn2; a_cur = a_nxt; w_cur = w_nxt;
        has_next = vt + (int)gridDim.x < ntiles;
        if (has_next) {
            decode(vt + gridDim.x, tm2, tn2);
            a_nxt = a_ptr(tm2); w_nxt = w_ptr(tn2);
        }
    }
#undef HD_STAGE
}

}  // namespace

extern "C" int64_t hd_limb_image_bytes(int64_t rows, int64_t cols) {
    if (rows < 0 || cols < 0) return -1;
    return ((rows + 127) / 128) * ((cols + 31) / 32) * (int64_t)kTile * 16;
}

extern "C" int hd_limb_pack_f32(const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* st) {
    HD_REQUIRE(rows >= 0 && cols >= 1 && ld >= cols, "hd_limb_pack_f32: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(in && out && is_aligned16(out), "hd_limb_pack_f32: null or misaligned pointer");
    HD_REQUIRE(rows < (1LL << 31) && cols < (1LL << 31), "hd_limb_pack_f32: size exceeds int32");
    const int nslab = (int)((cols + 31) / 32);
    const int64_t total = ((rows + 127) / 128) * nslab * 4 * 128;
    limb_pack<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st>>>(
        (const float*)in, ld, (int)rows, (int)cols, nslab, (uint4*)out, total);
    return check_launch("hd_limb_pack_f32");
}

extern "C" int hd_limb_unpack_f32(const void* in, int64_t rows, int64_t cols, void* out, int64_t ld, void* st) {
    HD_REQUIRE(rows >= 0 && cols >= 1 && ld >= cols, "hd_limb_unpack_f32: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(in && out, "hd_limb_unpack_f32: null pointer");
    const int nslab = (int)((cols + 31) / 32);
    const int64_t total = ((rows + 127) / 128) * nslab * 4 * 128;
    limb_unpack<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st>>>(
        (const uint4*)in, (int)rows, (int)cols, nslab, (float*)out, ld, total);
    return check_launch("hd_limb_unpack_f32");
}

extern "C" int hd_gemm_limbs_f32(const void* A, const void* W, const void* bias, const void* resid, void* C,
                                 void* CL, int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "hd_gemm_limbs_f32: bad shape M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && (C || CL), "hd_gemm_limbs_f32: null pointer");
    HD_REQUIRE(is_aligned16(A) && is_aligned16(W) && (!CL || is_aligned16(CL)),
               "hd_gemm_limbs_f32: limb images must be 16-byte aligned");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "hd_gemm_limbs_f32: size exceeds int32");
    HD_REQUIRE(epi >= 0 && epi <= 2 && (epi != 2 || resid), "hd_gemm_limbs_f32: bad epilogue / missing residual");
    HD_REQUIRE((!C && epi != 2) || ldc >= N, "hd_gemm_limbs_f32: leading dimension too small");
    const int tiles_m = (int)((M + 127) / 128);
    const int nslab = (int)((K + 31) / 32), nslab_out = (int)((N + 31) / 32);
    // tile: 'N' = 128x64 / 4 wavefronts (two workgroups per CU), 'W' = 128x128 / 4 wavefronts,
    // 'X' = 128x128 / 8 wavefronts (one workgroup per CU).  Persistent grid: resident workgroups
    // only (a multiple of 8, so a workgroup's tiles stay on one XCD), unless K fits one slab.
    char cfg = 'N';
    if (const char* e = getenv("HD_GEMM_LIMBS_TUNE")) cfg = e[0];
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    int persist = 1;
    if (const char* e = getenv("HD_GEMM_LIMBS_PERSIST")) persist = atoi(e);
#define HD_LAUNCH(BNT_, WAVES_, PER_CU_)                                                                   \
    {                                                                                                      \
        const int tiles_n = (int)((N + BNT_ - 1) / BNT_);                                                  \
        const int64_t ntiles = (int64_t)tiles_m * tiles_n;                                                 \
        HD_REQUIRE(ntiles < (1LL << 31), "hd_gemm_limbs_f32: too many tiles");                             \
        int64_t grid = ntiles;                                                                             \
        const int64_t resident = (int64_t)(n_cu * PER_CU_) / 8 * 8;                                        \
        if (persist && nslab >= 2 && resident >= 8 && grid > resident) grid = resident;                    \
        gemm_limbs<BNT_, WAVES_><<<dim3((unsigned)grid), dim3(WAVES_ * 64), 0, (hipStream_t)st>>>(         \
            (const uint4*)A, (const uint4*)W, (const float*)bias, (const float*)resid, (float*)C, (uint4*)CL,  \
            (int)M, (int)N, nslab, (int)ldc, epi, tiles_m, tiles_n, nslab_out);                            \
    }
    if (cfg == 'W') HD_LAUNCH(128, 4, 1)
    else if (cfg == 'X') HD_LAUNCH(128, 8, 1)
    else HD_LAUNCH(64, 4, 2)
#undef HD_LAUNCH
    return check_launch("hd_gemm_limbs_f32");
}
