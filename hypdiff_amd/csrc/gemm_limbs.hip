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
//
// Second format, "f16x2" (F16 = true): two fp16 limbs per element (22 significant bits) of the row
// scaled by a power of two, a * 2^e[row] = a_h + a_l, three limb products (hh, hl, lh) per fp32
// product on v_mfma_f32_16x16x32_f16 -- half the matrix-core work of the bf16 format.  fp16 has
// 5 exponent bits, so every row carries its own exponent e[row] (removed again, exactly, by
// v_ldexp_f32 on the accumulator).  For packed inputs and weights e comes from the exact row
// maximum.  For activations written by a GEMM's epilogue the row maximum of the OUTPUT is not
// known to a single tile, so e is taken from a bound that is: |c_mn| <= sqrt(K) max|a_m| max_n|w_n|_2
// + max|b| (Cauchy-Schwarz; max|a_m| is exact -- packed rows carry it, epilogues collect it with
// an integer atomicMax, deterministic), scaled to [2^13, 2^14).  fp16 keeps an absolute grid of
// 2^-24 below 2^-14 (subnormals are not flushed by the matrix cores), so a row whose true maximum
// is 2^-k of the bound is still represented to 2^-(38-k) of its maximum: the bound may be loose
// by 2^16 before the format loses anything against the 2^-22 of two limbs.  Error model: per
// product <= 3 * 2^-22 |a||w| worst case (representation 2 * 2^-22, dropped ll 2^-22), absolute
// 2^-38 * bound per element for rows far below their bound.
// Replaces x @ W^T of manifolds/poincare.py:96, manifolds/hyperboloid.py:122 and nn.Linear +
// F.gelu of models/direction_diffusion.py:56-69, ddpm.py:9-41.
#include "hd_common.cuh"
#include <type_traits>
#include "hd_gelu.cuh"
#include "hd_manifold.cuh"
#include "../../include/hypdiff.h"
#include <stdlib.h>

using namespace hd;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// limbs per element and chunks of one (row block, slab) tile: 24 KiB (bf16x3) / 16 KiB (f16x2)
template <bool F16> constexpr int kLimbs = F16 ? 2 : 3;
template <bool F16> constexpr int kTileOf = 4 * kLimbs<F16> * 128;

__device__ __forceinline__ unsigned pk_bf16(float x, float y) {      // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_f16(float x, float y) {       // 2 x v_cvt_f16_f32 (RNE)
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float hf_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float hf_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

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
// four (already scaled) fp32 -> two limbs of four fp16; v - h is exact in fp32
__device__ __forceinline__ void split4_f16(const float4 v, uint2& h, uint2& l) {
    h.x = pk_f16(v.x, v.y);
    h.y = pk_f16(v.z, v.w);
    l.x = pk_f16(v.x - hf_lo(h.x), v.y - hf_hi(h.x));
    l.y = pk_f16(v.z - hf_lo(h.y), v.w - hf_hi(h.y));
}

// row exponent e with bound * 2^e in [2^(top-1), 2^top); 0 for a zero / NaN bound; a bound that
// overflowed fp32 is taken as 2^128, so every finite value of the row still fits fp16
__device__ __forceinline__ int exp_for(float bound, int top) {
    if (!(bound > 0.f)) return 0;
    if (bound > 3.0e38f) return top - 128;
    const int e = top - __builtin_amdgcn_frexp_expf(bound);         // bound < 2^frexp_exp
    return e < -120 ? -120 : (e > 120 ? 120 : e);
}

__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// exact row maximum of |x| (bit pattern) and the row exponent that scales it to [2^14, 2^15)
__global__ __launch_bounds__(256) void row_absmax(const float* __restrict__ in, int64_t ld, int rows, int cols,
                                                  unsigned* __restrict__ amax, int* __restrict__ ex) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = in + (int64_t)row * ld;
    float m = 0.f;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, fabsf(p[c]));
#pragma unroll
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) {
        amax[row] = __float_as_uint(m);
        ex[row] = exp_for(m, 15);
    }
}

// fp32 row-major -> limb image.  One thread per (row, 8-column chunk); consecutive threads take
// consecutive rows, so each limb store of a wavefront is 1 KiB contiguous.
template <bool F16>
__global__ __launch_bounds__(256) void limb_pack(const float* __restrict__ in, int64_t ld, int rows, int cols,
                                                 int nslab, uint4* __restrict__ out, int64_t total,
                                                 const int* __restrict__ ex) {
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
    uint4* o = out + tile * kTileOf<F16> + kc * 128 + r;
    if constexpr (F16) {
        const int e_row = row < rows ? ex[row] : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ldexpf(v[e], e_row);
        uint2 h0, l0, h1, l1;
        split4_f16(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
        split4_f16(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        o[4 * 128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
        uint2 h0, m0, l0, h1, m1, l1;
        split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
        split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        o[4 * 128] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        o[8 * 128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// Several bf16x3 images in ONE launch (the weights of a training step: W_l for the forward pass and W_l^T for the
// input gradient, 9 small matrices for the mlp denoiser, each a launch of a few microseconds otherwise).  A job's
// logical matrix is src or src^T; threads of a wavefront take consecutive logical rows, so the transposed read is
// contiguous across lanes and the plain one is 32 bytes per lane.
constexpr int kMaxPackJobs = 16;
struct PackJob { const float* src; uint4* out; int64_t ld; int rows, cols, nslab, trans; unsigned block0; };
struct PackJobs { int n; PackJob j[kMaxPackJobs]; };

// One workgroup per (128-row block, 32-column slab) of an image: the 128 x 32 fp32 values go through LDS so that both
// sides are in runs -- the source is read in 128-byte row pieces (or, transposed, in 512-byte column pieces), the image is
// written in 2-KiB runs of one (limb, chunk column).  (Round 3: a thread used to fetch its own 32 bytes of its own row,
// 33 us for the ten weight images of the denoiser; same bits.)
__global__ __launch_bounds__(256) void limb_pack_multi(const PackJobs jobs) {
    __shared__ float st[128][33];
    int k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.j[k + 1].block0) ++k;
    const PackJob& J = jobs.j[k];
    const int64_t tile = (int64_t)(blockIdx.x - J.block0);                   // rb * nslab + s
    const int s = (int)(tile % J.nslab);
    const int64_t rb = tile / J.nslab;
    const int t = threadIdx.x;
    if (J.trans) {                                                            // image (r, c) = src[c][r]
        const int r = t & 127;
        const int64_t row = rb * 128 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = (t >> 7) + 2 * i, cg = s * 32 + c;
            st[r][c] = (row < J.rows && cg < J.cols) ? J.src[(int64_t)cg * J.ld + row] : 0.f;
        }
    } else {
        const int c = (t & 7) * 4, cg = s * 32 + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (t >> 3) + 32 * i;
            const int64_t row = rb * 128 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < J.rows) {
                const float* p = J.src + row * J.ld + cg;
                if (cg + 4 <= J.cols && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (cg < J.cols) v.x = p[0];
                    if (cg + 1 < J.cols) v.y = p[1];
                    if (cg + 2 < J.cols) v.z = p[2];
                    if (cg + 3 < J.cols) v.w = p[3];
                }
            }
            st[r][c] = v.x; st[r][c + 1] = v.y; st[r][c + 2] = v.z; st[r][c + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int id = t + 256 * h, r = id & 127, kc = id >> 7;
        const float* v = &st[r][kc * 8];
        uint4* o = J.out + tile * kTileOf<false> + kc * 128 + r;
        uint2 h0, m0, l0, h1, m1, l1;
        split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
        split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        o[4 * 128] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        o[8 * 128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// Row producers that write the f16x2 image of a GEMM operand directly, in one pass over the rows: a
// wavefront takes a row at a time, a lane owns the 8-column chunks lane, lane + 64, ...; the row maximum (and the
// row norm, where the map needs one) is a wavefront reduction, so the row is read once and no fp32
// copy of the operand is made.
//   PACK_MLP     cat(x.float(), t[:,None])            models/direction_diffusion.py:57-58
//   PACK_HB_LOG0 Hyperboloid.logmap0(x)               manifolds/hyperboloid.py:103-113 (operand of :122)
//   PACK_NORM    x itself; also writes |x|_2 per row  manifolds/poincare.py:94-96
enum PackMode { PACK_MLP = 0, PACK_HB_LOG0 = 1, PACK_NORM = 2 };
constexpr int kPackChunks = 3;                                    // chunks per lane at most: cols <= 1536
// NQ = chunks of 8 columns per lane = ceil(columns / 512): rows of <= 512 columns keep 32 row registers per lane
// instead of 96 (the three-chunk form ran at two workgroups' worth of registers for every width)
template <int MODE, typename TX, bool F16, int NQ>
__global__ __launch_bounds__(256) void row_pack_f16x2(const TX* __restrict__ x, const void* __restrict__ t, int t_kind,
                                                      float t_scalar, int rows, int dim, int nslab, Curv cv,
                                                      uint4* __restrict__ out, int* __restrict__ ex,
                                                      unsigned* __restrict__ amax, float* __restrict__ norms) {
    // A workgroup owns 16 consecutive rows, 4 per wavefront.  The limbs go to LDS first ([limb][chunk][row],
    // 17-chunk pitch: conflict-free both ways) and leave as 256-byte runs -- in the image the 16 rows of one
    // (limb, chunk) are contiguous, whereas a row's own chunks lie 2 KiB apart.  A wavefront loads its 4 rows
    // together and evaluates logmap0's scalar chain (fp32 arcosh: wavefront-uniform in a row-at-a-time
    // kernel, i.e. computed 64 times over) once for all of them, row i in lane i.
    extern __shared__ uint4 stage[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = nslab * 4;
    constexpr int RW = 4;
    constexpr int V = 16 / sizeof(TX);
    using VT = typename VecOf<TX>::type;
    float v[RW][NQ][8];
    float ss_l = 0.f, x0_l = 0.f;
#pragma unroll
    for (int it = 0; it < RW; ++it) {
        const int row_raw = blockIdx.x * 16 + wave * RW + it;
        const int row = row_raw < rows ? row_raw : rows - 1;
        const TX* xr = x + (int64_t)row * dim;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(xr) & 15) == 0;
        float tv = 0.f;
        if constexpr (MODE == PACK_MLP)
            tv = t_kind == 0 ? (float)((const int64_t*)t)[row] : (t_kind == 1 ? ((const float*)t)[row] : t_scalar);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c0 = (lane + 64 * q) * 8;
            if (c0 + 8 <= dim && vec_ok) {
#pragma unroll
                for (int g = 0; g < 8 / V; ++g) {
                    const VT pk = *reinterpret_cast<const VT*>(xr + c0 + g * V);
                    const TX* pe = reinterpret_cast<const TX*>(&pk);
#pragma unroll
                    for (int e = 0; e < V; ++e) v[it][q][g * V + e] = (float)pe[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = c0 + e;
                    v[it][q][e] = c < dim ? (float)xr[c] : (MODE == PACK_MLP && c == dim ? tv : 0.f);
                }
            }
        }
        if constexpr (MODE != PACK_MLP) {
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!(MODE == PACK_HB_LOG0 && q == 0 && e == 0 && lane == 0)) ss += v[it][q][e] * v[it][q][e];
#pragma unroll
            for (int off = 32; off; off >>= 1) ss += __shfl_xor(ss, off);
            if constexpr (MODE == PACK_NORM) {
                if (lane == 0 && row_raw < rows) norms[row] = sqrtf(ss);
            } else {
                const float x0 = __shfl(v[it][0][0], 0);
                if (lane == it) { ss_l = ss; x0_l = x0; }
            }
        }
    }
    float f_l = 1.f;
    if constexpr (MODE == PACK_HB_LOG0) {                       // hyperboloid.py:103-113, the 4 rows at once
        const float sK = (float)cv.sK;
        const float n = clamp_min(hd_sqrt(ss_l), (float)kMinNorm);
        const float theta = clamp_min(x0_l / sK, (float)(1.0 + Lim<float>::hb_eps));
        f_l = sK * arcosh_ref(theta) / n;
    }
#pragma unroll
    for (int it = 0; it < RW; ++it) {
        const int r16 = wave * RW + it;
        const int row = blockIdx.x * 16 + r16;
        if constexpr (MODE == PACK_HB_LOG0) {
            const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f_l), it));
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][q][e] *= f;
            if (lane == 0) v[it][0][0] = 0.f;
        }
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[it][q][e]));
#pragma unroll
        for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const int e_row = F16 ? exp_for(m, 15) : 0;
        if constexpr (F16) {
            if (lane == 0 && row < rows) {
                amax[row] = __float_as_uint(m);
                ex[row] = e_row;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ch = lane + 64 * q;
            if (ch >= nchunk) continue;
            if constexpr (F16) {
                uint2 h0, l0, h1, l1;
                split4_f16(make_float4(ldexpf(v[it][q][0], e_row), ldexpf(v[it][q][1], e_row), ldexpf(v[it][q][2], e_row),
                                       ldexpf(v[it][q][3], e_row)), h0, l0);
                split4_f16(make_float4(ldexpf(v[it][q][4], e_row), ldexpf(v[it][q][5], e_row), ldexpf(v[it][q][6], e_row),
                                       ldexpf(v[it][q][7], e_row)), h1, l1);
                stage[ch * 17 + r16] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                stage[(nchunk + ch) * 17 + r16] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            } else {                                             // three bf16 limbs, no row scaling
                uint2 h0, m0, l0, h1, m1, l1;
                split4(make_float4(v[it][q][0], v[it][q][1], v[it][q][2], v[it][q][3]), h0, m0, l0);
                split4(make_float4(v[it][q][4], v[it][q][5], v[it][q][6], v[it][q][7]), h1, m1, l1);
                stage[ch * 17 + r16] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                stage[(nchunk + ch) * 17 + r16] = make_uint4(m0.x, m0.y, m1.x, m1.y);
                stage[(2 * nchunk + ch) * 17 + r16] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    }
    __syncthreads();
    const int row16 = threadIdx.x & 15;
    const int64_t grow = (int64_t)blockIdx.x * 16 + row16;
    if (grow >= rows) return;
    uint4* obase = out + (grow >> 7) * nslab * kTileOf<F16> + (grow & 127);
    for (int p = threadIdx.x >> 4; p < kLimbs<F16> * nchunk; p += 16) {
        const int l = p / nchunk, ch = p - l * nchunk;
        obase[(int64_t)(ch >> 2) * kTileOf<F16> + l * 4 * 128 + (ch & 3) * 128] = stage[p * 17 + row16];
    }
}

// limb image -> fp32 row-major (limbs summed in fp32, small first; tests and debugging)
template <bool F16>
__global__ __launch_bounds__(256) void limb_unpack(const uint4* __restrict__ in, int rows, int cols, int nslab,
                                                   float* __restrict__ out, int64_t ld, int64_t total,
                                                   const int* __restrict__ ex) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int r = (int)(id & 127);
    const int64_t q = id >> 7;
    const int kc = (int)(q & 3);
    const int64_t tile = q >> 2;
    const int s = (int)(tile % nslab);
    const int64_t row = (tile / nslab) * 128 + r;
    if (row >= rows) return;
    const uint4* p = in + tile * kTileOf<F16> + kc * 128 + r;
    const int c0 = s * 32 + kc * 8;
    float a[8];
    if constexpr (F16) {
        const uint4 h = p[0], l = p[4 * 128];
        const unsigned hh[4] = {h.x, h.y, h.z, h.w}, ll[4] = {l.x, l.y, l.z, l.w};
        const int e_row = ex[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[2 * e] = ldexpf(hf_lo(ll[e]) + hf_lo(hh[e]), -e_row);
            a[2 * e + 1] = ldexpf(hf_hi(ll[e]) + hf_hi(hh[e]), -e_row);
        }
    } else {
        const uint4 h = p[0], m = p[4 * 128], l = p[8 * 128];
        const unsigned hh[4] = {h.x, h.y, h.z, h.w}, mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[2 * e] = (bf_lo(ll[e]) + bf_lo(mm[e])) + bf_lo(hh[e]);
            a[2 * e + 1] = (bf_hi(ll[e]) + bf_hi(mm[e])) + bf_hi(hh[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (c0 + e < cols) out[row * ld + c0 + e] = a[e];
}

// row scales of the f16x2 format (unused by bf16x3): exponents of the A rows (ea) and W rows (ew),
// exact row maxima of A (amax_a, bit patterns); outputs for the limb image CL: its row exponents
// (eo) and the running row maxima of the values written (amax_o, atomicMax on bit patterns: zeroed
// by the caller).  bound on |row of C| = amax_a * bmul + badd.
struct Scales {
    const int* ea;
    const unsigned* amax_a;
    const int* ew;
    int* eo;
    unsigned* amax_o;
    float bmul, badd;
};

// epi: 0 none, 1 gelu, 2 resid + gelu (ddpm.py:20-22; bf16x3 only).  C (fp32 row-major, ldc) and CL
// (limb image with nslab_out = ceil(N/32) slabs) are both optional outputs of the same values.
// write-out of 4 consecutive columns nb..nb+3 of row m (tile-local row mloc): v = accumulator + bias on
// entry; epilogue, fp32 and limb stores; rmax collects max|x| of what was written (f16x2).
// CHECK = false: interior tile, every column of the quad is < N (no per-element tests); vec = the fp32
// rows of C / resid are 16-byte aligned at every quad (tile-uniform).
template <bool F16, bool CHECK>
__device__ __forceinline__ void store_quad(float v[4], int m, int mloc, int nb, int half, int N, int nout, int ldc,
                                           int epi, bool vec, const float* __restrict__ resid, float* __restrict__ C,
                                           uint4* __restrict__ CL, size_t cl_tile_base, int eo_m, float& rmax,
                                           float* __restrict__ Z = nullptr) {
    if (Z) {                                                  // training: the pre-activation is kept for the backward pass
        float* zo = Z + (size_t)m * ldc + nb;
        if (vec && (!CHECK || nb + 3 < N)) {
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(zo));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!CHECK || nb + e < N) zo[e] = v[e];
        }
    }
    if (epi == 3) {                                           // backward: dz_below = g * gelu'(z_below), z_below = resid
        const float* rp = resid + (size_t)m * ldc + nb;
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && (!CHECK || nb + 3 < N)) {
            const float4 q = *reinterpret_cast<const float4*>(rp);
            r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!CHECK || nb + e < N) r[e] = rp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(r[e]);
    } else if (epi != 0) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (epi == 2) {
            const float* rp = resid + (size_t)m * ldc + nb;
            if (vec && (!CHECK || nb + 3 < N)) {
                const float4 q = *reinterpret_cast<const float4*>(rp);
                r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!CHECK || nb + e < N) r[e] = rp[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e] + gelu_erfc(v[e]);        // erf-GELU, single-polynomial erfc form
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (CHECK && nb + e >= N) v[e] = 0.f;
        if constexpr (F16) rmax = fmaxf(rmax, fabsf(v[e]));
    }
    if (C) {
        float* o = C + (size_t)m * ldc + nb;
        if (vec && (!CHECK || nb + 3 < N)) {
            if (Z || epi == 3) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(o));
            else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!CHECK || nb + e < N) o[e] = v[e];
        }
    }
    if (CL && (!CHECK || nb < nout)) {
        const int so = nb >> 5, kc = (nb & 31) >> 3;
        uint2* o = reinterpret_cast<uint2*>(CL + cl_tile_base + (size_t)so * kTileOf<F16> + kc * 128 + mloc) + half;
        if constexpr (F16) {
            uint2 h, l;
            split4_f16(make_float4(ldexpf(v[0], eo_m), ldexpf(v[1], eo_m), ldexpf(v[2], eo_m), ldexpf(v[3], eo_m)), h, l);
            o[0] = h;
            o[2 * 4 * 128] = l;
        } else {
            uint2 h, mm, l;
            split4(make_float4(v[0], v[1], v[2], v[3]), h, mm, l);
            o[0] = h;
            o[2 * 4 * 128] = mm;
            o[2 * 8 * 128] = l;
        }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 4 || N == 6 || N == 8 || N == 9 || N == 12, "add the literal");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// Persistent kernel: workgroup w takes tiles w, w + grid, w + 2 grid, ...; the first slabs of its
// next tile stream into LDS while the current tile's last slabs are multiplied and written out.
// v_mfma_f32_16x16x32_{bf16,f16} (one k-step per 32-deep slab): same cycles and LDS reads as
// 32x32x16, but the chip holds a higher clock on this shape under load (measured +13 % on the
// denoiser chain).
template <int BNT, int WAVES, bool F16, int NBUF = 2>
__global__ __launch_bounds__(WAVES * 64) void gemm_limbs(const uint4* __restrict__ Ag, const uint4* __restrict__ Wg,
                                                  const float* __restrict__ bias, const float* __restrict__ resid,
                                                  float* __restrict__ C, uint4* __restrict__ CL, int M, int N,
                                                  int nslab, int ldc, int epi, int tiles_m, int tiles_n,
                                                  int nslab_out, Scales sc, int phase_mode, int delay_ticks,
                                                  float* __restrict__ Z) {
    constexpr int L = kLimbs<F16>, kTile = kTileOf<F16>;
    constexpr int WCN = WAVES / 2;               // wavefronts are laid out 2 (rows) x WCN (columns)
    constexpr int WCOLS = BNT / WCN;             // columns per wavefront: 32 or 64
    constexpr int MT = 4, NT16 = WCOLS / 16;     // 16x16 MFMA tiles per wavefront
    constexpr int QA = 8 * L / WAVES, QW = (4 * L * BNT / 64) / WAVES;   // 1 KiB pieces per wavefront and slab
    constexpr int CH_A = 4 * L * 128, CH_W = 4 * L * BNT, CH_BUF = CH_A + CH_W;
    // NBUF = 1: one LDS buffer -- a slab's fragments are in registers before the multiply starts, so the next slab can
    // stream into the same buffer meanwhile (one slab in flight instead of two; 48 KB for the 128x128 tile, which
    // then runs two workgroups per CU with 64x64 per wavefront: a third of the LDS traffic per flop of the 128x64 form)
    __shared__ uint4 lds[NBUF * CH_BUF];
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
    // fp32 rows of C / resid: every 4-column quad is 16-byte aligned (quads start at multiples of 4)
    const bool vec = (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(resid) & 15) == 0 && (reinterpret_cast<uintptr_t>(Z) & 15) == 0;
    // operand maps of v_mfma_f32_16x16x32: lane (r16 = lane&15, kq = lane>>4) supplies row r16,
    // k = 8 kq .. 8 kq + 7 of the slab = chunk kc = kq.  First operand = W rows (n), second = A rows
    // (m): the accumulator is the C^T tile, D column = lane&15 = m, D rows 4 kq + reg = n.
    const int r16 = lane & 15, kq = lane >> 4;
    const int a_off = kq * 128 + wr * 64 + r16, w_off = CH_A + kq * BNT + wc * WCOLS + r16;

    int vt = blockIdx.x, tm, tn, tm2 = 0, tn2 = 0;
    decode(vt, tm, tn);
    // a next tile is prefetched across the tile boundary only when every tile has >= 2 slabs
    bool has_next = (NBUF == 1 || nslab >= 2) && vt + (int)gridDim.x < ntiles;
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
    if (phase_mode) {
        // De-phasing: the workgroups that share a CU all start together and their tiles take the same
        // time, so their write-outs (no MFMA work) would coincide for the whole launch.  The second
        // workgroup of a CU starts half a tile late: one's write-out then runs under the other's K loop.
        __shared__ int phase_s;
        if (tid == 0) {
            int ph;
            const int g = gridDim.x, s_ = blockIdx.x;
            const bool extra = s_ + (ntiles / g) * g < ntiles;      // this slot runs one tile more than the rest
            if (phase_mode == 1) ph = 2 * (s_ >= g / 2);
            else if (phase_mode == 2) ph = 2 * ((s_ >> 3) & 1);
            else if (phase_mode == 5) ph = extra ? 0 : 2 * (s_ >= g / 2);
            else if (phase_mode == 6) ph = extra ? 0 : 2 * (s_ >= g / 2) + ((s_ >> 3) & 1);
            else if (phase_mode == 7) ph = extra ? 0 : (s_ >= g / 2 ? 2 : 0) + ((s_ >> 3) & 1) * (s_ >= g / 2 ? -1 : 1);
            else {
                const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, 32 bits
                ph = 2 * (phase_mode == 3 ? (hw & 1) : ((hw >> 16) & 1));
            }
            phase_s = ph;
        }
        __syncthreads();
        if (phase_s) {
            const uint64_t t0 = wall_clock64(), wait = (uint64_t)delay_ticks * phase_s / 2;
            while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }
    HD_STAGE(a_cur, w_cur, 0, 0)
    if (NBUF == 2 && nslab > 1) HD_STAGE(a_cur, w_cur, 1, 1)
    for (;;) {
        f32x4 acc[MT][NT16];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT16; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int t = 0; t < nslab; ++t) {
            if (NBUF == 2 && (t + 1 < nslab || has_next)) wait_vmcnt<QA + QW>();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // slab t has landed, for every wave
            uint4 fa[MT][L], fw[NT16][L];
            const uint4* la = &lds[(NBUF == 2 ? ((pb + t) & 1) : 0) * CH_BUF];
#pragma unroll
            for (int l = 0; l < L; ++l) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i][l] = la[a_off + l * 4 * 128 + i * 16];
#pragma unroll
                for (int j = 0; j < NT16; ++j) fw[j][l] = la[w_off + l * 4 * BNT + j * 16];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // every wave holds its fragments: buffer free
            if constexpr (NBUF == 2) {
                if (t + 2 < nslab) HD_STAGE(a_cur, w_cur, t + 2, (pb + t) & 1)
                else if (has_next) HD_STAGE(a_nxt, w_nxt, t + 2 - nslab, (pb + t) & 1)
            } else {
                if (t + 1 < nslab) HD_STAGE(a_cur, w_cur, t + 1, 0)
                else if (has_next) HD_STAGE(a_nxt, w_nxt, 0, 0)
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT16; ++j) {
                    f32x4 c = acc[i][j];
                    if constexpr (F16) {
#define HD_MM(wl, al) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw[j][wl]), \
                                                                 __builtin_bit_cast(f16x8, fa[i][al]), c, 0, 0, 0);
                        HD_MM(1, 0) HD_MM(0, 1) HD_MM(0, 0)
#undef HD_MM
                    } else {
#define HD_MM(wl, al) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[j][wl]), \
                                                                  __builtin_bit_cast(bf16x8, fa[i][al]), c, 0, 0, 0);
                        HD_MM(0, 2) HD_MM(2, 0) HD_MM(1, 1) HD_MM(0, 1) HD_MM(1, 0) HD_MM(0, 0)
#undef HD_MM
                    }
                    acc[i][j] = c;
                }
        }
        // write-out of the C^T tile: a lane owns row m and 4 consecutive columns per accumulator.  Interior
        // tiles (every row < M, every column < N) take the copy without per-element tests.
        {
            const int m0 = tm * 128, n0 = tn * BNT;
            auto write_out = [&](auto check_c) {
                constexpr bool CHECK = decltype(check_c)::value;
                const size_t cl_base = (size_t)tm * nslab_out * kTile;
                int ewq[NT16][4];
                float bq[NT16][4];                                  // bias of this lane's columns, once per tile
#pragma unroll
                for (int j = 0; j < NT16; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + wc * WCOLS + j * 16 + 4 * kq + e;
                        const int nc = CHECK ? (n < N ? n : N - 1) : n;
                        bq[j][e] = bias ? bias[nc] : 0.f;
                        if constexpr (F16) ewq[j][e] = sc.ew[nc];
                    }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int mloc = wr * 64 + i * 16 + r16;
                    const int m = m0 + mloc;
                    if (CHECK && m >= M) {
                        // pad rows of the last row block: ZERO in the output image (the weight gradient contracts over
                        // image rows, hd_gemm_tn_limbs_f32), nothing else is written
                        if (CL) {
#pragma unroll
                            for (int j = 0; j < NT16; ++j) {
                                const int nb = n0 + wc * WCOLS + j * 16 + 4 * kq;
                                if (nb >= nout) continue;
                                uint2* o = reinterpret_cast<uint2*>(CL + cl_base + (size_t)(nb >> 5) * kTile + ((nb & 31) >> 3) * 128 +
                                                                    mloc) + (kq & 1);
                                o[0] = make_uint2(0u, 0u);
                                o[2 * 4 * 128] = make_uint2(0u, 0u);
                                if constexpr (!F16) o[2 * 8 * 128] = make_uint2(0u, 0u);
                            }
                        }
                        continue;
                    }
                    int ea_m = 0, eo_m = 0;
                    float rmax = 0.f;
                    if constexpr (F16) {
                        ea_m = sc.ea[m];
                        if (CL) {
                            eo_m = exp_for(__uint_as_float(sc.amax_a[m]) * sc.bmul + sc.badd, 14);
                            if (tn == 0 && wc == 0 && kq == 0) sc.eo[m] = eo_m;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NT16; ++j) {
                        const int nb = n0 + wc * WCOLS + j * 16 + 4 * kq;
                        if (CHECK && nb >= nout && nb >= N) continue;
                        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (F16) v[e] = ldexpf(v[e], -(ea_m + ewq[j][e]));
                            v[e] += bq[j][e];
                        }
                        store_quad<F16, CHECK>(v, m, mloc, nb, kq & 1, N, nout, ldc, epi, vec, resid, C, CL, cl_base,
                                               eo_m, rmax, Z);
                    }
                    if constexpr (F16) {
                        if (sc.amax_o) {                            // the 4 lanes of a row sit 16 apart
                            rmax = fmaxf(rmax, __shfl_xor(rmax, 16));
                            rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
                            if (kq == 0) atomicMax(&sc.amax_o[m], __float_as_uint(rmax));
                        }
                    }
                }
            };
            if (m0 + 128 <= M && n0 + BNT <= N) write_out(std::false_type{});
            else write_out(std::true_type{});
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

template <bool F16> int64_t image_bytes(int64_t rows, int64_t cols) {
    if (rows < 0 || cols < 0) return -1;
    return ((rows + 127) / 128) * ((cols + 31) / 32) * (int64_t)kTileOf<F16> * 16;
}

template <bool F16>
int pack_impl(const char* who, const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* ex,
              void* amax, void* st) {
    HD_REQUIRE(rows >= 0 && cols >= 1 && ld >= cols, "%s: bad shape", who);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(in && out && is_aligned16(out), "%s: null or misaligned pointer", who);
    HD_REQUIRE(!F16 || (ex && amax), "%s: null scale arrays", who);
    HD_REQUIRE(rows < (1LL << 31) && cols < (1LL << 31), "%s: size exceeds int32", who);
    const int nslab = (int)((cols + 31) / 32);
    const int64_t total = ((rows + 127) / 128) * nslab * 4 * 128;
    if (F16)
        row_absmax<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)st>>>(
            (const float*)in, ld, (int)rows, (int)cols, (unsigned*)amax, (int*)ex);
    limb_pack<F16><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st>>>(
        (const float*)in, ld, (int)rows, (int)cols, nslab, (uint4*)out, total, (const int*)ex);
    return check_launch(who);
}

template <bool F16>
int unpack_impl(const char* who, const void* in, const void* ex, int64_t rows, int64_t cols, void* out, int64_t ld,
                void* st) {
    HD_REQUIRE(rows >= 0 && cols >= 1 && ld >= cols, "%s: bad shape", who);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(in && out && (!F16 || ex), "%s: null pointer", who);
    const int nslab = (int)((cols + 31) / 32);
    const int64_t total = ((rows + 127) / 128) * nslab * 4 * 128;
    limb_unpack<F16><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)st>>>(
        (const uint4*)in, (int)rows, (int)cols, nslab, (float*)out, ld, total, (const int*)ex);
    return check_launch(who);
}

template <bool F16>
int gemm_impl(const char* who, const void* A, const void* W, const void* bias, const void* resid, void* C, void* CL,
              int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, Scales sc, void* st, void* Z = nullptr) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "%s: bad shape M=%lld N=%lld K=%lld", who, (long long)M, (long long)N,
               (long long)K);
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && (C || CL), "%s: null pointer", who);
    HD_REQUIRE(is_aligned16(A) && is_aligned16(W) && (!CL || is_aligned16(CL)),
               "%s: limb images must be 16-byte aligned", who);
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "%s: size exceeds int32", who);
    HD_REQUIRE(epi >= 0 && epi <= (F16 ? 1 : 3) && (epi < 2 || resid), "%s: bad epilogue / missing residual", who);
    HD_REQUIRE((!C && !Z && epi < 2) || ldc >= N, "%s: leading dimension too small", who);
    if (F16) {
        HD_REQUIRE(sc.ea && sc.ew, "%s: null row exponents", who);
        HD_REQUIRE(!CL || (sc.amax_a && sc.eo), "%s: a limb-image output needs amax_a and eo", who);
        if (!CL) sc.amax_o = nullptr;
    }
    const int tiles_m = (int)((M + 127) / 128);
    const int nslab = (int)((K + 31) / 32), nslab_out = (int)((N + 31) / 32);
    // tile: 'N' = 128x64 / 4 wavefronts, 'W' = 128x128 / 4 wavefronts, 'X' = 128x128 / 8 wavefronts, all with two LDS
    // buffers; 'T' = 128x64 / 4 wavefronts and 'S' = 128x128 / 4 wavefronts with ONE buffer (a slab's fragments sit in
    // registers before its multiplies start, so the next slab streams into the same buffer: one slab in flight).
    // Persistent grid: the workgroups that are resident at once (a multiple of 8, so a
    // workgroup's tiles stay on one XCD), unless K fits one slab.
    // Measured on the denoiser shapes (M = 15 777), chain of five GEMMs: bf16x3 'T' 485-497 us, 'S' 499-512 (de-phased),
    // 'N' 515-530 (round-2 default until then), same boxes; f16x2 'X' (two workgroups = 16 wavefronts per CU).
    // tuning knobs and the CU count are read ONCE per process (HD_GEMM_TUNE_RELOAD=1 re-reads them on every
    // launch: the tile-configuration sweeps of tools/bench_chain.py)
    struct Knobs { char cfg_bf, cfg_f16; int persist, per_cu, n_cu, reload, phase, delay_ns; };
    static Knobs knobs = [] {
        Knobs k{'T', 'X', 1, 0, 256, 0, -1, 500};
        if (const char* e = getenv("HD_GEMM_LIMBS_PHASE")) k.phase = atoi(e);
        if (const char* e = getenv("HD_GEMM_LIMBS_DELAY_NS")) k.delay_ns = atoi(e);
        if (const char* e = getenv("HD_GEMM_LIMBS_TUNE")) k.cfg_bf = e[0];
        if (const char* e = getenv("HD_GEMM_F16X2_TUNE")) k.cfg_f16 = e[0];
        if (const char* e = getenv("HD_GEMM_LIMBS_PERSIST")) k.persist = atoi(e);
        if (const char* e = getenv("HD_GEMM_LIMBS_PER_CU")) k.per_cu = atoi(e);
        if (const char* e = getenv("HD_GEMM_TUNE_RELOAD")) k.reload = atoi(e);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            k.n_cu = prop.multiProcessorCount;
        return k;
    }();
    if (knobs.reload) {
        if (const char* e = getenv("HD_GEMM_LIMBS_TUNE")) knobs.cfg_bf = e[0];
        if (const char* e = getenv("HD_GEMM_F16X2_TUNE")) knobs.cfg_f16 = e[0];
        if (const char* e = getenv("HD_GEMM_LIMBS_PHASE")) knobs.phase = atoi(e);
        if (const char* e = getenv("HD_GEMM_LIMBS_DELAY_NS")) knobs.delay_ns = atoi(e);
    }
    const char cfg = F16 ? knobs.cfg_f16 : knobs.cfg_bf;
    const int n_cu = knobs.n_cu, persist = knobs.persist, per_cu = knobs.per_cu;
    // de-phased start (phase mode 7) unless HD_GEMM_LIMBS_PHASE says otherwise; the single-buffer 128x64 form ('T')
    // measured better without it on three boxes (chain 485-497 us against 504-523)
    const int phase = knobs.phase >= 0 ? knobs.phase : (cfg == 'T' ? 0 : 7);
#define HD_LAUNCH(BNT_, WAVES_, PER_CU_) HD_LAUNCHB(BNT_, WAVES_, PER_CU_, 2)
#define HD_LAUNCHB(BNT_, WAVES_, PER_CU_, NBUF_)                                                                   \
    {                                                                                                      \
        const int tiles_n = (int)((N + BNT_ - 1) / BNT_);                                                  \
        const int64_t ntiles = (int64_t)tiles_m * tiles_n;                                                 \
        HD_REQUIRE(ntiles < (1LL << 31), "%s: too many tiles", who);                                       \
        int64_t grid = ntiles;                                                                             \
        const int64_t resident = (int64_t)(n_cu * (per_cu > 0 ? per_cu : PER_CU_)) / 8 * 8;                \
        if (persist && nslab >= 2 && resident >= 8 && grid > resident) grid = resident;                    \
        gemm_limbs<BNT_, WAVES_, F16, NBUF_><<<dim3((unsigned)grid), dim3(WAVES_ * 64), 0, (hipStream_t)st>>>( \
            (const uint4*)A, (const uint4*)W, (const float*)bias, (const float*)resid, (float*)C, (uint4*)CL,  \
            (int)M, (int)N, nslab, (int)ldc, epi, tiles_m, tiles_n, nslab_out, sc,                         \
            grid < ntiles ? phase : 0, (int)((int64_t)knobs.delay_ns * nslab / 10), (float*)Z);            \
    }
    if (cfg == 'W') HD_LAUNCH(128, 4, (F16 ? 2 : 1))
    else if (cfg == 'X') HD_LAUNCH(128, 8, (F16 ? 2 : 1))
    else if (cfg == 'S') HD_LAUNCHB(128, 4, 2, 1)
    else if (cfg == 'T') HD_LAUNCHB(64, 4, 2, 1)
    else HD_LAUNCH(64, 4, (F16 ? 3 : 2))
#undef HD_LAUNCHB
#undef HD_LAUNCH
    return check_launch(who);
}

}  // namespace

extern "C" int64_t hd_limb_image_bytes(int64_t rows, int64_t cols) { return image_bytes<false>(rows, cols); }
extern "C" int64_t hd_limb_image_bytes_f16x2(int64_t rows, int64_t cols) { return image_bytes<true>(rows, cols); }

extern "C" int hd_limb_pack_f32(const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* st) {
    return pack_impl<false>("hd_limb_pack_f32", in, ld, rows, cols, out, nullptr, nullptr, st);
}
extern "C" int hd_limb_pack_multi_f32(const hd_pack_job* jobs, int n_jobs, void* st) {
    HD_REQUIRE(n_jobs >= 0 && n_jobs <= kMaxPackJobs && (jobs || n_jobs == 0),
               "hd_limb_pack_multi_f32: between 0 and %d jobs per call", kMaxPackJobs);
    PackJobs pj;
    pj.n = 0;
    uint64_t blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const hd_pack_job& j = jobs[i];
        HD_REQUIRE(j.rows >= 0 && j.cols >= 1 && j.ld >= j.cols, "hd_limb_pack_multi_f32: job %d: bad shape", i);
        if (j.rows == 0) continue;
        HD_REQUIRE(j.src && j.image && is_aligned16(j.image), "hd_limb_pack_multi_f32: job %d: null or misaligned pointer", i);
        HD_REQUIRE(j.rows < (1LL << 31) && j.cols < (1LL << 31), "hd_limb_pack_multi_f32: job %d: size exceeds int32", i);
        const int64_t lrows = j.transposed ? j.cols : j.rows, lcols = j.transposed ? j.rows : j.cols;
        PackJob& o = pj.j[pj.n++];
        o.src = (const float*)j.src;
        o.out = (uint4*)j.image;
        o.ld = j.ld;
        o.rows = (int)lrows;
        o.cols = (int)lcols;
        o.nslab = (int)((lcols + 31) / 32);
        o.trans = j.transposed ? 1 : 0;
        o.block0 = (unsigned)blocks;
        blocks += (uint64_t)((lrows + 127) / 128) * o.nslab;               // one workgroup per tile and slab (512 chunks)
        HD_REQUIRE(blocks < (1ULL << 31), "hd_limb_pack_multi_f32: too many workgroups");
    }
    if (pj.n == 0) return HD_OK;
    limb_pack_multi<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>(pj);
    return check_launch("hd_limb_pack_multi_f32");
}
extern "C" int hd_limb_pack_f16x2_f32(const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* ex,
                                      void* amax, void* st) {
    return pack_impl<true>("hd_limb_pack_f16x2_f32", in, ld, rows, cols, out, ex, amax, st);
}

// rows wider than ~950 columns stage more than the 64 KiB of LDS a workgroup gets by default
template <typename K> static int allow_lds(K kernel, size_t bytes, const char* who) {
    if (bytes <= 64 * 1024) return HD_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%zu B of LDS): %s", who, bytes, hipGetErrorString(e)); return HD_EHIP; }
    return HD_OK;
}

template <int MODE, typename TX, bool F16>
static int launch_row_pack(const char* who, const TX* x, const void* t, int t_kind, float ts, int64_t rows, int64_t dim,
                           int64_t cols, Curv cv, void* out, void* ex, void* amax, void* norms, void* st) {
    const int nslab = (int)((cols + 31) / 32);
    const dim3 grid((unsigned)((rows + 15) / 16));
    const size_t lds = (size_t)kLimbs<F16> * nslab * 4 * 17 * sizeof(uint4);
    auto go = [&](auto nq_c) -> int {
        constexpr int NQ = decltype(nq_c)::value;
        if (int rc = allow_lds(row_pack_f16x2<MODE, TX, F16, NQ>, lds, who)) return rc;
        row_pack_f16x2<MODE, TX, F16, NQ><<<grid, 256, lds, (hipStream_t)st>>>(
            x, t, t_kind, ts, (int)rows, (int)dim, nslab, cv, (uint4*)out, (int*)ex, (unsigned*)amax, (float*)norms);
        return check_launch(who);
    };
    const int nq = (int)((cols + 511) / 512);
    return nq <= 1 ? go(std::integral_constant<int, 1>{}) : (nq == 2 ? go(std::integral_constant<int, 2>{})
                                                                    : go(std::integral_constant<int, 3>{}));
}

extern "C" int hd_mlp_pack_input_f16x2(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar,
                                       void* out, void* ex, void* amax, int64_t rows, int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim + 1 <= 64 * 8 * kPackChunks, "hd_mlp_pack_input_f16x2: dim out of range (<= %d)",
               64 * 8 * kPackChunks - 1);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && ex && amax && (t || t_kind == 2) && t_kind >= 0 && t_kind <= 2 && is_aligned16(out),
               "hd_mlp_pack_input_f16x2: bad argument");
    HD_REQUIRE(rows < (1LL << 31), "hd_mlp_pack_input_f16x2: size exceeds int32");
    if (x_dtype)
        return launch_row_pack<PACK_MLP, double, true>("hd_mlp_pack_input_f16x2", (const double*)x, t, t_kind, (float)t_scalar,
                                                       rows, dim, dim + 1, Curv{}, out, ex, amax, nullptr, st);
    return launch_row_pack<PACK_MLP, float, true>("hd_mlp_pack_input_f16x2", (const float*)x, t, t_kind, (float)t_scalar, rows,
                                                  dim, dim + 1, Curv{}, out, ex, amax, nullptr, st);
}

extern "C" int hd_row_pack_f16x2_f32(const void* x, int mode, hd_curv c, void* out, void* ex, void* amax, void* norms,
                                     int64_t rows, int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim <= 64 * 8 * kPackChunks, "hd_row_pack_f16x2_f32: dim out of range (<= %d)",
               64 * 8 * kPackChunks);
    HD_REQUIRE(mode == PACK_HB_LOG0 || mode == PACK_NORM, "hd_row_pack_f16x2_f32: mode must be 1 (hyperboloid logmap0) or 2 (rows + norms)");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && ex && amax && is_aligned16(out) && (mode != PACK_NORM || norms),
               "hd_row_pack_f16x2_f32: null or misaligned pointer");
    HD_REQUIRE(rows < (1LL << 31), "hd_row_pack_f16x2_f32: size exceeds int32");
    const Curv cv{c.c, c.sqrt_c, c.K, c.sqrt_K};
    if (mode == PACK_HB_LOG0)
        return launch_row_pack<PACK_HB_LOG0, float, true>("hd_row_pack_f16x2_f32", (const float*)x, nullptr, 2, 0.f, rows, dim,
                                                          dim, cv, out, ex, amax, nullptr, st);
    return launch_row_pack<PACK_NORM, float, true>("hd_row_pack_f16x2_f32", (const float*)x, nullptr, 2, 0.f, rows, dim, dim, cv,
                                                   out, ex, amax, norms, st);
}

// the same one-pass producers for the bf16x3 format (three limbs, no row exponents): 48 B of LDS per element
// of the 16-row stage, so rows of up to 1024 columns
constexpr int kPackMaxBf16x3 = 1024;
extern "C" int hd_mlp_pack_input_bf16x3(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar,
                                        void* out, int64_t rows, int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim + 1 <= kPackMaxBf16x3, "hd_mlp_pack_input_bf16x3: dim out of range (<= %d)",
               kPackMaxBf16x3 - 1);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && (t || t_kind == 2) && t_kind >= 0 && t_kind <= 2 && is_aligned16(out),
               "hd_mlp_pack_input_bf16x3: bad argument");
    HD_REQUIRE(rows < (1LL << 31), "hd_mlp_pack_input_bf16x3: size exceeds int32");
    if (x_dtype)
        return launch_row_pack<PACK_MLP, double, false>("hd_mlp_pack_input_bf16x3", (const double*)x, t, t_kind, (float)t_scalar,
                                                        rows, dim, dim + 1, Curv{}, out, nullptr, nullptr, nullptr, st);
    return launch_row_pack<PACK_MLP, float, false>("hd_mlp_pack_input_bf16x3", (const float*)x, t, t_kind, (float)t_scalar, rows,
                                                   dim, dim + 1, Curv{}, out, nullptr, nullptr, nullptr, st);
}

extern "C" int hd_row_pack_bf16x3_f32(const void* x, int mode, hd_curv c, void* out, void* norms, int64_t rows,
                                      int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim <= kPackMaxBf16x3, "hd_row_pack_bf16x3_f32: dim out of range (<= %d)",
               kPackMaxBf16x3);
    HD_REQUIRE(mode == PACK_HB_LOG0 || mode == PACK_NORM, "hd_row_pack_bf16x3_f32: mode must be 1 (hyperboloid logmap0) or 2 (rows + norms)");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && is_aligned16(out) && (mode != PACK_NORM || norms), "hd_row_pack_bf16x3_f32: null or misaligned pointer");
    HD_REQUIRE(rows < (1LL << 31), "hd_row_pack_bf16x3_f32: size exceeds int32");
    const Curv cv{c.c, c.sqrt_c, c.K, c.sqrt_K};
    if (mode == PACK_HB_LOG0)
        return launch_row_pack<PACK_HB_LOG0, float, false>("hd_row_pack_bf16x3_f32", (const float*)x, nullptr, 2, 0.f, rows, dim,
                                                           dim, cv, out, nullptr, nullptr, nullptr, st);
    return launch_row_pack<PACK_NORM, float, false>("hd_row_pack_bf16x3_f32", (const float*)x, nullptr, 2, 0.f, rows, dim, dim,
                                                    cv, out, nullptr, nullptr, norms, st);
}

extern "C" int hd_limb_unpack_f32(const void* in, int64_t rows, int64_t cols, void* out, int64_t ld, void* st) {
    return unpack_impl<false>("hd_limb_unpack_f32", in, nullptr, rows, cols, out, ld, st);
}
extern "C" int hd_limb_unpack_f16x2_f32(const void* in, const void* ex, int64_t rows, int64_t cols, void* out,
                                        int64_t ld, void* st) {
    return unpack_impl<true>("hd_limb_unpack_f16x2_f32", in, ex, rows, cols, out, ld, st);
}

extern "C" int hd_gemm_limbs_f32(const void* A, const void* W, const void* bias, const void* resid, void* C,
                                 void* CL, int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, void* st) {
    HD_REQUIRE(epi >= 0 && epi <= 2, "hd_gemm_limbs_f32: bad epilogue");
    return gemm_impl<false>("hd_gemm_limbs_f32", A, W, bias, resid, C, CL, M, N, K, ldc, epi, Scales{}, st);
}
extern "C" int hd_gemm_limbs_train_f32(const void* A, const void* W, const void* bias, const void* aux, void* Z, void* C,
                                       void* CL, int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, void* st) {
    HD_REQUIRE(epi == 0 || epi == 1 || epi == 3, "hd_gemm_limbs_train_f32: epilogue must be 0, 1 or 3");
    return gemm_impl<false>("hd_gemm_limbs_train_f32", A, W, bias, aux, C, CL, M, N, K, ldc, epi, Scales{}, st, Z);
}
extern "C" int hd_gemm_f16x2_f32(const void* A, const void* ea, const void* amax_a, const void* W, const void* ew,
                                 const void* bias, void* C, void* CL, void* eo, void* amax_o, int64_t M, int64_t N,
                                 int64_t K, int64_t ldc, int epi, float bound_mul, float bound_add, void* st) {
    Scales sc{(const int*)ea, (const unsigned*)amax_a, (const int*)ew, (int*)eo, (unsigned*)amax_o, bound_mul,
              bound_add};
    HD_REQUIRE(bound_mul >= 0.f && bound_add >= 0.f, "hd_gemm_f16x2_f32: negative bound coefficients");
    return gemm_impl<true>("hd_gemm_f16x2_f32", A, W, bias, nullptr, C, CL, M, N, K, ldc, epi, sc, st);
}
