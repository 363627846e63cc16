// fp32-accurate dense contraction on the bf16 matrix cores: C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]).
//
// Why: on gfx950 the exact-f32 MFMA (v_mfma_f32_32x32x2_f32, csrc/gemm.hip) runs at the
// vector-ALU rate, 1/16 of the bf16 MFMA rate.  Every fp32 operand is therefore split,
// exactly, into three bf16 limbs  a = a_h + a_m + a_l  (a_h = bf16(a), a_m = bf16(a - a_h),
// a_l = bf16(a - a_h - a_m); the two subtractions are exact in fp32, round-to-nearest limbs
// leave |a - a_h - a_m - a_l| <= 2^-26 |a|), and a product is evaluated as the six limb
// products whose weight is >= 2^-16:
//     a*w ~= a_h*w_h + a_h*w_m + a_m*w_h + a_m*w_m + a_h*w_l + a_l*w_h
// (each an exact bf16 x bf16 product, accumulated in fp32 by v_mfma_f32_32x32x16_bf16).  The
// dropped products (m*l, l*m, l*l) are below 2^-26 |a||w|, i.e. under the rounding of ONE fp32
// product, so the result carries fp32 accuracy (measured: same error against an fp64 product as
// the exact-f32 kernel, tests/test_gpu_parity.py) at 6/16 of the exact kernel's matrix-pipe time.
// No reduced-precision shortcut: nothing is rounded to bf16 that is not also carried by a
// lower limb.  Non-finite or > 3e38 inputs give NaN instead of inf (inf - inf in the split).
//
// Same operands, epilogues and launch geometry as gemm_f32_mfma (x @ W^T of HypLinear,
// manifolds/poincare.py:96, manifolds/hyperboloid.py:122; the denoisers' nn.Linear + bias +
// erf-GELU, models/direction_diffusion.py:56-69, ddpm.py:9-41).
//
// Tiling: 256 threads = 2x2 wavefronts over a 128 x BNT block tile, K in slabs of 32.  fp32 rows
// are loaded 16 bytes per lane into registers one slab ahead, split on the vector ALU and written
// to LDS as three bf16 planes per operand ([row][32 + 8 pad] bf16: the 80-byte pitch puts the
// sixteen 16-byte fragment reads of a lane group on sixteen distinct bank slots).  Per k-step of
// 16 a wavefront reads 3 limb fragments per 32-row block and issues 6 MFMAs per 32x32 output tile.
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

constexpr int BM = 128, BK = 32, PB = BK + 8;      // PB: LDS row pitch in bf16 elements (80 B)

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

// epi codes: see gemm.hip (0 none, 1 gelu, 2 residual + gelu, 3 input-gradient form)
template <int BNT>
__global__ __launch_bounds__(256) void gemm_f32_bf16x3(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ resid,
                                                       const float* __restrict__ aux, float* __restrict__ C,
                                                       float* __restrict__ C2,
                                                       int M, int N, int K, int lda, int ldw, int ldc, int epi,
                                                       int tiles_m, int tiles_n, int kps, float* __restrict__ partial) {
    constexpr int NT = BNT / 64;               // 32-col MFMA tiles per wavefront along N
    __shared__ __attribute__((aligned(16))) unsigned short sA[3][BM * PB];
    __shared__ __attribute__((aligned(16))) unsigned short sW[3][BNT * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order (see gemm.hip): every XCD walks a contiguous run of tiles, N fastest.
    int tile = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = tile & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BNT;

    // staging map: 8 threads cover one row's 32 floats; 32 rows per pass
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    float4 ra[BM / 32], rw[BNT / 32];
    bool kin = true;
    const float* pA[BM / 32];
    const float* pW[BNT / 32];
#pragma unroll
    for (int p = 0; p < BM / 32; ++p) {
        int r = m0 + srow + 32 * p;
        r = r < M ? r : M - 1;
        pA[p] = A + (size_t)r * lda;
    }
#pragma unroll
    for (int p = 0; p < BNT / 32; ++p) {
        int r = n0 + srow + 32 * p;
        r = r < N ? r : N - 1;
        pW[p] = W + (size_t)r * ldw;
    }
    // loads are unconditional (row and k clamped to valid addresses); a slab position beyond K
    // is zeroed when the registers are split and written to LDS one iteration later
#define HD_GLOAD(k0)                                                                        \
    {                                                                                       \
        kin = ((k0) + skq) < K;                                                             \
        const int kk = kin ? (k0) + skq : K - 4;                                            \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p)                                 \
            ra[p] = *reinterpret_cast<const float4*>(pA[p] + kk);                           \
        _Pragma("unroll") for (int p = 0; p < BNT / 32; ++p)                                \
            rw[p] = *reinterpret_cast<const float4*>(pW[p] + kk);                           \
    }
#define HD_SSTORE()                                                                         \
    {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p) {                               \
            float4 v = ra[p];                                                               \
            v.x = kin ? v.x : 0.f; v.y = kin ? v.y : 0.f; v.z = kin ? v.z : 0.f; v.w = kin ? v.w : 0.f; \
            uint2 h, m, l;                                                                  \
            split4(v, h, m, l);                                                             \
            const int o = (srow + 32 * p) * PB + skq;                                       \
            *reinterpret_cast<uint2*>(&sA[0][o]) = h;                                       \
            *reinterpret_cast<uint2*>(&sA[1][o]) = m;                                       \
            *reinterpret_cast<uint2*>(&sA[2][o]) = l;                                       \
        }                                                                                   \
        _Pragma("unroll") for (int p = 0; p < BNT / 32; ++p) {                              \
            float4 v = rw[p];                                                               \
            v.x = kin ? v.x : 0.f; v.y = kin ? v.y : 0.f; v.z = kin ? v.z : 0.f; v.w = kin ? v.w : 0.f; \
            uint2 h, m, l;                                                                  \
            split4(v, h, m, l);                                                             \
            const int o = (srow + 32 * p) * PB + skq;                                       \
            *reinterpret_cast<uint2*>(&sW[0][o]) = h;                                       \
            *reinterpret_cast<uint2*>(&sW[1][o]) = m;                                       \
            *reinterpret_cast<uint2*>(&sW[2][o]) = l;                                       \
        }                                                                                   \
    }
    // one K slab = two k-steps of 16; operand maps of v_mfma_f32_32x32x16_bf16: lane (r = lane&31,
    // h = lane>>5) holds row r, k = 8h .. 8h+7 of the step (16 contiguous bytes of its LDS row).
    // Small limb products first, so that they meet before being added to the large sum.
#define HD_COMPUTE()                                                                        \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                         \
        bf16x8 fa[2][3], fw[NT][3];                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                       \
            _Pragma("unroll") for (int l = 0; l < 3; ++l)                                   \
                fa[i][l] = *reinterpret_cast<const bf16x8*>(                                \
                    &sA[l][(wr * 64 + i * 32 + fi) * PB + s * 16 + fh * 8]);                \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                      \
            _Pragma("unroll") for (int l = 0; l < 3; ++l)                                   \
                fw[j][l] = *reinterpret_cast<const bf16x8*>(                                \
                    &sW[l][(wc * (BNT / 2) + j * 32 + fi) * PB + s * 16 + fh * 8]);         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                       \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                \
                f32x16 c = acc[i][j];                                                       \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fw[j][0], c, 0, 0, 0); \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][2], c, 0, 0, 0); \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fw[j][1], c, 0, 0, 0); \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fw[j][0], c, 0, 0, 0); \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][1], c, 0, 0, 0); \
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fw[j][0], c, 0, 0, 0); \
                acc[i][j] = c;                                                              \
            }                                                                               \
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    // a wavefront whose quadrant lies entirely beyond M or N only stages (see gemm.hip)
    const bool wave_live = (m0 + wr * 64 < M) && (n0 + wc * (BNT / 2) < N);
    // split-K (partial != nullptr): workgroup (tile, z = blockIdx.y) multiplies slabs [z kps, (z + 1) kps) only and stores
    // the raw accumulators in partial[z]; splitk_epilogue adds the splits in order and applies bias / epilogue
    const int nk_all = (K + BK - 1) / BK;
    const int kt0 = partial ? (int)blockIdx.y * kps : 0;
    const int nk = partial ? (kt0 + kps < nk_all ? kt0 + kps : nk_all) : nk_all;
    HD_GLOAD(kt0 * BK)
    if (!wave_live) {
        for (int kt = kt0; kt < nk; ++kt) {
            HD_SSTORE()
            __syncthreads();
            HD_GLOAD((kt + 1) * BK)
            __syncthreads();
        }
        return;
    }
    for (int kt = kt0; kt < nk; ++kt) {
        HD_SSTORE()
        __syncthreads();
        HD_GLOAD((kt + 1) * BK)                    // past the end: clamped re-read, never stored
        __builtin_amdgcn_sched_barrier(0);
        HD_COMPUTE()
        __syncthreads();
    }
#undef HD_GLOAD
#undef HD_SSTORE
#undef HD_COMPUTE

    // C/D map of the 32x32 tile (dtype-independent): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cn = n0 + wc * (BNT / 2) + j * 32 + fi;
            if (cn >= N) continue;
            if (partial) {
                float* po = partial + (size_t)blockIdx.y * M * N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rm = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (rm < M) po[(size_t)rm * N + cn] = acc[i][j][r];
                }
                continue;
            }
            const float bv = bias ? bias[cn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rm = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (rm < M) {
                    const size_t o = (size_t)rm * ldc + cn;
                    float v = acc[i][j][r] + bv;
                    if (epi == 3) {
                        if (resid) v += resid[o];
                        if (C) C[o] = v;
                        if (C2) C2[o] = v * gelu_erf_grad(aux[o]);
                    } else {
                        if (epi != 0 && C2) C2[o] = v;
                        if (epi == 1) v = gelu_erf(v);
                        else if (epi == 2) v = resid[o] + gelu_erf(v);
                        C[o] = v;
                    }
                }
            }
        }
}

// sum of the split-K partials in split order, then the write-out of gemm_f32_bf16x3 (same expressions)
__global__ __launch_bounds__(256) void splitk_epilogue(const float* __restrict__ partial, int splits,
                                                       const float* __restrict__ bias, const float* __restrict__ resid,
                                                       const float* __restrict__ aux, float* __restrict__ C,
                                                       float* __restrict__ C2, int M, int N, int ldc, int epi) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int rm = (int)(i / N), cn = (int)(i - (size_t)rm * N);
        float a = partial[i];
        for (int z = 1; z < splits; ++z) a += partial[(size_t)z * total + i];
        const size_t o = (size_t)rm * ldc + cn;
        float v = a + (bias ? bias[cn] : 0.f);
        if (epi == 3) {
            if (resid) v += resid[o];
            if (C) C[o] = v;
            if (C2) C2[o] = v * gelu_erf_grad(aux[o]);
        } else {
            if (epi != 0 && C2) C2[o] = v;
            if (epi == 1) v = gelu_erf(v);
            else if (epi == 2) v = resid[o] + gelu_erf(v);
            C[o] = v;
        }
    }
}

// split-K only pays where the tiles alone leave most of the chip idle (the graph-level DDPM's 200-row batches:
// 200 x 512 x 512 is 16 tiles of 16 slabs, 57 us of exposed load latency on 16 CUs)
inline int splitk_splits(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + 63) / 64), nk = (K + BK - 1) / BK;
    if (nk < 4) return 1;
    // few tiles: up to 256 workgroups of >= 2 slabs; a chip's worth of tiles with a long K (config 4's 2708-wide
    // identity features: 176 tiles of 85 slabs): up to 512 workgroups of >= 16 slabs
    int64_t s = tiles < 64 ? 256 / tiles : 512 / tiles;
    const int64_t cap = tiles < 64 ? nk / 2 : nk / 16;
    s = s < cap ? s : cap;
    return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}

}  // namespace

extern "C" int64_t hd_gemm_splitk_workspace_floats(int64_t M, int64_t N, int64_t K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    const int s = splitk_splits(M, N, K);
    return s > 1 ? (int64_t)s * M * N : 0;
}

static int gemm_bf16x3_impl(const void* A, const void* W, const void* bias, const void* resid, const void* aux,
                            void* C, void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                            int64_t ldc, int epi, void* workspace, void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "hd_gemm_f32_bf16x3: bad shape M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && (C || (epi == 3 && C2)), "hd_gemm_f32_bf16x3: null pointer");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "hd_gemm_f32_bf16x3: size exceeds int32");
    HD_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && is_aligned16(A) && is_aligned16(W),
               "hd_gemm_f32_bf16x3: K, lda, ldw must be multiples of 4 and A, W 16-byte aligned (pad with zeros)");
    HD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "hd_gemm_f32_bf16x3: leading dimension too small");
    HD_REQUIRE(epi >= 0 && epi <= 3 && (epi != 2 || resid), "hd_gemm_f32_bf16x3: bad epilogue / missing residual");
    HD_REQUIRE(epi != 3 || !C2 || aux, "hd_gemm_f32_bf16x3: gradient epilogue needs the saved pre-activation");
    const int tiles_m = (int)((M + BM - 1) / BM);
    const int splits = workspace ? splitk_splits(M, N, K) : 1;
    if (splits > 1) {
        HD_REQUIRE(is_aligned16(workspace), "hd_gemm_f32_bf16x3_splitk: misaligned workspace");
        const int nk = (int)((K + BK - 1) / BK), kps = (nk + splits - 1) / splits, zs = (nk + kps - 1) / kps;
        const int tiles_n = (int)((N + 63) / 64);
        gemm_f32_bf16x3<64><<<dim3((unsigned)(tiles_m * tiles_n), (unsigned)zs), dim3(256), 0, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, nullptr, nullptr, nullptr, nullptr, nullptr, (int)M, (int)N, (int)K, (int)lda,
            (int)ldw, (int)N, 0, tiles_m, tiles_n, kps, (float*)workspace);
        int blocks = (int)(((size_t)M * N + 255) / 256);
        blocks = blocks < 2048 ? blocks : 2048;
        splitk_epilogue<<<blocks, 256, 0, (hipStream_t)st>>>((const float*)workspace, zs, (const float*)bias, (const float*)resid,
                                                             (const float*)aux, (float*)C, (float*)C2, (int)M, (int)N, (int)ldc, epi);
        return check_launch("hd_gemm_f32_bf16x3_splitk");
    }
    // 128x64 tiles where 128x128 ones would leave CUs without a workgroup (M = 4096, N = 512: 128 -> 256 workgroups)
    int width = (int64_t)tiles_m * ((N + 127) / 128) < 256 ? 64 : 128;
    if (const char* e = getenv("HD_GEMM_SPLIT_TUNE")) {      // tuning hook: "n" = 128x64 tile, "w" = 128x128
        if (e[0] == 'n') width = 64; else if (e[0] == 'w') width = 128;
    }
    if (width == 64) {
        const int tiles_n = (int)((N + 63) / 64);
        gemm_f32_bf16x3<64><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (const float*)aux, (float*)C,
            (float*)C2, (int)M, (int)N, (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n, 0, nullptr);
    } else {
        const int tiles_n = (int)((N + 127) / 128);
        gemm_f32_bf16x3<128><<<dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (const float*)aux, (float*)C,
            (float*)C2, (int)M, (int)N, (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n, 0, nullptr);
    }
    return check_launch("hd_gemm_f32_bf16x3");
}

extern "C" int hd_gemm_f32_bf16x3(const void* A, const void* W, const void* bias, const void* resid, const void* aux,
                                  void* C, void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                  int64_t ldc, int epi, void* st) {
    return gemm_bf16x3_impl(A, W, bias, resid, aux, C, C2, M, N, K, lda, ldw, ldc, epi, nullptr, st);
}

extern "C" int hd_gemm_f32_bf16x3_splitk(const void* A, const void* W, const void* bias, const void* resid, const void* aux,
                                         void* C, void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                         int64_t ldc, int epi, void* workspace, void* st) {
    return gemm_bf16x3_impl(A, W, bias, resid, aux, C, C2, M, N, K, lda, ldw, ldc, epi, workspace, st);
}
