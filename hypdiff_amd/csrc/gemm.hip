// Dense contractions of the path: C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]).
//
// fp32 runs on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an
// fmaf chain, so the 1e-5 parity bar holds without any reduced-precision trick).
// Used by HypLinear's x @ W^T (manifolds/poincare.py:96, manifolds/hyperboloid.py:122)
// and by the denoiser's five Linear layers (models/direction_diffusion.py:56-69), whose
// bias add and exact-erf GELU are fused into the accumulator write-out.
//
// Tiling: 256 threads = 4 wavefronts in a 2x2 arrangement over a 128x128 block tile,
// each wavefront 64x64 = 2x2 MFMA tiles (64 accumulator registers).  K advances in
// slabs of 32 staged through LDS as [row][32 + 4 pad] floats: the 144-byte row pitch
// makes the 16-byte fragment reads of a 16-lane group hit 16 distinct 4-bank slots.
// Inside a slab lane-half h of the MFMA takes k = 16h .. 16h+15, i.e. each lane reads
// 16 contiguous floats of its row (4 x ds_read_b128) for 16 MFMA steps.  Global loads
// of slab t+1 are issued before the MFMAs of slab t (register staging).
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "../../include/hypdiff.h"
#include <stdlib.h>

using namespace hd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 4;

// epi: 0 none, 1 gelu, 2 residual + gelu (ddpm.py:20-22: x0 + gelu(layer(x0)));
//      with C2 != NULL modes 1 and 2 also store the pre-activation z = acc + bias (training).
//      3 = input-gradient form: g = acc (+ resid); C = g if C != NULL; C2 = g * gelu'(aux) if
//      C2 != NULL (the next layer-down's pre-activation gradient, fused into the write-out).
template <int BNT>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const float* __restrict__ A, const float* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ resid,
                                                     const float* __restrict__ aux, float* __restrict__ C,
                                                     float* __restrict__ C2,
                                                     int M, int N, int K, int lda, int ldw, int ldc, int epi,
                                                     int tiles_m, int tiles_n) {
    constexpr int NT = BNT / 64;               // 32-col MFMA tiles per wavefront along N: 2 (BN=128) or 1 (BN=64)
    __shared__ __attribute__((aligned(16))) float sA[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float sW[BNT * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // Tile order.  The hardware deals consecutive workgroup ids round-robin over the 8
    // XCDs (private L2 each).  Give every XCD a contiguous run of tile ids (bijective
    // remap) and walk N fastest inside it, so the workgroups that share one 128-row A
    // panel run back to back on ONE L2 and the (small) weight matrix stays L2-resident;
    // A then leaves HBM / Infinity Cache once instead of once per N tile (+7 % measured).
    int tile = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = tile & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BNT;

    // staging map: 8 threads cover one row's 128 B; 32 rows per pass
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    float4 ra[BM / 32], rw[BNT / 32];

    // Register staging, written as straight-line macros (a lambda capturing the arrays makes
    // them addressable and hipcc then parks them in scratch memory).  Loads are unconditional:
    // row and k are clamped to valid addresses so that no branch and no early s_waitcnt sits
    // between their issue and the MFMAs; a slab position beyond K is zeroed when the registers
    // are written to LDS one iteration later.
    bool kinA = true;
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
#define HD_GLOAD(RA, RW, KIN, k0)                                                           \
    {                                                                                       \
        KIN = ((k0) + skq) < K;                                                             \
        const int kk = KIN ? (k0) + skq : K - 4;                                            \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p)                                 \
            RA[p] = *reinterpret_cast<const float4*>(pA[p] + kk);                           \
        _Pragma("unroll") for (int p = 0; p < BNT / 32; ++p)                                \
            RW[p] = *reinterpret_cast<const float4*>(pW[p] + kk);                           \
    }
#define HD_SSTORE(RA, RW, KIN)                                                              \
    {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < BM / 32; ++p) {                               \
            float4 v = RA[p];                                                               \
            v.x = KIN ? v.x : 0.f; v.y = KIN ? v.y : 0.f; v.z = KIN ? v.z : 0.f; v.w = KIN ? v.w : 0.f; \
            *reinterpret_cast<float4*>(&sA[(srow + 32 * p) * PITCH + skq]) = v;             \
        }                                                                                   \
        _Pragma("unroll") for (int p = 0; p < BNT / 32; ++p) {                              \
            float4 v = RW[p];                                                               \
            v.x = KIN ? v.x : 0.f; v.y = KIN ? v.y : 0.f; v.z = KIN ? v.z : 0.f; v.w = KIN ? v.w : 0.f; \
            *reinterpret_cast<float4*>(&sW[(srow + 32 * p) * PITCH + skq]) = v;             \
        }                                                                                   \
    }
// one K slab: fragments from LDS, 16 * 2 * NT MFMAs
#define HD_COMPUTE()                                                                        \
    {                                                                                       \
        float4 fa[2][4], fw[NT][4];                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                     \
            const float* pa = &sA[(wr * 64 + i * 32 + fi) * PITCH + fh * 16];               \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) fa[i][q] = *reinterpret_cast<const float4*>(pa + 4 * q); \
        }                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                    \
            const float* pw = &sW[(wc * (BNT / 2) + j * 32 + fi) * PITCH + fh * 16];        \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) fw[j][q] = *reinterpret_cast<const float4*>(pw + 4 * q); \
        }                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                   \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                               \
                    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                        \
                        const float a = reinterpret_cast<const float*>(&fa[i][q])[e];       \
                        const float b = reinterpret_cast<const float*>(&fw[j][q])[e];       \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0); \
                    }                                                                       \
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    // A wavefront whose 64 x (BNT/2) quadrant lies entirely beyond M or N (ragged last block
    // row / column: N = 900 fills 14 column blocks and 4 columns of a 15th) skips its MFMAs --
    // wave-uniform, and the matrix pipe it would have occupied serves the other resident
    // workgroups of the SIMD.  It still takes part in the staging and the barriers.
    const bool wave_live = (m0 + wr * 64 < M) && (n0 + wc * (BNT / 2) < N);
    const int nk = (K + BK - 1) / BK;
    // Software pipeline: while slab t is multiplied out of LDS the loads of slab t+1 are in
    // flight into registers.  The sched_barrier keeps the loads where they are written: hipcc
    // otherwise sinks them below the MFMAs to reuse the fragment registers, and the next
    // iteration then waits out the full memory latency.  (A second register set / prefetch
    // distance 2 was measured: no gain, 32 more VGPRs.)
    HD_GLOAD(ra, rw, kinA, 0)
    if (!wave_live) {
        // staging-only copy of the loop (same barrier count).  A separate loop rather than a
        // predicate around HD_COMPUTE: conditionally updated accumulators leave the AGPRs and
        // cost the live path 30+ VGPRs, i.e. one resident workgroup per CU.
        for (int kt = 0; kt < nk; ++kt) {
            HD_SSTORE(ra, rw, kinA)
            __syncthreads();
            HD_GLOAD(ra, rw, kinA, (kt + 1) * BK)
            __syncthreads();
        }
        return;
    }
    for (int kt = 0; kt < nk; ++kt) {
        HD_SSTORE(ra, rw, kinA)
        __syncthreads();
        HD_GLOAD(ra, rw, kinA, (kt + 1) * BK)      // past the end: clamped re-read, never stored
        __builtin_amdgcn_sched_barrier(0);
        HD_COMPUTE()
        __syncthreads();
    }
#undef HD_GLOAD
#undef HD_SSTORE
#undef HD_COMPUTE

    // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cn = n0 + wc * (BNT / 2) + j * 32 + fi;
            if (cn >= N) continue;
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

// fp64 fallback for double-precision HypLinear: 64x64 tile, 4x4 outputs per thread, VALU fma
__global__ __launch_bounds__(256) void gemm_f64_valu(const double* __restrict__ A, const double* __restrict__ W,
                                                     const double* __restrict__ bias, double* __restrict__ C,
                                                     int M, int N, int K, int lda, int ldw, int ldc) {
    constexpr int T = 64, KB = 16;
    __shared__ double sA[KB][T + 1], sW[KB][T + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * T, n0 = blockIdx.y * T;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += KB) {
        for (int i = tid; i < T * KB; i += 256) {
            const int r = i / KB, k = i % KB;
            const int gm = m0 + r, gn = n0 + r, gk = k0 + k;
            sA[k][r] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.0;
            sW[k][r] = (gn < N && gk < K) ? W[(size_t)gn * ldw + gk] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[k][ty * 4 + i]; b[i] = sW[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
            if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = acc[i][j] + (bias ? bias[gn] : 0.0);
        }
}

}  // namespace

extern "C" int hd_gemm_f32_ex(const void* A, const void* W, const void* bias, const void* resid, const void* aux,
                              void* C, void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                              int64_t ldc, int epi, void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "hd_gemm_f32: bad shape M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && (C || (epi == 3 && C2)), "hd_gemm_f32: null pointer");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "hd_gemm_f32: size exceeds int32");
    HD_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && is_aligned16(A) && is_aligned16(W),
               "hd_gemm_f32: K, lda, ldw must be multiples of 4 and A, W 16-byte aligned (pad with zeros)");
    HD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "hd_gemm_f32: leading dimension too small");
    HD_REQUIRE(epi >= 0 && epi <= 3 && (epi != 2 || resid), "hd_gemm_f32: bad epilogue / missing residual");
    HD_REQUIRE(epi != 3 || !C2 || aux, "hd_gemm_f32: gradient epilogue needs the saved pre-activation");
    const dim3 blk(256);
    // Tile width and residency (workgroups per CU, capped by requesting unused dynamic LDS).
    // Measured on MI355X over the denoiser / HypLinear shapes (M = 15 777, N,K in 512..900):
    // the 128x64 tile at 4 workgroups per CU is best or within 1 % of best everywhere (less
    // padding of N, more workgroups per CU to overlap the LDS hand-off and the write-out of
    // one with the MFMAs of the others); large square problems prefer 128x128 at 2 per CU.
    const int tiles_m = (int)((M + BM - 1) / BM);
    int best_w = 64, best_r = 4;
    if (N >= 2048 && M >= 2048 && N % 128 == 0) { best_w = 128; best_r = 2; }
    if (const char* e = getenv("HD_GEMM_TUNE")) {          // tuning hook: "<n|w><R>"
        if (e[0] == 'n') best_w = 64; else if (e[0] == 'w') best_w = 128;
        if (e[0] && e[1] >= '1' && e[1] <= '4') best_r = e[1] - '0';
    }
    // LDS per CU 160 KiB: R resident workgroups <=> each asks for more than 160/(R+1) KiB
    const size_t static_lds = (size_t)(BM + best_w) * PITCH * sizeof(float);
    size_t want = (size_t)(160 * 1024) / (best_r + 1) + 1024;
    const size_t dyn = want > static_lds ? want - static_lds : 0;
    if (best_w == 64) {
        const int tiles_n = (int)((N + 63) / 64);
        gemm_f32_mfma<64><<<dim3((unsigned)(tiles_m * tiles_n)), blk, dyn, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (const float*)aux, (float*)C,
            (float*)C2, (int)M, (int)N, (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n);
    } else {
        const int tiles_n = (int)((N + 127) / 128);
        gemm_f32_mfma<128><<<dim3((unsigned)(tiles_m * tiles_n)), blk, dyn, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (const float*)aux, (float*)C,
            (float*)C2, (int)M, (int)N, (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n);
    }
    return check_launch("hd_gemm_f32");
}

extern "C" int hd_gemm_f32(const void* A, const void* W, const void* bias, const void* resid, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi,
                           void* st) {
    HD_REQUIRE(epi >= 0 && epi <= 2, "hd_gemm_f32: bad epilogue");
    return hd_gemm_f32_ex(A, W, bias, resid, nullptr, C, nullptr, M, N, K, lda, ldw, ldc, epi, st);
}

extern "C" int hd_gemm_f64(const void* A, const void* W, const void* bias, void* C, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldw, int64_t ldc, void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "hd_gemm_f64: bad shape");
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && C, "hd_gemm_f64: null pointer");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "hd_gemm_f64: size exceeds int32");
    HD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "hd_gemm_f64: leading dimension too small");
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
    gemm_f64_valu<<<grid, 256, 0, (hipStream_t)st>>>((const double*)A, (const double*)W, (const double*)bias,
                                                     (double*)C, (int)M, (int)N, (int)K, (int)lda, (int)ldw, (int)ldc);
    return check_launch("hd_gemm_f64");
}
