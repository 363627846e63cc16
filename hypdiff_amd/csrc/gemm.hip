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
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, PITCH = BK + 4;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// epi: 0 none, 1 gelu, 2 residual + gelu (ddpm.py:20-22: x0 + gelu(layer(x0)))
template <int BNT>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const float* __restrict__ A, const float* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ resid, float* __restrict__ C,
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

    auto gload = [&](int k0) {
        const bool kin = (k0 + skq) < K;
#pragma unroll
        for (int p = 0; p < BM / 32; ++p) {
            int r = m0 + srow + 32 * p;
            r = r < M ? r : M - 1;
            ra[p] = kin ? *reinterpret_cast<const float4*>(A + (size_t)r * lda + k0 + skq) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < BNT / 32; ++p) {
            int r = n0 + srow + 32 * p;
            r = r < N ? r : N - 1;
            rw[p] = kin ? *reinterpret_cast<const float4*>(W + (size_t)r * ldw + k0 + skq) : make_float4(0, 0, 0, 0);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int p = 0; p < BM / 32; ++p)
            *reinterpret_cast<float4*>(&sA[(srow + 32 * p) * PITCH + skq]) = ra[p];
#pragma unroll
        for (int p = 0; p < BNT / 32; ++p)
            *reinterpret_cast<float4*>(&sW[(srow + 32 * p) * PITCH + skq]) = rw[p];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    const int nk = (K + BK - 1) / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        sstore();
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * BK);
        float4 fa[2][4], fw[NT][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* pa = &sA[(wr * 64 + i * 32 + fi) * PITCH + fh * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) fa[i][q] = *reinterpret_cast<const float4*>(pa + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float* pw = &sW[(wc * (BNT / 2) + j * 32 + fi) * PITCH + fh * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) fw[j][q] = *reinterpret_cast<const float4*>(pw + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float a = reinterpret_cast<const float*>(&fa[i][q])[e];
                        const float b = reinterpret_cast<const float*>(&fw[j][q])[e];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

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
                    float v = acc[i][j][r] + bv;
                    if (epi == 1) v = gelu_erf(v);
                    else if (epi == 2) v = resid[(size_t)rm * ldc + cn] + gelu_erf(v);
                    C[(size_t)rm * ldc + cn] = v;
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

extern "C" int hd_gemm_f32(const void* A, const void* W, const void* bias, const void* resid, void* C,
                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi,
                           void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && K >= 1, "hd_gemm_f32: bad shape M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
    if (M == 0) return HD_OK;
    HD_REQUIRE(A && W && C, "hd_gemm_f32: null pointer");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "hd_gemm_f32: size exceeds int32");
    HD_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && is_aligned16(A) && is_aligned16(W),
               "hd_gemm_f32: K, lda, ldw must be multiples of 4 and A, W 16-byte aligned (pad with zeros)");
    HD_REQUIRE(lda >= K && ldw >= K && ldc >= N, "hd_gemm_f32: leading dimension too small");
    HD_REQUIRE(epi >= 0 && epi <= 2 && (epi != 2 || resid), "hd_gemm_f32: bad epilogue / missing residual");
    const dim3 blk(256);
    // narrow block tile when it wastes fewer padded columns or is needed to fill 256 CUs
    const int64_t t128 = ((N + 127) / 128) * ((M + 127) / 128);
    const bool narrow = ((N + 63) / 64) * 64 < ((N + 127) / 128) * 128 || t128 < 256;
    const int tiles_m = (int)((M + BM - 1) / BM);
    if (narrow) {
        const int tiles_n = (int)((N + 63) / 64);
        gemm_f32_mfma<64><<<dim3((unsigned)(tiles_m * tiles_n)), blk, 0, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (float*)C, (int)M, (int)N,
            (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n);
    } else {
        const int tiles_n = (int)((N + 127) / 128);
        gemm_f32_mfma<128><<<dim3((unsigned)(tiles_m * tiles_n)), blk, 0, (hipStream_t)st>>>(
            (const float*)A, (const float*)W, (const float*)bias, (const float*)resid, (float*)C, (int)M, (int)N,
            (int)K, (int)lda, (int)ldw, (int)ldc, epi, tiles_m, tiles_n);
    }
    return check_launch("hd_gemm_f32");
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
