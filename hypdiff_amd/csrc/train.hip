// Backward-pass kernels of the denoiser training step (models/direction_diffusion.py:335-363,
// ddpm.py:71-85 followed by loss.backward() in diff.py:113-118 / diff_graphset.py:64-75).
//
//  * gemm_tn_f32_mfma: weight gradients dW[P,Q] = sum_r A[r,P] * B[r,Q] (A = dZ, B = layer
//    input).  The reduction runs over the ROW index of both operands (tens of thousands of
//    latent rows) and the output is small, so the rows are split into slices (one workgroup per
//    slice and output tile) whose partial tiles reduce_slices sums in slice order (bit-reproducible,
//    no atomics).  Exact-f32 matrix cores as in gemm.hip; slabs of 32 rows are staged in LDS in
//    their natural [r][p] layout: the MFMA operand of lane (i, kk) is sA[2s+kk][i], i.e. 32
//    consecutive floats per half-wave (conflict-free ds_read_b32; at 64 cycles per
//    v_mfma_f32_32x32x2_f32 the LDS reads are nowhere near the critical path).
//  * transpose_f32: W -> W^T for the input-gradient GEMM (weights only: < 1 M elements).
//  * colsum_f32: bias gradients, two-stage column sums.
//  * mse kernels: F.mse_loss forward (difference + sum of squares) and its gradient.
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TP = 128, TQ = 128, TR = 32;

// Workgroup -> (row slice, output tile).  The hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs; every XCD gets a contiguous run of logical indices, i.e. whole
// row slices: all output tiles of a slice then run on ONE XCD and stream the same rows of A
// and B in step, so each operand row leaves HBM once and is shared through that XCD's L2
// (without this every XCD pulled all of A and B: 1.3 GB of fabric traffic per call, 39 TF).
__global__ __launch_bounds__(256) void gemm_tn_f32_mfma(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ C, int R, int P, int Q, int lda, int ldb,
                                                        int ldc, int chunk, int tiles_q, int tiles, int total,
                                                        size_t slice_stride) {
    __shared__ __attribute__((aligned(16))) float sA[TR * TP];
    __shared__ __attribute__((aligned(16))) float sB[TR * TQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int logical = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = logical & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (logical >> 3);
    }
    const int slice = logical / tiles, tile = logical - slice * tiles;
    const int tp = tile / tiles_q, tq = tile - tp * tiles_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int r_begin = slice * chunk;
    const int r_end = r_begin + chunk < R ? r_begin + chunk : R;

    // staging: each slab is 32 rows x 128 floats per operand -> 4 float4 per thread and operand
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const bool a_in = p0 + scol < P, b_in = q0 + scol < Q;      // P, Q multiples of 4
    const float* pa = A + (a_in ? p0 + scol : 0);
    const float* pb = B + (b_in ? q0 + scol : 0);
    float4 ra[4], rb[4];

    // Straight-line register staging as in gemm.hip (selects on whole float4 temporaries made
    // hipcc park the staging registers in scratch memory): loads are unconditional with the
    // row clamped into the slice; rows past its end and columns past P / Q are zeroed
    // component-wise when the registers go to LDS one iteration later.
#define HD_TN_LOAD(r0)                                                                       \
    {                                                                                        \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                      \
            int r = (r0) + srow + 8 * p;                                                     \
            r = r < r_end ? r : r_end - 1;                                                   \
            ra[p] = *reinterpret_cast<const float4*>(pa + (size_t)r * lda);                  \
            rb[p] = *reinterpret_cast<const float4*>(pb + (size_t)r * ldb);                  \
        }                                                                                    \
    }
#define HD_TN_STORE(r0)                                                                      \
    {                                                                                        \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                      \
            const bool rin = (r0) + srow + 8 * p < r_end;                                    \
            const bool oa = rin && a_in, ob = rin && b_in;                                   \
            float4 v = ra[p];                                                                \
            v.x = oa ? v.x : 0.f; v.y = oa ? v.y : 0.f; v.z = oa ? v.z : 0.f; v.w = oa ? v.w : 0.f; \
            *reinterpret_cast<float4*>(&sA[(srow + 8 * p) * TP + scol]) = v;                 \
            float4 u = rb[p];                                                                \
            u.x = ob ? u.x : 0.f; u.y = ob ? u.y : 0.f; u.z = ob ? u.z : 0.f; u.w = ob ? u.w : 0.f; \
            *reinterpret_cast<float4*>(&sB[(srow + 8 * p) * TQ + scol]) = u;                 \
        }                                                                                    \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    // a wavefront whose 64x64 quadrant lies entirely beyond P or Q (ragged last tile row /
    // column) skips its MFMAs; it still stages and synchronises
    const bool wave_live = (p0 + wr * 64 < P) && (q0 + wc * 64 < Q);
    const float* fpa = &sA[fh * TP + wr * 64 + fi];
    const float* fpb = &sB[fh * TQ + wc * 64 + fi];
    if (r_begin < r_end) {
        HD_TN_LOAD(r_begin)
        if (!wave_live) {       // staging-only copy of the loop (same barrier count), see gemm.hip
            for (int r0 = r_begin; r0 < r_end; r0 += TR) {
                HD_TN_STORE(r0)
                __syncthreads();
                HD_TN_LOAD(r0 + TR)
                __syncthreads();
            }
            return;
        }
        for (int r0 = r_begin; r0 < r_end; r0 += TR) {
            HD_TN_STORE(r0)
            __syncthreads();
            HD_TN_LOAD(r0 + TR)                                   // past the end: clamped re-read, zeroed
            __builtin_amdgcn_sched_barrier(0);
            // all fragments of the slab first (64 registers), then 64 MFMAs: the LDS latency of
            // step s+1.. hides behind the matrix pipe instead of stalling every group of four
            float fa[TR / 2][2], fb[TR / 2][2];
#pragma unroll
            for (int s = 0; s < TR / 2; ++s) {
                fa[s][0] = fpa[2 * s * TP];
                fa[s][1] = fpa[2 * s * TP + 32];
                fb[s][0] = fpb[2 * s * TQ];
                fb[s][1] = fpb[2 * s * TQ + 32];
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of the MFMAs (hipcc re-serialises them)
#pragma unroll
            for (int s = 0; s < TR / 2; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0], fb[s][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1], fb[s][1], acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }
#undef HD_TN_LOAD
#undef HD_TN_STORE

    float* out = C + (size_t)slice * slice_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cq = q0 + wc * 64 + j * 32 + fi;
        if (cq >= Q) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rp = p0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (rp < P) out[(size_t)rp * ldc + cq] = acc[i][j][r];
            }
    }
}

// out[i] = sum_s ws[s * stride + i], s ascending (fixed order)
__global__ __launch_bounds__(256) void reduce_slices(const float* __restrict__ ws, int S, size_t stride, size_t count,
                                                     float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float s = ws[i];
        for (int k = 1; k < S; ++k) s += ws[(size_t)k * stride + i];
        out[i] = s;
    }
}

__global__ __launch_bounds__(256) void transpose_f32(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                     int cols, int ld_in, int ld_out) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int r = ty; r < 64; r += 4)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(size_t)(r0 + r) * ld_in + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * ld_out + r0 + tx] = tile[tx][c];
}

// every W^T of a backward pass in ONE launch (24 weights of the graph-level DDPM: 23 launches of 8 us otherwise)
constexpr int kMaxTransposeJobs = 64;
struct TransposeJobs {
    int n;
    const float* in[kMaxTransposeJobs];
    float* out[kMaxTransposeJobs];
    int rows[kMaxTransposeJobs], cols[kMaxTransposeJobs], ld_in[kMaxTransposeJobs], ld_out[kMaxTransposeJobs];
    unsigned block0[kMaxTransposeJobs];
};

__global__ __launch_bounds__(256) void transpose_multi_f32(const TransposeJobs jobs) {
    __shared__ float tile[64][65];
    int k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.block0[k + 1]) ++k;
    const float* __restrict__ in = jobs.in[k];
    float* __restrict__ out = jobs.out[k];
    const int rows = jobs.rows[k], cols = jobs.cols[k], ld_in = jobs.ld_in[k], ld_out = jobs.ld_out[k];
    const int tiles_c = (cols + 63) >> 6, b = (int)(blockIdx.x - jobs.block0[k]);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = (b / tiles_c) * 64, c0 = (b % tiles_c) * 64;
    for (int r = ty; r < 64; r += 4)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[(size_t)(r0 + r) * ld_in + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) out[(size_t)(c0 + c) * ld_out + r0 + tx] = tile[tx][c];
}

// partial column sums: block = 64 columns x 4 row lanes, gridDim.y row chunks -> ws[chunk][N]
__global__ __launch_bounds__(256) void colsum_partial(const float* __restrict__ x, int M, int N, int ld, int chunk,
                                                      float* __restrict__ ws) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r_begin = blockIdx.y * chunk, r_end = r_begin + chunk < M ? r_begin + chunk : M;
    float s = 0.f;
    if (c < N)
        for (int r = r_begin + ty; r < r_end; r += 4) s += x[(size_t)r * ld + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) ws[(size_t)blockIdx.y * N + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

// d = a - b and per-block partial sums of d^2 (fp64)
__global__ __launch_bounds__(256) void mse_diff(const float* __restrict__ a, const float* __restrict__ b,
                                                float* __restrict__ d, size_t count, double* __restrict__ partial) {
    __shared__ double sm[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float v = a[i] - b[i];
        d[i] = v;
        s += (double)v * (double)v;
    }
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void mse_final(const double* __restrict__ partial, int n, double count,
                                                 float* __restrict__ loss) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((sm[0] + sm[1]) + (sm[2] + sm[3])) / count);
}

// out = d * (factor * g[0])
__global__ __launch_bounds__(256) void scale_by_dev(const float* __restrict__ d, const float* __restrict__ g,
                                                    float factor, float* __restrict__ out, size_t count) {
    const float f = factor * g[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) out[i] = d[i] * f;
}

// GCNConv tail (torch_geometric GCNConv: out = A_norm (x W^T) + bias, then F.gelu --
// models/direction_diffusion.py:28-37): out = gelu(x + bias[col]); z (optional) = x + bias
__global__ __launch_bounds__(256) void bias_gelu(const float* __restrict__ x, const float* __restrict__ bias,
                                                 float* __restrict__ out, float* __restrict__ z, size_t rows, int dim,
                                                 int act) {
    const size_t count = rows * (size_t)dim;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float v = x[i] + (bias ? bias[i % dim] : 0.f);
        if (z) z[i] = v;
        out[i] = act ? gelu_erf(v) : v;
    }
}
// dz = g * gelu'(z)
__global__ __launch_bounds__(256) void gelu_bwd(const float* __restrict__ g, const float* __restrict__ z,
                                                float* __restrict__ dz, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        dz[i] = g[i] * gelu_erf_grad(z[i]);
}

inline int pick_splits(int64_t R, int64_t tiles) {
    // about 4 workgroups per CU on 256 CUs, a multiple of 8 slices (whole slices per XCD),
    // every slice at least 64 rows
    int64_t s = (1024 / tiles) / 8 * 8;
    if (s < 8) s = 8;
    const int64_t smax = R / 64;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

constexpr int kMsePartials = 1024;

}  // namespace

extern "C" int64_t hd_gemm_tn_workspace_floats(int64_t R, int64_t P, int64_t Q) {
    if (R < 1 || P < 1 || Q < 1) return 0;
    const int64_t tiles = ((P + TP - 1) / TP) * ((Q + TQ - 1) / TQ);
    const int s = pick_splits(R, tiles);
    return s > 1 ? (int64_t)s * P * Q : 0;
}

extern "C" int hd_gemm_tn_f32(const void* A, const void* B, void* C, int64_t R, int64_t P, int64_t Q, int64_t lda,
                              int64_t ldb, int64_t ldc, void* workspace, void* st) {
    HD_REQUIRE(R >= 0 && P >= 1 && Q >= 1, "hd_gemm_tn_f32: bad shape R=%lld P=%lld Q=%lld", (long long)R,
               (long long)P, (long long)Q);
    HD_REQUIRE(A && B && C, "hd_gemm_tn_f32: null pointer");
    HD_REQUIRE(R < (1LL << 31) && P < (1LL << 31) && Q < (1LL << 31), "hd_gemm_tn_f32: size exceeds int32");
    HD_REQUIRE(P % 4 == 0 && Q % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && is_aligned16(A) && is_aligned16(B),
               "hd_gemm_tn_f32: P, Q, lda, ldb must be multiples of 4 and A, B 16-byte aligned (pad with zeros)");
    HD_REQUIRE(lda >= P && ldb >= Q && ldc >= Q, "hd_gemm_tn_f32: leading dimension too small");
    hipStream_t s = (hipStream_t)st;
    if (R == 0) {
        HD_REQUIRE(ldc == Q, "hd_gemm_tn_f32: empty reduction needs a contiguous output");
        if (hipMemsetAsync(C, 0, (size_t)P * Q * sizeof(float), s) != hipSuccess) return check_launch("hd_gemm_tn_f32");
        return HD_OK;
    }
    const int tiles_p = (int)((P + TP - 1) / TP), tiles_q = (int)((Q + TQ - 1) / TQ);
    const int splits = pick_splits(R, (int64_t)tiles_p * tiles_q);
    int chunk = (int)((R + splits - 1) / splits);
    chunk = (chunk + TR - 1) / TR * TR;
    const int nsl = (int)((R + chunk - 1) / chunk);
    if (nsl > 1) {
        HD_REQUIRE(workspace, "hd_gemm_tn_f32: workspace of hd_gemm_tn_workspace_floats() floats required");
        HD_REQUIRE(ldc == Q, "hd_gemm_tn_f32: split reduction needs a contiguous output (ldc == Q)");
        gemm_tn_f32_mfma<<<dim3((unsigned)(tiles_p * tiles_q * nsl)), 256, 0, s>>>(
            (const float*)A, (const float*)B, (float*)workspace, (int)R, (int)P, (int)Q, (int)lda, (int)ldb, (int)Q, chunk,
            tiles_q, tiles_p * tiles_q, tiles_p * tiles_q * nsl, (size_t)P * Q);
        const size_t count = (size_t)P * Q;
        int blocks = (int)((count + 255) / 256);
        blocks = blocks < 2048 ? blocks : 2048;
        reduce_slices<<<blocks, 256, 0, s>>>((const float*)workspace, nsl, count, count, (float*)C);
    } else {
        gemm_tn_f32_mfma<<<dim3((unsigned)(tiles_p * tiles_q)), 256, 0, s>>>(
            (const float*)A, (const float*)B, (float*)C, (int)R, (int)P, (int)Q, (int)lda, (int)ldb, (int)ldc, chunk,
            tiles_q, tiles_p * tiles_q, tiles_p * tiles_q, 0);
    }
    return check_launch("hd_gemm_tn_f32");
}

extern "C" int hd_transpose_f32(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                                void* st) {
    HD_REQUIRE(rows >= 0 && cols >= 0 && ld_in >= cols && ld_out >= rows, "hd_transpose_f32: bad shape");
    if (rows == 0 || cols == 0) return HD_OK;
    HD_REQUIRE(in && out && in != out, "hd_transpose_f32: null or aliased pointer");
    HD_REQUIRE(rows < (1LL << 31) && cols < (1LL << 31), "hd_transpose_f32: size exceeds int32");
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    HD_REQUIRE(grid.y < 65536, "hd_transpose_f32: too many row tiles");
    transpose_f32<<<grid, 256, 0, (hipStream_t)st>>>((const float*)in, (float*)out, (int)rows, (int)cols, (int)ld_in,
                                                      (int)ld_out);
    return check_launch("hd_transpose_f32");
}

extern "C" int hd_transpose_multi_f32(const hd_transpose_job* jobs, int n_jobs, void* st) {
    HD_REQUIRE(n_jobs >= 0 && n_jobs <= kMaxTransposeJobs && (jobs || n_jobs == 0),
               "hd_transpose_multi_f32: between 0 and %d jobs per call", kMaxTransposeJobs);
    TransposeJobs tj;
    tj.n = 0;
    uint64_t blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const hd_transpose_job& j = jobs[i];
        HD_REQUIRE(j.rows >= 0 && j.cols >= 0 && j.ld_in >= j.cols && j.ld_out >= j.rows, "hd_transpose_multi_f32: job %d: bad shape", i);
        if (j.rows == 0 || j.cols == 0) continue;
        HD_REQUIRE(j.in && j.out && j.in != j.out, "hd_transpose_multi_f32: job %d: null or aliased pointer", i);
        HD_REQUIRE(j.rows < (1LL << 31) && j.cols < (1LL << 31) && j.ld_in < (1LL << 31) && j.ld_out < (1LL << 31),
                   "hd_transpose_multi_f32: job %d: size exceeds int32", i);
        const int k = tj.n++;
        tj.in[k] = (const float*)j.in;
        tj.out[k] = (float*)j.out;
        tj.rows[k] = (int)j.rows;
        tj.cols[k] = (int)j.cols;
        tj.ld_in[k] = (int)j.ld_in;
        tj.ld_out[k] = (int)j.ld_out;
        tj.block0[k] = (unsigned)blocks;
        blocks += (uint64_t)((j.rows + 63) / 64) * ((j.cols + 63) / 64);
        HD_REQUIRE(blocks < (1ULL << 31), "hd_transpose_multi_f32: too many workgroups");
    }
    if (tj.n == 0) return HD_OK;
    transpose_multi_f32<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)st>>>(tj);
    return check_launch("hd_transpose_multi_f32");
}

extern "C" int64_t hd_colsum_workspace_floats(int64_t M, int64_t N) {
    if (M < 1 || N < 1) return 0;
    int64_t chunks = (M + 255) / 256;
    chunks = chunks < 64 ? chunks : 64;
    return chunks * N;
}

extern "C" int hd_colsum_f32(const void* x, void* out, int64_t M, int64_t N, int64_t ld, void* workspace, void* st) {
    HD_REQUIRE(M >= 0 && N >= 1 && ld >= N, "hd_colsum_f32: bad shape");
    HD_REQUIRE(out, "hd_colsum_f32: null pointer");
    hipStream_t s = (hipStream_t)st;
    if (M == 0) {
        if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s) != hipSuccess) return check_launch("hd_colsum_f32");
        return HD_OK;
    }
    HD_REQUIRE(x && workspace, "hd_colsum_f32: null pointer");
    HD_REQUIRE(M < (1LL << 31) && N < (1LL << 31), "hd_colsum_f32: size exceeds int32");
    int64_t chunks = (M + 255) / 256;
    chunks = chunks < 64 ? chunks : 64;
    const int chunk = (int)((M + chunks - 1) / chunks);
    const int nch = (int)((M + chunk - 1) / chunk);
    colsum_partial<<<dim3((unsigned)((N + 63) / 64), (unsigned)nch), 256, 0, s>>>((const float*)x, (int)M, (int)N, (int)ld,
                                                                                 chunk, (float*)workspace);
    int blocks = (int)((N + 255) / 256);
    reduce_slices<<<blocks, 256, 0, s>>>((const float*)workspace, nch, (size_t)N, (size_t)N, (float*)out);
    return check_launch("hd_colsum_f32");
}

extern "C" int64_t hd_mse_workspace_doubles(void) { return kMsePartials; }

extern "C" int hd_mse_forward_f32(const void* a, const void* b, void* diff, int64_t count, void* workspace, void* loss,
                                  void* st) {
    HD_REQUIRE(count >= 1, "hd_mse_forward_f32: empty input");
    HD_REQUIRE(a && b && diff && workspace && loss, "hd_mse_forward_f32: null pointer");
    int64_t blocks = (count + 255) / 256;
    blocks = blocks < kMsePartials ? blocks : kMsePartials;
    hipStream_t s = (hipStream_t)st;
    mse_diff<<<(unsigned)blocks, 256, 0, s>>>((const float*)a, (const float*)b, (float*)diff, (size_t)count,
                                              (double*)workspace);
    mse_final<<<1, 256, 0, s>>>((const double*)workspace, (int)blocks, (double)count, (float*)loss);
    return check_launch("hd_mse_forward_f32");
}

extern "C" int hd_mse_backward_f32(const void* diff, const void* grad_loss, void* grad_out, int64_t count, void* st) {
    HD_REQUIRE(count >= 1, "hd_mse_backward_f32: empty input");
    HD_REQUIRE(diff && grad_loss && grad_out, "hd_mse_backward_f32: null pointer");
    int64_t blocks = (count + 255) / 256;
    blocks = blocks < 4096 ? blocks : 4096;
    scale_by_dev<<<(unsigned)blocks, 256, 0, (hipStream_t)st>>>((const float*)diff, (const float*)grad_loss,
                                                                (float)(2.0 / (double)count), (float*)grad_out,
                                                                (size_t)count);
    return check_launch("hd_mse_backward_f32");
}

extern "C" int hd_bias_gelu_f32(const void* x, const void* bias, void* out, void* z, int64_t rows, int64_t dim, int act,
                                void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim < (1LL << 31), "hd_bias_gelu_f32: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out, "hd_bias_gelu_f32: null pointer");
    int64_t blocks = (rows * dim + 255) / 256;
    blocks = blocks < 4096 ? blocks : 4096;
    bias_gelu<<<(unsigned)blocks, 256, 0, (hipStream_t)st>>>((const float*)x, (const float*)bias, (float*)out, (float*)z,
                                                             (size_t)rows, (int)dim, act);
    return check_launch("hd_bias_gelu_f32");
}

extern "C" int hd_gelu_bwd_f32(const void* g, const void* z, void* dz, int64_t count, void* st) {
    HD_REQUIRE(count >= 0, "hd_gelu_bwd_f32: bad size");
    if (count == 0) return HD_OK;
    HD_REQUIRE(g && z && dz, "hd_gelu_bwd_f32: null pointer");
    int64_t blocks = (count + 255) / 256;
    blocks = blocks < 4096 ? blocks : 4096;
    gelu_bwd<<<(unsigned)blocks, 256, 0, (hipStream_t)st>>>((const float*)g, (const float*)z, (float*)dz, (size_t)count);
    return check_launch("hd_gelu_bwd_f32");
}

// ---- Adam over a flat gradient bucket (optim.Adam(diff.parameters(), lr) of diff.py:107, diff_graphset.py:66) --------
// One launch for every parameter tensor of the denoiser (torch's foreach Adam takes 7 multi-tensor launches, 120 us for
// 2.8 M parameters; this is one pass over p, g, m, v at HBM speed).  Arithmetic in the order of torch.optim.Adam's
// default path: m += (g - m)(1 - b1); v = v b2 + (1 - b2) g g; p += -(lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps)).
namespace {
constexpr int kMaxAdamJobs = 128;
struct AdamJobs {
    int n;
    float* p[kMaxAdamJobs];
    unsigned off[kMaxAdamJobs], count[kMaxAdamJobs], block0[kMaxAdamJobs];
};

// step counter on the device (a captured HIP graph replays the same kernel arguments): state = {t, -lr / (1 - b1^t),
// sqrt(1 - b2^t)} as doubles; one thread advances it ahead of adam_multi on the same stream
__global__ void adam_tick(double* __restrict__ state, double lr, double beta1, double beta2) {
    const double t = state[0] + 1.0;
    state[0] = t;
    state[1] = -(lr / (1.0 - pow(beta1, t)));
    state[2] = sqrt(1.0 - pow(beta2, t));
}

__global__ __launch_bounds__(256) void adam_multi(const AdamJobs jobs, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, float w1, float b2, float w2, float neg_step,
                                                  float bc2_sqrt, float eps, const double* __restrict__ state) {
    if (state) {
        neg_step = (float)state[1];
        bc2_sqrt = (float)state[2];
    }
    int k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.block0[k + 1]) ++k;
    float* __restrict__ p = jobs.p[k];
    const unsigned off = jobs.off[k], count = jobs.count[k];
    const unsigned base = (blockIdx.x - jobs.block0[k]) * 1024u + threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned i = base + e * 256u;
        if (i >= count) break;
        const float gi = g[off + i];
        float mi = m[off + i], vi = v[off + i];
        mi = mi + w1 * (gi - mi);
        vi = vi * b2 + w2 * gi * gi;
        m[off + i] = mi;
        v[off + i] = vi;
        p[i] = p[i] + neg_step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}
}  // namespace

extern "C" int hd_adam_step_f32(const hd_adam_job* jobs, int n_jobs, const void* grad_flat, void* exp_avg_flat,
                                void* exp_avg_sq_flat, double lr, double beta1, double beta2, double eps, int64_t step,
                                void* step_state, void* st) {
    HD_REQUIRE(n_jobs >= 0 && n_jobs <= kMaxAdamJobs && (jobs || n_jobs == 0), "hd_adam_step_f32: between 0 and %d jobs per call",
               kMaxAdamJobs);
    HD_REQUIRE((step >= 1 || step_state) && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
               "hd_adam_step_f32: bad hyper-parameters (step counts from 1)");
    AdamJobs aj;
    aj.n = 0;
    uint64_t blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (jobs[i].count == 0) continue;
        HD_REQUIRE(jobs[i].param && jobs[i].count > 0 && jobs[i].offset >= 0 && jobs[i].offset + jobs[i].count < (1LL << 32),
                   "hd_adam_step_f32: job %d: null parameter or range beyond 2^32 elements", i);
        aj.p[aj.n] = (float*)jobs[i].param;
        aj.off[aj.n] = (unsigned)jobs[i].offset;
        aj.count[aj.n] = (unsigned)jobs[i].count;
        aj.block0[aj.n] = (unsigned)blocks;
        blocks += (uint64_t)(jobs[i].count + 1023) / 1024;
        ++aj.n;
    }
    if (aj.n == 0) return HD_OK;
    HD_REQUIRE(grad_flat && exp_avg_flat && exp_avg_sq_flat, "hd_adam_step_f32: null pointer");
    HD_REQUIRE(blocks < (1ULL << 31), "hd_adam_step_f32: too many workgroups");
    // step_state: the count lives on the device; step > 0 there means "advance it now" (the first group of jobs of a step)
    if (step_state && step > 0) adam_tick<<<1, 1, 0, (hipStream_t)st>>>((double*)step_state, lr, beta1, beta2);
    const double tt = step >= 1 ? (double)step : 1.0;
    const double bc1 = 1.0 - pow(beta1, tt), bc2 = 1.0 - pow(beta2, tt);
    adam_multi<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st>>>(
        aj, (const float*)grad_flat, (float*)exp_avg_flat, (float*)exp_avg_sq_flat, (float)(1.0 - beta1), (float)beta2,
        (float)(1.0 - beta2), (float)(-(lr / bc1)), (float)sqrt(bc2), (float)eps, (const double*)step_state);
    return check_launch("hd_adam_step_f32");
}
