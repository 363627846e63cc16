// Weight gradient dW = dz^T @ x on the bf16 matrix cores, fp32-faithful (3 bf16 limbs per operand element, 6 exact
// limb products per fp32 product, as the limb GEMM of gemm_limbs.hip).  Autograd of diff.py:113-118 /
// diff_graphset.py:64-75 over models/direction_diffusion.py:56-69, ddpm.py:9-41; the reference has no kernel of its
// own here.  C[P][Q] = sum_r A[r][p] B[r][q]: the contraction index r is the ROW index of both operands, while a lane
// of v_mfma_f32_16x16x32_bf16 supplies 8 consecutive k of one row.  The transpose costs nothing extra: a thread
// loads the same 4 columns of 8 consecutive rows (8 float4 loads; a wavefront's loads are 512-byte runs of A and of
// B), so it holds 8 consecutive r for each of its 4 columns in registers, splits them into limbs and writes the 12
// operand chunks of 16 bytes straight in fragment order ([limb][kq][column], the slab layout of the limb images).
// 128x128 tile, 4 wavefronts of 64x64, ONE LDS buffer of 48 KB (a slab's 24 fragments are in registers before its
// multiplies start; the next slab's rows are already in flight into registers), two workgroups per CU.  The rows are
// cut into slices (one workgroup per tile and slice, partial tiles summed in slice order by a second kernel:
// bit-reproducible, no atomics), as in hd_gemm_tn_f32, which stays the exact-fp32 form.  The bias gradient
// db[p] = sum_r A[r][p] rides along: the workgroups of the first tile column add up the fp32 rows they stage anyway
// (registers, then the four wavefronts in order, then the slices in order), which replaces hd_colsum_f32's two launches.
#include <stdlib.h>
#include "hd_common.cuh"
#include "hd_limbs.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

constexpr int TP = 128, TQ = 128, TR = 32;

inline int tn_target_wgs() {
    static const int v = [] { const char* e = getenv("HD_TN_WGS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 512; }();
    return v;
}

inline int host_slices(int64_t R, int64_t tiles) {
    // one resident round: 256 CUs x 2 workgroups (212 VGPRs, 48 KB LDS); slices of at least 2 slabs (a 200-row batch
    // is 7 slabs: four slices instead of one workgroup per tile)
    int64_t s = tn_target_wgs() / tiles;
    const int64_t max_s = (R + 2 * TR - 1) / (2 * TR);
    s = s < 1 ? 1 : (s > max_s ? max_s : s);
    return (int)(s > 64 ? 64 : s);
}

__global__ __launch_bounds__(256) void gemm_tn_bf16x3(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ out, int R, int P, int Q, int lda, int ldb,
                                                      int ldo, int chunk, int tiles_q, int tiles, int total, size_t slice_stride,
                                                      float* __restrict__ cs_out) {
    __shared__ uint4 lds[2 * 3 * 4 * 128];                       // [operand][limb][kq][column]: 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup b runs on XCD b % 8: give each XCD a contiguous run of (slice, tile) pairs, so that the tiles that share
    // a slice's rows of A and B meet in one L2
    const int per_xcd = (gridDim.x + 7) >> 3;
    const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vid >= total) return;                                     // uniform: the grid is rounded up to a multiple of 8
    const int tile = vid % tiles, slice = vid / tiles;
    const int tp = tile / tiles_q, tq = tile - tp * tiles_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int r_begin = slice * chunk, r_end = r_begin + chunk < R ? r_begin + chunk : R;
    const int nslab = (r_end - r_begin + TR - 1) / TR;
    // staging role of this thread: rows 8 kq .. 8 kq + 7 of the slab (kq = wave), 4 columns of A (lanes 0..31) or of B
    const int kq_s = wave, half = lane >> 5, c4 = (lane & 31) * 4;
    const float* src = half ? B : A;
    const int ld = half ? ldb : lda;
    const int col = (half ? q0 : p0) + c4;
    const bool col_ok = col < (half ? Q : P);                    // P, Q multiples of 4: a quad is inside or outside
    float4 pre[8];
    const bool do_cs = cs_out != nullptr && tq == 0;             // uniform over the workgroup
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_slab = [&](int t) {
        const int rb = r_begin + t * TR + kq_s * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = rb + e;
            pre[e] = (col_ok && r < r_end) ? *reinterpret_cast<const float4*>(src + (size_t)r * ld + col)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_slab = [&]() {
        uint4* base = lds + half * (3 * 4 * 128) + kq_s * 128 + c4;
        if (do_cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs.x += pre[e].x; cs.y += pre[e].y; cs.z += pre[e].z; cs.w += pre[e].w; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = c == 0 ? pre[e].x : (c == 1 ? pre[e].y : (c == 2 ? pre[e].z : pre[e].w));
            uint4 h, m, l;
            limb_split8(v, h, m, l);
            base[c] = h;
            base[4 * 128 + c] = m;
            base[8 * 128 + c] = l;
        }
    };
    // operand maps of v_mfma_f32_16x16x32: lane (r16, kq) supplies column r16 of its 16-column block, k = 8 kq .. + 7
    const int r16 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int a_off = kq * 128 + wr * 64 + r16, b_off = 3 * 4 * 128 + kq * 128 + wc * 64 + r16;
    lf32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = lf32x4{0.f, 0.f, 0.f, 0.f};
    if (nslab > 0) load_slab(0);
    for (int t = 0; t < nslab; ++t) {
        store_slab();
        __syncthreads();                                          // the slab is in LDS
        uint4 fa[4][3], fb[4][3];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][l] = lds[a_off + l * 4 * 128 + i * 16];
                fb[i][l] = lds[b_off + l * 4 * 128 + i * 16];
            }
        __syncthreads();                                          // every wavefront holds its fragments: buffer free
        if (t + 1 < nslab) load_slab(t + 1);                      // in flight during the multiplies
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lf32x4 c = acc[i][j];
                // first operand = B columns (q), second = A columns (p): D column = lane & 15 = p, D rows 4 kq + reg = q
#define HD_MM(bl, al) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, fb[j][bl]), \
                                                          __builtin_bit_cast(lbf16x8, fa[i][al]), c, 0, 0, 0);
                HD_MM(0, 2) HD_MM(2, 0) HD_MM(1, 1) HD_MM(0, 1) HD_MM(1, 0) HD_MM(0, 0)
#undef HD_MM
                acc[i][j] = c;
            }
    }
    if (do_cs) {                                                  // the last slab's fragments are out of LDS (barrier above)
        float* red = reinterpret_cast<float*>(lds);
        if (half == 0) *reinterpret_cast<float4*>(red + wave * 128 + c4) = cs;
        __syncthreads();
        if (tid < 128 && p0 + tid < P)
            cs_out[(size_t)slice * P + p0 + tid] = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
    }
    float* o = out + (size_t)slice * slice_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + wr * 64 + i * 16 + r16;
        if (p >= P) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + wc * 64 + j * 16 + 4 * kq;
            if (q < Q)                                            // Q % 4 == 0: the quad is inside or outside
                *reinterpret_cast<float4*>(o + (size_t)p * ldo + q) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// The same product from the limb IMAGES that the forward and input-gradient launches write anyway (bf16x3,
// [row block 128][slab 32][limb][kc][row] chunks of 8 columns of one row): no fp32 copies of dz or of the layer inputs
// exist in training, and nothing is split here.  The contraction runs over image ROWS, while a chunk holds 8 COLUMNS of
// one row -- the transpose is gfx950's ds_read_b64_tr_b16: a slab of 32 rows goes to LDS by LDS-DMA as 48 sub-tiles
// [32 rows][16 columns] of 1 KB (operand, 32-column slab, limb, column half), and a 16-lane group reads a 4-row x
// 16-column block of one and receives it column-major, i.e. 4 consecutive k of its own column: two reads make one
// v_mfma_f32_16x16x32_bf16 operand.  A sub-tile holds the chunk of (row r, 8-column half h) at position
// 32 h + (r ^ 4 h): lanes 0..31 of the LDS-DMA fetch one 512-byte run of the image and lanes 32..63 another (whole
// 128-byte lines), and the 32 lanes of a half-wavefront of the transposing read (rows 8g..8g+3 of two groups, both
// halves, 8 bytes each) cover one 256-byte bank row: conflict-free.  (Round 3: the earlier [row][half] order, whose
// neighbouring lanes fetched from two runs 2 KB apart, measured the same to the microsecond; so did every slab served
// from the L2 (a timing build that re-read two slabs) and a start-up delay of 0.3-3 us for the second workgroup of a CU:
// the kernel runs 1.9-2.1 us per slab like the forward chain, and what separated five launches from that was ~22 us
// of fixed cost per launch -- ramp, 33 MB of partial tiles, the reduction launch -- hence the one-launch form below.)
// The pad rows of the last 128-row block must be ZERO in both images (hd_limb_pack*_f32 and the limb GEMM's write-out
// zero them).  The bias gradient is three extra multiplies per 16 columns with an all-ones operand in the first tile column.
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Frag8 { s16x4 lo, hi; };

// rows 8 kq .. + 3 of the sub-tile at `lo + imm`, rows 8 kq + 4 .. + 7 at `hi + imm`
__device__ __forceinline__ lbf16x8 tr_frag(const __attribute__((address_space(3))) char* lo,
                                           const __attribute__((address_space(3))) char* hi, int imm) {
    Frag8 f;
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(const_cast<__attribute__((address_space(3))) char*>(lo + imm)));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(const_cast<__attribute__((address_space(3))) char*>(hi + imm)));
    return __builtin_bit_cast(lbf16x8, f);
}

// one product C = A^T B cut into `tiles` 128 x 128 tiles and `nsl` row slices (whole slabs, dealt evenly): work item vid = slice * tiles + tile
struct TnJob {
    const uint4* Aimg;
    const uint4* Bimg;
    float* out;               // C itself (one slice) or the workspace of partial tiles
    float* cs_out;            // column sums of A (per slice), or null
    int R, P, Q, ldo, tiles_q, tiles, total;
    int slabs_q, slabs_rem;   // slice s covers slabs_q + (s < slabs_rem) slabs of 32 rows: slices differ by one slab at most
    size_t slice_stride;
};

// work item `vid` of job J by this workgroup; `lds` = 48 KB, free on entry (every wavefront holds no fragment of it) and on exit
__device__ __forceinline__ void tn_img_item(const TnJob& J, int vid, uint4* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = J.P, Q = J.Q, ldo = J.ldo;
    const int tile = vid % J.tiles, slice = vid / J.tiles;
    const int tp = tile / J.tiles_q, tq = tile - tp * J.tiles_q;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int nslab = J.slabs_q + (slice < J.slabs_rem ? 1 : 0);
    const int r_begin = (slice * J.slabs_q + (slice < J.slabs_rem ? slice : J.slabs_rem)) * TR;
    // staging: wavefront w brings sub-tiles 12 w .. 12 w + 11 (waves 0, 1: A; 2, 3: B); lane L fetches the chunk of column
    // half L >> 5, row (L & 31) ^ 4 (L >> 5) of the slab, into position L of the sub-tile
    const int op = wave >> 1;
    const uint4* img = op ? J.Bimg : J.Aimg;
    const int nsl_img = ((op ? Q : P) + 31) >> 5, s0 = (op ? q0 : p0) >> 5;
    const int half_l = lane >> 5;
    const int row_l = (lane & 31) ^ (4 * half_l);
    unsigned lane_off[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int rem = (wave & 1) * 12 + k;                     // (slab * 3 + limb) * 2 + column half, within the operand
        const int sl = rem / 6, l = (rem % 6) >> 1, kp = rem & 1;
        int sg = s0 + sl;
        sg = sg < nsl_img ? sg : nsl_img - 1;                    // beyond the image: any valid slab (columns never written out)
        lane_off[k] = (unsigned)(sg * 1536 + l * 512 + (2 * kp + half_l) * 128 + row_l);
    }
    auto stage = [&](int t) {
        const int r = r_begin + t * TR;
        const uint4* base = img + ((size_t)(r >> 7) * nsl_img * 1536 + (r & 127));
#pragma unroll
        for (int k = 0; k < 12; ++k) limb_glds16(base + lane_off[k], lds + (wave * 12 + k) * 64);
    };
    const int r16 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    // lane (rho = r16 >> 2, gamma = r16 & 3) reads columns 4 gamma .. + 3 of row 8 kq + rho (then + 4): half h = gamma >> 1,
    // position 32 h + (row ^ 4 h), 8 (gamma & 1) bytes into the chunk
    const int h_r = (r16 & 3) >> 1;
    const int lane_lo = 512 * h_r + 16 * ((8 * kq + (r16 >> 2)) ^ (4 * h_r)) + 8 * (r16 & 1);
    const int lane_hi = 512 * h_r + 16 * ((8 * kq + (r16 >> 2) + 4) ^ (4 * h_r)) + 8 * (r16 & 1);
    const auto* lbase = (const __attribute__((address_space(3))) char*)lds;
    const auto* a_lo = lbase + lane_lo + wr * 12288;
    const auto* a_hi = lbase + lane_hi + wr * 12288;
    const auto* b_lo = lbase + lane_lo + 24576 + wc * 12288;
    const auto* b_hi = lbase + lane_hi + 24576 + wc * 12288;
    lf32x4 acc[4][4], csa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        csa[i] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = lf32x4{0.f, 0.f, 0.f, 0.f};
    }
    // column sums of A (the bias gradient): the eight 16-column groups (wr, i) of a tile row are dealt over the wavefronts
    // (tq, wc) of that tile row that hold their fragments, so that no tile is 12 % longer than its neighbours (tiles that
    // share rows stay in phase in the L2); the sum of a column is the same sequence of products wherever it runs
    const int cs_slots = 2 * J.tiles_q, cs_mine = 2 * tq + wc;
    bool cs_own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs_own[i] = J.cs_out != nullptr && (wr * 4 + i) % cs_slots == cs_mine;   // wavefront-uniform
    const lbf16x8 ones = __builtin_bit_cast(lbf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    if (nslab > 0) stage(0);
    for (int t = 0; t < nslab; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // the slab is in LDS
        lbf16x8 fa[4][3], fb[4][3];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int imm = ((i >> 1) * 6 + l * 2 + (i & 1)) * 1024;
                fa[i][l] = tr_frag(a_lo, a_hi, imm);
                fb[i][l] = tr_frag(b_lo, b_hi, imm);
            }
        __syncthreads();                                          // every wavefront holds its fragments: buffer free
        if (t + 1 < nslab) stage(t + 1);                          // in flight during the multiplies
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lf32x4 c = acc[i][j];
#define HD_MM(bl, al) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][bl], fa[i][al], c, 0, 0, 0);
                HD_MM(0, 2) HD_MM(2, 0) HD_MM(1, 1) HD_MM(0, 1) HD_MM(1, 0) HD_MM(0, 0)
#undef HD_MM
                acc[i][j] = c;
            }
            if (cs_own[i]) {
                lf32x4 c = csa[i];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i][0], c, 0, 0, 0);
                csa[i] = c;
            }
        }
    }
    if (kq == 0) {                                                // every row of the ones product is the column sum
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + wr * 64 + i * 16 + r16;
            if (cs_own[i] && p < P) J.cs_out[(size_t)slice * P + p] = csa[i][0];
        }
    }
    float* o = J.out + (size_t)slice * J.slice_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + wr * 64 + i * 16 + r16;
        if (p >= P) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + wc * 64 + j * 16 + 4 * kq;
            if (q < Q)
                *reinterpret_cast<float4*>(o + (size_t)p * ldo + q) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_img(const TnJob J) {
    __shared__ __attribute__((aligned(1024))) uint4 lds[48 * 64];  // [operand][slab][limb][column half][32 rows x 16 columns]
    // workgroup b runs on XCD b % 8: every XCD a contiguous run of (slice, tile) pairs (shared rows meet in one L2)
    const int per_xcd = (gridDim.x + 7) >> 3;
    const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vid >= J.total) return;                                   // uniform
    tn_img_item(J, vid, lds);
}

// Several products that share the contraction rows (the five weight gradients of the denoiser's backward pass: they
// are independent of each other) in ONE launch of resident workgroups: workgroup b takes, job after job, the item
// vid_j = (v(b) - rot_j) mod grid of job j when the job has that many items, v(b) as above.  Jobs are cut exactly as
// hd_gemm_tn_limbs_f32 cuts them (same slices, same slab order, same reduction order: bit-identical results); rot_j
// is the running sum of the item counts, so the workgroups a job leaves idle are the first of the next one.  No
// workgroup waits for another: a job's items start as their workgroups finish the previous job's.
constexpr int kMaxTnJobs = 8;
struct TnMulti {
    TnJob job[kMaxTnJobs];
    int rot[kMaxTnJobs];
    int n;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_img_multi(const TnMulti M) {
    __shared__ __attribute__((aligned(1024))) uint4 lds[48 * 64];
    const int grid = (int)gridDim.x, per_xcd = grid >> 3;        // the host launches a multiple of 8 workgroups
    const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    for (int j = 0; j < M.n; ++j) {
        int vid = v - M.rot[j];
        vid += vid < 0 ? grid : 0;
        if (vid < M.job[j].total) tn_img_item(M.job[j], vid, lds);
    }
}

// out[i] = sum over slices, in slice order; the column sums (count2 of them per slice, 0 = none) follow the tiles
__global__ __launch_bounds__(256) void tn_reduce(const float* __restrict__ ws, int nsl, size_t count, float* __restrict__ out,
                                                 const float* __restrict__ ws2, size_t count2, float* __restrict__ out2) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count + count2; i += (size_t)gridDim.x * 256 * 4) {
        const bool second = i >= count;
        const float* src = second ? ws2 + (i - count) : ws + i;
        const size_t stride = second ? count2 : count;
        float4 s = *reinterpret_cast<const float4*>(src);
        for (int k = 1; k < nsl; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(second ? out2 + (i - count) : out + i) = s;
    }
}

// the same reduction for several products in one launch: blockIdx.y = job
struct TnReduceJob {
    const float* ws;
    const float* ws2;
    float* out;
    float* out2;
    size_t count, count2;
    int nsl;
};
struct TnReduceJobs {
    TnReduceJob job[kMaxTnJobs];
    int n;
};
__global__ __launch_bounds__(256) void tn_reduce_multi(const TnReduceJobs J) {
    const TnReduceJob& r = J.job[blockIdx.y];
    const size_t count = r.count, count2 = r.count2;
    const int nsl = r.nsl;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count + count2; i += (size_t)gridDim.x * 256 * 4) {
        const bool second = i >= count;
        const float* src = second ? r.ws2 + (i - count) : r.ws + i;
        const size_t stride = second ? count2 : count;
        float4 s = *reinterpret_cast<const float4*>(src);
        for (int k = 1; k < nsl; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(second ? r.out2 + (i - count) : r.out + i) = s;
    }
}

}  // namespace

extern "C" int64_t hd_gemm_tn_bf16x3_workspace_floats(int64_t R, int64_t P, int64_t Q) {
    if (R < 1 || P < 1 || Q < 1) return 0;
    const int64_t tiles = ((P + TP - 1) / TP) * ((Q + TQ - 1) / TQ);
    const int s = host_slices(R, tiles);
    return s > 1 ? (int64_t)s * (P * Q + P) : 0;                  // tiles, then column sums, per slice
}

extern "C" int hd_gemm_tn_bf16x3_f32(const void* A, const void* B, void* C, void* colsum_a, int64_t R, int64_t P, int64_t Q,
                                     int64_t lda, int64_t ldb, int64_t ldc, void* workspace, void* st) {
    HD_REQUIRE(R >= 0 && P >= 1 && Q >= 1, "hd_gemm_tn_bf16x3_f32: bad shape R=%lld P=%lld Q=%lld", (long long)R,
               (long long)P, (long long)Q);
    HD_REQUIRE(A && B && C, "hd_gemm_tn_bf16x3_f32: null pointer");
    HD_REQUIRE(R < (1LL << 31) && P < (1LL << 31) && Q < (1LL << 31), "hd_gemm_tn_bf16x3_f32: size exceeds int32");
    HD_REQUIRE(P % 4 == 0 && Q % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && is_aligned16(A) &&
                   is_aligned16(B) && is_aligned16(C),
               "hd_gemm_tn_bf16x3_f32: P, Q and the leading dimensions must be multiples of 4 and A, B, C 16-byte aligned");
    HD_REQUIRE(lda >= P && ldb >= Q && ldc >= Q, "hd_gemm_tn_bf16x3_f32: leading dimension too small");
    HD_REQUIRE(!colsum_a || is_aligned16(colsum_a), "hd_gemm_tn_bf16x3_f32: colsum_a must be 16-byte aligned");
    hipStream_t s = (hipStream_t)st;
    if (R == 0) {
        HD_REQUIRE(ldc == Q, "hd_gemm_tn_bf16x3_f32: empty reduction needs a contiguous output");
        if (hipMemsetAsync(C, 0, (size_t)P * Q * sizeof(float), s) != hipSuccess ||
            (colsum_a && hipMemsetAsync(colsum_a, 0, (size_t)P * sizeof(float), s) != hipSuccess))
            return check_launch("hd_gemm_tn_bf16x3_f32");
        return HD_OK;
    }
    const int tiles_p = (int)((P + TP - 1) / TP), tiles_q = (int)((Q + TQ - 1) / TQ);
    const int tiles = tiles_p * tiles_q;
    const int splits = host_slices(R, tiles);
    int chunk = (int)((R + splits - 1) / splits);
    chunk = (chunk + TR - 1) / TR * TR;
    const int nsl = (int)((R + chunk - 1) / chunk);
    HD_REQUIRE((int64_t)tiles * nsl < (1LL << 31), "hd_gemm_tn_bf16x3_f32: too many workgroups");
    if (nsl > 1) {
        HD_REQUIRE(workspace && is_aligned16(workspace), "hd_gemm_tn_bf16x3_f32: workspace of hd_gemm_tn_bf16x3_workspace_floats() floats required");
        HD_REQUIRE(ldc == Q, "hd_gemm_tn_bf16x3_f32: split reduction needs a contiguous output (ldc == Q)");
        const size_t count = (size_t)P * Q;
        float* cs_ws = colsum_a ? (float*)workspace + (size_t)nsl * count : nullptr;
        gemm_tn_bf16x3<<<dim3((unsigned)((tiles * nsl + 7) / 8 * 8)), 256, 0, s>>>(
            (const float*)A, (const float*)B, (float*)workspace, (int)R, (int)P, (int)Q, (int)lda, (int)ldb, (int)Q, chunk,
            tiles_q, tiles, tiles * nsl, (size_t)P * Q, cs_ws);
        const size_t count2 = colsum_a ? (size_t)P : 0;
        int blocks = (int)(((count + count2) / 4 + 255) / 256);
        blocks = blocks < 2048 ? blocks : 2048;
        tn_reduce<<<blocks, 256, 0, s>>>((const float*)workspace, nsl, count, (float*)C, cs_ws, count2, (float*)colsum_a);
    } else {
        gemm_tn_bf16x3<<<dim3((unsigned)((tiles + 7) / 8 * 8)), 256, 0, s>>>((const float*)A, (const float*)B, (float*)C, (int)R,
                                                                               (int)P, (int)Q, (int)lda, (int)ldb, (int)ldc,
                                                                               chunk, tiles_q, tiles, tiles, 0,
                                                                               (float*)colsum_a);
    }
    return check_launch("hd_gemm_tn_bf16x3_f32");
}

// plan of one image product: how hd_gemm_tn_limbs_f32 cuts it (tiles, slices), where its partial tiles go
struct TnPlan {
    TnJob job;
    int nsl;
    size_t count, count2;     // P * Q, P (or 0 without a bias gradient)
    float *ws, *cs_ws, *C, *cs;
};
static int tn_plan(const char* who, const void* Aimg, const void* Bimg, void* C, void* colsum_a, int64_t R, int64_t P,
                   int64_t Q, int64_t ldc, float* workspace, TnPlan& pl) {
    HD_REQUIRE(P >= 1 && Q >= 1, "%s: bad shape R=%lld P=%lld Q=%lld", who, (long long)R, (long long)P, (long long)Q);
    HD_REQUIRE(C && Aimg && Bimg, "%s: null pointer", who);
    HD_REQUIRE(P < (1LL << 24) && Q < (1LL << 24), "%s: size exceeds the index range", who);
    HD_REQUIRE(P % 4 == 0 && Q % 4 == 0 && ldc % 4 == 0 && ldc >= Q && is_aligned16(C) && is_aligned16(Aimg) && is_aligned16(Bimg),
               "%s: P, Q, ldc must be multiples of 4 and the images and C 16-byte aligned", who);
    HD_REQUIRE(!colsum_a || is_aligned16(colsum_a), "%s: colsum_a must be 16-byte aligned", who);
    const int tiles_p = (int)((P + TP - 1) / TP), tiles_q = (int)((Q + TQ - 1) / TQ);
    const int tiles = tiles_p * tiles_q;
    const int slabs = (int)((R + TR - 1) / TR);
    int nsl = host_slices(R, tiles);
    nsl = nsl < slabs ? nsl : slabs;
    HD_REQUIRE((int64_t)tiles * nsl < (1LL << 31), "%s: too many workgroups", who);
    pl.nsl = nsl;
    pl.count = (size_t)P * Q;
    pl.count2 = colsum_a ? (size_t)P : 0;
    pl.C = (float*)C;
    pl.cs = (float*)colsum_a;
    pl.ws = pl.cs_ws = nullptr;
    TnJob& J = pl.job;
    J.Aimg = (const uint4*)Aimg;
    J.Bimg = (const uint4*)Bimg;
    J.R = (int)R; J.P = (int)P; J.Q = (int)Q;
    J.slabs_q = slabs / nsl; J.slabs_rem = slabs % nsl;
    J.tiles_q = tiles_q; J.tiles = tiles; J.total = tiles * nsl;
    if (nsl > 1) {
        HD_REQUIRE(workspace && is_aligned16(workspace), "%s: workspace required", who);
        HD_REQUIRE(ldc == Q, "%s: split reduction needs a contiguous output (ldc == Q)", who);
        pl.ws = workspace;
        pl.cs_ws = colsum_a ? workspace + (size_t)nsl * pl.count : nullptr;
        J.out = pl.ws; J.cs_out = pl.cs_ws; J.ldo = (int)Q; J.slice_stride = pl.count;
    } else {
        J.out = (float*)C; J.cs_out = (float*)colsum_a; J.ldo = (int)ldc; J.slice_stride = 0;
    }
    return HD_OK;
}

extern "C" int hd_gemm_tn_limbs_f32(const void* Aimg, const void* Bimg, void* C, void* colsum_a, int64_t R, int64_t P,
                                    int64_t Q, int64_t ldc, void* workspace, void* st) {
    HD_REQUIRE(R >= 0 && R < (1LL << 31), "hd_gemm_tn_limbs_f32: bad shape R=%lld P=%lld Q=%lld", (long long)R, (long long)P,
               (long long)Q);
    hipStream_t s = (hipStream_t)st;
    if (R == 0) {
        HD_REQUIRE(P >= 1 && Q >= 1 && C && ldc == Q, "hd_gemm_tn_limbs_f32: empty reduction needs a contiguous output");
        if (hipMemsetAsync(C, 0, (size_t)P * Q * sizeof(float), s) != hipSuccess ||
            (colsum_a && hipMemsetAsync(colsum_a, 0, (size_t)P * sizeof(float), s) != hipSuccess))
            return check_launch("hd_gemm_tn_limbs_f32");
        return HD_OK;
    }
    TnPlan pl;
    if (const int rc = tn_plan("hd_gemm_tn_limbs_f32", Aimg, Bimg, C, colsum_a, R, P, Q, ldc, (float*)workspace, pl)) return rc;
    gemm_tn_img<<<dim3((unsigned)((pl.job.total + 7) / 8 * 8)), 256, 0, s>>>(pl.job);
    if (pl.nsl > 1) {
        int blocks = (int)(((pl.count + pl.count2) / 4 + 255) / 256);
        blocks = blocks < 2048 ? blocks : 2048;
        tn_reduce<<<blocks, 256, 0, s>>>(pl.ws, pl.nsl, pl.count, pl.C, pl.cs_ws, pl.count2, pl.cs);
    }
    return check_launch("hd_gemm_tn_limbs_f32");
}

// workspace of hd_gemm_tn_limbs_multi_f32: the jobs' own workspaces, one after the other (each a multiple of 4 floats)
extern "C" int64_t hd_gemm_tn_limbs_multi_workspace_floats(int64_t R, const hd_tn_job* jobs, int n_jobs) {
    if (!jobs || n_jobs < 1 || n_jobs > kMaxTnJobs) return -1;
    int64_t total = 0;
    for (int i = 0; i < n_jobs; ++i) total += hd_gemm_tn_bf16x3_workspace_floats(R, jobs[i].P, jobs[i].Q);
    return total;
}

extern "C" int hd_gemm_tn_limbs_multi_f32(int64_t R, const hd_tn_job* jobs, int n_jobs, void* workspace, void* st) {
    HD_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= kMaxTnJobs, "hd_gemm_tn_limbs_multi_f32: between 1 and %d jobs", kMaxTnJobs);
    HD_REQUIRE(R >= 0 && R < (1LL << 31), "hd_gemm_tn_limbs_multi_f32: bad row count");
    HD_REQUIRE(!workspace || is_aligned16(workspace), "hd_gemm_tn_limbs_multi_f32: misaligned workspace");
    hipStream_t s = (hipStream_t)st;
    if (R == 0) {                                                 // empty sums: the single-product entry point knows how
        for (int i = 0; i < n_jobs; ++i)
            if (const int rc = hd_gemm_tn_limbs_f32(jobs[i].a_img, jobs[i].b_img, jobs[i].c, jobs[i].colsum_a, 0, jobs[i].P,
                                                    jobs[i].Q, jobs[i].ldc, nullptr, st))
                return rc;
        return HD_OK;
    }
    TnMulti M{};
    TnReduceJobs RJ{};
    M.n = n_jobs;
    float* ws = (float*)workspace;
    int64_t most = 0, items = 0;
    int blocks = 1;
    TnPlan pl[kMaxTnJobs];
    for (int i = 0; i < n_jobs; ++i) {
        const hd_tn_job& j = jobs[i];
        if (const int rc = tn_plan("hd_gemm_tn_limbs_multi_f32", j.a_img, j.b_img, j.c, j.colsum_a, R, j.P, j.Q, j.ldc, ws, pl[i]))
            return rc;
        if (ws) ws += hd_gemm_tn_bf16x3_workspace_floats(R, j.P, j.Q);
        M.job[i] = pl[i].job;
        most = pl[i].job.total > most ? pl[i].job.total : most;
        items += pl[i].job.total;
        if (pl[i].nsl > 1) {
            TnReduceJob& r = RJ.job[RJ.n++];
            r.ws = pl[i].ws; r.nsl = pl[i].nsl; r.count = pl[i].count; r.out = pl[i].C;
            r.ws2 = pl[i].cs_ws; r.count2 = pl[i].count2; r.out2 = pl[i].cs;
            const int b = (int)(((r.count + r.count2) / 4 + 255) / 256);
            blocks = b > blocks ? b : blocks;
        }
    }
    // as many workgroups as the largest job has items (every job of one resident round, as hd_gemm_tn_limbs_f32 cuts
    // them, fits), a multiple of 8; job j's items are dealt from workgroup rot_j on
    const int grid = (int)((most + 7) / 8 * 8);
    int64_t rot = 0;
    for (int i = 0; i < n_jobs; ++i) {
        M.rot[i] = (int)(rot % grid);
        rot += pl[i].job.total;
    }
    (void)items;
    gemm_tn_img_multi<<<dim3((unsigned)grid), 256, 0, s>>>(M);
    if (RJ.n > 0) {
        blocks = blocks < 1024 ? blocks : 1024;
        tn_reduce_multi<<<dim3((unsigned)blocks, (unsigned)RJ.n), 256, 0, s>>>(RJ);
    }
    return check_launch("hd_gemm_tn_limbs_multi_f32");
}
