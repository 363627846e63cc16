// Hyperbolic k-means between encode and diffusion (labels / cluster centres u0 of the direction
// diffusion): hyperbolic_learning/hyperbolic_kmeans/hkmeans.py + util_hk.py of the reference, a
// numpy triple loop there (samples x clusters x gradient rounds in Python).  fp64 like numpy.
//
//   to_hyperboloid   util_hk.py:28-61   Poincare rows -> hyperboloid rows (time coordinate LAST)
//   grad_partial     hkmeans.py:26-39   per cluster k: sum over its members of d/(p^2-1) * x,
//                                       p = <theta_k, x>_L, d = arccosh(-p); one wavefront per row,
//                                       block partials in fixed order (no atomics)
//   update           hkmeans.py:41-70   Frechet-mean gradient step on each centre (exp map)
//   to_poincare      util_hk.py:118-133 centre back to the ball (+ the reference's proj)
//   assign           hkmeans.py:206-214 nearest centre by poincare_dist (util_hk.py:14-16)
//
// Quirk kept on purpose: minkowski_distance_gradient's `(...)**-1/2` parses as ((...)**-1)/2,
// so the "gradient" weight is d/(p^2-1), not d/sqrt(p^2-1).
#include "hd_common.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

constexpr int kCols = 9;           // columns per lane: rows up to 64*9 = 576 doubles
constexpr int kRowsPerBlock = 256; // rows handled by one block of grad_partial

__global__ __launch_bounds__(256) void hkm_to_hyperboloid(const double* __restrict__ y, double* __restrict__ h,
                                                          int64_t n, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= n) return;
    const double* yr = y + row * d;
    double ss = 0.0;
    for (int c = lane; c < d; c += 64) ss += yr[c] * yr[c];
    ss = group_sum<64>(ss);
    const double r = sqrt(ss), r2 = r * r;
    const double s = 2.0 / (1.0 - r2 + 1e-6);
    double* hr = h + row * (d + 1);
    for (int c = lane; c < d; c += 64) hr[c] = s * yr[c];
    if (lane == 0) hr[d] = s * (1.0 + r2) / 2.0;
}

// grid (chunks, K): block (c, k) sums w_i * h_i over rows i of chunk c with label k
__global__ __launch_bounds__(256) void hkm_grad_partial(const double* __restrict__ h, const int32_t* __restrict__ labels,
                                                        const double* __restrict__ theta, int64_t n, int d1,
                                                        double* __restrict__ ws, int32_t* __restrict__ cnt) {
    __shared__ double part[4][64 * kCols];
    __shared__ int pc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.y;
    const double* th = theta + (size_t)k * d1;
    double t[kCols], acc[kCols];
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        const int c = lane + 64 * j;
        t[j] = c < d1 ? th[c] : 0.0;
        acc[j] = 0.0;
    }
    int count = 0;
    const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
    const int64_t r1 = r0 + kRowsPerBlock < n ? r0 + kRowsPerBlock : n;
    for (int64_t row = r0 + wave; row < r1; row += 4) {
        if (labels[row] != k) continue;                 // wave-uniform
        const double* hr = h + row * d1;
        double x[kCols], p = 0.0;
#pragma unroll
        for (int j = 0; j < kCols; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < d1 ? hr[c] : 0.0;
            p += (c == d1 - 1) ? -t[j] * x[j] : t[j] * x[j];      // hyperboloid_dot: last coordinate negative
        }
        p = group_sum<64>(p);
        double dist = acosh(-p);
        if (dist != dist) dist = 1e-6;                  // hyperboloid_dist returns eps for NaN
        const double w = dist / (p * p - 1.0);
#pragma unroll
        for (int j = 0; j < kCols; ++j) acc[j] += w * x[j];
        ++count;
    }
#pragma unroll
    for (int j = 0; j < kCols; ++j) part[wave][lane + 64 * j] = acc[j];
    if (lane == 0) pc[wave] = count;
    __syncthreads();
    if (wave == 0) {
        double* out = ws + ((size_t)blockIdx.x * gridDim.y + k) * d1;
#pragma unroll
        for (int j = 0; j < kCols; ++j) {
            const int c = lane + 64 * j;
            if (c < d1) out[c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        }
        if (lane == 0) cnt[(size_t)blockIdx.x * gridDim.y + k] = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    }
}

// one block per cluster: mean gradient -> tangent projection -> exponential map
__global__ __launch_bounds__(64) void hkm_update(const double* __restrict__ ws, const int32_t* __restrict__ cnt,
                                                 int chunks, int K, int d1, double alpha, double* __restrict__ theta) {
    const int lane = threadIdx.x, k = blockIdx.x;
    double g[kCols], t[kCols];
    int total = 0;
#pragma unroll
    for (int j = 0; j < kCols; ++j) g[j] = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const double* p = ws + ((size_t)c * K + k) * d1;
#pragma unroll
        for (int j = 0; j < kCols; ++j) {
            const int col = lane + 64 * j;
            if (col < d1) g[j] += p[col];
        }
        total += cnt[(size_t)c * K + k];
    }
    if (total == 0) return;                             // empty cluster keeps its centre (hkmeans.py:138-139)
    double* th = theta + (size_t)k * d1;
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        const int col = lane + 64 * j;
        t[j] = col < d1 ? th[col] : 0.0;
        g[j] = -(g[j] / (double)total);                 // -grad_loss (2*mean(d * grad/2 ...) folded in w)
        dot += (col == d1 - 1) ? -t[j] * g[j] : t[j] * g[j];
    }
    dot = group_sum<64>(dot);
    double nv = 0.0;
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        g[j] = -alpha * (g[j] + dot * t[j]);            // v = -alpha * project_to_tangent(theta, -grad)
        nv += g[j] * g[j];
    }
    nv = sqrt(group_sum<64>(nv));
    const double ch = cosh(nv), sh = sinh(nv) / (nv + 1e-6);
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        const int col = lane + 64 * j;
        if (col < d1) th[col] = ch * t[j] + sh * g[j];
    }
}

__global__ __launch_bounds__(64) void hkm_to_poincare(const double* __restrict__ theta, int d1, double* __restrict__ cen) {
    const int lane = threadIdx.x, k = blockIdx.x, d = d1 - 1;
    const double* th = theta + (size_t)k * d1;
    const double den = (th[d] + 1.0) + 1e-6;
    double v[kCols], ss = 0.0;
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        const int col = lane + 64 * j;
        v[j] = col < d ? th[col] / den : 0.0;
        ss += v[j] * v[j];
    }
    const double nrm = sqrt(group_sum<64>(ss));
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        const int col = lane + 64 * j;
        if (col < d) cen[(size_t)k * d + col] = nrm >= 1.0 ? v[j] / nrm - 0.1 : v[j];    // util_hk.py:112-115
    }
}

// nearest centre per row; returns label and the squared distance (inertia term)
__global__ __launch_bounds__(256) void hkm_assign(const double* __restrict__ x, const double* __restrict__ cen,
                                                  int64_t n, int d, int K, int32_t* __restrict__ labels,
                                                  double* __restrict__ dist2) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (row >= n) return;
    const double* xr = x + row * d;
    double xx = 0.0;
    for (int c = lane; c < d; c += 64) xx += xr[c] * xr[c];
    xx = group_sum<64>(xx);
    const double nx = sqrt(xx);
    double best = 0.0;
    int bk = -1;
    for (int k = 0; k < K; ++k) {
        const double* cr = cen + (size_t)k * d;
        double dd = 0.0, cc = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double df = xr[c] - cr[c];
            dd += df * df;
            cc += cr[c] * cr[c];
        }
        dd = group_sum<64>(dd);
        cc = group_sum<64>(cc);
        const double nd = sqrt(dd), nc = sqrt(cc);
        const double a = 1.0 + 2.0 * (nd * nd) / ((1.0 - nx * nx) * (1.0 - nc * nc) + 1e-5);
        const double dist = acosh(a);
        // np.argmin: the first minimum; a NaN wins over any number (and the first NaN stays)
        const bool take = bk < 0 || (best == best && (dist != dist || dist < best));
        if (take) { best = dist; bk = k; }
    }
    if (lane == 0) {
        labels[row] = bk;
        if (dist2) dist2[row] = best * best;
    }
}

}  // namespace

extern "C" int64_t hd_hkm_workspace_doubles(int64_t rows, int64_t dim, int64_t K) {
    if (rows < 1 || dim < 1 || K < 1) return 0;
    const int64_t chunks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    // partial sums (chunks*K*(dim+1) doubles) followed by the int32 member counts (chunks*K)
    return chunks * K * (dim + 1) + (chunks * K + 1) / 2;
}

extern "C" int hd_hkm_to_hyperboloid(const double* y, double* h, int64_t rows, int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && dim < (1LL << 30), "hd_hkm_to_hyperboloid: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(y && h, "hd_hkm_to_hyperboloid: null pointer");
    hkm_to_hyperboloid<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)st>>>(y, h, rows, (int)dim);
    return check_launch("hd_hkm_to_hyperboloid");
}

extern "C" int hd_hkm_mean_step(const double* h, const int32_t* labels, double* theta, int64_t rows, int64_t dim,
                                int64_t K, double alpha, double* workspace, void* st) {
    HD_REQUIRE(rows >= 1 && dim >= 1 && K >= 1 && K <= 65535, "hd_hkm_mean_step: bad shape");
    HD_REQUIRE(dim + 1 <= 64 * kCols, "hd_hkm_mean_step: dim=%lld exceeds %d", (long long)dim, 64 * kCols - 1);
    HD_REQUIRE(h && labels && theta && workspace, "hd_hkm_mean_step: null pointer");
    const int64_t chunks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    HD_REQUIRE(chunks < (1LL << 31), "hd_hkm_mean_step: too many rows");
    const int d1 = (int)dim + 1;
    int32_t* cnt = reinterpret_cast<int32_t*>(workspace + chunks * K * d1);
    hkm_grad_partial<<<dim3((unsigned)chunks, (unsigned)K), 256, 0, (hipStream_t)st>>>(h, labels, theta, rows, d1,
                                                                                         workspace, cnt);
    hkm_update<<<(unsigned)K, 64, 0, (hipStream_t)st>>>(workspace, cnt, (int)chunks, (int)K, d1, alpha, theta);
    return check_launch("hd_hkm_mean_step");
}

extern "C" int hd_hkm_to_poincare(const double* theta, double* centres, int64_t K, int64_t dim, void* st) {
    HD_REQUIRE(K >= 1 && dim >= 1 && dim + 1 <= 64 * kCols, "hd_hkm_to_poincare: bad shape");
    HD_REQUIRE(theta && centres, "hd_hkm_to_poincare: null pointer");
    hkm_to_poincare<<<(unsigned)K, 64, 0, (hipStream_t)st>>>(theta, (int)dim + 1, centres);
    return check_launch("hd_hkm_to_poincare");
}

extern "C" int hd_hkm_assign(const double* x, const double* centres, int32_t* labels, double* dist2, int64_t rows,
                             int64_t dim, int64_t K, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && K >= 1 && dim < (1LL << 30) && K < (1LL << 30), "hd_hkm_assign: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && centres && labels, "hd_hkm_assign: null pointer");
    hkm_assign<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)st>>>(x, centres, rows, (int)dim, (int)K, labels, dist2);
    return check_launch("hd_hkm_assign");
}
