// Direction-diffusion step kernels (models/direction_diffusion.py of the reference).
// All are single-pass elementwise / reduction kernels bound by HBM bandwidth.  dtype
// promotion follows the reference: schedule tables fp64, getpri fp32, denoiser fp32,
// so q_sample / p_sample outputs are fp64 whatever the input dtype.
#include "hd_manifold.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

// getpri, :191-196 -- int64 t times a python float is an fp32 tensor in torch
__device__ __forceinline__ float getpri(int64_t t, float sqrt_c, float horizon, float prior) {
    return prior * tanhf(sqrt_c * (float)t / horizon);
}

__device__ __forceinline__ double ld(const void* p, int dtype, int64_t i) {
    return dtype ? ((const double*)p)[i] : (double)((const float*)p)[i];
}

// q_sample, :198-213 (+ tran_direction :154-163 in place on noise when restrict)
template <typename TX, typename TN>
__global__ __launch_bounds__(256) void q_sample_k(const TX* __restrict__ x0, TN* __restrict__ noise,
                                                  const int64_t* __restrict__ t, const int32_t* __restrict__ labels,
                                                  const float* __restrict__ signs, int64_t n_centres,
                                                  const double* __restrict__ sqrt_ac,
                                                  const double* __restrict__ sqrt_1mac, int64_t T,
                                                  double* __restrict__ out, int64_t rows, int64_t dim,
                                                  int restrict_, float sqrt_c, float horizon, float prior) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim, c = i - r * dim;
        int64_t tt = t[r];
        tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        TN nz = noise[i];
        if (restrict_) {
            int32_t lb = labels[r];
            lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
            nz = (TN)(nz * (TN)signs[(int64_t)lb * dim + c]);
            noise[i] = nz;
        }
        const TX x = x0[i];
        const float pri = getpri(t[r], sqrt_c, horizon, prior);
        // pri * x_start is evaluated in the promoted dtype of (fp32, TX)
        const double px = sizeof(TX) == 4 ? (double)(pri * (float)x) : (double)pri * (double)x;
        out[i] = sqrt_ac[tt] * (double)x + sqrt_1mac[tt] * (double)nz + px;
    }
}

// The training forward's three passes between the moments and the denoiser in one (:145-149 normalization,
// :198-213 q_sample with tran_direction, :57-58 the mlp's cat(x.float(), t)): x_start = (x - mean) / std in fp64, kept
// as fp32 for the loss (:362 casts it); x_t from x_start, the direction-constrained noise and the prior; the mlp's packed
// fp32 input row [x_t.float(), t / T, 0 ...].  Every expression is the one of standardize_k / q_sample_k /
// pack_input_k, so the results are bit-identical to the three launches.  The noise is written back (sign-flipped in
// place, as tran_direction does) only when the caller owns it.
template <typename TN>
__global__ __launch_bounds__(256) void q_sample_pack_k(const double* __restrict__ x, const double* __restrict__ m,
                                                       TN* __restrict__ noise, int write_noise,
                                                       const int64_t* __restrict__ t, const float* __restrict__ t_col,
                                                       const int32_t* __restrict__ labels,
                                                       const float* __restrict__ signs, int64_t n_centres,
                                                       const double* __restrict__ sqrt_ac,
                                                       const double* __restrict__ sqrt_1mac, int64_t T,
                                                       float* __restrict__ x_start, float* __restrict__ h, int64_t ldh,
                                                       int64_t rows, int64_t dim, float sqrt_c, float horizon, float prior) {
    const double n = m[2], mean = m[0] / n;
    const double var = (m[1] - m[0] * m[0] / n) / (n - 1.0);
    const double sd = sqrt(var > 0.0 ? var : 0.0);
    const double mul = 1.0 / sd, add = -mean / sd;
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim, c = i - r * dim;
        const int64_t traw = t[r];
        const int64_t tt = traw < 0 ? 0 : (traw >= T ? T - 1 : traw);
        int32_t lb = labels[r];
        lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
        const TN nz = (TN)(noise[i] * (TN)signs[(int64_t)lb * dim + c]);
        if (write_noise) noise[i] = nz;
        const double xs = fma(x[i], mul, add);
        x_start[i] = (float)xs;
        const float pri = getpri(traw, sqrt_c, horizon, prior);
        const double px = (double)pri * xs;
        const double xt = sqrt_ac[tt] * xs + sqrt_1mac[tt] * (double)nz + px;
        float* hr = h + r * ldh;
        hr[c] = (float)xt;
        if (c == 0) {
            hr[dim] = t_col ? t_col[r] : (float)traw / (float)T;
            for (int64_t e = dim + 1; e < ldh; ++e) hr[e] = 0.f;
        }
    }
}

// the same, four consecutive elements per lane with 16-byte accesses (dim % 4 == 0, ldh % 4 == 0, aligned bases): the
// element-wise form above spends its time on one 64-bit division per element
template <typename TN>
__global__ __launch_bounds__(256) void q_sample_pack_v4(const double* __restrict__ x, const double* __restrict__ m,
                                                        TN* __restrict__ noise, int write_noise,
                                                        const int64_t* __restrict__ t, const float* __restrict__ t_col,
                                                        const int32_t* __restrict__ labels,
                                                        const float* __restrict__ signs, int64_t n_centres,
                                                        const double* __restrict__ sqrt_ac,
                                                        const double* __restrict__ sqrt_1mac, int64_t T,
                                                        float* __restrict__ x_start, float* __restrict__ h, int64_t ldh,
                                                        int64_t rows, int64_t dim, float sqrt_c, float horizon, float prior) {
    const double n = m[2], mean = m[0] / n;
    const double var = (m[1] - m[0] * m[0] / n) / (n - 1.0);
    const double sd = sqrt(var > 0.0 ? var : 0.0);
    const double mul = 1.0 / sd, add = -mean / sd;
    const int64_t dq = dim / 4, total4 = rows * dq;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / dq, c = (q - r * dq) * 4, i = q * 4;
        const int64_t traw = t[r];
        const int64_t tt = traw < 0 ? 0 : (traw >= T ? T - 1 : traw);
        int32_t lb = labels[r];
        lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
        const float4 sg = *reinterpret_cast<const float4*>(signs + (int64_t)lb * dim + c);
        const float s4[4] = {sg.x, sg.y, sg.z, sg.w};
        TN nz[4];
        if constexpr (sizeof(TN) == 8) {
            const double2 a = *reinterpret_cast<const double2*>(noise + i), b = *reinterpret_cast<const double2*>(noise + i + 2);
            nz[0] = a.x; nz[1] = a.y; nz[2] = b.x; nz[3] = b.y;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(noise + i);
            nz[0] = a.x; nz[1] = a.y; nz[2] = a.z; nz[3] = a.w;
        }
        const double2 xa = *reinterpret_cast<const double2*>(x + i), xb = *reinterpret_cast<const double2*>(x + i + 2);
        const double xv[4] = {xa.x, xa.y, xb.x, xb.y};
        const float pri = getpri(traw, sqrt_c, horizon, prior);
        const double sa = sqrt_ac[tt], sm = sqrt_1mac[tt];
        float xs4[4], xt4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            nz[e] = (TN)(nz[e] * (TN)s4[e]);
            const double xs = fma(xv[e], mul, add);
            xs4[e] = (float)xs;
            const double px = (double)pri * xs;
            xt4[e] = (float)(sa * xs + sm * (double)nz[e] + px);
        }
        if (write_noise) {
            if constexpr (sizeof(TN) == 8) {
                *reinterpret_cast<double2*>(noise + i) = make_double2(nz[0], nz[1]);
                *reinterpret_cast<double2*>(noise + i + 2) = make_double2(nz[2], nz[3]);
            } else {
                *reinterpret_cast<float4*>(noise + i) = make_float4(nz[0], nz[1], nz[2], nz[3]);
            }
        }
        *reinterpret_cast<float4*>(x_start + i) = make_float4(xs4[0], xs4[1], xs4[2], xs4[3]);
        float* hr = h + r * ldh;
        *reinterpret_cast<float4*>(hr + c) = make_float4(xt4[0], xt4[1], xt4[2], xt4[3]);
        if (c == 0) {
            hr[dim] = t_col ? t_col[r] : (float)traw / (float)T;
            for (int64_t e = dim + 1; e < ldh; ++e) hr[e] = 0.f;
        }
    }
}

// p_sample after the denoiser, :248-291:
//   eps = ((sqrt(1/acp_t) + pri_t) * x_t - x0_raw) / sqrt(1/acp_t - 1)     (:230-236, unclamped x0)
//   x0  = clamp(x0_raw, -1, 1)                                           (:266-267, in place)
//   eps *= sign(direction[label])                                        (:289 -> :201)
//   out = sqrt(acp_t2) * x0 + sqrt(1-acp_t2) * eps + pri_t2 * x0          (:212-213)
template <typename TX>
__global__ __launch_bounds__(256) void p_sample_update_k(const TX* __restrict__ xt, float* __restrict__ x0,
                                                         const int32_t* __restrict__ labels,
                                                         const float* __restrict__ signs, int64_t n_centres,
                                                         double ra, double rm1, double a2, double b2, float pri_t,
                                                         float pri_t2, double* __restrict__ out, int64_t rows,
                                                         int64_t dim) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim, c = i - r * dim;
        int32_t lb = labels[r];
        lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
        const float raw = x0[i];
        double eps = ((ra + (double)pri_t) * (double)xt[i] - (double)raw) / rm1;
        eps = eps * (double)signs[(int64_t)lb * dim + c];
        const float xc = raw < -1.f ? -1.f : (raw > 1.f ? 1.f : raw);
        x0[i] = xc;
        out[i] = a2 * (double)xc + b2 * eps + (double)(pri_t2 * xc);
    }
}

// the same, four consecutive elements per lane with 16-byte accesses (dim % 4 == 0, aligned rows)
template <typename TX>
__global__ __launch_bounds__(256) void p_sample_update_v4(const TX* __restrict__ xt, float* __restrict__ x0,
                                                          const int32_t* __restrict__ labels,
                                                          const float* __restrict__ signs, int64_t n_centres,
                                                          double ra, double rm1, double a2, double b2, float pri_t,
                                                          float pri_t2, double* __restrict__ out, int64_t rows,
                                                          int64_t dim) {
    const int64_t total4 = rows * dim / 4;
    const int64_t dq = dim / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / dq, c = (q - r * dq) * 4, i = q * 4;
        int32_t lb = labels[r];
        lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
        double x[4];
        if constexpr (sizeof(TX) == 8) {
            const double2 a = *reinterpret_cast<const double2*>(xt + i), b = *reinterpret_cast<const double2*>(xt + i + 2);
            x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(xt + i);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        }
        const float4 rw = *reinterpret_cast<const float4*>(x0 + i);
        const float4 sg = *reinterpret_cast<const float4*>(signs + (int64_t)lb * dim + c);
        const float raw[4] = {rw.x, rw.y, rw.z, rw.w}, s4[4] = {sg.x, sg.y, sg.z, sg.w};
        float xc[4];
        double o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double eps = ((ra + (double)pri_t) * x[e] - (double)raw[e]) / rm1;
            eps = eps * (double)s4[e];
            xc[e] = raw[e] < -1.f ? -1.f : (raw[e] > 1.f ? 1.f : raw[e]);
            o[e] = a2 * (double)xc[e] + b2 * eps + (double)(pri_t2 * xc[e]);
        }
        *reinterpret_cast<float4*>(x0 + i) = make_float4(xc[0], xc[1], xc[2], xc[3]);
        *reinterpret_cast<double2*>(out + i) = make_double2(o[0], o[1]);
        *reinterpret_cast<double2*>(out + i + 2) = make_double2(o[2], o[3]);
    }
}

// tran_direction, :154-163: noise[i] *= sign(direction[labels[i]]) in place
template <typename TN>
__global__ __launch_bounds__(256) void tran_direction_k(TN* __restrict__ noise, const int32_t* __restrict__ labels,
                                                        const float* __restrict__ signs, int64_t n_centres,
                                                        int64_t rows, int64_t dim) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim, c = i - r * dim;
        int32_t lb = labels[r];
        lb = lb < 0 ? 0 : (lb >= n_centres ? (int32_t)n_centres - 1 : lb);
        noise[i] = (TN)(noise[i] * (TN)signs[(int64_t)lb * dim + c]);
    }
}

// mlp input: cat(x.float(), t[:,None]).float()  (:57-58), row pitch ld >= dim+1, zero padded
template <typename TX>
__global__ __launch_bounds__(256) void pack_input_k(const TX* __restrict__ x, const void* __restrict__ t, int t_kind,
                                                    float t_scalar, float* __restrict__ out, int64_t rows,
                                                    int64_t dim, int64_t ldo) {
    const int64_t total = rows * ldo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ldo, c = i - r * ldo;
        float v = 0.f;
        if (c < dim) v = (float)x[r * dim + c];
        else if (c == dim) v = t_kind == 0 ? (float)((const int64_t*)t)[r] : (t_kind == 1 ? ((const float*)t)[r] : t_scalar);
        out[i] = v;
    }
}

constexpr int kMomentBlocks = 1024;

// 16-byte loads, four independent fp64 accumulator pairs per lane (the fixed element -> lane ->
// block assignment keeps the reduction tree, hence the bits, reproducible)
template <typename TX>
__global__ __launch_bounds__(256) void moments_partial_k(const TX* __restrict__ x, int64_t count,
                                                         double* __restrict__ ws) {
    constexpr int V = 16 / sizeof(TX);
    using VT = typename VecOf<TX>::type;
    __shared__ double s1[4], s2[4];
    double a = 0.0, b = 0.0;
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? count / V : 0;
    const VT* xv = reinterpret_cast<const VT*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const VT t = xv[i];
        const TX* tp = reinterpret_cast<const TX*>(&t);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const double v = (double)tp[e];
            a += v;
            b += v * v;
        }
    }
    for (int64_t i = nvec * V + (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i];
        a += v;
        b += v * v;
    }
    a = group_sum<64>(a);
    b = group_sum<64>(b);
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        ws[2 * blockIdx.x + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
}

__global__ __launch_bounds__(256) void moments_final_k(const double* __restrict__ ws, int nblocks,
                                                       double* __restrict__ out, double count) {
    __shared__ double s1[4], s2[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += ws[2 * i]; b += ws[2 * i + 1]; }
    a = group_sum<64>(a);
    b = group_sum<64>(b);
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        out[1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
        out[2] = count;
    }
}

// normalization / renormal, :145-153: global mean and UNBIASED std from (sum, sumsq, count)
template <typename TX>
__global__ __launch_bounds__(256) void standardize_k(const TX* __restrict__ x, TX* __restrict__ out, int64_t count,
                                                     const double* __restrict__ m, int inverse) {
    constexpr int V = 16 / sizeof(TX);
    using VT = typename VecOf<TX>::type;
    const double n = m[2], mean = m[0] / n;
    const double var = (m[1] - m[0] * m[0] / n) / (n - 1.0);
    const double sd = sqrt(var > 0.0 ? var : 0.0);
    // one multiply-add per element in fp64: (v - mean) / sd = v * (1/sd) - mean/sd up to 1 ulp of fp64
    const double mul = inverse ? sd : 1.0 / sd;
    const double add = inverse ? mean : -mean / sd;
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int64_t nvec = al ? count / V : 0;
    const VT* xv = reinterpret_cast<const VT*>(x);
    VT* ov = reinterpret_cast<VT*>(out);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        VT t = xv[i];
        TX* tp = reinterpret_cast<TX*>(&t);
#pragma unroll
        for (int e = 0; e < V; ++e) tp[e] = (TX)fma((double)tp[e], mul, add);
        ov[i] = t;
    }
    for (int64_t i = nvec * V + (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        out[i] = (TX)fma((double)x[i], mul, add);
}

// ---- graph-level latent DDPM (ddpm.py / ddpm_utils.py), fp32 throughout -----------------
// Model.forward noising, ddpm.py:95-104: idx per row, tables fp32.
//   restrict == 0: sqrt(ab) * x + sqrt(1 - ab) * eps
//   restrict != 0: pri * x + sqrt(ab) * x + sign(x) * sqrt(1 - ab) * eps,  pri = deta*tanh(sqrt(c)*t/2000)
__global__ __launch_bounds__(256) void ddpm_noise_k(const float* __restrict__ x, const float* __restrict__ eps,
                                                    const int64_t* __restrict__ idx,
                                                    const float* __restrict__ alpha_bars, int64_t T,
                                                    float* __restrict__ out, int64_t rows, int64_t dim,
                                                    int restrict_, float deta, float sqrt_c, float horizon) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        int64_t t = idx[r];
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float ab = alpha_bars[t];
        const float sa = sqrtf(ab), sb = sqrtf(1.f - ab);
        const float xv = x[i], e = eps[i];
        if (!restrict_) {
            out[i] = sa * xv + sb * e;
        } else {
            const float pri = deta * tanhf(sqrt_c * (float)idx[r] / horizon);
            const float sg = xv > 0.f ? 1.f : (xv < 0.f ? -1.f : xv);
            out[i] = pri * xv + sa * xv + sg * sb * e;
        }
    }
}

// DDPMSampler._one_diffusion_step, ddpm_utils.py:36-45 (target pred_noise, restrict False):
//   x <- c1 * (x - c2 * eps) + c3 * noise,  c1 = sqrt(1/alpha_t), c2 = beta_t / sqrt(1 - ab_t),
//   c3 = sqrt((1 - ab_{t-1}) / (1 - ab_t) * beta_t); noise may be NULL (idx == 0: zeros)
__global__ __launch_bounds__(256) void ddpm_step_k(const float* __restrict__ x, const float* __restrict__ eps,
                                                   const float* __restrict__ noise, float c1, float c2, float c3,
                                                   float* __restrict__ out, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float mu = c1 * (x[i] - c2 * eps[i]);
        out[i] = mu + c3 * (noise ? noise[i] : 0.f);
    }
}

// Backbone.forward, ddpm.py:39-41: h += embedding_layer(idx) (rows gathered from the (T, dim) table)
__global__ __launch_bounds__(256) void embedding_add_k(float* __restrict__ h, const float* __restrict__ table,
                                                       const int64_t* __restrict__ idx, int64_t T, int64_t rows,
                                                       int64_t dim) {
    const int64_t total = rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim, c = i - r * dim;
        int64_t t = idx[r];
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        h[i] = h[i] + table[t * dim + c];
    }
}

// gradient of the embedding table of embedding_add (autograd of ddpm.py:39-41): out[t] = sum of the rows of g whose
// index is t, in ascending row order (deterministic, no atomics).  `sorted_idx` / `sorted_rows` = a stable sort of idx
// (values and positions): table row t owns one contiguous run, found by two binary searches -- nothing is read back to
// the host, so the step can be captured in a HIP graph.  One workgroup per table row, a thread per column.
__global__ __launch_bounds__(256) void embedding_grad_k(const float* __restrict__ g, const int64_t* __restrict__ sorted_idx,
                                                        const int64_t* __restrict__ sorted_rows, int64_t rows, int64_t dim,
                                                        float* __restrict__ out) {
    const int64_t t = blockIdx.x;
    int64_t lo = 0, hi = rows;                       // first position with sorted_idx >= t
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_idx[mid] < t) lo = mid + 1; else hi = mid; }
    const int64_t begin = lo;
    hi = rows;                                       // first position with sorted_idx > t
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_idx[mid] <= t) lo = mid + 1; else hi = mid; }
    const int64_t end = lo;
    for (int64_t c = threadIdx.x; c < dim; c += 256) {
        float acc = 0.f;
        for (int64_t k = begin; k < end; ++k) acc += g[sorted_rows[k] * dim + c];
        out[t * dim + c] = acc;
    }
}

inline int ew_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < 256 * 32 ? b : 256 * 32);
}

}  // namespace

extern "C" int hd_q_sample(const void* x0, int x_dtype, void* noise, int noise_dtype, const int64_t* t,
                           const int32_t* labels, const float* signs, int64_t n_centres, const double* sqrt_ac,
                           const double* sqrt_1mac, int64_t T, double* out, int64_t rows, int64_t dim,
                           int restrict_, double sqrt_c, double horizon, double prior, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && T >= 1, "hd_q_sample: bad shape rows=%lld dim=%lld T=%lld", (long long)rows,
               (long long)dim, (long long)T);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x0 && noise && t && sqrt_ac && sqrt_1mac && out, "hd_q_sample: null pointer");
    HD_REQUIRE(!restrict_ || (labels && signs && n_centres >= 1), "hd_q_sample: restrict needs labels and signs");
    const int g = ew_grid(rows * dim);
    hipStream_t s = (hipStream_t)st;
#define QS(TX, TN)                                                                                               \
    q_sample_k<TX, TN><<<g, 256, 0, s>>>((const TX*)x0, (TN*)noise, t, labels, signs, n_centres, sqrt_ac, sqrt_1mac, \
                                         T, out, rows, dim, restrict_, (float)sqrt_c, (float)horizon, (float)prior)
    if (!x_dtype && !noise_dtype) QS(float, float);
    else if (!x_dtype && noise_dtype) QS(float, double);
    else if (x_dtype && !noise_dtype) QS(double, float);
    else QS(double, double);
#undef QS
    return check_launch("hd_q_sample");
}

extern "C" int hd_q_sample_pack_train(const void* x, const void* moments3, void* noise, int noise_dtype, int write_noise,
                                      const int64_t* t, const float* t_col, const int32_t* labels, const float* signs, int64_t n_centres,
                                      const double* sqrt_ac, const double* sqrt_1mac, int64_t T, void* x_start,
                                      void* h, int64_t ldh, int64_t rows, int64_t dim, double sqrt_c, double horizon,
                                      double prior, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && T >= 1 && ldh >= dim + 1, "hd_q_sample_pack_train: bad shape rows=%lld dim=%lld T=%lld ldh=%lld",
               (long long)rows, (long long)dim, (long long)T, (long long)ldh);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && moments3 && noise && t && labels && signs && n_centres >= 1 && sqrt_ac && sqrt_1mac && x_start && h,
               "hd_q_sample_pack_train: null pointer");
    hipStream_t s = (hipStream_t)st;
    if (dim % 4 == 0 && ldh % 4 == 0 && is_aligned16(x) && is_aligned16(noise) && is_aligned16(signs) && is_aligned16(x_start) &&
        is_aligned16(h)) {
        const int g4 = ew_grid(rows * dim / 4);
        if (noise_dtype)
            q_sample_pack_v4<double><<<g4, 256, 0, s>>>((const double*)x, (const double*)moments3, (double*)noise, write_noise, t,
                                                       t_col, labels, signs, n_centres, sqrt_ac, sqrt_1mac, T, (float*)x_start,
                                                       (float*)h, ldh, rows, dim, (float)sqrt_c, (float)horizon, (float)prior);
        else
            q_sample_pack_v4<float><<<g4, 256, 0, s>>>((const double*)x, (const double*)moments3, (float*)noise, write_noise, t,
                                                      t_col, labels, signs, n_centres, sqrt_ac, sqrt_1mac, T, (float*)x_start,
                                                      (float*)h, ldh, rows, dim, (float)sqrt_c, (float)horizon, (float)prior);
        return check_launch("hd_q_sample_pack_train");
    }
    const int g = ew_grid(rows * dim);
    if (noise_dtype)
        q_sample_pack_k<double><<<g, 256, 0, s>>>((const double*)x, (const double*)moments3, (double*)noise, write_noise, t,
                                                  t_col, labels, signs, n_centres, sqrt_ac, sqrt_1mac, T, (float*)x_start, (float*)h,
                                                  ldh, rows, dim, (float)sqrt_c, (float)horizon, (float)prior);
    else
        q_sample_pack_k<float><<<g, 256, 0, s>>>((const double*)x, (const double*)moments3, (float*)noise, write_noise, t,
                                                 t_col, labels, signs, n_centres, sqrt_ac, sqrt_1mac, T, (float*)x_start, (float*)h,
                                                 ldh, rows, dim, (float)sqrt_c, (float)horizon, (float)prior);
    return check_launch("hd_q_sample_pack_train");
}

extern "C" int hd_p_sample_update(const void* x_t, int x_dtype, float* x0, const int32_t* labels,
                                  const float* signs, int64_t n_centres, double sqrt_recip_t,
                                  double sqrt_recipm1_t, double sqrt_ac_t2, double sqrt_1mac_t2, int64_t t,
                                  int64_t t2, double* out, int64_t rows, int64_t dim, double sqrt_c,
                                  double horizon, double prior, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1, "hd_p_sample_update: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x_t && x0 && labels && signs && out && n_centres >= 1, "hd_p_sample_update: null pointer");
    // fp32 prior of a scalar step, same expression as the per-row kernel path
    const float pri_t = (float)prior * tanhf((float)sqrt_c * (float)t / (float)horizon);
    const float pri_t2 = (float)prior * tanhf((float)sqrt_c * (float)t2 / (float)horizon);
    const bool v4 = dim % 4 == 0 && is_aligned16(x_t) && is_aligned16(x0) && is_aligned16(out) && is_aligned16(signs);
    const int g = ew_grid(v4 ? rows * dim / 4 : rows * dim);
#define HD_PSU(K_, T_)                                                                                         \
    K_<T_><<<g, 256, 0, (hipStream_t)st>>>((const T_*)x_t, x0, labels, signs, n_centres, sqrt_recip_t,        \
                                           sqrt_recipm1_t, sqrt_ac_t2, sqrt_1mac_t2, pri_t, pri_t2, out, rows, dim)
    if (v4) {
        if (x_dtype) HD_PSU(p_sample_update_v4, double);
        else HD_PSU(p_sample_update_v4, float);
    } else {
        if (x_dtype) HD_PSU(p_sample_update_k, double);
        else HD_PSU(p_sample_update_k, float);
    }
#undef HD_PSU
    return check_launch("hd_p_sample_update");
}

extern "C" int hd_tran_direction(void* noise, int noise_dtype, const int32_t* labels, const float* signs,
                                 int64_t n_centres, int64_t rows, int64_t dim, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1, "hd_tran_direction: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(noise && labels && signs && n_centres >= 1, "hd_tran_direction: null pointer");
    const int g = ew_grid(rows * dim);
    if (noise_dtype) tran_direction_k<double><<<g, 256, 0, (hipStream_t)st>>>((double*)noise, labels, signs, n_centres, rows, dim);
    else tran_direction_k<float><<<g, 256, 0, (hipStream_t)st>>>((float*)noise, labels, signs, n_centres, rows, dim);
    return check_launch("hd_tran_direction");
}

extern "C" int hd_mlp_pack_input(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar, float* out,
                                 int64_t rows, int64_t dim, int64_t ld_out, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && ld_out >= dim + 1, "hd_mlp_pack_input: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && (t || t_kind == 2) && t_kind >= 0 && t_kind <= 2, "hd_mlp_pack_input: bad argument");
    const int g = ew_grid(rows * ld_out);
    if (x_dtype)
        pack_input_k<double><<<g, 256, 0, (hipStream_t)st>>>((const double*)x, t, t_kind, (float)t_scalar, out, rows, dim, ld_out);
    else
        pack_input_k<float><<<g, 256, 0, (hipStream_t)st>>>((const float*)x, t, t_kind, (float)t_scalar, out, rows, dim, ld_out);
    return check_launch("hd_mlp_pack_input");
}

extern "C" int hd_ddpm_noise(const float* x, const float* eps, const int64_t* idx, const float* alpha_bars, int64_t T,
                             float* out, int64_t rows, int64_t dim, int restrict_, double deta, double sqrt_c,
                             double horizon, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && T >= 1, "hd_ddpm_noise: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && eps && idx && alpha_bars && out, "hd_ddpm_noise: null pointer");
    ddpm_noise_k<<<ew_grid(rows * dim), 256, 0, (hipStream_t)st>>>(x, eps, idx, alpha_bars, T, out, rows, dim, restrict_,
                                                                  (float)deta, (float)sqrt_c, (float)horizon);
    return check_launch("hd_ddpm_noise");
}

extern "C" int hd_ddpm_step(const float* x, const float* eps, const float* noise, double c1, double c2, double c3,
                            float* out, int64_t count, void* st) {
    HD_REQUIRE(count >= 0, "hd_ddpm_step: bad count");
    if (count == 0) return HD_OK;
    HD_REQUIRE(x && eps && out, "hd_ddpm_step: null pointer");
    ddpm_step_k<<<ew_grid(count), 256, 0, (hipStream_t)st>>>(x, eps, noise, (float)c1, (float)c2, (float)c3, out, count);
    return check_launch("hd_ddpm_step");
}

extern "C" int hd_embedding_add(float* h, const float* table, const int64_t* idx, int64_t T, int64_t rows, int64_t dim,
                                void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && T >= 1, "hd_embedding_add: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(h && table && idx, "hd_embedding_add: null pointer");
    embedding_add_k<<<ew_grid(rows * dim), 256, 0, (hipStream_t)st>>>(h, table, idx, T, rows, dim);
    return check_launch("hd_embedding_add");
}

extern "C" int hd_embedding_grad_f32(const float* g, const int64_t* sorted_idx, const int64_t* sorted_rows, int64_t rows,
                                     int64_t dim, int64_t T, float* out, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1 && T >= 1 && T < (1LL << 31), "hd_embedding_grad_f32: bad shape");
    HD_REQUIRE(out && (rows == 0 || (g && sorted_idx && sorted_rows)), "hd_embedding_grad_f32: null pointer");
    embedding_grad_k<<<dim3((unsigned)T), 256, 0, (hipStream_t)st>>>(g, sorted_idx, sorted_rows, rows, dim, out);
    return check_launch("hd_embedding_grad_f32");
}

extern "C" int64_t hd_moments_workspace_doubles(void) { return 2 * kMomentBlocks; }

extern "C" int hd_moments(const void* x, int dtype, int64_t count, double* workspace, double* out3, void* st) {
    HD_REQUIRE(count >= 1 && x && workspace && out3, "hd_moments: bad argument");
    int64_t nb = (count + 255) / 256;
    nb = nb < kMomentBlocks ? nb : kMomentBlocks;
    if (dtype) moments_partial_k<double><<<(int)nb, 256, 0, (hipStream_t)st>>>((const double*)x, count, workspace);
    else moments_partial_k<float><<<(int)nb, 256, 0, (hipStream_t)st>>>((const float*)x, count, workspace);
    int rc = check_launch("hd_moments(partial)");
    if (rc) return rc;
    moments_final_k<<<1, 256, 0, (hipStream_t)st>>>(workspace, (int)nb, out3, (double)count);
    return check_launch("hd_moments(final)");
}

// ---- hd_moments for callers that cannot allocate (HIP graph capture, tight serving loops) ----------
// The partial sums live in a file-scope device array instead of a caller workspace.  In host code the
// array's name is only a stand-in; its device address is resolved per GPU with hipGetSymbolAddress the
// first time the entry point runs on that GPU (with that GPU current), and only that pointer is handed
// to the kernels.  Every call on a GPU shares the GPU's one array: calls that overlap on one GPU race.
static __device__ __align__(256) double scratch[2 * kMomentBlocks];          // 16 KiB per GPU, held until exit
constexpr int kMaxDevices = 64;
static double* g_scratch_ptr[kMaxDevices];                                   // device address of `scratch`, per GPU
constexpr int64_t kNoAllocMaxCount = (int64_t)1 << 32;

static int scratch_of_current_device(double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("hd_moments_noalloc: hipGetDevice: %s", hipGetErrorString(e)); return HD_EHIP; }
    HD_REQUIRE(dev >= 0 && dev < kMaxDevices, "hd_moments_noalloc: device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    double* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        void* q = nullptr;
        e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !q) {
            set_error("hd_moments_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
            return HD_EHIP;
        }
        p = (double*)q;
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);      // racing first calls resolve the same address
    }
    *out = p;
    return HD_OK;
}

extern "C" int hd_moments_noalloc(const void* x, int dtype, int64_t count, double* out3, void* st) {
    HD_REQUIRE(count >= 1 && x && out3, "hd_moments_noalloc: bad argument");
    HD_REQUIRE(count <= kNoAllocMaxCount, "hd_moments_noalloc: count %lld exceeds the maximum of %lld elements",
               (long long)count, (long long)kNoAllocMaxCount);
    double* ws = nullptr;
    if (int rc = scratch_of_current_device(&ws)) return rc;
    // same kernels, grid and reduction tree as hd_moments; every slot the final pass reads is written by
    // the partial pass of the same call, so nothing of the previous call's contents is consumed
    return hd_moments(x, dtype, count, ws, out3, st);
}

extern "C" int hd_standardize(const void* x, void* out, int dtype, int64_t count, const double* moments3,
                              int inverse, void* st) {
    HD_REQUIRE(count >= 0, "hd_standardize: bad count");
    if (count == 0) return HD_OK;
    HD_REQUIRE(x && out && moments3, "hd_standardize: null pointer");
    const int g = ew_grid(count);
    if (dtype) standardize_k<double><<<g, 256, 0, (hipStream_t)st>>>((const double*)x, (double*)out, count, moments3, inverse);
    else standardize_k<float><<<g, 256, 0, (hipStream_t)st>>>((const float*)x, (float*)out, count, moments3, inverse);
    return check_launch("hd_standardize");
}
