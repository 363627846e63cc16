// Latent -> graph decoding: the tensor part of the reference's diff.generate
// (diff.py:327-352) and decode_adj (:355-378) for a whole graph set in one launch.
//   e = -log(1 / (h + 1e-8) - 1)                      logit of the (G, 1, n, n) latent
//   y = (e - min_g e) / (max_g e - min_g e) > 0.5     per-graph min-max normalisation, threshold
//   adj = decode_adj(y): row i of y lists, newest first, the links of node i+1 to its predecessors
//         -> symmetric (n+1, n+1) 0/1 matrix with adj[a][b] = y[max(a,b)-1][|a-b|-1]
// One workgroup per graph.  any[g] = 0 marks a graph with no entry above the threshold (the reference then
// keeps the input graph).  HBM-bound: 4 n^2 bytes read, 4 (n+1)^2 written per graph.
#include "hd_common.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

__device__ __forceinline__ float logit_ref(float h) { return -logf(1.f / (h + 1e-8f) - 1.f); }

__global__ __launch_bounds__(256) void generate_adj_k(const float* __restrict__ h, float* __restrict__ adj,
                                                      int32_t* __restrict__ any, int n) {
    __shared__ float s_min[4], s_max[4];
    __shared__ int s_nan[4], s_any[4];
    const float* hg = h + (size_t)blockIdx.x * n * n;
    float* out = adj + (size_t)blockIdx.x * (n + 1) * (n + 1);
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int i = threadIdx.x; i < n * n; i += 256) {
        const float e = logit_ref(hg[i]);
        nan |= (e != e);
        mn = fminf(mn, e);
        mx = fmaxf(mx, e);
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
        nan |= __shfl_xor(nan, off);
    }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = mn; s_max[threadIdx.x >> 6] = mx; s_nan[threadIdx.x >> 6] = nan; }
    __syncthreads();
    mn = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    nan = s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3];
    const float range = mx - mn;                            // 0 -> y = 0/0 = NaN -> nothing above 0.5
    int found = 0;
    const int m1 = n + 1;
    for (int i = threadIdx.x; i < m1 * m1; i += 256) {
        const int a = i / m1, b = i - a * m1;
        float v = 0.f;
        if (a != b && !nan) {
            const int hi = a > b ? a : b, dlt = a > b ? a - b : b - a;
            const float y = (logit_ref(hg[(hi - 1) * n + (dlt - 1)]) - mn) / range;
            v = y > 0.5f ? 1.f : 0.f;
        }
        out[i] = v;
    }
    // the reference tests torch.any over the WHOLE thresholded (n, n) matrix, not only the decoded part
    if (!nan)
        for (int i = threadIdx.x; i < n * n; i += 256) found |= ((logit_ref(hg[i]) - mn) / range > 0.5f);
#pragma unroll
    for (int off = 32; off; off >>= 1) found |= __shfl_xor(found, off);
    if ((threadIdx.x & 63) == 0) s_any[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0) any[blockIdx.x] = s_any[0] | s_any[1] | s_any[2] | s_any[3];
}

}  // namespace

extern "C" int hd_generate_adj_f32(const float* h, float* adj, int32_t* any, int64_t n_graphs, int64_t n, void* st) {
    HD_REQUIRE(n_graphs >= 0 && n >= 1 && n < 32768, "hd_generate_adj_f32: bad shape graphs=%lld n=%lld",
               (long long)n_graphs, (long long)n);
    if (n_graphs == 0) return HD_OK;
    HD_REQUIRE(h && adj && any, "hd_generate_adj_f32: null pointer");
    HD_REQUIRE(n_graphs < (1LL << 31), "hd_generate_adj_f32: too many graphs");
    generate_adj_k<<<(unsigned)n_graphs, 256, 0, (hipStream_t)st>>>(h, adj, any, (int)n);
    return check_launch("hd_generate_adj_f32");
}
