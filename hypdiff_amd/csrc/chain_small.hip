// A whole stack of nn.Linear layers on a SMALL batch in ONE launch: ddpm.Backbone at B = 200 (config 3,
// diff_graphset.py:191: 24 Linear layers of width 512) is 24 dependent 200 x 512 x 512 GEMMs -- 1.6 % of a
// chip-filling tile grid each, so a launch per layer spends its time on launch ramps (0.58 ms per sampler
// step in round 1).  Replaces MLP.forward / Backbone.forward of ddpm.py:9-41 for rows <= 512.
//
// Layout of the work: the rows are cut into 8 groups (one per XCD: workgroup b belongs to group b & 7, and the
// dispatcher deals workgroups round-robin over the XCDs, so a group's activations stay near one L2 -- a
// performance assumption only, correctness rests on agent-scope atomics).  A group of G workgroups
// walks the layers together; workgroup g owns output columns 16 g .. 16 g + 15 of every layer: its four
// wavefronts split K (k-steps w, w + 4, ...), each multiplying up to 4 row tiles of 16 rows with
// v_mfma_f32_16x16x32_bf16 on 3 bf16 limbs (6 MFMAs per fp32 product block: fp32-faithful, as the limb GEMM),
// the partial tiles meet in LDS, wavefront `it` finishes row tile `it`: bias, erf-GELU, residual, time
// embedding; the result goes out as fp32 (the next residual) and as limbs in the A-operand order of the next
// layer.  Weights come pre-split in exactly the per-lane operand order (hd_chain_pack_w_f32), straight from L2
// into registers -- with 16..64 rows per group a weight fragment is used by one wavefront only, LDS staging
// would add nothing -- and the loads of layer l + 1 are issued before the group waits for layer l's
// activations.  Between layers the group meets at a counter (group_arrive / group_wait below).  The launch is cooperative (every workgroup resident:
// the spin cannot starve a workgroup that has not started), and the spin is bounded by a 2 s wall clock after
// which the group raises the workspace's fail flag and writes NaN instead of hanging.
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "hd_limbs.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

constexpr int kGroups = 8, kG = 32;            // groups (XCDs) x workgroups per group = 256 workgroups
constexpr int kMaxW = 512;                     // widest layer (input or output)
constexpr int kMaxKS = kMaxW / 32, kMaxRT = 4; // k-steps per layer, row tiles of 16 per group
constexpr int kMaxRows = kGroups * kMaxRT * 16;
constexpr int kMaxLayers = 32;

struct Layer {
    const uint4* w;       // hd_chain_pack_w_f32 image
    const float* bias;    // may be null
    int K, N, epi, add_emb;
};
struct Chain {
    Layer l[kMaxLayers];
    int n;
};

// workspace: [0, 256) counters (one int per group, 32 bytes apart would be nicer for the atomics' lines but a
// group only touches its own word) + fail flag; then the limb images and fp32 copies of the activations, two of each
struct Ws {
    int* ctr;
    int* fail;
    uint4* actl[2];       // [group][it][ks][limb][lane]
    float* actf[2];       // [group][row in group][kMaxW]
};
constexpr size_t kActlBytes = (size_t)kGroups * kMaxRT * kMaxKS * 3 * 64 * sizeof(uint4);
constexpr size_t kActfBytes = (size_t)kMaxRows * kMaxW * sizeof(float);
constexpr size_t kWsBytes = 256 + 2 * kActlBytes + 2 * kActfBytes;

// Activations cross workgroups (and possibly XCDs, whose L2s are not coherent with each other for ordinary
// accesses).  An agent-scope release / acquire FENCE would do it, but on this chip it is a whole-L2 write-back /
// invalidate per workgroup and layer (measured: 38 us per layer, and the spinning acquires kept throwing the
// weights out of L2).  Instead every activation access is a relaxed agent-scope ATOMIC load / store of 8 bytes
// (global_load / global_store with sc1: coherent at the agent level per location, no cache-wide operation), the
// producer drains its stores (vmcnt(0)) before its workgroup bumps the group counter, and the consumer issues its
// loads only after it has seen the counter: no fences.
__device__ __forceinline__ void st_u2(void* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_u2(const void* p) {
    const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)q, (unsigned)(q >> 32));
}
__device__ __forceinline__ void st_u4(uint4* p, uint4 v) {
    st_u2(p, make_uint2(v.x, v.y));
    st_u2(reinterpret_cast<uint2*>(p) + 1, make_uint2(v.z, v.w));
}
__device__ __forceinline__ uint4 ld_u4(const uint4* p) {
    const uint2 a = ld_u2(p), b = ld_u2(reinterpret_cast<const uint2*>(p) + 1);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void st_f(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void group_arrive(int* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's activation stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true = the group is complete; false = timed out or another workgroup raised the fail flag
__device__ __forceinline__ bool group_wait(int* ctr, int target, int* fail) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        const uint64_t t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull ||
                __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    return ok_s != 0;
}

__global__ __launch_bounds__(256) void linear_chain_small(const float* __restrict__ x, int64_t ldx, int rows,
                                                          Chain ch, const float* __restrict__ emb,
                                                          const int64_t* __restrict__ emb_idx, int emb_rows,
                                                          float* __restrict__ out, int64_t ldo, Ws ws, int rg) {
    __shared__ float4 red[4][kMaxRT][64];
    const int grp = blockIdx.x & (kGroups - 1), g = blockIdx.x >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int r0 = grp * rg;                                   // first row of the group
    const int nr = rows - r0 < rg ? rows - r0 : rg;
    if (nr <= 0) return;                                       // (the whole group leaves: it has its own counter)
    const int RT = (nr + 15) >> 4;
    int* ctr = ws.ctr + grp;
    uint4* actl[2] = {ws.actl[0] + (size_t)grp * kMaxRT * kMaxKS * 3 * 64, ws.actl[1] + (size_t)grp * kMaxRT * kMaxKS * 3 * 64};
    float* actf[2] = {ws.actf[0] + (size_t)r0 * kMaxW, ws.actf[1] + (size_t)r0 * kMaxW};
    bool ok = true;

    // stage 0: x -> limb image (and fp32 copy) of the first layer's operand; (it, ks) pairs over the group's wavefronts
    {
        const int K0 = ch.l[0].K, KS0 = (K0 + 31) >> 5;
        for (int p = g * 4 + wave; p < RT * KS0; p += kG * 4) {
            const int it = p / KS0, ks = p - it * KS0;
            const int rl = it * 16 + r16, row = r0 + rl;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ks * 32 + 8 * kq + e;
                v[e] = (row < rows && k < K0) ? x[(size_t)row * ldx + k] : 0.f;
                if (k < K0) st_f(&actf[0][(size_t)rl * kMaxW + k], v[e]);
            }
            uint4 h, m, l;
            limb_split8(v, h, m, l);
            uint4* o = actl[0] + ((size_t)(it * kMaxKS + ks) * 3) * 64 + lane;
            st_u4(o, h); st_u4(o + 64, m); st_u4(o + 128, l);
        }
    }
    group_arrive(ctr);

    for (int li = 0; li < ch.n; ++li) {
        const Layer L = ch.l[li];
        const int cur = li & 1, nxt = cur ^ 1;
        const int KS = (L.K + 31) >> 5, NT = (L.N + 15) >> 4;
        const bool last = li == ch.n - 1;
        bool first_jt = true;
        for (int jt = g; jt < NT || first_jt; jt += kG) {
            const bool work = jt < NT;
            // this wavefront's weight fragments (k-steps wave, wave + 4, ...): issued before the wait below
            uint4 wf[kMaxKS / 4][3];
#pragma unroll
            for (int q = 0; q < kMaxKS / 4; ++q) {
                const int ks = wave + 4 * q;
#pragma unroll
                for (int l = 0; l < 3; ++l)
                    wf[q][l] = (work && ks < KS) ? L.w[((size_t)(jt * KS + ks) * 3 + l) * 64 + lane] : make_uint4(0, 0, 0, 0);
            }
            if (first_jt) {
                ok = group_wait(ctr, kG * (li + 1), ws.fail) && ok;     // layer li's operand is complete
                first_jt = false;
            }
            if (!work) break;
            // operand maps of v_mfma_f32_16x16x32: first = W rows (n = 16 jt + r16), second = activation rows
            // (m = r16), k = 8 kq .. 8 kq + 7 of the k-step; D column = lane & 15 = m, D rows 4 kq + reg = n
            // two row tiles at a time: all their fragment loads are issued before the first multiply, so the
            // tiles share one memory round trip (the loads are agent-scope atomics: the compiler keeps their order)
#pragma unroll
            for (int it0 = 0; it0 < kMaxRT; it0 += 2) {
                if (it0 < RT) {
                    uint4 af[2][kMaxKS / 4][3];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < kMaxKS / 4; ++q) {
                            const int ks = wave + 4 * q;
                            if (it0 + h < RT && ks < KS) {
                                const uint4* a = actl[cur] + ((size_t)((it0 + h) * kMaxKS + ks) * 3) * 64 + lane;
                                af[h][q][0] = ld_u4(a); af[h][q][1] = ld_u4(a + 64); af[h][q][2] = ld_u4(a + 128);
                            } else {
                                af[h][q][0] = af[h][q][1] = af[h][q][2] = make_uint4(0, 0, 0, 0);
                            }
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (it0 + h < RT) {
                            lf32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int q = 0; q < kMaxKS / 4; ++q) {
                                if (wave + 4 * q < KS) {
#define HD_MM(wv, av) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, wv), \
                                                          __builtin_bit_cast(lbf16x8, av), c, 0, 0, 0);
                                    HD_MM(wf[q][0], af[h][q][2]) HD_MM(wf[q][2], af[h][q][0]) HD_MM(wf[q][1], af[h][q][1])
                                    HD_MM(wf[q][0], af[h][q][1]) HD_MM(wf[q][1], af[h][q][0]) HD_MM(wf[q][0], af[h][q][0])
#undef HD_MM
                                }
                            }
                            red[wave][it0 + h][lane] = make_float4(c[0], c[1], c[2], c[3]);
                        }
                    }
                }
            }
            __syncthreads();
            if (wave < RT) {                                       // wavefront `it` finishes row tile `it`
                const int it = wave;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 p = red[w][it][lane];
                    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
                }
                const int rl = it * 16 + r16, row = r0 + rl, nb = jt * 16 + 4 * kq;
                const int64_t er = (L.add_emb && row < rows) ? emb_idx[row] : -1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nb + e;
                    float y = 0.f;
                    if (n < L.N) {
                        y = v[e] + (L.bias ? L.bias[n] : 0.f);
                        if (L.epi == 1) y = gelu_erfc(y);
                        else if (L.epi == 2) y = ld_f(&actf[cur][(size_t)rl * kMaxW + n]) + gelu_erfc(y);
                        if (er >= 0 && er < emb_rows) y += emb[er * (int64_t)L.N + n];
                        else if (L.add_emb && row < rows) y = __builtin_nanf("");    // index out of range (nn.Embedding raises)
                    }
                    v[e] = y;
                }
                if (last) {
                    if (row < rows) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb + e < L.N) out[(size_t)row * ldo + nb + e] = ok ? v[e] : __builtin_nanf("");
                    }
                } else {
                    float* fo = actf[nxt] + (size_t)rl * kMaxW + nb;
                    st_u2(fo, make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1])));
                    st_u2(fo + 2, make_uint2(__float_as_uint(v[2]), __float_as_uint(v[3])));
                    // columns 16 jt .. + 15 = half a k-step of the next layer: lane (r16, 2 (jt & 1) + (kq >> 1)),
                    // 8-byte half kq & 1
                    unsigned hh[2], mm[2], ll[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        hh[e] = limb_pk_bf16(v[2 * e], v[2 * e + 1]);
                        const float q0 = v[2 * e] - limb_lo(hh[e]), q1 = v[2 * e + 1] - limb_hi(hh[e]);
                        mm[e] = limb_pk_bf16(q0, q1);
                        ll[e] = limb_pk_bf16(q0 - limb_lo(mm[e]), q1 - limb_hi(mm[e]));
                    }
                    const int ln = r16 + 16 * (2 * (jt & 1) + (kq >> 1));
                    uint2* o = reinterpret_cast<uint2*>(actl[nxt] + ((size_t)(it * kMaxKS + (jt >> 1)) * 3) * 64 + ln) + (kq & 1);
                    st_u2(o, make_uint2(hh[0], hh[1]));
                    st_u2(o + 2 * 64, make_uint2(mm[0], mm[1]));
                    st_u2(o + 2 * 128, make_uint2(ll[0], ll[1]));
                }
            }
            __syncthreads();                                       // red[] is free again
        }
        if (!last) group_arrive(ctr);
    }
}

// W (N, K) fp32 row-major (row stride ld) -> [jt][ks][limb][lane] uint4: limb of W[16 jt + (lane & 15)][32 ks + 8 (lane >> 4) .. + 7]
__global__ __launch_bounds__(64) void chain_pack_w(const float* __restrict__ W, int64_t ld, int N, int K, int KS,
                                                   uint4* __restrict__ img) {
    const int jt = blockIdx.x / KS, ks = blockIdx.x - jt * KS, lane = threadIdx.x;
    const int nrow = 16 * jt + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (nrow < N && k0 + e < K) ? W[(int64_t)nrow * ld + k0 + e] : 0.f;
    uint4 h, m, l;
    limb_split8(v, h, m, l);
    uint4* o = img + ((size_t)(jt * KS + ks) * 3) * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

}  // namespace

extern "C" int64_t hd_chain_image_bytes(int64_t d_out, int64_t d_in) {
    if (d_out < 1 || d_in < 1) return -1;
    return ((d_out + 15) / 16) * ((d_in + 31) / 32) * 3 * 64 * 16;
}

extern "C" int64_t hd_chain_workspace_bytes(void) { return (int64_t)kWsBytes; }

extern "C" int hd_chain_pack_w_f32(const void* W, int64_t ld, int64_t d_out, int64_t d_in, void* img, void* st) {
    HD_REQUIRE(d_out >= 1 && d_in >= 1 && d_out <= kMaxW && d_in <= kMaxW && ld >= d_in,
               "hd_chain_pack_w_f32: layer wider than %d", kMaxW);
    HD_REQUIRE(W && img && is_aligned16(img), "hd_chain_pack_w_f32: null or misaligned pointer");
    const int KS = (int)((d_in + 31) / 32);
    chain_pack_w<<<dim3((unsigned)(((d_out + 15) / 16) * KS)), dim3(64), 0, (hipStream_t)st>>>(
        (const float*)W, ld, (int)d_out, (int)d_in, KS, (uint4*)img);
    return check_launch("hd_chain_pack_w_f32");
}

extern "C" int hd_linear_chain_small_f32(const void* x, int64_t ldx, int64_t rows, const hd_chain_layer* layers,
                                         int n_layers, const void* emb, const int64_t* emb_idx, int64_t emb_rows,
                                         void* out, int64_t ldo, void* workspace, int64_t workspace_bytes, void* st) {
    HD_REQUIRE(rows >= 0 && rows <= kMaxRows, "hd_linear_chain_small_f32: rows must be <= %d (use the GEMM chain beyond)", kMaxRows);
    HD_REQUIRE(layers && n_layers >= 1 && n_layers <= kMaxLayers, "hd_linear_chain_small_f32: 1..%d layers", kMaxLayers);
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out && workspace && is_aligned16(workspace) && workspace_bytes >= (int64_t)kWsBytes,
               "hd_linear_chain_small_f32: null pointer or workspace smaller than hd_chain_workspace_bytes()");
    Chain ch;
    ch.n = n_layers;
    bool any_emb = false;
    for (int i = 0; i < n_layers; ++i) {
        const hd_chain_layer& a = layers[i];
        HD_REQUIRE(a.wimg && is_aligned16(a.wimg), "hd_linear_chain_small_f32: layer %d: null or misaligned weight image", i);
        HD_REQUIRE(a.d_in >= 1 && a.d_in <= kMaxW && a.d_out >= 1 && a.d_out <= kMaxW,
                   "hd_linear_chain_small_f32: layer %d wider than %d", i, kMaxW);
        HD_REQUIRE(a.epi >= 0 && a.epi <= 2, "hd_linear_chain_small_f32: layer %d: epilogue must be 0, 1 or 2", i);
        HD_REQUIRE(a.epi != 2 || a.d_in == a.d_out, "hd_linear_chain_small_f32: layer %d: a residual layer keeps its width", i);
        HD_REQUIRE(i == 0 || a.d_in == layers[i - 1].d_out, "hd_linear_chain_small_f32: layer %d: d_in != previous d_out", i);
        HD_REQUIRE(i == n_layers - 1 || a.d_out % 32 == 0,
                   "hd_linear_chain_small_f32: layer %d: inner widths must be multiples of 32", i);
        ch.l[i] = Layer{(const uint4*)a.wimg, (const float*)a.bias, a.d_in, a.d_out, a.epi, a.add_emb};
        any_emb = any_emb || a.add_emb;
    }
    HD_REQUIRE(ldx >= layers[0].d_in && ldo >= layers[n_layers - 1].d_out, "hd_linear_chain_small_f32: leading dimension too small");
    HD_REQUIRE(!any_emb || (emb && emb_idx && emb_rows >= 1), "hd_linear_chain_small_f32: embedding table / indices missing");
    char* p = (char*)workspace;
    Ws ws;
    ws.ctr = (int*)p;
    ws.fail = (int*)(p + 128);
    ws.actl[0] = (uint4*)(p + 256);
    ws.actl[1] = (uint4*)(p + 256 + kActlBytes);
    ws.actf[0] = (float*)(p + 256 + 2 * kActlBytes);
    ws.actf[1] = (float*)(p + 256 + 2 * kActlBytes + kActfBytes);
    hipError_t e = hipMemsetAsync(workspace, 0, 256, (hipStream_t)st);
    if (e != hipSuccess) { set_error("hd_linear_chain_small_f32: hipMemsetAsync: %s", hipGetErrorString(e)); return HD_EHIP; }
    // rows per group: a multiple of 16, at most kMaxRT tiles
    int rg = (int)((rows + kGroups * 16 - 1) / (kGroups * 16)) * 16;
    const float* xp = (const float*)x;
    const float* ep = (const float*)emb;
    float* op = (float*)out;
    int rows_i = (int)rows, emb_rows_i = (int)(emb_rows < 0x7fffffff ? emb_rows : 0x7fffffff);
    void* args[] = {&xp, &ldx, &rows_i, &ch, &ep, &emb_idx, &emb_rows_i, &op, &ldo, &ws, &rg};
    e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(linear_chain_small), dim3(kGroups * kG), dim3(256), args,
                                   0, (hipStream_t)st);
    if (e != hipSuccess) {
        set_error("hd_linear_chain_small_f32: cooperative launch failed: %s", hipGetErrorString(e));
        return HD_EHIP;
    }
    return HD_OK;
}

// 1 after a launch whose group barrier timed out (its outputs are NaN); reads the workspace (synchronises the stream)
extern "C" int hd_chain_failed(const void* workspace, void* st) {
    int f = 0;
    if (!workspace) return 0;
    if (hipMemcpyAsync(&f, (const char*)workspace + 128, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)st) != hipSuccess) return -1;
    return f;
}
