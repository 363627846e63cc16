// A whole stack of nn.Linear layers on a SMALL batch in ONE launch: ddpm.Backbone at B = 200 (config 3,
// diff_graphset.py:191: 24 Linear layers of width 512) is 24 dependent 200 x 512 x 512 GEMMs -- 1.6 % of a
// chip-filling tile grid each, so a launch per layer spends its time on launch ramps (0.58 ms per sampler
// step in round 1).  Replaces MLP.forward / Backbone.forward of ddpm.py:9-41 for rows <= 512.
//
// Layout of the work: the row tiles (16 rows) are dealt to up to 16 groups of 16 workgroups (workgroup b is
// member b >> 4 of group b & 15; one or two row tiles per group).  A group walks the layers together; member g
// owns output columns 32 g .. 32 g + 31 of every layer (two 16-column MFMA tiles): its four wavefronts split K
// (k-steps w, w + 4, ...), each multiplying with v_mfma_f32_16x16x32_bf16 on 3 bf16 limbs (6 MFMAs per fp32
// product block: fp32-faithful, as the limb GEMM), the partial tiles meet in LDS, one wavefront per output tile
// finishes it: bias, erf-GELU, residual (the tile's previous value is still in that wavefront's registers), time
// embedding.  Weights come pre-split in exactly the per-lane operand order (hd_chain_pack_w_f32), straight from
// L2 into registers -- a weight fragment is used by one wavefront only, LDS staging would add nothing.  The 24
// matrices are 37 MB, so every layer's slice misses L2: it is requested as soon as the previous layer's multiplies
// have released the registers, one hand-over ahead of its use (the bias / embedding values a tile's write-out
// needs are loaded before that request: a wavefront's loads return in order).
//
// Between layers nothing synchronises but the data itself.  An activation element travels as ONE 8-byte word:
// its three bf16 limbs and a 16-bit tag (launch epoch, layer).  Words are written and read with relaxed
// agent-scope atomics (global_store / global_load ... sc1: coherent per location across CUs and XCDs, no
// cache-wide operation), an 8-byte atomic arrives whole, so a consumer that sees the tag it is waiting for has
// the element; it simply re-reads the words whose tag is still old.  One fabric round trip per layer instead of
// four (store drain, counter add, counter poll, data load: the counter version ran 8.6 us per layer at B = 200;
// agent-scope release / acquire FENCES, the textbook form, 38 us -- on this chip they write back / invalidate the
// whole L2 per workgroup and layer).  The two word buffers alternate by layer; a workgroup writes layer l's output
// only after it has read all of layer l's input, which exists only once every member has finished reading layer
// l - 1's input -- the buffer being overwritten.  Tags never repeat inside the workspace: the epoch (1..2046) lives
// in the workspace header, the last workgroup to leave a launch advances it and, when it wraps, zeroes the words
// (tag 0 is never waited for; the caller zero-fills a new workspace once).  The grid fits the chip at once (checked at
// the launch; see there why it is an ordinary launch and not a cooperative one by default), and every wait is
// bounded by a 2 s wall clock, after which the workgroup raises the fail flag and the outputs are NaN.
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "hd_limbs.cuh"
#include "../../include/hypdiff.h"
#include <stdlib.h>

using namespace hd;

namespace {

constexpr int kGroups = 16, kG = 16;           // groups x members = 256 workgroups
constexpr int kMaxW = 512;                     // widest layer (input or output)
constexpr int kMaxKS = kMaxW / 32, kMaxRT = 2; // k-steps per layer, row tiles of 16 per group
constexpr int kMaxRows = kGroups * kMaxRT * 16;
constexpr int kMaxLayers = 31;                 // 5-bit layer field of the tag, stage 0 included
constexpr int kEpochs = 2046;                  // epochs 1..2046: tags 0x0000 and 0xffff never occur

struct Layer {
    const uint4* w;       // hd_chain_pack_w_f32 image
    const float* bias;    // may be null
    int K, N, epi, add_emb;
};
struct Chain {
    Layer l[kMaxLayers];
    int n;
};

// workspace: header (int 0: launches completed mod kEpochs, int 1: workgroups that have left the running launch,
// int 32: fail flag of the running launch, int 33: sticky report for hd_chain_failed), then the two word buffers [group][row tile][k-step][element 0..7 of the lane][lane]: a
// wavefront's load / store instruction covers 512 contiguous bytes
typedef unsigned long long u64;
constexpr size_t kGroupWords = (size_t)kMaxRT * kMaxKS * 8 * 64;
constexpr size_t kBufWords = (size_t)kGroups * kGroupWords;
constexpr size_t kWsBytes = 256 + 2 * kBufWords * sizeof(u64);

__device__ __forceinline__ void st_w(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_w(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one activation element -> word: limbs h | m << 16 | l << 32, tag << 48 (x = h + m + l exactly)
__device__ __forceinline__ u64 make_word(float x, unsigned tag) {
    const unsigned h = limb_pk_bf16(x, 0.f) & 0xffffu;
    const float r1 = x - limb_lo(h);
    const unsigned m = limb_pk_bf16(r1, 0.f) & 0xffffu;
    const unsigned l = limb_pk_bf16(r1 - limb_lo(m), 0.f) & 0xffffu;
    return (u64)(h | (m << 16)) | ((u64)(l | (tag << 16)) << 32);
}

__global__ __launch_bounds__(256) void linear_chain_small(const float* __restrict__ x, int64_t ldx, int rows,
                                                          Chain ch, const float* __restrict__ emb,
                                                          const int64_t* __restrict__ emb_idx, int emb_rows,
                                                          float* __restrict__ out, int64_t ldo, int* hdr, u64* buf0,
                                                          u64* buf1, int RT, int withhold) {
    __shared__ float4 red[4][2 * kMaxRT][64];
    __shared__ int bad_s, last_s;
    const int grp = blockIdx.x & (kGroups - 1), g = blockIdx.x >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int epoch = hdr[0] + 1;
    int* fail = hdr + 32;
    const int r0 = grp * RT * 16;                              // first row of the group
    const int nrt = r0 < rows ? ((rows - r0 + 15) >> 4 < RT ? (rows - r0 + 15) >> 4 : RT) : 0;   // its row tiles
    // (two named pointers, not an array: indexing an array of pointers loses the global address space and the
    // accesses become flat_load / flat_store)
    u64* const gb0 = buf0 + (size_t)grp * kGroupWords;
    u64* const gb1 = buf1 + (size_t)grp * kGroupWords;
    bool ok = true;
    if (tid == 0) bad_s = 0;
    float py[4] = {0.f, 0.f, 0.f, 0.f};                        // this thread's 4 outputs of the previous layer
    // the output tile this wavefront finishes: column tile 2 g + jl_e, row tile it_e
    const int jl_e = wave / kMaxRT, it_e = wave % kMaxRT;
    const int row_e = r0 + it_e * 16 + r16;

    if (nrt > 0) {
        // stage 0: x -> words tagged (epoch, 0); (row tile, k-step) pairs over the group's wavefronts
        {
            const int K0 = ch.l[0].K, KS0 = (K0 + 31) >> 5;
            const unsigned tag = (unsigned)(epoch << 5);
            // (withhold: tests only -- group 0's first member never publishes its share of the input, so every consumer
            // of the group runs into the bounded wait: tests/test_ddpm.py::test_chain_small_timeout_and_recovery)
            for (int p = g * 4 + wave; p < nrt * KS0 && !(withhold && grp == 0 && g == 0); p += kG * 4) {
                const int it = p / KS0, ks = p - it * KS0;
                const int row = r0 + it * 16 + r16;
                u64* o = gb0 + (size_t)(it * kMaxKS + ks) * 8 * 64 + lane;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 32 + 8 * kq + e;
                    st_w(o + e * 64, make_word((row < rows && k < K0) ? x[(size_t)row * ldx + k] : 0.f, tag));
                }
            }
        }
        // this wavefront's weight fragments of a layer (k-steps wave, wave + 4, ...; column tiles 2 g, 2 g + 1)
        uint4 wf[2][kMaxKS / 4][3];
        auto load_w = [&](const Layer& L) {
            const int KS = (L.K + 31) >> 5, NT = (L.N + 15) >> 4;
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int q = 0; q < kMaxKS / 4; ++q) {
                    const int ks = wave + 4 * q;
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        wf[jl][q][l] = (2 * g + jl < NT && ks < KS) ? L.w[((size_t)((2 * g + jl) * KS + ks) * 3 + l) * 64 + lane]
                                                                    : make_uint4(0, 0, 0, 0);
                }
        };
        load_w(ch.l[0]);
        for (int li = 0; li < ch.n; ++li) {
            const Layer L = ch.l[li];
            const u64* in = (li & 1) ? gb1 : gb0;
            u64* nxt = (li & 1) ? gb0 : gb1;
            const int KS = (L.K + 31) >> 5, NT = (L.N + 15) >> 4;
            const bool last = li == ch.n - 1;
            const int njl = 2 * g + 1 < NT ? 2 : (2 * g < NT ? 1 : 0);     // this member's column tiles: 2 g, 2 g + 1
            if (njl == 0) {                                                // (narrow layer: nothing to do here)
                if (!last) load_w(ch.l[li + 1]);
                continue;
            }
            // what this wavefront's write-out will need from memory, requested first (loads return in order)
            const bool fin = jl_e < njl && it_e < nrt;
            const int nb_e = (2 * g + jl_e) * 16 + 4 * kq;
            float bq[4] = {0.f, 0.f, 0.f, 0.f}, eq[4] = {0.f, 0.f, 0.f, 0.f};
            bool emb_bad = false;
            if (fin) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (L.bias && nb_e + e < L.N) bq[e] = L.bias[nb_e + e];
                if (L.add_emb && row_e < rows) {
                    const int64_t er = emb_idx[row_e];
                    emb_bad = er < 0 || er >= emb_rows;             // nn.Embedding raises: the row comes back NaN
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!emb_bad && nb_e + e < L.N) eq[e] = emb[er * (int64_t)L.N + nb_e + e];
                }
            }
            // this wavefront's share of the input: lane (r16, kq) holds row r16, k = 32 ks + 8 kq .. + 7
            u64 aw[kMaxRT][kMaxKS / 4][8];
            const u64 want = (u64)((epoch << 5) | li);
            const uint64_t t0 = wall_clock64();
            // wait on ONE word per lane first: polling the whole share while the producers are busy floods the fabric
            if (wave < KS) {
                const u64* rep = in + (size_t)wave * 8 * 64 + lane;
                while (__any((ld_w(rep) >> 48) != want)) {
                    if (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
#pragma unroll
            for (int it = 0; it < kMaxRT; ++it)
#pragma unroll
                for (int q = 0; q < kMaxKS / 4; ++q) {
                    const int ks = wave + 4 * q;
                    const bool use = it < nrt && ks < KS;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        aw[it][q][e] = use ? ld_w(in + ((size_t)(it * kMaxKS + ks) * 8 + e) * 64 + lane) : (want << 48);
                }
            for (;;) {
                bool stale = false;
#pragma unroll
                for (int it = 0; it < kMaxRT; ++it)
#pragma unroll
                    for (int q = 0; q < kMaxKS / 4; ++q)
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if ((aw[it][q][e] >> 48) != want) {
                                aw[it][q][e] = ld_w(in + ((size_t)(it * kMaxKS + wave + 4 * q) * 8 + e) * 64 + lane);
                                stale = true;
                            }
                if (!__any(stale)) break;
                if (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bad_s = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            // operand maps of v_mfma_f32_16x16x32: first = W rows (n = 16 jt + r16), second = activation rows
            // (m = r16), k = 8 kq .. 8 kq + 7 of the k-step; D column = lane & 15 = m, D rows 4 kq + reg = n
            lf32x4 c[kMaxRT][2];
#pragma unroll
            for (int it = 0; it < kMaxRT; ++it) {
                c[it][0] = c[it][1] = lf32x4{0.f, 0.f, 0.f, 0.f};
                if (it < nrt) {
#pragma unroll
                    for (int q = 0; q < kMaxKS / 4; ++q) {
                        if (wave + 4 * q < KS) {
                            // 8 words -> the three limb fragments (8 bf16 each)
                            unsigned hh[4], mm[4], ll[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned lo0 = (unsigned)aw[it][q][2 * e], lo1 = (unsigned)aw[it][q][2 * e + 1];
                                const unsigned hi0 = (unsigned)(aw[it][q][2 * e] >> 32), hi1 = (unsigned)(aw[it][q][2 * e + 1] >> 32);
                                hh[e] = __builtin_amdgcn_perm(lo1, lo0, 0x05040100u);
                                mm[e] = __builtin_amdgcn_perm(lo1, lo0, 0x07060302u);
                                ll[e] = __builtin_amdgcn_perm(hi1, hi0, 0x05040100u);
                            }
                            const uint4 a0 = make_uint4(hh[0], hh[1], hh[2], hh[3]), a1 = make_uint4(mm[0], mm[1], mm[2], mm[3]),
                                        a2 = make_uint4(ll[0], ll[1], ll[2], ll[3]);
#pragma unroll
                            for (int jl = 0; jl < 2; ++jl) {
                                if (jl < njl) {
                                    lf32x4 cc = c[it][jl];
#define HD_MM(wv, av) cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8, wv), \
                                                           __builtin_bit_cast(lbf16x8, av), cc, 0, 0, 0);
                                    HD_MM(wf[jl][q][0], a2) HD_MM(wf[jl][q][2], a0) HD_MM(wf[jl][q][1], a1)
                                    HD_MM(wf[jl][q][0], a1) HD_MM(wf[jl][q][1], a0) HD_MM(wf[jl][q][0], a0)
#undef HD_MM
                                    c[it][jl] = cc;
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();                                       // the previous layer's write-out has read red[]
#pragma unroll
            for (int it = 0; it < kMaxRT; ++it)
#pragma unroll
                for (int jl = 0; jl < 2; ++jl)
                    red[wave][jl * kMaxRT + it][lane] = make_float4(c[it][jl][0], c[it][jl][1], c[it][jl][2], c[it][jl][3]);
            // the next layer's weights: the registers are free, and the request has a whole hand-over to land
            if (!last) load_w(ch.l[li + 1]);
            __syncthreads();
            ok = ok && bad_s == 0;
            if (fin) {                                             // this wavefront finishes output tile (jl_e, it_e)
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 p = red[w][wave][lane];
                    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = 0.f;
                    if (nb_e + e < L.N) {
                        y = v[e] + bq[e];
                        if (L.epi == 1) y = gelu_erfc(y);
                        else if (L.epi == 2) y = py[e] + gelu_erfc(y);
                        y += eq[e];
                        if (emb_bad) y = __builtin_nanf("");
                    }
                    v[e] = y;
                    py[e] = y;
                }
                if (last) {
                    if (row_e < rows) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb_e + e < L.N) out[(size_t)row_e * ldo + nb_e + e] = ok ? v[e] : __builtin_nanf("");
                    }
                } else {
                    // columns 16 jt + 4 kq .. + 3 = elements 4 (kq & 1) .. + 3 of lane (r16, 2 (jt & 1) + (kq >> 1)) of
                    // k-step jt >> 1 of the next layer
                    const int jt = 2 * g + jl_e;
                    const unsigned tag = (unsigned)((epoch << 5) | (li + 1));
                    const int ln = r16 + 16 * (2 * (jt & 1) + (kq >> 1));
                    u64* o = nxt + ((size_t)(it_e * kMaxKS + (jt >> 1)) * 8 + 4 * (kq & 1)) * 64 + ln;
#pragma unroll
                    for (int e = 0; e < 4; ++e) st_w(o + e * 64, make_word(v[e], tag));
                }
            }
        }
    }
    // the last workgroup to leave advances the epoch (and zeroes the words when it wraps)
    __syncthreads();
    if (tid == 0) {
        const int left = __hip_atomic_fetch_add(hdr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = left == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last_s) {
        const int done = hdr[0] + 1 == kEpochs ? 0 : hdr[0] + 1;
        if (done == 0) {
            for (size_t i = tid; i < kBufWords; i += blockDim.x) { st_w(buf0 + i, 0); st_w(buf1 + i, 0); }
        }
        __syncthreads();
        // the fail word is per launch: the last workgroup out moves it to the sticky report word (read and cleared by
        // hd_chain_failed) and clears it, so a launch that timed out poisons its own outputs only
        if (tid == 0) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { hdr[33] = 1; hdr[32] = 0; }
            hdr[0] = done; hdr[1] = 0;
        }
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
        HD_REQUIRE(a.epi != 2 || (i >= 1 && a.d_in == a.d_out),
                   "hd_linear_chain_small_f32: layer %d: a residual layer keeps its width and is not the first", i);
        HD_REQUIRE(i == 0 || a.d_in == layers[i - 1].d_out, "hd_linear_chain_small_f32: layer %d: d_in != previous d_out", i);
        HD_REQUIRE(i == n_layers - 1 || a.d_out % 32 == 0,
                   "hd_linear_chain_small_f32: layer %d: inner widths must be multiples of 32", i);
        ch.l[i] = Layer{(const uint4*)a.wimg, (const float*)a.bias, a.d_in, a.d_out, a.epi, a.add_emb};
        any_emb = any_emb || a.add_emb;
    }
    HD_REQUIRE(ldx >= layers[0].d_in && ldo >= layers[n_layers - 1].d_out, "hd_linear_chain_small_f32: leading dimension too small");
    HD_REQUIRE(!any_emb || (emb && emb_idx && emb_rows >= 1), "hd_linear_chain_small_f32: embedding table / indices missing");
    char* p = (char*)workspace;
    int* hdr = (int*)p;
    u64* buf0 = (u64*)(p + 256);
    u64* buf1 = buf0 + kBufWords;
    hipError_t e;
    // row tiles per group: 1 up to 256 rows, 2 beyond
    int RT = (int)(((rows + 15) / 16 + kGroups - 1) / kGroups);
    const float* xp = (const float*)x;
    const float* ep = (const float*)emb;
    float* op = (float*)out;
    int rows_i = (int)rows, emb_rows_i = (int)(emb_rows < 0x7fffffff ? emb_rows : 0x7fffffff);
    const char* wh = getenv("HD_CHAIN_TEST_WITHHOLD");                  // tests: force the time-out path (see the kernel)
    int withhold = wh && wh[0] == '1';
    void* args[] = {&xp, &ldx, &rows_i, &ch, &ep, &emb_idx, &emb_rows_i, &op, &ldo, &hdr, &buf0, &buf1, &RT, &withhold};
    // An ordinary launch on the caller's stream by default; HD_CHAIN_SMALL_LAUNCH=coop asks for the cooperative one.  The
    // cooperative launch goes through a queue of its own: two cross-queue hand-overs per call (10 + 28 us around a 175 us
    // kernel), and when that queue is the process's fifth or later (GPU_MAX_HW_QUEUES = 4) the hardware queues are
    // oversubscribed and every hand-over costs ~45 us: the sampler step measured 0.33 instead of 0.20 ms whenever the first
    // cooperative launch came after other streams had been created (profiles/r03_coop_order_trace.txt); the ordinary launch
    // measures 0.18 ms in either order.  Residency: the grid must fit the chip at once (checked here with the occupancy
    // query, else the cooperative launch is used); 256 workgroups of one wavefront per SIMD fit an idle MI355X twice over.
    // Beside other kernels a workgroup may start late -- the others wait for its data, other kernels finish, it starts --
    // and two launches of this kernel at once on one GPU remain forbidden (hypdiff.h); every wait is bounded, so a starved
    // launch ends in NaN outputs + hd_chain_failed, never in a hang.
    static const bool plain = [] {
        const char* v = getenv("HD_CHAIN_SMALL_LAUNCH");
        if (v && v[0] == 'c') return false;
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(linear_chain_small), 256, 0) != hipSuccess)
            return false;
        return (long)per_cu * prop.multiProcessorCount >= (long)kGroups * kG;
    }();
    if (plain) {
        linear_chain_small<<<dim3(kGroups * kG), dim3(256), 0, (hipStream_t)st>>>(xp, ldx, rows_i, ch, ep, emb_idx, emb_rows_i, op,
                                                                                  ldo, hdr, buf0, buf1, RT, withhold);
        return check_launch("hd_linear_chain_small_f32");
    }
    e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(linear_chain_small), dim3(kGroups * kG), dim3(256), args,
                                   0, (hipStream_t)st);
    if (e != hipSuccess) {
        set_error("hd_linear_chain_small_f32: cooperative launch failed: %s", hipGetErrorString(e));
        return HD_EHIP;
    }
    return HD_OK;
}

// 1 if a wait timed out in any launch on this workspace since the last query (the outputs of THAT launch are NaN; later
// launches are clean: the kernel clears its fail word when it leaves and keeps a report word for this query); clears the
// report.  Reads device memory: synchronises the stream once.  The launch itself never synchronises.
extern "C" int hd_chain_failed(void* workspace, void* st) {
    int f = 0;
    if (!workspace) return 0;
    if (hipMemcpyAsync(&f, (const char*)workspace + 132, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)st) != hipSuccess) return -1;
    if (f != 0 && hipMemsetAsync((char*)workspace + 132, 0, 4, (hipStream_t)st) != hipSuccess) return -1;
    return f != 0;
}
