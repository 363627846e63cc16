// A stack of nn.Linear (+ bias, + exact-erf GELU) layers over MANY rows as ONE persistent launch on the bf16 matrix
// cores: the mlp denoiser of models/direction_diffusion.py:56-69 (D+1 -> 800 -> 900 -> 800 -> 700 -> D over all
// latent rows of a sampling step), and the two stacks of its training step (forward with the pre-activations kept;
// the input-gradient GEMMs, epilogue 3).  Arithmetic, limb images and K order are those of csrc/gemm_limbs.hip
// (fp32-faithful 3-limb split, 6 exact limb products per fp32 product, 4 wavefronts, one LDS buffer, LDS-DMA
// staging) -- the results are bit-identical to one launch of it per layer; what is new is the schedule and the tile:
// 128 x 128 (a wavefront 64 x 64) instead of 128 x 64, because the K loop is bound by the bytes staged per MFMA, not
// by the MFMAs (measured: see the note above the kernel), with a 128 x 64 form for ragged last column tiles.
//
// An MLP is row-parallel: layer l + 1 of a 128-row block needs layer l of THAT row block only.  So instead of five
// launches, each a whole number of tile rounds on 512 workgroup slots with a ramp and a drain, all tiles of the chain
// (3968 at 15 777 rows) form one stream of
// work items (layer, row block, column tile) that the resident workgroups pull from device-side queues.  The only
// synchronisation is a counter per (layer, row block) that the tiles of a row block raise when their part of the
// limb image is in memory; a tile of the next layer polls it before it stages its A operand.
//
//   * Eight queues, one per XCD; row block rb belongs to queue rb % 8, so the producer and the consumers of a
//     row block's activations normally sit on one XCD.  A queue holds its items layer-major, then row block, then
//     column tile (N fastest: the tiles running at the same time share A panels through that XCD's L2).  A
//     workgroup pulls from the queue of the XCD it runs on (HW_REG_XCC_ID) and, when that is empty, from the others
//     (work stealing: the XCDs do not finish together: 124 row blocks are 15.5 per XCD).
//   * No deadlock whatever the residency: an item depends only on items EARLIER in its own queue, items are handed
//     out in queue order, and a workgroup holds at most its current item and the one it pulled ahead.  Layer-major
//     order puts ~100 items of the same queue between a tile and its consumers (64 run at once per XCD), so the
//     poll is satisfied the first time for ~80 % of the tiles; the others resolve their successor after the tile, at
//     the cost of an un-prefetched first slab.  Every wait is bounded (2 s of wall clock): on a time-out the
//     fail word is raised, every other wait ends at once, the launch runs to its end and the last workgroup fills
//     the last layer's outputs with NaN (fp32) / bf16 NaN limbs (image).
//   * Hand-off (MI355X_MICROARCH.md, "Valid forms"): the limb image is written with 16-byte `sc1` (write-through)
//     stores -- a lane pair's two 8-byte halves of a chunk are brought into one lane by v_permlane16_swap --
//     every storing wave waits vmcnt(0), the workgroup meets at a barrier, one lane adds to the counter
//     (agent scope).  The consumer: one lane polls the counter (agent-scope load), issues the agent acquire
//     (buffer_inv sc1), waits for it, the workgroup meets at a barrier, then LDS-DMA loads.  A region of an image is
//     written once per launch and read only after its counter is complete.
//   * Everything a tile needs to know about its successor is fetched while it multiplies, in its LAST five K slabs
//     (an item reserved early would sit on its workgroup while others go idle at the end of the launch): the queue
//     pull (a returning atomic), its decode, the counter poll (an atomic add of 0), the acquire, the hand-over to the
//     other wavefronts -- one step per slab, each consuming what the previous one left in flight (one register,
//     read after the vmcnt(0) that every slab starts with), each issued BEHIND that slab's staging loads, so no
//     wavefront waits on a round trip and no load is delayed.  The successor's first slab then streams into LDS under
//     the last multiply of the current tile, as in the per-layer kernel.
//   * Nothing this launch writes stays dirty in an L2: the fp32 output goes out write-through as well, so the launch
//     does not end on a cache write-back.  The last workgroup out is found in two levels (8 groups, then one word):
//     512 adds to one word are ~6 us when the workgroups finish together.
//   * The workspace (queue heads, counters) is zeroed by the LAST workgroup to leave, so the next launch on the
//     same stream finds it clean; the caller zero-fills it once when it allocates it.
#include "hd_common.cuh"
#include "hd_gelu.cuh"
#include "../../include/hypdiff.h"
#include <stdlib.h>
#include <type_traits>

using namespace hd;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxLayers = 8;
constexpr int kTile = 4 * 3 * 128;               // 16-byte chunks of one (row block, K slab) tile of a limb image
constexpr int BNT = 128;                         // wide tile 128 x 128, 2 x 2 wavefronts of 64 x 64; narrow tile 128 x 64
constexpr int NJ = BNT / 32;                     // 16-column accumulators per wavefront and 16-row group (wide tile)
constexpr int CH_A = 4 * 3 * 128, CH_W = 4 * 3 * BNT, CH_BUF = CH_A + CH_W;
constexpr int QA = 6, QW = 6;                    // 1 KiB LDS-DMA pieces per wavefront and slab (narrow tile: W pieces of two wavefronts only)
// workspace words (unsigned): queue head q at q * 32 (a 128-byte line each), then fail, exited, then the counters
constexpr int WS_FAIL = 256, WS_EXITED = 257, WS_REPORT = 258, WS_STAT = 260, WS_EXIT_GROUP = 288, WS_DONE = 288 + 256;   // WS_STAT..+3: slow paths, polls that failed (flag 8)
constexpr uint64_t kTimeoutTicks = 200000000ull; // 2 s of the 100 MHz wall clock

struct Layer {
    const uint4* W;          // limb image of the weight (N, K)
    const float* bias;       // N floats or null
    uint4* out_img;          // limb image of the output (M, N), or null
    float* out_f32;          // fp32 output (M, N) with row pitch ldc, or null
    float* z_out;            // fp32 pre-activation (M, N) with row pitch ldc, or null (training forward)
    const float* aux;        // epi 3: fp32 z of the layer below (M, N) with row pitch ldc
    int N, nslab, nslab_out, epi, ldc;
    int tiles_w, tiles_64;   // column tiles of a row block cut wide (128; the last one narrow when <= 64 columns are left) / narrow (64)
    int narrow_from;         // row blocks from this one on are cut into narrow tiles (tiles_m: none, 0: all)
};
struct ChainArgs {
    Layer L[kMaxLayers];
    const uint4* A0;         // limb image of the input (M, K_0)
    unsigned* ws;
    int n_layers, M, tiles_m;
    int delay_ticks;         // start-up stagger of every second workgroup (0 = none)
    unsigned long long* dbg; // HD_MLP_CHAIN_DEBUG_PTR: 8 wall-clock stamps per workgroup (probes), or null
    int gang;                // wide row blocks of a queue handed out in gangs of this many (decode)
    int flags;               // debugging (HD_MLP_CHAIN_FLAGS): 4 = no pull-ahead (every successor resolved blocking), 8 = count slow paths
};

__device__ __forceinline__ unsigned pk_bf16(float x, float y) {
    const f32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ void split4(const float v[4], u32x2& h, u32x2& m, u32x2& l) {
    h.x = pk_bf16(v[0], v[1]);
    h.y = pk_bf16(v[2], v[3]);
    const float r0 = v[0] - bf_lo(h.x), r1 = v[1] - bf_hi(h.x), r2 = v[2] - bf_lo(h.y), r3 = v[3] - bf_hi(h.y);
    m.x = pk_bf16(r0, r1);
    m.y = pk_bf16(r2, r3);
    l.x = pk_bf16(r0 - bf_lo(m.x), r1 - bf_hi(m.x));
    l.y = pk_bf16(r2 - bf_lo(m.y), r3 - bf_hi(m.y));
}
__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// 16-byte write-through store (sc1): the bytes are in memory, not in this XCD's L2, once vmcnt has counted it.
// The s_nop belongs to the store: a VALU write of the data registers needs a wait state after a store of more than 8
// bytes, and the compiler's hazard recogniser does not see into inline assembly (measured: without it the 128-wide
// build returned a wrong accumulator register in lanes 12-15 of every row in 143 of 150 launches).
__device__ __forceinline__ void store16_sc1(uint4* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// lanes L (even 16-lane row) and L + 16 (odd row) hold halves (a, b) of two different chunks; afterwards the even row
// holds both halves of its chunk in (a, b) and the odd row both halves of its own: v_permlane16_swap exchanges the
// odd rows of the first operand with the even rows of the second
__device__ __forceinline__ void pair_swap(u32x2& a, u32x2& b) {
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    a = u32x2{r0.x, r1.x};
    b = u32x2{r0.y, r1.y};
}
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Item { int q, idx; };

// queue q holds the row blocks q, q + 8, ...: nq of them, all layers
__device__ __forceinline__ int queue_blocks(int q, int tiles_m) { return q < tiles_m ? (tiles_m - q + 7) >> 3 : 0; }

// Column tiles of a layer: row blocks below the layer's `narrow_from` are cut into wide tiles (tw of them, numbered 0 ..),
// the row blocks from there on into narrow ones (t64 of them, numbered kNarrow + 0 ..).  tab = {tw[l], t64[l], narrow_from[l]}
// (three runs of kMaxLayers) plus the eight queue lengths, in LDS: indexed by values that live in vector registers (thread
// 0's successor bookkeeping), where an index into the kernel arguments would be a blocking global load.
constexpr int kNarrow = 1 << 20;
constexpr int TAB_TW = 0, TAB_T64 = kMaxLayers, TAB_NF = 2 * kMaxLayers, TAB_QLEN = 3 * kMaxLayers, TAB_GANG = 3 * kMaxLayers + 8, TAB_SIZE = 3 * kMaxLayers + 9;
// wide row blocks of layer l in queue q (which holds nq row blocks)
__device__ __forceinline__ int wide_blocks(const int* tab, int l, int q, int nq) {
    const int n = (tab[TAB_NF + l] - q + 7) >> 3;
    return n < 0 ? 0 : (n > nq ? nq : n);
}
__device__ __forceinline__ int queue_items(const int* tab, int n_layers, int tiles_m, int q) {
    const int nq = queue_blocks(q, tiles_m);
    int n = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int nw = wide_blocks(tab, l, q, nq);
        n += nw * tab[TAB_TW + l] + (nq - nw) * tab[TAB_T64 + l];
    }
    return n;
}
__device__ __forceinline__ void decode(const int* tab, int n_layers, int tiles_m, int q, int idx, int& l, int& rb, int& tn) {
    const int nq = queue_blocks(q, tiles_m);
    int nw = 0;
    for (l = 0;; ++l) {
        nw = wide_blocks(tab, l, q, nq);
        const int items = nw * tab[TAB_TW + l] + (nq - nw) * tab[TAB_T64 + l];
        if (idx < items || l + 1 >= n_layers) break;
        idx -= items;
    }
    const int tw = tab[TAB_TW + l], t64 = tab[TAB_T64 + l];
    int j;
    // wide row blocks in gangs of G: the tiles of a gang are handed out column tile by column tile (tn, then the row block
    // within the gang), so that the G tiles that read one slice of the W image start together (G = 1: N fastest)
    const int G = tab[TAB_GANG], gsz = G * tw, full = nw / G * gsz;
    if (idx < full) {
        const int grp = idx / gsz, r = idx - grp * gsz;
        tn = r / G;
        j = grp * G + (r - tn * G);
    } else if (idx < nw * tw) {
        const int r = idx - full;
        j = r / tw;
        tn = r - j * tw;
        j += nw / G * G;
    } else {
        const int r = idx - nw * tw;
        j = r / t64;
        tn = kNarrow + r - j * t64;
        j += nw;
    }
    rb = q + 8 * j;
}
// tiles of (layer l, row block rb): what the counter of that row block must reach
__device__ __forceinline__ int tiles_of(const int* tab, int l, int rb) {
    return rb < tab[TAB_NF + l] ? tab[TAB_TW + l] : tab[TAB_T64 + l];
}

// one lane: the counter of (layer l - 1, row block rb) has reached that row block's tile count, or the launch has failed
__device__ __forceinline__ bool dep_ready(const ChainArgs& P, const int* tab, int l, int rb) {
    if (l == 0) return true;
    return ld_agent(P.ws + WS_DONE + (l - 1) * P.tiles_m + rb) >= (unsigned)tiles_of(tab, l - 1, rb);
}

// Measured and not kept (same box, 501 us per chain as it is): the nine staging pieces of a wavefront dealt between the
// MFMA groups of the slab instead of issued ahead of them -- 523 us (two pieces ahead of each of the first groups) and 551 us
// (one per group): the in-kernel phase probe (-DHD_CHAIN_PHASE_PROBE, tools/probes/chain_stamps.py) has wave 0 spend ~40 %
// of a slab issuing the pieces, but a wave stalled on a piece inside its MFMA stream only lands the slab later.  Timing
// experiments (-DHD_CHAIN_EXPERIMENT) put the staging + barrier + fragment-read pipeline alone at ~420 us of the 500.
#ifndef HD_CHAIN_WO
#define HD_CHAIN_WO 2              // write-out: 0 = bias loads per column pair, scalar GELU; 1 = a tile's bias loads in one batch;
                                   // 2 = 1 + GELU on the packed fp32 pipe (same bits).  Same-box A/B in the bench step
                                   // (tools/probes/bench_ab.sh, twice): 183.6 / 184.1 -> 184.3 / 185.1 -> 185.5 / 184.9 TF
#endif
#ifndef HD_CHAIN_PRIO
#define HD_CHAIN_PRIO 0            // s_setprio: 1 = raised while the staging pieces are issued, 2 = during the MFMAs.  Measured
                                   // 458 / 464 us against 460-463 (tools/probes/chain_ab.py): no effect; neither has a start
                                   // stagger of a CU's second workgroup (HD_MLP_CHAIN_DELAY_NS 400 / 1000 / 3000: 462 / 466 / 460)
#endif
#ifndef HD_CHAIN_EXPERIMENT
#define HD_CHAIN_EXPERIMENT 0      // timing experiments only (results are wrong): 1 = two staging pieces of nine,
#endif                             // 2 = 8 MFMAs of 48, 3 = 6 fragment reads of 18, 4 = the multiplies as 48 v_mfma_f32_32x32x16_bf16, 5 = four of the six A pieces (the bytes of an fp32 A operand)
                                   // (4, measured on one box against the kernel as it is: 551-569 against 454-467 us per chain, with one
                                   // accumulator after the other as with the four in turn -- the 32-row shape, half as many instructions of
                                   // twice the length for the same fragments, is a fifth SLOWER here, so the 16x16x32 form stays)
                                   // (5, same way: 440-447 against 453-459 us -- a sixth fewer staged bytes buy 2-4 % at most, before the
                                   // VALU work of splitting an fp32 A operand in registers is paid: not pursued)
__device__ __forceinline__ void mlp_chain_body(const ChainArgs& P) {
    __shared__ uint4 lds[CH_BUF];
    __shared__ int nxt_s[8];                                  // successor: {queue, index, ready, layer, row block, column tile}
    __shared__ int tab[TAB_SIZE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int a_off = kq * 128 + wr * 64 + r16;
    unsigned* const ws = P.ws;
    const int M = P.M, tiles_m = P.tiles_m;
    // HW_REG_XCC_ID (id 20), bits 3:0: the XCD this workgroup runs on; locality only, never correctness
    const int my_xcd = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);

    if (threadIdx.x < kMaxLayers) {
        const bool used = threadIdx.x < (unsigned)P.n_layers;
        tab[TAB_TW + threadIdx.x] = used ? P.L[threadIdx.x].tiles_w : 1;
        tab[TAB_T64 + threadIdx.x] = used ? P.L[threadIdx.x].tiles_64 : 1;
        tab[TAB_NF + threadIdx.x] = used ? P.L[threadIdx.x].narrow_from : 0;
    }
    __syncthreads();
    if (threadIdx.x < 8) tab[TAB_QLEN + threadIdx.x] = queue_items(tab, P.n_layers, P.tiles_m, threadIdx.x);
    if (threadIdx.x == 8) tab[TAB_GANG] = P.gang;
    __syncthreads();
    // ---- thread 0's private scheduling state ----
    int curq = my_xcd, tried = 0;                             // queue being pulled from, queues found empty so far
    auto qlen = [&](int q) { return tab[TAB_QLEN + q]; };
    // blocking pull (start of the launch, end of a queue): next item of the first non-empty queue from curq on
    auto pull_blocking = [&]() -> Item {
        unsigned head[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) head[k] = ld_agent(ws + k * 32);     // eight loads in flight: one round trip
        while (tried < 8) {
            const int n = qlen(curq);
            unsigned hq = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) hq = k == curq ? head[k] : hq;
            if (n > 0 && hq < (unsigned)n) {
                const int idx = (int)__hip_atomic_fetch_add(ws + curq * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (idx < n) return Item{curq, idx};
            }
            curq = (curq + 1) & 7;
            ++tried;
        }
        return Item{-1, -1};
    };
    // blocking wait for an item's dependency, then the agent acquire
    auto wait_dep = [&](int l, int rb) {
        if (l == 0) return;
        const uint64_t t0 = wall_clock64();
        while (!dep_ready(P, tab, l, rb)) {
            if (ld_agent(ws + WS_FAIL)) break;
            if (wall_clock64() - t0 > kTimeoutTicks) {
                __hip_atomic_store(ws + WS_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    if (P.delay_ticks > 0 && (blockIdx.x >= gridDim.x / 2)) {   // optional stagger of the second workgroup of a CU
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)P.delay_ticks) __builtin_amdgcn_s_sleep(16);
    }

#define HD_STAMP(k) if (P.dbg && tid == 0) P.dbg[blockIdx.x * 8 + (k)] = wall_clock64();
    HD_STAMP(0)
    // ---- first item ----
    if (tid == 0) {
        const Item it = pull_blocking();
        if (it.q >= 0) {
            int l, rb, tn;
            decode(tab, P.n_layers, tiles_m, it.q, it.idx, l, rb, tn);
            wait_dep(l, rb);
        }
        nxt_s[0] = it.q; nxt_s[1] = it.idx; nxt_s[2] = 1;
    }
    __syncthreads();
    int cq = __builtin_amdgcn_readfirstlane(nxt_s[0]), cidx = __builtin_amdgcn_readfirstlane(nxt_s[1]);
    int seq = 0;                                              // tiles this workgroup has finished (debugging)
    (void)seq;
    int done_l = -1, done_rb = 0;                             // finished tile whose counter add is still owed

    // Tile kinds: 0 = wide (128 x 128, a wavefront 64 x 64); 1 / 2 = narrow (128 x 64, a wavefront 64 x 32) on rows
    // 0..63 / 64..127 of a 128-row block of the W image.  A narrow tile needs only the W pieces of that half -- piece
    // parity = wavefront parity, so two wavefronts stage W and two do not -- half the fragments and half the MFMAs.
    // Narrow tiles are the last column tile of a wide layer when at most 64 columns are left, and every tile of a layer
    // the host marks narrow (the last layer when it has too few wide tiles to fill the chip evenly at the end).
#define HD_STAGE(as_, ws_, t, kind_)                                                        \
    {                                                                                       \
        const uint4* as = (as_) + (size_t)(t) * kTile;                                      \
        const uint4* wsrc = (ws_) + (size_t)(t) * kTile;                                    \
        _Pragma("unroll") for (int q = 0; q < QA; ++q) {                                    \
            if (HD_CHAIN_EXPERIMENT == 1 && q) continue;                                    \
            if (HD_CHAIN_EXPERIMENT == 5 && q >= 4) continue;                               \
            const int piece = q * 4 + wave;                                                 \
            glds16(as + piece * 64, lds + piece * 64);                                      \
        }                                                                                   \
        if ((kind_) == 0 || ((kind_) - 1) == (wave & 1)) {                                  \
            _Pragma("unroll") for (int q = 0; q < QW; ++q) {                                \
                if (HD_CHAIN_EXPERIMENT == 1 && q) continue;                                \
                const int piece = q * 4 + wave;                                             \
                glds16(wsrc + piece * 64, lds + CH_A + piece * 64);                         \
            }                                                                               \
        }                                                                                   \
    }
    auto tile_kind = [&](int l, int tn) { return tn < kNarrow ? (P.L[l].N - tn * BNT <= 64 ? 1 : 0) : 1 + (tn & 1); };
    auto a_src = [&](int l, int rb) {
        const uint4* base = l == 0 ? P.A0 : P.L[l - 1].out_img;
        return base + (size_t)rb * P.L[l].nslab * kTile + lane;
    };
    auto w_src = [&](int l, int tn) { return P.L[l].W + (size_t)(tn < kNarrow ? tn : (tn - kNarrow) >> 1) * P.L[l].nslab * kTile + lane; };

    HD_STAMP(1)
    int stamped = 0;
    if (cq >= 0) {
        int l, rb, tn;
        decode(tab, P.n_layers, tiles_m, cq, cidx, l, rb, tn);
        l = __builtin_amdgcn_readfirstlane(l); rb = __builtin_amdgcn_readfirstlane(rb); tn = __builtin_amdgcn_readfirstlane(tn);
        const uint4 *a_cur = a_src(l, rb), *w_cur = w_src(l, tn);
        int kind = tile_kind(l, tn);
        HD_STAGE(a_cur, w_cur, 0, kind)
        for (;;) {
            const Layer& Ly = P.L[l];
            const int nslab = Ly.nslab;
            const bool narrow = kind != 0;
            const int w_off = CH_A + kq * BNT + (narrow ? (kind - 1) * 64 + wc * 32 : wc * 64) + r16;
            const unsigned long long tile_t0 = P.dbg ? wall_clock64() : 0ull;
            f32x4 acc[4][NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#if HD_CHAIN_EXPERIMENT == 4
            f32x16 acc16[2][2];                                // timing only: the slab's multiplies as 48 x 32x32x16 (wrong results)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc16[i][j][r] = 0.f;
#endif
            // successor, fetched asynchronously by thread 0 while this tile multiplies (nslab >= 4)
            // thread 0 only.  `pend` is the ONE register that receives what the op issued in the previous slab returns;
            // it is read at the start of the next slab's step, after the vmcnt(0) at the top of that slab, so the compiler
            // never has a reason to wait for it where it is issued.
            int p_idx = -1, p_l = 0, p_rb = 0, p_tn = 0, p_ready = 0, p_state = 0;   // 1 pull issued, 2 pulled, 3 poll issued
            unsigned pend = 0;
            int nq_ = -1, nidx_ = -1, nready_ = 0, nl_ = 0, nrb_ = 0, ntn_ = 0;   // every thread, after the hand-over slab
            // pull, decode, poll, acquire, hand-over in the LAST five slabs: an item is reserved only a few microseconds
            // before it starts, so the launch does not end on workgroups that sit on a reserved tile while others are idle
            const bool async = nslab >= 5 && !(P.flags & 4);
            const int t_poll = nslab - 3, t_pull = nslab - 5;
            const uint4 *a_n = nullptr, *w_n = nullptr;
#ifdef HD_CHAIN_PHASE_PROBE
            unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, pt = wall_clock64(), pn;
            const unsigned long long probe_c0 = clock64(), probe_w0 = pt;      // core clock against the 100 MHz wall clock
#define HD_PH(k) { pn = wall_clock64(); ph[k] += pn - pt; pt = pn; }
#else
#define HD_PH(k)
#endif
            // (Measured and not kept: a counted wait -- vmcnt(number of image stores of the last write-out) -- at a tile's
            // first slab, so that it waits for the slab's data and not for the write-through stores behind it, with the
            // counter add moved one slab down: 462 against 456-463 us, no difference; the stores have been acknowledged by
            // the time the write-out's arithmetic is over.)
            for (int t = 0; t < nslab; ++t) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                HD_PH(0)
#ifdef HD_CHAIN_PHASE_PROBE
                // slab-by-slab timeline of this workgroup's 4th and 5th tile (steady state): time at which each slab's data
                // had landed (wave 0)
                if (P.dbg && tid == 0 && (seq == 3 || seq == 4) && t < 32)
                    P.dbg[512 * 8 + 512 * 24 * 4 + 512 * 8 + ((size_t)blockIdx.x * 2 + (seq - 3)) * 32 + t] = wall_clock64();
#endif
                if (async && t == t_poll + 2 && tid == 0) {    // hand the successor to the other wavefronts
                    nxt_s[0] = curq; nxt_s[1] = p_idx; nxt_s[2] = p_ready; nxt_s[3] = p_l; nxt_s[4] = p_rb; nxt_s[5] = p_tn;
                }
                __builtin_amdgcn_s_barrier();                   // slab t has landed; every wave's older stores are done
                HD_PH(1)
                uint4 fa[4][3], fw[NJ][3];
#pragma unroll
                for (int li = 0; li < (HD_CHAIN_EXPERIMENT == 3 ? 1 : 3); ++li) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i][li] = lds[a_off + li * 4 * 128 + i * 16];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fw[j][li] = lds[w_off + li * 4 * BNT + j * 16];
                    if (NJ == 4 && !narrow) {
#pragma unroll
                        for (int j = 2; j < NJ; ++j) fw[j][li] = lds[w_off + li * 4 * BNT + j * 16];
                    }
                }
                if (async && t == t_poll + 2) {
                    nq_ = __builtin_amdgcn_readfirstlane(nxt_s[0]);
                    nidx_ = __builtin_amdgcn_readfirstlane(nxt_s[1]);
                    nready_ = __builtin_amdgcn_readfirstlane(nxt_s[2]);
                    nl_ = __builtin_amdgcn_readfirstlane(nxt_s[3]);
                    nrb_ = __builtin_amdgcn_readfirstlane(nxt_s[4]);
                    ntn_ = __builtin_amdgcn_readfirstlane(nxt_s[5]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (HD_CHAIN_EXPERIMENT == 3) {
#pragma unroll
                    for (int li = 1; li < 3; ++li) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) fa[i][li] = fa[i][0];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) fw[j][li] = fw[j][0];
                    }
                }
                HD_PH(2)
                __builtin_amdgcn_s_barrier();                   // every wave holds its fragments: the buffer is free
                HD_PH(3)
                // Thread 0's step of the successor protocol, on the four slabs that have one (a scalar test elsewhere).
                // Nothing may sit between the vmcnt(0) at the top of a slab and its fragment reads, and nothing may delay
                // this slab's staging loads: the value the previous step left in flight is taken over (one move, after
                // which the compiler has no reason to wait again), the loads go out, and only then does the step run --
                // under the loads' latency -- and issue its own operation behind them.
                // (As the code stands the step runs AHEAD of the staging loads, and the slab with the decode is 0.5 us longer
                // than the others (tools/probes/chain_stamps.py); moving the step behind the loads and the decode behind
                // the MFMAs measured 0.477 against 0.481 of the peak in the bench step, same box, twice: not kept.)
                const bool act = async && (t == t_pull || t == t_pull + 1 || t == t_poll || t == t_poll + 1);
                unsigned got = 0;
                if (act && tid == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(got) : "v"(pend));
                // where the next slab comes from: this tile's next slab, or the successor's first (none: nothing is staged)
                const bool more = t + 1 < nslab;
                const bool do_stage = more || (nidx_ >= 0 && nready_);
                if (!more && do_stage) { a_n = a_src(nl_, nrb_); w_n = w_src(nl_, ntn_); }
                const uint4* const sa = more ? a_cur + (size_t)(t + 1) * kTile : a_n;
                const uint4* const sw = more ? w_cur + (size_t)(t + 1) * kTile : w_n;
                if (t == 0 && done_l >= 0) {
                    // the previous tile's image stores were complete in every wavefront at this slab's first barrier (each
                    // waited vmcnt(0) before it): raise its row block's counter
                    if (tid == 0 && !((P.flags & 64) && done_l == 0 && done_rb == 0))   // 64: test knob, see the host side
                        __hip_atomic_fetch_add(ws + WS_DONE + done_l * tiles_m + done_rb, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    done_l = -1;
                }
                if (act && tid == 0) {
                    unsigned* addr = nullptr;
                    unsigned inc = 0;
                    if (t == t_pull) {
                        if (tried < 8 && qlen(curq) > 0) { addr = ws + curq * 32; inc = 1; p_state = 1; }
                    } else if (t == t_pull + 1) {
                        if (p_state == 1 && got < (unsigned)qlen(curq)) {
                            p_idx = (int)got;
                            decode(tab, P.n_layers, tiles_m, curq, p_idx, p_l, p_rb, p_tn);
                            p_state = 2;
                        }
                    } else if (t == t_poll) {
                        if (p_state == 2) {
                            if (p_l == 0) p_ready = 1;
                            else { addr = ws + WS_DONE + (p_l - 1) * tiles_m + p_rb; p_state = 3; }   // poll = add 0
                        }
                    } else {
                        if (p_state == 3 && got >= (unsigned)tiles_of(tab, p_l - 1, p_rb)) {
                            p_ready = 1;
                            // agent acquire for the successor's A operand: invalidate this CU's L1; the vmcnt(0) at the top of
                            // the next slab waits for it, the barrier behind that orders the other wavefronts' loads
                            asm volatile("buffer_inv sc1" ::: "memory");
                        }
                    }
                    if (addr) pend = __hip_atomic_fetch_add(addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                HD_PH(4)
#define HD_MM(wl, al) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[j][wl]), \
                                                                  __builtin_bit_cast(bf16x8, fa[i][al]), c, 0, 0, 0);
                if (do_stage) {
                    const int kind_next = more ? kind : tile_kind(nl_, ntn_);
                    if (HD_CHAIN_PRIO == 1) __builtin_amdgcn_s_setprio(3);
                    HD_STAGE(sa, sw, 0, kind_next)
                    if (HD_CHAIN_PRIO == 1) __builtin_amdgcn_s_setprio(0);
                }
                if (HD_CHAIN_PRIO == 2) __builtin_amdgcn_s_setprio(2);
#define HD_MMS(J0, J1)                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            _Pragma("unroll") for (int j = J0; j < J1; ++j) {                                                    \
                f32x4 c = acc[i][j];                                                                             \
                if (HD_CHAIN_EXPERIMENT != 2) { HD_MM(0, 2) HD_MM(2, 0) HD_MM(1, 1) HD_MM(0, 1) HD_MM(1, 0) }    \
                HD_MM(0, 0)                                                                                      \
                acc[i][j] = c;                                                                                   \
            }
#if HD_CHAIN_EXPERIMENT == 4
#define HD_MM32(ks, wl, al) _Pragma("unroll") for (int I = 0; I < 2; ++I) _Pragma("unroll") for (int J = 0; J < 2; ++J)        \
        acc16[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[2 * J + ks][wl]),                \
                                                              __builtin_bit_cast(bf16x8, fa[2 * I + ks][al]), acc16[I][J], 0, 0, 0);
                // the four accumulators in turn: no multiply waits for the one before it
                HD_MM32(0, 0, 2) HD_MM32(0, 2, 0) HD_MM32(0, 1, 1) HD_MM32(0, 0, 1) HD_MM32(0, 1, 0) HD_MM32(0, 0, 0)
                HD_MM32(1, 0, 2) HD_MM32(1, 2, 0) HD_MM32(1, 1, 1) HD_MM32(1, 0, 1) HD_MM32(1, 1, 0) HD_MM32(1, 0, 0)
#undef HD_MM32
#else
                HD_MMS(0, 2)
                if (NJ == 4 && !narrow) { HD_MMS(2, NJ) }
#endif
                if (HD_CHAIN_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#undef HD_MMS
#undef HD_MM
            }
            HD_PH(5)
#ifdef HD_CHAIN_PHASE_PROBE
            if (P.dbg && tid == 0)
            {
                for (int k = 0; k < 6; ++k) P.dbg[512 * 8 + 512 * 24 * 4 + (size_t)blockIdx.x * 8 + k] += ph[k];
                P.dbg[512 * 8 + 512 * 24 * 4 + (size_t)blockIdx.x * 8 + 6] += clock64() - probe_c0;
                P.dbg[512 * 8 + 512 * 24 * 4 + (size_t)blockIdx.x * 8 + 7] += wall_clock64() - probe_w0;
            }
#endif
            if (!stamped) HD_STAMP(2)
            if (P.dbg && tid == 0 && seq < 24) {
                unsigned long long* e = P.dbg + 512 * 8 + ((size_t)blockIdx.x * 24 + seq) * 4;
                e[0] = tile_t0; e[1] = wall_clock64(); e[3] = ((unsigned long long)l << 32) | ((unsigned)rb << 8) | (unsigned)tn;
            }
            // ---- write-out of the C^T tile: a lane owns row m and columns nb .. nb + 3 of accumulator (i, j) ----
            {
                const int m0 = rb * 128, n0 = tn < kNarrow ? tn * BNT + (narrow ? wc * 32 : wc * 64) : (tn - kNarrow) * 64 + wc * 32;
                const int N = Ly.N, epi = Ly.epi, ldc = Ly.ldc;
                const float* __restrict__ bias = Ly.bias;
                uint4* __restrict__ CL = Ly.out_img;
                float* __restrict__ C = Ly.out_f32;
                float* __restrict__ Z = Ly.z_out;
                const float* __restrict__ AUX = Ly.aux;
                // (the host checks that every fp32 pointer is 16-byte aligned)
                const bool vec = (ldc & 3) == 0;
                // this lane's chunk after the pair swap: even 16-lane rows keep the even accumulator of a pair (chunk
                // kc = kq / 2), odd rows the odd one (kc = 2 + kq / 2)
                const int kc_mine = (kq & 1) * 2 + (kq >> 1);
                // HD_CHAIN_WO >= 1: the tile's bias values of this lane, all loads in flight at once
                float bq4[NJ][4];
                if (HD_CHAIN_WO >= 1) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = n0 + j * 16 + 4 * kq + e;
                            bq4[j][e] = (bias && n < N && !(narrow && j >= 2)) ? bias[n] : 0.f;
                        }
                }
#pragma unroll
                for (int jp = 0; jp < NJ / 2; ++jp) {
                    if (jp && narrow) continue;
                    // the pair's 32 columns are exactly K slab `so` of the output image
                    const int nq = n0 + jp * 32;
                    const int so = nq >> 5;
                    const bool img = CL != nullptr && so < Ly.nslab_out;
                    uint4* const cl_base = img ? CL + ((size_t)rb * Ly.nslab_out + so) * kTile + kc_mine * 128 : nullptr;
                    float bq[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = nq + j * 16 + 4 * kq + e;
                            if (HD_CHAIN_WO >= 1) bq[j][e] = bq4[2 * jp + j][e];
                            else bq[j][e] = (bias && n < N) ? bias[n] : 0.f;
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int mloc = wr * 64 + i * 16 + r16;
                        const int m = m0 + mloc;
                        const bool row_ok = m < M;
                        float v[2][4];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int nb = nq + j * 16 + 4 * kq;
                            float x[4];
#pragma unroll
#if HD_CHAIN_EXPERIMENT == 4
                            for (int e = 0; e < 4; ++e) x[e] = acc16[i >> 1][(2 * jp + j) >> 1][((i & 1) * 2 + ((2 * jp + j) & 1)) * 4 + e] + bq[j][e];
#else
                            for (int e = 0; e < 4; ++e) x[e] = acc[i][2 * jp + j][e] + bq[j][e];
#endif
                            if (Z && row_ok) {                  // training: the pre-activation is kept for the backward pass
                                float* zo = Z + (size_t)m * ldc + nb;
                                if (vec && nb + 3 < N) {
                                    __builtin_nontemporal_store(f32x4{x[0], x[1], x[2], x[3]}, reinterpret_cast<f32x4*>(zo));
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (nb + e < N) zo[e] = x[e];
                                }
                            }
                            if (epi == 3) {                     // backward: dz_below = g * gelu'(z_below)
                                float r[4] = {0.f, 0.f, 0.f, 0.f};
                                if (row_ok) {
                                    const float* rp = AUX + (size_t)m * ldc + nb;
                                    if (vec && nb + 3 < N) {
                                        const float4 q = *reinterpret_cast<const float4*>(rp);
                                        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
                                    } else {
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            if (nb + e < N) r[e] = rp[e];
                                    }
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) x[e] *= gelu_erf_grad(r[e]);
                            } else if (epi == 1) {
#pragma unroll
                                for (int e = 0; e < 4; e += 2) {
                                    if (HD_CHAIN_WO >= 2) {
                                        const hd_f32x2 g = gelu_erfc2(hd_f32x2{x[e], x[e + 1]});
                                        x[e] = g.x; x[e + 1] = g.y;
                                    } else {
                                        x[e] = gelu_erfc(x[e]); x[e + 1] = gelu_erfc(x[e + 1]);
                                    }
                                }
                            }
                            // columns past N are the zero padding of the next layer's K; rows past M are written as zeros
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[j][e] = (row_ok && nb + e < N) ? x[e] : 0.f;
                        }
                        if (img) {
                            u32x2 h0, m0_, l0, h1, m1, l1;
                            split4(v[0], h0, m0_, l0);
                            split4(v[1], h1, m1, l1);
                            pair_swap(h0, h1);
                            pair_swap(m0_, m1);
                            pair_swap(l0, l1);
                            // even rows now hold {own j = 0 half, partner's j = 0 half} in (x0, x1); odd rows {partner's
                            // j = 1 half, own j = 1 half}: in both cases (first, second) = columns (8 kc .. + 3, + 4 .. + 7)
                            uint4* o = cl_base + mloc;
                            store16_sc1(o, u32x4{h0.x, h0.y, h1.x, h1.y});
                            store16_sc1(o + 4 * 128, u32x4{m0_.x, m0_.y, m1.x, m1.y});
                            store16_sc1(o + 8 * 128, u32x4{l0.x, l0.y, l1.x, l1.y});
                        }
                        if (C && row_ok) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int nb = nq + j * 16 + 4 * kq;
                                float* o = C + (size_t)m * ldc + nb;
                                if (vec && nb + 3 < N) {
                                    // write-through as well: nothing of this launch is left dirty in the L2s when it ends
                                    store16_sc1(reinterpret_cast<uint4*>(o), __builtin_bit_cast(u32x4, f32x4{v[j][0], v[j][1], v[j][2], v[j][3]}));
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (nb + e < N) o[e] = v[j][e];
                                }
                            }
                        }
                    }
                }
            }
            if (!stamped) { HD_STAMP(3) stamped = 1; }
            if (P.dbg && tid == 0 && seq < 24) P.dbg[512 * 8 + ((size_t)blockIdx.x * 24 + seq) * 4 + 2] = wall_clock64();
            ++seq;
            done_l = (l + 1 < P.n_layers) ? l : -1;            // the last layer feeds nobody
            done_rb = rb;
            if (async && nidx_ >= 0 && nready_) {               // successor already streaming into LDS
                cq = nq_; cidx = nidx_;
                l = nl_; rb = nrb_; tn = ntn_;
                a_cur = a_n; w_cur = w_n;
                kind = tile_kind(l, tn);
                continue;
            } else {
                // slow path: end of a queue, a dependency that was not ready at the poll, or a tile of < 4 slabs
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    if (done_l >= 0 && !((P.flags & 64) && done_l == 0 && done_rb == 0))
                        __hip_atomic_fetch_add(ws + WS_DONE + done_l * tiles_m + done_rb, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    if (P.flags & 8) {
                        atomicAdd(ws + WS_STAT, 1u);
                        if (async && p_idx >= 0) atomicAdd(ws + WS_STAT + 1, 1u);
                        if (async && p_state == 0) atomicAdd(ws + WS_STAT + 2, 1u);
                    }
                    Item it;
                    if (async && p_idx >= 0) it = Item{curq, p_idx};          // pulled, its dependency was not ready
                    else it = pull_blocking();
                    if (it.q >= 0) {
                        int l2, rb2, tn2;
                        decode(tab, P.n_layers, tiles_m, it.q, it.idx, l2, rb2, tn2);
                        wait_dep(l2, rb2);
                    }
                    nxt_s[0] = it.q; nxt_s[1] = it.idx; nxt_s[2] = 1;
                }
                done_l = -1;
                __syncthreads();
                cq = __builtin_amdgcn_readfirstlane(nxt_s[0]);
                cidx = __builtin_amdgcn_readfirstlane(nxt_s[1]);
                if (cq < 0) break;
                decode(tab, P.n_layers, tiles_m, cq, cidx, l, rb, tn);
                l = __builtin_amdgcn_readfirstlane(l); rb = __builtin_amdgcn_readfirstlane(rb);
                tn = __builtin_amdgcn_readfirstlane(tn);
                a_cur = a_src(l, rb); w_cur = w_src(l, tn);
                kind = tile_kind(l, tn);
                HD_STAGE(a_cur, w_cur, 0, kind)
                continue;
            }
        }
    }
#undef HD_STAGE
    // ---- leave: the last workgroup out cleans the workspace for the next launch (and reports a failed one) ----
    HD_STAMP(4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HD_STAMP(5)
    if (tid == 0) {
        // last one out, found in two levels (512 adds to one word are ~6 us of serialised atomics when the workgroups
        // finish together): eight groups by blockIdx, the last of a group adds to the top word
        const unsigned g = blockIdx.x & 7u, gsize = (gridDim.x - g + 7u) >> 3;
        unsigned last = 0;
        if (__hip_atomic_fetch_add(ws + WS_EXIT_GROUP + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
            last = __hip_atomic_fetch_add(ws + WS_EXITED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
        }
        nxt_s[0] = (int)last;
        nxt_s[1] = (int)ld_agent(ws + WS_FAIL);
    }
    __syncthreads();
    HD_STAMP(6)
    if (P.dbg && tid == 0) P.dbg[blockIdx.x * 8 + 7] = ((unsigned long long)seq << 8) | (unsigned)my_xcd;
    if (nxt_s[0]) {
        const int words = WS_DONE + (P.n_layers > 1 ? (P.n_layers - 1) * tiles_m : 0);
        // the fail word is per launch (cleared here: the next launch on this workspace is clean); what hd_mlp_chain_failed
        // reads is the report word, raised below and cleared only by that query
        for (int i = tid; i < words; i += 256)
            if (i != WS_REPORT && !(i >= WS_STAT && i < WS_STAT + 4)) __hip_atomic_store(ws + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nxt_s[1]) {                                         // a wait timed out somewhere: make the failure loud
            if (tid == 0) __hip_atomic_store(ws + WS_REPORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const Layer& Lz = P.L[P.n_layers - 1];
            if (Lz.out_f32)
                for (size_t i = tid; i < (size_t)M * Lz.ldc; i += 256)
                    if ((int)(i % Lz.ldc) < Lz.N) Lz.out_f32[i] = __builtin_nanf("");
            if (Lz.out_img) {                                   // every limb a bf16 NaN
                const size_t chunks = (size_t)tiles_m * Lz.nslab_out * kTile;
                for (size_t i = tid; i < chunks; i += 256) Lz.out_img[i] = make_uint4(0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u);
            }
        }
    }
}

// The launch of a whole stack, and the same body under a second name for a chain of ONE layer (a single GEMM over many
// rows, ops.gemm_limbs: HypLinear's 512 x 512 in the bench step): two kernels in a rocprofv3 trace, so that the average
// duration and the counters of the five-layer launch are its own.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void mlp_chain(const ChainArgs P) { mlp_chain_body(P); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_limbs_wide(const ChainArgs P) { mlp_chain_body(P); }

}  // namespace

extern "C" int64_t hd_mlp_chain_workspace_bytes(int64_t rows, int n_layers) {
    if (rows < 0 || n_layers < 1 || n_layers > kMaxLayers) return -1;
    const int64_t tiles_m = (rows + 127) / 128;
    return (WS_DONE + (int64_t)(n_layers - 1) * tiles_m) * 4 + 256;
}

extern "C" int hd_mlp_chain_limbs_f32(const void* A0, int64_t rows, const hd_mlp_chain_layer* layers, int n_layers,
                                      void* workspace, int64_t workspace_bytes, void* st) {
    HD_REQUIRE(n_layers >= 1 && n_layers <= kMaxLayers && layers, "hd_mlp_chain_limbs_f32: between 1 and %d layers", kMaxLayers);
    HD_REQUIRE(rows >= 0 && rows < (1LL << 31), "hd_mlp_chain_limbs_f32: bad row count");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(A0 && is_aligned16(A0) && workspace && is_aligned16(workspace), "hd_mlp_chain_limbs_f32: null or misaligned pointer");
    HD_REQUIRE(workspace_bytes >= hd_mlp_chain_workspace_bytes(rows, n_layers), "hd_mlp_chain_limbs_f32: workspace too small");
    ChainArgs P{};
    P.A0 = (const uint4*)A0;
    P.ws = (unsigned*)workspace;
    P.n_layers = n_layers;
    P.M = (int)rows;
    P.tiles_m = (int)((rows + 127) / 128);
    struct Knobs { int n_cu, delay_ns, flags, reload, per_cu, tail, grid, gang; };
    static Knobs knobs = [] {
        Knobs k{256, 0, 0, 0, 2, 0, 0, 1};
        if (const char* e = getenv("HD_MLP_CHAIN_PER_CU")) k.per_cu = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_DELAY_NS")) k.delay_ns = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_FLAGS")) k.flags = atoi(e);
        if (const char* e = getenv("HD_GEMM_TUNE_RELOAD")) k.reload = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_TAIL")) k.tail = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_GRID")) k.grid = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_GANG")) k.gang = atoi(e);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            k.n_cu = prop.multiProcessorCount;
        return k;
    }();
    if (knobs.reload) {
        if (const char* e = getenv("HD_MLP_CHAIN_DELAY_NS")) knobs.delay_ns = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_FLAGS")) knobs.flags = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_PER_CU")) knobs.per_cu = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_TAIL")) knobs.tail = atoi(e);
        if (const char* e = getenv("HD_MLP_CHAIN_GANG")) knobs.gang = atoi(e);
    }
    if (knobs.tail < 0 || knobs.tail > 100) knobs.tail = 0;
    const int per_cu = knobs.per_cu >= 1 && knobs.per_cu <= 2 ? knobs.per_cu : 2;
    const int64_t grid_full = (int64_t)knobs.n_cu * per_cu / 8 * 8 > 8 ? (int64_t)knobs.n_cu * per_cu / 8 * 8 : 8;
    int64_t items = 0;
    for (int i = 0; i < n_layers; ++i) {
        const hd_mlp_chain_layer& s = layers[i];
        HD_REQUIRE(s.N >= 1 && s.K >= 1 && s.N < (1LL << 31) && s.K < (1LL << 31), "hd_mlp_chain_limbs_f32: layer %d: bad shape", i);
        HD_REQUIRE(i == 0 || s.K == layers[i - 1].N, "hd_mlp_chain_limbs_f32: layer %d: K = %lld but layer %d has N = %lld", i,
                   (long long)s.K, i - 1, (long long)layers[i - 1].N);
        HD_REQUIRE(s.W && is_aligned16(s.W), "hd_mlp_chain_limbs_f32: layer %d: null or misaligned weight image", i);
        HD_REQUIRE(s.epi == 0 || s.epi == 1 || s.epi == 3, "hd_mlp_chain_limbs_f32: layer %d: epilogue must be 0 (none), 1 (GELU) or 3 (backward)", i);
        HD_REQUIRE(s.epi != 3 || (s.aux && !s.bias), "hd_mlp_chain_limbs_f32: layer %d: epilogue 3 needs aux and no bias", i);
        HD_REQUIRE(is_aligned16(s.out_f32) && is_aligned16(s.z_out) && is_aligned16(s.aux),
                   "hd_mlp_chain_limbs_f32: layer %d: fp32 pointers must be 16-byte aligned", i);
        const bool last = i == n_layers - 1;
        HD_REQUIRE(last || (s.out_image && is_aligned16(s.out_image)),
                   "hd_mlp_chain_limbs_f32: layer %d: an inner layer needs a (16-byte aligned) output image", i);
        HD_REQUIRE(!last || s.out_f32 || s.out_image, "hd_mlp_chain_limbs_f32: the last layer needs an output");
        const bool f32io = s.out_f32 || s.z_out || s.epi == 3;
        HD_REQUIRE(!f32io || s.ldc >= s.N, "hd_mlp_chain_limbs_f32: layer %d: leading dimension too small", i);
        HD_REQUIRE((int64_t)rows * (f32io ? s.ldc : 1) < (1LL << 40), "hd_mlp_chain_limbs_f32: output too large");
        Layer& L = P.L[i];
        L.W = (const uint4*)s.W;
        L.bias = (const float*)s.bias;
        L.out_img = (uint4*)s.out_image;
        L.out_f32 = (float*)s.out_f32;
        L.z_out = (float*)s.z_out;
        L.aux = s.epi == 3 ? (const float*)s.aux : nullptr;
        L.N = (int)s.N;
        L.nslab = (int)((s.K + 31) / 32);
        // Wide tiles everywhere by default.  HD_MLP_CHAIN_TAIL = t cuts the last t percent of the LAST layer's row blocks
        // (the items handed out last) into narrow tiles of half the length, so that the workgroups leave closer together
        // (with wide tiles they leave over a span of ~70 us).  Measured at t = 0 / 15 / 25 / 33 / 50 / 100: 452 / 455 / 452 /
        // 451 / 448 / 451 us per chain (tools/probes/chain_tail.py, alternating, same box): nothing outside the noise --
        // narrow tiles multiply at 85 % of the wide rate, which eats what the shorter drain gives -- so t = 0.
        // flag 32 = every layer narrow (the 128 x 64 tiling of csrc/gemm_limbs.hip): 521 us.
        L.tiles_w = (int)((s.N + BNT - 1) / BNT);
        L.tiles_64 = (int)((s.N + 63) / 64);
        L.narrow_from = P.tiles_m;
        if (knobs.flags & 32) L.narrow_from = 0;
        else if (last && n_layers > 1)
            L.narrow_from = (int)(P.tiles_m - (int64_t)P.tiles_m * knobs.tail / 100);
        items += (int64_t)L.narrow_from * L.tiles_w + (int64_t)(P.tiles_m - L.narrow_from) * L.tiles_64;
        L.nslab_out = (int)((s.N + 31) / 32);
        L.epi = s.epi;
        L.ldc = (int)(f32io ? s.ldc : s.N);
    }
    HD_REQUIRE(items < (1LL << 30), "hd_mlp_chain_limbs_f32: too many tiles");
    P.delay_ticks = knobs.delay_ns / 10;
    P.flags = knobs.flags;
    // test knob, read on every call: the tiles of (layer 0, row block 0) never report completion, so the tiles of layer 1 that
    // wait for them run into the bound (tests/test_mlp_chain.py: the launch ends, its outputs are NaN, the next one is clean)
    if (const char* e = getenv("HD_MLP_CHAIN_TEST_WITHHOLD")) P.flags |= atoi(e) ? 64 : 0;
    P.gang = knobs.gang >= 1 && knobs.gang <= 16 ? knobs.gang : 1;
    P.dbg = nullptr;
    if (knobs.reload)
        if (const char* e = getenv("HD_MLP_CHAIN_DEBUG_PTR")) P.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
    const int64_t total = items;
    // The kernel's registers (252) admit two workgroups per CU, its LDS (48 KB) three; unused dynamic LDS pads the
    // footprint so that exactly `per_cu` workgroups fit a CU (160 KB) and the grid fills the chip evenly.
    // HD_MLP_CHAIN_GRID: fewer workgroups than fill the chip (a multiple of 8), leaving slots to kernels of other streams.
    // In the bench's pipelined two-stream step (same box, tools/probes/bench_ab.sh): 512 / 496 / 480 / 448 / 384 workgroups
    // -> 0.637-0.643 / 0.653 / 0.631 / 0.629 / 0.634 ms per step while the chain itself goes 194 -> 192 / 189 / 181 / 168 TF:
    // a 1-2 % gain of the pipelined step for a 3-7 % loss of the kernel and of the one-stream step, so the default stays full.
    int64_t grid = knobs.grid >= 8 && knobs.grid < grid_full ? knobs.grid / 8 * 8 : grid_full;
    if (grid > total) grid = total;
    const size_t static_lds = CH_BUF * 16 + 64;
    const size_t want = per_cu == 2 ? 56 * 1024 : 100 * 1024;
    const size_t pad = want > static_lds ? want - static_lds : 0;
    static bool attr_set[64] = {};                  // per device: the attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("hd_mlp_chain_limbs_f32: no current device");
        return HD_EHIP;
    }
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                                100 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_limbs_wide), hipFuncAttributeMaxDynamicSharedMemorySize,
                                100 * 1024) != hipSuccess) {
            set_error("hd_mlp_chain_limbs_f32: hipFuncSetAttribute failed");
            return HD_EHIP;
        }
        attr_set[dev] = true;
    }
    if (n_layers == 1)
        gemm_limbs_wide<<<dim3((unsigned)grid), dim3(256), pad, (hipStream_t)st>>>(P);
    else
        mlp_chain<<<dim3((unsigned)grid), dim3(256), pad, (hipStream_t)st>>>(P);
    return check_launch("hd_mlp_chain_limbs_f32");
}

// 1 if a wait of any launch on this workspace timed out since the last query (the fp32 output of that launch is NaN);
// clears the flag.  Synchronises the stream.
extern "C" int hd_mlp_chain_failed(void* workspace, void* st) {
    if (!workspace) return 0;
    unsigned flag = 0;
    if (hipMemcpyAsync(&flag, (unsigned*)workspace + WS_REPORT, 4, hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)st) != hipSuccess) return -1;
    if (flag) {
        if (hipMemsetAsync((unsigned*)workspace + WS_REPORT, 0, 4, (hipStream_t)st) != hipSuccess) return -1;
    }
    return flag ? 1 : 0;
}
