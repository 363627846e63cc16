/* hypdiff.h -- C ABI of libhypdiff_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hyperbolic encode + direction-diffusion hot path of
 * RingBDStack/HypDiff.  The reference is pure Python on PyTorch and has no FFI of
 * its own; each entry point below replaces the chain of aten launches behind ONE
 * reference method (cited as file:line relative to the reference root).  The
 * Python classes in hypdiff_amd/ bind these symbols with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch);
 *    the library never allocates, frees or retains them, and never synchronises;
 *  - matrices are dense row-major and contiguous: (rows, dim), row stride = dim;
 *  - `_f32` / `_f64` suffix = element type of the row data;
 *  - `stream` is a hipStream_t (NULL = default stream); work is only enqueued;
 *  - return 0 on success, non-zero on failure (1 bad argument, 2 HIP error,
 *    3 unsupported); hd_last_error() returns the message of the calling thread's
 *    last failure.  No C++ exception crosses the boundary;
 *  - thread-safe and re-entrant: no global mutable state besides the per-thread
 *    error string.
 */
#ifndef HYPDIFF_H
#define HYPDIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Curvature bundle, computed by the host with the reference's precision rules:
 * c, c**0.5, K = 1/c, K**0.5 (a python-float c in double; a 1-element tensor c in
 * the tensor's dtype -- layers/hyp_layers.py:35, models/base_models.py:24). */
typedef struct hd_curv {
    double c, sqrt_c, K, sqrt_K;
} hd_curv;

/* activation codes for the tangent-space activation of HypAct */
enum { HD_ACT_IDENTITY = 0, HD_ACT_RELU = 1, HD_ACT_TANH = 2, HD_ACT_SIGMOID = 3, HD_ACT_ELU = 4,
       HD_ACT_LEAKY_RELU = 5 };

int hd_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf */
size_t hd_last_error(char* buf, size_t cap);

#define HD_DECL2(name, ...) int name##_f32(__VA_ARGS__); int name##_f64(__VA_ARGS__);

/* ---- Poincare ball, manifolds/poincare.py ------------------------------------- */
/* proj :42-47 */
HD_DECL2(hd_pb_proj, const void* x, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* expmap0 :73-77 */
HD_DECL2(hd_pb_expmap0, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* logmap0 :79-83.  Artanh (utils/math_utils.py:30-36): fp64 rows evaluate the reference's literal fp64 formula;
 * fp32 rows use the cancellation-free fp32 form 0.5 * log1p(2z / (1 - z)) -- the same values as the reference's
 * fp64 detour to <= 4 fp32 ulp (golden vectors), without two software fp64 logs per row */
HD_DECL2(hd_pb_logmap0, const void* p, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* proj(expmap0(proj_tan0(x))) -- the input lift of models/encoders.py:115-119 in one pass */
HD_DECL2(hd_pb_expmap0_proj, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* mobius_add :85-91 */
HD_DECL2(hd_pb_mobius_add, const void* x, const void* y, void* out, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* expmap(u, p) :55-64 -- first pointer is the base point p, second the tangent u */
HD_DECL2(hd_pb_expmap, const void* p, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* logmap(p1, p2) :66-71 */
HD_DECL2(hd_pb_logmap, const void* p1, const void* p2, void* out, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* the per-row loops of models/direction_diffusion.py:305-306,350-352 (logmap at the row's
 * cluster centre) and :328-330 (expmap at it): base point = centres[labels[row]] */
HD_DECL2(hd_pb_logmap_at, const void* centres, const int32_t* labels, int64_t n_centres, const void* x,
         void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
HD_DECL2(hd_pb_expmap_at, const void* centres, const int32_t* labels, int64_t n_centres, const void* u,
         void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* hd_pb_logmap_at_f64 on fp32 rows x (centres fp64, out fp64): diffusion.forward :350-352 runs on x.double() of the
 * encoder's fp32 output; the rows are widened on load -- the same bits, without the fp64 copy of x. */
int hd_pb_logmap_at_f32_f64(const void* centres, const int32_t* labels, int64_t n_centres, const void* x, void* out,
                            int64_t rows, int64_t dim, hd_curv c, void* stream);
/* sqdist :25-31 -> out (rows,) */
HD_DECL2(hd_pb_sqdist, const void* p1, const void* p2, void* out, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* tail of mobius_matvec :93-102 after mx = x @ W^T (hd_gemm): needs |x| per row (hd_rownorm).
 * flags bit0: follow with proj (HypLinear.forward, layers/hyp_layers.py:100-101);
 * flags bit1: then logmap0, i.e. emit the tangent vector HypAgg consumes (:134). */
HD_DECL2(hd_pb_matvec_epilogue, const void* mx, const void* xnorm, void* out, int64_t rows, int64_t dim,
         hd_curv c, int flags, void* stream)
/* HypAct.forward, layers/hyp_layers.py:170-173 */
HD_DECL2(hd_pb_hyp_act, const void* x, void* out, int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out,
         int act, void* stream)

/* Graph-VAE tail, models/hyp_layers.py:45-61 (HNNLayer = HypLinear :172-206 -> HypNorm :298-319 -> HypAct
 * :280-296; Poincare ball, c = 1 in the reference) as ONE pass over mx = x @ W^T (rows, dim).  flags bit0:
 * HypLinear's tail h = expmap0(mx), b = ptransp0(h, bias), h = expmap(u = h, p = b) (the reference's argument
 * order); bit1: HypNorm h = expmap0(LayerNorm(logmap0(h))) with ln_w / ln_b (dim,) and eps; bit2: HypAct
 * h = expmap0(act(logmap0(h))) (no proj).  bias (dim,) must be 16-byte aligned. */
HD_DECL2(hd_pb_vae_hnn_tail, const void* mx, const void* bias, const void* ln_w, const void* ln_b, void* out,
         int64_t rows, int64_t dim, hd_curv c, double eps, int flags, int act, void* stream)

/* ---- Hyperboloid, manifolds/hyperboloid.py (column 0 = time coordinate) --------- */
/* proj :43-52 */
HD_DECL2(hd_hb_proj, const void* x, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* proj_tan0 :64-68 */
HD_DECL2(hd_hb_proj_tan0, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* expmap0 :90-101 (includes its trailing proj) */
HD_DECL2(hd_hb_expmap0, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* logmap0 :103-113.  Arcosh (utils/math_utils.py:57-63): literal fp64 formula for fp64 rows; fp32 rows use
 * log1p((z - 1) + sqrt((z - 1)(z + 1))) in fp32 (<= 4 fp32 ulp from the reference's fp64 detour) */
HD_DECL2(hd_hb_logmap0, const void* x, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* proj(expmap0(proj_tan0(x))) -- input lift, models/encoders.py:115-119 */
HD_DECL2(hd_hb_expmap0_proj, const void* u, void* out, int64_t rows, int64_t dim, hd_curv c, void* stream)
/* sqdist :35-41 -> out (rows, 1) */
HD_DECL2(hd_hb_sqdist, const void* x, const void* y, void* out, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* tail of mobius_matvec :120-123 after mu = logmap0(x) @ W^T: expmap0; flags as above */
HD_DECL2(hd_hb_linear_epilogue, const void* mu, void* out, int64_t rows, int64_t dim, hd_curv c, int flags,
         void* stream)
HD_DECL2(hd_hb_hyp_act, const void* x, void* out, int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out,
         int act, void* stream)

/* HypLinear on the hyperboloid with a narrow input, ONE launch (layers/hyp_layers.py:98-103,
 * manifolds/hyperboloid.py:103-123): out = tail(logmap0(x, c) @ W^T) with tail = expmap0 (flags&1: proj,
 * flags&2: logmap0) as in hd_hb_linear_epilogue.  x (rows, d_in) fp32 contiguous, Wt = W^T (d_in, d_out) fp32
 * contiguous, 16-byte aligned; d_in <= 64, d_out <= 512 and a multiple of 4.  The contraction is an fp32 fma
 * chain over k (exact fp32, no limbs). */
/* hd_hb_linear_epilogue_f32 for rows of up to 512 columns (a multiple of 4; 16-byte aligned): four rows per
 * wavefront, the row's scalar chain evaluated once per wavefront instead of once per lane. */
int hd_hb_linear_tail_rows_f32(const void* mu, void* out, int64_t rows, int64_t dim, hd_curv c, int flags,
                               void* stream);
int hd_hb_linear_smallk_f32(const void* x, const void* Wt, void* out, int64_t rows, int64_t d_in, int64_t d_out,
                            hd_curv c, int flags, void* stream);
/* The same layer with the contraction on the bf16 matrix cores (fp32-faithful 3-limb split, as hd_gemm_limbs_f32):
 * d_in <= 32, d_out <= 512 and a multiple of 16.  Wimg = hd_smallk_pack_w_f32 of W (d_out, d_in) fp32 row-major
 * (row stride ld): hd_smallk_image_bytes(d_out) bytes, 16-byte aligned, built once per weight version. */
int64_t hd_smallk_image_bytes(int64_t d_out);
int hd_smallk_pack_w_f32(const void* W, int64_t ld, int64_t d_out, int64_t d_in, void* img, void* stream);
int hd_hb_linear_smallk_bf16x3_f32(const void* x, const void* Wimg, void* out, int64_t rows, int64_t d_in,
                                   int64_t d_out, hd_curv c, int flags, void* stream);

/* ---- backward of the row maps (autograd of the encoder as lp_train.py:66-82 trains it) ------
 * Each is the vector-Jacobian product of one forward entry point above: `saved` row + incoming
 * gradient -> outgoing gradient, clamps and torch.where branches differentiated the way torch
 * autograd differentiates the reference's expression, Artanh / Arcosh by their hand-written
 * backward (utils/math_utils.py:38-41,65-68).  which: 0 proj, 1 expmap0, 2 logmap0,
 * 3 proj(expmap0(proj_tan0(.))) (input lift). */
HD_DECL2(hd_pb_map_bwd, int which, const void* x, const void* dy, void* dx, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
HD_DECL2(hd_hb_map_bwd, int which, const void* x, const void* dy, void* dx, int64_t rows, int64_t dim, hd_curv c,
         void* stream)
/* VJP of hd_pb_matvec_epilogue: d_mx (rows, dim) and d|x| (rows,) from mx, |x| and d_out */
HD_DECL2(hd_pb_linear_tail_bwd, const void* mx, const void* xnorm, const void* dout, void* dmx, void* dxnorm,
         int64_t rows, int64_t dim, hd_curv c, int flags, void* stream)
/* VJP of hd_hb_linear_epilogue */
HD_DECL2(hd_hb_linear_tail_bwd, const void* mu, const void* dout, void* dmu, int64_t rows, int64_t dim, hd_curv c,
         int flags, void* stream)
/* tail of HypAgg (+HypAct) on a stored aggregate s = spmm(adj, xt): the training forward runs
 * hd_spmm_csr + this so that s is kept for the backward pass; and its VJP */
HD_DECL2(hd_pb_agg_tail, const void* s, void* out, int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out, int act,
         int fuse_act, void* stream)
HD_DECL2(hd_hb_agg_tail, const void* s, void* out, int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out, int act,
         int fuse_act, void* stream)
HD_DECL2(hd_pb_agg_tail_bwd, const void* s, const void* dout, void* ds, int64_t rows, int64_t dim, hd_curv c_in,
         hd_curv c_out, int act, int fuse_act, void* stream)
HD_DECL2(hd_hb_agg_tail_bwd, const void* s, const void* dout, void* ds, int64_t rows, int64_t dim, hd_curv c_in,
         hd_curv c_out, int act, int fuse_act, void* stream)
/* out = g + coef[row] * x / |x|: the gradient of the row norm (hd_rownorm) added to g */
HD_DECL2(hd_rownorm_bwd_add, const void* x, const void* coef, const void* g, void* out, int64_t rows, int64_t dim,
         void* stream)

/* VJPs of the ball's two-point maps called on their own (what torch autograd gives for manifolds/poincare.py:25-31,
 * 55-71, 85-91; _lambda_x is computed on .data there and is a constant here).  Kernel operand order as the forward
 * entry points: which = 0 mobius_add(x = a, y = b), 1 expmap(u = b, p = a), 2 logmap(p1 = a, p2 = b).  g = d loss / d out
 * (rows, dim); da, db (rows, dim) receive both gradients in one pass (da may alias g).  hd_*_sqdist_bwd: g (rows,) is
 * the gradient of the one scalar per row of Manifold.sqdist (poincare.py:25-31, hyperboloid.py:35-41, clamp at 50
 * included). */
HD_DECL2(hd_pb_map2_bwd, int which, const void* a, const void* b, const void* g, void* da, void* db, int64_t rows,
         int64_t dim, hd_curv c, void* stream)
HD_DECL2(hd_pb_sqdist_bwd, const void* a, const void* b, const void* g, void* da, void* db, int64_t rows, int64_t dim,
         hd_curv c, void* stream)
HD_DECL2(hd_hb_sqdist_bwd, const void* a, const void* b, const void* g, void* da, void* db, int64_t rows, int64_t dim,
         hd_curv c, void* stream)

/* LPModel.decode, models/base_models.py:152-159: sqdist(h[idx[:,0]], h[idx[:,1]]) for int64 index
 * pairs (n_pairs, 2) gathered in the kernel; fermi_dirac != 0 appends FermiDiracDecoder
 * (layers/layers.py:85-87) probs = 1/(exp((sqdist - r)/t) + 1).  out (n_pairs,) */
HD_DECL2(hd_pb_edge_decode, const void* h, const int64_t* pairs, int64_t n_rows, void* out, int64_t n_pairs,
         int64_t dim, hd_curv c, int fermi_dirac, double r, double t, void* stream)
HD_DECL2(hd_hb_edge_decode, const void* h, const int64_t* pairs, int64_t n_rows, void* out, int64_t n_pairs,
         int64_t dim, hd_curv c, int fermi_dirac, double r, double t, void* stream)

/* VJP of hd_*_edge_decode: gprob (n_pairs,) = d loss / d out; ge (2 * n_pairs, dim) receives
 * d h[pairs[e,0]] in row e and d h[pairs[e,1]] in row n_pairs + e (the caller sums the rows of
 * each node with a selection-matrix hd_spmm_csr: deterministic, no atomics). */
HD_DECL2(hd_pb_edge_decode_bwd, const void* h, const int64_t* pairs, int64_t n_rows, const void* gprob, void* ge,
         int64_t n_pairs, int64_t dim, hd_curv c, int fermi_dirac, double r, double t, void* stream)
HD_DECL2(hd_hb_edge_decode_bwd, const void* h, const int64_t* pairs, int64_t n_rows, const void* gprob, void* ge,
         int64_t n_pairs, int64_t dim, hd_curv c, int fermi_dirac, double r, double t, void* stream)

/* row L2 norms (unclamped), x.norm(dim=-1) of manifolds/poincare.py:95 -> out (rows,) */
HD_DECL2(hd_rownorm, const void* x, void* out, int64_t rows, int64_t dim, void* stream)

/* ---- HypAgg: tangent-space CSR SpMM, layers/hyp_layers.py:133-152 (use_att=0) --- */
/* COO (int64 row/col, the layout utils/data_utils.py:125-133 produces) -> CSR with int32
 * rowptr (n_rows+1) and col (nnz), entries kept in COO order inside a row.  flags[0]:
 * bit0 = input not row-sorted (sort stably on the host and call again), bit1 = index
 * out of range. */
int hd_csr_from_coo(const int64_t* row, const int64_t* col_in, int64_t nnz, int64_t n_rows, int64_t n_cols,
                    int32_t* rowptr, int32_t* col, int32_t* flags, void* stream);
/* Adjacency preprocessing on the device, utils/data_utils.py:103-133 (`process`): adj_train as a CSR
 * with ascending columns per row (int64 rowptr (n+1) / col, optional fp64 values, null = ones) ->
 * normalize(adj + I) = D^-1 (A + I) (fp64, 1/0 -> 0; normalize = 0 writes A + I unscaled -- the reference
 * adds the identity only when it normalises, so its drop-in always passes 1) as the COO the reference's
 * .tocoo() yields: row-major, columns descending inside a row (scipy's sparse product r_mat_inv.dot(mx)
 * reverses the merged row), int64 indices, fp32 values.  hd_adj_count writes every
 * output row's length (flags[0]: bit0 columns not strictly ascending, bit1 index out of range); the
 * caller prefix-sums them into out_ptr (n+1, starting at 0) and calls hd_adj_fill.
 * hd_feat_normalize: features <- normalize(features) (:106-107), fp32 out. */
int hd_adj_count(const int64_t* rowptr, const int64_t* col, int64_t n, int64_t* counts, int32_t* flags, void* stream);
int hd_adj_fill(const int64_t* rowptr, const int64_t* col, const double* val, int64_t n, const int64_t* out_ptr,
                int normalize, int64_t* out_row, int64_t* out_col, float* out_val, void* stream);
int hd_feat_normalize(const void* x, int x_dtype, int64_t rows, int64_t dim, float* out, void* stream);
/* plain torch.spmm(adj, x) :150.  long_rows (n_long entries, may be NULL/0): ids of the rows
 * with more than 512 stored entries (power-law hubs); each of them is summed by a whole
 * workgroup in 8 fixed slices instead of by one lane group. */
HD_DECL2(hd_spmm_csr, const int32_t* rowptr, const int32_t* col, const void* val, const void* x, void* out,
         int64_t rows, int64_t dim, const int32_t* long_rows, int64_t n_long, void* stream)
/* the same on a column slab of wider matrices: x rows have pitch ldx, out rows pitch ldo (elements, >= dim;
 * 16-byte loads when both are multiples of 4 (f32) / 2 (f64) and the bases are aligned).  A feature column's sum
 * does not depend on the other columns, so spmm over column slabs == spmm over whole rows bit for bit: this is
 * what lets a row-sharded HypAgg start on the first slab of the all-gathered tangent rows while the next slab is
 * still on the wire (hypdiff_amd/dist.py).  long_threshold: the threshold `long_rows` was built with (rows with MORE
 * entries; 0 = 512) -- the same one the fused aggregate of the whole graph takes (hd_*_agg, fuse_act bits 10-29), so
 * that the slab path sums every row in the same shape. */
HD_DECL2(hd_spmm_csr_strided, const int32_t* rowptr, const int32_t* col, const void* val, const void* x, int64_t ldx,
         void* out, int64_t ldo, int64_t rows, int64_t dim, const int32_t* long_rows, int64_t n_long,
         int64_t long_threshold, void* stream)
/* proj(expmap0(spmm(adj, xt))) :150-151, xt = logmap0(x) already in the tangent space (:134);
 * fuse_act bit 0 appends HypAct.forward (:170-173) with (c_in, c_out, act) in the same pass.  Bits 8-9 of fuse_act pin
 * the form of the fused row tail, which has two evaluation orders that differ by rounding: 0 = chosen from this
 * call's rows (four rows per wavefront from 8192 rows on; on the hyperboloid only without long rows), 1 = one row per
 * wavefront, 2 = four rows per wavefront.  A row shard of a larger graph passes the form the whole graph takes, so
 * that sharded and unsharded encodes agree bit for bit.  Bits 10-29 of fuse_act: the long-row threshold T the caller
 * built `long_rows` with (rows with MORE than T entries; 0 = 512).  Long rows are summed by a workgroup in 8 fixed
 * slices (bit-reproducible; differs from the sequential sum by rounding only), so T too is a property of the adjacency:
 * hypdiff_amd passes 64 for graphs of fewer than 8192 rows, where one wavefront walking a row of a few hundred neighbours
 * is the launch's critical path, and 512 otherwise; a row shard passes the whole graph's. */
HD_DECL2(hd_pb_agg, const int32_t* rowptr, const int32_t* col, const void* val, const void* xt, void* out,
         int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out, int act, int fuse_act,
         const int32_t* long_rows, int64_t n_long, void* stream)
HD_DECL2(hd_hb_agg, const int32_t* rowptr, const int32_t* col, const void* val, const void* xt, void* out,
         int64_t rows, int64_t dim, hd_curv c_in, hd_curv c_out, int act, int fuse_act,
         const int32_t* long_rows, int64_t n_long, void* stream)

/* ---- dense contractions ---------------------------------------------------------- */
/* C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]) on the exact-f32 matrix cores.
 * x @ W^T of manifolds/poincare.py:96 / hyperboloid.py:122; nn.Linear (+F.gelu) of
 * models/direction_diffusion.py:56-69.  epi: 0 none, 1 exact-erf GELU, 2 resid + GELU
 * (ddpm.py:20-22).  K, lda, ldw multiples of 4, A/W 16-byte aligned (pad with zeros). */
int hd_gemm_f32(const void* A, const void* W, const void* bias, const void* resid, void* C, int64_t M,
                int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi, void* stream);
/* Training form of hd_gemm_f32 (loss.backward() of diff.py:113-118 / diff_graphset.py:64-75 over
 * the Linear+GELU stacks of models/direction_diffusion.py:56-69 and ddpm.py:17-27).
 * epi 1 / 2 with C2 != NULL also store the pre-activation z = A W^T + bias into C2 (saved for
 * the backward pass).  epi 3 is the input-gradient form: g = A W^T (+ resid), A = dL/dz of a
 * layer and W its TRANSPOSED weight (hd_transpose_f32); C = g when C != NULL, and
 * C2 = g * gelu'(aux) when C2 != NULL (aux = saved pre-activation of the layer below), so the
 * next layer's dL/dz leaves the accumulators without a separate elementwise pass.  resid,
 * aux, C, C2 all have leading dimension ldc. */
int hd_gemm_f32_ex(const void* A, const void* W, const void* bias, const void* resid, const void* aux, void* C,
                   void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi,
                   void* stream);
/* Same contract as hd_gemm_f32_ex (operands, epilogues, alignment rules), evaluated on the bf16
 * matrix cores with fp32 accuracy: every fp32 operand is split exactly into three bf16 limbs and
 * each product is the sum of the six limb products of weight >= 2^-16, accumulated in fp32
 * (dropped terms < 2^-26 |a||w|, below one fp32 rounding).  6/16 of the exact-f32 MFMA time.
 * Replaces the same reference lines as hd_gemm_f32 (manifolds/poincare.py:96,
 * manifolds/hyperboloid.py:122, models/direction_diffusion.py:56-69, ddpm.py:17-27). */
int hd_gemm_f32_bf16x3(const void* A, const void* W, const void* bias, const void* resid, const void* aux, void* C,
                       void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi,
                       void* stream);
/* The same product with the K range split over workgroups where the tiles alone would leave the chip idle (the
 * graph-level DDPM's 200-row batches, ddpm.py:9-41): partial tiles in `workspace`
 * (hd_gemm_splitk_workspace_floats(M,N,K) floats; 0 = no split pays, workspace may be NULL), summed in split order, then
 * the same bias / epilogue.  Bit-reproducible; differs from the unsplit kernel only in the order of the fp32 sums. */
int64_t hd_gemm_splitk_workspace_floats(int64_t M, int64_t N, int64_t K);
int hd_gemm_f32_bf16x3_splitk(const void* A, const void* W, const void* bias, const void* resid, const void* aux, void* C,
                              void* C2, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int epi,
                              void* workspace, void* stream);
/* ---- the same contraction with operands pre-split into bf16 limbs ("limb images") ----
 * Limb image of a (rows, cols) fp32 matrix: 16-byte chunks of 8 bf16,
 *   chunk(rb, s, l, kc, r) = (((rb*nslab + s)*3 + l)*4 + kc)*128 + r,
 * rb = row/128, r = row%128, s = col/32, kc = (col%32)/8, l = limb (h, m, l), nslab =
 * ceil(cols/32); zero columns up to the slab boundary.  hd_limb_image_bytes gives the size.
 * hd_limb_pack_f32 builds it from row-major fp32 (weights: once per parameter version; inputs),
 * hd_limb_unpack_f32 sums the limbs back (tests).  hd_gemm_limbs_f32: A, W limb images of
 * (M,K) and (N,K); C = epi(A W^T + bias) as fp32 row-major (C, ldc) and/or as the limb image
 * of an (M,N) matrix (CL) that the next layer consumes directly; epi 0 none, 1 erf-GELU,
 * 2 resid + GELU (resid fp32 row-major, ldc).  Staging is a linear LDS-DMA copy; no split work
 * inside the contraction.  Same reference lines as hd_gemm_f32. */
/* The training forms of the same kernel (autograd of diff.py:113-118 over models/direction_diffusion.py:56-69; the
 * reference has no kernel of its own here).  Forward (epi 0 / 1): Z (optional, fp32 row-major, ldc) receives the
 * pre-activation A W^T + bias, C / CL the activated value.  Backward (epi 3): A = limb image of dz (M, N_l), W = limb
 * image of W_l^T (K_l, N_l), aux = z of the layer below (fp32, ldc): C / CL = (dz W_l) * gelu'(z_below). */
int hd_gemm_limbs_train_f32(const void* A, const void* W, const void* bias, const void* aux, void* Z, void* C, void* CL,
                            int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, void* stream);
int64_t hd_limb_image_bytes(int64_t rows, int64_t cols);
int hd_limb_pack_f32(const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* stream);
/* Several bf16x3 images in one launch (at most 16 jobs): job i packs src (rows x cols fp32, row stride ld) or, with
 * `transposed`, src^T (the cols x rows matrix W^T that the input gradient dz @ W contracts with; autograd of F.linear,
 * models/direction_diffusion.py:56-69 -- replaces hd_transpose_f32 + hd_limb_pack_f32) into `image`
 * (hd_limb_image_bytes of the packed matrix's shape). */
typedef struct hd_pack_job {
    const void* src;
    void* image;
    int64_t ld, rows, cols;
    int32_t transposed;
} hd_pack_job;
int hd_limb_pack_multi_f32(const hd_pack_job* jobs, int n_jobs, void* stream);
int hd_limb_unpack_f32(const void* in, int64_t rows, int64_t cols, void* out, int64_t ld, void* stream);
/* A stack of nn.Linear layers on a small batch (rows <= 512, widths <= 512) in ONE launch: MLP.forward /
 * Backbone.forward of ddpm.py:9-41 (config 3: 24 layers of width 512 at B = 200).  Layer i: y = x W_i^T + bias_i, then
 * epi 0 nothing, 1 erf-GELU, 2 x + GELU(y) (ddpm.py:20-22; d_in == d_out, not the first layer); add_emb != 0: y += emb[emb_idx[row]]
 * afterwards (ddpm.py:39, emb (emb_rows, d_out) fp32 contiguous; an index outside [0, emb_rows) gives a NaN row).  wimg
 * = hd_chain_pack_w_f32 of W_i (d_out, d_in) fp32 row-major, hd_chain_image_bytes(d_out, d_in) bytes, 16-byte aligned;
 * inner widths are multiples of 32.  Arithmetic as hd_gemm_limbs_f32 (3 bf16 limbs, fp32-faithful).  workspace:
 * hd_chain_workspace_bytes() bytes of device memory, 16-byte aligned, ZERO-FILLED ONCE by the caller when it is
 * allocated and from then on only passed to this function, private to the call while it runs (at most 31 layers).  The 256
 * workgroups exchange data among themselves, so they must all be resident: an ordinary launch when the occupancy query says
 * the grid fits the chip at once (HD_CHAIN_SMALL_LAUNCH=coop forces hipLaunchCooperativeKernel, which costs two cross-queue
 * hand-overs per call); never two launches of it at once on one GPU.  It never synchronises.  Every device-side
 * wait is bounded (2 s); a launch in which one timed out writes NaN outputs -- that launch only: the kernel clears its
 * fail word when it leaves, the next launch on the workspace is clean.  hd_chain_failed(workspace, stream) returns 1 if
 * that happened in any launch since the last query and clears the report; it reads device memory, i.e. it synchronises
 * the stream (once). */
typedef struct {
    const void* wimg;
    const void* bias;
    int32_t d_in, d_out, epi, add_emb;
} hd_chain_layer;
int64_t hd_chain_image_bytes(int64_t d_out, int64_t d_in);
int64_t hd_chain_workspace_bytes(void);
int hd_chain_pack_w_f32(const void* W, int64_t ld, int64_t d_out, int64_t d_in, void* img, void* stream);
int hd_linear_chain_small_f32(const void* x, int64_t ldx, int64_t rows, const hd_chain_layer* layers, int n_layers,
                              const void* emb, const int64_t* emb_idx, int64_t emb_rows, void* out, int64_t ldo,
                              void* workspace, int64_t workspace_bytes, void* stream);
int hd_chain_failed(void* workspace, void* stream);

/* One-pass producers of bf16x3 GEMM operands (no fp32 staging copy, one read of the rows): cat(x.float(),
 * t[:,None]) of models/direction_diffusion.py:57-58 (x f32/f64 by x_dtype 0/1, t as in hd_mlp_pack_input;
 * dim + 1 <= 1024), and the operands of HypLinear (dim <= 1024): mode 1 = Hyperboloid.logmap0(x, c)
 * (manifolds/hyperboloid.py:103-113, operand of :122), mode 2 = x itself with norms[row] = |x|_2
 * (manifolds/poincare.py:94-96). */
int hd_mlp_pack_input_bf16x3(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar, void* out,
                             int64_t rows, int64_t dim, void* stream);
int hd_row_pack_bf16x3_f32(const void* x, int mode, hd_curv c, void* out, void* norms, int64_t rows, int64_t dim,
                           void* stream);
int hd_gemm_limbs_f32(const void* A, const void* W, const void* bias, const void* resid, void* C, void* CL,
                      int64_t M, int64_t N, int64_t K, int64_t ldc, int epi, void* stream);
/* A stack of nn.Linear (+ bias, + exact-erf GELU) layers over MANY rows in ONE persistent launch: the mlp denoiser of
 * models/direction_diffusion.py:56-69 (fc1 .. fc5 with F.gelu between) over all latent rows of a sampling step.  Same
 * arithmetic and limb images as hd_gemm_limbs_f32, layer by layer (results are bit-identical to n_layers calls of it);
 * the tiles of all layers are one stream of work items pulled by the resident workgroups, a 128-row block flows from
 * layer to layer as soon as its own tiles are complete (csrc/gemm_chain.hip).  A0: limb image of the input (rows, K_0).
 * Layer i: W = limb image of the weight (N, K), bias (N floats or null), epi 0 none / 1 GELU / 3 backward
 * ((A W^T) * gelu'(aux): the input-gradient chain of the training step, dz_{l-1} = (dz_l W_l) gelu'(z_{l-1}), which is
 * row-parallel like the forward pass; bias must be null), z_out = optional fp32 copy of the pre-activation (what the
 * backward pass needs of the forward pass), as hd_gemm_limbs_train_f32 writes it; out_image = limb image
 * of the output (required for every layer but the last: hd_limb_image_bytes(rows, N) bytes, written by this call,
 * distinct buffers per layer), out_f32 = fp32 output with row pitch ldc (optional; the last layer needs one of the
 * two).  K of layer i must equal N of layer i - 1; at most 8 layers.  workspace: hd_mlp_chain_workspace_bytes(rows,
 * n_layers) bytes of device memory, 16-byte aligned, ZERO-FILLED ONCE by the caller when it is allocated and from then
 * on only passed to this function; private to the call while it runs (one workspace per stream).  Never synchronises.
 * Every device-side wait is bounded (2 s): after a time-out the last layer's outputs of THAT launch are NaN (its fail
 * word is cleared when the launch leaves: the next launch on the workspace is clean) and
 * hd_mlp_chain_failed(workspace, stream) -- which DOES synchronise the stream -- returns 1 once and clears the report. */
typedef struct hd_mlp_chain_layer {
    const void* W;
    const void* bias;
    void* out_image;
    void* out_f32;
    int64_t N, K, ldc;
    int32_t epi;
    void* z_out;        /* training forward: fp32 pre-activation A W^T + bias (rows x N, row pitch ldc), or null */
    const void* aux;    /* epi 3 only: fp32 z of the layer below (rows x N, row pitch ldc) */
} hd_mlp_chain_layer;
int64_t hd_mlp_chain_workspace_bytes(int64_t rows, int n_layers);
int hd_mlp_chain_limbs_f32(const void* A0, int64_t rows, const hd_mlp_chain_layer* layers, int n_layers, void* workspace,
                           int64_t workspace_bytes, void* stream);
int hd_mlp_chain_failed(void* workspace, void* stream);
/* ---- limb images of two row-scaled fp16 limbs ("f16x2"): half the matrix-core work ----
 * a * 2^ex[row] = a_h + a_l (fp16, 22 significant bits), three limb products (hh, hl, lh) per fp32
 * product on v_mfma_f32_16x16x32_f16, the row exponents removed exactly from the accumulator.
 * Image layout as above with 2 limbs: chunk(rb, s, l, kc, r) = (((rb*nslab + s)*2 + l)*4 + kc)*128 + r,
 * chunks of 8 fp16.  Every image carries ex (int32 per row) and amax (uint32 per row: bit pattern
 * of the exact max |x| of the row).  hd_limb_pack_f16x2_f32 derives both from the data.
 * hd_gemm_f16x2_f32: C = epi(A W^T + bias), epi 0 / 1 (erf-GELU), as fp32 (C, ldc) and/or as an
 * f16x2 image CL with row exponents eo taken from the bound |c_mn| <= amax_a[m] * bound_mul +
 * bound_add (caller: bound_mul >= sqrt(K) * max_n |w_n|_2, bound_add >= max |bias|) scaled into
 * [2^13, 2^14), and amax_o = exact row maxima of the values written (integer atomicMax; the
 * caller zeroes amax_o before the call).  Error: <= 3 * 2^-22 |a||w| per product, plus 2^-38 of
 * the row bound per stored element.  Same reference lines as hd_gemm_f32. */
int64_t hd_limb_image_bytes_f16x2(int64_t rows, int64_t cols);
int hd_limb_pack_f16x2_f32(const void* in, int64_t ld, int64_t rows, int64_t cols, void* out, void* ex, void* amax,
                           void* stream);
int hd_limb_unpack_f16x2_f32(const void* in, const void* ex, int64_t rows, int64_t cols, void* out, int64_t ld,
                             void* stream);
/* cat(x.float(), t[:,None]) (models/direction_diffusion.py:57-58) written directly as the f16x2 image
 * of the (rows, dim+1) denoiser input: x f32/f64 (x_dtype 0/1) row-major contiguous, t as in
 * hd_mlp_pack_input; dim + 1 <= 1536. */
int hd_mlp_pack_input_f16x2(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar, void* out,
                            void* ex, void* amax, int64_t rows, int64_t dim, void* stream);
/* GEMM operands of HypLinear written directly as f16x2 images, one pass over the rows (dim <= 1536):
 * mode 1 = Hyperboloid.logmap0(x, c) (manifolds/hyperboloid.py:103-113, the operand of :122),
 * mode 2 = x itself, with norms[row] = |x|_2 (manifolds/poincare.py:94-96). */
int hd_row_pack_f16x2_f32(const void* x, int mode, hd_curv c, void* out, void* ex, void* amax, void* norms,
                          int64_t rows, int64_t dim, void* stream);
int hd_gemm_f16x2_f32(const void* A, const void* ea, const void* amax_a, const void* W, const void* ew,
                      const void* bias, void* C, void* CL, void* eo, void* amax_o, int64_t M, int64_t N, int64_t K,
                      int64_t ldc, int epi, float bound_mul, float bound_add, void* stream);
/* Weight gradient dW[P,Q] = A[R,P]^T @ B[R,Q] (A = dL/dz, B = layer input; autograd of
 * F.linear, models/direction_diffusion.py:56-69).  The R rows are split over workgroups and
 * the partial tiles summed in slice order (bit-reproducible).  P, Q, lda, ldb multiples of 4;
 * workspace holds hd_gemm_tn_workspace_floats(R,P,Q) floats (0 = none needed). */
int64_t hd_gemm_tn_workspace_floats(int64_t R, int64_t P, int64_t Q);
int hd_gemm_tn_f32(const void* A, const void* B, void* C, int64_t R, int64_t P, int64_t Q, int64_t lda, int64_t ldb,
                   int64_t ldc, void* workspace, void* stream);
/* The same weight gradient on the bf16 matrix cores (3 bf16 limbs per operand element split in registers on the way
 * into LDS, 6 limb products per fp32 product: fp32-faithful as hd_gemm_limbs_f32); same arguments, C also 16-byte
 * aligned with ldc a multiple of 4; workspace of hd_gemm_tn_bf16x3_workspace_floats floats.  colsum_a (P floats, 16-byte
 * aligned, or NULL) also receives the bias gradient db[p] = sum_r A[r][p] in fp32, summed in a fixed order from the rows
 * the kernel stages anyway (in place of hd_colsum_f32). */
int64_t hd_gemm_tn_bf16x3_workspace_floats(int64_t R, int64_t P, int64_t Q);
int hd_gemm_tn_bf16x3_f32(const void* A, const void* B, void* C, void* colsum_a, int64_t R, int64_t P, int64_t Q,
                          int64_t lda, int64_t ldb, int64_t ldc, void* workspace, void* stream);
/* optimizer.step() of optim.Adam(diff.parameters(), lr) (diff.py:107,117; diff_graphset.py:66,74; torch defaults: no weight
 * decay, no amsgrad) for every parameter tensor in ONE launch: job i updates `count` fp32 elements at `param` from the
 * gradient / first / second moment elements [offset, offset + count) of three flat fp32 arrays (the gradient bucket of
 * the data-parallel trainer and two state arrays of the same layout).  `step` counts from 1.  At most 128 jobs.
 * step_state (3 doubles on the device, zero-filled once, or NULL): the step count lives there instead -- a call with
 * step > 0 advances it by one before the update, a call with step <= 0 (further job groups of the same step) uses it
 * as it is -- so that a HIP graph holding this call replays with the right bias corrections. */
typedef struct hd_adam_job {
    void* param;
    int64_t offset, count;
} hd_adam_job;
int hd_adam_step_f32(const hd_adam_job* jobs, int n_jobs, const void* grad_flat, void* exp_avg_flat, void* exp_avg_sq_flat,
                     double lr, double beta1, double beta2, double eps, int64_t step, void* step_state, void* stream);
/* The same weight gradient (and bias gradient) from bf16x3 limb IMAGES of A (R x P) and B (R x Q) -- the images the
 * forward and input-gradient launches of hd_gemm_limbs_train_f32 write anyway, so training keeps no fp32 copies of dz or
 * of the layer inputs.  The contraction runs over image rows (LDS-DMA + ds_read_b64_tr_b16 transposing reads); the pad
 * rows of the last 128-row block must be zero in both images (hd_limb_pack*_f32 and the limb GEMM's write-out zero them).
 * P, Q multiples of 4 with ceil(P/32), ceil(Q/32) the images' slab counts; C 16-byte aligned, contiguous when the rows
 * are split; workspace of hd_gemm_tn_bf16x3_workspace_floats(R,P,Q) floats. */
int hd_gemm_tn_limbs_f32(const void* Aimg, const void* Bimg, void* C, void* colsum_a, int64_t R, int64_t P, int64_t Q,
                         int64_t ldc, void* workspace, void* stream);
/* Several such products over the SAME R rows in one launch (at most 8 jobs): the five weight gradients of the mlp
 * denoiser's backward pass (autograd of models/direction_diffusion.py:56-69 under diff.py:113-118), which do not depend
 * on each other.  Every job is cut and summed exactly as hd_gemm_tn_limbs_f32 cuts and sums it (bit-identical results);
 * the resident workgroups walk the jobs one after the other without meeting, so four launch ramps, four drains and four
 * reduction launches go away.  workspace of hd_gemm_tn_limbs_multi_workspace_floats(R, jobs, n) floats (16-byte aligned;
 * -1: bad arguments). */
typedef struct hd_tn_job {
    const void* a_img;
    const void* b_img;
    void* c;
    void* colsum_a;     /* P floats or NULL */
    int64_t P, Q, ldc;
} hd_tn_job;
int64_t hd_gemm_tn_limbs_multi_workspace_floats(int64_t R, const hd_tn_job* jobs, int n_jobs);
int hd_gemm_tn_limbs_multi_f32(int64_t R, const hd_tn_job* jobs, int n_jobs, void* workspace, void* stream);
/* out[cols, rows] = in[rows, cols]^T */
int hd_transpose_f32(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                     void* stream);
/* Several transposes in one launch (at most 64 jobs): every W^T of a backward pass (autograd of F.linear, ddpm.py:17-27). */
typedef struct hd_transpose_job {
    const void* in;
    void* out;
    int64_t rows, cols, ld_in, ld_out;
} hd_transpose_job;
int hd_transpose_multi_f32(const hd_transpose_job* jobs, int n_jobs, void* stream);
/* bias gradient: out[n] = sum_m x[m, n]; workspace of hd_colsum_workspace_floats(M,N) floats */
int64_t hd_colsum_workspace_floats(int64_t M, int64_t N);
int hd_colsum_f32(const void* x, void* out, int64_t M, int64_t N, int64_t ld, void* workspace, void* stream);
/* F.mse_loss(a, b) (models/direction_diffusion.py:362; ddpm.py:83-85 `.square().mean()`):
 * diff = a - b (kept for the backward pass), loss[0] = mean(diff^2) (fp64 accumulation);
 * workspace of hd_mse_workspace_doubles() doubles.  Backward: grad_out = diff * 2/count *
 * grad_loss[0] (grad_loss is a DEVICE scalar: no host read-back inside loss.backward()). */
int64_t hd_mse_workspace_doubles(void);
int hd_mse_forward_f32(const void* a, const void* b, void* diff, int64_t count, void* workspace, void* loss,
                       void* stream);
int hd_mse_backward_f32(const void* diff, const void* grad_loss, void* grad_out, int64_t count, void* stream);
/* GCNConv tail of the gcndec denoiser (models/direction_diffusion.py:28-37; torch_geometric
 * GCNConv adds its bias after the propagation, F.gelu follows): out = act ? gelu(x + bias) :
 * x + bias; z (optional) receives x + bias for the backward pass.  hd_gelu_bwd: dz = g * gelu'(z). */
int hd_bias_gelu_f32(const void* x, const void* bias, void* out, void* z, int64_t rows, int64_t dim, int act,
                     void* stream);
int hd_gelu_bwd_f32(const void* g, const void* z, void* dz, int64_t count, void* stream);
/* double-precision x @ W^T (+bias) for fp64 HypLinear */
int hd_gemm_f64(const void* A, const void* W, const void* bias, void* C, int64_t M, int64_t N, int64_t K,
                int64_t lda, int64_t ldw, int64_t ldc, void* stream);

/* ---- direction diffusion, models/direction_diffusion.py ---------------------------- */
/* dtype codes: 0 = f32, 1 = f64 */
/* q_sample :198-213.  noise is modified IN PLACE when restrict != 0 (tran_direction :154-163
 * with signs = sign(direction) as an fp32 (n_centres, dim) table).  t int64 (rows,),
 * sqrt_ac / sqrt_1mac = get_alphas2 tables (:173-184, fp64, length T).  out is fp64. */
int hd_q_sample(const void* x0, int x_dtype, void* noise, int noise_dtype, const int64_t* t,
                const int32_t* labels, const float* signs, int64_t n_centres, const double* sqrt_ac,
                const double* sqrt_1mac, int64_t T, double* out, int64_t rows, int64_t dim, int restrict_,
                double sqrt_c, double horizon, double prior, void* stream);
/* The training forward between the moments and the denoiser in one pass (diffusion.forward :335-363 with graphset):
 * normalization (:145-149) with the given {sum, sumsq, count}, q_sample with restrict (:198-213), and the mlp's
 * cat(x_t.float(), t_col) (:57-58; t_col = t / T as the caller's fp32 tensor, or null: (float)t / (float)T is then formed
 * in the kernel, which is what torch's int64 / int true division gives).  x fp64 (rows, dim); outputs: x_start as fp32 (the loss target, :362) and the packed
 * fp32 input rows h (row pitch ldh >= dim + 1, zero padded).  Bit-identical to hd_standardize + hd_q_sample +
 * hd_mlp_pack_input.  noise is sign-flipped in place only when write_noise is set (the caller owns it). */
int hd_q_sample_pack_train(const void* x, const void* moments3, void* noise, int noise_dtype, int write_noise,
                           const int64_t* t, const float* t_col, const int32_t* labels, const float* signs, int64_t n_centres,
                           const double* sqrt_ac, const double* sqrt_1mac, int64_t T, void* x_start, void* h, int64_t ldh,
                           int64_t rows, int64_t dim, double sqrt_c, double horizon, double prior, void* stream);
/* tran_direction :154-163, in place: noise[i] *= signs[labels[i]] */
int hd_tran_direction(void* noise, int noise_dtype, const int32_t* labels, const float* signs, int64_t n_centres,
                      int64_t rows, int64_t dim, void* stream);
/* everything of p_sample :272-291 after the denoiser call: predict_noise_from_start :230-236 on
 * the unclamped x0, clamp_(-1,1) in place :266-267, q_sample(x0, t2, eps, restrict=True) :289.
 * The four table values are looked up on the host for the scalar steps t and t2. */
int hd_p_sample_update(const void* x_t, int x_dtype, float* x0, const int32_t* labels, const float* signs,
                       int64_t n_centres, double sqrt_recip_t, double sqrt_recipm1_t, double sqrt_ac_t2,
                       double sqrt_1mac_t2, int64_t t, int64_t t2, double* out, int64_t rows, int64_t dim,
                       double sqrt_c, double horizon, double prior, void* stream);
/* torch.cat((x.float(), t[:,None]), 1).float() :57-58 into a zero-padded (rows, ld_out) fp32
 * buffer.  t_kind: 0 = int64 per row, 1 = f32 per row, 2 = one scalar (t_scalar). */
int hd_mlp_pack_input(const void* x, int x_dtype, const void* t, int t_kind, double t_scalar, float* out,
                      int64_t rows, int64_t dim, int64_t ld_out, void* stream);
/* normalization / renormal :145-153: out3 = {sum, sum of squares, count} over all elements
 * (fp64 accumulation, fixed reduction tree -> bit-reproducible); workspace holds
 * hd_moments_workspace_doubles() doubles. */
int64_t hd_moments_workspace_doubles(void);
int hd_moments(const void* x, int dtype, int64_t count, double* workspace, double* out3, void* stream);
/* hd_moments without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE:
 * it exists for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  The partial sums go to a 16 KiB array that lives
 * in the library's code object on every GPU it runs on (until the process ends); its device address is
 * resolved once per GPU on the first call there (make that call before a capture begins).  Same
 * kernels and reduction tree as hd_moments: out3 is bit-identical.  count <= 2^32 elements, else
 * status 1.  Calls on ONE GPU share that GPU's array: calls that overlap there (two streams, two
 * threads) race -- serialise them.  Calls no hipMalloc / hipFree or any allocating routine. */
int hd_moments_noalloc(const void* x, int dtype, int64_t count, double* out3, void* stream);
/* inverse == 0: (x - mean) / std ; inverse != 0: x * std + mean ; unbiased std, from moments3 */
int hd_standardize(const void* x, void* out, int dtype, int64_t count, const double* moments3, int inverse,
                   void* stream);

/* ---- graph-level latent DDPM, ddpm.py / ddpm_utils.py (config 3), fp32 ---------------- */
/* Model.forward noising ddpm.py:95-104 (idx int64 per row, alpha_bars fp32 (T,)); restrict != 0
 * selects the geometric-constraint form :103-104 with getpri :63-69 (deta, horizon 2000). */
int hd_ddpm_noise(const float* x, const float* eps, const int64_t* idx, const float* alpha_bars, int64_t T,
                  float* out, int64_t rows, int64_t dim, int restrict_, double deta, double sqrt_c,
                  double horizon, void* stream);
/* DDPMSampler._one_diffusion_step ddpm_utils.py:36-45 (pred_noise, restrict False):
 * out = c1*(x - c2*eps) + c3*noise; noise NULL = zeros (idx 0) */
int hd_ddpm_step(const float* x, const float* eps, const float* noise, double c1, double c2, double c3, float* out,
                 int64_t count, void* stream);
/* Backbone.forward ddpm.py:39-41: h[r] += table[idx[r]] in place */
int hd_embedding_add(float* h, const float* table, const int64_t* idx, int64_t T, int64_t rows, int64_t dim,
                     void* stream);
/* Its table gradient (autograd of ddpm.py:39-41): out[t] (T, dim) = sum of the rows r of g (rows, dim) with idx[r] == t,
 * in ascending r.  sorted_idx / sorted_rows = the values and positions of a stable ascending sort of idx (indices outside
 * [0, T) contribute nothing).  No read-back: usable under HIP graph capture.  The same segment sum adds up the endpoint
 * gradients per node in the backward pass of LPModel.decode (models/base_models.py:152-159). */
int hd_embedding_grad_f32(const float* g, const int64_t* sorted_idx, const int64_t* sorted_rows, int64_t rows, int64_t dim,
                          int64_t T, float* out, void* stream);

/* ---- latent -> graph decoding, diff.py:327-378 --------------------------------------------------
 * The tensor part of generate (:334-349) and decode_adj (:355-378) for a set of graphs: h (n_graphs, n, n) fp32
 * latents in (0, 1) -> logit -> per-graph min-max normalisation -> threshold 0.5 -> symmetric 0/1 adjacency
 * adj (n_graphs, n+1, n+1) fp32; any[g] = 0 where nothing exceeds the threshold (the caller then keeps the input
 * graph, :343-344).  The networkx hand-off (get_graph :380-391) stays on the host. */
int hd_generate_adj_f32(const float* h, float* adj, int32_t* any, int64_t n_graphs, int64_t n, void* stream);

/* ---- hyperbolic k-means (labels / centres u0 between encode and diffusion), fp64 -------------
 * hyperbolic_learning/hyperbolic_kmeans/hkmeans.py + util_hk.py.  Hyperboloid rows carry the
 * time coordinate LAST, as util_hk.py does. */
/* poincare_pts_to_hyperboloid util_hk.py:28-61: y (rows, dim) -> h (rows, dim+1) */
int hd_hkm_to_hyperboloid(const double* y, double* h, int64_t rows, int64_t dim, void* stream);
/* one round of compute_mean hkmeans.py:56-70 for all K centres at once: theta (K, dim+1) is
 * updated in place from the members (labels int32 (rows,)) of each cluster; a cluster without
 * members keeps its centre.  workspace: hd_hkm_workspace_doubles(rows, dim, K) doubles. */
int64_t hd_hkm_workspace_doubles(int64_t rows, int64_t dim, int64_t K);
int hd_hkm_mean_step(const double* h, const int32_t* labels, double* theta, int64_t rows, int64_t dim, int64_t K,
                     double alpha, double* workspace, void* stream);
/* hyperboloid_pt_to_poincare util_hk.py:118-133 (+ proj :112-115): theta (K, dim+1) -> (K, dim) */
int hd_hkm_to_poincare(const double* theta, double* centres, int64_t K, int64_t dim, void* stream);
/* assignment step hkmeans.py:206-214: labels[i] = argmin_k poincare_dist(x_i, centre_k)
 * (util_hk.py:14-16); dist2 (optional) = that distance squared (the inertia terms) */
int hd_hkm_assign(const double* x, const double* centres, int32_t* labels, double* dist2, int64_t rows,
                  int64_t dim, int64_t K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPDIFF_H */
