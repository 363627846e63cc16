// COO (int64 indices, the layout of utils/data_utils.py:125-133) -> CSR with int32
// row pointers / columns, entries kept in COO order inside each row.
//
// The reference's adjacency is row-sorted (scipy .tocoo() of a CSR matrix), so the
// entry permutation is the identity and the row pointer is the position of each row's
// first entry: thread i compares row[i-1] with row[i] and fills the pointers of the
// rows that start there.  No atomics, bit-reproducible.  flags[0] receives
//   bit0: the COO is NOT row-sorted (the host then sorts stably and calls again)
//   bit1: an index is out of range (the host raises; nothing downstream may gather)
#include "hd_common.cuh"
#include "../../include/hypdiff.h"

using namespace hd;

namespace {

__global__ void coo_to_csr_sorted(const int64_t* __restrict__ row, const int64_t* __restrict__ colin,
                                  int64_t nnz, int64_t n_rows, int64_t n_cols, int32_t* __restrict__ rowptr,
                                  int32_t* __restrict__ col, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nnz == 0) {
        for (int64_t r = i; r <= n_rows; r += (int64_t)gridDim.x * blockDim.x) rowptr[r] = 0;
        return;
    }
    if (i >= nnz) return;
    const int64_t r = row[i], c = colin[i];
    int bad = 0;
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) bad |= 2;
    const int64_t prev = i == 0 ? -1 : row[i - 1];
    if (prev > r) bad |= 1;
    if (bad) atomicOr(flags, bad);
    col[i] = (int32_t)c;
    if (!(bad & 2) && prev >= -1 && prev < n_rows) {
        for (int64_t q = prev + 1; q <= r; ++q) rowptr[q] = (int32_t)i;
        if (i == nnz - 1)
            for (int64_t q = r + 1; q <= n_rows; ++q) rowptr[q] = (int32_t)nnz;
    }
}

}  // namespace

extern "C" int hd_csr_from_coo(const int64_t* row, const int64_t* col_in, int64_t nnz, int64_t n_rows,
                               int64_t n_cols, int32_t* rowptr, int32_t* col, int32_t* flags, void* st) {
    HD_REQUIRE(nnz >= 0 && n_rows >= 0 && n_cols >= 0, "hd_csr_from_coo: negative size");
    HD_REQUIRE(nnz < (1LL << 31) && n_rows < (1LL << 31) - 1, "hd_csr_from_coo: int32 index range exceeded");
    HD_REQUIRE(rowptr && flags && (nnz == 0 || (row && col_in && col)), "hd_csr_from_coo: null pointer");
    hipError_t e = hipMemsetAsync(flags, 0, sizeof(int32_t), (hipStream_t)st);
    if (e != hipSuccess) { set_error("hd_csr_from_coo: memset: %s", hipGetErrorString(e)); return HD_EHIP; }
    const int64_t work = nnz > 0 ? nnz : n_rows + 1;
    const int64_t blocks = (work + 255) / 256;
    coo_to_csr_sorted<<<(unsigned)blocks, 256, 0, (hipStream_t)st>>>(
        row, col_in, nnz, n_rows, n_cols, rowptr, col, flags);
    return check_launch("hd_csr_from_coo");
}

// ---- adjacency preprocessing on the device: utils/data_utils.py:103-133 ------------------------
// process(): adj <- normalize(adj + I) = D^-1 (A + I) in fp64 (r_inv = 1 / rowsum, inf -> 0), .tocoo()
// (row-major, columns DESCENDING inside a row: see adj_fill), indices int64, values cast to fp32.  Input: CSR of adj_train with
// ascending columns per row (scipy's canonical form), optional fp64 values (null = all ones).
// Two launches around the caller's prefix sum: adj_count gives every output row's length (the
// diagonal adds an entry only where A has none), adj_fill merges the diagonal into the row and
// writes indices and values.  One lane per row; preprocessing runs once per graph.
namespace {

__global__ void adj_count(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, int64_t n,
                          int64_t* __restrict__ counts, int32_t* __restrict__ flags) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    int bad = 0;
    if (e < b) bad |= 2;
    bool diag = false;
    int64_t prev = -1;
    for (int64_t k = b; k < e; ++k) {
        const int64_t c = col[k];
        if (c < 0 || c >= n) bad |= 2;
        if (c <= prev) bad |= 1;                // not strictly ascending (unsorted or duplicate entries)
        prev = c;
        diag |= c == r;
    }
    if (bad) atomicOr(flags, bad);
    counts[r] = (e >= b ? e - b : 0) + (diag ? 0 : 1);
}

__global__ void adj_fill(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                         const double* __restrict__ val, int64_t n, const int64_t* __restrict__ out_ptr,
                         int normalize, int64_t* __restrict__ orow, int64_t* __restrict__ ocol,
                         float* __restrict__ oval) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    double rowsum = 1.0;                         // the identity's entry; scipy sums a row in storage order
    bool diag = false;
    for (int64_t k = b; k < e; ++k) {
        diag |= col[k] == r;
    }
    // (adj + I).sum(1): the merged row in column order
    {
        double s = 0.0;
        bool placed = false;
        for (int64_t k = b; k < e; ++k) {
            const int64_t c = col[k];
            const double a = val ? val[k] : 1.0;
            if (!placed && c >= r) {
                placed = true;
                if (c == r) { s += a + 1.0; continue; }
                s += 1.0;
            }
            s += a;
        }
        if (!placed) s += 1.0;
        rowsum = s;
    }
    double rinv = 1.0 / rowsum;                  // np.power(rowsum, -1)
    if (isinf(rinv)) rinv = 0.0;                 // r_inv[np.isinf(r_inv)] = 0.
    // r_mat_inv.dot(mx) is scipy's sparse product: it emits every row in the reverse of the order it
    // touched the columns, i.e. the ascending merged row comes out DESCENDING -- the entry order of the
    // reference's COO, and with it the summation order of torch.spmm
    int64_t o = out_ptr[r + 1] - 1;
    bool placed = false;
    for (int64_t k = b; k < e; ++k) {
        const int64_t c = col[k];
        double a = val ? val[k] : 1.0;
        if (!placed && c >= r) {
            placed = true;
            if (c == r) a += 1.0;
            else {
                orow[o] = r; ocol[o] = r; oval[o] = (float)(normalize ? rinv * 1.0 : 1.0); --o;
            }
        }
        orow[o] = r; ocol[o] = c; oval[o] = (float)(normalize ? rinv * a : a); --o;
    }
    if (!placed) { orow[o] = r; ocol[o] = r; oval[o] = (float)(normalize ? rinv * 1.0 : 1.0); }
    (void)diag;
}

// features <- normalize(features) (:106-107,115-122): every row divided by its fp64 row sum
// (1/0 -> 0), result cast to fp32.  One wavefront per row.
template <typename TX>
__global__ __launch_bounds__(256) void feat_normalize(const TX* __restrict__ x, int64_t n, int64_t d,
                                                      float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const TX* p = x + row * d;
    double s = 0.0;
    for (int64_t c = lane; c < d; c += 64) s += (double)p[c];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    double rinv = 1.0 / s;
    if (isinf(rinv)) rinv = 0.0;
    for (int64_t c = lane; c < d; c += 64) out[row * d + c] = (float)(rinv * (double)p[c]);
}

}  // namespace

extern "C" int hd_adj_count(const int64_t* rowptr, const int64_t* col, int64_t n, int64_t* counts, int32_t* flags,
                            void* st) {
    HD_REQUIRE(n >= 0, "hd_adj_count: negative size");
    HD_REQUIRE(flags && (n == 0 || (rowptr && counts)), "hd_adj_count: null pointer");
    hipError_t e = hipMemsetAsync(flags, 0, sizeof(int32_t), (hipStream_t)st);
    if (e != hipSuccess) { set_error("hd_adj_count: memset: %s", hipGetErrorString(e)); return HD_EHIP; }
    if (n == 0) return HD_OK;
    adj_count<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)st>>>(rowptr, col, n, counts, flags);
    return check_launch("hd_adj_count");
}

extern "C" int hd_adj_fill(const int64_t* rowptr, const int64_t* col, const double* val, int64_t n,
                           const int64_t* out_ptr, int normalize, int64_t* out_row, int64_t* out_col, float* out_val,
                           void* st) {
    HD_REQUIRE(n >= 0, "hd_adj_fill: negative size");
    if (n == 0) return HD_OK;
    HD_REQUIRE(rowptr && out_ptr && out_row && out_col && out_val, "hd_adj_fill: null pointer");
    adj_fill<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)st>>>(rowptr, col, val, n, out_ptr, normalize, out_row,
                                                                       out_col, out_val);
    return check_launch("hd_adj_fill");
}

extern "C" int hd_feat_normalize(const void* x, int x_dtype, int64_t rows, int64_t dim, float* out, void* st) {
    HD_REQUIRE(rows >= 0 && dim >= 1, "hd_feat_normalize: bad shape");
    if (rows == 0) return HD_OK;
    HD_REQUIRE(x && out, "hd_feat_normalize: null pointer");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (x_dtype) feat_normalize<double><<<grid, 256, 0, (hipStream_t)st>>>((const double*)x, rows, dim, out);
    else feat_normalize<float><<<grid, 256, 0, (hipStream_t)st>>>((const float*)x, rows, dim, out);
    return check_launch("hd_feat_normalize");
}
