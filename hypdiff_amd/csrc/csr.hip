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
