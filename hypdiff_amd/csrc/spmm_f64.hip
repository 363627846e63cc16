// fp64 instantiation of the HypAgg SpMM (hd_spmm.cuh)
#include "hd_spmm.cuh"
HD_SPMM(double, f64)
