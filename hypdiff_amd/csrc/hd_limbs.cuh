// bf16x3 limb helpers shared by the row kernels that feed or use the matrix cores directly
// (the limb GEMM files keep their own file-local copies next to their format tables).
#pragma once
#include <hip/hip_runtime.h>

namespace hd {

typedef float lf32x2 __attribute__((ext_vector_type(2)));
typedef float lf32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lbf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned limb_pk_bf16(float x, float y) {      // v_cvt_pk_bf16_f32 (RNE)
    const lf32x2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, lbf16x2));
}
__device__ __forceinline__ float limb_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float limb_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// eight fp32 -> three limbs of eight bf16 (16 bytes each): v = h + m + l exactly (24 significand bits)
__device__ __forceinline__ void limb_split8(const float v[8], uint4& h, uint4& m, uint4& l) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hh[e] = limb_pk_bf16(v[2 * e], v[2 * e + 1]);
        const float r0 = v[2 * e] - limb_lo(hh[e]), r1 = v[2 * e + 1] - limb_hi(hh[e]);
        mm[e] = limb_pk_bf16(r0, r1);
        ll[e] = limb_pk_bf16(r0 - limb_lo(mm[e]), r1 - limb_hi(mm[e]));
    }
    h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    m = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// 16 bytes per lane, global -> LDS without passing through registers; `l` is the wavefront's base (the
// hardware adds 16 * lane)
__device__ __forceinline__ void limb_glds16(const uint4* g, uint4* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

}  // namespace hd
