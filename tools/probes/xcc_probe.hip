// prints which XCD (HW_REG_XCC_ID) each of 64 workgroups of a 512-workgroup grid ran on, next to blockIdx % 8
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    }
}
int main() {
    unsigned* d;
    hipMalloc(&d, 512 * 8);
    probe<<<512, 256>>>(d);
    unsigned h[1024];
    hipMemcpy(h, d, 512 * 8, hipMemcpyDeviceToHost);
    int hist[16] = {0}, agree = 0;
    for (int b = 0; b < 512; ++b) { hist[h[2 * b] & 15]++; agree += ((h[2 * b] & 7) == (h[0] + b) % 8); }
    for (int b = 0; b < 24; ++b) printf("block %d: xcc_id[3:0]=%u raw=0x%x\n", b, h[2 * b], h[2 * b + 1]);
    printf("hist:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\nround-robin agreement %d / 512\n", agree);
    return 0;
}
