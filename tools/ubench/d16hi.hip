// d16hi.hip -- does ds_read_u16_d16_hi keep the low half of its destination on this part?  (LLVM does not select the d16 loads for
// gfx950: with SRAM ECC the register file zeroes the half a 16-bit load does not write.)  Prints the register after a d16 + d16_hi pair.
//   hipcc --offload-arch=gfx950 -O3 -o d16hi d16hi.hip && ./d16hi
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    __shared__ unsigned short tab[256];
    tab[threadIdx.x] = (unsigned short)(0x1100 + threadIdx.x);
    tab[threadIdx.x + 64] = (unsigned short)(0x2200 + threadIdx.x);
    __syncthreads();
    unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)tab + 2u * threadIdx.x, a1 = a0 + 128u;
    unsigned c = 0xdeadbeefu;
    asm volatile("ds_read_u16_d16 %0, %1\n\tds_read_u16_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(a0), "v"(a1));
    out[threadIdx.x] = c;
}
int main() {
    unsigned* d;
    unsigned h[64];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane 0: %08x  lane 5: %08x  (both halves kept: 22001100 / 22051105)\n", h[0], h[5]);
    return 0;
}
