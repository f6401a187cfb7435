// LDS read cost by width and (mis)alignment, addresses random per lane inside a 1 KB window (the query slot of k_ungap1) or lane-private
// (its score table): cycles per wave-instruction per CU with every SIMD holding 4 waves that do nothing else (round 5).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsua.hip -o tools/ubench/ldsua && tools/ubench/ldsua
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) const unsigned char lds_u8;

template <int W /*bytes*/>
__global__ __launch_bounds__(1024, 1) void k(int iters, int align, int mode, unsigned* out, unsigned long long* clk) {
    __shared__ __align__(16) unsigned char buf[64 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) reinterpret_cast<unsigned*>(buf)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    const unsigned wavebase = (threadIdx.x >> 6) * 2048u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            h = h * 1664525u + 1013904223u;
            unsigned a;
            if (mode == 0) a = wavebase + (((h >> 8) % 1000u) & ~(unsigned)(align - 1));          // random inside the wave's 1 KB slot, aligned to `align`
            else a = ((h >> 8) % 400u) * 128u + (threadIdx.x & 31) * 4u;                          // lane-private table entry
            const lds_u8* p = (lds_u8*)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)buf + a);
            if (W == 16) { uint4 v; __builtin_memcpy(&v, (const void*)p, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
            else if (W == 8) { uint2 v; __builtin_memcpy(&v, (const void*)p, 8); acc += v.x ^ v.y; }
            else if (W == 4) { unsigned v; __builtin_memcpy(&v, (const void*)p, 4); acc += v; }
            else { unsigned short v; __builtin_memcpy(&v, (const void*)p, 2); acc += v; }
        }
    }
    if (acc == 0x12345) out[0] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0, clk[1] = wall_clock64() - w0;
}

template <int W>
int run(const char* name, int align, int mode, unsigned* o, unsigned long long* dclk) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256, iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(1024), 0, 0, iters, align, mode, o, dclk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2] = {0, 0};
    CK(hipMemcpy(hc, dclk, sizeof hc, hipMemcpyDeviceToHost));
    const double mhz = hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0;
    const double per_cu = 16.0 * iters * 8;   // wave-instructions per CU
    printf("%-40s align %2d: %7.3f ms = %6.2f cycles per wave-instruction per CU (%.0f MHz)\n", name, align, ms, ms * 1e-3 * mhz * 1e6 / per_cu, mhz);
    return 0;
}

int main() {
    unsigned* o;
    unsigned long long* dclk;
    CK(hipMalloc(&o, 64));
    CK(hipMalloc(&dclk, 64));
    for (int al : {16, 8, 4, 2, 1}) run<16>("ds_read_b128, random in 1 KB slot", al, 0, o, dclk);
    for (int al : {8, 4, 2, 1}) run<8>("ds_read_b64, random in 1 KB slot", al, 0, o, dclk);
    for (int al : {4, 2, 1}) run<4>("ds_read_b32, random in 1 KB slot", al, 0, o, dclk);
    run<2>("ds_read_u16, lane-private table", 2, 1, o, dclk);
    run<4>("ds_read_b32, lane-private table", 4, 1, o, dclk);
    return 0;
}
