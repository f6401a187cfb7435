// X-drop recurrence of k_ungap, 8 elements per step, all CUs busy at 8 waves per SIMD: cycles per element per SIMD for
//   A  the kernel's form:   s += t; drop = mp - s >= C; mp = max(mp, s); s = drop ? PIN : s          (add, sub, cmp, max, cndmask)
//   B  deferred drop test:  s += t; mp = max(mp, s); d = mp - s; acc = max3(acc, d_k, d_k+1)         (add, max, sub, half a max3)
//   C  A without the pin (lower bound of the add / max / compare part)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/xdrop.hip -o tools/ubench/xdrop && tools/ubench/xdrop
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define PIN (-(1 << 30))
#define C ((31 << 4) - 7)

template <int V>
__global__ __launch_bounds__(256) void k(int iters, const int* __restrict__ tab, int* out) {
    int t[8];
    for (int i = 0; i < 8; ++i) t[i] = tab[(threadIdx.x + i * 7) & 63];
    int s = 16, mp = 15, acc = 0, res = 0;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                s += t[k2];
                const bool drop = mp - s >= C;
                mp = max(mp, s);
                s = drop ? PIN : s;
            }
        } else if (V == 1) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2 += 2) {
                s += t[k2];
                mp = max(mp, s);
                const int d0 = mp - s;
                s += t[k2 + 1];
                mp = max(mp, s);
                const int d1 = mp - s;
                acc = max(acc, max(d0, d1));
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                s += t[k2];
                acc |= (mp - s >= C) ? 1 : 0;
                mp = max(mp, s);
            }
        }
        // chunk epilogue as in the kernel (keeps the chain honest across iterations)
        res += mp & 15;
        mp |= 15;
        s += 8;
        asm volatile("" : "+v"(s), "+v"(mp), "+v"(acc));
    }
    if (res + acc + s == 0x12345) out[0] = res;
}

template <int V>
int run(const char* name, const int* tab, int* o) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), 0, 0, iters, tab, o);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double elems_per_simd = (double)blocks * 4 / 1024 * iters * 8;   // wave-elements per SIMD
    printf("%-40s %7.3f ms = %5.2f cycles per element (wave) per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / elems_per_simd);
    return 0;
}

int main() {
    int h[64];
    for (int i = 0; i < 64; ++i) h[i] = ((i * 37 % 11) - 5) * 16 - 1;
    int *tab, *o;
    CK(hipMalloc(&tab, sizeof h));
    CK(hipMalloc(&o, 64));
    CK(hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice));
    run<0>("A add/sub/cmp/max/cndmask", tab, o);
    run<1>("B add/max/sub + max3 per pair", tab, o);
    run<2>("C add/sub/cmp/max + flag", tab, o);
    return 0;
}
