// Micro-benchmarks behind DESIGN.md's statements about the extension / aligner kernels (round 3):
//   A. rate of wave-instructions that GATHER 8 / 16 bytes per lane from an L2-resident array when every lane reads a
//      different cache line (the shape of k_ungap's subject-window loads), against the coalesced shape;
//   B. rate of ds_read_u16 when the 32 lanes of an access group spread over b banks (k_align / k_ungap score-table lookups).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/gather.hip -o /tmp/gather && /tmp/gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BYTES>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* __restrict__ base, unsigned mask, unsigned lane_stride, int iters, unsigned long long* out) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned off = (wave * 7919u * 64u + lane * lane_stride) & mask;
    unsigned long long acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (BYTES == 8) {
            unsigned long long w;
            __builtin_memcpy(&w, base + off, 8);
            acc += w;
        } else {
            unsigned long long w[2];
            __builtin_memcpy(w, base + off, 16);
            acc += w[0] ^ w[1];
        }
        off = (off + 8u * (BYTES / 8) + ((unsigned)acc & 0u)) & mask;   // next window of the same lane: +8 (or +16) bytes, as the X-drop walk does
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_lds(int banks, int iters, unsigned* out) {
    __shared__ unsigned short tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = (unsigned short)i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    // lane l reads dword (l % banks) + 32 * (l / banks) * k: `banks` distinct banks per 32-lane group, different addresses inside a bank
    unsigned idx = ((lane & 31) % banks) * 2 + ((lane & 31) / banks) * 64 * 2;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += tab[idx & 8191];
        idx += 64 * 2 * 8 + (acc & 0u);
    }
    if (acc == 0x1234567u) out[0] = acc;
}

int main() {
    const size_t N = 3u << 20;   // 3 MiB: L2-resident per XCD
    unsigned char* d;
    unsigned long long* o;
    CK(hipMalloc(&d, N + 64));
    CK(hipMalloc(&o, 64));
    CK(hipMemset(d, 1, N + 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 2000;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    const unsigned mask = (2u << 20) - 1;
    struct { const char* name; int bytes; unsigned stride; } cases[] = {{"8 B/lane, 64 lines per instruction", 8, 4099 * 8}, {"16 B/lane, 64 lines per instruction", 16, 4099 * 8},
                                                                         {"8 B/lane, coalesced (512 B per instruction)", 8, 8}, {"16 B/lane, coalesced (1 KiB per instruction)", 16, 16},
                                                                         {"8 B/lane, 300 B apart (neighbouring subjects)", 8, 300}, {"16 B/lane, 300 B apart", 16, 300}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (c.bytes == 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, d, mask, c.stride, iters, o);
            else hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, d, mask, c.stride, iters, o);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * 4 * iters;
        printf("gather %-48s %8.3f ms  %7.2f G wave-instr/s chip  = %6.1f cycles per instr per CU at 2.4 GHz\n", c.name, ms, instr / ms / 1e6,
               ms * 1e-3 * 2.4e9 / (instr / 256));
    }
    for (int banks : {32, 16, 8, 4, 2, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, banks, 4000, (unsigned*)o);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * 4 * 4000;
        printf("ds_read_u16, %2d banks per 32 lanes (%2d-way conflict): %8.3f ms = %6.1f LDS cycles per wave-instr per CU\n", banks, 32 / banks, ms,
               ms * 1e-3 * 2.4e9 / (instr / 256));
    }
    return 0;
}
