// k_sort.hip -- device radix sorts used for diagonal binning and candidate ordering.
// The sort itself is the rocPRIM library primitive (a plain library sort, like a plain
// library GEMM); everything around it (key construction, grouping, extension) is
// hand-written in the other k_*.hip files.
#include "common.h"
#include "kernels.h"
#include <hipcub/hipcub.hpp>

size_t sort_keys_u64_temp_bytes(size_t n, int bits) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys((void*)nullptr, bytes, (const u64*)nullptr, (u64*)nullptr, (int)n, 0, bits, (hipStream_t)0);
    return bytes;
}

// stable LSD radix sort on key bits [begin_bit, end_bit)
void sort_keys_u64(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, int begin_bit, int end_bit, hipStream_t st) {
    if (n == 0) return;
    HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, in, out, (int)n, begin_bit, end_bit, st));
}

size_t sort_pairs_u64_u32_temp_bytes(size_t n, int bits) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs((void*)nullptr, bytes, (const u64*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32*)nullptr,
                                       (int)n, 0, bits, (hipStream_t)0);
    return bytes;
}

void sort_pairs_u64_u32(void* temp, size_t temp_bytes, const u64* kin, u64* kout, const u32* vin, u32* vout, size_t n, int bits,
                        hipStream_t st) {
    if (n == 0) return;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, kin, kout, vin, vout, (int)n, 0, bits, st));
}

__global__ __launch_bounds__(256) void k_stride_gather(const u32* __restrict__ src, u32 stride, u32 n, u32* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[(size_t)i * stride];
}
void launch_stride_gather(const u32* src, u32 stride, u32 n, u32* dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_stride_gather, dim3((n + 255) / 256), dim3(256), 0, st, src, stride, n, dst);
}
