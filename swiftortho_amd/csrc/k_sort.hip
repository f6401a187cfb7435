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

// ---- segmented variant: hits leave the lookup kernel ordered by query, so only the (subject, diagonal)
// bits need sorting inside each query's segment (fewer radix passes than a device-wide sort that
// also carries the query bits).
__global__ __launch_bounds__(256) void k_query_segments(const u32* __restrict__ hoff, size_t T, const u32* __restrict__ qoff, u32 nq, int AS,
                                                        u32 H, u32* __restrict__ seg /*nq + 1*/) {
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (q > nq) return;
    const size_t t = ((size_t)qoff[q] + q) * (size_t)AS;  // first seed slot of query q (position-major ordinals)
    seg[q] = t < T ? hoff[t] : H;
}

void launch_query_segments(const u32* hoff, size_t T, const u32* qoff, u32 nq, int AS, u32 H, u32* seg, hipStream_t st) {
    hipLaunchKernelGGL(k_query_segments, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, hoff, T, qoff, nq, AS, H, seg);
}

size_t sort_keys_u64_seg_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)hipcub::DeviceSegmentedRadixSort::SortKeys((void*)nullptr, bytes, (const u64*)nullptr, (u64*)nullptr, (int)n, (int)nseg,
                                                      (const u32*)nullptr, (const u32*)nullptr, begin_bit, end_bit, (hipStream_t)0);
    return bytes;
}

void sort_keys_u64_seg(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit,
                       int end_bit, hipStream_t st) {
    if (n == 0) return;
    HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortKeys(temp, temp_bytes, in, out, (int)n, (int)nseg, seg, seg + 1, begin_bit, end_bit, st));
}
