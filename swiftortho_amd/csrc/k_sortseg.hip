// k_sortseg.hip -- the SEGMENTED library sorts (one segment per query): hit keys of the sorted path, candidate order of the bucketed
// passes.  A translation unit of its own: rocPRIM emits every kernel for each of its 13 target architectures, so a unit's code
// object is large and costs tens of milliseconds to load at its first launch -- the index build only needs the device-wide pair
// sort in k_sort.hip, and a search whose passes are all bucketed never loads this one.
#include "common.h"
#include "kernels.h"
#include <hipcub/hipcub.hpp>

// ---- segmented variant: hits leave the lookup kernel ordered by query, so only the (subject, diagonal)
// bits need sorting inside each query's segment (fewer radix passes than a device-wide sort that
// also carries the query bits).
// rocPRIM ships no tuned segmented-sort configuration for gfx950 and falls back to its generic one (6-bit digits,
// 128 x 17 keys per block: four passes over our 24 bits).  Measured on config 2 (247 M keys, 10 k segments;
// generic 5.09 ms; 8-bit digits with 256 x 8 keys 4.16, 256 x 16 keys 3.97-4.06 (default
// here), 256 x 32 6.0, 512 x 16 4.64, 1024 x 8 4.35.
template <int IPT, int BLOCK = 256, int BITS = 8>
using SegCfg = rocprim::segmented_radix_sort_config<BITS, rocprim::kernel_config<BLOCK, IPT>, rocprim::WarpSortConfig<8, 4, 256, 64, 16, 8, 256>, true>;

// Long segments (weight-6 seeds: tens of thousands of keys per query) want the 8-bit / 256 x 16 instance; short ones
// (weight-10 seeds: hundreds per query) are served better by the library default with its warp-sort size classes
// (config 3: 2.7 ms vs 3.7 ms).
// Digit width: the field is bs + bd bits wide (24 on config 2, 27 with 50 k-sequence chunks); 9-bit digits with 512 x 8
// keys cost the same per pass as 8-bit / 256 x 16 (3.96 vs 3.97 ms on config 2) and keep widths up to 27 at three passes.
// (10-bit digits with 1024 x 4 keys: 5.3 ms per three passes -- no better than four 8-bit ones.)
static int seg_variant(size_t n, u32 nseg, int width) {
    if (!(nseg && n / nseg >= 4096)) return 0;
    if (width > 24 && width <= 27) return 6;
    return 2;
}

template <class Cfg>
static hipError_t seg_sort(void* temp, size_t& bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* sb, const u32* se, int b0,
                           int b1, hipStream_t st) {
    return rocprim::segmented_radix_sort_keys<Cfg>(temp, bytes, in, out, (unsigned)n, nseg, sb, se, (unsigned)b0, (unsigned)b1, st, false);
}

static hipError_t seg_sort_dispatch(void* temp, size_t& bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* sb, const u32* se,
                                    int b0, int b1, hipStream_t st) {
    switch (seg_variant(n, nseg, b1 - b0)) {
        // (the instances that lost the sweep -- 256 x 8, 1024 x 8, 1024 x 4 with 10-bit digits -- are no longer compiled: rocPRIM emits
        // every kernel for each of its 13 target architectures, the library was 39 MB and its first sort spent 86 ms loading it)
        case 0: return seg_sort<rocprim::default_config>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
        case 4: return seg_sort<SegCfg<16, 512>>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
        case 6: return seg_sort<SegCfg<8, 512, 9>>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
        default: return seg_sort<SegCfg<16>>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
    }
}

size_t sort_keys_u64_seg_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)seg_sort_dispatch(nullptr, bytes, nullptr, nullptr, n, nseg, nullptr, nullptr, begin_bit, end_bit, (hipStream_t)0);
    return bytes;
}

void sort_keys_u64_seg(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit,
                       int end_bit, hipStream_t st) {
    if (n == 0) return;
    HIP_CHECK(seg_sort_dispatch(temp, temp_bytes, in, out, n, nseg, seg, seg + 1, begin_bit, end_bit, st));
}
void sort_keys_u64_seg2(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg_begin, const u32* seg_end, int begin_bit,
                        int end_bit, hipStream_t st) {
    if (n == 0) return;
    HIP_CHECK(seg_sort_dispatch(temp, temp_bytes, in, out, n, nseg, seg_begin, seg_end, begin_bit, end_bit, st));
}

// ---- keys-only variant for the candidate order of a bucketed pass (the position rides in the low bits of the sort word) ----------
// NOTE end_bit must stay below 64: rocPRIM's comparator for short segments builds its mask as (1 << (begin + bits)) - 1, which for
// begin + bits == 64 shifts by the word size and ends up comparing the bits BELOW begin_bit (ROCm 7.2) -- the caller keeps bit 63 free.
static hipError_t cand_keys_dispatch(void* temp, size_t& bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* sb, const u32* se, int b0,
                                     int b1, hipStream_t st) {
    const size_t avg = nseg ? n / nseg : 0;
    const int cfg = avg >= 3072 ? 2 : avg >= 1024 ? 1 : 0;
    if (cfg == 2) return seg_sort<SegCfg<16, 512>>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
    if (cfg == 1) return seg_sort<SegCfg<16>>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
    return seg_sort<rocprim::default_config>(temp, bytes, in, out, n, nseg, sb, se, b0, b1, st);
}
size_t sort_cand_keys_seg_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)cand_keys_dispatch(nullptr, bytes, nullptr, nullptr, n, nseg, nullptr, nullptr, begin_bit, end_bit, (hipStream_t)0);
    return bytes;
}
void sort_cand_keys_seg(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit, int end_bit,
                        hipStream_t st) {
    if (n == 0) return;
    if (end_bit >= 64) throw SoError("sort_cand_keys_seg: end_bit must be below 64");
    HIP_CHECK(cand_keys_dispatch(temp, temp_bytes, in, out, n, nseg, seg, seg + 1, begin_bit, end_bit, st));
}


// ---- members of every index bucket in DESCENDING entry order (order_chunk, host_index.hip): one segment per occupied bucket (tens of entries:
// the library's warp-sort size classes), once per index build and only for chunks whose passes are dense enough for the bucketed binning ----
size_t sort_keys_u64_seg_desc_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_keys_desc<rocprim::default_config>(nullptr, bytes, (const u64*)nullptr, (u64*)nullptr, (unsigned)n, nseg, (const u32*)nullptr,
                                                                           (const u32*)nullptr, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0, false);
    return bytes;
}
void sort_keys_u64_seg_desc(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit, int end_bit, hipStream_t st) {
    if (n == 0 || nseg == 0) return;
    if (end_bit >= 64) throw SoError("sort_keys_u64_seg_desc: end_bit must be below 64");
    HIP_CHECK(rocprim::segmented_radix_sort_keys_desc<rocprim::default_config>(temp, temp_bytes, in, out, (unsigned)n, nseg, seg, seg + 1, (unsigned)begin_bit,
                                                                               (unsigned)end_bit, st, false));
}
