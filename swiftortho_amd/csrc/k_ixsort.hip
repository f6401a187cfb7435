// k_ixsort.hip -- the index build's grouping step (fsearch.py:2240-2266: count per bucket, prefix, fill): the (bucket id, entry) pairs
// of a chunk, emitted in position order, grouped by ascending bucket id.  The members of a bucket may come out in any order (every
// consumer derives the visiting order from the entry itself; tests/test_gpu_parity.py check_index compares buckets as sets), so no
// stable sort is needed -- two counting passes over digits of the bucket id do it:
//
//   level 1   the id range [0, NC) is cut into <= 8192 BINS of W = 2^wsh ids.  IXS_G persistent workgroups each take a contiguous slice
//             of the pairs: k_ixs_count builds the slice's bin histogram in LDS and stores it as column g of a bins x IXS_G matrix, an
//             exclusive scan of the matrix in (bin, g) order is the scatter plan, k_ixs_scatter re-reads the slice and writes every pair
//             to its bin's region (an LDS fetch-add on the plan's entry gives the slot).  This pass is the expensive one: 0.56 of the
//             0.85 ms -- a 7000-way scatter keeps ~66 MB of partly written lines open per XCD against 4 MB of L2, so lines leave the
//             cache partly written.
//   level 2   k_ixs_bin, one workgroup per bin (~2000 pairs over 16384 ids at the default -M): LDS counters per id, count, scan,
//             scatter -- the bin's pairs land grouped by id.  Bins of more than 16384 ids (-M above 2^27) are first split by the id's
//             upper digit (<= 32 sub-ranges, through the level-1 input arrays as scratch), then every sub-range is grouped like a bin.
//
// Measured per 50 000-sequence chunk (14.7 M pairs, -M 120000000; profiles/r05_*_c3only_kernel_stats.csv): count 0.04 + plan scan 0.03 + scatter
// 0.56 + bins 0.22 = 0.85 ms, against 0.94 ms of the library radix sort (hipcub::DeviceRadixSort::SortPairs) it replaces.
#include "common.h"
#include "kernels.h"

#define IXS_G 512            // level-1 workgroups (= columns of the count matrix)
#define IXS_T1 1024          // their threads
#define IXS_BINS_MAX 8192
#define IXS_W 16384          // ids per level-2 counting pass (LDS counters)
#define IXS_T2 512

static inline int ixs_wsh(u32 NC) {   // log2 of the bin width: the fewest bits that leave <= IXS_BINS_MAX bins
    int wsh = 0;
    while ((((u64)NC + (1ull << wsh) - 1) >> wsh) > IXS_BINS_MAX) ++wsh;
    return wsh;
}
u32 ixsort_bins(u32 NC) { return (u32)(((u64)NC + (1ull << ixs_wsh(NC)) - 1) >> ixs_wsh(NC)); }
size_t ixsort_plan_elems(u32 NC) { return (size_t)ixsort_bins(NC) * IXS_G + 1; }

// Slice (= plan column) of this workgroup.  Workgroups are dealt round-robin to the 8 XCDs, so XCD x takes the x-th contiguous eighth of the
// slices: inside a bin the sub-ranges of neighbouring slices are neighbours too (~16 bytes of keys each), and a 128-byte output line is
// filled by ONE XCD's L2 instead of receiving a partial write-back from each of the eight (0.62 -> 0.56 ms of scatter per chunk).
__device__ __forceinline__ u32 ixs_col() { return (blockIdx.x & 7u) * (IXS_G / 8) + (blockIdx.x >> 3); }
__device__ __forceinline__ void ixs_slice(u32 E, u32& lo, u32& hi) {   // (multiples of 4 pairs but the end)
    const u64 per = (((u64)E + IXS_G - 1) / IXS_G + 3) & ~3ull;
    lo = (u32)min((u64)E, per * ixs_col()), hi = (u32)min((u64)E, per * (ixs_col() + 1));
}

__global__ __launch_bounds__(IXS_T1) void k_ixs_count(const u32* __restrict__ keys, u32 E, int wsh, u32 nbins, u32* __restrict__ plan /*[nbins][IXS_G] (+1)*/) {
    __shared__ u32 s_cnt[IXS_BINS_MAX];
    for (u32 i = threadIdx.x; i < nbins; i += IXS_T1) s_cnt[i] = 0;
    __syncthreads();
    u32 lo, hi;
    ixs_slice(E, lo, hi);
    for (u32 i = lo + threadIdx.x; i < hi; i += IXS_T1) {
        const u32 k = keys[i];
        if (k != 0xFFFFFFFFu) atomicAdd(&s_cnt[k >> wsh], 1u);   // (~0: the slot of an invalid window, k_index_windows_sparse)
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nbins; i += IXS_T1) plan[(size_t)i * IXS_G + ixs_col()] = s_cnt[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) plan[(size_t)nbins * IXS_G] = 0;   // (the scan leaves E there: the end of the last bin)
}

__global__ __launch_bounds__(IXS_T1) void k_ixs_scatter(const u32* __restrict__ keys, const u64* __restrict__ vals, u32 E, int wsh, u32 nbins,
                                                       const u32* __restrict__ plan /*scanned*/, u32* __restrict__ tk, u64* __restrict__ tv) {
    __shared__ u32 s_cur[IXS_BINS_MAX];
    for (u32 i = threadIdx.x; i < nbins; i += IXS_T1) s_cur[i] = plan[(size_t)i * IXS_G + ixs_col()];
    __syncthreads();
    u32 lo, hi;
    ixs_slice(E, lo, hi);
    for (u32 i = lo + threadIdx.x; i < hi; i += IXS_T1) {
        const u32 k = keys[i];
        if (k == 0xFFFFFFFFu) continue;
        const u32 at = atomicAdd(&s_cur[k >> wsh], 1u);
        tk[at] = k, tv[at] = vals[i];
    }
}

// counting pass of one id range inside a workgroup: pairs [a, b) of (sk, sv), ids in [idbase, idbase + IXS_W), to (dk, dv) at the same
// positions, grouped by id
// (sk / sv carry no __restrict__: the split path of k_ixs_bin hands in the scratch pairs the same workgroup has just written)
__device__ __forceinline__ void ixs_group(const u32* sk, const u64* sv, u32 a, u32 b, u32 idbase, u32* __restrict__ dk,
                                          u64* __restrict__ dv, u32* s_cnt, u32* s_ws) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER = IXS_W / IXS_T2;   // counters per thread in the scan
    for (int i = tid; i < IXS_W; i += IXS_T2) s_cnt[i] = 0;
    __syncthreads();
    for (u32 i = a + (u32)tid; i < b; i += IXS_T2) atomicAdd(&s_cnt[sk[i] - idbase], 1u);
    __syncthreads();
    {   // exclusive scan of the counters (each thread PER consecutive ones), + a
        u32 c[PER], tot = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) c[k] = s_cnt[tid * PER + k], tot += c[k];
        u32 inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
        }
        if (lane == 63) s_ws[w] = inc;
        __syncthreads();
        u32 run = a + inc - tot;
        for (int k = 0; k < w; ++k) run += s_ws[k];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            s_cnt[tid * PER + k] = run;
            run += c[k];
        }
    }
    __syncthreads();
    for (u32 i = a + (u32)tid; i < b; i += IXS_T2) {
        const u32 k = sk[i];
        const u32 at = atomicAdd(&s_cnt[k - idbase], 1u);
        dk[at] = k, dv[at] = sv[i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(IXS_T2) void k_ixs_bin(const u32* __restrict__ tk, const u64* __restrict__ tv, const u32* __restrict__ plan, int wsh,
                                                    u32* ak, u64* av /*scratch (the level-1 input), bins wider than IXS_W only: written, then re-read by this workgroup*/,
                                                    u32* __restrict__ ok, u64* __restrict__ ov) {
    __shared__ u32 s_cnt[IXS_W];
    __shared__ u32 s_ws[IXS_T2 / 64];
    __shared__ u32 s_sub[34];
    const u32 bin = blockIdx.x;
    const u32 a = plan[(size_t)bin * IXS_G], b = plan[(size_t)(bin + 1) * IXS_G];
    if (a == b) return;
    const u32 idbase = bin << wsh;
    if (wsh <= 14) {
        ixs_group(tk, tv, a, b, idbase, ok, ov, s_cnt, s_ws);
        return;
    }
    // a bin of 2^wsh > 16384 ids: split by the id's digit above bit 14 first (<= 32 sub-ranges: NC < 2^32, <= 8192 bins), then every sub-range
    const u32 nsub = 1u << (wsh - 14);
    const int tid = threadIdx.x;
    if (tid < 34) s_sub[tid] = 0;
    __syncthreads();
    for (u32 i = a + (u32)tid; i < b; i += IXS_T2) atomicAdd(&s_sub[(tk[i] - idbase) >> 14], 1u);
    __syncthreads();
    if (tid == 0) {
        u32 run = a;
        for (u32 j = 0; j <= nsub; ++j) {
            const u32 c = j < nsub ? s_sub[j] : 0u;
            s_sub[j] = run;
            run += c;
        }
    }
    __syncthreads();
    u32 sub_lo[32];   // (kept per thread: the cursors below overwrite the starts)
    for (u32 j = 0; j < nsub; ++j) sub_lo[j] = s_sub[j];
    const u32 sub_end = s_sub[nsub];
    __syncthreads();
    for (u32 i = a + (u32)tid; i < b; i += IXS_T2) {
        const u32 k = tk[i];
        const u32 at = atomicAdd(&s_sub[(k - idbase) >> 14], 1u);
        ak[at] = k, av[at] = tv[i];
    }
    __threadfence();
    __syncthreads();
    for (u32 j = 0; j < nsub; ++j) {
        const u32 sa = sub_lo[j], sb = j + 1 < nsub ? sub_lo[j + 1] : sub_end;
        if (sa < sb) ixs_group(ak, av, sa, sb, idbase + (j << 14), ok, ov, s_cnt, s_ws);   // (wave-uniform condition: barriers inside are safe)
    }
}

// keys in [0, NC), or ~0 = no pair in this slot (n slots in all).  plan: ixsort_plan_elems(NC) u32; (tk, tv), (kout, vout): E pairs, E = the
// number of valid slots = the plan's last word after ixsort_count; (kin, vin) are overwritten when NC > 2^27.
const u32* ixsort_count(const u32* kin, u32 n, u32 NC, u32* plan, u32* scan_tmp, hipStream_t st) {   // -> device word holding E
    const int wsh = ixs_wsh(NC);
    const u32 nbins = ixsort_bins(NC);
    hipLaunchKernelGGL(k_ixs_count, dim3(IXS_G), dim3(IXS_T1), 0, st, kin, n, wsh, nbins, plan);
    scan_u32(plan, plan, (size_t)nbins * IXS_G + 1, false, scan_tmp, st);
    return plan + (size_t)nbins * IXS_G;
}
void ixsort_finish(u32* kin, u64* vin, u32 n, u32 NC, const u32* plan, u32* tk, u64* tv, u32* kout, u64* vout, hipStream_t st) {
    if (!n) return;
    const int wsh = ixs_wsh(NC);
    const u32 nbins = ixsort_bins(NC);
    hipLaunchKernelGGL(k_ixs_scatter, dim3(IXS_G), dim3(IXS_T1), 0, st, kin, vin, n, wsh, nbins, plan, tk, tv);
    hipLaunchKernelGGL(k_ixs_bin, dim3(nbins), dim3(IXS_T2), 0, st, tk, tv, plan, wsh, kin, vin, kout, vout);
}
void ixsort_pairs(u32* kin, u64* vin, u32 E, u32 NC, u32* plan, u32* scan_tmp, u32* tk, u64* tv, u32* kout, u64* vout, hipStream_t st) {
    if (!E) return;
    (void)ixsort_count(kin, E, NC, plan, scan_tmp, st);
    ixsort_finish(kin, vin, E, NC, plan, tk, tv, kout, vout, st);
}
