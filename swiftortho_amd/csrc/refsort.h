// refsort.h -- the reference's non-stable fixed-pivot quicksort (fsearch.py:260-327; *_u twins
// 189-256) as device functions.  Ties decide cap membership, the top-vmax cut and output order,
// so the algorithm is reproduced move for move: insertion sort below 7 elements, pivot index
// l + 3 at exactly 7, else l + int(0.3745401188473625 * gap) (MT19937 re-seeded with 42 on every
// call), Hoare-style partition.  Sub-ranges are independent, so the order in which they are
// processed does not matter.
//
//   ref_qsort_range   one thread, explicit stack (smaller side first, depth <= log2 n)
//   wave_ref_qsort    one wave64, array in LDS: ranges >= WQS_PAR are partitioned by the whole
//                     wave with EXACTLY the permutation the sequential partition produces; smaller
//                     ranges are collected and sorted afterwards one per lane with ref_qsort_range.
#pragma once
#include "common.h"

// `limit`: only positions [0, limit) of the result are needed -- sub-ranges that start at or beyond
// it are left unsorted (they cannot influence earlier positions).
template <class T, class KeyFn>
__device__ inline void ref_qsort_range(T* x, int l0, int r0, KeyFn key, int limit = 0x7fffffff) {
    int stk[2 * 48];
    int sp = 0;
    stk[sp++] = l0;
    stk[sp++] = r0;
    while (sp > 0) {
        int r = stk[--sp], l = stk[--sp];
        if (r <= l || l >= limit) continue;
        int gap = r - l + 1;
        if (gap < 7) {
            for (int i = l; i < r + 1; ++i) {  // insort(x, l, r + 1)
                T v = x[i];
                auto pivot = key(v);
                int j = i - 1;
                while (j >= l) {
                    if (key(x[j]) <= pivot) break;
                    x[j + 1] = x[j];
                    --j;
                }
                x[j + 1] = v;
            }
            continue;
        }
        int m = (gap == 7) ? l + 3 : l + (int)(0.3745401188473625 * (double)gap);
        T t = x[l];
        x[l] = x[m];
        x[m] = t;
        auto pivot = key(x[l]);
        int i = l, j = r + 1;
        for (;;) {
            ++i;
            while (i <= r && key(x[i]) < pivot) ++i;
            --j;
            while (key(x[j]) > pivot) --j;
            if (i > j) break;
            t = x[i];
            x[i] = x[j];
            x[j] = t;
        }
        t = x[l];
        x[l] = x[j];
        x[j] = t;
        int med = j;
        int n1 = med - l, n2 = r - med;
        if (n1 > n2) {
            stk[sp++] = l, stk[sp++] = med - 1;
            stk[sp++] = med + 1, stk[sp++] = r;
        } else {
            stk[sp++] = med + 1, stk[sp++] = r;
            stk[sp++] = l, stk[sp++] = med - 1;
        }
    }
}

template <class T, class KeyFn>
__device__ inline void ref_qsort_dev(T* x, int n, KeyFn key, int limit = 0x7fffffff) {
    ref_qsort_range(x, 0, n - 1, key, limit);
}

// ---------------------------------------------------------------------------------------------
// Wave-parallel exact replay.  For a range [l, r] with pivot p = key(x[l]) (after the pivot swap):
//   L_1 < L_2 < ...   positions in (l, r] with key >= p   (where the upward scan stops)
//   R_1 > R_2 > ...   positions in [l, r] with key <= p   (where the downward scan stops; x[l] is the last)
// The sequential partition swaps (L_k, R_k) for k = 1..m, m = #{k : L_k < R_k} (swapped elements are
// never revisited, so the pairing is a function of the ORIGINAL array), then exchanges the pivot
// with j = max(R_{m+1}, L_m), or max(R_{m+2}, L_m) when L_{m+1} == R_{m+1} (the scans meet on an
// element equal to the pivot: a self-swap, one more scan step).  Ranks come from ballots.
// Block = exactly one wave (64 threads); x, Lpos, Rpos (n u16 each) and leaf (2 * WQS_LEAF ints) in LDS.
// ---------------------------------------------------------------------------------------------
#define WQS_PAR 32   // ranges below this go to the per-lane sequential leaves (>= 8: the gap == 7 pivot rule lives there);
                     // config 2 candidate sort: 64 -> 1.27 ms, 32 -> 1.04, 24 -> 1.05, 10 -> 1.40
#define WQS_LEAF 512

// T = element type (u32 words in LDS, or u64 words in global memory for segments that do not fit),
// PT = position type of the misfit lists (u16 / u32).
// LEAFCAP = capacity of the leaf list (2 ints each); when it fills up the collected leaves are sorted at once and the list starts over
// (a 300-element list produces 20-40 leaves).
// PF = 64-element steps a scan loads before the ballots consume them: 4 for arrays in LDS, 16 for arrays in global memory (a scan over
// a 30 000-element range is a chain of load -> ballot -> store round trips, ~2 us each: the deeper the batch, the fewer of them).
// leafbuf (arrays in global memory): 64 x WQS_LEAFBUF elements of LDS; a lane sorts its leaf there instead of in place (an insertion
// sort is a chain of dependent accesses: ~100 cycles each in LDS, ~1000 in global memory).
#define WQS_LEAFBUF (WQS_PAR + 1)   // odd stride: the lanes' copies start in different banks
template <class T, class KeyFn>
__device__ inline void wave_sort_leaves(T* x, const int* leaf, int nleaf, KeyFn key, int limit, T* leafbuf) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < nleaf; i += 64) {
        const int l = leaf[2 * i], r = leaf[2 * i + 1];
        if (!leafbuf) {
            ref_qsort_range(x, l, r, key, limit);
        } else if (l < limit) {
            T* b = leafbuf + lane * WQS_LEAFBUF;
            for (int k = 0; k <= r - l; ++k) b[k] = x[l + k];
            ref_qsort_range(b, 0, r - l, key, limit - l);
            for (int k = 0; k <= r - l; ++k) x[l + k] = b[k];
        }
    }
}

template <int LEAFCAP = WQS_LEAF, int PF = 4, class T, class PT, class KeyFn>
__device__ inline void wave_ref_qsort(T* x, int n, KeyFn key, int limit, PT* Lpos, PT* Rpos, int* leaf, T* leafbuf = nullptr) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int stk[2 * 40];
    int sp = 0, nleaf = 0;
    stk[sp++] = 0;
    stk[sp++] = n - 1;
    while (sp > 0) {  // wave-uniform control flow: every lane holds the same stack
        const int r = stk[--sp], l = stk[--sp];
        if (r <= l || l >= limit) continue;
        const int gap = r - l + 1;
        if (gap < WQS_PAR) {
            if (nleaf == LEAFCAP) {  // leaf list full: sort the collected leaves now, one per lane (disjoint ranges, independent of
                                     // everything still on the stack), and start a new list.  (Until round 4 a full list sent every
                                     // further range, whatever its size, to ONE lane: a 30 000-element list took 42 ms.)
                __syncthreads();
                wave_sort_leaves(x, leaf, nleaf, key, limit, leafbuf);
                __syncthreads();
                nleaf = 0;
            }
            if (lane == 0) leaf[2 * nleaf] = l, leaf[2 * nleaf + 1] = r;
            ++nleaf;
            continue;
        }
        const int m = l + (int)(0.3745401188473625 * (double)gap);
        if (lane == 0) {
            T t = x[l];
            x[l] = x[m];
            x[m] = t;
        }
        __syncthreads();
        const auto p = key(x[l]);
        int cntL = 0, cntR = 0;
        // the two scans read four 64-element steps ahead of the ballots that consume them (a load -> ballot -> store chain per step
        // leaves the wave waiting on memory once per step: the global-memory instance spent ~1 us per 64 elements)
        for (int base = l + 1; base <= r; base += PF * 64) {
            T v[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (PF > 4 && base + u * 64 > r) break;   // (wave-uniform: nothing of the range left for this step)
                const int t = base + u * 64 + lane;
                v[u] = x[t <= r ? t : r];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (PF > 4 && base + u * 64 > r) break;
                const int t = base + u * 64 + lane;
                const bool f = (t <= r) && !(key(v[u]) < p);
                const unsigned long long bal = __ballot(f);
                if (f) Lpos[cntL + __popcll(bal & lt)] = (PT)(t - l);
                cntL += __popcll(bal);
            }
        }
        for (int base = r; base >= l; base -= PF * 64) {
            T v[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (PF > 4 && base - u * 64 < l) break;
                const int t = base - u * 64 - lane;
                v[u] = x[t >= l ? t : l];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (PF > 4 && base - u * 64 < l) break;
                const int t = base - u * 64 - lane;
                const bool f = (t >= l) && !(key(v[u]) > p);
                const unsigned long long bal = __ballot(f);
                if (f) Rpos[cntR + __popcll(bal & lt)] = (PT)(t - l);
                cntR += __popcll(bal);
            }
        }
        __syncthreads();
        const int K = cntL < cntR ? cntL : cntR;
        int mm = 0;
        for (int base = 0; base < K; base += 64) {
            const int k = base + lane;
            const bool f = (k < K) && (Lpos[k < K ? k : 0] < Rpos[k < K ? k : 0]);
            mm += __popcll(__ballot(f));
        }
        for (int k = lane; k < mm; k += 64) {
            const int a = l + (int)Lpos[k], b = l + (int)Rpos[k];
            const T t = x[a];
            x[a] = x[b];
            x[b] = t;
        }
        const int Lm = mm > 0 ? (int)Lpos[mm - 1] : -1;
        const bool tie = (mm < cntL) && (Lpos[mm] == Rpos[mm]);
        const int Rn = tie ? (int)Rpos[mm + 1] : (int)Rpos[mm];
        const int jrel = Rn > Lm ? Rn : Lm;
        __syncthreads();
        if (lane == 0) {
            T t = x[l];
            x[l] = x[l + jrel];
            x[l + jrel] = t;
        }
        __syncthreads();
        const int med = l + jrel;
        if (med - l > r - med) {
            stk[sp++] = l, stk[sp++] = med - 1;
            stk[sp++] = med + 1, stk[sp++] = r;
        } else {
            stk[sp++] = med + 1, stk[sp++] = r;
            stk[sp++] = l, stk[sp++] = med - 1;
        }
    }
    __syncthreads();
    wave_sort_leaves(x, leaf, nleaf, key, limit, leafbuf);
    __syncthreads();
}
