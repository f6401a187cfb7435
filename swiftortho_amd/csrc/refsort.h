// refsort.h -- the reference's non-stable fixed-pivot quicksort (fsearch.py:260-327; *_u twins
// 189-256) as a device function.  Ties decide cap membership, the top-vmax cut and output order,
// so the algorithm is reproduced move for move: insertion sort below 7 elements, pivot index
// l + 3 at exactly 7, else l + int(0.3745401188473625 * gap) (MT19937 re-seeded with 42 on every
// call), Hoare-style partition.  Sub-ranges are independent, so an explicit stack that handles
// the smaller side first (depth <= log2 n) gives the same result as the recursion.
#pragma once
#include "common.h"

// `limit`: only positions [0, limit) of the result are needed -- sub-ranges that start at or beyond
// it are left unsorted (they cannot influence earlier positions: sub-ranges are independent).
template <class KeyFn>
__device__ inline void ref_qsort_dev(u32* x, int n, KeyFn key, int limit = 0x7fffffff) {
    int stk[2 * 48];
    int sp = 0;
    stk[sp++] = 0;
    stk[sp++] = n - 1;
    while (sp > 0) {
        int r = stk[--sp], l = stk[--sp];
        if (r <= l || l >= limit) continue;
        int gap = r - l + 1;
        if (gap < 7) {
            for (int i = l; i < r + 1; ++i) {  // insort(x, l, r + 1)
                u32 v = x[i];
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
        u32 t = x[l];
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
        // ranges [l, med-1] and [med+1, r]; push the larger first so the smaller is handled next
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
