// seedhash.h -- device code shared by the index-build and query-seed kernels: LDS staging of
// 5-bit hash classes and the reduced-alphabet spaced-seed FNV-1a hash (fsearch.py:519-556).
#pragma once
#include "common.h"

#define TILE_POS 256

struct HashLut {
    u32 v[MAX_ALPHA][32];
};

// Unpack the hash classes of packed positions [p0, p0 + TILE_POS + MAX_SEEDLEN) into LDS.
// `words` carries two zero pad words past the last group, so w + 1 is always readable.
__device__ __forceinline__ void stage_classes(const u32* __restrict__ words, u32 p0, u32 Ppad, u8* s_cls) {
    for (int k = threadIdx.x; k < TILE_POS + MAX_SEEDLEN; k += blockDim.x) {
        u32 p = p0 + (u32)k;
        u8 c = HCLS_SEP;
        if (p < Ppad) {
            u64 bit = (u64)p * 5u;
            u32 w = (u32)(bit >> 5), sh = (u32)(bit & 31u);
            u64 v = (u64)words[w] | ((u64)words[w + 1] << 32);
            c = (u8)((v >> sh) & 31u);
        }
        s_cls[k] = c;
    }
}

// Buckets of every seed pattern at one position for alphabet `a`.  bucket[s] is valid iff bit s
// of the returned mask is set: the window covers no class >= 30 and no earlier pattern of the
// same alphabet produced the same (bucket, position) (the reference's `visit` dict, 529, 554-556).
__device__ __forceinline__ u32 hash_position(const u8* cls /* LDS, at the position */, const SeedCfg& cfg, const u32* lut_a,
                                             u32* bucket /*[MAX_PATTERNS]*/) {
    u32 mask = 0;
    for (int s = 0; s < cfg.S; ++s) {
        const int k = cfg.klen[s];
        const u32 care = cfg.care[s];
        u32 n = 0x811c9dc5u;
        bool ok = true;
        for (int j = 0; j < k; ++j) {
            u32 c = cls[j];
            if (c >= HCLS_SEP) {
                ok = false;
                break;
            }
            if ((care >> j) & 1u) {
                n ^= lut_a[c];
                n *= 0x01000193u;
            }
        }
        n ^= (u32)s;
        n *= 0x01000193u;
        u32 b = n % cfg.nc;
        if (ok) {
            for (int s2 = 0; s2 < s; ++s2)
                if (((mask >> s2) & 1u) && bucket[s2] == b) ok = false;
        }
        bucket[s] = b;
        if (ok) mask |= 1u << s;
    }
    return mask;
}
