// k_index.hip -- per-chunk seed index build on the device (Fasta.build_msav, fsearch.py:2208-2280).
//
// The reference keeps a direct-addressed CSR: start[NC + 1] (480 MB at -M 120000000) + locus[E].  Only a few
// hundred thousand of the 120 M buckets are occupied, so here the bucket directory is an OPEN-ADDRESSED HASH MAP
// of the occupied bucket ids, a few MB that stay in L2 / Infinity Cache:
//
//   entries[0..E) u64   (subject_local << 32) | (tag << 24) | pos, grouped by ascending bucket id -- the very slot
//                       layout of the reference's CSR (tag = alphabet * S + pattern)
//   ub[0..U)      u32   occupied bucket ids, ascending;   ubeg[0..U] u32   first slot of each (ubeg[U] = E)
//   hkey / hval         hash map bucket id -> (first slot | count << 32), linear probing, load <= 1/2
//
// Build: (bucket, entry) pairs at fixed slots per position, invalid windows marked (k_index_windows_sparse: one pass; until round 6 count ->
// scan -> dense emit) -> grouping by bucket id (k_ixsort.hip: two counting passes, the first one's scan = the number of entries) -> run heads
// -> unique list, run lengths (threshold statistics) -> directory (bitmap + rank, or the hash map for very large -M).
// "bucket = hash % NC, no key check" -- collisions included -- is unchanged: the map is keyed by the bucket id.
//
// The reference stores bucket members in descending insertion order and later visits them in slot order; here
// the slot order inside a bucket is whatever the grouping leaves, because every downstream consumer re-derives the
// visiting order from the entry value itself (descending (subject, tag, pos) == descending insertion order) --
// until order_chunk (host_index.hip: the chunk's first dense pass, or so_chunk_download) puts the members in that very
// order.  The one order-dependent rule, "the very last locus slot is never read" (fsearch.py:2277, 2539), is kept by
// k_index_fixlast.
#include "common.h"
#include "kernels.h"
#include "seedhash.h"

// ONE pass instead of count + scan + emit (round 6): position p writes its A x S (bucket, entry) pairs to the fixed slots (p - p_lo) * AS + tag,
// an invalid window as bucket ~0 (never a bucket id: the directory kernels refuse NC - 1 and above) -- the grouping (k_ixsort.hip) skips those
// and its own scan yields the number of entries.  Saves the counting pass (0.17 ms per 50 k-sequence chunk), a scan and a host round trip.
__global__ __launch_bounds__(TILE_POS) void k_index_windows_sparse(const u32* __restrict__ words, const u32* __restrict__ pseq, const u32* __restrict__ off, u32 p_lo,
                                                                   u32 p_hi, u32 Ppad, u32 seq_lo, SeedCfg cfg, HashLut lut, u32 step, u32* __restrict__ bkt,
                                                                   u64* __restrict__ ent) {
    __shared__ u8 s_cls[TILE_POS + MAX_SEEDLEN];
    const u32 p0 = p_lo + blockIdx.x * TILE_POS;
    stage_classes(words, p0, Ppad, s_cls);
    __syncthreads();
    const u32 p = p0 + threadIdx.x;
    if (p >= p_hi) return;
    const u32 AS = (u32)(cfg.A * cfg.S);
    const size_t base = (size_t)(p - p_lo) * AS;
    bool live = s_cls[threadIdx.x] < HCLS_SEP;  // separator or x: no window starts here
    u32 j = 0, pos = 0;
    if (live) {
        j = pseq[p];
        pos = p - (off[j] + j);
        if (step > 1 && (pos % step) != 0) live = false;  // xrange(0, L - k + 1, step), fsearch.py:534
    }
    u32 bucket[MAX_PATTERNS];
    for (int a = 0; a < cfg.A; ++a) {
        const u32 mask = live ? hash_position(s_cls + threadIdx.x, cfg, lut.v[a], bucket) : 0u;
        for (int sd = 0; sd < cfg.S; ++sd) {
            const u32 tag = (u32)(a * cfg.S + sd);
            const bool ok = ((mask >> sd) & 1u) != 0;
            bkt[base + tag] = ok ? bucket[sd] : 0xFFFFFFFFu;
            if (ok) ent[base + tag] = ((u64)(j - seq_lo) << 32) | ((u64)tag << 24) | (u64)pos;
        }
    }
}

// ---- runs of equal bucket ids in the sorted pair list ---------------------------------------------------
__global__ __launch_bounds__(256) void k_run_heads(const u32* __restrict__ bkt, u32 E, u32* __restrict__ flags) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < E) flags[i] = (i == 0 || bkt[i] != bkt[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_run_list(const u32* __restrict__ bkt, const u32* __restrict__ flags, const u32* __restrict__ ridx,
                                                  u32 E, u32 U, u32* __restrict__ ub, u32* __restrict__ ubeg) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) ubeg[U] = E;
    if (i < E && flags[i]) {
        const u32 k = ridx[i];
        ub[k] = bkt[i];
        ubeg[k] = i;
    }
}

__global__ __launch_bounds__(256) void k_run_counts(const u32* __restrict__ ubeg, u32 U, u32* __restrict__ cnt) {
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (k < U) cnt[k] = ubeg[k + 1] - ubeg[k];
}

// ---- open-addressed map: bucket id -> (first slot | count << 32) -----------------------------------------
__device__ __forceinline__ u32 htab_hash(u32 b, int hshift) { return (b * 2654435761u) >> hshift; }

__global__ __launch_bounds__(256) void k_htab_insert(const u32* __restrict__ ub, const u32* __restrict__ ubeg, u32 U, u32* __restrict__ hkey,
                                                     u64* __restrict__ hval, int hshift, u32 hmask) {
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (k >= U) return;
    const u32 b = ub[k];
    u32 h = htab_hash(b, hshift);
    for (;;) {
        const u32 old = atomicCAS(&hkey[h], HTAB_EMPTY, b);
        if (old == HTAB_EMPTY) break;  // ids are unique: nobody else inserts b
        h = (h + 1u) & hmask;
    }
    hval[h] = (u64)ubeg[k] | ((u64)(ubeg[k + 1] - ubeg[k]) << 32);
}

// ---- bitmap + rank directory: bucket id -> position in the occupied-bucket list --------------------------------
// dir[w] (one 8-byte word per 32 bucket ids) = occupancy bits of buckets 32w .. 32w + 31 (low half) | number of occupied buckets
// below 32w (high half); bucket b is occupied iff its bit is set, and its place in ub / ubeg is high + popcount(bits below b).
// NC / 4 bytes (30 MB at -M 120000000: Infinity-Cache resident) against 12 bytes x 2 x U for the open-addressed map, and the build
// is one coalesced pass over the ascending list (the map's 15 M random compare-and-swaps were 1.2 ms of the 2.5 ms a config-3
// chunk takes to index).  Used when NC <= 2^28 (SOHIT_DIR_MAX); larger -M keep the map.
__global__ __launch_bounds__(256) void k_dir_build(const u32* __restrict__ ub, u32 U, u32* __restrict__ dir32) {
    const u32 k = blockIdx.x * 256u + threadIdx.x;
    if (k >= U) return;
    const u32 b = ub[k], w = b >> 5;
    if (k != 0 && (ub[k - 1] >> 5) == w) return;
    // the word's first occupied bucket (ub is ascending) collects the word's bits -- a handful of neighbours -- and stores the word
    // once: no atomics (round 3; 15 M atomic ORs were 0.32 ms per config-3 chunk)
    u32 bits = 1u << (b & 31u);
    for (u32 j = k + 1; j < U; ++j) {
        const u32 bj = ub[j];
        if ((bj >> 5) != w) break;
        bits |= 1u << (bj & 31u);
    }
    *reinterpret_cast<uint2*>(dir32 + 2 * (size_t)w) = make_uint2(bits, k);
}

// sum c, sum c^2, #non-empty, largest non-empty bucket id over counts[0..NC)
__global__ __launch_bounds__(256) void k_index_stats(const u32* __restrict__ counts, u32 NC, u64* __restrict__ stats /*[gridDim.x][4] partials*/) {
    u64 s1 = 0, s2 = 0;
    u32 nn = 0;
    u32 mbp1 = 0;  // (largest non-empty bucket id) + 1, 0 = none
    // branch-free, four independent 16-byte loads per thread and trip (the array is 4 x NC bytes, mostly zeros)
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < NC; i0 += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            v[u] = make_uint4(0, 0, 0, 0);
            if (i + 3 < NC) {
                v[u] = *reinterpret_cast<const uint4*>(counts + i);
            } else if (i < NC) {
                v[u].x = counts[i];
                if (i + 1 < NC) v[u].y = counts[i + 1];
                if (i + 2 < NC) v[u].z = counts[i + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = (u32)(i0 + u * stride);
            const u32 c[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1 += c[k];
                s2 += (u64)c[k] * c[k];
                nn += c[k] != 0;
                mbp1 = max(mbp1, c[k] ? i + k + 1u : 0u);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
        nn += __shfl_down(nn, o);
        u32 om = __shfl_down(mbp1, o);
        mbp1 = om > mbp1 ? om : mbp1;
    }
    // per-block partials, no atomics: same-address atomics serialise at ~90 per microsecond chip-wide,
    // which made 8 k wave-level atomics cost more than the 480 MB sweep itself
    __shared__ u64 s_part[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[w][0] = s1, s_part[w][1] = s2, s_part[w][2] = nn, s_part[w][3] = mbp1;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 a = 0, b = 0, c = 0, d = 0;
        for (int k = 0; k < 4; ++k) a += s_part[k][0], b += s_part[k][1], c += s_part[k][2], d = s_part[k][3] > d ? s_part[k][3] : d;
        u64* o = stats + 4 * (size_t)blockIdx.x;
        o[0] = a, o[1] = b, o[2] = c, o[3] = d;
    }
}

// second stage: reduce the per-block partials (nb x 4) into stats[0..4)
__global__ __launch_bounds__(256) void k_index_stats_final(const u64* __restrict__ part, u32 nb, u64* __restrict__ stats) {
    __shared__ u64 s_red[256][4];
    u64 a = 0, b = 0, c = 0, d = 0;
    for (u32 i = threadIdx.x; i < nb; i += 256) {
        a += part[4 * (size_t)i], b += part[4 * (size_t)i + 1], c += part[4 * (size_t)i + 2];
        d = part[4 * (size_t)i + 3] > d ? part[4 * (size_t)i + 3] : d;
    }
    s_red[threadIdx.x][0] = a, s_red[threadIdx.x][1] = b, s_red[threadIdx.x][2] = c, s_red[threadIdx.x][3] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_red[threadIdx.x][0] += s_red[threadIdx.x + o][0];
            s_red[threadIdx.x][1] += s_red[threadIdx.x + o][1];
            s_red[threadIdx.x][2] += s_red[threadIdx.x + o][2];
            if (s_red[threadIdx.x + o][3] > s_red[threadIdx.x][3]) s_red[threadIdx.x][3] = s_red[threadIdx.x + o][3];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) stats[threadIdx.x] = s_red[0][threadIdx.x];
}

// Keep the reference's "slot len(locus)-1 is never read": that slot belongs to the last non-empty
// bucket b* and, in the reference's descending order, holds b*'s smallest entry.  Move b*'s
// smallest entry to slot E-1; the lookup clamps every bucket end to E-1 exactly like get_bin_mem.
__global__ __launch_bounds__(256) void k_index_fixlast(u64* __restrict__ entries, u32 lo /*first slot of the last occupied bucket*/, u32 E) {
    __shared__ u64 s_min[256];
    __shared__ u32 s_idx[256];
    u64 mn = ~0ull;
    u32 mi = E - 1;
    for (u32 i = lo + threadIdx.x; i < E; i += 256) {
        u64 v = entries[i];
        if (v < mn) mn = v, mi = i;
    }
    s_min[threadIdx.x] = mn, s_idx[threadIdx.x] = mi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o && s_min[threadIdx.x + o] < s_min[threadIdx.x]) {
            s_min[threadIdx.x] = s_min[threadIdx.x + o];
            s_idx[threadIdx.x] = s_idx[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u32 i = s_idx[0];
        if (i != E - 1) {
            u64 a = entries[i];
            entries[i] = entries[E - 1];
            entries[E - 1] = a;
        }
    }
}


// ---- key deltas for the seed-lookup kernel ----------------------------------------------------------
// For one key layout (sh_subj, sh_diag) every index entry contributes a FIXED 64-bit addend to the sort
// key of any hit that visits it:   D = (subject << sh_subj) + ((maxslen - pos) << sh_diag) + tag,
// so the lookup kernel computes  key = kbase(seed) + D  with one 64-bit add.  D < 2^sh_q <= 2^63, which
// leaves bit 63 for the rare entries at offset 0 of their sequence (the reference attributes those to
// the previous non-empty sequence, fsearch.py:134-153 strict bisect): they carry
// (1 << 63) | (subject << 8) | tag and are resolved on a slow path.
__global__ __launch_bounds__(256) void k_encode_delta(const u64* __restrict__ entries, u32 E, int sh_subj, int sh_diag, u32 maxslen,
                                                      u64* __restrict__ dkeys) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= E) return;
    const u64 e = entries[i];
    const u32 j = (u32)(e >> 32), pos = (u32)e & 0xFFFFFFu, tag = (u32)(e >> 24) & 0xFFu;
    dkeys[i] = pos == 0 ? ((1ull << 63) | ((u64)j << 8) | (u64)tag)
                        : (((u64)j << sh_subj) + ((u64)(maxslen - pos) << sh_diag) + (u64)tag);
}

// Compact form (4 bytes per index entry), round 4: BANDED DIAGONAL IDS.  Every (subject, diagonal) pair of the chunk gets one
// integer  G = gbase[subject] + qpos - pos  that grows with (subject, diagonal) -- so a hit's sort key is still one addition,
// key = kbase(seed) + (c >> ba << sh_diag) + (c & amask)  with the entry's addend  c = (gbase[subject] - pos) << ba | tag  --
// but the number of ids a subject owns follows ITS length instead of the chunk's longest: with k = KeyLayout::bd "diagonal" bits
// and the pass's queries shorter than 2^bp, a subject of length <= C = 2^k - 2^bp owns one band of 2^k ids (gbase = band << k | C),
// a longer one owns ceil((length + 2^bp) / 2^k) consecutive bands (gbase = first band << k | length).  The KeyLayout's "subject" field
// then is the BAND id, and k_ungap maps a band back to (subject, gbase) through a table when some subject owns several; when none
// does, band == subject and the diagonal field is qpos - pos + C exactly as before round 4.  One 30 000-residue protein in a chunk
// used to widen the diagonal field of every hit of the chunk to 16 bits, which pushed the hit word of the bucketed binning past
// 32 bits and the whole pass to the sorted path.
// An entry at offset 0 of its sequence is resolved HERE, once per index entry instead of once per visiting hit: the
// reference's strict `soas[j] < x` (fsearch.py:2685-2688) attributes it to the previous non-empty sequence of the chunk at
// sst = that sequence's length.  When there is no such sequence the reference resolves index -1 and the hit never scores: those
// entries are all-ones (never a real value: the host keeps band and diagonal bits + ba <= 31), which the lookup kernels read as "drop".
__global__ __launch_bounds__(256) void k_encode_band32(const u64* __restrict__ entries, u32 E, int ba, const u32* __restrict__ gbase /*per chunk sequence*/,
                                                       const u32* __restrict__ roff /*chunk-local offsets*/, u32* __restrict__ dk32) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= E) return;
    const u64 e = entries[i];
    u32 j = (u32)(e >> 32);
    u32 pos = (u32)e & 0xFFFFFFu;
    const u32 tag = (u32)(e >> 24) & 0xFFu;
    if (pos == 0) {
        while (j > 0 && roff[j] == roff[j - 1]) --j;
        if (j == 0) {
            dk32[i] = 0xFFFFFFFFu;
            return;
        }
        j -= 1;
        pos = roff[j + 1] - roff[j];
    }
    dk32[i] = ((gbase[j] - pos) << ba) | tag;
}

void launch_encode_band32(const u64* entries, u32 E, int ba, const u32* gbase, const u32* roff, u32* dk32, hipStream_t st) {
    if (!E) return;
    hipLaunchKernelGGL(k_encode_band32, dim3((E + 255) / 256), dim3(256), 0, st, entries, E, ba, gbase, roff, dk32);
}

void launch_encode_delta(const u64* entries, u32 E, int sh_subj, int sh_diag, u32 maxslen, u64* dkeys, hipStream_t st) {
    if (!E) return;
    hipLaunchKernelGGL(k_encode_delta, dim3((E + 255) / 256), dim3(256), 0, st, entries, E, sh_subj, sh_diag, maxslen, dkeys);
}

void launch_index_windows_sparse(const u32* words, const u32* pseq, const u32* off, u32 p_lo, u32 p_hi, u32 Ppad, u32 seq_lo, const SeedCfg& cfg, const HashLut& lut, u32 step,
                                 u32* bkt, u64* ent, hipStream_t st) {
    if (p_hi <= p_lo) return;
    const u32 nb = (p_hi - p_lo + TILE_POS - 1) / TILE_POS;
    hipLaunchKernelGGL(k_index_windows_sparse, dim3(nb), dim3(TILE_POS), 0, st, words, pseq, off, p_lo, p_hi, Ppad, seq_lo, cfg, lut, step, bkt, ent);
}

void launch_run_heads(const u32* bkt, u32 E, u32* flags, hipStream_t st) {
    if (E) hipLaunchKernelGGL(k_run_heads, dim3((E + 255) / 256), dim3(256), 0, st, bkt, E, flags);
}
void launch_run_list(const u32* bkt, const u32* flags, const u32* ridx, u32 E, u32 U, u32* ub, u32* ubeg, u32* cnt, hipStream_t st) {
    if (!E) return;
    hipLaunchKernelGGL(k_run_list, dim3((E + 255) / 256), dim3(256), 0, st, bkt, flags, ridx, E, U, ub, ubeg);
    hipLaunchKernelGGL(k_run_counts, dim3((U + 255) / 256), dim3(256), 0, st, ubeg, U, cnt);
}
void launch_dir_build(const u32* ub, u32 U, u64* dir, hipStream_t st) {
    if (U) hipLaunchKernelGGL(k_dir_build, dim3((U + 255) / 256), dim3(256), 0, st, ub, U, reinterpret_cast<u32*>(dir));
}

void launch_htab_insert(const u32* ub, const u32* ubeg, u32 U, u32* hkey, u64* hval, int hshift, u32 hmask, hipStream_t st) {
    if (U) hipLaunchKernelGGL(k_htab_insert, dim3((U + 255) / 256), dim3(256), 0, st, ub, ubeg, U, hkey, hval, hshift, hmask);
}

// stats_buf: 4 results followed by INDEX_STATS_BLOCKS x 4 per-block partials
void launch_index_stats(const u32* counts, u32 NC, u64* stats_buf, hipStream_t st) {
    u32 nb = (u32)std::min<size_t>(INDEX_STATS_BLOCKS, ((size_t)NC / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_index_stats, dim3(nb), dim3(256), 0, st, counts, NC, stats_buf + 4);
    hipLaunchKernelGGL(k_index_stats_final, dim3(1), dim3(256), 0, st, stats_buf + 4, nb, stats_buf);
}

void launch_index_fixlast(u64* entries, u32 lo, u32 E, hipStream_t st) {
    if (E == 0) return;
    hipLaunchKernelGGL(k_index_fixlast, dim3(1), dim3(256), 0, st, entries, lo, E);
}
