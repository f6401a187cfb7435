// k_bucket.hip -- diagonal binning WITHOUT a sort (fsearch.py:2679-2719: the `hits` dict keyed (subject, qst - sst)).
//
// The reference drops every seed hit into a dict keyed by (subject, diagonal).  The first MI355X version of this stage
// wrote an 8-byte key per hit and ran a library segmented radix sort over them (77 B of HBM traffic per hit).  What the
// extension kernel needs is weaker than sorted order: the hits of one (query, subject, diagonal) group next to each other,
// by ascending query position, and neighbouring groups close in the subject array.  This file produces exactly that:
//
//   1. tiles           a query's hit ordinals are cut into tiles of <= 1024 (k_bkt_ntiles / k_bkt_tiledesc), so a tile never
//                      straddles two queries.
//   2. k_bkt_pass      the seed-lookup kernel, run twice.  A BUCKET is (query, range of W = 2^wb diagonal bands -- one per chunk
//                      sequence, several for the sequences too long for one, see k_encode_band32); inside
//                      its bucket a hit is ONE 32-bit word  subject_low | diagonal | qpos.  Pass 1 counts the hits of
//                      every (range, tile) in a per-wave LDS histogram and stores the counts range-major; an exclusive scan
//                      of that matrix IS the scatter plan (buckets range-major, tiles in order inside a bucket); pass 2
//                      recomputes the hits from the 4-byte index addends (L2 / Infinity-Cache resident) and writes each to
//                      its final place.  No global atomics, 4 bytes written per hit instead of 8.
//   3. k_bkt_group     one workgroup per bucket (1-2 k hits): an exact bucket-local sort by (subject, diagonal, qpos) shaped for
//                      the data -- LDS counting sort by subject, then a rank inside each subject's (mostly tiny) segment.
//   The extension kernel (k_group.hip: k_ungap) then walks the grouped keys exactly as it walked the sorted ones.
//
// The bucket-local sort is a full one, so the scatter needs no stable ranks: an LDS fetch-add hands out slots inside a tile.
// Scope: one alphabet x one seed pattern, compact index addends and wb + bd + bp <= 31 with wb <= 10; otherwise, or when one
// subject alone brings more than BG_CAP hits to a query, the host runs the pass on the sorted path.
#include "common.h"
#include "kernels.h"

#define BK_WAVES 4
#define BK_ITERS 16   // 64-hit steps per wave tile (32: scatter 0.72 -> 1.0 ms on config 2: register pressure)
#define BK_HITS (64 * BK_ITERS)
#define BK_SEEDS 256

__device__ __forceinline__ void bk_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ u32 bk_scan_max(u32 x) {  // inclusive max-scan over the wave (DPP)
    // (a 16-bit v_max_u16_dpp form, 2.4 against 4.2 cycles in isolation, was measured here: no difference in the kernel)
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}

__device__ __forceinline__ u32 bk_scan_add(u32 x) {  // inclusive add-scan over the wave (DPP; __shfl_up compiles to ds_bpermute round trips)
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

// ---- tiles: <= BK_HITS consecutive hit ordinals of ONE query ------------------------------------------------------
__global__ __launch_bounds__(256) void k_bkt_ntiles(const u32* __restrict__ qseg /*first hit ordinal per pass query, + end*/, u32 nqp,
                                                    u32* __restrict__ ntile /*nqp + 1*/) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i > nqp) return;
    ntile[i] = i < nqp ? (qseg[i + 1] - qseg[i] + BK_HITS - 1) / BK_HITS : 0u;
}

// tile t -> {query (relative to the pass), first ordinal, first seed, last seed}
__global__ __launch_bounds__(256) void k_bkt_tiledesc(const u32* __restrict__ qseg, const u32* __restrict__ t0 /*nqp + 1*/, u32 nqp, u32 NT,
                                                      const u32* __restrict__ cs_hoff, u32 K, uint4* __restrict__ td) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    if (t >= NT) return;
    u32 lo = 0, hi = nqp;  // largest q with t0[q] <= t (t0[0] == 0; queries without hits own no tile and are skipped by `<=`)
    while (hi - lo > 1) {
        const u32 m = (lo + hi) >> 1;
        if (t0[m] <= t) lo = m;
        else hi = m;
    }
    const u32 q = lo;
    const u32 first = qseg[q] + (t - t0[q]) * BK_HITS;
    const u32 last = min(first + BK_HITS, qseg[q + 1]) - 1u;
    u32 a = 0, b = K;  // largest k with cs_hoff[k] <= first
    while (b - a > 1) {
        const u32 m = (a + b) >> 1;
        if (cs_hoff[m] <= first) a = m;
        else b = m;
    }
    const u32 k0 = a;
    b = K;
    while (b - a > 1) {
        const u32 m = (a + b) >> 1;
        if (cs_hoff[m] <= last) a = m;
        else b = m;
    }
    td[t] = make_uint4(q, first, k0, a);
}

// One hit: index addend c (k_encode_band32 with ba == 0: gbase[subject] - pos, or all-ones for the offset-0 entries the reference
// drops, fsearch.py:2685-2688) + the seed's query position = the hit's banded diagonal id G (band << bd | diagonal inside the band;
// the addition may carry into the band bits: that is the next band of a long subject) -> band range and 32-bit word.  Branch-free:
// the kernels' scalar unit, shared by the four SIMDs of a CU, was saturated by the exec-mask bookkeeping of a per-hit slow path
// (round 2: 1100 scalar against 800 vector instructions per tile).
__device__ __forceinline__ bool bk_hit(u32 c, u32 qpos, const BktLayout& L, u32& range, u32& word) {
    const u32 g = c + qpos;
    const int gb = L.wb + L.bd;
    range = g >> gb;
    word = ((g & ((1u << gb) - 1u)) << L.bp) | qpos;
    return (int)c >= 0;
}

// Per-wave LDS histogram of the tile's hits per range, REPLICATED: lane l counts in copy l mod ncopy.  Hits of one seed arrive
// ordered by subject, so neighbouring lanes mostly hit the SAME range: one counter per range would serialise them on one
// address (SQ_LDS_ADDR_CONFLICT 84 % of the LDS cycles), and copies a power-of-two stride apart would still share the bank.
// The copies therefore sit `cstride` = Rp + 32 / ncopy words apart: the ncopy counters of one range fall into ncopy different
// banks spread evenly over the 32.
#define BK_HIST_WORDS(staged) ((staged) ? 640 : 560)   // 560 + the 896 phase-A words: seven 4-wave workgroups per CU
struct BkHist {
    u32 ncopy, cstride;   // copies (power of two), words between two copies
    u32 gbase;            // staged scatter: where the per-run global offsets live inside the histogram array
};
static BkHist bk_hist_layout(u32 R, bool staged, bool skewed) {
    u32 Rp = 1;
    while (Rp < R) Rp <<= 1;
    BkHist h;
    if (!skewed) {   // round-2 layout (SOHIT_BK_SKEW=0): copies Rp words apart
        h.ncopy = std::max(1u, std::min(8u, 512u / Rp));
        h.cstride = Rp;
    } else {
        const u32 room = BK_HIST_WORDS(staged) - (staged ? Rp : 0u);
        h.ncopy = 8;
        while (h.ncopy > 1 && h.ncopy * (Rp + 32u / h.ncopy) > room) h.ncopy >>= 1;
        h.cstride = h.ncopy > 1 ? Rp + 32u / h.ncopy : Rp;
    }
    h.gbase = h.ncopy * h.cstride;
    return h;
}
#define BK_STAGE_RMAX 128   // the staged scatter keeps one global offset per non-empty run of the tile in LDS

// SCATTER = false: mat[tile * R + range] = hits of the tile in the range (every entry written);
// SCATTER = true : mat holds the exclusive scan of those counts IN RANGE-MAJOR ORDER (k_bkt_colsum / k_bkt_colscan) = where the
// tile's hits of each range go in `out`.  The matrix is stored tile-major so that a tile's R entries are one contiguous run for
// the wave that writes / reads them (range-major storage cost R scattered 4-byte accesses per tile in both passes).
// STAGED (scatter only, R <= BK_STAGE_RMAX): the tile's words are first put in LDS ordered by range -- the per-range LDS
// counters give every hit its place -- and then written out position by position: one store instruction covers 64
// consecutive staged words = two to four runs of consecutive addresses, instead of 64 lanes landing in ~R different lines
// (the texture-address path takes a cycle per distinct line per instruction: the round-2 kernel spent more time issuing
// its sixteen scattered stores than reading the index).
#ifndef BK_NT
#define BK_NT 1   // index entries are read once per pass: non-temporal loads
#endif
template <bool SCATTER, bool STAGED, int WPE = 6>
__global__ __launch_bounds__(64 * BK_WAVES, WPE) void k_bkt_pass(const uint4* __restrict__ td, const u32* __restrict__ qseg, u32 NT,
                                                              const u32* __restrict__ cs_hoff, const u32* __restrict__ cs_base,
                                                              const u64* __restrict__ cs_kbase, const u32* __restrict__ dk32,
                                                              BktLayout L, BkHist HL, u32* __restrict__ mat,
                                                              u32* __restrict__ out) {
    // phase A: owner marks (u16 x BK_HITS) | seed bases (u32 x BK_SEEDS) | seed qpos (u16 x BK_SEEDS); phase B (staged): the words
    constexpr u32 MEMW = STAGED ? BK_HITS : BK_HITS / 2 + BK_SEEDS + BK_SEEDS / 2;
    __shared__ u32 s_mem_all[BK_WAVES][MEMW];
    __shared__ u32 s_hist_all[BK_WAVES][BK_HIST_WORDS(STAGED)];
    static_assert(BK_HITS / 2 + BK_SEEDS + BK_SEEDS / 2 <= BK_HITS, "phase A arrays must fit the staging area");
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs, so XCD x takes the x-th contiguous eighth of the
    // tiles: the tiles of one query, and the output lines they fill, stay in one L2.
    const u32 per = gridDim.x >> 3;  // the grid is a multiple of 8 workgroups
    const u32 lb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const u32 t = __builtin_amdgcn_readfirstlane(lb * BK_WAVES + w);
    if (t >= NT) return;
    u16* s_owner = reinterpret_cast<u16*>(s_mem_all[w]);
    u32* s_base = s_mem_all[w] + BK_HITS / 2;
    u16* s_qpos = reinterpret_cast<u16*>(s_mem_all[w] + BK_HITS / 2 + BK_SEEDS);
    u32* s_stage = s_mem_all[w];
    u32* s_hist = s_hist_all[w];
    const uint4 d = td[t];
    const u32 q = __builtin_amdgcn_readfirstlane(d.x), lo = __builtin_amdgcn_readfirstlane(d.y);
    const u32 k0 = __builtin_amdgcn_readfirstlane(d.z), k1 = __builtin_amdgcn_readfirstlane(d.w);
    const u32 len = min((u32)BK_HITS, __builtin_amdgcn_readfirstlane(qseg[q + 1]) - lo);
    const u32 ns = k1 - k0 + 1;
    const u32 pmask = (1u << L.bp) - 1u;
    const u32 ncopy = HL.ncopy, cstride = HL.cstride;
    const u32 cbase = (lane & (ncopy - 1u)) * cstride;
#pragma unroll
    for (u32 i = 0; i < BK_ITERS / 4; ++i) reinterpret_cast<uint2*>(s_owner)[i * 64 + lane] = make_uint2(0, 0);
    for (u32 r = lane; r < HL.gbase; r += 64) s_hist[r] = 0;
    bk_wave_sync();
    for (u32 i = lane; i < ns; i += 64) {
        if (i < BK_SEEDS) {
            s_base[i] = cs_base[k0 + i];
            s_qpos[i] = (u16)((u32)(cs_kbase[k0 + i] >> L.sh_qpos) & pmask);
        }
        if (i) {
            const u32 o = cs_hoff[k0 + i] - lo;
            if (o < len) s_owner[o] = (u16)i;
        }
    }
    bk_wave_sync();
    u32 word[BK_ITERS], slot[BK_ITERS];  // slot: histogram word (copy, range) | rank << 10; ~0u = no hit
    u32 carry = 0;
    // all 16 index reads of the tile are issued before the first one is used (a dependent load -> LDS fetch-add chain per
    // step would leave the wave waiting on L2 sixteen times).  Wave-uniform choice of where the seeds' bases come from: LDS
    // when the tile's seeds fit the stage (a per-lane choice between an LDS and a global address compiles to flat loads).
    if (ns <= BK_SEEDS) {
#pragma unroll
        for (int it = 0; it < BK_ITERS; ++it) {
            const u32 hl = it * 64 + lane;
            const u32 inc = bk_scan_max((u32)s_owner[hl]);
            const u32 a = max(inc, carry);
            carry = max(carry, (u32)__builtin_amdgcn_readlane((int)inc, 63));
            slot[it] = (u32)s_qpos[a];                                    // the seed's qpos, for now
            word[it] = BK_NT ? __builtin_nontemporal_load(&dk32[s_base[a] + lo + min(hl, len - 1u)]) : dk32[s_base[a] + lo + min(hl, len - 1u)];   // the index addend, for now
        }
    } else {
#pragma unroll
        for (int it = 0; it < BK_ITERS; ++it) {
            const u32 hl = it * 64 + lane;
            const u32 inc = bk_scan_max((u32)s_owner[hl]);
            const u32 a = max(inc, carry);
            carry = max(carry, (u32)__builtin_amdgcn_readlane((int)inc, 63));
            slot[it] = (u32)(cs_kbase[k0 + a] >> L.sh_qpos) & pmask;
            word[it] = dk32[cs_base[k0 + a] + lo + min(hl, len - 1u)];
        }
    }
#pragma unroll
    for (int it = 0; it < BK_ITERS; ++it) {
        const u32 hl = it * 64 + lane;
        u32 r, wd;
        const bool ok = bk_hit(word[it], slot[it], L, r, wd) && hl < len;
        word[it] = wd;
        slot[it] = 0xFFFFFFFFu;
        if (ok) slot[it] = (cbase + r) | (atomicAdd(&s_hist[cbase + r], 1u) << 10);  // LDS fetch-add: a unique slot inside (tile, range, copy)
    }
    bk_wave_sync();
    if (!SCATTER) {
        for (u32 r = lane; r < L.R; r += 64) {   // tile-major: the tile's R counts are one contiguous run (zeros included)
            u32 n = 0;
            if (ncopy == 8) {
#pragma unroll
                for (u32 cpy = 0; cpy < 8; ++cpy) n += s_hist[cpy * cstride + r];
            } else {
                for (u32 cpy = 0; cpy < ncopy; ++cpy) n += s_hist[cpy * cstride + r];
            }
            mat[(size_t)t * L.R + r] = n;
        }
        return;
    }
    if (!STAGED) {
        for (u32 r = lane; r < L.R; r += 64) {  // counts -> global start of every (range, copy) share
            u32 run = mat[(size_t)t * L.R + r];
            for (u32 cpy = 0; cpy < ncopy; ++cpy) {
                const u32 n = s_hist[cpy * cstride + r];
                s_hist[cpy * cstride + r] = run;
                run += n;
            }
        }
        bk_wave_sync();
#pragma unroll
        for (int it = 0; it < BK_ITERS; ++it)
            if (slot[it] != 0xFFFFFFFFu) out[s_hist[slot[it] & 1023u] + (slot[it] >> 10)] = word[it];
        return;
    }
    // ---- staged scatter -------------------------------------------------------------------------------------------------
    // counts -> place of every (range, copy) share inside the tile (ranges ascending), and for every non-empty run its
    // global start minus its local start, listed in run order
    u32 total = 0, nrun = 0;
    u32 myhead[BK_STAGE_RMAX / 64];   // local start of this lane's range in round k (~0u: empty)
#pragma unroll
    for (u32 k = 0; k < BK_STAGE_RMAX / 64; ++k) {
        const u32 r = k * 64 + lane;
        myhead[k] = 0xFFFFFFFFu;
        if (k * 64 >= L.R) continue;   // wave-uniform
        const u32 rr = min(r, L.R - 1u);   // lanes past the last range read a valid counter and contribute nothing
        u32 cnt8[8];
        u32 n = 0;
        if (ncopy == 8) {   // wave-uniform; the usual case (up to 64 ranges): eight independent reads
#pragma unroll
            for (u32 cpy = 0; cpy < 8; ++cpy) cnt8[cpy] = s_hist[cpy * cstride + rr];
#pragma unroll
            for (u32 cpy = 0; cpy < 8; ++cpy) n += cnt8[cpy];
        } else {
            for (u32 cpy = 0; cpy < ncopy; ++cpy) n += s_hist[cpy * cstride + rr];
        }
        if (r >= L.R) n = 0;
        const u32 inc = bk_scan_add(n);
        const unsigned long long nz = __ballot(n != 0);
        u32 run = total + inc - n;
        if (n) {
            const u32 j = nrun + __builtin_amdgcn_mbcnt_hi((u32)(nz >> 32), __builtin_amdgcn_mbcnt_lo((u32)nz, 0u));
            s_hist[HL.gbase + j] = mat[(size_t)t * L.R + r] - run;
            myhead[k] = run;
        }
        if (r < L.R) {
            if (ncopy == 8) {
#pragma unroll
                for (u32 cpy = 0; cpy < 8; ++cpy) {
                    s_hist[cpy * cstride + r] = run;
                    run += cnt8[cpy];
                }
            } else {
                for (u32 cpy = 0; cpy < ncopy; ++cpy) {
                    const u32 c = s_hist[cpy * cstride + r];
                    s_hist[cpy * cstride + r] = run;
                    run += c;
                }
            }
        }
        total += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        nrun += (u32)__popcll(nz);
    }
    bk_wave_sync();   // (also: phase A's arrays are dead, the staging area may be written)
#pragma unroll
    for (int it = 0; it < BK_ITERS; ++it)
        if (slot[it] != 0xFFFFFFFFu) s_stage[s_hist[slot[it] & 1023u] + (slot[it] >> 10)] = word[it];
    bk_wave_sync();
    // words are < 2^31 (wb + bd + bp <= 31): bit 31 marks the first word of every run
#pragma unroll
    for (u32 k = 0; k < BK_STAGE_RMAX / 64; ++k)
        if (myhead[k] != 0xFFFFFFFFu) s_stage[myhead[k]] |= 0x80000000u;
    bk_wave_sync();
    u32 jc = 0;   // runs that started before this step
#pragma unroll
    for (int it = 0; it < BK_ITERS; ++it) {
        const u32 i = it * 64 + lane;
        if ((u32)it * 64u >= total) break;   // wave-uniform
        const u32 v = i < total ? s_stage[i] : 0u;
        const unsigned long long hb = __ballot((v >> 31) != 0);
        const u32 j = jc + __builtin_amdgcn_mbcnt_hi((u32)(hb >> 32), __builtin_amdgcn_mbcnt_lo((u32)hb, 0u)) + (v >> 31) - 1u;
        if (i < total) out[s_hist[HL.gbase + j] + i] = v & 0x7FFFFFFFu;   // (non-temporal stores: measured, no difference)
        jc += (u32)__popcll(hb);
    }
}

// ================================================================================================================
// the count pass WITHOUT the index entries: range boundaries per index bucket (round 6)
// ================================================================================================================
// k_bkt_pass<count> reads every visited index entry a second time (4 B per hit: 3.3 GB per 928 M-hit launch) only to learn how many of a
// tile's hits fall into each band range.  The band range of a hit is a function of its SUBJECT alone (a subject's bands lie inside one
// range: the host checks that for the few multi-band subjects and keeps the counting pass otherwise), so with the members of every index
// bucket in descending entry order (order_chunk: subject descending -- the reference's own CSR order) the entries of one range are one
// run of the bucket, and a table of the runs' boundaries, built once per (chunk, key layout, range width), turns the count into
// arithmetic: tile x seed x range -> overlap of two intervals.
//   rtab[u * (R + 1) + r] = entries of occupied bucket u whose range is >= r  (r = 0 ... R; the one entry per chunk the reference drops,
//   offset 0 of the first sequence, has the smallest value, sits last and is in no range): range r's run is [rtab[u][r + 1], rtab[u][r]).
__global__ __launch_bounds__(256) void k_rtab_build(const u32* __restrict__ dk32, u32 E, const u32* __restrict__ ubeg /*U + 1*/, u32 U, int gb, u32 R,
                                                    u16* __restrict__ rtab, u32* __restrict__ row_of_slot, u32* __restrict__ flag) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= E) return;
    u32 lo = 0, hi = U;   // the bucket of slot i: largest u with ubeg[u] <= i
    while (hi - lo > 1) {
        const u32 m = (lo + hi) >> 1;
        if (ubeg[m] <= i) lo = m;
        else hi = m;
    }
    const u32 u = lo, beg = ubeg[u], cnt = ubeg[u + 1] - beg, li = i - beg;
    if (li == 0) row_of_slot[i] = u;
    if (cnt > 0xFFFFu) {   // (16-bit boundaries: such a bucket is far above every frequency threshold; the host keeps the counting pass)
        if (li == 0) atomicOr(flag, 2u);
        return;
    }
    const u32 c = dk32[i];
    const int ri = (int)c >= 0 ? (int)min(c >> gb, R - 1u) : -1;
    int rp = (int)R;
    if (li) {
        const u32 cp = dk32[i - 1];
        rp = (int)cp >= 0 ? (int)min(cp >> gb, R - 1u) : -1;
    }
    if (ri > rp) atomicOr(flag, 1u);   // not in descending order
    u16* row = rtab + (size_t)u * (R + 1u);
    for (int r = ri + 1; r <= rp; ++r) row[r] = (u16)li;
    if (li == cnt - 1u)
        for (int r = 0; r <= ri; ++r) row[r] = (u16)cnt;
}

// mat[tile * R + range] = hits of the tile in the range, as k_bkt_pass<count> writes them.  A wave per tile.  The loads form a chain -- tile
// descriptor -> seed (hit offset, slot base) -> bucket ordinal -> its row of boundaries -- so the tile's seeds (a handful in a dense pass) are
// first read side by side, lane = seed, and then taken one after the other with lane = range: four memory round trips per tile, not four per
// seed (0.51 -> 0.2 ms per 906 k tiles).
__global__ __launch_bounds__(64 * BK_WAVES) void k_bkt_count_tab(const uint4* __restrict__ td, const u32* __restrict__ qseg, u32 NT, const u32* __restrict__ cs_hoff,
                                                                  const u32* __restrict__ cs_base, const u32* __restrict__ row_of_slot, const u16* __restrict__ rtab, u32 R,
                                                                  u32* __restrict__ mat) {
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 t = __builtin_amdgcn_readfirstlane(blockIdx.x * BK_WAVES + w);
    if (t >= NT) return;
    const uint4 d = td[t];
    const u32 q = __builtin_amdgcn_readfirstlane(d.x), first = __builtin_amdgcn_readfirstlane(d.y);
    const u32 k0 = __builtin_amdgcn_readfirstlane(d.z), k1 = __builtin_amdgcn_readfirstlane(d.w);
    const u32 last = first + min((u32)BK_HITS, __builtin_amdgcn_readfirstlane(qseg[q + 1]) - first);   // one past the tile's last ordinal
    for (u32 r0 = 0; r0 < R; r0 += 64) {   // (one round unless a chunk has more than 64 ranges)
        const u32 r = min(r0 + lane, R - 1u);
        u32 n = 0;
        for (u32 a0 = k0; a0 <= k1; a0 += 64) {
            // lane = seed: the part of its bucket that lies inside the tile, relative to the bucket's first slot
            const u32 a = min(a0 + lane, k1);
            const u32 h0 = cs_hoff[a];
            const u32 o_lo = max(first, h0), o_hi = a == k1 ? last : min(last, cs_hoff[a + 1]);
            const bool live = a0 + lane <= k1 && o_hi > o_lo;
            const u32 row = live ? row_of_slot[cs_base[a] + h0] : 0u;
            const u32 x0 = o_lo - h0, x1 = o_hi - h0;
            const u32 ns = min(64u, k1 - a0 + 1u);
            // lane = range: overlap of the range's run [row[r + 1], row[r]) with the seed's part
            // (four seeds' rows requested before the first is used; a seed outside the tile reads row 0 and counts nothing)
            for (u32 j = 0; j < ns; j += 4) {
                u32 lo[4], hi[4];
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    const int jj = (int)min(j + k, ns - 1u);
                    const bool lv = j + k < ns && __builtin_amdgcn_readlane((int)live, jj) != 0;
                    const u16* rw = rtab + (size_t)(u32)__builtin_amdgcn_readlane((int)row, jj) * (R + 1u);
                    lo[k] = max((u32)rw[r + 1], (u32)__builtin_amdgcn_readlane((int)x0, jj));
                    hi[k] = lv ? min((u32)rw[r], (u32)__builtin_amdgcn_readlane((int)x1, jj)) : 0u;
                }
#pragma unroll
                for (u32 k = 0; k < 4; ++k) n += hi[k] > lo[k] ? hi[k] - lo[k] : 0u;
            }
        }
        if (r0 + lane < R) mat[(size_t)t * R + r] = n;
    }
}

// The same with a wave per QUERY (R <= 64): the query's seeds are walked once, in order, the current tile's counts in the lanes; a tile is
// written when the seed that reaches its end has been added.  A wave per tile paid its chain of four dependent memory round trips for
// three or four seeds (0.42 ms per 906 k tiles: latency); here the chain is paid once per 64 seeds.
__global__ __launch_bounds__(64 * BK_WAVES) void k_bkt_count_tabq(const uint4* __restrict__ td, const u32* __restrict__ qseg, const u32* __restrict__ t0 /*nqp + 1*/, u32 nqp,
                                                                   const u32* __restrict__ cs_hoff, const u32* __restrict__ cs_base, const u32* __restrict__ row_of_slot,
                                                                   const u16* __restrict__ rtab, u32 R, u32* __restrict__ mat) {
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 q = __builtin_amdgcn_readfirstlane(blockIdx.x * BK_WAVES + w);
    if (q >= nqp) return;
    const u32 tf = __builtin_amdgcn_readfirstlane(t0[q]), te = __builtin_amdgcn_readfirstlane(t0[q + 1]);
    if (te <= tf) return;   // no hits: no tile
    const u32 qe = __builtin_amdgcn_readfirstlane(qseg[q + 1]);
    const u32 ka = __builtin_amdgcn_readfirstlane(td[tf].z), kb = __builtin_amdgcn_readfirstlane(td[te - 1u].w);
    const u32 r = min(lane, R - 1u);
    u32 acc = 0, tcur = tf;
    u32 tile_lo = __builtin_amdgcn_readfirstlane(qseg[q]), tile_hi = min(tile_lo + (u32)BK_HITS, qe);
    for (u32 a0 = ka; a0 <= kb; a0 += 64) {
        const u32 a = min(a0 + lane, kb);
        const u32 h0 = cs_hoff[a], h1 = a < kb ? cs_hoff[a + 1] : qe;   // the seed's hit ordinals
        const u32 row = row_of_slot[cs_base[a] + h0];
        const u32 ns = min(64u, kb - a0 + 1u);
        for (u32 j0 = 0; j0 < ns; j0 += 4) {
            u32 lo[4], hi[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {   // four seeds' rows requested before the first is used
                const u16* rw = rtab + (size_t)(u32)__builtin_amdgcn_readlane((int)row, (int)min(j0 + k, ns - 1u)) * (R + 1u);
                lo[k] = rw[r + 1], hi[k] = rw[r];
            }
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                if (j0 + k >= ns) break;   // (wave-uniform)
                const u32 s0 = (u32)__builtin_amdgcn_readlane((int)h0, (int)(j0 + k)), s1 = (u32)__builtin_amdgcn_readlane((int)h1, (int)(j0 + k));
                for (;;) {   // the tiles the seed's ordinals [s0, s1) reach into
                    const u32 x0 = max(lo[k], max(s0, tile_lo) - s0), x1 = min(hi[k], min(s1, tile_hi) - s0);
                    acc += x1 > x0 ? x1 - x0 : 0u;
                    if (s1 < tile_hi) break;
                    if (lane < R) mat[(size_t)tcur * R + r] = acc;   // the tile is complete
                    acc = 0, ++tcur, tile_lo = tile_hi, tile_hi = min(tile_hi + (u32)BK_HITS, qe);
                    if (s1 <= tile_lo) break;
                }
            }
        }
    }
}

// SOHIT_COUNT_TAB=2 (tests): the table's counts against the counting pass's
__global__ __launch_bounds__(256) void k_u32_differ(const u32* __restrict__ a, const u32* __restrict__ b, size_t n, u32* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(flag, 1u);
}
void launch_u32_differ(const u32* a, const u32* b, size_t n, u32* flag, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_u32_differ, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n, flag);
}
void launch_rtab_build(const u32* dk32, u32 E, const u32* ubeg, u32 U, int gb, u32 R, u16* rtab, u32* row_of_slot, u32* flag, hipStream_t st) {
    if (E && U) hipLaunchKernelGGL(k_rtab_build, dim3((E + 255) / 256), dim3(256), 0, st, dk32, E, ubeg, U, gb, R, rtab, row_of_slot, flag);
}
void launch_bkt_count_tab(const void* td, const u32* qseg, const u32* t0, u32 nqp, u32 NT, const u32* cs_hoff, const u32* cs_base, const u32* row_of_slot, const u16* rtab,
                          u32 R, u32* mat, hipStream_t st) {
    if (NT && R <= 64) {
        hipLaunchKernelGGL(k_bkt_count_tabq, dim3((nqp + BK_WAVES - 1) / BK_WAVES), dim3(64 * BK_WAVES), 0, st, (const uint4*)td, qseg, t0, nqp, cs_hoff, cs_base, row_of_slot,
                           rtab, R, mat);
        return;
    }
    if (NT) hipLaunchKernelGGL(k_bkt_count_tab, dim3((NT + BK_WAVES - 1) / BK_WAVES), dim3(64 * BK_WAVES), 0, st, (const uint4*)td, qseg, NT, cs_hoff, cs_base, row_of_slot,
                               rtab, R, mat);
}

// ================================================================================================================
// grouping kernel: hits of a bucket -> 64-bit keys in (subject, diagonal, qpos) order
// ================================================================================================================
// One workgroup per bucket (1-2 k hits over a few hundred subjects).  An exact sort, but shaped for the data: an LDS
// counting sort by the subject inside the range (one fetch-add per hit), then every hit finds its rank inside its subject's
// segment by comparing with the segment's other hits.  Nearly all segments hold a handful of hits (a random seed match or
// two); the few long ones (the query itself, a homolog: hundreds of hits on one or two diagonals) are ranked by whole waves,
// every lane reading the same LDS word (broadcast).  Hit words of one query are distinct (an index entry meets a query window
// once), so ranks are unique.  A bucket above BG_CAP hits is done in sub-passes over sub-ranges of its subjects.
#define BG_THREADS 512
#ifndef BG_CAP
#define BG_CAP 4096       // hits sorted at a time (8 per thread, kept in registers between the phases)
#endif
#ifndef BG_BINS
#define BG_BINS 1024      // subjects per range (wb <= 10)
#endif
#define BG_SMALL 16       // segments up to this size are ranked by their own hits' threads
#define BG_NBIG 256       // longer segments per bucket handled by whole waves (more: the slow way, still exact)
#define BG_SUB (BG_CAP * 2 / 3)   // sub-passes of an oversized bucket are sized for this many hits on average
#define BG_NONE 0xFFFFFFFFu
static_assert(BG_CAP <= 4096 && BG_BINS % BG_THREADS == 0 && BG_CAP % BG_THREADS == 0, "s_big packs start in 12 bits and size in 13");

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt: every barrier would then wait for the key stores
// of the previous phase and for the next bucket's prefetched hits.  All data the phases exchange lives in LDS; global loads are
// consumed by the thread that issued them (the compiler counts those).
__device__ __forceinline__ void bg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bucket b (range-major) occupies hits[bext[b] .. bext[b + 1])
__global__ __launch_bounds__(256) void k_bkt_extents(const u32* __restrict__ mat, const u32* __restrict__ t0, u32 NT, u32 R, u32 nqp, u32 nb,
                                                     const u32* __restrict__ total, u32* __restrict__ bext /*nb + 1*/) {
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b > nb) return;
    u32 r = b / nqp;
    const u32 qrel = b - r * nqp;
    // the first tile of a query without hits is the next query's first tile: its scanned entry is where the bucket would start;
    // past the last tile that is the first entry of the next range (scan order: position r * NT + t), or the total
    u32 t = b == nb ? NT : t0[qrel];
    if (t >= NT) t = 0, r += 1;
    bext[b] = r >= R ? *total : mat[(size_t)t * R + r];
}

// ---- exclusive scan of the tile-major count matrix in RANGE-MAJOR order ------------------------------------------------
// position of entry (tile t, range r) in the scan order = r * NT + t.  Blocks of BK_TB consecutive tiles: column sums per block
// (k_bkt_colsum, stored range-major), one short scan over those R * NTB partials (scan_u32), then every block rewrites its
// entries as running sums from its column bases (k_bkt_colscan).  Thread = one range: a wave touches R contiguous words per tile.
#define BK_TB 128
__global__ __launch_bounds__(64) void k_bkt_colsum(const u32* __restrict__ mat, u32 NT, u32 R, u32 NTB, u32* __restrict__ partT /*[R][NTB]*/) {
    const u32 blk = blockIdx.x, t0 = blk * BK_TB, t1 = min(NT, t0 + BK_TB);
    for (u32 r = threadIdx.x; r < R; r += 64) {
        u32 s = 0;
#pragma unroll 8
        for (u32 t = t0; t < t1; ++t) s += mat[(size_t)t * R + r];
        partT[(size_t)r * NTB + blk] = s;
    }
}
__global__ __launch_bounds__(64) void k_bkt_colscan(u32* __restrict__ mat, u32 NT, u32 R, u32 NTB, const u32* __restrict__ baseT /*[R][NTB], scanned*/) {
    const u32 blk = blockIdx.x, t0 = blk * BK_TB, t1 = min(NT, t0 + BK_TB);
    for (u32 r = threadIdx.x; r < R; r += 64) {
        u32 run = baseT[(size_t)r * NTB + blk];
        u32 t = t0;
        for (; t + 8 <= t1; t += 8) {   // eight loads in flight, stores in order
            u32 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = mat[(size_t)(t + k) * R + r];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mat[(size_t)(t + k) * R + r] = run;
                run += v[k];
            }
        }
        for (; t < t1; ++t) {
            const u32 v = mat[(size_t)t * R + r];
            mat[(size_t)t * R + r] = run;
            run += v;
        }
    }
}

// (8 waves per SIMD at the price of 7 spilled dwords: budgets of 7 / 6 waves measured again with the 32-bit output -- 91.7 / 89.8 against
// 80.5 ms of grouping per step on the 100 k weight-6 set)
template <bool W32 /*output: the sorted words themselves (bit 31 = first word of the bucket) instead of 64-bit keys*/>
__global__ __launch_bounds__(BG_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bkt_group(const u32* __restrict__ hits, const u32* __restrict__ bext /*nb + 1*/, u32 nb,
                                                            BktLayout L, KeyLayout kl, u64* __restrict__ keys, u32* __restrict__ words32, u32* __restrict__ fallback) {
    __shared__ u32 s_srt[BG_CAP];          // the hits, grouped by subject
    __shared__ u32 s_bin[BG_BINS + 1];     // per subject: count -> scatter cursor (= end of its segment afterwards)
    __shared__ u32 s_big[BG_NBIG];         // work units of long segments: start | size << 12 | 64-member block << 25
    __shared__ u32 s_wsum[BG_THREADS / 64];
    __shared__ u32 s_ctl[4];               // [0] hits of the sub-pass, [1] long segments, [2] refused
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32 dmask = (1u << L.bd) - 1u, pmask = (1u << L.bp) - 1u;
    const int sshift = L.bd + L.bp;
    const u32 W = 1u << L.wb;
    constexpr int HPT = BG_CAP / BG_THREADS;
    constexpr int BPT = BG_BINS / BG_THREADS;
    // software pipeline over the persistent loop: extents are fetched two buckets ahead, hits one bucket ahead
    const u32 G = gridDim.x;
    u32 e0 = 0, e1 = 0, f0 = 0, f1 = 0, nhit[HPT];
    auto fetch_extent = [&](u32 b, u32& lo, u32& hi) {
        lo = hi = 0;
        if (b < nb) lo = bext[b], hi = bext[b + 1];
    };
    auto fetch_hits = [&](u32 lo, u32 hi) {
        const u32 cnt = hi - lo;
#pragma unroll
        for (int k = 0; k < HPT; ++k) {
            const u32 j = (u32)tid + (u32)k * BG_THREADS;
            nhit[k] = (cnt <= BG_CAP && j < cnt) ? hits[lo + j] : BG_NONE;
        }
    };
    fetch_extent(blockIdx.x, e0, e1);
    fetch_hits(e0, e1);
    fetch_extent(blockIdx.x + G, f0, f1);
    for (u32 b = blockIdx.x; b < nb; b += G) {
        const u32 b0 = e0, n = e1 - e0;
        u32 my[HPT];
#pragma unroll
        for (int k = 0; k < HPT; ++k) my[k] = nhit[k];
        e0 = f0, e1 = f1;
        fetch_hits(e0, e1);                 // bucket b + G (its extent arrived during the previous bucket)
        fetch_extent(b + 2 * G, f0, f1);    // bucket b + 2G
        if (n == 0) continue;
        const u32 r = b / L.nqp, qrel = b - r * L.nqp;
        const u64 kq = ((u64)(L.qa + qrel) << kl.sh_q) | ((u64)(r << L.wb) << kl.sh_subj);  // the bucket's share of every key
        u32 K = 1;
        while ((u64)K * BG_SUB < n && n > BG_CAP && K < W) K <<= 1;
        const u32 wsub = W / K;  // subjects per sub-pass
        u32 cursor = b0;
        for (u32 sp = 0; sp < K; ++sp) {
            const u32 slo = sp * wsub;
            // ---- A: clear the subject bins ----
            bg_barrier();  // the previous round's LDS is no longer read
#pragma unroll
            for (int i = 0; i < BPT; ++i) s_bin[i * BG_THREADS + tid] = 0;
            if (tid < 4) s_ctl[tid] = 0;
            bg_barrier();
            // ---- B: count per subject ----
            if (K == 1) {
#pragma unroll
                for (int k = 0; k < HPT; ++k)
                    if (my[k] != BG_NONE) atomicAdd(&s_bin[my[k] >> sshift], 1u);
            } else {
                u32 mine = 0;
                for (u32 j = (u32)tid; j < n; j += BG_THREADS) {
                    const u32 sj = (hits[b0 + j] >> sshift) - slo;
                    if (sj < wsub) atomicAdd(&s_bin[sj], 1u), ++mine;
                }
                if (mine) atomicAdd(&s_ctl[0], mine);
            }
            bg_barrier();
            const u32 npass = K == 1 ? n : s_ctl[0];
            if (npass > BG_CAP) {  // one subject sub-range still holds too many hits: the host reruns the pass on the sorted path
                if (tid == 0) atomicOr(fallback, 1u);
                break;
            }
            // ---- C: exclusive prefix over the bins (2 per thread): segment starts ----
            {
                u32 c[BPT], tot = 0;
#pragma unroll
                for (int k = 0; k < BPT; ++k) {
                    c[k] = s_bin[tid * BPT + k];
                    tot += c[k];
                }
                u32 inc = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const u32 x = __shfl_up(inc, o);
                    if (lane >= o) inc += x;
                }
                if (lane == 63) s_wsum[w] = inc;
                bg_barrier();
                u32 run = inc - tot;
                for (int k = 0; k < w; ++k) run += s_wsum[k];
#pragma unroll
                for (int k = 0; k < BPT; ++k) {
                    s_bin[tid * BPT + k] = run;
                    if (c[k] > BG_SMALL) {  // a long segment: whole waves rank it, 64 members per work unit
                        const u32 nu = (c[k] + 63u) >> 6;
                        const u32 i = atomicAdd(&s_ctl[1], nu);
                        for (u32 u = 0; u < nu && i + u < BG_NBIG; ++u) s_big[i + u] = run | (c[k] << 12) | (u << 25);  // 12 + 13 + 7 bits
                    }
                    run += c[k];
                }
            }
            bg_barrier();
            // ---- D: scatter by subject (the bin becomes the cursor: afterwards it holds the END of its segment) ----
            if (K == 1) {
#pragma unroll
                for (int k = 0; k < HPT; ++k)
                    if (my[k] != BG_NONE) s_srt[atomicAdd(&s_bin[my[k] >> sshift], 1u)] = my[k];
            } else {
                for (u32 j = (u32)tid; j < n; j += BG_THREADS) {
                    const u32 hw = hits[b0 + j];
                    const u32 sj = (hw >> sshift) - slo;
                    if (sj < wsub) s_srt[atomicAdd(&s_bin[sj], 1u)] = hw;
                }
            }
            bg_barrier();
            // ---- E: rank inside the subject's segment = final position; short segments by their own hits' threads ----
            const u32 nbig = min(s_ctl[1], (u32)BG_NBIG);
            const bool big_all = s_ctl[1] > BG_NBIG;  // more long segments than the list holds: everything the slow way
            for (u32 p = (u32)tid; p < npass; p += BG_THREADS) {
                const u32 x = s_srt[p];
                const u32 sj = (x >> sshift) - slo;
                const u32 a = sj ? s_bin[sj - 1] : 0u, e = s_bin[sj];
                if (e - a > BG_SMALL && !big_all) continue;
                u32 rank = 0;
                for (u32 k = a; k < e; ++k) rank += (s_srt[k] < x) ? 1u : 0u;
                if (W32) words32[cursor + a + rank] = x | ((cursor + a + rank == b0) ? 0x80000000u : 0u);
                else keys[cursor + a + rank] = kq + ((u64)(x >> sshift) << kl.sh_subj) + ((u64)((x >> L.bp) & dmask) << kl.sh_diag) +
                                               ((u64)(x & pmask) << kl.sh_qpos);
            }
            if (!big_all) {
                for (u32 i = (u32)w; i < nbig; i += BG_THREADS / 64) {  // a wave per unit; every lane reads the same word (broadcast)
                    const u32 a = s_big[i] & 0xFFFu, sz = (s_big[i] >> 12) & 0x1FFFu, mi = (s_big[i] >> 25) * 64u + (u32)lane;
                    const u32 x = s_srt[a + min(mi, sz - 1u)];
                    u32 rank = 0;
                    for (u32 k = 0; k < sz; k += 4) {
                        u32 v[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = s_srt[a + min(k + (u32)t, sz - 1u)];
#pragma unroll
                        for (int t = 0; t < 4; ++t) rank += (k + (u32)t < sz && v[t] < x) ? 1u : 0u;
                    }
                    if (mi < sz) {
                        if (W32) words32[cursor + a + rank] = x | ((cursor + a + rank == b0) ? 0x80000000u : 0u);
                        else keys[cursor + a + rank] = kq + ((u64)(x >> sshift) << kl.sh_subj) + ((u64)((x >> L.bp) & dmask) << kl.sh_diag) +
                                                       ((u64)(x & pmask) << kl.sh_qpos);
                    }
                }
            }
            cursor += npass;
        }
    }
}

// ================================================================================================================
// best diagonal per (query, subject), bucket by bucket (fsearch.py:2709-2719)
// ================================================================================================================
// The pass records k_ungap appended (a few per cent of the hits) are binned into the SAME (query, subject range) buckets -- count,
// scan, scatter; the returning count atomic doubles as the record's rank inside its bucket, so the scatter needs none -- and the
// first-touch key of every record is computed on the way (what k_first_touch does on the sorted path).  A bucket holds ONE query
// and <= BG_BINS subjects, so "first group attaining the maximum per subject" is a reduction over an LDS table indexed by the
// subject inside the range: a 64-bit max over (score, inverted first-touch key) picks the best group -- ties go to the group visited
// first, the reference's strict `>` -- and a 64-bit min keeps the subject's earliest first-touch key (its place in the candidate
// order).  The reduction runs twice: once to count the candidates per bucket, and after a scan of the counts once more to write them
// densely (one global counter for all buckets serialises ~1e5 same-address atomics: 1.2 ms on config 2 against 0.1 ms for the
// second reduction).  This replaces the radix sort of all pass records by (query, subject) + segment flags + scan + k_best.
__device__ __forceinline__ u32 rec_bucket(u64 qs, int bs, const BktLayout& L) {
    return (((u32)qs & ((1u << bs) - 1u)) >> L.wb) * L.nqp + ((u32)(qs >> bs) - L.qa);
}

__global__ __launch_bounds__(256) void k_rec_count(const u64* __restrict__ p_qs, u32 n, int bs, BktLayout L, u32* __restrict__ bcnt,
                                                   u32* __restrict__ rnk) {
    // neighbouring records come from one wave of k_ungap, i.e. from one or two buckets: one atomic per distinct bucket of the wave
    // (a returning atomic per record serialises on those few addresses: 1.4 ms instead of 0.1 ms on config 2)
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const u64 qs = i < n ? p_qs[i] : UG_REC_NONE;
    const bool have = qs != UG_REC_NONE;   // (k_ungap1 leaves the unused slots of its reserved pieces marked)
    const u32 bk = have ? rec_bucket(qs, bs, L) : 0xFFFFFFFFu;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(have);
    u32 r = 0;
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const u32 b0 = __shfl(bk, leader);
        const unsigned long long m = __ballot(have && bk == b0);
        u32 base = 0;
        if (lane == leader) base = atomicAdd(&bcnt[b0], (u32)__popcll(m));
        base = __shfl(base, leader);
        if (have && bk == b0) r = base + (u32)__popcll(m & lt);
        todo &= ~m;
    }
    if (have) rnk[i] = r;
}

__global__ __launch_bounds__(256) void k_rec_scatter(const u64* __restrict__ p_qs, const u64* __restrict__ p_sd, const u64* __restrict__ p_ft,
                                                     const u32* __restrict__ rnk, u32 n, KeyLayout kl, BktLayout L, int ft_bits_entry, int bsp,
                                                     const u32* __restrict__ roff, const u32* __restrict__ boff, u64* __restrict__ q_qs,
                                                     u64* __restrict__ q_sd, u64* __restrict__ q_ft) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u64 qs = p_qs[i];
    if (qs == UG_REC_NONE) return;
    const u32 o = boff[rec_bucket(qs, kl.bs, L)] + rnk[i];
    q_qs[o] = qs, q_sd[o] = p_sd[i], q_ft[o] = ft_key_of_head(p_ft[i], kl, ft_bits_entry, bsp, roff);
}

#define BB_THREADS 256
template <bool WRITE>
__global__ __launch_bounds__(BB_THREADS) void k_bkt_best(const u64* __restrict__ q_qs, const u64* __restrict__ q_sd, const u64* __restrict__ q_ft,
                                                        const u32* __restrict__ boff /*nb + 1*/, u32 nb, BktLayout L, int bs, u32 seq_lo,
                                                        u32* __restrict__ ccnt /*!WRITE: out; WRITE: scanned, in*/, u64* __restrict__ c_ft,
                                                        u32* __restrict__ c_q, u32* __restrict__ c_rec, int bsp, int idx_bits) {
    // idx_bits > 0 (WRITE): instead of (first-touch key, query) the kernel writes ONE sort word per candidate into c_ft:
    // first-touch word << idx_bits | position inside the query's segment -- the candidate order then is a keys-only segmented sort
    // on the word's upper bits, and the row gather reads the position out of the sorted word (no index array, no key-build pass)
    __shared__ unsigned long long s_best[BG_BINS], s_min[BG_BINS];
    __shared__ u32 s_pref[BG_BINS];
    __shared__ u32 s_wsum[BB_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32 W = 1u << L.wb;
    const u64 FTM = (1ull << 44) - 1ull;  // first-touch keys fit 44 bits (host check), scores 20
    constexpr int SPT = BG_BINS / BB_THREADS;
    for (u32 b = blockIdx.x; b < nb; b += gridDim.x) {
        const u32 b0 = boff[b], n = boff[b + 1] - b0;
        // candidates are laid out QUERY-major (the buckets are range-major): a query's candidates of all ranges sit together, so
        // the gather that follows the candidate-order sort (k_emit_cands) stays inside a few KB per query instead of fetching one
        // 128-byte line per 16-byte record
        const u32 qrel = b - (b / L.nqp) * L.nqp;
        const u32 ci = qrel * L.R + b / L.nqp;
        if (n == 0) {
            if (!WRITE && tid == 0) ccnt[ci] = 0;
            continue;
        }
        __syncthreads();
        for (u32 i = (u32)tid; i < W; i += BB_THREADS) s_best[i] = 0ull, s_min[i] = ~0ull;
        __syncthreads();
        for (u32 i = (u32)tid; i < n; i += BB_THREADS) {
            const u32 s = (u32)q_qs[b0 + i] & (W - 1u);
            const u64 ft = q_ft[b0 + i];
            atomicMax(&s_best[s], ((q_sd[b0 + i] >> 32) << 44) | (FTM - ft));
            if (WRITE) atomicMin(&s_min[s], (unsigned long long)ft);
        }
        __syncthreads();
        // subjects with a candidate -> dense positions (subject order inside the bucket)
        u32 c[SPT], tot = 0;
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const u32 i = (u32)tid * SPT + (u32)k;
            c[k] = (i < W && s_best[i] != 0ull) ? 1u : 0u;
            tot += c[k];
        }
        u32 inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
        }
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        u32 run = inc - tot, total = 0;
        for (int k = 0; k < BB_THREADS / 64; ++k) {
            if (k < w) run += s_wsum[k];
            total += s_wsum[k];
        }
        if (!WRITE) {
            if (tid == 0) ccnt[ci] = total;
            continue;
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const u32 i = (u32)tid * SPT + (u32)k;
            if (i < W) s_pref[i] = run;
            run += c[k];
        }
        __syncthreads();
        const u32 base = ccnt[ci];
        const u32 seg0 = idx_bits ? ccnt[qrel * L.R] : 0u;   // first candidate of the query
        const u64 pmk = (1ull << bsp) - 1ull;
        const u32 gq = L.qa + qrel;
        for (u32 i = (u32)tid; i < n; i += BB_THREADS) {
            const u64 qs = q_qs[b0 + i], sd = q_sd[b0 + i], ft = q_ft[b0 + i];
            const u32 s = (u32)qs & (W - 1u);
            if ((((sd >> 32) << 44) | (FTM - ft)) != s_best[s]) continue;  // not the subject's best group
            const u32 o = base + s_pref[s];
            const int bdist = (int)(u32)sd;
            u32 qi, qj;
            if (bdist > 0) qi = 0, qj = (u32)bdist;
            else qi = (u32)(-bdist), qj = 0;
            if (idx_bits) {
                const u64 ft = s_min[s];
                c_ft[o] = ((((ft >> bsp) << 1) | ((ft & pmk) == pmk ? 1ull : 0ull)) << idx_bits) | (u64)(o - seg0);
            } else {
                c_ft[o] = s_min[s];
                c_q[o] = gq;
            }
            *reinterpret_cast<uint4*>(c_rec + 4 * (size_t)o) = make_uint4(((u32)qs & ((1u << bs) - 1u)) + seq_lo, (u32)(sd >> 32), qi, qj);
        }
    }
}

// ---- launch wrappers -------------------------------------------------------------------------------
u32 bkt_tile_hits() { return BK_HITS; }

void launch_bkt_ntiles(const u32* qseg, u32 nqp, u32* ntile, hipStream_t st) {
    hipLaunchKernelGGL(k_bkt_ntiles, dim3((nqp + 1 + 255) / 256), dim3(256), 0, st, qseg, nqp, ntile);
}

void launch_bkt_tiledesc(const u32* qseg, const u32* t0, u32 nqp, u32 NT, const u32* cs_hoff, u32 K, void* td, hipStream_t st) {
    if (!NT) return;
    hipLaunchKernelGGL(k_bkt_tiledesc, dim3((NT + 255) / 256), dim3(256), 0, st, qseg, t0, nqp, NT, cs_hoff, K, (uint4*)td);
}

void launch_bkt_pass(bool scatter, const void* td, const u32* qseg, u32 NT, const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase,
                     const u32* dk32, const u32* roff, const BktLayout& L, u32* mat, u32* out, hipStream_t st) {
    if (!NT) return;
    const dim3 g(((NT + BK_WAVES - 1) / BK_WAVES + 7u) & ~7u), bl(64 * BK_WAVES);  // multiple of 8: the XCD-aware tile order
    // the scatter is staged through LDS whenever the tile's runs fit the stage (the direct scatter serves passes with more ranges)
    const bool staged = scatter && L.R <= BK_STAGE_RMAX;
    const BkHist HL = bk_hist_layout(L.R, staged, true);
    // (the count pass at a budget of 8 waves per SIMD cannot be met either -- LDS, not registers, bounds it at 7: 0.752 against 0.749 ms)
    if (!scatter) hipLaunchKernelGGL((k_bkt_pass<false, false>), g, bl, 0, st, (const uint4*)td, qseg, NT, cs_hoff, cs_base, cs_kbase, dk32, L, HL, mat, out);
    // waves per SIMD the register budget is cut for.  Round 3: 5 (93 VGPRs; 6 spilled four dwords and was 4-5 % slower).  Round 4: the
    // banded hit word needs fewer registers -- 6 fits in 80 VGPRs without a spill and is 6 % faster (1.89 -> 1.77 ms per 928 M-hit
    // launch); a budget of 7 cannot be met (the compiler falls back to 5 waves: 1.88 ms)
    else if (staged) hipLaunchKernelGGL((k_bkt_pass<true, true, 6>), g, bl, 0, st, (const uint4*)td, qseg, NT, cs_hoff, cs_base, cs_kbase, dk32, L, HL, mat, out);
    else hipLaunchKernelGGL((k_bkt_pass<true, false>), g, bl, 0, st, (const uint4*)td, qseg, NT, cs_hoff, cs_base, cs_kbase, dk32, L, HL, mat, out);
}

void launch_bkt_extents(const u32* mat, const u32* t0, u32 NT, u32 R, u32 nqp, u32 nb, const u32* total, u32* bext, hipStream_t st) {
    hipLaunchKernelGGL(k_bkt_extents, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, mat, t0, NT, R, nqp, nb, total, bext);
}

u32 bkt_scan_blocks(u32 NT) { return (NT + BK_TB - 1) / BK_TB; }
void launch_bkt_colsum(const u32* mat, u32 NT, u32 R, u32* partT, hipStream_t st) {
    if (NT) hipLaunchKernelGGL(k_bkt_colsum, dim3(bkt_scan_blocks(NT)), dim3(64), 0, st, mat, NT, R, bkt_scan_blocks(NT), partT);
}
void launch_bkt_colscan(u32* mat, u32 NT, u32 R, const u32* baseT, hipStream_t st) {
    if (NT) hipLaunchKernelGGL(k_bkt_colscan, dim3(bkt_scan_blocks(NT)), dim3(64), 0, st, mat, NT, R, bkt_scan_blocks(NT), baseT);
}

int bkt_max_wb() {
    int lg = 0;
    while ((1 << lg) < BG_BINS) ++lg;
    return lg;
}

void launch_bkt_group(const u32* hits, const u32* bext, u32 nb, const BktLayout& L, const KeyLayout& kl, u64* keys, u32* words32, u32* fallback,
                      hipStream_t st) {
    if (!nb) return;
    // persistent workgroups striding over the buckets, range-major: the chip writes one subject range at a time
    const u32 grid = std::min<u32>(nb, 256u * 4u);
    if (words32) hipLaunchKernelGGL(k_bkt_group<true>, dim3(grid), dim3(BG_THREADS), 0, st, hits, bext, nb, L, kl, keys, words32, fallback);
    else hipLaunchKernelGGL(k_bkt_group<false>, dim3(grid), dim3(BG_THREADS), 0, st, hits, bext, nb, L, kl, keys, words32, fallback);
}

void launch_rec_count(const u64* p_qs, u32 n, int bs, const BktLayout& L, u32* bcnt, u32* rnk, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_rec_count, dim3((n + 255) / 256), dim3(256), 0, st, p_qs, n, bs, L, bcnt, rnk);
}

void launch_rec_scatter(const u64* p_qs, const u64* p_sd, const u64* p_ft, const u32* rnk, u32 n, const KeyLayout& kl, const BktLayout& L,
                        int ft_bits_entry, int bsp, const u32* roff, const u32* boff, u64* q_qs, u64* q_sd, u64* q_ft, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_rec_scatter, dim3((n + 255) / 256), dim3(256), 0, st, p_qs, p_sd, p_ft, rnk, n, kl, L, ft_bits_entry, bsp, roff, boff,
                       q_qs, q_sd, q_ft);
}

void launch_bkt_best(bool write, const u64* q_qs, const u64* q_sd, const u64* q_ft, const u32* boff, u32 nb, const BktLayout& L, int bs,
                     u32 seq_lo, u32* ccnt, u64* c_ft, u32* c_q, u32* c_rec, int bsp, int idx_bits, hipStream_t st) {
    if (!nb) return;
    const u32 grid = std::min<u32>(nb, 256u * 8u);
    if (write) hipLaunchKernelGGL(k_bkt_best<true>, dim3(grid), dim3(BB_THREADS), 0, st, q_qs, q_sd, q_ft, boff, nb, L, bs, seq_lo, ccnt, c_ft, c_q, c_rec, bsp, idx_bits);
    else hipLaunchKernelGGL(k_bkt_best<false>, dim3(grid), dim3(BB_THREADS), 0, st, q_qs, q_sd, q_ft, boff, nb, L, bs, seq_lo, ccnt, c_ft, c_q, c_rec, bsp, 0);
}

// rows of a bucketed pass in candidate order: sorted[i] = first-touch word << idx_bits | position inside the query's segment
// (k_bkt_best); a workgroup per query copies its records in that order and leaves the query's count
__global__ __launch_bounds__(256) void k_emit_cands_seg(const u64* __restrict__ sorted, const u32* __restrict__ seg /*nqp + 1*/, u32 qa, int idx_bits,
                                                        const u32* __restrict__ c_rec, u32* __restrict__ out_q, u32* __restrict__ out_rec,
                                                        u32* __restrict__ qcnt) {
    const u32 a = seg[blockIdx.x], n = seg[blockIdx.x + 1] - a;
    if (!n) return;
    const u32 q = qa + blockIdx.x;
    const u64 mask = (1ull << idx_bits) - 1ull;
    for (u32 i = threadIdx.x; i < n; i += 256) {
        const u32 r = a + (u32)(sorted[a + i] & mask);
        out_q[a + i] = q;
        *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)(a + i)) = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)r);
    }
    if (threadIdx.x == 0) qcnt[q] = n;
}

// ---- candidate order of a bucketed pass, hand-written (round 5): sort + row gather in ONE kernel --------------------------------------
// A workgroup per query.  A sort word is first-touch word << idx_bits | position inside the query's segment (unique, so an exact sort
// gives the stable library sort's order), and the top of the first-touch word is the QUERY POSITION of the candidate's first hit: the
// n ~ 8000 candidates of a query spread over its few hundred positions, a few dozen each.  So: an LDS counting sort by the word's top
// CO_DBITS bits (count, scan, scatter: the words land grouped by that digit), then every word finds its rank inside its digit's group
// by comparing with the group's other words (neighbouring threads read the same LDS words: broadcasts) -- digit group start + rank is
// the candidate's final place, and its record is copied there at once.  A query with more than CO_CAP candidates in one chunk is done
// in sub-passes over ranges of the digit; a single digit group above CO_CAP words (never seen: the frequency cap bounds the hits of one
// query position) raises `fallback` and the host orders the pass with the library sort.
// Before: segmented library radix sort 1.76 ms + gather kernel 1.32 ms per 928 M-hit pass; this kernel: see DESIGN.md.
#define CO_THREADS 1024
#define CO_CAP 16384
#define CO_DBITS 11
__global__ __launch_bounds__(CO_THREADS) void k_cand_order_seg(const u64* __restrict__ words, const u32* __restrict__ seg /*nqp + 1*/, u32 qa, int idx_bits,
                                                              int word_bits /*idx_bits + width of the first-touch word*/, const u32* __restrict__ c_rec,
                                                              u32* __restrict__ out_q, u32* __restrict__ out_rec, u32* __restrict__ qcnt,
                                                              u32* __restrict__ fallback) {
    __shared__ u64 s_key[CO_CAP];
    __shared__ u32 s_cnt[(1 << CO_DBITS) + 1];
    __shared__ u32 s_ws[CO_THREADS / 64];
    __shared__ u32 s_ctl[2];
    const u32 a = seg[blockIdx.x], n = seg[blockIdx.x + 1] - a;
    if (!n) return;
    const u32 q = qa + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dsh = max(word_bits - CO_DBITS, idx_bits);   // the digit: bits [dsh, word_bits) (fewer than CO_DBITS when the word is short)
    const u32 ND = 1u << min(CO_DBITS, word_bits - dsh);
    const u64 mask = (1ull << idx_bits) - 1ull;
    constexpr int PER = (1 << CO_DBITS) / CO_THREADS;
    u32 d_lo = 0, placed = 0;   // sub-pass: digits [d_lo, d_hi); candidates placed by earlier sub-passes
    while (d_lo < ND) {
        // ---- count per digit (>= d_lo) ----
        for (u32 i = (u32)tid; i <= ND; i += CO_THREADS) s_cnt[i] = 0;
        __syncthreads();
        for (u32 i = (u32)tid; i < n; i += CO_THREADS) {
            const u32 d = (u32)(words[a + i] >> dsh) & (ND - 1u);
            if (d >= d_lo) atomicAdd(&s_cnt[d], 1u);
        }
        __syncthreads();
        // ---- exclusive scan; d_hi = first digit whose group would end beyond CO_CAP ----
        {
            u32 c[PER], tot = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) c[k] = (u32)(tid * PER + k) < ND ? s_cnt[tid * PER + k] : 0u, tot += c[k];
            u32 inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 x = __shfl_up(inc, o);
                if (lane >= o) inc += x;
            }
            if (lane == 63) s_ws[w] = inc;
            if (tid == 0) s_ctl[0] = ND, s_ctl[1] = 0;
            __syncthreads();
            u32 run = inc - tot;
            for (int k = 0; k < w; ++k) run += s_ws[k];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const u32 d = (u32)(tid * PER + k);
                if (d < ND) {
                    s_cnt[d] = run;
                    if (run + c[k] > CO_CAP && run <= CO_CAP && d >= d_lo) atomicMin(&s_ctl[0], d);   // the first group that does not fit
                }
                run += c[k];
            }
            if (tid == CO_THREADS - 1) s_cnt[ND] = run;
        }
        __syncthreads();
        const u32 d_hi = s_ctl[0];
        if (d_hi == d_lo) {   // one digit group alone exceeds the LDS sort: the host reorders the pass with the library sort
            if (tid == 0) atomicOr(fallback, 2u);
            return;
        }
        const u32 m = s_cnt[d_hi];   // candidates of this sub-pass (group starts are relative to it)
        __syncthreads();
        // ---- scatter by digit: the counter becomes the cursor (= the END of its group afterwards) ----
        for (u32 i = (u32)tid; i < n; i += CO_THREADS) {
            const u64 x = words[a + i];
            const u32 d = (u32)(x >> dsh) & (ND - 1u);
            if (d >= d_lo && d < d_hi) s_key[atomicAdd(&s_cnt[d], 1u)] = x;
        }
        __syncthreads();
        // ---- rank inside the digit's group = final place; copy the record ----
        for (u32 p = (u32)tid; p < m; p += CO_THREADS) {
            const u64 x = s_key[p];
            const u32 d = (u32)(x >> dsh) & (ND - 1u);
            const u32 ga = d > d_lo ? s_cnt[d - 1] : 0u, ge = s_cnt[d];
            u32 rank = 0;
            for (u32 k = ga; k < ge; ++k) rank += s_key[k] < x ? 1u : 0u;
            const u32 o = a + placed + ga + rank;
            out_q[o] = q;
            *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)o) = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)(a + (u32)(x & mask)));
        }
        __syncthreads();
        placed += m;
        d_lo = d_hi;
    }
    if (tid == 0) qcnt[q] = n;
}
void launch_cand_order_seg(const u64* words, const u32* seg, u32 nqp, u32 qa, int idx_bits, int word_bits, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                           u32* fallback, hipStream_t st) {
    if (nqp) hipLaunchKernelGGL(k_cand_order_seg, dim3(nqp), dim3(CO_THREADS), 0, st, words, seg, qa, idx_bits, word_bits, c_rec, out_q, out_rec, qcnt, fallback);
}

void launch_emit_cands_seg(const u64* sorted, const u32* seg, u32 nqp, u32 qa, int idx_bits, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                           hipStream_t st) {
    if (nqp) hipLaunchKernelGGL(k_emit_cands_seg, dim3(nqp), dim3(256), 0, st, sorted, seg, qa, idx_bits, c_rec, out_q, out_rec, qcnt);
}
