// k_seed.hip -- query side of the seed stage (Fasta.find_msav_m up to the hit walk,
// fsearch.py:2645-2691): window hashing, bucket-bounds fetch, k-mer self-score order,
// high-frequency cap, and the load-balanced seed-lookup kernel that turns every visited index
// entry into one 64-bit sort key (query | subject | diagonal | qpos | as | tag).
#include "common.h"
#include "kernels.h"
#include "refsort.h"
#include "seedhash.h"

// ---- query window hashes (chunk independent) -----------------------------------------------------
// qbucket[as * Ppad + p] = bucket of alphabet a / pattern s at packed position p, ~0u when the
// window is invalid or a duplicate (bucket, position) within the alphabet.  Queries: step 1 (2658).
__global__ __launch_bounds__(TILE_POS) void k_qhash(const u32* __restrict__ words, u32 Ppad, SeedCfg cfg, HashLut lut,
                                                    u32* __restrict__ qbucket) {
    __shared__ u8 s_cls[TILE_POS + MAX_SEEDLEN];
    const u32 p0 = blockIdx.x * TILE_POS;
    stage_classes(words, p0, Ppad, s_cls);
    __syncthreads();
    const u32 p = p0 + threadIdx.x;
    if (p >= Ppad) return;
    u32 bucket[MAX_PATTERNS];
    const bool dead = s_cls[threadIdx.x] >= HCLS_SEP;
    for (int a = 0; a < cfg.A; ++a) {
        u32 mask = dead ? 0u : hash_position(s_cls + threadIdx.x, cfg, lut.v[a], bucket);
        for (int s = 0; s < cfg.S; ++s) qbucket[(size_t)(a * cfg.S + s) * Ppad + p] = ((mask >> s) & 1u) ? bucket[s] : 0xFFFFFFFFu;
    }
}

// ---- bucket bounds against one chunk index (get_bin_mem, fsearch.py:2530-2541) -------------------
// sbeg/scnt[as][p]: first slot and clamped size of the bucket; pcnt[p] = sum over as (hist[qst][2]).
// The bucket directory is the chunk's bitmap + rank table (DIR; k_index.hip: one 8-byte read, then the bucket's two slot bounds) or,
// for very large -M, its open-addressed map (one probe sequence per window).
template <bool DIR>
__global__ __launch_bounds__(256) void k_bounds(const u32* __restrict__ qbucket, u32 Ppad, int AS, const u32* __restrict__ hkey,
                                                const u64* __restrict__ hval, int hshift, u32 hmask, const uint2* __restrict__ dir,
                                                const u32* __restrict__ ubeg, u32 NC, u32 E, u32* __restrict__ sbeg,
                                                u32* __restrict__ scnt, u32* __restrict__ pcnt) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= Ppad) return;
    const i64 L = (i64)E - 1;  // self.L = len(self.locus) - 1
    u32 tot = 0;
    for (int as = 0; as < AS; ++as) {
        const size_t t = (size_t)p * AS + as;  // seed ordinal: position-major, so hits are generated in (query, qpos, as) order
        const u32 b = qbucket[(size_t)as * Ppad + p];
        u32 beg = 0, cnt = 0;
        if (b != 0xFFFFFFFFu && b + 1u < NC) {  // bucket NC-1: `except: st = ed` (always empty)
            if (DIR) {
                const uint2 d = dir[b >> 5];
                const u32 bit = b & 31u;
                if ((d.x >> bit) & 1u) {
                    const u32 k = d.y + (u32)__popc(d.x & ((1u << bit) - 1u));
                    const i64 st = (i64)ubeg[k];
                    i64 ed = (i64)ubeg[k + 1];  // == start[b + 1]
                    ed = ed < L ? ed : L;
                    if (ed > st) beg = (u32)st, cnt = (u32)(ed - st);
                }
            } else {
                u32 h = (b * 2654435761u) >> hshift;
                for (;;) {
                    const u32 k = hkey[h];
                    if (k == b) {
                        const u64 v = hval[h];
                        const i64 st = (i64)(u32)v;
                        i64 ed = st + (i64)(u32)(v >> 32);  // == start[b + 1]
                        ed = ed < L ? ed : L;
                        if (ed > st) beg = (u32)st, cnt = (u32)(ed - st);
                        break;
                    }
                    if (k == HTAB_EMPTY) break;
                    h = (h + 1u) & hmask;
                }
            }
        }
        sbeg[t] = beg;
        scnt[t] = cnt;
        tot += cnt;
    }
    pcnt[p] = tot;
}

// ---- k-mer self-score order (fsearch.py:2647-2656, 2660, 2668) -----------------------------------
// kscs over the shortest seed span on the MASKED query, then the reference quicksort of positions
// by -ksc.  korder[qoff[q] + r] = r-th position.  One wave per query with the
// ((KSC_BIAS - ksc) << 12 | position) words in LDS and the wave replaying the reference quicksort;
// queries longer than LDS_SORT_MAX windows take the one-thread-per-query global-memory kernel.
#define LDS_SORT_MAX 4096
#define KSC_BIAS (1 << 18)

template <int CAP, int LO>  // serves queries with LO < windows <= CAP
__global__ __launch_bounds__(64) void k_ksc_order_lds(const u8* __restrict__ q_scls, const u32* __restrict__ qoff, u32 nq, int mink,
                                                      const signed char* __restrict__ b62c, u32* __restrict__ korder,
                                                      const u32* __restrict__ list /*or null: every query; else the grid runs over it*/,
                                                      u8* __restrict__ have) {
    __shared__ signed char s_self[SCLS_N];
    constexpr int LEAFCAP = CAP <= 512 ? 128 : CAP <= 1024 ? 256 : WQS_LEAF;  // leaf list sized with the instance (LDS = residency)
    __shared__ u32 s_x[CAP];
    __shared__ u16 s_L[CAP], s_R[CAP];
    __shared__ int s_leaf[2 * LEAFCAP];
    if (threadIdx.x < SCLS_N) s_self[threadIdx.x] = b62c[threadIdx.x * SCLS_N + threadIdx.x];
    __syncthreads();
    const u32 q = list ? list[blockIdx.x] : blockIdx.x;
    if (list && have[q]) return;   // computed for an earlier chunk of this batch
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    if (nk <= LO || nk > CAP) return;
    const u8* c = q_scls + base;
    for (int i = threadIdx.x; i < nk; i += 64) {
        int sc = 0;
        for (int j = 0; j < mink; ++j) sc += s_self[c[i + j]];
        s_x[i] = ((u32)(KSC_BIAS - sc) << 12) | (u32)i;
    }
    __syncthreads();
    wave_ref_qsort<LEAFCAP>(s_x, nk, [](u32 v) { return (int)(v >> 12); }, 0x7fffffff, s_L, s_R, s_leaf);
    for (int i = threadIdx.x; i < nk; i += 64) korder[base + i] = s_x[i] & 0xFFFu;
    if (list && threadIdx.x == 0) have[q] = 1;
}

// Queries with more than LDS_SORT_MAX windows (proteins above ~4100 residues): the same wave-parallel replay on 64-bit words
// ((KSC_BIAS - ksc) << 32 | position) and 32-bit misfit lists in global scratch (slices at the query's residue offset).  Until
// round 4 one THREAD sorted such a query: a 30 000-residue protein took 370 ms, seven times the whole config-3 search.
__global__ __launch_bounds__(64) void k_ksc_order_g(const u8* __restrict__ q_scls, const u32* __restrict__ qoff, u32 q0, u32 nq, int mink,
                                                    const signed char* __restrict__ b62c /*24x24*/, u64* __restrict__ gx, u32* __restrict__ gL,
                                                    u32* __restrict__ gR, u32* __restrict__ korder, const u32* __restrict__ list, u8* __restrict__ have) {
    __shared__ signed char s_self[SCLS_N];
    __shared__ u64 s_leafbuf[64 * WQS_LEAFBUF];
    __shared__ int s_leaf[2 * WQS_LEAF];
    if (threadIdx.x < SCLS_N) s_self[threadIdx.x] = b62c[threadIdx.x * SCLS_N + threadIdx.x];
    __syncthreads();
    const u32 q = list ? list[blockIdx.x] : q0 + blockIdx.x;
    if (q >= nq || q < q0) return;
    if (list && have[q]) return;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    if (nk <= LDS_SORT_MAX) return;  // done by k_ksc_order_lds (or nothing to do)
    const u8* c = q_scls + base;
    u64* x = gx + base;
    for (int i = threadIdx.x; i < nk; i += 64) {
        int sc = 0;
        for (int j = 0; j < mink; ++j) sc += s_self[c[i + j]];
        x[i] = ((u64)(u32)(KSC_BIAS - sc) << 32) | (u32)i;
    }
    __syncthreads();
    wave_ref_qsort<WQS_LEAF, 16>(x, nk, [](u64 v) { return (int)(v >> 32); }, 0x7fffffff, gL + base, gR + base, s_leaf, s_leafbuf);
    for (int i = threadIdx.x; i < nk; i += 64) korder[base + i] = (u32)x[i];
    if (list && threadIdx.x == 0) have[q] = 1;
}

// ---- high-frequency cap (fsearch.py:2667-2677) ---------------------------------------------------
// Positions are taken in korder until the running bucket total exceeds threshold * len; the test precedes the
// add, so position r is taken iff the total of the positions before it is <= the limit: a prefix of the order.
// One wave per query: 64 counts per step, wave prefix sum (u64), carried total; also returns the query's hits.
__global__ __launch_bounds__(256) void k_cap(const u32* __restrict__ korder, const u32* __restrict__ qoff, u32 q0, u32 nq, int mink,
                                             const u32* __restrict__ pcnt, i64 threshold, u8* __restrict__ mark,
                                             unsigned long long* __restrict__ qhits, const u32* __restrict__ list, u32 nlist) {
    // (with a list -- the queries k_cap_all left open --: the grid runs over it, a query outside [q0, nq) is another call's)
    const u32 w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (list && w >= nlist) return;
    const u32 q = list ? list[w] : q0 + w;
    const int lane = threadIdx.x & 63;
    if (q >= nq || q < q0) return;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    unsigned long long cum = 0;  // total of the positions taken so far (wave-uniform)
    if (nk > 0) {
        const u32 pbase = base + q;  // packed position of residue 0
        const i64 thr = threshold * (i64)ql;
        for (int r0 = 0; r0 < nk; r0 += 64) {
            const int r = r0 + lane;
            u32 pos = 0;
            unsigned long long v = 0;
            if (r < nk) {
                pos = korder[base + r];
                v = pcnt[pbase + pos];
            }
            unsigned long long inc = v;  // inclusive prefix over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const bool take = (r < nk) && ((i64)(cum + inc - v) <= thr);
            if (take) mark[pbase + pos] = 1;
            const unsigned long long tb = __ballot(take);
            // taken lanes form a prefix of the wave: the new total is the inclusive prefix of the last taken lane
            const int ntake = __popcll(tb);
            if (ntake) cum += __shfl(inc, ntake - 1);
            if (ntake < 64 && ntake < nk - r0) break;  // a position was refused: everything after it is refused too
        }
    }
    if (lane == 0) qhits[q] = cum;  // seed hits this query will visit in this chunk
}

// The cap without the order: the loop above refuses a position only when the total of the positions BEFORE it exceeds the limit, so a
// query whose windows' counts sum to no more than the limit keeps every window whatever the order (fsearch.py:2667-2677: the `break`
// is never reached).  One wave per query adds the counts up; a query at or below its limit has all its windows marked and its total
// returned, any other is appended to open_list (*over counts them; their order in the list is whatever the atomics make it, and nothing
// depends on it) -- the host then has the k-mer orders of those queries computed and runs k_cap over the list.
__global__ __launch_bounds__(256) void k_cap_all(const u32* __restrict__ qoff, u32 nq, int mink, const u32* __restrict__ pcnt, i64 threshold,
                                                 u8* __restrict__ mark, unsigned long long* __restrict__ qhits, unsigned long long* __restrict__ over,
                                                 u32* __restrict__ open_list) {
    const u32 q = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    const u32 pbase = base + q;
    unsigned long long sum = 0;
    for (int r = lane; r < nk; r += 64) sum += pcnt[pbase + r];
#pragma unroll
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
    if (nk > 0 && (i64)sum > threshold * (i64)ql) {
        if (lane == 0) open_list[atomicAdd(over, 1ull)] = q, qhits[q] = ~0ull;
        return;
    }
    for (int r = lane; r < nk; r += 64) mark[pbase + r] = 1;
    if (lane == 0) qhits[q] = sum;
}

// ---- effective per-seed hit counts + compaction of non-empty seeds --------------------------------
// (a pass touches the seed slots of ITS queries only, [AS * p_lo, AS * p_hi): with several passes per chunk -- one per query length
// class -- whole-batch sweeps per pass added up)
__global__ __launch_bounds__(256) void k_effcnt(const u8* __restrict__ mark, const u32* __restrict__ scnt, int AS, size_t t_lo, size_t t_hi,
                                                u32* __restrict__ eff, u32* __restrict__ nz) {
    const size_t t = t_lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= t_hi) return;
    const u32 p = (u32)(t / AS);
    u32 c = mark[p] ? scnt[t] : 0u;
    eff[t] = c;
    nz[t] = c ? 1u : 0u;
}

// Compacted seed list for the lookup kernel: one record per cap-selected seed window with a non-empty
// bucket.  cs_hoff = first hit ordinal, cs_base = bucket begin - first hit ordinal (so that hit h reads
// slot cs_base + h, u32 wrap-around intended), cs_kbase = the part of the sort key that does not
// depend on the index entry: q, qpos (twice: diagonal field and position field) and as.  Per hit
//   key = kbase + D(entry)      (k_index.hip: k_encode_delta).
__global__ __launch_bounds__(256) void k_compact_seeds(const u8* __restrict__ mark, const u32* __restrict__ scnt, const u32* __restrict__ hoff,
                                                       const u32* __restrict__ cidx, const u32* __restrict__ sbeg,
                                                       const u32* __restrict__ q_pseq, const u32* __restrict__ qoff, size_t t_lo, size_t t_hi, int AS,
                                                       KeyLayout kl, u32* __restrict__ cs_hoff, u32* __restrict__ cs_base,
                                                       u64* __restrict__ cs_kbase) {
    const size_t t = t_lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= t_hi) return;
    const u32 as = (u32)(t % AS), p = (u32)(t / AS);
    if (!mark[p] || !scnt[t]) return;   // the slot's effective count (k_effcnt's rule)
    const u32 k = cidx[t];
    const u32 q = q_pseq[p];
    const u32 qpos = p - (qoff[q] + q);
    cs_hoff[k] = hoff[t];
    cs_base[k] = sbeg[t] - hoff[t];
    cs_kbase[k] = ((u64)q << kl.sh_q) | ((u64)qpos << kl.sh_diag) | ((u64)qpos << kl.sh_qpos) | ((u64)as << kl.sh_as);
}

// ---- the seed-lookup kernel ------------------------------------------------------------------------
// Hit h (0 <= h < H) is index slot cs_base[k] + h of the compacted seed k that owns it.
// Work unit = one WAVE: LW_HITS consecutive hit ordinals, no block-level barrier anywhere.  The wave
// loads the seeds that cover its range (base / kbase into LDS, start marks into an owner map) and then
// walks 64 consecutive ordinals per step: owner = wave prefix-max of the marks (DPP scan, carried
// across steps in an SGPR), lanes read consecutive index slots (coalesced 8-byte key deltas, all
// LW_ITERS loads in flight before the first use) and write consecutive 8-byte keys.  A full tile with
// <= LW_SEEDS seeds takes the branch-free fast path; the last (partial) tile and tiles of many tiny
// buckets take a generic loop.
// Subject resolution follows the reference's strict `soas[j] < x` rule: an entry at offset 0 of chunk
// sequence j belongs to the previous non-empty sequence at sst = its length; at the chunk start it
// resolves to index -1, can never score, and is dropped (key = ~0).  Such entries are flagged in
// bit 63 of their delta and redone after the main loop.
#define LW_SEEDS 256
#define LW_WAVES 4
static constexpr int lw_iters() { return 16; }  // 64-hit steps per wave tile (config 2: 4 -> 0.54 ms per launch, 8 -> 0.48 ms, 16 -> 0.46 ms)

// first compacted seed of every lookup wave: largest k with cs_hoff[k] <= wave * LW_HITS
__global__ __launch_bounds__(256) void k_lookup_blockfirst(const u32* __restrict__ cs_hoff, u32 K, u32 H, u32 nw, u32 LW_HITS,
                                                           u32* __restrict__ wave_first) {
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b > nw) return;
    if (b == nw) {
        wave_first[b] = K;
        return;
    }
    const u32 h = b * LW_HITS;
    u32 lo = 0, hi = K;  // cs_hoff[0] == 0 <= h
    while (hi - lo > 1) {
        u32 m = (lo + hi) >> 1;
        if (cs_hoff[m] <= h) lo = m;
        else hi = m;
    }
    wave_first[b] = lo;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive max-scan over the 64 lanes (values >= 0): 4 row shifts + 2 row broadcasts, all DPP
__device__ __forceinline__ u32 wave_scan_max(u32 x) {
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));  // row_shr:1
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));  // row_shr:2
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));  // row_shr:4
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));  // row_shr:8
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false)); // row_bcast:15 -> rows 1, 3
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false)); // row_bcast:31 -> rows 2, 3
    return x;
}

// the reference's resolution of an entry at offset 0 of its sequence (flagged delta)
__device__ __forceinline__ u64 lookup_offset0_key(u32 j, u32 tag, u64 kb, const u32* __restrict__ roff, const KeyLayout& kl, u32 maxslen) {
    // previous NON-EMPTY sequence at sst = its length; none -> index -1 -> dropped
    while (j > 0 && roff[j] == roff[j - 1]) --j;
    if (j == 0) return ~0ull;
    j -= 1;
    const u32 sst = roff[j + 1] - roff[j];
    return kb + ((u64)j << kl.sh_subj) + ((u64)(maxslen - sst) << kl.sh_diag) + (u64)tag;
}

// ENT = u32: compact addends (k_encode_delta32), ENT = u64: full addends (k_encode_delta)
template <class ENT>
struct Addend;
template <>
struct Addend<u64> {
    static __device__ __forceinline__ u64 expand(u64 d, const KeyLayout&) { return d; }
    static __device__ __forceinline__ bool offset0(u64 d, const KeyLayout&) { return (d >> 63) != 0; }
    static __device__ __forceinline__ u32 subj(u64 d, const KeyLayout&) { return (u32)(d >> 8) & 0x7FFFFFFFu; }
    static __device__ __forceinline__ u32 tag(u64 d, const KeyLayout&) { return (u32)d & 0xFFu; }
};
template <>
struct Addend<u32> {
    static __device__ __forceinline__ u64 expand(u32 c, const KeyLayout& kl) {
        return kl.ba ? (((u64)(c >> kl.ba) << kl.sh_diag) + (u64)(c & ((1u << kl.ba) - 1u))) : ((u64)c << kl.sh_diag);
    }
    // k_encode_band32 resolves offset-0 entries itself; the ones the reference drops are all-ones (real addends stay below 2^31)
    static __device__ __forceinline__ bool offset0(u32 c, const KeyLayout&) { return (int)c < 0; }
    static __device__ __forceinline__ u32 subj(u32, const KeyLayout&) { return 0u; }   // lookup_offset0_key(0, ...) = dropped
    static __device__ __forceinline__ u32 tag(u32, const KeyLayout&) { return 0u; }
};

template <int LW_ITERS, class ENT>
__global__ __launch_bounds__(64 * LW_WAVES, 8) void k_lookup(const u32* __restrict__ cs_hoff, const u32* __restrict__ cs_base,
                                                          const u64* __restrict__ cs_kbase, const u32* __restrict__ wave_first, u32 K,
                                                          u32 H, u32 nw, const ENT* __restrict__ dkeys,
                                                          const u32* __restrict__ roff /*chunk off*/, KeyLayout kl, u32 maxslen,
                                                          u64* __restrict__ keys) {
    constexpr u32 LW_HITS = 64 * LW_ITERS;
    __shared__ u16 s_owner_all[LW_WAVES][LW_HITS];
    __shared__ u32 s_base_all[LW_WAVES][LW_SEEDS];
    __shared__ u64 s_kb_all[LW_WAVES][LW_SEEDS];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 wid = __builtin_amdgcn_readfirstlane(blockIdx.x * LW_WAVES + w);
    if (wid >= nw) return;
    u16* s_owner = s_owner_all[w];
    u32* s_base = s_base_all[w];
    u64* s_kb = s_kb_all[w];
    const u32 k0 = __builtin_amdgcn_readfirstlane(wave_first[wid]);
    u32 k1 = __builtin_amdgcn_readfirstlane(wave_first[wid + 1]);
    if (k1 >= K) k1 = K - 1;
    const u32 ns = k1 - k0 + 1;  // <= LW_HITS + 1
    const u32 lo = wid * LW_HITS;
    {
        // LW_HITS u16 marks = LW_ITERS * 128 B: 8 B per lane and store
#pragma unroll
        for (u32 i = 0; i < LW_ITERS / 4; ++i) reinterpret_cast<uint2*>(s_owner)[i * 64 + lane] = make_uint2(0, 0);
    }
    wave_lds_sync();
    // seed i (relative to k0) starts at ordinal cs_hoff[k0 + i]; seed 0 covers the tile start
    for (u32 i = lane; i < ns; i += 64) {
        if (i < LW_SEEDS) {
            s_base[i] = cs_base[k0 + i];
            s_kb[i] = cs_kbase[k0 + i];
        }
        if (i) {
            const u32 o = cs_hoff[k0 + i] - lo;
            if (o < LW_HITS) s_owner[o] = (u16)i;
        }
    }
    wave_lds_sync();
    u64* const kp = keys + lo;  // wave-uniform base: the per-lane part is lane * 8 + constant
    if (ns <= LW_SEEDS && lo + LW_HITS <= H) {
        // ---- fast path: full tile, every seed staged in LDS ----
        ENT e[LW_ITERS];
        u32 own[LW_ITERS];
        u32 carry = 0;
#pragma unroll
        for (int it = 0; it < LW_ITERS; ++it) {
            const u32 inc = wave_scan_max((u32)s_owner[it * 64 + lane]);
            const u32 a = max(inc, carry);
            carry = max(carry, (u32)__builtin_amdgcn_readlane((int)inc, 63));
            own[it] = a;
            const u32 byte_off = (s_base[a] + (lo + it * 64 + lane)) * (u32)sizeof(ENT);  // < 2^32: chunk entries < 2^29 (host check)
            e[it] = *reinterpret_cast<const ENT*>(reinterpret_cast<const char*>(dkeys) + byte_off);
        }
        u32 fix = 0;
#pragma unroll
        for (int it = 0; it < LW_ITERS; ++it) {
            const u64 key = s_kb[own[it]] + Addend<ENT>::expand(e[it], kl);
            // streamed once, read back by the sort: non-temporal stores keep the 1.9 GB key stream from evicting
            // the index addends out of L2 / Infinity Cache (0.64 -> 0.47 ms on config 2)
            __builtin_nontemporal_store(key, kp + it * 64 + lane);
            fix |= (u32)Addend<ENT>::offset0(e[it], kl) << it;
        }
        if (fix) {  // rare: entries at offset 0 of their sequence
#pragma unroll 1
            for (int it = 0; it < LW_ITERS; ++it) {
                if (!((fix >> it) & 1u)) continue;
                const u32 hl = it * 64 + lane;
                u32 a = 0;
                for (u32 i = 1; i < ns; ++i)
                    if (cs_hoff[k0 + i] - lo <= hl) a = i;
                const ENT d = dkeys[s_base[a] + lo + hl];
                kp[hl] = lookup_offset0_key(Addend<ENT>::subj(d, kl), Addend<ENT>::tag(d, kl), s_kb[a], roff, kl, maxslen);
            }
        }
        return;
    }
    // ---- generic path: partial last tile, or more seeds than the LDS stage holds ----
    u32 carry = 0;
#pragma unroll 1
    for (int it = 0; it < LW_ITERS; ++it) {
        const u32 hl = it * 64 + lane, h = lo + hl;
        const u32 inc = wave_scan_max((u32)s_owner[hl]);
        const u32 a = max(inc, carry);
        carry = max(carry, (u32)__builtin_amdgcn_readlane((int)inc, 63));
        if (h >= H) continue;
        const u32 base = a < LW_SEEDS ? s_base[a] : cs_base[k0 + a];
        const u64 kb = a < LW_SEEDS ? s_kb[a] : cs_kbase[k0 + a];
        const ENT d = dkeys[base + h];
        keys[h] = Addend<ENT>::offset0(d, kl) ? lookup_offset0_key(Addend<ENT>::subj(d, kl), Addend<ENT>::tag(d, kl), kb, roff, kl, maxslen)
                                              : kb + Addend<ENT>::expand(d, kl);
    }
}

// ---- launch wrappers ------------------------------------------------------------------------------
void launch_qhash(const u32* words, u32 Ppad, const SeedCfg& cfg, const HashLut& lut, u32* qbucket, hipStream_t st) {
    if (!Ppad) return;
    hipLaunchKernelGGL(k_qhash, dim3((Ppad + TILE_POS - 1) / TILE_POS), dim3(TILE_POS), 0, st, words, Ppad, cfg, lut, qbucket);
}

void launch_bounds(const u32* qbucket, u32 Ppad, int AS, const u32* hkey, const u64* hval, int hshift, u32 hmask, const u64* dir, const u32* ubeg,
                   u32 NC, u32 E, u32* sbeg, u32* scnt, u32* pcnt, hipStream_t st) {
    if (!Ppad) return;
    if (dir) hipLaunchKernelGGL(k_bounds<true>, dim3((Ppad + 255) / 256), dim3(256), 0, st, qbucket, Ppad, AS, hkey, hval, hshift, hmask,
                                reinterpret_cast<const uint2*>(dir), ubeg, NC, E, sbeg, scnt, pcnt);
    else hipLaunchKernelGGL(k_bounds<false>, dim3((Ppad + 255) / 256), dim3(256), 0, st, qbucket, Ppad, AS, hkey, hval, hshift, hmask,
                            (const uint2*)nullptr, ubeg, NC, E, sbeg, scnt, pcnt);
}

void launch_ksc_order(const u8* q_scls, const u32* qoff, u32 nq, u32 q_long /*first batch slot that may hold more than LDS_SORT_MAX windows*/, int mink,
                      const signed char* b62c, u64* gx, u32* gL, u32* gR /*global scratch per residue: only when q_long < nq*/, u32* korder, hipStream_t st,
                      hipStream_t st_long /*where the global-scratch instance runs (the caller orders it against st)*/) {
    if (!nq) return;
    // (one wave per query: the instance for the typical protein keeps 4 KB of LDS so that a CU holds 32 of them)
    hipLaunchKernelGGL((k_ksc_order_lds<512, 0>), dim3(nq), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, nullptr, nullptr);
    hipLaunchKernelGGL((k_ksc_order_lds<1024, 512>), dim3(nq), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, nullptr, nullptr);
    hipLaunchKernelGGL((k_ksc_order_lds<LDS_SORT_MAX, 1024>), dim3(nq), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, nullptr, nullptr);
    if (q_long < nq)
        hipLaunchKernelGGL(k_ksc_order_g, dim3(nq - q_long), dim3(64), 0, st_long, q_scls, qoff, q_long, nq, mink, b62c, gx, gL, gR, korder, nullptr, nullptr);
}
// ... of the listed queries that do not have theirs yet (have[q] == 0; set here); the global-scratch instance only when the caller saw a
// listed query that needs it (q_long < nq: slots from q_long on)
void launch_ksc_order_list(const u8* q_scls, const u32* qoff, u32 nq, const u32* list, u32 nlist, u8* have, u32 q_long, int mink, const signed char* b62c,
                           u64* gx, u32* gL, u32* gR, u32* korder, hipStream_t st, hipStream_t st_long) {
    if (!nlist) return;
    hipLaunchKernelGGL((k_ksc_order_lds<512, 0>), dim3(nlist), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, list, have);
    hipLaunchKernelGGL((k_ksc_order_lds<1024, 512>), dim3(nlist), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, list, have);
    hipLaunchKernelGGL((k_ksc_order_lds<LDS_SORT_MAX, 1024>), dim3(nlist), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder, list, have);
    if (q_long < nq)
        hipLaunchKernelGGL(k_ksc_order_g, dim3(nlist), dim3(64), 0, st_long, q_scls, qoff, q_long, nq, mink, b62c, gx, gL, gR, korder, list, have);
}
int ksc_lds_max() { return LDS_SORT_MAX; }

void launch_cap(const u32* korder, const u32* qoff, u32 q0, u32 nq /*queries [q0, nq)*/, int mink, const u32* pcnt, i64 threshold, u8* mark,
                unsigned long long* qhits, const u32* list /*or null; else: those of the nlist listed queries that lie in [q0, nq)*/, u32 nlist, hipStream_t st) {
    if (nq <= q0 || (list && !nlist)) return;
    const u32 n = list ? nlist : nq - q0;
    hipLaunchKernelGGL(k_cap, dim3((n + 3) / 4), dim3(256), 0, st, korder, qoff, q0, nq, mink, pcnt, threshold, mark, qhits, list, nlist);
}

void launch_cap_all(const u32* qoff, u32 nq, int mink, const u32* pcnt, i64 threshold, u8* mark, unsigned long long* qhits,
                    unsigned long long* over /*zeroed here*/, u32* open_list /*nq slots*/, hipStream_t st) {
    if (!nq) return;
    HIP_CHECK(hipMemsetAsync(over, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_cap_all, dim3((nq + 3) / 4), dim3(256), 0, st, qoff, nq, mink, pcnt, threshold, mark, qhits, over, open_list);
}

void launch_effcnt(const u8* mark, const u32* scnt, int AS, u32 p_lo, u32 p_hi, u32* eff, u32* nz, hipStream_t st) {
    const size_t t_lo = (size_t)AS * p_lo, t_hi = (size_t)AS * p_hi;
    if (t_hi <= t_lo) return;
    hipLaunchKernelGGL(k_effcnt, dim3((unsigned)((t_hi - t_lo + 255) / 256)), dim3(256), 0, st, mark, scnt, AS, t_lo, t_hi, eff, nz);
}

void launch_compact_seeds(const u8* mark, const u32* scnt, const u32* hoff, const u32* cidx, const u32* sbeg, const u32* q_pseq, const u32* qoff,
                          u32 p_lo, u32 p_hi, int AS, const KeyLayout& kl, u32* cs_hoff, u32* cs_base, u64* cs_kbase, hipStream_t st) {
    const size_t t_lo = (size_t)AS * p_lo, t_hi = (size_t)AS * p_hi;
    if (t_hi <= t_lo) return;
    hipLaunchKernelGGL(k_compact_seeds, dim3((unsigned)((t_hi - t_lo + 255) / 256)), dim3(256), 0, st, mark, scnt, hoff, cidx, sbeg, q_pseq, qoff,
                       t_lo, t_hi, AS, kl, cs_hoff, cs_base, cs_kbase);
}

u32 lookup_num_blocks(u32 H) { return (H + 64u * lw_iters() - 1) / (64u * lw_iters()); }  // number of lookup WAVES

void launch_lookup_blockfirst(const u32* cs_hoff, u32 K, u32 H, u32* wave_first, hipStream_t st) {
    u32 nw = lookup_num_blocks(H);
    hipLaunchKernelGGL(k_lookup_blockfirst, dim3((nw + 1 + 255) / 256), dim3(256), 0, st, cs_hoff, K, H, nw, 64u * lw_iters(), wave_first);
}

void launch_lookup(const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase, const u32* wave_first, u32 K, u32 H,
                   const void* dkeys, bool compact, const u32* roff, const KeyLayout& kl, u32 maxslen, u64* keys, hipStream_t st) {
    if (!H) return;
    const u32 nw = lookup_num_blocks(H);
    const dim3 g((nw + LW_WAVES - 1) / LW_WAVES), bl(64 * LW_WAVES);
    if (compact)
        hipLaunchKernelGGL((k_lookup<lw_iters(), u32>), g, bl, 0, st, cs_hoff, cs_base, cs_kbase, wave_first, K, H, nw, (const u32*)dkeys, roff, kl, maxslen, keys);
    else
        hipLaunchKernelGGL((k_lookup<lw_iters(), u64>), g, bl, 0, st, cs_hoff, cs_base, cs_kbase, wave_first, K, H, nw, (const u64*)dkeys, roff, kl, maxslen, keys);
}
