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
__global__ __launch_bounds__(256) void k_bounds(const u32* __restrict__ qbucket, u32 Ppad, int AS, const u32* __restrict__ start,
                                                u32 NC, u32 E, u32* __restrict__ sbeg, u32* __restrict__ scnt,
                                                u32* __restrict__ pcnt) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= Ppad) return;
    const i64 L = (i64)E - 1;  // self.L = len(self.locus) - 1
    u32 tot = 0;
    for (int as = 0; as < AS; ++as) {
        const size_t t = (size_t)as * Ppad + p;
        const u32 b = qbucket[t];
        u32 beg = 0, cnt = 0;
        if (b != 0xFFFFFFFFu) {
            i64 st = start[b];
            i64 ed = (b + 1u < NC) ? (i64)start[b + 1] : st;  // bucket NC-1: `except: st = ed`
            ed = ed < L ? ed : L;
            if (ed > st) beg = (u32)st, cnt = (u32)(ed - st);
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
// ((KSC_BIAS - ksc) << 12 | position) words in LDS and lane 0 replaying the reference quicksort;
// queries longer than LDS_SORT_MAX windows take the one-thread-per-query global-memory kernel.
#define LDS_SORT_MAX 4096
#define KSC_BIAS (1 << 18)

__global__ __launch_bounds__(64) void k_ksc_order_lds(const u8* __restrict__ q_scls, const u32* __restrict__ qoff, u32 nq, int mink,
                                                      const signed char* __restrict__ b62c, u32* __restrict__ korder) {
    __shared__ signed char s_self[SCLS_N];
    __shared__ u32 s_x[LDS_SORT_MAX];
    if (threadIdx.x < SCLS_N) s_self[threadIdx.x] = b62c[threadIdx.x * SCLS_N + threadIdx.x];
    __syncthreads();
    const u32 q = blockIdx.x;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    if (nk <= 0 || nk > LDS_SORT_MAX) return;
    const u8* c = q_scls + base;
    for (int i = threadIdx.x; i < nk; i += 64) {
        int sc = 0;
        for (int j = 0; j < mink; ++j) sc += s_self[c[i + j]];
        s_x[i] = ((u32)(KSC_BIAS - sc) << 12) | (u32)i;
    }
    __syncthreads();
    if (threadIdx.x == 0) ref_qsort_dev(s_x, nk, [](u32 v) { return (int)(v >> 12); });
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += 64) korder[base + i] = s_x[i] & 0xFFFu;
}

__global__ __launch_bounds__(64) void k_ksc_order(const u8* __restrict__ q_scls, const u32* __restrict__ qoff, u32 nq, int mink,
                                                  const signed char* __restrict__ b62c /*24x24*/, int* __restrict__ ksc,
                                                  u32* __restrict__ korder) {
    __shared__ signed char s_self[SCLS_N];
    if (threadIdx.x < SCLS_N) s_self[threadIdx.x] = b62c[threadIdx.x * SCLS_N + threadIdx.x];
    __syncthreads();
    const u32 q = blockIdx.x * 64u + threadIdx.x;
    if (q >= nq) return;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    if (nk <= LDS_SORT_MAX) return;  // done by k_ksc_order_lds (or nothing to do)
    const u8* c = q_scls + base;
    int* k = ksc + base;
    u32* x = korder + base;
    int sc = 0;
    for (int i = 0; i < mink; ++i) sc += s_self[c[i]];
    k[0] = sc;
    x[0] = 0;
    for (int i = 1; i < nk; ++i) {
        sc = sc - s_self[c[i - 1]] + s_self[c[i - 1 + mink]];
        k[i] = sc;
        x[i] = (u32)i;
    }
    ref_qsort_dev(x, nk, [k](u32 i) { return -k[i]; });
}

// ---- high-frequency cap (fsearch.py:2667-2677) ---------------------------------------------------
__global__ __launch_bounds__(64) void k_cap(const u32* __restrict__ korder, const u32* __restrict__ qoff, u32 nq, int mink,
                                            const u32* __restrict__ pcnt, i64 threshold, u8* __restrict__ mark) {
    const u32 q = blockIdx.x * 64u + threadIdx.x;
    if (q >= nq) return;
    const u32 base = qoff[q];
    const int ql = (int)(qoff[q + 1] - base);
    const int nk = ql - mink + 1;
    if (nk <= 0) return;
    const u32 pbase = base + q;  // packed position of residue 0
    const i64 thr = threshold * (i64)ql;
    i64 cum = 0;
    for (int r = 0; r < nk; ++r) {
        if (cum > thr) break;
        u32 pos = korder[base + r];
        cum += pcnt[pbase + pos];
        mark[pbase + pos] = 1;
    }
}

// ---- effective per-seed hit counts + compaction of non-empty seeds --------------------------------
__global__ __launch_bounds__(256) void k_effcnt(const u8* __restrict__ mark, const u32* __restrict__ scnt, u32 Ppad, int AS,
                                                u32* __restrict__ eff, u32* __restrict__ nz) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)AS * Ppad) return;
    const u32 p = (u32)(t % Ppad);
    u32 c = mark[p] ? scnt[t] : 0u;
    eff[t] = c;
    nz[t] = c ? 1u : 0u;
}

// cs_*[k] for the k-th non-empty seed: first hit ordinal, first slot, query, (as << 24) | qpos
__global__ __launch_bounds__(256) void k_compact_seeds(const u32* __restrict__ eff, const u32* __restrict__ hoff,
                                                       const u32* __restrict__ cidx, const u32* __restrict__ sbeg,
                                                       const u32* __restrict__ q_pseq, const u32* __restrict__ qoff, u32 Ppad, int AS,
                                                       u32* __restrict__ cs_hoff, u32* __restrict__ cs_beg, u32* __restrict__ cs_q,
                                                       u32* __restrict__ cs_qa) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)AS * Ppad) return;
    if (!eff[t]) return;
    const u32 as = (u32)(t / Ppad), p = (u32)(t % Ppad);
    const u32 k = cidx[t];
    const u32 q = q_pseq[p];
    cs_hoff[k] = hoff[t];
    cs_beg[k] = sbeg[t];
    cs_q[k] = q;
    cs_qa[k] = (as << 24) | (p - (qoff[q] + q));
}

// first compacted seed of every lookup block: largest k with cs_hoff[k] <= blk * LK_HITS
#define LK_THREADS 256
#define LK_PER_THREAD 8
#define LK_HITS (LK_THREADS * LK_PER_THREAD)

__global__ __launch_bounds__(256) void k_lookup_blockfirst(const u32* __restrict__ cs_hoff, u32 K, u32 H, u32 nblk,
                                                           u32* __restrict__ blk_first) {
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b > nblk) return;
    if (b == nblk) {
        blk_first[b] = K;
        return;
    }
    const u32 h = b * LK_HITS;
    u32 lo = 0, hi = K;  // cs_hoff[0] == 0 <= h
    while (hi - lo > 1) {
        u32 m = (lo + hi) >> 1;
        if (cs_hoff[m] <= h) lo = m;
        else hi = m;
    }
    blk_first[b] = lo;
}

// ---- the seed-lookup kernel ------------------------------------------------------------------------
// Hit h (0 <= h < H) is slot cs_beg[k] + (h - cs_hoff[k]) of the compacted seed k that owns it.
// Each block owns LK_HITS consecutive hit ordinals, stages the (<= LK_HITS + 1) seeds that cover
// them in LDS, and each wave walks 64 consecutive hits per step: lanes read consecutive index
// slots (coalesced 8-byte entries) and write consecutive 8-byte keys.
// Per hit: subject resolution with the reference's strict `soas[j] < x` rule (an entry at offset 0
// of chunk sequence j >= 1 belongs to sequence j-1 at sst = len(j-1); offset 0 of the chunk's
// first sequence resolves to index -1 and can never score -> dropped, key = ~0).
__global__ __launch_bounds__(LK_THREADS) void k_lookup(const u32* __restrict__ cs_hoff, const u32* __restrict__ cs_beg,
                                                       const u32* __restrict__ cs_q, const u32* __restrict__ cs_qa,
                                                       const u32* __restrict__ blk_first, u32 K, u32 H,
                                                       const u64* __restrict__ entries, const u32* __restrict__ roff /*chunk off*/,
                                                       KeyLayout kl, u64* __restrict__ keys) {
    __shared__ u32 s_off[LK_HITS + 2];
    __shared__ u32 s_beg[LK_HITS + 2];
    __shared__ u32 s_q[LK_HITS + 2];
    __shared__ u32 s_qa[LK_HITS + 2];
    const u32 k0 = blk_first[blockIdx.x];
    u32 k1 = blk_first[blockIdx.x + 1];  // last seed that can start inside this block (inclusive)
    if (k1 >= K) k1 = K - 1;
    const u32 ns = k1 - k0 + 1;  // <= LK_HITS + 1
    for (u32 i = threadIdx.x; i < ns; i += LK_THREADS) {
        s_off[i] = cs_hoff[k0 + i];
        s_beg[i] = cs_beg[k0 + i];
        s_q[i] = cs_q[k0 + i];
        s_qa[i] = cs_qa[k0 + i];
    }
    __syncthreads();
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 hbase = blockIdx.x * LK_HITS + wave * (64 * LK_PER_THREAD);
    u32 lo = 0;
#pragma unroll 2
    for (int it = 0; it < LK_PER_THREAD; ++it) {
        const u32 h = hbase + it * 64 + lane;
        if (h >= H) break;
        // largest i in [lo, ns) with s_off[i] <= h
        u32 a = lo, b = ns;
        while (b - a > 1) {
            u32 m = (a + b) >> 1;
            if (s_off[m] <= h) a = m;
            else b = m;
        }
        lo = a;
        const u32 slot = s_beg[a] + (h - s_off[a]);
        const u64 e = entries[slot];
        const u32 qa = s_qa[a];
        const u32 qpos = qa & 0xFFFFFFu, as = qa >> 24;
        u32 j = (u32)(e >> 32), tag = (u32)(e >> 24) & 0xFFu, pos = (u32)e & 0xFFFFFFu;
        u64 key;
        if (pos == 0 && j == 0) {
            key = ~0ull;
        } else {
            u32 sst = pos;
            if (pos == 0) {
                j -= 1;
                sst = roff[j + 1] - roff[j];
            }
            const u64 diag = (u64)((i64)qpos - (i64)sst + kl.diag_off);
            key = ((u64)s_q[a] << kl.sh_q) | ((u64)j << kl.sh_subj) | (diag << kl.sh_diag) | ((u64)qpos << kl.sh_qpos) |
                  ((u64)as << kl.sh_as) | (u64)tag;
        }
        keys[h] = key;
    }
}

// ---- launch wrappers ------------------------------------------------------------------------------
void launch_qhash(const u32* words, u32 Ppad, const SeedCfg& cfg, const HashLut& lut, u32* qbucket, hipStream_t st) {
    if (!Ppad) return;
    hipLaunchKernelGGL(k_qhash, dim3((Ppad + TILE_POS - 1) / TILE_POS), dim3(TILE_POS), 0, st, words, Ppad, cfg, lut, qbucket);
}

void launch_bounds(const u32* qbucket, u32 Ppad, int AS, const u32* start, u32 NC, u32 E, u32* sbeg, u32* scnt, u32* pcnt,
                   hipStream_t st) {
    if (!Ppad) return;
    hipLaunchKernelGGL(k_bounds, dim3((Ppad + 255) / 256), dim3(256), 0, st, qbucket, Ppad, AS, start, NC, E, sbeg, scnt, pcnt);
}

void launch_ksc_order(const u8* q_scls, const u32* qoff, u32 nq, int mink, const signed char* b62c, int* ksc, u32* korder,
                      hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_ksc_order_lds, dim3(nq), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, korder);
    hipLaunchKernelGGL(k_ksc_order, dim3((nq + 63) / 64), dim3(64), 0, st, q_scls, qoff, nq, mink, b62c, ksc, korder);
}

void launch_cap(const u32* korder, const u32* qoff, u32 nq, int mink, const u32* pcnt, i64 threshold, u8* mark, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_cap, dim3((nq + 63) / 64), dim3(64), 0, st, korder, qoff, nq, mink, pcnt, threshold, mark);
}

void launch_effcnt(const u8* mark, const u32* scnt, u32 Ppad, int AS, u32* eff, u32* nz, hipStream_t st) {
    size_t T = (size_t)AS * Ppad;
    if (!T) return;
    hipLaunchKernelGGL(k_effcnt, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, mark, scnt, Ppad, AS, eff, nz);
}

void launch_compact_seeds(const u32* eff, const u32* hoff, const u32* cidx, const u32* sbeg, const u32* q_pseq, const u32* qoff,
                          u32 Ppad, int AS, u32* cs_hoff, u32* cs_beg, u32* cs_q, u32* cs_qa, hipStream_t st) {
    size_t T = (size_t)AS * Ppad;
    if (!T) return;
    hipLaunchKernelGGL(k_compact_seeds, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, eff, hoff, cidx, sbeg, q_pseq, qoff,
                       Ppad, AS, cs_hoff, cs_beg, cs_q, cs_qa);
}

u32 lookup_num_blocks(u32 H) { return (H + LK_HITS - 1) / LK_HITS; }

void launch_lookup_blockfirst(const u32* cs_hoff, u32 K, u32 H, u32* blk_first, hipStream_t st) {
    u32 nblk = lookup_num_blocks(H);
    hipLaunchKernelGGL(k_lookup_blockfirst, dim3((nblk + 1 + 255) / 256), dim3(256), 0, st, cs_hoff, K, H, nblk, blk_first);
}

void launch_lookup(const u32* cs_hoff, const u32* cs_beg, const u32* cs_q, const u32* cs_qa, const u32* blk_first, u32 K, u32 H,
                   const u64* entries, const u32* roff, const KeyLayout& kl, u64* keys, hipStream_t st) {
    if (!H) return;
    hipLaunchKernelGGL(k_lookup, dim3(lookup_num_blocks(H)), dim3(LK_THREADS), 0, st, cs_hoff, cs_beg, cs_q, cs_qa, blk_first, K, H,
                       entries, roff, kl, keys);
}
