// k_phase2.hip -- phase 2 of blastp (fsearch.py:3028-3110) around the banded aligner:
// chunk-major candidate gather, reference quicksort by -ungapped score, top-vmax task list,
// sequential early-stop replay, reference quicksort by -bit, top-v output records.
#include "common.h"
#include "kernels.h"
#include "refsort.h"

// per-chunk candidate region (sorted by query, then first-touch) -> per-query chunk-major layout
//   dst = qcoff[q] + prior[q] + (i - cqoff[q])
__global__ __launch_bounds__(256) void k_gather_cands(const u32* __restrict__ src_q, const u32* __restrict__ src_rec, u32 n,
                                                      const u32* __restrict__ cqoff /*excl scan of this chunk's per-query counts*/,
                                                      const u32* __restrict__ prior /*sum of earlier chunks' counts*/,
                                                      const u32* __restrict__ qcoff, u32* __restrict__ dst_rec) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u32 q = src_q[i];
    const u32 d = qcoff[q] + prior[q] + (i - cqoff[q]);
    *reinterpret_cast<uint4*>(dst_rec + 4 * (size_t)d) = *reinterpret_cast<const uint4*>(src_rec + 4 * (size_t)i);
}

__global__ __launch_bounds__(256) void k_add_u32(u32* __restrict__ acc, const u32* __restrict__ x, u32 n) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

// qsort(hits, key=-score) (3051) per query + number of alignment tasks min(n, vmax) (3059, 3062).
// One wave per query: the candidates' (inverted score << 12 | index) words are staged in LDS and
// the wave replays the reference quicksort there (wave_ref_qsort: exact wave-parallel partitions);
// only ranges that reach into the first vmax positions are sorted.  Queries with more than
// LDS_SORT_MAX candidates take the global-memory kernel below.
#define LDS_SORT_MAX 4096
#define SCORE_CAP ((1u << 20) - 1u)

// tiles of one candidate: 1 on the normal path; on the long path (either sequence >= 4096,
// fsearch.py:3068, 3085) one per 4096-residue step of the query from qi (range(qi, li, chk), 1487)
__device__ __forceinline__ u32 cand_tiles(u32 lq, u32 ls, u32 qi) {
    if (lq < LONG_SEQ && ls < LONG_SEQ) return 1u;
    return qi < lq ? (lq - qi + LONG_SEQ - 1) / LONG_SEQ : 0u;
}

// CAP = LDS capacity of this instance; it serves the queries with LO < n <= CAP (a small instance keeps
// 4x more waves resident for the typical few-hundred-candidate lists).
template <int CAP, int LO>
__global__ __launch_bounds__(64) void k_csort_lds(const u32* __restrict__ rec, const u32* __restrict__ qcoff, u32 nq, u32 vmax,
                                                  const u32* __restrict__ qoff, const u32* __restrict__ roff, u32* __restrict__ perm,
                                                  u32* __restrict__ ntask, u32* __restrict__ ntile) {
    constexpr int LEAFCAP = CAP <= 512 ? 128 : CAP <= 1024 ? 256 : WQS_LEAF;  // leaf list sized with the instance (LDS = residency)
    __shared__ u32 s_x[CAP];
    __shared__ u16 s_L[CAP], s_R[CAP];
    __shared__ int s_leaf[2 * LEAFCAP];
    const u32 q = blockIdx.x;
    const u32 c0 = qcoff[q];
    const int n = (int)(qcoff[q + 1] - c0);
    if (n > CAP || (n <= LO && LO > 0)) return;  // another instance (or k_csort) handles it
    const u32* r = rec + 4 * (size_t)c0;
    for (int i = threadIdx.x; i < n; i += 64) {
        u32 sc = r[4 * (size_t)i + 1];
        sc = sc > SCORE_CAP ? SCORE_CAP : sc;
        s_x[i] = ((SCORE_CAP - sc) << 12) | (u32)i;
    }
    __syncthreads();
    wave_ref_qsort<LEAFCAP>(s_x, n, [](u32 v) { return (int)(v >> 12); }, (int)vmax, s_L, s_R, s_leaf);
    const int m = n < (int)vmax ? n : (int)vmax;
    const u32 lq = qoff[q + 1] - qoff[q];
    u32 tiles = 0;
    for (int i = threadIdx.x; i < m; i += 64) {
        const u32 c = s_x[i] & 0xFFFu;
        perm[c0 + i] = c;
        const u32 subj = r[4 * (size_t)c], qi = r[4 * (size_t)c + 2];
        tiles += cand_tiles(lq, roff[subj + 1] - roff[subj], qi);
    }
    for (int o = 32; o > 0; o >>= 1) tiles += __shfl_down(tiles, o);
    if (threadIdx.x == 0) ntask[q] = (u32)m, ntile[q] = tiles;
}

// queries with more than LDS_SORT_MAX candidates: the same exact wave-parallel quicksort on 64-bit words
// ((inverted score << 32) | index) and 32-bit misfit lists in global scratch (slices of the arrays at
// the query's candidate offset).  __syncthreads() in a one-wave block orders the global accesses.
__global__ __launch_bounds__(64) void k_csort(const u32* __restrict__ rec, const u32* __restrict__ qcoff, u32 nq, u32 vmax,
                                              const u32* __restrict__ qoff, const u32* __restrict__ roff, u32* __restrict__ perm,
                                              u32* __restrict__ ntask, u32* __restrict__ ntile, u64* __restrict__ gx,
                                              u32* __restrict__ gL, u32* __restrict__ gR) {
    __shared__ int s_leaf[2 * WQS_LEAF];
    const u32 q = blockIdx.x;
    const u32 c0 = qcoff[q];
    const int n = (int)(qcoff[q + 1] - c0);
    if (n <= LDS_SORT_MAX) return;  // done by k_csort_lds
    const u32* r = rec + 4 * (size_t)c0;
    u64* x = gx + c0;
    for (int i = threadIdx.x; i < n; i += 64) {
        u32 sc = r[4 * (size_t)i + 1];
        sc = sc > SCORE_CAP ? SCORE_CAP : sc;
        x[i] = ((u64)(SCORE_CAP - sc) << 32) | (u32)i;
    }
    __syncthreads();
    // (16-step scan batches and LDS leaf buffers, which help the k-mer order of very long queries, cost time here: 37 -> 48 ms per
    // step on the 100k weight-6 set -- the vmax cut prunes most of the recursion, what is left are a few long scans)
    wave_ref_qsort(x, n, [](u64 v) { return (int)(v >> 32); }, (int)vmax, gL + c0, gR + c0, s_leaf);
    const u32 m = (u32)n < vmax ? (u32)n : vmax;
    const u32 lq = qoff[q + 1] - qoff[q];
    u32 tiles = 0;
    for (u32 i = threadIdx.x; i < m; i += 64) {
        const u32 c = (u32)x[i];
        perm[c0 + i] = c;
        const u32 subj = r[4 * (size_t)c], qi = r[4 * (size_t)c + 2];
        tiles += cand_tiles(lq, roff[subj + 1] - roff[subj], qi);
    }
    for (int o = 32; o > 0; o >>= 1) tiles += __shfl_down(tiles, o);
    if (threadIdx.x == 0) ntask[q] = m, ntile[q] = tiles;
}

// tasks of a query in rank order; rk_slot[roffc[q] + r] = first task slot (relative to toff[q]) of rank r.
// One wave per query: lanes take ranks 64 at a time, a wave prefix sum of the per-rank tile counts
// gives the slots (1 per rank unless a sequence is >= 4096 aa).
__global__ __launch_bounds__(64) void k_mktasks(const u32* __restrict__ rec, const u32* __restrict__ qcoff, const u32* __restrict__ perm,
                                                const u32* __restrict__ ntask, const u32* __restrict__ roffc, const u32* __restrict__ toff,
                                                u32 nq, const u32* __restrict__ qoff, const u32* __restrict__ roff,
                                                AlnTask* __restrict__ tasks, u32* __restrict__ rk_slot) {
    const u32 q = blockIdx.x;
    const int lane = threadIdx.x;
    const u32 c0 = qcoff[q], t0 = toff[q], nt = ntask[q], r0 = roffc[q];
    const u32 lq = qoff[q + 1] - qoff[q];
    u32 carry = 0;
    for (u32 kb = 0; kb < nt; kb += 64) {
        const u32 k = kb + lane;
        AlnTask t;
        u32 ls = 0, tiles = 0;
        if (k < nt) {
            const u32 c = c0 + perm[c0 + k];
            const uint4 v = *reinterpret_cast<const uint4*>(rec + 4 * (size_t)c);
            t.q = q, t.subj = v.x, t.score = v.y, t.qi = v.z, t.qj = v.w, t.rank = k;
            ls = roff[t.subj + 1] - roff[t.subj];
            tiles = cand_tiles(lq, ls, t.qi);
        }
        u32 inc = tiles;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
        }
        const u32 total = __shfl(inc, 63);
        u32 slot = carry + inc - tiles;
        carry += total;
        if (k >= nt) continue;
        rk_slot[r0 + k] = slot;
        if (lq < LONG_SEQ && ls < LONG_SEQ) {
            t.qe = lq, t.se = ls;
            tasks[t0 + slot] = t;
        } else {
            // kswat_st_long (1480-1498): tile i over [qi + 4096 i, +4096) x [qj + 4096 i, +4096), each aligned from its corner
            const u32 qi0 = t.qi, qj0 = t.qj;
            for (u32 i = qi0, j = qj0; i < lq; i += LONG_SEQ, j += LONG_SEQ) {
                t.qi = i, t.qe = min(lq, i + LONG_SEQ);
                t.qj = min(j, ls), t.se = min(ls, j + LONG_SEQ);
                if (j >= ls) t.qj = ls, t.se = ls;  // sqj[j:jed] is empty
                tasks[t0 + slot++] = t;
            }
        }
    }
}

// ---- sequential stop rule in rounds -----------------------------------------------------------------
// The reference aligns a query's sorted candidates one by one and stops after `mmiss` consecutive
// misses (3052-3054, 3062-3104), so only a prefix of the top-vmax list is ever aligned.  The
// device aligns that list in growing rounds: each round aligns the next B ranks of every query
// that has not stopped, then k_stop_round_w replays the sequential rule over them.  Per-query state
// st_state[5*q + {0 next rank, 1 unmch, 2 bv, 3 nsel, 4 done}].
__device__ __forceinline__ u32 rank_slot(const u32* rk_slot, u32 r0, u32 r, u32 nt, u32 ntile_q) {
    return r < nt ? rk_slot[r0 + r] : ntile_q;
}

// rcnt[q] = ranks aligned this round, tcnt[q] = their tasks.  A query can only stop once `unmch`
// reaches ceil(mmiss), so a round aligns exactly the ranks that are needed before the rule could
// fire if they all miss (at least minr, to bound the number of rounds): the device then aligns
// almost exactly the prefix the reference's sequential loop aligns.
__global__ __launch_bounds__(256) void k_round_counts(const u32* __restrict__ ntask, const u32* __restrict__ ntile,
                                                      const u32* __restrict__ roffc, const u32* __restrict__ rk_slot,
                                                      const u32* __restrict__ qcoff, const u32* __restrict__ st_state, u32 nq,
                                                      double max_miss, u32 minr, u32* __restrict__ rcnt, u32* __restrict__ tcnt) {
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (q > nq) return;
    u32 c = 0, tc = 0;
    if (q < nq && !st_state[5 * (size_t)q + 4]) {
        const u32 nt = ntask[q], next = st_state[5 * (size_t)q], unmch = st_state[5 * (size_t)q + 1];
        const u32 left = nt - next;
        const u32 n = qcoff[q + 1] - qcoff[q];
        double mmiss = (double)n * max_miss + 1;
        const double inv = 100. / mmiss;
        mmiss = mmiss > inv ? mmiss : inv;
        mmiss = mmiss > 10. ? mmiss : 10.;
        mmiss = mmiss < 120. ? mmiss : 120.;
        const u32 cm = (u32)ceil(mmiss);
        u32 need = cm > unmch ? cm - unmch : 1u;
        need = need < minr ? minr : need;
        c = left < need ? left : need;
        tc = rank_slot(rk_slot, roffc[q], next + c, nt, ntile[q]) - rank_slot(rk_slot, roffc[q], next, nt, ntile[q]);
    }
    rcnt[q] = c;
    tcnt[q] = tc;
}

// ---- first round with SPECULATIVE TRACES (round 3) ----------------------------------------------------------------------------------
// Every reported row is aligned twice: score-only in the rounds, traced afterwards.  Candidates arrive ordered by ungapped score, and
// a candidate whose UNGAPPED score alone would already pass the e-value test almost always passes with gaps: the leading tasks of a
// query's first round that satisfy that test (a prefix: scores descend) are aligned with traces right away (k_align<true>, the trace
// kept in a slab of its own, its position in tpos[slot]); rows that end up reported and have a trace only need the traceback walk.
// A wrong guess costs the difference between the traced and the packed kernel for one alignment; results do not depend on the guess.
// tcnt_pk[q] = tasks of the round for the score-only kernel, scnt[q] = tasks traced right away, *any_rank += ranks of the round.
__global__ __launch_bounds__(256) void k_round_counts_spec(const u32* __restrict__ ntask, const u32* __restrict__ ntile,
                                                           const u32* __restrict__ roffc, const u32* __restrict__ rk_slot,
                                                           const u32* __restrict__ qcoff, const u32* __restrict__ st_state, u32 nq,
                                                           double max_miss, u32 minr, const AlnTask* __restrict__ tasks,
                                                           const u32* __restrict__ toff, const u32* __restrict__ qoff, const u32* __restrict__ roff,
                                                           const int* __restrict__ bittab, int bittab_n, i64 D, double expect,
                                                           u32* __restrict__ rcnt, u32* __restrict__ tcnt_pk, u32* __restrict__ scnt,
                                                           u32* __restrict__ any_rank) {
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    u32 c = 0, tc = 0, ns = 0;
    if (q < nq && !st_state[5 * (size_t)q + 4]) {
        const u32 nt = ntask[q], next = st_state[5 * (size_t)q], unmch = st_state[5 * (size_t)q + 1];
        const u32 left = nt - next;
        const u32 n = qcoff[q + 1] - qcoff[q];
        double mmiss = (double)n * max_miss + 1;
        const double inv = 100. / mmiss;
        mmiss = mmiss > inv ? mmiss : inv;
        mmiss = mmiss > 10. ? mmiss : 10.;
        mmiss = mmiss < 120. ? mmiss : 120.;
        const u32 cm = (u32)ceil(mmiss);
        u32 need = cm > unmch ? cm - unmch : 1u;
        need = need < minr ? minr : need;
        c = left < need ? left : need;
        const u32 s0 = rank_slot(rk_slot, roffc[q], next, nt, ntile[q]);
        tc = rank_slot(rk_slot, roffc[q], next + c, nt, ntile[q]) - s0;
        const i64 li = (i64)(qoff[q + 1] - qoff[q]);
        const size_t base = (size_t)toff[q] + s0;
        for (; ns < tc; ++ns) {
            const AlnTask tk = tasks[base + ns];
            const i64 lj = (i64)(roff[tk.subj + 1] - roff[tk.subj]);
            const int sc = (int)tk.score < bittab_n ? (int)tk.score : bittab_n - 1;
            const int bit = bittab[sc];
            const double p2 = bit > 1074 ? 0.0 : ldexp(1.0, -bit);
            if (!((double)(D * li * lj) * p2 <= expect)) break;
        }
    }
    if (q <= nq) rcnt[q] = c, tcnt_pk[q] = tc - ns, scnt[q] = ns;
    u32 sum = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += (u32)__shfl_xor((int)sum, o);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(any_rank, sum);
}

__global__ __launch_bounds__(64) void k_round_idx_spec(const u32* __restrict__ tcnt_pk, const u32* __restrict__ scnt, const u32* __restrict__ poff,
                                                       const u32* __restrict__ soff, const u32* __restrict__ toff, const u32* __restrict__ ntask,
                                                       const u32* __restrict__ ntile, const u32* __restrict__ roffc, const u32* __restrict__ rk_slot,
                                                       const u32* __restrict__ st_state, u32 nq, u32* __restrict__ ridx, u32* __restrict__ sidx) {
    const u32 q = blockIdx.x;
    const u32 ns = scnt[q], n = tcnt_pk[q] + ns;
    if (!n) return;
    const u32 base = toff[q] + rank_slot(rk_slot, roffc[q], st_state[5 * (size_t)q], ntask[q], ntile[q]);
    const u32 po = poff[q], so = soff[q];
    for (u32 k = threadIdx.x; k < n; k += 64) {
        if (k < ns) sidx[so + k] = base + k;
        else ridx[po + (k - ns)] = base + k;
    }
}

// reported rows -> those that still need the traced alignment (no trace yet: flag 1) and those that only need the walk
__global__ __launch_bounds__(256) void k_trace_flags(const u32* __restrict__ sel_idx, u32 n, const u32* __restrict__ tpos, u32* __restrict__ flags) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = tpos[sel_idx[i]] == 0xFFFFFFFFu ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_trace_split(const u32* __restrict__ sel_idx, u32 n, const u32* __restrict__ flags, const u32* __restrict__ fscan,
                                                     u32* __restrict__ list_b /*flag 1*/, u32* __restrict__ list_a /*flag 0*/) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) list_b[fscan[i]] = sel_idx[i];
    else list_a[i - fscan[i]] = sel_idx[i];
}

__global__ __launch_bounds__(64) void k_round_idx(const u32* __restrict__ tcnt, const u32* __restrict__ troff, const u32* __restrict__ toff,
                                                  const u32* __restrict__ ntask, const u32* __restrict__ ntile,
                                                  const u32* __restrict__ roffc, const u32* __restrict__ rk_slot,
                                                  const u32* __restrict__ st_state, u32 nq, u32* __restrict__ ridx) {
    const u32 q = blockIdx.x;
    const u32 n = tcnt[q];
    if (!n) return;
    const u32 base = toff[q] + rank_slot(rk_slot, roffc[q], st_state[5 * (size_t)q], ntask[q], ntile[q]), o = troff[q];
    for (u32 k = threadIdx.x; k < n; k += 64) ridx[o + k] = base + k;
}

// A WAVE per query (round 3): a thread per query walked up to ~90 ranks with three dependent loads per rank, one memory round trip
// each, for every query in lockstep (0.21 ms per round on config 3).  Here a lane loads one task of the round, all at once; the hit flags become a ballot, and the sequential rule runs over the bits of that mask in registers.  Queries with tiled (long)
// candidates -- several tasks per rank -- keep the serial walk, done by lane 0.  Also adds the round's cells to the query's counter
// (qcells; k_sum_u64 adds them up at the end: no pass over every task's result, no zero-fill of the results).
__global__ __launch_bounds__(64) void k_stop_round_w(const AlnTask* __restrict__ tasks, const AlnRes* __restrict__ res,
                                                     const u32* __restrict__ qcoff, const u32* __restrict__ ntask,
                                                     const u32* __restrict__ ntile, const u32* __restrict__ roffc,
                                                     const u32* __restrict__ rk_slot, const u32* __restrict__ toff,
                                                     const u32* __restrict__ rcnt, u32 nq, const u32* __restrict__ qoff,
                                                     const u32* __restrict__ roff, const int* __restrict__ bittab, int bittab_n, i64 D,
                                                     double expect, double max_miss, i64 v, u32* __restrict__ sel,
                                                     u32* __restrict__ st_state, int* __restrict__ bits, unsigned long long* __restrict__ qcells) {
    const u32 q = blockIdx.x, lane = threadIdx.x;
    const u32 nr = rcnt[q];
    if (!nr) return;
    u32* S = st_state + 5 * (size_t)q;
    const u32 t0 = toff[q], r0 = roffc[q], nt = ntask[q], ntl = ntile[q];
    const u32 n = qcoff[q + 1] - qcoff[q];
    double mmiss = (double)n * max_miss + 1;
    const double inv = 100. / mmiss;
    mmiss = mmiss > inv ? mmiss : inv;
    mmiss = mmiss > 10. ? mmiss : 10.;
    mmiss = mmiss < 120. ? mmiss : 120.;
    const i64 li = (i64)(qoff[q + 1] - qoff[q]);
    u32 r = S[0];
    i64 unmch = S[1], bv = S[2];
    u32 nsel = S[3];
    bool done = false;
    const u32 rend = r + nr;
    const u32 sA = rank_slot(rk_slot, r0, r, nt, ntl), sB = rank_slot(rk_slot, r0, rend, nt, ntl);
    unsigned long long cells = 0;
    bool simple = true;   // every rank of the round is exactly one task
    for (u32 k = lane; k < nr; k += 64) simple = simple && (rank_slot(rk_slot, r0, r + k + 1, nt, ntl) - rank_slot(rk_slot, r0, r + k, nt, ntl) == 1u);
    if (!__all(simple)) {
        // tiled candidates: the serial walk (lane 0); the other lanes only add up cells
        for (u32 s = sA + lane; s < sB; s += 64) cells += (unsigned long long)(u32)res[t0 + s].cells;
        if (lane == 0) {
            for (; r < rend; ++r) {
                const u32 s0 = rank_slot(rk_slot, r0, r, nt, ntl), s1 = rank_slot(rk_slot, r0, r + 1, nt, ntl);
                bool hit = false;
                for (u32 s = s0; s < s1; ++s) {  // one task, or the tiles of a long candidate (3085-3096)
                    const AlnTask tk = tasks[t0 + s];
                    const AlnRes a = res[t0 + s];
                    const i64 lj = (i64)(roff[tk.subj + 1] - roff[tk.subj]);
                    const int sc = a.maxscore < bittab_n ? a.maxscore : bittab_n - 1;
                    const int bit = bittab[sc];
                    bits[t0 + s] = bit;
                    const double p2 = bit > 1074 ? 0.0 : ldexp(1.0, -bit);
                    const double e = (double)(D * li * lj) * p2;  // bit2e (1086), full sequence lengths
                    if (e <= expect) {
                        sel[t0 + nsel++] = s;
                        hit = true;
                        bv += 1;
                    }
                }
                if (hit) unmch = 0;
                else unmch += 1;
                if ((double)unmch >= mmiss || (double)bv >= (double)v + mmiss) {
                    done = true;
                    ++r;
                    break;
                }
            }
            if (r >= nt) done = true;
            S[0] = r, S[1] = (u32)unmch, S[2] = (u32)bv, S[3] = nsel, S[4] = done ? 1u : 0u;
        }
    } else {
        // one task per rank: 64 ranks per step
        u32 covered = 0;   // ranks whose cells are counted
        for (u32 c0 = 0; c0 < nr && !done; c0 += 64) {
            covered = min(nr, c0 + 64u);
            const u32 k = c0 + lane, s = sA + k;
            bool hit = false;
            int bit = 0;
            if (k < nr) {
                const AlnTask tk = tasks[t0 + s];
                const AlnRes a = res[t0 + s];
                cells += (unsigned long long)(u32)a.cells;
                const i64 lj = (i64)(roff[tk.subj + 1] - roff[tk.subj]);
                const int sc = a.maxscore < bittab_n ? a.maxscore : bittab_n - 1;
                bit = bittab[sc];
                const double p2 = bit > 1074 ? 0.0 : ldexp(1.0, -bit);
                hit = (double)(D * li * lj) * p2 <= expect;
            }
            const unsigned long long hb = __ballot(hit);
            const u32 cnt = min(64u, nr - c0);
            u32 used = cnt;   // ranks of this step the rule looks at
            for (u32 i = 0; i < cnt; ++i) {   // (wave-uniform: registers only)
                if ((hb >> i) & 1ull) unmch = 0, bv += 1;
                else unmch += 1;
                if ((double)unmch >= mmiss || (double)bv >= (double)v + mmiss) {
                    done = true;
                    used = i + 1;
                    break;
                }
            }
            if (lane < used) {
                bits[t0 + s] = bit;
                if (hit) sel[t0 + nsel + (u32)__popcll(hb & ((1ull << lane) - 1ull))] = s;
            }
            nsel += (u32)__popcll(used >= 64 ? hb : (hb & ((1ull << used) - 1ull)));
            r += used;
        }
        // the cells of ranks behind a stop inside the round were computed too
        for (u32 k = covered + lane; k < nr; k += 64) cells += (unsigned long long)(u32)res[t0 + sA + k].cells;
        if (r >= nt) done = true;
        if (lane == 0) S[0] = r, S[1] = (u32)unmch, S[2] = (u32)bv, S[3] = nsel, S[4] = done ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    if (lane == 0 && cells) qcells[q] += cells;
}

__global__ __launch_bounds__(256) void k_sum_u64(const unsigned long long* __restrict__ x, u32 n, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_w[4];
    unsigned long long c = 0;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) c += x[i];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (c) atomicAdd(total, c);
    }
}

// qsort_u(m8s, key=-bit) (3108) + first v (3109).  A wave per query replays the reference quicksort on (inverted bit score, list
// position) words in LDS (lists of up to FSEL_CAP rows: -v 500 keeps them there); longer lists take the one-thread replay.  (One
// thread per query was 9 ms per batch on the 1 M-protein run, where every query reports its full 500 rows: 2.4 s of 121.)
#define FSEL_CAP 1024
#define FSEL_BITMAX ((1 << 20) - 1)
__global__ __launch_bounds__(64) void k_final_select_lds(const u32* __restrict__ toff, u32 nq, i64 v, u32* __restrict__ sel,
                                                         const u32* __restrict__ st_state, const int* __restrict__ bits,
                                                         u32* __restrict__ nout) {
    __shared__ u32 s_x[FSEL_CAP], s_v[FSEL_CAP];
    __shared__ u16 s_L[FSEL_CAP], s_R[FSEL_CAP];
    __shared__ int s_leaf[2 * 256];
    const u32 q = blockIdx.x;
    const u32 t0 = toff[q];
    const int n = (int)st_state[5 * (size_t)q + 3];
    if (n > FSEL_CAP) return;  // k_final_select
    bool fits = true;
    for (int i = threadIdx.x; i < n; i += 64) {
        const u32 r = sel[t0 + i];
        const int b = bits[t0 + r];
        fits = fits && b >= 0 && b <= FSEL_BITMAX;
        s_v[i] = r;
        s_x[i] = ((u32)(FSEL_BITMAX - b) << 12) | (u32)i;
    }
    // a bit score outside 20 bits cannot be packed: such a list (none exists with integer alignment scores) is left to the serial kernel
    if (!__all(fits)) return;
    __syncthreads();
    wave_ref_qsort<256>(s_x, n, [](u32 w) { return (int)(w >> 12); }, 0x7fffffff, s_L, s_R, s_leaf);
    for (int i = threadIdx.x; i < n; i += 64) sel[t0 + i] = s_v[s_x[i] & 0xFFFu];
    const i64 vv = v > 0 ? v : 0;
    if (threadIdx.x == 0) nout[q] = (i64)n < vv ? (u32)n : (u32)vv;
}

__global__ __launch_bounds__(64) void k_final_select(const u32* __restrict__ toff, u32 nq, i64 v, u32* __restrict__ sel,
                                                     const u32* __restrict__ st_state, const int* __restrict__ bits,
                                                     u32* nout, const u32* done /*== nout*/) {
    const u32 q = blockIdx.x * 64u + threadIdx.x;
    if (q >= nq || done[q] != 0xFFFFFFFFu) return;  // the wave kernel wrote this query's count
    const u32 t0 = toff[q];
    const u32 nsel = st_state[5 * (size_t)q + 3];
    const int* b = bits + t0;
    ref_qsort_dev(sel + t0, (int)nsel, [b](u32 r) { return -b[r]; });
    const i64 vv = v > 0 ? v : 0;
    nout[q] = (i64)nsel < vv ? nsel : (u32)vv;
}

// final records: 12 x i32 per reported row
// task slots of the rows that will be reported (the traced second aligner pass runs on these only)
__global__ __launch_bounds__(64) void k_selected_idx(const u32* __restrict__ toff, const u32* __restrict__ sel, const u32* __restrict__ nout,
                                                     const u32* __restrict__ ooff, u32 nq, u32* __restrict__ idx) {
    const u32 q = blockIdx.x;   // a wave per query, a lane per row (a thread per query walked its rows one dependent load at a time)
    if (q >= nq) return;
    const u32 t0 = toff[q], no = nout[q], o0 = ooff[q];
    for (u32 k = threadIdx.x; k < no; k += 64) idx[o0 + k] = t0 + sel[t0 + k];
}

__global__ __launch_bounds__(64) void k_emit_hits(const AlnTask* __restrict__ tasks, const AlnRes* __restrict__ res,
                                                  const u32* __restrict__ toff, const u32* __restrict__ sel, const u32* __restrict__ nout,
                                                  const u32* __restrict__ ooff, const int* __restrict__ bits, u32 q0, u32 nq,
                                                  int* __restrict__ out) {
    const u32 q = q0 + blockIdx.x;   // queries [q0, nq) of the batch: a wave per query, a lane per row
    if (q >= nq) return;
    const u32 t0 = toff[q], no = nout[q], o0 = ooff[q];
    for (u32 k = threadIdx.x; k < no; k += 64) {
        const u32 r = sel[t0 + k];
        const AlnTask tk = tasks[t0 + r];
        const AlnRes a = res[t0 + r];
        int* o = out + 12 * (size_t)(o0 + k);
        o[0] = (int)q, o[1] = (int)tk.subj, o[2] = a.aln, o[3] = a.aln - a.matches, o[4] = a.gap, o[5] = a.qst + 1, o[6] = a.qed,
        o[7] = a.sst + 1, o[8] = a.sed, o[9] = bits[t0 + r], o[10] = (int)tk.score, o[11] = a.matches;
    }
}

// so_hit records (include/sohit.h: 2 x i64, 2 x f64, 12 x i32 = 80 bytes) built on the device from the 12-int rows of k_emit_hits,
// for the device-resident result path (so_search_device).  Same IEEE expressions as the host emission in host_phase2.hip:
// identity = matches * (100. / aln) (fsearch.py:1458-1459, 1471), e = D * qlen * slen * 2^-bit (1086) with the powers of two from
// a table filled by libm on the host.  The library is built with -ffp-contract=off.
struct DevHit {
    long long qidx, sidx;
    double identity, evalue;
    int aln, mis, gap, qst, qed, sst, sed, bit, qlen, slen, matches, ungapped;
};
static_assert(sizeof(DevHit) == 80, "so_hit layout");

// qid != null: the batch holds its queries in length-class order (slot v[0] is query q_lo + qid[v[0]] of the file) and the records
// are written in FILE order: slot s's rows, [ooff[s], ooff[s + 1]) here, start at ostart[qid[s]] there.
__global__ __launch_bounds__(256) void k_make_hits(const int* __restrict__ rows, u32 n, long long q_lo, const u32* __restrict__ qid,
                                                   const u32* __restrict__ ooff, const u32* __restrict__ ostart, const u32* __restrict__ qoff_abs,
                                                   const u32* __restrict__ roff, long long D, const double* __restrict__ p2tab, int p2n,
                                                   DevHit* __restrict__ out) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int* v = rows + 12 * (size_t)i;
    DevHit h;
    u32 dst = i;
    if (qid) {
        const u32 s = (u32)v[0], o = qid[s];
        h.qidx = q_lo + o;
        dst = ostart[o] + (i - ooff[s]);
    } else {
        h.qidx = q_lo + v[0];
    }
    h.sidx = v[1];
    h.aln = v[2], h.mis = v[3], h.gap = v[4], h.qst = v[5], h.qed = v[6], h.sst = v[7], h.sed = v[8], h.bit = v[9];
    h.ungapped = v[10], h.matches = v[11];
    h.qlen = (int)(qoff_abs[h.qidx + 1] - qoff_abs[h.qidx]);
    h.slen = (int)(roff[h.sidx + 1] - roff[h.sidx]);
    h.identity = (double)h.matches * (100. / (double)h.aln);
    const double pw = (h.bit >= 0 && h.bit < p2n) ? p2tab[h.bit] : (h.bit < 0 ? ldexp(1.0, -h.bit) : 0.0);
    h.evalue = (double)(D * (long long)h.qlen * (long long)h.slen) * pw;
    out[dst] = h;
}

// dst[idx[i]] = src[i]
__global__ __launch_bounds__(256) void k_scatter_u32(const u32* __restrict__ src, const u32* __restrict__ idx, u32 n, u32* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}


// ---- launch wrappers -------------------------------------------------------------------------------
void launch_gather_cands(const u32* src_q, const u32* src_rec, u32 n, const u32* cqoff, const u32* prior, const u32* qcoff,
                         u32* dst_rec, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather_cands, dim3((n + 255) / 256), dim3(256), 0, st, src_q, src_rec, n, cqoff, prior, qcoff, dst_rec);
}

void launch_add_u32(u32* acc, const u32* x, u32 n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_add_u32, dim3((n + 255) / 256), dim3(256), 0, st, acc, x, n);
}

void launch_csort(const u32* rec, const u32* qcoff, u32 nq, u32 vmax, const u32* qoff, const u32* roff, u32* perm, u32* ntask,
                  u32* ntile, u64* gx, u32* gL, u32* gR, hipStream_t st, hipStream_t st_g /*k_csort's (may equal st)*/) {
    if (!nq) return;
    if (gx) hipLaunchKernelGGL(k_csort, dim3(nq), dim3(64), 0, st_g, rec, qcoff, nq, vmax, qoff, roff, perm, ntask, ntile, gx, gL, gR);
    hipLaunchKernelGGL((k_csort_lds<512, 0>), dim3(nq), dim3(64), 0, st, rec, qcoff, nq, vmax, qoff, roff, perm, ntask, ntile);
    hipLaunchKernelGGL((k_csort_lds<1024, 512>), dim3(nq), dim3(64), 0, st, rec, qcoff, nq, vmax, qoff, roff, perm, ntask, ntile);
    hipLaunchKernelGGL((k_csort_lds<2048, 1024>), dim3(nq), dim3(64), 0, st, rec, qcoff, nq, vmax, qoff, roff, perm, ntask, ntile);
    hipLaunchKernelGGL((k_csort_lds<LDS_SORT_MAX, 2048>), dim3(nq), dim3(64), 0, st, rec, qcoff, nq, vmax, qoff, roff, perm, ntask, ntile);
}

int csort_lds_max() { return LDS_SORT_MAX; }

void launch_mktasks(const u32* rec, const u32* qcoff, const u32* perm, const u32* ntask, const u32* roffc, const u32* toff, u32 nq,
                    const u32* qoff, const u32* roff, AlnTask* tasks, u32* rk_slot, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_mktasks, dim3(nq), dim3(64), 0, st, rec, qcoff, perm, ntask, roffc, toff, nq, qoff, roff, tasks, rk_slot);
}

void launch_round_counts(const u32* ntask, const u32* ntile, const u32* roffc, const u32* rk_slot, const u32* qcoff,
                         const u32* st_state, u32 nq, double max_miss, u32 minr, u32* rcnt, u32* tcnt, hipStream_t st) {
    hipLaunchKernelGGL(k_round_counts, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, ntask, ntile, roffc, rk_slot, qcoff, st_state, nq,
                       max_miss, minr, rcnt, tcnt);
}

void launch_round_counts_spec(const u32* ntask, const u32* ntile, const u32* roffc, const u32* rk_slot, const u32* qcoff, const u32* st_state, u32 nq,
                              double max_miss, u32 minr, const AlnTask* tasks, const u32* toff, const u32* qoff, const u32* roff, const int* bittab,
                              int bittab_n, i64 D, double expect, u32* rcnt, u32* tcnt_pk, u32* scnt, u32* any_rank, hipStream_t st) {
    hipLaunchKernelGGL(k_round_counts_spec, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, ntask, ntile, roffc, rk_slot, qcoff, st_state, nq, max_miss,
                       minr, tasks, toff, qoff, roff, bittab, bittab_n, D, expect, rcnt, tcnt_pk, scnt, any_rank);
}
void launch_round_idx_spec(const u32* tcnt_pk, const u32* scnt, const u32* poff, const u32* soff, const u32* toff, const u32* ntask, const u32* ntile,
                           const u32* roffc, const u32* rk_slot, const u32* st_state, u32 nq, u32* ridx, u32* sidx, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_round_idx_spec, dim3(nq), dim3(64), 0, st, tcnt_pk, scnt, poff, soff, toff, ntask, ntile, roffc, rk_slot, st_state, nq, ridx, sidx);
}
void launch_trace_flags(const u32* sel_idx, u32 n, const u32* tpos, u32* flags, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_trace_flags, dim3((n + 255) / 256), dim3(256), 0, st, sel_idx, n, tpos, flags);
}
void launch_trace_split(const u32* sel_idx, u32 n, const u32* flags, const u32* fscan, u32* list_b, u32* list_a, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_trace_split, dim3((n + 255) / 256), dim3(256), 0, st, sel_idx, n, flags, fscan, list_b, list_a);
}

void launch_round_idx(const u32* tcnt, const u32* troff, const u32* toff, const u32* ntask, const u32* ntile, const u32* roffc,
                      const u32* rk_slot, const u32* st_state, u32 nq, u32* ridx, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_round_idx, dim3(nq), dim3(64), 0, st, tcnt, troff, toff, ntask, ntile, roffc, rk_slot, st_state, nq, ridx);
}

void launch_stop_round_w(const AlnTask* tasks, const AlnRes* res, const u32* qcoff, const u32* ntask, const u32* ntile, const u32* roffc,
                         const u32* rk_slot, const u32* toff, const u32* rcnt, u32 nq, const u32* qoff, const u32* roff, const int* bittab,
                         int bittab_n, i64 D, double expect, double max_miss, i64 v, u32* sel, u32* st_state, int* bits,
                         unsigned long long* qcells, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_stop_round_w, dim3(nq), dim3(64), 0, st, tasks, res, qcoff, ntask, ntile, roffc, rk_slot, toff, rcnt, nq, qoff, roff,
                       bittab, bittab_n, D, expect, max_miss, v, sel, st_state, bits, qcells);
}

void launch_sum_u64(const unsigned long long* x, u32 n, unsigned long long* total, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_sum_u64, dim3(std::min<u32>(64u, (n + 255) / 256)), dim3(256), 0, st, x, n, total);
}

void launch_final_select(const u32* toff, u32 nq, i64 v, u32* sel, const u32* st_state, const int* bits, u32* nout, hipStream_t st) {
    if (!nq) return;
    // nout arrives filled with 0xFFFFFFFF: the wave kernel writes the count of every query it handles, the serial one takes the rest
    hipLaunchKernelGGL(k_final_select_lds, dim3(nq), dim3(64), 0, st, toff, nq, v, sel, st_state, bits, nout);
    hipLaunchKernelGGL(k_final_select, dim3((nq + 63) / 64), dim3(64), 0, st, toff, nq, v, sel, st_state, bits, nout, nout);
}

void launch_selected_idx(const u32* toff, const u32* sel, const u32* nout, const u32* ooff, u32 nq, u32* idx, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_selected_idx, dim3(nq), dim3(64), 0, st, toff, sel, nout, ooff, nq, idx);
}

void launch_emit_hits(const AlnTask* tasks, const AlnRes* res, const u32* toff, const u32* sel, const u32* nout, const u32* ooff,
                      const int* bits, u32 q0, u32 q1, int* out, hipStream_t st) {
    if (q1 <= q0) return;
    hipLaunchKernelGGL(k_emit_hits, dim3(q1 - q0), dim3(64), 0, st, tasks, res, toff, sel, nout, ooff, bits, q0, q1, out);
}


void launch_make_hits(const int* rows, u32 n, i64 q_lo, const u32* qid, const u32* ooff, const u32* ostart, const u32* qoff_abs, const u32* roff, i64 D,
                      const double* p2tab, int p2n, void* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_make_hits, dim3((n + 255) / 256), dim3(256), 0, st, rows, n, (long long)q_lo, qid, ooff, ostart, qoff_abs, roff, (long long)D,
                       p2tab, p2n, (DevHit*)out);
}

void launch_scatter_u32(const u32* src, const u32* idx, u32 n, u32* dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_scatter_u32, dim3((n + 255) / 256), dim3(256), 0, st, src, idx, n, dst);
}

// ---- alignment tasks of a launch ordered by their row count ------------------------------------------------
// k_align runs four alignments per wave: a wave lasts as long as its longest one.  Key = 8191 - R (longest first), R = rows of the
// band (see k_align); the launch list is then sorted on it.  Bit 13 of the key is clear for the tasks the packed 16-bit aligner cannot
// take -- neither  11 * min(rows, columns)  nor the smaller of the two sequences' score bounds (k_seq_bound) fits its cells -- so
// they sort in front of the others and the launch list splits into a k_align<false> part and a k_align_pk part; *n_wide counts them.
__global__ __launch_bounds__(256) void k_task_rows(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 n,
                                                   const u32* __restrict__ qoff, const u32* __restrict__ roff, const u32* __restrict__ qbound,
                                                   const u32* __restrict__ rbound, int pk_len, u32 pk_score, u32* __restrict__ n_wide,
                                                   unsigned long long* __restrict__ cells_wide, u64* __restrict__ keys) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    bool wide = false;
    u32 wcells = 0;
    if (t < n) {
        const AlnTask tk = tasks[ridx ? ridx[t] : t];
        const int lq = min((int)(qoff[tk.q + 1] - qoff[tk.q]), (int)tk.qe), ls = min((int)(roff[tk.subj + 1] - roff[tk.subj]), (int)tk.se);
        const int la = lq - min((int)tk.qi, lq), lb = ls - min((int)tk.qj, ls);
        const int ncols = min(la, lb), nrows = max(la, lb);
        if (n_wide) wide = ncols > pk_len && min(qbound[tk.q], rbound[tk.subj]) > pk_score;
        if (wide) {   // band cells of the task, as the aligners count them (lane l owns band offsets 2l and 2l + 1)
            const int R = min(nrows, ncols + 16);
            for (int l = 0; l < 16; ++l)
                wcells += (u32)(max(0, min(R, ncols + 16 - 2 * l) - max(1, 17 - 2 * l) + 1) + max(0, min(R, ncols + 15 - 2 * l) - max(1, 16 - 2 * l) + 1));
        }
        keys[t] = (u64)(8191 - min(min(nrows, ncols + 16), 8191)) | (wide ? 0ull : 8192ull);
    }
    if (n_wide) {
        const unsigned long long wb = __ballot(wide);
        if (wb) {   // (wave-uniform)
            for (int o = 32; o > 0; o >>= 1) wcells += (u32)__shfl_xor((int)wcells, o);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(n_wide, (u32)__popcll(wb));
                if (cells_wide) atomicAdd(cells_wide, (unsigned long long)wcells);
            }
        }
    }
}

void launch_task_rows(const AlnTask* tasks, const u32* ridx, u32 n, const u32* qoff, const u32* roff, const u32* qbound, const u32* rbound, int pk_len,
                      u32 pk_score, u32* n_wide, unsigned long long* cells_wide, u64* keys, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_task_rows, dim3((n + 255) / 256), dim3(256), 0, st, tasks, ridx, n, qoff, roff, qbound, rbound, pk_len, pk_score, n_wide, cells_wide,
                       keys);
}
