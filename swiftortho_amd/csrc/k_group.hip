// k_group.hip -- diagonal binning and chained ungapped X-drop extension
// (fsearch.py:2679-2719: hits dict keyed (subject, qst - sst), qsort + lis per group,
// get_ungap_scores 2497-2509, ungap 2454-2494, best diagonal per subject, guess_start 2544-2553).
//
// Input: the seed-hit keys of one (query batch, chunk) sorted ascending, so that the hits of one
// (query, subject, diagonal) group are contiguous and ordered by query position.  qsort-by-qst +
// lis-by-sst of a single-diagonal group == its distinct query positions in ascending order.
#include "common.h"
#include "kernels.h"

// ---- segment heads of a flagged list ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_list(const u32* __restrict__ flags, const u32* __restrict__ gidx, u32 H,
                                                    u32* __restrict__ ghead) {
    const u32 h = blockIdx.x * 256u + threadIdx.x;
    if (h >= H) return;
    if (flags[h]) ghead[gidx[h]] = h;
}

// ---- chained ungapped extension -----------------------------------------------------------------

__device__ __forceinline__ u64 load8u(const u8* p) {  // unaligned 8-byte global load
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

// ---- group walk + chained ungapped extension --------------------------------------------------------
// One WAVE owns the groups whose first hit lies in a fixed range of the sorted key array.  It scans
// that range 64 keys at a time for group heads (key prefix != previous key's), keeps them in a small
// LDS ring, and every lane runs a state machine
//     NEED_GROUP -> NEXT_HIT -> RIGHT pass -> LEFT pass -> NEXT_HIT -> ... -> (group done) -> NEED_GROUP
// pulling the next group as soon as it finishes one: a wave's time is (sum of its lanes' work) / 64
// instead of 64 x the slowest group, and no separate head-flag / scan / list kernels are needed.
// One loop iteration = CPI 8-residue chunks for every lane that is inside a pass (a lane whose right pass ends goes
// straight on with its left pass in the next chunk); the bookkeeping part (group fetch, hit walk) runs only when enough
// lanes wait for it.  Passing groups (score >= 25) are
// buffered in LDS and flushed to one of UG_SHARDS regions with ONE atomic per flush (wave-ballot
// compaction inside the wave).   p_qs = (q << bs) | subject_local, p_sd = (score << 32) | (u32)dist,
// p_ft = the head hit's key (or its position when ft_walk): k_first_touch turns it into the first-touch key.
#define UW_WAVES 4
#define UW_RANGE 4096   // head positions owned by one wave (large passes; launch_ungap gives the waves of a small pass shorter ranges)
#define UW_QCAP 128     // group-head ring (u32 hit indices)
#define UW_PCAP 64     // buffered pass records per wave
#define UW_WAIT 20      // run the bookkeeping part when this many lanes wait for it
#define UW_CPI 3        // 8-residue chunks per loop iteration (the right -> left hand-over sits between them)
#define UG_PIN8 (-(1 << 30))  // running word after an X-drop: below anything reachable, 8 more sentinel steps still fit int32

enum { PH_NEED = 0, PH_HIT = 1, PH_RIGHT = 2, PH_LEFT = 3, PH_DONE = 4, PH_FIN = 5 };

// BANDS (compact index addends where some subject owns several diagonal bands, k_encode_band32): the key's subject field is a band id
// and btab[band] = (chunk sequence, gbase) resolves it -- sst - qpos = gbase - (band << bd | diagonal); the pass records then carry
// the sequence id and a head key rebuilt in the record layout klr (sequence bits, diagonal + klr.diag_off).  Without BANDS klr == kl.
// GALLOP (passes of queries of 4096 residues and more): runs of seeds that start inside the previous segment are skipped with a
// doubling / halving search instead of one bookkeeping visit each.  The self hit of a 30 000-residue protein is a group of 30 000
// such seeds (45 ms per chunk for that one lane); for ordinary lengths the extra code in the bookkeeping path costs more than the
// visits it saves (measured on the 100k weight-6 set: 640 -> 699 ms with it always on), hence a template switch.
// W32 (bucketed passes, round 4): the input is not 64-bit keys but the buckets' own 32-bit hit words, band_low << (bd + bp) | diagonal
// << bp | qpos, in (band, diagonal, qpos) order inside each bucket as k_bkt_group leaves them, bit 31 set on the FIRST word of every
// bucket; the bucket (query, band range) of a word follows from its position (bext).  A group = equal word >> bp inside one bucket.
// Neither the grouping kernel writes nor this one reads the 8-byte keys (8 B per hit each way), and every per-group field is a 32-bit
// shift instead of a 64-bit one.  Records are always written in the record layout (klr's fields), as with BANDS.
template <int CPI /*chunks per loop iteration*/, bool BANDS, bool GALLOP, bool W32, bool COUNT /*count the b62 lookups (Fasta.ungap's `flag`) into stat[0]*/>
__global__ __launch_bounds__(64 * UW_WAVES, 8) void k_ungap(const u64* __restrict__ keys, const u32* __restrict__ words, const u32* __restrict__ bext, u32 nb, BktLayout L,
                                                         u32 H, KeyLayout kl, int rbs, int rsh_subj, int rsh_diag, int rdoff /*klr's fields (BANDS / W32)*/,
                                                         const uint2* __restrict__ btab, int ft_walk /*bit 1 (W32): singleton groups are k_ungap1's, skip them*/, u32 wait_n, u32 range /*head positions per wave*/,
                                                         const u8* __restrict__ q_scls, const u32* __restrict__ qoff,
                                                         const u8* __restrict__ r_scls,
                                                         const u32* __restrict__ roff /*chunk-local offsets (absolute values)*/,
                                                         const signed char* __restrict__ b62g, u32* __restrict__ shard_cnt /*[UG_SHARDS]*/,
                                                         u32 shard_cap, u64* __restrict__ p_qs, u64* __restrict__ p_sd,
                                                         u64* __restrict__ p_ft, unsigned long long* __restrict__ group_count, unsigned long long* __restrict__ stat) {
    // score table addressed by ONE v_perm per element: (query class << 8) | (subject class * 4).  Row stride 256 B; the
    // * 4 spreads the 24 subject classes over 24 LDS banks (a row stride of 64 dwords keeps the bank = column / 4).
    // 16-bit entries T = (score << 4) - 1 at byte offset (query class << 8) | (subject class * 4): adding T to the running
    // word advances (score << 4 | 15 - k) by one element in ONE add (see the chunk loop).  Row 31 (query class of the
    // elements past a pass limit) holds a byte-uniform sentinel, so whatever stray byte the subject side supplies -- at
    // whatever alignment -- reads as an immediate X-drop.
    __shared__ short s_b62[32 * 128];
    __shared__ u32 s_queue_all[UW_WAVES][UW_QCAP];   // head position | (1 << 31) when the group is a singleton
    __shared__ u64 s_qkey_all[UW_WAVES][UW_QCAP];    // the head's (masked) key
    __shared__ u64 s_pb_all[UW_WAVES][3][UW_PCAP];
    for (int i = threadIdx.x; i < 32 * 128; i += 64 * UW_WAVES) {
        const int a = i >> 7, b = (i & 127) >> 1;  // b: subject class of byte column 2 * (i & 127) (the slots between the * 4
                                                   // columns are never read by in-sequence bytes)
        const int v = (a < SCLS_N && b < SCLS_N) ? (int)b62g[a * SCLS_N + b] : -4;
        s_b62[i] = (a == 31) ? (short)0xF7F7 /* -2057: far below the X-drop threshold */ : (short)((v << 4) - 1);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 wid = blockIdx.x * UW_WAVES + w;
    const u64 ra = (u64)wid * range;
    if (ra >= H) return;
    const u32 a0 = (u32)ra, b0 = (u32)min((u64)H, ra + range);
    u32* s_queue = s_queue_all[w];
    u64* s_qkey = s_qkey_all[w];
    u32* s_qw = reinterpret_cast<u32*>(s_qkey_all[w]);   // W32: the head's word ...
    u32* s_qr = s_qw + UW_QCAP;                          // ... and its bucket: query << 10 | band range
    u64* s_pqs = s_pb_all[w][0];
    u64* s_psd = s_pb_all[w][1];
    u64* s_pft = s_pb_all[w][2];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const u64 kmask = (kl.total >= 64) ? ~0ull : ((1ull << kl.total) - 1ull);
    const u64 qall = (1ull << kl.bq) - 1ull;
    const u64 pmask = (1ull << kl.bp) - 1ull;
    const u32 shard = blockIdx.x & (UG_SHARDS - 1);
    const bool skip_single = (ft_walk & 2) != 0;
    ft_walk &= 1;
    const u8* q_m16 = q_scls - 16;  // both class arrays have 16 readable bytes in front (left windows start up to 8 bytes early)
    const u8* r_m16 = r_scls - 16;

    // W32: the bucket the scan position lies in (wave-uniform): id, its end, its query (relative to the pass) and band range
    const u32 WM = 0x7FFFFFFFu;
    const u32 pmask32 = (1u << L.bp) - 1u;
    const int gb = L.wb + L.bd;
    u32 bcur = 0, bend = 0, bqrel = 0, brange = 0;
    if (W32) {
        u32 lo_b = 0, hi_b = nb;   // largest b with bext[b] <= a0; a0 < H = bext[nb], so that bucket is not empty
        while (hi_b - lo_b > 1) {
            const u32 m = (lo_b + hi_b) >> 1;
            if (bext[m] <= a0) lo_b = m;
            else hi_b = m;
        }
        bcur = __builtin_amdgcn_readfirstlane(lo_b);
        bend = __builtin_amdgcn_readfirstlane(bext[bcur + 1]);
        brange = bcur / L.nqp;
        bqrel = bcur - brange * L.nqp;
    }
    // wave-uniform bookkeeping
    u32 cur = a0;              // next position to scan for heads
    u32 qfront = 0, qback = 0;  // ring counters
    u32 npb = 0;               // buffered pass records
    u32 ngroups = 0;
    unsigned long long nst = 0;   // COUNT: elements scored by this lane

    // lane state
    int phase = PH_NEED;
    u32 h = 0;
    u64 gpre = 0;
    u32 gq = 0, gsubj = 0;
    // All hits of a group lie on one diagonal, so the subject coordinate is always (query coordinate + dlt):
    //   dlt = sst - qst;  sbd = subject offset + dlt (subject byte of query position p is r_scls[sbd + p]);
    //   lim = min(ql, sl - dlt): first query coordinate past either sequence;
    //   lo  = lower bound of the next segment in query coordinates: max(0, -dlt) before the first segment
    //         (qlo = slo = 0, fsearch.py:2455-2464), then the previous segment's right end.
    int dlt = 0, sbd = 0, lim = 0, lo = 0;
    u32 qb = 0;
    int prev_qpos = -1, scores = 0;
    bool single = false, havekey = false;
    u64 hkey = 0;  // the head hit's key
    u32 hw = 0, gw = 0;  // W32: the head hit's word, and the group's word >> bp
    u32 h0 = 0;    // the head hit's position (ft_walk mode)
    u32 cq = 0xFFFFFFFFu, cqb = 0;  // last query looked up (groups arrive sorted by query)
    int cql = 0;
    int Qst = 0, ci = 0, cn = 0, score8 = 16 /*running word: (score << 4) + position nibble*/, mp = 15 /*packed running maximum*/, best = -1, r_qed = 0;
    int qcur = 0, scur = 0;  // byte offsets of the next chunk in the two class arrays
    int dstep = 8;           // +8 on the right pass, -8 on the left pass
    u32 selA = 0x03020100u, selB = 0x07060504u;  // v_perm selectors of the low / high half of a chunk in element order
    bool stop = false;

    for (;;) {
        const unsigned long long waitb = __ballot(phase == PH_NEED || phase == PH_HIT);
        const unsigned long long workb = __ballot(phase == PH_RIGHT || phase == PH_LEFT);
        if (!waitb && !workb) break;
        if (waitb && (!workb || (u32)__popcll(waitb) >= wait_n)) {
            // ---- refill the head ring -----------------------------------------------------------------
            while (W32 && qback - qfront < 64u && cur < b0) {
                const u32 pos = cur + (u32)lane;
                const bool valid = pos < b0;
                u32 wv = pos < H ? words[pos] : 0x80000000u;   // (past the end: reads as the start of another bucket)
                u32 wp = (u32)__shfl_up((int)wv, 1);
                if (lane == 0) wp = cur == 0 ? 0u : words[cur - 1];
                u32 wn = (u32)__shfl_down((int)wv, 1);
                if (skip_single && lane == 63) wn = pos + 1u < H ? words[pos + 1u] : 0x80000000u;   // (exact: both kernels must agree on what a singleton is)
                bool head = valid && ((wv >> 31) != 0 || ((wv & WM) >> L.bp) != ((wp & WM) >> L.bp));
                const bool sing = (lane < 63 || skip_single) && ((wn >> 31) != 0 || ((wn & WM) >> L.bp) != ((wv & WM) >> L.bp));
                if (skip_single && sing) head = false;
                // buckets of the 64 positions: the scan's bucket, except behind a bucket start other than its own (rare: a bucket holds
                // a thousand hits or more) -- those starts are walked one by one, skipping empty buckets
                u32 lqr = ((L.qa + bqrel) << 10) | brange;
                unsigned long long mb = __ballot(valid && (wv >> 31) != 0);
                while (mb) {
                    const int ml = (int)__builtin_ctzll(mb);
                    mb &= mb - 1ull;
                    if (cur + (u32)ml >= bend) {   // (== bend: the next non-empty bucket starts here)
                        do {
                            ++bcur;
                            if (++bqrel == L.nqp) bqrel = 0, ++brange;
                        } while (__builtin_amdgcn_readfirstlane(bext[bcur + 1]) <= cur + (u32)ml);
                        bend = __builtin_amdgcn_readfirstlane(bext[bcur + 1]);
                        if (lane >= ml) lqr = ((L.qa + bqrel) << 10) | brange;
                    }
                }
                const unsigned long long hb = __ballot(head);
                if (head) {
                    const u32 slot = (qback + (u32)__popcll(hb & lt)) & (UW_QCAP - 1);
                    s_queue[slot] = pos | (sing ? 0x80000000u : 0u);
                    s_qw[slot] = wv & WM;
                    s_qr[slot] = lqr;
                }
                qback += (u32)__popcll(hb);
                cur += 64;
            }
            while (!W32 && qback - qfront < 64u && cur < b0) {
                const u32 pos = cur + (u32)lane;
                u64 k = ~0ull, kp = ~0ull;
                if (pos < H) k = keys[pos] & kmask;
                kp = __shfl_up(k, 1);
                if (lane == 0) kp = (cur == 0) ? ~0ull : (keys[cur - 1] & kmask);
                const bool valid = (pos < b0) && ((k >> kl.sh_q) != qall);
                const bool head = valid && (cur + (u32)lane == 0 || (k >> kl.sh_diag) != (kp >> kl.sh_diag));
                // singleton: the next key starts another group (lane 63 would need one more load: treated as unknown)
                const u64 kn = __shfl_down(k, 1);
                const bool sing = (lane < 63) && ((kn >> kl.sh_diag) != (k >> kl.sh_diag));
                const unsigned long long hb = __ballot(head);
                if (head) {
                    const u32 slot = (qback + (u32)__popcll(hb & lt)) & (UW_QCAP - 1);
                    s_queue[slot] = pos | (sing ? 0x80000000u : 0u);
                    s_qkey[slot] = k;
                }
                qback += (u32)__popcll(hb);
                cur += 64;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- finished groups: buffer the passing ones ----------------------------------------------
            // (a lane in PH_HIT discovers the end of its group below; results are buffered one iteration later)
            // ---- hand out groups ------------------------------------------------------------------------
            {
                const unsigned long long needb = __ballot(phase == PH_NEED);
                const u32 avail = qback - qfront;
                if (phase == PH_NEED) {
                    const u32 r = (u32)__popcll(needb & lt);
                    if (r < avail) {
                        const u32 qe = s_queue[(qfront + r) & (UW_QCAP - 1)];
                        u64 k0 = 0;
                        h = qe & 0x7FFFFFFFu;
                        h0 = h;
                        single = (qe >> 31) != 0;
                        havekey = true;
                        if (W32) {
                            hw = s_qw[(qfront + r) & (UW_QCAP - 1)];
                            const u32 qr = s_qr[(qfront + r) & (UW_QCAP - 1)];
                            gw = hw >> L.bp;
                            gq = qr >> 10;
                            const u32 G = ((qr & 1023u) << gb) | gw;
                            if (BANDS) {
                                const uint2 be = btab[G >> L.bd];
                                gsubj = be.x;
                                dlt = (int)(be.y - G);  // sst - qpos
                            } else {
                                gsubj = G >> L.bd;
                                dlt = (int)kl.diag_off - (int)(G & ((1u << L.bd) - 1u));
                            }
                        } else {
                            k0 = s_qkey[(qfront + r) & (UW_QCAP - 1)];
                            hkey = k0;
                            gpre = k0 >> kl.sh_diag;
                            gq = (u32)((k0 >> kl.sh_q) & qall);
                        }
                        if (W32) {
                        } else if (BANDS) {
                            const u32 G = (u32)((k0 >> kl.sh_diag) & ((1ull << (kl.bs + kl.bd)) - 1ull));
                            const uint2 be = btab[G >> kl.bd];
                            gsubj = be.x;
                            dlt = (int)(be.y - G);  // sst - qpos
                        } else {
                            gsubj = (u32)((k0 >> kl.sh_subj) & ((1ull << kl.bs) - 1ull));
                            dlt = (int)(kl.diag_off - (i64)((k0 >> kl.sh_diag) & ((1ull << kl.bd) - 1ull)));  // sst - qpos
                        }
                        if (gq != cq) cq = gq, cqb = qoff[gq], cql = (int)(qoff[gq + 1] - cqb);
                        qb = cqb;
                        const u32 sb = roff[gsubj];
                        const int sl = (int)(roff[gsubj + 1] - sb);
                        sbd = (int)sb + dlt;
                        lim = min(cql, sl - dlt);
                        lo = max(0, -dlt);
                        prev_qpos = -1, scores = 0;
                        phase = PH_HIT;
                    } else if (cur >= b0) {
                        phase = PH_DONE;  // nothing left to hand out
                    }
                }
                const u32 taken = min((u32)__popcll(needb), avail);
                qfront += taken;
                ngroups += taken;
            }
            // ---- walk to the next distinct seed of the group, or finish the group --------------------------
            bool fin = false;
            if (phase == PH_HIT) {
                bool in_group = false;
                u64 k = 0;
                u32 kw = 0;
                if (havekey) {
                    k = hkey, kw = hw, in_group = true, havekey = false;  // the head's key came with the hand-out
                } else if (h < H) {
                    if (W32) {
                        kw = words[h];
                        in_group = (kw >> 31) == 0 && (kw >> L.bp) == gw;
                    } else {
                        k = keys[h] & kmask;
                        in_group = (k >> kl.sh_diag) == gpre;
                    }
                }
                if (!in_group) {
                    fin = true;
                } else {
                    const int qpos = W32 ? (int)(kw & pmask32) : (int)((k >> kl.sh_qpos) & pmask);
                    if (qpos == prev_qpos) {
                        ++h;  // duplicate (qst, sst) pair: dropped by lis()
                    } else if (GALLOP && prev_qpos >= 0 && qpos <= lo) {
                        // A later seed that starts inside the previous segment: off = lo - qpos moves its start to lo, both passes get
                        // zero steps (2460-2476: `qlo < qst` fails) and it adds nothing -- lo, the score and the group's ends stay as
                        // they are.  Such seeds come in runs (a homolog's diagonal is covered by its first extension; the self hit of a
                        // 30 000-residue protein is 30 000 of them): gallop to the first hit of the group that is not covered.
                        ++h;
                        // (one loop for both phases of the search: first doubling steps while the probed hit is covered, then halving
                        // them inside [h, h + step - 1], whose last hit is not)
                        u32 step = 1;
                        bool grow = true;
                        for (;;) {
                            if (!grow) {
                                if (step <= 1u) break;
                                step >>= 1;
                            }
                            const u32 t = h + step - 1u;
                            bool cv = false;
                            if (t < H) {
                                if (W32) {
                                    const u32 kk = words[t];
                                    cv = (kk >> 31) == 0 && (kk >> L.bp) == gw && (int)(kk & pmask32) <= lo;
                                } else {
                                    const u64 kk = keys[t] & kmask;
                                    cv = (kk >> kl.sh_diag) == gpre && (int)((kk >> kl.sh_qpos) & pmask) <= lo;
                                }
                            }
                            if (cv) h += step;
                            if (grow) {
                                if (cv) step <<= 1;
                                else grow = false;
                            }
                        }
                    } else {
                        prev_qpos = qpos;
                        // Fasta.ungap set-up (2455-2464): first seed unbounded, later ones bounded by the previous segment's end;
                        // off = max(qlo - qst, slo - sst, 0) and both differences are equal on a diagonal
                        Qst = qpos + max(lo - qpos, 0);
                        cn = (lo < Qst) ? lim - Qst : 0;  // qlo < qst and slo < sst; min(ql - qst, sl - sst) steps
                        ci = 0, score8 = 16, mp = 15, best = -1, stop = false;
                        qcur = (int)qb + Qst + 16, scur = sbd + Qst + 16;  // offsets from (array - 16): never negative
                        dstep = 8, selA = 0x03020100u, selB = 0x07060504u;
                        phase = PH_RIGHT;
                    }
                }
            }
            if (fin) phase = PH_FIN;
        }
        // ---- one 8-residue chunk for every lane inside a pass ---------------------------------------------------
        // Branch-free: one load path for both directions (left-pass windows are byte-reversed), elements past
        // the pass limit get class 31 (scores -128 against everything: an immediate X-drop), and after a
        // drop the running score is pinned far below zero so later elements can neither raise the maximum
        // nor matter -- exactly the sequential loop's `break`.
        if (phase == PH_RIGHT || phase == PH_LEFT) {
#pragma unroll
          for (int rep = 0; rep < CPI; ++rep) {
            if (ci < cn && !stop) {
                // right: bytes [Qst + ci, +8), element k in byte k;  left: bytes [Qst - 8 - ci, +8), element k in byte 7 - k
                // (qcur / scur walk by +-8 per chunk).  A left window may start up to 8 bytes before its sequence (the arrays
                // have 16 readable bytes in front): those elements lie past the pass limit and are never active.
                const u64 q0 = load8u(q_m16 + (u32)qcur), s0 = load8u(r_m16 + (u32)scur);  // scalar base + 32-bit offset
                qcur += dstep;
                scur += dstep;
                // element order: one v_perm per 32-bit half with the pass's selectors (identity on the right pass, byte
                // reversal across the two halves on the left pass)
                u64 qw = (u64)__builtin_amdgcn_perm((u32)(q0 >> 32), (u32)q0, selA) | ((u64)__builtin_amdgcn_perm((u32)(q0 >> 32), (u32)q0, selB) << 32);
                u64 sw = (u64)__builtin_amdgcn_perm((u32)(s0 >> 32), (u32)s0, selA) | ((u64)__builtin_amdgcn_perm((u32)(s0 >> 32), (u32)s0, selB) << 32);
                const int m = cn - ci;
                if (m < 8) qw = (qw & ~(~0ull << (8 * m))) | (0x1F1F1F1F1F1F1F1Full << (8 * m));  // elements >= m: exactly class 31 (row 31 = -128)
                int sc[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const u32 qh = (u32)(k < 4 ? qw : qw >> 32), sh = (u32)(k < 4 ? sw : sw >> 32);
                    // byte0 = subject byte k, byte1 = query byte k, bytes 2-3 = 0   (selector 0x0c = constant 0x00)
                    const u32 boff = __builtin_amdgcn_perm(qh, sh, 0x0c0c0400u + (u32)(k & 3) * 0x0101u);
                    sc[k] = *reinterpret_cast<const short*>(reinterpret_cast<const char*>(s_b62) + boff);
                }
                // The running word R carries (score << 4) + (15 - k) after element k: score and position of the element in one
                // word, advanced by ONE add of the table entry (score << 4) - 1 (the chunk starts at (score + 1) << 4).  The
                // running maximum mp keeps the largest R seen: a later equal score has a smaller low nibble and loses, the
                // incoming maximum carries 15 and wins every tie -- the FIRST position of the maximum, as the reference's
                // strict `>`.  With d = mp - R = ((max - ns) << 4) + (nibble difference in [0, 7]) the X-drop test
                // ns + 30 < max  is  d >= (31 << 4) - 7.  Five VALU per element.
                const int mp_in = mp;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (COUNT) nst += (score8 > (UG_PIN8 >> 1) && k < m) ? 1u : 0u;   // the pass is alive and the element inside its limits
                    score8 += sc[k];
                    const bool drop = mp - score8 >= ((DROPX + 1) << 4) - 7;
                    mp = max(mp, score8);
                    score8 = drop ? UG_PIN8 : score8;
                }
                if (mp != mp_in) best = ci + (15 - (mp & 15));
                mp |= 15;
                stop = score8 < (UG_PIN8 >> 1);
                score8 += 8;  // low nibble 8 -> (score + 1) << 4 for the next chunk
                ci += 8;
            }
            if (phase == PH_RIGHT && (ci >= cn || stop)) {
                r_qed = Qst + max(best, 0);
                // left pass from (Qst - 1, Sst - 1); the score continues from the maximum (2479-2492)
                score8 = (mp & ~15) + 16;
                stop = false, best = -1, ci = 0;
                qcur = (int)qb + Qst + 8, scur = sbd + Qst + 8;  // (Qst - 8) + 16
                dstep = -8, selA = 0x04050607u, selB = 0x00010203u;
                cn = (Qst - 1 < lim) ? Qst - 1 - lo : 0;  // min(qst - 1 - qlo, sst - 1 - slo) steps
                phase = PH_LEFT;
            }
          }
            if (phase == PH_LEFT && (ci >= cn || stop)) {
                // get_ungap_scores (2497-2509): segment maxima add up; the next seed is bounded by this segment's right end.
                // The left end (start of the first segment) only enters guess_start, which needs no end point at all: every
                // segment of a group lies on the group's diagonal, so (sst0 - qst0) + (sed - qed) = 2 * (sst - qst).
                scores += mp >> 4;
                lo = r_qed;
                ++h;
                phase = single ? PH_FIN : PH_HIT;  // a singleton group is complete: no second visit to find its end
            }
        }
        // ---- buffer the groups that just finished with score >= 25 (wave-ballot compaction into LDS) ------------
        {
            const bool fin = phase == PH_FIN;
            const bool pass = fin && scores >= MIN_UNGAP;
            const unsigned long long pb = __ballot(pass);
            if (pb) {
                const u32 np = (u32)__popcll(pb);
                if (npb + np > UW_PCAP) {  // flush (wave-uniform)
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&shard_cnt[shard], npb);
                    base = __shfl(base, 0);
                    for (u32 i = (u32)lane; i < npb; i += 64) {
                        const size_t o = (size_t)shard * shard_cap + base + i;
                        p_qs[o] = s_pqs[i], p_sd[o] = s_psd[i], p_ft[o] = s_pft[i];
                    }
                    npb = 0;
                    __builtin_amdgcn_wave_barrier();
                }
                if (pass) {
                    const int dist = dlt;  // guess_start (2544-2553): floor(2 * (sst - qst) / 2) = the diagonal
                    const u32 i = npb + (u32)__popcll(pb & lt);
                    s_pqs[i] = ((u64)gq << ((BANDS || W32) ? rbs : kl.bs)) | gsubj;
                    s_psd[i] = ((u64)(u32)scores << 32) | (u64)(u32)dist;
                    // k_first_touch / k_rec_scatter turn this into the first-touch key
                    u64 hk = hkey;
                    if (W32) hk = ((u64)gsubj << rsh_subj) | ((u64)(u32)(rdoff - dlt) << rsh_diag) | ((u64)(hw & pmask32) << kl.sh_qpos);
                    else if (BANDS) hk = ((u64)gsubj << rsh_subj) | ((u64)(u32)(rdoff - dlt) << rsh_diag) | (((hkey >> kl.sh_qpos) & pmask) << kl.sh_qpos);
                    s_pft[i] = ft_walk ? (u64)h0 : hk;
                }
                npb += np;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (fin) phase = PH_NEED;
        }
    }
    if (npb) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&shard_cnt[shard], npb);
        base = __shfl(base, 0);
        for (u32 i = (u32)lane; i < npb; i += 64) {
            const size_t o = (size_t)shard * shard_cap + base + i;
            p_qs[o] = s_pqs[i], p_sd[o] = s_psd[i], p_ft[o] = s_pft[i];
        }
    }
    if (lane == 0 && ngroups) atomicAdd(&group_count[shard], (unsigned long long)ngroups);
    if (COUNT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nst += __shfl_xor(nst, o);
        if (lane == 0 && nst) atomicAdd(&stat[0], nst);
    }
}

// ---- first-touch keys of the passing groups ----------------------------------------------------------
// Group visiting order, "first best diagonal wins" and the candidate order are functions of the minimum, over a
// group's hits, of (emission order (as, qpos), then index slot order == descending (j, tag, pos)).  Only the passing
// groups (a few percent) need it, so it is computed here, one thread per pass record, instead of inside the extension
// kernel for every hit.  With one alphabet x one seed pattern (as == 0 everywhere) the minimum is the head hit's (the
// hits of a group are ordered by query position) and the record carries that key; otherwise (WALK) it carries the
// head's position and the group's hits are read back from the sorted key array.
template <bool WALK>
__global__ __launch_bounds__(256) void k_first_touch(const u64* __restrict__ keys, u32 H, KeyLayout kl, int ft_bits_entry, int bsp,
                                                     const u32* __restrict__ roff, u64* __restrict__ p_ft, u32 n) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u64 kmask = (kl.total >= 64) ? ~0ull : ((1ull << kl.total) - 1ull);
    const u64 pmask = (1ull << kl.bp) - 1ull, amask = (1ull << kl.ba) - 1ull;
    const u64 jmax = (1ull << (kl.bs + 1)) - 1ull, pmax = (1ull << bsp) - 1ull;
    u32 h = WALK ? (u32)p_ft[i] : 0u;
    const u64 k0 = WALK ? (keys[h] & kmask) : p_ft[i];
    const u64 gpre = k0 >> kl.sh_diag;
    const u32 gsubj = (u32)((k0 >> kl.sh_subj) & ((1ull << kl.bs) - 1ull));
    const i64 gdiag = (i64)((k0 >> kl.sh_diag) & ((1ull << kl.bd) - 1ull)) - kl.diag_off;  // qpos - sst
    const int sl = (int)(roff[gsubj + 1] - roff[gsubj]);
    u64 ft = ~0ull;
    for (u64 k = k0;;) {
        const int qpos = (int)((k >> kl.sh_qpos) & pmask);
        const u32 as = kl.ba ? (u32)((k >> kl.sh_as) & amask) : 0u;
        const u32 tag = kl.ba ? (u32)(k & amask) : 0u;
        const int sst = (int)((i64)qpos - gdiag);
        u32 j = gsubj, pos = (u32)sst;
        if (sst == sl) j = gsubj + 1, pos = 0;  // offset-0 entry of the next chunk sequence
        const u64 inv = ((jmax - j) << (kl.ba + bsp)) | ((amask - tag) << bsp) | (pmax - pos);
        const u64 f = ((((u64)as << kl.bp) | (u64)qpos) << ft_bits_entry) | inv;
        ft = f < ft ? f : ft;
        if (!WALK) break;
        if (++h >= H) break;
        k = keys[h] & kmask;
        if ((k >> kl.sh_diag) != gpre) break;
    }
    p_ft[i] = ft;
}

void launch_first_touch(bool walk, const u64* keys, u32 H, const KeyLayout& kl, int ft_bits_entry, int bsp, const u32* roff, u64* p_ft, u32 n,
                        hipStream_t st) {
    if (!n) return;
    if (walk) hipLaunchKernelGGL((k_first_touch<true>), dim3((n + 255) / 256), dim3(256), 0, st, keys, H, kl, ft_bits_entry, bsp, roff, p_ft, n);
    else hipLaunchKernelGGL((k_first_touch<false>), dim3((n + 255) / 256), dim3(256), 0, st, keys, H, kl, ft_bits_entry, bsp, roff, p_ft, n);
}

// shard offsets (exclusive scan over UG_SHARDS counters) + total
__global__ void k_shard_scan(const u32* __restrict__ shard_cnt, u32* __restrict__ shard_off /*[UG_SHARDS + 1]*/) {
    if (threadIdx.x == 0) {
        u32 s = 0;
        for (int k = 0; k < UG_SHARDS; ++k) {
            shard_off[k] = s;
            s += shard_cnt[k];
        }
        shard_off[UG_SHARDS] = s;
    }
}

__global__ __launch_bounds__(256) void k_compact_shards(const u32* __restrict__ shard_cnt, const u32* __restrict__ shard_off, u32 shard_cap,
                                                        const u64* __restrict__ a0, const u64* __restrict__ a1, const u64* __restrict__ a2,
                                                        u64* __restrict__ b0, u64* __restrict__ b1, u64* __restrict__ b2) {
    const u32 shard = blockIdx.y;
    const u32 n = shard_cnt[shard], o = shard_off[shard];
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const size_t s = (size_t)shard * shard_cap + i;
        b0[o + i] = a0[s], b1[o + i] = a1[s], b2[o + i] = a2[s];
    }
}

// ---- best diagonal per (query, subject) -----------------------------------------------------------
// pass records sorted by p_qs (idx = permutation).  Segment heads -> one candidate per segment:
// best = max score, ties -> smallest first-touch key (first visited wins, strict `>` at 2709);
// the candidate's order key = smallest first-touch key among the segment's passing groups.
__global__ __launch_bounds__(256) void k_seg_flags(const u64* __restrict__ sorted_qs, u32 n, u32* __restrict__ flags, u32* __restrict__ zero_word) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (i == 0 && zero_word) *zero_word = 0;  // k_qseg's running maximum
    flags[i] = (i == 0 || sorted_qs[i] != sorted_qs[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_best(const u64* __restrict__ sorted_qs, const u32* __restrict__ idx, const u32* __restrict__ shead,
                                              u32 nseg, u32 n, const u64* __restrict__ p_sd, const u64* __restrict__ p_ft, u32 seq_lo,
                                              int bs, u64* __restrict__ c_ft, u32* __restrict__ c_q, u32* __restrict__ c_rec /*4 per cand*/) {
    const u32 s = blockIdx.x * 256u + threadIdx.x;
    if (s >= nseg) return;
    const u32 i0 = shead[s], i1 = (s + 1 < nseg) ? shead[s + 1] : n;
    u64 minft = ~0ull, bft = ~0ull;
    u32 bscore = 0;
    int bdist = 0;
    for (u32 i = i0; i < i1; ++i) {
        const u32 r = idx[i];
        const u64 sd = p_sd[r], ft = p_ft[r];
        const u32 sc = (u32)(sd >> 32);
        minft = ft < minft ? ft : minft;
        if (sc > bscore || (sc == bscore && ft < bft)) bscore = sc, bft = ft, bdist = (int)(u32)sd;
    }
    const u64 qs = sorted_qs[i0];
    c_ft[s] = minft;
    c_q[s] = (u32)(qs >> bs);
    u32 qi, qj;
    if (bdist > 0) qi = 0, qj = (u32)bdist;
    else qi = (u32)(-bdist), qj = 0;
    // [global subject id, ungapped score, qi, qj]: one 16-byte store
    *reinterpret_cast<uint4*>(c_rec + 4 * (size_t)s) = make_uint4(((u32)qs & ((1u << bs) - 1u)) + seq_lo, bscore, qi, qj);
}

// gather helpers for the two-pass (ft, then stable q) ordering
__global__ __launch_bounds__(256) void k_gather_u32_as_u64(const u32* __restrict__ src, const u32* __restrict__ idx, u32 n,
                                                           u64* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// Sort word of a candidate: (query, first-touch key) with the key's low `bsp` bits -- the inverted subject offset --
// reduced to ONE bit.  Two candidates of one query are different subjects, so their keys can only agree down to
// (as, qpos, j, tag) when one of them is the offset-0 quirk entry (attributed to j = subject + 1 with pos = 0, the
// all-ones inverted offset) and the other a real entry of subject j (pos >= 1): the real one is visited first.
// One radix pass fewer than sorting the full key.
__global__ __launch_bounds__(256) void k_combine_q_ft(const u32* __restrict__ c_q, const u64* __restrict__ c_ft, u32 n, int ftbits, int bsp,
                                                      u64* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u64 ft = c_ft[i], pm = (1ull << bsp) - 1ull;
    dst[i] = ((u64)c_q[i] << (ftbits - bsp + 1)) | ((ft >> bsp) << 1) | ((ft & pm) == pm ? 1ull : 0ull);
}

__global__ __launch_bounds__(256) void k_iota(u32* __restrict__ p, u32 n) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = i;
}

// final order -> chunk candidate region; the region is sorted by query, so per-query counts are
// segment lengths: the thread at a segment's last element knows them without atomics.
__global__ __launch_bounds__(256) void k_emit_cands(const u32* __restrict__ order, u32 n, const u64* __restrict__ sorted_key, int qshift,
                                                    const u32* __restrict__ c_rec, u32* __restrict__ out_q, u32* __restrict__ out_rec,
                                                    u32* __restrict__ seg_first) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u32 r = order[i];
    const u32 q = (u32)(sorted_key[i] >> qshift);  // the query sits on top of the sort word: read in order, not gathered
    out_q[i] = q;
    const uint4 v = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)r);
    *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)i) = v;
    if (i == 0 || (u32)(sorted_key[i - 1] >> qshift) != q) seg_first[q] = i;
}

__global__ __launch_bounds__(256) void k_seg_counts(const u32* __restrict__ out_q, u32 n, const u32* __restrict__ seg_first,
                                                    u32* __restrict__ qcnt) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u32 q = out_q[i];
    if (i + 1 == n || out_q[i + 1] != q) qcnt[q] = i + 1 - seg_first[q];
}

// ---- sparse path: a query's candidates ordered inside LDS ------------------------------------------------------
// k_best leaves the candidates (query, subject)-major, so the (query, first-touch) order the reference's spill file has is a
// sort of the first-touch word INSIDE each query's segment: tens to hundreds of candidates per query and pass on the sparse
// path.  k_qseg finds the segments from the sorted pass records (lower bound of q << bs; the exclusive scan of the head flags at
// that record = the query's first candidate) together with the longest one, so the host -- which fetches the candidate total with
// the same copy -- knows whether every segment fits the LDS sort; k_cand_order_lds then does, in ONE launch, what the library
// path needs an index fill, a key build, a 7-pass device-wide radix sort, a gather and a count kernel for.
__global__ __launch_bounds__(256) void k_qseg(const u64* __restrict__ sorted_qs, u32 n, const u32* __restrict__ gidx, const u32* __restrict__ total,
                                              int bs, u32 q0, u32 nq /*queries [q0, nq) of the batch: the pass's*/, u32* __restrict__ seg /*entries [q0, nq]*/,
                                              u32* __restrict__ maxseg) {
    const u32 q = q0 + blockIdx.x * 256u + threadIdx.x;
    auto first_cand = [&](u32 qq) -> u32 {
        const u64 key = (u64)qq << bs;
        u32 lo = 0, hi = n;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (sorted_qs[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        return lo < n ? gidx[lo] : *total;
    };
    u32 len = 0;
    if (q <= nq) {
        const u32 a = first_cand(q);
        seg[q] = a;
        if (q < nq) len = first_cand(q + 1) - a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, (u32)__shfl_xor((int)len, o));
    if ((threadIdx.x & 63) == 0 && len) atomicMax(maxseg, len);
}

// One wave per query: sort word = (first-touch word << LB) | position in the segment (unique: ties keep the (query, subject) order,
// as the stable library sort does), bitonic network over the next power of two, then the records are gathered in that order.
template <int CAP, int LB>
__global__ __launch_bounds__(64) void k_cand_order_lds(const u64* __restrict__ c_ft, const u32* __restrict__ c_rec, const u32* __restrict__ seg, u32 q0,
                                                       int bsp, u32* __restrict__ out_q, u32* __restrict__ out_rec, u32* __restrict__ qcnt) {
    __shared__ u64 s_key[CAP];
    const u32 q = q0 + blockIdx.x, lane = threadIdx.x;
    const u32 a = seg[q], n = seg[q + 1] - a;
    if (!n) return;
    if (lane == 0) qcnt[q] = n;
    if (n == 1) {
        if (lane == 0) {
            out_q[a] = q;
            *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)a) = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)a);
        }
        return;
    }
    u32 P = 2;
    while (P < n) P <<= 1;
    const u64 pm = (1ull << bsp) - 1ull;
    for (u32 i = lane; i < P; i += 64) {
        u64 k = ~0ull;
        if (i < n) {
            const u64 ft = c_ft[a + i];
            k = ((((ft >> bsp) << 1) | ((ft & pm) == pm ? 1ull : 0ull)) << LB) | (u64)i;
        }
        s_key[i] = k;
    }
    __syncthreads();
    for (u32 k = 2; k <= P; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 t = lane; t < (P >> 1); t += 64) {
                const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const u64 x = s_key[i], y = s_key[l];
                if ((x > y) == ((i & k) == 0)) s_key[i] = y, s_key[l] = x;
            }
            __syncthreads();
        }
    for (u32 i = lane; i < n; i += 64) {
        const u32 r = (u32)s_key[i] & ((1u << LB) - 1u);
        out_q[a + i] = q;
        *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)(a + i)) = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)(a + r));
    }
}

// ---- sparse path, round 5: best diagonal per subject AND candidate order of a pass, per query, in LDS --------------------------
// The pass records of a sparse pass (tens per query) used to be sorted device-wide by (query, subject) -- a 7-launch library radix sort of
// all records --, flagged, scanned, reduced (k_best) and then ordered per query (k_cand_order_lds).  Here the records are counted per query
// (k_rec_count with one bucket per query: the returning atomic is the record's rank), scattered to their query's segment
// (k_qrec_scatter), and ONE wave per query does the rest: a bitonic sort of (subject, position) words, a reduction per run of equal
// subjects -- best = max score, ties to the smaller first-touch key; order key = the run's smallest first-touch key (k_best's rules) --
// and a second bitonic sort of the candidates by (first-touch word, position in subject order) (k_cand_order_lds's word).  The
// candidates' records go to a scratch at the segment's start, their order to `perm`; k_q_emit copies them to the dense candidate store
// once the per-query counts are scanned.
__global__ __launch_bounds__(256) void k_qrec_scatter(const u64* __restrict__ p_qs, const u64* __restrict__ p_sd, const u64* __restrict__ p_ft,
                                                      const u32* __restrict__ rnk, u32 n, int bs, u32 qa, const u32* __restrict__ qoff,
                                                      u32* __restrict__ o_q, u32* __restrict__ o_subj, u64* __restrict__ o_sd, u64* __restrict__ o_ft) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u64 qs = p_qs[i];
    if (qs == UG_REC_NONE) return;
    const u32 q = (u32)(qs >> bs), s = qoff[q - qa] + rnk[i];
    o_q[s] = q, o_subj[s] = (u32)qs & ((1u << bs) - 1u), o_sd[s] = p_sd[i], o_ft[s] = p_ft[i];
}

#define QB_LB 11   // position bits of the two sort words (segments of up to 2048 records)
template <int P>
__device__ __forceinline__ void qb_bitonic(u64* s_key, u32 Pn, u32 lane) {   // ascending, Pn a power of two <= P, one wave
    for (u32 k = 2; k <= Pn; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 t = lane; t < (Pn >> 1); t += 64) {
                const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const u64 x = s_key[i], y = s_key[l];
                if ((x > y) == ((i & k) == 0)) s_key[i] = y, s_key[l] = x;
            }
            __syncthreads();
        }
}

// CAP: records of a query this instance holds; it serves the queries with lo_n < n <= CAP (the largest instance flags longer ones:
// the host then takes the sorting path for the pass)
template <int CAP>
__global__ __launch_bounds__(64) void k_q_best(const u32* __restrict__ qoff, u32 qa, const u32* __restrict__ o_subj, const u64* __restrict__ o_sd,
                                               const u64* __restrict__ o_ft, u32 seq_lo, int bsp, u32 lo_n, bool last, u32* __restrict__ t_rec,
                                               u32* __restrict__ perm, u32* __restrict__ qcnt, u32* __restrict__ fallback) {
    __shared__ u64 s_key[CAP], s_key2[CAP];
    const u32 qr = blockIdx.x, lane = threadIdx.x;
    const u32 a = qoff[qr], n = qoff[qr + 1] - a;
    if (n == 0 || n <= lo_n) return;
    if (n > (u32)CAP) {
        if (last && lane == 0) atomicOr(fallback, 1u);
        return;
    }
    const u64 pm = (1ull << bsp) - 1ull;
    auto ftword = [&](u64 ft) { return ((ft >> bsp) << 1) | ((ft & pm) == pm ? 1ull : 0ull); };
    auto store_rec = [&](u32 at, u32 subj, u32 score, int dist) {   // [global subject id, ungapped score, qi, qj]
        u32 qi, qj;
        if (dist > 0) qi = 0, qj = (u32)dist;
        else qi = (u32)(-dist), qj = 0;
        *reinterpret_cast<uint4*>(t_rec + 4 * (size_t)at) = make_uint4(subj + seq_lo, score, qi, qj);
    };
    if (n == 1) {
        if (lane == 0) {
            const u64 sd = o_sd[a];
            store_rec(a, o_subj[a], (u32)(sd >> 32), (int)(u32)sd);
            perm[a] = 0;
            qcnt[qa + qr] = 1;
        }
        return;
    }
    u32 P = 2;
    while (P < n) P <<= 1;
    for (u32 i = lane; i < P; i += 64) s_key[i] = i < n ? (((u64)o_subj[a + i] << QB_LB) | (u64)i) : ~0ull;
    __syncthreads();
    qb_bitonic<CAP>(s_key, P, lane);
    // runs of equal subjects -> candidates, numbered in subject order
    const unsigned long long lt = (1ull << lane) - 1ull;
    u32 nc = 0;
    for (u32 c0 = 0; c0 < n; c0 += 64) {
        const u32 i = c0 + lane;
        bool head = false;
        u32 subj = 0;
        if (i < n) {
            subj = (u32)(s_key[i] >> QB_LB);
            head = i == 0 || (u32)(s_key[i - 1] >> QB_LB) != subj;
        }
        const unsigned long long hm = __ballot(head);
        if (head) {
            const u32 ci = nc + (u32)__popcll(hm & lt);
            u64 minft = ~0ull, bft = ~0ull;
            u32 bscore = 0;
            int bdist = 0;
            for (u32 j = i; j < n && (u32)(s_key[j] >> QB_LB) == subj; ++j) {
                const u32 r = (u32)s_key[j] & ((1u << QB_LB) - 1u);
                const u64 sd = o_sd[a + r], ft = o_ft[a + r];
                const u32 sc = (u32)(sd >> 32);
                minft = ft < minft ? ft : minft;
                if (sc > bscore || (sc == bscore && ft < bft)) bscore = sc, bft = ft, bdist = (int)(u32)sd;
            }
            s_key2[ci] = (ftword(minft) << QB_LB) | (u64)ci;
            store_rec(a + ci, subj, bscore, bdist);
        }
        nc += (u32)__popcll(hm);
    }
    u32 P2 = 2;
    while (P2 < nc) P2 <<= 1;
    __syncthreads();
    for (u32 i = nc + lane; i < P2; i += 64) s_key2[i] = ~0ull;
    __syncthreads();
    if (nc > 1) qb_bitonic<CAP>(s_key2, P2, lane);
    for (u32 k = lane; k < nc; k += 64) perm[a + k] = (u32)s_key2[k] & ((1u << QB_LB) - 1u);
    if (lane == 0) qcnt[qa + qr] = nc;
}

// candidate k of query q (k < qcnt[q]) to the dense store: slot s = qoff[q - qa] + k of the scratch holds its rank-k candidate's
// position in perm
__global__ __launch_bounds__(256) void k_q_emit(const u32* __restrict__ qoff, u32 qa, u32 nqp, const u32* __restrict__ o_q, const u32* __restrict__ qcnt,
                                                const u32* __restrict__ coff, const u32* __restrict__ perm, const u32* __restrict__ t_rec,
                                                u32* __restrict__ out_q, u32* __restrict__ out_rec) {
    const u32 s = blockIdx.x * 256u + threadIdx.x;
    if (s >= qoff[nqp]) return;
    const u32 q = o_q[s], a = qoff[q - qa], k = s - a;
    if (k >= qcnt[q]) return;
    const u32 dst = coff[q - qa] + k;
    out_q[dst] = q;
    *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)dst) = *reinterpret_cast<const uint4*>(t_rec + 4 * (size_t)(a + perm[s]));
}

int q_best_max() { return 2048; }
void launch_qrec_scatter(const u64* p_qs, const u64* p_sd, const u64* p_ft, const u32* rnk, u32 n, int bs, u32 qa, const u32* qoff, u32* o_q, u32* o_subj,
                         u64* o_sd, u64* o_ft, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_qrec_scatter, dim3((n + 255) / 256), dim3(256), 0, st, p_qs, p_sd, p_ft, rnk, n, bs, qa, qoff, o_q, o_subj, o_sd, o_ft);
}
void launch_q_best(const u32* qoff, u32 qa, u32 nqp, const u32* o_subj, const u64* o_sd, const u64* o_ft, u32 seq_lo, int bsp, u32* t_rec, u32* perm,
                   u32* qcnt, u32* fallback, hipStream_t st) {
    if (!nqp) return;
    hipLaunchKernelGGL((k_q_best<128>), dim3(nqp), dim3(64), 0, st, qoff, qa, o_subj, o_sd, o_ft, seq_lo, bsp, 0u, false, t_rec, perm, qcnt, fallback);
    hipLaunchKernelGGL((k_q_best<512>), dim3(nqp), dim3(64), 0, st, qoff, qa, o_subj, o_sd, o_ft, seq_lo, bsp, 128u, false, t_rec, perm, qcnt, fallback);
    hipLaunchKernelGGL((k_q_best<2048>), dim3(nqp), dim3(64), 0, st, qoff, qa, o_subj, o_sd, o_ft, seq_lo, bsp, 512u, true, t_rec, perm, qcnt, fallback);
}
void launch_q_emit(const u32* qoff, u32 qa, u32 nqp, u32 nslots, const u32* o_q, const u32* qcnt, const u32* coff, const u32* perm, const u32* t_rec,
                   u32* out_q, u32* out_rec, hipStream_t st) {
    if (nslots) hipLaunchKernelGGL(k_q_emit, dim3((nslots + 255) / 256), dim3(256), 0, st, qoff, qa, nqp, o_q, qcnt, coff, perm, t_rec, out_q, out_rec);
}

int cand_order_lds_max() { return 2048; }
int cand_order_lds_key_bits() { return 64 - 11; }  // widest first-touch word the packed sort word holds

void launch_qseg(const u64* sorted_qs, u32 n, const u32* gidx, const u32* total, int bs, u32 q0, u32 q1, u32* seg, u32* maxseg, hipStream_t st) {
    hipLaunchKernelGGL(k_qseg, dim3((q1 - q0 + 1 + 255) / 256), dim3(256), 0, st, sorted_qs, n, gidx, total, bs, q0, q1, seg, maxseg);
}

void launch_cand_order_lds(const u64* c_ft, const u32* c_rec, const u32* seg, u32 q0, u32 q1, u32 maxseg, int bsp, u32* out_q, u32* out_rec, u32* qcnt,
                           hipStream_t st) {
    if (q1 <= q0) return;
    if (maxseg <= 256)
        hipLaunchKernelGGL((k_cand_order_lds<256, 11>), dim3(q1 - q0), dim3(64), 0, st, c_ft, c_rec, seg, q0, bsp, out_q, out_rec, qcnt);
    else
        hipLaunchKernelGGL((k_cand_order_lds<2048, 11>), dim3(q1 - q0), dim3(64), 0, st, c_ft, c_rec, seg, q0, bsp, out_q, out_rec, qcnt);
}

// ---- launch wrappers -------------------------------------------------------------------------------
void launch_group_list(const u32* flags, const u32* gidx, u32 H, u32* ghead, hipStream_t st) {
    if (!H) return;
    hipLaunchKernelGGL(k_group_list, dim3((H + 255) / 256), dim3(256), 0, st, flags, gidx, H, ghead);
}

// pass-list capacity per shard: every block can emit at most the groups whose heads it owns
u32 ungap_num_blocks(u32 H) { return (u32)(((u64)H + (u64)UW_RANGE * UW_WAVES - 1) / ((u64)UW_RANGE * UW_WAVES)); }
u32 ungap_shard_cap(u32 H) {
    const u32 nblk = ungap_num_blocks(H);
    return ((nblk + UG_SHARDS - 1) / UG_SHARDS) * (UW_RANGE * UW_WAVES);
}

void launch_ungap(const u64* keys, u32 H, const KeyLayout& kl, const KeyLayout& klr, const void* btab, bool gallop, bool ft_walk, const u8* q_scls, const u32* qoff,
                  const u8* r_scls, const u32* roff, const signed char* b62g, u32* shard_cnt, u32 shard_cap, u64* p_qs, u64* p_sd,
                  u64* p_ft, unsigned long long* group_count, hipStream_t st, const u32* words, const u32* bext, u32 nb, const BktLayout* L, bool skip_single,
                  unsigned long long* stat) {
    if (!H) return;
    const u32 wait_n = 20;   // waiting lanes that trigger the bookkeeping section (16 ... 32 measured alike)
    BktLayout L0 = BktLayout();
    L0.nqp = 1;
    // A wave walks its range's groups one after the other on each lane: with 4096 positions per wave a pass of a few million hits (the
    // long length classes of a mixed batch) fills a fraction of the 8192 wave slots and lasts as long as one wave does.  Passes below
    // half a fill get shorter ranges, down to 256 positions (heterogeneous 100 k set: 14.4 -> 13.0 ms of extension per step; config 3's
    // passes keep 4096 -- shorter ranges there only add table set-ups).
    u32 range = UW_RANGE;
    while (range > 256u && (u64)H / range < 4096ull) range >>= 1;
    const dim3 g((unsigned)(((u64)H + (u64)range * UW_WAVES - 1) / ((u64)range * UW_WAVES))), bl(64 * UW_WAVES);
#define UG_LAUNCH(K) hipLaunchKernelGGL(K, g, bl, 0, st, keys, words, bext, nb, words ? *L : L0, H, kl, klr.bs, klr.sh_subj, klr.sh_diag, (int)klr.diag_off, (const uint2*)btab, \
                                        (ft_walk ? 1 : 0) | (skip_single && words ? 2 : 0), wait_n, range, q_scls, qoff, r_scls, roff, b62g, shard_cnt, shard_cap, p_qs, p_sd, p_ft, group_count, stat)
#define UG_PICK(C, BA, GA, W, CT) UG_LAUNCH((k_ungap<C, BA, GA, W, CT>))
#define UG_BY_FLAGS(W, CT)                   \
    do {                                     \
        if (gallop) {                        \
            if (btab) UG_PICK(3, true, true, W, CT);   \
            else UG_PICK(3, false, true, W, CT);       \
        } else {                             \
            if (btab) UG_PICK(3, true, false, W, CT);  \
            else UG_PICK(3, false, false, W, CT);      \
        }                                    \
    } while (0)
    if (stat) {   // counting instances (tests, bench's per-kernel work rates)
        if (words) UG_BY_FLAGS(true, true);
        else UG_BY_FLAGS(false, true);
        return;
    }
    if (words) {   // the buckets' 32-bit words (k_bkt_group), bucketed passes
        UG_BY_FLAGS(true, false);
        return;
    }
    UG_BY_FLAGS(false, false);   // (three chunk steps per loop iteration: instances with one and two measured slower, rounds 1-2)
#undef UG_BY_FLAGS
#undef UG_PICK
#undef UG_LAUNCH
}

void launch_shard_scan(const u32* shard_cnt, u32* shard_off, hipStream_t st) {
    hipLaunchKernelGGL(k_shard_scan, dim3(1), dim3(64), 0, st, shard_cnt, shard_off);
}

void launch_compact_shards(const u32* shard_cnt, u32* shard_off, u32 shard_cap, const u64* a0, const u64* a1, const u64* a2, u64* b0,
                           u64* b1, u64* b2, hipStream_t st) {
    hipLaunchKernelGGL(k_compact_shards, dim3(64, UG_SHARDS), dim3(256), 0, st, shard_cnt, shard_off, shard_cap, a0, a1, a2, b0, b1, b2);
}

void launch_seg_flags(const u64* sorted_qs, u32 n, u32* flags, u32* zero_word, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_seg_flags, dim3((n + 255) / 256), dim3(256), 0, st, sorted_qs, n, flags, zero_word);
}

void launch_best(const u64* sorted_qs, const u32* idx, const u32* shead, u32 nseg, u32 n, const u64* p_sd, const u64* p_ft,
                 u32 seq_lo, int bs, u64* c_ft, u32* c_q, u32* c_rec, hipStream_t st) {
    if (!nseg) return;
    hipLaunchKernelGGL(k_best, dim3((nseg + 255) / 256), dim3(256), 0, st, sorted_qs, idx, shead, nseg, n, p_sd, p_ft, seq_lo, bs, c_ft,
                       c_q, c_rec);
}

void launch_gather_u32_as_u64(const u32* src, const u32* idx, u32 n, u64* dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather_u32_as_u64, dim3((n + 255) / 256), dim3(256), 0, st, src, idx, n, dst);
}

void launch_combine_q_ft(const u32* c_q, const u64* c_ft, u32 n, int ftbits, int bsp, u64* dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_combine_q_ft, dim3((n + 255) / 256), dim3(256), 0, st, c_q, c_ft, n, ftbits, bsp, dst);
}

void launch_iota(u32* p, u32 n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
}

void launch_emit_cands(const u32* order, u32 n, const u64* sorted_key, int qshift, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                       u32* seg_first, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_emit_cands, dim3((n + 255) / 256), dim3(256), 0, st, order, n, sorted_key, qshift, c_rec, out_q, out_rec, seg_first);
    hipLaunchKernelGGL(k_seg_counts, dim3((n + 255) / 256), dim3(256), 0, st, out_q, n, seg_first, qcnt);
}
