// k_ungapq.hip -- the singleton groups of a SPARSE seed pass, a wave per QUERY (fsearch.py:2681-2713: the `hits` dict of one query against
// one chunk, Fasta.ungap with one seed).
//
// A pass of the long seeds (BASELINE config 3: seed 11111011111, a few hundred index entries per query and chunk) is too thin for the
// bucketed binning -- a (query, band range) bucket would hold nine hits -- so it went  k_lookup (8-byte key per hit) -> segmented library
// radix sort -> k_ungap  (0.47 + 0.94 + 2.6 ms per config-3 chunk).  About half of such a pass's hits are random matches alone on their
// (subject, diagonal): they need no order at all, only the knowledge that they ARE alone.  This kernel finds that out without a sort:
//
//   * a wave takes a query (work counter), puts its score classes into an LDS slot (k_ungap1's, sentinel-bounded) and walks the query's
//     seeds TWICE, 64 seeds at a time, every lane the entries of its own seed: pass 1 marks  hash(G)  of every hit's banded diagonal id
//     G = addend + qpos  in a table of 4096 two-bit counters (seen once / seen again), pass 2 regenerates the hits (the entries are in
//     L1 / L2 by then) and looks them up: "seen once" is PROOF that no other hit of this query shares the diagonal;
//   * such a hit goes straight into k_ungap1's ring and is extended by the next free lane with k_ungap1's step (both passes at once in the
//     two 16-bit halves, 16 residues per step, lane-private score table) -- no key is written, sorted or read for it;
//   * every other hit (the diagonals with two and more hits, plus the few singletons that share a counter with another diagonal) is written
//     as the 8-byte key k_lookup would have written, at its own place of the key array, which the host has filled with all-ones (= dropped)
//     before: the sorted path then sorts and walks only what is left.
// The pass records have k_ungap1's form (head key in the record layout), into the same list as k_ungap's.
#include "ungap1.h"

#define UQ_WAVES 16
#define UQ_RING 128
#define UQ_TBITS 12                       // 4096 two-bit counters = 1 KB per wave
#define UQ_QCAP 512                       // longest query the slot holds
#define UQ_QSLOT (UQ_QCAP + 2 * U1_QPAD)
// per wave: query slot, counters, ring (subject byte, diagonal id, query position)
#define UQ_TILE 256                       // hit ordinals per tile of the owner marks (mark, slot base, query position: 8 bytes per ordinal)
#define UQ_WAVE_BYTES (UQ_QSLOT + (1 << UQ_TBITS) / 4 + UQ_RING * 4 + UQ_RING * 4 + UQ_RING * 2 + UQ_TILE * 8)

__device__ __forceinline__ u32 bk_scan_max16(u32 x) {  // inclusive max-scan over the wave (DPP)
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}

__device__ __forceinline__ u32 uq_slot(u32 g) { return (g * 2654435761u) >> (32 - UQ_TBITS); }

// first compacted seed of every pass query: smallest k with cs_hoff[k] >= qseg[q] (queries without hits: their successor's)
__global__ __launch_bounds__(256) void k_uq_first(const u32* __restrict__ qseg, u32 nqp, const u32* __restrict__ cs_hoff, u32 K, u32* __restrict__ qk) {
    const u32 q = blockIdx.x * 256u + threadIdx.x;
    if (q > nqp) return;
    const u32 x = qseg[q];
    u32 lo = 0, hi = K;   // first k in [0, K] with cs_hoff[k] >= x
    while (lo < hi) {
        const u32 m = (lo + hi) >> 1;
        if (cs_hoff[m] < x) lo = m + 1;
        else hi = m;
    }
    qk[q] = lo;
}

template <bool BANDS, bool COUNT>
__global__ __launch_bounds__(64 * UQ_WAVES, 1) __attribute__((amdgpu_num_sgpr(96))) void k_ungapq(
    const u32* __restrict__ qseg /*first hit ordinal of every pass query, + end*/, u32 nqp, u32 qa, const u32* __restrict__ qk, const u32* __restrict__ cs_hoff,
    const u32* __restrict__ cs_base, const u64* __restrict__ cs_kbase, const u32* __restrict__ dk32, int bp, int bd, int sh_q, int sh_qpos, int sh_diag, int diag_off, int rbs,
    int rsh_subj, int rsh_diag, int rdoff, const uint2* __restrict__ btab, u32 wait_n, const u8* __restrict__ q_scls, const u32* __restrict__ qoff,
    const u8* __restrict__ r_ug, const u32* __restrict__ roff, const signed char* __restrict__ b62g, u32* __restrict__ work_ctr, u32* __restrict__ shard_cnt,
    u64* __restrict__ p_qs, u64* __restrict__ p_sd, u64* __restrict__ p_ft, unsigned long long* __restrict__ group_count, u64* __restrict__ keys,
    u64* __restrict__ keys_sorted, u32* __restrict__ seg_end, unsigned long long* __restrict__ stat /*COUNT: [0] += b62 lookups, [1] += groups*/) {
    constexpr int TSH = 3;
    constexpr u32 TBYTES = (u32)U1_ROWS << (8 + TSH);
    __shared__ __align__(16) unsigned char uq_smem[TBYTES + UQ_WAVES * UQ_WAVE_BYTES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (u32 i = threadIdx.x; i < U1_ROWS * 32u * 32u; i += 64 * UQ_WAVES) {   // (u1_fill_table strides by U1_WAVES threads)
        const u32 q = i >> 10, s = (i >> 5) & 31u, l = i & 31u;
        const int v = (q < SCLS_N && s < SCLS_N) ? (int)b62g[q * SCLS_N + s] : -100;
        *reinterpret_cast<short*>(uq_smem + ((q << 11) | (s << 6) | (l << 1))) = (short)v;
    }
    __syncthreads();
    unsigned char* wbase = uq_smem + TBYTES + (u32)w * UQ_WAVE_BYTES;
    u8* qslot = wbase;
    u32* tab = reinterpret_cast<u32*>(wbase + UQ_QSLOT);
    u32* ring_s = tab + (1 << UQ_TBITS) / 16;
    u32* ring_g = ring_s + UQ_RING;
    u16* ring_q = reinterpret_cast<u16*>(ring_g + UQ_RING);
    u32* mbase = reinterpret_cast<u32*>(ring_q + UQ_RING);   // (16-byte aligned: the slot, counters and rings before it are multiples of 16 bytes)
    u16* marks = reinterpret_cast<u16*>(mbase + UQ_TILE);
    u16* mqpos = marks + UQ_TILE;
    static_assert(UQ_TILE == 256, "tile_marks clears the marks with two words per lane");
    static_assert(U1_PCAP * 8 <= 7 * 64 && UQ_WAVES <= U1_ROWS, "pass buffer in a table row's tail");
    u64* s_pass = reinterpret_cast<u64*>(uq_smem + ((u32)w << 11) + 25 * 64);   // (diagonal id | score << 32), query position in s_passq
    u16* s_passq = reinterpret_cast<u16*>(uq_smem + ((u32)w << 11) + 25 * 64 + U1_PCAP * 8);   // 64 more bytes of the same row tail
    static_assert(U1_PCAP * 8 + U1_PCAP * 2 <= 7 * 64, "row tail");
    const unsigned long long lt = (1ull << lane) - 1ull;
    const u32 lanebase = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)uq_smem + ((u32)(lane & 31) << 1);
    const u32 pmask = (1u << bp) - 1u;
    const u32 dmask = (1u << bd) - 1u;
    u32 npb = 0, ngroups = 0;
    u32 ch_pos = 0, ch_end = 0;   // the wave's reserved piece of the pass list
    U1Track tr_unused;
    U1Count ct;
    ct.n = 0;
    ct.start();

    for (;;) {
        u32 qrel = 0;
        if (lane == 0) qrel = atomicAdd(work_ctr, 1u);
        qrel = (u32)__builtin_amdgcn_readfirstlane((int)qrel);
        if (qrel >= nqp) break;
        const u32 h0 = (u32)__builtin_amdgcn_readfirstlane((int)qseg[qrel]), h1 = (u32)__builtin_amdgcn_readfirstlane((int)qseg[qrel + 1]);
        if (h0 == h1) {
            if (lane == 0) seg_end[qrel] = h0;
            continue;
        }
        const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)qk[qrel]), k1 = (u32)__builtin_amdgcn_readfirstlane((int)qk[qrel + 1]);
        const u32 gq = qa + qrel;
        const u32 qb0 = (u32)__builtin_amdgcn_readfirstlane((int)qoff[gq]);
        const int ql = (int)((u32)__builtin_amdgcn_readfirstlane((int)qoff[gq + 1]) - qb0);
        // ---- the query's classes into the slot: [pad][classes, position 0 = sentinel][pad]; the counters cleared ----
        for (int i = lane * 16; i < UQ_QSLOT; i += 64 * 16) {
            const int p0 = i - U1_QPAD;
            uint4 v = make_uint4(0x18181818u, 0x18181818u, 0x18181818u, 0x18181818u);
            if (p0 >= 0 && p0 < ql) {
                v = u1_load16(q_scls + qb0 + (u32)p0);
                u32* d = reinterpret_cast<u32*>(&v);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int nv = min(max(ql - p0 - 4 * k, 0), 4);
                    const u32 m = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
                    d[k] = (d[k] & m) | (0x18181818u & ~m);
                }
                if (p0 == 0) d[0] = (d[0] & 0xFFFFFF00u) | 0x18u;
            }
            *reinterpret_cast<uint4*>(qslot + i) = v;
        }
        reinterpret_cast<uint4*>(tab)[lane] = make_uint4(0, 0, 0, 0);   // 256 words
        u1_wave_sync();
        const int qr_max = U1_QPAD + ql;
        const u64 kq = (u64)gq << sh_q;   // the query's share of a sort key (one alphabet x one pattern: the as / tag fields are 0)

        // The query's hits are its seeds' index slots one after the other: hit ordinal o (0 <= o < n) belongs to the seed that started at or before
        // it.  Both walks take the ordinals 64 at a time, a lane each -- every lane busy, a seed's entries read as one run -- in TILES of UQ_TILE
        // ordinals: the seeds that start inside the tile leave their number at their first ordinal in an LDS array, and a prefix maximum over
        // it (carried from step to step, and from tile to tile) is the owner of every ordinal (k_lookup's scheme).  Walking seed by seed, a
        // lane per seed, had the wave spend eight turns on a round whose seeds hold 1.5 hits on average.
        const u32 n = h1 - h0;
        u32 t_next = k0;          // first seed not yet marked
        u32 c_base = 0, c_qpos = 0;   // the seed that owns the ordinals in front of the tile's first mark: its slot base and query position
        auto tile_marks = [&](u32 tb) {   // seeds starting in [tb, tb + UQ_TILE): position + 1, slot base and query position at their first ordinal
            reinterpret_cast<u32*>(marks)[lane] = 0, reinterpret_cast<u32*>(marks)[64 + lane] = 0;   // UQ_TILE u16 marks
            u1_wave_sync();
            for (;;) {   // (wave-uniform trip count)
                const u32 k = t_next + (u32)lane;
                u32 o = 0xFFFFFFFFu;
                if (k < k1) o = cs_hoff[k] - h0 - tb;
                const bool in = o < (u32)UQ_TILE;
                if (in) {
                    marks[o] = (u16)(o + 1u);
                    mbase[o] = cs_base[k] + h0;   // (u32 wrap-around intended: cs_base = first slot - first ordinal of the pass)
                    mqpos[o] = (u16)((u32)(cs_kbase[k] >> sh_qpos) & pmask);
                }
                const u32 cnt = (u32)__popcll(__ballot(in));
                t_next += cnt;
                if (cnt < 64u) break;
            }
            u1_wave_sync();
        };
        // one step: the ordinals tb + 64 st + lane -> (valid, addend, query position, ordinal); ONE trip to memory per step (the index entry)
        auto gen_step = [&](u32 tb, u32 st, u32& c, u32& qpos, u32& o) -> bool {
            o = tb + 64u * st + (u32)lane;
            const u32 p1 = bk_scan_max16((u32)marks[64u * st + (u32)lane]);   // first ordinal (+ 1) of the owner inside this step's part of the tile, 0: before it
            u32 base = c_base;
            qpos = c_qpos;
            if (p1) base = mbase[p1 - 1u], qpos = (u32)mqpos[p1 - 1u];
            c_base = (u32)__builtin_amdgcn_readlane((int)base, 63), c_qpos = (u32)__builtin_amdgcn_readlane((int)qpos, 63);
            c = 0xFFFFFFFFu;
            if (o >= n) return false;
            c = dk32[base + o];
            return (int)c >= 0;
        };
        // ---- pass 1: the two-bit counters ----
        for (u32 tb = 0; tb < n; tb += UQ_TILE) {
            tile_marks(tb);
            const u32 nst = (min(n - tb, (u32)UQ_TILE) + 63u) / 64u;
            for (u32 st = 0; st < nst; ++st) {
                u32 c, qpos, o;
                if (gen_step(tb, st, c, qpos, o)) {
                    const u32 t = uq_slot(c + qpos), sh = (t & 15u) * 2u;
                    const u32 old = atomicOr(&tab[t >> 4], 1u << sh);
                    if ((old >> sh) & 1u) atomicOr(&tab[t >> 4], 2u << sh);
                }
            }
        }
        u1_wave_sync();

        // ---- pass 2: hits alone on their diagonal -> ring -> extension; the others -> their 8-byte keys ----
        auto flush = [&]() {   // the buffered singletons (all of this query) -> pass records, one per lane
            const u32 at = u1_reserve(npb, lane, ch_pos, ch_end, &shard_cnt[0], U1_CHUNK);
            if ((u32)lane < npb) {
                const u64 e = s_pass[lane];
                const u32 G = (u32)e, qp = (u32)s_passq[lane];
                u32 gsubj;
                int dlt;
                if (BANDS) {
                    const uint2 be = btab[G >> bd];
                    gsubj = be.x;
                    dlt = (int)(be.y - G);
                } else {
                    gsubj = G >> bd;
                    dlt = diag_off - (int)(G & dmask);
                }
                p_qs[at] = ((u64)gq << rbs) | gsubj;
                p_sd[at] = (e & 0xFFFFFFFF00000000ull) | (u64)(u32)dlt;
                p_ft[at] = ((u64)gsubj << rsh_subj) | ((u64)(u32)(rdoff - dlt) << rsh_diag) | ((u64)qp << sh_qpos);
            }
            npb = 0;
        };
        t_next = k0, c_base = 0, c_qpos = 0;
        u32 g_tb = 0, g_st = 0, g_nst = (min(n, (u32)UQ_TILE) + 63u) / 64u;
        tile_marks(0);
        bool more = true;   // the generator has hits left
        u32 ns = 0;         // hits left to the sorted path so far (they sit at the front of the query's part of the key array)
        u32 rfront = 0, rback = 0;
        // lane state
        bool working = false;
        int qR = 0, qL = 0;
        i64 sR = 0, sL = 0;
        u32 hG = 0, hq = 0;
        pk16 S = {0, 0}, M = {0, 0};
        for (;;) {
            const unsigned long long idleb = __ballot(!working);
            if (idleb && ((u32)__popcll(idleb) >= wait_n || idleb == ~0ull)) {
                // ---- refill the ring: 64 hit ordinals per turn ----
                while (rback - rfront <= UQ_RING - 64u && more) {
                    u32 c, qpos, o;
                    const bool valid = gen_step(g_tb, g_st, c, qpos, o);
                    bool sing = false;
                    u32 G = 0;
                    if (valid) {
                        G = c + qpos;
                        const u32 t = uq_slot(G);
                        sing = ((tab[t >> 4] >> ((t & 15u) * 2u)) & 3u) == 1u;
                    }
                    const unsigned long long sb_ = __ballot(sing), kb_ = __ballot(valid && !sing);
                    if (valid && !sing) keys[h0 + ns + (u32)__popcll(kb_ & lt)] = kq + ((u64)qpos << sh_diag) + ((u64)qpos << sh_qpos) + ((u64)c << sh_diag);   // k_lookup's key
                    ns += (u32)__popcll(kb_);
                    if (sing) {
                        u32 gsubj;
                        int dlt;
                        if (BANDS) {
                            const uint2 be = btab[G >> bd];
                            gsubj = be.x;
                            dlt = (int)(be.y - G);   // sst - qpos
                        } else {
                            gsubj = G >> bd;
                            dlt = diag_off - (int)(G & dmask);
                        }
                        const u32 slot = (rback + (u32)__popcll(sb_ & lt)) & (UQ_RING - 1);
                        ring_s[slot] = roff[gsubj] + (u32)((int)qpos + dlt);   // byte of (subject, sst) in r_ug
                        ring_g[slot] = G;
                        ring_q[slot] = (u16)qpos;
                    }
                    rback += (u32)__popcll(sb_);
                    if (++g_st >= g_nst) {
                        g_st = 0, g_tb += UQ_TILE;
                        if (g_tb < n) {
                            g_nst = (min(n - g_tb, (u32)UQ_TILE) + 63u) / 64u;
                            tile_marks(g_tb);
                        } else {
                            more = false;
                        }
                    }
                }
                u1_wave_sync();
                // ---- hand out ----
                const u32 avail = rback - rfront;
                if (!working) {
                    const u32 rk = (u32)__popcll(idleb & lt);
                    if (rk < avail) {
                        const u32 slot = (rfront + rk) & (UQ_RING - 1);
                        const u32 sa = ring_s[slot];
                        hG = ring_g[slot], hq = (u32)ring_q[slot];
                        qR = U1_QPAD + (int)hq, qL = qR - 16;
                        sR = (i64)sa, sL = (i64)sa - 16;
                        S = pk16{0, 0}, M = pk16{0, 0};
                        if (COUNT) ct.start();
                        working = true;
                    }
                }
                const u32 taken = min((u32)__popcll(idleb), avail);
                rfront += taken;
                ngroups += taken;
                u1_wave_sync();   // ring slots may be overwritten by the next refill only after these reads
            }
            if (!__ballot(working)) {
                if (!more && rback == rfront) break;   // query done
                continue;
            }
            bool fin = false;
            if (working) {
                const uint4 qr4 = u1_lds16(qslot, qR), ql4 = u1_lds16(qslot, qL);
                const uint4 sr4 = u1_load16(r_ug + sR), sl4 = u1_load16(r_ug + sL);
                const u32 msk = u1_step<TSH, false, COUNT>(qr4, ql4, sr4, sl4, lanebase, S, M, tr_unused, 0, ct);
                qR = min(qR + 16, qr_max), qL = max(qL - 16, 0);
                sR += 16, sL -= 16;
                fin = msk == 0xFFFFFFFFu;   // both passes have ended
            }
            // ---- finished singletons: buffer the ones that reach MIN_UNGAP ----
            {
                const int score = (int)M.x + (int)M.y;
                bool todo = fin && score >= MIN_UNGAP;
                for (;;) {   // (one round unless more lanes pass in a step than the buffer has room for)
                    const unsigned long long pb = __ballot(todo);
                    if (!pb) break;
                    const u32 room = U1_PCAP - npb, rk = (u32)__popcll(pb & lt);
                    if (todo && rk < room) {
                        s_pass[npb + rk] = ((u64)(u32)score << 32) | hG;
                        s_passq[npb + rk] = (u16)hq;
                        todo = false;
                    }
                    npb += min((u32)__popcll(pb), room);
                    u1_wave_sync();
                    if (npb == U1_PCAP) {
                        flush();
                        u1_wave_sync();
                    }
                }
                if (fin) working = false;
            }
        }
        if (npb) {   // the buffered records belong to this query
            flush();
            u1_wave_sync();
        }
        // the sorted path's view of this query: keys [h0, h0 + ns) to sort, dropped keys behind them in the sort's OUTPUT array
        if (lane == 0) seg_end[qrel] = h0 + ns;
        for (u32 i = h0 + ns + (u32)lane; i < h1; i += 64) keys_sorted[i] = ~0ull;
    }
    for (u32 i = ch_pos + (u32)lane; i < ch_end; i += 64) p_qs[i] = UG_REC_NONE, p_ft[i] = 0;   // unused slots of the wave's last piece: skipped downstream
    if (lane == 0 && ngroups) atomicAdd(&group_count[0], (unsigned long long)ngroups);
    if (COUNT) {
        unsigned long long nst = ct.n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nst += __shfl_xor(nst, o);
        if (lane == 0) atomicAdd(&stat[0], nst), atomicAdd(&stat[1], (unsigned long long)ngroups);
    }
}

u32 ungapq_qcap() { return UQ_QCAP; }

// qk: nqp + 1 words of scratch.  Afterwards a query's keys to sort are keys[qseg[q] .. seg_end[q]); keys_sorted (the sort's output array) holds
// dropped keys (all-ones) behind them up to qseg[q + 1].
void launch_ungapq(u32 ncu, const u32* qseg, u32 nqp, u32 qa, u32* qk, const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase, u32 K, const u32* dk32,
                   const KeyLayout& kl, const KeyLayout& klr, const void* btab, const u8* q_scls, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g,
                   u32* work_ctr, u32* shard_cnt, u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, u64* keys, u64* keys_sorted, u32* seg_end,
                   unsigned long long* stat, hipStream_t st) {
    if (!nqp) return;
    hipLaunchKernelGGL(k_uq_first, dim3((nqp + 1 + 255) / 256), dim3(256), 0, st, qseg, nqp, cs_hoff, K, qk);
    static_assert(((size_t)U1_ROWS << 11) + (size_t)UQ_WAVES * UQ_WAVE_BYTES <= 160 * 1024, "LDS of a CU");
    const dim3 g(std::min<u32>(ncu, (nqp + UQ_WAVES - 1) / UQ_WAVES)), bl(64 * UQ_WAVES);
#define UQ_GO(B, CT)                                                                                                                                              \
    hipLaunchKernelGGL((k_ungapq<B, CT>), g, bl, 0, st, qseg, nqp, qa, qk, cs_hoff, cs_base, cs_kbase, dk32, kl.bp, kl.bd, kl.sh_q, kl.sh_qpos, kl.sh_diag, (int)kl.diag_off, klr.bs,  \
                       klr.sh_subj, klr.sh_diag, (int)klr.diag_off, (const uint2*)btab, U1_WAIT, q_scls, qoff, r_ug, roff, b62g, work_ctr, shard_cnt, p_qs, p_sd, p_ft,      \
                       group_count, keys, keys_sorted, seg_end, stat)
    if (stat) {
        if (btab) UQ_GO(true, true);
        else UQ_GO(false, true);
    } else if (btab) UQ_GO(true, false);
    else UQ_GO(false, false);
#undef UQ_GO
}
