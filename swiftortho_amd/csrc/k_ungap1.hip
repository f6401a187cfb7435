// k_ungap1.hip -- ungapped X-drop extension of the SINGLETON groups of a bucketed seed pass
// (fsearch.py:2454-2494 Fasta.ungap, 2497-2509 get_ungap_scores with ONE seed, 2696-2713 the group loop).
//
// 84 % of the (query, subject, diagonal) groups of a dense pass hold one seed hit (1.73e12 groups for 2.06e12 hits on the 1 M-protein
// run).  For such a group the reference's chain collapses: no lower bounds from a previous segment, no hit walk, and the left pass --
// which starts at the right pass's maximum WITH that score (2479-2492) -- adds what it would add to any starting score, so
//     group score = (maximum of the right pass from 0) + (maximum of the left pass from 0)
// and the two passes are independent.  k_ungap (k_group.hip) spends 40 % of its instructions on the state machine that chains seeds and
// runs its chunk steps with 36 of 64 lanes; this kernel does nothing but X-drop steps:
//
//   * one WAVE per bucket (= one query x 2^wb diagonal bands, 1-3 k hits, words sorted by (band, diagonal, qpos) by k_bkt_group): the
//     query's score classes sit in an LDS slot, the wave scans the bucket's words 64 at a time, keeps the singletons (a head whose
//     successor starts another group) as 8-byte entries (subject byte offset, hit word) in an LDS ring, and a lane that finishes takes
//     the next entry.  Groups of two and more hits are left to k_ungap, which skips the singletons (`skip_single`);
//   * BOTH passes at once in the two 16-bit halves of a register (low: right, high: left), 16 residues per step and direction: four
//     16-byte loads per step (query from LDS, subject from global memory) instead of one 8-byte pair per 8 residues;
//   * NO limits: position 0 of a sequence is never scored by the reference (`qlo < qst`, strict, with qlo = 0: a right pass that would
//     start there does not run, a left pass stops in front of it) and neither is the position behind its last residue, so both carry a
//     SENTINEL class (score -100 against everything: an X-drop at once) in the arrays this kernel reads -- the subject side in a
//     copy of the reference's class array made for it (`r_ug`: class * 8, position 0 of every sequence and the pads = sentinel; the
//     position behind a sequence is the next one's position 0), the query side in the LDS slot.  A pass ends where the reference's
//     loop condition or its X-drop ends it, and nothing has to be counted or masked;
//   * a LANE-PRIVATE score table (one 32-bit entry per (query class, subject class, lane mod 32): no LDS bank conflicts; address = one
//     v_perm + one v_lshl_or);
//   * the X-drop is tested and made sticky once per group of three elements, on the group's lowest running score: a score that fell 31
//     below the maximum cannot climb back to it within two further elements (2 * 11 < 31), so neither a late pin (running score :=
//     -8192) nor a maximum that rose earlier in the same group changes the outcome.
// Per element pair: 4 address + 1 combine + 3 packed ALU instructions + 5 per group of three = 9.7, against 12.5 per ELEMENT in k_ungap.
#include "ungap1.h"

// class * mul with sentinels: mul = 8: the subject side (r_ug), mul = 1: the query side of the chain kernel (q_ug); see the head of
// the file.  `out` points U1_UG_PAD bytes into its allocation.
__global__ __launch_bounds__(256) void k_make_ug(const u8* __restrict__ scls, size_t n, u32 mul, u8* __restrict__ out) {
    const i64 j = (i64)((size_t)blockIdx.x * 256u + threadIdx.x) - U1_UG_PAD;
    if (j < (i64)n + U1_UG_PAD) out[j] = (j >= 0 && j < (i64)n) ? (u8)(scls[j] * mul) : (u8)(U1_SENT * mul);
}
__global__ __launch_bounds__(256) void k_ug_starts(const u32* __restrict__ off, u32 nseq, u32 mul, u8* __restrict__ out) {
    const u32 s = blockIdx.x * 256u + threadIdx.x;
    if (s < nseq && off[s + 1] > off[s]) out[off[s]] = (u8)(U1_SENT * mul);
}

void launch_make_ug(const u8* scls, const u32* off, u32 nseq, size_t nres, u32 mul, u8* out, hipStream_t st) {
    const size_t tot = nres + 2 * (size_t)U1_UG_PAD;
    hipLaunchKernelGGL(k_make_ug, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, scls, nres, mul, out);
    if (nseq) hipLaunchKernelGGL(k_ug_starts, dim3((nseq + 255) / 256), dim3(256), 0, st, off, nseq, mul, out);
}

// ================================================================================================================
// singleton groups
// ================================================================================================================
// QCAP: longest query the slot holds; WGS: workgroups per CU the LDS budget is cut for (512-residue slots and 16-bit table entries:
// two workgroups = 8 waves per SIMD; 1024-residue slots or 32-bit entries: one).
// mlist (nullable): the heads of the groups of two and more hits are appended to it as position | bucket << 32, for k_ungap2 (pieces
// of U1_LCHUNK entries per wave, unused entries = UG_REC_NONE); without it those groups are k_ungap's (skip_single).
template <bool BANDS, int TSH, int QCAP, int WGS, bool COUNT>
__global__ __launch_bounds__(64 * U1_WAVES, WGS) __attribute__((amdgpu_num_sgpr(80))) void k_ungap1(   // (8 waves per SIMD need <= 96 SGPRs with VCC and the rest)
    const u32* __restrict__ words, const u32* __restrict__ bext, u32 nb, BktLayout L, int sh_qpos, int diag_off, int rbs, int rsh_subj, int rsh_diag, int rdoff,
    const uint2* __restrict__ btab, u32 wait_n, const u8* __restrict__ q_scls, const u32* __restrict__ qoff, const u8* __restrict__ r_ug, const u32* __restrict__ roff,
    const signed char* __restrict__ b62g, u32* __restrict__ work_ctr, u32* __restrict__ shard_cnt, u64* __restrict__ p_qs, u64* __restrict__ p_sd,
    u64* __restrict__ p_ft, unsigned long long* __restrict__ group_count, u64* __restrict__ mlist, u32* __restrict__ mlist_cnt,
    unsigned long long* __restrict__ stat /*COUNT: [0] += b62 lookups, [1] += groups*/) {
    constexpr u32 TBYTES = (u32)U1_ROWS << (8 + TSH);
    constexpr int QSLOT = QCAP + 2 * U1_QPAD;
    __shared__ __align__(16) unsigned char u1_smem[TBYTES + U1_WAVES * U1_WAVE_BYTES(QCAP, TSH)];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u1_fill_table<TSH>(u1_smem, b62g);
    __syncthreads();
    unsigned char* wbase = u1_smem + TBYTES + (u32)w * U1_WAVE_BYTES(QCAP, TSH);
    u8* qslot = wbase;
    u32* ring_s = reinterpret_cast<u32*>(wbase + QSLOT);
    u32* ring_w = ring_s + U1_RING;
    static_assert(U1_PCAP * 8 <= 7 * 64 && U1_WAVES <= U1_ROWS, "pass buffer in a table row's tail");
    u64* s_pass = reinterpret_cast<u64*>(TSH == 3 ? u1_smem + ((u32)w << 11) + 25 * 64 : wbase + QSLOT + U1_RING * 8);
    const unsigned long long lt = (1ull << lane) - 1ull;
    // LDS byte address of the lane's copy of table entry (0, 0): a lookup's address is ONE v_lshl_add on top of the v_perm
    const u32 lanebase = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)u1_smem + ((u32)(lane & 31) << (TSH == 4 ? 2 : 1));
    const u32 WM = 0x7FFFFFFFu;
    const u32 pmask = (1u << L.bp) - 1u;
    const int gb = L.wb + L.bd;
    u32 npb = 0, ngroups = 0;
    u32 ch_pos = 0, ch_end = 0;   // the wave's reserved piece of the pass list: next free slot, end
    u32 lc_pos = 0, lc_end = 0;   // ... and of the chain list
    U1Track tr_unused;
    U1Count ct;
    ct.n = 0;
    ct.start();

    for (;;) {
        // ---- next bucket (range-major: the chip works on one subject range at a time) ----
        u32 b = 0;
        if (lane == 0) b = atomicAdd(work_ctr, 1u);
        b = (u32)__builtin_amdgcn_readfirstlane((int)b);
        if (b >= nb) break;
        const u32 e0 = (u32)__builtin_amdgcn_readfirstlane((int)bext[b]), e1 = (u32)__builtin_amdgcn_readfirstlane((int)bext[b + 1]);
        if (e0 == e1) continue;
        const u32 r = b / L.nqp, qrel = b - r * L.nqp, gq = L.qa + qrel;
        const u32 qb0 = (u32)__builtin_amdgcn_readfirstlane((int)qoff[gq]);
        const int ql = (int)((u32)__builtin_amdgcn_readfirstlane((int)qoff[gq + 1]) - qb0);
        // ---- the query's classes into the slot: [pad][classes, position 0 = sentinel][pad] ----
        for (int i = lane * 16; i < QSLOT; i += 64 * 16) {
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
        u1_wave_sync();
        const int qr_max = U1_QPAD + ql;
        auto flush = [&]() {   // the buffered singletons (all of this bucket) -> pass records, one per lane
            const u32 at = u1_reserve(npb, lane, ch_pos, ch_end, &shard_cnt[0], U1_CHUNK);
            if ((u32)lane < npb) {
                const u64 e = s_pass[lane];
                u1_record<BANDS>((u32)e, e & 0xFFFFFFFF00000000ull, r, gq, L, diag_off, rbs, rsh_subj, rsh_diag, rdoff, sh_qpos, btab, p_qs, p_sd, p_ft, at);
            }
            npb = 0;
        };
        u32 cur = e0, rfront = 0, rback = 0;
        // lane state
        bool working = false;
        int qR = 0, qL = 0;
        i64 sR = 0, sL = 0;
        u32 hw = 0;
        pk16 S = {0, 0}, M = {0, 0};
        for (;;) {
            const unsigned long long idleb = __ballot(!working);
            if (idleb && ((u32)__popcll(idleb) >= wait_n || idleb == ~0ull)) {
                // ---- refill the ring: scan 64 words, keep the singletons ----
                while (rback - rfront <= U1_RING - 64u && cur < e1) {
                    const u32 pos = cur + (u32)lane;
                    const bool valid = pos < e1;
                    const u32 wv = valid ? (words[pos] & WM) : 0u;
                    u32 wp = (u32)__shfl_up((int)wv, 1);
                    if (lane == 0) wp = cur == e0 ? ~wv : (words[cur - 1] & WM);
                    u32 wn = (u32)__shfl_down((int)wv, 1);
                    if (lane == 63) wn = pos + 1u < e1 ? (words[pos + 1u] & WM) : ~wv;
                    const bool head = valid && (wv >> L.bp) != (wp >> L.bp);
                    const bool last = pos + 1u >= e1 || (wn >> L.bp) != (wv >> L.bp);
                    const bool sing = head && last;
                    const unsigned long long sb_ = __ballot(sing);
                    if (sing) {
                        const u32 G = (r << gb) | (wv >> L.bp);
                        u32 gsubj;
                        int dlt;
                        if (BANDS) {
                            const uint2 be = btab[G >> L.bd];
                            gsubj = be.x;
                            dlt = (int)(be.y - G);   // sst - qpos
                        } else {
                            gsubj = G >> L.bd;
                            dlt = diag_off - (int)(G & ((1u << L.bd) - 1u));
                        }
                        const u32 slot = (rback + (u32)__popcll(sb_ & lt)) & (U1_RING - 1);
                        ring_s[slot] = roff[gsubj] + (u32)((int)(wv & pmask) + dlt);   // byte of (subject, sst) in r_ug
                        ring_w[slot] = wv;
                    }
                    rback += (u32)__popcll(sb_);
                    if (mlist) {   // heads of longer groups -> the chain list
                        const bool mh = head && !last;
                        const unsigned long long mb = __ballot(mh);
                        if (mb) {
                            if (lc_end - lc_pos < 64u) {   // (wave-uniform) the piece may not hold this scan's heads: leave its rest unused
                                for (u32 i = lc_pos + (u32)lane; i < lc_end; i += 64) mlist[i] = UG_REC_NONE;
                                u32 nbase = 0;
                                if (lane == 0) nbase = atomicAdd(mlist_cnt, (u32)U1_LCHUNK);
                                lc_pos = (u32)__builtin_amdgcn_readfirstlane((int)nbase);
                                lc_end = lc_pos + U1_LCHUNK;
                            }
                            if (mh) mlist[lc_pos + (u32)__popcll(mb & lt)] = (u64)pos | ((u64)b << 32);
                            lc_pos += (u32)__popcll(mb);
                        }
                    }
                    cur += 64;
                }
                u1_wave_sync();
                // ---- hand out ----
                const u32 avail = rback - rfront;
                if (!working) {
                    const u32 rk = (u32)__popcll(idleb & lt);
                    if (rk < avail) {
                        const u32 slot = (rfront + rk) & (U1_RING - 1);
                        const u32 sa = ring_s[slot];
                        hw = ring_w[slot];
                        const int qpos = (int)(hw & pmask);
                        qR = U1_QPAD + qpos, qL = qR - 16;
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
                if (cur >= e1 && rback == rfront) break;   // bucket done
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
            // ---- finished singletons: buffer the ones that reach MIN_UNGAP (hit word | score << 32; expanded when flushed) ----
            {
                const int score = (int)M.x + (int)M.y;
                bool todo = fin && score >= MIN_UNGAP;
                for (;;) {   // (one round unless more lanes pass in a step than the buffer has room for)
                    const unsigned long long pb = __ballot(todo);
                    if (!pb) break;
                    const u32 room = U1_PCAP - npb, rk = (u32)__popcll(pb & lt);
                    if (todo && rk < room) {
                        s_pass[npb + rk] = ((u64)(u32)score << 32) | hw;
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
        if (npb) {   // the buffered records belong to this bucket: their (query, band range) is wave-uniform here
            flush();
            u1_wave_sync();
        }
    }
    for (u32 i = ch_pos + (u32)lane; i < ch_end; i += 64) p_qs[i] = UG_REC_NONE;   // unused slots of the wave's last piece: k_rec_count / k_rec_scatter skip them
    if (mlist)
        for (u32 i = lc_pos + (u32)lane; i < lc_end; i += 64) mlist[i] = UG_REC_NONE;
    if (lane == 0 && ngroups) atomicAdd(&group_count[0], (unsigned long long)ngroups);
    if (COUNT) {
        unsigned long long nst = ct.n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nst += __shfl_xor(nst, o);
        if (lane == 0) atomicAdd(&stat[0], nst), atomicAdd(&stat[1], (unsigned long long)ngroups);
    }
}

// ================================================================================================================
// groups of two and more hits: chains (get_ungap_scores, fsearch.py:2497-2509)
// ================================================================================================================
// Input: the chain list k_ungap1 wrote (head position | bucket << 32).  Every later seed of a group is bounded on the left by the
// previous segment's right end `lo` = max_qed -- a seed at or in front of it adds nothing (off moves its start to lo, `qlo < qst`
// fails in both passes), one behind it runs its right pass freely and its left pass down to lo + 1.  What the singleton kernel does not
// need: (1) WHERE the right pass reached its maximum (max_qed: the FIRST position of the maximum) -- at every X-drop test point the group
// of three elements in which the maximum last rose is remembered with its three running scores; (2) the left bound -- the left window's
// bytes at positions <= lo are replaced by the sentinel class through a 16-byte mask read from a 17-entry LDS table; (3) groups of many
// buckets -- hence many queries -- share a wave, so the query side is read from global memory like the subject side (q_ug: the batch's
// score classes with position 0 of every query = sentinel).  The hits behind a seed are read four at a time, one step ahead.
#define U2_RING 128
struct U2Entry {   // 32 bytes
    u32 sa, w, pos, b, qb, e1, pad0, pad1;
};
template <bool BANDS, int TSH, bool COUNT>
__global__ __launch_bounds__(64 * U1_WAVES, 1) void k_ungap2(const u64* __restrict__ mlist, const u32* __restrict__ mlist_cnt, const u32* __restrict__ words,
                                                            const u32* __restrict__ bext, BktLayout L, int sh_qpos, int diag_off, int rbs, int rsh_subj, int rsh_diag,
                                                            int rdoff, const uint2* __restrict__ btab, u32 wait_n, const u8* __restrict__ q_ug, const u32* __restrict__ qoff,
                                                            const u8* __restrict__ r_ug, const u32* __restrict__ roff, const signed char* __restrict__ b62g,
                                                            u32* __restrict__ work_ctr, u32* __restrict__ shard_cnt, u64* __restrict__ p_qs, u64* __restrict__ p_sd,
                                                            u64* __restrict__ p_ft, unsigned long long* __restrict__ group_count,
                                                            unsigned long long* __restrict__ stat /*COUNT: [0] += b62 lookups, [2] += groups*/) {
    constexpr u32 TBYTES = (u32)U1_ROWS << (8 + TSH);
    constexpr u32 WBYTES = U2_RING * 32 + U1_PCAP * 16;
    __shared__ __align__(16) unsigned char u2_smem[TBYTES + 17 * 16 + U1_WAVES * WBYTES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u1_fill_table<TSH>(u2_smem, b62g);
    // s_lmask[n] = 16 bytes, the lowest n of them 0xFF (n = 0 ... 16)
    uint4* s_lmask = reinterpret_cast<uint4*>(u2_smem + TBYTES);
    if (threadIdx.x < 17u * 4u) {
        const int n = (int)threadIdx.x >> 2, d = (int)threadIdx.x & 3;
        const int nv = min(max(n - 4 * d, 0), 4);
        reinterpret_cast<u32*>(s_lmask)[threadIdx.x] = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
    }
    __syncthreads();
    unsigned char* wbase = u2_smem + TBYTES + 17 * 16 + (u32)w * WBYTES;
    U2Entry* ring = reinterpret_cast<U2Entry*>(wbase);
    uint4* s_pass = reinterpret_cast<uint4*>(wbase + U2_RING * 32);   // (head word, score, bucket, -)
    const unsigned long long lt = (1ull << lane) - 1ull;
    const u32 lanebase = (u32)(size_t)(__attribute__((address_space(3))) unsigned char*)u2_smem + ((u32)(lane & 31) << (TSH == 4 ? 2 : 1));
    const u32 WM = 0x7FFFFFFFu;
    const u32 pmask = (1u << L.bp) - 1u;
    const int gb = L.wb + L.bd;
    const u32 nlist = *mlist_cnt;
    u32 npb = 0, ngroups = 0;
    u32 ch_pos = 0, ch_end = 0;
    u32 rfront = 0, rback = 0;
    bool more = true;   // the list has unread entries
    u32 wk_pos = 0, wk_end = 0;   // the wave's piece of the list
    auto flush = [&]() {
        const u32 at = u1_reserve(npb, lane, ch_pos, ch_end, &shard_cnt[0], U1_CHUNK);
        if ((u32)lane < npb) {
            const uint4 e = s_pass[lane];
            const u32 r = e.z / L.nqp, gq = L.qa + (e.z - r * L.nqp);
            u1_record<BANDS>(e.x, (u64)e.y << 32, r, gq, L, diag_off, rbs, rsh_subj, rsh_diag, rdoff, sh_qpos, btab, p_qs, p_sd, p_ft, at);
        }
        npb = 0;
    };
    // lane state
    bool working = false;      // inside a segment (a seed's two passes)
    bool walking = false;      // between segments, looking for the group's next seed behind `lo`
    i64 qR = 0, qL = 0, sR = 0, sL = 0;   // byte offsets of the next windows in q_ug / r_ug
    i64 sa0 = 0, qa0 = 0;      // byte of the head hit in r_ug; byte of the query's position 0 in q_ug
    u32 hw = 0, hb = 0, he1 = 0;   // the group's head word, bucket, end of the bucket's words
    u32 h = 0;                 // position of the current seed's hit
    uint4 nx = make_uint4(0, 0, 0, 0);   // the four words behind it
    int scores = 0, lo = -1, Qst = 0, eb = 0;
    pk16 S = {0, 0}, M = {0, 0};
    U1Track tr;
    tr.Mprev = pk16{0, 0}, tr.sv1 = tr.sv2 = tr.sv3 = pk16{0, 0}, tr.gsel = 0;
    U1Count ct;
    ct.n = 0;
    ct.start();
    auto start_segment = [&](int qpos) {   // both passes of the seed at query position qpos, on the group's diagonal
        Qst = qpos;
        qR = qa0 + qpos, qL = qR - 16;
        sR = sa0 + (qpos - (int)(hw & pmask)), sL = sR - 16;
        S = pk16{0, 0}, M = pk16{0, 0};
        tr.Mprev = pk16{0, 0}, tr.gsel = 0, eb = 0;
        if (COUNT) ct.start();
        working = true;
    };
    for (;;) {
        const unsigned long long idleb = __ballot(!working && !walking);
        if (idleb && ((u32)__popcll(idleb) >= wait_n || !__ballot(working))) {
            // ---- refill the ring: 64 list entries, decoded ----
            while (more && rback - rfront <= U2_RING - 64u) {
                if (wk_pos == wk_end) {   // the wave's piece of the list is used up: the next one (one same-address atomic per 2048 entries)
                    u32 c0 = 0;
                    if (lane == 0) c0 = atomicAdd(work_ctr, 2048u);
                    wk_pos = (u32)__builtin_amdgcn_readfirstlane((int)c0);
                    wk_end = min(wk_pos + 2048u, nlist);
                    if (wk_pos >= nlist) {
                        wk_pos = wk_end = 0;
                        more = false;
                        break;
                    }
                }
                const u32 c0 = wk_pos, c1 = min(wk_pos + 64u, wk_end);
                wk_pos = c1;
                const u64 e = c0 + (u32)lane < c1 ? mlist[c0 + (u32)lane] : UG_REC_NONE;
                const bool valid = e != UG_REC_NONE;
                const unsigned long long vb = __ballot(valid);
                if (valid) {
                    const u32 pos = (u32)e, b = (u32)(e >> 32);
                    const u32 wv = words[pos] & WM;
                    const u32 r = b / L.nqp, gq = L.qa + (b - r * L.nqp);
                    const u32 G = (r << gb) | (wv >> L.bp);
                    u32 gsubj;
                    int dlt;
                    if (BANDS) {
                        const uint2 be = btab[G >> L.bd];
                        gsubj = be.x;
                        dlt = (int)(be.y - G);   // sst - qpos
                    } else {
                        gsubj = G >> L.bd;
                        dlt = diag_off - (int)(G & ((1u << L.bd) - 1u));
                    }
                    U2Entry en;
                    en.sa = roff[gsubj] + (u32)((int)(wv & pmask) + dlt);
                    en.w = wv, en.pos = pos, en.b = b, en.qb = qoff[gq], en.e1 = bext[b + 1], en.pad0 = en.pad1 = 0;
                    ring[(rback + (u32)__popcll(vb & lt)) & (U2_RING - 1)] = en;
                }
                rback += (u32)__popcll(vb);
            }
            u1_wave_sync();
            // ---- hand out ----
            const u32 avail = rback - rfront;
            if (!working && !walking) {
                const u32 rk = (u32)__popcll(idleb & lt);
                if (rk < avail) {
                    const U2Entry en = ring[(rfront + rk) & (U2_RING - 1)];
                    sa0 = (i64)en.sa, qa0 = (i64)en.qb;
                    hw = en.w, hb = en.b, he1 = en.e1, h = en.pos;
                    nx = u1_load16(reinterpret_cast<const u8*>(words + h + 1));   // (the array is readable 8 words past its end)
                    scores = 0, lo = -1;
                    start_segment((int)(hw & pmask));
                }
            }
            const u32 taken = min((u32)__popcll(idleb), avail);
            rfront += taken;
            ngroups += taken;
            u1_wave_sync();   // ring slots may be overwritten by the next refill only after these reads
        }
        if (!__ballot(working || walking)) {
            if (!more && rback == rfront) break;
            continue;
        }
        bool fin = false;      // the lane's group is complete
        if (working) {
            const uint4 qr4 = u1_load16(q_ug + qR), sr4 = u1_load16(r_ug + sR), sl4 = u1_load16(r_ug + sL);
            uint4 ql4 = u1_load16(q_ug + qL);
            {   // positions <= lo of the left window (query positions [qL - qa0, + 16)): sentinel
                const int ninv = min(max(lo + 1 - (int)(qL - qa0), 0), 16);
                const uint4 mk = s_lmask[ninv];
                ql4.x = (ql4.x & ~mk.x) | (0x18181818u & mk.x), ql4.y = (ql4.y & ~mk.y) | (0x18181818u & mk.y);
                ql4.z = (ql4.z & ~mk.z) | (0x18181818u & mk.z), ql4.w = (ql4.w & ~mk.w) | (0x18181818u & mk.w);
            }
            const u32 msk = u1_step<TSH, true, COUNT>(qr4, ql4, sr4, sl4, lanebase, S, M, tr, eb, ct);
            qR += 16, qL -= 16, sR += 16, sL -= 16, eb += 16;
            if (msk == 0xFFFFFFFFu) {   // both passes have ended
                // the segment's maximum joins the chain's score; the next seed is bounded by max_qed: the first position of the right
                // pass's maximum, the seed's own position when nothing it scored was positive (2463, 2468-2469)
                scores += (int)M.x + (int)M.y;
                const int am = tr.gsel + (tr.sv1.x == M.x ? 0 : (tr.sv2.x == M.x ? 1 : 2));
                lo = M.x > 0 ? Qst + am : Qst;
                working = false, walking = true;
            }
        }
        if (walking && !working) {
            // ---- the group's next seed behind lo, among the four hits read ahead (a seed at or in front of lo adds nothing) ----
            const u32* nxd = reinterpret_cast<const u32*>(&nx);
            bool found = false, ended = false;
            int qp = 0;
            u32 adv = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!found && !ended) {
                    const u32 c = nxd[i] & WM;
                    if (h + 1u + (u32)i >= he1 || (nxd[i] >> 31) != 0 || (c >> L.bp) != (hw >> L.bp)) ended = true;
                    else {
                        adv = (u32)i + 1u;
                        qp = (int)(c & pmask);
                        if (qp > lo) found = true;
                    }
                }
            }
            if (ended) {
                walking = false;
                fin = true;
            } else {
                h += found ? adv : 4u;
                nx = u1_load16(reinterpret_cast<const u8*>(words + h + 1));
                if (found) {
                    walking = false;
                    start_segment(qp);
                }
            }
        }
        // ---- finished groups: buffer the ones that reach MIN_UNGAP (head word, score, bucket; expanded when flushed) ----
        {
            bool todo = fin && scores >= MIN_UNGAP;
            for (;;) {   // (one round unless more lanes pass in a step than the buffer has room for)
                const unsigned long long pb = __ballot(todo);
                if (!pb) break;
                const u32 room = U1_PCAP - npb, rk = (u32)__popcll(pb & lt);
                if (todo && rk < room) {
                    s_pass[npb + rk] = make_uint4(hw, (u32)scores, hb, 0u);
                    todo = false;
                }
                npb += min((u32)__popcll(pb), room);
                u1_wave_sync();
                if (npb == U1_PCAP) {
                    flush();
                    u1_wave_sync();
                }
            }
        }
    }
    if (npb) flush();
    for (u32 i = ch_pos + (u32)lane; i < ch_end; i += 64) p_qs[i] = UG_REC_NONE;
    if (lane == 0 && ngroups) atomicAdd(&group_count[0], (unsigned long long)ngroups);
    if (COUNT) {
        unsigned long long nst = ct.n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nst += __shfl_xor(nst, o);
        if (lane == 0) atomicAdd(&stat[0], nst), atomicAdd(&stat[2], (unsigned long long)ngroups);
    }
}

u32 ungap1_qcap() { return U1_QCAP; }
size_t ungap1_list_slack(u32 ncu) { return (size_t)ncu * 3 * U1_WAVES * U1_CHUNK; }
size_t ungap1_mlist_cap(u32 H, u32 ncu) { return (size_t)H / 2 + (size_t)ncu * 2 * U1_WAVES * U1_LCHUNK + (size_t)H / 16 + 64; }   // (+ the entries pieces leave unused: < 64 per 1024)

template <bool BANDS, int TSH, int QCAP, int WGS, bool COUNT>
static void ungap1_launch(u32 ncu, const u32* words, const u32* bext, u32 nb, const BktLayout& L, const KeyLayout& kl, const KeyLayout& klr, const void* btab,
                          u32 wait_n, const u8* q_scls, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g, u32* work_ctr, u32* shard_cnt,
                          u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, u64* mlist, u32* mlist_cnt, unsigned long long* stat, hipStream_t st) {
    static bool said = false;
    if (!said && tune().debug) {
        int nblk = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_ungap1<BANDS, TSH, QCAP, WGS, COUNT>, 64 * U1_WAVES, 0);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_ungap1<BANDS, TSH, QCAP, WGS, COUNT>));
        fprintf(stderr, "[sohit] k_ungap1<%d,%d,%d,%d>: occupancy query %d blocks per CU (err %d), static LDS %zu, regs %d\n", (int)BANDS, TSH, QCAP, WGS, nblk, (int)e,
                fa.sharedSizeBytes, fa.numRegs);
        said = true;
    }
    static_assert(WGS * (((size_t)U1_ROWS << (8 + TSH)) + (size_t)U1_WAVES * U1_WAVE_BYTES(QCAP, TSH)) <= 160 * 1024, "LDS of a CU");
    hipLaunchKernelGGL((k_ungap1<BANDS, TSH, QCAP, WGS, COUNT>), dim3(ncu * WGS), dim3(64 * U1_WAVES), 0, st, words, bext, nb, L, kl.sh_qpos, (int)kl.diag_off, klr.bs,
                       klr.sh_subj, klr.sh_diag, (int)klr.diag_off, (const uint2*)btab, wait_n, q_scls, qoff, r_ug, roff, b62g, work_ctr, shard_cnt, p_qs, p_sd, p_ft,
                       group_count, mlist, mlist_cnt, stat);
}

// 16-bit table entries; 512-residue query slots and two workgroups per CU (8 waves per SIMD) for the passes whose queries fit, else one
// workgroup with 1024-residue or U1_QCAP slots.  (An instance with 32-bit entries -- every lane of a 32-lane group its own bank -- measured
// slower at its one workgroup per CU, round 5.)
void launch_ungap1(u32 ncu, u32 pmaxq, const u32* words, const u32* bext, u32 nb, const BktLayout& L, const KeyLayout& kl, const KeyLayout& klr,
                   const void* btab, u32 wait_n, const u8* q_scls, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g, u32* work_ctr, u32* shard_cnt,
                   u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, u64* mlist, u32* mlist_cnt, unsigned long long* stat, hipStream_t st) {
    if (!nb) return;
#define U1_GO(B, T, Q, W) ungap1_launch<B, T, Q, W, false>(ncu, words, bext, nb, L, kl, klr, btab, wait_n, q_scls, qoff, r_ug, roff, b62g, work_ctr, shard_cnt, p_qs, p_sd, p_ft, group_count, mlist, mlist_cnt, nullptr, st)
#define U1_GOC(B) ungap1_launch<B, 3, U1_QCAP, 1, true>(ncu, words, bext, nb, L, kl, klr, btab, wait_n, q_scls, qoff, r_ug, roff, b62g, work_ctr, shard_cnt, p_qs, p_sd, p_ft, group_count, mlist, mlist_cnt, stat, st)
    if (stat) {   // counting instance
        if (btab) U1_GOC(true);
        else U1_GOC(false);
        return;
    }
    if (pmaxq <= 512) {
        if (btab) U1_GO(true, 3, 512, 2);
        else U1_GO(false, 3, 512, 2);
    } else if (pmaxq <= 1024) {
        if (btab) U1_GO(true, 3, 1024, 1);
        else U1_GO(false, 3, 1024, 1);
    } else {
        if (btab) U1_GO(true, 3, U1_QCAP, 1);
        else U1_GO(false, 3, U1_QCAP, 1);
    }
#undef U1_GO
#undef U1_GOC
}

void launch_ungap2(u32 ncu, const u64* mlist, const u32* mlist_cnt, const u32* words, const u32* bext, const BktLayout& L, const KeyLayout& kl, const KeyLayout& klr,
                   const void* btab, u32 wait_n, const u8* q_ug, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g, u32* work_ctr, u32* shard_cnt,
                   u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, unsigned long long* stat, hipStream_t st) {
#define U2_GO(B, CT) hipLaunchKernelGGL((k_ungap2<B, 3, CT>), dim3(ncu), dim3(64 * U1_WAVES), 0, st, mlist, mlist_cnt, words, bext, L, kl.sh_qpos, (int)kl.diag_off, klr.bs, klr.sh_subj, \
                                    klr.sh_diag, (int)klr.diag_off, (const uint2*)btab, wait_n, q_ug, qoff, r_ug, roff, b62g, work_ctr, shard_cnt, p_qs, p_sd, p_ft, group_count, stat)
    if (stat) {
        if (btab) U2_GO(true, true);
        else U2_GO(false, true);
    } else if (btab) U2_GO(true, false);
    else U2_GO(false, false);
#undef U2_GO
}
