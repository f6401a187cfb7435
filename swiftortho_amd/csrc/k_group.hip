// k_group.hip -- diagonal binning and chained ungapped X-drop extension
// (fsearch.py:2679-2719: hits dict keyed (subject, qst - sst), qsort + lis per group,
// get_ungap_scores 2497-2509, ungap 2454-2494, best diagonal per subject, guess_start 2544-2553).
//
// Input: the seed-hit keys of one (query batch, chunk) sorted ascending, so that the hits of one
// (query, subject, diagonal) group are contiguous and ordered by query position.  qsort-by-qst +
// lis-by-sst of a single-diagonal group == its distinct query positions in ascending order.
#include "common.h"
#include "kernels.h"

// ---- group heads ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_flags(const u64* __restrict__ keys, u32 H, KeyLayout kl, u32* __restrict__ flags,
                                                     u32* __restrict__ hvalid) {
    const u32 h = blockIdx.x * 256u + threadIdx.x;
    if (h >= H) return;
    const u64 mask = (kl.total >= 64) ? ~0ull : ((1ull << kl.total) - 1ull);
    const u64 qall = (1ull << kl.bq) - 1ull;
    const u64 k = keys[h] & mask;
    const bool valid = (k >> kl.sh_q) != qall;
    bool head = false;
    if (valid) {
        if (h == 0) head = true;
        else head = ((keys[h - 1] & mask) >> kl.sh_diag) != (k >> kl.sh_diag);
        bool last = (h + 1 == H) || (((keys[h + 1] & mask) >> kl.sh_q) == qall);
        if (last) *hvalid = h + 1;
    }
    flags[h] = head ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_group_list(const u32* __restrict__ flags, const u32* __restrict__ gidx, u32 H,
                                                    u32* __restrict__ ghead) {
    const u32 h = blockIdx.x * 256u + threadIdx.x;
    if (h >= H) return;
    if (flags[h]) ghead[gidx[h]] = h;
}

// ---- chained ungapped extension -----------------------------------------------------------------
struct UG {
    int max_score, max_qst, max_qed, max_sst, max_sed;
};

#define B62_LD 36  // LDS score table: 32 rows of 36 bytes (9 dwords: rows rotate through all 32 banks); any 5-bit class pair indexes inside it

__device__ __forceinline__ u64 load8u(const u8* p) {  // unaligned 8-byte global load
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

// Fasta.ungap (fsearch.py:2454-2494); qlo/slo already resolved to >= 0.
// The reference's per-residue loops are evaluated in chunks of 8 residues: two unaligned 8-byte
// loads, eight LDS score lookups issued back to back, and predicated (branch-free) max / X-drop
// updates.  Scores, maxima and end points are exactly those of the sequential loops: an element
// is applied only while `k < remaining && !stopped`.  qabs/sabs = absolute offsets of the two
// sequences inside their (padded) class arrays, used to keep the left-pass loads in bounds.
__device__ __forceinline__ UG ungap_dev(const u8* __restrict__ q, int ql, i64 qabs, const u8* __restrict__ s, int sl, i64 sabs, int Qst,
                                        int Sst, int qlo, int slo, const signed char* b62c) {
    const int off = max(max(qlo - Qst, slo - Sst), 0);
    Qst += off;
    Sst += off;
    int max_score = 0, score = 0, best = -1;
    bool stop = false;
    // right pass: t-th step scores (Qst + t, Sst + t) while qlo < qst < ql and slo < sst < sl
    int n = (qlo < Qst && slo < Sst) ? min(ql - Qst, sl - Sst) : 0;
    for (int i = 0; i < n && !stop; i += 8) {
        const u64 qw = load8u(q + Qst + i), sw = load8u(s + Sst + i);
        const int m = n - i;
        int sc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sc[k] = b62c[(((u32)(qw >> (8 * k)) & 31u) * B62_LD) + ((u32)(sw >> (8 * k)) & 31u)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool act = (k < m) && !stop;
            const int ns = score + sc[k];
            const bool better = act && (ns > max_score);
            stop = stop || (act && !better && (ns + DROPX < max_score));
            score = act ? ns : score;
            max_score = better ? ns : max_score;
            best = better ? i + k : best;
        }
    }
    const int max_qed = best >= 0 ? Qst + best : Qst, max_sed = best >= 0 ? Sst + best : Sst;
    // left pass from (Qst - 1, Sst - 1), score continues from the maximum
    score = max_score;
    stop = false;
    best = -1;
    n = (Qst - 1 < ql && Sst - 1 < sl) ? min(Qst - 1 - qlo, Sst - 1 - slo) : 0;
    for (int i = 0; i < n && !stop; i += 8) {
        // bytes [p - 7, p] with p = Qst - 1 - i; element k lives in byte 7 - k
        i64 qa = (i64)Qst - 8 - i, sa = (i64)Sst - 8 - i;
        u64 qw, sw;
        if (qabs + qa >= 0) qw = load8u(q + qa);
        else qw = load8u(q - qabs) << (8 * (int)(-(qabs + qa)));  // array start: missing low bytes are never active
        if (sabs + sa >= 0) sw = load8u(s + sa);
        else sw = load8u(s - sabs) << (8 * (int)(-(sabs + sa)));
        const int m = n - i;
        int sc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sc[k] = b62c[(((u32)(qw >> (8 * (7 - k))) & 31u) * B62_LD) + ((u32)(sw >> (8 * (7 - k))) & 31u)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool act = (k < m) && !stop;
            const int ns = score + sc[k];
            const bool better = act && (ns > max_score);
            stop = stop || (act && !better && (ns + DROPX < max_score));
            score = act ? ns : score;
            max_score = better ? ns : max_score;
            best = better ? i + k : best;
        }
    }
    const int max_qst = best >= 0 ? Qst - 1 - best : Qst - 1, max_sst = best >= 0 ? Sst - 1 - best : Sst - 1;
    return {max_score, max_qst, max_qed, max_sst, max_sed};
}

// One thread per (query, subject, diagonal) group.  Passing groups (score >= 25) are appended to
// the pass list with a wave-ballot compaction: one atomicAdd per wave, lanes take consecutive slots.
//   p_qs[i] = (q << 32) | subject_local     p_sd[i] = (score << 32) | (u32)dist      p_ft[i] = first-touch key
__global__ __launch_bounds__(256) void k_ungap(const u64* __restrict__ keys, const u32* __restrict__ ghead, u32 G, u32 Hvalid,
                                               KeyLayout kl, int ft_bits_entry, int bsp, const u8* __restrict__ q_scls,
                                               const u32* __restrict__ qoff, const u8* __restrict__ r_scls,
                                               const u32* __restrict__ roff /*chunk-local offsets (absolute values)*/,
                                               const signed char* __restrict__ b62g, u32* __restrict__ shard_cnt /*[UG_SHARDS]*/,
                                               u32 shard_cap, u64* __restrict__ p_qs, u64* __restrict__ p_sd, u64* __restrict__ p_ft,
                                               unsigned long long* __restrict__ step_shards /*[UG_SHARDS]*/) {
    __shared__ signed char s_b62[32 * B62_LD];
    __shared__ u32 s_wcnt[4];
    __shared__ u32 s_base;
    for (int i = threadIdx.x; i < 32 * B62_LD; i += 256) {
        const int a = i / B62_LD, b = i % B62_LD;
        s_b62[i] = (a < SCLS_N && b < SCLS_N) ? b62g[a * SCLS_N + b] : (signed char)-4;
    }
    __syncthreads();
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    bool pass = false;
    u64 o_qs = 0, o_sd = 0, o_ft = 0;
    if (g < G) {
        const u32 h0 = ghead[g], h1 = (g + 1 < G) ? ghead[g + 1] : Hvalid;
        const u64 k0 = keys[h0];
        const u32 q = (u32)((k0 >> kl.sh_q) & ((1ull << kl.bq) - 1ull));
        const u32 subj = (u32)((k0 >> kl.sh_subj) & ((1ull << kl.bs) - 1ull));
        const i64 diag = (i64)((k0 >> kl.sh_diag) & ((1ull << kl.bd) - 1ull)) - kl.diag_off;  // qpos - sst
        const u32 qb = qoff[q], sb = roff[subj];
        const int ql = (int)(qoff[q + 1] - qb), sl = (int)(roff[subj + 1] - sb);
        const u8* qs = q_scls + qb;
        const u8* ss = r_scls + sb;
        const u64 pmask = (1ull << kl.bp) - 1ull, amask = (1ull << kl.ba) - 1ull;
        int prev_qpos = -1;
        int scores = 0, x0 = 0, y0 = 0, x = 0, y = 0;
        bool first = true;
        u64 ft = ~0ull;
        for (u32 h = h0; h < h1; ++h) {
            const u64 k = keys[h];
            const int qpos = (int)((k >> kl.sh_qpos) & pmask);
            const u32 as = kl.ba ? (u32)((k >> kl.sh_as) & amask) : 0u;
            const u32 tag = kl.ba ? (u32)(k & amask) : 0u;
            const int sst = (int)((i64)qpos - diag);
            // first-touch key: emission order (as, qpos) ascending, then index slot order ==
            // descending (true subject j, tag, pos)
            {
                u32 j = subj, pos = (u32)sst;
                if (sst == sl) j = subj + 1, pos = 0;  // offset-0 entry of the next chunk sequence
                const u64 jmax = (1ull << (kl.bs + 1)) - 1ull, tmax = amask, pmax = (1ull << bsp) - 1ull;
                u64 inv = ((jmax - j) << (kl.ba + bsp)) | ((tmax - tag) << bsp) | (pmax - pos);
                u64 emit = ((u64)as << kl.bp) | (u64)qpos;
                u64 f = (emit << ft_bits_entry) | inv;
                ft = f < ft ? f : ft;
            }
            if (qpos == prev_qpos) continue;  // duplicate (qst, sst) pair: dropped by lis()
            prev_qpos = qpos;
            if (first) {
                UG u = ungap_dev(qs, ql, (i64)qb, ss, sl, (i64)sb, qpos, sst, 0, 0, s_b62);
                scores = u.max_score, x0 = u.max_qst, y0 = u.max_sst, x = u.max_qed, y = u.max_sed;
                first = false;
            } else {
                UG u = ungap_dev(qs, ql, (i64)qb, ss, sl, (i64)sb, qpos, sst, x, y, s_b62);
                scores += u.max_score, x = u.max_qed, y = u.max_sed;
            }
        }
        if (scores >= MIN_UNGAP) {
            pass = true;
            // guess_start over [[x0, y0], [x, y]]: floor(((y0 - x0) + (y - x)) / 2)
            int d2 = (y0 - x0) + (y - x);
            int dist = (d2 >= 0) ? d2 / 2 : -((-d2 + 1) / 2);
            o_qs = ((u64)q << 32) | subj;
            o_sd = ((u64)(u32)scores << 32) | (u64)(u32)dist;
            o_ft = ft;
        }
    }
    // Wave-ballot compaction, aggregated per block, into one of UG_SHARDS regions: a single global
    // counter would serialise ~3 M same-address atomics (one per wave) and dominated this kernel;
    // with one atomic per block spread over 64 addresses the tail is free.  Shard s owns slots
    // [s * shard_cap, (s + 1) * shard_cap); k_compact_shards makes the list contiguous afterwards.
    const unsigned long long bal = __ballot(pass);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_wcnt[w] = (u32)__popcll(bal);
    __syncthreads();
    const u32 shard = blockIdx.x & (UG_SHARDS - 1);
    if (threadIdx.x == 0) {
        const u32 tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        s_base = tot ? atomicAdd(&shard_cnt[shard], tot) : 0u;
    }
    __syncthreads();
    if (pass) {
        u32 i = s_base + (u32)__popcll(bal & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; ++k) i += s_wcnt[k];
        const size_t o = (size_t)shard * shard_cap + i;
        p_qs[o] = o_qs, p_sd[o] = o_sd, p_ft[o] = o_ft;
    }
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
__global__ __launch_bounds__(256) void k_seg_flags(const u64* __restrict__ sorted_qs, u32 n, u32* __restrict__ flags) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || sorted_qs[i] != sorted_qs[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_best(const u64* __restrict__ sorted_qs, const u32* __restrict__ idx, const u32* __restrict__ shead,
                                              u32 nseg, u32 n, const u64* __restrict__ p_sd, const u64* __restrict__ p_ft, u32 seq_lo,
                                              u64* __restrict__ c_ft, u32* __restrict__ c_q, u32* __restrict__ c_rec /*4 per cand*/) {
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
    c_q[s] = (u32)(qs >> 32);
    u32 qi, qj;
    if (bdist > 0) qi = 0, qj = (u32)bdist;
    else qi = (u32)(-bdist), qj = 0;
    c_rec[4 * s + 0] = (u32)qs + seq_lo;  // global subject id
    c_rec[4 * s + 1] = bscore;
    c_rec[4 * s + 2] = qi;
    c_rec[4 * s + 3] = qj;
}

// gather helpers for the two-pass (ft, then stable q) ordering
__global__ __launch_bounds__(256) void k_gather_u32_as_u64(const u32* __restrict__ src, const u32* __restrict__ idx, u32 n,
                                                           u64* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ __launch_bounds__(256) void k_combine_q_ft(const u32* __restrict__ c_q, const u64* __restrict__ c_ft, u32 n, int ftbits,
                                                      u64* __restrict__ dst) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = ((u64)c_q[i] << ftbits) | c_ft[i];
}

__global__ __launch_bounds__(256) void k_iota(u32* __restrict__ p, u32 n) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) p[i] = i;
}

// final order -> chunk candidate region; the region is sorted by query, so per-query counts are
// segment lengths: the thread at a segment's last element knows them without atomics.
__global__ __launch_bounds__(256) void k_emit_cands(const u32* __restrict__ order, u32 n, const u32* __restrict__ c_q,
                                                    const u32* __restrict__ c_rec, u32* __restrict__ out_q, u32* __restrict__ out_rec,
                                                    u32* __restrict__ seg_first) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u32 r = order[i];
    const u32 q = c_q[r];
    out_q[i] = q;
    const uint4 v = *reinterpret_cast<const uint4*>(c_rec + 4 * (size_t)r);
    *reinterpret_cast<uint4*>(out_rec + 4 * (size_t)i) = v;
    if (i == 0 || c_q[order[i - 1]] != q) seg_first[q] = i;
}

__global__ __launch_bounds__(256) void k_seg_counts(const u32* __restrict__ out_q, u32 n, const u32* __restrict__ seg_first,
                                                    u32* __restrict__ qcnt) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const u32 q = out_q[i];
    if (i + 1 == n || out_q[i + 1] != q) qcnt[q] = i + 1 - seg_first[q];
}

// ---- launch wrappers -------------------------------------------------------------------------------
void launch_group_flags(const u64* keys, u32 H, const KeyLayout& kl, u32* flags, u32* hvalid, hipStream_t st) {
    HIP_CHECK(hipMemsetAsync(hvalid, 0, sizeof(u32), st));
    if (!H) return;
    hipLaunchKernelGGL(k_group_flags, dim3((H + 255) / 256), dim3(256), 0, st, keys, H, kl, flags, hvalid);
}

void launch_group_list(const u32* flags, const u32* gidx, u32 H, u32* ghead, hipStream_t st) {
    if (!H) return;
    hipLaunchKernelGGL(k_group_list, dim3((H + 255) / 256), dim3(256), 0, st, flags, gidx, H, ghead);
}

u32 ungap_shard_cap(u32 G) {
    const u32 nblk = (G + 255) / 256;
    return ((nblk + UG_SHARDS - 1) / UG_SHARDS) * 256u;
}

void launch_ungap(const u64* keys, const u32* ghead, u32 G, u32 Hvalid, const KeyLayout& kl, int ft_bits_entry, int bsp,
                  const u8* q_scls, const u32* qoff, const u8* r_scls, const u32* roff, const signed char* b62g, u32* shard_cnt,
                  u32 shard_cap, u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* step_shards, hipStream_t st) {
    if (!G) return;
    hipLaunchKernelGGL(k_ungap, dim3((G + 255) / 256), dim3(256), 0, st, keys, ghead, G, Hvalid, kl, ft_bits_entry, bsp, q_scls, qoff,
                       r_scls, roff, b62g, shard_cnt, shard_cap, p_qs, p_sd, p_ft, step_shards);
}

void launch_shard_scan(const u32* shard_cnt, u32* shard_off, hipStream_t st) {
    hipLaunchKernelGGL(k_shard_scan, dim3(1), dim3(64), 0, st, shard_cnt, shard_off);
}

void launch_compact_shards(const u32* shard_cnt, u32* shard_off, u32 shard_cap, const u64* a0, const u64* a1, const u64* a2, u64* b0,
                           u64* b1, u64* b2, hipStream_t st) {
    hipLaunchKernelGGL(k_compact_shards, dim3(64, UG_SHARDS), dim3(256), 0, st, shard_cnt, shard_off, shard_cap, a0, a1, a2, b0, b1, b2);
}

void launch_seg_flags(const u64* sorted_qs, u32 n, u32* flags, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_seg_flags, dim3((n + 255) / 256), dim3(256), 0, st, sorted_qs, n, flags);
}

void launch_best(const u64* sorted_qs, const u32* idx, const u32* shead, u32 nseg, u32 n, const u64* p_sd, const u64* p_ft,
                 u32 seq_lo, u64* c_ft, u32* c_q, u32* c_rec, hipStream_t st) {
    if (!nseg) return;
    hipLaunchKernelGGL(k_best, dim3((nseg + 255) / 256), dim3(256), 0, st, sorted_qs, idx, shead, nseg, n, p_sd, p_ft, seq_lo, c_ft,
                       c_q, c_rec);
}

void launch_gather_u32_as_u64(const u32* src, const u32* idx, u32 n, u64* dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather_u32_as_u64, dim3((n + 255) / 256), dim3(256), 0, st, src, idx, n, dst);
}

void launch_combine_q_ft(const u32* c_q, const u64* c_ft, u32 n, int ftbits, u64* dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_combine_q_ft, dim3((n + 255) / 256), dim3(256), 0, st, c_q, c_ft, n, ftbits, dst);
}

void launch_iota(u32* p, u32 n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
}

void launch_emit_cands(const u32* order, u32 n, const u32* c_q, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                       u32* seg_first, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_emit_cands, dim3((n + 255) / 256), dim3(256), 0, st, order, n, c_q, c_rec, out_q, out_rec, seg_first);
    hipLaunchKernelGGL(k_seg_counts, dim3((n + 255) / 256), dim3(256), 0, st, out_q, n, seg_first, qcnt);
}
