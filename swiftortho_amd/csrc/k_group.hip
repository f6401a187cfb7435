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
    int max_score, max_qst, max_qed, max_sst, max_sed, steps;
};

// Fasta.ungap (fsearch.py:2454-2494); qlo/slo already resolved to >= 0
__device__ __forceinline__ UG ungap_dev(const u8* __restrict__ q, int ql, const u8* __restrict__ s, int sl, int Qst, int Sst, int qlo,
                                        int slo, const signed char* b62c) {
    int off = max(max(qlo - Qst, slo - Sst), 0);
    Qst += off;
    Sst += off;
    int qst = Qst, sst = Sst;
    int score = 0, max_score = 0, max_qed = qst, max_sed = sst, steps = 0;
    while (qlo < qst && qst < ql && slo < sst && sst < sl) {
        ++steps;
        score += b62c[q[qst] * SCLS_N + s[sst]];
        if (score > max_score) max_score = score, max_qed = qst, max_sed = sst;
        else if (score + DROPX < max_score) break;
        ++qst, ++sst;
    }
    qst = Qst - 1, sst = Sst - 1;
    score = max_score;
    int max_qst = qst, max_sst = sst;
    while (ql > qst && qst > qlo && sl > sst && sst > slo) {
        ++steps;
        score += b62c[q[qst] * SCLS_N + s[sst]];
        if (score > max_score) max_score = score, max_qst = qst, max_sst = sst;
        else if (score + DROPX < max_score) break;
        --qst, --sst;
    }
    return {max_score, max_qst, max_qed, max_sst, max_sed, steps};
}

// One thread per (query, subject, diagonal) group.  Passing groups (score >= 25) are appended to
// the pass list with a wave-ballot compaction: one atomicAdd per wave, lanes take consecutive slots.
//   p_qs[i] = (q << 32) | subject_local     p_sd[i] = (score << 32) | (u32)dist      p_ft[i] = first-touch key
__global__ __launch_bounds__(256) void k_ungap(const u64* __restrict__ keys, const u32* __restrict__ ghead, u32 G, u32 Hvalid,
                                               KeyLayout kl, int ft_bits_entry, int bsp, const u8* __restrict__ q_scls,
                                               const u32* __restrict__ qoff, const u8* __restrict__ r_scls,
                                               const u32* __restrict__ roff /*chunk-local offsets (absolute values)*/,
                                               const signed char* __restrict__ b62g, u32* __restrict__ pass_count,
                                               u64* __restrict__ p_qs, u64* __restrict__ p_sd, u64* __restrict__ p_ft,
                                               unsigned long long* __restrict__ step_count) {
    __shared__ signed char s_b62[SCLS_N * SCLS_N];
    for (int i = threadIdx.x; i < SCLS_N * SCLS_N; i += 256) s_b62[i] = b62g[i];
    __syncthreads();
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    bool pass = false;
    u64 o_qs = 0, o_sd = 0, o_ft = 0;
    int steps_total = 0;
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
                UG u = ungap_dev(qs, ql, ss, sl, qpos, sst, 0, 0, s_b62);
                scores = u.max_score, x0 = u.max_qst, y0 = u.max_sst, x = u.max_qed, y = u.max_sed;
                steps_total += u.steps;
                first = false;
            } else {
                UG u = ungap_dev(qs, ql, ss, sl, qpos, sst, x, y, s_b62);
                scores += u.max_score, x = u.max_qed, y = u.max_sed;
                steps_total += u.steps;
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
    // wave-ballot compaction
    const unsigned long long bal = __ballot(pass);
    const int lane = threadIdx.x & 63;
    u32 base = 0;
    if (bal) {
        if (lane == (int)__ffsll((unsigned long long)bal) - 1) base = atomicAdd(pass_count, (u32)__popcll(bal));
        base = __shfl(base, __ffsll((unsigned long long)bal) - 1);
        if (pass) {
            u32 i = base + (u32)__popcll(bal & ((1ull << lane) - 1ull));
            p_qs[i] = o_qs, p_sd[i] = o_sd, p_ft[i] = o_ft;
        }
    }
    // work counter (b62 lookups == the reference's `flag`)
    for (int o = 32; o > 0; o >>= 1) steps_total += __shfl_down(steps_total, o);
    if (lane == 0 && steps_total) atomicAdd(step_count, (unsigned long long)steps_total);
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

void launch_ungap(const u64* keys, const u32* ghead, u32 G, u32 Hvalid, const KeyLayout& kl, int ft_bits_entry, int bsp,
                  const u8* q_scls, const u32* qoff, const u8* r_scls, const u32* roff, const signed char* b62g, u32* pass_count,
                  u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* step_count, hipStream_t st) {
    if (!G) return;
    hipLaunchKernelGGL(k_ungap, dim3((G + 255) / 256), dim3(256), 0, st, keys, ghead, G, Hvalid, kl, ft_bits_entry, bsp, q_scls, qoff,
                       r_scls, roff, b62g, pass_count, p_qs, p_sd, p_ft, step_count);
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
