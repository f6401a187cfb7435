// k_prep.hip -- sequence-set layout kernels: separator-padded hash-class stream packed to
// 5 bits per residue (seeding stages) and the 1-byte score-class stream (extension stages).
//
// Packed stream of a sequence set with offsets off[0..n]: sequence j occupies packed positions
// [off[j] + j, off[j+1] + j) followed by ONE separator (class 30) at off[j+1] + j.  A seed window
// is valid iff it covers no class >= 30, which implements both "window fits in the sequence"
// (fsearch.py:534) and "no x/X anywhere in the span" (538-540) without a boundary lookup.
#include "common.h"
#include "kernels.h"

// one thread per packed position: owner sequence (binary search) + hash class
__global__ __launch_bounds__(256) void k_layout(const u8* __restrict__ res, const u32* __restrict__ off, u32 nseq, u32 P /*=nres+nseq*/,
                                                u32 Ppad, const u8* __restrict__ hmap /*256*/, u32* __restrict__ pseq,
                                                u8* __restrict__ pcls) {
    __shared__ u8 s_hmap[256];
    s_hmap[threadIdx.x] = hmap[threadIdx.x];
    __syncthreads();
    u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= Ppad) return;
    if (p >= P) {
        pcls[p] = HCLS_SEP;
        if (pseq) pseq[p] = nseq;
        return;
    }
    // largest j with off[j] + j <= p
    u32 lo = 0, hi = nseq;  // invariant: poff(lo) <= p < poff(hi) where poff(nseq) = P
    while (hi - lo > 1) {
        u32 m = (lo + hi) >> 1;
        if (off[m] + m <= p) lo = m;
        else hi = m;
    }
    u32 pos = p - (off[lo] + lo), len = off[lo + 1] - off[lo];
    if (pseq) pseq[p] = lo;
    pcls[p] = (pos == len) ? (u8)HCLS_SEP : s_hmap[res[off[lo] + pos]];
}

// one thread per 32 positions -> five 32-bit words (position p at bits [5p, 5p+5) of the stream)
__global__ __launch_bounds__(256) void k_pack5(const u8* __restrict__ pcls, u32 Ppad /*multiple of 32*/, u32* __restrict__ words) {
    u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g * 32u >= Ppad) return;
    const u8* c = pcls + (size_t)g * 32;
    u64 acc = 0;
    int nb = 0, w = 0;
    u32 out[5];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        acc |= (u64)(c[k] & 31u) << nb;
        nb += 5;
        if (nb >= 32) {
            out[w++] = (u32)acc;
            acc >>= 32;
            nb -= 32;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) words[(size_t)g * 5 + k] = out[k];
}

// raw byte -> BLOSUM62 score class; scls4 (optional) = class * 4, the column index of k_ungap's LDS score table
__global__ __launch_bounds__(256) void k_scls(const u8* __restrict__ res, size_t n, const u8* __restrict__ smap, u8* __restrict__ scls,
                                              u8* __restrict__ scls4) {
    __shared__ u8 s_map[256];
    s_map[threadIdx.x] = smap[threadIdx.x];
    __syncthreads();
    // the 16 bytes in front of and the 64 bytes behind both arrays belong to them (k_ungap's windows reach up to 8 bytes
    // past either end): class 0, so that no lookup offset ever depends on what the allocation held before
    if (blockIdx.x == 0 && threadIdx.x < SCLS_PAD_FRONT + SCLS_PAD_BACK) {
        const ptrdiff_t o = threadIdx.x < SCLS_PAD_FRONT ? (ptrdiff_t)threadIdx.x - SCLS_PAD_FRONT : (ptrdiff_t)(n + threadIdx.x - SCLS_PAD_FRONT);
        scls[o] = 0;
        if (scls4) scls4[o] = 0;
    }
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        uchar4 v = *reinterpret_cast<const uchar4*>(res + i);
        uchar4 o = make_uchar4(s_map[v.x], s_map[v.y], s_map[v.z], s_map[v.w]);
        *reinterpret_cast<uchar4*>(scls + i) = o;
        if (scls4) *reinterpret_cast<uchar4*>(scls4 + i) = make_uchar4(o.x << 2, o.y << 2, o.z << 2, o.w << 2);
    } else {
        for (int k = 0; k < 4; ++k)
            if (i + k < n) {
                scls[i + k] = s_map[res[i + k]];
                if (scls4) scls4[i + k] = (u8)(s_map[res[i + k]] << 2);
            }
    }
}

// ---- class arrays with SENTINELS behind every sequence (the packed aligner's, k_align16.hip) ------------------------------------
// Sequence s's classes at off[s] + PCLS_PAD * s, followed by PCLS_PAD bytes of the sentinel class (24; * 4 in the column array); the
// two arrays point PCLS_PAD bytes into their allocations (sentinels in front of sequence 0 too).  A band that runs off the END of a
// sequence then reads sentinels where the plain arrays hold the next sequence: the aligner's last groups need no window masks.
__global__ __launch_bounds__(256) void k_pad_cls(const u8* __restrict__ scls, const u32* __restrict__ off, u32 nseq, u8* __restrict__ out,
                                                 u8* __restrict__ out4) {
    const u32 s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s == 0 && lane < PCLS_PAD) out[(ptrdiff_t)lane - PCLS_PAD] = 24, out4[(ptrdiff_t)lane - PCLS_PAD] = 96;
    if (s >= nseq) return;
    const u32 a = off[s], e = off[s + 1];
    u8* o = out + a + (size_t)PCLS_PAD * s;
    u8* o4 = out4 + a + (size_t)PCLS_PAD * s;
    for (u32 i = lane; i < e - a; i += 64) {
        const u8 c = scls[a + i];
        o[i] = c, o4[i] = (u8)(c << 2);
    }
    if (lane < PCLS_PAD) o[e - a + lane] = 24, o4[e - a + lane] = 96;
}
void launch_pad_cls(const u8* scls, const u32* off, u32 nseq, u8* out, u8* out4, hipStream_t st) {
    hipLaunchKernelGGL(k_pad_cls, dim3(std::max(1u, (nseq + 3) / 4)), dim3(256), 0, st, scls, off, nseq, out, out4);
}

// ---- upper bound of any local alignment score a sequence can take part in -------------------------------------
// Every column of an alignment scores at most the row maximum of its residue's class (gaps and mismatches only lower the sum),
// so  sum over the sequence of max(0, max_b b62[class][b])  bounds the score of every alignment the sequence is one side of.
// The packed 16-bit aligner (k_align16.hip) takes a task when the smaller of its two sides' bounds fits its cells: protein
// self-scores average ~5.5 per residue where the worst case (W against W) is 11, which doubles the lengths it can take.
struct RowMax {
    u8 v[32];
};
__global__ __launch_bounds__(256) void k_seq_bound(const u8* __restrict__ scls, const u32* __restrict__ off, u32 nseq, RowMax rm,
                                                   u32* __restrict__ bound) {
    const u32 s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (s >= nseq) return;
    const u32 a = off[s], e = off[s + 1];
    u32 sum = 0;
    for (u32 i = a + lane; i < e; i += 64) sum += rm.v[scls[i] & 31u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += (u32)__shfl_xor((int)sum, o);
    if (lane == 0) bound[s] = sum;
}

void launch_seq_bound(const u8* scls, const u32* off, u32 nseq, const signed char* b62c_host /*SCLS_N x SCLS_N*/, u32* bound, hipStream_t st) {
    if (!nseq) return;
    RowMax rm;
    for (int a = 0; a < 32; ++a) {
        int m = 0;
        if (a < SCLS_N)
            for (int b = 0; b < SCLS_N; ++b) m = std::max(m, (int)b62c_host[a * SCLS_N + b]);
        rm.v[a] = (u8)m;
    }
    hipLaunchKernelGGL(k_seq_bound, dim3((nseq + 3) / 4), dim3(256), 0, st, scls, off, nseq, rm, bound);
}

// ---- SEG-like query masking (fsearch.py:2872-2928; entropy 2854-2868; Counter 157-177) ------------
// The reference slides a 12-residue window and updates the entropy incrementally,  ent += t(leaving) ; ent +=
// t(entering),  so its rounding depends on the order of those additions -- but each addend depends only on the
// window: the reference's per-symbol counter always equals (occurrences in the current window) + a per-symbol
// constant (its Counter starts the first window at 2 * occ - 1 instead of occ).  One wave per query: the lanes
// compute the two addends of up to SEG_TILE positions in parallel (counts by direct comparison over the 12
// residues), then lane 0 replays the additions in order from LDS, then walks the mask into the output.
// All logarithms come from host tables of libm values (lg12[k] = log(k / 12.), lgn[w][j] = log(j / w),
// log2v = log(2)), so the device only performs IEEE fp64 multiply / subtract / divide / add / compare (the
// library is built with -ffp-contract=off) and reproduces the reference's doubles bit for bit.  symmap folds
// the upper-cased byte to one of <= 64 symbols.  Only output[:n] is produced (2996, 3034).
struct SegTab {
    double lg12[64];
    double lgn[13][32];
    double log2v;
};
#define SEG_STAGE_SMALL 1024  // instance for queries up to this length: 6 KB of LDS per block, so a CU keeps ~26 of them resident
#define SEG_STAGE_MAX 4096    // queries up to this length keep their residues and mask in LDS (CAP > 0)

// CAP > 0: the query's residues and mask live in LDS -- the sequential parts (addition replay, output walk) are chains of dependent
// loads; from LDS a step costs ~100 cycles, from global memory ~600.  CAP = 0: no staging (queries longer than SEG_STAGE_MAX).
// One wave per query, and the sequential parts run on one lane: throughput is the number of queries resident per CU, i.e. LDS per
// block -- hence the small instance (CAP 1024, 128-step tiles: 6 KB) for the typical protein next to the 4096 / 512 one (17 KB).
template <int CAP, int TILE>
__global__ __launch_bounds__(64) void k_seg(const u8* __restrict__ raw, const u32* __restrict__ src_off, u32 q_lo, const u32* __restrict__ qid, u32 nq,
                                            const u32* __restrict__ dst_off, const u8* __restrict__ symmap /*256: upper-cased byte*/,
                                            const u8* __restrict__ upmap /*256*/, const SegTab* __restrict__ tab, u8* __restrict__ mk,
                                            u8* __restrict__ out, int min_len /*this instance serves lengths > min_len*/, u32 q_first /*first slot of the grid*/) {
    constexpr bool STAGE = CAP > 0;
    constexpr int SEG_TILE = TILE;
    __shared__ u8 s_sym[256], s_up[256];
    __shared__ int s_off[64];  // reference counter - occurrences in the window, per symbol
    __shared__ double s_t1[SEG_TILE], s_t2[SEG_TILE];
    __shared__ u8 s_S[STAGE ? CAP : 4], s_m[STAGE ? CAP : 4];
    const int lane = threadIdx.x;
    const u32 q = q_first + blockIdx.x;
    if (q >= nq) return;
    const u32 sq = q_lo + (qid ? qid[q] : q);   // the batch may hold its queries in another order than the file
    const int n = (int)(src_off[sq + 1] - src_off[sq]);
    if (n <= 0) return;
    if ((STAGE && n > CAP) || n <= min_len) return;  // another instance serves this length
    for (int i = lane; i < 256; i += 64) s_sym[i] = symmap[i], s_up[i] = upmap[i];
    s_off[lane] = 0;
    __syncthreads();
    const u8* Sg = raw + src_off[sq];
    u8* o = out + dst_off[q];
    u8* mg = mk + dst_off[q];
    if (STAGE) {
        for (int i = lane; i < n; i += 64) s_S[i] = Sg[i];
        __syncthreads();
    }
    const u8* S = STAGE ? s_S : Sg;
    u8* m = STAGE ? s_m : mg;
    const double minent = 2.2, window = 12.;
    const double log2v = tab->log2v;
    const int w = n < 12 ? n : 12;
    double ent = 0;  // lane 0 only
    int prev = 0;
    if (lane == 0) {
        // first window: Counter(seq) then one more per char -> 2 * occ - 1 (157-177, 2876-2880); entropy over the
        // distinct symbols in first-seen order (2854-2868)
        int cnt[12], symv[12], nd = 0;
        for (int i = 0; i < w; ++i) {
            const int c = s_sym[S[i]];
            int k = 0;
            while (k < nd && symv[k] != c) ++k;
            if (k == nd) symv[nd] = c, cnt[nd] = 1, ++nd;
            else cnt[k] += 2;
        }
        for (int k = 0; k < nd; ++k) {
            const int j = cnt[k];
            const double freq = (double)j / ((double)w * 1.);
            ent -= freq * tab->lgn[w][j];
            s_off[symv[k]] = j - (j + 1) / 2;  // occ = (j + 1) / 2
        }
        ent /= log2v;
        prev = ent < minent ? 1 : 0;
        m[0] = (u8)prev;
    }
    __syncthreads();
    const int last = n - 12;  // sliding steps i = 1 .. last
    for (int i0 = 1; i0 <= last; i0 += SEG_TILE) {
        const int cntp = min(SEG_TILE, last - i0 + 1);
        // ---- the two addends of every step of the tile, all lanes ----
        for (int r = lane; r < cntp; r += 64) {
            const int i = i0 + r;
            const int pre = s_sym[S[i - 1]], cur = s_sym[S[i + 11]];
            double t1 = 0., t2 = 0.;  // pre == cur: the reference leaves ent alone; adding +0. twice does the same
            if (pre != cur) {
                int tp = 0, tc = 0;  // occurrences in the window before the step, S[i - 1 .. i + 10]
                for (int k = -1; k < 11; ++k) {
                    const int c = s_sym[S[i + k]];
                    tp += (c == pre), tc += (c == cur);
                }
                const int pre_count = tp + s_off[pre], cur_count = tc + s_off[cur];
                const int pre_after = pre_count - 1, cur_after = cur_count + 1;
                double a = (double)pre_count / window, b = (double)pre_after / window;
                if (pre_after != 0) {
                    t1 = (a * tab->lg12[pre_count] - b * tab->lg12[pre_after]) / log2v;
                    if (t1 == 0) t1 = a * tab->lg12[pre_count] / log2v;
                } else {
                    t1 = a * tab->lg12[pre_count] / log2v;
                }
                a = (double)cur_count / window;
                b = (double)cur_after / window;
                if (cur_count != 0) {
                    t2 = (a * tab->lg12[cur_count] - b * tab->lg12[cur_after]) / log2v;
                    if (t2 == 0) t2 = -b * tab->lg12[cur_after] / log2v;
                } else {
                    t2 = -b * tab->lg12[cur_after] / log2v;
                }
            }
            s_t1[r] = t1, s_t2[r] = t2;
        }
        __syncthreads();
        // ---- the additions, in the reference's order ----
        if (lane == 0) {
            // eight steps' addends are fetched before the dependent chain of additions consumes them (same additions, same order)
            int r = 0;
            for (; r + 8 <= cntp; r += 8) {
                double a[8], b[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = s_t1[r + k], b[k] = s_t2[r + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    ent += a[k];
                    ent += b[k];
                    m[i0 + r + k] = (u8)(ent < minent ? 1 : 0);
                }
            }
            for (; r < cntp; ++r) {
                ent += s_t1[r];
                ent += s_t2[r];
                m[i0 + r] = (u8)(ent < minent ? 1 : 0);
            }
        }
        __syncthreads();
    }
    // ---- the output walk (2918-2928): an unmasked position copies its residue and moves on by one, a masked one emits twelve 'x'
    // and moves on by twelve; output and input positions advance together, so out[p] belongs to input position p.  All lanes: 64
    // positions per step, the run of unmasked ones in front of the first masked one is copied at once.  (One lane walking a
    // 30 000-residue query through global memory took 8 ms.)
    if (!STAGE) {   // the mask was written to global memory by lane 0: make it visible to the other lanes' loads
        __threadfence();
        __syncthreads();
    }
    const int Nws = n - 12 > 0 ? n - 12 : 0;
    const int tail = m[Nws];  // if mask[Nws]: mask[Nws:] = 1 ; positions past n - 12 are otherwise 0
    int st = 0;   // wave-uniform
    while (st < n) {
        const int p = st + lane;
        int mv = 0;
        if (p < n) mv = p <= Nws ? (int)m[p] : tail;
        const unsigned long long mb = __ballot(p < n && mv != 0);
        const int run = mb ? (int)__builtin_ctzll(mb) : min(64, n - st);   // unmasked positions in front
        if (lane < run) o[p] = s_up[S[p]];
        st += run;
        if (mb) {
            if (lane < 12 && st + lane < n) o[st + lane] = 'x';
            st += 12;
        }
    }
}

// no masking (-F other than T): batch residues = raw residues
__global__ __launch_bounds__(256) void k_copy_range(const u8* __restrict__ src, u8* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

#define SEG_STAGE_GIANT 32768  // ... and up to this length in an instance of one wave per CU (72 KB of LDS): a 30 000-residue query's replay and walk
                               // out of global memory took 1.3 ms (0.3 of it the additions themselves)

// q_mid / q_long: the slots in front of them hold no query above SEG_STAGE_SMALL / SEG_STAGE_MAX residues (a class-ordered batch keeps
// its long queries at the end: the two long instances are launched over the tail only; 0: anywhere).  st_long (may equal st): the
// stream of the instance for the queries above SEG_STAGE_MAX -- one wave per query, milliseconds for a giant, beside the other two.
void launch_seg(const u8* raw, const u32* src_off, u32 q_lo, const u32* qid, u32 nq, const u32* dst_off, const u8* symmap, const u8* upmap,
                const void* tab, u8* mk, u8* out, u32 max_len, u32 q_mid, u32 q_long, hipStream_t st, hipStream_t st_long) {
    if (!nq) return;
    // instances over the same slots, each serving its length range: (0, 1024], (1024, 4096], (4096, 32768] staged in LDS, longer ones unstaged
    if (max_len > SEG_STAGE_MAX && q_long < nq) {
        hipLaunchKernelGGL((k_seg<SEG_STAGE_GIANT, 512>), dim3(nq - q_long), dim3(64), 0, st_long, raw, src_off, q_lo, qid, nq, dst_off, symmap, upmap,
                           (const SegTab*)tab, mk, out, SEG_STAGE_MAX, q_long);
        if (max_len > SEG_STAGE_GIANT)
            hipLaunchKernelGGL((k_seg<0, 512>), dim3(nq - q_long), dim3(64), 0, st_long, raw, src_off, q_lo, qid, nq, dst_off, symmap, upmap,
                               (const SegTab*)tab, mk, out, SEG_STAGE_GIANT, q_long);
    }
    hipLaunchKernelGGL((k_seg<SEG_STAGE_SMALL, 128>), dim3(nq), dim3(64), 0, st, raw, src_off, q_lo, qid, nq, dst_off, symmap, upmap, (const SegTab*)tab, mk,
                       out, 0, 0u);
    if (max_len > SEG_STAGE_SMALL && q_mid < nq)
        hipLaunchKernelGGL((k_seg<SEG_STAGE_MAX, 512>), dim3(nq - q_mid), dim3(64), 0, st, raw, src_off, q_lo, qid, nq, dst_off, symmap, upmap,
                           (const SegTab*)tab, mk, out, SEG_STAGE_SMALL, q_mid);
}

// unmasked queries of a batch that holds them in another order than the file: a wave per sequence
__global__ __launch_bounds__(256) void k_gather_seqs(const u8* __restrict__ raw, const u32* __restrict__ src_off, u32 q_lo, const u32* __restrict__ qid, u32 nq,
                                                     const u32* __restrict__ dst_off, u8* __restrict__ out) {
    const u32 q = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= nq) return;
    const u32 sq = q_lo + qid[q];
    const u32 a = src_off[sq], n = src_off[sq + 1] - a, o = dst_off[q];
    for (u32 i = lane; i < n; i += 64) out[o + i] = raw[a + i];
}
void launch_gather_seqs(const u8* raw, const u32* src_off, u32 q_lo, const u32* qid, u32 nq, const u32* dst_off, u8* out, hipStream_t st) {
    if (nq) hipLaunchKernelGGL(k_gather_seqs, dim3((nq + 3) / 4), dim3(256), 0, st, raw, src_off, q_lo, qid, nq, dst_off, out);
}

void launch_copy_range(const u8* src, u8* dst, size_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_copy_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

void launch_layout(const u8* res, const u32* off, u32 nseq, u32 P, u32 Ppad, const u8* hmap, u32* pseq, u8* pcls, u32* words,
                   hipStream_t st) {
    if (Ppad == 0) return;
    hipLaunchKernelGGL(k_layout, dim3((Ppad + 255) / 256), dim3(256), 0, st, res, off, nseq, P, Ppad, hmap, pseq, pcls);
    u32 groups = Ppad / 32;
    hipLaunchKernelGGL(k_pack5, dim3((groups + 255) / 256), dim3(256), 0, st, pcls, Ppad, words);
}

void launch_scls(const u8* res, size_t n, const u8* smap, u8* scls, u8* scls4, hipStream_t st) {
    size_t thr = std::max<size_t>((n + 3) / 4, 1);  // n == 0: the pads alone
    hipLaunchKernelGGL(k_scls, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, res, n, smap, scls, scls4);
}
