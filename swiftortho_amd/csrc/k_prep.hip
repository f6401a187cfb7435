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

// ---- SEG-like query masking (fsearch.py:2872-2928; entropy 2854-2868; Counter 157-177) ------------
// One thread per query, sequential like the reference (the entropy is updated incrementally and
// its rounding is order dependent).  All logarithms come from host tables of libm values
// (lg12[k] = log(k / 12.), lgn[w][j] = log(j / w), log2v = log(2)), so the device only performs
// IEEE fp64 multiply / subtract / divide / compare (the library is built with -ffp-contract=off)
// and reproduces the reference's doubles bit for bit.  symmap folds the upper-cased byte to one
// of <= 64 symbols; per-thread counters live in LDS.  Only output[:n] is produced (2996, 3034).
struct SegTab {
    double lg12[64];
    double lgn[13][32];
    double log2v;
};

__global__ __launch_bounds__(64) void k_seg(const u8* __restrict__ raw, const u32* __restrict__ src_off, u32 q_lo, u32 nq,
                                            const u32* __restrict__ dst_off, const u8* __restrict__ symmap /*256: upper-cased byte*/,
                                            const u8* __restrict__ upmap /*256*/, const SegTab* __restrict__ tab, u8* __restrict__ mk,
                                            u8* __restrict__ out) {
    __shared__ u8 s_cnt[64][64];  // [symbol][thread]: conflict-free per-thread counters
    __shared__ u8 s_sym[256], s_up[256];
    for (int i = threadIdx.x; i < 256; i += 64) s_sym[i] = symmap[i], s_up[i] = upmap[i];
    for (int k = 0; k < 64; ++k) s_cnt[k][threadIdx.x] = 0;
    __syncthreads();
    const u32 q = blockIdx.x * 64u + threadIdx.x;
    if (q >= nq) return;
    const u8* S = raw + src_off[q_lo + q];
    const int n = (int)(src_off[q_lo + q + 1] - src_off[q_lo + q]);
    u8* o = out + dst_off[q];
    u8* m = mk + dst_off[q];
    if (n <= 0) return;
    const double minent = 2.2, window = 12.;
    const int tx = threadIdx.x;
#define CNT(c) s_cnt[c][tx]
    const int w = n < 12 ? n : 12;
    for (int i = 0; i < w; ++i) {  // Counter(seq) + one more per char: 2 * occ - 1
        const int c = s_sym[S[i]];
        CNT(c) = CNT(c) == 0 ? 1 : CNT(c) + 2;
    }
    double ent = 0;
    for (int i = 0; i < w; ++i) {  // values() in first-seen order
        const int c = s_sym[S[i]];
        bool first = true;
        for (int k = 0; k < i; ++k) first = first && (s_sym[S[k]] != c);
        if (!first) continue;
        const int j = CNT(c);
        const double freq = (double)j / ((double)w * 1.);
        ent -= freq * tab->lgn[w][j];
    }
    ent /= tab->log2v;
    int prev = ent < minent ? 1 : 0;
    m[0] = (u8)prev;
    for (int i = 1; i < n - 12 + 1; ++i) {
        const int pre = s_sym[S[i - 1]], cur = s_sym[S[i + 11]];
        if (pre == cur) {
            m[i] = (u8)prev;
            continue;
        }
        const int pre_count = CNT(pre);
        CNT(pre) = pre_count - 1;
        const int cur_count = CNT(cur);
        CNT(cur) = cur_count + 1;
        const int pre_after = pre_count - 1, cur_after = cur_count + 1;
        double a = (double)pre_count / window, b = (double)pre_after / window, t;
        if (pre_after != 0) {
            t = (a * tab->lg12[pre_count] - b * tab->lg12[pre_after]) / tab->log2v;
            if (t == 0) t = a * tab->lg12[pre_count] / tab->log2v;
        } else {
            t = a * tab->lg12[pre_count] / tab->log2v;
        }
        ent += t;
        a = (double)cur_count / window;
        b = (double)cur_after / window;
        if (cur_count != 0) {
            t = (a * tab->lg12[cur_count] - b * tab->lg12[cur_after]) / tab->log2v;
            if (t == 0) t = -b * tab->lg12[cur_after] / tab->log2v;
        } else {
            t = -b * tab->lg12[cur_after] / tab->log2v;
        }
        ent += t;
        prev = ent < minent ? 1 : 0;
        m[i] = (u8)prev;
    }
#undef CNT
    const int Nws = n - 12 > 0 ? n - 12 : 0;
    const int tail = m[Nws];  // if mask[Nws]: mask[Nws:] = 1 ; positions past n - 12 are otherwise 0
    int st = 0, oo = 0;
    while (st < n) {
        const int mv = st <= Nws ? (int)m[st] : tail;
        if (mv == 0) {
            o[oo++] = s_up[S[st]];
            st += 1;
        } else {
            for (int k = 0; k < 12 && oo < n; ++k) o[oo++] = 'x';
            st += 12;
        }
    }
}

// no masking (-F other than T): batch residues = raw residues
__global__ __launch_bounds__(256) void k_copy_range(const u8* __restrict__ src, u8* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

void launch_seg(const u8* raw, const u32* src_off, u32 q_lo, u32 nq, const u32* dst_off, const u8* symmap, const u8* upmap,
                const void* tab, u8* mk, u8* out, hipStream_t st) {
    if (!nq) return;
    hipLaunchKernelGGL(k_seg, dim3((nq + 63) / 64), dim3(64), 0, st, raw, src_off, q_lo, nq, dst_off, symmap, upmap, (const SegTab*)tab, mk,
                       out);
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
    if (n == 0) return;
    size_t thr = (n + 3) / 4;
    hipLaunchKernelGGL(k_scls, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, res, n, smap, scls, scls4);
}
