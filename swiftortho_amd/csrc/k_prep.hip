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

// raw byte -> BLOSUM62 score class
__global__ __launch_bounds__(256) void k_scls(const u8* __restrict__ res, size_t n, const u8* __restrict__ smap, u8* __restrict__ scls) {
    __shared__ u8 s_map[256];
    s_map[threadIdx.x] = smap[threadIdx.x];
    __syncthreads();
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        uchar4 v = *reinterpret_cast<const uchar4*>(res + i);
        uchar4 o = make_uchar4(s_map[v.x], s_map[v.y], s_map[v.z], s_map[v.w]);
        *reinterpret_cast<uchar4*>(scls + i) = o;
    } else {
        for (int k = 0; k < 4; ++k)
            if (i + k < n) scls[i + k] = s_map[res[i + k]];
    }
}

void launch_layout(const u8* res, const u32* off, u32 nseq, u32 P, u32 Ppad, const u8* hmap, u32* pseq, u8* pcls, u32* words,
                   hipStream_t st) {
    if (Ppad == 0) return;
    hipLaunchKernelGGL(k_layout, dim3((Ppad + 255) / 256), dim3(256), 0, st, res, off, nseq, P, Ppad, hmap, pseq, pcls);
    u32 groups = Ppad / 32;
    hipLaunchKernelGGL(k_pack5, dim3((groups + 255) / 256), dim3(256), 0, st, pcls, Ppad, words);
}

void launch_scls(const u8* res, size_t n, const u8* smap, u8* scls, hipStream_t st) {
    if (n == 0) return;
    size_t thr = (n + 3) / 4;
    hipLaunchKernelGGL(k_scls, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, res, n, smap, scls);
}
