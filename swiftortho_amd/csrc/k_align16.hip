// k_align16.hip -- score-only banded gapped extension in PACKED 16-bit arithmetic (kswat_st, fsearch.py:1357-1416).
//
// Same recurrence, same band and the same 16-lane anti-diagonal mapping as k_align.hip -- lane l of a DPP row owns row
// i = m - l of iteration m and its two band cells d = 2l, 2l + 1 -- but every 32-bit register carries TWO alignments, one
// per 16-bit half: a 16-lane row works on tasks (2g, 2g + 1) of the launch list, a wave on eight, and every DP instruction
// is a `v_pk_*_i16` (or a bitwise / byte-permute op that is half-agnostic).  The early-stop rounds of phase 2 (the bulk of
// the aligner's work on BASELINE config 3) only need each candidate's maximum, which is what this kernel returns; the rows
// that end up reported are aligned again by k_align<true> with traces.
//
// A cell's value is TAGGED:  w = (score << 2 | tag) + 44,  tag = 3 diagonal, 2 left ('-'), 1 up ('|'), 0 stop ('*').
//   * one max over the four candidates resolves the score AND the reference's trace priority diag > left > up > stop
//     (fsearch.py:1404-1411): equal scores differ in the tag;
//   * the gap cost a cell charges its right / lower neighbour depends on its own trace only (extend -1 iff it is that gap
//     direction, else open -11): the two outgoing candidates are  w + cI[tag]  and  w + cD[tag]  with 4-entry byte tables
//     held in a register and indexed by ONE v_perm for both halves;
//   * the bias 44 = 11 << 2 makes "score -11, tag 0" the integer 0, i.e. what a DPP move with bound_ctrl hands a lane without
//     a source -- exactly what an out-of-band neighbour contributes (score 0 stepping out with an opened gap) -- and it always
//     loses against the stop candidate 44.
// No validity predicates: residues outside a sequence read as a SENTINEL class whose scores are -100 against everything, so
// a cell outside the matrix never takes its diagonal candidate, cells above / left of the matrix come out as score 0 with a
// stop trace (the reference's boundary cells: 0 and a non-extending trace, 1379-1389), and whatever cells right of / below the
// matrix hold is derived from in-matrix cells minus gap costs: it never reaches back into the matrix and never exceeds the
// in-matrix maximum.  Only the first and last groups of an alignment need the sentinels (their windows are masked); interior
// groups run on raw windows.
// Range: (score << 2) + 47 must fit int16: 11 * min(rows, columns) <= 8179; launch_align_pk() is only used when the longest
// possible window allows it (host_phase2.hip), longer inputs take the 32-bit kernel.
// Per cell pair: 11 packed VALU + 1 DPP move + 2 address permutes + 1 pack of the two looked-up scores = 15, i.e. 7.5 per cell
// against 13 in k_align<false>.
#include "common.h"
#include "kernels.h"
#include <type_traits>

#define KB 16
#define PK_SENT_ROW 24          // sentinel row class
#define PK_SENT_COL4 96         // sentinel column class * 4
// Score table: one 16-bit entry per (row class, column class, lane mod 32) at byte (row << 11) | (col << 6) | (lane32 << 1).
// With one copy of the table a lookup's bank is a function of the column class alone -- 24 classes of very unequal frequency
// spread 32 lanes over a dozen banks, ~5 deep: measured on config 3, the round-2 aligner spent ~20 of its 25 ms per step in LDS
// bank conflicts, which is why halving its VALU work alone changed nothing.  Here every lane reads its own copy: lanes 2k and
// 2k + 1 share a bank (two 16-bit entries per dword), nothing else collides -- at most 2-way.  Address = one v_perm
// (row << 8 | col * 8) and one v_lshl_or (<< 3 | lane32 * 2).
#define PK_TAB (25 * 2048)
#define PK_STOP 0x002C002Cu     // score 0, tag 0 (+ 44) in both halves
#define PK_TAG3 0x00030003u
#define PK_CI 0xD3FCD5D6u       // low bytes of (candidate for the right neighbour) - w by tag: -42, -43, -4, -45
#define PK_CD 0xD2D3FCD5u       // low bytes of (candidate for the lower neighbour) - w by tag: -43, -4, -45, -46

typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) {
    const pk16 r = __builtin_bit_cast(pk16, a) + __builtin_bit_cast(pk16, b);
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) {
    const pk16 r = __builtin_elementwise_max(__builtin_bit_cast(pk16, a), __builtin_bit_cast(pk16, b));
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 pk_dpp_shr1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ u32 pk_dpp_shl1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true); }

__device__ __forceinline__ u32 pk_load4u(const u8* p) {
    u32 w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
__device__ __forceinline__ u64 pk_load8u(const u8* p) {
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

// bytes [lo, hi) of a 64-bit word set (0 <= lo, hi <= 8; empty when hi <= lo)
__device__ __forceinline__ u64 pk_bytemask(int lo, int hi) {
    const u64 a = hi >= 8 ? ~0ull : ((1ull << (8 * hi)) - 1ull);
    const u64 b = lo >= 8 ? ~0ull : ((1ull << (8 * lo)) - 1ull);
    return a & ~b;
}

// One cell pair.  I / D: the candidates the left / upper neighbours published; Wd3: the diagonal neighbour's value with its tag
// bits forced to 3; S: 4 x substitution score per half.  Publishes w (value), the two outgoing candidates and w | 3.
__device__ __forceinline__ void pk_cell(u32 I, u32 D, u32 Wd3, u32 S, u32& w, u32& Iout, u32& Dout, u32& W3) {
    const u32 M = pk_add(Wd3, S);
    w = pk_max(pk_max(I, D), pk_max(M, PK_STOP));
    u32 sel;   // per half: byte 0 = the tag (picks a byte of the cost table), byte 1 = 0x0d (constant 0xff: the costs are negative)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(w), "s"(PK_TAG3), "v"(0x0D000D00u));   // (one op; the compiler emits and + or with literals)
    Iout = pk_add(w, __builtin_amdgcn_perm(0u, PK_CI, sel));
    Dout = pk_add(w, __builtin_amdgcn_perm(0u, PK_CD, sel));
    W3 = w | PK_TAG3;
}

struct PkSide {                 // one of the two alignments of a 16-lane row
    const u8* rp;               // row classes of the current group's four iterations (this lane)
    const u8* cp;               // column classes * 4, one column early
    int nrows, ncols, R;
    int int_hi;                 // last group start m0 that needs no window masks
    int ridx0, cidx0;           // index (into the sequence) of byte 0 of the two windows at m0 = 8
    u32 slot;
    int ncell;
};

// the class arrays are the PADDED ones (k_pad_cls): sequence s at off[s] + PCLS_PAD * s, PCLS_PAD sentinel bytes behind it
__device__ __forceinline__ void pk_setup(PkSide& s, const AlnTask& tk, u32 slot, int l, const u8* __restrict__ q_scls,
                                         const u8* __restrict__ q_scls4, const u32* __restrict__ qoff, const u8* __restrict__ r_scls,
                                         const u8* __restrict__ r_scls4, const u32* __restrict__ roff) {
    const u32 qb = qoff[tk.q], sb = roff[tk.subj];
    const int lenq = (int)(qoff[tk.q + 1] - qb), lens = (int)(roff[tk.subj + 1] - sb);
    const int lq = min(lenq, (int)tk.qe), ls = min(lens, (int)tk.se);
    q_scls += (size_t)PCLS_PAD * tk.q, q_scls4 += (size_t)PCLS_PAD * tk.q, r_scls += (size_t)PCLS_PAD * tk.subj, r_scls4 += (size_t)PCLS_PAD * tk.subj;
    const int qi = min((int)tk.qi, lq), qj = min((int)tk.qj, ls);
    const int la = lq - qi, lb = ls - qj;
    const bool swp = !(la < lb);  // abs(qed - qst) < abs(sed - sst) -> no swap (1364-1369)
    s.ncols = swp ? lb : la, s.nrows = swp ? la : lb;
    const u8* ccls = swp ? (r_scls4 + sb + qj) : (q_scls4 + qb + qi);
    const u8* rcls = swp ? (q_scls + qb + qi) : (r_scls + sb + qj);
    s.R = min(s.nrows, s.ncols + KB);
    // Both windows end where their sequences end (no tile of a long sequence): what lies behind them are the arrays' sentinels, PCLS_PAD
    // bytes of them -- rows are read up to R + 18, columns up to ncols + 37 -- so the groups up to the alignment's last need no masks.
    // Else (and for every group in front of m0 = 17): the masked path.
    s.int_hi = ((int)tk.qe >= lenq && (int)tk.se >= lens) ? s.R + 15 : min(s.R, s.ncols) - 3;
    s.ridx0 = 8 - l - 1, s.cidx0 = 8 + l - KB - 1;
    s.rp = rcls + s.ridx0, s.cp = ccls + s.cidx0;
    s.slot = slot;
    // cells the reference evaluates (a counter the oracle keeps too): the lane's two band offsets' row ranges
    const int lo_e = max(1, 17 - 2 * l), lo_o = max(1, 16 - 2 * l);
    s.ncell = max(0, min(s.R, s.ncols + 16 - 2 * l) - lo_e + 1) + max(0, min(s.R, s.ncols + 15 - 2 * l) - lo_o + 1);
}

#define PK_THREADS 1024   // two workgroups (2 x 51 KB of table) per CU = 8 waves per SIMD
// TRACE = true (round 5): the same cells with their traces written and the position of the maximum kept, for the alignments that are
// reported -- k_align<true>'s job at 10.5 instead of 15 vector instructions per cell.  A cell's trace IS its tag: the four iterations of a
// group shift their (odd tag << 2 | even tag) nibbles into a register, both alignments at once, and every second group ends with one
// store per alignment into the 32-bit kernel's own trace layout (word TRACE_WORD((m >> 3) - 1, lane), iteration m in nibble 7 - (m & 7): a group is
// the upper or the lower half of such a word).  The codes differ -- tag 3 diagonal, 1 up against trace 1 diagonal, 3 up -- so the
// result says which it is (AlnRes.pad = 1) and k_traceback swaps the two when it reads them.  The maximum's position: four 32-bit keys
// (alignment A / B x even / odd cell) of (value | 3) << 16 | (0xFFFF - iteration), two instructions per cell and alignment.
template <bool TRACE>
__global__ __launch_bounds__(PK_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_align_pk(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 ntasks,
                                                  const u8* __restrict__ q_scls, const u8* __restrict__ q_scls4, const u32* __restrict__ qoff,
                                                  const u8* __restrict__ r_scls, const u8* __restrict__ r_scls4, const u32* __restrict__ roff,
                                                  const signed char* __restrict__ b62g, AlnRes* __restrict__ out, u32* __restrict__ trace,
                                                  u32 trace_stride, const u32* __restrict__ tofs, u32* __restrict__ tpos_out, u32 tpos_base,
                                                  u32 t0 /*first list position of this launch: ridx, tofs and the traces are the whole list's*/) {
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[PK_TAB];
    for (int i = threadIdx.x; i < 25 * 32 * 32; i += PK_THREADS) {
        const int a = i >> 10, b = (i >> 5) & 31;   // row class, column class; i & 31 = the lane's copy
        const int v = (a < SCLS_N && b < SCLS_N) ? 4 * (int)b62g[a * SCLS_N + b] : -400;
        reinterpret_cast<short*>(s_tab)[i] = (short)v;
    }
    __syncthreads();
    const u32 lane2 = (threadIdx.x & 31u) << 1;
    const u32 g = blockIdx.x * (PK_THREADS / 16) + (threadIdx.x >> 4);   // 16-lane row = task pair
    const int l = threadIdx.x & 15;
    if (t0 + 2u * g >= ntasks) return;
    const u32 tA = t0 + 2u * g, tB = min(tA + 1u, ntasks - 1u);  // odd tail: the pair is (last, last), written once
    PkSide A, B;
    {
        const u32 sa = ridx ? ridx[tA] : tA, sb = ridx ? ridx[tB] : tB;
        pk_setup(A, tasks[sa], sa, l, q_scls, q_scls4, qoff, r_scls, r_scls4, roff);
        pk_setup(B, tasks[sb], sb, l, q_scls, q_scls4, qoff, r_scls, r_scls4, roff);
    }
    u32 W3e = PK_STOP | PK_TAG3, W3o = PK_STOP | PK_TAG3, Io_out = 0, Do_out = 0;  // results of iteration m - 1
    u32 key = PK_STOP;
    // TRACE: (value | 3) << 16 | (0xFFFF - m) of the best even / odd cell of A and of B (first strict maximum in row-major order = the largest
    // such key inside a lane: i = m - l); the group's trace nibbles; where the two alignments' traces live
    u32 keyEA = 47u << 16, keyOA = keyEA, keyEB = keyEA, keyOB = keyEA, tw = 0, twh = 0;
    u32 tuA = 0, tuB = 0;
    u32 *trA = nullptr, *trB = nullptr;
    if (TRACE) {
        tuA = tofs ? tofs[tA] : tA, tuB = tofs ? tofs[tB] : tB;
        trA = trace + (size_t)tuA * trace_stride + TRACE_WORD(0, l);
        trB = trace + (size_t)tuB * trace_stride + TRACE_WORD(0, l);
    }
    const int mendA = A.R + 15, mendB = tB != tA ? B.R + 15 : -1;
    const int m_end = max(A.R, B.R) + 15;
    const int int_hi = min(A.int_hi, B.int_hi);  // groups m0 in [17, int_hi]: every window byte of both sides is inside its sequence or a sentinel behind it

    auto four = [&](int m0, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
        u32 rwA = pk_load4u(A.rp), rwB = pk_load4u(B.rp);
        u64 cwA = pk_load8u(A.cp), cwB = pk_load8u(B.cp);
        if (EDGE) {   // bytes outside [0, rows) / [0, columns) -> sentinel classes
            const int ra = A.ridx0 + (m0 - 8), rb = B.ridx0 + (m0 - 8), ca = A.cidx0 + (m0 - 8), cb = B.cidx0 + (m0 - 8);
            const u32 mra = (u32)pk_bytemask(max(-ra, 0), min(max(A.nrows - ra, 0), 4)), mrb = (u32)pk_bytemask(max(-rb, 0), min(max(B.nrows - rb, 0), 4));
            const u64 mca = pk_bytemask(max(-ca, 0), min(max(A.ncols - ca, 0), 8)), mcb = pk_bytemask(max(-cb, 0), min(max(B.ncols - cb, 0), 8));
            rwA = (rwA & mra) | (0x18181818u & ~mra), rwB = (rwB & mrb) | (0x18181818u & ~mrb);
            cwA = (cwA & mca) | (0x6060606060606060ull & ~mca), cwB = (cwB & mcb) | (0x6060606060606060ull & ~mcb);   // (sentinel column 24, * 4)
        }
        // column classes arrive * 4; the table wants * 8 (no carry between bytes: every class * 4 is < 128)
        cwA <<= 1, cwB <<= 1;
        // all sixteen score lookups of the group first (two cells x four iterations x two alignments)
        u32 S0[4], S1[4];
        u32 ad[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (row class << 8) | column class * 8: bytes 1 and 0 picked by one v_perm (selector bytes 0-3: second source)
            const u32 selE = 0x0C0C0000u | ((u32)(4 + k) << 8) | (u32)k;
            const u32 selO = 0x0C0C0000u | ((u32)(4 + k) << 8) | (u32)(k + 1);
            const u32 a0A = __builtin_amdgcn_perm(rwA, (u32)cwA, selE), a0B = __builtin_amdgcn_perm(rwB, (u32)cwB, selE);
            const u32 a1A = k < 3 ? __builtin_amdgcn_perm(rwA, (u32)cwA, selO) : __builtin_amdgcn_perm(rwA, (u32)(cwA >> 32), 0x0C0C0700u);
            const u32 a1B = k < 3 ? __builtin_amdgcn_perm(rwB, (u32)cwB, selO) : __builtin_amdgcn_perm(rwB, (u32)(cwB >> 32), 0x0C0C0700u);
            const u32 lds0 = (u32)(uintptr_t)s_tab;
            ad[4 * k] = lds0 + ((a0A << 3) | lane2), ad[4 * k + 1] = lds0 + ((a0B << 3) | lane2);
            ad[4 * k + 2] = lds0 + ((a1A << 3) | lane2), ad[4 * k + 3] = lds0 + ((a1B << 3) | lane2);
        }
        // alignment A's entry is read zero-extended, alignment B's straight into the HIGH half (ds_read_u16_d16_hi; with SRAM ECC on, as
        // on gfx950, a d16 load clears the other half instead of keeping it): the pair is joined by a plain OR (full-rate) instead of
        // the byte permute the compiler emits for (b << 16) | a.
        // (one statement per load, so that a result may reuse its own address register; the wait statement below ties all sixteen
        // results to the counter the compiler cannot see)
        u32 lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("ds_read_u16 %0, %1" : "=v"(lo[k]) : "v"(ad[2 * k]));
            asm volatile("ds_read_u16_d16_hi %0, %1" : "=v"(hi[k]) : "v"(ad[2 * k + 1]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]), "+v"(hi[0]), "+v"(hi[1]),
                       "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
#pragma unroll
        for (int k = 0; k < 4; ++k) S0[k] = lo[2 * k] | hi[2 * k], S1[k] = lo[2 * k + 1] | hi[2 * k + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 we, Ie_out, De_out, wo;
            pk_cell(pk_dpp_shr1(Io_out), Do_out, W3e, S0[k], we, Ie_out, De_out, W3e);
            pk_cell(Ie_out, pk_dpp_shl1(De_out), W3o, S1[k], wo, Io_out, Do_out, W3o);
            if (TRACE) {
                const u32 rk = 0xFFFFu - (u32)(m0 + k);   // (wave-uniform)
                keyEA = max(keyEA, __builtin_amdgcn_alignbit(W3e, rk << 16, 16)), keyOA = max(keyOA, __builtin_amdgcn_alignbit(W3o, rk << 16, 16));
                keyEB = max(keyEB, (W3e & 0xFFFF0000u) | rk), keyOB = max(keyOB, (W3o & 0xFFFF0000u) | rk);
                // (plain 32-bit shifts: a half holds exactly the four nibbles of a group and is cleared behind its store; the bits the
                // shift by 2 carries across the halves fall outside the mask)
                tw = (((wo << 2) & 0x000C000Cu) | (we & 0x00030003u)) | (tw << 4);
            } else {
                key = pk_max(key, pk_max(we, wo));
            }
        }
        if (TRACE) {
            // the group is the upper (m0 & 4 == 0) or the lower half of word [(m0 >> 3) - 1][l]: whole words are stored, by the lower half's
            // group (an alignment that ends inside the upper half's group still gets its word: the lower half then holds cells nobody
            // reads; the longer partner's last upper half is stored behind the loop)
            if (!(m0 & 4)) {
                twh = tw;
            } else {
                const u32 at = TRACE_WORD((m0 >> 3) - 1, 0);
                if (m0 - 4 <= mendA) trA[at] = (twh << 16) | (tw & 0xFFFFu);
                if (m0 - 4 <= mendB) trB[at] = (twh & 0xFFFF0000u) | (tw >> 16);
            }
            tw = 0;
        }
        A.rp += 4, A.cp += 4, B.rp += 4, B.cp += 4;
    };
    // every ACTIVE lane's group m0 is interior: one compare against the exec mask (m0 is wave-uniform)
    auto all_interior = [&](int m0) {
        return m0 >= 17 && __builtin_amdgcn_sicmp(m0, int_hi, 41 /*ICMP_SLE*/) == __builtin_amdgcn_read_exec();
    };
    for (int m0 = 8; m0 <= m_end; m0 += 4) {
        if (all_interior(m0)) {
            do {
                four(m0, std::false_type{});
                m0 += 4;
            } while (all_interior(m0));
        }
        four(m0, std::true_type{});
    }
    if (TRACE) {   // a last group that is an upper half: left-aligned, as k_align<true> stores it
        const int ml = 8 + ((m_end - 8) & ~3);
        if (!(ml & 4)) {
            const u32 at = TRACE_WORD((ml >> 3) - 1, 0);
            if (ml <= mendA) trA[at] = twh << 16;
            if (ml <= mendB) trB[at] = twh & 0xFFFF0000u;
        }
    }
    // reduce over the 16 lanes: maximum per half; cells per side
    int ncA = A.ncell, ncB = B.ncell;
    if (TRACE) {
        // lane best: max score, then smallest i, then the even cell (smaller j); over the lanes: max score, smallest i, smallest j
        auto lane_best = [&](u32 kE, u32 kO, int& best, int& bi, int& bj) {
            const int sE = ((int)(kE >> 16) - 47) >> 2, sO = ((int)(kO >> 16) - 47) >> 2;
            const int iE = 0xFFFF - (int)(kE & 0xFFFFu) - l, iO = 0xFFFF - (int)(kO & 0xFFFFu) - l;
            const bool takeO = (sO > sE) || (sO == sE && iO < iE);
            best = takeO ? sO : sE, bi = takeO ? iO : iE;
            bj = bi + 2 * l - KB + (takeO ? 1 : 0);
            if (best == 0) bi = 0, bj = 0;  // nothing scored: (i_max, j_max) stay (0, 0) (1391)
        };
        int bA, iA, jA, bB, iB, jB;
        lane_best(keyEA, keyOA, bA, iA, jA);
        lane_best(keyEB, keyOB, bB, iB, jB);
        for (int msk = 8; msk > 0; msk >>= 1) {
            const int ob = __shfl_xor(bA, msk, 16), oi = __shfl_xor(iA, msk, 16), oj = __shfl_xor(jA, msk, 16);
            if (ob > bA || (ob == bA && (oi < iA || (oi == iA && oj < jA)))) bA = ob, iA = oi, jA = oj;
            const int pb = __shfl_xor(bB, msk, 16), pi = __shfl_xor(iB, msk, 16), pj = __shfl_xor(jB, msk, 16);
            if (pb > bB || (pb == bB && (pi < iB || (pi == iB && pj < jB)))) bB = pb, iB = pi, jB = pj;
            ncA += __shfl_xor(ncA, msk, 16);
            ncB += __shfl_xor(ncB, msk, 16);
        }
        if (l != 0) return;
        AlnRes r;
        r.aln = 0, r.matches = 0, r.gap = 0, r.pad = 1 /*traces = tags*/, r.sst = 0, r.sed = 0;
        r.maxscore = bA, r.qst = iA, r.qed = jA, r.cells = ncA;   // (i_max, j_max) parked for k_traceback
        out[A.slot] = r;
        if (tpos_out) tpos_out[A.slot] = tpos_base + tuA;
        if (tB != tA) {
            r.maxscore = bB, r.qst = iB, r.qed = jB, r.cells = ncB;
            out[B.slot] = r;
            if (tpos_out) tpos_out[B.slot] = tpos_base + tuB;
        }
        return;
    }
    for (int msk = 8; msk > 0; msk >>= 1) {
        key = pk_max(key, (u32)__shfl_xor((int)key, msk, 16));
        ncA += __shfl_xor(ncA, msk, 16);
        ncB += __shfl_xor(ncB, msk, 16);
    }
    if (l != 0) return;
    AlnRes r;
    r.aln = 0, r.matches = 0, r.gap = 0, r.pad = 0, r.qst = 0, r.qed = 0, r.sst = 0, r.sed = 0;
    r.maxscore = ((int)(short)(key & 0xFFFFu) - 44) >> 2, r.cells = ncA;
    out[A.slot] = r;
    if (tB != tA) {
        r.maxscore = ((int)(short)(key >> 16) - 44) >> 2, r.cells = ncB;
        out[B.slot] = r;
    }
}

// k_align_pk joins its two alignments' table entries with ds_read_u16_d16_hi + OR, which needs the d16 load to CLEAR the half it does
// not write.  That is what a part with SRAM ECC does (every MI300 / MI355X; the compiler never emits d16_hi for such a target because
// of it); a part that PRESERVES the other half would leave the register's old bits under alignment A's entry.  Asked once per process:
// one lane loads into a register holding all ones.  false -> the host sends every score-only alignment to k_align<false>.
__global__ void k_d16_probe(u32* __restrict__ out) {
    __shared__ u32 s_w[2];
    s_w[0] = 0x12345678u, s_w[1] = 0u;
    __syncthreads();
    u32 v = 0xFFFFFFFFu;
    asm volatile("ds_read_u16_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"((u32)(uintptr_t)s_w));
    out[0] = v;
}
bool align_pk_supported(hipStream_t st) {
    static int state = -1;   // (one device kind per process)
    if (state < 0) {
        u32* d = nullptr;
        u32 h = 0;
        if (hipMalloc((void**)&d, sizeof(u32)) != hipSuccess) return false;
        hipLaunchKernelGGL(k_d16_probe, dim3(1), dim3(1), 0, st, d);
        const bool ok = hipMemcpyAsync(&h, d, sizeof(u32), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        (void)hipFree(d);
        state = (ok && h == 0x56780000u) ? 1 : 0;
    }
    return state == 1;
}

// largest min(rows, columns) whose scores fit the packed cells: (11 * n << 2) + 47 + 44 (one more substitution) <= 32767
int align_pk_max_len() { return 740; }
// ... or whose score bound does: (score << 2) + 47 + 44 <= 32767
u32 align_pk_max_score() { return 8169; }

void launch_align_pk(const AlnTask* tasks, const u32* ridx, u32 ntasks, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g, AlnRes* out,
                     hipStream_t st) {
    if (!ntasks) return;
    const u32 pairs = (ntasks + 1) / 2;
    hipLaunchKernelGGL((k_align_pk<false>), dim3((pairs + PK_THREADS / 16 - 1) / (PK_THREADS / 16)), dim3(PK_THREADS), 0, st, tasks, ridx, ntasks, pk.q, pk.q4, qoff,
                       pk.r, pk.r4, roff, b62g, out, (u32*)nullptr, 0u, (const u32*)nullptr, (u32*)nullptr, 0u, 0u);
}

// with traces (the walk is k_traceback's), list positions [t0, t1) of a launch list: ridx, tofs (or the position itself) and the trace room
// are the whole list's, as k_align<true> and k_traceback see them; tofs / tpos_out / tpos_base as for k_align<true>
void launch_align_pk_traced(const AlnTask* tasks, const u32* ridx, u32 t0, u32 t1, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g,
                            u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out, u32* tpos_out, u32 tpos_base, hipStream_t st) {
    if (t1 <= t0) return;
    const u32 pairs = (t1 - t0 + 1) / 2;
    hipLaunchKernelGGL((k_align_pk<true>), dim3((pairs + PK_THREADS / 16 - 1) / (PK_THREADS / 16)), dim3(PK_THREADS), 0, st, tasks, ridx, t1, pk.q, pk.q4, qoff,
                       pk.r, pk.r4, roff, b62g, out, trace, trace_stride, tofs, tpos_out, tpos_base, t0);
}
