// k_align.hip -- banded gapped extension with traceback (kswat_st, fsearch.py:1357-1476).
//
// Reference semantics kept exactly: columns = side with the shorter remainder
// (ties -> subject on columns); band j in [i-16, i+15]; single-matrix affine approximation where a
// gap is "extended" (-1) only if the neighbour's trace is the same gap direction, else opened (-11);
// trace priority diag > left > up > stop; every out-of-band / boundary neighbour reads as score 0 with
// a non-extending trace; first strict maximum in row-major order wins; traceback walks through
// row 0 ('-'), column 0 ('|') and the left boundary cell (i, i-17) ('|') until a stop cell.
//
// CDNA4 mapping: one alignment per 16 lanes (one DPP row), 4 alignments per wave64.  Band cell
// d = j - i + 16 in [0, 32); anti-diagonal step t = 2i + d.  At step t lane l owns d = 2l + (t & 1):
//   left (i, d-1) and up (i-1, d+1) were produced at step t-1 (own lane or the DPP-row neighbour),
//   diag (i-1, d) at step t-2 by the same lane.
// One loop iteration = one row per lane (its even and odd cell), one DPP move per cell, residues
// streamed through 8-byte register windows, no LDS traffic for the DP state; LDS holds only the
// score table.  2-bit traces go to a per-task scratch slab; the traceback caches its trace word.
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <type_traits>

#define KB 16  // kbound

// DP values are kept BIASED by +11 (= -GO): a score b is held as b + 11 and a gap input (score + gap cost) as that + 11, so the
// value every out-of-band / boundary neighbour contributes -- score 0 stepping out with an opened gap, 0 - 11 -- is the integer 0,
// which is what a DPP move with bound_ctrl writes into the lane that has no source: no `old` operand to initialise per cell.
// Equalities between the candidates of a cell (the trace priority) and the order of scores are unchanged by the bias.
#define BIAS 11
// neighbour lane's value inside the 16-lane DPP row; lanes without a source get 0 (the biased -11)
__device__ __forceinline__ int dpp_row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_row_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true); }

__device__ __forceinline__ u32 load4u(const u8* p) {  // unaligned 4-byte global load
    u32 w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
__device__ __forceinline__ u64 load8u(const u8* p) {  // unaligned 8-byte global load
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

// class bytes [idx, idx + 4) of a sequence; idx runs from -9 (band cells left of / above the matrix) to len + 37 (below /
// right of it).  Bytes outside [0, len) are don't-care -- they only ever feed invalid cells, and every
// byte of a class array, pads included, is a class < SCLS_N (or 4 x one) and indexes inside the score table -- so the load is
// unconditional: the class arrays carry 16 readable bytes in front and 64 behind, and inside the array the neighbours are other
// sequences' residues.  (The bounds-checked version of this helper was 126 of the 274 instructions of a four-row step.)
__device__ __forceinline__ u32 win4(const u8* __restrict__ base, int idx) { return load4u(base + idx); }

#define GO (-11)
#define GE (-1)

// ---- lane predicates as 64-bit wave masks ------------------------------------------------------------
// Every predicate of a band cell is an integer compare, so it is kept as the SGPR-pair mask the compare
// produces; the boolean algebra (validity, trace priority) then runs on the scalar unit, and the few
// per-lane selections read the mask directly (v_cndmask / v_addc with an SGPR-pair operand).  The
// compiler cannot be made to keep an `i1` in this form across `&&` / `||`, hence the three helpers.
typedef unsigned long long m64;
#define ICMP_EQ 32
#define ICMP_ULT 36
__device__ __forceinline__ int sel_bias(int x, m64 m) {  // m ? x : BIAS (the biased score 0)
    int r;
    asm("v_cndmask_b32_e64 %0, 11, %1, %2" : "=v"(r) : "v"(x), "s"(m));
    return r;
}
__device__ __forceinline__ int gapcost(m64 ext) {  // ext ? GE (-1) : GO (-11)
    int r;
    asm("v_cndmask_b32_e64 %0, -11, -1, %1" : "=v"(r) : "s"(ext));
    return r;
}
__device__ __forceinline__ u32 shl1_in(u32 x, m64 bit) {  // 2 * x + bit
    u32 r;
    m64 cout;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(cout) : "v"(x), "s"(bit));
    return r;
}

// One band cell, all values biased (see BIAS).  I / D arrive ready-made from the producing cells (their score plus the gap cost
// that applies when stepping out of them: extend iff their own trace is that gap direction, else
// open); Bd = diagonal neighbour's score.  Publishes B, Iout (for the cell to the right),
// Dout (for the cell below) and the two bits of the trace code  0 '*', 1 '\\', 2 '-', 3 '|'
// (priority diag > left > up, fsearch.py:1404-1411) as masks t0 (bit 0) and t1 (bit 1).
// EDGE = false: the caller knows the cell is valid in every active lane (band interior), `valid` is ignored.
template <bool EDGE>
__device__ __forceinline__ void dp_cell(m64 valid, int I, int D, int Bd, int s, int& B, int& Iout, int& Dout, m64& t0, m64& t1) {
    const int M = Bd + s;
    const int b0 = max(max(I, D), max(M, BIAS));
    m64 isM = __builtin_amdgcn_sicmp(b0, M, ICMP_EQ), eI = __builtin_amdgcn_sicmp(b0, I, ICMP_EQ), eD = __builtin_amdgcn_sicmp(b0, D, ICMP_EQ);
    if (EDGE) isM &= valid, eI &= valid, eD &= valid;
    eI &= ~isM;
    eD &= ~(isM | eI);
    const int b = EDGE ? sel_bias(b0, valid) : b0;
    B = b;
    t0 = isM | eD;
    t1 = eI | eD;
    Iout = b + gapcost(eI);
    Dout = b + gapcost(eD);
}

// LDS score table addressed by ONE v_perm per cell: byte offset (row class << 8) | (column class * 4), the column classes read
// from the pre-scaled copy of the class array (k_scls).  Row stride 256 B; the * 4 spreads the 24 column classes over 24 banks.
#define AL_TAB (SCLS_N * 256)

// trace codes: 0 '*' (stop), 1 '\\' (diag), 2 '-' (left), 3 '|' (up)
// TRACE = false: score-only run (maximum and its cell): the early-stop rule (k_stop_round_w) needs nothing else, and only the
// few alignments that end up reported are run again with TRACE = true for the traceback.
template <bool TRACE>
__global__ __launch_bounds__(256) void k_align(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 ntasks,
                                               const u8* __restrict__ q_scls, const u8* __restrict__ q_scls4, const u32* __restrict__ qoff,
                                               const u8* __restrict__ r_scls, const u8* __restrict__ r_scls4, const u32* __restrict__ roff,
                                               const signed char* __restrict__ b62g, u32* __restrict__ trace, u32 trace_stride,
                                               const u32* __restrict__ tofs /*TRACE: where launch position tid's trace starts, in units of trace_stride words
                                                                              (k_trace_units + scan: every task takes what ITS band needs); null: tid*/,
                                               AlnRes* __restrict__ out, u32* __restrict__ tpos_out, u32 tpos_base) {
    __shared__ signed char s_b62[AL_TAB];
    for (int i = threadIdx.x; i < SCLS_N * 64; i += 256) {
        const int a = i >> 6, b = i & 63;
        s_b62[a * 256 + b * 4] = b < SCLS_N ? b62g[a * SCLS_N + b] : (signed char)-4;
    }
    __syncthreads();
    const u32 tid = blockIdx.x * 16u + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (tid >= ntasks) return;
    const u32 slot = ridx ? ridx[tid] : tid;  // task / result slot; the trace slab is per launch position
    const AlnTask tk = tasks[slot];
    const u32 qb = qoff[tk.q], sb = roff[tk.subj];
    const int lq = min((int)(qoff[tk.q + 1] - qb), (int)tk.qe), ls = min((int)(roff[tk.subj + 1] - sb), (int)tk.se);
    const int qi = min((int)tk.qi, lq), qj = min((int)tk.qj, ls);
    const int la = lq - qi, lb = ls - qj;
    const bool swp = !(la < lb);  // abs(qed - qst) < abs(sed - sst) -> no swap (1364-1369)
    const int ncols = swp ? lb : la, nrows = swp ? la : lb;
    const u8* ccls = swp ? (r_scls4 + sb + qj) : (q_scls4 + qb + qi);  // column classes * 4
    const u8* rcls = swp ? (q_scls + qb + qi) : (r_scls + sb + qj);
    const int R = min(nrows, ncols + KB);  // rows beyond ncols + 16 have an empty band
    const u32 tunit = (TRACE && tofs) ? tofs[tid] : tid;
    u32* tr = trace + (size_t)tunit * trace_stride;

    // Iteration m: every lane handles ONE row i = m - l and its two band cells
    //   even cell: d = 2l   (column j0 = m + l - 16)        odd cell: d = 2l + 1   (column j0 + 1)
    // even: left = lane l-1's odd cell of iteration m-1, up = own odd cell of m-1, diag = own even cell of m-1
    // odd : left = own even cell of m,               up = lane l+1's even cell of m, diag = own odd cell of m-1
    // Per iteration a lane consumes one new row class (row i) and one new column class (column j0 + 1;
    // column j0's class is last iteration's).  Both stream through register windows loaded once per
    // four-iteration group (4 row bytes, 8 column bytes starting one column early).
    // A cell (i, d) exists for i in [max(1, 17 - d), min(R, ncols + 16 - d)]: in iteration terms m - l in that range, one
    // unsigned compare per cell against per-lane constants (first iteration, count).
    const int lo_e = max(1, 17 - 2 * l), lo_o = max(1, 16 - 2 * l);
    const int cnt_e = max(0, min(R, ncols + 16 - 2 * l) - lo_e + 1), cnt_o = max(0, min(R, ncols + 15 - 2 * l) - lo_o + 1);
    const int m_e = l + lo_e, m_o = l + lo_o;  // first iteration with a valid even / odd cell
    int Be = BIAS, Bo = BIAS, Io_out = 0, Do_out = 0;  // results of iteration m-1 (biased)
    u32 keyE = TRACE ? ((u32)BIAS << 13) : (u32)BIAS, keyO = keyE;  // best even / odd cell: score << 13 | (8191 - i)
    u32 tw = 0;
    const int m_end = R + 15;
    // four iterations; EDGE = false when every cell of every active lane is inside its band and matrix.
    // rw = the row classes of the four iterations; cw0 / cw1 = column classes * 4 starting ONE column early (five are used:
    // iteration k's even cell sits in column byte k, its odd cell in byte k + 1)
    auto four = [&](int m0, u32 rw, u32 cw0, u32 cw1, auto edge) {
        constexpr bool EDGE = decltype(edge)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + k;
            m64 ve = ~0ull, vo = ~0ull;
            if (EDGE) {
                ve = __builtin_amdgcn_uicmp((u32)(m - m_e), (u32)cnt_e, ICMP_ULT);
                vo = __builtin_amdgcn_uicmp((u32)(m - m_o), (u32)cnt_o, ICMP_ULT);
            }
            // (row class << 8) | column class * 4: bytes 1 and 0 picked by one v_perm (selector bytes 0-3: second source)
            const u32 a0 = __builtin_amdgcn_perm(rw, cw0, 0x0C0C0000u | ((u32)(4 + k) << 8) | (u32)k);
            const u32 a1 = k < 3 ? __builtin_amdgcn_perm(rw, cw0, 0x0C0C0000u | ((u32)(4 + k) << 8) | (u32)(k + 1))
                                 : __builtin_amdgcn_perm(rw, cw1, 0x0C0C0700u);
            const int s0 = s_b62[a0], s1 = s_b62[a1];
            int nBe, Ie_out, De_out, nBo, nIo, nDo;
            m64 te0, te1, to0, to1;
            dp_cell<EDGE>(ve, dpp_row_shr1(Io_out), Do_out, Be, s0, nBe, Ie_out, De_out, te0, te1);
            dp_cell<EDGE>(vo, Ie_out, dpp_row_shl1(De_out), Bo, s1, nBo, nIo, nDo, to0, to1);
            Be = nBe, Bo = nBo, Io_out = nIo, Do_out = nDo;
            // first strict maximum in row-major order == largest (score, 8191 - i) key; inside a lane i = m - l, so the lane keeps
            // (score, 8191 - m) -- the second half is wave-uniform, an SGPR -- and converts at the end; keys of invalid cells
            // (score 0) are < any scoring one
            if (TRACE) {
                const u32 rk = (u32)(8191 - m);
                keyE = max(keyE, ((u32)nBe << 13) | rk);
                keyO = max(keyO, ((u32)nBo << 13) | rk);
            } else {  // score-only: the position of the maximum is not needed
                keyE = max(keyE, max((u32)nBe, (u32)nBo));
            }
            // the iteration's 4 trace bits (odd cell: bits 3-2, even cell: bits 1-0) are shifted into the trace word, one v_addc
            // per bit; invalid cells shift in zeros
            if (TRACE) tw = shl1_in(shl1_in(shl1_in(shl1_in(tw, to1), to0), te1), te0);
        }
        // Trace words are laid out by ITERATION, not by row: word TRACE_WORD((m >> 3) - 1, lane) (kernels.h) holds the lane's cells of iterations
        // 8 (m >> 3) .. + 7, iteration m in nibble 7 - (m & 7) -- so all lanes store together, every second group (a row-aligned
        // layout has two lanes of every row storing in EVERY iteration: a masked store sequence per iteration, 40 of the 158
        // VALU of a traced group).  A lane's last group may end half a word: stored left-aligned.
        if (TRACE) {
            const bool full = (m0 & 4) != 0;
            if (full || m0 + 4 > m_end) tr[TRACE_WORD((m0 >> 3) - 1, l)] = full ? tw : (tw << 16);
            if (full) tw = 0;
        }
    };
    const int int_hi = min(R, ncols) - 3;  // groups m0 in [17, int_hi] are interior for this alignment (all 16 lanes, all 4 steps)
    // every ACTIVE lane's group m0 is interior: one compare against the exec mask (m0 is wave-uniform)
    auto all_interior = [&](int m0) {
        return m0 >= 17 && __builtin_amdgcn_sicmp(m0, int_hi, 41 /*ICMP_SLE*/) == __builtin_amdgcn_read_exec();
    };
    const u8* rp = rcls + (8 - l - 1);       // row classes of the group's four iterations
    const u8* cp = ccls + (8 + l - KB - 1);  // column classes, one column early
    // (fetching the windows one group ahead was measured and is slower: 35.1 against 33.4 ms of align rounds on config 3)
    for (int m0 = 8; m0 <= m_end; m0 += 4, rp += 4, cp += 4) {
        if (all_interior(m0)) {
            // the interior groups of the wave's active alignments run in a loop of their own (one code path: no copies where the
            // two variants of a group would join); an interior group is at least 18 iterations away from its alignment's last
            do {
                const u64 cw = load8u(cp);
                four(m0, load4u(rp), (u32)cw, (u32)(cw >> 32), std::false_type{});
                m0 += 4, rp += 4, cp += 4;
            } while (all_interior(m0));
        }
        const u64 cw = load8u(cp);
        four(m0, load4u(rp), (u32)cw, (u32)(cw >> 32), std::true_type{});
    }
    // lane best: max score, then smallest i, then the even cell (smaller j)
    int best, bi, bj;
    {
        const int sE = (TRACE ? (int)(keyE >> 13) : (int)keyE) - BIAS, sO = TRACE ? (int)(keyO >> 13) - BIAS : 0;
        const int iE = 8191 - (int)(keyE & 8191u) - l, iO = 8191 - (int)(keyO & 8191u) - l;  // m -> i (score-only: unused)
        const bool takeO = (sO > sE) || (sO == sE && iO < iE);
        best = takeO ? sO : sE;
        bi = takeO ? iO : iE;
        bj = bi + 2 * l - KB + (takeO ? 1 : 0);
        if (best == 0) bi = 0, bj = 0;  // nothing scored: (i_max, j_max) stay (0, 0) (1391)
    }
    // reduce (best, bi, bj) over the 16 lanes: max score, then smallest i, then smallest j; the cells evaluated (a counter the
    // oracle keeps too) are the lane's two band offsets' row ranges
    int ncell = cnt_e + cnt_o;
    for (int msk = 8; msk > 0; msk >>= 1) {
        const int ob = __shfl_xor(best, msk, 16), oi = __shfl_xor(bi, msk, 16), oj = __shfl_xor(bj, msk, 16);
        const int oc = __shfl_xor(ncell, msk, 16);
        ncell += oc;
        if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) best = ob, bi = oi, bj = oj;
    }
    // the traceback runs in k_traceback (one thread per alignment: 64 walks per wave instead of 4)
    if (l != 0) return;
    AlnRes r;
    r.maxscore = best, r.aln = 0, r.matches = 0, r.gap = 0, r.cells = ncell, r.pad = 0;
    r.qst = bi, r.qed = bj, r.sst = 0, r.sed = 0;  // (i_max, j_max) parked for k_traceback
    out[slot] = r;
    if (TRACE && tpos_out) tpos_out[slot] = tpos_base + tunit;   // where this task's trace lives (speculative traces)
}

// Traceback from the best cell until a stop cell (1418-1443), walking through row 0 ('-'), column 0
// ('|') and the left boundary cell (i, i-17) ('|'), plus the alignment statistics (1454-1471).
// One thread per alignment; the 2-bit traces come from the slab k_align wrote in the same launch
// sequence (kernel boundary = visibility), with the current trace word cached in a register.
//
// LONG walks (wave mode, round 5).  A thread's walk is a chain of dependent steps -- ~0.4 us each once the wave is alone on its SIMD --
// so a launch lasted as long as its longest walk: 1.6 ms for the 4096 columns of a giant's self-alignment, whatever else it held
// (log-normal set: 4 launches, 7.2 ms per step).  The first `nw` blocks of the launch therefore take the first nw list positions
// (the lists are ordered longest band first) ONE WALK PER WAVE when the band has at least `wrows` rows: the 64 lanes look at the next
// 64 cells up the current diagonal together (their trace codes and residues: one memory round trip), the leading run of plain
// diagonal columns -- code 1, no '-' residue -- is added in one go (columns, matches by popcount; behind a non-gap column the gap
// machine's three counters are all f(-1)), and the column that ends the run takes the single step of the thread walk, served from the
// lane that looked at it.  The other blocks walk 64 alignments each as before and skip what a wave took.
template <bool WAVE>
__device__ __forceinline__ void traceback_one(const u32 tid, const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx,
                                              const u8* __restrict__ q_res, const u32* __restrict__ qoff, const u8* __restrict__ r_res,
                                              const u32* __restrict__ roff, const u32* __restrict__ trace, u32 trace_stride,
                                              const u32* __restrict__ tpos, const u32* __restrict__ tofs, AlnRes* __restrict__ out, u32 nw, int wrows);

__global__ __launch_bounds__(64) void k_traceback(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 ntasks,
                                                  const u8* __restrict__ q_res, const u32* __restrict__ qoff,
                                                  const u8* __restrict__ r_res, const u32* __restrict__ roff,
                                                  const u32* __restrict__ trace, u32 trace_stride, const u32* __restrict__ tpos,
                                                  const u32* __restrict__ tofs, AlnRes* __restrict__ out, u32 nw, int wrows) {
    if (blockIdx.x < nw) {
        traceback_one<true>(blockIdx.x, tasks, ridx, q_res, qoff, r_res, roff, trace, trace_stride, tpos, tofs, out, nw, wrows);
        return;
    }
    const u32 tid = (blockIdx.x - nw) * 64u + threadIdx.x;
    if (tid >= ntasks) return;
    traceback_one<false>(tid, tasks, ridx, q_res, qoff, r_res, roff, trace, trace_stride, tpos, tofs, out, nw, wrows);
}

__device__ __forceinline__ uint4 u1_load16g(const u8* p) {   // unaligned 16-byte global load
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

template <bool WAVE>
__device__ __forceinline__ void traceback_one(const u32 tid, const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx,
                                              const u8* __restrict__ q_res, const u32* __restrict__ qoff, const u8* __restrict__ r_res,
                                              const u32* __restrict__ roff, const u32* __restrict__ trace, u32 trace_stride,
                                              const u32* __restrict__ tpos, const u32* __restrict__ tofs, AlnRes* __restrict__ out, u32 nw, int wrows) {
    const u32 slot = ridx ? ridx[tid] : tid;
    const AlnTask tk = tasks[slot];
    const u32 qb = qoff[tk.q], sb = roff[tk.subj];
    const int lq = min((int)(qoff[tk.q + 1] - qb), (int)tk.qe), ls = min((int)(roff[tk.subj + 1] - sb), (int)tk.se);
    const int qi = min((int)tk.qi, lq), qj = min((int)tk.qj, ls);
    const int la = lq - qi, lb = ls - qj;
    {   // who walks this alignment: a wave, if the band is long and the list position is among the first nw
        const bool by_wave = tid < nw && min(max(la, lb), min(la, lb) + KB) >= wrows;
        if (by_wave != WAVE) return;
    }
    AlnRes r = out[slot];
    const bool swp = !(la < lb);
    const u8* craw = swp ? (r_res + sb + qj) : (q_res + qb + qi);
    const u8* rraw = swp ? (q_res + qb + qi) : (r_res + sb + qj);
    // tpos: traces kept per task (speculative); tofs: per launch position, variable size; else one stride per launch position
    const u32* tr = trace + (size_t)(tpos ? tpos[slot] : tofs ? tofs[tid] : tid) * trace_stride;
    const int bi = r.qst, bj = r.qed;
    const int tagged = r.pad & 1;   // traces written by k_align_pk<true>: 3 diagonal, 1 up (here: 1 diagonal, 3 up) -- swapped on reading
    if (WAVE) {
        const int t = threadIdx.x;
        int i = bi, j = bj, AL = 0, matches = 0, fm1 = 0, f0 = 0, f1 = 0;
        while (i > 0 || j > 0) {
            const int d = j - i + KB;
            const bool inband = i > 0 && j > 0 && d >= 0;   // (wave-uniform)
            // lane t looks at cell (i - t, j - t) of the diagonal; outside the band interior every lane looks at (i, j)
            const int it = inband ? i - t : i, jt = inband ? j - t : j;
            int tcl = 0;
            if (inband && it > 0 && jt > 0) {
                const int m = it + (d >> 1);
                tcl = (int)((tr[TRACE_WORD((m >> 3) - 1, d >> 1)] >> (((7 - (m & 7)) << 2) + ((d & 1) << 1))) & 3u);
                tcl ^= (tcl & tagged) << 1;
            }
            const int a0l = jt > 0 ? (int)craw[jt - 1] : (int)'-', a1l = it > 0 ? (int)rraw[it - 1] : (int)'-';
            int run = 0;
            if (inband) {
                const u64 okm = __ballot(it > 0 && jt > 0 && tcl == 1 && a0l != '-' && a1l != '-');
                run = ~okm ? __builtin_ctzll(~okm) : 64;
                if (run) {
                    const u64 eqm = __ballot(a0l == a1l);
                    AL += run, matches += __builtin_popcountll(run < 64 ? eqm & ((1ull << run) - 1ull) : eqm);
                    f0 = fm1, f1 = fm1;
                    i -= run, j -= run;
                }
                if (run == 64) continue;
                if (!(i > 0 || j > 0)) break;
            }
            // the column at (i, j), looked at by lane `run`: the thread walk's step
            int tc;
            if (i == 0) tc = 2;
            else if (j == 0) tc = 3;
            else if (d < 0) tc = 3;  // left boundary cell (i, i-17): '|'
            else tc = __builtin_amdgcn_readlane(tcl, run);
            if (tc == 0) break;
            ++AL;
            const int a0 = tc != 3 ? __builtin_amdgcn_readlane(a0l, run) : (int)'-', a1 = tc != 2 ? __builtin_amdgcn_readlane(a1l, run) : (int)'-';
            matches += (a0 == a1) ? 1 : 0;
            const bool g0 = a0 == '-', g1 = a1 == '-';
            const int nm1 = g0 ? 1 + f0 : (g1 ? 1 + f1 : fm1), n0 = g1 ? 1 + f1 : fm1, n1 = g0 ? 1 + f0 : fm1;
            fm1 = nm1, f0 = n0, f1 = n1;
            if (tc != 3) --j;
            if (tc != 2) --i;
        }
        if (t == 0) {
            r.aln = AL, r.matches = matches, r.gap = fm1;
            if (swp) r.qst = i + qi, r.qed = bi + qi, r.sst = j + qj, r.sed = bj + qj;
            else r.qst = j + qi, r.qed = bj + qi, r.sst = i + qj, r.sed = bi + qj;
            out[slot] = r;
        }
        return;
    }
    // The reference derives its statistics from the two aligned STRINGS, gap columns spelled '-' (1454-1471): identity
    // compares characters, and the gap counter is a three-state machine over them (op = -1 / 0 / 1; a '-' in string 0 opens
    // when op != 0, else one in string 1 opens when op != 1, else op resets) -- so a run of L gap columns counts ceil(L / 2)
    // openings, and a literal '-' residue behaves like a gap character.  The walk below runs from the END of the alignment,
    // the machine from its start: f_s = openings counted from the current column to the end if the machine enters it in
    // state s; three counters, updated per column, and the answer is f(-1) at the first column.
    int i = bi, j = bj, AL = 0, matches = 0, fm1 = 0, f0 = 0, f1 = 0;
    // The walk in rounds of EIGHT BRANCH-FREE columns behind one checkpoint (round 6).  The 64 walks of a wave reach their boundaries (next
    // trace word, next residues) at different steps; written with a branch per case the loop compiled to ~25 exec-mask regions per column and
    // the kernel was bound by the scalar unit (150 scalar beside 100 vector instructions per column).  Now every eighth column each lane makes
    // sure it holds what eight columns can touch -- its lane's group of four trace blocks and the group behind it (two aligned 16-byte
    // pieces, TRACE_WORD), 16 residues of either sequence ending at the current column -- loading only what is missing (a piece fetched
    // again comes from the Infinity Cache, not the L1: unconditional reloads made the launch twice as long), and the columns are selects: m
    // falls by at most one per column, so by at most one block per round; a gap column that moves the walk to another lane's words waits
    // for the next round.  Config 3, 720 k walks per launch: 1.19 -> 0.83 ms.  Measured and not kept: 32-byte residue windows (1.17 ms), a
    // run of up to eight plain diagonal columns taken in one go from the shifted trace nibbles + two single columns per round (1.22 ms: the
    // wave lasts as long as its slowest lane's rounds, and the 64-bit arithmetic of the run is not cheaper than eight selected columns).
    // Finished walks idle until the wave's longest ends, as before.
    bool run = true;   // no stop cell met yet
    auto byte16 = [](const uint4& v, int o) -> int {
        const u32 lo = (o & 8) ? v.z : v.x, hi = (o & 8) ? v.w : v.y;
        return (int)__builtin_amdgcn_ubfe((o & 4) ? hi : lo, (u32)(o & 3) << 3, 8u);
    };
    auto word4 = [](const uint4& v, int blk) -> u32 {
        const u32 lo = (blk & 1) ? v.y : v.x, hi = (blk & 1) ? v.w : v.z;
        return (blk & 2) ? hi : lo;
    };
    int gkey = -1, ngkey = -1;   // first words (TRACE_WORD(block & ~3, lane)) of the two trace pieces held; -1: none
    uint4 gv = make_uint4(0, 0, 0, 0), ngv = make_uint4(0, 0, 0, 0);
    int cb = 0x40000000, rb = 0x40000000;   // residue positions cb .. cb + 15 / rb .. rb + 15 are held (none yet)
    uint4 cw = make_uint4(0, 0, 0, 0), rw = make_uint4(0, 0, 0, 0);
    while (__ballot(run && (i > 0 || j > 0)) != 0ull) {
        {   // ---- checkpoint ----
            const int d = j - i + KB;
            const bool inb = run && i > 0 && j > 0 && d >= 0;
            const int m = i + (d >> 1), blk = (m >> 3) - 1;
            const int g = (int)TRACE_WORD(blk & ~3, d >> 1);
            if (inb && g != gkey) {
                if (g == ngkey) gv = ngv;
                else gv = *reinterpret_cast<const uint4*>(tr + g);
                gkey = g;
            }
            if (inb && g >= 64 && ngkey != g - 64) ngkey = g - 64, ngv = *reinterpret_cast<const uint4*>(tr + (g - 64));
        }
        // the round's residues are positions j - 8 .. j - 1 at most (the arrays are padded behind: a window may reach past its sequence)
        // (32-byte windows -- half the line fetches, a select level more per residue -- measured: 1.17 against 0.83 ms per launch)
        if (run && j > 0 && (j - 8 < cb || j > cb + 16)) cb = max(j - 16, 0), cw = u1_load16g(craw + cb);
        if (run && i > 0 && (i - 8 < rb || i > rb + 16)) rb = max(i - 16, 0), rw = u1_load16g(rraw + rb);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool alive = run && (i > 0 || j > 0);
            const int d = j - i + KB;
            const bool inb = i > 0 && j > 0 && d >= 0;
            const int m = i + (d >> 1), blk = (m >> 3) - 1;  // the iteration that computed the cell in lane d >> 1 (k_align's trace layout)
            const int g = (int)TRACE_WORD(blk & ~3, d >> 1);
            const bool cur = g == gkey, prv = g == ngkey;
            const u32 w = prv ? word4(ngv, blk) : word4(gv, blk);
            int code = (int)((w >> (((7 - (m & 7)) << 2) + ((d & 1) << 1))) & 3u);
            code ^= (code & tagged) << 1;
            const bool held = !inb || cur || prv;            // else: another lane's words, or a block further back -- next round
            const int tc = i == 0 ? 2 : (j == 0 ? 3 : (d < 0 ? 3 : code));   // row 0: '-', column 0 and the left boundary cell (i, i-17): '|'
            const bool go = alive && held && tc != 0;
            run = run && !(alive && held && tc == 0);
            // the column's two characters (1419-1432)
            const int a0 = tc != 3 ? byte16(cw, j - 1 - cb) : (int)'-', a1 = tc != 2 ? byte16(rw, i - 1 - rb) : (int)'-';
            const bool g0 = a0 == '-', g1 = a1 == '-';
            const int nm1 = g0 ? 1 + f0 : (g1 ? 1 + f1 : fm1), n0 = g1 ? 1 + f1 : fm1, n1 = g0 ? 1 + f0 : fm1;
            AL += go ? 1 : 0;
            matches += (go && a0 == a1) ? 1 : 0;
            fm1 = go ? nm1 : fm1, f0 = go ? n0 : f0, f1 = go ? n1 : f1;
            j -= (go && tc != 3) ? 1 : 0;
            i -= (go && tc != 2) ? 1 : 0;
        }
    }
    r.aln = AL, r.matches = matches, r.gap = fm1;
    if (swp) {  // rows = query, columns = subject (1473-1474)
        r.qst = i + qi, r.qed = bi + qi, r.sst = j + qj, r.sed = bj + qj;
    } else {
        r.qst = j + qi, r.qed = bj + qi, r.sst = i + qj, r.sed = bi + qj;
    }
    out[slot] = r;
}

u32 align_trace_stride(int max_cols_plus) {
    // words per task: 16 per block of 8 iterations, iterations 8 .. R + 15, R <= max_cols_plus; rounded to groups of four blocks (256 B)
    u32 w = (u32)((max_cols_plus + 15) / 8 + 1) * 16u;
    return (w + 63u) & ~63u;   // whole groups of four blocks (TRACE_WORD)
}

// Trace words of one task: what a band of R rows writes (align_trace_stride(R + 1)), in units of TRACE_UNIT words.  A launch whose
// tasks get their trace room from the exclusive scan of these (tofs) needs sum(units) instead of tasks x the longest band's stride:
// with one 4096-row window in the batch every 300-row alignment used to own -- and its traceback to stride over -- 33 KB.
#define TRACE_UNIT 32
__global__ __launch_bounds__(256) void k_trace_units(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 n, const u32* __restrict__ qoff,
                                                     const u32* __restrict__ roff, u32* __restrict__ units) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    if (t > n) return;
    if (t == n) {
        units[t] = 0;
        return;
    }
    const AlnTask tk = tasks[ridx ? ridx[t] : t];
    const int lq = min((int)(qoff[tk.q + 1] - qoff[tk.q]), (int)tk.qe), ls = min((int)(roff[tk.subj + 1] - roff[tk.subj]), (int)tk.se);
    const int la = lq - min((int)tk.qi, lq), lb = ls - min((int)tk.qj, ls);
    const int R = min(max(la, lb), min(la, lb) + KB);
    const u32 w = (u32)((R + 1 + 15) / 8 + 1) * 16u;
    units[t] = ((w + 63u) & ~63u) / TRACE_UNIT;
}
u32 align_trace_unit() { return TRACE_UNIT; }
void launch_trace_units(const AlnTask* tasks, const u32* ridx, u32 n, const u32* qoff, const u32* roff, u32* units /*n + 1*/, hipStream_t st) {
    hipLaunchKernelGGL(k_trace_units, dim3((n + 1 + 255) / 256), dim3(256), 0, st, tasks, ridx, n, qoff, roff, units);
}

// list positions offered to the wave walks (SOHIT_TRACE_WAVE_ROWS = 0: none)
static inline u32 traceback_waves(u32 ntasks) {
    return tune().trace_wave_rows > 0 ? (u32)std::min<long long>(ntasks, std::max<long long>(0, tune().trace_wave_max)) : 0u;
}

// with_traceback = false: scores only (trace may be null); true: traces + traceback statistics.  n_wide (with_traceback): the leading
// list positions that need the 32-bit cells; the rest is aligned by the packed kernel (ntasks: all of them by the 32-bit one)
void launch_align(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u8* q_scls, const u8* q_scls4, const u32* qoff,
                  const u8* r_res, const u8* r_scls, const u8* r_scls4, const u32* roff, const signed char* b62g, u32* trace, u32 trace_stride,
                  const u32* tofs, AlnRes* out, bool with_traceback, hipStream_t st, u32 n_wide, PkCls pk) {
    if (!ntasks) return;
    if (!with_traceback) {
        hipLaunchKernelGGL((k_align<false>), dim3((ntasks + 15) / 16), dim3(256), 0, st, tasks, ridx, ntasks, q_scls, q_scls4, qoff, r_scls, r_scls4,
                           roff, b62g, trace, trace_stride, (const u32*)nullptr, out, (u32*)nullptr, 0u);
        return;
    }
    launch_align_traced(tasks, ridx, ntasks, q_scls, q_scls4, qoff, r_scls, r_scls4, roff, b62g, trace, trace_stride, tofs, out, nullptr, 0u, st, n_wide, pk);
    const u32 nw = traceback_waves(ntasks);
    hipLaunchKernelGGL(k_traceback, dim3(nw + (ntasks + 63) / 64), dim3(64), 0, st, tasks, ridx, ntasks, q_res, qoff, r_res, roff, trace,
                       trace_stride, (const u32*)nullptr, tofs, out, nw, (int)tune().trace_wave_rows);
}

void launch_align_traced(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_scls, const u8* q_scls4, const u32* qoff, const u8* r_scls,
                         const u8* r_scls4, const u32* roff, const signed char* b62g, u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out,
                         u32* tpos_out, u32 tpos_base, hipStream_t st, u32 n_wide, PkCls pk) {
    if (!ntasks) return;
    n_wide = pk.q ? std::min(n_wide, ntasks) : ntasks;   // (no padded arrays given: everything by the 32-bit kernel)
    if (n_wide)
        hipLaunchKernelGGL((k_align<true>), dim3((n_wide + 15) / 16), dim3(256), 0, st, tasks, ridx, n_wide, q_scls, q_scls4, qoff, r_scls, r_scls4, roff,
                           b62g, trace, trace_stride, tofs, out, tpos_out, tpos_base);
    launch_align_pk_traced(tasks, ridx, n_wide, ntasks, pk, qoff, roff, b62g, trace, trace_stride, tofs, out, tpos_out, tpos_base, st);
}

// the walks alone over list positions whose traces sit at tofs[position] (the alignments were made by an earlier launch_align_traced)
void launch_traceback_tofs(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u32* qoff, const u8* r_res, const u32* roff,
                           const u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out, hipStream_t st) {
    if (!ntasks) return;
    const u32 nw = traceback_waves(ntasks);
    hipLaunchKernelGGL(k_traceback, dim3(nw + (ntasks + 63) / 64), dim3(64), 0, st, tasks, ridx, ntasks, q_res, qoff, r_res, roff, trace, trace_stride,
                       (const u32*)nullptr, tofs, out, nw, (int)tune().trace_wave_rows);
}

void launch_traceback(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u32* qoff, const u8* r_res, const u32* roff,
                      const u32* trace, u32 trace_stride, const u32* tpos, AlnRes* out, hipStream_t st) {
    if (!ntasks) return;
    const u32 nw = traceback_waves(ntasks);
    hipLaunchKernelGGL(k_traceback, dim3(nw + (ntasks + 63) / 64), dim3(64), 0, st, tasks, ridx, ntasks, q_res, qoff, r_res, roff, trace, trace_stride, tpos,
                       (const u32*)nullptr, out, nw, (int)tune().trace_wave_rows);
}
