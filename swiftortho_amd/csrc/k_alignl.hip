// k_alignl.hip -- score-only banded gapped extension, ONE LANE PER ALIGNMENT PAIR (kswat_st, fsearch.py:1357-1416).
//
// k_align_pk (k_align16.hip) spreads an alignment over the 16 lanes of a DPP row on the anti-diagonal: two cells per lane and
// iteration, neighbours through DPP moves.  That mapping pays for itself with a 15-iteration drain per task, masked groups at the
// start, cells of the band's corners that no lane can skip, and two DPP moves per cell pair: 14 lane-instruction slots per band
// cell delivered on BASELINE config 3 against 9.2 in its interior loop (profiles/r05_sq_align.txt; the tasks of a round are 150
// rows on average, so the fixed costs weigh twice what they would on a full-length band).
//
// The early-stop rounds of phase 2 hold MILLIONS of independent tasks, ordered by band rows (sort_by_rows, host.hip).  So here a lane
// owns two whole alignments (the two 16-bit halves of every register, as in k_align_pk) and walks their band row by row, cell by
// cell: the 32 cells of the previous row stay in 2 x 32 registers (value with tag 3, candidate for the cell below), the left
// neighbour is the value just computed, and nothing crosses lanes -- no DPP, no drain, no masked groups, no reduction at the end.
// Neighbouring lanes hold tasks of (nearly) the same number of rows, so a wave's lanes finish together.
//
// The arithmetic is k_align_pk's, bit for bit: tagged cells  w = (score << 2 | tag) + 44  (tag 3 diagonal, 2 left, 1 up, 0 stop: one
// max resolves score and trace priority), the gap costs of the two outgoing candidates picked by the cell's own tag with one v_perm
// each, sentinel classes (-100 against everything) wherever a window leaves its sequence, the lane-private score table
// (row class, column class, lane mod 32) at (row << 11) | (col << 6) | (lane32 << 1).  See the head of k_align16.hip.
//
// Row i (1-based DP row; residue i - 1) holds the band cells d = 0 .. 31 at DP column j = i + d - 16 (residue j - 1):
//     I = left neighbour's candidate (d = 0: the out-of-band boundary cell, "score -11" = 0)
//     D = candidate of (i - 1, d + 1)              (d = 31: never written by the DP = 0, fsearch.py:1379-1389)
//     M = value of (i - 1, d) with tag 3 + 4 x substitution score
// Columns left of the matrix (j < 1) read sentinel classes put into the window when it is set up; rows / columns behind a sequence
// read the 48 sentinel bytes k_pad_cls leaves behind every sequence, and a lane whose alignment has ended feeds sentinel rows.
// Only tasks whose windows END where their sequences end come here (no tile of a 4096+-residue sequence: the host decides).
//
// Per cell pair: 4 address + 1 combine + 10 recurrence + 1 maximum = 16 vector instructions = 8 per cell.
#include "common.h"
#include "kernels.h"

#define AL_TAB (25 * 2048)
#define AL_STOP 0x002C002Cu     // score 0, tag 0 (+ 44) in both halves
#define AL_TAG3 0x00030003u
#define AL_CI 0xD3FCD5D6u       // low bytes of (candidate for the right neighbour) - w by tag: -42, -43, -4, -45
#define AL_CD 0xD2D3FCD5u       // low bytes of (candidate for the lower neighbour) - w by tag: -43, -4, -45, -46
#define AL_SENT_ROW4 0x18181818u    // four sentinel row classes (24)
#define AL_SENT_COL8 0xC0C0C0C0u    // four sentinel column classes (24) * 8
#define AL_THREADS 512
#define AL_G 4             // rows per block = cells per step (one of every row of the block)

typedef short al_pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 al_add(u32 a, u32 b) {
    const al_pk16 r = __builtin_bit_cast(al_pk16, a) + __builtin_bit_cast(al_pk16, b);
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 al_max(u32 a, u32 b) {
    const al_pk16 r = __builtin_elementwise_max(__builtin_bit_cast(al_pk16, a), __builtin_bit_cast(al_pk16, b));
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ u32 al_load4(const u8* p) {
    u32 w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

struct AlSide {
    const u8* rcls;   // row classes, residue 0 of the window
    const u8* ccls;   // column classes * 4, residue 0 of the window
    int nrows, ncols, R, ncell;
    u32 slot;
};

__device__ __forceinline__ void al_setup(AlSide& s, const AlnTask& tk, u32 slot, const u8* __restrict__ q_scls, const u8* __restrict__ q_scls4,
                                         const u32* __restrict__ qoff, const u8* __restrict__ r_scls, const u8* __restrict__ r_scls4,
                                         const u32* __restrict__ roff) {
    const u32 qb = qoff[tk.q], sb = roff[tk.subj];
    const int lenq = (int)(qoff[tk.q + 1] - qb), lens = (int)(roff[tk.subj + 1] - sb);
    const int lq = min(lenq, (int)tk.qe), ls = min(lens, (int)tk.se);
    q_scls += (size_t)PCLS_PAD * tk.q, q_scls4 += (size_t)PCLS_PAD * tk.q, r_scls += (size_t)PCLS_PAD * tk.subj, r_scls4 += (size_t)PCLS_PAD * tk.subj;
    const int qi = min((int)tk.qi, lq), qj = min((int)tk.qj, ls);
    const int la = lq - qi, lb = ls - qj;
    const bool swp = !(la < lb);  // abs(qed - qst) < abs(sed - sst) -> no swap (1364-1369)
    s.ncols = swp ? lb : la, s.nrows = swp ? la : lb;
    s.ccls = swp ? (r_scls4 + sb + qj) : (q_scls4 + qb + qi);
    s.rcls = swp ? (q_scls + qb + qi) : (r_scls + sb + qj);
    s.R = min(s.nrows, s.ncols + 16);
    s.slot = slot;
    // cells the reference evaluates (the counter the oracle keeps too; k_align_pk sums the same terms over its 16 lanes)
    int nc = 0;
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        const int lo_e = max(1, 17 - 2 * l), lo_o = max(1, 16 - 2 * l);
        nc += max(0, min(s.R, s.ncols + 16 - 2 * l) - lo_e + 1) + max(0, min(s.R, s.ncols + 15 - 2 * l) - lo_o + 1);
    }
    s.ncell = nc;
}

__global__ __launch_bounds__(AL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_align_lane(
    const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 ntasks, const u8* __restrict__ q_scls, const u8* __restrict__ q_scls4,
    const u32* __restrict__ qoff, const u8* __restrict__ r_scls, const u8* __restrict__ r_scls4, const u32* __restrict__ roff,
    const signed char* __restrict__ b62g, AlnRes* __restrict__ out, u32* __restrict__ work_ctr /*zeroed*/, unsigned long long* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[AL_TAB];
    unsigned long long t_core = 0, t_real = 0;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) t_core = __builtin_readcyclecounter(), t_real = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < 25 * 32 * 32; i += AL_THREADS) {
        const int a = i >> 10, b = (i >> 5) & 31;   // row class, column class; i & 31 = the lane's copy
        const int v = (a < SCLS_N && b < SCLS_N) ? 4 * (int)b62g[a * SCLS_N + b] : -400;
        reinterpret_cast<short*>(s_tab)[i] = (short)v;
    }
    __syncthreads();
    const u32 lane2 = (threadIdx.x & 31u) << 1;
    const u32 lds0 = (u32)(uintptr_t)s_tab;
    const u32 npairs = (ntasks + 1u) / 2u;
    // PERSISTENT waves: a wave takes the next 64 task pairs of the list (ordered by rows, longest first) until none is left.  A grid
    // of one workgroup per 512 pairs left whole workgroups' wave slots idle behind their longest wave, and the chip's tail behind
    // the last of 2 000 workgroups: a ninth of the launch.
    for (;;) {
    u32 blk = 0;
    if ((threadIdx.x & 63u) == 0) blk = atomicAdd(work_ctr, 1u);
    blk = (u32)__builtin_amdgcn_readfirstlane((int)blk);
    if (blk * 64u >= npairs) break;
    const u32 pair = blk * 64u + (threadIdx.x & 63u);
    const bool live = pair < npairs;
    const u32 tA = live ? 2u * pair : 0u, tB = min(tA + 1u, ntasks - 1u);  // odd tail: the pair is (last, last), written once
    AlSide A, B;
    {
        const u32 sa = ridx ? ridx[tA] : tA, sb = ridx ? ridx[tB] : tB;
        al_setup(A, tasks[sa], sa, q_scls, q_scls4, qoff, r_scls, r_scls4, roff);
        al_setup(B, tasks[sb], sb, q_scls, q_scls4, qoff, r_scls, r_scls4, roff);
    }
    if (!live) A.R = B.R = 0;
    // rows the wave walks: its longest alignment (the list is ordered by rows: the lanes differ by little)
    int rmax = max(A.R, B.R);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rmax = max(rmax, __shfl_xor(rmax, o));
    rmax = __builtin_amdgcn_readfirstlane(rmax);

    u32 W3[32], Dc[33];
#pragma unroll
    for (int d = 0; d < 32; ++d) W3[d] = AL_STOP | AL_TAG3, Dc[d] = 0;   // row 0: score 0, trace '-' (1379-1380); its candidates lose to STOP
    Dc[32] = 0;
    u32 key = AL_STOP;
    // column windows: byte k of cw <-> column residue i0 - 17 + k (i0 = first row of the block), classes * 8; residues < 0: sentinels
    u32 cwA[9], cwB[9];
#pragma unroll
    for (int k = 0; k < 4; ++k) cwA[k] = AL_SENT_COL8, cwB[k] = AL_SENT_COL8;
#pragma unroll
    for (int k = 4; k < 9; ++k) cwA[k] = al_load4(A.ccls + 4 * (k - 4)) << 1, cwB[k] = al_load4(B.ccls + 4 * (k - 4)) << 1;
    const u8* rpA = A.rcls;
    const u8* rpB = B.rcls;
    const u8* cpA = A.ccls + 20;   // next column dword: residues i0 + 19 ...
    const u8* cpB = B.ccls + 20;
    u32 rwA = al_load4(rpA), rwB = al_load4(rpB);

    for (int i0 = 1; i0 <= rmax; i0 += 4) {
        // a lane whose alignment has ended feeds sentinel rows and stops advancing (rows R + 1 ... of its last block read the sentinels
        // behind the sequences)
        const bool onA = i0 <= A.R, onB = i0 <= B.R;
        if (!onA) rwA = AL_SENT_ROW4;
        if (!onB) rwB = AL_SENT_ROW4;
        // the next block's row classes and column dword, on their way while this block is computed
        rpA += onA ? 4 : 0, rpB += onB ? 4 : 0;
        const u32 nrwA = al_load4(rpA), nrwB = al_load4(rpB);
        const u32 ncwA = al_load4(cpA) << 1, ncwB = al_load4(cpB) << 1;
        cpA += onA ? 4 : 0, cpB += onB ? 4 : 0;

        // The block's four rows are walked SKEWED by two cells -- step s computes cell s - 2 r of row r, r = 0 .. 3: a cell needs the
        // upper row's cells d and d + 1, done one and two steps earlier -- so a lane runs four independent dependency chains (a row alone is
        // one chain of five dependent instructions per cell: four waves per SIMD left the vector unit idle a quarter of the time).
        // score lookups of one step: addresses, then eight 16-bit reads -- alignment A's entry zero-extended, alignment B's straight into
        // the high half (ds_read_u16_d16_hi clears the other half on this part: k_align16.hip) -- in flight while the previous step is computed
        u32 lo[AL_G], hi[AL_G];
        auto issue = [&](int st) {
#pragma unroll
            for (int r = 0; r < AL_G; ++r) {
                const int d = st - 2 * r, cb = r + d;
                if (d < 0 || d > 31) continue;
                const u32 sel = 0x0C0C0000u | ((u32)(4 + r) << 8) | (u32)(cb & 3);
                const u32 aA = __builtin_amdgcn_perm(rwA, cwA[cb >> 2], sel), aB = __builtin_amdgcn_perm(rwB, cwB[cb >> 2], sel);
                const u32 adA = lds0 + ((aA << 3) | lane2), adB = lds0 + ((aB << 3) | lane2);
                asm volatile("ds_read_u16 %0, %1" : "=v"(lo[r]) : "v"(adA));
                asm volatile("ds_read_u16_d16_hi %0, %1" : "=v"(hi[r]) : "v"(adB));
            }
        };
        auto wait = [&]() {
            static_assert(AL_G == 4, "the wait statement names every result register");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
        };
        u32 I[AL_G];   // left neighbour's candidate per row; left of the band: the boundary cell (i, i - 17), score 0 and a non-extending trace -> "score -11"
#pragma unroll
        for (int r = 0; r < AL_G; ++r) I[r] = 0;
        issue(0);
#pragma unroll
        for (int st = 0; st < 32 + 2 * (AL_G - 1); ++st) {
            wait();
            u32 S[AL_G];
#pragma unroll
            for (int r = 0; r < AL_G; ++r) S[r] = lo[r] | hi[r];
            if (st + 1 < 32 + 2 * (AL_G - 1)) issue(st + 1);
            __builtin_amdgcn_sched_barrier(0);   // the reads stay in front of the step's arithmetic (the scheduler sinks them to the next wait otherwise)
#pragma unroll
            for (int r = 0; r < AL_G; ++r) {
                const int d = st - 2 * r;
                if (d < 0 || d > 31) continue;
                const u32 M = al_add(W3[d], S[r]);
                const u32 w = al_max(al_max(I[r], Dc[d + 1]), al_max(M, AL_STOP));
                u32 sel;   // per half: byte 0 = the tag (picks a byte of the cost table), byte 1 = 0x0d (constant 0xff: the costs are negative)
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(w), "s"(AL_TAG3), "v"(0x0D000D00u));
                I[r] = al_add(w, __builtin_amdgcn_perm(0u, AL_CI, sel));
                Dc[d] = al_add(w, __builtin_amdgcn_perm(0u, AL_CD, sel));
                W3[d] = w | AL_TAG3;
                key = al_max(key, w);
            }
        }
        // slide the windows by four residues
#pragma unroll
        for (int k = 0; k < 8; ++k) cwA[k] = cwA[k + 1], cwB[k] = cwB[k + 1];
        cwA[8] = ncwA, cwB[8] = ncwB;
        rwA = nrwA, rwB = nrwB;
    }
    if (live) {
    AlnRes r;
    r.aln = 0, r.matches = 0, r.gap = 0, r.pad = 0, r.qst = 0, r.qed = 0, r.sst = 0, r.sed = 0;
    r.maxscore = ((int)(short)(key & 0xFFFFu) - 44) >> 2, r.cells = A.ncell;
    out[A.slot] = r;
    if (tB != tA) {
        r.maxscore = ((int)(short)(key >> 16) - 44) >> 2, r.cells = B.ncell;
        out[B.slot] = r;
    }
    }
    }
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[0] = __builtin_readcyclecounter() - t_core, dbg[1] = __builtin_amdgcn_s_memrealtime() - t_real;
}

void launch_align_lane(const AlnTask* tasks, const u32* ridx, u32 ntasks, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g, AlnRes* out,
                       u32* work_ctr /*one word, zeroed here*/, u32 ncu, hipStream_t st) {
    if (!ntasks) return;
    const u32 pairs = (ntasks + 1) / 2;
    HIP_CHECK(hipMemsetAsync(work_ctr, 0, sizeof(u32), st));
    const u32 grid = std::min<u32>((pairs + AL_THREADS - 1) / AL_THREADS, ncu * 2u);   // two workgroups per CU fill it (LDS, registers)
    static unsigned long long* d_dbg = nullptr;
    const bool dbg = tune().debug;
    if (dbg && !d_dbg) HIP_CHECK(hipMalloc((void**)&d_dbg, 16));
    hipLaunchKernelGGL(k_align_lane, dim3(grid), dim3(AL_THREADS), 0, st, tasks, ridx, ntasks, pk.q, pk.q4, qoff, pk.r, pk.r4, roff, b62g, out, work_ctr,
                       dbg ? d_dbg : nullptr);
    if (dbg) {   // SOHIT_DEBUG: the shader clock the launch ran at (core-clock counter over the constant 100 MHz one, first workgroup)
        unsigned long long h[2];
        HIP_CHECK(hipMemcpyAsync(h, d_dbg, 16, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        fprintf(stderr, "[sohit] k_align_lane: %u tasks, %.3f ms, shader clock %.0f MHz\n", ntasks, (double)h[1] / 1e5, h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0);
    }
}
