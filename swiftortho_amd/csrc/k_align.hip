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
// So every step is one row_shr / row_shl DPP move plus ~25 VALU ops, no LDS traffic for the DP
// state; LDS holds only the 24x24 score table.  2-bit traces go to a per-task scratch slab.
#include "common.h"
#include "kernels.h"

#define KB 16  // kbound

__device__ __forceinline__ int dpp_row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_row_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true); }

// trace codes: 0 '*' (stop), 1 '\\' (diag), 2 '-' (left), 3 '|' (up)
__global__ __launch_bounds__(256) void k_align(const AlnTask* __restrict__ tasks, const u32* __restrict__ ridx, u32 ntasks,
                                               const u8* __restrict__ q_res,
                                               const u8* __restrict__ q_scls, const u32* __restrict__ qoff,
                                               const u8* __restrict__ r_res, const u8* __restrict__ r_scls,
                                               const u32* __restrict__ roff, const signed char* __restrict__ b62g,
                                               u32* __restrict__ trace, u32 trace_stride, AlnRes* __restrict__ out) {
    __shared__ signed char s_b62[SCLS_N * SCLS_N];
    for (int i = threadIdx.x; i < SCLS_N * SCLS_N; i += 256) s_b62[i] = b62g[i];
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
    const u8* ccls = swp ? (r_scls + sb + qj) : (q_scls + qb + qi);
    const u8* rcls = swp ? (q_scls + qb + qi) : (r_scls + sb + qj);
    const u8* craw = swp ? (r_res + sb + qj) : (q_res + qb + qi);
    const u8* rraw = swp ? (q_res + qb + qi) : (r_res + sb + qj);
    const int R = min(nrows, ncols + KB);  // rows beyond ncols + 16 have an empty band
    u32* tr = trace + (size_t)tid * trace_stride;

    int pv = 0, pv2 = 0;  // own packed results (B << 2 | trace) of steps t-1, t-2
    int best = 0, bi = 0, bj = 0, ncell = 0;
    u32 tw = 0;
    int rc = 0, cc = 0;  // current row / column residue classes
    const int t_end = 2 * R + 31;
    for (int t = 2 + KB; t <= t_end; ++t) {
        const int par = t & 1;
        const int d = 2 * l + par;
        const int i = (t - d) >> 1;
        const int j = i + d - KB;
        const int nbL = dpp_row_shr1(pv);
        const int nbR = dpp_row_shl1(pv);
        const int left = par ? pv : nbL;
        const int up = par ? nbR : pv;
        const int dg = pv2;
        const bool valid = (i >= 1) && (i <= R) && (j >= 1) && (j <= ncols);
        int B = 0, trc = 0;
        if (valid) {
            // residues change every other step; reloading both keeps the code simple (L1 hits)
            rc = rcls[i - 1];
            cc = ccls[j - 1];
            const int I = (left >> 2) + (((left & 3) == 2) ? -1 : -11);
            const int M = (dg >> 2) + s_b62[rc * SCLS_N + cc];
            const int D = (up >> 2) + (((up & 3) == 3) ? -1 : -11);
            B = max(max(0, I), max(M, D));
            trc = (B == M) ? 1 : (B == I) ? 2 : (B == D) ? 3 : 0;
            ++ncell;
            if (B > best) best = B, bi = i, bj = j;
        }
        if (i >= 1 && i <= R) {
            tw |= (u32)trc << ((((i - 1) & 7) << 2) + (par << 1));
            if (par && ((((i - 1) & 7) == 7) || i == R)) {
                tr[((i - 1) >> 3) * 16 + l] = tw;
                tw = 0;
            }
        }
        pv2 = pv;
        pv = (B << 2) | trc;
    }
    // reduce (best, bi, bj) over the 16 lanes: max score, then smallest i, then smallest j
    for (int m = 8; m > 0; m >>= 1) {
        const int ob = __shfl_xor(best, m, 16), oi = __shfl_xor(bi, m, 16), oj = __shfl_xor(bj, m, 16);
        const int oc = __shfl_xor(ncell, m, 16);
        ncell += oc;
        if (ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)))) best = ob, bi = oi, bj = oj;
    }
    // make the trace words of all 16 lanes visible to the lane that walks them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (l != 0) return;
    int i = bi, j = bj, AL = 0, matches = 0, gaps = 0, run = 0, rtype = 0;
    while (i > 0 || j > 0) {
        int tc;
        if (i == 0) tc = 2;
        else if (j == 0) tc = 3;
        else {
            const int d = j - i + KB;
            if (d < 0) tc = 3;  // left boundary cell (i, i-17): '|'
            else tc = (int)((tr[((i - 1) >> 3) * 16 + (d >> 1)] >> ((((i - 1) & 7) << 2) + ((d & 1) << 1))) & 3u);
        }
        if (tc == 0) break;
        ++AL;
        if (tc == 1) {
            matches += (craw[j - 1] == rraw[i - 1]) ? 1 : 0;
            --i, --j;
            run = 0, rtype = 0;
        } else {
            // a run of L same-direction gap columns counts ceil(L / 2) openings (1462-1469)
            if (rtype != tc) run = 0, rtype = tc;
            if ((run & 1) == 0) ++gaps;
            ++run;
            if (tc == 2) --j;
            else --i;
        }
    }
    AlnRes r;
    r.maxscore = best, r.aln = AL, r.matches = matches, r.gap = gaps, r.cells = ncell, r.pad = 0;
    if (swp) {  // rows = query, columns = subject (1473-1474)
        r.qst = i + qi, r.qed = bi + qi, r.sst = j + qj, r.sed = bj + qj;
    } else {
        r.qst = j + qi, r.qed = bj + qi, r.sst = i + qj, r.sed = bi + qj;
    }
    out[slot] = r;
}

u32 align_trace_stride(int max_cols_plus) {
    // words per task: ceil(R / 8) * 16, R <= max_cols_plus; rounded to 32 words (128 B)
    u32 w = (u32)((max_cols_plus + 7) / 8) * 16u;
    return (w + 31u) & ~31u;
}

void launch_align(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u8* q_scls, const u32* qoff, const u8* r_res,
                  const u8* r_scls, const u32* roff, const signed char* b62g, u32* trace, u32 trace_stride, AlnRes* out,
                  hipStream_t st) {
    if (!ntasks) return;
    hipLaunchKernelGGL(k_align, dim3((ntasks + 15) / 16), dim3(256), 0, st, tasks, ridx, ntasks, q_res, q_scls, qoff, r_res, r_scls, roff,
                       b62g, trace, trace_stride, out);
}
