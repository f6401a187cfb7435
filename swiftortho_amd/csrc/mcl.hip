// mcl.hip -- Markov clustering of one block of the orthology graph on the device:
// the matrix loop of SwiftOrtho's bin/find_cluster.py `mcl` (652-689) with `normalize` (636-646), as called by `mcl_xyz`
// (1425-1467) -- column normalisation, expansion (sparse x sparse), inflation, pruning, convergence test every 5th round.
//
// The reference runs this loop through scipy on float32 CSR matrices and its groups depend on the exact pruning decisions, so
// every stage reproduces scipy's ARITHMETIC ORDER, not just its mathematics:
//   * column sums accumulate in storage order (row-major; scipy: ones @ A -> csc_matvec over the transposed view), one thread
//     per column walking a stable (column, position) sort of the entries;
//   * expansion is the SMMP row product of scipy's csr_matmat: for every row i, for every stored a_ij in storage order, for every
//     stored b_jk in storage order, sums[k] += a_ij * b_jk in float32 without contraction; the output row lists its columns in
//     REVERSE order of first touch and drops exact zeros.  One wave per row: the j loop is sequential, the lanes take the entries
//     of row j (distinct columns), a hash table (LDS for rows of up to 1024 products, global scratch beyond) keeps sums[k], and
//     an order array indexed by the first-touch ordinal gives the output order without a sort.  Stored zeros (pruned entries)
//     take part exactly like scipy's: they touch columns and can decide the order;
//   * inflation: float32 power through the double-precision pow rounded once = the correctly rounded value.  numpy's float32 power is
//     CPU-dispatched (AVX512: an SVML routine that differs from libm's powf and from the correctly rounded value by one ulp in ~21 %
//     of inputs), so the reference's own last bits depend on the host; a last-bit difference changes a pruning decision only for a
//     value within one ulp of the threshold, and contracting runs are insensitive to it (goldens + random-graph tests: same structure,
//     same read-out, values within one ulp);
//   * pruning keeps the entry and stores 0 (the reference assigns into .data), so the matrix structure -- which the reference's
//     final read-out zips against -- is scipy's.
// Product code behind the C ABI (so_mcl); the scipy loop survives only in tests/ as the oracle.
#include "common.h"
#include "kernels.h"
#include "../../include/sohit.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

#define MCL_EMPTY 0xFFFFFFFFu
#define MCL_LDS_P 1024u          // rows with at most this many products (or old entries) use the LDS tables
#define MCL_LDS_T 2048u

struct Csr {
    u32 n = 0, nnz = 0;
    DevBuf<u32> rp;      // n + 1 row pointers
    DevBuf<u32> idx;
    DevBuf<float> val;
};

__device__ __forceinline__ u32 mcl_hash(u32 k, u32 tmask) { return (k * 2654435761u) & tmask; }

// loads / stores that other lanes of the SAME wave must see in a later step (global scratch path): device-scope atomics bypass
// the per-CU vector cache; LDS pointers (flat) behave the same way
__device__ __forceinline__ float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_u(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// the same ordering point when the tables live in LDS: nothing has to reach, or be re-read from, memory.  (The agent-scope pair above
// writes the L2 back and invalidates the vector cache; the expansion kernel executed it once per matrix entry of every row, tables in
// LDS or not -- 2.3 ms per pass over config 5's 89 k short rows where the arithmetic needs a tenth of that.)
__device__ __forceinline__ void wave_fence_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- column normalisation ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mcl_colkeys(const u32* __restrict__ idx, u32 nnz, u64* __restrict__ keys, u32* __restrict__ pos) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p < nnz) keys[p] = idx[p], pos[p] = p;
}
// y[c] = entries of column c summed in storage order (float32, sequential), flags: [0] some column sums to 0, [1] some sum > 0,
// [2] some sum < 0 or NaN (then y.min() != 0), [3] the smallest column index whose sum is not 0 (atomicMin; preset to ~0)
__global__ __launch_bounds__(256) void k_mcl_colsum(const u64* __restrict__ ckeys /*sorted columns*/, const u32* __restrict__ cpos, u32 nnz,
                                                    const float* __restrict__ val, u32 n, float* __restrict__ y, u32* __restrict__ flags) {
    const u32 c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    u32 lo = 0, hi = nnz;   // first entry with column >= c
    while (lo < hi) {
        const u32 m = (lo + hi) >> 1;
        if (ckeys[m] < c) lo = m + 1;
        else hi = m;
    }
    float s = 0.f;
    for (u32 e = lo; e < nnz && ckeys[e] == c; ++e) s = __fadd_rn(s, val[cpos[e]]);
    y[c] = s;
    if (s == 0.f) atomicOr(&flags[0], 1u);
    if (s > 0.f) atomicOr(&flags[1], 1u);
    if (s < 0.f || s != s) atomicOr(&flags[2], 1u);   // (a NaN sum makes numpy's y.min() NaN: `y.min() == 0` is false, the addend 1e-8)
    if (s != 0.f) atomicMin(&flags[3], c);   // (NaN != 0 too, as numpy's nonzero() sees it)
}
// normalize() (find_cluster.py:636-646): `y = np.asarray(cs)[0]` is the 1-D vector of column sums;
// `if y.min() == 0 and y.max() > 0: y += y.nonzero()[0].min() / 1e3` adds (index of the FIRST column with a non-zero sum) / 1000,
// rounded to float32 like every scalar added in place to a float32 array -- 0.0 only while column 0 itself sums to something --
// `else: y += 1e-8`; then data /= y[column].  (Round 3 had the addend fixed at 0.0: a block whose first gene has only zero-weight or
// fully pruned edges divided 0 by 0 where the reference divides 0 by k / 1000.)
__global__ __launch_bounds__(256) void k_mcl_divide(const u32* __restrict__ idx, float* __restrict__ val, u32 nnz, const float* __restrict__ y,
                                                    const u32* __restrict__ flags) {
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= nnz) return;
    const bool min_is_zero = flags[0] && !flags[2];
    const float eps = (min_is_zero && flags[1]) ? (float)((double)flags[3] / 1e3) : 1e-8f;
    val[p] = __fdiv_rn(val[p], __fadd_rn(y[idx[p]], eps));
}

// ---- expansion --------------------------------------------------------------------------------------------------
// products of row i = sum over its entries a_ij of the length of row j
__global__ __launch_bounds__(256) void k_mcl_products(const u32* __restrict__ rp, const u32* __restrict__ idx, u32 n, u32* __restrict__ P) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    u64 s = 0;
    for (u32 jj = rp[i]; jj < rp[i + 1]; ++jj) s += rp[idx[jj] + 1] - rp[idx[jj]];
    P[i] = (u32)std::min<u64>(s, 0xFFFFFFFEull);
}

// One wave per row.  WRITE = false: ccnt[i] = stored entries of output row i; WRITE = true: the entries go to (cidx, cval) at crp[i].
// Rows above MCL_LDS_P products use the global scratch at soff[i] (u32 units): [T keys][T sums][P order].
// Three instances per pass, by the row's number of products: up to MCL_SMALL_P on a 2.5 KB table (every CU then holds its 32 waves:
// after the first rounds nearly every row of a Markov matrix is that short), up to MCL_LDS_P on the 20 KB one, and the rest on a 40 KB
// table up to MCL_BIG_P products, else in global scratch -- a family of 34 genes squares to ~1150 products per row, just above the
// 20 KB instance's limit, and clearing and probing its tables in HBM cost config 5's 89 k-row block 7 ms per pass (1.64 -> 1.33 s
// for the 100 rounds of that block).  Table sizes only move slots around: the order of the output and every sum are the same.
#define MCL_SMALL_P 128u
#define MCL_SMALL_T 256u
#define MCL_BIG_P 2048u
#define MCL_BIG_T 4096u
template <bool WRITE, int TIER /*0: <= MCL_SMALL_P products, 1: <= MCL_LDS_P, 2: the rest*/>
__global__ __launch_bounds__(64) void k_mcl_spgemm(const u32* __restrict__ rp, const u32* __restrict__ idx, const float* __restrict__ val,
                                                   const u32* __restrict__ rows /*the rows of this instance's tier (host-built list)*/, u32 nrows,
                                                   const u32* __restrict__ P, const u64* __restrict__ soff, u32* __restrict__ scratch,
                                                   u32* __restrict__ ccnt, const u32* __restrict__ crp, u32* __restrict__ cidx, float* __restrict__ cval) {
    constexpr u32 TL = TIER == 0 ? MCL_SMALL_T : TIER == 1 ? MCL_LDS_T : MCL_BIG_T, PL = TIER == 0 ? MCL_SMALL_P : TIER == 1 ? MCL_LDS_P : MCL_BIG_P;
    __shared__ u32 s_keys[TL];
    __shared__ float s_sums[TL];
    __shared__ u32 s_ord[PL];
    if (blockIdx.x >= nrows) return;
    const u32 i = rows[blockIdx.x];
    const u32 lane = threadIdx.x;
    const u32 Pi = P[i];
    u32 *keys, *ord;
    float* sums;
    u32 T;
    if (Pi <= PL) {
        keys = s_keys, sums = s_sums, ord = s_ord, T = TL;
    } else {
        T = 1;
        while (T < 2u * Pi) T <<= 1;
        u32* base = scratch + soff[i];
        keys = base, sums = reinterpret_cast<float*>(base + T), ord = base + 2 * (size_t)T;
    }
    const bool in_lds = Pi <= PL;   // (wave-uniform)
    for (u32 t = lane; t < T; t += 64) st_u(&keys[t], MCL_EMPTY), st_f(&sums[t], 0.f);
    for (u32 t = lane; t < Pi; t += 64) st_u(&ord[t], MCL_EMPTY);
    if (in_lds) wave_fence_lds();
    else wave_fence();
    // The entries a_ij of row i are consumed one after the other (scipy's accumulation order), but their descriptors -- column j, value,
    // extent of row j -- are fetched 64 at a time by the lanes and handed out by lane index: per entry only row j's own entries are
    // still a dependent load (round 3 walked three dependent loads deep per entry: a short row cost ~100 us of pure latency).
    u32 stepbase = 0;
    const u32 r0 = rp[i], rn = rp[i + 1] - r0;
    for (u32 eb = 0; eb < rn; eb += 64) {
        u32 mrb = 0, mrc = 0;
        float mv = 0.f;
        if (eb + lane < rn) {
            const u32 j = idx[r0 + eb + lane];
            mv = val[r0 + eb + lane];
            mrb = rp[j];
            mrc = rp[j + 1] - mrb;
        }
        const u32 cnt = min(64u, rn - eb);
        for (u32 t = 0; t < cnt; ++t) {
            const u32 rb = (u32)__shfl((int)mrb, (int)t), rc = (u32)__shfl((int)mrc, (int)t);
            const float v = __shfl(mv, (int)t);
            for (u32 kk0 = 0; kk0 < rc; kk0 += 64) {
                const u32 kk = kk0 + lane;
                if (kk < rc) {
                    const u32 k = idx[rb + kk];
                    const float prod = __fmul_rn(v, val[rb + kk]);
                    u32 slot = mcl_hash(k, T - 1);
                    for (;;) {
                        const u32 cur = atomicCAS(&keys[slot], MCL_EMPTY, k);
                        if (cur == MCL_EMPTY) {
                            st_u(&ord[stepbase + kk], slot);   // first touch of column k: its ordinal in scipy's linked list
                            break;
                        }
                        if (cur == k) break;
                        slot = (slot + 1) & (T - 1);
                    }
                    st_f(&sums[slot], __fadd_rn(ld_f(&sums[slot]), prod));   // the columns of one row are distinct: no other lane owns this slot now
                }
            }
            if (in_lds) wave_fence_lds();
            else wave_fence();
            stepbase += rc;
        }
    }
    // output in reverse first-touch order, exact zeros dropped
    u32 outn = 0;
    const u32 base = WRITE ? crp[i] : 0u;
    for (u32 top = Pi; top > 0; top -= min(top, 64u)) {
        const bool have = lane < top;
        const u32 slot = have ? ld_u(&ord[top - 1 - lane]) : MCL_EMPTY;
        const float s = slot != MCL_EMPTY ? ld_f(&sums[slot]) : 0.f;
        const bool keep = slot != MCL_EMPTY && s != 0.f;
        const unsigned long long kb = __ballot(keep);
        if (WRITE && keep) {
            const u32 o = base + outn + (u32)__popcll(kb & ((1ull << lane) - 1ull));
            cidx[o] = ld_u(&keys[slot]);
            cval[o] = s;
        }
        outn += (u32)__popcll(kb);
    }
    if (!WRITE && lane == 0) ccnt[i] = outn;
}

// ---- inflation, pruning -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mcl_pow(float* __restrict__ val, u32 nnz, double p) {
    const u32 e = blockIdx.x * 256u + threadIdx.x;
    if (e < nnz) val[e] = (float)pow((double)val[e], p);
}
__global__ __launch_bounds__(256) void k_mcl_prune(float* __restrict__ val, u32 nnz, float thr) {
    const u32 e = blockIdx.x * 256u + threadIdx.x;
    if (e < nnz && val[e] < thr) val[e] = 0.f;
}

// ---- convergence: max over all positions of |x - x_old| - rtol * |x_old| (float32, elementwise as scipy's binops) ----
__global__ __launch_bounds__(64) void k_mcl_diff(const u32* __restrict__ rp, const u32* __restrict__ idx, const float* __restrict__ val,
                                                 const u32* __restrict__ orp, const u32* __restrict__ oidx, const float* __restrict__ oval, u32 n,
                                                 float rtol, const u64* __restrict__ soff, u32* __restrict__ scratch, u32* __restrict__ maxbits) {
    __shared__ u32 s_keys[MCL_LDS_T];
    __shared__ float s_vals[MCL_LDS_T];
    __shared__ u32 s_seen[MCL_LDS_T];
    const u32 i = blockIdx.x;
    if (i >= n) return;
    const u32 lane = threadIdx.x;
    const u32 ob = orp[i], oc = orp[i + 1] - ob, xb = rp[i], xc = rp[i + 1] - xb;
    u32 *keys, *seen;
    float* vals;
    u32 T;
    if (oc <= MCL_LDS_P) {
        keys = s_keys, vals = s_vals, seen = s_seen, T = MCL_LDS_T;
    } else {
        T = 1;
        while (T < 2u * oc) T <<= 1;
        u32* base = scratch + soff[i];
        keys = base, vals = reinterpret_cast<float*>(base + T), seen = base + 2 * (size_t)T;
    }
    for (u32 t = lane; t < T; t += 64) st_u(&keys[t], MCL_EMPTY), st_u(&seen[t], 0u);
    if (oc <= MCL_LDS_P) wave_fence_lds();
    else wave_fence();
    for (u32 e = lane; e < oc; e += 64) {
        const u32 k = oidx[ob + e];
        u32 slot = mcl_hash(k, T - 1);
        for (;;) {
            const u32 cur = atomicCAS(&keys[slot], MCL_EMPTY, k);
            if (cur == MCL_EMPTY || cur == k) break;
            slot = (slot + 1) & (T - 1);
        }
        st_f(&vals[slot], oval[ob + e]);
    }
    if (oc <= MCL_LDS_P) wave_fence_lds();
    else wave_fence();
    float m = 0.f;
    for (u32 e = lane; e < xc; e += 64) {
        const u32 k = idx[xb + e];
        const float a = val[xb + e];
        float b = 0.f;
        u32 slot = mcl_hash(k, T - 1);
        for (;;) {
            const u32 cur = ld_u(&keys[slot]);
            if (cur == MCL_EMPTY) break;
            if (cur == k) {
                b = ld_f(&vals[slot]);
                st_u(&seen[slot], 1u);
                break;
            }
            slot = (slot + 1) & (T - 1);
        }
        const float t = __fsub_rn(fabsf(__fsub_rn(a, b)), __fmul_rn(rtol, fabsf(b)));
        if (m == m) m = (t != t) ? t : fmaxf(m, t);   // scipy's .max() hands a NaN on (np.maximum.reduce); fmaxf would drop it
    }
    if (oc <= MCL_LDS_P) wave_fence_lds();
    else wave_fence();
    for (u32 t = lane; t < T; t += 64) {
        if (ld_u(&keys[t]) != MCL_EMPTY && ld_u(&seen[t]) == 0u) {   // stored in x_old only
            const float b = ld_f(&vals[t]);
            const float tt = __fsub_rn(fabsf(b), __fmul_rn(rtol, fabsf(b)));
            if (m == m) m = (tt != tt) ? tt : fmaxf(m, tt);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(m, o);
        m = (m != m) ? m : (other != other) ? other : fmaxf(m, other);
    }
    // non-negative floats order like their bit patterns, and a NaN's pattern (0x7FC00000 after fabsf) lies above every finite one's:
    // the maximum the host reads back is NaN when any row saw one, and `NaN <= atol` is false there as in the reference
    if (lane == 0 && (m > 0.f || m != m)) atomicMax(maxbits, __float_as_uint(fabsf(m)) | (m != m ? 0x7FC00000u : 0u));
}

struct Mcl {
    hipStream_t st = nullptr;
    DevBuf<u64> ck, ck2;
    DevBuf<u32> cp, cp2, flags, P, ccnt, scratch, small, rows;
    std::vector<u32> tier_rows;
    DevBuf<u64> soff;
    DevBuf<float> y;
    DevBuf<u8> sort_tmp;
    DevBuf<u32> scan_tmp;
    std::vector<u32> hP;
    std::vector<u64> hoff;

    void normalize(Csr& x) {
        if (!x.nnz) return;
        ck.ensure(x.nnz + 2), ck2.ensure(x.nnz + 2), cp.ensure(x.nnz + 2), cp2.ensure(x.nnz + 2), y.ensure((size_t)x.n + 2), flags.ensure(4);
        hipLaunchKernelGGL(k_mcl_colkeys, dim3((x.nnz + 255) / 256), dim3(256), 0, st, x.idx.p, x.nnz, ck.p, cp.p);
        const int bits = ceil_log2((u64)x.n + 1);
        sort_tmp.ensure(sort_pairs_u64_u32_temp_bytes(x.nnz, bits) + 256);
        sort_pairs_u64_u32(sort_tmp.p, sort_tmp.cap, ck.p, ck2.p, cp.p, cp2.p, x.nnz, bits, st);   // stable: positions ascend inside a column
        HIP_CHECK(hipMemsetAsync(flags.p, 0, 3 * sizeof(u32), st));
        HIP_CHECK(hipMemsetAsync(flags.p + 3, 0xFF, sizeof(u32), st));
        hipLaunchKernelGGL(k_mcl_colsum, dim3((x.n + 255) / 256), dim3(256), 0, st, ck2.p, cp2.p, x.nnz, x.val.p, x.n, y.p, flags.p);
        hipLaunchKernelGGL(k_mcl_divide, dim3((x.nnz + 255) / 256), dim3(256), 0, st, x.idx.p, x.val.p, x.nnz, y.p, flags.p);
    }

    // per-row scratch offsets (u32 units) for the rows whose table does not fit LDS; rows are processed in ranges whose scratch fits `budget`
    void plan(const std::vector<u32>& need /*per row: products or old entries*/, u32 words_per_unit_extra, std::vector<std::pair<u32, u32>>& ranges,
              size_t budget_words) {
        const u32 n = (u32)need.size();
        hoff.assign(n, 0);
        ranges.clear();
        u32 lo = 0;
        size_t used = 0;
        for (u32 i = 0; i < n; ++i) {
            size_t w = 0;
            if (need[i] > MCL_LDS_P) {
                size_t T = 1;
                while (T < 2 * (size_t)need[i]) T <<= 1;
                w = 2 * T + (words_per_unit_extra ? (size_t)need[i] : T);
            }
            if (w > budget_words) throw SoError("so_mcl: one matrix row needs more scratch than the device budget");
            if (used + w > budget_words) {
                ranges.emplace_back(lo, i);
                lo = i, used = 0;
            }
            hoff[i] = used;
            used += w;
        }
        ranges.emplace_back(lo, n);
    }

    void expand(const Csr& x, Csr& c) {
        const u32 n = x.n;
        c.n = n;
        c.rp.ensure((size_t)n + 2);
        P.ensure((size_t)n + 2), ccnt.ensure((size_t)n + 2), soff.ensure((size_t)n + 2);
        hipLaunchKernelGGL(k_mcl_products, dim3((n + 255) / 256), dim3(256), 0, st, x.rp.p, x.idx.p, n, P.p);
        hP.resize(n);
        HIP_CHECK(hipMemcpyAsync(hP.data(), P.p, (size_t)n * sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<std::pair<u32, u32>> ranges;
        const size_t budget = (size_t)1 << 28;   // 1 GiB of u32 scratch per range of rows
        plan(hP, 1, ranges, budget);
        size_t maxw = 0;
        for (auto& r : ranges) {
            size_t w = 0;
            for (u32 i = r.first; i < r.second; ++i) {
                if (hP[i] > MCL_LDS_P) {
                    size_t T = 1;
                    while (T < 2 * (size_t)hP[i]) T <<= 1;
                    w = std::max(w, (size_t)hoff[i] + 2 * T + hP[i]);
                }
            }
            maxw = std::max(maxw, w);
        }
        scratch.ensure(maxw + 64);
        HIP_CHECK(hipMemcpyAsync(soff.p, hoff.data(), (size_t)n * sizeof(u64), hipMemcpyHostToDevice, st));
        // The three instances of the kernel each get the LIST of their rows (by products per row, known on the host anyway).  Until
        // round 4 every instance was launched over all rows and a workgroup of another tier left at once: after the first rounds
        // nearly every row of a Markov matrix is a tier-0 row, and the two big-table instances spent 1.6 ms each per pass starting
        // 89 k workgroups (40 KB of LDS apiece) that had nothing to do -- two thirds of the loop's time on config 5.
        // tier_rows = [tier 0 | tier 1 | tier 2] per range of rows; tier_at[range][t] = (first, count)
        tier_rows.assign(n, 0);
        std::vector<std::array<std::pair<u32, u32>, 3>> tier_at(ranges.size());
        {
            u32 at = 0;
            for (size_t ri = 0; ri < ranges.size(); ++ri)
                for (int t = 0; t < 3; ++t) {
                    const u32 first = at;
                    for (u32 i = ranges[ri].first; i < ranges[ri].second; ++i) {
                        const u32 Pi = hP[i];
                        const int ti = Pi <= MCL_SMALL_P ? 0 : Pi <= MCL_LDS_P ? 1 : 2;
                        if (ti == t) tier_rows[at++] = i;
                    }
                    tier_at[ri][t] = {first, at - first};
                }
        }
        rows.ensure((size_t)n + 4);
        if (n) HIP_CHECK(hipMemcpyAsync(rows.p, tier_rows.data(), (size_t)n * sizeof(u32), hipMemcpyHostToDevice, st));
        auto launch_tiers = [&](bool write, size_t ri) {
            for (int t = 0; t < 3; ++t) {
                const u32 first = tier_at[ri][t].first, cnt = tier_at[ri][t].second;
                if (!cnt) continue;
                const dim3 g(cnt), bl(64);
                const u32* rl = rows.p + first;
#define MCL_GO(W, T) hipLaunchKernelGGL((k_mcl_spgemm<W, T>), g, bl, 0, st, x.rp.p, x.idx.p, x.val.p, rl, cnt, P.p, soff.p, scratch.p, ccnt.p, \
                                        W ? c.rp.p : nullptr, W ? c.idx.p : nullptr, W ? c.val.p : nullptr)
                if (write) {
                    if (t == 0) MCL_GO(true, 0);
                    else if (t == 1) MCL_GO(true, 1);
                    else MCL_GO(true, 2);
                } else {
                    if (t == 0) MCL_GO(false, 0);
                    else if (t == 1) MCL_GO(false, 1);
                    else MCL_GO(false, 2);
                }
#undef MCL_GO
            }
        };
        for (size_t ri = 0; ri < ranges.size(); ++ri) launch_tiers(false, ri);
        HIP_CHECK(hipMemsetAsync(ccnt.p + n, 0, sizeof(u32), st));
        scan_tmp.ensure(scan_u32_temp_elems((size_t)n + 1) + 8);
        const u32* tot = scan_u32(ccnt.p, c.rp.p, (size_t)n + 1, false, scan_tmp.p, st);
        u32 nnz = 0;
        HIP_CHECK(hipMemcpyAsync(&nnz, tot, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        c.nnz = nnz;
        c.idx.ensure((size_t)nnz + 2), c.val.ensure((size_t)nnz + 2);
        for (size_t ri = 0; ri < ranges.size(); ++ri) launch_tiers(true, ri);
    }

    bool converged(const Csr& x, const Csr& old, float rtol, float atol) {
        const u32 n = x.n;
        std::vector<u32> orp((size_t)n + 1), oc(n);
        HIP_CHECK(hipMemcpyAsync(orp.data(), old.rp.p, ((size_t)n + 1) * sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (u32 i = 0; i < n; ++i) oc[i] = orp[i + 1] - orp[i];
        std::vector<std::pair<u32, u32>> ranges;
        plan(oc, 0, ranges, (size_t)1 << 28);
        if (ranges.size() != 1) throw SoError("so_mcl: convergence scratch exceeds the device budget");
        size_t maxw = 0;
        for (u32 i = 0; i < n; ++i)
            if (oc[i] > MCL_LDS_P) {
                size_t T = 1;
                while (T < 2 * (size_t)oc[i]) T <<= 1;
                maxw = std::max(maxw, (size_t)hoff[i] + 3 * T);
            }
        scratch.ensure(maxw + 64);
        soff.ensure((size_t)n + 2);
        HIP_CHECK(hipMemcpyAsync(soff.p, hoff.data(), (size_t)n * sizeof(u64), hipMemcpyHostToDevice, st));
        small.ensure(4);
        HIP_CHECK(hipMemsetAsync(small.p, 0, 4 * sizeof(u32), st));
        hipLaunchKernelGGL(k_mcl_diff, dim3(n), dim3(64), 0, st, x.rp.p, x.idx.p, x.val.p, old.rp.p, old.idx.p, old.val.p, n, rtol, soff.p, scratch.p, small.p);
        u32 bits = 0;
        HIP_CHECK(hipMemcpyAsync(&bits, small.p, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        float m;
        memcpy(&m, &bits, 4);
        return m <= atol;
    }
};

void copy_csr(const Csr& a, Csr& b, hipStream_t st) {
    b.n = a.n, b.nnz = a.nnz;
    b.rp.ensure((size_t)a.n + 2), b.idx.ensure((size_t)a.nnz + 2), b.val.ensure((size_t)a.nnz + 2);
    HIP_CHECK(hipMemcpyAsync(b.rp.p, a.rp.p, ((size_t)a.n + 1) * sizeof(u32), hipMemcpyDeviceToDevice, st));
    if (a.nnz) {
        HIP_CHECK(hipMemcpyAsync(b.idx.p, a.idx.p, (size_t)a.nnz * sizeof(u32), hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(b.val.p, a.val.p, (size_t)a.nnz * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
}

thread_local std::string g_mcl_err;

}  // namespace

extern "C" {

const char* so_mcl_last_error(void) { return g_mcl_err.c_str(); }

void so_mcl_free(so_mcl_result* r) {
    if (!r) return;
    free(r->indptr), free(r->indices), free(r->data);
    memset(r, 0, sizeof *r);
}

int so_mcl(int device, int64_t n, const int64_t* indptr, const int32_t* indices, const float* data, double inflation, int32_t max_rounds,
           int32_t check_every, double prune, double rtol, double atol, so_mcl_result* out) {
    try {
        if (!out) throw SoError("so_mcl: result pointer is NULL");
        memset(out, 0, sizeof *out);
        if (n < 0 || n > 0x7FFFFFF0ll || !indptr) throw SoError("so_mcl: bad matrix");
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) throw SoError("so_mcl: no HIP device available (libsohit has no CPU fallback)");
        if (device < 0 || device >= nd) throw SoError("so_mcl: device index out of range");
        HIP_CHECK(hipSetDevice(device));
        const int64_t nnz0 = indptr[n];
        if (nnz0 < 0 || nnz0 > 0x7FFFFFF0ll) throw SoError("so_mcl: matrix too large for 32-bit positions");
        if (check_every < 1) check_every = 1;
        Mcl m;
        HIP_CHECK(hipStreamCreate(&m.st));
        struct Guard {
            hipStream_t s;
            ~Guard() { (void)hipStreamSynchronize(s), (void)hipStreamDestroy(s); }
        } guard{m.st};
        Csr x, old, c;
        x.n = (u32)n, x.nnz = (u32)nnz0;
        x.rp.ensure((size_t)n + 2), x.idx.ensure((size_t)nnz0 + 2), x.val.ensure((size_t)nnz0 + 2);
        {
            std::vector<u32> rp((size_t)n + 1);
            for (int64_t i = 0; i <= n; ++i) rp[(size_t)i] = (u32)indptr[i];
            HIP_CHECK(hipMemcpy(x.rp.p, rp.data(), ((size_t)n + 1) * sizeof(u32), hipMemcpyHostToDevice));
            if (nnz0) {
                HIP_CHECK(hipMemcpy(x.idx.p, indices, (size_t)nnz0 * sizeof(u32), hipMemcpyHostToDevice));
                HIP_CHECK(hipMemcpy(x.val.p, data, (size_t)nnz0 * sizeof(float), hipMemcpyHostToDevice));
            }
        }
        const float prunef = (float)prune, rtolf = (float)rtol, atolf = (float)atol;
        const double pw = (double)(float)inflation;   // numpy raises the float32 data to a float32 exponent
        int rounds = 0, conv = 0;
        Csr *px = &x, *pc = &c;
        for (int i = 0; i < max_rounds; ++i) {
            ++rounds;
            m.normalize(*px);
            if (i % check_every == 0) copy_csr(*px, old, m.st);
            m.expand(*px, *pc);
            if (pc->nnz) hipLaunchKernelGGL(k_mcl_pow, dim3((pc->nnz + 255) / 256), dim3(256), 0, m.st, pc->val.p, pc->nnz, pw);
            std::swap(px, pc);
            if (i % check_every == 0 && i > 0 && m.converged(*px, old, rtolf, atolf)) {
                conv = 1;
                break;
            }
            if (px->nnz) hipLaunchKernelGGL(k_mcl_prune, dim3((px->nnz + 255) / 256), dim3(256), 0, m.st, px->val.p, px->nnz, prunef);
        }
        HIP_CHECK(hipStreamSynchronize(m.st));
        HIP_CHECK(hipGetLastError());
        out->n = n, out->nnz = px->nnz, out->rounds = rounds, out->converged = conv;
        out->indptr = (int64_t*)malloc(((size_t)n + 1) * sizeof(int64_t));
        out->indices = (int32_t*)malloc(std::max<size_t>(1, px->nnz) * sizeof(int32_t));
        out->data = (float*)malloc(std::max<size_t>(1, px->nnz) * sizeof(float));
        if (!out->indptr || !out->indices || !out->data) {
            so_mcl_free(out);
            throw SoError("so_mcl: out of host memory");
        }
        std::vector<u32> rp((size_t)n + 1);
        HIP_CHECK(hipMemcpy(rp.data(), px->rp.p, ((size_t)n + 1) * sizeof(u32), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i <= n; ++i) out->indptr[i] = rp[(size_t)i];
        if (px->nnz) {
            HIP_CHECK(hipMemcpy(out->indices, px->idx.p, (size_t)px->nnz * sizeof(u32), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(out->data, px->val.p, (size_t)px->nnz * sizeof(float), hipMemcpyDeviceToHost));
        }
        g_mcl_err.clear();
        return 0;
    } catch (const std::exception& e) {
        g_mcl_err = e.what();
        return 1;
    }
}

}  // extern "C"
