// k_util.hip -- hand-written device-wide prefix sum (reduce / scan block sums / apply).
// Wave = 64 lanes on gfx950; 256-thread blocks = 4 waves; 16 B per lane loads.
#include "common.h"
#include "kernels.h"

#define SCAN_THREADS 256
#define SCAN_ITERS 8
#define SCAN_TILE (SCAN_THREADS * 4 * SCAN_ITERS)  // items per block

// four values at a pointer that is only 4-byte aligned (seed_pass scans sub-ranges that start at any slot): still one dwordx4 access
__device__ __forceinline__ uint4 ld4(const u32* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void st4(u32* p, const uint4& v) { __builtin_memcpy(p, &v, 16); }

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ u32 block_excl_scan_u32(u32 v, u32* lds4, u32* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 inc = wave_incl_scan_u32(v, lane);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        u32 s = lds4[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const u32* __restrict__ in, size_t n, u32* __restrict__ blocksums) {
    __shared__ u32 lds4[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE;
    u32 s = 0;
#pragma unroll
    for (int it = 0; it < SCAN_ITERS; ++it) {
        size_t i = base + (size_t)it * SCAN_THREADS * 4 + (size_t)threadIdx.x * 4;
        if (i + 3 < n) {
            uint4 v = ld4(in + i);
            s += v.x + v.y + v.z + v.w;
        } else {
            for (int k = 0; k < 4; ++k)
                if (i + k < n) s += in[i + k];
        }
    }
    u32 tot;
    block_excl_scan_u32(s, lds4, &tot);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = tot;
}

// single block: exclusive scan of blocksums[0..nb) in place; blocksums[nb] = grand total
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocksums(u32* __restrict__ blocksums, size_t nb) {
    __shared__ u32 lds4[4];
    u32 carry = 0;
    for (size_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        size_t i = b0 + threadIdx.x;
        u32 v = i < nb ? blocksums[i] : 0;
        u32 tot;
        u32 ex = block_excl_scan_u32(v, lds4, &tot);
        if (i < nb) blocksums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) blocksums[nb] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const u32* __restrict__ in, u32* __restrict__ out, size_t n,
                                                             const u32* __restrict__ blocksums, int inclusive) {
    __shared__ u32 lds4[4];
    size_t base = (size_t)blockIdx.x * SCAN_TILE;
    u32 carry = blocksums[blockIdx.x];
#pragma unroll 1
    for (int it = 0; it < SCAN_ITERS; ++it) {
        size_t i = base + (size_t)it * SCAN_THREADS * 4 + (size_t)threadIdx.x * 4;
        u32 a = 0, b = 0, c = 0, d = 0;
        if (i + 3 < n) {
            uint4 v = ld4(in + i);
            a = v.x, b = v.y, c = v.z, d = v.w;
        } else {
            if (i < n) a = in[i];
            if (i + 1 < n) b = in[i + 1];
            if (i + 2 < n) c = in[i + 2];
            if (i + 3 < n) d = in[i + 3];
        }
        u32 s = a + b + c + d, tot;
        u32 ex = carry + block_excl_scan_u32(s, lds4, &tot);
        u32 o0, o1, o2, o3;
        if (inclusive) o0 = ex + a, o1 = o0 + b, o2 = o1 + c, o3 = o2 + d;
        else o0 = ex, o1 = ex + a, o2 = o1 + b, o3 = o2 + c;
        if (i + 3 < n) {
            st4(out + i, make_uint4(o0, o1, o2, o3));
        } else {
            if (i < n) out[i] = o0;
            if (i + 1 < n) out[i + 1] = o1;
            if (i + 2 < n) out[i + 2] = o2;
            if (i + 3 < n) out[i + 3] = o3;
        }
        carry += tot;
    }
}

// ---- the seed slots' two scans in one (seed_pass, one alphabet x one pattern) ----------------------------------------------------
// A pass needs, per seed slot t, the exclusive prefix of its effective hit count c(t) = mark[t] ? scnt[t] : 0 (hit ordinals) and of
// [c(t) != 0] (compacted seed index).  Until round 6: k_effcnt wrote c and the flag to two arrays and two scans read them back -- 37
// bytes per slot through six launches.  Here the pair travels as ONE 64-bit value, flag count << 32 | hits (a pass holds fewer than 2^32
// hits: seed_stage), computed from mark / scnt where it is needed: 5 bytes read by the reduce, 5 read + 8 written by the apply.
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up((unsigned long long)v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u64 block_excl_scan_u64(u64 v, u64* lds4, u64* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u64 inc = wave_incl_scan_u64(v, lane);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        const u64 s = lds4[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
// four slots from i on: their effective counts (0 beyond n)
__device__ __forceinline__ uint4 eff4(const u8* __restrict__ mark, const u32* __restrict__ scnt, size_t i, size_t n) {
    uint4 c = make_uint4(0, 0, 0, 0);
    if (i + 3 < n) {
        u32 m;
        __builtin_memcpy(&m, mark + i, 4);
        const uint4 v = ld4(scnt + i);
        c.x = (m & 0xFFu) ? v.x : 0u, c.y = (m & 0xFF00u) ? v.y : 0u, c.z = (m & 0xFF0000u) ? v.z : 0u, c.w = (m & 0xFF000000u) ? v.w : 0u;
    } else {
        if (i < n) c.x = mark[i] ? scnt[i] : 0u;
        if (i + 1 < n) c.y = mark[i + 1] ? scnt[i + 1] : 0u;
        if (i + 2 < n) c.z = mark[i + 2] ? scnt[i + 2] : 0u;
    }
    return c;
}
__device__ __forceinline__ u64 pair4(const uint4& c) {
    return ((u64)((c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0)) << 32) + (u64)c.x + c.y + c.z + c.w;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_effscan_reduce(const u8* __restrict__ mark, const u32* __restrict__ scnt, size_t n,
                                                                 u64* __restrict__ blocksums) {
    __shared__ u64 lds4[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    u64 s = 0;
#pragma unroll
    for (int it = 0; it < SCAN_ITERS; ++it) s += pair4(eff4(mark, scnt, base + (size_t)it * SCAN_THREADS * 4 + (size_t)threadIdx.x * 4, n));
    u64 tot;
    block_excl_scan_u64(s, lds4, &tot);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocksums64(u64* __restrict__ blocksums, size_t nb) {
    __shared__ u64 lds4[4];
    u64 carry = 0;
    for (size_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
        const size_t i = b0 + threadIdx.x;
        const u64 v = i < nb ? blocksums[i] : 0;
        u64 tot;
        const u64 ex = block_excl_scan_u64(v, lds4, &tot);
        if (i < nb) blocksums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) blocksums[nb] = carry;   // low word: the pass's hits, high word: its non-empty seeds
}

__global__ __launch_bounds__(SCAN_THREADS) void k_effscan_apply(const u8* __restrict__ mark, const u32* __restrict__ scnt, size_t n,
                                                                const u64* __restrict__ blocksums, u32* __restrict__ hoff, u32* __restrict__ cidx) {
    __shared__ u64 lds4[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    u64 carry = blocksums[blockIdx.x];
#pragma unroll 1
    for (int it = 0; it < SCAN_ITERS; ++it) {
        const size_t i = base + (size_t)it * SCAN_THREADS * 4 + (size_t)threadIdx.x * 4;
        const uint4 c = eff4(mark, scnt, i, n);
        u64 tot;
        const u64 ex = carry + block_excl_scan_u64(pair4(c), lds4, &tot);
        const u32 h0 = (u32)ex, h1 = h0 + c.x, h2 = h1 + c.y, h3 = h2 + c.z;
        const u32 k0 = (u32)(ex >> 32), k1 = k0 + (c.x != 0), k2 = k1 + (c.y != 0), k3 = k2 + (c.z != 0);
        if (i + 3 < n) {
            st4(hoff + i, make_uint4(h0, h1, h2, h3));
            st4(cidx + i, make_uint4(k0, k1, k2, k3));
        } else {
            if (i < n) hoff[i] = h0, cidx[i] = k0;
            if (i + 1 < n) hoff[i + 1] = h1, cidx[i + 1] = k1;
            if (i + 2 < n) hoff[i + 2] = h2, cidx[i + 2] = k2;
        }
        carry += tot;
    }
}

size_t effscan_temp_elems(size_t n) { return 2 * ((n + SCAN_TILE - 1) / SCAN_TILE + 2) + 2; }   // in u32 (the buffer is used as u64)

// hoff / cidx = exclusive prefixes of c(t) / [c(t) != 0] over the n slots at mark / scnt; returns the device address of the totals,
// two u32: {hits, non-empty seeds}.  temp: effscan_temp_elems(n) u32, 8-byte aligned.
const u32* effscan(const u8* mark, const u32* scnt, size_t n, u32* hoff, u32* cidx, u32* temp, hipStream_t st) {
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    u64* t64 = reinterpret_cast<u64*>(temp);
    if (nb == 0) {
        HIP_CHECK(hipMemsetAsync(t64, 0, sizeof(u64), st));
        return temp;
    }
    hipLaunchKernelGGL(k_effscan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, mark, scnt, n, t64);
    hipLaunchKernelGGL(k_scan_blocksums64, dim3(1), dim3(SCAN_THREADS), 0, st, t64, nb);
    hipLaunchKernelGGL(k_effscan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, mark, scnt, n, t64, hoff, cidx);
    return reinterpret_cast<const u32*>(t64 + nb);
}

// short inputs (per-query counters: one value per query of a batch): ONE block of 1024 threads walks the input 16 K values at a
// time with a running carry -- one launch instead of three.  Sixteen values per thread (four 16-byte loads issued together): a
// 100 k-query batch is seven steps, ~15 us (round 3, before: 256 threads x 4 values, 98 steps, 70 us per scan, 16 scans per step).
#define SS_THREADS 1024
#define SS_ITEMS 16
__global__ __launch_bounds__(SS_THREADS) void k_scan_small(const u32* __restrict__ in, u32* __restrict__ out, size_t n, u32* __restrict__ total,
                                                           int inclusive) {
    __shared__ u32 s_w[SS_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 carry = 0;
    for (size_t i0 = 0; i0 < n; i0 += (size_t)SS_THREADS * SS_ITEMS) {
        const size_t i = i0 + (size_t)threadIdx.x * SS_ITEMS;
        u32 v[SS_ITEMS];
        if (i + SS_ITEMS <= n) {
#pragma unroll
            for (int k = 0; k < SS_ITEMS; k += 4) {
                const uint4 x = ld4(in + i + k);
                v[k] = x.x, v[k + 1] = x.y, v[k + 2] = x.z, v[k + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SS_ITEMS; ++k) v[k] = i + k < n ? in[i + k] : 0u;
        }
        u32 s = 0;
#pragma unroll
        for (int k = 0; k < SS_ITEMS; ++k) s += v[k];
        const u32 inc = wave_incl_scan_u32(s, lane);
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        u32 base = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < SS_THREADS / 64; ++k) {
            const u32 x = s_w[k];
            base += k < w ? x : 0u;
            tot += x;
        }
        __syncthreads();
        u32 run = carry + base + inc - s;   // exclusive prefix of this thread's first value
#pragma unroll
        for (int k = 0; k < SS_ITEMS; ++k) {
            const u32 x = v[k];
            v[k] = inclusive ? run + x : run;
            run += x;
        }
        if (i + SS_ITEMS <= n) {
#pragma unroll
            for (int k = 0; k < SS_ITEMS; k += 4) st4(out + i + k, make_uint4(v[k], v[k + 1], v[k + 2], v[k + 3]));
        } else {
#pragma unroll
            for (int k = 0; k < SS_ITEMS; ++k)
                if (i + k < n) out[i + k] = v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
#define SCAN_SMALL_TILES 4   // up to 32 k values in ONE launch (config 2's 10 k queries: 14.2 against 14.5 ms); above that the three kernels are faster -- a 100 k-value scan
                            // takes 26 us in one workgroup and 5 + 5 + 7 in three launches, and a config-3 step holds 21 of them (46.5 -> 46.15 ms)

size_t scan_u32_temp_elems(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 2; }

// out[i] = sum(in[0..i)) (exclusive) or sum(in[0..i]) (inclusive); in may alias out; 4-byte alignment suffices.
// temp needs scan_u32_temp_elems(n) u32; temp[nblocks] receives the grand total (device side).
const u32* scan_u32(const u32* in, u32* out, size_t n, bool inclusive, u32* temp, hipStream_t st) {
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 0) {
        HIP_CHECK(hipMemsetAsync(temp, 0, sizeof(u32), st));
        return temp;
    }
    if (nb <= SCAN_SMALL_TILES) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SS_THREADS), 0, st, in, out, n, temp + nb, inclusive ? 1 : 0);
        return temp + nb;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, temp);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_THREADS), 0, st, temp, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, temp, inclusive ? 1 : 0);
    return temp + nb;
}

// ---- tiny helpers ------------------------------------------------------------------------------
__global__ void k_fill_u32(u32* p, size_t n, u32 v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void fill_u32(u32* p, size_t n, u32 v, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}

// first hit ordinal of every query of the batch (position-major seed ordinals): the segments of the per-query sorts and the tiles of the
// bucketed passes.  (Lives here, not next to the segmented sorts: launching it must not load their large code object.)
__global__ __launch_bounds__(256) void k_query_segments(const u32* __restrict__ hoff, const u32* __restrict__ qoff, u32 qa, u32 qb, int AS,
                                                        u32 H, u32* __restrict__ seg /*entries [qa, qb] written*/) {
    const u32 q = qa + blockIdx.x * 256u + threadIdx.x;
    if (q > qb) return;
    const size_t t = ((size_t)qoff[q] + q) * (size_t)AS;  // first seed slot of query q (position-major ordinals)
    seg[q] = q < qb ? hoff[t] : H;                        // (hoff: the pass's exclusive scan, valid on the pass's slots)
}

void launch_query_segments(const u32* hoff, const u32* qoff, u32 qa, u32 qb, int AS, u32 H, u32* seg, hipStream_t st) {
    hipLaunchKernelGGL(k_query_segments, dim3((qb - qa + 1 + 255) / 256), dim3(256), 0, st, hoff, qoff, qa, qb, AS, H, seg);
}

