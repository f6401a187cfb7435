// common.h -- shared declarations of libsohit.so (host orchestration + HIP kernels, gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "tune.h"

typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

struct SoError : std::runtime_error {
    explicit SoError(const std::string& m) : std::runtime_error(m) {}
};
// a query batch's 32-bit candidate store is full: the caller halves the batch
struct CandOverflow : SoError {
    CandOverflow() : SoError("a query batch collected >= 2^32 candidates") {}
};
// a device allocation failed: inside a query batch the caller frees the batch's buffers and halves the batch
struct DevOom : SoError {
    explicit DevOom(size_t bytes) : SoError("HIP error out of memory: device allocation of " + std::to_string(bytes) + " bytes failed") {}
};

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw SoError(std::string("HIP error ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" +   \
                          std::to_string(__LINE__) + " in " #expr);                                       \
    } while (0)

extern thread_local int g_poison;   // Tune::poison of the context whose call runs on this thread (-1: none)

// Grow-only device buffer.
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    void ensure(size_t n, bool keep = false, hipStream_t st = 0) {
        if (n <= cap) return;
        size_t nc = n + n / 8 + 64;
        T* np_ = nullptr;
        {
            const hipError_t e = hipMalloc((void**)&np_, nc * sizeof(T));
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                throw DevOom(nc * sizeof(T));
            }
            HIP_CHECK(e);
        }
        // SOHIT_POISON=<byte>: fill every fresh allocation (tests: results must not depend on what device memory held before)
        // (the fill is queued on the null stream, which the non-blocking streams -- index build, side streams -- are not ordered with: wait for it)
        if (g_poison >= 0) {
            HIP_CHECK(hipMemset(np_, g_poison & 0xFF, nc * sizeof(T)));
            HIP_CHECK(hipStreamSynchronize(0));
        }
        if (keep && p && cap) {
            HIP_CHECK(hipMemcpyAsync(np_, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        if (p) (void)hipFree(p);
        p = np_;
        cap = nc;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// ------------------------------------------------------------------------------------------------
// Residue classes
//   hash class (5 bit, packed):  0..29 = distinct values of code[byte] present in the run's alphabet
//                                 map, 30 = sequence separator, 31 = 'x'/'X' (window rejected);
//                                 both 30 and 31 invalidate a seed window (fsearch.py:538-540).
//   score class (8 bit array):    0..22 = the 23 BLOSUM62 letters (either case), 23 = anything else (-4)
// ------------------------------------------------------------------------------------------------
#define HCLS_SEP 30
#define HCLS_X 31
#define SCLS_N 24
#define MAX_PATTERNS 8   // seed patterns per run (S)
#define MAX_ALPHA 4      // alphabets per run (A)
#define MAX_SEEDLEN 32   // longest spaced-seed pattern
#define MIN_UNGAP 25     // self.min, fsearch.py:2224
#define DROPX 30
#define UG_SHARDS 1    // pass-list regions of k_ungap (power of two).  One region: a wave appends ~64 records per atomic, a few
                       // hundred thousand appends per launch -- far below the same-address atomic rate -- and the list needs no compaction

struct SeedCfg {
    int S, A;                       // #patterns, #alphabets
    int klen[MAX_PATTERNS];         // pattern lengths
    u32 care[MAX_PATTERNS];         // bit j set = position j is a '1' (care) position
    int mink;                       // shortest pattern length (fsearch.py:2219)
    u32 lut[MAX_ALPHA][32];         // hash class -> code[ch] value per alphabet (fsearch.py:543-547)
    u32 nc;                         // bucket count
};

// Sort-key layout of one seed hit: [q | subj | diag | qpos | as | tag] from MSB to LSB.  With the compact (banded) index addends
// `subj` counts diagonal bands (k_encode_band32) and diag_off is the offset of a one-band subject's diagonals.
struct KeyLayout {
    int bq, bs, bd, bp, ba;         // bit widths
    int sh_tag, sh_as, sh_qpos, sh_diag, sh_subj, sh_q;
    int total;
    i64 diag_off;                   // added to (qpos - sst) to make it non-negative
    void finish() {
        sh_tag = 0;
        sh_as = ba;
        sh_qpos = 2 * ba;
        sh_diag = sh_qpos + bp;
        sh_subj = sh_diag + bd;
        sh_q = sh_subj + bs;
        total = sh_q + bq;
    }
};

// Bucketed diagonal binning (k_bucket.hip): a bucket = (query of the pass, range of 2^wb diagonal bands); a hit inside a
// bucket is one 32-bit word  band_low << (bd + bp) | diagonal << bp | qpos.  (Bands: KeyLayout's "subject" field counts diagonal
// bands of 2^bd ids -- one per chunk sequence, several for a sequence too long for one; k_index.hip, k_encode_band32.)  The pass
// records of the best-diagonal reduction are binned by (query, range of 2^wb chunk SEQUENCES) with a layout of their own.
#define BKT_RMAX 512          // ranges per chunk (per-wave LDS histogram of the count / scatter passes)
struct BktLayout {
    int wb, bd, bp;           // bits: band (or sequence) inside the range, diagonal, query position
    int sh_q, sh_qpos;        // where q / qpos sit in the seeds' 64-bit key bases (KeyLayout)
    u32 nqp, qa;              // queries in the pass, first one (batch-local index)
    u32 R;                    // ranges: bucket id = range * nqp + (q - qa)
    u32 maxslen;              // (unused by the kernels)
};

#ifdef __HIPCC__
// First-touch key of a group whose only / first hit has key k0 (one alphabet x one pattern: the hits of a group are ordered by query
// position, so the minimum over the group is the head's): (as, qpos, ~j, ~tag, ~pos) -- emission order, then index slot order ==
// descending (j, tag, pos).  An entry that sits at the end of subject j is the offset-0 entry of chunk sequence j + 1 (see k_lookup).
__device__ __forceinline__ u64 ft_key_of_head(u64 k0, const KeyLayout& kl, int ft_bits_entry, int bsp, const u32* __restrict__ roff) {
    const u64 pmask = (1ull << kl.bp) - 1ull, amask = (1ull << kl.ba) - 1ull;
    const u64 jmax = (1ull << (kl.bs + 1)) - 1ull, pmax = (1ull << bsp) - 1ull;
    const u32 gsubj = (u32)((k0 >> kl.sh_subj) & ((1ull << kl.bs) - 1ull));
    const i64 gdiag = (i64)((k0 >> kl.sh_diag) & ((1ull << kl.bd) - 1ull)) - kl.diag_off;  // qpos - sst
    const int sl = (int)(roff[gsubj + 1] - roff[gsubj]);
    const int qpos = (int)((k0 >> kl.sh_qpos) & pmask);
    const u32 as = kl.ba ? (u32)((k0 >> kl.sh_as) & amask) : 0u;
    const u32 tag = kl.ba ? (u32)(k0 & amask) : 0u;
    const int sst = (int)((i64)qpos - gdiag);
    u32 j = gsubj, pos = (u32)sst;
    if (sst == sl) j = gsubj + 1, pos = 0;
    const u64 inv = ((jmax - j) << (kl.ba + bsp)) | ((amask - tag) << bsp) | (pmax - pos);
    return ((((u64)as << kl.bp) | (u64)qpos) << ft_bits_entry) | inv;
}
#endif

// One alignment task / result (phase 2).
struct AlnTask {
    u32 q;        // query index local to the batch
    u32 subj;     // global subject id
    u32 qi, qj;   // start offsets (fsearch.py:3063, 3069-3070); tile starts for the long path
    u32 score;    // ungapped candidate score
    u32 rank;     // position in the query's sorted candidate list
    u32 qe, se;   // exclusive ends of the aligned windows: sequence lengths, or tile ends (kswat_st_long, 1487-1490)
};
#define LONG_SEQ 4096  // fsearch.py:3068 / kswat_st_long chk

struct AlnRes {
    int maxscore, aln, matches, gap;
    int qst, qed, sst, sed;  // 0-based start (exclusive) / end as kswat_st returns them
    int cells;
    int pad;
};

static inline int ceil_log2(u64 x) {  // bits needed to hold values in [0, x)
    int b = 0;
    while ((1ull << b) < x) ++b;
    return b < 1 ? 1 : b;
}
