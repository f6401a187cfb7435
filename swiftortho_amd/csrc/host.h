// host.h -- what the host_*.hip files of libsohit.so share: the context, the resident sequence sets and chunk indexes, the per-batch
// scratch, the small timing / result-buffer helpers and the entry points of each stage.  Internal (not part of include/sohit.h).
//
// Product code.  Nothing here may call into oracle/ (the CPU restatement is test infrastructure);
// there is no CPU fallback for any device stage.  Host-side work is limited to what the reference
// also does outside its hot loops: FASTA indexing (fsearch.py:1543-1553, 2182-2199), SEG-like
// query masking (2872-2928; table-driven, bit-identical libm logs), the per-chunk mu+2sd threshold
// from exact device-side integer sums (746-761, 2248-2250), and text formatting (43-61, 3234-3243).
//   host_load.hip    parameters, constant tables, FASTA sets resident in HBM, SEG masking on the host
//   host_index.hip   per-chunk index build, Fasta.load of the reference's index files, banded diagonal ids
//   host_seed.hip    batch preparation, the seed stage (bounds, cap, lookup + binning, ungapped extension, best diagonal, candidate order)
//   host_phase2.hip  banded alignments in rounds, stop rule, traces, row emission
//   host_search.hip  the batched search, result-array cache, work pre-pass
//   host_abi.hip     row formatting and the C ABI of include/sohit.h
#pragma once
#include "common.h"
#include "kernels.h"
#include "seedhash.h"
#include "../../include/sohit.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <malloc.h>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <sys/stat.h>

inline double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// libm entry points reached through volatile pointers so that no compiler folds pow(x, 2) / log
extern double (*volatile p_pow)(double, double);
extern double (*volatile p_log)(double);
extern double (*volatile p_log10)(double);
extern double (*volatile p_sqrt)(double);


// ---------------------------------------------------------------------------------------------
// FASTA set: host index (fsearch.py:1543-1553) + device-resident arrays
// ---------------------------------------------------------------------------------------------
struct SeqSet {
    std::string data;            // raw file bytes
    std::vector<i64> rec;        // record offsets (idx)
    std::vector<u32> hd_beg, hd_len;  // header span of each record (without '>')
    std::vector<u32> id_len;          // ... and of its first blank-free word (the id columns of the output)
    i64 N = 0;
    std::vector<u32> off;        // [N+1] residue offsets
    std::vector<u8> res;         // concatenated residues (raw bytes)
    u32 maxlen = 0;
    // device
    DevBuf<u8> d_res, d_scls_store, d_scls4_store, d_pcls_store, d_pcls4_store;
    struct { u8* p = nullptr; } d_pcls, d_pcls4;   // classes / classes * 4 with PCLS_PAD sentinels behind every sequence (k_align_pk)
    struct { u8* p = nullptr; } d_scls;   // score classes; 16 readable bytes in front (k_ungap's left-pass windows start up to 8 bytes early)
    struct { u8* p = nullptr; } d_scls4;  // score class * 4 (k_ungap's subject side: column offset in its LDS table), same padding
    DevBuf<u32> d_off, d_words, d_pseq;
    DevBuf<u8> d_ug_store;       // k_ungap1's subject side (class * 8 with sentinels), made on first use
    bool ug_valid = false;
    DevBuf<u32> d_bound;         // per sequence: upper bound of any alignment score it can take part in (k_seq_bound)
    u32 P = 0, Ppad = 0;
    HashLut lut;

    void parse() {
        rec.clear();
        rec.push_back(0);
        const i64 n = (i64)data.size();
        for (const char *d0 = data.data(), *q = n > 1 ? (const char*)memchr(d0 + 1, '>', (size_t)n - 1) : nullptr; q;
             q = q + 1 < d0 + n ? (const char*)memchr(q + 1, '>', (size_t)(d0 + n - q - 1)) : nullptr)
            if (q[-1] == '\n') rec.push_back((i64)(q - d0));
        N = (i64)rec.size();
        off.assign((size_t)N + 1, 0);
        hd_beg.resize((size_t)N);
        hd_len.resize((size_t)N);
        id_len.resize((size_t)N);
        res.clear();
        res.reserve(data.size());
        maxlen = 0;
        for (i64 x = 0; x < N; ++x) {
            i64 st = rec[x], ed = (x == N - 1) ? n : rec[x + 1];
            i64 p = st;
            {
                const void* nl = memchr(data.data() + st, '\n', (size_t)(ed - st));
                p = nl ? (i64)((const char*)nl - data.data()) : ed;
            }
            hd_beg[x] = (u32)std::min<i64>(st + 1, p);
            hd_len[x] = (u32)(p > st ? p - st - 1 : 0);
            {
                const void* sp = hd_len[x] ? memchr(data.data() + hd_beg[x], ' ', hd_len[x]) : nullptr;
                id_len[x] = sp ? (u32)((const char*)sp - (data.data() + hd_beg[x])) : hd_len[x];
            }
            ++p;
            while (p < ed) {
                const void* nl = memchr(data.data() + p, '\n', (size_t)(ed - p));
                const i64 q = nl ? (i64)((const char*)nl - data.data()) : ed;
                res.insert(res.end(), data.begin() + p, data.begin() + q);
                p = q + 1;
            }
            if (res.size() >= 0xFFFFFF00ull) throw SoError("sequence set exceeds 4 Gi residues");
            off[x + 1] = (u32)res.size();
            maxlen = std::max(maxlen, off[x + 1] - off[x]);
        }
    }
    std::string header(i64 x) const { return data.substr(hd_beg[x], hd_len[x]); }
    std::string ident(i64 x) const {
        std::string h = header(x);
        return h.substr(0, h.find(' '));
    }
    u32 len(i64 x) const { return off[x + 1] - off[x]; }
};


struct ChunkIndex {
    u64 s2 = 0;  // sum of squared bucket sizes (seed hits a reference-like query window expects: s2 / E)
    i64 seq_lo = 0, seq_hi = 0;
    u32 p_lo = 0, p_hi = 0;
    u32 E = 0;
    i64 threshold = 0;
    u32 maxslen = 0;
    DevBuf<u64> entries;  // E, grouped by ascending bucket id (the reference's CSR slot layout)
    DevBuf<u32> ub, ubeg, ucnt;  // occupied bucket ids (ascending), their first slots (+ E), their sizes
    u32 U = 0;            // occupied buckets
    DevBuf<u32> hkey;     // open-addressed map bucket id -> hval = first slot | count << 32
    DevBuf<u64> hval;
    int hshift = 31;
    u32 hmask = 0;
    DevBuf<u64> dir;      // bitmap + rank directory (k_dir_build) used instead of the map when NC <= 2^28
    bool use_dir = false;
    DevBuf<u64> dkeys;    // E: per-entry key addends for the layout (d_sh_subj, d_sh_diag) -- k_encode_delta
    int d_sh_subj = -1, d_sh_diag = -1;
    // compact (4-byte) addends, one set per key layout in use -- k_encode_band32.  A layout = (tag bits, query-position bits, diagonal
    // bits k): the chunk's (subject, diagonal) pairs are numbered in bands of 2^k ids, one band for a sequence of length <= C =
    // 2^k - 2^bp, several for a longer one.  A search with queries of several length classes alternates between a few layouts per
    // chunk, so the sets are kept (at most ten, least recently used first out).
    struct BandEnc {
        int ba = -1, bp = -1, k = -1;
        bool multi_ok = false;   // built with several bands per long subject allowed (one alphabet x one pattern only)
        bool multi = false;      // ... and some subject does own several: btab resolves bands
        u32 nband = 0;           // bands of the chunk (== sequences unless multi)
        u32 C = 0;               // diagonal offset of the one-band subjects
        DevBuf<u32> dk32, gbase; // E addends; per chunk sequence: (first band << k) + (C or, for a multi-band subject, its length)
        DevBuf<u64> btab;        // nband x (chunk sequence | gbase << 32)
        std::vector<std::pair<u32, u32>> spans;   // (first band, bands) of the subjects that own several (host copy: range_table checks them)
        u64 used = 0;
        // the index was rebuilt (so_drop_index + so_build_index) over the same sequences: layout, gbase and btab -- functions of the sequence
        // lengths alone -- still hold, only the entries' addends are encoded again
        bool stale = false;
        u64 ref_gen = 0;
        i64 seq_lo = -1, seq_hi = -1;
    };
    std::vector<std::unique_ptr<BandEnc>> encs;
    u64 enc_clock = 0;
    // The count pass of the bucketed binning without the index entries (k_bkt_count_tab): every bucket's members in descending entry
    // order (order_chunk: the reference's own CSR order; lazily, at the chunk's first dense pass -- a sparse search never pays for it) and,
    // per (key layout, range width), the boundaries of the band ranges inside every occupied bucket.
    bool ordered = false;
    DevBuf<u32> row_of_slot;    // E: occupied-bucket ordinal at every bucket's first slot
    struct RangeTab {
        int ba = -1, bp = -1, k = -1, wb = -1;   // the band encoding it belongs to, the range width
        bool multi_ok = false;
        u32 R = 0;
        bool ok = false;        // false: a multi-band subject straddles a range boundary, a bucket above 65535 entries, ... -- the counting pass stays
        DevBuf<u16> tab;        // U x (R + 1)
        u64 used = 0;
    };
    std::vector<std::unique_ptr<RangeTab>> rtabs;
};


struct so_ctx {
    int device = 0;
    u32 ncu = 256;          // compute units of the device
    Tune tune;
    hipStream_t st = nullptr;
    // params
    std::string seeds, alphabet;
    i64 nc = 0, chunk = 50000, step = 1, v = 500, thr = -1;
    double expect = 1e-3, max_miss = 1e-3;
    bool filter = true, profile = false;
    SeedCfg cfg;
    std::vector<std::array<int, 256>> codes;
    std::string err;
    // constant device tables
    DevBuf<u8> d_smap, d_hmap;
    DevBuf<signed char> d_b62c;
    DevBuf<int> d_bittab;
    u8 smap[256];
    signed char b62c[SCLS_N * SCLS_N];
    static const int BITTAB_N = 1 << 16;
    // sets
    SeqSet ref, qry;
    std::string ref_path;   // file the reference was read from ("" when it came from memory) and its size / mtime then
    long long ref_fsize = -1, ref_mtime_ns = -1;
    bool ref_loaded = false, qry_loaded = false, index_built = false;
    u64 ref_gen = 0;        // bumped by every reference load
    // band_plan()'s answers: a function of a chunk's sequence lengths only, so they outlive index rebuilds (cleared with the reference)
    struct BandPlan { i64 lo, hi; int bp; bool multi_ok; int k; u64 nband; };
    std::vector<BandPlan> band_plans;
    i64 r_lo = -1, r_hi = -1;
    std::vector<std::unique_ptr<ChunkIndex>> chunks;
    std::vector<std::unique_ptr<ChunkIndex>> spare_chunks;  // dropped chunk objects: their device buffers are reused by the next build
    // masked query cache of the last batch / search (for so_masked_query)
    std::vector<std::string> masked;     // indexed by qidx - masked_lo
    i64 masked_lo = 0;
    std::vector<std::vector<u32>> last_cands;  // per query of last search: 4 x u32 per cand
    u64 qry_gen = 0;                           // bumped by every query load: what a batch's cached slot layout is good for
    i64 last_q_lo = 0;
    so_counters cnt;
    // scratch
    DevBuf<u32> d_scan_tmp, d_tmp32a, d_tmp32b;
    DevBuf<u64> d_stats;
    DevBuf<u32> d_small;  // parked scan totals (stash_u32)
    DevBuf<u32> ix_bkt, ix_bkt2, ix_flags, ix_ridx, ix_plan, ix_tk;  // index build scratch
    DevBuf<u64> ix_ent, ix_tv;
    DevBuf<u8> d_pcls;
    DevBuf<u8> d_sort_tmp;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t st_rows = nullptr;  // result rows leave on their own stream: the download of one batch overlaps the next batch's kernels
    hipEvent_t ev_rows = nullptr, ev_rows_done = nullptr;
    hipEvent_t ev_part[8] = {nullptr};   // emission range p's rows have arrived (the worker converts a range while the next is traced)
    // the k-mer order of queries too long for the LDS sort (one wave each, milliseconds for a 30 000-residue protein) runs beside the
    // batch's other preparation and the seed passes of the shorter length classes
    hipStream_t st_side = nullptr;
    hipEvent_t ev_side_go = nullptr, ev_korder = nullptr;
    hipStream_t st_ug = nullptr;       // k_ungap beside k_ungap1 (Tune::ug1_overlap)
    hipEvent_t ev_ug_go = nullptr, ev_ug_done = nullptr;
    hipEvent_t ev_bnd_go = nullptr, ev_bnd_done = nullptr;   // the next chunk's k_bounds on st_side (chunk_qhits)
    bool rows_in_flight = false;
    bool count_tab_now = false;   // this (batch, chunk)'s bucketed passes count from the range boundaries (seed_stage decides: SOHIT_COUNT_TAB)
    size_t max_hits_per_pass = (size_t)1 << 30;
    u32 max_batch = 131072;  // queries per device batch.  Round 3, config 3 (100k queries), same box: 25000 64.1 ms, 33334 63.0, 2 x 50000 63.0,
                             // 65536 + 34464 62.1, one batch of 100000 59.1 -- larger launches beat overlapping one batch's row download with
                             // the next batch's kernels (round 2, when the host side was slower: 65536 109 ms, 131072 115)
    // device SEG: tables, symbol folding of the loaded query set
    DevBuf<u8> d_segtab, d_symmap, d_upmap, d_segmask;
    bool seg_on_device = false;
    bool q_present[256];               // bytes that can occur in (masked) query residues
    void* pinned = nullptr;            // pinned host staging for result rows
    size_t pinned_cap = 0;
    // host-side row emission of batch k runs on a worker thread while the GPU processes batch k + 1
    struct EmitJob {
        std::thread th;
        bool active = false;
        size_t base = 0, n = 0;
        std::atomic<i64> dropped{0};
        std::exception_ptr err;
    } emit;
    unsigned long long* h_qhits = nullptr;  // pinned: per-query hit counts of one (batch, chunk)
    size_t h_qhits_cap = 0;
    unsigned long long* h_small = nullptr;  // pinned scratch for the small device -> host reads (counts, totals): 1 KB
    std::map<std::string, double> tm;  // per-stage wall ms (only with params.profile)
    std::map<std::string, double> lt;  // wall ms of the last loads, always kept: load.ref_parse / load.ref_h2d_layout / load.qry_parse / load.qry_h2d (reported beside the step)
    std::shared_ptr<void> batch;       // persistent per-batch scratch (struct Batch)
    // device-resident results (so_search_device): so_hit records stay in HBM until the caller has exchanged them
    // the library sorts' code objects (rocPRIM: megabytes each) are loaded by their first launch: a thread started by so_create does two
    // tiny sorts while the caller reads and parses its FASTA files; joined before the first index build
    std::thread warm;
    bool dev_out = false;
    DevBuf<u8> d_hits;
    size_t d_hits_n = 0;
    DevBuf<double> d_p2tab;
};

static inline u64 cand_limit() { return tune().cand_limit > 0 ? (u64)tune().cand_limit : 0xFFFFFFF0ull; }


// ---------------------------------------------------------------------------------------------
// SEG-like masking (fsearch.py:2872-2928; entropy 2854-2868; Counter 157-177).  Only output[:n]
// is used downstream (2996, 3034).  Logs come from a table of libm values so the arithmetic is
// bit-identical to evaluating log() in place.
// ---------------------------------------------------------------------------------------------
struct SegTables {
    double lg12[64];      // log(k / 12.)
    double lgn[13][32];   // log(j / n), n = first-window length (1..12)
    double log2v;
    SegTables() {
        log2v = p_log(2);
        for (int k = 1; k < 64; ++k) lg12[k] = p_log((double)k / 12.);
        lg12[0] = 0;
        for (int n = 1; n <= 12; ++n)
            for (int j = 1; j < 32; ++j) lgn[n][j] = p_log((double)j / ((double)n * 1.));
    }
};


template <class F>
void parallel_for(i64 n, F f) {
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < 200000 || nt == 1) {  // thread start-up costs ~0.3 ms: only worth it for big loops
        for (i64 i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<i64> next(0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (;;) {
                i64 b = next.fetch_add(4096);
                if (b >= n) break;
                for (i64 i = b; i < std::min(n, b + 4096); ++i) f(i);
            }
        });
    for (auto& t : th) t.join();
}


// ---------------------------------------------------------------------------------------------
// Search
// ---------------------------------------------------------------------------------------------
#define QCLASSES_MAX 5
struct Batch {
    i64 q_lo = 0, q_hi = 0;  // absolute query ordinals
    u32 nq = 0;
    std::vector<u8> h_res;   // masked residues
    std::vector<u32> h_off;  // [nq+1]
    u32 maxqlen = 0;
    // Queries are held in LENGTH-CLASS order inside a batch (stable inside a class): batch slot i is query q_lo + qid[i] of the file.
    // Per-query results do not depend on their neighbours (find_hit.py:107-146 relies on the same fact), seed passes never mix
    // classes -- so key widths follow the pass's longest query, not the batch's -- and the rows are put back in file order when they
    // reach the host.  One class only (the usual protein set below 1024 residues): qid is the identity and `permuted` is false.
    std::vector<u32> qid;
    std::vector<u8> qcls;    // length class per slot (query_class)
    // The slot layout (qid, qcls, h_off, maxqlen and their device copies d_qid, dev.d_off) is a function of the loaded queries and the
    // range alone: a second search over the same range finds it made (0.5-1 ms of host loops and two pageable uploads in front of the
    // batch's first kernel, with the GPU idle).
    u64 lay_gen = 0;
    i64 lay_lo = -1, lay_hi = -1;
    bool lay_classes = false;
    bool permuted = false;
    // functions of the slot layout, made with it (the host loops over a 100 k-query batch that found them again for every search and every
    // chunk were ~0.1 ms of GPU idle per chunk): are the classes in order; first slot of every class (+ nq); longest query of every class and
    // of every block of 256 slots
    bool cls_sorted = true;
    u32 cls_start[QCLASSES_MAX + 1] = {0};
    u32 cls_maxq[QCLASSES_MAX] = {0};
    std::vector<u32> blkmax;
    // slots [q_defer, nq): the length class whose longest members' k-mer order is still being computed on the side stream; their
    // frequency cap (and everything after it) waits for ev_korder, the classes before them do not
    u32 q_defer = 0;
    bool korder_async = false;
    // k-mer orders are computed when a query reaches its frequency cap (chunk_qhits): its own while such queries are few, every query's
    // at the first chunk where they are the majority; ksc_long = first batch slot whose order needs global scratch
    bool korder_ready = false;   // every query's order is there
    u32 ksc_long = 0;
    DevBuf<u32> open_list;       // this chunk's queries above their cap while !korder_ready (k_cap_all), n_open of them
    u32 n_open = 0;
    DevBuf<u8> kord_have;        // per batch slot: its order has been computed (order_open_queries)
    bool kord_have_clear = false;
    DevBuf<u32> d_qid, d_ocnt, d_ostart;
    SeqSet dev;              // device arrays only (d_res = masked raw, d_scls, d_off, d_words, d_pseq)
    DevBuf<u32> qbucket, korder, sbeg, scnt, pcnt, eff, nz, hoff, cidx;
    // k_bounds of the NEXT chunk runs on the side stream beside this chunk's seed stage (one random directory line per window: memory
    // latency, no LDS), into a second set of arrays that chunk_qhits swaps in (SOHIT_BOUNDS_AHEAD)
    DevBuf<u32> sbeg2, scnt2, pcnt2;
    int bnd_ci = -1;         // chunk whose bounds the second set holds (or is being filled with): -1 none
    DevBuf<int> ksc;
    DevBuf<u8> mark;
    DevBuf<u32> cs_hoff, cs_beg, blk_first, qseg, sel_idx;
    DevBuf<unsigned long long> qhits;
    DevBuf<u64> cs_kbase;
    DevBuf<u64> keys, keys2;
    DevBuf<u32> hits32, hits32s, bmat, bpart, bt0, btd, bext, bflag, bcnt, bccnt;  // bucketed binning (k_bucket.hip)
    DevBuf<u64> mlist;   // heads of the groups of two and more hits of a bucketed pass (k_ungap1 -> k_ungap2)
    DevBuf<u32> flags, gidx, ghead;
    DevBuf<u64> p_qs, p_sd, p_ft, p_qs2, tmp64, q_qs, q_sd, q_ft;
    DevBuf<u32> shard;
    DevBuf<unsigned long long> stepshard;
    DevBuf<u32> pidx, pidx2, shead;
    DevBuf<u64> c_ft, c_ft2;
    DevBuf<u32> c_q, c_rec, order, order2;
    DevBuf<u32> counters;  // [0] pass count, [1] hvalid
    DevBuf<unsigned long long> ucount;  // [0] ungap steps, [1] cells
    // candidate store
    DevBuf<u32> cand_q, cand_rec;             // all chunks' regions concatenated
    std::vector<u32> chunk_base;              // region start per chunk (+ total)
    DevBuf<u32> ccnt;                         // [nchunks][nq] per-query counts
    DevBuf<unsigned long long> qcells;        // DP cells per query (phase2)
    DevBuf<u32> tpos, spcnt, spoff, sidx, spec_trace, sel_a, sel_b;   // speculative traces of the first aligner round (phase2)
    DevBuf<u64> gx;
    DevBuf<u32> gL, gR;
    DevBuf<u32> segfirst, st_state, rcnt, tcnt, roff, ridx, ridx2, ntile, roffc, rk_slot, order_tmp;
    DevBuf<u32> cqoff, prior, qtot, qcoff, fin_rec, perm, ntask, toff, sel, nout, ooff;
    DevBuf<AlnTask> tasks;
    DevBuf<AlnRes> ares;
    DevBuf<int> bits, outrec;
    DevBuf<u32> trace;
    DevBuf<u32> tr_units, tr_ofs;   // trace room per task of a launch list and its exclusive scan (k_trace_units)
    DevBuf<u32> tl_sorted, al_sorted;   // the trace pass's lists ordered by band rows (mixed-length batches)
};

// length classes of the queries: < 512 residues, < 1024, < 2048, < 4096, longer (the aligner's tiled path).  A pass's key widths and
// bucket ranges follow its longest query, and the seed hits a query brings to a bucket grow with its length: inside a class
// they differ by a factor of two (eight in the first), so the grouping kernel's buckets stay near their target size.
#define QCLASSES 5
static_assert(QCLASSES <= QCLASSES_MAX, "Batch::cls_start");
inline u8 query_class(u32 len) { return len < 512 ? 0 : len < 1024 ? 1 : len < 2048 ? 2 : len < 4096 ? 3 : 4; }

// wall-clock stage laps (stream-synchronising, so only when profiling)
struct StageClock {
    so_ctx* c;
    double t = 0;
    explicit StageClock(so_ctx* c_) : c(c_) {
        if (c->profile) {
            (void)hipStreamSynchronize(c->st);
            t = wall();
        }
    }
    void lap(const char* name) {
        if (!c->profile) return;
        (void)hipStreamSynchronize(c->st);
        double n = wall();
        c->tm[name] += (n - t) * 1e3;
        t = n;
    }
};

struct ProfTimer {
    so_ctx* c;
    double* ms;
    int64_t* launches;
    bool on;
    ProfTimer(so_ctx* c_, double* ms_, int64_t* l_) : c(c_), ms(ms_), launches(l_), on(c_->profile) {
        if (on) HIP_CHECK(hipEventRecord(c->ev0, c->st));
    }
    void stop() {
        if (!on) return;
        HIP_CHECK(hipEventRecord(c->ev1, c->st));
        HIP_CHECK(hipEventSynchronize(c->ev1));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, c->ev0, c->ev1));
        *ms += t;
        *launches += 1;
        on = false;
    }
};

// One released result array is kept for the next search (process-wide, SOHIT_HIT_CACHE=0 turns it off): a config-3 result is 130 MB,
// and handing that back to the kernel page by page and faulting it in again costs more than 17 ms per search -- a fifth of the step.
// Only arrays between 1 MiB and 2 GiB are kept; the larger of (cached, released) survives, so_destroy() drops it.
struct HitCache {
    std::mutex mu;
    so_hit* p = nullptr;
    size_t bytes = 0;
    static bool enabled() {
        return tune().hit_cache;
    }
    so_hit* take(size_t& cap_rows) {
        std::lock_guard<std::mutex> g(mu);
        so_hit* r = p;
        cap_rows = bytes / sizeof(so_hit);
        p = nullptr, bytes = 0;
        return r;
    }
    // returning a large array to the system (munmap of 130 MB: ~18 ms) is not the caller's business: a detached thread does it
    static void release(so_hit* q) {
        if (!q) return;
        if (malloc_usable_size(q) < ((size_t)8 << 20)) {
            free(q);
            return;
        }
        try {
            std::thread([q] { free(q); }).detach();
        } catch (...) {
            free(q);
        }
    }
    void give(so_hit* q) {
        if (!q) return;
        const size_t b = malloc_usable_size(q);
        if (!enabled() || b < ((size_t)1 << 20) || b > ((size_t)2 << 30)) {  // a 1 M-protein result (24 GB) is not worth holding on to
            release(q);
            return;
        }
        so_hit* drop = q;
        {
            std::lock_guard<std::mutex> g(mu);
            if (b > bytes) drop = p, p = q, bytes = b;
        }
        release(drop);
    }
    void clear() {
        size_t n;
        free(take(n));
    }
};
extern HitCache g_hit_cache;


// growable result array handed to the caller as-is: no zero-fill, no final copy
struct HitBuf {
    so_hit* p = nullptr;
    size_t n = 0, cap = 0;
    void grow(size_t extra) {
        if (n + extra <= cap) return;
        if (!p) p = g_hit_cache.take(cap);  // the previous search's array, pages still mapped
        if (n + extra <= cap) return;
        size_t nc = std::max<size_t>(n + extra, cap + cap / 2 + 1024);
        so_hit* np_ = (so_hit*)realloc(p, nc * sizeof(so_hit));
        if (!np_) throw SoError("out of host memory for the result rows");
        p = np_;
        cap = nc;
    }
    so_hit* release() {
        so_hit* r = p ? p : (so_hit*)malloc(sizeof(so_hit));
        p = nullptr;
        n = cap = 0;
        return r;
    }
    ~HitBuf() { g_hit_cache.give(p); }
};

struct HostRow {
    int v[12];
};

std::vector<std::string> split(const std::string& s, char sep);
void build_score_maps(u8 smap[256], signed char b62c[SCLS_N * SCLS_N]);
void nr_table(const std::string& gaa, int tbl[256]);
void byte_presence(const u8* bytes, size_t n, bool present[256]);
void build_hash_classes(const bool present[256], const std::vector<std::array<int, 256>>& codes, u8 hmap[256], HashLut& lut);
void set_tune(const Tune* t);
// host_load.hip: parameters, constant tables, sequence sets
void set_params(so_ctx* c, const so_params* p);
void upload_constants(so_ctx* c);
void layout_set(so_ctx* c, SeqSet& s, const bool present[256], size_t nres, u32 nseq);
void upload_set(so_ctx* c, SeqSet& s, const u8* residues, const std::vector<u32>& off, u32 nseq, const bool* present_in = nullptr);
const SegTables& seg_tables();
void seg_mask(const u8* S, int n, u8* out);
void load_ref_common(so_ctx* c, i64 r_lo, i64 r_hi);
void load_queries_common(so_ctx* c, bool parsed = false);
void file_stamp(const char* path, long long& size, long long& mtime_ns);
bool read_file(const char* path, std::string& out);
// host_index.hip: chunk indexes (built, or read from the reference's files), band encodings
i64 chunk_threshold(so_ctx* c, const u32* d_counts /*sizes of the occupied buckets, ascending bucket order*/, u64 s1, u64 s2, u64 nn);
void warm_sort_modules(int device);
void build_index(so_ctx* c);
void load_index(so_ctx* c, const char* prefix);
void band_plan(so_ctx* c, ChunkIndex& ch, int bp, bool multi_ok, int* k_out, u64* nband_out);
ChunkIndex::BandEnc* band_encoding(so_ctx* c, ChunkIndex& ch, int ba, int bp, bool multi_ok);
void order_chunk(so_ctx* c, ChunkIndex& ch);
const ChunkIndex::RangeTab* range_table(so_ctx* c, ChunkIndex& ch, const ChunkIndex::BandEnc& e, int wb, u32 R);
// host_seed.hip: batch preparation and the seed stage
void prepare_batch(so_ctx* c, Batch& b, i64 q_lo, i64 q_hi);
void order_queries(so_ctx* c, Batch& b);
void order_open_queries(so_ctx* c, Batch& b, const unsigned long long* qh);
void* small_host(so_ctx* c);
u32 d2h_u32(so_ctx* c, const u32* p);
void stash_u32(so_ctx* c, const u32* p, int slot);
void d2h_pair(so_ctx* c, const u32* second, u32& a, u32& b);
void ensure_sort_tmp(so_ctx* c, size_t bytes);
const unsigned long long* chunk_qhits(so_ctx* c, Batch& b, int ci);
void chunk_qhits_deferred(so_ctx* c, Batch& b, int ci);
bool class_takes_sorted_path(so_ctx* c, ChunkIndex& ch, u32 maxq, unsigned long long hits, unsigned long long nq);
void seed_stage(so_ctx* c, Batch& b, int ci);
void seed_pass(so_ctx* c, Batch& b, int ci, u32 qa, u32 qb, double t0, StageClock& sc);
// host_phase2.hip: candidate order, banded alignments in rounds, row emission
void emit_join(so_ctx* c, HitBuf& out);
void phase2(so_ctx* c, Batch& b, HitBuf& out);
// host_search.hip: batches of a search, the work pre-pass
void search_loaded(so_ctx* c, i64 q_lo, i64 q_hi, HitBuf& out);
void query_work(so_ctx* c, i64 q_lo, i64 q_hi, u64* out);
// host_abi.hip: row text
void format_hit_into(so_ctx* c, const so_hit& h, std::vector<char>& out);
std::string format_hit(so_ctx* c, const so_hit& h);
extern std::string g_create_err;


template <class F>
int guarded(so_ctx* c, F f) {
    try {
        if (!c) return 1;
        set_tune(&c->tune);
        g_poison = (int)c->tune.poison;
        HIP_CHECK(hipSetDevice(c->device));
        f();
        c->err.clear();
        return 0;
    } catch (const std::exception& e) {
        if (c) c->err = e.what();
        return 1;
    }
}
