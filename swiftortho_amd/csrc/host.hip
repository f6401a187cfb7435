// host.hip -- libsohit.so host side: context, FASTA sets resident in HBM, per-chunk index build,
// the batched search pipeline, row formatting and the C ABI of include/sohit.h.
//
// Product code.  Nothing here may call into oracle/ (the CPU restatement is test infrastructure);
// there is no CPU fallback for any device stage.  Host-side work is limited to what the reference
// also does outside its hot loops: FASTA indexing (fsearch.py:1543-1553, 2182-2199), SEG-like
// query masking (2872-2928; table-driven, bit-identical libm logs), the per-chunk mu+2sd threshold
// from exact device-side integer sums (746-761, 2248-2250), and text formatting (43-61, 3234-3243).
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

namespace {

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// libm entry points reached through volatile pointers so that no compiler folds pow(x, 2) / log
double (*volatile p_pow)(double, double) = pow;
double (*volatile p_log)(double) = log;
double (*volatile p_log10)(double) = log10;
double (*volatile p_sqrt)(double) = sqrt;

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t q = s.find(sep, p);
        if (q == std::string::npos) {
            out.push_back(s.substr(p));
            break;
        }
        out.push_back(s.substr(p, q - p));
        p = q + 1;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// BLOSUM62 by score class (fsearch.py:330-346).  Classes 0..22 = ARNDCQEGHILKMFPSTWYVBZX in either
// case, class 23 = every other byte (-4 against everything, itself included).
// ---------------------------------------------------------------------------------------------
const char B62_ORDER[] = "ARNDCQEGHILKMFPSTWYVBZX";
const signed char B62_ROWS[23][23] = {
    {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0, -2, -1, 0},
    {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3, -1, 0, -1},
    {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3, 3, 0, -1},
    {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3, 4, 1, -1},
    {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -3, -2},
    {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2, 0, 3, -1},
    {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1},
    {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3, -1, -2, -1},
    {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3, 0, 0, -1},
    {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3, -3, -3, -1},
    {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1, -4, -3, -1},
    {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2, 0, 1, -1},
    {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1, -3, -1, -1},
    {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1, -3, -3, -1},
    {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2, -2, -1, -2},
    {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2, 0, 0, 0},
    {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0, -1, -1, 0},
    {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3, -4, -3, -2},
    {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1, -3, -2, -1},
    {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4, -3, -2, -1},
    {-2, -1, 3, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4, 1, -1},
    {-1, 0, 0, 1, -3, 3, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -3, -2, -2, 1, 4, -1},
    {0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1, -1, -1, -1},
};

void build_score_maps(u8 smap[256], signed char b62c[SCLS_N * SCLS_N]) {
    for (int i = 0; i < 256; ++i) smap[i] = SCLS_N - 1;
    for (int k = 0; k < 23; ++k) {
        smap[(u8)B62_ORDER[k]] = (u8)k;
        smap[(u8)(B62_ORDER[k] + 32)] = (u8)k;
    }
    for (int a = 0; a < SCLS_N; ++a)
        for (int b = 0; b < SCLS_N; ++b) b62c[a * SCLS_N + b] = (a < 23 && b < 23) ? B62_ROWS[a][b] : -4;
}

// generate_nr_tbl (fsearch.py:406-422), bytes 0..255 only
void nr_table(const std::string& gaa, int tbl[256]) {
    for (int i = 0; i < 256; ++i) tbl[i] = i;
    std::string up = gaa;
    for (auto& c : up) c = (char)toupper((unsigned char)c);
    for (auto& grp : split(up, ',')) {
        int flag = 1024;
        for (unsigned char c : grp) flag = std::min(flag, (int)c);
        for (unsigned char c : grp) {
            tbl[c] = flag;
            tbl[(unsigned char)tolower(c)] = flag;
        }
    }
}

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

// 5-bit hash classes for one residue array under the run's alphabets
void byte_presence(const u8* bytes, size_t n, bool present[256]) {
    u64 cnt[256] = {0};
    for (size_t i = 0; i < n; ++i) cnt[bytes[i]]++;
    for (int b = 0; b < 256; ++b) present[b] = cnt[b] != 0;
}

void build_hash_classes(const bool present[256], const std::vector<std::array<int, 256>>& codes, u8 hmap[256], HashLut& lut) {
    memset(&lut, 0, sizeof lut);
    std::vector<std::vector<int>> tuples;
    for (int b = 0; b < 256; ++b) {
        hmap[b] = HCLS_X;  // absent bytes never occur; x/X reject the window
        if (!present[b] || b == 'x' || b == 'X') continue;
        std::vector<int> t;
        for (auto& c : codes) t.push_back(c[b]);
        size_t k = 0;
        for (; k < tuples.size(); ++k)
            if (tuples[k] == t) break;
        if (k == tuples.size()) {
            if (tuples.size() >= HCLS_SEP)
                throw SoError("more than 30 distinct residue codes in the input: cannot pack hash classes into 5 bits");
            tuples.push_back(t);
            for (size_t a = 0; a < codes.size(); ++a) lut.v[a][k] = (u32)t[a];
        }
        hmap[b] = (u8)k;
    }
}

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
        u64 used = 0;
    };
    std::vector<std::unique_ptr<BandEnc>> encs;
    u64 enc_clock = 0;
};

}  // namespace

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
    DevBuf<u32> ix_pcount, ix_bkt, ix_bkt2, ix_flags, ix_ridx, ix_plan, ix_tk;  // index build scratch
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
    bool rows_in_flight = false;
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

// the switches the launch helpers see: those of the context whose API call ran last (one context per process in every use of the
// library; tests create theirs one after the other)
static std::atomic<const Tune*> g_tune{nullptr};
const Tune& tune() {
    static const Tune dflt;
    const Tune* t = g_tune.load(std::memory_order_acquire);
    return t ? *t : dflt;
}
void set_tune(const Tune* t) { g_tune.store(t, std::memory_order_release); }
int g_poison = -1;
static inline u64 cand_limit() { return tune().cand_limit > 0 ? (u64)tune().cand_limit : 0xFFFFFFF0ull; }

namespace {

std::string g_create_err;

void set_params(so_ctx* c, const so_params* p) {
    c->seeds = p->seeds ? p->seeds : "111111";
    c->alphabet = p->alphabet ? p->alphabet : "AST,CFILMVY,DN,EQ,G,H,KR,P,W";
    if (c->alphabet == "aa9") c->alphabet = "AST,CFILMVY,DN,EQ,G,H,KR,P,W";
    if (c->alphabet == "aa20") c->alphabet = "A,S,T,C,F,I,L,M,V,Y,D,N,E,Q,G,H,K,R,P,W";
    c->nc = p->nc;
    c->chunk = p->chunk > 0 ? p->chunk : 50000;
    c->step = p->step;
    c->v = p->max_hits;
    c->thr = p->thr;
    c->expect = p->expect;
    c->max_miss = std::max(p->max_miss, 1e-3);  // fsearch.py:2970
    c->filter = p->filter != 0;
    c->profile = p->profile != 0;
    if (c->step < 1) throw SoError("-j (seed step) must be >= 1");
    auto pats = split(c->seeds, ',');
    if (c->nc < 1) {
        // NC = ht < 1 and bins or ht (fsearch.py:2228-2231); self.scale was overwritten with the parameter (-1, line 2216), so
        // bins = min(int(pow(-1, mw)) * nssp * 5, 128Mi) = 5 * nssp for an even maximum seed weight and negative for an odd
        // one (an empty table and an IndexError in the reference): refused.
        int mw = 0;
        for (auto& sp : pats) mw = std::max(mw, (int)std::count(sp.begin(), sp.end(), '1'));
        c->nc = std::min<i64>((mw % 2 == 0 ? 1 : -1) * (i64)pats.size() * 5, 128ll * 1024 * 1024);
        if (c->nc < 1) throw SoError("-M < 1 with an odd maximum seed weight: the reference derives a negative bucket count (fsearch.py:2228-2231); pass -M");
    }
    if (c->nc > 0xFFFFFFF0ll) throw SoError("-M (bucket count) must be < 2^32");
    auto alphas = split(c->alphabet, '/');
    if (pats.empty() || (int)pats.size() > MAX_PATTERNS) throw SoError("1.." + std::to_string(MAX_PATTERNS) + " seed patterns supported");
    if (alphas.empty() || (int)alphas.size() > MAX_ALPHA) throw SoError("1.." + std::to_string(MAX_ALPHA) + " alphabets supported");
    memset(&c->cfg, 0, sizeof c->cfg);
    c->cfg.S = (int)pats.size();
    c->cfg.A = (int)alphas.size();
    c->cfg.nc = (u32)c->nc;
    c->cfg.mink = 1 << 30;
    for (int s = 0; s < c->cfg.S; ++s) {
        const std::string& sp = pats[s];
        if (sp.empty() || sp.size() > MAX_SEEDLEN) throw SoError("seed pattern length must be 1.." + std::to_string(MAX_SEEDLEN));
        c->cfg.klen[s] = (int)sp.size();
        c->cfg.mink = std::min(c->cfg.mink, (int)sp.size());
        u32 care = 0;
        for (size_t j = 0; j < sp.size(); ++j)
            if (sp[j] != '0') care |= 1u << j;  // fsearch.py:541 `space[j] != '0'`
        c->cfg.care[s] = care;
    }
    c->codes.clear();
    for (auto& a : alphas) {
        std::array<int, 256> t;
        nr_table(a, t.data());
        c->codes.push_back(t);
    }
}

void upload_constants(so_ctx* c) {
    build_score_maps(c->smap, c->b62c);
    c->d_smap.ensure(256);
    c->d_hmap.ensure(256);
    c->d_b62c.ensure(SCLS_N * SCLS_N);
    HIP_CHECK(hipMemcpy(c->d_smap.p, c->smap, 256, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c->d_b62c.p, c->b62c, SCLS_N * SCLS_N, hipMemcpyHostToDevice));
    // score2bit (fsearch.py:1066-1071) tabulated on the host: the device never evaluates it
    std::vector<int> bt(so_ctx::BITTAB_N);
    for (int s = 0; s < so_ctx::BITTAB_N; ++s) bt[s] = (int)((.267 * (double)s + 3.1941832122778293) / 0.69314718055994529);
    c->d_bittab.ensure(so_ctx::BITTAB_N);
    HIP_CHECK(hipMemcpy(c->d_bittab.p, bt.data(), bt.size() * sizeof(int), hipMemcpyHostToDevice));
    c->d_stats.ensure(4 + 4 * INDEX_STATS_BLOCKS + 8);
}

// device-resident arrays of a sequence set given its (possibly masked) residues
// Derived device arrays of a sequence set whose residues (s.d_res) and offsets (s.d_off) are already
// on the device: score classes, 5-bit hash-class stream, owner map.  `present` = bytes that can occur.
void layout_set(so_ctx* c, SeqSet& s, const bool present[256], size_t nres, u32 nseq) {
    u8 hmap[256];
    build_hash_classes(present, c->codes, hmap, s.lut);
    s.d_scls_store.ensure(nres + SCLS_PAD_FRONT + SCLS_PAD_BACK);
    s.d_scls.p = s.d_scls_store.p + SCLS_PAD_FRONT;
    s.d_scls4_store.ensure(nres + SCLS_PAD_FRONT + SCLS_PAD_BACK);
    s.d_scls4.p = s.d_scls4_store.p + SCLS_PAD_FRONT;
    HIP_CHECK(hipMemcpyAsync(c->d_hmap.p, hmap, 256, hipMemcpyHostToDevice, c->st));
    launch_scls(s.d_res.p, nres, c->d_smap.p, s.d_scls.p, s.d_scls4.p, c->st);
    s.d_pcls_store.ensure(nres + (size_t)PCLS_PAD * (nseq + 2) + 64), s.d_pcls4_store.ensure(nres + (size_t)PCLS_PAD * (nseq + 2) + 64);
    s.d_pcls.p = s.d_pcls_store.p + PCLS_PAD, s.d_pcls4.p = s.d_pcls4_store.p + PCLS_PAD;
    launch_pad_cls(s.d_scls.p, s.d_off.p, nseq, s.d_pcls.p, s.d_pcls4.p, c->st);
    s.ug_valid = false;
    s.d_bound.ensure((size_t)nseq + 4);
    launch_seq_bound(s.d_scls.p, s.d_off.p, nseq, c->b62c, s.d_bound.p, c->st);
    if ((u64)nres + nseq + 64 > 0xFFFFFFF0ull) throw SoError("sequence set too large for 32-bit packed positions");
    s.P = (u32)(nres + nseq);
    s.Ppad = (s.P + 31u) & ~31u;
    if (s.Ppad == 0) s.Ppad = 32;
    s.d_words.ensure((size_t)s.Ppad / 32 * 5 + 4);
    HIP_CHECK(hipMemsetAsync(s.d_words.p, 0, ((size_t)s.Ppad / 32 * 5 + 4) * sizeof(u32), c->st));
    s.d_pseq.ensure(s.Ppad);
    c->d_pcls.ensure(s.Ppad);
    launch_layout(s.d_res.p, s.d_off.p, nseq, s.P, s.Ppad, c->d_hmap.p, s.d_pseq.p, c->d_pcls.p, s.d_words.p, c->st);
    HIP_CHECK(hipStreamSynchronize(c->st));  // hmap (stack) must outlive the copy
}

// host residues -> device, then layout
void upload_set(so_ctx* c, SeqSet& s, const u8* residues, const std::vector<u32>& off, u32 nseq, const bool* present_in = nullptr) {
    const size_t nres = off[nseq];
    bool present[256];
    if (present_in) memcpy(present, present_in, sizeof present);
    else byte_presence(residues, nres, present);
    s.d_res.ensure(nres + 64);
    s.d_off.ensure((size_t)nseq + 1);
    HIP_CHECK(hipMemcpyAsync(s.d_res.p, residues, nres, hipMemcpyHostToDevice, c->st));
    HIP_CHECK(hipMemcpyAsync(s.d_off.p, off.data(), ((size_t)nseq + 1) * sizeof(u32), hipMemcpyHostToDevice, c->st));
    layout_set(c, s, present, nres, nseq);
}

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
const SegTables& seg_tables() {
    static SegTables t;
    return t;
}

void seg_mask(const u8* S, int n, u8* out) {
    const SegTables& T = seg_tables();
    const double minent = 2.2, window = 12.;
    if (n <= 0) return;
    std::vector<u8> s((size_t)n);
    for (int i = 0; i < n; ++i) s[i] = (S[i] >= 'a' && S[i] <= 'z') ? (u8)(S[i] - 32) : S[i];
    int counts[256];
    int order[12], norder = 0;
    const int w = std::min(n, 12);
    bool seen[256];
    for (int i = 0; i < w; ++i) seen[s[i]] = false, counts[s[i]] = 0;
    // Counter(seq): first occurrence 0, then +1 each; the explicit loop adds 1 per char -> 2*occ - 1
    for (int i = 0; i < w; ++i) {
        u8 c = s[i];
        if (!seen[c]) seen[c] = true, counts[c] = 0, order[norder++] = c;
        else counts[c] += 1;
    }
    for (int i = 0; i < w; ++i) counts[s[i]] += 1;
    double ent = 0;
    for (int k = 0; k < norder; ++k) {
        int j = counts[order[k]];
        double freq = (double)j / ((double)w * 1.);
        ent -= freq * T.lgn[w][j];
    }
    ent /= T.log2v;
    // characters entering later start from 0
    std::vector<u8> mask((size_t)n, 0);
    if (ent < minent) mask[0] = 1;
    bool touched[256];
    memset(touched, 0, sizeof touched);
    for (int k = 0; k < norder; ++k) touched[order[k]] = true;
    for (int i = 1; i < n - 12 + 1; ++i) {
        const u8 pre = s[i - 1], cur = s[i + 11];
        if (pre == cur) {
            mask[i] = mask[i - 1];
            continue;
        }
        if (!touched[cur]) touched[cur] = true, counts[cur] = 0;
        const int pre_count = counts[pre];
        counts[pre] -= 1;
        const int cur_count = counts[cur];
        counts[cur] += 1;
        double a = (double)pre_count / window, b = (double)counts[pre] / window;
        double t;
        if (counts[pre] != 0) {
            t = (a * T.lg12[pre_count] - b * T.lg12[counts[pre]]) / T.log2v;
            if (t == 0) t = a * T.lg12[pre_count] / T.log2v;
        } else {
            t = a * T.lg12[pre_count] / T.log2v;
        }
        ent += t;
        a = (double)cur_count / window;
        b = (double)counts[cur] / window;
        if (cur_count != 0) {
            t = (a * T.lg12[cur_count] - b * T.lg12[counts[cur]]) / T.log2v;
            if (t == 0) t = -b * T.lg12[counts[cur]] / T.log2v;
        } else {
            t = -b * T.lg12[counts[cur]] / T.log2v;
        }
        ent += t;
        if (ent < minent) mask[i] = 1;
    }
    const int Nws = std::max(0, n - 12);
    if (mask[Nws] == 1)
        for (int i = Nws; i < n; ++i) mask[i] = 1;
    int st = 0, o = 0;
    while (st < n) {
        if (mask[st] == 0) {
            out[o++] = s[st];
            st += 1;
        } else {
            for (int k = 0; k < 12 && o < n; ++k) out[o++] = 'x';
            st += 12;
        }
    }
}

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
// Reference side
// ---------------------------------------------------------------------------------------------
void load_ref_common(so_ctx* c, i64 r_lo, i64 r_hi) {
    const double t0 = wall();
    c->ref.parse();
    const double t1 = wall();
    c->r_lo = r_lo, c->r_hi = r_hi;
    upload_set(c, c->ref, c->ref.res.data(), c->ref.off, (u32)c->ref.N);   // (ends with a stream synchronisation)
    c->lt["load.ref_parse"] = (t1 - t0) * 1e3, c->lt["load.ref_h2d_layout"] = (wall() - t1) * 1e3;
    c->ref_loaded = true;
    c->band_plans.clear();
    c->index_built = false;
    c->chunks.clear();
    c->cnt.ref_seqs = c->ref.N;
    c->cnt.ref_aa = (i64)c->ref.res.size();
}

// threshold = int(mu + 2 sd) (fsearch.py:2248-2250, 746-761) from exact integer sums; when the
// value is too close to an integer for that to be safe, replay the reference's sequential
// floating-point loops over the counts in bucket order.
i64 chunk_threshold(so_ctx* c, const u32* d_counts /*sizes of the occupied buckets, ascending bucket order*/, u64 s1, u64 s2, u64 nn) {
    const double N = (double)(nn + 1);
    const double mu = (double)s1 / N;
    long double lm = (long double)mu;
    long double ss = (long double)s2 - 2.0L * lm * (long double)s1 + (long double)nn * lm * lm;
    if (ss < 0) ss = 0;
    long double T = lm + 2.0L * sqrtl(ss / (long double)N);
    long double fl = floorl(T);
    long double margin = ((long double)nn * 8e-16L + 1e-11L) * (T + 1.0L);
    const bool forced = tune().exact_threshold;
    if (!forced && T - fl > margin && (fl + 1.0L) - T > margin) return (i64)fl;
    // exact replay
    std::vector<u32> counts((size_t)nn);  // the reference walks all NC counts and skips the zeros: same sequence
    if (nn) HIP_CHECK(hipMemcpyAsync(counts.data(), d_counts, (size_t)nn * sizeof(u32), hipMemcpyDeviceToHost, c->st));
    HIP_CHECK(hipStreamSynchronize(c->st));
    i64 Nn = 1;
    double m = 0.;
    for (u32 v : counts)
        if (v > 0) m += (double)v, Nn += 1;
    m /= (double)Nn;
    double sd = 0.;
    for (u32 v : counts)
        if (v > 0) sd += p_pow((double)v - m, 2);
    sd = p_sqrt(sd / (double)Nn);
    return (i64)(m + 2 * sd);
}

void* small_host(so_ctx* c);
u32 d2h_u32(so_ctx* c, const u32* p);
void ensure_sort_tmp(so_ctx* c, size_t bytes);

// first launches of the library sorts (they load their code objects: ~15 ms for the device-wide unit, ~40 ms for the segmented one)
void warm_sort_modules(int device) {
    try {
        HIP_CHECK(hipSetDevice(device));
        hipStream_t st;
        HIP_CHECK(hipStreamCreate(&st));
        {
            DevBuf<u32> k, k2, seg;
            DevBuf<u64> v, v2;
            DevBuf<u8> tmp;
            k.ensure(8), k2.ensure(8), seg.ensure(8), v.ensure(8), v2.ensure(8);
            HIP_CHECK(hipMemsetAsync(k.p, 0, 8 * sizeof(u32), st));
            HIP_CHECK(hipMemsetAsync(v.p, 0, 8 * sizeof(u64), st));
            const u32 sg[2] = {0, 2};
            HIP_CHECK(hipMemcpyAsync(seg.p, sg, sizeof sg, hipMemcpyHostToDevice, st));
            tmp.ensure(std::max(sort_pairs_u64_u32_temp_bytes(2, 8), sort_keys_u64_seg_temp_bytes(2, 1, 0, 8)) + 256);
            sort_pairs_u64_u32(tmp.p, tmp.cap, v.p, v2.p, k.p, k2.p, 2, 8, st);   // (the index build no longer sorts: the sorted path's units)
            sort_keys_u64_seg(tmp.p, tmp.cap, v.p, v2.p, 2, 1, seg.p, 0, 8, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        (void)hipStreamDestroy(st);
    } catch (...) {
        // a failure here is the caller's business at its own first sort
    }
}

void build_index(so_ctx* c) {
    if (!c->ref_loaded) throw SoError("so_build_index: no reference loaded");
    if (c->index_built) return;
    if (c->warm.joinable()) c->warm.join();
    const double t0 = wall();
    c->chunks.clear();
    const i64 N = c->ref.N;
    i64 Start = c->r_lo == -1 ? 0 : std::max<i64>(0, c->r_lo);   // makedb, fsearch.py:2286-2288
    i64 End = c->r_hi == -1 ? N : c->r_hi;
    const u32 NC = (u32)c->nc;
    c->cnt.index_entries = 0;
    const bool dbg = tune().debug_index;   // wall laps of the build's steps (stderr)
    double tl = wall();
    auto dlap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->st);
        const double n = wall();
        fprintf(stderr, "[sohit index] %-28s %.3f ms\n", what, (n - tl) * 1e3);
        tl = n;
    };
    for (i64 s = Start; s < End; s += c->chunk) {
        std::unique_ptr<ChunkIndex> ch;
        if (!c->spare_chunks.empty()) {
            ch = std::move(c->spare_chunks.back());
            c->spare_chunks.pop_back();
            ch->d_sh_subj = ch->d_sh_diag = -1;  // key addends belong to the old entries
            for (auto& e : ch->encs) e->k = -1;
        } else {
            ch = std::make_unique<ChunkIndex>();
        }
        i64 e = std::min(s + c->chunk, End);
        ch->seq_lo = std::min<i64>(std::max<i64>(0, s), N);  // build_msav clamps, 2233-2234
        ch->seq_hi = std::min<i64>(e, N);
        if (ch->seq_hi < ch->seq_lo) ch->seq_hi = ch->seq_lo;
        ch->p_lo = c->ref.off[ch->seq_lo] + (u32)ch->seq_lo;
        ch->p_hi = c->ref.off[ch->seq_hi] + (u32)ch->seq_hi;
        ch->maxslen = 0;
        for (i64 j = ch->seq_lo; j < ch->seq_hi; ++j) ch->maxslen = std::max(ch->maxslen, c->ref.len(j));
        {
            // SOHIT_DIR_MAX: largest -M served by the bitmap + rank directory (NC / 4 bytes per chunk)
            ch->use_dir = (u64)NC <= (tune().dir_max >= 0 ? (u64)tune().dir_max : (1ull << 28));
        }
        // 1. windows per position -> exclusive scan -> (bucket, entry) pairs in position order
        const u32 npos = ch->p_hi - ch->p_lo;
        c->ix_pcount.ensure((size_t)npos + 4);
        c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)npos + 1) + 8);
        dlap("alloc pcount / scan tmp");
        launch_index_windows(false, c->ref.d_words.p, c->ref.d_pseq.p, c->ref.d_off.p, ch->p_lo, ch->p_hi, c->ref.Ppad, (u32)ch->seq_lo, c->cfg,
                             c->ref.lut, (u32)c->step, c->ix_pcount.p, nullptr, nullptr, c->st);
        u32 E = 0;
        if (npos) E = d2h_u32(c, scan_u32(c->ix_pcount.p, c->ix_pcount.p, npos, false, c->d_scan_tmp.p, c->st));
        dlap("windows (count) + scan");
        if ((u64)E >= (1ull << 29)) throw SoError("chunk index exceeds 2^29 entries (the lookup kernel addresses 8-byte slots with 32-bit byte offsets); lower -c");
        ch->E = E;
        ch->U = 0;
        ch->entries.ensure((size_t)E + 4);
        u64 s2 = 0;
        if (E) {
            c->ix_bkt.ensure((size_t)E + 4), c->ix_bkt2.ensure((size_t)E + 4), c->ix_ent.ensure((size_t)E + 4);
            dlap("alloc entries / pairs");
            launch_index_windows(true, c->ref.d_words.p, c->ref.d_pseq.p, c->ref.d_off.p, ch->p_lo, ch->p_hi, c->ref.Ppad, (u32)ch->seq_lo, c->cfg,
                                 c->ref.lut, (u32)c->step, c->ix_pcount.p, c->ix_bkt.p, c->ix_ent.p, c->st);
            // 2. group by bucket id (ascending): the slot layout of the reference's CSR
            const int bbits = ceil_log2((u64)NC);
            dlap("windows (emit)");
            // (hand-written: two counting passes, k_ixsort.hip; members of a bucket land in no particular order)
            (void)bbits;
            c->ix_plan.ensure(ixsort_plan_elems((u32)NC) + 8), c->ix_tk.ensure((size_t)E + 4), c->ix_tv.ensure((size_t)E + 4);
            c->d_scan_tmp.ensure(scan_u32_temp_elems(ixsort_plan_elems((u32)NC)) + 8);
            dlap("alloc grouping scratch");
            ixsort_pairs(c->ix_bkt.p, c->ix_ent.p, E, (u32)NC, c->ix_plan.p, c->d_scan_tmp.p, c->ix_tk.p, c->ix_tv.p, c->ix_bkt2.p, ch->entries.p, c->st);
            dlap("pair grouping");
            // 3. runs -> occupied bucket list, first slots, sizes
            c->ix_flags.ensure((size_t)E + 4), c->ix_ridx.ensure((size_t)E + 4);
            c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)E + 1) + 8);
            launch_run_heads(c->ix_bkt2.p, E, c->ix_flags.p, c->st);
            const u32 U = d2h_u32(c, scan_u32(c->ix_flags.p, c->ix_ridx.p, E, false, c->d_scan_tmp.p, c->st));
            ch->U = U;
            ch->ub.ensure((size_t)U + 4), ch->ubeg.ensure((size_t)U + 4), ch->ucnt.ensure((size_t)U + 4);
            launch_run_list(c->ix_bkt2.p, c->ix_flags.p, c->ix_ridx.p, E, U, ch->ub.p, ch->ubeg.p, ch->ucnt.p, c->st);
            dlap("run heads / list");
            // 4. threshold statistics over the occupied buckets (sum c = E, sum c^2, count = U)
            launch_index_stats(ch->ucnt.p, U, c->d_stats.p, c->st);
            u64* stats = (u64*)small_host(c);
            u32* last_lo_h = (u32*)(stats + 4);
            HIP_CHECK(hipMemcpyAsync(stats, c->d_stats.p, 4 * sizeof(u64), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipMemcpyAsync(last_lo_h, ch->ubeg.p + (U - 1), sizeof(u32), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipStreamSynchronize(c->st));
            s2 = stats[1];
            const u32 last_lo = *last_lo_h;
            dlap("stats");
            // 5. bucket directory: bitmap + rank table over the NC bucket ids, or (very large -M) an open-addressed map, load <= 1/2
            if (ch->use_dir) {
                const size_t nd = (size_t)NC / 32 + 2;
                ch->dir.ensure(nd);
                HIP_CHECK(hipMemsetAsync(ch->dir.p, 0, nd * sizeof(u64), c->st));
                launch_dir_build(ch->ub.p, U, ch->dir.p, c->st);
            } else {
                u32 cap = 1024;
                int lg = 10;
                while (cap < 2 * U) cap <<= 1, ++lg;
                ch->hkey.ensure(cap), ch->hval.ensure(cap);
                ch->hshift = 32 - lg, ch->hmask = cap - 1;
                HIP_CHECK(hipMemsetAsync(ch->hkey.p, 0xFF, (size_t)cap * sizeof(u32), c->st));
                launch_htab_insert(ch->ub.p, ch->ubeg.p, U, ch->hkey.p, ch->hval.p, ch->hshift, ch->hmask, c->st);
            }
            // 6. the reference never reads the last locus slot: park the last bucket's smallest entry there
            launch_index_fixlast(ch->entries.p, last_lo, E, c->st);
        } else if (ch->use_dir) {
            const size_t nd = (size_t)NC / 32 + 2;
            ch->dir.ensure(nd);
            ch->ubeg.ensure(4);
            HIP_CHECK(hipMemsetAsync(ch->dir.p, 0, nd * sizeof(u64), c->st));
        } else {
            ch->hkey.ensure(1024), ch->hval.ensure(1024);
            ch->hshift = 22, ch->hmask = 1023;
            HIP_CHECK(hipMemsetAsync(ch->hkey.p, 0xFF, 1024 * sizeof(u32), c->st));
        }
        dlap("directory + fixlast");
        ch->threshold = chunk_threshold(c, ch->ucnt.p, (u64)E, s2, (u64)ch->U);
        ch->s2 = s2;
        HIP_CHECK(hipStreamSynchronize(c->st));
        c->cnt.index_entries += ch->E;
        c->chunks.push_back(std::move(ch));
    }
    c->spare_chunks.clear();
    c->cnt.n_chunks = (i64)c->chunks.size();
    c->index_built = true;
    c->cnt.index_ms += (wall() - t0) * 1e3;
}

// ---------------------------------------------------------------------------------------------
// Fasta.load (fsearch.py:2355-2444): chunk indexes read back from the reference's on-disk format -- `<prefix>.<k>.idx` (locus:
// int32 per slot = soas[j] + pos, a bucket's members in the reference's slot order), `.soas` (prefix lengths of the chunk's
// sequences), `.bin` (start[NC] + trailer `offset;offend;max weight;threshold;NC;seeds;alphabet` + its length in one byte) -- and
// made resident in the layout build_index() produces.  Like the reference's load, the sequences come from the FASTA file (loaded
// with so_load_ref), the threshold from the trailer.  The FILE's slot order is kept (so the slot the reference never reads,
// fsearch.py:2277 / 2539, is the file's last one: no k_index_fixlast); an entry's tag (alphabet x pattern), which the file does not
// hold and the consumers derive the visiting order from, is recovered by hashing the entry's window under every (alphabet, pattern)
// and matching the bucket: equal (subject, position) members of one bucket were inserted in ascending tag order, i.e. stand in
// descending tag order in the file.
// ---------------------------------------------------------------------------------------------
bool read_file(const char* path, std::string& out);

// bucket of the window at `pos` of `seq` under (alphabet a, pattern s), fsearch.py:519-556; false = no window (too short, x / X inside)
bool host_window_bucket(const so_ctx* c, const u8* seq, u32 len, u32 pos, int a, int s, u32* bucket) {
    const int k = c->cfg.klen[s];
    if ((u64)pos + (u64)k > (u64)len) return false;
    const u32 care = c->cfg.care[s];
    u32 n = 0x811c9dc5u;
    for (int j = 0; j < k; ++j) {
        const u8 ch = seq[pos + j];
        if (ch == 'x' || ch == 'X') return false;
        if ((care >> j) & 1u) n = (n ^ (u32)c->codes[a][ch]) * 0x01000193u;
    }
    n = (n ^ (u32)s) * 0x01000193u;
    *bucket = n % (u32)c->nc;
    return true;
}

void load_index(so_ctx* c, const char* prefix) {
    if (!c->ref_loaded) throw SoError("so_load_index: no reference loaded (the index files hold no sequences: so_load_ref first)");
    if (!prefix || !*prefix) throw SoError("so_load_index: empty prefix");
    if (c->warm.joinable()) c->warm.join();
    const double t0 = wall();
    c->chunks.clear();
    c->spare_chunks.clear();
    c->index_built = false;
    c->cnt.index_entries = 0;
    const u32 NC = (u32)c->nc;
    const int AS = c->cfg.A * c->cfg.S;
    for (int k = 0;; ++k) {
        const std::string name = std::string(prefix) + "." + std::to_string(k);
        std::string bin, idx, soas_b;
        if (!read_file((name + ".bin").c_str(), bin)) {
            if (k == 0) throw SoError("so_load_index: cannot read " + name + ".bin");
            break;
        }
        if (!read_file((name + ".idx").c_str(), idx)) throw SoError("so_load_index: cannot read " + name + ".idx");
        if (!read_file((name + ".soas").c_str(), soas_b)) throw SoError("so_load_index: cannot read " + name + ".soas");
        // trailer (fsearch.py:2380-2387): the last byte is its length
        if (bin.empty()) throw SoError("so_load_index: " + name + ".bin is empty");
        const size_t tl = (u8)bin.back();
        if (bin.size() < tl + 1) throw SoError("so_load_index: " + name + ".bin has no parameter trailer");
        const size_t tbeg = bin.size() - tl - 1;
        auto f = split(bin.substr(tbeg, tl), ';');
        if (f.size() != 7) throw SoError("so_load_index: " + name + ".bin: malformed parameter trailer");
        i64 offset, nc_f, thr_f;
        try {
            offset = std::stoll(f[0]), thr_f = std::stoll(f[3]), nc_f = std::stoll(f[4]);
        } catch (...) {
            throw SoError("so_load_index: " + name + ".bin: malformed parameter trailer");
        }
        if (f[6] == "aa9") f[6] = "AST,CFILMVY,DN,EQ,G,H,KR,P,W";
        if (f[6] == "aa20") f[6] = "A,S,T,C,F,I,L,M,V,Y,D,N,E,Q,G,H,K,R,P,W";
        if (nc_f != (i64)NC || f[5] != c->seeds || f[6] != c->alphabet)
            throw SoError("so_load_index: " + name + " was built with -M " + f[4] + " -s " + f[5] + " -r " + f[6] + ", the context with -M " +
                          std::to_string(NC) + " -s " + c->seeds + " -r " + c->alphabet);
        if (tbeg != (size_t)NC * 4) throw SoError("so_load_index: " + name + ".bin does not hold NC start values");
        if (soas_b.size() < 4 || soas_b.size() % 4 || idx.size() % 4) throw SoError("so_load_index: " + name + ": truncated .soas / .idx");
        const u32* start = reinterpret_cast<const u32*>(bin.data());
        const u32* soas = reinterpret_cast<const u32*>(soas_b.data());
        const u32* locus = reinterpret_cast<const u32*>(idx.data());
        const i64 M = (i64)(soas_b.size() / 4) - 1;
        const u64 E64 = idx.size() / 4;
        if (E64 >= (1ull << 29)) throw SoError("chunk index exceeds 2^29 entries (the lookup kernel addresses 8-byte slots with 32-bit byte offsets)");
        const u32 E = (u32)E64;
        offset = std::max<i64>(offset, 0);   // build_msav writes `start` unclamped (-1 = from the first sequence)
        if (offset + M > c->ref.N || soas[0] != 0) throw SoError("so_load_index: " + name + " does not belong to the loaded reference (sequence range)");
        for (i64 j = 0; j < M; ++j)
            if (soas[j + 1] - soas[j] != c->ref.len(offset + j))
                throw SoError("so_load_index: " + name + " does not belong to the loaded reference (sequence lengths)");
        auto ch = std::make_unique<ChunkIndex>();
        ch->seq_lo = offset, ch->seq_hi = offset + M;
        ch->p_lo = c->ref.off[ch->seq_lo] + (u32)ch->seq_lo;
        ch->p_hi = c->ref.off[ch->seq_hi] + (u32)ch->seq_hi;
        ch->maxslen = 0;
        for (i64 j = ch->seq_lo; j < ch->seq_hi; ++j) ch->maxslen = std::max(ch->maxslen, c->ref.len(j));
        ch->use_dir = (u64)NC <= (tune().dir_max >= 0 ? (u64)tune().dir_max : (1ull << 28));
        ch->E = E;
        ch->threshold = thr_f;
        // occupied buckets: start[b] = first slot of bucket b, bucket b ends where b + 1 begins (the last one at E)
        std::vector<u32> ub, ubeg, ucnt;
        u64 s2 = 0;
        for (u32 b = 0; b < NC; ++b) {
            const u64 st = start[b], ed = b + 1 < NC ? (u64)start[b + 1] : (u64)E;
            if (ed < st || ed > E) throw SoError("so_load_index: " + name + ".bin: start values are not a prefix sum of the .idx slots");
            if (ed > st) ub.push_back(b), ubeg.push_back((u32)st), ucnt.push_back((u32)(ed - st)), s2 += (ed - st) * (ed - st);
        }
        if (E && (ub.empty() || ubeg[0] != 0)) throw SoError("so_load_index: " + name + ".bin: slots in front of the first bucket");
        const u32 U = (u32)ub.size();
        ubeg.push_back(E);
        ch->U = U, ch->s2 = s2;
        // slots -> entries
        std::vector<u64> ent((size_t)E);
        std::atomic<bool> bad(false);
        const u8* res = c->ref.res.data();
        parallel_for((i64)E, [&](i64 i) {
            const u32 x = locus[i];
            const u32* p = std::upper_bound(soas, soas + M + 1, x);   // offset 0 of sequence j is soas[j] itself: largest j with soas[j] <= x
            const i64 j = (p - soas) - 1;
            if (j < 0 || j >= M || x - soas[j] >= (1u << 24)) {
                bad = true;
                return;
            }
            ent[(size_t)i] = ((u64)j << 32) | (u64)(x - soas[j]);
        });
        if (bad) throw SoError("so_load_index: " + name + ".idx: slot outside the chunk's sequences");
        if (AS > 1) {
            parallel_for((i64)U, [&](i64 kb) {
                const u32 b = ub[(size_t)kb];
                std::vector<u64> seen;   // (subject, pos) of the slots of this bucket so far
                for (u32 i = ubeg[(size_t)kb]; i < ubeg[(size_t)kb + 1]; ++i) {
                    const u64 e = ent[i];
                    const i64 j = (i64)(e >> 32);
                    const u32 pos = (u32)e;
                    const u8* sq = res + c->ref.off[offset + j];
                    const u32 ln = c->ref.len(offset + j);
                    int tags[MAX_ALPHA * MAX_PATTERNS], nt = 0;
                    for (int a = 0; a < c->cfg.A; ++a) {
                        u32 bk[MAX_PATTERNS];
                        bool ok[MAX_PATTERNS];
                        for (int s = 0; s < c->cfg.S; ++s) {
                            ok[s] = host_window_bucket(c, sq, ln, pos, a, s, &bk[s]);
                            for (int s2i = 0; ok[s] && s2i < s; ++s2i)
                                if (ok[s2i] && bk[s2i] == bk[s]) ok[s] = false;   // the reference's `visit`
                            if (ok[s] && bk[s] == b) tags[nt++] = a * c->cfg.S + s;
                        }
                    }
                    const int before = (int)std::count(seen.begin(), seen.end(), e);
                    if (before >= nt) {
                        bad = true;
                        return;
                    }
                    seen.push_back(e);
                    ent[i] = e | ((u64)tags[nt - 1 - before] << 24);
                }
            });
            if (bad) throw SoError("so_load_index: " + name + ".idx does not match the loaded reference under these seeds (a slot's window does not hash to its bucket)");
        }
        ch->entries.ensure((size_t)E + 4);
        ch->ub.ensure((size_t)U + 4), ch->ubeg.ensure((size_t)U + 4), ch->ucnt.ensure((size_t)U + 4);
        if (E) HIP_CHECK(hipMemcpyAsync(ch->entries.p, ent.data(), (size_t)E * sizeof(u64), hipMemcpyHostToDevice, c->st));
        if (U) {
            HIP_CHECK(hipMemcpyAsync(ch->ub.p, ub.data(), (size_t)U * sizeof(u32), hipMemcpyHostToDevice, c->st));
            HIP_CHECK(hipMemcpyAsync(ch->ucnt.p, ucnt.data(), (size_t)U * sizeof(u32), hipMemcpyHostToDevice, c->st));
        }
        HIP_CHECK(hipMemcpyAsync(ch->ubeg.p, ubeg.data(), ((size_t)U + 1) * sizeof(u32), hipMemcpyHostToDevice, c->st));
        if (ch->use_dir) {
            const size_t nd = (size_t)NC / 32 + 2;
            ch->dir.ensure(nd);
            HIP_CHECK(hipMemsetAsync(ch->dir.p, 0, nd * sizeof(u64), c->st));
            if (U) launch_dir_build(ch->ub.p, U, ch->dir.p, c->st);
        } else {
            u32 cap = 1024;
            int lg = 10;
            while (cap < 2 * U) cap <<= 1, ++lg;
            ch->hkey.ensure(cap), ch->hval.ensure(cap);
            ch->hshift = 32 - lg, ch->hmask = cap - 1;
            HIP_CHECK(hipMemsetAsync(ch->hkey.p, 0xFF, (size_t)cap * sizeof(u32), c->st));
            if (U) launch_htab_insert(ch->ub.p, ch->ubeg.p, U, ch->hkey.p, ch->hval.p, ch->hshift, ch->hmask, c->st);
        }
        HIP_CHECK(hipStreamSynchronize(c->st));   // the host vectors must outlive the copies
        c->cnt.index_entries += ch->E;
        c->chunks.push_back(std::move(ch));
    }
    c->cnt.n_chunks = (i64)c->chunks.size();
    c->index_built = true;
    c->cnt.index_ms += (wall() - t0) * 1e3;
}

// ---------------------------------------------------------------------------------------------
// Search
// ---------------------------------------------------------------------------------------------
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
    // slots [q_defer, nq): the length class whose longest members' k-mer order is still being computed on the side stream; their
    // frequency cap (and everything after it) waits for ev_korder, the classes before them do not
    u32 q_defer = 0;
    bool korder_async = false;
    DevBuf<u32> d_qid, d_ocnt, d_ostart;
    SeqSet dev;              // device arrays only (d_res = masked raw, d_scls, d_off, d_words, d_pseq)
    DevBuf<u32> qbucket, korder, sbeg, scnt, pcnt, eff, nz, hoff, cidx;
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
inline u8 query_class(u32 len) { return len < 512 ? 0 : len < 1024 ? 1 : len < 2048 ? 2 : len < 4096 ? 3 : 4; }

void prepare_batch(so_ctx* c, Batch& b, i64 q_lo, i64 q_hi) {
    if (b.korder_async) HIP_CHECK(hipStreamSynchronize(c->st_side));   // (a batch that never reached its long queries' passes)
    b.korder_async = false;
    b.q_lo = q_lo, b.q_hi = q_hi, b.nq = (u32)(q_hi - q_lo);
    const SeqSet& Q = c->qry;
    const bool classes_on = tune().qclass;   // SOHIT_QCLASS=0: file order
    const bool lay_cached = b.lay_gen == c->qry_gen && b.lay_lo == q_lo && b.lay_hi == q_hi && b.lay_classes == classes_on && (c->seg_on_device || !c->filter);
    if (!lay_cached) {
    b.lay_lo = -1;
    b.h_off.assign((size_t)b.nq + 1, 0);
    b.maxqlen = 0;
    b.qid.resize(b.nq), b.qcls.resize(b.nq);
    {
        u32 cnt[QCLASSES] = {0}, at[QCLASSES] = {0};
        for (u32 i = 0; i < b.nq; ++i) cnt[classes_on ? query_class(Q.len(q_lo + i)) : 0]++;
        for (int k = 1; k < QCLASSES; ++k) at[k] = at[k - 1] + cnt[k - 1];
        b.permuted = false;
        for (u32 i = 0; i < b.nq; ++i) {
            const u8 k = classes_on ? query_class(Q.len(q_lo + i)) : 0;
            b.permuted |= at[k] != i;
            b.qcls[at[k]] = k;
            b.qid[at[k]++] = i;
        }
    }
    for (u32 i = 0; i < b.nq; ++i) {
        u32 ln = Q.len(q_lo + b.qid[i]);
        b.h_off[i + 1] = b.h_off[i] + ln;
        b.maxqlen = std::max(b.maxqlen, ln);
    }
    }
    const u32* d_qid = nullptr;
    if (b.permuted) {
        b.d_qid.ensure((size_t)b.nq + 4);
        if (!lay_cached) HIP_CHECK(hipMemcpyAsync(b.d_qid.p, b.qid.data(), (size_t)b.nq * sizeof(u32), hipMemcpyHostToDevice, c->st));
        d_qid = b.d_qid.p;
    }
    const size_t nres_b = b.h_off[b.nq];
    c->masked_lo = q_lo;
    if (c->seg_on_device || !c->filter) {
        // residues never leave the device: SEG kernel (or plain copy) from the resident raw queries
        b.dev.d_res.ensure(nres_b + 64);
        b.dev.d_off.ensure((size_t)b.nq + 1);
        if (!lay_cached) HIP_CHECK(hipMemcpyAsync(b.dev.d_off.p, b.h_off.data(), ((size_t)b.nq + 1) * sizeof(u32), hipMemcpyHostToDevice, c->st));
        b.lay_gen = c->qry_gen, b.lay_lo = q_lo, b.lay_hi = q_hi, b.lay_classes = classes_on;
        if (c->filter) {
            c->d_segmask.ensure(nres_b + 64);
            // a class-ordered batch keeps its long queries at the end: the instances for them start there, and the one for the queries
            // above 4096 residues (one wave per query: 0.4 ms for a 30 000-residue giant) runs on the side stream beside the others
            u32 q_mid = 0, q_long = 0;
            bool ordered = true;
            for (u32 i = 1; i < b.nq && ordered; ++i) ordered = b.qcls[i] >= b.qcls[i - 1];
            if (ordered && tune().qclass) {
                while (q_mid < b.nq && b.qcls[q_mid] < 2) ++q_mid;      // classes 0, 1: below 1024 residues
                q_long = q_mid;
                while (q_long < b.nq && b.qcls[q_long] < 4) ++q_long;   // class 4: 4096 and more
            }
            const bool seg_aside = b.maxqlen > 4096 && tune().seg_aside;
            if (seg_aside) {
                HIP_CHECK(hipEventRecord(c->ev_side_go, c->st));   // (the batch's offsets are on their way)
                HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_side_go, 0));
            }
            launch_seg(c->qry.d_res.p, c->qry.d_off.p, (u32)q_lo, d_qid, b.nq, b.dev.d_off.p, c->d_symmap.p, c->d_upmap.p, c->d_segtab.p,
                       c->d_segmask.p, b.dev.d_res.p, b.maxqlen, q_mid, q_long, c->st, seg_aside ? c->st_side : c->st);
            if (seg_aside) {
                HIP_CHECK(hipEventRecord(c->ev_ug_done, c->st_side));
                HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_ug_done, 0));
            }
        } else if (b.permuted) {
            launch_gather_seqs(c->qry.d_res.p, c->qry.d_off.p, (u32)q_lo, d_qid, b.nq, b.dev.d_off.p, b.dev.d_res.p, c->st);
        } else {
            launch_copy_range(c->qry.d_res.p + Q.off[q_lo], b.dev.d_res.p, nres_b, c->st);
        }
        layout_set(c, b.dev, c->q_present, nres_b, b.nq);
        b.h_res.clear();
    } else {
        // more than 64 distinct residue bytes: SEG on the host (same arithmetic, same tables)
        b.h_res.resize(nres_b + 16);
        const u8* src = Q.res.data();
        parallel_for((i64)b.nq, [&](i64 i) { seg_mask(src + Q.off[q_lo + b.qid[i]], (int)Q.len(q_lo + b.qid[i]), b.h_res.data() + b.h_off[i]); });
        upload_set(c, b.dev, b.h_res.data(), b.h_off, b.nq);
    }
    const int AS = c->cfg.A * c->cfg.S;
    const u32 Ppad = b.dev.Ppad;
    const size_t T = (size_t)AS * Ppad;
    b.qbucket.ensure(T);
    launch_qhash(b.dev.d_words.p, Ppad, c->cfg, b.dev.lut, b.qbucket.p, c->st);
    const size_t nres = b.h_off[b.nq];
    b.korder.ensure(nres + 1);
    {
        // queries with more windows than the LDS sort holds use global scratch; in a class-ordered batch they are the tail
        u32 q_long = b.nq;
        for (u32 i = 0; i < b.nq; ++i)
            if ((i64)(b.h_off[i + 1] - b.h_off[i]) - c->cfg.mink + 1 > (i64)ksc_lds_max()) {
                q_long = i;
                break;
            }
        if (q_long < b.nq) b.gx.ensure(nres + 4), b.gL.ensure(nres + 4), b.gR.ensure(nres + 4);
        // in a class-ordered batch the long ones are the tail of the last class: their order is computed on the side stream
        // (SOHIT_KSC_ASYNC=0: on the batch's stream)
        const bool async_on = tune().ksc_async;
        bool ordered = true;
        for (u32 i = 1; i < b.nq && ordered; ++i) ordered = b.qcls[i] >= b.qcls[i - 1];
        b.korder_async = async_on && q_long < b.nq && ordered && b.qcls[q_long] != b.qcls[0];
        if (b.korder_async) {
            b.q_defer = q_long;
            while (b.q_defer > 0 && b.qcls[b.q_defer - 1] == b.qcls[q_long]) --b.q_defer;
            HIP_CHECK(hipEventRecord(c->ev_side_go, c->st));   // the batch's class arrays are on the device
            HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_side_go, 0));
        }
        launch_ksc_order(b.dev.d_scls.p, b.dev.d_off.p, b.nq, q_long, c->cfg.mink, c->d_b62c.p, b.gx.p, b.gL.p, b.gR.p, b.korder.p, c->st,
                         b.korder_async ? c->st_side : c->st);
        if (b.korder_async) HIP_CHECK(hipEventRecord(c->ev_korder, c->st_side));
    }
    b.sbeg.ensure(T), b.scnt.ensure(T), b.eff.ensure(T + 4), b.nz.ensure(T + 4), b.hoff.ensure(T + 4), b.cidx.ensure(T + 4);
    b.pcnt.ensure(Ppad), b.mark.ensure(Ppad);
    b.counters.ensure(8);
    b.ucount.ensure(4);
    HIP_CHECK(hipMemsetAsync(b.ucount.p, 0, 4 * sizeof(unsigned long long), c->st));
    c->d_scan_tmp.ensure(scan_u32_temp_elems(std::max<size_t>(T, (size_t)c->nc + 1)) + 8);
    // seed windows hashed (valid or not): one per (as, residue)
    c->cnt.seed_windows += (i64)AS * (i64)nres;
}

// pinned destination: a pageable 4-byte read costs ~30 us per sync through the staging path, a pinned one ~10
void* small_host(so_ctx* c) {
    if (!c->h_small) HIP_CHECK(hipHostMalloc((void**)&c->h_small, 1024, hipHostMallocDefault));
    return c->h_small;
}

u32 d2h_u32(so_ctx* c, const u32* p) {
    u32* v = (u32*)small_host(c);
    HIP_CHECK(hipMemcpyAsync(v, p, sizeof(u32), hipMemcpyDeviceToHost, c->st));
    HIP_CHECK(hipStreamSynchronize(c->st));
    return *v;
}

// Totals of two scans that share d_scan_tmp, fetched with ONE synchronisation: the first total is parked in a
// device word while the second scan runs.
void stash_u32(so_ctx* c, const u32* p, int slot) {
    c->d_small.ensure(16);
    HIP_CHECK(hipMemcpyAsync(c->d_small.p + slot, p, sizeof(u32), hipMemcpyDeviceToDevice, c->st));
}
void d2h_pair(so_ctx* c, const u32* second, u32& a, u32& b) {
    stash_u32(c, second, 1);
    u32* v = (u32*)small_host(c);
    HIP_CHECK(hipMemcpyAsync(v, c->d_small.p, 2 * sizeof(u32), hipMemcpyDeviceToHost, c->st));
    HIP_CHECK(hipStreamSynchronize(c->st));
    a = v[0], b = v[1];
}

void ensure_sort_tmp(so_ctx* c, size_t bytes) { c->d_sort_tmp.ensure(bytes + 256); }

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

// Compact index addends of chunk `ch` for a pass whose queries are shorter than 2^bp (see ChunkIndex::BandEnc, k_encode_band32):
// picks the diagonal width k, numbers the bands, encodes the entries.  multi_ok = a long subject may own several bands (the
// kernels resolve bands through a table: one alphabet x one pattern only); otherwise k is wide enough for the longest subject.
// Returns null when band + diagonal + tag bits exceed 31 (the pass then uses the 8-byte addends).
// diagonal bits k and number of bands of the chunk for queries below 2^bp residues (kept in the ctx: see band_plans)
void band_plan(so_ctx* c, ChunkIndex& ch, int bp, bool multi_ok, int* k_out, u64* nband_out) {
    for (const auto& pl : c->band_plans)
        if (pl.lo == ch.seq_lo && pl.hi == ch.seq_hi && pl.bp == bp && pl.multi_ok == multi_ok) {
            *k_out = pl.k, *nband_out = pl.nband;
            return;
        }
    const u32 nseq = (u32)(ch.seq_hi - ch.seq_lo);
    const u64 qcap = 1ull << bp;
    int kmax = bp + 1;
    while ((1ull << kmax) < (u64)ch.maxslen + qcap) ++kmax;   // every subject in one band
    auto bands_at = [&](int k) -> u64 {
        const u64 C = (1ull << k) - qcap;
        u64 n = 0;
        for (i64 j = ch.seq_lo; j < ch.seq_hi; ++j) {
            const u64 sl = c->ref.len(j);
            n += sl <= C ? 1ull : (sl + qcap + (1ull << k) - 1) >> k;
        }
        return n;
    };
    int k = kmax;
    u64 nband = nseq;
    if (multi_ok) {
        int best_bits = ceil_log2(std::max<u64>(nseq, 2)) + kmax;
        for (int kk = kmax - 1; kk > bp; --kk) {   // (ties keep the wider k: fewer subjects with several bands)
            const u64 nb = bands_at(kk);
            const int bits = ceil_log2(std::max<u64>(nb, 2)) + kk;
            if (bits < best_bits) best_bits = bits, k = kk, nband = nb;
        }
    }
    c->band_plans.push_back({ch.seq_lo, ch.seq_hi, bp, multi_ok, k, nband});
    *k_out = k, *nband_out = nband;
}

ChunkIndex::BandEnc* band_encoding(so_ctx* c, ChunkIndex& ch, int ba, int bp, bool multi_ok) {
    ++ch.enc_clock;
    for (auto& e : ch.encs)
        if (e->k >= 0 && e->ba == ba && e->bp == bp && e->multi_ok == multi_ok) {
            e->used = ch.enc_clock;
            return e->k == 0 ? nullptr : e.get();   // k == 0: "does not fit" remembered
        }
    const u32 nseq = (u32)(ch.seq_hi - ch.seq_lo);
    const u64 qcap = 1ull << bp;
    int k;
    u64 nband;
    band_plan(c, ch, bp, multi_ok, &k, &nband);
    // slot: a stale one, else a new one, else the least recently used
    ChunkIndex::BandEnc* e = nullptr;
    for (auto& x : ch.encs)
        if (x->k < 0) e = x.get();
    if (!e && ch.encs.size() < 10) {   // (one per query-position width in use: 7 ... 15 bits; fewer slots than widths and a step that cycles through them misses every time)
        ch.encs.push_back(std::make_unique<ChunkIndex::BandEnc>());
        e = ch.encs.back().get();
    }
    if (!e) {
        e = ch.encs[0].get();
        for (auto& x : ch.encs)
            if (x->used < e->used) e = x.get();
    }
    e->ba = ba, e->bp = bp, e->multi_ok = multi_ok, e->used = ch.enc_clock;
    if (ceil_log2(std::max<u64>(nband, 2)) + k + ba > 31) {
        e->k = 0;
        return nullptr;
    }
    e->k = k, e->nband = (u32)nband, e->C = (u32)((1ull << k) - qcap), e->multi = nband != nseq;
    std::vector<u32> gbase((size_t)nseq + 1);
    std::vector<u64> btab;
    if (e->multi) btab.resize((size_t)nband);
    u32 band = 0;
    for (u32 j = 0; j < nseq; ++j) {
        const u32 sl = c->ref.len(ch.seq_lo + j);
        const bool one = sl <= e->C;
        const u32 nb = one ? 1u : (u32)(((u64)sl + qcap + (1ull << k) - 1) >> k);
        gbase[j] = (band << k) + (one ? e->C : sl);
        if (e->multi)
            for (u32 t = 0; t < nb; ++t) btab[(size_t)band + t] = (u64)j | ((u64)gbase[j] << 32);
        band += nb;
    }
    e->gbase.ensure((size_t)nseq + 4);
    e->dk32.ensure((size_t)ch.E + 4);
    if (nseq) HIP_CHECK(hipMemcpyAsync(e->gbase.p, gbase.data(), (size_t)nseq * sizeof(u32), hipMemcpyHostToDevice, c->st));
    if (e->multi) {
        e->btab.ensure((size_t)nband + 4);
        HIP_CHECK(hipMemcpyAsync(e->btab.p, btab.data(), (size_t)nband * sizeof(u64), hipMemcpyHostToDevice, c->st));
    }
    launch_encode_band32(ch.entries.p, ch.E, ba, e->gbase.p, c->ref.d_off.p + ch.seq_lo, e->dk32.p, c->st);
    HIP_CHECK(hipStreamSynchronize(c->st));   // the host vectors must outlive the copies
    return e;
}

// seed stage of one (batch, chunk): candidates appended to the batch's candidate store
void seed_pass(so_ctx* c, Batch& b, int ci, u32 qa, u32 qb, double t0, StageClock& sc);

// bucket bounds of every query window in chunk ci, frequency cap, and the number of index entries each query of the
// batch will visit there (pinned host array, valid until the next call)
const unsigned long long* chunk_qhits(so_ctx* c, Batch& b, int ci) {
    ChunkIndex& ch = *c->chunks[ci];
    const int AS = c->cfg.A * c->cfg.S;
    const u32 Ppad = b.dev.Ppad, NC = (u32)c->nc;
    {
        ProfTimer pt(c, &c->cnt.bounds_ms, &c->cnt.bounds_launches);
        launch_bounds(b.qbucket.p, Ppad, AS, ch.hkey.p, ch.hval.p, ch.hshift, ch.hmask, ch.use_dir ? ch.dir.p : nullptr, ch.ubeg.p, NC, ch.E, b.sbeg.p,
                      b.scnt.p, b.pcnt.p, c->st);
        pt.stop();
        if (c->profile) c->cnt.bounds_bytes += (i64)8 * AS * (i64)b.h_off[b.nq];
    }
    HIP_CHECK(hipMemsetAsync(b.mark.p, 0, Ppad, c->st));
    i64 threshold = ch.threshold;
    if (c->thr >= 1 || threshold == 0) threshold = c->thr;  // `thr < 1 and DB.threshold or thr`, fsearch.py:2992
    b.qhits.ensure((size_t)b.nq + 2);
    const u32 n1 = b.korder_async ? b.q_defer : b.nq;   // (the last length class follows in chunk_qhits_deferred)
    launch_cap(b.korder.p, b.dev.d_off.p, 0, n1, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, c->st);
    if (c->h_qhits_cap < b.nq) {  // pinned: a pageable read of this array costs more than the kernels around it
        if (c->h_qhits) (void)hipHostFree(c->h_qhits);
        c->h_qhits_cap = (size_t)b.nq + 1024;
        HIP_CHECK(hipHostMalloc((void**)&c->h_qhits, c->h_qhits_cap * sizeof(unsigned long long), hipHostMallocDefault));
    }
    unsigned long long* qh = c->h_qhits;
    HIP_CHECK(hipMemcpyAsync(qh, b.qhits.p, (size_t)n1 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->st));
    HIP_CHECK(hipStreamSynchronize(c->st));
    return qh;
}

// ... and of the length class that waited for the side stream's k-mer orders (once per batch: later chunks find them done)
void chunk_qhits_deferred(so_ctx* c, Batch& b, int ci) {
    ChunkIndex& ch = *c->chunks[ci];
    HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_korder, 0));
    b.korder_async = false;
    i64 threshold = ch.threshold;
    if (c->thr >= 1 || threshold == 0) threshold = c->thr;
    launch_cap(b.korder.p, b.dev.d_off.p, b.q_defer, b.nq, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, c->st);
    HIP_CHECK(hipMemcpyAsync(c->h_qhits + b.q_defer, b.qhits.p + b.q_defer, (size_t)(b.nq - b.q_defer) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->st));
    HIP_CHECK(hipStreamSynchronize(c->st));
}

// Would a pass of `nq` queries no longer than `maxq` with `hits` seed hits in this chunk take the sorted path?  The tests of
// group_bucketed (seed_pass) on a whole length class: no compact banded addends for that query width, several seed patterns, or fewer
// hits per (query, widest subject range) than the sparse limit.
bool class_takes_sorted_path(so_ctx* c, ChunkIndex& ch, u32 maxq, unsigned long long hits, unsigned long long nq) {
    const int AS = c->cfg.A * c->cfg.S;
    if (!tune().bucket || AS != 1 || UG_SHARDS != 1 || !nq) return true;
    if (tune().lk_wide) return true;
    const int bp = ceil_log2(std::max<u32>(maxq, 2));
    const bool bands_ok = tune().bands;
    int k;
    u64 nband;
    band_plan(c, ch, bp, bands_ok, &k, &nband);   // (the layout only: no encoding is built for the question)
    const int bs = ceil_log2(std::max<u64>(nband, 2));
    if (bs + k > 31 || bp > 16) return true;
    const int wb_hi = std::min(std::min(31 - k - bp, bs), bkt_max_wb());
    int wb_lo = 0;
    while ((nband + (1ull << wb_lo) - 1) >> wb_lo > BKT_RMAX) ++wb_lo;
    if (wb_hi < wb_lo) return true;
    const unsigned long long sparse = (unsigned long long)std::max(0ll, tune().bucket_min);
    return hits / (nq * ((nband + (1ull << wb_hi) - 1) >> wb_hi)) < sparse;
}

void seed_stage(so_ctx* c, Batch& b, int ci) {
    ChunkIndex& ch = *c->chunks[ci];
    const u32 nseq_chunk = (u32)(ch.seq_hi - ch.seq_lo);
    b.chunk_base.push_back(b.chunk_base.empty() ? 0u : b.chunk_base.back());
    if (nseq_chunk == 0 || ch.E == 0 || b.nq == 0) return;
    StageClock sc(c);
    // Split the batch into query sub-ranges whose seed hits fit the per-pass budget (keys, sort
    // scratch and group arrays are sized by it; 32-bit hit ordinals need < 2^32 per pass).
    const unsigned long long* qh = chunk_qhits(c, b, ci);
    const unsigned long long budget = c->max_hits_per_pass;
    // Pass groups.  A pass holds one length class so that its key fields are as narrow as its queries allow -- what the bucketed
    // binning needs.  A class whose pass would take the sorted path anyway (class_takes_sorted_path: typically too few hits per query
    // and subject range, the long seeds) gains nothing from a pass of its own and pays a dozen launches and three host round trips for it:
    // neighbouring sparse classes are searched as ONE pass (heterogeneous 100 k set, headline seed: ten passes per step -> four).
    // The class still waiting for its k-mer orders (korder_async) is never merged into an earlier pass.
    int grp[QCLASSES];
    {
        const bool merge_on = tune().pass_merge;
        unsigned long long hits[QCLASSES] = {0}, cnt[QCLASSES] = {0};
        u32 maxq[QCLASSES] = {0};
        const u32 known = b.korder_async ? b.q_defer : b.nq;   // (the deferred class's counts arrive later)
        for (u32 q = 0; q < known; ++q) {
            const int k = b.qcls[q];
            hits[k] += qh[q], cnt[k]++, maxq[k] = std::max(maxq[k], b.h_off[q + 1] - b.h_off[q]);
        }
        const int held = b.korder_async ? (int)b.qcls[b.q_defer] : -1;
        bool sp[QCLASSES];
        for (int k = 0; k < QCLASSES; ++k) sp[k] = merge_on && k != held && cnt[k] && class_takes_sorted_path(c, ch, maxq[k], hits[k], cnt[k]);
        grp[0] = 0;
        for (int k = 1; k < QCLASSES; ++k) {
            // (an empty class between two sparse ones does not separate them)
            int j = k - 1;
            while (j > 0 && !cnt[j] && j != held) --j;
            grp[k] = grp[k - 1] + ((sp[k] && sp[j] && grp[j] == grp[k - 1]) || (!cnt[k] && k != held) ? 0 : 1);
        }
    }
    u32 qa = 0;
    while (qa < b.nq) {
        if (b.korder_async && qa >= b.q_defer) chunk_qhits_deferred(c, b, ci);
        unsigned long long acc = 0;
        u32 qb = qa;
        while (qb < b.nq && (qb == qa || (acc + qh[qb] <= budget && grp[b.qcls[qb]] == grp[b.qcls[qa]]))) acc += qh[qb++];
        if (acc >= 0xFFFFFFF0ull) throw SoError("a single query visits >= 2^32 index entries in one chunk: lower -c");
        if (acc) seed_pass(c, b, ci, qa, qb, wall(), sc), ++c->cnt.seed_passes;
        if (tune().debug) fprintf(stderr, "[sohit] chunk %d pass queries [%u, %u) classes %d..%d hits %llu\n", ci, qa, qb, (int)b.qcls[qa], (int)b.qcls[qb - 1], acc);
        qa = qb;
    }
}

// one hit-budgeted pass of the seed stage: batch queries [qa, qb) = packed positions [p_lo, p_hi)
void seed_pass(so_ctx* c, Batch& b, int ci, u32 qa, u32 qb, double t0, StageClock& sc) {
    ChunkIndex& ch = *c->chunks[ci];
    const u32 p_lo = b.h_off[qa] + qa, p_hi = b.h_off[qb] + qb;
    const int AS = c->cfg.A * c->cfg.S;
    const u32 nseq_chunk = (u32)(ch.seq_hi - ch.seq_lo);
    u32* qcnt = b.ccnt.p + (size_t)ci * b.nq;
    // hit counts, their scan (hit ordinals) and the seed compaction run over the pass's own seed slots [t_lo, t_lo + Tp)
    const size_t t_lo = (size_t)AS * p_lo, Tp = (size_t)AS * (p_hi - p_lo);
    launch_effcnt(b.mark.p, b.scnt.p, AS, p_lo, p_hi, b.eff.p, b.nz.p, c->st);
    const u32* dH = scan_u32(b.eff.p + t_lo, b.hoff.p + t_lo, Tp, false, c->d_scan_tmp.p, c->st);
    stash_u32(c, dH, 0);  // the scan's total lives in d_scan_tmp: park it before the next scan
    const u32* dK = scan_u32(b.nz.p + t_lo, b.cidx.p + t_lo, Tp, false, c->d_scan_tmp.p, c->st);
    u32 H, K;
    d2h_pair(c, dK, H, K);
    sc.lap("seed.bounds_cap_scan");
    c->cnt.seed_hits += H;
    if (H == 0 || K == 0) {
        c->cnt.seed_ms += (wall() - t0) * 1e3;
        return;
    }
    // key layout.  Query-position bits follow the PASS's longest query (passes hold one length class, seed_stage); subject and
    // diagonal bits come from the compact banded addends when they fit (band_encoding), else from the chunk's longest sequence.
    u32 pmaxq = 0;
    for (u32 q = qa; q < qb; ++q) pmaxq = std::max(pmaxq, b.h_off[q + 1] - b.h_off[q]);
    KeyLayout kl;
    kl.bq = ceil_log2((u64)b.nq + 1);
    kl.bp = ceil_log2(std::max<u32>(pmaxq, 2));
    kl.ba = AS > 1 ? ceil_log2((u64)AS) : 0;
    const bool force_wide = tune().lk_wide;
    const bool bands_ok = tune().bands;   // SOHIT_BANDS=0: one band per subject whatever its length
    ChunkIndex::BandEnc* enc = force_wide ? nullptr : band_encoding(c, ch, kl.ba, kl.bp, AS == 1 && bands_ok);
    const bool compact = enc != nullptr;
    const u32 nunit = compact ? enc->nband : nseq_chunk;   // what the key's subject field counts
    if (compact) {
        kl.bs = ceil_log2(std::max<u32>(enc->nband, 2));
        kl.bd = enc->k;
        kl.diag_off = enc->C;
    } else {
        kl.bs = ceil_log2(nseq_chunk);
        kl.bd = ceil_log2((u64)pmaxq + ch.maxslen + 1);
        kl.diag_off = ch.maxslen;
    }
    kl.finish();
    // layout of the pass records: chunk sequences and plain diagonals
    KeyLayout klr = kl;
    const void* btab = nullptr;
    if (compact && enc->multi) {
        klr.bs = ceil_log2(nseq_chunk);
        klr.bd = ceil_log2((u64)pmaxq + ch.maxslen + 1);
        klr.diag_off = ch.maxslen;
        klr.finish();
        btab = enc->btab.p;
    }
    // k_ungap's GALLOP variant from this query length on (SOHIT_UG_GALLOP: tuning switch)
    const u32 gallop_min = (u32)tune().ug_gallop;
    const int bsp = ceil_log2((u64)ch.maxslen + 1);
    const int ft_bits_entry = (klr.bs + 1) + klr.ba + bsp;
    const bool ft_walk = AS > 1;  // several (alphabet, pattern) combinations: a group's first-touch key needs all its hits
    if (kl.total > 64) throw SoError("sort key needs " + std::to_string(kl.total) + " bits (> 64): lower SOHIT_BATCH or -c");
    if (klr.sh_subj + klr.bs > 64) throw SoError("pass-record key exceeds 64 bits: sequences too long for this build");
    if (kl.ba + kl.bp + ft_bits_entry > 64) throw SoError("first-touch key exceeds 64 bits: sequences too long for this build");
    b.cs_hoff.ensure((size_t)K + 2), b.cs_beg.ensure((size_t)K + 2), b.cs_kbase.ensure((size_t)K + 2);
    launch_compact_seeds(b.eff.p, b.hoff.p, b.cidx.p, b.sbeg.p, b.dev.d_pseq.p, b.dev.d_off.p, p_lo, p_hi, AS, kl, b.cs_hoff.p, b.cs_beg.p,
                         b.cs_kbase.p, c->st);
    // per-entry key addends: the 4-byte compact form whenever the fields fit, else 8-byte ones for this layout
    const u32* dk32 = compact ? enc->dk32.p : nullptr;
    if (!compact && (ch.d_sh_subj != kl.sh_subj || ch.d_sh_diag != kl.sh_diag)) {
        ch.dkeys.ensure((size_t)ch.E + 2);
        launch_encode_delta(ch.entries.p, ch.E, kl.sh_subj, kl.sh_diag, ch.maxslen, ch.dkeys.p, c->st);
        ch.d_sh_subj = kl.sh_subj, ch.d_sh_diag = kl.sh_diag;
    }
    // pass-list buffers (k_ungap / k_bkt_ungap append the groups that reach MIN_UNGAP)
    const u32 shard_cap = ungap_shard_cap(H);
    const size_t pcap = (size_t)shard_cap * UG_SHARDS + 2 + ungap1_list_slack(c->ncu);   // (+ the unused slots of k_ungap1's reserved pieces)
    b.p_qs.ensure(pcap), b.p_sd.ensure(pcap), b.p_ft.ensure(pcap);
    b.shard.ensure(2 * UG_SHARDS + 8);
    b.stepshard.ensure(UG_SHARDS + 4);   // group counts, then (Tune::count_steps) b62 lookups, singleton groups, chained groups
    b.bflag.ensure(8);
    double t1 = wall();
    bool bbest = false;   // set by group_bucketed: pass records were flushed per bucket, k_bkt_best reduces them
    BktLayout bL;
    u32 bnb = 0;
    memset(&bL, 0, sizeof bL);
    unsigned long long* const ugstat = c->tune.count_steps ? b.stepshard.p + UG_SHARDS : nullptr;   // the counting instances of the extension kernels
    auto reset_pass_lists = [&] {
        HIP_CHECK(hipMemsetAsync(b.shard.p, 0, (2 * UG_SHARDS + 8) * sizeof(u32), c->st));
        HIP_CHECK(hipMemsetAsync(b.stepshard.p, 0, (UG_SHARDS + 4) * sizeof(unsigned long long), c->st));
        HIP_CHECK(hipMemsetAsync(b.bflag.p, 0, 8 * sizeof(u32), c->st));
    };

    // ---- diagonal binning, bucketed (k_bucket.hip): no sort; count -> scan -> scatter (4 B per hit) -> LDS hash grouping ----
    // One alphabet x one pattern, compact index addends and wb + bd + bp <= 31 (wb counts diagonal bands: one per chunk sequence,
    // several for the few sequences too long for one); returns false when the pass has to take the sorted path below.
    auto group_bucketed = [&]() -> bool {
        const bool enabled = tune().bucket;
        if (!enabled || AS != 1 || !compact || UG_SHARDS != 1) return false;
        const u32 nqp = qb - qa;
        const int wb_hi = std::min(std::min(31 - kl.bd - kl.bp, kl.bs), bkt_max_wb());   // hit word < 2^31; subjects per range <= the sort's bins
        int wb_lo = 0;
        while (((u64)nunit + (1ull << wb_lo) - 1) >> wb_lo > BKT_RMAX) ++wb_lo;
        if (wb_hi < wb_lo || kl.bp > 16) return false;
        // widest subject range whose average bucket is a few hits per thread of the workgroup that groups it
        const u32 target = (u32)std::max(1ll, tune().bucket_avg);
        int wb = wb_hi;
        // sparse passes (long seeds: a few hundred hits per query) would leave the grouping kernel walking mostly empty
        // buckets: the sorted path handles those well (its segments are short)
        const u32 sparse = (u32)std::max(0ll, tune().bucket_min);
        if ((u64)H / ((u64)nqp * (((u64)nunit + (1ull << wb) - 1) >> wb)) < sparse) return false;
        while (wb > wb_lo && (u64)H / ((u64)nqp * (((u64)nunit + (1ull << wb) - 1) >> wb)) > target) --wb;
        BktLayout L;
        L.wb = wb, L.bd = kl.bd, L.bp = kl.bp, L.sh_q = kl.sh_q, L.sh_qpos = kl.sh_qpos, L.nqp = nqp, L.qa = qa;
        L.R = (u32)(((u64)nunit + (1ull << wb) - 1) >> wb);
        L.maxslen = ch.maxslen;
        if ((u64)L.R * nqp >= 0xFFFFFFF0ull) return false;
        const u32 nb = L.R * nqp;
        // tiles: <= 1024 consecutive hit ordinals of one query
        b.qseg.ensure((size_t)b.nq + 4);
        launch_query_segments(b.hoff.p, b.dev.d_off.p, qa, qb, AS, H, b.qseg.p, c->st);
        const u32* qseg = b.qseg.p + qa;
        b.bt0.ensure((size_t)nqp + 4);
        launch_bkt_ntiles(qseg, nqp, b.bt0.p, c->st);
        c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)nqp + 1) + 8);
        const u32 NT = d2h_u32(c, scan_u32(b.bt0.p, b.bt0.p, (size_t)nqp + 1, false, c->d_scan_tmp.p, c->st));
        if ((u64)L.R * NT >= 0xFFFFFFF0ull) return false;
        b.btd.ensure(4 * (size_t)NT + 8);
        launch_bkt_tiledesc(qseg, b.bt0.p, nqp, NT, b.cs_hoff.p, K, b.btd.p, c->st);
        // count per (tile, range), stored tile-major; its exclusive scan in range-major order is the scatter plan
        const size_t nm = (size_t)L.R * NT;
        b.bmat.ensure(nm + 4);
        {
            ProfTimer pt(c, &c->cnt.count_ms, &c->cnt.count_launches);
            launch_bkt_pass(false, b.btd.p, qseg, NT, b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, dk32, c->ref.d_off.p + ch.seq_lo, L, b.bmat.p,
                            nullptr, c->st);
            pt.stop();
        }
        const size_t npart = (size_t)L.R * bkt_scan_blocks(NT);
        b.bpart.ensure(npart + 4);
        c->d_scan_tmp.ensure(scan_u32_temp_elems(npart + 1) + 8);
        launch_bkt_colsum(b.bmat.p, NT, L.R, b.bpart.p, c->st);
        const u32* dHv = scan_u32(b.bpart.p, b.bpart.p, npart, false, c->d_scan_tmp.p, c->st);
        stash_u32(c, dHv, 2);   // the total lives in d_scan_tmp: park it (k_bkt_extents reads it after later scans)
        launch_bkt_colscan(b.bmat.p, NT, L.R, b.bpart.p, c->st);
        const u32 Hv = d2h_u32(c, c->d_small.p + 2);  // hits kept (all but the dropped offset-0 ones)
        sc.lap("seed.bucket_count");
        b.hits32.ensure((size_t)H + 2);
        {
            ProfTimer pt(c, &c->cnt.lookup_ms, &c->cnt.lookup_launches);
            launch_bkt_pass(true, b.btd.p, qseg, NT, b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, dk32, c->ref.d_off.p + ch.seq_lo, L, b.bmat.p,
                            b.hits32.p, c->st);
            pt.stop();
            if (c->profile) c->cnt.lookup_bytes += (i64)8 * (i64)H;
        }
        t1 = wall();
        sc.lap("seed.bucket_scatter");
        // the grouped hits leave as the buckets' own 32-bit words (k_ungap's W32 input) unless SOHIT_UG_W32=0 asks for the 64-bit keys
        const bool w32 = tune().ug_w32;
        if (w32) b.hits32s.ensure((size_t)H + 8);   // (k_ungap1's chains read four words ahead)
        else b.keys2.ensure((size_t)H + 2);
        b.bext.ensure((size_t)nb + 4);
        launch_bkt_extents(b.bmat.p, b.bt0.p, NT, L.R, nqp, nb, c->d_small.p + 2, b.bext.p, c->st);
        {
            ProfTimer pt(c, &c->cnt.bgroup_ms, &c->cnt.bgroup_launches);
            launch_bkt_group(b.hits32.p, b.bext.p, nb, L, kl, w32 ? nullptr : b.keys2.p, w32 ? b.hits32s.p : nullptr, b.bflag.p, c->st);
            pt.stop();
        }
        // a group too large for a wave's LDS table (or pool) leaves key slots unwritten: never walk them -- sorted path instead
        const u32 refused = d2h_u32(c, b.bflag.p);
        sc.lap("group.bucket_group");
        if (refused) {
            if (tune().debug) fprintf(stderr, "[sohit] bucketed pass refused (flag %u): sorted path\n", refused);
            return false;
        }
        if (tune().debug) fprintf(stderr, "[sohit] bucketed pass: wb %d ranges %u buckets %u tiles %u hits %u of %u\n", wb, L.R, nb, NT, Hv, H);
        // best diagonal per subject bucket by bucket (k_bkt_best) when first-touch keys fit its 44-bit field; else the sorted path below
        // ... and a chained ungapped score fits the 20 bits k_bkt_best packs above them (at most 11 per residue of the shorter sequence)
        bbest = tune().bucket_best && !ft_walk && (kl.ba + kl.bp + ft_bits_entry <= 44) &&
                (u64)std::min<u32>(pmaxq, ch.maxslen) * 11ull < (1ull << 20);
        // singleton groups (84 % of a dense pass's groups) go to k_ungap1, the chains of two and more seeds stay with k_ungap
        const bool ug1 = w32 && c->tune.ug1 && pmaxq <= ungap1_qcap() && bbest;   // (its pass list has unused slots: only the bucketed reduction skips them)
        const bool side = ug1 && c->tune.ug1_overlap;
        hipStream_t ust = side ? c->st_ug : c->st;
        const bool ug2 = ug1 && c->tune.ug1_chain;   // ... and the longer groups to k_ungap2, through the list of their heads k_ungap1 writes
        if (ug1 && !c->ref.ug_valid) {
            const size_t nres = c->ref.res.size();
            c->ref.d_ug_store.ensure(nres + 2 * (size_t)U1_UG_PAD);
            launch_make_ug(c->ref.d_scls.p, c->ref.d_off.p, (u32)c->ref.N, nres, 8u, c->ref.d_ug_store.p + U1_UG_PAD, c->st);
            c->ref.ug_valid = true;
        }
        if (ug2) {
            if (!b.dev.ug_valid) {
                const size_t nres = b.h_off[b.nq];
                b.dev.d_ug_store.ensure(nres + 2 * (size_t)U1_UG_PAD);
                launch_make_ug(b.dev.d_scls.p, b.dev.d_off.p, b.nq, nres, 1u, b.dev.d_ug_store.p + U1_UG_PAD, c->st);
                b.dev.ug_valid = true;
            }
            b.mlist.ensure(ungap1_mlist_cap(Hv, c->ncu));
        }
        if (side) {   // the chains' kernel on a second stream, beside k_ungap1 (both append to the pass list)
            HIP_CHECK(hipEventRecord(c->ev_ug_go, c->st));
            HIP_CHECK(hipStreamWaitEvent(c->st_ug, c->ev_ug_go, 0));
        }
        if (ug1)
            launch_ungap1(c->ncu, (int)c->tune.ug1_variant, pmaxq, b.hits32s.p, b.bext.p, nb, L, kl, klr, btab, (u32)std::max(1ll, c->tune.ug1_wait), b.dev.d_scls.p, b.dev.d_off.p,
                          c->ref.d_ug_store.p + U1_UG_PAD, c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.bflag.p + 1, b.shard.p, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p,
                          ug2 ? b.mlist.p : nullptr, b.bflag.p + 3, ugstat, c->st);
        if (ug2)
            launch_ungap2(c->ncu, b.mlist.p, b.bflag.p + 3, b.hits32s.p, b.bext.p, L, kl, klr, btab, (u32)std::max(1ll, c->tune.ug1_wait), b.dev.d_ug_store.p + U1_UG_PAD, b.dev.d_off.p,
                          c->ref.d_ug_store.p + U1_UG_PAD, c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.bflag.p + 2, b.shard.p, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p,
                          ugstat, c->st);
        else
            launch_ungap(w32 ? nullptr : b.keys2.p, Hv, kl, klr, btab, pmaxq >= gallop_min, ft_walk, b.dev.d_scls.p, b.dev.d_off.p, c->ref.d_scls4.p,
                         c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.shard.p, shard_cap, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p, ust, w32 ? b.hits32s.p : nullptr,
                         b.bext.p, nb, &L, ug1, ugstat);
        if (side) {
            HIP_CHECK(hipEventRecord(c->ev_ug_done, c->st_ug));
            HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_ug_done, 0));
        }
        // the pass records are binned by (query, range of 2^wb chunk SEQUENCES): the same layout unless bands and sequences differ
        bL = L;
        bL.R = (u32)(((u64)nseq_chunk + (1ull << wb) - 1) >> wb);
        bnb = bL.R * nqp;
        c->cnt.hits_bucketed += H;
        return true;
    };

    // ---- the sorted path: 8-byte keys, segmented radix sort by (subject, diagonal), group walk over the sorted keys ----
    auto group_sorted = [&] {
    b.blk_first.ensure((size_t)lookup_num_blocks(H) + 2);
    launch_lookup_blockfirst(b.cs_hoff.p, K, H, b.blk_first.p, c->st);
    b.keys.ensure((size_t)H + 2), b.keys2.ensure((size_t)H + 2);
    {
        ProfTimer pt(c, &c->cnt.lookup_ms, &c->cnt.lookup_launches);
        launch_lookup(b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, b.blk_first.p, K, H, compact ? (const void*)dk32 : (const void*)ch.dkeys.p,
                      compact, c->ref.d_off.p + ch.seq_lo, kl, ch.maxslen, b.keys.p, c->st);
        pt.stop();
        if (c->profile) c->cnt.lookup_bytes += (i64)8 * (i64)H;
    }
    t1 = wall();
    sc.lap("seed.compact_lookup");
    // Hits are generated in (query, qpos, as) order (position-major seed ordinals) and the radix sorts are
    // stable, so only the (subject, diagonal) bits need sorting, inside each query's segment: 2 radix passes
    // fewer than a device-wide sort of the (query, subject, diagonal) bits.  One block sorts one segment, so
    // passes with few queries (huge per-query hit lists) use the device-wide sort instead.
    // (Dropped hits carry ~0 and sort last in their segment.)
    const int seg_mode = (int)tune().segsort;
    const u32 nseg = qb - qa;
    if (seg_mode && nseg >= 256) {
        b.qseg.ensure((size_t)b.nq + 4);
        launch_query_segments(b.hoff.p, b.dev.d_off.p, qa, qb, AS, H, b.qseg.p, c->st);
        ensure_sort_tmp(c, sort_keys_u64_seg_temp_bytes(H, nseg, kl.sh_diag, kl.sh_q));
        sort_keys_u64_seg(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.keys.p, b.keys2.p, H, nseg, b.qseg.p + qa, kl.sh_diag, kl.sh_q, c->st);
    } else {
    ensure_sort_tmp(c, sort_keys_u64_temp_bytes(H, kl.total));
    sort_keys_u64(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.keys.p, b.keys2.p, H, kl.sh_diag, kl.total, c->st);
    }
    sc.lap("group.sort_keys");
    // group walk + chained ungapped extension (the kernel finds the group heads itself)
    launch_ungap(b.keys2.p, H, kl, klr, btab, pmaxq >= gallop_min, ft_walk, b.dev.d_scls.p, b.dev.d_off.p, c->ref.d_scls4.p, c->ref.d_off.p + ch.seq_lo,
                 c->d_b62c.p, b.shard.p, shard_cap, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p, c->st, nullptr, nullptr, 0, nullptr, false, ugstat);
    };

    const bool lk_ablation = tune().lk_variant == 1 || tune().lk_variant == 2;
    reset_pass_lists();
    if (lk_ablation || !group_bucketed()) {
        bbest = false;
        group_sorted();
    }
    if (lk_ablation) return;  // ablation runs time the lookup only: keys are not valid
    u64* c_ftp = nullptr;   // candidates of the pass: first-touch key, query, [subject, score, qi, qj]
    u32 *c_qp = nullptr, *c_recp = nullptr;
    u32 NS = 0, maxseg = 0xFFFFFFFFu;  // candidates of the pass; the longest per-query segment (sparse path only)
    int cand_idx_bits = 0, cand_ftw = 0;   // > 0: k_bkt_best wrote sort words (first-touch word << idx_bits | position in the query's segment)
    for (;;) {
        // contiguous pass list; the group counters and the pass total come back in one synchronisation
        u32* shard_off = b.shard.p + UG_SHARDS;
        launch_shard_scan(b.shard.p, shard_off, c->st);
        u32 NP;
        {
            static_assert((UG_SHARDS + 4) * sizeof(unsigned long long) + sizeof(u32) <= 1024, "h_small too small");
            unsigned long long* gc = (unsigned long long*)small_host(c);
            u32* np = (u32*)(gc + UG_SHARDS + 4);
            HIP_CHECK(hipMemcpyAsync(gc, b.stepshard.p, (UG_SHARDS + 4) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipMemcpyAsync(np, shard_off + UG_SHARDS, sizeof(u32), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipStreamSynchronize(c->st));
            for (int k = 0; k < UG_SHARDS; ++k) c->cnt.groups += (i64)gc[k];
            c->cnt.ungap_steps += (i64)gc[UG_SHARDS], c->cnt.groups_single += (i64)gc[UG_SHARDS + 1], c->cnt.groups_chain += (i64)gc[UG_SHARDS + 2];
            NP = *np;
        }
        sc.lap("group.ungap");
        const u64 *q_qs = b.p_qs.p, *q_sd = b.p_sd.p, *q_ft = b.p_ft.p;  // one region: already contiguous
        if (NP && UG_SHARDS > 1) {
            b.q_qs.ensure((size_t)NP + 2), b.q_sd.ensure((size_t)NP + 2), b.q_ft.ensure((size_t)NP + 2);
            launch_compact_shards(b.shard.p, shard_off, shard_cap, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.q_qs.p, b.q_sd.p, b.q_ft.p, c->st);
            q_qs = b.q_qs.p, q_sd = b.q_sd.p, q_ft = b.q_ft.p;
        }
        if (bbest && NP) {
            // pass records -> the hit buckets, then one LDS reduction per bucket (k_bucket.hip): no sort of the records
            static_assert(UG_SHARDS == 1, "the bucketed best-diagonal path reads one contiguous pass list");
            b.bcnt.ensure((size_t)bnb + 2), b.bccnt.ensure((size_t)bnb + 2), b.pidx.ensure((size_t)NP + 2);
            b.q_qs.ensure((size_t)NP + 2), b.q_sd.ensure((size_t)NP + 2), b.q_ft.ensure((size_t)NP + 2);
            c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)bnb + 2) + 8);
            HIP_CHECK(hipMemsetAsync(b.bcnt.p, 0, ((size_t)bnb + 2) * sizeof(u32), c->st));
            launch_rec_count(b.p_qs.p, NP, klr.bs, bL, b.bcnt.p, b.pidx.p, c->st);
            scan_u32(b.bcnt.p, b.bcnt.p, (size_t)bnb + 1, false, c->d_scan_tmp.p, c->st);
            launch_rec_scatter(b.p_qs.p, b.p_sd.p, b.p_ft.p, b.pidx.p, NP, klr, bL, ft_bits_entry, bsp, c->ref.d_off.p + ch.seq_lo, b.bcnt.p, b.q_qs.p,
                               b.q_sd.p, b.q_ft.p, c->st);
            HIP_CHECK(hipMemsetAsync(b.bccnt.p + bnb, 0, 2 * sizeof(u32), c->st));
            launch_bkt_best(false, b.q_qs.p, b.q_sd.p, b.q_ft.p, b.bcnt.p, bnb, bL, klr.bs, (u32)ch.seq_lo, b.bccnt.p, nullptr, nullptr, nullptr, bsp, 0,
                            c->st);
            NS = d2h_u32(c, scan_u32(b.bccnt.p, b.bccnt.p, (size_t)bnb + 1, false, c->d_scan_tmp.p, c->st));
            if (tune().debug)
                fprintf(stderr, "[sohit] seed pass: queries %u..%u hits %u seeds %u pass records %u candidates %u (bucketed best)\n", qa, qb, H, K, NP, NS);
            b.c_ft.ensure((size_t)NS + 2), b.c_q.ensure((size_t)NS + 2), b.c_rec.ensure(4 * (size_t)NS + 8);
            c_ftp = b.c_ft.p, c_qp = b.c_q.p, c_recp = b.c_rec.p;
            {
                // candidate order as a keys-only segmented sort: the sort word = first-touch word << idx_bits | position inside the query's
                // segment (written by k_bkt_best itself) when both fit 63 bits (bit 63 stays free: see sort_cand_keys_seg) -- no index
                // array, no key-build pass, 8 instead of 12 bytes per candidate and radix pass (SOHIT_CAND_KEYS=0: the pairs sort)
                const bool cand_keys = tune().cand_keys;
                const bool cand_seg0 = tune().cand_segsort;
                const int ftw = kl.ba + kl.bp + ft_bits_entry - bsp + 1;
                if (cand_keys && cand_seg0 && bL.nqp >= 256 && ftw < 63 && 63 - ftw >= klr.bs + 1) cand_idx_bits = 63 - ftw, cand_ftw = ftw;
            }
            launch_bkt_best(true, b.q_qs.p, b.q_sd.p, b.q_ft.p, b.bcnt.p, bnb, bL, klr.bs, (u32)ch.seq_lo, b.bccnt.p, c_ftp, c_qp, c_recp, bsp,
                            cand_idx_bits, c->st);
            break;
        }
        // first-touch keys of the passing groups (k_ungap left the head hit's key / position in the third array)
        if (NP) launch_first_touch(ft_walk, b.keys2.p, H, klr, ft_bits_entry, bsp, c->ref.d_off.p + ch.seq_lo, const_cast<u64*>(q_ft), NP, c->st);
        if (NP == 0) break;
        // Sparse pass (tens of records per query): best diagonal per subject and the candidate order by one wave per query, in LDS -- no
        // sort of the records (k_q_best, k_group.hip).  A query with more records than the LDS instance holds sends the pass down the
        // sorting path below (SOHIT_QBEST=0: always).
        // (not for a pass that holds queries of 4096 residues and more: a giant brings more records than the largest instance sorts, and
        // one such query sends the whole pass down the old path after the new one has run)
        if (tune().qbest && pmaxq < 4096 && (kl.ba + kl.bp + ft_bits_entry) - bsp + 1 <= cand_order_lds_key_bits() && qb > qa) {
            const u32 nqp = qb - qa;
            BktLayout Lq{};
            Lq.wb = 31, Lq.nqp = nqp, Lq.qa = qa, Lq.R = 1;   // one "bucket" per query: k_rec_count's returning atomic is the record's rank in it
            b.bcnt.ensure((size_t)nqp + 2), b.bccnt.ensure((size_t)nqp + 2), b.pidx.ensure((size_t)NP + 2), b.pidx2.ensure((size_t)NP + 2);
            b.order.ensure((size_t)NP + 2), b.order2.ensure((size_t)NP + 2), b.p_qs2.ensure((size_t)NP + 2), b.tmp64.ensure((size_t)NP + 2);
            b.c_rec.ensure(4 * (size_t)NP + 8), b.bflag.ensure(4);
            c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)nqp + 2) + 8);
            HIP_CHECK(hipMemsetAsync(b.bcnt.p, 0, ((size_t)nqp + 2) * sizeof(u32), c->st));
            launch_rec_count(q_qs, NP, klr.bs, Lq, b.bcnt.p, b.pidx.p, c->st);
            scan_u32(b.bcnt.p, b.bcnt.p, (size_t)nqp + 1, false, c->d_scan_tmp.p, c->st);
            launch_qrec_scatter(q_qs, q_sd, q_ft, b.pidx.p, NP, klr.bs, qa, b.bcnt.p, b.pidx2.p /*query*/, b.order.p /*subject*/, b.p_qs2.p /*score, distance*/,
                                b.tmp64.p /*first-touch key*/, c->st);
            HIP_CHECK(hipMemsetAsync(b.bflag.p, 0, sizeof(u32), c->st));
            launch_q_best(b.bcnt.p, qa, nqp, b.order.p, b.p_qs2.p, b.tmp64.p, (u32)ch.seq_lo, bsp, b.c_rec.p, b.order2.p, qcnt, b.bflag.p, c->st);
            HIP_CHECK(hipMemcpyAsync(b.bccnt.p, qcnt + qa, (size_t)nqp * sizeof(u32), hipMemcpyDeviceToDevice, c->st));
            HIP_CHECK(hipMemsetAsync(b.bccnt.p + nqp, 0, 2 * sizeof(u32), c->st));
            const u32* dT = scan_u32(b.bccnt.p, b.bccnt.p, (size_t)nqp + 1, false, c->d_scan_tmp.p, c->st);
            stash_u32(c, dT, 0);
            u32 flag = 0;
            d2h_pair(c, b.bflag.p, NS, flag);
            if (!flag) {
                if (tune().debug) fprintf(stderr, "[sohit] seed pass: queries %u..%u hits %u seeds %u pass records %u candidates %u (per query)\n", qa, qb, H, K, NP, NS);
                if (NS) {
                    const u32 base = b.chunk_base.back();
                    if ((u64)base + NS >= cand_limit()) throw CandOverflow();   // (SOHIT_CAND_LIMIT: tests lower the limit to exercise the split)
                    b.cand_q.ensure((size_t)base + NS + 4, true, c->st);
                    b.cand_rec.ensure(4 * ((size_t)base + NS) + 16, true, c->st);
                    launch_q_emit(b.bcnt.p, qa, nqp, NP, b.pidx2.p, qcnt, b.bccnt.p, b.order2.p, b.c_rec.p, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base, c->st);
                    b.chunk_base.back() = base + NS;
                    c->cnt.candidates += NS;
                }
                sc.lap("group.best_order");
                c->cnt.seed_ms += (t1 - t0) * 1e3;
                c->cnt.group_ms += (wall() - t1) * 1e3;
                return;
            }
            // (a query above the LDS instance: k_q_best has written counts for the others -- the path below writes every query's again)
            NS = 0;
        }
        // best diagonal per (query, subject): sort pass records by (q, subject)
        b.pidx.ensure((size_t)NP + 2), b.pidx2.ensure((size_t)NP + 2), b.p_qs2.ensure((size_t)NP + 2);
        launch_iota(b.pidx.p, NP, c->st);
        ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(NP, 64));
        sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, q_qs, b.p_qs2.p, b.pidx.p, b.pidx2.p, NP, klr.bs + kl.bq, c->st);
        b.flags.ensure((size_t)NP + 4), b.gidx.ensure((size_t)NP + 4);
        c->d_small.ensure(16);
        launch_seg_flags(b.p_qs2.p, NP, b.flags.p, c->d_small.p, c->st);
        const u32* dS = scan_u32(b.flags.p, b.gidx.p, NP, false, c->d_scan_tmp.p, c->st);
        // per-query candidate segments and the longest one (d_small[0]), fetched with the candidate total
        b.qseg.ensure((size_t)b.nq + 4);
        launch_qseg(b.p_qs2.p, NP, b.gidx.p, dS, klr.bs, qa, qb, b.qseg.p, c->d_small.p, c->st);
        d2h_pair(c, dS, maxseg, NS);
        b.shead.ensure((size_t)NS + 2);
        if (tune().debug) fprintf(stderr, "[sohit] seed pass: queries %u..%u hits %u seeds %u pass records %u candidates %u\n", qa, qb, H, K, NP, NS);
        launch_group_list(b.flags.p, b.gidx.p, NP, b.shead.p, c->st);
        b.c_ft.ensure((size_t)NS + 2), b.c_q.ensure((size_t)NS + 2), b.c_rec.ensure(4 * (size_t)NS + 8);
        c_ftp = b.c_ft.p, c_qp = b.c_q.p, c_recp = b.c_rec.p;
        launch_best(b.p_qs2.p, b.pidx2.p, b.shead.p, NS, NP, q_sd, q_ft, (u32)ch.seq_lo, klr.bs, c_ftp, c_qp, c_recp, c->st);
        break;
    }
    if (NS == 0) {
        c->cnt.seed_ms += (t1 - t0) * 1e3;
        c->cnt.group_ms += (wall() - t1) * 1e3;
        return;
    }
    const int ftbits = kl.ba + kl.bp + ft_bits_entry;
    const bool cand_lds = tune().cand_lds;
    if (!bbest && cand_lds && maxseg <= (u32)cand_order_lds_max() && ftbits - bsp + 1 <= cand_order_lds_key_bits()) {
        // sparse path: every query's candidates fit the LDS sort -- ordered and written to the candidate store by one kernel
        const u32 base = b.chunk_base.back();
        if ((u64)base + NS >= cand_limit()) throw CandOverflow();   // (SOHIT_CAND_LIMIT: tests lower the limit to exercise the split)
        b.cand_q.ensure((size_t)base + NS + 4, true, c->st);
        b.cand_rec.ensure(4 * ((size_t)base + NS) + 16, true, c->st);
        launch_cand_order_lds(c_ftp, c_recp, b.qseg.p, qa, qb, maxseg, bsp, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base, qcnt, c->st);
        b.chunk_base.back() = base + NS;
        c->cnt.candidates += NS;
        sc.lap("group.best_order");
        c->cnt.seed_ms += (t1 - t0) * 1e3;
        c->cnt.group_ms += (wall() - t1) * 1e3;
        return;
    }
    if (cand_idx_bits) {
        const u32 base = b.chunk_base.back();
        if ((u64)base + NS >= cand_limit()) throw CandOverflow();   // (SOHIT_CAND_LIMIT: tests lower the limit to exercise the split)
        b.qseg.ensure((size_t)b.nq + 4), b.tmp64.ensure((size_t)NS + 2);
        launch_stride_gather(b.bccnt.p, bL.R, bL.nqp + 1, b.qseg.p, c->st);
        b.cand_q.ensure((size_t)base + NS + 4, true, c->st);
        b.cand_rec.ensure(4 * ((size_t)base + NS) + 16, true, c->st);
        bool lib_sort = !tune().cand_order_lds;
        if (!lib_sort) {   // sort + row gather per query in one hand-written kernel (k_cand_order_seg)
            HIP_CHECK(hipMemsetAsync(b.bflag.p, 0, sizeof(u32), c->st));
            launch_cand_order_seg(c_ftp, b.qseg.p, bL.nqp, bL.qa, cand_idx_bits, cand_idx_bits + cand_ftw, c_recp, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base,
                                  qcnt, b.bflag.p, c->st);
            lib_sort = d2h_u32(c, b.bflag.p) != 0;   // (a digit group above the LDS sort: never seen; the library path then redoes the pass's order)
        }
        if (lib_sort) {   // (SOHIT_CAND_ORDER_LDS=0: the library's segmented radix sort, then the gather)
            ensure_sort_tmp(c, sort_cand_keys_seg_temp_bytes(NS, bL.nqp, cand_idx_bits, cand_idx_bits + cand_ftw));
            sort_cand_keys_seg(c->d_sort_tmp.p, c->d_sort_tmp.cap, c_ftp, b.tmp64.p, NS, bL.nqp, b.qseg.p, cand_idx_bits, cand_idx_bits + cand_ftw, c->st);
            launch_emit_cands_seg(b.tmp64.p, b.qseg.p, bL.nqp, bL.qa, cand_idx_bits, c_recp, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base, qcnt, c->st);
        }
        b.chunk_base.back() = base + NS;
        c->cnt.candidates += NS;
        sc.lap("group.best_order");
        c->cnt.seed_ms += (t1 - t0) * 1e3;
        c->cnt.group_ms += (wall() - t1) * 1e3;
        return;
    }
    // order candidates by (query, first-touch): one sort on (q << ftbits | ft) when that fits 64 bits,
    // else sort by first-touch and then stable-sort by query; only the populated bits are sorted
    b.order.ensure((size_t)NS + 2), b.order2.ensure((size_t)NS + 2), b.tmp64.ensure((size_t)NS + 2), b.c_ft2.ensure((size_t)NS + 2);
    launch_iota(b.order.p, NS, c->st);
    int qshift = 0;  // where the query sits in the final sort's key stream (b.c_ft2)
    if (ftbits + kl.bq <= 64) {
        launch_combine_q_ft(c_qp, c_ftp, NS, ftbits, bsp, b.tmp64.p, c->st);
        const bool cand_seg = tune().cand_segsort;
        if (bbest && cand_seg && bL.nqp >= 256) {
            // k_bkt_best wrote the candidates query-major: a query's segment = [ccnt[q * R], ccnt[(q + 1) * R]); only the first-touch bits
            // are sorted, inside the segments (the query bits stay on top of the sort word for k_emit_cands)
            b.qseg.ensure((size_t)b.nq + 4);
            launch_stride_gather(b.bccnt.p, bL.R, bL.nqp + 1, b.qseg.p, c->st);
            ensure_sort_tmp(c, sort_pairs_u64_u32_seg_temp_bytes(NS, bL.nqp, 0, ftbits - bsp + 1));
            sort_pairs_u64_u32_seg(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, b.order.p, b.order2.p, NS, bL.nqp, b.qseg.p, 0,
                                   ftbits - bsp + 1, c->st);
        } else {
            ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(NS, 64));
            sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, b.order.p, b.order2.p, NS, ftbits - bsp + 1 + kl.bq, c->st);
        }
        std::swap(b.order.p, b.order2.p);
        std::swap(b.order.cap, b.order2.cap);
        qshift = ftbits - bsp + 1;
    } else {
        ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(NS, 64));
        sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, c_ftp, b.c_ft2.p, b.order.p, b.order2.p, NS, ftbits, c->st);
        launch_gather_u32_as_u64(c_qp, b.order2.p, NS, b.tmp64.p, c->st);
        sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, b.order2.p, b.order.p, NS, kl.bq, c->st);
    }
    // append to the candidate store
    const u32 base = b.chunk_base.back();
    if ((u64)base + NS >= cand_limit()) throw CandOverflow();  // search_loaded() splits the batch and runs the halves (SOHIT_CAND_LIMIT: tests)
    b.cand_q.ensure((size_t)base + NS + 4, true, c->st);
    b.cand_rec.ensure(4 * ((size_t)base + NS) + 16, true, c->st);
    b.segfirst.ensure((size_t)b.nq + 4);
    launch_emit_cands(b.order.p, NS, b.c_ft2.p, qshift, c_recp, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base, qcnt, b.segfirst.p, c->st);
    b.chunk_base.back() = base + NS;
    c->cnt.candidates += NS;
    sc.lap("group.best_order");  // (synchronises when profiling; otherwise the next pass is queued behind this one)
    c->cnt.seed_ms += (t1 - t0) * 1e3;
    c->cnt.group_ms += (wall() - t1) * 1e3;
}

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
HitCache g_hit_cache;

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

// wait for the row-emission job of the previous batch (if any) and apply its rare post-filter
void emit_join(so_ctx* c, HitBuf& out) {
    if (!c->emit.active) return;
    c->emit.th.join();
    c->emit.active = false;
    if (c->emit.err) {
        std::exception_ptr e = c->emit.err;
        c->emit.err = nullptr;
        std::rethrow_exception(e);
    }
    if (c->emit.dropped.load()) {
        // entry_point re-checks e <= expect (3234).  k_stop_round_w applied the same test to the same
        // doubles, so this never fires; kept as the reference has it.
        const double expect = c->expect;
        size_t wpos = c->emit.base;
        for (size_t k = c->emit.base; k < c->emit.base + c->emit.n; ++k)
            if (out.p[k].evalue <= expect) out.p[wpos++] = out.p[k];
        out.n = wpos;
    }
}

void phase2(so_ctx* c, Batch& b, HitBuf& out) {
    const double t0 = wall();
    StageClock sc(c);
    const int nchunks = (int)c->chunks.size();
    const u32 nq = b.nq;
    if (nq == 0) return;
    const u32 Ntot = b.chunk_base.empty() ? 0u : b.chunk_base.back();
    b.qtot.ensure((size_t)nq + 4), b.qcoff.ensure((size_t)nq + 4), b.prior.ensure((size_t)nq + 4), b.cqoff.ensure((size_t)nq + 4);
    HIP_CHECK(hipMemsetAsync(b.qtot.p, 0, ((size_t)nq + 4) * sizeof(u32), c->st));
    for (int ci = 0; ci < nchunks; ++ci) launch_add_u32(b.qtot.p, b.ccnt.p + (size_t)ci * nq, nq, c->st);
    c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)nq + 1) + 8);
    scan_u32(b.qtot.p, b.qcoff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
    b.fin_rec.ensure(4 * (size_t)Ntot + 16);
    HIP_CHECK(hipMemsetAsync(b.prior.p, 0, ((size_t)nq + 4) * sizeof(u32), c->st));
    for (int ci = 0; ci < nchunks; ++ci) {
        const u32 lo = ci == 0 ? 0u : b.chunk_base[ci - 1], hi = b.chunk_base[ci];
        const u32* cc = b.ccnt.p + (size_t)ci * nq;
        if (hi > lo) {
            scan_u32(cc, b.cqoff.p, nq, false, c->d_scan_tmp.p, c->st);
            launch_gather_cands(b.cand_q.p + lo, b.cand_rec.p + 4 * (size_t)lo, hi - lo, b.cqoff.p, b.prior.p, b.qcoff.p, b.fin_rec.p,
                                c->st);
        }
        launch_add_u32(b.prior.p, cc, nq, c->st);
    }
    sc.lap("phase2.gather");
    // candidate dump for so_query_candidates (tests only)
    if (tune().keep_cands) {
        std::vector<u32> qcoff((size_t)nq + 1), rec(4 * (size_t)Ntot + 4);
        HIP_CHECK(hipMemcpyAsync(qcoff.data(), b.qcoff.p, ((size_t)nq + 1) * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        if (Ntot) HIP_CHECK(hipMemcpyAsync(rec.data(), b.fin_rec.p, 4 * (size_t)Ntot * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        HIP_CHECK(hipStreamSynchronize(c->st));
        for (u32 q = 0; q < nq; ++q) {
            auto& dst = c->last_cands[(size_t)(b.q_lo - c->last_q_lo) + b.qid[q]];
            dst.assign(rec.begin() + 4 * (size_t)qcoff[q], rec.begin() + 4 * (size_t)qcoff[q + 1]);
        }
    }
    if (Ntot == 0) {
        c->cnt.phase2_ms += (wall() - t0) * 1e3;
        return;
    }
    const u32 vmax = (u32)std::max<i64>(100, std::max<i64>(c->v + 100, (i64)((double)c->v * 1.1)));  // fsearch.py:3059
    // ranks = candidates considered (top vmax); tasks = alignments (1 per rank, or one per 4096-tile
    // of a long candidate, kswat_st_long)
    b.perm.ensure((size_t)Ntot + 4), b.ntask.ensure((size_t)nq + 4), b.ntile.ensure((size_t)nq + 4);
    b.toff.ensure((size_t)nq + 4), b.roffc.ensure((size_t)nq + 4);
    HIP_CHECK(hipMemsetAsync(b.ntask.p, 0, ((size_t)nq + 4) * sizeof(u32), c->st));
    HIP_CHECK(hipMemsetAsync(b.ntile.p, 0, ((size_t)nq + 4) * sizeof(u32), c->st));
    {
        // queries with more candidates than the LDS sort holds need global scratch for the wave sort
        // (through the pinned per-query buffer of the seed stage: a pageable read of nq words costs more than the kernels around it)
        if (c->h_qhits_cap < nq) {
            if (c->h_qhits) (void)hipHostFree(c->h_qhits);
            c->h_qhits_cap = (size_t)nq + 1024;
            HIP_CHECK(hipHostMalloc((void**)&c->h_qhits, c->h_qhits_cap * sizeof(unsigned long long), hipHostMallocDefault));
        }
        const u32* qt = reinterpret_cast<const u32*>(c->h_qhits);
        HIP_CHECK(hipMemcpyAsync(c->h_qhits, b.qtot.p, (size_t)nq * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        HIP_CHECK(hipStreamSynchronize(c->st));
        u32 mx = 0;
        for (u32 i = 0; i < nq; ++i) mx = std::max(mx, qt[i]);
        u64* gx = nullptr;
        u32 *gL = nullptr, *gR = nullptr;
        if ((int)mx > csort_lds_max()) {
            b.gx.ensure((size_t)Ntot + 4), b.gL.ensure((size_t)Ntot + 4), b.gR.ensure((size_t)Ntot + 4);
            gx = b.gx.p, gL = b.gL.p, gR = b.gR.p;
        }
        // (the lists too long for the LDS instances are sorted in global scratch, a wave each: beside the LDS instances, on the side stream)
        const bool cs_aside = gx && tune().csort_aside;
        if (cs_aside) {
            HIP_CHECK(hipEventRecord(c->ev_ug_go, c->st));
            HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_ug_go, 0));
        }
        launch_csort(b.fin_rec.p, b.qcoff.p, nq, vmax, b.dev.d_off.p, c->ref.d_off.p, b.perm.p, b.ntask.p, b.ntile.p, gx, gL, gR, c->st,
                     cs_aside ? c->st_side : c->st);
        if (cs_aside) {
            HIP_CHECK(hipEventRecord(c->ev_ug_done, c->st_side));
            HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_ug_done, 0));
        }
    }
    const u32* dNRk = scan_u32(b.ntask.p, b.roffc.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
    stash_u32(c, dNRk, 0);
    const u32* dNT = scan_u32(b.ntile.p, b.toff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
    u32 NRK, NT;
    d2h_pair(c, dNT, NRK, NT);
    sc.lap("phase2.csort");
    b.tasks.ensure((size_t)NT + 4), b.ares.ensure((size_t)NT + 4), b.bits.ensure((size_t)NT + 4), b.sel.ensure((size_t)NT + 4);
    b.rk_slot.ensure((size_t)NRK + 4);
    b.qcells.ensure((size_t)nq + 2);
    HIP_CHECK(hipMemsetAsync(b.qcells.p, 0, ((size_t)nq + 2) * sizeof(unsigned long long), c->st));   // cells per query, added up by the stop rule
    launch_mktasks(b.fin_rec.p, b.qcoff.p, b.perm.p, b.ntask.p, b.roffc.p, b.toff.p, nq, b.dev.d_off.p, c->ref.d_off.p, b.tasks.p,
                   b.rk_slot.p, c->st);
    // k_align runs four alignments per wave and a wave lasts as long as its longest one: every launch list is ordered by band rows,
    // longest first (one 13-bit radix sort; config 3: align rounds 33.6 -> 28.4 ms, sort included).  SOHIT_ALIGN_SORT=0: as listed.
    const bool align_sort = tune().align_sort;
    // The packed 16-bit aligner takes a task whose scores fit its cells: 11 * min(rows, columns), or the smaller of the two sequences'
    // score bounds (k_seq_bound), within range.  That is a property of the TASK: a launch list is split into the tasks it cannot take
    // (k_task_rows clears bit 13 of their sort key, so they lead the sorted list, and counts them) and the rest.  Only batches that hold
    // a query AND a reference sequence above the length limit can contain such tasks at all.
    const u32 maxwin_q = std::min<u32>(b.maxqlen, LONG_SEQ), maxwin_s = std::min<u32>(c->ref.maxlen, LONG_SEQ);
    const bool pk_on = tune().align_pk && align_pk_supported(c->st);
    const bool pk_mixed = pk_on && (int)std::min(maxwin_q, maxwin_s) > align_pk_max_len();
    const PkCls pkc{b.dev.d_pcls.p, b.dev.d_pcls4.p, c->ref.d_pcls.p, c->ref.d_pcls4.p};
    const bool traced_pk = pk_on && tune().align_pk_trace;   // traced alignments by the packed kernel too (SOHIT_ALIGN_PK_TRACE=0: k_align<true>)
    auto sort_by_rows = [&](const u32* list, u32 n, u32* n_wide) -> const u32* {
        const bool split = n_wide && pk_mixed;
        if (n_wide) *n_wide = pk_on ? 0u : n;
        if (!split && (!align_sort || n < 4096)) return list;
        b.tmp64.ensure((size_t)n + 2), b.c_ft2.ensure((size_t)n + 2), b.ridx2.ensure((size_t)n + 2);
        ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(n, 64));
        // (ordering inside blocks of 2^k queries instead of globally -- key = query block << 13 | rows -- was measured: 24.9-25.2 ms of
        // align rounds for k = 7 ... 13 against 24.8-25.2)
        c->d_small.ensure(16);
        if (split) HIP_CHECK(hipMemsetAsync(c->d_small.p + 12, 0, sizeof(u32), c->st));
        launch_task_rows(b.tasks.p, list, n, b.dev.d_off.p, c->ref.d_off.p, b.dev.d_bound.p, c->ref.d_bound.p, align_pk_max_len(), align_pk_max_score(),
                         split ? c->d_small.p + 12 : nullptr, b.ucount.p + 2, b.tmp64.p, c->st);
        sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, list, b.ridx2.p, n, split ? 14 : 13, c->st);
        if (split) *n_wide = d2h_u32(c, c->d_small.p + 12);
        return b.ridx2.p;
    };
    // banded alignments in rounds (see k_round_counts / k_stop_round_w)
    const int maxrows = (int)std::min<u32>(std::max(maxwin_q, maxwin_s), std::min(maxwin_q, maxwin_s) + 16);
    const u32 stride = align_trace_stride(maxrows + 1);
    const size_t budget_words = (size_t)1 << 30;  // 4 GiB of trace scratch for the fixed-stride slabs
    const u32 slab = (u32)std::max<size_t>(16, std::min<size_t>(std::max<u32>(NT, 1), budget_words / std::max<u32>(stride, 1)));
    // Traces take what each task's own band needs: room per task (k_trace_units), scanned into b.tr_ofs; up to 8 GiB per launch list,
    // beyond that the list falls back to slabs of the batch-wide stride.  (`stride` follows the longest window of the batch: one
    // 4096-residue pair and every 300-row alignment owned 33 KB of trace, which its traceback then strode over.)
    const size_t var_budget_words = (size_t)1 << 31;
    const u32 TU = align_trace_unit();
    auto trace_offsets = [&](const u32* list, u32 n) -> size_t {   // -> words the list's traces need
        // the offsets are a 32-bit scan of units: a list whose total could wrap (no task needs more units than the batch-wide stride holds)
        // takes the slab path -- "does not fit" for both callers
        if ((u64)n * ((u64)(stride + TU - 1) / TU + 1) >= (1ull << 32)) return ~(size_t)0;
        b.tr_units.ensure((size_t)n + 4), b.tr_ofs.ensure((size_t)n + 4);
        c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)n + 1) + 8);
        launch_trace_units(b.tasks.p, list, n, b.dev.d_off.p, c->ref.d_off.p, b.tr_units.p, c->st);
        return (size_t)d2h_u32(c, scan_u32(b.tr_units.p, b.tr_ofs.p, (size_t)n + 1, false, c->d_scan_tmp.p, c->st)) * TU;
    };
    b.st_state.ensure(5 * (size_t)nq + 8), b.rcnt.ensure((size_t)nq + 4), b.tcnt.ensure((size_t)nq + 4), b.roff.ensure((size_t)nq + 4);
    b.order_tmp.ensure((size_t)nq + 4);
    b.ridx.ensure((size_t)NT + 4);
    HIP_CHECK(hipMemsetAsync(b.st_state.p, 0, (5 * (size_t)nq + 8) * sizeof(u32), c->st));
    sc.lap("phase2.mktasks");
    u32 aligned_total = 0;
    // Speculative traces (k_round_counts_spec): in the FIRST round, the leading tasks of every query whose ungapped score alone would pass
    // the e-value test are aligned with traces at once; reported rows that have one skip the second alignment.  SOHIT_SPEC=0: off.
    // SOHIT_SPEC=0 / 1: off / on whatever the size (default: on from 2^21 tasks; below that the extra launches cost more than they save:
    // config 2, 0.55 M tasks, 16.7 -> 17.1 ms).  SOHIT_SPEC_SLACK: the guess tests the ungapped score against expect x this (default 1e3:
    // config 3 keeps 1.44 M traces, all of them of reported rows, 175 k rows are left for the second pass; 1: 1.30 M / 315 k; 1e6: 1.56 M /
    // 57 k with 1.3 k traces unused -- a wrong guess costs about as much as a right one saves).
    const bool spec_on = tune().spec >= 0 ? tune().spec != 0 : NT >= (1u << 21);
    const double spec_slack = tune().spec_slack;
    u32 spec_cap = NT;   // (8 GiB of kept traces at most: checked on the list's actual trace sizes below)
    if (tune().spec_cap >= 0) spec_cap = (u32)tune().spec_cap;   // (tests: the round that does not fit)
    u32 nspec = 0;
    if (spec_on) {
        b.tpos.ensure((size_t)NT + 4);
        HIP_CHECK(hipMemsetAsync(b.tpos.p, 0xFF, ((size_t)NT + 4) * sizeof(u32), c->st));   // 0xFFFFFFFF = no trace kept
    }
    bool first_round = true;
    for (u32 minr = 8;; minr = minr < 256 ? minr * 2 : minr) {
        u32 NR = 0, RR = 0, NS = 0;
        bool spec_round = spec_on && first_round;
        first_round = false;
        if (spec_round) {
            b.spcnt.ensure((size_t)nq + 4), b.spoff.ensure((size_t)nq + 4), b.sidx.ensure((size_t)NT + 4);
            c->d_small.ensure(16);
            HIP_CHECK(hipMemsetAsync(c->d_small.p + 3, 0, sizeof(u32), c->st));
            launch_round_counts_spec(b.ntask.p, b.ntile.p, b.roffc.p, b.rk_slot.p, b.qcoff.p, b.st_state.p, nq, c->max_miss, minr, b.tasks.p, b.toff.p,
                                     b.dev.d_off.p, c->ref.d_off.p, c->d_bittab.p, so_ctx::BITTAB_N, c->ref.N, c->expect * spec_slack, b.rcnt.p, b.tcnt.p, b.spcnt.p,
                                     c->d_small.p + 3, c->st);
            stash_u32(c, scan_u32(b.tcnt.p, b.roff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st), 0);
            stash_u32(c, scan_u32(b.spcnt.p, b.spoff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st), 1);
            u32* v = (u32*)small_host(c);
            HIP_CHECK(hipMemcpyAsync(v, c->d_small.p, 4 * sizeof(u32), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipStreamSynchronize(c->st));
            NR = v[0], NS = v[1], RR = v[3];
            if (NS > spec_cap) spec_round = false;   // the traces would not fit: this round again, without them
        }
        if (!spec_round) {
            NS = 0;
            launch_round_counts(b.ntask.p, b.ntile.p, b.roffc.p, b.rk_slot.p, b.qcoff.p, b.st_state.p, nq, c->max_miss, minr, b.rcnt.p,
                                b.tcnt.p, c->st);
            const u32* dNR = scan_u32(b.tcnt.p, b.roff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
            stash_u32(c, dNR, 0);
            // ranks left this round (a round may hold ranks with zero tiles only)
            const u32* dRR = scan_u32(b.rcnt.p, b.order_tmp.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
            d2h_pair(c, dRR, NR, RR);
        }
        if (RR == 0) break;
        if (spec_round) {
            launch_round_idx_spec(b.tcnt.p, b.spcnt.p, b.roff.p, b.spoff.p, b.toff.p, b.ntask.p, b.ntile.p, b.roffc.p, b.rk_slot.p, b.st_state.p, nq,
                                  b.ridx.p, b.sidx.p, c->st);
            if (NS) {
                u32 nw_s = 0;   // (leading tasks of the ordered list whose scores need 32-bit cells)
                const u32* slist = sort_by_rows(b.sidx.p, NS, traced_pk ? &nw_s : nullptr);
                if (!traced_pk) nw_s = NS;
                const size_t tw = trace_offsets(slist, NS);
                ProfTimer pt(c, &c->cnt.align_ms, &c->cnt.align_launches);
                if (tw <= var_budget_words) {
                    b.spec_trace.ensure(tw + 64);
                    launch_align_traced(b.tasks.p, slist, NS, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_scls.p, c->ref.d_scls4.p,
                                        c->ref.d_off.p, c->d_b62c.p, b.spec_trace.p, TU, b.tr_ofs.p, b.ares.p, b.tpos.p, 0u, c->st, nw_s, pkc);
                    nspec = NS;
                } else {   // the traces would not fit after all: these tasks score-only, like the rest of the round
                    launch_align(b.tasks.p, slist, NS, b.dev.d_res.p, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_scls.p,
                                 c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p, nullptr, stride, nullptr, b.ares.p, false, c->st, 0u);
                }
                pt.stop();
            }
        } else if (NR) {
            launch_round_idx(b.tcnt.p, b.roff.p, b.toff.p, b.ntask.p, b.ntile.p, b.roffc.p, b.rk_slot.p, b.st_state.p, nq, b.ridx.p, c->st);
        }
        if (NR) {
            // score-only: the stop rule needs the maximum alone; the reported rows are traced in a second pass below
            u32 n_wide = 0;
            const u32* rlist = sort_by_rows(b.ridx.p, NR, &n_wide);
            ProfTimer pt(c, &c->cnt.align_ms, &c->cnt.align_launches);
            // score-only: the packed 16-bit kernel (two alignments per register) for every task whose scores fit it, the 32-bit one for the
            // n_wide tasks at the head of the list that do not
            // (the few wide tasks of a mixed batch are its longest: a launch of their own lasts as long as one 4096-row band, ~0.5 ms per
            // round with the GPU nearly idle -- so they run beside the packed kernel, on st_side, which is idle in phase 2; st_ug would
            // not do: it shares its hardware queue with the batch's stream on this runtime -- four queues, dealt round-robin)
            const bool wide_aside = n_wide && NR > n_wide && tune().wide_aside;
            hipStream_t wst = wide_aside ? c->st_side : c->st;
            if (wide_aside) {
                HIP_CHECK(hipEventRecord(c->ev_ug_go, c->st));
                HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_ug_go, 0));
            }
            if (n_wide)
                launch_align(b.tasks.p, rlist, n_wide, b.dev.d_res.p, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_scls.p,
                             c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p, nullptr, stride, nullptr, b.ares.p, false, wst, 0u);
            if (wide_aside) HIP_CHECK(hipEventRecord(c->ev_ug_done, c->st_side));
            if (NR > n_wide)
                launch_align_pk(b.tasks.p, rlist + n_wide, NR - n_wide, pkc, b.dev.d_off.p, c->ref.d_off.p, c->d_b62c.p, b.ares.p, c->st);
            if (wide_aside) HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_ug_done, 0));
            pt.stop();
            c->cnt.align_wide += n_wide;
        }
        launch_stop_round_w(b.tasks.p, b.ares.p, b.qcoff.p, b.ntask.p, b.ntile.p, b.roffc.p, b.rk_slot.p, b.toff.p, b.rcnt.p, nq,
                            b.dev.d_off.p, c->ref.d_off.p, c->d_bittab.p, so_ctx::BITTAB_N, c->ref.N, c->expect, c->max_miss, c->v, b.sel.p,
                            b.st_state.p, b.bits.p, b.qcells.p, c->st);
        aligned_total += NR + NS;
    }
    launch_sum_u64(b.qcells.p, nq, b.ucount.p + 1, c->st);
    sc.lap("phase2.align_rounds");
    c->cnt.alignments += aligned_total;
    b.nout.ensure((size_t)nq + 4), b.ooff.ensure((size_t)nq + 4);
    HIP_CHECK(hipMemsetAsync(b.nout.p, 0xFF, (size_t)nq * sizeof(u32), c->st));  // 0xFFFFFFFF = not selected yet
    HIP_CHECK(hipMemsetAsync(b.nout.p + nq, 0, 4 * sizeof(u32), c->st));
    launch_final_select(b.toff.p, nq, c->v, b.sel.p, b.st_state.p, b.bits.p, b.nout.p, c->st);
    const u32* dNO = scan_u32(b.nout.p, b.ooff.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
    // The reported rows leave in up to EMIT_PARTS query ranges: a range's rows are traced, written and sent to the host while the next
    // range is being traced (one batch per search leaves nothing else to hide the download behind).  The ranges' first rows come
    // back with the row total: ooff at every (nq / parts)-th query.
    enum { EMIT_PARTS_MAX = 8 };
    // SOHIT_EMIT_PARTS (1-8, default 4) / SOHIT_EMIT_MIN_ROWS (default 2^18: smaller results leave in one piece): tuning and test switches
    // (with kept traces -- nspec -- the last stage is short: fewer ranges, SOHIT_SPEC_PARTS, default 2)
    const int EMIT_PARTS = std::min<int>(EMIT_PARTS_MAX, std::max(1, nspec ? (int)tune().spec_parts : (int)tune().emit_parts));
    const u32 emit_min_rows = (u32)std::max(1ll, tune().emit_min_rows);
    // (config 3, one batch: 1 part 57.0 ms per step, 4 parts 56.0)
    const u32 qstep = (nq + EMIT_PARTS - 1) / EMIT_PARTS;
    c->d_small.ensure(16);
    launch_stride_gather(b.ooff.p, qstep, (nq + qstep - 1) / qstep, c->d_small.p + 4, c->st);   // d_small[4 + p] = first row of range p
    stash_u32(c, dNO, 0);
    u32 NO, part_row[EMIT_PARTS_MAX + 1];
    std::vector<u32> h_ooff;   // (permuted batch, host rows: the slots' first rows, for the file-order placement below -- fetched with the totals)
    {
        u32* v = (u32*)small_host(c);
        HIP_CHECK(hipMemcpyAsync(v, c->d_small.p, 12 * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        if (b.permuted && !c->dev_out) {
            h_ooff.resize((size_t)nq + 1);
            HIP_CHECK(hipMemcpyAsync(h_ooff.data(), b.ooff.p, ((size_t)nq + 1) * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        }
        HIP_CHECK(hipStreamSynchronize(c->st));
        NO = v[0];
        for (int p = 0; p <= EMIT_PARTS; ++p) part_row[p] = (u64)p * qstep < nq ? v[4 + p] : NO;
    }
    sc.lap("phase2.stop");
    if (NO) {
        // second aligner pass, with traces + traceback, over the rows that are reported (a few percent of the alignments)
        const int parts = (c->dev_out || NO < emit_min_rows) ? 1 : EMIT_PARTS;
        u32 part_lo[EMIT_PARTS_MAX] = {0}, part_hi[EMIT_PARTS_MAX] = {0};   // rows of emission range p (what ev_part[p] stands for)
        b.sel_idx.ensure((size_t)NO + 4);
        launch_selected_idx(b.toff.p, b.sel.p, b.nout.p, b.ooff.p, nq, b.sel_idx.p, c->st);
        u32 maxpart = NO;
        if (parts > 1) {
            maxpart = 0;
            for (int p = 0; p < parts; ++p) maxpart = std::max(maxpart, part_row[p + 1] - part_row[p]);
        }
        b.outrec.ensure(12 * (size_t)NO + 16);
        if (c->rows_in_flight) {  // the previous batch's rows may still be on their way out of b.outrec
            HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_rows_done, 0));
            c->rows_in_flight = false;
        }
        if (!c->dev_out) {
            emit_join(c, out);  // the previous batch's job reads the staging buffer and writes into `out`
            // pinned staging buffer: pageable D2H runs at ~1 GB/s, pinned at PCIe speed
            if (c->pinned_cap < (size_t)NO * sizeof(HostRow)) {
                if (c->pinned) (void)hipHostFree(c->pinned);
                c->pinned_cap = (size_t)NO * sizeof(HostRow) * 5 / 4 + 4096;
                HIP_CHECK(hipHostMalloc(&c->pinned, c->pinned_cap, hipHostMallocDefault));
            }
        }
        const u32* slist = b.sel_idx.p;  // (ordering this pass by rows too costs more than it saves: 9.1 -> 9.9 ms on config 3)
        // Kept traces: rows that have one only need the walk, the others are aligned with traces now.  The row list is split stably
        // (flags, scan, scatter); range p's rows without a trace are list B's [pb[p], pb[p + 1]), the others list A's
        // [first row - pb[p], ...): the scan values at the ranges' first rows come back in one small copy.
        u32 pb[EMIT_PARTS_MAX + 1] = {0};
        if (nspec) {
            b.flags.ensure((size_t)NO + 4), b.gidx.ensure((size_t)NO + 4), b.sel_b.ensure((size_t)NO + 4), b.sel_a.ensure((size_t)NO + 4);
            c->d_scan_tmp.ensure(scan_u32_temp_elems((size_t)NO + 1) + 8);
            launch_trace_flags(slist, NO, b.tpos.p, b.flags.p, c->st);
            const u32* dNB = scan_u32(b.flags.p, b.gidx.p, NO, false, c->d_scan_tmp.p, c->st);
            launch_trace_split(slist, NO, b.flags.p, b.gidx.p, b.sel_b.p, b.sel_a.p, c->st);
            u32* v = (u32*)small_host(c);
            HIP_CHECK(hipMemcpyAsync(v + parts, dNB, sizeof(u32), hipMemcpyDeviceToHost, c->st));
            for (int p = 1; p < parts; ++p) {
                if (part_row[p] < NO) HIP_CHECK(hipMemcpyAsync(v + p, b.gidx.p + part_row[p], sizeof(u32), hipMemcpyDeviceToHost, c->st));
            }
            HIP_CHECK(hipStreamSynchronize(c->st));
            pb[parts] = v[parts];
            for (int p = 1; p < parts; ++p) pb[p] = part_row[p] < NO ? v[p] : pb[parts];
            if (tune().debug) fprintf(stderr, "[sohit] kept traces %u, reported rows %u, of them without a trace %u\n", nspec, NO, pb[parts]);
        }
        // the list aligned with traces now: the rows without a kept trace (nspec), or all rows; its traces take their own sizes
        // (b.tr_ofs) when the whole list fits the budget, else slabs of the batch-wide stride
        const u32* tlist = nspec ? b.sel_b.p : slist;
        const u32 tn = nspec ? pb[parts] : NO;
        const u32* alist = b.sel_a.p;   // (nspec) rows that only need the walk
        // On a batch of mixed lengths the lists are ordered by band rows inside each emission range: k_align runs four alignments per
        // wave and k_traceback sixty-four walks, and either lasts as long as its longest (on uniform lengths the sort costs more than
        // it saves -- config 3: 9.1 -> 9.9 ms -- hence the test).  The traces' offsets follow the ordered list.
        const bool order_rows = (b.permuted || (u64)b.maxqlen * b.nq > 3ull * b.h_off[b.nq] / 2) &&
                                tune().trace_sort;
        // tasks of emission range p's traced list that need the 32-bit cells (they lead the ordered range); a mixed batch whose lists are
        // not ordered keeps the 32-bit kernel for all of them
        u32 nwide_part[EMIT_PARTS_MAX] = {0};
        if (order_rows) {
            // [t0, t1) of `in`, longest band first, to the same range of `out`; split: the wide tasks first, returns their number
            auto order_list = [&](const u32* in, u32 t0, u32 t1, u32* out, bool split) -> u32 {
                const u32 n = t1 - t0;
                if (!n) return 0u;
                b.tmp64.ensure((size_t)n + 2), b.c_ft2.ensure((size_t)n + 2);
                ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(n, 64));
                c->d_small.ensure(16);
                if (split) HIP_CHECK(hipMemsetAsync(c->d_small.p + 12, 0, sizeof(u32), c->st));
                launch_task_rows(b.tasks.p, in + t0, n, b.dev.d_off.p, c->ref.d_off.p, b.dev.d_bound.p, c->ref.d_bound.p, split ? align_pk_max_len() : 0,
                                 split ? align_pk_max_score() : 0u, split ? c->d_small.p + 12 : nullptr, nullptr, b.tmp64.p, c->st);
                sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, in + t0, out + t0, n, split ? 14 : 13, c->st);
                return split ? d2h_u32(c, c->d_small.p + 12) : 0u;
            };
            b.tl_sorted.ensure((size_t)tn + 4);
            if (nspec) b.al_sorted.ensure((size_t)(NO - tn) + 4);
            for (int p = 0; p < parts; ++p) {
                const u32 r0 = parts > 1 ? part_row[p] : 0u, r1 = parts > 1 ? part_row[p + 1] : NO;
                const bool split = traced_pk && pk_mixed;
                if (nspec) {
                    nwide_part[p] = order_list(tlist, pb[p], pb[p + 1], b.tl_sorted.p, split);
                    (void)order_list(alist, r0 - pb[p], r1 - pb[p + 1], b.al_sorted.p, false);
                } else {
                    nwide_part[p] = order_list(tlist, r0, r1, b.tl_sorted.p, split);
                }
            }
            tlist = b.tl_sorted.p;
            if (nspec) alist = b.al_sorted.p;
        }
        const size_t tw = tn ? trace_offsets(tlist, tn) : 0;
        const bool tvar = tw <= var_budget_words;
        b.trace.ensure(tvar ? tw + 64 : (size_t)std::min(slab, std::max<u32>(maxpart, 1)) * stride + 64);
        auto align_traced = [&](u32 t0, u32 t1, int p) {   // tasks [t0, t1) of tlist = emission range p's
            if (t1 <= t0) return;
            // the range's leading tasks that take the 32-bit kernel
            const u32 nw = !traced_pk ? t1 - t0 : (order_rows ? std::min(nwide_part[p], t1 - t0) : (pk_mixed ? t1 - t0 : 0u));
            if (tvar) {
                launch_align(b.tasks.p, tlist + t0, t1 - t0, b.dev.d_res.p, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_scls.p,
                             c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p, b.trace.p, TU, b.tr_ofs.p + t0, b.ares.p, true, c->st, nw, pkc);
                return;
            }
            for (u32 t = t0; t < t1; t += slab) {
                const u32 n = std::min(slab, t1 - t);
                launch_align(b.tasks.p, tlist + t, n, b.dev.d_res.p, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_scls.p,
                             c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p, b.trace.p, stride, nullptr, b.ares.p, true, c->st,
                             std::min(n, nw > t - t0 ? nw - (t - t0) : 0u), pkc);
            }
        };
        for (int p = 0; p < parts; ++p) {
            const u32 r0 = parts > 1 ? part_row[p] : 0u, r1 = parts > 1 ? part_row[p + 1] : NO;
            const u32 qa = parts > 1 ? std::min<u32>(nq, (u32)p * qstep) : 0u, qb = parts > 1 ? std::min<u32>(nq, (u32)(p + 1) * qstep) : nq;
            if (r1 > r0 && nspec) {
                const u32 b0 = pb[p], b1 = pb[p + 1], a0 = r0 - b0, a1 = r1 - b1;
                ProfTimer pt(c, &c->cnt.align_ms, &c->cnt.align_launches);
                align_traced(b0, b1, p);
                launch_traceback(b.tasks.p, alist + a0, a1 - a0, b.dev.d_res.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_off.p, b.spec_trace.p, TU,
                                 b.tpos.p, b.ares.p, c->st);
                pt.stop();
            } else if (r1 > r0) {
                ProfTimer pt(c, &c->cnt.align_ms, &c->cnt.align_launches);
                align_traced(r0, r1, p);
                pt.stop();
            }
            launch_emit_hits(b.tasks.p, b.ares.p, b.toff.p, b.sel.p, b.nout.p, b.ooff.p, b.bits.p, qa, qb, b.outrec.p, c->st);
            if (!c->dev_out && r1 > r0) {
                // the range's rows are downloaded on a second stream, behind the kernel that wrote them
                HIP_CHECK(hipEventRecord(c->ev_rows, c->st));
                HIP_CHECK(hipStreamWaitEvent(c->st_rows, c->ev_rows, 0));
                HIP_CHECK(hipMemcpyAsync((char*)c->pinned + (size_t)r0 * sizeof(HostRow), b.outrec.p + 12 * (size_t)r0, (size_t)(r1 - r0) * sizeof(HostRow),
                                         hipMemcpyDeviceToHost, c->st_rows));
                HIP_CHECK(hipEventRecord(c->ev_part[p], c->st_rows));
            }
            part_lo[p] = r0, part_hi[p] = r1;
        }
        sc.lap("phase2.trace_pass");
        {   // SOHIT_TEST_OOM_PHASE2=1 (tests): the first multi-query batch of the process fails here, as a device allocation of the
            // emission stage would -- search_loaded() reruns it as two halves
            static bool fired = false;
            if (!fired && nq > 1 && tune().test_oom_phase2) {
                fired = true;
                throw DevOom(0);
            }
        }
        if (c->dev_out) {
            // device-resident results: the so_hit records are built in HBM and appended to the ctx's result buffer
            if (!c->d_p2tab.p) {
                std::vector<double> p2(1200);
                for (int k = 0; k < 1200; ++k) p2[k] = p_pow(2, (double)(-k));  // bit2e's pow(2, -bit): exact powers of two from libm
                c->d_p2tab.ensure(1200);
                HIP_CHECK(hipMemcpy(c->d_p2tab.p, p2.data(), 1200 * sizeof(double), hipMemcpyHostToDevice));
            }
            c->d_hits.ensure((c->d_hits_n + NO) * sizeof(so_hit) + 256, true, c->st);
            const u32 *d_qid = nullptr, *d_ostart = nullptr;
            if (b.permuted) {   // records in file order: row counts scattered to file order, scanned
                b.d_ocnt.ensure((size_t)nq + 4), b.d_ostart.ensure((size_t)nq + 4);
                launch_scatter_u32(b.nout.p, b.d_qid.p, nq, b.d_ocnt.p, c->st);
                HIP_CHECK(hipMemsetAsync(b.d_ocnt.p + nq, 0, sizeof(u32), c->st));
                scan_u32(b.d_ocnt.p, b.d_ostart.p, (size_t)nq + 1, false, c->d_scan_tmp.p, c->st);
                d_qid = b.d_qid.p, d_ostart = b.d_ostart.p;
            }
            launch_make_hits(b.outrec.p, NO, b.q_lo, d_qid, b.ooff.p, d_ostart, c->qry.d_off.p, c->ref.d_off.p, c->ref.N, c->d_p2tab.p, 1200,
                             c->d_hits.p + c->d_hits_n * sizeof(so_hit), c->st);
            c->d_hits_n += NO;
            sc.lap("phase2.emit_device");
            c->cnt.phase2_ms += (wall() - t0) * 1e3;
            return;
        }
        const HostRow* rows = (const HostRow*)c->pinned;
        // the worker below waits for the last range's copy, the main thread goes on to the next batch (whose row kernel in turn waits
        // for that copy before it overwrites the device rows)
        HIP_CHECK(hipEventRecord(c->ev_rows_done, c->st_rows));
        c->rows_in_flight = true;
        if (c->profile) HIP_CHECK(hipEventSynchronize(c->ev_rows_done));
        sc.lap("phase2.emit_d2h");
        const i64 D = c->ref.N;
        // pow(2, -bit) (bit2e, fsearch.py:1086) tabulated once with libm: exact powers of two, 0 past the subnormals
        static std::vector<double> p2;
        if (p2.empty()) {
            p2.resize(1200);
            for (int k = 0; k < 1200; ++k) p2[k] = p_pow(2, (double)(-k));
        }
        const size_t base = out.n;
        out.grow(NO);
        so_hit* dst = out.p + base;
        out.n = base + NO;
        const double expect = c->expect;
        const i64 q_lo = b.q_lo;
        const double* p2p = p2.data();
        // A batch that holds its queries in length-class order hands the rows over in that order; they are written in FILE order:
        // slot s's rows, [ooff[s], ooff[s + 1]) of the download, start at row ostart[qid[s]] -- place[s] = {query, destination - source}.
        std::shared_ptr<std::vector<std::pair<u32, i64>>> place;
        if (b.permuted) {
            const std::vector<u32>& ooff = h_ooff;
            std::vector<u32> ocnt((size_t)nq + 1, 0);
            for (u32 s = 0; s < nq; ++s) ocnt[b.qid[s]] = ooff[s + 1] - ooff[s];
            u32 run = 0;
            for (u32 o = 0; o < nq; ++o) {
                const u32 n = ocnt[o];
                ocnt[o] = run;
                run += n;
            }
            place = std::make_shared<std::vector<std::pair<u32, i64>>>(nq);
            for (u32 s = 0; s < nq; ++s) (*place)[s] = {b.qid[s], (i64)ocnt[b.qid[s]] - (i64)ooff[s]};
        }
        c->emit.base = base, c->emit.n = NO;
        c->emit.dropped.store(0);
        c->emit.active = true;
        // The worker converts range p's rows as soon as they have arrived, while the GPU traces range p + 1: behind the last copy only
        // the last range is left (it used to wait for ALL rows: ~1.3 ms of a config-3 step with the GPU idle).  Its threads are started
        // once and walk the ranges together.
        struct PartSpan { u32 lo, hi; };
        std::array<PartSpan, EMIT_PARTS_MAX> spans{};
        for (int p = 0; p < parts; ++p) spans[(size_t)p] = {part_lo[p], part_hi[p]};
        c->emit.th = std::thread([c, rows, dst, NO, D, expect, q_lo, p2p, place, spans, parts] {
            try {
                HIP_CHECK(hipSetDevice(c->device));
                auto convert = [&](i64 i) {
                    const int* v = rows[i].v;
                    so_hit h;
                    i64 di = i;
                    if (place) {
                        const auto& pl = (*place)[(size_t)v[0]];
                        h.qidx = q_lo + pl.first;
                        di = i + pl.second;
                    } else {
                        h.qidx = q_lo + v[0];
                    }
                    h.sidx = v[1];
                    h.aln = v[2], h.mis = v[3], h.gap = v[4], h.qst = v[5], h.qed = v[6], h.sst = v[7], h.sed = v[8], h.bit = v[9];
                    h.ungapped = v[10], h.matches = v[11];
                    h.qlen = (int32_t)c->qry.len(h.qidx);
                    h.slen = (int32_t)c->ref.len(h.sidx);
                    // idy: one += 1. per identical column, then idy *= (100. / AL) (fsearch.py:1458-1459, 1471)
                    h.identity = (double)h.matches * (100. / (double)h.aln);
                    // bit2e (1086): D * len(sqi) * len(sqj) * pow(2, -bit)
                    const double pw = (h.bit >= 0 && h.bit < 1200) ? p2p[h.bit] : p_pow(2, (double)(-h.bit));
                    h.evalue = (double)(D * (i64)h.qlen * (i64)h.slen) * pw;
                    if (!(h.evalue <= expect)) c->emit.dropped.fetch_add(1);
                    dst[di] = h;
                };
                const unsigned nt = NO < 200000 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
                std::array<std::atomic<i64>, EMIT_PARTS_MAX> next;
                for (auto& n : next) n.store(0);
                std::exception_ptr werr;
                std::mutex wmu;
                auto worker = [&] {
                    try {
                        HIP_CHECK(hipSetDevice(c->device));
                        for (int p = 0; p < parts; ++p) {
                            const i64 lo = spans[(size_t)p].lo, n = (i64)spans[(size_t)p].hi - lo;
                            if (n <= 0) continue;
                            HIP_CHECK(hipEventSynchronize(c->ev_part[p]));   // the range's rows have arrived in the pinned buffer
                            for (;;) {
                                const i64 b0 = next[(size_t)p].fetch_add(4096);
                                if (b0 >= n) break;
                                for (i64 i = b0; i < std::min(n, b0 + 4096); ++i) convert(lo + i);
                            }
                        }
                    } catch (...) {
                        std::lock_guard<std::mutex> g(wmu);
                        werr = std::current_exception();
                    }
                };
                std::vector<std::thread> th;
                for (unsigned t = 1; t < nt; ++t) th.emplace_back(worker);
                worker();
                for (auto& t : th) t.join();
                if (werr) std::rethrow_exception(werr);
            } catch (...) {
                c->emit.err = std::current_exception();
            }
        });
    }
    sc.lap("phase2.emit_host");
    c->cnt.phase2_ms += (wall() - t0) * 1e3;
}

void search_loaded(so_ctx* c, i64 q_lo, i64 q_hi, HitBuf& out) {
    if (!c->ref_loaded) throw SoError("so_search: no reference loaded");
    if (!c->qry_loaded) throw SoError("so_search: no queries loaded");
    build_index(c);
    struct EmitGuard {  // an exception must not leave a worker writing into a result buffer that is being freed
        so_ctx* c;
        ~EmitGuard() {
            if (c->emit.active) {
                c->emit.th.join();
                c->emit.active = false;
                c->emit.err = nullptr;
            }
        }
    } emit_guard{c};
    const double t0 = wall();
    const i64 N = c->qry.N, D = c->ref.N;
    i64 st = std::min<i64>(std::max<i64>(0, q_lo), N);       // fsearch.py:2980
    i64 ed = std::min<i64>(q_hi < 0 ? D : q_hi, N);          // 2981 (uses D when -u < 0)
    if (ed < st) ed = st;
    c->last_q_lo = st;
    if (tune().keep_cands) c->last_cands.assign((size_t)(ed - st), std::vector<u32>());   // (tests: so_query_candidates)
    else c->last_cands.clear();
    c->masked.clear();
    const int nchunks = (int)c->chunks.size();
    if (tune().batch > 0) c->max_batch = (u32)tune().batch;
    if (tune().max_hits > 0) c->max_hits_per_pass = (size_t)tune().max_hits;
    // A batch's candidate store is indexed with 32 bits (and costs 20 bytes of HBM per candidate).  A query has at most one candidate per
    // reference sequence and at most one per seed hit; the hits a query expects follow from the index itself (a window drawn like the
    // reference's own hits sum(c^2) / sum(c) entries per chunk).  Batches are sized so that the estimate stays below 2^32 -- the sparse
    // weight-10 seed of config 3 then runs its 100k queries as ONE batch (59.1 ms against 60.4 for 65536 + 34464 and 62.1 for three
    // batches of 37580, same box), the 1 M-protein run keeps its 3758-query batches (106.1 s; 16384-query batches 110.5 s).  The
    // estimate can be wrong (queries unlike the reference): nothing is emitted before a batch's seed stage has finished, so a batch whose
    // store would overflow (seed_stage throws CandOverflow) or whose buffers do not fit the device (DevOom) is run again as two halves.
    double est_hits = 0;
    {
        long double s2 = 0, e1 = 0;
        for (auto& ch : c->chunks) s2 += (long double)ch->s2, e1 += (long double)ch->E;
        const double avg_qlen = N ? (double)c->qry.off[(size_t)N] / (double)N : 0.;
        if (e1 > 0) est_hits = (double)(s2 / e1) * (double)c->chunks.size() * avg_qlen;
    }
    const double est_cands = std::max(1., std::min((double)std::max<i64>(D, 1), est_hits));
    i64 batch_q = std::max<i64>(1, c->max_batch);
    if (tune().batch <= 0) batch_q = std::min<i64>(batch_q, std::max<i64>(1024, (i64)((double)0xE0000000ull / est_cands)));
    std::function<void(i64, i64)> run_batch = [&](i64 b0, i64 b1) {
        const so_counters keep = c->cnt;
        const size_t keep_rows = out.n, keep_dev_rows = c->d_hits_n;   // what the batch may have appended before it failed
        try {
            if (!c->batch) c->batch = std::make_shared<Batch>();
            Batch& b = *static_cast<Batch*>(c->batch.get());
            b.chunk_base.clear();
            StageClock scp(c);
            prepare_batch(c, b, b0, b1);
            scp.lap("prepare_batch");
            b.ccnt.ensure((size_t)std::max(1, nchunks) * b.nq + 4);
            HIP_CHECK(hipMemsetAsync(b.ccnt.p, 0, ((size_t)std::max(1, nchunks) * b.nq + 4) * sizeof(u32), c->st));
            for (int ci = 0; ci < nchunks; ++ci) seed_stage(c, b, ci);
            phase2(c, b, out);
            unsigned long long uc[3] = {0, 0, 0};
            HIP_CHECK(hipMemcpyAsync(uc, b.ucount.p, sizeof uc, hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipStreamSynchronize(c->st));
            c->cnt.cells += (i64)uc[1];
            c->cnt.cells_wide += (i64)uc[2];
            c->cnt.n_queries += b.nq;
            c->cnt.query_aa += b.h_off[b.nq];
            if (tune().keep_masked) {
                if (c->masked.empty()) c->masked_lo = st;
                if (b.h_res.empty() && b.h_off[b.nq]) {
                    b.h_res.resize(b.h_off[b.nq] + 16);
                    HIP_CHECK(hipMemcpy(b.h_res.data(), b.dev.d_res.p, b.h_off[b.nq], hipMemcpyDeviceToHost));
                }
                const size_t m0 = c->masked.size();
                c->masked.resize(m0 + b.nq);
                for (u32 i = 0; i < b.nq; ++i)
                    c->masked[m0 + b.qid[i]].assign((const char*)b.h_res.data() + b.h_off[i], (size_t)(b.h_off[i + 1] - b.h_off[i]));
            }
        } catch (const SoError& e) {
            const bool oom = dynamic_cast<const DevOom*>(&e) != nullptr;
            if (!oom && !dynamic_cast<const CandOverflow*>(&e)) throw;
            if (b1 - b0 < 2) throw SoError(oom ? std::string(e.what()) : std::string("one query collected >= 2^32 candidates"));
            (void)hipStreamSynchronize(c->st);
            (void)hipStreamSynchronize(c->st_rows);   // row downloads the failed attempt had queued
            (void)hipStreamSynchronize(c->st_side);
            c->rows_in_flight = false;
            if (oom) c->batch.reset();  // hand the batch's buffers back before the halves allocate theirs
            c->cnt = keep;
            out.n = keep_rows, c->d_hits_n = keep_dev_rows;   // (a failure in phase 2 comes after rows of the batch may have been appended)
            const i64 mid = b0 + (b1 - b0) / 2;
            run_batch(b0, mid);
            run_batch(mid, b1);
        }
    };
    for (i64 b0 = st; b0 < ed; b0 += batch_q) run_batch(b0, std::min<i64>(ed, b0 + batch_q));
    emit_join(c, out);  // the last batch's rows
    c->cnt.rows += c->dev_out ? (i64)c->d_hits_n : (i64)out.n;
    c->cnt.total_ms += (wall() - t0) * 1e3;
}

// ---------------------------------------------------------------------------------------------
// Row formatting (entry_point, fsearch.py:3234-3243; f2s 43-61)
// ---------------------------------------------------------------------------------------------
// "%f" of a double, as glibc prints it: six decimals, the EXACT binary value rounded half-to-even.  Done with 128-bit integers for
// |x| < 1e15 (x = m * 2^e exactly; m * 10^6 fits 74 bits) -- three of these per row were most of the 0.25 us a row took in snprintf;
// anything else (huge, inf, nan) goes through snprintf.  Returns the number of characters written (no terminator).
int fmt_f6(double x, char* out) {
    if (!(std::fabs(x) < 1e15)) return snprintf(out, 400, "%f", x);
    char* o = out;
    if (std::signbit(x)) *o++ = '-', x = -x;
    unsigned __int128 q = 0;
    if (x != 0) {
        int ex;
        const double fr = frexp(x, &ex);                  // x = fr * 2^ex, 0.5 <= fr < 1
        const u64 m = (u64)ldexp(fr, 53);                 // 53-bit integer mantissa
        const int e = ex - 53;
        const unsigned __int128 P = (unsigned __int128)m * 1000000u;
        if (e >= 0) {
            q = P << e;                                   // (x < 1e15 < 2^50: e <= -3 in fact)
        } else if (-e < 100) {
            const int sh = -e;
            q = P >> sh;
            const unsigned __int128 rem = P & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }                                                 // else: below 2^-26 of a unit of the last place: 0
    }
    const u64 ip = (u64)(q / 1000000u);
    u32 fp = (u32)(q % 1000000u);
    char tmp[24];
    int n = 0;
    u64 v = ip;
    do tmp[n++] = (char)('0' + v % 10), v /= 10;
    while (v);
    while (n) *o++ = tmp[--n];
    *o++ = '.';
    for (int k = 5; k >= 0; --k) o[k] = (char)('0' + fp % 10), fp /= 10;
    o += 6;
    return (int)(o - out);
}

std::string fmt_f(double x) {
    char buf[400];
    return std::string(buf, (size_t)fmt_f6(x, buf));
}

// f2s (fsearch.py:43-61) into `out`; returns the length
int f2s_into(double e, char* out) {
    if (e <= 0) {
        out[0] = '0';
        return 1;
    }
    if (e < 1e-3) {
        double a = p_log10(e);
        a -= (double)(i64)a;
        if (a < 0) {
            double t = 1 + a;
            a = (t != 0) ? t : a;
        }
        const double b = p_pow(10, a);
        char sb[400], pb[400];
        const int sl = fmt_f6(p_log10(e / b), sb), pl = fmt_f6(b, pb);
        const char* sd = (const char*)memchr(sb, '.', (size_t)sl);
        const int sn = sd ? (int)(sd - sb) : 0;                          // the exponent: everything in front of the point
        const char* pd = (const char*)memchr(pb, '.', (size_t)pl);
        const int pn = std::min(pl, pd ? (int)(pd - pb) + 3 : 2);       // the mantissa cut behind its second decimal
        memcpy(out, pb, (size_t)pn);
        out[pn] = 'e';
        memcpy(out + pn + 1, sb, (size_t)sn);
        return pn + 1 + sn;
    }
    return fmt_f6(e, out);
}

std::string f2s(double e) {
    char buf[900];
    return std::string(buf, (size_t)f2s_into(e, buf));
}

inline char* put_int(char* o, long long v) {
    if (v < 0) *o++ = '-', v = -v;   // (never LLONG_MIN here)
    char tmp[24];
    int n = 0;
    do tmp[n++] = (char)('0' + v % 10), v /= 10;
    while (v);
    while (n) *o++ = tmp[--n];
    return o;
}

// one row of the 16-column file appended to `out` (entry_point, fsearch.py:3234-3243)
void format_hit_into(so_ctx* c, const so_hit& h, std::vector<char>& out) {
    if (h.qidx < 0 || h.qidx >= c->qry.N || h.sidx < 0 || h.sidx >= c->ref.N) throw SoError("so_format_hit: hit does not belong to the loaded files");
    const SeqSet &Q = c->qry, &R = c->ref;
    const size_t ql = Q.id_len[(size_t)h.qidx], sl = R.id_len[(size_t)h.sidx], hl = R.hd_len[(size_t)h.sidx];
    const size_t at = out.size();
    out.resize(at + ql + sl + hl + 1400);   // two ids, the header, 14 numbers (three of them doubles: up to 400 characters each)
    char* o = out.data() + at;
    memcpy(o, Q.data.data() + Q.hd_beg[(size_t)h.qidx], ql), o += ql, *o++ = '\t';
    memcpy(o, R.data.data() + R.hd_beg[(size_t)h.sidx], sl), o += sl, *o++ = '\t';
    {   // identity: "%f" cut behind its second decimal
        char b[400];
        const int n = fmt_f6(h.identity, b);
        const char* d = (const char*)memchr(b, '.', (size_t)n);
        const int k = std::min(n, d ? (int)(d - b) + 3 : 2);
        memcpy(o, b, (size_t)k), o += k, *o++ = '\t';
    }
    for (int v : {h.aln, h.mis, h.gap, h.qst, h.qed, h.sst, h.sed}) o = put_int(o, v), *o++ = '\t';
    o += f2s_into(h.evalue, o), *o++ = '\t';
    o = put_int(o, h.bit), *o++ = '\t';
    o = put_int(o, h.qlen), *o++ = '\t';
    o = put_int(o, h.slen), *o++ = '\t';
    o = put_int(o, (long long)h.qidx), *o++ = '\t';
    memcpy(o, R.data.data() + R.hd_beg[(size_t)h.sidx], hl), o += hl, *o++ = '\n';
    out.resize((size_t)(o - out.data()));
}

std::string format_hit(so_ctx* c, const so_hit& h) {
    std::vector<char> v;
    format_hit_into(c, h, v);
    return std::string(v.data(), v.size());
}

void file_stamp(const char* path, long long& size, long long& mtime_ns) {
    struct stat sb;
    size = mtime_ns = -1;
    if (stat(path, &sb) == 0) size = (long long)sb.st_size, mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
}

bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)std::max<long>(0, n));
    bool ok = n <= 0 || fread(&out[0], 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// queries: parse, make the raw residues resident, and prepare the device SEG symbol folding
void load_queries_common(so_ctx* c, bool parsed = false) {
    SeqSet& Q = c->qry;
    ++c->qry_gen;
    const double t0 = wall();
    if (!parsed) Q.parse();
    const double t1 = wall();
    const size_t nres = Q.res.size();
    Q.d_res.ensure(nres + 64);
    Q.d_off.ensure((size_t)Q.N + 1);
    if (nres) HIP_CHECK(hipMemcpyAsync(Q.d_res.p, Q.res.data(), nres, hipMemcpyHostToDevice, c->st));
    HIP_CHECK(hipMemcpyAsync(Q.d_off.p, Q.off.data(), ((size_t)Q.N + 1) * sizeof(u32), hipMemcpyHostToDevice, c->st));
    bool raw_present[256];
    byte_presence(Q.res.data(), nres, raw_present);
    u8 up[256], sym[256];
    for (int b = 0; b < 256; ++b) up[b] = (b >= 'a' && b <= 'z') ? (u8)(b - 32) : (u8)b;
    memset(c->q_present, 0, sizeof c->q_present);
    if (c->filter) {
        // masked residues are upper-cased raw bytes plus 'x'
        for (int b = 0; b < 256; ++b)
            if (raw_present[b]) c->q_present[up[b]] = true;
        c->q_present['x'] = true;
        int nsym = 0;
        int id[256];
        for (int b = 0; b < 256; ++b) id[b] = -1;
        for (int b = 0; b < 256; ++b)
            if (raw_present[b] && id[up[b]] < 0) id[up[b]] = nsym++;
        c->seg_on_device = nsym <= 64;
        for (int b = 0; b < 256; ++b) sym[b] = (u8)((id[up[b]] >= 0 && id[up[b]] < 64) ? id[up[b]] : 0);
        const SegTables& T = seg_tables();
        c->d_segtab.ensure(sizeof(SegTables));
        c->d_symmap.ensure(256);
        c->d_upmap.ensure(256);
        HIP_CHECK(hipMemcpyAsync(c->d_segtab.p, &T, sizeof(SegTables), hipMemcpyHostToDevice, c->st));
        HIP_CHECK(hipMemcpyAsync(c->d_symmap.p, sym, 256, hipMemcpyHostToDevice, c->st));
        HIP_CHECK(hipMemcpyAsync(c->d_upmap.p, up, 256, hipMemcpyHostToDevice, c->st));
    } else {
        memcpy(c->q_present, raw_present, sizeof raw_present);
        c->seg_on_device = false;
    }
    HIP_CHECK(hipStreamSynchronize(c->st));
    c->qry_loaded = true;
    c->lt["load.qry_parse"] = (t1 - t0) * 1e3, c->lt["load.qry_h2d"] = (wall() - t1) * 1e3;
}

// per-query seed-hit counts over all chunks (what the lookup kernel will visit): the work estimate used to shard queries
void query_work(so_ctx* c, i64 q_lo, i64 q_hi, u64* out) {
    if (!c->ref_loaded) throw SoError("so_query_work: no reference loaded");
    if (!c->qry_loaded) throw SoError("so_query_work: no queries loaded");
    build_index(c);
    const i64 N = c->qry.N;
    i64 st = std::min<i64>(std::max<i64>(0, q_lo), N), ed = std::min<i64>(q_hi < 0 ? N : q_hi, N);
    if (tune().batch > 0) c->max_batch = (u32)tune().batch;
    const bool prof = c->profile;
    c->profile = false;  // a pre-pass, not part of any timed stage
    try {
        for (i64 b0 = st; b0 < ed; b0 += c->max_batch) {
            const i64 b1 = std::min<i64>(ed, b0 + c->max_batch);
            if (!c->batch) c->batch = std::make_shared<Batch>();
            Batch& b = *static_cast<Batch*>(c->batch.get());
            const so_counters keep = c->cnt;
            prepare_batch(c, b, b0, b1);
            c->cnt = keep;
            for (u32 i = 0; i < b.nq; ++i) out[b0 - st + i] = 0;
            for (int ci = 0; ci < (int)c->chunks.size(); ++ci) {
                ChunkIndex& ch = *c->chunks[ci];
                if (ch.seq_hi == ch.seq_lo || ch.E == 0 || b.nq == 0) continue;
                const unsigned long long* qh = chunk_qhits(c, b, ci);
                if (b.korder_async) chunk_qhits_deferred(c, b, ci);
                for (u32 i = 0; i < b.nq; ++i) out[b0 - st + b.qid[i]] += qh[i];
            }
        }
    } catch (...) {
        c->profile = prof;
        throw;
    }
    c->profile = prof;
}

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

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int so_abi_version(void) { return SOHIT_ABI_VERSION; }

so_ctx* so_create(int device, const so_params* params) {
    so_ctx* c = nullptr;
    try {
        if (!params) throw SoError("so_create: params is NULL");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) throw SoError("so_create: no HIP device available (libsohit has no CPU fallback)");
        if (device < 0 || device >= n) throw SoError("so_create: device index out of range");
        HIP_CHECK(hipSetDevice(device));
        c = new so_ctx();
        c->device = device;
        c->tune.read();
        set_tune(&c->tune);
        g_poison = (int)c->tune.poison;
        {
            int ncu = 0;
            HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
            c->ncu = (u32)std::max(1, ncu);
        }
        memset(&c->cnt, 0, sizeof c->cnt);
        set_params(c, params);
        HIP_CHECK(hipStreamCreate(&c->st));
        HIP_CHECK(hipEventCreate(&c->ev0));
        HIP_CHECK(hipEventCreate(&c->ev1));
        HIP_CHECK(hipStreamCreateWithFlags(&c->st_rows, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_rows_done, hipEventDisableTiming));
        for (auto& e : c->ev_part) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK(hipStreamCreateWithFlags(&c->st_side, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_side_go, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_korder, hipEventDisableTiming));
        HIP_CHECK(hipStreamCreateWithFlags(&c->st_ug, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_ug_go, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_ug_done, hipEventDisableTiming));
        upload_constants(c);
        if (c->tune.warm) c->warm = std::thread(warm_sort_modules, device);
        g_create_err.clear();
        return c;
    } catch (const std::exception& e) {
        g_create_err = e.what();
        delete c;
        return nullptr;
    }
}

void so_destroy(so_ctx* c) {
    if (!c) return;
    if (c->warm.joinable()) c->warm.join();
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_qhits) (void)hipHostFree(c->h_qhits);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
    if (c->ev_rows_done) (void)hipEventDestroy(c->ev_rows_done);
    for (auto& e : c->ev_part)
        if (e) (void)hipEventDestroy(e);
    if (c->st_rows) (void)hipStreamDestroy(c->st_rows);
    if (c->ev_side_go) (void)hipEventDestroy(c->ev_side_go);
    if (c->ev_korder) (void)hipEventDestroy(c->ev_korder);
    if (c->st_side) (void)hipStreamDestroy(c->st_side);
    if (c->ev_ug_go) (void)hipEventDestroy(c->ev_ug_go);
    if (c->ev_ug_done) (void)hipEventDestroy(c->ev_ug_done);
    if (c->st_ug) (void)hipStreamDestroy(c->st_ug);
    if (c->st) (void)hipStreamDestroy(c->st);
    if (&c->tune == &tune()) set_tune(nullptr);
    delete c;
    g_hit_cache.clear();
}

const char* so_last_error(const so_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int so_load_ref(so_ctx* c, const char* path, int64_t r_lo, int64_t r_hi) {
    return guarded(c, [&] {
        if (!read_file(path, c->ref.data)) throw SoError(std::string("cannot read reference FASTA ") + path);
        load_ref_common(c, r_lo, r_hi);
        c->ref_path = path;
        file_stamp(path, c->ref_fsize, c->ref_mtime_ns);
    });
}

int so_load_ref_mem(so_ctx* c, const char* bytes, int64_t n, int64_t r_lo, int64_t r_hi) {
    return guarded(c, [&] {
        c->ref.data.assign(bytes, (size_t)n);
        load_ref_common(c, r_lo, r_hi);
        c->ref_path.clear();
    });
}

int so_build_index(so_ctx* c) {
    return guarded(c, [&] { build_index(c); });
}

int so_load_index(so_ctx* c, const char* prefix) {
    return guarded(c, [&] { load_index(c, prefix); });
}

int so_drop_index(so_ctx* c) {
    return guarded(c, [&] {
        for (auto& ch : c->chunks) c->spare_chunks.push_back(std::move(ch));  // keep the allocations (480 MB `start` per chunk)
        c->chunks.clear();
        c->index_built = false;
    });
}

int so_load_queries(so_ctx* c, const char* path) {
    return guarded(c, [&] {
        // an all-vs-all run names ONE file twice: its parsed form is copied from the reference side instead of read and parsed again
        long long fs = -1, fm = -1;
        if (path) file_stamp(path, fs, fm);
        if (c->ref_loaded && path && c->ref_path == path && fs >= 0 && fs == c->ref_fsize && fm == c->ref_mtime_ns) {
            const SeqSet& R = c->ref;
            SeqSet& Q = c->qry;
            Q.data = R.data, Q.rec = R.rec, Q.hd_beg = R.hd_beg, Q.hd_len = R.hd_len, Q.id_len = R.id_len, Q.N = R.N, Q.off = R.off, Q.res = R.res,
            Q.maxlen = R.maxlen;
            load_queries_common(c, true);
            return;
        }
        if (!read_file(path, c->qry.data)) throw SoError(std::string("cannot read query FASTA ") + path);
        load_queries_common(c);
    });
}

int so_load_queries_mem(so_ctx* c, const char* bytes, int64_t n) {
    return guarded(c, [&] {
        c->qry.data.assign(bytes, (size_t)n);
        load_queries_common(c);
    });
}

int64_t so_num_queries(const so_ctx* c) { return c && c->qry_loaded ? c->qry.N : -1; }
int64_t so_num_refs(const so_ctx* c) { return c && c->ref_loaded ? c->ref.N : -1; }
int64_t so_ref_len(const so_ctx* c, int64_t j) { return (c && c->ref_loaded && j >= 0 && j < c->ref.N) ? (int64_t)c->ref.len(j) : -1; }
int64_t so_query_len(const so_ctx* c, int64_t q) { return (c && c->qry_loaded && q >= 0 && q < c->qry.N) ? (int64_t)c->qry.len(q) : -1; }

int so_search_loaded(so_ctx* c, int64_t q_lo, int64_t q_hi, so_hit** hits, int64_t* n_hits) {
    return guarded(c, [&] {
        if (!hits || !n_hits) throw SoError("so_search: output pointers are NULL");
        *hits = nullptr;
        *n_hits = 0;
        HitBuf out;
        search_loaded(c, q_lo, q_hi, out);
        *n_hits = (int64_t)out.n;
        *hits = out.release();
    });
}

int so_search(so_ctx* c, const char* qry_path, int64_t q_lo, int64_t q_hi, so_hit** hits, int64_t* n_hits) {
    int rc = so_load_queries(c, qry_path);
    if (rc) return rc;
    return so_search_loaded(c, q_lo, q_hi, hits, n_hits);
}

void so_free_hits(so_hit* hits) { g_hit_cache.give(hits); }

int so_search_device(so_ctx* c, int64_t q_lo, int64_t q_hi, const so_hit** d_hits, int64_t* n_hits) {
    return guarded(c, [&] {
        if (!n_hits) throw SoError("so_search_device: n_hits is NULL");
        *n_hits = 0;
        if (d_hits) *d_hits = nullptr;
        c->d_hits_n = 0;
        c->dev_out = true;
        HitBuf none;
        try {
            search_loaded(c, q_lo, q_hi, none);
        } catch (...) {
            c->dev_out = false;
            throw;
        }
        c->dev_out = false;
        HIP_CHECK(hipStreamSynchronize(c->st));
        *n_hits = (int64_t)c->d_hits_n;
        if (d_hits) *d_hits = (const so_hit*)c->d_hits.p;
    });
}

int so_device_hits_copy(so_ctx* c, void* dst_device, int64_t n_hits) {
    return guarded(c, [&] {
        if (n_hits < 0 || (size_t)n_hits > c->d_hits_n) throw SoError("so_device_hits_copy: more records requested than the last so_search_device produced");
        if (n_hits && !dst_device) throw SoError("so_device_hits_copy: destination is NULL");
        if (n_hits) HIP_CHECK(hipMemcpyAsync(dst_device, c->d_hits.p, (size_t)n_hits * sizeof(so_hit), hipMemcpyDeviceToDevice, c->st));
        HIP_CHECK(hipStreamSynchronize(c->st));
    });
}

int so_query_work(so_ctx* c, int64_t q_lo, int64_t q_hi, uint64_t* work) {
    return guarded(c, [&] {
        if (!work) throw SoError("so_query_work: output is NULL");
        query_work(c, q_lo, q_hi, work);
    });
}

int64_t so_format_hit(so_ctx* c, const so_hit* hit, char* buf, int64_t cap) {
    int64_t need = -1;
    guarded(c, [&] {
        std::string r = format_hit(c, *hit);
        need = (int64_t)r.size();
        if (buf && cap > 0) {
            size_t k = std::min<size_t>(r.size(), (size_t)cap - 1);
            memcpy(buf, r.data(), k);
            buf[k] = 0;
        }
    });
    return need;
}

int so_write_sc(so_ctx* c, const so_hit* hits, int64_t n, const char* path, const char* mode) {
    return guarded(c, [&] {
        FILE* f = fopen(path, (mode && mode[0] == 'a') ? "ab" : "wb");
        if (!f) throw SoError(std::string("cannot open output ") + path);
        // Rows are formatted in slabs of 16384 by a few threads that take slabs in order from a counter; the calling thread writes
        // every slab as soon as it and all slabs before it are done, so formatting and writing overlap and the threads live as long
        // as the call (round 3 started eight threads per 262144 rows and wrote between the groups).  A formatter waits while it is more
        // than 4 * nt slabs ahead of the writer: with a slow disk the text of a 300 M-row result would otherwise pile up in memory.
        const int64_t SLAB = 16384;
        const int64_t nslab = (n + SLAB - 1) / SLAB;
        const unsigned nt = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min((unsigned)std::max(1ll, tune().write_threads), std::max(1u, std::thread::hardware_concurrency())), nslab));
        std::vector<std::vector<char>> bufs((size_t)nslab);
        std::vector<std::atomic<int>> ready((size_t)nslab);
        for (auto& r : ready) r.store(0);
        std::atomic<int64_t> next(0), written(0);
        std::atomic<bool> failed(false);
        std::exception_ptr err;
        std::mutex mu;
        std::vector<std::thread> th;
        const int64_t ahead = 4 * (int64_t)nt;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&] {
                for (;;) {
                    const int64_t k = next.fetch_add(1);
                    if (k >= nslab || failed.load()) break;
                    while (k - written.load(std::memory_order_acquire) >= ahead && !failed.load()) std::this_thread::yield();
                    if (failed.load()) break;
                    try {
                        std::vector<char>& b = bufs[(size_t)k];
                        const int64_t lo = k * SLAB, hi = std::min<int64_t>(n, lo + SLAB);
                        b.reserve((size_t)(hi - lo) * 128);
                        for (int64_t i = lo; i < hi; ++i) format_hit_into(c, hits[i], b);
                    } catch (...) {
                        std::lock_guard<std::mutex> g(mu);
                        if (!err) err = std::current_exception();
                        failed.store(true);
                    }
                    ready[(size_t)k].store(1, std::memory_order_release);
                }
            });
        bool ok = true;
        for (int64_t k = 0; k < nslab && !failed.load(); ++k) {
            while (!ready[(size_t)k].load(std::memory_order_acquire) && !failed.load()) std::this_thread::yield();
            if (failed.load()) break;
            std::vector<char>& b = bufs[(size_t)k];
            if (ok && !b.empty()) ok = fwrite(b.data(), 1, b.size(), f) == b.size();
            std::vector<char>().swap(b);
            written.store(k + 1, std::memory_order_release);
        }
        failed.store(failed.load() || !ok);   // (a short write: let waiting formatters go)
        for (auto& x : th) x.join();
        if (fclose(f) != 0) ok = false;
        if (err) std::rethrow_exception(err);
        if (!ok) throw SoError(std::string("short write to ") + path + " (disk full or I/O error)");
    });
}

// "%f" of v[0..n) (the library's own exact formatter) and f2s (fsearch.py:43-61) of the same values, one per line: "<%f>\t<f2s>\n".
// Host-only (no ctx, no GPU): lets the CPU tests compare the formatter with printf over millions of values.
int64_t so_fmt_rows(const double* v, int64_t n, char* out, int64_t cap) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (cap - w < 1400) return -1;
        w += fmt_f6(v[i], out + w);
        out[w++] = '\t';
        w += f2s_into(v[i], out + w);
        out[w++] = '\n';
    }
    return w;
}

// One switch of tune.h's table by its environment name (with or without the SOHIT_ prefix), for this context from now on.
int so_set_option(so_ctx* c, const char* name, const char* value) {
    return guarded(c, [&] {
        if (!name || !value) throw SoError("so_set_option: name or value is NULL");
        std::string n = name;
        if (n.rfind("SOHIT_", 0) != 0) n = "SOHIT_" + n;
        bool found = false;
#define X_B(f) c->tune.f = atoi(value) != 0;
#define X_I(f) c->tune.f = strtoll(value, nullptr, 0);
#define X_D(f) c->tune.f = atof(value);
#define X_P(f) c->tune.f = atoi(value) != 0;
#define X(kind, field, env, dflt, text) \
    if (!found && n == env) {           \
        X_##kind(field) found = true;   \
    }
        SOHIT_TUNE_TABLE(X)
#undef X
#undef X_B
#undef X_I
#undef X_D
#undef X_P
        if (!found) throw SoError("so_set_option: unknown switch " + n);
        g_poison = (int)c->tune.poison;
    });
}

int so_set_profile(so_ctx* c, int on) {
    if (!c) return 1;
    c->profile = on != 0;
    return 0;
}

int64_t so_bucket_count(const so_ctx* c) { return c ? c->nc : -1; }

int so_get_counters(const so_ctx* c, so_counters* out) {
    if (!c || !out) return 1;
    *out = c->cnt;
    return 0;
}

int64_t so_timing_report(const so_ctx* c, char* buf, int64_t cap) {
    if (!c) return -1;
    std::string s;
    for (const auto* m : {&c->tm, &c->lt})
        for (auto& kv : *m) {
            char tmp[128];
            snprintf(tmp, sizeof tmp, "%s=%.3f;", kv.first.c_str(), kv.second);
            s += tmp;
        }
    if (buf && cap > 0) {
        size_t k = std::min<size_t>(s.size(), (size_t)cap - 1);
        memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return (int64_t)s.size();
}

int so_reset_counters(so_ctx* c) {
    if (!c) return 1;
    c->tm.clear();
    so_counters keep = c->cnt;
    memset(&c->cnt, 0, sizeof c->cnt);
    c->cnt.ref_seqs = keep.ref_seqs, c->cnt.ref_aa = keep.ref_aa, c->cnt.n_chunks = keep.n_chunks;
    c->cnt.index_entries = keep.index_entries;
    return 0;
}

int64_t so_chunk_threshold(const so_ctx* c, int64_t k) { return (c && k >= 0 && k < (int64_t)c->chunks.size()) ? c->chunks[k]->threshold : -1; }
int64_t so_chunk_entries(const so_ctx* c, int64_t k) { return (c && k >= 0 && k < (int64_t)c->chunks.size()) ? (int64_t)c->chunks[k]->E : -1; }

int so_chunk_download(so_ctx* c, int64_t k, uint32_t* start, uint64_t* entries) {
    return guarded(c, [&] {
        if (k < 0 || k >= (int64_t)c->chunks.size()) throw SoError("so_chunk_download: no such chunk");
        ChunkIndex& ch = *c->chunks[k];
        if (start) {  // the reference's direct-addressed start[NC + 1], rebuilt from the occupied-bucket list
            std::vector<u32> ub(ch.U), ubeg((size_t)ch.U + 1);
            if (ch.U) {
                HIP_CHECK(hipMemcpy(ub.data(), ch.ub.p, (size_t)ch.U * sizeof(u32), hipMemcpyDeviceToHost));
                HIP_CHECK(hipMemcpy(ubeg.data(), ch.ubeg.p, ((size_t)ch.U + 1) * sizeof(u32), hipMemcpyDeviceToHost));
            }
            size_t k = 0;
            for (size_t bk = 0; bk <= (size_t)c->nc; ++bk) {
                while (k < ch.U && ub[k] < bk) ++k;  // start[b] = first slot of the first occupied bucket >= b
                start[bk] = k < ch.U ? ubeg[k] : ch.E;
            }
        }
        if (entries && ch.E) HIP_CHECK(hipMemcpy(entries, ch.entries.p, (size_t)ch.E * sizeof(u64), hipMemcpyDeviceToHost));
    });
}

int64_t so_masked_query(so_ctx* c, int64_t q, char* buf, int64_t cap) {
    if (!c) return -1;
    i64 k = q - c->masked_lo;
    if (k < 0 || k >= (i64)c->masked.size()) return -1;
    const std::string& s = c->masked[(size_t)k];
    if (buf && cap > 0) memcpy(buf, s.data(), std::min<size_t>(s.size(), (size_t)cap));
    return (int64_t)s.size();
}

int64_t so_query_candidates(so_ctx* c, int64_t q, uint32_t* out4, int64_t cap) {
    if (!c) return -1;
    i64 k = q - c->last_q_lo;
    if (k < 0 || k >= (i64)c->last_cands.size()) return -1;
    const auto& v = c->last_cands[(size_t)k];
    const int64_t n = (int64_t)v.size() / 4;
    if (out4)
        for (int64_t i = 0; i < std::min(n, cap) * 4; ++i) out4[i] = v[(size_t)i];
    return n;
}

}  // extern "C"
