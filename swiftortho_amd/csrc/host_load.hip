// host_load.hip -- libsohit.so host side: parameters, constant tables, FASTA sets resident in HBM, SEG masking on the host (see host.h).
#include "host.h"

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


// the switches the launch helpers see: those of the context whose API call runs (or ran last) on THIS thread -- guarded() sets them per call,
// so two contexts driven from two threads each see their own (so_free_hits, which has no context, follows the calling thread's last one)
static thread_local const Tune* t_tune = nullptr;
const Tune& tune() {
    static const Tune dflt;
    return t_tune ? *t_tune : dflt;
}
void set_tune(const Tune* t) { t_tune = t; }
thread_local int g_poison = -1;


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
void upload_set(so_ctx* c, SeqSet& s, const u8* residues, const std::vector<u32>& off, u32 nseq, const bool* present_in) {
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
    ++c->ref_gen;
    c->band_plans.clear();
    c->index_built = false;
    c->chunks.clear();
    c->cnt.ref_seqs = c->ref.N;
    c->cnt.ref_aa = (i64)c->ref.res.size();
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
void load_queries_common(so_ctx* c, bool parsed) {
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
