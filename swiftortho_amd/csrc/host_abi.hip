// host_abi.hip -- libsohit.so host side: row formatting and the C ABI of include/sohit.h (see host.h).
#include "host.h"

std::string g_create_err;


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
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_bnd_go, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->ev_bnd_done, hipEventDisableTiming));
        upload_constants(c);
        c->warm = std::thread(warm_sort_modules, device);
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
    if (c->ev_bnd_go) (void)hipEventDestroy(c->ev_bnd_go);
    if (c->ev_bnd_done) (void)hipEventDestroy(c->ev_bnd_done);
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
        order_chunk(c, ch);   // (members of every bucket in the reference's CSR order, if no dense pass has asked for that yet)
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
        if (entries && ch.E) {
            HIP_CHECK(hipStreamSynchronize(c->st));
            HIP_CHECK(hipMemcpy(entries, ch.entries.p, (size_t)ch.E * sizeof(u64), hipMemcpyDeviceToHost));
        }
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

