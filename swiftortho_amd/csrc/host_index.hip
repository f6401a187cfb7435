// host_index.hip -- libsohit.so host side: per-chunk index build, Fasta.load of the reference's index files, banded diagonal ids (see host.h).
#include "host.h"


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
            ch = std::move(c->spare_chunks.front());   // (in order: chunk k gets chunk k's object back, with its band layouts)
            c->spare_chunks.erase(c->spare_chunks.begin());
            ch->d_sh_subj = ch->d_sh_diag = -1;  // key addends belong to the old entries
            for (auto& e : ch->encs) e->stale = true;   // (band_encoding re-encodes the entries; the layout survives when the sequences are the same)
            ch->ordered = false;
            for (auto& t : ch->rtabs) t->k = -1;       // (range boundaries belong to the old entries; the buffers are kept)
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
        // 1. (bucket, entry) pairs, ONE pass: position p's A x S windows at the fixed slots (p - p_lo) * AS + tag, an invalid window as bucket ~0
        //    (until round 6: windows counted per position, scanned, emitted densely -- a counting pass, a scan and a host round trip more)
        const u32 npos = ch->p_hi - ch->p_lo;
        const int AS = c->cfg.A * c->cfg.S;
        const u64 nslots64 = (u64)npos * (u64)AS;
        if (nslots64 >= (1ull << 31)) throw SoError("chunk holds 2^31 seed-window slots or more; lower -c");
        const u32 nslots = (u32)nslots64;
        u32 E = 0;
        ch->U = 0;
        u64 s2 = 0;
        if (nslots) {
            c->ix_bkt.ensure((size_t)nslots + 4), c->ix_ent.ensure((size_t)nslots + 4);
            c->ix_plan.ensure(ixsort_plan_elems((u32)NC) + 8);
            c->d_scan_tmp.ensure(scan_u32_temp_elems(std::max<size_t>(ixsort_plan_elems((u32)NC), (size_t)nslots + 1)) + 8);
            dlap("alloc pairs / plan");
            launch_index_windows_sparse(c->ref.d_words.p, c->ref.d_pseq.p, c->ref.d_off.p, ch->p_lo, ch->p_hi, c->ref.Ppad, (u32)ch->seq_lo, c->cfg, c->ref.lut,
                                        (u32)c->step, c->ix_bkt.p, c->ix_ent.p, c->st);
            // 2. group by bucket id (ascending): the slot layout of the reference's CSR.  Hand-written: two counting passes, k_ixsort.hip; the
            //    first one's scan leaves the number of entries; members of a bucket land in no particular order (order_chunk)
            E = d2h_u32(c, ixsort_count(c->ix_bkt.p, nslots, (u32)NC, c->ix_plan.p, c->d_scan_tmp.p, c->st));
            dlap("windows + level-1 count");
        }
        if ((u64)E >= (1ull << 29)) throw SoError("chunk index exceeds 2^29 entries (the lookup kernel addresses 8-byte slots with 32-bit byte offsets); lower -c");
        ch->E = E;
        ch->entries.ensure((size_t)E + 4);
        if (E) {
            c->ix_bkt2.ensure((size_t)E + 4), c->ix_tk.ensure((size_t)E + 4), c->ix_tv.ensure((size_t)E + 4);
            dlap("alloc entries / grouping scratch");
            ixsort_finish(c->ix_bkt.p, c->ix_ent.p, nslots, (u32)NC, c->ix_plan.p, c->ix_tk.p, c->ix_tv.p, c->ix_bkt2.p, ch->entries.p, c->st);
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
        if (e->stale && (e->ref_gen != c->ref_gen || e->seq_lo != ch.seq_lo || e->seq_hi != ch.seq_hi)) e->k = -1, e->stale = false;   // other sequences: nothing to keep
    for (auto& e : ch.encs)
        if (e->k >= 0 && e->ba == ba && e->bp == bp && e->multi_ok == multi_ok) {
            e->used = ch.enc_clock;
            if (e->stale && e->k > 0) {   // same sequences, new entries: the addends alone (no host loop, no upload, no round trip)
                e->dk32.ensure((size_t)ch.E + 4);
                launch_encode_band32(ch.entries.p, ch.E, ba, e->gbase.p, c->ref.d_off.p + ch.seq_lo, e->dk32.p, c->st);
            }
            e->stale = false;
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
    e->stale = false, e->ref_gen = c->ref_gen, e->seq_lo = ch.seq_lo, e->seq_hi = ch.seq_hi;
    if (ceil_log2(std::max<u64>(nband, 2)) + k + ba > 31) {
        e->k = 0;
        return nullptr;
    }
    e->k = k, e->nband = (u32)nband, e->C = (u32)((1ull << k) - qcap), e->multi = nband != nseq;
    std::vector<u32> gbase((size_t)nseq + 1);
    std::vector<u64> btab;
    if (e->multi) btab.resize((size_t)nband);
    u32 band = 0;
    e->spans.clear();
    for (u32 j = 0; j < nseq; ++j) {
        const u32 sl = c->ref.len(ch.seq_lo + j);
        const bool one = sl <= e->C;
        const u32 nb = one ? 1u : (u32)(((u64)sl + qcap + (1ull << k) - 1) >> k);
        if (nb > 1) e->spans.emplace_back(band, nb);
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


// ---------------------------------------------------------------------------------------------
// The count pass without the index entries (k_bucket.hip: k_rtab_build, k_bkt_count_tab)
// ---------------------------------------------------------------------------------------------
// Members of every bucket in descending entry order = subject descending (the order the reference's CSR has, which the hand-written
// grouping of the build gave up): the entries of one band range become one run of the bucket.  Done at the chunk's first dense pass;
// every band encoding's addends follow the slots, so they are encoded again when next used (`stale`).  The smallest member of the last
// bucket ends up in the slot the reference never reads, where k_index_fixlast had put it.
void order_chunk(so_ctx* c, ChunkIndex& ch) {
    if (ch.ordered) return;
    if (ch.E && ch.U) {
        const u32 nseq = (u32)std::max<i64>(1, ch.seq_hi - ch.seq_lo);
        const int end_bit = std::min(63, 32 + ceil_log2((u64)nseq + 1));
        c->ix_ent.ensure((size_t)ch.E + 4);
        ensure_sort_tmp(c, sort_keys_u64_seg_desc_temp_bytes(ch.E, ch.U, 0, end_bit));
        sort_keys_u64_seg_desc(c->d_sort_tmp.p, c->d_sort_tmp.cap, ch.entries.p, c->ix_ent.p, ch.E, ch.U, ch.ubeg.p, 0, end_bit, c->st);
        HIP_CHECK(hipMemcpyAsync(ch.entries.p, c->ix_ent.p, (size_t)ch.E * sizeof(u64), hipMemcpyDeviceToDevice, c->st));
        for (auto& e : ch.encs) e->stale = true;
        ch.d_sh_subj = ch.d_sh_diag = -1;
        for (auto& t : ch.rtabs) t->k = -1;
    }
    ch.ordered = true;
}

// Range boundaries of every occupied bucket for band encoding `e` and ranges of 2^wb bands; nullptr-like (ok == false) when the table
// cannot stand in for the counting pass: a subject whose bands straddle a range boundary (its hits' range would depend on the query
// position), a bucket above 65535 entries, a table above 1 GiB.  Up to four tables per chunk are kept (least recently used first out).
const ChunkIndex::RangeTab* range_table(so_ctx* c, ChunkIndex& ch, const ChunkIndex::BandEnc& e, int wb, u32 R) {
    ++ch.enc_clock;
    for (auto& t : ch.rtabs)
        if (t->k == e.k && t->ba == e.ba && t->bp == e.bp && t->multi_ok == e.multi_ok && t->wb == wb && t->R == R) {
            t->used = ch.enc_clock;
            return t.get();
        }
    ChunkIndex::RangeTab* t = nullptr;
    for (auto& x : ch.rtabs)
        if (x->k < 0) t = x.get();
    if (!t && ch.rtabs.size() < 4) {
        ch.rtabs.push_back(std::make_unique<ChunkIndex::RangeTab>());
        t = ch.rtabs.back().get();
    }
    if (!t) {
        t = ch.rtabs[0].get();
        for (auto& x : ch.rtabs)
            if (x->used < t->used) t = x.get();
    }
    t->ba = e.ba, t->bp = e.bp, t->k = e.k, t->multi_ok = e.multi_ok, t->wb = wb, t->R = R, t->used = ch.enc_clock, t->ok = false;
    if (!ch.ordered || !ch.E || !ch.U || e.ba != 0) return t;
    for (const auto& sp : e.spans)
        if ((sp.first >> wb) != ((sp.first + sp.second - 1) >> wb)) return t;   // a subject's bands in two ranges
    const size_t cells = (size_t)ch.U * ((size_t)R + 1);
    if (cells * sizeof(u16) > ((size_t)1 << 30)) return t;
    t->tab.ensure(cells + 8);
    ch.row_of_slot.ensure((size_t)ch.E + 4);
    c->d_small.ensure(32);
    HIP_CHECK(hipMemsetAsync(c->d_small.p + 14, 0, sizeof(u32), c->st));
    launch_rtab_build(e.dk32.p, ch.E, ch.ubeg.p, ch.U, wb + e.k, R, t->tab.p, ch.row_of_slot.p, c->d_small.p + 14, c->st);
    const u32 flag = d2h_u32(c, c->d_small.p + 14);
    if (flag && tune().debug) fprintf(stderr, "[sohit] range table refused (flag %u): counting pass\n", flag);
    t->ok = flag == 0;
    return t;
}
