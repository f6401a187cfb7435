// host_seed.hip -- libsohit.so host side: batch preparation and the seed stage -- bucket bounds, frequency cap, lookup + diagonal binning, ungapped extension, best diagonal, candidate order (see host.h).
#include "host.h"

// The k-mer orders of the batch's queries (fsearch.py:2660-2666: windows by self score, the reference's own quicksort replayed: 1.3 ms
// for 100 k queries).  Only the frequency cap reads them, and only for a query whose windows together exceed its limit in the chunk at
// hand (k_cap_all leaves it open).  chunk_qhits asks for the orders of the open queries alone while they are a minority of the batch
// (order_open_queries: once per query and batch) and for everybody's at the first chunk where they are not (order_queries;
// SOHIT_KSC_LAZY=0: prepare_batch does).  Either way the queries that need global scratch -- in a class-ordered batch the tail of the
// last class -- have theirs computed on the side stream while the classes before them are searched (korder_async, q_defer).
static void defer_long_class(so_ctx* c, Batch& b, u32 q_long) {
    b.korder_async = q_long < b.nq && b.cls_sorted && b.qcls[q_long] != b.qcls[0];
    if (!b.korder_async) return;
    b.q_defer = q_long;
    while (b.q_defer > 0 && b.qcls[b.q_defer - 1] == b.qcls[q_long]) --b.q_defer;
    HIP_CHECK(hipEventRecord(c->ev_side_go, c->st));   // the batch's class arrays (and this chunk's list) are on the device
    HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_side_go, 0));
}

void order_queries(so_ctx* c, Batch& b) {
    const u32 q_long = b.ksc_long;
    const size_t nres = b.h_off[b.nq];
    if (q_long < b.nq) b.gx.ensure(nres + 4), b.gL.ensure(nres + 4), b.gR.ensure(nres + 4);
    defer_long_class(c, b, q_long);
    launch_ksc_order(b.dev.d_scls.p, b.dev.d_off.p, b.nq, q_long, c->cfg.mink, c->d_b62c.p, b.gx.p, b.gL.p, b.gR.p, b.korder.p, c->st,
                     b.korder_async ? c->st_side : c->st);
    if (b.korder_async) HIP_CHECK(hipEventRecord(c->ev_korder, c->st_side));
    b.korder_ready = true;
}

// ... of the b.n_open queries in b.open_list (qh: the host's copy of b.qhits, ~0 for those)
void order_open_queries(so_ctx* c, Batch& b, const unsigned long long* qh) {
    u32 q_long = b.nq;   // the first open query that needs global scratch
    for (u32 q = b.ksc_long; q < b.nq && q_long == b.nq; ++q)
        if (qh[q] == ~0ull && (i64)(b.h_off[q + 1] - b.h_off[q]) - c->cfg.mink + 1 > (i64)ksc_lds_max()) q_long = q;
    const size_t nres = b.h_off[b.nq];
    if (q_long < b.nq) b.gx.ensure(nres + 4), b.gL.ensure(nres + 4), b.gR.ensure(nres + 4);
    if (!b.kord_have_clear) {
        b.kord_have.ensure((size_t)b.nq + 4);
        HIP_CHECK(hipMemsetAsync(b.kord_have.p, 0, b.nq, c->st));
        b.kord_have_clear = true;
    }
    defer_long_class(c, b, q_long);
    launch_ksc_order_list(b.dev.d_scls.p, b.dev.d_off.p, b.nq, b.open_list.p, b.n_open, b.kord_have.p, q_long, c->cfg.mink, c->d_b62c.p, b.gx.p, b.gL.p,
                          b.gR.p, b.korder.p, c->st, b.korder_async ? c->st_side : c->st);
    if (b.korder_async) HIP_CHECK(hipEventRecord(c->ev_korder, c->st_side));
}

void prepare_batch(so_ctx* c, Batch& b, i64 q_lo, i64 q_hi) {
    if (b.korder_async || b.bnd_ci >= 0) HIP_CHECK(hipStreamSynchronize(c->st_side));   // (a batch that never reached its long queries' passes / its last chunks)
    b.korder_async = false;
    b.bnd_ci = -1;
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
    b.blkmax.assign(((size_t)b.nq + 255) / 256, 0u);
    for (int k = 0; k < QCLASSES; ++k) b.cls_maxq[k] = 0;
    b.cls_sorted = true;
    for (u32 i = 0; i < b.nq; ++i) {
        u32 ln = Q.len(q_lo + b.qid[i]);
        b.h_off[i + 1] = b.h_off[i] + ln;
        b.maxqlen = std::max(b.maxqlen, ln);
        b.blkmax[i >> 8] = std::max(b.blkmax[i >> 8], ln);
        b.cls_maxq[b.qcls[i]] = std::max(b.cls_maxq[b.qcls[i]], ln);
        if (i && b.qcls[i] < b.qcls[i - 1]) b.cls_sorted = false;
    }
    {
        u32 at = 0;
        for (int k = 0; k < QCLASSES; ++k) {
            b.cls_start[k] = at;
            while (b.cls_sorted && at < b.nq && b.qcls[at] == k) ++at;
        }
        b.cls_start[QCLASSES] = b.nq;
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
            const bool ordered = b.cls_sorted;
            if (ordered && tune().qclass) {
                q_mid = b.cls_start[2];      // classes 0, 1: below 1024 residues
                q_long = b.cls_start[4];     // class 4: 4096 and more
            }
            const bool seg_aside = b.maxqlen > 4096;
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
        for (u32 blk = 0; blk < (u32)b.blkmax.size() && q_long == b.nq; ++blk) {   // (the first slot whose query is too long: block maxima first)
            if ((i64)b.blkmax[blk] - c->cfg.mink + 1 <= (i64)ksc_lds_max()) continue;
            for (u32 i = blk << 8; i < std::min<u32>(b.nq, (blk + 1) << 8); ++i)
                if ((i64)(b.h_off[i + 1] - b.h_off[i]) - c->cfg.mink + 1 > (i64)ksc_lds_max()) {
                    q_long = i;
                    break;
                }
        }
        b.ksc_long = q_long;
        b.korder_ready = false, b.kord_have_clear = false, b.n_open = 0;
        if (!tune().ksc_lazy) order_queries(c, b);
    }
    b.sbeg.ensure(T), b.scnt.ensure(T), b.hoff.ensure(T + 4), b.cidx.ensure(T + 4);   // (eff / nz: only with several alphabets or patterns, seed_pass)
    b.pcnt.ensure(Ppad), b.mark.ensure(Ppad);
    b.counters.ensure(8);
    b.ucount.ensure(4);
    HIP_CHECK(hipMemsetAsync(b.ucount.p, 0, 4 * sizeof(unsigned long long), c->st));
    c->d_scan_tmp.ensure(std::max(scan_u32_temp_elems(std::max<size_t>(T, (size_t)c->nc + 1)), effscan_temp_elems(T)) + 8);
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
    c->d_small.ensure(32);
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

// bucket bounds of every query window in chunk ci, frequency cap, and the number of index entries each query of the
// batch will visit there (pinned host array, valid until the next call)
const unsigned long long* chunk_qhits(so_ctx* c, Batch& b, int ci) {
    ChunkIndex& ch = *c->chunks[ci];
    const int AS = c->cfg.A * c->cfg.S;
    const u32 Ppad = b.dev.Ppad, NC = (u32)c->nc;
    if (b.bnd_ci == ci) {   // computed on the side stream beside the previous chunk's seed stage: take the second set of arrays
        HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_bnd_done, 0));
        std::swap(b.sbeg.p, b.sbeg2.p), std::swap(b.sbeg.cap, b.sbeg2.cap);
        std::swap(b.scnt.p, b.scnt2.p), std::swap(b.scnt.cap, b.scnt2.cap);
        std::swap(b.pcnt.p, b.pcnt2.p), std::swap(b.pcnt.cap, b.pcnt2.cap);
    } else {
        if (b.bnd_ci >= 0) HIP_CHECK(hipStreamWaitEvent(c->st, c->ev_bnd_done, 0));   // (another chunk's: its arrays are simply not used)
        ProfTimer pt(c, &c->cnt.bounds_ms, &c->cnt.bounds_launches);
        launch_bounds(b.qbucket.p, Ppad, AS, ch.hkey.p, ch.hval.p, ch.hshift, ch.hmask, ch.use_dir ? ch.dir.p : nullptr, ch.ubeg.p, NC, ch.E, b.sbeg.p,
                      b.scnt.p, b.pcnt.p, c->st);
        pt.stop();
        if (c->profile) c->cnt.bounds_bytes += (i64)8 * AS * (i64)b.h_off[b.nq];
    }
    b.bnd_ci = -1;
    if (tune().bounds_ahead && !c->profile) {   // (the stage clocks want every kernel on the batch's stream)
        int nx = ci + 1;
        while (nx < (int)c->chunks.size() && (c->chunks[nx]->seq_hi == c->chunks[nx]->seq_lo || c->chunks[nx]->E == 0)) ++nx;
        if (nx < (int)c->chunks.size()) {
            ChunkIndex& cn = *c->chunks[nx];
            const size_t T = (size_t)AS * Ppad;
            b.sbeg2.ensure(T), b.scnt2.ensure(T), b.pcnt2.ensure(Ppad);
            HIP_CHECK(hipEventRecord(c->ev_bnd_go, c->st));   // (what was queued so far may still read the second set: it held the previous chunk's bounds)
            HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_bnd_go, 0));
            launch_bounds(b.qbucket.p, Ppad, AS, cn.hkey.p, cn.hval.p, cn.hshift, cn.hmask, cn.use_dir ? cn.dir.p : nullptr, cn.ubeg.p, NC, cn.E, b.sbeg2.p, b.scnt2.p,
                          b.pcnt2.p, c->st_side);
            HIP_CHECK(hipEventRecord(c->ev_bnd_done, c->st_side));
            b.bnd_ci = nx;
        }
    }
    HIP_CHECK(hipMemsetAsync(b.mark.p, 0, Ppad, c->st));
    i64 threshold = ch.threshold;
    if (c->thr >= 1 || threshold == 0) threshold = c->thr;  // `thr < 1 and DB.threshold or thr`, fsearch.py:2992
    b.qhits.ensure((size_t)b.nq + 2);
    if (c->h_qhits_cap < (size_t)b.nq + 1) {  // pinned: a pageable read of this array costs more than the kernels around it
        if (c->h_qhits) (void)hipHostFree(c->h_qhits);
        c->h_qhits_cap = (size_t)b.nq + 1024;
        HIP_CHECK(hipHostMalloc((void**)&c->h_qhits, c->h_qhits_cap * sizeof(unsigned long long), hipHostMallocDefault));
    }
    unsigned long long* qh = c->h_qhits;
    if (!b.korder_ready) {
        // no orders yet: a query that stays below its cap keeps all its windows (k_cap_all).  The others get their orders now and the ordered
        // cap -- they alone while they are a minority of the batch (the long seeds: a few per cent); at the first chunk where they are not,
        // every query's order is computed and the batch's later chunks go to the ordered cap at once
        b.open_list.ensure((size_t)b.nq + 4);
        launch_cap_all(b.dev.d_off.p, b.nq, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, b.qhits.p + b.nq, b.open_list.p, c->st);
        HIP_CHECK(hipMemcpyAsync(qh, b.qhits.p, ((size_t)b.nq + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->st));
        HIP_CHECK(hipStreamSynchronize(c->st));
        const unsigned long long open = qh[b.nq];
        if (tune().debug) fprintf(stderr, "[sohit] chunk %d: %llu of %u queries reach their frequency cap\n", ci, open, b.nq);
        b.n_open = 0;
        if (!open) return qh;
        if (open * 2 <= b.nq) {
            if (c->profile) c->tm["seed.kmer_orders_open_chunks"] += (double)(1u << std::min(ci, 30));   // (tests: which chunks asked; one batch per search there)
            b.n_open = (u32)open;
            order_open_queries(c, b, qh);
            const u32 n1 = b.korder_async ? b.q_defer : b.nq;   // (the open queries of the last length class follow in chunk_qhits_deferred)
            launch_cap(b.korder.p, b.dev.d_off.p, 0, n1, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, b.open_list.p, b.n_open, c->st);
            HIP_CHECK(hipMemcpyAsync(qh, b.qhits.p, (size_t)n1 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->st));
            HIP_CHECK(hipStreamSynchronize(c->st));
            return qh;
        }
        if (c->profile) c->tm["seed.kmer_orders_all_at_chunk"] += ci + 1;
        order_queries(c, b);
        HIP_CHECK(hipMemsetAsync(b.mark.p, 0, Ppad, c->st));
    }
    const u32 n1 = b.korder_async ? b.q_defer : b.nq;   // (the last length class follows in chunk_qhits_deferred)
    launch_cap(b.korder.p, b.dev.d_off.p, 0, n1, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, nullptr, 0, c->st);
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
    // (while only the open queries have orders -- !korder_ready -- the others of the class keep what k_cap_all gave them)
    launch_cap(b.korder.p, b.dev.d_off.p, b.q_defer, b.nq, c->cfg.mink, b.pcnt.p, threshold, b.mark.p, b.qhits.p, b.korder_ready ? nullptr : b.open_list.p,
               b.korder_ready ? 0 : b.n_open, c->st);
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
    unsigned long long cls_hits[QCLASSES] = {0};
    bool whole_classes = false;
    {
        const bool merge_on = tune().pass_merge;
        unsigned long long hits[QCLASSES] = {0}, cnt[QCLASSES] = {0};
        u32 maxq[QCLASSES] = {0};
        const u32 known = b.korder_async ? b.q_defer : b.nq;   // (the deferred class's counts arrive later)
        if (b.cls_sorted && known == b.nq) {   // the classes are slot ranges: their sizes and longest queries came with the layout
            for (int k = 0; k < QCLASSES; ++k) {
                cnt[k] = b.cls_start[k + 1] - b.cls_start[k], maxq[k] = b.cls_maxq[k];
                unsigned long long h = 0;
                for (u32 q = b.cls_start[k]; q < b.cls_start[k + 1]; ++q) h += qh[q];
                hits[k] = h;
            }
        } else {
            for (u32 q = 0; q < known; ++q) {
                const int k = b.qcls[q];
                hits[k] += qh[q], cnt[k]++, maxq[k] = std::max(maxq[k], b.h_off[q + 1] - b.h_off[q]);
            }
        }
        whole_classes = b.cls_sorted && known == b.nq;   // a pass may take a class in one step when all of it fits
        for (int k = 0; k < QCLASSES; ++k) cls_hits[k] = hits[k];
        {   // Ordering the chunk's buckets and building a table of range boundaries costs ~1.2 ms per 50 k-sequence chunk (once per index build)
            // and saves 0.58 ms per 10^9-hit pass: worth it when this batch alone brings the chunk two passes (the 10 k-protein config 2 has
            // one pass in all: 14.3 -> 14.7 ms with it)
            unsigned long long tot = 0;
            for (int k = 0; k < QCLASSES; ++k) tot += hits[k];
            c->count_tab_now = tune().count_tab >= 2 || (tune().count_tab == 1 && (ch.ordered || tot >= 2 * budget));
            if (tune().debug) fprintf(stderr, "[sohit] chunk %d: %llu seed hits in this batch, budget %llu per pass, count_tab %lld -> %d\n", ci, tot, budget, (long long)tune().count_tab, (int)c->count_tab_now);
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
        while (whole_classes && qb < b.nq && qb == b.cls_start[b.qcls[qb]] && acc + cls_hits[b.qcls[qb]] <= budget && grp[b.qcls[qb]] == grp[b.qcls[qa]])
            acc += cls_hits[b.qcls[qb]], qb = b.cls_start[b.qcls[qb] + 1];   // (what the query-by-query loop below would do, without the 100 k steps)
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
    u32 H, K;
    if (AS == 1) {   // (slot = packed position: both prefixes from mark / scnt in one 64-bit scan, k_util.hip)
        c->d_scan_tmp.ensure(effscan_temp_elems(Tp) + 8);
        const u32* dHK = effscan(b.mark.p + p_lo, b.scnt.p + t_lo, Tp, b.hoff.p + t_lo, b.cidx.p + t_lo, c->d_scan_tmp.p, c->st);
        u32* v = (u32*)small_host(c);
        HIP_CHECK(hipMemcpyAsync(v, dHK, 2 * sizeof(u32), hipMemcpyDeviceToHost, c->st));
        HIP_CHECK(hipStreamSynchronize(c->st));
        H = v[0], K = v[1];
    } else {
        b.eff.ensure((size_t)AS * b.dev.Ppad + 4), b.nz.ensure((size_t)AS * b.dev.Ppad + 4);
        launch_effcnt(b.mark.p, b.scnt.p, AS, p_lo, p_hi, b.eff.p, b.nz.p, c->st);
        const u32* dH = scan_u32(b.eff.p + t_lo, b.hoff.p + t_lo, Tp, false, c->d_scan_tmp.p, c->st);
        stash_u32(c, dH, 0);  // the scan's total lives in d_scan_tmp: park it before the next scan
        const u32* dK = scan_u32(b.nz.p + t_lo, b.cidx.p + t_lo, Tp, false, c->d_scan_tmp.p, c->st);
        d2h_pair(c, dK, H, K);
    }
    sc.lap("seed.bounds_cap_scan");
    c->cnt.seed_hits += H;
    if (H == 0 || K == 0) {
        c->cnt.seed_ms += (wall() - t0) * 1e3;
        return;
    }
    // key layout.  Query-position bits follow the PASS's longest query (passes hold one length class, seed_stage); subject and
    // diagonal bits come from the compact banded addends when they fit (band_encoding), else from the chunk's longest sequence.
    u32 pmaxq = 0;
    for (u32 q = qa; q < qb;) {   // (whole blocks of 256 slots through their maxima)
        if ((q & 255u) == 0 && q + 256u <= qb) pmaxq = std::max(pmaxq, b.blkmax[q >> 8]), q += 256u;
        else pmaxq = std::max(pmaxq, b.h_off[q + 1] - b.h_off[q]), ++q;
    }
    KeyLayout kl;
    kl.bq = ceil_log2((u64)b.nq + 1);
    kl.bp = ceil_log2(std::max<u32>(pmaxq, 2));
    kl.ba = AS > 1 ? ceil_log2((u64)AS) : 0;
    const bool force_wide = tune().lk_wide;
    const bool bands_ok = tune().bands;   // SOHIT_BANDS=0: one band per subject whatever its length
    // A pass dense enough for the bucketed binning whatever the range width: the chunk's buckets are put in order first (once per index
    // build), so that the count pass can read range boundaries instead of the entries (order_chunk / range_table, host_index.hip)
    if (tune().bucket && c->count_tab_now && AS == 1 && !force_wide && !ch.ordered &&
        (u64)H * 1024ull >= (u64)std::max(1ll, tune().bucket_min) * (u64)(qb - qa) * (u64)std::max<u32>(nseq_chunk, 1))
        order_chunk(c, ch);
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
    const u32 gallop_min = 1024;   // k_ungap's GALLOP variant from this query length on
    const int bsp = ceil_log2((u64)ch.maxslen + 1);
    const int ft_bits_entry = (klr.bs + 1) + klr.ba + bsp;
    const bool ft_walk = AS > 1;  // several (alphabet, pattern) combinations: a group's first-touch key needs all its hits
    if (kl.total > 64) throw SoError("sort key needs " + std::to_string(kl.total) + " bits (> 64): lower SOHIT_BATCH or -c");
    if (klr.sh_subj + klr.bs > 64) throw SoError("pass-record key exceeds 64 bits: sequences too long for this build");
    if (kl.ba + kl.bp + ft_bits_entry > 64) throw SoError("first-touch key exceeds 64 bits: sequences too long for this build");
    b.cs_hoff.ensure((size_t)K + 2), b.cs_beg.ensure((size_t)K + 2), b.cs_kbase.ensure((size_t)K + 2);
    launch_compact_seeds(b.mark.p, b.scnt.p, b.hoff.p, b.cidx.p, b.sbeg.p, b.dev.d_pseq.p, b.dev.d_off.p, p_lo, p_hi, AS, kl, b.cs_hoff.p, b.cs_beg.p,
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
            // hits per (tile, range): from the range boundaries of the ordered index buckets (no entry is read), else by the counting pass
            const ChunkIndex::RangeTab* rt = (c->count_tab_now && ch.ordered) ? range_table(c, ch, *enc, wb, L.R) : nullptr;
            const bool tab = rt && rt->ok;
            ProfTimer pt(c, &c->cnt.count_ms, &c->cnt.count_launches);
            if (tab) launch_bkt_count_tab(b.btd.p, qseg, b.bt0.p, nqp, NT, b.cs_hoff.p, b.cs_beg.p, ch.row_of_slot.p, rt->tab.p, L.R, b.bmat.p, c->st);
            else launch_bkt_pass(false, b.btd.p, qseg, NT, b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, dk32, c->ref.d_off.p + ch.seq_lo, L, b.bmat.p, nullptr, c->st);
            pt.stop();
            if (tab && c->profile) c->tm["seed.bucket_count_tab_launches"] += 1;
            if (tab && tune().count_tab == 2) {   // tests: the counting pass must give the same matrix
                b.bpart.ensure(nm + 4);
                launch_bkt_pass(false, b.btd.p, qseg, NT, b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, dk32, c->ref.d_off.p + ch.seq_lo, L, b.bpart.p, nullptr, c->st);
                c->d_small.ensure(32);
                HIP_CHECK(hipMemsetAsync(c->d_small.p + 14, 0, sizeof(u32), c->st));
                launch_u32_differ(b.bmat.p, b.bpart.p, nm, c->d_small.p + 14, c->st);
                const u32 nd = d2h_u32(c, c->d_small.p + 14);
                if (nd) throw SoError("range-table counts differ from the counting pass in " + std::to_string(nd) + " cells");
            }
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
        if (ug1)
            launch_ungap1(c->ncu, pmaxq, b.hits32s.p, b.bext.p, nb, L, kl, klr, btab, U1_WAIT, b.dev.d_scls.p, b.dev.d_off.p,
                          c->ref.d_ug_store.p + U1_UG_PAD, c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.bflag.p + 1, b.shard.p, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p,
                          ug2 ? b.mlist.p : nullptr, b.bflag.p + 3, ugstat, c->st);
        if (ug2)
            launch_ungap2(c->ncu, b.mlist.p, b.bflag.p + 3, b.hits32s.p, b.bext.p, L, kl, klr, btab, U1_WAIT, b.dev.d_ug_store.p + U1_UG_PAD, b.dev.d_off.p,
                          c->ref.d_ug_store.p + U1_UG_PAD, c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.bflag.p + 2, b.shard.p, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p,
                          ugstat, c->st);
        else
            launch_ungap(w32 ? nullptr : b.keys2.p, Hv, kl, klr, btab, pmaxq >= gallop_min, ft_walk, b.dev.d_scls.p, b.dev.d_off.p, c->ref.d_scls4.p,
                         c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.shard.p, shard_cap, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p, c->st, w32 ? b.hits32s.p : nullptr,
                         b.bext.p, nb, &L, ug1, ugstat);
        // the pass records are binned by (query, range of 2^wb chunk SEQUENCES): the same layout unless bands and sequences differ
        bL = L;
        bL.R = (u32)(((u64)nseq_chunk + (1ull << wb) - 1) >> wb);
        bnb = bL.R * nqp;
        c->cnt.hits_bucketed += H;
        return true;
    };

    // ---- the sorted path: 8-byte keys, segmented radix sort by (subject, diagonal), group walk over the sorted keys ----
    auto group_sorted = [&] {
    b.keys.ensure((size_t)H + 2), b.keys2.ensure((size_t)H + 2);
    const u32 nseg = qb - qa;
    // Sparse passes of short queries (one alphabet x one pattern, compact addends): the hits that are ALONE on their diagonal -- about half
    // of them -- are found and extended by k_ungapq, a wave per query, without a key being written for them; the rest are written as
    // k_lookup's keys into an array of all-ones (= dropped) and take the sort + k_ungap below.
    const bool uq = tune().ungapq && tune().segsort && AS == 1 && compact && !ft_walk && UG_SHARDS == 1 && pmaxq <= ungapq_qcap() && nseg >= 256;
    if (uq || nseg >= 256) {
        b.qseg.ensure((size_t)b.nq + 4);
        launch_query_segments(b.hoff.p, b.dev.d_off.p, qa, qb, AS, H, b.qseg.p, c->st);
    }
    if (uq) {
        if (!c->ref.ug_valid) {
            const size_t nres = c->ref.res.size();
            c->ref.d_ug_store.ensure(nres + 2 * (size_t)U1_UG_PAD);
            launch_make_ug(c->ref.d_scls.p, c->ref.d_off.p, (u32)c->ref.N, nres, 8u, c->ref.d_ug_store.p + U1_UG_PAD, c->st);
            c->ref.ug_valid = true;
        }
        b.blk_first.ensure((size_t)nseg + 4), b.flags.ensure((size_t)nseg + 4);
        ProfTimer pt(c, &c->cnt.lookup_ms, &c->cnt.lookup_launches);
        launch_ungapq(c->ncu, b.qseg.p + qa, nseg, qa, b.blk_first.p, b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, K, dk32, kl, klr, btab, b.dev.d_scls.p, b.dev.d_off.p,
                      c->ref.d_ug_store.p + U1_UG_PAD, c->ref.d_off.p + ch.seq_lo, c->d_b62c.p, b.bflag.p + 1, b.shard.p, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p,
                      b.keys.p, b.keys2.p, b.flags.p, ugstat, c->st);
        pt.stop();
        if (c->profile) c->cnt.lookup_bytes += (i64)8 * (i64)H;
    } else {
    b.blk_first.ensure((size_t)lookup_num_blocks(H) + 2);
    launch_lookup_blockfirst(b.cs_hoff.p, K, H, b.blk_first.p, c->st);
    {
        ProfTimer pt(c, &c->cnt.lookup_ms, &c->cnt.lookup_launches);
        launch_lookup(b.cs_hoff.p, b.cs_beg.p, b.cs_kbase.p, b.blk_first.p, K, H, compact ? (const void*)dk32 : (const void*)ch.dkeys.p,
                      compact, c->ref.d_off.p + ch.seq_lo, kl, ch.maxslen, b.keys.p, c->st);
        pt.stop();
        if (c->profile) c->cnt.lookup_bytes += (i64)8 * (i64)H;
    }
    }
    t1 = wall();
    sc.lap("seed.compact_lookup");
    // Hits are generated in (query, qpos, as) order (position-major seed ordinals) and the radix sorts are
    // stable, so only the (subject, diagonal) bits need sorting, inside each query's segment: 2 radix passes
    // fewer than a device-wide sort of the (query, subject, diagonal) bits.  One block sorts one segment, so
    // passes with few queries (huge per-query hit lists) use the device-wide sort instead.
    // (Dropped hits carry ~0 and sort last in their segment.)
    const int seg_mode = (int)tune().segsort;
    if (seg_mode && nseg >= 256) {
        ensure_sort_tmp(c, sort_keys_u64_seg_temp_bytes(H, nseg, kl.sh_diag, kl.sh_q));
        if (uq) sort_keys_u64_seg2(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.keys.p, b.keys2.p, H, nseg, b.qseg.p + qa, b.flags.p, kl.sh_diag, kl.sh_q, c->st);
        else sort_keys_u64_seg(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.keys.p, b.keys2.p, H, nseg, b.qseg.p + qa, kl.sh_diag, kl.sh_q, c->st);
    } else {
    ensure_sort_tmp(c, sort_keys_u64_temp_bytes(H, kl.total));
    sort_keys_u64(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.keys.p, b.keys2.p, H, kl.sh_diag, kl.total, c->st);
    }
    sc.lap("group.sort_keys");
    // group walk + chained ungapped extension (the kernel finds the group heads itself)
    launch_ungap(b.keys2.p, H, kl, klr, btab, pmaxq >= gallop_min, ft_walk, b.dev.d_scls.p, b.dev.d_off.p, c->ref.d_scls4.p, c->ref.d_off.p + ch.seq_lo,
                 c->d_b62c.p, b.shard.p, shard_cap, b.p_qs.p, b.p_sd.p, b.p_ft.p, b.stepshard.p, c->st, nullptr, nullptr, 0, nullptr, false, ugstat);
    };

    reset_pass_lists();
    if (!group_bucketed()) {
        bbest = false;
        group_sorted();
    }
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
                // array, no key-build pass, 8 instead of 12 bytes per candidate and radix pass
                const int ftw = kl.ba + kl.bp + ft_bits_entry - bsp + 1;
                if (bL.nqp >= 256 && ftw < 63 && 63 - ftw >= klr.bs + 1) cand_idx_bits = 63 - ftw, cand_ftw = ftw;
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
        // sorting path below.
        // (not for a pass that holds queries of 4096 residues and more: a giant brings more records than the largest instance sorts, and
        // one such query sends the whole pass down the old path after the new one has run)
        if (pmaxq < 4096 && (kl.ba + kl.bp + ft_bits_entry) - bsp + 1 <= cand_order_lds_key_bits() && qb > qa) {
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
        c->d_small.ensure(32);
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
    if (!bbest && maxseg <= (u32)cand_order_lds_max() && ftbits - bsp + 1 <= cand_order_lds_key_bits()) {
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
        // sort + row gather per query in one hand-written kernel (k_cand_order_seg)
        HIP_CHECK(hipMemsetAsync(b.bflag.p, 0, sizeof(u32), c->st));
        launch_cand_order_seg(c_ftp, b.qseg.p, bL.nqp, bL.qa, cand_idx_bits, cand_idx_bits + cand_ftw, c_recp, b.cand_q.p + base, b.cand_rec.p + 4 * (size_t)base,
                              qcnt, b.bflag.p, c->st);
        if (d2h_u32(c, b.bflag.p) != 0) {   // a digit group above the LDS sort (never seen): the library's segmented radix sort redoes the pass's order
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
        ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(NS, 64));
        sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, b.order.p, b.order2.p, NS, ftbits - bsp + 1 + kl.bq, c->st);
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
