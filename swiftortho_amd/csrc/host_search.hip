// host_search.hip -- libsohit.so host side: the batched search, the result-array cache, the per-query work pre-pass (see host.h).
#include "host.h"

HitCache g_hit_cache;


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
