// host_phase2.hip -- libsohit.so host side: phase 2 -- candidates in the reference's order, banded alignments in rounds, stop rule, traces, row emission (see host.h).
#include "host.h"


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
        const bool cs_aside = gx != nullptr;
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
    // score-only rounds by k_align_lane (a lane per alignment pair) when every task's windows end where its sequences end: no tiles
    const bool lane_on = pk_on && tune().align_lane && b.maxqlen < LONG_SEQ && c->ref.maxlen < LONG_SEQ;
    const bool traced_pk = pk_on;   // traced alignments by the packed kernel too (k_align<true> keeps the tasks whose scores need 32-bit cells)
    auto sort_by_rows = [&](const u32* list, u32 n, u32* n_wide) -> const u32* {
        const bool split = n_wide && pk_mixed;
        if (n_wide) *n_wide = pk_on ? 0u : n;
        if (!split && (!align_sort || n < 4096)) return list;
        b.tmp64.ensure((size_t)n + 2), b.c_ft2.ensure((size_t)n + 2), b.ridx2.ensure((size_t)n + 2);
        ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(n, 64));
        // (ordering inside blocks of 2^k queries instead of globally -- key = query block << 13 | rows -- was measured: 24.9-25.2 ms of
        // align rounds for k = 7 ... 13 against 24.8-25.2)
        c->d_small.ensure(32);
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
    // config 2, 0.55 M tasks, 16.7 -> 17.1 ms).  SOHIT_SPEC_SLACK: the guess tests the ungapped score against expect x this (default 1e6 since the end of
    // round 6 -- with the cheaper walk and the one-launch alignment of the rows left over, config 3 interleaved: 1e2 47.70 ms, 1e3 47.56, 1e4 47.51, 1e5 47.38,
    // 1e6 47.21, 1e7 47.17, 1e8 47.28, 1e10 53.2 (the first round's traces no longer fit); the weight-6 and mixed-length sets do not care.  Round 5, 1e3:
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
            c->d_small.ensure(32);
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
            const bool wide_aside = n_wide && NR > n_wide;
            hipStream_t wst = wide_aside ? c->st_side : c->st;
            if (wide_aside) {
                HIP_CHECK(hipEventRecord(c->ev_ug_go, c->st));
                HIP_CHECK(hipStreamWaitEvent(c->st_side, c->ev_ug_go, 0));
            }
            if (n_wide)
                launch_align(b.tasks.p, rlist, n_wide, b.dev.d_res.p, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_scls.p,
                             c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p, nullptr, stride, nullptr, b.ares.p, false, wst, 0u);
            if (wide_aside) HIP_CHECK(hipEventRecord(c->ev_ug_done, c->st_side));
            if (NR > n_wide) {
                // (a lane walks a whole alignment alone: a launch lasts at least one alignment's ~0.4 ms however few tasks it holds -- the last
                // rounds of config 3, 6.6 k and 64 tasks, took 0.63 and 0.40 ms; sixteen lanes per pair finish those in 0.1)
                if (lane_on && NR - n_wide >= (1u << 18)) {
                    c->d_small.ensure(32);
                    launch_align_lane(b.tasks.p, rlist + n_wide, NR - n_wide, pkc, b.dev.d_off.p, c->ref.d_off.p, c->d_b62c.p, b.ares.p, c->d_small.p + 13, c->ncu, c->st);
                } else {
                    launch_align_pk(b.tasks.p, rlist + n_wide, NR - n_wide, pkc, b.dev.d_off.p, c->ref.d_off.p, c->d_b62c.p, b.ares.p, c->st);
                }
            }
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
    // (with kept traces -- nspec -- on a batch of mixed lengths the last stage orders and launches per range: two ranges, 66.0 against 66.9 ms with
    // four on the log-normal set; a uniform batch takes four since its left-over rows are aligned in one launch: config 3 1 range 48.26 ms, 2 47.19, 3 46.98,
    // 4 46.91, 5 47.06, 6 47.16, 8 47.47)
    const int EMIT_PARTS = std::min<int>(EMIT_PARTS_MAX, std::max(1, (nspec && (b.permuted || pk_mixed)) ? 2 : (int)tune().emit_parts));
    const u32 emit_min_rows = (u32)std::max(1ll, tune().emit_min_rows);
    // (config 3, one batch: 1 part 57.0 ms per step, 4 parts 56.0)
    const u32 qstep = (nq + EMIT_PARTS - 1) / EMIT_PARTS;
    c->d_small.ensure(32);
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
        const bool order_rows = b.permuted || (u64)b.maxqlen * b.nq > 3ull * b.h_off[b.nq] / 2;
        // tasks of emission range p's traced list that need the 32-bit cells (they lead the ordered range); a mixed batch whose lists are
        // not ordered keeps the 32-bit kernel for all of them
        u32 nwide_part[EMIT_PARTS_MAX] = {0};
        if (order_rows) {
            // [t0, t1) of `in`, longest band first, to the same range of `out`; wide (non-null): the wide tasks first, their number counted there
            // (the ranges' counts come back in ONE copy behind the loop: a synchronisation per range stalled the range-by-range overlap)
            c->d_small.ensure(32);
            const bool split = traced_pk && pk_mixed;
            if (split) HIP_CHECK(hipMemsetAsync(c->d_small.p + 16, 0, EMIT_PARTS_MAX * sizeof(u32), c->st));
            auto order_list = [&](const u32* in, u32 t0, u32 t1, u32* out, u32* wide) {
                const u32 n = t1 - t0;
                if (!n) return;
                b.tmp64.ensure((size_t)n + 2), b.c_ft2.ensure((size_t)n + 2);
                ensure_sort_tmp(c, sort_pairs_u64_u32_temp_bytes(n, 64));
                launch_task_rows(b.tasks.p, in + t0, n, b.dev.d_off.p, c->ref.d_off.p, b.dev.d_bound.p, c->ref.d_bound.p, wide ? align_pk_max_len() : 0,
                                 wide ? align_pk_max_score() : 0u, wide, nullptr, b.tmp64.p, c->st);
                sort_pairs_u64_u32(c->d_sort_tmp.p, c->d_sort_tmp.cap, b.tmp64.p, b.c_ft2.p, in + t0, out + t0, n, wide ? 14 : 13, c->st);
            };
            b.tl_sorted.ensure((size_t)tn + 4);
            if (nspec) b.al_sorted.ensure((size_t)(NO - tn) + 4);
            for (int p = 0; p < parts; ++p) {
                const u32 r0 = parts > 1 ? part_row[p] : 0u, r1 = parts > 1 ? part_row[p + 1] : NO;
                u32* wide = split ? c->d_small.p + 16 + p : nullptr;
                if (nspec) {
                    order_list(tlist, pb[p], pb[p + 1], b.tl_sorted.p, wide);
                    order_list(alist, r0 - pb[p], r1 - pb[p + 1], b.al_sorted.p, nullptr);
                } else {
                    order_list(tlist, r0, r1, b.tl_sorted.p, wide);
                }
            }
            if (split) {
                u32* v = (u32*)small_host(c);
                HIP_CHECK(hipMemcpyAsync(v, c->d_small.p + 16, EMIT_PARTS_MAX * sizeof(u32), hipMemcpyDeviceToHost, c->st));
                HIP_CHECK(hipStreamSynchronize(c->st));
                for (int p = 0; p < parts; ++p) nwide_part[p] = v[p];
            }
            tlist = b.tl_sorted.p;
            if (nspec) alist = b.al_sorted.p;
        }
        const size_t tw = tn ? trace_offsets(tlist, tn) : 0;
        const bool tvar = tw <= var_budget_words;
        b.trace.ensure(tvar ? tw + 64 : (size_t)std::min(slab, std::max<u32>(maxpart, 1)) * stride + 64);
        // The rows without a kept trace are aligned in ONE launch in front of the ranges when none of them needs the 32-bit kernel (uniform
        // sets): a range's share (config 3: 87 k tasks = 1.3 fillings of the chip) left half a filling idle -- 0.35 + 1.0 ms in two launches,
        // 0.7 in one; the ranges then only walk their kept traces (config 3, interleaved runs: 47.49 against 47.61 ms -- the second range's
        // kept-trace walk now runs beside the first range's row download and pays for it).
        // (only behind kept traces: without them this list is every reported row, and its ranges' downloads overlap the later ranges' alignments)
        const bool hoist = nspec && tvar && traced_pk && !pk_mixed && parts > 1 && tn > 0;
        if (hoist) {
            ProfTimer pt(c, &c->cnt.align_ms, &c->cnt.align_launches);
            launch_align_traced(b.tasks.p, tlist, tn, b.dev.d_scls.p, b.dev.d_scls4.p, b.dev.d_off.p, c->ref.d_scls.p, c->ref.d_scls4.p, c->ref.d_off.p, c->d_b62c.p,
                                b.trace.p, TU, b.tr_ofs.p, b.ares.p, nullptr, 0u, c->st, 0u, pkc);
            // ... and walked at once, while their traces are in the L2 (walked range by range, the second range's came back from the Infinity
            // Cache behind the first range's 0.9 GB of kept traces: 0.94 instead of 0.27 ms)
            launch_traceback_tofs(b.tasks.p, tlist, tn, b.dev.d_res.p, b.dev.d_off.p, c->ref.d_res.p, c->ref.d_off.p, b.trace.p, TU, b.tr_ofs.p, b.ares.p, c->st);
            pt.stop();
        }
        auto align_traced = [&](u32 t0, u32 t1, int p) {   // tasks [t0, t1) of tlist = emission range p's
            if (t1 <= t0) return;
            if (hoist) return;
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
