// kernels.h -- launch wrappers implemented in the k_*.hip translation units.
#pragma once
#include "common.h"

struct HashLut;

// k_util.hip
size_t scan_u32_temp_elems(size_t n);
size_t effscan_temp_elems(size_t n);
const u32* effscan(const u8* mark, const u32* scnt, size_t n, u32* hoff, u32* cidx, u32* temp, hipStream_t st);
const u32* scan_u32(const u32* in, u32* out, size_t n, bool inclusive, u32* temp, hipStream_t st);  // returns device ptr to total
void fill_u32(u32* p, size_t n, u32 v, hipStream_t st);

// k_sort.hip
size_t sort_keys_u64_temp_bytes(size_t n, int bits);
void sort_keys_u64(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, int begin_bit, int end_bit, hipStream_t st);
// seg[q] = first hit ordinal of batch query q in the pass, q in [qa, qb] (seg[qb] = H)
void launch_query_segments(const u32* hoff, const u32* qoff, u32 qa, u32 qb, int AS, u32 H, u32* seg, hipStream_t st);
size_t sort_keys_u64_seg_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit);
void sort_keys_u64_seg(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit,
                       int end_bit, hipStream_t st);
// ... with the segments' ends given apart from their starts (seg_end[k] <= seg_begin[k + 1]: what lies between is not touched)
void sort_keys_u64_seg2(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg_begin, const u32* seg_end, int begin_bit,
                        int end_bit, hipStream_t st);
void launch_stride_gather(const u32* src, u32 stride, u32 n, u32* dst, hipStream_t st);   // dst[i] = src[i * stride]
size_t sort_pairs_u64_u32_temp_bytes(size_t n, int bits);
void sort_pairs_u64_u32(void* temp, size_t temp_bytes, const u64* kin, u64* kout, const u32* vin, u32* vout, size_t n, int bits,
                        hipStream_t st);

// k_ixsort.hip: (bucket id, entry) pairs grouped by ascending bucket id, members of a bucket in no particular order.  plan:
// ixsort_plan_elems(NC) u32, scan_tmp: scan_u32_temp_elems of that, (tk, tv): E pairs of scratch; (kin, vin) are scratch too when NC > 2^27
size_t ixsort_plan_elems(u32 NC);
void ixsort_pairs(u32* kin, u64* vin, u32 E, u32 NC, u32* plan, u32* scan_tmp, u32* tk, u64* tv, u32* kout, u64* vout, hipStream_t st);
// the same in two steps over n SLOTS of which some hold no pair (key ~0): the count and its scan leave the number of pairs in the returned device word
const u32* ixsort_count(const u32* kin, u32 n, u32 NC, u32* plan, u32* scan_tmp, hipStream_t st);
void ixsort_finish(u32* kin, u64* vin, u32 n, u32 NC, const u32* plan, u32* tk, u64* tv, u32* kout, u64* vout, hipStream_t st);
void launch_index_windows_sparse(const u32* words, const u32* pseq, const u32* off, u32 p_lo, u32 p_hi, u32 Ppad, u32 seq_lo, const SeedCfg& cfg, const HashLut& lut, u32 step,
                                 u32* bkt, u64* ent, hipStream_t st);

// k_prep.hip
void launch_layout(const u8* res, const u32* off, u32 nseq, u32 P, u32 Ppad, const u8* hmap, u32* pseq, u8* pcls, u32* words,
                   hipStream_t st);
// score classes of n residues; scls / scls4 point SCLS_PAD_FRONT bytes into allocations of n + SCLS_PAD_FRONT + SCLS_PAD_BACK
// bytes, and the pads are zeroed here
#define PCLS_PAD 48   // sentinel bytes behind every sequence of the packed aligner's class arrays (k_pad_cls)
struct PkCls {   // the four padded class arrays of a (query batch, reference) pair
    const u8 *q, *q4, *r, *r4;
};
void launch_pad_cls(const u8* scls, const u32* off, u32 nseq, u8* out /*PCLS_PAD bytes into an allocation of nres + PCLS_PAD * (nseq + 2) + 64*/, u8* out4,
                    hipStream_t st);
#define SCLS_PAD_FRONT 16
#define SCLS_PAD_BACK 64
void launch_scls(const u8* res, size_t n, const u8* smap, u8* scls, u8* scls4 /*nullable: class * 4*/, hipStream_t st);
// bound[s] = sum over sequence s of max(0, row maximum of its residues' score classes): no alignment the sequence takes part in scores more
void launch_seq_bound(const u8* scls, const u32* off, u32 nseq, const signed char* b62c_host, u32* bound, hipStream_t st);
// batch slot q holds sequence q_lo + (qid ? qid[q] : q) of the source set
void launch_seg(const u8* raw, const u32* src_off, u32 q_lo, const u32* qid, u32 nq, const u32* dst_off, const u8* symmap, const u8* upmap,
                const void* tab, u8* mk, u8* out, u32 max_len, u32 q_mid, u32 q_long, hipStream_t st, hipStream_t st_long);
void launch_gather_seqs(const u8* raw, const u32* src_off, u32 q_lo, const u32* qid, u32 nq, const u32* dst_off, u8* out, hipStream_t st);
void launch_copy_range(const u8* src, u8* dst, size_t n, hipStream_t st);

// k_index.hip
#define HTAB_EMPTY 0xFFFFFFFFu
void launch_run_heads(const u32* bkt, u32 E, u32* flags, hipStream_t st);
void launch_run_list(const u32* bkt, const u32* flags, const u32* ridx, u32 E, u32 U, u32* ub, u32* ubeg, u32* cnt, hipStream_t st);
void launch_dir_build(const u32* ub, u32 U, u64* dir /*zeroed, NC / 32 + 1 words*/, hipStream_t st);
void launch_htab_insert(const u32* ub, const u32* ubeg, u32 U, u32* hkey, u64* hval, int hshift, u32 hmask, hipStream_t st);
#define INDEX_STATS_BLOCKS 2048
void launch_index_stats(const u32* counts, u32 NC, u64* stats_buf /*4 + 4 * INDEX_STATS_BLOCKS*/, hipStream_t st);
void launch_encode_band32(const u64* entries, u32 E, int ba, const u32* gbase /*per chunk sequence*/, const u32* roff /*chunk-local*/, u32* dk32, hipStream_t st);
void launch_encode_delta(const u64* entries, u32 E, int sh_subj, int sh_diag, u32 maxslen, u64* dkeys, hipStream_t st);
void launch_index_fixlast(u64* entries, u32 lo, u32 E, hipStream_t st);

// k_seed.hip
void launch_qhash(const u32* words, u32 Ppad, const SeedCfg& cfg, const HashLut& lut, u32* qbucket, hipStream_t st);
void launch_bounds(const u32* qbucket, u32 Ppad, int AS, const u32* hkey, const u64* hval, int hshift, u32 hmask, const u64* dir /*or null: the map*/,
                   const u32* ubeg, u32 NC, u32 E, u32* sbeg, u32* scnt, u32* pcnt, hipStream_t st);
void launch_ksc_order(const u8* q_scls, const u32* qoff, u32 nq, u32 q_long /*first batch slot that may hold more than ksc_lds_max() windows*/, int mink,
                      const signed char* b62c, u64* gx, u32* gL, u32* gR /*global scratch per residue: only when q_long < nq*/, u32* korder, hipStream_t st,
                      hipStream_t st_long /*stream of the global-scratch instance: st, or a side stream the caller orders against st*/);
int ksc_lds_max();
void launch_ksc_order_list(const u8* q_scls, const u32* qoff, u32 nq, const u32* list, u32 nlist, u8* have, u32 q_long, int mink, const signed char* b62c,
                           u64* gx, u32* gL, u32* gR, u32* korder, hipStream_t st, hipStream_t st_long);
void launch_cap_all(const u32* qoff, u32 nq, int mink, const u32* pcnt, i64 threshold, u8* mark, unsigned long long* qhits, unsigned long long* over,
                    u32* open_list, hipStream_t st);
void launch_cap(const u32* korder, const u32* qoff, u32 q0, u32 nq /*batch slots [q0, nq)*/, int mink, const u32* pcnt, i64 threshold, u8* mark,
                unsigned long long* qhits, const u32* list, u32 nlist, hipStream_t st);
// (these two work on the pass's seed slots only, [AS * p_lo, AS * p_hi))
void launch_effcnt(const u8* mark, const u32* scnt, int AS, u32 p_lo, u32 p_hi, u32* eff, u32* nz, hipStream_t st);
void launch_compact_seeds(const u8* mark, const u32* scnt, const u32* hoff, const u32* cidx, const u32* sbeg, const u32* q_pseq, const u32* qoff,
                          u32 p_lo, u32 p_hi, int AS, const KeyLayout& kl, u32* cs_hoff, u32* cs_beg, u64* cs_kbase, hipStream_t st);
u32 lookup_num_blocks(u32 H);
void launch_lookup_blockfirst(const u32* cs_hoff, u32 K, u32 H, u32* wave_first, hipStream_t st);
void launch_lookup(const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase, const u32* wave_first, u32 K, u32 H,
                   const void* dkeys /*u32 (compact) or u64 addends*/, bool compact, const u32* roff, const KeyLayout& kl, u32 maxslen,
                   u64* keys, hipStream_t st);

// k_group.hip
void launch_group_list(const u32* flags, const u32* gidx, u32 H, u32* ghead, hipStream_t st);
u32 ungap_shard_cap(u32 H);
// klr: layout of the pass records (sequence bits / diagonal offset; == kl unless btab); btab: band -> (chunk sequence, gbase) when
// some subject owns several diagonal bands (k_encode_band32), else null
// gallop: the pass holds queries long enough for runs of covered seeds worth skipping in one step (k_ungap's GALLOP)
void launch_ungap(const u64* keys, u32 H, const KeyLayout& kl, const KeyLayout& klr, const void* btab, bool gallop, bool ft_walk, const u8* q_scls, const u32* qoff,
                  const u8* r_scls, const u32* roff, const signed char* b62g, u32* shard_cnt, u32 shard_cap, u64* p_qs, u64* p_sd,
                  u64* p_ft, unsigned long long* group_count, hipStream_t st,
                  // bucketed passes: instead of `keys`, the buckets' sorted 32-bit words (launch_bkt_group with words32), their extents and layout
                  const u32* words = nullptr, const u32* bext = nullptr, u32 nb = 0, const BktLayout* L = nullptr,
                  bool skip_single = false /*bucketed passes: the singleton groups were k_ungap1's*/,
                  unsigned long long* stat = nullptr /*non-null: stat[0] += b62 lookups (the reference's `flag`, fsearch.py:2467, 2482)*/);

// k_ungap1.hip: the singleton groups of a bucketed pass (queries up to U1_QCAP residues); pass records are appended like k_ungap's
#define UG_REC_NONE 0xFFFFFFFFFFFFFFFFull   // p_qs of an unused pass-list slot (k_ungap1 reserves the list in pieces)
#define U1_QCAP 2048          // longest query (its classes sit in an LDS slot per wave; packed 16-bit scores: 11 * 2048 + the pin's 8192 < 2^15)
#define U1_UG_PAD 4096        // sentinel bytes in front of and behind the subject-side array (a dropped pass keeps reading while the other one runs)
u32 ungap1_qcap();
size_t ungap1_list_slack(u32 ncu);   // pass-list slots its waves may leave unused
// class * mul of every residue, position 0 of every sequence and both pads = the sentinel class (mul = 8: the subject side r_ug; mul = 1:
// the chain kernel's query side q_ug); `out` points U1_UG_PAD bytes into an allocation of nres + 2 * U1_UG_PAD bytes
void launch_make_ug(const u8* scls, const u32* off, u32 nseq, size_t nres, u32 mul, u8* out, hipStream_t st);
size_t ungap1_mlist_cap(u32 H, u32 ncu);   // entries of the chain list of a pass of H hits
// mlist / mlist_cnt (nullable; counter zeroed): the heads of the groups of two and more hits, for launch_ungap2; without them those
// groups are launch_ungap's (skip_single)
#define U1_WAIT 4u   // k_ungap1 / k_ungap2: idle lanes that trigger a hand-out
void launch_ungap1(u32 ncu, u32 pmaxq, const u32* words, const u32* bext, u32 nb, const BktLayout& L, const KeyLayout& kl, const KeyLayout& klr,
                   const void* btab, u32 wait_n, const u8* q_scls, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g, u32* work_ctr /*zeroed*/,
                   u32* shard_cnt, u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, u64* mlist, u32* mlist_cnt,
                   unsigned long long* stat /*nullable: [0] += b62 lookups, [1] += singleton groups*/, hipStream_t st);
// sparse passes, a wave per query (k_ungapq.hip): hits alone on their diagonal extended at once, the others written as k_lookup's keys
u32 ungapq_qcap();
void launch_ungapq(u32 ncu, const u32* qseg, u32 nqp, u32 qa, u32* qk, const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase, u32 K, const u32* dk32,
                   const KeyLayout& kl, const KeyLayout& klr, const void* btab, const u8* q_scls, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g,
                   u32* work_ctr, u32* shard_cnt, u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, u64* keys, u64* keys_sorted, u32* seg_end,
                   unsigned long long* stat, hipStream_t st);
void launch_ungap2(u32 ncu, const u64* mlist, const u32* mlist_cnt, const u32* words, const u32* bext, const BktLayout& L, const KeyLayout& kl, const KeyLayout& klr,
                   const void* btab, u32 wait_n, const u8* q_ug, const u32* qoff, const u8* r_ug, const u32* roff, const signed char* b62g, u32* work_ctr /*zeroed*/,
                   u32* shard_cnt, u64* p_qs, u64* p_sd, u64* p_ft, unsigned long long* group_count, unsigned long long* stat /*nullable: [0], [2] += chained groups*/,
                   hipStream_t st);
void launch_first_touch(bool walk, const u64* keys, u32 H, const KeyLayout& kl, int ft_bits_entry, int bsp, const u32* roff, u64* p_ft, u32 n,
                        hipStream_t st);
void launch_shard_scan(const u32* shard_cnt, u32* shard_off, hipStream_t st);
void launch_compact_shards(const u32* shard_cnt, u32* shard_off, u32 shard_cap, const u64* a0, const u64* a1, const u64* a2, u64* b0,
                           u64* b1, u64* b2, hipStream_t st);
void launch_seg_flags(const u64* sorted_qs, u32 n, u32* flags, u32* zero_word /*set to 0 (may be null)*/, hipStream_t st);
int cand_order_lds_max();
int cand_order_lds_key_bits();
// (both over the pass's queries [q0, q1) of the batch)
void launch_qseg(const u64* sorted_qs, u32 n, const u32* gidx, const u32* total, int bs, u32 q0, u32 q1, u32* seg, u32* maxseg, hipStream_t st);
// k_group.hip: best diagonal + candidate order of a sparse pass per query in LDS (no sort of the pass records)
int q_best_max();
void launch_qrec_scatter(const u64* p_qs, const u64* p_sd, const u64* p_ft, const u32* rnk, u32 n, int bs, u32 qa, const u32* qoff, u32* o_q, u32* o_subj,
                         u64* o_sd, u64* o_ft, hipStream_t st);
void launch_q_best(const u32* qoff, u32 qa, u32 nqp, const u32* o_subj, const u64* o_sd, const u64* o_ft, u32 seq_lo, int bsp, u32* t_rec, u32* perm,
                   u32* qcnt, u32* fallback, hipStream_t st);
void launch_q_emit(const u32* qoff, u32 qa, u32 nqp, u32 nslots, const u32* o_q, const u32* qcnt, const u32* coff, const u32* perm, const u32* t_rec,
                   u32* out_q, u32* out_rec, hipStream_t st);
void launch_cand_order_lds(const u64* c_ft, const u32* c_rec, const u32* seg, u32 q0, u32 q1, u32 maxseg, int bsp, u32* out_q, u32* out_rec, u32* qcnt,
                           hipStream_t st);
void launch_best(const u64* sorted_qs, const u32* idx, const u32* shead, u32 nseg, u32 n, const u64* p_sd, const u64* p_ft,
                 u32 seq_lo, int bs, u64* c_ft, u32* c_q, u32* c_rec, hipStream_t st);
void launch_gather_u32_as_u64(const u32* src, const u32* idx, u32 n, u64* dst, hipStream_t st);
void launch_iota(u32* p, u32 n, hipStream_t st);
void launch_combine_q_ft(const u32* c_q, const u64* c_ft, u32 n, int ftbits, int bsp, u64* dst, hipStream_t st);
void launch_emit_cands(const u32* order, u32 n, const u64* sorted_key /*the final sort's key stream*/, int qshift /*query = key >> qshift*/,
                       const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt, u32* seg_first, hipStream_t st);

// k_align.hip
// Trace words of one alignment (k_align<true>, k_align_pk<true> -> k_traceback): word (b, l) holds lane l's two band cells of the eight
// iterations 8 (b + 1) .. + 7.  Laid out in groups of FOUR blocks, lane-major inside a group: a walk stays in one lane for many columns and
// steps back through the blocks, so its next four words are one aligned 16-byte piece (a 64-byte line = 4 lanes x 4 blocks) -- with the
// blocks 64 bytes apart (rounds 1-5) every word a walk read was a line of its own, and k_traceback was bound by those line fetches (1.44 M
// walks of config 3: ~35 lines of trace each).  The aligner's 16 lanes store 16 bytes apart instead of side by side; a line is
// complete after four blocks.
#define TRACE_WORD(b, l) ((((u32)(b) >> 2) << 6) | ((u32)(l) << 2) | ((u32)(b) & 3u))
u32 align_trace_stride(int max_rows);
void launch_align(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u8* q_scls, const u8* q_scls4, const u32* qoff,
                  const u8* r_res, const u8* r_scls, const u8* r_scls4, const u32* roff, const signed char* b62g, u32* trace, u32 trace_stride,
                  const u32* tofs /*traces: start of launch position t's trace in units of trace_stride words, or null = t*/, AlnRes* out,
                  bool with_traceback, hipStream_t st, u32 n_wide /*with_traceback: leading positions for the 32-bit kernel, the rest packed*/,
                  PkCls pk = PkCls{nullptr, nullptr, nullptr, nullptr});
// trace room each task of a launch list needs, in units of align_trace_unit() words (+ a 0 behind the last): scanned, they are `tofs`
u32 align_trace_unit();
void launch_trace_units(const AlnTask* tasks, const u32* ridx, u32 n, const u32* qoff, const u32* roff, u32* units /*n + 1*/, hipStream_t st);

// k_align16.hip: score-only aligner in packed 16-bit arithmetic, two alignments per register
bool align_pk_supported(hipStream_t st);   // the d16 load behaviour k_align_pk relies on (probed once per process)
int align_pk_max_len();   // largest min(rows, columns) it can score whatever the residues
u32 align_pk_max_score(); // largest alignment score its cells hold
// one lane per alignment pair, score-only, persistent waves (k_alignl.hip): tasks whose windows end where their sequences end
void launch_align_lane(const AlnTask* tasks, const u32* ridx, u32 ntasks, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g, AlnRes* out,
                       u32* work_ctr, u32 ncu, hipStream_t st);
void launch_align_pk(const AlnTask* tasks, const u32* ridx, u32 ntasks, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g, AlnRes* out,
                     hipStream_t st);

// k_phase2.hip
void launch_gather_cands(const u32* src_q, const u32* src_rec, u32 n, const u32* cqoff, const u32* prior, const u32* qcoff,
                         u32* dst_rec, hipStream_t st);
void launch_add_u32(u32* acc, const u32* x, u32 n, hipStream_t st);
void launch_csort(const u32* rec, const u32* qcoff, u32 nq, u32 vmax, const u32* qoff, const u32* roff, u32* perm, u32* ntask,
                  u32* ntile, u64* gx /*scratch, only for queries with > csort_lds_max() candidates; may be null*/, u32* gL, u32* gR,
                  hipStream_t st, hipStream_t st_g);
int csort_lds_max();
void launch_mktasks(const u32* rec, const u32* qcoff, const u32* perm, const u32* ntask, const u32* roffc, const u32* toff, u32 nq,
                    const u32* qoff, const u32* roff, AlnTask* tasks, u32* rk_slot, hipStream_t st);
void launch_round_counts(const u32* ntask, const u32* ntile, const u32* roffc, const u32* rk_slot, const u32* qcoff,
                         const u32* st_state, u32 nq, double max_miss, u32 minr, u32* rcnt, u32* tcnt, hipStream_t st);
void launch_round_idx(const u32* tcnt, const u32* troff, const u32* toff, const u32* ntask, const u32* ntile, const u32* roffc,
                      const u32* rk_slot, const u32* st_state, u32 nq, u32* ridx, hipStream_t st);
void launch_final_select(const u32* toff, u32 nq, i64 v, u32* sel, const u32* st_state, const int* bits, u32* nout, hipStream_t st);
void launch_selected_idx(const u32* toff, const u32* sel, const u32* nout, const u32* ooff, u32 nq, u32* idx, hipStream_t st);
void launch_emit_hits(const AlnTask* tasks, const AlnRes* res, const u32* toff, const u32* sel, const u32* nout, const u32* ooff,
                      const int* bits, u32 q0, u32 q1 /*queries [q0, q1) of the batch*/, int* out, hipStream_t st);
// so_hit records on the device (80 bytes each) from k_emit_hits rows; qoff_abs = offsets of the whole loaded query set
// qid / ooff / ostart (all null, or all set): batch slots in length-class order, records written in file order (k_make_hits)
void launch_make_hits(const int* rows, u32 n, i64 q_lo, const u32* qid, const u32* ooff, const u32* ostart, const u32* qoff_abs, const u32* roff, i64 D,
                      const double* p2tab, int p2n, void* out, hipStream_t st);
void launch_scatter_u32(const u32* src, const u32* idx, u32 n, u32* dst, hipStream_t st);   // dst[idx[i]] = src[i]

// k_bucket.hip: diagonal binning without a sort (query-aligned tiles, count -> scan -> scatter, LDS hash grouping)
u32 bkt_tile_hits();
void launch_bkt_ntiles(const u32* qseg, u32 nqp, u32* ntile, hipStream_t st);
// the count pass from per-bucket range boundaries (k_bucket.hip; order_chunk / range_table in host_index.hip)
void launch_rtab_build(const u32* dk32, u32 E, const u32* ubeg, u32 U, int gb, u32 R, u16* rtab, u32* row_of_slot, u32* flag, hipStream_t st);
void launch_bkt_count_tab(const void* td, const u32* qseg, const u32* t0 /*first tile of every pass query, + NT*/, u32 nqp, u32 NT, const u32* cs_hoff, const u32* cs_base,
                          const u32* row_of_slot, const u16* rtab, u32 R, u32* mat, hipStream_t st);
void launch_u32_differ(const u32* a, const u32* b, size_t n, u32* flag, hipStream_t st);
size_t sort_keys_u64_seg_desc_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit);
void sort_keys_u64_seg_desc(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg /*nseg + 1*/, int begin_bit, int end_bit,
                            hipStream_t st);
void launch_bkt_tiledesc(const u32* qseg, const u32* t0, u32 nqp, u32 NT, const u32* cs_hoff, u32 K, void* td /*uint4 x NT*/, hipStream_t st);
void launch_bkt_pass(bool scatter, const void* td, const u32* qseg, u32 NT, const u32* cs_hoff, const u32* cs_base, const u64* cs_kbase,
                     const u32* dk32, const u32* roff, const BktLayout& L, u32* mat, u32* out, hipStream_t st);
int bkt_max_wb();
void launch_bkt_extents(const u32* mat, const u32* t0, u32 NT, u32 R, u32 nqp, u32 nb, const u32* total, u32* bext /*nb + 1*/, hipStream_t st);
// exclusive scan of the tile-major count matrix in range-major order: column sums per block of tiles, scan_u32 over them, running sums
u32 bkt_scan_blocks(u32 NT);
void launch_bkt_colsum(const u32* mat, u32 NT, u32 R, u32* partT /*R x bkt_scan_blocks(NT)*/, hipStream_t st);
void launch_bkt_colscan(u32* mat, u32 NT, u32 R, const u32* baseT, hipStream_t st);
// keys (64-bit, (subject, diagonal, qpos) order per bucket) or, when words32 is given instead, the sorted 32-bit words themselves with
// bit 31 set on the first word of every bucket (k_ungap's W32 input)
void launch_bkt_group(const u32* hits, const u32* bext, u32 nb, const BktLayout& L, const KeyLayout& kl, u64* keys, u32* words32, u32* fallback,
                      hipStream_t st);
// best diagonal per (query, subject), bucket by bucket: pass records binned into the hit buckets (count -> scan -> scatter, first-touch
// keys computed on the way), then one LDS reduction per bucket, run once to count the candidates and once to write them
void launch_rec_count(const u64* p_qs, u32 n, int bs, const BktLayout& L, u32* bcnt, u32* rnk, hipStream_t st);
void launch_rec_scatter(const u64* p_qs, const u64* p_sd, const u64* p_ft, const u32* rnk, u32 n, const KeyLayout& kl, const BktLayout& L,
                        int ft_bits_entry, int bsp, const u32* roff, const u32* boff, u64* q_qs, u64* q_sd, u64* q_ft, hipStream_t st);
void launch_bkt_best(bool write, const u64* q_qs, const u64* q_sd, const u64* q_ft, const u32* boff, u32 nb, const BktLayout& L, int bs,
                     u32 seq_lo, u32* ccnt, u64* c_ft, u32* c_q, u32* c_rec, int bsp, int idx_bits /*> 0: c_ft receives sort words*/, hipStream_t st);
// sort words (first-touch word << idx_bits | position in the query's segment) -> the query's records in first-touch order, its count
// (word_bits = idx_bits + width of the first-touch word; *fallback |= 2 when a query's digit group exceeds the LDS sort: order the pass otherwise)
void launch_cand_order_seg(const u64* words, const u32* seg, u32 nqp, u32 qa, int idx_bits, int word_bits, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                           u32* fallback, hipStream_t st);
void launch_emit_cands_seg(const u64* sorted, const u32* seg, u32 nqp, u32 qa, int idx_bits, const u32* c_rec, u32* out_q, u32* out_rec, u32* qcnt,
                           hipStream_t st);
size_t sort_cand_keys_seg_temp_bytes(size_t n, u32 nseg, int begin_bit, int end_bit);
void sort_cand_keys_seg(void* temp, size_t temp_bytes, const u64* in, u64* out, size_t n, u32 nseg, const u32* seg, int begin_bit, int end_bit,
                        hipStream_t st);

void launch_round_counts_spec(const u32* ntask, const u32* ntile, const u32* roffc, const u32* rk_slot, const u32* qcoff, const u32* st_state, u32 nq,
                              double max_miss, u32 minr, const AlnTask* tasks, const u32* toff, const u32* qoff, const u32* roff, const int* bittab,
                              int bittab_n, i64 D, double expect, u32* rcnt, u32* tcnt_pk, u32* scnt, u32* any_rank, hipStream_t st);
void launch_round_idx_spec(const u32* tcnt_pk, const u32* scnt, const u32* poff, const u32* soff, const u32* toff, const u32* ntask, const u32* ntile,
                           const u32* roffc, const u32* rk_slot, const u32* st_state, u32 nq, u32* ridx, u32* sidx, hipStream_t st);
void launch_trace_flags(const u32* sel_idx, u32 n, const u32* tpos, u32* flags, hipStream_t st);
void launch_trace_split(const u32* sel_idx, u32 n, const u32* flags, const u32* fscan, u32* list_b, u32* list_a, hipStream_t st);
// k_align.hip: the two halves of launch_align(..., true) on their own (speculative traces: the walk runs long after the alignment)
void launch_align_traced(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_scls, const u8* q_scls4, const u32* qoff, const u8* r_scls,
                         const u8* r_scls4, const u32* roff, const signed char* b62g, u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out,
                         u32* tpos_out /*tofs: receives tpos_base + tofs[t]*/, u32 tpos_base, hipStream_t st, u32 n_wide, PkCls pk);
// k_align16.hip: list positions [t0, t1) by the packed kernel, traces in the same layout (codes = tags: AlnRes.pad = 1)
void launch_align_pk_traced(const AlnTask* tasks, const u32* ridx, u32 t0, u32 t1, PkCls pk, const u32* qoff, const u32* roff, const signed char* b62g,
                            u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out, u32* tpos_out, u32 tpos_base, hipStream_t st);
void launch_traceback(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u32* qoff, const u8* r_res, const u32* roff,
                      const u32* trace, u32 trace_stride, const u32* tpos, AlnRes* out, hipStream_t st);
void launch_traceback_tofs(const AlnTask* tasks, const u32* ridx, u32 ntasks, const u8* q_res, const u32* qoff, const u8* r_res, const u32* roff,
                           const u32* trace, u32 trace_stride, const u32* tofs, AlnRes* out, hipStream_t st);
void launch_stop_round_w(const AlnTask* tasks, const AlnRes* res, const u32* qcoff, const u32* ntask, const u32* ntile, const u32* roffc,
                         const u32* rk_slot, const u32* toff, const u32* rcnt, u32 nq, const u32* qoff, const u32* roff, const int* bittab,
                         int bittab_n, i64 D, double expect, double max_miss, i64 v, u32* sel, u32* st_state, int* bits,
                         unsigned long long* qcells /*[nq], += cells of the round*/, hipStream_t st);
void launch_sum_u64(const unsigned long long* x, u32 n, unsigned long long* total, hipStream_t st);
// keys[t] = 8191 - band rows of task t, | 8192 when the packed aligner can take it (n_wide != null: the others are counted there; null: all can)
void launch_task_rows(const AlnTask* tasks, const u32* ridx, u32 n, const u32* qoff, const u32* roff, const u32* qbound, const u32* rbound, int pk_len,
                      u32 pk_score, u32* n_wide, unsigned long long* cells_wide /*+= their band cells*/, u64* keys, hipStream_t st);
