// tune.h -- every SOHIT_* switch of libsohit.so in ONE table.  The environment is read ONCE, by so_create (Tune::read), into the
// context's Tune; the launch helpers in the k_*.hip files read the current context's copy through tune() -- no getenv() anywhere else.
// None of the switches changes results, except the two ablation variants marked so.  tools/diag/README.md lists what they are for.
#pragma once
#include <cstdlib>

//   kind  field            environment variable      default   meaning
//   B = boolean (unset: default, else atoi != 0), I = integer (atoll), D = double (atof), P = presence (set to anything)
#define SOHIT_TUNE_TABLE(X)                                                                                                                                     \
    /* ---- ungapped extension ---- */                                                                                                                          \
    X(B, ug1, "SOHIT_UG1", 1, "bucketed passes: singleton groups to k_ungap1 (0: everything to k_ungap)")                                                     \
    X(B, ug1_chain, "SOHIT_UG1_CHAIN", 1, "... and the groups of two and more hits to k_ungap2 (0: to k_ungap, which then skips the singletons)")              \
    X(B, count_steps, "SOHIT_UG_COUNT", 0, "counting instances of the extension kernels: so_counters.ungap_steps, groups_single, groups_chain")                \
    X(B, ungapq, "SOHIT_UNGAPQ", 1, "sparse passes of queries up to 512 residues: hits alone on their diagonal by k_ungapq, a wave per query (0: every hit through the sorted keys)")     \
    X(B, ug_w32, "SOHIT_UG_W32", 1, "bucketed passes hand k_ungap the buckets' 32-bit words (0: 64-bit keys; no k_ungap1 then)")                               \
    /* ---- seed stage: which path a pass takes ---- */                                                                                                         \
    X(B, bucket, "SOHIT_BUCKET", 1, "bucketed diagonal binning (0: every pass on the sorted path)")                                                            \
    X(I, bucket_min, "SOHIT_BUCKET_MIN", 192, "hits per bucket below which a pass takes the sorted path (0: bucketed whenever possible)")                      \
    X(I, bucket_avg, "SOHIT_BUCKET_AVG", 2560, "average bucket size the band-range width is chosen for")                                                       \
    X(I, count_tab, "SOHIT_COUNT_TAB", 1, "bucketed passes: hits per (tile, range) from the range boundaries of the ordered index buckets when a batch brings a chunk two passes and more (0: always the counting pass over the entries; 2: always both, compared -- tests; 3: always)") \
    X(B, bucket_best, "SOHIT_BUCKET_BEST", 1, "best diagonal per subject by the per-bucket reduction (0: sort of the pass records)")                           \
    X(B, bands, "SOHIT_BANDS", 1, "diagonal bands follow the subjects' lengths (0: one band per subject)")                                                     \
    X(B, lk_wide, "SOHIT_LK_WIDE", 0, "8-byte index addends whatever the field widths")                                                                        \
    X(B, qclass, "SOHIT_QCLASS", 1, "queries of a batch in length-class order (0: file order)")                                                                \
    X(B, pass_merge, "SOHIT_PASS_MERGE", 1, "neighbouring length classes that take the sorted path anyway share one pass")                                     \
    X(B, ksc_lazy, "SOHIT_KSC_LAZY", 1, "k-mer orders computed for the queries that reach their frequency cap, when they do (a query below it keeps every window whatever the order; 0: for every query, with the batch)") \
    X(B, bounds_ahead, "SOHIT_BOUNDS_AHEAD", 1, "the next chunk's bucket bounds (k_bounds) on the side stream beside this chunk's seed stage (0: in front of its own)")     \
    X(I, segsort, "SOHIT_SEGSORT", 1, "sorted path: segmented sort of the keys inside each query (0: device-wide sort)")                                        \
    X(I, write_threads, "SOHIT_WRITE_THREADS", 48, "so_write_sc: formatter threads (at most the host's cores)")                                                                          \
    /* ---- candidate order ---- */                                                                                                                             \
    X(I, cand_limit, "SOHIT_CAND_LIMIT", 0, "tests: candidate-store size at which a batch is split (0: 2^32 - 16)")                                            \
    /* ---- phase 2 ---- */                                                                                                                                     \
    X(B, align_pk, "SOHIT_ALIGN_PK", 1, "packed 16-bit score-only aligner where the scores fit")                                                               \
    X(B, align_lane, "SOHIT_ALIGN_LANE", 1, "score-only rounds: one lane per alignment pair (k_align_lane) when no sequence reaches 4096 residues (0: k_align_pk)")            \
    X(B, align_sort, "SOHIT_ALIGN_SORT", 1, "launch lists ordered by band rows")                                                                               \
    X(I, trace_wave_rows, "SOHIT_TRACE_WAVE_ROWS", 1024, "traceback: bands of this many rows and more are walked by a wave of their own (0: never)")         \
    X(I, trace_wave_max, "SOHIT_TRACE_WAVE_MAX", 32768, "... among the first this many positions of a launch list")                                            \
    X(I, spec, "SOHIT_SPEC", -1, "speculative traces in the first round: 0 off, 1 on, -1 from 2^21 tasks on")                                                  \
    X(D, spec_slack, "SOHIT_SPEC_SLACK", 1e6, "... the guess tests the ungapped score against expect x this (tests: 1e30 = every first-round task traced, 1e-30 = none)")          \
    X(I, spec_cap, "SOHIT_SPEC_CAP", -1, "tests: most speculative traces kept (-1: all)")                                                                      \
    X(I, emit_parts, "SOHIT_EMIT_PARTS", 4, "... and otherwise")                                                                                               \
    X(I, emit_min_rows, "SOHIT_EMIT_MIN_ROWS", 1 << 18, "rows below which the emission is one part")                                                           \
    X(P, test_oom_phase2, "SOHIT_TEST_OOM_PHASE2", 0, "tests: phase 2 of the first batch fails once with an out-of-memory error")                              \
    /* ---- batches, memory ---- */                                                                                                                             \
    X(I, batch, "SOHIT_BATCH", 0, "queries per device batch (0: 131072, lowered when the candidate estimate asks)")                                            \
    X(I, max_hits, "SOHIT_MAX_HITS", 0, "seed hits per pass (0: 2^30)")                                                                                        \
    X(I, poison, "SOHIT_POISON", -1, "tests: byte every fresh device allocation is filled with (-1: none)")                                                    \
    X(B, hit_cache, "SOHIT_HIT_CACHE", 1, "one released result array is kept for the next search")                                                             \
    X(P, keep_cands, "SOHIT_KEEP_CANDS", 0, "tests: keep every query's candidate list (so_query_candidates)")                                                  \
    X(P, keep_masked, "SOHIT_KEEP_MASKED", 0, "tests: keep the masked queries (so_masked_query)")                                                              \
    /* ---- index ---- */                                                                                                                                       \
    X(P, exact_threshold, "SOHIT_EXACT_THRESHOLD", 0, "threshold always by the sequential fp64 replay")                                                        \
    X(I, dir_max, "SOHIT_DIR_MAX", -1, "largest -M served by the bitmap + rank directory (-1: 2^31)")                                                          \
    /* ---- traces ---- */                                                                                                                                      \
    X(P, debug, "SOHIT_DEBUG", 0, "per-pass lines on stderr")                                                                                                  \
    X(P, debug_index, "SOHIT_DEBUG_INDEX", 0, "wall laps of the index build on stderr")

struct Tune {
#define X_B(f, d) bool f = (d) != 0;
#define X_I(f, d) long long f = (d);
#define X_D(f, d) double f = (d);
#define X_P(f, d) bool f = false;
#define X(kind, field, env, dflt, text) X_##kind(field, dflt)
    SOHIT_TUNE_TABLE(X)
#undef X
#undef X_B
#undef X_I
#undef X_D
#undef X_P
    void read() {
#define X_B(f, e) if (const char* v = getenv(e)) f = atoi(v) != 0;
#define X_I(f, e) if (const char* v = getenv(e)) f = strtoll(v, nullptr, 0);
#define X_D(f, e) if (const char* v = getenv(e)) f = atof(v);
#define X_P(f, e) f = getenv(e) != nullptr;
#define X(kind, field, env, dflt, text) X_##kind(field, env)
        SOHIT_TUNE_TABLE(X)
#undef X
#undef X_B
#undef X_I
#undef X_D
#undef X_P
    }
};

// the switches of the context whose API call runs on this thread (set by so_create and by every guarded entry point)
const Tune& tune();
void set_tune(const Tune* t);
