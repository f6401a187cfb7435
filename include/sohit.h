/* sohit.h -- C ABI of libsohit.so, the MI355X-native replacement for SwiftOrtho's
 * seed-and-extend search core.
 *
 * What it replaces.  The reference has no in-process API for this path: its
 * bin/find_hit.py shells out to the RPython-built native `lib/fsearch-c`, one process
 * per query block (find_hit.py:119-132, 191-192), and the native's whole interface is
 * its flag list (lib/fsearch.py:3187-3188, 3215-3216) plus the 16-column row it prints
 * (fsearch.py:3242-3243).  Each entry point below therefore cites the piece of that
 * process boundary it stands in for.  bin/find_hit.py of THIS repository binds these
 * symbols with ctypes; INTEGRATION.md shows the stub a SwiftOrtho maintainer would add.
 *
 * Conventions: plain C types only; inputs are borrowed, result buffers are allocated
 * by the library and released with so_free_hits(); every function that can fail
 * returns 0 on success / non-zero on error and leaves a message retrievable with
 * so_last_error(); nothing aborts.  One so_ctx per GPU per process; a ctx is not
 * thread-safe.  There is NO CPU fallback: without a usable HIP device so_create()
 * fails.
 */
#ifndef SOHIT_H
#define SOHIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOHIT_ABI_VERSION 3

typedef struct so_ctx so_ctx;

/* The native's flags (fsearch.py:3187-3188; defaults as driven by find_hit.py:227-228). */
typedef struct so_params {
    const char *seeds;    /* -s  comma-separated spaced-seed patterns, e.g. "111111"          */
    const char *alphabet; /* -r  reduced alphabet groups; '/' separates several alphabets     */
    int64_t nc;           /* -M  number of hash buckets NC; < 1 = the reference's default (2228-2231) */
    int64_t chunk;        /* -c  reference sequences per index chunk (fsearch.py:2283-2295)   */
    int64_t step;         /* -j  reference seed step (2244); queries always use step 1 (2658) */
    int64_t max_hits;     /* -v  rows reported per query                                      */
    int64_t thr;          /* -t  >=1 overrides the per-chunk mu+2sd seed-frequency threshold  */
    double expect;        /* -e  e-value cutoff                                               */
    double max_miss;      /* -m  early-stop ratio (fsearch.py:2970, 3052-3054)                */
    int32_t filter;       /* -F  1 = SEG-like query masking ('T'), 0 = off                    */
    int32_t profile;      /* 1 = time the headline kernels with HIP events (so_get_counters)  */
} so_params;

/* One reported alignment == one output row (fsearch.py:3074-3076, 3242-3243).
 * qst/sst are the 1-based starts exactly as printed (reference prints qst+1, sst+1). */
typedef struct so_hit {
    int64_t qidx;     /* query ordinal in the query file (column 15)            */
    int64_t sidx;     /* subject ordinal in the reference file                  */
    double identity;  /* matches * (100. / aln), full precision (column 3 is truncated text) */
    double evalue;    /* D * qlen * slen * 2^-bit (column 11)                   */
    int32_t aln;      /* alignment length (column 4)                            */
    int32_t mis;      /* non-identical columns, gaps included (column 5)        */
    int32_t gap;      /* gap "openings": ceil(run/2) per gap run (column 6)     */
    int32_t qst, qed; /* columns 7, 8                                           */
    int32_t sst, sed; /* columns 9, 10                                          */
    int32_t bit;      /* truncated integer bit score (column 12)                */
    int32_t qlen;     /* column 13                                              */
    int32_t slen;     /* column 14                                              */
    int32_t matches;  /* identical columns                                      */
    int32_t ungapped; /* ungapped (seed-stage) score of the candidate           */
} so_hit;

/* Work and time counters of the last so_build_index()/so_search*() calls. */
typedef struct so_counters {
    int64_t n_queries, query_aa;       /* queries searched, their residues                 */
    int64_t ref_seqs, ref_aa, n_chunks;
    int64_t seed_windows;              /* query seed windows hashed                        */
    int64_t seed_hits;                 /* index entries visited by the lookup kernel (H)   */
    int64_t groups;                    /* distinct (subject, diagonal) groups              */
    int64_t candidates;                /* candidates with ungapped score >= 25             */
    int64_t alignments;                /* banded alignments computed                       */
    int64_t cells;                     /* banded DP cells computed                         */
    int64_t rows;                      /* reported rows                                    */
    int64_t index_entries;             /* sum of index entries over chunks                 */
    /* headline-kernel timing (HIP events on the ctx stream; only when params.profile)     */
    int64_t lookup_launches;           /* launches of the seed-lookup kernel               */
    double lookup_ms;                  /* their summed duration                            */
    int64_t lookup_bytes;              /* algorithmic bytes read by them       */
    int64_t bounds_launches;           /* launches of the query hash+bucket-bounds kernel  */
    double bounds_ms;
    int64_t bounds_bytes;
    int64_t align_launches;
    double align_ms;
    double index_ms, seed_ms, group_ms, phase2_ms, total_ms; /* host-side stage wall times */
    int64_t count_launches;            /* launches of the lookup kernel's COUNT pass (bucketed binning) */
    double count_ms;                   /* their summed duration; lookup_* then describe its SCATTER pass */
    /* which path the work took (round 4: both are per-task / per-pass decisions, not per-batch ones) */
    int64_t hits_bucketed;             /* seed hits binned by the bucketed passes (k_bkt_pass); the rest took the sorted path */
    int64_t align_wide;                /* score-only alignments the packed 16-bit aligner could not take (32-bit kernel)       */
    int64_t cells_wide;                /* their band cells (`cells` counts every task of the early-stop rounds once)            */
    int64_t seed_passes;               /* seed passes run (one per length class and chunk; sparse neighbouring classes share one) */
    /* round 5 (ABI 2).  With SOHIT_UG_COUNT=1 in the environment of so_create the extension kernels run their counting instances: */
    int64_t ungap_steps;               /* b62 lookups of the ungapped extension = the reference's `flag` (fsearch.py:2467, 2482)  */
    int64_t groups_single;             /* groups of one seed hit, extended by k_ungap1                                            */
    int64_t groups_chain;              /* groups of two and more hits, chained by k_ungap2 (the rest of `groups`: k_ungap)        */
    /* round 6 (ABI 3): the third kernel of the seed-lookup + diagonal-binning stage, so that the STAGE (count + scatter + grouping) can be
     * priced against the HBM roofline, not one pass of it (params.profile) */
    int64_t bgroup_launches;           /* launches of the bucket grouping kernel (k_bkt_group)                                    */
    double bgroup_ms;                  /* their summed duration                                                                   */
} so_counters;

/* Lifetime.  Replaces: spawning `fsearch-c` with its flags (find_hit.py:119-123). */
so_ctx *so_create(int device, const so_params *params);
void so_destroy(so_ctx *ctx);
const char *so_last_error(const so_ctx *ctx); /* ctx may be NULL: last so_create() error */
int so_abi_version(void);
/* One tuning / diagnostic switch (the SOHIT_* names of swiftortho_amd/csrc/tune.h, with or without the prefix; none changes results) for
 * this context from now on.  so_create reads the same names from the environment once. */
int so_set_option(so_ctx *ctx, const char *name, const char *value);

/* Reference side.  Replaces: Fasta(open(ref)) + DB.makedb()/build_msav() per chunk
 * (fsearch.py:2975-2979, 2990, 2208-2295).  r_lo/r_hi are -L/-U (-1 = all).
 * so_load_ref* parses the FASTA and makes the residues resident in HBM;
 * so_build_index builds every chunk's index on the device (idempotent until the next load). */
int so_load_ref(so_ctx *ctx, const char *fasta_path, int64_t r_lo, int64_t r_hi);
int so_load_ref_mem(so_ctx *ctx, const char *fasta_bytes, int64_t nbytes, int64_t r_lo, int64_t r_hi);
int so_build_index(so_ctx *ctx);
int so_drop_index(so_ctx *ctx); /* forget the built chunk indexes (residues stay resident) */
/* Replaces: Fasta.load (fsearch.py:2355-2444).  Reads the chunk indexes `<prefix>.<k>.idx / .soas / .bin`, k = 0, 1, ... (the files
 * Fasta.write produces, fsearch.py:2298-2352; swiftortho_amd.fsearch.makedb writes the same bytes) and makes them resident in place
 * of a built index: locus slots -> entries, start[] -> bucket directory, threshold from the .bin trailer.  As in the reference the
 * files hold no sequences -- so_load_ref() the FASTA file first -- and the context must have been created with the trailer's -M, -s
 * and -r (an error otherwise, as when sequence lengths and the .soas file disagree).  Searches then run from the loaded chunks until
 * the next so_load_ref / so_drop_index. */
int so_load_index(so_ctx *ctx, const char *prefix);

/* Query side.  Replaces: Fasta(open(qry)) (fsearch.py:2971-2973).  Parses the FASTA and
 * makes the query residues resident in HBM. */
int so_load_queries(so_ctx *ctx, const char *fasta_path);
int so_load_queries_mem(so_ctx *ctx, const char *fasta_bytes, int64_t nbytes);
int64_t so_num_queries(const so_ctx *ctx);
int64_t so_num_refs(const so_ctx *ctx);
int64_t so_query_len(const so_ctx *ctx, int64_t qidx);
int64_t so_ref_len(const so_ctx *ctx, int64_t sidx); /* residues of reference sequence sidx (the index exporter's `soas`) */

/* The search.  Replaces: blastp(qry, ref, ..., st=-l, ed=-u) (fsearch.py:2968-3121):
 * queries [q_lo, q_hi) of the loaded query file against the loaded reference; rows come
 * back in the order the reference yields them (ascending query, descending bit).
 * so_search() = so_load_queries() + so_build_index() (if needed) + so_search_loaded(). */
int so_search_loaded(so_ctx *ctx, int64_t q_lo, int64_t q_hi, so_hit **hits, int64_t *n_hits);
int so_search(so_ctx *ctx, const char *qry_fasta_path, int64_t q_lo, int64_t q_hi, so_hit **hits, int64_t *n_hits);
/* Releases a result array.  The library may keep ONE released array (the largest between 1 MiB and 2 GiB) for the next search of this process
 * instead of returning its pages to the system -- a 100k-protein result is 130 MB and unmapping + refaulting it costs ~17 ms per
 * search; so_destroy() drops it, SOHIT_HIT_CACHE=0 disables it. */
void so_free_hits(so_hit *hits);

/* Multi-GPU support.  The reference runs one fsearch-c process per query block and joins the part files with `cat`
 * (find_hit.py:107-146); here one process per GPU searches a query shard and the hit records are exchanged over RCCL, so
 * they must be able to stay in HBM until after the exchange:
 * so_search_device() = so_search_loaded() whose so_hit records (same 80-byte layout, same values) are left in device
 * memory owned by the ctx (*d_hits, valid until the next so_search* call on the ctx or so_destroy);
 * so_device_hits_copy() copies the first n_hits of them device-to-device into a caller-owned device buffer (e.g. the
 * send buffer of a collective).
 * so_query_work() fills work[0 .. q_hi-q_lo) with the number of index entries each query visits over all chunks (one
 * hash + bucket-bounds + frequency-cap pre-pass, no search): the cost estimate used to balance query shards. */
int so_search_device(so_ctx *ctx, int64_t q_lo, int64_t q_hi, const so_hit **d_hits, int64_t *n_hits);
int so_device_hits_copy(so_ctx *ctx, void *dst_device, int64_t n_hits);
int so_query_work(so_ctx *ctx, int64_t q_lo, int64_t q_hi, uint64_t *work);

/* Output.  Replaces: the row formatter of entry_point (fsearch.py:3234-3258; f2s 43-61):
 * writes the 16-column tab-separated rows for hits of the currently loaded query and
 * reference files.  mode "w" or "a" (-O).  so_format_hit() renders one row into buf. */
int so_write_sc(so_ctx *ctx, const so_hit *hits, int64_t n_hits, const char *path, const char *mode);
int64_t so_format_hit(so_ctx *ctx, const so_hit *hit, char *buf, int64_t cap);

/* Introspection (tests, bench). */
/* switch params.profile (HIP-event kernel timers + stage laps, which synchronise the stream) on or off */
int so_set_profile(so_ctx *ctx, int on);
/* NC actually in use: -M, or the reference's `bins` default when -M < 1 (fsearch.py:2228-2231) */
int64_t so_bucket_count(const so_ctx *ctx);
/* Host-only helper of the row formatter (tests): for v[0..n) writes "<%f of v>\t<f2s(v)>\n" lines (f2s: fsearch.py:43-61) into out;
 * returns the bytes written or -1 when cap is too small (allow 1400 per value). */
int64_t so_fmt_rows(const double *v, int64_t n, char *out, int64_t cap);
int so_get_counters(const so_ctx *ctx, so_counters *out);
int so_reset_counters(so_ctx *ctx);
/* "stage=ms;stage=ms;..." wall-clock laps of the pipeline stages (only with params.profile) */
int64_t so_timing_report(const so_ctx *ctx, char *buf, int64_t cap);
int64_t so_chunk_threshold(const so_ctx *ctx, int64_t chunk);
int64_t so_chunk_entries(const so_ctx *ctx, int64_t chunk);
/* copies start[0..NC] (uint32, NC+1 values) / entries (uint64) of one chunk's index to host, every bucket's members in descending entry
 * order -- the slot order of the reference's CSR (fsearch.py:2240-2266) and of its index files.  (The build's grouping kernels place a
 * bucket's members with atomics; they are put in order by the chunk's first dense seed pass or by this call, whichever comes first.) */
int so_chunk_download(so_ctx *ctx, int64_t chunk, uint32_t *start, uint64_t *entries);
/* masked (SEG-filtered, upper-cased) bytes of query qidx as used for seeding and alignment */
int64_t so_masked_query(so_ctx *ctx, int64_t qidx, char *buf, int64_t cap);
/* candidates [subject, ungapped score, qi, qj] x uint32 of query qidx from the last
 * so_search_loaded() call, in the order the reference's spill file holds them (chunk-major) */
int64_t so_query_candidates(so_ctx *ctx, int64_t qidx, uint32_t *out4, int64_t cap);

/* Markov clustering of one block of the orthology graph.  Replaces: the matrix loop of bin/find_cluster.py
 * `mcl` (652-689) with `normalize` (636-646) as `mcl_xyz` (1425-1467) calls it on a float32 scipy csr_matrix -- column
 * normalisation, expansion (sparse x sparse), inflation, pruning below `prune`, at most max_rounds rounds, convergence test
 * every check_every-th round -- with scipy's arithmetic order (csrc/mcl.hip).  Input: an n x n CSR matrix (rows = indptr,
 * columns = indices, float32 data; inputs are borrowed).  Output: the matrix the loop ends with, in scipy's storage order and
 * with its explicitly stored zeros (the reference's read-out depends on both); arrays allocated by the library, released with
 * so_mcl_free().  Returns 0 / non-zero with a message in so_mcl_last_error().  No so_ctx: the call owns a stream of `device`. */
typedef struct so_mcl_result {
    int64_t n, nnz;
    int32_t rounds;    /* rounds executed                                  */
    int32_t converged; /* 1 = left the loop through the convergence test   */
    int64_t *indptr;   /* n + 1                                            */
    int32_t *indices;  /* nnz                                              */
    float *data;       /* nnz                                              */
} so_mcl_result;
int so_mcl(int device, int64_t n, const int64_t *indptr, const int32_t *indices, const float *data, double inflation, int32_t max_rounds,
           int32_t check_every, double prune, double rtol, double atol, so_mcl_result *out);
void so_mcl_free(so_mcl_result *result);
const char *so_mcl_last_error(void);

/* Host-side tokeniser of tab-separated text for the stages behind the search (csrc/tsv.hip; no device, no so_ctx).  Replaces: the
 * per-line `split('\t')` + `float()` loops of bin/find_orth.py (blastparse, 58-125) and bin/find_cluster.py (1425-1467) as the
 * numpy tokeniser of swiftortho_amd/find_orth.py restates them.
 *   so_tsv_lines  start offset of every line of buf[0..n) (a line ends at its '\n'); returns the number of lines.
 *   so_tsv_scan   for every line and each requested column cols[c]: [beg, beg + len) of the field (numeric columns: stripped of ASCII
 *                 white space), the number of tabs of the line, and for numeric columns the value with status 0 = plain decimal
 *                 number (strtod), 1 = empty or missing, 2 = anything else (the caller applies Python's float()).  Arrays are
 *                 [ncols][nline].  numeric[c]: 0 = text column (bounds), 1 = number (bounds, value, status), 2 = number, status only,
 *                 3 = number, value and status -- what is not asked for is not written.  Returns 0.
 *   so_tsv_codes  two columns of byte strings -> codes into their sorted distinct list (byte order, a proper prefix first: numpy's
 *                 order of fixed-width byte strings); returns the number of distinct strings, or minus it when cap is too small. */
int64_t so_tsv_lines(const char *buf, int64_t n, int64_t *line_start, int64_t cap);
int so_tsv_scan(const char *buf, int64_t n, const int64_t *line_start, int64_t nline, int32_t ncols, const int32_t *cols, const uint8_t *numeric,
                int32_t *ntab, int64_t *beg, int32_t *len, double *val, uint8_t *status);
int64_t so_tsv_codes(const char *buf, int64_t nrows, const int64_t *beg_a, const int32_t *len_a, const int64_t *beg_b, const int32_t *len_b,
                     int64_t *code_a, int64_t *code_b, int64_t *name_beg, int32_t *name_len, int64_t cap);
/* The output side of the same stages: n lines  kind \t name[x] \t name[y] \t repr(v) \n  (find_orth.py prints its relations with
 * Python's repr of the score, 1181-1226); names = concatenated ids, name_off[k] .. name_off[k + 1] the k-th.  Returns the bytes written,
 * or minus the bytes needed when cap is too small.  so_py_repr: repr() of n doubles, one per line (tests). */
int64_t so_format_pairs(const char *kind, int32_t kind_len, const char *names, const int64_t *name_off, const int64_t *x, const int64_t *y,
                        const double *v, int64_t n, char *out, int64_t cap);
int64_t so_py_repr(const double *v, int64_t n, char *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SOHIT_H */
