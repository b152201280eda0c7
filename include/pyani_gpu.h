/* pyani_gpu.h — C ABI of libpyani_gpu.so, the MI355X (gfx950) engine behind pyani's hot path.
 *
 * pyani (the reference) is 100 % Python and has no FFI: the boundary this library slots behind is the
 * module API of pyani/tetra.py (and, next, the process/file boundary between pyani/anim.py and
 * nucmer/delta-filter).  Each entry point below names the reference interface it replaces (file:line,
 * relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a pyani maintainer would add.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer; the library keeps no caller pointer after return.
 *   - every function returns PG_OK (0) or a negative PG_E_* code; pg_last_error(ctx) has the message.
 *   - no exceptions cross the ABI; there is NO CPU fallback: without a usable GPU pg_create fails.
 *   - one context per process and device; calls on one context must not be concurrent.
 *   - k-mer index convention: first base most significant, A=0 C=1 G=2 T=3 (== sorted() string order
 *     used by pyani/tetra.py:176), i.e. "ACGT" -> 0*64 + 1*16 + 2*4 + 3.
 */
#ifndef PYANI_GPU_H
#define PYANI_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_E_ARG (-1)      /* bad argument */
#define PG_E_NODEVICE (-2) /* no usable HIP device (the product path never falls back to CPU) */
#define PG_E_HIP (-3)      /* HIP runtime error, see pg_last_error */
#define PG_E_IO (-4)       /* file could not be read */
#define PG_E_NOMEM (-5)
#define PG_E_KEYSET (-6)   /* genomes have different observed-tetramer key sets: pyani raises AssertionError (tetra.py:174-175) */
#define PG_E_EMPTY (-7)    /* empty key set: pyani raises ZeroDivisionError (tetra.py:181) */
#define PG_E_RNA (-8)      /* sequence contains U/u: Biopython complements it asymmetrically (U->A); unsupported */
#define PG_E_CAPACITY (-9) /* a per-pair work buffer overflowed (pg_anim_result.status; pg_anim_alignments_batch) */
#define PG_E_INTERNAL (-10) /* an internal consistency check failed (traceback pass), see pg_last_error */

typedef struct pg_ctx pg_ctx;

/* ---- context ------------------------------------------------------------------------------------------- */
const char* pg_version(void);
/* device: HIP device ordinal (use LOCAL_RANK for one-process-per-GPU jobs). */
int pg_create(pg_ctx** out, int device);
void pg_destroy(pg_ctx* ctx);
const char* pg_last_error(const pg_ctx* ctx);
/* blocks until all work queued by this context has finished */
int pg_sync(pg_ctx* ctx);

/* ---- genome store: 2-bit codes + 1-bit "clean" mask resident in HBM ------------------------------------- */
/* Replaces Bio.SeqIO.parse(filename, "fasta") + str(rec.seq).upper() (pyani/tetra.py:98-99) and
 * pyani_files.get_sequence_lengths (pyani/pyani_files.py:128-142: total_len = sum of len(record)).
 * seq: concatenated record sequences (ASCII, any case, IUPAC allowed); rec_off[0..n_rec]: record boundaries. */
int pg_add_genome(pg_ctx* ctx, const uint8_t* seq, const uint64_t* rec_off, uint32_t n_rec, int32_t* genome_id_out);
/* Same, reading a FASTA file (records = '>' blocks; whitespace inside sequence lines removed). */
int pg_add_fasta(pg_ctx* ctx, const char* path, int32_t* genome_id_out, uint64_t* total_len_out, uint32_t* n_rec_out);
/* Many files at once: read + parse + pack on `threads` host threads (0 = all hardware threads); ids are assigned in
 * input order.  Replaces the per-file Bio.SeqIO loops of pyani_files.get_sequence_lengths (pyani_files.py:128-142) and
 * tetra.calculate_tetra_zscores (tetra.py:66-74) on the ingest side.  Any of the three output arrays may be NULL. */
int pg_add_fasta_batch(pg_ctx* ctx, const char* const* paths, uint32_t n, uint32_t threads, int32_t* genome_ids_out,
                       uint64_t* total_len_out, uint32_t* n_rec_out);
int pg_genome_count(const pg_ctx* ctx);
int pg_genome_length(const pg_ctx* ctx, int32_t genome_id, uint64_t* total_len_out, uint32_t* n_rec_out);
/* Drop all genomes (device arena is kept for reuse). */
int pg_clear_genomes(pg_ctx* ctx);
/* Push not-yet-resident genomes to HBM now (otherwise done lazily by the first compute call). */
int pg_upload(pg_ctx* ctx);
/* Bytes of the packed representation the count kernel streams for these genomes (2-bit codes + 1-bit mask,
 * padding excluded): the "algorithmic bytes" of SURVEY.md §8(d).  genome_ids == NULL means all genomes. */
int pg_tetra_algorithmic_bytes(const pg_ctx* ctx, const int32_t* genome_ids, uint32_t n, uint64_t* bytes_out,
                               uint64_t* bases_out);

/* ---- TETRA ----------------------------------------------------------------------------------------------
 * kernel 1: exact integer k-mer counts over both strands, including the reference's quirk that the last
 * tetranucleotide of each strand of each record is not counted.  Replaces the counting loops of
 * calculate_tetra_zscore (pyani/tetra.py:98-116).  c2: n x 16, c3: n x 64, c4: n x 256 (host buffers). */
int pg_tetra_counts(pg_ctx* ctx, const int32_t* genome_ids, uint32_t n, uint64_t* c2, uint64_t* c3, uint64_t* c4);

/* Z-scores from counts in the reference's IEEE-754 operation order (pyani/tetra.py:119-138), computed on the
 * device in fp64 without FMA contraction.  z: n x 256; present: n x 256 (1 where the tetramer is a key of the
 * reference's result dict, i.e. c4 > 0). */
int pg_tetra_zscores(pg_ctx* ctx, const uint64_t* c2, const uint64_t* c3, const uint64_t* c4, uint32_t n, double* z,
                     uint8_t* present);

/* kernel 2: all-vs-all Pearson matrix, bit-identical to calculate_correlations (pyani/tetra.py:158-194):
 * sequential sums in tetramer order, diagonal = 1.0.  out: n x n row-major (host).
 * PG_E_KEYSET if the present[] rows differ, PG_E_EMPTY if they are all-zero. */
int pg_tetra_corr(pg_ctx* ctx, const double* z, const uint8_t* present, uint32_t n, double* out);

/* Fused, device-resident pipeline for a batch of stored genomes: counts -> Z -> Pearson with no host round
 * trip in between; replaces calculate_tetra (scripts/average_nucleotide_identity.py:582-612).
 * Any of z_out / present_out / corr_out may be NULL.  Row/column order = order of genome_ids. */
int pg_tetra_matrix(pg_ctx* ctx, const int32_t* genome_ids, uint32_t n, double* z_out, uint8_t* present_out,
                    double* corr_out);
/* Asynchronous form used for throughput runs: queues one pass (results land in an internal pinned buffer,
 * fetched by pg_tetra_matrix_fetch after pg_sync).  Queue any number of passes, then pg_sync once. */
/* fetch_z != 0: Z and presence are copied back too (otherwise only the matrix: the reference's calculate_tetra
 * returns just the correlation DataFrame). */
int pg_tetra_matrix_enqueue(pg_ctx* ctx, const int32_t* genome_ids, uint32_t n, int fetch_z);
int pg_tetra_matrix_fetch(pg_ctx* ctx, uint32_t n, double* z_out, uint8_t* present_out, double* corr_out);

/* Multi-GPU building blocks (one process per GPU; the exchange itself is an RCCL all-gather done by the host
 * layer on these device buffers — SURVEY.md §8(e)).  d_* are DEVICE pointers owned by the caller.
 * Both calls return after their work has completed on the device. */
int pg_tetra_zscores_dev(pg_ctx* ctx, const int32_t* genome_ids, uint32_t n, double* d_z, uint8_t* d_present);
/* rows [row0, row0+nrows) of the n x n matrix from the full (all-gathered) d_z / d_present; d_out: nrows x n */
int pg_tetra_corr_rows_dev(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n, uint32_t row0,
                           uint32_t nrows, double* d_out);

/* ---- ANIm ------------------------------------------------------------------------------------------------
 * In-process replacement for the `nucmer --mum` + `delta-filter -1` jobs pyani builds in
 * construct_nucmer_cmdline (pyani/anim.py:240-289), runs through run_multiprocessing (run_multiprocessing.py:56-152)
 * and reduces with parse_delta (anim.py:292-411).  One result per ORDERED pair:
 *   ref_ids[i] = the genome in nucmer's *reference* role  (pyani's query genome, fname1 of anim.py:280)
 *   qry_ids[i] = the genome in nucmer's *query* role      (pyani's subject genome, fname2)
 * The tuple (ref_aln_len, qry_aln_len, identity, sim_errors) is what parse_delta returns for the pair's .filter
 * file.  status: 0 = ok; PG_ANIM_NO_ALIGNMENT = no alignment survived (parse_delta would raise ZeroDivisionError,
 * anim.py:396); PG_E_CAPACITY = internal buffers overflowed for this pair.
 * MUMmer itself is third-party and absent from the reference tree: behaviour is reconstructed and calibrated against
 * the MUMmer output files the reference's tests hold (every fixture reproduced exactly).
 * reserved = number of alignments BEFORE the 1-to-1 filter (what nucmer's .delta would hold). */
typedef struct {
  int64_t ref_aln_len, qry_aln_len, sim_errors, n_alignments;
  double identity;
  int32_t status;
  int32_t reserved;
} pg_anim_result;
#define PG_ANIM_NO_ALIGNMENT 1
/* maxmatch = 0: pyani's default `nucmer --mum` (anchors unique in both genomes); maxmatch != 0: `--maxmatch`
 * (anim.py:246-289: every maximal match is an anchor) — same pipeline without the uniqueness filter; no MUMmer output
 * for this mode exists among the reference's fixtures, so it is checked against the scalar CPU statement of the same search
 * (genomes with and without repeats).  filter_1to1 = 0 reproduces pyani's --nofilter (reduction over the unfiltered alignments).
 * The pairs may come in any order and may repeat; results are written in the caller's order.  pyani compares every pair in both
 * directions (anim.py:216-233): when (A, B) and (B, A) are both in ONE call they share their seeding (same maximal exact
 * matches), so one call with the whole job is up to 1.6 x faster than the directions in separate calls — same results. */
int pg_anim_pairs(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                  int filter_1to1, pg_anim_result* out);
/* The non-blocking form (round 6; the counterpart of pg_tetra_matrix_enqueue / _fetch): pyani's runner keeps all its cores busy
 * across job boundaries (one multiprocessing.Pool for the whole job list, run_multiprocessing.py:130-144); a blocking pg_anim_pairs
 * pays the sequential tails of its last kernels (one forced re-alignment can hold a single wave for 0.1 s) with the rest of the
 * GPU idle.  pg_anim_pairs_enqueue copies the id lists, starts the call on a host thread of the context and returns a ticket at
 * once; pg_anim_pairs_fetch waits for that call and writes its n_pairs results (the status of the call is fetch's return value).
 * At most TWO calls are in flight per context (PG_E_CAPACITY beyond): they run on disjoint halves of the context's four
 * (stream, scratch) worker slots with half the match budget each, so the tail of call k overlaps the front of call k + 1.
 * Results are those of pg_anim_pairs, bit for bit.  Every ticket must be fetched (pg_destroy waits for unfetched calls). */
int pg_anim_pairs_enqueue(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                          int filter_1to1, uint64_t* ticket);
int pg_anim_pairs_fetch(pg_ctx* ctx, uint64_t ticket, pg_anim_result* out, uint64_t n_pairs);

/* The extension algorithm after seeding and clustering is MUMmer 3.23's own postnuc (extendClusters + its alignment engine:
 * dynamic anti-diagonal band trimmed at breaklen * 3 below the best score, backward search + forced forward re-alignment,
 * MUMmer's tie order): it reproduces every alignment record (coordinates and error counts) of the nucmer output files the
 * reference's tests hold and is what replaces pyani's `nucmer` job (pyani/anim.py:240-289).  There is no other extender: the
 * approximate fixed-band one of rounds 1-2 was retired in round 5.  pg_anim_set_extender stays in the ABI for callers built against the
 * round-4 header: PG_EXTENDER_NUCMER is accepted (a no-op), every other value is PG_E_ARG. */
#define PG_EXTENDER_NUCMER 0
int pg_anim_set_extender(pg_ctx* ctx, int extender);
/* How many host worker threads (each with its own HIP stream and scratch) share one pg_anim_pairs / pg_anib_pairs call: 1 ... 4,
 * default 2 (the launches of two streams overlap: one worker's read-backs and sequential tails hide behind the other's kernels).
 * The counterpart of pyani's --workers inside ONE device (subcmd_anim.py:392-396 spreads jobs over CPU cores; over devices it is
 * pyani_amd.multi.MultiEngine).  Results do not depend on it; 1 gives un-overlapped per-stage timings (pg_profile_get). */
int pg_anim_set_workers(pg_ctx* ctx, int workers);

/* Engine counters of the nucmer extender since the last reset, for measurement (bench.py's VALU-issue roofline: cells per second
 * against the instruction count per cell): out[64].  out[0..2] = calls / anti-diagonals / DP cells of all register engines;
 * out[32 + 4 k + {0, 1, 2}] = the same for the diagonal engine in kernel class k (0 match-to-match gaps, 1 forward extensions,
 * 2 backward searches run ahead, 3 the units' walks, 4 forced runs in the narrow kernel, 5 / 6 forced runs in windows of up to 1024 /
 * 2048 diagonals, 7 forced runs on a group of four waves: 3072 ... 8192 diagonals); the rest: development counters.
 * Synchronises the context's streams. */
int pg_anim_counters(pg_ctx* ctx, uint64_t* out, int reset);

/* Work-memory budget of pg_anim_pairs: at most max_pairs ordered pairs and max_matches exact matches in flight (about 384 bytes
 * of device scratch per match, grown on demand; default 131072 pairs / 512 Mi matches, split over the context's two workers =
 * launches of up to 65536 pairs / 256 Mi matches, ~68 GB each).  Larger calls are split transparently; results do not depend on
 * the split. */
int pg_anim_set_batch_budget(pg_ctx* ctx, uint32_t max_pairs, uint64_t max_matches);

/* The alignment records of ONE ordered pair: the content of the .delta file nucmer would write (kept == 3 marks the
 * records delta-filter -1 keeps, i.e. the .filter file), minus the indel offset lists, which parse_delta ignores
 * (anim.py:374-393); pg_anim_alignments_batch below returns them too.  Coordinates as in MUMmer's alignment header lines
 * (pyani/nucmer.py:333-351): 1-based, closed, relative to the record, qs > qe on the reverse strand; ref_rec / qry_rec
 * = ordinal of the FASTA record (the '>' line names the ids).  At most `cap` records are written, *n_out = how many
 * there are.  Used for alignment-level parity tests and to write recovery files (pyani_amd.anim.write_delta). */
typedef struct {
  int32_t ref_rec, qry_rec;
  int32_t rs, re, qs, qe;
  int32_t errors;
  int32_t kept;   /* 3 = survives delta-filter -1 (bit 0: reference-side LIS, bit 1: query-side LIS) */
} pg_anim_alignment;
int pg_anim_pair_alignments(pg_ctx* ctx, int32_t ref_id, int32_t qry_id, pg_anim_alignment* out, uint32_t cap, uint32_t* n_out);

/* The same for MANY ordered pairs in one call — what pyani's nucmer jobs leave on disk for a whole run (the .delta files of
 * pyani/anim.py:240-289, one per pair; kept == 3 marks the records of the .filter files), without running the search once per
 * pair.  aln_offsets[n_pairs + 1]: pair i owns records [aln_offsets[i], aln_offsets[i + 1]) of the result, in MUMmer's output
 * order (forward-strand alignments of the pair, then reverse-strand ones, each in the order postnuc creates them).
 * with_indels != 0 adds a traceback pass on the GPU (one thread per search / forced piece of the alignments' paths, scalar engine
 * with a backpointer store) and yields every record's .delta indel offset list (pyani/nucmer.py:170-290 parses them: positive =
 * a reference base facing a gap, negative = a query base facing a gap, distances between consecutive indels; the terminating 0
 * is not stored); *n_indels = how many numbers all lists hold together.
 * The result stays in the context until the next call; pg_anim_alignments_read copies it out: out[aln_offsets[n_pairs]],
 * indel_offsets[n_alignments + 1] and indels[*n_indels] (both may be NULL). */
int pg_anim_alignments_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch, int with_indels,
                             uint64_t* aln_offsets, uint64_t* n_indels);
int pg_anim_alignments_read(pg_ctx* ctx, pg_anim_alignment* out, uint64_t* indel_offsets, int64_t* indels);

/* The reduction alone, on alignment records supplied by the caller (e.g. parsed from existing MUMmer .delta/.filter
 * files — pyani's --recovery mode): replaces delta-filter -1 (apply_filter != 0; scripts/delta_filter_wrapper.py:70-93)
 * and parse_delta (anim.py:292-411).  Pair p owns records [offsets[p], offsets[p+1]).  Coordinates are the 1-based
 * closed MUMmer coordinates (qs > qe for reverse-strand hits); rseq / qseq are per-pair sequence (record) ordinals. */
int pg_anim_reduce(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                   const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                   int apply_filter, pg_anim_result* out);

/* ---- ANIb reduction -------------------------------------------------------------------------------------
 * parse_blast_tab of pyani/anib.py:569-667 (mode "ANIb", BLAST+ 15-column rows) for a batch of ordered pairs.
 * Pair p owns rows [offsets[p], offsets[p+1]) in file order; frag[] = ordinal of the query fragment (0-based, < n_frags[p]).
 * Per row: ani_alnlen = length - gaps, ani_alnids = ani_alnlen - mismatch; rows with ani_alnlen/qlen > 0.7 and
 * ani_alnids/qlen > 0.3 are kept, then the first kept row of every fragment.  Outputs per pair: aln_length =
 * sum(ani_alnlen), sim_errors = sum(mismatch) + sum(gaps), pid = mean(pident) over the kept rows in fragment order
 * (0 if none) — sequential fp64 sum, within 1e-12 (relative) of pandas' pairwise mean. */
int pg_anib_reduce(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                   const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen, const double* pident,
                   int64_t* aln_length_out, int64_t* sim_errors_out, double* pid_out);

/* ---- ANIb fragment mode -----------------------------------------------------------------------------------
 * In-process replacement for the BLAST+ jobs pyani builds in construct_blastn_cmdline (pyani/anib.py:451-471: the
 * `fragsize`-nt fragments of the query genome, anib.py:164-203, searched with blastn -task blastn against the subject
 * genome) followed by parse_blast_tab (anib.py:569-667).  One result per ORDERED pair (qry_ids[i] = the fragmented genome,
 * sbj_ids[i] = the BLAST database genome):  aln_length = sum of (length - gaps), sim_errors = sum of (mismatch + gaps),
 * pid = mean pident, each over the FIRST row of every fragment among the rows with coverage > 0.7 and identity > 0.3 (rows in
 * BLAST's order: best score first; anib.py:640-660 filters, then drops duplicates keeping the first) — the tuple parse_blast_tab
 * returns.  BLAST+ is third-party and absent from the reference tree; the search is a restatement of its documented
 * behaviour for this command line (blastn scoring 2 / -3 / 5 / 2, X-drop 150 bits, e-value 1e-15; seeds: exact 16-mers, then
 * blastn's 11-mer words for fragments those leave without a reportable hit), checked against the BLAST+ tables the reference's
 * tests hold (level of agreement per fixture).
 * Limits: fragsize <= 1020 (pyani's default and maximum in practice; larger values are rejected with PG_E_ARG — the fragment's
 * DP lives in LDS).  Query genomes of any bacterial or fungal size: up to 15 872 fragments (16.1 Mb at 1020 nt) the per-fragment
 * counters sit in LDS, beyond that in HBM (round 5: there used to be a hard limit); only a query of more than ~10^6 fragments
 * (1 Gb) comes back with status = PG_E_CAPACITY (n_frags set, everything else 0), the call going on with the others. */
typedef struct {
  int64_t aln_length, sim_errors;
  double pid;
  int32_t n_frags, n_kept;   /* fragments of the query genome / fragments that contributed */
  int32_t status, reserved;  /* 0 = ok, PG_E_CAPACITY = the query genome has more fragments than a launch can hold (see above) */
} pg_anib_result;
int pg_anib_pairs(pg_ctx* ctx, const int32_t* qry_ids, const int32_t* sbj_ids, uint64_t n_pairs, uint32_t fragsize, pg_anib_result* out);
/* The table of ONE ordered pair, row for row as pyani reads it (anib.py:609-624): at most 4 rows per fragment (2 strands x
 * 2 anchors), best score first; frag = 0-based fragment ordinal (frag00001 = 0), coordinates 1-based as BLAST prints them
 * (sstart > send on the minus strand), srec = ordinal of the subject record, score = raw alignment score. */
typedef struct {
  int32_t frag, length, mismatch, gaps, nident, qlen, qstart, qend, sstart, send, srec, score;
} pg_anib_row;
int pg_anib_pair_rows(pg_ctx* ctx, int32_t qry_id, int32_t sbj_id, uint32_t fragsize, pg_anib_row* out, uint32_t cap, uint32_t* n_out);

/* ---- sketch mode: an opt-in ESTIMATE in the shape of pyani's fastANI wrapper -------------------------
 * Replaces the `fastANI -q <query> -r <ref> --fragLen 3000 -k 16 --minFraction 0.2` job of pyani/fastani.py:193-229
 * (construct_fastani_cmdline) and the line parse_fastani_file reads back (fastani.py:231-270): ANI estimate, matching fragments,
 * query fragments.  Definition (pyani_amd/csrc/pg_sketch_core.h): canonical 16-mers sampled 1 in `scale` by a hash (FracMinHash);
 * the query's records cut into non-overlapping fragments of `frag_len` bases; per fragment the share C of its sampled k-mers that
 * occur anywhere in the reference, identity = C^(1/16); a fragment matches with >= 2 hits and identity >= 0.80; ani = mean identity
 * of the matching fragments (a FRACTION, as ComparisonResult.ani), status PG_SKETCH_NO_RESULT when fewer than min_fraction of
 * the fragments match (fastANI writes no line then).  Sketches are built on first use and cached per genome.  An estimate with
 * its own columns: nothing of it enters the exact ANIm / ANIb results.  Limits: k = 16 only; <= 24 576 fragments per query. */
typedef struct {
  double ani;          /* mean identity estimate of the matching fragments, 0 ... 1 (0 when status != 0) */
  int32_t matches;     /* fragments with an identity estimate >= 0.80 */
  int32_t fragments;   /* fragments of the query genome */
  int32_t status;      /* 0 = ok, PG_SKETCH_NO_RESULT */
  int32_t reserved;
} pg_sketch_result;
#define PG_SKETCH_NO_RESULT 1
int pg_sketch_pairs(pg_ctx* ctx, const int32_t* qry_ids, const int32_t* ref_ids, uint64_t n_pairs, int32_t frag_len, int32_t scale,
                    double min_fraction, pg_sketch_result* out);

/* ---- measurement ---------------------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream. */
int pg_profile_enable(pg_ctx* ctx, int on);
/* Restrict event bracketing to the kernels in `kernel_mask` (bit i = PG_K_*) and to every `every_n`-th launch of
 * each, so that measuring inside a timed region costs next to nothing.  Default: all kernels, every launch. */
int pg_profile_config(pg_ctx* ctx, uint32_t kernel_mask, uint32_t every_n);
int pg_profile_reset(pg_ctx* ctx);
#define PG_K_TETRA_COUNT 0
#define PG_K_TETRA_FINALIZE 1
#define PG_K_TETRA_STATS 2
#define PG_K_TETRA_PAIRS 3
/* ANIm stages (pg_anim_pairs); SEED, CLUSTER, EXTLANE and FINISH bracket exactly one kernel per launch, the others a
 * short group of kernels that belong together */
#define PG_K_ANIM_SEED 4     /* anim_seed_kernel: LDS-resident reference groups, streamed query lists */
#define PG_K_ANIM_HIT 5      /* anim_hoff_kernel + anim_hit_scatter_kernel + anim_hit_kernel + anim_scatter_kernel */
#define PG_K_ANIM_CLUSTER 6  /* anim_cluster_wave_kernel (one launch; + anim_cluster_prep_kernel when few units) */
#define PG_K_ANIM_GAPS 7     /* anim_postnuc_gaplist / gaplane<16,32,59> / gapbig kernels (match-to-match alignments) */
#define PG_K_ANIM_EXTLANE 8  /* anim_postnuc_forced_kernel (narrow bands) + anim_postnuc_forced_wide / _huge / _strips kernels */
#define PG_K_ANIM_EXTEND 9   /* anim_postnuc_kernel (the units' walks) */
#define PG_K_ANIM_FINISH 10  /* anim_finish_kernel */
#define PG_K_ANIB_BUCKET 11  /* anib_bucket_kernel: seeds clipped to fragments, counting sort by fragment */
#define PG_K_ANIB_FRAG 12    /* anib_frag_kernel: anchors + X-drop extensions, one wave per (pair, fragment) */
#define PG_K_ANIM_FWD 13     /* anim_postnuc_fwd_kernel: the forward extension off every cluster, ahead of the units' walks */
#define PG_K_ANIM_BWD 14     /* anim_postnuc_rehearse_kernel + anim_postnuc_bwd_kernel: the walks rehearsed, their backward searches run ahead */
#define PG_K_SKETCH_PAIRS 15 /* sketch_pairs_kernel: the sketch mode's containment pass (pg_sketch_pairs) */
#define PG_K__COUNT 16
/* total milliseconds and number of launches of kernel `which` since the last reset (synchronises). */
int pg_profile_get(pg_ctx* ctx, int which, double* total_ms_out, uint64_t* launches_out);
const char* pg_kernel_name(int which);

#ifdef __cplusplus
}
#endif
#endif /* PYANI_GPU_H */
