// pg_internal.h — shared declarations of libpyani_gpu.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "pyani_gpu.h"

// ---- HBM layout of the genome store -------------------------------------------------------------------------
// One "base stream" per genome: its records back to back with ONE dirty separator base between records, then
// dirty padding up to a multiple of PG_SUPER bases (always at least one dirty base at the end).  Streams of all
// genomes are laid out in one arena:
//   codes : 2 bits/base, base s of the arena in bits [2*(s%16), +2) of codes32[s/16]   (A=0 C=1 G=2 T=3, dirty=0)
//   mask  : 1 bit/base,  bit (s%32) of mask32[s/32] = 1 iff the base is one of ACGT ("clean")
// A k-mer window is valid iff all its k mask bits are 1; separators and padding make windows that would cross a
// record or genome boundary invalid for free, so the count kernel needs no record table.
constexpr uint32_t PG_SUPER = 65536;          // bases per super-tile = one block iteration (1024 lanes x 64 bases)
constexpr uint32_t PG_ACC_WORDS = 16 + 64 + 256;  // per-genome accumulator: E2 | E3 | F4 (unsigned long long each)

struct PgGenome {
  uint64_t total_len = 0;    // sum of record lengths (pyani_files.get_sequence_lengths)
  uint32_t n_rec = 0;
  uint64_t stream_len = 0;   // total_len + (n_rec-1) separators
  uint64_t padded_len = 0;   // multiple of PG_SUPER, > stream_len
  uint64_t arena_start = 0;  // base offset in the device arena (multiple of PG_SUPER)
  bool resident = false;
  std::vector<uint32_t> codes, mask;  // host copy until uploaded
  std::array<uint32_t, 256> quirk{};  // tetramers the reference does not count (last window of each strand)
  std::vector<int32_t> rec_start;     // stream position of each record's first base; [n_rec] = stream_len + 1
};

struct PgEventPair {
  hipEvent_t a, b;
  int which;
};

struct pg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::mutex mu;
  std::vector<PgGenome> genomes;
  uint64_t arena_used = 0;  // bases
  // device arena
  uint32_t* d_codes = nullptr;
  uint32_t* d_mask = nullptr;
  uint64_t arena_cap = 0;  // bases (device), excludes the trailing guard super-tile
  uint32_t* d_quirk = nullptr;
  uint32_t quirk_cap = 0;  // genomes
  uint32_t n_resident = 0;
  // batch scratch (device)
  std::vector<int32_t> batch_ids;  // ids of the cached work list
  uint32_t* d_seg_tile0 = nullptr;   // batch: first arena super-tile of each genome
  uint32_t* d_seg_prefix = nullptr;  // batch + 1: cumulative super-tile counts
  uint32_t* d_batch_gid = nullptr;
  uint32_t n_work = 0, batch_cap = 0;
  unsigned long long* d_acc = nullptr;      // batch x PG_ACC_WORDS (zero between passes)
  unsigned long long* d_counts = nullptr;   // batch x (16+64+256): c2 | c3 | c4
  double* d_dev = nullptr;                  // batch x 256 compacted deviations
  double* d_ss = nullptr;                   // batch
  unsigned long long* d_keybits = nullptr;  // batch x 4: bitmap of observed tetramers
  // One result block per pass so that ONE device-to-host copy returns everything:
  //   [ flags: 4 x int32 ] [ z: n x 256 f64 ] [ corr: n x n f64 ] [ present: n x 256 u8 ]
  uint8_t* d_result = nullptr;
  uint8_t* h_result = nullptr;  // pinned
  uint64_t result_cap = 0;      // bytes
  int32_t* d_flags = nullptr;   // [0]=keyset mismatch, [1]=n present keys      (pointers into d_result)
  double* d_z = nullptr;
  double* d_corr = nullptr;
  uint8_t* d_present = nullptr;
  unsigned long long* h_counts = nullptr;  // pinned
  uint32_t h_batch_cap = 0;
  // profiling
  bool profiling = false;
  uint32_t prof_mask = 0xFFFFFFFFu, prof_every = 1;
  uint64_t prof_seen[PG_K__COUNT] = {};
  bool prof_open = false;
  std::vector<PgEventPair> events;
  double prof_ms[PG_K__COUNT] = {};
  uint64_t prof_n[PG_K__COUNT] = {};
  int num_cu = 256;
  void* anim_scratch = nullptr;  // AnimScratch (pg_anim.hip) of worker 0, grows on demand
  // ANIm / fragment-mode calls are split over two host workers, each with its own stream and scratch: while one worker's
  // launch is in a low-occupancy tail (one wave per unit, slowest unit = launch time) the other's kernels fill the GPU
  // result of the latest pg_anim_alignments_batch (caller's pair order)
  std::vector<pg_anim_alignment> aln_store;
  std::vector<uint64_t> aln_indel_off;
  std::vector<int64_t> aln_indels;
  static constexpr int MAX_WORKERS = 4;
  void* anim_scratch_w[MAX_WORKERS] = {nullptr, nullptr, nullptr, nullptr};   // workers 1.. (index 0 unused: worker 0 = anim_scratch)
  hipStream_t stream_w[MAX_WORKERS] = {nullptr, nullptr, nullptr, nullptr};   // workers 1.. (index 0 unused: worker 0 = stream)
  void* anim_lists = nullptr;      // per-genome seed lists, shared by the workers (guarded by anim_mu)
  void* sketch_store = nullptr;    // per-genome k-mer sketches of the sketch mode (pg_sketch.hip), built on first use
  std::mutex anim_mu, err_mu, prof_mu;
  int anib_word_tier = 1;      // fragment mode: search failed fragments again with blastn-sized (11-mer) seeds
  int anim_pn_window_max = 2048;   // forced runs: the widest single-wave window (development: smaller values push runs on to the group kernel)
  int anim_pn_group_max = 8184;    // ... and the widest band the group of four waves takes (development: 0 = everything beyond one wave on the strips)
  int anim_gap_lanes = 1;      // postnuc: small match-to-match gaps on one lane each (0: all gaps on the wave engine; tests compare the two)
  int anim_bwd_ahead = 1;      // backward searches ahead of the units' walks (pga_postnuc.inc); PYANI_ANIM_BWD_AHEAD=0 (development switch): inside them       // PG_EXTENDER_NUCMER (pg_anim_set_extender)
  int anim_workers = 2;
  // pg_anim_pairs_enqueue / _fetch: two lanes, lane L drives worker slots 2 L and 2 L + 1
  struct AnimAsync {
    std::thread th;
    std::vector<int32_t> r, q;
    std::vector<pg_anim_result> out;
    uint64_t ticket = 0;
    int rc = 0;
    bool busy = false;
  } anim_async[2];
  uint64_t anim_next_ticket = 1;
  std::mutex anim_async_mu;
  uint32_t anim_batch_pairs = 131072;         // ordered pairs in flight (split over the two workers: 65536 per launch; every launch pays its slowest unit once)
  uint64_t anim_scratch_matches_held = 0;     // match slots the workers' scratch already holds (counts as available to anim_match_budget)
  uint64_t anim_batch_matches = 512ull << 20; // exact matches in flight (~384 B of scratch each, grown on demand: at most ~136 GB of the 288 GB)
};

int pg_fail(pg_ctx* ctx, int code, const std::string& msg);
#define PG_HIP(ctx, call)                                                                                     \
  do {                                                                                                        \
    hipError_t _e = (call);                                                                                   \
    if (_e != hipSuccess)                                                                                     \
      return pg_fail((ctx), PG_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e));                     \
  } while (0)

// profiling helpers (pg_api.cpp).  Events are recorded on the calling thread's stream: pg_tls_stream when set (the ANIm
// workers), else the context's.
extern thread_local hipStream_t pg_tls_stream;
void pg_prof_begin(pg_ctx* ctx, int which);
void pg_prof_end(pg_ctx* ctx);

// kernels' host launchers (pg_tetra.hip)
int pg_launch_tetra_count(pg_ctx* ctx, uint32_t n_batch);
int pg_launch_tetra_finalize(pg_ctx* ctx, uint32_t n_batch, unsigned long long* d_acc_in);
int pg_launch_tetra_stats(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n);
int pg_launch_tetra_pairs(pg_ctx* ctx, uint32_t n, uint32_t row0, uint32_t nrows, double* d_out, bool mirror);

// ANIm (pg_anim.hip): all pairs sharing one reference genome
int pg_anim_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                       const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                       int apply_filter, pg_anim_result* out);
// frag != nullptr: fragment mode (ANIb) — ref_ids are the subject genomes, qry_ids the fragmented ones; the batch stops after
// the seeding stage and runs the fragment kernels instead of clustering / extension
struct PgFragArgs {
  int32_t fragsize;
  pg_anib_result* out;        // n_pairs results (host)
  pg_anib_row* rows_out;      // optional: the rows of pair 0 (host), at most rows_cap; *n_rows_out = how many there are
  uint32_t rows_cap;
  uint32_t* n_rows_out;
  uint64_t max_slots;         // (pair, fragment) slots per launch
};
int pg_anim_run_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, int filter_1to1, int maxmatch,
                      uint64_t max_matches, pg_anim_result* out_host, uint32_t* n_done,
                      const PgFragArgs* frag = nullptr);   // ref_ids grouped (equal ids adjacent)
void pg_anim_set_worker(pg_ctx* ctx, int worker);   // binds the calling thread to worker 0 .. MAX_WORKERS-1 (stream + scratch) for run_batch
void pg_anim_free_scratch(pg_ctx* ctx);
void pg_anim_release_worker_scratch(pg_ctx* ctx);   // launch scratch of every worker slot (seed lists stay); idle context only
int pg_anim_fetch_alignments(pg_ctx* ctx, int32_t ref_id, int32_t qry_id, uint32_t n, pg_anim_alignment* out);   // after a 1-pair batch
// Where pg_anim_run_batch leaves the alignment records of its pairs when the calling thread has set one (pg_anim_alignments_batch):
// appended pair after pair in the order of the call's arrays; with_indels adds the traceback pass and every alignment's .delta list.
// Development knobs (launch shapes, alternative code paths that must give the same results, counters): environment variables
// honoured ONLY under PYANI_DEV_KNOBS=1 — the test suite sets it — so that a stray variable in a production environment cannot
// change what the library launches.
inline const char* pg_dev_env(const char* name) {
  const char* on = getenv("PYANI_DEV_KNOBS");
  return on && on[0] == '1' ? getenv(name) : nullptr;
}
constexpr uint32_t PG_FRAG_MAX_FRAGS = 15872;      // fragments per query genome in fragment mode (pga_frag.inc: LDS counters)
struct PgAlnSink {
  bool with_indels = false;
  std::vector<pg_anim_alignment> alns;
  std::vector<uint32_t> pair_count;               // alignments of each pair, in arrival order
  std::vector<std::vector<int64_t>> indels;       // parallel to alns (with_indels)
};
int pg_anim_counters_read(pg_ctx* ctx, uint64_t* out /*[64]*/, int reset);
void pg_anim_set_sink(PgAlnSink* sink);           // thread-local; nullptr = none
void pg_anim_drop_lists(pg_ctx* ctx);   // per-genome seed lists: must go when the genome store is cleared
void pg_sketch_drop(pg_ctx* ctx);       // ... and the sketches of the sketch mode (pg_sketch.hip)
int pg_anib_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                       const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen,
                       const double* pident, int64_t* aln_out, int64_t* err_out, double* pid_out);
