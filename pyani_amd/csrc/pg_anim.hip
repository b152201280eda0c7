// pg_anim.hip — gfx950 kernels of the ANIm engine (kernel family 3 of BASELINE.json's north star): replaces the
// `nucmer --mum` + `delta-filter -1` processes pyani shells out to (pyani/anim.py:240-289) and the parse_delta
// reduction (anim.py:292-411) with an in-process pipeline over the 2-bit/1-bit packed genomes already resident in HBM.
//
//   A1 anim_list_kernel      per genome, once: its 16-mers as (k-mer, position) lists partitioned into 2048 hash groups
//                            (every position for the reference role, every 5th strand position for the query role)
//   A2 anim_seed_kernel      one workgroup per (reference, group): the group's reference k-mers become a hash table in
//                            LDS; the same group of every query of that reference streams through it (coalesced,
//                            sequential HBM reads; no random global access except to verify / extend actual hits)
//   A3 anim_cluster_wave_kernel  one WAVE per (pair, strand): MUM filter (packed radix sorts + wave-scan containment
//                            flags), mgaps clustering (lock-free union-find), chain extraction (register / LDS resident)
//   A4x the extension stage  MUMmer's own postnuc / sw_align (pga_postnuc.inc, pga_postnuc_diag.inc; statement pg_nucmer_core.h,
//                            pg_nucmer_diag.h): match-to-match gaps, forward extensions and backward searches as pre-passes
//                            (one wave or one lane per call), the units' sequential cluster walks, the forced re-alignments
//                            as certified bands on diagonal-window engines of 128 ... 8192 diagonals
//   A5 anim_finish_kernel    one wave per pair: 1-to-1 filter (delta-filter -1), parse_delta reduction -> pg_anim_result
//
// The kernels live in include files, in pipeline order: pga_seed.inc (A1/A2), pga_cluster.inc (A3), pga_postnuc.inc +
// pga_postnuc_diag.inc (A4x), pga_finish.inc (A5), pga_frag.inc (fragment mode); this file holds the shared descriptors and
// the host driver.  (The fixed-band "banded64" extender of rounds 1-2 was retired in round 5: it was not exact.)
//
// Every kernel has a scalar statement in pg_anim_core.h that compiles for the host (tools/anim_debug); the two are kept
// in lock-step and compared on the GPU by tests/test_anim_gpu.py.  Limits: genomes up to ~14 Mb (a reference k-mer
// group must fit a 16384-slot LDS table; PG_E_CAPACITY otherwise), chain scores < 2^24.
#include <tuple>
#include "pg_internal.h"
#include "pg_anim_core.h"
#include "pg_nucmer_core.h"
#include "pg_nucmer_diag.h"
#include "pg_anib_core.h"
#include "pg_anim_trace.h"

using namespace pga;

namespace {

constexpr unsigned long long SLOT_EMPTY = ~0ull;
constexpr int FRAG_SEED_MIN = 16;   // fragment mode keeps every sampled 16-mer hit (SEED_K)

struct RefDesc {
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;  // n_rec + 1 entries
  int32_t n_rec;
};

struct UnitDesc {   // one (pair, query strand)
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;
  int32_t n_rec;
  int32_t strand;
  int32_t pair;  // index into the batch's pair list
  int32_t ref;   // index into the batch's reference list
};

// 16 bases starting at stream position p (p + 16 <= len): codes in 32 bits (first base low), clean bits in 16
__device__ __forceinline__ void get16(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t p,
                                      uint32_t& c, uint32_t& m) {
  const uint32_t cw = p >> 4, cs = 2 * (p & 15);
  const uint64_t lo = (uint64_t)codes[cw] | ((uint64_t)codes[cw + 1] << 32);
  c = (uint32_t)(lo >> cs);
  const uint32_t mw = p >> 5, ms = p & 31;
  const uint64_t ml = (uint64_t)mask[mw] | ((uint64_t)mask[mw + 1] << 32);
  m = (uint32_t)(ml >> ms) & 0xFFFFu;
}

// 2-bit codes / clean bits of the n <= 32 stream positions p0 .. p0 + n - 1 of a packed genome (positions outside are not clean)
__device__ __forceinline__ void packed_window(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t len, int64_t p0,
                                              uint64_t& c, uint32_t& ok) {
  c = 0; ok = 0;
  if (p0 >= len || p0 + 32 <= 0) return;
  const int64_t q0 = p0 < 0 ? 0 : p0;                 // first position actually read
  const int sh = (int)(q0 - p0);                      // it lands at window slot sh
  const int32_t w = (int32_t)(q0 >> 4), lastw = (len - 1) >> 4;
  const uint32_t c0 = codes[w], c1 = codes[w + 1 <= lastw ? w + 1 : lastw], c2 = codes[w + 2 <= lastw ? w + 2 : lastw];
  const int cs = 2 * (int)(q0 & 15);
  uint64_t lo = ((uint64_t)c1 << 32) | c0;
  uint64_t win = cs ? ((lo >> cs) | ((uint64_t)c2 << (64 - cs))) : lo;
  const int32_t mw = (int32_t)(q0 >> 5), lastm = (len - 1) >> 5;
  const uint64_t m = ((uint64_t)mask[mw + 1 <= lastm ? mw + 1 : lastm] << 32) | mask[mw];
  uint32_t okw = (uint32_t)(m >> (q0 & 31));
  const int64_t avail = len - q0;                      // positions from q0 that exist
  if (avail < 32) okw &= (avail <= 0 ? 0u : ((1u << avail) - 1u));
  c = sh ? (win << (2 * sh)) : win;
  ok = sh ? (okw << sh) : okw;
}

#include "pga_seed.inc"
#include "pga_cluster.inc"
#include "pga_postnuc.inc"
#include "pga_finish.inc"
#include "pga_frag.inc"

}  // namespace

// ---- host driver ---------------------------------------------------------------------------------------------------
// One batch of ordered pairs (any mix of references): ref_ids[i] = nucmer's reference (pyani's query genome, anim.py:280).
// Scratch lives in the context and only grows.  The per-unit kernels are latency-bound single-thread code, so the
// batch should be as large as memory allows: thousands of units in flight are what fills the GPU.
namespace {
// per-genome seed lists (built once per resident genome and role, dropped by pg_clear_genomes); shared by the two workers of a
// context: built under ctx->anim_mu and complete (stream synchronised) before the lock is released
struct GenomeIdx {
  uint64_t *ref_list = nullptr, *qry_list = nullptr, *qry_list1 = nullptr;   // qry_list1: every position (fragment mode)
  uint32_t *ref_goff = nullptr, *qry_goff = nullptr, *qry_goff1 = nullptr;
  uint32_t ref_max = 0;   // largest reference group (sizes the LDS table)
  uint32_t* word_start = nullptr;   // fragment mode, word tier: 4^11 + 1 bucket offsets of the genome's 11-mers ...
  int32_t* word_pos = nullptr;      // ... and their positions
};
struct AnimLists { std::vector<GenomeIdx> gidx; };
thread_local PgAlnSink* tls_sink = nullptr;      // set by pg_anim_alignments_batch around its run_batch calls
thread_local int tls_worker = 0;   // which of the context's two (stream, scratch) sets the calling thread drives

struct AnimScratch {
  size_t units = 0, pairs = 0, refs = 0, recs = 0, wl = 0, matches = 0;
  uint32_t* list_cnt = nullptr;   // 2 * SEED_GROUPS counters used by this worker's list builds
  SeedRef* srefs_d = nullptr;
  SeedQry* sqry_d = nullptr;
  SeedSlice* slice_d = nullptr;   // [SEED_GROUPS][pairs of the batch]
  size_t slice_pairs = 0;
  int32_t* recs_d = nullptr;
  RefDesc* refs_d = nullptr;
  UnitDesc* units_d = nullptr;
  uint32_t *mem_count = nullptr, *moff = nullptr, *choff_d = nullptr;
  int32_t *nch = nullptr, *status = nullptr;
  pg_anim_result* out = nullptr;
  // per-match arrays (sliced by moff)
  Match *mem = nullptr, *cm = nullptr;
  int32_t *iscratch = nullptr, *order = nullptr;
  Chain* chains = nullptr;
  FinishScratch S{};
  uint2* wl_d = nullptr;
  Match* seedbuf = nullptr;   // batch-wide append buffer of the seed pass
  size_t seed_cap = 0;
  uint32_t* seed_total = nullptr;   // [0] matches appended, [1] hits recorded
  Match* hits_d = nullptr;          // hits recorded by the probe kernel for anim_hit_kernel
  Match* hits_sorted = nullptr;     // the same, dealt into per-unit slices (hoff)
  uint32_t *hit_count = nullptr, *hoff = nullptr, *hit_cursor = nullptr;   // per unit
  size_t hit_cap = 0;
  int32_t* mirror_d = nullptr;      // per pair: the partner pair (roles swapped) that receives this pair's matches transposed, or -1
  BigUnit* big_d = nullptr;         // cluster stage: units whose chains are extracted by ranges, the range work items, their counts
  uint2* ranges_d = nullptr;
  RangeOut* range_out = nullptr;
  size_t big_cap = 0, range_cap = 0;
  bool lds_attr_set = false;        // anim_seed_kernel's dynamic-LDS limit has been raised on this context's device
  // A4x, the postnuc extension stage (pga_postnuc.inc)
  pgn::PnAln* pn = nullptr;         // per-unit alignment lists, sliced by moff like the per-match arrays
  uint8_t* pn_fused = nullptr;      // per chain: already extended / fused / shadowed
  int32_t* pn_n = nullptr;          // per unit: alignments (< 0: capacity)
  uint32_t* pn_cursor = nullptr;    // unit hand-out counter of the persistent waves
  uint32_t* pn_gscratch = nullptr;  // [waves][PN_GLOBAL_WORDS] anti-diagonals too wide for LDS
  uint32_t* pn_wide = nullptr;      // request slots handed from the narrow forced kernel to the wide one
  PnForcedReq* pn_reqs = nullptr;   // the launch's deferred forced runs (at most one per alignment started: <= chains)
  pgn::PnGap* pn_gaps = nullptr;    // match-to-match alignments by match slot
  pgn::PnFwd* pn_fwd = nullptr;     // forward extensions by cluster (moff-relative position in the unit's order)
  pgn::PnBwd* pn_bwd = nullptr;      // backward searches run ahead of the walks, by cluster (as pn_fwd)
  pgn::PnTurn* pn_tlog = nullptr;    // the walks' turn logs (pgn::PnPairSync), sliced by moff: at most one turn per cluster
  int32_t* pn_born = nullptr;        // per alignment: the key of the turn that pushed it
  uint32_t* pn_porder = nullptr;     // pairs by descending cluster count of their larger strand
  pgn::PnPiece* pn_pieces = nullptr; // traceback runs: the walks' pieces (pn_piece_base)
  uint32_t* pn_npieces = nullptr;    // per unit
  size_t pn_piece_cap = 0, pn_npieces_cap = 0;
  uint8_t* tr_arena = nullptr;       // traceback pass: the jobs' slabs
  uint32_t* tr_out = nullptr;        // ... their paths (run-length coded)
  PnTraceJob* tr_jobs = nullptr;
  unsigned long long *tr_off = nullptr, *tr_cursor = nullptr;
  int32_t* tr_cnt = nullptr;
  size_t tr_arena_cap = 0, tr_out_cap = 0, tr_jobs_cap = 0;
  PnGapTask* pn_tasks = nullptr;    // [3 size classes][pn_cap] small gaps for the lane kernel
  uint32_t* pn_order = nullptr;     // units by descending cluster count
  size_t pn_cap = 0, pn_units = 0, pn_waves = 0, pn_req_cap = 0, pn_req_n = 0;   // pn_req_n: slots of pn_reqs the latest launch used (its req_cap)
  // fragment mode (ANIb)
  int32_t* fr_tables = nullptr;     // frag_pos | frag_len | rec_frag0 of every distinct query genome of the batch
  FragPair* fr_pairs = nullptr;
  uint32_t *fr_slot_pair = nullptr, *fr_off = nullptr, *fr_nrows = nullptr;
  uint64_t* fr_ebase = nullptr;
  FragSeed* fr_entries = nullptr;
  FragRow* fr_rows = nullptr;
  pg_anib_result* fr_out = nullptr;
  uint32_t *fr_list = nullptr, *fr_nlist = nullptr, *fr_wtmp = nullptr;   // word tier: slots to search again, their number, scan scratch
  WordIdx* fr_widx = nullptr;
  size_t fr_list_cap = 0, fr_widx_cap = 0;
  size_t fr_tables_cap = 0, fr_pairs_cap = 0, fr_slots_cap = 0, fr_off_cap = 0, fr_units_cap = 0, fr_entries_cap = 0;
};

template <typename T>
int regrow(pg_ctx* ctx, T*& p, size_t n) {
  if (p) PG_HIP(ctx, hipFree(p));
  p = nullptr;
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)));
  return PG_OK;
}
}  // namespace

static AnimScratch* anim_scratch(pg_ctx* ctx) {
  void*& slot = tls_worker ? ctx->anim_scratch_w[tls_worker] : ctx->anim_scratch;
  if (!slot) slot = new AnimScratch();
  return static_cast<AnimScratch*>(slot);
}
static AnimLists* anim_lists(pg_ctx* ctx) {   // (callers hold ctx->anim_mu)
  if (!ctx->anim_lists) ctx->anim_lists = new AnimLists();
  return static_cast<AnimLists*>(ctx->anim_lists);
}
static hipStream_t cur_stream(pg_ctx* ctx) { return tls_worker ? ctx->stream_w[tls_worker] : ctx->stream; }
void pg_anim_set_worker(pg_ctx* ctx, int worker) {
  tls_worker = worker > 0 && worker < pg_ctx::MAX_WORKERS ? worker : 0;
  pg_tls_stream = cur_stream(ctx);
}

void pg_anim_set_sink(PgAlnSink* sink) { tls_sink = sink; }
int pg_anim_counters_read(pg_ctx* ctx, uint64_t* out, int reset) {
  PG_HIP(ctx, hipDeviceSynchronize());
  unsigned long long a[32], b[32], z[32] = {0};
  PG_HIP(ctx, hipMemcpyFromSymbol(a, HIP_SYMBOL(g_pn_stats), sizeof(a)));
  PG_HIP(ctx, hipMemcpyFromSymbol(b, HIP_SYMBOL(g_pn_kstats), sizeof(b)));
  for (int i = 0; i < 32; ++i) { out[i] = a[i]; out[32 + i] = b[i]; }
  if (reset) {
    PG_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pn_stats), z, sizeof(z)));
    PG_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pn_kstats), z, sizeof(z)));
  }
  return PG_OK;
}

void pg_anim_drop_lists(pg_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->anim_mu);
  AnimLists* A = static_cast<AnimLists*>(ctx->anim_lists);
  if (!A) return;
  for (auto& g : A->gidx) {
    void* ptrs[] = {g.ref_list, g.qry_list, g.ref_goff, g.qry_goff, g.qry_list1, g.qry_goff1, g.word_start, g.word_pos};
    for (void* p : ptrs) if (p) (void)hipFree(p);
  }
  A->gidx.clear();
}

// Build the seed lists the batch needs and does not have yet: reference role for `ref_genomes`, query role for `qry_genomes`.
static int anim_ensure_lists(pg_ctx* ctx, AnimScratch* A, const std::vector<int32_t>& ref_genomes, const std::vector<int32_t>& qry_genomes,
                             int qstep) {
  std::lock_guard<std::mutex> lk(ctx->anim_mu);   // one worker builds at a time; a list is complete before anyone else sees it
  AnimLists* LS = anim_lists(ctx);
  if (LS->gidx.size() < ctx->genomes.size()) LS->gidx.resize(ctx->genomes.size());
  // A list counts as built when its pointer is set, and it outlives this call (shared by the workers): whatever this call
  // allocated is taken back if anything after the allocation fails, so that no later call seeds from a half-built list.
  std::vector<std::pair<uint64_t**, uint32_t**>> mine;
  const int rc_all = [&]() -> int {
  int rc;
  bool built = false;
  if (!A->list_cnt && (rc = regrow(ctx, A->list_cnt, (size_t)2 * SEED_GROUPS))) return rc;
  std::vector<int32_t> fresh_refs;
  for (int role = 0; role < 2; ++role) {
    for (int32_t gid : role ? qry_genomes : ref_genomes) {
      GenomeIdx& X = LS->gidx[gid];
      uint64_t*& qlist = qstep == 1 ? X.qry_list1 : X.qry_list;
      uint32_t*& qgoff = qstep == 1 ? X.qry_goff1 : X.qry_goff;
      if (role ? qlist != nullptr : X.ref_list != nullptr) continue;
      const PgGenome& G = ctx->genomes[gid];
      const int32_t len = (int32_t)G.stream_len;
      const uint32_t n_sub = role ? 2 * SEED_GROUPS : SEED_GROUPS;
      const size_t bound = role ? 2 * ((size_t)len / qstep + 1) : (size_t)len + 1;
      uint64_t*& list = role ? qlist : X.ref_list;
      uint32_t*& goff = role ? qgoff : X.ref_goff;
      mine.emplace_back(&list, &goff);
      if ((rc = regrow(ctx, list, bound))) return rc;
      if ((rc = regrow(ctx, goff, (size_t)n_sub + 2))) return rc;
      const uint32_t* codes = ctx->d_codes + G.arena_start / 16;
      const uint32_t* mask = ctx->d_mask + G.arena_start / 32;
      const int32_t n_idx = role ? len / qstep + 1 : len;
      const dim3 grid((uint32_t)(n_idx + LIST_CHUNK - 1) / LIST_CHUNK, role ? 2 : 1);
      PG_HIP(ctx, hipMemsetAsync(A->list_cnt, 0, (size_t)n_sub * 4, cur_stream(ctx)));
      if (grid.x)   // (an empty genome still gets its all-zero offset table from the scan)
        hipLaunchKernelGGL(anim_list_kernel, grid, dim3(LIST_BLOCK), 0, cur_stream(ctx), codes, mask, len, role, A->list_cnt,
                           (const uint32_t*)nullptr, (uint64_t*)nullptr, 0, qstep);
      hipLaunchKernelGGL(anim_list_scan_kernel, dim3(1), dim3(64), 0, cur_stream(ctx), A->list_cnt, goff, n_sub);
      if (grid.x)
        hipLaunchKernelGGL(anim_list_kernel, grid, dim3(LIST_BLOCK), 0, cur_stream(ctx), codes, mask, len, role, A->list_cnt,
                           (const uint32_t*)goff, list, 1, qstep);
      if (!role) fresh_refs.push_back(gid);
      built = true;
    }
  }
  PG_HIP(ctx, hipGetLastError());
  if (built) PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
  for (int32_t gid : fresh_refs)
    PG_HIP(ctx, hipMemcpy(&LS->gidx[gid].ref_max, LS->gidx[gid].ref_goff + SEED_GROUPS + 1, 4, hipMemcpyDeviceToHost));
  return PG_OK;
  }();
  if (rc_all != PG_OK) {
    (void)hipStreamSynchronize(cur_stream(ctx));      // nothing of this call may still be writing into them
    for (auto& pr : mine) {
      if (*pr.first) { (void)hipFree(*pr.first); *pr.first = nullptr; }
      if (*pr.second) { (void)hipFree(*pr.second); *pr.second = nullptr; }
    }
  }
  return rc_all;
}

static void anim_free_one(pg_ctx* ctx, void*& slot);
// the workers' launch scratch only (the per-genome seed lists stay): the context must be idle
void pg_anim_release_worker_scratch(pg_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) (void)hipStreamSynchronize(ctx->stream_w[w]);
  anim_free_one(ctx, ctx->anim_scratch);
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) anim_free_one(ctx, ctx->anim_scratch_w[w]);
  ctx->anim_scratch_matches_held = 0;
}
void pg_anim_free_scratch(pg_ctx* ctx) {
  pg_anim_drop_lists(ctx);
  delete static_cast<AnimLists*>(ctx->anim_lists);
  ctx->anim_lists = nullptr;
  anim_free_one(ctx, ctx->anim_scratch);
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) anim_free_one(ctx, ctx->anim_scratch_w[w]);
  ctx->anim_scratch_matches_held = 0;
}
static void anim_free_one(pg_ctx* ctx, void*& slot) {
  AnimScratch* A = static_cast<AnimScratch*>(slot);
  if (!A) return;
  void* ptrs[] = {A->mirror_d, A->big_d, A->ranges_d, A->range_out, A->hits_sorted, A->hit_count, A->hoff, A->hit_cursor, A->hits_d, A->slice_d, A->choff_d, A->list_cnt, A->srefs_d, A->sqry_d, A->recs_d, A->refs_d, A->units_d, A->mem_count, A->moff, A->nch, A->status, A->out, A->mem, A->cm,
                  A->iscratch, A->order, A->chains, A->S.alns, A->S.a_rrec,
                  A->S.a_qrec, A->S.idx, A->S.from, A->S.sc, A->wl_d, A->seedbuf, A->seed_total, A->fr_tables, A->fr_pairs,
                  A->fr_slot_pair, A->fr_off, A->fr_nrows, A->fr_ebase, A->fr_entries, A->fr_rows, A->fr_out, A->fr_list, A->fr_nlist, A->fr_wtmp, A->fr_widx,
                  A->pn, A->pn_fused, A->pn_n, A->pn_cursor, A->pn_gscratch, A->pn_reqs, A->pn_wide, A->pn_gaps, A->pn_fwd, A->pn_bwd, A->pn_tlog, A->pn_born, A->pn_porder, A->pn_tasks, A->pn_order, A->pn_pieces, A->pn_npieces, A->tr_arena, A->tr_out, A->tr_jobs, A->tr_off, A->tr_cursor, A->tr_cnt};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete A;
  slot = nullptr;
}

// One batch of ordered pairs (ref_ids grouped).  The seed pass appends every unit's matches to one buffer and counts them
// per (pair, strand) unit; a scatter then gives every per-match array exactly the slice it needs, which is what lets
// thousands of units be in flight at once within the HBM budget.
// If the batch needs more than max_matches, only its first n_done pairs are processed (the caller continues from there).
static_assert(sizeof(FragRow) == sizeof(pg_anib_row), "FragRow is pg_anib_row");
static int anib_frag_stage(pg_ctx* ctx, AnimScratch* A, const int32_t* qry_ids, uint32_t n_pairs, const std::vector<uint32_t>& cnt,
                           const PgFragArgs& F, const std::vector<int32_t>& ref_list, const std::vector<uint32_t>& ref_of_pair);

// The alignment records of a finished batch -> the caller's sink; with_indels: the traceback pass (anim_trace_kernel: one
// search / forced piece of the walks per thread, the scalar engine with its backpointer store) and the .delta lists
// (pg_anim_trace.h).  Host work here is list management: sizing the jobs' slabs from the wave engine's own bookkeeping,
// batching them into the arena, stitching the pieces' paths.
static int anim_collect(pg_ctx* ctx, AnimScratch* A, const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, const pg_anim_result* res,
                        const std::vector<uint32_t>& choff, PgAlnSink& sink) {
  const uint32_t n_units = 2 * n_pairs;
  int rc;
  std::vector<uint32_t> moff((size_t)n_units + 1);
  PG_HIP(ctx, hipMemcpy(moff.data(), A->moff, moff.size() * 4, hipMemcpyDeviceToHost));
  const size_t total = moff[n_units];
  std::vector<Aln> al(total ? total : 1);
  std::vector<int32_t> rr(total ? total : 1), qr(total ? total : 1);
  if (total) {
    PG_HIP(ctx, hipMemcpy(al.data(), A->S.alns, total * sizeof(Aln), hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(rr.data(), A->S.a_rrec, total * 4, hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(qr.data(), A->S.a_qrec, total * 4, hipMemcpyDeviceToHost));
  }
  const size_t first_aln = sink.alns.size();
  std::vector<size_t> pair_first(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (res[p].status == PG_E_CAPACITY) return pg_fail(ctx, PG_E_CAPACITY, "anim: work buffers overflowed for a pair of the batch");
    const PgGenome& G = ctx->genomes[ref_ids[p]];
    const PgGenome& H = ctx->genomes[qry_ids[p]];
    const uint32_t n = (uint32_t)res[p].reserved;
    pair_first[p] = sink.alns.size();
    for (uint32_t i = 0; i < n; ++i) {
      const Aln& a = al[moff[2 * p] + i];            // forward stream coordinates, half-open
      const int32_t r_ = rr[moff[2 * p] + i], q_ = qr[moff[2 * p] + i], ro = G.rec_start[r_], qo = H.rec_start[q_];
      pg_anim_alignment x;
      x.ref_rec = r_; x.qry_rec = q_;
      x.rs = a.rs - ro + 1; x.re = a.re - ro;
      x.qs = a.strand ? a.qe - qo : a.qs - qo + 1;
      x.qe = a.strand ? a.qs - qo + 1 : a.qe - qo;
      x.errors = a.errors; x.kept = a.keep;
      sink.alns.push_back(x);
    }
    sink.pair_count.push_back(n);
  }
  if (!sink.with_indels) return PG_OK;
  sink.indels.resize(sink.alns.size());
  const size_t n_wl = choff[n_units];
  if (!total || !n_wl) return PG_OK;      // no clusters at all: no alignments, nothing to trace
  // ---- the walks' pieces
  std::vector<pgn::PnAln> pn(total);
  std::vector<int32_t> pn_n(n_units);
  std::vector<uint32_t> npieces(n_units);
  const size_t piece_total = pn_piece_base(total, (uint32_t)n_wl, n_units);
  std::vector<pgn::PnPiece> pieces(piece_total ? piece_total : 1);
  PG_HIP(ctx, hipMemcpy(pn.data(), A->pn, total * sizeof(pgn::PnAln), hipMemcpyDeviceToHost));
  PG_HIP(ctx, hipMemcpy(pn_n.data(), A->pn_n, (size_t)n_units * 4, hipMemcpyDeviceToHost));
  PG_HIP(ctx, hipMemcpy(npieces.data(), A->pn_npieces, (size_t)n_units * 4, hipMemcpyDeviceToHost));
  PG_HIP(ctx, hipMemcpy(pieces.data(), A->pn_pieces, piece_total * sizeof(pgn::PnPiece), hipMemcpyDeviceToHost));
  const size_t req_cap = A->pn_req_n;      // as the launch laid the two request lists out
  std::vector<PnForcedReq> reqs(req_cap);
  PG_HIP(ctx, hipMemcpy(reqs.data(), A->pn_reqs, req_cap * sizeof(PnForcedReq), hipMemcpyDeviceToHost));
  // ---- one job per search / forced piece
  struct JobRef { uint32_t unit; uint32_t piece; uint64_t bytes; };      // piece: index in the unit's list
  std::vector<PnTraceJob> jobs;
  std::vector<JobRef> refs_;
  for (uint32_t u = 0; u < n_units; ++u) {
    const size_t pb = pn_piece_base(moff[u], choff[u], u);
    if (pn_n[u] < 0 || npieces[u] > pn_piece_cap((size_t)moff[u + 1] - moff[u], choff[u + 1] - choff[u]))
      return pg_fail(ctx, PG_E_CAPACITY, "anim traceback: a unit's piece list overflowed");
    for (uint32_t k = 0; k < npieces[u]; ++k) {
      const pgn::PnPiece& P = pieces[pb + k];
      if (P.kind == pgn::PIECE_MATCH || P.kind == pgn::PIECE_VISIT) continue;
      PnTraceJob J{};
      J.unit = u; J.m_o = P.m_o; J.A0 = P.A0; J.B0 = P.B0; J.A1 = P.A1; J.B1 = P.B1;
      uint64_t cells = P.cells;
      uint32_t wmax = P.wmax;
      if (P.kind == pgn::PIECE_FORCED) {
        J.tA = P.A1; J.tB = P.B1;
        if (P.aux >= req_cap) return pg_fail(ctx, PG_E_INTERNAL, "anim traceback: forced piece without its request");
        J.band_w = reqs[P.aux].w;
        const int32_t N = P.A1 - P.A0 + 1, M_ = P.B1 - P.B0 + 1;
        cells = 0; wmax = 0;
        for (int32_t d = 1; d <= N + M_; ++d) {      // the cells of the certified band, as the engine clips them
          int32_t lo = d - N > 0 ? d - N : 0, hi = d < M_ ? d : M_;
          if (J.band_w >= 0) pgn::forced_band_clip(d, N, M_, J.band_w, lo, hi);
          if (hi >= lo) { cells += (uint64_t)(hi - lo + 1); if ((uint32_t)(hi - lo + 1) > wmax) wmax = (uint32_t)(hi - lo + 1); }
        }
      } else {
        J.tA = P.tA; J.tB = P.tB; J.band_w = -1;
        if (P.cells == 0xFFFFFFFFu) return pg_fail(ctx, PG_E_CAPACITY, "anim traceback: a search outgrew the register engine");
      }
      const int32_t N = J.tA - J.A0 + 1, M_ = J.tB - J.B0 + 1;
      J.cap = wmax + 2; J.dcap = (uint32_t)(N + M_ + 4); J.rle_cap = (uint32_t)(N + M_ + 4); J.bp_cap = cells;
      const uint64_t bytes = (uint64_t)J.cap * 3 * sizeof(pgn::Cell) + (uint64_t)J.dcap * 8 + (uint64_t)J.rle_cap * 4 + ((cells + 15) & ~15ull) + 16;
      jobs.push_back(J);
      refs_.push_back(JobRef{u, k, (bytes + 15) & ~15ull});
    }
  }
  // ---- batches that fit the arena (largest jobs first: threads of a wave get pieces of similar size)
  std::vector<uint32_t> order(jobs.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return refs_[a].bytes > refs_[b].bytes; });
  std::vector<std::vector<uint32_t>> rle(jobs.size());
  const uint64_t ARENA = 6ull << 30, OUT_MAX = 192ull << 20;      // bytes of slabs / path entries per launch
  if (!A->tr_cursor && (rc = regrow(ctx, A->tr_cursor, 2))) return rc;
  for (size_t b0 = 0; b0 < order.size();) {
    uint64_t bytes = 0, outs = 0;
    size_t b1 = b0;
    while (b1 < order.size() && (b1 == b0 || (bytes + refs_[order[b1]].bytes <= ARENA && outs + jobs[order[b1]].rle_cap <= OUT_MAX))) {
      bytes += refs_[order[b1]].bytes; outs += jobs[order[b1]].rle_cap; ++b1;
    }
    const size_t nb = b1 - b0;
    std::vector<PnTraceJob> hj(nb);
    uint64_t at = 0;
    for (size_t k = 0; k < nb; ++k) { hj[k] = jobs[order[b0 + k]]; hj[k].slab = at; at += refs_[order[b0 + k]].bytes; }
    if (bytes > A->tr_arena_cap) { if ((rc = regrow(ctx, A->tr_arena, (size_t)bytes))) return rc; A->tr_arena_cap = (size_t)bytes; }
    if (outs > A->tr_out_cap) { if ((rc = regrow(ctx, A->tr_out, (size_t)outs))) return rc; A->tr_out_cap = (size_t)outs; }
    if (nb > A->tr_jobs_cap) {
      if ((rc = regrow(ctx, A->tr_jobs, nb))) return rc;
      if ((rc = regrow(ctx, A->tr_off, nb))) return rc;
      if ((rc = regrow(ctx, A->tr_cnt, nb))) return rc;
      A->tr_jobs_cap = nb;
    }
    PG_HIP(ctx, hipMemcpyAsync(A->tr_jobs, hj.data(), nb * sizeof(PnTraceJob), hipMemcpyHostToDevice, cur_stream(ctx)));
    PG_HIP(ctx, hipMemsetAsync(A->tr_cursor, 0, 8, cur_stream(ctx)));
    hipLaunchKernelGGL(anim_trace_kernel, dim3((uint32_t)((nb + 63) / 64)), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->tr_jobs, (uint32_t)nb,
                       A->tr_arena, A->tr_out, (unsigned long long)outs, A->tr_cursor, A->tr_off, A->tr_cnt);
    PG_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> off(nb);
    std::vector<int32_t> cnt(nb);
    unsigned long long used = 0;
    PG_HIP(ctx, hipMemcpyAsync(off.data(), A->tr_off, nb * 8, hipMemcpyDeviceToHost, cur_stream(ctx)));
    PG_HIP(ctx, hipMemcpyAsync(cnt.data(), A->tr_cnt, nb * 4, hipMemcpyDeviceToHost, cur_stream(ctx)));
    PG_HIP(ctx, hipMemcpyAsync(&used, A->tr_cursor, 8, hipMemcpyDeviceToHost, cur_stream(ctx)));
    PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    std::vector<uint32_t> out((size_t)used ? (size_t)used : 1);
    if (used) PG_HIP(ctx, hipMemcpy(out.data(), A->tr_out, (size_t)used * 4, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < nb; ++k) {
      if (cnt[k] < 0) return pg_fail(ctx, PG_E_INTERNAL, "anim traceback: a piece did not repeat in the scalar engine");
      rle[order[b0 + k]].assign(out.begin() + (size_t)off[k], out.begin() + (size_t)off[k] + (size_t)cnt[k]);
    }
    b0 = b1;
  }
  // ---- stitch: per unit, the paths of its alignments; then the pair's records in MUMmer's print order (PIECE_VISIT)
  size_t job_at = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const size_t n_rec = (size_t)pn_n[2 * p] + (size_t)pn_n[2 * p + 1];
    if (n_rec != sink.pair_count[sink.pair_count.size() - n_pairs + p]) return pg_fail(ctx, PG_E_INTERNAL, "anim traceback: record counts differ");
    std::vector<std::vector<int64_t>> lists(n_rec);
    std::vector<std::tuple<int32_t, int32_t, int32_t>> key(n_rec);      // (visit, strand, index in the unit)
    for (uint32_t u = 2 * p; u < 2 * p + 2; ++u) {
      const size_t pb = pn_piece_base(moff[u], choff[u], u);
      std::vector<int32_t> job_of(npieces[u], -1);
      for (uint32_t k = 0; k < npieces[u]; ++k)
        if (pieces[pb + k].kind == pgn::PIECE_SEARCH || pieces[pb + k].kind == pgn::PIECE_FORCED) job_of[k] = (int32_t)job_at++;
      std::vector<std::vector<int64_t>> deltas;
      std::vector<int32_t> visit;
      std::string why;
      static const uint32_t none = 0;
      if (!pgt::unit_deltas(pieces.data() + pb, (int32_t)npieces[u], pn.data() + moff[u], pn_n[u],
                            [&](int32_t k, int32_t& c) -> const uint32_t* { const auto& v = rle[(size_t)job_of[k]]; c = (int32_t)v.size(); return v.empty() ? &none : v.data(); },
                            deltas, visit, &why))
        return pg_fail(ctx, PG_E_INTERNAL, "anim traceback: " + why);
      const size_t base = (u & 1) ? (size_t)pn_n[u - 1] : 0;
      for (int32_t i = 0; i < pn_n[u]; ++i) { lists[base + (size_t)i] = std::move(deltas[(size_t)i]); key[base + (size_t)i] = std::make_tuple(visit[(size_t)i], (int32_t)(u & 1), i); }
    }
    std::vector<size_t> perm(n_rec);
    for (size_t i = 0; i < n_rec; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    std::vector<pg_anim_alignment> recs(n_rec);
    for (size_t i = 0; i < n_rec; ++i) recs[i] = sink.alns[pair_first[p] + perm[i]];
    for (size_t i = 0; i < n_rec; ++i) { sink.alns[pair_first[p] + i] = recs[i]; sink.indels[pair_first[p] + i] = std::move(lists[perm[i]]); }
  }
  (void)first_aln;
  return PG_OK;
}

int pg_anim_run_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, int filter_1to1, int maxmatch,
                      uint64_t max_matches, pg_anim_result* out_host, uint32_t* n_done, const PgFragArgs* frag) {
  AnimScratch* A = anim_scratch(ctx);
  int rc;
  (void)hipGetLastError();   // launch checks below must only see this batch's errors
  if (frag) {   // fragment mode: a launch holds at most max_slots (pair, fragment) slots
    uint64_t slots = 0;
    uint32_t fit = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
      const PgGenome& Q = ctx->genomes[qry_ids[p]];
      uint64_t nf = 0;
      for (uint32_t r = 0; r < Q.n_rec; ++r) nf += ((uint64_t)(Q.rec_start[r + 1] - 1 - Q.rec_start[r]) + frag->fragsize - 1) / frag->fragsize;
      if (p > 0 && slots + nf > frag->max_slots) break;
      slots += nf;
      fit = p + 1;
    }
    n_pairs = fit;
  }
  uint32_t n_units = 2 * n_pairs;
  const int qstep = frag ? FRAG_QSTEP : SEED_STEP;   // query-strand sampling of the seed lists
  std::vector<int32_t> ref_list;
  std::vector<uint32_t> ref_of_pair(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (ref_list.empty() || ref_list.back() != ref_ids[p]) ref_list.push_back(ref_ids[p]);
    ref_of_pair[p] = (uint32_t)ref_list.size() - 1;
  }
  const uint32_t n_refs = (uint32_t)ref_list.size();
  std::vector<RefDesc> refs(n_refs);
  std::vector<int32_t> recs;
  std::vector<uint32_t> ref_rec_off(n_refs), qry_rec_off(n_pairs);
  int32_t max_rlen = 0, max_qlen = 0;
  for (uint32_t r = 0; r < n_refs; ++r) {
    const PgGenome& G = ctx->genomes[ref_list[r]];
    refs[r].codes = ctx->d_codes + G.arena_start / 16;
    refs[r].mask = ctx->d_mask + G.arena_start / 32;
    refs[r].len = (int32_t)G.stream_len;
    refs[r].n_rec = (int32_t)G.n_rec;
    ref_rec_off[r] = (uint32_t)recs.size();
    recs.insert(recs.end(), G.rec_start.begin(), G.rec_start.end());
    if ((int32_t)G.stream_len > max_rlen) max_rlen = (int32_t)G.stream_len;
  }
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    qry_rec_off[p] = (uint32_t)recs.size();
    recs.insert(recs.end(), Q.rec_start.begin(), Q.rec_start.end());
    if ((int32_t)Q.stream_len > max_qlen) max_qlen = (int32_t)Q.stream_len;
  }
  if (recs.size() > A->recs) { if ((rc = regrow(ctx, A->recs_d, recs.size()))) return rc; A->recs = recs.size(); }
  if (n_refs > A->refs) {
    if ((rc = regrow(ctx, A->refs_d, n_refs))) return rc;
    if ((rc = regrow(ctx, A->srefs_d, n_refs))) return rc;
    A->refs = n_refs;
  }
  if (n_units > A->units) {
    if ((rc = regrow(ctx, A->units_d, n_units))) return rc;
    if ((rc = regrow(ctx, A->mem_count, n_units))) return rc;
    if ((rc = regrow(ctx, A->moff, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->choff_d, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->hit_count, n_units))) return rc;
    if ((rc = regrow(ctx, A->hoff, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->hit_cursor, n_units))) return rc;
    if ((rc = regrow(ctx, A->nch, n_units))) return rc;
    A->units = n_units;
  }
  if (n_pairs > A->pairs) {
    if ((rc = regrow(ctx, A->status, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->sqry_d, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->out, n_pairs))) return rc;
    A->pairs = n_pairs;
  }
  for (uint32_t r = 0; r < n_refs; ++r) refs[r].rec_start = A->recs_d + ref_rec_off[r];
  std::vector<UnitDesc> units(n_units);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    for (int s = 0; s < 2; ++s) {
      UnitDesc& U = units[2 * p + s];
      U.codes = ctx->d_codes + Q.arena_start / 16;
      U.mask = ctx->d_mask + Q.arena_start / 32;
      U.len = (int32_t)Q.stream_len;
      U.rec_start = A->recs_d + qry_rec_off[p];
      U.n_rec = (int32_t)Q.n_rec;
      U.strand = s;
      U.pair = (int32_t)p;
      U.ref = (int32_t)ref_of_pair[p];
    }
  }
  PG_HIP(ctx, hipMemcpyAsync(A->recs_d, recs.data(), recs.size() * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
  PG_HIP(ctx, hipMemcpyAsync(A->refs_d, refs.data(), n_refs * sizeof(RefDesc), hipMemcpyHostToDevice, cur_stream(ctx)));
  PG_HIP(ctx, hipMemcpyAsync(A->units_d, units.data(), n_units * sizeof(UnitDesc), hipMemcpyHostToDevice, cur_stream(ctx)));
  // seeding: LDS-resident reference groups, streamed query groups; one pass appends (unit, match) records and the
  // per-unit counts it leaves are exact even if the buffer overflowed.
  // Roles: when the launch holds a pair in BOTH directions, (A, B) and (B, A), only one of them is seeded — the maximal exact
  // matches of the two are the same set, and anim_hit_kernel appends each match a second time, transposed, for the partner
  // (`mirror`).  The seeded direction is the one whose reference has more pairs in the launch (longer query streams per LDS
  // table); equal counts: decided by the ids' parity, so that every genome is the table for half of its partners.
  std::vector<int32_t> mirror(n_pairs);
  std::vector<uint8_t> seeded(n_pairs);
  std::vector<SeedRef> srefs;
  std::vector<SeedQry> sqry(n_pairs);
  std::vector<GenomeIdx> LSv;
  uint32_t slots = 256, n_srefs = 0;
  if (n_pairs > A->slice_pairs) {
    if ((rc = regrow(ctx, A->slice_d, (size_t)n_pairs * SEED_GROUPS))) return rc;
    if ((rc = regrow(ctx, A->mirror_d, n_pairs))) return rc;
    A->slice_pairs = n_pairs;
  }
  const uint32_t slice_stride = n_pairs;   // the table is laid out for the whole batch even if only a prefix is seeded again
  const bool use_mirror = !frag && !pg_dev_env("PYANI_ANIM_NO_MIRROR");
  auto prepare = [&](uint32_t limit) -> int {   // roles, seed lists and descriptors for the pairs [0, limit)
    std::fill(mirror.begin(), mirror.end(), -1);
    std::fill(seeded.begin(), seeded.end(), (uint8_t)0);
    std::fill(seeded.begin(), seeded.begin() + limit, (uint8_t)1);
    if (use_mirror && limit > 1) {
      std::vector<uint32_t> deg(ctx->genomes.size(), 0);
      for (uint32_t p = 0; p < limit; ++p) ++deg[ref_ids[p]];
      std::vector<uint32_t> idx(limit);
      for (uint32_t p = 0; p < limit; ++p) idx[p] = p;
      auto key = [&](uint32_t p) {   // unordered pair, then direction, then position: partners end up next to each other
        const uint32_t a = (uint32_t)ref_ids[p], b = (uint32_t)qry_ids[p];
        return std::make_tuple(a < b ? a : b, a < b ? b : a, a < b ? 0 : 1, p);
      };
      std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return key(x) < key(y); });
      for (uint32_t i = 0; i < limit;) {
        uint32_t j = i;
        while (j < limit && std::get<0>(key(idx[j])) == std::get<0>(key(idx[i])) && std::get<1>(key(idx[j])) == std::get<1>(key(idx[i]))) ++j;
        uint32_t m = i;   // [i, m): direction min -> max, [m, j): the other direction (a pair listed twice pairs up once)
        while (m < j && std::get<2>(key(idx[m])) == 0) ++m;
        if (ref_ids[idx[i]] != qry_ids[idx[i]])
          for (uint32_t k = 0; i + k < m && m + k < j; ++k) {
            const uint32_t p = idx[i + k], p2 = idx[m + k];
            const uint32_t a = (uint32_t)ref_ids[p], b = (uint32_t)qry_ids[p];
            const bool first = deg[a] != deg[b] ? deg[a] > deg[b] : ((a < b) != (((a + b) & 1u) != 0));
            if (first) { mirror[p] = (int32_t)p2; seeded[p2] = 0; } else { mirror[p2] = (int32_t)p; seeded[p] = 0; }
          }
        i = j;
      }
    }
    std::vector<int32_t> seed_refs, seed_qrys;
    for (uint32_t p = 0; p < limit; ++p)
      if (seeded[p]) {
        if (seed_refs.empty() || seed_refs.back() != ref_ids[p]) seed_refs.push_back(ref_ids[p]);
        seed_qrys.push_back(qry_ids[p]);
      }
    std::sort(seed_qrys.begin(), seed_qrys.end());
    seed_qrys.erase(std::unique(seed_qrys.begin(), seed_qrys.end()), seed_qrys.end());
    int rc2;
    if ((rc2 = anim_ensure_lists(ctx, A, seed_refs, seed_qrys, qstep))) return rc2;
    {   // (entries of genomes this batch uses are complete and never change while the genomes are resident; the vector itself
        // may be resized by another worker, so take the pointers under the lock)
      std::lock_guard<std::mutex> lk(ctx->anim_mu);
      LSv = anim_lists(ctx)->gidx;
    }
    uint32_t max_group = 1;
    for (int32_t g : seed_refs) if (LSv[g].ref_max > max_group) max_group = LSv[g].ref_max;
    slots = 256;
    while (slots < 2 * max_group) slots <<= 1;
    if (pg_dev_env("PYANI_SEED_SLOTS_MIN")) {   // development: a larger LDS table (lower load, shorter probe sequences, fewer workgroups per CU)
      const uint32_t want = (uint32_t)atoi(pg_dev_env("PYANI_SEED_SLOTS_MIN"));
      while (slots < want && slots < SEED_MAX_SLOTS) slots <<= 1;
    }
    if (slots > SEED_MAX_SLOTS)
      return pg_fail(ctx, PG_E_CAPACITY, "anim seeding: a reference k-mer group does not fit the LDS table (genome too large or too repetitive)");
    srefs.clear();   // one entry per reference with seeded pairs: [pair_begin, pair_end) spans them (pairs in between that
                     // are not seeded have empty slices)
    for (uint32_t p = 0; p < n_pairs; ++p) {
      sqry[p] = SeedQry{nullptr, nullptr};
      if (p >= limit || !seeded[p]) continue;
      const GenomeIdx& X = LSv[qry_ids[p]];
      sqry[p] = qstep == 1 ? SeedQry{X.qry_list1, X.qry_goff1} : SeedQry{X.qry_list, X.qry_goff};
      if (srefs.empty() || srefs.back().list != LSv[ref_ids[p]].ref_list)
        srefs.push_back(SeedRef{LSv[ref_ids[p]].ref_list, LSv[ref_ids[p]].ref_goff, p, p + 1});
      srefs.back().pair_end = p + 1;
    }
    n_srefs = (uint32_t)srefs.size();
    PG_HIP(ctx, hipMemcpyAsync(A->sqry_d, sqry.data(), n_pairs * sizeof(SeedQry), hipMemcpyHostToDevice, cur_stream(ctx)));
    PG_HIP(ctx, hipMemcpyAsync(A->mirror_d, mirror.data(), n_pairs * sizeof(int32_t), hipMemcpyHostToDevice, cur_stream(ctx)));
    if (n_srefs) PG_HIP(ctx, hipMemcpyAsync(A->srefs_d, srefs.data(), n_srefs * sizeof(SeedRef), hipMemcpyHostToDevice, cur_stream(ctx)));
    hipLaunchKernelGGL(anim_slice_kernel, dim3(n_pairs), dim3(256), 0, cur_stream(ctx), A->sqry_d, slice_stride, A->slice_d);
    return PG_OK;
  };
  if (!A->lds_attr_set) {   // per context = per device (the attribute is a property of the function ON a device)
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(anim_seed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(SEED_MAX_SLOTS * 8 + SEED_STAGE_BYTES)));
    A->lds_attr_set = true;
  }
  // The append buffer of the seed pass and the hit buffers hold the batch budget for a large call (no overflow re-runs), but no more
  // than the call can produce: a pair of genomes cannot have more maximal matches than a quarter of its bases (a one-pair call
  // — smoke(), pg_anim_pair_alignments — used to reserve 3 x 8 GB for a few MB of matches).  Both grow on overflow (below).
  size_t call_bound = 0;
  for (uint32_t p = 0; p < n_pairs && call_bound <= max_matches; ++p)
    call_bound += (size_t)(ctx->genomes[ref_ids[p]].stream_len + ctx->genomes[qry_ids[p]].stream_len) / 4 + 8192;
  const size_t budget = call_bound < max_matches ? call_bound : (size_t)max_matches;
  if (!A->seed_total) {
    if ((rc = regrow(ctx, A->seed_total, 2))) return rc;
  }
  if (budget + 1024 > A->seed_cap) {
    A->seed_cap = budget + 1024;
    if ((rc = regrow(ctx, A->seedbuf, A->seed_cap))) return rc;
  }
  std::vector<uint32_t> cnt(n_units), moff;
  uint32_t total = 0, pairs_fit = 0;
  // hit buffer: the matches of the budget plus the chance 16-mer hits of unrelated pairs (~1200 per 5 Mb unit)
  {
    const size_t want = budget + (size_t)(frag ? 8 * 4096 : 4096) * n_units + 1024;   // (every position sampled: 5 x the chance hits)
    if (want > A->hit_cap) {
      if ((rc = regrow(ctx, A->hits_d, want))) return rc;
      if ((rc = regrow(ctx, A->hits_sorted, want))) return rc;
      A->hit_cap = want;
    }
  }
  for (int attempt = 0;; ++attempt) {
    if (attempt == 8) return pg_fail(ctx, PG_E_CAPACITY, "anim seeding: buffers still overflow after repeated splitting");
    uint32_t counts[2] = {0, 0};   // matches appended, hits recorded
    if ((rc = prepare(n_pairs))) return rc;   // (again after a split: a pair whose partner left the launch is seeded itself)
    PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, cur_stream(ctx)));
    PG_HIP(ctx, hipMemsetAsync(A->seed_total, 0, 8, cur_stream(ctx)));   // [0] matches, [1] hits
    PG_HIP(ctx, hipMemsetAsync(A->hit_count, 0, n_units * 4, cur_stream(ctx)));
    pg_prof_begin(ctx, PG_K_ANIM_SEED);
    if (n_srefs)
    hipLaunchKernelGGL(anim_seed_kernel, dim3(n_srefs, SEED_GROUPS), dim3(SEED_BLOCK), (size_t)slots * 8 + SEED_STAGE_BYTES, cur_stream(ctx),
                       A->refs_d, A->units_d, A->srefs_d, A->sqry_d, A->slice_d, slice_stride, slots - 1, A->hits_d,
                       (uint32_t)A->hit_cap, A->seed_total + 1, A->hit_count, qstep);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipGetLastError());   // a rejected launch (LDS size) must not surface only at the end of the batch
    pg_prof_begin(ctx, PG_K_ANIM_HIT);
    // hits -> per-unit slices, then one workgroup per unit verifies / extends them
    hipLaunchKernelGGL(anim_hoff_kernel, dim3(1), dim3(1024), 0, cur_stream(ctx), A->hit_count, n_units, A->hoff, A->hit_cursor);
    hipLaunchKernelGGL(anim_hit_scatter_kernel, dim3((uint32_t)ctx->num_cu * 8u), dim3(256), 0, cur_stream(ctx), A->hits_d, A->seed_total + 1,
                       (uint32_t)A->hit_cap, A->hoff, A->hit_cursor, A->hits_sorted);
    hipLaunchKernelGGL(anim_hit_kernel, dim3(n_units), dim3(256), 0, cur_stream(ctx), A->refs_d, A->units_d, A->hits_sorted, A->hoff,
                       A->seed_total + 1, (uint32_t)A->hit_cap, A->seedbuf, (uint32_t)A->seed_cap, A->seed_total, A->mem_count,
                       frag ? FRAG_SEED_MIN : MIN_MATCH, qstep, use_mirror ? A->mirror_d : (const int32_t*)nullptr);
    pg_prof_end(ctx);
    PG_HIP(ctx, hipMemcpyAsync(cnt.data(), A->mem_count, n_units * 4, hipMemcpyDeviceToHost, cur_stream(ctx)));
    PG_HIP(ctx, hipMemcpyAsync(counts, A->seed_total, 8, hipMemcpyDeviceToHost, cur_stream(ctx)));
    PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    total = counts[0];
    if (counts[1] > A->hit_cap) {   // hits were dropped: the counts are incomplete -> seed half as many pairs
      if (n_pairs == 1) {
        A->hit_cap = (size_t)counts[1] + 1024;
        if ((rc = regrow(ctx, A->hits_d, A->hit_cap))) return rc;
        if ((rc = regrow(ctx, A->hits_sorted, A->hit_cap))) return rc;
      } else {
        n_pairs = (n_pairs + 1) / 2;
        n_units = 2 * n_pairs;
      }
      continue;
    }
    uint64_t tot = 0, raw = 0;
    pairs_fit = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
      const uint64_t need = (uint64_t)cnt[2 * p] + cnt[2 * p + 1] + 4;   // per unit: count + 1 (never empty), rounded up to even
      if (p > 0 && tot + need > max_matches) break;
      tot += need;
      raw += need - 4;
      pairs_fit = p + 1;
    }
    n_pairs = pairs_fit;
    n_units = 2 * n_pairs;
    if (total <= A->seed_cap) break;
    // overflow: make room for the prefix of pairs that fits the batch budget and seed that prefix again
    A->seed_cap = (size_t)(raw + raw / 8 + 1024);
    if ((rc = regrow(ctx, A->seedbuf, A->seed_cap))) return rc;
  }
  *n_done = n_pairs;
  moff.assign((size_t)n_units + 1, 0);
  for (uint32_t u = 0; u < n_units; ++u) moff[u + 1] = moff[u] + ((cnt[u] + 2) & ~1u);   // even slice sizes: 8-byte aligned sub-slices
  const size_t M = moff[n_units];
  if (M > A->matches) {
    const size_t cap = M + M / 4;
    if ((rc = regrow(ctx, A->mem, cap))) return rc;
    if ((rc = regrow(ctx, A->cm, cap))) return rc;
    if ((rc = regrow(ctx, A->iscratch, cap * 8))) return rc;
    if ((rc = regrow(ctx, A->chains, cap))) return rc;
    if ((rc = regrow(ctx, A->order, cap))) return rc;
    if ((rc = regrow(ctx, A->S.alns, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_rrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_qrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.idx, cap))) return rc;
    if ((rc = regrow(ctx, A->S.from, cap))) return rc;
    if ((rc = regrow(ctx, A->S.sc, cap))) return rc;
    __atomic_fetch_add(&ctx->anim_scratch_matches_held, (uint64_t)(cap - A->matches), __ATOMIC_RELAXED);      // (what pg_api.cpp's anim_match_budget may count as available)
    A->matches = cap;
  }
  PG_HIP(ctx, hipMemcpyAsync(A->moff, moff.data(), ((size_t)n_units + 1) * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
  PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, cur_stream(ctx)));
  PG_HIP(ctx, hipMemsetAsync(A->status, 0, n_pairs * 4, cur_stream(ctx)));
  ClusterOut O{A->moff, A->cm, A->chains, A->nch, A->order, A->status};
  pg_prof_begin(ctx, PG_K_ANIM_HIT);
  if (total)
    hipLaunchKernelGGL(anim_scatter_kernel, dim3((total + 255) / 256), dim3(256), 0, cur_stream(ctx), A->seedbuf, total, A->moff, n_units,
                       A->mem_count, A->mem);
  pg_prof_end(ctx);
  if (frag) {   // fragment mode: the matches of every unit are in place; the rest of the batch is the fragment kernels
    PG_HIP(ctx, hipGetLastError());
    return anib_frag_stage(ctx, A, qry_ids, n_pairs, cnt, *frag, ref_list, ref_of_pair);
  }
  // Units with >= split_min matches ("big": pairs of related genomes) get their chains from many waves (pga_cluster.inc,
  // anim_chain_range_kernel); every other unit is finished by the one wave that filters and clusters it.
  const int split_min = pg_dev_env("PYANI_ANIM_SPLIT_MIN") ? atoi(pg_dev_env("PYANI_ANIM_SPLIT_MIN")) : 2048;   // (<= 0: never split)
  const int range_entries = pg_dev_env("PYANI_ANIM_RANGE_ENTRIES") && atoi(pg_dev_env("PYANI_ANIM_RANGE_ENTRIES")) > 0
                                ? atoi(pg_dev_env("PYANI_ANIM_RANGE_ENTRIES")) : CHAIN_RANGE_ENTRIES;
  std::vector<BigUnit> big;
  std::vector<uint2> ranges;
  if (split_min > 0 && !pg_dev_env("PYANI_ANIM_SCALAR_CLUSTER"))
    for (uint32_t u = 0; u < n_units; ++u)
      if (cnt[u] >= (uint32_t)split_min) {
        uint32_t R = cnt[u] / (uint32_t)range_entries;
        R = R < 1 ? 1 : (R > (uint32_t)CHAIN_RANGES_MAX ? (uint32_t)CHAIN_RANGES_MAX : R);
        for (uint32_t r = 0; r < R; ++r) ranges.push_back(make_uint2((uint32_t)big.size(), r));
        big.push_back(BigUnit{u, (uint32_t)(ranges.size() - R), R, 0});
      }
  if (!big.empty()) {
    if (big.size() > A->big_cap) { if ((rc = regrow(ctx, A->big_d, big.size() + big.size() / 4))) return rc; A->big_cap = big.size() + big.size() / 4; }
    if (ranges.size() > A->range_cap) {
      const size_t c = ranges.size() + ranges.size() / 4;
      if ((rc = regrow(ctx, A->ranges_d, c))) return rc;
      if ((rc = regrow(ctx, A->range_out, c))) return rc;
      A->range_cap = c;
    }
    PG_HIP(ctx, hipMemcpyAsync(A->big_d, big.data(), big.size() * sizeof(BigUnit), hipMemcpyHostToDevice, cur_stream(ctx)));
    PG_HIP(ctx, hipMemcpyAsync(A->ranges_d, ranges.data(), ranges.size() * sizeof(uint2), hipMemcpyHostToDevice, cur_stream(ctx)));
  }
  const int split_arg = big.empty() ? 0x7fffffff : split_min;
  pg_prof_begin(ctx, PG_K_ANIM_CLUSTER);
  if (pg_dev_env("PYANI_ANIM_SCALAR_CLUSTER") && !maxmatch)   // debugging aid: the one-thread-per-unit statement of the same algorithm
    hipLaunchKernelGGL(anim_cluster_kernel, dim3((n_units + 63) / 64), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, n_units,
                       A->mem, A->mem_count, A->iscratch, O);
  else {
    // the front half (MUM filter, union-find, grouping) of a big unit is shared by the PREP_WAVES waves of one workgroup: a
    // single wave needs ~10 ms for the sorts of 50 000 matches, and the launch would wait for the slowest of them (measured
    // on C4, cluster stage per grid: one wave per big unit 1.95 s, workgroup 0.51 s; PYANI_ANIM_WAVE_PREP=1 forces the former)
    const bool prep = !big.empty() && !pg_dev_env("PYANI_ANIM_WAVE_PREP");
    if (prep)
      hipLaunchKernelGGL(anim_cluster_prep_kernel, dim3((uint32_t)big.size()), dim3(PREP_THREADS), 0, cur_stream(ctx), A->refs_d, A->units_d,
                         A->mem, A->mem_count, A->iscratch, O, maxmatch, A->big_d);
    hipLaunchKernelGGL(anim_cluster_wave_kernel, dim3(n_units), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->mem,
                       A->mem_count, A->iscratch, O, prep ? 1 : 0, maxmatch, split_arg);
    if (!big.empty()) {
      hipLaunchKernelGGL(anim_chain_range_kernel, dim3((uint32_t)ranges.size()), dim3(64), 0, cur_stream(ctx), A->units_d, A->mem, A->iscratch,
                         O, A->big_d, A->ranges_d, A->range_out);
      hipLaunchKernelGGL(anim_chain_merge_kernel, dim3((uint32_t)big.size()), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->iscratch, O,
                         A->big_d, A->range_out);
    }
  }
  pg_prof_end(ctx);
  // work list of (unit, chain): one wave each
  std::vector<int32_t> nch(n_units);
  PG_HIP(ctx, hipMemcpyAsync(nch.data(), A->nch, n_units * 4, hipMemcpyDeviceToHost, cur_stream(ctx)));
  PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
  // (unit, chain) work list, one wave each: offsets by a host prefix over the per-unit chain counts, entries on device
  std::vector<uint32_t> choff((size_t)n_units + 1, 0);
  for (uint32_t u = 0; u < n_units; ++u) choff[u + 1] = choff[u] + (uint32_t)nch[u];
  const size_t n_wl = choff[n_units];
  {
    // A4x: MUMmer's own extension algorithm, one wave per unit (persistent waves, units handed out longest first would be
    // better still: a unit's time is ~ its clusters; the cursor takes them in batch order)
    const size_t Mp = (M + 15) & ~(size_t)15;
    if (Mp > A->pn_cap) {
      if ((rc = regrow(ctx, A->pn, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_fused, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_gaps, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_fwd, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_bwd, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_tlog, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_born, Mp))) return rc;
      if ((rc = regrow(ctx, A->pn_tasks, 4 * Mp))) return rc;      // three lane classes + the wave engine's list
      A->pn_cap = Mp;
    }
    if (n_units > A->pn_units) {
      if ((rc = regrow(ctx, A->pn_n, (size_t)n_units + n_units / 2))) return rc;
      if ((rc = regrow(ctx, A->pn_order, (size_t)n_units + n_units / 2))) return rc;
      if ((rc = regrow(ctx, A->pn_porder, (size_t)n_units + n_units / 2))) return rc;
      A->pn_units = (size_t)n_units + n_units / 2;
    }
    {
      std::vector<uint32_t> uorder(n_units);
      for (uint32_t u = 0; u < n_units; ++u) uorder[u] = u;
      std::stable_sort(uorder.begin(), uorder.end(), [&](uint32_t a, uint32_t b) { return nch[a] > nch[b]; });
      PG_HIP(ctx, hipMemcpyAsync(A->pn_order, uorder.data(), (size_t)n_units * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
      std::vector<uint32_t> porder(n_pairs);      // the walk kernel takes PAIRS (one wave per strand): by the larger strand's cluster count
      for (uint32_t p = 0; p < n_pairs; ++p) porder[p] = p;
      std::stable_sort(porder.begin(), porder.end(), [&](uint32_t a, uint32_t b) { return std::max(nch[2 * a], nch[2 * a + 1]) > std::max(nch[2 * b], nch[2 * b + 1]); });
      PG_HIP(ctx, hipMemcpyAsync(A->pn_porder, porder.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
      PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));     // (uorder / porder are locals)
    }
    if (!A->pn_cursor && (rc = regrow(ctx, A->pn_cursor, 24))) return rc;
    const uint32_t pn_waves = (uint32_t)ctx->num_cu * 12u;   // forced kernels with the LDS store: 12 KiB of LDS each: 12 per CU
    const uint32_t pn_walk_waves = (uint32_t)ctx->num_cu * 8u;   // the walk / rehearsal kernels: 219 / 173 VGPRs, two waves per SIMD — one persistent wave per resident slot
    const uint32_t pn_waves_pre = (uint32_t)ctx->num_cu * 32u;   // gap / forward / backward pre-passes and the narrow forced kernel: no LDS, diagonal engine only, <= 64 registers: 8 per SIMD
    const uint32_t pn_waves_scr = ctx->anim_gap_lanes ? (pn_waves > pn_walk_waves ? pn_waves : pn_walk_waves) : pn_waves_pre;      // (only the walks, the wide forced kernel and the all-gaps form of the gap kernel use the global scratch)
    if (pn_waves_scr > A->pn_waves) { if ((rc = regrow(ctx, A->pn_gscratch, (size_t)pn_waves_scr * PN_GLOBAL_WORDS))) return rc; A->pn_waves = pn_waves_scr; }
    const bool trace = tls_sink && tls_sink->with_indels;      // the walks list their pieces and align everything themselves
    const bool bwd_ahead = ctx->anim_bwd_ahead != 0;
    if (trace) {
      const size_t need = pn_piece_base(Mp, (uint32_t)n_wl, n_units) + 16;
      if (need > A->pn_piece_cap) { if ((rc = regrow(ctx, A->pn_pieces, need))) return rc; A->pn_piece_cap = need; }
      if (n_units > A->pn_npieces_cap) { if ((rc = regrow(ctx, A->pn_npieces, (size_t)n_units + 16))) return rc; A->pn_npieces_cap = (size_t)n_units + 16; }
      PG_HIP(ctx, hipMemsetAsync(A->pn_npieces, 0, (size_t)n_units * 4, cur_stream(ctx)));
    }
    const size_t req_cap = Mp + 16;      // one slot per match slot: a walk records at most one forced run per alignment it starts, and starts at most one per match
    A->pn_req_n = req_cap;
    if (req_cap > A->pn_req_cap) { if ((rc = regrow(ctx, A->pn_reqs, req_cap + req_cap / 2))) return rc; if ((rc = regrow(ctx, A->pn_wide, 3 * (req_cap + req_cap / 2)))) return rc; A->pn_req_cap = req_cap + req_cap / 2; }
    PG_HIP(ctx, hipMemsetAsync(A->pn_cursor, 0, 96, cur_stream(ctx)));   // [0] unit cursor, [1] forced runs recorded ([7]: the long ones), [2] forced-run cursor, [3] chain / big-gap cursor, [4..6] small gaps by class ([11]: gaps left to the wave engine), [8] cluster cursor of the forward extensions, [9] unit cursor of the rehearsal, [10] cluster cursor of the backward searches, [12] / [13] wide forced runs: count / cursor, [14] / [15] huge ones, [16] / [17] the strips' list
    if (n_wl && trace) PG_HIP(ctx, hipMemcpyAsync(A->choff_d, choff.data(), choff.size() * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
    if (n_wl && !trace) {     // the (unit, chain) work list, then every cluster's match-to-match alignments
      if (n_wl > A->wl) { if ((rc = regrow(ctx, A->wl_d, n_wl + n_wl / 2))) return rc; A->wl = n_wl + n_wl / 2; }
      PG_HIP(ctx, hipMemcpyAsync(A->choff_d, choff.data(), choff.size() * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
      hipLaunchKernelGGL(anim_wl_kernel, dim3(n_units), dim3(64), 0, cur_stream(ctx), A->choff_d, A->wl_d);
      pg_prof_begin(ctx, PG_K_ANIM_GAPS);
      const int lane_small = ctx->anim_gap_lanes;
      if (lane_small) {     // small gaps: one LANE each, by size class
        hipLaunchKernelGGL(anim_postnuc_gaplist_kernel, dim3((uint32_t)((n_wl + 255) / 256)), dim3(256), 0, cur_stream(ctx), A->units_d, O, A->wl_d,
                           (uint32_t)n_wl, A->pn_tasks, A->pn_cap, A->pn_cursor + 4);
        const dim3 lg((uint32_t)ctx->num_cu * 8u);
        hipLaunchKernelGGL((anim_postnuc_gaplane_kernel<16>), lg, dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->pn_tasks, A->pn_cursor + 4, A->pn_gaps);
        hipLaunchKernelGGL((anim_postnuc_gaplane_kernel<32>), lg, dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->pn_tasks + A->pn_cap, A->pn_cursor + 5, A->pn_gaps);
        hipLaunchKernelGGL((anim_postnuc_gaplane_kernel<PN_SMALL>), lg, dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->pn_tasks + 2 * A->pn_cap, A->pn_cursor + 6, A->pn_gaps);
      }
      if (lane_small)
        hipLaunchKernelGGL(anim_postnuc_gapbig_kernel, dim3(pn_waves_pre), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->pn_cursor + 3,
                           A->pn_gaps, A->pn_tasks + 3 * A->pn_cap, A->pn_cursor + 11);
      else
        hipLaunchKernelGGL(anim_postnuc_gap_kernel, dim3(pn_waves_pre), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->wl_d, (uint32_t)n_wl,
                           A->pn_cursor + 3, A->pn_gaps, A->pn_gscratch);
      pg_prof_end(ctx);
      pg_prof_begin(ctx, PG_K_ANIM_FWD);
      hipLaunchKernelGGL(anim_postnuc_fwd_kernel, dim3(pn_waves_pre), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->wl_d, (uint32_t)n_wl,
                         A->pn_cursor + 8, A->pn_fwd, A->pn_gscratch);
      pg_prof_end(ctx);
      pg_prof_begin(ctx, PG_K_ANIM_BWD);
      if (bwd_ahead) {     // the walks rehearsed without their backward searches, then the searches they predict, one wave each
        PG_HIP(ctx, hipMemsetAsync(A->pn_bwd, 0, (size_t)M * sizeof(pgn::PnBwd), cur_stream(ctx)));
        hipLaunchKernelGGL(anim_postnuc_rehearse_kernel, dim3(pn_walk_waves < n_units ? pn_walk_waves : n_units), dim3(64), 0, cur_stream(ctx), A->refs_d,
                           A->units_d, n_units, O, A->pn_cursor + 9, A->pn, A->pn_fused, A->pn_gscratch, A->pn_gaps, A->pn_fwd, A->pn_order, A->pn_bwd);
        hipLaunchKernelGGL(anim_postnuc_bwd_kernel, dim3(pn_waves_pre), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, O, A->wl_d, (uint32_t)n_wl,
                           A->pn_cursor + 10, A->pn_bwd, A->pn_gscratch);
      }
      pg_prof_end(ctx);
    }
    pg_prof_begin(ctx, PG_K_ANIM_EXTEND);
    if (n_wl)
      hipLaunchKernelGGL(anim_postnuc_kernel, dim3(pn_walk_waves / 2 < n_pairs ? pn_walk_waves / 2 : n_pairs), dim3(128), 0, cur_stream(ctx), A->refs_d, A->units_d,
                         n_pairs, O, A->pn_cursor, A->pn, A->pn_fused, A->pn_n, A->pn_gscratch, A->pn_reqs, A->pn_cursor + 1, (uint32_t)req_cap,
                         trace ? nullptr : A->pn_gaps, trace ? nullptr : A->pn_fwd, A->pn_porder, trace ? A->pn_pieces : nullptr, A->pn_npieces, A->choff_d,
                         bwd_ahead && !trace ? A->pn_bwd : nullptr, A->pn_tlog, A->pn_born);
    else
      PG_HIP(ctx, hipMemsetAsync(A->pn_n, 0, (size_t)n_units * 4, cur_stream(ctx)));
    pg_prof_end(ctx);
    pg_prof_begin(ctx, PG_K_ANIM_EXTLANE);     // (the forced re-alignments, deferred: pga_postnuc.inc)
    if (n_wl) {      // narrow bands first (five waves per SIMD), then the runs that asked for a wide one (pga_postnuc.inc, pn_forced_wave)
      const uint32_t win_max = (uint32_t)ctx->anim_pn_window_max, group_max = (uint32_t)ctx->anim_pn_group_max;
      hipLaunchKernelGGL(anim_postnuc_forced_kernel, dim3(pn_waves_pre), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->pn_reqs,
                         A->pn_cursor + 1, (uint32_t)req_cap, A->pn_cursor + 2, A->pn_n, A->pn_wide, A->pn_cursor + 12, win_max);
      hipLaunchKernelGGL(anim_postnuc_forced_wide_kernel, dim3(pn_waves), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->pn_reqs,
                         A->pn_cursor + 1, (uint32_t)req_cap, A->pn_cursor + 13, A->pn_n, A->pn_gscratch, A->pn_wide, A->pn_cursor + 12,
                         A->pn_wide + req_cap, A->pn_cursor + 14, win_max);
      // runs whose band spans more than one wave's 2048 diagonals: a workgroup of four waves each (2 workgroups per CU: the 8192-diagonal form holds 242 VGPRs); what the group
      // cannot hold either: the column strips, one wave per run (a list that is empty on every workload seen so far)
      hipLaunchKernelGGL(anim_postnuc_forced_huge_kernel, dim3((uint32_t)ctx->num_cu * 2u), dim3(64 * PN_HUGE_WAVES), 0, cur_stream(ctx), A->refs_d,
                         A->units_d, A->pn_reqs, A->pn_cursor + 15, A->pn_n, A->pn_wide + req_cap, A->pn_cursor + 14, A->pn_wide + 2 * req_cap,
                         A->pn_cursor + 16, group_max);
      hipLaunchKernelGGL(anim_postnuc_forced_strips_kernel, dim3(pn_waves), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->pn_reqs,
                         A->pn_cursor + 17, A->pn_n, A->pn_gscratch, A->pn_wide + 2 * req_cap, A->pn_cursor + 16);
    }
    pg_prof_end(ctx);
    if (pg_dev_env("PYANI_PN_STATS")) {   // development: what the engines did in this launch
      PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
      unsigned long long st[32], zero[32] = {0};
      PG_HIP(ctx, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pn_stats), sizeof(st)));
      PG_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pn_stats), zero, sizeof(zero)));
      fprintf(stderr, "[pn-stats] units %llu clusters %llu | regs: calls %llu steps %llu cells %llu moves %llu overflows %llu | lds: calls %llu steps %llu cells %llu | "
                      "global: calls %llu steps %llu cells %llu\n", st[11], st[12], st[0], st[1], st[2], st[9], st[10], st[3], st[4], st[5], st[6], st[7], st[8]);
      fprintf(stderr, "[pn-stats] searches of the gap + units kernels: %llu calls, %.1f ms inside the engine (summed over waves); shadow tests that asked for the synteny's current alignment: %llu\n", st[22], st[21] / 1e5, st[31]);
      fprintf(stderr, "[pn-stats] forced passes by engine (127 / 255 / 511 cells / strips): %llu %llu %llu %llu passes, %.1f %.1f %.1f %.1f ms summed over waves\n",
              st[27], st[28], st[29], st[30], st[23] / 1e5, st[24] / 1e5, st[25] / 1e5, st[26] / 1e5);
      {
        unsigned long long ks[32], kz[32] = {0};
        PG_HIP(ctx, hipMemcpyFromSymbol(ks, HIP_SYMBOL(g_pn_kstats), sizeof(ks)));
        PG_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_pn_kstats), kz, sizeof(kz)));
        const char* kn[8] = {"gaps", "forward", "backward-ahead", "walks", "forced narrow", "forced 512-1024", "forced 2048", "forced group 8192"};
        for (int k = 0; k < 8; ++k)
          fprintf(stderr, "[pn-stats] diagonal engine in %-16s: %llu calls, %llu anti-diagonals, %llu cells\n", kn[k], ks[4 * k], ks[4 * k + 1], ks[4 * k + 2]);
        fprintf(stderr, "[pn-stats] walk kernel scans: shadow test %.1f ms over %llu rows of 64 alignments, reverse-target search %.1f ms in %llu calls (summed over waves)\n",
                ks[3] / 1e5, ks[7], ks[11] / 1e5, ks[15]);
        fprintf(stderr, "[pn-stats] run-ahead results the walk took: forward %llu of %llu computed (%.4f), backward %llu found ready of %llu searches run ahead; %llu searches left to the walk itself\n",
                ks[19], ks[4], ks[4] ? (double)ks[19] / (double)ks[4] : 0.0, ks[23], ks[8], ks[12]);
      }
      for (int k = 13; k <= 17; k += 4)      // ticks of the 100 MHz wall clock -> ms
        fprintf(stderr, "[pn-stats] %s: busy %.1f ms summed over waves, span %.1f ms, longest item %.1f ms (size %llu)\n", k == 13 ? "units" : "forced",
                st[k] / 1e5, st[k + 2] ? (st[k + 2] - ~st[k + 3]) / 1e5 : 0.0, (st[k + 1] >> 20) / 1e5, st[k + 1] & 0xFFFFFull);
    }
  }
  pg_prof_begin(ctx, PG_K_ANIM_FINISH);
  hipLaunchKernelGGL(anim_finish_kernel, dim3(n_pairs), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, n_pairs,
                     O, A->pn, A->pn_n, A->S, filter_1to1, A->out);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  PG_HIP(ctx, hipMemcpyAsync(out_host, A->out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, cur_stream(ctx)));
  PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
  if (tls_sink && (rc = anim_collect(ctx, A, ref_ids, qry_ids, n_pairs, out_host, choff, *tls_sink))) return rc;
  return PG_OK;
}

// Fragment mode after seeding: A->mem / A->moff / A->mem_count hold every unit's exact matches (>= 16, sampled), A->units_d /
// A->refs_d the descriptors.  Builds the fragment tables of the batch's query genomes, runs F1-F3 (pga_frag.inc), returns
// the pair results (and, optionally, the rows of pair 0).
// The word index of one genome (pga_frag.inc), built once and kept with its seed lists.
static int anib_ensure_word_index(pg_ctx* ctx, AnimScratch* A, int32_t gid) {
  int rc;
  std::lock_guard<std::mutex> lk(ctx->anim_mu);
  AnimLists* LS = anim_lists(ctx);
  if (LS->gidx.size() < ctx->genomes.size()) LS->gidx.resize(ctx->genomes.size());
  GenomeIdx& X = LS->gidx[gid];
  if (X.word_start) return PG_OK;
  const PgGenome& G = ctx->genomes[gid];
  const int32_t len = (int32_t)G.stream_len;
  const uint32_t* codes = ctx->d_codes + G.arena_start / 16;
  const uint32_t* mask = ctx->d_mask + G.arena_start / 32;
  uint32_t* start = nullptr;
  int32_t* pos = nullptr;
  if (!A->fr_wtmp && (rc = regrow(ctx, A->fr_wtmp, (size_t)WORD_BUCKETS + 1024 + 16))) return rc;   // fill cursors | block sums
  if ((rc = regrow(ctx, start, (size_t)WORD_BUCKETS + 1))) return rc;
  if ((rc = regrow(ctx, pos, (size_t)(len > 0 ? len : 1)))) { (void)hipFree(start); return rc; }
  hipStream_t st = cur_stream(ctx);
  const uint32_t grid = (uint32_t)((len + 255) / 256);
  hipError_t e = hipMemsetAsync(start, 0, ((size_t)WORD_BUCKETS + 1) * 4, st);
  if (e == hipSuccess && grid) hipLaunchKernelGGL(anib_word_count_kernel, dim3(grid), dim3(256), 0, st, codes, mask, len, start, pos, 0);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anib_word_scan1_kernel, dim3(WORD_BUCKETS / 4096u), dim3(1024), 0, st, start, A->fr_wtmp + WORD_BUCKETS);
    hipLaunchKernelGGL(anib_word_scan2_kernel, dim3(1), dim3(1024), 0, st, A->fr_wtmp + WORD_BUCKETS, WORD_BUCKETS / 4096u, start + WORD_BUCKETS);
    hipLaunchKernelGGL(anib_word_scan3_kernel, dim3(WORD_BUCKETS / 1024u), dim3(1024), 0, st, start, A->fr_wtmp + WORD_BUCKETS, A->fr_wtmp);
    if (grid) hipLaunchKernelGGL(anib_word_count_kernel, dim3(grid), dim3(256), 0, st, codes, mask, len, A->fr_wtmp, pos, 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(start); (void)hipFree(pos); return pg_fail(ctx, PG_E_HIP, hipGetErrorString(e)); }
  X.word_start = start; X.word_pos = pos;     // published complete
  return PG_OK;
}

static int anib_frag_stage(pg_ctx* ctx, AnimScratch* A, const int32_t* qry_ids, uint32_t n_pairs, const std::vector<uint32_t>& cnt,
                           const PgFragArgs& F, const std::vector<int32_t>& ref_list, const std::vector<uint32_t>& ref_of_pair) {
  int rc;
  const uint32_t n_units = 2 * n_pairs;
  // fragment tables, one per distinct query genome
  std::vector<int32_t> tables;
  struct Tab { size_t pos, len, rec0; int32_t n_frags; };
  std::vector<Tab> tab_of(ctx->genomes.size(), Tab{0, 0, 0, -1});
  std::vector<FragPair> fp(n_pairs);
  std::vector<size_t> tab_ref(n_pairs);
  uint64_t slots = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const int32_t gid = qry_ids[p];
    Tab& T = tab_of[gid];
    if (T.n_frags < 0) {
      const PgGenome& Q = ctx->genomes[gid];
      std::vector<int32_t> pos, len, rec0;
      for (uint32_t r = 0; r < Q.n_rec; ++r) {
        rec0.push_back((int32_t)pos.size());
        const int32_t r0 = Q.rec_start[r], r1 = Q.rec_start[r + 1] - 1;
        for (int32_t f0 = r0; f0 < r1; f0 += F.fragsize) { pos.push_back(f0); len.push_back(r1 - f0 < F.fragsize ? r1 - f0 : F.fragsize); }
      }
      T.n_frags = (int32_t)pos.size();
      T.pos = tables.size(); tables.insert(tables.end(), pos.begin(), pos.end());
      T.len = tables.size(); tables.insert(tables.end(), len.begin(), len.end());
      T.rec0 = tables.size(); tables.insert(tables.end(), rec0.begin(), rec0.end());
    }
    fp[p].n_frags = T.n_frags;
    fp[p].slot0 = (uint32_t)slots;
    slots += (uint64_t)T.n_frags;
  }
  if (slots >= (1ull << 31)) return pg_fail(ctx, PG_E_CAPACITY, "fragment mode: too many (pair, fragment) slots in one launch");
  if (tables.size() + 1 > A->fr_tables_cap) { if ((rc = regrow(ctx, A->fr_tables, tables.size() + 1024))) return rc; A->fr_tables_cap = tables.size() + 1024; }
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const Tab& T = tab_of[qry_ids[p]];
    fp[p].frag_pos = A->fr_tables + T.pos; fp[p].frag_len = A->fr_tables + T.len; fp[p].rec_frag0 = A->fr_tables + T.rec0;
  }
  if (n_pairs > A->fr_pairs_cap) {
    if ((rc = regrow(ctx, A->fr_pairs, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->fr_out, n_pairs))) return rc;
    A->fr_pairs_cap = n_pairs;
  }
  if (n_units > A->fr_units_cap) { if ((rc = regrow(ctx, A->fr_ebase, n_units))) return rc; A->fr_units_cap = n_units; }
  const size_t n_off = 2 * (size_t)slots + 2 * (size_t)n_pairs;
  if (n_off > A->fr_off_cap) { if ((rc = regrow(ctx, A->fr_off, n_off + n_off / 4))) return rc; A->fr_off_cap = n_off + n_off / 4; }
  if (slots > A->fr_slots_cap) {
    const size_t cap = (size_t)slots + (size_t)slots / 4;
    if ((rc = regrow(ctx, A->fr_slot_pair, cap))) return rc;
    if ((rc = regrow(ctx, A->fr_nrows, cap))) return rc;
    if ((rc = regrow(ctx, A->fr_rows, cap * FRAG_ROWS))) return rc;
    A->fr_slots_cap = cap;
  }
  std::vector<uint32_t> slot_pair((size_t)slots);
  for (uint32_t p = 0; p < n_pairs; ++p) std::fill(slot_pair.begin() + fp[p].slot0, slot_pair.begin() + fp[p].slot0 + fp[p].n_frags, p);
  // a match is clipped into at most 1 + (fragment boundaries it crosses) seeds: <= count + n_frags per unit
  std::vector<uint64_t> ebase(n_units);
  uint64_t n_entries = 0;
  for (uint32_t u = 0; u < n_units; ++u) { ebase[u] = n_entries; n_entries += (uint64_t)cnt[u] + (uint64_t)fp[u / 2].n_frags; }
  if (n_entries > A->fr_entries_cap) { if ((rc = regrow(ctx, A->fr_entries, (size_t)n_entries + (size_t)n_entries / 4))) return rc; A->fr_entries_cap = (size_t)n_entries + (size_t)n_entries / 4; }
  if (!tables.empty()) PG_HIP(ctx, hipMemcpyAsync(A->fr_tables, tables.data(), tables.size() * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
  PG_HIP(ctx, hipMemcpyAsync(A->fr_pairs, fp.data(), n_pairs * sizeof(FragPair), hipMemcpyHostToDevice, cur_stream(ctx)));
  if (slots) PG_HIP(ctx, hipMemcpyAsync(A->fr_slot_pair, slot_pair.data(), (size_t)slots * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
  PG_HIP(ctx, hipMemcpyAsync(A->fr_ebase, ebase.data(), n_units * 8, hipMemcpyHostToDevice, cur_stream(ctx)));
  pg_prof_begin(ctx, PG_K_ANIB_BUCKET);
  hipLaunchKernelGGL(anib_bucket_kernel, dim3(n_units), dim3(256), 0, cur_stream(ctx), A->units_d, A->fr_pairs, A->mem, A->moff, A->mem_count,
                     A->fr_ebase, F.fragsize, A->fr_off, A->fr_entries);
  pg_prof_end(ctx);
  pg_prof_begin(ctx, PG_K_ANIB_FRAG);
  if (slots)
    hipLaunchKernelGGL(anib_frag_kernel, dim3((uint32_t)slots), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->fr_pairs, A->fr_slot_pair,
                       A->fr_off, A->fr_entries, A->fr_ebase, A->fr_rows, A->fr_nrows, (const uint32_t*)nullptr, (const WordIdx*)nullptr);
  pg_prof_end(ctx);
  hipLaunchKernelGGL(anib_reduce_pairs_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, cur_stream(ctx), A->fr_pairs, n_pairs, A->fr_rows, A->fr_nrows,
                     A->fr_out);
  PG_HIP(ctx, hipGetLastError());
  PG_HIP(ctx, hipMemcpyAsync(F.out, A->fr_out, n_pairs * sizeof(pg_anib_result), hipMemcpyDeviceToHost, cur_stream(ctx)));
  PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
  // ---- word tier (pga_frag.inc): pairs with some, but not all, fragments reportable get their other fragments searched again
  // with blastn-sized seeds; the subjects' word indices are built on first use
  if (slots && ctx->anib_word_tier) {
    std::vector<WordIdx> widx(ref_list.size(), WordIdx{nullptr, nullptr});
    bool any = false;
    for (uint32_t p = 0; p < n_pairs; ++p) {
      if (!(F.out[p].n_kept > 0 && F.out[p].n_kept < F.out[p].n_frags)) continue;
      const uint32_t r = ref_of_pair[p];
      if (widx[r].start) continue;
      if ((rc = anib_ensure_word_index(ctx, A, ref_list[r]))) return rc;
      const GenomeIdx& X = anim_lists(ctx)->gidx[ref_list[r]];
      widx[r] = WordIdx{X.word_start, X.word_pos};
      any = true;
    }
    if (any) {
      if (widx.size() > A->fr_widx_cap) { if ((rc = regrow(ctx, A->fr_widx, widx.size() + 16))) return rc; A->fr_widx_cap = widx.size() + 16; }
      if ((size_t)slots > A->fr_list_cap) { if ((rc = regrow(ctx, A->fr_list, (size_t)slots + (size_t)slots / 4))) return rc; A->fr_list_cap = (size_t)slots + (size_t)slots / 4; }
      if (!A->fr_nlist && (rc = regrow(ctx, A->fr_nlist, 4))) return rc;
      PG_HIP(ctx, hipMemcpyAsync(A->fr_widx, widx.data(), widx.size() * sizeof(WordIdx), hipMemcpyHostToDevice, cur_stream(ctx)));
      PG_HIP(ctx, hipMemsetAsync(A->fr_nlist, 0, 4, cur_stream(ctx)));
      hipLaunchKernelGGL(anib_failed_kernel, dim3((uint32_t)((slots + 255) / 256)), dim3(256), 0, cur_stream(ctx), A->fr_pairs, A->fr_slot_pair,
                         (uint32_t)slots, A->fr_rows, A->fr_nrows, A->fr_out, A->fr_widx, A->units_d, A->fr_list, A->fr_nlist);
      uint32_t n_list = 0;
      PG_HIP(ctx, hipMemcpyAsync(&n_list, A->fr_nlist, 4, hipMemcpyDeviceToHost, cur_stream(ctx)));
      PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
      if (n_list) {
        pg_prof_begin(ctx, PG_K_ANIB_FRAG);
        hipLaunchKernelGGL(anib_frag_kernel, dim3(n_list), dim3(64), 0, cur_stream(ctx), A->refs_d, A->units_d, A->fr_pairs, A->fr_slot_pair,
                           A->fr_off, A->fr_entries, A->fr_ebase, A->fr_rows, A->fr_nrows, (const uint32_t*)A->fr_list, (const WordIdx*)A->fr_widx);
        pg_prof_end(ctx);
        hipLaunchKernelGGL(anib_reduce_pairs_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, cur_stream(ctx), A->fr_pairs, n_pairs, A->fr_rows,
                           A->fr_nrows, A->fr_out);
        PG_HIP(ctx, hipGetLastError());
        PG_HIP(ctx, hipMemcpyAsync(F.out, A->fr_out, n_pairs * sizeof(pg_anib_result), hipMemcpyDeviceToHost, cur_stream(ctx)));
        PG_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
      }
    }
  }
  if (F.n_rows_out) {   // the table of pair 0
    const uint32_t nf = (uint32_t)fp[0].n_frags;
    std::vector<uint32_t> nr(nf);
    std::vector<FragRow> rows((size_t)nf * FRAG_ROWS);
    if (nf) {
      PG_HIP(ctx, hipMemcpy(nr.data(), A->fr_nrows + fp[0].slot0, nf * 4, hipMemcpyDeviceToHost));
      PG_HIP(ctx, hipMemcpy(rows.data(), A->fr_rows + (size_t)fp[0].slot0 * FRAG_ROWS, rows.size() * sizeof(FragRow), hipMemcpyDeviceToHost));
    }
    uint32_t n = 0;
    for (uint32_t f = 0; f < nf; ++f)
      for (uint32_t i = 0; i < nr[f]; ++i, ++n)
        if (F.rows_out && n < F.rows_cap) F.rows_out[n] = *reinterpret_cast<const pg_anib_row*>(&rows[(size_t)f * FRAG_ROWS + i]);
    *F.n_rows_out = n;
  }
  return PG_OK;
}

// The alignment records of the pair a 1-pair batch has just processed (slice 0 of the finish scratch), converted to
// MUMmer's per-record 1-based closed coordinates.
int pg_anim_fetch_alignments(pg_ctx* ctx, int32_t ref_id, int32_t qry_id, uint32_t n, pg_anim_alignment* out) {
  AnimScratch* A = anim_scratch(ctx);
  std::vector<Aln> al(n);
  std::vector<int32_t> rr(n), qr(n);
  if (n) {
    PG_HIP(ctx, hipMemcpy(al.data(), A->S.alns, n * sizeof(Aln), hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(rr.data(), A->S.a_rrec, n * 4, hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(qr.data(), A->S.a_qrec, n * 4, hipMemcpyDeviceToHost));
  }
  const PgGenome& G = ctx->genomes[ref_id];
  const PgGenome& H = ctx->genomes[qry_id];
  for (uint32_t i = 0; i < n; ++i) {
    const Aln& a = al[i];            // forward stream coordinates, half-open
    const int32_t ro = G.rec_start[rr[i]], qo = H.rec_start[qr[i]];
    pg_anim_alignment x;
    x.ref_rec = rr[i]; x.qry_rec = qr[i];
    x.rs = a.rs - ro + 1; x.re = a.re - ro;
    x.qs = a.strand ? a.qe - qo : a.qs - qo + 1;
    x.qe = a.strand ? a.qs - qo + 1 : a.qe - qo;
    x.errors = a.errors; x.kept = a.keep;
    out[i] = x;
  }
  return PG_OK;
}

int pg_anim_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                       const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                       int apply_filter, pg_anim_result* out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<Aln> h(n);
  for (uint64_t i = 0; i < n; ++i) {
    Aln a;
    a.strand = qs[i] > qe[i];
    a.rs = (rs[i] < re[i] ? rs[i] : re[i]) - 1; a.re = rs[i] < re[i] ? re[i] : rs[i];
    a.qs = (qs[i] < qe[i] ? qs[i] : qe[i]) - 1; a.qe = qs[i] < qe[i] ? qe[i] : qs[i];
    a.errors = errors[i];
    a.keep = apply_filter ? 0 : 3;
    h[i] = a;
  }
  uint64_t* d_off = nullptr; Aln* d_a = nullptr; int32_t *d_rg = nullptr, *d_qg = nullptr, *d_idx = nullptr, *d_from = nullptr;
  double* d_sc = nullptr; pg_anim_result* d_out = nullptr;
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* p : to_free) if (p) (void)hipFree(p); };
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_a, n + 1); AA(d_rg, n + 1); AA(d_qg, n + 1); AA(d_idx, n + 1); AA(d_from, n + 1); AA(d_sc, n + 1);
  AA(d_out, n_pairs + 1);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_a, h.data(), n * sizeof(Aln), hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_rg, rseq, n * 4, hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_qg, qseq, n * 4, hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anim_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, cur_stream(ctx), n_pairs, d_off, d_a, d_rg, d_qg,
                       d_idx, d_from, d_sc, apply_filter, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, cur_stream(ctx));
  if (e == hipSuccess) e = hipStreamSynchronize(cur_stream(ctx));
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anim reduce: ") + hipGetErrorString(e));
  return PG_OK;
}

// ---- ANIb: parse_blast_tab reduction (pyani/anib.py:641-665), one thread per ordered pair ---------------------------
namespace {
__global__ __launch_bounds__(64) void anib_reduce_kernel(uint32_t n_pairs, const uint64_t* __restrict__ offsets,
                                                         const uint64_t* __restrict__ foff, const int32_t* __restrict__ frag,
                                                         const int32_t* __restrict__ length, const int32_t* __restrict__ mismatch,
                                                         const int32_t* __restrict__ gaps, const int32_t* __restrict__ qlen,
                                                         const double* __restrict__ pident, int64_t* first_row,
                                                         int64_t* __restrict__ aln_out, int64_t* __restrict__ err_out,
                                                         double* __restrict__ pid_out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int64_t* first = first_row + foff[p];
  const uint64_t nf = foff[p + 1] - foff[p];
  for (uint64_t f = 0; f < nf; ++f) first[f] = -1;
  for (uint64_t i = offsets[p]; i < offsets[p + 1]; ++i) {
    const int32_t alnlen = length[i] - gaps[i], alnids = alnlen - mismatch[i];
    const double cov = (double)alnlen / (double)qlen[i], pid = (double)alnids / (double)qlen[i];
    if (cov > 0.7 && pid > 0.3 && (uint64_t)frag[i] < nf && first[frag[i]] < 0) first[frag[i]] = (int64_t)i;
  }
  int64_t aln = 0, err = 0, cnt = 0;
  double sum = 0.0;
  for (uint64_t f = 0; f < nf; ++f) {
    const int64_t i = first[f];
    if (i < 0) continue;
    aln += length[i] - gaps[i];
    err += (int64_t)mismatch[i] + gaps[i];
    sum = sum + pident[i];
    ++cnt;
  }
  aln_out[p] = aln;
  err_out[p] = err;
  pid_out[p] = cnt ? sum / (double)cnt : 0.0;
}
}  // namespace

int pg_anib_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                       const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen,
                       const double* pident, int64_t* aln_out, int64_t* err_out, double* pid_out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<uint64_t> foff(n_pairs + 1, 0);
  for (uint32_t p = 0; p < n_pairs; ++p) foff[p + 1] = foff[p] + n_frags[p];
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* q : to_free) if (q) (void)hipFree(q); };
  uint64_t *d_off = nullptr, *d_foff = nullptr;
  int32_t *d_frag = nullptr, *d_len = nullptr, *d_mm = nullptr, *d_gap = nullptr, *d_ql = nullptr;
  double *d_pid = nullptr, *d_pout = nullptr;
  int64_t *d_first = nullptr, *d_aln = nullptr, *d_err = nullptr;
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_foff, n_pairs + 1); AA(d_frag, n + 1); AA(d_len, n + 1); AA(d_mm, n + 1); AA(d_gap, n + 1);
  AA(d_ql, n + 1); AA(d_pid, n + 1); AA(d_first, foff[n_pairs] + 1); AA(d_aln, n_pairs); AA(d_err, n_pairs); AA(d_pout, n_pairs);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess) e = hipMemcpyAsync(d_foff, foff.data(), (n_pairs + 1) * 8, hipMemcpyHostToDevice, cur_stream(ctx));
  const struct { void* d; const void* h; size_t b; } cp[] = {{d_frag, frag, n * 4}, {d_len, length, n * 4}, {d_mm, mismatch, n * 4},
                                                           {d_gap, gaps, n * 4}, {d_ql, qlen, n * 4}, {d_pid, pident, n * 8}};
  for (const auto& c : cp)
    if (e == hipSuccess && c.b) e = hipMemcpyAsync(c.d, c.h, c.b, hipMemcpyHostToDevice, cur_stream(ctx));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anib_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, cur_stream(ctx), n_pairs, d_off, d_foff, d_frag, d_len,
                       d_mm, d_gap, d_ql, d_pid, d_first, d_aln, d_err, d_pout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(aln_out, d_aln, n_pairs * 8, hipMemcpyDeviceToHost, cur_stream(ctx));
  if (e == hipSuccess) e = hipMemcpyAsync(err_out, d_err, n_pairs * 8, hipMemcpyDeviceToHost, cur_stream(ctx));
  if (e == hipSuccess) e = hipMemcpyAsync(pid_out, d_pout, n_pairs * 8, hipMemcpyDeviceToHost, cur_stream(ctx));
  if (e == hipSuccess) e = hipStreamSynchronize(cur_stream(ctx));
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anib reduce: ") + hipGetErrorString(e));
  return PG_OK;
}
