// pg_sketch.hip — the SKETCH mode on the GPU (definition: pg_sketch_core.h; reference interface it stands in for:
// pyani/fastani.py:193-270, construct_fastani_cmdline / parse_fastani_file).  An opt-in ESTIMATE with its own result struct — nothing
// here touches the exact ANIm / ANIb results.  Integer / byte work over the 2-bit packed genomes already resident in HBM; no MFMA.
//
//   S1 sketch_scan_kernel   once per genome (cached with its frag_len / scale): one coalesced pass over the packed stream; every lane
//                           rolls the canonical 16-mers of 32 consecutive start positions out of three code words and two mask
//                           words, keeps the sampled ones (mix32(kmer) & (scale - 1) == 0: 1 in 16), and
//                             - inserts them into the genome's k-mer SET (open addressing in HBM, <= 1/2 load: the reference role),
//                             - appends (k-mer, fragment) for those that lie inside a fragment (the query role), counting per fragment.
//                           Two launches: count (sizes the arrays exactly), then fill.
//   S2 sketch_pairs_kernel  one workgroup per (query genome, up to 4 reference genomes): the query's occurrence list streams through
//                           once (coalesced), every k-mer probes the references' sets (1.2 MB each: L2-resident while the launch
//                           works through one reference's queries), hits are counted per fragment in LDS; then per reference the
//                           fragments' identities (pgs::frag_identity) are summed in fragment order — the definition's order, so the
//                           double comes out bit-identical to the host statement.
// Cost per ordered pair of 5 Mb genomes at scale 16: 3 x 10^5 probes + 2.4 MB / 4 of list traffic: the 10^6 pairs of C4 take seconds.
#include <algorithm>
#include <vector>

#include "pg_internal.h"
#include <mutex>
#include "pg_sketch_core.h"

namespace {

struct SketchGenome {
  bool built = false;
  int32_t frag_len = 0, scale = 0;
  uint32_t n_frags = 0, n_occ = 0, cap_mask = 0;
  uint32_t *occ_kmer = nullptr, *occ_frag = nullptr, *frag_n = nullptr, *tab = nullptr;
  int32_t* rec_tab = nullptr;      // [2 (n_rec + 1)]: rec_start | frag_base (device)
};
struct SketchStore { std::vector<SketchGenome> g; uint32_t* counters = nullptr; };

void free_genome(SketchGenome& S) {
  for (void* p : {(void*)S.occ_kmer, (void*)S.occ_frag, (void*)S.frag_n, (void*)S.tab, (void*)S.rec_tab}) if (p) (void)hipFree(p);
  S = SketchGenome{};
}

// Device memory for a sketch.  The ANIm engine keeps its per-launch scratch and per-genome seed lists for reuse (after a 1000-genome
// grid: ~200 GB of the 288); a sketch that does not fit beside them takes their place — they are rebuilt on the next ANIm call.
template <typename T>
int sk_malloc(pg_ctx* ctx, T*& p, size_t n) {
  // (PYANI_SKETCH_ALLOC_FAIL under PYANI_DEV_KNOBS=1: every first attempt counts as failed, so that a test can walk the fallback)
  static const bool fail_first = pg_dev_env("PYANI_SKETCH_ALLOC_FAIL") != nullptr;
  if (!fail_first && hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)) == hipSuccess) return PG_OK;
  (void)hipGetLastError();
  p = nullptr;
  // The ANIm launch scratch takes the sketches' place — only on an IDLE context: never under an enqueued ANIm call (its thread is using
  // that scratch), and only after everything the worker streams were given has finished.
  {
    std::lock_guard<std::mutex> lk(ctx->anim_async_mu);
    if (ctx->anim_async[0].busy || ctx->anim_async[1].busy)
      return pg_fail(ctx, PG_E_NOMEM, "sketch: device memory is short and enqueued ANIm calls are in flight (fetch them first: their launch scratch cannot be released under them)");
  }
  PG_HIP(ctx, hipDeviceSynchronize());
  pg_anim_free_scratch(ctx);
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
  return PG_OK;
}

// record of stream position p (rec_start[r] <= p < rec_start[r + 1])
__device__ __forceinline__ int rec_of(const int32_t* __restrict__ rec_start, int n_rec, int32_t p) {
  int lo = 0, hi = n_rec - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rec_start[mid] <= p) lo = mid; else hi = mid - 1; }
  return lo;
}

// counters: [0] sampled k-mers (all), [1] occurrences inside fragments (fill pass: the append cursor)
template <bool FILL>
__global__ __launch_bounds__(256) void sketch_scan_kernel(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int64_t stream_len,
                                                           const int32_t* __restrict__ rec_tab, int n_rec, int32_t frag_len, uint32_t scale,
                                                           uint32_t log2_scale, uint32_t* __restrict__ counters, uint32_t* __restrict__ occ_kmer,
                                                           uint32_t* __restrict__ occ_frag, uint32_t* __restrict__ frag_n, uint32_t* __restrict__ tab,
                                                           uint32_t cap_mask) {
  const int32_t* rec_start = rec_tab;
  const int32_t* frag_base = rec_tab + (n_rec + 1);
  const int64_t n_chunks = (stream_len + 31) / 32;      // chunk c = start positions 32 c .. 32 c + 31
  uint32_t n_all = 0, n_in = 0;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
    // bases 32 c .. 32 c + 47 = code words 2 c, 2 c + 1, 2 c + 2; mask words c, c + 1 (the arena is padded: reads past the stream see dirty bases)
    const uint64_t c01 = (uint64_t)codes[2 * c] | ((uint64_t)codes[2 * c + 1] << 32);
    const uint32_t c2 = codes[2 * c + 2];
    const uint64_t m = (uint64_t)mask[c] | ((uint64_t)mask[c + 1] << 32);
    uint32_t f = 0, r = 0;
    int rec = -1;
    int32_t rec_lo = 0, rec_hi = -1, fb = 0, n_full = 0;
    for (int t = 0; t < 47; ++t) {      // base 32 c + t enters the window; the k-mer that ENDS on it starts at 32 c + t - 15
      const uint32_t code = t < 32 ? (uint32_t)(c01 >> (2 * t)) & 3u : (c2 >> (2 * (t - 32))) & 3u;
      f = pgs::roll_fwd(f, code); r = pgs::roll_rc(r, code);
      const int s = t - 15;
      if (s < 0) continue;
      const int64_t p = 32 * c + s;
      if (p + 16 > stream_len || ((m >> s) & 0xFFFFull) != 0xFFFFull) continue;      // an ambiguity symbol, a record end, the stream's end
      const uint32_t canon = f < r ? f : r;
      if (!pgs::sampled(canon, scale)) continue;
      ++n_all;
      if (FILL) {      // the genome's k-mer set
        uint32_t slot = pgs::slot_of(canon, log2_scale, cap_mask);
        for (;;) {
          const uint32_t old = atomicCAS(&tab[slot], pgs::EMPTY, canon);
          if (old == pgs::EMPTY || old == canon) break;
          slot = (slot + 1u) & cap_mask;
        }
      }
      if (p < rec_lo || p > rec_hi) {      // (a chunk of 32 positions rarely leaves its record)
        rec = rec_of(rec_start, n_rec, (int32_t)p);
        rec_lo = rec_start[rec]; rec_hi = rec_start[rec + 1] - 2;      // last base of the record
        fb = frag_base[rec]; n_full = (rec_hi - rec_lo + 1) / frag_len;
      }
      const int32_t x = (int32_t)p - rec_lo, j = x / frag_len;
      if (j >= n_full || x - j * frag_len + 16 > frag_len) continue;      // the record's tail, or a k-mer across two fragments
      ++n_in;
      if (FILL) {
        const uint32_t at = atomicAdd(&counters[1], 1u);
        occ_kmer[at] = canon; occ_frag[at] = (uint32_t)(fb + j);
        atomicAdd(&frag_n[fb + j], 1u);
      }
    }
  }
  if (!FILL) {      // one atomic per wave and counter
    for (int o = 32; o > 0; o >>= 1) { n_all += __shfl_xor(n_all, o, 64); n_in += __shfl_xor(n_in, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (n_all) atomicAdd(&counters[0], n_all); if (n_in) atomicAdd(&counters[1], n_in); }
  }
}

struct SketchJob {      // one workgroup: a query against up to SK_REFS references
  const uint32_t *occ_kmer, *occ_frag, *frag_n;
  uint32_t n_occ, n_frags, n_refs, log2_scale;
  const uint32_t* tab[4];
  uint32_t cap_mask[4];
  uint32_t out[4];      // pair indices of the call
  double min_fraction;
};
constexpr int SK_REFS = 4;

__global__ __launch_bounds__(256) void sketch_pairs_kernel(const SketchJob* __restrict__ jobs, pg_sketch_result* __restrict__ out) {
  extern __shared__ uint32_t hits[];      // [n_refs][n_frags]
  const SketchJob J = jobs[blockIdx.x];
  const uint32_t nf = J.n_frags, nr = J.n_refs;
  for (uint32_t i = threadIdx.x; i < nf * nr; i += blockDim.x) hits[i] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < J.n_occ; i += blockDim.x) {
    const uint32_t km = __builtin_nontemporal_load(J.occ_kmer + i), fr = __builtin_nontemporal_load(J.occ_frag + i);
    const uint32_t h = pgs::mix32(km) >> J.log2_scale;
#pragma unroll
    for (uint32_t g = 0; g < SK_REFS; ++g) {
      if (g >= nr) break;
      const uint32_t* tab = J.tab[g];
      const uint32_t cm = J.cap_mask[g];
      uint32_t slot = h & cm, v;
      while ((v = tab[slot]) != pgs::EMPTY) {
        if (v == km) { atomicAdd(&hits[g * nf + fr], 1u); break; }
        slot = (slot + 1u) & cm;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nr) {      // the definition's order: fragments ascending, one running sum (bit-identical to the host statement)
    const uint32_t g = threadIdx.x;
    double sum = 0.0;
    int32_t matches = 0;
    for (uint32_t f = 0; f < nf; ++f) {
      const uint32_t n = J.frag_n[f], h = hits[g * nf + f];
      if (pgs::frag_matches(h, n)) { sum = sum + pgs::frag_identity(h, n); ++matches; }
    }
    pg_sketch_result o;
    o.matches = matches; o.fragments = (int32_t)nf;
    const bool enough = matches > 0 && (double)matches >= J.min_fraction * (double)nf;
    o.ani = enough ? sum / (double)matches : 0.0;
    o.status = enough ? 0 : PG_SKETCH_NO_RESULT; o.reserved = 0;
    out[J.out[g]] = o;
  }
}

SketchStore* store_of(pg_ctx* ctx) {
  if (!ctx->sketch_store) ctx->sketch_store = new SketchStore();
  return static_cast<SketchStore*>(ctx->sketch_store);
}

int build_sketch(pg_ctx* ctx, SketchStore* ST, int32_t gid, int32_t frag_len, int32_t scale, uint32_t log2_scale) {
  SketchGenome& S = ST->g[gid];
  if (S.built && S.frag_len == frag_len && S.scale == scale) return PG_OK;
  free_genome(S);
  const PgGenome& G = ctx->genomes[gid];
  std::vector<int32_t> rec_tab(2 * (G.n_rec + 1));
  uint32_t nf = 0;
  for (uint32_t r = 0; r <= G.n_rec; ++r) {
    rec_tab[r] = G.rec_start[r];
    rec_tab[G.n_rec + 1 + r] = (int32_t)nf;
    if (r < G.n_rec) nf += (uint32_t)((G.rec_start[r + 1] - 1 - G.rec_start[r]) / frag_len);
  }
  int rc;
  if ((rc = sk_malloc(ctx, S.rec_tab, rec_tab.size()))) return rc;
  PG_HIP(ctx, hipMemcpyAsync(S.rec_tab, rec_tab.data(), rec_tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!ST->counters && (rc = sk_malloc(ctx, ST->counters, 2))) return rc;
  PG_HIP(ctx, hipMemsetAsync(ST->counters, 0, 8, ctx->stream));
  const uint32_t* codes = ctx->d_codes + G.arena_start / 16;
  const uint32_t* mask = ctx->d_mask + G.arena_start / 32;
  const dim3 grid((uint32_t)std::min<uint64_t>((G.stream_len / 32 + 255) / 256 + 1, (uint64_t)ctx->num_cu * 8));
  hipLaunchKernelGGL((sketch_scan_kernel<false>), grid, dim3(256), 0, ctx->stream, codes, mask, (int64_t)G.stream_len, S.rec_tab, (int)G.n_rec, frag_len,
                     (uint32_t)scale, log2_scale, ST->counters, nullptr, nullptr, nullptr, nullptr, 0u);
  uint32_t cnt[2];
  PG_HIP(ctx, hipMemcpyAsync(cnt, ST->counters, 8, hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t cap = 1024;
  while (cap < 2 * cnt[0]) cap <<= 1;
  S.cap_mask = cap - 1; S.n_occ = cnt[1]; S.n_frags = nf; S.frag_len = frag_len; S.scale = scale;
  if ((rc = sk_malloc(ctx, S.tab, (size_t)cap))) return rc;
  if ((rc = sk_malloc(ctx, S.occ_kmer, (size_t)cnt[1] + 1))) return rc;
  if ((rc = sk_malloc(ctx, S.occ_frag, (size_t)cnt[1] + 1))) return rc;
  if ((rc = sk_malloc(ctx, S.frag_n, (size_t)nf + 1))) return rc;
  PG_HIP(ctx, hipMemsetAsync(S.tab, 0xFF, (size_t)cap * 4, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(S.frag_n, 0, (size_t)(nf + 1) * 4, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(ST->counters, 0, 8, ctx->stream));
  hipLaunchKernelGGL((sketch_scan_kernel<true>), grid, dim3(256), 0, ctx->stream, codes, mask, (int64_t)G.stream_len, S.rec_tab, (int)G.n_rec, frag_len,
                     (uint32_t)scale, log2_scale, ST->counters, S.occ_kmer, S.occ_frag, S.frag_n, S.tab, S.cap_mask);
  PG_HIP(ctx, hipGetLastError());
  S.built = true;
  return PG_OK;
}

}  // namespace

void pg_sketch_drop(pg_ctx* ctx) {
  if (!ctx->sketch_store) return;
  SketchStore* ST = static_cast<SketchStore*>(ctx->sketch_store);
  for (auto& s : ST->g) free_genome(s);
  if (ST->counters) (void)hipFree(ST->counters);
  delete ST;
  ctx->sketch_store = nullptr;
}

extern "C" int pg_sketch_pairs(pg_ctx* ctx, const int32_t* qry_ids, const int32_t* ref_ids, uint64_t n_pairs, int32_t frag_len, int32_t scale,
                               double min_fraction, pg_sketch_result* out) {
  if (!ctx || !out || (n_pairs && (!qry_ids || !ref_ids))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (frag_len < 64 || scale < 1 || scale > 4096 || (scale & (scale - 1)) || !(min_fraction >= 0.0 && min_fraction <= 1.0))
    return pg_fail(ctx, PG_E_ARG, "sketch: frag_len >= 64, scale a power of two <= 4096, 0 <= min_fraction <= 1");
  for (uint64_t i = 0; i < n_pairs; ++i)
    if (qry_ids[i] < 0 || (size_t)qry_ids[i] >= ctx->genomes.size() || ref_ids[i] < 0 || (size_t)ref_ids[i] >= ctx->genomes.size())
      return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  if (n_pairs == 0) return PG_OK;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  SketchStore* ST = store_of(ctx);
  if (ST->g.size() < ctx->genomes.size()) ST->g.resize(ctx->genomes.size());
  uint32_t log2_scale = 0;
  while ((1 << log2_scale) < scale) ++log2_scale;
  std::vector<char> need(ctx->genomes.size(), 0);
  for (uint64_t i = 0; i < n_pairs; ++i) { need[qry_ids[i]] = 1; need[ref_ids[i]] = 1; }
  for (size_t g = 0; g < need.size(); ++g)
    if (need[g] && (rc = build_sketch(ctx, ST, (int32_t)g, frag_len, scale, log2_scale))) return rc;
  // jobs: the pairs by query, up to SK_REFS references per workgroup (fewer when the query's fragment counters would not fit LDS)
  std::vector<uint64_t> idx(n_pairs);
  for (uint64_t i = 0; i < n_pairs; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return qry_ids[a] != qry_ids[b] ? qry_ids[a] < qry_ids[b] : ref_ids[a] < ref_ids[b]; });
  std::vector<SketchJob> jobs;
  size_t lds_max = 0;
  for (uint64_t a = 0; a < n_pairs;) {
    const SketchGenome& Q = ST->g[qry_ids[idx[a]]];
    const size_t per_ref = (size_t)std::max<uint32_t>(Q.n_frags, 1u) * 4;
    if (per_ref > 96 * 1024) return pg_fail(ctx, PG_E_CAPACITY, "sketch: more than 24 576 fragments in one query genome");
    const uint32_t g_max = (uint32_t)std::min<size_t>(SK_REFS, (96 * 1024) / per_ref);
    SketchJob J{};
    J.occ_kmer = Q.occ_kmer; J.occ_frag = Q.occ_frag; J.frag_n = Q.frag_n; J.n_occ = Q.n_occ; J.n_frags = Q.n_frags; J.log2_scale = log2_scale;
    J.min_fraction = min_fraction;
    uint32_t g = 0;
    while (a < n_pairs && g < g_max && qry_ids[idx[a]] == qry_ids[idx[a - g]]) {
      const SketchGenome& Rf = ST->g[ref_ids[idx[a]]];
      J.tab[g] = Rf.tab; J.cap_mask[g] = Rf.cap_mask; J.out[g] = (uint32_t)idx[a];
      ++g; ++a;
    }
    J.n_refs = g;
    lds_max = std::max(lds_max, per_ref * g);
    jobs.push_back(J);
  }
  SketchJob* d_jobs = nullptr;
  pg_sketch_result* d_out = nullptr;
  struct Guard { SketchJob*& j; pg_sketch_result*& o; ~Guard() { if (j) (void)hipFree(j); if (o) (void)hipFree(o); } } guard{d_jobs, d_out};   // every exit path
  if ((rc = sk_malloc(ctx, d_jobs, jobs.size()))) return rc;
  if ((rc = sk_malloc(ctx, d_out, (size_t)n_pairs))) return rc;
  PG_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(SketchJob), hipMemcpyHostToDevice, ctx->stream));
  if (lds_max > 48 * 1024) {      // up to 96 KiB of dynamic LDS: fits gfx950's 160 KiB per workgroup; a part that refuses it gets a clear status, not a launch failure
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sketch_pairs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024)) != hipSuccess) {
      (void)hipGetLastError();
      return pg_fail(ctx, PG_E_CAPACITY, "sketch: the job's fragment counters need up to 96 KiB of LDS per workgroup, which this device does not grant");
    }
  }
  pg_prof_begin(ctx, PG_K_SKETCH_PAIRS);
  hipLaunchKernelGGL(sketch_pairs_kernel, dim3((uint32_t)jobs.size()), dim3(256), std::max<size_t>(lds_max, 16), ctx->stream, d_jobs, d_out);
  pg_prof_end(ctx);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n_pairs * sizeof(pg_sketch_result), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("sketch: ") + hipGetErrorString(e));
  return PG_OK;
}
