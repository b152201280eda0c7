// pg_api.cpp — C ABI of libpyani_gpu.so: context, genome store (FASTA -> 2-bit/1-bit packed arena in HBM) and the
// TETRA entry points declared in include/pyani_gpu.h.  Host-side C++ only; kernels live in pg_tetra.hip.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <functional>
#include <new>
#include <thread>

#include "pg_internal.h"

int pg_fail(pg_ctx* ctx, int code, const std::string& msg) {
  if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = msg; }
  return code;
}

static const char* const KERNEL_NAMES[PG_K__COUNT] = {"tetra_count_kernel", "tetra_finalize_kernel", "tetra_stats_kernel",
                                                      "tetra_pairs_kernel", "anim_seed_kernel", "anim_hit_kernels",
                                                      "anim_cluster_wave_kernel",
                                                      "anim_postnuc_gap_kernels", "anim_postnuc_forced_kernels",
                                                      "anim_postnuc_kernel", "anim_finish_kernel", "anib_bucket_kernel",
                                                      "anib_frag_kernel", "anim_postnuc_fwd_kernel", "anim_postnuc_rehearse_kernel+anim_postnuc_bwd_kernel",
                                                      "sketch_pairs_kernel"};

// ---- profiling ----------------------------------------------------------------------------------------------
thread_local hipStream_t pg_tls_stream = nullptr;
namespace { thread_local PgEventPair tls_open_pair; thread_local bool tls_open = false; }
void pg_prof_begin(pg_ctx* ctx, int which) {
  tls_open = false;
  if (!ctx->profiling || !((ctx->prof_mask >> which) & 1u)) return;
  {
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    if ((ctx->prof_seen[which]++ % ctx->prof_every) != 0) return;
  }
  PgEventPair p;
  p.which = which;
  if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
  (void)hipEventRecord(p.a, pg_tls_stream ? pg_tls_stream : ctx->stream);
  tls_open_pair = p;
  tls_open = true;
}
void pg_prof_end(pg_ctx* ctx) {
  if (!tls_open) return;
  tls_open = false;
  (void)hipEventRecord(tls_open_pair.b, pg_tls_stream ? pg_tls_stream : ctx->stream);
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  ctx->events.push_back(tls_open_pair);
}
static void prof_drain(pg_ctx* ctx) {
  for (auto& p : ctx->events) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->prof_ms[p.which] += ms;
      ctx->prof_n[p.which] += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  ctx->events.clear();
}

// ---- packing --------------------------------------------------------------------------------------------------
namespace {

struct Lut {
  uint8_t v[256];
  Lut() {
    std::memset(v, 4, sizeof(v));
    v['A'] = v['a'] = 0;
    v['C'] = v['c'] = 1;
    v['G'] = v['g'] = 2;
    v['T'] = v['t'] = 3;
    v['U'] = v['u'] = 5;  // flagged: Biopython's reverse complement maps U->A (asymmetric), unsupported
  }
};
const Lut LUT;

inline uint32_t rc4_index(uint32_t x) {
  uint32_t c = 255u - x, r = 0;
  for (int i = 0; i < 4; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
  return r;
}

// Build the packed stream of one genome.  Returns PG_OK or PG_E_RNA.
int pack_genome(const uint8_t* seq, const uint64_t* rec_off, uint32_t n_rec, PgGenome& g) {
  g.n_rec = n_rec;
  g.total_len = n_rec ? rec_off[n_rec] - rec_off[0] : 0;
  g.stream_len = g.total_len + (n_rec > 1 ? n_rec - 1 : 0);
  g.padded_len = ((g.stream_len + 1 + PG_SUPER - 1) / PG_SUPER) * PG_SUPER;  // >= 1 dirty base at the end
  g.codes.assign(g.padded_len / 16, 0u);
  g.mask.assign(g.padded_len / 32, 0u);
  g.quirk.fill(0);
  g.rec_start.clear();
  uint64_t s = 0;  // stream position
  bool has_u = false;
  for (uint32_t r = 0; r < n_rec; ++r) {
    const uint8_t* p = seq + rec_off[r];
    const uint64_t L = rec_off[r + 1] - rec_off[r];
    if (r > 0) ++s;  // separator stays dirty (zero-initialised)
    g.rec_start.push_back((int32_t)s);
    uint64_t i = 0;
    // head: until s is 32-aligned
    auto put = [&](uint64_t pos, uint8_t c) {
      if (c < 4) {
        g.codes[pos >> 4] |= (uint32_t)c << (2 * (pos & 15));
        g.mask[pos >> 5] |= 1u << (pos & 31);
      } else if (c == 5) {
        has_u = true;
      }
    };
    while (i < L && (s & 31)) put(s++, LUT.v[p[i++]]);
    // body: 32 bases -> two code words + one mask word
    while (i + 32 <= L) {
      uint64_t cw = 0;
      uint32_t mw = 0;
      for (int k = 0; k < 32; ++k) {
        const uint8_t c = LUT.v[p[i + k]];
        const uint32_t ok = c < 4;
        cw |= (uint64_t)(c & 3u & (0u - ok)) << (2 * k);
        mw |= ok << k;
        has_u |= (c == 5);
      }
      g.codes[s >> 4] = (uint32_t)cw;
      g.codes[(s >> 4) + 1] = (uint32_t)(cw >> 32);
      g.mask[s >> 5] = mw;
      s += 32;
      i += 32;
    }
    while (i < L) put(s++, LUT.v[p[i++]]);
    // the reference never counts the LAST tetranucleotide of either strand of a record (tetra.py:106):
    // forward strand: last4(rec); reverse strand: its last window is rc(first4(rec)).
    if (L >= 4) {
      uint32_t f = 0, l = 0;
      bool fok = true, lok = true;
      for (int k = 0; k < 4; ++k) {
        const uint8_t cf = LUT.v[p[k]], cl = LUT.v[p[L - 4 + k]];
        fok = fok && cf < 4;
        lok = lok && cl < 4;
        f = f * 4 + (cf & 3u);
        l = l * 4 + (cl & 3u);
      }
      if (lok) g.quirk[l] += 1;
      if (fok) g.quirk[rc4_index(f)] += 1;
    }
  }
  g.rec_start.push_back((int32_t)g.stream_len + 1);
  return has_u ? PG_E_RNA : PG_OK;
}

int read_file(const char* path, std::vector<uint8_t>& buf) {
  FILE* fh = std::fopen(path, "rb");
  if (!fh) return -1;
  std::fseek(fh, 0, SEEK_END);
  const long sz = std::ftell(fh);
  std::fseek(fh, 0, SEEK_SET);
  if (sz < 0) { std::fclose(fh); return -1; }
  buf.resize((size_t)sz);
  const size_t got = sz ? std::fread(buf.data(), 1, (size_t)sz, fh) : 0;
  std::fclose(fh);
  return got == (size_t)sz ? 0 : -1;
}

// FASTA text -> concatenated record sequences + offsets.  Record = '>' line + following lines; lines before the
// first '>' are ignored; spaces, CR and line breaks are removed from sequence lines, trailing blanks stripped.
void parse_fasta(const std::vector<uint8_t>& txt, std::vector<uint8_t>& seq, std::vector<uint64_t>& off) {
  seq.clear();
  off.clear();
  seq.reserve(txt.size());
  size_t i = 0;
  const size_t n = txt.size();
  bool in_rec = false;
  while (i < n) {
    size_t e = i;
    while (e < n && txt[e] != '\n') ++e;
    if (txt[i] == '>') {
      off.push_back(seq.size());
      in_rec = true;
    } else if (in_rec) {
      size_t end = e;
      while (end > i && (txt[end - 1] == ' ' || txt[end - 1] == '\t' || txt[end - 1] == '\r' || txt[end - 1] == '\v' ||
                         txt[end - 1] == '\f'))
        --end;
      for (size_t k = i; k < end; ++k)
        if (txt[k] != ' ' && txt[k] != '\r') seq.push_back(txt[k]);
    }
    i = e + 1;
  }
  off.push_back(seq.size());
  if (off.size() == 1) off.clear();  // no records at all
}

int add_packed(pg_ctx* ctx, PgGenome&& g, int32_t* id_out) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  g.arena_start = ctx->arena_used;
  ctx->arena_used += g.padded_len;
  ctx->genomes.push_back(std::move(g));
  if (id_out) *id_out = (int32_t)ctx->genomes.size() - 1;
  return PG_OK;
}

template <typename T>
int dev_realloc(pg_ctx* ctx, T*& p, size_t n_new) {
  if (p) PG_HIP(ctx, hipFree(p));
  p = nullptr;
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), n_new * sizeof(T)));
  return PG_OK;
}
template <typename T>
int host_realloc(pg_ctx* ctx, T*& p, size_t n_new) {
  if (p) PG_HIP(ctx, hipHostFree(p));
  p = nullptr;
  PG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&p), n_new * sizeof(T), hipHostMallocDefault));
  return PG_OK;
}

int ensure_batch_scratch(pg_ctx* ctx, uint32_t n) {
  if (n > ctx->batch_cap) {
    const uint32_t cap = std::max<uint32_t>(n, ctx->batch_cap * 2);
    int rc;
    if ((rc = dev_realloc(ctx, ctx->d_batch_gid, cap))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_seg_tile0, cap))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_seg_prefix, (size_t)cap + 1))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_acc, (size_t)cap * PG_ACC_WORDS))) return rc;
    PG_HIP(ctx, hipMemsetAsync(ctx->d_acc, 0, (size_t)cap * PG_ACC_WORDS * 8, ctx->stream));
    if ((rc = dev_realloc(ctx, ctx->d_counts, (size_t)cap * PG_ACC_WORDS))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_dev, (size_t)cap * 256))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_ss, (size_t)cap))) return rc;
    if ((rc = dev_realloc(ctx, ctx->d_keybits, (size_t)cap * 4))) return rc;
    ctx->batch_cap = cap;
    ctx->batch_ids.clear();
  }
  if (n > ctx->h_batch_cap) {
    const uint32_t cap = std::max<uint32_t>(n, ctx->h_batch_cap * 2);
    int rc;
    if ((rc = host_realloc(ctx, ctx->h_counts, (size_t)cap * PG_ACC_WORDS))) return rc;
    ctx->h_batch_cap = cap;
  }
  return PG_OK;
}

// result block layout for a pass over n genomes
struct ResultLayout {
  uint64_t off_z, off_corr, off_present, bytes, bytes_corr_only;
  explicit ResultLayout(uint32_t n, bool corr) {
    off_corr = 16;
    off_z = off_corr + (corr ? (uint64_t)n * n * 8 : 0);
    off_present = off_z + (uint64_t)n * 256 * 8;
    bytes = off_present + (uint64_t)n * 256;
    bytes_corr_only = off_z;
  }
};

int ensure_result(pg_ctx* ctx, uint32_t n, bool corr) {
  const ResultLayout L(n, corr);
  if (L.bytes > ctx->result_cap) {
    const uint64_t cap = std::max<uint64_t>(L.bytes, ctx->result_cap + ctx->result_cap / 2);
    int rc;
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = dev_realloc(ctx, ctx->d_result, cap))) return rc;
    if ((rc = host_realloc(ctx, ctx->h_result, cap))) return rc;
    PG_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, 16, ctx->stream));
    ctx->result_cap = cap;
  }
  ctx->d_flags = reinterpret_cast<int32_t*>(ctx->d_result);
  ctx->d_z = reinterpret_cast<double*>(ctx->d_result + L.off_z);
  ctx->d_corr = reinterpret_cast<double*>(ctx->d_result + L.off_corr);
  ctx->d_present = ctx->d_result + L.off_present;
  return PG_OK;
}

// Make `ids` the current batch: genomes resident, work list (super-tile -> batch row) on the device.
int ensure_batch(pg_ctx* ctx, const int32_t* ids, uint32_t n) {
  int rc;
  for (uint32_t i = 0; i < n; ++i)
    if (ids[i] < 0 || (size_t)ids[i] >= ctx->genomes.size()) return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  if ((rc = pg_upload(ctx))) return rc;
  if ((rc = ensure_batch_scratch(ctx, n))) return rc;
  if (ctx->batch_ids.size() == n && std::equal(ids, ids + n, ctx->batch_ids.begin()) && n > 0) return PG_OK;
  std::vector<uint32_t> tile0(n), prefix(n + 1, 0), gid(n);
  for (uint32_t b = 0; b < n; ++b) {
    const PgGenome& g = ctx->genomes[ids[b]];
    gid[b] = (uint32_t)ids[b];
    tile0[b] = (uint32_t)(g.arena_start / PG_SUPER);
    prefix[b + 1] = prefix[b] + (uint32_t)(g.padded_len / PG_SUPER);
  }
  ctx->n_work = prefix[n];
  if (n) {
    PG_HIP(ctx, hipMemcpyAsync(ctx->d_seg_tile0, tile0.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(ctx->d_batch_gid, gid.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  PG_HIP(ctx, hipMemcpyAsync(ctx->d_seg_prefix, prefix.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
  ctx->batch_ids.assign(ids, ids + n);
  return PG_OK;
}

int check_flags(pg_ctx* ctx, uint32_t n) {
  const int32_t* h_flags = reinterpret_cast<const int32_t*>(ctx->h_result);
  if (n >= 2) {
    if (h_flags[0] & 1) return pg_fail(ctx, PG_E_KEYSET, "genomes have different observed-tetranucleotide key sets");
    if (h_flags[1] == 0) return pg_fail(ctx, PG_E_EMPTY, "no tetranucleotide observed in any genome");
  }
  return PG_OK;
}

}  // namespace

// ---- context ------------------------------------------------------------------------------------------------
extern "C" {

const char* pg_version(void) { return "pyani_gpu 0.1.0 (gfx950)"; }

const char* pg_kernel_name(int which) { return (which >= 0 && which < PG_K__COUNT) ? KERNEL_NAMES[which] : ""; }

int pg_create(pg_ctx** out, int device) {
  if (!out) return PG_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PG_E_NODEVICE;
  if (device < 0 || device >= ndev) return PG_E_ARG;
  pg_ctx* ctx = new (std::nothrow) pg_ctx();
  if (!ctx) return PG_E_NOMEM;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return PG_E_HIP;
  }
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w)
    if (hipStreamCreateWithFlags(&ctx->stream_w[w], hipStreamNonBlocking) != hipSuccess) { delete ctx; return PG_E_HIP; }
  if (const char* g = pg_dev_env("PYANI_ANIM_BWD_AHEAD")) ctx->anim_bwd_ahead = atoi(g) != 0;
  if (const char* g = pg_dev_env("PYANI_PN_WINDOW_MAX")) ctx->anim_pn_window_max = atoi(g);   // development switches of the forced kernels (tests: same results)
  if (const char* g = pg_dev_env("PYANI_PN_GROUP_MAX")) ctx->anim_pn_group_max = atoi(g);
  if (const char* g = pg_dev_env("PYANI_ANIM_GAP_LANES")) ctx->anim_gap_lanes = atoi(g) != 0;   // development switch (tests: both forms, same results)
  if (const char* w = pg_dev_env("PYANI_ANIM_WORKERS")) {   // development switch
    ctx->anim_workers = atoi(w);
    if (ctx->anim_workers < 1) ctx->anim_workers = 1;
    if (ctx->anim_workers > pg_ctx::MAX_WORKERS) ctx->anim_workers = pg_ctx::MAX_WORKERS;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    ctx->num_cu = prop.multiProcessorCount;
  (void)hipGetLastError();   // the runtime's "last error" is per thread and sticky: start clean
  *out = ctx;
  return PG_OK;
}

void pg_destroy(pg_ctx* ctx) {
  if (!ctx) return;
  for (auto& J : ctx->anim_async) if (J.th.joinable()) J.th.join();      // enqueued ANIm calls nobody fetched
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) (void)hipStreamSynchronize(ctx->stream_w[w]);
  pg_anim_free_scratch(ctx);
  pg_sketch_drop(ctx);
  prof_drain(ctx);
  void* dev[] = {ctx->d_codes, ctx->d_mask, ctx->d_quirk, ctx->d_seg_tile0, ctx->d_seg_prefix, ctx->d_batch_gid, ctx->d_acc,
                 ctx->d_counts, ctx->d_z, ctx->d_present, ctx->d_dev, ctx->d_ss, ctx->d_flags, ctx->d_corr};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  void* host[] = {ctx->h_result, ctx->h_counts};
  for (void* p : host)
    if (p) (void)hipHostFree(p);
  (void)hipStreamDestroy(ctx->stream);
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) (void)hipStreamDestroy(ctx->stream_w[w]);
  (void)hipGetLastError();   // teardown errors must not surface in a later context's launch checks
  delete ctx;
}

const char* pg_last_error(const pg_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pg_sync(pg_ctx* ctx) {
  if (!ctx) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int w = 1; w < pg_ctx::MAX_WORKERS; ++w) PG_HIP(ctx, hipStreamSynchronize(ctx->stream_w[w]));
  return PG_OK;
}

// ---- genome store -------------------------------------------------------------------------------------------
int pg_add_genome(pg_ctx* ctx, const uint8_t* seq, const uint64_t* rec_off, uint32_t n_rec, int32_t* genome_id_out) {
  if (!ctx || !rec_off || (!seq && n_rec && rec_off[n_rec] > rec_off[0])) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (uint32_t r = 0; r < n_rec; ++r)
    if (rec_off[r + 1] < rec_off[r]) return pg_fail(ctx, PG_E_ARG, "record offsets must be non-decreasing");
  PgGenome g;
  const int rc = pack_genome(seq, rec_off, n_rec, g);
  if (rc == PG_E_RNA)
    return pg_fail(ctx, PG_E_RNA, "sequence contains U/u (RNA); Biopython complements it asymmetrically, unsupported");
  return add_packed(ctx, std::move(g), genome_id_out);
}

int pg_add_fasta(pg_ctx* ctx, const char* path, int32_t* genome_id_out, uint64_t* total_len_out, uint32_t* n_rec_out) {
  if (!ctx || !path) return pg_fail(ctx, PG_E_ARG, "bad argument");
  std::vector<uint8_t> txt, seq;
  std::vector<uint64_t> off;
  if (read_file(path, txt) != 0) return pg_fail(ctx, PG_E_IO, std::string("cannot read ") + path);
  parse_fasta(txt, seq, off);
  const uint64_t zero_off[1] = {0};
  const uint32_t n_rec = off.empty() ? 0 : (uint32_t)off.size() - 1;
  PgGenome g;
  const int rc = pack_genome(seq.data(), off.empty() ? zero_off : off.data(), n_rec, g);
  if (rc == PG_E_RNA) return pg_fail(ctx, PG_E_RNA, std::string(path) + ": sequence contains U/u (RNA), unsupported");
  if (total_len_out) *total_len_out = g.total_len;
  if (n_rec_out) *n_rec_out = n_rec;
  return add_packed(ctx, std::move(g), genome_id_out);
}

// Host ingest: read + parse + 2-bit pack of many FASTA files on `threads` host threads; genome ids are
// assigned in input order whatever the completion order.  On error nothing is added.
int pg_add_fasta_batch(pg_ctx* ctx, const char* const* paths, uint32_t n, uint32_t threads, int32_t* genome_ids_out,
                       uint64_t* total_len_out, uint32_t* n_rec_out) {
  if (!ctx || (n && !paths)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (uint32_t i = 0; i < n; ++i) if (!paths[i]) return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (threads == 0) threads = std::thread::hardware_concurrency();
  if (threads == 0) threads = 1;
  if (threads > n) threads = n ? n : 1;
  std::vector<PgGenome> packed(n);
  std::vector<int> status(n, PG_OK);
  std::vector<uint32_t> nrec(n, 0);
  std::atomic<uint32_t> next{0};
  auto worker = [&]() {
    std::vector<uint8_t> txt, seq;
    std::vector<uint64_t> off;
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= n) break;
      txt.clear(); seq.clear(); off.clear();
      if (read_file(paths[i], txt) != 0) { status[i] = PG_E_IO; continue; }
      parse_fasta(txt, seq, off);
      const uint64_t zero_off[1] = {0};
      nrec[i] = off.empty() ? 0 : (uint32_t)off.size() - 1;
      status[i] = pack_genome(seq.data(), off.empty() ? zero_off : off.data(), nrec[i], packed[i]);
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  for (uint32_t i = 0; i < n; ++i) {
    if (status[i] == PG_E_IO) return pg_fail(ctx, PG_E_IO, std::string("cannot read ") + paths[i]);
    if (status[i] == PG_E_RNA) return pg_fail(ctx, PG_E_RNA, std::string(paths[i]) + ": sequence contains U/u (RNA), unsupported");
    if (status[i] != PG_OK) return pg_fail(ctx, status[i], std::string(paths[i]) + ": cannot pack");
  }
  for (uint32_t i = 0; i < n; ++i) {
    if (total_len_out) total_len_out[i] = packed[i].total_len;
    if (n_rec_out) n_rec_out[i] = nrec[i];
    int32_t id = -1;
    const int rc = add_packed(ctx, std::move(packed[i]), &id);
    if (rc != PG_OK) return rc;
    if (genome_ids_out) genome_ids_out[i] = id;
  }
  return PG_OK;
}

int pg_genome_count(const pg_ctx* ctx) { return ctx ? (int)ctx->genomes.size() : PG_E_ARG; }

int pg_genome_length(const pg_ctx* ctx, int32_t id, uint64_t* total_len_out, uint32_t* n_rec_out) {
  if (!ctx || id < 0 || (size_t)id >= ctx->genomes.size()) return PG_E_ARG;
  if (total_len_out) *total_len_out = ctx->genomes[id].total_len;
  if (n_rec_out) *n_rec_out = ctx->genomes[id].n_rec;
  return PG_OK;
}

int pg_clear_genomes(pg_ctx* ctx) {
  if (!ctx) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pg_anim_drop_lists(ctx);
  pg_sketch_drop(ctx);
  ctx->genomes.clear();
  ctx->arena_used = 0;
  ctx->n_resident = 0;
  ctx->batch_ids.clear();
  return PG_OK;
}

int pg_upload(pg_ctx* ctx) {
  if (!ctx) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t ng = (uint32_t)ctx->genomes.size();
  if (ctx->n_resident == ng) return PG_OK;
  // arena (+ one zero guard super-tile so look-ahead loads past the last genome stay in bounds and read "dirty")
  if (ctx->arena_used > ctx->arena_cap) {
    const uint64_t cap = std::max<uint64_t>(ctx->arena_used, ctx->arena_cap + ctx->arena_cap / 2);
    uint32_t *nc = nullptr, *nm = nullptr;
    PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&nc), (cap + PG_SUPER) / 4));
    PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&nm), (cap + PG_SUPER) / 8));
    PG_HIP(ctx, hipMemsetAsync(nc, 0, (cap + PG_SUPER) / 4, ctx->stream));
    PG_HIP(ctx, hipMemsetAsync(nm, 0, (cap + PG_SUPER) / 8, ctx->stream));
    uint64_t res_bases = 0;
    for (uint32_t i = 0; i < ctx->n_resident; ++i) res_bases = ctx->genomes[i].arena_start + ctx->genomes[i].padded_len;
    if (res_bases && ctx->d_codes) {
      PG_HIP(ctx, hipMemcpyAsync(nc, ctx->d_codes, res_bases / 4, hipMemcpyDeviceToDevice, ctx->stream));
      PG_HIP(ctx, hipMemcpyAsync(nm, ctx->d_mask, res_bases / 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_codes) PG_HIP(ctx, hipFree(ctx->d_codes));
    if (ctx->d_mask) PG_HIP(ctx, hipFree(ctx->d_mask));
    ctx->d_codes = nc;
    ctx->d_mask = nm;
    ctx->arena_cap = cap;
  }
  if (ng > ctx->quirk_cap) {
    const uint32_t cap = std::max<uint32_t>(ng, ctx->quirk_cap * 2);
    uint32_t* nq = nullptr;
    PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&nq), (size_t)cap * 256 * 4));
    if (ctx->n_resident && ctx->d_quirk)
      PG_HIP(ctx, hipMemcpy(nq, ctx->d_quirk, (size_t)ctx->n_resident * 256 * 4, hipMemcpyDeviceToDevice));
    if (ctx->d_quirk) PG_HIP(ctx, hipFree(ctx->d_quirk));
    ctx->d_quirk = nq;
    ctx->quirk_cap = cap;
  }
  for (uint32_t i = ctx->n_resident; i < ng; ++i) {
    PgGenome& g = ctx->genomes[i];
    PG_HIP(ctx, hipMemcpyAsync(ctx->d_codes + g.arena_start / 16, g.codes.data(), g.codes.size() * 4,
                               hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(ctx->d_mask + g.arena_start / 32, g.mask.data(), g.mask.size() * 4, hipMemcpyHostToDevice,
                               ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(ctx->d_quirk + (size_t)i * 256, g.quirk.data(), 256 * 4, hipMemcpyHostToDevice,
                               ctx->stream));
  }
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t i = ctx->n_resident; i < ng; ++i) {
    PgGenome& g = ctx->genomes[i];
    g.resident = true;
    std::vector<uint32_t>().swap(g.codes);  // the HBM copy is now the only one
    std::vector<uint32_t>().swap(g.mask);
  }
  ctx->n_resident = ng;
  return PG_OK;
}

int pg_tetra_algorithmic_bytes(const pg_ctx* ctx, const int32_t* ids, uint32_t n, uint64_t* bytes_out, uint64_t* bases_out) {
  if (!ctx) return PG_E_ARG;
  uint64_t bytes = 0, bases = 0;
  const uint32_t cnt = ids ? n : (uint32_t)ctx->genomes.size();
  for (uint32_t i = 0; i < cnt; ++i) {
    const int32_t id = ids ? ids[i] : (int32_t)i;
    if (id < 0 || (size_t)id >= ctx->genomes.size()) return PG_E_ARG;
    const PgGenome& g = ctx->genomes[id];
    bases += g.total_len;
    bytes += (g.stream_len + 3) / 4 + (g.stream_len + 7) / 8 + PG_ACC_WORDS * 8;  // codes + mask read, counts written
  }
  if (bytes_out) *bytes_out = bytes;
  if (bases_out) *bases_out = bases;
  return PG_OK;
}

// ---- TETRA ----------------------------------------------------------------------------------------------------
namespace {
// queue: D2H of the first `bytes` of the result block
int fetch_result_async(pg_ctx* ctx, uint64_t bytes) {
  PG_HIP(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_result, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return PG_OK;
}
}  // namespace

int pg_tetra_counts(pg_ctx* ctx, const int32_t* ids, uint32_t n, uint64_t* c2, uint64_t* c3, uint64_t* c4) {
  if (!ctx || (n && !ids)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch(ctx, ids, n))) return rc;
  if ((rc = ensure_result(ctx, n, false))) return rc;
  if (n == 0) return PG_OK;
  if ((rc = pg_launch_tetra_count(ctx, n))) return rc;
  if ((rc = pg_launch_tetra_finalize(ctx, n, ctx->d_acc))) return rc;
  PG_HIP(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_counts, (size_t)n * PG_ACC_WORDS * 8, hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t g = 0; g < n; ++g) {
    const unsigned long long* src = ctx->h_counts + (size_t)g * PG_ACC_WORDS;
    if (c2) std::memcpy(c2 + (size_t)g * 16, src, 16 * 8);
    if (c3) std::memcpy(c3 + (size_t)g * 64, src + 16, 64 * 8);
    if (c4) std::memcpy(c4 + (size_t)g * 256, src + 80, 256 * 8);
  }
  return PG_OK;
}

int pg_tetra_zscores(pg_ctx* ctx, const uint64_t* c2, const uint64_t* c3, const uint64_t* c4, uint32_t n, double* z,
                     uint8_t* present) {
  if (!ctx || (n && (!c2 || !c3 || !c4))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch_scratch(ctx, n))) return rc;
  if ((rc = ensure_result(ctx, n, false))) return rc;
  if (n == 0) return PG_OK;
  for (uint32_t g = 0; g < n; ++g) {
    unsigned long long* dst = ctx->h_counts + (size_t)g * PG_ACC_WORDS;
    std::memcpy(dst, c2 + (size_t)g * 16, 16 * 8);
    std::memcpy(dst + 16, c3 + (size_t)g * 64, 64 * 8);
    std::memcpy(dst + 80, c4 + (size_t)g * 256, 256 * 8);
  }
  PG_HIP(ctx, hipMemcpyAsync(ctx->d_counts, ctx->h_counts, (size_t)n * PG_ACC_WORDS * 8, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = pg_launch_tetra_finalize(ctx, n, nullptr))) return rc;
  const ResultLayout L(n, false);
  if ((rc = fetch_result_async(ctx, L.bytes))) return rc;
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (z) std::memcpy(z, ctx->h_result + L.off_z, (size_t)n * 256 * 8);
  if (present) std::memcpy(present, ctx->h_result + L.off_present, (size_t)n * 256);
  return PG_OK;
}

int pg_tetra_corr(pg_ctx* ctx, const double* z, const uint8_t* present, uint32_t n, double* out) {
  if (!ctx || (n && (!z || !present || !out))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch_scratch(ctx, n))) return rc;
  if ((rc = ensure_result(ctx, n, true))) return rc;
  if (n == 0) return PG_OK;
  const ResultLayout L(n, true);
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h_result is reused as the upload staging area
  std::memcpy(ctx->h_result + L.off_z, z, (size_t)n * 256 * 8);
  std::memcpy(ctx->h_result + L.off_present, present, (size_t)n * 256);
  PG_HIP(ctx, hipMemcpyAsync(ctx->d_z, ctx->h_result + L.off_z, (size_t)n * 256 * 8, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(ctx->d_present, ctx->h_result + L.off_present, (size_t)n * 256, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = pg_launch_tetra_stats(ctx, ctx->d_z, ctx->d_present, n))) return rc;
  if ((rc = pg_launch_tetra_pairs(ctx, n, 0, n, ctx->d_corr, true))) return rc;
  if ((rc = fetch_result_async(ctx, L.bytes))) return rc;
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = check_flags(ctx, n))) return rc;
  std::memcpy(out, ctx->h_result + L.off_corr, (size_t)n * n * 8);
  return PG_OK;
}

int pg_tetra_matrix_enqueue(pg_ctx* ctx, const int32_t* ids, uint32_t n, int fetch_z) {
  if (!ctx || (n && !ids)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch(ctx, ids, n))) return rc;
  if ((rc = ensure_result(ctx, n, true))) return rc;
  if (n == 0) return PG_OK;
  if ((rc = pg_launch_tetra_count(ctx, n))) return rc;
  if ((rc = pg_launch_tetra_finalize(ctx, n, ctx->d_acc))) return rc;
  if ((rc = pg_launch_tetra_pairs(ctx, n, 0, n, ctx->d_corr, true))) return rc;
  const ResultLayout L(n, true);
  return fetch_result_async(ctx, fetch_z ? L.bytes : L.bytes_corr_only);
}

int pg_tetra_matrix_fetch(pg_ctx* ctx, uint32_t n, double* z_out, uint8_t* present_out, double* corr_out) {
  if (!ctx) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const ResultLayout L(n, true);
  if (L.bytes > ctx->result_cap) return pg_fail(ctx, PG_E_ARG, "fetch larger than last batch");
  if (z_out) std::memcpy(z_out, ctx->h_result + L.off_z, (size_t)n * 256 * 8);
  if (present_out) std::memcpy(present_out, ctx->h_result + L.off_present, (size_t)n * 256);
  if (corr_out) {
    const int rc = check_flags(ctx, n);
    if (rc) return rc;
    std::memcpy(corr_out, ctx->h_result + L.off_corr, (size_t)n * n * 8);
  }
  return PG_OK;
}

int pg_tetra_matrix(pg_ctx* ctx, const int32_t* ids, uint32_t n, double* z_out, uint8_t* present_out, double* corr_out) {
  const int rc = pg_tetra_matrix_enqueue(ctx, ids, n, (z_out || present_out) ? 1 : 0);
  if (rc) return rc;
  return pg_tetra_matrix_fetch(ctx, n, z_out, present_out, corr_out);
}

int pg_tetra_zscores_dev(pg_ctx* ctx, const int32_t* ids, uint32_t n, double* d_z, uint8_t* d_present) {
  if (!ctx || (n && (!ids || !d_z || !d_present))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch(ctx, ids, n))) return rc;
  if ((rc = ensure_result(ctx, n, false))) return rc;
  if (n == 0) return PG_OK;
  if ((rc = pg_launch_tetra_count(ctx, n))) return rc;
  if ((rc = pg_launch_tetra_finalize(ctx, n, ctx->d_acc))) return rc;
  PG_HIP(ctx, hipMemcpyAsync(d_z, ctx->d_z, (size_t)n * 256 * 8, hipMemcpyDeviceToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(d_present, ctx->d_present, (size_t)n * 256, hipMemcpyDeviceToDevice, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PG_OK;
}

int pg_tetra_corr_rows_dev(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n, uint32_t row0,
                           uint32_t nrows, double* d_out) {
  if (!ctx || (n && (!d_z || !d_present)) || row0 + (uint64_t)nrows > n || (nrows && !d_out))
    return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch_scratch(ctx, n))) return rc;
  if ((rc = ensure_result(ctx, 1, false))) return rc;  // only the flags words are used
  if (n == 0) return PG_OK;
  ctx->batch_ids.clear();  // d_dev / d_ss no longer describe the cached batch
  if ((rc = pg_launch_tetra_stats(ctx, d_z, d_present, n))) return rc;
  if ((rc = pg_launch_tetra_pairs(ctx, n, row0, nrows, d_out, false))) return rc;
  if ((rc = fetch_result_async(ctx, 16))) return rc;
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return check_flags(ctx, n);
}

// ---- ANIm -------------------------------------------------------------------------------------------------------
// Ordered pairs grouped by their LDS-table genome are cut into chunks (whole launches); the context's workers (two host
// threads, each with its own stream and scratch) take chunks in turn, so that one launch's low-occupancy tail overlaps the
// other's streaming kernels.  run(chunk_begin, chunk_end, worker) processes order[chunk_begin .. chunk_end).
static int anim_run_chunks(pg_ctx* ctx, const std::vector<std::pair<uint64_t, uint64_t>>& chunks,
                           const std::function<int(uint64_t, uint64_t, int)>& run, int slot_base = 0, int max_workers = 0) {
  const int cap_w = max_workers > 0 ? max_workers : ctx->anim_workers;
  const int workers = (int)std::min<size_t>((size_t)cap_w, chunks.size() ? chunks.size() : 1);
  std::atomic<size_t> next{0};
  std::atomic<int> first_rc{PG_OK};
  auto body = [&](int w) {
    (void)hipSetDevice(ctx->device);
    pg_anim_set_worker(ctx, slot_base + w);
    for (size_t c; (c = next++) < chunks.size() && first_rc.load() == PG_OK;) {
      int rc;
      try { rc = run(chunks[c].first, chunks[c].second, w); }      // (a worker thread must not let an exception escape: std::terminate)
      catch (const std::bad_alloc&) { rc = pg_fail(ctx, PG_E_NOMEM, "out of host memory in an ANIm worker"); }
      catch (const std::exception& e) { rc = pg_fail(ctx, PG_E_HIP, std::string("ANIm worker: ") + e.what()); }
      int expect = PG_OK;
      if (rc != PG_OK) first_rc.compare_exchange_strong(expect, rc);
    }
    pg_anim_set_worker(ctx, 0);
    pg_tls_stream = nullptr;
  };
  std::vector<std::thread> others;
  for (int w = 1; w < workers; ++w) others.emplace_back(body, w);
  body(0);
  for (auto& th : others) th.join();
  return first_rc.load();
}

// chunk boundaries over pairs sorted by table genome: at most max_pairs pairs and max_refs distinct table genomes each
static std::vector<std::pair<uint64_t, uint64_t>> anim_chunks(const int32_t* table_ids, const std::vector<uint64_t>& order, uint64_t max_pairs,
                                                              uint32_t max_refs) {
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  const uint64_t n = order.size();
  uint64_t i = 0;
  while (i < n) {
    uint64_t j = i;
    uint32_t nrefs = 0;
    int32_t last = -1;
    while (j < n && j - i < max_pairs) {
      const int32_t rid = table_ids[order[j]];
      if (rid != last) { if (nrefs == max_refs) break; ++nrefs; last = rid; }
      ++j;
    }
    chunks.push_back({i, j});
    i = j;
  }
  return chunks;
}

// The match budget of a call: the configured one, but never more than what the device has FREE right now can hold (~384 B of
// scratch per exact match in flight: pg_anim.hip's per-match arrays) — several processes may share one GPU (ranks of a debugging
// run, another job's context), and a budget sized for an empty 288 GB device would then over-commit it.  Scratch already held by
// this context counts as available (it is reused).
static uint64_t anim_match_budget(pg_ctx* ctx) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return ctx->anim_batch_matches; }
  const uint64_t fit = (uint64_t)(0.7 * (double)free_b) / 384u;
  const uint64_t floor_ = 4ull << 20;      // (always enough for a few related 5 Mb pairs; a launch that still does not fit reports PG_E_HIP / NOMEM)
  return std::max<uint64_t>(floor_, std::min<uint64_t>(ctx->anim_batch_matches, fit + __atomic_load_n(&ctx->anim_scratch_matches_held, __ATOMIC_RELAXED)));
}

int pg_anim_set_batch_budget(pg_ctx* ctx, uint32_t max_pairs, uint64_t max_matches) {
  if (!ctx || max_pairs == 0 || max_matches < 1024) return pg_fail(ctx, PG_E_ARG, "bad argument");
  ctx->anim_batch_pairs = max_pairs;
  ctx->anim_batch_matches = max_matches;
  return PG_OK;
}

int pg_anim_set_extender(pg_ctx* ctx, int extender) {      // (deprecation stub: the one extender left is MUMmer's own)
  if (!ctx) return PG_E_ARG;
  if (extender != PG_EXTENDER_NUCMER) return pg_fail(ctx, PG_E_ARG, "pg_anim_set_extender: only PG_EXTENDER_NUCMER exists (the approximate banded64 extender was retired)");
  return PG_OK;
}

int pg_anim_set_workers(pg_ctx* ctx, int workers) {
  if (!ctx || workers < 1 || workers > pg_ctx::MAX_WORKERS) return pg_fail(ctx, PG_E_ARG, "workers must be 1 ... 4");
  ctx->anim_workers = workers;
  return PG_OK;
}

int pg_anim_counters(pg_ctx* ctx, uint64_t* out, int reset) {
  if (!ctx || !out) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  return pg_anim_counters_read(ctx, out, reset);
}

// slot_base / lane_workers: which of the context's worker slots the call drives (pg_anim_pairs: all of them from 0; the lanes of
// pg_anim_pairs_enqueue: two each); budget_div: the share of the match budget one of its workers may take
static int anim_pairs_body(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                           int filter_1to1, pg_anim_result* out, int slot_base, int lane_workers, int budget_div);

int pg_anim_pairs(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                  int filter_1to1, pg_anim_result* out) {
  if (ctx) {      // the blocking call owns every worker slot: not while enqueued calls are in flight
    std::lock_guard<std::mutex> lk(ctx->anim_async_mu);
    if (ctx->anim_async[0].busy || ctx->anim_async[1].busy) return pg_fail(ctx, PG_E_ARG, "pg_anim_pairs while enqueued calls are in flight: fetch them first");
  }
  return anim_pairs_body(ctx, ref_ids, qry_ids, n_pairs, maxmatch, filter_1to1, out, 0, 0, 1);
}

int pg_anim_pairs_enqueue(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                          int filter_1to1, uint64_t* ticket) {
  if (!ctx || !ticket || (n_pairs && (!ref_ids || !qry_ids))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  int rc;
  if ((rc = pg_upload(ctx))) return rc;      // (on the caller's thread: the lanes never race to upload)
  std::lock_guard<std::mutex> lk(ctx->anim_async_mu);
  int lane = -1;
  for (int l = 0; l < 2; ++l) if (!ctx->anim_async[l].busy) { lane = l; break; }
  if (lane < 0) return pg_fail(ctx, PG_E_CAPACITY, "two enqueued ANIm calls are in flight: fetch one first");
  // Blocking calls before this one may have grown two worker slots to most of the device (a 1000-genome grid leaves ~200 GB of launch
  // scratch for reuse): the other lane's slots would find nothing left.  With no call in flight and less than 40 % of the device free,
  // the slots' scratch is given back first (they grow again to what enqueued calls need: half-sized launches).
  if (!ctx->anim_async[0].busy && !ctx->anim_async[1].busy) {
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(ctx->device);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b < 0.4 * (double)total_b) pg_anim_release_worker_scratch(ctx);
    (void)hipGetLastError();
  }
  pg_ctx::AnimAsync& J = ctx->anim_async[lane];
  J.r.assign(ref_ids, ref_ids + n_pairs);
  J.q.assign(qry_ids, qry_ids + n_pairs);
  J.out.assign(n_pairs, pg_anim_result{});
  J.ticket = ctx->anim_next_ticket++;
  J.rc = PG_OK;
  J.busy = true;
  const int lane_workers = std::min(2, std::max(1, ctx->anim_workers));
  J.th = std::thread([ctx, &J, lane, lane_workers, maxmatch, filter_1to1]() {
    (void)hipSetDevice(ctx->device);
    try { J.rc = anim_pairs_body(ctx, J.r.data(), J.q.data(), J.r.size(), maxmatch, filter_1to1, J.out.data(), 2 * lane, lane_workers, 2); }
    catch (const std::bad_alloc&) { J.rc = pg_fail(ctx, PG_E_NOMEM, "out of host memory in an enqueued ANIm call"); }
    catch (const std::exception& e) { J.rc = pg_fail(ctx, PG_E_HIP, std::string("enqueued ANIm call: ") + e.what()); }
  });
  *ticket = J.ticket;
  return PG_OK;
}

int pg_anim_pairs_fetch(pg_ctx* ctx, uint64_t ticket, pg_anim_result* out, uint64_t n_pairs) {
  if (!ctx || (n_pairs && !out)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  pg_ctx::AnimAsync* J = nullptr;
  {
    std::lock_guard<std::mutex> lk(ctx->anim_async_mu);
    for (int l = 0; l < 2; ++l) if (ctx->anim_async[l].busy && ctx->anim_async[l].ticket == ticket) J = &ctx->anim_async[l];
  }
  if (!J) return pg_fail(ctx, PG_E_ARG, "no enqueued ANIm call has this ticket");
  if (J->th.joinable()) J->th.join();
  int rc = J->rc;
  if (rc == PG_OK && n_pairs != J->out.size()) rc = pg_fail(ctx, PG_E_ARG, "pg_anim_pairs_fetch: n_pairs differs from the enqueued call's");
  if (rc == PG_OK && n_pairs) memcpy(out, J->out.data(), n_pairs * sizeof(pg_anim_result));
  std::lock_guard<std::mutex> lk(ctx->anim_async_mu);
  J->r.clear(); J->q.clear(); J->out.clear(); J->out.shrink_to_fit();
  J->busy = false;
  return rc;
}

static int anim_pairs_body(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch,
                           int filter_1to1, pg_anim_result* out, int slot_base, int lane_workers, int budget_div) {
  if (!ctx || (n_pairs && (!ref_ids || !qry_ids || !out))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  for (uint64_t i = 0; i < n_pairs; ++i)
    if (ref_ids[i] < 0 || (size_t)ref_ids[i] >= ctx->genomes.size() || qry_ids[i] < 0 ||
        (size_t)qry_ids[i] >= ctx->genomes.size())
      return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  // Launch order.  A pair whose reverse (roles swapped) is in the call too shares its seeding with it, if both are in one
  // launch (pg_anim.hip: roles).  Every pair gets a HUB genome: its reference, or — with the reverse present — the genome
  // of the two that is the reference of more pairs of the call (ties: by the ids' parity, so that a full grid makes every
  // genome the hub of half its partners).  A pair and its reverse have the same hub; pairs are ordered by hub, launches are
  // runs of whole hub groups, and inside a launch the pairs are grouped by reference (one k-mer table per seeded reference).
  std::vector<uint64_t> order(n_pairs);
  std::vector<int32_t> hub(n_pairs);
  {
    std::vector<uint32_t> deg(ctx->genomes.size(), 0);
    for (uint64_t i = 0; i < n_pairs; ++i) { order[i] = i; ++deg[ref_ids[i]]; hub[i] = ref_ids[i]; }
    auto lo = [&](uint64_t i) { return std::min(ref_ids[i], qry_ids[i]); };
    auto hi = [&](uint64_t i) { return std::max(ref_ids[i], qry_ids[i]); };
    // stable counting sorts over the genome ids (a call has up to 10^6 pairs: comparison sorts cost a quarter of a second here)
    std::vector<uint64_t> tmp(n_pairs);
    std::vector<uint64_t> start(ctx->genomes.size() + 1);
    auto sort_by = [&](auto key) {
      std::fill(start.begin(), start.end(), 0);
      for (uint64_t i = 0; i < n_pairs; ++i) ++start[(size_t)key(order[i]) + 1];
      for (size_t g = 0; g + 1 < start.size(); ++g) start[g + 1] += start[g];
      for (uint64_t i = 0; i < n_pairs; ++i) tmp[start[(size_t)key(order[i])]++] = order[i];
      order.swap(tmp);
    };
    sort_by(hi);
    sort_by(lo);   // by (lo, hi), the call's order inside
    for (uint64_t i = 0; i < n_pairs;) {
      uint64_t j = i;
      bool fwd = false, rev = false;
      while (j < n_pairs && lo(order[j]) == lo(order[i]) && hi(order[j]) == hi(order[i])) {
        (ref_ids[order[j]] <= qry_ids[order[j]] ? fwd : rev) = true;
        ++j;
      }
      if (fwd && rev) {
        const int32_t a = lo(order[i]), b = hi(order[i]);
        const int32_t h = deg[a] != deg[b] ? (deg[a] > deg[b] ? a : b) : ((((a + b) & 1) == 0) ? a : b);
        for (uint64_t k = i; k < j; ++k) hub[order[k]] = h;
      }
      i = j;
    }
    for (uint64_t i = 0; i < n_pairs; ++i) order[i] = i;   // back to the call's order, then by (hub, reference)
    sort_by([&](uint64_t i) { return ref_ids[i]; });
    sort_by([&](uint64_t i) { return hub[i]; });
  }
  // launches: as many pairs as the scratch budget allows, at most MAX_REFS hubs; with W workers the launches are 1/W as
  // large and each worker gets 1/W of the match budget, so the memory in use is the same
  // (measured, MI355X r04: one family of 25 genomes = 600 related pairs takes 0.58 s on one worker and 0.71 s split over two — each half
  // keeps the launch's sequential tails and they contend; four families = 2 400 pairs: two workers win)
  const int W = n_pairs >= 1024 ? (lane_workers > 0 ? lane_workers : ctx->anim_workers) : 1;
  // (an enqueued call shares the device with one other call: half the pairs and half the matches in flight each, so that the
  // four worker slots together hold what the two of a blocking call hold)
  const uint32_t MAX_PAIRS = std::max<uint32_t>(1u, ctx->anim_batch_pairs / (uint32_t)(W * budget_div)), MAX_REFS = 256;
  const uint64_t max_matches = std::max<uint64_t>(4ull << 20, anim_match_budget(ctx) / ((uint64_t)W * (uint64_t)budget_div));
  // about equal launches, a multiple of W of them, none above the per-launch budget
  const uint64_t cap = MAX_PAIRS ? MAX_PAIRS : 1;
  const uint64_t n_target = (uint64_t)W * ((n_pairs + (uint64_t)W * cap - 1) / ((uint64_t)W * cap));
  const uint64_t target = n_target ? std::min<uint64_t>(cap, (n_pairs + n_target - 1) / n_target) : cap;
  std::vector<std::pair<uint64_t, uint64_t>> chunks;
  for (uint64_t i = 0; i < n_pairs;) {
    uint64_t j = i;
    uint32_t hubs = 0;
    while (j < n_pairs && hubs < MAX_REFS) {
      uint64_t g = j;   // the hub group [j, g)
      while (g < n_pairs && hub[order[g]] == hub[order[j]]) ++g;
      if (j > i && g - i > target) break;
      j = g; ++hubs;
    }
    if (j - i > cap) j = i + cap;   // one hub with more pairs than a launch holds: cut it (pairs cut off from their reverse are seeded themselves)
    {   // the launch's pairs grouped by reference (stable counting sort of the range)
      std::vector<uint64_t> cnt(ctx->genomes.size() + 1, 0), part(order.begin() + i, order.begin() + j);
      for (uint64_t x : part) ++cnt[(size_t)ref_ids[x] + 1];
      for (size_t g = 0; g + 1 < cnt.size(); ++g) cnt[g + 1] += cnt[g];
      for (uint64_t x : part) order[i + cnt[(size_t)ref_ids[x]]++] = x;
    }
    chunks.push_back({i, j});
    i = j;
  }
  return anim_run_chunks(ctx, chunks, [&](uint64_t i, uint64_t j, int) -> int {
    std::vector<int32_t> r, q;
    std::vector<pg_anim_result> res;
    while (i < j) {
      r.clear(); q.clear();
      for (uint64_t k = i; k < j; ++k) { r.push_back(ref_ids[order[k]]); q.push_back(qry_ids[order[k]]); }
      res.assign(j - i, pg_anim_result{});
      uint32_t done = 0;
      const int rc2 = pg_anim_run_batch(ctx, r.data(), q.data(), (uint32_t)(j - i), filter_1to1, maxmatch != 0, max_matches, res.data(), &done);
      if (rc2) return rc2;
      for (uint64_t k = i; k < i + done; ++k) out[order[k]] = res[k - i];
      i += done;   // pairs beyond the match budget are taken up by the next launch
    }
    return PG_OK;
  }, slot_base, W);
}

int pg_anim_pair_alignments(pg_ctx* ctx, int32_t ref_id, int32_t qry_id, pg_anim_alignment* out, uint32_t cap, uint32_t* n_out) {
  if (!ctx || !n_out || (cap && !out)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (ref_id < 0 || (size_t)ref_id >= ctx->genomes.size() || qry_id < 0 || (size_t)qry_id >= ctx->genomes.size())
    return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  pg_anim_result res{};
  uint32_t done = 0;
  if ((rc = pg_anim_run_batch(ctx, &ref_id, &qry_id, 1, 1, 0, anim_match_budget(ctx), &res, &done))) return rc;
  if (res.status == PG_E_CAPACITY) return pg_fail(ctx, PG_E_CAPACITY, "anim: work buffers overflowed for this pair");
  *n_out = (uint32_t)res.reserved;
  const uint32_t n = *n_out < cap ? *n_out : cap;
  std::vector<pg_anim_alignment> tmp(*n_out);
  if ((rc = pg_anim_fetch_alignments(ctx, ref_id, qry_id, *n_out, tmp.data()))) return rc;
  for (uint32_t i = 0; i < n; ++i) out[i] = tmp[i];
  return PG_OK;
}

static int anim_alignments_batch_body(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch, int with_indels,
                                      uint64_t* aln_offsets, uint64_t* n_indels);
// No exception may cross the C ABI (the call grows host vectors: sink, indel lists, the stored result), and the thread's sink must
// not outlive the call whichever way it ends.
int pg_anim_alignments_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch, int with_indels,
                             uint64_t* aln_offsets, uint64_t* n_indels) {
  struct SinkReset { ~SinkReset() { pg_anim_set_sink(nullptr); } } reset;
  try { return anim_alignments_batch_body(ctx, ref_ids, qry_ids, n_pairs, maxmatch, with_indels, aln_offsets, n_indels); }
  catch (const std::bad_alloc&) { return pg_fail(ctx, PG_E_NOMEM, "out of host memory while collecting alignment records"); }
  catch (const std::exception& e) { return pg_fail(ctx, PG_E_INTERNAL, std::string("pg_anim_alignments_batch: ") + e.what()); }
}
static int anim_alignments_batch_body(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint64_t n_pairs, int maxmatch, int with_indels,
                                      uint64_t* aln_offsets, uint64_t* n_indels) {
  if (!ctx || !aln_offsets || (n_pairs && (!ref_ids || !qry_ids))) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (uint64_t i = 0; i < n_pairs; ++i)
    if (ref_ids[i] < 0 || (size_t)ref_ids[i] >= ctx->genomes.size() || qry_ids[i] < 0 || (size_t)qry_ids[i] >= ctx->genomes.size())
      return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  ctx->aln_store.clear(); ctx->aln_indel_off.assign(1, 0); ctx->aln_indels.clear();
  // launches: the call's pairs grouped by reference genome (its seed table is built once per launch), a bounded number each
  std::vector<uint64_t> order(n_pairs);
  for (uint64_t i = 0; i < n_pairs; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return ref_ids[a] < ref_ids[b]; });
  const uint64_t LAUNCH = with_indels ? 64 : 1024;      // (a traceback launch keeps its walks' pieces and their paths on the host)
  PgAlnSink sink;
  sink.with_indels = with_indels != 0;
  pg_anim_set_sink(&sink);
  std::vector<int32_t> r, q;
  std::vector<pg_anim_result> res;
  rc = PG_OK;
  for (uint64_t i = 0; i < n_pairs && rc == PG_OK;) {
    uint64_t j = std::min<uint64_t>(n_pairs, i + LAUNCH);
    r.clear(); q.clear();
    for (uint64_t k = i; k < j; ++k) { r.push_back(ref_ids[order[k]]); q.push_back(qry_ids[order[k]]); }
    res.assign(j - i, pg_anim_result{});
    uint32_t done = 0;
    rc = pg_anim_run_batch(ctx, r.data(), q.data(), (uint32_t)(j - i), 1, maxmatch != 0, anim_match_budget(ctx), res.data(), &done);
    i += done;
  }
  pg_anim_set_sink(nullptr);
  if (rc) return rc;
  // back to the caller's order
  std::vector<uint64_t> first(n_pairs + 1, 0), at_sorted(n_pairs + 1, 0);
  for (uint64_t k = 0; k < n_pairs; ++k) at_sorted[k + 1] = at_sorted[k] + sink.pair_count[k];
  for (uint64_t k = 0; k < n_pairs; ++k) first[order[k] + 1] = sink.pair_count[k];
  for (uint64_t i = 0; i < n_pairs; ++i) first[i + 1] += first[i];
  ctx->aln_store.resize(sink.alns.size());
  std::vector<uint64_t> src_of(sink.alns.size());
  for (uint64_t k = 0; k < n_pairs; ++k)
    for (uint64_t t = 0; t < sink.pair_count[k]; ++t) { ctx->aln_store[first[order[k]] + t] = sink.alns[at_sorted[k] + t]; src_of[first[order[k]] + t] = at_sorted[k] + t; }
  uint64_t total = 0;
  if (with_indels) {
    ctx->aln_indel_off.assign(sink.alns.size() + 1, 0);
    for (uint64_t a = 0; a < sink.alns.size(); ++a) { total += sink.indels[src_of[a]].size(); ctx->aln_indel_off[a + 1] = total; }
    ctx->aln_indels.reserve(total);
    for (uint64_t a = 0; a < sink.alns.size(); ++a) ctx->aln_indels.insert(ctx->aln_indels.end(), sink.indels[src_of[a]].begin(), sink.indels[src_of[a]].end());
  }
  for (uint64_t i = 0; i <= n_pairs; ++i) aln_offsets[i] = first[i];
  if (n_indels) *n_indels = total;
  return PG_OK;
}

int pg_anim_alignments_read(pg_ctx* ctx, pg_anim_alignment* out, uint64_t* indel_offsets, int64_t* indels) {
  if (!ctx || (!out && !ctx->aln_store.empty())) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (size_t i = 0; i < ctx->aln_store.size(); ++i) out[i] = ctx->aln_store[i];
  if (indel_offsets) {
    if (ctx->aln_indel_off.size() != ctx->aln_store.size() + 1) return pg_fail(ctx, PG_E_ARG, "the stored result has no indel lists (with_indels was 0)");
    for (size_t i = 0; i < ctx->aln_indel_off.size(); ++i) indel_offsets[i] = ctx->aln_indel_off[i];
  }
  if (indels) for (size_t i = 0; i < ctx->aln_indels.size(); ++i) indels[i] = ctx->aln_indels[i];
  return PG_OK;
}

int pg_anim_reduce(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                   const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                   int apply_filter, pg_anim_result* out) {
  if (!ctx || !offsets || (n_pairs && !out)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (offsets[n_pairs] && (!rseq || !qseq || !rs || !re || !qs || !qe || !errors)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (uint32_t p = 0; p < n_pairs; ++p)
    if (offsets[p + 1] < offsets[p]) return pg_fail(ctx, PG_E_ARG, "offsets must be non-decreasing");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  if (n_pairs == 0) return PG_OK;
  return pg_anim_reduce_run(ctx, n_pairs, offsets, rseq, qseq, rs, re, qs, qe, errors, apply_filter, out);
}

int pg_anib_reduce(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                   const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen, const double* pident,
                   int64_t* aln_length_out, int64_t* sim_errors_out, double* pid_out) {
  if (!ctx || !offsets || (n_pairs && (!n_frags || !aln_length_out || !sim_errors_out || !pid_out)))
    return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (offsets[n_pairs] && (!frag || !length || !mismatch || !gaps || !qlen || !pident)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  for (uint64_t i = 0; i < offsets[n_pairs]; ++i)
    if (qlen[i] <= 0) return pg_fail(ctx, PG_E_ARG, "qlen must be positive");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  if (n_pairs == 0) return PG_OK;
  return pg_anib_reduce_run(ctx, n_pairs, offsets, n_frags, frag, length, mismatch, gaps, qlen, pident, aln_length_out,
                            sim_errors_out, pid_out);
}

// ---- ANIb fragment mode ----------------------------------------------------------------------------------------
static constexpr uint64_t ANIB_MAX_SLOTS = 4ull << 20;   // (pair, fragment) slots per launch: 192 B of rows each

int pg_anib_pairs(pg_ctx* ctx, const int32_t* qry_ids, const int32_t* sbj_ids, uint64_t n_pairs, uint32_t fragsize, pg_anib_result* out) {
  if (!ctx || fragsize == 0 || fragsize > 1020 || (n_pairs && (!qry_ids || !sbj_ids || !out)))
    return pg_fail(ctx, PG_E_ARG, "bad argument (fragment sizes up to pyani's 1020 are supported)");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  for (uint64_t i = 0; i < n_pairs; ++i)
    if (qry_ids[i] < 0 || (size_t)qry_ids[i] >= ctx->genomes.size() || sbj_ids[i] < 0 || (size_t)sbj_ids[i] >= ctx->genomes.size())
      return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  // A query genome with more fragments than a launch's per-genome counters hold (15 872: 16.1 Mb at 1020 nt) cannot be searched:
  // its pairs get status = PG_E_CAPACITY, the others are computed (the call does not fail for them).
  std::vector<uint64_t> order;
  order.reserve(n_pairs);
  {
    std::vector<int64_t> nfrag(ctx->genomes.size(), -1);
    for (uint64_t i = 0; i < n_pairs; ++i) {
      int64_t& nf = nfrag[(size_t)qry_ids[i]];
      if (nf < 0) {
        const PgGenome& Q = ctx->genomes[(size_t)qry_ids[i]];
        nf = 0;
        for (uint32_t r = 0; r < Q.n_rec; ++r) nf += ((int64_t)(Q.rec_start[r + 1] - 1 - Q.rec_start[r]) + fragsize - 1) / fragsize;
      }
      if (nf > (int64_t)ANIB_MAX_SLOTS / pg_ctx::MAX_WORKERS) {      // (more fragments than one launch holds: a query genome beyond ~1 Gb)
        out[i] = pg_anib_result{};
        out[i].n_frags = (int32_t)nf;
        out[i].status = PG_E_CAPACITY;
      } else order.push_back(i);
    }
  }
  n_pairs = order.size();      // (pairs are addressed through order[] from here on)
  // grouped by subject genome (the LDS-table role of the seeding stage), then in chunks the launch budgets allow
  std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return sbj_ids[a] < sbj_ids[b]; });
  const int W = n_pairs >= 64 ? ctx->anim_workers : 1;
  const uint32_t MAX_PAIRS = ctx->anim_batch_pairs / W, MAX_REFS = 256;
  const uint64_t max_matches = anim_match_budget(ctx) / W, max_slots = ANIB_MAX_SLOTS / W;
  // (fragment launches are bounded by their (pair, fragment) slots: cut the chunks so that both workers get several)
  const auto chunks = anim_chunks(sbj_ids, order, std::max<uint64_t>(1, std::min<uint64_t>(MAX_PAIRS, W >= 2 ? (n_pairs + 4 * W - 1) / (4 * W) : n_pairs)), MAX_REFS);
  return anim_run_chunks(ctx, chunks, [&](uint64_t i, uint64_t j, int) -> int {
    std::vector<int32_t> s, q;
    std::vector<pg_anib_result> res;
    while (i < j) {
      s.clear(); q.clear();
      for (uint64_t k = i; k < j; ++k) { s.push_back(sbj_ids[order[k]]); q.push_back(qry_ids[order[k]]); }
      res.assign(j - i, pg_anib_result{});
      PgFragArgs F{(int32_t)fragsize, res.data(), nullptr, 0, nullptr, max_slots};
      uint32_t done = 0;
      const int rc2 = pg_anim_run_batch(ctx, s.data(), q.data(), (uint32_t)(j - i), 0, 1, max_matches, nullptr, &done, &F);
      if (rc2) return rc2;
      for (uint64_t k = i; k < i + done; ++k) out[order[k]] = res[k - i];
      i += done;
    }
    return PG_OK;
  });
}

int pg_anib_pair_rows(pg_ctx* ctx, int32_t qry_id, int32_t sbj_id, uint32_t fragsize, pg_anib_row* out, uint32_t cap, uint32_t* n_out) {
  if (!ctx || !n_out || fragsize == 0 || fragsize > 1020 || (cap && !out)) return pg_fail(ctx, PG_E_ARG, "bad argument");
  if (qry_id < 0 || (size_t)qry_id >= ctx->genomes.size() || sbj_id < 0 || (size_t)sbj_id >= ctx->genomes.size())
    return pg_fail(ctx, PG_E_ARG, "genome id out of range");
  PG_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = pg_upload(ctx))) return rc;
  pg_anib_result res{};
  PgFragArgs F{(int32_t)fragsize, &res, out, cap, n_out, ANIB_MAX_SLOTS};
  uint32_t done = 0;
  return pg_anim_run_batch(ctx, &sbj_id, &qry_id, 1, 0, 1, anim_match_budget(ctx), nullptr, &done, &F);
}

// ---- measurement -----------------------------------------------------------------------------------------------
int pg_profile_enable(pg_ctx* ctx, int on) {
  if (!ctx) return PG_E_ARG;
  ctx->profiling = on != 0;
  return PG_OK;
}
int pg_profile_config(pg_ctx* ctx, uint32_t kernel_mask, uint32_t every_n) {
  if (!ctx || every_n == 0) return PG_E_ARG;
  ctx->prof_mask = kernel_mask;
  ctx->prof_every = every_n;
  return PG_OK;
}
int pg_profile_reset(pg_ctx* ctx) {
  if (!ctx) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  prof_drain(ctx);
  for (int i = 0; i < PG_K__COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; ctx->prof_seen[i] = 0; }
  return PG_OK;
}
int pg_profile_get(pg_ctx* ctx, int which, double* total_ms_out, uint64_t* launches_out) {
  if (!ctx || which < 0 || which >= PG_K__COUNT) return PG_E_ARG;
  PG_HIP(ctx, hipSetDevice(ctx->device));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  prof_drain(ctx);
  if (total_ms_out) *total_ms_out = ctx->prof_ms[which];
  if (launches_out) *launches_out = ctx->prof_n[which];
  return PG_OK;
}

}  // extern "C"
