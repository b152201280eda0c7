// oracle/anim_cpu.cpp — CPU statement of the ANIm pair search.   TEST / MEASUREMENT INFRASTRUCTURE ONLY.
//
// What it is: the scalar functions of pyani_amd/csrc/pg_anim_core.h (MUM filter, mgaps clustering, banded affine extension,
// 1-to-1 LIS filter, parse_delta reduction — the restatement of what `nucmer --mum` + `delta-filter -1` + pyani's
// parse_delta (pyani/anim.py:240-289, 292-411; scripts/delta_filter_wrapper.py:70-93) compute, calibrated on the MUMmer
// output files the reference's tests hold) compiled for the HOST, fed by a sparse host 16-mer index (below) instead of the
// GPU's sampled LDS seeding, one ordered pair per thread (two walker threads while a pair is in its extension stage).  MUMmer itself is third-party and absent from /root/reference and
// from this image, so this is the only same-box CPU comparison there is: bench.py's `cpu_baseline` leg times it on all
// host cores ("own-cpu") and tests use it as the scalar statement the GPU pipeline must equal.
// Nothing under pyani_amd/ loads this library.
//   g++ -O2 -std=c++17 -pthread -fPIC -shared -Ipyani_amd/csrc oracle/anim_cpu.cpp -o oracle/libanimcpu.so
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <time.h>
#include <vector>
#include "pg_anim_core.h"
#include "pg_nucmer_core.h"
using namespace pga;

namespace {
struct Genome {
  std::vector<uint32_t> codes, mask;
  std::vector<int32_t> rec_start;  // stream position of each record's first base; last entry = stream length + 1
  int64_t len = 0;
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};

// records back to back with ONE dirty separator between them (the layout of pg_add_genome)
Genome pack(const uint8_t* seq, const uint64_t* rec_off, uint32_t n_rec) {
  Genome g;
  int64_t len = 0;
  for (uint32_t r = 0; r < n_rec; ++r) len += (int64_t)(rec_off[r + 1] - rec_off[r]) + (r ? 1 : 0);
  g.len = len;
  g.codes.assign(len / 16 + 2, 0);
  g.mask.assign(len / 32 + 2, 0);
  int64_t p = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    if (r) ++p;
    g.rec_start.push_back((int32_t)p);
    for (uint64_t i = rec_off[r]; i < rec_off[r + 1]; ++i, ++p) {
      int c = -1;
      switch (seq[i]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; }
      if (c >= 0) { g.codes[p >> 4] |= (uint32_t)c << (2 * (p & 15)); g.mask[p >> 5] |= 1u << (p & 31); }
    }
  }
  g.rec_start.push_back((int32_t)g.len + 1);
  return g;
}

// Seeding as a CPU program would do it (the exhaustive sorted 20-mer table — 80 MB per pair, one lookup per query
// base — cost 8.9 s for an UNRELATED 5 Mb pair with every host core busy; a baseline should not be slow for no reason): a SPARSE
// index in MUMmer 4's spirit.  Every reference position's 16-mer goes into a bucket table (counting sort on the top 22 bits, the
// other 10 bits kept beside the position: 8 B per base, no comparison sort); of the query strand only every 5th position is looked
// up — a match of >= 20 bases contains a whole 16-mer at one of its first five positions — a hit is extended to the left (5 or
// more bases: an earlier sampled position reports it) and to the right, and kept if it is >= 20 long.  Same maximal matches as
// the exhaustive table (the tests hold the results against the GPU's and the fixtures'), a fifth of the random memory accesses.
constexpr int SEED_K = 16, SEED_STEP = MIN_MATCH - SEED_K + 1, INDEX_BITS = 22;
static_assert(SEED_STEP == 5, "a match of MIN_MATCH bases must contain a sampled SEED_K-mer");
struct KmerTable {
  std::vector<uint64_t> tab;     // (low 10 bits of the 16-mer) << 32 | position, grouped by the 16-mer's top 22 bits
  std::vector<uint32_t> start;   // start[b] .. start[b + 1]: the entries of bucket b
};

void build_table(const Genome& G, KmerTable& T) {
  const SeqView R = G.view();
  constexpr int K = SEED_K, shift = 2 * K - INDEX_BITS;
  T.start.assign((size_t(1) << INDEX_BITS) + 2, 0);
  uint32_t v = 0;
  int run = 0;
  for (int64_t p = 0; p < R.len; ++p) {
    if (!R.clean(p)) { run = 0; v = 0; continue; }
    v = (v << 2) | (uint32_t)R.base(p);
    if (++run >= K) ++T.start[(v >> shift) + 2];
  }
  for (size_t b = 2; b < T.start.size(); ++b) T.start[b] += T.start[b - 1];
  T.tab.resize(T.start.back());
  v = 0; run = 0;
  for (int64_t p = 0; p < R.len; ++p) {      // start[b + 1] is bucket b's fill cursor; afterwards it is bucket b's end = bucket b + 1's start
    if (!R.clean(p)) { run = 0; v = 0; continue; }
    v = (v << 2) | (uint32_t)R.base(p);
    if (++run >= K) T.tab[T.start[(v >> shift) + 1]++] = ((uint64_t)(v & ((1u << shift) - 1u)) << 32) | (uint32_t)(p - K + 1);
  }
}

// all maximal exact matches >= MIN_MATCH between the reference and one query strand
template <typename QV>
void find_mems(const Genome& G, const KmerTable& T, const QV& Q, int strand, std::vector<Match>& out) {
  const SeqView R = G.view();
  constexpr int K = SEED_K, shift = 2 * K - INDEX_BITS;
  uint32_t v = 0;
  int run = 0;
  for (int64_t e = 0; e < Q.len(); ++e) {
    if (!Q.clean(e)) { run = 0; v = 0; continue; }
    v = (v << 2) | (uint32_t)Q.base(e);
    if (++run < K) continue;
    const int64_t q = e - K + 1;
    if (q % SEED_STEP) continue;
    const uint32_t b = v >> shift;
    const uint64_t rem = (uint64_t)(v & ((1u << shift) - 1u));
    for (uint32_t t = T.start[b]; t < T.start[b + 1]; ++t) {
      if ((T.tab[t] >> 32) != rem) continue;
      const int64_t r = (int64_t)(uint32_t)T.tab[t];
      int32_t left = 0;
      while (left < SEED_STEP && R.clean(r - 1 - left) && Q.clean(q - 1 - left) && R.base(r - 1 - left) == Q.base(q - 1 - left)) ++left;
      if (left >= SEED_STEP) continue;      // the match holds an earlier sampled position: reported from there
      int32_t L = K;
      while (R.clean(r + L) && Q.clean(q + L) && R.base(r + L) == Q.base(q + L)) ++L;
      if (left + L >= MIN_MATCH) out.push_back(Match{(int32_t)(r - left), (int32_t)(q - left), left + L, strand});
    }
  }
}

struct Result {   // = pg_anim_result (include/pyani_gpu.h)
  int64_t ref_aln_len, qry_aln_len, sim_errors, n_alignments;
  double identity;
  int32_t status;
  int32_t reserved;
};

// the extension stage is the postnuc statement (pg_nucmer_core.h: MUMmer's own extension algorithm, scalar engine)
static double thread_cpu_seconds() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
// helper_cpu_s: CPU seconds of the second walk's thread (the caller times its own thread)
Result run_pair(const Genome& G, const Genome& H, int filter_1to1, int maxmatch, double* helper_cpu_s = nullptr) {
  const SeqView R = G.view();
  std::vector<Aln> alns;
  std::vector<int32_t> a_rrec, a_qrec;
  KmerTable tab;
  build_table(G, tab);
  const int nq = (int)H.rec_start.size() - 1;
  // seeding + MUM filter + mgaps per strand; then the two strands' walks side by side (pgn::PnPairSync: MUMmer walks the clusters of
  // both strands of a record pair in ONE list — the walks consult each other where that matters)
  struct Strand { std::vector<Chain> chains; std::vector<Match> cm; std::vector<int32_t> co; int n_chains = 0; std::vector<pgn::PnAln> al; int na = 0;
                  std::vector<pgn::PnTurn> tlog; std::vector<int32_t> born; };
  Strand S[2];
  for (int strand = 0; strand < 2; ++strand) {
    StrandView Q{H.view(), strand};
    std::vector<Match> mem;
    find_mems(G, tab, Q, strand, mem);
    int n = (int)mem.size();
    if (!maxmatch) n = mum_filter(mem.data(), n, strand, [&](int32_t q) { return record_of(H.rec_start.data(), nq, strand ? (int32_t)(H.len - 1 - q) : q); });
    else std::sort(mem.begin(), mem.end(), [](const Match& a, const Match& b) { return a.q != b.q ? a.q < b.q : a.r < b.r; });      // mgaps' By_Start2: query start, then reference start
    mem.resize(n);
    std::vector<int32_t> rrec(n), qrec(n), parent(n), score(n), from(n), adj(n), order(n);
    for (int i = 0; i < n; ++i) {
      rrec[i] = record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, mem[i].r);
      const int32_t qf = strand ? (int32_t)(H.len - 1 - mem[i].q) : mem[i].q;
      qrec[i] = record_of(H.rec_start.data(), nq, qf);
    }
    Strand& T = S[strand];
    T.chains.resize(n + 1); T.cm.resize(n + 1);
    int n_cm = 0;
    mgaps_strand(mem.data(), n, strand, rrec.data(), qrec.data(), parent.data(), score.data(), from.data(), adj.data(),
                 order.data(), T.chains.data(), T.n_chains, (int)T.chains.size(), T.cm.data(), n_cm, (int)T.cm.size());
    T.n_chains = split_chains_by_ref_record(T.chains.data(), T.n_chains, T.cm.data(), [&](int32_t r) { return record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, r); });
    T.co.resize(T.n_chains);
    for (int i = 0; i < T.n_chains; ++i) T.co[i] = i;
    std::sort(T.co.begin(), T.co.end(), [&](int a, int b) { return chain_before(T.chains.data(), T.cm.data(), a, b); });
    T.al.resize(T.n_chains + 1); T.tlog.resize(T.n_chains + 1); T.born.resize(T.n_chains + 1);
  }
  pgn::PnHostShared shared;
  auto walk = [&](int strand) {
    Strand& T = S[strand];
    StrandView Q{H.view(), strand};
    const int cap = 1 << 14;   // widest anti-diagonal: MAX_ALIGNMENT_LENGTH + 1 cells
    std::vector<pgn::Cell> d0(cap), d1(cap), d2(cap);
    pgn::ScalarEngine<SeqView, StrandView> eng{R, Q, d0.data(), d1.data(), d2.data(), cap};
    std::vector<uint8_t> fused(T.n_chains + 1);
    pgn::PnPairSync<pgn::PnHostPrim> sync{pgn::PnHostPrim{&shared, strand}, T.tlog.data(), S[1 - strand].tlog.data(), T.born.data()};
    T.na = pgn::postnuc_unit(eng, T.chains.data(), T.cm.data(), T.co.data(), T.n_chains,
        [&](int c, int32_t& rl, int32_t& rh, int32_t& ql, int32_t& qh) {
          rl = G.rec_start[T.chains[c].rrec]; rh = G.rec_start[T.chains[c].rrec + 1] - 1;
          ql = H.rec_start[T.chains[c].qrec]; qh = H.rec_start[T.chains[c].qrec + 1] - 1;
          if (strand) { const int32_t a = (int32_t)H.len - qh, b = (int32_t)H.len - ql; ql = a; qh = b; } },
        fused.data(), T.al.data(), (int)T.al.size(), sync, strand);
    if (T.na < 0) T.na = -1 - T.na;
  };
  {
    double other_cpu = 0.0;
    std::thread other([&]() { const double t0 = thread_cpu_seconds(); walk(1); other_cpu = thread_cpu_seconds() - t0; });
    walk(0);
    other.join();
    if (helper_cpu_s) *helper_cpu_s = other_cpu;
  }
  for (int strand = 0; strand < 2; ++strand) {
    const Strand& T = S[strand];
    for (int i = 0; i < T.na; ++i) {
      Aln a{T.al[i].sA, T.al[i].eA + 1, T.al[i].sB, T.al[i].eB + 1, T.al[i].errors, strand, 0};
      a_rrec.push_back(record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, a.rs));
      if (strand) { const int32_t qs = (int32_t)H.len - a.qe, qe = (int32_t)H.len - a.qs; a.qs = qs; a.qe = qe; }
      a_qrec.push_back(record_of(H.rec_start.data(), nq, a.qs));
      alns.push_back(a);
    }
  }
  const int n = (int)alns.size();
  std::vector<int32_t> idx(n + 1), from(n + 1);
  std::vector<double> sc(n + 1);
  if (!filter_1to1) for (auto& a : alns) a.keep = 3;
  else {
    lis_filter(alns.data(), n, 0, a_rrec.data(), a_qrec.data(), idx.data(), sc.data(), from.data());
    lis_filter(alns.data(), n, 1, a_qrec.data(), a_rrec.data(), idx.data(), sc.data(), from.data());
  }
  const PairResult pr = reduce_pair(alns.data(), n, a_rrec.data(), a_qrec.data(), idx.data());
  Result out{};
  out.ref_aln_len = pr.ref_aln_len; out.qry_aln_len = pr.qry_aln_len; out.sim_errors = pr.sim_errors;
  out.n_alignments = pr.n_alignments;
  out.identity = pr.aligned ? (double)pr.weighted / (double)pr.aligned : 0.0;
  out.status = pr.n_alignments ? 0 : 1;
  out.reserved = n;
  return out;
}
}  // namespace

extern "C" {
// seqs[g] / rec_offs[g] / n_recs[g]: genome g as pg_add_genome takes it.  One ordered pair per thread (threads = 0: all
// hardware threads); seconds_out[i] = the CPU seconds pair i took (table build included — one process per pair is how
// pyani's runner does it, run_multiprocessing.py:130-144).  Returns 0.
int anim_cpu_pairs(const uint8_t* const* seqs, const uint64_t* const* rec_offs, const uint32_t* n_recs, uint32_t n_genomes,
                   const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, int maxmatch, int filter_1to1, int threads,
                   Result* out, double* seconds_out) {
  std::vector<Genome> G(n_genomes);
  std::vector<char> used(n_genomes, 0);
  for (uint32_t i = 0; i < n_pairs; ++i) { used[ref_ids[i]] = 1; used[qry_ids[i]] = 1; }
  unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  {
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&]() { for (uint32_t g; (g = next++) < n_genomes;) if (used[g]) G[g] = pack(seqs[g], rec_offs[g], n_recs[g]); });
    for (auto& th : pool) th.join();
  }
  std::atomic<uint32_t> next{0};
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt && t < n_pairs; ++t)
    pool.emplace_back([&]() {
      for (uint32_t i; (i = next++) < n_pairs;) {
        // CPU seconds of the pair: this thread's + the second walk's thread's (the two strands' walks run side by side, so wall time
        // would flatter the baseline whenever cores are idle)
        const double t0 = thread_cpu_seconds();
        double helper = 0.0;
        out[i] = run_pair(G[ref_ids[i]], G[qry_ids[i]], filter_1to1, maxmatch, &helper);
        if (seconds_out) seconds_out[i] = thread_cpu_seconds() - t0 + helper;
      }
    });
  for (auto& th : pool) th.join();
  return 0;
}
}
