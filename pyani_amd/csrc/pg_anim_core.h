// pg_anim_core.h — per-pair logic of the ANIm engine either side of the extension stage (MUM filter, mgaps clustering,
// 1-to-1 filter, parse_delta reduction; the extension stage itself is pg_nucmer_core.h) as plain C++ that compiles for the
// device (hipcc) AND for the host.
// The host build exists only for tools/anim_debug (a development harness in the GPU-less build container); the
// product runs these functions inside HIP kernels (pg_anim.hip).
//
// What it emulates: `nucmer --mum` + `delta-filter -1` of MUMmer 3.23 as pyani drives them (pyani/anim.py:280-288),
// reduced as pyani.anim.parse_delta does (anim.py:292-411).  MUMmer's source is NOT in the reference tree; the
// behaviour below is reconstructed from its published defaults (-l 20 -c 65 -g 90 -d 0.12 -D 5 -b 200) and calibrated
// at the alignment-record level against the real MUMmer output the reference's tests hold (tests/golden/anim/):
// with the rules below all 505 alignment records of the 17 fixture pairs that have both genomes, and all 12 734
// delta-filter decisions of the 27 .delta/.filter fixture pairs, are reproduced exactly.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PG_HD __host__ __device__ __forceinline__
#else
#define PG_HD inline
#endif

namespace pga {

// ---- parameters (nucmer / mgaps / postnuc defaults used by pyani) ---------------------------------------------
constexpr int MIN_MATCH = 20;        // nucmer -l
constexpr int MIN_CLUSTER = 65;      // nucmer -c  (mgaps -l)
constexpr int MAX_GAP = 90;          // nucmer -g  (mgaps -s)
constexpr int DIAG_DIFF = 5;         // nucmer -D  (mgaps -d)
constexpr double DIAG_FACTOR = 0.12; // nucmer -d  (mgaps -f)
#ifdef PGA_BREAK_LEN
constexpr int BREAK_LEN = PGA_BREAK_LEN;
#else
constexpr int BREAK_LEN = 200;       // nucmer -b  (anti-diagonals without a new high score)
#endif
constexpr int32_t NEG_INF = -(1 << 28);

// ---- packed genome view ------------------------------------------------------------------------------------------
// Same layout as the TETRA arena (pg_internal.h): 2-bit codes, 1-bit clean mask, records separated by one dirty base.
struct SeqView {
  const uint32_t* codes;
  const uint32_t* mask;
  int64_t len;  // stream length in bases (records + separators)
  PG_HD int base(int64_t p) const { return (int)((codes[p >> 4] >> (2 * (p & 15))) & 3u); }
  PG_HD bool clean(int64_t p) const { return p >= 0 && p < len && ((mask[p >> 5] >> (p & 31)) & 1u); }
};

// A query strand view: strand 0 = forward, 1 = reverse complement (position p of the strand = base len-1-p complemented)
struct StrandView {
  SeqView s;
  int rc;
  PG_HD int64_t len() const { return s.len; }
  PG_HD bool clean(int64_t p) const { return rc ? s.clean(s.len - 1 - p) : s.clean(p); }
  PG_HD int base(int64_t p) const { return rc ? 3 - s.base(s.len - 1 - p) : s.base(p); }
};

struct Match {   // exact match: ref [r, r+len), query-strand [q, q+len)
  int32_t r, q, len;
  int32_t strand;
};

struct Aln {     // alignment in ref / query-strand coordinates, half-open
  int32_t rs, re, qs, qe;
  int32_t errors;
  int32_t strand;
  int32_t keep;  // used by the 1-to-1 filter
};

}  // namespace pga

// =====================================================================================================================
// Per-pair bookkeeping after seeding: MUM filter -> clusters/chains -> (extension, elsewhere) -> 1-to-1 filter -> reduce
// All routines are single-threaded over small arrays (one GPU thread per ordered pair; thousands of pairs in flight).
// =====================================================================================================================
namespace pga {

template <typename T, typename Less>
PG_HD void heapsort(T* a, int n, Less less) {
  auto sift = [&](int root, int end) {
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && less(a[child], a[child + 1])) ++child;
      if (!less(a[root], a[child])) break;
      T t = a[root]; a[root] = a[child]; a[child] = t;
      root = child;
    }
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
  for (int end = n - 1; end > 0; --end) {
    T t = a[0]; a[0] = a[end]; a[end] = t;
    sift(0, end);
  }
}

// record index of stream position p: offsets[k] = first base of record k, offsets[n_rec] = stream length + 1
PG_HD int record_of(const int32_t* rec_start, int n_rec, int32_t p) {
  int lo = 0, hi = n_rec;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rec_start[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// MUM filter (`mummer -mum`): keep maximal matches whose string occurs once in the reference and once in the query
// strand.  A match is NOT unique in the reference iff another match (other ref position) covers its whole query
// interval, and NOT unique in the query iff another match OF THE SAME QUERY RECORD covers its whole ref interval: `mummer`
// indexes the whole reference file (all records) and streams the query file one sequence and strand at a time, so
// mumuniqueinquery never sees the candidates of another query record — a repeat that sits once in each of two contigs of a
// draft assembly is unique in either (round 4: the filter used to scan the strand stream of all records as one sequence and
// lost those matches; found by fuzzing against the independent restatement the tests use, tools/anim_fuzz_multirecord.py).
// m[0..n) of ONE strand; the `strand` field is used as scratch flag and restored.  QREC(strand position) -> query record
// (any monotone labelling: only equality is used).  Returns the new count (order: by q).
template <typename QREC>
PG_HD int mum_filter(Match* m, int n, int strand, QREC&& qrec_of) {
  for (int i = 0; i < n; ++i) m[i].strand = 0;
  // query-interval containment (a query interval lies inside one record: the candidates of other records cannot cover it)
  heapsort(m, n, [](const Match& a, const Match& b) { return a.q < b.q || (a.q == b.q && a.len > b.len); });
  int32_t maxend = -1;
  for (int i = 0; i < n; ++i) {
    const int32_t e = m[i].q + m[i].len;
    if (e <= maxend) m[i].strand = 1;
    else if (i + 1 < n && m[i + 1].q == m[i].q && m[i + 1].len == m[i].len) m[i].strand = 1;
    if (e > maxend) maxend = e;
  }
  // ref-interval containment, per query record
  heapsort(m, n, [&](const Match& a, const Match& b) {
    const int32_t ra = qrec_of(a.q), rb = qrec_of(b.q);
    return ra != rb ? ra < rb : (a.r < b.r || (a.r == b.r && a.len > b.len)); });
  maxend = -1;
  int32_t seg = -1;
  for (int i = 0; i < n; ++i) {
    const int32_t e = m[i].r + m[i].len, rec = qrec_of(m[i].q);
    if (rec != seg) { seg = rec; maxend = -1; }
    if (e <= maxend) m[i].strand = 1;
    else if (i + 1 < n && m[i + 1].r == m[i].r && m[i + 1].len == m[i].len && qrec_of(m[i + 1].q) == rec) m[i].strand = 1;
    if (e > maxend) maxend = e;
  }
  int k = 0;
  for (int i = 0; i < n; ++i)
    if (!m[i].strand) m[k++] = m[i];
  for (int i = 0; i < k; ++i) m[i].strand = strand;
  heapsort(m, k, [](const Match& a, const Match& b) { return a.q < b.q || (a.q == b.q && a.r < b.r); });
  return k;
}

struct Chain {      // a cluster chain: matches cm[first .. first+count)
  int32_t first, count;
  int32_t strand;
  int32_t rrec, qrec;  // record indices (ref, query-forward)
};

// mgaps scans ALL earlier matches of a cluster for a match's best predecessor.  The chain DP here looks at the latest CHAIN_WINDOW
// live ones first (the wave form keeps exactly those in registers) and goes on to the rest only when they could matter: a
// predecessor j outside the window yields at most score[j] + len_i, and being earlier it wins ties, so the window's answer stands
// whenever  max(score of the entries that left the window) + len_i  <  the window's best candidate  — along a collinear run the
// scores grow by a match length per entry and the test never fails; when it does (round 5: it used to be a documented
// deviation, "the chain DP looks back 64 matches") the scan continues to the cluster's first entry.  Same answer as the full scan.
constexpr int CHAIN_WINDOW = 64;
#if !defined(__HIP_DEVICE_COMPILE__)
static long g_chain_full_scans = 0, g_chain_full_wins = 0;      // host statement only (development / tests): certificate failures, and how many changed the predecessor
#endif

// mgaps: union-find clustering of one strand's MUMs (sorted by q) + best-chain extraction.
// scratch: parent[n], score[n], from[n], adj[n], used[n] (int32 each).  Appends chains to chains[]/cm[].
// Matches of different QUERY records never join (mummer / mgaps run per query sequence).  The reference is ONE text to them —
// nucmer's prenuc joins the reference records with a separator, exactly the stream layout here — so a cluster may hold matches
// of two reference records (a query contig that runs across the junction of two adjacent reference contigs); postnuc then cuts
// such a cluster where the record changes: split_chains_by_ref_record below, applied after the -l 65 test on the whole.
PG_HD void mgaps_strand(const Match* m, int n, int strand, const int32_t* rrec_of, const int32_t* qrec_of,
                        int32_t* parent, int32_t* score, int32_t* from, int32_t* adj, int32_t* order,
                        Chain* chains, int& n_chains, int max_chains, Match* cm, int& n_cm, int max_cm) {
  auto find = [&](int x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  };
  for (int i = 0; i < n; ++i) parent[i] = i;
  for (int i = 0; i < n; ++i) {
    const int32_t iend = m[i].q + m[i].len, idiag = m[i].q - m[i].r;
    for (int j = i + 1; j < n; ++j) {
      const int32_t sep = m[j].q - iend;
      if (sep > MAX_GAP) break;
      if (qrec_of[i] != qrec_of[j]) continue;
      int32_t dd = (m[j].q - m[j].r) - idiag;
      if (dd < 0) dd = -dd;
      int32_t lim = (int32_t)(DIAG_FACTOR * sep);
      if (lim < DIAG_DIFF) lim = DIAG_DIFF;
      if (dd <= lim) { const int a = find(i), b = find(j); if (a != b) parent[a] = b; }
    }
  }
  for (int i = 0; i < n; ++i) { parent[i] = find(i); order[i] = i; }
  // group by cluster id, keep q order inside
  heapsort(order, n, [&](int a, int b) { return parent[a] < parent[b] || (parent[a] == parent[b] && a < b); });
  int g0 = 0;
  while (g0 < n) {
    int g1 = g0;
    while (g1 < n && parent[order[g1]] == parent[order[g0]]) ++g1;
    // cluster = order[g0..g1); extract chains until nothing is left (adj doubles as "removed" marker via from = -2)
    int remaining = g1 - g0;
    for (int k = g0; k < g1; ++k) from[order[k]] = -1;
    while (remaining > 0) {
      int best = -1;
      int win_tail = -1, win_n = 0;      // the window: the live entries at positions win_tail .. k - 1 (at most CHAIN_WINDOW of them)
      int32_t win_out = NEG_INF;         // the largest score among the live entries that have left it
      for (int k = g0; k < g1; ++k) {
        const int i = order[k];
        if (from[i] == -2) continue;
        score[i] = m[i].len; from[i] = -1; adj[i] = 0;
        int32_t bc = NEG_INF, bj = -1, bol = 0;   // best predecessor; equal scores: the EARLIEST one (mgaps scans j = 0 .. i-1 with a strict >)
        auto consider = [&](int j) {
          int32_t ol = m[j].r + m[j].len - m[i].r;
          if (ol < 0) ol = 0;
          const int32_t ol2 = m[j].q + m[j].len - m[i].q;
          if (ol2 > ol) ol = ol2;
          int32_t dd = (m[i].q - m[i].r) - (m[j].q - m[j].r);
          if (dd < 0) dd = -dd;
          const int32_t cand = score[j] + m[i].len - (ol + dd);
          if (cand >= bc) { bc = cand; bj = j; bol = ol; }
        };
        for (int kk = k - 1; kk >= win_tail && win_tail >= 0; --kk) if (from[order[kk]] != -2) consider(order[kk]);      // the window: near to far
        if (win_out > NEG_INF / 2 && win_out + m[i].len >= bc) {      // an entry that left the window could reach the best candidate: the rest
#if !defined(__HIP_DEVICE_COMPILE__)
          ++g_chain_full_scans;
          const int bj_w = bj;
#endif
          for (int kk = win_tail - 1; kk >= g0; --kk) if (from[order[kk]] != -2) consider(order[kk]);
#if !defined(__HIP_DEVICE_COMPILE__)
          if (bj != bj_w && bc > score[i]) ++g_chain_full_wins;
#endif
        }
        if (bc > score[i]) { score[i] = bc; from[i] = bj; adj[i] = bol; }
        if (best < 0 || score[i] > score[best]) best = i;
        if (win_n == 0) win_tail = k;
        if (win_n < CHAIN_WINDOW) ++win_n;
        else {      // the oldest entry leaves the window
          if (score[order[win_tail]] > win_out) win_out = score[order[win_tail]];
          do ++win_tail; while (from[order[win_tail]] == -2);
        }
      }
      // walk the chain
      int32_t total = 0, cnt = 0;
      for (int i = best; i >= 0; i = from[i]) { total += m[i].len; ++cnt; }
      if (total >= MIN_CLUSTER && n_chains < max_chains && n_cm + cnt <= max_cm) {
        Chain c;
        c.first = n_cm; c.count = cnt; c.strand = strand; c.rrec = rrec_of[best]; c.qrec = qrec_of[best];
        int pos = n_cm + cnt;
        for (int i = best; i >= 0; i = from[i]) {
          Match t = m[i];
          t.r += adj[i]; t.q += adj[i]; t.len -= adj[i];
          cm[--pos] = t;
        }
        n_cm += cnt;
        chains[n_chains++] = c;
      }
      for (int i = best; i >= 0;) { const int nx = from[i]; from[i] = -2; --remaining; i = nx; }
    }
    g0 = g1;
  }
}

// postnuc reads mgaps' clusters in joined-reference coordinates and maps every match back to its record: a cluster whose
// matches lie in more than one reference record becomes one cluster per run of matches of the same record (the -l 65 test was
// applied to the whole by mgaps).  In place: the chains keep their order, the pieces of a cut chain take consecutive slots
// (chains[] must have room for one chain per match).  RREC(ref stream position) -> reference record.  Returns the new count.
template <typename RREC>
PG_HD int split_chains_by_ref_record(Chain* chains, int n_chains, const Match* cm, RREC&& rrec_of) {
  int extra = 0;
  for (int c = 0; c < n_chains; ++c)
    for (int k = 1; k < chains[c].count; ++k)
      if (rrec_of(cm[chains[c].first + k].r) != rrec_of(cm[chains[c].first + k - 1].r)) ++extra;
  if (!extra) return n_chains;
  int w = n_chains + extra;
  for (int c = n_chains - 1; c >= 0; --c) {        // from the back: a chain's pieces land at slots >= its own
    const Chain C = chains[c];
    int end = C.first + C.count;
    for (int k = C.count - 1; k >= 0; --k) {
      const bool cut = k == 0 || rrec_of(cm[C.first + k].r) != rrec_of(cm[C.first + k - 1].r);
      if (!cut) continue;
      Chain p = C;
      p.first = C.first + k; p.count = end - p.first; p.rrec = rrec_of(cm[p.first].r);
      chains[--w] = p;
      end = p.first;
    }
  }
  return n_chains + extra;
}

// ---- chains in reference order (the extension stage that turns them into alignments is pg_nucmer_core.h) --------------------
// Reference order of the chains: by the reference start of the first match.  Chains that start on the same reference base (only
// --maxmatch produces them: one reference copy anchored by several query copies) have NO defined order in MUMmer — postnuc sorts
// its clusters with std::sort, which is not stable, over an input order that depends on mgaps' union-by-size roots — so the engine
// and the tests' CPU checker of nucmer both put them in a canonical one: by the query-strand start of the first match (two chains of one strand
// cannot share both starts), then by index.  A TOTAL order that does not depend on how a cluster stage numbers its components, so
// every form of it (radix sorts on the GPU, heapsort, std::sort) lists the chains identically.
PG_HD bool chain_before(const Chain* chains, const Match* cm, int a, int b) {
  const Match &ma = cm[chains[a].first], &mb = cm[chains[b].first];
  return ma.r != mb.r ? ma.r < mb.r : (ma.q != mb.q ? ma.q < mb.q : a < b);
}

// delta-filter -1 (1-to-1: intersection of the best alignment sets on the reference and on the query), restated from
// MUMmer 3.23's published algorithm (DeltaGraph_t::flagRLIS / flagQLIS + ScoreLocal): per sequence, alignments sorted by
// start (ties: input order); weighted LIS with integer scores  score_i = max(own_i, max_j<i score_j + gain(i, j)),
// own_i = trunc(len_i * idy_i^2), gain = trunc((len_i - olap) * idy_i^2), and a predecessor j is not allowed when the
// overlap exceeds LIS_MAX_OLAP of either alignment; first best wins (strict >).  LIS_MAX_OLAP is 100 %: the -1 path of
// MUMmer 3.23's delta-filter does not apply the 75 % that its usage text gives for -o (measured on the 27 real
// .delta -> .filter pairs of tests/golden/anim: 100 reproduces all 12 734 keep/drop decisions, 75 gets 32 of them wrong —
// overlaps between 75 % and 100 % do occur there, tests/test_anim_cpu.py::test_one_to_one_filter_matches_delta_filter).
// side 0 = reference coordinates, 1 = query coordinates.  idx: scratch order; sc (as int64) / from: scratch.
constexpr double LIS_MAX_OLAP = 100.0;
PG_HD double lis_idy(const Aln& a) {
  const double tot = (double)((a.re - a.rs) + (a.qe - a.qs));
  return tot > 0 ? 1.0 - 2.0 * a.errors / tot : 0.0;
}
PG_HD int64_t lis_gain(int64_t len_i, int64_t len_j, int64_t olap, double idy_i, bool& allowed) {
  allowed = !(olap > 0 && ((double)olap / (double)len_i * 100.0 > LIS_MAX_OLAP || (double)olap / (double)len_j * 100.0 > LIS_MAX_OLAP));
  return (int64_t)((double)(len_i - olap) * (idy_i * idy_i));
}
PG_HD void lis_filter(Aln* a, int n, int side, const int32_t* grp, const int32_t* ogrp, int32_t* idx, double* sc_, int32_t* from) {
  int64_t* sc = reinterpret_cast<int64_t*>(sc_);
  auto lo = [&](int i) { return side == 0 ? a[i].rs : a[i].qs; };   // a[] carries FORWARD query coordinates here
  auto hi = [&](int i) { return side == 0 ? a[i].re : a[i].qe; };
  auto olo = [&](int i) { return side == 0 ? a[i].qs : a[i].rs; };   // the start on the OTHER sequence (ogrp: its record)
  bool ok;
  for (int i = 0; i < n; ++i) { idx[i] = i; sc[i] = lis_gain(hi(i) - lo(i), 1, 0, lis_idy(a[i]), ok); }   // own scores
  // by start; equal starts: the higher-scoring alignment first — with the fixtures' equal-start pairs this is the order that
  // reproduces delta-filter's choices.  Equal start AND equal score (two copies of a duplicated region): delta-filter's std::sort
  // leaves them in an order MUMmer does not define; here, so that the result does not depend on how the alignments were listed
  // (by strand on the GPU, by record pair in a .delta file): by the other sequence's record, then the start there, then strand
  heapsort(idx, n, [&](int x, int y) {
    if (grp[x] != grp[y]) return grp[x] < grp[y];
    if (lo(x) != lo(y)) return lo(x) < lo(y);
    if (sc[x] != sc[y]) return sc[x] > sc[y];
    if (ogrp[x] != ogrp[y]) return ogrp[x] < ogrp[y];
    if (olo(x) != olo(y)) return olo(x) < olo(y);
    if (a[x].strand != a[y].strand) return a[x].strand < a[y].strand;
    return x < y; });
  int g0 = 0;
  while (g0 < n) {
    int g1 = g0;
    while (g1 < n && grp[idx[g1]] == grp[idx[g0]]) ++g1;
    int best = -1;
    for (int k = g0; k < g1; ++k) {
      const int i = idx[k];
      const int64_t len = hi(i) - lo(i);
      const double idy = lis_idy(a[i]);
      from[i] = -1;   // sc[i] = own score already
      for (int kk = g0; kk < k; ++kk) {
        const int j = idx[kk];
        int64_t ol = (int64_t)hi(j) - lo(i);
        if (ol < 0) ol = 0;
        const int64_t g = lis_gain(len, hi(j) - lo(j), ol, idy, ok);
        if (!ok) continue;
        const int64_t cand = sc[j] + g;
        if (cand > sc[i]) { sc[i] = cand; from[i] = j; }
      }
      if (best < 0 || sc[i] > sc[best]) best = i;
    }
    for (int i = best; i >= 0; i = from[i]) a[i].keep |= (1 << side);
    g0 = g1;
  }
}

struct PairResult {
  int64_t ref_aln_len, qry_aln_len, sim_errors, n_alignments;
  int64_t weighted, aligned;  // identity = weighted / aligned (one correctly rounded division, anim.py:396)
};

// pyani.anim.parse_delta over the kept alignments (anim.py:355-411): sums + per-sequence interval unions of the
// 1-based closed intervals.  a[] must carry FORWARD coordinates; rgrp/qgrp = record ids; idx: scratch.
PG_HD PairResult reduce_pair(const Aln* a, int n, const int32_t* rgrp, const int32_t* qgrp, int32_t* idx) {
  PairResult r{0, 0, 0, 0, 0, 0};
  int k = 0;
  for (int i = 0; i < n; ++i)
    if (a[i].keep == 3) {
      const int64_t rl = a[i].re - a[i].rs, ql = a[i].qe - a[i].qs;
      r.aligned += rl + ql;
      r.sim_errors += a[i].errors;
      r.weighted += rl + ql - 2 * (int64_t)a[i].errors;
      r.n_alignments += 1;
      idx[k++] = i;
    }
  for (int side = 0; side < 2; ++side) {
    auto grp = [&](int i) { return side == 0 ? rgrp[i] : qgrp[i]; };
    auto lo = [&](int i) { return side == 0 ? a[i].rs : a[i].qs; };
    auto hi = [&](int i) { return side == 0 ? a[i].re : a[i].qe; };
    heapsort(idx, k, [&](int x, int y) { return grp(x) < grp(y) || (grp(x) == grp(y) && lo(x) < lo(y)); });
    int64_t total = 0;
    int32_t cb = 0, ce = 0, cg = -1;
    bool open = false;
    for (int t = 0; t < k; ++t) {
      const int i = idx[t];
      // closed 1-based interval [lo+1, hi]; intervals that overlap or share an end point are merged (anim.py:399-409)
      if (open && grp(i) == cg && lo(i) + 1 <= ce) { if (hi(i) > ce) ce = hi(i); }
      else { if (open) total += ce - cb; cb = lo(i); ce = hi(i); cg = grp(i); open = true; }
    }
    if (open) total += ce - cb;
    if (side == 0) r.ref_aln_len = total; else r.qry_aln_len = total;
  }
  return r;
}

}  // namespace pga
