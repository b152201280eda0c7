// pg_anim_core.h — per-pair logic of the ANIm engine (MUM filter, mgaps-style clustering, banded affine extension,
// 1-to-1 filter, parse_delta reduction) as plain C++ that compiles for the device (hipcc) AND for the host.
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
// A target search reaches its target only if the target's anti-diagonal lies fewer than BREAK_LEN - TARGET_SLACK steps past the
// best cell: out of sample (round 2) MUMmer left every junction unfused whose target sat exactly 199 steps past the best cell
// and fused the ones at 198 (host sweep over all 25 192 fixture records: slack 0 / 1 / 2 -> 25 006 / 25 022 / 25 014 exact).
#ifdef PGA_TARGET_SLACK
constexpr int TARGET_SLACK = PGA_TARGET_SLACK;
#else
constexpr int TARGET_SLACK = 1;
#endif
constexpr int SC_MATCH = 3, SC_MISMATCH = -7, SC_GAP_OPEN = -10, SC_GAP_EXT = -7;
constexpr int BAND = 64;             // DP band: diagonal offsets -32 .. +31 around the start diagonal

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

// ---- banded affine extension -----------------------------------------------------------------------------------
// Anti-diagonal order (d = i + j), band of BAND diagonals k = j - i in [-BAND/2, BAND/2).  This is the SCALAR statement
// of the algorithm; pg_anim.hip holds the wave-cooperative version (one lane per diagonal, neighbours through DPP)
// that computes exactly the same cells, checks and tie-breaks — the two must stay in lock-step.
//   H: best score of a path ending at (i, j);  X: ending in a gap that consumed a ref base;  Y: ... a query base.
//   X = max(H(i-1,j) + OPEN, X(i-1,j) + EXT)   (ties: open)      Y likewise from (i,j-1)
//   H = max(diag + match/mismatch, X, Y)        (ties: diag, then X, then Y);  errors ride along.
// Free search: the end is the best cell (ties: larger d, then larger k).  Every CHECK_EVERY anti-diagonals the search
// stops once the best cell lies BREAK_LEN anti-diagonals back (nucmer -b; fitted: ">= 200", a tie at step 201 is too late) or no cell is alive.
// Target search (tr >= 0): runs to d = tr + tq and reports whether the target cell was reached by a live path,
// subject to the same break rule on the way (and to TARGET_SLACK at the target itself).
struct ExtResult {
  int32_t di, dj;      // bases consumed on ref / query at the chosen end
  int32_t score, errors;
  int32_t reached;     // target reached (only when a target was given)
};

constexpr int32_t NEG_INF = -(1 << 28);
constexpr int CHECK_EVERY = 1;
constexpr int TARGET_TRIM_MAX = 19;  // a following chain whose first match overlaps this one's end by < MIN_MATCH bases is still a target
constexpr int GAP_DIAG_MAX = 64;  // same-diagonal gaps up to this length are first tried as pure substitutions

struct DpCell { int32_t h, he, x, xe, y, ye; };

// One cell update, shared by the scalar and the wave version.  up = cell (i-1, j), left = cell (i, j-1),
// diag_h/diag_he = H of (i-1, j-1); ok = the two bases match.
PG_HD DpCell dp_cell(bool has_up, int32_t up_h, int32_t up_he, int32_t up_x, int32_t up_xe, bool has_left, int32_t left_h,
                     int32_t left_he, int32_t left_y, int32_t left_ye, bool has_diag, int32_t diag_h, int32_t diag_he, bool ok) {
  // Every choice is the lexicographic maximum of (score, -errors): among equally scoring paths the one with fewer
  // errors wins.  (This is what lets the device keep score and errors in one 32-bit key and use plain integer max.)
  DpCell c{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0};
  if (has_up) {
    const int32_t ho = up_h + SC_GAP_OPEN, xo = up_x + SC_GAP_EXT;
    if (ho > xo || (ho == xo && up_he <= up_xe)) { c.x = ho; c.xe = up_he + 1; } else { c.x = xo; c.xe = up_xe + 1; }
    if (c.x < NEG_INF / 2) c.x = NEG_INF;
  }
  if (has_left) {
    const int32_t ho = left_h + SC_GAP_OPEN, yo = left_y + SC_GAP_EXT;
    if (ho > yo || (ho == yo && left_he <= left_ye)) { c.y = ho; c.ye = left_he + 1; } else { c.y = yo; c.ye = left_ye + 1; }
    if (c.y < NEG_INF / 2) c.y = NEG_INF;
  }
  if (has_diag && diag_h > NEG_INF / 2) { c.h = diag_h + (ok ? SC_MATCH : SC_MISMATCH); c.he = diag_he + (ok ? 0 : 1); }
  if (c.x > c.h || (c.x == c.h && c.x > NEG_INF / 2 && c.xe < c.he)) { c.h = c.x; c.he = c.xe; }
  if (c.y > c.h || (c.y == c.h && c.y > NEG_INF / 2 && c.ye < c.he)) { c.h = c.y; c.he = c.ye; }
  return c;
}

template <typename RefT, typename QryT>
PG_HD ExtResult extend_banded(const RefT& R, const QryT& Q, int64_t r0, int64_t q0, int dir, int32_t rmax, int32_t qmax,
                              int32_t tr, int32_t tq) {
  constexpr int W = BAND / 2;
  DpCell cur[BAND];                 // latest cell of every diagonal (index l <-> k = l - W)
  int32_t bs[BAND], bd[BAND], be[BAND];  // per-diagonal best: score, d, errors
  for (int l = 0; l < BAND; ++l) { cur[l] = DpCell{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0}; bs[l] = NEG_INF; bd[l] = 0; be[l] = 0; }
  ExtResult res{0, 0, 0, 0, 0};
  bool targeted = tr >= 0;
  // band placement: diagonals k = l - W + koff.  A free search is centred on the start diagonal; a target search is
  // centred between the start diagonal and the target's, so diagonal shifts of up to ~60 bases can be bridged.
  int koff = 0;
  if (targeted) {
    koff = (tq - tr) / 2;
    if (koff > W - 2) koff = W - 2;
    if (koff < -(W - 2)) koff = -(W - 2);
    const int lt = (tq - tr) - koff + W;
    if (lt < 0 || lt >= BAND || tr > rmax || tq > qmax) { targeted = false; koff = 0; }   // unreachable: free search
  }
  if (targeted && tr == 0 && tq == 0) { res.reached = 1; return res; }
  cur[W - koff].h = 0; bs[W - koff] = 0;   // cell (0, 0) lies on diagonal 0
  const int32_t d_end = targeted ? tr + tq : rmax + qmax;
  int32_t gbest = 0, gbest_d = 0;
  for (int32_t d = 1; d <= d_end; ++d) {
    DpCell nxt[BAND];
    bool alive = false;
    for (int l = 0; l < BAND; ++l) {
      const int k = l - W + koff;
      if ((d + k) & 1) { nxt[l] = cur[l]; continue; }        // this diagonal has no cell on anti-diagonal d
      const int32_t i = (d - k) / 2, j = (d + k) / 2;
      if (i < 0 || j < 0 || i > rmax || j > qmax) { nxt[l] = DpCell{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0}; continue; }
      const bool has_up = i >= 1 && l + 1 < BAND, has_left = j >= 1 && l >= 1, has_diag = i >= 1 && j >= 1;
      bool ok = false;
      if (has_diag) {
        const int64_t rp = dir > 0 ? r0 + (i - 1) : r0 - i, qp = dir > 0 ? q0 + (j - 1) : q0 - j;
        ok = R.clean(rp) && Q.clean(qp) && R.base(rp) == Q.base(qp);
      }
      const DpCell& U = cur[has_up ? l + 1 : l];
      const DpCell& L = cur[has_left ? l - 1 : l];
      nxt[l] = dp_cell(has_up, U.h, U.he, U.x, U.xe, has_left, L.h, L.he, L.y, L.ye, has_diag, cur[l].h, cur[l].he, ok);
      if (nxt[l].h > NEG_INF / 2) {
        alive = true;
        if (nxt[l].h > bs[l] || (nxt[l].h == bs[l] && d >= bd[l])) { bs[l] = nxt[l].h; bd[l] = d; be[l] = nxt[l].he; }
      }
    }
    for (int l = 0; l < BAND; ++l) cur[l] = nxt[l];
    if ((d % CHECK_EVERY) == 0 || d == d_end) {
      gbest = NEG_INF; gbest_d = 0;
      for (int l = 0; l < BAND; ++l)
        if (bs[l] > gbest || (bs[l] == gbest && bd[l] >= gbest_d)) { gbest = bs[l]; gbest_d = bd[l]; }
      if (d - gbest_d >= BREAK_LEN) break;
      // `alive` of the last anti-diagonal only; two dead anti-diagonals in a row cannot revive
      bool any = alive;
      for (int l = 0; l < BAND && !any; ++l) any = cur[l].h > NEG_INF / 2;
      if (!any) break;
    }
    if (targeted && d == d_end) {
      const int l = (tq - tr) - koff + W;
      if (cur[l].h > NEG_INF / 2 && d - gbest_d < BREAK_LEN - TARGET_SLACK) { res.di = tr; res.dj = tq; res.score = cur[l].h; res.errors = cur[l].he; res.reached = 1; return res; }
    }
  }
  // best cell: max score, ties -> larger d, then larger k
  int bl = W; gbest = NEG_INF; gbest_d = -1;
  for (int l = 0; l < BAND; ++l)
    if (bs[l] > gbest || (bs[l] == gbest && bd[l] >= gbest_d)) { gbest = bs[l]; gbest_d = bd[l]; bl = l; }
  const int k = bl - W + koff;
  res.score = gbest; res.errors = be[bl]; res.di = (gbest_d - k) / 2; res.dj = (gbest_d + k) / 2;
  return res;
}

// Full (un-banded) global alignment of a small rectangle: ref [r0, r0+n) x query [q0, q0+m), forward direction, same
// recurrences and tie-breaks as dp_cell.  Used to bridge cluster junctions whose diagonal shift exceeds the band
// (an indel of 60+ bases between two clusters that nucmer still fuses).  min(n, m) <= THIN_MAX; returns -1 otherwise.
#ifdef PGA_THIN_LONG
constexpr int THIN_MAX = 63, THIN_LONG = PGA_THIN_LONG;
#else
constexpr int THIN_MAX = 63, THIN_LONG = 511;
#endif
// THIN_WHOLE: the same DP with the shorter side up to two wave strips, for the ERROR COUNT of a bridged junction over the whole
// junction (between the last match of one chain and the first match of the next); which junctions are bridged is still decided
// on rectangles of at most THIN_MAX (host sweep, round 2: 63 / 126 / 189 -> 25 022 / 25 038 / 25 034 records exact).
#ifdef PGA_THIN_WHOLE
constexpr int THIN_WHOLE = PGA_THIN_WHOLE;
#else
constexpr int THIN_WHOLE = 2 * THIN_MAX;
#endif
struct RectResult { int32_t score, errors; };   // errors < 0: the rectangle is too large for the thin DP
template <typename RefT, typename QryT>
PG_HD RectResult thin_rect_errors(const RefT& R, const QryT& Q, int64_t r0, int32_t n, int64_t q0, int32_t m, int32_t max_short = THIN_MAX) {
  if (n < 0 || m < 0 || n > THIN_LONG || m > THIN_LONG || (n > max_short && m > max_short)) return RectResult{NEG_INF, -1};
  // rows = ref bases, columns = query bases (no transposition: both builds walk the same cells)
  DpCell row[THIN_LONG + 1], nrow[THIN_LONG + 1];
  row[0] = DpCell{0, 0, NEG_INF, 0, NEG_INF, 0};
  for (int32_t j = 1; j <= m; ++j)
    row[j] = dp_cell(false, 0, 0, 0, 0, true, row[j - 1].h, row[j - 1].he, row[j - 1].y, row[j - 1].ye, false, 0, 0, false);
  for (int32_t i = 1; i <= n; ++i) {
    nrow[0] = dp_cell(true, row[0].h, row[0].he, row[0].x, row[0].xe, false, 0, 0, 0, 0, false, 0, 0, false);
    for (int32_t j = 1; j <= m; ++j) {
      const bool ok = R.clean(r0 + i - 1) && Q.clean(q0 + j - 1) && R.base(r0 + i - 1) == Q.base(q0 + j - 1);
      nrow[j] = dp_cell(true, row[j].h, row[j].he, row[j].x, row[j].xe, true, nrow[j - 1].h, nrow[j - 1].he, nrow[j - 1].y,
                        nrow[j - 1].ye, true, row[j - 1].h, row[j - 1].he, ok);
    }
    for (int32_t j = 0; j <= m; ++j) row[j] = nrow[j];
  }
  return RectResult{row[m].h, row[m].he};
}

// Global alignment of the gap between two chained matches: ref gap n, query gap m (both small); returns errors.
template <typename RefT, typename QryT>
PG_HD int32_t gap_errors(const RefT& R, const QryT& Q, int64_t r0, int32_t n, int64_t q0, int32_t m) {
  if (n == 0) return m;
  if (m == 0) return n;
  if (n == m && n <= GAP_DIAG_MAX) {
    // A gap on one diagonal with e <= 2 mismatches: the straight path scores 3n - 10e >= 3n - 20, any path with an
    // insertion/deletion pair at most 3(n-1) - 20 -> the diagonal is the unique optimum and the DP would return e.
    int32_t err = 0;
    for (int32_t t = 0; t < n; ++t)
      err += (R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t)) ? 0 : 1;
    if (err <= 2) return err;
  }
  const ExtResult e = extend_banded(R, Q, r0, q0, +1, n, m, n, m);
  if (e.reached) return e.errors;
  // target outside the band (|n - m| >= BAND/2) or pruned: count the diagonal part + the length difference
  int32_t k = n < m ? n : m, err = (n > m ? n - m : m - n);
  for (int32_t t = 0; t < k; ++t) {
    const bool ok = R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t);
    err += ok ? 0 : 1;
  }
  return err;
}

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

constexpr int CHAIN_LOOKBACK = 64;  // mgaps scans all earlier matches of the cluster; we bound the scan (documented)

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
      for (int k = g0; k < g1; ++k) {
        const int i = order[k];
        if (from[i] == -2) continue;
        score[i] = m[i].len; from[i] = -1; adj[i] = 0;
        int seen = 0;
        int32_t bc = NEG_INF, bj = -1, bol = 0;   // best predecessor; equal scores: the EARLIEST one (mgaps scans j = 0 .. i-1 with a strict >)
        for (int kk = k - 1; kk >= g0 && seen < CHAIN_LOOKBACK; --kk) {
          const int j = order[kk];
          if (from[j] == -2) continue;
          ++seen;
          int32_t ol = m[j].r + m[j].len - m[i].r;
          if (ol < 0) ol = 0;
          const int32_t ol2 = m[j].q + m[j].len - m[i].q;
          if (ol2 > ol) ol = ol2;
          int32_t dd = (m[i].q - m[i].r) - (m[j].q - m[j].r);
          if (dd < 0) dd = -dd;
          const int32_t cand = score[j] + m[i].len - (ol + dd);
          if (cand >= bc) { bc = cand; bj = j; bol = ol; }
        }
        if (bc > score[i]) { score[i] = bc; from[i] = bj; adj[i] = bol; }
        if (best < 0 || score[i] > score[best]) best = i;
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

// ---- chains -> alignments -------------------------------------------------------------------------------------------
// postnuc semantics reconstructed from the fixtures: every chain is extended forward freely (stop = best-scoring
// cell once BREAK_LEN anti-diagonals pass without a new high score); it is extended BACKWARD towards the end of the
// preceding alignment as a target — if the target cell is reached before the break criterion fires, the two are
// fused into one alignment (this is how nucmer bridges ~100-base junk between two clusters), otherwise the chain
// starts a new alignment at its own best backward cell.
// Reference order of the chains: by the start of the first match; chains that start on the same reference base (only with
// --maxmatch: one reference copy anchored by several query copies) in the order they were extracted — a TOTAL order, so
// that every form of the cluster stage (radix sort of (r, chain), heapsort, std::sort) lists them identically.
PG_HD bool chain_before(const Chain* chains, const Match* cm, int a, int b) {
  const int32_t ra = cm[chains[a].first].r, rb = cm[chains[b].first].r;
  return ra != rb ? ra < rb : a < b;
}
// nearest predecessor (chain_is_predecessor) / following chain of the same (ref record, query record) in ref order (looks 8
// entries each way)
// Chain p can be the predecessor of chain c (the alignment c's backward search aims at, may fuse with, and must not run into):
// same records, and p's last match ends before c's first match starts in BOTH sequences, give or take the overlap a target may
// have (TARGET_TRIM_MAX, as forward_target).  The chain before c in reference order may belong to another copy of a repeat,
// far ahead in the query: with it as "predecessor" c never met the collinear chain one or two places further back that MUMmer
// fuses it with (round 2, out of sample: +6 records; the bound itself is not sensitive, 0 / 19 / 100 / 10^5 -> +6 / +6 / +8 / +8).
PG_HD bool chain_is_predecessor(const Chain* chains, const Match* cm, int p, int c) {
  if (chains[p].rrec != chains[c].rrec || chains[p].qrec != chains[c].qrec) return false;
  const Match& l = cm[chains[p].first + chains[p].count - 1];
  const Match& f = cm[chains[c].first];
  return l.r + l.len <= f.r + TARGET_TRIM_MAX && l.q + l.len <= f.q + TARGET_TRIM_MAX;
}
PG_HD void chain_neighbours(const Chain* chains, const Match* cm, const int32_t* order, int n, int32_t* prev_of, int32_t* next_of) {
  for (int k = 0; k < n; ++k) {
    const int c = order[k];
    int p = -1, q = -1;
    for (int kk = k - 1; kk >= 0 && kk >= k - 8 && p < 0; --kk)
      if (chain_is_predecessor(chains, cm, order[kk], c)) p = order[kk];
    for (int kk = k + 1; kk < n && kk <= k + 8 && q < 0; ++kk)
      if (chains[order[kk]].rrec == chains[c].rrec && chains[order[kk]].qrec == chains[c].qrec) q = order[kk];
    prev_of[c] = p;
    next_of[c] = q;
  }
}

struct ChainFwd {
  int32_t first_r, first_q;     // first match start
  int32_t inner_err;            // errors of the gaps between chained matches
  int32_t lr, lq;               // end of the last chained match
  int32_t re, qe, err_fwd;      // end after the forward extension (== next chain's first match when reached)
  int32_t reached;              // forward extension landed exactly on chain `target`'s first match -> fuse
  int32_t target;               // the first following chain that lies strictly ahead in both sequences (or -1)
};
struct ChainBwd {
  int32_t rs, qs, err_back;     // start after the backward extension (== target cell when reached)
  int32_t reached;              // landed exactly on the previous chain's forward end -> fuse
                                // (2 = through the thin-rectangle bridge: err_back already includes the bridge)
};

// Gap fills between the chained matches; returns the end of the last match through er/eq.
template <typename RefT, typename QryT>
PG_HD int32_t chain_inner_errors(const RefT& R, const QryT& Q, const Match* cm, const Chain& c, int32_t& er, int32_t& eq) {
  const Match& f = cm[c.first];
  int32_t inner = 0;
  er = f.r + f.len; eq = f.q + f.len;
  for (int k = 1; k < c.count; ++k) {
    Match t = cm[c.first + k];
    int32_t trim = er - t.r;                 // chained matches may still overlap the running end
    if (eq - t.q > trim) trim = eq - t.q;
    if (trim > 0) { t.r += trim; t.q += trim; t.len -= trim; }
    if (t.len <= 0) continue;
    inner += gap_errors(R, Q, er, t.r - er, eq, t.q - eq);
    er = t.r + t.len; eq = t.q + t.len;
  }
  return inner;
}

// Forward target of a chain that ends at (er, eq): the start of the following chain's first match.  If that match
// overlaps this chain's end in ONE sequence by fewer than MIN_MATCH bases (a short tandem repeat at the edge of an
// indel), the target is the match trimmed by the overlap, as between the matches of one cluster (fixture: Blochmannia
// NC_007292 / NC_020075, one 791 kb alignment across a 46-base insertion flanked by an 11-mer repeat); if it overlaps in
// BOTH sequences the two clusters stay two alignments (fixture: the Caulobacter pair around 3.73 Mb).
PG_HD void forward_target(int32_t er, int32_t eq, int32_t nr, int32_t nq, int32_t nlen, int32_t& tr, int32_t& tq) {
  tr = -1; tq = -1;
  if (nr < 0 || (nr < er && nq < eq)) return;   // overlapping in BOTH sequences: the two clusters stay two alignments
  int32_t trim = er - nr;
  if (eq - nq > trim) trim = eq - nq;
  if (trim < 0) trim = 0;
  if (trim >= nlen || trim > TARGET_TRIM_MAX) return;
  tr = nr + trim - er; tq = nq + trim - eq;
}

// MUMmer's DP works on at most MAX_ALIGNMENT_LENGTH = 10000 bases per call; an extension off a cluster end therefore
// never exceeds 9999 bases in either direction — visible in the fixtures as alignments that stop exactly there.
constexpr int32_t MAX_EXT_FWD = 9999, MAX_EXT_BWD = 9999;
PG_HD int32_t cap_ext(int32_t v, int32_t cap) { return v < cap ? v : cap; }

// Forward target: walk the following chains (same strand and records, ref order) and take the first whose first match
// lies strictly ahead of (er, eq) in both sequences and within MUMmer's 10 kb DP limit; chains skipped on the way
// overlap this one and end up shadowed.
PG_HD int32_t pick_forward_target(const Chain* chains, const Match* cm, const int32_t* next_of, int c, int32_t er, int32_t eq,
                                  int32_t& nr, int32_t& nq) {
  nr = -1; nq = -1;
  int t = next_of[c];
  for (int hops = 0; t >= 0 && hops < 8; ++hops, t = next_of[t]) {
    const Match& nf = cm[chains[t].first];
    int32_t tr, tq;
    forward_target(er, eq, nf.r, nf.q, nf.len, tr, tq);
    if (tr >= 0) { nr = er + tr; nq = eq + tq; return t; }
  }
  return -1;
}

// Forward extension off the end (er, eq) of a chain towards the target match start (nr, nq) (or freely if nr < 0).
// MUMmer's DP handles at most 10 kb per call: a farther target is approached in 10 kb free-search chunks, continuing
// while a chunk runs into its length limit (the region is still alignable) and giving up as soon as one ends earlier.
// EXT is the DP routine (scalar extend_banded on the host, the wave-cooperative one on the device).
constexpr int MAX_FWD_CHUNKS = 64;
template <typename EXT>
PG_HD void forward_extension(EXT&& ext, int32_t er, int32_t eq, int32_t r_hi, int32_t q_hi, int32_t nr, int32_t nq, int32_t& re,
                             int32_t& qe, int32_t& errors, int32_t& reached) {
  int32_t cr = er, cq = eq, err = 0;
  reached = 0;
  for (int chunk = 0; chunk < MAX_FWD_CHUNKS; ++chunk) {
    int32_t tr = -1, tq = -1;
    if (nr >= 0) { tr = nr - cr; tq = nq - cq; }
    const bool near = nr >= 0 && tr >= 0 && tq >= 0 && tr <= MAX_EXT_FWD && tq <= MAX_EXT_FWD;
    ExtResult x = ext(cr, cq, cap_ext(r_hi - cr, MAX_EXT_FWD), cap_ext(q_hi - cq, MAX_EXT_FWD), near ? tr : -1, near ? tq : -1);
    if (near && !x.reached && tr != tq)   // the band was shifted towards an unreachable target: search freely instead
      x = ext(cr, cq, cap_ext(r_hi - cr, MAX_EXT_FWD), cap_ext(q_hi - cq, MAX_EXT_FWD), -1, -1);
    err += x.errors; cr += x.di; cq += x.dj;
    if (near) { reached = x.reached; break; }
    const bool hit_cap = x.di >= MAX_EXT_FWD - 100 || x.dj >= MAX_EXT_FWD - 100;
    if (nr < 0 || tr < 0 || tq < 0 || !hit_cap) break;   // no target, or the chunk ended on its own: stop here
  }
  re = cr; qe = cq; errors = err;
}

template <typename RefT, typename QryT>
PG_HD ChainFwd extend_chain_fwd(const RefT& R, const QryT& Q, const Match* cm, const Chain* chains, const int32_t* next_of, int c,
                                int32_t r_hi, int32_t q_hi) {
  ChainFwd e;
  e.first_r = cm[chains[c].first].r; e.first_q = cm[chains[c].first].q;
  int32_t er, eq;
  e.inner_err = chain_inner_errors(R, Q, cm, chains[c], er, eq);
  e.lr = er; e.lq = eq;
  int32_t nr, nq;
  e.target = pick_forward_target(chains, cm, next_of, c, er, eq, nr, nq);
  forward_extension([&](int32_t cr, int32_t cq, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
                      return extend_banded(R, Q, cr, cq, +1, rmax, qmax, tr, tq); },
                    er, eq, r_hi, q_hi, nr, nq, e.re, e.qe, e.err_fwd, e.reached);
  return e;
}

// Junction with a diagonal shift beyond the band: the free backward search stopped at (e.rs, e.qs), its best cell.  nucmer's
// DP band is dynamic: it widens by one diagonal per anti-diagonal and is trimmed from its edges where the score has fallen
// more than  GOOD_SCORE * breaklen = 3 * 200 = 600  below the best cell, and the search ends breaklen anti-diagonals after the
// best cell.  So the previous alignment's end (prev_re, prev_qe) is reached — and the two alignments fused — iff it lies
// within the break length of the best cell (n + m <= 200) AND the optimal path over the residual rectangle between the two
// costs no more than the X-drop.  Out of sample (the ten 85 % Caulobacter pairs, 101 such junctions): nucmer fused every
// junction whose residual rectangle scores >= -612 and none below -623 (one exception at -605); 615 separates them, i.e.
// indels up to ~87 bases between two clusters are bridged, longer ones end the alignment.  Round 1 had no score test (fitted
// on the Blochmannia pairs, whose largest such indel is 80 bases) and fused 240 junctions nucmer does not.
// RECT(r0, n, q0, m) -> {score, errors} of the optimal global path (errors < 0: too large for the thin DP).
constexpr int32_t BRIDGE_XDROP =
#ifdef PGA_BRIDGE_XDROP
    PGA_BRIDGE_XDROP;
#else
    615;
#endif
template <typename RECT, typename RECTW>
PG_HD void bridge_junction(ChainBwd& e, int32_t prev_re, int32_t prev_qe, int32_t tr, int32_t tq, int32_t first_r, int32_t first_q,
                           int32_t prev_lr, int32_t prev_lq, int32_t prev_err_fwd, RECT&& rect, RECTW&& rect_whole) {
  if (e.reached || prev_re < 0 || tr < 0) return;
  int32_t shift = tq - tr;
  if (shift < 0) shift = -shift;
  if (shift < BAND - 2) return;                            // reachable shifts are decided by the (shifted-band) target search
#ifdef PGA_BRIDGE_SHIFT_MAX
  if (shift > PGA_BRIDGE_SHIFT_MAX) return;
#endif
  const int32_t n = e.rs - prev_re, m = e.qs - prev_qe;
  if (n < 0 || m < 0 || n + m > BREAK_LEN) return;
  const RectResult resid = rect(prev_re, n, prev_qe, m);
  if (resid.errors < 0 || resid.score < -BRIDGE_XDROP) return;
  // errors of the fused alignment: prefer the optimal path over the WHOLE junction, from the end of the previous chain's
  // last match to this chain's first match (its free forward extension is then replaced: minus prev_err_fwd); then from the
  // previous forward end; else the residual rectangle behind the free backward search
  RectResult x = prev_lr >= 0 && prev_lr <= prev_re && prev_lq <= prev_qe ? rect_whole(prev_lr, first_r - prev_lr, prev_lq, first_q - prev_lq)
                                                                           : RectResult{NEG_INF, -1};
  if (x.errors >= 0) { e.err_back = x.errors - prev_err_fwd; }
  else {
    x = rect(prev_re, tr, prev_qe, tq);
    if (x.errors >= 0) { e.err_back = x.errors; }
    else { e.err_back += resid.errors; }
  }
  e.rs = prev_re; e.qs = prev_qe; e.reached = 2;
}

// The backward search of a chain never enters the matches of the chain before it — if that chain is its collinear
// predecessor: its last match ends before this chain's first match in both sequences, on a diagonal the band can reach.  The
// chain before it in reference order may just as well belong to another copy of a repeat, hundreds of kilobases away in the
// query; stopping at ITS matches cut alignments short that MUMmer extends over them (round 2, out of sample: host sweep of
// the shift bound over 25 192 records, none / 61 / 100 / 1000 / unbounded -> 25 057 / 25 057 / 25 056 / 25 044 / 25 022 exact).
constexpr int PREV_LIMIT_SHIFT =
#ifdef PGA_PREV_LIMIT_SHIFT
    PGA_PREV_LIMIT_SHIFT;
#else
    BAND - 3;
#endif
PG_HD void limit_backward_by_prev(int32_t first_r, int32_t first_q, int32_t prev_lr, int32_t prev_lq, int32_t& r_lo, int32_t& q_lo) {
  if (prev_lr > first_r || prev_lq > first_q) return;
  int32_t sh = (first_q - prev_lq) - (first_r - prev_lr);
  if (sh < 0) sh = -sh;
  if (sh > PREV_LIMIT_SHIFT) return;
  if (prev_lr > r_lo) r_lo = prev_lr;
  if (prev_lq > q_lo) q_lo = prev_lq;
}

// prev_re/prev_qe: forward end of the preceding chain (same strand and records), or -1 if there is none;
// prev_lr/prev_lq: end of its last match — the backward search never needs to enter the previous chain's matches;
// prev_fr/prev_fq: its first match start; my_lr/my_lq: end of THIS chain's last match.  If the previous chain's span
// already covers this chain entirely, the stitch will shadow it and no backward search is needed at all.
template <typename RefT, typename QryT>
PG_HD ChainBwd extend_chain_bwd(const RefT& R, const QryT& Q, int32_t first_r, int32_t first_q, int32_t r_lo, int32_t q_lo,
                                int32_t prev_re, int32_t prev_qe, int32_t prev_lr, int32_t prev_lq, int32_t prev_fr,
                                int32_t prev_fq, int32_t my_lr, int32_t my_lq, bool prev_reached_me, int32_t prev_err_fwd) {
  // no search needed: the previous chain's forward extension already landed on this chain's first match (fusion),
  // or its span covers this chain entirely (the stitch will shadow it)
  if (prev_reached_me ||
      (prev_re >= 0 && prev_fr <= first_r && prev_fq <= first_q && prev_re >= my_lr && prev_qe >= my_lq)) {
    ChainBwd e;
    e.rs = first_r; e.qs = first_q; e.err_back = 0; e.reached = 0;
    return e;
  }
  int32_t tr = -1, tq = -1;
  if (prev_re >= 0 && first_r >= prev_re && first_q >= prev_qe) { tr = first_r - prev_re; tq = first_q - prev_qe; }
  if (prev_re >= 0) limit_backward_by_prev(first_r, first_q, prev_lr, prev_lq, r_lo, q_lo);
  ExtResult b = extend_banded(R, Q, first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD),
                              cap_ext(first_q - q_lo, MAX_EXT_BWD), tr, tq);
  if (tr >= 0 && !b.reached && tr != tq)   // the band was shifted towards an unreachable target: search freely instead
    b = extend_banded(R, Q, first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), -1, -1);
  ChainBwd e;
  e.rs = first_r - b.di; e.qs = first_q - b.dj; e.err_back = b.errors;
  e.reached = (tr >= 0 && b.reached) ? 1 : 0;
  bridge_junction(e, prev_re, prev_qe, tr, tq, first_r, first_q, prev_lr, prev_lq, prev_err_fwd,
                  [&](int32_t r0, int32_t n, int32_t q0, int32_t m) { return thin_rect_errors(R, Q, r0, n, q0, m); },
                  [&](int32_t r0, int32_t n, int32_t q0, int32_t m) { return thin_rect_errors(R, Q, r0, n, q0, m, THIN_WHOLE); });
  return e;
}

// Sequential stitch of one strand's chains (order[] = sorted by first-match ref start).  prev_of / next_of: the
// neighbouring chains of the same records in that order (or -1).  A chain is fused into the running alignment when
// the previous chain's forward extension reached its first match, or when its own backward extension reached the
// previous chain's forward end; chains lying inside an existing alignment are shadowed.
PG_HD int stitch_chains(const ChainFwd* fw, const ChainBwd* bw, const Match* cm, const Chain* chains, const int32_t* order,
                        const int32_t* prev_of, const int32_t* next_of, int n, int strand, int32_t* aln_of, Aln* out, int n_out,
                        int max_out) {
  const int out0 = n_out;
  for (int k = 0; k < n; ++k) aln_of[order[k]] = -1;
  for (int k = 0; k < n; ++k) {
    const int c = order[k];
    if (aln_of[c] >= 0) continue;                       // already fused forward into an earlier alignment
    const Match& l = cm[chains[c].first + chains[c].count - 1];
    const int p = prev_of[c];
    int ai = -1;
    if (p >= 0 && bw[c].reached && aln_of[p] >= 0 && out[aln_of[p]].re == fw[p].re && out[aln_of[p]].qe == fw[p].qe) {
      ai = aln_of[p];                                    // bridge the junk between the two chains
      out[ai].errors += bw[c].err_back + fw[c].inner_err;
    } else {
      bool shadow = false;
      for (int t = out0; t < n_out && !shadow; ++t)
        if (fw[c].first_r >= out[t].rs && l.r + l.len <= out[t].re && fw[c].first_q >= out[t].qs && l.q + l.len <= out[t].qe) {
          shadow = true;
          aln_of[c] = t;
        }
      if (shadow) continue;
      if (n_out >= max_out) continue;
      Aln a;
      a.rs = bw[c].rs; a.qs = bw[c].qs; a.re = a.rs; a.qe = a.qs; a.strand = strand; a.keep = 0;
      a.errors = bw[c].err_back + fw[c].inner_err;
      ai = n_out;
      out[n_out++] = a;
    }
    int cur = c;
    for (;;) {
      out[ai].errors += fw[cur].err_fwd;
      out[ai].re = fw[cur].re; out[ai].qe = fw[cur].qe;
      aln_of[cur] = ai;
      if (!fw[cur].reached) break;
      const int t = fw[cur].target;
      if (t < 0 || aln_of[t] >= 0) break;
      out[ai].errors += fw[t].inner_err;
      cur = t;
    }
  }
  return n_out;
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
PG_HD void lis_filter(Aln* a, int n, int side, const int32_t* grp, int32_t* idx, double* sc_, int32_t* from) {
  int64_t* sc = reinterpret_cast<int64_t*>(sc_);
  auto lo = [&](int i) { return side == 0 ? a[i].rs : a[i].qs; };   // a[] carries FORWARD query coordinates here
  auto hi = [&](int i) { return side == 0 ? a[i].re : a[i].qe; };
  bool ok;
  for (int i = 0; i < n; ++i) { idx[i] = i; sc[i] = lis_gain(hi(i) - lo(i), 1, 0, lis_idy(a[i]), ok); }   // own scores
  // by start; equal starts: the higher-scoring alignment first (then input order) — with the fixtures' equal-start
  // pairs this is the order that reproduces delta-filter's choices
  heapsort(idx, n, [&](int x, int y) {
    if (grp[x] != grp[y]) return grp[x] < grp[y];
    if (lo(x) != lo(y)) return lo(x) < lo(y);
    if (sc[x] != sc[y]) return sc[x] > sc[y];
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
