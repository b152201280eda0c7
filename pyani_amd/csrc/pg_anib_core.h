// pg_anib_core.h — fragment mode of the aligner (BASELINE.json configs[4]): the search pyani's ANIb
// delegates to BLAST+,   blastn -task blastn -query <1020-nt fragments of genome Q> -db <genome S> -xdrop_gap_final 150
// -dust no -evalue 1e-15 -max_target_seqs 1   (pyani/anib.py:451-471), as plain C++ that compiles for the device (hipcc)
// AND for the host (tools/anib_debug, oracle/): every fragment gets its best local alignment against the subject genome,
// described by the columns pyani reads from the BLAST table (anib.py:609-624):
//     length (alignment columns), mismatch, gaps (gap characters), nident -> pident = 100 * nident / length,
//     qstart / qend (1-based in the fragment), sstart / send (1-based in the subject record, sstart > send on the minus strand)
// and those rows go through parse_blast_tab's arithmetic (anib.py:641-665 == pg_anib_reduce).
//
// BLAST+ itself is third-party and absent from the reference tree; what is restated here is its documented behaviour for
// this command line — blastn scoring  reward 2 / penalty -3 / gap open 5 / gap extend 2  (a gap of k bases costs 5 + 2k),
// a gapped local alignment grown from exact word seeds, highest-scoring HSP first in the table — and it is calibrated
// against the BLAST+ tables the reference's tests hold for the four Caulobacter genomes (tests/golden/anib/blastn/*.blast_tab).
//
// Method per (fragment, strand): the seeds (exact matches, from the same LDS-table seeding as ANIm) vote for a subject
// diagonal; a banded Smith-Waterman (FRAG_BAND diagonals around it, anti-diagonal order, affine gaps) finds the best local
// alignment, each DP state carrying the statistics of its best path (mismatches, gap bases, start cell), so no traceback
// is kept.  Ties: diagonal move, then gap in the subject (query base consumed), then gap in the query — and among cells of equal
// score the earliest anti-diagonal, then the lowest diagonal; host and device walk the same cells in the same order.
#pragma once
#include "pg_anim_core.h"

namespace pga {

constexpr int FRAG_SIZE = 1020;                 // pyani_config.FRAGSIZE
constexpr int FRAG_MATCH = 2, FRAG_MISMATCH = -3, FRAG_GAP_OPEN = -7, FRAG_GAP_EXT = -2;   // first gap base -(5 + 2), further -2
#ifndef PGA_FRAG_BAND
#define PGA_FRAG_BAND 64
#endif
constexpr int FRAG_BAND = PGA_FRAG_BAND;                   // diagonals of the DP band (one lane each on the device)
constexpr int FRAG_SLACK = 200;                // subject bases an extension may use beyond the fragment's own length

struct FragStat { int32_t mm, gaps; };            // mismatches and gap bases of the best path into a DP state
struct FragCell { int32_t h, x, y; FragStat hs, xs, ys; };   // H: any end; X: ends in a gap consuming a query base; Y: ... a subject base

struct FragExt {             // one directional extension off an anchor
  int32_t score;             // best score (>= 0; 0 = no extension)
  int32_t di, dj;            // query / subject bases consumed at the best cell
  int32_t mm, gaps;          // mismatches / gap bases on the way
};

struct FragHit {             // one HSP: the row of the BLAST table
  int32_t score;
  int32_t length, mismatch, gaps, nident;
  int32_t qs, qe;            // query (fragment) interval, 0-based half-open, on the searched strand of the fragment
  int32_t ss, se;            // subject interval, 0-based half-open, stream coordinates
};

// Gapped X-drop extension off an anchor, as BLAST grows an HSP from a seed: cell (0, 0) scores 0, qbase(t) / sbase(t) give the
// t-th base away from the anchor in the direction of the extension (0..3; 4 / 5 for dirty or out of range: never equal), at most
// qmax / smax bases.  Anti-diagonal order; the band covers the FRAG_BAND diagonals K = j - i in [koff, koff + FRAG_BAND), starts
// centred on the anchor's diagonal and FOLLOWS the alignment: every FRAG_TRACK anti-diagonals it is re-centred on the diagonal of
// the best live H (ties: lowest diagonal), by an even number of diagonals (the cell / anti-diagonal parity pattern is kept),
// at most FRAG_SHIFT_MAX; states that leave the band are lost, new ones start dead.  A cell more than FRAG_XDROP below the best
// score so far is dead (BLAST's -xdrop_gap_final 150 bits = 166 in raw 2 / -3 scores); the search ends when a whole anti-diagonal
// pair is dead.  The end is the best cell (ties: earliest anti-diagonal, then lowest diagonal).  On the device the 64 lanes of a
// wave are the 64 diagonals and a re-centring is one wave shift of the state registers.
#ifndef PGA_FRAG_XDROP
#define PGA_FRAG_XDROP 166
#endif
constexpr int FRAG_TRACK = 16, FRAG_SHIFT_MAX = 8, FRAG_XDROP = PGA_FRAG_XDROP;
template <typename QB, typename SB>
PG_HD FragExt frag_extend(QB&& qbase, int32_t qmax, SB&& sbase, int32_t smax) {
  constexpr int32_t NEG = -(1 << 28);
  FragCell cur[FRAG_BAND], prv[FRAG_BAND];   // latest cell of every diagonal (index k <-> diagonal K = k + koff)
  const FragStat z{0, 0};
  const FragCell dead{NEG, NEG, NEG, z, z, z};
  for (int k = 0; k < FRAG_BAND; ++k) cur[k] = dead;
  int32_t koff = -FRAG_BAND / 2;
  cur[0 - koff].h = 0;                       // the anchor cell (0, 0) on diagonal 0
  FragExt best{0, 0, 0, 0, 0};
  int dead_run = 0;
  for (int32_t d = 1; d <= qmax + smax; ++d) {
    if ((d % FRAG_TRACK) == 0) {             // re-centre the band on the best live H
      int32_t bh = NEG / 2, bk = -1;
      for (int k = 0; k < FRAG_BAND; ++k) if (cur[k].h > bh) { bh = cur[k].h; bk = k; }
      if (bk >= 0) {
        int32_t s = bk - FRAG_BAND / 2;
        if (s > FRAG_SHIFT_MAX) s = FRAG_SHIFT_MAX;
        if (s < -FRAG_SHIFT_MAX) s = -FRAG_SHIFT_MAX;
        s &= ~1;
        if (s != 0) {
          for (int k = 0; k < FRAG_BAND; ++k) prv[k] = cur[k];
          for (int k = 0; k < FRAG_BAND; ++k) { const int f = k + s; cur[k] = (f >= 0 && f < FRAG_BAND) ? prv[f] : dead; }
          koff += s;
        }
      }
    }
    for (int k = 0; k < FRAG_BAND; ++k) prv[k] = cur[k];
    bool alive = false;
    for (int k = 0; k < FRAG_BAND; ++k) {
      const int32_t K = k + koff;
      if ((d + K) & 1) continue;                       // no cell of this diagonal on this anti-diagonal
      const int32_t i = (d - K) / 2, j = (d + K) / 2;
      if (i < 0 || j < 0 || i > qmax || j > smax) { cur[k] = dead; continue; }
      FragCell c = dead;
      if (i >= 1 && k + 1 < FRAG_BAND) {               // X: gap consuming a query base, from (i-1, j): diagonal K+1
        const FragCell& u = prv[k + 1];
        const int32_t ho = u.h + FRAG_GAP_OPEN, xo = u.x + FRAG_GAP_EXT;
        if (u.x > NEG / 2 && xo >= ho) { c.x = xo; c.xs = u.xs; } else if (u.h > NEG / 2) { c.x = ho; c.xs = u.hs; }
        if (c.x > NEG / 2) c.xs.gaps += 1;
      }
      if (j >= 1 && k >= 1) {                          // Y: gap consuming a subject base, from (i, j-1): diagonal K-1
        const FragCell& l = prv[k - 1];
        const int32_t ho = l.h + FRAG_GAP_OPEN, yo = l.y + FRAG_GAP_EXT;
        if (l.y > NEG / 2 && yo >= ho) { c.y = yo; c.ys = l.ys; } else if (l.h > NEG / 2) { c.y = ho; c.ys = l.hs; }
        if (c.y > NEG / 2) c.ys.gaps += 1;
      }
      if (i >= 1 && j >= 1 && prv[k].h > NEG / 2) {    // H: diagonal move from (i-1, j-1): same diagonal, anti-diagonal d-2
        const int qb = qbase(i - 1), sb = sbase(j - 1);
        const bool ok = qb < 4 && qb == sb;
        c.h = prv[k].h + (ok ? FRAG_MATCH : FRAG_MISMATCH);
        c.hs = prv[k].hs;
        if (!ok) c.hs.mm += 1;
      }
      if (c.x > c.h) { c.h = c.x; c.hs = c.xs; }
      if (c.y > c.h) { c.h = c.y; c.hs = c.ys; }
      if (c.h < best.score - FRAG_XDROP) c.h = NEG;    // X-drop
      if (c.x < best.score - FRAG_XDROP) c.x = NEG;
      if (c.y < best.score - FRAG_XDROP) c.y = NEG;
      cur[k] = c;
      if (c.h > NEG / 2 || c.x > NEG / 2 || c.y > NEG / 2) alive = true;
      if (c.h > best.score) { best.score = c.h; best.di = i; best.dj = j; best.mm = c.hs.mm; best.gaps = c.hs.gaps; }
    }
    dead_run = alive ? 0 : dead_run + 1;
    if (dead_run >= 2) break;
  }
  return best;
}

// The HSP grown from an exact anchor  query [aq, aq + alen)  ==  subject [as, as + alen)  (both within their limits): leftward
// and rightward extension + the anchor itself.  q_at(p) / s_at(p): base at absolute query / subject position p (4 / 5 outside).
template <typename QA, typename SA>
PG_HD FragHit frag_hsp(QA&& q_at, int32_t qlen, SA&& s_at, int64_t s_lo, int64_t s_hi, int32_t aq, int64_t as, int32_t alen) {
  const FragExt R = frag_extend([&](int32_t t) { return q_at(aq + alen + t); }, qlen - (aq + alen),
                                [&](int32_t t) { return s_at(as + alen + t); }, (int32_t)(s_hi - (as + alen) < FRAG_SIZE + FRAG_SLACK ? s_hi - (as + alen) : FRAG_SIZE + FRAG_SLACK));
  const FragExt L = frag_extend([&](int32_t t) { return q_at(aq - 1 - t); }, aq,
                                [&](int32_t t) { return s_at(as - 1 - t); }, (int32_t)(as - s_lo < FRAG_SIZE + FRAG_SLACK ? as - s_lo : FRAG_SIZE + FRAG_SLACK));
  FragHit h;
  h.score = FRAG_MATCH * alen + L.score + R.score;
  h.qs = aq - L.di; h.qe = aq + alen + R.di;
  h.ss = (int32_t)(as - L.dj); h.se = (int32_t)(as + alen + R.dj);
  h.gaps = L.gaps + R.gaps; h.mismatch = L.mm + R.mm;
  const int32_t m = ((h.qe - h.qs) + (h.se - h.ss) - h.gaps) / 2;       // diagonal columns
  h.length = m + h.gaps; h.nident = m - h.mismatch;
  return h;
}

}  // namespace pga
