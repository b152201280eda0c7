// pg_anib_core.h — fragment mode of the aligner (BASELINE.json configs[4]): the search pyani's ANIb
// delegates to BLAST+,   blastn -task blastn -query <1020-nt fragments of genome Q> -db <genome S> -xdrop_gap_final 150
// -dust no -evalue 1e-15 -max_target_seqs 1   (pyani/anib.py:451-471), as plain C++ that compiles for the device (hipcc)
// AND for the host (the development harness and the CPU checker build): every fragment gets its best local alignment against the subject genome,
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
#include <cmath>
#include "pg_anim_core.h"

namespace pga {

constexpr int FRAG_SIZE = 1020;                 // pyani_config.FRAGSIZE
constexpr int FRAG_QSTEP = 1;                   // seeds: every 16-mer of the fragmented genome is looked up (ANIm samples every 5th)
constexpr int FRAG_MATCH = 2, FRAG_MISMATCH = -3, FRAG_GAP_OPEN = -7, FRAG_GAP_EXT = -2;   // first gap base -(5 + 2), further -2
#ifndef PGA_FRAG_BAND
#define PGA_FRAG_BAND 256
#endif
constexpr int FRAG_BAND = PGA_FRAG_BAND;                   // diagonals of the DP band (4 per lane on the device): a 5 + 2k gap of ~80 bases still fits the X-drop
constexpr int FRAG_SLACK = 200;                // subject bases an extension may use beyond the fragment's own length

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
// qmax / smax bases.  Anti-diagonal order (d = i + j); the band covers the FRAG_BAND diagonals K = j - i in [koff, koff + FRAG_BAND),
// starts centred on the anchor's diagonal and FOLLOWS the alignment: before every FRAG_TRACK-th anti-diagonal it is re-centred on
// the diagonal of the best live H (ties: lowest diagonal), by an even number of diagonals (the cell / anti-diagonal parity pattern
// is kept), at most FRAG_SHIFT_MAX; states that leave the band are lost, new ones start dead.  A state more than FRAG_XDROP below
// `xbest` is dead (BLAST's -xdrop_gap_final 150 bits = 166 in raw 2 / -3 scores), xbest being the best score as it stood before the
// last anti-diagonal that is a multiple of FRAG_XSYNC (a wave refreshes its shared copy that often); the search ends after two
// dead anti-diagonals.  The end is the best cell: highest score, ties -> earliest anti-diagonal, then lowest diagonal.
// The device version (pga_frag.inc: the band in LDS, 4 diagonals per lane) walks the same cells and applies the same rules.
#ifndef PGA_FRAG_XDROP
#define PGA_FRAG_XDROP 166
#endif
constexpr int FRAG_TRACK = 16, FRAG_SHIFT_MAX = 8, FRAG_XDROP = PGA_FRAG_XDROP, FRAG_XSYNC = 4;
constexpr int32_t FRAG_NEG = -(1 << 28);

// one cell: u = state of diagonal K+1 (cell (i-1, j)), l = diagonal K-1 (cell (i, j-1)), g = this diagonal's previous cell (i-1, j-1)
struct FragCellIn { int32_t h, x, y, hs, xs, ys; };      // stats packed: mismatches << 16 | gap bases
PG_HD FragCellIn frag_cell(bool has_u, const FragCellIn& u, bool has_l, const FragCellIn& l, bool has_g, const FragCellIn& g, bool ok,
                           int32_t xbest) {
  // straight-line: selects and NON-short-circuit logic only (& and | on the flags) — on the device every branch here cost a
  // round of exec-mask bookkeeping per cell
  constexpr int32_t LIVE = FRAG_NEG / 2;
  FragCellIn c;
  {                                                      // X: gap consuming a query base (ties: extend the open gap)
    const int32_t ho = u.h + FRAG_GAP_OPEN, xo = u.x + FRAG_GAP_EXT;
    const bool ext = (u.x > LIVE) & (xo >= ho), any = has_u & (ext | (u.h > LIVE));
    c.x = any ? (ext ? xo : ho) : FRAG_NEG;
    c.xs = any ? (ext ? u.xs : u.hs) + 1 : 0;
  }
  {                                                      // Y: gap consuming a subject base
    const int32_t ho = l.h + FRAG_GAP_OPEN, yo = l.y + FRAG_GAP_EXT;
    const bool ext = (l.y > LIVE) & (yo >= ho), any = has_l & (ext | (l.h > LIVE));
    c.y = any ? (ext ? yo : ho) : FRAG_NEG;
    c.ys = any ? (ext ? l.ys : l.hs) + 1 : 0;
  }
  const bool diag = has_g & (g.h > LIVE);
  c.h = diag ? g.h + (ok ? FRAG_MATCH : FRAG_MISMATCH) : FRAG_NEG;
  c.hs = diag ? g.hs + (ok ? 0 : 65536) : 0;
  const bool tx = c.x > c.h;
  c.h = tx ? c.x : c.h; c.hs = tx ? c.xs : c.hs;
  const bool ty = c.y > c.h;
  c.h = ty ? c.y : c.h; c.hs = ty ? c.ys : c.hs;
  const int32_t floor_ = xbest - FRAG_XDROP;
  c.h = c.h < floor_ ? FRAG_NEG : c.h;
  c.x = c.x < floor_ ? FRAG_NEG : c.x;
  c.y = c.y < floor_ ? FRAG_NEG : c.y;
  return c;
}

template <typename QB, typename SB>
PG_HD FragExt frag_extend(QB&& qbase, int32_t qmax, SB&& sbase, int32_t smax) {
  if (qmax <= 0 || smax <= 0) return FragExt{0, 0, 0, 0, 0};   // nothing to extend into (the anchor reaches the fragment's end)
  FragCellIn S[FRAG_BAND];                  // latest cell of every diagonal (index k <-> diagonal K = k + koff)
  const FragCellIn dead{FRAG_NEG, FRAG_NEG, FRAG_NEG, 0, 0, 0};
  for (int k = 0; k < FRAG_BAND; ++k) S[k] = dead;
  int32_t koff = -FRAG_BAND / 2;
  S[0 - koff].h = 0;                        // the anchor cell (0, 0) on diagonal 0
  FragExt best{0, 0, 0, 0, 0};
  int32_t xbest = 0;
  int dead_run = 0;
  for (int32_t d = 1; d <= qmax + smax; ++d) {
    if ((d % FRAG_TRACK) == 0) {            // re-centre the band on the best live H
      int32_t bh = FRAG_NEG / 2, bk = -1;
      for (int k = 0; k < FRAG_BAND; ++k) if (S[k].h > bh) { bh = S[k].h; bk = k; }
      if (bk >= 0) {
        int32_t s = bk - FRAG_BAND / 2;
        if (s > FRAG_SHIFT_MAX) s = FRAG_SHIFT_MAX;
        if (s < -FRAG_SHIFT_MAX) s = -FRAG_SHIFT_MAX;
        s &= ~1;
        if (s > 0) { for (int k = 0; k < FRAG_BAND; ++k) S[k] = k + s < FRAG_BAND ? S[k + s] : dead; }
        if (s < 0) { for (int k = FRAG_BAND - 1; k >= 0; --k) S[k] = k + s >= 0 ? S[k + s] : dead; }
        koff += s;
      }
    }
    if ((d % FRAG_XSYNC) == 0) xbest = best.score;
    bool alive = false;
    // cells of one anti-diagonal read only diagonals of the other parity (last written one step ago) and their own previous
    // cell (two steps ago), so updating in place is the same as updating from a copy
    for (int k = 0; k < FRAG_BAND; ++k) {
      const int32_t K = k + koff;
      if ((d + K) & 1) continue;
      const int32_t i = (d - K) / 2, j = (d + K) / 2;
      if (i < 0 || j < 0 || i > qmax || j > smax) { S[k] = dead; continue; }
      bool ok = false;
      if (i >= 1 && j >= 1) { const int qb = qbase(i - 1), sb = sbase(j - 1); ok = qb < 4 && qb == sb; }
      const FragCellIn c = frag_cell(i >= 1 && k + 1 < FRAG_BAND, S[k + 1 < FRAG_BAND ? k + 1 : k], j >= 1 && k >= 1, S[k >= 1 ? k - 1 : k],
                                     i >= 1 && j >= 1, S[k], ok, xbest);
      S[k] = c;
      if (c.h > FRAG_NEG / 2 || c.x > FRAG_NEG / 2 || c.y > FRAG_NEG / 2) alive = true;
      if (c.h > best.score) { best.score = c.h; best.di = i; best.dj = j; best.mm = c.hs >> 16; best.gaps = c.hs & 0xFFFF; }
    }
    dead_run = alive ? 0 : dead_run + 1;
    if (dead_run >= 2) break;
  }
  return best;
}

// left extension + anchor + right extension -> the HSP's table row
PG_HD FragHit frag_join(const FragExt& L, const FragExt& R, int32_t aq, int64_t as, int32_t alen) {
  FragHit h;
  h.score = FRAG_MATCH * alen + L.score + R.score;
  h.qs = aq - L.di; h.qe = aq + alen + R.di;
  h.ss = (int32_t)(as - L.dj); h.se = (int32_t)(as + alen + R.dj);
  h.gaps = L.gaps + R.gaps; h.mismatch = L.mm + R.mm;
  const int32_t m = ((h.qe - h.qs) + (h.se - h.ss) - h.gaps) / 2;       // diagonal columns
  h.length = m + h.gaps; h.nident = m - h.mismatch;
  return h;
}

// ---- anchors ---------------------------------------------------------------------------------------------------------
// The exact matches of one (fragment, strand), clipped to the fragment: s = subject stream position, q = position in the fragment
// (on the searched strand), len.  Candidate anchors: every match scores the total length of the matches within FRAG_VOTE_WIN
// diagonals of its own; the best-scoring match (ties: longer, then smaller q, then smaller s) marks the locus, and the ANCHOR is
// the longest match within FRAG_VOTE_WIN diagonals of it (same ties) — in a tandem repeat the short off-diagonal copies can
// out-vote the long true match, but they cannot out-grow it.  The second candidate is found the same way among the matches at
// least FRAG_VOTE_FAR diagonals away from the first anchor.  Returns the number of candidates (0..2), their indices in cand[].
struct FragSeed { int32_t s, q, len; };
constexpr int FRAG_VOTE_WIN = 16, FRAG_VOTE_FAR = 48, FRAG_MAX_SEEDS = 64, FRAG_MIN_CLIP = 11;
PG_HD bool frag_seed_before(const FragSeed& a, const FragSeed& o) {      // a is preferred to o at equal votes / as the longer anchor
  return a.len > o.len || (a.len == o.len && (a.q < o.q || (a.q == o.q && a.s < o.s)));
}
// votes[] receives each candidate's locus score.  frag_keep_candidate: a candidate whose locus holds fewer than 32 matched bases
// (one chance 16-mer) is not extended when another candidate of the fragment (either strand) holds at least 64.
constexpr int FRAG_WEAK_VOTES = 32, FRAG_STRONG_VOTES = 64;
PG_HD bool frag_keep_candidate(int32_t votes, int32_t best_votes_of_fragment) {
  return !(votes < FRAG_WEAK_VOTES && best_votes_of_fragment >= FRAG_STRONG_VOTES);
}
PG_HD int frag_pick_anchors(const FragSeed* e, int n, int* cand, int32_t* votes_out) {
  int nc = 0;
  int64_t first_diag = 0;
  for (int round = 0; round < 2; ++round) {
    int best = -1;
    int64_t best_votes = -1;
    for (int a = 0; a < n; ++a) {
      const int64_t da = (int64_t)e[a].s - e[a].q;
      if (round == 1 && (da - first_diag < FRAG_VOTE_FAR && first_diag - da < FRAG_VOTE_FAR)) continue;
      int64_t votes = 0;
      for (int b = 0; b < n; ++b) {
        const int64_t db = (int64_t)e[b].s - e[b].q;
        if (db - da <= FRAG_VOTE_WIN && da - db <= FRAG_VOTE_WIN) votes += e[b].len;
      }
      if (best < 0 || votes > best_votes || (votes == best_votes && frag_seed_before(e[a], e[best]))) { best = a; best_votes = votes; }
    }
    if (best < 0) break;
    const int64_t dl = (int64_t)e[best].s - e[best].q;
    int anchor = best;
    for (int b = 0; b < n; ++b) {
      const int64_t db = (int64_t)e[b].s - e[b].q;
      if (db - dl > FRAG_VOTE_WIN || dl - db > FRAG_VOTE_WIN) continue;
      if (round == 1 && (db - first_diag < FRAG_VOTE_FAR && first_diag - db < FRAG_VOTE_FAR)) continue;
      if (frag_seed_before(e[b], e[anchor])) anchor = b;
    }
    votes_out[nc] = (int32_t)(best_votes > 0x7FFFFFFF ? 0x7FFFFFFF : best_votes);
    cand[nc++] = anchor;
    first_diag = (int64_t)e[anchor].s - e[anchor].q;
  }
  return nc;
}

// The word tier (blastn's word size, `-task blastn`: anib.py:465-471): a fragment that the 16-mer seeds leave without a reportable
// HSP is searched again with every 11-mer of either strand; a hit becomes a seed if at least WORD_FLANK_MIN of the WORD_FLANK
// bases on its left OR on its right match on its diagonal (chance: 8 +- 2.4 of 32; 22 centres the agreement with the BLAST+
// tables of the reference's tests: the CPU checker of the tests (anib_cpu.cpp) has the statement, profiles/r03_anib_blast_agreement.json the level).
constexpr int WORD_K = 11, WORD_FLANK = 32, WORD_FLANK_MIN = 22, WORD_MAX_SEEDS = 512;

// BLAST's e-value for raw score S (blastn 2 / -3, gap costs 5 / 2: lambda = 0.625, K = 0.41), search space m * n without length
// adjustment; pyani runs blastn with -evalue 1e-15 (anib.py:466).
PG_HD bool frag_evalue_ok(int32_t score, int32_t qlen, int64_t slen) {
  const double bits_nat = 0.625 * (double)score - (-0.8915981192837836);   // lambda * S - ln K
  return (double)qlen * (double)slen * exp(-bits_nat) <= 1e-15;
}

// The HSP grown from an exact anchor  query [aq, aq + alen)  ==  subject [as, as + alen)  (both within their limits): leftward
// and rightward extension + the anchor itself.  q_at(p) / s_at(p): base at absolute query / subject position p (4 / 5 outside).
template <typename QA, typename SA>
PG_HD FragHit frag_hsp(QA&& q_at, int32_t qlen, SA&& s_at, int64_t s_lo, int64_t s_hi, int32_t aq, int64_t as, int32_t alen) {
  const int64_t room_r = s_hi - (as + alen), room_l = as - s_lo;
  const int32_t cap = FRAG_SIZE + FRAG_SLACK;
  const FragExt R = frag_extend([&](int32_t t) { return q_at(aq + alen + t); }, qlen - (aq + alen),
                                [&](int32_t t) { return s_at(as + alen + t); }, (int32_t)(room_r < cap ? room_r : cap));
  const FragExt L = frag_extend([&](int32_t t) { return q_at(aq - 1 - t); }, aq,
                                [&](int32_t t) { return s_at(as - 1 - t); }, (int32_t)(room_l < cap ? room_l : cap));
  return frag_join(L, R, aq, as, alen);
}

}  // namespace pga
