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
constexpr int FRAG_XDROP_PRELIM = 33;          // blastn's preliminary gapped extension: 30 bits in raw 2 / -3 scores (frag_hsp: chance hits only)
constexpr int32_t FRAG_NEG = -(1 << 28);

// one cell: u = state of diagonal K+1 (cell (i-1, j)), l = diagonal K-1 (cell (i, j-1)), g = this diagonal's previous cell (i-1, j-1)
struct FragCellIn { int32_t h, x, y, hs, xs, ys; };      // stats packed: mismatches << 16 | gap bases
PG_HD FragCellIn frag_cell(bool has_u, const FragCellIn& u, bool has_l, const FragCellIn& l, bool has_g, const FragCellIn& g, bool ok,
                           int32_t xbest, int32_t abs_floor, int32_t xdrop) {
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
  const int32_t rel_ = xbest - xdrop;                                 // xdrop: FRAG_XDROP, or FRAG_XDROP_PRELIM for the preliminary look at a chance hit
  const int32_t floor_ = rel_ > abs_floor ? rel_ : abs_floor;      // abs_floor: frag_hsp's second look (FRAG_NEG: none)
  c.h = c.h < floor_ ? FRAG_NEG : c.h;
  c.x = c.x < floor_ ? FRAG_NEG : c.x;
  c.y = c.y < floor_ ? FRAG_NEG : c.y;
  return c;
}

template <typename QB, typename SB>
PG_HD FragExt frag_extend(QB&& qbase, int32_t qmax, SB&& sbase, int32_t smax, int32_t abs_floor = FRAG_NEG, int32_t xdrop = FRAG_XDROP) {
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
                                     i >= 1 && j >= 1, S[k], ok, xbest, abs_floor, xdrop);
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

// ---- where blastn starts its gapped alignment (round 6) ---------------------------------------------------------------------------
// Rounds 2-5 grew an HSP from the LONGEST exact seed of the fragment's best locus.  The DP itself already followed blastn's rules
// (fed with blastn's start points it gives blastn's rows: 2 050 of 2 050 on NC_002696 vs NC_010338, profiles/
// r06_anib_product_vs_blastn_restatement.json), but where two alignments of equal score exist the left (reversed) and the right DP pick
// different ones, so WHERE the alignment is split decides one row in twenty-five.  blastn splits it like this (blast_gapalign.c,
// na_ungapped.c; the tests compare with an independent restatement of blastn that shares nothing with this file):
//   * every exact word of BL_WORD bases is a hit; on a diagonal the hits are taken left to right, one that starts inside the stretch
//     already explored is dropped, the others are extended without gaps (reward 2 / penalty -3, stop BL_X_UNGAPPED below the best:
//     20 bits) and become INITIAL HSPs when they score BL_TRIGGER (27 bits) or more;
//   * the best initial HSP (score, then subject start, length, query start) is aligned first, from its word's first base moved to the
//     next 4-base boundary of the subject record (1..4 bases);
//   * before the final alignment that point is kept if it lies in a run of more than BL_START_RUN identities, else moved to the middle
//     of the first such run on its diagonal, else to the middle of the longest run there.
// The product takes these steps on the diagonals of its own seeds (16-mers, or 11-mers where the word tier ran): one diagonal walk
// per seed diagonal — no extra DP pass (blastn bounds the last step by its preliminary X = 30-bit alignment; here the fragment's
// whole diagonal is searched, which moves 4 of 2 059 reported rows on the table above).
constexpr int BL_WORD = 11, BL_X_UNGAPPED = 22, BL_TRIGGER = 28, BL_START_RUN = 20;
constexpr int BL_LOCAL_FULL = 60;                          // below: a chance hit, its own stretch is its diagonal's initial HSP (frag_diag_best_init)
constexpr int BL_WEAK_SCORE = 64, BL_STRONG_SCORE = 128;   // a candidate below 64 is not aligned when the fragment has one of 128 or more

struct FragInit { int32_t score, q_start, len, q_off; };   // best initial HSP of ONE diagonal (s = q + diag): ungapped stretch + its word

// a precedes b in blastn's order of initial HSPs
PG_HD bool frag_init_before(const FragInit& a, int64_t adiag, const FragInit& b, int64_t bdiag) {
  if (a.score != b.score) return a.score > b.score;
  const int64_t as = a.q_start + adiag, bs = b.q_start + bdiag;
  if (as != bs) return as < bs;
  if (a.len != b.len) return a.len > b.len;
  return a.q_start < b.q_start;
}

// The one-hit diagonal procedure on one diagonal of a fragment strand.  match(p): query base p == subject base p + diag, both clean
// and inside their sequences (false outside).  Returns the diagonal's best initial HSP (score 0: none).
// (ungapped X-drop extension of the word at `a`: left of it, then right from its first base)
template <typename M>
PG_HD FragInit frag_ungapped(M&& match, int32_t qlen, int32_t a) {
  int32_t score = 0, sum = 0, q_beg = a, q_end = a;
  for (int32_t t = a - 1; t >= 0; --t) {
    sum += match(t) ? FRAG_MATCH : FRAG_MISMATCH;
    if (sum > 0) { q_beg = t; score += sum; sum = 0; }
    else if (sum < -BL_X_UNGAPPED) break;
  }
  sum = 0;
  for (int32_t t = a; t < qlen; ++t) {
    sum += match(t) ? FRAG_MATCH : FRAG_MISMATCH;
    if (sum > 0) { q_end = t + 1; score += sum; sum = 0; }
    else if (sum < -BL_X_UNGAPPED) break;
  }
  return FragInit{score, q_beg, q_end - q_beg, a};
}

// the whole diagonal, left to right: every exact run of BL_WORD or more that starts outside the stretch already explored is a hit
// (second: the diagonal's second initial HSP in blastn's order, score 0 if none — the preliminary stage looks at both)
template <typename M>
PG_HD FragInit frag_diag_walk(M&& match, int32_t qlen, int64_t diag, FragInit* second = nullptr) {
  FragInit best{0, 0, 0, 0}, next{0, 0, 0, 0};
  int32_t last_hit = 0, p = 0;
  while (p < qlen) {
    if (!match(p)) { ++p; continue; }
    const int32_t a = p;
    while (p < qlen && match(p)) ++p;                 // the maximal exact run [a, p)
    if (p - a < BL_WORD || a < last_hit) continue;
    const FragInit h = frag_ungapped(match, qlen, a);
    const int32_t q_end = h.q_start + h.len;
    last_hit = q_end > a + BL_WORD ? q_end : a + BL_WORD;
    if (h.score < BL_TRIGGER) continue;
    if (best.score == 0 || frag_init_before(h, diag, best, diag)) { next = best; best = h; }
    else if (next.score == 0 || frag_init_before(h, diag, next, diag)) next = h;
  }
  if (second) *second = next;
  return best;
}

// seed_q: where the diagonal's longest seed starts in the fragment strand.  A seed whose own ungapped stretch scores less than
// BL_LOCAL_FULL is a chance hit (2.4 of them per fragment and 5 Mb of unrelated subject): its diagonal holds nothing else (another
// exact 11-mer on the same diagonal of 1020 random bases: p ~ 2e-4), so that stretch IS the diagonal's initial HSP and the walk over
// the whole diagonal is skipped — on the GPU the walk of every chance hit cost as much as the rest of the fragment's work.
template <typename M>
PG_HD FragInit frag_diag_best_init(M&& match, int32_t qlen, int64_t diag, int32_t seed_q, FragInit* second = nullptr) {
  if (second) *second = FragInit{0, 0, 0, 0};
  {
    const FragInit local = frag_ungapped(match, qlen, seed_q);
    if (local.score < BL_LOCAL_FULL) {
      if (local.score < BL_TRIGGER) return FragInit{0, 0, 0, 0};
      int32_t word = seed_q, run = 0;                   // blastn's word: the first exact run of BL_WORD inside the stretch
      for (int32_t t = local.q_start; t < local.q_start + local.len; ++t) {
        run = match(t) ? run + 1 : 0;
        if (run >= BL_WORD) { word = t - run + 1; break; }
      }
      return FragInit{local.score, local.q_start, local.len, word};
    }
  }
  return frag_diag_walk(match, qlen, diag, second);
}

// Up to two candidates among the per-diagonal initial HSPs init[0..n) (score 0 = none) on diagonals diag[]: the best-supported locus'
// first initial HSP in blastn's order, then the same among the diagonals at least FRAG_VOTE_FAR away from it.  Returns their number.
// locus_n[c] (optional): how many initial HSPs the candidate's locus holds — 1 = a LONE hit (what a chance match looks like; a weak
// true alignment has neighbours on nearby diagonals), which is what frag_hsp's preliminary look is for.
// max_cand > 2 (FRAG_MAX_CAND): a repeat family (rRNA operons, insertion elements: a dozen copies with initial HSPs of a few hundred
// each) — blastn aligns them all and the table's first row is the best FINAL score, which the two best-supported loci need not hold
// (NC_002696 vs NC_010338, fragment 3015).  Candidates beyond the second must be strong (BL_STRONG_SCORE): chance hits never open
// a third round.  The caller grows the preliminary alignments of all of them and the final ones of the FRAG_MAX_FINALS best
// (frag_prelim_before; a preliminary score orders copies of a repeat only roughly) and keeps the fragment's FRAG_KEEP_ROWS best rows.
constexpr int FRAG_MAX_CAND = 8, FRAG_MAX_FINALS = 4, FRAG_KEEP_ROWS = 4;      // final alignments per strand; rows kept per fragment (the best)
PG_HD int frag_pick_inits(const FragInit* init, const int64_t* diag, int n, int* cand, int* locus_n = nullptr, int max_cand = 2) {
  int nc = 0;
  for (int round = 0; round < max_cand; ++round) {
    // the locus: the diagonal neighbourhood (FRAG_VOTE_WIN) holding the largest total of initial-HSP scores — blastn aligns every
    // initial HSP and the table's first row is the best FINAL score, which a long alignment in many pieces wins over one strong repeat
    int locus = -1;
    int64_t locus_sum = -1;
    for (int a = 0; a < n; ++a) {
      if (init[a].score <= 0) continue;
      bool taken = false;
      for (int p = 0; p < round; ++p) { const int64_t dd = diag[a] - diag[cand[p]]; taken = taken || (dd < FRAG_VOTE_FAR && -dd < FRAG_VOTE_FAR); }
      if (taken) continue;
      int64_t sum = 0;
      for (int b = 0; b < n; ++b) {
        const int64_t dd = diag[b] - diag[a];
        if (init[b].score > 0 && dd <= FRAG_VOTE_WIN && -dd <= FRAG_VOTE_WIN) sum += init[b].score;
      }
      if (locus < 0 || sum > locus_sum || (sum == locus_sum && frag_init_before(init[a], diag[a], init[locus], diag[locus]))) { locus = a; locus_sum = sum; }
    }
    if (locus < 0) break;
    // inside it, blastn's first initial HSP
    int best = locus, members = 0;
    for (int b = 0; b < n; ++b) {
      const int64_t dd = diag[b] - diag[locus];
      if (init[b].score <= 0 || dd > FRAG_VOTE_WIN || -dd > FRAG_VOTE_WIN) continue;
      bool taken = false;
      for (int p = 0; p < round; ++p) { const int64_t d0 = diag[b] - diag[cand[p]]; taken = taken || (d0 < FRAG_VOTE_FAR && -d0 < FRAG_VOTE_FAR); }
      if (taken) continue;
      ++members;
      if (frag_init_before(init[b], diag[b], init[best], diag[best])) best = b;
    }
    if (round >= 2 && init[best].score < BL_STRONG_SCORE) break;
    if (locus_n) locus_n[nc] = members;
    cand[nc++] = best;
  }
  return nc;
}
PG_HD bool frag_keep_init(int32_t score, int32_t best_score_of_fragment) {
  return !(score < BL_WEAK_SCORE && best_score_of_fragment >= BL_STRONG_SCORE);
}
// A candidate that gets frag_hsp's preliminary look: a chance-sized initial HSP that stands alone in its locus.
PG_HD bool frag_init_is_lone_weak(int32_t score, int locus_members) { return score < BL_LOCAL_FULL && locus_members <= 1; }

// The point the gapped alignment grows from: the word's first base moved to the next 4-base boundary of the subject record, then
// blastn's start rule on that diagonal of the fragment.  q_off: the word in the fragment strand; s_rel = its subject position
// relative to the subject record's first base; lo / hi: the range of fragment positions whose subject base is inside the record.
template <typename M>
PG_HD int32_t frag_start_point(M&& match, int32_t qlen, int32_t q_off, int64_t s_rel, int32_t lo, int32_t hi, int32_t init_score) {
  int32_t g = q_off + 4 - (int32_t)(s_rel & 3);
  if (g >= qlen) g = q_off;
  if (init_score < BL_LOCAL_FULL) return g;          // a chance hit: no run of BL_START_RUN anywhere near, the point stays in its word
  int32_t score = -1;
  for (int32_t t = g; t < hi && match(t); ++t) if (++score > BL_START_RUN) return g;
  for (int32_t t = g; t >= lo && match(t); --t) if (++score > BL_START_RUN) return g;
  int32_t max_score = 0, max_offset = lo, run = 0;
  bool prev = false, m = false;
  int32_t i = lo;
  for (; i < hi; ++i) {
    m = match(i);
    if (m != prev) {
      prev = m;
      if (m) run = 1;
      else if (run > max_score) { max_score = run; max_offset = i - run / 2; }
    } else if (m) {
      if (++run > BL_START_RUN) return i - BL_START_RUN / 2;
    }
  }
  if (m && run > max_score) { max_score = run; max_offset = i - run / 2; }
  return max_score > 0 ? max_offset : g;
}

// The same rule as blastn applies it (BlastGetStartForGappedAlignmentNucl): the search for a run is bounded by the PRELIMINARY
// alignment's box — [q0, q1) x [s0, s1) in fragment / record-relative subject coordinates, grown from the point g (subject gs) under
// the preliminary X-drop.  rec_lo: the first fragment position whose subject base on this diagonal is inside the record.
template <typename M>
PG_HD int32_t frag_start_point_boxed(M&& match, int32_t g, int64_t gs, int32_t q0, int32_t q1, int64_t s0, int64_t s1, int32_t rec_lo) {
  int32_t score = -1;
  for (int32_t t = g; t < q1 && match(t); ++t) if (++score > BL_START_RUN) return g;
  for (int32_t t = g; t >= rec_lo && match(t); --t) if (++score > BL_START_RUN) return g;
  const int64_t off64 = (gs - s0) < (int64_t)(g - q0) ? (gs - s0) : (int64_t)(g - q0);
  const int32_t off = (int32_t)off64, lo = g - off;
  const int64_t s_start = gs - off;
  const int64_t len64 = (s1 - s_start) < (int64_t)(q1 - lo) ? (s1 - s_start) : (int64_t)(q1 - lo);
  const int32_t hi = lo + (int32_t)len64;
  int32_t max_score = 0, max_offset = lo, run = 0;
  bool prev = false, m = false;
  int32_t i = lo;
  for (; i < hi; ++i) {
    m = match(i);
    if (m != prev) {
      prev = m;
      if (m) run = 1;
      else if (run > max_score) { max_score = run; max_offset = i - run / 2; }
    } else if (m) {
      if (++run > BL_START_RUN) return i - BL_START_RUN / 2;
    }
  }
  if (m && run > max_score) { max_score = run; max_offset = i - run / 2; }
  return max_score > 0 ? max_offset : g;
}

// ---- blastn's preliminary stage (round 6, second half) ----------------------------------------------------------------------------
// blastn does not align its best initial HSP once: EVERY initial HSP, best first, is grown under the preliminary X-drop (30 bits)
// unless it lies inside the box of a better preliminary alignment on a nearby diagonal (blast_itree.c, s_HSPIsContained); of
// preliminary alignments with a common end the better one stays, and the final alignment is grown from the start point OF THE BEST
// PRELIMINARY ALIGNMENT — which need not be the best initial HSP's: a strong stretch walled in by a divergent patch loses to a weaker
// one whose 30-bit alignment gets across (NC_002696 vs NC_010338, fragment 1618: initial HSPs of 787 and 437; the first one's
// preliminary alignment stops at 462 of 1020 bases with 787, the second covers the fragment with 1 309 and gives the table's row).
// The product takes these steps over the initial HSPs of a candidate's neighbourhood (FRAG_VOTE_FAR diagonals), at most
// BL_MAX_PRELIMS preliminary alignments per candidate.
struct FragPrelim { int32_t score, q0, q1, g; int64_t s0, s1, diag; };   // box [q0, q1] x [s0, s1], grown from (g, g + diag)
constexpr int BL_MIN_DIAG_SEPARATION = 50, BL_MAX_PRELIMS = 4;
PG_HD bool frag_init_contained(const FragInit& in, int64_t diag, const FragPrelim& t) {
  if (in.score > t.score) return false;
  const int64_t q0 = in.q_start, q1 = q0 + in.len, s0 = q0 + diag, s1 = s0 + in.len;
  const bool inside = q0 >= t.q0 && q0 <= t.q1 && s0 >= t.s0 && s0 <= t.s1 && q1 >= t.q0 && q1 <= t.q1 && s1 >= t.s0 && s1 <= t.s1;
  if (!inside) return false;
  int64_t a = (t.q0 - t.s0) - (q0 - s0), b = (t.q1 - t.s1) - (q1 - s1);
  a = a < 0 ? -a : a; b = b < 0 ? -b : b;
  return a < BL_MIN_DIAG_SEPARATION || b < BL_MIN_DIAG_SEPARATION;
}
PG_HD bool frag_prelim_before(const FragPrelim& a, const FragPrelim& b) {
  if (a.score != b.score) return a.score > b.score;
  if (a.s0 != b.s0) return a.s0 < b.s0;
  if (a.s1 != b.s1) return a.s1 > b.s1;
  if (a.q0 != b.q0) return a.q0 < b.q0;
  return a.q1 > b.q1;
}
// the word's first base moved to the next 4-base boundary of the subject record
PG_HD int32_t frag_word_start(int32_t q_off, int64_t s_rel, int32_t qlen) {
  const int32_t g = q_off + 4 - (int32_t)(s_rel & 3);
  return g >= qlen ? q_off : g;
}

// Two table rows of one fragment that begin or end at the same point of the same strand: blastn keeps the better one whole and cuts
// the other where the better one ends (Blast_HSPListPurgeHSPsWithCommonEndpoints) — typically a weak neighbour whose X-drop
// extension ran across a gap into the main alignment.  What is left of it never covers 70 % of a fragment, so for parse_blast_tab
// dropping it is the same as cutting it: the product drops it (the statistics of a path are carried forward, there is no script to cut).
PG_HD bool frag_rows_share_end(int32_t aqs, int32_t aqe, int32_t ass, int32_t ase, int32_t bqs, int32_t bqe, int32_t bss, int32_t bse) {
  const bool a_minus = ass > ase, b_minus = bss > bse;
  if (a_minus != b_minus) return false;
  return (aqs == bqs && ass == bss) || (aqe == bqe && ase == bse);
}

// blastn's e-value cut (blast_stat.c): E = searchsp * K * exp(-lambda * S) on the EFFECTIVE search space — query and database
// lengths shortened by the length adjustment l, the fixed point of  l = alpha / lambda * ln(K (m - l)(n - N l)) + beta  (gapped
// parameters of 2 / -3 / 5 / 2: lambda 0.625, K 0.41, alpha 0.8, beta -2), database = the whole subject genome (n bases, N records).
PG_HD int32_t frag_length_adjustment(int32_t qlen, int64_t db_len, int32_t db_seqs) {
  const double K = 0.41, logK = -0.8915981192837836, a_d_l = 0.8 / 0.625, beta = -2.0;
  const double m = (double)qlen, n = (double)db_len, N = (double)db_seqs;
  double ell = 0, ell_min = 0, ell_max, ell_next = 0;
  bool converged = false;
  {
    const double a = N, mb = m * N + n, c = n * m - (m > n ? m : n) / K;
    if (c < 0) return 0;
    ell_max = 2 * c / (mb + sqrt(mb * mb - 4 * a * c));
  }
  for (int i = 1; i <= 20; ++i) {
    ell = ell_next;
    const double ss = (m - ell) * (n - N * ell);
    const double ell_bar = a_d_l * (logK + log(ss)) + beta;
    if (ell_bar >= ell) {
      ell_min = ell;
      if (ell_bar - ell_min <= 1.0) { converged = true; break; }
      if (ell_min == ell_max) break;
    } else {
      ell_max = ell;
    }
    if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
    else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) / 2;
  }
  int32_t adj = (int32_t)ell_min;
  if (converged) {
    ell = ceil(ell_min);
    if (ell <= ell_max) {
      const double ss = (m - ell) * (n - N * ell);
      if (a_d_l * (logK + log(ss)) + beta >= ell) adj = (int32_t)ell;
    }
  }
  return adj;
}
PG_HD bool frag_evalue_ok_db(int32_t score, int32_t qlen, int64_t db_len, int32_t db_seqs) {
  const int32_t adj = frag_length_adjustment(qlen, db_len, db_seqs);
  int64_t eff_db = db_len - (int64_t)db_seqs * adj;
  if (eff_db <= 0) eff_db = 1;
  int32_t eff_q = qlen - adj;
  if (eff_q <= 0) eff_q = 1;
  const double bits_nat = 0.625 * (double)score - (-0.8915981192837836);   // lambda * S - ln K
  return (double)eff_db * (double)eff_q * exp(-bits_nat) <= 1e-15;
}

// The word tier (blastn's word size, `-task blastn`: anib.py:465-471): a fragment that the 16-mer seeds leave without a reportable
// HSP is searched again with every 11-mer of either strand; a hit becomes a seed if at least WORD_FLANK_MIN of the WORD_FLANK
// bases on its left OR on its right match on its diagonal (chance: 8 +- 2.4 of 32).  The bar is a PRE-filter only — a fragment strand
// holds ~1 000 chance 11-mer hits and the list WORD_MAX_SEEDS — since blastn's own stages (initial HSP at the gap trigger, preliminary
// and final gapped alignment, e-value) decide what is reported.  18 = 4 sigma above chance (~0.4 chance survivors per fragment) and
// below what a 2/3-identity alignment shows (21.3: coding regions that differ at every third base have FEW exact 11-mers, each with
// ~21 of 32 flank matches — rounds 3-5 asked for 22 and lost exactly those fragments: 3 ... 5 per 2 600 on the 78 - 84 % pairs of the
// reference's tests, +0.02 pp mean identity; 16 admits enough chance seeds to crowd the list again).
constexpr int WORD_K = 11, WORD_FLANK = 32, WORD_FLANK_MIN = 18, WORD_MAX_SEEDS = 512;
// In the fragment's seed list (FRAG_MAX_SEEDS entries) the word tier's seeds come FIRST, up to FRAG_WORD_FIRST of them (longest first):
// each passed the flank test (chance: ~1e-7 per hit), while a fragment with a low-complexity stretch can hold more than FRAG_MAX_SEEDS
// chance 16-mers, all longer than the 11 ... 15-base words of a real 72 % alignment (NC_002696 vs NC_010338, fragments 1573, 2911, 2930:
// found by the tier and cut from the list again by the length ranking of rounds 3-5).
constexpr int FRAG_WORD_FIRST = 32;

// BLAST's e-value for raw score S (blastn 2 / -3, gap costs 5 / 2: lambda = 0.625, K = 0.41), search space m * n without length
// adjustment; pyani runs blastn with -evalue 1e-15 (anib.py:466).
PG_HD bool frag_evalue_ok(int32_t score, int32_t qlen, int64_t slen) {
  const double bits_nat = 0.625 * (double)score - (-0.8915981192837836);   // lambda * S - ln K
  return (double)qlen * (double)slen * exp(-bits_nat) <= 1e-15;
}

// blastn reports an alignment whole only if no proper prefix and no proper suffix of it scores more than all of it: otherwise the
// better-scoring part is an HSP of its own (found from its own words) and this one, sharing an end with it, is cut there
// (Blast_HSPListPurgeHSPsWithCommonEndpoints + re-evaluation).  With the alignment grown from a point, a prefix can only beat the
// whole if the LEFT extension's running score fell below minus the right side's total on its way (and the other way round), which
// needs that total to be below the X-drop: only then is the side grown a second time, with that floor.
PG_HD bool frag_second_look(int32_t other_total, int32_t this_score) { return other_total < FRAG_XDROP && this_score > 0; }

// After the preliminary look at a lone chance-sized hit: the final alignment is made if the preliminary score passes the e-value
// cut-off (blastn's rule) — or if it has grown to BL_PRELIM_RESCUE, which chance does not do (a lone 16-mer's stretch scores 32 - 45):
// such a hit lies in a real, weak alignment, which blastn reaches from another of its 11-mer words that the 16-mer seeding does not
// see (NC_002696 vs NC_010338, fragment 2494: the one seed's preliminary alignment stops at 56, BLAST+ reports 965 columns there).
constexpr int BL_PRELIM_RESCUE = 50;
template <typename OK>
PG_HD bool frag_prelim_goes_on(int32_t prelim_score, OK&& reportable) { return prelim_score >= BL_PRELIM_RESCUE || reportable(prelim_score); }

// The HSP grown from an exact anchor  query [aq, aq + alen)  ==  subject [as, as + alen)  (both within their limits): leftward
// and rightward extension + the anchor itself.  q_at(p) / s_at(p): base at absolute query / subject position p (4 / 5 outside).
// weak: the candidate is a chance hit (its initial HSP scores less than BL_LOCAL_FULL).  blastn aligns every initial HSP with a
// PRELIMINARY X-drop of 30 bits first and drops those whose preliminary score misses the e-value cut-off before the final 150-bit
// alignment is ever made; a chance hit dies there within ~20 bases.  The product takes that look for weak candidates only (a strong
// one passes it anyway): the first version ran every chance hit — 2.4 per fragment and unrelated 5 Mb subject — under the final
// X-drop, ~5 x the cells, which was most of what the fragment kernel did on a C5 grid.
template <typename QA, typename SA, typename OK>
PG_HD FragHit frag_hsp(QA&& q_at, int32_t qlen, SA&& s_at, int64_t s_lo, int64_t s_hi, int32_t aq, int64_t as, int32_t alen, OK&& reportable,
                       bool weak = false) {
  const int64_t room_r = s_hi - (as + alen), room_l = as - s_lo;
  const int32_t cap = FRAG_SIZE + FRAG_SLACK;
  auto qr = [&](int32_t t) { return q_at(aq + alen + t); };
  auto sr = [&](int32_t t) { return s_at(as + alen + t); };
  auto ql = [&](int32_t t) { return q_at(aq - 1 - t); };
  auto sl = [&](int32_t t) { return s_at(as - 1 - t); };
  const int32_t nr = (int32_t)(room_r < cap ? room_r : cap), nl = (int32_t)(room_l < cap ? room_l : cap);
  if (weak) {
    const FragExt Rp = frag_extend(qr, qlen - (aq + alen), sr, nr, FRAG_NEG, FRAG_XDROP_PRELIM);
    const FragExt Lp = frag_extend(ql, aq, sl, nl, FRAG_NEG, FRAG_XDROP_PRELIM);
    if (!frag_prelim_goes_on(Rp.score + Lp.score + FRAG_MATCH * alen, reportable)) return frag_join(Lp, Rp, aq, as, alen);      // (fails the cut-off: the caller drops it)
  }
  FragExt R = frag_extend(qr, qlen - (aq + alen), sr, nr);
  FragExt L = frag_extend(ql, aq, sl, nl);
  // The second look (frag_second_look): where one side scores less than the X-drop, the other side may have crossed a dip deeper than
  // that side is worth — blastn then reports the far part as an HSP of its own and cuts this one at it.  That side is grown again
  // with the running score not allowed below minus the other side's total; without such a dip the result is the same extension.
  const int32_t r_total = R.score + FRAG_MATCH * alen, l_total = L.score;
  if (reportable(r_total + l_total)) {      // (a second look can only lower the score: a row that fails the e-value already is not looked at again)
    if (frag_second_look(r_total, L.score)) L = frag_extend(ql, aq, sl, nl, -r_total);
    if (frag_second_look(l_total, R.score)) R = frag_extend(qr, qlen - (aq + alen), sr, nr, -l_total);
  }
  return frag_join(L, R, aq, as, alen);
}

}  // namespace pga
