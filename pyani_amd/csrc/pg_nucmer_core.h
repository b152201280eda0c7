// pg_nucmer_core.h — the extension stage of the ANIm engine as MUMmer 3.23's postnuc defines it: `extendClusters` over the
// clusters of one (reference, query strand) unit, with the alignment engine (`sw_align`'s anti-diagonal DP: dynamic band that
// grows by one cell per side and step, trimmed from its edges at breaklen * 3 below the best score, break after breaklen
// anti-diagonals without a new best, three states per cell with MUMmer's tie order) as a template parameter.
// Plain C++ that compiles for the device (pg_anim.hip: wave-cooperative engine, pga_postnuc.inc) AND for the host
// (ScalarEngine below: the statement the GPU must equal; tools/anim_debug and the CPU checker of the tests).
//
// What it restates, and what pins it: pyani runs `nucmer --mum` per ordered pair (pyani/anim.py:240-289); nucmer's extension
// step is postnuc.  The tests' CPU checker of nucmer is an independent restatement of the same published algorithm with traceback and
// MUMmer's own data structures; it reproduces all 25 192 alignment records (and every indel list) of the MUMmer output files
// the reference's tests hold.  This file differs from it in form, not in results: stream coordinates, fixed arrays, and error
// counts that ride along with the scores (every state carries the errors of its chosen path, choices follow MUMmer's tie order,
// so no traceback is stored) — tests/test_anim_cpu.py holds the two against each other and against the fixtures.
#pragma once
#include <stdint.h>
#include "pg_anim_core.h"
#if !defined(__HIP_DEVICE_COMPILE__)
#include <thread>
#endif

namespace pgn {
using pga::Chain;
using pga::Match;

constexpr int32_t GOOD_SCORE = 3, BAD_SCORE = -7, OPEN_GAP_SCORE = -10, CONT_GAP_SCORE = -7;   // sw_alignscore.hh, nucleotides
constexpr int32_t BREAK_LEN = 200;                       // nucmer -b
constexpr int32_t MAX_DIFF = GOOD_SCORE * BREAK_LEN;     // a band-edge cell further below the best score is trimmed
constexpr int32_t MAX_ALIGNMENT_LENGTH = 10000;
enum : unsigned { DIRECTION_BIT = 1, SEARCH_BIT = 2, FORCED_BIT = 4, OPTIMAL_BIT = 8 };
constexpr unsigned FORWARD_ALIGN = 1, FORCED_FORWARD_ALIGN = 5, BACKWARD_SEARCH = 2;

// An alignment under construction, MUMmer's coordinates transplanted to the packed streams: positions are stream indices,
// BOTH ends inclusive (sA = first aligned reference base, eA = last); B in query-STRAND coordinates.
struct PnAln {
  int32_t sA, sB, eA, eB;
  int32_t errors;
  int32_t chain;   // the chain (cluster) it started from: names the (reference record, query record) it belongs to
};

// The forward alignment from a cluster's match to its next one, computed ahead of the unit's walk (pga_postnuc.inc).
struct PnGap { int32_t eA, eB, errors, reached; };
// The forward extension off a cluster's last match (towards its forward target cluster): a function of the unit's clusters alone —
// getForwardTargetCluster looks at the clusters' matches, not at what has been aligned or fused so far, and the extension starts on
// the last base of the cluster's last match whichever way the walk came to it — so an engine may have all of them ready
// (postnuc_forward: the GPU runs one wave per cluster before the units' sequential walks).
struct PnFwd { int32_t eA, eB, errors, targetk, reached; };
// A backward search run AHEAD of the walk (one per cluster, for the alignment that starts on its first match): its arguments as
// a DP-free rehearsal of the walk predicted them (postnuc_rehearse: the walk with every backward search answered "found
// nothing" — alignment ENDS do not depend on backward searches, only starts and merges do, so the targets come out right almost
// always) and its result.  The real walk takes the result only if its own arguments are these (else it searches itself):
// a wrong prediction costs time, never a result.  state: 0 = no search predicted, 1 = predicted, 2 = result present.
struct PnBwd { int32_t sA, sB, tA, tB; uint32_t m_o; int32_t rA, rB, reached, state; };

// ---- packed DP words ----------------------------------------------------------------------------------------------------
// One 32-bit word per state:  (score + SCORE_BIAS) << 17 | state << 15 | errors.
// MUMmer compares scores only and breaks ties by state (scoreEdit and maxScore alike: MATCH, then INSERT, then DELETE).  With the
// state of ORIGIN in the two bits under the score, one unsigned max over the three candidates is exactly that rule — the three
// candidates of a choice always come from three different states, so the error bits below never decide — and the errors of the
// chosen path ride along for free.  After a choice the word is re-labelled with the state it now belongs to.
// A word whose score field is 0 is UNREACHABLE (any low bits).  Fields: 15 bits of score, bias 2700: an alignment of at most
// 10 001 x 10 001 bases cannot score above 30 003, so the field's 32 768 values cover [-2700, 30 067] (rounds 3-4 had the bias
// at 1024 and wasted 1 700 values at the top).  A trimmed search keeps its live cells within MAX_DIFF + a few gap steps of the
// best score, which starts at 3: it never comes near the floor.  A forced run may push cells below -2700, where they saturate to
// unreachable: harmless off the optimal path (they are thousands of points under it), a deviation from MUMmer's plain integers
// only if the OPTIMAL path of a forced rectangle has a prefix below -2700 (386 mismatches in a row) — and a rectangle whose corner
// can then not be reached at all is reported (forced_verdict below), never counted as 0 errors.  tests/test_anim_cpu.py holds
// both cases.  15 bits of errors: one call aligns at most 20 002 bases.
constexpr uint32_t SCORE_BIAS = 2700u, SCORE_SHIFT = 17u;
constexpr uint32_t W_ONE = 1u << SCORE_SHIFT;                 // one score point; also: the smallest live word
constexpr uint32_t W_STATE = 3u << 15, W_ERR = 0x7FFFu;
enum : uint32_t { ST_DELETE = 0u << 15, ST_INSERT = 1u << 15, ST_MATCH = 2u << 15 };   // DELETE consumes a B base, INSERT an A base
PG_HD uint32_t w_make(int32_t score, uint32_t errors, uint32_t state) { return ((uint32_t)(score + (int32_t)SCORE_BIAS) << SCORE_SHIFT) | state | errors; }
PG_HD int32_t w_score(uint32_t w) { return (int32_t)(w >> SCORE_SHIFT) - (int32_t)SCORE_BIAS; }
PG_HD uint32_t w_errors(uint32_t w) { return w & W_ERR; }
PG_HD uint32_t w_relabel(uint32_t w, uint32_t state) { return (w & ~W_STATE) | state; }
// Saturating unsigned subtraction (one v_sub_u32 with clamp on the device).
PG_HD uint32_t w_sat_sub(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_sub_sat(a, b);
#else
  return a > b ? a - b : 0u;
#endif
}
// w + cost (cost < 0) and one more error.  A word that falls to score field 0 is unreachable whatever its low bits hold (they
// cannot carry into the field: at most 20 002 + 3 * 32768 < 2^17), so the subtraction may simply saturate — no test, no select.
PG_HD uint32_t w_gap(uint32_t w, int32_t cost) { return w_sat_sub(w, ((uint32_t)(-cost) << SCORE_SHIFT) - 1u); }
PG_HD uint32_t w_step(uint32_t w, bool same) {     // diagonal step from the best state of (i-1, j-1)
  const uint32_t hit = w + ((uint32_t)GOOD_SCORE << SCORE_SHIFT), miss = w_sat_sub(w, ((uint32_t)(-BAD_SCORE) << SCORE_SHIFT) - 1u);
  return w >= W_ONE ? (same ? hit : miss) : 0u;    // (an unreachable word must stay one: matches would lift it back into the field)
}
// The NORMALISED frame of the trimmed searches on the wave engines (pg_nucmer_diag.h, diag_lane_step<NORM>): on anti-diagonal d
// the score field holds score - norm_offset(d) + NORM_BIAS.  A path to (i, j) has at most min(i, j) <= floor(d / 2) matches, so
// score - norm_offset(d) <= 0: the bias is the top of the field.  A search of 10 001 x 10 001 bases ends at d = 20 002 with the
// offset at 30 003, and its best score is >= -10 from the first anti-diagonal on: the trimming threshold's field is >= 32 767 -
// 30 003 - 10 - MAX_DIFF = 2 154 — the same distance from the floor the plain frame keeps (SCORE_BIAS - 10 - MAX_DIFF = 2 090).
constexpr uint32_t NORM_BIAS = 32767u;
PG_HD int32_t norm_offset(int32_t d) { return GOOD_SCORE * (d >> 1); }
// the diagonal step in that frame: a match keeps the word (an unreachable word stays unreachable by itself), a mismatch pays
// BAD - GOOD and one error.  miss_window: the slot's window of MISMATCH bits, the cell's own bit on top.
PG_HD uint32_t w_step_norm(uint32_t w, uint32_t miss_window) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (as written for the host the compiler folds shift + and back into v_cmp + v_cndmask: two half-rate instructions; the shift is
  // therefore opaque to it — v_ashrrev_i32, v_and_b32 and v_sub_u32 clamp are the three full-rate ones)
  uint32_t sign;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(sign) : "v"(miss_window));
#else
  const uint32_t sign = (uint32_t)((int32_t)miss_window >> 31);
#endif
  return w_sat_sub(w, sign & (((uint32_t)(GOOD_SCORE - BAD_SCORE) << SCORE_SHIFT) - 1u));
}
// w_step with the match bit as the SIGN of a window word (the wave engines' forced runs, which stay in the plain frame): the sign
// spread over the word selects the price — add GOOD + |BAD| on a match, then pay |BAD| and one error either way (a match never
// saturates: it ends above where it started) — in full-rate instructions; the reach test stays a compare and a select.
PG_HD uint32_t w_step_mask(uint32_t w, uint32_t match_mask) {      // match_mask: all ones on a match, 0 otherwise
  constexpr uint32_t MISS = ((uint32_t)(-BAD_SCORE) << SCORE_SHIFT) - 1u, LIFT = ((uint32_t)GOOD_SCORE << SCORE_SHIFT) + MISS;
  const uint32_t t = w_sat_sub(w + (match_mask & LIFT), MISS);
  return w >= W_ONE ? t : 0u;
}
PG_HD uint32_t w_step_window(uint32_t w, uint32_t match_window) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t sign;
  asm("v_ashrrev_i32 %0, 31, %1" : "=v"(sign) : "v"(match_window));
  return w_step_mask(w, sign);
#else
  return w_step(w, (int32_t)match_window < 0);
#endif
}
// bit `bit` of `word` as a match mask (one v_bfe_i32 on the device)
PG_HD uint32_t w_bit_mask(uint32_t word, int bit) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_sbfe((int)word, (unsigned)bit, 1u);
#else
  return 0u - ((word >> bit) & 1u);
#endif
}
PG_HD uint32_t w_max3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a > b ? a : b; return m > c ? m : c; }

struct Cell { uint32_t D, I, M, X; };   // the three states and X = the best of them (labelled with the winner's state)

// One cell from its three neighbours: L = (i, j-1) and U = (i-1, j) on the previous anti-diagonal, G = best state of (i-1, j-1);
// a neighbour that does not exist (outside its anti-diagonal's computed range, or the matrix) is all zeros.
PG_HD Cell cell_update(const Cell& L, const Cell& U, uint32_t G, bool same) {
  Cell c;
  c.D = w_relabel(w_max3(w_gap(L.D, CONT_GAP_SCORE), w_gap(L.I, OPEN_GAP_SCORE), w_gap(L.M, OPEN_GAP_SCORE)), ST_DELETE);
  c.I = w_relabel(w_max3(w_gap(U.D, OPEN_GAP_SCORE), w_gap(U.I, CONT_GAP_SCORE), w_gap(U.M, OPEN_GAP_SCORE)), ST_INSERT);
  c.M = w_relabel(w_step(G, same), ST_MATCH);
  c.X = w_max3(c.D, c.I, c.M);
  return c;
}

// ---- traceback (the .delta indel lists; optional: pg_anim_alignments_batch) ------------------------------------------------
// The riding error counts make a traceback unnecessary for ANI; MUMmer's .delta files also hold each alignment's path as
// indel offsets.  For those, a run of the scalar engine can store one byte per computed cell — the state of ORIGIN of each of
// the cell's three states (the two state bits of the winning candidate word BEFORE it is re-labelled: that is what the word
// format keeps them for) and the state of the cell's best word — anti-diagonal after anti-diagonal, and walk back from the
// finish cell: MATCH steps to (i - 1, j - 1), INSERT to (i - 1, j), DELETE to (i, j - 1), each into the recorded state of origin.
struct PnTrace {
  uint8_t* bp;         // [bp_cap] one byte per computed cell: origin of D | origin of I << 2 | origin of M << 4 | state of X << 6
  uint64_t bp_cap;
  uint32_t* doff;      // [dcap] offset of anti-diagonal d's first cell in bp (doff[last + 1] = cells used)
  int32_t* dlo;        // [dcap] its lowest column
  int32_t dcap;
  uint64_t used;
  int32_t finish_ct, finish_j, overflow;
};
PG_HD Cell cell_update_bp(const Cell& L, const Cell& U, uint32_t G, bool same, uint8_t& bp) {
  const uint32_t d = w_max3(w_gap(L.D, CONT_GAP_SCORE), w_gap(L.I, OPEN_GAP_SCORE), w_gap(L.M, OPEN_GAP_SCORE));
  const uint32_t i = w_max3(w_gap(U.D, OPEN_GAP_SCORE), w_gap(U.I, CONT_GAP_SCORE), w_gap(U.M, OPEN_GAP_SCORE));
  const uint32_t m = w_step(G, same);
  Cell c;
  c.D = w_relabel(d, ST_DELETE); c.I = w_relabel(i, ST_INSERT); c.M = w_relabel(m, ST_MATCH);
  c.X = w_max3(c.D, c.I, c.M);
  bp = (uint8_t)(((d >> 15) & 3u) | (((i >> 15) & 3u) << 2) | (((m >> 15) & 3u) << 4) | (((c.X >> 15) & 3u) << 6));
  return c;
}
// The path from the finish cell back to the start, run-length coded in REVERSE order: entry = op << 28 | count, op 0 = DELETE
// (a B base alone), 1 = INSERT (an A base alone), 2 = MATCH column.  Returns the number of entries, -1 if `cap` is too small or
// the trace is broken.  The first and last columns are the two corner base pairs (or gap ops standing in for them).
PG_HD int32_t pn_trace_back(const PnTrace& tr, uint32_t* rle, int32_t cap) {
  int32_t d = tr.finish_ct, j = tr.finish_j, n = 0;
  if (d <= 0) return 0;
  if (j < tr.dlo[d] || (uint32_t)(j - tr.dlo[d]) >= tr.doff[d + 1] - tr.doff[d]) return -1;
  uint32_t st = tr.bp[tr.doff[d] + (uint32_t)(j - tr.dlo[d])] >> 6;
  uint32_t cur_op = 3u, cur_n = 0u;
  while (d > 0) {
    if (j < tr.dlo[d] || (uint32_t)(j - tr.dlo[d]) >= tr.doff[d + 1] - tr.doff[d] || st > 2u) return -1;
    const uint8_t b = tr.bp[tr.doff[d] + (uint32_t)(j - tr.dlo[d])];
    const uint32_t from = (b >> (2u * st)) & 3u;
    if (st != cur_op || cur_n == 0x0FFFFFFFu) {
      if (cur_n) { if (n >= cap) return -1; rle[n++] = (cur_op << 28) | cur_n; }
      cur_op = st; cur_n = 0u;
    }
    ++cur_n;
    if (st == 2u) { d -= 2; j -= 1; } else if (st == 1u) { d -= 1; } else { d -= 1; j -= 1; }
    st = from;
  }
  if (cur_n) { if (n >= cap) return -1; rle[n++] = (cur_op << 28) | cur_n; }
  return d == 0 && j == 0 ? n : -1;
}

// A piece of an alignment's path, in the order the walk lays them down (postnuc_unit, `eng.piece`): an exact match, a search /
// alignment of the engine from (A0, B0) towards the target (tA, tB) that ended on (A1, B1), or a forced run between two corners.
// PIECE_VISIT (aln = -1) marks the walk arriving at the next cluster in reference order (A0 = the reference start of its first
// match): MUMmer walks the clusters of BOTH strands of a sequence pair in one list sorted by that start and prints alignments in
// the order it creates them, so an alignment's place in the .delta file is (the visit it was created in, forward strand first).
enum : uint32_t { PIECE_MATCH = 0, PIECE_SEARCH = 1, PIECE_FORCED = 2, PIECE_VISIT = 3 };
struct PnPiece {
  int32_t aln;                 // the alignment (index in the unit's al[]) it belongs to
  uint32_t kind, m_o;
  int32_t A0, B0, A1, B1;      // first and last base pair (inclusive corners; A1/B1 = where the engine finished)
  int32_t tA, tB;              // PIECE_SEARCH: the target the engine was given
  uint32_t cells, wmax;        // engine bookkeeping of the call (cells computed, widest anti-diagonal): sizes the traceback store
  uint32_t aux;                // PIECE_FORCED on the GPU: the slot of its deferred request (the band it certified with is found there)
};

// ---- the scans of extendClusters ---------------------------------------------------------------------------------------------
PG_HD bool pn_close_enough(int32_t a, int32_t b) {
  const int32_t lesser = a < b ? a : b, greater = a < b ? b : a;
  return greater < BREAK_LEN || lesser * GOOD_SCORE + (greater - lesser) * CONT_GAP_SCORE >= 0;
}
PG_HD bool pn_same_records(const Chain* chains, int a, int b) { return chains[a].rrec == chains[b].rrec && chains[a].qrec == chains[b].qrec; }

// The three scans of extendClusters, one candidate at a time (the scalar engines walk them in MUMmer's order; the GPU's wave
// engine evaluates 64 candidates per step and reduces to the same answer — the selection rules are order-free once stated as
// "the first close-enough candidate in scan order, else the smallest distance, ties to the earlier one in scan order").
struct PnCand { bool valid, close; int32_t dist, a, b; };
// isShadowedCluster: alignment x (same records) contains the cluster [sA, eA] x [sB, eB]
PG_HD bool pn_shadow_hit(const Chain* chains, const PnAln& x, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) {
  return pn_same_records(chains, x.chain, c) && x.eA >= eA && x.eB >= eB && x.sA <= sA && x.sB <= sB;
}
// getReverseTargetAlignment: alignment x as the target of a backward search that starts at (sA, sB)
PG_HD PnCand pn_reverse_cand(const Chain* chains, const PnAln& x, int c, int32_t sA, int32_t sB) {
  PnCand r{false, false, 0, x.eA, x.eB};
  if (!pn_same_records(chains, x.chain, c) || !(x.eA <= sA && x.eB <= sB)) return r;
  int32_t lesser = sA - x.eA, greater = sB - x.eB;
  if (lesser > greater) { const int32_t t = lesser; lesser = greater; greater = t; }
  r.valid = true; r.close = pn_close_enough(lesser, greater); r.dist = (greater << 1) - lesser;
  return r;
}
// getForwardTargetCluster: cluster tc as the target of a forward extension off a cluster that ends at (sA, sB); if its first
// match overlaps that end but its last one does not, the target is its first match that starts at or after the end in both
PG_HD PnCand pn_forward_cand(const Chain* chains, const Match* cm, int tc, int c, int32_t sA, int32_t sB) {
  PnCand r{false, false, 0, 0, 0};
  if (!pn_same_records(chains, tc, c)) return r;
  const Match* tm = cm + chains[tc].first;
  const int tn = chains[tc].count;
  int32_t eA = tm[0].r, eB = tm[0].q;
  if ((eA < sA || eB < sB) && tm[tn - 1].r >= sA && tm[tn - 1].q >= sB)
    for (int x = 0; x < tn && (eA < sA || eB < sB); ++x) { eA = tm[x].r; eB = tm[x].q; }
  if (!(eA >= sA && eB >= sB)) return r;
  int32_t lesser = eA - sA, greater = eB - sB;
  if (lesser > greater) { const int32_t t = lesser; lesser = greater; greater = t; }
  r.valid = true; r.close = pn_close_enough(lesser, greater); r.dist = (greater << 1) - lesser; r.a = eA; r.b = eB;
  return r;
}
// the scans in MUMmer's own order (what ScalarEngine / DiagEngine use; PnWaveEngine has lane-parallel forms of the same three)
struct PnScalarScans {
  PG_HD bool shadowed(const Chain* chains, const PnAln* al, int from, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    for (int t = from; t >= 0; --t) if (pn_shadow_hit(chains, al[t], c, sA, eA, sB, eB)) return true;
    return false;
  }
  // the EARLIEST of the unit's alignments al[0 .. n_al) that contains the cluster, -1 if none does (whatever the walk may see of them)
  PG_HD int shadow_first(const Chain* chains, const PnAln* al, int n_al, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    for (int t = 0; t < n_al; ++t) if (pn_shadow_hit(chains, al[t], c, sA, eA, sB, eB)) return t;
    return -1;
  }
  PG_HD int reverse_target(const Chain* chains, const PnAln* al, int cura, int c, int32_t sA, int32_t sB, int32_t dist) const {
    int tgt = -1;
    for (int t = cura - 1; t >= 0; --t) {
      const PnCand x = pn_reverse_cand(chains, al[t], c, sA, sB);
      if (!x.valid) continue;
      if (x.close) { tgt = t; break; }
      if (x.dist < dist) { tgt = t; dist = x.dist; }
    }
    return tgt;
  }
  PG_HD int forward_target(const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, int c, int32_t sA, int32_t sB,
                           int32_t dist, int32_t& targetA, int32_t& targetB) const {
    int targetk = -1;
    for (int k = curk + 1; k < n; ++k) {
      const PnCand x = pn_forward_cand(chains, cm, order[k], c, sA, sB);
      if (!x.valid) continue;
      if (x.close) { targetk = k; targetA = x.a; targetB = x.b; break; }
      if (x.dist < dist) { targetk = k; targetA = x.a; targetB = x.b; dist = x.dist; }
    }
    return targetk;
  }
};

// ---- one walk over BOTH strands (round 5) -----------------------------------------------------------------------------------
// MUMmer's extendClusters walks the clusters of one (reference record, query record) pair — a "synteny" — of BOTH query strands in
// one list sorted by reference start, and isShadowedCluster looks at that synteny's alignments from its CURRENT one (the alignment
// the latest non-skipped cluster left the walk on: the one it pushed, or the one its backward search merged into) backwards.  The
// engine walks each strand as a unit of its own (and all record pairs of a strand in one list), so "the current alignment" of the
// synteny may belong to the OTHER strand's walk, and a cluster's shadow test must see exactly the alignments of its own strand
// that were made before that one — rounds 3-4 scanned from the unit's own current alignment instead (deviation (1):
// the answers differ when an alignment of the cluster's strand and synteny lies beyond the current one, which takes a backward
// merge into an older alignment, or when a turn of the other strand moved the current alignment).
// The two walks of a pair run side by side and meet only where it matters.  Every walk keeps a log of its non-skipped TURNS (a
// turn = the cluster the walk's `prev` cursor stands on plus the chain of forward targets it fuses: MUMmer's while-loop from one
// `curc = ++prev` to the next) keyed by the turn's place in the joint order, key = 2 * (reference start of the turn's first
// cluster) + strand (equal starts: forward strand first, as a stable sort of MUMmer's cluster list — forward clusters are read
// first — leaves them), and publishes the key of the turn it is in.  A shadow test first looks whether ANY alignment of the unit
// contains the cluster; only then (rare) does it need the synteny's current alignment: the later of the latest entry of its own
// log and the latest entry of the other walk's log below its own key — for the latter it waits until the other walk has passed
// that key (the waits cannot deadlock: each walk waits only for keys below its own).  If the current alignment is the other
// strand's, the test sees its own alignments born (turn key) before that one.
struct PnTurn { int32_t key, rrec, qrec, aln, born, pad; };      // after the turn: the synteny's current alignment = this unit's al[aln], born in the turn `born`
constexpr int32_t PN_KEY_DONE = 0x7FFFFFFF;
PG_HD int32_t pn_turn_key(int32_t ref_start, int strand) { return 2 * ref_start + (strand ? 1 : 0); }
// PRIM: how the two walks reach each other's words — the host (two threads, std::atomic) and the GPU (two waves of a workgroup:
// progress words in LDS, logs in global memory read past the vector L1) supply
//   int32_t ld(const int32_t* p)      a load that sees the other walk's latest store        void st(int32_t* p, int32_t v)   its store
//   void publish_log(int32_t n_log)   make this walk's log (its stores so far) visible and announce its length;   void publish_key(int32_t key)   its position (after the log)
//   int32_t other_key() / other_nlog()         the other walk's position / log length (key first)         void pause()
template <typename PRIM>
struct PnPairSync {
  PRIM prim;
  PnTurn* log;                 // this walk's turns
  const PnTurn* other_log;     // the other walk's (nullptr: there is none — a walk by itself, e.g. the rehearsal)
  int32_t* born;               // per alignment of this unit: the key of the turn that pushed it
  int32_t n_log = 0, key = 0;
  long waits = 0, asked = 0, differs = 0;   // (development counters; differs: shadow tests that the unit's own current alignment would have answered otherwise)
  // Publication is LAZY: the log's length and this walk's position go out (behind a store fence) every PUBLISH_EVERY-th logged turn,
  // before this walk waits for the other one (else two waiting walks could each sit on news the other needs) and at the end — a
  // fence per turn (10^7 turns per C4 launch) cost the walk kernel a third of its time, and the other walk only ever looks at
  // this one's log inside the rare shadow test that found a containing alignment: it just waits a little longer there.
  static constexpr int PUBLISH_EVERY = 16;
  int32_t unpublished = 0;
  PG_HD void publish() { prim.publish_log(n_log); prim.publish_key(key); unpublished = 0; }
  PG_HD void begin_turn(int32_t k) { key = k; pushed_aln = -1; if (unpublished >= PUBLISH_EVERY) publish(); }
  int32_t pushed_aln = -1;     // the alignment this turn pushed (its birth key is `key`: not read back through memory in the same turn)
  PG_HD void pushed(int aln) { prim.st(born + aln, key); pushed_aln = aln; }
  PG_HD void end_turn(int32_t rrec, int32_t qrec, int aln) {
    PnTurn* e = log + n_log;
    const int32_t b = aln == pushed_aln ? key : prim.ld(born + aln);      // (else: a merge target, born in an earlier turn)
    prim.st(&e->key, key); prim.st(&e->rrec, rrec); prim.st(&e->qrec, qrec); prim.st(&e->aln, aln); prim.st(&e->born, b);
    ++n_log; ++unpublished;
  }
  PG_HD void finish() { key = PN_KEY_DONE; publish(); }
  // isShadowedCluster's scan range for a cluster of synteny (rrec, qrec) at the current turn: the largest index of this unit's
  // alignment list the test may look at (-1: none).  n_al: alignments of this unit so far.
  PG_HD int visible(int32_t rrec, int32_t qrec, int n_al, int /* cura */) {
    ++asked;
    publish();      // (this walk's own log entries are read back below, and the other walk may be waiting for this one's news)
    int32_t o_key = -1, o_aln = -1;
    for (int t = n_log - 1; t >= 0; --t)
      if (prim.ld(&log[t].rrec) == rrec && prim.ld(&log[t].qrec) == qrec) { o_key = prim.ld(&log[t].key); o_aln = prim.ld(&log[t].aln); break; }
    int32_t x_key = -1, x_born = 0;
    if (other_log) {
      while (prim.other_key() <= key) { ++waits; prim.pause(); }
      for (int t = prim.other_nlog() - 1; t >= 0; --t) {
        const int32_t k = prim.ld(&other_log[t].key);
        if (k > key) continue;      // (the other walk is ahead: turns that come after this one in the joint order)
        if (prim.ld(&other_log[t].rrec) == rrec && prim.ld(&other_log[t].qrec) == qrec) { x_key = k; x_born = prim.ld(&other_log[t].born); break; }
      }
    }
    if (x_key > o_key) {      // the synteny's current alignment is the other strand's: this strand's alignments born before it
      int cnt = 0;
      while (cnt < n_al && prim.ld(born + cnt) < x_born) ++cnt;
      return cnt - 1;
    }
    return o_aln;      // this strand's (-1: the synteny has no alignment yet)
  }
};
// a walk by itself (the DP-free rehearsal: a wrong guess costs time, never a result): sees its own list up to its current alignment
struct PnNoSync {
  long differs = 0;
  PG_HD void begin_turn(int32_t) {}
  PG_HD void pushed(int) {}
  PG_HD void end_turn(int32_t, int32_t, int) {}
  PG_HD void finish() {}
  PG_HD int visible(int32_t, int32_t, int, int cura) const { return cura; }
};
#if !defined(__HIP_DEVICE_COMPILE__)
// the host statement's two walks of a pair: one thread per strand (tools/anim_debug and the CPU baseline's host build), GCC / clang atomics
struct PnHostShared { int32_t key[2] = {-1, -1}, nlog[2] = {0, 0}; };
struct PnHostPrim {
  PnHostShared* sh;
  int me;
  int32_t ld(const int32_t* p) const { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
  void st(int32_t* p, int32_t v) const { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
  void publish_log(int32_t n_log) const { __atomic_store_n(&sh->nlog[me], n_log, __ATOMIC_RELEASE); }
  void publish_key(int32_t key) const { __atomic_store_n(&sh->key[me], key, __ATOMIC_RELEASE); }
  int32_t other_key() const { return __atomic_load_n(&sh->key[1 - me], __ATOMIC_ACQUIRE); }
  int32_t other_nlog() const { return __atomic_load_n(&sh->nlog[1 - me], __ATOMIC_ACQUIRE); }
  void pause() const { std::this_thread::yield(); }
};
#endif

// ---- forced alignments: the whole rectangle, computed as a certified band ------------------------------------------------
// A FORCED alignment (the forward re-alignment of a backward extension, up to 10 000 x 10 000) is the optimal global path of its
// rectangle under the tie order above; MUMmer fills the whole rectangle (no trimming, no break).  The same path comes out of
// the cells within w diagonals of the corner-to-corner span [min(0, M - N), max(0, M - N)] whenever every path that leaves
// them scores strictly less than the score S_w found inside: such a path spends at least |M - N| + 2 (w + 1) gap bases in at
// least two runs, so it scores at most  3 (min(N, M) - (w + 1)) - 7 (|M - N| + 2 (w + 1)) - 6.  If S_w beats that bound, the
// optimal path, every comparison along it (a competitor's banded value is <= its full value, which lost under the same tie
// order) and therefore the riding error count are those of the full rectangle; otherwise w doubles.  Near-identical genomes
// certify at w = 32 where the rectangle has 10^8 cells.
// Band sequence: w = 28, 60, 124, 252, 508, 1020, then the whole rectangle (band cells = w + |M - N| / 2 + 2: chosen so that
// successive bands fill the GPU's register engines of 127 / 255 / 511 / 1023 cells).
constexpr int32_t FORCED_BAND_FIRST = 28;
PG_HD int32_t forced_band_next(int32_t w) { return 2 * w + 4; }
// The band to try after band w gave score S without a certificate: a wider band can only score higher, so the first width w'
// with  17 (w' + 1) > 3 min(N, M) - 7 |M - N| - 6 - S  is CERTAIN to certify — and nothing wider is needed: the run's cells grow
// with w'.  (Rounds 1-3 went up a doubling sequence 28, 60, 124, ... to fill fixed register engines: a run that needed w = 1100
// was computed at 2044.  The diagonal-window engines of round 4 take any width.)
PG_HD int32_t forced_band_after(int32_t w, int32_t N, int32_t M, int32_t S) {
  const int64_t mn = N < M ? N : M, df = N < M ? M - N : N - M;
  const int64_t need = ((int64_t)GOOD_SCORE * mn + (int64_t)CONT_GAP_SCORE * df + 2 * (OPEN_GAP_SCORE - CONT_GAP_SCORE) - S) / (GOOD_SCORE - 2 * CONT_GAP_SCORE);
  const int64_t mx = N > M ? N : M;
  int64_t nw = need > (int64_t)w + 1 ? need : (int64_t)w + 1;
  // a corner word at the score floor says nothing about how far off the band is (S is a bound, not a score): double, so that a run
  // that has to end as "give up" (forced_verdict) gets to the whole rectangle in a few passes instead of four diagonals at a time
  if (S <= -(int32_t)SCORE_BIAS && nw < 2 * ((int64_t)w + 1)) nw = 2 * ((int64_t)w + 1);
  nw = (nw + 3) & ~(int64_t)3;
  return (int32_t)(nw < mx ? nw : mx);
}
PG_HD int64_t forced_outside_bound(int32_t N, int32_t M, int32_t w) {
  const int64_t mn = N < M ? N : M, df = N < M ? M - N : N - M;
  return (int64_t)GOOD_SCORE * (mn - (w + 1)) + (int64_t)CONT_GAP_SCORE * (df + 2 * (int64_t)(w + 1)) + 2 * (OPEN_GAP_SCORE - CONT_GAP_SCORE);
}
// What the band loop of a forced run does with the result of band w (score: of the corner's word):  0 = accept, 1 = grow the band,
// 2 = give up: even the whole rectangle leaves the corner unreachable, i.e. every path to it fell out of the score field — the
// caller flags the unit (PG_E_CAPACITY on the pair) instead of adding the error count of a word that holds none.
PG_HD int forced_verdict(bool reached, int32_t score, bool whole, int32_t N, int32_t M, int32_t w) {
  if (score <= -(int32_t)SCORE_BIAS) return whole ? 2 : 1;
  return (whole || (reached && (int64_t)score > forced_outside_bound(N, M, w))) ? 0 : 1;
}
// band of a forced run in cell coordinates: diagonal k = j - i = 2 j - Dct within [kmin - w, kmax + w]
PG_HD void forced_band_clip(int32_t Dct, int32_t N, int32_t M, int32_t w, int32_t& lo, int32_t& hi) {
  const int32_t kmin = (M - N < 0 ? M - N : 0) - w, kmax = (M - N > 0 ? M - N : 0) + w;
  const int32_t blo = (Dct + kmin + 1) >> 1, bhi = (Dct + kmax) >> 1;   // ceil((Dct + kmin) / 2), floor((Dct + kmax) / 2)
  if (lo < blo) lo = blo;
  if (hi > bhi) hi = bhi;
}

// ---- the scalar engine (host statement; also the definition the wave engine follows cell for cell) -----------------------
// SEQ: a_ok(p) / a_base(p) / b_ok(p) / b_base(p) on stream positions (b: strand coordinates).
template <typename RefT, typename QryT>
struct ScalarEngine {
  const RefT& R;
  const QryT& Q;
  Cell *d0, *d1, *d2;     // three anti-diagonals of capacity cap each (rotating)
  int32_t cap;
  int32_t overflow = 0;
  long cells = 0;
  const PnGap* gaps = nullptr;     // match-to-match alignments computed beforehand, by match slot (host experiments: what the GPU's gap pre-pass leaves)
  PG_HD bool gap_ready(int32_t slot, PnGap& g) const { if (!gaps || pieces) return false; g = gaps[slot]; return g.reached >= 0; }
  // gap_run — the DEFINITION of the hook (see postnuc_unit): the walk stands on match m of a cluster (matches mm[0 .. count), their
  // match-to-match alignments at slots first_slot + t) with its alignment ending exactly where mm[m] starts.  Returns j >= m such
  // that every alignment from match t to match t + 1, m <= t < j, is ready, reached its target and ended on it; their errors are
  // added.  The walk then goes on at match j as if it had taken those steps one by one (a wave does them 64 at a time).
  PG_HD int gap_run(const Match* mm, int32_t first_slot, int m, int count, int32_t& errors) const {
    if (!gaps || pieces) return m;
    int t = m;
    for (; t + 1 < count; ++t) {
      const PnGap g = gaps[first_slot + t];
      if (!(g.reached > 0 && g.eA == mm[t + 1].r && g.eB == mm[t + 1].q)) break;
      errors += g.errors;
    }
    return t;
  }
  const PnFwd* fwd = nullptr;      // forward extensions computed beforehand (postnuc_forward_all), by position in `order`
  PG_HD bool fwd_ready(int k, PnFwd& f) const { if (!fwd) return false; f = fwd[k]; return true; }
  const PnBwd* bwd = nullptr;      // backward searches run ahead (postnuc_rehearse + one align per predicted call), by position in `order`
  PG_HD bool bwd_ready(int k, PnBwd& b) const { if (!bwd) return false; b = bwd[k]; return b.state == 2; }
  PG_HD void bwd_key(int, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  long searches = 0, search_cells = 0;      // backward searches this engine ran itself (host experiments)
  PnTrace* trace = nullptr;        // set: the next run() stores its backpointers there
  PnPiece* pieces = nullptr;       // set: the walk's pieces are listed here (piece_cap entries; n_pieces counts on past it)
  int32_t piece_cap = 0, n_pieces = 0;
  uint32_t last_cells = 0, last_wmax = 0;    // of the latest run()
  PG_HD void piece(uint32_t kind, int32_t aln, int32_t A0, int32_t B0, int32_t A1, int32_t B1, int32_t tA, int32_t tB, unsigned m_o) {
    if (!pieces) return;
    if (n_pieces < piece_cap) pieces[n_pieces] = PnPiece{aln, kind, m_o, A0, B0, A1, B1, tA, tB, last_cells, last_wmax, 0u};
    ++n_pieces;
  }
  // a forced alignment between two known corners: its error count (see the call sites in postnuc_unit)
  PG_HD int32_t forced_errors(int32_t A0, int32_t A1, int32_t B0, int32_t B1, PnAln*) {
    int32_t err = 0, a = A1, b = B1;
    align(A0, a, B0, b, FORCED_FORWARD_ALIGN, err);
    return err;
  }
  // the scans of extendClusters in MUMmer's own order
  PG_HD bool shadowed(const Chain* chains, const PnAln* al, int from, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadowed(chains, al, from, c, sA, eA, sB, eB); }
  PG_HD int shadow_first(const Chain* chains, const PnAln* al, int n_al, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadow_first(chains, al, n_al, c, sA, eA, sB, eB); }
  PG_HD int reverse_target(const Chain* chains, const PnAln* al, int cura, int c, int32_t sA, int32_t sB, int32_t dist) const {
    return PnScalarScans().reverse_target(chains, al, cura, c, sA, sB, dist); }
  PG_HD int forward_target(const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, int c, int32_t sA, int32_t sB,
                           int32_t dist, int32_t& targetA, int32_t& targetB) const {
    return PnScalarScans().forward_target(chains, cm, order, n, curk, c, sA, sB, dist, targetA, targetB); }
  PG_HD bool same(int64_t pa, int64_t pb) const { return R.clean(pa) && Q.clean(pb) && R.base(pa) == Q.base(pb); }

  // Aligns A[Astart .. Aend] with B[Bstart .. Bend] (inclusive; walking backwards when DIRECTION_BIT is clear).  Returns whether
  // the target corner was reached; Aend / Bend = the finish position; errors = errors of the path to it (not in SEARCH mode
  // — the value is computed anyway and ignored by the callers).
  PG_HD bool align(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t& errors) {
    if (!(m_o & FORCED_BIT)) {
      const bool r = run(Astart, Aend, Bstart, Bend, m_o, -1, errors);
      if (m_o & SEARCH_BIT) { ++searches; search_cells += last_cells; }
      return r;
    }
    const bool fwd = m_o & DIRECTION_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    for (int32_t w = FORCED_BAND_FIRST;;) {
      int32_t a = Aend, b = Bend, score = 0;
      const bool whole = w >= (N > M ? N : M);
      const bool reached = run(Astart, a, Bstart, b, m_o, whole ? -1 : w, errors, &score);
      const int v = overflow ? 0 : forced_verdict(reached, score, whole, N, M, w);
      if (v == 2) { overflow = 1; Aend = a; Bend = b; errors = 0; return false; }
      if (v == 0) { Aend = a; Bend = b; return reached; }
      w = forced_band_after(w, N, M, score);
    }
  }
  // band_w < 0: MUMmer's own band (dynamic, trimmed unless forced); >= 0: a forced run confined to the certified band
  PG_HD bool run(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors,
                 int32_t* score_out = nullptr) {
    const bool fwd = m_o & DIRECTION_BIT, forced = m_o & FORCED_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    Cell *p2 = d0, *p1 = d1, *cur = d2;
    int32_t p2lo = 0, p2hi = -1, p1lo = 0, p1hi = 0;
    p1[0] = Cell{0u, 0u, w_make(0, 0, ST_MATCH), w_make(0, 0, ST_MATCH)};
    int32_t high = -(1 << 30), FinishCt = 0, FinishJ = 0;
    uint32_t high_w = 0;
    int32_t jlo = 0, jhi = 1, Dct;
    PnTrace* const tr = trace;
    uint32_t wmax = 0;
    uint64_t ncells = 0;
    if (tr) { tr->used = 0; tr->overflow = 0; tr->doff[0] = 0; tr->dlo[0] = 0; if (tr->dcap > 1) tr->doff[1] = 0; }
    for (Dct = 1; Dct <= N + M && (forced || Dct - FinishCt <= BREAK_LEN) && jlo <= jhi; ++Dct) {
      int32_t lo = jlo, hi = jhi;
      if (lo < Dct - N) lo = Dct - N;
      if (lo < 0) lo = 0;
      if (hi > M) hi = M;
      if (hi > Dct) hi = Dct;
      if (band_w >= 0) forced_band_clip(Dct, N, M, band_w, lo, hi);
      if (lo > hi) break;
      if (hi - lo + 1 > cap) { overflow = 1; break; }
      if ((uint32_t)(hi - lo + 1) > wmax) wmax = (uint32_t)(hi - lo + 1);
      ncells += (uint64_t)(hi - lo + 1);
      if (tr) {
        if (Dct + 2 > tr->dcap || tr->used + (uint64_t)(hi - lo + 1) > tr->bp_cap) { tr->overflow = 1; overflow = 1; break; }
        tr->doff[Dct] = (uint32_t)tr->used; tr->dlo[Dct] = lo;
      }
      for (int32_t j = lo; j <= hi; ++j) {
        const int32_t i = Dct - j;
        const bool hasL = j >= 1 && j - 1 >= p1lo && j - 1 <= p1hi, hasU = i >= 1 && j >= p1lo && j <= p1hi;
        const bool hasG = i >= 1 && j >= 1 && j - 1 >= p2lo && j - 1 <= p2hi;
        bool sm = false;
        if (hasG) sm = same(fwd ? (int64_t)Astart + i - 1 : (int64_t)Astart - i + 1, fwd ? (int64_t)Bstart + j - 1 : (int64_t)Bstart - j + 1);
        uint8_t bp = 0;
        const Cell c = tr ? cell_update_bp(hasL ? p1[j - 1 - p1lo] : Cell{0, 0, 0, 0}, hasU ? p1[j - p1lo] : Cell{0, 0, 0, 0},
                                           hasG ? p2[j - 1 - p2lo].X : 0u, sm, bp)
                          : cell_update(hasL ? p1[j - 1 - p1lo] : Cell{0, 0, 0, 0}, hasU ? p1[j - p1lo] : Cell{0, 0, 0, 0},
                                        hasG ? p2[j - 1 - p2lo].X : 0u, sm);
        if (tr) tr->bp[tr->used + (uint64_t)(j - lo)] = bp;
        cur[j - lo] = c;
        const int32_t s = w_score(c.X);
        if (s >= high) { high = s; high_w = c.X; FinishCt = Dct; FinishJ = j; }
      }
      if (tr) { tr->used += (uint64_t)(hi - lo + 1); tr->doff[Dct + 1] = (uint32_t)tr->used; }
      cells += hi - lo + 1;
      int32_t tlo = lo, thi = hi;
      if (!forced) {
        while (tlo <= thi && high - w_score(cur[tlo - lo].X) > MAX_DIFF) ++tlo;
        while (thi >= tlo && high - w_score(cur[thi - lo].X) > MAX_DIFF) --thi;
      }
      jlo = tlo; jhi = thi + 1;
      if (tlo > thi) { jlo = 1; jhi = 0; }
      Cell* t = p2; p2 = p1; p1 = cur; cur = t;
      p2lo = p1lo; p2hi = p1hi; p1lo = lo; p1hi = hi;
    }
    --Dct;
    bool reached = false;
    uint32_t fin_w = high_w;
    if (Dct == N + M && !overflow) {
      if (!(m_o & OPTIMAL_BIT)) { reached = true; FinishCt = N + M; FinishJ = M; fin_w = p1[M - p1lo].X; }
      else if (FinishCt == Dct) reached = true;
    }
    const int32_t fi = FinishCt - FinishJ, fj = FinishJ;
    Aend = fwd ? Astart + fi - 1 : Astart - fi + 1;
    Bend = fwd ? Bstart + fj - 1 : Bstart - fj + 1;
    errors = (int32_t)w_errors(fin_w);
    if (score_out) *score_out = w_score(fin_w);
    last_cells = ncells > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ncells; last_wmax = wmax;
    if (tr) { tr->finish_ct = FinishCt; tr->finish_j = FinishJ; }
    return reached;
  }
};

// ---- the same engine laid out by DIAGONAL (the statement of the quad-lane kernel, pga_postnuc_quad.inc) ---------------------
// Cell (i, j) lives on diagonal k = j - i, slot l = k + DIAG_HALF of a fixed window of DIAG_WINDOW diagonals around the start
// cell; anti-diagonal Dct touches the slots of its own parity.  Per slot: X = best-state word of the latest cell on that
// diagonal (two anti-diagonals old when the slot's turn comes: the diagonal neighbour G), D / I = the gap states of that cell
// (read once, by the neighbours l - 1 / l + 1 on the next anti-diagonal).  The gap states need only two candidates each:
//   DELETE = max(D(left) + CONT, X(left) + OPEN),  INSERT = max(I(up) + CONT, X(up) + OPEN)
// — the third candidate of MUMmer's scoreEdit is dominated (a gap state re-opened from itself costs OPEN < CONT) and the state
// labels settle every tie as before.  A slot whose turn comes while it is outside the anti-diagonal's range is zeroed, which
// is what "not computed" means to its later readers.  A range that would leave the window makes the call start over in the
// general engine (overflow = true): MUMmer's band is 150-190 diagonals wide wherever two sequences align and follows the
// alignment's net indels, so 256 diagonals hold it unless an extension drifts more than ~40 diagonals off its start.
constexpr int32_t DIAG_WINDOW = 256, DIAG_HALF = 128;
struct DiagSlot { uint32_t X, D, I; };
PG_HD void diag_cell(const DiagSlot& left, const DiagSlot& up, uint32_t G, bool same, DiagSlot& out) {
  const uint32_t dc = w_gap(left.D, CONT_GAP_SCORE), dx = w_gap(left.X, OPEN_GAP_SCORE);
  const uint32_t ic = w_gap(up.I, CONT_GAP_SCORE), ix = w_gap(up.X, OPEN_GAP_SCORE);
  const uint32_t d = w_relabel(dc > dx ? dc : dx, ST_DELETE), i = w_relabel(ic > ix ? ic : ix, ST_INSERT);
  const uint32_t m = w_relabel(w_step(G, same), ST_MATCH);
  out.D = d; out.I = i; out.X = w_max3(d, i, m);
}

template <typename RefT, typename QryT>
struct DiagScalarEngine {
  const RefT& R;
  const QryT& Q;
  int32_t overflow = 0;
  long cells = 0, fallbacks = 0;
  PG_HD bool same(int64_t pa, int64_t pb) const { return R.clean(pa) && Q.clean(pb) && R.base(pa) == Q.base(pb); }
  // false: the band left the window (nothing is returned)
  PG_HD bool run(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors,
                 int32_t& score, bool& reached) {
    const bool fwd = m_o & DIRECTION_BIT, forced = m_o & FORCED_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    DiagSlot S[DIAG_WINDOW];
    for (int l = 0; l < DIAG_WINDOW; ++l) S[l] = DiagSlot{0u, 0u, 0u};
    S[DIAG_HALF].X = w_make(0, 0, ST_MATCH);
    int32_t high = -(1 << 30), FinishCt = 0, FinishK = 0;
    uint32_t high_w = 0;
    int32_t ka = 0, kb = 0;     // surviving diagonals of the latest anti-diagonal
    bool empty = false;
    int32_t Dct;
    for (Dct = 1; Dct <= N + M && (forced || Dct - FinishCt <= BREAK_LEN) && !empty; ++Dct) {
      int32_t lo = ka - 1, hi = kb + 1;                 // k range; cells exist on k == Dct (mod 2): ka / kb had parity Dct - 1
      const int32_t c1 = -Dct > Dct - 2 * N ? -Dct : Dct - 2 * N, c2 = 2 * M - Dct < Dct ? 2 * M - Dct : Dct;
      if (lo < c1) lo = c1;
      if (hi > c2) hi = c2;
      if (band_w >= 0) {
        const int32_t kmin = (M - N < 0 ? M - N : 0) - band_w, kmax = (M - N > 0 ? M - N : 0) + band_w;
        if (lo < kmin) lo = kmin;
        if (hi > kmax) hi = kmax;
      }
      if ((lo + Dct) & 1) ++lo;
      if ((hi + Dct) & 1) --hi;
      if (lo > hi) break;
      if (lo < -DIAG_HALF || hi >= DIAG_HALF) return false;
      DiagSlot nw[DIAG_WINDOW / 2];
      const int par = Dct & 1;                           // slots l = k + DIAG_HALF have the parity of k (DIAG_HALF is even)
      int32_t bestk = -(1 << 30), best_l = -1; uint32_t bestw = 0;
      for (int l = par; l < DIAG_WINDOW; l += 2) {
        const int32_t k = l - DIAG_HALF;
        DiagSlot out{0u, 0u, 0u};
        if (k >= lo && k <= hi) {
          const int32_t i = (Dct - k) / 2, j = (Dct + k) / 2;
          const DiagSlot zero{0u, 0u, 0u};
          const DiagSlot& left = l >= 1 ? S[l - 1] : zero;
          const DiagSlot& up = l + 1 < DIAG_WINDOW ? S[l + 1] : zero;
          bool sm = false;
          if (i >= 1 && j >= 1) sm = same(fwd ? (int64_t)Astart + i - 1 : (int64_t)Astart - i + 1, fwd ? (int64_t)Bstart + j - 1 : (int64_t)Bstart - j + 1);
          diag_cell(j >= 1 ? left : zero, i >= 1 ? up : zero, (i >= 1 && j >= 1) ? S[l].X : 0u, sm, out);
          ++cells;
          const int32_t sc = w_score(out.X);
          if (sc >= bestk) { bestk = sc; best_l = l; bestw = out.X; }
        }
        nw[l >> 1] = out;
      }
      for (int l = par; l < DIAG_WINDOW; l += 2) S[l] = nw[l >> 1];
      if (best_l >= 0 && bestk >= high) { high = bestk; high_w = bestw; FinishCt = Dct; FinishK = best_l - DIAG_HALF; }
      int32_t ta = lo, tb = hi;
      if (!forced) {
        while (ta <= tb && high - w_score(S[ta + DIAG_HALF].X) > MAX_DIFF) ta += 2;
        while (tb >= ta && high - w_score(S[tb + DIAG_HALF].X) > MAX_DIFF) tb -= 2;
      }
      ka = ta; kb = tb;
      if (ta > tb) empty = true;
    }
    --Dct;
    reached = false;
    uint32_t fin_w = high_w;
    if (Dct == N + M) {
      if (!(m_o & OPTIMAL_BIT)) { reached = true; FinishCt = N + M; FinishK = M - N; fin_w = S[M - N + DIAG_HALF].X; }
      else if (FinishCt == Dct) reached = true;
    }
    const int32_t fi = (FinishCt - FinishK) / 2, fj = (FinishCt + FinishK) / 2;
    Aend = fwd ? Astart + fi - 1 : Astart - fi + 1;
    Bend = fwd ? Bstart + fj - 1 : Bstart - fj + 1;
    errors = (int32_t)w_errors(fin_w);
    score = w_score(fin_w);
    return true;
  }
};

// The diagonal-window engine with the general one behind it: what the GPU's quad-lane kernel + wave-engine fallback compute.
template <typename RefT, typename QryT>
struct DiagEngine {
  DiagScalarEngine<RefT, QryT> fast;
  ScalarEngine<RefT, QryT> slow;
  PG_HD bool gap_ready(int32_t, PnGap&) const { return false; }     // (scalar engines align match to match as they go)
  PG_HD void piece(uint32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  PG_HD bool bwd_ready(int, PnBwd&) const { return false; }
  PG_HD void bwd_key(int, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  const PnFwd* fwd = nullptr;      // forward extensions computed beforehand (postnuc_forward_all), by position in `order`
  PG_HD bool fwd_ready(int k, PnFwd& f) const { if (!fwd) return false; f = fwd[k]; return true; }
  // a forced alignment between two known corners: its error count (see the call sites in postnuc_unit)
  PG_HD int32_t forced_errors(int32_t A0, int32_t A1, int32_t B0, int32_t B1, PnAln*) {
    int32_t err = 0, a = A1, b = B1;
    align(A0, a, B0, b, FORCED_FORWARD_ALIGN, err);
    return err;
  }
  // the scans of extendClusters in MUMmer's own order
  PG_HD bool shadowed(const Chain* chains, const PnAln* al, int from, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadowed(chains, al, from, c, sA, eA, sB, eB); }
  PG_HD int shadow_first(const Chain* chains, const PnAln* al, int n_al, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadow_first(chains, al, n_al, c, sA, eA, sB, eB); }
  PG_HD int reverse_target(const Chain* chains, const PnAln* al, int cura, int c, int32_t sA, int32_t sB, int32_t dist) const {
    return PnScalarScans().reverse_target(chains, al, cura, c, sA, sB, dist); }
  PG_HD int forward_target(const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, int c, int32_t sA, int32_t sB,
                           int32_t dist, int32_t& targetA, int32_t& targetB) const {
    return PnScalarScans().forward_target(chains, cm, order, n, curk, c, sA, sB, dist, targetA, targetB); }
  long stat_cells[16] = {0}, stat_calls[16] = {0};   // by class: 0 trimmed search / alignment, 1 + log2(w / 32) forced band, 15 whole
  PG_HD bool run(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors, int32_t& score) {
    int32_t a = Aend, b = Bend;
    bool reached = false;
    const long c0 = fast.cells + slow.cells;
    int cls = 0;
    if (m_o & FORCED_BIT) { cls = 15; int32_t w = FORCED_BAND_FIRST; for (int t = 0; t < 12; ++t, w = forced_band_next(w)) if (band_w == w) cls = 1 + t; }
    struct Tally { DiagEngine* e; int cls; long c0; ~Tally() { e->stat_cells[cls] += e->fast.cells + e->slow.cells - c0; e->stat_calls[cls] += 1; } } tally{this, cls, c0};
    if (fast.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached)) { Aend = a; Bend = b; return reached; }
    ++fast.fallbacks;
    return slow.run(Astart, Aend, Bstart, Bend, m_o, band_w, errors, &score);
  }
  PG_HD bool align(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t& errors) {
    int32_t score = 0;
    if (!(m_o & FORCED_BIT)) return run(Astart, Aend, Bstart, Bend, m_o, -1, errors, score);
    const bool fwd = m_o & DIRECTION_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    for (int32_t w = FORCED_BAND_FIRST;;) {
      int32_t a = Aend, b = Bend;
      const bool whole = w >= (N > M ? N : M);
      const bool reached = run(Astart, a, Bstart, b, m_o, whole ? -1 : w, errors, score);
      const int v = slow.overflow ? 0 : forced_verdict(reached, score, whole, N, M, w);
      if (v == 2) { slow.overflow = 1; Aend = a; Bend = b; errors = 0; return false; }
      if (v == 0) { Aend = a; Bend = b; return reached; }
      w = forced_band_after(w, N, M, score);
    }
  }
};

// ---- postnuc: extendClusters ---------------------------------------------------------------------------------------------
// chains[order[k]], k = 0 .. n-1: the unit's clusters by the reference start of their first match (ties: extraction order);
// cm: their matches.  BOUNDS(c, r_lo, r_hi, q_lo, q_hi): the records of chain c as half-open stream ranges (q: strand
// coordinates).  fused[n] / al[max_al]: scratch and output.  Returns the number of alignments (al[] in creation order, as
// MUMmer prints them), or -1 - count when max_al was too small.
// The walk rehearsed without its backward searches: every other call goes to the real engine (which has them ready: the match-to-match
// and forward pre-passes ran before), a backward search is noted (bwd[k] = its arguments) and answered "found nothing".
// engines that can take a run of reached match-to-match alignments at once (ScalarEngine::gap_run is the definition)
template <typename T> T& pn_declref();      // (unevaluated contexts only)
template <typename E, typename = void> struct pn_has_gap_run { static constexpr bool value = false; };
template <typename E>
struct pn_has_gap_run<E, decltype((void)pn_declref<E>().gap_run(pn_declref<const Match*>(), 0, 0, 0, pn_declref<int32_t>()))> {
  static constexpr bool value = true;
};
template <typename ENG>
struct PnRehearsal {
  ENG& e;
  PnBwd* out;
  PG_HD bool gap_ready(int32_t slot, PnGap& g) const { return e.gap_ready(slot, g); }
  template <typename E2 = ENG>
  PG_HD auto gap_run(const Match* mm, int32_t first_slot, int m, int count, int32_t& errors) const -> decltype(pn_declref<E2>().gap_run(mm, first_slot, m, count, errors)) {
    return e.gap_run(mm, first_slot, m, count, errors); }
  PG_HD bool fwd_ready(int k, PnFwd& f) const { return e.fwd_ready(k, f); }
  PG_HD bool bwd_ready(int, PnBwd&) const { return false; }
  PG_HD void bwd_key(int k, int32_t sA, int32_t sB, int32_t tA, int32_t tB, unsigned m_o) {
#if defined(__HIP_DEVICE_COMPILE__)
    if ((threadIdx.x & 63) != 0) return;      // (a wave rehearses as one: every lane holds the same values)
#endif
    out[k] = PnBwd{sA, sB, tA, tB, m_o, 0, 0, 0, 1};
  }
  PG_HD void piece(uint32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  PG_HD int32_t forced_errors(int32_t, int32_t, int32_t, int32_t, PnAln*) { return 0; }
  PG_HD bool shadowed(const Chain* chains, const PnAln* al, int from, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return e.shadowed(chains, al, from, c, sA, eA, sB, eB); }
  PG_HD int shadow_first(const Chain* chains, const PnAln* al, int n_al, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return e.shadow_first(chains, al, n_al, c, sA, eA, sB, eB); }
  PG_HD int reverse_target(const Chain* chains, const PnAln* al, int cura, int c, int32_t sA, int32_t sB, int32_t dist) const {
    return e.reverse_target(chains, al, cura, c, sA, sB, dist); }
  PG_HD int forward_target(const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, int c, int32_t sA, int32_t sB,
                           int32_t dist, int32_t& targetA, int32_t& targetB) const {
    return e.forward_target(chains, cm, order, n, curk, c, sA, sB, dist, targetA, targetB); }
  PG_HD bool align(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t& errors) {
    if (m_o & SEARCH_BIT) { Aend = Astart; Bend = Bstart; errors = 0; return false; }
    return e.align(Astart, Aend, Bstart, Bend, m_o, errors);
  }
};

// extendForward off the last match of cluster order[curk]: target search (getForwardTargetCluster), clamps, alignment
template <typename ENG, typename BOUNDS>
PG_HD PnFwd postnuc_forward(ENG& eng, const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, BOUNDS&& bounds,
                            int aln = -1) {
  const int c = order[curk];
  const Chain& C = chains[c];
  const Match& ml = cm[C.first + C.count - 1];
  int32_t r_lo, r_hi, q_lo, q_hi;
  bounds(c, r_lo, r_hi, q_lo, q_hi);
  unsigned m_o = FORWARD_ALIGN;
  int32_t targetA = r_hi - 1, targetB = q_hi - 1;
  const int32_t sA = ml.r + ml.len - 1, sB = ml.q + ml.len - 1;
  const int32_t d0 = (targetA - sA) < (targetB - sB) ? (targetA - sA) : (targetB - sB);
  const int targetk = eng.forward_target(chains, cm, order, n, curk, c, sA, sB, d0, targetA, targetB);
  if (targetk < 0) m_o |= OPTIMAL_BIT;
  bool overflow = false;
  if (targetA - sA + 1 > MAX_ALIGNMENT_LENGTH) { targetA = sA + MAX_ALIGNMENT_LENGTH - 1; overflow = true; m_o |= OPTIMAL_BIT; }
  if (targetB - sB + 1 > MAX_ALIGNMENT_LENGTH) { targetB = sB + MAX_ALIGNMENT_LENGTH - 1; if (!overflow) m_o |= OPTIMAL_BIT; overflow = true; }
  int32_t err = 0;
  const int32_t tA = targetA, tB = targetB;
  bool reached = eng.align(sA, targetA, sB, targetB, m_o, err);
  if (aln >= 0) eng.piece(PIECE_SEARCH, aln, sA, sB, targetA, targetB, tA, tB, m_o);
  if (reached && overflow) reached = false;
  return PnFwd{targetA, targetB, err, targetk, reached ? 1 : 0};
}

template <typename ENG, typename BOUNDS, typename SYNC>
PG_HD int postnuc_unit(ENG& eng, const Chain* chains, const Match* cm, const int32_t* order, int n, BOUNDS&& bounds, uint8_t* fused,
                       PnAln* al, int max_al, SYNC& sync, int strand) {
  for (int k = 0; k < n; ++k) fused[k] = 0;
  int n_al = 0, cura = -1;
  // The current alignment lives in A (registers) and is written to al[cura] only when something is about to read al[] or the
  // walk moves on to another alignment: with every field update a read-modify-write of global memory, a unit of 50 000 matches
  // spent ~50 us per match waiting on its own stores (the longest unit of a launch took 2.4 s with all its DP work done ahead).
  PnAln A{0, 0, 0, 0, 0, 0};
  bool target_reached = false, full = false;
  int prev = 0, curk = 0, targetk = -1;
  int32_t targetA = 0, targetB = 0;
  int32_t turn_rrec = 0, turn_qrec = 0;
  while (curk < n) {
    const int c = order[curk];
    const Chain C = chains[c];
    const Match* mm = cm + C.first;
    const Match mf = mm[0];
    const Match ml = mm[C.count - 1];
    int32_t r_lo, r_hi, q_lo, q_hi;
    bounds(c, r_lo, r_hi, q_lo, q_hi);
    if (!target_reached) {      // a new turn of the walk (see PnPairSync): the cluster `prev` stands on
      eng.piece(PIECE_VISIT, -1, mf.r, mf.q, 0, 0, 0, 0, 0u);
      sync.begin_turn(pn_turn_key(mf.r, strand));
      turn_rrec = C.rrec; turn_qrec = C.qrec;
      bool skip = fused[curk] != 0;
      if (!skip) {   // isShadowedCluster: inside an alignment of the same records and strand, from the synteny's CURRENT alignment backwards
        if (cura >= 0) al[cura] = A;
        const int first = eng.shadow_first(chains, al, n_al, c, mf.r, ml.r + ml.len - 1, mf.q, ml.q + ml.len - 1);
        if (first >= 0) { const int vis = sync.visible(C.rrec, C.qrec, n_al, cura); skip = first <= vis; if (skip != (first <= cura)) ++sync.differs; }
      }
      if (skip) { fused[curk] = 1; curk = ++prev; continue; }
    }
    for (int m = 0; m < C.count; ++m) {
      Match Mp = mm[m];
      if (target_reached) {
        if (A.eA != Mp.r || A.eB != Mp.q) continue;     // matches of the target cluster before the target match
        if constexpr (pn_has_gap_run<ENG>::value) {
          // the matches of a cluster whose match-to-match alignments are ready and all reached the next match: taken in one go (per
          // match the walk would add the gap's errors and step onto the next match — two dependent loads each, which is what a
          // unit's walk spent its time on: ~2 us per match, 23 us per cluster on C4)
          const int j = eng.gap_run(mm, C.first, m, C.count, A.errors);
          if (j != m) { m = j; Mp = mm[m]; A.eA = Mp.r; A.eB = Mp.q; }
        }
        A.eA += Mp.len - 1; A.eB += Mp.len - 1;
        eng.piece(PIECE_MATCH, cura, Mp.r, Mp.q, Mp.r + Mp.len - 1, Mp.q + Mp.len - 1, 0, 0, 0u);
      } else {
        if (n_al >= max_al) { full = true; break; }
        if (cura >= 0) al[cura] = A;      // (the one the walk leaves; the scan below reads the alignments made so far)
        A = PnAln{Mp.r, Mp.q, Mp.r + Mp.len - 1, Mp.q + Mp.len - 1, 0, c};
        cura = n_al++;
        sync.pushed(cura);
        // getReverseTargetAlignment: the latest earlier alignment that ends at or before this start in both sequences and is
        // close enough; failing that, the one at the smallest distance if that beats the distance to the sequence starts
        const int32_t d0 = (A.sA - r_lo + 1) < (A.sB - q_lo + 1) ? (A.sA - r_lo + 1) : (A.sB - q_lo + 1);
        const int tgt = eng.reverse_target(chains, al, cura, c, A.sA, A.sB, d0);
        // extendBackward: search back towards the target's end; reached = merge (the gap is re-aligned forwards, forced)
        {
          unsigned m_o = BACKWARD_SEARCH;
          int32_t tA, tB;
          bool overflow = false;
          PnAln T{0, 0, 0, 0, 0, 0};
          if (tgt >= 0) { T = al[tgt]; tA = T.eA; tB = T.eB; } else { tA = r_lo; tB = q_lo; m_o |= OPTIMAL_BIT; }
          if (A.sA - tA + 1 > MAX_ALIGNMENT_LENGTH) { tA = A.sA - MAX_ALIGNMENT_LENGTH + 1; overflow = true; m_o |= OPTIMAL_BIT; }
          if (A.sB - tB + 1 > MAX_ALIGNMENT_LENGTH) { tB = A.sB - MAX_ALIGNMENT_LENGTH + 1; if (!overflow) m_o |= OPTIMAL_BIT; overflow = true; }
          int32_t err = 0;
          bool reached;
          PnBwd bw;
          if (m == 0) eng.bwd_key(curk, A.sA, A.sB, tA, tB, m_o | SEARCH_BIT);      // (a rehearsal notes what it was asked)
          if (m == 0 && eng.bwd_ready(curk, bw) && bw.sA == A.sA && bw.sB == A.sB && bw.tA == tA && bw.tB == tB && bw.m_o == (m_o | SEARCH_BIT)) {
            tA = bw.rA; tB = bw.rB; reached = bw.reached != 0;      // searched ahead of the walk with exactly these arguments
          } else
            reached = eng.align(A.sA, tA, A.sB, tB, m_o | SEARCH_BIT, err);
          if (overflow || tgt < 0) reached = false;
          if (reached) {
            // forced re-alignments only contribute their error count (the corner is reached by definition): the engine may
            // return it now (scalar engines) or add it to the alignment later (the GPU defers them to a kernel of their own,
            // which runs after this walk has written al[tgt] for the last time)
            T.errors += eng.forced_errors(T.eA, A.sA, T.eB, A.sB, al + tgt);
            eng.piece(PIECE_FORCED, tgt, T.eA, T.eB, A.sA, A.sB, 0, 0, FORCED_FORWARD_ALIGN);
            T.eA = A.eA; T.eB = A.eB;
            --n_al;
            cura = tgt;
            A = T;
          } else {
            if (tA != A.sA || tB != A.sB) {
              A.errors += eng.forced_errors(tA, A.sA, tB, A.sB, al + cura);
              eng.piece(PIECE_FORCED, cura, tA, tB, A.sA, A.sB, 0, 0, FORCED_FORWARD_ALIGN);
            }
            A.sA = tA; A.sB = tB;
          }
        }
        eng.piece(PIECE_MATCH, cura, Mp.r, Mp.q, Mp.r + Mp.len - 1, Mp.q + Mp.len - 1, 0, 0, 0u);
      }
      // extendForward: to the next match of the cluster, or from its last match towards the target cluster
      if (m + 1 < C.count) {
        // match to match inside a cluster: the call depends on the two matches only (the alignment ends on this match's last
        // base), so an engine may have it ready (the GPU runs all of them in a pass of their own before the units)
        PnGap g;
        if (eng.gap_ready(C.first + m, g)) {
          A.errors += g.errors; A.eA = g.eA; A.eB = g.eB;
          target_reached = g.reached != 0;
          continue;
        }
        const Match Mn = mm[m + 1];
        targetA = Mn.r; targetB = Mn.q;
        bool overflow = false;
        unsigned m_o = FORWARD_ALIGN;
        if (targetA - A.eA + 1 > MAX_ALIGNMENT_LENGTH) { targetA = A.eA + MAX_ALIGNMENT_LENGTH - 1; overflow = true; m_o |= OPTIMAL_BIT; }
        if (targetB - A.eB + 1 > MAX_ALIGNMENT_LENGTH) { targetB = A.eB + MAX_ALIGNMENT_LENGTH - 1; if (!overflow) m_o |= OPTIMAL_BIT; overflow = true; }
        int32_t err = 0;
        const int32_t tA = targetA, tB = targetB;
        bool reached = eng.align(A.eA, targetA, A.eB, targetB, m_o, err);
        eng.piece(PIECE_SEARCH, cura, A.eA, A.eB, targetA, targetB, tA, tB, m_o);
        if (reached && overflow) reached = false;
        A.errors += err;
        A.eA = targetA; A.eB = targetB;
        target_reached = reached;
      } else {
        // off the last match (the alignment ends on its last base here, however the walk entered the cluster): PnFwd
        PnFwd f;
        if (!eng.fwd_ready(curk, f)) f = postnuc_forward(eng, chains, cm, order, n, curk, bounds, cura);
        targetk = f.targetk;
        A.errors += f.errors; A.eA = f.eA; A.eB = f.eB;
        target_reached = f.reached != 0;
      }
    }
    if (full) break;
    if (targetk < 0) target_reached = false;
    fused[curk] = 1;
    if (!target_reached) { sync.end_turn(turn_rrec, turn_qrec, cura); curk = ++prev; } else curk = targetk;
  }
  if (cura >= 0) al[cura] = A;
  sync.finish();
  return full ? -1 - n_al : n_al;
}

}  // namespace pgn
