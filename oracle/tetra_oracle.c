/* oracle/tetra_oracle.c — CPU restatement of pyani's TETRA path.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (pyani_amd/) never imports, links or calls it.
 *
 * It follows the reference literally (both strands are materialised and scanned exactly like the Python
 * loops), deliberately NOT using the closed form the GPU kernels use, so that the two are independent.
 *
 * Reference (read-only, /root/reference):
 *   pyani/tetra.py:78-139   calculate_tetra_zscore   -> orc_tetra_counts + orc_tetra_zscores
 *   pyani/tetra.py:143-153  tetra_clean              -> clean k-mer test (codes >= 0)
 *   pyani/tetra.py:158-194  calculate_correlations   -> orc_tetra_corr
 *
 * Pinned against the reference's own goldens (tests/golden/ref_tetra_zscore_NC_002696.json =
 * tests/fixtures/targets/tetra/zscore.json, tests/target_TETRA_output/TETRA_correlations.tab) and against
 * vectors produced by importing the reference itself (tools/make_goldens.py): see tests/test_oracle_tetra.py.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC   (no FMA contraction: the float op order is the contract)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* upper-case + code: A=0 C=1 G=2 T=3, anything else -1 (tetra.py:99 uppercases; tetra.py:143-153 defines clean) */
static int code_of(unsigned char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

/* k-mer index with the first base most significant == sorted() string order (tetra.py:176) */
static int kmer_index(const int8_t *s, int k) {
  int v = 0;
  for (int j = 0; j < k; ++j) {
    if (s[j] < 0) return -1;
    v = v * 4 + s[j];
  }
  return v;
}

/* One strand of one record, exactly the loop structure of tetra.py:102-116. */
static void count_strand(const int8_t *seq, int64_t L, uint64_t *c2, uint64_t *c3, uint64_t *c4) {
  int idx;
  /* for i in range(len(seq[:-4])): seq[:-4] has max(L-4,0) symbols */
  int64_t n = L - 4 > 0 ? L - 4 : 0;
  for (int64_t i = 0; i < n; ++i) {
    if ((idx = kmer_index(seq + i, 2)) >= 0) c2[idx]++;
    if ((idx = kmer_index(seq + i, 3)) >= 0) c3[idx]++;
    if ((idx = kmer_index(seq + i, 4)) >= 0) c4[idx]++;
  }
  /* stragglers (tetra.py:112-116).  Python slices clamp; a slice shorter than k lands under a key of the
   * wrong length that is never read, so only full-length slices count. */
  if (L >= 4) {
    if ((idx = kmer_index(seq + L - 4, 3)) >= 0) c3[idx]++; /* seq[-4:-1] */
    if ((idx = kmer_index(seq + L - 3, 3)) >= 0) c3[idx]++; /* seq[-3:]   */
    if ((idx = kmer_index(seq + L - 4, 2)) >= 0) c2[idx]++; /* seq[-4:-2] */
    if ((idx = kmer_index(seq + L - 3, 2)) >= 0) c2[idx]++; /* seq[-3:-1] */
    if ((idx = kmer_index(seq + L - 2, 2)) >= 0) c2[idx]++; /* seq[-2:]   */
  } else if (L == 3) {
    /* seq[-4:-1] = first 2 symbols (dinucleotide key in the TRI dict: never read); seq[-3:] = all 3 */
    if ((idx = kmer_index(seq, 3)) >= 0) c3[idx]++;
    /* seq[-4:-2] = first symbol (len 1: never read); seq[-3:-1] = first 2; seq[-2:] = last 2 */
    if ((idx = kmer_index(seq, 2)) >= 0) c2[idx]++;
    if ((idx = kmer_index(seq + 1, 2)) >= 0) c2[idx]++;
  } else if (L == 2) {
    /* tri slices have len <= 2: never read as trimers.  seq[-4:-2]='' ; seq[-3:-1]=first symbol; seq[-2:]=both */
    if ((idx = kmer_index(seq, 2)) >= 0) c2[idx]++;
  }
  /* L <= 1: nothing of length >= 2 */
}

/* Counts for one genome.  seq = concatenated records (ASCII), rec_off[0..n_rec] record boundaries.
 * c2[16], c3[64], c4[256] are ACCUMULATED into (caller zeroes).  Returns 0, or -1 on allocation failure. */
int orc_tetra_counts(const unsigned char *seq, const uint64_t *rec_off, uint32_t n_rec, uint64_t *c2, uint64_t *c3,
                     uint64_t *c4) {
  for (uint32_t r = 0; r < n_rec; ++r) {
    const int64_t L = (int64_t)(rec_off[r + 1] - rec_off[r]);
    int8_t *fwd = (int8_t *)malloc((size_t)(L > 0 ? L : 1));
    int8_t *rev = (int8_t *)malloc((size_t)(L > 0 ? L : 1));
    if (!fwd || !rev) { free(fwd); free(rev); return -1; }
    for (int64_t i = 0; i < L; ++i) fwd[i] = (int8_t)code_of(seq[rec_off[r] + i]);
    /* reverse complement (tetra.py:99, Biopython Seq.reverse_complement): IUPAC ambiguity symbols complement
     * to ambiguity symbols, unknown characters stay as they are; the one asymmetric case is U/u, which
     * Biopython's DNA complement table maps to A/a (so it is dirty forward but clean on the reverse strand). */
    for (int64_t i = 0; i < L; ++i) {
      const unsigned char ch = seq[rec_off[r] + (L - 1 - i)];
      if (ch == 'U' || ch == 'u') rev[i] = 0;
      else rev[i] = fwd[L - 1 - i] < 0 ? (int8_t)-1 : (int8_t)(3 - fwd[L - 1 - i]);
    }
    count_strand(fwd, L, c2, c3, c4);
    count_strand(rev, L, c2, c3, c4);
    free(fwd);
    free(rev);
  }
  return 0;
}

/* Z-scores for n genomes in the reference's operation order (tetra.py:119-138).
 * present[t] = 1 iff tetramer t was observed (it is a key of the reference's result dict). */
void orc_tetra_zscores(const uint64_t *c2, const uint64_t *c3, const uint64_t *c4, uint32_t n, double *z,
                       uint8_t *present) {
  for (uint32_t g = 0; g < n; ++g) {
    const uint64_t *g2 = c2 + 16 * (size_t)g, *g3 = c3 + 64 * (size_t)g, *g4 = c4 + 256 * (size_t)g;
    for (int t = 0; t < 256; ++t) {
      double *zo = z + 256 * (size_t)g + t;
      uint8_t *po = present + 256 * (size_t)g + t;
      if (g4[t] == 0) { *zo = 0.0; *po = 0; continue; }
      const uint64_t a = g3[t >> 2], b = g3[t & 63], den = g2[(t >> 2) & 15];
      /* 1.0 * c3[abc] * c3[bcd] / c2[bc]  — left to right */
      const double e = ((1.0 * (double)a) * (double)b) / (double)den;
      /* sqrt(exp * (den - a) * (den - b) / (den * den)); integer subexpressions are exact Python ints */
      const double sd = sqrt(((e * (double)(den - a)) * (double)(den - b)) / (double)(den * den));
      if (sd != 0.0) *zo = ((double)g4[t] - e) / sd;
      else *zo = 1.0 / (double)(den * den); /* ZeroDivisionError branch, tetra.py:135-138 */
      *po = 1;
    }
  }
}

/* Pearson matrix (tetra.py:170-193).  Returns 0, or -2 if two genomes have different key sets
 * (the reference raises AssertionError, tetra.py:174-175), -3 if the common key set is empty (the reference
 * raises ZeroDivisionError at tetra.py:181).  Diagonal = 1.0 literal (fillna(1.0), :171). */
int orc_tetra_corr(const double *z, const uint8_t *present, uint32_t n, double *out) {
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < n; ++j) out[(size_t)i * n + j] = 1.0;
  for (uint32_t i = 0; i + 1 < n; ++i) {
    for (uint32_t j = i + 1; j < n; ++j) {
      const double *z1 = z + 256 * (size_t)i, *z2 = z + 256 * (size_t)j;
      const uint8_t *p1 = present + 256 * (size_t)i, *p2 = present + 256 * (size_t)j;
      if (memcmp(p1, p2, 256) != 0) return -2;
      int cnt = 0;
      double s1 = 0.0, s2 = 0.0; /* Python sum() starts from int 0; 0 + x == x exactly (and -0.0 -> 0.0 like 0.0 + -0.0) */
      for (int t = 0; t < 256; ++t)
        if (p1[t]) { s1 = s1 + z1[t]; s2 = s2 + z2[t]; ++cnt; }
      if (cnt == 0) return -3; /* Python: sum([]) / 0 -> ZeroDivisionError (tetra.py:181) */
      const double m1 = s1 / (double)cnt, m2 = s2 / (double)cnt;
      double dp = 0.0, ss1 = 0.0, ss2 = 0.0;
      for (int t = 0; t < 256; ++t)
        if (p1[t]) {
          const double d1 = z1[t] - m1, d2 = z2[t] - m2;
          dp = dp + d1 * d2;
          ss1 = ss1 + d1 * d1;
          ss2 = ss2 + d2 * d2;
        }
      const double r = dp / sqrt(ss1 * ss2);
      out[(size_t)i * n + j] = r;
      out[(size_t)j * n + i] = r;
    }
  }
  return 0;
}
