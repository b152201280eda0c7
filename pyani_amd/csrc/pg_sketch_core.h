// pg_sketch_core.h — the definition of the SKETCH mode ("fastANI-style sketch mode"), plain C++ for the device
// (pg_sketch.hip) and for the host checker of the tests.
//
// What it stands in for: pyani's fastANI wrapper (pyani/fastani.py:193-270) shells out to `fastANI -q query -r ref --fragLen 3000
// -k 16 --minFraction 0.2` and reads back one line: query, reference, ANI estimate (percent), orthologous matches, query fragments
// (parse_fastani_file -> ComparisonResult(reference, query, ani, matches, fragments)).  fastANI (third-party, absent from the
// reference tree and the image) maps every non-overlapping fragLen piece of the query to the reference through MinHash sketches of
// its 16-mers and averages the pieces' identity estimates.  This mode computes an estimate of the same SHAPE — same inputs, same
// three outputs, same minFraction rule — with an estimator that needs no mapping step:
//
//   * k = 16: a 16-mer of the 2-bit alphabet IS a 32-bit integer; canonical form = min(forward, reverse complement); windows with
//     an ambiguity symbol or across a record end do not exist (the mask bit of the packed stream, as everywhere in this engine);
//   * FracMinHash sampling: a canonical k-mer belongs to every sketch iff mix32(kmer) & (scale - 1) == 0 (scale = 16 by default:
//     ~190 sampled k-mers per 3 000-base fragment, ~3 x 10^5 per 5 Mb genome);
//   * the query genome's records are cut into non-overlapping fragments of frag_len bases (a record's tail shorter than that is
//     dropped, as fastANI does); a sampled k-mer belongs to the fragment that contains all 16 of its bases;
//   * per fragment: n = its sampled k-mer occurrences, h = those that occur ANYWHERE in the reference genome (either strand);
//     containment C = h / n, and since a k-mer survives iff none of its k bases changed, identity = C^(1/16) (four square roots:
//     correctly rounded on host and device, so the estimate is reproducible bit for bit);
//   * a fragment MATCHES iff h >= 2 and identity >= 0.80 (fastANI's floor: below that its mapper finds nothing either);
//   * ANI = mean identity of the matching fragments (summed in fragment order), matches = how many, fragments = all of them;
//     fewer matches than min_fraction * fragments: no result (fastANI writes an empty file; parse_fastani_file raises).
//
// It is an ESTIMATE with its own columns, never written into the exact ANIm / ANIb matrices; tests/test_sketch_gpu.py holds the GPU
// against the numpy restatement of this definition (bit-exact) and prices the estimate against the exact engine on C3 pairs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PGS_HD __host__ __device__ __forceinline__
#else
#define PGS_HD inline
#endif

namespace pgs {

constexpr int K = 16;
constexpr double MIN_IDENTITY = 0.80;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;      // (no canonical 16-mer has this value: min(x, rc x) < 2^32 - 1)

PGS_HD uint32_t mix32(uint32_t h) {          // murmur3's finaliser: a bijection of the 32-bit k-mer
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// forward word: first base in the HIGH bits; rc word likewise for the reverse complement
PGS_HD uint32_t roll_fwd(uint32_t f, uint32_t code) { return (f << 2) | code; }
PGS_HD uint32_t roll_rc(uint32_t r, uint32_t code) { return (r >> 2) | ((3u - code) << 30); }
PGS_HD bool sampled(uint32_t canon, uint32_t scale) { return (mix32(canon) & (scale - 1u)) == 0u; }
PGS_HD uint32_t slot_of(uint32_t canon, uint32_t log2_scale, uint32_t cap_mask) { return (mix32(canon) >> log2_scale) & cap_mask; }
// identity estimate of a fragment with h of n sampled k-mers found: (h / n)^(1/16)
PGS_HD double frag_identity(uint32_t h, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
  double c = (double)h / (double)n;
  c = __dsqrt_rn(c); c = __dsqrt_rn(c); c = __dsqrt_rn(c); c = __dsqrt_rn(c);
  return c;
#else
  double c = (double)h / (double)n;
  c = __builtin_sqrt(c); c = __builtin_sqrt(c); c = __builtin_sqrt(c); c = __builtin_sqrt(c);
  return c;
#endif
}
PGS_HD bool frag_matches(uint32_t h, uint32_t n) { return n > 0u && h >= 2u && frag_identity(h, n) >= MIN_IDENTITY; }

}  // namespace pgs
