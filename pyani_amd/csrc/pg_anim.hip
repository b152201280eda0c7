// pg_anim.hip — gfx950 kernels of the ANIm engine (kernel family 3 of BASELINE.json's north star): replaces the
// `nucmer --mum` + `delta-filter -1` processes pyani shells out to (pyani/anim.py:240-289) and the parse_delta
// reduction (anim.py:292-411) with an in-process pipeline over the 2-bit/1-bit packed genomes already resident in HBM.
//
//   A1 anim_list_kernel      per genome, once: its 16-mers as (k-mer, position) lists partitioned into 2048 hash groups
//                            (every position for the reference role, every 5th strand position for the query role)
//   A2 anim_seed_kernel      one workgroup per (reference, group): the group's reference k-mers become a hash table in
//                            LDS; the same group of every query of that reference streams through it (coalesced,
//                            sequential HBM reads; no random global access except to verify / extend actual hits)
//   A3 anim_cluster_wave_kernel  one WAVE per (pair, strand): MUM filter (packed radix sorts + wave-scan containment
//                            flags), mgaps clustering (lock-free union-find), chain extraction (register / LDS resident)
//   A4 anim_gaps_kernel      one wave per chain: trims the chained matches, settles trivial gaps, emits GapTasks
//      the banded affine DP (pga::extend_banded) in two forms with identical results:
//      anim_gapdp_lane_kernel / anim_extdp_lane_kernel   one LANE per search, 64 per wave, the band in registers (small
//                            gaps row by row; extensions and larger gaps by anti-diagonals in persistent waves)
//      extend_wave           one WAVE per search (64 lanes = 64 diagonals, DPP neighbour exchange): takes over the
//                            searches the lane waves hand over mid-way when they run thin, and the rare later calls
//      anim_extreq_kernel    writes the DP calls of every chain down as requests for the lanes (policy code, no DP)
//      anim_extend_kernel    one wave per chain: forward extension towards the next chain, then backward extension /
//                            junction bridge, consuming the lanes' answers
//   A5 anim_finish_kernel    one wave per pair: stitch/fuse chains, 1-to-1 filter, parse_delta reduction -> pg_anim_result
//
// Every kernel has a scalar statement in pg_anim_core.h that compiles for the host (tools/anim_debug); the two are kept
// in lock-step and compared on the GPU by tests/test_anim_gpu.py.  Limits: genomes up to ~14 Mb (a reference k-mer
// group must fit a 16384-slot LDS table; PG_E_CAPACITY otherwise), chain scores < 2^24.
#include "pg_internal.h"
#include "pg_anim_core.h"

using namespace pga;

namespace {

constexpr unsigned long long SLOT_EMPTY = ~0ull;
constexpr int MAX_HITS = 4096;  // copies of one seed k-mer examined per lookup (a bound for pathological repeats only)

struct RefDesc {
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;  // n_rec + 1 entries
  int32_t n_rec;
};

struct UnitDesc {   // one (pair, query strand)
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;
  int32_t n_rec;
  int32_t strand;
  int32_t pair;  // index into the batch's pair list
  int32_t ref;   // index into the batch's reference list
};

// 16 bases starting at stream position p (p + 16 <= len): codes in 32 bits (first base low), clean bits in 16
__device__ __forceinline__ void get16(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t p,
                                      uint32_t& c, uint32_t& m) {
  const uint32_t cw = p >> 4, cs = 2 * (p & 15);
  const uint64_t lo = (uint64_t)codes[cw] | ((uint64_t)codes[cw + 1] << 32);
  c = (uint32_t)(lo >> cs);
  const uint32_t mw = p >> 5, ms = p & 31;
  const uint64_t ml = (uint64_t)mask[mw] | ((uint64_t)mask[mw + 1] << 32);
  m = (uint32_t)(ml >> ms) & 0xFFFFu;
}

// ---- A1/A2: seeding ------------------------------------------------------------------------------------------------
// Every maximal exact match of length >= MIN_MATCH (20) contains, whatever its offset, a query-strand position that is a
// multiple of SEED_STEP and starts a SEED_K-mer lying wholly inside the match (SEED_K + SEED_STEP - 1 == MIN_MATCH).  So
// the reference lists its 16-mers at EVERY position and the query strand is looked up at every 5th position only; of
// the sampled positions inside one match, the first (left extension < SEED_STEP) is the one that reports it.
constexpr int SEED_K = 16, SEED_STEP = 5;
static_assert(SEED_K + SEED_STEP - 1 == MIN_MATCH, "sampling must not miss a minimal-length match");

// 16-mer (first base in the low bits) of strand `strand` at strand position q (q + 16 <= len), false if a base is dirty
__device__ __forceinline__ bool seed_kmer(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t len,
                                          int32_t strand, int32_t q, uint32_t& k) {
  uint32_t c, m;
  if (strand == 0) {
    get16(codes, mask, q, c, m);
    k = c;
  } else {
    get16(codes, mask, len - SEED_K - q, c, m);  // forward window holding the same bases
    uint32_t x = __brev(c);
    x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
    k = ~x;
  }
  return m == 0xFFFFu;
}

// Hash of a seed k-mer: the top SEED_GROUP_BITS select the group (partition of the k-mer space shared by all genomes),
// bits 5.. select the slot inside the group's LDS table.
constexpr int SEED_GROUP_BITS = 11, SEED_GROUPS = 1 << SEED_GROUP_BITS;
constexpr uint32_t SEED_MAX_SLOTS = 16384;   // 128 KiB of LDS
__device__ __forceinline__ uint32_t seed_hash(uint32_t k) { return k * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t seed_group(uint32_t h) { return h >> (32 - SEED_GROUP_BITS); }

// Per-genome seed lists.  role 0 (reference): every stream position, 1 sub-list per group; role 1 (query): every
// SEED_STEP-th position of both strands, sub-list index = 2 * group + strand.
// Entry (64 bit): [63:43] low 21 bits of the k-mer hash (the hash is a bijection of the 32-bit k-mer and its top 11 bits
// are the group, so these 21 bits identify the k-mer within its group) | [42:33] the SEED_STEP bases to the LEFT of the
// k-mer, nearest first | [32] 1 = all of them exist and are clean | [31:0] position.  With both flags set, the
// left-maximality test of a hit needs no memory access at all.
// pass 0 counts into cnt[], pass 1 writes at goff[] + cursor (cnt[] re-zeroed in between by anim_list_scan_kernel).
constexpr uint64_t SEED_KEY_SHIFT = 43;
constexpr int LIST_BLOCK = 1024, LIST_CHUNK = 16384;   // positions (or sampled positions) per workgroup
__global__ __launch_bounds__(LIST_BLOCK) void anim_list_kernel(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask,
                                                               int32_t len, int role, uint32_t* __restrict__ cnt,
                                                               const uint32_t* __restrict__ goff, uint64_t* __restrict__ list, int pass) {
  // Sub-list counters are kept per workgroup in LDS; the global counters see one atomic per (workgroup, non-empty
  // sub-list) instead of one per k-mer.  pass 1 counts again, reserves a range per sub-list, then writes.
  __shared__ uint32_t s_cnt[2 * SEED_GROUPS];
  const int32_t strand = role ? (int32_t)blockIdx.y : 0;
  const uint32_t n_sub = role ? 2 * SEED_GROUPS : SEED_GROUPS;
  const int32_t idx0 = blockIdx.x * LIST_CHUNK;
  for (uint32_t i = threadIdx.x; i < n_sub; i += LIST_BLOCK) s_cnt[i] = 0;
  __syncthreads();
  auto kmer_of = [&](int32_t idx, uint32_t& h, int32_t& p) -> bool {
    p = role ? idx * SEED_STEP : idx;
    if (p + SEED_K > len) return false;
    uint32_t k;
    if (!seed_kmer(codes, mask, len, strand, p, k)) return false;
    h = seed_hash(k);
    return true;
  };
  auto sub_of = [&](uint32_t h) { const uint32_t g = seed_group(h); return role ? 2 * g + (uint32_t)strand : g; };
  for (int32_t t = threadIdx.x; t < LIST_CHUNK; t += LIST_BLOCK) {
    uint32_t h; int32_t p;
    if (kmer_of(idx0 + t, h, p)) atomicAdd(&s_cnt[sub_of(h)], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_sub; i += LIST_BLOCK) {
    const uint32_t c = s_cnt[i];
    uint32_t base = 0;
    if (c) base = atomicAdd(&cnt[i], c);
    s_cnt[i] = pass ? goff[i] + base : 0;   // pass 1: this workgroup's write cursor in sub-list i
  }
  if (!pass) return;
  __syncthreads();
  const StrandView V{SeqView{codes, mask, len}, strand};
  for (int32_t t = threadIdx.x; t < LIST_CHUNK; t += LIST_BLOCK) {
    uint32_t h; int32_t p;
    if (!kmer_of(idx0 + t, h, p)) continue;
    const uint32_t at = atomicAdd(&s_cnt[sub_of(h)], 1u);
    uint64_t left = 0, flag = 1;
    for (int j = 1; j <= SEED_STEP; ++j) {
      if (!V.clean(p - j)) { flag = 0; left = 0; break; }
      left |= (uint64_t)V.base(p - j) << (2 * (j - 1));
    }
    list[at] = ((uint64_t)(h & 0x1FFFFFu) << SEED_KEY_SHIFT) | (left << 33) | (flag << 32) | (uint32_t)p;
  }
}

// goff[0..n] = exclusive prefix of cnt[0..n), goff[n + 1] = max(cnt); cnt re-zeroed.  One wave; n is 2048 or 4096.
__global__ __launch_bounds__(64) void anim_list_scan_kernel(uint32_t* __restrict__ cnt, uint32_t* __restrict__ goff, uint32_t n) {
  const uint32_t lane = threadIdx.x;
  uint32_t run = 0, mx = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t c = cnt[base + lane];
    cnt[base + lane] = 0;
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
    goff[base + lane] = run + incl - c;
    run += __shfl(incl, 63, 64);
    mx = c > mx ? c : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
  if (lane == 0) { goff[n] = run; goff[n + 1] = mx; }
}

struct SeedRef {            // one per reference of the batch
  const uint64_t* list;
  const uint32_t* goff;     // SEED_GROUPS + 2
  uint32_t pair_begin, pair_end;
};
struct SeedQry {            // one per pair of the batch (its query genome)
  const uint64_t* list;
  const uint32_t* goff;     // 2 * SEED_GROUPS + 2
};
// Per batch, transposed: slice[g * n_pairs + p] = where pair p's query keeps group g.  A wave reads the descriptors of
// 64 of its pairs with ONE coalesced load instead of chasing pair -> offset table -> entries once per pair.
struct SeedSlice { uint32_t begin, n0, n1; };   // strand-0 entries [begin, begin + n0), strand-1 [begin + n0, begin + n0 + n1)
__global__ __launch_bounds__(256) void anim_slice_kernel(const SeedQry* __restrict__ sqry, uint32_t n_pairs, SeedSlice* __restrict__ slice) {
  const uint32_t p = blockIdx.x;
  const uint32_t* goff = sqry[p].goff;
  for (uint32_t g = threadIdx.x; g < SEED_GROUPS; g += 256) {
    const uint32_t o0 = goff[2 * g], o1 = goff[2 * g + 1], o2 = goff[2 * g + 2];
    slice[(size_t)g * n_pairs + p] = SeedSlice{o0, o1 - o0, o2 - o1};
  }
}

// One hit of a sampled query k-mer (strand position q) on reference position r: report the maximal match it lies in,
// unless an earlier sampled position of the same match does.  Returns false if nothing is to be appended.
__device__ __forceinline__ bool seed_hit(const RefDesc& R, const SeqView& RV, const UnitDesc& U0, const StrandView& QV, int strand,
                                         int32_t r, int32_t q, int32_t left, Match& out) {
  if (left < 0) {   // left context not decidable from the list entries (sequence start / ambiguity symbol nearby)
    left = 0;
    while (left < SEED_STEP && RV.clean(r - 1 - left) && QV.clean(q - 1 - left) && RV.base(r - 1 - left) == QV.base(q - 1 - left)) ++left;
  }
  if (left == SEED_STEP) return false;
  int32_t L = SEED_K;
  // right extension, 16 bases per step (word compare of the packed codes and masks), then base by base near a sequence
  // end.  Single-exit loops (state in `n`): break / continue shapes cost a lot of exec-mask bookkeeping.
  int n = 16;
  while (n == 16 && r + L + 16 <= R.len && q + L + 16 <= U0.len) {
    uint32_t rc_, rm_, qc_, qm_;
    get16(R.codes, R.mask, r + L, rc_, rm_);
    uint32_t fc, fm;
    get16(U0.codes, U0.mask, strand == 0 ? q + L : U0.len - 16 - (q + L), fc, fm);
    uint32_t rv = __brev(fc);
    rv = ((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1);
    qc_ = strand == 0 ? fc : ~rv;
    qm_ = strand == 0 ? fm : __brev(fm) >> 16;
    const uint32_t x = rc_ ^ qc_;
    const uint32_t diff = (x | (x >> 1)) & 0x55555555u;
    const uint32_t bad = ~(rm_ & qm_) & 0xFFFFu;
    const int nd = diff ? (__ffs(diff) - 1) >> 1 : 16;
    const int nb = bad ? __ffs(bad) - 1 : 16;
    n = nd < nb ? nd : nb;
    L += n;
  }
  if (n == 16)   // fewer than 16 bases left in one of the sequences
    while (RV.clean(r + L) && QV.clean(q + L) && RV.base(r + L) == QV.base(q + L)) ++L;
  if (left + L < MIN_MATCH) return false;
  out = Match{r - left, q - left, left + L, 0};
  return true;
}

// Workgroup (g, r): LDS table of reference r's group g, then every query of r streams its group-g entries through it.
// Each of the 16 WAVES takes every 16th pair and keeps SEED_UNROLL coalesced 512-byte loads in flight, so the stream is
// bandwidth- rather than latency-bound.  Matches are appended to one batch-wide buffer (the `strand` field carries the
// unit index until the scatter); unit_count[] is exact even when the buffer overflows, which is what the host uses to
// size the slices (and to re-run a prefix).
constexpr int SEED_BLOCK = 1024, SEED_UNROLL = 4;
constexpr uint32_t SEED_STAGE = 48;    // hits staged in LDS per wave (64 KiB table + staging: two workgroups per CU)
constexpr size_t SEED_STAGE_BYTES = (SEED_BLOCK / 64) * (SEED_STAGE * sizeof(Match) + 4);
__global__ __launch_bounds__(SEED_BLOCK) void anim_seed_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                               const SeedRef* __restrict__ srefs, const SeedQry* __restrict__ sqry,
                                                               const SeedSlice* __restrict__ slice, uint32_t n_pairs,
                                                               uint32_t slot_mask, Match* __restrict__ buf, uint32_t cap,
                                                               uint32_t* __restrict__ total, uint32_t* __restrict__ hit_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
  Match* stage = reinterpret_cast<Match*>(tab + slot_mask + 1);                    // [waves][SEED_STAGE]
  uint32_t* stage_n = reinterpret_cast<uint32_t*>(stage + (SEED_BLOCK / 64) * SEED_STAGE);   // [waves]
  const uint32_t g = blockIdx.x;
  const SeedRef SR = srefs[blockIdx.y];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  for (uint32_t i = tid; i <= slot_mask; i += SEED_BLOCK) tab[i] = SLOT_EMPTY;
  if (tid < SEED_BLOCK / 64) stage_n[tid] = 0;
  __syncthreads();
  for (uint32_t e = SR.goff[g] + tid; e < SR.goff[g + 1]; e += SEED_BLOCK) {
    const unsigned long long v = SR.list[e];
    uint32_t slot = (uint32_t)(v >> (SEED_KEY_SHIFT + 5)) & slot_mask;   // hash bits 5.. (slot_mask <= 2^14 - 1)
    while (atomicCAS(&tab[slot], SLOT_EMPTY, v) != SLOT_EMPTY) slot = (slot + 1) & slot_mask;
  }
  __syncthreads();
  // this wave's share of the reference's pairs: a contiguous range, its descriptors fetched 64 at a time
  const uint32_t n_mine_all = SR.pair_end - SR.pair_begin;
  const uint32_t per_wave = (n_mine_all + SEED_BLOCK / 64 - 1) / (SEED_BLOCK / 64);
  const uint32_t my_begin = SR.pair_begin + wave * per_wave;
  const uint32_t my_end = my_begin + per_wave < SR.pair_end ? my_begin + per_wave : SR.pair_end;
  const SeedSlice* __restrict__ row = slice + (size_t)g * n_pairs;
  for (uint32_t chunk = my_begin; chunk < my_end; chunk += 64) {
    SeedSlice mine{0, 0, 0};
    const uint64_t* mylist = nullptr;
    if (chunk + lane < my_end) { mine = row[chunk + lane]; mylist = sqry[chunk + lane].list; }
    const uint32_t in_chunk = my_end - chunk < 64 ? my_end - chunk : 64;
    // The chunk's work as a sequence of row blocks (<= SEED_UNROLL rows of 64 entries of one (pair, strand) slice),
    // software-pipelined: the loads of block k+1 are in flight while block k is looked up.
    struct Blk { uint32_t j, strand, e0, e_end; const uint64_t* list; bool valid; };
    auto slice_of = [&](uint32_t j, uint32_t strand, Blk& o) {
      const uint32_t begin = (uint32_t)__builtin_amdgcn_readlane((int)mine.begin, j);
      const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.n0, j), n1 = (uint32_t)__builtin_amdgcn_readlane((int)mine.n1, j);
      o.j = j; o.strand = strand;
      o.e0 = strand ? begin + n0 : begin;
      o.e_end = o.e0 + (strand ? n1 : n0);
      o.list = reinterpret_cast<const uint64_t*>(
          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)((unsigned long long)mylist >> 32), j) << 32) |
          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(unsigned long long)mylist, j));
    };
    auto first_from = [&](uint32_t j, uint32_t strand) {   // first non-empty slice at or after (j, strand)
      Blk o{0, 0, 0, 0, nullptr, false};
      for (; j < in_chunk; ++j, strand = 0)
        for (; strand < 2; ++strand) {
          slice_of(j, strand, o);
          if (o.e0 < o.e_end) { o.valid = true; return o; }
        }
      return o;
    };
    auto next_of = [&](const Blk& c) {
      if (c.e0 + 64 * SEED_UNROLL < c.e_end) { Blk o = c; o.e0 += 64 * SEED_UNROLL; return o; }
      return c.strand == 0 ? first_from(c.j, 1) : first_from(c.j + 1, 0);
    };
    auto load = [&](const Blk& c, unsigned long long (&qv)[SEED_UNROLL]) {
#pragma unroll
      for (int t = 0; t < SEED_UNROLL; ++t) {
        const uint32_t e = c.e0 + 64 * t + lane;
        qv[t] = e < c.e_end ? __builtin_nontemporal_load(&c.list[e]) : SLOT_EMPTY;
      }
    };
    auto process = [&](const Blk& c, const unsigned long long (&qv)[SEED_UNROLL], bool last) {
      const uint32_t unit = 2 * (chunk + c.j) + c.strand;
#pragma unroll
      for (int t = 0; t < SEED_UNROLL; ++t) {
        if (qv[t] == SLOT_EMPTY) continue;
        const uint32_t key = (uint32_t)(qv[t] >> SEED_KEY_SHIFT);
        const uint32_t qctx = (uint32_t)(qv[t] >> 32) & 0x7FFu;   // bit 0: flag, bits 1..10: left bases
        const int32_t q = (int32_t)(uint32_t)qv[t];
        uint32_t slot = (key >> 5) & slot_mask;
        // one exit condition and no break / continue inside: the compiler turns anything else into a state machine of
        // exec-mask bookkeeping, and the per-CU scalar unit is a bottleneck of this kernel
        int hits = 0;
        unsigned long long v = tab[slot];
        while (v != SLOT_EMPTY) {   // load factor <= 1/2: every probe sequence ends
          if ((uint32_t)(v >> SEED_KEY_SHIFT) == key) {
            int32_t left = -1;
            bool report = true;
            const uint32_t rctx = (uint32_t)(v >> 32) & 0x7FFu;
            if (rctx & qctx & 1u) {
              const uint32_t x = (rctx ^ qctx) >> 1;
              const uint32_t diff = (x | (x >> 1)) & 0x155u;
              left = diff ? (__ffs(diff) - 1) >> 1 : SEED_STEP;
              report = left != SEED_STEP;   // inside a longer match: an earlier sampled position reports it
            }
            if (report) {
              // a hit that may start a match: handed to anim_hit_kernel (verification / extension need the sequences and
              // many registers; keeping them out of this kernel doubles its waves per SIMD).  Staged per wave in LDS.
              const Match m{(int32_t)(uint32_t)v, q, left, (int32_t)unit};
              const uint32_t at = atomicAdd(&stage_n[wave], 1u);
              if (at < SEED_STAGE) {
                stage[wave * SEED_STAGE + at] = m;
              } else {   // staging buffer full (a burst of hits): straight to the global buffer
                const uint32_t ga = atomicAdd(total, 1u);
                atomicAdd(&hit_count[unit], 1u);
                if (ga < cap) buf[ga] = m;
              }
            }
            ++hits;
          }
          slot = (slot + 1) & slot_mask;
          v = hits < MAX_HITS ? tab[slot] : SLOT_EMPTY;
        }
      }
      // uniform point: flush once the buffer is half full (or at the very end).  Staged hits sit in processing order,
      // i.e. in runs of one unit: every run adds its length to its unit's hit count (one atomic per run, not per hit).
      __builtin_amdgcn_wave_barrier();
      uint32_t n_st = stage_n[wave];
      if (n_st > SEED_STAGE) n_st = SEED_STAGE;
      if (n_st >= SEED_STAGE / 2 || (n_st && last)) {
        static_assert(SEED_STAGE <= 64, "one staged hit per lane at flush time");
        uint32_t base = 0;
        if (lane == 0) {
          base = atomicAdd(total, n_st);
          stage_n[wave] = 0;
        }
        base = __shfl(base, 0);
        Match m{0, 0, 0, -1};
        if (lane < n_st) m = stage[wave * SEED_STAGE + lane];
        const int32_t prev_unit = __shfl_up(m.strand, 1, 64);
        const bool start = lane < n_st && (lane == 0 || prev_unit != m.strand);
        const uint64_t starts = __ballot(start);
        if (start) {
          const uint64_t later = starts >> 1 >> lane;   // starts after this lane
          const uint32_t run = later ? (uint32_t)__ffsll((unsigned long long)later) : n_st - lane;
          atomicAdd(&hit_count[(uint32_t)m.strand], run);
        }
        if (lane < n_st && base + lane < cap) buf[base + lane] = m;
        __builtin_amdgcn_wave_barrier();
      }
    };
    unsigned long long qa[SEED_UNROLL], qb[SEED_UNROLL];
    Blk A = first_from(0, 0);
    if (A.valid) load(A, qa);
    while (A.valid) {
      Blk B = next_of(A);
      if (B.valid) load(B, qb);
      process(A, qa, !B.valid && chunk + 64 >= my_end);
      if (!B.valid) break;
      A = next_of(B);
      if (A.valid) load(A, qa);
      process(B, qb, !A.valid && chunk + 64 >= my_end);
    }
  }
}

// hoff[0..n] = exclusive prefix of cnt[0..n); cursor[] zeroed.  One workgroup (n <= 2 * pairs of a launch).
__global__ __launch_bounds__(1024) void anim_hoff_kernel(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ hoff,
                                                         uint32_t* __restrict__ cursor) {
  __shared__ uint32_t s_part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n + 1023) / 1024, lo = tid * per, hi = lo + per < n ? lo + per : n;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += cnt[i];
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) { uint32_t run = 0; for (int i = 0; i < 1024; ++i) { const uint32_t t = s_part[i]; s_part[i] = run; run += t; } hoff[n] = run; }
  __syncthreads();
  uint32_t run = s_part[tid];
  for (uint32_t i = lo; i < hi; ++i) { hoff[i] = run; run += cnt[i]; cursor[i] = 0; }
}

// hits -> per-unit slices (hoff): same dealing as anim_scatter_kernel, records unchanged
__global__ __launch_bounds__(256) void anim_hit_scatter_kernel(const Match* __restrict__ buf, const uint32_t* __restrict__ n_hits, uint32_t cap,
                                                               const uint32_t* __restrict__ hoff, uint32_t* __restrict__ cursor,
                                                               Match* __restrict__ out) {
  const uint32_t n = *n_hits < cap ? *n_hits : cap;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    Match m{0, 0, 0, 0};
    if (i < n) m = buf[i];
    const uint32_t u = (uint32_t)m.strand;
    bool todo = i < n;
    while (true) {
      const uint64_t rest = __ballot(todo);
      if (!rest) break;
      const int leader = __ffsll((unsigned long long)rest) - 1;
      const uint32_t lu = __shfl(u, leader);
      const uint64_t same = __ballot(todo && u == lu);
      uint32_t at = 0;
      if ((int)lane == leader) at = atomicAdd(&cursor[lu], (uint32_t)__popcll(same));
      at = __shfl(at, leader);
      if (todo && u == lu) {
        out[(size_t)hoff[u] + at + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = m;
        todo = false;
      }
    }
  }
}

// One WORKGROUP per unit walks that unit's recorded hits {r, q, left (-1: undecided), unit}: decide the left extension
// where the list entries could not, extend to the right, and append matches of at least MIN_MATCH bases to the batch
// buffer (the `strand` field carries the unit until the scatter) while counting them — exact even if the buffer
// overflows.  A unit's hits touch only its own two genomes (≈ 4 MB packed): processed by one workgroup, i.e. on one XCD,
// they are served by that XCD's L2 instead of one HBM line fetch per access.
__global__ __launch_bounds__(256) void anim_hit_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                       const Match* __restrict__ hits, const uint32_t* __restrict__ hoff,
                                                       const uint32_t* __restrict__ n_hits, uint32_t hit_cap,
                                                       Match* __restrict__ buf, uint32_t cap, uint32_t* __restrict__ total,
                                                       uint32_t* __restrict__ unit_count) {
  const uint32_t unit = blockIdx.x;
  const uint32_t h0 = hoff[unit], h1 = hoff[unit + 1];
  if (h0 == h1 || *n_hits > hit_cap) return;   // (hits were dropped: the slices are incomplete, the host retries with fewer pairs)
  const uint32_t lane = threadIdx.x & 63u;
  const UnitDesc U0 = units[unit];
  const RefDesc R = refs[U0.ref];
  const SeqView RV{R.codes, R.mask, R.len};
  const StrandView QV{SeqView{U0.codes, U0.mask, U0.len}, U0.strand};
  for (uint32_t base = h0; base < h1; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    Match m{0, 0, 0, (int32_t)unit};
    bool have = false;
    if (i < h1) {
      const Match h = hits[i];
      have = seed_hit(R, RV, U0, QV, U0.strand, h.r, h.q, h.len, m);
    }
    const uint64_t got = __ballot(have);
    if (got) {
      uint32_t at = 0;
      if (lane == 0) {
        const uint32_t c = (uint32_t)__popcll(got);
        at = atomicAdd(total, c);
        atomicAdd(&unit_count[unit], c);
      }
      at = __shfl(at, 0);
      if (have) {
        at += (uint32_t)__popcll(got & ((1ull << lane) - 1ull));
        m.strand = (int32_t)unit;
        if (at < cap) buf[at] = m;
      }
    }
  }
}

// Deal the appended matches into their units' slices (moff[u] .. moff[u+1]); units >= n_units wait for the next batch.
// The seed kernel flushes bursts of one unit, so a wave usually sees one to three distinct units: one atomic per
// distinct unit and wave instead of one per match.
__global__ __launch_bounds__(256) void anim_scatter_kernel(const Match* __restrict__ buf, uint32_t n, const uint32_t* __restrict__ moff,
                                                           uint32_t n_units, uint32_t* __restrict__ cursor, Match* __restrict__ mem) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  Match m{0, 0, 0, -1};
  if (i < n) m = buf[i];
  const uint32_t u = (uint32_t)m.strand;
  bool todo = i < n && u < n_units;
  while (true) {
    const uint64_t rest = __ballot(todo);
    if (!rest) break;
    const int leader = __ffsll((unsigned long long)rest) - 1;
    const uint32_t lu = __shfl(u, leader);
    const uint64_t same = __ballot(todo && u == lu);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&cursor[lu], (uint32_t)__popcll(same));
    base = __shfl(base, leader);
    if (todo && u == lu) {
      m.strand = (int32_t)(u & 1u);
      mem[(size_t)moff[u] + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = m;
      todo = false;
    }
  }
}

struct ClusterOut {   // per-match arrays are sliced by moff[] (a chain has >= 1 match, so chains fit the same slices)
  const uint32_t* moff; // [U + 1]
  Match* cm;
  Chain* chains;
  int32_t* n_chains;    // [U]
  int32_t* order;       // chains sorted by first-match ref start
  int32_t* prev_of;
  int32_t* next_of;
  int32_t* status;      // [P]
};

__global__ __launch_bounds__(64) void anim_cluster_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units, uint32_t n_units,
                                                          Match* __restrict__ mem, const uint32_t* __restrict__ mem_count,
                                                          int32_t* __restrict__ iscratch, ClusterOut O) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  O.n_chains[u] = 0;
  const size_t off = O.moff[u];
  const uint32_t cap = O.moff[u + 1] - O.moff[u];
  uint32_t n0 = mem_count[u];
  if (n0 > cap) { atomicOr(&O.status[U.pair], 1); n0 = cap; }
  if (n0 == 0) return;
  Match* m = mem + off;
  const int n = mum_filter(m, (int)n0, U.strand);
  int32_t* s = iscratch + off * 7;
  int32_t *rrec = s, *qrec = s + cap, *parent = s + 2 * (size_t)cap, *score = s + 3 * (size_t)cap, *from = s + 4 * (size_t)cap,
          *adj = s + 5 * (size_t)cap, *order = s + 6 * (size_t)cap;
  for (int i = 0; i < n; ++i) {
    rrec[i] = record_of(R.rec_start, R.n_rec, m[i].r);
    const int32_t qf = U.strand ? U.len - 1 - m[i].q : m[i].q;
    qrec[i] = record_of(U.rec_start, U.n_rec, qf);
  }
  int n_chains = 0, n_cm = 0;
  Chain* chains = O.chains + off;
  Match* cm = O.cm + off;
  mgaps_strand(m, n, U.strand, rrec, qrec, parent, score, from, adj, order, chains, n_chains, (int)cap, cm, n_cm, (int)cap);
  int32_t* co = O.order + off;
  for (int i = 0; i < n_chains; ++i) co[i] = i;
  heapsort(co, n_chains, [&](int a, int b) { return cm[chains[a].first].r < cm[chains[b].first].r; });
  chain_neighbours(chains, co, n_chains, O.prev_of + off, O.next_of + off);
  O.n_chains[u] = n_chains;
}

__device__ __forceinline__ int32_t from_lane_below(int32_t v, int32_t fill) {  // lane l <- lane l-1 (lane 0 <- fill)
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
// the same shifts for unsigned DP keys whose "nothing there" value is 0: bound_ctrl delivers it, no fill register needed
__device__ __forceinline__ uint32_t dpp_from_above0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dpp_from_below0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ long long wave_max64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// Wave-wide max / min of a 32-bit value through DPP (no LDS crossbar: a handful of cycles instead of six dependent
// ds_bpermute round trips).  Every lane gets the result.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define PG_DPP_MAX(ctrl, rmask) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v = t_ > v ? t_ : v; }
  PG_DPP_MAX(0xB1, 0xf)    // quad_perm [1,0,3,2]
  PG_DPP_MAX(0x4E, 0xf)    // quad_perm [2,3,0,1]
  PG_DPP_MAX(0x141, 0xf)   // row_half_mirror
  PG_DPP_MAX(0x140, 0xf)   // row_mirror: every lane of a row holds the row's max
  PG_DPP_MAX(0x142, 0xa)   // row_bcast15 into rows 1 and 3
  PG_DPP_MAX(0x143, 0xc)   // row_bcast31 into rows 2 and 3: lane 63 holds the wave's max
#undef PG_DPP_MAX
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }

// =====================================================================================================================
// A3, wave-cooperative: one WAVE per (pair, strand) unit.  Same results as the scalar statement (pga::mum_filter +
// pga::mgaps_strand, which the one-thread kernel above runs): stable LSD radix sorts instead of heapsorts, wave scans
// for the containment flags, a lock-free union-find, and a chain DP whose 64-deep look-back lives in the 64 lanes.
// =====================================================================================================================
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// Orders this wave's LDS accesses (cross-lane read-after-write through LDS) without waiting for its outstanding global
// stores, which a __syncthreads() of a one-wave workgroup would do (s_waitcnt vmcnt(0): ~1-2 us per use).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Stable LSD radix sort of packed (key << 32 | val) words by `passes` 8-bit digits of the key.  Returns the buffer that
// holds the result (a or b).  One 8-byte scattered store per element and pass; a pass whose digit is the same for
// every element moves nothing; global loads are issued four 64-element rows ahead of their use.  hist: 256 words of LDS.
__device__ uint64_t* wave_radix_sort(uint64_t* a, uint64_t* b, int n, int passes, uint32_t* hist) {
  const int lane = threadIdx.x & 63;
  constexpr int U = 4;
  uint64_t *src = a, *dst = b;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 32 + 8 * pass;
    __syncthreads();   // the previous pass's (or the caller's) global stores have landed
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    wave_lds_fence();
    for (int base = 0; base < n; base += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int t = 0; t < U; ++t) { const int i = base + 64 * t + lane; kk[t] = i < n ? src[i] : 0ull; }
#pragma unroll
      for (int t = 0; t < U; ++t) if (base + 64 * t + lane < n) atomicAdd(&hist[(uint32_t)(kk[t] >> shift) & 255u], 1u);
    }
    wave_lds_fence();
    uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
    if (__any(c0 == (uint32_t)n || c1 == (uint32_t)n || c2 == (uint32_t)n || c3 == (uint32_t)n)) continue;   // constant digit
    {  // exclusive scan of the 256 bins: 4 bins per lane
      const uint32_t tot = c0 + c1 + c2 + c3;
      uint32_t incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      uint32_t ex = incl - tot;
      wave_lds_fence();
      hist[4 * lane] = ex; ex += c0;
      hist[4 * lane + 1] = ex; ex += c1;
      hist[4 * lane + 2] = ex; ex += c2;
      hist[4 * lane + 3] = ex;
    }
    wave_lds_fence();
    for (int base = 0; base < n; base += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int t = 0; t < U; ++t) { const int i = base + 64 * t + lane; kk[t] = i < n ? src[i] : 0ull; }
#pragma unroll
      for (int t = 0; t < U; ++t) {   // rows in order: the sort is stable
        const bool act = base + 64 * t + lane < n;
        const uint32_t d = (uint32_t)(kk[t] >> shift) & 255u;
        uint64_t peers = __ballot(act);
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
          const uint64_t vote = __ballot((d >> bb) & 1u);
          peers &= ((d >> bb) & 1u) ? vote : ~vote;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
        uint32_t pos = 0;
        if (act) pos = hist[d] + rank;
        wave_lds_fence();   // every lane has read its bin before the bin's first lane advances it
        if (act && rank == 0) hist[d] += (uint32_t)__popcll(peers);
        wave_lds_fence();
        if (act) dst[pos] = kk[t];
      }
    }
    uint64_t* tmp = src; src = dst; dst = tmp;
  }
  __syncthreads();
  return src;
}

// containment flags over elements in sorted order (ascending start, ties: longer first): flag[idx] |= 1 if an earlier
// element reaches at least as far, or if the next element has the same start and length.
__device__ void wave_containment_flags(const uint64_t* order, const int32_t* start, const int32_t* len, int n, int32_t* flag) {
  const int lane = threadIdx.x & 63;
  int32_t carry = -1;
  for (int base = 0; base < n; base += 64) {
    const int t = base + lane;
    const bool act = t < n;
    const uint32_t idx = act ? (uint32_t)order[t] : 0u;
    const int32_t st = act ? start[idx] : 0, ln = act ? len[idx] : 0;
    const int32_t e = act ? st + ln : -1;
    int32_t incl = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o, 64); if (lane >= o && v > incl) incl = v; }
    int32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -1;
    const int32_t prevmax = excl > carry ? excl : carry;
    if (act) {
      bool f = e <= prevmax;
      if (!f && t + 1 < n) { const uint32_t nx = (uint32_t)order[t + 1]; f = start[nx] == st && len[nx] == ln; }
      if (f) flag[idx] = 1;
    }
    const int32_t last = __shfl(incl, 63, 64);
    if (last > carry) carry = last;
  }
}

// (single-exit loops throughout: on this hardware a divergent loop with break / continue / return inside compiles to a
// state machine of exec-mask bookkeeping that costs more than the work)
__device__ __forceinline__ int uf_find(int32_t* parent, int x) {
  // with path halving: links only ever move to an ancestor (a smaller index), so the races are benign
  int p = parent[x];
  while (p != x) {
    const int gp = parent[p];
    if (gp != p) parent[x] = gp;
    x = gp;          // == p when p is the root: the loop then ends
    p = parent[x];
  }
  return x;
}
__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b) {   // larger root -> smaller root (deterministic roots)
  bool done = false;
  while (!done) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    done = a == b || atomicCAS(&parent[hi], hi, lo) == hi;
  }
}

// =====================================================================================================================
// A3a, workgroup-cooperative: the front half of the per-unit work (MUM filter, union-find, grouping by cluster) run by
// PREP_WAVES waves per unit, so that the largest unit of a launch no longer sets the launch time on one wave's latency
// chain.  Same algorithms as the wave versions above (which remain as the single-wave statement): the stable counting
// sort gives every wave a contiguous chunk and its own histogram row, offsets are the prefix over (digit, wave).
// =====================================================================================================================
constexpr int PREP_WAVES = 16, PREP_THREADS = PREP_WAVES * 64;

__device__ uint64_t* block_radix_sort(uint64_t* a, uint64_t* b, int n, int passes, uint32_t* hist /*[PREP_WAVES][256]*/,
                                      uint32_t* s_misc /*[4]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int U = 4;
  const int chunk = ((n + PREP_WAVES - 1) / PREP_WAVES + 63) & ~63;   // whole rows per wave
  const int c0 = wave * chunk < n ? wave * chunk : n, c1 = c0 + chunk < n ? c0 + chunk : n;
  uint64_t *src = a, *dst = b;
  uint32_t* myh = hist + wave * 256;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 32 + 8 * pass;
    __syncthreads();   // previous pass's (or the caller's) global stores have landed; hist free
    for (int i = tid; i < PREP_WAVES * 256; i += PREP_THREADS) hist[i] = 0;
    if (tid == 0) s_misc[0] = 0;
    __syncthreads();
    for (int base = c0; base < c1; base += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int t = 0; t < U; ++t) { const int i = base + 64 * t + lane; kk[t] = i < c1 ? src[i] : 0ull; }
#pragma unroll
      for (int t = 0; t < U; ++t) if (base + 64 * t + lane < c1) atomicAdd(&myh[(uint32_t)(kk[t] >> shift) & 255u], 1u);
    }
    __syncthreads();
    // digit totals, constant-digit test, exclusive prefix over (digit, wave): thread d owns digit d
    uint32_t col = 0;
    if (tid < 256) {
      for (int w = 0; w < PREP_WAVES; ++w) col += hist[w * 256 + tid];
      if (col == (uint32_t)n) s_misc[0] = 1;
    }
    __syncthreads();
    if (s_misc[0]) continue;   // every key has the same digit: nothing moves (uniform decision)
    if (wave == 0) {           // exclusive scan of the 256 totals by wave 0 (4 per lane), result into s_tot via hist row reuse
      uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      for (int w = 0; w < PREP_WAVES; ++w) {
        t0 += hist[w * 256 + 4 * lane]; t1 += hist[w * 256 + 4 * lane + 1];
        t2 += hist[w * 256 + 4 * lane + 2]; t3 += hist[w * 256 + 4 * lane + 3];
      }
      const uint32_t tot = t0 + t1 + t2 + t3;
      uint32_t incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      uint32_t ex = incl - tot;
      // turn every column into running offsets: hist[w][d] = base[d] + sum_{w' < w} count[w'][d]
      uint32_t basev[4] = {ex, ex + t0, ex + t0 + t1, ex + t0 + t1 + t2};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t run = basev[k];
        for (int w = 0; w < PREP_WAVES; ++w) { const uint32_t c = hist[w * 256 + 4 * lane + k]; hist[w * 256 + 4 * lane + k] = run; run += c; }
      }
    }
    __syncthreads();
    for (int base = c0; base < c1; base += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int t = 0; t < U; ++t) { const int i = base + 64 * t + lane; kk[t] = i < c1 ? src[i] : 0ull; }
#pragma unroll
      for (int t = 0; t < U; ++t) {   // rows in order: the sort is stable
        const bool act = base + 64 * t + lane < c1;
        const uint32_t d = (uint32_t)(kk[t] >> shift) & 255u;
        uint64_t peers = __ballot(act);
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
          const uint64_t vote = __ballot((d >> bb) & 1u);
          peers &= ((d >> bb) & 1u) ? vote : ~vote;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
        uint32_t pos = 0;
        if (act) pos = myh[d] + rank;
        wave_lds_fence();
        if (act && rank == 0) myh[d] += (uint32_t)__popcll(peers);
        wave_lds_fence();
        if (act) dst[pos] = kk[t];
      }
    }
    uint64_t* tmp = src; src = dst; dst = tmp;
  }
  __syncthreads();
  return src;
}

// containment flags (see wave_containment_flags): every wave scans its chunk of the sorted order; the running maximum
// entering a chunk is the maximum of the earlier chunks' ends.
__device__ void block_containment_flags(const uint64_t* order, const int32_t* start, const int32_t* len, int n, int32_t* flag,
                                        int32_t* s_carry /*[PREP_WAVES]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = ((n + PREP_WAVES - 1) / PREP_WAVES + 63) & ~63;
  const int c0 = wave * chunk < n ? wave * chunk : n, c1 = c0 + chunk < n ? c0 + chunk : n;
  int32_t mx = -1;
  for (int t = c0 + lane; t < c1; t += 64) { const uint32_t idx = (uint32_t)order[t]; const int32_t e = start[idx] + len[idx]; mx = e > mx ? e : mx; }
  mx = (int32_t)wave_max_u32((uint32_t)(mx + 1)) - 1;
  if (lane == 0) s_carry[wave] = mx;
  __syncthreads();
  int32_t carry = -1;
  for (int w = 0; w < wave; ++w) carry = s_carry[w] > carry ? s_carry[w] : carry;
  for (int base = c0; base < c1; base += 64) {
    const int t = base + lane;
    const bool act = t < c1;
    const uint32_t idx = act ? (uint32_t)order[t] : 0u;
    const int32_t st = act ? start[idx] : 0, ln = act ? len[idx] : 0;
    const int32_t e = act ? st + ln : -1;
    int32_t incl = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o, 64); if (lane >= o && v > incl) incl = v; }
    int32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -1;
    const int32_t prevmax = excl > carry ? excl : carry;
    if (act) {
      bool f = e <= prevmax;
      if (!f && t + 1 < n) { const uint32_t nx = (uint32_t)order[t + 1]; f = start[nx] == st && len[nx] == ln; }
      if (f) flag[idx] = 1;
    }
    const int32_t last = __shfl(incl, 63, 64);
    if (last > carry) carry = last;
  }
  __syncthreads();
}

__global__ __launch_bounds__(PREP_THREADS) void anim_cluster_prep_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                                         Match* __restrict__ mem, const uint32_t* __restrict__ mem_count,
                                                                         int32_t* __restrict__ iscratch, ClusterOut O, int maxmatch) {
  __shared__ uint32_t hist[PREP_WAVES * 256];
  __shared__ uint32_t s_misc[4];
  __shared__ int32_t s_carry[PREP_WAVES];
  __shared__ uint32_t s_cnt[PREP_WAVES];
  const uint32_t u = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const size_t off = O.moff[u];
  const uint32_t cap = O.moff[u + 1] - O.moff[u];
  uint32_t n0 = mem_count[u];
  if (n0 > cap) { if (tid == 0) atomicOr(&O.status[U.pair], 1); n0 = cap; }
  if (n0 == 0) { if (tid == 0) O.n_chains[u] = 0; return; }
  Match* m = mem + off;
  int32_t* sa = iscratch + off * 8;
  int32_t *sb = sa + cap, *sc = sa + 2 * (size_t)cap, *sd = sa + 3 * (size_t)cap, *se = sa + 4 * (size_t)cap,
          *sf = sa + 5 * (size_t)cap, *sg = sa + 6 * (size_t)cap, *sh = sa + 7 * (size_t)cap;
  uint64_t *P0 = reinterpret_cast<uint64_t*>(sa), *P1 = reinterpret_cast<uint64_t*>(sc);
  const int n_in = (int)n0;
  // ---- MUM filter (as in the wave kernel) ---------------------------------------------------------------------------
  for (int i = tid; i < n_in; i += PREP_THREADS) { const Match t = m[i]; se[i] = t.r; sf[i] = t.q; sg[i] = t.len; sh[i] = 0; }
  __syncthreads();
  const int start_passes = (R.len > U.len ? R.len : U.len) < (1 << 24) ? 3 : 4;
  const uint64_t* qsorted = nullptr;
  for (int side = maxmatch ? 0 : 1; side >= 0; --side) {
    const int32_t* start = side == 0 ? sf : se;
    uint64_t* r0 = P0;
    if (maxmatch) {   // --maxmatch keeps every maximal match: no uniqueness filter, but a total order (q, len desc, r)
      for (int i = tid; i < n_in; i += PREP_THREADS) P0[i] = ((uint64_t)(uint32_t)se[i] << 32) | (uint32_t)i;
      r0 = block_radix_sort(P0, P1, n_in, start_passes, hist, s_misc);
      for (int i = tid; i < n_in; i += PREP_THREADS) {
        const uint32_t idx = (uint32_t)r0[i], l = (uint32_t)sg[idx];
        r0[i] = ((uint64_t)(0xFFFFFFu - (l > 0xFFFFFFu ? 0xFFFFFFu : l)) << 32) | idx;
      }
    } else {
      for (int i = tid; i < n_in; i += PREP_THREADS) {
        const uint32_t l = (uint32_t)sg[i];
        P0[i] = ((uint64_t)(0xFFFFFFu - (l > 0xFFFFFFu ? 0xFFFFFFu : l)) << 32) | (uint32_t)i;
      }
    }
    uint64_t* r1 = block_radix_sort(r0, r0 == P0 ? P1 : P0, n_in, 3, hist, s_misc);
    for (int i = tid; i < n_in; i += PREP_THREADS) { const uint32_t idx = (uint32_t)r1[i]; r1[i] = ((uint64_t)(uint32_t)start[idx] << 32) | idx; }
    uint64_t* r2 = block_radix_sort(r1, r1 == P0 ? P1 : P0, n_in, start_passes, hist, s_misc);
    if (!maxmatch) block_containment_flags(r2, start, sg, n_in, sh, s_carry);
    qsorted = r2;
  }
  // survivors in q order: ordered compaction over the waves' chunks
  int n = 0;
  {
    uint32_t* keepidx = reinterpret_cast<uint32_t*>(qsorted == P0 ? P1 : P0);
    const int chunk = ((n_in + PREP_WAVES - 1) / PREP_WAVES + 63) & ~63;
    const int c0 = wave * chunk < n_in ? wave * chunk : n_in, c1 = c0 + chunk < n_in ? c0 + chunk : n_in;
    uint32_t mine = 0;
    for (int base = c0; base < c1; base += 64) {
      const int t = base + lane;
      const bool keep = t < c1 && sh[(uint32_t)qsorted[t]] == 0;
      mine += (uint32_t)__popcll(__ballot(keep));
    }
    if (lane == 0) s_cnt[wave] = mine;
    __syncthreads();
    uint32_t at = 0;
    for (int w = 0; w < PREP_WAVES; ++w) { if (w < wave) at += s_cnt[w]; n += (int)s_cnt[w]; }
    for (int base = c0; base < c1; base += 64) {
      const int t = base + lane;
      const uint32_t idx = t < c1 ? (uint32_t)qsorted[t] : 0u;
      const bool keep = t < c1 && sh[idx] == 0;
      const uint64_t bm = __ballot(keep);
      if (keep) keepidx[at + __popcll(bm & lanemask_lt())] = idx;
      at += (uint32_t)__popcll(bm);
    }
    __syncthreads();
    for (int i = tid; i < n; i += PREP_THREADS) { const uint32_t idx = keepidx[i]; m[i] = Match{se[idx], sf[idx], sg[idx], U.strand}; }
    __syncthreads();
  }
  // ---- clustering (mgaps): union-find over all threads, then grouping by root -------------------------------------------
  int32_t *rrec = sa, *qrec = sb, *parent = sc, *order = sh;
  uint64_t *Q0 = reinterpret_cast<uint64_t*>(sd), *Q1 = reinterpret_cast<uint64_t*>(sf);
  for (int i = tid; i < n; i += PREP_THREADS) {
    rrec[i] = record_of(R.rec_start, R.n_rec, m[i].r);
    const int32_t qf = U.strand ? U.len - 1 - m[i].q : m[i].q;
    qrec[i] = record_of(U.rec_start, U.n_rec, qf);
    parent[i] = i;
  }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < n; i += PREP_THREADS) {
    const Match mi = m[i];
    const int32_t iend = mi.q + mi.len, idiag = mi.q - mi.r;
    int j = i + 1;
    int32_t sep = j < n ? m[j].q - iend : MAX_GAP + 1;
    while (sep <= MAX_GAP) {
      const Match mj = m[j];
      if (rrec[i] == rrec[j] && qrec[i] == qrec[j]) {
        int32_t dd = (mj.q - mj.r) - idiag;
        if (dd < 0) dd = -dd;
        int32_t lim = (int32_t)(DIAG_FACTOR * sep);
        if (lim < DIAG_DIFF) lim = DIAG_DIFF;
        if (dd <= lim) uf_union(parent, i, j);
      }
      ++j;
      sep = j < n ? m[j].q - iend : MAX_GAP + 1;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < n; i += PREP_THREADS) Q0[i] = ((uint64_t)(uint32_t)uf_find(parent, i) << 32) | (uint32_t)i;
  __syncthreads();
  const uint64_t* rs_ = block_radix_sort(Q0, Q1, n, 4, hist, s_misc);
  for (int i = tid; i < n; i += PREP_THREADS) { const uint64_t x = rs_[i]; parent[i] = (int32_t)(x >> 32); order[i] = (int32_t)(uint32_t)x; }
  if (tid == 0) O.n_chains[u] = n;   // handed to the chain kernel (which overwrites it with the chain count)
}

#ifdef PGA_DP_STATS
__device__ unsigned long long g_cl_stats[16];   // per phase: sum of cycles [0..5], max [6..11], max n_in [12]
#define CL_MARK(ph) do { const unsigned long long t_now = __builtin_readcyclecounter(); if (lane == 0) { \
    atomicAdd(&g_cl_stats[ph], t_now - t_mark); atomicMax(&g_cl_stats[6 + ph], t_now - t_mark); } t_mark = t_now; } while (0)
#else
#define CL_MARK(ph) do {} while (0)
#endif
__global__ __launch_bounds__(64) void anim_cluster_wave_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                               Match* __restrict__ mem, const uint32_t* __restrict__ mem_count,
                                                               int32_t* __restrict__ iscratch, ClusterOut O, int prepared, int maxmatch) {
  __shared__ uint32_t hist[256];
  constexpr int WALK_CHUNK = 1024;   // 4 KiB: keeps 32 one-wave workgroups per CU
  __shared__ int32_t s_from[WALK_CHUNK];
  const uint32_t u = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const size_t off = O.moff[u];
  const uint32_t cap = O.moff[u + 1] - O.moff[u];
  uint32_t n0 = mem_count[u];
  if (n0 > cap) { if (lane == 0) atomicOr(&O.status[U.pair], 1); n0 = cap; }
  if (n0 == 0) { if (lane == 0) O.n_chains[u] = 0; return; }
  Match* m = mem + off;
  Match* cm = O.cm + off;
  // scratch slices (8 x cap ints): a..h; (a,b) and (c,d) double as the two packed sort buffers
  int32_t* sa = iscratch + off * 8;
  int32_t *sb = sa + cap, *sc = sa + 2 * (size_t)cap, *sd = sa + 3 * (size_t)cap, *se = sa + 4 * (size_t)cap,
          *sf = sa + 5 * (size_t)cap, *sg = sa + 6 * (size_t)cap, *sh = sa + 7 * (size_t)cap;
  uint64_t *P0 = reinterpret_cast<uint64_t*>(sa), *P1 = reinterpret_cast<uint64_t*>(sc);
  const int n_in = (int)n0;
#ifdef PGA_DP_STATS
  unsigned long long t_mark = __builtin_readcyclecounter();
  if (lane == 0) atomicMax(&g_cl_stats[12], (unsigned long long)n_in);
#endif
  int n = 0;
  int32_t *rrec = sa, *qrec = sb, *parent = sc, *from = se, *adj = sf, *order = sh;
  uint64_t *Q0 = reinterpret_cast<uint64_t*>(sd), *Q1 = reinterpret_cast<uint64_t*>(sf);   // (sd,se) and (sf,sg): sort buffers
  if (prepared) {   // anim_cluster_prep_kernel has done the MUM filter, the union-find and the grouping
    n = O.n_chains[u];
  } else {
  // ---- MUM filter -----------------------------------------------------------------------------------------------
  // SoA copies: se = r, sf = q, sg = len; flags in sh
  for (int i = lane; i < n_in; i += 64) { const Match t = m[i]; se[i] = t.r; sf[i] = t.q; sg[i] = t.len; sh[i] = 0; }
  __syncthreads();
  const int start_passes = (R.len > U.len ? R.len : U.len) < (1 << 24) ? 3 : 4;
  const uint64_t* qsorted = nullptr;
  for (int side = maxmatch ? 0 : 1; side >= 0; --side) {
    const int32_t* start = side == 0 ? sf : se;   // reference intervals first, then query intervals (whose order is reused)
    // sort by (start asc, len desc): LSD = len-desc key first, then start
    uint64_t* r0 = P0;
    if (maxmatch) {   // --maxmatch keeps every maximal match: no uniqueness filter, but a total order (q, len desc, r)
      for (int i = lane; i < n_in; i += 64) P0[i] = ((uint64_t)(uint32_t)se[i] << 32) | (uint32_t)i;
      r0 = wave_radix_sort(P0, P1, n_in, start_passes, hist);
      for (int i = lane; i < n_in; i += 64) {
        const uint32_t idx = (uint32_t)r0[i], l = (uint32_t)sg[idx];
        r0[i] = ((uint64_t)(0xFFFFFFu - (l > 0xFFFFFFu ? 0xFFFFFFu : l)) << 32) | idx;
      }
    } else {
      for (int i = lane; i < n_in; i += 64) {
        const uint32_t l = (uint32_t)sg[i];
        P0[i] = ((uint64_t)(0xFFFFFFu - (l > 0xFFFFFFu ? 0xFFFFFFu : l)) << 32) | (uint32_t)i;
      }
    }
    uint64_t* r1 = wave_radix_sort(r0, r0 == P0 ? P1 : P0, n_in, 3, hist);
    for (int i = lane; i < n_in; i += 64) { const uint32_t idx = (uint32_t)r1[i]; r1[i] = ((uint64_t)(uint32_t)start[idx] << 32) | idx; }
    uint64_t* r2 = wave_radix_sort(r1, r1 == P0 ? P1 : P0, n_in, start_passes, hist);
    if (!maxmatch) wave_containment_flags(r2, start, sg, n_in, sh);
    __syncthreads();
    qsorted = r2;
  }
  // survivors in q order (distinct q among survivors): the query-side order is still there -> compact
  {
    uint32_t* keepidx = reinterpret_cast<uint32_t*>(qsorted == P0 ? P1 : P0);   // the other sort buffer is free
    for (int base = 0; base < n_in; base += 64) {
      const int t = base + lane;
      const uint32_t idx = t < n_in ? (uint32_t)qsorted[t] : 0u;
      const bool keep = t < n_in && sh[idx] == 0;
      const uint64_t bm = __ballot(keep);
      if (keep) keepidx[n + __popcll(bm & lanemask_lt())] = idx;
      n += (int)__popcll(bm);
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const uint32_t idx = keepidx[i]; m[i] = Match{se[idx], sf[idx], sg[idx], U.strand}; }
    __syncthreads();
  }
  CL_MARK(0);
  // ---- clustering (mgaps) ------------------------------------------------------------------------------------------
  for (int i = lane; i < n; i += 64) {
    rrec[i] = record_of(R.rec_start, R.n_rec, m[i].r);
    const int32_t qf = U.strand ? U.len - 1 - m[i].q : m[i].q;
    qrec[i] = record_of(U.rec_start, U.n_rec, qf);
    parent[i] = i;
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const Match mi = m[i];
    const int32_t iend = mi.q + mi.len, idiag = mi.q - mi.r;
    int j = i + 1;
    int32_t sep = j < n ? m[j].q - iend : MAX_GAP + 1;
    while (sep <= MAX_GAP) {
      const Match mj = m[j];
      if (rrec[i] == rrec[j] && qrec[i] == qrec[j]) {
        int32_t dd = (mj.q - mj.r) - idiag;
        if (dd < 0) dd = -dd;
        int32_t lim = (int32_t)(DIAG_FACTOR * sep);
        if (lim < DIAG_DIFF) lim = DIAG_DIFF;
        if (dd <= lim) uf_union(parent, i, j);
      }
      ++j;
      sep = j < n ? m[j].q - iend : MAX_GAP + 1;
    }
  }
  __threadfence_block();
  __syncthreads();
  CL_MARK(1);
  {  // group by root (stable: q order inside a cluster): radix sort of (root, index)
    for (int i = lane; i < n; i += 64) Q0[i] = ((uint64_t)(uint32_t)uf_find(parent, i) << 32) | (uint32_t)i;
    __syncthreads();
    const uint64_t* rs_ = wave_radix_sort(Q0, Q1, n, 4, hist);
    for (int i = lane; i < n; i += 64) { const uint64_t x = rs_[i]; parent[i] = (int32_t)(x >> 32); order[i] = (int32_t)(uint32_t)x; }
    __syncthreads();   // parent[] now = root id of the i-th element in grouped order
  }
  }   // !prepared
  CL_MARK(2);
  // ---- chain extraction per cluster ----------------------------------------------------------------------------------
  // grouped list: order[t] = match index, parent[t] = its root.  score/from/adj are indexed by LIST POSITION here.
  // `lst` (compacted working list of the current cluster) lives in adj's slice after use... keep it simple: a cluster's
  // live entries are kept contiguous in order[g0 .. g0+live).
  int n_chains = 0, n_cm = 0;
  Chain* chains = O.chains + off;
  int g0 = 0;
  while (g0 < n) {
    // ---- fast path: clusters of at most 64 matches (nearly all of them) are handled in registers -----------------------
    // One window = the next 64 list positions, lane t <-> position g0 + t: two rounds of global latency per window
    // instead of a dozen per cluster.  A cluster that reaches the window's end is retried at the start of the next
    // window; only one that starts at lane 0 and still does not end takes the general path below.
    {
      const int posn = g0 + lane;
      const bool valid = posn < n;
      const int32_t root = valid ? parent[posn] : -1;
      const int32_t idx = valid ? order[posn] : 0;
      Match mt{0, 0, 0, 0};
      int32_t rr = 0, qq = 0;
      if (valid) { mt = m[idx]; rr = rrec[idx]; qq = qrec[idx]; }
      int c0 = 0;
      bool general = false;
      while (c0 < 64 && g0 + c0 < n) {
        const int32_t r0 = __builtin_amdgcn_readlane(root, c0);
        const uint64_t diff = __ballot(root != r0) & ~((1ull << c0) - 1ull);
        const int c1 = diff ? __ffsll((long long)diff) - 1 : 64;
        if (c1 == 64 && g0 + 64 < n) { general = c0 == 0; break; }
        uint64_t L = (c1 == 64 ? ~0ull : (1ull << c1) - 1ull) & ~((1ull << c0) - 1ull);   // live members
        while (L) {
          int32_t my_sc = NEG_INF, my_from = -1, my_adj = 0, my_tot = 0, my_cnt = 0;
          int32_t best_sc = NEG_INF, best_k = -1;
          uint64_t done = 0;
          for (uint64_t rem = L; rem; rem &= rem - 1) {
            const int k = __ffsll((long long)rem) - 1;
            const int32_t mr = __builtin_amdgcn_readlane(mt.r, k), mq = __builtin_amdgcn_readlane(mt.q, k), ml = __builtin_amdgcn_readlane(mt.len, k);
            int32_t cand = NEG_INF, ol = 0;
            if ((done >> lane) & 1ull) {
              ol = mt.r + mt.len - mr;
              if (ol < 0) ol = 0;
              const int32_t ol2 = mt.q + mt.len - mq;
              if (ol2 > ol) ol = ol2;
              int32_t dd = (mq - mr) - (mt.q - mt.r);
              if (dd < 0) dd = -dd;
              cand = my_sc + ml - (ol + dd);
            }
            // best candidate: max cand, ties -> nearest predecessor (largest lane)
            // (chain scores are sums of match lengths: < 2^24; invalid candidates become key 0)
            const uint32_t key = wave_max_u32(cand > NEG_INF / 2 ? ((uint32_t)(cand + (1 << 24)) << 6) | (uint32_t)lane : 0u);
            const int32_t bc = key ? (int32_t)(key >> 6) - (1 << 24) : NEG_INF;
            const int bl = (int)(key & 63u);
            int32_t sc_k = ml, fr_k = -1, ad_k = 0, tot_k = ml, cnt_k = 1;
            if (bc > sc_k) {
              sc_k = bc; fr_k = bl; ad_k = __builtin_amdgcn_readlane(ol, bl);
              tot_k += __builtin_amdgcn_readlane(my_tot, bl); cnt_k += __builtin_amdgcn_readlane(my_cnt, bl);
            }
            if (lane == k) { my_sc = sc_k; my_from = fr_k; my_adj = ad_k; my_tot = tot_k; my_cnt = cnt_k; }
            if (sc_k > best_sc) { best_sc = sc_k; best_k = k; }
            done |= 1ull << k;
          }
          const int32_t total = __builtin_amdgcn_readlane(my_tot, best_k), cnt = __builtin_amdgcn_readlane(my_cnt, best_k);
          const bool emit = total >= MIN_CLUSTER && n_chains < (int)cap && n_cm + cnt <= (int)cap;
          uint64_t M = 0;
          int32_t my_pos = -1;
          {
            int kk = best_k, pos = n_cm + cnt;
            while (kk >= 0) {
              M |= 1ull << kk;
              --pos;
              if (lane == kk) my_pos = pos;
              kk = __builtin_amdgcn_readlane(my_from, kk);
            }
          }
          if (emit) {
            if (lane == best_k) {
              Chain c;
              c.first = n_cm; c.count = cnt; c.strand = U.strand; c.rrec = rr; c.qrec = qq;
              chains[n_chains] = c;
            }
            if ((M >> lane) & 1ull) {
              Match t = mt;
              t.r += my_adj; t.q += my_adj; t.len -= my_adj;
              cm[my_pos] = t;
            }
            n_chains += 1; n_cm += cnt;
          }
          L &= ~M;
        }
        c0 = c1;
      }
      if (!general) { g0 += c0; continue; }
    }
    // ---- general path: a cluster of more than 64 matches ----------------------------------------------------------------
#ifdef PGA_DP_STATS
    const unsigned long long t_gen = __builtin_readcyclecounter();
#endif
    int g1 = g0 + 1;
    {  // cluster end: first position whose root differs (wave search)
      const int32_t root = parent[g0];
      for (;;) {
        const int t = g1 + lane;
        const uint64_t diff = __ballot(t >= n || parent[t] != root);
        if (diff) { g1 += __ffsll((long long)diff) - 1; break; }
        g1 += 64;
      }
    }
    int live = g1 - g0;
    while (live > 0) {
#ifdef PGA_DP_STATS
      if (lane == 0) { atomicAdd(&g_cl_stats[13], 1ull); atomicAdd(&g_cl_stats[14], (unsigned long long)live); }
#endif
      // Chain DP over the live entries order[g0 .. g0+live), 64 at a time: the entries are gathered lane-parallel (one
      // round of global latency per 64), then handed out one by one through readlane.  Lane l's window registers hold
      // the entry at position k-1-l: start, length, best score, and the matched bases / members of the best chain
      // ending there, so the winner's totals are known without walking it.
      int32_t wr = 0, wq = 0, wl = 0, wsc = NEG_INF, wtot = 0, wcnt = 0;
      int32_t best_sc = NEG_INF, best_k = -1, best_tot = 0, best_cnt = 0;
      for (int kb = 0; kb < live; kb += 64) {
        const int kt = kb + lane;
        Match mt{0, 0, 0, 0};
        if (kt < live) mt = m[order[g0 + kt]];
        int32_t my_from = -1, my_adj = 0;
        const int kend = live - kb < 64 ? live - kb : 64;
        for (int t = 0; t < kend; ++t) {
          const int k = kb + t;
          const int32_t mr = __builtin_amdgcn_readlane(mt.r, t), mq = __builtin_amdgcn_readlane(mt.q, t),
                        ml = __builtin_amdgcn_readlane(mt.len, t);
          int32_t cand = NEG_INF, ol = 0;
          if (lane < k && wsc > NEG_INF / 2) {
            ol = wr + wl - mr;
            if (ol < 0) ol = 0;
            const int32_t ol2 = wq + wl - mq;
            if (ol2 > ol) ol = ol2;
            int32_t dd = (mq - mr) - (wq - wr);
            if (dd < 0) dd = -dd;
            cand = wsc + ml - (ol + dd);
          }
          // best candidate: max cand, ties -> nearest predecessor (smallest lane)
          const uint32_t key = wave_max_u32(cand > NEG_INF / 2 ? ((uint32_t)(cand + (1 << 24)) << 6) | (uint32_t)(63 - lane) : 0u);
          const int32_t bc = key ? (int32_t)(key >> 6) - (1 << 24) : NEG_INF;
          const int bl = 63 - (int)(key & 63u);
          int32_t sc_k = ml, fr_k = -1, ad_k = 0, tot_k = ml, cnt_k = 1;
          if (bc > sc_k) {
            sc_k = bc; fr_k = k - 1 - bl; ad_k = __builtin_amdgcn_readlane(ol, bl);
            tot_k += __builtin_amdgcn_readlane(wtot, bl); cnt_k += __builtin_amdgcn_readlane(wcnt, bl);
          }
          if (lane == t) { my_from = fr_k; my_adj = ad_k; }
          if (sc_k > best_sc) { best_sc = sc_k; best_k = k; best_tot = tot_k; best_cnt = cnt_k; }
          // slide the window: lane l <- lane l-1, lane 0 <- entry k
          wr = from_lane_below(wr, mr); wq = from_lane_below(wq, mq); wl = from_lane_below(wl, ml);
          wsc = from_lane_below(wsc, sc_k); wtot = from_lane_below(wtot, tot_k); wcnt = from_lane_below(wcnt, cnt_k);
        }
        if (kt < live) { from[g0 + kt] = my_from; adj[g0 + kt] = my_adj; }
      }
      __threadfence_block();
      __syncthreads();
      const int32_t total = best_tot, cnt = best_cnt;
      const bool emit = total >= MIN_CLUSTER && n_chains < (int)cap && n_cm + cnt <= (int)cap;
      if (emit && lane == 0) {
        const int first_idx = order[g0 + best_k];
        Chain c;
        c.first = n_cm; c.count = cnt; c.strand = U.strand; c.rrec = rrec[first_idx]; c.qrec = qrec[first_idx];
        chains[n_chains] = c;
      }
      // Walk the best chain backwards (from[k] < k always) and tag its members from[k] = -3 - (slot in cm), or -2 if
      // the chain is dropped.  The pointer chase runs in LDS: from[] is staged WALK_CHUNK entries at a time, high to low.
      {
        int k = best_k, pos = n_cm + cnt;
        for (int chunk = (best_k / WALK_CHUNK) * WALK_CHUNK; chunk >= 0 && k >= 0; chunk -= WALK_CHUNK) {
          const int hi = chunk + WALK_CHUNK < live ? chunk + WALK_CHUNK : live;
          for (int t = chunk + lane; t < hi; t += 64) s_from[t - chunk] = from[g0 + t];
          __syncthreads();
          if (lane == 0) {
            while (k >= chunk) {
              const int nx = s_from[k - chunk];
              from[g0 + k] = emit ? -3 - (--pos) : -2;
              k = nx;
            }
          }
          k = __shfl(k, 0, 64); pos = __shfl(pos, 0, 64);
          __syncthreads();
        }
      }
      __threadfence_block();
      __syncthreads();
      // gather the tagged members into cm (lane-parallel) and compact the live list (drop them), preserving order
      int kept = 0;
      for (int base = 0; base < live; base += 64) {
        const int k = base + lane;
        const int32_t f = k < live ? from[g0 + k] : 0;
        const bool keep = k < live && f > -2;
        const int32_t idx = k < live ? order[g0 + k] : 0;
        if (k < live && f <= -3) {
          Match t = m[idx];
          const int32_t a = adj[g0 + k];
          t.r += a; t.q += a; t.len -= a;
          cm[-3 - f] = t;
        }
        const uint64_t bmask = __ballot(keep);
        __syncthreads();
        if (keep) order[g0 + kept + __popcll(bmask & lanemask_lt())] = idx;
        kept += (int)__popcll(bmask);
        __syncthreads();
      }
      if (emit) { n_chains += 1; n_cm += cnt; }
      live = kept;
    }
    g0 = g1;
#ifdef PGA_DP_STATS
    if (lane == 0) atomicAdd(&g_cl_stats[15], __builtin_readcyclecounter() - t_gen);
#endif
  }
  CL_MARK(3);
  // ---- chains in reference order + neighbours ------------------------------------------------------------------------
  int32_t* co = O.order + off;
  {
    __syncthreads();
    for (int i = lane; i < n_chains; i += 64) Q0[i] = ((uint64_t)(uint32_t)cm[chains[i].first].r << 32) | (uint32_t)i;
    __syncthreads();
    const uint64_t* cs_ = wave_radix_sort(Q0, Q1, n_chains, 4, hist);
    for (int i = lane; i < n_chains; i += 64) co[i] = (int32_t)(uint32_t)cs_[i];
    __threadfence_block();
    __syncthreads();
  }
  int32_t* prev_of = O.prev_of + off;
  int32_t* next_of = O.next_of + off;
  for (int k = lane; k < n_chains; k += 64) {
    const int c = co[k];
    int p = -1, q = -1;
    for (int kk = k - 1; kk >= 0 && kk >= k - 8 && p < 0; --kk)
      if (chains[co[kk]].rrec == chains[c].rrec && chains[co[kk]].qrec == chains[c].qrec) p = co[kk];
    for (int kk = k + 1; kk < n_chains && kk <= k + 8 && q < 0; ++kk)
      if (chains[co[kk]].rrec == chains[c].rrec && chains[co[kk]].qrec == chains[c].qrec) q = co[kk];
    prev_of[c] = p;
    next_of[c] = q;
  }
  if (lane == 0) O.n_chains[u] = n_chains;
  CL_MARK(4);
}

__device__ __forceinline__ void chain_bounds(const RefDesc& R, const UnitDesc& U, const Chain& c, int32_t& r_lo, int32_t& r_hi,
                                             int32_t& q_lo, int32_t& q_hi) {
  r_lo = R.rec_start[c.rrec]; r_hi = R.rec_start[c.rrec + 1] - 1;
  q_lo = U.rec_start[c.qrec]; q_hi = U.rec_start[c.qrec + 1] - 1;
  if (U.strand) { const int32_t a = U.len - q_hi, b = U.len - q_lo; q_lo = a; q_hi = b; }
}

// ---- wave-cooperative banded DP: the 64 lanes of a wave ARE the 64 diagonals of the band ----------------------------
// Same cells, checks and tie-breaks as pga::extend_banded (pg_anim_core.h); neighbours' cells arrive through DPP
// wave shifts, so a step costs a handful of VALU ops per lane and no LDS.  All lanes return the same result.
// LDS staging of the two sequences for one wave: base codes (0-3, 4 = dirty / out of range) of consumed indices
// t = 0, 1, 2, ... in a 256-entry ring.  Cell (i, j) compares ring_r[i-1] with ring_q[j-1]; on anti-diagonal d every lane
// needs indices within [d/2 - 17, d/2 + 15], so the ring is topped up 64 entries at a time, one base per lane.
struct WaveSeq {
  uint8_t* ring_r;
  uint8_t* ring_q;
  int32_t loaded;  // indices [0, loaded) have been staged (uniform)
};

__device__ __forceinline__ void wave_seq_fill(WaveSeq& ws, const SeqView& R, const StrandView& Q, int64_t r0, int64_t q0, int dir,
                                              int32_t rmax, int32_t qmax, int lane) {
  const int32_t t = ws.loaded + lane;
  uint8_t rb = 4, qb = 5;   // dirty / out of range: two codes that never compare equal
  if (t < rmax) { const int64_t rp = dir > 0 ? r0 + t : r0 - 1 - t; if (R.clean(rp)) rb = (uint8_t)R.base(rp); }
  if (t < qmax) { const int64_t qp = dir > 0 ? q0 + t : q0 - 1 - t; if (Q.clean(qp)) qb = (uint8_t)Q.base(qp); }
  ws.ring_r[t & 255] = rb;
  ws.ring_q[t & 255] = qb;
  ws.loaded += 64;
}

#ifdef PGA_DP_STATS   // development aid: per call-site DP step / cycle totals and a log2 histogram of steps per call
__device__ unsigned long long g_dp_stats[3][40];
__device__ int g_dp_site;
#endif

// A search that anim_extdp_lane_kernel hands over mid-way: the latest H of the 64 diagonals, X / Y of the 32 cells of
// anti-diagonal d (diagonals l = (d + koff) mod 2, + 2, ...), and the best cell so far (its key with d in the low bits;
// bpay = its error field << 6 | its diagonal).
struct ExtDump {
  uint32_t H[64], X[32], Y[32];
  uint32_t best, bpay;
  int32_t d, pad_;
};

__device__ ExtResult extend_wave(const SeqView& R, const StrandView& Q, int64_t r0, int64_t q0, int dir, int32_t rmax,
                                 int32_t qmax, int32_t tr, int32_t tq, const ExtDump* resume = nullptr) {
  constexpr int W = BAND / 2;
#ifdef PGA_DP_STATS
  const unsigned long long t_begin = __builtin_readcyclecounter();
  int32_t n_steps = 0;
#endif
  static_assert(BAND == 64, "one lane per diagonal");
  __shared__ uint8_t s_ring[2][256];
  const int lane = threadIdx.x & 63;
  ExtResult res{0, 0, 0, 0, 0};
  bool targeted = tr >= 0;
  int koff = 0;   // band placement, see pga::extend_banded
  if (targeted) {
    koff = (tq - tr) / 2;
    if (koff > W - 2) koff = W - 2;
    if (koff < -(W - 2)) koff = -(W - 2);
    const int lt = (tq - tr) - koff + W;
    if (lt < 0 || lt >= BAND || tr > rmax || tq > qmax) { targeted = false; koff = 0; }
  }
  if (targeted && tr == 0 && tq == 0) { res.reached = 1; return res; }
  const int k = lane - W + koff;
  // Cell (i, j) = ((d - k) / 2, (d + k) / 2) of this lane's diagonal exists on anti-diagonal d iff d_lo <= d <= d_hi.
  //
  // Score and error count of a cell travel as ONE unsigned key  K = (score + 65536) << 15 | (32767 - errors):  integer
  // max on keys is exactly dp_cell's rule "higher score, then fewer errors", a move is one (saturating) add / subtract of
  // a constant, a wave shift moves both fields at once, and 0 is "dead" (what a DPP shift delivers at the band's edges
  // and what saturation keeps for negative moves).  Ranges: |score| <= 3 * 9999 and errors <= 2 * 9999 within MUMmer's
  // 10 kb DP limit, so no field overflows; a dead cell cannot climb to K_LIVE (score -32768) within 10^4 matches.
  // The per-lane best is one word too: (score + 65536) << 15 | d, so "higher score, ties: the later cell" is again max.
  constexpr uint32_t K_LIVE = 32768u << 15, K_TOP = 0xFFFF8000u;
  constexpr uint32_t K_OPEN = (uint32_t)(-SC_GAP_OPEN) * 32768u + 1u, K_EXT = (uint32_t)(-SC_GAP_EXT) * 32768u + 1u;
  constexpr uint32_t K_MATCH = (uint32_t)SC_MATCH * 32768u, K_MISMATCH = (uint32_t)(-SC_MISMATCH) * 32768u + 1u;
  const int32_t d_lo = k < 0 ? -k : k;
  const int32_t d_hi = (2 * rmax + k) < (2 * qmax - k) ? (2 * rmax + k) : (2 * qmax - k);
  uint32_t H = 0, X = 0, Y = 0;
  uint32_t best = K_LIVE, be = 0;   // per-lane best cell: (score + 65536) << 15 | d, and that cell's key
  if (lane == W - koff) { H = (65536u << 15) | 32767u; best = 65536u << 15; be = H; }
  const int32_t d_end = targeted ? tr + tq : rmax + qmax;
  // A resumed search picks up after anti-diagonal resume->d: every lane takes its diagonal's H, the lanes of that
  // anti-diagonal's parity their X / Y as well (the others' have been consumed), the best cell's lane the best.
  int32_t d_start = 1;
  if (resume) {
    d_start = resume->d + 1;
    H = resume->H[lane]; X = 0; Y = 0;
    if (((resume->d + k) & 1) == 0) { X = resume->X[lane >> 1]; Y = resume->Y[lane >> 1]; }
    best = K_LIVE; be = 0;
    if (lane == (int)(resume->bpay & 63u)) { best = resume->best; be = resume->bpay >> 6; }
  }
  WaveSeq ws{s_ring[0], s_ring[1], 0};
  if (resume) { const int32_t first = (d_start >> 1) - 64; ws.loaded = first > 0 ? (first & ~63) : 0; }
  __syncthreads();  // previous user of the ring (same wave) is done
  wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
  wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
  while ((d_start >> 1) + 36 > ws.loaded) wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);   // (resume only)
  __syncthreads();
  // ring positions of this lane's next cell (it has one on every other anti-diagonal; both advance by one each time)
  const int32_t d_first = ((d_start + k) & 1) ? d_start + 1 : d_start;
  uint32_t ir = (uint32_t)(((d_first - k) >> 1) - 1) & 255u, iq = (uint32_t)(((d_first + k) >> 1) - 1) & 255u;
  uint32_t par = (uint32_t)(d_start + k) & 1u;   // (d + k) & 1 at the first step; toggles every step
  uint8_t rb = ws.ring_r[ir], qb = ws.ring_q[iq];   // bases of the next cell, read one cell ahead (LDS latency off the path)
  // Break rule with PER-STEP semantics (as the scalar code) at the price of one wave reduction every CHECK steps:
  // g_key / t_prev = global best score (as a key with d = 0) and its anti-diagonal as of the last check; every lane
  // remembers the first step since then at which it matched or beat it (fimp) and a snapshot of its best.
  constexpr int CHECK = 16;
  uint32_t g_key = 65536u << 15;
  int32_t t_prev = 0, fimp = 0x7FFFFFFF;
  if (resume) { g_key = resume->best & K_TOP; t_prev = (int32_t)(resume->best & 32767u); }   // the hand-over is a check point
  uint32_t sbest = best, sbe = be;
  for (int32_t d = d_start; d <= d_end; ++d) {
#ifdef PGA_DP_STATS
    ++n_steps;
#endif
    if ((d >> 1) + 36 > ws.loaded) {   // uniform; covers the diagonals of a shifted band (|koff| <= 30)
      wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
      __syncthreads();
    }
    const uint32_t up_H = dpp_from_above0(H), up_X = dpp_from_above0(X);
    const uint32_t lf_H = dpp_from_below0(H), lf_Y = dpp_from_below0(Y);
    if (par == 0) {
      const bool ok = rb == qb;   // dirty codes differ (4 vs 5): never equal
      ir = (ir + 1) & 255u; iq = (iq + 1) & 255u;
      rb = ws.ring_r[ir]; qb = ws.ring_q[iq];   // staged at least 18 entries ahead of any cell of the next two steps
      const uint32_t xa = __builtin_elementwise_sub_sat(up_H, K_OPEN), xb = __builtin_elementwise_sub_sat(up_X, K_EXT);
      const uint32_t ya = __builtin_elementwise_sub_sat(lf_H, K_OPEN), yb = __builtin_elementwise_sub_sat(lf_Y, K_EXT);
      const uint32_t nx = xa > xb ? xa : xb, ny = ya > yb ? ya : yb;
      uint32_t nh = ok ? H + K_MATCH : __builtin_elementwise_sub_sat(H, K_MISMATCH);
      nh = nh > nx ? nh : nx;
      nh = nh > ny ? nh : ny;
      const bool alive = d >= d_lo && d <= d_hi;
      H = alive ? nh : 0u; X = alive ? nx : 0u; Y = alive ? ny : 0u;
      const uint32_t cellkey = (H & K_TOP) | (uint32_t)d;
      if (cellkey > best) be = H;                          // higher score, or the same score on a later cell
      best = cellkey > best ? cellkey : best;
      const int32_t imp = cellkey >= g_key ? d : 0x7FFFFFFF;   // matched or beat the best known at the last check
      fimp = imp < fimp ? imp : fimp;
    }
    par ^= 1u;
    if ((d % CHECK) == 0 || d == d_end) {
      const uint32_t key = wave_max_u32(best);            // max score, ties: larger d
      const int32_t t = (int32_t)(key & 32767u);
      const int32_t b = t_prev + BREAK_LEN;          // step at which the per-step rule (d - best_d >= BREAK_LEN) fires without an improvement
      const int32_t d1 = (int32_t)wave_min_u32((uint32_t)fimp);
      if (b <= d && d1 > b) {                        // it fired before the first improvement of this interval
        best = sbest; be = sbe;                      // results as of the last check (nothing global changed until b)
        break;
      }
      if (!__any(H >= K_LIVE)) break;
      g_key = key & K_TOP; t_prev = t; fimp = 0x7FFFFFFF;
      sbest = best; sbe = be;
    }
    if (targeted && d == d_end) {
      const int lt = (tq - tr) - koff + W;
      const uint32_t tH = (uint32_t)__shfl((int)H, lt, 64);
      if (tH >= K_LIVE) {
        res.di = tr; res.dj = tq; res.score = (int32_t)(tH >> 15) - 65536; res.errors = 32767 - (int32_t)(tH & 32767u); res.reached = 1;
        break;
      }
    }
  }
#ifdef PGA_DP_STATS
  if (lane == 0) {
    const int site = dir < 0 ? 2 : (tr >= 0 && tr == rmax && tq == qmax ? 0 : 1);
    atomicAdd(&g_dp_stats[site][0], 1ull);
    atomicAdd(&g_dp_stats[site][1], (unsigned long long)n_steps);
    atomicAdd(&g_dp_stats[site][2], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&g_dp_stats[site][8 + (31 - __clz(n_steps | 1))], 1ull);
  }
#endif
  if (res.reached) return res;
  // best cell: max score, ties -> larger d, then larger diagonal
  const long long key = wave_max64(((long long)best << 6) | (long long)lane);
  const int bl = (int)(key & 63);
  const uint32_t bkey = (uint32_t)(key >> 6);
  const int32_t gd = (int32_t)(bkey & 32767u);
  res.score = (int32_t)(bkey >> 15) - 65536;
  res.errors = 32767 - (int32_t)((uint32_t)__shfl((int)be, bl, 64) & 32767u);
  const int kk = bl - W + koff;
  res.di = (gd - kk) / 2; res.dj = (gd + kk) / 2;
  return res;
}

// Wave version of pga::thin_rect_errors: full DP of a small rectangle whose SHORT side (<= 63) is spread over the
// lanes; skewed wavefront: lane t works on line t of the short side, step s handles the cells with long-side index
// s - t, so the three predecessors are the lane's own previous cell and the lower neighbour's last two cells.
__device__ int32_t thin_rect_errors_wave(const SeqView& R, const StrandView& Q, int64_t r0, int32_t n, int64_t q0, int32_t m) {
  if (n < 0 || m < 0 || n > THIN_LONG || m > THIN_LONG || (n > THIN_MAX && m > THIN_MAX)) return -1;
  __shared__ uint8_t s_long[THIN_LONG + 1];
  const int lane = threadIdx.x & 63;
  const bool lanes_q = m <= THIN_MAX;            // lanes over query columns (rows = ref) or over ref rows
  const int32_t n_short = lanes_q ? m : n, n_long = lanes_q ? n : m;
  __syncthreads();
  for (int32_t t = lane; t < n_long; t += 64) {  // stage the long side's bases
    uint8_t b = 4;
    if (lanes_q) { if (R.clean(r0 + t)) b = (uint8_t)R.base(r0 + t); }
    else { if (Q.clean(q0 + t)) b = (uint8_t)Q.base(q0 + t); }
    s_long[t] = b;
  }
  uint8_t mine = 4;                              // this lane's base on the short side (line `lane`, 1-based)
  if (lane >= 1 && lane <= n_short) {
    if (lanes_q) { if (Q.clean(q0 + lane - 1)) mine = (uint8_t)Q.base(q0 + lane - 1); }
    else { if (R.clean(r0 + lane - 1)) mine = (uint8_t)R.base(r0 + lane - 1); }
  }
  __syncthreads();
  const DpCell dead{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0};
  DpCell cur = dead, prev = dead;
  for (int32_t s = 0; s <= n_short + n_long; ++s) {
    // neighbour (lane - 1): its cell of step s-1 (same long index) and of step s-2 (long index - 1)
    DpCell nb1, nb2;
    nb1.h = from_lane_below(cur.h, NEG_INF); nb1.he = from_lane_below(cur.he, 0);
    nb1.x = from_lane_below(cur.x, NEG_INF); nb1.xe = from_lane_below(cur.xe, 0);
    nb1.y = from_lane_below(cur.y, NEG_INF); nb1.ye = from_lane_below(cur.ye, 0);
    nb2.h = from_lane_below(prev.h, NEG_INF); nb2.he = from_lane_below(prev.he, 0);
    const int32_t u = s - lane;                  // long-side index of this lane's cell
    if (lane <= n_short && u >= 0 && u <= n_long) {
      DpCell c;
      if (lane == 0 && u == 0) {
        c = DpCell{0, 0, NEG_INF, 0, NEG_INF, 0};
      } else {
        const bool ok = lane >= 1 && u >= 1 && mine < 4 && s_long[u - 1] == mine;
        if (lanes_q) {   // i = u (rows, long), j = lane: up = own previous cell, left = neighbour (step s-1), diag = neighbour (s-2)
          c = dp_cell(u >= 1, cur.h, cur.he, cur.x, cur.xe, lane >= 1, nb1.h, nb1.he, nb1.y, nb1.ye, u >= 1 && lane >= 1, nb2.h, nb2.he, ok);
        } else {         // i = lane (rows, short), j = u: up = neighbour (step s-1), left = own previous cell, diag = neighbour (s-2)
          c = dp_cell(lane >= 1, nb1.h, nb1.he, nb1.x, nb1.xe, u >= 1, cur.h, cur.he, cur.y, cur.ye, u >= 1 && lane >= 1, nb2.h, nb2.he, ok);
        }
      }
      prev = cur;
      cur = c;
    }
  }
  return __shfl(cur.he, n_short, 64);            // cell (n, m) lives in lane n_short after the last step
}

__device__ int32_t gap_errors_wave(const SeqView& R, const StrandView& Q, int64_t r0, int32_t n, int64_t q0, int32_t m) {
  if (n == 0) return m;
  if (m == 0) return n;
  if (n == m && n <= GAP_DIAG_MAX) {   // see pga::gap_errors: <= 2 substitutions on one diagonal need no DP
    int32_t err = 0;
    for (int32_t t = 0; t < n; ++t) err += (R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t)) ? 0 : 1;
    if (err <= 2) return err;
  }
  const ExtResult e = extend_wave(R, Q, r0, q0, +1, n, m, n, m);
  if (e.reached) return e.errors;
  int32_t kq = n < m ? n : m, err = (n > m ? n - m : m - n);
  for (int32_t t = 0; t < kq; ++t) err += (R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t)) ? 0 : 1;
  return err;
}

// work list entries (unit, chain) of unit u at wl[choff[u] .. choff[u+1])
__global__ __launch_bounds__(64) void anim_wl_kernel(const uint32_t* __restrict__ choff, uint2* __restrict__ wl) {
  const uint32_t u = blockIdx.x, b = choff[u], n = choff[u + 1] - b;
  for (uint32_t c = threadIdx.x; c < n; c += 64) wl[b + c] = make_uint2(u, c);
}

// ---- A4a: gaps between the chained matches ---------------------------------------------------------------------------
// One wave per chain walks its matches 64 at a time (pga::chain_inner_errors' trimming, lane = match).  Gaps that need
// no DP (empty on one side, or <= 2 substitutions on one diagonal) are settled in the lane; the others become GapTasks
// for the DP kernels, so that a chain with 10^5 matches no longer occupies a single wave for its whole DP work.
struct GapTask {
  uint32_t unit;
  int32_t chain;
  int32_t r0, n, q0, m;
};
constexpr int GAP_LANE_MAX = 63;   // gaps up to 63 x 63 are solved by one lane each in anim_gapdp_lane_kernel (64 tasks per wave)
constexpr int GAP_CLASSES = 4;     // size classes of those gaps (max side <= 16 / 31 / 47 / 63): a wave gets tasks of one class

__device__ __forceinline__ int32_t wave_sum32(int32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(64) void anim_gaps_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                       ClusterOut O, const uint2* __restrict__ wl, ChainFwd* __restrict__ fw,
                                                       GapTask* __restrict__ tasks, uint8_t* __restrict__ task_cls) {
  const uint32_t u = wl[blockIdx.x].x;
  const int32_t c = (int32_t)wl[blockIdx.x].y;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const SeqView RV{R.codes, R.mask, R.len};
  const StrandView QV{SeqView{U.codes, U.mask, U.len}, U.strand};
  const size_t off = O.moff[u];
  const Chain ch = O.chains[off + c];
  const Match* cm = O.cm + off + ch.first;
  const int lane = threadIdx.x & 63;
  const Match f = cm[0];
  int32_t er = f.r + f.len, eq = f.q + f.len;   // uniform: end of the last kept match
  int32_t inner = 0;                              // per-lane partial sum
  for (int32_t base = 1; base < ch.count; base += 64) {
    const int32_t kq = base + lane;
    const bool valid = kq < ch.count;
    Match t = valid ? cm[kq] : Match{0, 0, 0, 0};
    const Match pm = (valid && lane > 0) ? cm[kq - 1] : Match{0, 0, 0, 0};
    // optimistic: the predecessor is kept, so the running end is the predecessor's end
    int32_t ger = lane == 0 ? er : pm.r + pm.len, geq = lane == 0 ? eq : pm.q + pm.len;
    int32_t trim = ger - t.r;
    if (geq - t.q > trim) trim = geq - t.q;
    if (trim < 0) trim = 0;
    bool skip = valid && t.len - trim <= 0;
    if (__any(skip)) {
      // a match swallowed by the running end: redo this block serially (uniform loop, every lane keeps its own result)
      for (int i = 0; i < 64 && base + i < ch.count; ++i) {
        const int32_t tr_ = __shfl(t.r, i, 64), tq_ = __shfl(t.q, i, 64), tl_ = __shfl(t.len, i, 64);
        int32_t tm = er - tr_;
        if (eq - tq_ > tm) tm = eq - tq_;
        if (tm < 0) tm = 0;
        const bool sk = tl_ - tm <= 0;
        if (lane == i) { ger = er; geq = eq; trim = tm; skip = sk; }
        if (!sk) { er = tr_ + tl_; eq = tq_ + tl_; }
      }
    } else {
      const int last = (ch.count - base < 64 ? ch.count - base : 64) - 1;
      er = __shfl(t.r + t.len, last, 64);
      eq = __shfl(t.q + t.len, last, 64);
    }
    bool hard = false;
    int32_t gn = 0, gm = 0;
    if (valid && !skip) {
      gn = t.r + trim - ger; gm = t.q + trim - geq;
      if (gn == 0) inner += gm;
      else if (gm == 0) inner += gn;
      else {
        hard = true;
        if (gn == gm && gn <= GAP_DIAG_MAX) {
          int32_t e = 0;
          for (int32_t x = 0; x < gn; ++x)
            e += (RV.clean(ger + x) && QV.clean(geq + x) && RV.base(ger + x) == QV.base(geq + x)) ? 0 : 1;
          if (e <= 2) { inner += e; hard = false; }
        }
      }
    }
    // A hard gap becomes a GapTask in the slot of the match it precedes (slots are unique, so no counter is contended);
    // its size class goes to the byte plane that anim_gapsort_kernel turns into per-class task lists.
    if (hard) {
      const int32_t mx = gn > gm ? gn : gm;
      const size_t slot = off + ch.first + kq;
      tasks[slot] = GapTask{u, c, ger, gn, geq, gm};
      task_cls[slot] = (uint8_t)(mx > GAP_LANE_MAX ? GAP_CLASSES : mx <= 16 ? 0 : mx <= 31 ? 1 : mx <= 47 ? 2 : 3);
    }
  }
  inner = wave_sum32(inner);
  if (lane == 0) {
    ChainFwd e;
    e.first_r = f.r; e.first_q = f.q;
    e.inner_err = inner;
    e.lr = er; e.lq = eq;
    e.re = er; e.qe = eq; e.err_fwd = 0; e.reached = 0; e.target = -1;
    fw[off + c] = e;
  }
}

// Task lists by size class from the class plane (0xFF = no task in the slot): list k holds the slots of the gaps with
// both sides <= 16 / 31 / 47 / 63 (k = 0..3, one LANE each in anim_gapdp_lane_kernel, so a wave gets 64 tasks of one
// size) and of the larger ones (k = 4: requests for anim_extdp_lane_kernel, see anim_gapreq_kernel).  A block sorts 4096 slots with LDS counters
// and reserves its share of every list with one global atomic per class.
constexpr int GAPSORT_BLOCK = 256;
__global__ __launch_bounds__(GAPSORT_BLOCK) void anim_gapsort_kernel(const uint8_t* __restrict__ task_cls, uint32_t n_slots,
                                                                     uint32_t* __restrict__ lists, uint32_t* __restrict__ n_tasks) {
  __shared__ uint32_t cnt[GAP_CLASSES + 1], gbase[GAP_CLASSES + 1];
  const uint32_t n_vec = (n_slots + 15u) / 16u;   // the plane is padded to whole 16-byte words
  for (uint32_t v0 = blockIdx.x * GAPSORT_BLOCK; v0 < n_vec; v0 += gridDim.x * GAPSORT_BLOCK) {
    if (threadIdx.x <= GAP_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t v = v0 + threadIdx.x;
    uint4 w = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (v < n_vec) w = reinterpret_cast<const uint4*>(task_cls)[v];
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t mine[GAP_CLASSES + 1] = {0, 0, 0, 0, 0};
    if ((w.x & w.y & w.z & w.w) != ~0u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t c = (ww[i >> 2] >> (8 * (i & 3))) & 0xFFu;
#pragma unroll
        for (int k = 0; k <= GAP_CLASSES; ++k) mine[k] += c == (uint32_t)k;
      }
    }
    uint32_t lbase[GAP_CLASSES + 1];
#pragma unroll
    for (int k = 0; k <= GAP_CLASSES; ++k) lbase[k] = mine[k] ? atomicAdd(&cnt[k], mine[k]) : 0u;
    __syncthreads();
    if (threadIdx.x <= GAP_CLASSES) gbase[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&n_tasks[threadIdx.x], cnt[threadIdx.x]) : 0u;
    __syncthreads();
    if ((w.x & w.y & w.z & w.w) != ~0u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t c = (ww[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        if (c <= (uint32_t)GAP_CLASSES) {
          uint32_t at = 0;
#pragma unroll
          for (int k = 0; k <= GAP_CLASSES; ++k)
            if (c == (uint32_t)k) at = gbase[k] + lbase[k]++;
          lists[(size_t)c * n_slots + at] = v * 16u + (uint32_t)i;
        }
      }
    }
    __syncthreads();
  }
}


// ---- small gaps: one LANE per GapTask ---------------------------------------------------------------------------
// 64 positions of a sequence starting at p0 (any sign): 2-bit codes in c[0..3] (position p0 in the low bits of c[0]) and
// ok bit k = position p0 + k lies inside the sequence and is clean.
__device__ __forceinline__ void seq_window64(const SeqView& s, int64_t p0, uint32_t c[4], uint64_t& ok) {
  const int64_t last_c = (s.len - 1) >> 4, last_m = (s.len - 1) >> 5;
  const int64_t w0 = p0 >> 4, m0 = p0 >> 5;   // floor
  uint32_t w[5], mw[3];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int64_t i = w0 + k;
    i = i < 0 ? 0 : i > last_c ? last_c : i;   // a clamped word only stands in for positions outside the sequence
    w[k] = s.codes[i];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int64_t i = m0 + k;
    i = i < 0 ? 0 : i > last_m ? last_m : i;
    mw[k] = s.mask[i];
  }
  const uint32_t sc = 2u * (uint32_t)(p0 & 15), sm = (uint32_t)(p0 & 31);
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = __funnelshift_r(w[k], w[k + 1], sc);
  ok = (uint64_t)__funnelshift_r(mw[0], mw[1], sm) | ((uint64_t)__funnelshift_r(mw[1], mw[2], sm) << 32);
  const int64_t lo = p0 < 0 ? -p0 : 0, hi = s.len - p0;   // window bits [lo, hi) are inside the sequence
  const uint64_t below_hi = hi >= 64 ? ~0ull : hi <= 0 ? 0ull : ((1ull << hi) - 1ull);
  const uint64_t below_lo = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
  ok &= below_hi & ~below_lo;
}
__device__ __forceinline__ uint32_t rev_fields2(uint32_t x) {   // the 16 two-bit fields of x in reverse order
  x = __brev(x);
  return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
__device__ __forceinline__ uint32_t spread16(uint32_t x) {   // bit k of the low half -> bit 2k
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  return (x | (x << 1)) & 0x55555555u;
}

// Cells j = J .. C-1 of one DP row of the lane kernel below, as nested uniform ifs (columns beyond the wave's largest m
// are skipped with one forward jump, and H / X stay in fixed registers).  hdiag = H(i-1, j-1), hleft = H(i, j-1),
// yleft = Y(i, j-1).  In the banded variant only H of a cell outside [blo, blo + bwid] is killed: the X of a cell right
// of the band derives only from killed cells above it and its Y only feeds cells further right; left of the band it is
// the other way round (Y derives from killed cells, X only feeds cells further down) — neither reaches a band cell.
template <int C, bool BANDED, int J>
struct LaneRow {
  static constexpr int QW = (C + 14) / 16;
  static __device__ __forceinline__ void run(uint32_t (&H)[C], uint32_t (&X)[C], const uint32_t (&eq)[QW], bool row0, int32_t m_max,
                                             uint32_t blo, uint32_t bwid, uint32_t hdiag, uint32_t hleft, uint32_t yleft) {
    constexpr uint32_t K_START = (65536u << 15) | 32767u;
    constexpr uint32_t K_OPEN = (uint32_t)(-SC_GAP_OPEN) * 32768u + 1u, K_EXT = (uint32_t)(-SC_GAP_EXT) * 32768u + 1u;
    constexpr uint32_t K_MATCH = (uint32_t)SC_MATCH * 32768u, K_MISMATCH = (uint32_t)(-SC_MISMATCH) * 32768u + 1u;
    if (J > m_max) return;   // uniform
    const uint32_t up_h = H[J], up_x = X[J];
    const uint32_t xa = __builtin_elementwise_sub_sat(up_h, K_OPEN), xb = __builtin_elementwise_sub_sat(up_x, K_EXT);
    const uint32_t nx = xa > xb ? xa : xb;
    uint32_t ny = 0, nh;
    if (J == 0) {
      nh = row0 ? K_START : nx;
    } else {
      const uint32_t ya = __builtin_elementwise_sub_sat(hleft, K_OPEN), yb = __builtin_elementwise_sub_sat(yleft, K_EXT);
      ny = ya > yb ? ya : yb;
      constexpr int Q = J > 0 ? J - 1 : 0;
      const uint32_t bit = (eq[Q >> 4] >> (2 * (Q & 15))) & 1u;
      nh = __umul24(bit, K_MATCH + K_MISMATCH) + __builtin_elementwise_sub_sat(hdiag, K_MISMATCH);
      nh = nh > nx ? nh : nx;
      nh = nh > ny ? nh : ny;
    }
    if (BANDED) nh = ((uint32_t)J - blo <= bwid) ? nh : 0u;
    H[J] = nh; X[J] = nx;
    LaneRow<C, BANDED, J + 1>::run(H, X, eq, row0, m_max, blo, bwid, up_h, nh, ny);
  }
};
template <int C, bool BANDED>
struct LaneRow<C, BANDED, C> {
  static constexpr int QW = (C + 14) / 16;
  static __device__ __forceinline__ void run(uint32_t (&)[C], uint32_t (&)[C], const uint32_t (&)[QW], bool, int32_t, uint32_t, uint32_t,
                                             uint32_t, uint32_t, uint32_t) {}
};

// Small gaps, one LANE per GapTask (64 tasks of one size class per wave): the cells of the wave DP of a targeted gap
// fill — the n x m rectangle restricted to the 64 diagonals centred between start and target (pga::extend_banded) —
// row by row, the previous row's H and X keys held in REGISTERS (the column loop is fully unrolled, C - 1 = the largest
// gap side of the class) and the same unsigned keys, so every choice is again "higher score, then fewer errors".
//   * rows beyond a lane's n are masked off, so its registers end holding row n; columns beyond its m compute garbage
//     that never flows back (a cell only reads columns <= its own) and that stays far below K_LIVE;
//   * the reference base of the row is compared with all query bases at once (xor of the 2-bit codes, clean masks
//     folded in), a cell takes its bit of that word: mismatch penalty always, + (match + mismatch) * bit;
//   * BANDED = false (n + m <= 62): every diagonal of the rectangle is inside the band, no range test per cell;
//     BANDED = true: H of the cells outside [i + klo, min(m, i + khi)] is killed (see LaneRow); a target outside the
//     band falls back to the diagonal count exactly like pga::gap_errors.
// A gap of at most 63 + 63 anti-diagonals can never trigger the break rule of the extension DP.
// ~10 (13 banded) VALU per cell for 64 tasks at once, against ~35 per anti-diagonal for ONE task in the wave version.
template <int C, bool BANDED>
__global__ __launch_bounds__(64) void anim_gapdp_lane_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                             ClusterOut O, const GapTask* __restrict__ tasks, const uint32_t* __restrict__ list,
                                                             const uint32_t* __restrict__ n_tasks, ChainFwd* __restrict__ fw) {
  constexpr int W = BAND / 2;
  constexpr int QW = (C + 14) / 16;   // code words that hold query bases 0 .. C-2
  constexpr uint32_t K_LIVE = 32768u << 15;
  const int lane = threadIdx.x & 63;
  const uint32_t n_all = *n_tasks;
  for (uint32_t base = blockIdx.x * 64u; base < n_all; base += gridDim.x * 64u) {
    const bool valid = base + lane < n_all;
    GapTask T{0, 0, 0, 0, 0, 0};
    if (valid) T = tasks[list[base + lane]];
    const UnitDesc U = units[T.unit];
    const RefDesc R = refs[U.ref];
    const SeqView RV{R.codes, R.mask, R.len};
    const SeqView QS{U.codes, U.mask, U.len};
    const int32_t n = T.n, m = T.m;
    // band placement and target test of pga::extend_banded for tr = n, tq = m
    int koff = (m - n) / 2;
    if (koff > W - 2) koff = W - 2;
    if (koff < -(W - 2)) koff = -(W - 2);
    const int lt = (m - n) - koff + W;
    const bool in_band = valid && lt >= 0 && lt < BAND;
    const int klo = koff - W, khi = koff + W - 1;   // diagonals j - i inside the band
    const int32_t n_max = (int32_t)wave_max_u32(in_band ? (uint32_t)n : 0u), m_max = (int32_t)wave_max_u32(in_band ? (uint32_t)m : 0u);
    // the two windows: reference bases r0 .. r0+63, query-strand bases q0 .. q0+63
    uint32_t rc[4], qc[4];
    uint64_t rok, qok;
    seq_window64(RV, T.r0, rc, rok);
    if (U.strand) {   // strand position p = forward position len-1-p, complemented
      uint32_t f[4];
      uint64_t fok;
      seq_window64(QS, U.len - 1 - (int64_t)T.q0 - 63, f, fok);
#pragma unroll
      for (int k = 0; k < 4; ++k) qc[k] = ~rev_fields2(f[3 - k]);
      qok = ((uint64_t)__brev((uint32_t)fok) << 32) | (uint64_t)__brev((uint32_t)(fok >> 32));
    } else {
      seq_window64(QS, T.q0, qc, qok);
    }
    uint32_t qs[QW];   // clean query bases, one bit per 2-bit field
#pragma unroll
    for (int k = 0; k < QW; ++k) qs[k] = spread16((uint32_t)(qok >> (16 * k)) & 0xFFFFu);
    uint32_t H[C], X[C];
#pragma unroll
    for (int j = 0; j < C; ++j) { H[j] = 0; X[j] = 0; }
    for (int32_t i = 0; i <= n_max; ++i) {
      if (in_band && i <= n) {
        // eq bit 2j = query base j equals the reference base of this row (row 0 has none)
        uint32_t sel = (i >= 1 && (rok & 1ull)) ? ~0u : 0u;
        const uint32_t rb = (rc[0] & 3u) * 0x55555555u;
        uint32_t eq[QW];
#pragma unroll
        for (int k = 0; k < QW; ++k) {
          const uint32_t x = qc[k] ^ rb;
          eq[k] = ~(x | (x >> 1)) & qs[k] & sel;
        }
        if (i >= 1) {   // uniform
          rc[0] = __funnelshift_r(rc[0], rc[1], 2); rc[1] = __funnelshift_r(rc[1], rc[2], 2);
          rc[2] = __funnelshift_r(rc[2], rc[3], 2); rc[3] >>= 2;
          rok >>= 1;
        }
        uint32_t blo = 0, bwid = 0;
        if (BANDED) {
          const int32_t lo = i + klo > 0 ? i + klo : 0, hi = i + khi < m ? i + khi : m;
          blo = hi >= lo ? (uint32_t)lo : (1u << 20);
          bwid = hi >= lo ? (uint32_t)(hi - lo) : 0u;
        }
        int32_t m_row = m_max;
        asm volatile("" : "+s"(m_row));   // keeps the 64 column tests as scalar compares in the row instead of 64 hoisted masks
        LaneRow<C, BANDED, 0>::run(H, X, eq, i == 0, m_row, blo, bwid, 0u, 0u, 0u);
      }
    }
    if (valid) {
      uint32_t tH = 0;
#pragma unroll
      for (int j = 0; j < C; ++j) tH = m == j ? H[j] : tH;
      int32_t err;
      if (in_band && tH >= K_LIVE) {
        err = 32767 - (int32_t)(tH & 32767u);
      } else {   // target outside the band or pruned: diagonal part + length difference (pga::gap_errors)
        const StrandView QV{QS, U.strand};
        const int32_t kq = n < m ? n : m;
        err = n > m ? n - m : m - n;
        for (int32_t t = 0; t < kq; ++t)
          err += (RV.clean(T.r0 + t) && QV.clean(T.q0 + t) && RV.base(T.r0 + t) == QV.base(T.q0 + t)) ? 0 : 1;
      }
      if (err) atomicAdd(&fw[O.moff[T.unit] + T.chain].inner_err, err);
    }
  }
}

// ---- extension DP, one LANE per task -------------------------------------------------------------------------------
// The first DP calls of a chain in a phase need nothing but the chain tables (and the results of the calls before them),
// so they are written down as ExtReqs by anim_extreq_kernel (one round per call), solved here 64 to a wave, and picked up
// by anim_extend_kernel in place of its own extend_wave calls; whatever the lanes have not delivered still runs there.
struct ExtReq {   // everything a lane needs, so that taking a request costs two dependent loads (request, sequence words)
  const uint32_t* rcodes;
  const uint32_t* rmask;
  const uint32_t* qcodes;
  const uint32_t* qmask;
  int32_t rlen, qlen;
  int32_t strand, dir;
  int32_t r0, q0, rmax, qmax, tr, tq;
  uint32_t chain;   // index into the work list: where the result goes
  int32_t pad_;
};
struct ExtArgs {   // the arguments of one DP call, for the check in anim_extend_kernel
  int32_t r0, q0, dir, rmax, qmax, tr, tq;
};
struct ExtPre {
  ExtResult res;
  ExtArgs args;
  int32_t valid;
};
constexpr int EXT_ROUNDS = 2;   // DP calls per chain and phase that may go to the lanes (a third one is rare)
constexpr int EXT_TAIL_LANES = 24, EXT_TAIL_BLOCKS = 64;   // see the tail rule in anim_extdp_lane_kernel
constexpr uint32_t EXT_DUMP_CAP = 1u << 17;

// One sequence of a lane task as a stream: element t = the base at stored position start + sgn * t (complemented for a
// reverse query strand), valid while t < tmax and the position is a clean base.  Up to 32 elements are buffered (2-bit
// codes, and valid flags as 01 per element), the next one in the low bits.
struct LaneSeq {
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len, start, sgn, tmax;
  uint32_t comp;
  uint32_t bc_lo, bc_hi, bo_lo, bo_hi;
  int32_t cnt, chunk;   // buffered elements; next 16-element chunk to fetch
};
__device__ __forceinline__ void lane_seq_fetch(const LaneSeq& s, uint32_t& c, uint32_t& okf) {
  const int32_t t0 = s.chunk * 16;
  const int32_t p0 = s.sgn > 0 ? s.start + t0 : s.start - t0 - 15;
  const int32_t last_c = (s.len - 1) >> 4, last_m = (s.len - 1) >> 5;
  int32_t w0 = p0 >> 4, w1 = w0 + 1, m0 = p0 >> 5, m1 = m0 + 1;   // floor; a clamped word only stands in for outside positions
  w0 = w0 < 0 ? 0 : w0 > last_c ? last_c : w0; w1 = w1 < 0 ? 0 : w1 > last_c ? last_c : w1;
  m0 = m0 < 0 ? 0 : m0 > last_m ? last_m : m0; m1 = m1 < 0 ? 0 : m1 > last_m ? last_m : m1;
  c = __funnelshift_r(s.codes[w0], s.codes[w1], 2u * (uint32_t)(p0 & 15));
  uint32_t ok = __funnelshift_r(s.mask[m0], s.mask[m1], (uint32_t)(p0 & 31)) & 0xFFFFu;
  const int32_t lo = p0 < 0 ? -p0 : 0, hi = s.len - p0;   // window bits [lo, hi) lie inside the sequence
  uint32_t in = hi >= 16 ? 0xFFFFu : hi <= 0 ? 0u : ((1u << hi) - 1u);
  in &= lo >= 16 ? 0u : ~((1u << lo) - 1u);
  ok &= in;
  if (s.sgn < 0) { c = rev_fields2(c); ok = __brev(ok) >> 16; }
  if (s.comp) c = ~c;
  const int32_t nv = s.tmax - t0;
  ok &= nv >= 16 ? 0xFFFFu : nv <= 0 ? 0u : ((1u << nv) - 1u);
  okf = spread16(ok);
}
__device__ __forceinline__ void lane_seq_append(LaneSeq& s) {   // needs cnt <= 16
  uint32_t c, okf;
  lane_seq_fetch(s, c, okf);
  const uint32_t sh = 2u * (uint32_t)s.cnt;
  const uint64_t bc = (((uint64_t)s.bc_hi << 32) | s.bc_lo) | ((uint64_t)c << sh);
  const uint64_t bo = (((uint64_t)s.bo_hi << 32) | s.bo_lo) | ((uint64_t)okf << sh);
  s.bc_lo = (uint32_t)bc; s.bc_hi = (uint32_t)(bc >> 32);
  s.bo_lo = (uint32_t)bo; s.bo_hi = (uint32_t)(bo >> 32);
  s.cnt += 16; s.chunk += 1;
}
__device__ __forceinline__ void lane_seq_pop(LaneSeq& s, uint32_t& c, uint32_t& o) {
  c = s.bc_lo & 3u; o = s.bo_lo & 1u;
  s.bc_lo = __funnelshift_r(s.bc_lo, s.bc_hi, 2); s.bc_hi >>= 2;
  s.bo_lo = __funnelshift_r(s.bo_lo, s.bo_hi, 2); s.bo_hi >>= 2;
  s.cnt -= 1;
}
// The two windows a lane compares on one anti-diagonal.  Its 32 cells sit on every other diagonal l = P, P+2, ...;
// cell t' needs ref element i0 - t' and query element j0 + t', so the ref window is kept REVERSED (element i0 in field 0)
// and the query window forward (element j0 in field 0): one xor compares all 32 pairs.  Stepping to the next
// anti-diagonal of the other parity moves exactly one of them by one element.
struct LaneWin {
  uint32_t rc_lo, rc_hi, ro_lo, ro_hi;   // ref: codes, valid flags (01 per element)
  uint32_t qc_lo, qc_hi, qo_lo, qo_hi;   // query
};
__device__ __forceinline__ void lane_win_push_ref(LaneWin& w, LaneSeq& s) {
  uint32_t c, o;
  lane_seq_pop(s, c, o);
  w.rc_hi = __funnelshift_r(w.rc_lo, w.rc_hi, 30); w.rc_lo = (w.rc_lo << 2) | c;
  w.ro_hi = __funnelshift_r(w.ro_lo, w.ro_hi, 30); w.ro_lo = (w.ro_lo << 2) | o;
}
__device__ __forceinline__ void lane_win_push_qry(LaneWin& w, LaneSeq& s) {
  uint32_t c, o;
  lane_seq_pop(s, c, o);
  w.qc_lo = __funnelshift_r(w.qc_lo, w.qc_hi, 2); w.qc_hi = (w.qc_hi >> 2) | (c << 30);
  w.qo_lo = __funnelshift_r(w.qo_lo, w.qo_hi, 2); w.qo_hi = (w.qo_hi >> 2) | (o << 30);
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b_uniform, uint32_t c) {   // a * b + c, a and b below 2^24
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}

// Cells l = P + 2T, P + 2T + 2, ... of one anti-diagonal of a lane task: pga::dp_cell on the unsigned keys of
// extend_wave, then the running best: higher score, ties the later anti-diagonal, then the larger diagonal (ascending l
// with >=).  bpay = the best cell's key << 6 | l: its error field and its diagonal.
// Registers: H[l] = H of the latest cell of diagonal l (the neighbours l - 1 and l + 1 belong to the other parity and
// are not written in this step).  The X and Y of a cell are read exactly once, by its neighbours on the next
// anti-diagonal, so ONE set of 32 serves both parities: oX / oY hold the previous anti-diagonal's (cell l + 1 is entry
// T + P, cell l - 1 entry T + P - 1), nX / nY receive this one's.
// LIMITS: H of the cells with i > rmax or j > qmax (bit l of alive_lo/hi clear) is killed — X of a cell past rmax only
// feeds cells further past it and its Y derives from killed cells, and the other way round past qmax.
template <int P, int T, bool LIMITS>
struct ExtLaneCell {
  static __device__ __forceinline__ void run(uint32_t (&H)[64], const uint32_t (&oX)[32], const uint32_t (&oY)[32], uint32_t (&nX)[32],
                                             uint32_t (&nY)[32], uint32_t e_lo, uint32_t e_hi, uint32_t d, uint32_t alive_lo,
                                             uint32_t alive_hi, uint32_t& best, uint32_t& bpay) {
    constexpr int L = P + 2 * T;
    constexpr uint32_t K_TOP = 0xFFFF8000u;
    constexpr uint32_t K_OPEN = (uint32_t)(-SC_GAP_OPEN) * 32768u + 1u, K_EXT = (uint32_t)(-SC_GAP_EXT) * 32768u + 1u;
    constexpr uint32_t K_MATCH = (uint32_t)SC_MATCH * 32768u, K_MISMATCH = (uint32_t)(-SC_MISMATCH) * 32768u + 1u;
    uint32_t nx = 0, ny = 0;
    if (L + 1 < 64) {
      const uint32_t xa = __builtin_elementwise_sub_sat(H[L + 1 < 64 ? L + 1 : 0], K_OPEN);
      const uint32_t xb = __builtin_elementwise_sub_sat(oX[T + P < 32 ? T + P : 0], K_EXT);
      nx = xa > xb ? xa : xb;
    }
    if (L >= 1) {
      const uint32_t ya = __builtin_elementwise_sub_sat(H[L >= 1 ? L - 1 : 0], K_OPEN);
      const uint32_t yb = __builtin_elementwise_sub_sat(oY[T + P >= 1 ? T + P - 1 : 0], K_EXT);
      ny = ya > yb ? ya : yb;
    }
    const uint32_t bit = ((T < 16 ? e_lo : e_hi) >> (2 * (T & 15))) & 1u;
    uint32_t nh = mad_u24(bit, K_MATCH + K_MISMATCH, __builtin_elementwise_sub_sat(H[L], K_MISMATCH));
    nh = nh > nx ? nh : nx;
    nh = nh > ny ? nh : ny;
    if (LIMITS) nh &= (uint32_t)((int32_t)((L < 32 ? alive_lo : alive_hi) << (31 - (L & 31))) >> 31);
    H[L] = nh; nX[T] = nx; nY[T] = ny;
    // P = 1 walks the cells upwards (ties: >=), P = 0 downwards (ties: >): either way the largest diagonal wins a tie, and
    // an entry of oX / oY is dead by the time the same entry of nX / nY is written, so the two can share registers
    const uint32_t ck = (nh & K_TOP) | d;
    bpay = (P == 1 ? ck >= best : ck > best) ? ((nh << 6) | (uint32_t)L) : bpay;
    best = ck > best ? ck : best;
    ExtLaneCell<P, (P == 1 ? T + 1 : T - 1), LIMITS>::run(H, oX, oY, nX, nY, e_lo, e_hi, d, alive_lo, alive_hi, best, bpay);
  }
};
template <bool LIMITS>
struct ExtLaneCell<0, -1, LIMITS> {
  static __device__ __forceinline__ void run(uint32_t (&)[64], const uint32_t (&)[32], const uint32_t (&)[32], uint32_t (&)[32],
                                             uint32_t (&)[32], uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t&, uint32_t&) {}
};
template <bool LIMITS>
struct ExtLaneCell<1, 32, LIMITS> {
  static __device__ __forceinline__ void run(uint32_t (&)[64], const uint32_t (&)[32], const uint32_t (&)[32], uint32_t (&)[32],
                                             uint32_t (&)[32], uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t&, uint32_t&) {}
};

// The per-lane scalars of a task.
struct LaneTask {
  uint32_t chain;          // where the result goes
  int32_t d, d_end, koff, lt, targeted, tr, tq;
  int32_t c1, c2;          // 2 * rmax - 32 + koff and 2 * qmax + 32 - koff: cell l is inside the limits iff d - c1 <= l <= c2 - d
  uint32_t best, bpay;
};
__device__ __forceinline__ ExtResult lane_best_result(const LaneTask& t) {
  ExtResult r{0, 0, 0, 0, 0};
  const int32_t gd = (int32_t)(t.best & 32767u), kk = (int32_t)(t.bpay & 63u) - BAND / 2 + t.koff;
  r.score = (int32_t)(t.best >> 15) - 65536; r.errors = 32767 - (int32_t)((t.bpay >> 6) & 32767u);
  r.di = (gd - kk) / 2; r.dj = (gd + kk) / 2;
  return r;
}

// One anti-diagonal (register parity P) for the lanes with go set.
template <int P, bool LIMITS>
__device__ __forceinline__ void ext_lane_step(bool go, uint32_t (&H)[64], uint32_t (&X)[32], uint32_t (&Y)[32], LaneWin& w, LaneSeq& rs,
                                              LaneSeq& qs, LaneTask& t) {
  if (go) {
    t.d += 1;
    if (P == 0) lane_win_push_ref(w, rs); else lane_win_push_qry(w, qs);
    const uint32_t x_lo = w.rc_lo ^ w.qc_lo, x_hi = w.rc_hi ^ w.qc_hi;
    const uint32_t e_lo = ~(x_lo | (x_lo >> 1)) & w.ro_lo & w.qo_lo, e_hi = ~(x_hi | (x_hi >> 1)) & w.ro_hi & w.qo_hi;
    uint32_t alive_lo = ~0u, alive_hi = ~0u;
    if (LIMITS) {
      const int32_t lo = t.d - t.c1 > 0 ? t.d - t.c1 : 0, hi = t.c2 - t.d;   // diagonals lo .. hi are inside both sequences
      uint64_t m = lo >= 64 ? 0ull : (~0ull << lo);
      m &= hi < 0 ? 0ull : hi >= 63 ? ~0ull : ((2ull << hi) - 1ull);
      alive_lo = (uint32_t)m; alive_hi = (uint32_t)(m >> 32);
    }
    uint32_t nX[32], nY[32];
    ExtLaneCell<P, (P == 1 ? 0 : 31), LIMITS>::run(H, X, Y, nX, nY, e_lo, e_hi, (uint32_t)t.d, alive_lo, alive_hi, t.best, t.bpay);
#pragma unroll
    for (int k = 0; k < 32; ++k) { X[k] = nX[k]; Y[k] = nY[k]; }
  }
}

// 32 anti-diagonals of every active lane, with the checks of pga::extend_banded after each: break rule first, then the
// end / the target.  LIMITS (uniform): some lane may come within reach of rmax / qmax during the block.
template <bool LIMITS>
__device__ __forceinline__ void ext_lane_block(bool& active, uint32_t (&H)[64], uint32_t (&X)[32], uint32_t (&Y)[32], LaneWin& w, LaneSeq& rs,
                                               LaneSeq& qs, LaneTask& t, ExtPre* __restrict__ pre) {
  constexpr uint32_t K_LIVE = 32768u << 15;
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      // a lane's next anti-diagonal d + 1 lives on the diagonals l = d + 1 + koff (mod 2)
      const bool go = active && (((t.d + 1 + t.koff) & 1) == P);
      if (P == 0) ext_lane_step<0, LIMITS>(go, H, X, Y, w, rs, qs, t); else ext_lane_step<1, LIMITS>(go, H, X, Y, w, rs, qs, t);
      bool fin = false, want = false;
      if (go) {
        if (t.d - (int32_t)(t.best & 32767u) >= BREAK_LEN) fin = true;
        else if (t.d == t.d_end) { fin = true; want = t.targeted != 0; }
      }
      if (__any(fin)) {   // uniform
        uint32_t tH = 0;
        if (__any(want)) {
#pragma unroll
          for (int l = 0; l < 64; ++l) tH = t.lt == l ? H[l] : tH;
        }
        if (fin) {
          ExtResult r;
          if (want && tH >= K_LIVE) {
            r.di = t.tr; r.dj = t.tq; r.score = (int32_t)(tH >> 15) - 65536; r.errors = 32767 - (int32_t)(tH & 32767u); r.reached = 1;
          } else {
            r = lane_best_result(t);
          }
          pre[t.chain].res = r;
          pre[t.chain].valid = 1;
          active = false;
        }
      }
    }
  }
}

// Persistent waves, one per SIMD: every lane runs one request at a time (pga::extend_banded: same cells, checks and
// tie-breaks as extend_wave), in lock-step anti-diagonals, and takes the next request when it is done.  The free
// searches (list A: length unknown, up to 20 000 anti-diagonals) are handed out before the target searches (list B), so
// that the long ones start early.  Register parity: step s of the wave updates the diagonals l = s mod 2, so a lane
// whose band offset is odd simply starts one step later.  Sequence buffers are topped up, dead searches detected and
// free lanes refilled every 32 steps.
__global__ __launch_bounds__(64) void anim_extdp_lane_kernel(const ExtReq* __restrict__ reqs_a, const ExtReq* __restrict__ reqs_b,
                                                             const uint32_t* __restrict__ n_reqs, uint32_t* __restrict__ cursor,
                                                             ExtPre* __restrict__ pre, ExtDump* __restrict__ dumps, uint32_t dump_cap,
                                                             uint32_t* __restrict__ n_dumps, int tail_lanes, int tail_blocks) {
  constexpr int W = BAND / 2;
  constexpr uint32_t K_LIVE = 32768u << 15;
  const int lane = threadIdx.x & 63;
  const uint32_t n_a = n_reqs[0], n_all = n_a + n_reqs[1];
  uint32_t H[64], X[32], Y[32];   // see ExtLaneCell
#pragma unroll
  for (int l = 0; l < 64; ++l) { H[l] = 0; X[l >> 1] = 0; Y[l >> 1] = 0; }
  LaneWin w{0, 0, 0, 0, 0, 0, 0, 0};
  LaneSeq rs{nullptr, nullptr, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0}, qs = rs;
  LaneTask t{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool active = false, drained = false;   // drained (uniform): the lists have been handed out
  int blocks_after = 0;                   // 32-step blocks run since then
  for (;;) {
    // ---- the tail: once the lists are empty a thinly occupied wave (a lane step costs the same for 1 or 64 searches, and
    // one lane needs ~10x the time of a whole wave for the same anti-diagonals) hands its searches over to
    // anim_extend_kernel, state and all; so does any wave after EXT_TAIL_BLOCKS more blocks
    if (drained) {
      const uint64_t am = __ballot(active);
      if (am && (__popcll(am) < tail_lanes || blocks_after >= tail_blocks)) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(n_dumps, (uint32_t)__popcll(am));
        at = (uint32_t)__shfl((int)at, 0, 64);
        if (active) {
          const uint32_t slot = at + (uint32_t)__popcll(am & lanemask_lt());
          if (slot < dump_cap) {
            ExtDump* o = dumps + slot;
#pragma unroll
            for (int l = 0; l < 64; ++l) o->H[l] = H[l];
#pragma unroll
            for (int l = 0; l < 32; ++l) { o->X[l] = X[l]; o->Y[l] = Y[l]; }
            o->best = t.best; o->bpay = t.bpay; o->d = t.d; o->pad_ = 0;
            pre[t.chain].res.di = (int32_t)slot;
            pre[t.chain].valid = 2;
          }   // (no slot left: valid stays 0 and the wave kernel searches from the start)
          active = false;
        }
      }
      ++blocks_after;
    }
    // ---- every 32 steps: buffers, dead searches, refill ----------------------------------------------------------
    if (active) {
      if (rs.cnt <= 16) lane_seq_append(rs);
      if (qs.cnt <= 16) lane_seq_append(qs);
      uint32_t mx = 0;
#pragma unroll
      for (int l = 0; l < 64; ++l) mx = H[l] > mx ? H[l] : mx;
      if (mx < K_LIVE) {   // nothing alive: the best so far stands (as the break of extend_wave)
        pre[t.chain].res = lane_best_result(t);
        pre[t.chain].valid = 1;
        active = false;
      }
    }
    const uint64_t idle = __ballot(!active);
    if (idle && !drained) {   // uniform
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(cursor, (uint32_t)__popcll(idle));
      at = (uint32_t)__shfl((int)at, 0, 64);
      if (at + (uint32_t)__popcll(idle) >= n_all) drained = true;
      const uint32_t mine = at + (uint32_t)__popcll(idle & lanemask_lt());
      int32_t used_r = 0, used_q = 0;
      const bool fresh = !active && mine < n_all;
      if (fresh) {
        const ExtReq q = mine < n_a ? reqs_a[mine] : reqs_b[mine - n_a];
        t.chain = q.chain;
        bool targeted = q.tr >= 0;
        int koff = 0, lt = 0;
        if (targeted) {   // band placement and target test of pga::extend_banded
          koff = (q.tq - q.tr) / 2;
          if (koff > W - 2) koff = W - 2;
          if (koff < -(W - 2)) koff = -(W - 2);
          lt = (q.tq - q.tr) - koff + W;
          if (lt < 0 || lt >= BAND || q.tr > q.rmax || q.tq > q.qmax) { targeted = false; koff = 0; }
        }
        t.koff = koff; t.lt = lt; t.targeted = targeted ? 1 : 0; t.tr = q.tr; t.tq = q.tq;
        t.d = 0; t.d_end = targeted ? q.tr + q.tq : q.rmax + q.qmax;
        t.c1 = 2 * q.rmax - W + koff; t.c2 = 2 * q.qmax + W - koff;
#pragma unroll
        for (int l = 0; l < 64; ++l) { H[l] = (l == W - koff) ? ((65536u << 15) | 32767u) : 0u; X[l >> 1] = 0; Y[l >> 1] = 0; }
        t.best = 65536u << 15; t.bpay = (32767u << 6) | (uint32_t)(W - koff);
        // the two streams (see LaneSeq); query strand position p = stored position len-1-p, complemented
        rs.codes = q.rcodes; rs.mask = q.rmask; rs.len = q.rlen; rs.comp = 0; rs.tmax = q.rmax;
        rs.start = q.dir > 0 ? q.r0 : q.r0 - 1; rs.sgn = q.dir;
        qs.codes = q.qcodes; qs.mask = q.qmask; qs.len = q.qlen; qs.comp = q.strand ? 1u : 0u; qs.tmax = q.qmax;
        if (!q.strand) { qs.start = q.dir > 0 ? q.q0 : q.q0 - 1; qs.sgn = q.dir; }
        else { qs.start = q.dir > 0 ? q.qlen - 1 - q.q0 : q.qlen - q.q0; qs.sgn = -q.dir; }
        rs.bc_lo = rs.bc_hi = rs.bo_lo = rs.bo_hi = 0; rs.cnt = 0; rs.chunk = 0;
        qs.bc_lo = qs.bc_hi = qs.bo_lo = qs.bo_hi = 0; qs.cnt = 0; qs.chunk = 0;
        lane_seq_append(rs); lane_seq_append(rs);
        lane_seq_append(qs); lane_seq_append(qs);
        w = LaneWin{0, 0, 0, 0, 0, 0, 0, 0};
        // elements already inside the windows just before the first anti-diagonal (which pushes one more of its own)
        if (koff & 1) { used_r = (31 - koff) / 2; used_q = (koff + 31) / 2; }
        else { used_r = 16 - koff / 2; used_q = koff / 2 + 15; }
        active = true;
        if (targeted && q.tr == 0 && q.tq == 0) {   // already there
          pre[t.chain].res = ExtResult{0, 0, 0, 0, 1};
          pre[t.chain].valid = 1;
          active = false; used_r = 0; used_q = 0;
        }
      }
      const int32_t roll = (int32_t)wave_max_u32((uint32_t)(used_r > used_q ? used_r : used_q));
      for (int32_t it = 0; it < roll; ++it) {
        if (it < used_r) lane_win_push_ref(w, rs);
        if (it < used_q) lane_win_push_qry(w, qs);
      }
      if (fresh && active) {
        if (rs.cnt <= 16) lane_seq_append(rs);
        if (qs.cnt <= 16) lane_seq_append(qs);
      }
    }
    if (!__any(active)) break;
    // ---- 32 anti-diagonals ---------------------------------------------------------------------------------------------
    const bool near_limit = active && (t.d + 34 - t.c1 > 0 || t.c2 - (t.d + 34) < 63);
    if (__any(near_limit)) ext_lane_block<true>(active, H, X, Y, w, rs, qs, t, pre);
    else ext_lane_block<false>(active, H, X, Y, w, rs, qs, t, pre);
  }
}

// DP call number `round` of every chain of the work list in the given phase, one THREAD per chain: the same policy code
// as anim_extend_kernel, with a DP routine that replays the delivered results of the earlier calls and only writes the
// arguments of this one down — as a request for the lanes (free searches in list A, target searches in list B) and into
// pre[round] for the check in anim_extend_kernel.  valid = 0 until a lane delivers.
__global__ __launch_bounds__(256) void anim_extreq_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units, ClusterOut O,
                                                          const uint2* __restrict__ wl, uint32_t n_wl, ChainFwd* fw, ChainBwd* __restrict__ bw,
                                                          int phase, int round, ExtPre* __restrict__ pre_all, ExtReq* __restrict__ reqs_a,
                                                          ExtReq* __restrict__ reqs_b, uint32_t* __restrict__ n_reqs,
                                                          uint32_t* __restrict__ wave_list) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool final_round = round == EXT_ROUNDS;   // every call answered: finish the chain here; else: list it for anim_extend_kernel
  bool have = false, lost = false;
  ExtArgs a{0, 0, 0, 0, 0, 0, 0};
  ExtReq q;
  if (i < n_wl) {
    const uint32_t u = wl[i].x;
    const int32_t c = (int32_t)wl[i].y;
    const UnitDesc U = units[u];
    const RefDesc R = refs[U.ref];
    const size_t off = O.moff[u];
    const Chain ch = O.chains[off + c];
    int32_t r_lo, r_hi, q_lo, q_hi;
    chain_bounds(R, U, ch, r_lo, r_hi, q_lo, q_hi);
    ChainFwd* fwu = fw + off;
    const Match* cm = O.cm + off;
    int call = 0;
    const auto ext = [&](int32_t cr, int32_t cq, int dir, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
      const ExtArgs now{cr, cq, dir, rmax, qmax, tr, tq};
      const int k = call++;
      if (k < round && !lost) {   // an earlier call: its result, if a lane delivered it
        const ExtPre p = pre_all[(size_t)k * n_wl + i];
        if (p.valid == 1 && p.args.r0 == cr && p.args.q0 == cq && p.args.dir == dir && p.args.rmax == rmax && p.args.qmax == qmax &&
            p.args.tr == tr && p.args.tq == tq)
          return p.res;
        lost = true;
      } else if (k == round && !lost) {
        have = true; a = now;
        if (final_round) lost = true;   // a call the lanes were not asked
      }
      return ExtResult{0, 0, 0, 0, 1};   // ends the policy code quickly
    };
    if (phase == 0) {
      const int32_t er = fwu[c].lr, eq = fwu[c].lq;
      int32_t nr, nq, re, qe, err_fwd, reached;
      const int32_t target = pick_forward_target(O.chains + off, cm, O.next_of + off, c, er, eq, nr, nq);
      forward_extension([&](int32_t cr, int32_t cq, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
                          return ext(cr, cq, +1, rmax, qmax, tr, tq); },
                        er, eq, r_hi, q_hi, nr, nq, re, qe, err_fwd, reached);
      if (final_round && !lost) {   // field-wise, as anim_extend_kernel
        fwu[c].re = re; fwu[c].qe = qe; fwu[c].err_fwd = err_fwd; fwu[c].reached = reached; fwu[c].target = target;
      }
    } else {
      const int32_t p = O.prev_of[off + c];
      const int32_t first_r = fwu[c].first_r, first_q = fwu[c].first_q;
      const int32_t prev_re = p >= 0 ? fwu[p].re : -1, prev_qe = p >= 0 ? fwu[p].qe : -1;
      const bool shadowed = p >= 0 && ((fwu[p].reached && fwu[p].target == c) ||
                                       (fwu[p].first_r <= first_r && fwu[p].first_q <= first_q && prev_re >= fwu[c].lr && prev_qe >= fwu[c].lq));
      if (!shadowed) {
        int32_t tr = -1, tq = -1;
        if (prev_re >= 0 && first_r >= prev_re && first_q >= prev_qe) { tr = first_r - prev_re; tq = first_q - prev_qe; }
        if (p >= 0) {
          const int32_t plr = fwu[p].lr, plq = fwu[p].lq;
          if (plr <= first_r && plq <= first_q) {
            if (plr > r_lo) r_lo = plr;
            if (plq > q_lo) q_lo = plq;
          }
        }
        const int32_t rmax = cap_ext(first_r - r_lo, MAX_EXT_BWD), qmax = cap_ext(first_q - q_lo, MAX_EXT_BWD);
        ExtResult b = ext(first_r, first_q, -1, rmax, qmax, tr, tq);
        if (tr >= 0 && !b.reached && tr != tq) b = ext(first_r, first_q, -1, rmax, qmax, -1, -1);
        if (final_round && !lost) {   // as anim_extend_kernel; a junction that needs the rectangle DP goes there
          ChainBwd e;
          e.rs = first_r - b.di; e.qs = first_q - b.dj; e.err_back = b.errors;
          e.reached = (tr >= 0 && b.reached) ? 1 : 0;
          bridge_junction(e, prev_re, prev_qe, tr, tq, first_r, first_q, p >= 0 ? fwu[p].lr : -1, p >= 0 ? fwu[p].lq : -1,
                          p >= 0 ? fwu[p].err_fwd : 0, [&](int32_t, int32_t, int32_t, int32_t) { lost = true; return -1; });
          if (!lost) bw[off + c] = e;
        }
      } else if (final_round) {
        bw[off + c] = ChainBwd{first_r, first_q, 0, 0};   // will be shadowed
      }
    }
    if (final_round) {
      have = lost;   // "have" now means: has work for the wave kernel
    } else {
      have = have && !lost;
      ExtPre* mine = pre_all + (size_t)round * n_wl + i;
      mine->args = a;
      mine->valid = 0;
      q.rcodes = R.codes; q.rmask = R.mask; q.qcodes = U.codes; q.qmask = U.mask;
      q.rlen = (int32_t)R.len; q.qlen = (int32_t)U.len; q.strand = U.strand; q.dir = a.dir;
      q.r0 = a.r0; q.q0 = a.q0; q.rmax = a.rmax; q.qmax = a.qmax; q.tr = a.tr; q.tq = a.tq;
      q.chain = i; q.pad_ = 0;
    }
  }
  if (final_round) {
    const uint64_t mw = __ballot(have);
    if (mw) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&n_reqs[0], (uint32_t)__popcll(mw));
      at = (uint32_t)__shfl((int)at, 0, 64);
      if (have) wave_list[at + (uint32_t)__popcll(mw & lanemask_lt())] = i;
    }
    return;
  }
  // the lane kernel decides "free search" exactly like pga::extend_banded; here only the order of the hand-out depends on it
  const bool free_search = have && (a.tr < 0 || a.tr > a.rmax || a.tq > a.qmax);
  const uint64_t ma = __ballot(free_search), mb = __ballot(have && !free_search);
  if (ma) {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&n_reqs[0], (uint32_t)__popcll(ma));
    at = (uint32_t)__shfl((int)at, 0, 64);
    if (free_search) reqs_a[at + (uint32_t)__popcll(ma & lanemask_lt())] = q;
  }
  if (mb) {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&n_reqs[1], (uint32_t)__popcll(mb));
    at = (uint32_t)__shfl((int)at, 0, 64);
    if (have && !free_search) reqs_b[at + (uint32_t)__popcll(mb & lanemask_lt())] = q;
  }
}

// The gaps too large for anim_gapdp_lane_kernel (a side > 63) are target searches like any other: anim_gapreq_kernel turns
// the entries of their list into requests for anim_extdp_lane_kernel (one THREAD per entry; a target outside the band
// is not worth a search, pga::gap_errors counts the diagonal then), anim_gapdp_kernel picks the results up.
__global__ __launch_bounds__(256) void anim_gapreq_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                          const GapTask* __restrict__ tasks, const uint32_t* __restrict__ list, uint32_t n,
                                                          ExtPre* __restrict__ pre, ExtReq* __restrict__ reqs_b, uint32_t* __restrict__ n_reqs) {
  constexpr int W = BAND / 2;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool have = false;
  ExtReq q;
  if (i < n) {
    const GapTask T = tasks[list[i]];
    const UnitDesc U = units[T.unit];
    const RefDesc R = refs[U.ref];
    int koff = (T.m - T.n) / 2;
    if (koff > W - 2) koff = W - 2;
    if (koff < -(W - 2)) koff = -(W - 2);
    const int lt = (T.m - T.n) - koff + W;
    have = lt >= 0 && lt < BAND;
    pre[i].args = ExtArgs{T.r0, T.q0, +1, T.n, T.m, T.n, T.m};
    pre[i].valid = have ? 0 : 3;   // 3: no search needed
    q.rcodes = R.codes; q.rmask = R.mask; q.qcodes = U.codes; q.qmask = U.mask;
    q.rlen = (int32_t)R.len; q.qlen = (int32_t)U.len; q.strand = U.strand; q.dir = +1;
    q.r0 = T.r0; q.q0 = T.q0; q.rmax = T.n; q.qmax = T.m; q.tr = T.n; q.tq = T.m;
    q.chain = i; q.pad_ = 0;
  }
  const uint64_t mb = __ballot(have);
  if (mb) {
    uint32_t at = 0;
    if (lane == 0) at = atomicAdd(&n_reqs[1], (uint32_t)__popcll(mb));
    at = (uint32_t)__shfl((int)at, 0, 64);
    if (have) reqs_b[at + (uint32_t)__popcll(mb & lanemask_lt())] = q;
  }
}

// One wave per large gap at a time (grid-stride over their list): the lanes' result, the rest of a search they handed
// over, or the whole of pga::gap_errors.
__global__ __launch_bounds__(64) void anim_gapdp_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                        ClusterOut O, const GapTask* __restrict__ tasks, const uint32_t* __restrict__ list,
                                                        uint32_t n, const ExtPre* __restrict__ pre, const ExtDump* __restrict__ dumps,
                                                        ChainFwd* __restrict__ fw) {
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const GapTask T = tasks[list[i]];
    const UnitDesc U = units[T.unit];
    const RefDesc R = refs[U.ref];
    const SeqView RV{R.codes, R.mask, R.len};
    const StrandView QV{SeqView{U.codes, U.mask, U.len}, U.strand};
    const ExtPre pr = pre[i];
    int32_t err;
    if (pr.valid == 0) {
      err = gap_errors_wave(RV, QV, T.r0, T.n, T.q0, T.m);
    } else {
      ExtResult e{0, 0, 0, 0, 0};
      if (pr.valid == 1) e = pr.res;
      else if (pr.valid == 2) e = extend_wave(RV, QV, T.r0, T.q0, +1, T.n, T.m, T.n, T.m, dumps + pr.res.di);
      if (e.reached) {
        err = e.errors;
      } else {   // target outside the band or pruned: the diagonal part + the length difference (pga::gap_errors)
        const int32_t kq = T.n < T.m ? T.n : T.m;
        int32_t part = 0;
        for (int32_t t = threadIdx.x & 63; t < kq; t += 64)
          part += (RV.clean(T.r0 + t) && QV.clean(T.q0 + t) && RV.base(T.r0 + t) == QV.base(T.q0 + t)) ? 0 : 1;
        err = wave_sum32(part) + (T.n > T.m ? T.n - T.m : T.m - T.n);
      }
    }
    if ((threadIdx.x & 63) == 0 && err) atomicAdd(&fw[O.moff[T.unit] + T.chain].inner_err, err);
  }
}

// One WAVE per chain (work list wl: unit, chain).  phase 0: forward extension off the last match (the rest of
// pga::extend_chain_fwd is anim_gaps_kernel + anim_gapdp_kernel); phase 1: backward extension towards the previous
// chain's forward end (extend_chain_bwd).
__global__ __launch_bounds__(64) void anim_extend_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                         ClusterOut O, const uint2* __restrict__ wl,
                                                         ChainFwd* __restrict__ fw, ChainBwd* __restrict__ bw, int phase,
                                                         const ExtPre* __restrict__ pre, uint32_t n_wl, const ExtDump* __restrict__ dumps,
                                                         const uint32_t* __restrict__ wave_list, const uint32_t* __restrict__ n_list,
                                                         uint32_t* __restrict__ cursor) {
  // persistent waves over the chains anim_extreq_kernel's final round could not finish
  const uint32_t n_todo = *n_list;
  for (;;) {
  uint32_t todo = 0;
  if ((threadIdx.x & 63) == 0) todo = atomicAdd(cursor, 1u);
  todo = (uint32_t)__shfl((int)todo, 0, 64);
  if (todo >= n_todo) return;
  const uint32_t ci = wave_list[todo];
  const uint32_t u = wl[ci].x;
  const int32_t c = (int32_t)wl[ci].y;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const SeqView RV{R.codes, R.mask, R.len};
  const StrandView QV{SeqView{U.codes, U.mask, U.len}, U.strand};
  // the chain's first DP calls of the phase may have been solved by a lane of anim_extdp_lane_kernel already
  int call = 0;
  const auto ext = [&](int32_t cr, int32_t cq, int dir, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
    const int k = call++;
    if (k < EXT_ROUNDS) {   // uniform
      const ExtPre pr = pre[(size_t)k * n_wl + ci];
      const bool same = pr.args.r0 == cr && pr.args.q0 == cq && pr.args.dir == dir && pr.args.rmax == rmax && pr.args.qmax == qmax &&
                        pr.args.tr == tr && pr.args.tq == tq;
#ifdef PGA_DP_STATS
      if ((threadIdx.x & 63) == 0) atomicAdd(&g_dp_stats[dir < 0 ? 2 : 1][pr.valid == 1 && same ? 3 : pr.valid == 2 && same ? 6 : !same ? 5 : 4], 1ull);
#endif
      if (pr.valid == 1 && same) return pr.res;
      if (pr.valid == 2 && same) return extend_wave(RV, QV, cr, cq, dir, rmax, qmax, tr, tq, dumps + pr.res.di);   // handed over mid-way
    }
    return ExtResult{0, 0, 0, 0, -1};   // not delivered
  };
  const size_t off = O.moff[u];
  const Chain ch = O.chains[off + c];
  int32_t r_lo, r_hi, q_lo, q_hi;
  chain_bounds(R, U, ch, r_lo, r_hi, q_lo, q_hi);
  ChainFwd* fwu = fw + off;
  const Match* cm = O.cm + off;
  if (phase == 0) {
    ChainFwd e = fwu[c];   // first match, last match end and gap errors come from anim_gaps_kernel / anim_gapdp_kernel
    const int32_t er = e.lr, eq = e.lq;
    int32_t nr, nq, re, qe, err_fwd, reached;
    const int32_t target = pick_forward_target(O.chains + off, cm, O.next_of + off, c, er, eq, nr, nq);
    forward_extension([&](int32_t cr, int32_t cq, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
                        const ExtResult x = ext(cr, cq, +1, rmax, qmax, tr, tq);
                        return x.reached >= 0 ? x : extend_wave(RV, QV, cr, cq, +1, rmax, qmax, tr, tq); },
                      er, eq, r_hi, q_hi, nr, nq, re, qe, err_fwd, reached);
    if ((threadIdx.x & 63) == 0) {   // field-wise: inner_err may still be receiving atomics from the gap DP kernel
      fwu[c].re = re; fwu[c].qe = qe; fwu[c].err_fwd = err_fwd; fwu[c].reached = reached; fwu[c].target = target;
    }
  } else {
    const int32_t p = O.prev_of[off + c];
    const int32_t first_r = fwu[c].first_r, first_q = fwu[c].first_q;
    const int32_t prev_re = p >= 0 ? fwu[p].re : -1, prev_qe = p >= 0 ? fwu[p].qe : -1;
    if (p >= 0 && ((fwu[p].reached && fwu[p].target == c) ||
                   (fwu[p].first_r <= first_r && fwu[p].first_q <= first_q && prev_re >= fwu[c].lr && prev_qe >= fwu[c].lq))) {
      if ((threadIdx.x & 63) == 0) bw[off + c] = ChainBwd{first_r, first_q, 0, 0};  // will be shadowed
      continue;
    }
    int32_t tr = -1, tq = -1;
    if (prev_re >= 0 && first_r >= prev_re && first_q >= prev_qe) { tr = first_r - prev_re; tq = first_q - prev_qe; }
    if (p >= 0) {   // never search into the previous chain's matches (same rule as pga::extend_chain_bwd)
      const int32_t plr = fwu[p].lr, plq = fwu[p].lq;
      if (plr <= first_r && plq <= first_q) {   // collinear predecessor only
        if (plr > r_lo) r_lo = plr;
        if (plq > q_lo) q_lo = plq;
      }
    }
    ExtResult b = ext(first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), tr, tq);
    if (b.reached < 0)
      b = extend_wave(RV, QV, first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), tr, tq);
    if (tr >= 0 && !b.reached && tr != tq) {   // shifted band, unreachable target: search freely (as pga::extend_chain_bwd)
      b = ext(first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), -1, -1);
      if (b.reached < 0)
        b = extend_wave(RV, QV, first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), -1, -1);
    }
    ChainBwd e;
    e.rs = first_r - b.di; e.qs = first_q - b.dj; e.err_back = b.errors;
    e.reached = (tr >= 0 && b.reached) ? 1 : 0;
    bridge_junction(e, prev_re, prev_qe, tr, tq, first_r, first_q, p >= 0 ? fwu[p].lr : -1, p >= 0 ? fwu[p].lq : -1,
                    p >= 0 ? fwu[p].err_fwd : 0,
                    [&](int32_t r0, int32_t n, int32_t q0, int32_t m) { return thin_rect_errors_wave(RV, QV, r0, n, q0, m); });
    if ((threadIdx.x & 63) == 0) bw[off + c] = e;
  }
  }
}

struct FinishScratch {   // per-alignment arrays: pair p owns the slice [moff[2p], moff[2p+2])
  Aln* alns;
  int32_t* a_rrec;
  int32_t* a_qrec;
  int32_t* idx;
  int32_t* from;
  double* sc;
  int32_t* aln_of;  // per chain
};

// pga::lis_filter with the O(n^2) look-back spread over the lanes of one wave.  Every lane runs the (cheap) serial parts
// redundantly and writes identical values, so no intra-wave memory ordering is needed; candidates are evaluated by the
// same expression as the scalar code and the reduction keeps its tie rule (the earliest predecessor reaching the max).
__device__ void lis_filter_wave(Aln* a, int n, int side, const int32_t* grp, int32_t* idx, double* sc_, int32_t* from) {
  const int lane = threadIdx.x & 63;
  int64_t* sc = reinterpret_cast<int64_t*>(sc_);
  auto lo = [&](int i) { return side == 0 ? a[i].rs : a[i].qs; };
  auto hi = [&](int i) { return side == 0 ? a[i].re : a[i].qe; };
  bool ok;
  for (int i = 0; i < n; ++i) { idx[i] = i; sc[i] = lis_gain(hi(i) - lo(i), 1, 0, lis_idy(a[i]), ok); }
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
      const int32_t lo_i = lo(i);
      long long bc = sc[i];   // own score; this lane's best candidate and its predecessor's rank (kk)
      int bk = 0x7FFFFFFF;
      for (int kk = g0 + lane; kk < k; kk += 64) {
        const int j = idx[kk];
        int64_t ol = (int64_t)hi(j) - lo_i;
        if (ol < 0) ol = 0;
        bool allowed;
        const int64_t g = lis_gain(len, hi(j) - lo(j), ol, idy, allowed);
        if (!allowed) continue;
        const long long cand = sc[j] + g;
        if (cand > bc) { bc = cand; bk = kk; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const long long oc = __shfl_xor(bc, o, 64);
        const int ok2 = __shfl_xor(bk, o, 64);
        if (oc > bc || (oc == bc && ok2 < bk)) { bc = oc; bk = ok2; }
      }
      sc[i] = bc;
      from[i] = bk == 0x7FFFFFFF ? -1 : idx[bk];
      if (best < 0 || sc[i] > sc[best]) best = i;
    }
    for (int i = best; i >= 0; i = from[i]) a[i].keep |= (1 << side);
    g0 = g1;
  }
}

// One WAVE per ordered pair: stitch both strands' chains into alignments, 1-to-1 filter, parse_delta reduction.
__global__ __launch_bounds__(64) void anim_finish_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units, uint32_t n_pairs,
                                                         ClusterOut O, const ChainFwd* __restrict__ fw, const ChainBwd* __restrict__ bw,
                                                         FinishScratch S, int filter_1to1, pg_anim_result* __restrict__ out) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pairs) return;
  const RefDesc R = refs[units[2 * p].ref];
  const size_t poff = O.moff[2 * p];
  const int cap_a = (int)(O.moff[2 * p + 2] - O.moff[2 * p]);
  Aln* alns = S.alns + poff;
  int32_t* a_rrec = S.a_rrec + poff;
  int32_t* a_qrec = S.a_qrec + poff;
  int32_t* idx = S.idx + poff;
  int32_t* from = S.from + poff;
  double* sc = S.sc + poff;
  int n = 0;
  for (int strand = 0; strand < 2; ++strand) {
    const uint32_t u = 2 * p + strand;  // units are laid out pair-major: (pair, fwd), (pair, rev)
    const UnitDesc U = units[u];
    const size_t off = O.moff[u];
    const int before = n;
    n = stitch_chains(fw + off, bw + off, O.cm + off, O.chains + off, O.order + off, O.prev_of + off, O.next_of + off,
                      O.n_chains[u], strand, S.aln_of + off, alns, n, cap_a);
    for (int i = before; i < n; ++i) {
      Aln& a = alns[i];
      a_rrec[i] = record_of(R.rec_start, R.n_rec, a.rs);
      if (strand) { const int32_t qs = U.len - a.qe, qe = U.len - a.qs; a.qs = qs; a.qe = qe; }  // forward coordinates
      a_qrec[i] = record_of(U.rec_start, U.n_rec, a.qs);
    }
  }
  if (filter_1to1) {
    lis_filter_wave(alns, n, 0, a_rrec, idx, sc, from);
    lis_filter_wave(alns, n, 1, a_qrec, idx, sc, from);
  } else {
    for (int i = 0; i < n; ++i) alns[i].keep = 3;
  }
  const PairResult r = reduce_pair(alns, n, a_rrec, a_qrec, idx);
  pg_anim_result o;
  o.ref_aln_len = r.ref_aln_len;
  o.qry_aln_len = r.qry_aln_len;
  o.sim_errors = r.sim_errors;
  o.n_alignments = r.n_alignments;
  o.identity = r.aligned > 0 ? (double)r.weighted / (double)r.aligned : 0.0;  // int/int true division (anim.py:396)
  o.status = O.status[p] ? PG_E_CAPACITY : (r.n_alignments == 0 ? PG_ANIM_NO_ALIGNMENT : 0);
  o.reserved = n;   // alignments before the filter
  if ((threadIdx.x & 63) == 0) out[p] = o;
}

// reduction of caller-supplied alignment records (pg_anim_reduce): one thread per pair
__global__ __launch_bounds__(64) void anim_reduce_kernel(uint32_t n_pairs, const uint64_t* __restrict__ offsets, Aln* alns,
                                                         const int32_t* __restrict__ rgrp, const int32_t* __restrict__ qgrp,
                                                         int32_t* idx, int32_t* from, double* sc, int apply_filter,
                                                         pg_anim_result* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const uint64_t o = offsets[p];
  const int n = (int)(offsets[p + 1] - o);
  if (apply_filter) {
    lis_filter(alns + o, n, 0, rgrp + o, idx + o, sc + o, from + o);
    lis_filter(alns + o, n, 1, qgrp + o, idx + o, sc + o, from + o);
  }
  const PairResult r = reduce_pair(alns + o, n, rgrp + o, qgrp + o, idx + o);
  pg_anim_result res;
  res.ref_aln_len = r.ref_aln_len; res.qry_aln_len = r.qry_aln_len; res.sim_errors = r.sim_errors;
  res.n_alignments = r.n_alignments;
  res.identity = r.aligned > 0 ? (double)r.weighted / (double)r.aligned : 0.0;
  res.status = r.n_alignments == 0 ? PG_ANIM_NO_ALIGNMENT : 0;
  res.reserved = 0;
  out[p] = res;
}

template <typename T>
int anim_alloc(pg_ctx* ctx, T*& p, size_t n) {
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
  return PG_OK;
}

}  // namespace

// ---- host driver ---------------------------------------------------------------------------------------------------
// One batch of ordered pairs (any mix of references): ref_ids[i] = nucmer's reference (pyani's query genome, anim.py:280).
// Scratch lives in the context and only grows.  The per-unit kernels are latency-bound single-thread code, so the
// batch should be as large as memory allows: thousands of units in flight are what fills the GPU.
namespace {
struct AnimScratch {
  size_t units = 0, pairs = 0, refs = 0, recs = 0, wl = 0, matches = 0;
  // per-genome seed lists (built once per resident genome and role, dropped by pg_clear_genomes)
  struct GenomeIdx {
    uint64_t *ref_list = nullptr, *qry_list = nullptr;
    uint32_t *ref_goff = nullptr, *qry_goff = nullptr;
    uint32_t ref_max = 0;   // largest reference group (sizes the LDS table)
  };
  std::vector<GenomeIdx> gidx;
  uint32_t* list_cnt = nullptr;   // 2 * SEED_GROUPS counters shared by the list builds
  SeedRef* srefs_d = nullptr;
  SeedQry* sqry_d = nullptr;
  SeedSlice* slice_d = nullptr;   // [SEED_GROUPS][pairs of the batch]
  size_t slice_pairs = 0;
  int32_t* recs_d = nullptr;
  RefDesc* refs_d = nullptr;
  UnitDesc* units_d = nullptr;
  uint32_t *mem_count = nullptr, *moff = nullptr, *choff_d = nullptr;
  int32_t *nch = nullptr, *status = nullptr;
  pg_anim_result* out = nullptr;
  // per-match arrays (sliced by moff)
  Match *mem = nullptr, *cm = nullptr;
  int32_t *iscratch = nullptr, *order = nullptr, *prev = nullptr, *next = nullptr, *alnof = nullptr;
  Chain* chains = nullptr;
  ChainFwd* fw = nullptr;
  ChainBwd* bw = nullptr;
  FinishScratch S{};
  uint2* wl_d = nullptr;
  GapTask* tasks_d = nullptr; // gaps that need the DP (at most one per match)
  size_t tasks = 0;
  Match* seedbuf = nullptr;   // batch-wide append buffer of the seed pass
  size_t seed_cap = 0;
  uint32_t* seed_total = nullptr;   // [0] matches appended, [1] hits recorded
  uint32_t* gap_counts = nullptr;   // gap tasks per size class + the wave list
  ExtReq* ext_reqs = nullptr;       // DP requests for the lanes: free searches, then (at n_wl) target searches
  ExtPre* ext_pre = nullptr;        // [EXT_ROUNDS][n_wl] arguments of the chains' first DP calls and the delivered results
  ExtDump* ext_dumps = nullptr;     // searches the lanes hand over to the wave kernel mid-way
  uint32_t* ext_wave = nullptr;     // chains left for the wave kernel of a phase
  uint32_t* ext_counts = nullptr;   // request list lengths, hand-out cursor, hand-over count of a lane launch
  size_t ext_cap = 0;
  uint8_t* task_cls = nullptr;      // size class of the GapTask in every match slot (0xFF = none)
  uint32_t* task_lists = nullptr;   // [GAP_CLASSES + 1][slots] slot lists by class
  Match* hits_d = nullptr;          // hits recorded by the probe kernel for anim_hit_kernel
  Match* hits_sorted = nullptr;     // the same, dealt into per-unit slices (hoff)
  uint32_t *hit_count = nullptr, *hoff = nullptr, *hit_cursor = nullptr;   // per unit
  size_t hit_cap = 0;
};

template <typename T>
int regrow(pg_ctx* ctx, T*& p, size_t n) {
  if (p) PG_HIP(ctx, hipFree(p));
  p = nullptr;
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)));
  return PG_OK;
}
}  // namespace

static AnimScratch* anim_scratch(pg_ctx* ctx) {
  if (!ctx->anim_scratch) ctx->anim_scratch = new AnimScratch();
  return static_cast<AnimScratch*>(ctx->anim_scratch);
}

void pg_anim_drop_lists(pg_ctx* ctx) {
  AnimScratch* A = static_cast<AnimScratch*>(ctx->anim_scratch);
  if (!A) return;
  for (auto& g : A->gidx) {
    void* ptrs[] = {g.ref_list, g.qry_list, g.ref_goff, g.qry_goff};
    for (void* p : ptrs) if (p) (void)hipFree(p);
  }
  A->gidx.clear();
}

// Build the seed lists the batch needs and does not have yet: reference role for `ref_genomes`, query role for `qry_genomes`.
static int anim_ensure_lists(pg_ctx* ctx, AnimScratch* A, const std::vector<int32_t>& ref_genomes, const std::vector<int32_t>& qry_genomes) {
  int rc;
  if (A->gidx.size() < ctx->genomes.size()) A->gidx.resize(ctx->genomes.size());
  if (!A->list_cnt && (rc = regrow(ctx, A->list_cnt, (size_t)2 * SEED_GROUPS))) return rc;
  std::vector<int32_t> fresh_refs;
  for (int role = 0; role < 2; ++role) {
    for (int32_t gid : role ? qry_genomes : ref_genomes) {
      AnimScratch::GenomeIdx& X = A->gidx[gid];
      if (role ? X.qry_list != nullptr : X.ref_list != nullptr) continue;
      const PgGenome& G = ctx->genomes[gid];
      const int32_t len = (int32_t)G.stream_len;
      const uint32_t n_sub = role ? 2 * SEED_GROUPS : SEED_GROUPS;
      const size_t bound = role ? 2 * ((size_t)len / SEED_STEP + 1) : (size_t)len + 1;
      uint64_t*& list = role ? X.qry_list : X.ref_list;
      uint32_t*& goff = role ? X.qry_goff : X.ref_goff;
      if ((rc = regrow(ctx, list, bound))) return rc;
      if ((rc = regrow(ctx, goff, (size_t)n_sub + 2))) return rc;
      const uint32_t* codes = ctx->d_codes + G.arena_start / 16;
      const uint32_t* mask = ctx->d_mask + G.arena_start / 32;
      const int32_t n_idx = role ? len / SEED_STEP + 1 : len;
      const dim3 grid((uint32_t)(n_idx + LIST_CHUNK - 1) / LIST_CHUNK, role ? 2 : 1);
      PG_HIP(ctx, hipMemsetAsync(A->list_cnt, 0, (size_t)n_sub * 4, ctx->stream));
      if (grid.x)   // (an empty genome still gets its all-zero offset table from the scan)
        hipLaunchKernelGGL(anim_list_kernel, grid, dim3(LIST_BLOCK), 0, ctx->stream, codes, mask, len, role, A->list_cnt,
                           (const uint32_t*)nullptr, (uint64_t*)nullptr, 0);
      hipLaunchKernelGGL(anim_list_scan_kernel, dim3(1), dim3(64), 0, ctx->stream, A->list_cnt, goff, n_sub);
      if (grid.x)
        hipLaunchKernelGGL(anim_list_kernel, grid, dim3(LIST_BLOCK), 0, ctx->stream, codes, mask, len, role, A->list_cnt,
                           (const uint32_t*)goff, list, 1);
      if (!role) fresh_refs.push_back(gid);
    }
  }
  PG_HIP(ctx, hipGetLastError());
  if (!fresh_refs.empty()) {
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int32_t gid : fresh_refs)
      PG_HIP(ctx, hipMemcpy(&A->gidx[gid].ref_max, A->gidx[gid].ref_goff + SEED_GROUPS + 1, 4, hipMemcpyDeviceToHost));
  }
  return PG_OK;
}

void pg_anim_free_scratch(pg_ctx* ctx) {
  AnimScratch* A = static_cast<AnimScratch*>(ctx->anim_scratch);
  if (!A) return;
  pg_anim_drop_lists(ctx);
  void* ptrs[] = {A->ext_reqs, A->ext_pre, A->ext_dumps, A->ext_wave, A->ext_counts, A->gap_counts, A->task_cls, A->task_lists, A->hits_sorted, A->hit_count, A->hoff, A->hit_cursor, A->hits_d, A->slice_d, A->choff_d, A->list_cnt, A->srefs_d, A->sqry_d, A->recs_d, A->refs_d, A->units_d, A->mem_count, A->moff, A->nch, A->status, A->out, A->mem, A->cm,
                  A->iscratch, A->order, A->prev, A->next, A->alnof, A->chains, A->fw, A->bw, A->S.alns, A->S.a_rrec,
                  A->S.a_qrec, A->S.idx, A->S.from, A->S.sc, A->wl_d, A->seedbuf, A->seed_total, A->tasks_d};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete A;
  ctx->anim_scratch = nullptr;
}

// One batch of ordered pairs (ref_ids grouped).  The seed pass appends every unit's matches to one buffer and counts them
// per (pair, strand) unit; a scatter then gives every per-match array exactly the slice it needs, which is what lets
// thousands of units be in flight at once within the HBM budget.
// If the batch needs more than max_matches, only its first n_done pairs are processed (the caller continues from there).
int pg_anim_run_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, int filter_1to1, int maxmatch,
                      uint64_t max_matches, pg_anim_result* out_host, uint32_t* n_done) {
  AnimScratch* A = anim_scratch(ctx);
  uint32_t n_units = 2 * n_pairs;
  int rc;
  (void)hipGetLastError();   // launch checks below must only see this batch's errors
  std::vector<int32_t> ref_list;
  std::vector<uint32_t> ref_of_pair(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (ref_list.empty() || ref_list.back() != ref_ids[p]) ref_list.push_back(ref_ids[p]);
    ref_of_pair[p] = (uint32_t)ref_list.size() - 1;
  }
  const uint32_t n_refs = (uint32_t)ref_list.size();
  std::vector<RefDesc> refs(n_refs);
  std::vector<int32_t> recs;
  std::vector<uint32_t> ref_rec_off(n_refs), qry_rec_off(n_pairs);
  int32_t max_rlen = 0, max_qlen = 0;
  for (uint32_t r = 0; r < n_refs; ++r) {
    const PgGenome& G = ctx->genomes[ref_list[r]];
    refs[r].codes = ctx->d_codes + G.arena_start / 16;
    refs[r].mask = ctx->d_mask + G.arena_start / 32;
    refs[r].len = (int32_t)G.stream_len;
    refs[r].n_rec = (int32_t)G.n_rec;
    ref_rec_off[r] = (uint32_t)recs.size();
    recs.insert(recs.end(), G.rec_start.begin(), G.rec_start.end());
    if ((int32_t)G.stream_len > max_rlen) max_rlen = (int32_t)G.stream_len;
  }
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    qry_rec_off[p] = (uint32_t)recs.size();
    recs.insert(recs.end(), Q.rec_start.begin(), Q.rec_start.end());
    if ((int32_t)Q.stream_len > max_qlen) max_qlen = (int32_t)Q.stream_len;
  }
  if (recs.size() > A->recs) { if ((rc = regrow(ctx, A->recs_d, recs.size()))) return rc; A->recs = recs.size(); }
  if (n_refs > A->refs) {
    if ((rc = regrow(ctx, A->refs_d, n_refs))) return rc;
    if ((rc = regrow(ctx, A->srefs_d, n_refs))) return rc;
    A->refs = n_refs;
  }
  if (n_units > A->units) {
    if ((rc = regrow(ctx, A->units_d, n_units))) return rc;
    if ((rc = regrow(ctx, A->mem_count, n_units))) return rc;
    if ((rc = regrow(ctx, A->moff, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->choff_d, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->hit_count, n_units))) return rc;
    if ((rc = regrow(ctx, A->hoff, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->hit_cursor, n_units))) return rc;
    if ((rc = regrow(ctx, A->nch, n_units))) return rc;
    A->units = n_units;
  }
  if (n_pairs > A->pairs) {
    if ((rc = regrow(ctx, A->status, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->sqry_d, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->out, n_pairs))) return rc;
    A->pairs = n_pairs;
  }
  for (uint32_t r = 0; r < n_refs; ++r) refs[r].rec_start = A->recs_d + ref_rec_off[r];
  std::vector<UnitDesc> units(n_units);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    for (int s = 0; s < 2; ++s) {
      UnitDesc& U = units[2 * p + s];
      U.codes = ctx->d_codes + Q.arena_start / 16;
      U.mask = ctx->d_mask + Q.arena_start / 32;
      U.len = (int32_t)Q.stream_len;
      U.rec_start = A->recs_d + qry_rec_off[p];
      U.n_rec = (int32_t)Q.n_rec;
      U.strand = s;
      U.pair = (int32_t)p;
      U.ref = (int32_t)ref_of_pair[p];
    }
  }
  PG_HIP(ctx, hipMemcpyAsync(A->recs_d, recs.data(), recs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(A->refs_d, refs.data(), n_refs * sizeof(RefDesc), hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(A->units_d, units.data(), n_units * sizeof(UnitDesc), hipMemcpyHostToDevice, ctx->stream));
  // seeding: LDS-resident reference groups, streamed query groups; one pass appends (unit, match) records and the
  // per-unit counts it leaves are exact even if the buffer overflowed
  {
    std::vector<int32_t> qry_list(qry_ids, qry_ids + n_pairs);
    std::sort(qry_list.begin(), qry_list.end());
    qry_list.erase(std::unique(qry_list.begin(), qry_list.end()), qry_list.end());
    if ((rc = anim_ensure_lists(ctx, A, ref_list, qry_list))) return rc;
  }
  uint32_t max_group = 1;
  for (uint32_t r = 0; r < n_refs; ++r) if (A->gidx[ref_list[r]].ref_max > max_group) max_group = A->gidx[ref_list[r]].ref_max;
  uint32_t slots = 256;
  while (slots < 2 * max_group) slots <<= 1;
  if (slots > SEED_MAX_SLOTS)
    return pg_fail(ctx, PG_E_CAPACITY, "anim seeding: a reference k-mer group does not fit the LDS table (genome too large or too repetitive)");
  std::vector<SeedRef> srefs(n_refs);
  std::vector<SeedQry> sqry(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) sqry[p] = SeedQry{A->gidx[qry_ids[p]].qry_list, A->gidx[qry_ids[p]].qry_goff};
  auto fill_srefs = [&](uint32_t limit) {
    for (uint32_t r = 0; r < n_refs; ++r) srefs[r] = SeedRef{A->gidx[ref_list[r]].ref_list, A->gidx[ref_list[r]].ref_goff, 0, 0};
    for (uint32_t p = 0; p < limit; ++p) {
      SeedRef& S = srefs[ref_of_pair[p]];
      if (S.pair_end == 0) S.pair_begin = p;
      S.pair_end = p + 1;
    }
  };
  PG_HIP(ctx, hipMemcpyAsync(A->sqry_d, sqry.data(), n_pairs * sizeof(SeedQry), hipMemcpyHostToDevice, ctx->stream));
  if (n_pairs > A->slice_pairs) {
    if ((rc = regrow(ctx, A->slice_d, (size_t)n_pairs * SEED_GROUPS))) return rc;
    A->slice_pairs = n_pairs;
  }
  const uint32_t slice_stride = n_pairs;   // the table is laid out for the whole batch even if only a prefix is seeded again
  hipLaunchKernelGGL(anim_slice_kernel, dim3(n_pairs), dim3(256), 0, ctx->stream, A->sqry_d, n_pairs, A->slice_d);
  static bool lds_attr_set = false;
  if (!lds_attr_set) {
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(anim_seed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(SEED_MAX_SLOTS * 8 + SEED_STAGE_BYTES)));
    lds_attr_set = true;
  }
  if (!A->seedbuf) {
    A->seed_cap = (size_t)max_matches + 1024;   // the whole batch budget (2.4 GB by default): no overflow re-runs
    if ((rc = regrow(ctx, A->seedbuf, A->seed_cap))) return rc;
    if ((rc = regrow(ctx, A->seed_total, 2))) return rc;
    if ((rc = regrow(ctx, A->gap_counts, GAP_CLASSES + 1))) return rc;
  }
  std::vector<uint32_t> cnt(n_units), moff;
  uint32_t total = 0, pairs_fit = 0;
  // hit buffer: the matches of the batch budget plus the chance 16-mer hits of unrelated pairs (~1200 per 5 Mb unit)
  {
    const size_t want = (size_t)max_matches + (size_t)4096 * n_units + 1024;
    if (want > A->hit_cap) {
      if ((rc = regrow(ctx, A->hits_d, want))) return rc;
      if ((rc = regrow(ctx, A->hits_sorted, want))) return rc;
      A->hit_cap = want;
    }
  }
  for (int attempt = 0;; ++attempt) {
    if (attempt == 8) return pg_fail(ctx, PG_E_CAPACITY, "anim seeding: buffers still overflow after repeated splitting");
    uint32_t counts[2] = {0, 0};   // matches appended, hits recorded
    fill_srefs(n_pairs);
    PG_HIP(ctx, hipMemcpyAsync(A->srefs_d, srefs.data(), n_refs * sizeof(SeedRef), hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, ctx->stream));
    PG_HIP(ctx, hipMemsetAsync(A->seed_total, 0, 8, ctx->stream));   // [0] matches, [1] hits
    PG_HIP(ctx, hipMemsetAsync(A->hit_count, 0, n_units * 4, ctx->stream));
    hipLaunchKernelGGL(anim_seed_kernel, dim3(SEED_GROUPS, n_refs), dim3(SEED_BLOCK), (size_t)slots * 8 + SEED_STAGE_BYTES, ctx->stream,
                       A->refs_d, A->units_d, A->srefs_d, A->sqry_d, A->slice_d, slice_stride, slots - 1, A->hits_d,
                       (uint32_t)A->hit_cap, A->seed_total + 1, A->hit_count);
    // hits -> per-unit slices, then one workgroup per unit verifies / extends them
    hipLaunchKernelGGL(anim_hoff_kernel, dim3(1), dim3(1024), 0, ctx->stream, A->hit_count, n_units, A->hoff, A->hit_cursor);
    hipLaunchKernelGGL(anim_hit_scatter_kernel, dim3((uint32_t)ctx->num_cu * 8u), dim3(256), 0, ctx->stream, A->hits_d, A->seed_total + 1,
                       (uint32_t)A->hit_cap, A->hoff, A->hit_cursor, A->hits_sorted);
    hipLaunchKernelGGL(anim_hit_kernel, dim3(n_units), dim3(256), 0, ctx->stream, A->refs_d, A->units_d, A->hits_sorted, A->hoff,
                       A->seed_total + 1, (uint32_t)A->hit_cap, A->seedbuf, (uint32_t)A->seed_cap, A->seed_total, A->mem_count);
    PG_HIP(ctx, hipMemcpyAsync(cnt.data(), A->mem_count, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(counts, A->seed_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    total = counts[0];
    if (counts[1] > A->hit_cap) {   // hits were dropped: the counts are incomplete -> seed half as many pairs
      if (n_pairs == 1) {
        A->hit_cap = (size_t)counts[1] + 1024;
        if ((rc = regrow(ctx, A->hits_d, A->hit_cap))) return rc;
        if ((rc = regrow(ctx, A->hits_sorted, A->hit_cap))) return rc;
      } else {
        n_pairs = (n_pairs + 1) / 2;
        n_units = 2 * n_pairs;
      }
      continue;
    }
    uint64_t tot = 0, raw = 0;
    pairs_fit = 0;
    for (uint32_t p = 0; p < n_pairs; ++p) {
      const uint64_t need = (uint64_t)cnt[2 * p] + cnt[2 * p + 1] + 4;   // per unit: count + 1 (never empty), rounded up to even
      if (p > 0 && tot + need > max_matches) break;
      tot += need;
      raw += need - 4;
      pairs_fit = p + 1;
    }
    n_pairs = pairs_fit;
    n_units = 2 * n_pairs;
    if (total <= A->seed_cap) break;
    // overflow: make room for the prefix of pairs that fits the batch budget and seed that prefix again
    A->seed_cap = (size_t)(raw + raw / 8 + 1024);
    if ((rc = regrow(ctx, A->seedbuf, A->seed_cap))) return rc;
  }
  *n_done = n_pairs;
  uint32_t n_nonempty = 0;
  for (uint32_t u = 0; u < n_units; ++u) n_nonempty += cnt[u] >= 1024;   // units with real work (unrelated pairs have ~50 chance matches)
  moff.assign((size_t)n_units + 1, 0);
  for (uint32_t u = 0; u < n_units; ++u) moff[u + 1] = moff[u] + ((cnt[u] + 2) & ~1u);   // even slice sizes: 8-byte aligned sub-slices
  const size_t M = moff[n_units];
  if (M > A->matches) {
    const size_t cap = M + M / 4;
    if ((rc = regrow(ctx, A->mem, cap))) return rc;
    if ((rc = regrow(ctx, A->cm, cap))) return rc;
    if ((rc = regrow(ctx, A->iscratch, cap * 8))) return rc;
    if ((rc = regrow(ctx, A->chains, cap))) return rc;
    if ((rc = regrow(ctx, A->order, cap))) return rc;
    if ((rc = regrow(ctx, A->prev, cap))) return rc;
    if ((rc = regrow(ctx, A->next, cap))) return rc;
    if ((rc = regrow(ctx, A->alnof, cap))) return rc;
    if ((rc = regrow(ctx, A->fw, cap))) return rc;
    if ((rc = regrow(ctx, A->bw, cap))) return rc;
    if ((rc = regrow(ctx, A->S.alns, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_rrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_qrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.idx, cap))) return rc;
    if ((rc = regrow(ctx, A->S.from, cap))) return rc;
    if ((rc = regrow(ctx, A->S.sc, cap))) return rc;
    A->matches = cap;
  }
  A->S.aln_of = A->alnof;
  PG_HIP(ctx, hipMemcpyAsync(A->moff, moff.data(), ((size_t)n_units + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->status, 0, n_pairs * 4, ctx->stream));
  ClusterOut O{A->moff, A->cm, A->chains, A->nch, A->order, A->prev, A->next, A->status};
  if (total)
    hipLaunchKernelGGL(anim_scatter_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, A->seedbuf, total, A->moff, n_units,
                       A->mem_count, A->mem);
  if (getenv("PYANI_ANIM_SCALAR_CLUSTER") && !maxmatch)   // debugging aid: the one-thread-per-unit statement of the same algorithm
    hipLaunchKernelGGL(anim_cluster_kernel, dim3((n_units + 63) / 64), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, n_units,
                       A->mem, A->mem_count, A->iscratch, O);
  else if ((n_nonempty > 3000 && !getenv("PYANI_ANIM_SPLIT_CLUSTER")) || getenv("PYANI_ANIM_WAVE_PREP"))
    // thousands of units with matches: one wave per unit already fills the machine, and the radix scatters are bound by
    // HBM's partial-line write rate, which more waves per unit only congest (measured: C3 574 ms vs 724 ms split)
    hipLaunchKernelGGL(anim_cluster_wave_kernel, dim3(n_units), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, A->mem,
                       A->mem_count, A->iscratch, O, 0, maxmatch);
  else {   // few units: PREP_WAVES waves share each unit's sorts / union-find so that the largest unit is not the launch time
    hipLaunchKernelGGL(anim_cluster_prep_kernel, dim3(n_units), dim3(PREP_THREADS), 0, ctx->stream, A->refs_d, A->units_d, A->mem,
                       A->mem_count, A->iscratch, O, maxmatch);
    hipLaunchKernelGGL(anim_cluster_wave_kernel, dim3(n_units), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, A->mem,
                       A->mem_count, A->iscratch, O, 1, maxmatch);
  }
  // work list of (unit, chain): one wave each
  std::vector<int32_t> nch(n_units);
  PG_HIP(ctx, hipMemcpyAsync(nch.data(), A->nch, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // (unit, chain) work list, one wave each: offsets by a host prefix over the per-unit chain counts, entries on device
  std::vector<uint32_t> choff((size_t)n_units + 1, 0);
  for (uint32_t u = 0; u < n_units; ++u) choff[u + 1] = choff[u] + (uint32_t)nch[u];
  const size_t n_wl = choff[n_units];
  if (n_wl) {
    if (n_wl > A->wl) { if ((rc = regrow(ctx, A->wl_d, n_wl + n_wl / 2))) return rc; A->wl = n_wl + n_wl / 2; }
    uint32_t* choff_d = A->choff_d;
    PG_HIP(ctx, hipMemcpyAsync(choff_d, choff.data(), choff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(anim_wl_kernel, dim3(n_units), dim3(64), 0, ctx->stream, choff_d, A->wl_d);
    // gap tasks: one slot per match (sparse), a class byte per slot, and GAP_CLASSES + 1 slot lists
    const size_t Mp = (M + 15) & ~(size_t)15;
    if (Mp > A->tasks) {
      if ((rc = regrow(ctx, A->tasks_d, Mp))) return rc;
      if ((rc = regrow(ctx, A->task_cls, Mp))) return rc;
      if ((rc = regrow(ctx, A->task_lists, (GAP_CLASSES + 1) * Mp))) return rc;
      A->tasks = Mp;
    }
    PG_HIP(ctx, hipMemsetAsync(A->gap_counts, 0, (GAP_CLASSES + 1) * 4, ctx->stream));
    PG_HIP(ctx, hipMemsetAsync(A->task_cls, 0xFF, Mp, ctx->stream));
    hipLaunchKernelGGL(anim_gaps_kernel, dim3((uint32_t)n_wl), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O, A->wl_d,
                       A->fw, A->tasks_d, A->task_cls);
    hipLaunchKernelGGL(anim_gapsort_kernel, dim3((uint32_t)ctx->num_cu * 8u), dim3(GAPSORT_BLOCK), 0, ctx->stream, A->task_cls,
                       (uint32_t)Mp, A->task_lists, A->gap_counts);
    const dim3 lane_grid((uint32_t)ctx->num_cu * 8u);
    hipLaunchKernelGGL((anim_gapdp_lane_kernel<17, false>), lane_grid, dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O, A->tasks_d,
                       A->task_lists, A->gap_counts, A->fw);
    hipLaunchKernelGGL((anim_gapdp_lane_kernel<32, false>), lane_grid, dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O, A->tasks_d,
                       A->task_lists + Mp, A->gap_counts + 1, A->fw);
    hipLaunchKernelGGL((anim_gapdp_lane_kernel<48, true>), lane_grid, dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O, A->tasks_d,
                       A->task_lists + 2 * Mp, A->gap_counts + 2, A->fw);
    hipLaunchKernelGGL((anim_gapdp_lane_kernel<64, true>), lane_grid, dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O, A->tasks_d,
                       A->task_lists + 3 * Mp, A->gap_counts + 3, A->fw);
    // the larger gaps go through the lanes of the extension DP (anim_gapreq_kernel); their number sizes the buffers
    uint32_t gap_counts[GAP_CLASSES + 1];
    PG_HIP(ctx, hipMemcpyAsync(gap_counts, A->gap_counts, sizeof(gap_counts), hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n_big = gap_counts[GAP_CLASSES], n_ext = n_wl > n_big ? n_wl : n_big;
    // development / test knobs of the hand-over rule (results do not depend on them: tests/test_anim_gpu.py)
    const int tail_lanes = getenv("PYANI_EXT_TAIL_LANES") ? atoi(getenv("PYANI_EXT_TAIL_LANES")) : EXT_TAIL_LANES;
    const int tail_blocks = getenv("PYANI_EXT_TAIL_BLOCKS") ? atoi(getenv("PYANI_EXT_TAIL_BLOCKS")) : EXT_TAIL_BLOCKS;
    uint32_t dump_cap = EXT_DUMP_CAP;
    if (getenv("PYANI_EXT_DUMP_CAP") && (uint32_t)atoi(getenv("PYANI_EXT_DUMP_CAP")) < dump_cap) dump_cap = (uint32_t)atoi(getenv("PYANI_EXT_DUMP_CAP"));
    if (!A->ext_dumps && (rc = regrow(ctx, A->ext_dumps, EXT_DUMP_CAP))) return rc;
    if (!A->ext_counts && (rc = regrow(ctx, A->ext_counts, 8))) return rc;
    if (n_ext > A->ext_cap) {
      const size_t cap = n_ext + n_ext / 2;
      if ((rc = regrow(ctx, A->ext_reqs, 2 * cap))) return rc;
      if ((rc = regrow(ctx, A->ext_pre, EXT_ROUNDS * cap))) return rc;
      if ((rc = regrow(ctx, A->ext_wave, cap))) return rc;
      A->ext_cap = cap;
    }
    if (n_big) {
      PG_HIP(ctx, hipMemsetAsync(A->ext_counts, 0, 20, ctx->stream));   // [0], [1] list lengths, [2] hand-out cursor, [4] handed over
      hipLaunchKernelGGL(anim_gapreq_kernel, dim3((uint32_t)((n_big + 255) / 256)), dim3(256), 0, ctx->stream, A->refs_d, A->units_d,
                         A->tasks_d, A->task_lists + GAP_CLASSES * Mp, (uint32_t)n_big, A->ext_pre, A->ext_reqs, A->ext_counts);
      hipLaunchKernelGGL(anim_extdp_lane_kernel, dim3((uint32_t)ctx->num_cu * 4u), dim3(64), 0, ctx->stream, A->ext_reqs, A->ext_reqs,
                         A->ext_counts, A->ext_counts + 2, A->ext_pre, A->ext_dumps, dump_cap, A->ext_counts + 4, tail_lanes, tail_blocks);
      hipLaunchKernelGGL(anim_gapdp_kernel, dim3((uint32_t)ctx->num_cu * 32u), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O,
                         A->tasks_d, A->task_lists + GAP_CLASSES * Mp, (uint32_t)n_big, A->ext_pre, A->ext_dumps, A->fw);
    }
    for (int phase = 0; phase < 2; ++phase) {
      // the first DP calls of every chain: written down, solved one per LANE, then consumed by the wave kernel
      PG_HIP(ctx, hipMemsetAsync(A->ext_counts + 4, 0, 4, ctx->stream));   // [4] searches handed over mid-way
      for (int round = 0; round < EXT_ROUNDS; ++round) {
        PG_HIP(ctx, hipMemsetAsync(A->ext_counts, 0, 16, ctx->stream));   // [0], [1] list lengths, [2] hand-out cursor
        hipLaunchKernelGGL(anim_extreq_kernel, dim3((uint32_t)((n_wl + 255) / 256)), dim3(256), 0, ctx->stream, A->refs_d, A->units_d, O,
                           A->wl_d, (uint32_t)n_wl, A->fw, A->bw, phase, round, A->ext_pre, A->ext_reqs, A->ext_reqs + n_wl, A->ext_counts,
                           A->ext_wave);
        hipLaunchKernelGGL(anim_extdp_lane_kernel, dim3((uint32_t)ctx->num_cu * 4u), dim3(64), 0, ctx->stream, A->ext_reqs,
                           A->ext_reqs + n_wl, A->ext_counts, A->ext_counts + 2, A->ext_pre + (size_t)round * n_wl, A->ext_dumps,
                           dump_cap, A->ext_counts + 4, tail_lanes, tail_blocks);
      }
      // final round: chains whose calls were all answered are finished by a thread each; the others (handed-over searches,
      // third calls, junction rectangles) are listed for the wave kernel
      PG_HIP(ctx, hipMemsetAsync(A->ext_counts, 0, 12, ctx->stream));   // [0] list length, [2] hand-out cursor
      hipLaunchKernelGGL(anim_extreq_kernel, dim3((uint32_t)((n_wl + 255) / 256)), dim3(256), 0, ctx->stream, A->refs_d, A->units_d, O,
                         A->wl_d, (uint32_t)n_wl, A->fw, A->bw, phase, EXT_ROUNDS, A->ext_pre, A->ext_reqs, A->ext_reqs + n_wl,
                         A->ext_counts, A->ext_wave);
      hipLaunchKernelGGL(anim_extend_kernel, dim3((uint32_t)ctx->num_cu * 32u), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O,
                         A->wl_d, A->fw, A->bw, phase, A->ext_pre, (uint32_t)n_wl, A->ext_dumps, A->ext_wave, A->ext_counts,
                         A->ext_counts + 2);
    }
  }
  hipLaunchKernelGGL(anim_finish_kernel, dim3(n_pairs), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, n_pairs,
                     O, A->fw, A->bw, A->S, filter_1to1, A->out);
  PG_HIP(ctx, hipGetLastError());
  PG_HIP(ctx, hipMemcpyAsync(out_host, A->out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
#ifdef PGA_DP_STATS
  {
    unsigned long long st[3][40];
    PG_HIP(ctx, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_dp_stats), sizeof(st)));
    unsigned long long cl[16];
    PG_HIP(ctx, hipMemcpyFromSymbol(cl, HIP_SYMBOL(g_cl_stats), sizeof(cl)));
    fprintf(stderr, "[cluster-stats] phases mumfilter/unionfind/rootsort/chains/tail: sum cycles %llu %llu %llu %llu %llu  max %llu %llu %llu %llu %llu  max n_in %llu\n",
            cl[0], cl[1], cl[2], cl[3], cl[4], cl[6], cl[7], cl[8], cl[9], cl[10], cl[12]);
    fprintf(stderr, "[cluster-stats] general-path rounds %llu, sum of live entries over rounds %llu, cycles in the general path %llu\n", cl[13], cl[14], cl[15]);
    const char* names[3] = {"gap", "fwd", "bwd"};
    for (int k = 0; k < 3; ++k) {
      fprintf(stderr, "[dp-stats] %s calls %llu steps %llu cycles %llu  first calls: delivered %llu, handed over %llu, not valid %llu, other arguments %llu  hist(log2 steps):",
              names[k], st[k][0], st[k][1], st[k][2], st[k][3], st[k][6], st[k][4], st[k][5]);
      for (int b = 0; b < 20; ++b) fprintf(stderr, " %llu", st[k][8 + b]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return PG_OK;
}

// The alignment records of the pair a 1-pair batch has just processed (slice 0 of the finish scratch), converted to
// MUMmer's per-record 1-based closed coordinates.
int pg_anim_fetch_alignments(pg_ctx* ctx, int32_t ref_id, int32_t qry_id, uint32_t n, pg_anim_alignment* out) {
  AnimScratch* A = anim_scratch(ctx);
  std::vector<Aln> al(n);
  std::vector<int32_t> rr(n), qr(n);
  if (n) {
    PG_HIP(ctx, hipMemcpy(al.data(), A->S.alns, n * sizeof(Aln), hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(rr.data(), A->S.a_rrec, n * 4, hipMemcpyDeviceToHost));
    PG_HIP(ctx, hipMemcpy(qr.data(), A->S.a_qrec, n * 4, hipMemcpyDeviceToHost));
  }
  const PgGenome& G = ctx->genomes[ref_id];
  const PgGenome& H = ctx->genomes[qry_id];
  for (uint32_t i = 0; i < n; ++i) {
    const Aln& a = al[i];            // forward stream coordinates, half-open
    const int32_t ro = G.rec_start[rr[i]], qo = H.rec_start[qr[i]];
    pg_anim_alignment x;
    x.ref_rec = rr[i]; x.qry_rec = qr[i];
    x.rs = a.rs - ro + 1; x.re = a.re - ro;
    x.qs = a.strand ? a.qe - qo : a.qs - qo + 1;
    x.qe = a.strand ? a.qs - qo + 1 : a.qe - qo;
    x.errors = a.errors; x.kept = a.keep;
    out[i] = x;
  }
  return PG_OK;
}

int pg_anim_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                       const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                       int apply_filter, pg_anim_result* out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<Aln> h(n);
  for (uint64_t i = 0; i < n; ++i) {
    Aln a;
    a.strand = qs[i] > qe[i];
    a.rs = (rs[i] < re[i] ? rs[i] : re[i]) - 1; a.re = rs[i] < re[i] ? re[i] : rs[i];
    a.qs = (qs[i] < qe[i] ? qs[i] : qe[i]) - 1; a.qe = qs[i] < qe[i] ? qe[i] : qs[i];
    a.errors = errors[i];
    a.keep = apply_filter ? 0 : 3;
    h[i] = a;
  }
  uint64_t* d_off = nullptr; Aln* d_a = nullptr; int32_t *d_rg = nullptr, *d_qg = nullptr, *d_idx = nullptr, *d_from = nullptr;
  double* d_sc = nullptr; pg_anim_result* d_out = nullptr;
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* p : to_free) if (p) (void)hipFree(p); };
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_a, n + 1); AA(d_rg, n + 1); AA(d_qg, n + 1); AA(d_idx, n + 1); AA(d_from, n + 1); AA(d_sc, n + 1);
  AA(d_out, n_pairs + 1);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_a, h.data(), n * sizeof(Aln), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_rg, rseq, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_qg, qseq, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anim_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, n_pairs, d_off, d_a, d_rg, d_qg,
                       d_idx, d_from, d_sc, apply_filter, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anim reduce: ") + hipGetErrorString(e));
  return PG_OK;
}

// ---- ANIb: parse_blast_tab reduction (pyani/anib.py:641-665), one thread per ordered pair ---------------------------
namespace {
__global__ __launch_bounds__(64) void anib_reduce_kernel(uint32_t n_pairs, const uint64_t* __restrict__ offsets,
                                                         const uint64_t* __restrict__ foff, const int32_t* __restrict__ frag,
                                                         const int32_t* __restrict__ length, const int32_t* __restrict__ mismatch,
                                                         const int32_t* __restrict__ gaps, const int32_t* __restrict__ qlen,
                                                         const double* __restrict__ pident, int64_t* first_row,
                                                         int64_t* __restrict__ aln_out, int64_t* __restrict__ err_out,
                                                         double* __restrict__ pid_out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int64_t* first = first_row + foff[p];
  const uint64_t nf = foff[p + 1] - foff[p];
  for (uint64_t f = 0; f < nf; ++f) first[f] = -1;
  for (uint64_t i = offsets[p]; i < offsets[p + 1]; ++i) {
    const int32_t alnlen = length[i] - gaps[i], alnids = alnlen - mismatch[i];
    const double cov = (double)alnlen / (double)qlen[i], pid = (double)alnids / (double)qlen[i];
    if (cov > 0.7 && pid > 0.3 && (uint64_t)frag[i] < nf && first[frag[i]] < 0) first[frag[i]] = (int64_t)i;
  }
  int64_t aln = 0, err = 0, cnt = 0;
  double sum = 0.0;
  for (uint64_t f = 0; f < nf; ++f) {
    const int64_t i = first[f];
    if (i < 0) continue;
    aln += length[i] - gaps[i];
    err += (int64_t)mismatch[i] + gaps[i];
    sum = sum + pident[i];
    ++cnt;
  }
  aln_out[p] = aln;
  err_out[p] = err;
  pid_out[p] = cnt ? sum / (double)cnt : 0.0;
}
}  // namespace

int pg_anib_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                       const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen,
                       const double* pident, int64_t* aln_out, int64_t* err_out, double* pid_out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<uint64_t> foff(n_pairs + 1, 0);
  for (uint32_t p = 0; p < n_pairs; ++p) foff[p + 1] = foff[p] + n_frags[p];
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* q : to_free) if (q) (void)hipFree(q); };
  uint64_t *d_off = nullptr, *d_foff = nullptr;
  int32_t *d_frag = nullptr, *d_len = nullptr, *d_mm = nullptr, *d_gap = nullptr, *d_ql = nullptr;
  double *d_pid = nullptr, *d_pout = nullptr;
  int64_t *d_first = nullptr, *d_aln = nullptr, *d_err = nullptr;
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_foff, n_pairs + 1); AA(d_frag, n + 1); AA(d_len, n + 1); AA(d_mm, n + 1); AA(d_gap, n + 1);
  AA(d_ql, n + 1); AA(d_pid, n + 1); AA(d_first, foff[n_pairs] + 1); AA(d_aln, n_pairs); AA(d_err, n_pairs); AA(d_pout, n_pairs);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_foff, foff.data(), (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  const struct { void* d; const void* h; size_t b; } cp[] = {{d_frag, frag, n * 4}, {d_len, length, n * 4}, {d_mm, mismatch, n * 4},
                                                           {d_gap, gaps, n * 4}, {d_ql, qlen, n * 4}, {d_pid, pident, n * 8}};
  for (const auto& c : cp)
    if (e == hipSuccess && c.b) e = hipMemcpyAsync(c.d, c.h, c.b, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anib_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, n_pairs, d_off, d_foff, d_frag, d_len,
                       d_mm, d_gap, d_ql, d_pid, d_first, d_aln, d_err, d_pout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(aln_out, d_aln, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(err_out, d_err, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(pid_out, d_pout, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anib reduce: ") + hipGetErrorString(e));
  return PG_OK;
}
