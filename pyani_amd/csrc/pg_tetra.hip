// pg_tetra.hip — hand-written gfx950 (CDNA4, wave64) kernels for pyani's TETRA path.
//
//   K0 tetra_count     HBM-streaming k-mer histogram over the 2-bit/1-bit packed genome arena   (HBM roofline)
//   K1 tetra_finalize  marginals + reverse-complement fold + quirk -> c2/c3/c4 -> 256 Z-scores   (tiny)
//   K2 tetra_stats     per-genome mean / deviations / sum of squares, sequential order           (tiny)
//   K3 tetra_pairs     all-vs-all Pearson, sequential 256-term dot products in fp64              (latency bound)
//
// Reference semantics: pyani/tetra.py:98-138 (counts, Z) and :158-194 (Pearson); closed form in SURVEY.md App. A.
// Built with -ffp-contract=off: the fp64 operation ORDER is part of the contract (bit-exact results).
//
// K0 design (integer/byte work — no MFMA on purpose):
//   see pg_tetra_count.h: one 1024-thread workgroup per CU streams 64 bases per lane per tile and counts
//   heptamers at stride 4 with LDS atomics (one atomic per 4 bases), folded to tetramers at flush.
#include "pg_internal.h"

namespace {

#include "pg_tetra_count.h"

// ---- K1: counts + Z-scores + per-genome statistics ------------------------------------------------------------
__device__ __forceinline__ uint32_t rc_index(uint32_t x, int k) {
  // reverse complement of a k-mer index (first base most significant): complement = 3 - digit, then reverse digits
  uint32_t c = ((1u << (2 * k)) - 1u) - x, r = 0;
  for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
  return r;
}

// Statistics of one genome (tetra.py:181-189) for the Pearson kernel, by one workgroup of NT threads.
// zrow/prow: Z and presence of the 256 tetramers in LDS.  Only the two SUMS are order-sensitive, so only they are
// sequential (one wave, every lane redundantly, operands from a compacted LDS array so the loop has no data-dependent
// branch and its LDS reads pipeline); compaction and the deviations themselves are computed in parallel.
//   mean = (((0 + z0) + z1) + ...) / cnt   over the present keys in tetramer order   (CPython-3.10 sum())
//   dev[k] = z[k] - mean ;  ss = ((d0*d0 + d1*d1) + ...)
template <int NT>
__device__ __forceinline__ void genome_stats(const double* zrow, const uint8_t* prow, double* zc, double* dsq,
                                             double* bcast, uint32_t* wave_cnt, double* __restrict__ dev_g,
                                             double* __restrict__ ss_g) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // 1. compact the present keys (stable): rank = present keys before t
  for (uint32_t base = 0; base < 256; base += NT) {
    const uint32_t t = base + tid;
    const unsigned long long bits = __ballot(prow[t] != 0);
    if (lane == 0) wave_cnt[t >> 6] = (uint32_t)__popcll(bits);
  }
  __syncthreads();
  const uint32_t cnt = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  for (uint32_t base = 0; base < 256; base += NT) {
    const uint32_t t = base + tid;
    const bool p = prow[t] != 0;
    const unsigned long long bits = __ballot(p);
    uint32_t rank = (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < (t >> 6); ++w) rank += wave_cnt[w];
    if (p) zc[rank] = zrow[t];
  }
  __syncthreads();
  // 2. sequential mean
  if (tid < 64) {
    double s = 0.0;
    uint32_t k = 0;
    for (; k + 8 <= cnt; k += 8) {
      const double a0 = zc[k], a1 = zc[k + 1], a2 = zc[k + 2], a3 = zc[k + 3], a4 = zc[k + 4], a5 = zc[k + 5],
                   a6 = zc[k + 6], a7 = zc[k + 7];
      s = s + a0; s = s + a1; s = s + a2; s = s + a3; s = s + a4; s = s + a5; s = s + a6; s = s + a7;
    }
    for (; k < cnt; ++k) s = s + zc[k];
    if (tid == 0) bcast[0] = cnt ? s / (double)cnt : 0.0;
  }
  __syncthreads();
  // 3. deviations in parallel
  const double m = bcast[0];
  for (uint32_t k = tid; k < cnt; k += NT) {
    const double d = zc[k] - m;
    dev_g[k] = d;
    dsq[k] = d * d;
  }
  __syncthreads();
  // 4. sequential sum of squares
  if (tid < 64) {
    double acc = 0.0;
    uint32_t k = 0;
    for (; k + 8 <= cnt; k += 8) {
      const double a0 = dsq[k], a1 = dsq[k + 1], a2 = dsq[k + 2], a3 = dsq[k + 3], a4 = dsq[k + 4], a5 = dsq[k + 5],
                   a6 = dsq[k + 6], a7 = dsq[k + 7];
      acc = acc + a0; acc = acc + a1; acc = acc + a2; acc = acc + a3; acc = acc + a4; acc = acc + a5; acc = acc + a6; acc = acc + a7;
    }
    for (; k < cnt; ++k) acc = acc + dsq[k];
    if (tid == 0) *ss_g = acc;
  }
}

// One 256-thread block per genome.  If acc != nullptr: build c2/c3/c4 from the K0 accumulators (and re-zero them for
// the next pass); else read c2/c3/c4 from counts.  Then the Z-scores in the reference's operation order, the
// observed-key bitmap, and the statistics the Pearson kernel needs.
__global__ __launch_bounds__(256) void tetra_finalize_kernel(unsigned long long* __restrict__ acc,
                                                             const uint32_t* __restrict__ quirk,
                                                             const uint32_t* __restrict__ batch_gid,
                                                             unsigned long long* __restrict__ counts,
                                                             double* __restrict__ z, uint8_t* __restrict__ present,
                                                             double* __restrict__ dev, double* __restrict__ ss,
                                                             unsigned long long* __restrict__ keybits,
                                                             int32_t* __restrict__ flags) {
  __shared__ unsigned long long F4[256], F3[64], F2[16], c4[256], c3[64], c2[16];
  __shared__ double zrow[256], zc[256], dsq[256], bcast[1];
  __shared__ uint8_t prow[256];
  __shared__ uint32_t wave_cnt[4];
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  if (g == 0 && t == 0) flags[0] = 0;  // key-set mismatch flag, raised by the pairs kernel later in the stream
  unsigned long long* cg = counts + (size_t)g * PG_ACC_WORDS;
  if (acc) {
    unsigned long long* a = acc + (size_t)g * PG_ACC_WORDS;
    F4[t] = a[80 + t];
    a[80 + t] = 0;
    unsigned long long e3 = 0, e2 = 0;
    if (t < 64) { e3 = a[16 + t]; a[16 + t] = 0; }
    if (t < 16) { e2 = a[t]; a[t] = 0; }
    __syncthreads();
    if (t < 64) F3[t] = F4[4 * t] + F4[4 * t + 1] + F4[4 * t + 2] + F4[4 * t + 3] + e3;
    __syncthreads();
    if (t < 16) F2[t] = F3[4 * t] + F3[4 * t + 1] + F3[4 * t + 2] + F3[4 * t + 3] + e2;
    __syncthreads();
    c4[t] = F4[t] + F4[rc_index(t, 4)] - (unsigned long long)quirk[(size_t)batch_gid[g] * 256 + t];
    if (t < 64) c3[t] = F3[t] + F3[rc_index(t, 3)];
    if (t < 16) c2[t] = F2[t] + F2[rc_index(t, 2)];
    __syncthreads();
    cg[80 + t] = c4[t];
    if (t < 64) cg[16 + t] = c3[t];
    if (t < 16) cg[t] = c2[t];
  } else {
    c4[t] = cg[80 + t];
    if (t < 64) c3[t] = cg[16 + t];
    if (t < 16) c2[t] = cg[t];
    __syncthreads();
  }
  const unsigned long long obs = c4[t];
  double zv = 0.0;
  uint8_t pv = 0;
  if (obs != 0) {
    const unsigned long long a = c3[t >> 2], b = c3[t & 63u], den = c2[(t >> 2) & 15u];
    const double e = ((1.0 * (double)a) * (double)b) / (double)den;                                   // tetra.py:121-123
    const double sd = sqrt(((e * (double)(den - a)) * (double)(den - b)) / (double)(den * den));      // tetra.py:129-132
    if (sd != 0.0) zv = ((double)obs - e) / sd;                                                        // tetra.py:134
    else zv = 1.0 / (double)(den * den);                                                               // tetra.py:135-138
    pv = 1;
  }
  z[(size_t)g * 256 + t] = zv;
  present[(size_t)g * 256 + t] = pv;
  zrow[t] = zv;
  prow[t] = pv;
  const unsigned long long bits = __ballot(pv != 0);
  if ((t & 63u) == 0) keybits[(size_t)g * 4 + (t >> 6)] = bits;
  __syncthreads();
  genome_stats<256>(zrow, prow, zc, dsq, bcast, wave_cnt, dev + (size_t)g * 256, ss + g);
}

// ---- K2: statistics from given Z rows (host-provided Z, or the all-gathered Z of a multi-GPU job) -----------
__global__ __launch_bounds__(256) void tetra_stats_kernel(const double* __restrict__ z, const uint8_t* __restrict__ present,
                                                          uint32_t n, double* __restrict__ dev, double* __restrict__ ss,
                                                          unsigned long long* __restrict__ keybits,
                                                          int32_t* __restrict__ flags) {
  __shared__ double zrow[256], zc[256], dsq[256], bcast[1];
  __shared__ uint8_t prow[256];
  __shared__ uint32_t wave_cnt[4];
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  if (g == 0 && t == 0) flags[0] = 0;
  const uint8_t pv = present[(size_t)g * 256 + t];
  zrow[t] = z[(size_t)g * 256 + t];
  prow[t] = pv;
  const unsigned long long bits = __ballot(pv != 0);
  if ((t & 63u) == 0) keybits[(size_t)g * 4 + (t >> 6)] = bits;
  __syncthreads();
  genome_stats<256>(zrow, prow, zc, dsq, bcast, wave_cnt, dev + (size_t)g * 256, ss + g);
}

// ---- K3: Pearson matrix ---------------------------------------------------------------------------------------
// 16x16 tile of pairs per 256-thread block, one pair per thread.  Both 16-row panels of deviations are staged in LDS
// once (2 x 16 x 256 f64 = 64 KiB), then each thread accumulates its pair's dot product strictly in k order (each
// product rounded, then added: no FMA), as tetra.py:186-188 does.  r(i,j) == r(j,i) bitwise, so with `mirror` only
// upper-triangle tiles are computed and both cells written.
constexpr int K3_T = 16, K3_LD = 256 + 2;  // +2 doubles: rows start 4 banks apart -> conflict-free ds_read_b64/b128

__global__ __launch_bounds__(256) void tetra_pairs_kernel(const double* __restrict__ dev, const double* __restrict__ ss,
                                                          const unsigned long long* __restrict__ keybits,
                                                          int32_t* __restrict__ flags, uint32_t n, uint32_t row0,
                                                          uint32_t nrows, double* __restrict__ out, int mirror) {
  extern __shared__ __attribute__((aligned(16))) double k3_lds[];
  double* A = k3_lds;                  // [K3_T][K3_LD]
  double* B = k3_lds + K3_T * K3_LD;   // [K3_T][K3_LD]
  const uint32_t ti = blockIdx.y, tj = blockIdx.x;
  if (mirror && tj < ti) return;
  const unsigned long long k0 = keybits[0], k1 = keybits[1], k2 = keybits[2], k3 = keybits[3];
  const uint32_t cnt = (uint32_t)(__popcll(k0) + __popcll(k1) + __popcll(k2) + __popcll(k3));
  const uint32_t i0 = row0 + ti * K3_T, j0 = tj * K3_T;
  const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
  if (threadIdx.x < 2 * K3_T) {  // key sets of this tile's genomes must equal genome 0's (tetra.py:174-175)
    const uint32_t g = threadIdx.x < K3_T ? i0 + threadIdx.x : j0 + threadIdx.x - K3_T;
    if (g < n) {
      const unsigned long long* kb = keybits + (size_t)g * 4;
      const unsigned long long d = (kb[0] ^ k0) | (kb[1] ^ k1) | (kb[2] ^ k2) | (kb[3] ^ k3);
      if (d) atomicOr(&flags[0], 1);
    }
  }
  if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0) flags[1] = (int32_t)cnt;
  {  // stage both panels: thread t fetches column t of all 16 rows (coalesced); all 32 loads are issued before
     // the first LDS store.  Row indices are clamped, not predicated: out-of-range rows are never written out.
    double va[K3_T], vb[K3_T];
    const uint32_t i_hi = min(row0 + nrows, n) - 1, j_hi = n - 1;
#pragma unroll
    for (int r = 0; r < K3_T; ++r) {
      va[r] = dev[(size_t)min(i0 + r, i_hi) * 256 + threadIdx.x];
      vb[r] = dev[(size_t)min(j0 + r, j_hi) * 256 + threadIdx.x];
    }
#pragma unroll
    for (int r = 0; r < K3_T; ++r) {
      A[r * K3_LD + threadIdx.x] = va[r];
      B[r * K3_LD + threadIdx.x] = vb[r];
    }
  }
  __syncthreads();
  const double* a = A + ty * K3_LD;
  const double* b = B + tx * K3_LD;
  double acc = 0.0;
  uint32_t k = 0;
  for (; k + 8 <= cnt; k += 8) {
    const double p0 = a[k] * b[k], p1 = a[k + 1] * b[k + 1], p2 = a[k + 2] * b[k + 2], p3 = a[k + 3] * b[k + 3],
                 p4 = a[k + 4] * b[k + 4], p5 = a[k + 5] * b[k + 5], p6 = a[k + 6] * b[k + 6], p7 = a[k + 7] * b[k + 7];
    acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3; acc = acc + p4; acc = acc + p5; acc = acc + p6; acc = acc + p7;
  }
  for (; k < cnt; ++k) acc = acc + a[k] * b[k];
  const uint32_t i = i0 + ty, j = j0 + tx;
  if (i >= row0 + nrows || i >= n || j >= n) return;
  const double r = (i == j) ? 1.0 : acc / sqrt(ss[i] * ss[j]);   // tetra.py:171 (diag), :190-192
  if (!mirror) {
    out[(size_t)(i - row0) * n + j] = r;
  } else if (j >= i) {
    out[(size_t)i * n + j] = r;
    out[(size_t)j * n + i] = r;
  }
}

}  // namespace

// ---- host launchers ---------------------------------------------------------------------------------------------
// configuration chosen on MI355X with tools/microbench/count_bench.hip (profiles/archive/r01_count_variants.txt)
constexpr int K0_PF = 3, K0_BLOCK = 512, K0_BLOCKS_PER_CU = 2;

int pg_launch_tetra_count(pg_ctx* ctx, uint32_t n_batch) {
  auto kern = tetra_count_kernel<K0_PF, 0, K0_BLOCK>;   // 73 KiB of static LDS per workgroup
  // d_acc is zero here: zeroed at allocation and re-zeroed by the finalize kernel after every pass
  if (ctx->n_work == 0) return PG_OK;
  const uint32_t want = (uint32_t)ctx->num_cu * K0_BLOCKS_PER_CU, tiles = ctx->n_work * (1024 / K0_BLOCK);
  const uint32_t grid = tiles < want ? tiles : want;
  pg_prof_begin(ctx, PG_K_TETRA_COUNT);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(K0_BLOCK), 0, ctx->stream, ctx->d_codes, ctx->d_mask,
                     ctx->d_seg_tile0, ctx->d_seg_prefix, n_batch, ctx->d_acc);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_finalize(pg_ctx* ctx, uint32_t n_batch, unsigned long long* d_acc_in) {
  if (n_batch == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_FINALIZE);
  hipLaunchKernelGGL(tetra_finalize_kernel, dim3(n_batch), dim3(256), 0, ctx->stream, d_acc_in, ctx->d_quirk,
                     ctx->d_batch_gid, ctx->d_counts, ctx->d_z, ctx->d_present, ctx->d_dev, ctx->d_ss, ctx->d_keybits,
                     ctx->d_flags);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_stats(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n) {
  if (n == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_STATS);
  hipLaunchKernelGGL(tetra_stats_kernel, dim3(n), dim3(256), 0, ctx->stream, d_z, d_present, n, ctx->d_dev, ctx->d_ss,
                     ctx->d_keybits, ctx->d_flags);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_pairs(pg_ctx* ctx, uint32_t n, uint32_t row0, uint32_t nrows, double* d_out, bool mirror) {
  if (n == 0 || nrows == 0) return PG_OK;
  const dim3 grid((n + K3_T - 1) / K3_T, (nrows + K3_T - 1) / K3_T);
  const size_t lds_bytes = 2 * K3_T * K3_LD * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tetra_pairs_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  pg_prof_begin(ctx, PG_K_TETRA_PAIRS);
  hipLaunchKernelGGL(tetra_pairs_kernel, grid, dim3(256), lds_bytes, ctx->stream, ctx->d_dev, ctx->d_ss, ctx->d_keybits,
                     ctx->d_flags, n, row0, nrows, d_out, mirror ? 1 : 0);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}
