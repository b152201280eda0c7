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

// ---- K1: counts + Z-scores --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rc_index(uint32_t x, int k) {
  // reverse complement of a k-mer index (first base most significant): complement = 3 - digit, then reverse digits
  uint32_t c = ((1u << (2 * k)) - 1u) - x, r = 0;
  for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
  return r;
}

// One 256-thread block per genome.  If acc != nullptr: build c2/c3/c4 from the K0 accumulators (and store them
// to counts); else read c2/c3/c4 from counts.  Then the Z-scores in the reference's operation order.
__global__ __launch_bounds__(256) void tetra_finalize_kernel(const unsigned long long* __restrict__ acc,
                                                             const uint32_t* __restrict__ quirk,
                                                             const uint32_t* __restrict__ batch_gid,
                                                             unsigned long long* __restrict__ counts,
                                                             double* __restrict__ z, uint8_t* __restrict__ present) {
  __shared__ unsigned long long F4[256], F3[64], F2[16], c4[256], c3[64], c2[16];
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  unsigned long long* cg = counts + (size_t)g * PG_ACC_WORDS;
  if (acc) {
    const unsigned long long* a = acc + (size_t)g * PG_ACC_WORDS;
    F4[t] = a[80 + t];
    __syncthreads();
    if (t < 64) F3[t] = F4[4 * t] + F4[4 * t + 1] + F4[4 * t + 2] + F4[4 * t + 3] + a[16 + t];
    __syncthreads();
    if (t < 16) F2[t] = F3[4 * t] + F3[4 * t + 1] + F3[4 * t + 2] + F3[4 * t + 3] + a[t];
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
  if (!z) return;
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
}

// ---- K2: per-genome statistics (sequential sums, CPython-3.10 sum() order) ------------------------------------
// One thread per genome.  dev[g][0..cnt) = z - mean over the present keys in tetramer order; ss[g] = sum(dev^2).
// flags[0] |= 1 if present[g] differs from present[0] (AssertionError in the reference); flags[1] = cnt.
__global__ __launch_bounds__(64) void tetra_stats_kernel(const double* __restrict__ z, const uint8_t* __restrict__ present,
                                                         uint32_t n, double* __restrict__ dev, double* __restrict__ ss,
                                                         int32_t* __restrict__ flags) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const double* zg = z + (size_t)g * 256;
  const uint8_t* pg = present + (size_t)g * 256;
  const uint8_t* p0 = present;
  double s = 0.0;
  int cnt = 0;
  bool same = true;
  for (int t = 0; t < 256; ++t) {
    const uint8_t p = pg[t];
    same = same && (p == p0[t]);
    if (p) { s = s + zg[t]; ++cnt; }
  }
  if (!same) atomicOr(&flags[0], 1);
  if (g == 0) flags[1] = cnt;
  double* dg = dev + (size_t)g * 256;
  double acc = 0.0;
  if (cnt > 0) {
    const double m = s / (double)cnt;
    int k = 0;
    for (int t = 0; t < 256; ++t) {
      if (pg[t]) {
        const double d = zg[t] - m;
        dg[k++] = d;
        acc = acc + d * d;
      }
    }
  }
  ss[g] = acc;
}

// ---- K3: Pearson matrix ---------------------------------------------------------------------------------------
// 32x32 tile of pairs per 256-thread block, 2x2 pairs per thread; the 256-term dot product of each pair is
// accumulated strictly in k order (each product rounded, then added: no FMA), as tetra.py:186-188 does.
constexpr int K3_T = 32, K3_KC = 64;

__global__ __launch_bounds__(256) void tetra_pairs_kernel(const double* __restrict__ dev, const double* __restrict__ ss,
                                                          const int32_t* __restrict__ flags, uint32_t n, uint32_t row0,
                                                          uint32_t nrows, double* __restrict__ out, int mirror) {
  __shared__ double A[K3_T][K3_KC + 1], B[K3_T][K3_KC + 1];
  const uint32_t ti = blockIdx.y, tj = blockIdx.x;
  if (mirror && tj < ti) return;  // upper-triangle tiles only; results are mirrored (r(i,j) == r(j,i) bitwise)
  const int cnt = flags[1];
  const uint32_t i0 = row0 + ti * K3_T, j0 = tj * K3_T;
  const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int k0 = 0; k0 < cnt; k0 += K3_KC) {
    const int kc = min(K3_KC, cnt - k0);
    for (int e = threadIdx.x; e < K3_T * K3_KC; e += 256) {
      const int r = e / K3_KC, k = e % K3_KC;
      const uint32_t gi = i0 + r, gj = j0 + r;
      A[r][k] = (gi < row0 + nrows && gi < n && k < kc) ? dev[(size_t)gi * 256 + k0 + k] : 0.0;
      B[r][k] = (gj < n && k < kc) ? dev[(size_t)gj * 256 + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const double a0 = A[ty][k], a1 = A[ty + 16][k], b0 = B[tx][k], b1 = B[tx + 16][k];
      acc[0][0] = acc[0][0] + a0 * b0;
      acc[0][1] = acc[0][1] + a0 * b1;
      acc[1][0] = acc[1][0] + a1 * b0;
      acc[1][1] = acc[1][1] + a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i >= row0 + nrows || i >= n || j >= n) continue;
      const double r = (i == j) ? 1.0 : acc[a][b] / sqrt(ss[i] * ss[j]);   // tetra.py:171 (diag), :190-192
      if (!mirror) {
        out[(size_t)(i - row0) * n + j] = r;
      } else if (j >= i) {
        out[(size_t)i * n + j] = r;
        out[(size_t)j * n + i] = r;
      }
    }
}

}  // namespace

// ---- host launchers ---------------------------------------------------------------------------------------------
// configuration chosen on MI355X with tools/microbench/count_bench.hip (profiles/r01_count_variants.txt)
constexpr int K0_PF = 2, K0_REPL = 1;

int pg_launch_tetra_count(pg_ctx* ctx, uint32_t n_batch) {
  static bool attr_set = false;
  const size_t lds_bytes = K0Lds<K0_REPL>::WORDS * sizeof(uint32_t);
  auto kern = tetra_count_kernel<K0_PF, 0, K0_REPL>;
  if (!attr_set) {
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  PG_HIP(ctx, hipMemsetAsync(ctx->d_acc, 0, (size_t)n_batch * PG_ACC_WORDS * sizeof(unsigned long long), ctx->stream));
  if (ctx->n_work == 0) return PG_OK;
  const uint32_t grid = ctx->n_work < (uint32_t)ctx->num_cu ? ctx->n_work : (uint32_t)ctx->num_cu;
  pg_prof_begin(ctx, PG_K_TETRA_COUNT);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(K0_BLOCK), lds_bytes, ctx->stream, ctx->d_codes, ctx->d_mask,
                     ctx->d_seg_tile0, ctx->d_seg_prefix, n_batch, ctx->d_acc);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_finalize(pg_ctx* ctx, uint32_t n_batch, const unsigned long long* d_acc_in) {
  if (n_batch == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_FINALIZE);
  hipLaunchKernelGGL(tetra_finalize_kernel, dim3(n_batch), dim3(256), 0, ctx->stream, d_acc_in, ctx->d_quirk,
                     ctx->d_batch_gid, ctx->d_counts, ctx->d_z, ctx->d_present);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_stats(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n) {
  PG_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 2 * sizeof(int32_t), ctx->stream));
  if (n == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_STATS);
  hipLaunchKernelGGL(tetra_stats_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_z, d_present, n, ctx->d_dev,
                     ctx->d_ss, ctx->d_flags);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_pairs(pg_ctx* ctx, uint32_t n, uint32_t row0, uint32_t nrows, double* d_out, bool mirror) {
  if (n == 0 || nrows == 0) return PG_OK;
  const dim3 grid((n + K3_T - 1) / K3_T, (nrows + K3_T - 1) / K3_T);
  pg_prof_begin(ctx, PG_K_TETRA_PAIRS);
  hipLaunchKernelGGL(tetra_pairs_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_dev, ctx->d_ss, ctx->d_flags, n, row0,
                     nrows, d_out, mirror ? 1 : 0);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}
