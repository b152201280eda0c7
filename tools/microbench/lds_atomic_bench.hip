// lds_atomic_bench.hip — measures the LDS atomic (ds_add_u32, no return) rate on gfx950 for the access patterns the
// TETRA count kernel can choose between.  Standalone: hipcc --offload-arch=gfx950 -O3 lds_atomic_bench.hip -o lds_bench
// Output: one line per (pattern, waves/CU): atomics per clock per CU (assuming the measured kernel clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int LDS_WORDS = 32768;  // 128 KiB

// pattern 0: 1024 bins x 32 replicas, lane -> bank lane%32 (conflict-free)
// pattern 1: 2048 bins x 16 replicas (lanes l and l+16 may collide: <=2-way)
// pattern 2: 16384 bins, no replication (random banks)
// pattern 3: 4096 bins x 8 replicas
// pattern 4: all lanes same address (worst case)
// pattern 5: conflict-free but with plain ds_write_b32 instead of atomic (rate reference)
template <int PATTERN>
__global__ __launch_bounds__(1024) void bench(const uint32_t* __restrict__ keys, uint32_t iters, uint32_t* out,
                                              unsigned long long* cycles) {
  extern __shared__ uint32_t lds[];
  for (uint32_t i = threadIdx.x; i < LDS_WORDS; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  uint32_t x = keys[blockIdx.x * blockDim.x + threadIdx.x];
  uint32_t k[16];  // 16 random keys per lane, re-randomised cheaply per iteration (2 VALU per atomic)
  for (int u = 0; u < 16; ++u) { x = x * 1664525u + 1013904223u; k[u] = x >> 8; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      k[u] = k[u] + 0x9E3779u * (uint32_t)(u + 1);
      const uint32_t r = k[u] >> 3;
      uint32_t idx;
      if (PATTERN == 0 || PATTERN == 5) idx = ((r & 1023u) << 5) | (lane & 31u);
      else if (PATTERN == 1) idx = ((r & 2047u) << 4) | (lane & 15u);
      else if (PATTERN == 2) idx = r & 16383u;
      else if (PATTERN == 3) idx = ((r & 4095u) << 3) | (lane & 7u);
      else idx = 7;
      if (PATTERN == 5) lds[idx] = r;
      else __hip_atomic_fetch_add(&lds[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (uint32_t i = threadIdx.x; i < LDS_WORDS; i += blockDim.x) s += lds[i];
  atomicAdd(out, s + x);
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int P>
void run(const char* name, int threads, const uint32_t* d_keys, uint32_t* d_out, unsigned long long* d_cyc, int ncu) {
  const uint32_t iters = 2000;
  const size_t lds = LDS_WORDS * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(bench<P>, dim3(ncu), dim3(threads), lds, 0, d_keys, 10u, d_out, d_cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<P>, dim3(ncu), dim3(threads), lds, 0, d_keys, iters, d_out, d_cyc);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> cyc(ncu);
  CK(hipMemcpy(cyc.data(), d_cyc, ncu * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto c : cyc) avg += (double)c;
  avg /= ncu;
  const double atomics_per_block = (double)threads * iters * 16;
  // s_memtime/readcyclecounter ticks at a constant 100 MHz on gfx9; use wall time with an assumed 2.4 GHz too
  printf("%-34s waves/CU=%2d  %8.3f ms  %7.2f atomics/ns/CU  => %6.2f lanes/clk/CU @2.4GHz  (ticks/block %.0f)\n", name,
         threads / 64, ms, atomics_per_block / (ms * 1e6), atomics_per_block / (ms * 1e6) / 2.4, avg);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
  std::vector<uint32_t> keys((size_t)ncu * 1024);
  uint32_t s = 12345;
  for (auto& k : keys) { s = s * 1103515245u + 12345u; k = s; }
  uint32_t *d_keys, *d_out; unsigned long long* d_cyc;
  CK(hipMalloc(&d_keys, keys.size() * 4)); CK(hipMalloc(&d_out, 4)); CK(hipMalloc(&d_cyc, ncu * 8));
  CK(hipMemcpy(d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
  for (int threads : {256, 512, 1024}) {
    run<0>("ds_add 1024x32 conflict-free", threads, d_keys, d_out, d_cyc, ncu);
    run<1>("ds_add 2048x16 (<=2-way)", threads, d_keys, d_out, d_cyc, ncu);
    run<3>("ds_add 4096x8  (<=4-way)", threads, d_keys, d_out, d_cyc, ncu);
    run<2>("ds_add 16384 random banks", threads, d_keys, d_out, d_cyc, ncu);
    run<4>("ds_add same address", threads, d_keys, d_out, d_cyc, ncu);
    run<5>("ds_write_b32 conflict-free", threads, d_keys, d_out, d_cyc, ncu);
  }
  return 0;
}
