// Do L2-resident loads of one wave wait behind HBM loads of other waves of the same CU?
// Three "consumer" waves per block time dependent 16-byte loads from a 64 KiB table (L2 hits after the first pass);
// the fourth wave either idles (mode 0) or streams a large buffer from HBM with many loads in flight (mode 1), or
// streams the same small table (mode 2: same instruction mix, no HBM).  Prints the consumers' mean load latency.
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_order_probe vmem_order_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ table, int table_units, const uint4* __restrict__ big,
                                             long long big_units, int iters, int mode, unsigned long long* cycles,
                                             unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < 3) {
    unsigned idx = (blockIdx.x * 197 + wave * 61 + lane) % table_units;
    unsigned long long total = 0;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      const uint4 v = table[idx];
      acc += v.x;                                            // dependent use: waits for the load
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = __builtin_readcyclecounter();
      total += t1 - t0;
      idx = (idx * 1103515245u + 12345u + (acc & 1)) % table_units;
    }
    if (lane == 0) atomicAdd(cycles, total);
    if (acc == 0x12345678u) sink[0] = acc;
  } else if (mode != 0) {
    // streamer: 16 independent 16-byte loads in flight per lane, then use them
    unsigned acc = 0;
    const long long span = mode == 1 ? big_units : table_units;
    const uint4* src = mode == 1 ? big : table;
    long long pos = ((long long)blockIdx.x * 4099 * 64 + lane) % span;
    for (int i = 0; i < iters / 4; ++i) {
      uint4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(pos + (long long)u * 64 * 1021) % span];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u].y;
      pos = (pos + 16LL * 64 * 1021 + 7919) % span;
    }
    if (acc == 0x12345678u) sink[1] = acc;
  }
}

int main() {
  const int table_units = 4096;                              // 64 KiB
  const long long big_units = (1LL << 30) / 16 * 2;          // 2 GiB
  uint4 *table, *big;
  unsigned long long* cycles;
  unsigned* sink;
  hipMalloc(&table, table_units * 16);
  hipMalloc(&big, big_units * 16);
  hipMalloc(&cycles, 8);
  hipMalloc(&sink, 8);
  hipMemset(table, 1, table_units * 16);
  hipMemset(big, 1, big_units * 16);
  const int blocks = 256 * 3, iters = 4000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cycles, 0, 8);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, table, table_units, big, big_units, iters, mode, cycles, sink);
      hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cycles, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): mean latency of an L2-resident 16-byte load = %.0f shader-clock cycles\n", mode,
           mode == 0 ? "other wave idle" : mode == 1 ? "other wave streams HBM" : "other wave streams the L2-resident table",
           (double)c / ((double)blocks * 3 * iters));
  }
  return 0;
}
