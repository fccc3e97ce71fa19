// Probe: sustained MFMA rate from registers on random operands, by data type (fp16 / bf16) and shape
// (32x32x16 / 16x16x32), with the shader clock the chip holds meanwhile (s_memtime ticks per 100 MHz s_memrealtime tick).
// Operands are random bit patterns of normal magnitude (every lane different), refreshed from a small table so that
// the values do not repeat.  One wave per SIMD (256 blocks x 256 threads) and two.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: f16 32x32x16, 1: bf16 32x32x16, 2: f16 16x16x32, 3: bf16 16x16x32
__global__ __launch_bounds__(256, 2) void probe(const u32x4* __restrict__ tab, float* out, long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = tab[(lane + 64 * i) & 1023]; b[i] = tab[(lane * 7 + 64 * i + 13) & 1023]; }
  f32x16 acc[4];
  f32x4 acc4[16];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND < 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (KIND == 0)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i * 2 + j], 0, 0, 0);
          else
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i * 2 + j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (KIND == 2)
            acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc4[i * 4 + j], 0, 0, 0);
          else
            acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc4[i * 4 + j], 0, 0, 0);
        }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int KIND>
void run(const char* name, const u32x4* tab, float* out, long long* clk, int blocks) {
  // per iteration: KIND < 2: 4 MFMAs of 32768 FLOP; else 16 MFMAs of 16384 FLOP
  const double flop_it = KIND < 2 ? 4 * 32768.0 : 16 * 16384.0;
  const int iters = KIND < 2 ? 100000 : 50000;
  long long h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, tab, out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-16s blocks %4d rep %d %8.2f ms %8.1f TFLOP/s  clock %.2f GHz\n", name, blocks, rep, ms,
           (double)blocks * 4 * iters * flop_it / ms / 1e9, (double)h[0] / (double)h[1] / 10.0);
  }
}

int main() {
  u32x4* tab; float* out; long long* clk;
  (void)hipMalloc(&tab, 1024 * 16);
  (void)hipMalloc(&out, 4096 * 256 * 4);
  (void)hipMalloc(&clk, 4096 * 16);
  // random fp16-pair bit patterns with exponents near 1.0 (valid, normal, both signs) -- also sane as bf16 pairs
  unsigned hbuf[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) {
    unsigned lo = (rand() & 0x83ff) | 0x3800 | ((rand() & 3) << 10), hi = (rand() & 0x83ff) | 0x3800 | ((rand() & 3) << 10);
    hbuf[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(tab, hbuf, sizeof hbuf, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<0>("f16 32x32x16", tab, out, clk, blocks);
    run<1>("bf16 32x32x16", tab, out, clk, blocks);
    run<2>("f16 16x16x32", tab, out, clk, blocks);
    run<3>("bf16 16x16x32", tab, out, clk, blocks);
  }
  return 0;
}
