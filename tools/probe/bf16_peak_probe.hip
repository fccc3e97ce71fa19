// Probe: sustained rate of v_mfma_f32_32x32x16_bf16 from registers, and the shader clock the chip actually holds
// while doing it (s_memtime ticks per 100 MHz s_memrealtime tick).
// build: hipcc --offload-arch=gfx950 -O3 -o bf16_peak_probe bf16_peak_probe.hip ; run: ./bf16_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 3) void probe(float* out, long long* clk, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  float* out; long long* clk;
  (void)hipMalloc(&out, 4096 * 256 * 4);
  (void)hipMalloc(&clk, 4096 * 16);
  long long h[2];
  for (int blocks : {256, 512, 768, 1024, 2048}) {
    for (int iters : {20000, 200000}) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, clk, 10);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
      printf("blocks %5d iters %7d  %9.3f ms  %8.1f TFLOP/s   s_memtime/s_memrealtime = %.2f (x100 MHz)\n", blocks, iters,
             ms, flops / ms / 1e9, (double)h[0] / (double)h[1]);
    }
  }
  return 0;
}
