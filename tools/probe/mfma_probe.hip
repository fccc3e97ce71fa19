// Probe: how close can a wave-level fp32 MFMA loop with LDS-fed operands get to the 32x32x2 f32 issue rate?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(256, 3) void probe(float* out, int iters, int rs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 12 * 1024; i += 256) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 a0 = {1.f, 2.f, 3.f, 4.f}, a1 = {0.5f, 0.25f, 0.125f, 1.f};
  float b0 = 1.f + lane, b1 = 2.f + lane;
  const float* xs = lds + 4096 + (lane & 31) + (lane >> 5) * rs;
  for (int it = 0; it < iters; ++it) {
    if (VARIANT >= 2) {
      a0 = *reinterpret_cast<const float4*>(&lds[((it & 7) * 2 + 0) * 256 + lane * 4]);
      a1 = *reinterpret_cast<const float4*>(&lds[((it & 7) * 2 + 1) * 256 + lane * 4]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (VARIANT >= 1) {
        b0 = xs[(e * 2) * rs + (it & 7)];
        b1 = xs[(e * 2) * rs + (it & 7) + 32];
      }
      const float av0 = e == 0 ? a0.x : e == 1 ? a0.y : e == 2 ? a0.z : a0.w;
      const float av1 = e == 0 ? a1.x : e == 1 ? a1.y : e == 2 ? a1.z : a1.w;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, int blocks, float* out) {
  const int iters = 2000, rs = 324;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 12 * 4096, 0, out, 10, rs);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 12 * 4096, 0, out, iters, rs);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-34s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 512, 768, 1024}) {
    run<0>("pure mfma (regs)", blocks, out);
    run<1>("+ B from LDS (ds_read per 4 mfma)", blocks, out);
    run<2>("+ A from LDS (b128 per 16 mfma)", blocks, out);
  }
  return 0;
}
