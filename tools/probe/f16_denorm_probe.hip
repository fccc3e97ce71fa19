// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950, and what do the packed conversions do?
// build: hipcc --offload-arch=gfx950 -O3 -o f16_denorm_probe f16_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float a_val, float b_val, float* out, float* cv) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
  // split of x into h + l in fp16
  if (threadIdx.x == 0) {
    const f32x2 v = {a_val, b_val};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = {v.x - (float)h.x, v.y - (float)h.y};
    const f16x2 l = __builtin_convertvector(r, f16x2);
    cv[0] = (float)h.x; cv[1] = (float)l.x; cv[2] = (float)h.y; cv[3] = (float)l.y;
  }
}

int main() {
  float *out, *cv;
  hipMalloc(&out, 4); hipMalloc(&cv, 16);
  const float tests[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /* 2^-20 */, 1.0f}, {5.9604644775390625e-08f /* 2^-24 */, 1.0f},
                            {3.0517578125e-05f /* 2^-15 */, 3.0517578125e-05f}, {0.0123456789f, 0.333333333f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t[0], t[1], out, cv);
    float h, c[4];
    hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    hipMemcpy(c, cv, 16, hipMemcpyDeviceToHost);
    printf("a=%.9g b=%.9g: mfma sum of 16 products = %.9g (exact %.9g) | split a: h=%.9g l=%.9g (h+l-a=%.3g)  b: h=%.9g l=%.9g (h+l-b=%.3g)\n",
           t[0], t[1], h, 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1], c[0], c[1],
           (double)c[0] + c[1] - t[0], c[2], c[3], (double)c[2] + c[3] - t[1]);
  }
  return 0;
}
