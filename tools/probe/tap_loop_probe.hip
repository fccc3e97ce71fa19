// Probe: the bf16x3 tap loop in isolation.  Each wave runs the consumer loop of the conv kernel (6 weight fragments
// per tap from global/L2, 6 activation fragments per tap from LDS, 24 MFMAs) with no staging, barriers or epilogue.
// Variants switch the weight loads and the LDS reads off; the grid sets the number of waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tap_loop_probe tap_loop_probe.hip ; run: ./tap_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct C0 { static constexpr int value = 0; };
struct C1 { static constexpr int value = 1; };

template <int LOAD_A, int LOAD_B, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(const uint4* __restrict__ w, float* out, int n_it, int wfrags, int randomize) {
  extern __shared__ __attribute__((aligned(16))) uint4 xq[];
  const int lane = threadIdx.x & 63;
  const int ncols = 320;
  for (int i = threadIdx.x; i < 6 * ncols + 4 * 384; i += 256) {
    if (randomize) {     // bf16 pairs with random sign / mantissa and exponents near 1.0
      unsigned h = (i + 1) * 2654435761u;
      unsigned v[4];
      for (int q = 0; q < 4; ++q) { h = h * 1664525u + 1013904223u; v[q] = (h & 0x807f807fu) | 0x3f003f00u; }
      xq[i] = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
      xq[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    }
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  uint4 Ah[2][2], Am[2][2], Al[2], Bh[2][2], Bm[2][2], Bl[2];
  const int wave = threadIdx.x >> 6;
  const unsigned wofs0 = lane, wofs1 = 192 + lane;
  const int colbase = (wave & 3) * 64 + (lane & 31) + (lane >> 5) * ncols;
  auto lda = [&](uint4 (&d)[2], long long base, int plane) {
    if (LOAD_A == 2) {
      const int slot = 6 * ncols + (int)(base / 384 & 3) * 384 + plane * 64;
      d[0] = xq[slot + wofs0];
      d[1] = xq[slot + wofs1];
    } else {
      d[0] = (w + base + plane * 64)[wofs0];
      d[1] = (w + base + plane * 64)[wofs1];
    }
  };
  auto ldb = [&](uint4 (&d)[2], int plane, int off) {
    d[0] = xq[plane * 2 * ncols + off];
    d[1] = xq[plane * 2 * ncols + off + 32];
  };
  auto mma = [&](const uint4 (&af)[2], const uint4 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                            __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
  };
  lda(Ah[0], 0, 0); lda(Am[0], 0, 1); lda(Al, 0, 2);
  lda(Ah[1], 0, 0); lda(Am[1], 0, 1);
  ldb(Bh[0], 0, colbase); ldb(Bm[0], 1, colbase); ldb(Bl, 2, colbase);
  ldb(Bh[1], 0, colbase); ldb(Bm[1], 1, colbase);
  int it = 0, boff = colbase, k = 0;
  auto step = [&](auto P) {
    constexpr int c = decltype(P)::value, n = c ^ 1;
    const int itn = it + 1 == wfrags ? 0 : it + 1;
    const long long wb = (long long)itn * 384;       // two m-tiles, 3 planes, 64 lanes per tap
    const int boffn = k == 10 ? colbase : boff + 1;
    if (LOAD_A) { lda(Ah[n], wb, 0); lda(Am[n], wb, 1); }
    __builtin_amdgcn_sched_barrier(0);
    mma(Ah[c], Bl);
    __builtin_amdgcn_sched_barrier(0);
    if (LOAD_B) { ldb(Bh[n], 0, boffn); ldb(Bm[n], 1, boffn); ldb(Bl, 2, boffn); }
    __builtin_amdgcn_sched_barrier(0);
    mma(Al, Bh[c]);
    __builtin_amdgcn_sched_barrier(0);
    if (LOAD_A) lda(Al, wb, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(Am[c], Bm[c]); mma(Ah[c], Bm[c]); mma(Am[c], Bh[c]); mma(Ah[c], Bh[c]);
    __builtin_amdgcn_sched_barrier(0);
    it = itn; boff = boffn; k = k == 10 ? 0 : k + 1;
  };
  for (int s = 0; s < n_it; s += 2) { step(C0{}); step(C1{}); }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int LA, int LB, int WPS>
void run(const char* name, const uint4* w, float* out, int wfrags, int randomize) {
  const int n_it = 16384, blocks = 256 * WPS;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<LA, LB, WPS>), dim3(blocks), dim3(256), (6 * 320 + 4 * 384) * 16, 0, w, out, 64, wfrags, randomize);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<LA, LB, WPS>), dim3(blocks), dim3(256), (6 * 320 + 4 * 384) * 16, 0, w, out, n_it, wfrags, randomize);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * n_it * 24 * 32768.0;
  printf("%s %-28s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s bf16 (%5.1f fp32-equivalent)\n", randomize ? "random" : "zeros ", name, WPS, ms, flops / ms / 1e9,
         flops / ms / 1e9 / 6);
}

int main() {
  const int wfrags = 88;    // 8 chunks x 11 taps: 88 x 6 KiB = 528 KiB of weight fragments
  uint4* w; float* out;
  (void)hipMalloc(&w, (size_t)wfrags * 384 * 16);
  (void)hipMemset(w, 0, (size_t)wfrags * 384 * 16);
  (void)hipMalloc(&out, 1024 * 256 * 4);
  for (int randomize = 0; randomize < 2; ++randomize) {
    if (randomize) {
      std::vector<unsigned> hw((size_t)wfrags * 384 * 4);
      unsigned h = 12345u;
      for (auto& x : hw) { h = h * 1664525u + 1013904223u; x = (h & 0x807f807fu) | 0x3c003c00u; }
      (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    }
    run<0, 0, 1>("mfma only", w, out, wfrags, randomize);
    run<0, 0, 3>("mfma only", w, out, wfrags, randomize);
    run<1, 1, 1>("+ A and B", w, out, wfrags, randomize);
    run<1, 1, 2>("+ A and B", w, out, wfrags, randomize);
    run<1, 1, 3>("+ A and B", w, out, wfrags, randomize);
    run<2, 1, 1>("+ A and B both LDS", w, out, wfrags, randomize);
    run<2, 1, 2>("+ A and B both LDS", w, out, wfrags, randomize);
    run<2, 1, 3>("+ A and B both LDS", w, out, wfrags, randomize);
  }
  return 0;
}
