// Probe: buffer_load_dwordx4 ... lds (LDS-DMA through a raw buffer descriptor) on gfx950.
//   1. lanes whose byte offset lies outside [0, num_records) must deliver zeros to LDS (negative offsets included);
//   2. a wave-instruction writes LDS at (wave-uniform base) + lane * 16.
// Build: hipcc --offload-arch=gfx950 -O3 glds_oob_probe.hip -o glds_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

__global__ void k(const uint4* __restrict__ src, uint4* dst, int n_units, int t0) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  lds[threadIdx.x] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_units * 16, 0x00020000);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + wave * 64), 16, (t0 + wave * 64 + lane) * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  dst[threadIdx.x] = lds[threadIdx.x];
}

int main() {
  const int n = 100;                       // valid units
  std::vector<uint4> h(n + 512);
  for (int i = 0; i < n + 512; ++i) h[i] = make_uint4(i + 1, 2 * i + 1, 3 * i + 1, 0x1000 + i);
  uint4 *d_src, *d_dst;
  hipMalloc(&d_src, (n + 512) * 16);
  hipMalloc(&d_dst, 256 * 16);
  hipMemcpy(d_src, h.data(), (n + 512) * 16, hipMemcpyHostToDevice);
  int bad = 0;
  for (int t0 : {-40, 0, 30, -300}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 256 * 16, 0, d_src, d_dst, n, t0);
    std::vector<uint4> o(256);
    hipMemcpy(o.data(), d_dst, 256 * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 256; ++i) {
      const int t = t0 + i;
      const bool in = t >= 0 && t < n;
      const uint4 e = in ? h[t] : make_uint4(0, 0, 0, 0);
      if (o[i].x != e.x || o[i].y != e.y || o[i].z != e.z || o[i].w != e.w) {
        if (bad < 10) printf("t0 %d unit %d (t %d): got %08x %08x %08x %08x want %08x ...\n", t0, i, t, o[i].x, o[i].y, o[i].z, o[i].w, e.x);
        ++bad;
      }
    }
  }
  printf(bad ? "FAIL: %d mismatches\n" : "OK: out-of-range lanes deliver zeros, LDS image is lane-linear (%d)\n", bad);
  return bad != 0;
}
