// LDS-DMA convolution on v_mfma_f32_16x16x32_f16 (conv_gl16_kernel): the 7- and 11-tap staged-input f16x2 convs of the
// generator's ResBlocks (reference modules.py:183-225, the convs1 / convs2 of ResBlock1).
//
// Same data path as conv_gl_kernel (kernels_conv.hip): the packed weight stream and the staged activations go straight
// from memory into LDS by `buffer_load_dwordx4 ... lds`, a counted s_waitcnt and one s_barrier per step, fragments by
// ds_read_b128.  What differs is the matrix instruction.  One 16 x 16 x 32 MFMA does in 4 passes the work two
// 32 x 32 x 16 ones do per 16 x 16 outputs in 8, with half the accumulator traffic per FLOP; on this chip, whose
// matrix throughput under load is rationed by power rather than by issue slots, that is worth 9-13 % on these layers.
//
// The 32-deep contraction of one instruction covers TWO steps of the existing packing, not 32 channels of one tap:
// step s = chunk * K + tap of the flat (16-channel chunk, tap) sequence is what one ring slot of weights and one
// offset into an activation buffer hold, and a "pair step" p consumes the flat steps 2p and 2p + 1.  Lanes 0..31 of
// an operand fragment carry flat step 2p (k-groups 0 and 1: the two 8-channel halves of its chunk), lanes 32..63 flat
// step 2p + 1 -- so the weight packing [Cin/16][K][M/32][2 planes][64 lanes][8 fp16], the staged activation format
// and the DMA requests are exactly those of conv_gl_kernel, and only the fragment addresses differ per lane half:
//   A (weights)      unit (k-half kh, row r) of an m-tile's plane sits at kh * 32 + r; rows r = 16 * (i & 1) + (lane & 15)
//                    of 16-row tile i, kh = (lane >> 4) & 1; lanes 32..63 read the next ring slot (S is even, so the
//                    two slots of a pair step are neighbours);
//   B (activations)  unit (half kh, plane, column); column = tile column + tap * dilation; lanes 32..63 add their own
//                    step's buffer and tap offset (the same buffer one tap on, or the next chunk's buffer at tap 0
//                    when the pair straddles a chunk boundary: K is odd).
// With K odd a trip of the unrolled loop covers four chunks = 2K pair steps, so buffer indices (two buffers), tap
// offsets and the register-set alternation are compile-time constants.
//
// Software pipeline inside a pair step (wave tile 64 x 32 * WN = 4 x 2 * WN accumulator tiles of 16 x 16): the weight
// fragments of the whole step (8) are double-buffered across steps, the activation fragments go by sub-steps of one or
// two 16-column tiles (two sets): the reads of sub-step q + 1 are in flight while the 12 or 24 MFMAs of sub-step q run,
// and the last sub-step's MFMAs run after the next step's barrier, under that step's first reads.
//
// Results differ from conv_gl_kernel / conv_f16x2_kernel in the last bits only (the same products, summed 32 at a time
// instead of 16).
#include "common.h"

#include <cstdio>
#include <cstdlib>

namespace vits {
namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int V>
struct IntC { static constexpr int value = V; };

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int floor_mod(int x, int m) { return ((x % m) + m) % m; }

// Activation tile requested at pair step j of a trip (a trip = four chunks = 2K pair steps, two buffers): the chunk,
// counted from the trip's first, whose buffer the barrier of this step has just released -- chunk c's last reader is
// pair step floor((c * K + K - 1) / 2), so chunk c + 2 goes out at the step after it.  -1: none.
constexpr int x_chunk_at(int j, int K) {
  return j == 0 ? 1 : j == (K + 1) / 2 ? 2 : j == K ? 3 : j == (3 * K + 1) / 2 ? 4 : -1;
}
// DMA instructions a wave has issued after those of pair step j's operands (requested LA pair steps ahead): the
// requests of the LA - 1 pair steps in between
constexpr int dma_issued_since(int j, int K, int LA, int WPK, int APC) {
  int n = 0;
  for (int q = j - LA + 1; q <= j - 1; ++q) n += 2 * WPK + (x_chunk_at(floor_mod(q, 2 * K), K) >= 0 ? APC : 0);
  return n;
}

__device__ __forceinline__ void report_range(int* flag, float amax) {
  if (flag && !(amax <= 65504.f)) atomicOr(flag, 1);
}

// eight consecutive channels of one time step -> the 16-byte units (planes h, l) of the staged f16x2 format
__device__ __forceinline__ void split_f16x2(const float (&v)[8], uint4& oh, uint4& ol) {
  unsigned h4[4], l4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x2 e = {fminf(fmaxf(v[2 * u], -65504.f), 65504.f), fminf(fmaxf(v[2 * u + 1], -65504.f), 65504.f)};
    const f16x2 h = __builtin_convertvector(e, f16x2);
    const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
    h4[u] = __builtin_bit_cast(unsigned, h);
    l4[u] = __builtin_bit_cast(unsigned, l);
  }
  oh = make_uint4(h4[0], h4[1], h4[2], h4[3]);
  ol = make_uint4(l4[0], l4[1], l4[2], l4[3]);
}

// Epilogue for the 16 x 16 accumulator layout.  Tile (i, jn) of lane 16 g + c holds rows 16 i + 4 g + r (r = 0..3) of
// column 16 jn + c: a store of one register would write 64-byte runs.  v_permlane16_swap of the registers of two
// neighbouring column tiles (odd 16-lane rows of the first against even rows of the second) leaves lane 32 h + c32 with
//   first register:  row 16 i + 8 h + r,      column 32 jp + c32
//   second register: row 16 i + 8 h + 4 + r,  column 32 jp + c32
// i.e. 128-byte runs per row and, per lane, the eight consecutive channels of one staged 16-byte unit.
// F_OUT_STAGED: bias, leaky-relu(out_slope), staged f16x2 store.  Otherwise bias, relu, masks, residual, accumulate as
// in the 32 x 32 epilogue, an fp32 store, and optionally a second, staged copy (out2).  The residual and accumulate
// loads of a 16-row block go out before its first store (out may alias res).
template <int NT>
__device__ __forceinline__ void epilogue16(const ConvArgs& a, f32x4 (&acc)[4][NT], int row0, int tw0, int lane, int b) {
  constexpr int NP = NT / 2;                     // 32-column pairs
  const int h = lane >> 5, c32 = lane & 31;
  const bool staged_out = a.flags & F_OUT_STAGED;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int flags = a.flags;
  float* out_b = a.out + (long long)b * a.out_bs;
  const float* res_b = a.res ? a.res + (long long)b * a.res_bs : nullptr;
  uint4* stg_b = staged_out ? reinterpret_cast<uint4*>(a.out + (long long)b * a.out_bs)
                 : a.out2   ? reinterpret_cast<uint4*>(a.out2 + (long long)b * a.out2_bs)
                            : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r0 = row0 + 16 * i + 8 * h;        // the lane's eight rows: r0 .. r0 + 7
    float bz[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bz[r] = a.bias ? a.bias[r0 + r] : 0.f;
    float rv[NP][8], ov[NP][8];
    if (!staged_out && (flags & (F_RES | F_RES_SUB))) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int t = tw0 + 32 * jp + c32;
        const int tc = t < a.L ? t : a.L - 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) rv[jp][r] = res_b[(long long)(r0 + r) * a.res_cs + tc];
      }
    }
    if (!staged_out && (flags & F_ACCUM)) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int t = tw0 + 32 * jp + c32;
        const int tc = t < a.L ? t : a.L - 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) ov[jp][r] = out_b[(long long)(r0 + r) * a.out_cs + tc];
      }
    }
    const long long unit0 = (long long)(r0 >> 3) * 2 * a.L;    // 16-byte unit of (channel group, plane h, t = 0)
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lo = acc[i][2 * jp][r], hi4 = acc[i][2 * jp + 1][r];
        // (the wait states stand in for the hazard checks the compiler does not make inside an asm statement)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi4));
        v[r] = lo;
        v[4 + r] = hi4;
      }
      const int t = tw0 + 32 * jp + c32;
      if (t >= a.L) continue;
      if (staged_out) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bz[r];
      } else {
        const float m = t < len_b ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float e = v[r] + bz[r];
          if (flags & F_RELU) e = fmaxf(e, 0.f);
          if (flags & F_MASK_PRE) e *= m;
          if (flags & F_RES) e += rv[jp][r];
          if (flags & F_RES_SUB) e = rv[jp][r] - e;
          if (flags & F_ACCUM) e += ov[jp][r];
          if (flags & F_MASK_POST) e *= m;
          v[r] = e;
#if defined(GL16_KO) && (GL16_KO & 2)
          if (a.L < 0)
#endif
          out_b[(long long)(r0 + r) * a.out_cs + t] = e;
        }
      }
      if (stg_b) {
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          v[r] = fmaxf(v[r], v[r] * a.out_slope);
          amax = fmaxf(amax, fabsf(v[r]));
        }
        report_range(a.range_flag, amax);
        uint4 oh, ol;
        split_f16x2(v, oh, ol);
#if defined(GL16_KO) && (GL16_KO & 2)
        if (a.L < 0 || (oh.x ^ ol.y) == 0x12345678u)
#endif
        {
          stg_b[unit0 + t] = oh;
          stg_b[unit0 + a.L + t] = ol;
        }
      }
    }
  }
}

}  // namespace

// S: ring slots of one flat step's weights each (even; the look-ahead is (S - 2) / 2 pair steps).
#ifdef GL16_STAMP
__device__ unsigned long long* g_gl16_stamps = nullptr;      // [slots][2]: shader cycles, 100 MHz ticks of a block's main loop
__device__ unsigned long long g_gl16_stamp_slots = 0;
#endif

#ifdef GL16_TIMELINE
__device__ unsigned long long* g_gl16_timeline = nullptr;    // [slots][6]: 100 MHz ticks at block entry, loop start, loop end,
__device__ unsigned long long g_gl16_timeline_slots = 0;     // block end; HW_ID; XCC_ID
#endif

template <int WN, int WGM, int WGN, int K, int S>
__global__ __launch_bounds__(256, 2) void conv_gl16_kernel(const ConvArgs a) {
#ifdef GL16_TIMELINE
  // Diagnostic build only (tools/build_variant.sh timeline kernels_gl16.hip -DGL16_TIMELINE): where a block's time goes.
  const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int SUBN = WN == 2 ? 2 : 1;          // 16-column tiles per sub-step (the wider wave tiles have 96 / 128
                                                 // accumulator registers: two sets of four activation fragments would spill)
  constexpr int NSUB = 2 * WN / SUBN;            // sub-steps per pair step
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  constexpr int NCOLS = BN + MAX_HALO;           // tile columns: a multiple of 64, halo included
  constexpr int WU = BM * 4;                     // 16-byte units of one flat step's weights
  constexpr int WPK = WU / 256;                  // weight DMA instructions per wave and flat step
  constexpr int APC = NCOLS / 64;                // activation DMA instructions per wave and chunk (wave w copies slab w)
  constexpr int LA = (S - 2) / 2;                // pair steps between a request and its use
  constexpr int TRIP = 2 * K;                    // pair steps per loop trip (four chunks)
  constexpr int NT = 2 * WN;                     // 16-column accumulator tiles per wave
  static_assert(WN >= 2 && WN <= 4 && NSUB % 2 == 0, "wave tile 64 x 64, 64 x 96 or 64 x 128; sub-steps alternate two register sets");
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(K % 2 == 1 && K >= 5, "odd tap counts of at least five (an activation tile goes out (K - 1) / 2 pair steps ahead)");
  static_assert(S % 2 == 0 && S >= 4 && LA < (K - 1) / 2, "weights are requested after the activation tile they go with");
  static_assert(WU % 256 == 0 && NCOLS % 64 == 0, "whole wave-instructions");
  static_assert(2 * NCOLS * 16 + 256 < 65536, "fragment offsets fit the 16-bit offset field");
  static_assert(x_chunk_at(TRIP - 1, K) < 0 && x_chunk_at(TRIP - 2, K) < 0, "no activation request in the steps before a trip");

  extern __shared__ __attribute__((aligned(16))) uint4 gl_lds[];
  uint4* wq = gl_lds;                            // [S][BM/32][2 planes][64 units]
  uint4* xq = gl_lds + S * WU;                   // [NBUF][2 halves][2 planes][NCOLS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int b = blockIdx.z;
  int tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map), rowblk = blockIdx.y;
  if (gridDim.y == 2) xcd_tile_rows2(blockIdx.x, blockIdx.y, gridDim.x, a.xcd_map, tile, rowblk);
  const int t0 = tile * BN;
  const int mt0 = rowblk * (BM / 32);
  const int nchunks = a.Cin / 16;
  const int total = nchunks * K;
  const int tstart = t0 + a.min_off;
  const int dil = a.tstep[0];

  // ---- DMA requests (as in conv_gl_kernel) ----
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.Cin * K * a.M * 4, 0x00020000);
  const uint4* xs_b = reinterpret_cast<const uint4*>(a.x + (long long)b * a.x_bs);
  // One vector register per stream serves every request: the weights' per-lane part is lane * 16 with the rest of the
  // address in the scalar offset; the activations' is (tile start + lane) * 16 with the 64-column pieces in the
  // instruction's offset field -- they have to stay on the vector side, which is what the range check looks at.
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned xv0 = (unsigned)((tstart + lane) * 16);
  auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {
    const int sc = s < total ? s : total - 1;
    const int unit0 = (sc * a.mt_total + mt0) * 128;
#pragma unroll
    for (int j = 0; j < WPK; ++j) {
      const int piece = wave + 4 * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wq + slot * WU + piece * 64), 16, lane16,
                                               (unit0 + piece * 64) * 16, 0, 0);
    }
  };
  auto issue_x = [&](int c, int buf) __attribute__((always_inline)) {
    const int cc = c < nchunks ? c : nchunks - 1;
    // this wave's slab: channel group 2 * cc + (wave >> 1), plane wave & 1; its run over time is one buffer, so that
    // t < 0 and t >= L are out of range and read as zero
    const uint4* slab = xs_b + (long long)((cc * 2 + (wave >> 1)) * 2 + (wave & 1)) * a.L;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(slab), 0, a.L * 16, 0x00020000);
    uint4* dst = xq + (buf * 4 + wave) * NCOLS;
    auto piece = [&](auto jj) __attribute__((always_inline)) {
      constexpr int j = decltype(jj)::value;
      if constexpr (j < 4)      // the instruction offset moves the memory address AND the LDS address
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, xv0, 0, 1024 * j, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(dst + 64 * j), 16, xv0 + 1024u * j, 0, 0, 0);
    };
    static_for<0, APC>(piece);
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- the requests of the pair steps before the first one, in the loop's own order (they hold no activation request) ----
  issue_x(0, 0);
#pragma unroll
  for (int v = -LA; v < 0; ++v) {
    issue_w(2 * (v + LA), (2 * (v + LA)) % S);
    issue_w(2 * (v + LA) + 1, (2 * (v + LA) + 1) % S);
  }

  // Fragment addresses (bytes).  Lane = 16 * kg + c15; k-group kg: bit 0 = 8-channel half of the chunk, bit 1 = which
  // flat step of the pair.
  const int c15 = lane & 15, kh = (lane >> 4) & 1, hi = lane >> 5;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void*)gl_lds;
  const unsigned a_lane = lds_base + (unsigned)(((wm * 4) * 64 + kh * 32 + c15 + hi * WU) * 16);
  const unsigned b_lane = lds_base + (unsigned)((S * WU + kh * 2 * NCOLS + wn * 32 * WN + c15) * 16);
  const unsigned himask = hi ? 0xffffffffu : 0u;

  u32x4 A[2][8], B[2][2 * SUBN];                 // A[set][2 * i + plane], B[set][2 * jj + plane]
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) A[0][i] = A[1][i] = z;
#pragma unroll
    for (int i = 0; i < 2 * SUBN; ++i) B[0][i] = B[1][i] = z;
  }
  // Reads are ds_read_b128 statements of our own: a read the compiler can see is made to wait for every LDS-DMA
  // request in flight, which would drain the look-ahead at every step.
  auto read_a = [&](u32x4 (&f)[8], unsigned addr) __attribute__((always_inline)) {
    asm volatile(
        "ds_read_b128 %0, %8\n\t"
        "ds_read_b128 %1, %8 offset:1024\n\t"
        "ds_read_b128 %2, %8 offset:256\n\t"
        "ds_read_b128 %3, %8 offset:1280\n\t"
        "ds_read_b128 %4, %8 offset:2048\n\t"
        "ds_read_b128 %5, %8 offset:3072\n\t"
        "ds_read_b128 %6, %8 offset:2304\n\t"
        "ds_read_b128 %7, %8 offset:3328"
        : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
        : "v"(addr)
        : "memory");
  };
  // the activation fragments of sub-step q: the sub-step's column offset rides in the instruction's offset field, so a
  // pair step has ONE address register
  auto read_b = [&](u32x4 (&f)[2 * SUBN], unsigned addr, auto qq) __attribute__((always_inline)) {
    constexpr int o = decltype(qq)::value * SUBN * 256;
    if constexpr (SUBN == 2)
      asm volatile(
          "ds_read_b128 %0, %4 offset:%c5\n\t"
          "ds_read_b128 %1, %4 offset:%c6\n\t"
          "ds_read_b128 %2, %4 offset:%c7\n\t"
          "ds_read_b128 %3, %4 offset:%c8"
          : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
          : "v"(addr), "i"(o), "i"(o + NCOLS * 16), "i"(o + 256), "i"(o + NCOLS * 16 + 256)
          : "memory");
    else
      asm volatile(
          "ds_read_b128 %0, %2 offset:%c3\n\t"
          "ds_read_b128 %1, %2 offset:%c4"
          : "=&v"(f[0]), "=&v"(f[1])
          : "v"(addr), "i"(o), "i"(o + NCOLS * 16)
          : "memory");
  };
  // every LDS read issued so far has landed; names the registers so that nothing uses them before
  auto wait_reads = [&](u32x4 (&fa)[8], u32x4 (&fb)[2 * SUBN]) __attribute__((always_inline)) {
    if constexpr (SUBN == 2)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                     "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                     "+v"(fb[0]), "+v"(fb[1])
                   :
                   : "memory");
  };
  // the three products of sub-step q (SUBN 16-column tiles): h * l, h * h, l * h
  auto mma_sub = [&](const u32x4 (&fa)[8], const u32x4 (&fb)[2 * SUBN], auto qq) __attribute__((always_inline)) {
    constexpr int q = decltype(qq)::value;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
      const int pa = pr == 2 ? 1 : 0, pb = pr == 0 ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < SUBN; ++jj)
          acc[i][SUBN * q + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[2 * i + pa]),
                                                                         __builtin_bit_cast(f16x8, fb[2 * jj + pb]),
                                                                         acc[i][SUBN * q + jj], 0, 0, 0);
    }
  };

#ifdef GL16_STAMP
  // Diagnostic build only (tools/build_variant.sh stamp kernels_gl16.hip -DGL16_STAMP): the shader clock the chip holds
  // inside the main loop = shader-cycle counter / 100 MHz wall counter, stamped once around the loop.  The values go to a
  // buffer of their own that nothing else reads; no output depends on them.
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef GL16_TIMELINE
  const unsigned long long tl_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
  int slot_rd = 0;                               // ring slot of flat step 2p (even); the slots of p - 1 are written
  int p = 0;                                     // pair step
  for (int c0 = 0; c0 < nchunks; c0 += 4) {
    auto step = [&](auto jj) __attribute__((always_inline)) {
      constexpr int j = decltype(jj)::value;
      constexpr int ca = (2 * j) / K, ka = (2 * j) % K;            // flat step 2p: chunk (of the trip) and tap
      constexpr int cb = (2 * j + 1) / K, kb = (2 * j + 1) % K;    // flat step 2p + 1
      constexpr int allowed = dma_issued_since(j, K, LA, WPK, APC);
      constexpr int xc = x_chunk_at(j, K);
      u32x4(&Acur)[8] = A[j & 1];
      u32x4(&Aprev)[8] = A[(j & 1) ^ 1];
      // this wave's DMA share of the step has landed, and so have the previous step's last reads
      if constexpr (SUBN == 2)
        asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                     : "+v"(Aprev[0]), "+v"(Aprev[1]), "+v"(Aprev[2]), "+v"(Aprev[3]), "+v"(Aprev[4]), "+v"(Aprev[5]),
                       "+v"(Aprev[6]), "+v"(Aprev[7]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[1][2]), "+v"(B[1][3])
                     : "n"(allowed)
                     : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%10) lgkmcnt(0)"
                     : "+v"(Aprev[0]), "+v"(Aprev[1]), "+v"(Aprev[2]), "+v"(Aprev[3]), "+v"(Aprev[4]), "+v"(Aprev[5]),
                       "+v"(Aprev[6]), "+v"(Aprev[7]), "+v"(B[1][0]), "+v"(B[1][1])
                     : "n"(allowed)
                     : "memory");
      __builtin_amdgcn_s_barrier();
      // per-lane offset of this step's activation units: buffer and tap of the lane half's own flat step.  The tap
      // offsets are products with the dilation: taken as new at every step, they are computed here (scalar work) instead
      // of being kept in 2K vector registers for the whole trip
      int dilv = dil;
      asm volatile("" : "+s"(dilv));
      const unsigned off_lo = (unsigned)(((ca & 1) * 4 * NCOLS + ka * dilv) * 16);
      const unsigned off_hi = (unsigned)(((cb & 1) * 4 * NCOLS + kb * dilv) * 16);
      const unsigned xk = b_lane + off_lo + (himask & (off_hi - off_lo));
      read_a(Acur, a_lane + (unsigned)(slot_rd * WU * 16));
      read_b(B[0], xk, IntC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      if (j > 0 || c0 > 0) mma_sub(Aprev, B[1], IntC<NSUB - 1>{});
      __builtin_amdgcn_sched_barrier(0);
      // requests of pair step p + LA go into the slots pair step p - 1 has just left; they go after the MFMAs because
      // a vector-memory request can hold the wave at issue until the load path has room
      {
        const int wr = slot_rd == 0 ? S - 2 : slot_rd - 2;
        if constexpr (xc >= 0) issue_x(c0 + xc, xc & 1);
        issue_w(2 * (p + LA), wr);
        issue_w(2 * (p + LA) + 1, wr + 1);
      }
      auto sub = [&](auto qq) __attribute__((always_inline)) {
        constexpr int q = decltype(qq)::value;
        wait_reads(Acur, B[q & 1]);
        read_b(B[(q + 1) & 1], xk, IntC<q + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mma_sub(Acur, B[q & 1], IntC<q>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      static_for<0, NSUB - 1>(sub);
      ++p;
      slot_rd = slot_rd + 2 == S ? 0 : slot_rd + 2;
    };
    static_for<0, TRIP>(step);
  }
  // the last sub-step of the last pair step (TRIP is even: its weights are in the second set)
  wait_reads(A[1], B[1]);
  __builtin_amdgcn_sched_barrier(0);
  mma_sub(A[1], B[1], IntC<NSUB - 1>{});
#ifdef GL16_STAMP
  {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* sb = g_gl16_stamps;
    if (sb && threadIdx.x == 0) {
      const unsigned long long blk = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
      if (blk < g_gl16_stamp_slots) {
        sb[2 * blk] = st1 - st0;
        sb[2 * blk + 1] = rt1 - rt0;
      }
    }
  }
#endif
#ifdef GL16_TIMELINE
  const unsigned long long tl_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
  // nothing of this block may still be on its way into LDS when the block's LDS is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#if defined(GL16_KO) && (GL16_KO & 1)
  if (a.L < 0)                                   // knock-out (measurement only): no epilogue at all
#endif
  epilogue16<NT>(a, acc, (mt0 + wm * 2) * 32, t0 + wn * 32 * WN, lane, b);
#ifdef GL16_TIMELINE
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the epilogue's stores have been accepted
    const unsigned long long tl_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long* sb = g_gl16_timeline;
    if (sb && threadIdx.x == 0) {
      const unsigned long long blk = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
      if (blk < g_gl16_timeline_slots) {
        sb[6 * blk] = tl_entry;
        sb[6 * blk + 1] = tl_loop0;
        sb[6 * blk + 2] = tl_loop1;
        sb[6 * blk + 3] = tl_end;
        sb[6 * blk + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        sb[6 * blk + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
      }
    }
  }
#endif
}

namespace {

bool gl16_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_CONV_GL16");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <int WN, int WGM, int WGN, int K, int S>
hipError_t launch_cfg(ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  constexpr size_t lds = ((size_t)S * BM * 4 + (size_t)2 * 4 * (BN + MAX_HALO)) * sizeof(uint4);
  static_assert(lds <= 80 * 1024, "two blocks to a CU");
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(&conv_gl16_kernel<WN, WGM, WGN, K, S>), lds); e != hipSuccess)
    return e;
  dim3 grid((a.L + BN - 1) / BN, a.M / BM, a.B);
  char name[96];
  snprintf(name, sizeof name, "conv_gl16_kernel<%d,%d,%d,%d,%d>", WN, WGM, WGN, K, S);
  const ProfTicket tk = prof_begin(stream);
  hipLaunchKernelGGL((conv_gl16_kernel<WN, WGM, WGN, K, S>), grid, dim3(256), lds, stream, a);
  const hipError_t err = hipGetLastError();
  prof_end(tk, name, false, a, stream);
  return err;
}

}  // namespace

// The 7- and 11-tap shapes of the LDS-DMA conv (the launcher in kernels_conv.hip has already checked everything the
// two kernels share: staged input, rows % 64, grid size).
bool conv_gl16_takes(const ConvArgs& a) {
  return gl16_enabled() && (a.K == 7 || a.K == 11) && a.Cin % 64 == 0;
}

// Block tile width for a 128-row-block launch: 256, 192 or 128 time steps.  A launch is a number of rounds of the chip's
// 2 * CUs block slots, and a round costs a block's time whether the round is full or not: 16 rows of 16 512 steps are
// 1040 tiles of 256 on 1024 slots -- a third round for 16 blocks -- but 1376 tiles of 192, three rounds of blocks three
// quarters the size.  Cost = rounds * (width + 48): the 48 stands for a block's fixed part (prologue, epilogue, the
// weight stream, which does not shrink with the tile).  VITS_CONV_GL16_BN=<width> fixes the choice (A/B measurements).
int conv_gl16_tile_width(const ConvArgs& a) {
  static const int fixed = [] {
    const char* e = getenv("VITS_CONV_GL16_BN");
    return e && e[0] ? atoi(e) : 0;
  }();
  if (fixed == 128 || fixed == 192 || fixed == 256) return fixed;
  const long long slots = 2LL * device_cu_count();
  int best = 0;
  long long best_cost = 0;
  for (int bn : {256, 192, 128}) {
    const long long blocks = (long long)((a.L + bn - 1) / bn) * (a.M / 128) * a.B;
    const long long cost = ((blocks + slots - 1) / slots) * (bn + 48);
    if (!best || cost < best_cost) { best = bn; best_cost = cost; }
  }
  return best;
}

// bn: block tile width for 128-row blocks (M % 128 == 0), 0 for 64-row blocks (64 x 256 tiles)
hipError_t launch_conv_gl16(ConvArgs& a, int bn, hipStream_t stream) {
  if (bn == 256) return a.K == 7 ? launch_cfg<4, 2, 2, 7, 4>(a, stream) : launch_cfg<4, 2, 2, 11, 4>(a, stream);
  if (bn == 192) return a.K == 7 ? launch_cfg<3, 2, 2, 7, 4>(a, stream) : launch_cfg<3, 2, 2, 11, 4>(a, stream);
  if (bn == 128) return a.K == 7 ? launch_cfg<2, 2, 2, 7, 6>(a, stream) : launch_cfg<2, 2, 2, 11, 6>(a, stream);
  return a.K == 7 ? launch_cfg<2, 1, 4, 7, 6>(a, stream) : launch_cfg<2, 1, 4, 11, 6>(a, stream);
}

}  // namespace vits

#ifdef GL16_TIMELINE
// Diagnostic build only: where the timeline goes (a device buffer of 6 * slots 64-bit words the caller owns).
extern "C" int vits_debug_gl16_timeline_buffer(void* buf, unsigned long long slots) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(vits::g_gl16_timeline), &p, sizeof p) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(vits::g_gl16_timeline_slots), &slots, sizeof slots) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef GL16_STAMP
// Diagnostic build only: where the stamps go (a device buffer of 2 * slots 64-bit words the caller owns).
extern "C" int vits_debug_gl16_stamp_buffer(void* buf, unsigned long long slots) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(vits::g_gl16_stamps), &p, sizeof p) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(vits::g_gl16_stamp_slots), &slots, sizeof slots) != hipSuccess) return -1;
  return 0;
}
#endif
