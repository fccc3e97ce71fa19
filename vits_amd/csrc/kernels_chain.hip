// A whole ResBlock in one launch (gfx950, wave64): x -> x + c2(lrelu(c1(lrelu(x)))), once per dilation, with the fp32
// carrier of a time tile kept in registers across the pairs (reference modules.py:206-219, ResBlock1.forward; its
// output joins the stage's sum as in models.py:292-298).
//
// The 32- and 64-channel stages of the generator are bandwidth-bound: a ResBlock done pair by pair reads and writes its
// carrier three times (resblock_pair_kernel), done conv by conv six times.  Here a block owns W1 consecutive time
// steps: it reads x once, runs every pair on the tile -- the operand of each conv (leaky-relu'd, converted, zero outside
// the sequence, as the conv's zero padding requires) goes through ONE LDS tile that the x operand and the intermediate
// take turns in -- and writes the result once.  A tile loses (K - 1) / 2 * (dilation + 1) columns on either side per
// pair to its neighbours' data, R in all; the block stores the inner W1 - 2 R columns and its neighbours recompute the
// rest (12 columns a side for 3 taps, 36 for 7, 60 for 11 with dilations 1, 3, 5).
//
// Everything lives in the accumulator layout of v_mfma_f32_32x32x16 (lane = column l31 of an n-tile, rows
// 8 g + 4 par + q for register 4 g + q): the carrier, the global loads and stores (runs of 32 time steps per row), and
// the operand writes, which are 8-byte pieces -- four consecutive channels of a column, the two lane halves completing
// a 16-byte unit.  No value changes lanes, so there is no transposition through LDS.
//
// Every output sees the same products in the same order as in resblock_pair_kernel and in the two-launch pair (chunk by
// chunk, tap by tap; per tap h*l, h*h, l*h in f16x2), then + bias, + x, and for the last pair + the stage's sum: the
// same bytes as the pair-by-pair path.
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace vits {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Measurement builds (tools/build_variant.sh): -DCHAIN_KO=<bits> removes a part of the kernel to see what the rest
// costs -- 1: no MFMAs, 2: no weight loads inside the tap loops, 4: operand tiles written without the conversion
// arithmetic, 8: no fragment reads from LDS, 16: global loads and stores of NJ consecutive steps a lane (the same bytes in
// 8- / 16-byte pieces; with 15, the memory skeleton alone: 3.7 -> 5.2 TB/s at 32 channels in bf16, but 4 % on the whole
// kernel, whose conversion / LDS / MFMA phases take as long as the skeleton).  Results are wrong in such a build; the
// default build has none of it.
#ifndef CHAIN_KO
#define CHAIN_KO 0
#endif
constexpr int CHAIN_MAX_PAIRS = 4;
template <int V>
struct IntC { static constexpr int value = V; };

struct ChainArgs {
  const float* x; long long x_bs; int x_cs;
  float* out; long long out_bs; int out_cs;
  int B, L;
  int accumulate;                  // out += result (read-modify-write)
  float slope;                     // leaky-relu slope before every conv
  int* range_flag;                 // f16x2: OR 1 into it when an in-sequence activation was clamped to the fp16 range
  int xcd_map;
  int npair;                       // pairs in the chain
  int dil[CHAIN_MAX_PAIRS];        // dilation of each pair's first conv (the second conv's is 1)
  int R;                           // columns a tile loses on each side: sum over the pairs of (K - 1) / 2 * (dil + 1)
  int H;                           // zero columns on each side of the operand tile: (K - 1) / 2 * the largest dilation
  const uint4* w1[CHAIN_MAX_PAIRS]; const float* b1[CHAIN_MAX_PAIRS];
  const uint4* w2[CHAIN_MAX_PAIRS]; const float* b2[CHAIN_MAX_PAIRS];
};

// weight fragments kept in flight: a multiple that divides the 2 * NT taps of a pair, so that a tap's ring slot does
// not depend on the pair it belongs to
constexpr int chain_ring(int nt2, int want) {
  int d = want;
  while (nt2 % d) --d;
  return d;
}

// MT = channels / 32.  NP: operand planes (2: f16x2, fp16 h / l, three products a tap; 1: plain bf16, one product).
// WR: 32-row tiles per wave (MT / WR waves side by side in rows, the rest of the block's four side by side in time).
// NJ: 32-column tiles per wave.  The block's tile is W1 = 32 NJ * (waves in time) steps wide.
// NW: waves per block.  With one wave (32 channels: the wave holds every channel of its columns) a block synchronises
// with nobody, and the waves of a SIMD drift apart: one converts operands while another multiplies.
// BPC: waves per SIMD the register budget is set for.
template <int MT, int K, int NP, int WR, int NJ, int BPC, int NW = 4>
__global__ __launch_bounds__(64 * NW, BPC) void resblock_chain_kernel(const ChainArgs a) {
  static_assert(MT % WR == 0 && NW % (MT / WR) == 0, "whole waves in rows and in time");
  constexpr int WGM = MT / WR, WGN = NW / WGM, CW = 32 * NJ, W1 = CW * WGN, NCH = 2 * MT;
  constexpr int NT = NCH * K;                               // taps per conv
  // depth of the weight ring: what the registers of the build leave room for
#ifndef CHAIN_RING3
#define CHAIN_RING3 4
#endif
#ifndef CHAIN_RING128
#define CHAIN_RING128 2
#endif
  constexpr int D = chain_ring(2 * NT, BPC >= 3 ? (NP == 2 ? CHAIN_RING3 : 4) : (MT == 4 && NP == 2) ? CHAIN_RING128 : WR == 2 ? 4 : (NJ == 4 && NP == 2) ? 3 : 8);
  static_assert((2 * NT) % D == 0 && D >= 2, "ring slots must line up across pairs");
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int BN = W1 - 2 * a.R;
  const int t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int COLS = W1 + 2 * a.H;                            // operand tile [NCH chunks][NP planes][2 halves][COLS] of 16 bytes
  const int c0 = wn * CW + l31;                             // this lane's column in its n-tile 0 (+ 32 j)
  const int tl = t0 - a.R + c0;                             // ... and its time step
  constexpr int h2 = (K - 1) / 2;

  // every conv's bias goes to LDS once (behind the operand tile: [pair][conv][32 MT] floats); read back in the
  // accumulator layout, four consecutive rows a read
  float* bias_lds = reinterpret_cast<float*>(lds + NCH * NP * 2 * COLS);
  for (int i = tid; i < a.npair * 2 * 32 * MT; i += 64 * NW) {
    const int pc = i / (32 * MT), row = i % (32 * MT);
    const float* bp = (pc & 1) ? a.b2[pc >> 1] : a.b1[pc >> 1];
    bias_lds[i] = bp ? bp[row] : 0.f;
  }
  // zero columns on either side of the tile: what a conv reads beyond the tile's edge (results there are the
  // neighbours' to compute, but they must not be garbage: they are range-checked and recirculate)
  for (int i = tid; i < 2 * a.H * NCH * NP * 2; i += 64 * NW) {
    const int row = i / (2 * a.H), c = i % (2 * a.H);
    lds[row * COLS + (c < a.H ? c : W1 + c)] = make_uint4(0, 0, 0, 0);
  }

  // weight ring: tap g of the virtual sequence [first conv's NT taps, second conv's NT taps, the next pair's first
  // taps] lives in slot g % D
  uint4 Aq[D][WR][NP];
  const uint4 *wa = a.w1[0], *wb = a.w2[0], *wc = a.w1[a.npair > 1 ? 1 : 0];
  // (a wave-uniform base per tap plus one 32-bit lane offset: the addresses live in scalar registers)
  const unsigned lane_bytes = lane * 16u;
  auto fetch_a = [&](int g) __attribute__((always_inline)) {
    const uint4* base = g < NT ? wa : g < 2 * NT ? wb : wc;
    const int tap = g < NT ? g : g < 2 * NT ? g - NT : g - 2 * NT;
    const char* wk = reinterpret_cast<const char*>(base + ((long long)tap * MT + wm * WR) * (64 * NP));
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        Aq[g % D][i][p] = *reinterpret_cast<const uint4*>(wk + 1024 * (i * NP + p) + lane_bytes);
  };
#pragma unroll
  for (int g = 0; g < D; ++g) fetch_a(g);

  // the carrier: this wave's rows of its columns, in the accumulator layout, from one round trip to memory
  float xr[WR][NJ][16];
  if constexpr (CHAIN_KO & 16) {                            // mock: the same bytes with NJ consecutive steps a lane
    typedef float fvec __attribute__((ext_vector_type(NJ)));
    const float* xb = a.x + (long long)b * a.x_bs;
    int t = t0 - a.R + wn * CW + NJ * l31;
    t = t >= 0 && t + NJ <= a.L ? t : 0;
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const fvec v = *reinterpret_cast<const fvec*>(xb + (long long)(32 * (wm * WR + i) + 4 * par + (r & 3) + 8 * (r >> 2)) * a.x_cs + t);
#pragma unroll
        for (int j = 0; j < NJ; ++j) xr[i][j][r] = v[j];
      }
  } else {
    const float* xb = a.x + (long long)b * a.x_bs;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int t = tl + 32 * j;
      const bool ok = t >= 0 && t < a.L;                    // outside the sequence the carrier starts at zero and is
      const int tc = ok ? t : 0;                            // never used: operands are zeroed there, nothing is stored
#pragma unroll
      for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = xb[(long long)(32 * (wm * WR + i) + 4 * par + (r & 3) + 8 * (r >> 2)) * a.x_cs + tc];
          xr[i][j][r] = ok ? v : 0.f;
        }
    }
  }

  f32x16 acc[WR][NJ];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto mma = [&](const uint4 (&af)[WR][NP], int pa, const auto& bf, int j0) __attribute__((always_inline)) {
    constexpr int NB = sizeof(bf) / sizeof(uint4);          // n-tiles j0 .. j0 + NB - 1
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        const int j = j0 + jj;
        if constexpr (CHAIN_KO & 1) {
          acc[i][j][0] += __builtin_bit_cast(float, af[i][pa].x ^ bf[jj].x);
        } else if constexpr (NP == 2)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][pa]),
                                                             __builtin_bit_cast(f16x8, bf[jj]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][pa]),
                                                              __builtin_bit_cast(bf16x8, bf[jj]), acc[i][j], 0, 0, 0);
      }
  };
  // LDS addresses: one vector register per conv (the lane's column and half, made opaque so that nothing derived from
  // it is computed ahead and kept), plus per tap a wave-uniform offset -- chunk, and for the dilated conv tap * stride,
  // a run-time value made opaque at every tap -- added where the tap is read; n-tile, and for the undilated conv the
  // tap, sit in the instruction's offset field.  Left to itself the compiler keeps an address register per tap, plane
  // and half alive across the whole loop over the pairs (2 to 4 K of them) and spills the carrier to make room.
  auto conv = [&](auto first_tap, auto dilated, int reach, int step) __attribute__((always_inline)) {
    constexpr int g0 = decltype(first_tap)::value;
    int base = par * COLS + a.H - reach + c0;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int g = g0 + chunk * K + k;
        int uni = chunk * (2 * NP) * COLS;
        if constexpr (decltype(dilated)::value != 0) {
          uni += k * step;
          asm volatile("" : "+s"(uni));
        }
        const int off = base + uni + (decltype(dilated)::value != 0 ? 0 : k);
        // the activation fragments of JG n-tiles at a time (all of them, or two where four in two planes would be 32
        // registers in flight); each accumulator still takes its tap's products in the order h * l, h * h, l * h
        constexpr int JG = (NJ == 4 && NP == 2) ? 2 : NJ;
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += JG) {
          uint4 Bh[JG], Bl[JG];
#pragma unroll
          for (int j = 0; j < JG; ++j) {
            if constexpr (CHAIN_KO & 8) {
              Bh[j] = make_uint4(off, j, k, chunk);
              Bl[j] = make_uint4(j, off, chunk, k);
            } else {
              Bh[j] = lds[off + (j0 + j) * 32];
              if constexpr (NP == 2) Bl[j] = lds[2 * COLS + off + (j0 + j) * 32];
            }
          }
          if constexpr (NP == 2) {
            mma(Aq[g % D], 0, Bl, j0);                     // h * l
            mma(Aq[g % D], 0, Bh, j0);                     // h * h
            mma(Aq[g % D], NP - 1, Bh, j0);                // l * h
          } else {
            mma(Aq[g % D], 0, Bh, j0);
          }
          if constexpr (JG != NJ) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!(CHAIN_KO & 2)) fetch_a(g + D);
#ifdef CHAIN_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#else
        // the 128-channel f16x2 build has no register to spare for fragments of later taps read ahead of time
        if constexpr (MT == 4 && NP == 2) __builtin_amdgcn_sched_barrier(0);
#endif
      }
  };
  // One accumulator tile (row tile i of the wave, n-tile j) -> leaky-relu, (clamp, split into fp16 h / l | round to
  // bf16), zero outside the sequence -> operand tile
  // A conv's operand is zero outside the sequence (its zero padding): 1 or 0 per column, multiplied in before the
  // leaky-relu (exact for the columns that stay; two packed multiplies and a max per pair of values in all)
  float inside[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) inside[j] = (tl + 32 * j >= 0 && tl + 32 * j < a.L) ? 1.f : 0.f;
  auto emit = [&](int i, int j, float& amax, auto&& v) __attribute__((always_inline)) {
    const f32x2 keep = {inside[j], inside[j]}, slope2 = {a.slope, a.slope};
    const int col = a.H + c0 + 32 * j;
    const int rt = wm * WR + i;
    uint2 oh[4], ol[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x2 e = {v(4 * g + 2 * u), v(4 * g + 2 * u + 1)};
        if constexpr (CHAIN_KO & 4) {
          (u ? oh[g].y : oh[g].x) = __builtin_bit_cast(unsigned, e[0]);
          (u ? ol[g].y : ol[g].x) = __builtin_bit_cast(unsigned, e[1]);
        } else {
          e = e * keep;
          const f32x2 es = e * slope2;
          e[0] = fmaxf(e[0], es[0]);
          e[1] = fmaxf(e[1], es[1]);
          if constexpr (NP == 2) {
            amax = fmaxf(amax, fmaxf(fabsf(e[0]), fabsf(e[1])));
            e[0] = __builtin_amdgcn_fmed3f(e[0], -65504.f, 65504.f);
            e[1] = __builtin_amdgcn_fmed3f(e[1], -65504.f, 65504.f);
          }
        }
        if constexpr (CHAIN_KO & 4) {
        } else if constexpr (NP == 2) {
          const f16x2 h = __builtin_convertvector(e, f16x2);
          const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
          (u ? oh[g].y : oh[g].x) = __builtin_bit_cast(unsigned, h);
          (u ? ol[g].y : ol[g].x) = __builtin_bit_cast(unsigned, l);
        } else {
          (u ? oh[g].y : oh[g].x) = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2));
        }
      }
      if (g & 1) {
        // A 16-byte unit (eight channels of a column) is four channels of lane l and four of lane l + 32.  The two
        // lanes trade one half each -- l gives its part of the odd unit for l + 32's part of the even one -- and
        // write whole units: l the even one, l + 32 the odd one, 512 contiguous bytes per half wave.
        // (the wait states stand in for the hazard checks the compiler does not make inside an asm statement)
        if constexpr (NP == 2)
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                       "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                       : "+v"(oh[g - 1].x), "+v"(oh[g].x), "+v"(oh[g - 1].y), "+v"(oh[g].y),
                         "+v"(ol[g - 1].x), "+v"(ol[g].x), "+v"(ol[g - 1].y), "+v"(ol[g].y));
        else
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                       : "+v"(oh[g - 1].x), "+v"(oh[g].x), "+v"(oh[g - 1].y), "+v"(oh[g].y));
        const int chunk = 2 * rt + (g >> 1);
        lds[((chunk * NP + 0) * 2 + par) * COLS + col] = make_uint4(oh[g - 1].x, oh[g - 1].y, oh[g].x, oh[g].y);
        if constexpr (NP == 2)
          lds[((chunk * NP + 1) * 2 + par) * COLS + col] = make_uint4(ol[g - 1].x, ol[g - 1].y, ol[g].x, ol[g].y);
#ifdef CHAIN_SCHED_BARRIER2
        __builtin_amdgcn_sched_barrier(0);       // eight values at a time: the conversion's temporaries stay few
#else
        if constexpr (MT == 4 && NP == 2) __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
  };

  auto load_bias = [&](const float4* bias, int i, float (&bz)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = bias[8 * (wm * WR + i) + 2 * g + par];      // rows 32 rt + 8 g + 4 par .. + 3
      bz[4 * g] = v.x; bz[4 * g + 1] = v.y; bz[4 * g + 2] = v.z; bz[4 * g + 3] = v.w;
    }
  };
  auto report = [&](float amax) __attribute__((always_inline)) {
    if constexpr (NP == 2) {
      if (a.range_flag && !(amax <= 65504.f)) atomicOr(a.range_flag, 1);
    }
  };
  for (int p = 0; p < a.npair; ++p) {
    const int dil = a.dil[p];
    const float4* bias1 = reinterpret_cast<const float4*>(bias_lds + (2 * p) * 32 * MT);
    const float4* bias2 = reinterpret_cast<const float4*>(bias_lds + (2 * p + 1) * 32 * MT);
    if (p > 0) {
      wa = a.w1[p];
      wb = a.w2[p];
      wc = a.w1[p + 1 < a.npair ? p + 1 : p];               // the last pair's look-ahead lands on valid memory, unused
      __syncthreads();                                      // every wave has read its last fragment of the previous pair
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) emit(i, j, amax, [&](int r) { return xr[i][j][r]; });
    report(amax);
    __syncthreads();
    zero_acc();
    conv(IntC<0>{}, IntC<1>{}, h2 * dil, dil);
    __syncthreads();                                        // every wave has read the x operand
    amax = 0.f;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      float bz[16];
      load_bias(bias1, i, bz);
#pragma unroll
      for (int j = 0; j < NJ; ++j) emit(i, j, amax, [&](int r) { return acc[i][j][r] + bz[r]; });
    }
    report(amax);
    __syncthreads();
    zero_acc();
    conv(IntC<NT>{}, IntC<0>{}, h2, 1);
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      float bz[16];
      load_bias(bias2, i, bz);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = acc[i][j][r] + bz[r];             // c2 + bias, then + x: the order of the pair kernels
          xr[i][j][r] = y + xr[i][j][r];
        }
    }
  }

  // ---- out = carrier (+ out) on the columns this block owns ----
  if constexpr (CHAIN_KO & 16) {
    typedef float fvec __attribute__((ext_vector_type(NJ)));
    float* ob = a.out + (long long)b * a.out_bs;
    const int c = wn * CW + NJ * l31, t = t0 - a.R + c;
    if (c >= a.R && c + NJ <= W1 - a.R && t + NJ <= a.L) {
#pragma unroll
      for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          fvec v;
#pragma unroll
          for (int j = 0; j < NJ; ++j) v[j] = xr[i][j][r];
          fvec* dst = reinterpret_cast<fvec*>(ob + (long long)(32 * (wm * WR + i) + 4 * par + (r & 3) + 8 * (r >> 2)) * a.out_cs + t);
          if (a.accumulate) v += *dst;
          *dst = v;
        }
    }
  } else {
    float* ob = a.out + (long long)b * a.out_bs;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = c0 + 32 * j, t = tl + 32 * j;
      const bool mine = c >= a.R && c < W1 - a.R && t < a.L;
      const int tc = mine ? t : 0;
      if (a.accumulate) {                                   // all loads of the n-tile before its first store
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[i][j][r] = ob[(long long)(32 * (wm * WR + i) + 4 * par + (r & 3) + 8 * (r >> 2)) * a.out_cs + tc];
      }
      if (mine) {
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = xr[i][j][r];
            if (a.accumulate) v += acc[i][j][r];
            ob[(long long)(32 * (wm * WR + i) + 4 * par + (r & 3) + 8 * (r >> 2)) * a.out_cs + t] = v;
          }
      }
    }
  }
}

// Which ResBlocks take the whole-chain kernel.  VITS_CHAIN_F16X2 / VITS_CHAIN_BF16: lists of "<channels>x<taps>"
// (measurement knobs; "none" for the pair-by-pair path everywhere).
bool listed(const char* env, const char* dflt, int C, int K) {
  const char* e = getenv(env);
  const std::string v = std::string(",") + (e ? e : dflt) + ",";
  return v.find("," + std::to_string(C) + "x" + std::to_string(K) + ",") != std::string::npos;
}

}  // namespace

static int chain_radius(const ConvDesc* c1, int npair) {
  int R = 0;
  for (int p = 0; p < npair; ++p) R += (c1[p].K - 1) / 2 * (c1[p].tstep[0] + 1);
  return R;
}

bool chain_fusable(const ConvDesc* c1, const ConvDesc* c2, int npair) {
  if (npair < 1 || npair > CHAIN_MAX_PAIRS) return false;
  for (int p = 0; p < npair; ++p) {
    // 32 or 64 channels; 128 in plain bf16 with three taps (one build: 2 x 2 waves of 64 rows x 64 steps)
    const bool channels = c1[p].M == 32 || c1[p].M == 64 || (c1[p].M == 128 && c1[p].K == 3);
    if (!pair_shape_fusable(c1[p], c2[p]) || !channels) return false;
    if (c1[p].M != c1[0].M || c1[p].K != c1[0].K || c1[p].prec != c1[0].prec) return false;
  }
  return chain_radius(c1, npair) <= 64;                     // a 256-step tile keeps at least half of its columns
}

bool chain_profitable(const ConvDesc* c1, const ConvDesc* c2, int npair) {
  if (!chain_fusable(c1, c2, npair)) return false;
  const int C = c1[0].M, K = c1[0].K;
  return c1[0].prec == PREC_BF16 ? listed("VITS_CHAIN_BF16", "32x3,64x3,32x7,64x7,32x11,128x3", C, K)
                                 : listed("VITS_CHAIN_F16X2", "32x3,64x3,32x7,128x3", C, K);
}

namespace {
template <int MT, int K, int NP, int WR, int NJ, int BPC, int NW>
hipError_t launch_build(ChainArgs& a, const char* name, const ConvArgs& ta, hipStream_t stream) {
  constexpr int W1 = 32 * NJ * (NW / (MT / WR));
  const int BN = W1 - 2 * a.R;
  if (BN < 32) return hipErrorInvalidValue;
  const size_t lds = (size_t)2 * MT * NP * 2 * (W1 + 2 * a.H) * sizeof(uint4) + (size_t)a.npair * 2 * 32 * MT * sizeof(float);
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(&resblock_chain_kernel<MT, K, NP, WR, NJ, BPC, NW>), lds);
      e != hipSuccess)
    return e;
  dim3 grid((a.L + BN - 1) / BN, 1, a.B);
  const ProfTicket tk = prof_begin(stream);
  hipLaunchKernelGGL((resblock_chain_kernel<MT, K, NP, WR, NJ, BPC, NW>), grid, dim3(64 * NW), lds, stream, a);
  const hipError_t err = hipGetLastError();
  prof_end(tk, name, false, ta, stream);
  return err;
}

template <int MT, int NP, int WR, int NJ, int BPC, int NW>
hipError_t launch_taps(int K, ChainArgs& a, const char* name, const ConvArgs& ta, hipStream_t stream) {
  switch (K) {
    case 3: return launch_build<MT, 3, NP, WR, NJ, BPC, NW>(a, name, ta, stream);
    case 5: return launch_build<MT, 5, NP, WR, NJ, BPC, NW>(a, name, ta, stream);
    case 7: return launch_build<MT, 7, NP, WR, NJ, BPC, NW>(a, name, ta, stream);
    case 9: return launch_build<MT, 9, NP, WR, NJ, BPC, NW>(a, name, ta, stream);
    case 11: return launch_build<MT, 11, NP, WR, NJ, BPC, NW>(a, name, ta, stream);
  }
  return hipErrorInvalidValue;
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && e[0] ? atoi(e) : dflt;
}
}  // namespace

// How a ResBlock of npair pairs is cut into launches: seg[i] pairs each, in order.  One segment = the whole ResBlock
// in one launch; segments of one pair go to the pair kernel; 0 = no plan (every pair on its own, by whatever path the
// caller has for it).  Beyond the whole-ResBlock list: 64 channels, 11 taps in plain bf16 -- the whole block loses too
// many columns a side (60 of a 256-step tile) but its first two pairs lose as many as its third alone (30), and fused
// they take 6.4 ms where two pair launches take 7.4 (64 x 10 s).  VITS_CHAIN_SPLIT=0 switches that off.
int chain_plan(const ConvDesc* c1, const ConvDesc* c2, int npair, int* seg) {
  static const bool split = env_int("VITS_CHAIN_SPLIT", 1) != 0;
  // f16x2, 7 taps: the first two pairs in one launch and the third on its own lose 18 columns a side instead of 36
  // (three products a tap make the recomputed columns three times as dear as in bf16).  At 16 x 3 s, 32 channels: 740 us
  // against 753 whole and 783 pair by pair; 64 channels: 1140 against 1228 whole and 1188 as six LDS-DMA conv launches;
  // at 64 x 10 s 9.3 / 10.0 / 10.0 ms and 15.2 / 16.8 / 16.1 ms.  VITS_CHAIN_SPLIT_F16X2: the list ("none": off).
  if (split && npair == 3 && c1[0].prec == PREC_F16X2 && listed("VITS_CHAIN_SPLIT_F16X2", "32x7,64x7", c1[0].M, c1[0].K) &&
      chain_fusable(c1, c2, 2) && chain_fusable(c1 + 2, c2 + 2, 1)) {
    seg[0] = 2;
    seg[1] = 1;
    return 2;
  }
  if (chain_profitable(c1, c2, npair)) {
    seg[0] = npair;
    return 1;
  }
  if (split && npair == 3 && c1[0].prec == PREC_BF16 && c1[0].M == 64 && c1[0].K == 11 && chain_fusable(c1, c2, 2) &&
      pair_fusable(c1[2], c2[2])) {
    seg[0] = 2;
    seg[1] = 1;
    return 2;
  }
  return 0;
}

hipError_t launch_resblock_chain(const ConvDesc* c1, const ConvDesc* c2, int npair, const PairArgs& pa, hipStream_t stream) {
  if (!chain_fusable(c1, c2, npair) || pa.B <= 0 || pa.L <= 0 || pa.B > 65535) return hipErrorInvalidValue;
  const int K = c1[0].K, MT = c1[0].M / 32, NP = c1[0].prec == PREC_BF16 ? 1 : 2;
  ChainArgs a;
  memset(&a, 0, sizeof a);
  a.x = pa.x; a.x_bs = pa.x_bs; a.x_cs = pa.x_cs;
  a.out = pa.out; a.out_bs = pa.out_bs; a.out_cs = pa.out_cs;
  a.B = pa.B; a.L = pa.L; a.accumulate = pa.accumulate; a.slope = pa.slope; a.range_flag = pa.range_flag;
  a.xcd_map = xcd_map_enabled();
  a.npair = npair;
  int dmax = 1;
  for (int p = 0; p < npair; ++p) {
    a.dil[p] = c1[p].tstep[0];
    dmax = std::max(dmax, a.dil[p]);
    a.w1[p] = reinterpret_cast<const uint4*>(c1[p].w); a.b1[p] = c1[p].bias;
    a.w2[p] = reinterpret_cast<const uint4*>(c2[p].w); a.b2[p] = c2[p].bias;
  }
  a.R = chain_radius(c1, npair);
  a.H = (K - 1) / 2 * dmax;
  ConvArgs ta;
  memset(&ta, 0, sizeof ta);
  ta.M = c1[0].M; ta.Cin = c1[0].M; ta.K = 2 * npair * K; ta.L = a.L; ta.B = a.B;    // timing row: every conv's FLOPs
  ta.flags = a.accumulate ? F_ACCUM : 0;                    // bytes: x once, the result once, the sum when accumulated
  char name[64];
  snprintf(name, sizeof name, NP == 1 ? "resblock_chain_bf16_kernel<%d,%d>" : "resblock_chain_kernel<%d,%d>", MT, K);
  // Builds (VITS_CHAIN_BUILD_32 / _64: measurement knobs):
  //   32 channels  0: four waves side by side, 256-step tile (f16x2)    1: the same, 512 steps (bf16: one plane only)
  //                2: one wave a block, 128 steps -- no block barrier at all; measured slower (f16x2 k = 3: 423 against
  //                   351 us at 16 x 3 s; bf16: 2699 against 2435 at 64 x 10 s)
  //   64 channels  bf16: four waves of 64 rows, 256 steps.  f16x2: 1: 2 x 2 waves, 256 steps, two blocks to a CU;
  //                0: 128 steps, three blocks (k = 3: 722 against 651 us)
  static const int wide_from = env_int("VITS_CHAIN_WIDE_R", 0);
  static const int build32 = env_int("VITS_CHAIN_BUILD_32", -1), build64 = env_int("VITS_CHAIN_BUILD_64", 1);
  if (MT == 4)                                                                   // chain_fusable: three taps
  {
    if (NP == 1) return launch_build<4, 3, 1, 2, 2, 2, 4>(a, name, ta, stream);
    return launch_build<4, 3, 2, 2, 2, 2, 4>(a, name, ta, stream);
  }
  if (MT == 1) {
    int build = build32 >= 0 ? build32 : (NP == 1 && a.R >= wide_from) ? 1 : 0;
    if (build == 2 && 128 - 2 * a.R < 64) build = NP == 1 ? 1 : 0;       // a 128-step tile would keep too little
    if (NP == 2) {
      if (build == 2) return launch_taps<1, 2, 1, 4, 2, 1>(K, a, name, ta, stream);
      return launch_taps<1, 2, 1, 2, 3, 4>(K, a, name, ta, stream);
    }
    if (build == 2) return launch_taps<1, 1, 1, 4, 2, 1>(K, a, name, ta, stream);
    if (build == 1) return launch_taps<1, 1, 1, 4, 2, 4>(K, a, name, ta, stream);
    return launch_taps<1, 1, 1, 2, 3, 4>(K, a, name, ta, stream);
  }
  if (NP == 2) {
    if (build64 == 1 || 128 - 2 * a.R < 32) return launch_taps<2, 2, 1, 4, 2, 4>(K, a, name, ta, stream);
    return launch_taps<2, 2, 1, 2, 3, 4>(K, a, name, ta, stream);
  }
  return launch_taps<2, 1, 2, 2, 2, 4>(K, a, name, ta, stream);
}

}  // namespace vits
