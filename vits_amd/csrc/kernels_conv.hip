// MFMA implicit-GEMM convolution for gfx950 (wave64).
//
// One kernel family serves every conv1d / conv_transpose1d on the synthesis path (reference: nn.Conv1d uses in
// attentions.py:196-199,373-374, models.py:116-120,263-283,543,567,657-658, modules.py:129-142,187-202,314-316,
// frame_prior_network.py:39-47).
//
//   GEMM view:  D[row, t] = sum over (ci, tap) of  W[row, ci, tap] * X[ci, t + toff(tap)]
//     M = rows (output channels; for a transposed conv: (channel, phase) pairs), N = time, K = Cin * taps.
//   A operand (weights) comes pre-packed in exact fragment order, so each lane fetches 16 bytes per MFMA group
//   straight from global memory (L2-resident, 1 KiB per wave instruction, no LDS round trip).
//   B operand (activations) is staged per 16 input channels into an LDS tile with the input prologue (scale,
//   leaky-relu, conditioning bias, length mask, zero padding) applied once at staging.
//   The next channel chunk is prefetched into registers while the current chunk is multiplied.
//   C/D: column = time = lane & 31, so every accumulator register stores 128 contiguous bytes per half wave.
//
// Four arithmetic modes share tiling, prologue and epilogue:
//   PREC_F32     v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (64 FLOP/clk/SIMD).
//   PREC_BF16X3  every fp32 operand is split exactly into three bf16 terms (x = h + m + l, 24 significant bits) and
//                the six largest cross products (hh, hm, mh, hl, lh, mm) run on v_mfma_f32_32x32x16_bf16 with fp32
//                accumulation: fp32-grade results (dropped terms < 2^-24 relative) at 16/6 of the fp32 MFMA rate.
//   PREC_F16X2   every fp32 operand is split into two fp16 terms (22 significant bits) and the three largest cross
//                products run on v_mfma_f32_32x32x16_f16: half the matrix work of PREC_BF16X3, relative error per
//                product about 2^-22; operands must lie inside the fp16 range.
//   PREC_BF16    both operands rounded to bf16, one product, fp32 accumulation (8 significant bits per operand).
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace vits {

// ---- optional per-kernel timing (HIP events on the launch stream), off by default ----
namespace {
struct ProfClass {
  double flops = 0.0, ms = 0.0, bytes = 0.0;
  long long count = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
bool g_prof_on = false;
std::mutex g_prof_mu;
std::map<std::string, ProfClass> g_prof;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t prof_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void prof_enable(bool on) { g_prof_on = on; }
bool prof_enabled() { return g_prof_on; }

void prof_reset() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& kv : g_prof) {
    for (auto& p : kv.second.pending) {
      g_event_pool.push_back(p.first);
      g_event_pool.push_back(p.second);
    }
  }
  g_prof.clear();
}

// Resolves pending events (synchronises on them) and returns the number of classes.
int prof_collect(std::vector<ProfRow>& rows) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  rows.clear();
  for (auto& kv : g_prof) {
    ProfClass& c = kv.second;
    for (auto& p : c.pending) {
      float ms = 0.f;
      (void)hipEventSynchronize(p.second);
      (void)hipEventElapsedTime(&ms, p.first, p.second);
      c.ms += ms;
      g_event_pool.push_back(p.first);
      g_event_pool.push_back(p.second);
    }
    c.pending.clear();
    rows.push_back(ProfRow{kv.first, c.flops, c.ms, c.bytes, c.count});
  }
  return (int)rows.size();
}

int device_cu_count();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// f16x2 operands are clamped to the fp16 range at conversion; a caller that wants to know passes a device word, and
// every thread that clamped a value ORs 1 into it (a not-taken branch otherwise).
__device__ __forceinline__ void report_f16_range(int* flag, float amax) {
  if (flag && !(amax <= 65504.f)) atomicOr(flag, 1);
}

// The 16 values a lane holds of one 32 x 32 accumulator tile (m-tile mt, its column at time t) -> the staged f16x2
// format [C/8][2 planes][L] of 16-byte units: leaky-relu(out_slope), clamp, h = fp16(v), l = fp16(v - h).  Register
// group g holds channels 8g + 4*par .. + 3: half a unit, one 8-byte store per plane.
__device__ __forceinline__ void staged_f16x2_store(const ConvArgs& a, const float (&fv)[16], uint2* out_b, int mt, int par,
                                                   int t) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float e = fv[4 * g + q];
      v[q] = fmaxf(e, e * a.out_slope);
    }
    report_f16_range(a.range_flag, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    const long long unit = (long long)(mt * 4 + g) * 2 * a.L + t;
    uint2 oh, ol;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x2 e = {fminf(fmaxf(v[2 * u], -65504.f), 65504.f), fminf(fmaxf(v[2 * u + 1], -65504.f), 65504.f)};
      const f16x2 h = __builtin_convertvector(e, f16x2);
      const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
      (u ? oh.y : oh.x) = __builtin_bit_cast(unsigned, h);
      (u ? ol.y : ol.x) = __builtin_bit_cast(unsigned, l);
    }
    out_b[unit * 2 + par] = oh;
    out_b[(unit + a.L) * 2 + par] = ol;
  }
}

// The same for the staged bf16 format [C/8][L] of 16-byte units (eight bf16): leaky-relu(out_slope), round to nearest.
__device__ __forceinline__ void staged_bf16_store(const ConvArgs& a, const float (&fv)[16], uint2* out_b, int mt, int par,
                                                  int t) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float e = fv[4 * g + q];
      v[q] = fmaxf(e, e * a.out_slope);
    }
    uint2 o;
    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
    out_b[((long long)(mt * 4 + g) * a.L + t) * 2 + par] = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Epilogue shared by both arithmetic modes (the 32x32 C/D layout does not depend on the operand type).
// Addresses are a wave-uniform row base (scalar registers) plus ONE per-lane element offset shared by all rows.
// All loads of a tile are issued before its first store: out may alias res (in-place residual update), so a
// per-element load -> store sequence would serialise 16 memory round trips.
// ---------------------------------------------------------------------------------------------------------
// LN: the deferred-LayerNorm options of ConvArgs (ln_res_*, ln_out_stats) are honoured (the bf16x3 kernels).
// ln_pre: the residual's (mean, 1 / std) per n-tile when the caller has already combined them (else null).
// EXTRA: the optional features a kernel's epilogue is built with -- each costs registers, and the kernels that run three
// blocks to a CU have none to spare (with all of them built in, every such kernel spilled 48-60 bytes):
//   EPI_LN     the deferred-LayerNorm options of ConvArgs (ln_res_*, ln_out_stats): the split-K kernel
//   EPI_SPLIT  a second destination for the rows from split_row on: the kernels that run the flow (fp32, bf16x3)
//   EPI_OUT2   a second, staged copy of the result: the staged-input f16x2 kernels (EPI_OUT2B: in the bf16 format, the
//              LDS-DMA bf16 kernel)
constexpr int EPI_LN = 1, EPI_SPLIT = 2, EPI_OUT2 = 4, EPI_OUT2B = 8;
template <int WM, int WN, int MODE, bool RES_ALL = false, int EXTRA = 0>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[WM][WN], const int (&mt)[WM],
                                              const bool (&mt_ok)[WM], int wn, int par, int l31, int t0, int b,
                                              int len_b, const float (*ln_pre)[2] = nullptr) {
  const float* cbias_b = a.cbias ? a.cbias + (long long)b * a.cbias_bs : nullptr;
  float* out_b = a.out + (long long)b * a.out_bs;
  if constexpr (MODE == MODE_PLAIN) {
    const float* res_b = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    constexpr bool LN = (EXTRA & EPI_LN) != 0;
    int flags = a.flags;
    if constexpr ((EXTRA & EPI_SPLIT) != 0) {
      if (a.split_row > 0 && mt[0] * 32 >= a.split_row) {    // the wave's rows belong to the second destination
        out_b = a.out_hi + (long long)b * a.out_hi_bs - (long long)a.split_row * a.out_cs;
        flags = a.flags_hi;
        res_b = nullptr;
      }
    }
    // Residual loads go out before the first store of the rows they cover (out may alias res element for element,
    // which this order keeps correct).  RES_ALL: the whole wave tile at once (one memory round trip, WM * WN * 16
    // registers); otherwise one 32-row block at a time.
    constexpr int RI = RES_ALL ? WM : 1;
    float rv[RI][WN][16];
    // deferred LayerNorm of the residual: this lane's time steps' mean and 1 / std, once
    float rmean[WN], rrstd[WN];
    bool ln_res = false;
    if constexpr (LN) {
      ln_res = a.ln_res_stats && res_b;
      if (ln_res) {
        const int units = a.M >> 6;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int t = t0 + (wn * WN + j) * 32 + l31;
          if (ln_pre) {
            rmean[j] = ln_pre[j][0];
            rrstd[j] = ln_pre[j][1];
          } else {
            ln_combine(a.ln_res_stats + ((long long)b * a.L + (t < a.L ? t : a.L - 1)) * units * 2, units, rmean[j], rrstd[j]);
          }
        }
      }
    }
    auto load_res = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + (wn * WN + j) * 32 + l31;
        const unsigned lo_res = (unsigned)(4 * par * a.res_cs + (t < a.L ? t : a.L - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[slot][j][r] = (res_b + (long long)(mt[i] * 32 + (r & 3) + 8 * (r >> 2)) * a.res_cs)[lo_res];
        if constexpr (LN) {
          if (ln_res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = mt[i] * 32 + 4 * par + (r & 3) + 8 * (r >> 2);
              rv[slot][j][r] = (rv[slot][j][r] - rmean[j]) * rrstd[j] * a.ln_res_gamma[row] + a.ln_res_beta[row];
            }
          }
        }
      }
    };
    // statistics of the result for a deferred LayerNorm: (sum, M2) of the lane's 16 values per m-tile, merged below
    float psum[WM][WN], pm2[WM][WN];
    if constexpr (LN) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) psum[i][j] = pm2[i][j] = 0.f;
    }
    if constexpr (RES_ALL) {
      if (flags & (F_RES | F_RES_SUB)) {
#pragma unroll
        for (int i = 0; i < WM; ++i) load_res(i, i);
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if (!mt_ok[i]) continue;
      const int rowu = mt[i] * 32;
      const int ri = RES_ALL ? i : 0;
      if constexpr (!RES_ALL) {
        if (flags & (F_RES | F_RES_SUB)) load_res(i, 0);
      }
      float bz[16];                        // whole-array branches: the 16 loads of a table are issued back to back
      const int row0 = rowu + 4 * par;
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] = a.bias[row0 + (r & 3) + 8 * (r >> 2)];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] = 0.f;
      }
      if (cbias_b) {
        float cz[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cz[r] = cbias_b[row0 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] += cz[r];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + (wn * WN + j) * 32 + l31;
        if (t < a.L) {
          const float m = t < len_b ? 1.f : 0.f;
          const unsigned lo_out = (unsigned)(4 * par * a.out_cs + t);
          float ov[16];
          if (flags & F_ACCUM) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ov[r] = (out_b + (long long)(rowu + (r & 3) + 8 * (r >> 2)) * a.out_cs)[lo_out];
          }
          float fv[(EXTRA & (EPI_OUT2 | EPI_OUT2B | EPI_LN)) ? 16 : 1];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] + bz[r];
            if (flags & F_RELU) v = fmaxf(v, 0.f);
            if (flags & F_MASK_PRE) v *= m;
            if (flags & F_RES) v += rv[ri][j][r];
            if (flags & F_RES_SUB) v = rv[ri][j][r] - v;
            if (flags & F_ACCUM) v += ov[r];
            if (flags & F_MASK_POST) v *= m;
            if constexpr ((EXTRA & (EPI_OUT2 | EPI_OUT2B | EPI_LN)) != 0) fv[r] = v;
            (out_b + (long long)(rowu + (r & 3) + 8 * (r >> 2)) * a.out_cs)[lo_out] = v;
          }
          if constexpr ((EXTRA & EPI_OUT2) != 0) {
            if (a.out2)
              staged_f16x2_store(a, fv, reinterpret_cast<uint2*>(a.out2 + (long long)b * a.out2_bs), mt[i], par, t);
          }
          if constexpr ((EXTRA & EPI_OUT2B) != 0) {
            if (a.out2)
              staged_bf16_store(a, fv, reinterpret_cast<uint2*>(a.out2 + (long long)b * a.out2_bs), mt[i], par, t);
          }
          if constexpr (LN) {
            if (a.ln_out_stats) {
              float s16 = 0.f, q16 = 0.f;
#pragma unroll
              for (int r = 0; r < 16; ++r) s16 += fv[r];
              const float mu = s16 * (1.f / 16.f);
#pragma unroll
              for (int r = 0; r < 16; ++r) q16 += (fv[r] - mu) * (fv[r] - mu);
              psum[i][j] = s16;
              pm2[i][j] = q16;
            }
          }
        }
      }
    }
    if constexpr ((EXTRA & EPI_LN) != 0 && WM == 2) {
      if (a.ln_out_stats && mt_ok[0] && mt_ok[1]) {
        // Chan's merge of (sum, M2) pairs: the lane's two m-tiles (16 + 16 values), then lane ^ 32 (32 + 32): the 64
        // rows of the unit at this time step
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int t = t0 + (wn * WN + j) * 32 + l31;
          const float d1 = (psum[0][j] - psum[1][j]) * (1.f / 16.f);
          float sum = psum[0][j] + psum[1][j];
          float m2 = pm2[0][j] + pm2[1][j] + d1 * d1 * 8.f;              // n_a n_b / (n_a + n_b) = 16 * 16 / 32
          const float osum = __shfl_xor(sum, 32), om2 = __shfl_xor(m2, 32);
          const float d2 = (sum - osum) * (1.f / 32.f);
          m2 = m2 + om2 + d2 * d2 * 16.f;                                // 32 * 32 / 64
          sum += osum;
          if (par == 0 && t < a.L) {
            float* so = a.ln_out_stats + (((long long)b * a.L + t) * (a.M >> 6) + (mt[0] >> 1)) * 2;
            so[0] = sum;
            so[1] = m2;
          }
        }
      }
    }
  } else if constexpr (MODE == MODE_GATE) {
    if (mt_ok[0]) {
      const int hrows = a.M >> 1;
      const int rowu = mt[0] * 32;
      float ba[16], bs[16];                // whole-table branches: the loads of a table go out back to back
      const int row0 = rowu + 4 * par;
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ba[r] = a.bias[row0 + (r & 3) + 8 * (r >> 2)];
          bs[r] = a.bias[row0 + (r & 3) + 8 * (r >> 2) + hrows];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ba[r] = bs[r] = 0.f;
      }
      if (cbias_b) {
        float ca[16], cs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ca[r] = cbias_b[row0 + (r & 3) + 8 * (r >> 2)];
          cs[r] = cbias_b[row0 + (r & 3) + 8 * (r >> 2) + hrows];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ba[r] += ca[r];
          bs[r] += cs[r];
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + (wn * WN + j) * 32 + l31;
        if (t >= a.L) continue;
        const unsigned lo_out = (unsigned)(4 * par * a.out_cs + t);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float va = acc[0][j][r] + ba[r], vs = acc[WM - 1][j][r] + bs[r];
          // tanh(a) * sigmoid(s) on the hardware exp: tanh(a) = 1 - 2 / (1 + e^(2a)) (saturates cleanly at +-1),
          // absolute error ~1e-7 against libm
          const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * va));
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-vs));
          (out_b + (long long)(rowu + (r & 3) + 8 * (r >> 2)) * a.out_cs)[lo_out] = th * sg;
        }
      }
    }
  } else if constexpr (MODE == MODE_UP8 || MODE == MODE_UP4) {
    // rows of an m-tile: 8*g + 4*par + jj  ->  channel = cb*8 + 2*g + par, phase = jj (+4 for the second class)
    constexpr int UP = MODE == MODE_UP8 ? 8 : 4;
    constexpr bool OUT2 = (EXTRA & (EPI_OUT2 | EPI_OUT2B)) != 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if (!mt_ok[i]) continue;
      const int cb = MODE == MODE_UP8 ? (mt[i] >> 1) : mt[i];
      const int ph0 = MODE == MODE_UP8 ? 4 * (mt[i] & 1) : 0;
      float bz[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bz[g] = a.bias ? a.bias[mt[i] * 32 + 8 * g + 4 * par] : 0.f;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + (wn * WN + j) * 32 + l31;
        if (t >= a.L) continue;
        const unsigned lo_out = (unsigned)(par * a.out_cs + UP * t + ph0);
        float v[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[g][q] = acc[i][j][4 * g + q] + bz[g];
          *reinterpret_cast<float4*>(out_b + (long long)(cb * 8 + 2 * g) * a.out_cs + lo_out) =
              make_float4(v[g][0], v[g][1], v[g][2], v[g][3]);
        }
        if constexpr (OUT2) {
          // A second copy in the staged operand format of the convs that read this tensor (leaky-relu(out_slope)
          // applied): a 16-byte unit is eight consecutive channels of one time step, and a lane holds four phases of
          // every other channel.  Lanes l and l + 32 (even and odd channels of the same steps) trade two phases each:
          // afterwards lane l has all eight channels of phases 0, 1 and lane l + 32 those of phases 2, 3.
          if (a.out2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              // (the wait states stand in for the hazard checks the compiler does not make inside an asm statement)
              asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v[g][0]), "+v"(v[g][2]));
              asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v[g][1]), "+v"(v[g][3]));
            }
            const long long Lo = (long long)a.L * UP;
            uint4* sb = reinterpret_cast<uint4*>(a.out2 + (long long)b * a.out2_bs);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const long long to = (long long)UP * t + ph0 + 2 * par + u;
              float c8[8];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                c8[2 * g] = fmaxf(v[g][u], v[g][u] * a.out_slope);
                c8[2 * g + 1] = fmaxf(v[g][2 + u], v[g][2 + u] * a.out_slope);
              }
              if constexpr ((EXTRA & EPI_OUT2) != 0) {
                unsigned h4[4], l4[4];
                float amax = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  amax = fmaxf(amax, fmaxf(fabsf(c8[2 * q]), fabsf(c8[2 * q + 1])));
                  f32x2 e = {fminf(fmaxf(c8[2 * q], -65504.f), 65504.f), fminf(fmaxf(c8[2 * q + 1], -65504.f), 65504.f)};
                  const f16x2 h = __builtin_convertvector(e, f16x2);
                  const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
                  h4[q] = __builtin_bit_cast(unsigned, h);
                  l4[q] = __builtin_bit_cast(unsigned, l);
                }
                report_f16_range(a.range_flag, amax);
                sb[(long long)cb * 2 * Lo + to] = make_uint4(h4[0], h4[1], h4[2], h4[3]);
                sb[((long long)cb * 2 + 1) * Lo + to] = make_uint4(l4[0], l4[1], l4[2], l4[3]);
              } else {
                unsigned o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  o4[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{c8[2 * q], c8[2 * q + 1]}, bf16x2));
                sb[(long long)cb * Lo + to] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
              }
            }
          }
        }
      }
    }
  } else {  // MODE_UP2: m-tile = 32 channels of one phase
    if constexpr (WM == 2) {
      // the wave holds both phases of its 32 channels (m-tiles 2 cb, 2 cb + 1): one 8-byte store per output pair, so a
      // row is written in whole runs instead of every other float by each of two instructions
      if (mt_ok[0] && mt_ok[1] && (mt[0] & 1) == 0 && mt[1] == mt[0] + 1) {
        const int cb = mt[0] >> 1;
        float bz0[16], bz1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bz0[r] = a.bias ? a.bias[mt[0] * 32 + (r & 3) + 8 * (r >> 2) + 4 * par] : 0.f;
          bz1[r] = a.bias ? a.bias[mt[1] * 32 + (r & 3) + 8 * (r >> 2) + 4 * par] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int t = t0 + (wn * WN + j) * 32 + l31;
          if (t >= a.L) continue;
          const unsigned lo_out = (unsigned)(4 * par * a.out_cs + 2 * t);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float2 v;
            v.x = acc[0][j][r] + bz0[r];
            v.y = acc[1][j][r] + bz1[r];
            *reinterpret_cast<float2*>(out_b + (long long)(cb * 32 + (r & 3) + 8 * (r >> 2)) * a.out_cs + lo_out) = v;
          }
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if (!mt_ok[i]) continue;
      const int cb = mt[i] >> 1, ph = mt[i] & 1;
      float bz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bz[r] = a.bias ? a.bias[mt[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * par] : 0.f;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int t = t0 + (wn * WN + j) * 32 + l31;
        if (t >= a.L) continue;
        const unsigned lo_out = (unsigned)(4 * par * a.out_cs + 2 * t + ph);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          (out_b + (long long)(cb * 32 + (r & 3) + 8 * (r >> 2)) * a.out_cs)[lo_out] = acc[i][j][r] + bz[r];
      }
    }
  }
}

// m-tiles handled by a wave (clamped so that out-of-range tiles still read valid weights; never stored)
template <int WM, int WGM, int MODE>
__device__ __forceinline__ void wave_mtiles(const ConvArgs& a, int wm, int (&mt)[WM], bool (&mt_ok)[WM]) {
  if constexpr (MODE == MODE_GATE) {
    const int half = a.mt_total >> 1;
    const int p = blockIdx.y * WGM + wm;
    mt_ok[0] = mt_ok[WM - 1] = p < half;
    mt[0] = p < half ? p : half - 1;
    mt[WM - 1] = mt[0] + half;
  } else {
    const int base = (blockIdx.y * WGM + wm) * WM;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      mt_ok[i] = base + i < a.mt_total;
      mt[i] = mt_ok[i] ? base + i : a.mt_total - 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// PREC_F32 kernel.  VEC = true: the input rows are 16-byte aligned and L % 4 == 0, so activations are staged with
// dwordx4 loads and ds_write_b128 (a group of four time steps is then entirely inside or outside [0, L)).
// LDS tile: [16 channels][rs] fp32; MFMA B fragments are conflict-free ds_read_b32 (lanes = consecutive t).
// ---------------------------------------------------------------------------------------------------------
// EX: optional epilogue features (EPI_*), each an instantiation of its own
template <int WM, int WN, int WGM, int WGN, int MODE, bool VEC, int EX = 0>
__global__ __launch_bounds__(256, 3) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int BN = 32 * WN * WGN;
  constexpr int NCOLS = VEC ? (BN + MAX_HALO + 4 + 255) / 256 : (BN + MAX_HALO + 63) / 64;
  constexpr bool TWO_CLASS = (MODE == MODE_UP8 || MODE == MODE_UP2);
  constexpr int NB = TWO_CLASS ? 2 : 1;
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(!TWO_CLASS || WM == 2, "two-class transposed convs pair the classes in one wave");
  static_assert(MODE != MODE_GATE || WM == 2, "gate pairs the tanh and sigmoid rows in one wave");

  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile indices stay scalar
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int rs = a.rs;

  int mt[WM];
  bool mt_ok[WM];
  wave_mtiles<WM, WGM, MODE>(a, wm, mt, mt_ok);

  // ---- activation staging (global -> registers -> LDS), input prologue applied on the way ----
  // LDS column c of the tile holds time step tstart + c.  VEC: tstart is rounded down to a multiple of 4.
  const float* xb = a.x + (long long)b * a.x_bs;
  const int tfirst = t0 + a.min_off;
  const int tstart = VEC ? (tfirst & ~3) : tfirst;
  const int lead = tfirst - tstart;                       // 0 when !VEC
  const int wcols = a.width + lead;                       // columns to stage
  auto prologue = [&](float v, float cb, int t) -> float {
    v = (v + cb) * a.in_scale;
    if (a.flags & F_IN_LRELU) v = fmaxf(v, v * a.in_slope);       // leaky relu, 0 <= slope <= 1
    if (a.flags & F_IN_MASK) v = t < len_b ? v : 0.f;
    return v;
  };
  float st[4][NCOLS][VEC ? 4 : 1];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* xr = xb + (long long)(chunk * CI_CHUNK + wave * 4 + r) * a.x_cs;
#pragma unroll
      for (int q = 0; q < NCOLS; ++q) {
        if constexpr (VEC) {
          const int col = 4 * (lane + 64 * q), t = tstart + col;
          const bool ok = col < wcols && t >= 0 && t < a.L;
          // clamped address + select instead of a branch around the load
          const float4 v = *reinterpret_cast<const float4*>(xr + (ok ? t : 0));
          st[r][q][0] = ok ? v.x : 0.f;
          st[r][q][1] = ok ? v.y : 0.f;
          st[r][q][2] = ok ? v.z : 0.f;
          st[r][q][3] = ok ? v.w : 0.f;
        } else {
          const int col = lane + 64 * q, t = tstart + col;
          const bool ok = col < wcols && t >= 0 && t < a.L;
          const float v = xr[ok ? t : 0];
          st[r][q][0] = ok ? v : 0.f;
        }
      }
    }
  };
  auto store_chunk = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 4 + r;
      const float cb = a.in_cbias ? a.in_cbias[(long long)b * a.in_cbias_bs + chunk * CI_CHUNK + row] : 0.f;
#pragma unroll
      for (int q = 0; q < NCOLS; ++q) {
        if constexpr (VEC) {
          const int col = 4 * (lane + 64 * q), t = tstart + col;
          const bool ok = t >= 0 && t < a.L;              // zero padding stays zero (not prologue(0))
          float4 v;
          v.x = ok ? prologue(st[r][q][0], cb, t) : 0.f;
          v.y = ok ? prologue(st[r][q][1], cb, t + 1) : 0.f;
          v.z = ok ? prologue(st[r][q][2], cb, t + 2) : 0.f;
          v.w = ok ? prologue(st[r][q][3], cb, t + 3) : 0.f;
          if (col < wcols) *reinterpret_cast<float4*>(&xs[row * rs + col]) = v;
        } else {
          const int col = lane + 64 * q, t = tstart + col;
          const bool ok = t >= 0 && t < a.L;
          if (col < wcols) xs[row * rs + col] = ok ? prologue(st[r][q][0], cb, t) : 0.f;
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- weight fragment stream: one float4 per (m-tile, iteration), iteration = (chunk, tap, 8-channel half) ----
  // Two fragment register sets ping-pong (A0 for the even half, A1 for the odd half of each tap), so each load has
  // a full 4*WM*WN-MFMA half-iteration to land and no register copies sit between MFMAs.
  const long long it_stride = (long long)a.mt_total * 64;   // in float4
  const float4* wp[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) wp[i] = reinterpret_cast<const float4*>(a.w) + (long long)mt[i] * 64 + lane;
  const int nchunks = a.Cin / CI_CHUNK;
  const int n_it = nchunks * a.K * 2;
  float4 A0[WM], A1[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) A0[i] = wp[i][0];
  int it = 0;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  auto half_step = [&](const float4 (&af)[WM], const int (&boff)[NB], int sq) {
    float bv[4][NB][WN];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[e][c][j] = xs[(sq * 8 + e * 2) * rs + boff[c] + j * 32];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e][TWO_CLASS ? i : 0][j], acc[i][j], 0, 0, 0);
      }
  };

  const int colbase = wn * WN * 32 + l31 + par * rs - a.min_off + lead;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
    int boff[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) boff[c] = colbase + a.toff0[c];
    for (int k = 0; k < a.K; ++k) {
#pragma unroll
      for (int i = 0; i < WM; ++i) A1[i] = wp[i][(long long)(it + 1) * it_stride];
      __builtin_amdgcn_sched_barrier(0);   // keep the fragment prefetch ahead of the MFMA half-step
      half_step(A0, boff, 0);
      it += 2;
      if (it < n_it) {
#pragma unroll
        for (int i = 0; i < WM; ++i) A0[i] = wp[i][(long long)it * it_stride];
      }
      __builtin_amdgcn_sched_barrier(0);
      half_step(A1, boff, 1);
#pragma unroll
      for (int c = 0; c < NB; ++c) boff[c] += a.tstep[c];
    }
    if (chunk + 1 < nchunks) {
      __syncthreads();
      store_chunk(chunk + 1);
      __syncthreads();
    }
  }

  conv_epilogue<WM, WN, MODE, true, EX>(a, acc, mt, mt_ok, wn, par, l31, t0, b, len_b);
}

// ---------------------------------------------------------------------------------------------------------
// PREC_BF16X3 kernel.
// Weights: packed [Cin/16][tap][M/32][3 planes][64 lanes][8 bf16]; lane l of m-tile mt holds W[mt*32 + (l & 31)]
//   [16*chunk + 8*(l >> 5) + j][tap], j = 0..7: the A fragment of v_mfma_f32_32x32x16_bf16, one 16-byte load per plane.
// Activations: a thread owns a tile column; it loads the chunk's 16 channels of that time step (lanes = consecutive
//   t, coalesced), applies the prologue, splits each value into three bf16 terms and writes them as
//   LDS[plane][half][column][8 bf16] (16-byte units): ds_write_b128 and ds_read_b128 are both lane-contiguous.
//   The B fragment of lane l, n-tile column c is LDS[plane][l >> 5][c] = channels 8*(l >> 5) .. +7 of that column.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16(__bf16 lo, __bf16 hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}

// EX: optional epilogue features (EPI_*), each an instantiation of its own
template <int WM, int WN, int WGM, int WGN, int MODE, int EX = 0>
__global__ __launch_bounds__(256, 3) void conv_bf16x3_kernel(const ConvArgs a) {
  constexpr int BN = 32 * WN * WGN;
  constexpr int NCT = (BN + MAX_HALO + 255) / 256;       // tile columns per thread
  constexpr bool TWO_CLASS = (MODE == MODE_UP8 || MODE == MODE_UP2);
  constexpr int NB = TWO_CLASS ? 2 : 1;
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(!TWO_CLASS || WM == 2, "two-class transposed convs pair the classes in one wave");
  static_assert(MODE != MODE_GATE || WM == 2, "gate pairs the tanh and sigmoid rows in one wave");

  extern __shared__ __attribute__((aligned(16))) uint4 xq[];    // [3 planes][2 halves][ncols] of 8 bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int ncols = a.rs;

  int mt[WM];
  bool mt_ok[WM];
  wave_mtiles<WM, WGM, MODE>(a, wm, mt, mt_ok);

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tstart = t0 + a.min_off;
  const int wcols = a.width;
  float st[NCT][16];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col - lane >= wcols) continue;              // none of this wave's columns is in the tile: no requests
      const bool ok = col < wcols && t >= 0 && t < a.L;
      const float* xc = xb + (long long)chunk * 16 * a.x_cs + (ok ? t : 0);
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) {
        const float v = xc[(long long)ci * a.x_cs];
        st[q][ci] = ok ? v : 0.f;
      }
    }
  };
  auto store_chunk = [&](int chunk) {
    const float* cbp = a.in_cbias ? a.in_cbias + (long long)b * a.in_cbias_bs + chunk * 16 : nullptr;
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col < wcols) {
        bool ok = t >= 0 && t < a.L;
        if (a.flags & F_IN_MASK) ok = ok && t < len_b;
        unsigned ph[8], pm[8], pl[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          __bf16 h[2], m[2], l[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ci = 2 * c2 + u;
            float v = (st[q][ci] + (cbp ? cbp[ci] : 0.f)) * a.in_scale;
            if (a.flags & F_IN_LRELU) v = fmaxf(v, v * a.in_slope);
            v = ok ? v : 0.f;
            h[u] = (__bf16)v;                       // x = h + m + l exactly to 24 significant bits
            const float r1 = v - (float)h[u];
            m[u] = (__bf16)r1;
            const float r2 = r1 - (float)m[u];
            l[u] = (__bf16)r2;
          }
          ph[c2] = pack_bf16(h[0], h[1]);
          pm[c2] = pack_bf16(m[0], m[1]);
          pl[c2] = pack_bf16(l[0], l[1]);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          xq[(0 * 2 + hf) * ncols + col] = make_uint4(ph[4 * hf], ph[4 * hf + 1], ph[4 * hf + 2], ph[4 * hf + 3]);
          xq[(1 * 2 + hf) * ncols + col] = make_uint4(pm[4 * hf], pm[4 * hf + 1], pm[4 * hf + 2], pm[4 * hf + 3]);
          xq[(2 * 2 + hf) * ncols + col] = make_uint4(pl[4 * hf], pl[4 * hf + 1], pl[4 * hf + 2], pl[4 * hf + 3]);
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight fragments: index (((chunk*K + tap) * mt_total + mt) * 3 + plane) * 64 + lane, in 16-byte units
  const uint4* w16 = reinterpret_cast<const uint4*>(a.w);
  const uint4* wp[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) wp[i] = w16 + (long long)mt[i] * 192 + lane;
  const long long it_stride = (long long)a.mt_total * 192;
  const int nchunks = a.Cin / 16;
  const int n_it = nchunks * a.K;
  uint4 A[3][WM];                 // planes h, m, l of the current tap
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < WM; ++i) A[p][i] = wp[i][p * 64];
  int it = 0;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  auto mma = [&](const uint4 (&af)[WM], const uint4 (&bf)[NB][WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                            __builtin_bit_cast(bf16x8, bf[TWO_CLASS ? i : 0][j]),
                                                            acc[i][j], 0, 0, 0);
  };

  const int colbase = wn * WN * 32 + l31 + par * ncols - a.min_off;
  uint4 B[3][NB][WN];             // planes h, m, l of the current tap
  auto ldb = [&](int p, const int (&off)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int j = 0; j < WN; ++j) B[p][c][j] = xq[p * 2 * ncols + off[c] + j * 32];
  };
  // Every operand plane is refetched for the next tap right after its last use in this one (weights from L2,
  // activations from LDS), and the order of the six products puts the h planes' uses first: the refetches have two
  // to five MFMA groups to land, with no second register set.  The reloads are unconditional (a chunk's last tap
  // re-reads its own activation fragments, the last tap of all its own weights), so the code stays straight-line and
  // the compiler waits for exactly the plane it needs.
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
    int boff[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) boff[c] = colbase + a.toff0[c];
    ldb(0, boff);
    ldb(1, boff);
    ldb(2, boff);
    for (int k = 0; k < a.K; ++k) {
      ++it;
      const long long nxt = (long long)(it < n_it ? it : n_it - 1) * it_stride;
      int boffn[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) boffn[c] = k + 1 < a.K ? boff[c] + a.tstep[c] : boff[c];
      mma(A[0], B[2]);                                   // h * l
      ldb(2, boffn);
      mma(A[0], B[1]);                                   // h * m
      mma(A[0], B[0]);                                   // h * h
#pragma unroll
      for (int i = 0; i < WM; ++i) A[0][i] = wp[i][nxt];
      mma(A[1], B[1]);                                   // m * m
      ldb(1, boffn);
      mma(A[1], B[0]);                                   // m * h
#pragma unroll
      for (int i = 0; i < WM; ++i) A[1][i] = wp[i][nxt + 64];
      mma(A[2], B[0]);                                   // l * h
#pragma unroll
      for (int i = 0; i < WM; ++i) A[2][i] = wp[i][nxt + 128];
      ldb(0, boffn);
#pragma unroll
      for (int c = 0; c < NB; ++c) boff[c] = boffn[c];
    }
    if (chunk + 1 < nchunks) {
      __syncthreads();
      store_chunk(chunk + 1);
      __syncthreads();
    }
  }

  conv_epilogue<WM, WN, MODE, true, EX>(a, acc, mt, mt_ok, wn, par, l31, t0, b, len_b);
}

// ---------------------------------------------------------------------------------------------------------
// helpers shared by the bf16x3 kernels below
// ---------------------------------------------------------------------------------------------------------

// (v0, v1) -> packed bf16 pairs h, m, l with v = h + m + l exactly (low half = v0)
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  const f32x2 v = {v0, v1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r1 = {v0 - __builtin_bit_cast(float, h << 16), v1 - __builtin_bit_cast(float, h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 r2 = {r1.x - __builtin_bit_cast(float, m << 16), r1.y - __builtin_bit_cast(float, m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

template <int V>
struct IntC { static constexpr int value = V; };

// Epilogue for F_OUT_STAGED (MODE_PLAIN): result + bias (+ conditioning bias) -> leaky-relu(out_slope) -> the consumer's
// operand format.  A lane holds, per 8-row group g of an m-tile, the four channels 8g + 4*par .. + 3 of its time step:
// half of a 16-byte unit, so one 8-byte store per group (and per plane).  PLANES = 1: bf16; PLANES = 2: fp16 h / l split.
template <int WM, int WN, int PLANES>
__device__ __forceinline__ void conv_epilogue_staged(const ConvArgs& a, f32x16 (&acc)[WM][WN], const int (&mt)[WM],
                                                     const bool (&mt_ok)[WM], int wn, int par, int l31, int t0, int b) {
  const float* cbias_b = a.cbias ? a.cbias + (long long)b * a.cbias_bs : nullptr;
  uint2* out_b = reinterpret_cast<uint2*>(a.out + (long long)b * a.out_bs);      // 8-byte units
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    if (!mt_ok[i]) continue;
    const int row0 = mt[i] * 32 + 4 * par;
    float bz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bz[r] = a.bias ? a.bias[row0 + (r & 3) + 8 * (r >> 2)] : 0.f;
    if (cbias_b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bz[r] += cbias_b[row0 + (r & 3) + 8 * (r >> 2)];
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int t = t0 + (wn * WN + j) * 32 + l31;
      if (t >= a.L) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float e = acc[i][j][4 * g + q] + bz[4 * g + q];
          e = fmaxf(e, e * a.out_slope);
          v[q] = e;
        }
        // 16-byte unit of (channel group cg, plane p, time t): ((cg * PLANES + p) * L + t)
        const long long unit = (long long)(mt[i] * 4 + g) * PLANES * a.L + t;
        if constexpr (PLANES == 1) {
          uint2 o;
          o.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          o.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          out_b[unit * 2 + par] = o;
        } else {
          uint2 oh, ol;
          report_f16_range(a.range_flag, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x2 e = {fminf(fmaxf(v[2 * u], -65504.f), 65504.f), fminf(fmaxf(v[2 * u + 1], -65504.f), 65504.f)};
            const f16x2 h = __builtin_convertvector(e, f16x2);
            const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
            (u ? oh.y : oh.x) = __builtin_bit_cast(unsigned, h);
            (u ? ol.y : ol.x) = __builtin_bit_cast(unsigned, l);
          }
          out_b[unit * 2 + par] = oh;
          out_b[(unit + a.L) * 2 + par] = ol;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// PREC_BF16 kernel: one bf16 product per multiply-accumulate.  Both operands are rounded to nearest-even bf16 (the
// weights on the host at packing time, the activations at staging after the input prologue) and accumulated in fp32
// on v_mfma_f32_32x32x16_bf16.  Same tiling, staging order and epilogue as the bf16x3 kernel with one plane instead
// of three: weights [Cin/16][tap][M/32][64 lanes][8 bf16], LDS [2 halves][ncols] of 8 bf16.  A sixth of the
// matrix work for the same activation traffic, so the long layers of the generator run at the rate HBM delivers
// their fp32 inputs, residuals and outputs.
// ---------------------------------------------------------------------------------------------------------
template <int WM, int WN, int WGM, int WGN, int MODE, bool STG = false>
__global__ __launch_bounds__(256, WM * WN > 4 ? 2 : 3) void conv_bf16_kernel(const ConvArgs a) {
  constexpr int BN = 32 * WN * WGN;
  constexpr int NCT = (BN + MAX_HALO + 255) / 256;       // tile columns per thread
  constexpr bool TWO_CLASS = (MODE == MODE_UP8 || MODE == MODE_UP2);
  constexpr int NB = TWO_CLASS ? 2 : 1;
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(!TWO_CLASS || WM == 2, "two-class transposed convs pair the classes in one wave");
  static_assert(MODE != MODE_GATE || WM == 2, "gate pairs the tanh and sigmoid rows in one wave");
  static_assert(!STG || MODE == MODE_PLAIN, "staged activations are a plain-conv format");
  constexpr int NPU = 1;                                  // 16-byte units per (channel group, time step)

  extern __shared__ __attribute__((aligned(16))) uint4 xq[];    // [2 halves][ncols] of 8 bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int ncols = a.rs;

  int mt[WM];
  bool mt_ok[WM];
  wave_mtiles<WM, WGM, MODE>(a, wm, mt, mt_ok);

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tstart = t0 + a.min_off;
  const int wcols = a.width;
  float st[STG ? 1 : NCT][16];
  uint4 sp[STG ? NCT : 1][2 * NPU];                         // staged input: the chunk's units of a column, as stored
  const uint4* xs = reinterpret_cast<const uint4*>(xb);
  auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (STG) {
#pragma unroll
      for (int q = 0; q < NCT; ++q) {
        const int col = tid + 256 * q, t = tstart + col;
        if (col - lane >= wcols) continue;              // none of this wave's columns is in the tile: no requests
        const bool ok = col < wcols && t >= 0 && t < a.L;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int p = 0; p < NPU; ++p)
            sp[q][hf * NPU + p] = xs[((long long)(chunk * 2 + hf) * NPU + p) * a.L + (ok ? t : 0)];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col - lane >= wcols) continue;              // none of this wave's columns is in the tile: no requests
      const bool ok = col < wcols && t >= 0 && t < a.L;
      const float* xc = xb + (long long)chunk * 16 * a.x_cs + (ok ? t : 0);
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) {
        const float v = xc[(long long)ci * a.x_cs];
        st[q][ci] = ok ? v : 0.f;
      }
    }
  };
  auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (STG) {
#pragma unroll
      for (int q = 0; q < NCT; ++q) {
        const int col = tid + 256 * q, t = tstart + col;
        if (col < wcols) {
          const bool ok = t >= 0 && t < a.L;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int p = 0; p < NPU; ++p)
              xq[(p * 2 + hf) * ncols + col] = ok ? sp[q][hf * NPU + p] : make_uint4(0, 0, 0, 0);
        }
      }
      return;
    }
    const float* cbp = a.in_cbias ? a.in_cbias + (long long)b * a.in_cbias_bs + chunk * 16 : nullptr;
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col < wcols) {
        bool ok = t >= 0 && t < a.L;
        if (a.flags & F_IN_MASK) ok = ok && t < len_b;
        unsigned ph[8];
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          f32x2 v;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ci = 2 * c2 + u;
            float e = (st[q][ci] + (cbp ? cbp[ci] : 0.f)) * a.in_scale;
            if (a.flags & F_IN_LRELU) e = fmaxf(e, e * a.in_slope);
            v[u] = ok ? e : 0.f;
          }
          ph[c2] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        }
        xq[col] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        xq[ncols + col] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight fragments: index ((chunk*K + tap) * mt_total + mt) * 64 + lane, in 16-byte units
  const uint4* w16 = reinterpret_cast<const uint4*>(a.w);
  const uint4* wp[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) wp[i] = w16 + (long long)mt[i] * 64 + lane;
  const long long it_stride = (long long)a.mt_total * 64;
  const int nchunks = a.Cin / 16;
  const int n_it = nchunks * a.K;
  uint4 A[WM], An[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) A[i] = wp[i][0];
  int it = 0;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int colbase = wn * WN * 32 + l31 + par * ncols - a.min_off;
  uint4 B[NB][WN], Bn[NB][WN];
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
    int boff[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      boff[c] = colbase + a.toff0[c];
#pragma unroll
      for (int j = 0; j < WN; ++j) B[c][j] = xq[boff[c] + j * 32];
    }
    for (int k = 0; k < a.K; ++k) {
      // operands of the next tap are requested before this tap's products (a chunk's last tap re-reads its own
      // activation fragments, the last tap of all its own weights: straight-line code, exact waits)
      ++it;
      const long long nxt = (long long)(it < n_it ? it : n_it - 1) * it_stride;
#pragma unroll
      for (int i = 0; i < WM; ++i) An[i] = wp[i][nxt];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (k + 1 < a.K) boff[c] += a.tstep[c];
#pragma unroll
        for (int j = 0; j < WN; ++j) Bn[c][j] = xq[boff[c] + j * 32];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]),
                                                              __builtin_bit_cast(bf16x8, B[TWO_CLASS ? i : 0][j]),
                                                              acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < WM; ++i) A[i] = An[i];
#pragma unroll
      for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int j = 0; j < WN; ++j) B[c][j] = Bn[c][j];
    }
    if (chunk + 1 < nchunks) {
      __syncthreads();
      store_chunk(chunk + 1);
      __syncthreads();
    }
  }

  if constexpr (MODE == MODE_PLAIN) {
    if (a.flags & F_OUT_STAGED) {
      conv_epilogue_staged<WM, WN, NPU>(a, acc, mt, mt_ok, wn, par, l31, t0, b);
      return;
    }
  }
  conv_epilogue<WM, WN, MODE, true, (MODE == MODE_UP8 || MODE == MODE_UP4) ? EPI_OUT2B : 0>(a, acc, mt, mt_ok, wn, par, l31, t0, b, len_b);
}

// ---------------------------------------------------------------------------------------------------------
// PREC_F16X2 kernel: every fp32 operand is split into two fp16 terms, x = h + l with h = fp16(x) and
// l = fp16(x - h) (22 significant bits; fp16 subnormals are produced by the conversions and honoured by the MFMA, so
// the absolute representation error is at most 2^-25 for |x| < 1/2), and the three largest cross products
// (h*l, h*h, l*h) run on v_mfma_f32_32x32x16_f16 with fp32 accumulation: half the matrix work of the bf16x3 split for
// a relative error per product of about 2^-22 instead of 2^-24.  Activations are clamped to the fp16 range
// (+-65504) at staging.  Weights: [Cin/16][tap][M/32][2 planes][64 lanes][8 fp16]; LDS [2 planes][2 halves][ncols].
// ---------------------------------------------------------------------------------------------------------
// RING > 0: the variant for launches that cannot fill the chip (at most two blocks per CU).  There a block is one
// serial chain whose every tap waits for its weight fragments, and with few blocks per XCD those come from beyond L2
// (~1800 cycles a tap measured at batch 1, against 384 cycles of MFMAs).  The variant keeps the fragments of the next
// RING taps in flight in a register ring (slots are compile-time: the tap loop is flat and unrolled RING times) and
// spends the registers that occupancy does not need.  Same operands in the same order: bit-identical results.
// EX: optional epilogue features beyond the staged copy (EPI_SPLIT: WaveNet's res / skip conv with two destinations)
template <int WM, int WN, int WGM, int WGN, int MODE, bool STG = false, int RING = 0, int EX = 0>
__global__ __launch_bounds__(256, RING ? 2 : WM * WN > 4 ? 2 : 3) void conv_f16x2_kernel(const ConvArgs a) {
  constexpr int BN = 32 * WN * WGN;
  constexpr int NCT = (BN + MAX_HALO + 255) / 256;       // tile columns per thread
  constexpr bool TWO_CLASS = (MODE == MODE_UP8 || MODE == MODE_UP2);
  constexpr int NB = TWO_CLASS ? 2 : 1;
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(!TWO_CLASS || WM == 2, "two-class transposed convs pair the classes in one wave");
  static_assert(MODE != MODE_GATE || WM == 2, "gate pairs the tanh and sigmoid rows in one wave");
  static_assert(!STG || MODE == MODE_PLAIN, "staged activations are a plain-conv format");
  constexpr int NPU = 2;                                  // 16-byte units per (channel group, time step)

  extern __shared__ __attribute__((aligned(16))) uint4 xq[];    // [2 planes][2 halves][ncols] of 8 fp16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int ncols = a.rs;

  int mt[WM];
  bool mt_ok[WM];
  wave_mtiles<WM, WGM, MODE>(a, wm, mt, mt_ok);

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tstart = t0 + a.min_off;
  const int wcols = a.width;
  float st[STG ? 1 : NCT][16];
  uint4 sp[STG ? NCT : 1][2 * NPU];                         // staged input: the chunk's units of a column, as stored
  const uint4* xs = reinterpret_cast<const uint4*>(xb);
  auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (STG) {
#pragma unroll
      for (int q = 0; q < NCT; ++q) {
        const int col = tid + 256 * q, t = tstart + col;
        if (col - lane >= wcols) continue;              // none of this wave's columns is in the tile: no requests
        const bool ok = col < wcols && t >= 0 && t < a.L;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int p = 0; p < NPU; ++p)
            sp[q][hf * NPU + p] = xs[((long long)(chunk * 2 + hf) * NPU + p) * a.L + (ok ? t : 0)];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col - lane >= wcols) continue;              // none of this wave's columns is in the tile: no requests
      const bool ok = col < wcols && t >= 0 && t < a.L;
      const float* xc = xb + (long long)chunk * 16 * a.x_cs + (ok ? t : 0);
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) {
        const float v = xc[(long long)ci * a.x_cs];
        st[q][ci] = ok ? v : 0.f;
      }
    }
  };
  auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (STG) {
#pragma unroll
      for (int q = 0; q < NCT; ++q) {
        const int col = tid + 256 * q, t = tstart + col;
        if (col < wcols) {
          const bool ok = t >= 0 && t < a.L;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int p = 0; p < NPU; ++p)
              xq[(p * 2 + hf) * ncols + col] = ok ? sp[q][hf * NPU + p] : make_uint4(0, 0, 0, 0);
        }
      }
      return;
    }
    const float* cbp = a.in_cbias ? a.in_cbias + (long long)b * a.in_cbias_bs + chunk * 16 : nullptr;
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = tid + 256 * q, t = tstart + col;
      if (col < wcols) {
        bool ok = t >= 0 && t < a.L;
        if (a.flags & F_IN_MASK) ok = ok && t < len_b;
        unsigned ph[8], pl[8];
        float amax = 0.f;                               // largest magnitude of the column's 16 values, before the clamp
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          f32x2 v;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ci = 2 * c2 + u;
            float e = (st[q][ci] + (cbp ? cbp[ci] : 0.f)) * a.in_scale;
            if (a.flags & F_IN_LRELU) e = fmaxf(e, e * a.in_slope);
            amax = fmaxf(amax, fabsf(e));
            e = fminf(fmaxf(e, -65504.f), 65504.f);
            v[u] = ok ? e : 0.f;
          }
          const f16x2 h = __builtin_convertvector(v, f16x2);
          const f32x2 r = v - __builtin_convertvector(h, f32x2);
          const f16x2 l = __builtin_convertvector(r, f16x2);
          ph[c2] = __builtin_bit_cast(unsigned, h);
          pl[c2] = __builtin_bit_cast(unsigned, l);
        }
        report_f16_range(a.range_flag, amax);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          xq[(0 * 2 + hf) * ncols + col] = make_uint4(ph[4 * hf], ph[4 * hf + 1], ph[4 * hf + 2], ph[4 * hf + 3]);
          xq[(1 * 2 + hf) * ncols + col] = make_uint4(pl[4 * hf], pl[4 * hf + 1], pl[4 * hf + 2], pl[4 * hf + 3]);
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight fragments: index (((chunk*K + tap) * mt_total + mt) * 2 + plane) * 64 + lane, in 16-byte units
  const uint4* w16 = reinterpret_cast<const uint4*>(a.w);
  const uint4* wp[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) wp[i] = w16 + (long long)mt[i] * 128 + lane;
  const long long it_stride = (long long)a.mt_total * 128;
  const int nchunks = a.Cin / 16;
  const int n_it = nchunks * a.K;
  constexpr int D = RING > 0 ? RING : 1;
  uint4 A[2][WM], An[2][WM];      // planes h, l of the current and of the next tap
  uint4 Aq[D][2][WM];             // RING: planes h, l of the next D taps, tap i in slot i % D
  auto fetch_a = [&](int itx, uint4 (&dst)[2][WM]) __attribute__((always_inline)) {
    const long long o = (long long)(itx < n_it ? itx : n_it - 1) * it_stride;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < WM; ++i) dst[p][i] = wp[i][o + p * 64];
  };
  if constexpr (RING > 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) fetch_a(d, Aq[d]);
  } else {
    fetch_a(0, A);
  }
  int it = 0;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  auto mma = [&](const uint4 (&af)[WM], const uint4 (&bf)[NB][WN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]),
                                                           __builtin_bit_cast(f16x8, bf[TWO_CLASS ? i : 0][j]),
                                                           acc[i][j], 0, 0, 0);
  };

  const int colbase = wn * WN * 32 + l31 + par * ncols - a.min_off;
  uint4 B[2][NB][WN];             // planes h, l of the current tap
  auto ldb = [&](int p, const int (&off)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
      for (int j = 0; j < WN; ++j) B[p][c][j] = xq[p * 2 * ncols + off[c] + j * 32];
  };
  // The next tap's weight fragments (L2) are requested at the top of a tap into a second register set; each
  // activation plane (LDS) is refetched right after its last use.  All reloads are unconditional (a chunk's last tap
  // re-reads its own activation fragments, the last tap of all its own weights): straight-line code, exact waits.
  if constexpr (RING > 0) {
    int chunk = 0, k = 0;
    int boff[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) boff[c] = colbase + a.toff0[c];
    if (nchunks > 1) load_chunk(1);
    ldb(0, boff);
    ldb(1, boff);
    for (int it0 = 0; it0 < n_it; it0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (it0 + d < n_it) {
          int boffn[NB];
#pragma unroll
          for (int c = 0; c < NB; ++c) boffn[c] = k + 1 < a.K ? boff[c] + a.tstep[c] : boff[c];
          mma(Aq[d][0], B[1]);                             // h * l
          ldb(1, boffn);
          mma(Aq[d][0], B[0]);                             // h * h
          mma(Aq[d][1], B[0]);                             // l * h
          ldb(0, boffn);
          fetch_a(it0 + d + D, Aq[d]);
#pragma unroll
          for (int c = 0; c < NB; ++c) boff[c] = boffn[c];
          if (++k == a.K) {
            k = 0;
            ++chunk;
            if (chunk < nchunks) {
              __syncthreads();
              store_chunk(chunk);
              __syncthreads();
              if (chunk + 1 < nchunks) load_chunk(chunk + 1);
#pragma unroll
              for (int c = 0; c < NB; ++c) boff[c] = colbase + a.toff0[c];
              ldb(0, boff);
              ldb(1, boff);
            }
          }
        }
      }
    }
  } else
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk + 1 < nchunks) load_chunk(chunk + 1);
    int boff[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) boff[c] = colbase + a.toff0[c];
    ldb(0, boff);
    ldb(1, boff);
    for (int k = 0; k < a.K; ++k) {
      ++it;
      const long long nxt = (long long)(it < n_it ? it : n_it - 1) * it_stride;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < WM; ++i) An[p][i] = wp[i][nxt + p * 64];
      int boffn[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) boffn[c] = k + 1 < a.K ? boff[c] + a.tstep[c] : boff[c];
      mma(A[0], B[1]);                                   // h * l
      ldb(1, boffn);
      mma(A[0], B[0]);                                   // h * h
      mma(A[1], B[0]);                                   // l * h
      ldb(0, boffn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < WM; ++i) A[p][i] = An[p][i];
#pragma unroll
      for (int c = 0; c < NB; ++c) boff[c] = boffn[c];
    }
    if (chunk + 1 < nchunks) {
      __syncthreads();
      store_chunk(chunk + 1);
      __syncthreads();
    }
  }

  if constexpr (MODE == MODE_PLAIN) {
    if (a.flags & F_OUT_STAGED) {
      conv_epilogue_staged<WM, WN, NPU>(a, acc, mt, mt_ok, wn, par, l31, t0, b);
      return;
    }
  }
  conv_epilogue<WM, WN, MODE, true, (STG || MODE == MODE_UP8 || MODE == MODE_UP4) ? EPI_OUT2 : EX>(a, acc, mt, mt_ok, wn, par, l31, t0, b, len_b);
}

// ---------------------------------------------------------------------------------------------------------
// PREC_F16X2 with a staged input, every operand through LDS by LDS-DMA (conv_gl_kernel; MODE_PLAIN, K taps known at
// compile time).
//
// A CU's vector-memory results come back in issue order, so a weight fragment requested one tap ahead (an L2 hit)
// returns behind whatever activation tile some wave of the CU is streaming from memory, and the matrix pipe idles for
// that round trip on every tap.  Here no value the tap loop needs is requested less than D taps ahead, and none of
// it passes through registers on its way in:
//   * weights of one (chunk, tap) step for the block's BM rows are one contiguous run of the packed stream
//     ([M/32][2 planes][64 lanes] 16-byte units): BM * 64 bytes, copied by `buffer_load_dwordx4 ... lds` into a ring of
//     S = D + 1 slots, D steps ahead of their use;
//   * the staged activations of a 16-channel chunk ([2 halves][2 planes] runs over time, see F_OUT_STAGED) are copied
//     the same way, A chunks ahead, into one of A + 1 buffers; lanes outside [0, L) lie outside the buffer
//     descriptor's range and deliver zeros, which is the convolution's padding;
//   * a step is: counted s_waitcnt vmcnt (everything older than the last D - 1 steps has landed) -> s_barrier (every
//     wave's share has) -> fragment reads (ds_read_b128, conflict-free) -> the DMA requests of step + D into the slot
//     the barrier has just released -> 12 MFMAs.
// The DMA pattern is the same at every step (indices are clamped at the end, the first D steps' requests are issued
// before the loop in the order the loop would have issued them), so the counts are compile-time constants.
// Same operands in the same order as conv_f16x2_kernel: bit-identical results.
// Block = 4 waves (WGM x WGN), wave tile 64 x 64, two blocks to a CU.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned gl_u32x4 __attribute__((ext_vector_type(4)));

constexpr int gl_floor_mod(int x, int m) { return ((x % m) + m) % m; }
constexpr int gl_floor_div(int x, int m) { return (x - gl_floor_mod(x, m)) / m; }
// activation tiles requested during the D - 1 steps before tap k (requests go out at tap 0 of a chunk)
constexpr int gl_tap0_in_window(int k, int K, int D) {
  int n = 0;
  for (int j = 1; j <= D - 1; ++j) n += gl_floor_mod(k - j, K) == 0 ? 1 : 0;
  return n;
}

template <int I, int N, typename F>
__device__ __forceinline__ void gl_static_for(F& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    gl_static_for<I + 1, N>(f);
  }
}

// one step's operand fragments: weights (m-tile 0/1, planes h/l) and activations (n-tile 0..3, planes h/l)
struct GlFrags {
  gl_u32x4 ah0, al0, ah1, al1, bh0, bh1, bl0, bl1, bh2, bh3, bl2, bl3;
};
__device__ __forceinline__ void gl_zero_frags(GlFrags& f) {
  const gl_u32x4 z = {0u, 0u, 0u, 0u};
  f.ah0 = f.al0 = f.ah1 = f.al1 = f.bh0 = f.bh1 = f.bl0 = f.bl1 = f.bh2 = f.bh3 = f.bl2 = f.bl3 = z;
}

template <int WN, int WGM, int WGN, int K, int D, int A>
__global__ __launch_bounds__(256, 2) void conv_gl_kernel(const ConvArgs a) {
  constexpr int WM = 2;
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  static_assert(WN == 2 || WN == 4, "wave tile 64 x 64 or 64 x 128");
  constexpr int S = D + 1, NBUF = A + 1;
  constexpr int NCOLS = BN + MAX_HALO;           // tile columns: a multiple of 64, halo included
  constexpr int WU = BM * 4;                     // 16-byte units of one step's weights
  constexpr int WPK = WU / 64 / 4;               // weight DMA instructions per wave and step
  constexpr int APC = NCOLS / 64;                // activation DMA instructions per wave and chunk (wave w copies slab w)
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(A * K >= D, "an activation tile must have left the request window when it is first read");
  static_assert(WU % 256 == 0 && NCOLS % 64 == 0, "whole wave-instructions");

  extern __shared__ __attribute__((aligned(16))) uint4 gl_lds[];
  uint4* wq = gl_lds;                            // [S][BM/32][2 planes][64 lanes]
  uint4* xq = gl_lds + S * WU;                   // [NBUF][2 halves][2 planes][NCOLS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  int tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map), rowblk = blockIdx.y;
  if (gridDim.y == 2) xcd_tile_rows2(blockIdx.x, blockIdx.y, gridDim.x, a.xcd_map, tile, rowblk);
  const int t0 = tile * BN;
  const int mt0 = rowblk * (BM / 32);
  int mt[WM];
  bool mt_ok[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    mt[i] = mt0 + wm * WM + i;
    mt_ok[i] = true;                             // the launcher only takes M % BM == 0
  }
  const int nchunks = a.Cin / 16;
  const int total = nchunks * K;
  const int tstart = t0 + a.min_off;
  const int dil = a.tstep[0];

  // ---- DMA requests ----
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.Cin * K * a.M * 4, 0x00020000);
  const uint4* xs_b = reinterpret_cast<const uint4*>(a.x + (long long)b * a.x_bs);
  auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {
    const int sc = s < total ? s : total - 1;
    const int unit0 = (sc * a.mt_total + mt0) * 128;
#pragma unroll
    for (int j = 0; j < WPK; ++j) {
      const int piece = wave + 4 * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wq + slot * WU + piece * 64), 16,
                                               (unit0 + piece * 64 + lane) * 16, 0, 0, 0);
    }
  };
  auto issue_x = [&](int c, int buf) __attribute__((always_inline)) {
    const int cc = c < nchunks ? c : nchunks - 1;
    // this wave's slab: channel group 2 * cc + (wave >> 1), plane wave & 1; its run over time is one buffer, so that
    // t < 0 and t >= L are out of range and read as zero
    const uint4* slab = xs_b + (long long)((cc * 2 + (wave >> 1)) * 2 + (wave & 1)) * a.L;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(slab), 0, a.L * 16, 0x00020000);
#pragma unroll
    for (int j = 0; j < APC; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(xq + (buf * 4 + wave) * NCOLS + 64 * j), 16,
                                               (tstart + 64 * j + lane) * 16, 0, 0, 0);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- the requests of the steps before the first one, in the loop's own order ----
#pragma unroll
  for (int c = 0; c < A; ++c)
    if ((c - A) * K < -D) issue_x(c, c % NBUF);
#pragma unroll
  for (int v = -D; v < 0; ++v) {
    if (gl_floor_mod(v, K) == 0) issue_x(gl_floor_div(v, K) + A, (gl_floor_div(v, K) + A) % NBUF);
    issue_w(v + D, (v + D) % S);
  }

  auto mma = [&](const gl_u32x4 (&af)[WM], const gl_u32x4 (&bf)[WN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]),
                                                           __builtin_bit_cast(f16x8, bf[j]), acc[i][j], 0, 0, 0);
      }
  };

  int slot_rd = 0, slot_wr = D % S;              // ring slot of the current step, and of step + D
  int buf_rd = 0, buf_wr = A % NBUF;             // buffer of the current chunk, and of chunk + A
  // Fragment reads are written as ds_read_b128 statements of our own: a read the compiler can see is made to wait for
  // EVERY LDS-DMA request in flight (it cannot tell the ring slots apart), which would drain the look-ahead at every
  // step.  Byte addresses: fragment (m-tile i, plane p) of a slot at a_addr + (2 * i + p) * 1024; activation unit
  // (half par, plane p, column) at b_addr + p * NCOLS * 16 + 512 * j, tap k a further k * dil * 16.
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void*)gl_lds;
  const unsigned a_addr = lds_base + (unsigned)(((wm * WM * 2) * 64 + lane) * 16);
  const unsigned b_addr = lds_base + (unsigned)((S * WU + par * 2 * NCOLS + wn * 32 * WN + l31) * 16);
  // The fragments of step u are read while the MFMAs of step u - 1 run: two register sets, alternating.  A step is
  //   wait (this wave's DMA share of step u has landed; the reads of step u - 1 have) -> s_barrier (every wave's)
  //   -> reads of step u -> DMA requests of step u + D, into the slot step u - 1 has just left -> MFMAs of step u - 1.
  GlFrags f0, f1;
  gl_zero_frags(f1);
  auto read_frags = [&](GlFrags& f, unsigned wa, unsigned xk) __attribute__((always_inline)) {
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072"
        : "=&v"(f.ah0), "=&v"(f.al0), "=&v"(f.ah1), "=&v"(f.al1)
        : "v"(wa)
        : "memory");
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:512\n\t"
        "ds_read_b128 %2, %4 offset:%c5\n\t"
        "ds_read_b128 %3, %4 offset:%c6"
        : "=&v"(f.bh0), "=&v"(f.bh1), "=&v"(f.bl0), "=&v"(f.bl1)
        : "v"(xk), "i"(NCOLS * 16), "i"(NCOLS * 16 + 512)
        : "memory");
    if constexpr (WN == 4)
      asm volatile(
          "ds_read_b128 %0, %4 offset:1024\n\t"
          "ds_read_b128 %1, %4 offset:1536\n\t"
          "ds_read_b128 %2, %4 offset:%c5\n\t"
          "ds_read_b128 %3, %4 offset:%c6"
          : "=&v"(f.bh2), "=&v"(f.bh3), "=&v"(f.bl2), "=&v"(f.bl3)
          : "v"(xk), "i"(NCOLS * 16 + 1024), "i"(NCOLS * 16 + 1536)
          : "memory");
  };
  // the three products of a step, from a register set whose reads have been waited for
  auto mma3 = [&](const GlFrags& f) __attribute__((always_inline)) {
    const gl_u32x4 Ah[WM] = {f.ah0, f.ah1}, Al[WM] = {f.al0, f.al1};
    gl_u32x4 Bh[WN], Bl[WN];
    Bh[0] = f.bh0; Bh[1] = f.bh1; Bl[0] = f.bl0; Bl[1] = f.bl1;
    if constexpr (WN == 4) { Bh[2] = f.bh2; Bh[3] = f.bh3; Bl[2] = f.bl2; Bl[3] = f.bl3; }
    mma(Ah, Bl);                                       // h * l
    mma(Ah, Bh);                                       // h * h
    mma(Al, Bh);                                       // l * h
  };
  int s = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 2) {        // two chunks a trip: the register sets swap at compile time
    auto step = [&](auto jj) __attribute__((always_inline)) {
      constexpr int j = decltype(jj)::value;
      constexpr int k = j % K;
      constexpr int allowed = (D - 1) * WPK + APC * gl_tap0_in_window(k, K, D);
      GlFrags& cur = (j & 1) ? f1 : f0;
      GlFrags& prev = (j & 1) ? f0 : f1;
      // the wait names every register of the previous set, so nothing touches them before their reads have landed
      if constexpr (WN == 4)
        asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                     : "+v"(prev.ah0), "+v"(prev.al0), "+v"(prev.ah1), "+v"(prev.al1), "+v"(prev.bh0), "+v"(prev.bh1),
                       "+v"(prev.bl0), "+v"(prev.bl1), "+v"(prev.bh2), "+v"(prev.bh3), "+v"(prev.bl2), "+v"(prev.bl3)
                     : "n"(allowed)
                     : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)"
                     : "+v"(prev.ah0), "+v"(prev.al0), "+v"(prev.ah1), "+v"(prev.al1), "+v"(prev.bh0), "+v"(prev.bh1),
                       "+v"(prev.bl0), "+v"(prev.bl1)
                     : "n"(allowed)
                     : "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(cur, a_addr + (unsigned)(slot_rd * WU * 16),
                 b_addr + (unsigned)(buf_rd * 4 * NCOLS * 16) + (unsigned)(k * dil * 16));
      // the scheduling barriers keep the MFMAs (register-only, not ordered by a memory clobber) between the reads of
      // this step and its DMA requests; those go last because a vector-memory request can hold the wave at issue until
      // the load path has room, and here that wait runs into the one for the next step's data
      __builtin_amdgcn_sched_barrier(0);
      if (j > 0 || c0 > 0) mma3(prev);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k == 0) issue_x(c0 + j / K + A, buf_wr);
      issue_w(s + D, slot_wr);
      ++s;
      slot_rd = slot_rd + 1 == S ? 0 : slot_rd + 1;
      slot_wr = slot_wr + 1 == S ? 0 : slot_wr + 1;
      if constexpr (k == K - 1) {
        buf_rd = buf_rd + 1 == NBUF ? 0 : buf_rd + 1;
        buf_wr = buf_wr + 1 == NBUF ? 0 : buf_wr + 1;
      }
    };
    gl_static_for<0, 2 * K>(step);
  }
  // the last step's products (an even number of steps: its fragments are in the second set)
  if constexpr (WN == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f1.ah0), "+v"(f1.al0), "+v"(f1.ah1), "+v"(f1.al1), "+v"(f1.bh0), "+v"(f1.bh1), "+v"(f1.bl0),
                   "+v"(f1.bl1), "+v"(f1.bh2), "+v"(f1.bh3), "+v"(f1.bl2), "+v"(f1.bl3)
                 :
                 : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f1.ah0), "+v"(f1.al0), "+v"(f1.ah1), "+v"(f1.al1), "+v"(f1.bh0), "+v"(f1.bh1), "+v"(f1.bl0),
                   "+v"(f1.bl1)
                 :
                 : "memory");
  __builtin_amdgcn_sched_barrier(0);
  mma3(f1);
  // nothing of this block may still be on its way into LDS when the block's LDS is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (a.flags & F_OUT_STAGED) {
    conv_epilogue_staged<WM, WN, 2>(a, acc, mt, mt_ok, wn, par, l31, t0, b);
    return;
  }
  // residual loads one 32-row block ahead of its stores for the 64 x 128 wave tile (the whole tile's would not fit in
  // registers beside the accumulators)
  conv_epilogue<WM, WN, MODE_PLAIN, WN == 2, EPI_OUT2>(a, acc, mt, mt_ok, wn, par, l31, t0, b, a.lengths ? a.lengths[b] : 0x7fffffff);
}

// ---------------------------------------------------------------------------------------------------------
// PREC_BF16 with a staged input, every operand through LDS by LDS-DMA (conv_glb_kernel): conv_gl_kernel's data path
// with one operand plane.  With one product a tap, a 64 x 64 wave tile of the register-staging kernel asks the vector
// cache for half a kilobyte of weight fragments per MFMA -- all it delivers -- and every wave of a block fetches the
// same ones; here a block's weights cross the load path once.
//
// The staged bf16 tensor is [C/8][L][8]: one run over time per 8-channel group, so the four waves of a block copy the
// four groups of a PAIR of 16-channel chunks, and a step is (chunk pair, tap): the weights of both chunks at that tap
// (two runs of the packed stream [Cin/16][tap][M/32][64 lanes][8 bf16], BM * 32 bytes each) in one ring slot, 16 MFMAs
// per wave (64 x 128 wave tile) between two barriers.  Per output the products are summed tap by tap over a chunk
// pair, not chunk by chunk as in conv_bf16_kernel: the same products, another fp32 summation order.
// ---------------------------------------------------------------------------------------------------------
// one step's operand fragments: weights a[h][i] (chunk h of the pair, m-tile i), activations b[h][j] (n-tile j)
struct GlbFrags {
  gl_u32x4 a00, a01, a10, a11, b00, b01, b02, b03, b10, b11, b12, b13;
};

template <int WN, int WGM, int WGN, int K, int D, int A>
__global__ __launch_bounds__(256, 2) void conv_glb_kernel(const ConvArgs a) {
  constexpr int WM = 2;
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  static_assert(WN == 2 || WN == 4, "wave tile 64 x 64 or 64 x 128");
  constexpr int S = D + 1, NBUF = A + 1;
  constexpr int NCOLS = BN + MAX_HALO;           // tile columns: a multiple of 64, halo included
  constexpr int HP = BM / 32;                    // 64-unit pieces of one chunk's weights (one per m-tile)
  constexpr int WU = 2 * HP * 64;                // 16-byte units of one step's weights (both chunks of the pair)
  constexpr int WPK = WU / 64 / 4;               // weight DMA instructions per wave and step
  constexpr int APC = NCOLS / 64;                // activation DMA instructions per wave and chunk pair (wave w: group w)
  static_assert(WGM * WGN == 4, "four waves per block");
  static_assert(A * K >= D, "an activation tile must have left the request window when it is first read");
  static_assert(WU % 256 == 0 && NCOLS % 64 == 0, "whole wave-instructions");
  static_assert(HP * 1024 + 1024 < 65536 && 3 * NCOLS * 16 + 2048 < 65536, "fragment offsets fit the offset field");

  extern __shared__ __attribute__((aligned(16))) uint4 gl_lds[];
  uint4* wq = gl_lds;                            // [S][2 chunks][BM/32][64 lanes]
  uint4* xq = gl_lds + S * WU;                   // [NBUF][2 chunks][2 halves][NCOLS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  int tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map), rowblk = blockIdx.y;
  if (gridDim.y == 2) xcd_tile_rows2(blockIdx.x, blockIdx.y, gridDim.x, a.xcd_map, tile, rowblk);
  const int t0 = tile * BN;
  const int mt0 = rowblk * HP;
  int mt[WM];
  bool mt_ok[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    mt[i] = mt0 + wm * WM + i;
    mt_ok[i] = true;                             // the launcher only takes M % BM == 0
  }
  const int npairs = a.Cin / 32;
  const int total = npairs * K;
  const int tstart = t0 + a.min_off;
  const int dil = a.tstep[0];

  // ---- DMA requests ----
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.Cin * K * a.M * 2, 0x00020000);
  const uint4* xs_b = reinterpret_cast<const uint4*>(a.x + (long long)b * a.x_bs);
  auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {
    const int sc = s < total ? s : total - 1;
    const int pr = sc / K, k = sc - pr * K;
#pragma unroll
    for (int j = 0; j < WPK; ++j) {
      const int piece = wave + 4 * j;            // of the slot: chunk h = piece / HP, m-tile piece % HP
      const int h = piece / HP, mi = piece - h * HP;
      const int unit0 = (((2 * pr + h) * K + k) * a.mt_total + mt0 + mi) * 64;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wq + slot * WU + piece * 64), 16,
                                               (unit0 + lane) * 16, 0, 0, 0);
    }
  };
  auto issue_x = [&](int c, int buf) __attribute__((always_inline)) {
    const int cc = c < npairs ? c : npairs - 1;
    // this wave's run: channel group 4 * cc + wave (chunk wave >> 1 of the pair, half wave & 1) over time, one buffer
    // descriptor, so that t < 0 and t >= L are out of range and read as zero
    const uint4* slab = xs_b + (long long)(cc * 4 + wave) * a.L;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(slab), 0, a.L * 16, 0x00020000);
#pragma unroll
    for (int j = 0; j < APC; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(xq + (buf * 4 + wave) * NCOLS + 64 * j), 16,
                                               (tstart + 64 * j + lane) * 16, 0, 0, 0);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- the requests of the steps before the first one, in the loop's own order ----
#pragma unroll
  for (int c = 0; c < A; ++c)
    if ((c - A) * K < -D) issue_x(c, c % NBUF);
#pragma unroll
  for (int v = -D; v < 0; ++v) {
    if (gl_floor_mod(v, K) == 0) issue_x(gl_floor_div(v, K) + A, (gl_floor_div(v, K) + A) % NBUF);
    issue_w(v + D, (v + D) % S);
  }

  int slot_rd = 0, slot_wr = D % S;              // ring slot of the current step, and of step + D
  int buf_rd = 0, buf_wr = A % NBUF;             // buffer of the current chunk pair, and of pair + A
  // Byte addresses: weight fragment (chunk h, m-tile i) of a slot at a_addr + (h * HP + i) * 1024; activation unit
  // (chunk h, half par, column) at b_addr + h * 2 * NCOLS * 16 + 512 * j, tap k a further k * dil * 16.
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void*)gl_lds;
  const unsigned a_addr = lds_base + (unsigned)(((wm * WM) * 64 + lane) * 16);
  const unsigned b_addr = lds_base + (unsigned)((S * WU + par * NCOLS + wn * 32 * WN + l31) * 16);
  GlbFrags f0, f1;
  {
    const gl_u32x4 z = {0u, 0u, 0u, 0u};
    f1.a00 = f1.a01 = f1.a10 = f1.a11 = f1.b00 = f1.b01 = f1.b02 = f1.b03 = f1.b10 = f1.b11 = f1.b12 = f1.b13 = z;
    f0 = f1;
  }
  auto read_frags = [&](GlbFrags& f, unsigned wa, unsigned xk) __attribute__((always_inline)) {
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:%c5\n\t"
        "ds_read_b128 %3, %4 offset:%c6"
        : "=&v"(f.a00), "=&v"(f.a01), "=&v"(f.a10), "=&v"(f.a11)
        : "v"(wa), "i"(HP * 1024), "i"(HP * 1024 + 1024)
        : "memory");
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:512\n\t"
        "ds_read_b128 %2, %4 offset:%c5\n\t"
        "ds_read_b128 %3, %4 offset:%c6"
        : "=&v"(f.b00), "=&v"(f.b01), "=&v"(f.b10), "=&v"(f.b11)
        : "v"(xk), "i"(2 * NCOLS * 16), "i"(2 * NCOLS * 16 + 512)
        : "memory");
    if constexpr (WN == 4)
      asm volatile(
          "ds_read_b128 %0, %4 offset:1024\n\t"
          "ds_read_b128 %1, %4 offset:1536\n\t"
          "ds_read_b128 %2, %4 offset:%c5\n\t"
          "ds_read_b128 %3, %4 offset:%c6"
          : "=&v"(f.b02), "=&v"(f.b03), "=&v"(f.b12), "=&v"(f.b13)
          : "v"(xk), "i"(2 * NCOLS * 16 + 1024), "i"(2 * NCOLS * 16 + 1536)
          : "memory");
  };
  // the products of a step, first chunk of the pair first, from a register set whose reads have been waited for
  auto mma2 = [&](const GlbFrags& f) __attribute__((always_inline)) {
    const gl_u32x4 Af[2][WM] = {{f.a00, f.a01}, {f.a10, f.a11}};
    gl_u32x4 Bf[2][WN];
    Bf[0][0] = f.b00; Bf[0][1] = f.b01; Bf[1][0] = f.b10; Bf[1][1] = f.b11;
    if constexpr (WN == 4) { Bf[0][2] = f.b02; Bf[0][3] = f.b03; Bf[1][2] = f.b12; Bf[1][3] = f.b13; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Af[h][i]),
                                                              __builtin_bit_cast(bf16x8, Bf[h][j]), acc[i][j], 0, 0, 0);
  };
  int s = 0;
  for (int c0 = 0; c0 < npairs; c0 += 2) {         // two chunk pairs a trip: the register sets swap at compile time
    auto step = [&](auto jj) __attribute__((always_inline)) {
      constexpr int j = decltype(jj)::value;
      constexpr int k = j % K;
      constexpr int allowed = (D - 1) * WPK + APC * gl_tap0_in_window(k, K, D);
      GlbFrags& cur = (j & 1) ? f1 : f0;
      GlbFrags& prev = (j & 1) ? f0 : f1;
      // the wait names every register of the previous set, so nothing touches them before their reads have landed
      if constexpr (WN == 4)
        asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                     : "+v"(prev.a00), "+v"(prev.a01), "+v"(prev.a10), "+v"(prev.a11), "+v"(prev.b00), "+v"(prev.b01),
                       "+v"(prev.b10), "+v"(prev.b11), "+v"(prev.b02), "+v"(prev.b03), "+v"(prev.b12), "+v"(prev.b13)
                     : "n"(allowed)
                     : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)"
                     : "+v"(prev.a00), "+v"(prev.a01), "+v"(prev.a10), "+v"(prev.a11), "+v"(prev.b00), "+v"(prev.b01),
                       "+v"(prev.b10), "+v"(prev.b11)
                     : "n"(allowed)
                     : "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(cur, a_addr + (unsigned)(slot_rd * WU * 16),
                 b_addr + (unsigned)(buf_rd * 4 * NCOLS * 16) + (unsigned)(k * dil * 16));
      __builtin_amdgcn_sched_barrier(0);
      if (j > 0 || c0 > 0) mma2(prev);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k == 0) issue_x(c0 + j / K + A, buf_wr);
      issue_w(s + D, slot_wr);
      ++s;
      slot_rd = slot_rd + 1 == S ? 0 : slot_rd + 1;
      slot_wr = slot_wr + 1 == S ? 0 : slot_wr + 1;
      if constexpr (k == K - 1) {
        buf_rd = buf_rd + 1 == NBUF ? 0 : buf_rd + 1;
        buf_wr = buf_wr + 1 == NBUF ? 0 : buf_wr + 1;
      }
    };
    gl_static_for<0, 2 * K>(step);
  }
  // the last step's products (an even number of steps: its fragments are in the second set)
  if constexpr (WN == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f1.a00), "+v"(f1.a01), "+v"(f1.a10), "+v"(f1.a11), "+v"(f1.b00), "+v"(f1.b01), "+v"(f1.b10),
                   "+v"(f1.b11), "+v"(f1.b02), "+v"(f1.b03), "+v"(f1.b12), "+v"(f1.b13)
                 :
                 : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f1.a00), "+v"(f1.a01), "+v"(f1.a10), "+v"(f1.a11), "+v"(f1.b00), "+v"(f1.b01), "+v"(f1.b10),
                   "+v"(f1.b11)
                 :
                 : "memory");
  __builtin_amdgcn_sched_barrier(0);
  mma2(f1);
  // nothing of this block may still be on its way into LDS when the block's LDS is handed to the next one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (a.flags & F_OUT_STAGED) {
    conv_epilogue_staged<WM, WN, 1>(a, acc, mt, mt_ok, wn, par, l31, t0, b);
    return;
  }
  conv_epilogue<WM, WN, MODE_PLAIN, WN == 2, EPI_OUT2B>(a, acc, mt, mt_ok, wn, par, l31, t0, b, a.lengths ? a.lengths[b] : 0x7fffffff);
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
ProfTicket prof_begin(hipStream_t stream) {
  ProfTicket t;
  if (!g_prof_on) return t;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    t.e0 = prof_event();
    t.e1 = prof_event();
  }
  t.on = true;
  (void)hipEventRecord(t.e0, stream);
  return t;
}

void prof_end(const ProfTicket& t, const char* name, bool gate, const ConvArgs& a, hipStream_t stream) {
  if (!t.on) return;
  (void)hipEventRecord(t.e1, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfClass& c = g_prof[name];
  c.flops += 2.0 * a.M * a.Cin * a.K * (double)a.L * a.B;     // algorithmic: dense over every position
  // every input and output element once, fp32: rows * L outputs (half of that after the gate), residual and
  // accumulated-output reads where the epilogue has them, the packed weights once
  const double outs = (gate ? 0.5 : 1.0) * a.M;
  const int reads = ((a.flags & (F_RES | F_RES_SUB)) ? 1 : 0) + ((a.flags & F_ACCUM) ? 1 : 0);
  const bool plain_bf16 = strstr(name, "conv_bf16_kernel") || strstr(name, "conv_glb_bf16");
  const double wbytes = plain_bf16 ? 2.0
                        : strstr(name, "conv_mfma") || strstr(name, "conv_f16x2") || strstr(name, "conv_gl") ||
                                strstr(name, "resblock_pair") || strstr(name, "resblock_chain") ? 4.0 : 6.0;
  const double sb = plain_bf16 ? 2.0 : 4.0;                                   // bytes per staged element
  const double in_b = (a.flags & F_IN_STAGED) ? sb : 4.0, out_b = (a.flags & F_OUT_STAGED) ? sb : 4.0;
  c.bytes += (double)a.L * a.B * (in_b * a.Cin + outs * (out_b + 4.0 * reads + (a.out2 ? sb : 0.0))) + wbytes * a.M * a.Cin * a.K;
  c.count += 1;
  c.pending.emplace_back(t.e0, t.e1);
}

template <typename Launch>
static hipError_t timed_launch(const char* name, bool gate, const ConvArgs& a, hipStream_t stream, Launch&& launch) {
  const ProfTicket t = prof_begin(stream);
  launch();
  const hipError_t err = hipGetLastError();
  prof_end(t, name, gate, a, stream);
  return err;
}

template <int WM, int WN, int WGM, int WGN, int MODE, bool VEC>
static hipError_t launch_f32(ConvArgs& a, hipStream_t stream) {
  constexpr int BN = 32 * WN * WGN;
  a.width = BN + (a.width);          // caller passes halo span in a.width
  a.rs = (a.width + 3 + 3) & ~3;     // + up to 3 lead columns when the tile start is rounded down to a multiple of 4
  const int units = MODE == MODE_GATE ? a.mt_total / 2 : (a.mt_total + WM - 1) / WM;
  dim3 grid((a.L + BN - 1) / BN, (units + WGM - 1) / WGM, a.B);
  const size_t lds = (size_t)CI_CHUNK * a.rs * sizeof(float);
  char name[96];
  if constexpr (MODE == MODE_PLAIN) {
    if (a.split_row) {                                 // the instantiation with a second destination
      snprintf(name, sizeof name, "conv_mfma_kernel<%d,%d,%d,%d,%d,%d,split>", WM, WN, WGM, WGN, MODE, (int)VEC);
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, WGM, WGN, MODE, VEC, EPI_SPLIT>), grid, dim3(256), lds, stream, a);
      });
    }
  }
  if (a.split_row) return hipErrorInvalidValue;
  snprintf(name, sizeof name, "conv_mfma_kernel<%d,%d,%d,%d,%d,%d>", WM, WN, WGM, WGN, MODE, (int)VEC);
  return timed_launch(name, MODE == MODE_GATE, a, stream, [&] {
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, WGM, WGN, MODE, VEC>), grid, dim3(256), lds, stream, a);
  });
}

template <int WM, int WN, int WGM, int WGN, int MODE>
static hipError_t launch_b3(ConvArgs& a, hipStream_t stream) {
  constexpr int BN = 32 * WN * WGN;
  a.width = BN + (a.width);
  a.rs = a.width;                    // columns per (plane, half) slab
  const int units = MODE == MODE_GATE ? a.mt_total / 2 : (a.mt_total + WM - 1) / WM;
  dim3 grid((a.L + BN - 1) / BN, (units + WGM - 1) / WGM, a.B);
  const size_t lds = (size_t)6 * a.rs * sizeof(uint4);
  char name[96];
  // deferred LayerNorm is the split-K kernel's: its consumers pay ~2 us there, here (every thread of a block with the
  // statistics and 32 more registers) a frame-rate conv took 80 us instead of 45
  if (a.ln_in_stats || a.ln_res_stats || a.ln_out_stats) return hipErrorInvalidValue;
  if constexpr (MODE == MODE_PLAIN) {
    if (a.split_row) {                                 // the instantiation with a second destination
      snprintf(name, sizeof name, "conv_bf16x3_kernel<%d,%d,%d,%d,%d,split>", WM, WN, WGM, WGN, MODE);
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_bf16x3_kernel<WM, WN, WGM, WGN, MODE, EPI_SPLIT>), grid, dim3(256), lds, stream, a);
      });
    }
  }
  if (a.split_row) return hipErrorInvalidValue;
  snprintf(name, sizeof name, "conv_bf16x3_kernel<%d,%d,%d,%d,%d>", WM, WN, WGM, WGN, MODE);
  return timed_launch(name, MODE == MODE_GATE, a, stream, [&] {
    hipLaunchKernelGGL((conv_bf16x3_kernel<WM, WN, WGM, WGN, MODE>), grid, dim3(256), lds, stream, a);
  });
}

template <int WN, int WGM, int WGN, int K, int D, int A>
static hipError_t launch_glb_cfg(ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  constexpr size_t lds = ((size_t)(D + 1) * BM * 4 + (size_t)(A + 1) * 4 * (BN + MAX_HALO)) * sizeof(uint4);
  static_assert(lds <= 80 * 1024, "two blocks to a CU");
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(&conv_glb_kernel<WN, WGM, WGN, K, D, A>), lds); e != hipSuccess) return e;
  dim3 grid((a.L + BN - 1) / BN, a.M / BM, a.B);
  char name[96];
  snprintf(name, sizeof name, "conv_glb_bf16_kernel<%d,%d,%d,%d,%d,%d>", WN, WGM, WGN, K, D, A);
  return timed_launch(name, false, a, stream, [&] {
    hipLaunchKernelGGL((conv_glb_kernel<WN, WGM, WGN, K, D, A>), grid, dim3(256), lds, stream, a);
  });
}

// Staged-input plain bf16 convs of the generator (3, 7 or 11 taps, rows in 128s, channels in 64s) whose grid fills the
// chip go to the LDS-DMA kernel; VITS_CONV_GLB=0 keeps them on the register-staging kernel (A/B measurements).
// shape and grid of a launch the kernel takes: 0 = none, else the block tile width
static int glb_tile_width(int M, int Cin, int K, int dil, int B, int L) {
  static const bool on = !(getenv("VITS_CONV_GLB") && getenv("VITS_CONV_GLB")[0] == '0');
  if (!on || M % 64 || Cin % 64 || (K != 3 && K != 7 && K != 11) || dil < 1) return 0;
  if ((long long)Cin * K * M * 2 >= (1LL << 31) || (long long)L * 16 >= (1LL << 31)) return 0;
  const bool tall = M % 128 == 0;                             // 128-row blocks, else 64 x 256
  auto blocks_of = [&](int bn) { return (long long)((L + bn - 1) / bn) * (M / (tall ? 128 : 64)) * B; };
  const long long slots = 2LL * device_cu_count();            // two blocks to a CU
  const bool wide = !tall || blocks_of(256) > slots;
  if (blocks_of(wide ? 256 : 128) <= slots / 2) return 0;
  return wide ? 256 : 128;
}

// A staged-input plain bf16 conv of this shape and size runs on the LDS-DMA kernel (which is also the one that can
// write the second, staged copy of its result): what the generator asks before it stages a stage's input.
bool conv_takes_glb(const ConvDesc& d, int B, int L) {
  return d.prec == PREC_BF16 && d.mode == MODE_PLAIN && glb_tile_width(d.M, d.Cin, d.K, d.tstep[0], B, L) > 0;
}

static bool try_launch_glb(ConvArgs& a, hipStream_t stream, hipError_t& err) {
  if (!(a.flags & F_IN_STAGED) || a.in_cbias || a.cbias || (a.flags & (F_IN_MASK | F_IN_LRELU)) || a.split_row) return false;
  const int width = glb_tile_width(a.M, a.Cin, a.K, a.tstep[0], a.B, a.L);
  if (!width) return false;
  const bool wide = width == 256;
  if (a.M % 128)
    err = a.K == 3 ? launch_glb_cfg<2, 1, 4, 3, 4, 2>(a, stream) : a.K == 7 ? launch_glb_cfg<2, 1, 4, 7, 4, 1>(a, stream)
                                                                           : launch_glb_cfg<2, 1, 4, 11, 4, 1>(a, stream);
  else if (wide)
    err = a.K == 3 ? launch_glb_cfg<4, 2, 2, 3, 3, 1>(a, stream) : a.K == 7 ? launch_glb_cfg<4, 2, 2, 7, 3, 1>(a, stream)
                                                                           : launch_glb_cfg<4, 2, 2, 11, 3, 1>(a, stream);
  else
    err = a.K == 3 ? launch_glb_cfg<2, 2, 2, 3, 4, 2>(a, stream) : a.K == 7 ? launch_glb_cfg<2, 2, 2, 7, 4, 1>(a, stream)
                                                                           : launch_glb_cfg<2, 2, 2, 11, 4, 1>(a, stream);
  return true;
}

template <int WM, int WN, int WGM, int WGN, int MODE>
static hipError_t launch_b1(ConvArgs& a, hipStream_t stream) {
  if constexpr (MODE == MODE_PLAIN) {
    hipError_t err = hipSuccess;
    if (try_launch_glb(a, stream, err)) return err;
  }
  // of the plain convs only the LDS-DMA kernel writes the second copy (see conv_takes_glb)
  if (a.out2 && MODE != MODE_UP8 && MODE != MODE_UP4) return hipErrorInvalidValue;
  constexpr int BN = 32 * WN * WGN;
  a.width = BN + (a.width);
  a.rs = a.width;                    // columns per half slab
  const int units = MODE == MODE_GATE ? a.mt_total / 2 : (a.mt_total + WM - 1) / WM;
  dim3 grid((a.L + BN - 1) / BN, (units + WGM - 1) / WGM, a.B);
  const size_t lds = (size_t)2 * a.rs * sizeof(uint4);
  char name[96];
  if constexpr (MODE == MODE_PLAIN) {
    if (a.flags & F_IN_STAGED) {
      snprintf(name, sizeof name, "conv_bf16_kernel<%d,%d,%d,%d,%d,staged>", WM, WN, WGM, WGN, MODE);
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_bf16_kernel<WM, WN, WGM, WGN, MODE, true>), grid, dim3(256), lds, stream, a);
      });
    }
  }
  if (a.flags & F_IN_STAGED) return hipErrorInvalidValue;
  snprintf(name, sizeof name, "conv_bf16_kernel<%d,%d,%d,%d,%d>", WM, WN, WGM, WGN, MODE);
  return timed_launch(name, MODE == MODE_GATE, a, stream, [&] {
    hipLaunchKernelGGL((conv_bf16_kernel<WM, WN, WGM, WGN, MODE>), grid, dim3(256), lds, stream, a);
  });
}

constexpr int F16X2_RING = 6;
// The ring variant takes grids of at most VITS_CONV_RING blocks per CU (default 2, what its register count lets be
// resident; 0 keeps every launch on the occupancy-oriented kernel, for A/B measurements).
// A front-end launch beside a generator (throughput mode) stays on the occupancy-oriented kernel whatever its grid: the
// ring variant's blocks need a slot as large as the generator's own blocks and wait for one (C2: +0.45 % without it).
static int f16x2_ring_blocks_per_cu() {
  static const int n = [] {
    const char* e = getenv("VITS_CONV_RING");
    return e && e[0] ? atoi(e) : 2;
  }();
  return front_end_overlapped() ? 0 : n;
}

// VITS_CONV_GL=0 keeps staged-input convs on the register-staged kernel (A/B measurements).
static bool gl_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_CONV_GL");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <int WN, int WGM, int WGN, int K, int D, int A>
static hipError_t launch_gl_cfg(ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 64 * WGM, BN = 32 * WN * WGN;
  constexpr size_t lds = ((size_t)(D + 1) * BM * 4 + (size_t)(A + 1) * 4 * (BN + MAX_HALO)) * sizeof(uint4);
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(&conv_gl_kernel<WN, WGM, WGN, K, D, A>), lds); e != hipSuccess) return e;
  dim3 grid((a.L + BN - 1) / BN, a.M / BM, a.B);
  char name[96];
  snprintf(name, sizeof name, "conv_gl_kernel<%d,%d,%d,%d,%d,%d>", WN, WGM, WGN, K, D, A);
  return timed_launch(name, false, a, stream, [&] {
    hipLaunchKernelGGL((conv_gl_kernel<WN, WGM, WGN, K, D, A>), grid, dim3(256), lds, stream, a);
  });
}

// VITS_CONV_GL_WIDE=0: 128 x 128 block tiles for the >= 128-row layers instead of 128 x 256 (A/B measurements).
static bool gl_wide() {
  static const bool on = [] {
    const char* e = getenv("VITS_CONV_GL_WIDE");
    return !(e && e[0] == '0');
  }();
  return on;
}

// Staged-input plain convs of the generator (3, 7 or 11 taps) whose grid fills the chip go to the LDS-DMA kernel.
// What bounds it is the CU's load path: every block pulls the whole packed weight stream through it once per output
// tile, so tiles are as wide in time as two blocks' LDS allows -- 128 rows x 256 steps (wave tile 64 x 128) when the
// rows allow, else 64 x 256.  Returns false when the launch is not one of them.
static bool try_launch_gl(ConvArgs& a, hipStream_t stream, hipError_t& err) {
  if (!gl_enabled() || !(a.flags & F_IN_STAGED) || a.in_cbias || a.cbias || (a.flags & (F_IN_MASK | F_IN_LRELU))) return false;
  if (a.M % 64 || a.Cin % 32 || (a.K != 3 && a.K != 7 && a.K != 11) || a.tstep[0] < 1) return false;   // chunks go in pairs
  if ((long long)a.Cin * a.K * a.M * 4 >= (1LL << 31) || (long long)a.L * 16 >= (1LL << 31)) return false;
  const bool tall = a.M % 128 == 0;
  auto blocks_of = [&](int bn) { return (long long)((a.L + bn - 1) / bn) * (a.M / (tall ? 128 : 64)) * a.B; };
  const long long slots = 2LL * device_cu_count();            // two blocks to a CU
  // 128 x 256 tiles when that still gives every slot a block, else 128 x 128; a grid that cannot even give every CU a
  // block stays with the register-ring variant of the other kernel (batch-1 latency)
  const bool wide = tall && gl_wide() && blocks_of(256) > slots;
  static const long long min_blocks = [] {
    const char* e = getenv("VITS_CONV_GL_MIN_BLOCKS");
    return e && e[0] ? atoll(e) : -1LL;
  }();
  if (blocks_of(wide || !tall ? 256 : 128) <= (min_blocks >= 0 ? min_blocks : slots / 2)) return false;
  // 7 and 11 taps: the 16 x 16 x 32 build, with the tile width that needs the fewest block rounds
  if (conv_gl16_takes(a)) {
    err = launch_conv_gl16(a, tall ? conv_gl16_tile_width(a) : 0, stream);
    return true;
  }
  // look-ahead D steps; the activation tile goes out A chunks ahead with A * K >= D
  if (wide)
    err = a.K == 3 ? launch_gl_cfg<4, 2, 2, 3, 3, 1>(a, stream) : a.K == 7 ? launch_gl_cfg<4, 2, 2, 7, 3, 1>(a, stream)
                                                                          : launch_gl_cfg<4, 2, 2, 11, 3, 1>(a, stream);
  else if (tall)
    err = a.K == 3 ? launch_gl_cfg<2, 2, 2, 3, 4, 2>(a, stream) : a.K == 7 ? launch_gl_cfg<2, 2, 2, 7, 4, 1>(a, stream)
                                                                          : launch_gl_cfg<2, 2, 2, 11, 4, 1>(a, stream);
  else
    err = a.K == 3 ? launch_gl_cfg<2, 1, 4, 3, 4, 2>(a, stream) : a.K == 7 ? launch_gl_cfg<2, 1, 4, 7, 4, 1>(a, stream)
                                                                          : launch_gl_cfg<2, 1, 4, 11, 4, 1>(a, stream);
  return true;
}

template <int WM, int WN, int WGM, int WGN, int MODE>
static hipError_t launch_h2(ConvArgs& a, hipStream_t stream) {
  if constexpr (MODE == MODE_PLAIN) {
    hipError_t err = hipSuccess;
    if (try_launch_gl(a, stream, err)) return err;
  }
  constexpr int BN = 32 * WN * WGN;
  a.width = BN + (a.width);
  a.rs = a.width;                    // columns per (plane, half) slab
  const int units = MODE == MODE_GATE ? a.mt_total / 2 : (a.mt_total + WM - 1) / WM;
  dim3 grid((a.L + BN - 1) / BN, (units + WGM - 1) / WGM, a.B);
  const size_t lds = (size_t)4 * a.rs * sizeof(uint4);
  char name[96];
  if (a.split_row) {
    // two destinations: one tile shape is built with that epilogue (the 128-row block of launch_conv's direct route)
    if constexpr (MODE == MODE_PLAIN && WM == 2 && WN == 2 && WGM == 2 && WGN == 2) {
      if (a.flags & F_IN_STAGED) return hipErrorInvalidValue;
      const bool ring = (long long)grid.x * grid.y * grid.z <= (long long)f16x2_ring_blocks_per_cu() * device_cu_count();
      snprintf(name, sizeof name, "conv_f16x2_kernel<%d,%d,%d,%d,%d,split%s>", WM, WN, WGM, WGN, MODE, ring ? ",ring" : "");
      return timed_launch(name, false, a, stream, [&] {
        if (ring)
          hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE, false, F16X2_RING, EPI_SPLIT>), grid, dim3(256), lds,
                             stream, a);
        else
          hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE, false, 0, EPI_SPLIT>), grid, dim3(256), lds, stream, a);
      });
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (MODE == MODE_PLAIN && WM == 2 && (WN == 2 || (WN == 1 && WGM == 2))) {
    // a grid of at most two blocks per CU is a set of serial chains: the weight-ring variant
    if ((long long)grid.x * grid.y * grid.z <= (long long)f16x2_ring_blocks_per_cu() * device_cu_count()) {
      const bool stg = a.flags & F_IN_STAGED;
      snprintf(name, sizeof name, "conv_f16x2_kernel<%d,%d,%d,%d,%d,%s,%d>", WM, WN, WGM, WGN, MODE,
               stg ? "staged" : "false", F16X2_RING);
      return timed_launch(name, false, a, stream, [&] {
        if (stg)
          hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE, true, F16X2_RING>), grid, dim3(256), lds,
                             stream, a);
        else
          hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE, false, F16X2_RING>), grid, dim3(256), lds,
                             stream, a);
      });
    }
  }
  if constexpr (MODE == MODE_PLAIN) {
    if (a.flags & F_IN_STAGED) {
      snprintf(name, sizeof name, "conv_f16x2_kernel<%d,%d,%d,%d,%d,staged>", WM, WN, WGM, WGN, MODE);
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE, true>), grid, dim3(256), lds, stream, a);
      });
    }
  }
  if (a.flags & F_IN_STAGED) return hipErrorInvalidValue;
  snprintf(name, sizeof name, "conv_f16x2_kernel<%d,%d,%d,%d,%d>", WM, WN, WGM, WGN, MODE);
  return timed_launch(name, MODE == MODE_GATE, a, stream, [&] {
    hipLaunchKernelGGL((conv_f16x2_kernel<WM, WN, WGM, WGN, MODE>), grid, dim3(256), lds, stream, a);
  });
}

// ---------------------------------------------------------------------------------------------------------
// PREC_BF16X3, split-K kernel for short sequences (MODE_PLAIN / MODE_GATE).
//
// At phoneme and frame rate (L = 50 .. a few hundred) a conv has few output tiles but a deep reduction (up to
// 768 channels x 3 taps), and the one-role kernel walks that reduction as one serial chain per block.  Here the four
// waves of a block all own the SAME 64 x (32 * WN) output tile and split the channel chunks between them
// (wave w takes chunks w, w + 4, ...): each wave stages its chunks into a private LDS region, so the main loop has no
// block barrier at all, and the chain is a quarter as long.  The four partial accumulators are then summed through LDS
// in a fixed order ((w0 + w2) + (w1 + w3)) and wave 0 runs the epilogue.
// ---------------------------------------------------------------------------------------------------------
// NW = waves per block that share the reduction: 4, or 8 (WN = 1 only) for the deepest reductions on the smallest
// grids (768 channels at phoneme rate: six blocks on the chip, so the chain length is all that counts).
// EX: optional features (EPI_SPLIT: a second destination; EPI_LN: deferred LayerNorm of input, residual and result),
// each an instantiation of its own so that the plain one keeps its registers
template <int WN, int MODE, int NW = 4, int EX = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : WN <= 2 ? 2 : 1) void conv_b3ks_kernel(const ConvArgs a) {
  static_assert(NW == 4 || (NW == 8 && WN == 1), "eight waves: one n-tile");
  constexpr int WM = 2;
  constexpr int BN = 32 * WN;
  constexpr int NCT = (BN + MAX_HALO + 63) / 64;             // tile columns per lane
  static_assert(MODE == MODE_PLAIN || MODE == MODE_GATE, "one row class only");

  extern __shared__ __attribute__((aligned(16))) uint4 xq[];    // [4 waves][3 planes][2 halves][ncols] of 8 bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * BN;
  const int len_b = a.lengths ? a.lengths[b] : 0x7fffffff;
  const int ncols = a.rs;
  uint4* xw = xq + wave * 6 * ncols;

  int mt[WM];
  bool mt_ok[WM];
  if constexpr (MODE == MODE_GATE) {
    const int half = a.mt_total >> 1;
    const int p = blockIdx.y;
    mt_ok[0] = mt_ok[1] = p < half;
    mt[0] = p < half ? p : half - 1;
    mt[1] = mt[0] + half;
  } else {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      mt_ok[i] = (int)blockIdx.y * WM + i < a.mt_total;
      mt[i] = mt_ok[i] ? (int)blockIdx.y * WM + i : a.mt_total - 1;
    }
  }

  const float* xb = a.x + (long long)b * a.x_bs;
  const int tstart = t0 + a.min_off;
  const int wcols = a.width;
  constexpr bool LN = (EX & EPI_LN) != 0;
  float st[NCT][16], lng[LN ? 16 : 1], lnb[LN ? 16 : 1];
  auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = lane + 64 * q, t = tstart + col;
      if (64 * q >= wcols) continue;                   // narrow tile: this column slot is empty for every lane
      const unsigned toff = (col < wcols && t >= 0 && t < a.L) ? (unsigned)t : 0u;
      const float* xc = xb + (long long)chunk * 16 * a.x_cs;
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) st[q][ci] = (xc + (long long)ci * a.x_cs)[toff];    // raw: store_chunk zeroes
    }
    if constexpr (LN) {
      if (a.ln_in_stats) {                             // the chunk's LayerNorm gains and biases travel with it
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) {
          lng[ci] = a.ln_in_gamma[chunk * 16 + ci];
          lnb[ci] = a.ln_in_beta[chunk * 16 + ci];
        }
      }
    }
  };
  // deferred LayerNorm of the input: mean and 1 / std of this lane's tile columns, once -- requested behind the first
  // chunk's loads, so that the two round trips overlap
  float lnm[LN ? NCT : 1], lnr[LN ? NCT : 1];
  auto ln_columns = [&]() __attribute__((always_inline)) {
    if constexpr (LN) {
      if (!a.ln_in_stats) return;
      const int units = a.Cin >> 6;
#pragma unroll
      for (int q = 0; q < NCT; ++q) {
        const int t = tstart + lane + 64 * q;
        const int tc = t < 0 ? 0 : t < a.L ? t : a.L - 1;
        ln_combine(a.ln_in_stats + ((long long)b * a.L + tc) * units * 2, units, lnm[q], lnr[q]);
      }
    }
  };
  auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
    const float* cbp = a.in_cbias ? a.in_cbias + (long long)b * a.in_cbias_bs + chunk * 16 : nullptr;
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
      const int col = lane + 64 * q, t = tstart + col;
      if (64 * q >= wcols) continue;
      bool ok = t >= 0 && t < a.L;
      if (a.flags & F_IN_MASK) ok = ok && t < len_b;
      unsigned ph[8], pm[8], pl[8];
#pragma unroll
      for (int c2 = 0; c2 < 8; ++c2) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float x = st[q][2 * c2 + u];
          if constexpr (LN) {
            if (a.ln_in_stats) x = (x - lnm[q]) * lnr[q] * lng[2 * c2 + u] + lnb[2 * c2 + u];
          }
          if (cbp) x += cbp[2 * c2 + u];
          x *= a.in_scale;
          if (a.flags & F_IN_LRELU) x = fmaxf(x, x * a.in_slope);
          v[u] = ok ? x : 0.f;
        }
        split3_pair(v[0], v[1], ph[c2], pm[c2], pl[c2]);
      }
      if (col < wcols) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          xw[(0 * 2 + hf) * ncols + col] = make_uint4(ph[4 * hf], ph[4 * hf + 1], ph[4 * hf + 2], ph[4 * hf + 3]);
          xw[(1 * 2 + hf) * ncols + col] = make_uint4(pm[4 * hf], pm[4 * hf + 1], pm[4 * hf + 2], pm[4 * hf + 3]);
          xw[(2 * 2 + hf) * ncols + col] = make_uint4(pl[4 * hf], pl[4 * hf + 1], pl[4 * hf + 2], pl[4 * hf + 3]);
        }
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint4* w16 = reinterpret_cast<const uint4*>(a.w);
  unsigned wofs[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) wofs[i] = (unsigned)(mt[i] * 192 + lane);
  const long long it_stride = (long long)a.mt_total * 192;
  const int nchunks = a.Cin / 16;
  const int my_chunks = nchunks > wave ? (nchunks - wave + NW - 1) / NW : 0;
  const int colbase = l31 + par * ncols - a.min_off;

  uint4 Ah[2][WM], Am[2][WM], Al[WM], Bh[2][WN], Bm[2][WN], Bl[WN];
  auto lda = [&](uint4 (&d)[WM], long long base, int plane) {
#pragma unroll
    for (int i = 0; i < WM; ++i) d[i] = (w16 + base + plane * 64)[wofs[i]];
  };
  auto ldb = [&](uint4 (&d)[WN], int plane, int off) {
#pragma unroll
    for (int j = 0; j < WN; ++j) d[j] = xw[plane * 2 * ncols + off + j * 32];
  };
  auto mma = [&](const uint4 (&af)[WM], const uint4 (&bf)[WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                            __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
  };

  if (my_chunks > 0) {
    int chunk = wave, k = 0, boff = colbase + a.toff0[0];
    const long long w0 = (long long)chunk * a.K * it_stride;
    lda(Ah[0], w0, 0);
    lda(Am[0], w0, 1);
    lda(Al, w0, 2);
    load_chunk(chunk);
    ln_columns();
    store_chunk(chunk);
    if (chunk + NW < nchunks) load_chunk(chunk + NW);
    ldb(Bh[0], 0, boff);
    ldb(Bm[0], 1, boff);
    ldb(Bl, 2, boff);

    // One tap with register set c current and set n prefetched (same schedule as the wave-specialised consumer).
    auto step = [&](auto P) __attribute__((always_inline)) {
      constexpr int c = decltype(P)::value, n = c ^ 1;
      const bool last = k == a.K - 1;
      // next tap: the same chunk, or tap 0 of this wave's next chunk (clamped to the last one at the very end)
      const int nchunk = last ? (chunk + NW < nchunks ? chunk + NW : chunk) : chunk;
      const int nk = last ? 0 : k + 1;
      const long long wb = ((long long)nchunk * a.K + nk) * it_stride;
      const int boffn = last ? boff : boff + a.tstep[0];
      lda(Ah[n], wb, 0);
      lda(Am[n], wb, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(Ah[c], Bl);                                           // h * l
      __builtin_amdgcn_sched_barrier(0);
      if (!last) {
        ldb(Bh[n], 0, boffn);
        ldb(Bm[n], 1, boffn);
        ldb(Bl, 2, boffn);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(Al, Bh[c]);                                           // l * h
      __builtin_amdgcn_sched_barrier(0);
      lda(Al, wb, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(Am[c], Bm[c]);                                        // m * m
      mma(Ah[c], Bm[c]);                                        // h * m
      mma(Am[c], Bh[c]);                                        // m * h
      mma(Ah[c], Bh[c]);                                        // h * h
      __builtin_amdgcn_sched_barrier(0);
      k = nk;
      boff = boffn;
      if (last) {
        // this wave's LDS region is free (its reads were consumed by the MFMAs above): stage the next chunk
        const bool more = chunk + NW < nchunks;
        chunk += NW;
        boff = colbase + a.toff0[0];
        if (more) {
          store_chunk(chunk);
          if (chunk + NW < nchunks) load_chunk(chunk + NW);
          ldb(Bh[n], 0, boff);
          ldb(Bm[n], 1, boff);
          ldb(Bl, 2, boff);
        }
      }
    };
    const int S = my_chunks * a.K;
    int s = 0;
    for (; s + 1 < S; s += 2) {
      step(IntC<0>{});
      step(IntC<1>{});
    }
    if (s < S) step(IntC<0>{});
  }

  // the residual's LayerNorm statistics (wave 0 runs the epilogue): requested before the reduction's barriers
  float ln_pre[WN][2];
  const bool ln_res = LN && a.ln_res_stats && a.res && wave == 0;
  if (ln_res) {
    const int units = a.M >> 6;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int t = t0 + j * 32 + l31;
      ln_combine(a.ln_res_stats + ((long long)b * a.L + (t < a.L ? t : a.L - 1)) * units * 2, units, ln_pre[j][0], ln_pre[j][1]);
    }
  }
  // ---- fixed-order reduction of the four partial sums through LDS: (w0 + w2) + (w1 + w3) ----
  __syncthreads();                                              // every wave is done with its staging region
  float* red = reinterpret_cast<float*>(xq);                    // [NW / 2 slots][WM * WN * 16][64 lanes]
  constexpr int NACC = WM * WN * 16;
  auto put = [&](int slot) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(slot * NACC + (i * WN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(slot * NACC + (i * WN + j) * 16 + r) * 64 + lane];
  };
  if constexpr (NW == 8) {                                      // waves 4..7 into 0..3 first, then as with four
    if (wave >= 4) put(wave - 4);
    __syncthreads();
    if (wave < 4) add(wave);
    __syncthreads();
  }
  if (wave == 2 || wave == 3) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    add(0);
    conv_epilogue<WM, WN, MODE, false, EX>(a, acc, mt, mt_ok, 0, par, l31, t0, b, len_b, ln_res ? ln_pre : nullptr);
  }
}

bool xcd_map_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_XCD_MAP");
    return !(e && e[0] == '0');
  }();
  return on;
}

int device_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

// VITS_CONV_KS=0 keeps short sequences on the one-role kernel (for A/B measurements).
static bool ks_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_CONV_KS");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <int WN, int MODE, int NW = 4>
static hipError_t launch_b3ks(ConvArgs& a, hipStream_t stream) {
  constexpr int BN = 32 * WN;
  a.width = BN + (a.width);
  a.rs = a.width;
  const int units = MODE == MODE_GATE ? a.mt_total / 2 : (a.mt_total + 1) / 2;
  dim3 grid((a.L + BN - 1) / BN, units, a.B);
  const size_t stage = (size_t)NW * 6 * a.rs * sizeof(uint4), red = (size_t)WN * 16384 * (NW / 4);
  const size_t lds = stage > red ? stage : red;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  char name[96];
  const bool ln = a.ln_in_stats || a.ln_res_stats || a.ln_out_stats;
  if (NW == 4) snprintf(name, sizeof name, "conv_b3ks_kernel<%d,%d%s>", WN, MODE, ln ? ",ln" : a.split_row ? ",split" : "");
  else snprintf(name, sizeof name, "conv_b3ks_kernel<%d,%d,%d%s>", WN, MODE, NW, ln ? ",ln" : a.split_row ? ",split" : "");
  if constexpr (MODE == MODE_PLAIN) {
    if (ln && a.split_row) return hipErrorInvalidValue;
    if (ln)
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_b3ks_kernel<WN, MODE, NW, EPI_LN>), grid, dim3(64 * NW), lds, stream, a);
      });
    if (a.split_row)
      return timed_launch(name, false, a, stream, [&] {
        hipLaunchKernelGGL((conv_b3ks_kernel<WN, MODE, NW, EPI_SPLIT>), grid, dim3(64 * NW), lds, stream, a);
      });
  }
  if (ln || a.split_row) return hipErrorInvalidValue;
  return timed_launch(name, MODE == MODE_GATE, a, stream, [&] {
    hipLaunchKernelGGL((conv_b3ks_kernel<WN, MODE, NW>), grid, dim3(64 * NW), lds, stream, a);
  });
}

// Split-K is for launches whose one-role grid would leave most CUs with a single long reduction chain: few output
// tiles (at most two blocks per CU) and at least four channel chunks to share out.  Returns false when it does not apply.
// VITS_CONV_KS8=0 keeps the deepest reductions on four waves (for A/B measurements).
static bool ks8_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_CONV_KS8");
    return !(e && e[0] == '0');
  }();
  return on;
}

// Tile width (n-tiles of 32 steps: 1..3) the split-K kernel would take this conv with, 0 when it is not a split-K
// launch; *eight: eight waves share the reduction.  `halo` = max_off - min_off of the conv.
static int ks_choice(bool gate, int M, int Cin, int K, int L, int B, int halo, bool* eight) {
  *eight = false;
  // rows: whole 64-row units, or (plain convs) a last unit with one 32-row tile -- the 96-row projections of the flow
  if (!ks_enabled() || Cin < 64 || (gate ? M % 64 != 0 : M % 32 != 0)) return 0;
  const int n32 = (L + 31) / 32, mt_total = M / 32;
  const int units = gate ? mt_total / 2 : (mt_total + 1) / 2;
  const int cus = device_cu_count();
  // Tile width: the launch is latency-bound, so what counts is how many block "rounds" the chip needs (blocks of
  // width 1 and 2 fit two to a CU, width 3 one) times the length of a block's chain (a fixed part + the tile width).
  int wn = 0;
  long long best = 0, best_blocks = 0;
  for (int c = 1; c <= 3; ++c) {
    const long long blocks = (long long)((n32 + c - 1) / c) * units * B;
    const long long slots = (long long)cus * (c <= 2 ? 2 : 1);
    const long long cost = ((blocks + slots - 1) / slots) * (2 + c);
    if (wn == 0 || cost < best) { best = cost; wn = c; best_blocks = blocks; }
  }
  // beyond two blocks per CU the one-role kernel wins, except for deep reductions (>= 256 channels, >= 5 taps)
  if (best_blocks > 2LL * cus && !(Cin >= 256 && K >= 5)) return 0;
  // eight waves when the chip is nearly empty and the reduction is deep: at most one block for every fourth CU, >= 32 chunks
  *eight = wn == 1 && Cin >= 512 && best_blocks * 4 <= cus && (size_t)8 * 6 * (32 + halo) * sizeof(uint4) <= 64 * 1024 &&
           ks8_enabled();
  return wn;
}

bool conv_takes_split_k(const ConvDesc& d, int B, int L) {
  bool eight;
  if (d.w_alt && alt_prefers_f16x2(B, L)) return false;
  return d.prec == PREC_BF16X3 && (d.mode == MODE_PLAIN || d.mode == MODE_GATE) &&
         ks_choice(d.mode == MODE_GATE, d.M, d.Cin, d.K, L, B, d.max_off - d.min_off, &eight) > 0;
}

template <int MODE>
static bool try_launch_ks(ConvArgs& a, hipStream_t stream, hipError_t& err) {
  bool eight;
  const int wn = ks_choice(MODE == MODE_GATE, a.M, a.Cin, a.K, a.L, a.B, a.width, &eight);
  if (wn == 0) return false;
  if (eight) {
    err = launch_b3ks<1, MODE, 8>(a, stream);
    return true;
  }
  err = wn == 1 ? launch_b3ks<1, MODE>(a, stream) : wn == 2 ? launch_b3ks<2, MODE>(a, stream) : launch_b3ks<3, MODE>(a, stream);
  return true;
}

template <int WM, int WN, int WGM, int WGN, int MODE>
static hipError_t launch_cfg(ConvArgs& a, int prec, hipStream_t stream) {
  if (prec == PREC_BF16X3) return launch_b3<WM, WN, WGM, WGN, MODE>(a, stream);
  if (prec == PREC_BF16) return launch_b1<WM, WN, WGM, WGN, MODE>(a, stream);
  if (prec == PREC_F16X2) return launch_h2<WM, WN, WGM, WGN, MODE>(a, stream);
  // dwordx4 staging needs 16-byte aligned rows and whole groups of four time steps
  const bool vec = a.L % 4 == 0 && a.x_cs % 4 == 0 && a.x_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  if (vec) return launch_f32<WM, WN, WGM, WGN, MODE, true>(a, stream);
  return launch_f32<WM, WN, WGM, WGN, MODE, false>(a, stream);
}

// VITS_CONV_SMALL128: the largest grid, in quarter blocks per CU, that gets 128 x 64 tiles in place of the tall or
// wide ones (default 4 = one block per CU; 0: never).
static int small128_limit() {
  static const int n = [] {
    const char* e = getenv("VITS_CONV_SMALL128");
    return e && e[0] ? atoi(e) : 4;
  }();
  return n;
}

// When a bf16x3 conv that carries an f16x2 packing (ConvDesc::w_alt) takes it: 0 never, 1 (default) for launches too
// large for the split-K kernel, 2 always (tests and measurements: the split-K kernel is then bypassed for such convs).
// VITS_FE_F16X2 sets the initial value.
static std::atomic<int> g_alt_mode{[] {
  const char* e = getenv("VITS_FE_F16X2");
  return e && e[0] ? atoi(e) : 1;
}()};
void set_alt_f16x2_mode(int mode) { g_alt_mode.store(mode < 0 ? 0 : mode > 2 ? 2 : mode); }
int alt_f16x2_mode() { return g_alt_mode.load(); }
static thread_local bool g_fe_overlapped = false;
void set_front_end_overlapped(bool on) { g_fe_overlapped = on; }
bool front_end_overlapped() { return g_fe_overlapped; }

// A conv that has the f16x2 packing leaves the split-K kernel to it: always in mode 2; in mode 1 when the front end
// runs beside a generator (throughput mode), from VITS_FE_F16X2_COLS (default 2048) time steps over the whole batch on.
// The split-K kernel is built for latency (four waves share one output tile's reduction) and is the faster one on an
// otherwise idle chip (a plain call at 16 x 258 frames: 13.7 against 14.6 ms), but it spreads a conv over several
// hundred blocks that each hold a block slot while they wait on memory; the one-role kernel does the same conv in half
// the products and a fraction of the block time, which is what the generator running beside it gets back (C2 in
// throughput mode: +1 %).
bool alt_prefers_f16x2(int B, int L) {
  static const long long min_cols = [] {
    const char* e = getenv("VITS_FE_F16X2_COLS");
    return e && e[0] ? atoll(e) : 2048LL;
  }();
  const int mode = alt_f16x2_mode();
  return mode >= 2 || (mode == 1 && g_fe_overlapped && (long long)B * L >= min_cols);
}

hipError_t launch_conv(const ConvDesc& d, ConvArgs a, hipStream_t stream) {
  if (d.Cin % 16 != 0 || d.M % 32 != 0 || d.K < 1 || d.K > MAX_TAPS) return hipErrorInvalidValue;
  if (d.max_off - d.min_off > MAX_HALO) return hipErrorInvalidValue;
  if (a.B <= 0 || a.L <= 0) return hipErrorInvalidValue;
  if (a.B > 65535) return hipErrorInvalidValue;
  a.w = d.w;
  a.bias = d.bias;
  a.xcd_map = xcd_map_enabled();
  a.Cin = d.Cin;
  a.M = d.M;
  a.K = d.K;
  a.mt_total = d.M / 32;
  a.min_off = d.min_off;
  a.width = d.max_off - d.min_off;
  for (int c = 0; c < 2; ++c) {
    a.toff0[c] = d.toff0[c];
    a.tstep[c] = d.tstep[c];
  }
  const int L = a.L, M = d.M;
  int p = d.prec;
  if ((a.flags & (F_IN_STAGED | F_OUT_STAGED)) && (d.mode != MODE_PLAIN || (p != PREC_BF16 && p != PREC_F16X2)))
    return hipErrorInvalidValue;
  if ((a.flags & F_OUT_STAGED) && (a.flags & (F_RELU | F_MASK_PRE | F_RES | F_RES_SUB | F_ACCUM | F_MASK_POST)))
    return hipErrorInvalidValue;
  if ((a.flags & (F_IN_STAGED | F_OUT_STAGED)) && (d.Cin % 16 || M % 32)) return hipErrorInvalidValue;
  if (a.split_row && (d.mode != MODE_PLAIN || a.split_row % 64 || a.split_row >= M || !a.out_hi || a.out2 ||
                      (a.flags & (F_OUT_STAGED | F_IN_STAGED)) || ((a.flags | a.flags_hi) & F_RES_SUB)))
    return hipErrorInvalidValue;
  // a second, staged copy of the result: the staged-input f16x2 / bf16 kernels write it, and the x8 / x4 transposed convs
  const bool up84 = d.mode == MODE_UP8 || d.mode == MODE_UP4;
  if (a.out2 && ((p != PREC_F16X2 && p != PREC_BF16) || (a.flags & F_OUT_STAGED) ||
                 !(up84 || (d.mode == MODE_PLAIN && (a.flags & F_IN_STAGED)))))
    return hipErrorInvalidValue;
  if (a.split_row && p != PREC_F32 && p != PREC_BF16X3) return hipErrorInvalidValue;       // the kernels that run the flow
  // deferred LayerNorm: the bf16x3 kernels' 64-row wave tiles only; statistics are per 64-channel unit
  if (a.ln_in_stats || a.ln_res_stats || a.ln_out_stats) {
    if (d.mode != MODE_PLAIN || p != PREC_BF16X3 || M % 64 || a.split_row) return hipErrorInvalidValue;
    if (a.ln_in_stats && (d.Cin % 64 || !a.ln_in_gamma || !a.ln_in_beta)) return hipErrorInvalidValue;
    if (a.ln_res_stats && (!a.res || !a.ln_res_gamma || !a.ln_res_beta)) return hipErrorInvalidValue;
  }
  if (p == PREC_BF16X3 && (d.mode == MODE_PLAIN || d.mode == MODE_GATE)) {
    // the f16x2 packing, where the conv has one and uses nothing the f16x2 kernel's epilogue is not built with (a
    // second destination, deferred LayerNorm)
    // (a second destination: the 128-row tile shape only)
    static const bool split_too = [] {            // VITS_FE_SPLIT_F16X2=0: two-destination convs stay bf16x3 (A/B)
      const char* e = getenv("VITS_FE_SPLIT_F16X2");
      return !(e && e[0] == '0');
    }();
    const bool alt_ok = d.w_alt && !a.ln_in_stats && !a.ln_res_stats && !a.ln_out_stats && M % 64 == 0 &&
                        (!a.split_row || (split_too && M % 128 == 0 && d.mode == MODE_PLAIN));
    const int alt = alt_ok ? alt_f16x2_mode() : 0;
    if (!(alt_ok && alt_prefers_f16x2(a.B, a.L))) {
      hipError_t err = hipSuccess;
      ConvArgs ak = a;
      if (d.mode == MODE_PLAIN ? try_launch_ks<MODE_PLAIN>(ak, stream, err) : try_launch_ks<MODE_GATE>(ak, stream, err))
        return err;
    }
    if (alt > 0) {                               // too large for the split-K kernel (or forced)
      a.w = d.w_alt;
      p = PREC_F16X2;
    }
  }
  // A front-end conv on its f16x2 packing whose grid is a few hundred blocks whichever the tile (32 x 258 frames), on
  // an otherwise idle chip: what counts is the length of a block's chain.  128 x 64 tiles (two m-tiles x one n-tile a
  // wave: half the chain of the 64 x 256 and 256 x 64 tiles, and a row of 258 steps wastes a fifth of its last tile,
  // not half) while the grid still fits the chip's block slots; rows beyond M in the last 128-row block are computed
  // and not stored.  Plain calls at 16 x 258 with the packing forced: 15.08 -> 14.61 ms.  Beside a generator the generic
  // choice stays (fewer, larger blocks: +0.2 %).  VITS_FE_TILE64=0 keeps the generic choice everywhere (A/B).
  static const bool fe_tile64 = [] {
    const char* e = getenv("VITS_FE_TILE64");
    return !(e && e[0] == '0');
  }();
  const long long row_blocks64 = d.mode == MODE_GATE ? (M / 2 + 63) / 64 : (M + 127) / 128;
  const bool alt_small = fe_tile64 && !front_end_overlapped() && p == PREC_F16X2 && d.prec != PREC_F16X2 && !a.split_row && M % 64 == 0 &&
                         (long long)((L + 63) / 64) * row_blocks64 * a.B <= 2LL * device_cu_count();
  switch (d.mode) {
    case MODE_PLAIN:
      if (a.split_row && p == PREC_F16X2) return launch_cfg<2, 2, 2, 2, MODE_PLAIN>(a, p, stream);
      if (alt_small) return launch_cfg<2, 1, 2, 2, MODE_PLAIN>(a, p, stream);
      if (M % 64 == 0) {
        if (L <= 96 && M % 128 == 0) return launch_cfg<2, 1, 2, 2, MODE_PLAIN>(a, p, stream);
        // bf16x3 convs the split-K kernel declined (phoneme rate in a large batch: 64 x 160): the same 128 x 64 tiles
        // while their grid fits the block slots -- half the chain, and a row of 160 steps wastes a sixth of its tiles
        // instead of three eighths of one 256-step tile
        static const bool b3_tile64 = [] {
          const char* e = getenv("VITS_B3_TILE64");
          return !(e && e[0] == '0');
        }();
        if (b3_tile64 && p == PREC_BF16X3 && !a.split_row && L <= 512 &&
            (long long)((L + 63) / 64) * ((M + 127) / 128) * a.B <= 2LL * device_cu_count())
          return launch_cfg<2, 1, 2, 2, MODE_PLAIN>(a, p, stream);
        // a grid that leaves most CUs empty (batch 1): 128 x 64 tiles, twice the blocks, half the chain
        if (p == PREC_F16X2 && M % 128 == 0 && small128_limit() > 0 &&
            (long long)((L + 63) / 64) * (M / 128) * a.B * 4 <= (long long)small128_limit() * device_cu_count())
          return launch_cfg<2, 1, 2, 2, MODE_PLAIN>(a, p, stream);
        // taller block tiles when there are enough rows: one staged activation tile then feeds 128 / 256 rows
        if (M % 256 == 0) return launch_cfg<2, 2, 4, 1, MODE_PLAIN>(a, p, stream);
        if (M % 128 == 0) return launch_cfg<2, 2, 2, 2, MODE_PLAIN>(a, p, stream);
        return launch_cfg<2, 2, 1, 4, MODE_PLAIN>(a, p, stream);
      }
      if (L > 1024 && p == PREC_F32) return launch_cfg<1, 4, 1, 4, MODE_PLAIN>(a, p, stream);
      if (L > 192) return launch_cfg<1, 2, 1, 4, MODE_PLAIN>(a, p, stream);
      return launch_cfg<1, 1, 1, 4, MODE_PLAIN>(a, p, stream);
    case MODE_GATE:
      if (M % 64 != 0) return hipErrorInvalidValue;
      if (L <= 96 || alt_small) return launch_cfg<2, 1, 2, 2, MODE_GATE>(a, p, stream);
      return launch_cfg<2, 2, 1, 4, MODE_GATE>(a, p, stream);
    case MODE_UP8:
      if (M % 64 != 0) return hipErrorInvalidValue;
      // taller block tiles: one staged input tile feeds 256 rows (eight phases of 32 channels) instead of 64 or 128.
      // At 16 x 3 s: 512 -> 256 channels 142 -> 98 us, 256 -> 128 channels 215 -> 164 us; batch-1 latency unchanged.
      if (M % 256 == 0) return launch_cfg<2, 2, 4, 1, MODE_UP8>(a, p, stream);
      if (L <= 1024) return launch_cfg<2, 1, 2, 2, MODE_UP8>(a, p, stream);
      return launch_cfg<2, 2, 1, 4, MODE_UP8>(a, p, stream);
    case MODE_UP4:
      if (M % 64 != 0) return hipErrorInvalidValue;
      // 128 rows x 64 steps: one staged input tile feeds twice the rows (155 -> 140 us for 128 -> 64 channels at 16 x 3 s)
      if (M % 128 == 0) return launch_cfg<2, 2, 2, 2, MODE_UP4>(a, p, stream);
      return launch_cfg<2, 2, 1, 4, MODE_UP4>(a, p, stream);
    case MODE_UP2:
      if (M % 64 != 0) return hipErrorInvalidValue;
      return launch_cfg<2, 2, 1, 4, MODE_UP2>(a, p, stream);
    default:
      return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fused ResBlock pair for the 32-channel stage (PREC_F16X2): out = x + c2(lrelu(c1(lrelu(x)))) (+ out).
// The stage is bandwidth-bound, and the unfused pair moves five tensor passes (x in, t out, t in, x in again, out);
// here a block computes c1 on 256 time steps into LDS (leaky-relu'd, split into fp16 h / l, in c2's operand layout,
// zero outside the sequence as c2's padding requires) and c2 on the inner 224 of them, so x is read once (plus halo;
// the raw values of the block's own columns stay in registers and are the residual) and only the result is written.  Same accumulation order per output as the two separate kernels: bit-identical.
//   block = 4 waves side by side in time; c1: wave tile 32 rows x 64 steps; c2: n-tiles 2w, 2w + 1 of the 7 valid ones.
//   Y column c <-> time t0 - 16 + c;  c2 output column o <-> time t0 + o, reads Y columns o + 16 + k - (K-1)/2.
// ---------------------------------------------------------------------------------------------------------
constexpr int PAIR_YOFF = 16;

// MT = channels / 32: MT waves side by side in rows (one 32-row tile each), 4 / MT in time (64 steps each).
// Three blocks to a CU: a fourth (Y overlaid on the dead x tile, 128 registers) measured 4 to 11 % slower.
// K (taps) is a template parameter so that both tap loops unroll completely: the weight fragments of the next PAIR_AHEAD
// taps are then kept in flight in a register ring with compile-time slots.  A fragment load is an L2 hit, but vector
// memory returns in order within a CU, so behind the other blocks' streaming x loads it takes ~2000 cycles instead of
// ~560; fetched only one tap ahead of its MFMAs (192 cycles a tap) the matrix pipe sat idle most of the time.
constexpr int PAIR_AHEAD = 3;
// The two-row-tile build (64 channels, one plane) has two blocks to a CU instead of three to cover a fragment's round
// trip, and registers to spare: seven taps ahead (per launch at 64 x 10 s, ring of 3 / 5 / 7: k = 11 4.60 / 4.01 / 3.83 ms,
// k = 7 3.24 / 2.90 / 2.90).  The one-row-tile bf16 build gains nothing from a deeper ring (8: k = 7 6 % slower).
constexpr int PAIR_TALL_AHEAD = 7;

// NP: operand planes.  2: PREC_F16X2 (fp16 h / l, three products a tap).  1: PREC_BF16 (operands rounded to bf16, one
// product a tap; the same staging, the same order of additions as conv_bf16_kernel with a staged intermediate).
// WR: 32-row tiles per wave.  1: MT waves side by side in rows.  2 (64 channels, one plane): every wave owns all 64 rows
// of its 64 columns, so an activation fragment read from LDS feeds two MFMAs and a weight fragment two as well --
// with one product a tap the 32 x 64 wave tile asks LDS and the vector cache for a kilobyte per MFMA, which is all
// either delivers -- and the tile is 256 steps wide (224 outputs: a seventh of the first conv recomputed, not a third).
template <int MT, int K, int NP = 2, int WR = 1>
__global__ __launch_bounds__(256, WR == 2 ? 2 : 3) void resblock_pair_kernel(const PairArgs a, const uint4* __restrict__ w1,
                                                               const float* __restrict__ b1,
                                                               const uint4* __restrict__ w2,
                                                               const float* __restrict__ b2, int dil) {
  static_assert(MT % WR == 0, "whole waves in rows");
  constexpr int WGM = MT / WR, WGN = 4 / WGM, W1 = 64 * WGN, BN = W1 - 32, NCH = 2 * MT;
  constexpr int CPT = NCH * W1 / 256;                       // 16-channel chunks of its column a thread loads and keeps
  constexpr int NT = NCH * K, D = WR == 2 ? PAIR_TALL_AHEAD : PAIR_AHEAD;               // taps per conv; depth of the weight ring
  constexpr int OP = BN + 8;                                // row pitch of the result tile in LDS (floats)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int par = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, t0 = xcd_tile(blockIdx.x, gridDim.x, a.xcd_map) * BN;
  const int h1 = (K - 1) / 2 * dil, h2 = (K - 1) / 2;
  const int ncols = W1 + 2 * h1;                            // x tile: times t0 - 16 - h1 ... , one column per time step
  uint4* xq = lds;                                          // [NP planes][2 halves][ncols]: one 16-channel chunk at a time
  uint4* yq = lds + 2 * NP * ncols;                         // [NCH chunks][NP planes][2 halves][W1] (+ slack)
  const float* xb = a.x + (long long)b * a.x_bs;
  const int tx0 = t0 - PAIR_YOFF - h1;

  // The whole x tile is requested at once, before anything else: one memory round trip per block instead of one per
  // chunk plus one for the residual.
  //   * thread (c, cp) = (tid % W1, tid / W1) owns tile column c + h1 (time t0 - 16 + c: the W1 columns under the first
  //     conv's outputs, which include every column the block writes) for the chunks CPT cp .. CPT cp + CPT - 1: 16 CPT
  //     raw values that stay in registers to the end -- they are the residual of the block's own outputs, so x is
  //     never read twice;
  //   * the 2 * h1 halo columns on either side: thread tid < 2 * h1 * NCH takes (column, chunk) item tid.
  const int oc = tid % W1, ocp = tid / W1;
  const int ot = t0 - PAIR_YOFF + oc;                       // this thread's own column: its time step
  const bool o_ok = ot >= 0 && ot < a.L;
  const int nh = 2 * h1 * NCH;                              // halo items
  const int hidx = tid / NCH, hch = tid % NCH;              // this thread's halo item: column index, chunk
  const int hcol = hidx < h1 ? hidx : W1 + hidx;            // tile column: left halo [0, h1), right halo [W1 + h1, ncols)
  const int ht = tx0 + hcol;
  const bool h_has = tid < nh, h_ok = h_has && ht >= 0 && ht < a.L;
  float st[CPT][16], sb[16];
  {
    const float* xo = xb + (long long)(ocp * 16 * CPT) * a.x_cs + (o_ok ? ot : 0);
#pragma unroll
    for (int u = 0; u < CPT; ++u)
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) st[u][ci] = xo[(long long)(u * 16 + ci) * a.x_cs];
    const float* xh = xb + (long long)(hch * 16) * a.x_cs + (h_ok ? ht : 0);
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) sb[ci] = h_has ? xh[(long long)ci * a.x_cs] : 0.f;
  }
  // 16 channels of one tile column -> leaky-relu, clamp, split into fp16 h / l, four 16-byte units of xq
  auto put = [&](const float (&v16)[16], int col, bool ok) __attribute__((always_inline)) {
    unsigned ph[8], pl[8];
    float amax = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2) {
      f32x2 v;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float e = v16[2 * c2 + u];
        e = fmaxf(e, e * a.slope);
        if constexpr (NP == 2) {
          amax = fmaxf(amax, fabsf(e));
          e = fminf(fmaxf(e, -65504.f), 65504.f);
        }
        v[u] = ok ? e : 0.f;
      }
      if constexpr (NP == 2) {
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
        ph[c2] = __builtin_bit_cast(unsigned, h);
        pl[c2] = __builtin_bit_cast(unsigned, l);
      } else {
        ph[c2] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
      }
    }
    if constexpr (NP == 2) {
      if (ok) report_f16_range(a.range_flag, amax);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      xq[(0 * 2 + hf) * ncols + col] = make_uint4(ph[4 * hf], ph[4 * hf + 1], ph[4 * hf + 2], ph[4 * hf + 3]);
      if constexpr (NP == 2)
        xq[(1 * 2 + hf) * ncols + col] = make_uint4(pl[4 * hf], pl[4 * hf + 1], pl[4 * hf + 2], pl[4 * hf + 3]);
    }
  };
  auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
    if (chunk / CPT == ocp) put(st[chunk % CPT], oc + h1, o_ok);       // chunk is a compile-time constant
    if (h_has && hch == chunk) put(sb, hcol, h_ok);
  };

  f32x16 acc[WR][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // one product of a tap: plane pa of the weights (every row tile of the wave) x the given activation fragments
  auto mma = [&](const uint4 (&af)[WR][NP], int pa, const uint4 (&bf)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (NP == 2)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][pa]),
                                                             __builtin_bit_cast(f16x8, bf[j]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][pa]),
                                                              __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
      }
  };

  // weight ring: tap g of the virtual sequence [first conv's NT taps, second conv's NT taps] lives in slot g % D
  uint4 Aq[D][WR][NP];
  auto fetch_a = [&](int g) __attribute__((always_inline)) {
    if (g < 2 * NT) {
      const uint4* wk = (g < NT ? w1 : w2) + ((long long)(g < NT ? g : g - NT) * MT + wm * WR) * (64 * NP) + lane;
#pragma unroll
      for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) Aq[g % D][i][p] = wk[64 * (i * NP + p)];
    }
  };
#pragma unroll
  for (int g = 0; g < D; ++g) fetch_a(g);

  // ---- first conv: rows 32 * wm .., Y columns [64 * wn, 64 * wn + 64) ----
  zero_acc();
  {
    const int colbase = wn * 64 + l31 + par * ncols;
    auto taps = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int g = chunk * K + k;
        const int off = colbase + k * dil;
        uint4 Bh[2], Bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          Bh[j] = xq[off + j * 32];
          if constexpr (NP == 2) Bl[j] = xq[2 * ncols + off + j * 32];
        }
        if constexpr (NP == 2) {
          mma(Aq[g % D], 0, Bl);                           // h * l
          mma(Aq[g % D], 0, Bh);                           // h * h
          mma(Aq[g % D], NP - 1, Bh);                      // l * h
        } else {
          mma(Aq[g % D], 0, Bh);
        }
        fetch_a(g + D);
      }
    };
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      if (chunk > 0) __syncthreads();                      // every wave has read the previous chunk
      store_chunk(chunk);
      __syncthreads();
      taps(chunk);
    }
  }
  // ---- Y = split(lrelu(c1 + bias)), zero outside the sequence; 8-byte pieces (four channels of a half) ----
  {
    uint2* y2 = reinterpret_cast<uint2*>(yq);
    float ymax = 0.f;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
    const int rt = wm * WR + i;                              // this accumulator's 32-row tile
    float bz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bz[r] = b1 ? b1[32 * rt + 4 * par + (r & 3) + 8 * (r >> 2)] : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = wn * 64 + j * 32 + l31;
      const int t = t0 - PAIR_YOFF + c;
      const bool inside = t >= 0 && t < a.L;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 oh, ol;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x2 e;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float v = acc[i][j][4 * g + 2 * u + q] + bz[4 * g + 2 * u + q];
            v = fmaxf(v, v * a.slope);
            if constexpr (NP == 2) {
              ymax = fmaxf(ymax, fabsf(v));
              v = fminf(fmaxf(v, -65504.f), 65504.f);
            }
            e[q] = inside ? v : 0.f;
          }
          if constexpr (NP == 2) {
            const f16x2 h = __builtin_convertvector(e, f16x2);
            const f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, f32x2), f16x2);
            (u ? oh.y : oh.x) = __builtin_bit_cast(unsigned, h);
            (u ? ol.y : ol.x) = __builtin_bit_cast(unsigned, l);
          } else {
            (u ? oh.y : oh.x) = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2));
          }
        }
        const int chunk2 = 2 * rt + (g >> 1), half = g & 1;
        y2[(((chunk2 * NP + 0) * 2 + half) * W1 + c) * 2 + par] = oh;
        if constexpr (NP == 2) y2[(((chunk2 * NP + 1) * 2 + half) * W1 + c) * 2 + par] = ol;
      }
    }
    }
    if constexpr (NP == 2) report_f16_range(a.range_flag, ymax);
  }
  __syncthreads();
  // ---- second conv on the inner BN steps ----
  zero_acc();
  {
    const int obase = wn * 64 + l31 + PAIR_YOFF - h2 + par * W1;
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int g = NT + chunk * K + k;
        const int off = (chunk * 2 * NP) * W1 + obase + k;
        uint4 Bh[2], Bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          Bh[j] = yq[off + j * 32];
          if constexpr (NP == 2) Bl[j] = yq[2 * W1 + off + j * 32];
        }
        if constexpr (NP == 2) {
          mma(Aq[g % D], 0, Bl);
          mma(Aq[g % D], 0, Bh);
          mma(Aq[g % D], NP - 1, Bh);
        } else {
          mma(Aq[g % D], 0, Bh);
        }
        fetch_a(g + D);
      }
  }
  // ---- out = c2 + bias + x (+ out) ----
  // The result goes through LDS (the Y tile is dead once every wave has finished the second conv) into the layout
  // the x tile was loaded in: thread (c, cp) takes column o = c - 16 of its 32 rows, adds the raw x it has kept since
  // the load, and stores runs of 64 consecutive time steps per row.  Same operations in the same order as an epilogue
  // in the accumulator layout: c2 + bias, + x, + out.
  {
    float* ob = a.out + (long long)b * a.out_bs;
    const int o = oc - PAIR_YOFF, t = t0 + o;
    const bool mine = o >= 0 && o < BN && t < a.L;
    constexpr int NR = 16 * CPT;                           // rows this thread finishes: those it loaded
    constexpr bool EARLY = CPT <= 2;                       // room for the accumulated output beside everything else
    float ov[EARLY ? NR : 16];
    if (EARLY && a.accumulate) {                           // requested before the barriers below
      const int tc = mine ? t : 0;
#pragma unroll
      for (int i = 0; i < (EARLY ? NR : 0); ++i) ov[i] = ob[(long long)(NR * ocp + i) * a.out_cs + tc];
    }
    __syncthreads();                                       // every wave has read its last Y fragment
    float* oq = reinterpret_cast<float*>(yq);              // [32 * MT rows][OP]
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int rt = wm * WR + i;
      float bz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bz[r] = b2 ? b2[32 * rt + 4 * par + (r & 3) + 8 * (r >> 2)] : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + l31;
        if (col < BN) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            oq[(32 * rt + 4 * par + (r & 3) + 8 * (r >> 2)) * OP + col] = acc[i][j][r] + bz[r];
        }
      }
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        if constexpr (!EARLY) {
          if (a.accumulate) {
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) ov[ci] = ob[(long long)(NR * ocp + 16 * u + ci) * a.out_cs + t];
          }
        }
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) {
          const int i = 16 * u + ci;
          float v = oq[(NR * ocp + i) * OP + o];
          v += st[u][ci];
          if (a.accumulate) v += ov[EARLY ? i : ci];
          ob[(long long)(NR * ocp + i) * a.out_cs + t] = v;
        }
      }
    }
  }
}

// Where the fused pair is the faster choice (measured at 16 x 3 s, per launch): every 32-channel pair (k = 3: 190 us
// against 363 for the two launches, k = 7: 277 / 372, k = 11: 363 / 442).  At 64 channels the tile is 96 steps wide (the
// intermediate has to fit in LDS three blocks to a CU), and the halo re-reads and the wider first conv eat the saved
// traffic: k = 3 wins (300 / 346), k = 7 and 11 lose (557 and 806 us against ~590 for both).  VITS_PAIR64 (a list of
// tap counts, default "3"; empty for none) chooses the 64-channel pairs that are fused.
static bool pair64_taps(int K) {
  static const std::string v = getenv("VITS_PAIR64") ? getenv("VITS_PAIR64") : "3";
  return ("," + v + ",").find("," + std::to_string(K) + ",") != std::string::npos;
}

bool pair_profitable(const ConvDesc& c1, const ConvDesc& c2) {
  // plain bf16: one product a tap, so the recomputed halo of the 96-step tiles costs little and every layer is bound
  // by its traffic, which the fused pair halves: every 32- and 64-channel pair
  if (pair_fusable(c1, c2) && c1.prec == PREC_BF16) {
    static const std::string v = getenv("VITS_PAIR64_BF16") ? getenv("VITS_PAIR64_BF16") : "3,5,7,9,11";
    return c1.M == 32 || ("," + v + ",").find("," + std::to_string(c1.K) + ",") != std::string::npos;
  }
  return pair_fusable(c1, c2) && (c1.M == 32 || pair64_taps(c1.K));
}

// everything a fused pair needs but the channel count (which differs between the pair and the whole-ResBlock kernel)
bool pair_shape_fusable(const ConvDesc& c1, const ConvDesc& c2) {
  return (c1.prec == PREC_F16X2 || c1.prec == PREC_BF16) && c2.prec == c1.prec && c1.mode == MODE_PLAIN && c2.mode == MODE_PLAIN &&
         c1.M == c1.Cin && c2.M == c1.M && c2.Cin == c1.M && c1.K == c2.K && (c1.K & 1) && c2.tstep[0] == 1 &&
         (c1.K - 1) / 2 * c1.tstep[0] <= 25 && (c2.K - 1) / 2 <= PAIR_YOFF && c1.K >= 3 && c1.K <= 11;
}

bool pair_fusable(const ConvDesc& c1, const ConvDesc& c2) {
  return pair_shape_fusable(c1, c2) && (c1.M == 32 || c1.M == 64);
}

hipError_t launch_resblock_pair(const ConvDesc& c1, const ConvDesc& c2, const PairArgs& a, hipStream_t stream) {
  if (!pair_fusable(c1, c2) || a.B <= 0 || a.L <= 0 || a.B > 65535) return hipErrorInvalidValue;
  const int K = c1.K, dil = c1.tstep[0], MT = c1.M / 32;
  const int NP = c1.prec == PREC_BF16 ? 1 : 2;
  // 64 channels in one plane: every wave takes both row tiles (a 256-step tile, two blocks to a CU).  Measured at
  // 64 x 10 s per launch with the chains one after the other, against 96-step tiles at three blocks to a CU: k = 3 1.96
  // against 2.17 ms, k = 7 3.20 / 3.34, k = 11 4.57 / 4.34; with the chains side by side, as the generator runs them,
  // the whole step is shortest with the wide tile at every tap count (VITS_PAIR_BF16_TALL=<taps>: the largest tap
  // count that takes it; 0 = never; measured 0 / 7 / 11: 6245 / 6266 / 6405 audio-s/s).
  static const int tall_taps = getenv("VITS_PAIR_BF16_TALL") ? atoi(getenv("VITS_PAIR_BF16_TALL")) : 11;
  const int WR = (NP == 1 && MT == 2 && K <= tall_taps) ? 2 : 1;
  const int W1 = 64 * (4 / (MT / WR)), BN = W1 - 32, nch = 2 * MT;
  const int ncols = W1 + 2 * ((K - 1) / 2 * dil);
  // operand tiles (+ slack for the invalid n-tile); the result tile ([32 * MT rows][BN + 8] fp32) reuses the Y tile's
  // space and, with one plane, needs more than that
  size_t lds = ((size_t)2 * NP * ncols + (size_t)nch * 2 * NP * W1 + 64) * sizeof(uint4);
  lds = std::max(lds, (size_t)2 * NP * ncols * sizeof(uint4) + (size_t)32 * MT * (BN + 8) * sizeof(float));
  dim3 grid((a.L + BN - 1) / BN, 1, a.B);
  ConvArgs ta;
  memset(&ta, 0, sizeof ta);
  ta.M = c1.M; ta.Cin = c1.M; ta.K = 2 * K; ta.L = a.L; ta.B = a.B;      // timing row: both convs' FLOPs
  ta.flags = a.accumulate ? F_ACCUM : 0;       // bytes: x once (the residual is the same tensor), the result, the sum
  char name[64];
  snprintf(name, sizeof name, NP == 1 ? "resblock_pair_bf16_kernel<%d,%d>" : "resblock_pair_kernel<%d,%d>", MT, K);
  const uint4* w1 = reinterpret_cast<const uint4*>(c1.w);
  const uint4* w2 = reinterpret_cast<const uint4*>(c2.w);
  PairArgs pa = a;
  pa.xcd_map = xcd_map_enabled();
  auto go = [&](auto mt, auto kk) {
    if (NP == 1 && WR == 2) {
      if constexpr (decltype(mt)::value == 2)
        hipLaunchKernelGGL((resblock_pair_kernel<2, decltype(kk)::value, 1, 2>), grid, dim3(256), lds, stream,
                           pa, w1, c1.bias, w2, c2.bias, dil);
    } else if (NP == 1)
      hipLaunchKernelGGL((resblock_pair_kernel<decltype(mt)::value, decltype(kk)::value, 1>), grid, dim3(256), lds, stream,
                         pa, w1, c1.bias, w2, c2.bias, dil);
    else
      hipLaunchKernelGGL((resblock_pair_kernel<decltype(mt)::value, decltype(kk)::value, 2>), grid, dim3(256), lds, stream,
                         pa, w1, c1.bias, w2, c2.bias, dil);
  };
  auto by_taps = [&](auto mt) {
    switch (K) {
      case 3: go(mt, IntC<3>{}); break;
      case 5: go(mt, IntC<5>{}); break;
      case 7: go(mt, IntC<7>{}); break;
      case 9: go(mt, IntC<9>{}); break;
      default: go(mt, IntC<11>{}); break;
    }
  };
  return timed_launch(name, false, ta, stream, [&] {
    if (MT == 1) by_taps(IntC<1>{});
    else by_taps(IntC<2>{});
  });
}

}  // namespace vits
