// Bandwidth-bound and small kernels of the MVTS synthesis path (gfx950, wave64): channel-first LayerNorm,
// windowed relative-position attention, length regulator (prefix sum + gather), embeddings, conditioning GEMV,
// 1-channel convs and the scalar control math.  Every kernel reads and writes with lanes on the time axis, the
// contiguous axis of the [B, C, L] layout, so global accesses are coalesced.
#include "common.h"
#include <math.h>
#include <cstdlib>

namespace vits {

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over channels of [B, C, L]  (reference modules.py:15-27; nn.LayerNorm in frame_prior_network.py:83,95)
// biased variance, eps inside the sqrt.
// Block = 16 time steps x 16 channel groups: thread (t, g) keeps channels g, g + 16, ... (NV of them) in registers,
// so x is read once, every load of a thread is independent and in flight together, and the two reductions (mean,
// then the centred sum of squares) run over registers: shuffles across the four groups of a wave, LDS across waves.
// ---------------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* x, float* y,   // x may be y (in place)
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const int* __restrict__ lengths, int C, int L, int mask_out) {
  __shared__ float red[2][4][16];
  const int tl = threadIdx.x & 15, g = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 16 + tl;
  const bool valid = t < L;
  const float* xb = x + (long long)b * C * L + (valid ? t : 0);
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = g + 16 * i;
    v[i] = c < C ? xb[(long long)c * L] : 0.f;
  }
  auto block_sum = [&](float s, int slot) {
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if ((threadIdx.x & 63) < 16) red[slot][wave][tl] = s;
    __syncthreads();
    return red[slot][0][tl] + red[slot][1][tl] + red[slot][2][tl] + red[slot][3][tl];
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  const float mean = block_sum(s, 0) / (float)C;
  s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = g + 16 * i < C ? v[i] - mean : 0.f;
    s += d * d;
  }
  const float var = block_sum(s, 1) / (float)C;
  const float rstd = 1.f / sqrtf(var + 1e-5f);
  const bool keep = !mask_out || !lengths || t < lengths[b];
  if (valid) {
    float* yb = y + (long long)b * C * L + t;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = g + 16 * i;
      if (c < C) {
        const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
        yb[(long long)c * L] = keep ? o : 0.f;
      }
    }
  }
}

hipError_t launch_layer_norm(const float* x, float* y, const float* gamma, const float* beta, const int* lengths,
                             int B, int C, int L, bool mask_out, hipStream_t s) {
  if (B <= 0 || C <= 0 || C > 1024 || L <= 0 || B > 65535) return hipErrorInvalidValue;
  dim3 grid((L + 15) / 16, B);
  const int m = mask_out ? 1 : 0;
#define LN_LAUNCH(NV) hipLaunchKernelGGL(layer_norm_kernel<NV>, grid, dim3(256), 0, s, x, y, gamma, beta, lengths, C, L, m)
  if (C <= 192) LN_LAUNCH(12);
  else if (C <= 256) LN_LAUNCH(16);
  else if (C <= 768) LN_LAUNCH(48);
  else LN_LAUNCH(64);
#undef LN_LAUNCH
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Self-attention with windowed relative-position keys/values (reference attentions.py:226-257, banded form):
//   s_ij = (q_i/sqrt(dk)).k_j + [|j-i|<=W] (q_i/sqrt(dk)).Ek[j-i+W];   masked (i or j beyond the length) -> -1e4
//   p = softmax_j(s);   o_i = sum_j p_ij v_j + sum_{|j-i|<=W} p_ij Ev[j-i+W]
// qkv is [B, 3*H*DK, L] (q rows, then k rows, then v rows).  A block handles 4 * QPW queries of one (batch, head),
// QPW per wave.  K and V go through LDS in chunks of 64 keys (coalesced loads, everything in flight at once):
//   scores:  lanes = keys;     q broadcast from LDS, the whole score row stays in registers (NCH chunks)
//   p . v:   lanes = channels; p_ij of a key is read from its lane with v_readlane, v from LDS (row stride 65)
// ---------------------------------------------------------------------------------------------------------
constexpr int ATT_DK = 96;
constexpr int ATT_W = 4;
typedef float att_f32x2 __attribute__((ext_vector_type(2)));

template <int NCH, int QPW>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ ek,
                                                         const float* __restrict__ ev,
                                                         const int* __restrict__ lengths, float* __restrict__ out,
                                                         int H, int L) {
  constexpr int DK = ATT_DK, W = ATT_W, NR = 2 * ATT_W + 1, NQ = 4 * QPW, KS = 65;
  extern __shared__ __attribute__((aligned(16))) float att_s[];
  float* kv_s = att_s;                       // [DK][KS]   one 64-key chunk of K or V
  float* q_s = kv_s + DK * KS;               // [DK][NQ]   (16-byte aligned: DK * KS is a multiple of 4)
  float* qe_s = q_s + DK * NQ;               // [NQ][NR]
  float* pw_s = qe_s + NQ * NR;              // [NQ][NR]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * NQ, h = blockIdx.y, b = blockIdx.z;
  const int C = H * DK;
  const int len = lengths ? lengths[b] : L;
  const float* qb = qkv + ((long long)b * 3 * C + h * DK) * L;
  const float* kb = qb + (long long)C * L;
  const float* vb = kb + (long long)C * L;
  const float inv = sqrtf((float)DK);

  {                                                  // all loads of the staging pass are issued before the first LDS write
    constexpr int NU = DK * NQ / 256;
    float tmp[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + 256 * u, c = idx / NQ, i = i0 + idx % NQ;
      tmp[u] = qb[(long long)c * L + (i < L ? i : L - 1)];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + 256 * u, i = i0 + idx % NQ;
      q_s[idx] = i < L ? tmp[u] / inv : 0.f;
    }
  }
  for (int idx = tid; idx < NQ * NR; idx += 256) pw_s[idx] = 0.f;
  for (int idx = tid; idx < NR * DK; idx += 256) kv_s[idx] = ek[idx];      // relative key embeddings, via LDS
  __syncthreads();
  for (int idx = tid; idx < NQ * NR; idx += 256) {
    const int qi = idx / NR, r = idx % NR;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < DK; ++c) s += q_s[c * NQ + qi] * kv_s[r * DK + c];
    qe_s[idx] = s;
  }

  auto stage = [&](const float* src, int ch) __attribute__((always_inline)) {   // kv_s[c][j] = src[c][64*ch + j]
    constexpr int NU = DK * 64 / 256;
    float tmp[NU];
    const int j = tid & 63, jj = ch * 64 + j;
    const float* sp = src + (jj < L ? jj : L - 1);
#pragma unroll
    for (int u = 0; u < NU; ++u) tmp[u] = sp[(long long)(4 * u + (tid >> 6)) * L];
#pragma unroll
    for (int u = 0; u < NU; ++u) kv_s[(4 * u + (tid >> 6)) * KS + j] = jj < L ? tmp[u] : 0.f;
  };

  const int qw = wave * QPW;
  float sc[QPW][NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();                                   // previous chunk consumed (first pass: qe_s written)
    if (ch * 64 < L) stage(kb, ch);
    __syncthreads();
    const int j = ch * 64 + lane;
    att_f32x2 acc[QPW / 2];                            // pairs of queries: packed fp32 FMAs
#pragma unroll
    for (int n = 0; n < QPW / 2; ++n) acc[n] = att_f32x2{0.f, 0.f};
    if (ch * 64 < L) {
#pragma unroll 4
      for (int c = 0; c < DK; ++c) {
        const float kv = kv_s[c * KS + lane];
        const att_f32x2 kv2 = {kv, kv};
#pragma unroll
        for (int m = 0; m < QPW / 4; ++m) {
          const float4 q4 = *reinterpret_cast<const float4*>(&q_s[c * NQ + qw + 4 * m]);
          acc[2 * m + 0] += att_f32x2{q4.x, q4.y} * kv2;
          acc[2 * m + 1] += att_f32x2{q4.z, q4.w} * kv2;
        }
      }
    }
#pragma unroll
    for (int n = 0; n < QPW; ++n) {
      const int i = i0 + qw + n;
      const int r = j - i + W;
      float v = acc[n >> 1][n & 1];
      if (r >= 0 && r < NR) v += qe_s[(qw + n) * NR + r];
      if (!(i < len && j < len)) v = -1e4f;
      sc[n][ch] = j < L ? v : -INFINITY;
    }
  }
  // softmax over the row (all chunks, all lanes)
#pragma unroll
  for (int n = 0; n < QPW; ++n) {
    float m = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) m = fmaxf(m, sc[n][ch]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const float e = expf(sc[n][ch] - m);     // exp(-inf) = 0 for keys beyond L
      sc[n][ch] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float rinv = 1.f / sum;
    const int i = i0 + qw + n;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const float p = sc[n][ch] * rinv;
      sc[n][ch] = p;
      const int j = ch * 64 + lane, r = j - i + W;
      if (r >= 0 && r < NR && j < L) pw_s[(qw + n) * NR + r] = p;
    }
  }
  // o = p . v with lanes on channels: lane handles channel `lane` and, for lane < 32, channel 64 + lane
  att_f32x2 o01[QPW];                                  // (channel lane, channel 64 + lane) per query
#pragma unroll
  for (int n = 0; n < QPW; ++n) o01[n] = att_f32x2{0.f, 0.f};
  const int c1 = lane < DK - 64 ? 64 + lane : lane;     // lanes >= 32 repeat their first channel (result unused)
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    __syncthreads();
    if (ch * 64 < L) stage(vb, ch);
    __syncthreads();
    if (ch * 64 < L) {
      const int nj = L - ch * 64 < 64 ? L - ch * 64 : 64;
      for (int j = 0; j < nj; ++j) {
        const att_f32x2 v01 = {kv_s[lane * KS + j], kv_s[c1 * KS + j]};
#pragma unroll
        for (int n = 0; n < QPW; ++n) {
          const float p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc[n][ch]), j));
          o01[n] += att_f32x2{p, p} * v01;
        }
      }
    }
  }
  float* ob = out + ((long long)b * C + h * DK) * L;
#pragma unroll
  for (int slot = 0; slot < 2; ++slot) {
    if (slot == 1 && lane >= DK - 64) break;
    const int c = slot == 0 ? lane : 64 + lane;
    float evr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) evr[r] = ev[r * DK + c];
#pragma unroll
    for (int n = 0; n < QPW; ++n) {
      const int i = i0 + qw + n;
      if (i < L) {
        float v = o01[n][slot];
#pragma unroll
        for (int r = 0; r < NR; ++r) v += pw_s[(qw + n) * NR + r] * evr[r];
        ob[(long long)c * L + i] = v;
      }
    }
  }
}

template <int NCH, int QPW>
static void attention_launch(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out,
                             int B, int H, int L, hipStream_t s) {
  constexpr int NQ = 4 * QPW;
  dim3 grid((L + NQ - 1) / NQ, H, B);
  const size_t lds = (size_t)(ATT_DK * 65 + ATT_DK * NQ + 2 * NQ * (2 * ATT_W + 1)) * sizeof(float);
  hipLaunchKernelGGL((attention_kernel<NCH, QPW>), grid, dim3(256), lds, s, qkv, emb_k, emb_v, lengths, out, H, L);
}

// ---------------------------------------------------------------------------------------------------------
// The same attention on the matrix cores, for long sequences (L >= 256).
//   S'[j][i] = k_j . q_i  (keys x queries: the transposed score tile) on v_mfma_f32_32x32x16 with every fp32 operand
//   split into NP terms:  NP = 3: three bf16 terms (exact) and the six largest cross products, as the bf16x3 convs;
//                         NP = 2: fp16 h + l (22 bits, clamped to the fp16 range) and three products, as the f16x2 convs;
//   O[c][i]  = sum_j v[c][j] p[j][i]: the 32 x 32 accumulator layout of S' (a lane holds one query column and, per
//   register, keys 8g + 4*par + q) is used as the B operand of this product as it stands -- the sum over keys does not
//   care in which order its terms sit in the MFMA's k slots, so V is staged in that same key order instead of the
//   probabilities being shuffled across lanes.
// A block of NW waves owns 32 * NW queries of one (batch, head); keys and values pass through LDS in tiles of 64,
// split into planes once per block.  One pass over the keys with a running row maximum: a lane's registers of O all
// belong to its own query, so when a tile raises the maximum the partial sums are rescaled by one factor per lane
// (exp(old - new)) -- the scores are computed once.  The +-4 relative-position terms: q . Ek through one more MFMA tile
// (the nine embeddings as nine "keys"), added to the scores inside the band; the band's scores are kept per query,
// turned into probabilities against the final maximum and multiplied with Ev as one last k step of the p . v product.
// ---------------------------------------------------------------------------------------------------------
typedef float att_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 att_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 att_f16x2 __attribute__((ext_vector_type(2)));

// (v0, v1) -> packed bf16 pairs h, m, l with v = h + m + l exactly (low half = v0)
__device__ __forceinline__ void att_split3(float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
  const att_f32x2 v = {v0, v1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, att_bf16x2));
  const att_f32x2 r1 = {v0 - __builtin_bit_cast(float, h << 16), v1 - __builtin_bit_cast(float, h & 0xffff0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, att_bf16x2));
  const att_f32x2 r2 = {r1.x - __builtin_bit_cast(float, m << 16), r1.y - __builtin_bit_cast(float, m & 0xffff0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, att_bf16x2));
}

// eight fp32 values -> one 16-byte unit per plane; NP = 2: amax collects the largest magnitude seen (range report)
template <int NP>
__device__ __forceinline__ void att_split_unit(const float (&v)[8], uint4 (&pl)[NP], float& amax) {
  unsigned w[NP][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (NP == 3) {
      att_split3(v[2 * e], v[2 * e + 1], w[0][e], w[1][e], w[2][e]);
    } else {
      amax = fmaxf(amax, fmaxf(fabsf(v[2 * e]), fabsf(v[2 * e + 1])));
      const att_f32x2 c = {fminf(fmaxf(v[2 * e], -65504.f), 65504.f), fminf(fmaxf(v[2 * e + 1], -65504.f), 65504.f)};
      const att_f16x2 h = __builtin_convertvector(c, att_f16x2);
      const att_f16x2 l = __builtin_convertvector(c - __builtin_convertvector(h, att_f32x2), att_f16x2);
      w[0][e] = __builtin_bit_cast(unsigned, h);
      w[1][e] = __builtin_bit_cast(unsigned, l);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) pl[p] = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
}

// the cross products of two split operands, smallest first.  Three planes: a_h b_l, a_l b_h, a_m b_m, a_h b_m, a_m b_h,
// a_h b_h; two planes: a_h b_l, a_l b_h, a_h b_h
template <int NP>
__device__ __forceinline__ att_f32x16 att_mma_split(const uint4 (&a)[NP], const uint4 (&b)[NP], att_f32x16 c) {
  if constexpr (NP == 3) {
    auto mm = [&](const uint4& x, const uint4& y) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf16x8, x), __builtin_bit_cast(att_bf16x8, y), c, 0, 0, 0);
    };
    mm(a[0], b[2]);
    mm(a[2], b[0]);
    mm(a[1], b[1]);
    mm(a[0], b[1]);
    mm(a[1], b[0]);
    mm(a[0], b[0]);
  } else {
    auto mm = [&](const uint4& x, const uint4& y) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, x), __builtin_bit_cast(att_f16x8, y), c, 0, 0, 0);
    };
    mm(a[0], b[1]);
    mm(a[1], b[0]);
    mm(a[0], b[0]);
  }
  return c;
}

constexpr int ATT_KT = 64;                         // keys per LDS tile
constexpr int ATT_NCG = ATT_DK / 8;                // 16-byte channel groups per head (12)

template <int NW, int NP>
__global__ __launch_bounds__(64 * NW, 2) void attention_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ ek,
                                                                  const float* __restrict__ ev,
                                                                  const int* __restrict__ lengths, float* __restrict__ out,
                                                                  int H, int L, int* range_flag) {
  constexpr int DK = ATT_DK, W = ATT_W, NR = 2 * ATT_W + 1, NQ = 32 * NW, NT = 64 * NW, NCG = ATT_NCG;
  constexpr int KQ_UNITS = NP * NCG * ATT_KT;      // K tile: [NP planes][12 groups][64 keys]
  constexpr int VQ_UNITS = NP * DK * 8;            // V tile: [NP planes][96 channels][4 k steps x 2 halves], keys permuted
  static_assert(NP * NCG * NQ <= KQ_UNITS + VQ_UNITS, "the Q staging area fits the K / V tile space");
  extern __shared__ __attribute__((aligned(16))) uint4 att_m[];
  uint4* kq = att_m;
  uint4* vq = att_m + KQ_UNITS;
  uint4* evq = vq + VQ_UNITS;                      // Ev as one k step of V: [NP planes][96 channels][2 halves]
  float* qe_s = reinterpret_cast<float*>(evq + NP * DK * 2);    // [NQ][NR]   q . Ek
  float* pw_s = qe_s + NQ * NR;                    // [NQ][NR]   band scores (-inf: no such key)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = lane >> 5, l31 = lane & 31;
  const int i0 = blockIdx.x * NQ, h = blockIdx.y, b = blockIdx.z;
  const int C = H * DK;
  const int len = lengths ? lengths[b] : L;
  const float* qb = qkv + ((long long)b * 3 * C + h * DK) * L;
  const float* kb = qb + (long long)C * L;
  const float* vb = kb + (long long)C * L;
  const float qscale = 1.f / sqrtf((float)DK);
  const int il = 32 * wave + l31;                  // this lane's query inside the block
  const int iq = i0 + il;
  float amax = 0.f;                                // NP = 2: the largest operand magnitude this thread converted

  // ---- Q: staged (scaled, split) through the K / V tile space, then held in registers as B fragments ----
  for (int u = tid; u < NCG * NQ; u += NT) {
    const int cg = u / NQ, i = u % NQ;
    const int ii = i0 + i < L ? i0 + i : L - 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = qb[(long long)(8 * cg + e) * L + ii] * qscale;
    uint4 pl[NP];
    att_split_unit<NP>(v, pl, amax);
#pragma unroll
    for (int p = 0; p < NP; ++p) att_m[(p * NCG + cg) * NQ + i] = pl[p];
  }
  for (int idx = tid; idx < NQ * NR; idx += NT) pw_s[idx] = -INFINITY;
  __syncthreads();
  uint4 Qf[6][NP];                                 // [k step of 16 channels][plane]
#pragma unroll
  for (int ks = 0; ks < 6; ++ks)
#pragma unroll
    for (int p = 0; p < NP; ++p) Qf[ks][p] = att_m[(p * NCG + 2 * ks + par) * NQ + il];
  __syncthreads();

  // ---- Ek as a 32-"key" tile (rows >= 9 zero) for q . Ek; Ev as one k step of values ----
  for (int u = tid; u < NCG * 32; u += NT) {
    const int cg = u / 32, r = u % 32;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r < NR ? ek[r * DK + 8 * cg + e] : 0.f;
    uint4 pl[NP];
    att_split_unit<NP>(v, pl, amax);
#pragma unroll
    for (int p = 0; p < NP; ++p) kq[(p * NCG + cg) * ATT_KT + r] = pl[p];
  }
  for (int u = tid; u < DK * 2; u += NT) {
    const int c = u >> 1, hp = u & 1;              // k slot e of half hp <-> embedding row 8 * (e >> 2) + 4 * hp + (e & 3)
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * (e >> 2) + 4 * hp + (e & 3);
      v[e] = r < NR ? ev[r * DK + c] : 0.f;
    }
    uint4 pl[NP];
    att_split_unit<NP>(v, pl, amax);
#pragma unroll
    for (int p = 0; p < NP; ++p) evq[(p * DK + c) * 2 + hp] = pl[p];
  }
  __syncthreads();
  {
    att_f32x16 e;
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      uint4 a[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) a[p] = kq[(p * NCG + 2 * ks + par) * ATT_KT + l31];
      e = att_mma_split<NP>(a, Qf[ks], e);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rel = 8 * (r >> 2) + 4 * par + (r & 3);
      if (rel < NR) qe_s[il * NR + rel] = e[r];
    }
  }

  auto stage_k = [&](int jb) __attribute__((always_inline)) {
    for (int u = tid; u < NCG * ATT_KT; u += NT) {
      const int cg = u / ATT_KT, j = u % ATT_KT;
      const bool ok = jb + j < L;
      const float* kp = kb + (long long)(8 * cg) * L + (ok ? jb + j : L - 1);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = kp[(long long)e * L];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : 0.f;
      uint4 pl[NP];
      att_split_unit<NP>(v, pl, amax);
#pragma unroll
      for (int p = 0; p < NP; ++p) kq[(p * NCG + cg) * ATT_KT + j] = pl[p];
    }
  };
  auto stage_v = [&](int jb) __attribute__((always_inline)) {
    for (int u = tid; u < DK * 8; u += NT) {
      const int c = u >> 3, s8 = u & 7;            // s8 = 2 * (k step of 16 keys) + half
      const int k0 = jb + 16 * (s8 >> 1) + 4 * (s8 & 1);
      const float* vp = vb + (long long)c * L;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = k0 + 8 * (e >> 2) + (e & 3);
        v[e] = vp[j < L ? j : L - 1];
        v[e] = j < L ? v[e] : 0.f;
      }
      uint4 pl[NP];
      att_split_unit<NP>(v, pl, amax);
#pragma unroll
      for (int p = 0; p < NP; ++p) vq[(p * DK + c) * 8 + s8] = pl[p];
    }
  };
  // S' of the staged key tile for this wave's 32 queries, relative-position term and masking applied
  auto scores = [&](int jb, bool near, att_f32x16 (&sc)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[mt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        uint4 a[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = kq[(p * NCG + 2 * ks + par) * ATT_KT + 32 * mt + l31];
        sc[mt] = att_mma_split<NP>(a, Qf[ks], sc[mt]);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jb + 32 * mt + 8 * (r >> 2) + 4 * par + (r & 3);
        float v = sc[mt][r];
        if (near) {
          const int d = j - iq + W;
          if (d >= 0 && d < NR) v += qe_s[il * NR + d];
        }
        if (!(iq < len && j < len)) v = -1e4f;
        sc[mt][r] = j < L ? v : -INFINITY;
      }
  };

  // ---- one pass: scores, running maximum, p = exp(s - max), row sums, p . v ----
  att_f32x16 o[3];
#pragma unroll
  for (int mc = 0; mc < 3; ++mc)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mc][r] = 0.f;
  float lsum = 0.f, mx = -INFINITY;
  for (int jb = 0; jb < L; jb += ATT_KT) {
    __syncthreads();
    stage_k(jb);
    stage_v(jb);
    __syncthreads();
    // the band |j - i| <= W touches this tile only when the tile's keys come near the wave's queries
    const int wq0 = i0 + 32 * wave;
    const bool near = jb <= wq0 + 31 + W && jb + ATT_KT - 1 >= wq0 - W;
    att_f32x16 sc[2];
    scores(jb, near, sc);
    float tm = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tm = fmaxf(tm, sc[mt][r]);
    tm = fmaxf(tm, __shfl_xor(tm, 32));            // the other half of the tile's keys for the same query
    // a tile always holds a key below L, whose score is finite: the new maximum is finite, and exp(-inf - it) = 0
    const float mnew = fmaxf(mx, tm);
    const float alpha = expf(mx - mnew);
    mx = mnew;
    lsum *= alpha;
#pragma unroll
    for (int mc = 0; mc < 3; ++mc)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mc][r] *= alpha;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (near) {
          const int j = jb + 32 * mt + 8 * (r >> 2) + 4 * par + (r & 3);
          const int d = j - iq + W;
          if (d >= 0 && d < NR && j < L) pw_s[il * NR + d] = sc[mt][r];
        }
        const float p = expf(sc[mt][r] - mx);      // exp(-inf) = 0 for keys beyond L
        sc[mt][r] = p;
        lsum += p;
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = sc[mt][8 * hh + e];
        uint4 pb[NP];
        float unused = 0.f;                        // probabilities lie in [0, 1]
        att_split_unit<NP>(pv, pb, unused);
        const int s8 = 2 * (2 * mt + hh) + par;
#pragma unroll
        for (int mc = 0; mc < 3; ++mc) {
          uint4 a[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) a[p] = vq[(p * DK + 32 * mc + l31) * 8 + s8];
          o[mc] = att_mma_split<NP>(a, pb, o[mc]);
        }
      }
  }
  lsum += __shfl_xor(lsum, 32);
  // ---- the band's probabilities (against the final maximum) times Ev: one more k step (LDS accesses of one wave
  // complete in order, and pw_s rows are written and read by the two lanes of the same query only) ----
  {
    float pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * (e >> 2) + 4 * par + (e & 3);
      pv[e] = r < NR ? expf(pw_s[il * NR + r] - mx) : 0.f;
    }
    uint4 pb[NP];
    float unused = 0.f;
    att_split_unit<NP>(pv, pb, unused);
#pragma unroll
    for (int mc = 0; mc < 3; ++mc) {
      uint4 a[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) a[p] = evq[(p * DK + 32 * mc + l31) * 2 + par];
      o[mc] = att_mma_split<NP>(a, pb, o[mc]);
    }
  }
  if (iq < L) {
    const float rinv = 1.f / lsum;
    float* ob = out + ((long long)b * C + h * DK) * L + iq;
#pragma unroll
    for (int mc = 0; mc < 3; ++mc)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(long long)(32 * mc + 4 * par + (r & 3) + 8 * (r >> 2)) * L] = o[mc][r] * rinv;
  }
  if constexpr (NP == 2) {
    if (range_flag && !(amax <= 65504.f)) atomicOr(range_flag, 1);
  }
}

template <int NW, int NP>
static hipError_t attention_mfma_launch(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out,
                                  int B, int H, int L, int* range_flag, hipStream_t s) {
  constexpr int NQ = 32 * NW;
  const size_t lds = (size_t)(NP * ATT_NCG * ATT_KT + NP * ATT_DK * 8 + NP * ATT_DK * 2) * sizeof(uint4) +
                     (size_t)2 * NQ * (2 * ATT_W + 1) * sizeof(float);
  static LdsAttr attr;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(&attention_mfma_kernel<NW, NP>), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((attention_mfma_kernel<NW, NP>), dim3((L + NQ - 1) / NQ, H, B), dim3(64 * NW), lds, s, qkv, emb_k, emb_v,
                     lengths, out, H, L, range_flag);
  return hipGetLastError();
}

// VITS_ATT_MFMA: the shortest sequence that takes the matrix-core kernel (default 256; 0 = never).
static int att_mfma_min_len() {
  static const int n = [] {
    const char* e = getenv("VITS_ATT_MFMA");
    return e && e[0] ? atoi(e) : 256;
  }();
  return n;
}

hipError_t launch_attention(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out,
                            int B, int H, int DK, int L, int window, hipStream_t s, bool f16x2, int* range_flag) {
  if (DK != ATT_DK || window != ATT_W || B <= 0 || L <= 0 || L > 16384 || B > 65535) return hipErrorInvalidValue;
  // The matrix-core kernel is a block-serial walk over the keys (a barrier pair per 64 keys): it wins where its grid
  // gives at least every second CU a block or the row is long (the other kernel's cost per query grows with L twice:
  // scores and p.v on the vector ALUs); a short row in a small batch (batch-1 latency at 258 frames: 10 blocks) stays
  // with the other one.  (16 x 258 frames, 160 blocks: 60 against 67 us alone, and beside the generator in throughput
  // mode its 160 two-wave blocks take less of the chip than the other kernel's 544: +0.5 % on the whole job.)
  const long long blocks64 = (long long)((L + 63) / 64) * H * B;
  if (att_mfma_min_len() > 0 && L >= att_mfma_min_len() && (L >= 2 * att_mfma_min_len() || 2 * blocks64 >= device_cu_count())) {
    // 128 queries a block when that still gives every CU two blocks, else 64
    const bool wide = (long long)((L + 127) / 128) * H * B >= 2LL * device_cu_count();
    if (f16x2)
      return wide ? attention_mfma_launch<4, 2>(qkv, emb_k, emb_v, lengths, out, B, H, L, range_flag, s)
                  : attention_mfma_launch<2, 2>(qkv, emb_k, emb_v, lengths, out, B, H, L, range_flag, s);
    return wide ? attention_mfma_launch<4, 3>(qkv, emb_k, emb_v, lengths, out, B, H, L, nullptr, s)
                : attention_mfma_launch<2, 3>(qkv, emb_k, emb_v, lengths, out, B, H, L, nullptr, s);
  }
  if (L > 2048) return hipErrorInvalidValue;       // the lanes-on-keys kernel holds a whole score row in registers
  const int nch = (L + 63) / 64;
#define ATT_LAUNCH(N, Q) attention_launch<N, Q>(qkv, emb_k, emb_v, lengths, out, B, H, L, s)
  // The score row lives in registers: QPW * NCH floats per lane, so longer rows take fewer queries per wave.
  // Short rows have a choice: 16, 8 or 4 queries per wave.  A block re-stages every key and value of its (batch, head),
  // so fewer, larger blocks do less work, but phoneme-rate and batch-1 launches would leave most CUs idle: take the
  // largest block that still gives two blocks per CU, else the smallest.
  int qpw = 16;
  if (nch <= 5) {
    const long long per_row = (long long)H * B, want = 2LL * device_cu_count();
    while (qpw > 4 && ((L + 4 * qpw - 1) / (4 * qpw)) * per_row < want) qpw >>= 1;
  }
#define ATT_PICK(N) (qpw == 16 ? ATT_LAUNCH(N, 16) : qpw == 8 ? ATT_LAUNCH(N, 8) : ATT_LAUNCH(N, 4))
  if (nch <= 1) ATT_PICK(1);
  else if (nch <= 2) ATT_PICK(2);
  else if (nch <= 3) ATT_PICK(3);
  else if (nch <= 5) ATT_PICK(5);
  else if (nch <= 8) ATT_LAUNCH(8, 8);
  else if (nch <= 12) ATT_LAUNCH(12, 8);
  else if (nch <= 16) ATT_LAUNCH(16, 4);
  else if (nch <= 24) ATT_LAUNCH(24, 4);
  else ATT_LAUNCH(32, 4);
#undef ATT_PICK
#undef ATT_LAUNCH
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// embedding lookups and conditioning
// ---------------------------------------------------------------------------------------------------------
// x[b][c][p] = table[ids[b][p]][c] * sqrt(C) * (p < len[b])     (reference models.py:176-180)
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                             const int* __restrict__ lengths, float* __restrict__ x, int B, int P, int C, int n_vocab,
                             int* err_flag) {
  const long long n = (long long)B * C * P;
  const float scale = sqrtf((float)C);
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const int p = idx % P;
    const int c = (idx / P) % C;
    const int b = idx / ((long long)P * C);
    long long id = ids[(long long)b * P + p];
    if (id < 0 || id >= n_vocab) {
      if (err_flag) atomicOr(err_flag, 1);
      id = 0;
    }
    x[idx] = p < lengths[b] ? table[id * C + c] * scale : 0.f;
  }
}

hipError_t launch_embed(const int64_t* ids, const float* table, const int* lengths, float* x, int B, int P, int C,
                        int n_vocab, int* err_flag, hipStream_t s) {
  const long long n = (long long)B * C * P;
  if (n <= 0) return hipErrorInvalidValue;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(embed_kernel, dim3(blocks), dim3(256), 0, s, ids, table, lengths, x, B, P, C, n_vocab, err_flag);
  return hipGetLastError();
}

// out[b][row] = bias[row] + sum_ci w[row][ci] * emb_g[sid[b]][ci]: every 1x1 conditioning conv on the speaker
// embedding (reference models.py:129,282-287,551,719-720; modules.py:149; frame_prior_network.py:126) in one launch.
__global__ __launch_bounds__(256) void cond_gemv_kernel(const float* __restrict__ emb_g, const int64_t* __restrict__ sid,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int rows, int gin, int n_speakers,
                                                         int* err_flag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (row >= rows) return;
  long long sp = sid[b];
  if (sp < 0 || sp >= n_speakers) {
    if (err_flag && lane == 0) atomicOr(err_flag, 2);
    sp = 0;
  }
  const float* g = emb_g + sp * gin;
  const float* wr = w + (long long)row * gin;
  float s = 0.f;
  for (int c = lane; c < gin; c += 64) s += wr[c] * g[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[(long long)b * rows + row] = s + bias[row];
}

hipError_t launch_cond_gemv(const float* emb_g, const int64_t* sid, const float* w, const float* bias, float* out,
                            int B, int rows, int gin, int n_speakers, int* err_flag, hipStream_t s) {
  if (B <= 0 || rows <= 0 || B > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cond_gemv_kernel, dim3((rows + 3) / 4, B), dim3(256), 0, s, emb_g, sid, w, bias, out, rows, gin,
                     n_speakers, err_flag);
  return hipGetLastError();
}

// y[b][c][t] = (x[b][c][t] + cb[b][c]) * (mask ? t < len[b] : 1)      (reference models.py:551-553)
__global__ void add_cbias_mask_kernel(const float* __restrict__ x, const float* __restrict__ cb, int cb_bs,
                                      const int* __restrict__ lengths, float* __restrict__ y, int B, int C, int L,
                                      int mask) {
  const long long n = (long long)B * C * L;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const int t = idx % L;
    const int c = (idx / L) % C;
    const int b = idx / ((long long)L * C);
    const float v = x[idx] + cb[(long long)b * cb_bs + c];
    y[idx] = (!mask || t < lengths[b]) ? v : 0.f;
  }
}

hipError_t launch_add_cbias_mask(const float* x, const float* cb, int cb_bs, const int* lengths, float* y, int B, int C,
                                 int L, bool mask, hipStream_t s) {
  const long long n = (long long)B * C * L;
  if (n <= 0) return hipErrorInvalidValue;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_cbias_mask_kernel, dim3(blocks), dim3(256), 0, s, x, cb, cb_bs, lengths, y, B, C, L,
                     mask ? 1 : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// conv with ONE output channel: out[b][t] = act(bias + sum_{c,k} w[c*K+k] * pro(x[b][c][t+k-(K-1)/2]))
// (reference models.py:120 duration proj, :543 proj_f0, :278 conv_post; frame_prior_network.py:101 linear_layer)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_to1_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, const int* __restrict__ lengths, int C,
                                                        int K, int L, int flags, float in_scale, float in_slope,
                                                        int tanh_out, int mask_out) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const int len = lengths ? lengths[b] : 0x7fffffff;
  const int pad = (K - 1) / 2;
  const float* xb = x + (long long)b * x_bs;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (long long)c * x_cs;
    for (int k = 0; k < K; ++k) {
      const int tt = t + k - pad;
      bool ok = tt >= 0 && tt < L;
      if (flags & F_IN_MASK) ok = ok && tt < len;
      float v = ok ? xr[tt] * in_scale : 0.f;
      if (flags & F_IN_LRELU) v = v > 0.f ? v : v * in_slope;
      acc += w[c * K + k] * v;
    }
  }
  if (bias) acc += bias[0];
  if (tanh_out) acc = tanhf(acc);
  if (mask_out && t >= len) acc = 0.f;
  out[(long long)b * L + t] = acc;
}

// Short sequences: block = 16 time steps x 16 channel groups (channels g, g + 16, ...), partial sums reduced with
// shuffles across the four groups of a wave and through LDS across waves: the reduction over up to 768 channels is
// no longer one thread's serial chain.
__global__ __launch_bounds__(256) void conv_to1_split_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              const int* __restrict__ lengths, int C, int K, int L,
                                                              int flags, float in_scale, float in_slope, int tanh_out,
                                                              int mask_out) {
  __shared__ float red[4][16];
  const int tl = threadIdx.x & 15, g = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 16 + tl;
  const int len = lengths ? lengths[b] : 0x7fffffff;
  const int pad = (K - 1) / 2;
  const float* xb = x + (long long)b * x_bs;
  float acc = 0.f;
  if (t < L) {
#pragma unroll 4
    for (int c = g; c < C; c += 16) {
      const float* xr = xb + (long long)c * x_cs;
      for (int k = 0; k < K; ++k) {
        const int tt = t + k - pad;
        bool ok = tt >= 0 && tt < L;
        if (flags & F_IN_MASK) ok = ok && tt < len;
        float v = ok ? xr[tt] * in_scale : 0.f;
        if (flags & F_IN_LRELU) v = v > 0.f ? v : v * in_slope;
        acc += w[c * K + k] * v;
      }
    }
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if ((threadIdx.x & 63) < 16) red[wave][tl] = acc;
  __syncthreads();
  if (threadIdx.x < 16 && t < L) {
    acc = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
    if (bias) acc += bias[0];
    if (tanh_out) acc = tanhf(acc);
    if (mask_out && t >= len) acc = 0.f;
    out[(long long)b * L + t] = acc;
  }
}

// Long sequences with 16-byte aligned rows: a thread produces four consecutive outputs from float4 loads of the row
// segment [t - 4, t + 8) (K <= 9), so a loaded value serves up to K taps from registers.
template <int K>
__global__ __launch_bounds__(256) void conv_to1_vec4_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             const int* __restrict__ lengths, int C, int L,
                                                             int flags, float in_scale, float in_slope, int tanh_out,
                                                             int mask_out) {
  const int b = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= L) return;
  const int len = lengths ? lengths[b] : 0x7fffffff;
  constexpr int pad = (K - 1) / 2;
  const float* xb = x + (long long)b * x_bs;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (long long)c * x_cs;
    float v[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int tq = t - 4 + 4 * q;
      float4 f = {0.f, 0.f, 0.f, 0.f};
      if (tq >= 0 && tq < L) f = *reinterpret_cast<const float4*>(xr + tq);
      v[4 * q + 0] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const int tt = t - 4 + e;
      float u = v[e] * in_scale;
      if ((flags & F_IN_MASK) && tt >= len) u = 0.f;
      if (flags & F_IN_LRELU) u = u > 0.f ? u : u * in_slope;
      v[e] = u;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wk = w[c * K + k];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += wk * v[4 + o + k - pad];   // input position t + o + k - pad
    }
  }
  float4 r;
  float* rp = &r.x;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    float a = acc[o];
    if (bias) a += bias[0];
    if (tanh_out) a = tanhf(a);
    if (mask_out && t + o >= len) a = 0.f;
    rp[o] = a;
  }
  *reinterpret_cast<float4*>(out + (long long)b * L + t) = r;
}

hipError_t launch_conv_to1(const float* x, long long x_bs, int x_cs, const float* w, const float* bias, float* out,
                           const int* lengths, int B, int C, int K, int L, int flags, float in_scale, float in_slope,
                           bool tanh_out, bool mask_out, hipStream_t s) {
  if (B <= 0 || L <= 0 || B > 65535) return hipErrorInvalidValue;
  if (L <= 2048) {
    hipLaunchKernelGGL(conv_to1_split_kernel, dim3((L + 15) / 16, B), dim3(256), 0, s, x, x_bs, x_cs, w, bias, out,
                       lengths, C, K, L, flags, in_scale, in_slope, tanh_out ? 1 : 0, mask_out ? 1 : 0);
    return hipGetLastError();
  }
  if (K == 7 && L % 4 == 0 && x_cs % 4 == 0 && x_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(conv_to1_vec4_kernel<7>, dim3((L / 4 + 255) / 256, B), dim3(256), 0, s, x, x_bs, x_cs, w, bias, out,
                       lengths, C, L, flags, in_scale, in_slope, tanh_out ? 1 : 0, mask_out ? 1 : 0);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(conv_to1_kernel, dim3((L + 255) / 256, B), dim3(256), 0, s, x, x_bs, x_cs, w, bias, out, lengths,
                     C, K, L, flags, in_scale, in_slope, tanh_out ? 1 : 0, mask_out ? 1 : 0);
  return hipGetLastError();
}

// x[b][c][t] += bias[c] + sum_k w[c*3+k] * sig[b][t+k-1]   (zero padded; reference models.py:657-658,742,752)
__global__ void prenet_add_kernel(float* __restrict__ x, const float* __restrict__ sig, const float* __restrict__ w,
                                  const float* __restrict__ bias, int B, int C, int L) {
  const long long n = (long long)B * C * L;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const int t = idx % L;
    const int c = (idx / L) % C;
    const int b = idx / ((long long)L * C);
    const float* sb = sig + (long long)b * L;
    const float s0 = t > 0 ? sb[t - 1] : 0.f, s1 = sb[t], s2 = t + 1 < L ? sb[t + 1] : 0.f;
    x[idx] += (w[c * 3] * s0 + w[c * 3 + 1] * s1 + w[c * 3 + 2] * s2) + bias[c];
  }
}

hipError_t launch_prenet_add(float* x, const float* sig, const float* w, const float* bias, int B, int C, int L,
                             hipStream_t s) {
  const long long n = (long long)B * C * L;
  if (n <= 0) return hipErrorInvalidValue;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(prenet_add_kernel, dim3(blocks), dim3(256), 0, s, x, sig, w, bias, B, C, L);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// scalar control math  (reference models.py:732-733, 737-743, 747-753, 763)
// ---------------------------------------------------------------------------------------------------------
__global__ void duration_finalize_kernel(const float* __restrict__ logw, const int* __restrict__ lengths, float scale,
                                         float* __restrict__ dur, int B, int P) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * P) return;
  const int b = idx / P, p = idx % P;
  const float m = p < lengths[b] ? 1.f : 0.f;
  dur[idx] = ceilf((expf(logw[idx]) * m - 1.f) * scale);
}

hipError_t launch_duration_finalize(const float* logw, const int* lengths, float scale, float* dur, int B, int P,
                                    hipStream_t s) {
  hipLaunchKernelGGL(duration_finalize_kernel, dim3((B * P + 255) / 256), dim3(256), 0, s, logw, lengths, scale, dur, B, P);
  return hipGetLastError();
}

__global__ void pitch_math_kernel(const float* __restrict__ in, int in_is_hz, float scale, float* __restrict__ lf0,
                                  float* __restrict__ f0, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float v = in[idx];
  const float l = in_is_hz ? (2595.f * log10f(1.f + v / 700.f)) / 500.f : v * scale;
  lf0[idx] = l;
  f0[idx] = (powf(10.f, l * 500.f / 2590.f) - 1.f) * 700.f;      // 2590, not 2595: as the reference (models.py:743)
}

hipError_t launch_pitch_math(const float* in, bool in_is_hz, float scale, float* lf0, float* f0, int n, hipStream_t s) {
  hipLaunchKernelGGL(pitch_math_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, in_is_hz ? 1 : 0, scale, lf0, f0, n);
  return hipGetLastError();
}

__global__ void energy_math_kernel(const float* __restrict__ in, int in_is_raw, float scale, float* __restrict__ norm,
                                   float* __restrict__ energy, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float v = in[idx];
  const float ne = in_is_raw ? (v - 60.f) / 36.f : (((v * 36.f + 60.f) * scale) - 60.f) / 36.f;
  norm[idx] = ne;
  energy[idx] = ne * 36.f + 60.f;
}

hipError_t launch_energy_math(const float* in, bool in_is_raw, float scale, float* norm, float* energy, int n,
                              hipStream_t s) {
  hipLaunchKernelGGL(energy_math_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, in_is_raw ? 1 : 0, scale, norm,
                     energy, n);
  return hipGetLastError();
}

// z_p = m_p + noise * exp(logs_p) * noise_scale, with stats = [B, 2C, L] holding m_p then logs_p; z gets a copy.
// lengths != nullptr: positions t >= lengths[b] are zeroed (posterior encoder, reference models.py:249).
__global__ void sample_kernel(const float* __restrict__ stats, const float* __restrict__ noise, float noise_scale,
                              float* __restrict__ z_p, float* __restrict__ z, int B, int C, int L,
                              const int* __restrict__ lengths) {
  const long long n = (long long)B * C * L;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    const long long cl = idx % ((long long)C * L);
    const long long b = idx / ((long long)C * L);
    const float m = stats[b * 2 * C * L + cl];
    const float ls = stats[b * 2 * C * L + (long long)C * L + cl];
    float v = m + noise[idx] * expf(ls) * noise_scale;
    if (lengths && (int)(cl % L) >= lengths[b]) v = 0.f;
    z_p[idx] = v;
    z[idx] = v;
  }
}

hipError_t launch_sample(const float* stats, const float* noise, float noise_scale, float* z_p, float* z, int B, int C,
                         int L, hipStream_t s, const int* lengths) {
  const long long n = (long long)B * C * L;
  if (n <= 0) return hipErrorInvalidValue;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(sample_kernel, dim3(blocks), dim3(256), 0, s, stats, noise, noise_scale, z_p, z, B, C, L, lengths);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// fp32 activations -> the staged f16x2 operand format of the conv kernels: y = leaky-relu(x * scale, slope) clamped to
// the fp16 range, h = fp16(y), l = fp16(y - h), written as [B][C/8][2 planes h, l][L] 16-byte units of eight channels
// (the input prologue of the ResBlock convs, reference modules.py:208-209, applied once for all three ResBlocks that
// read the same tensor).  A thread owns one (channel group, time step): eight coalesced loads, two 16-byte stores.
// ---------------------------------------------------------------------------------------------------------
typedef float st_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 st_f16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stage_f16x2_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C, int L,
                                                          float scale, float slope, int* range_flag) {
  const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* xb = x + ((long long)b * C + cg * 8) * L + t;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = xb[(long long)i * L];
  unsigned ph[4], pl[4];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st_f32x2 e;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float y = v[2 * i + u] * scale;
      y = fmaxf(y, y * slope);
      amax = fmaxf(amax, fabsf(y));
      e[u] = fminf(fmaxf(y, -65504.f), 65504.f);
    }
    const st_f16x2 h = __builtin_convertvector(e, st_f16x2);
    const st_f16x2 l = __builtin_convertvector(e - __builtin_convertvector(h, st_f32x2), st_f16x2);
    ph[i] = __builtin_bit_cast(unsigned, h);
    pl[i] = __builtin_bit_cast(unsigned, l);
  }
  if (range_flag && !(amax <= 65504.f)) atomicOr(range_flag, 1);
  uint4* ob = out + ((long long)b * (C / 8) + cg) * 2 * L + t;
  ob[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
  ob[L] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}

hipError_t launch_stage_f16x2(const float* x, float* out, int B, int C, int L, float scale, float slope, int* range_flag,
                              hipStream_t s) {
  if (B <= 0 || C <= 0 || C % 8 || L <= 0 || B > 65535 || C / 8 > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stage_f16x2_kernel, dim3((L + 255) / 256, C / 8, B), dim3(256), 0, s, x,
                     reinterpret_cast<uint4*>(out), C, L, scale, slope, range_flag);
  return hipGetLastError();
}

// The same for the plain-bf16 arithmetic: y = leaky-relu(x * scale, slope) rounded to nearest-even bf16, written as
// [B][C/8][L] 16-byte units of eight channels -- the values conv_bf16_kernel converts at staging.
typedef __bf16 st_bf16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stage_bf16_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C, int L,
                                                         float scale, float slope) {
  const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* xb = x + ((long long)b * C + cg * 8) * L + t;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = xb[(long long)i * L];
  unsigned p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st_f32x2 e;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float y = v[2 * i + u] * scale;
      e[u] = fmaxf(y, y * slope);
    }
    p[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, st_bf16x2));
  }
  out[((long long)b * (C / 8) + cg) * L + t] = make_uint4(p[0], p[1], p[2], p[3]);
}

hipError_t launch_stage_bf16(const float* x, float* out, int B, int C, int L, float scale, float slope, hipStream_t s) {
  if (B <= 0 || C <= 0 || C % 8 || L <= 0 || B > 65535 || C / 8 > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stage_bf16_kernel, dim3((L + 255) / 256, C / 8, B), dim3(256), 0, s, x,
                     reinterpret_cast<uint4*>(out), C, L, scale, slope);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// length regulator (reference models.py:426-455): phoneme p of row b is repeated max(int(dur[b][p]), 0) times.
// plan: integer durations -> inclusive prefix sum per row (block scan in LDS);  gather: per frame, binary search
// of the prefix sums, then a coalesced copy of the phoneme's channel vector.
// ---------------------------------------------------------------------------------------------------------
constexpr int LR_MAX_P = 2048;

__device__ inline int dur_to_int(const void* dur, int dtype, long long i) {
  long long v;
  switch (dtype) {
    case 0: { const float f = ((const float*)dur)[i]; v = (long long)f; break; }     // int(): truncation
    case 1: v = ((const int64_t*)dur)[i]; break;
    case 2: v = ((const int*)dur)[i]; break;
    default: { const double f = ((const double*)dur)[i]; v = (long long)f; break; }
  }
  if (v < 0) v = 0;
  if (v > (1 << 24)) v = 1 << 24;
  return (int)v;
}

// report (optional): [gridDim.x + 2] = the frame counts again, then the words flag0 and flag1 point to (0 for a null
// pointer) -- everything the host reads after the plan, in one place, so that it takes one copy and no gather kernel
__global__ __launch_bounds__(256) void lr_plan_kernel(const void* __restrict__ dur, int dtype, int P,
                                                       int* __restrict__ cum, int* __restrict__ totals,
                                                       int* __restrict__ report, const int* flag0, const int* flag1) {
  __shared__ int buf[2][LR_MAX_P];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int p = tid; p < P; p += 256) buf[0][p] = dur_to_int(dur, dtype, (long long)b * P + p);
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < P; off <<= 1) {
    for (int p = tid; p < P; p += 256) buf[cur ^ 1][p] = buf[cur][p] + (p >= off ? buf[cur][p - off] : 0);
    __syncthreads();
    cur ^= 1;
  }
  for (int p = tid; p < P; p += 256) cum[(long long)b * P + p] = buf[cur][p];
  if (tid == 0) {
    totals[b] = buf[cur][P - 1];
    if (report) {
      report[b] = buf[cur][P - 1];
      if (b == 0) {
        report[gridDim.x] = flag0 ? *flag0 : 0;
        report[gridDim.x + 1] = flag1 ? *flag1 : 0;
      }
    }
  }
}

hipError_t launch_lr_plan(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, hipStream_t s, int* report,
                          const int* flag0, const int* flag1) {
  if (B <= 0 || P <= 0 || P > LR_MAX_P || dur_dtype < 0 || dur_dtype > 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lr_plan_kernel, dim3(B), dim3(256), 0, s, dur, dur_dtype, P, cum, totals, report, flag0, flag1);
  return hipGetLastError();
}

constexpr int LR_CH_PER_BLOCK = 16;

__global__ __launch_bounds__(256) void lr_gather_kernel(const float* __restrict__ x, const int* __restrict__ cum,
                                                         const int* __restrict__ totals, float* __restrict__ out, int C,
                                                         int P, int T) {
  __shared__ int cs[LR_MAX_P];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int p = tid; p < P; p += 256) cs[p] = cum[(long long)b * P + p];
  __syncthreads();
  const int t = blockIdx.x * 256 + tid;
  if (t >= T) return;
  const int total = totals[b];
  int src = -1;
  if (t < total) {               // first p with cum[p] > t
    int lo = 0, hi = P - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cs[mid] > t) hi = mid; else lo = mid + 1;
    }
    src = lo;
  }
  // blockIdx.z: a group of LR_CH_PER_BLOCK channels (a thread walking all channels alone is one long chain of loads:
  // 44 us for one 3-second utterance)
  const int c0 = blockIdx.z * LR_CH_PER_BLOCK, c1 = c0 + LR_CH_PER_BLOCK < C ? c0 + LR_CH_PER_BLOCK : C;
  const float* xb = x + (long long)b * C * P;
  float* ob = out + (long long)b * C * T + t;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) ob[(long long)c * T] = src >= 0 ? xb[(long long)c * P + src] : 0.f;
}

hipError_t launch_lr_gather(const float* x, const int* cum, const int* totals, float* out, int B, int C, int P, int T,
                            hipStream_t s) {
  if (B <= 0 || P <= 0 || P > LR_MAX_P || T <= 0 || B > 65535) return hipErrorInvalidValue;
  if (C <= 0 || (C + LR_CH_PER_BLOCK - 1) / LR_CH_PER_BLOCK > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lr_gather_kernel, dim3((T + 255) / 256, B, (C + LR_CH_PER_BLOCK - 1) / LR_CH_PER_BLOCK), dim3(256), 0,
                     s, x, cum, totals, out, C, P, T);
  return hipGetLastError();
}

}  // namespace vits
