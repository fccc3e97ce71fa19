// Shared declarations for the MVTS synthesis kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

namespace vits {

// ---------------------------------------------------------------------------------------------------------
// conv1d / conv_transpose1d as an fp32 MFMA implicit GEMM:  out[row, t] = sum_{ci, tap} W[row, ci, tap] * x[ci, t + toff(tap)]
// ---------------------------------------------------------------------------------------------------------
enum ConvMode : int {
  MODE_PLAIN = 0,   // rows are output channels
  MODE_GATE  = 1,   // rows [0, M/2) and [M/2, M) are combined as tanh(a) * sigmoid(s)  (WaveNet gate)
  MODE_UP8   = 2,   // transposed conv, stride 8, kernel 16, padding 4: rows are (channel, phase) pairs
  MODE_UP4   = 3,   // transposed conv, stride 4, kernel 4,  padding 0
  MODE_UP2   = 4,   // transposed conv, stride 2, kernel 4,  padding 1
};

enum ConvPrec : int {
  PREC_F32    = 0,   // v_mfma_f32_32x32x2_f32, packed fp32 weights
  PREC_BF16X3 = 1,   // three-term bf16 split of both operands on v_mfma_f32_32x32x16_bf16 (fp32-grade results)
  PREC_F16X2  = 3,   // two-term fp16 split of both operands, three products on v_mfma_f32_32x32x16_f16 (2^-22 relative)
  PREC_BF16   = 2,   // operands rounded to bf16, one product per MAC on v_mfma_f32_32x32x16_bf16, fp32 accumulate
};

enum ConvFlags : int {
  F_IN_LRELU  = 1 << 0,   // leaky-relu(in_slope) on the input
  F_IN_MASK   = 1 << 1,   // input *= (t < len[b])
  F_RELU      = 1 << 2,   // relu on the result
  F_MASK_PRE  = 1 << 3,   // result *= mask before residual / accumulate
  F_RES       = 1 << 4,   // result += res
  F_RES_SUB   = 1 << 5,   // result = res - result
  F_ACCUM     = 1 << 6,   // result += out (read-modify-write)
  F_MASK_POST = 1 << 7,   // result *= mask last
  // Staged activations (MODE_PLAIN, PREC_BF16 / PREC_F16X2 only): a tensor that only the next conv reads is written
  // by its producer already in that conv's LDS operand format -- leaky-relu(out_slope) applied, converted, eight
  // channels of one time step per 16-byte unit: bf16 [C/8][L][8], f16x2 [C/8][2 planes h, l][L][8] (a plane of a channel
  // group is one contiguous run over time, which LDS-DMA copies 64 time steps at a time) -- so the consumer stages
  // it with plain 16-byte copies (no prologue, no conversion).  x_bs / out_bs stay in floats.
  F_OUT_STAGED = 1 << 8,  // write the result in staged format (bias / conditioning bias only; no residual, mask, accumulate)
  F_IN_STAGED  = 1 << 9,  // the input is in staged format
};

constexpr int CI_CHUNK = 16;    // input channels staged per LDS tile
constexpr int MAX_TAPS = 12;
constexpr int MAX_HALO = 64;    // max (max_off - min_off) a tile may need

struct ConvDesc {          // one packed convolution, immutable after load
  const void* w = nullptr;       // PREC_F32: fp32 [Cin/16][K][2][M/32][64][4]; PREC_BF16X3: bf16 [Cin/16][K][M/32][3][64][8]; PREC_BF16: bf16 [Cin/16][K][M/32][64][8];
                                 // PREC_F16X2: fp16 [Cin/16][K][M/32][2][64][8]
  int prec = PREC_F32;
  // optional second packing of the same weights for PREC_F16X2 (a PREC_BF16X3 conv of the front end): launches too
  // large for the split-K kernel -- frame-rate convs of long batches -- then run three fp16 products instead of six
  // bf16 ones; short launches, where masks and durations are decided, keep bf16x3
  const void* w_alt = nullptr;
  const float* bias = nullptr;   // [M] per-row bias, may be null
  int Cin = 0, M = 0, K = 0, mode = MODE_PLAIN;
  int toff0[2] = {0, 0};         // input time offset of tap k, per row class: toff0 + k * tstep
  int tstep[2] = {0, 0};
  int min_off = 0, max_off = 0;
  int up = 1;                    // output time stride (1, 2, 4, 8)
  int cout = 0;                  // logical output channels
};

struct ConvArgs {
  const float* x; long long x_bs; int x_cs;          // input [B][Cin][*], batch / channel strides in floats
  const void* w; const float* bias;
  const float* cbias; int cbias_bs;                  // per-(batch,row) additive bias, or null
  const float* in_cbias; int in_cbias_bs;            // per-(batch,ci) additive input bias, or null
  float* out; long long out_bs; int out_cs;
  const float* res; long long res_bs; int res_cs;
  const int* lengths;                                // per-batch valid length, or null (all valid)
  int B, Cin, M, K, L;                               // L = number of time positions (input length)
  int mt_total;                                      // M / 32
  int min_off, width;                                // LDS tile covers t0 + min_off .. + width
  int rs;                                            // LDS row stride in floats
  int flags;
  float in_scale, in_slope;
  float out_slope;                                   // F_OUT_STAGED: leaky-relu slope the consumer would apply
  int toff0[2], tstep[2];                            // tap k reads x[t + toff0 + k * tstep] (per row class)
  // MODE_PLAIN, optional second destination: rows >= split_row (a multiple of 64; 0 = off) go to out_hi (row
  // split_row is its row 0) under flags_hi, with no residual; rows below it to out / res under flags.  One launch then
  // serves WaveNet's res_skip conv: the residual half updates x, the skip half accumulates into the skip sum.
  int split_row; int flags_hi;
  float* out_hi; long long out_hi_bs;
  float* out2; long long out2_bs;                    // MODE_PLAIN, optional: a second copy of the result in the staged
                                                     // f16x2 format (leaky-relu(out_slope) applied), for the next conv
  int* range_flag;                                   // PREC_F16X2: OR 1 into *range_flag when an activation lies outside
                                                     // the fp16 range and was clamped (null: not reported)
  // Deferred LayerNorm over channels (PREC_BF16X3, MODE_PLAIN, split-K kernel: conv_takes_split_k).  A producer leaves its result un-normalised and writes,
  // per time step and 64-row unit, (sum, sum of squared deviations from the unit's mean) to ln_out_stats
  // [B][L][M / 64][2]; a consumer normalises on the fly -- its input at staging (ln_in_*: statistics of the Cin input
  // channels) and / or its residual in the epilogue (ln_res_*: of the M rows) -- as (v - mean) * rstd * gamma[c] + beta[c]
  // with eps 1e-5 and the biased variance (reference modules.py:15-27).
  const float* ln_in_stats; const float* ln_in_gamma; const float* ln_in_beta;
  const float* ln_res_stats; const float* ln_res_gamma; const float* ln_res_beta;
  float* ln_out_stats;
  int xcd_map;                                        // launch_conv: blockIdx.x -> time tile goes through xcd_tile()
};

// Which time tile the block with blockIdx.x = bx of a row of ntiles blocks takes.  Workgroups go to the chip's eight
// XCDs round-robin in launch order, and each XCD has its own L2: with the identity mapping the neighbours of a tile --
// which read the same halo columns and the other halves of its cache lines -- always sit on other XCDs, and every one
// of them fetches those lines from memory again.  Here the blocks an XCD receives (bx = f, f + 8, f + 16, ...) take
// consecutive tiles, so neighbouring tiles share an L2 and run at about the same time.  A bijection for any ntiles.
__device__ __forceinline__ int xcd_tile(int bx, int ntiles, int on) {
  if (!on || ntiles < 16) return bx;
  const int f = bx & 7, q = ntiles >> 3, r = ntiles & 7;
  return f * q + (f < r ? f : r) + (bx >> 3);
}

// Time tile and row block of a block of a grid of ntiles x 2 blocks per batch row (a 256-row layer in 128-row blocks).
// The two row blocks of a time tile read the same activation tile; workgroups are dealt to the XCDs round-robin in
// launch order (x fastest), so with the plain mapping the two sit ntiles apart, on different XCDs as a rule, and the
// tile is fetched from memory twice.  Here the blocks l and l + 8 of a plane -- same XCD, consecutive in its queue --
// are the two row blocks of one tile, so the second finds the tile in the XCD's L2.  A bijection for any ntiles: the
// last (2 ntiles) % 16 blocks of a plane keep row blocks side by side.
__device__ __forceinline__ void xcd_tile_rows2(int bx, int by, int ntiles, int on, int& tile, int& rowblk) {
  if (!on) { tile = bx; rowblk = by; return; }
  const int l = bx + ntiles * by, full = (2 * ntiles) & ~15;
  if (l < full) {
    const int q = l >> 3;
    tile = ((q >> 1) << 3) + (l & 7);
    rowblk = q & 1;
  } else {
    const int r = l - full;
    tile = (full >> 1) + (r >> 1);
    rowblk = r & 1;
  }
}

// mean and 1 / sqrt(var + eps) of one time step from the per-unit partial statistics (Chan's combination of
// (sum, M2) pairs over units of 64 channels)
__device__ __forceinline__ void ln_combine(const float* __restrict__ st, int units, float& mean, float& rstd) {
  float total = 0.f;
  for (int u = 0; u < units; ++u) total += st[2 * u];
  mean = total / (64.f * (float)units);
  float m2 = 0.f;
  for (int u = 0; u < units; ++u) {
    const float d = st[2 * u] * (1.f / 64.f) - mean;
    m2 += st[2 * u + 1] + 64.f * d * d;
  }
  rstd = 1.f / sqrtf(m2 / (64.f * (float)units) + 1e-5f);
}

// A kernel that uses more than 64 KiB of dynamic LDS has to be given the attribute once per DEVICE (a model can be
// moved to another GPU of the process).  One of these per kernel instantiation (a function-local static); ensure()
// sets the attribute on the current device the first time it is called there.  Setting it twice is harmless, so a
// relaxed flag per device is all the synchronisation concurrent callers need.
struct LdsAttr {
  static constexpr int kMaxDevices = 64;
  std::atomic<bool> done[kMaxDevices];
  LdsAttr() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
  hipError_t ensure(const void* kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (tracked && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && tracked) done[dev].store(true, std::memory_order_release);
    return e;
  }
};

hipError_t launch_conv(const ConvDesc& d, ConvArgs a, hipStream_t stream);
// true when launch_conv would run this conv on the split-K kernel for a batch of B rows of L steps (the only kernel
// that honours ConvArgs::ln_*)
bool conv_takes_split_k(const ConvDesc& d, int B, int L);
bool alt_prefers_f16x2(int B, int L);   // a conv with the f16x2 packing skips the split-K kernel at this size
// per host thread: the front-end stage being launched runs beside a generator on other streams (throughput mode)
void set_front_end_overlapped(bool on);
bool front_end_overlapped();
void set_alt_f16x2_mode(int mode);      // ConvDesc::w_alt: 0 never, 1 launches beyond the split-K kernel (default), 2 always
int alt_f16x2_mode();

// One ResBlock pair in one launch (32-channel stage, PREC_F16X2): out = x + c2(lrelu(c1(lrelu(x)))) [+ out], with the
// tensor between the two convs kept in LDS.  x and out must not overlap (blocks read a halo of x).
struct PairArgs {
  const float* x; long long x_bs; int x_cs;
  float* out; long long out_bs; int out_cs;
  int B, L;
  int accumulate;                // out += result (read-modify-write)
  float slope;                   // leaky-relu slope before both convs
  int* range_flag = nullptr;     // OR 1 into it when an activation was clamped to the fp16 range (null: not reported)
  int xcd_map = 0;               // launch_resblock_pair: blockIdx.x -> time tile goes through xcd_tile()
};
bool xcd_map_enabled();          // VITS_XCD_MAP=0 switches the mapping off (A/B measurements)
bool pair_shape_fusable(const ConvDesc& c1, const ConvDesc& c2); // ... everything but the channel count
bool pair_fusable(const ConvDesc& c1, const ConvDesc& c2);       // the kernel supports the shape (32 or 64 channels)
bool pair_profitable(const ConvDesc& c1, const ConvDesc& c2);    // ... and is the faster choice for it
hipError_t launch_resblock_pair(const ConvDesc& c1, const ConvDesc& c2, const PairArgs& a, hipStream_t stream);

// A whole ResBlock -- npair pairs of the same width, tap count and arithmetic, one dilation each -- in one launch
// (kernels_chain.hip): the carrier stays on chip across the pairs.  a.x is the ResBlock's input, a.out its output (or
// the sum it joins, with a.accumulate); they must not overlap.
bool chain_fusable(const ConvDesc* c1, const ConvDesc* c2, int npair);      // the kernel supports the chain
bool chain_profitable(const ConvDesc* c1, const ConvDesc* c2, int npair);   // ... and is the faster choice for it
hipError_t launch_resblock_chain(const ConvDesc* c1, const ConvDesc* c2, int npair, const PairArgs& a, hipStream_t stream);
// the launches a ResBlock is cut into: seg[i] consecutive pairs each (a segment of one pair is a fused-pair launch);
// returns their number, 0 when the ResBlock has no such plan
int chain_plan(const ConvDesc* c1, const ConvDesc* c2, int npair, int* seg);

// Optional per-kernel timing of the conv launches with HIP events on the launch stream (off by default).
struct ProfRow {
  std::string name;      // kernel instantiation, e.g. conv_mfma_kernel<2,2,1,4,0> = <WM,WN,WGM,WGN,MODE>
  double flops, ms;      // algorithmic FLOPs (2*rows*cin*taps*L*B) and summed event time
  double bytes;          // algorithmic HBM bytes: input + output (+ residual, + accumulated output) + packed weights
  long long count;
};
void prof_enable(bool on);
bool prof_enabled();
int device_cu_count();      // compute units of the current device (256 on MI355X), cached
void prof_reset();
int prof_collect(std::vector<ProfRow>& rows);
// Timing hooks for a conv launcher: prof_begin records the first event on the launch stream (when timing is on),
// prof_end the second, and books the launch's algorithmic FLOPs and bytes under `name`.
struct ProfTicket {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool on = false;
};
ProfTicket prof_begin(hipStream_t stream);
void prof_end(const ProfTicket& t, const char* name, bool gate, const ConvArgs& a, hipStream_t stream);

// LDS-DMA conv on v_mfma_f32_16x16x32_f16 (kernels_gl16.hip): the 7- / 11-tap shapes of the staged-input f16x2 convs.
// launch_conv's LDS-DMA branch asks conv_gl16_takes after its own checks (staged input, rows % 64, grid size).
bool conv_gl16_takes(const ConvArgs& a);
int conv_gl16_tile_width(const ConvArgs& a);          // 256 / 192 / 128 time steps per 128-row block
hipError_t launch_conv_gl16(ConvArgs& a, int bn, hipStream_t stream);   // bn = 0: 64-row blocks

// ---------------------------------------------------------------------------------------------------------
// other kernels (kernels_misc.hip)
// ---------------------------------------------------------------------------------------------------------
hipError_t launch_layer_norm(const float* x, float* y, const float* gamma, const float* beta, const int* lengths,
                             int B, int C, int L, bool mask_out, hipStream_t s);
// f16x2: the matrix-core kernel (long sequences) splits its operands into fp16 h + l (three products) instead of three
// bf16 terms (six); range_flag: where it reports an operand beyond the fp16 range (clamped), or null
hipError_t launch_attention(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out,
                            int B, int H, int DK, int L, int window, hipStream_t s, bool f16x2 = false,
                            int* range_flag = nullptr);
hipError_t launch_embed(const int64_t* ids, const float* table, const int* lengths, float* x, int B, int P, int C,
                        int n_vocab, int* err_flag, hipStream_t s);
hipError_t launch_cond_gemv(const float* emb_g, const int64_t* sid, const float* w, const float* bias, float* out,
                            int B, int rows, int gin, int n_speakers, int* err_flag, hipStream_t s);
hipError_t launch_add_cbias_mask(const float* x, const float* cb, int cb_bs, const int* lengths, float* y, int B, int C,
                                 int L, bool mask, hipStream_t s);
// out[b][t] = act(bias + sum_{c,k} w[c*K+k] * pro(x[b][c][t + k - pad]))
hipError_t launch_conv_to1(const float* x, long long x_bs, int x_cs, const float* w, const float* bias, float* out,
                           const int* lengths, int B, int C, int K, int L, int flags, float in_scale, float in_slope,
                           bool tanh_out, bool mask_out, hipStream_t s);
hipError_t launch_prenet_add(float* x, const float* sig, const float* w, const float* bias, int B, int C, int L,
                             hipStream_t s);
hipError_t launch_duration_finalize(const float* logw, const int* lengths, float scale, float* dur, int B, int P,
                                    hipStream_t s);
hipError_t launch_pitch_math(const float* in, bool in_is_hz, float scale, float* lf0, float* f0, int n, hipStream_t s);
hipError_t launch_energy_math(const float* in, bool in_is_raw, float scale, float* norm, float* energy, int n,
                              hipStream_t s);
hipError_t launch_sample(const float* stats, const float* noise, float noise_scale, float* z_p, float* z, int B, int C,
                         int L, hipStream_t s, const int* lengths = nullptr);
// x [B, C, L] fp32 -> staged f16x2 operand format [B][C/8][2 planes][L][8 fp16] of leaky-relu(x * scale, slope)
// the same for the staged bf16 format ([B][C/8][L] units of eight bf16, round to nearest even)
hipError_t launch_stage_bf16(const float* x, float* out, int B, int C, int L, float scale, float slope, hipStream_t s);
// a staged-input plain bf16 conv of this shape at this batch and length runs on the LDS-DMA kernel
bool conv_takes_glb(const ConvDesc& d, int B, int L);
hipError_t launch_stage_f16x2(const float* x, float* out, int B, int C, int L, float scale, float slope, int* range_flag,
                              hipStream_t s);
// length regulator
hipError_t launch_lr_plan(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, hipStream_t s,
                          int* report = nullptr, const int* flag0 = nullptr, const int* flag1 = nullptr);
hipError_t launch_lr_gather(const float* x, const int* cum, const int* totals, float* out, int B, int C, int P, int T,
                            hipStream_t s);

}  // namespace vits
