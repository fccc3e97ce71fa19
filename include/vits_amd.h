/* C ABI of libvits_amd.so: the MI355X (gfx950) implementation of the MVTS synthesis path.
 *
 * The reference (xirui123/VITS) is pure Python and has no FFI; the interface it exposes for this path is the
 * Python method SynthesizerTrn.infer (reference models.py:717-767) on a module built by models.py:582-667 and
 * filled by utils.load_checkpoint (utils.py:19-49).  Each entry point below names the reference code it replaces.
 * The Python class vits_amd.SynthesizerTrn binds these with ctypes and reproduces the reference method's
 * signature, return tuple and error behaviour.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name starts with host_; tensors are fp32,
 * contiguous, channel-first [B, C, L]; phoneme ids and speaker ids are int64; lengths are int32.  All work is
 * enqueued on `stream` (a hipStream_t passed as void*) and nothing synchronises unless stated.  Functions
 * return 0 on success and a negative code on failure; vits_last_error() then holds a message (thread-local).
 * No function allocates device memory: the caller passes a workspace sized by the matching *_workspace_bytes.
 */
#ifndef VITS_AMD_H
#define VITS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITS_OK 0
#define VITS_ERR_INVALID (-1)   /* bad argument / shape                      */
#define VITS_ERR_MISSING (-2)   /* a tensor the model needs was never set    */
#define VITS_ERR_HIP (-3)       /* a HIP call failed                         */
#define VITS_ERR_WORKSPACE (-4) /* workspace too small                       */

typedef struct vits_model vits_model;

/* Hyper-parameters: the constructor arguments of SynthesizerTrn (reference models.py:582-606) plus the
 * dimensions the reference hard-codes (models.py:540,642,644; frame_prior_network.py:65-67; attentions.py:11). */
typedef struct vits_config {
  int n_vocab, hidden_channels, inter_channels, filter_channels, n_heads, n_layers, kernel_size;
  int gin_channels, n_speakers;
  int pitch_layers, dur_filter, energy_filter;
  int n_flows, flow_layers, flow_kernel, window_size;
  int n_upsamples;
  int upsample_rates[8], upsample_kernel_sizes[8], upsample_initial_channel;
  int n_resblock_kernels, resblock_kernel_sizes[8], n_resblock_dilations, resblock_dilation_sizes[8][8];
  int resblock;   /* 1: modules.ResBlock1 (pairs of convs, reference modules.py:183-225); 2: modules.ResBlock2 (single
                   * convs, modules.py:228-252); 0 is read as 1 */
} vits_config;

const char* vits_last_error(void);
const char* vits_version(void);

/* ---- model set-up (replaces module construction + utils.load_checkpoint, reference utils.py:19-49) ---- */
int vits_model_create(const vits_config* cfg, vits_model** out);
/* Releases a handle vits_model_create returned.  A pointer that is not a live handle (destroyed before, or never
 * created by this library) is left alone: nothing is freed and vits_last_error() says so. */
void vits_model_destroy(vits_model* m);
/* 1 if m is a live handle of this library, else 0.  Every entry point that takes a model checks this first. */
int vits_model_alive(const vits_model* m);
/* How many HIP streams vits_generator may use for this model: the caller's, plus n_streams - 1 of the library's own
 * (the generator's independent ResBlock chains, reference models.py:292-298, run side by side on them).  Default:
 * one per ResBlock kernel size.  A caller that drives further streams of its own lowers it to stay within the
 * process's hardware queues (4 by default).  No counterpart in the reference. */
int vits_model_set_generator_streams(vits_model* m, int n_streams);
/* Tells the library that this model's front-end stages (frame prior, flow) are being run beside a generator call on
 * other streams (a pipeline over batches): they then pick kernels that take less of the chip's block slots, which the
 * generator gets back, over kernels with the shortest latency of their own.  Same values either way at the fp32-grade
 * tolerance (the f16x2 packing of the frame-rate convs is used from smaller sizes on).  Default off.  No counterpart
 * in the reference. */
int vits_model_set_front_end_overlapped(vits_model* m, int on);
/* The encoders' post-LN sub-layers (reference attentions.py:32-44, modules.py:15-27): with on != 0 (the default, unless
 * VITS_LN_FUSED=0) a LayerNorm is not a kernel of its own -- the conv that produces x + f(x) writes per-time-step
 * statistics and the convs that consume the normalised tensor apply it while they load; with 0 every LayerNorm is a
 * separate launch.  Same values to rounding (the statistics are merged from 64-channel partial sums). */
int vits_model_set_ln_fused(vits_model* m, int on);
/* Register a device int32 (zeroed by the caller) that the f16x2 kernels OR 1 into whenever they clamp an activation
 * to the fp16 range (+-65504) during vits_generator; NULL switches reporting off.  No counterpart in the reference
 * (its fp32 convs have no such range). */
int vits_model_set_range_flag(vits_model* m, int* dev_flag);
/* Register a plain fp32 tensor (bias, LayerNorm gain, embedding table ...) under `name`. The memory stays owned
 * by the caller and must outlive the model. */
int vits_model_set_tensor(vits_model* m, const char* name, const float* dev_ptr, int64_t numel);
/* Register a convolution whose weight was folded (weight-norm) and packed by vits_pack_conv_weight*.
 * mode: 0 plain, 1 gated (tanh*sigmoid halves), 2/3/4 transposed conv with stride 8/4/2.
 * precision: 0 = fp32 MFMA on fp32-packed weights; 1 = three-term bf16 split ("bf16x3": both operands are split
 *   exactly into three bf16 terms, six cross products on the bf16 MFMA, fp32 accumulate; fp32-grade results) on
 *   weights packed by vits_pack_conv_weight_bf16x3; 3 = two-term fp16 split ("f16x2": x = h + l in fp16, three cross
 *   products on the fp16 MFMA, fp32 accumulate; about 2^-22 relative error per product, operands must lie inside the
 *   fp16 range, activations are clamped to +-65504) on weights packed by vits_pack_conv_weight_f16x2; 2 = plain bf16 (both operands rounded to nearest-even bf16, one
 *   product per multiply-accumulate, fp32 accumulate; about 8e-3 waveform max-abs against fp32 when used for the
 *   generator) on weights packed by vits_pack_conv_weight_bf16.
 * rows = packed GEMM rows (cout, or cout*stride for a transposed conv); bias has `rows` entries or is NULL. */
int vits_model_set_conv(vits_model* m, const char* name, const void* packed_w, const float* bias_rows, int cin,
                        int rows, int taps, int dilation, int mode, int precision);
/* A second packing of a bf16x3 conv1d's weights in the f16x2 format (same rows / cin / taps; vits_model_set_conv of
 * that name comes first): launches of that conv too large for the short-sequence kernel -- frame-rate convs of long
 * batches (the frame-prior encoder, reference models.py:497-501, the projection, :569-572, the flow's WaveNets,
 * modules.py:144-172) -- then run in f16x2 arithmetic, three fp16 products where bf16x3 takes six bf16 ones.  Optional:
 * without it such a conv runs in bf16x3 at every size. */
int vits_model_set_conv_alt(vits_model* m, const char* name, const void* packed_w_f16x2);
/* When a conv with such a second packing takes it (process-wide; no counterpart in the reference, a tuning knob):
 * 0 never, 1 (default) launches too large for the short-sequence kernel, 2 always (tests, measurements). */
void vits_set_front_end_f16x2(int mode);
/* Resolve every name the synthesis path needs; fails with VITS_ERR_MISSING naming the first absent tensor. */
int vits_model_finalize(vits_model* m);

/* Host-side packing of one conv weight into MFMA fragment order.  host_w is the folded weight laid out as
 * [rows][cin][taps] (row order for transposed convs: see vits_amd/weights.py); host_out gets rows*cin*taps floats. */
int vits_pack_conv_weight(const float* host_w, int rows, int cin, int taps, float* host_out);
/* Same for the bf16x3 mode: host_out gets 3*rows*cin*taps uint16 (bf16 bit patterns). */
int vits_pack_conv_weight_bf16x3(const float* host_w, int rows, int cin, int taps, uint16_t* host_out);
/* Same for the f16x2 mode: host_out gets 2*rows*cin*taps uint16 (fp16 bit patterns); fails if a weight is outside
 * the fp16 range. */
int vits_pack_conv_weight_f16x2(const float* host_w, int rows, int cin, int taps, uint16_t* host_out);
/* Same for the plain bf16 mode: host_out gets rows*cin*taps uint16 (bf16 bit patterns, round to nearest even). */
int vits_pack_conv_weight_bf16(const float* host_w, int rows, int cin, int taps, uint16_t* host_out);

/* ---- stages of SynthesizerTrn.infer ---- */

/* Conditioning vectors from the speaker embedding: g = emb_g(sid) (reference models.py:719-720) pushed through
 * every cond 1x1 conv (models.py:129,287,551; modules.py:149; frame_prior_network.py:126).  cond: [B, cond_rows].
 * A speaker-less model (n_speakers == 0: the reference runs with g = None, models.py:721-722, and applies no
 * conditioning conv) has no such call -- vits_speaker_cond fails for it -- and every stage takes an all-zero cond;
 * its energy must then be given (vits_variance_adaptor with energy_is_tensor), as in the reference, whose energy
 * predictor cannot run without g (frame_prior_network.py:125). */
int vits_cond_rows(const vits_model* m);
int vits_speaker_cond(const vits_model* m, const int64_t* sid, int B, float* cond, int* err_flag, void* stream);

/* TextEncoder.forward (reference models.py:175-181): x [B, C, P]. */
int64_t vits_encoder_workspace_bytes(const vits_model* m, int B, int L);
int vits_text_encoder(const vits_model* m, const int64_t* phonemes, const int* lengths, int B, int P, float* x,
                      int* err_flag, void* ws, int64_t ws_bytes, void* stream);

/* attentions.Encoder.forward (reference attentions.py:32-44) in place on x [B, C, L]; which: 0 enc_p.encoder,
 * 1 pitch_predictor.pitch_net, 2 frame_prior_net.fft_block. */
int vits_encoder_stack(const vits_model* m, int which, float* x, const int* lengths, int B, int L, void* ws,
                       int64_t ws_bytes, void* stream);

/* Variance adaptor (reference models.py:726-753): duration / pitch / energy prediction or override and the two
 * prenet updates of x (in place).  *_mode: 0 = predict and scale by *_scale, 1 = per-phoneme tensor override
 * (pitch in Hz, energy in raw units; duration override is handled by the caller and skips the predictor).
 * Outputs: dur_out [B, P] float (only when dur_mode == 0), f0_out, energy_out [B, P]. */
int64_t vits_variance_workspace_bytes(const vits_model* m, int B, int P);
int vits_variance_adaptor(const vits_model* m, float* x, const int* lengths, const float* cond, int B, int P,
                          int dur_mode, float dur_scale, float* dur_out, int pitch_mode, float pitch_scale,
                          const float* pitch_hz, float* f0_out, int energy_mode, float energy_scale,
                          const float* energy_raw, float* energy_out, void* ws, int64_t ws_bytes, void* stream);

/* LengthRegulator (reference models.py:418-460).  plan: integer durations -> inclusive prefix sums cum [B, P] and
 * frame counts totals [B] (dur_dtype: 0 f32, 1 i64, 2 i32, 3 f64).  The caller reads totals to size x_frame, then
 * gather writes x_frame [B, C, T] (zero beyond each row's frame count). */
int vits_length_regulate_plan(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, void* stream);
/* The same, and report [B + 2] receives the frame counts again followed by the device words flag0 and flag1 point to
 * (0 for NULL) as they stand when the plan runs: the caller's one device-to-host copy then covers the frame counts and
 * its two status words (id-range errors of the embedding stages, the fp16-range flag) without a gather of its own. */
int vits_length_regulate_plan_report(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, int* report,
                                     const int* flag0, const int* flag1, void* stream);
int vits_length_regulate(const float* x, const int* cum, const int* totals, int B, int C, int P, int T, float* x_frame,
                         void* stream);

/* FramePriorNet + Projection + sampling (reference models.py:760-763): x_frame [B, C, T] is transformed in place;
 * stats [B, 2C, T] receives (m_p, logs_p); z_p = m_p + noise * exp(logs_p) * noise_scale; z gets a copy of z_p. */
int vits_frame_prior(const vits_model* m, float* x_frame, const int* frame_lengths, int B, int T, const float* noise,
                     float noise_scale, float* stats, float* z_p, float* z, void* ws, int64_t ws_bytes, void* stream);

/* ResidualCouplingBlock.forward(reverse=True) (reference models.py:209-216, modules.py:320-339) in place on z. */
int64_t vits_flow_workspace_bytes(const vits_model* m, int B, int T);
int vits_flow_reverse(const vits_model* m, float* z, const int* frame_lengths, const float* cond, int B, int T,
                      void* ws, int64_t ws_bytes, void* stream);

/* vits_flow_reverse for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the same kernels and results, with the three work tensors in a fixed
 * 18 MiB array inside the library's device image instead of a caller workspace.  It has no counterpart in the
 * reference.  Limit: B * T <= vits_flow_noalloc_max_positions() (8192) and hidden_channels <= 192; beyond it the call
 * fails with VITS_ERR_WORKSPACE.  It calls no allocating routine: the array's address on a GPU is looked up by
 * vits_model_finalize (with that GPU current; the first look-up on a device loads the device image, which allocates),
 * so a model must have been finalized on the GPU the call runs on -- on any other GPU of the process the first call
 * does that look-up itself and is not safe under capture.  Every call on one GPU uses that GPU's single copy of the
 * array: calls that overlap on one GPU (different streams or threads) race. */
int64_t vits_flow_noalloc_max_positions(void);
int vits_flow_reverse_noalloc(const vits_model* m, float* z, const int* frame_lengths, const float* cond, int B, int T,
                              void* stream);

/* ResidualCouplingBlock.forward(reverse=False) (reference models.py:209-212, modules.py:320-335) in place on z: the
 * z -> z_p direction used by voice conversion (models.py:774).  Same workspace as vits_flow_reverse. */
int vits_flow_forward(const vits_model* m, float* z, const int* frame_lengths, const float* cond, int B, int T,
                      void* ws, int64_t ws_bytes, void* stream);

/* PosteriorEncoder.forward (reference models.py:243-250), available when the enc_q.* convs were registered
 * ("enc_q.pre", "enc_q.in.N" gated, "enc_q.res.N", "enc_q.skip.N", "enc_q.proj"; their conditioning rows appended to
 * cond.w / cond.b).  spec [B, vits_posterior_channels(m), T]: the linear spectrogram with its channel count rounded
 * up to a multiple of 16, the added channels zero (the packed enc_q.pre weight is padded the same way).
 * stats [B, 2*inter, T] receives (m_q, logs_q); z = (m_q + noise * exp(logs_q)) * mask. */
int vits_posterior_channels(const vits_model* m);          /* -1 when the model has no posterior encoder */
int64_t vits_posterior_workspace_bytes(const vits_model* m, int B, int T);
int vits_posterior_encoder(const vits_model* m, const float* spec, const int* frame_lengths, const float* cond, int B,
                           int T, const float* noise, float* stats, float* z, void* ws, int64_t ws_bytes, void* stream);

/* Generator.forward (reference models.py:284-303) on (z * mask)[:, :, :t_dec]: z has row stride T;
 * wav [B, 1, t_dec * prod(upsample_rates)]. */
int64_t vits_generator_workspace_bytes(const vits_model* m, int B, int t_dec);
int vits_generator(const vits_model* m, const float* z, const int* frame_lengths, const float* cond, int B, int T,
                   int t_dec, float* wav, void* ws, int64_t ws_bytes, void* stream);

/* ---- per-kernel timing of the conv launches (HIP events on the launch stream; off by default).  No counterpart
 * in the reference.  vits_profile_read(-1, ...) resolves pending events (synchronises) and returns the number of
 * rows; vits_profile_read(i, ...) returns row i: kernel instantiation name, summed algorithmic FLOPs
 * (2 * rows * cin * taps * L * B per launch), summed milliseconds and launch count since the last reset.
 * While timing is enabled vits_generator keeps all its launches on the caller's stream (no side streams), so the
 * per-kernel intervals do not overlap. */
void vits_profile_enable(int on);
void vits_profile_reset(void);
int vits_profile_read(int index, char* name, int name_len, double* flops, double* ms, int64_t* count);
/* Algorithmic HBM bytes summed over the launches of row `index` (same row order as vits_profile_read): every input
 * and output element once in fp32, residual / accumulated-output reads where the epilogue has them, packed weights. */
int vits_profile_read_bytes(int index, double* bytes);

/* ---- single operators, exposed for unit parity tests ----
 * vits_op_conv1d flags (OR of): 1 leaky-relu(in_slope) on the input, 2 input mask, 4 relu, 8 mask before residual,
 * 16 += res, 32 res - result, 64 += out, 128 mask last, 256 write the result in the staged operand format of the next
 * conv (leaky-relu(in_slope) applied, converted; plain convs in precision 2 / 3 only), 512 the input is in that format.
 * Staged format, per batch row: 16-byte units of eight channels of one time step, precision 2: bf16 [C/8][L][8];
 * precision 3: fp16 [C/8][2 planes h, l][L][8]. */
/* Device int32 the f16x2 single operators below OR 1 into when they clamp an activation to the fp16 range (thread-local
 * setting; NULL, the default, switches it off). */
void vits_op_set_range_flag(int* dev_flag);
int vits_op_conv1d(const void* packed_w, const float* bias_rows, int cin, int rows, int taps, int dilation, int mode,
                   int precision, const float* x, int B, int L, const int* lengths, const float* cbias,
                   const float* in_cbias, const float* res, float* out, int flags, float in_scale, float in_slope,
                   void* stream);
/* vits_op_conv1d without masks and per-batch biases, with a second copy of the result in the staged operand format
 * (leaky-relu(out_slope) applied, then converted) in out2: what the generator's stages hand to the ResBlock convs
 * (reference models.py:290-298: x = ups[i](x), then every ResBlock starts with leaky_relu(x, 0.1)).  Precision 2 / 3
 * only; plain convs with flag 512 (staged input), and the x8 / x4 transposed convs (mode 2 / 3), whose out2 covers the
 * L * 8 or L * 4 output steps.  out2 may be NULL (then this is vits_op_conv1d). */
int vits_op_conv1d_out2(const void* packed_w, const float* bias_rows, int cin, int rows, int taps, int dilation, int mode,
                        int precision, const float* x, int B, int L, const float* res, float* out, float* out2, int flags,
                        float in_scale, float in_slope, float out_slope, void* stream);
/* One ResBlock1 pair (reference modules.py:206-219, one iteration of its loop) in a single launch:
 * out = x + conv2(lrelu(conv1(lrelu(x)))) (+ out when accumulate), conv1 dilated, conv2 dilation 1, both with `taps`
 * taps (3, 5, 7, 9 or 11; (taps - 1) / 2 * dilation <= 25) and f16x2-packed weights [C][C][taps]; C = 32 or 64;
 * x and out [B, C, L], not overlapping.  Anything else is VITS_ERR_INVALID. */
int vits_op_resblock_pair(const void* packed_w1, const float* bias1, const void* packed_w2, const float* bias2,
                          int channels, int taps, int dilation, const float* x, float* out, int B, int L,
                          int accumulate, float slope, void* stream);
/* The same pair with bf16-packed weights and bf16 operands (precision 2; precision 3 is vits_op_resblock_pair): what
 * the plain-bf16 generator mode launches for its 32- and 64-channel stages. */
int vits_op_resblock_pair_prec(const void* packed_w1, const float* bias1, const void* packed_w2, const float* bias2,
                               int channels, int taps, int dilation, int precision, const float* x, float* out, int B,
                               int L, int accumulate, float slope, void* stream);
/* A whole ResBlock1 (reference modules.py:206-219: every iteration of its loop) in a single launch: for each of the
 * npair (1..4) dilations x = x + conv2(lrelu(conv1(lrelu(x)))), then out = x (+ out when accumulate: the sum of a
 * stage's ResBlocks, models.py:292-298).  packed_w1 / bias1 / packed_w2 / bias2: host arrays of npair device pointers
 * (weights packed for `precision`, 3 = f16x2 or 2 = bf16; a bias array or an entry of it may be NULL); C = 32 or 64,
 * or 128 with 3 taps (the same bytes as the convs called one by one there: the pair ops stop at 64 channels);
 * the same tap count for every conv; the halos of the pairs, (taps - 1) / 2 * (dilation + 1) each, add up to at most
 * 64 steps.  Same bytes as npair calls of vits_op_resblock_pair(_prec).  x and out [B, C, L], not overlapping. */
int vits_op_resblock_chain(const void* const* packed_w1, const float* const* bias1, const void* const* packed_w2,
                           const float* const* bias2, int channels, int taps, const int* dilations, int npair,
                           int precision, const float* x, float* out, int B, int L, int accumulate, float slope,
                           void* stream);
int vits_op_layer_norm(const float* x, float* y, const float* gamma, const float* beta, const int* lengths, int B, int C,
                       int L, int mask_out, void* stream);
int vits_op_attention(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out, int B,
                      int H, int DK, int L, int window, void* stream);
/* The same with the arithmetic of the matrix-core kernel (rows of 256 steps and more) chosen: precision 1 = three bf16
 * terms per operand (what vits_op_attention runs), 3 = fp16 h + l, what the frame-rate encoder runs beside its f16x2
 * convs (reference attentions.py:216-257 at frame rate, frame_prior_network.py).  Shorter rows run on the vector-ALU
 * kernel in fp32 either way. */
int vits_op_attention_prec(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out, int B,
                           int H, int DK, int L, int window, int precision, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITS_AMD_H */
