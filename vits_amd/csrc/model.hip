// Host side of libvits_amd.so: model registry, stage orchestration and the C ABI declared in include/vits_amd.h.
// Each stage function enqueues hand-written kernels (kernels_conv.hip, kernels_misc.hip) on the caller's stream and
// allocates nothing; activations live in the caller's workspace.
#include "common.h"
#include "../../include/vits_amd.h"

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace vits;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// Every handle vits_model_create has returned and vits_model_destroy has not yet released.  The C ABI hands out raw
// pointers; a caller that destroys one twice, or passes a pointer the library never made (a copied or stale handle),
// must get an error, not a double free.
std::mutex g_models_mu;
std::set<const vits_model*> g_models;

bool model_alive(const vits_model* m) {
  if (!m) return false;
  std::lock_guard<std::mutex> lk(g_models_mu);
  return g_models.count(m) != 0;
}

// Device word the f16x2 kernels OR 1 into when an activation leaves the fp16 range (set per entry-point call from the
// model; read by conv_simple and the pair launcher on the same host thread).
thread_local int* g_range_flag = nullptr;
struct RangeFlagScope {
  int* prev;
  explicit RangeFlagScope(int* f) : prev(g_range_flag) { g_range_flag = f; }
  ~RangeFlagScope() { g_range_flag = prev; }
};
// The same for "this front-end stage runs beside a generator on other streams" (launch_conv's choice of kernel).
struct OverlapScope {
  bool prev;
  explicit OverlapScope(bool on) : prev(front_end_overlapped()) { set_front_end_overlapped(on); }
  ~OverlapScope() { set_front_end_overlapped(prev); }
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(VITS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

struct Plain {
  const float* p = nullptr;
  int64_t n = 0;
};

struct EncLayer {
  ConvDesc qkv, o, ffn1, ffn2;
  const float *ek, *ev, *ln1g, *ln1b, *ln2g, *ln2b;
};

struct FlowLayer {
  ConvDesc pre, post;
  std::vector<ConvDesc> in, res, skip;
  std::vector<ConvDesc> rs;      // res and skip halves as one conv of 2C rows (when registered): one launch per layer
};

struct ResBlock {
  std::vector<ConvDesc> c1, c2;
};

}  // namespace

// VITS_LN_FUSED=0: every LayerNorm of the encoders is a kernel of its own (the default of vits_model_set_ln_fused).
static bool ln_fused_default() {
  static const bool on = [] {
    const char* e = getenv("VITS_LN_FUSED");
    return !(e && e[0] == '0');
  }();
  return on;
}

struct vits_model {
  vits_config cfg{};
  std::map<std::string, Plain> plain;
  std::map<std::string, ConvDesc> convs;
  bool finalized = false;
  // resolved
  std::vector<EncLayer> enc[3];
  const float *symbol_emb = nullptr, *emb_g = nullptr, *cond_w = nullptr, *cond_b = nullptr;
  ConvDesc dp_conv1, dp_conv2, en_conv1, en_conv2, project, conv_pre;
  const float *dp_ln1g, *dp_ln1b, *dp_ln2g, *dp_ln2b, *dp_proj_w, *dp_proj_b;
  const float *pitch_proj_w, *pitch_proj_b, *pitch_pre_w, *pitch_pre_b;
  const float *en_ln1g, *en_ln1b, *en_ln2g, *en_ln2b, *en_lin_w, *en_lin_b, *en_pre_w, *en_pre_b;
  std::vector<FlowLayer> flows;
  std::vector<ConvDesc> ups;
  std::vector<ResBlock> rbs;
  const float* conv_post_w = nullptr;
  int cond_rows = 0, off_dp = 0, off_pitch = 0, off_energy = 0, off_dec = 0, off_flow = 0, off_encq = 0;
  // posterior encoder (voice conversion only; present when the checkpoint's enc_q.* convs were registered)
  bool has_encq = false;
  int encq_cin = 0;                  // spectrogram channels rounded up to a multiple of 16
  ConvDesc encq_pre, encq_proj;
  FlowLayer encq;                    // its WN: in / res / skip lists (pre and post unused)
  int total_up = 1;
  // side streams + events so that the generator's independent ResBlock chains can overlap (one chain's
  // memory-bound phases run under another chain's MFMA phases); created at finalize on the then-current device
  hipStream_t side[8] = {};
  hipEvent_t ev_fork = nullptr, ev_done[8] = {};
  int n_side = 0;
  int side_limit = 8;                // vits_model_set_generator_streams: how many of them vits_generator may use
  bool ln_fused = ln_fused_default();   // vits_model_set_ln_fused: the encoders' LayerNorms ride in the neighbouring convs
  int* range_flag = nullptr;         // vits_model_set_range_flag
  bool beside_generator = false;     // vits_model_set_front_end_overlapped
};

namespace {

const char* kEncNames[3] = {"enc_p.encoder", "pitch_predictor.pitch_net", "frame_prior_net.fft_block"};

struct Arena {
  char* base;
  int64_t size, used = 0;
  Arena(void* p, int64_t n) : base((char*)p), size(n) {}
  float* take(int64_t floats) {
    const int64_t bytes = (floats * 4 + 255) & ~int64_t(255);
    if (used + bytes > size) return nullptr;
    float* r = (float*)(base + used);
    used += bytes;
    return r;
  }
};

inline int64_t pad256(int64_t floats) { return (floats * 4 + 255) & ~int64_t(255); }

int make_desc(ConvDesc& d, const void* w, const float* bias, int cin, int rows, int taps, int dilation, int mode,
              int prec = PREC_F32) {
  if (cin <= 0 || cin % CI_CHUNK || rows <= 0 || rows % 32 || taps < 1 || taps > MAX_TAPS || dilation < 1)
    return fail(VITS_ERR_INVALID, "conv: cin must be a multiple of 16, rows of 32, 1 <= taps <= 12");
  d = ConvDesc();
  d.w = w;
  d.bias = bias;
  d.Cin = cin;
  d.M = rows;
  d.K = taps;
  d.mode = mode;
  if (prec != PREC_F32 && prec != PREC_BF16X3 && prec != PREC_BF16 && prec != PREC_F16X2) return fail(VITS_ERR_INVALID, "conv: unknown precision");
  d.prec = prec;
  switch (mode) {
    case MODE_PLAIN:
    case MODE_GATE:
      if (mode == MODE_GATE && rows % 64) return fail(VITS_ERR_INVALID, "gated conv: rows must be a multiple of 64");
      d.toff0[0] = d.toff0[1] = -((taps - 1) / 2) * dilation;
      d.tstep[0] = d.tstep[1] = dilation;
      d.min_off = d.toff0[0];
      d.max_off = d.toff0[0] + (taps - 1) * dilation;
      d.up = 1;
      d.cout = mode == MODE_GATE ? rows / 2 : rows;
      break;
    case MODE_UP8:
    case MODE_UP2:
      // stride u, kernel 2u, padding u/2: phase r < u/2 reads x[t], x[t-1]; phase r >= u/2 reads x[t+1], x[t]
      if (taps != 2 || rows % 64) return fail(VITS_ERR_INVALID, "transposed conv (stride 8 / 2): taps must be 2, rows % 64 == 0");
      d.toff0[0] = 0; d.tstep[0] = -1;
      d.toff0[1] = 1; d.tstep[1] = -1;
      d.min_off = -1; d.max_off = 1;
      d.up = mode == MODE_UP8 ? 8 : 2;
      d.cout = rows / d.up;
      break;
    case MODE_UP4:
      if (taps != 1 || rows % 64) return fail(VITS_ERR_INVALID, "transposed conv (stride 4): taps must be 1, rows % 64 == 0");
      d.up = 4;
      d.cout = rows / 4;
      break;
    default:
      return fail(VITS_ERR_INVALID, "conv: unknown mode");
  }
  if (d.max_off - d.min_off > MAX_HALO) return fail(VITS_ERR_INVALID, "conv: receptive field wider than 64 samples");
  return VITS_OK;
}

ConvArgs base_args(const float* x, long long x_bs, int x_cs, float* out, long long out_bs, int out_cs, int B, int L,
                   const int* lengths, int flags) {
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.x_bs = x_bs; a.x_cs = x_cs;
  a.out = out; a.out_bs = out_bs; a.out_cs = out_cs;
  a.B = B; a.L = L; a.lengths = lengths; a.flags = flags;
  a.in_scale = 1.f; a.in_slope = 1.f;
  a.range_flag = g_range_flag;
  return a;
}

// contiguous [B, Cin, L] -> [B, cout, L*up]
// floats per batch row of a staged activation tensor with C channels (bf16: 2 bytes per element, f16x2: 4)
inline long long staged_row_floats(int prec, int C, int L) { return prec == PREC_BF16 ? (long long)C * L / 2 : (long long)C * L; }

ConvArgs conv_simple_args(const ConvDesc& d, const float* x, float* out, int B, int L, const int* lengths, int flags,
                          const float* res = nullptr, const float* cbias = nullptr, int cbias_bs = 0,
                          const float* in_cbias = nullptr, int in_cbias_bs = 0, float in_scale = 1.f, float in_slope = 1.f,
                          float out_slope = 1.f, float* out2 = nullptr) {
  ConvArgs a = base_args(x, (long long)d.Cin * L, L, out, (long long)d.cout * L * d.up, L * d.up, B, L, lengths, flags);
  a.out2 = out2; a.out2_bs = staged_row_floats(d.prec, d.cout, L * d.up);   // staged copy: f16x2 as many bytes as fp32, bf16 half
  if (flags & F_IN_STAGED) a.x_bs = staged_row_floats(d.prec, d.Cin, L);
  if (flags & F_OUT_STAGED) a.out_bs = staged_row_floats(d.prec, d.cout, L);
  a.out_slope = out_slope;
  a.res = res; a.res_bs = (long long)d.cout * L; a.res_cs = L;
  a.cbias = cbias; a.cbias_bs = cbias_bs;
  a.in_cbias = in_cbias; a.in_cbias_bs = in_cbias_bs;
  a.in_scale = in_scale; a.in_slope = in_slope;
  a.range_flag = g_range_flag;
  return a;
}

int conv_simple(const ConvDesc& d, const float* x, float* out, int B, int L, const int* lengths, int flags,
                hipStream_t s, const float* res = nullptr, const float* cbias = nullptr, int cbias_bs = 0,
                const float* in_cbias = nullptr, int in_cbias_bs = 0, float in_scale = 1.f, float in_slope = 1.f,
                float out_slope = 1.f, float* out2 = nullptr) {
  HIP_TRY(launch_conv(d, conv_simple_args(d, x, out, B, L, lengths, flags, res, cbias, cbias_bs, in_cbias, in_cbias_bs,
                                          in_scale, in_slope, out_slope, out2), s));
  return VITS_OK;
}

#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != VITS_OK) return rc_; \
  } while (0)

int64_t encoder_ws_floats(const vits_model* m, int B, int L) {
  const int64_t C = m->cfg.hidden_channels, F = m->cfg.filter_channels;
  // q|k|v, attention output, FFN hidden, and two tables of LayerNorm statistics (per time step and 64-channel unit)
  return (pad256((int64_t)B * 3 * C * L) + pad256((int64_t)B * C * L) + pad256((int64_t)B * F * L) +
          2 * pad256((int64_t)B * L * ((C + 63) / 64) * 2)) / 4;
}

int run_encoder(const vits_model* m, int which, float* x, const int* lengths, int B, int L, Arena& ar, hipStream_t s) {
  const int C = m->cfg.hidden_channels, F = m->cfg.filter_channels, H = m->cfg.n_heads;
  float* qkv = ar.take((int64_t)B * 3 * C * L);
  float* att = ar.take((int64_t)B * C * L);
  float* h = ar.take((int64_t)B * F * L);
  if (!qkv || !att || !h) return fail(VITS_ERR_WORKSPACE, "encoder: workspace too small");
  const auto& layers = m->enc[which];
  // Post-LN sub-layers, x = norm(x + f(x)) (reference attentions.py:32-44).  Deferred form: the conv that produces
  // x + f(x) leaves it un-normalised and writes per-time-step statistics; the convs that read it (q|k|v, the FFN's first
  // conv) normalise at staging and the next residual add normalises in its epilogue, so only the encoder's last
  // LayerNorm is a kernel.  Needs whole 64-channel units and every conv of the stack on the split-K kernel (short
  // grids: phoneme rate, and frame rate at small batches), where a consumer pays ~2 us for it.
  bool deferred = m->ln_fused && C % 64 == 0 && !layers.empty();
  for (const EncLayer& l : layers)
    deferred = deferred && conv_takes_split_k(l.qkv, B, L) && conv_takes_split_k(l.o, B, L) &&
               conv_takes_split_k(l.ffn1, B, L) && conv_takes_split_k(l.ffn2, B, L);
  float* st[2] = {nullptr, nullptr};
  if (deferred) {
    st[0] = ar.take((int64_t)B * L * (C / 64) * 2);
    st[1] = ar.take((int64_t)B * L * (C / 64) * 2);
    if (!st[0] || !st[1]) return fail(VITS_ERR_WORKSPACE, "encoder: workspace too small");
  }
  const float *pg = nullptr, *pb = nullptr, *ps = nullptr;      // the pending LayerNorm of x: gain, bias, statistics
  for (size_t i = 0; i < layers.size(); ++i) {
    const EncLayer& l = layers[i];
    // an encoder whose convs have the f16x2 packing (frame rate) runs its attention in the same arithmetic
    const bool att_f16x2 = l.qkv.w_alt && l.o.w_alt && alt_f16x2_mode() > 0;
    if (!deferred) {
      TRY(conv_simple(l.qkv, x, qkv, B, L, lengths, 0, s));
      HIP_TRY(launch_attention(qkv, l.ek, l.ev, lengths, att, B, H, C / H, L, m->cfg.window_size, s, att_f16x2, g_range_flag));
      TRY(conv_simple(l.o, att, x, B, L, lengths, F_RES, s, x));
      HIP_TRY(launch_layer_norm(x, x, l.ln1g, l.ln1b, lengths, B, C, L, false, s));
      TRY(conv_simple(l.ffn1, x, h, B, L, lengths, F_IN_MASK | F_RELU, s));
      TRY(conv_simple(l.ffn2, h, x, B, L, lengths, F_IN_MASK | F_MASK_PRE | F_RES, s, x));
      HIP_TRY(launch_layer_norm(x, x, l.ln2g, l.ln2b, lengths, B, C, L, i + 1 == layers.size(), s));
      continue;
    }
    const bool last = i + 1 == layers.size();
    {
      ConvArgs a = conv_simple_args(l.qkv, x, qkv, B, L, lengths, 0);
      a.ln_in_stats = ps; a.ln_in_gamma = pg; a.ln_in_beta = pb;
      HIP_TRY(launch_conv(l.qkv, a, s));
    }
    HIP_TRY(launch_attention(qkv, l.ek, l.ev, lengths, att, B, H, C / H, L, m->cfg.window_size, s, att_f16x2, g_range_flag));
    {
      ConvArgs a = conv_simple_args(l.o, att, x, B, L, lengths, F_RES, x);
      a.ln_res_stats = ps; a.ln_res_gamma = pg; a.ln_res_beta = pb;
      a.ln_out_stats = st[0];
      HIP_TRY(launch_conv(l.o, a, s));
    }
    {
      ConvArgs a = conv_simple_args(l.ffn1, x, h, B, L, lengths, F_IN_MASK | F_RELU);
      a.ln_in_stats = st[0]; a.ln_in_gamma = l.ln1g; a.ln_in_beta = l.ln1b;
      HIP_TRY(launch_conv(l.ffn1, a, s));
    }
    {
      ConvArgs a = conv_simple_args(l.ffn2, h, x, B, L, lengths, F_IN_MASK | F_MASK_PRE | F_RES, x);
      a.ln_res_stats = st[0]; a.ln_res_gamma = l.ln1g; a.ln_res_beta = l.ln1b;
      a.ln_out_stats = last ? nullptr : st[1];
      HIP_TRY(launch_conv(l.ffn2, a, s));
    }
    if (last) HIP_TRY(launch_layer_norm(x, x, l.ln2g, l.ln2b, lengths, B, C, L, true, s));
    pg = l.ln2g; pb = l.ln2b; ps = st[1];
  }
  return VITS_OK;
}

const float* need_plain(vits_model* m, const std::string& name, int64_t numel, std::string& missing) {
  auto it = m->plain.find(name);
  if (it == m->plain.end() || (numel > 0 && it->second.n != numel)) {
    if (missing.empty()) missing = name + (it == m->plain.end() ? " (not set)" : " (wrong size)");
    return nullptr;
  }
  return it->second.p;
}

ConvDesc need_conv(vits_model* m, const std::string& name, int cin, int rows, int taps, int mode, std::string& missing) {
  auto it = m->convs.find(name);
  if (it == m->convs.end()) {
    if (missing.empty()) missing = name + " (conv not set)";
    return ConvDesc();
  }
  const ConvDesc& d = it->second;
  if (d.Cin != cin || d.M != rows || d.K != taps || d.mode != mode) {
    if (missing.empty()) {
      char buf[256];
      snprintf(buf, sizeof buf, "%s (conv shape: got cin %d rows %d taps %d mode %d, need %d %d %d %d)", name.c_str(),
               d.Cin, d.M, d.K, d.mode, cin, rows, taps, mode);
      missing = buf;
    }
    return ConvDesc();
  }
  return d;
}

}  // namespace

extern "C" {

const char* vits_last_error(void) { return g_err.c_str(); }
const char* vits_version(void) {
  return "vits_amd 0.2 (gfx950; MFMA convs: f16x2 / bf16x3 / fp32 / bf16, fp32 accumulate)";
}

int vits_model_create(const vits_config* cfg, vits_model** out) {
  if (!cfg || !out) return fail(VITS_ERR_INVALID, "null argument");
  const vits_config& c = *cfg;
  if (c.hidden_channels != c.inter_channels) return fail(VITS_ERR_INVALID, "hidden_channels must equal inter_channels");
  if (c.hidden_channels % 32 || c.filter_channels % 32 || c.dur_filter % 32 || c.energy_filter % 32)
    return fail(VITS_ERR_INVALID, "channel counts must be multiples of 32");
  if (c.n_heads <= 0 || c.hidden_channels / c.n_heads != 96 || c.window_size != 4)
    return fail(VITS_ERR_INVALID, "attention kernel is built for head dim 96 and window 4");
  if (c.n_flows <= 0 || c.n_flows % 2) return fail(VITS_ERR_INVALID, "n_flows must be even (channel flips are folded into weights)");
  if ((c.inter_channels / 2) % CI_CHUNK) return fail(VITS_ERR_INVALID, "inter_channels / 2 must be a multiple of 16");
  if (c.n_upsamples <= 0 || c.n_upsamples > 8 || c.n_resblock_kernels <= 0 || c.n_resblock_kernels > 8 ||
      c.n_resblock_dilations <= 0 || c.n_resblock_dilations > 8 || c.resblock < 0 || c.resblock > 2)
    return fail(VITS_ERR_INVALID, "generator: bad upsample / resblock counts or resblock type");
  // n_speakers == 0: speaker-less (no conditioning anywhere; gin_channels is then irrelevant).  n_speakers == 1 is a
  // configuration the reference itself cannot run (it creates no embedding table for it, models.py:660, and looks a
  // speaker up all the same, :719)
  if (c.n_speakers < 0 || c.n_speakers == 1 || (c.n_speakers > 1 && c.gin_channels <= 0))
    return fail(VITS_ERR_INVALID, "n_speakers must be 0 (speaker-less) or > 1 with gin_channels > 0");
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    if (!((u == 8 && k == 16) || (u == 4 && k == 4) || (u == 2 && k == 4)))
      return fail(VITS_ERR_INVALID, "generator: supported (stride, kernel) pairs are (8,16), (4,4), (2,4)");
    if ((c.upsample_initial_channel >> (i + 1)) % 32) return fail(VITS_ERR_INVALID, "generator: channels must stay multiples of 32");
  }
  vits_model* m = new vits_model();
  m->cfg = c;
  {
    std::lock_guard<std::mutex> lk(g_models_mu);
    g_models.insert(m);
  }
  *out = m;
  return VITS_OK;
}

// The side streams the generator's ResBlock chains fork onto belong to the process, two per device, shared by every
// model: a HIP process schedules four hardware queues at once, and side streams of a model's own would add two more
// per model (a second model in the process ran its pipeline 6.7 % slower for them: 82.1 against 77.0 ms a step at
// 64 x 10 s).  Models that run at the same time on one device then share them, which serialises their chains and keeps
// them correct (every model waits on events of its own).  They live until the process ends.
namespace {
constexpr int kSideDevices = 64, kSideStreams = 2;
std::mutex g_side_mu;
hipStream_t g_side[kSideDevices][kSideStreams] = {};
int shared_side_stream(int i, hipStream_t* out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kSideDevices || i < 0 || i >= kSideStreams) return fail(VITS_ERR_INVALID, "side stream: device or index out of range");
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (!g_side[dev][i]) HIP_TRY(hipStreamCreateWithFlags(&g_side[dev][i], hipStreamNonBlocking));
  *out = g_side[dev][i];
  return VITS_OK;
}
}  // namespace

static void release_streams(vits_model* m) {
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  for (auto& e : m->ev_done) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  m->ev_fork = nullptr;
  m->n_side = 0;
}

void vits_model_destroy(vits_model* m) {
  if (!m) return;
  {
    // only a live handle of this library is released: a second destroy of the same pointer, or a pointer that
    // vits_model_create never returned, is ignored (and reported through vits_last_error)
    std::lock_guard<std::mutex> lk(g_models_mu);
    if (g_models.erase(m) == 0) {
      g_err = "vits_model_destroy: not a live model handle (already destroyed, or not created by this library)";
      return;
    }
  }
  release_streams(m);
  delete m;
}

int vits_model_alive(const vits_model* m) { return model_alive(m) ? 1 : 0; }

int vits_model_set_generator_streams(vits_model* m, int n_streams) {
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "set_generator_streams: not a live model handle");
  if (n_streams < 1) return fail(VITS_ERR_INVALID, "set_generator_streams: at least the caller's stream");
  m->side_limit = n_streams - 1;
  return VITS_OK;
}

int vits_model_set_front_end_overlapped(vits_model* m, int on) {
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "set_front_end_overlapped: not a live model handle");
  m->beside_generator = on != 0;
  return VITS_OK;
}

int vits_model_set_ln_fused(vits_model* m, int on) {
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "set_ln_fused: not a live model handle");
  m->ln_fused = on != 0;
  return VITS_OK;
}

int vits_model_set_range_flag(vits_model* m, int* dev_flag) {
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "set_range_flag: not a live model handle");
  m->range_flag = dev_flag;
  return VITS_OK;
}

int vits_model_set_tensor(vits_model* m, const char* name, const float* p, int64_t numel) {
  if (!model_alive(m) || !name || !p || numel <= 0) return fail(VITS_ERR_INVALID, "set_tensor: bad argument");
  m->plain[name] = Plain{p, numel};
  m->finalized = false;
  return VITS_OK;
}

int vits_model_set_conv(vits_model* m, const char* name, const void* w, const float* bias, int cin, int rows, int taps,
                        int dilation, int mode, int precision) {
  if (!model_alive(m) || !name || !w) return fail(VITS_ERR_INVALID, "set_conv: bad argument");
  ConvDesc d;
  TRY(make_desc(d, w, bias, cin, rows, taps, dilation, mode, precision));
  m->convs[name] = d;
  m->finalized = false;
  return VITS_OK;
}

int vits_model_set_conv_alt(vits_model* m, const char* name, const void* w_f16x2) {
  if (!model_alive(m) || !name || !w_f16x2) return fail(VITS_ERR_INVALID, "set_conv_alt: bad argument");
  auto it = m->convs.find(name);
  if (it == m->convs.end()) return fail(VITS_ERR_MISSING, "set_conv_alt: no such conv (vits_model_set_conv comes first)");
  if (it->second.prec != PREC_BF16X3 || (it->second.mode != MODE_PLAIN && it->second.mode != MODE_GATE))
    return fail(VITS_ERR_INVALID, "set_conv_alt: only a bf16x3 conv1d takes a second packing");
  it->second.w_alt = w_f16x2;
  m->finalized = false;
  return VITS_OK;
}

namespace { int flow_scratch_for_current_device(void** out); }

int vits_model_finalize(vits_model* m) {
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "finalize: not a live model handle");
  const vits_config& c = m->cfg;
  const int C = c.hidden_channels, F = c.filter_channels, G = c.gin_channels, half = c.inter_channels / 2;
  const int NR = 2 * c.window_size + 1, DK = C / c.n_heads;
  std::string miss;
  m->symbol_emb = need_plain(m, "enc_p.symbol_emb", (int64_t)c.n_vocab * C, miss);
  // speaker-less (n_speakers == 0; reference models.py:719-722: g = None): no speaker table, no conditioning matrix;
  // the caller passes an all-zero conditioning vector instead of calling vits_speaker_cond
  m->emb_g = c.n_speakers > 0 ? need_plain(m, "emb_g", (int64_t)c.n_speakers * G, miss) : nullptr;
  const int ch0 = c.upsample_initial_channel;
  m->off_dp = 0; m->off_pitch = C; m->off_energy = 2 * C; m->off_dec = 3 * C; m->off_flow = 3 * C + ch0;
  m->cond_rows = m->off_flow + c.n_flows * 2 * C * c.flow_layers;
  // optional posterior encoder: its layer count and input width come from what was registered
  m->has_encq = m->convs.count("enc_q.pre") > 0;
  m->encq = FlowLayer();
  if (m->has_encq) {
    int nlq = 0;
    while (m->convs.count("enc_q.in." + std::to_string(nlq))) ++nlq;
    if (nlq == 0) return fail(VITS_ERR_MISSING, "model is missing enc_q.in.0 (conv not set)");
    m->encq_cin = m->convs["enc_q.pre"].Cin;
    m->encq_pre = need_conv(m, "enc_q.pre", m->encq_cin, C, 1, MODE_PLAIN, miss);
    m->encq_proj = need_conv(m, "enc_q.proj", C, 2 * c.inter_channels, 1, MODE_PLAIN, miss);
    for (int l = 0; l < nlq; ++l) {
      m->encq.in.push_back(need_conv(m, "enc_q.in." + std::to_string(l), C, 2 * C, c.flow_kernel, MODE_GATE, miss));
      if (l + 1 < nlq) m->encq.res.push_back(need_conv(m, "enc_q.res." + std::to_string(l), C, C, 1, MODE_PLAIN, miss));
      if (l + 1 < nlq && m->convs.count("enc_q.rs." + std::to_string(l)))
        m->encq.rs.push_back(need_conv(m, "enc_q.rs." + std::to_string(l), C, 2 * C, 1, MODE_PLAIN, miss));
      m->encq.skip.push_back(need_conv(m, "enc_q.skip." + std::to_string(l), C, C, 1, MODE_PLAIN, miss));
    }
    m->off_encq = m->cond_rows;
    m->cond_rows += 2 * C * nlq;
  }
  m->cond_w = c.n_speakers > 0 ? need_plain(m, "cond.w", (int64_t)m->cond_rows * G, miss) : nullptr;
  m->cond_b = c.n_speakers > 0 ? need_plain(m, "cond.b", m->cond_rows, miss) : nullptr;
  const int nlayers[3] = {c.n_layers, c.pitch_layers, c.n_layers};
  for (int e = 0; e < 3; ++e) {
    m->enc[e].clear();
    for (int i = 0; i < nlayers[e]; ++i) {
      const std::string p = std::string(kEncNames[e]) + "." + std::to_string(i);
      EncLayer l;
      l.qkv = need_conv(m, p + ".qkv", C, 3 * C, 1, MODE_PLAIN, miss);
      l.o = need_conv(m, p + ".o", C, C, 1, MODE_PLAIN, miss);
      l.ffn1 = need_conv(m, p + ".ffn1", C, F, c.kernel_size, MODE_PLAIN, miss);
      l.ffn2 = need_conv(m, p + ".ffn2", F, C, c.kernel_size, MODE_PLAIN, miss);
      l.ek = need_plain(m, p + ".emb_rel_k", (int64_t)NR * DK, miss);
      l.ev = need_plain(m, p + ".emb_rel_v", (int64_t)NR * DK, miss);
      l.ln1g = need_plain(m, p + ".ln1.g", C, miss);
      l.ln1b = need_plain(m, p + ".ln1.b", C, miss);
      l.ln2g = need_plain(m, p + ".ln2.g", C, miss);
      l.ln2b = need_plain(m, p + ".ln2.b", C, miss);
      m->enc[e].push_back(l);
    }
  }
  const int DF = c.dur_filter, EF = c.energy_filter;
  m->dp_conv1 = need_conv(m, "dp.conv1", C, DF, 3, MODE_PLAIN, miss);
  m->dp_conv2 = need_conv(m, "dp.conv2", DF, DF, 3, MODE_PLAIN, miss);
  m->dp_ln1g = need_plain(m, "dp.ln1.g", DF, miss); m->dp_ln1b = need_plain(m, "dp.ln1.b", DF, miss);
  m->dp_ln2g = need_plain(m, "dp.ln2.g", DF, miss); m->dp_ln2b = need_plain(m, "dp.ln2.b", DF, miss);
  m->dp_proj_w = need_plain(m, "dp.proj.w", DF, miss); m->dp_proj_b = need_plain(m, "dp.proj.b", 1, miss);
  m->pitch_proj_w = need_plain(m, "pitch.proj.w", C, miss); m->pitch_proj_b = need_plain(m, "pitch.proj.b", 1, miss);
  m->pitch_pre_w = need_plain(m, "pitch_prenet.w", (int64_t)C * 3, miss); m->pitch_pre_b = need_plain(m, "pitch_prenet.b", C, miss);
  m->en_conv1 = need_conv(m, "energy.conv1", C, EF, 3, MODE_PLAIN, miss);
  m->en_conv2 = need_conv(m, "energy.conv2", EF, EF, 3, MODE_PLAIN, miss);
  m->en_ln1g = need_plain(m, "energy.ln1.g", EF, miss); m->en_ln1b = need_plain(m, "energy.ln1.b", EF, miss);
  m->en_ln2g = need_plain(m, "energy.ln2.g", EF, miss); m->en_ln2b = need_plain(m, "energy.ln2.b", EF, miss);
  m->en_lin_w = need_plain(m, "energy.lin.w", EF, miss); m->en_lin_b = need_plain(m, "energy.lin.b", 1, miss);
  m->en_pre_w = need_plain(m, "energy_prenet.w", (int64_t)C * 3, miss); m->en_pre_b = need_plain(m, "energy_prenet.b", C, miss);
  m->project = need_conv(m, "project", C, 2 * C, 1, MODE_PLAIN, miss);
  m->flows.clear();
  for (int f = 0; f < c.n_flows; ++f) {
    const std::string p = "flow." + std::to_string(f);
    FlowLayer fl;
    fl.pre = need_conv(m, p + ".pre", half, C, 1, MODE_PLAIN, miss);
    fl.post = need_conv(m, p + ".post", C, half, 1, MODE_PLAIN, miss);
    for (int l = 0; l < c.flow_layers; ++l) {
      fl.in.push_back(need_conv(m, p + ".in." + std::to_string(l), C, 2 * C, c.flow_kernel, MODE_GATE, miss));
      if (l + 1 < c.flow_layers) fl.res.push_back(need_conv(m, p + ".res." + std::to_string(l), C, C, 1, MODE_PLAIN, miss));
      if (l + 1 < c.flow_layers && m->convs.count(p + ".rs." + std::to_string(l)))
        fl.rs.push_back(need_conv(m, p + ".rs." + std::to_string(l), C, 2 * C, 1, MODE_PLAIN, miss));
      fl.skip.push_back(need_conv(m, p + ".skip." + std::to_string(l), C, C, 1, MODE_PLAIN, miss));
    }
    m->flows.push_back(fl);
  }
  m->conv_pre = need_conv(m, "dec.conv_pre", c.inter_channels, ch0, 7, MODE_PLAIN, miss);
  m->ups.clear();
  m->rbs.clear();
  m->total_up = 1;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int u = c.upsample_rates[i], cin = ch0 >> i, cout = ch0 >> (i + 1);
    const int mode = u == 8 ? MODE_UP8 : u == 4 ? MODE_UP4 : MODE_UP2;
    m->ups.push_back(need_conv(m, "dec.ups." + std::to_string(i), cin, cout * u, c.upsample_kernel_sizes[i] / u, mode, miss));
    m->total_up *= u;
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      ResBlock rb;
      const std::string p = "dec.rb." + std::to_string(i * c.n_resblock_kernels + j);
      for (int q = 0; q < c.n_resblock_dilations; ++q) {
        rb.c1.push_back(need_conv(m, p + ".c1." + std::to_string(q), cout, cout, c.resblock_kernel_sizes[j], MODE_PLAIN, miss));
        if (c.resblock == 2) continue;             // modules.ResBlock2: one conv per dilation
        rb.c2.push_back(need_conv(m, p + ".c2." + std::to_string(q), cout, cout, c.resblock_kernel_sizes[j], MODE_PLAIN, miss));
      }
      m->rbs.push_back(rb);
    }
  }
  m->conv_post_w = need_plain(m, "dec.conv_post.w", (int64_t)(ch0 >> c.n_upsamples) * 7, miss);
  if (!miss.empty()) return fail(VITS_ERR_MISSING, "model is missing " + miss);
  release_streams(m);
  const char* env = getenv("VITS_GEN_STREAMS");
  const int want = env ? atoi(env) : c.n_resblock_kernels;
  const int n_side = std::max(0, std::min(std::min(want, c.n_resblock_kernels) - 1, kSideStreams));
  HIP_TRY(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  for (int i = 0; i < c.n_resblock_kernels; ++i) HIP_TRY(hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming));
  for (int i = 0; i < n_side; ++i) {
    TRY(shared_side_stream(i, &m->side[i]));
    m->n_side = i + 1;
  }
  // vits_flow_reverse_noalloc's array: its address on this device is looked up here, at load time, because the first
  // look-up on a device makes the runtime load this file's device image (the array included), which allocates --
  // the entry point itself must not, and after this it only reads the pointer
  void* scratch = nullptr;
  TRY(flow_scratch_for_current_device(&scratch));
  m->finalized = true;
  return VITS_OK;
}

int vits_pack_conv_weight(const float* w, int rows, int cin, int taps, float* out) {
  if (!w || !out || rows <= 0 || rows % 32 || cin <= 0 || cin % CI_CHUNK || taps < 1)
    return fail(VITS_ERR_INVALID, "pack: rows % 32 == 0 and cin % 16 == 0 required");
  const int mt_total = rows / 32, nchunks = cin / CI_CHUNK;
  int64_t o = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int k = 0; k < taps; ++k)
      for (int sq = 0; sq < 2; ++sq)
        for (int mt = 0; mt < mt_total; ++mt)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int row = mt * 32 + (lane & 31);
              const int ci = chunk * CI_CHUNK + sq * 8 + e * 2 + (lane >> 5);
              out[o++] = w[((int64_t)row * cin + ci) * taps + k];
            }
  return VITS_OK;
}

// round-to-nearest-even fp32 -> bf16 (bit pattern) and back, on the host
static inline uint16_t f2bf(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int vits_pack_conv_weight_bf16x3(const float* w, int rows, int cin, int taps, uint16_t* out) {
  if (!w || !out || rows <= 0 || rows % 32 || cin <= 0 || cin % 16 || taps < 1)
    return fail(VITS_ERR_INVALID, "pack: rows % 32 == 0 and cin % 16 == 0 required");
  // [cin/16][taps][rows/32][3 planes][64 lanes][8]: lane l of m-tile mt holds, for j = 0..7, the three bf16 terms
  // (h, m, l with h + m + l = w to 24 bits) of W[mt*32 + (l & 31)][16*chunk + 8*(l >> 5) + j][tap]
  const int mt_total = rows / 32, nchunks = cin / 16;
  int64_t o = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int k = 0; k < taps; ++k)
      for (int mt = 0; mt < mt_total; ++mt) {
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int row = mt * 32 + (lane & 31);
            const int ci = chunk * 16 + 8 * (lane >> 5) + j;
            const float v = w[((int64_t)row * cin + ci) * taps + k];
            const uint16_t h = f2bf(v);
            const float r1 = v - bf2f(h);
            const uint16_t m = f2bf(r1);
            const float r2 = r1 - bf2f(m);
            const uint16_t l = f2bf(r2);
            out[o + (0 * 64 + lane) * 8 + j] = h;
            out[o + (1 * 64 + lane) * 8 + j] = m;
            out[o + (2 * 64 + lane) * 8 + j] = l;
          }
        o += 3 * 64 * 8;
      }
  return VITS_OK;
}

int vits_pack_conv_weight_bf16(const float* w, int rows, int cin, int taps, uint16_t* out) {
  if (!w || !out || rows <= 0 || rows % 32 || cin <= 0 || cin % 16 || taps < 1)
    return fail(VITS_ERR_INVALID, "pack: rows % 32 == 0 and cin % 16 == 0 required");
  // [cin/16][taps][rows/32][64 lanes][8]: lane l of m-tile mt holds bf16(W[mt*32 + (l & 31)][16*chunk + 8*(l >> 5) + j][tap])
  const int mt_total = rows / 32, nchunks = cin / 16;
  int64_t o = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int k = 0; k < taps; ++k)
      for (int mt = 0; mt < mt_total; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int row = mt * 32 + (lane & 31);
            const int ci = chunk * 16 + 8 * (lane >> 5) + j;
            out[o++] = f2bf(w[((int64_t)row * cin + ci) * taps + k]);
          }
  return VITS_OK;
}

int vits_pack_conv_weight_f16x2(const float* w, int rows, int cin, int taps, uint16_t* out) {
  if (!w || !out || rows <= 0 || rows % 32 || cin <= 0 || cin % 16 || taps < 1)
    return fail(VITS_ERR_INVALID, "pack: rows % 32 == 0 and cin % 16 == 0 required");
  // [cin/16][taps][rows/32][2 planes][64 lanes][8]: h = fp16(w), l = fp16(w - h), round to nearest even, subnormals kept
  const int mt_total = rows / 32, nchunks = cin / 16;
  int64_t o = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int k = 0; k < taps; ++k)
      for (int mt = 0; mt < mt_total; ++mt) {
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int row = mt * 32 + (lane & 31);
            const int ci = chunk * 16 + 8 * (lane >> 5) + j;
            const float v = w[((int64_t)row * cin + ci) * taps + k];
            if (!(std::fabs(v) <= 65504.f)) return fail(VITS_ERR_INVALID, "pack_f16x2: weight outside the fp16 range");
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            std::memcpy(&out[o + (0 * 64 + lane) * 8 + j], &h, 2);
            std::memcpy(&out[o + (1 * 64 + lane) * 8 + j], &l, 2);
          }
        o += 2 * 64 * 8;
      }
  return VITS_OK;
}

#define CHECK_MODEL(m)                                                                                  \
  if (!model_alive(m)) return fail(VITS_ERR_INVALID, "not a live model handle");                        \
  if (!(m)->finalized) return fail(VITS_ERR_INVALID, "model not finalized")

int vits_cond_rows(const vits_model* m) { return model_alive(m) && m->finalized ? m->cond_rows : -1; }

int vits_speaker_cond(const vits_model* m, const int64_t* sid, int B, float* cond, int* err_flag, void* stream) {
  CHECK_MODEL(m);
  if (m->cfg.n_speakers == 0)
    return fail(VITS_ERR_INVALID, "speaker_cond: the model is speaker-less; its conditioning vector is all zeros");
  HIP_TRY(launch_cond_gemv(m->emb_g, sid, m->cond_w, m->cond_b, cond, B, m->cond_rows, m->cfg.gin_channels,
                           m->cfg.n_speakers, err_flag, (hipStream_t)stream));
  return VITS_OK;
}

int64_t vits_encoder_workspace_bytes(const vits_model* m, int B, int L) {
  if (!model_alive(m)) return -1;
  return encoder_ws_floats(m, B, L) * 4;
}

int vits_text_encoder(const vits_model* m, const int64_t* phonemes, const int* lengths, int B, int P, float* x,
                      int* err_flag, void* ws, int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(launch_embed(phonemes, m->symbol_emb, lengths, x, B, P, m->cfg.hidden_channels, m->cfg.n_vocab, err_flag, s));
  Arena ar(ws, ws_bytes);
  return run_encoder(m, 0, x, lengths, B, P, ar, s);
}

int vits_encoder_stack(const vits_model* m, int which, float* x, const int* lengths, int B, int L, void* ws,
                       int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  if (which < 0 || which > 2) return fail(VITS_ERR_INVALID, "encoder index must be 0, 1 or 2");
  Arena ar(ws, ws_bytes);
  return run_encoder(m, which, x, lengths, B, L, ar, (hipStream_t)stream);
}

int64_t vits_variance_workspace_bytes(const vits_model* m, int B, int P) {
  if (!model_alive(m)) return -1;
  const int64_t C = m->cfg.hidden_channels, EF = m->cfg.energy_filter, DF = m->cfg.dur_filter;
  const int64_t big = std::max<int64_t>(std::max<int64_t>(EF, DF), C);
  return 2 * pad256((int64_t)B * big * P) + 6 * pad256((int64_t)B * P) + encoder_ws_floats(m, B, P) * 4;
}

int vits_variance_adaptor(const vits_model* m, float* x, const int* lengths, const float* cond, int B, int P,
                          int dur_mode, float dur_scale, float* dur_out, int pitch_mode, float pitch_scale,
                          const float* pitch_hz, float* f0_out, int energy_mode, float energy_scale,
                          const float* energy_raw, float* energy_out, void* ws, int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  hipStream_t s = (hipStream_t)stream;
  const int C = m->cfg.hidden_channels, EF = m->cfg.energy_filter, DF = m->cfg.dur_filter;
  const int big = std::max(std::max(EF, DF), C);
  const int cr = m->cond_rows;
  Arena ar(ws, ws_bytes);
  float* h1 = ar.take((int64_t)B * big * P);
  float* h2 = ar.take((int64_t)B * big * P);
  float* v0 = ar.take((int64_t)B * P);   // logw / predictor outputs
  float* lf0 = ar.take((int64_t)B * P);
  float* ne = ar.take((int64_t)B * P);
  if (!h1 || !h2 || !v0 || !lf0 || !ne) return fail(VITS_ERR_WORKSPACE, "variance adaptor: workspace too small");
  // duration (reference models.py:125-139, 729-733)
  if (dur_mode == 0) {
    if (!dur_out) return fail(VITS_ERR_INVALID, "dur_out is required when durations are predicted");
    TRY(conv_simple(m->dp_conv1, x, h1, B, P, lengths, F_IN_MASK | F_RELU, s, nullptr, nullptr, 0, cond + m->off_dp, cr));
    HIP_TRY(launch_layer_norm(h1, h1, m->dp_ln1g, m->dp_ln1b, lengths, B, DF, P, false, s));
    TRY(conv_simple(m->dp_conv2, h1, h2, B, P, lengths, F_IN_MASK | F_RELU, s));
    HIP_TRY(launch_layer_norm(h2, h2, m->dp_ln2g, m->dp_ln2b, lengths, B, DF, P, false, s));
    HIP_TRY(launch_conv_to1(h2, (long long)DF * P, P, m->dp_proj_w, m->dp_proj_b, v0, lengths, B, DF, 1, P, F_IN_MASK,
                            1.f, 1.f, false, true, s));
    HIP_TRY(launch_duration_finalize(v0, lengths, dur_scale, dur_out, B, P, s));
  }
  // pitch (reference models.py:547-556, 736-743)
  if (pitch_mode == 0) {
    HIP_TRY(launch_add_cbias_mask(x, cond + m->off_pitch, cr, lengths, h1, B, C, P, true, s));
    TRY(run_encoder(m, 1, h1, lengths, B, P, ar, s));
    HIP_TRY(launch_conv_to1(h1, (long long)C * P, P, m->pitch_proj_w, m->pitch_proj_b, v0, lengths, B, C, 1, P, 0, 1.f,
                            1.f, false, false, s));
    HIP_TRY(launch_pitch_math(v0, false, pitch_scale, lf0, f0_out, B * P, s));
  } else {
    if (!pitch_hz) return fail(VITS_ERR_INVALID, "pitch_hz is required for a pitch override");
    HIP_TRY(launch_pitch_math(pitch_hz, true, 1.f, lf0, f0_out, B * P, s));
  }
  HIP_TRY(launch_prenet_add(x, lf0, m->pitch_pre_w, m->pitch_pre_b, B, C, P, s));
  // energy (reference frame_prior_network.py:104-129, models.py:746-753): no masking anywhere
  if (energy_mode == 0) {
    TRY(conv_simple(m->en_conv1, x, h1, B, P, nullptr, F_RELU, s, nullptr, nullptr, 0, cond + m->off_energy, cr));
    HIP_TRY(launch_layer_norm(h1, h1, m->en_ln1g, m->en_ln1b, nullptr, B, EF, P, false, s));
    TRY(conv_simple(m->en_conv2, h1, h2, B, P, nullptr, F_RELU, s));
    HIP_TRY(launch_layer_norm(h2, h2, m->en_ln2g, m->en_ln2b, nullptr, B, EF, P, false, s));
    HIP_TRY(launch_conv_to1(h2, (long long)EF * P, P, m->en_lin_w, m->en_lin_b, v0, nullptr, B, EF, 1, P, 0, 1.f, 1.f,
                            false, false, s));
    HIP_TRY(launch_energy_math(v0, false, energy_scale, ne, energy_out, B * P, s));
  } else {
    if (!energy_raw) return fail(VITS_ERR_INVALID, "energy_raw is required for an energy override");
    HIP_TRY(launch_energy_math(energy_raw, true, 1.f, ne, energy_out, B * P, s));
  }
  HIP_TRY(launch_prenet_add(x, ne, m->en_pre_w, m->en_pre_b, B, C, P, s));
  return VITS_OK;
}

int vits_length_regulate_plan(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, void* stream) {
  if (!dur || !cum || !totals) return fail(VITS_ERR_INVALID, "length_regulate_plan: null argument");
  if (P > 2048) return fail(VITS_ERR_INVALID, "length regulator supports at most 2048 phonemes per utterance");
  HIP_TRY(launch_lr_plan(dur, dur_dtype, B, P, cum, totals, (hipStream_t)stream));
  return VITS_OK;
}

int vits_length_regulate_plan_report(const void* dur, int dur_dtype, int B, int P, int* cum, int* totals, int* report,
                                     const int* flag0, const int* flag1, void* stream) {
  if (!dur || !cum || !totals || !report) return fail(VITS_ERR_INVALID, "length_regulate_plan_report: null argument");
  if (P > 2048) return fail(VITS_ERR_INVALID, "length regulator supports at most 2048 phonemes per utterance");
  HIP_TRY(launch_lr_plan(dur, dur_dtype, B, P, cum, totals, (hipStream_t)stream, report, flag0, flag1));
  return VITS_OK;
}

int vits_length_regulate(const float* x, const int* cum, const int* totals, int B, int C, int P, int T, float* x_frame,
                         void* stream) {
  if (!x || !cum || !totals || !x_frame) return fail(VITS_ERR_INVALID, "length_regulate: null argument");
  HIP_TRY(launch_lr_gather(x, cum, totals, x_frame, B, C, P, T, (hipStream_t)stream));
  return VITS_OK;
}

int vits_frame_prior(const vits_model* m, float* x_frame, const int* frame_lengths, int B, int T, const float* noise,
                     float noise_scale, float* stats, float* z_p, float* z, void* ws, int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  hipStream_t s = (hipStream_t)stream;
  RangeFlagScope range_scope(m->range_flag);       // frame-rate convs of long batches run in f16x2
  OverlapScope overlap_scope(m->beside_generator);
  if (T > 16384) return fail(VITS_ERR_INVALID, "at most 16384 frames per utterance");
  Arena ar(ws, ws_bytes);
  TRY(run_encoder(m, 2, x_frame, frame_lengths, B, T, ar, s));
  TRY(conv_simple(m->project, x_frame, stats, B, T, frame_lengths, F_MASK_PRE, s));
  HIP_TRY(launch_sample(stats, noise, noise_scale, z_p, z, B, m->cfg.inter_channels, T, s));
  return VITS_OK;
}

int64_t vits_flow_workspace_bytes(const vits_model* m, int B, int T) {
  if (!model_alive(m)) return -1;
  return 3 * pad256((int64_t)B * m->cfg.hidden_channels * T);
}

// VITS_WN_RS=0 keeps WaveNet's res and skip 1x1 convs as two launches (A/B measurements).
static bool res_skip_one_launch() {
  static const bool on = [] {
    const char* e = getenv("VITS_WN_RS");
    return !(e && e[0] == '0');
  }();
  return on;
}

// modules.WN.forward (reference modules.py:144-172) on h (updated in place), result in `skip`: per layer the gated conv
// with its slice of the conditioning, then the 1x1 res / skip convs (the reference's single res_skip conv, split).
static int run_wn(const FlowLayer& wn, float* h, float* acts, float* skip, const float* cf, int cr, int C, int B, int T,
                  const int* lengths, hipStream_t s) {
  const int NL = (int)wn.in.size();
  for (int l = 0; l < NL; ++l) {
    TRY(conv_simple(wn.in[l], h, acts, B, T, lengths, 0, s, nullptr, cf + (long long)l * 2 * C, cr));
    if (l + 1 < NL && (int)wn.rs.size() == NL - 1 && C % 64 == 0 && res_skip_one_launch()) {
      // the reference's single res_skip conv (modules.py:164-170): rows [0, C) update x, rows [C, 2C) go to the skip sum
      ConvArgs a = base_args(acts, (long long)C * T, T, h, (long long)C * T, T, B, T, lengths, F_RES | F_MASK_POST);
      a.res = h; a.res_bs = (long long)C * T; a.res_cs = T;
      a.split_row = C; a.out_hi = skip; a.out_hi_bs = (long long)C * T; a.flags_hi = l ? F_ACCUM : 0;
      HIP_TRY(launch_conv(wn.rs[l], a, s));
    } else if (l + 1 < NL) {
      TRY(conv_simple(wn.res[l], acts, h, B, T, lengths, F_RES | F_MASK_POST, s, h));
      TRY(conv_simple(wn.skip[l], acts, skip, B, T, lengths, l ? F_ACCUM : 0, s));
    } else {
      TRY(conv_simple(wn.skip[l], acts, skip, B, T, lengths, (l ? F_ACCUM : 0) | F_MASK_POST, s));
    }
  }
  return VITS_OK;
}

// One pass of the coupling block over z, in place.  reverse: layers nf-1 .. 0, x1 = (x1 - m) * mask (synthesis,
// reference models.py:213-216, modules.py:336-339); forward: layers 0 .. nf-1, x1 = m + x1 * mask (voice conversion,
// models.py:209-212, modules.py:331-335).
static int run_flow(const vits_model* m, float* z, const int* lengths, const float* cond, int B, int T, void* ws,
                    int64_t ws_bytes, hipStream_t s, bool reverse) {
  const int C = m->cfg.hidden_channels, IC = m->cfg.inter_channels, half = IC / 2, NL = m->cfg.flow_layers;
  const int cr = m->cond_rows, nf = m->cfg.n_flows;
  RangeFlagScope range_scope(m->range_flag);       // frame-rate convs of long batches run in f16x2
  OverlapScope overlap_scope(m->beside_generator);
  Arena ar(ws, ws_bytes);
  float* h = ar.take((int64_t)B * C * T);
  float* acts = ar.take((int64_t)B * C * T);
  float* skip = ar.take((int64_t)B * C * T);
  if (!h || !acts || !skip) return fail(VITS_ERR_WORKSPACE, "flow: workspace too small");
  for (int step = 0; step < nf; ++step) {
    const int f = reverse ? nf - 1 - step : step;
    // The reference flips all channels around every coupling layer (models.py:206-216, modules.py:266-273).  Here z
    // stays in place: after an odd number of flips the layer reads the upper half as x0 and rewrites the lower half,
    // and the channel reversal inside each half is folded into pre's input order and post's output order at load.
    // Layer f has seen nf - f flips on the way back and f on the way forward: the same parity, because nf is even.
    const bool flipped = (nf - f) & 1;
    float* x0 = z + (flipped ? (long long)half * T : 0);
    float* x1 = z + (flipped ? 0 : (long long)half * T);
    const FlowLayer& fl = m->flows[f];
    {
      ConvArgs a = base_args(x0, (long long)IC * T, T, h, (long long)C * T, T, B, T, lengths, F_MASK_PRE);
      HIP_TRY(launch_conv(fl.pre, a, s));
    }
    TRY(run_wn(fl, h, acts, skip, cond + m->off_flow + (long long)f * 2 * C * NL, cr, C, B, T, lengths, s));
    {
      ConvArgs a = base_args(skip, (long long)C * T, T, x1, (long long)IC * T, T, B, T, lengths,
                             F_MASK_PRE | (reverse ? F_RES_SUB : F_RES) | F_MASK_POST);
      a.res = x1; a.res_bs = (long long)IC * T; a.res_cs = T;
      HIP_TRY(launch_conv(fl.post, a, s));
    }
  }
  return VITS_OK;
}

int vits_flow_reverse(const vits_model* m, float* z, const int* lengths, const float* cond, int B, int T, void* ws,
                      int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  return run_flow(m, z, lengths, cond, B, T, ws, ws_bytes, (hipStream_t)stream, true);
}

// ---- vits_flow_reverse for callers that cannot allocate ----
// The coupling block's three work tensors (h, gate output, skip sum; 3 * B * hidden * T floats) live in a fixed array
// in the library's device image instead of a caller workspace.  Every GPU the library runs on holds its own copy of
// the array at its own address for the life of the process; host code never uses the array's name as an address
// (it is only a host-side stand-in), it uses the per-device pointer hipGetSymbolAddress returned.
#define VITS_NOALLOC_MAX_POSITIONS 8192      /* B * T */
#define VITS_NOALLOC_MAX_HIDDEN 192
constexpr int64_t kFlowScratchBytes = 3 * (int64_t)VITS_NOALLOC_MAX_POSITIONS * VITS_NOALLOC_MAX_HIDDEN * 4;   // 18 MiB
static __device__ __align__(256) __attribute__((used)) unsigned char g_flow_scratch[kFlowScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_flow_scratch_ptr[kMaxDevices] = {};

// Address of the current device's copy of g_flow_scratch (resolved once per device, with that device current).
int flow_scratch_for_current_device(void** out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(VITS_ERR_INVALID, "flow_reverse_noalloc: device index out of range");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_flow_scratch_ptr[dev]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_flow_scratch)));
    if (!p) return fail(VITS_ERR_HIP, "hipGetSymbolAddress returned a null address for the flow scratch");
    g_flow_scratch_ptr[dev] = p;
  }
  *out = g_flow_scratch_ptr[dev];
  return VITS_OK;
}
}  // namespace

int64_t vits_flow_noalloc_max_positions(void) { return VITS_NOALLOC_MAX_POSITIONS; }

int vits_flow_reverse_noalloc(const vits_model* m, float* z, const int* lengths, const float* cond, int B, int T,
                              void* stream) {
  CHECK_MODEL(m);
  if (!z || !lengths || !cond) return fail(VITS_ERR_INVALID, "flow_reverse_noalloc: null argument");
  if (B <= 0 || T <= 0) return fail(VITS_ERR_INVALID, "flow_reverse_noalloc: B and T must be positive");
  if (m->cfg.hidden_channels > VITS_NOALLOC_MAX_HIDDEN || (int64_t)B * T > VITS_NOALLOC_MAX_POSITIONS ||
      vits_flow_workspace_bytes(m, B, T) > kFlowScratchBytes)
    return fail(VITS_ERR_WORKSPACE, "flow_reverse_noalloc: B * T exceeds 8192 positions (or hidden_channels 192), the "
                                    "size of the library's fixed scratch; use vits_flow_reverse with a workspace");
  void* scratch = nullptr;
  TRY(flow_scratch_for_current_device(&scratch));
  // nothing in the scratch has to start at zero: h, the gate output and the skip sum are each written in full
  // (every row, every t < T of every batch row) before they are read
  return run_flow(m, z, lengths, cond, B, T, scratch, kFlowScratchBytes, (hipStream_t)stream, true);
}

void vits_op_set_range_flag(int* dev_flag) { g_range_flag = dev_flag; }

int vits_flow_forward(const vits_model* m, float* z, const int* lengths, const float* cond, int B, int T, void* ws,
                      int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  return run_flow(m, z, lengths, cond, B, T, ws, ws_bytes, (hipStream_t)stream, false);
}

int vits_posterior_channels(const vits_model* m) {
  return model_alive(m) && m->finalized && m->has_encq ? m->encq_cin : -1;
}

int64_t vits_posterior_workspace_bytes(const vits_model* m, int B, int T) {
  if (!model_alive(m)) return -1;
  return 3 * pad256((int64_t)B * m->cfg.hidden_channels * T);
}

int vits_posterior_encoder(const vits_model* m, const float* spec, const int* lengths, const float* cond, int B, int T,
                           const float* noise, float* stats, float* z, void* ws, int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  if (!m->has_encq) return fail(VITS_ERR_MISSING, "posterior encoder: the model was loaded without enc_q.* weights");
  if (!spec || !lengths || !cond || !noise || !stats || !z) return fail(VITS_ERR_INVALID, "posterior encoder: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int C = m->cfg.hidden_channels, IC = m->cfg.inter_channels;
  Arena ar(ws, ws_bytes);
  float* h = ar.take((int64_t)B * C * T);
  float* acts = ar.take((int64_t)B * C * T);
  float* skip = ar.take((int64_t)B * C * T);
  if (!h || !acts || !skip) return fail(VITS_ERR_WORKSPACE, "posterior encoder: workspace too small");
  // x = pre(spec) * mask; x = WN(x, mask, g); stats = proj(x) * mask; z = (m + eps * exp(logs)) * mask
  TRY(conv_simple(m->encq_pre, spec, h, B, T, lengths, F_MASK_PRE, s));
  TRY(run_wn(m->encq, h, acts, skip, cond + m->off_encq, m->cond_rows, C, B, T, lengths, s));
  TRY(conv_simple(m->encq_proj, skip, stats, B, T, lengths, F_MASK_PRE, s));
  HIP_TRY(launch_sample(stats, noise, 1.f, z, z, B, IC, T, s, lengths));
  return VITS_OK;
}

static int64_t gen_buf_floats(const vits_model* m, int B, int t_dec) {
  int64_t mx = (int64_t)m->cfg.upsample_initial_channel * t_dec;
  int64_t L = t_dec;
  for (int i = 0; i < m->cfg.n_upsamples; ++i) {
    L *= m->cfg.upsample_rates[i];
    mx = std::max<int64_t>(mx, (int64_t)(m->cfg.upsample_initial_channel >> (i + 1)) * L);
  }
  return mx * B;
}

// VITS_GEN_STAGED=0 keeps fp32 tensors between the two convs of a ResBlock pair (measurement knob).
static bool staged_pairs_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_GEN_STAGED");
    return !(e && e[0] == '0');
  }();
  return on;
}

// VITS_GEN_FUSED=0 keeps the 32-channel ResBlock pairs as two launches (measurement knob).
static bool fused_pairs_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_GEN_FUSED");
    return !(e && e[0] == '0');
  }();
  return on;
}

// VITS_GEN_CHAIN=0 keeps every ResBlock pair by pair (measurement knob).
static bool whole_resblocks_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_GEN_CHAIN");
    return !(e && e[0] == '0');
  }();
  return on;
}

// VITS_GEN_UPS_STAGE=0: the staged copy of a stage's input comes from the conversion kernel, not from the transposed
// conv's epilogue (measurement knob).
static bool ups_stage_enabled() {
  static const bool on = [] {
    const char* e = getenv("VITS_GEN_UPS_STAGE");
    return !(e && e[0] == '0');
  }();
  return on;
}

// How one stage of the generator runs (decided once per call: it depends on batch and length), shared by the workspace
// query and vits_generator so that the two cannot disagree.
struct StagePlan {
  bool stage_x = false, stage_bf16 = false;      // x converted once into the staged f16x2 / bf16 operand format (XG)
  int nseg[8] = {0};                             // > 0: ResBlock j runs as that many whole-chain / fused-pair launches
  int seg[8][8];
  bool fuse_chain[8] = {false};                  // every pair of ResBlock j in one launch each
  int bufs[8] = {0};                             // work tensors of the stage's size ResBlock j needs (Y, T, YG)
  int total_bufs = 0;
};

static StagePlan plan_stage(const vits_model* m, int i, int B, int L) {
  const vits_config& c = m->cfg;
  const int nk = c.n_resblock_kernels, nd = c.n_resblock_dilations;
  const bool single = c.resblock == 2;            // modules.ResBlock2 (reference modules.py:228-252)
  StagePlan p;
  // f16x2 arithmetic: every conv of an unfused pair takes its input in the staged operand format (leaky-relu applied,
  // split into fp16 h / l) -- x is converted once for all chains, and the second conv of a pair writes its result
  // twice, as fp32 (the residual of the next pair) and staged (the input of the next pair's first conv).
  // Plain-bf16 arithmetic: the same where the stage's convs run on the LDS-DMA kernel (the one that writes the second
  // copy), i.e. for rows in 128s at sizes that fill the chip; decided per stage, for every chain at once.
  for (int j = 0; j < nk && staged_pairs_enabled() && !single; ++j) {
    const ResBlock& rb = m->rbs[i * nk + j];
    bool fused = fused_pairs_enabled();
    for (int q = 0; q < nd && fused; ++q) fused = pair_profitable(rb.c1[q], rb.c2[q]);
    if (!fused && rb.c1[0].prec == PREC_F16X2 && rb.c2[0].prec == PREC_F16X2) p.stage_x = true;
  }
  if (!p.stage_x && staged_pairs_enabled() && !single) {
    p.stage_bf16 = true;
    for (int j = 0; j < nk && p.stage_bf16; ++j) {
      const ResBlock& rb = m->rbs[i * nk + j];
      for (int q = 0; q < nd && p.stage_bf16; ++q)
        p.stage_bf16 = conv_takes_glb(rb.c1[q], B, L) && conv_takes_glb(rb.c2[q], B, L);
    }
  }
  for (int j = 0; j < nk; ++j) {
    const ResBlock& rb = m->rbs[i * nk + j];
    // One decision per chain: the fused pair ping-pongs between Y and T (its blocks read a halo of their input, so
    // it cannot run in place) while the two-launch pair keeps T for the tensor between its convs and updates Y in
    // place; a chain that mixed the two would hand a fused pair's output in T to a two-launch pair that overwrites
    // T before reading it.  So a chain is fused only when every one of its pairs is.
    p.fuse_chain[j] = fused_pairs_enabled() && !single;
    for (int q = 0; q < nd && p.fuse_chain[j]; ++q) p.fuse_chain[j] = pair_profitable(rb.c1[q], rb.c2[q]);
    p.nseg[j] = whole_resblocks_enabled() && !single && nd <= 8 ? chain_plan(rb.c1.data(), rb.c2.data(), nd, p.seg[j]) : 0;
    if (p.nseg[j] > 0) p.bufs[j] = p.nseg[j] > 1 ? 2 : 0;       // a whole ResBlock in one launch needs none
    else if (single || p.fuse_chain[j]) p.bufs[j] = nd > 1 ? 2 : 0;
    else {
      const int p1 = rb.c1[0].prec;
      const bool staged_in = staged_pairs_enabled() && ((p1 == PREC_F16X2 && p.stage_x) || (p1 == PREC_BF16 && p.stage_bf16));
      p.bufs[j] = staged_in ? 3 : 2;
    }
    p.total_bufs += p.bufs[j];
  }
  return p;
}

// floats of the work tensors every ResBlock chain of one stage needs, the largest over the stages
static int64_t gen_scratch_floats(const vits_model* m, int B, int t_dec) {
  int64_t L = t_dec, mx = 0;
  for (int i = 0; i < m->cfg.n_upsamples; ++i) {
    L *= m->cfg.upsample_rates[i];
    const int64_t one = pad256((int64_t)(m->cfg.upsample_initial_channel >> (i + 1)) * L * B) / 4;
    mx = std::max(mx, one * plan_stage(m, i, B, (int)L).total_bufs);
  }
  return mx;
}

int64_t vits_generator_workspace_bytes(const vits_model* m, int B, int t_dec) {
  if (!model_alive(m)) return -1;
  // x, the sum of the chains, x in staged form (the largest stage's size each); then, sized per stage, what the
  // ResBlock chains of that stage need: nothing for a ResBlock that runs whole, its carrier and the tensor between the
  // two convs of a pair otherwise, and the carrier in staged form for unfused pairs
  return 3 * pad256(gen_buf_floats(m, B, t_dec)) + pad256(gen_scratch_floats(m, B, t_dec));
}

int vits_generator(const vits_model* m, const float* z, const int* lengths, const float* cond, int B, int T, int t_dec,
                   float* wav, void* ws, int64_t ws_bytes, void* stream) {
  CHECK_MODEL(m);
  hipStream_t s = (hipStream_t)stream;
  if (t_dec <= 0 || t_dec > T) return fail(VITS_ERR_INVALID, "generator: 0 < t_dec <= T required");
  const vits_config& c = m->cfg;
  const int IC = c.inter_channels, ch0 = c.upsample_initial_channel, nk = c.n_resblock_kernels, nd = c.n_resblock_dilations;
  Arena ar(ws, ws_bytes);
  const int64_t nbuf = gen_buf_floats(m, B, t_dec);
  float *X = ar.take(nbuf), *XS = ar.take(nbuf), *XG = ar.take(nbuf), *Y[8], *Tt[8], *YG[8];
  const int64_t scratch_floats = gen_scratch_floats(m, B, t_dec);
  float* scratch = ar.take(scratch_floats);       // the chains' work tensors, carved anew for every stage
  if (!X || !XS || !XG || !scratch) return fail(VITS_ERR_WORKSPACE, "generator: workspace too small");
  // per-kernel event timing needs the launches of one step serialised on one stream
  const int n_side = prof_enabled() ? 0 : std::max(0, std::min(m->n_side, m->side_limit));
  RangeFlagScope range_scope(m->range_flag);
  // conv_pre on (z * mask)[:, :, :t_dec] + cond(g)   (reference models.py:285-287, 765)
  {
    ConvArgs a = base_args(z, (long long)IC * T, T, XS, (long long)ch0 * t_dec, t_dec, B, t_dec, lengths, F_IN_MASK);
    a.cbias = cond + m->off_dec; a.cbias_bs = m->cond_rows;
    a.range_flag = g_range_flag;
    HIP_TRY(launch_conv(m->conv_pre, a, s));
  }
  float in_scale = 1.f;
  int L = t_dec;
  for (int i = 0; i < c.n_upsamples; ++i) {
    // x = ups[i](leaky_relu(x, 0.1)); the division by num_kernels of the previous stage rides on the input scale.
    // Where the stage's ResBlocks read x in the staged operand format, the x8 / x4 transposed convs write that copy
    // themselves (same values as the conversion kernel: leaky-relu(0.1) of the fp32 result, then the split).
    const StagePlan plan = plan_stage(m, i, B, L * c.upsample_rates[i]);
    const bool ups_stages = ups_stage_enabled() && (m->ups[i].mode == MODE_UP8 || m->ups[i].mode == MODE_UP4) &&
                            ((plan.stage_x && m->ups[i].prec == PREC_F16X2) || (plan.stage_bf16 && m->ups[i].prec == PREC_BF16));
    TRY(conv_simple(m->ups[i], XS, X, B, L, nullptr, F_IN_LRELU, s, nullptr, nullptr, 0, nullptr, 0, in_scale, 0.1f, 0.1f,
                    ups_stages ? XG : nullptr));
    L *= c.upsample_rates[i];
    // The nk ResBlocks read the same x and are independent until their outputs are summed (models.py:292-298):
    // chain j runs on its own stream; its last conv accumulates into XS after chain j-1's has (same order of
    // additions as the reference: (rb0 + rb1) + rb2).
    // f16x2 arithmetic: every conv of an unfused pair takes its input in the staged operand format (leaky-relu applied,
    // split into fp16 h / l) -- x is converted once for all chains here, and the second conv of a pair writes its
    // result twice, as fp32 (the residual of the next pair) and staged (the input of the next pair's first conv).
    // Plain-bf16 arithmetic: the same where the stage's convs run on the LDS-DMA kernel (the one that writes the second
    // copy), i.e. for rows in 128s at sizes that fill the chip; decided per stage, for every chain at once.
    const bool stage_x = plan.stage_x, stage_bf16 = plan.stage_bf16;
    const bool single = c.resblock == 2;            // modules.ResBlock2 (reference modules.py:228-252)
    {
      const int64_t one = pad256((int64_t)(ch0 >> (i + 1)) * L * B) / 4;
      if (one * plan.total_bufs > scratch_floats) return fail(VITS_ERR_WORKSPACE, "generator: workspace too small");
      float* next = scratch;
      for (int j = 0; j < nk; ++j) {
        Y[j] = Tt[j] = YG[j] = nullptr;
        if (plan.bufs[j] >= 2) { Y[j] = next; Tt[j] = next + one; next += 2 * one; }
        if (plan.bufs[j] >= 3) { YG[j] = next; next += one; }
      }
    }
    if (stage_x && !ups_stages)
      HIP_TRY(launch_stage_f16x2(X, XG, B, m->rbs[i * nk].c1[0].Cin, L, 1.f, 0.1f, g_range_flag, s));
    if (stage_bf16 && !ups_stages)
      HIP_TRY(launch_stage_bf16(X, XG, B, m->rbs[i * nk].c1[0].Cin, L, 1.f, 0.1f, s));
    if (n_side > 0) {
      HIP_TRY(hipEventRecord(m->ev_fork, s));
      for (int j = 0; j < n_side; ++j) HIP_TRY(hipStreamWaitEvent(m->side[j], m->ev_fork, 0));
    }
    for (int j = 0; j < nk; ++j) {
      // One side stream: the last chain (the widest kernel, about as much work as the others together) keeps the
      // caller's stream and the others share the side stream; more: a stream per chain.
      hipStream_t sj = n_side == 0 ? s
                       : n_side + 1 >= nk ? (j == 0 ? s : m->side[j - 1])
                       : (j == nk - 1 ? s : m->side[std::min(j, n_side - 1)]);
      const ResBlock& rb = m->rbs[i * nk + j];
      const float* yin = X;
      const bool fuse_chain = plan.fuse_chain[j];
      const int nseg = plan.nseg[j];
      const int* seg = plan.seg[j];
      if (nseg > 0) {
        // the ResBlock in nseg launches (one: x is read once and the result goes straight into the stage's sum); the
        // carrier between two segments takes turns in Y and T
        int q0 = 0;
        for (int g = 0; g < nseg; ++g) {
          const bool last = g + 1 == nseg;
          float* dst = last ? XS : ((g & 1) ? Tt[j] : Y[j]);
          if (last && j > 0 && n_side > 0) HIP_TRY(hipStreamWaitEvent(sj, m->ev_done[j - 1], 0));
          PairArgs pa;
          pa.x = yin; pa.x_bs = (long long)rb.c1[0].Cin * L; pa.x_cs = L;
          pa.out = dst; pa.out_bs = (long long)rb.c1[0].Cin * L; pa.out_cs = L;
          pa.B = B; pa.L = L; pa.accumulate = last && j > 0; pa.slope = 0.1f;
          pa.range_flag = g_range_flag;
          // a segment of one pair: the pair kernel where it is the better one (32 channels; 64 in bf16), else the
          // whole-ResBlock kernel on that one pair (64 channels in f16x2: 384 against 499 us at 16 x 3 s, 7 taps)
          if (seg[g] == 1 && pair_profitable(rb.c1[q0], rb.c2[q0])) HIP_TRY(launch_resblock_pair(rb.c1[q0], rb.c2[q0], pa, sj));
          else HIP_TRY(launch_resblock_chain(rb.c1.data() + q0, rb.c2.data() + q0, seg[g], pa, sj));
          q0 += seg[g];
          yin = dst;
        }
        if (n_side > 0) HIP_TRY(hipEventRecord(m->ev_done[j], sj));
        continue;
      }
      for (int q = 0; q < nd; ++q) {
        if (single) {
          // x = c(leaky_relu(x)) + x per dilation; the carrier alternates between Y and T because a conv with a halo
          // cannot update its own input, and the last conv adds the chain into the sum of the stage
          const bool last = q + 1 == nd;
          float* dst = last ? XS : ((q & 1) ? Tt[j] : Y[j]);
          if (last && j > 0 && n_side > 0) HIP_TRY(hipStreamWaitEvent(sj, m->ev_done[j - 1], 0));
          TRY(conv_simple(rb.c1[q], yin, dst, B, L, nullptr, F_IN_LRELU | F_RES | (last && j > 0 ? F_ACCUM : 0), sj, yin,
                          nullptr, 0, nullptr, 0, 1.f, 0.1f, 0.1f));
          yin = dst;
          continue;
        }
        if (fuse_chain) {
          // the whole pair in one launch, the tensor between the convs never leaves LDS; Y and T take turns as the
          // chain's carrier
          const bool last = q + 1 == nd;
          float* dst = last ? XS : ((q & 1) ? Tt[j] : Y[j]);
          if (last && j > 0 && n_side > 0) HIP_TRY(hipStreamWaitEvent(sj, m->ev_done[j - 1], 0));
          PairArgs pa;
          pa.x = yin; pa.x_bs = (long long)rb.c1[q].Cin * L; pa.x_cs = L;
          pa.out = dst; pa.out_bs = (long long)rb.c1[q].Cin * L; pa.out_cs = L;
          pa.B = B; pa.L = L; pa.accumulate = last && j > 0; pa.slope = 0.1f;
          pa.range_flag = g_range_flag;
          HIP_TRY(launch_resblock_pair(rb.c1[q], rb.c2[q], pa, sj));
          yin = dst;
          continue;
        }
        // The tensor between the two convs of a pair has one reader: in the bf16 / f16x2 modes the first conv writes
        // it already leaky-relu'd and converted, in the second conv's operand format (same values, no conversion at
        // staging; half the bytes in bf16 mode).
        const int p1 = rb.c1[q].prec;
        const bool staged = staged_pairs_enabled() && p1 == rb.c2[q].prec && (p1 == PREC_BF16 || p1 == PREC_F16X2);
        // the first conv's input is staged as well
        const bool staged_in = staged && ((p1 == PREC_F16X2 && stage_x) || (p1 == PREC_BF16 && stage_bf16));
        const float* c1_in = staged_in ? (q == 0 ? XG : YG[j]) : yin;
        TRY(conv_simple(rb.c1[q], c1_in, Tt[j], B, L, nullptr,
                        (staged_in ? F_IN_STAGED : F_IN_LRELU) | (staged ? F_OUT_STAGED : 0), sj, nullptr, nullptr, 0,
                        nullptr, 0, 1.f, 0.1f, 0.1f));
        const bool last = q + 1 == nd;
        float* dst = last ? XS : Y[j];
        const int flags = (staged ? F_IN_STAGED : F_IN_LRELU) | F_RES | (last && j > 0 ? F_ACCUM : 0);
        if (last && j > 0 && n_side > 0) HIP_TRY(hipStreamWaitEvent(sj, m->ev_done[j - 1], 0));
        TRY(conv_simple(rb.c2[q], Tt[j], dst, B, L, nullptr, flags, sj, yin, nullptr, 0, nullptr, 0, 1.f, 0.1f, 0.1f,
                        staged_in && !last ? YG[j] : nullptr));
        yin = Y[j];
      }
      if (n_side > 0) HIP_TRY(hipEventRecord(m->ev_done[j], sj));
    }
    if (n_side > 0) HIP_TRY(hipStreamWaitEvent(s, m->ev_done[nk - 1], 0));
    in_scale = 1.f / (float)nk;
  }
  // tanh(conv_post(leaky_relu(x, 0.01)))   (reference models.py:299-301)
  HIP_TRY(launch_conv_to1(XS, (long long)(ch0 >> c.n_upsamples) * L, L, m->conv_post_w, nullptr, wav, nullptr, B,
                          ch0 >> c.n_upsamples, 7, L, F_IN_LRELU, in_scale, 0.01f, true, false, s));
  return VITS_OK;
}

void vits_set_front_end_f16x2(int mode) { set_alt_f16x2_mode(mode); }

// ---- per-kernel timing ----
void vits_profile_enable(int on) { prof_enable(on != 0); }
void vits_profile_reset(void) { prof_reset(); }

int vits_profile_read(int index, char* name, int name_len, double* flops, double* ms, int64_t* count) {
  static thread_local std::vector<ProfRow> rows;
  if (index < 0) return prof_collect(rows);
  if (index >= (int)rows.size() || !name || name_len <= 0) return fail(VITS_ERR_INVALID, "profile_read: bad index");
  snprintf(name, name_len, "%s", rows[index].name.c_str());
  if (flops) *flops = rows[index].flops;
  if (ms) *ms = rows[index].ms;
  if (count) *count = rows[index].count;
  return VITS_OK;
}

int vits_profile_read_bytes(int index, double* bytes) {
  std::vector<ProfRow> rows;
  const int n = prof_collect(rows);
  if (index < 0 || index >= n || !bytes) return fail(VITS_ERR_INVALID, "profile_read_bytes: bad index");
  *bytes = rows[index].bytes;
  return VITS_OK;
}

// ---- single operators ----
int vits_op_conv1d(const void* w, const float* bias, int cin, int rows, int taps, int dilation, int mode, int precision,
                   const float* x, int B, int L, const int* lengths, const float* cbias, const float* in_cbias,
                   const float* res, float* out, int flags, float in_scale, float in_slope, void* stream) {
  ConvDesc d;
  TRY(make_desc(d, w, bias, cin, rows, taps, dilation, mode, precision));
  return conv_simple(d, x, out, B, L, lengths, flags, (hipStream_t)stream, res, cbias, rows, in_cbias, cin, in_scale,
                     in_slope, in_slope);
}

int vits_op_conv1d_out2(const void* w, const float* bias, int cin, int rows, int taps, int dilation, int mode, int precision,
                        const float* x, int B, int L, const float* res, float* out, float* out2, int flags, float in_scale,
                        float in_slope, float out_slope, void* stream) {
  ConvDesc d;
  TRY(make_desc(d, w, bias, cin, rows, taps, dilation, mode, precision));
  return conv_simple(d, x, out, B, L, nullptr, flags, (hipStream_t)stream, res, nullptr, 0, nullptr, 0, in_scale, in_slope,
                     out_slope, out2);
}

int vits_op_resblock_pair(const void* w1, const float* b1, const void* w2, const float* b2, int channels, int taps,
                          int dilation, const float* x, float* out, int B, int L, int accumulate, float slope,
                          void* stream) {
  return vits_op_resblock_pair_prec(w1, b1, w2, b2, channels, taps, dilation, PREC_F16X2, x, out, B, L, accumulate, slope,
                                    stream);
}

int vits_op_resblock_pair_prec(const void* w1, const float* b1, const void* w2, const float* b2, int channels, int taps,
                               int dilation, int precision, const float* x, float* out, int B, int L, int accumulate,
                               float slope, void* stream) {
  if (precision != PREC_F16X2 && precision != PREC_BF16)
    return fail(VITS_ERR_INVALID, "resblock pair: precision must be f16x2 (3) or bf16 (2)");
  ConvDesc c1, c2;
  TRY(make_desc(c1, w1, b1, channels, channels, taps, dilation, MODE_PLAIN, precision));
  TRY(make_desc(c2, w2, b2, channels, channels, taps, 1, MODE_PLAIN, precision));
  if (!pair_fusable(c1, c2))
    return fail(VITS_ERR_INVALID, "resblock pair: 32 or 64 channels, 3 / 5 / 7 / 9 / 11 taps, first-conv halo <= 25 required");
  if (x == out) return fail(VITS_ERR_INVALID, "resblock pair: cannot run in place");
  PairArgs pa;
  pa.x = x; pa.x_bs = (long long)channels * L; pa.x_cs = L;
  pa.out = out; pa.out_bs = (long long)channels * L; pa.out_cs = L;
  pa.B = B; pa.L = L; pa.accumulate = accumulate; pa.slope = slope;
  pa.range_flag = g_range_flag;
  HIP_TRY(launch_resblock_pair(c1, c2, pa, (hipStream_t)stream));
  return VITS_OK;
}

int vits_op_resblock_chain(const void* const* w1, const float* const* b1, const void* const* w2, const float* const* b2,
                           int channels, int taps, const int* dilations, int npair, int precision, const float* x,
                           float* out, int B, int L, int accumulate, float slope, void* stream) {
  if (precision != PREC_F16X2 && precision != PREC_BF16)
    return fail(VITS_ERR_INVALID, "resblock chain: precision must be f16x2 (3) or bf16 (2)");
  if (!w1 || !w2 || !dilations || !x || !out || npair < 1 || npair > 4)
    return fail(VITS_ERR_INVALID, "resblock chain: null argument or more than 4 pairs");
  ConvDesc c1[4], c2[4];
  for (int p = 0; p < npair; ++p) {
    TRY(make_desc(c1[p], w1[p], b1 ? b1[p] : nullptr, channels, channels, taps, dilations[p], MODE_PLAIN, precision));
    TRY(make_desc(c2[p], w2[p], b2 ? b2[p] : nullptr, channels, channels, taps, 1, MODE_PLAIN, precision));
  }
  if (!chain_fusable(c1, c2, npair))
    return fail(VITS_ERR_INVALID, "resblock chain: 1 to 4 pairs of 32 or 64 channels (128 with 3 taps) and 3 / 5 / 7 / 9 / 11 taps whose "
                                  "halos add up to at most 64 steps a side required");
  if (x == out) return fail(VITS_ERR_INVALID, "resblock chain: cannot run in place");
  PairArgs pa;
  pa.x = x; pa.x_bs = (long long)channels * L; pa.x_cs = L;
  pa.out = out; pa.out_bs = (long long)channels * L; pa.out_cs = L;
  pa.B = B; pa.L = L; pa.accumulate = accumulate; pa.slope = slope;
  pa.range_flag = g_range_flag;
  HIP_TRY(launch_resblock_chain(c1, c2, npair, pa, (hipStream_t)stream));
  return VITS_OK;
}

int vits_op_layer_norm(const float* x, float* y, const float* gamma, const float* beta, const int* lengths, int B, int C,
                       int L, int mask_out, void* stream) {
  HIP_TRY(launch_layer_norm(x, y, gamma, beta, lengths, B, C, L, mask_out != 0, (hipStream_t)stream));
  return VITS_OK;
}

int vits_op_attention(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out, int B,
                      int H, int DK, int L, int window, void* stream) {
  HIP_TRY(launch_attention(qkv, emb_k, emb_v, lengths, out, B, H, DK, L, window, (hipStream_t)stream));
  return VITS_OK;
}

int vits_op_attention_prec(const float* qkv, const float* emb_k, const float* emb_v, const int* lengths, float* out, int B,
                           int H, int DK, int L, int window, int precision, void* stream) {
  if (precision != PREC_BF16X3 && precision != PREC_F16X2)
    return fail(VITS_ERR_INVALID, "attention: precision must be bf16x3 (1) or f16x2 (3)");
  HIP_TRY(launch_attention(qkv, emb_k, emb_v, lengths, out, B, H, DK, L, window, (hipStream_t)stream,
                           precision == PREC_F16X2, g_range_flag));
  return VITS_OK;
}

}  // extern "C"
