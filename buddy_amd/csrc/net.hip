// Host-side graph of the NCSN++ score network (reference networks/ncsnpp.py:281-449 + STFT/iSTFT wrapper :473-506)
// and its input-VJP, driving the gfx950 kernels of igemm.hip / ops.hip.  One call = one forward (or one VJP) of the
// whole network for a batch of utterances; activations live in one HBM arena owned by the handle (sized by a dry run).
// The VJP is a reverse "tape" of closures recorded during the forward (no weight gradients: inference only).
#include "common.h"
#include "net.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <unordered_map>

namespace buddy {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return BUDDY_ERR_HIP; } } while (0)

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
const char* last_error() { return g_err.c_str(); }

static const float INV_SQRT2 = 0.70710678118654752440f;

// ------------------------------------------------------------------------------------------------ parameter layout
struct PSpec { std::string name; std::vector<int> shape; long long off; long long numel; };

static long long numel_of(const std::vector<int>& s) { long long n = 1; for (int v : s) n *= v; return n; }

// Same enumeration as buddy_amd/synth.py:module_specs (reference construction order ncsnpp.py:157-274).
static std::vector<PSpec> build_specs(const NetCfg& c) {
  std::vector<PSpec> v;
  long long off = 0;
  int idx = 0;
  auto add = [&](const std::string& suffix, std::vector<int> shape) {
    PSpec s; s.name = "all_modules." + std::to_string(idx) + "." + suffix; s.shape = shape; s.off = off; s.numel = numel_of(shape);
    off += s.numel; v.push_back(s);
  };
  const int nf = c.nf, in_ch = 2;
  auto resblock = [&](int cin, int cout, bool resample) {
    add("GroupNorm_0.weight", {cin}); add("GroupNorm_0.bias", {cin});
    add("Conv_0.weight", {cout, cin, 3, 3}); add("Conv_0.bias", {cout});
    add("Dense_0.weight", {cout, nf * 4}); add("Dense_0.bias", {cout});
    add("GroupNorm_1.weight", {cout}); add("GroupNorm_1.bias", {cout});
    add("Conv_1.weight", {cout, cout, 3, 3}); add("Conv_1.bias", {cout});
    if (cin != cout || resample) { add("Conv_2.weight", {cout, cin, 1, 1}); add("Conv_2.bias", {cout}); }
    ++idx;
  };
  add("W", {nf}); ++idx;
  add("weight", {nf * 4, nf * 2}); add("bias", {nf * 4}); ++idx;
  add("weight", {nf * 4, nf * 4}); add("bias", {nf * 4}); ++idx;
  add("weight", {nf, in_ch, 3, 3}); add("bias", {nf}); ++idx;
  std::vector<int> hs_c{nf};
  int ch = nf;
  for (int l = 0; l < c.nlev; ++l) {
    for (int b = 0; b < c.nrb; ++b) { int co = nf * c.ch_mult[l]; resblock(ch, co, false); ch = co; hs_c.push_back(ch); }
    if (l != c.nlev - 1) {
      resblock(ch, ch, true);
      add("Conv_0.weight", {ch, in_ch, 1, 1}); add("Conv_0.bias", {ch}); ++idx;
      hs_c.push_back(ch);
    }
  }
  resblock(ch, ch, false);
  add("GroupNorm_0.weight", {ch}); add("GroupNorm_0.bias", {ch});
  for (int k = 0; k < 4; ++k) { add("NIN_" + std::to_string(k) + ".W", {ch, ch}); add("NIN_" + std::to_string(k) + ".b", {ch}); }
  ++idx;
  resblock(ch, ch, false);
  for (int l = c.nlev - 1; l >= 0; --l) {
    for (int b = 0; b < c.nrb + 1; ++b) { int co = nf * c.ch_mult[l]; resblock(ch + hs_c.back(), co, false); hs_c.pop_back(); ch = co; }
    add("weight", {ch}); add("bias", {ch}); ++idx;
    add("weight", {in_ch, ch, 3, 3}); add("bias", {in_ch}); ++idx;
    if (l != 0) resblock(ch, ch, true);
  }
  PSpec s; s.name = "output_layer.weight"; s.shape = {2, in_ch, 1, 1}; s.off = off; s.numel = 4; off += 4; v.push_back(s);
  PSpec t; t.name = "output_layer.bias"; t.shape = {2}; t.off = off; t.numel = 2; off += 2; v.push_back(t);
  return v;
}

long long param_count(const NetCfg& c) {
  auto v = build_specs(c);
  return v.back().off + v.back().numel;
}

// ------------------------------------------------------------------------------------------------ tensors / arena
struct Tens {
  float* p = nullptr; float* g = nullptr; int ginit = 0;
  int B = 0, H = 0, W = 0, C = 0;
  double* csum = nullptr; bool has_csum = false;   // per-(utterance, channel) (sum, sum of squares) left by the producer (GroupNorm statistics)
  long long numel() const { return (long long)B * H * W * C; }
};
struct View { Tens* a = nullptr; Tens* b = nullptr; int C() const { return a->C + (b ? b->C : 0); } };

struct Arena {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0; bool dry = false; bool overflow = false;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* r = dry ? (void*)(uintptr_t)(0x10000 + off) : (void*)(base + off);
    off += bytes;
    if (off > peak) peak = off;
    if (!dry && off > cap) overflow = true;
    return r;
  }
  float* f(long long n) { return (float*)alloc((size_t)n * 4); }
};

// One prepared operand form of a 3x3 convolution's weights: u = fp32 (direct / Winograd-domain), x = the bf16x3 stage image of u (wgemm.hip)
struct WVar { float* u = nullptr; void* x = nullptr; void* x2 = nullptr; };   // fp32 form, bf16x3 stage image, f16x2 stage image
// raw: the torch OIHW tensor on the device.  3x3 convolutions of the ResBlocks get their operand forms LAZILY (conv_weights below): one
// (direction, kernel variant, arithmetic) per layer is ever built for a given workload, on the GPU (wprep.hip).  wf / wb: forms prepared at
// creation (the small 2-channel convolutions, 1x1 convolutions)
struct ConvW { int cin = 0, cout = 0, taps = 0; const float* raw = nullptr; float* wf = nullptr; float* wb = nullptr; float* bias = nullptr;
               mutable WVar var[2][5]; };                      // [forward | data-gradient][direct, F(2x2), F(4x4), F(6x6), F(6x6) sub-pixel up form]
struct GNW { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct ResW { GNW gn0, gn1; ConvW c0, c1, c2; bool has_c2 = false; int cin = 0, cout = 0, dense_off = 0; };
struct AttnW { GNW gn; float* Wt[4]; float* Wn[4]; float* b[4]; int C = 0; };

struct W3Img { const void* img; int N, K; const void* img2 = nullptr; };      // img: bf16x3 stage image; img2: f16x2 stage image (general f16x2 form), where the shape allows
// Everything derived from the parameters: shared (read-only after preparation) by a handle and its replicas (net_replica)
struct Weights {
  NetCfg cfg;
  std::vector<PSpec> specs;
  float* dparams = nullptr;    // raw parameters (device)
  float* dpacked = nullptr;    // small packed weights (device): 2-channel convolutions, transposed 1x1 / NIN, Dense_0 stack, DFT bases
  // modules in execution order
  float* Wf = nullptr; float* lin1_w = nullptr; float* lin1_b = nullptr; float* lin2_w = nullptr; float* lin2_b = nullptr;
  float* dense_w = nullptr; float* dense_b = nullptr; int dense_total = 0;
  ConvW conv_in;                       // 2 -> nf (c2in fwd, c2out bwd)
  std::vector<ResW> res;               // in module order
  std::vector<ConvW> combine;          // 1x1 2 -> C
  AttnW attn;
  std::vector<GNW> pyr_gn; std::vector<ConvW> pyr_conv;   // C -> 2 heads, top level first
  float* out_w = nullptr; float* out_b = nullptr;
  float* basisF = nullptr; float* basisI = nullptr;      // [2Fb][Kp], [Kp][2Fb]
  unsigned char* dpacked3 = nullptr;   // bf16x3 stage images of the 1x1 / NIN weights
  std::unordered_map<const float*, W3Img> w3;   // 1x1 / NIN weights [N][K] (fp32, device) -> their bf16x3 stage image and the shape it was packed for
  // lazily prepared 3x3 operand forms
  std::mutex mu;
  std::vector<void*> lazy_allocs; size_t lazy_bytes = 0; int lazy_count = 0;
  float* prep_tmp = nullptr; size_t prep_cap = 0;        // fp32 staging of a Winograd-domain weight set on its way to the bf16x3 image
  ~Weights() {
    if (dparams) (void)hipFree(dparams);
    if (dpacked) (void)hipFree(dpacked);
    if (dpacked3) (void)hipFree(dpacked3);
    if (prep_tmp) (void)hipFree(prep_tmp);
    for (void* q : lazy_allocs) (void)hipFree(q);
  }
};

struct Net {
  NetCfg cfg;
  int Fb = 0, Kp = 0, pad = 0;
  std::shared_ptr<Weights> W;
  // per-(B,L) state
  int B = 0, L = 0, T = 0, Tp = 0, Lp = 0;
  float* inv_env = nullptr; int env_len = 0; int env_Tp = -1;
  Arena arena;
  std::deque<Tens> pool;
  std::vector<std::function<void()>> tape;
  std::vector<std::pair<int, Tens*>> taps;
  double* partial = nullptr; float* red = nullptr;       // reduction scratch
  bool have_tape = false;
  hipStream_t st = nullptr;
  // saved for vjp
  Tens* spec = nullptr; Tens* pyr0 = nullptr;
  const float* k_cin = nullptr; const float* k_cskip = nullptr; const float* k_cout = nullptr;

  int rsv_B = 0, rsv_L = 0, rsv_vjp = -1;   // shape the arena was last sized for (the sizing dry run is skipped while it still fits)
  Options opt;                 // this handle's launcher options (options.hip): attention core, GEMM arithmetic, every A/B switch; from the environment
                               // defaults at creation, changed with buddy_ncsnpp_set_option; the launchers read it through cur_opt() while a call runs
  bool prep_failed = false;    // a lazily prepared weight form could not be built (out of memory): the call reports it
  bool fir = false;            // fir=True: FIR (1,3,3,1) resampling instead of nearest / box (reference up_or_down_sampling.py:195-257)
  float* w4_scratch = nullptr; size_t w4_cap = 0, w4_need = 0;   // V / M buffers of the three-pass F(4x4,3x3) convolutions (floats)
  // gemm = f16x2: abs-max of V per (convolution of this call, utterance): vslots x B x [VMAX_SUB][VMAX_STRIDE] words (common.h), zeroed at the start of every call
  unsigned* vmax = nullptr; size_t vmax_cap = 0; int vslot = 0, vslots = 0, vdry = 0;   // vslots: convolutions per call counted by the sizing dry run
  int vslot_need[2] = {0, 0};  // slots a forward / an input-VJP call used on the reserved shape (0: not known yet, zero all of them)
  bool dry() const { return arena.dry; }
  Tens* mk(int B_, int H, int W, int C, bool grad) {
    pool.emplace_back();
    Tens* t = &pool.back();
    t->B = B_; t->H = H; t->W = W; t->C = C;
    t->p = arena.f(t->numel());
    if (grad) t->g = arena.f(t->numel());
    t->csum = (double*)arena.alloc((size_t)B_ * C * 2 * sizeof(double));
    return t;
  }
  float* tmp(long long n) { return arena.f(n); }
};

// ------------------------------------------------------------------------------------------------ weight packing (host)
namespace {
struct Packer {
  std::vector<float> buf;
  long long put(const std::vector<float>& v) {
    long long off = (long long)buf.size();
    buf.insert(buf.end(), v.begin(), v.end());
    while (buf.size() % 64) buf.push_back(0.f);
    return off;
  }
};

// conv 3x3 weights are torch OIHW with H = frequency (ky), W = time (kx).  Our spatial axes are swapped
// (H = time, W = frequency), so tap (dy, dx) of our layout reads W[o][i][ky = dx][kx = dy].
inline float w3(const float* w, int O, int I, int o, int i, int dy, int dx) { return w[(((long long)o * I + i) * 3 + dx) * 3 + dy]; }

std::vector<float> transpose2(const float* w, int R, int Cc) {   // [R][Cc] -> [Cc][R]
  std::vector<float> r((size_t)R * Cc);
  for (int i = 0; i < R; ++i) for (int j = 0; j < Cc; ++j) r[(size_t)j * R + i] = w[(size_t)i * Cc + j];
  return r;
}
}  // namespace

static const PSpec* find_spec(const std::vector<PSpec>& v, const std::string& n) {
  for (auto& s : v) if (s.name == n) return &s;
  return nullptr;
}


// Build the shared weight store: the raw parameters go to the device once (every tensor start aligned to 256 B); the host prepares only the
// small operands (2-channel convolutions, transposed 1x1 / NIN matrices, the Dense_0 stack, the windowed DFT bases: ~4 M floats).  The 3x3
// convolutions -- 98 % of the parameters -- are NOT touched here: conv_weights() derives the one operand form a layer needs on first use.
static int weights_create(const float* hp, long long n, const NetCfg& cfg, std::shared_ptr<Weights>* out) {
  auto Wp = std::make_shared<Weights>();
  Weights* N = Wp.get();
  N->cfg = cfg;
  N->specs = build_specs(cfg);
  if (n != param_count(cfg)) { set_error("parameter blob size mismatch"); return BUDDY_ERR_ARG; }
  if (cfg.n_fft % 2) { set_error("n_fft must be even"); return BUDDY_ERR_ARG; }
  {  // the GroupNorm partial-sum scratch is sized for at most 1024 channels; the widest tensor is a skip concatenation (2 x the level width)
    int cmax = 0;
    for (int l = 0; l < cfg.nlev; ++l) cmax = std::max(cmax, cfg.nf * cfg.ch_mult[l]);
    if (2 * cmax > 1024) { set_error("nf * max(ch_mult) must be <= 512 (concatenated skip tensors of up to 1024 channels)"); return BUDDY_ERR_ARG; }
  }
  // GroupNorm kernels (stand-alone and fused) normalise float4 channel quads with ONE group's statistics: min(C / 4, 32) groups must give a
  // multiple of four channels per group for EVERY normalised tensor, skip concatenations included -- C <= 128 (four per group) or C % 128 == 0.
  // nf = 32 and nf = 128 with the reference's multipliers satisfy it; e.g. nf = 64, ch_mult (1,2,2,2) has a 192-channel concatenation (six per group)
  for (const PSpec& ps : N->specs) {
    if (ps.name.find("GroupNorm_") == std::string::npos) continue;        // the pyramid GroupNorms have the widths of the blocks' GroupNorm_1
    const int C = ps.shape[0];
    if (C % 4 || (C > 128 && C % 128)) {
      set_error("unsupported width: a GroupNorm over " + std::to_string(C) + " channels (" + ps.name + ") has " + std::to_string(C / (C / 4 < 32 ? C / 4 : 32)) +
                " channels per group; the kernels need a multiple of 4 (C <= 128 or C % 128 == 0)");
      return BUDDY_ERR_ARG;
    }
  }
  const int Fb = cfg.n_fft / 2 + 1, Kp = (cfg.n_fft + 3) / 4 * 4;
  if (Fb % (1 << (cfg.nlev - 1))) { set_error("frequency bins not divisible by 2^(levels-1)"); return BUDDY_ERR_ARG; }
  std::vector<long long> doff(N->specs.size());
  {
    long long o = 0;
    for (size_t i = 0; i < N->specs.size(); ++i) { doff[i] = o; o += (N->specs[i].numel + 63) / 64 * 64; }
    HIPCHK(hipMalloc(&N->dparams, (size_t)o * 4));
    HIPCHK(hipMemsetAsync(N->dparams, 0, (size_t)o * 4, nullptr));
    // consecutive tensors whose padded and source offsets advance together go up as one copy
    size_t i = 0;
    while (i < N->specs.size()) {
      size_t j = i; long long len = N->specs[i].numel;
      while (j + 1 < N->specs.size() && N->specs[j].numel % 64 == 0) { ++j; len += N->specs[j].numel; }
      HIPCHK(hipMemcpyAsync(N->dparams + doff[i], hp + N->specs[i].off, (size_t)len * 4, hipMemcpyHostToDevice, nullptr));
      i = j + 1;
    }
  }

  Packer pk;
  struct Fix { float** dst; long long off; };
  std::vector<Fix> fixes;
  std::vector<std::pair<float**, std::pair<int, int>>> plain3;               // 1x1 / NIN weights [N][K]: registered in N->w3 by pointer
  auto raw = [&](const std::string& name) -> float* {
    for (size_t i = 0; i < N->specs.size(); ++i) if (N->specs[i].name == name) return N->dparams + doff[i];
    return nullptr;
  };
  auto host = [&](const std::string& name) -> const float* {
    const PSpec* s = find_spec(N->specs, name);
    return s ? hp + s->off : nullptr;
  };
  auto packed = [&](float** dst, const std::vector<float>& v) { fixes.push_back({dst, pk.put(v)}); };

  const int nf = cfg.nf;
  int idx = 0;
  auto pre = [&]() { return "all_modules." + std::to_string(idx) + "."; };
  std::vector<float> dense_w_all, dense_b_all;
  auto load_gn = [&](GNW& g, const std::string& p, int C) { g.gamma = raw(p + ".weight"); g.beta = raw(p + ".bias"); g.C = C; };
  auto load_conv3 = [&](ConvW& c, const std::string& p, int cin, int cout) {
    c.cin = cin; c.cout = cout; c.taps = 9; c.bias = raw(p + ".bias"); c.raw = raw(p + ".weight");
  };
  auto load_res = [&](int cin, int cout, bool resample) {
    ResW r; r.cin = cin; r.cout = cout;
    const std::string p = pre();
    load_gn(r.gn0, p + "GroupNorm_0", cin); load_gn(r.gn1, p + "GroupNorm_1", cout);
    N->res.push_back(r);
    ResW& R = N->res.back();
    load_conv3(R.c0, p + "Conv_0", cin, cout);
    load_conv3(R.c1, p + "Conv_1", cout, cout);
    R.dense_off = (int)dense_b_all.size();
    const float* dw = host(p + "Dense_0.weight"); const float* db = host(p + "Dense_0.bias");
    dense_w_all.insert(dense_w_all.end(), dw, dw + (size_t)cout * nf * 4);
    dense_b_all.insert(dense_b_all.end(), db, db + cout);
    R.has_c2 = (cin != cout) || resample;
    if (R.has_c2) {
      R.c2.cin = cin; R.c2.cout = cout; R.c2.taps = 1; R.c2.bias = raw(p + "Conv_2.bias");
      R.c2.wf = raw(p + "Conv_2.weight");                                  // [cout][cin] already k-contiguous
      packed(&R.c2.wb, transpose2(host(p + "Conv_2.weight"), cout, cin));  // [cin][cout]
      plain3.push_back({&R.c2.wf, {cout, cin}}); plain3.push_back({&R.c2.wb, {cin, cout}});
    }
    ++idx;
  };
  N->res.reserve(64);
  N->Wf = raw(pre() + "W"); ++idx;
  N->lin1_w = raw(pre() + "weight"); N->lin1_b = raw(pre() + "bias"); ++idx;
  N->lin2_w = raw(pre() + "weight"); N->lin2_b = raw(pre() + "bias"); ++idx;
  {  // input conv 2 -> nf
    const float* w = host(pre() + "weight");
    N->conv_in.cin = 2; N->conv_in.cout = nf; N->conv_in.taps = 9; N->conv_in.bias = raw(pre() + "bias");
    std::vector<float> f((size_t)nf * 18), b((size_t)9 * nf * 2);
    for (int o = 0; o < nf; ++o) for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) for (int i = 0; i < 2; ++i) {
      f[((size_t)o * 9 + dy * 3 + dx) * 2 + i] = w3(w, nf, 2, o, i, dy, dx);
      b[((size_t)(dy * 3 + dx) * nf + o) * 2 + i] = w3(w, nf, 2, o, i, 2 - dy, 2 - dx);
    }
    packed(&N->conv_in.wf, f); packed(&N->conv_in.wb, b);
    ++idx;
  }
  std::vector<int> hs_c{nf};
  int ch = nf;
  for (int l = 0; l < cfg.nlev; ++l) {
    for (int b = 0; b < cfg.nrb; ++b) { int co = nf * cfg.ch_mult[l]; load_res(ch, co, false); ch = co; hs_c.push_back(ch); }
    if (l != cfg.nlev - 1) {
      load_res(ch, ch, true);
      ConvW c; c.cin = 2; c.cout = ch; c.taps = 1; c.bias = raw(pre() + "Conv_0.bias");
      c.wf = raw(pre() + "Conv_0.weight"); c.wb = c.wf;   // [C][2]: both kernels index it the same way for one tap
      N->combine.push_back(c); ++idx;
      hs_c.push_back(ch);
    }
  }
  load_res(ch, ch, false);
  {
    AttnW& a = N->attn; a.C = ch;
    const std::string p = pre();
    load_gn(a.gn, p + "GroupNorm_0", ch);
    for (int k = 0; k < 4; ++k) {
      a.Wn[k] = raw(p + "NIN_" + std::to_string(k) + ".W"); a.b[k] = raw(p + "NIN_" + std::to_string(k) + ".b");
      packed(&a.Wt[k], transpose2(host(p + "NIN_" + std::to_string(k) + ".W"), ch, ch));
      plain3.push_back({&a.Wt[k], {ch, ch}}); plain3.push_back({&a.Wn[k], {ch, ch}});
    }
    ++idx;
  }
  load_res(ch, ch, false);
  N->pyr_gn.resize(cfg.nlev); N->pyr_conv.resize(cfg.nlev);
  for (int l = cfg.nlev - 1, j = 0; l >= 0; --l, ++j) {
    for (int b = 0; b < cfg.nrb + 1; ++b) { int co = nf * cfg.ch_mult[l]; load_res(ch + hs_c.back(), co, false); hs_c.pop_back(); ch = co; }
    load_gn(N->pyr_gn[j], "all_modules." + std::to_string(idx), ch); ++idx;
    {
      const float* w = host(pre() + "weight");   // [2][ch][3][3]
      ConvW& c = N->pyr_conv[j]; c.cin = ch; c.cout = 2; c.taps = 9; c.bias = raw(pre() + "bias");
      std::vector<float> f((size_t)9 * ch * 2), b((size_t)ch * 18);
      for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) for (int i = 0; i < ch; ++i) for (int o = 0; o < 2; ++o) {
        f[((size_t)(dy * 3 + dx) * ch + i) * 2 + o] = w3(w, 2, ch, o, i, dy, dx);
        b[((size_t)i * 9 + dy * 3 + dx) * 2 + o] = w3(w, 2, ch, o, i, 2 - dy, 2 - dx);
      }
      packed(&c.wf, f); packed(&c.wb, b);
      ++idx;
    }
    if (l != 0) load_res(ch, ch, true);
  }
  N->out_w = raw("output_layer.weight"); N->out_b = raw("output_layer.bias");
  N->dense_total = (int)dense_b_all.size();
  packed(&N->dense_w, dense_w_all); packed(&N->dense_b, dense_b_all);

  // DFT bases with the periodic Hann window folded in (reference ncsnpp.py:464,473-496 via the stft / istft of PyTorch).
  {
    const int nfft = cfg.n_fft;
    std::vector<float> bf((size_t)2 * Fb * Kp, 0.f), bi((size_t)Kp * 2 * Fb, 0.f);
    const double PI = 3.14159265358979323846;
    std::vector<double> ct(nfft), sn(nfft);                   // cos / sin of 2 pi j / nfft: one evaluation per angle instead of Fb per angle
    for (int j = 0; j < nfft; ++j) { ct[j] = std::cos(2.0 * PI * (double)j / nfft); sn[j] = std::sin(2.0 * PI * (double)j / nfft); }
    for (int k = 0; k < nfft; ++k) {
      const double w = 0.5 - 0.5 * ct[k];
      for (int f = 0; f < Fb; ++f) {
        const int j = (int)((long long)f * k % nfft);
        bf[((size_t)f * 2 + 0) * Kp + k] = (float)(w * ct[j]);
        bf[((size_t)f * 2 + 1) * Kp + k] = (float)(-w * sn[j]);
        const double cf = (f == 0 || f == nfft / 2) ? 1.0 : 2.0;
        bi[(size_t)k * 2 * Fb + f * 2 + 0] = (float)(cf / nfft * ct[j] * w);
        bi[(size_t)k * 2 * Fb + f * 2 + 1] = (float)(-cf / nfft * sn[j] * w);
      }
    }
    packed(&N->basisF, bf); packed(&N->basisI, bi);
  }
  HIPCHK(hipMalloc(&N->dpacked, pk.buf.size() * 4));
  HIPCHK(hipMemcpyAsync(N->dpacked, pk.buf.data(), pk.buf.size() * 4, hipMemcpyHostToDevice, nullptr));
  for (auto& f : fixes) *f.dst = N->dpacked + f.off;
  std::vector<size_t> plain_off;
  size_t pack3_bytes = 0;
  for (auto& j : plain3) {
    plain_off.push_back(pack3_bytes);
    if (wgemm_supported(j.second.first, j.second.second)) pack3_bytes += (wgemm_packed_bytes(1, j.second.first, j.second.second) + 255) / 256 * 256;
  }
  std::vector<size_t> plain_off2;
  size_t pack2_bytes = 0;
  for (auto& j : plain3) {
    plain_off2.push_back(pack2_bytes);
    if (wgemm_f16x2_supported(j.second.first, j.second.second)) pack2_bytes += (wgemm_f16x2_packed_bytes(1, j.second.first, j.second.second) + 255) / 256 * 256;
  }
  if (pack3_bytes) {     // the 1x1 / NIN matrices once more, split into three bf16 planes in the GEMM's LDS stage order (+ the two-term f16 image of the f16x2 general form)
    HIPCHK(hipMalloc(&N->dpacked3, pack3_bytes + pack2_bytes));
    for (size_t i = 0; i < plain3.size(); ++i) {
      const int n3 = plain3[i].second.first, k3 = plain3[i].second.second;
      if (!wgemm_supported(n3, k3) || *plain3[i].first == nullptr) continue;
      wgemm_pack_weights(*plain3[i].first, N->dpacked3 + plain_off[i], 1, n3, k3, nullptr);
      W3Img im{N->dpacked3 + plain_off[i], n3, k3};
      if (wgemm_f16x2_supported(n3, k3)) {
        wgemm_f16x2_pack_weights(*plain3[i].first, N->dpacked3 + pack3_bytes + plain_off2[i], 1, n3, k3, nullptr);
        im.img2 = N->dpacked3 + pack3_bytes + plain_off2[i];
      }
      N->w3[*plain3[i].first] = im;
    }
  }
  HIPCHK(hipDeviceSynchronize());     // the host staging buffers go out of scope here
  *out = Wp;
  return BUDDY_OK;
}

static int net_from_weights(std::shared_ptr<Weights> Wp, Net** out) {
  Net* N = new Net();
  N->W = std::move(Wp);
  N->cfg = N->W->cfg;
  if (int rc = options_check()) { delete N; return rc; }      // an unknown BUDDY_* variable or a bad value fails loudly here
  N->opt = default_options();
  N->Fb = N->cfg.n_fft / 2 + 1;
  N->Kp = (N->cfg.n_fft + 3) / 4 * 4;
  N->pad = N->cfg.n_fft / 2;
  *out = N;
  return BUDDY_OK;
}

int net_create(const float* hp, long long n, const NetCfg& cfg, Net** out) {
  std::shared_ptr<Weights> Wp;
  if (int rc = weights_create(hp, n, cfg, &Wp)) return rc;
  return net_from_weights(std::move(Wp), out);
}

// A second handle on the SAME prepared weights (read-only, reference-counted): own activation arena, VJP tape and settings.  Costs nothing but
// the arena of its first forward -- concurrent sub-batches (buddy_amd/testing/concurrent.py) and per-stream handles share one weight store.
int net_replica(Net* src, Net** out) {
  Net* N = nullptr;
  if (int rc = net_from_weights(src->W, &N)) return rc;
  N->opt = src->opt; N->fir = src->fir;
  *out = N;
  return BUDDY_OK;
}

int net_weight_bytes(Net* N, long long* params, long long* packed, long long* lazy, int* lazy_forms) {
  Weights* Wt = N->W.get();
  std::lock_guard<std::mutex> lk(Wt->mu);
  long long o = 0;
  for (auto& s : Wt->specs) o += (s.numel + 63) / 64 * 64;
  if (params) *params = o * 4;
  long long pk = 0;
  for (auto& kv : Wt->w3) pk += (long long)wgemm_packed_bytes(1, kv.second.N, kv.second.K);
  if (packed) *packed = pk;     // bf16x3 images of the 1x1 / NIN matrices (the small fp32 packs are < 20 MB and not counted separately)
  if (lazy) *lazy = (long long)Wt->lazy_bytes + (long long)Wt->prep_cap * 4;
  if (lazy_forms) *lazy_forms = Wt->lazy_count;
  return BUDDY_OK;
}

int net_set_option(Net* N, const char* key, int value) {
  if (int rc = option_set(N->opt, key, value)) return rc;
  N->rsv_vjp = -1;             // options change which temporaries a call allocates: size the arena again
  // ... and which kernels the recorded backward closures would launch against a tape and arena built under the old ones (abs-max slots, attention
  // workspace, fused forms): a saved forward does not survive an option change -- vjp reports BUDDY_ERR_STATE until the next forward(save = 1)
  N->tape.clear(); N->have_tape = false;
  return BUDDY_OK;
}
int net_get_option(Net* N, const char* key, int* value) { return option_get(N->opt, key, value); }
int net_set_attention(Net* N, int mode) { if (mode < 0 || mode > 4) { set_error("attention mode must be 0..4"); return BUDDY_ERR_ARG; } return net_set_option(N, "attention", mode); }
int net_set_gemm(Net* N, int mode) { if (mode < 0 || mode > 2) { set_error("gemm mode must be 0 (fp32 MFMA), 1 (bf16x3) or 2 (f16x2)"); return BUDDY_ERR_ARG; } return net_set_option(N, "gemm", mode); }
int net_set_fir(Net* N, int fir) { N->fir = fir != 0; N->rsv_vjp = -1; N->tape.clear(); N->have_tape = false; return BUDDY_OK; }
void net_destroy(Net* N) {
  if (!N) return;
  if (N->w4_scratch) (void)hipFree(N->w4_scratch);
  if (N->vmax) (void)hipFree(N->vmax);
  if (N->arena.base) (void)hipFree(N->arena.base);
  if (N->inv_env) (void)hipFree(N->inv_env);
  delete N;
}

// The operand form `kind` (0 direct, 2 / 4 / 6 = Winograd F(kind x kind, 3x3), 61 = F(6x6,3x3) of the sub-pixel up form) of a 3x3 convolution for one direction, built on first use on
// the GPU from the raw OIHW tensor (wprep.hip) and cached in the shared store.  want_x: 1 the bf16x3 / 2 the f16x2 stage image (the fp32 form is then only a
// staging buffer, reused for the next layer); 0: the fp32 form itself is kept.  The preparing stream is drained before the pointer is
// published, so a replica on another stream may use it at once.
static const WVar* conv_weights(Net* N, const ConvW& c, bool dgrad, int kind, int want_x) {
  Weights* Wt = N->W.get();
  const int ki = kind == 0 ? 0 : kind == 2 ? 1 : kind == 4 ? 2 : kind == 6 ? 3 : 4;
  WVar& v = c.var[dgrad ? 1 : 0][ki];
  std::lock_guard<std::mutex> lk(Wt->mu);
  if (want_x == 2 ? v.x2 != nullptr : want_x ? v.x != nullptr : v.u != nullptr) return &v;
  const int ph = kind == 61 ? 4 : 1;                          // sub-pixel up form: four phase kernels per output channel (wprep.hip)
  const int Co = dgrad ? c.cin : ph * c.cout, Ci = dgrad ? ph * c.cout : c.cin;
  const size_t nfl = (size_t)conv3_weight_floats(c.cout, c.cin, kind);
  hipStream_t st = N->st;
  float* u = v.u;
  if (u == nullptr) {
    if (want_x) {
      if (Wt->prep_cap < nfl) {
        if (Wt->prep_tmp) (void)hipFree(Wt->prep_tmp);
        Wt->prep_tmp = nullptr; Wt->prep_cap = 0;
        if (hipMalloc(&Wt->prep_tmp, nfl * 4) != hipSuccess) { N->prep_failed = true; set_error("out of memory preparing convolution weights"); return nullptr; }
        Wt->prep_cap = nfl;
      }
      u = Wt->prep_tmp;
    } else {
      if (hipMalloc(&u, nfl * 4) != hipSuccess) { N->prep_failed = true; set_error("out of memory preparing convolution weights"); return nullptr; }
      Wt->lazy_allocs.push_back(u); Wt->lazy_bytes += nfl * 4;
    }
    if (launch_conv3_weight_prep(c.raw, c.cout, c.cin, dgrad, kind, u, st) != BUDDY_OK) { N->prep_failed = true; set_error("bad weight form"); return nullptr; }
  }
  void* x = nullptr;
  if (want_x) {
    const int P = kind == 4 ? 36 : 64;
    const size_t bytes = want_x == 2 ? wgemm_f16x2_packed_bytes(P, Co, Ci) : wgemm_packed_bytes(P, Co, Ci);
    if (hipMalloc(&x, bytes) != hipSuccess) { N->prep_failed = true; set_error("out of memory preparing convolution weights"); return nullptr; }
    Wt->lazy_allocs.push_back(x); Wt->lazy_bytes += bytes;
    if (want_x == 2) wgemm_f16x2_pack_weights(u, x, P, Co, Ci, st);
    else wgemm_pack_weights(u, x, P, Co, Ci, st);
  }
  (void)hipStreamSynchronize(st);
  if (want_x == 2) v.x2 = x; else if (want_x) v.x = x; else v.u = u;
  ++Wt->lazy_count;
  return &v;
}

// ------------------------------------------------------------------------------------------------ op helpers
static Src2 src_of(const View& v) {
  Src2 s; s.p0 = v.a->p; s.p1 = v.b ? v.b->p : nullptr; s.C0 = v.a->C; s.ld0 = v.a->C; s.ld1 = v.b ? v.b->C : 0;
  return s;
}
static Dst2 gdst_of(const View& v) {
  Dst2 d; d.p0 = v.a->g; d.p1 = v.b ? v.b->g : nullptr; d.C0 = v.a->C; d.ld0 = v.a->C; d.ld1 = v.b ? v.b->C : 0;
  d.acc0 = v.a->ginit; d.acc1 = v.b ? v.b->ginit : 0;
  v.a->ginit = 1; if (v.b) v.b->ginit = 1;
  return d;
}
static IgemmParams ig_base() {
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = 1; p.H = 1; p.W = 1;
  return p;
}
static inline int gn_groups(int C) { int g = C / 4; return g < 32 ? g : 32; }

// conv3x3 over an NHWC tensor -> out, with the optional GroupNorm fusions of the three-pass Winograd paths
struct Conv3 {
  const float* a = nullptr;                 // input (B, H, W, Cin); unused when gn describes the input
  int B = 0, H = 0, W = 0, Cin = 0, Cout = 0;
  const ConvW* w = nullptr; bool dgrad = false;   // weights; dgrad: the data-gradient operands (Cin / Cout are those of THIS convolution)
  const float* bias = nullptr; const float* bias_bn = nullptr; int ld_bn = 0;   // per-channel bias, per-(utterance, channel) bias (time embedding)
  const float* res = nullptr; int ldRes = 0; int res_mode = 0;                  // residual: 1 same pixel, 2 nearest-upsampled (H/2 x W/2)
  float alpha = 1.f, out_scale = 1.f;
  float* out = nullptr;
  // gn / gn_tmp: the input is act(GroupNorm(gn->x)) -- or, with gn->da, that GroupNorm's BACKWARD applied to the gradient gn->da; the three-pass
  // paths evaluate it inside their input transform (the backward form: F(6x6,3x3) only), any other path materialises it into gn_tmp first
  const W4Gn* gn = nullptr; float* gn_tmp = nullptr;
  // stat_out: the tensor `out` belongs to -- its per-channel sums are left by the output transform where the shape allows.  direct: the only
  // reader of those statistics follows immediately: leave the partials in N->partial and return their chunk count instead of collapsing them
  Tens* stat_out = nullptr; bool direct = false;
  // bwd_gn (data-gradient convolutions): `out` is the gradient w.r.t. act(GroupNorm(bwd_gn->x)); on the F(6x6,3x3) path the output transform
  // leaves that GroupNorm's backward-sum partials in N->partial and the call returns their chunk count (0: run the reduction pass)
  const W4Gn* bwd_gn = nullptr;
  // up (sub-pixel forms, launch_wino6): 1 = out (2H, 2W, Cout) = conv3x3(nearest-upsample x2 of the (H, W, Cin) input); 2 = its data-gradient: the
  // input is the (2H, 2W, Cin) gradient, out is (H, W, Cout).  H, W are the LOW resolution in both.  The caller checks conv3_up_ok first.
  int up = 0;
};
// The BigGAN up block's Conv_0 (layerspp.py:246-257: h = upsample(act(GroupNorm_0(x))), Conv_0(h)) as one three-pass convolution on the low-resolution
// grid: no upsampled activation in HBM, V 4x smaller, the data-gradient's M and da 4x smaller.  Needs the F(6x6,3x3) path with the GroupNorm
// fusions on the low-resolution geometry.
// the next convolution's abs-max slots (one per utterance; zeroed by begin_call)
static unsigned* vmax_slot(Net* N, int B) {
  if (N->vmax == nullptr || N->vslot >= N->vslots || (size_t)(N->vslot + 1) * B * VMAX_SUB * VMAX_STRIDE > N->vmax_cap) { N->prep_failed = true; set_error("abs-max slots of the f16x2 GEMM exhausted"); return nullptr; }
  return N->vmax + (size_t)(N->vslot++) * B * VMAX_SUB * VMAX_STRIDE;
}
static bool conv3_up_ok(Net* N, int B, int H, int W, int Cin, int Cout) {
  if (H < 7 || W < 7) return false;
  const Options& o = N->opt;
  const bool on = o.conv == 0 && o.gn_fuse && o.gn_fuse_bwdin && o.gn_fuse_bwd && o.upconv;     // A/B switches
  if (!on || Cin % 8 || Cout % 8 || H < 6 || W < 6) return false;
  IgemmParams p = ig_base();
  p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.ldC = Cout; p.ld_bias_bn = 4;
  if (!wino6_supported(p) || !wino6_pays(p)) return false;
  const int sc = wino6_stat_chunks(p, 1);
  return sc > 0 && (long long)sc * Cout <= 256LL * 1024 && (long long)wino6_stat_chunks(p, 0) * Cin <= 256LL * 1024;
}
static int conv3(Net* N, const Conv3& c) {
  const float* a = c.a; const int B = c.B, H = c.H, W = c.W, Cin = c.Cin, Cout = c.Cout;
  if (N->prep_failed) return -1;      // an earlier convolution of this call could not be prepared: launch nothing more on its unwritten output, the call reports the error
  if (N->dry()) ++N->vdry;            // convolutions per call: the abs-max slots of the f16x2 GEMM are sized from this count
  // Winograd forms exist for channel counts that are multiples of 8 (every ResBlock convolution of the supported family)
  const bool wino_ok = c.w != nullptr && c.w->raw != nullptr && Cin % 8 == 0 && Cout % 8 == 0;
  const W4Gn* gn = c.gn; float* gn_tmp = c.gn_tmp; Tens* stat_out = c.stat_out; const W4Gn* bwd_gn = c.bwd_gn; const bool direct = c.direct;
  // option conv = 1 direct | 2 wino2 | 3 wino4 | 0 (default) three-pass F(6x6,3x3) on the large layers, three-pass F(4x4,3x3) where the shape allows,
  // else fused F(2x2,3x3), else direct
  const int cmode = N->opt.conv;
  const bool use_wino = cmode != 1, use_wino4 = cmode != 1 && cmode != 2, use_wino6 = use_wino4 && cmode != 3;
  if (N->dry()) {
    if (use_wino4 && wino_ok && H % 4 == 0 && W % 4 == 0) {
      const size_t need = (size_t)36 * ((size_t)B * H * W / 16) * (size_t)(Cin + Cout);
      if (need > N->w4_need) N->w4_need = need;
    }
    if (use_wino6 && wino_ok && H >= 6 && W >= 6) {
      const size_t th = c.up == 1 ? H / 7 + 1 : c.up == 2 ? (H + 6) / 7 : (H + 5) / 6, tw = c.up == 1 ? W / 7 + 1 : c.up == 2 ? (W + 6) / 7 : (W + 5) / 6;
      const size_t need = (size_t)64 * ((size_t)B * th * tw) * (size_t)((c.up == 2 ? 4 : 1) * Cin + (c.up == 1 ? 4 : 1) * Cout);
      if (need > N->w4_need) N->w4_need = need;
    }
    return 0;
  }
  if (c.up) {
    IgemmParams p = ig_base();
    p.A0 = a; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = c.out; p.ldC = Cout;
    p.bias_n = c.bias; p.bias_bn = c.bias_bn; p.ld_bias_bn = c.ld_bn; p.rows_per_batch = H * W; p.alpha = c.alpha; p.out_scale = c.out_scale;
    int x3 = wgemm_supported((c.up == 1 ? 4 : 1) * Cout, (c.up == 2 ? 4 : 1) * Cin) ? N->opt.gemm : 0;     // 1 bf16x3, 2 f16x2
    if (x3 == 2 && !wgemm_f16x2_supported((c.up == 1 ? 4 : 1) * Cout, (c.up == 2 ? 4 : 1) * Cin)) x3 = 1;
    const WVar* wv = conv_weights(N, *c.w, c.dgrad, 61, x3);
    unsigned* vm = x3 == 2 ? vmax_slot(N, B) : nullptr;
    if (x3 == 2 && !vm) return -1;
    if (!wv || N->w4_scratch == nullptr) { if (wv) { N->prep_failed = true; set_error("convolution scratch buffer missing"); } return -1; }
    long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf, c.up);
    const bool want_bwd = c.up == 2 && bwd_gn != nullptr;
    const int sc = (c.up == 1 ? stat_out != nullptr : want_bwd) ? wino6_stat_chunks(p, c.up) : 0;
    const bool stat = sc > 0 && (long long)sc * Cout <= 256LL * 1024;
    IgemmParams pr = p; pr.M = 4 * p.M;                       // the direct-convolution work is that of the (2H, 2W) grid
    const double xr = wino6_exec_ratio(p, c.up);
    igemm_prof_record(pr, 9, 1, N->st, true, xr);
    launch_wino6(p, wv->u, N->w4_scratch, N->w4_scratch + vf, N->st, gn, stat ? N->partial : nullptr, (stat && want_bwd) ? bwd_gn : nullptr,
                 x3 == 2 ? wv->x2 : x3 ? wv->x : nullptr, c.up, x3, vm);
    igemm_prof_record(pr, 9, 1, N->st, false, xr);
    if (stat && (want_bwd || direct)) return sc;
    if (stat && stat_out) { launch_csum_collapse(N->partial, sc, B, Cout, stat_out->csum, N->st); stat_out->has_csum = true; }
    return 0;
  }
  IgemmParams p = ig_base();
  p.A0 = a; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout;
  p.Bt = nullptr; p.ldB = 9 * Cin; p.C = c.out; p.ldC = Cout;
  p.bias_n = c.bias; p.bias_bn = c.bias_bn; p.ld_bias_bn = c.ld_bn; p.rows_per_batch = H * W;
  p.res = c.res; p.ldRes = c.ldRes; p.res_mode = c.res_mode; p.alpha = c.alpha; p.out_scale = c.out_scale;
  const bool fuse_gn = N->opt.gn_fuse != 0;
  const bool w6 = use_wino6 && wino_ok && N->w4_scratch != nullptr && wino6_supported(p) && wino6_pays(p);
  const bool w4 = !w6 && use_wino4 && wino_ok && N->w4_scratch != nullptr && wino4_supported(p);
  const bool fuse_bwd_in = N->opt.gn_fuse_bwdin != 0;
  if (gn != nullptr && gn->da != nullptr && !(w6 && fuse_gn && fuse_bwd_in)) {       // GroupNorm backward as the input: only F(6x6,3x3) fuses it
    Dst2 d; d.p0 = gn_tmp; d.p1 = nullptr; d.C0 = Cin; d.ld0 = Cin; d.ld1 = 0; d.acc0 = 0; d.acc1 = 0;
    launch_gn_bwd_apply(gn->x, gn->stats, gn->gamma, gn->beta, gn->da, B, H, W, Cin, gn->G, 0, gn->silu, nullptr, 0, 0.f, gn->red, d, N->st);
    p.A0 = gn_tmp; gn = nullptr;
  }
  if (gn != nullptr && gn->da == nullptr && !((w4 || w6) && fuse_gn)) {
    launch_gn_apply(gn->x, gn->stats, gn->gamma, gn->beta, B, H, W, Cin, gn->G, 0, gn->silu, gn_tmp, nullptr, N->st);
    p.A0 = gn_tmp; gn = nullptr;
  }
  const bool x3 = N->opt.gemm >= 1 && wgemm_supported(Cout, Cin);     // the batched GEMM pass in split arithmetic: only the stage image is needed
  if (w6) {
    const int xf = !x3 ? 0 : (N->opt.gemm == 2 && !wgemm_f16x2_supported(Cout, Cin)) ? 1 : N->opt.gemm;     // 1 bf16x3, 2 f16x2
    const WVar* wv = conv_weights(N, *c.w, c.dgrad, 6, xf);
    if (!wv) return -1;
    unsigned* vm = xf == 2 ? vmax_slot(N, B) : nullptr;
    if (xf == 2 && !vm) return -1;
    const float* U6 = wv->u; const void* U6x = xf == 2 ? wv->x2 : xf ? wv->x : nullptr;
    long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf);
    const bool fuse_bwd = N->opt.gn_fuse_bwd != 0;
    const bool want_bwd = bwd_gn != nullptr && fuse_gn && fuse_bwd;
    const int sc = ((stat_out != nullptr && fuse_gn) || want_bwd) ? wino6_stat_chunks(p) : 0;
    const bool stat = sc > 0 && (long long)sc * Cout <= 256LL * 1024;
    const double xr = wino6_exec_ratio(p);
    igemm_prof_record(p, 9, 1, N->st, true, xr);
    launch_wino6(p, U6, N->w4_scratch, N->w4_scratch + vf, N->st, gn, stat ? N->partial : nullptr, (stat && want_bwd) ? bwd_gn : nullptr, U6x, 0, xf, vm);
    igemm_prof_record(p, 9, 1, N->st, false, xr);
    if (stat && (want_bwd || direct)) return sc;
    if (stat && stat_out) { launch_csum_collapse(N->partial, sc, B, Cout, stat_out->csum, N->st); stat_out->has_csum = true; }
  } else if (w4) {
    const WVar* wv = conv_weights(N, *c.w, c.dgrad, 4, x3 ? 1 : 0);       // F(4x4,3x3): the small layers stay on bf16x3 in both split modes
    if (!wv) return -1;
    const float* U4 = wv->u; const void* U4x = x3 ? wv->x : nullptr;
    long long vf = 0, mf = 0; wino4_scratch(p, &vf, &mf);
    const int sc = (stat_out != nullptr && fuse_gn) ? wino4_stat_chunks(p) : 0;
    const bool stat = sc > 0 && (long long)sc * Cout <= 256LL * 1024;     // N->partial holds 256 x 1024 (chunk, channel) pairs per utterance
    igemm_prof_record(p, 9, 1, N->st, true, 0.25);
    launch_wino4(p, U4, N->w4_scratch, N->w4_scratch + vf, N->st, gn, stat ? N->partial : nullptr, U4x);
    igemm_prof_record(p, 9, 1, N->st, false, 0.25);
    if (stat) { launch_csum_collapse(N->partial, sc, B, Cout, stat_out->csum, N->st); stat_out->has_csum = true; }
  } else if (use_wino && wino_ok && wino_supported(p)) {
    const WVar* wv = conv_weights(N, *c.w, c.dgrad, 2, false);
    if (!wv) return -1;
    igemm_prof_record(p, 9, 1, N->st, true);
    launch_wino(p, wv->u, N->st);
    igemm_prof_record(p, 9, 1, N->st, false);
  } else {
    const WVar* wv = conv_weights(N, *c.w, c.dgrad, 0, false);
    if (!wv) return -1;
    p.Bt = wv->u;
    launch_igemm(p, 9, false, false, 1, N->st);
  }
  return 0;
}
// The general GEMMs in f16x2 arithmetic (round 6; only with gemm = f16x2): option gen_f16x2 = 1 every shape (DEFAULT), 0 never (the exact bf16x3 split),
// 2 the smaller launches only.  The first version took each row's power of two in a pre-pass that re-read A from beyond L2 and measured equal to bf16x3
// (331.8 vs 328.4 us, 129.0 vs 128.8 us per launch); with the scale found on the way (wgemm_f16x2_gen_kernel) every 1x1 / NIN / skip-path shape of the
// network is faster, 6.83 -> 5.98 ms over one forward + VJP at B = 8 (tools/gemm_shapes.py, profiles/README.md round 6) and 1.2 - 1.5x per launch in
// isolation (tools/gen_gemm_one.py).
static bool gen_f16x2_on(const Net* N, long long M) {
  if (N->opt.gemm != 2 || N->opt.gen_f16x2 == 0) return false;
  return N->opt.gen_f16x2 == 1 || M <= 600000;
}
// a plain row-major GEMM against a registered [N][K] weight (1x1 convolution, NIN) in bf16x3 arithmetic when the handle's mode asks for it
static bool try_wgemm(Net* N, const IgemmParams& p) {
  if (N->opt.gemm < 1 || p.bias_m || p.bias_bn || p.res_mode || p.out_scale != 1.f || p.sA || p.sC) return false;
  const auto it = N->W->w3.find(p.Bt);
  if (it == N->W->w3.end() || p.ldB != p.Cin || it->second.N != p.N || it->second.K != p.Cin) return false;
  if (!wgemm_general_supported(p.N, p.Cin, p.A1 ? p.C0 : 0, p.ldA0, p.A1 ? p.ldA1 : 0, p.ldC, p.A0, p.A1, p.C, p.bias_n)) return false;
  igemm_prof_record(p, 1, 1, N->st, true, 1.0);
  if (gen_f16x2_on(N, p.M) && it->second.img2 && wgemm_f16x2_supported(p.N, p.Cin))
    launch_wgemm_f16x2_general(p.A0, p.ldA0, p.A1, p.ldA1, p.C0, it->second.img2, p.C, p.ldC, p.M, p.N, p.Cin, p.bias_n, p.alpha, p.accumulate, N->st);
  else
    launch_wgemm_bf16x3_general(p.A0, p.ldA0, p.A1, p.ldA1, p.C0, it->second.img, p.C, p.ldC, p.M, p.N, p.Cin, p.bias_n, p.alpha, p.accumulate, N->st);
  igemm_prof_record(p, 1, 1, N->st, false, 1.0);
  return true;
}
// 1x1 conv / per-pixel linear over a (possibly two-source) view
static void conv1(Net* N, Src2 a, long long M, int Cin, const float* wt, int Cout, const float* bias, float alpha, float* out, int accumulate) {
  if (N->dry()) return;
  IgemmParams p = ig_base();
  p.A0 = a.p0; p.A1 = a.p1; p.C0 = a.C0; p.ldA0 = a.ld0; p.ldA1 = a.ld1; p.Cin = Cin; p.M = (int)M; p.N = Cout;
  p.Bt = wt; p.ldB = Cin; p.C = out; p.ldC = Cout; p.bias_n = bias; p.alpha = alpha; p.accumulate = accumulate;
  if (try_wgemm(N, p)) return;
  launch_igemm(p, 1, false, false, 1, N->st);
}
static Src2 single(const float* p, int C) { Src2 s; s.p0 = p; s.p1 = nullptr; s.C0 = C; s.ld0 = C; s.ld1 = 0; return s; }

// GroupNorm statistics of a (concatenated) view from the per-channel sums its tensors carry; a tensor without them gets one reduction pass
static void view_stats(Net* N, const View& x, int HW, int G, float* stats) {
  hipStream_t st = N->st;
  const int B = x.a->B;
  for (Tens* t : {x.a, x.b})
    if (t != nullptr && !t->has_csum) { launch_chan_sums(t->p, B, HW, t->C, N->partial, t->csum, st); t->has_csum = true; }
  launch_gn_stats_csum(x.a->csum, x.b ? x.b->csum : nullptr, x.a->C, B, HW, x.C(), G, 1e-6f, stats, st);
}

// ------------------------------------------------------------------------------------------------ composite ops
static Tens* resblock(Net* N, const ResW& R, View x, int mode, const float* temb_all, bool rec) {
  const int B = x.a->B, H = x.a->H, W = x.a->W, Cin = R.cin, Cout = R.cout;
  const int Ho = mode == 1 ? H / 2 : (mode == 2 ? H * 2 : H), Wo = mode == 1 ? W / 2 : (mode == 2 ? W * 2 : W);
  const int G0 = gn_groups(Cin), G1 = gn_groups(Cout);
  hipStream_t st = N->st;
  Tens* out = N->mk(B, Ho, Wo, Cout, rec);
  Tens* h1 = N->mk(B, Ho, Wo, Cout, false);
  float* stats0 = N->tmp((long long)B * G0 * 2);
  float* stats1 = N->tmp((long long)B * G1 * 2);
  const size_t mark = N->arena.off;
  const bool firm = N->fir && mode != 0;                  // FIR resampling: explicit passes on h = act(GN(x)) and on x (layerspp.py:246-255)
  float* a0 = N->tmp((long long)B * Ho * Wo * Cin);
  float* xr = (mode == 1 || firm) ? N->tmp((long long)B * Ho * Wo * Cin) : nullptr;
  float* xs = R.has_c2 ? N->tmp((long long)B * ((mode == 2 && !firm) ? H * W : Ho * Wo) * Cout) : nullptr;
  float* a1 = N->tmp((long long)B * Ho * Wo * Cout);
  float* a0f = firm ? N->tmp((long long)B * H * W * Cin) : nullptr;
  // up block without FIR: Conv_0 in its sub-pixel form on the (H, W) grid (GroupNorm_0 + SiLU inside its input transform, no upsampled activation)
  const bool up6 = mode == 2 && !firm && conv3_up_ok(N, B, H, W, Cin, Cout);
  if (N->dry()) {                                         // sizing pass: let the convolutions note their F(4x4,3x3) scratch need
    Conv3 c; c.B = B; c.H = Ho; c.W = Wo; c.Cin = Cin; c.Cout = Cout; c.w = &R.c0;
    if (up6) { c.H = H; c.W = W; c.up = 1; }
    conv3(N, c);
    c.H = Ho; c.W = Wo; c.up = 0; c.Cin = Cout; c.w = &R.c1; conv3(N, c);
  }
  if (!N->dry()) {
    view_stats(N, x, H * W, G0, stats0);
    W4Gn g0{src_of(x), stats0, R.gn0.gamma, R.gn0.beta, G0, 1}, g1{single(h1->p, Cout), stats1, R.gn1.gamma, R.gn1.beta, G1, 1};
    if (firm) {
      launch_gn_apply(src_of(x), stats0, R.gn0.gamma, R.gn0.beta, B, H, W, Cin, G0, 0, 1, a0f, nullptr, st);
      if (mode == 2) { launch_fir_up2(a0f, a0, B, H, W, Cin, 1.f, 0, st); launch_fir_up2(x.a->p, xr, B, H, W, Cin, 1.f, 0, st); }
      else { launch_fir_down2(a0f, a0, B, H, W, Cin, 1.f, 0, st); launch_fir_down2(x.a->p, xr, B, H, W, Cin, 1.f, 0, st); }
    } else if (mode != 0 && !up6)
    launch_gn_apply(src_of(x), stats0, R.gn0.gamma, R.gn0.beta, B, H, W, Cin, G0, mode, 1, a0, xr, st);
    // same resolution: act(GroupNorm(.)) is applied by the convolution's input transform (a0 / a1 are only its fallback buffers)
    Conv3 c0; c0.a = a0; c0.B = B; c0.H = Ho; c0.W = Wo; c0.Cin = Cin; c0.Cout = Cout; c0.w = &R.c0; c0.bias = R.c0.bias;
    c0.bias_bn = temb_all + R.dense_off; c0.ld_bn = N->W->dense_total; c0.out = h1->p;
    c0.gn = (mode == 0 || up6) ? &g0 : nullptr; c0.gn_tmp = a0; c0.stat_out = h1; c0.direct = true;
    if (up6) { c0.H = H; c0.W = W; c0.up = 1; }
    const int ch1 = conv3(N, c0);
    if (ch1 > 0) launch_gn_stats_partial(N->partial, ch1, B, Ho * Wo, Cout, G1, 1e-6f, stats1, st);   // h1 has this one reader
    else { View vh1; vh1.a = h1; view_stats(N, vh1, Ho * Wo, G1, stats1); }
    const float* res; int res_mode = 1;
    if (R.has_c2) {
      if (mode == 1 || firm) conv1(N, single(xr, Cin), (long long)B * Ho * Wo, Cin, R.c2.wf, Cout, R.c2.bias, 1.f, xs, 0);
      else conv1(N, src_of(x), (long long)B * H * W, Cin, R.c2.wf, Cout, R.c2.bias, 1.f, xs, 0);
      res = xs; if (mode == 2 && !firm) res_mode = 2;
    } else {
      res = x.a->p;   // identity skip: single source, same resolution, Cin == Cout
    }
    Conv3 c1; c1.a = a1; c1.B = B; c1.H = Ho; c1.W = Wo; c1.Cin = Cout; c1.Cout = Cout; c1.w = &R.c1; c1.bias = R.c1.bias;
    c1.res = res; c1.ldRes = Cout; c1.res_mode = res_mode; c1.out_scale = INV_SQRT2; c1.out = out->p;
    c1.gn = &g1; c1.gn_tmp = a1; c1.stat_out = out;
    conv3(N, c1);
  }
  N->arena.off = mark;
  if (rec) {
    const ResW* Rp = &R;
    N->tape.push_back([=]() {
      Net* n = N; hipStream_t s = n->st;
      const size_t mk = n->arena.off;
      const float* dout = out->g;
      const float* extra; int extra_mode = 1; float extra_scale = 1.f;
      Dst2 d0 = gdst_of(x);
      // skip path through Conv_2 at the block's input resolution: its 1x1 data-gradient GEMM takes the GroupNorm_0 backward's apply pass as its epilogue
      // (wgemm.hip GNB): the block's input gradient in one launch, neither the 1x1 result nor a separate apply pass in HBM
      const W3Img* c2i = nullptr;
      const float* c2a = dout;                               // A operand of that GEMM (the pooled gradient for the sub-pixel up block)
      const bool fuse_c2_on = n->opt.c2_fuse != 0;     // A/B switch
      if (fuse_c2_on && Rp->has_c2 && !firm && (mode == 0 || (mode == 2 && up6)) && n->opt.gemm >= 1 && Cin % 4 == 0 && (Cin / G0) % 4 == 0) {
        const auto it = n->W->w3.find(Rp->c2.wb);
        if (it != n->W->w3.end() && it->second.N == Cin && it->second.K == Cout && wgemm_gnbwd_supported(Cin, Cout, Cout, src_of(x), d0, dout, dout))
          c2i = &it->second;
      }
      if (c2i) {
        if (mode == 2) {
          float* pooled = n->tmp((long long)B * H * W * Cout);
          if (!n->dry()) launch_pool2(dout, pooled, B, Ho, Wo, Cout, 1.f, 0, s);
          c2a = pooled;
        }
        extra = nullptr;
      } else
      if (Rp->has_c2) {
        float* tx;
        if (firm) {                                        // d x_resampled at (Ho, Wo), then the FIR adjoint back to (H, W)
          float* txr = n->tmp((long long)B * Ho * Wo * Cin);
          tx = n->tmp((long long)B * H * W * Cin);
          conv1(n, single(dout, Cout), (long long)B * Ho * Wo, Cout, Rp->c2.wb, Cin, nullptr, INV_SQRT2, txr, 0);
          if (!n->dry()) {
            if (mode == 2) launch_fir_down2(txr, tx, B, Ho, Wo, Cin, 4.f, 0, s);      // up^T = 4 down
            else launch_fir_up2(txr, tx, B, Ho, Wo, Cin, 0.25f, 0, s);                // down^T = up / 4
          }
        } else if (mode == 2) {
          float* pooled = n->tmp((long long)B * H * W * Cout);
          tx = n->tmp((long long)B * H * W * Cin);
          if (!n->dry()) {
            launch_pool2(dout, pooled, B, Ho, Wo, Cout, 1.f, 0, s);
            conv1(n, single(pooled, Cout), (long long)B * H * W, Cout, Rp->c2.wb, Cin, nullptr, INV_SQRT2, tx, 0);
          }
        } else {
          tx = n->tmp((long long)B * Ho * Wo * Cin);
          conv1(n, single(dout, Cout), (long long)B * Ho * Wo, Cout, Rp->c2.wb, Cin, nullptr, INV_SQRT2, tx, 0);
          if (mode == 1) extra_mode = 2;
        }
        extra = tx;
      } else {
        extra = dout; extra_scale = INV_SQRT2;
      }
      float* da1 = n->tmp((long long)B * Ho * Wo * Cout);
      float* dh1 = n->tmp((long long)B * Ho * Wo * Cout);
      float* da0 = n->tmp((long long)B * Ho * Wo * Cin);
      // on the F(6x6,3x3) path the data-gradient convolutions leave the backward sums of the GroupNorm their output feeds (same resolution)
      const W4Gn b1{single(h1->p, Cout), stats1, Rp->gn1.gamma, Rp->gn1.beta, G1, 1}, b0{src_of(x), stats0, Rp->gn0.gamma, Rp->gn0.beta, G0, 1};
      Conv3 d1c; d1c.a = dout; d1c.B = B; d1c.H = Ho; d1c.W = Wo; d1c.Cin = Cout; d1c.Cout = Cout; d1c.w = &Rp->c1; d1c.dgrad = true;
      d1c.alpha = INV_SQRT2; d1c.out = da1; d1c.bwd_gn = &b1;
      const int s1 = conv3(n, d1c);
      // GroupNorm_1 backward: its two per-group means here, its apply pass inside the input transform of the Conv_0 data-gradient (dh1 is
      // only that convolution's fallback buffer)
      if (!n->dry())
        launch_gn_bwd_sums(single(h1->p, Cout), stats1, Rp->gn1.gamma, Rp->gn1.beta, da1, B, Ho, Wo, Cout, G1, 0, 1, n->partial, n->red, s, s1);
      W4Gn gb1{single(h1->p, Cout), stats1, Rp->gn1.gamma, Rp->gn1.beta, G1, 1};
      gb1.da = da1; gb1.ldda = Cout; gb1.red = n->red;
      Conv3 d0c; d0c.a = dh1; d0c.B = B; d0c.H = Ho; d0c.W = Wo; d0c.Cin = Cout; d0c.Cout = Cin; d0c.w = &Rp->c0; d0c.dgrad = true;
      d0c.out = da0; d0c.gn = &gb1; d0c.gn_tmp = dh1; d0c.bwd_gn = (mode == 0 || up6) ? &b0 : nullptr;
      if (up6) { d0c.H = H; d0c.W = W; d0c.up = 2; }      // da0 comes out at (H, W): the nearest-upsample's adjoint is inside the convolution
      const int s0 = conv3(n, d0c);
      if (c2i) {
        if (!n->dry()) {
          launch_gn_bwd_sums(src_of(x), stats0, Rp->gn0.gamma, Rp->gn0.beta, da0, B, H, W, Cin, G0, 0, 1, n->partial, n->red, s, s0);
          IgemmParams pr = ig_base(); pr.M = B * H * W; pr.N = Cin; pr.Cin = Cout;
          igemm_prof_record(pr, 1, 1, s, true, 1.0);
          if (gen_f16x2_on(n, (long long)B * H * W) && c2i->img2 && wgemm_f16x2_supported(Cin, Cout))
            launch_wgemm_f16x2_gnbwd(c2a, Cout, c2i->img2, (long long)B * H * W, Cin, Cout, INV_SQRT2, src_of(x), da0, stats0, n->red, Rp->gn0.gamma,
                                     Rp->gn0.beta, G0, 1, H * W, d0, s);
          else
          launch_wgemm_bf16x3_gnbwd(c2a, Cout, c2i->img, (long long)B * H * W, Cin, Cout, INV_SQRT2, src_of(x), da0, stats0, n->red, Rp->gn0.gamma,
                                    Rp->gn0.beta, G0, 1, H * W, d0, s);
          igemm_prof_record(pr, 1, 1, s, false, 1.0);
        }
      } else
      if (firm) {
        float* da0f = n->tmp((long long)B * H * W * Cin);
        if (!n->dry()) {
          if (mode == 2) launch_fir_down2(da0, da0f, B, Ho, Wo, Cin, 4.f, 0, s);
          else launch_fir_up2(da0, da0f, B, Ho, Wo, Cin, 0.25f, 0, s);
          launch_gn_bwd(src_of(x), stats0, Rp->gn0.gamma, Rp->gn0.beta, da0f, B, H, W, Cin, G0, 0, 1, extra, 1, 1.f, n->partial, n->red, d0, s);
        }
      } else
      if (!n->dry())
        launch_gn_bwd(src_of(x), stats0, Rp->gn0.gamma, Rp->gn0.beta, da0, B, H, W, Cin, G0, up6 ? 0 : mode, 1, extra, extra_mode, extra_scale,
                      n->partial, n->red, d0, s, s0);
      n->arena.off = mk;
    });
  }
  return out;
}

static void gemm_b(Net* N, const float* A, int ldA, long long sA, bool tA, const float* Bt, int ldB, long long sB, bool tB, float* C, int ldC,
                   long long sC, int M, int Nn, int K, const float* bias_n, const float* bias_m, float alpha, int accumulate, int batch) {
  if (N->dry()) return;
  IgemmParams p = ig_base();
  p.A0 = A; p.ldA0 = ldA; p.sA = sA; p.Bt = Bt; p.ldB = ldB; p.sB = sB; p.C = C; p.ldC = ldC; p.sC = sC;
  p.M = M; p.N = Nn; p.Cin = K; p.bias_n = bias_n; p.bias_m = bias_m; p.alpha = alpha; p.accumulate = accumulate;
  if (!tA && !tB && batch == 1 && try_wgemm(N, p)) return;
  launch_igemm(p, 1, tA, tB, batch, N->st);
}

// Attention mode of a handle: 4 = auto (default): fp32 throughout, the materialised T x T form (the reference's own formulation, layerspp.py:82-86) while
// the matrix is small -- T <= ATTN_MATRIX_MAX_T: 16.8 MB / utterance at 4 s -- and the online-softmax (flash) kernels beyond (905 MB / utterance at 30 s
// never exists).  Measured at B = 8, T = 2048 (tools/ab_env.sh BUDDY_ATTN flash matrix): 65.4 -> 64.6 ms/step: six plain batched GEMMs at 100+ TFLOP/s
// beat kernels that run one wave per SIMD.  The choice depends on T alone: a row's arithmetic does not depend on the batch it is in.
// 0 = flash, fp32 operands; 1 / 2 = flash with bf16 / f16 MFMA operands (opt-in fast mode); 3 = always the materialised form.
// Default from BUDDY_ATTN (auto | matrix | flash | bf16 | f16; options.hip), changed per handle with buddy_ncsnpp_set_attention / _set_option.
constexpr int ATTN_MATRIX_MAX_T = 4096;
static bool attn_use_flash(const Net* N, int C, int T) {
  if (!flash_attn_supported(C) || N->opt.attn == 3) return false;
  return N->opt.attn != 4 || T > ATTN_MATRIX_MAX_T;
}
static int attn_prec(const Net* N) { return N->opt.attn == 1 || N->opt.attn == 2 ? N->opt.attn : 0; }

static Tens* attnblock_flash(Net* N, const AttnW& A, Tens* x, bool rec) {
  const int B = x->B, H = x->H, W = x->W, C = A.C, T = H * W, G = gn_groups(C);
  hipStream_t st = N->st;
  const float scale = 1.f / std::sqrt((float)C);
  const long long BTC = (long long)B * T * C;
  Tens* out = N->mk(B, H, W, C, rec);
  float* stats = N->tmp((long long)B * G * 2);
  float* q = N->tmp(BTC); float* k = N->tmp(BTC); float* v = N->tmp(BTC); float* O = N->tmp(BTC);
  float* lse = N->tmp((long long)B * T);
  const size_t mark = N->arena.off;
  float* hn = N->tmp(BTC);
  // few utterances: the kernels' sequential loops are split over more workgroups (attn.hip: split_range); workspace from the arena
  // 16-bit modes: the operand arrays of the pre-pass (attn16.hip) live in the same scratch
  const int prec = attn_prec(N);
  const int splits = prec ? 1 : flash_attn_splits(B, T);
  float* aws = prec ? N->tmp(flash_attn16_ws_floats(B, T, C)) : splits > 1 ? N->tmp(flash_attn_ws_floats(B, T, C, splits)) : nullptr;
  if (!N->dry()) {
    { View vx; vx.a = x; view_stats(N, vx, T, G, stats); }
    launch_gn_apply(single(x->p, C), stats, A.gn.gamma, A.gn.beta, B, H, W, C, G, 0, 0, hn, nullptr, st);
    gemm_b(N, hn, C, 0, false, A.Wt[0], C, 0, false, q, C, 0, B * T, C, C, A.b[0], nullptr, 1.f, 0, 1);
    gemm_b(N, hn, C, 0, false, A.Wt[1], C, 0, false, k, C, 0, B * T, C, C, A.b[1], nullptr, 1.f, 0, 1);
    gemm_b(N, hn, C, 0, false, A.Wt[2], C, 0, false, v, C, 0, B * T, C, C, A.b[2], nullptr, 1.f, 0, 1);
    if (prec) launch_flash_attn16_fwd(q, k, v, O, lse, B, T, C, scale, prec, aws, st);
    else launch_flash_attn_fwd(q, k, v, O, lse, B, T, C, scale, aws, splits, st);
    IgemmParams p = ig_base();
    p.A0 = O; p.ldA0 = C; p.Cin = C; p.M = B * T; p.N = C; p.Bt = A.Wt[3]; p.ldB = C; p.C = out->p; p.ldC = C; p.bias_n = A.b[3];
    p.res = x->p; p.ldRes = C; p.res_mode = 1; p.out_scale = INV_SQRT2;
    launch_igemm(p, 1, false, false, 1, st);
  }
  N->arena.off = mark;
  if (rec) {
    const AttnW* Ap = &A;
    N->tape.push_back([=]() {
      Net* n = N; hipStream_t s = n->st;
      const size_t mk = n->arena.off;
      const float* dout = out->g;
      float* dO = n->tmp(BTC); float* dq = n->tmp(BTC); float* dk = n->tmp(BTC); float* dv = n->tmp(BTC); float* dhn = n->tmp(BTC);
      float* dl = n->tmp((long long)B * T);
      const int bprec = attn_prec(n);
      const int bsplits = bprec ? 1 : flash_attn_splits(B, T);
      float* bws = bprec ? n->tmp(flash_attn16_ws_floats(B, T, C)) : bsplits > 1 ? n->tmp(flash_attn_ws_floats(B, T, C, bsplits)) : nullptr;
      gemm_b(n, dout, C, 0, false, Ap->Wn[3], C, 0, false, dO, C, 0, B * T, C, C, nullptr, nullptr, INV_SQRT2, 0, 1);
      if (!n->dry()) {
        if (bprec) launch_flash_attn16_bwd(q, k, v, O, dO, lse, dl, dq, dk, dv, B, T, C, scale, bprec, bws, s);
        else launch_flash_attn_bwd(q, k, v, O, dO, lse, dl, dq, dk, dv, B, T, C, scale, bws, bsplits, s);
      }
      gemm_b(n, dq, C, 0, false, Ap->Wn[0], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 0, 1);
      gemm_b(n, dk, C, 0, false, Ap->Wn[1], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 1, 1);
      gemm_b(n, dv, C, 0, false, Ap->Wn[2], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 1, 1);
      View xv; xv.a = x;
      Dst2 d = gdst_of(xv);
      if (!n->dry())
        launch_gn_bwd(single(x->p, C), stats, Ap->gn.gamma, Ap->gn.beta, dhn, B, H, W, C, G, 0, 0, dout, 1, INV_SQRT2, n->partial, n->red, d, s);
      n->arena.off = mk;
    });
  }
  return out;
}

void launch_transpose_sq(const float* src, float* dst, int batch, int n, hipStream_t st);   // ops.hip: dst[b] = src[b]^T, n x n, n % 32 == 0
static Tens* attnblock(Net* N, const AttnW& A, Tens* x, bool rec) {
  if (attn_use_flash(N, A.C, x->H * x->W)) return attnblock_flash(N, A, x, rec);
  const int B = x->B, H = x->H, W = x->W, C = A.C, T = H * W, G = gn_groups(C);
  hipStream_t st = N->st;
  const float scale = 1.f / std::sqrt((float)C);
  Tens* out = N->mk(B, H, W, C, rec);
  float* stats = N->tmp((long long)B * G * 2);
  float* q = N->tmp((long long)B * T * C);
  float* k = N->tmp((long long)B * T * C);
  float* vT = N->tmp((long long)B * T * C);
  float* P = N->tmp((long long)B * T * T);
  const size_t mark = N->arena.off;
  float* hn = N->tmp((long long)B * T * C);
  float* O = N->tmp((long long)B * T * C);
  if (!N->dry()) {
    { View vx; vx.a = x; view_stats(N, vx, T, G, stats); }
    launch_gn_apply(single(x->p, C), stats, A.gn.gamma, A.gn.beta, B, H, W, C, G, 0, 0, hn, nullptr, st);
    gemm_b(N, hn, C, 0, false, A.Wt[0], C, 0, false, q, C, 0, B * T, C, C, A.b[0], nullptr, 1.f, 0, 1);
    gemm_b(N, hn, C, 0, false, A.Wt[1], C, 0, false, k, C, 0, B * T, C, C, A.b[1], nullptr, 1.f, 0, 1);
    gemm_b(N, A.Wt[2], C, 0, false, hn, C, (long long)T * C, false, vT, T, (long long)C * T, C, T, C, nullptr, A.b[2], 1.f, 0, B);
    gemm_b(N, q, C, (long long)T * C, false, k, C, (long long)T * C, false, P, T, (long long)T * T, T, T, C, nullptr, nullptr, scale, 0, B);
    launch_softmax_rows(P, B * T, T, st);
    gemm_b(N, P, T, (long long)T * T, false, vT, T, (long long)C * T, false, O, C, (long long)T * C, T, C, T, nullptr, nullptr, 1.f, 0, B);
    IgemmParams p = ig_base();
    p.A0 = O; p.ldA0 = C; p.Cin = C; p.M = B * T; p.N = C; p.Bt = A.Wt[3]; p.ldB = C; p.C = out->p; p.ldC = C; p.bias_n = A.b[3];
    p.res = x->p; p.ldRes = C; p.res_mode = 1; p.out_scale = INV_SQRT2;
    launch_igemm(p, 1, false, false, 1, st);
  }
  N->arena.off = mark;
  if (rec) {
    const AttnW* Ap = &A;
    N->tape.push_back([=]() {
      Net* n = N; hipStream_t s = n->st;
      const size_t mk = n->arena.off;
      const float* dout = out->g;
      const long long TC = (long long)T * C, TT = (long long)T * T;
      float* dO = n->tmp(B * TC); float* dP = n->tmp(B * TT); float* dvT = n->tmp(B * TC);
      float* dq = n->tmp(B * TC); float* dk = n->tmp(B * TC); float* dhn = n->tmp(B * TC);
      // P^T and dS^T by a tiled transpose (T % 32 == 0): the two products that need them then take the row-major-A kernel (and its half-height tiles when
      // the grid is small) instead of the doubly transposed one (272 us per launch at B = 8 against 41 + ~150)
      const bool use_tr = n->opt.attn_tr != 0;     // A/B switch
      const bool tr = use_tr && T % 32 == 0;
      float* Tr = tr ? n->tmp(B * TT) : nullptr;
      gemm_b(n, dout, C, 0, false, Ap->Wn[3], C, 0, false, dO, C, 0, B * T, C, C, nullptr, nullptr, INV_SQRT2, 0, 1);
      gemm_b(n, dO, C, TC, false, vT, T, TC, true, dP, T, TT, T, T, C, nullptr, nullptr, 1.f, 0, B);        // dP = dO V^T
      if (tr) {
        if (!n->dry()) launch_transpose_sq(P, Tr, B, T, s);
        gemm_b(n, Tr, T, TT, false, dO, C, TC, true, dvT, C, TC, T, C, T, nullptr, nullptr, 1.f, 0, B);     // dV = P^T dO  (dvT holds dV [T][C] here)
      } else
      gemm_b(n, dO, C, TC, true, P, T, TT, true, dvT, T, TC, C, T, T, nullptr, nullptr, 1.f, 0, B);         // dV^T = dO^T P
      if (!n->dry()) launch_softmax_bwd_rows(P, dP, B * T, T, s);                                           // dP <- dS
      gemm_b(n, dP, T, TT, false, k, C, TC, true, dq, C, TC, T, C, T, nullptr, nullptr, scale, 0, B);        // dq = scale dS k
      if (tr) {
        if (!n->dry()) launch_transpose_sq(dP, Tr, B, T, s);
        gemm_b(n, Tr, T, TT, false, q, C, TC, true, dk, C, TC, T, C, T, nullptr, nullptr, scale, 0, B);      // dk = scale dS^T q
      } else
      gemm_b(n, dP, T, TT, true, q, C, TC, true, dk, C, TC, T, C, T, nullptr, nullptr, scale, 0, B);         // dk = scale dS^T q
      gemm_b(n, dq, C, 0, false, Ap->Wn[0], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 0, 1);
      gemm_b(n, dk, C, 0, false, Ap->Wn[1], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 1, 1);
      if (tr) gemm_b(n, dvT, C, 0, false, Ap->Wn[2], C, 0, false, dhn, C, 0, B * T, C, C, nullptr, nullptr, 1.f, 1, 1);
      else
      gemm_b(n, dvT, T, TC, true, Ap->Wn[2], C, 0, false, dhn, C, TC, T, C, C, nullptr, nullptr, 1.f, 1, B);
      View xv; xv.a = x;
      Dst2 d = gdst_of(xv);
      if (!n->dry())
        launch_gn_bwd(single(x->p, C), stats, Ap->gn.gamma, Ap->gn.beta, dhn, B, H, W, C, G, 0, 0, dout, 1, INV_SQRT2, n->partial, n->red, d, s);
      n->arena.off = mk;
    });
  }
  return out;
}

// ------------------------------------------------------------------------------------------------ network forward
static int ensure_env(Net* N, int Tp) {
  if (N->env_Tp == Tp) return BUDDY_OK;
  const int nfft = N->cfg.n_fft, hop = N->cfg.hop;
  const int len = nfft + hop * (Tp - 1);
  std::vector<double> env(len, 0.0);
  const double PI = 3.14159265358979323846;
  for (int t = 0; t < Tp; ++t)
    for (int k = 0; k < nfft; ++k) { const double w = 0.5 - 0.5 * std::cos(2.0 * PI * k / nfft); env[t * hop + k] += w * w; }
  std::vector<float> inv(len);
  for (int i = 0; i < len; ++i) inv[i] = env[i] > 1e-11 ? (float)(1.0 / (double)(float)env[i]) : 0.f;
  if (N->inv_env) (void)hipFree(N->inv_env);
  HIPCHK(hipMalloc(&N->inv_env, (size_t)len * 4));
  HIPCHK(hipMemcpy(N->inv_env, inv.data(), (size_t)len * 4, hipMemcpyHostToDevice));
  N->env_len = len; N->env_Tp = Tp;
  return BUDDY_OK;
}

static void run_forward(Net* N, const float* x, const float* cnoise, const float* cin_b, const float* cskip_b, const float* cout_b, float* y,
                        int B, int L, bool rec) {
  const NetCfg& c = N->cfg;
  hipStream_t st = N->st;
  const int nf = c.nf, Fb = N->Fb, Kp = N->Kp, hop = c.hop, nfft = c.n_fft;
  const int T = 1 + L / hop, Tp = (T + 15) / 16 * 16;
  const int Lp = (L + nfft + 8 + 3) / 4 * 4;
  N->B = B; N->L = L; N->T = T; N->Tp = Tp; N->Lp = Lp;
  N->pool.clear(); N->tape.clear(); N->taps.clear();
  N->arena.off = 0;
  // the EDM scalars are needed again by the VJP: keep private copies (the caller's buffers may be gone by then)
  N->k_cin = N->k_cskip = N->k_cout = nullptr;
  if (cin_b) {
    float* sc = N->tmp(3 * (long long)B);
    if (!N->dry()) {
      (void)hipMemcpyAsync(sc, cin_b, (size_t)B * 4, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(sc + B, cskip_b, (size_t)B * 4, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync(sc + 2 * B, cout_b, (size_t)B * 4, hipMemcpyDeviceToDevice, st);
    }
    N->k_cin = sc; N->k_cskip = sc + B; N->k_cout = sc + 2 * B;
    cin_b = N->k_cin; cskip_b = N->k_cskip; cout_b = N->k_cout;
  }
  // reduction scratch: chunks <= 256, C <= 1024
  N->partial = (double*)N->arena.alloc((size_t)B * 256 * 1024 * 2 * sizeof(double));
  N->red = N->tmp((long long)B * 32 * 2);

  // STFT as GEMM over the reflect-padded signal
  float* xp = N->tmp((long long)B * Lp);
  Tens* spec = N->mk(B, Tp, Fb, 2, rec);
  N->spec = spec;
  if (!N->dry()) {
    launch_reflect_pad(x, xp, B, L, N->pad, Lp, 1.f, cin_b, st);
    (void)hipMemsetAsync(spec->p, 0, (size_t)spec->numel() * 4, st);
    gemm_b(N, xp, hop, Lp, false, N->W->basisF, Kp, 0, false, spec->p, 2 * Fb, (long long)Tp * 2 * Fb, T, 2 * Fb, Kp, nullptr, nullptr, 1.f, 0, B);
  }
  // time embedding (reference ncsnpp.py:299-318) and all Dense_0 projections in one launch (layerspp.py:263)
  float* four = N->tmp((long long)B * 2 * nf);
  float* t1 = N->tmp((long long)B * 4 * nf);
  float* temb = N->tmp((long long)B * 4 * nf);
  float* temb_all = N->tmp((long long)B * N->W->dense_total);
  if (!N->dry()) {
    launch_fourier(cnoise, N->W->Wf, four, B, nf, st);
    launch_linear(four, N->W->lin1_w, N->W->lin1_b, t1, B, 2 * nf, 4 * nf, 0, st);
    launch_linear(t1, N->W->lin2_w, N->W->lin2_b, temb, B, 4 * nf, 4 * nf, 1, st);
    launch_linear(temb, N->W->dense_w, N->W->dense_b, temb_all, B, 4 * nf, N->W->dense_total, 1, st);
  }

  int mi = 3, ri = 0, ci = 0;
  auto tap = [&](int idx, Tens* t) { N->taps.push_back({idx, t}); };
  // input conv
  Tens* h0 = N->mk(B, Tp, Fb, nf, rec);
  if (!N->dry()) launch_conv_c2in(spec->p, N->W->conv_in.wf, N->W->conv_in.bias, nullptr, 0, h0->p, nf, B, Tp, Fb, nf, 9, 0, st);
  if (rec) N->tape.push_back([=]() {
    if (N->dry()) { spec->ginit = 1; return; }
    launch_conv_c2out(h0->g, nf, N->W->conv_in.wb, nullptr, nullptr, spec->g, B, Tp, Fb, nf, 9, spec->ginit, N->st);
    spec->ginit = 1;
  });
  tap(mi, h0); ++mi;
  std::vector<Tens*> hs{h0};
  Tens* pyr_in = spec;
  for (int l = 0; l < c.nlev; ++l) {
    for (int b = 0; b < c.nrb; ++b) {
      View v; v.a = hs.back();
      Tens* h = resblock(N, N->W->res[ri++], v, 0, temb_all, rec); tap(mi, h); ++mi;
      hs.push_back(h);
    }
    if (l != c.nlev - 1) {
      View v; v.a = hs.back();
      Tens* h = resblock(N, N->W->res[ri++], v, 1, temb_all, rec); tap(mi, h); ++mi;
      Tens* pin = N->mk(B, pyr_in->H / 2, pyr_in->W / 2, 2, rec);
      Tens* prev = pyr_in;
      if (!N->dry()) {
        if (N->fir) launch_fir_down2(prev->p, pin->p, B, prev->H, prev->W, 2, 1.f, 0, st);          // downsample_2d (layerspp.py:159)
        else launch_pool2(prev->p, pin->p, B, prev->H, prev->W, 2, 0.25f, 0, st);                   // F.avg_pool2d (layerspp.py:156)
      }
      if (rec) N->tape.push_back([=]() {
        if (!N->dry()) {
          if (N->fir) launch_fir_up2(pin->g, prev->g, B, pin->H, pin->W, 2, 0.25f, prev->ginit, N->st);
          else launch_up2_acc(pin->g, prev->g, B, pin->H, pin->W, 2, 0.25f, prev->ginit, N->st);
        }
        prev->ginit = 1;
      });
      pyr_in = pin;
      const ConvW& cw = N->W->combine[ci++];
      Tens* hc = N->mk(B, h->H, h->W, h->C, rec);
      if (!N->dry()) launch_conv_c2in(pin->p, cw.wf, cw.bias, h->p, h->C, hc->p, h->C, B, h->H, h->W, h->C, 1, 0, st);   // Combine 'sum'
      if (rec) { const ConvW* cwp = &cw; N->tape.push_back([=]() {
        if (!N->dry()) {
          launch_axpy(h->g, hc->g, 1.f, h->numel(), h->ginit, N->st);
          launch_conv_c2out(hc->g, h->C, cwp->wb, nullptr, nullptr, pin->g, B, h->H, h->W, h->C, 1, pin->ginit, N->st);
        }
        h->ginit = 1; pin->ginit = 1;
      }); }
      tap(mi, hc); ++mi;
      hs.push_back(hc);
    }
  }
  Tens* h = hs.back();
  { View v; v.a = h; h = resblock(N, N->W->res[ri++], v, 0, temb_all, rec); tap(mi, h); ++mi; }
  h = attnblock(N, N->W->attn, h, rec); tap(mi, h); ++mi;
  { View v; v.a = h; h = resblock(N, N->W->res[ri++], v, 0, temb_all, rec); tap(mi, h); ++mi; }
  Tens* pyr = nullptr;
  for (int l = c.nlev - 1, j = 0; l >= 0; --l, ++j) {
    for (int b = 0; b < c.nrb + 1; ++b) {
      View v; v.a = h; v.b = hs.back(); hs.pop_back();
      h = resblock(N, N->W->res[ri++], v, 0, temb_all, rec); tap(mi, h); ++mi;
    }
    {  // pyramid head: GroupNorm -> SiLU -> conv3x3 C->2, plus nearest-upsampled previous pyramid (ncsnpp.py:391-412)
      const GNW& gw = N->W->pyr_gn[j]; const ConvW& cw = N->W->pyr_conv[j];
      const int C = h->C, G = gn_groups(C), Hh = h->H, Ww = h->W;
      Tens* np = N->mk(B, Hh, Ww, 2, rec);
      float* stats = N->tmp((long long)B * G * 2);
      const size_t mark = N->arena.off;
      float* a = N->tmp(h->numel());
      Tens* prevp = pyr; Tens* hh = h;
      if (!N->dry()) {
        { View vx; vx.a = h; view_stats(N, vx, Hh * Ww, G, stats); }
        launch_gn_apply(single(h->p, C), stats, gw.gamma, gw.beta, B, Hh, Ww, C, G, 0, 1, a, nullptr, st);
        if (N->fir) {                                      // upsample_2d of the running pyramid (layerspp.py:122) instead of nearest
          launch_conv_c2out(a, C, cw.wf, cw.bias, nullptr, np->p, B, Hh, Ww, C, 9, 0, st);
          if (prevp) launch_fir_up2(prevp->p, np->p, B, Hh / 2, Ww / 2, 2, 1.f, 1, st);
        } else
        launch_conv_c2out(a, C, cw.wf, cw.bias, prevp ? prevp->p : nullptr, np->p, B, Hh, Ww, C, 9, 0, st);
      }
      N->arena.off = mark;
      if (rec) { const GNW* gp = &gw; const ConvW* cp = &cw; N->tape.push_back([=]() {
        const size_t mk = N->arena.off;
        float* da = N->tmp(hh->numel());
        if (!N->dry()) {
          if (prevp && N->fir) launch_fir_down2(np->g, prevp->g, B, Hh, Ww, 2, 4.f, prevp->ginit, N->st);
          else if (prevp) launch_pool2(np->g, prevp->g, B, Hh, Ww, 2, 1.f, prevp->ginit, N->st);
          launch_conv_c2in(np->g, cp->wb, nullptr, nullptr, 0, da, C, B, Hh, Ww, C, 9, 0, N->st);
        }
        if (prevp) prevp->ginit = 1;
        View hv; hv.a = hh;
        Dst2 d = gdst_of(hv);
        if (!N->dry()) launch_gn_bwd(single(hh->p, C), stats, gp->gamma, gp->beta, da, B, Hh, Ww, C, G, 0, 1, nullptr, 0, 0.f, N->partial, N->red, d, N->st);
        N->arena.off = mk;
      }); }
      pyr = np; mi += 2; tap(mi - 1, pyr);
    }
    if (l != 0) { View v; v.a = h; h = resblock(N, N->W->res[ri++], v, 2, temb_all, rec); tap(mi, h); ++mi; }
  }
  N->pyr0 = pyr;
  // output layer (1x1, 2->2), iSTFT as GEMM + overlap-add with the EDM skip/out scaling folded in
  float* o2 = N->tmp((long long)B * Tp * Fb * 2);
  float* frames = N->tmp((long long)B * Tp * Kp);
  if (!N->dry()) {
    launch_mix2(pyr->p, N->W->out_w, N->W->out_b, o2, (long long)B * Tp * Fb, 0, 0, st);
    gemm_b(N, o2, 2 * Fb, 0, false, N->W->basisI, 2 * Fb, 0, false, frames, Kp, 0, B * Tp, Kp, 2 * Fb, nullptr, nullptr, 1.f, 0, 1);
    launch_ola(frames, Kp, Tp, nfft, hop, N->inv_env, y, B, L, N->pad, cskip_b ? x : nullptr, cskip_b, cout_b, st);
  }
  N->have_tape = rec;
}

static void run_vjp(Net* N, const float* cot, float* gx) {
  const NetCfg& c = N->cfg;
  const int B = N->B, L = N->L, T = N->T, Tp = N->Tp, Fb = N->Fb, Kp = N->Kp, hop = c.hop, nfft = c.n_fft;
  hipStream_t st = N->st;
  for (auto& t : N->pool) t.ginit = 0;
  const size_t mark = N->arena.off;
  float* dframes = N->tmp((long long)B * Tp * Kp);
  float* do2 = N->tmp((long long)B * Tp * Fb * 2);
  if (!N->dry()) {
    launch_ola_adj(cot, B, L, N->pad, Tp, nfft, hop, N->inv_env, N->k_cout, dframes, Kp, st);
    gemm_b(N, dframes, Kp, 0, false, N->W->basisI, 2 * Fb, 0, true, do2, 2 * Fb, 0, B * Tp, 2 * Fb, Kp, nullptr, nullptr, 1.f, 0, 1);
    launch_mix2(do2, N->W->out_w, nullptr, N->pyr0->g, (long long)B * Tp * Fb, 1, 0, st);
  }
  N->pyr0->ginit = 1;
  for (int i = (int)N->tape.size() - 1; i >= 0; --i) N->tape[i]();
  float* dfx = N->tmp((long long)B * T * Kp);
  if (!N->dry()) {
    gemm_b(N, N->spec->g, 2 * Fb, (long long)Tp * 2 * Fb, false, N->W->basisF, Kp, 0, true, dfx, Kp, (long long)T * Kp, T, Kp, 2 * Fb, nullptr, nullptr,
           1.f, 0, B);
    launch_unpad_adj(dfx, Kp, T, nfft, hop, B, L, N->pad, 1.f, N->k_cin, N->k_cskip ? cot : nullptr, N->k_cskip, gx, st);
  }
  N->arena.off = mark;
}

int net_reserve(Net* N, int B, int L, int with_vjp, long long* bytes) {
  OptScope scope(&N->opt);
  Arena saved = N->arena;
  N->w4_need = 0;
  N->arena = Arena(); N->arena.dry = true;
  // the dry run takes the allocation sequence of the form with per-utterance EDM scalars (a superset of the one without): the pointers
  // only have to be non-null, nothing is dereferenced or launched while arena.dry is set
  static const float dry_scalars[1] = {0.f};
  N->vdry = 0;
  run_forward(N, nullptr, nullptr, dry_scalars, dry_scalars, dry_scalars, nullptr, B, L, with_vjp != 0);
  const int conv_fwd = N->vdry;
  if (with_vjp) run_vjp(N, nullptr, nullptr);
  const int conv_slots = std::max(conv_fwd, N->vdry - conv_fwd) + 1;
  const size_t need = N->arena.peak + (1 << 20);
  N->arena = saved;
  N->pool.clear(); N->tape.clear(); N->taps.clear(); N->have_tape = false;
  if (bytes) *bytes = (long long)need + (long long)N->w4_need * 4;
  if (N->w4_cap < N->w4_need) {
    if (N->w4_scratch) (void)hipFree(N->w4_scratch);
    N->w4_scratch = nullptr; N->w4_cap = 0;
    HIPCHK(hipMalloc(&N->w4_scratch, N->w4_need * 4));
    N->w4_cap = N->w4_need;
  }
  if (N->arena.cap < need) {
    if (N->arena.base) (void)hipFree(N->arena.base);
    N->arena.base = nullptr; N->arena.cap = 0;
    HIPCHK(hipMalloc(&N->arena.base, need));
    N->arena.cap = need;
  }
  N->vslots = conv_slots;
  if (N->opt.gemm == 2 && N->vmax_cap < (size_t)conv_slots * B * VMAX_SUB * VMAX_STRIDE) {
    if (N->vmax) (void)hipFree(N->vmax);
    N->vmax = nullptr; N->vmax_cap = 0;
    HIPCHK(hipMalloc(&N->vmax, (size_t)conv_slots * B * VMAX_SUB * VMAX_STRIDE * 4));
    N->vmax_cap = (size_t)conv_slots * B * VMAX_SUB * VMAX_STRIDE;
  }
  N->vslot_need[0] = N->vslot_need[1] = 0;
  N->rsv_B = B; N->rsv_L = L; N->rsv_vjp = with_vjp != 0;
  return BUDDY_OK;
}

int net_forward(Net* N, const float* x, const float* cnoise, const float* cin_b, const float* cskip_b, const float* cout_b, float* y, int B, int L,
                int save, hipStream_t st) {
  if (B < 1 || L < N->cfg.n_fft) { set_error("bad B or L"); return BUDDY_ERR_ARG; }
  OptScope scope(&N->opt);          // the launchers of this call read THIS handle's options
  const int T = 1 + L / N->cfg.hop, Tp = (T + 15) / 16 * 16;
  if (Tp % (1 << (N->cfg.nlev - 1))) { set_error("frames not divisible"); return BUDDY_ERR_ARG; }
  int rc = BUDDY_OK;
  if (!(N->rsv_B == B && N->rsv_L == L && N->rsv_vjp >= (save != 0))) rc = net_reserve(N, B, L, save, nullptr);   // host dry run only on a new shape
  if (rc) return rc;
  rc = ensure_env(N, Tp);
  if (rc) return rc;
  N->st = st;
  N->arena.dry = false; N->arena.overflow = false;
  if (N->opt.gemm == 2) { HIPCHK(hipMemsetAsync(N->vmax, 0, (size_t)(N->vslot_need[0] ? N->vslot_need[0] : N->vslots) * B * VMAX_SUB * VMAX_STRIDE * 4, st)); N->vslot = 0; }
  run_forward(N, x, cnoise, cin_b, cskip_b, cout_b, y, B, L, save != 0);
  N->vslot_need[0] = N->vslot;
  if (N->arena.overflow) { set_error("arena overflow"); return BUDDY_ERR_STATE; }
  if (N->prep_failed) { N->prep_failed = false; return BUDDY_ERR_HIP; }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

int net_vjp(Net* N, const float* cot, float* gx, hipStream_t st) {
  if (!N->have_tape) { set_error("vjp without a saved forward"); return BUDDY_ERR_STATE; }
  OptScope scope(&N->opt);
  N->st = st;
  if (N->opt.gemm == 2) { HIPCHK(hipMemsetAsync(N->vmax, 0, (size_t)(N->vslot_need[1] ? N->vslot_need[1] : N->vslots) * N->rsv_B * VMAX_SUB * VMAX_STRIDE * 4, st)); N->vslot = 0; }
  run_vjp(N, cot, gx);
  N->vslot_need[1] = N->vslot;
  if (N->arena.overflow) { set_error("arena overflow"); return BUDDY_ERR_STATE; }
  if (N->prep_failed) { N->prep_failed = false; return BUDDY_ERR_HIP; }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

int net_get_tap(Net* N, int module_idx, const float** p, int dims[4]) {
  for (auto& t : N->taps)
    if (t.first == module_idx) { *p = t.second->p; dims[0] = t.second->B; dims[1] = t.second->H; dims[2] = t.second->W; dims[3] = t.second->C; return BUDDY_OK; }
  set_error("no such tap");
  return BUDDY_ERR_ARG;
}

}  // namespace buddy
