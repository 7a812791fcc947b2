// Blind subband-filtering reverb operator on gfx950: batched over U utterances, hand-written forward AND analytic
// backward (no autograd), one C-ABI call per optimize_op (reference testing/EulerHeunSamplerDPS.py:71-113) and one per
// likelihood evaluation (:61-69).  Restates reference testing/operators/subband_filtering.py (SubbandFiltering :8-136,
// BlindSubbandFiltering :142-351), utils/reverb_utils.py:3-23 (hilbert / minimum_phase_version), utils/losses.py:59-64
// (l2_comp_stft_summean) and torch Adam (the optimizer the reference constructs, EulerHeunSamplerDPS.py:193) for the shipped op_hp (fix_EQ_extremes, single exponential per band set
// E >= 1, minimum_phase, fix_direct_path, clamp_decay, enforce_long_decay_in_second_exponential).
//
// Layouts: spectrograms [U][T][LDS_=1028] floats = 513 interleaved complex bins + 2 zero pad floats (16-B aligned rows,
// K % 4 == 0 for the MFMA GEMMs); subband filters are frame-major [U][Nf][1028] so that the filter IS a spectrogram of the
// RIR (cons() = iSTFT -> minimum phase -> STFT needs no transposes).  STFT/iSTFT (n_fft 1024, hann(512) zero-padded, hop
// 128) run as DFT-GEMMs on the fp32 MFMA kernel (K = 512: the window support); the 25856-point FFTs of the minimum-phase
// projection are two-stage Cooley-Tukey (101 x 256) direct kernels.
#include "common.h"
#include "net.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace buddy {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return BUDDY_ERR_HIP; } } while (0)

namespace {
constexpr int FB = 513, LDSP = 1028, NFFT = 1024, WIN = 512, HOP = 128;
constexpr int N2 = 25856, F1 = 101, F2 = 256;     // N2 = F1 * F2
constexpr double PI = 3.14159265358979323846;

// ------------------------------------------------------------------ small kernels
// Y[u][t][f] = sum_k H[u][k][f] * X[u][t + 1 - k][f]   (reference subband_filtering :67-74, one pre-impulse frame)
// A thread owns FOUR consecutive frames of one (utterance, bin): per tap it loads one H value and one new X frame (the other three
// slide through registers), i.e. 2 loads per 4 complex MACs instead of 8.  Per-output summation order is k ascending, as before.
constexpr int FT = 4;
__global__ __launch_bounds__(256) void fir_kernel_sb(const float* X, long long xs, const float* H, float* Y, int U, int T, int Nf) {
  const int TG = (T + FT - 1) / FT;
  const long long total = (long long)U * TG * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB); const int t0 = (int)((i / FB) % TG) * FT; const int u = (int)(i / ((long long)FB * TG));
    const float2* Xu = reinterpret_cast<const float2*>(X + (long long)u * xs) + f;
    const float2* Hu = reinterpret_cast<const float2*>(H + (long long)u * Nf * LDSP) + f;
    auto ldx = [&](int tt) { return (tt >= 0 && tt < T) ? Xu[(long long)tt * (LDSP / 2)] : make_float2(0.f, 0.f); };
    float ar[FT], ai[FT];
#pragma unroll
    for (int j = 0; j < FT; ++j) { ar[j] = 0.f; ai[j] = 0.f; }
    // window w[j] = X[t0 + j + 1 - k]; at k = 0: X[t0+1 .. t0+4]
    float2 w[FT];
#pragma unroll
    for (int j = 0; j < FT; ++j) w[j] = ldx(t0 + j + 1);
    const int kmax = (t0 + FT < Nf) ? t0 + FT : Nf - 1;          // beyond k = t + 1 every index is negative
    for (int k = 0; k <= kmax && k < Nf; ++k) {
      const float2 h = Hu[(long long)k * (LDSP / 2)];
#pragma unroll
      for (int j = 0; j < FT; ++j) { ar[j] += h.x * w[j].x - h.y * w[j].y; ai[j] += h.x * w[j].y + h.y * w[j].x; }
#pragma unroll
      for (int j = FT - 1; j > 0; --j) w[j] = w[j - 1];
      w[0] = ldx(t0 - k);
    }
#pragma unroll
    for (int j = 0; j < FT; ++j)
      if (t0 + j < T) reinterpret_cast<float2*>(Y + ((long long)u * T + t0 + j) * LDSP)[f] = make_float2(ar[j], ai[j]);
  }
}
// GX[u][t'][f] = sum_k conj(H[k]) * GY[t' - 1 + k]
__global__ __launch_bounds__(256) void fir_adjx_kernel(const float* GY, const float* H, float* GX, int U, int T, int Nf) {
  const long long total = (long long)U * T * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB); const int t = (int)((i / FB) % T); const int u = (int)(i / ((long long)FB * T));
    const float2* Gu = reinterpret_cast<const float2*>(GY + (long long)u * T * LDSP);
    const float2* Hu = reinterpret_cast<const float2*>(H + (long long)u * Nf * LDSP);
    float ar = 0.f, ai = 0.f;
    for (int k = 0; k < Nf; ++k) {
      const int tt = t - 1 + k;
      if (tt < 0) continue;
      if (tt >= T) break;
      const float2 h = Hu[(long long)k * (LDSP / 2) + f], g = Gu[(long long)tt * (LDSP / 2) + f];
      ar += h.x * g.x + h.y * g.y; ai += h.x * g.y - h.y * g.x;
    }
    reinterpret_cast<float2*>(GX + ((long long)u * T + t) * LDSP)[f] = make_float2(ar, ai);
  }
}
// GH[u][k][f] (+)= sum_t conj(X[t + 1 - k]) * GY[t]
// A thread owns FOUR consecutive taps k0..k0+3 of one (utterance, bin) and one QUARTER of the frame range (per frame: one GY load and
// one new X frame, the other three slide through registers); the four partial sums are added in fixed order through LDS.
constexpr int GH_SEG = 4, GH_F = 64;
__global__ __launch_bounds__(256) void fir_gradh_kernel(const float* X, long long xs, const float* GY, float* GH, int U, int T, int Nf, int accumulate) {
  __shared__ float2 red[GH_SEG][FT][GH_F];
  const int KG = (Nf + FT - 1) / FT, FGn = (FB + GH_F - 1) / GH_F;
  const int fl = threadIdx.x & (GH_F - 1), seg = threadIdx.x / GH_F;
  const int fg = blockIdx.x % FGn, kg = (blockIdx.x / FGn) % KG, u = blockIdx.x / (FGn * KG);
  const int f = fg * GH_F + fl, k0 = kg * FT;
  const bool ok = f < FB;
  const float2* Xu = reinterpret_cast<const float2*>(X + (long long)u * xs) + (ok ? f : 0);
  const float2* Gu = reinterpret_cast<const float2*>(GY + (long long)u * T * LDSP) + (ok ? f : 0);
  auto ldx = [&](int tt) { return (tt >= 0 && tt < T) ? Xu[(long long)tt * (LDSP / 2)] : make_float2(0.f, 0.f); };
  float ar[FT], ai[FT];
#pragma unroll
  for (int j = 0; j < FT; ++j) { ar[j] = 0.f; ai[j] = 0.f; }
  const int ts = k0 > 0 ? k0 - 1 : 0;                       // first frame with a non-negative index for tap k0
  const int len = (T - ts + GH_SEG - 1) / GH_SEG;
  int t = ts + seg * len;
  const int te = (t + len < T) ? t + len : T;
  float2 w[FT];                                             // window w[j] = X[t + 1 - k0 - j]
#pragma unroll
  for (int j = 0; j < FT; ++j) w[j] = ldx(t + 1 - k0 - j);
  for (; t + 8 <= te; t += 8) {                             // eight frames per trip: all 16 loads are issued before the arithmetic
    float2 g8[8], x8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { g8[q] = Gu[(long long)(t + q) * (LDSP / 2)]; x8[q] = ldx(t + q + 2 - k0); }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int j = 0; j < FT; ++j) { ar[j] += w[j].x * g8[q].x + w[j].y * g8[q].y; ai[j] += w[j].x * g8[q].y - w[j].y * g8[q].x; }
#pragma unroll
      for (int j = FT - 1; j > 0; --j) w[j] = w[j - 1];
      w[0] = x8[q];
    }
  }
  for (; t < te; ++t) {
    const float2 g = Gu[(long long)t * (LDSP / 2)];
#pragma unroll
    for (int j = 0; j < FT; ++j) { ar[j] += w[j].x * g.x + w[j].y * g.y; ai[j] += w[j].x * g.y - w[j].y * g.x; }
#pragma unroll
    for (int j = FT - 1; j > 0; --j) w[j] = w[j - 1];
    w[0] = ldx(t + 2 - k0);
  }
#pragma unroll
  for (int j = 0; j < FT; ++j) red[seg][j][fl] = make_float2(ar[j], ai[j]);
  __syncthreads();
  if (seg == 0 && ok) {
#pragma unroll
    for (int j = 0; j < FT; ++j) {
      if (k0 + j >= Nf) continue;
      float r = red[0][j][fl].x, im = red[0][j][fl].y;
#pragma unroll
      for (int sg = 1; sg < GH_SEG; ++sg) { r += red[sg][j][fl].x; im += red[sg][j][fl].y; }
      float2* o = reinterpret_cast<float2*>(GH + ((long long)u * Nf + k0 + j) * LDSP) + f;
      if (accumulate) { r += o->x; im += o->y; }
      *o = make_float2(r, im);
    }
  }
}

// ---- LDS-staged forms of the two FIR kernels of the optimisation loop (round 3).  The register-tiled kernels above re-load every X / H value
// from L2 once per 4 complex MACs and sit at 12 % of the VALU rate on dependent loads (8 frames per thread instead of 4 changed nothing: latency, not
// bytes); here a workgroup stages its (frames x 32 bins) slab once, the tap loop reads LDS only (two conflict-free ds_read_b64 per 8 complex MACs).
typedef float v2f __attribute__((ext_vector_type(2)));
// complex multiply-accumulate as TWO packed fmas (v_pk_fma_f32): acc += h.x * (w.x, w.y); acc += h.y * (-w.y, w.x)   [h * w]
__device__ __forceinline__ v2f cmac(v2f acc, float2 h, float2 w) {
  acc = __builtin_elementwise_fma(v2f{h.x, h.x}, v2f{w.x, w.y}, acc);
  return __builtin_elementwise_fma(v2f{-h.y, h.y}, v2f{w.y, w.x}, acc);
}
// acc += conj(w) * g:  (w.x g.x + w.y g.y,  w.x g.y - w.y g.x)
__device__ __forceinline__ v2f cmac_conj(v2f acc, float2 w, float2 g) {
  acc = __builtin_elementwise_fma(v2f{w.x, w.x}, v2f{g.x, g.y}, acc);
  return __builtin_elementwise_fma(v2f{w.y, -w.y}, v2f{g.y, g.x}, acc);
}
constexpr int FL_BINS = 32, FL_FT = 8, FL_TB = 64;           // bins per workgroup, frames per thread, frames per workgroup (8 frame groups)
// Y[u][t][f] = sum_k H[u][k][f] X[u][t + 1 - k][f]; grid (ceil(T / 64), ceil(FB / 32), U), 256 threads; LDS (64 + 2 Nf - 1) x 32 complex.
// One launch filters up to TWO independent inputs with the same H (FirSegs: the reconstruction term's STFT(x_den) and the regulariser's STFT(delta));
// the x tiles of the second follow those of the first.  Every launch of the captured loop costs ~5 us on top of its work.
struct FirSeg { const float* X; long long xs; float* Y; int T; int tiles; };
struct FirSegs { FirSeg s[2]; };
// Nf is a template parameter (the 25 856-point minimum-phase transforms fix it at 100 anyway): slab sizes, the tap loop and every row bound are
// compile-time, so the staging registers are exactly the rows that exist and the only conditions left are the tile's edges.
template <int Nf>
__global__ __launch_bounds__(256) void fir_sb_lds_kernel(FirSegs segs, const float* __restrict__ H) {
  extern __shared__ float2 fl_smem[];
  constexpr int nx = FL_TB + Nf - 1;                         // frames t0 - Nf + 2 ... t0 + 64
  float2* Xs = fl_smem;                                      // [nx][32]
  float2* Hs = fl_smem + nx * FL_BINS;                       // [Nf][32]
  const bool second = (int)blockIdx.x >= segs.s[0].tiles;
  const float* __restrict__ X = second ? segs.s[1].X : segs.s[0].X;
  float* __restrict__ Y = second ? segs.s[1].Y : segs.s[0].Y;
  const long long xs = second ? segs.s[1].xs : segs.s[0].xs;
  const int T = second ? segs.s[1].T : segs.s[0].T;
  const int u = blockIdx.z, f0 = blockIdx.y * FL_BINS, t0 = ((int)blockIdx.x - (second ? segs.s[0].tiles : 0)) * FL_TB;
  const int tid = threadIdx.x, b = tid & 31, g = tid >> 5;
  const float2* Xu = reinterpret_cast<const float2*>(X + (long long)u * xs);
  const float2* Hu = reinterpret_cast<const float2*>(H + (long long)u * Nf * LDSP);
  const bool fok = f0 + b < FB;
  const int xbase = t0 - Nf + 2;
  {   // all global loads of the slab are issued before the first LDS store (a load -> store loop would pay one L2 round trip per row);
      // unconditional loads from clamped rows / bins, the condition applied to the value (a predicated load is a branch with its own exec-mask
      // handling and wait)
    constexpr int NXR = (nx + 7) / 8, NHR = (Nf + 7) / 8;
    float2 vx[NXR], vh[NHR];
    const int fb = fok ? f0 + b : FB - 1;
#pragma unroll
    for (int i = 0; i < NXR; ++i) {
      const int r = g + 8 * i, tt = xbase + r;
      const bool okx = fok && tt >= 0 && tt < T;
      const float2 q = Xu[(long long)min(max(tt, 0), T - 1) * (LDSP / 2) + fb];
      vx[i] = okx ? q : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NHR; ++i) {
      const int k = g + 8 * i;
      const float2 q = Hu[(long long)min(k, Nf - 1) * (LDSP / 2) + fb];
      vh[i] = fok ? q : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NXR; ++i) { const int r = g + 8 * i; if (8 * i + 7 < nx || r < nx) Xs[r * FL_BINS + b] = vx[i]; }
#pragma unroll
    for (int i = 0; i < NHR; ++i) { const int k = g + 8 * i; if (8 * i + 7 < Nf || k < Nf) Hs[k * FL_BINS + b] = vh[i]; }
  }
  __syncthreads();
  const int tg = g * FL_FT;                                  // this thread's outputs: t0 + tg + j
  v2f ac[FL_FT];
  float2 w[FL_FT];                                           // w[j] = X[t0 + tg + j + 1 - k]  -> Xs row (tg + j + 1 - k) - (2 - Nf)
#pragma unroll
  for (int j = 0; j < FL_FT; ++j) { ac[j] = v2f{0.f, 0.f}; w[j] = Xs[(tg + j + Nf - 1) * FL_BINS + b]; }
#pragma unroll 8
  for (int k = 0; k < Nf; ++k) {                             // k ascending per output, as in the register-tiled kernel (unroll = window depth: the shift is renaming)
    const float2 h = Hs[k * FL_BINS + b];
#pragma unroll
    for (int j = 0; j < FL_FT; ++j) ac[j] = cmac(ac[j], h, w[j]);
#pragma unroll
    for (int j = FL_FT - 1; j > 0; --j) w[j] = w[j - 1];
    const int r = tg + Nf - 2 - k;                           // row of X[t0 + tg - k]; r = -1 only for the value shifted in after the last tap (never used)
    w[0] = Xs[max(r, 0) * FL_BINS + b];
  }
  if (fok) {
#pragma unroll
    for (int j = 0; j < FL_FT; ++j)
      if (t0 + tg + j < T) reinterpret_cast<float2*>(Y + ((long long)u * T + t0 + tg + j) * LDSP)[f0 + b] = make_float2(ac[j].x, ac[j].y);
  }
}
constexpr int FIR_NF = 100;
// GH[u][k][f] (+)= sum_t conj(X[u][t + 1 - k][f]) GY[u][t][f]; grid (ceil(Nf / 16), ceil(FB / 32), U), 256 threads = 32 bins x 2 tap groups of 8 x 4
// frame slots; frames in chunks of 64 (slot s takes frames 16 s ... 16 s + 15 of each chunk); the four slot sums are added in fixed order
constexpr int GL_TPT = 17, GL_TAPS = 2 * GL_TPT, GL_CH = 64, GL_SLOT = 16;      // taps per thread / per workgroup; frames per chunk / per slot
struct GradSeg { const float* X; long long xs; const float* GY; int T; };
struct GradSegs { GradSeg s[2]; int n; };
// one launch sums the tap gradients of up to two (X, GY) pairs; each pair is accumulated from zero and reduced on its own, the second sum is added to
// the first (exactly what two launches, the second accumulating into GH, did)
__global__ __launch_bounds__(256) void fir_gradh_lds_kernel(GradSegs segs, float* __restrict__ GH, int Nf, int accumulate) {
  __shared__ float2 sm[(GL_CH + GL_TAPS - 1 + GL_CH) * FL_BINS];
  __shared__ float2 tot_s[GL_TAPS][FL_BINS];                 // the sums of the segments done so far (written and read by the same sl == 0 thread)
  float2* Xs = sm;                                           // [GL_CH + GL_TAPS - 1][32]: frames c0 + 1 - (k0 + GL_TAPS - 1) ... c0 + 64 - k0
  float2* Gs = sm + (GL_CH + GL_TAPS - 1) * FL_BINS;         // [GL_CH][32]
  float2 (*red)[GL_TAPS][FL_BINS] = reinterpret_cast<float2 (*)[GL_TAPS][FL_BINS]>(sm);     // [4][GL_TAPS][32], after the chunk loop (same memory)
  static_assert(4 * GL_TAPS <= GL_CH + GL_TAPS - 1 + GL_CH, "the slot sums reuse the slab memory");
  // grid (bin tiles, U, tap blocks): the linear workgroup id modulo 8 (= the XCD it lands on: 17 * 8 is a multiple of 8) does not depend on the tap block,
  // so the blocks that read the same (X, GY) slab share one L2
  const int u = blockIdx.y, f0 = blockIdx.x * FL_BINS, k0 = blockIdx.z * GL_TAPS;
  const int tid = threadIdx.x, b = tid & 31, kg = (tid >> 5) & 1, sl = tid >> 6;
  const bool fok = f0 + b < FB;
  const int kk = k0 + GL_TPT * kg;                           // this thread's taps kk ... kk + GL_TPT - 1
  const int fb = fok ? f0 + b : FB - 1;
  for (int sg_i = 0; sg_i < segs.n; ++sg_i) {
    const int T = segs.s[sg_i].T;
    const float2* Xu = reinterpret_cast<const float2*>(segs.s[sg_i].X + (long long)u * segs.s[sg_i].xs);
    const float2* Gu = reinterpret_cast<const float2*>(segs.s[sg_i].GY + (long long)u * T * LDSP);
    v2f ac[GL_TPT];
#pragma unroll
    for (int j = 0; j < GL_TPT; ++j) ac[j] = v2f{0.f, 0.f};
    // the next chunk's slab is requested before the arithmetic of the current one (registers), stored to LDS after it
    constexpr int NXR = (GL_CH + GL_TAPS - 1 + 7) / 8, NGR = GL_CH / 8;
    const int g8 = tid >> 5;
    float2 vx[NXR], vg[NGR];
    auto fetch = [&](int c0) {
      const int xb = c0 + 1 - (k0 + GL_TAPS - 1);
#pragma unroll
      for (int i = 0; i < NXR; ++i) {                        // unconditional loads from clamped rows / bins, the condition applied to the value
        const int r = g8 + 8 * i, tt = xb + r;
        const bool okx = r < GL_CH + GL_TAPS - 1 && fok && tt >= 0 && tt < T;
        const float2 q = Xu[(long long)min(max(tt, 0), T - 1) * (LDSP / 2) + fb];
        vx[i] = okx ? q : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NGR; ++i) {
        const int tt = c0 + g8 + 8 * i;
        const float2 q = Gu[(long long)min(tt, T - 1) * (LDSP / 2) + fb];
        vg[i] = (fok && tt < T) ? q : make_float2(0.f, 0.f);
      }
    };
    fetch(0);
    for (int c0 = 0; c0 < T; c0 += GL_CH) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NXR; ++i) { const int r = g8 + 8 * i; if (r < GL_CH + GL_TAPS - 1) Xs[r * FL_BINS + b] = vx[i]; }
#pragma unroll
      for (int i = 0; i < NGR; ++i) Gs[(g8 + 8 * i) * FL_BINS + b] = vg[i];
      __syncthreads();
      if (c0 + GL_CH < T) fetch(c0 + GL_CH);
      // frame t = c0 + 16 sl + q: X[t + 1 - kk - j] = Xs row (16 sl + q + GL_TAPS - 1 - GL_TPT kg - j)
      const int base = GL_SLOT * sl + GL_TAPS - 1 - GL_TPT * kg;
      float2 w[GL_TPT];
#pragma unroll
      for (int j = 0; j < GL_TPT; ++j) w[j] = Xs[(base - j) * FL_BINS + b];
#pragma unroll
      for (int q = 0; q < GL_SLOT; ++q) {
        const float2 gy = Gs[(GL_SLOT * sl + q) * FL_BINS + b];
#pragma unroll
        for (int j = 0; j < GL_TPT; ++j) ac[j] = cmac_conj(ac[j], w[j], gy);
#pragma unroll
        for (int j = GL_TPT - 1; j > 0; --j) w[j] = w[j - 1];
        if (q + 1 < GL_SLOT) w[0] = Xs[(base + q + 1) * FL_BINS + b];
      }
    }
    __syncthreads();                                         // red may still be read by the previous segment's reduction
#pragma unroll
    for (int j = 0; j < GL_TPT; ++j) red[sl][GL_TPT * kg + j][b] = make_float2(ac[j].x, ac[j].y);
    __syncthreads();
    if (sl == 0) {
#pragma unroll
      for (int j = 0; j < GL_TPT; ++j) {
        float r = red[0][GL_TPT * kg + j][b].x, im = red[0][GL_TPT * kg + j][b].y;
#pragma unroll
        for (int sg = 1; sg < 4; ++sg) { r += red[sg][GL_TPT * kg + j][b].x; im += red[sg][GL_TPT * kg + j][b].y; }
        const float2 pr = tot_s[GL_TPT * kg + j][b];
        tot_s[GL_TPT * kg + j][b] = sg_i == 0 ? make_float2(r, im) : make_float2(r + pr.x, im + pr.y);
      }
    }
  }
  if (sl == 0 && fok) {
#pragma unroll
    for (int j = 0; j < GL_TPT; ++j) {
      const int k = kk + j;
      if (k >= Nf) continue;
      float2* o = reinterpret_cast<float2*>(GH + ((long long)u * Nf + k) * LDSP) + f0 + b;
      float r = tot_s[GL_TPT * kg + j][b].x, im = tot_s[GL_TPT * kg + j][b].y;
      if (accumulate) { r += o->x; im += o->y; }
      *o = make_float2(r, im);
    }
  }
}

// compressed spectrum: Xc = (|X| + 1e-8)^p * exp(j angle X)   (reference losses.py:59-64)
__device__ __forceinline__ float2 compress(float2 x, float p) {
  const float r = sqrtf(x.x * x.x + x.y * x.y);
  if (r == 0.f) return make_float2(powf(1e-8f, p), 0.f);      // angle(0) = 0
  const float rho = powf(r + 1e-8f, p) / r;
  return make_float2(x.x * rho, x.y * rho);
}
__global__ __launch_bounds__(256) void compress_kernel(const float* X, float* Xc, long long rows, float p) {
  const long long total = rows * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / FB; const int f = (int)(i % FB);
    reinterpret_cast<float2*>(Xc + r * LDSP)[f] = compress(reinterpret_cast<const float2*>(X + r * LDSP)[f], p);
  }
}
// loss_u = kappa * sum_{t,f} |Rc - comp(Xh)|^2 ;  G = d loss / d Xh (as dRe + j dIm).  partial sums per block (deterministic).
// Up to two independent terms per launch (LossJobs: blocks of the second follow those of the first).  Rc == nullptr: the target is the compressed
// spectrum of what G holds on entry (the regulariser's detached STFT(rir + t n), overwritten in place by its gradient; was compress_kernel).
struct LossJob { const float* Rc; const float* Xh; float* G; double* partial; int T; float kappa; int blocks; };
struct LossJobs { LossJob j[2]; };
__global__ __launch_bounds__(256) void comp_loss_kernel(LossJobs jobs, float p) {
  __shared__ double red[256];
  const bool second = (int)blockIdx.x >= jobs.j[0].blocks;
  const LossJob& jb = jobs.j[second ? 1 : 0];
  const float* Rc = jb.Rc; const float* Xh = jb.Xh; float* G = jb.G;
  const int T = jb.T, nblk = jb.blocks, bx = (int)blockIdx.x - (second ? jobs.j[0].blocks : 0);
  const float kappa = jb.kappa;
  const int u = blockIdx.y;
  const long long total = (long long)T * FB;
  double acc = 0.0;
  for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)nblk * 256) {
    const long long t = i / FB; const int f = (int)(i % FB);
    const long long row = ((long long)u * T + t) * LDSP;
    const float2 x = reinterpret_cast<const float2*>(Xh + row)[f];
    const float2 rc = Rc ? reinterpret_cast<const float2*>(Rc + row)[f] : compress(reinterpret_cast<const float2*>(G + row)[f], p);
    const float r = sqrtf(x.x * x.x + x.y * x.y);
    float2 g = make_float2(0.f, 0.f);
    float2 xc;
    if (r == 0.f) {
      xc = make_float2(powf(1e-8f, p), 0.f);
    } else {
      const float ir = 1.f / r, re = r + 1e-8f;
      const float cr = x.x * ir, ci = x.y * ir;             // e^{j theta}
      const float rho = powf(re, p);
      xc = make_float2(rho * cr, rho * ci);
      const float dr = xc.x - rc.x, di = xc.y - rc.y;       // D = Xc_hat - Rc
      const float a = dr * cr + di * ci;                    // Re(conj(D) e^{j theta})
      const float b = dr * ci - di * cr;                    // Im(conj(D) e^{j theta})
      const float gr = 2.f * kappa * a * p * (rho / re);    // d rho / d r = p (r + eps)^(p - 1): one powf and a division instead of two powf
      const float gt = -2.f * kappa * rho * b * ir;         // (1/r) dL/dtheta
      g = make_float2(gr * cr - gt * ci, gr * ci + gt * cr);
    }
    const float dr = xc.x - rc.x, di = xc.y - rc.y;
    acc += (double)(dr * dr + di * di);
    if (G) reinterpret_cast<float2*>(G + row)[f] = g;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) jb.partial[(long long)u * nblk + bx] = red[0];
}
__global__ void loss_finalize_kernel(const double* partial, int nblk, float kappa, float* loss, int accumulate) {
  const int u = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += partial[(long long)u * nblk + i];
  const float v = (float)(s * (double)kappa);
  loss[u] = accumulate ? loss[u] + v : v;
}

// ---- filter design (reference :212-251) ----
struct DesignTabs { const int* idx; const float* frac; const float* corr; const float* dpm; const int* fge; };   // per-bin knot index / fraction; OLA corr[Nf]; dpm[Nf][FB]
// dm[u][n][j], j = 0..K-1 knots (rows 0 and K-1 are zero): sum_e w[e][j-1] * exp(decay[e][j-1])^(-n)
__global__ void design_dm_kernel(const float* decay, const float* wts, float* logdm, float* dmv, int U, int E, int NB, int Nf) {
  const int K = NB + 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U * Nf * K) return;
  const int j = i % K, n = (i / K) % Nf, u = i / (K * Nf);
  float v = 0.f;
  if (j >= 1 && j <= NB)
    for (int e = 0; e < E; ++e) v += wts[((long long)u * E + e) * NB + j - 1] * powf(expf(decay[((long long)u * E + e) * NB + j - 1]), -(float)n);
  dmv[i] = v;
  logdm[i] = logf(v + 1e-6f);
}
__global__ __launch_bounds__(256) void design_A_kernel(const float* logdm, DesignTabs tb, float* A, float* Apre, int U, int K, int Nf) {
  const long long total = (long long)U * Nf * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB), n = (int)((i / FB) % Nf), u = (int)(i / ((long long)FB * Nf));
    const float* l = logdm + ((long long)u * Nf + n) * K;
    const int j = tb.idx[f];
    const float v0 = l[j], v1 = l[j + 1];
    const float e = expf(v0 + tb.frac[f] * (v1 - v0));
    Apre[i] = e;
    A[i] = (e + 1e-6f) / tb.corr[n] + tb.dpm[(long long)n * FB + f];
  }
}
// design_dm_kernel, design_A_kernel and the H0 frames Fin[u][k+1][f] = A[u][k][f] * exp(j phi[u][k][f]) (rows 0 and Nf+1 stay zero) in one pass:
// a block owns one (utterance, frame) row -- its K knot values first (LDS), then the 513 bins.  Two nodes less in the captured loop.
__global__ __launch_bounds__(256) void design_row_kernel(const float* decay, const float* wts, DesignTabs tb, const float* phi, float* logdm, float* dmv, float* A,
                                                         float* Apre, float* Fin, int U, int E, int NB, int Nf, int* step_inc) {
  __shared__ float l[64];
  const int K = NB + 2;
  const int n = blockIdx.x % Nf, u = blockIdx.x / Nf;
  if (step_inc && blockIdx.x == 0 && threadIdx.x == 0) *step_inc += 1;     // captured loop: the Adam step counter advances with the first kernel of an iteration
  for (int j = threadIdx.x; j < K; j += 256) {
    float v = 0.f;
    if (j >= 1 && j <= NB)
      for (int e = 0; e < E; ++e) v += wts[((long long)u * E + e) * NB + j - 1] * powf(expf(decay[((long long)u * E + e) * NB + j - 1]), -(float)n);
    const float lg = logf(v + 1e-6f);
    const long long o = ((long long)u * Nf + n) * K + j;
    dmv[o] = v; logdm[o] = lg; l[j] = lg;
  }
  __syncthreads();
  const long long row = ((long long)u * Nf + n) * FB;
  const float cn = tb.corr[n];
  float2* F = reinterpret_cast<float2*>(Fin + ((long long)u * (Nf + 2) + n + 1) * LDSP);
  for (int f = threadIdx.x; f < FB; f += 256) {
    const int j = tb.idx[f];
    const float v0 = l[j], v1 = l[j + 1];
    const float e = expf(v0 + tb.frac[f] * (v1 - v0));
    const float a = (e + 1e-6f) / cn + tb.dpm[(long long)n * FB + f];
    Apre[row + f] = e;
    A[row + f] = a;
    float sn, cs; sincosf(phi[row + f], &sn, &cs);
    F[f] = make_float2(a * cs, a * sn);
  }
}
// backward of the H0 frames and of the knot interpolation in one pass
//   gA = Re(conj(e^{j phi}) G_Fin), gphi = A Im(...);  g_logdm[u][n][j] = sum_f [idx(f) == j] (1 - frac) gi + [idx(f) + 1 == j] frac gi,  gi = gA / corr[n] * Apre
// a block owns one (utterance, frame) row of 513 bins.  Phase 1 forms gphi and
// gi = gA * Apre per bin (gi parked in LDS, gA never reaches memory); phase 2: wave w reduces the knots j = w, w + 4, ... -- lanes stride over the
// contiguous bins with idx in {j - 1, j} (idx[] is non-decreasing), fixed-order butterfly.  The one-thread-per-knot form walked ~50 dependent
// loads per thread (28 us per call).
__global__ __launch_bounds__(256) void h0_bwd_knots_kernel(const float* GFin, const float* A, const float* Apre, const float* phi, DesignTabs tb, const float* dmv,
                                                           float* gphi, float* gdm, int U, int K, int Nf) {
  __shared__ float gi[FB + 3];
  __shared__ int id_s[FB + 3];
  __shared__ float fr_s[FB + 3];
  const int n = blockIdx.x % Nf, u = blockIdx.x / Nf;
  const long long row = ((long long)u * Nf + n) * FB;
  const float2* G = reinterpret_cast<const float2*>(GFin + ((long long)u * (Nf + 2) + n + 1) * LDSP);
  for (int f = threadIdx.x; f < FB; f += 256) {
    const float2 g = G[f];
    const float a = A[row + f], ap = Apre[row + f];
    float sn, cs; sincosf(phi[row + f], &sn, &cs);
    gphi[row + f] = a * (-g.x * sn + g.y * cs);
    gi[f] = (g.x * cs + g.y * sn) * ap;
    id_s[f] = tb.idx[f]; fr_s[f] = tb.frac[f];
  }
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = w; j < K; j += 4) {
    const int f_lo = tb.fge[j], f_hi = tb.fge[j + 2];          // first bins with idx >= j - 1 and idx >= j + 1
    float acc = 0.f;
    for (int f = f_lo + lane; f < f_hi; f += 64) {
      const int id = id_s[f];
      if (id == j) acc += (1.f - fr_s[f]) * gi[f];
      else if (id + 1 == j) acc += fr_s[f] * gi[f];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const long long o = ((long long)u * Nf + n) * K + j;
    if (lane == 0) gdm[o] = acc / tb.corr[n] / (dmv[o] + 1e-6f);      // d/d dm of log(dm + 1e-6)
  }
}
// one WAVE per parameter (u, e, b): lanes take the frames n = lane, lane + 64, ...; fixed-order butterfly sum (one thread per parameter looping
// over the Nf frames with a powf each took 35 us on a single workgroup, ten times per sampler step)
__global__ __launch_bounds__(256) void design_bwd_params_kernel(const float* gdm, const float* decay, const float* wts, float* gdecay, float* gw, int U, int E,
                                                                int NB, int Nf) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= U * E * NB) return;
  const int b = i % NB, u = i / (E * NB);
  const int K = NB + 2;
  const float base = expf(decay[i]), w = wts[i];
  float gd = 0.f, gwv = 0.f;
  for (int n = lane; n < Nf; n += 64) {
    const float pw = powf(base, -(float)n);
    const float g = gdm[((long long)u * Nf + n) * K + b + 1];
    gwv += g * pw;
    gd += g * w * (-(float)n) * pw;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { gd += __shfl_xor(gd, off, 64); gwv += __shfl_xor(gwv, off, 64); }
  if (lane == 0) { gdecay[i] = gd; gw[i] = gwv; }
}

// ---- 25856-point complex FFT, two stages (N2 = 101 * 256): n = 256 n1 + n2, k = k1 + 101 k2 ----
// stage 1: Y1[u][n2][k1] = tw(n2 k1) * sum_{n1} x[256 n1 + n2] W101^(n1 k1).  101 is prime: a naive 101-point DFT per column n2.  A block parks
// its S1_COLS = 4 columns of x and the twiddles in LDS; THREAD k1 forms the outputs of all four columns, so one (bank-conflicted: the index n1 k1
// mod 101 is scattered over the lanes) twiddle read and one 16/32-byte row read serve four complex MACs.  The one-output-per-thread form (two LDS
// reads per MAC, 404 outputs on 256 threads) was bound by the LDS pipe: 8 us per pass.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
constexpr int S1_COLS = 4, S1_THREADS = 128;
// acc[c] += W[(n1 k1) mod 101] * X[n1][c], n1 = 0 .. 100 ascending
__device__ __forceinline__ void dft101x4(const float2* X, const float2* W, int k1, v2f (&acc)[4]) {
  const float4* X4 = reinterpret_cast<const float4*>(X);
  int idx = 0, n1 = 0;
  for (; n1 + 4 <= F1; n1 += 4) {
    float2 w[4]; float4 xa[4], xb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[j] = W[idx]; xa[j] = X4[2 * (n1 + j)]; xb[j] = X4[2 * (n1 + j) + 1]; idx += k1; if (idx >= F1) idx -= F1; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = cmac(acc[0], w[j], make_float2(xa[j].x, xa[j].y)); acc[1] = cmac(acc[1], w[j], make_float2(xa[j].z, xa[j].w));
      acc[2] = cmac(acc[2], w[j], make_float2(xb[j].x, xb[j].y)); acc[3] = cmac(acc[3], w[j], make_float2(xb[j].z, xb[j].w));
    }
  }
  for (; n1 < F1; ++n1) {
    const float2 w = W[idx]; const float4 xa = X4[2 * n1], xb = X4[2 * n1 + 1];
    acc[0] = cmac(acc[0], w, make_float2(xa.x, xa.y)); acc[1] = cmac(acc[1], w, make_float2(xa.z, xa.w));
    acc[2] = cmac(acc[2], w, make_float2(xb.x, xb.y)); acc[3] = cmac(acc[3], w, make_float2(xb.z, xb.w));
    idx += k1; if (idx >= F1) idx -= F1;
  }
}
// The input is a REAL signal of Lr samples zero-padded to N2, and it is never materialised: it is the overlap-add of the 512-sample frames
// fr[u][0..Tsrc) at j = n + Q times env[j] (what ola_kernel wrote), or, with fr == nullptr, the array xr[u][0..Lr).  ZERO0: sample 0 forced to zero
// (hm[0] is a constant of the projection).
struct S1In { const float* xr; int Lr; const float* fr; int Tsrc, Q; const float* env; };
template <bool ZERO0>
__global__ __launch_bounds__(S1_THREADS) void fft_stage1_kernel(S1In in, float2* y1, const float2* w101, const float2* twN, int sign) {
  __shared__ float2 W[F1];
  __shared__ __align__(16) float X[F1 * S1_COLS];             // real
  const int u = blockIdx.y, n20 = blockIdx.x * S1_COLS;
  for (int i = threadIdx.x; i < F1; i += S1_THREADS) W[i] = make_float2(w101[i].x, sign * w101[i].y);
  for (int i = threadIdx.x; i < F1 * S1_COLS; i += S1_THREADS) {
    const int n1 = i / S1_COLS, c = i - n1 * S1_COLS, n = F2 * n1 + n20 + c;
    float v;
    const bool inr = n < in.Lr && (!ZERO0 || n > 0);
    const int nc = n < in.Lr ? n : 0;                       // loads are unconditional (clamped index), the condition selects the value afterwards
    if (in.fr) {
      const int j = nc + in.Q, tq = j >> 7, m0 = j & (HOP - 1);
      float fv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {                         // ascending frame order, as ola_kernel
        const int tt = tq - 3 + k;
        const bool okf = tt >= 0 && tt < in.Tsrc;
        const float q = in.fr[((long long)u * in.Tsrc + (okf ? tt : 0)) * WIN + m0 + HOP * (3 - k)];
        fv[k] = okf ? q : 0.f;
      }
      const float e = in.env[j];
      v = (((0.f + fv[0]) + fv[1]) + fv[2]) + fv[3];
      v *= e;
    } else {
      v = in.xr[(long long)u * in.Lr + nc];
    }
    X[i] = inr ? v : 0.f;
  }
  __syncthreads();
  const int k1 = threadIdx.x;
  if (k1 >= F1) return;
  float2 t[S1_COLS];
#pragma unroll
  for (int c = 0; c < S1_COLS; ++c) t[c] = twN[(long long)(n20 + c) * F1 + k1];     // requested before the 101-term sum, needed after it
  v2f acc[S1_COLS];
#pragma unroll
  for (int c = 0; c < S1_COLS; ++c) acc[c] = v2f{0.f, 0.f};
  const float4* X4 = reinterpret_cast<const float4*>(X);
  int idx = 0;
#pragma unroll 4
  for (int n1 = 0; n1 < F1; ++n1) {                          // real input: one packed fma per column and term
    const float2 w = W[idx]; const float4 x = X4[n1];
    const v2f wv{w.x, w.y};
    acc[0] = __builtin_elementwise_fma(v2f{x.x, x.x}, wv, acc[0]); acc[1] = __builtin_elementwise_fma(v2f{x.y, x.y}, wv, acc[1]);
    acc[2] = __builtin_elementwise_fma(v2f{x.z, x.z}, wv, acc[2]); acc[3] = __builtin_elementwise_fma(v2f{x.w, x.w}, wv, acc[3]);
    idx += k1; if (idx >= F1) idx -= F1;
  }
#pragma unroll
  for (int c = 0; c < S1_COLS; ++c)
    y1[(long long)u * N2 + (long long)(n20 + c) * F1 + k1] = cmul(make_float2(acc[c].x, acc[c].y), make_float2(t[c].x, sign * t[c].y));
}
// stage 2: X[k1 + 101 k2] = scale * sum_{n2} Y1[n2][k1] W256^(n2 k2), a 256-point DFT per (utterance, k1) column done as
// 16 x 16: n2 = 16 a + r, k2 = b + 16 c  =>  W256^(n2 k2) = W16^(a b) * W256^(r b) * W16^(r c).  A block owns 16 columns; thread (col, r)
// does the 16-point DFT over a in registers (radix 4 x 4), applies W256^(r b), exchanges through LDS, thread (col, b) does the one over r.
// y[b] = sum_a x[a] exp(SGN * 2 pi i a b / 16)
template <int SGN>
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  // w^e = exp(SGN 2 pi i e / 16) for the exponents a0*b0 in {0,1,2,3,4,6,9}
  const float2 w1 = make_float2(C1, SGN * S1), w2 = make_float2(C2, SGN * C2), w3 = make_float2(S1, SGN * C1), w4 = make_float2(0.f, (float)SGN),
               w6 = make_float2(-C2, SGN * C2), w9 = make_float2(-C1, -SGN * S1);
  float2 u[4][4];                                           // u[a0][b0] = sum_{a1} x[4 a1 + a0] j^(a1 b0),  j = SGN * i
#pragma unroll
  for (int a0 = 0; a0 < 4; ++a0) {
    const float2 x0 = x[a0], x1 = x[4 + a0], x2 = x[8 + a0], x3 = x[12 + a0];
    const float2 s02 = make_float2(x0.x + x2.x, x0.y + x2.y), d02 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 s13 = make_float2(x1.x + x3.x, x1.y + x3.y), d13 = make_float2(x1.x - x3.x, x1.y - x3.y);
    const float2 jd = make_float2(-SGN * d13.y, SGN * d13.x);            // j * (x1 - x3)
    u[a0][0] = make_float2(s02.x + s13.x, s02.y + s13.y);
    u[a0][1] = make_float2(d02.x + jd.x, d02.y + jd.y);
    u[a0][2] = make_float2(s02.x - s13.x, s02.y - s13.y);
    u[a0][3] = make_float2(d02.x - jd.x, d02.y - jd.y);
  }
  u[1][1] = cmul(u[1][1], w1); u[1][2] = cmul(u[1][2], w2); u[1][3] = cmul(u[1][3], w3);
  u[2][1] = cmul(u[2][1], w2); u[2][2] = cmul(u[2][2], w4); u[2][3] = cmul(u[2][3], w6);
  u[3][1] = cmul(u[3][1], w3); u[3][2] = cmul(u[3][2], w6); u[3][3] = cmul(u[3][3], w9);
#pragma unroll
  for (int b0 = 0; b0 < 4; ++b0) {                          // y[4 b1 + b0] = sum_{a0} u[a0][b0] j^(a0 b1)
    const float2 x0 = u[0][b0], x1 = u[1][b0], x2 = u[2][b0], x3 = u[3][b0];
    const float2 s02 = make_float2(x0.x + x2.x, x0.y + x2.y), d02 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 s13 = make_float2(x1.x + x3.x, x1.y + x3.y), d13 = make_float2(x1.x - x3.x, x1.y - x3.y);
    const float2 jd = make_float2(-SGN * d13.y, SGN * d13.x);
    x[b0] = make_float2(s02.x + s13.x, s02.y + s13.y);
    x[4 + b0] = make_float2(d02.x + jd.x, d02.y + jd.y);
    x[8 + b0] = make_float2(s02.x - s13.x, s02.y - s13.y);
    x[12 + b0] = make_float2(d02.x - jd.x, d02.y - jd.y);
  }
}
// ---- the chain of four transforms of the minimum-phase projection (and of its backward), five kernels -------------------------------------------
// Consecutive transforms alternate between the two index splittings of N2 = 101 * 256, so that the LAST stage of one and the FIRST stage of the
// next work on the same data in the same workgroup, with the elementwise step between them applied in registers:
//   form I  (above):  n = 256 n1 + n2, k = k1 + 101 k2:   [101-point over n1, twiddle]  ->  [256-point over n2]
//   form II        :  n = m1 + 101 m2, k = 256 q1 + q2:   [256-point over m2, twiddle W_N^(m1 q2)]  ->  [101-point over m1]
// (both take and deliver NATURAL order).  FFT_1 (I) | pw | FFT_2 (II) | pw | FFT_3 (I) | pw | FFT_4 (II)  becomes
//   stage1  ->  [256 . pw . 256]  ->  [101 . pw . 101]  ->  [256 . pw . 256]  ->  stageB
// instead of 8 stage kernels + 2 elementwise ones; the intermediate images are y1[u][n2][k1] (after a 101-point stage) and z[u][q2][m1] (after a
// 256-point stage of form II) -- the same [256][101] shape, so twN[q2 * 101 + m1] serves both twiddles.
constexpr int S2_COLS = 8, S2_SHIFT = 3, S2_PITCH = 16 * S2_COLS + 4, S2_THREADS = 16 * S2_COLS;
// 256-point DFT of v over the index held as (register a, thread r): in  v[a] = x[16 a + r];  out v[c] = X[b + 16 c] with b = r
template <int SGN>
__device__ __forceinline__ void dft256(float2 (&v)[16], float2* S, const float2* W, int col, int r) {
  dft16<SGN>(v);
#pragma unroll
  for (int b = 0; b < 16; ++b) S[b * S2_PITCH + r * S2_COLS + col] = cmul(v[b], W[(r * b) & (F2 - 1)]);
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) v[rr] = S[r * S2_PITCH + rr * S2_COLS + col];
  dft16<SGN>(v);
}
// the elementwise steps (reference reverb_utils.py:9-23 and their adjoints), on element k of utterance u:
//   PW 0  X = FFT([h, 0]):          Hf = X, M = |X|,                          x' = log(M + 1e-8)
//   PW 1  X = hilbert(log M):       phim = -Im X,                             x' = M e^{j phim}
//   PW 2  X = FFT(g) / N2 (= G_Z):  gM = Re(conj(X) e^{j phim}),              x' = -M Im(conj(X) e^{j phim})
//   PW 3  X = hilbert(g_phi):       g = gM + Im X / (M + 1e-8),               x' = g Hf / M   (0 where M = 0)
struct MpArrays { float2* Hf; float* M; float* phim; float* gM; };
struct MpIn { float m, ph, g; float2 h; };            // what the step reads of the stored arrays (loaded for all 16 elements of a thread up front:
                                                      // behind the step's own stores the compiler may not hoist them, and each would cost a round trip)
template <int PW>
__device__ __forceinline__ MpIn mp_load(long long i, const MpArrays& A) {
  MpIn r; r.m = 0.f; r.ph = 0.f; r.g = 0.f; r.h = make_float2(0.f, 0.f);
  if (PW >= 1) r.m = A.M[i];
  if (PW == 2) r.ph = A.phim[i];
  if (PW == 3) { r.g = A.gM[i]; r.h = A.Hf[i]; }
  return r;
}
template <int PW>
__device__ __forceinline__ float2 mp_pointwise(float2 X, long long i, const MpIn& in, const MpArrays& A) {
  if (PW == 0) {
    const float m = sqrtf(X.x * X.x + X.y * X.y);
    A.Hf[i] = X; A.M[i] = m;
    return make_float2(logf(m + 1e-8f), 0.f);
  } else if (PW == 1) {
    const float ph = -X.y;
    A.phim[i] = ph;
    float sn, cs; sincosf(ph, &sn, &cs);
    return make_float2(in.m * cs, in.m * sn);
  } else if (PW == 2) {
    float sn, cs; sincosf(in.ph, &sn, &cs);
    const float a = X.x * cs + X.y * sn, b = X.x * sn - X.y * cs;
    A.gM[i] = a;
    return make_float2(-in.m * b, 0.f);
  } else {
    const float m = in.m;
    const float g = in.g + X.y / (m + 1e-8f);
    return m > 0.f ? make_float2(g * in.h.x / m, g * in.h.y / m) : make_float2(0.f, 0.f);
  }
}
// [256-point stage of form I] . pw . [256-point stage of form II + twiddle]: y1[u][n2][k1] -> z[u][q2][m1], m1 = k1.  A block owns 16 columns k1.
template <int SGN, int PW>
__global__ __launch_bounds__(S2_THREADS) void fft_mid256_kernel(const float2* y1, float2* z, const float2* w256, const float2* twN, float scale, MpArrays A) {
  __shared__ float2 S[16 * S2_PITCH];
  __shared__ float2 W[F2];
  for (int i = threadIdx.x; i < F2; i += S2_THREADS) W[i] = make_float2(w256[i].x, SGN * w256[i].y);
  const int u = blockIdx.y;
  const int col = threadIdx.x & (S2_COLS - 1), r = threadIdx.x >> S2_SHIFT;
  const int k1 = blockIdx.x * S2_COLS + col;
  const bool ok = k1 < F1;
  const float2* yu = y1 + (long long)u * N2;
  float2 v[16];
  MpIn in[16];
  const int k1c = ok ? k1 : 0;                              // unconditional loads (clamped column), the out-of-range columns are never stored
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = yu[(16 * a + r) * F1 + k1c];
  const long long e0 = (long long)u * N2 + k1c + (long long)F1 * r;                 // element c of this thread: e0 + 101 * 16 c
#pragma unroll
  for (int c = 0; c < 16; ++c) in[c] = mp_load<PW>(e0 + (long long)F1 * 16 * c, A);
  __syncthreads();
  dft256<SGN>(v, S, W, col, r);                             // v[c] = X[k1 + 101 (r + 16 c)]
  if (ok) {
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = mp_pointwise<PW>(make_float2(v[c].x * scale, v[c].y * scale), e0 + (long long)F1 * 16 * c, in[c], A);
  }
  __syncthreads();                                          // S is reused
  dft256<SGN>(v, S, W, col, r);                             // m2 = r + 16 c was (thread r, register c): v[c'] = sum over m2, q2 = r + 16 c'
  if (ok) {
    float2* zu = z + (long long)u * N2;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int q2 = r + 16 * c;
      const float2 t = twN[q2 * F1 + k1];
      zu[q2 * F1 + k1] = cmul(v[c], make_float2(t.x, SGN * t.y));
    }
  }
}
// [101-point stage of form II, sign -] . Hilbert window . [101-point stage of form I + twiddle, sign +]: z[u][q2][m1] -> y1[u][n2][k1], n2 = q2
__global__ __launch_bounds__(S1_THREADS) void fft_mid101_kernel(const float2* z, float2* y1, const float2* w101, const float2* twN) {
  __shared__ float2 Wm[F1], Wp[F1];
  __shared__ __align__(16) float2 X[F1 * S1_COLS], X2[F1 * S1_COLS];
  const int u = blockIdx.y, n20 = blockIdx.x * S1_COLS;
  for (int i = threadIdx.x; i < F1; i += S1_THREADS) { const float2 w = w101[i]; Wm[i] = make_float2(w.x, -w.y); Wp[i] = w; }
  const float2* zu = z + (long long)u * N2 + (long long)n20 * F1;
  {
    constexpr int NT = (F1 * S1_COLS + S1_THREADS - 1) / S1_THREADS;
    float2 v[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int i = threadIdx.x + S1_THREADS * q; v[q] = i < F1 * S1_COLS ? zu[i] : make_float2(0.f, 0.f); }
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int i = threadIdx.x + S1_THREADS * q; if (i < F1 * S1_COLS) { const int c = i / F1, m1 = i - c * F1; X[m1 * S1_COLS + c] = v[q]; } }
  }
  __syncthreads();
  const int k1 = threadIdx.x;                                // first pass: q1
  float2 t[S1_COLS];
  if (k1 < F1) {
#pragma unroll
    for (int c = 0; c < S1_COLS; ++c) t[c] = twN[(long long)(n20 + c) * F1 + k1];
    v2f acc[S1_COLS];
#pragma unroll
    for (int c = 0; c < S1_COLS; ++c) acc[c] = v2f{0.f, 0.f};
    dft101x4(X, Wm, k1, acc);                                // element n = 256 q1 + n2 of the inner spectrum
#pragma unroll
    for (int c = 0; c < S1_COLS; ++c) {
      const float w = (F2 * k1 + n20 + c) < N2 / 2 ? 2.f : 0.f;
      X2[k1 * S1_COLS + c] = make_float2(acc[c].x * w, acc[c].y * w);
    }
  }
  __syncthreads();
  if (k1 < F1) {
    v2f acc[S1_COLS];
#pragma unroll
    for (int c = 0; c < S1_COLS; ++c) acc[c] = v2f{0.f, 0.f};
    dft101x4(X2, Wp, k1, acc);
#pragma unroll
    for (int c = 0; c < S1_COLS; ++c) y1[(long long)u * N2 + (long long)(n20 + c) * F1 + k1] = cmul(make_float2(acc[c].x, acc[c].y), t[c]);
  }
}
// last stage of a form-II transform with REAL output: xr[u][n] = scale * Re sum_{m1} z[u][q2][m1] W101^(sign m1 q1), n = 256 q1 + q2 < Lo;
// first_set: sample 0 := first (the minimum-phase filter's leading tap is a constant of the projection)
__global__ __launch_bounds__(S1_THREADS) void fft_stageB_real_kernel(const float2* z, float* xr, int Lo, const float2* w101, int sign, float scale, int first_set, float first) {
  __shared__ float2 W[F1];
  __shared__ __align__(16) float2 X[F1 * S1_COLS];
  const int u = blockIdx.y, n20 = blockIdx.x * S1_COLS;
  if (n20 >= Lo) return;                                     // (uniform) nothing of this block's columns survives even for q1 = 0
  for (int i = threadIdx.x; i < F1; i += S1_THREADS) W[i] = make_float2(w101[i].x, -sign * w101[i].y);   // conjugated: Re(v w) = v . conj(w) as a dot product
  const float2* zu = z + (long long)u * N2 + (long long)n20 * F1;
  {
    constexpr int NT = (F1 * S1_COLS + S1_THREADS - 1) / S1_THREADS;
    float2 v[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int i = threadIdx.x + S1_THREADS * q; v[q] = i < F1 * S1_COLS ? zu[i] : make_float2(0.f, 0.f); }
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int i = threadIdx.x + S1_THREADS * q; if (i < F1 * S1_COLS) { const int c = i / F1, m1 = i - c * F1; X[m1 * S1_COLS + c] = v[q]; } }
  }
  __syncthreads();
  const int q1 = threadIdx.x;
  if (q1 >= F1 || F2 * q1 + n20 >= Lo) return;
  v2f acc[S1_COLS];
#pragma unroll
  for (int c = 0; c < S1_COLS; ++c) acc[c] = v2f{0.f, 0.f};
  const float4* X4 = reinterpret_cast<const float4*>(X);
  int idx = 0;
#pragma unroll 4
  for (int m1 = 0; m1 < F1; ++m1) {                          // (re, im) partial products of the real part: one packed fma per column and term
    const float2 w = W[idx]; const float4 xa = X4[2 * m1], xb = X4[2 * m1 + 1];
    const v2f wv{w.x, w.y};
    acc[0] = __builtin_elementwise_fma(v2f{xa.x, xa.y}, wv, acc[0]); acc[1] = __builtin_elementwise_fma(v2f{xa.z, xa.w}, wv, acc[1]);
    acc[2] = __builtin_elementwise_fma(v2f{xb.x, xb.y}, wv, acc[2]); acc[3] = __builtin_elementwise_fma(v2f{xb.z, xb.w}, wv, acc[3]);
    idx += q1; if (idx >= F1) idx -= F1;
  }
#pragma unroll
  for (int c = 0; c < S1_COLS; ++c) {
    const int n = F2 * q1 + n20 + c;
    if (n < Lo) xr[(long long)u * Lo + n] = (first_set && n == 0) ? first : (acc[c].x + acc[c].y) * scale;
  }
}

// ---- 1024-point FFTs for the operator STFT (n_fft 1024, hann(512) zero-padded, hop 128: reference subband_filtering.py:41-80) ----
// One wave per frame, 16 points per lane, three in-register passes (16 x 16 x 4) with two LDS exchanges:
//   n = 64 a + r, k = b + 16 (c0 + 4 c1):  W^(nk) = W16^(ab) * W1024^(rb) * W4^(r1 c0) * W64^(r0 c0) * W16^(r0 c1),  r = 16 r1 + r0.
// Replaces the four DFT-as-GEMM products (forward, inverse and their two adjoints: 2 x 512 x 1028 MACs per frame on the matrix cores) by
// 5 N log2 N flops per frame; the window, the 1/sqrt(sum w^2) norm, the one-sided factors {1, 2, ..., 2, 1}/N and the scale are folded in.
constexpr int FLD = 66;                       // LDS row stride (complex) of the [16][64] exchange image
// The twiddles a lane needs depend on the lane only: its row W1024^(r b), b < 16, of the [64][16] table (`Wrow`, 128 contiguous bytes per lane)
// and W64^(r0 c0), c0 = 1..3 (`W64t`, [16][4]).  They are loaded straight into registers while the frame's samples are in flight (Fft1024Tw);
// the first version parked the 1024-entry table in LDS per workgroup: 8 KB more LDS (3 instead of 4 workgroups per CU) and a load -> store -> barrier
// prologue before the first butterfly.
struct Fft1024Tw { float2 row[16]; float2 c[3]; };
__device__ __forceinline__ void fft1024_load_tw(Fft1024Tw& t, const float2* __restrict__ Wrow, const float2* __restrict__ W64t, int r) {
  const float4* p = reinterpret_cast<const float4*>(Wrow + r * 16);
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float4 q = p[i]; t.row[2 * i] = make_float2(q.x, q.y); t.row[2 * i + 1] = make_float2(q.z, q.w); }
  const float4* c = reinterpret_cast<const float4*>(W64t + (r & 15) * 4);
  const float4 c0 = c[0], c1 = c[1];
  t.c[0] = make_float2(c0.z, c0.w); t.c[1] = make_float2(c1.x, c1.y); t.c[2] = make_float2(c1.z, c1.w);
}
// LDS written by this wave, read by this wave: no workgroup barrier, only the wave's own ordering
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int SGN>
__device__ __forceinline__ void fft1024_core(float2 (&v)[16], float2* S, const Fft1024Tw& tw, int r) {
  dft16<SGN>(v);                                                       // over a: v[b] = sum_a x[64 a + r] W16^(a b)
#pragma unroll
  for (int b = 0; b < 16; ++b) S[b * FLD + r] = cmul(v[b], make_float2(tw.row[b].x, SGN * tw.row[b].y));
  wave_lds_sync();                                                     // the exchange image S is private to the wave
  const int r0 = r & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                        // 4-point DFTs over r1, in place
    const int b = 4 * j + (r >> 4);
    float2* q = S + b * FLD + r0;
    const float2 x0 = q[0], x1 = q[16], x2 = q[32], x3 = q[48];
    const float2 s02 = make_float2(x0.x + x2.x, x0.y + x2.y), d02 = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 s13 = make_float2(x1.x + x3.x, x1.y + x3.y), d13 = make_float2(x1.x - x3.x, x1.y - x3.y);
    const float2 jd = make_float2(-SGN * d13.y, SGN * d13.x);           // (SGN i) (x1 - x3)
    const float2 z0 = make_float2(s02.x + s13.x, s02.y + s13.y), z1 = make_float2(d02.x + jd.x, d02.y + jd.y);
    const float2 z2 = make_float2(s02.x - s13.x, s02.y - s13.y), z3 = make_float2(d02.x - jd.x, d02.y - jd.y);
    const float2 w1 = tw.c[0], w2 = tw.c[1], w3 = tw.c[2];                                          // W64^(r0 c0)
    q[0] = z0;
    q[16] = cmul(z1, make_float2(w1.x, SGN * w1.y));
    q[32] = cmul(z2, make_float2(w2.x, SGN * w2.y));
    q[48] = cmul(z3, make_float2(w3.x, SGN * w3.y));
  }
  wave_lds_sync();                                                     // the exchange image S is private to the wave
  const float2* q = S + (r & 15) * FLD + 16 * (r >> 4);               // lane = (b = r & 15, c0 = r >> 4)
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = q[i];
  dft16<SGN>(v);                                                       // over r0: v[c1] = X[r + 64 c1]
}
// rows of real, windowed 512-sample frames -> [rows][1028] one-sided spectra.  frame (row) starts at src[(row / Tn) * sA + (row % Tn) * hop];
// out[k] = fac * (cf ? cf(k) : 1) * sum_n src[n] win[n] exp(-2 pi i k n / 1024)
//
// Where the frames come from (R2cSrc): every elementwise / gather neighbour of the transform is folded into its first load, because each node
// of the captured optimisation loop costs ~5 us whatever it does.  Frame t of utterance u covers the samples s = 128 t + n - P, n < 512, of
//   GATHER = false: the signal sig[u][0..Ls)                                                        (was: a zero-padded copy)
//   GATHER = true : the overlap-add of fr[u][0..Tsrc)[512] at j = s + Q, times envA[j]              (was: ola_kernel -> signal -> pad_const)
// zero outside [0, Ls); then (+ add_scale * add[u][s]) and (* envB[128 t + n]) when given           (was: pad_const's noise term / ola_adj_kernel)
// Every product is formed in the order the separate kernels formed it, so the results are bit-identical to the unfused chain.
struct R2cSrc {
  const float* sig; const float* fr; int Ls, P, Tsrc, Q; const float* envA; const float* envB;
  const float* add; float add_scale; const float* add_scale_dev;
};
// One launch transforms up to THREE independent frame sets (R2cJobs; same fac / cf): the blocks of job i + 1 follow those of job i.
struct R2cJob { R2cSrc sc; int Tn; int rows; int blocks; float* out; };
struct R2cJobs { R2cJob j[3]; };
template <bool GATHER>
__global__ __launch_bounds__(256) void fft1024_r2c_kernel(R2cJobs jobs, const float* __restrict__ win, const float2* __restrict__ Wrow,
                                                          const float2* __restrict__ W64t, float fac, int cf) {
  __shared__ float2 S[4][16 * FLD];
  const int w = threadIdx.x >> 6, r = threadIdx.x & 63;
  Fft1024Tw tw; fft1024_load_tw(tw, Wrow, W64t, r);
  int blk = blockIdx.x;
  const int ji = blk < jobs.j[0].blocks ? 0 : (blk < jobs.j[0].blocks + jobs.j[1].blocks ? 1 : 2);
  if (ji >= 1) blk -= jobs.j[0].blocks;
  if (ji == 2) blk -= jobs.j[1].blocks;
  const R2cSrc& sc = jobs.j[ji].sc;
  const int Tn = jobs.j[ji].Tn;
  float* __restrict__ out = jobs.j[ji].out;
  const long long row = (long long)blk * 4 + w;
  const bool ok = row < jobs.j[ji].rows;
  const int u = ok ? (int)(row / Tn) : 0, t = ok ? (int)(row % Tn) : 0;
  const float add_scale = sc.add ? (sc.add_scale_dev ? *sc.add_scale_dev : sc.add_scale) : 0.f;
  float2 v[16];
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = make_float2(0.f, 0.f);
  if (ok) {
    // all loads of the wave's 8 x 64 samples are issued before the first use (a load -> use loop would pay one L2 round trip per sample), and
    // UNCONDITIONALLY, from an index clamped into the array, the condition applied to the loaded value afterwards: a `cond ? p[i] : 0` load is a
    // branch with its own exec-mask save / restore and wait (the first version of this kernel spent 765 scalar and 151 wait instructions on them);
    // sample j of the overlap-add has exactly four candidate frames j / 128 - 3 .. j / 128, summed in ascending frame order as ola_kernel does
    float fv[8][GATHER ? 4 : 1], ea[8], eb[8], ad[8];
    bool in[8], fin[8][GATHER ? 4 : 1];
    const long long ubase = (long long)u * sc.Ls;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int n = 64 * a + r, s = HOP * t + n - sc.P;
      in[a] = s >= 0 && s < sc.Ls;
      const int sc_s = in[a] ? s : 0;
      if (GATHER) {
        const int j = sc_s + sc.Q, tq = j >> 7, m0 = j & (HOP - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int tt = tq - 3 + k;
          fin[a][k] = tt >= 0 && tt < sc.Tsrc;
          fv[a][k] = sc.fr[((long long)u * sc.Tsrc + (fin[a][k] ? tt : 0)) * WIN + m0 + HOP * (3 - k)];
        }
        ea[a] = sc.envA[j];
      } else {
        fv[a][0] = sc.sig[ubase + sc_s];
      }
      if (sc.add) ad[a] = sc.add[ubase + sc_s];
      if (sc.envB) eb[a] = sc.envB[HOP * t + n];
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      float val = fv[a][0];
      if (GATHER) {
        val = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) val += fin[a][k] ? fv[a][k] : 0.f;
        val *= ea[a];
      }
      if (sc.add) val += add_scale * ad[a];
      if (sc.envB) val = val * eb[a];
      v[a].x = (in[a] ? val : 0.f) * win[64 * a + r];
    }
  }
  fft1024_core<-1>(v, S[w], tw, r);
  if (!ok) return;
  float2* o = reinterpret_cast<float2*>(out + row * LDSP);
#pragma unroll
  for (int c1 = 0; c1 < 8; ++c1) {
    const int k = r + 64 * c1;
    const float g = (cf && k != 0) ? 2.f * fac : fac;
    o[k] = make_float2(v[c1].x * g, v[c1].y * g);
  }
  if (r == 0) o[512] = make_float2(v[8].x * fac, v[8].y * fac);
  if (r == 1) o[513] = make_float2(0.f, 0.f);
}
// [rows][1028] one-sided spectra -> [rows][512] real frames: frames[n] = fac * win[n] * Re sum_{k <= 512} (cf ? cf(k) : 1) in[k] exp(+2 pi i k n / 1024)
struct C2rJob { const float* in; float* frames; int rows; int blocks; };
struct C2rJobs { C2rJob j[2]; };
__global__ __launch_bounds__(256) void fft1024_c2r_kernel(C2rJobs jobs, const float* __restrict__ win, const float2* __restrict__ Wrow,
                                                          const float2* __restrict__ W64t, float fac, int cf) {
  __shared__ float2 S[4][16 * FLD];
  const int w = threadIdx.x >> 6, r = threadIdx.x & 63;
  Fft1024Tw tw; fft1024_load_tw(tw, Wrow, W64t, r);
  const bool second = (int)blockIdx.x >= jobs.j[0].blocks;
  const float* __restrict__ in = second ? jobs.j[1].in : jobs.j[0].in;
  float* __restrict__ frames = second ? jobs.j[1].frames : jobs.j[0].frames;
  const long long row = (long long)((int)blockIdx.x - (second ? jobs.j[0].blocks : 0)) * 4 + w;
  const bool ok = row < (second ? jobs.j[1].rows : jobs.j[0].rows);
  const float2* f = reinterpret_cast<const float2*>(in + (ok ? row : 0) * LDSP);
  float2 v[16];
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    const int k = 64 * a + r;
    float2 x = make_float2(0.f, 0.f);
    if (ok && k <= 512) { x = f[k]; if (cf && k != 0 && k != 512) { x.x *= 2.f; x.y *= 2.f; } }
    v[a] = x;
  }
  fft1024_core<1>(v, S[w], tw, r);
  if (!ok) return;
#pragma unroll
  for (int c1 = 0; c1 < 8; ++c1) { const int n = r + 64 * c1; frames[row * WIN + n] = fac * win[n] * v[c1].x; }
}

// ---- Adam (torch Adam (the optimizer the reference constructs, EulerHeunSamplerDPS.py:193) single-tensor arithmetic: lerp, addcmul, addcdiv) + projection ----
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, const int* step_dev, const float2* bc_tab) {
  if (step_dev) { const float2 bc = bc_tab[*step_dev]; bc1 = bc.x; bc2_sqrt = bc.y; }   // captured-graph mode: bias corrections by table
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i];
    if (wd != 0.f) gi += wd * p[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}
__global__ void set_scalar_kernel(float* dst, float v) { *dst = v; }
__global__ void step_inc_kernel(int* step) { *step += 1; }
__global__ void project_kernel(float* decay, float* wts, int U, int E, int NB, float dmin, float dmax, float wlo, float whi, int clamp_decay, int long2nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U * NB) return;
  const int b = i % NB, u = i / NB;
  float d0 = 0.f, w0 = 0.f;
  for (int e = 0; e < E; ++e) {
    float* d = decay + ((long long)u * E + e) * NB + b;
    float* w = wts + ((long long)u * E + e) * NB + b;
    if (clamp_decay) {
      float hi = dmax;
      if (e > 0 && long2nd) hi = fminf(d0 / 1.01f, dmax);
      *d = fminf(fmaxf(*d, dmin), hi);
      if (e == 0) d0 = *d;
    }
    if (e == 0) { *w = fminf(fmaxf(*w, wlo), whi); w0 = *w; }
    else *w = fminf(fmaxf(*w, wlo), w0);
  }
}
// One launch for the whole parameter update of an iteration of the captured loop (was step_inc + 3 x adam + project = 5 nodes): blocks
// [0, pb) take the phases, the rest one (utterance, band) pair per wave -- the gradients of its E decays and E weights (was design_bwd_params_kernel),
// Adam on them, then the projection of exactly those values (project_params reads nothing else).  Same per-element arithmetic as adam_kernel / project_kernel.  The step counter is advanced
// by the FIRST kernel of the iteration (design_dm_kernel), so every block here reads the same value.
__device__ __forceinline__ float adam_one(float p, float gi, float* m, float* v, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  if (wd != 0.f) gi += wd * p;
  const float mi = *m + (gi - *m) * (1.f - b1);
  const float vi = *v * b2 + (1.f - b2) * gi * gi;
  *m = mi; *v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  return p - (lr / bc1) * (mi / denom);
}
struct AdamAll { float *decay, *wts, *phi; const float* gphi; float *m_d, *v_d, *m_w, *v_w, *m_p, *v_p; };
__global__ __launch_bounds__(256) void adam_all_kernel(AdamAll a, const float* gdm, int Nf, long long np, int pb, int U, int E, int NB, float lr, float b1, float b2,
                                                       float eps, float wd, const int* step_dev, const float2* bc_tab, float dmin, float dmax, float wlo, float whi,
                                                       int clamp_decay, int long2nd) {
  const float2 bc = bc_tab[*step_dev];
  if ((int)blockIdx.x < pb) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < np; i += (long long)pb * 256)
      a.phi[i] = adam_one(a.phi[i], a.gphi[i], a.m_p + i, a.v_p + i, lr, b1, b2, eps, wd, bc.x, bc.y);
    return;
  }
  // one WAVE per (utterance, band): for each of its E exponentials the gradient of (decay, weight) from g_dm exactly as design_bwd_params_kernel
  // forms it (lanes over the frames, fixed-order butterfly -- every lane ends with the sums), then Adam and the projection
  const int i = ((int)blockIdx.x - pb) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= U * NB) return;
  const int b = i % NB, u = i / NB;
  const int K = NB + 2;
  float d0 = 0.f, w0 = 0.f;
  for (int e = 0; e < E; ++e) {
    const long long q = ((long long)u * E + e) * NB + b;
    const float dq = a.decay[q], wq = a.wts[q];
    const float base = expf(dq);
    float gd = 0.f, gwv = 0.f;
    for (int n = lane; n < Nf; n += 64) {
      const float pw = powf(base, -(float)n);
      const float g = gdm[((long long)u * Nf + n) * K + b + 1];
      gwv += g * pw;
      gd += g * wq * (-(float)n) * pw;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { gd += __shfl_xor(gd, off, 64); gwv += __shfl_xor(gwv, off, 64); }
    float md = a.m_d[q], vd = a.v_d[q], mw = a.m_w[q], vw = a.v_w[q];
    float d = adam_one(dq, gd, &md, &vd, lr, b1, b2, eps, wd, bc.x, bc.y);
    float w = adam_one(wq, gwv, &mw, &vw, lr, b1, b2, eps, wd, bc.x, bc.y);
    if (clamp_decay) {
      float hi = dmax;
      if (e > 0 && long2nd) hi = fminf(d0 / 1.01f, dmax);
      d = fminf(fmaxf(d, dmin), hi);
      if (e == 0) d0 = d;
    }
    if (e == 0) { w = fminf(fmaxf(w, wlo), whi); w0 = w; }
    else w = fminf(fmaxf(w, wlo), w0);
    if (lane == 0) { a.decay[q] = d; a.wts[q] = w; a.m_d[q] = md; a.v_d[q] = vd; a.m_w[q] = mw; a.v_w[q] = vw; }
  }
}
// reference layout (U, F, Nf) <-> frame-major (U, Nf, F)
__global__ __launch_bounds__(256) void transpose_fk_kernel(const float* src, float* dst, int U, int Nf, int to_frame_major) {
  const long long total = (long long)U * Nf * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB), k = (int)((i / FB) % Nf), u = (int)(i / ((long long)FB * Nf));
    const long long a = ((long long)u * FB + f) * Nf + k;       // reference index
    if (to_frame_major) dst[i] = src[a]; else dst[a] = src[i];
  }
}
__global__ __launch_bounds__(256) void angle_kernel(const float* H, float* phi, int U, int Nf) {
  const long long total = (long long)U * Nf * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB); const long long r = i / FB;
    const float2 h = reinterpret_cast<const float2*>(H + r * LDSP)[f];
    phi[i] = atan2f(h.y, h.x);
  }
}
__global__ __launch_bounds__(256) void unit_phase_kernel(const float* Nz, float* phi, int U, int Nf) {   // phi = angle(N[:, 1:]) frame k <- noise frame k+1
  const long long total = (long long)U * Nf * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB), k = (int)((i / FB) % Nf), u = (int)(i / ((long long)FB * Nf));
    const float2 h = reinterpret_cast<const float2*>(Nz + ((long long)u * (Nf + 1) + k + 1) * LDSP)[f];
    phi[i] = atan2f(h.y, h.x);
  }
}
__global__ __launch_bounds__(256) void copy_h_kernel(const float* H, float* out, int U, int Nf) {   // frame-major padded -> reference (U,F,Nf,2)
  const long long total = (long long)U * Nf * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB), k = (int)((i / FB) % Nf), u = (int)(i / ((long long)FB * Nf));
    const float2 h = reinterpret_cast<const float2*>(H + ((long long)u * Nf + k) * LDSP)[f];
    reinterpret_cast<float2*>(out)[((long long)u * FB + f) * Nf + k] = h;
  }
}

__global__ __launch_bounds__(256) void copy_spec_kernel(const float* X, float* out, int U, int T) {   // padded frame-major -> reference (U,F,T,2)
  const long long total = (long long)U * T * FB;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % FB), t = (int)((i / FB) % T), u = (int)(i / ((long long)FB * T));
    reinterpret_cast<float2*>(out)[((long long)u * FB + f) * T + t] = reinterpret_cast<const float2*>(X + ((long long)u * T + t) * LDSP)[f];
  }
}
// reference (U,F,T,2) -> padded frame-major [U][T][LDSP] (pad floats zeroed): the inverse of copy_spec_kernel / copy_h_kernel (T = Nf)
__global__ __launch_bounds__(256) void spec_from_ref_kernel(const float* in, float* X, int U, int T) {
  const long long total = (long long)U * T * (LDSP / 2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % (LDSP / 2)), t = (int)((i / (LDSP / 2)) % T), u = (int)(i / ((long long)(LDSP / 2) * T));
    reinterpret_cast<float2*>(X + ((long long)u * T + t) * LDSP)[f] = f < FB ? reinterpret_cast<const float2*>(in)[((long long)u * FB + f) * T + t] : make_float2(0.f, 0.f);
  }
}
// plain minimum-phase output: hm = Re(o)[:Lm] (no direct-path override)

inline int gridf(long long total) { long long g = (total + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1; return (int)g; }
}  // namespace

// =================================================================== host-side operator object
struct BlindOp {
  BlindOpCfg c;
  int U = 0, L = 0, T = 0, Td = 0, Lr = 0, Lh = 0, Lm = 0, E = 0, NB = 0, K = 0, Nf = 0;
  hipStream_t st = nullptr;
  std::vector<void*> allocs;
  // tables
  float *ones = nullptr, *env_T = nullptr, *env_d = nullptr, *env_c = nullptr;
  float norm = 1.f;
  int *idx = nullptr, *fge = nullptr; float *frac = nullptr, *corr = nullptr, *dpm = nullptr;
  float2 *w101 = nullptr, *w256 = nullptr, *twN = nullptr, *w1024r = nullptr, *w64t = nullptr; float* win = nullptr;
  // parameters + Adam state
  float *decay = nullptr, *wts = nullptr, *phi = nullptr;
  float *m_d = nullptr, *v_d = nullptr, *m_w = nullptr, *v_w = nullptr, *m_p = nullptr, *v_p = nullptr;
  int adam_step = 0;
  // captured optimize_op graph (launch-bound loop: ~72 small kernels per Adam iteration): per-call inputs live at fixed device addresses
  static constexpr int MAXSTEP = 65536;
  int* d_step = nullptr; float* d_scal = nullptr; float2* bc_tab = nullptr; float tab_b1 = -1.f, tab_b2 = -1.f;
  float *xden_buf = nullptr, *noise_buf = nullptr; int noise_iters = 0;
  hipStream_t cap_stream = nullptr; hipGraphExec_t gexec = nullptr; int g_iters = -1; float g_hp[7] = {0, 0, 0, 0, 0, 0, 0};
  // state
  float *H = nullptr, *Yc = nullptr, *Xdelta = nullptr;
  // work buffers
  float *frames = nullptr, *X1 = nullptr, *X2 = nullptr, *X3 = nullptr, *Ybuf = nullptr, *sig1 = nullptr, *sig2 = nullptr;
  // second scratch set: the RIR-regulariser chain of an iteration shares every launch with the reconstruction chain (param_grads)
  float *frames_b = nullptr, *X2_b = nullptr, *X3_b = nullptr, *Ybuf_b = nullptr; double* partial_b = nullptr;
  int loss_norm = 0;                 // 0 l2_comp_stft_summean (default), 1 l2_comp_stft_sum, 2 l2_comp_stft_mean
  bool fused_loop = false;           // inside the captured optimisation loop: step counter in design_dm, no loss finalisation, one Adam launch
  bool big_lds = false;              // fir_sb_lds_kernel may take > 64 KB of dynamic LDS (set once at creation, outside any stream capture)
  float *A = nullptr, *Apre = nullptr, *logdm = nullptr, *dmv = nullptr, *gdm = nullptr, *Fin = nullptr, *GFin = nullptr, *GH = nullptr;
  float *gphi = nullptr, *gdecay = nullptr, *gw = nullptr, *hm = nullptr, *gh0 = nullptr;
  float2 *c1 = nullptr, *c2 = nullptr, *Hf = nullptr;
  float *Mabs = nullptr, *phim = nullptr, *gM = nullptr;
  double* partial = nullptr; float* losses = nullptr;
  float *rir = nullptr, *Rc = nullptr;

  template <typename Tp> int dalloc(Tp** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(Tp)) != hipSuccess) return 1;
    if (hipMemset(q, 0, n * sizeof(Tp)) != hipSuccess) return 1;
    allocs.push_back(q); *p = (Tp*)q;
    return 0;
  }
  ~BlindOp() {
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    for (void* q : allocs) (void)hipFree(q);
  }

  // ---- primitive routines (all batched over U, on stream st) ----
  // the four STFT-type transforms: 1024-point FFT kernels (round 1 ran them as DFT GEMMs on the matrix cores: 42 us per launch instead of 9)
  // up to three (source, frame count, output) jobs in one launch; all jobs of a launch gather from frames, or none does
  struct R2cReq { R2cSrc sc; int Tn; float* out; };
  void r2c_multi(const R2cReq* rq, int n, float fac, int cf) {
    R2cJobs jobs; std::memset(&jobs, 0, sizeof(jobs));
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
      jobs.j[i].sc = rq[i].sc; jobs.j[i].Tn = rq[i].Tn; jobs.j[i].rows = U * rq[i].Tn; jobs.j[i].blocks = (U * rq[i].Tn + 3) / 4; jobs.j[i].out = rq[i].out;
      blocks += (unsigned)jobs.j[i].blocks;
    }
    if (rq[0].sc.fr) hipLaunchKernelGGL(fft1024_r2c_kernel<true>, dim3(blocks), dim3(256), 0, st, jobs, (const float*)win, (const float2*)w1024r, (const float2*)w64t, fac, cf);
    else hipLaunchKernelGGL(fft1024_r2c_kernel<false>, dim3(blocks), dim3(256), 0, st, jobs, (const float*)win, (const float2*)w1024r, (const float2*)w64t, fac, cf);
  }
  void r2c(const R2cSrc& sc, int Tn, float* out, float fac, int cf) { const R2cReq rq{sc, Tn, out}; r2c_multi(&rq, 1, fac, cf); }
  void c2r2(const float* in0, long long rows0, float* fr0, const float* in1, long long rows1, float* fr1, float fac, int cf) {
    C2rJobs jobs; std::memset(&jobs, 0, sizeof(jobs));
    jobs.j[0].in = in0; jobs.j[0].frames = fr0; jobs.j[0].rows = (int)rows0; jobs.j[0].blocks = (int)((rows0 + 3) / 4);
    jobs.j[1].in = in1; jobs.j[1].frames = fr1; jobs.j[1].rows = (int)rows1; jobs.j[1].blocks = (int)((rows1 + 3) / 4);
    hipLaunchKernelGGL(fft1024_c2r_kernel, dim3((unsigned)(jobs.j[0].blocks + jobs.j[1].blocks)), dim3(256), 0, st, jobs, (const float*)win, (const float2*)w1024r, (const float2*)w64t, fac, cf);
  }
  void c2r(const float* in, long long rows, float* fr, float fac, int cf) { c2r2(in, rows, fr, nullptr, 0, nullptr, fac, cf); }
  // X[u][t][:] = scale * STFT frames of s (frame t starts at sample 128 t - P), Tn frames
  void stft(const float* s, int Ls, int P, int Tn, float scale, float* X, const float* add = nullptr, float add_scale = 0.f, const float* add_scale_dev = nullptr) {
    r2c(R2cSrc{s, nullptr, Ls, P, 0, 0, nullptr, nullptr, add, add_scale, add_scale_dev}, Tn, X, scale, 0);
  }
  // y[u][s] = sum_t frames_t[s + Q - 128 t] * inv_env[s + Q],  frames = scale * iDFT(Y) * window
  void istft(const float* Y, int Tn, int Q, const float* inv_env, int Ls, float scale, float* y) {
    c2r(Y, (long long)U * Tn, frames, scale / NFFT, 1);
    launch_ola(frames, WIN, Tn, WIN, HOP, inv_env, y, U, Ls, Q, nullptr, nullptr, nullptr, st);
  }
  // X = sx * STFT(istft(Y) (+ add_scale * add)) without the signal in between: the overlap-add happens in the load of the second transform.
  // The frames buffer stays valid afterwards (stft_of_frames below repeats the second half on it, e.g. with and without the noise term).
  void istft_stft(const float* Y, int Tn, int Q, const float* inv_env, int Ls, float si, int P, int Tx, float sx, float* X, const float* add = nullptr,
                  float add_scale = 0.f, const float* add_scale_dev = nullptr) {
    c2r(Y, (long long)U * Tn, frames, si / NFFT, 1);
    stft_of_frames(Tn, Q, inv_env, Ls, P, Tx, sx, X, add, add_scale, add_scale_dev);
  }
  void stft_of_frames(int Tn, int Q, const float* inv_env, int Ls, int P, int Tx, float sx, float* X, const float* add = nullptr, float add_scale = 0.f,
                      const float* add_scale_dev = nullptr) {
    r2c(R2cSrc{nullptr, frames, Ls, P, Tn, Q, inv_env, nullptr, add, add_scale, add_scale_dev}, Tx, X, sx, 0);
  }
  // adjoint of stft: g_s from G_X
  void stft_adj(const float* GX, int Ls, int P, int Tn, float scale, float* gs) {
    c2r(GX, (long long)U * Tn, frames, scale, 0);
    launch_ola(frames, WIN, Tn, WIN, HOP, ones, gs, U, Ls, P, nullptr, nullptr, nullptr, st);
  }
  // adjoint of istft: G_Y from g_y
  void istft_adj(const float* gy, int Tn, int Q, const float* inv_env, int Ls, float scale, float* GY) {
    r2c(R2cSrc{gy, nullptr, Ls, Q, 0, 0, nullptr, inv_env, nullptr, 0.f, nullptr}, Tn, GY, scale / NFFT, 1);
  }
  // G_Y = istft_adj(stft_adj(G_X)) without the signal in between (both adjoints act on the same Ls samples)
  void stft_adj_istft_adj(const float* GX, int Ls, int P, int Tx, float sx, int Tn, int Q, const float* inv_env, float si, float* GY) {
    c2r(GX, (long long)U * Tx, frames, sx, 0);
    r2c(R2cSrc{nullptr, frames, Ls, Q, Tx, P, ones, inv_env, nullptr, 0.f, nullptr}, Tn, GY, si / NFFT, 1);
  }
  // first stage (form I) of a transform of a REAL signal of Lr_ samples zero-padded to N2 (zero0: sample 0 forced to zero) -> c2
  void fft_first(const S1In& in, bool zero0, int sign) {
    const dim3 g1(F2 / S1_COLS, U);
    if (zero0) hipLaunchKernelGGL(fft_stage1_kernel<true>, g1, dim3(S1_THREADS), 0, st, in, c2, (const float2*)w101, (const float2*)twN, sign);
    else hipLaunchKernelGGL(fft_stage1_kernel<false>, g1, dim3(S1_THREADS), 0, st, in, c2, (const float2*)w101, (const float2*)twN, sign);
  }
  template <int SGN, int PW> void fft_mid256(float scale) {           // c2 -> c1
    const MpArrays a{Hf, Mabs, phim, gM};
    hipLaunchKernelGGL((fft_mid256_kernel<SGN, PW>), dim3(cdiv(F1, S2_COLS), U), dim3(S2_THREADS), 0, st, (const float2*)c2, c1, (const float2*)w256, (const float2*)twN, scale, a);
  }
  void fft_mid101() {                                                  // c1 -> c2
    hipLaunchKernelGGL(fft_mid101_kernel, dim3(F2 / S1_COLS, U), dim3(S1_THREADS), 0, st, (const float2*)c1, c2, (const float2*)w101, (const float2*)twN);
  }
  void fft_last_real(float* out, int Lo, int sign, float scale, bool first_set, float first) {       // c1 -> out
    hipLaunchKernelGGL(fft_stageB_real_kernel, dim3(F2 / S1_COLS, U), dim3(S1_THREADS), 0, st, (const float2*)c1, out, Lo, (const float2*)w101, sign, scale, first_set ? 1 : 0, first);
  }
  DesignTabs tabs() const { DesignTabs t; t.idx = idx; t.frac = frac; t.corr = corr; t.dpm = dpm; t.fge = fge; return t; }

  void design() {              // the filter magnitudes alone (buddy_blindop_design_filter)
    hipLaunchKernelGGL(design_dm_kernel, dim3(cdiv(U * Nf * K, 256)), dim3(256), 0, st, (const float*)decay, (const float*)wts, logdm, dmv, U, E, NB, Nf);
    hipLaunchKernelGGL(design_A_kernel, dim3(gridf((long long)U * Nf * FB)), dim3(256), 0, st, (const float*)logdm, tabs(), A, Apre, U, K, Nf);
  }
  // minimum_phase_version (reference reverb_utils.py:9-23) of hin (U, <= Lm samples, zero-padded to N2): out[u][0..Lo) = its real part
  // (sample 0 := first when first_set).  Leaves Hf = FFT([hin, 0]), Mabs = |Hf| and phim for the backward pass.
  void minphase_core(const S1In& hin, float* out, int Lo, bool first_set = false, float first = 0.f) {
    fft_first(hin, false, -1);                       // FFT([hin, zeros]) ...
    fft_mid256<-1, 0>(1.f);                          // ... | log |.| | FFT ...
    fft_mid101();                                    // ... | Hilbert window | IFFT ...
    fft_mid256<1, 1>(1.f / N2);                      // ... | M exp(-j Im .) | IFFT ...
    fft_last_real(out, Lo, +1, 1.f / N2, first_set, first);
  }
  // A = design(decay, weights) and H = cons(A * exp(j phi))   (reference :212-251, :333-351)
  void cons_forward() {
    hipLaunchKernelGGL(design_row_kernel, dim3(U * Nf), dim3(256), 0, st, (const float*)decay, (const float*)wts, tabs(), (const float*)phi, logdm, dmv, A, Apre, Fin, U, E, NB,
                       Nf, fused_loop ? d_step : (int*)nullptr);
    // h0 = istft(Fin) is never materialised: the first transform of the projection gathers it from the synthesis frames
    c2r(Fin, (long long)U * (Nf + 2), frames, 1.f / NFFT, 1);
    minphase_core(S1In{nullptr, Lh, frames, Nf + 2, WIN, env_c}, hm, Lm, true, (float)(WIN / (HOP * 2.0)));
    stft(hm, Lm, WIN - HOP, Nf, 1.f, H);          // frames 1..Nf of the centred STFT: frame k starts at 128 (k+1) - 512
  }
  // G_Fin from G_H
  void cons_backward(const float* GHin) {
    c2r(GHin, (long long)U * Nf, frames, 1.f, 0);                              // ghm = stft_adj(G_H), gathered from the frames by the first transform
    fft_first(S1In{nullptr, Lm, frames, Nf, WIN - HOP, ones}, true, -1);       // GZ = FFT([0, ghm[1:], zeros]) / N2 ...
    fft_mid256<-1, 2>(1.f / N2);                     // ... | gM, g_phi | FFT ...
    fft_mid101();                                    // ... | Hilbert window | IFFT ...
    fft_mid256<1, 3>(1.f / N2);                      // ... | G_H | N2 * IFFT
    fft_last_real(gh0, Lh, +1, 1.f, false, 0.f);     // g_h0 = Re(N2 * IFFT(GH))[:Lh]
    istft_adj(gh0, Nf + 2, WIN, env_c, Lh, 1.f, GFin);
  }
  void update_H() { cons_forward(); }
  bool fir_lds_ok() const {
    const bool lds = cur_opt().fir_lds != 0;
    return lds && big_lds && Nf == FIR_NF;
  }
  // Y0 = FIR(X0, H) and, when X1b is given, Y1 = FIR(X1b, H) in the same launch (LDS kernel only)
  void fir2(const float* X0, long long xs0, int T0, float* Y0, const float* X1b, long long xs1, int T1, float* Y1) {
    FirSegs sg; std::memset(&sg, 0, sizeof(sg));
    sg.s[0] = FirSeg{X0, xs0, Y0, T0, (T0 + FL_TB - 1) / FL_TB};
    if (X1b) sg.s[1] = FirSeg{X1b, xs1, Y1, T1, (T1 + FL_TB - 1) / FL_TB};
    const size_t sm = (size_t)(FL_TB + 2 * Nf - 1) * FL_BINS * sizeof(float2);
    hipLaunchKernelGGL(fir_sb_lds_kernel<FIR_NF>, dim3(sg.s[0].tiles + sg.s[1].tiles, (FB + FL_BINS - 1) / FL_BINS, U), dim3(256), sm, st, sg, (const float*)H);
  }
  void fir(const float* X, long long xs, int Tn, float* Y) {
    if (fir_lds_ok()) fir2(X, xs, Tn, Y, nullptr, 0, 0, nullptr);
    else hipLaunchKernelGGL(fir_kernel_sb, dim3(gridf((long long)U * ((Tn + FT - 1) / FT) * FB)), dim3(256), 0, st, X, xs, (const float*)H, Y, U, Tn, Nf);
  }
  // GH (+)= gradient of the taps from (X0, GY0) [+ (X1b, GY1)]
  void gradh2(const float* X0, long long xs0, const float* GY0, int T0, const float* X1b, long long xs1, const float* GY1, int T1, int accumulate) {
    GradSegs sg; std::memset(&sg, 0, sizeof(sg));
    sg.s[0] = GradSeg{X0, xs0, GY0, T0}; sg.n = 1;
    if (X1b) { sg.s[1] = GradSeg{X1b, xs1, GY1, T1}; sg.n = 2; }
    hipLaunchKernelGGL(fir_gradh_lds_kernel, dim3((FB + FL_BINS - 1) / FL_BINS, U, (Nf + GL_TAPS - 1) / GL_TAPS), dim3(256), 0, st, sg, GH, Nf, accumulate);
  }
  void gradh(const float* X, long long xs, const float* GY, int Tn, int accumulate) {
    const bool lds = cur_opt().fir_lds != 0;
    if (lds) gradh2(X, xs, GY, Tn, nullptr, 0, nullptr, 0, accumulate);
    else
      hipLaunchKernelGGL(fir_gradh_kernel, dim3(U * ((Nf + FT - 1) / FT) * ((FB + GH_F - 1) / GH_F)), dim3(256), 0, st, X, xs, GY, GH, U, Tn, Nf, accumulate);
  }
  // loss_u (+)= kappa * sum |Rc - comp(Xh)|^2, G optional; a second term (Rc1 == nullptr: target = compress(G1 on entry)) may share the launch
  static constexpr int LOSS_BLK = 64, LOSS_BLK_B = 32;
  void comp_loss2(const float* Rc0, const float* Xh0, float* G0, int T0, float w0, float* out0, const float* Rc1, const float* Xh1, float* G1, int T1, float w1,
                  float* out1, bool two) {
    LossJobs jobs; std::memset(&jobs, 0, sizeof(jobs));
    // normalisation of the loss family (utils/losses.py:46-64): summean = mean over frames of the sum over bins (the shipped one), sum, mean over both
    auto kap = [&](float w, int Tn) { return loss_norm == 0 ? w / (float)Tn : (loss_norm == 1 ? w : w / ((float)Tn * (float)FB)); };
    const float k0 = kap(w0, T0), k1 = two ? kap(w1, T1) : 0.f;
    jobs.j[0] = LossJob{Rc0, Xh0, G0, partial, T0, k0, LOSS_BLK};
    if (two) jobs.j[1] = LossJob{Rc1, Xh1, G1, partial_b, T1, k1, LOSS_BLK_B};
    hipLaunchKernelGGL(comp_loss_kernel, dim3(LOSS_BLK + (two ? LOSS_BLK_B : 0), U), dim3(256), 0, st, jobs, c.comp);
    if (!fused_loop) {       // the loop only needs the gradient; the loss VALUES are read through buddy_blindop_param_grads / rec_loss_grad
      hipLaunchKernelGGL(loss_finalize_kernel, dim3(U), dim3(32), 0, st, (const double*)partial, LOSS_BLK, k0, out0, 0);
      if (two) hipLaunchKernelGGL(loss_finalize_kernel, dim3(U), dim3(32), 0, st, (const double*)partial_b, LOSS_BLK_B, k1, out1, 0);
    }
  }
  void comp_loss(const float* Rcx, const float* Xh, float* G, int Tn, float weight, float* out, int accumulate) {
    (void)accumulate;
    comp_loss2(Rcx, Xh, G, Tn, weight, out, nullptr, nullptr, nullptr, 0, 0.f, nullptr, false);
  }
  // RIR-noise regulariser (reference :94-100): loss(rir, (rir + t n).detach()), gradient w.r.t. the subband filter's output left in X2
  void reg_chain(const float* noise, float t_op, const float* t_op_dev, float w_reg) {
    fir(Xdelta, 0, Td, Ybuf);                                                                       // rir = istft(FIR(Xdelta, H)), never materialised:
    istft_stft(Ybuf, Td, WIN + WIN / 2, env_d, Lr, norm, WIN, Td, 1.f / norm, X3, noise, t_op, t_op_dev);   // STFT(rir + t n)
    hipLaunchKernelGGL(compress_kernel, dim3(gridf((long long)U * Td * FB)), dim3(256), 0, st, (const float*)X3, Rc, (long long)U * Td, c.comp);
    stft_of_frames(Td, WIN + WIN / 2, env_d, Lr, WIN, Td, 1.f / norm, X2);                           // STFT(rir) from the same frames
    comp_loss(Rc, X2, X3, Td, w_reg, losses + U, 0);
    stft_adj_istft_adj(X3, Lr, WIN, Td, 1.f / norm, Td, WIN + WIN / 2, env_d, norm, X2);
  }
  void degrade(const float* x, float* y) {
    stft(x, L, WIN, T, 1.f / norm, X1);
    fir(X1, (long long)T * LDSP, T, Ybuf);
    istft(Ybuf, T, WIN + WIN / 2, env_T, L, norm, y);
  }
  void time_rir(float* out) {
    fir(Xdelta, 0, Td, Ybuf);
    istft(Ybuf, Td, WIN + WIN / 2, env_d, Lr, norm, out);
  }
};

static void host_window(std::vector<float>& w, double& norm2) {          // hann(512) (periodic) and sum of squares (the STFT's 1/sqrt(sum w^2) norm)
  w.resize(WIN);
  norm2 = 0;
  for (int n = 0; n < WIN; ++n) { const double v = 0.5 - 0.5 * std::cos(2.0 * PI * n / WIN); w[n] = (float)v; norm2 += (double)(float)v * (double)(float)v; }
}
static std::vector<float> inv_env_of(const std::vector<float>& w, int Tn) {
  std::vector<double> e((size_t)(Tn - 1) * HOP + NFFT, 0.0);
  for (int t = 0; t < Tn; ++t) for (int n = 0; n < WIN; ++n) e[(size_t)t * HOP + n] += (double)w[n] * (double)w[n];
  std::vector<float> r(e.size());
  for (size_t i = 0; i < e.size(); ++i) r[i] = e[i] > 1e-11 ? (float)(1.0 / (double)(float)e[i]) : 0.f;
  return r;
}
#define UP(dst, vec) do { if (o->dalloc(&o->dst, (vec).size())) { set_error("hipMalloc failed"); delete o; return BUDDY_ERR_HIP; } \
  if (hipMemcpy(o->dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy failed"); delete o; return BUDDY_ERR_HIP; } } while (0)
#define DA(dst, n) do { if (o->dalloc(&o->dst, (size_t)(n))) { set_error("hipMalloc failed"); delete o; return BUDDY_ERR_HIP; } } while (0)

int blindop_create(const BlindOpCfg& cfg, int U, int L, BlindOp** out) {
  if (cfg.n_fft != NFFT || cfg.win != WIN || cfg.hop != HOP) { set_error("operator STFT must be 1024/512/128"); return BUDDY_ERR_ARG; }
  if (cfg.num_knots < 3 || cfg.num_knots > 64 || cfg.Nf < 4 || cfg.E < 1) { set_error("bad operator config"); return BUDDY_ERR_ARG; }
  BlindOp* o = new BlindOp();
  o->c = cfg; o->U = U; o->L = L; o->Nf = cfg.Nf; o->E = cfg.E; o->K = cfg.num_knots; o->NB = cfg.num_knots - 2;
  o->T = 1 + (L + WIN) / HOP;
  o->Lh = HOP * cfg.Nf; o->Lm = o->Lh + HOP; o->Lr = o->Lh + 1024; o->Td = 1 + (o->Lr + WIN) / HOP;
  if (2 * o->Lm != N2) { set_error("Nf must be 100 (25856-point minimum-phase FFT)"); delete o; return BUDDY_ERR_ARG; }
  std::vector<float> w; double norm2;
  host_window(w, norm2);
  o->norm = (float)std::sqrt((double)(float)norm2);
  std::vector<float> eT = inv_env_of(w, o->T), ed = inv_env_of(w, o->Td), ec = inv_env_of(w, cfg.Nf + 2);
  UP(env_T, eT); UP(env_d, ed); UP(env_c, ec);
  std::vector<float> ones((size_t)((o->T > o->Td ? o->T : o->Td) + 8) * HOP + NFFT, 1.f); UP(ones, ones);
  // interpolation tables (torchcde LinearInterpolation semantics: bucketize(q, knots) - 1, clamped)
  std::vector<int> idx(FB); std::vector<float> frac(FB);
  for (int f = 0; f < FB; ++f) {
    const float q = (float)f * (float)cfg.sample_rate / (float)NFFT;      // rfftfreq
    int b = 0;
    while (b < o->K && cfg.knots[b] < q) ++b;                              // bucketize (right = False): first index with knot >= q
    int i0 = b - 1; if (i0 < 0) i0 = 0; if (i0 > o->K - 2) i0 = o->K - 2;
    idx[f] = i0; frac[f] = (q - cfg.knots[i0]) / (cfg.knots[i0 + 1] - cfg.knots[i0]);
  }
  UP(idx, idx); UP(frac, frac);
  std::vector<int> fge((size_t)o->K + 3);                                    // fge[v + 1] = first bin with idx >= v, v = -1 .. K + 1 (idx is non-decreasing)
  for (int v = -1; v <= o->K + 1; ++v) { int f = 0; while (f < FB && idx[f] < v) ++f; fge[v + 1] = f; }
  UP(fge, fge);
  std::vector<float> corr(cfg.Nf, 1.f);
  { const int Kc = WIN / HOP - 1; double ws = 0; for (int n = 0; n < WIN; ++n) ws += w[n];
    for (int k = 0; k < Kc; ++k) { double s = 0; for (int n = (Kc - k) * HOP; n < WIN; ++n) s += w[n]; corr[k] = (float)((float)ws / (float)s); } }
  UP(corr, corr);
  // FFT twiddles (double precision tables)
  std::vector<float2> w101(F1), w256(F2), tw((size_t)F2 * F1);
  for (int i = 0; i < F1; ++i) w101[i] = make_float2((float)std::cos(2 * PI * i / F1), (float)(std::sin(2 * PI * i / F1)));
  for (int i = 0; i < F2; ++i) w256[i] = make_float2((float)std::cos(2 * PI * i / F2), (float)(std::sin(2 * PI * i / F2)));
  for (int n2 = 0; n2 < F2; ++n2) for (int k1 = 0; k1 < F1; ++k1) {
    const double a = 2 * PI * (double)((long long)n2 * k1) / N2;
    tw[(size_t)n2 * F1 + k1] = make_float2((float)std::cos(a), (float)(std::sin(a)));
  }
  UP(w101, w101); UP(w256, w256); UP(twN, tw);
  { std::vector<float2> wr(64 * 16), w64(16 * 4);     // per-lane twiddle rows of the 1024-point kernels: W1024^(r b) and W64^(r0 c0) = W1024^(16 r0 c0)
    auto W1k = [](int i) { i &= 1023; return make_float2((float)std::cos(2 * PI * i / 1024), (float)std::sin(2 * PI * i / 1024)); };
    for (int r = 0; r < 64; ++r) for (int b = 0; b < 16; ++b) wr[r * 16 + b] = W1k(r * b);
    for (int r0 = 0; r0 < 16; ++r0) for (int c0 = 0; c0 < 4; ++c0) w64[r0 * 4 + c0] = W1k(16 * r0 * c0);
    UP(w1024r, wr); UP(w64t, w64); UP(win, w); }
  const int U_ = U, Nf = cfg.Nf, Td = o->Td;
  const int T = o->T > Td ? o->T : Td;            // work buffers hold either the signal (T frames) or the time-RIR (Td frames)
  const int Lmax = L > o->Lr ? L : o->Lr;
  const size_t specT = (size_t)U_ * T * LDSP + 8, specH = (size_t)U_ * (Nf + 2) * LDSP + 8;
  DA(decay, U_ * o->E * o->NB); DA(wts, U_ * o->E * o->NB); DA(phi, (size_t)U_ * Nf * FB);
  DA(m_d, U_ * o->E * o->NB); DA(v_d, U_ * o->E * o->NB); DA(m_w, U_ * o->E * o->NB); DA(v_w, U_ * o->E * o->NB);
  DA(m_p, (size_t)U_ * Nf * FB); DA(v_p, (size_t)U_ * Nf * FB);
  DA(H, specH); DA(Yc, specT); DA(Xdelta, (size_t)Td * LDSP + 8);
  DA(frames, (size_t)U_ * (T + 2) * WIN);
  DA(X1, specT); DA(X2, specT); DA(X3, specT); DA(Ybuf, specT); DA(sig1, (size_t)U_ * (Lmax + 8)); DA(sig2, (size_t)U_ * (Lmax + 8));
  DA(A, (size_t)U_ * Nf * FB); DA(Apre, (size_t)U_ * Nf * FB); DA(logdm, U_ * Nf * o->K); DA(dmv, U_ * Nf * o->K); DA(gdm, U_ * Nf * o->K);
  DA(Fin, specH); DA(GFin, specH); DA(GH, specH); DA(gphi, (size_t)U_ * Nf * FB);
  DA(gdecay, U_ * o->E * o->NB); DA(gw, U_ * o->E * o->NB);
  DA(hm, (size_t)U_ * o->Lm); DA(gh0, (size_t)U_ * o->Lh);
  DA(c1, (size_t)U_ * N2); DA(c2, (size_t)U_ * N2); DA(Hf, (size_t)U_ * N2);
  DA(Mabs, (size_t)U_ * N2); DA(phim, (size_t)U_ * N2); DA(gM, (size_t)U_ * N2);
  DA(partial, (size_t)U_ * 64); DA(losses, (size_t)U_ * 4);
  o->big_lds = hipFuncSetAttribute((const void*)fir_sb_lds_kernel<FIR_NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
  {
    const size_t specD = (size_t)U_ * Td * LDSP + 8;
    DA(frames_b, (size_t)U_ * (Td + 2) * WIN);
    DA(X2_b, specD); DA(X3_b, specD); DA(Ybuf_b, specD); DA(partial_b, (size_t)U_ * 64);
  }
  DA(rir, (size_t)U_ * o->Lr); DA(Rc, (size_t)U_ * Td * LDSP + 8);
  DA(dpm, (size_t)Nf * FB);
  DA(d_step, 4); DA(d_scal, 4); DA(bc_tab, BlindOp::MAXSTEP); DA(xden_buf, (size_t)U_ * L);
  // direct-path magnitude correction |STFT(2 delta)|[:, 1:] (reference :201-205) and STFT of the unit impulse, computed on device
  {
    const int saveU = o->U; o->U = 1; o->st = nullptr;
    std::vector<float> imp((size_t)o->Lr, 0.f);
    imp[0] = (float)(WIN / (HOP * 2.0));
    float* tmp_sig = o->rir;              // reuse as staging (Lr floats per utterance)
    HIPCHK(hipMemcpy(tmp_sig, imp.data(), (size_t)o->Lh * 4, hipMemcpyHostToDevice));
    // raw centred STFT of h (length Lh): frame t starts at 128 t - 512; columns 1..Nf
    o->stft(tmp_sig, o->Lh, WIN - HOP, Nf, 1.f, o->GH);
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> hs((size_t)Nf * LDSP);
    HIPCHK(hipMemcpy(hs.data(), o->GH, hs.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> dpm((size_t)Nf * FB);
    for (int k = 0; k < Nf; ++k) for (int f = 0; f < FB; ++f) {
      const float re = hs[(size_t)k * LDSP + 2 * f], im = hs[(size_t)k * LDSP + 2 * f + 1];
      dpm[(size_t)k * FB + f] = std::sqrt(re * re + im * im);
    }
    HIPCHK(hipMemcpy(o->dpm, dpm.data(), dpm.size() * 4, hipMemcpyHostToDevice));
    imp[0] = 1.f;
    HIPCHK(hipMemcpy(tmp_sig, imp.data(), (size_t)o->Lr * 4, hipMemcpyHostToDevice));
    o->stft(tmp_sig, o->Lr, WIN, Td, 1.f / o->norm, o->Xdelta);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(o->GH, 0, specH * 4));
    o->U = saveU;
  }
  *out = o;
  return BUDDY_OK;
}
void blindop_destroy(BlindOp* o) { delete o; }

int blindop_set_params(BlindOp* o, const float* decay, const float* wts, const float* phases_ref, int reset_adam, hipStream_t st) {
  o->st = st;
  const size_t nb = (size_t)o->U * o->E * o->NB * 4;
  if (decay) HIPCHK(hipMemcpyAsync(o->decay, decay, nb, hipMemcpyDeviceToDevice, st));
  if (wts) HIPCHK(hipMemcpyAsync(o->wts, wts, nb, hipMemcpyDeviceToDevice, st));
  if (phases_ref) hipLaunchKernelGGL(transpose_fk_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, phases_ref, o->phi, o->U, o->Nf, 1);
  if (reset_adam) {
    o->adam_step = 0;
    HIPCHK(hipMemsetAsync(o->d_step, 0, 4, st));
    HIPCHK(hipMemsetAsync(o->m_d, 0, nb, st)); HIPCHK(hipMemsetAsync(o->v_d, 0, nb, st));
    HIPCHK(hipMemsetAsync(o->m_w, 0, nb, st)); HIPCHK(hipMemsetAsync(o->v_w, 0, nb, st));
    const size_t np = (size_t)o->U * o->Nf * FB * 4;
    HIPCHK(hipMemsetAsync(o->m_p, 0, np, st)); HIPCHK(hipMemsetAsync(o->v_p, 0, np, st));
  }
  return BUDDY_OK;
}
int blindop_get_params(BlindOp* o, float* decay, float* wts, float* phases_ref, hipStream_t st) {
  o->st = st;
  const size_t nb = (size_t)o->U * o->E * o->NB * 4;
  if (decay) HIPCHK(hipMemcpyAsync(decay, o->decay, nb, hipMemcpyDeviceToDevice, st));
  if (wts) HIPCHK(hipMemcpyAsync(wts, o->wts, nb, hipMemcpyDeviceToDevice, st));
  if (phases_ref) hipLaunchKernelGGL(transpose_fk_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->phi, phases_ref, o->U, o->Nf, 0);
  return BUDDY_OK;
}
// H from the current parameters; with noise (U, Nf*128 samples): phases := angle(STFT(noise)/norm)[:, 1:], then phases := angle(H)
int blindop_update_H(BlindOp* o, const float* noise, hipStream_t st) {
  o->st = st;
  if (noise) {
    o->stft(noise, o->Lh, WIN, o->Nf + 1, 1.f / o->norm, o->X1);     // centred frames 0..Nf; we need 1..Nf
    hipLaunchKernelGGL(unit_phase_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->X1, o->phi, o->U, o->Nf);
  }
  o->update_H();
  if (noise) hipLaunchKernelGGL(angle_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->H, o->phi, o->U, o->Nf);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
int blindop_get_H(BlindOp* o, float* out, hipStream_t st) {
  hipLaunchKernelGGL(copy_h_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->H, out, o->U, o->Nf);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
int blindop_set_y(BlindOp* o, const float* y, hipStream_t st) {
  o->st = st;
  o->stft(y, o->L, WIN, o->T, 1.f / o->norm, o->X1);
  hipLaunchKernelGGL(compress_kernel, dim3(gridf((long long)o->U * o->T * FB)), dim3(256), 0, st, (const float*)o->X1, o->Yc, (long long)o->U * o->T, o->c.comp);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
int blindop_degrade(BlindOp* o, const float* x, float* y, hipStream_t st) { o->st = st; o->degrade(x, y); HIPCHK(hipGetLastError()); return BUDDY_OK; }
int blindop_time_rir(BlindOp* o, float* out, hipStream_t st) { o->st = st; o->time_rir(out); HIPCHK(hipGetLastError()); return BUDDY_OK; }

// ---- per-function views for the parity tests (each is the piece optimize_op / the likelihood call internally) ----
// design_filter (:241-251) from the current decay / weights: A in the reference layout (U, 513, Nf)
int blindop_design_filter(BlindOp* o, float* A_ref, hipStream_t st) {
  o->st = st;
  o->design();
  hipLaunchKernelGGL(transpose_fk_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->A, A_ref, o->U, o->Nf, 0);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// apply_stft (:41-52) of x (U, L): (U, 513, T, 2), T = 1 + (L + 512) / 128
int blindop_apply_stft(BlindOp* o, const float* x, float* X_ref, hipStream_t st) {
  o->st = st;
  o->stft(x, o->L, WIN, o->T, 1.f / o->norm, o->X2);
  hipLaunchKernelGGL(copy_spec_kernel, dim3(gridf((long long)o->U * o->T * FB)), dim3(256), 0, st, (const float*)o->X2, X_ref, o->U, o->T);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// minimum_phase_version (reverb_utils.py:9-23) of h (U, 128 * (Nf + 1)) -- the size cons() uses; no direct-path override
int blindop_minphase(BlindOp* o, const float* h, float* out, hipStream_t st) {
  o->st = st;
  o->minphase_core(S1In{h, o->Lm, nullptr, 0, 0, nullptr}, out, o->Lm);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// project_params (:298-331) on the current decay / weights
int blindop_project(BlindOp* o, hipStream_t st) {
  o->st = st;
  hipLaunchKernelGGL(project_kernel, dim3(cdiv(o->U * o->NB, 256)), dim3(256), 0, st, o->decay, o->wts, o->U, o->E, o->NB, o->c.min_decay, o->c.max_decay,
                     o->c.w_lo, o->c.w_hi, o->c.clamp_decay, o->c.long2nd);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// Adam moments (torch Adam (the optimizer the reference constructs, EulerHeunSamplerDPS.py:193) exp_avg / exp_avg_sq) in the reference layouts; any pointer may be NULL
int blindop_get_adam(BlindOp* o, float* m_decay, float* v_decay, float* m_wts, float* v_wts, float* m_phases, float* v_phases, int* step, hipStream_t st) {
  o->st = st;
  const size_t nb = (size_t)o->U * o->E * o->NB * 4;
  if (m_decay) HIPCHK(hipMemcpyAsync(m_decay, o->m_d, nb, hipMemcpyDeviceToDevice, st));
  if (v_decay) HIPCHK(hipMemcpyAsync(v_decay, o->v_d, nb, hipMemcpyDeviceToDevice, st));
  if (m_wts) HIPCHK(hipMemcpyAsync(m_wts, o->m_w, nb, hipMemcpyDeviceToDevice, st));
  if (v_wts) HIPCHK(hipMemcpyAsync(v_wts, o->v_w, nb, hipMemcpyDeviceToDevice, st));
  const dim3 g(gridf((long long)o->U * o->Nf * FB));
  if (m_phases) hipLaunchKernelGGL(transpose_fk_kernel, g, dim3(256), 0, st, (const float*)o->m_p, m_phases, o->U, o->Nf, 0);
  if (v_phases) hipLaunchKernelGGL(transpose_fk_kernel, g, dim3(256), 0, st, (const float*)o->v_p, v_phases, o->U, o->Nf, 0);
  if (step) *step = o->adam_step;
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

// ---- the differentiable surface the REFERENCE's own sampler needs (round 6): it autograds through operator.degradation (testing/EulerHeunSamplerDPS.py:61-69),
// through update_H / get_time_RIR inside optimize_op (:71-113) and through get_loss(...)(y, y_hat) (utils/losses.py:26-71).  Each entry below is the
// vector-Jacobian product of one of those pieces, built from the adjoint kernels the fused likelihood / optimisation calls already use; the Python classes
// wrap them as torch.autograd.Function, so an unmodified torch-loop sampler runs on the HIP operator.
static int frames_of(const BlindOp* o, int len) { return len == o->L ? o->T : (len == o->Lr ? o->Td : -1); }
// x (U, L), g_y (U, L) -> g_x = (d degrade / d x)^T g_y and / or g_H (U, 513, Nf, 2) = (d degrade / d H)^T g_y, for the CURRENT H
int blindop_degrade_vjp(BlindOp* o, const float* x, const float* g_y, float* g_x, float* g_H_ref, hipStream_t st) {
  o->st = st;
  const int U = o->U, T = o->T, L = o->L;
  o->istft_adj(g_y, T, WIN + WIN / 2, o->env_T, L, o->norm, o->X2);                         // gradient w.r.t. the filtered spectrogram
  if (g_H_ref) {
    o->stft(x, L, WIN, T, 1.f / o->norm, o->X1);
    o->gradh(o->X1, (long long)T * LDSP, o->X2, T, 0);
    hipLaunchKernelGGL(copy_h_kernel, dim3(gridf((long long)U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->GH, g_H_ref, U, o->Nf);
  }
  if (g_x) {
    hipLaunchKernelGGL(fir_adjx_kernel, dim3(gridf((long long)U * T * FB)), dim3(256), 0, st, (const float*)o->X2, (const float*)o->H, o->X3, U, T, o->Nf);
    o->stft_adj(o->X3, L, WIN, T, 1.f / o->norm, g_x);
  }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// g_rir (U, Lr) -> g_H (U, 513, Nf, 2) = (d get_time_RIR / d H)^T g_rir
int blindop_time_rir_vjp(BlindOp* o, const float* g_rir, float* g_H_ref, hipStream_t st) {
  o->st = st;
  o->istft_adj(g_rir, o->Td, WIN + WIN / 2, o->env_d, o->Lr, o->norm, o->X2);
  o->gradh(o->Xdelta, 0LL, o->X2, o->Td, 0);
  hipLaunchKernelGGL(copy_h_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->GH, g_H_ref, o->U, o->Nf);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// g_H (U, 513, Nf, 2) -> gradients of (decay, weights, phases) through cons(design_filter(.) exp(j phases)): the state of the LAST update_H is used
// (A, the minimum-phase spectra); any output may be NULL
int blindop_update_H_vjp(BlindOp* o, const float* g_H_ref, float* g_decay, float* g_wts, float* g_phases_ref, hipStream_t st) {
  o->st = st;
  const int U = o->U, Nf = o->Nf;
  hipLaunchKernelGGL(spec_from_ref_kernel, dim3(gridf((long long)U * Nf * (LDSP / 2))), dim3(256), 0, st, g_H_ref, o->GH, U, Nf);
  o->cons_backward(o->GH);
  hipLaunchKernelGGL(h0_bwd_knots_kernel, dim3(U * Nf), dim3(256), 0, st, (const float*)o->GFin, (const float*)o->A, (const float*)o->Apre, (const float*)o->phi, o->tabs(),
                     (const float*)o->dmv, o->gphi, o->gdm, U, o->K, Nf);
  hipLaunchKernelGGL(design_bwd_params_kernel, dim3(cdiv(U * o->E * o->NB, 4)), dim3(256), 0, st, (const float*)o->gdm, (const float*)o->decay, (const float*)o->wts, o->gdecay, o->gw, U, o->E, o->NB, Nf);
  const size_t nb = (size_t)U * o->E * o->NB * 4;
  if (g_decay) HIPCHK(hipMemcpyAsync(g_decay, o->gdecay, nb, hipMemcpyDeviceToDevice, st));
  if (g_wts) HIPCHK(hipMemcpyAsync(g_wts, o->gw, nb, hipMemcpyDeviceToDevice, st));
  if (g_phases_ref) hipLaunchKernelGGL(transpose_fk_kernel, dim3(gridf((long long)U * Nf * FB)), dim3(256), 0, st, (const float*)o->gphi, g_phases_ref, U, Nf, 0);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// apply_stft (:41-52) of signals of the bound length L (T frames) or of the time-RIR length Lr (Td frames), and its adjoint
int blindop_stft_len(BlindOp* o, const float* x, int len, float* X_ref, hipStream_t st) {
  o->st = st;
  const int Tn = frames_of(o, len);
  if (Tn < 0) { set_error("apply_stft: the handle transforms signals of its bound length or of its time-RIR length only"); return BUDDY_ERR_ARG; }
  o->stft(x, len, WIN, Tn, 1.f / o->norm, o->X2);
  hipLaunchKernelGGL(copy_spec_kernel, dim3(gridf((long long)o->U * Tn * FB)), dim3(256), 0, st, (const float*)o->X2, X_ref, o->U, Tn);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
int blindop_stft_len_adj(BlindOp* o, const float* G_ref, int len, float* g_x, hipStream_t st) {
  o->st = st;
  const int Tn = frames_of(o, len);
  if (Tn < 0) { set_error("apply_stft adjoint: the handle transforms signals of its bound length or of its time-RIR length only"); return BUDDY_ERR_ARG; }
  hipLaunchKernelGGL(spec_from_ref_kernel, dim3(gridf((long long)o->U * Tn * (LDSP / 2))), dim3(256), 0, st, G_ref, o->X3, o->U, Tn);
  o->stft_adj(o->X3, len, WIN, Tn, 1.f / o->norm, g_x);
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// loss_u = weight * l2_comp_stft_summean(a_u, b_u) (utils/losses.py:59-64) of two signals of length len in {L, Lr}, with the gradient w.r.t. either
// argument (NULL: not wanted).  The formula is symmetric in (a, b); each gradient is one pass of the loss kernel against the other side's compressed spectrum.
int blindop_stft_loss(BlindOp* o, const float* a, const float* b, int len, float weight, float* loss, float* g_a, float* g_b, hipStream_t st) {
  o->st = st;
  const int U = o->U, Tn = frames_of(o, len);
  if (Tn < 0) { set_error("stft loss: signals of the handle's bound length or of its time-RIR length only"); return BUDDY_ERR_ARG; }
  const dim3 gc(gridf((long long)U * Tn * FB));
  o->stft(a, len, WIN, Tn, 1.f / o->norm, o->X1);
  o->stft(b, len, WIN, Tn, 1.f / o->norm, o->X2);
  if (g_b || !g_a) {
    hipLaunchKernelGGL(compress_kernel, gc, dim3(256), 0, st, (const float*)o->X1, o->Ybuf, (long long)U * Tn, o->c.comp);
    o->comp_loss(o->Ybuf, o->X2, g_b ? o->X3 : nullptr, Tn, weight, loss, 0);
    if (g_b) o->stft_adj(o->X3, len, WIN, Tn, 1.f / o->norm, g_b);
  }
  if (g_a) {
    hipLaunchKernelGGL(compress_kernel, gc, dim3(256), 0, st, (const float*)o->X2, o->Ybuf, (long long)U * Tn, o->c.comp);
    o->comp_loss(o->Ybuf, o->X1, o->X3, Tn, weight, loss, 0);
    o->stft_adj(o->X3, len, WIN, Tn, 1.f / o->norm, g_a);
  }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}
// the compression exponent of the spectral losses, (0, 1]; the cached compressed observation must be set again afterwards (buddy_blindop_set_y)
int blindop_set_compression(BlindOp* o, float comp) {
  if (!(comp > 0.f && comp <= 1.f)) { set_error("compression factor must be in (0, 1]"); return BUDDY_ERR_ARG; }
  if (comp != o->c.comp && o->gexec) { (void)hipGraphExecDestroy(o->gexec); o->gexec = nullptr; }      // the captured loop holds the exponent as a kernel argument
  o->c.comp = comp;
  return BUDDY_OK;
}
// which member of the l2_comp_stft family the loss entries evaluate: 0 summean (utils/losses.py:59-64), 1 sum (:46-50), 2 mean (:52-57)
int blindop_set_loss_norm(BlindOp* o, int mode) {
  if (mode < 0 || mode > 2) { set_error("loss normalisation: 0 summean, 1 sum, 2 mean"); return BUDDY_ERR_ARG; }
  if (mode != o->loss_norm && o->gexec) { (void)hipGraphExecDestroy(o->gexec); o->gexec = nullptr; }      // kappa is a kernel argument of the captured loop
  o->loss_norm = mode;
  return BUDDY_OK;
}
int blindop_lengths(BlindOp* o, int* L, int* Lr, int* T, int* Td) { if (L) *L = o->L; if (Lr) *Lr = o->Lr; if (T) *T = o->T; if (Td) *Td = o->Td; return BUDDY_OK; }

// likelihood: loss_u = w_rec * l2_comp_stft_summean(y, degrade(x_den)), g = d sum_u loss_u / d x_den  (uses the CURRENT H)
int blindop_rec_loss_grad(BlindOp* o, const float* x_den, float weight, float* loss, float* g_x, hipStream_t st) {
  o->st = st;
  const int U = o->U, T = o->T, L = o->L;
  o->stft(x_den, L, WIN, T, 1.f / o->norm, o->X1);
  o->fir(o->X1, (long long)T * LDSP, T, o->Ybuf);
  o->istft_stft(o->Ybuf, T, WIN + WIN / 2, o->env_T, L, o->norm, WIN, T, 1.f / o->norm, o->X2);
  o->comp_loss(o->Yc, o->X2, g_x ? o->X3 : nullptr, T, weight, loss, 0);
  if (g_x) {
    o->stft_adj_istft_adj(o->X3, L, WIN, T, 1.f / o->norm, T, WIN + WIN / 2, o->env_T, o->norm, o->X2);
    hipLaunchKernelGGL(fir_adjx_kernel, dim3(gridf((long long)U * T * FB)), dim3(256), 0, st, (const float*)o->X2, (const float*)o->H, o->X3, U, T, o->Nf);
    o->stft_adj(o->X3, L, WIN, T, 1.f / o->norm, g_x);
  }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

// informed likelihood (reference RIROperator.degradation = fast_apply_RIR, reverb.py:33-35, + the same STFT loss): time-domain FIR with the
// known RIR (U, M) instead of the subband filter; loss_u = weight * l2_comp_stft_summean(y, x_den * rir), g = d sum_u loss_u / d x_den
void launch_fir(const float* x, const float* h, long long h_stride, float* y, int B, int L, int M, int adjoint, hipStream_t st);
int blindop_fir_loss_grad(BlindOp* o, const float* x_den, const float* rir, long long rir_stride, int M, float weight, float* loss, float* g_x,
                          hipStream_t st) {
  o->st = st;
  const int U = o->U, T = o->T, L = o->L;
  launch_fir(x_den, rir, rir_stride, o->sig1, U, L, M, 0, st);
  o->stft(o->sig1, L, WIN, T, 1.f / o->norm, o->X2);
  o->comp_loss(o->Yc, o->X2, g_x ? o->X3 : nullptr, T, weight, loss, 0);
  if (g_x) {
    o->stft_adj(o->X3, L, WIN, T, 1.f / o->norm, o->sig2);
    launch_fir(o->sig2, rir, rir_stride, g_x, U, L, M, 1, st);
  }
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

// one gradient evaluation of  rec_loss_params(y, degrade(x_den)) + reg(rir, rir + t_op * noise)  w.r.t. (decay, weights, phases);
// H is rebuilt from the parameters first (update_H at the top of each optimize_op iteration, reference :83)
static int param_grads(BlindOp* o, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, bool have_Xd, const float* t_op_dev = nullptr) {
  const int U = o->U, T = o->T, Td = o->Td, L = o->L, Nf = o->Nf;
  hipStream_t st = o->st;
  const float norm = o->norm;
  o->update_H();
  if (!have_Xd) o->stft(x_den, L, WIN, T, 1.f / norm, o->X1);          // X1 = STFT(x_den) stays valid across the iterations
  if (noise && o->fir_lds_ok()) {
    // The reconstruction term and the RIR-noise regulariser (reference :94-100: loss(rir, (rir + t n).detach())) are the same chain
    //   FIR by H -> iSTFT -> STFT -> compressed-spectrum loss -> adjoints -> tap gradient
    // on two inputs (STFT(x_den), T frames; STFT(delta), Td frames): every kernel of the chain takes both as two jobs of ONE launch (7 launches
    // instead of 16 -- a launch of the captured loop costs ~5 us before it does anything).  The regulariser's scratch is the *_b set.
    const int Q = WIN + WIN / 2;
    o->fir2(o->X1, (long long)T * LDSP, T, o->Ybuf, o->Xdelta, 0LL, Td, o->Ybuf_b);
    o->c2r2(o->Ybuf, (long long)U * T, o->frames, o->Ybuf_b, (long long)U * Td, o->frames_b, norm / NFFT, 1);
    {   // STFT(istft(.)) of both, and STFT(rir + t n) from the regulariser's frames (the overlap-add happens in the load)
      const BlindOp::R2cReq rq[3] = {
          {R2cSrc{nullptr, o->frames, L, WIN, T, Q, o->env_T, nullptr, nullptr, 0.f, nullptr}, T, o->X2},
          {R2cSrc{nullptr, o->frames_b, o->Lr, WIN, Td, Q, o->env_d, nullptr, nullptr, 0.f, nullptr}, Td, o->X2_b},
          {R2cSrc{nullptr, o->frames_b, o->Lr, WIN, Td, Q, o->env_d, nullptr, noise, t_op, t_op_dev}, Td, o->X3_b}};
      o->r2c_multi(rq, 3, 1.f / norm, 0);
    }
    o->comp_loss2(o->Yc, o->X2, o->X3, T, w_rec, o->losses, nullptr, o->X2_b, o->X3_b, Td, w_reg, o->losses + U, true);
    o->c2r2(o->X3, (long long)U * T, o->frames, o->X3_b, (long long)U * Td, o->frames_b, 1.f / norm, 0);
    {   // istft_adj(stft_adj(.)) of both
      const BlindOp::R2cReq rq[2] = {
          {R2cSrc{nullptr, o->frames, L, Q, T, WIN, o->ones, o->env_T, nullptr, 0.f, nullptr}, T, o->X2},
          {R2cSrc{nullptr, o->frames_b, o->Lr, Q, Td, WIN, o->ones, o->env_d, nullptr, 0.f, nullptr}, Td, o->X2_b}};
      o->r2c_multi(rq, 2, norm / NFFT, 1);
    }
    o->gradh2(o->X1, (long long)T * LDSP, o->X2, T, o->Xdelta, 0LL, o->X2_b, Td, 0);
  } else {
    // reconstruction term
    o->fir(o->X1, (long long)T * LDSP, T, o->Ybuf);
    o->istft_stft(o->Ybuf, T, WIN + WIN / 2, o->env_T, L, norm, WIN, T, 1.f / norm, o->X2);
    o->comp_loss(o->Yc, o->X2, o->X3, T, w_rec, o->losses, 0);
    o->stft_adj_istft_adj(o->X3, L, WIN, T, 1.f / norm, T, WIN + WIN / 2, o->env_T, norm, o->X2);
    o->gradh(o->X1, (long long)T * LDSP, o->X2, T, 0);
    if (noise) {
      o->reg_chain(noise, t_op, t_op_dev, w_reg);
      o->gradh(o->Xdelta, 0LL, o->X2, Td, 1);
    }
  }
  o->cons_backward(o->GH);
  hipLaunchKernelGGL(h0_bwd_knots_kernel, dim3(U * Nf), dim3(256), 0, st, (const float*)o->GFin, (const float*)o->A, (const float*)o->Apre, (const float*)o->phi, o->tabs(),
                     (const float*)o->dmv, o->gphi, o->gdm, U, o->K, Nf);
  if (!o->fused_loop)      // the captured loop forms these two inside its Adam kernel
    hipLaunchKernelGGL(design_bwd_params_kernel, dim3(cdiv(U * o->E * o->NB, 4)), dim3(256), 0, st, (const float*)o->gdm, (const float*)o->decay, (const float*)o->wts, o->gdecay, o->gw, U, o->E, o->NB, Nf);
  return BUDDY_OK;
}

int blindop_param_grads(BlindOp* o, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, float* g_decay, float* g_wts,
                        float* g_phases_ref, float* losses, hipStream_t st) {
  o->st = st;
  param_grads(o, x_den, noise, t_op, w_rec, w_reg, false);
  const size_t nb = (size_t)o->U * o->E * o->NB * 4;
  if (g_decay) HIPCHK(hipMemcpyAsync(g_decay, o->gdecay, nb, hipMemcpyDeviceToDevice, st));
  if (g_wts) HIPCHK(hipMemcpyAsync(g_wts, o->gw, nb, hipMemcpyDeviceToDevice, st));
  if (g_phases_ref) hipLaunchKernelGGL(transpose_fk_kernel, dim3(gridf((long long)o->U * o->Nf * FB)), dim3(256), 0, st, (const float*)o->gphi, g_phases_ref, o->U, o->Nf, 0);
  if (losses) HIPCHK(hipMemcpyAsync(losses, o->losses, (size_t)o->U * 2 * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipGetLastError());
  return BUDDY_OK;
}

// One Adam iteration of optimize_op; dev = captured-graph mode (step counter, bias corrections and t_op read from device memory).
static void optimize_iteration(BlindOp* o, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, float lr, float b1, float b2,
                               float wd, bool have_Xd, bool dev) {
  const int U = o->U;
  hipStream_t st = o->st;
  const long long nb = (long long)U * o->E * o->NB, np = (long long)U * o->Nf * FB;
  o->fused_loop = dev;
  param_grads(o, x_den, noise, t_op, w_rec, w_reg, have_Xd, dev ? o->d_scal : nullptr);
  o->fused_loop = false;
  o->adam_step += 1;
  if (dev) {
    const AdamAll a{o->decay, o->wts, o->phi, o->gphi, o->m_d, o->v_d, o->m_w, o->v_w, o->m_p, o->v_p};
    const int pb = (int)std::min<long long>((np + 255) / 256, 2048), db = cdiv(U * o->NB, 4);
    hipLaunchKernelGGL(adam_all_kernel, dim3(pb + db), dim3(256), 0, st, a, (const float*)o->gdm, o->Nf, np, pb, U, o->E, o->NB, lr, b1, b2, 1e-8f, wd, (const int*)o->d_step,
                       (const float2*)o->bc_tab, o->c.min_decay, o->c.max_decay, o->c.w_lo, o->c.w_hi, o->c.clamp_decay, o->c.long2nd);
    return;
  }
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, o->d_step);
  const float bc1 = 1.f - std::pow(b1, (float)o->adam_step), bc2s = std::sqrt(1.f - std::pow(b2, (float)o->adam_step));
  const int* sd = nullptr;
  hipLaunchKernelGGL(adam_kernel, dim3(gridf(nb)), dim3(256), 0, st, o->decay, (const float*)o->gdecay, o->m_d, o->v_d, nb, lr, b1, b2, 1e-8f, wd, bc1, bc2s, sd, (const float2*)o->bc_tab);
  hipLaunchKernelGGL(adam_kernel, dim3(gridf(nb)), dim3(256), 0, st, o->wts, (const float*)o->gw, o->m_w, o->v_w, nb, lr, b1, b2, 1e-8f, wd, bc1, bc2s, sd, (const float2*)o->bc_tab);
  hipLaunchKernelGGL(adam_kernel, dim3(gridf(np)), dim3(256), 0, st, o->phi, (const float*)o->gphi, o->m_p, o->v_p, np, lr, b1, b2, 1e-8f, wd, bc1, bc2s, sd, (const float2*)o->bc_tab);
  hipLaunchKernelGGL(project_kernel, dim3(cdiv(U * o->NB, 256)), dim3(256), 0, st, o->decay, o->wts, U, o->E, o->NB, o->c.min_decay, o->c.max_decay,
                     o->c.w_lo, o->c.w_hi, o->c.clamp_decay, o->c.long2nd);
}

// n_iters iterations of optimize_op (reference :71-113): update_H, losses, backward, Adam step on [decay, weights, phases], projection.
// noise: (n_iters, U, Lr) standard normal draws for the RIR regulariser (NULL disables it).
// The loop is ~72 small launches per iteration and launch-bound, so it is captured ONCE into a hipGraph (on a private stream; the
// caller's stream may be the legacy default stream, which cannot capture) and replayed on the caller's stream every sampler step;
// the per-call inputs (x_den, noise, t_op) are first copied to fixed device buffers.  BUDDY_OP_GRAPH=0 keeps the eager launches.
int blindop_optimize(BlindOp* o, const float* x_den, const float* noise, float t_op, int n_iters, float w_rec, float w_reg, float lr, float b1,
                     float b2, float wd, hipStream_t st) {
  o->st = st;
  const int U = o->U;
  const bool want_graph = cur_opt().op_graph != 0;
  const bool use_graph = want_graph && n_iters > 0 && o->adam_step + n_iters < BlindOp::MAXSTEP;
  if (!use_graph) {
    for (int it = 0; it < n_iters; ++it)
      optimize_iteration(o, x_den, noise ? noise + (long long)it * U * o->Lr : nullptr, t_op, w_rec, w_reg, lr, b1, b2, wd, it > 0, false);
    HIPCHK(hipGetLastError());
    return BUDDY_OK;
  }
  if (o->tab_b1 != b1 || o->tab_b2 != b2) {               // bias-correction table, same host arithmetic as the eager path
    std::vector<float2> tab(BlindOp::MAXSTEP);
    for (int k = 0; k < BlindOp::MAXSTEP; ++k) tab[k] = make_float2(1.f - std::pow(b1, (float)k), std::sqrt(1.f - std::pow(b2, (float)k)));
    HIPCHK(hipMemcpy(o->bc_tab, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    o->tab_b1 = b1; o->tab_b2 = b2;
  }
  if (noise && o->noise_iters < n_iters) {
    if (o->dalloc(&o->noise_buf, (size_t)n_iters * U * o->Lr)) { return BUDDY_ERR_HIP; }
    o->noise_iters = n_iters;
    if (o->gexec) { (void)hipGraphExecDestroy(o->gexec); o->gexec = nullptr; }
  }
  HIPCHK(hipMemcpyAsync(o->xden_buf, x_den, (size_t)U * o->L * 4, hipMemcpyDeviceToDevice, st));
  if (noise) HIPCHK(hipMemcpyAsync(o->noise_buf, noise, (size_t)n_iters * U * o->Lr * 4, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, o->d_scal, t_op);
  const float hp[7] = {w_rec, w_reg, lr, b1, b2, wd, noise ? 1.f : 0.f};
  if (!o->gexec || o->g_iters != n_iters || std::memcmp(hp, o->g_hp, sizeof(hp)) != 0) {
    if (o->gexec) { (void)hipGraphExecDestroy(o->gexec); o->gexec = nullptr; }
    if (!o->cap_stream) HIPCHK(hipStreamCreateWithFlags(&o->cap_stream, hipStreamNonBlocking));
    const bool prof = igemm_prof_enabled();
    igemm_prof_enable(0);                                // no event records inside the captured region
    const int step0 = o->adam_step;
    HIPCHK(hipStreamBeginCapture(o->cap_stream, hipStreamCaptureModeThreadLocal));
    o->st = o->cap_stream;
    for (int it = 0; it < n_iters; ++it)
      optimize_iteration(o, o->xden_buf, noise ? o->noise_buf + (long long)it * U * o->Lr : nullptr, t_op, w_rec, w_reg, lr, b1, b2, wd, it > 0, true);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(o->cap_stream, &graph);
    o->st = st; o->adam_step = step0;
    igemm_prof_enable(prof ? 1 : 0);
    if (ce != hipSuccess || !graph) { set_error(std::string("optimize_op graph capture failed: ") + hipGetErrorString(ce)); return BUDDY_ERR_HIP; }
    const hipError_t ie = hipGraphInstantiate(&o->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { o->gexec = nullptr; set_error(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(ie)); return BUDDY_ERR_HIP; }
    o->g_iters = n_iters; std::memcpy(o->g_hp, hp, sizeof(hp));
  }
  HIPCHK(hipGraphLaunch(o->gexec, st));
  o->adam_step += n_iters;
  return BUDDY_OK;
}

}  // namespace buddy
