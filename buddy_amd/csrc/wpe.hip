// Weighted prediction error (WPE) dereverberation, single channel, batch "full statistics" variant, complex128 -- the warm start
// `wpe_scaled` of the blind sampler (reference testing/EulerHeunSamplerDPS.py:32-54 calls nara_wpe.wpe.wpe(Y, taps=50, delay=2,
// iterations=5, statistics_mode='full'); nara_wpe is a third-party package absent from the reference tree, restated from its
// published algorithm -- parity unpinned, see buddy_amd/utils/wpe.py).
//
// One workgroup per (utterance, frequency bin) row of T frames.  Per iteration:
//   lambda_t = max(|x_t|^2, 1e-10 max_t |x_t|^2);   R = sum_t yt_t yt_t^H / lambda_t,  P = sum_t yt_t conj(y_t) / lambda_t,
//   yt_t = [y_{t-delay}, ..., y_{t-delay-taps+1}];   G = R^{-1} P (Cholesky in LDS);   x_t = y_t - G^H yt_t.
// Runs once per utterance outside the sampling loop; written for clarity, not speed (fp64 VALU, K = 50).
#include "common.h"

namespace buddy {
namespace {
struct c128 { double x, y; };
__device__ __forceinline__ c128 cmul(c128 a, c128 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c128 cmulc(c128 a, c128 b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a * conj(b)

constexpr int WPE_NT = 256;

__global__ __launch_bounds__(WPE_NT) void wpe_kernel(const c128* __restrict__ Y, c128* __restrict__ X, double* __restrict__ inv_all, int T, int K,
                                                     int delay, int iters) {
  extern __shared__ double smem[];
  c128* R = reinterpret_cast<c128*>(smem);                 // K x K, lower triangle is overwritten by the Cholesky factor
  c128* P = R + K * K;                                      // K
  c128* G = P + K;                                          // K
  double* red = reinterpret_cast<double*>(G + K);           // WPE_NT
  const int row = blockIdx.x, tid = threadIdx.x;
  const c128* y = Y + (long long)row * T;
  c128* x = X + (long long)row * T;
  double* inv = inv_all + (long long)row * T;
  for (int t = tid; t < T; t += WPE_NT) x[t] = y[t];
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // inverse power of the current estimate
    double pm = 0.0;
    for (int t = tid; t < T; t += WPE_NT) { const double p = x[t].x * x[t].x + x[t].y * x[t].y; pm = p > pm ? p : pm; }
    red[tid] = pm;
    __syncthreads();
    for (int s = WPE_NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = red[tid] > red[tid + s] ? red[tid] : red[tid + s]; __syncthreads(); }
    const double floor_p = 1e-10 * red[0];
    __syncthreads();
    for (int t = tid; t < T; t += WPE_NT) { const double p = x[t].x * x[t].x + x[t].y * x[t].y; inv[t] = 1.0 / (p > floor_p ? p : floor_p); }
    __syncthreads();
    // correlation matrix (lower triangle, mirrored) and vector
    const int ntri = K * (K + 1) / 2;
    for (int e = tid; e < ntri + K; e += WPE_NT) {
      if (e < ntri) {
        int i = 0; while ((i + 1) * (i + 2) / 2 <= e) ++i;   // row of the packed lower triangle
        const int j = e - i * (i + 1) / 2;
        c128 acc = {0.0, 0.0};
        for (int t = delay + i; t < T; ++t) {
          const c128 a = y[t - delay - i], b = y[t - delay - j];
          const double w = inv[t];
          acc.x += w * (a.x * b.x + a.y * b.y); acc.y += w * (a.y * b.x - a.x * b.y);
        }
        R[i * K + j] = acc; R[j * K + i] = {acc.x, -acc.y};
      } else {
        const int i = e - ntri;
        c128 acc = {0.0, 0.0};
        for (int t = delay + i; t < T; ++t) {
          const c128 a = y[t - delay - i], b = y[t];
          const double w = inv[t];
          acc.x += w * (a.x * b.x + a.y * b.y); acc.y += w * (a.y * b.x - a.x * b.y);
        }
        P[i] = acc;
      }
    }
    __syncthreads();
    // Cholesky R = L L^H (right-looking); L overwrites the lower triangle
    for (int c = 0; c < K; ++c) {
      const double dcc = R[c * K + c].x;
      const double l = dcc > 0.0 ? sqrt(dcc) : 0.0;
      const double il = l > 0.0 ? 1.0 / l : 0.0;           // an all-zero row (silent bin) yields G = 0 instead of NaNs
      __syncthreads();
      if (tid == 0) R[c * K + c] = {l, 0.0};
      for (int i = c + 1 + tid; i < K; i += WPE_NT) { R[i * K + c].x *= il; R[i * K + c].y *= il; }
      __syncthreads();
      const int m = K - c - 1;
      for (int e = tid; e < m * m; e += WPE_NT) {
        const int i = c + 1 + e / m, j = c + 1 + e % m;
        if (j <= i) { const c128 v = cmulc(R[i * K + c], R[j * K + c]); R[i * K + j].x -= v.x; R[i * K + j].y -= v.y; }
      }
      __syncthreads();
    }
    if (tid == 0) {                                         // L z = P,  L^H G = z   (K = 50: serial)
      for (int i = 0; i < K; ++i) {
        c128 s = P[i];
        for (int j = 0; j < i; ++j) { const c128 v = cmul(R[i * K + j], G[j]); s.x -= v.x; s.y -= v.y; }
        const double l = R[i * K + i].x;
        G[i] = l > 0.0 ? c128{s.x / l, s.y / l} : c128{0.0, 0.0};
      }
      for (int i = K - 1; i >= 0; --i) {
        c128 s = G[i];
        for (int j = i + 1; j < K; ++j) { const c128 lj = R[j * K + i]; const c128 v = cmul(c128{lj.x, -lj.y}, G[j]); s.x -= v.x; s.y -= v.y; }
        const double l = R[i * K + i].x;
        G[i] = l > 0.0 ? c128{s.x / l, s.y / l} : c128{0.0, 0.0};
      }
    }
    __syncthreads();
    for (int t = tid; t < T; t += WPE_NT) {
      c128 s = y[t];
      const int kmax = t - delay < K - 1 ? t - delay : K - 1;
      for (int k = 0; k <= kmax; ++k) { const c128 v = cmulc(y[t - delay - k], G[k]); s.x -= v.x; s.y -= v.y; }   // x_t = y_t - sum_k conj(G_k) y_{t-delay-k}
      x[t] = s;
    }
    __syncthreads();
  }
}

// ---- STFT / iSTFT of the warm start (nara_wpe.utils.stft / istft conventions: size 512, shift 128, periodic Blackman analysis window,
// size - shift zeros of "fading" on both sides, tail zero-padded to a whole frame, bi-orthogonal synthesis window), complex128.
// 512-point transforms as direct sums against a twiddle table in LDS: 0.13 GFLOP per 4 s utterance, once per run -- clarity over speed.
constexpr int WS = 512, WH = 128, WF = WS / 2 + 1, WPAD = WS - WH;

__device__ __forceinline__ double blackman_periodic(int n) {
  const double a = 2.0 * M_PI * (double)n / (double)WS;
  return 0.42 - 0.5 * cos(a) + 0.08 * cos(2.0 * a);
}

// Y[(b * WF + f) * T + t] = sum_n w[n] ypad[t * WH + n] e^{-2 pi i f n / WS};   grid (T, B), 256 threads
__global__ __launch_bounds__(256) void wpe_stft_kernel(const float* __restrict__ y, c128* __restrict__ Y, int L, int T) {
  __shared__ double xw[WS], tc[WS], ts[WS];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  for (int n = tid; n < WS; n += 256) {
    const long long p = (long long)t * WH + n - WPAD;                   // index into the un-padded signal
    const double v = (p >= 0 && p < L) ? (double)y[(long long)b * L + p] : 0.0;
    xw[n] = v * blackman_periodic(n);
    double sn, cs; sincos(-2.0 * M_PI * (double)n / (double)WS, &sn, &cs);
    tc[n] = cs; ts[n] = sn;
  }
  __syncthreads();
  for (int f = tid; f < WF; f += 256) {
    double re = 0.0, im = 0.0;
    for (int n = 0; n < WS; ++n) { const int k = (f * n) & (WS - 1); re += xw[n] * tc[k]; im += xw[n] * ts[k]; }
    Y[((long long)b * WF + f) * T + t] = {re, im};
  }
}

// out[b][j] = sum over the (at most 4) frames covering padded position j + WPAD of wsyn[n] * irfft(Z[:, t])[n];  grid (ceil(L / WH), B), 512 threads:
// thread (q, m) computes sample n = q * WH + m of frame h + ... so that all four land on output position h * WH + m
__global__ __launch_bounds__(512) void wpe_istft_kernel(const c128* __restrict__ Z, float* __restrict__ out, int L, int T) {
  __shared__ double tc[WS], ts[WS], acc[4][WH];
  __shared__ c128 zf[4][WF];
  const int hb = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int q = tid >> 7, m = tid & (WH - 1);
  const int h = hb + WPAD / WH;                                         // hop index in the padded signal
  for (int n = tid; n < WS; n += 512) { double sn, cs; sincos(2.0 * M_PI * (double)n / (double)WS, &sn, &cs); tc[n] = cs; ts[n] = sn; }
  for (int e = tid; e < 4 * WF; e += 512) {
    const int qq = e / WF, f = e - qq * WF, t = h - qq;
    zf[qq][f] = (t >= 0 && t < T) ? Z[((long long)b * WF + f) * T + t] : c128{0.0, 0.0};
  }
  __syncthreads();
  {
    const int n = q * WH + m;                                           // frame t = h - q contributes its sample n to padded position h * WH + m
    double s = zf[q][0].x + ((n & 1) ? -zf[q][WS / 2].x : zf[q][WS / 2].x);      // irfft ignores Im of the DC and Nyquist bins
    for (int f = 1; f < WS / 2; ++f) { const int k = (f * n) & (WS - 1); s += 2.0 * (zf[q][f].x * tc[k] - zf[q][f].y * ts[k]); }
    // synthesis window: analysis window / sum of its squares over the WS / WH overlapping positions
    double den = 0.0;
    for (int r = 0; r < WS / WH; ++r) { const double w = blackman_periodic(r * WH + m); den += w * w; }
    acc[q][m] = s * (1.0 / WS) * blackman_periodic(n) / den;
  }
  __syncthreads();
  if (tid < WH) {
    const long long j = (long long)hb * WH + tid;
    if (j < L) out[(long long)b * L + j] = (float)(((acc[3][tid] + acc[2][tid]) + acc[1][tid]) + acc[0][tid]);   // overlap-add in ascending frame order
  }
}
}  // namespace

size_t wpe_lds_bytes(int K) { return (size_t)(K * K + 2 * K) * sizeof(double) * 2 + WPE_NT * sizeof(double); }

void launch_wpe(const double* Y, double* X, double* inv_scratch, int rows, int T, int taps, int delay, int iters, hipStream_t st) {
  hipLaunchKernelGGL(wpe_kernel, dim3(rows), dim3(WPE_NT), wpe_lds_bytes(taps), st, reinterpret_cast<const c128*>(Y), reinterpret_cast<c128*>(X),
                     inv_scratch, T, taps, delay, iters);
}

int wpe_frames(int L) {                                      // frames of nara_wpe.utils.stft(size 512, shift 128, fading, pad)
  long long n = (long long)L + 2 * WPAD;
  if (n < WS) n = WS;
  else if ((n + WH - WS) % WH) n += WH - (n + WH - WS) % WH;
  return (int)((n - WS) / WH + 1);
}

size_t wpe_workspace_bytes(int B, int L) {                   // Y, X (complex128) + inverse-power scratch
  const size_t rows = (size_t)B * WF, T = (size_t)wpe_frames(L);
  return rows * T * (2 * sizeof(c128) + sizeof(double));
}

void launch_wpe_dereverb(const float* y, float* out, void* work, int B, int L, int taps, int delay, int iters, hipStream_t st) {
  const int T = wpe_frames(L);
  const size_t rows = (size_t)B * WF;
  c128* Y = reinterpret_cast<c128*>(work);
  c128* X = Y + rows * T;
  double* inv = reinterpret_cast<double*>(X + rows * T);
  hipLaunchKernelGGL(wpe_stft_kernel, dim3(T, B), dim3(256), 0, st, y, Y, L, T);
  hipLaunchKernelGGL(wpe_kernel, dim3((unsigned)rows), dim3(WPE_NT), wpe_lds_bytes(taps), st, Y, X, inv, T, taps, delay, iters);
  hipLaunchKernelGGL(wpe_istft_kernel, dim3((L + WH - 1) / WH, B), dim3(512), 0, st, X, out, L, T);
}

}  // namespace buddy
