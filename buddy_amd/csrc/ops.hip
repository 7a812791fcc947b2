// HBM-bound kernels of the score network and STFT glue (gfx950): GroupNorm statistics / apply(+SiLU, +2x resample)
// and their input-gradients, 2-channel direct convs, pooling, softmax rows, small linears, reflect-pad / overlap-add.
// All tensors fp32 NHWC (see common.h); every kernel moves float4 per lane along the channel axis (coalesced 16 B/lane).
#include "common.h"
#include <cstdlib>

namespace buddy {
namespace {

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + expf(-z)); }
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = 1.f / (1.f + expf(-z));
  return s * (1.f + z * (1.f - s));
}
__device__ __forceinline__ const float* src_ptr(const Src2& x, long long pix, int c) {
  return (x.p1 != nullptr && c >= x.C0) ? x.p1 + pix * x.ld1 + (c - x.C0) : x.p0 + pix * x.ld0 + c;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// streaming form (round 6): non-temporal, for tensors a kernel reads exactly once in whole 128-byte lines (lanes along the channel quads) -- the
// activation / gradient streams of the GroupNorm kernels: 4.03 -> 4.24 TB/s for the group (the streaming ubench: reads 6.3 -> 7.1 TB/s)
typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p) { const f32x4s v = __builtin_nontemporal_load(reinterpret_cast<const f32x4s*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// gradient of the (resampled) activation arriving at full-resolution pixel (b,h,w), channels c..c+3
__device__ __forceinline__ float4 da_eff(const float* da, int mode, int b, int h, int w, int H, int W, int C, int c) {
  if (mode == 0) return ld4(da + (((long long)b * H + h) * W + w) * C + c);
  if (mode == 1) {
    const int H2 = H >> 1, W2 = W >> 1;
    return mul4(ld4(da + (((long long)b * H2 + (h >> 1)) * W2 + (w >> 1)) * C + c), 0.25f);
  }
  const int H2 = H << 1, W2 = W << 1;
  const float* q = da + (((long long)b * H2 + 2 * h) * W2 + 2 * w) * C + c;
  return add4(add4(ld4(q), ld4(q + C)), add4(ld4(q + (long long)W2 * C), ld4(q + (long long)W2 * C + C)));
}

// ------------------------------------------------------------------ per-channel two-value reduction over pixels
// KIND 0: (x, x^2) for the forward statistics.  KIND 1: (dxhat, dxhat*xhat) for the backward.
struct RedArgs {
  Src2 x; int B, H, W, C, G; int chunks, ppc;
  const float* stats; const float* gamma; const float* beta; const float* da; int mode; int silu;
  double* partial;
};

template <int KIND>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const RedArgs a) {
  __shared__ double red[256 * 8];
  const int q = a.C >> 2, pl = 256 / q;
  const int tid = threadIdx.x, quad = tid % q, lp = tid / q;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int HW = a.H * a.W;
  const int p0 = chunk * a.ppc, p1 = min(HW, p0 + a.ppc);
  double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
  if (lp < pl) {
    const int c = quad * 4;
    float mean = 0.f, rstd = 0.f; float4 gm = make_float4(0, 0, 0, 0), bt = gm;
    if (KIND == 1) {
      const int g = c / (a.C / a.G);
      mean = a.stats[((long long)b * a.G + g) * 2]; rstd = a.stats[((long long)b * a.G + g) * 2 + 1];
      gm = ld4(a.gamma + c); bt = ld4(a.beta + c);
    }
    if (KIND == 1 && a.mode == 0) {
      // backward sums, same resolution: fp32 partials over strips of 8 pixels (two loads in flight per tensor), flushed into the fp64 sums
      // (the forward statistics stay fp64 throughout: E[x^2] - mean^2 cancels)
      const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
      const bool second = a.x.p1 != nullptr && c >= a.x.C0;
      const float* xs = (second ? a.x.p1 + (c - a.x.C0) : a.x.p0 + c) + (long long)b * HW * (second ? a.x.ld1 : a.x.ld0);
      const long long ldx = second ? a.x.ld1 : a.x.ld0;
      const float* ds = a.da + (long long)b * HW * a.C + c;
      int p = p0 + lp;
      while (p < p1) {
        float fs[4] = {0.f, 0.f, 0.f, 0.f}, ft[4] = {0.f, 0.f, 0.f, 0.f};
        auto one = [&](float4 v, float4 d) {
          const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mean) * rstd;
            const float z = xh * gv[j] + bv[j];
            const float dxh = dv[j] * (a.silu ? dsilu_f(z) : 1.f) * gv[j];
            fs[j] += dxh; ft[j] += dxh * xh;
          }
        };
        int n = 0;
        for (; n < 8 && p + pl < p1; n += 2, p += 2 * pl) {
          const float4 v0 = ld4s(xs + (long long)p * ldx), v1 = ld4s(xs + (long long)(p + pl) * ldx);
          const float4 d0 = ld4s(ds + (long long)p * a.C), d1 = ld4s(ds + (long long)(p + pl) * a.C);
          one(v0, d0); one(v1, d1);
        }
        if (n < 8 && p < p1) { one(ld4s(xs + (long long)p * ldx), ld4s(ds + (long long)p * a.C)); p += pl; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += (double)fs[j]; t[j] += (double)ft[j]; }
      }
    } else
    for (int p = p0 + lp; p < p1; p += pl) {
      const float4 v = ld4s(src_ptr(a.x, (long long)b * HW + p, c));
      const float xv[4] = {v.x, v.y, v.z, v.w};
      if (KIND == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += (double)xv[j]; t[j] += (double)xv[j] * (double)xv[j]; }
      } else {
        const int h = p / a.W, w = p - h * a.W;
        const float4 d = da_eff(a.da, a.mode, b, h, w, a.H, a.W, a.C, c);
        const float dv[4] = {d.x, d.y, d.z, d.w};
        const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mean) * rstd;
          const float z = xh * gv[j] + bv[j];
          const float dxh = dv[j] * (a.silu ? dsilu_f(z) : 1.f) * gv[j];
          s[j] += (double)dxh; t[j] += (double)dxh * (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = t[j]; }
  __syncthreads();
  if (tid < q) {
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int l = 0; l < pl; ++l)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += red[(l * q + tid) * 8 + j];
    double* o = a.partial + (((long long)b * a.chunks + chunk) * a.C + tid * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j * 2] = r[j]; o[j * 2 + 1] = r[4 + j]; }
  }
}

// one workgroup of 256 threads per (b, group): combine chunk partials in a fixed order (lane-strided sums, wave shuffle, four wave results through
// LDS).  KIND 0 -> (mean, rstd); KIND 1 -> (mean dxhat, mean dxhat*xhat).  (One wave per group took 18 us on ~2000 partial pairs: 45 launches per step.)
template <int KIND>
__global__ __launch_bounds__(256) void group_finalize_kernel(const double* partial, float* out, int C, int G, int chunks, int HW, float eps) {
  __shared__ double red[8];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpg = C / G;
  double s = 0, t = 0;
  for (int i = tid; i < chunks * cpg; i += 256) {
    const int ch = i / cpg, c = g * cpg + (i - ch * cpg);
    const double2 v = *reinterpret_cast<const double2*>(partial + (((long long)b * chunks + ch) * C + c) * 2);
    s += v.x; t += v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); t += __shfl_down(t, off, 64); }
  if (lane == 0) { red[2 * w] = s; red[2 * w + 1] = t; }
  __syncthreads();
  if (tid == 0) {
    s = ((red[0] + red[2]) + red[4]) + red[6]; t = ((red[1] + red[3]) + red[5]) + red[7];
    const double n = (double)cpg * (double)HW;
    float* o = out + ((long long)b * G + g) * 2;
    if (KIND == 0) {
      const double mean = s / n;
      double var = t / n - mean * mean;
      if (var < 0) var = 0;
      o[0] = (float)mean; o[1] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      o[0] = (float)(s / n); o[1] = (float)(t / n);
    }
  }
}

// partial[b][chunks][C][2] -> csum[b][c][2]: block = 16 channels x 16 chunk slices, fixed summation order
__global__ __launch_bounds__(256) void csum_collapse_kernel(const double* __restrict__ partial, int chunks, int C, double* __restrict__ csum) {
  __shared__ double red[256 * 2];
  const int tid = threadIdx.x, cl = tid & 15, sl = tid >> 4, b = blockIdx.y;
  const int c = blockIdx.x * 16 + cl;
  double s = 0, t = 0;
  if (c < C)
    for (int ch = sl; ch < chunks; ch += 16) {
      const double2 v = *reinterpret_cast<const double2*>(partial + (((long long)b * chunks + ch) * C + c) * 2);
      s += v.x; t += v.y;
    }
  red[tid * 2] = s; red[tid * 2 + 1] = t;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int l = 1; l < 16; ++l) { s += red[(l * 16 + cl) * 2]; t += red[(l * 16 + cl) * 2 + 1]; }
    csum[((long long)b * C + c) * 2] = s; csum[((long long)b * C + c) * 2 + 1] = t;
  }
}

// one wave per (b, group) of a (concatenated) view: (mean, rstd) from the per-channel sums of its sources
__global__ __launch_bounds__(64) void group_finalize_csum_kernel(const double* __restrict__ c0, const double* __restrict__ c1, int C0, int C, int G,
                                                                 int HW, float eps, float* __restrict__ out) {
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int cpg = C / G;
  double s = 0, t = 0;
  for (int i = lane; i < cpg; i += 64) {
    const int c = g * cpg + i;
    const double* p = (c1 != nullptr && c >= C0) ? c1 + ((long long)b * (C - C0) + (c - C0)) * 2 : c0 + ((long long)b * C0 + c) * 2;
    s += p[0]; t += p[1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); t += __shfl_down(t, off, 64); }
  if (lane == 0) {
    const double n = (double)cpg * (double)HW;
    float* o = out + ((long long)b * G + g) * 2;
    const double mean = s / n;
    double var = t / n - mean * mean;
    if (var < 0) var = 0;
    o[0] = (float)mean; o[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ------------------------------------------------------------------ GroupNorm apply (+SiLU) with optional 2x resample
__global__ __launch_bounds__(256) void gn_apply_kernel(Src2 x, const float* stats, const float* gamma, const float* beta, int B, int H,
                                                       int W, int C, int G, int mode, int silu, float* out, float* pooled_raw) {
  const int q = C >> 2, cpg = C / G;
  const int Ho = (mode == 1) ? H >> 1 : H, Wo = (mode == 1) ? W >> 1 : W;   // iteration space (mode 2 iterates inputs)
  const long long total = (long long)B * Ho * Wo * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int quad = (int)(idx % q); long long pix = idx / q;
    const int w = (int)(pix % Wo); pix /= Wo; const int h = (int)(pix % Ho); const int b = (int)(pix / Ho);
    const int c = quad * 4, g = c / cpg;
    const float mean = stats[((long long)b * G + g) * 2], rstd = stats[((long long)b * G + g) * 2 + 1];
    const float4 gm = ld4(gamma + c), bt = ld4(beta + c);
    auto act = [&](float4 v) {
      float4 z = make_float4((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y,
                             (v.z - mean) * rstd * gm.z + bt.z, (v.w - mean) * rstd * gm.w + bt.w);
      if (silu) z = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
      return z;
    };
    if (mode == 0) {
      const long long p = ((long long)b * H + h) * W + w;
      st4(out + p * C + c, act(ld4(src_ptr(x, p, c))));
    } else if (mode == 1) {
      const long long p = ((long long)b * H + 2 * h) * W + 2 * w;
      const float4 v0 = ld4(src_ptr(x, p, c)), v1 = ld4(src_ptr(x, p + 1, c)), v2 = ld4(src_ptr(x, p + W, c)), v3 = ld4(src_ptr(x, p + W + 1, c));
      const long long po = ((long long)b * Ho + h) * Wo + w;
      st4(out + po * C + c, mul4(add4(add4(act(v0), act(v1)), add4(act(v2), act(v3))), 0.25f));
      if (pooled_raw) st4(pooled_raw + po * C + c, mul4(add4(add4(v0, v1), add4(v2, v3)), 0.25f));
    } else {
      const long long p = ((long long)b * H + h) * W + w;
      const float4 a = act(ld4(src_ptr(x, p, c)));
      const int W2 = W << 1;
      float* o = out + ((((long long)b * (H << 1)) + 2 * h) * W2 + 2 * w) * C + c;
      st4(o, a); st4(o + C, a); st4(o + (long long)W2 * C, a); st4(o + (long long)W2 * C + C, a);
    }
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da,
                                                           int B, int H, int W, int C, int G, int mode, int silu, const float* extra,
                                                           int extra_mode, float extra_scale, const float* red, Dst2 dx) {
  const int q = C >> 2, cpg = C / G;
  const long long total = (long long)B * H * W * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int quad = (int)(idx % q); long long pix = idx / q;
    const long long p = pix;
    const int w = (int)(pix % W); pix /= W; const int h = (int)(pix % H); const int b = (int)(pix / H);
    const int c = quad * 4, g = c / cpg;
    const float mean = stats[((long long)b * G + g) * 2], rstd = stats[((long long)b * G + g) * 2 + 1];
    const float m1 = red[((long long)b * G + g) * 2], m2 = red[((long long)b * G + g) * 2 + 1];
    const float4 gm = ld4(gamma + c), bt = ld4(beta + c);
    const float4 v = ld4s(src_ptr(x, p, c));
    const float4 d = da_eff(da, mode, b, h, w, H, W, C, c);
    const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
    const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xv[j] - mean) * rstd;
      const float z = xh * gv[j] + bv[j];
      const float dxh = dv[j] * (silu ? dsilu_f(z) : 1.f) * gv[j];
      r[j] = rstd * (dxh - m1 - xh * m2);
    }
    if (extra_mode == 1) {
      const float4 e = ld4(extra + p * C + c);
      r[0] += extra_scale * e.x; r[1] += extra_scale * e.y; r[2] += extra_scale * e.z; r[3] += extra_scale * e.w;
    } else if (extra_mode == 2) {
      const float4 e = ld4(extra + ((((long long)b * (H >> 1)) + (h >> 1)) * (W >> 1) + (w >> 1)) * C + c);
      const float s = 0.25f * extra_scale;
      r[0] += s * e.x; r[1] += s * e.y; r[2] += s * e.z; r[3] += s * e.w;
    }
    float* o; int acc;
    if (dx.p1 != nullptr && c >= dx.C0) { o = dx.p1 + p * dx.ld1 + (c - dx.C0); acc = dx.acc1; }
    else { o = dx.p0 + p * dx.ld0 + c; acc = dx.acc0; }
    float4 res = make_float4(r[0], r[1], r[2], r[3]);
    if (acc) res = add4(res, ld4(o));
    st4(o, res);
  }
}

// ------------------------------------------------------------------ mode-0 fast paths (most GroupNorms of the network)
// grid (chunks, B), block = q * pl threads (q = C/4 channel quads, pl pixels in flight per block): a thread keeps its channel quad, so
// gamma/beta/statistics, the concat-source choice and all address arithmetic except one multiply-add per pixel leave the loop; four
// pixels per trip with the loads issued first.
struct GnFast { int HW, C, G, q, pl, ppc; };
__global__ __launch_bounds__(256) void gn_apply_m0_kernel(Src2 x, const float* stats, const float* gamma, const float* beta, GnFast a, int silu, float* out) {
  const int tid = threadIdx.x, quad = tid % a.q, lp = tid / a.q, b = blockIdx.y;
  const int c = quad * 4, g = c / (a.C / a.G);
  const float mean = stats[((long long)b * a.G + g) * 2], rstd = stats[((long long)b * a.G + g) * 2 + 1];
  const float4 gm = ld4(gamma + c), bt = ld4(beta + c);
  const bool second = x.p1 != nullptr && c >= x.C0;
  const float* src = second ? x.p1 + (c - x.C0) : x.p0 + c;
  const long long ld = second ? x.ld1 : x.ld0;
  src += (long long)b * a.HW * ld;
  float* dst = out + (long long)b * a.HW * a.C + c;
  const int p0 = blockIdx.x * a.ppc, p1 = min(a.HW, p0 + a.ppc);
  auto act = [&](float4 v) {
    float4 z = make_float4((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y,
                           (v.z - mean) * rstd * gm.z + bt.z, (v.w - mean) * rstd * gm.w + bt.w);
    if (silu) z = make_float4(silu_f(z.x), silu_f(z.y), silu_f(z.z), silu_f(z.w));
    return z;
  };
  int p = p0 + lp;
  for (; p + 3 * a.pl < p1; p += 4 * a.pl) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ld4s(src + (long long)(p + j * a.pl) * ld);
#pragma unroll
    for (int j = 0; j < 4; ++j) st4(dst + (long long)(p + j * a.pl) * a.C, act(v[j]));
  }
  for (; p < p1; p += a.pl) st4(dst + (long long)p * a.C, act(ld4s(src + (long long)p * ld)));
}

// extra_mode 0 / 1 (same-resolution extra gradient), da at the same resolution.  EX: an extra gradient exists; ACC: some destination accumulates.
// Both are template parameters so that every load of a pixel pair is unconditional and in flight together: as runtime `ptr ? ld4(..) : 0` they
// were branches with a vmcnt(0) between the streams (x, da, extra -> wait -> previous value).  A destination that does not accumulate still reads
// its (arena) memory under ACC and discards the value by a select.
template <bool EX, bool ACC>
__global__ __launch_bounds__(256) void gn_bwd_apply_m0_kernel(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da,
                                                              GnFast a, int silu, const float* extra, float extra_scale, const float* red, Dst2 dx) {
  const int tid = threadIdx.x, quad = tid % a.q, lp = tid / a.q, b = blockIdx.y;
  const int c = quad * 4, g = c / (a.C / a.G);
  const float mean = stats[((long long)b * a.G + g) * 2], rstd = stats[((long long)b * a.G + g) * 2 + 1];
  const float m1 = red[((long long)b * a.G + g) * 2], m2 = red[((long long)b * a.G + g) * 2 + 1];
  const float4 gm = ld4(gamma + c), bt = ld4(beta + c);
  const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
  const bool second = x.p1 != nullptr && c >= x.C0;
  const float* src = second ? x.p1 + (c - x.C0) : x.p0 + c;
  const long long ld = second ? x.ld1 : x.ld0;
  src += (long long)b * a.HW * ld;
  const float* dap = da + (long long)b * a.HW * a.C + c;
  const float* ex = EX ? extra + (long long)b * a.HW * a.C + c : nullptr;
  const bool dsecond = dx.p1 != nullptr && c >= dx.C0;
  float* o = dsecond ? dx.p1 + (c - dx.C0) : dx.p0 + c;
  const long long ldo = dsecond ? dx.ld1 : dx.ld0;
  const bool acc = ACC && (dsecond ? dx.acc1 : dx.acc0) != 0;
  o += (long long)b * a.HW * ldo;
  const int p0 = blockIdx.x * a.ppc, p1 = min(a.HW, p0 + a.ppc);
  auto one = [&](float4 v, float4 d, float4 e, float4 prev) {
    const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w}, ev[4] = {e.x, e.y, e.z, e.w}, pv[4] = {prev.x, prev.y, prev.z, prev.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xv[j] - mean) * rstd;
      const float z = xh * gv[j] + bv[j];
      const float dxh = dv[j] * (silu ? dsilu_f(z) : 1.f) * gv[j];
      r[j] = rstd * (dxh - m1 - xh * m2);
      if (EX) r[j] += extra_scale * ev[j];
      if (ACC) r[j] = acc ? r[j] + pv[j] : r[j];
    }
    return make_float4(r[0], r[1], r[2], r[3]);
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = p0 + lp;
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the parameter loads are complete, no wait inside the loop refers back to them (and to the stores)
  for (; p + a.pl < p1; p += 2 * a.pl) {
    float4 v[2], d[2], e[2], pr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long long pp = p + j * a.pl;
      v[j] = ld4s(src + pp * ld); d[j] = ld4s(dap + pp * a.C); e[j] = EX ? ld4s(ex + pp * a.C) : z4; pr[j] = ACC ? ld4(o + pp * ldo) : z4;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) st4(o + (long long)(p + j * a.pl) * ldo, one(v[j], d[j], e[j], pr[j]));
  }
  for (; p < p1; p += a.pl)
    st4(o + (long long)p * ldo, one(ld4s(src + (long long)p * ld), ld4s(dap + (long long)p * a.C), EX ? ld4s(ex + (long long)p * a.C) : z4, ACC ? ld4(o + (long long)p * ldo) : z4));
}

// ------------------------------------------------------------------ elementwise helpers
__global__ __launch_bounds__(256) void axpy_kernel(float* dst, const float* src, float alpha, long long n4, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = mul4(ld4(src + i * 4), alpha);
    if (accumulate) v = add4(v, ld4(dst + i * 4));
    st4(dst + i * 4, v);
  }
}

__global__ __launch_bounds__(256) void pool2_kernel(const float* src, float* dst, int B, int H, int W, int C, float scale, int accumulate) {
  const int q = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * Ho * Wo * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int quad = (int)(idx % q); long long pix = idx / q;
    const int w = (int)(pix % Wo); pix /= Wo; const int h = (int)(pix % Ho); const int b = (int)(pix / Ho);
    const float* s = src + ((((long long)b * H) + 2 * h) * W + 2 * w) * C + quad * 4;
    float4 v = mul4(add4(add4(ld4(s), ld4(s + C)), add4(ld4(s + (long long)W * C), ld4(s + (long long)W * C + C))), scale);
    float* o = dst + idx * 4;
    if (accumulate) v = add4(v, ld4(o));
    st4(o, v);
  }
}

// C == 2 variant (input pyramid): float2 per pixel
__global__ __launch_bounds__(256) void pool2_c2_kernel(const float* src, float* dst, int B, int H, int W, float scale, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    long long pix = idx;
    const int w = (int)(pix % Wo); pix /= Wo; const int h = (int)(pix % Ho); const int b = (int)(pix / Ho);
    const float2* s = reinterpret_cast<const float2*>(src) + (((long long)b * H) + 2 * h) * W + 2 * w;
    const float2 a = s[0], b1 = s[1], c = s[W], d = s[W + 1];
    float2 v = make_float2(((a.x + b1.x) + (c.x + d.x)) * scale, ((a.y + b1.y) + (c.y + d.y)) * scale);
    float2* o = reinterpret_cast<float2*>(dst) + idx;
    if (accumulate) { v.x += o->x; v.y += o->y; }
    *o = v;
  }
}

__global__ __launch_bounds__(256) void up2_acc_kernel(const float* src, float* dst, int B, int Hs, int Ws, int C, float scale, int accumulate) {
  const int cv = C >> 1;   // float2 granularity so that C == 2 works too
  const int H = Hs << 1, W = Ws << 1;
  const long long total = (long long)B * H * W * cv;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cc = (int)(idx % cv); long long pix = idx / cv;
    const int w = (int)(pix % W); pix /= W; const int h = (int)(pix % H); const int b = (int)(pix / H);
    const float2 s = *reinterpret_cast<const float2*>(src + ((((long long)b * Hs) + (h >> 1)) * Ws + (w >> 1)) * C + cc * 2);
    float2* o = reinterpret_cast<float2*>(dst + idx * 2);
    float2 v = make_float2(s.x * scale, s.y * scale);
    if (accumulate) { v.x += o->x; v.y += o->y; }
    *o = v;
  }
}

// one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* S, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* r = S + (long long)row * cols;
  float m = -INFINITY;
  for (int j = lane; j < cols; j += 64) m = fmaxf(m, r[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) { const float e = expf(r[j] - m); r[j] = e; s += e; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float inv = 1.f / s;
  for (int j = lane; j < cols; j += 64) r[j] *= inv;
}

__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* P, float* dP, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = P + (long long)row * cols;
  float* d = dP + (long long)row * cols;
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) s += p[j] * d[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  for (int j = lane; j < cols; j += 64) d[j] = p[j] * (d[j] - s);
}

// dst[b][j][i] = src[b][i][j], n x n per batch entry, 32 x 32 tiles through LDS (both sides coalesced); n % 32 == 0
__global__ __launch_bounds__(256) void transpose_sq_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  __shared__ float tile[32][33];
  const long long base = (long long)blockIdx.z * n * n;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[ty + 8 * r][tx] = src[base + (long long)(i0 + ty + 8 * r) * n + j0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[base + (long long)(j0 + ty + 8 * r) * n + i0 + tx] = tile[tx][ty + 8 * r];
}

// y[b][n] = sum_k act(x[b][k]) W[n][k] + bias[n]; one wave per output
__global__ __launch_bounds__(256) void linear_kernel(const float* x, const float* Wt, const float* bias, float* y, int B, int K, int N, int silu_in) {
  const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= (long long)B * N) return;
  const int b = (int)(o / N), n = (int)(o % N);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    float v = x[(long long)b * K + k];
    if (silu_in) v = silu_f(v);
    s += v * Wt[(long long)n * K + k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) y[o] = s + (bias ? bias[n] : 0.f);
}

__global__ void fourier_kernel(const float* cnoise, const float* Wf, float* out, int B, int nf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nf) return;
  const int b = i / nf, j = i % nf;
  const float pr = cnoise[b] * Wf[j] * 2.f * 3.14159265358979323846f;   // reference layerspp.py:40 (fp32 product order)
  out[(long long)b * 2 * nf + j] = sinf(pr);
  out[(long long)b * 2 * nf + nf + j] = cosf(pr);
}

__global__ __launch_bounds__(256) void mix2_kernel(const float* x, const float* w, const float* b, float* y, long long npix, int transpose, int accumulate) {
  const float w00 = w[0], w01 = w[1], w10 = w[2], w11 = w[3];
  const float b0 = b ? b[0] : 0.f, b1 = b ? b[1] : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    float2 o;
    if (!transpose) { o.x = w00 * v.x + w01 * v.y + b0; o.y = w10 * v.x + w11 * v.y + b1; }
    else { o.x = w00 * v.x + w10 * v.y; o.y = w01 * v.x + w11 * v.y; }
    float2* d = reinterpret_cast<float2*>(y) + i;
    if (accumulate) { o.x += d->x; o.y += d->y; }
    *d = o;
  }
}

// ------------------------------------------------------------------ 2-channel direct convs
// x [B][H][W][2] -> y [.., Cout]; w [Cout][TAPS][2]; thread = (pixel, 4 output channels)
template <int TAPS>
__global__ __launch_bounds__(256) void conv_c2in_kernel(const float* x, const float* w, const float* bias, const float* add, int add_ld,
                                                        float* y, int ldY, int B, int H, int W, int Cout, int accumulate) {
  const int q = Cout >> 2;
  const long long total = (long long)B * H * W * q;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int quad = (int)(idx % q); long long pix = idx / q;
    const long long p = pix;
    const int wq = (int)(pix % W); pix /= W; const int h = (int)(pix % H);
    const int c = quad * 4;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bias ? bias[c + j] : 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int dy = (TAPS == 9) ? t / 3 - 1 : 0, dx = (TAPS == 9) ? t % 3 - 1 : 0;
      if ((unsigned)(h + dy) >= (unsigned)H || (unsigned)(wq + dx) >= (unsigned)W) continue;
      const float2 v = reinterpret_cast<const float2*>(x)[p + dy * W + dx];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2 ww = *reinterpret_cast<const float2*>(w + ((long long)(c + j) * TAPS + t) * 2);
        acc[j] += ww.x * v.x + ww.y * v.y;
      }
    }
    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (add) r = add4(r, ld4(add + p * add_ld + c));
    float* o = y + p * ldY + c;
    if (accumulate) r = add4(r, ld4(o));
    st4(o, r);
  }
}

// Same op for Cout/4 dividing 256: a thread keeps its channel quad for the whole grid-stride loop, so its TAPS x 4 x 2 weights live
// in registers (the generic kernel re-loads them per pixel: 36 weight loads per 16-byte store).
template <int TAPS>
__global__ __launch_bounds__(256) void conv_c2in_reg_kernel(const float* x, const float* w, const float* bias, const float* add, int add_ld,
                                                            float* y, int ldY, int B, int H, int W, int Cout, int accumulate) {
  const int q = Cout >> 2;
  const int quad = threadIdx.x % q, c = quad * 4;
  float2 wr[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[t][j] = *reinterpret_cast<const float2*>(w + ((long long)(c + j) * TAPS + t) * 2);
  float b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = bias ? bias[c + j] : 0.f;
  const int ppb = 256 / q;                                  // pixels per block iteration
  const long long npix = (long long)B * H * W;
  for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / q; p < npix; p += (long long)gridDim.x * ppb) {
    const int wq = (int)(p % W); const int h = (int)((p / W) % H);
    float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int dy = (TAPS == 9) ? t / 3 - 1 : 0, dx = (TAPS == 9) ? t % 3 - 1 : 0;
      if ((unsigned)(h + dy) >= (unsigned)H || (unsigned)(wq + dx) >= (unsigned)W) continue;
      const float2 v = reinterpret_cast<const float2*>(x)[p + dy * W + dx];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += wr[t][j].x * v.x + wr[t][j].y * v.y;
    }
    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (add) r = add4(r, ld4(add + p * add_ld + c));
    float* o = y + p * ldY + c;
    if (accumulate) r = add4(r, ld4(o));
    st4(o, r);
  }
}

// 3x3 form of the same op with FOUR consecutive pixels (along W) per thread: the 3 x 6 input neighbourhood is loaded once (18 float2 loads for four
// outputs instead of 36) and the row / column predicates and address arithmetic are shared (the one-pixel form spends more issue slots on those than
// on its 72 FMAs).  W % 4 == 0.
__global__ __launch_bounds__(256) void conv_c2in_reg4_kernel(const float* x, const float* w, const float* bias, const float* add, int add_ld,
                                                             float* y, int ldY, int B, int H, int W, int Cout, int accumulate) {
  const int q = Cout >> 2;
  const int quad = threadIdx.x % q, c = quad * 4;
  float2 wr[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[t][j] = *reinterpret_cast<const float2*>(w + ((long long)(c + j) * 9 + t) * 2);
  float b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = bias ? bias[c + j] : 0.f;
  const int ppb = 256 / q;                                  // pixel groups per block iteration
  const int W4 = W >> 2;
  const long long ngrp = (long long)B * H * W4;
  for (long long gi = (long long)blockIdx.x * ppb + threadIdx.x / q; gi < ngrp; gi += (long long)gridDim.x * ppb) {
    const int wg = (int)(gi % W4) * 4; const int h = (int)((gi / W4) % H);
    const long long p0 = (gi / W4) * W + wg;                 // pixel index of the first of the four outputs
    float2 in[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool rok = (unsigned)(h + r - 1) < (unsigned)H;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        const int wx = wg + cc - 1;
        in[r][cc] = (rok && (unsigned)wx < (unsigned)W) ? reinterpret_cast<const float2*>(x)[p0 + (long long)(r - 1) * W + cc - 1] : make_float2(0.f, 0.f);
      }
    }
    float4 ad[4], pr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ad[u] = add ? ld4(add + (p0 + u) * add_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      pr[u] = accumulate ? ld4(y + (p0 + u) * ldY + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
      for (int t = 0; t < 9; ++t) {                          // tap order as in the one-pixel kernel (out-of-image taps contribute exact zeros)
        const float2 v = in[t / 3][u + t % 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += wr[t][j].x * v.x + wr[t][j].y * v.y;
      }
      float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (add) r = add4(r, ad[u]);
      if (accumulate) r = add4(r, pr[u]);
      st4(y + (p0 + u) * ldY + c, r);
    }
  }
}

// x [B][H][W][Cin] (ldX) -> y [B][H][W][2]; w [TAPS][Cin][2]; LPP = Cin/4 lanes cooperate on one pixel
template <int TAPS>
__global__ __launch_bounds__(256) void conv_c2out_kernel(const float* x, int ldX, const float* w, const float* bias, const float* up_add,
                                                         float* y, int B, int H, int W, int Cin, int accumulate) {
  const int lpp = Cin >> 2, ppb = 256 / lpp;
  const int tid = threadIdx.x, sub = tid % lpp, slot = tid / lpp;
  const int c = sub * 4;
  float wr[TAPS][8];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[t][j] = w[((long long)t * Cin + c) * 2 + j];
  const long long npix = (long long)B * H * W;
  const long long ngroups = (npix + ppb - 1) / ppb;
  for (long long gidx = blockIdx.x; gidx < ngroups; gidx += gridDim.x) {
    const long long p = gidx * ppb + slot;
    float s0 = 0.f, s1 = 0.f;
    int b = 0, h = 0, wq = 0;
    if (p < npix) {
      long long pix = p;
      wq = (int)(pix % W); pix /= W; h = (int)(pix % H); b = (int)(pix / H);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = (TAPS == 9) ? t / 3 - 1 : 0, dx = (TAPS == 9) ? t % 3 - 1 : 0;
        if ((unsigned)(h + dy) >= (unsigned)H || (unsigned)(wq + dx) >= (unsigned)W) continue;
        const float4 v = ld4(x + (p + dy * W + dx) * ldX + c);
        s0 += v.x * wr[t][0] + v.y * wr[t][2] + v.z * wr[t][4] + v.w * wr[t][6];
        s1 += v.x * wr[t][1] + v.y * wr[t][3] + v.z * wr[t][5] + v.w * wr[t][7];
      }
    }
    for (int off = lpp >> 1; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (sub == 0 && p < npix) {
      if (bias) { s0 += bias[0]; s1 += bias[1]; }
      if (up_add) {
        const float2 u = reinterpret_cast<const float2*>(up_add)[(((long long)b * (H >> 1)) + (h >> 1)) * (W >> 1) + (wq >> 1)];
        s0 += u.x; s1 += u.y;
      }
      float2* o = reinterpret_cast<float2*>(y) + p;
      if (accumulate) { s0 += o->x; s1 += o->y; }
      *o = make_float2(s0, s1);
    }
  }
}

// 3x3, Cin -> 2, LDS-tiled: a block owns an 8 x 32 tile of output pixels (one per thread).  Per 32-channel chunk the (10 x 34)-pixel halo is
// staged once (coalesced 128 B per pixel; the generic kernel re-reads every input element nine times through L1/L2), then each thread sums its
// 9 taps x 32 channels from LDS; the weights are wave-uniform (scalar loads).  Channel order of the sum: chunk, tap, channel.
constexpr int C2O_TH = 8, C2O_TW = 32, C2O_CK = 32, C2O_PITCH = C2O_CK + 4;
__global__ __launch_bounds__(256) void conv_c2out_tiled_kernel(const float* __restrict__ x, int ldX, const float* __restrict__ w, const float* bias,
                                                               const float* up_add, float* y, int B, int H, int W, int Cin, int accumulate) {
  __shared__ __attribute__((aligned(16))) float tile[(C2O_TH + 2) * (C2O_TW + 2) * C2O_PITCH];
  __shared__ __attribute__((aligned(16))) float wsh[9 * C2O_CK * 2];     // this chunk's weights: [tap][channel][2] (was 576 scalar loads per thread and chunk)
  const int tid = threadIdx.x;
  const int tx = tid & (C2O_TW - 1), ty = tid / C2O_TW;
  const int nbx = (W + C2O_TW - 1) / C2O_TW, nby = (H + C2O_TH - 1) / C2O_TH;
  int blk = blockIdx.x;
  const int bxi = blk % nbx; blk /= nbx;
  const int byi = blk % nby; const int b = blk / nby;
  const int h0 = byi * C2O_TH, w0 = bxi * C2O_TW;
  const int h = h0 + ty, wq = w0 + tx;
  float s0 = 0.f, s1 = 0.f;
  constexpr int NPX = (C2O_TH + 2) * (C2O_TW + 2);        // 340 halo pixels, 8 float4 each
  constexpr int NLD = (NPX * (C2O_CK / 4) + 255) / 256;    // 11 float4 per thread
  for (int c0 = 0; c0 < Cin; c0 += C2O_CK) {
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {                        // all global loads of the slab in flight before the first LDS store
      const int e = tid + 256 * i;
      const int px = e >> 3, c4 = (e & 7) * 4;
      const int hr = px / (C2O_TW + 2), wc = px - hr * (C2O_TW + 2);
      const int gh = h0 - 1 + hr, gw = w0 - 1 + wc;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NPX * (C2O_CK / 4) && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) v[i] = ld4(x + (((long long)b * H + gh) * W + gw) * ldX + c0 + c4);
    }
    float wv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + 256 * i; wv[i] = e < 9 * C2O_CK * 2 ? w[((long long)(e / (C2O_CK * 2)) * Cin + c0) * 2 + e % (C2O_CK * 2)] : 0.f; }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i;
      if (e < NPX * (C2O_CK / 4)) *reinterpret_cast<float4*>(tile + (e >> 3) * C2O_PITCH + (e & 7) * 4) = v[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + 256 * i; if (e < 9 * C2O_CK * 2) wsh[e] = wv[i]; }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* src = tile + ((ty + t / 3) * (C2O_TW + 2) + tx + t % 3) * C2O_PITCH;
      const float* wt = wsh + t * C2O_CK * 2;
#pragma unroll
      for (int c4 = 0; c4 < C2O_CK; c4 += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(src + c4);
        const float4 wa = *reinterpret_cast<const float4*>(wt + c4 * 2), wb = *reinterpret_cast<const float4*>(wt + c4 * 2 + 4);
        s0 += vv.x * wa.x + vv.y * wa.z + vv.z * wb.x + vv.w * wb.z;
        s1 += vv.x * wa.y + vv.y * wa.w + vv.z * wb.y + vv.w * wb.w;
      }
    }
    __syncthreads();
  }
  if (h < H && wq < W) {
    const long long p = ((long long)b * H + h) * W + wq;
    if (bias) { s0 += bias[0]; s1 += bias[1]; }
    if (up_add) {
      const float2 u = reinterpret_cast<const float2*>(up_add)[(((long long)b * (H >> 1)) + (h >> 1)) * (W >> 1) + (wq >> 1)];
      s0 += u.x; s1 += u.y;
    }
    float2* o = reinterpret_cast<float2*>(y) + p;
    if (accumulate) { s0 += o->x; s1 += o->y; }
    *o = make_float2(s0, s1);
  }
}

// Strip form (round 6).  The kernel above is LDS-bound: every thread reads 9 taps x 8 quads of halo values AND the 18 weight quads per tap row -- 864 reads of
// 16 bytes per pixel, 14.5 GB per level-0 launch against the LDS's 78 TB/s = 0.19 of the measured 0.22 ms (HBM: 0.71 GB).  Here a thread owns ONE channel quad
// of the chunk and a strip of 8 pixels of one row: its 6 weight quads of a tap row sit in registers while the row's halo values pass (they were re-read per pixel), a halo value is read
// once and feeds the up to three pixels it is a tap of (30 reads per strip instead of 72): 48 instead of 1728 reads of 16 bytes per thread and chunk for the same
// multiply-adds.  The eight quads of a pixel live in eight neighbouring lanes: one butterfly (3 steps) at the end, lane q of the group stores pixel q of the strip.
// Order of a pixel's sum: per quad (chunk, halo row, halo column, tap), then the butterfly over the quads.
__global__ __launch_bounds__(256, 3) void conv_c2out_strip_kernel(const float* __restrict__ x, int ldX, const float* __restrict__ w, const float* bias,
                                                               const float* up_add, float* y, int B, int H, int W, int Cin, int accumulate) {
  __shared__ __attribute__((aligned(16))) float tile[(C2O_TH + 2) * (C2O_TW + 2) * C2O_PITCH];
  __shared__ __attribute__((aligned(16))) float wsh[9 * C2O_CK * 2];
  const int tid = threadIdx.x;
  const int q = tid & 7, strip = tid >> 3, ry = strip >> 2, sx = (strip & 3) * 8;
  const int nbx = (W + C2O_TW - 1) / C2O_TW, nby = (H + C2O_TH - 1) / C2O_TH;
  int blk = blockIdx.x;
  const int bxi = blk % nbx; blk /= nbx;
  const int byi = blk % nby; const int b = blk / nby;
  const int h0 = byi * C2O_TH, w0 = bxi * C2O_TW;
  float s[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i][0] = 0.f; s[i][1] = 0.f; }
  constexpr int NPX = (C2O_TH + 2) * (C2O_TW + 2);
  constexpr int NLD = (NPX * (C2O_CK / 4) + 255) / 256;
  for (int c0 = 0; c0 < Cin; c0 += C2O_CK) {
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i;
      const int px = e >> 3, c4 = (e & 7) * 4;
      const int hr = px / (C2O_TW + 2), wc = px - hr * (C2O_TW + 2);
      const int gh = h0 - 1 + hr, gw = w0 - 1 + wc;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NPX * (C2O_CK / 4) && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) v[i] = ld4(x + (((long long)b * H + gh) * W + gw) * ldX + c0 + c4);
    }
    float wv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + 256 * i; wv[i] = e < 9 * C2O_CK * 2 ? w[((long long)(e / (C2O_CK * 2)) * Cin + c0) * 2 + e % (C2O_CK * 2)] : 0.f; }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i;
      if (e < NPX * (C2O_CK / 4)) *reinterpret_cast<float4*>(tile + (e >> 3) * C2O_PITCH + (e & 7) * 4) = v[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int e = tid + 256 * i; if (e < 9 * C2O_CK * 2) wsh[e] = wv[i]; }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
      float4 wa[3], wb[3];                                    // this quad's weights of tap row r: [dx] (4 channels x 2 outputs, interleaved)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        wa[dx] = *reinterpret_cast<const float4*>(wsh + (r * 3 + dx) * C2O_CK * 2 + q * 8);
        wb[dx] = *reinterpret_cast<const float4*>(wsh + (r * 3 + dx) * C2O_CK * 2 + q * 8 + 4);
      }
      const float* row = tile + ((ry + r) * (C2O_TW + 2) + sx) * C2O_PITCH + q * 4;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float4 vv = *reinterpret_cast<const float4*>(row + j * C2O_PITCH);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int px = j - dx;
          if (px >= 0 && px < 8) {
            s[px][0] += vv.x * wa[dx].x + vv.y * wa[dx].z + vv.z * wb[dx].x + vv.w * wb[dx].z;
            s[px][1] += vv.x * wa[dx].y + vv.y * wa[dx].w + vv.z * wb[dx].y + vv.w * wb[dx].w;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);                      // one tap row at a time: with all 30 halo reads hoisted the kernel needs 230 VGPRs
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1)
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i][0] += __shfl_xor(s[i][0], o); s[i][1] += __shfl_xor(s[i][1], o); }
  float s0 = s[0][0], s1 = s[0][1];
#pragma unroll
  for (int i = 1; i < 8; ++i) if (q == i) { s0 = s[i][0]; s1 = s[i][1]; }
  const int h = h0 + ry, wq = w0 + sx + q;
  if (h < H && wq < W) {
    const long long p = ((long long)b * H + h) * W + wq;
    if (bias) { s0 += bias[0]; s1 += bias[1]; }
    if (up_add) {
      const float2 u = reinterpret_cast<const float2*>(up_add)[(((long long)b * (H >> 1)) + (h >> 1)) * (W >> 1) + (wq >> 1)];
      s0 += u.x; s1 += u.y;
    }
    float2* o = reinterpret_cast<float2*>(y) + p;
    if (accumulate) { s0 += o->x; s1 += o->y; }
    *o = make_float2(s0, s1);
  }
}

// ------------------------------------------------------------------ STFT glue
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* x, float* xp, int B, int L, int pad, int Lp, float scale, const float* scale_b) {
  const long long total = (long long)B * Lp;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int b = (int)(idx / Lp), j = (int)(idx % Lp);
    float v = 0.f;
    if (j < L + 2 * pad) {
      int i = j - pad;
      if (i < 0) i = -i;
      if (i >= L) i = 2 * (L - 1) - i;
      v = x[(long long)b * L + i] * scale * (scale_b ? scale_b[b] : 1.f);
    }
    xp[idx] = v;
  }
}

__global__ __launch_bounds__(256) void ola_kernel(const float* frames, int ldF, int Tp, int n_fft, int hop, const float* inv_env, float* y, int B,
                                                  int L, int pad, const float* xin, const float* cskip_b, const float* cout_b) {
  const long long total = (long long)B * L;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int b = (int)(idx / L), s = (int)(idx % L);
    const int j = s + pad;
    int t0 = (j - n_fft + hop) / hop; if (t0 < 0) t0 = 0;     // ceil((j - n_fft + 1)/hop) for j-n_fft+1 > 0
    int t1 = j / hop; if (t1 > Tp - 1) t1 = Tp - 1;
    float acc = 0.f;
    for (int t = t0; t <= t1; ++t) {
      const int n = j - t * hop;
      if (n >= 0 && n < n_fft) acc += frames[((long long)b * Tp + t) * ldF + n];
    }
    acc *= inv_env[j];
    if (xin) acc = cskip_b[b] * xin[idx] + cout_b[b] * acc;
    y[idx] = acc;
  }
}

__global__ __launch_bounds__(256) void ola_adj_kernel(const float* g, int B, int L, int pad, int Tp, int n_fft, int hop, const float* inv_env,
                                                      const float* cout_b, float* frames, int ldF) {
  const long long total = (long long)B * Tp * ldF;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int n = (int)(idx % ldF); const long long bt = idx / ldF;
    const int t = (int)(bt % Tp), b = (int)(bt / Tp);
    float v = 0.f;
    if (n < n_fft) {
      const int j = t * hop + n, s = j - pad;
      if (s >= 0 && s < L) v = g[(long long)b * L + s] * inv_env[j] * (cout_b ? cout_b[b] : 1.f);
    }
    frames[idx] = v;
  }
}

__global__ __launch_bounds__(256) void unpad_adj_kernel(const float* dframes, int ldF, int T, int n_fft, int hop, int B, int L, int pad, float scale,
                                                        const float* scale_b, const float* g_out, const float* cskip_b, float* dx) {
  const long long total = (long long)B * L;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int b = (int)(idx / L), s = (int)(idx % L);
    auto U = [&](int j) {
      int t0 = (j - n_fft + hop) / hop; if (t0 < 0) t0 = 0;
      int t1 = j / hop; if (t1 > T - 1) t1 = T - 1;
      float a = 0.f;
      for (int t = t0; t <= t1; ++t) {
        const int n = j - t * hop;
        if (n >= 0 && n < n_fft) a += dframes[((long long)b * T + t) * ldF + n];
      }
      return a;
    };
    float v = U(s + pad);
    if (s >= 1 && s <= pad) v += U(pad - s);
    if (s >= L - 1 - pad && s <= L - 2) v += U(pad + 2 * (L - 1) - s);
    v *= scale * (scale_b ? scale_b[b] : 1.f);
    if (g_out) v += cskip_b[b] * g_out[idx];
    dx[idx] = v;
  }
}

inline int grid_for(long long total) {
  long long g = (total + 255) / 256;
  const long long cap = 1LL << cur_opt().ew_grid;            // option ew_grid (default 16: 58.2 -> 58.0 ms/step A/B, results identical): 12 = rounds 1-5 (4096 workgroups, long grid-stride loops)
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


// ------------------------------------------------------------------ FIR resampling (fir=True): upfirdn2d with the (1,3,3,1) kernel
// reference networks/ncsnpp_utils/up_or_down_sampling.py:195-257 (upsample_2d / downsample_2d, factor 2, zero padding) -> op/upfirdn2d_kernel.cu.
// For k = (1,3,3,1) the separable taps are, per axis:  up:   y[2i] = (x[i-1] + 3 x[i]) / 4,  y[2i+1] = (3 x[i] + x[i+1]) / 4
//                                                      down: y[i]  = (x[2i-1] + 3 x[2i] + 3 x[2i+1] + x[2i+2]) / 8       (x = 0 outside)
// and the two are adjoint up to a factor: up^T = 4 down, down^T = up / 4 (so the VJP needs no third kernel).  NHWC, any C.
__global__ __launch_bounds__(256) void fir_up2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, float scale, int accumulate) {
  const long long total = (long long)B * 2 * H * 2 * W * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C); long long r = idx / C;
    const int ox = (int)(r % (2 * W)); r /= 2 * W;
    const int oy = (int)(r % (2 * H)); const int b = (int)(r / (2 * H));
    const int iy = oy >> 1, ix = ox >> 1;
    const int y0 = (oy & 1) ? iy : iy - 1, x0 = (ox & 1) ? ix : ix - 1;          // the two contributing rows / columns are (y0, y0 + 1)
    const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int yy = y0 + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const float wy = dy ? 1.f - wy0 : wy0;
      float row = 0.f;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xx = x0 + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        row += (dx ? 1.f - wx0 : wx0) * x[(((long long)b * H + yy) * W + xx) * C + c];
      }
      acc += wy * row;
    }
    acc *= scale;
    y[idx] = accumulate ? y[idx] + acc : acc;
  }
}
// (H, W) -> (H/2, W/2)
__global__ __launch_bounds__(256) void fir_down2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, float scale, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C); long long r = idx / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
    const float wt[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const int yy = 2 * oy - 1 + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      float row = 0.f;
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int xx = 2 * ox - 1 + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        row += wt[dx] * x[(((long long)b * H + yy) * W + xx) * C + c];
      }
      acc += wt[dy] * row;
    }
    acc *= scale;
    y[idx] = accumulate ? y[idx] + acc : acc;
  }
}
}  // namespace

// ================================================================== launchers
int gn_num_chunks(int HW) {
  int c = HW / 64;
  if (c < 1) c = 1;
  if (c > 256) c = 256;
  return c;
}

static GnFast gn_fast(int HW, int C, int G) {
  GnFast a; a.HW = HW; a.C = C; a.G = G; a.q = C / 4; a.pl = 256 / a.q; if (a.pl < 1) a.pl = 1;
  // pixels per thread = option gn_trips (default 4 since round 6: short-lived workgroups, like the one-shot copy of the streaming ubench: 4.34 -> 4.54 TB/s for the
  // GroupNorm group; 32 = rounds 2-5: ~4096 workgroups over the batch at the big layers, 8 trips of 4 pixels per thread).  Elementwise kernels: results identical.
  const int trips = cur_opt().gn_trips;
  int chunks = HW / (a.pl * trips); if (chunks < 1) chunks = 1; if (chunks > (trips >= 32 ? 512 : 32768)) chunks = trips >= 32 ? 512 : 32768;
  a.ppc = (HW + chunks - 1) / chunks;
  a.ppc = (a.ppc + a.pl - 1) / a.pl * a.pl;
  return a;
}

static RedArgs make_red(Src2 x, int B, int H, int W, int C, int G, double* partial) {
  RedArgs a{};
  a.x = x; a.B = B; a.H = H; a.W = W; a.C = C; a.G = G;
  a.chunks = gn_num_chunks(H * W);
  a.ppc = (H * W + a.chunks - 1) / a.chunks;
  a.partial = partial;
  return a;
}

void launch_gn_stats(Src2 x, int B, int HW, int C, int G, float eps, double* partial, float* stats, hipStream_t st) {
  RedArgs a = make_red(x, B, 1, HW, C, G, partial);
  prof_hbm_begin(4.0 * B * HW * C, st);                                   // one read of x
  hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(a.chunks, B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(group_finalize_kernel<0>, dim3(G, B), dim3(256), 0, st, (const double*)partial, stats, C, G, a.chunks, HW, eps);
  prof_hbm_end(st);
}

void launch_csum_collapse(const double* partial, int chunks, int B, int C, double* csum, hipStream_t st) {
  hipLaunchKernelGGL(csum_collapse_kernel, dim3((C + 15) / 16, B), dim3(256), 0, st, partial, chunks, C, csum);
}
void launch_chan_sums(const float* x, int B, int HW, int C, double* partial, double* csum, hipStream_t st) {
  Src2 s; s.p0 = x; s.p1 = nullptr; s.C0 = C; s.ld0 = C; s.ld1 = 0;
  RedArgs a = make_red(s, B, 1, HW, C, 1, partial);
  prof_hbm_begin(4.0 * B * HW * C, st);                                   // one read of x
  hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(a.chunks, B), dim3(256), 0, st, a);
  launch_csum_collapse(partial, a.chunks, B, C, csum, st);
  prof_hbm_end(st);
}
void launch_gn_stats_partial(const double* partial, int chunks, int B, int HW, int C, int G, float eps, float* stats, hipStream_t st) {
  hipLaunchKernelGGL(group_finalize_kernel<0>, dim3(G, B), dim3(256), 0, st, partial, stats, C, G, chunks, HW, eps);
}
void launch_gn_stats_csum(const double* csum0, const double* csum1, int C0, int B, int HW, int C, int G, float eps, float* stats, hipStream_t st) {
  hipLaunchKernelGGL(group_finalize_csum_kernel, dim3(G, B), dim3(64), 0, st, csum0, csum1, csum1 ? C0 : C, C, G, HW, eps, stats);
}

void launch_gn_apply(Src2 x, const float* stats, const float* gamma, const float* beta, int B, int H, int W, int C, int G, int mode, int silu,
                     float* out, float* pooled_raw, hipStream_t st) {
  const long long total = (long long)B * (mode == 1 ? (H / 2) * (W / 2) : H * W) * (C / 4);
  const double n_in = (double)B * H * W * C, n_out = mode == 1 ? n_in / 4 : (mode == 2 ? n_in * 4 : n_in);
  prof_hbm_begin(4.0 * (n_in + n_out + (pooled_raw ? n_out : 0.0)), st);   // read x, write the (resampled) activation
  const bool fast = cur_opt().gn_fast != 0;
  if (fast && mode == 0 && C % 4 == 0 && C / 4 <= 256) {
    GnFast a = gn_fast(H * W, C, G);
    hipLaunchKernelGGL(gn_apply_m0_kernel, dim3((H * W + a.ppc - 1) / a.ppc, B), dim3(a.q * a.pl), 0, st, x, stats, gamma, beta, a, silu, out);
  } else {
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, stats, gamma, beta, B, H, W, C, G, mode, silu, out, pooled_raw);
  }
  prof_hbm_end(st);
}

// GroupNorm backward in two callable halves: the two per-group means of the backward (`red`), and the apply pass that needs them
void launch_gn_bwd_sums(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C, int G, int mode,
                        int silu, double* partial, float* red, hipStream_t st, int ready_chunks) {
  if (ready_chunks > 0) {                                    // the partials came with da (data-gradient epilogue): no pass over (x, da)
    hipLaunchKernelGGL(group_finalize_kernel<1>, dim3(G, B), dim3(256), 0, st, (const double*)partial, red, C, G, ready_chunks, H * W, 0.f);
    return;
  }
  RedArgs a = make_red(x, B, H, W, C, G, partial);
  a.stats = stats; a.gamma = gamma; a.beta = beta; a.da = da; a.mode = mode; a.silu = silu;
  const double n_in = (double)B * H * W * C, n_da = mode == 1 ? n_in / 4 : (mode == 2 ? n_in * 4 : n_in);
  prof_hbm_begin(4.0 * (n_in + n_da), st);                   // reads x and da
  hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(a.chunks, B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(group_finalize_kernel<1>, dim3(G, B), dim3(256), 0, st, (const double*)partial, red, C, G, a.chunks, H * W, 0.f);
  prof_hbm_end(st);
}
void launch_gn_bwd_apply(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C, int G, int mode,
                         int silu, const float* extra, int extra_mode, float extra_scale, const float* red, Dst2 dx, hipStream_t st) {
  const double n_in = (double)B * H * W * C, n_da = mode == 1 ? n_in / 4 : (mode == 2 ? n_in * 4 : n_in);
  // reads x and da again, writes dx (+ reads the extra gradient)
  prof_hbm_begin(4.0 * (n_in + n_da + n_in + (extra_mode ? (extra_mode == 2 ? n_in / 4 : n_in) : 0.0)), st);
  const long long total = (long long)B * H * W * (C / 4);
  const bool fast = cur_opt().gn_fast != 0;
  if (fast && mode == 0 && extra_mode != 2 && C % 4 == 0 && C / 4 <= 256) {
    GnFast g = gn_fast(H * W, C, G);
    const bool ex = extra_mode == 1 && extra != nullptr, acc = dx.acc0 != 0 || (dx.p1 != nullptr && dx.acc1 != 0);
    const dim3 grid((H * W + g.ppc - 1) / g.ppc, B), block(g.q * g.pl);
#define GN_M0(E, A) hipLaunchKernelGGL((gn_bwd_apply_m0_kernel<E, A>), grid, block, 0, st, x, stats, gamma, beta, da, g, silu, ex ? extra : nullptr, extra_scale, red, dx)
    if (ex && acc) GN_M0(true, true); else if (ex) GN_M0(true, false); else if (acc) GN_M0(false, true); else GN_M0(false, false);
#undef GN_M0
  } else
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, stats, gamma, beta, da, B, H, W, C, G, mode, silu, extra,
                     extra_mode, extra_scale, red, dx);
  prof_hbm_end(st);
}
void launch_gn_bwd(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C, int G, int mode,
                   int silu, const float* extra, int extra_mode, float extra_scale, double* partial, float* red, Dst2 dx, hipStream_t st,
                   int ready_chunks) {
  launch_gn_bwd_sums(x, stats, gamma, beta, da, B, H, W, C, G, mode, silu, partial, red, st, ready_chunks);
  launch_gn_bwd_apply(x, stats, gamma, beta, da, B, H, W, C, G, mode, silu, extra, extra_mode, extra_scale, red, dx, st);
}

void launch_axpy(float* dst, const float* src, float alpha, long long n, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, dst, src, alpha, n / 4, accumulate);
}

void launch_fir_up2(const float* x, float* y, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(fir_up2_kernel, dim3(grid_for((long long)B * 4 * H * W * C)), dim3(256), 0, st, x, y, B, H, W, C, scale, accumulate);
}
void launch_fir_down2(const float* x, float* y, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(fir_down2_kernel, dim3(grid_for((long long)B * (H / 2) * (W / 2) * C)), dim3(256), 0, st, x, y, B, H, W, C, scale, accumulate);
}

void launch_pool2(const float* src, float* dst, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st) {
  if (C == 2) {
    hipLaunchKernelGGL(pool2_c2_kernel, dim3(grid_for((long long)B * (H / 2) * (W / 2))), dim3(256), 0, st, src, dst, B, H, W, scale, accumulate);
  } else {
    hipLaunchKernelGGL(pool2_kernel, dim3(grid_for((long long)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, st, src, dst, B, H, W, C, scale, accumulate);
  }
}

void launch_up2_acc(const float* src, float* dst, int B, int Hs, int Ws, int C, float scale, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(up2_acc_kernel, dim3(grid_for((long long)B * Hs * Ws * 4 * (C / 2))), dim3(256), 0, st, src, dst, B, Hs, Ws, C, scale, accumulate);
}

void launch_transpose_sq(const float* src, float* dst, int batch, int n, hipStream_t st) {
  hipLaunchKernelGGL(transpose_sq_kernel, dim3(n / 32, n / 32, batch), dim3(256), 0, st, src, dst, n);
}
void launch_softmax_rows(float* S, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, S, rows, cols);
}

void launch_softmax_bwd_rows(const float* P, float* dP, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, P, dP, rows, cols);
}

void launch_linear(const float* x, const float* W, const float* b, float* y, int B, int K, int N, int silu_in, hipStream_t st) {
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv((long long)B * N, 4)), dim3(256), 0, st, x, W, b, y, B, K, N, silu_in);
}

void launch_fourier(const float* cnoise, const float* Wf, float* out, int B, int nf, hipStream_t st) {
  hipLaunchKernelGGL(fourier_kernel, dim3(cdiv(B * nf, 256)), dim3(256), 0, st, cnoise, Wf, out, B, nf);
}

void launch_mix2(const float* x, const float* w, const float* b, float* y, long long npix, int transpose, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(mix2_kernel, dim3(grid_for(npix)), dim3(256), 0, st, x, w, b, y, npix, transpose, accumulate);
}

void launch_conv_c2in(const float* x, const float* w, const float* bias, const float* add, int add_ld, float* y, int ldY, int B, int H, int W,
                      int Cout, int taps, int accumulate, hipStream_t st) {
  const long long total = (long long)B * H * W * (Cout / 4);
  if (Cout % 4 == 0 && 256 % (Cout / 4) == 0) {
    const int ppb = 256 / (Cout / 4);
    long long g = ((long long)B * H * W + ppb - 1) / ppb;
    if (g > 256 * 8) g = 256 * 8;
    const bool four = cur_opt().c2in4 != 0;
    if (taps == 9 && four && W % 4 == 0) {
      long long g4 = ((long long)B * H * (W / 4) + ppb - 1) / ppb;
      if (g4 > 256 * 16) g4 = 256 * 16;
      hipLaunchKernelGGL(conv_c2in_reg4_kernel, dim3((int)g4), dim3(256), 0, st, x, w, bias, add, add_ld, y, ldY, B, H, W, Cout, accumulate);
    } else if (taps == 9)
      hipLaunchKernelGGL(conv_c2in_reg_kernel<9>, dim3((int)g), dim3(256), 0, st, x, w, bias, add, add_ld, y, ldY, B, H, W, Cout, accumulate);
    else
      hipLaunchKernelGGL(conv_c2in_reg_kernel<1>, dim3((int)g), dim3(256), 0, st, x, w, bias, add, add_ld, y, ldY, B, H, W, Cout, accumulate);
    return;
  }
  if (taps == 9)
    hipLaunchKernelGGL(conv_c2in_kernel<9>, dim3(grid_for(total)), dim3(256), 0, st, x, w, bias, add, add_ld, y, ldY, B, H, W, Cout, accumulate);
  else
    hipLaunchKernelGGL(conv_c2in_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, x, w, bias, add, add_ld, y, ldY, B, H, W, Cout, accumulate);
}

void launch_conv_c2out(const float* x, int ldX, const float* w, const float* bias, const float* up_add, float* y, int B, int H, int W, int Cin,
                       int taps, int accumulate, hipStream_t st) {
  const int ppb = 256 / (Cin / 4);
  long long groups = ((long long)B * H * W + ppb - 1) / ppb;
  int grid = (int)(groups < 256 * 8 ? groups : 256 * 8);
  const bool tiled = cur_opt().c2out_tiled != 0;
  if (taps == 9 && cur_opt().c2out_tiled == 2 && Cin % C2O_CK == 0 && ldX % 4 == 0) {
    const int blocks = B * ((H + C2O_TH - 1) / C2O_TH) * ((W + C2O_TW - 1) / C2O_TW);
    hipLaunchKernelGGL(conv_c2out_strip_kernel, dim3(blocks), dim3(256), 0, st, x, ldX, w, bias, up_add, y, B, H, W, Cin, accumulate);
  } else if (taps == 9 && tiled && Cin % C2O_CK == 0 && ldX % 4 == 0) {
    const int blocks = B * ((H + C2O_TH - 1) / C2O_TH) * ((W + C2O_TW - 1) / C2O_TW);
    hipLaunchKernelGGL(conv_c2out_tiled_kernel, dim3(blocks), dim3(256), 0, st, x, ldX, w, bias, up_add, y, B, H, W, Cin, accumulate);
  } else if (taps == 9)
    hipLaunchKernelGGL(conv_c2out_kernel<9>, dim3(grid), dim3(256), 0, st, x, ldX, w, bias, up_add, y, B, H, W, Cin, accumulate);
  else
    hipLaunchKernelGGL(conv_c2out_kernel<1>, dim3(grid), dim3(256), 0, st, x, ldX, w, bias, up_add, y, B, H, W, Cin, accumulate);
}

void launch_reflect_pad(const float* x, float* xp, int B, int L, int pad, int Lp, float scale, const float* scale_b, hipStream_t st) {
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((long long)B * Lp)), dim3(256), 0, st, x, xp, B, L, pad, Lp, scale, scale_b);
}

void launch_ola(const float* frames, int ldF, int Tp, int n_fft, int hop, const float* inv_env, float* y, int B, int L, int pad, const float* xin,
                const float* cskip_b, const float* cout_b, hipStream_t st) {
  hipLaunchKernelGGL(ola_kernel, dim3(grid_for((long long)B * L)), dim3(256), 0, st, frames, ldF, Tp, n_fft, hop, inv_env, y, B, L, pad, xin, cskip_b, cout_b);
}

void launch_ola_adj(const float* g, int B, int L, int pad, int Tp, int n_fft, int hop, const float* inv_env, const float* cout_b, float* frames,
                    int ldF, hipStream_t st) {
  hipLaunchKernelGGL(ola_adj_kernel, dim3(grid_for((long long)B * Tp * ldF)), dim3(256), 0, st, g, B, L, pad, Tp, n_fft, hop, inv_env, cout_b, frames, ldF);
}

void launch_unpad_adj(const float* dframes, int ldF, int T, int n_fft, int hop, int B, int L, int pad, float scale, const float* scale_b,
                      const float* g_out, const float* cskip_b, float* dx, hipStream_t st) {
  hipLaunchKernelGGL(unpad_adj_kernel, dim3(grid_for((long long)B * L)), dim3(256), 0, st, dframes, ldF, T, n_fft, hop, B, L, pad, scale, scale_b, g_out, cskip_b, dx);
}

}  // namespace buddy
