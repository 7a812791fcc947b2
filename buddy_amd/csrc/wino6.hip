// Winograd F(6x6, 3x3) convolution in three passes (gfx950, fp32, NHWC, stride 1, pad 1) -- the large layers of the score network.
// Same call sites, epilogue semantics and GroupNorm fusions as the F(4x4,3x3) form (wino4.hip; reference ddpm_conv3x3,
// networks/ncsnpp_utils/layers.py:119-126, and its data-gradient), with 64 / 36 = 1.78 multiply-adds per output and (cin, cout) pair instead
// of 2.25, and transformed tensors V / M that are 1.78x the activation instead of 2.25x:
//   1. input transform   V[pos][tile][cin]  = (B^T d B)[pos]      8x8 input patch per 6x6 output tile
//   2. 64 batched GEMMs  M[pos][tile][cout] = V[pos] x U[pos]^T    igemm_kernel<1> with batch = 64 (fp32 MFMA)
//   3. output transform  Y = A^T M A + bias / time-embedding bias / residual / scale / accumulate
// Interpolation points {0, +-1, +-2, +-1/2, inf}.  Round-off (fp32 throughout) is about twice that of F(4x4,3x3): 1.0e-5 against 5.0e-6 of
// the abs-max on unit-variance data with 128 input channels (direct form 3e-7); unit test tolerance 1e-4 like F(4x4,3x3).
// H and W need not be multiples of 6: tiles overhang the image, overhanging inputs read as zero padding, overhanging outputs are not
// written -- the caller uses this form where the overhang costs less than the transform saves (wino6_pays).
//
// A thread cannot hold an 8x8 patch of float4 (256 VGPRs), so both transforms are separable passes through LDS: in the first phase a
// thread owns one COLUMN of one tile for one channel quad (8 float4), in the second one ROW.  Lanes run along the channel quads, so every
// global access is a run of 16-byte pieces (512 B for 128 channels) and the LDS image [row][col][quad] is conflict free.  A workgroup of
// 256 threads = 8 columns x QC channel quads x (32 / QC) tiles, QC <= 32.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace buddy {
namespace {
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }   // as wino4.hip
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// t = B^T d (8 -> 8)
__device__ __forceinline__ void bt8(const float4 (&d)[8], float4 (&t)[8]) {
  const float4 e0 = d[2] + d[6] - 4.25f * d[4], o0 = d[1] + d[5] - 4.25f * d[3];
  const float4 e1 = d[6] + 0.25f * d[2] - 1.25f * d[4], o1 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
  const float4 e2 = d[6] + 4.f * d[2] - 5.f * d[4], o2 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
  t[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
  t[1] = e0 + o0; t[2] = e0 - o0;
  t[3] = e1 + o1; t[4] = e1 - o1;
  t[5] = e2 + o2; t[6] = e2 - o2;
  t[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
}
// y = A^T m (8 -> 6)
__device__ __forceinline__ void at8(const float4 (&m)[8], float4 (&y)[6]) {
  const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
  y[0] = m[0] + s12 + s34 + s56;
  y[1] = d12 + 2.f * d34 + 0.5f * d56;
  y[2] = s12 + 4.f * s34 + 0.25f * s56;
  y[3] = d12 + 8.f * d34 + 0.125f * d56;
  y[4] = s12 + 16.f * s34 + 0.0625f * s56;
  y[5] = d12 + 32.f * d34 + 0.03125f * d56 + m[7];
}

// y = A^T m (8 -> 7): F(7x7,2x2) on the same points / the same B^T (the sub-pixel up forms: every phase of conv3x3(upsample x2) is a 2x2-tap
// correlation on the low-resolution grid)
__device__ __forceinline__ void at8(const float4 (&m)[8], float4 (&y)[7]) {
  const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
  y[0] = m[0] + s12 + s34 + s56;
  y[1] = d12 + 2.f * d34 + 0.5f * d56;
  y[2] = s12 + 4.f * s34 + 0.25f * s56;
  y[3] = d12 + 8.f * d34 + 0.125f * d56;
  y[4] = s12 + 16.f * s34 + 0.0625f * s56;
  y[5] = d12 + 32.f * d34 + 0.03125f * d56;
  y[6] = s12 + 64.f * s34 + 0.015625f * s56 + m[7];
}

struct W6Geo { int B, H, W, TH, TW, QC, TPB; long long Mt; int xcd; int Cl; };   // TH x TW tiles per utterance, Mt = B * TH * TW; Cl: see S2D / UP

// grid (ceil(Mt / TPB), ceil(q / QC)); V[(pos * Mt + tile) * Cin + c]
// GN 1: the input is act(GroupNorm(x)) of a (channel-concatenated) view, applied while loading (zero padding applies to the ACTIVATED tensor)
// GN 2: the input is the GroupNorm backward of the gradient gn.da: rstd * (dxhat - m1 - xhat * m2), dxhat = da * act'(z) * gamma (the tensor the
//       separate apply pass would write and this transform read back)
// TS: tile stride = outputs per tile and axis: 6 = F(6x6,3x3); 7 = F(7x7,2x2) of the sub-pixel up forms (same 8x8 patch, same B^T).
// S2D (sub-pixel form of the data-gradient of conv3x3(nearest-upsample x2), launch_wino6 up = 2; TS = 7): the input is a (2H, 2W, Cl) tensor read as
//       its space-to-depth image (H, W, 4 Cl): channel c' = ph * Cl + c, ph = 2 py + px, is pixel (2 y + py, 2 x + px) of channel c; the patch of phase
//       (py, px) starts at (7 ty - py, 7 tx - px) (the even phase reads rows i, i + 1 of its image, the odd one i - 1, i).  Single source.
template <int GN, bool S2D = false, int TS = 6>
__global__ __launch_bounds__(256) void w6_input_kernel(const float* __restrict__ x, int ldX, const W4Gn gn, float* __restrict__ V, int Cin,
                                                       const W6Geo geo, unsigned* __restrict__ vmax) {
  __shared__ float4 lds[32 * 64];
  const int tid = threadIdx.x, QC = geo.QC;
  const int ql = tid % QC, col = (tid / QC) & 7, tl = tid / (QC * 8);
  // XCD-aware tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (own L2 each), which puts the two tiles that share
  // two of their eight patch columns -- and the tile row below, which shares two rows -- on different L2s: the 1.78x patch overlap was fetched
  // from the fabric almost in full (PMC r03: 2.09x the input instead of ~1.1x).  Each XCD gets a contiguous range of tiles instead.
  int bx = blockIdx.x, by = blockIdx.y;
  if (geo.xcd) {
    const int n = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = n >> 3, r8 = n & 7, xcd = lin & 7, k = lin >> 3;
    const int nl = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
    bx = nl % gridDim.x; by = nl / gridDim.x;
  }
  const int quad = by * QC + ql, c = quad * 4;
  const long long tile = (long long)bx * geo.TPB + tl;
  const bool live = tile < geo.Mt && c < Cin;
  const int H = geo.H, W = geo.W;
  int b = 0, ty = 0, tx = 0;
  if (live) { tx = (int)(tile % geo.TW); ty = (int)((tile / geo.TW) % geo.TH); b = (int)(tile / ((long long)geo.TW * geo.TH)); }
  if (live) {
    float mean = 0.f, rstd = 0.f, m1 = 0.f, m2 = 0.f;
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
    const float* da = nullptr;
    const int ph = S2D ? c / geo.Cl : 0, cl = S2D ? c - ph * geo.Cl : c, py = ph >> 1, px = ph & 1;   // cl: channel of the source tensor
    if (GN) {
      const int g = cl / ((S2D ? geo.Cl : Cin) / gn.G);
      mean = gn.stats[((long long)b * gn.G + g) * 2]; rstd = gn.stats[((long long)b * gn.G + g) * 2 + 1];
      if (GN == 2) { m1 = gn.red[((long long)b * gn.G + g) * 2]; m2 = gn.red[((long long)b * gn.G + g) * 2 + 1]; da = gn.da + cl; }
      gm = ld4(gn.gamma + cl); bt = ld4(gn.beta + cl);
      const bool second = !S2D && gn.x.p1 != nullptr && c >= gn.x.C0;
      x = second ? gn.x.p1 + (c - gn.x.C0) : gn.x.p0 + cl;
      ldX = second ? gn.x.ld1 : gn.x.ld0;
    } else {
      x += cl;
    }
    const int gx = TS * tx - (S2D ? px : 1) + col, gy0 = TS * ty - (S2D ? py : 1);
    float4 d[8], t[8];
    // All loads of the column first, UNCONDITIONAL, from coordinates clamped into the image; the zero padding is applied to the value afterwards.
    // With the load inside `if (inside)` next to the GroupNorm arithmetic the compiler kept every row's load -> wait -> SiLU chain to itself (10
    // loads, 18 vmcnt waits in the GN 1 instantiation against 8 / 1 in the plain one): eight load latencies one after the other per thread.
    const bool okx = (unsigned)gx < (unsigned)W;
    const int cx = min(max(gx, 0), W - 1);
    float4 g4[GN == 2 ? 8 : 1];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int cy = min(max(gy0 + r, 0), H - 1);
      const long long pix = S2D ? ((long long)b * (2 * H) + 2 * cy + py) * (2 * W) + 2 * cx + px : ((long long)b * H + cy) * W + cx;
      d[r] = ld4(x + pix * ldX);
      if (GN == 2) g4[r] = ld4(da + pix * gn.ldda);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool ok = okx && (unsigned)(gy0 + r) < (unsigned)H;
      float4 v = d[r];
      if (GN == 1) {
        v = make_float4((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y, (v.z - mean) * rstd * gm.z + bt.z,
                        (v.w - mean) * rstd * gm.w + bt.w);
        if (gn.silu) v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
      }
      if (GN == 2) {
        const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {g4[r].x, g4[r].y, g4[r].z, g4[r].w};
        const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mean) * rstd;
          const float dxh = dv[j] * (gn.silu ? dsilu_f(xh * gv[j] + bv[j]) : 1.f) * gv[j];
          o[j] = rstd * (dxh - m1 - xh * m2);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
      }
      d[r] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bt8(d, t);                                               // column: t[:, col] = B^T d[:, col]
#pragma unroll
    for (int r = 0; r < 8; ++r) lds[(tl * 64 + r * 8 + col) * QC + ql] = t[r];
  }
  __syncthreads();
  float vm = 0.f;
  if (live) {
    const int r = col;                                       // this thread's row in the second phase
    float4 d[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = lds[(tl * 64 + r * 8 + j) * QC + ql];
    bt8(d, t);                                               // row: v[r, :] = t[r, :] B
    float* out = V + tile * Cin + c;
    const long long ps = geo.Mt * Cin;
#pragma unroll
    for (int j = 0; j < 8; ++j) st4(out + (long long)(r * 8 + j) * ps, t[j]);
    if (vmax) {
#pragma unroll
      for (int j = 0; j < 8; ++j) vm = fmaxf(fmaxf(vm, fmaxf(fabsf(t[j].x), fabsf(t[j].y))), fmaxf(fabsf(t[j].z), fabsf(t[j].w)));
    }
  }
  // f16x2 GEMM (wgemm.hip): the abs-max of V per UTTERANCE, from which the GEMM derives its power-of-two operand scale.  Device-scope atomics execute at
  // the memory side on this part (~250 ns each, serialised per address: one atomic per (tile, row) group onto one word per utterance cost 0.95 ms per
  // launch, r05g), so: ONE atomic max per tile (workgroup-level reduction through LDS) onto one of VMAX_SUB partial words per utterance, each in its own
  // 128-byte line; the GEMM combines the partial words.  On the float's bit pattern: non-negative floats order like unsigned integers.
  if (vmax) {
    __shared__ float wmx[256];
    for (int o = QC >> 1; o > 0; o >>= 1) vm = fmaxf(vm, __shfl_xor(vm, o));
    if (ql == 0) wmx[tl * 8 + col] = vm;
    __syncthreads();
    if (tid < geo.TPB) {
      const long long tw = (long long)bx * geo.TPB + tid;
      if (tw < geo.Mt) {
        float m = wmx[tid * 8];
#pragma unroll
        for (int j = 1; j < 8; ++j) m = fmaxf(m, wmx[tid * 8 + j]);
        const int bw = (int)(tw / ((long long)geo.TW * geo.TH)), sub = (int)((tw + 13 * by) & (VMAX_SUB - 1));
        atomicMax(vmax + ((long long)bw * VMAX_SUB + sub) * VMAX_STRIDE, __float_as_uint(m));
      }
    }
  }
}

// grid (B * chunks, ceil(q / QC)): workgroup (b, chunk) walks tiles [chunk * TPB * TL, (chunk + 1) * TPB * TL) of utterance b, TPB at a time.
// STAT 1: per-(utterance, channel) partial (sum, sum of squares) of the values written, fp64, one per workgroup:
// stat[((b * chunks + chunk) * N + n) * 2 + {0, 1}] (the layout csum_collapse_kernel reads).
// STAT 2 (data-gradient convolutions): the value written is da, the gradient w.r.t. act(GroupNorm(x)) of the view bg.x; the partials are the two
// sums that GroupNorm's backward needs, (sum dxhat, sum dxhat * xhat) with dxhat = da * act'(z) * gamma -- x is read here at the output pixels and
// the reduction pass over (x, da) disappears.  fp32 over the (at most 36 x walk) pixels a thread sees, fp64 beyond.
// UP (sub-pixel form of conv3x3(nearest-upsample x2), launch_wino6 up = 1): M holds 4 N columns, column n' = ph * N + n = output channel n of phase
// ph = 2 py + px; tile pixel (y, x) of phase ph is pixel (2 y + py, 2 x + px) of the (2H, 2W, N) output (depth-to-space).  The statistics partials
// of phase ph are chunks [ph * chunks, (ph + 1) * chunks) of 4 * chunks per utterance.  No residual.  TS = 7: tile row r of phase py is row
// 7 ty + r - py of the phase image (the patch starts at low-resolution row 7 ty - 1 for both phases).
// NTM (A/B switch w6_nt): M is read with non-temporal loads -- every 16-byte piece is read exactly once, in whole 512-byte rows per instruction (the
// streaming ubench: reads 6.3 -> 7.1 TB/s with them, tools/hbm_bw.py; the GEMM's V rows, read in four partial-line pieces, lost 45 % with them).
typedef float f32x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) { const f32x4w v = __builtin_nontemporal_load(reinterpret_cast<const f32x4w*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
template <int STAT, bool UP = false, int TS = 6, bool NTM = false>
__global__ __launch_bounds__(256) void w6_output_kernel(const float* __restrict__ Mb, const IgemmParams p, const W6Geo geo, int chunks, int TL,
                                                        double* __restrict__ stat, const W4Gn bg) {
  __shared__ float4 lds[32 * TS * 8];
  __shared__ double red[STAT ? 256 * 8 : 1];
  const int tid = threadIdx.x, QC = geo.QC;
  const int ql = tid % QC, col = (tid / QC) & 7, tl = tid / (QC * 8);
  const int N = p.N, H = geo.H, W = geo.W;
  const int quad = blockIdx.y * QC + ql, nq = quad * 4;       // nq: column of M
  const int ph = UP ? nq / N : 0, n = UP ? nq - ph * N : nq, py = ph >> 1, px = ph & 1;
  const int NM = UP ? 4 * N : N;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  const int tpb = geo.TH * geo.TW;                            // tiles per utterance
  const bool chan = nq < NM;
  const long long ps = geo.Mt * NM;
  double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
  float fs[4] = {0.f, 0.f, 0.f, 0.f}, ft[4] = {0.f, 0.f, 0.f, 0.f};
  float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
  float mean = 0.f, rstd = 0.f;
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
  const float* xsrc = nullptr; int ldx = 0;
  if (chan) {
    if (p.bias_n) add = ld4(p.bias_n + n);
    if (p.bias_bn) add = add + ld4(p.bias_bn + (long long)b * p.ld_bias_bn + n);
    if (STAT == 2) {
      const int g = n / (N / bg.G);
      mean = bg.stats[((long long)b * bg.G + g) * 2]; rstd = bg.stats[((long long)b * bg.G + g) * 2 + 1];
      gm = ld4(bg.gamma + n); bt = ld4(bg.beta + n);
      const bool second = bg.x.p1 != nullptr && n >= bg.x.C0;
      xsrc = second ? bg.x.p1 + (n - bg.x.C0) : bg.x.p0 + n;
      ldx = second ? bg.x.ld1 : bg.x.ld0;
    }
  }
  // Every load above (bias, statistics, gamma / beta) is complete before the loop: otherwise the first use of `add` inside the loop carries a
  // vmcnt(0) on every pixel (the compiler cannot count memory operations across the back edge), which also waits for the previous pixel's store.
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), expcnt / lgkmcnt untouched
  for (int it = 0; it < TL; ++it) {
    const int lt = (chunk * TL + it) * geo.TPB + tl;          // tile within the utterance
    const bool live = chan && lt < tpb;
    const long long tile = (long long)b * tpb + lt;
    // STAT 2: this thread's row of x is requested now, together with the M loads of phase 1, not after the LDS exchange
    float4 xpre[TS];
    if (STAT == 2 && live && col < TS) {
      const int ty = lt / geo.TW, tx = lt - ty * geo.TW, hh = min(TS * ty + col, H - 1);
#pragma unroll
      for (int cc = 0; cc < TS; ++cc) {                       // unconditional (clamped pixel): only read where the output pixel exists
        const int ww = min(TS * tx + cc, W - 1);
        xpre[cc] = ld4(xsrc + (((long long)b * H + hh) * W + ww) * ldx);
      }
    }
    if (live) {
      const float* src = Mb + tile * NM + nq;
      float4 m[8], s[TS];
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = NTM ? ld4nt(src + (long long)(i * 8 + col) * ps) : ld4(src + (long long)(i * 8 + col) * ps);
      at8(m, s);                                             // column: s[:, col] = A^T m[:, col]
#pragma unroll
      for (int r = 0; r < TS; ++r) lds[(tl * (TS * 8) + r * 8 + col) * QC + ql] = s[r];
    }
    __syncthreads();
    if (live && col < TS) {
      const int r = col, ty = lt / geo.TW, tx = lt - ty * geo.TW;
      const int hh = TS * ty + r - (UP ? py : 0);
      float4 m[8], y[TS];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = lds[(tl * (TS * 8) + r * 8 + j) * QC + ql];
      at8(m, y);                                             // row: y[r, :] = s[r, :] A
      if ((unsigned)hh < (unsigned)H) {
#pragma unroll
        for (int cc = 0; cc < TS; ++cc) {
          const int ww = TS * tx + cc - (UP ? px : 0);
          if ((unsigned)ww < (unsigned)W) {
            const long long pix = UP ? ((long long)b * (2 * H) + 2 * hh + py) * (2 * W) + 2 * ww + px : ((long long)b * H + hh) * W + ww;
            float4 v = p.alpha * y[cc] + add;
            if (UP) {}
            else if (p.res_mode == 1) v = v + ld4(p.res + pix * p.ldRes + n);
            else if (p.res_mode == 2) v = v + ld4(p.res + (((long long)b * (H >> 1) + (hh >> 1)) * (W >> 1) + (ww >> 1)) * p.ldRes + n);
            v = p.out_scale * v;
            float* dst = p.C + pix * p.ldC + n;
            if (p.accumulate) v = v + ld4(dst);
            st4(dst, v);
            if (STAT == 2) {
              const float4 xv4 = xpre[cc];
              const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {v.x, v.y, v.z, v.w};
              const float gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - mean) * rstd;
                const float dxh = dv[j] * (bg.silu ? dsilu_f(xh * gv[j] + bv[j]) : 1.f) * gv[j];
                fs[j] += dxh; ft[j] += dxh * xh;
              }
            }
            if (STAT == 1) {
              ssum[0] += (double)v.x; ssum[1] += (double)v.y; ssum[2] += (double)v.z; ssum[3] += (double)v.w;
              ssq[0] += (double)v.x * (double)v.x; ssq[1] += (double)v.y * (double)v.y; ssq[2] += (double)v.z * (double)v.z;
              ssq[3] += (double)v.w * (double)v.w;
            }
          }
        }
      }
    }
    if (STAT == 2) {                                          // flush the fp32 strip sums of this tile row
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[j] += (double)fs[j]; ssq[j] += (double)ft[j]; fs[j] = 0.f; ft[j] = 0.f; }
    }
    __syncthreads();
  }
  if (STAT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = ssum[j]; red[tid * 8 + 4 + j] = ssq[j]; }
    __syncthreads();
    if (tid < QC && chan) {                                   // tid < QC: ql == tid, col == 0, tl == 0
      double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < 256 / QC; ++l)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += red[(l * QC + tid) * 8 + j];
      double* o = stat + ((UP ? ((long long)b * 4 + ph) * chunks + chunk : (long long)b * chunks + chunk) * N + n) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j * 2] = r[j]; o[j * 2 + 1] = r[4 + j]; }
    }
  }
}

// tiles per axis: F(6x6,3x3) ceil(n / 6); sub-pixel up form (F(7x7,2x2) tiles on the low-resolution grid, patch origin 7 t - 1): forward
// floor(n / 7) + 1 (the odd phase's last output row n - 1 is row n - 1 + 1 of the tiling), data-gradient ceil(n / 7)
int tiles_axis(int n, int up) { return up == 1 ? n / 7 + 1 : up == 2 ? (n + 6) / 7 : (n + 5) / 6; }
W6Geo geometry(const IgemmParams& p, int C, int up = 0) {
  W6Geo g;
  g.H = p.H; g.W = p.W; g.B = p.M / (p.H * p.W);
  g.TH = tiles_axis(p.H, up); g.TW = tiles_axis(p.W, up);
  g.Mt = (long long)g.B * g.TH * g.TW;
  const int q = C / 4;
  g.QC = q >= 32 ? 32 : (q > 16 ? 32 : (q > 8 ? 16 : (q > 4 ? 8 : (q > 2 ? 4 : (q > 1 ? 2 : 1)))));
  g.TPB = 32 / g.QC;
  const int xcd = cur_opt().w6_xcd != 0;     // A/B switch of the XCD-aware tile order of the input transform
  g.xcd = xcd;
  g.Cl = C;
  return g;
}
// iterations of TPB tiles a workgroup of the output transform walks: about 512 workgroups (= statistics partials) per utterance (the sub-pixel
// up form has four partials per workgroup row: 128)
int out_walk(const W6Geo& g, int up = 0) { const int t = g.TH * g.TW, per = g.TPB * (up == 1 ? 128 : 512); return std::max(1, (t + per - 1) / per); }
}  // namespace

bool wino6_supported(const IgemmParams& p) {
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return p.H >= 6 && p.W >= 6 && p.Cin % 4 == 0 && p.N % 4 == 0 && p.A1 == nullptr && p.bias_m == nullptr && p.ldA0 % 4 == 0 && p.ldC % 4 == 0 &&
         (p.res_mode == 0 || p.ldRes % 4 == 0) && (p.res_mode != 2 || (p.H % 2 == 0 && p.W % 2 == 0)) &&
         (p.bias_bn == nullptr || p.ld_bias_bn % 4 == 0) && al16(p.A0) && al16(p.C) && al16(p.res) && al16(p.bias_n) && al16(p.bias_bn) &&
         (long long)((p.H + 5) / 6) * ((p.W + 5) / 6) * (p.M / (p.H * p.W)) * 64 < (1LL << 31);
}
// worth it against F(4x4,3x3): executed multiply-adds 64 per 6x6 tile (overhang included) against 36 per 4x4 tile (at least 5 % fewer), and at least 64
// tiles per utterance.  Round 4: the thresholds were 10 % / 128 tiles, which left the bottleneck level of the shipped network (32 x 64 per 4 s utterance: 66
// tiles, 8 % fewer multiply-adds) on F(4x4,3x3); that path has no GroupNorm-backward fusions (sums in the output transform, apply in the input
// transform), so moving it here saves those passes too: 65.46 -> 64.98 ms/step A/B, one denoiser evaluation 117.0 / 112.7 -> 115.3 / 110.4 dB to float64.
bool wino6_pays(const IgemmParams& p) {
  const double tiles6 = (double)((p.H + 5) / 6) * ((p.W + 5) / 6), tiles4 = (double)p.H * p.W / 16.0;
  return tiles6 * 64.0 <= 0.95 * tiles4 * 36.0 && tiles6 >= 64;
}
void wino6_scratch(const IgemmParams& p, long long* v_floats, long long* m_floats, int up) {
  const long long Mt = (long long)(p.M / (p.H * p.W)) * tiles_axis(p.H, up) * tiles_axis(p.W, up);
  *v_floats = 64 * Mt * p.Cin * (up == 2 ? 4 : 1); *m_floats = 64 * Mt * p.N * (up == 1 ? 4 : 1);
}
int wino6_stat_chunks(const IgemmParams& p, int up) {
  if (p.N % 4) return 0;
  const W6Geo g = geometry(p, up == 1 ? 4 * p.N : p.N, up);
  const int per = g.TPB * out_walk(g, up);
  return (up == 1 ? 4 : 1) * ((g.TH * g.TW + per - 1) / per);
}
double wino6_exec_ratio(const IgemmParams& p, int up) {       // executed / direct-convolution multiply-adds (sub-pixel forms: 4 x 64 per tile, 4 M pixels)
  const W6Geo g = geometry(p, p.N, up);
  return 64.0 * (double)g.Mt / (9.0 * (double)p.M);
}

// up = 1: y (2H, 2W, N) = conv3x3(nearest-upsample x2 of the (H, W, Cin) input): ONE input transform at the low resolution, a GEMM with 4 N columns
//         (the four sub-pixel phases' 3x3 kernels, conv3_weight_prep kind 61: every output phase sees a 2x2 subset of the upsampled taps, summed
//         into a zero-padded 3x3 kernel on the low-resolution grid) and a depth-to-space output transform.  U6: [64][4 N][Cin].
// up = 2: its data-gradient: p.Cin = channels of the (2H, 2W) gradient, read space-to-depth as 4 Cin channels; output (H, W, N).  U6: [64][N][4 Cin].
// p.H, p.W, p.M describe the LOW resolution in both.
void launch_wino6(const IgemmParams& p, const float* U6, float* V, float* Mb, hipStream_t st, const W4Gn* gn, double* stat, const W4Gn* bwd_gn,
                  const void* U6x, int up, int xform, unsigned* vmax) {
  // xform: the arithmetic of the stage image U6x: 1 = bf16x3, 2 = f16x2 (then vmax: one zeroed slot per utterance for this launch's abs-max of V)
  if (xform != 2) vmax = nullptr;
  const int CinG = up == 2 ? 4 * p.Cin : p.Cin, NG = up == 1 ? 4 * p.N : p.N;       // K and N of the batched GEMM
  W6Geo gi = geometry(p, CinG, up), go = geometry(p, NG, up);
  gi.Cl = p.Cin; go.Cl = p.N;
  const long long Mt = gi.Mt;
  const int plevel = igemm_prof_level();
  const bool prof = plevel >= 2, prof_gemm = plevel == 1;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (prof) { for (auto& e : ev) (void)hipEventCreate(&e); (void)hipEventRecord(ev[0], st); }
  if (prof_gemm) { (void)hipEventCreate(&ev[1]); (void)hipEventCreate(&ev[2]); }
  const dim3 grid_in((unsigned)((Mt + gi.TPB - 1) / gi.TPB), (unsigned)((CinG / 4 + gi.QC - 1) / gi.QC));
  if (up == 2) {
    if (gn && gn->da) hipLaunchKernelGGL((w6_input_kernel<2, true, 7>), grid_in, dim3(256), 0, st, (const float*)nullptr, 0, *gn, V, CinG, gi, vmax);
    else hipLaunchKernelGGL((w6_input_kernel<0, true, 7>), grid_in, dim3(256), 0, st, p.A0, p.ldA0, W4Gn{}, V, CinG, gi, vmax);
  } else if (up == 1) {
    if (gn) hipLaunchKernelGGL((w6_input_kernel<1, false, 7>), grid_in, dim3(256), 0, st, (const float*)nullptr, 0, *gn, V, p.Cin, gi, vmax);
    else hipLaunchKernelGGL((w6_input_kernel<0, false, 7>), grid_in, dim3(256), 0, st, p.A0, p.ldA0, W4Gn{}, V, p.Cin, gi, vmax);
  }
  else if (gn && gn->da) hipLaunchKernelGGL(w6_input_kernel<2>, grid_in, dim3(256), 0, st, (const float*)nullptr, 0, *gn, V, p.Cin, gi, vmax);
  else if (gn) hipLaunchKernelGGL(w6_input_kernel<1>, grid_in, dim3(256), 0, st, (const float*)nullptr, 0, *gn, V, p.Cin, gi, vmax);
  else hipLaunchKernelGGL(w6_input_kernel<0>, grid_in, dim3(256), 0, st, p.A0, p.ldA0, W4Gn{}, V, p.Cin, gi, vmax);
  if (prof || prof_gemm) (void)hipEventRecord(ev[1], st);
  IgemmParams g; std::memset(&g, 0, sizeof(g));
  g.A0 = V; g.ldA0 = CinG; g.sA = Mt * CinG; g.Cin = CinG;
  g.Bt = U6; g.ldB = CinG; g.sB = (long long)NG * CinG;
  g.C = Mb; g.ldC = NG; g.sC = Mt * NG;
  g.M = (int)Mt; g.N = NG; g.H = 1; g.W = 1; g.rows_per_batch = 1; g.alpha = 1.f; g.out_scale = 1.f;
  g.tag = 36;                                                 // the Winograd-domain batched GEMM instantiation (36 or 64 positions)
  igemm_prof_enable(0);
  if (U6x != nullptr && xform == 2 && wgemm_supported(NG, CinG)) launch_wgemm_f16x2(V, U6x, Mb, Mt, NG, CinG, 64, vmax, gi.TH * gi.TW, st);
  else if (U6x != nullptr && wgemm_supported(NG, CinG)) launch_wgemm_bf16x3(V, U6x, Mb, Mt, NG, CinG, 64, st);
  else launch_igemm(g, 1, false, false, 64, st);
  igemm_prof_enable(plevel);
  if (prof || prof_gemm) (void)hipEventRecord(ev[2], st);
  const int TL = out_walk(go, up), per = go.TPB * TL, chunks = (go.TH * go.TW + per - 1) / per;
  const dim3 grid_out((unsigned)(go.B * chunks), (unsigned)((NG / 4 + go.QC - 1) / go.QC));
  if (up == 1 && cur_opt().w6_nt) {
    if (stat) hipLaunchKernelGGL((w6_output_kernel<1, true, 7, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, W4Gn{});
    else hipLaunchKernelGGL((w6_output_kernel<0, true, 7, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  } else if (up == 1) {
    if (stat) hipLaunchKernelGGL((w6_output_kernel<1, true, 7>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, W4Gn{});
    else hipLaunchKernelGGL((w6_output_kernel<0, true, 7>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  } else if (up == 2 && cur_opt().w6_nt) {
    if (stat && bwd_gn) hipLaunchKernelGGL((w6_output_kernel<2, false, 7, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, *bwd_gn);
    else hipLaunchKernelGGL((w6_output_kernel<0, false, 7, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  } else if (up == 2) {
    if (stat && bwd_gn) hipLaunchKernelGGL((w6_output_kernel<2, false, 7>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, *bwd_gn);
    else hipLaunchKernelGGL((w6_output_kernel<0, false, 7>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  }
  else if (cur_opt().w6_nt) {
    if (stat && bwd_gn) hipLaunchKernelGGL((w6_output_kernel<2, false, 6, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, *bwd_gn);
    else if (stat) hipLaunchKernelGGL((w6_output_kernel<1, false, 6, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, W4Gn{});
    else hipLaunchKernelGGL((w6_output_kernel<0, false, 6, true>), grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  }
  else if (stat && bwd_gn) hipLaunchKernelGGL(w6_output_kernel<2>, grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, *bwd_gn);
  else if (stat) hipLaunchKernelGGL(w6_output_kernel<1>, grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, stat, W4Gn{});
  else hipLaunchKernelGGL(w6_output_kernel<0>, grid_out, dim3(256), 0, st, (const float*)Mb, p, go, chunks, TL, (double*)nullptr, W4Gn{});
  const double mt = (double)Mt, m = (double)p.M;
  if (up) {                                                   // sub-pixel forms: the GEMM's own K / N; the (2H, 2W) side has 4 m pixels
    const double kg = CinG, ng = NG, gf = 2.0 * 64.0 * mt * kg * ng, gb = 4.0 * 64.0 * (mt * kg + mt * ng + ng * kg);
    if (prof_gemm) prof_w4_push(nullptr, ev[1], ev[2], nullptr, gf, 0.0, 0.0, gb);
    if (prof) {
      (void)hipEventRecord(ev[3], st);
      const double in_reads = (gn && gn->da) ? 2.0 : 1.0, out_extra = (stat && bwd_gn) ? 1.0 : 0.0;
      prof_w4_push(ev[0], ev[1], ev[2], ev[3], gf, 4.0 * (in_reads * m * kg + 64.0 * mt * kg), 4.0 * (64.0 * mt * ng + m * ng * (1.0 + out_extra)), gb);
    }
    return;
  }
  if (prof_gemm) prof_w4_push(nullptr, ev[1], ev[2], nullptr, 2.0 * 64.0 * mt * p.Cin * p.N, 0.0, 0.0, 4.0 * 64.0 * (mt * p.Cin + mt * p.N + (double)p.N * p.Cin));
  if (prof) {
    (void)hipEventRecord(ev[3], st);
    // algorithmic bytes of the transform passes incl. what the fused GroupNorm work reads: the backward apply reads x AND da (input side), the
    // backward-sum epilogue reads x at the output pixels
    const double in_reads = (gn && gn->da) ? 2.0 : 1.0, out_extra = (stat && bwd_gn) ? 1.0 : 0.0;
    prof_w4_push(ev[0], ev[1], ev[2], ev[3], 2.0 * 64.0 * mt * p.Cin * p.N, 4.0 * (in_reads * m * p.Cin + 64.0 * mt * p.Cin),
                 4.0 * (64.0 * mt * p.N + m * p.N * ((p.res_mode ? 2.0 : 1.0) + out_extra)), 4.0 * 64.0 * (mt * p.Cin + mt * p.N + (double)p.N * p.Cin));
  }
}

// host: U6[pos][cout][cin] = (G g G^T)[pos] from tap-major packed weights wt[cout][(dy*3+dx)*Cin + cin]
void wino6_transform_weights(const float* wt, int Cout, int Cin, float* U) {
  static const double G[8][3] = {{1, 0, 0}, {-2.0 / 9, -2.0 / 9, -2.0 / 9}, {-2.0 / 9, 2.0 / 9, -2.0 / 9}, {1.0 / 90, 1.0 / 45, 2.0 / 45},
                                 {1.0 / 90, -1.0 / 45, 2.0 / 45}, {32.0 / 45, 16.0 / 45, 8.0 / 45}, {32.0 / 45, -16.0 / 45, 8.0 / 45}, {0, 0, 1}};
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      double g[3][3], t[8][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = wt[(size_t)o * 9 * Cin + (size_t)(a * 3 + b) * Cin + i];
      for (int xi = 0; xi < 8; ++xi) for (int b = 0; b < 3; ++b) t[xi][b] = G[xi][0] * g[0][b] + G[xi][1] * g[1][b] + G[xi][2] * g[2][b];
      for (int xi = 0; xi < 8; ++xi) for (int nu = 0; nu < 8; ++nu) {
        const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
        U[((size_t)(xi * 8 + nu) * Cout + o) * Cin + i] = (float)u;
      }
    }
}

}  // namespace buddy
