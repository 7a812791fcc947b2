// Winograd F(4x4, 3x3) convolution in three passes (gfx950, fp32, NHWC, stride 1, pad 1) -- for the large 3x3 layers, where the fused
// F(2x2,3x3) kernel (wino.hip) is limited by its LDS traffic per MFMA and not by the matrix pipe.  Same call sites and epilogue semantics as
// igemm_kernel<9> / wino3_kernel (reference ddpm_conv3x3, networks/ncsnpp_utils/layers.py:119-126, and its data-gradient):
//   1. input transform   V[pos][tile][cin]  = (B^T d B)[pos]      one thread per (6x6 input tile, 4 channels), HBM-bound: 1 read, 2.25 writes
//   2. 36 batched GEMMs  M[pos][tile][cout] = V[pos] x U[pos]^T    igemm_kernel<1> with batch = 36 (fp32 MFMA): 2.25 multiply-adds per output
//                                                                  and (cin, cout) pair instead of 9 (direct) or 4 (F(2x2,3x3))
//   3. output transform  Y = A^T M A + bias / time-embedding bias / residual / scale / accumulate, float4 per lane along cout
// Transform constants are the standard interpolation points {0, +-1, +-2, inf}; everything stays fp32 (round-off grows by about one decimal
// digit over the direct form: unit test tolerance 1e-4 instead of 2e-5).
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

// one 6-vector of the input transform: t = B^T d
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&t)[6]) {
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = d[3] + d[4] - 4.f * (d[1] + d[2]);
  t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  t[3] = 2.f * (d[3] - d[1]) + d[4] - d[2];
  t[4] = 2.f * (d[1] - d[3]) + d[4] - d[2];
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// one 6-vector of the output transform: y = A^T m (4 outputs)
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&y)[4]) {
  const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y[0] = m[0] + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m[5];
}

// SiLU on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each): the input transform evaluates it 2.25x per input value (tile halo),
// and with the exact expf + IEEE division it became ALU-bound (measured 301 us per launch; 212 us with these, 160 us without the GroupNorm --
// still cheaper than the 94 us apply pass + its 2 x tensor traffic that it replaces; a two-channels-per-thread form with 5 waves per SIMD
// instead of 3 was slower, 238 us: profiles/README.md r02c)
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }

// thread = (tile, channel quad); V[(pos * Mt + tile) * Cin + c]
// tile0 / Mc: this launch covers tiles [tile0, tile0 + Mc) and V holds only that chunk (whole tensor: tile0 = 0, Mc = all tiles)
// GN: the convolution's input is act(GroupNorm(x)) of a (channel-concatenated) view x (reference layerspp.py:243-245, 257-258): normalise and
// activate on the fly while loading the tile, so the activated tensor never exists in HBM (the zero padding applies to the ACTIVATED tensor)
template <bool GN>
__global__ __launch_bounds__(256) void w4_input_kernel(const float* __restrict__ x, int ldX, const W4Gn gn, float* __restrict__ V, int B, int H, int W,
                                                       int Cin, long long tile0, long long Mc) {
  const int q = Cin >> 2, TH = H >> 2, TW = W >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Mc * q) return;
  const int c = (int)(idx % q) * 4;
  const long long ltile = idx / q, tile = tile0 + ltile;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
  const int gy0 = 4 * ty - 1, gx0 = 4 * tx - 1;
  float mean = 0.f, rstd = 0.f;
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f), bt = gm;
  if (GN) {
    const int g = c / (Cin / gn.G);
    mean = gn.stats[((long long)b * gn.G + g) * 2]; rstd = gn.stats[((long long)b * gn.G + g) * 2 + 1];
    gm = ld4(gn.gamma + c); bt = ld4(gn.beta + c);
    const bool second = gn.x.p1 != nullptr && c >= gn.x.C0;
    x = second ? gn.x.p1 + (c - gn.x.C0) : gn.x.p0 + c;
    ldX = second ? gn.x.ld1 : gn.x.ld0;
  } else {
    x += c;
  }
  float4 d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      const int gy = gy0 + r, gx = gx0 + cc;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        float4 v = ld4(x + (((long long)b * H + gy) * W + gx) * ldX);
        if (GN) {
          v = make_float4((v.x - mean) * rstd * gm.x + bt.x, (v.y - mean) * rstd * gm.y + bt.y, (v.z - mean) * rstd * gm.z + bt.z,
                          (v.w - mean) * rstd * gm.w + bt.w);
          if (gn.silu) v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
        }
        d[r][cc] = v;
      } else {
        d[r][cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
  for (int cc = 0; cc < 6; ++cc) {                          // columns: t[:, cc] = B^T d[:, cc]
    float4 col[6], t[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = d[r][cc];
    bt6(col, t);
#pragma unroll
    for (int r = 0; r < 6; ++r) d[r][cc] = t[r];
  }
  float* out = V + ltile * Cin + c;
  const long long ps = Mc * Cin;
#pragma unroll
  for (int r = 0; r < 6; ++r) {                             // rows: v[r, :] = t[r, :] B
    float4 t[6];
    bt6(d[r], t);
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) st4(out + (long long)(r * 6 + cc) * ps, t[cc]);
  }
}

// thread = (tile, cout quad); Mb[(pos * Mt + tile) * N + n]
// STAT: also leave the per-(utterance, channel) sum and sum of squares of the values written (fp64), one partial per workgroup, for the
// GroupNorm that consumes this tensor (reference layerspp.py:243, 257): stat[((b * chunks + chunk) * N + n) * 2 + {0, 1}], chunks = tiles per
// utterance / (256 / q).  Requires 256 % q == 0 and tiles per utterance % (256 / q) == 0 (wino4_stat_chunks): a workgroup then holds whole
// tiles of one utterance and every thread is live.
template <bool STAT>
__global__ __launch_bounds__(256) void w4_output_kernel(const float* __restrict__ Mb, const IgemmParams p, int B, long long tile0, long long Mc,
                                                        double* __restrict__ stat) {
  const int H = p.H, W = p.W, N = p.N;
  const int q = N >> 2, TH = H >> 2, TW = W >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (!STAT && idx >= Mc * q) return;
  double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
  const int n = (int)(idx % q) * 4;
  const long long ltile = idx / q, tile = tile0 + ltile;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
  const float* src = Mb + ltile * N + n;
  const long long ps = Mc * N;
  float4 s[4][6];                                          // s = A^T m  (4 x 6)
#pragma unroll
  for (int cc = 0; cc < 6; ++cc) {
    float4 col[6], y[4];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = ld4(src + (long long)(r * 6 + cc) * ps);
    at6(col, y);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r][cc] = y[r];
  }
  float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias_n) add = ld4(p.bias_n + n);
  if (p.bias_bn) add = add + ld4(p.bias_bn + (long long)b * p.ld_bias_bn + n);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float4 y[4];
    at6(s[r], y);
    const int hh = 4 * ty + r;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int ww = 4 * tx + cc;
      const long long pix = ((long long)b * H + hh) * W + ww;
      float4 v = p.alpha * y[cc] + add;
      if (p.res_mode == 1) v = v + ld4(p.res + pix * p.ldRes + n);
      else if (p.res_mode == 2) v = v + ld4(p.res + (((long long)b * (H >> 1) + (hh >> 1)) * (W >> 1) + (ww >> 1)) * p.ldRes + n);
      v = p.out_scale * v;
      float* dst = p.C + pix * p.ldC + n;
      if (p.accumulate) v = v + ld4(dst);
      st4(dst, v);
      if (STAT) {
        ssum[0] += (double)v.x; ssum[1] += (double)v.y; ssum[2] += (double)v.z; ssum[3] += (double)v.w;
        ssq[0] += (double)v.x * (double)v.x; ssq[1] += (double)v.y * (double)v.y; ssq[2] += (double)v.z * (double)v.z;
        ssq[3] += (double)v.w * (double)v.w;
      }
    }
  }
  if (STAT) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x, pl = 256 / q;
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = ssum[j]; red[tid * 8 + 4 + j] = ssq[j]; }
    __syncthreads();
    if (tid < q) {
      double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < pl; ++l)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += red[(l * q + tid) * 8 + j];
      const long long tpb = (long long)TW * TH, first = (long long)blockIdx.x * pl;      // first tile of this workgroup (tile0 == 0)
      const long long bb = first / tpb, chunk = (first % tpb) / pl, chunks = tpb / pl;
      double* o = stat + ((bb * chunks + chunk) * N + tid * 4) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j * 2] = r[j]; o[j * 2 + 1] = r[4 + j]; }
    }
  }
}
}  // namespace

// partial sums per utterance the STAT epilogue writes (0: the shape does not allow it)
int wino4_stat_chunks(const IgemmParams& p) {
  const int q = p.N / 4;
  if (q < 1 || q > 256 || 256 % q) return 0;
  const long long tpb = (long long)(p.H / 4) * (p.W / 4);
  const int pl = 256 / q;
  return tpb % pl ? 0 : (int)(tpb / pl);
}

bool wino4_supported(const IgemmParams& p) {
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return p.H % 4 == 0 && p.W % 4 == 0 && p.Cin % 4 == 0 && p.N % 4 == 0 && p.A1 == nullptr && p.bias_m == nullptr && p.ldA0 % 4 == 0 &&
         p.ldC % 4 == 0 && (p.res_mode == 0 || p.ldRes % 4 == 0) && (p.bias_bn == nullptr || p.ld_bias_bn % 4 == 0) && al16(p.A0) && al16(p.C) &&
         al16(p.res) && al16(p.bias_n) && al16(p.bias_bn) && (long long)p.M / 16 * 36 < (1LL << 31);
}
// floats of scratch for the transformed input (V) and the transformed output (M)
void wino4_scratch(const IgemmParams& p, long long* v_floats, long long* m_floats) {
  const long long Mt = (long long)p.M / 16;
  *v_floats = 36 * Mt * p.Cin; *m_floats = 36 * Mt * p.N;
}

void launch_wino4(const IgemmParams& p, const float* U4, float* V, float* Mb, hipStream_t st, const W4Gn* gn, double* stat, const void* U4x) {
  const int B = p.M / (p.H * p.W);
  const long long Mt = (long long)p.M / 16;
  const int plevel = igemm_prof_level();
  const bool prof = plevel >= 2;                            // the caller brackets the three passes as ONE 3x3 convolution; passes timed here
  const bool prof_gemm = plevel == 1;                       // level 1: only the batched GEMM is bracketed
  // the passes take a tile range [t0, t0 + Mc): one range = the whole tensor.  (Running them chunk by chunk so that V / M stay in the 256 MB
  // Infinity Cache was measured and lost at every chunk size: +3...+27 % per convolution, profiles/README.md r02.)
  const long long chunk = Mt;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  if (prof) { for (auto& e : ev) (void)hipEventCreate(&e); (void)hipEventRecord(ev[0], st); }
  if (prof_gemm) { (void)hipEventCreate(&ev[1]); (void)hipEventCreate(&ev[2]); }
  for (long long t0 = 0; t0 < Mt; t0 += chunk) {
    const long long Mc = std::min(chunk, Mt - t0);
    const dim3 gi((unsigned)((Mc * (p.Cin / 4) + 255) / 256));
    if (gn) hipLaunchKernelGGL(w4_input_kernel<true>, gi, dim3(256), 0, st, (const float*)nullptr, 0, *gn, V, B, p.H, p.W, p.Cin, t0, Mc);
    else hipLaunchKernelGGL(w4_input_kernel<false>, gi, dim3(256), 0, st, p.A0, p.ldA0, W4Gn{}, V, B, p.H, p.W, p.Cin, t0, Mc);
    if (prof || prof_gemm) (void)hipEventRecord(ev[1], st);
    IgemmParams g; std::memset(&g, 0, sizeof(g));
    g.A0 = V; g.ldA0 = p.Cin; g.sA = Mc * p.Cin; g.Cin = p.Cin;
    g.Bt = U4; g.ldB = p.Cin; g.sB = (long long)p.N * p.Cin;
    g.C = Mb; g.ldC = p.N; g.sC = Mc * p.N;
    g.M = (int)Mc; g.N = p.N; g.H = 1; g.W = 1; g.rows_per_batch = 1; g.alpha = 1.f; g.out_scale = 1.f;
    g.tag = 36;
    igemm_prof_enable(0);
    if (U4x != nullptr && wgemm_supported(p.N, p.Cin)) launch_wgemm_bf16x3(V, U4x, Mb, Mc, p.N, p.Cin, 36, st);
    else launch_igemm(g, 1, false, false, 36, st);
    igemm_prof_enable(plevel);
    if (prof || prof_gemm) (void)hipEventRecord(ev[2], st);
    const dim3 go((unsigned)((Mc * (p.N / 4) + 255) / 256));
    if (stat) hipLaunchKernelGGL(w4_output_kernel<true>, go, dim3(256), 0, st, (const float*)Mb, p, B, t0, Mc, stat);
    else hipLaunchKernelGGL(w4_output_kernel<false>, go, dim3(256), 0, st, (const float*)Mb, p, B, t0, Mc, (double*)nullptr);
  }
  if (prof_gemm) {
    const double mt = (double)Mt;
    prof_w4_push(nullptr, ev[1], ev[2], nullptr, 2.0 * 36.0 * mt * p.Cin * p.N, 0.0, 0.0, 4.0 * 36.0 * (mt * p.Cin + mt * p.N + (double)p.N * p.Cin));
  }
  if (prof) {
    (void)hipEventRecord(ev[3], st);
    const double mt = (double)Mt, m = (double)p.M;
    prof_w4_push(ev[0], ev[1], ev[2], ev[3], 2.0 * 36.0 * mt * p.Cin * p.N, 4.0 * (m * p.Cin + 36.0 * mt * p.Cin),
                 4.0 * (36.0 * mt * p.N + m * p.N * (p.res_mode ? 2.0 : 1.0)), 4.0 * 36.0 * (mt * p.Cin + mt * p.N + (double)p.N * p.Cin));
  }
}

// host: U4[pos][cout][cin] = (G g G^T)[pos] from tap-major packed weights wt[cout][(dy*3+dx)*Cin + cin]
void wino4_transform_weights(const float* wt, int Cout, int Cin, float* U) {
  static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      double g[3][3], t[6][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = wt[(size_t)o * 9 * Cin + (size_t)(a * 3 + b) * Cin + i];
      for (int xi = 0; xi < 6; ++xi) for (int b = 0; b < 3; ++b) t[xi][b] = G[xi][0] * g[0][b] + G[xi][1] * g[1][b] + G[xi][2] * g[2][b];
      for (int xi = 0; xi < 6; ++xi) for (int nu = 0; nu < 6; ++nu) {
        const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
        U[((size_t)(xi * 6 + nu) * Cout + o) * Cin + i] = (float)u;
      }
    }
}

}  // namespace buddy
