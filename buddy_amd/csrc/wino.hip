// Fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (gfx950), NHWC, stride 1, pad 1.
//
// Replaces the same call sites as igemm_kernel<9> (reference ddpm_conv3x3, networks/ncsnpp_utils/layers.py:119-126, and
// its data-gradient) with 4*Cin instead of 9*Cin multiply-adds per output: Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A.
// Everything stays fp32 (v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain); only the summation order differs from the
// direct form (transform entries are 0, +-1, +-1/2).
//
// Mapping.  A wave owns 16 output tiles (2x2 pixels each) and ALL 16 transform positions for 32 output channels:
//   * lane l = (tile = l & 15, q = l >> 4) loads its tile's 4x4 input patch for channels 4q..4q+3 of the current 16-channel
//     chunk as sixteen float4 (coalesced 64 B per pixel across the four q), transforms it IN REGISTERS and thereby already
//     holds the MFMA A-operands of all 16 positions (A[row = tile][k = 4q + j], j = the float4 component = MFMA k-step): the
//     input never goes through LDS;
//   * the pre-transformed weights U[pos][cout][k] of the chunk (16 x 32 x 16 floats = 32 KiB) are staged through LDS once per
//     workgroup (4 waves = 64 tiles share them) and read as float4 B-fragments;
//   * 16 positions x 2 column tiles x 4 k-steps = 128 MFMAs (16x16x4, 32 cycles each) per chunk and wave = 4096 matrix cycles;
//   * the 16 position accumulators of one (tile, cout) sit in the same lane and register index, so the output transform
//     A^T M A is pure register arithmetic in the epilogue (no cross-lane traffic), fused with bias / time-embedding bias /
//     residual / 1/sqrt(2) like the direct kernel.
#include "common.h"
#include <cstdlib>
#include <cstdint>

namespace buddy {
namespace {
constexpr int WN = 32;      // output channels per workgroup
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---------------------------------------------------------------------------------------------------------------------
// v3: like v2 but 8-channel chunks and 4-wave workgroups (4 x 16 tiles), 56 KiB of LDS: TWO independent workgroups per CU, so one
// group's DMA-wait / patch-read / transform phase runs under the other group's MFMA phase (v2's single 8-wave group serialised
// the two phases: ablation 3.4 ms + 3.9 ms = 7.1 ms).  Lane (tile, q) owns channels 2q, 2q+1 of the chunk (float2 operands,
// two MFMA k-steps); weights are laid out U[Cin/8][16][Cout][8].
constexpr int BTX = 16, RC = 2 * BTX + 2;
constexpr int V3_WK = 8;
constexpr int V3_USZ = 16 * WN * V3_WK;      // floats per weight buffer

template <int ABL, int NW, int NBUF>
__global__ __launch_bounds__(64 * NW, 2) void wino3_kernel(const IgemmParams p, const float* __restrict__ Uw, const float* __restrict__ zeros) {
  constexpr int V3_BTY = NW, V3_RR = 2 * V3_BTY + 2, V3_RPIX = V3_RR * RC, V3_RPL = (V3_RPIX + 127) / 128 * 128;   // pixels covered by the load rounds
  // LDS images are built for ds_read_b64: a half-wave (16 tiles x 2 k-pairs) must cover 32 distinct 8-B slots of the 256-B bank row.
  // region: [k-pair q][pixel] float2 with a row pitch = 1 (mod 32) pixels (tile stride is 2 pixels = 16 B, the next k-pair fills the gaps);
  // weights: [pos][column tile][k-pair][16 couts] float2 (128 B per k-pair row)
  constexpr int V3_RP = ((V3_RPIX + 31) / 32) * 32 + 1, V3_RSZ = (4 * V3_RP * 2 + 3) / 4 * 4;
  constexpr int NT_ = 64 * NW, NREG = (2 * V3_RPL + NT_ - 1) / NT_, NU = 1024 / NT_;
  __shared__ __attribute__((aligned(16))) float smem[NBUF * (V3_USZ + V3_RSZ)];
  float* Us = smem;
  float* Rs = smem + NBUF * V3_USZ;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = tid >> 6;
  const int tl = lane & 15, q = lane >> 4;
  const int H = p.H, W = p.W, TH = H >> 1, TW = W >> 1;
  const int N = p.N, Cin = p.Cin;
  const int nNb = N / WN, nbx = TW / BTX, nby = TH / V3_BTY;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int qq = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + k;
  }
  const int n0 = (lid % nNb) * WN;
  int rest = lid / nNb;
  const int bx = rest % nbx; rest /= nbx;
  const int by = rest % nby; const int b = rest / nby;
  const int gy0 = 2 * by * V3_BTY - 1, gx0 = 2 * bx * BTX - 1;

  int rpix[NREG], rh[NREG];                             // region DMA: 2 * RP float4 = [half (4 ch)][pixel]
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int Lf = i * NT_ + tid;
    const int hi = Lf & 1, pix = Lf >> 1;               // lane pairs read the two 16-B halves of one pixel's 32-B chunk slice
    rh[i] = hi;
    int gp = -1;
    if (pix < V3_RPIX) {
      const int rr = pix / RC, rc = pix - rr * RC;
      const int gy = gy0 + rr, gx = gx0 + rc;
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) gp = (b * H + gy) * W + gx;
    }
    rpix[i] = gp;
  }
  // operand staging: plain global loads into registers one chunk ahead (they fly under the MFMA block), then ds_write into
  // the other LDS buffer.  (LDS-DMA was tried first: its ~25 GB/s/CU landing rate is below the ~21 GB/s/CU this loop needs at
  // full matrix rate plus hipcc serialises a builtin DMA against the next ds_read -- see profiles/README.md.)
  float4 rA[NREG], uA[NU], rB[NREG], uB[NU];          // two register sets: loads are issued TWO chunks ahead of their use
  auto gload = [&](int kc, float4 (&rreg)[NREG], float4 (&ureg)[NU]) {
    int cc = kc * V3_WK;
    const float* src = p.A0; int ld = p.ldA0;
    if (p.A1 != nullptr && cc >= p.C0) { src = p.A1; ld = p.ldA1; cc -= p.C0; }
#pragma unroll
    for (int i = 0; i < NREG; ++i)
      rreg[i] = rpix[i] >= 0 ? ld4(src + (long long)rpix[i] * ld + cc + 4 * rh[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NU; ++i) {                      // U chunk: 1024 float4 = [pos][cout][half]; halves swapped for couts with bit 3 set
      const int Lf = i * NT_ + tid;
      const int pos = Lf >> 6, co = (Lf >> 1) & 31, hs = Lf & 1;
      ureg[i] = ld4(Uw + (((long long)kc * 16 + pos) * N + n0 + co) * V3_WK + hs * 4);
    }
  };
  auto lstore = [&](int buf, const float4 (&rreg)[NREG], const float4 (&ureg)[NU]) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int Lf = i * NT_ + tid, pix = Lf >> 1;
      if (Lf < 2 * V3_RPL && pix < V3_RPIX) {
        float2* r0 = reinterpret_cast<float2*>(Rs + buf * V3_RSZ) + (2 * rh[i]) * V3_RP + pix;
        r0[0] = make_float2(rreg[i].x, rreg[i].y);
        r0[V3_RP] = make_float2(rreg[i].z, rreg[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int Lf = i * NT_ + tid;
      const int pos = Lf >> 6, co = (Lf >> 1) & 31, hs = Lf & 1;
      float2* u0 = reinterpret_cast<float2*>(Us + buf * V3_USZ) + ((pos * 2 + (co >> 4)) * 4 + 2 * hs) * 16 + (co & 15);
      u0[0] = make_float2(ureg[i].x, ureg[i].y);
      u0[16] = make_float2(ureg[i].z, ureg[i].w);
    }
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = Cin / V3_WK;
  const int pbase = (2 * wid) * RC + 2 * tl;
  const int roff = (q * V3_RP + pbase) * 2;                               // float offset of this lane's k-pair row / patch corner
  // one chunk: patch reads + transform + 64 MFMAs from LDS buffer `cur`
  auto compute = [&](int kc, int cur) {
    const float* Rb = Rs + cur * V3_RSZ + roff;
    const float* Ub = Us + cur * V3_USZ;
    float2 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[r][c] = (ABL == 2) ? make_float2(1.f + r, 2.f + c + kc) : *reinterpret_cast<const float2*>(Rb + (r * RC + c) * 2);
    if (ABL != 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float2 a0 = d[0][c], a1 = d[1][c], a2 = d[2][c], a3 = d[3][c];
        d[0][c] = make_float2(a0.x - a2.x, a0.y - a2.y); d[1][c] = make_float2(a1.x + a2.x, a1.y + a2.y);
        d[2][c] = make_float2(a2.x - a1.x, a2.y - a1.y); d[3][c] = make_float2(a1.x - a3.x, a1.y - a3.y);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 a0 = d[r][0], a1 = d[r][1], a2 = d[r][2], a3 = d[r][3];
        d[r][0] = make_float2(a0.x - a2.x, a0.y - a2.y); d[r][1] = make_float2(a1.x + a2.x, a1.y + a2.y);
        d[r][2] = make_float2(a2.x - a1.x, a2.y - a1.y); d[r][3] = make_float2(a1.x - a3.x, a1.y - a3.y);
      }
    }
    if (ABL == 1) {
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) asm volatile("" :: "v"(d[pos >> 2][pos & 3].x), "v"(d[pos >> 2][pos & 3].y));
      return;
    }
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
      const float2 b0 = *reinterpret_cast<const float2*>(Ub + (((pos * 2 + 0) * 4 + q) * 16 + tl) * 2);
      const float2 b1 = *reinterpret_cast<const float2*>(Ub + (((pos * 2 + 1) * 4 + q) * 16 + tl) * 2);
      const float2 a = d[pos >> 2][pos & 3];
      acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[pos][0], 0, 0, 0);
      acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[pos][1], 0, 0, 0);
      acc[pos][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[pos][0], 0, 0, 0);
      acc[pos][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[pos][1], 0, 0, 0);
    }
  };

  // software pipeline, depth 2: at chunk kc the loads of chunk kc+2 are issued, chunk kc+1 (loaded during chunk kc-1) is written
  // to the other LDS buffer after the MFMAs; one barrier per chunk.  Unrolled by two so the register sets are named statically.
  gload(0, rA, uA);
  lstore(0, rA, uA);
  if (nchunks > 1) gload(1, rB, uB);
  __syncthreads();
  for (int kc = 0; kc < nchunks; kc += 2) {
    if (ABL != 3 && kc + 2 < nchunks) gload(kc + 2, rA, uA);
    compute(kc, 0);
    if (kc + 1 < nchunks) lstore(1, rB, uB);
    __syncthreads();
    if (kc + 1 < nchunks) {
      if (ABL != 3 && kc + 3 < nchunks) gload(kc + 3, rB, uB);
      compute(kc + 1, 1);
      if (kc + 2 < nchunks) lstore(0, rA, uA);
      __syncthreads();
    }
  }

  if (p.wide_epi) {
    // Output transform in registers, then the (2 NW x 32)-pixel x 32-cout tile goes through LDS (the operand buffers are free now) so that
    // every pixel's 32 couts leave as 128 contiguous bytes: float4 stores, float4 bias / residual reads.
    constexpr int EP = WN + 4;                            // floats per staged pixel row
    float* S = smem;
    __syncthreads();                                      // all waves are done reading the last chunk's buffers
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        float m[16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) m[pos] = acc[pos][nt][r];
        const float s0 = m[0] + m[4] + m[8], s1 = m[1] + m[5] + m[9], s2 = m[2] + m[6] + m[10], s3 = m[3] + m[7] + m[11];
        const float u0 = m[4] - m[8] - m[12], u1 = m[5] - m[9] - m[13], u2 = m[6] - m[10] - m[14], u3 = m[7] - m[11] - m[15];
        const float yv[2][2] = {{s0 + s1 + s2, s1 - s2 - s3}, {u0 + u1 + u2, u1 - u2 - u3}};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx)
            S[((2 * wid + dy) * (2 * BTX) + 2 * (q * 4 + r) + dx) * EP + nt * 16 + tl] = yv[dy][dx];
      }
    __syncthreads();
    const int c4 = (tid & 7) * 4, n = n0 + c4;
    float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias_n) add = ld4(p.bias_n + n);
    if (p.bias_bn) add = f4add(add, ld4(p.bias_bn + (long long)b * p.ld_bias_bn + n));
    constexpr int NPIX = 2 * V3_BTY * 2 * BTX;            // pixels of the workgroup tile
#pragma unroll
    for (int i = 0; i < NPIX / (NT_ / 8); ++i) {
      const int pl = (tid >> 3) + (NT_ / 8) * i;
      const int hh = 2 * by * V3_BTY + pl / (2 * BTX), ww = 2 * bx * BTX + pl % (2 * BTX);
      const long long pix = ((long long)b * H + hh) * W + ww;
      const float4 y4 = *reinterpret_cast<const float4*>(S + pl * EP + c4);
      float4 v = make_float4(p.alpha * y4.x + add.x, p.alpha * y4.y + add.y, p.alpha * y4.z + add.z, p.alpha * y4.w + add.w);
      if (p.res_mode == 1) v = f4add(v, ld4(p.res + pix * p.ldRes + n));
      else if (p.res_mode == 2) v = f4add(v, ld4(p.res + (((long long)b * (H >> 1) + (hh >> 1)) * (W >> 1) + (ww >> 1)) * p.ldRes + n));
      v = make_float4(v.x * p.out_scale, v.y * p.out_scale, v.z * p.out_scale, v.w * p.out_scale);
      float* dst = p.C + pix * p.ldC + n;
      if (p.accumulate) v = f4add(v, ld4(dst));
      *reinterpret_cast<float4*>(dst) = v;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ey = by * V3_BTY + wid, ex = bx * BTX + q * 4 + r;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + nt * 16 + tl;
      float m[16];
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) m[pos] = acc[pos][nt][r];
      const float s0 = m[0] + m[4] + m[8], s1 = m[1] + m[5] + m[9], s2 = m[2] + m[6] + m[10], s3 = m[3] + m[7] + m[11];
      const float u0 = m[4] - m[8] - m[12], u1 = m[5] - m[9] - m[13], u2 = m[6] - m[10] - m[14], u3 = m[7] - m[11] - m[15];
      const float yv[2][2] = {{s0 + s1 + s2, s1 - s2 - s3}, {u0 + u1 + u2, u1 - u2 - u3}};
      float add = 0.f;
      if (p.bias_n) add += p.bias_n[n];
      if (p.bias_bn) add += p.bias_bn[(long long)b * p.ld_bias_bn + n];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int hh = 2 * ey + dy, ww = 2 * ex + dx;
          const long long pix = ((long long)b * H + hh) * W + ww;
          float v = p.alpha * yv[dy][dx] + add;
          if (p.res_mode == 1) v += p.res[pix * p.ldRes + n];
          else if (p.res_mode == 2) v += p.res[(((long long)b * (H >> 1) + (hh >> 1)) * (W >> 1) + (ww >> 1)) * p.ldRes + n];
          v *= p.out_scale;
          float* dst = p.C + pix * p.ldC + n;
          if (p.accumulate) v += *dst;
          *dst = v;
        }
    }
  }
}
}  // namespace

bool wino_supported(const IgemmParams& p) {
  if (!((p.H / 2) % 4 == 0 && (p.W / 2) % BTX == 0 && (long long)p.M < (1LL << 31))) return false;
  return (p.H % 2 == 0) && (p.W % 2 == 0) && (p.Cin % V3_WK == 0) && (p.N % WN == 0) && (p.A1 == nullptr || p.C0 % V3_WK == 0) && p.bias_m == nullptr;
}

static float* g_zero_page = nullptr;
void launch_wino(const IgemmParams& p_in, const float* Uw, hipStream_t st) {
  IgemmParams p = p_in;
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool wide = (p.ldC % 4 == 0) && al16(p.C);
    if (p.bias_n) wide = wide && al16(p.bias_n);
    if (p.bias_bn) wide = wide && al16(p.bias_bn) && (p.ld_bias_bn % 4 == 0);
    if (p.res_mode) wide = wide && al16(p.res) && (p.ldRes % 4 == 0);
    const bool force_scalar = cur_opt().wino_epi == 0;     // A/B switch
    p.wide_epi = (wide && !force_scalar) ? 1 : 0;
  }
  if (!g_zero_page) { (void)hipMalloc(&g_zero_page, 256); (void)hipMemset(g_zero_page, 0, 256); }
  const int B = p.M / (p.H * p.W);
  const int TH = p.H / 2, TW = p.W / 2;
  const int abl = cur_opt().wino_abl;     // timing ablations (wrong results): 1 no MFMA, 2 no patch reads, 3 no DMA
  const int geo = cur_opt().wino_geo;    // 42: 4-wave workgroups, 2 per CU (default, fastest); 82: 8-wave workgroups
  const float* z = g_zero_page;
  if (TH % 8 == 0 && geo / 10 == 8) {
    const int grid = B * (TH / 8) * (TW / BTX) * (p.N / WN);
    if (abl == 1) hipLaunchKernelGGL((wino3_kernel<1, 8, 2>), dim3(grid), dim3(512), 0, st, p, Uw, z);
    else if (abl == 2) hipLaunchKernelGGL((wino3_kernel<2, 8, 2>), dim3(grid), dim3(512), 0, st, p, Uw, z);
    else if (abl == 3) hipLaunchKernelGGL((wino3_kernel<3, 8, 2>), dim3(grid), dim3(512), 0, st, p, Uw, z);
    else hipLaunchKernelGGL((wino3_kernel<0, 8, 2>), dim3(grid), dim3(512), 0, st, p, Uw, z);
  } else {
    const int grid = B * (TH / 4) * (TW / BTX) * (p.N / WN);
    hipLaunchKernelGGL((wino3_kernel<0, 4, 2>), dim3(grid), dim3(256), 0, st, p, Uw, z);
  }
}

// host: U[Cin/8][pos][cout][8] from tap-major packed weights wt[cout][(dy*3+dx)*Cin + cin]
void wino_transform_weights(const float* wt, int Cout, int Cin, float* U) {
  const int wk = V3_WK;
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      double g[3][3], t[4][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = wt[(size_t)o * 9 * Cin + (size_t)(a * 3 + b) * Cin + i];
      for (int xi = 0; xi < 4; ++xi) for (int b = 0; b < 3; ++b) t[xi][b] = G[xi][0] * g[0][b] + G[xi][1] * g[1][b] + G[xi][2] * g[2][b];
      for (int xi = 0; xi < 4; ++xi) for (int nu = 0; nu < 4; ++nu) {
        const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
        U[(((size_t)(i / wk) * 16 + xi * 4 + nu) * Cout + o) * wk + (i % wk)] = (float)u;
      }
    }
}

}  // namespace buddy
