// Device-side preparation of the 3x3 convolution weights (reference ddpm_conv3x3, networks/ncsnpp_utils/layers.py:119-126): from the raw
// torch OIHW tensor in HBM straight to the operand form ONE kernel variant reads -- direct [Co][9][Ci], Winograd F(2x2,3x3) [Ci/8][16][Co][8],
// F(4x4,3x3) [36][Co][Ci] or F(6x6,3x3) [64][Co][Ci] -- for the forward or the data-gradient direction (taps flipped, roles of the channel
// axes exchanged).  Round 3 did all of this on one host thread for every variant of every layer at handle creation (2.6 GB computed and
// uploaded, ~7 s); here a layer's variant is produced on first use by one launch (a thread per (co, ci) pair, G g G^T in fp64 like the host
// restatement kept in wino*.hip for the unit tests), the whole network in a few milliseconds.
//
// Spatial axes: the network runs NHWC with H = time frames and W = frequency bins, torch weights are [O][I][ky = frequency][kx = time], so tap
// (dy, dx) of this layout reads w[o][i][ky = dx][kx = dy].
#include "common.h"

namespace buddy {
namespace {

__device__ __forceinline__ void g_rows(const int kind, double G[8][3]) {
  if (kind == 2) {
    const double g[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) G[r][c] = g[r][c];
  } else if (kind == 4) {
    const double g[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) G[r][c] = g[r][c];
  } else {
    const double g[8][3] = {{1, 0, 0}, {-2.0 / 9, -2.0 / 9, -2.0 / 9}, {-2.0 / 9, 2.0 / 9, -2.0 / 9}, {1.0 / 90, 1.0 / 45, 2.0 / 45},
                            {1.0 / 90, -1.0 / 45, 2.0 / 45}, {32.0 / 45, 16.0 / 45, 8.0 / 45}, {32.0 / 45, -16.0 / 45, 8.0 / 45}, {0, 0, 1}};
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) G[r][c] = g[r][c];
  }
}

// One thread per (co, ci) of THIS convolution (ci fastest: the stores of every position are coalesced along Ci).  O, I: the raw tensor's dims.
// fp64 products and sums are kept un-contracted (fp contract off: no fma) so the result is the host restatement's, bit for bit.
template <int KIND>
__global__ __launch_bounds__(256) void conv3_weight_prep_kernel(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int Co = dgrad ? I : O, Ci = dgrad ? O : I;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)Co * Ci) return;
  const int co = (int)(idx / Ci), ci = (int)(idx % Ci);
  const int o = dgrad ? ci : co, i = dgrad ? co : ci;
  const float* src = w + ((long long)o * I + i) * 9;
  float raw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) raw[k] = src[k];
  double g[3][3];                                             // g[dy][dx] of this convolution
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ey = dgrad ? 2 - dy : dy, ex = dgrad ? 2 - dx : dx;
      g[dy][dx] = (double)raw[ex * 3 + ey];
    }
  if (KIND == 0) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) out[((long long)co * 9 + dy * 3 + dx) * Ci + ci] = (float)g[dy][dx];
    return;
  }
  constexpr int R = KIND == 2 ? 4 : (KIND == 4 ? 6 : 8);
  double G[8][3];
  g_rows(KIND, G);
  double t[R][3];
#pragma unroll
  for (int xi = 0; xi < R; ++xi)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      t[xi][b] = G[xi][0] * g[0][b] + G[xi][1] * g[1][b] + G[xi][2] * g[2][b];
#pragma unroll
  for (int xi = 0; xi < R; ++xi)
#pragma unroll
    for (int nu = 0; nu < R; ++nu) {
      const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
      long long dst;
      if (KIND == 2) dst = (((long long)(ci / 8) * 16 + xi * 4 + nu) * Co + co) * 8 + (ci % 8);
      else dst = ((long long)(xi * R + nu) * Co + co) * Ci + ci;
      out[dst] = (float)u;
    }
}

// Sub-pixel form of conv3x3(nearest-upsample x2 of x) (the BigGAN up block's Conv_0, reference layerspp.py:246-257 with up_or_down_sampling.py
// naive_upsample_2d) in the Winograd domain.  Output pixel (2 i + py, 2 j + px) only ever reads the low-resolution pixels (i - 1 + py, i + py) x
// (j - 1 + px, j + px): per axis the three taps fold into TWO taps on the low-resolution grid,
//   py = 0: [g0, g1 + g2] on rows (i - 1, i),   py = 1: [g0 + g1, g2] on rows (i, i + 1)
// so each phase is a 2x2-tap correlation, evaluated as F(7x7,2x2) on the F(6x6,3x3) points (same 8x8 patch and B^T; G rows c [1, p]; wino6.hip):
// the convolution is one input transform at the low resolution, a GEMM with 4 O columns (phase-major: column ph * O + o, ph = 2 py + px) and a
// depth-to-space output transform; its data-gradient is a GEMM from the 4 O space-to-depth channels of the gradient to I channels with each
// phase's taps flipped.  One thread per (co', ci') of the Winograd-domain GEMM: forward [64][4 O][I], data-gradient [64][I][4 O].  fp64, un-contracted.
__global__ __launch_bounds__(256) void conv3_up_weight_prep_kernel(const float* __restrict__ w, int O, int I, int dgrad, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int Co = dgrad ? I : 4 * O, Ci = dgrad ? 4 * O : I;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)Co * Ci) return;
  const int co = (int)(idx / Ci), ci = (int)(idx % Ci);
  const int po = dgrad ? ci : co, ph = po / O, o = po - ph * O, i = dgrad ? co : ci, py = ph >> 1, px = ph & 1;
  const float* src = w + ((long long)o * I + i) * 9;
  double g[3][3];                                             // g[dy][dx] in this layout (dy: time, dx: frequency) = w[o][i][ky = dx][kx = dy]
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) g[dy][dx] = (double)src[dx * 3 + dy];
  double k[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ey = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 2 ? 1 : 0), ex = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 2 ? 1 : 0);
      k[ey][ex] = k[ey][ex] + g[dy][dx];
    }
  if (dgrad) {                                                // the adjoint of a correlation: both axes flipped
    const double t0 = k[0][0]; k[0][0] = k[1][1]; k[1][1] = t0;
    const double t1 = k[0][1]; k[0][1] = k[1][0]; k[1][0] = t1;
  }
  double G[8][3];
  g_rows(6, G);                                               // F(7x7,2x2) on the same points: the first two columns (c, c p); the point at infinity takes the last tap
  G[7][1] = 1.0;
  double t[8][2];
#pragma unroll
  for (int xi = 0; xi < 8; ++xi)
#pragma unroll
    for (int b = 0; b < 2; ++b) t[xi][b] = G[xi][0] * k[0][b] + G[xi][1] * k[1][b];
#pragma unroll
  for (int xi = 0; xi < 8; ++xi)
#pragma unroll
    for (int nu = 0; nu < 8; ++nu) {
      const double u = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1];
      out[((long long)(xi * 8 + nu) * Co + co) * Ci + ci] = (float)u;
    }
}

}  // namespace

long long conv3_weight_floats(int O, int I, int kind) {
  const long long n = (long long)O * I;
  return kind == 0 ? 9 * n : kind == 2 ? 16 * n : kind == 4 ? 36 * n : kind == 6 ? 64 * n : kind == 61 ? 256 * n : 0;
}

int launch_conv3_weight_prep(const float* w_oihw, int O, int I, bool dgrad, int kind, float* out, hipStream_t st) {
  const long long n = (long long)O * I;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (kind) {
    case 0: hipLaunchKernelGGL(conv3_weight_prep_kernel<0>, grid, block, 0, st, w_oihw, O, I, dgrad ? 1 : 0, out); break;
    case 2: if ((dgrad ? O : I) % 8) return BUDDY_ERR_ARG;
            hipLaunchKernelGGL(conv3_weight_prep_kernel<2>, grid, block, 0, st, w_oihw, O, I, dgrad ? 1 : 0, out); break;
    case 4: hipLaunchKernelGGL(conv3_weight_prep_kernel<4>, grid, block, 0, st, w_oihw, O, I, dgrad ? 1 : 0, out); break;
    case 6: hipLaunchKernelGGL(conv3_weight_prep_kernel<6>, grid, block, 0, st, w_oihw, O, I, dgrad ? 1 : 0, out); break;
    case 61: hipLaunchKernelGGL(conv3_up_weight_prep_kernel, dim3((unsigned)((4 * n + 255) / 256)), block, 0, st, w_oihw, O, I, dgrad ? 1 : 0, out); break;
    default: return BUDDY_ERR_ARG;
  }
  return BUDDY_OK;
}

}  // namespace buddy
