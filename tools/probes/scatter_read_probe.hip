// What does the output transform's structure cost against its bare access pattern?  Level-0 shape (29 696 tiles, 128 channels): a stand-in that reads the 64
// position planes of M (512 B per plane and tile) and writes 36 x 512 B per tile, made step by step more like w6_output_kernel<1>:
//   mode 0  one workgroup per tile, loads -> add -> stores (no LDS)          mode 1  the same from a tile-blocked layout of M
//   mode 2  + walk of 8 tiles per workgroup                                   mode 3  + LDS exchange with two barriers per tile (the separable transform)
//   mode 4  + residual read (512 B per output pixel)                          mode 5  + fp64 statistics (sum, sum of squares per channel)
//   mode 6  as 5, but a WAVE owns all 8 columns of 8 channel quads: the exchange is wave-private (no workgroup barrier), global pieces are 128 B
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probes/scatter_read_probe.hip -o /tmp/srp && /tmp/srp
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 128, TB = 256, TL = 8;
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ M, const float* __restrict__ res, float* __restrict__ y, double* __restrict__ st, int Mt) {
  __shared__ float4 lds[32 * 6 * 8];
  const int tid = threadIdx.x, ql = tid & 31, col = tid >> 5;
  const int walk = MODE >= 2 ? TL : 1;
  double s0 = 0, s1 = 0;
  for (int it = 0; it < walk; ++it) {
    const int tile = blockIdx.x * walk + it;
    if (tile >= Mt) break;
    float4 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int pos = i * 8 + col;
      const long long off = MODE == 1 ? (((long long)(tile / TB) * 64 + pos) * TB + (tile % TB)) * C : ((long long)pos * Mt + tile) * C;
      m[i] = *reinterpret_cast<const float4*>(M + off + ql * 4);
    }
    float4 o[6];
    if (MODE >= 3) {
#pragma unroll
      for (int r = 0; r < 6; ++r) lds[(r * 8 + col) * 32 + ql] = add4(m[r], m[r + 2 > 7 ? 7 : r + 2]);
      __syncthreads();
      if (col < 6) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = lds[(col * 8 + j) * 32 + ql];
#pragma unroll
        for (int j = 0; j < 6; ++j) o[j] = add4(t[j], t[j + 2]);
      }
    } else {
      float4 acc = m[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) acc = add4(acc, m[i]);
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = acc;
    }
    if (col < 6) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const long long pix = ((long long)tile * 36 + col * 6 + j) * C + ql * 4;
        float4 v = o[j];
        if (MODE >= 4) v = add4(v, *reinterpret_cast<const float4*>(res + pix));
        *reinterpret_cast<float4*>(y + pix) = v;
        if (MODE >= 5) { s0 += (double)v.x + (double)v.y + (double)v.z + (double)v.w; s1 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w; }
      }
    }
    if (MODE >= 3) __syncthreads();
  }
  if (MODE >= 5 && s0 + s1 == 12345.678) st[blockIdx.x * 256 + tid] = s0 + s1;
}

// mode 6: wave w, lane l: column l >> 3, channel quad 8 w + (l & 7)
__global__ __launch_bounds__(256) void k6(const float* __restrict__ M, const float* __restrict__ res, float* __restrict__ y, double* __restrict__ st, int Mt) {
  __shared__ float4 lds[4][6 * 8 * 8];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, col = l >> 3, q = w * 8 + (l & 7);
  float4* L = lds[w];
  double s0 = 0, s1 = 0;
  for (int it = 0; it < TL; ++it) {
    const int tile = blockIdx.x * TL + it;
    if (tile >= Mt) break;
    float4 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = *reinterpret_cast<const float4*>(M + ((long long)(i * 8 + col) * Mt + tile) * C + q * 4);
#pragma unroll
    for (int r = 0; r < 6; ++r) L[(r * 8 + col) * 8 + (l & 7)] = add4(m[r], m[r + 2 > 7 ? 7 : r + 2]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (col < 6) {
      float4 t[8], o[6];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = L[(col * 8 + j) * 8 + (l & 7)];
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = add4(t[j], t[j + 2]);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const long long pix = ((long long)tile * 36 + col * 6 + j) * C + q * 4;
        float4 v = add4(o[j], *reinterpret_cast<const float4*>(res + pix));
        *reinterpret_cast<float4*>(y + pix) = v;
        s0 += (double)v.x + (double)v.y + (double)v.z + (double)v.w; s1 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (s0 + s1 == 12345.678) st[blockIdx.x * 256 + tid] = s0 + s1;
}
float run6(const float* M, const float* res, float* y, double* st, int Mt) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k6, dim3((Mt + TL - 1) / TL), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k6, dim3((Mt + TL - 1) / TL), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

// mode 7: as 5 with the next tile's loads issued before the exchange of the current one; WALK tiles per workgroup
template <int WALK, bool PF>
__global__ __launch_bounds__(256) void k7(const float* __restrict__ M, const float* __restrict__ res, float* __restrict__ y, double* __restrict__ st, int Mt) {
  __shared__ float4 lds[32 * 6 * 8];
  const int tid = threadIdx.x, ql = tid & 31, col = tid >> 5;
  double s0 = 0, s1 = 0;
  float4 mn[8];
  auto ld = [&](int tile) {
    tile = tile < Mt ? tile : Mt - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) mn[i] = *reinterpret_cast<const float4*>(M + ((long long)(i * 8 + col) * Mt + tile) * C + ql * 4);
  };
  if (PF) ld(blockIdx.x * WALK);
  for (int it = 0; it < WALK; ++it) {
    const int tile = blockIdx.x * WALK + it;
    if (tile >= Mt) break;
    if (!PF) ld(tile);
    float4 m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = mn[i];
    if (PF && it + 1 < WALK) ld(tile + 1);
#pragma unroll
    for (int r = 0; r < 6; ++r) lds[(r * 8 + col) * 32 + ql] = add4(m[r], m[r + 2 > 7 ? 7 : r + 2]);
    __syncthreads();
    if (col < 6) {
      float4 t[8], o[6];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = lds[(col * 8 + j) * 32 + ql];
#pragma unroll
      for (int j = 0; j < 6; ++j) o[j] = add4(t[j], t[j + 2]);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const long long pix = ((long long)tile * 36 + col * 6 + j) * C + ql * 4;
        float4 v = add4(o[j], *reinterpret_cast<const float4*>(res + pix));
        *reinterpret_cast<float4*>(y + pix) = v;
        s0 += (double)v.x + (double)v.y + (double)v.z + (double)v.w; s1 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
    __syncthreads();
  }
  if (s0 + s1 == 12345.678) st[blockIdx.x * 256 + tid] = s0 + s1;
}
template <int WALK, bool PF> float run7(const float* M, const float* res, float* y, double* st, int Mt) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k7<WALK, PF>), dim3((Mt + WALK - 1) / WALK), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k7<WALK, PF>), dim3((Mt + WALK - 1) / WALK), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

// mode 9: no exchange at all -- a THREAD owns a whole tile for one channel pair (64 float2 loads = 512 B per wave and position plane, the separable transform in
// registers, 36 float2 stores), a wave = 128 channels of one tile, 4 tiles per workgroup, WALK tile groups per workgroup
__device__ __forceinline__ float2 add2(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
template <int WALK>
__global__ __launch_bounds__(256) void k9(const float* __restrict__ M, const float* __restrict__ res, float* __restrict__ y, double* __restrict__ st, int Mt) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  double s0 = 0, s1 = 0;
  for (int it = 0; it < WALK; ++it) {
    const int tile = (blockIdx.x * WALK + it) * 4 + w;
    if (tile >= Mt) break;
    float2 m[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) m[i][j] = *reinterpret_cast<const float2*>(M + ((long long)(i * 8 + j) * Mt + tile) * C + l * 2);
    float2 t[6][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 6; ++r) t[r][j] = add2(m[r][j], add2(m[r + 1][j], m[r + 2][j]));
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const long long pix = ((long long)tile * 36 + r * 6 + c) * C + l * 2;
        float2 v = add2(add2(t[r][c], add2(t[r][c + 1], t[r][c + 2])), *reinterpret_cast<const float2*>(res + pix));
        *reinterpret_cast<float2*>(y + pix) = v;
        s0 += (double)v.x + (double)v.y; s1 += (double)v.x * v.x + (double)v.y * v.y;
      }
  }
  if (s0 + s1 == 12345.678) st[blockIdx.x * 256 + tid] = s0 + s1;
}
template <int WALK> float run9(const float* M, const float* res, float* y, double* st, int Mt) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int g = (Mt + 4 * WALK - 1) / (4 * WALK);
  hipLaunchKernelGGL((k9<WALK>), dim3(g), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k9<WALK>), dim3(g), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
template <int MODE> float run(const float* M, const float* res, float* y, double* st, int Mt) {
  const int walk = MODE >= 2 ? TL : 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3((Mt + walk - 1) / walk), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<MODE>, dim3((Mt + walk - 1) / walk), dim3(256), 0, 0, M, res, y, st, Mt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  const int Mt = 29696;
  float *M, *y, *res; double* st;
  (void)hipMalloc(&M, (size_t)64 * Mt * C * 4); (void)hipMalloc(&y, (size_t)36 * Mt * C * 4); (void)hipMalloc(&res, (size_t)36 * Mt * C * 4); (void)hipMalloc(&st, (size_t)Mt * 256 * 8);
  (void)hipMemset(M, 0, (size_t)64 * Mt * C * 4); (void)hipMemset(res, 0, (size_t)36 * Mt * C * 4);
  const double rd = 64.0 * Mt * C * 4, wr = 36.0 * Mt * C * 4;
  const float t[6] = {run<0>(M, res, y, st, Mt), run<1>(M, res, y, st, Mt), run<2>(M, res, y, st, Mt), run<3>(M, res, y, st, Mt), run<4>(M, res, y, st, Mt), run<5>(M, res, y, st, Mt)};
  const char* nm[6] = {"bare pattern", "blocked layout", "+ 8-tile walk", "+ LDS exchange, 2 barriers", "+ residual read", "+ fp64 statistics"};
  { const float t6 = run6(M, res, y, st, Mt); printf("mode 6 %-28s %.3f ms  %.0f GB/s\n", "wave-private exchange", t6, (rd + 2 * wr) / t6 / 1e6); }
  { const float a = run7<8, true>(M, res, y, st, Mt), b = run7<2, false>(M, res, y, st, Mt), c = run7<1, false>(M, res, y, st, Mt), d = run7<16, true>(M, res, y, st, Mt);
    printf("mode 7 walk 8 + prefetch %.3f ms %.0f GB/s | walk 2 %.3f ms %.0f | walk 1 %.3f ms %.0f | walk 16 + prefetch %.3f ms %.0f\n", a, (rd + 2 * wr) / a / 1e6, b, (rd + 2 * wr) / b / 1e6, c, (rd + 2 * wr) / c / 1e6, d, (rd + 2 * wr) / d / 1e6); }
  { const float a = run9<1>(M, res, y, st, Mt), b = run9<2>(M, res, y, st, Mt), c = run9<4>(M, res, y, st, Mt);
    printf("mode 9 thread-per-(tile, channel pair), no exchange: walk 1 %.3f ms %.0f GB/s | walk 2 %.3f ms %.0f | walk 4 %.3f ms %.0f\n", a, (rd + 2 * wr) / a / 1e6, b, (rd + 2 * wr) / b / 1e6, c, (rd + 2 * wr) / c / 1e6); }
  for (int i = 0; i < 6; ++i) { const double by = rd + wr + (i >= 4 ? wr : 0.0); printf("mode %d %-28s %.3f ms  %.0f GB/s\n", i, nm[i], t[i], by / t[i] / 1e6); }
  return 0;
}
