// Does the position-plane layout of M / V ([pos][tile][C]: 64 reads of 512 B, 15+ MB apart, per tile) cost HBM efficiency against a tile-blocked layout
// ([tile / 256][pos][tile % 256][C]: the same 64 reads 128 KB apart)?  A stand-in for the output transform's access pattern: one workgroup per tile, thread =
// (column 0..7, channel quad 0..31) reads 8 float4 (positions i * 8 + col), adds them and writes six float4 of "output".  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/scatter_read_probe.hip -o /tmp/srp && /tmp/srp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int C = 128, TB = 256;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ M, float* __restrict__ y, int Mt) {
  const int tile = blockIdx.x, tid = threadIdx.x, ql = tid & 31, col = tid >> 5;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pos = i * 8 + col;
    const long long off = MODE == 0 ? ((long long)pos * Mt + tile) * C : (((long long)(tile / TB) * 64 + pos) * TB + (tile % TB)) * C;
    m[i] = *reinterpret_cast<const float4*>(M + off + ql * 4);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc.x += m[i].x; acc.y += m[i].y; acc.z += m[i].z; acc.w += m[i].w; }
  if (col < 6) {
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(y + (((long long)tile * 36 + col * 6 + j) * C) + ql * 4) = acc;
  }
}
int main() {
  const int Mt = 29696;                                        // 116 blocks of 256 tiles (level 0 at B = 8: 29 584)
  float *M, *y;
  hipMalloc(&M, (size_t)64 * Mt * C * 4); hipMalloc(&y, (size_t)36 * Mt * C * 4);
  hipMemset(M, 0, (size_t)64 * Mt * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(Mt), dim3(256), 0, 0, M, y, Mt); else hipLaunchKernelGGL(k<1>, dim3(Mt), dim3(256), 0, 0, M, y, Mt);
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(Mt), dim3(256), 0, 0, M, y, Mt); else hipLaunchKernelGGL(k<1>, dim3(Mt), dim3(256), 0, 0, M, y, Mt); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      const double by = (64.0 + 36.0) * Mt * C * 4;
      printf("%s layout: %.3f ms  %.0f GB/s (read %.2f GB + write %.2f GB)\n", mode == 0 ? "plane  " : "blocked", ms, by / ms / 1e6, 64.0 * Mt * C * 4 / 1e9, 36.0 * Mt * C * 4 / 1e9);
    }
  return 0;
}
