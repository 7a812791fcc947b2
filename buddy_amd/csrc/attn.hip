// Single-head self-attention over all H*W positions without materialising the T x T matrix (gfx950, fp32 MFMA 16x16x4, wave64).
// Replaces the three einsum / softmax steps of reference AttnBlockpp.forward (networks/ncsnpp_utils/layerspp.py:82-86):
//     w = einsum('bchw,bcij->bhwij', q, k) * C^-0.5 ; w = softmax(w over ij) ; h = einsum('bhwij,bcij->bchw', w, v)
// and their gradients.  q, k, v, O and all gradients are token-major [B][T][C] (NHWC activations flattened), C in {64, 128, 256}.
//
// Forward  (flash_fwd_kernel): a workgroup owns 64 query rows (4 waves x 16 rows), walks the keys / values in blocks of 32 staged through
//   LDS; per block: S = q K^T (MFMA, q fragments live in registers), online softmax with the running row max / row sum kept per lane and
//   reduced across the 16 lanes that share a row by wave shuffles, P -> wave-private LDS tile -> A operand of O += P V.  Writes O and the
//   row log-sum-exp L = m + log(l).  The 905 MB/utterance attention matrix of a 30 s input is never formed.
// Backward (two kernels, no atomics, deterministic): with D = rowsum(dO o O),
//   flash_bwd_dq_kernel  per 64 query rows:  P = exp(S - L), dP = dO V^T, dS = P o (dP - D) * scale, dq += dS K
//   flash_bwd_dkv_kernel per 64 key rows:    the same tiles transposed (S^T = K q^T, dP^T = V dO^T), dv += P^T dO, dk += dS^T q
// Everything is fp32 (exact-fp32 MFMA, expf/logf); only the summation order differs from the materialised form.  The 16-bit-operand kernels
// (attention mode bf16 | f16) are in attn16.hip.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace buddy {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BR = 64, BC = 32, PLD = BC + 4;
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// row `row` of a [T][ld] matrix (4 floats at `col`), zeros for row >= T -- as an UNCONDITIONAL load from the clamped row with the condition applied to
// the value: `row < T ? ld4(..) : 0` compiles to an exec-mask save / branch / restore with its own wait per load (72 such regions and 208 waits in
// the first flash_bwd_dkv_kernel<256>), which keeps the loads of a stage from flying together
__device__ __forceinline__ float4 ld4_row(const float* base, int row, int T, long long ld, int col) {
  const float4 v = ld4(base + (long long)(row < T ? row : T - 1) * ld + col);
  return row < T ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float ld1_row(const float* base, int row, int T) { const float v = base[row < T ? row : T - 1]; return row < T ? v : 0.f; }
__device__ __forceinline__ f32x4 zero_acc() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
// Split z of gridDim.z walks the 32-row blocks [lo, hi) of the loop dimension (keys in the forward / dq kernels, queries in the dkv kernel).  A 4 s
// utterance has 2048 tokens: 32 workgroups of 64 rows per utterance, each walking 64 blocks one after the other on ONE wave per SIMD -- at B = 1 the
// three kernels took as long as at B = 8 (0.47 / 0.60 / 0.74 ms) on an eighth of the chip.  The launchers split the loop when the grid is small; the
// partial results are combined in fixed order (no atomics: results do not depend on the schedule).
__device__ __forceinline__ void split_range(int T, int& lo, int& hi) {
  const int nb = (T + BC - 1) / BC, per = (nb + (int)gridDim.z - 1) / (int)gridDim.z;
  lo = (int)blockIdx.z * per * BC;
  hi = min(T, lo + per * BC);
}
// reduce over the 16 lanes that hold one accumulator row (lanes with equal lane >> 4)
__device__ __forceinline__ float row_max16(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// stage rows [r0, r0 + 32) of a [T][C] matrix into LDS [32][C + 4] (rows >= T zero-filled)
template <int C, int NT = 256>
__device__ __forceinline__ void stage32(const float* __restrict__ src, int r0, int T, float* dst) {
  constexpr int LD = C + 4, Q = C / 4;
  for (int i = threadIdx.x; i < 32 * Q; i += NT) {
    const int r = i / Q, c = (i - r * Q) * 4;
    const float4 v = ld4_row(src, r0 + r, T, C, c);
    *reinterpret_cast<float4*>(dst + r * LD + c) = v;
  }
}

// the same staging split in two (global -> registers, registers -> LDS) so the loads of block j + 1 fly under the matrix work of block j
template <int C, int NT = 256>
__device__ __forceinline__ void fetch32(const float* __restrict__ src, int r0, int T, float4 (&r)[32 * (C / 4) / NT]) {
  constexpr int Q = C / 4;
#pragma unroll
  for (int n = 0; n < 32 * Q / NT; ++n) {
    const int i = threadIdx.x + NT * n, row = i / Q, c = (i - row * Q) * 4;
    r[n] = ld4_row(src, r0 + row, T, C, c);
  }
}
template <int C, int NT = 256>
__device__ __forceinline__ void put32(const float4 (&r)[32 * (C / 4) / NT], float* dst) {
  constexpr int LD = C + 4, Q = C / 4;
#pragma unroll
  for (int n = 0; n < 32 * Q / NT; ++n) {
    const int i = threadIdx.x + NT * n, row = i / Q, c = (i - row * Q) * 4;
    *reinterpret_cast<float4*>(dst + row * LD + c) = r[n];
  }
}

// acc[t] (16 x 16, t = 0, 1) = sum_k A[row][k] * Bs[16 t + col][k]  with A in registers (float4 per 16-wide k chunk) and Bs in LDS
template <int C>
__device__ __forceinline__ void tile_abt(const float4 (&a)[C / 16], const float* Bs, int i, int g, f32x4 (&acc)[2]) {
  constexpr int LD = C + 4;
#pragma unroll
  for (int kk = 0; kk < C / 16; ++kk) {
    const float av[4] = {a[kk].x, a[kk].y, a[kk].z, a[kk].w};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b = *reinterpret_cast<const float4*>(Bs + (16 * t + i) * LD + 16 * kk + 4 * g);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[t], 0, 0, 0);
    }
  }
}
// o[c] (16 x 16 each, c < C / 16) += Ps[row][k] * Bs[k][16 c + col], k < 32; Ps is the wave's 16 x 32 tile in LDS (row stride PLD)
template <int C>
__device__ __forceinline__ void tile_pb(const float* Ps, const float* Bs, int i, int g, f32x4 (&o)[C / 16]) {
  constexpr int LD = C + 4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 p = *reinterpret_cast<const float4*>(Ps + i * PLD + 16 * h + 4 * g);
    const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* brow = Bs + (16 * h + 4 * g + j) * LD + i;
#pragma unroll
      for (int c = 0; c < C / 16; ++c) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[j], brow[16 * c], o[c], 0, 0, 0);
    }
  }
}

// NW waves per workgroup = 16 NW query rows share every staged key / value block (NW = 8: half the L2 -> LDS traffic per FLOP of NW = 4)
template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void flash_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            float* __restrict__ O, float* __restrict__ Lse, int T, float scale) {
  // gridDim.z > 1: O / Lse are the partial buffers [split][B][T][C] / [split][B][T] (each split normalised by its own row sum; flash_combine_kernel)
  constexpr int LD = C + 4;
  __shared__ __attribute__((aligned(16))) float Ks[32 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LD];
  __shared__ __attribute__((aligned(16))) float Ps[NW][16 * PLD];
  int jlo, jhi; split_range(T, jlo, jhi);
  O += (long long)blockIdx.z * gridDim.y * T * C; Lse += (long long)blockIdx.z * gridDim.y * T;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const long long base = (long long)b * T * C;
  const int row_a = blockIdx.x * 16 * NW + 16 * w + i;       // this lane's A-operand row
  const int row0 = blockIdx.x * 16 * NW + 16 * w + 4 * g;    // first of this lane's four accumulator rows
  float4 qa[C / 16];
#pragma unroll
  for (int kk = 0; kk < C / 16; ++kk) qa[kk] = ld4_row(q + base, row_a, T, C, 16 * kk + 4 * g);
  f32x4 o[C / 16];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) o[c] = zero_acc();
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
  constexpr int NT = 64 * NW;
  static_assert(32 * (C / 4) % NT == 0, "staging assumes whole float4 rounds per thread");
  float4 pk[32 * (C / 4) / NT], pv[32 * (C / 4) / NT];
  fetch32<C, NT>(k + base, jlo, T, pk);
  fetch32<C, NT>(v + base, jlo, T, pv);
  for (int j0 = jlo; j0 < jhi; j0 += BC) {
    __syncthreads();
    put32<C, NT>(pk, Ks);
    put32<C, NT>(pv, Vs);
    __syncthreads();
    if (j0 + BC < jhi) { fetch32<C, NT>(k + base, j0 + BC, T, pk); fetch32<C, NT>(v + base, j0 + BC, T, pv); }
    f32x4 s[2] = {zero_acc(), zero_acc()};
    tile_abt<C>(qa, Ks, i, g, s);
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0 = (j0 + i < T) ? s[0][r] * scale : -INFINITY;
      float s1 = (j0 + 16 + i < T) ? s[1][r] * scale : -INFINITY;
      const float mx = row_max16(fmaxf(s0, s1));
      const float mn = fmaxf(m[r], mx);                       // finite: column j0 of every block is valid
      alpha[r] = expf(m[r] - mn);
      s0 = expf(s0 - mn); s1 = expf(s1 - mn);
      l[r] = l[r] * alpha[r] + row_sum16(s0 + s1);
      m[r] = mn;
      Ps[w][(4 * g + r) * PLD + i] = s0;
      Ps[w][(4 * g + r) * PLD + 16 + i] = s1;
    }
#pragma unroll
    for (int c = 0; c < C / 16; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[c][r] *= alpha[r];
    __syncthreads();
    tile_pb<C>(Ps[w], Vs, i, g, o);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + r;
    if (row >= T) continue;
    const float inv = 1.f / l[r];
#pragma unroll
    for (int c = 0; c < C / 16; ++c) O[base + (long long)row * C + 16 * c + i] = o[c][r] * inv;
    if (i == 0) Lse[(long long)b * T + row] = m[r] + logf(l[r]);
  }
}

// forward partials -> O, Lse: one wave per (utterance, row); Lse = log sum_z exp(Lse_z), O = sum_z exp(Lse_z - Lse) O_z, splits in fixed order
template <int C>
__global__ __launch_bounds__(256) void flash_combine_kernel(const float* __restrict__ Op, const float* __restrict__ Lp, float* __restrict__ O,
                                                            float* __restrict__ Lse, long long rows, int ns) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float mx = -INFINITY;
  for (int z = 0; z < ns; ++z) mx = fmaxf(mx, Lp[z * rows + row]);
  float sum = 0.f;
  for (int z = 0; z < ns; ++z) sum += expf(Lp[z * rows + row] - mx);
  const float inv = 1.f / sum;
  for (int c = lane * 4; c < C; c += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < ns; ++z) {
      const float w = expf(Lp[z * rows + row] - mx) * inv;
      const float4 o = ld4(Op + (z * rows + row) * C + c);
      acc.x += w * o.x; acc.y += w * o.y; acc.z += w * o.z; acc.w += w * o.w;
    }
    *reinterpret_cast<float4*>(O + row * C + c) = acc;
  }
  if (lane == 0) Lse[row] = mx + logf(sum);
}
// out = sum_z part[z] (n4 float4 per part), z ascending
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, long long n4, int ns) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 a = ld4(part + i * 4);
    for (int z = 1; z < ns; ++z) { const float4 b = ld4(part + (z * n4 + i) * 4); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    *reinterpret_cast<float4*>(out + i * 4) = a;
  }
}

// D[b][t] = sum_c dO * O
template <int C>
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ D, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 a = ld4(dO + row * C + c), o = ld4(O + row * C + c);
    acc += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) D[row] = acc;
}

template <int C>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           const float* __restrict__ dO, const float* __restrict__ Lse, const float* __restrict__ D,
                                                           float* __restrict__ dq, int T, float scale) {
  // gridDim.z > 1: dq is the partial buffer [split][B][T][C] (summed by sum_parts_kernel)
  constexpr int LD = C + 4;
  __shared__ __attribute__((aligned(16))) float Ks[32 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LD];
  __shared__ __attribute__((aligned(16))) float Ps[4][16 * PLD];
  int jlo, jhi; split_range(T, jlo, jhi);
  dq += (long long)blockIdx.z * gridDim.y * T * C;
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const long long base = (long long)b * T * C;
  const int row_a = blockIdx.x * BR + 16 * w + i, row0 = blockIdx.x * BR + 16 * w + 4 * g;
  float4 qa[C / 16], da[C / 16];
#pragma unroll
  for (int kk = 0; kk < C / 16; ++kk) {
    qa[kk] = ld4_row(q + base, row_a, T, C, 16 * kk + 4 * g);
    da[kk] = ld4_row(dO + base, row_a, T, C, 16 * kk + 4 * g);
  }
  float lse[4], dl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lse[r] = ld1_row(Lse + (long long)b * T, row0 + r, T);     // clamped to row T - 1 (launchers guarantee T >= 1)
    dl[r] = ld1_row(D + (long long)b * T, row0 + r, T);
  }
  f32x4 acc[C / 16];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) acc[c] = zero_acc();
  float4 pk[32 * (C / 4) / 256], pv[32 * (C / 4) / 256];
  fetch32<C>(k + base, jlo, T, pk);
  fetch32<C>(v + base, jlo, T, pv);
  for (int j0 = jlo; j0 < jhi; j0 += BC) {
    __syncthreads();
    put32<C>(pk, Ks);
    put32<C>(pv, Vs);
    __syncthreads();
    if (j0 + BC < jhi) { fetch32<C>(k + base, j0 + BC, T, pk); fetch32<C>(v + base, j0 + BC, T, pv); }
    f32x4 s[2] = {zero_acc(), zero_acc()}, dp[2] = {zero_acc(), zero_acc()};
    tile_abt<C>(qa, Ks, i, g, s);
    tile_abt<C>(da, Vs, i, g, dp);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool ok = (j0 + 16 * t + i < T) && (row0 + r < T);
        const float p = ok ? expf(s[t][r] * scale - lse[r]) : 0.f;
        Ps[w][(4 * g + r) * PLD + 16 * t + i] = p * (dp[t][r] - dl[r]) * scale;
      }
    }
    __syncthreads();
    tile_pb<C>(Ps[w], Ks, i, g, acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + r;
    if (row >= T) continue;
#pragma unroll
    for (int c = 0; c < C / 16; ++c) dq[base + (long long)row * C + 16 * c + i] = acc[c][r];
  }
}

template <int C>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ dO, const float* __restrict__ Lse, const float* __restrict__ D,
                                                            float* __restrict__ dk, float* __restrict__ dv, int T, float scale) {
  constexpr int LD = C + 4;
  __shared__ __attribute__((aligned(16))) float Qs[32 * LD];
  __shared__ __attribute__((aligned(16))) float Os[32 * LD];          // dO rows of the current query block
  __shared__ __attribute__((aligned(16))) float Ps[4][16 * PLD];      // P^T tile
  __shared__ __attribute__((aligned(16))) float Ss[4][16 * PLD];      // dS^T tile
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const long long base = (long long)b * T * C;
  int ilo, ihi; split_range(T, ilo, ihi);          // gridDim.z > 1: dk / dv are partial buffers [split][B][T][C]
  dk += (long long)blockIdx.z * gridDim.y * T * C; dv += (long long)blockIdx.z * gridDim.y * T * C;
  const int row_a = blockIdx.x * BR + 16 * w + i, row0 = blockIdx.x * BR + 16 * w + 4 * g;     // key / value rows
  float4 ka[C / 16], va[C / 16];
#pragma unroll
  for (int kk = 0; kk < C / 16; ++kk) {
    ka[kk] = ld4_row(k + base, row_a, T, C, 16 * kk + 4 * g);
    va[kk] = ld4_row(v + base, row_a, T, C, 16 * kk + 4 * g);
  }
  f32x4 gk[C / 16], gv[C / 16];
#pragma unroll
  for (int c = 0; c < C / 16; ++c) { gk[c] = zero_acc(); gv[c] = zero_acc(); }
  float4 pq[32 * (C / 4) / 256], po[32 * (C / 4) / 256];
  fetch32<C>(q + base, ilo, T, pq);
  fetch32<C>(dO + base, ilo, T, po);
  for (int i0 = ilo; i0 < ihi; i0 += BC) {
    __syncthreads();
    put32<C>(pq, Qs);
    put32<C>(po, Os);
    __syncthreads();
    if (i0 + BC < ihi) { fetch32<C>(q + base, i0 + BC, T, pq); fetch32<C>(dO + base, i0 + BC, T, po); }
    f32x4 s[2] = {zero_acc(), zero_acc()}, dp[2] = {zero_acc(), zero_acc()};
    tile_abt<C>(ka, Qs, i, g, s);              // S^T[key row][query col]
    tile_abt<C>(va, Os, i, g, dp);             // dP^T
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qc = i0 + 16 * t + i;          // this lane's query column
      const bool qok = qc < T;
      const float lse = ld1_row(Lse + (long long)b * T, qc, T), dl = ld1_row(D + (long long)b * T, qc, T);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (qok && row0 + r < T) ? expf(s[t][r] * scale - lse) : 0.f;
        Ps[w][(4 * g + r) * PLD + 16 * t + i] = p;
        Ss[w][(4 * g + r) * PLD + 16 * t + i] = p * (dp[t][r] - dl) * scale;
      }
    }
    __syncthreads();
    tile_pb<C>(Ps[w], Os, i, g, gv);
    tile_pb<C>(Ss[w], Qs, i, g, gk);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + r;
    if (row >= T) continue;
#pragma unroll
    for (int c = 0; c < C / 16; ++c) {
      dk[base + (long long)row * C + 16 * c + i] = gk[c][r];
      dv[base + (long long)row * C + 16 * c + i] = gv[c][r];
    }
  }
}

}  // namespace

bool flash_attn_supported(int C) { return C == 64 || C == 128 || C == 256; }

// Loop splits of the fp32 kernels: a function of T ALONE -- what fills the chip for ONE utterance (cdiv(T, 64) workgroups per split, up to 256 in
// all; at least four 32-row blocks per split, every split non-empty).  The count fixes the summation order of every row, so it must not depend on
// the batch: row b of a batched call equals the B = 1 call bit for bit (tests/test_hip_fullsize.py, tests/test_hip_multirank.py: a sharded run
// equals the single-process run).  Measured at T = 2048 (tools/attn_split_bench.py, forward / backward ms): B = 1: none 0.47 / 1.31, 8 splits
// 0.076 / 0.203; B = 2: 0.47 / 1.31 -> 0.141 / 0.379; B = 4: 0.47 / 1.31 -> 0.270 / 0.746; B = 8: 0.477 / 1.317 -> 0.535 / 1.530 (the price of
// the rule: +0.27 ms on a 68 ms step; a batch-dependent count would save it and lose the reproducibility).  Long form (T = 15008): no split.
// BUDDY_ATTN_SPLIT=n forces n (1 = never split).
int flash_attn_splits(int B, int T) {
  (void)B;
  const int force = cur_opt().attn_split;
  const int nb = cdiv(T, BC);
  const long long wgs = (long long)cdiv(T, BR);
  int ns = force > 0 ? force : (int)(256 / (wgs > 0 ? wgs : 1));
  if (ns > nb / 4) ns = nb / 4;
  if (ns < 1) ns = 1;
  while (ns > 1 && (long long)cdiv(nb, ns) * (ns - 1) >= nb) --ns;
  return ns;
}
// floats of workspace the split forms need (forward: ns x (B T C + B T); backward: 2 ns x B T C)
long long flash_attn_ws_floats(int B, int T, int C, int splits) { return splits > 1 ? 2LL * splits * B * T * C : 0; }

// O [B][T][C], Lse [B][T]; fp32 operands (exact-fp32 MFMA); the 16-bit-operand kernels are in attn16.hip.  splits > 1 (ws = flash_attn_ws_floats
// floats): the key loop is split over blockIdx.z and the partial (O, Lse) pairs are combined by flash_combine_kernel
void launch_flash_attn_fwd(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, float* ws, int splits,
                           hipStream_t st) {
  const dim3 grid(cdiv(T, BR), B), block(256);
  if (splits > 1 && ws != nullptr) {
    const long long rows = (long long)B * T;
    float* Op = ws; float* Lp = ws + (long long)splits * rows * C;
    const dim3 gs(cdiv(T, BR), B, splits), gc((unsigned)((rows + 3) / 4));
#define FA_FWDS(CC)                                                                                              \
    hipLaunchKernelGGL((flash_fwd_kernel<CC, 4>), gs, block, 0, st, q, k, v, Op, Lp, T, scale);                    \
    hipLaunchKernelGGL(flash_combine_kernel<CC>, gc, block, 0, st, Op, Lp, O, Lse, rows, splits);
    if (C == 64) { FA_FWDS(64) } else if (C == 128) { FA_FWDS(128) } else { FA_FWDS(256) }
#undef FA_FWDS
    return;
  }
  // fp32: 128-row workgroups (8 waves) once there are enough of them to fill the chip, 64-row ones otherwise (BUDDY_ATTN_NW=4|8 forces one)
  const int force_nw = cur_opt().attn_nw;
  const bool wide = force_nw ? force_nw == 8 : (long long)cdiv(T, 128) * B >= 256;
  const dim3 grid8(cdiv(T, 128), B), block8(512);
#define FA_FWD(CC)                                                                                                           \
  if (wide) hipLaunchKernelGGL((flash_fwd_kernel<CC, 8>), grid8, block8, 0, st, q, k, v, O, Lse, T, scale);                      \
  else hipLaunchKernelGGL((flash_fwd_kernel<CC, 4>), grid, block, 0, st, q, k, v, O, Lse, T, scale);
  if (C == 64) { FA_FWD(64) } else if (C == 128) { FA_FWD(128) } else { FA_FWD(256) }
#undef FA_FWD
}
// dq, dk, dv [B][T][C]; D [B][T] scratch
void launch_flash_attn_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq,
                           float* dk, float* dv, int B, int T, int C, float scale, float* ws, int splits, hipStream_t st) {
  const long long rows = (long long)B * T;
  const dim3 grid(cdiv(T, BR), B), block(256), gd((unsigned)((rows + 3) / 4));
  const float* Dc = D;
  if (splits > 1 && ws != nullptr) {     // split loops (keys for dq, queries for dk / dv), partials summed in split order
    const dim3 gs(cdiv(T, BR), B, splits);
    const long long n4 = rows * C / 4;
    const dim3 gr((unsigned)std::min<long long>((n4 + 255) / 256, 256 * 16));
    float* P0 = ws; float* P1 = ws + (long long)splits * rows * C;
#define FA_BWDS(CC)                                                                                              \
    hipLaunchKernelGGL(attn_delta_kernel<CC>, gd, block, 0, st, dO, O, D, rows);                                   \
    hipLaunchKernelGGL(flash_bwd_dq_kernel<CC>, gs, block, 0, st, q, k, v, dO, Lse, Dc, P0, T, scale);             \
    hipLaunchKernelGGL(sum_parts_kernel, gr, block, 0, st, P0, dq, n4, splits);                                    \
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<CC>, gs, block, 0, st, q, k, v, dO, Lse, Dc, P0, P1, T, scale);        \
    hipLaunchKernelGGL(sum_parts_kernel, gr, block, 0, st, P0, dk, n4, splits);                                    \
    hipLaunchKernelGGL(sum_parts_kernel, gr, block, 0, st, P1, dv, n4, splits);
    if (C == 64) { FA_BWDS(64) } else if (C == 128) { FA_BWDS(128) } else { FA_BWDS(256) }
#undef FA_BWDS
    return;
  }
#define FA_BWD(CC)                                                                                                           \
  hipLaunchKernelGGL(attn_delta_kernel<CC>, gd, block, 0, st, dO, O, D, rows);                                                 \
  hipLaunchKernelGGL(flash_bwd_dq_kernel<CC>, grid, block, 0, st, q, k, v, dO, Lse, Dc, dq, T, scale);                         \
  hipLaunchKernelGGL(flash_bwd_dkv_kernel<CC>, grid, block, 0, st, q, k, v, dO, Lse, Dc, dk, dv, T, scale);
  if (C == 64) { FA_BWD(64) } else if (C == 128) { FA_BWD(128) } else { FA_BWD(256) }
#undef FA_BWD
}

}  // namespace buddy
