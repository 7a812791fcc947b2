// Sampler-side kernels (gfx950): per-utterance (row) elementwise / reductions and the time-domain RIR operator.
#include "common.h"
#include <algorithm>

namespace buddy {
namespace {

__global__ __launch_bounds__(256) void axpby_rows_kernel(const float* x, const float* y, const float* a, const float* c, float* out, int B, int L) {
  const long long total = (long long)B * L;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / L);
    float v = a[b] * x[i];
    if (y) v += c[b] * y[i];
    out[i] = v;
  }
}

// x_hat[b][i] = x[b][i] + sqrt(t_hat^2 - t^2) * eps[b][i]      (reference EulerHeunSampler.py:41-45)
__global__ __launch_bounds__(256) void perturb_kernel(const float* x, const float* eps, float scale, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = x[i] + scale * eps[i];
}
// Euler(-Heun) update with the likelihood term (reference EulerHeunSamplerDPS.py:128-157, diff_params/edm.py:83-96):
//   x_den' = x_den * (speech_scaling / std_b) if rescale            (per utterance b)
//   d      = -t * (x_den' - x_hat) / t^2 + lh                         (ODE integrand + likelihood score)
//   out    = base + dt * (w_prev * d_prev + w_cur * d)                (Euler: base = x_hat, w_prev = 0, w_cur = 1; Heun: 0.5 / 0.5)
// also writes d (for the Heun corrector) and the rescaled x_den'.
__global__ __launch_bounds__(256) void dps_update_kernel(const float* x_hat, const float* x_den, const float* lh, const float* lh_scale_b, const float* den_scale_b,
                                                         const float* base, const float* d_prev, float t, float dt, float w_prev, float w_cur,
                                                         float* out, float* d_out, float* x_den_out, int B, int L) {
  const long long n = (long long)B * L;
  const float inv_t = 1.f / t;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / L);
    const float xd = x_den[i] * (den_scale_b ? den_scale_b[b] : 1.f);
    const float score = (xd - x_hat[i]) * inv_t * inv_t;
    float d = -t * score;
    if (lh) d += lh_scale_b ? lh_scale_b[b] * lh[i] : lh[i];
    float acc = w_cur * d;
    if (d_prev) acc += w_prev * d_prev[i];
    out[i] = base[i] + dt * acc;
    if (d_out) d_out[i] = d;
    if (x_den_out) x_den_out[i] = xd;
  }
}

// one block per row; double accumulation (the reference's y.std(), torch.norm run in fp32 with pairwise summation --
// double keeps us within fp32 round-off of either order)
__global__ __launch_bounds__(256) void row_moments_kernel(const float* x, double* out, int L) {
  __shared__ double s1[256], s2[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* r = x + (long long)b * L;
  double a = 0, q = 0;
  for (int i = tid; i < L; i += 256) { const double v = r[i]; a += v; q += v * v; }
  s1[tid] = a; s2[tid] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s1[tid] += s1[tid + off]; s2[tid] += s2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { out[b * 2] = s1[0]; out[b * 2 + 1] = s2[0]; }
}

// One scalar per utterance from a row's moments, in ONE launch (round 6: the sampler formed these with ~10 tensor expressions per step, each its own
// 4 us kernel, and an 8-workgroup moments kernel of 100 us): one workgroup of 1024 threads per row, fp64 accumulation.
//   mode 0: out[b] = p0 / std_b        unbiased standard deviation (Tensor.std()): constraint_speech_magnitude, EulerHeunSamplerDPS.py:127-129
//   mode 1: out[b] = p0 / (||row_b||_2 / p1 + 1e-8)                                 the guidance normaliser zeta / (normguide + 1e-8), :66-69
// The float roundings follow the tensor expressions they replace: std / norm rounded to fp32 first, then fp32 division(s).
__global__ __launch_bounds__(1024) void row_scale_kernel(const float* __restrict__ x, float* __restrict__ out, int L, int mode, float p0, float p1) {
  __shared__ double s1[1024], s2[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* r = x + (long long)b * L;
  double a = 0, q = 0;
  if ((L & 3) == 0 && ((uintptr_t)r & 15) == 0) {
    const float4* r4 = reinterpret_cast<const float4*>(r);
    for (int i = tid; i < L / 4; i += 1024) {
      const float4 v = r4[i];
      a += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int i = tid; i < L; i += 1024) { const double v = r[i]; a += v; q += v * v; }
  }
  s1[tid] = a; s2[tid] = q;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) { s1[tid] += s1[tid + off]; s2[tid] += s2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    if (mode == 0) {
      double var = (s2[0] - s1[0] * s1[0] / L) / (L - 1);
      if (var < 0) var = 0;
      out[b] = p0 / (float)sqrt(var);
    } else {
      const float nrm = (float)sqrt(s2[0]);
      out[b] = p0 / (nrm / p1 + 1e-8f);
    }
  }
}
// out[k][b] = v[k], k < 4: the four EDM preconditioning scalars of one sigma, broadcast over the batch (the host evaluates diff_params/edm.py:44-75 in fp32)
__global__ void fill_rows4_kernel(float* out, int B, float v0, float v1, float v2, float v3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4 * B) { const int k = i / B; out[i] = k == 0 ? v0 : k == 1 ? v1 : k == 2 ? v2 : v3; }
}

// Direct-form FIR, 1024 outputs per block, taps staged through LDS in tiles of 1024.
//   forward: y[n] = sum_m h[m] x[n - m]        adjoint: y[n] = sum_m h[m] x[n + m]
constexpr int FIR_OUT = 1024, FIR_TAPS = 1024;
template <bool ADJ>
__global__ __launch_bounds__(256) void fir_kernel(const float* x, const float* hbase, long long h_stride, float* y, int L, int M) {
  __shared__ float hs[FIR_TAPS];
  __shared__ float xs[FIR_OUT + FIR_TAPS];
  const int b = blockIdx.y, n0 = blockIdx.x * FIR_OUT, tid = threadIdx.x;
  const float* xr = x + (long long)b * L;
  const float* h = hbase + (long long)b * h_stride;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < M; m0 += FIR_TAPS) {
    __syncthreads();
    for (int i = tid; i < FIR_TAPS; i += 256) hs[i] = (m0 + i < M) ? h[m0 + i] : 0.f;
    // window of x needed: forward covers indices [n0 - m0 - (TAPS-1), n0 - m0 + OUT), adjoint [n0 + m0, n0 + m0 + OUT + TAPS - 1)
    const int base = ADJ ? (n0 + m0) : (n0 - m0 - (FIR_TAPS - 1));
    for (int i = tid; i < FIR_OUT + FIR_TAPS; i += 256) {
      const int g = base + i;
      xs[i] = (g >= 0 && g < L) ? xr[g] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int m = 0; m < FIR_TAPS; ++m) {
      const float hv = hs[m];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = tid + 256 * j;
        const int idx = ADJ ? (o + m) : (o - m + FIR_TAPS - 1);
        acc[j] += hv * xs[idx];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + tid + 256 * j;
    if (n < L) y[(long long)b * L + n] = acc[j];
  }
}

}  // namespace

void launch_axpby_rows(const float* x, const float* y, const float* a, const float* c, float* out, int B, int L, hipStream_t st) {
  long long g = ((long long)B * L + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(axpby_rows_kernel, dim3((int)g), dim3(256), 0, st, x, y, a, c, out, B, L);
}
void launch_perturb(const float* x, const float* eps, float scale, float* out, long long n, hipStream_t st) {
  long long g = (n + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(perturb_kernel, dim3((int)g), dim3(256), 0, st, x, eps, scale, out, n);
}
void launch_dps_update(const float* x_hat, const float* x_den, const float* lh, const float* lh_scale_b, const float* den_scale_b, const float* base, const float* d_prev,
                       float t, float dt, float w_prev, float w_cur, float* out, float* d_out, float* x_den_out, int B, int L, hipStream_t st) {
  long long g = ((long long)B * L + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(dps_update_kernel, dim3((int)g), dim3(256), 0, st, x_hat, x_den, lh, lh_scale_b, den_scale_b, base, d_prev, t, dt, w_prev, w_cur, out, d_out,
                     x_den_out, B, L);
}
void launch_row_scale(const float* x, float* out, int B, int L, int mode, float p0, float p1, hipStream_t st) {
  hipLaunchKernelGGL(row_scale_kernel, dim3(B), dim3(1024), 0, st, x, out, L, mode, p0, p1);
}
void launch_fill_rows4(float* out, int B, float v0, float v1, float v2, float v3, hipStream_t st) {
  hipLaunchKernelGGL(fill_rows4_kernel, dim3((4 * B + 255) / 256), dim3(256), 0, st, out, B, v0, v1, v2, v3);
}
void launch_row_moments(const float* x, double* out, int B, int L, hipStream_t st) {
  hipLaunchKernelGGL(row_moments_kernel, dim3(B), dim3(256), 0, st, x, out, L);
}
void launch_fir(const float* x, const float* h, long long h_stride, float* y, int B, int L, int M, int adjoint, hipStream_t st) {
  dim3 grid(cdiv(L, FIR_OUT), B);
  if (adjoint) hipLaunchKernelGGL(fir_kernel<true>, grid, dim3(256), 0, st, x, h, h_stride, y, L, M);
  else hipLaunchKernelGGL(fir_kernel<false>, grid, dim3(256), 0, st, x, h, h_stride, y, L, M);
}

}  // namespace buddy

// ---- calibration micro-benchmark: pure fp32 MFMA issue rate (no memory traffic), to read the matrix-core peak the chip
// actually sustains at its power-managed clock on random operands (MI355X_MICROARCH.md: DVFS give-back) ----
namespace buddy {
namespace {
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_ubench_kernel(const float* seed, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x;
  float a0 = seed[tid], a1 = seed[tid + 256], b0 = seed[tid + 512], b1 = seed[tid + 768];
  f32x16_t c00, c01, c10, c11;
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
    a0 = -a0; b1 = -b1;     // keep operands toggling without growing the accumulators
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
// the same for v_mfma_f32_32x32x16_bf16 on random bf16 operand bits (three "planes" per side, rotated like the bf16x3 GEMM does): what the matrix
// pipe sustains on THIS box for bf16 under its power budget, with no memory traffic at all
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_ubench_bf16_kernel(const float* seed, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x;
  bf16x8_t a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    u32x4_t ua, ub;
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // two bf16 per dword, exponents kept near 1 so the accumulators stay finite: sign/mantissa bits from the seed
      const unsigned int r = __float_as_uint(seed[(tid * 7 + q * 4 + j) & 1023]), t = __float_as_uint(seed[(tid * 13 + q * 4 + j + 512) & 1023]);
      ua[j] = (r & 0x807F807Fu) | 0x3F003F00u; ub[j] = (t & 0x807F807Fu) | 0x3F003F00u;
    }
    a[q] = (bf16x8_t)ua; b[q] = (bf16x8_t)ub;
  }
  f32x16_t c[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q + k) % 3], b[q], c[k], 0, 0, 0);
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
}  // namespace
void launch_mfma_ubench_bf16(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, hipStream_t st) {
  hipLaunchKernelGGL(mfma_ubench_bf16_kernel, dim3(blocks), dim3(256), 0, st, seed, out, iters, clk);
}
void launch_mfma_ubench(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, hipStream_t st) {
  hipLaunchKernelGGL(mfma_ubench_kernel, dim3(blocks), dim3(256), 0, st, seed, out, iters, clk);
}

// ---- calibration micro-benchmark: what this box's HBM sustains for a plain streaming kernel (MI355X_MICROARCH.md: 8.0 TB/s nominal, 6.29 TB/s measured
// for a float4 copy).  mode 0 copy (n16 x 16 B read + written), 1 read (summed, stored only under a condition that never holds), 2 write; nt: the
// non-temporal forms of the loads / stores.  A workgroup moves CONTIGUOUS chunks of UNROLL x 256 sixteen-byte words (UNROLL independent requests per
// thread, consecutive threads on consecutive words) and strides over the array by the grid; blocks <= 0 asks for the one-shot form: exactly one chunk
// per workgroup, UNROLL = -blocks in {1, 2, 4, 8} (|blocks| = 1 is the classic "one float4 per thread" copy).
namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int MODE, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void hbm_ubench_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, long long n16) {
  const long long chunk = (long long)UNROLL * 256, stride = (long long)gridDim.x * chunk;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (long long base = (long long)blockIdx.x * chunk; base < n16; base += stride) {
    const long long i = base + threadIdx.x;
    if (base + chunk <= n16) {
      f32x4_t v[UNROLL];
      if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (MODE == 1) { acc += v[u]; continue; }
        const f32x4_t w = MODE == 2 ? f32x4_t{1.f, 2.f, 3.f, (float)u} : v[u];
        if (NT) __builtin_nontemporal_store(w, dst + i + u * 256); else dst[i + u * 256] = w;
      }
    } else {
      for (long long j = i; j < n16; j += 256) {
        if (MODE == 1) acc += src[j];
        else dst[j] = MODE == 2 ? f32x4_t{1.f, 2.f, 3.f, 4.f} : src[j];
      }
    }
  }
  if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 1.2345678e30f) dst[0] = acc;
}
// mode 3: READ in the access pattern of the f16x2 GEMM's A operand (wgemm.hip: rows of ROWB bytes; lane (row r = lane & 31, half h = lane >> 5) of a wave
// reads 64 consecutive bytes of its row per K-stage as four 16-byte loads, two 32-row tiles per wave; a stage covers 128 bytes of every row) -- every 128-byte
// line is requested in partial pieces by four instructions, and a row's lines at different times.  STAGED = false: all stages' loads of a wave in flight at
// once; true: one stage ahead, like the kernel.  A workgroup of 4 waves covers 256 consecutive rows.
template <int ROWB, bool STAGED>
__global__ __launch_bounds__(256) void hbm_ubench_rows_kernel(const char* __restrict__ src, f32x4_t* __restrict__ dst, long long rows) {
  constexpr int S = ROWB / 128;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long r0 = ((long long)blockIdx.x * 4 + wid) * 64;
  if (r0 + 64 > rows) return;
  const char* p0 = src + (r0 + (lane & 31)) * ROWB + 64 * (lane >> 5);
  const char* p1 = p0 + 32LL * ROWB;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (!STAGED) {
    f32x4_t v[S][8];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[s][j] = *reinterpret_cast<const f32x4_t*>(p0 + s * 128 + 16 * j); v[s][4 + j] = *reinterpret_cast<const f32x4_t*>(p1 + s * 128 + 16 * j); }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[s][j];
  } else {
    f32x4_t v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const f32x4_t*>(p0 + 16 * j); v[4 + j] = *reinterpret_cast<const f32x4_t*>(p1 + 16 * j); }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      f32x4_t c[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = v[j];
      if (s + 1 < S) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const f32x4_t*>(p0 + (s + 1) * 128 + 16 * j); v[4 + j] = *reinterpret_cast<const f32x4_t*>(p1 + (s + 1) * 128 + 16 * j); }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += c[j];
      __builtin_amdgcn_s_sleep(16);            // ~1000 cycles of "matrix work" per stage
      __builtin_amdgcn_s_sleep(16);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345678e30f) dst[0] = acc;
}
template <int MODE, bool NT>
void hbm_ubench_launch(const f32x4_t* s, f32x4_t* d, long long n16, int blocks, hipStream_t st) {
  const int u = blocks > 0 ? 8 : -blocks;
  const long long chunks = (n16 + (long long)u * 256 - 1) / ((long long)u * 256);
  const unsigned g = (unsigned)(blocks > 0 ? std::min<long long>(blocks, chunks) : chunks);
  switch (u) {
    case 1: hipLaunchKernelGGL((hbm_ubench_kernel<MODE, NT, 1>), dim3(g), dim3(256), 0, st, s, d, n16); break;
    case 2: hipLaunchKernelGGL((hbm_ubench_kernel<MODE, NT, 2>), dim3(g), dim3(256), 0, st, s, d, n16); break;
    case 4: hipLaunchKernelGGL((hbm_ubench_kernel<MODE, NT, 4>), dim3(g), dim3(256), 0, st, s, d, n16); break;
    default: hipLaunchKernelGGL((hbm_ubench_kernel<MODE, NT, 8>), dim3(g), dim3(256), 0, st, s, d, n16);
  }
}
}  // namespace
int launch_hbm_ubench(const void* src, void* dst, long long bytes, int mode, int nt, int blocks, hipStream_t st) {
  const long long n16 = bytes / 16;
  if (mode == 3 || mode == 4) {       // rows pattern: nt = row bytes (512 | 1024 | 2048), all loads in flight (3) / staged (4)
    const long long rows = bytes / nt;
    const unsigned g = (unsigned)(rows / 256);
    const char* sc = reinterpret_cast<const char*>(src); f32x4_t* dd = reinterpret_cast<f32x4_t*>(dst);
    if (g < 1) return BUDDY_ERR_ARG;
    const unsigned lds = blocks > 1 && blocks <= 64 ? (unsigned)blocks * 1024u : 0u;      // blocks = KB of (unused) dynamic LDS: bounds the workgroups per CU
#define BUDDY_ROWS(RB) do { if (mode == 3) hipLaunchKernelGGL((hbm_ubench_rows_kernel<RB, false>), dim3(g), dim3(256), lds, st, sc, dd, rows); \
                            else hipLaunchKernelGGL((hbm_ubench_rows_kernel<RB, true>), dim3(g), dim3(256), lds, st, sc, dd, rows); } while (0)
    if (nt == 512) BUDDY_ROWS(512); else if (nt == 1024) BUDDY_ROWS(1024); else if (nt == 2048) BUDDY_ROWS(2048); else return BUDDY_ERR_ARG;
#undef BUDDY_ROWS
    return BUDDY_OK;
  }
  if (mode < 0 || mode > 2 || n16 < 1 || blocks == 0 || (blocks < 0 && blocks != -1 && blocks != -2 && blocks != -4 && blocks != -8) || n16 / 256 > 0x7fffffffLL)
    return BUDDY_ERR_ARG;
  const f32x4_t* s = reinterpret_cast<const f32x4_t*>(src); f32x4_t* d = reinterpret_cast<f32x4_t*>(dst);
  if (mode == 0) { if (nt) hbm_ubench_launch<0, true>(s, d, n16, blocks, st); else hbm_ubench_launch<0, false>(s, d, n16, blocks, st); }
  else if (mode == 1) { if (nt) hbm_ubench_launch<1, true>(s, d, n16, blocks, st); else hbm_ubench_launch<1, false>(s, d, n16, blocks, st); }
  else { if (nt) hbm_ubench_launch<2, true>(s, d, n16, blocks, st); else hbm_ubench_launch<2, false>(s, d, n16, blocks, st); }
  return BUDDY_OK;
}
}  // namespace buddy
