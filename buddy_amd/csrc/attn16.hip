// Single-head self-attention with 16-bit MFMA operands (bf16 or f16, fp32 accumulation, fp32 softmax statistics) -- the reduced-precision
// attention core BASELINE configs[4] names ("fp16 MFMA attention path") for reference AttnBlockpp.forward (networks/ncsnpp_utils/layerspp.py:82-86)
// and its three input gradients.  Opt-in (attention mode bf16 | f16): not the arithmetic of the reference; the fp32 kernels are in attn.hip.
//
// Design (gfx950, v_mfma_f32_32x32x16_{bf16,f16}, wave64, one wave per SIMD with the whole 512-entry register file):
//  * a pre-pass converts q, k, v (and dO) ONCE into 16-bit operand arrays: token-major rows [B][Tp][C] and channel-major transposes [B][C][Tp]
//    (Tp = T rounded up to 128, pad rows zero).  q is stored pre-multiplied by C^-1/2 log2(e), so the logits come out of the matrix pipe in the
//    log2 domain and the exponentials are bare v_exp_f32.
//  * every product is computed TRANSPOSED so that the softmax axis is lane-local: S^T = K Q^T leaves, per lane, one query column with 16 keys per
//    32-key tile in its accumulator registers (the other 16 in lane ^ 32): the row maximum is a register chain + ONE half-wave exchange, the row sum is
//    kept per lane and combined once at the end, and the exponentiated tile converts IN REGISTERS into the B operand of O^T += V^T P^T -- no LDS
//    round trip, no shuffles.  The register order of an accumulator tile is the key order (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the transposed
//    arrays are stored with exactly that permutation inside every 16 tokens, so an A fragment of V^T (K^T, Q^T, dO^T) is one 16-byte LDS read.
//  * a workgroup = 4 waves x 32 query (or key) rows; K / V^T blocks stream through a double-buffered LDS ring filled by LDS-DMA
//    (global_load_lds_dwordx4: no staging registers, no ds_write pass); LDS images are XOR-swizzled on the SOURCE address (the DMA writes
//    lane-linear) and on the read, conflict-free for ds_read_b128; one barrier per block.
//  * backward: dq kernel per 128 query rows (S^T and dP^T recomputed, dS^T -> B operand of dq^T += K^T dS^T); dk / dv kernel per 128 key rows
//    with both accumulator sets (256 registers) resident.  Deterministic: no atomics, fixed summation order.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace buddy {
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <typename T16> struct V8;
template <> struct V8<__bf16> { typedef bf16x8 t; };
template <> struct V8<_Float16> { typedef f16x8 t; };
__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr int TPAD = 128;            // token padding of the 16-bit operand arrays
constexpr float RESCALE_THR = 8.f;   // the running maximum is only raised (and O rescaled) when a row's maximum grew by more than 2^8

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// combine a per-lane value with the one of lane ^ 32 (v_permlane32_swap: r[0] = the low half's value in both halves, r[1] = the high half's)
__device__ __forceinline__ float half_max(float x) {
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  const unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
// acc *= alpha with the accumulator tile staying in the ACCUMULATOR registers (AGPRs) for the compiler: any plain VALU statement on the tile makes hipcc keep
// the loop-carried accumulators in arch VGPRs and copy all of them to / from the accumulator file around the MFMAs of EVERY iteration (128 + 128 moves per
// key block in the first build) -- the rescale is rare (deferred), so its read / multiply / write triple lives inside the asm statement
__device__ __forceinline__ void scale_acc(f32x16& o, float alpha) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x = o[r], t;
    asm("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32_e32 %1, %2, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x), "=&v"(t) : "v"(alpha));
    o[r] = x;
  }
}

// XOR swizzle of the 16-byte slot index of an LDS row with SL slots: rows 512 / 256 B -> row & 15, 128 B -> (row >> 1) & 7, 64 B -> (row >> 2) & 3:
// the 16 lanes of a ds_read_b128 group (16 consecutive-mod-16 rows, same logical slot) land on 16 different 16-byte bank slots
template <int SL> __device__ __forceinline__ int swz(int row) { return SL >= 16 ? (row & 15) : SL == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform `lds` + 16 * lane (the hardware adds the lane part), source per lane
__device__ __forceinline__ void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// stage an R-row x SL-slot tile (row r of the source at base + r * ld elements) into its swizzled LDS image; 4 waves, R * SL / 256 DMAs per wave.
// The per-lane part of the source address is loop-invariant (TileOff, 32-bit byte offsets computed once); the block origin is a wave-uniform pointer, so
// a DMA is `global_load_lds_dwordx4 v_off, s[base]` with nothing per-lane to advance between blocks (as 64-bit per-lane pointers they cost two VGPRs
// and one add per DMA instruction and block: 32 VGPRs in the dk / dv kernel)
template <int R, int SL> struct TileOff { unsigned o[R * SL / 256]; };
template <int R, int SL, typename T16>
__device__ __forceinline__ TileOff<R, SL> tile_offsets(long long ld, int w, int lane) {
  constexpr int NI = R * SL / 256;
  static_assert(R * SL % 256 == 0, "whole DMA rounds per wave");
  TileOff<R, SL> t;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = (w * NI + i) * 64 + lane, r = p / SL, sp = p % SL, sg = sp ^ swz<SL>(r);
    t.o[i] = (unsigned)(((long long)r * ld + sg * 8) * (long long)sizeof(T16));
  }
  return t;
}
template <int R, int SL, typename T16>
__device__ __forceinline__ void stage_tile(const T16* __restrict__ base, const TileOff<R, SL>& t, char* lds, int w) {
  constexpr int NI = R * SL / 256;
#pragma unroll
  for (int i = 0; i < NI; ++i) glds16(reinterpret_cast<const char*>(base) + t.o[i], lds + (w * NI + i) * 1024);
}
// A fragment (32 rows x 16 k) of a row-major [rows][C] tile: lane (row = r0 + (lane & 31), hi = lane >> 5) reads k = 16 kk + 8 hi .. + 7
template <int SL, typename T16>
__device__ __forceinline__ typename V8<T16>::t frag_rows(const char* tile, int row, int fsw, int kk, int hi) {
  return *reinterpret_cast<const typename V8<T16>::t*>(tile + row * (SL * 16) + (((2 * kk + hi) ^ fsw) << 4));
}

// The MFMA phases are written as explicit software pipelines: N fragment reads through a ring of D register sets, the read of item i + D issued right
// behind the MFMA of item i -- hipcc otherwise emits `ds_read_b128 -> s_waitcnt lgkmcnt(0) -> v_mfma` into ONE register set for every MFMA (the LDS
// latency fully exposed, round 5 first build: 0.25 of peak) and its scheduler re-serialises a plain source-level pipeline, so every step is fenced with
// sched_barrier(0); items are ordered so that consecutive MFMAs go to different accumulators (a dependent
// 32x32x16 MFMA issues every 64 cycles, an independent one every 32).
#define FA16_PIPE(N, D, LOAD, MMA)                                   \
  {                                                                  \
    v8 fr_[D];                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < (D); ++i_) fr_[i_] = LOAD(i_); \
    __builtin_amdgcn_sched_barrier(0);                               \
    _Pragma("unroll") for (int i_ = 0; i_ < (N); ++i_) {             \
      MMA(i_, fr_[i_ % (D)]);                                        \
      if (i_ + (D) < (N)) fr_[i_ % (D)] = LOAD(i_ + (D));            \
      __builtin_amdgcn_sched_barrier(0);                             \
    }                                                                \
  }
// The same in BATCHES of D for the kernels where hipcc's wait-count pass only ever emits lgkmcnt(0) (forward, dq: every wait drains all reads in
// flight, so a ring stalls for a full LDS latency every D MFMAs): two register batches; batch b + 1 is requested right behind the FIRST MFMA of batch b
// -- the one wait per batch (in front of that MFMA) finds only batch b outstanding, issued D - 1 MFMAs earlier.
#define FA16_BATCH(N, D, LOAD, MMA)                                  \
  {                                                                  \
    static_assert((N) % (D) == 0, "whole batches");                  \
    v8 fr_[2][D];                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < (D); ++i_) fr_[0][i_] = LOAD(i_); \
    __builtin_amdgcn_sched_barrier(0);                               \
    _Pragma("unroll") for (int b_ = 0; b_ < (N) / (D); ++b_) {       \
      MMA(b_ * (D), fr_[b_ & 1][0]);                                 \
      if (b_ + 1 < (N) / (D)) { _Pragma("unroll") for (int i_ = 0; i_ < (D); ++i_) fr_[(b_ + 1) & 1][i_] = LOAD((b_ + 1) * (D) + i_); } \
      __builtin_amdgcn_sched_barrier(0);                             \
      _Pragma("unroll") for (int i_ = 1; i_ < (D); ++i_) { MMA(b_ * (D) + i_, fr_[b_ & 1][i_]); } \
      __builtin_amdgcn_sched_barrier(0);                             \
    }                                                                \
  }

// ---------------------------------------------------------------------------------------------------------------- operand pre-pass
// src fp32 [B][T][C] -> rows [B][Tp][C] and / or trans [B][C][Tp] (16-bit, times mult; tokens >= T zero; trans: token order permuted inside 16s)
template <typename T16>
__global__ __launch_bounds__(256) void cvt16_kernel(const float* __restrict__ src, T16* __restrict__ rows, T16* __restrict__ trans, int T, int Tp, int C,
                                                    float mult) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = (threadIdx.x >> 4) + 16 * i, c = (threadIdx.x & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + t < T) v = *reinterpret_cast<const float4*>(src + ((long long)b * T + t0 + t) * C + c0 + c);
    v.x *= mult; v.y *= mult; v.z *= mult; v.w *= mult;
    if (rows) {
      T16 o[4] = {(T16)v.x, (T16)v.y, (T16)v.z, (T16)v.w};
      *reinterpret_cast<uint2*>(rows + ((long long)b * Tp + t0 + t) * C + c0 + c) = *reinterpret_cast<const uint2*>(o);
    }
    tile[t][c] = v.x; tile[t][c + 1] = v.y; tile[t][c + 2] = v.z; tile[t][c + 3] = v.w;
  }
  if (!trans) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (threadIdx.x >> 3) + 32 * i, g = threadIdx.x & 7;
    T16 o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = 8 * g + j;                                              // stored position within the 64-token tile
      const int idx = (p & ~15) + 4 * ((p >> 3) & 1) + (p & 3) + 8 * ((p >> 2) & 1);
      o[j] = (T16)tile[idx][c];
    }
    *reinterpret_cast<uint4*>(trans + ((long long)b * C + c0 + c) * Tp + t0 + 8 * g) = *reinterpret_cast<const uint4*>(o);
  }
}
// backward statistics: D[b][t] = sum_c dO O (fp32, also returned to the caller), Dp / L2p [B][Tp]: D and Lse log2(e) padded (D 0, L2 +inf: P = 0)
template <int C>
__global__ __launch_bounds__(256) void stats16_kernel(const float* __restrict__ dO, const float* __restrict__ O, const float* __restrict__ Lse,
                                                      float* __restrict__ D, float* __restrict__ Dp, float* __restrict__ L2p, int T, int Tp, int B) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // over B * Tp
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * Tp) return;
  const int b = (int)(row / Tp), t = (int)(row - (long long)b * Tp);
  if (t >= T) { if (lane == 0) { Dp[row] = 0.f; L2p[row] = INFINITY; } return; }
  const long long src = ((long long)b * T + t) * C;
  float acc = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(dO + src + c), o = *reinterpret_cast<const float4*>(O + src + c);
    acc += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) { D[(long long)b * T + t] = acc; Dp[row] = acc; L2p[row] = Lse[(long long)b * T + t] * kLog2e; }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// O [B][T][C] fp32, Lse [B][T] (natural log).  gridDim = (cdiv(T, 128), B, splits); splits > 1: O / Lse are partial buffers [split][...]
template <int C, typename T16>
__global__ __launch_bounds__(256, 1) void fa16_fwd_kernel(const T16* __restrict__ Qh, const T16* __restrict__ Kh, const T16* __restrict__ Vt,
                                                         float* __restrict__ O, float* __restrict__ Lse, int T, int Tp) {
  typedef typename V8<T16>::t v8;
  constexpr int BN = 64, KS = C / 16, DT = C / 32, SLK = C / 8, SLV = BN / 8, KT = BN * C * 2, VT = C * BN * 2, BUF = KT + VT;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.y;
  const int lq = lane & 31, hi = lane >> 5;
  const int qrow = blockIdx.x * 128 + 32 * w + lq;
  const int nbt = (T + BN - 1) / BN, per = (nbt + (int)gridDim.z - 1) / (int)gridDim.z;
  const int jlo = (int)blockIdx.z * per * BN, jhi = min(T, jlo + per * BN);
  O += (long long)blockIdx.z * gridDim.y * T * C; Lse += (long long)blockIdx.z * gridDim.y * T;
  const T16* Kb = Kh + (long long)b * Tp * C;
  const T16* Vb = Vt + (long long)b * C * Tp;
  v8 qf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) qf[kk] = *reinterpret_cast<const v8*>(Qh + ((long long)b * Tp + qrow) * C + 16 * kk + 8 * hi);
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
  float m = -INFINITY, l = 0.f;
  const int fk = swz<SLK>(lq), fv = swz<SLV>(lq);
  const TileOff<BN, SLK> ok = tile_offsets<BN, SLK, T16>(C, w, lane);
  const TileOff<C, SLV> ov = tile_offsets<C, SLV, T16>(Tp, w, lane);
  if (jlo < jhi) {
    stage_tile<BN, SLK, T16>(Kb + (long long)jlo * C, ok, smem, w);
    stage_tile<C, SLV, T16>(Vb + jlo, ov, smem + KT, w);
  }
  int it = 0;
  for (int j0 = jlo; j0 < jhi; j0 += BN, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j0 + BN < jhi) {
      char* nb = smem + ((it + 1) & 1) * BUF;
      stage_tile<BN, SLK, T16>(Kb + (long long)(j0 + BN) * C, ok, nb, w);
      stage_tile<C, SLV, T16>(Vb + j0 + BN, ov, nb + KT, w);
    }
    const char* kb = smem + (it & 1) * BUF;
    const char* vb = kb + KT;
    f32x16 s[2] = {zero16(), zero16()};
#define LD_(i) frag_rows<SLK, T16>(kb, 32 * ((i) & 1) + lq, fk, (i) >> 1, hi)
#define MM_(i, f) s[(i) & 1] = mma32(f, qf[(i) >> 1], s[(i) & 1])
    FA16_BATCH(2 * KS, 8, LD_, MM_)
#undef LD_
#undef MM_
    if (j0 + BN > T) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (j0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi >= T) s[t][r] = -INFINITY;
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
    mx = half_max(mx);
    const float mn = fmaxf(m, mx);
    if (__any(mn > m + RESCALE_THR)) {              // wave-uniform: rescale EVERYTHING accumulated at the old maximum, exactly once
      const float alpha = fexp2(m - mn);
      l *= alpha;
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last P V MFMAs have long retired; the asm reads below are not hazard-checked by hipcc
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) scale_acc(o[dt], alpha);
      m = mn;
    }
    v8 pf[4];
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fexp2(s[t][r] - m);
        ls += p;
        pf[2 * t + (r >> 3)][r & 7] = (T16)p;
      }
    l += ls;
#define LD_(i) *reinterpret_cast<const v8*>(vb + (32 * ((i) % DT) + lq) * (SLV * 16) + (((2 * ((i) / DT) + hi) ^ fv) << 4))
#define MM_(i, f) o[(i) % DT] = mma32(f, pf[(i) / DT], o[(i) % DT])
    FA16_BATCH(4 * DT, (DT >= 8 ? 8 : DT), LD_, MM_)
#undef LD_
#undef MM_
  }
  const float lt = half_sum(l), inv = 1.f / lt;
  if (qrow < T) {
    float* orow = O + ((long long)b * T + qrow) * C;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * hi) =
            make_float4(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
    if (hi == 0) Lse[(long long)b * T + qrow] = (m + log2f(lt)) * kLn2;
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward: dq
// per 128 query rows: S^T = K Qs^T, dP^T = V dO^T, G^T = P^T o (dP^T - D), dq^T += K^T G^T; dq = scale * (...)
template <int C, typename T16>
__global__ __launch_bounds__(256, 1) void fa16_dq_kernel(const T16* __restrict__ Qh, const T16* __restrict__ Kh, const T16* __restrict__ Kt,
                                                        const T16* __restrict__ Vh, const T16* __restrict__ dOh, const float* __restrict__ L2p,
                                                        const float* __restrict__ Dp, float* __restrict__ dq, int T, int Tp, float scale) {
  typedef typename V8<T16>::t v8;
  constexpr int BN = 32, KS = C / 16, DT = C / 32, SLK = C / 8, SLT = BN / 8, RT = BN * C * 2, BUF = 3 * RT;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.y;
  const int lq = lane & 31, hi = lane >> 5;
  const int qrow = blockIdx.x * 128 + 32 * w + lq;
  const T16* Kb = Kh + (long long)b * Tp * C;
  const T16* Vb = Vh + (long long)b * Tp * C;
  const T16* Ktb = Kt + (long long)b * C * Tp;
  v8 qf[KS], df[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    qf[kk] = *reinterpret_cast<const v8*>(Qh + ((long long)b * Tp + qrow) * C + 16 * kk + 8 * hi);
    df[kk] = *reinterpret_cast<const v8*>(dOh + ((long long)b * Tp + qrow) * C + 16 * kk + 8 * hi);
  }
  const float l2 = L2p[(long long)b * Tp + qrow], dl = Dp[(long long)b * Tp + qrow];      // pad rows: +inf / 0 -> P = 0
  f32x16 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = zero16();
  const int fk = swz<SLK>(lq), ft = swz<SLT>(lq);
  const TileOff<BN, SLK> orow = tile_offsets<BN, SLK, T16>(C, w, lane);
  const TileOff<C, SLT> otr = tile_offsets<C, SLT, T16>(Tp, w, lane);
  stage_tile<BN, SLK, T16>(Kb, orow, smem, w);
  stage_tile<BN, SLK, T16>(Vb, orow, smem + RT, w);
  stage_tile<C, SLT, T16>(Ktb, otr, smem + 2 * RT, w);
  int it = 0;
  for (int j0 = 0; j0 < T; j0 += BN, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {                                                    // unconditional (the last iteration re-stages its own block into the idle buffer): the loop
      const int jn = j0 + BN < T ? j0 + BN : j0;         // body stays ONE basic block, which hipcc's wait-count pass needs for counted lgkmcnt waits
      char* nb = smem + ((it + 1) & 1) * BUF;
      stage_tile<BN, SLK, T16>(Kb + (long long)jn * C, orow, nb, w);
      stage_tile<BN, SLK, T16>(Vb + (long long)jn * C, orow, nb + RT, w);
      stage_tile<C, SLT, T16>(Ktb + jn, otr, nb + 2 * RT, w);
    }
    const char* kb = smem + (it & 1) * BUF;
    const char* vb = kb + RT;
    const char* tb = kb + 2 * RT;
    f32x16 sd[2] = {zero16(), zero16()};              // [0] S^T, [1] dP^T: two independent accumulation chains, interleaved
#define LD_(i) frag_rows<SLK, T16>(((i) & 1) ? vb : kb, lq, fk, (i) >> 1, hi)
#define MM_(i, f) sd[(i) & 1] = mma32(f, ((i) & 1) ? df[(i) >> 1] : qf[(i) >> 1], sd[(i) & 1])
    FA16_BATCH(2 * KS, 8, LD_, MM_)
#undef LD_
#undef MM_
    const f32x16 s = sd[0], dp = sd[1];
    v8 gf[2];
    const int kleft = T - j0 - 4 * hi;                 // keys >= T (the zero pad rows of K / V): G = 0.  Branch-free: a tail-only branch splits the loop body
#pragma unroll                                         // into basic blocks and hipcc's wait-count pass then drains the fragment ring at every wait
    for (int r = 0; r < 16; ++r) {
      float g = fexp2(s[r] - l2) * (dp[r] - dl);
      g = ((r & 3) + 8 * (r >> 2) < kleft) ? g : 0.f;
      gf[r >> 3][r & 7] = (T16)g;
    }
#define LD_(i) *reinterpret_cast<const v8*>(tb + (32 * ((i) % DT) + lq) * (SLT * 16) + (((2 * ((i) / DT) + hi) ^ ft) << 4))
#define MM_(i, f) acc[(i) % DT] = mma32(f, gf[(i) / DT], acc[(i) % DT])
    FA16_BATCH(2 * DT, (DT >= 8 ? 8 : DT), LD_, MM_)
#undef LD_
#undef MM_
  }
  if (qrow < T) {
    float* orow = dq + ((long long)b * T + qrow) * C;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * hi) =
            make_float4(acc[dt][4 * g] * scale, acc[dt][4 * g + 1] * scale, acc[dt][4 * g + 2] * scale, acc[dt][4 * g + 3] * scale);
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward: dk, dv
// per 128 key rows: S = Qs K^T, dP = dO V^T (query rows in the accumulator registers), P = exp2(S - L2), G = P o (dP - D),
// dv^T += dO^T P, dk^T += Qs^T G; dk = ln 2 * (...) (Qs carries scale log2 e)
template <int C, typename T16>
__global__ __launch_bounds__(256, 1) void fa16_dkv_kernel(const T16* __restrict__ Qh, const T16* __restrict__ Qt, const T16* __restrict__ Kh,
                                                         const T16* __restrict__ Vh, const T16* __restrict__ dOh, const T16* __restrict__ dOt,
                                                         const float* __restrict__ L2p, const float* __restrict__ Dp, float* __restrict__ dk,
                                                         float* __restrict__ dv, int T, int Tp) {
  typedef typename V8<T16>::t v8;
  constexpr int BN = 32, KS = C / 16, DT = C / 32, SLK = C / 8, SLT = BN / 8, RT = BN * C * 2, BUF = 4 * RT + 256;   // + the block's 32 L2 and 32 D values
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.y;
  const int lq = lane & 31, hi = lane >> 5;
  const int krow = blockIdx.x * 128 + 32 * w + lq;
  // the row statistics of a query block ride with its tiles: lanes 0-7 of wave 0 move the 32 L2 values, lanes 8-15 the 32 D values (one exec-masked DMA;
  // held in registers across the S / dP phase they cost 32 VGPRs the 256-accumulator kernel does not have)
  auto stage_ld = [&](int i0n, char* dst) {
    if (w == 0 && lane < 16) glds16((lane < 8 ? L2p + (long long)b * Tp + i0n + 4 * lane : Dp + (long long)b * Tp + i0n + 4 * (lane - 8)), dst);
  };
  const T16* Qb = Qh + (long long)b * Tp * C;
  const T16* Ob = dOh + (long long)b * Tp * C;
  const T16* Qtb = Qt + (long long)b * C * Tp;
  const T16* Otb = dOt + (long long)b * C * Tp;
  v8 kf[KS], vf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    kf[kk] = *reinterpret_cast<const v8*>(Kh + ((long long)b * Tp + krow) * C + 16 * kk + 8 * hi);
    vf[kk] = *reinterpret_cast<const v8*>(Vh + ((long long)b * Tp + krow) * C + 16 * kk + 8 * hi);
  }
  f32x16 gk[DT], gv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { gk[dt] = zero16(); gv[dt] = zero16(); }
  const int fk = swz<SLK>(lq), ft = swz<SLT>(lq);
  const TileOff<BN, SLK> orow = tile_offsets<BN, SLK, T16>(C, w, lane);
  const TileOff<C, SLT> otr = tile_offsets<C, SLT, T16>(Tp, w, lane);
  stage_tile<BN, SLK, T16>(Qb, orow, smem, w);
  stage_tile<BN, SLK, T16>(Ob, orow, smem + RT, w);
  stage_tile<C, SLT, T16>(Qtb, otr, smem + 2 * RT, w);
  stage_tile<C, SLT, T16>(Otb, otr, smem + 3 * RT, w);
  stage_ld(0, smem + 4 * RT);
  int it = 0;
  for (int i0 = 0; i0 < T; i0 += BN, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i0 + BN < T) {
      char* nb = smem + ((it + 1) & 1) * BUF;
      stage_tile<BN, SLK, T16>(Qb + (long long)(i0 + BN) * C, orow, nb, w);
      stage_tile<BN, SLK, T16>(Ob + (long long)(i0 + BN) * C, orow, nb + RT, w);
      stage_tile<C, SLT, T16>(Qtb + i0 + BN, otr, nb + 2 * RT, w);
      stage_tile<C, SLT, T16>(Otb + i0 + BN, otr, nb + 3 * RT, w);
      stage_ld(i0 + BN, nb + 4 * RT);
    }
    const char* qb = smem + (it & 1) * BUF;
    const char* ob = qb + RT;
    const char* qtb = qb + 2 * RT;
    const char* otb = qb + 3 * RT;
    // the fragment addresses are recomputed per block (opaque copies of the swizzle terms): hoisted out of the loop they are ~24 loop-invariant VGPRs,
    // which this kernel (256 accumulators + 128 resident operand registers) pays for with spilled K / V fragments reloaded from scratch in every block
    int fk_ = fk, ft_ = ft;
    asm volatile("" : "+v"(fk_), "+v"(ft_));
    f32x16 sd[2] = {zero16(), zero16()};              // [0] S, [1] dP
#define LD_(i) frag_rows<SLK, T16>(((i) & 1) ? ob : qb, lq, fk_, (i) >> 1, hi)
#define MM_(i, f) sd[(i) & 1] = mma32(f, ((i) & 1) ? vf[(i) >> 1] : kf[(i) >> 1], sd[(i) & 1])
    FA16_PIPE(2 * KS, 4, LD_, MM_)
#undef LD_
#undef MM_
    const f32x16 s = sd[0], dp = sd[1];
    v8 pf[2], gf[2];
    // the 16 query rows of this lane's accumulator registers: i0 + 8 g + 4 hi + (0..3), g = 0..3 (pad rows: L2 = +inf -> P = 0)
    const char* ldt = qb + 4 * RT;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(ldt + (8 * g + 4 * hi) * 4), d4 = *reinterpret_cast<const float4*>(ldt + 128 + (8 * g + 4 * hi) * 4);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const float p = fexp2(s[r] - lv[e]);
        pf[r >> 3][r & 7] = (T16)p;
        gf[r >> 3][r & 7] = (T16)(p * (dp[r] - dv4[e]));
      }
    }
    // items: (h, dt, which): dv^T[dt] += dO^T P, dk^T[dt] += Qs^T G; consecutive MFMAs on different accumulators
#define LD_(i) *reinterpret_cast<const v8*>((((i) & 1) ? qtb : otb) + (32 * (((i) >> 1) % DT) + lq) * (SLT * 16) + (((2 * ((i) / (2 * DT)) + hi) ^ ft_) << 4))
#define MM_(i, f) { if ((i) & 1) gk[((i) >> 1) % DT] = mma32(f, gf[(i) / (2 * DT)], gk[((i) >> 1) % DT]); else gv[((i) >> 1) % DT] = mma32(f, pf[(i) / (2 * DT)], gv[((i) >> 1) % DT]); }
    FA16_PIPE(4 * DT, 4, LD_, MM_)
#undef LD_
#undef MM_
  }
  if (krow < T) {
    float* krow_p = dk + ((long long)b * T + krow) * C;
    float* vrow_p = dv + ((long long)b * T + krow) * C;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(krow_p + 32 * dt + 8 * g + 4 * hi) =
            make_float4(gk[dt][4 * g] * kLn2, gk[dt][4 * g + 1] * kLn2, gk[dt][4 * g + 2] * kLn2, gk[dt][4 * g + 3] * kLn2);
        *reinterpret_cast<float4*>(vrow_p + 32 * dt + 8 * g + 4 * hi) = make_float4(gv[dt][4 * g], gv[dt][4 * g + 1], gv[dt][4 * g + 2], gv[dt][4 * g + 3]);
      }
  }
}

template <typename T16>
void cvt16(const float* src, T16* rows, T16* trans, int B, int T, int Tp, int C, float mult, hipStream_t st) {
  hipLaunchKernelGGL(cvt16_kernel<T16>, dim3(Tp / 64, C / 64, B), dim3(256), 0, st, src, rows, trans, T, Tp, C, mult);
}
template <typename T16>
void fwd16(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, void* ws, hipStream_t st) {
  const int Tp = cdiv(T, TPAD) * TPAD;
  const long long n = (long long)B * Tp * C;
  T16* Qh = (T16*)ws; T16* Kh = Qh + n; T16* Vt = Kh + n;
  cvt16<T16>(q, Qh, nullptr, B, T, Tp, C, scale * kLog2e, st);
  cvt16<T16>(k, Kh, nullptr, B, T, Tp, C, 1.f, st);
  cvt16<T16>(v, nullptr, Vt, B, T, Tp, C, 1.f, st);
  const dim3 grid(cdiv(T, 128), B), block(256);
#define FA16_FWD(CC) hipLaunchKernelGGL((fa16_fwd_kernel<CC, T16>), grid, block, 0, st, (const T16*)Qh, (const T16*)Kh, (const T16*)Vt, O, Lse, T, Tp)
  if (C == 64) FA16_FWD(64); else if (C == 128) FA16_FWD(128); else FA16_FWD(256);
#undef FA16_FWD
}
template <typename T16>
void bwd16(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq, float* dk, float* dv,
           int B, int T, int C, float scale, void* ws, hipStream_t st) {
  const int Tp = cdiv(T, TPAD) * TPAD;
  const long long n = (long long)B * Tp * C;
  T16* Qh = (T16*)ws; T16* Qt = Qh + n; T16* Kh = Qt + n; T16* Kt = Kh + n; T16* Vh = Kt + n; T16* dOh = Vh + n; T16* dOt = dOh + n;
  float* Dp = (float*)(dOt + n); float* L2p = Dp + (long long)B * Tp;
  cvt16<T16>(q, Qh, Qt, B, T, Tp, C, scale * kLog2e, st);
  cvt16<T16>(k, Kh, Kt, B, T, Tp, C, 1.f, st);
  cvt16<T16>(v, Vh, nullptr, B, T, Tp, C, 1.f, st);
  cvt16<T16>(dO, dOh, dOt, B, T, Tp, C, 1.f, st);
  const dim3 grid(cdiv(T, 128), B), block(256), gs((unsigned)(((long long)B * Tp + 3) / 4));
#define FA16_BWD(CC)                                                                                                                   \
  hipLaunchKernelGGL(stats16_kernel<CC>, gs, block, 0, st, dO, O, Lse, D, Dp, L2p, T, Tp, B);                                            \
  hipLaunchKernelGGL((fa16_dq_kernel<CC, T16>), grid, block, 0, st, (const T16*)Qh, (const T16*)Kh, (const T16*)Kt, (const T16*)Vh,       \
                     (const T16*)dOh, (const float*)L2p, (const float*)Dp, dq, T, Tp, scale);                                            \
  hipLaunchKernelGGL((fa16_dkv_kernel<CC, T16>), grid, block, 0, st, (const T16*)Qh, (const T16*)Qt, (const T16*)Kh, (const T16*)Vh,      \
                     (const T16*)dOh, (const T16*)dOt, (const float*)L2p, (const float*)Dp, dk, dv, T, Tp);
  if (C == 64) { FA16_BWD(64) } else if (C == 128) { FA16_BWD(128) } else { FA16_BWD(256) }
#undef FA16_BWD
}
}  // namespace

// floats of workspace of the 16-bit attention launchers (the larger, backward, need: seven operand arrays + two padded statistics rows)
long long flash_attn16_ws_floats(int B, int T, int C) {
  const long long Tp = (long long)cdiv(T, TPAD) * TPAD;
  return (7 * B * Tp * C * 2 + 2 * B * Tp * 4 + 3) / 4;
}
void launch_flash_attn16_fwd(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, int prec, float* ws,
                             hipStream_t st) {
  if (prec == 1) fwd16<__bf16>(q, k, v, O, Lse, B, T, C, scale, ws, st);
  else fwd16<_Float16>(q, k, v, O, Lse, B, T, C, scale, ws, st);
}
void launch_flash_attn16_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq,
                             float* dk, float* dv, int B, int T, int C, float scale, int prec, float* ws, hipStream_t st) {
  if (prec == 1) bwd16<__bf16>(q, k, v, O, dO, Lse, D, dq, dk, dv, B, T, C, scale, ws, st);
  else bwd16<_Float16>(q, k, v, O, dO, Lse, D, dq, dk, dv, B, T, C, scale, ws, st);
}

}  // namespace buddy
