// Winograd-domain batched GEMM  M[p] (tiles x Cout) = V[p] (tiles x Cin) . U[p]^T (Cout x Cin),  p < positions: the middle pass of the three-pass 3x3
// convolutions (reference ddpm_conv3x3, networks/ncsnpp_utils/layers.py:119-126) -- 94 % of the score network's FLOPs -- in two split arithmetics on the
// 16-bit matrix pipe: "f16x2" (the default since round 5: wgemm_f16x2_kernel / wgemm_f16x2_rt2_kernel, second half of this file) and "bf16x3" (this kernel;
// also the general form of the 1x1 convolutions / NIN layers / DFT GEMMs and the ResBlock skip path's fused GroupNorm-backward form in both modes).
//
// bf16x3 arithmetic.  Every fp32 operand is split EXACTLY into three bf16 terms by truncation, x = hi + mid + lo (8 + 8 + 8 significant bits), and
// the product is accumulated in fp32 as  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid  on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are
// exact in fp32; the dropped terms mid*lo, lo*mid, lo*lo are <= 2^-23 of a product, the fp32 rounding level).  Six bf16 MFMAs (32 cycles, 16 k)
// replace eight fp32 MFMAs (64 cycles, 2 k each): 2.67x fewer matrix-pipe cycles per multiply-add at the accuracy of the fp32 path
// (unit test vs fp64: the same 1e-4 bound as the fp32 kernel; measured in profiles/README.md).
//
// Structure (round 2's drop-in variant of the fp32 loop lost to LDS latency and two barriers per 48 MFMAs; this loop is built for the split):
//  * workgroup = 4 waves x 32 rows of V, all 128 columns (output channels) of one channel block: a wave's A operand (V rows) is PRIVATE, so it
//    goes global -> registers in MFMA fragment order, is split in registers (5.5 VALU per element, once) and never touches LDS;
//  * the k index an MFMA lane holds is free as long as A and B agree: lane (row r, half h) takes k = 32 s + 16 h + 0..15 of K-stage s, i.e. 64
//    contiguous bytes per lane, a whole 128-byte line per row and stage;
//  * the weights are split once, at handle creation (wgemm_pack_weights), into the exact LDS image of a stage: [k chunk][column block][plane][lane]
//    x 16 B, so a stage is a linear 24 KB copy and every fragment read is one conflict-free ds_read_b128;
//  * LDS is double-buffered: ONE barrier per K-stage of 48 MFMAs per wave; global loads for stage s + 1 are issued before the MFMAs of stage s;
//  * the accumulator is C^T (weights as the MFMA's first operand): a lane holds 4 consecutive output channels of one row -> 16-byte stores.
#include "common.h"
#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <cstdlib>

namespace buddy {
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WBM = 128, WBN = 128, WKS = 32, WNT = 256;
constexpr int STAGE_BYTES = WBN * WKS * 6;                    // 24 KB: 2 k-chunks x 4 column blocks x 3 planes x 1 KB
constexpr int FRAG = 1024;                                    // bytes of one (chunk, column block, plane) fragment block: 64 lanes x 16 B

struct Split3 { bf16x8 p[3]; };
// exact three-way split of 8 fp32 values by truncation (hi = top 16 bits; mid = top 16 bits of x - hi; lo = x - hi - mid, <= 8 significant bits)
__device__ __forceinline__ Split3 split3(const float4 a, const float4 b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned int h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned int u = __float_as_uint(x[i]);
    h[i] = u;
    const float r = x[i] - __uint_as_float(u & 0xFFFF0000u);
    m[i] = __float_as_uint(r);
    l[i] = __float_as_uint(r - __uint_as_float(m[i] & 0xFFFF0000u));
  }
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {                               // v_perm_b32: the high halves of two dwords -> one packed pair
    ph[q] = __builtin_amdgcn_perm(h[2 * q + 1], h[2 * q], 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(m[2 * q + 1], m[2 * q], 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(l[2 * q + 1], l[2 * q], 0x07060302u);
  }
  Split3 s;
  s.p[0] = (bf16x8)ph; s.p[1] = (bf16x8)pm; s.p[2] = (bf16x8)pl;
  return s;
}

// U fp32 [P][Cout][Cin] -> stage images [P][Cout/128][Cin/32][2][4][3][64] x 16 B; one thread per 16-byte element
__global__ __launch_bounds__(256) void wgemm_pack_kernel(const float* __restrict__ U, u32x4* __restrict__ out, int P, int Cout, int Cin) {
  const long long n16 = (long long)P * Cout * Cin * 6 / 16;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const int lane = (int)(i & 63);
  long long r = i >> 6;
  const int q = (int)(r % 3); r /= 3;
  const int cb = (int)(r & 3); r >>= 2;
  const int kc = (int)(r & 1); r >>= 1;
  const int S = Cin / WKS, NB = Cout / WBN;
  const int s = (int)(r % S); r /= S;
  const int nb = (int)(r % NB); r /= NB;
  const int p = (int)r;
  const int n = nb * WBN + cb * 32 + (lane & 31), k = s * WKS + 16 * (lane >> 5) + 8 * kc;
  const float* src = U + ((long long)p * Cout + n) * Cin + k;
  const Split3 sp = split3(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
  out[i] = (u32x4)sp.p[q];
}

struct WgemmArgs {
  const float* V; const unsigned char* U3; float* M;
  int Mt, Cin, Cout, S, NB;                                    // rows per position, K, N, K-stages, column blocks
  int pz, gx;                                                  // pz > 0: positions folded into a 1-D grid (pz positions x gx workgroups each), XCD x owns positions x mod 8
  long long sV, sM;                                            // strides between positions (floats)
  // general form (GEN = true; 1x1 convolutions, NIN): A from up to two sources (channel concatenation, split at C0), row strides, C = alpha * A W^T
  // + bias [+ C]
  const float* A1; int C0, ldA0, ldA1, ldC; const float* bias_n; float alpha; int accumulate;
  // GNB epilogue (general form): C = alpha * A W^T + the GroupNorm backward's apply pass, written to a two-destination view
  Src2 gxv; Dst2 gd; const float* gda; const float* gstats; const float* gred; const float* ggamma; const float* gbeta; int gG, gsilu, gHW;
  // f16x2 form: per-utterance abs-max of V (float bits, written by the input transform), rows per utterance, per-position inverse weight scales
  const unsigned* vmax; int tpu; const float* uinv;
  int gnt;                                                     // GNB epilogue of the f16x2 forms: x and da (each read once, whole lines per instruction) by non-temporal loads
};
typedef float f32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt_g(const float* p) { const f32x4g v = __builtin_nontemporal_load(reinterpret_cast<const f32x4g*>(p)); return make_float4(v.x, v.y, v.z, v.w); }

__device__ __forceinline__ float dsilu_g(float z) {
  const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// GEN: the general form (two-source A, row strides, alpha / bias / accumulate epilogue).  EPI: the accumulator tile goes through a wave-private LDS
// slab (the weight buffers are free after the last stage) and leaves as 256-byte row pieces instead of 32-byte pieces per lane pair.
// Measured and rejected (profiles/README.md r03a): a second stage of A in flight (190 VGPRs: -3 %), 64 rows per wave (256+ VGPRs: -25 %), two
// instead of three workgroups per CU (-2...4 %).
// GNB (with GEN): the skip path's 1x1 data-gradient of a ResBlock (Conv_2^T, layerspp.py:262-264) and the GroupNorm_0 backward's apply pass in ONE launch:
// out[row][c] = alpha * (A W^T)[row][c] + rstd * (dxhat - m1 - xhat * m2), dxhat = da * act'(z) * gamma -- the block's whole input gradient.  The separate
// apply kernel read x, da and the materialised 1x1 result and wrote dx (5 C-channel streams with the GEMM's store); here the epilogue reads x and da and
// writes dx (3).  x / out are two-source / two-destination channel views split at a multiple of 128 (a column block never straddles).
// Measured and rejected in round 5 (profiles/README.md): the weight stage by LDS-DMA instead of through 24 staging registers + six ds_write_b128 (152
// instead of 168 VGPRs, +-0.5 %); s_setprio(1) around the MFMA bursts (the builtin fences hipcc's read / MFMA interleave: 0.50 -> 0.76 ms); any run-time
// branch inside the K loop (same effect: 65 -> 81 ms/step); a from-scratch large-tile kernel (tools/probes/wgemm2_large_tile.hip: 256-row workgroups, one wave
// per SIMD, both operands by LDS-DMA, cross-stage split pipelining): bit-identical and within +-5 % on every shape -- two unrelated structures, one
// throughput: the shape is bounded by the power budget of its instruction mix (pipe 49 % busy at 2.06 GHz; a register-only loop: 100 % at 1.57 GHz).
template <bool GEN, bool EPI, bool GNB = false>
__global__ __launch_bounds__(WNT, 3) void wgemm_bf16x3_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware order (hardware places workgroup b of the flattened grid on XCD b % 8, each XCD with its own L2).
  //  * batched form, positions a multiple of 8 (a.pz > 0: 1-D grid of pz * tiles * NB workgroups): XCD x takes the positions x, x + 8, ... and walks
  //    all their tiles, so a position's weight panel (K x 128 x 6 B per column block) is fetched into ONE L2 instead of all eight (r03 PMC:
  //    the GEMM fetched 1.42x its V bytes; 8 x 25 MB of panels per 256 -> 256 convolution were most of the excess);
  //  * otherwise (blockIdx.z = position): each XCD gets a contiguous range of logical tiles.
  // In both, the column blocks of one row tile are adjacent, so the second column block finds its V rows in the same L2.
  int lid, p;
  if (a.pz > 0) {
    const int orig = blockIdx.x, xcd = orig & 7, k = orig >> 3;     // k-th workgroup of this XCD: gx * pz / 8 of them
    lid = k % a.gx; p = xcd + 8 * (k / a.gx);
  } else {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    p = blockIdx.z;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * WBM;
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const unsigned char* __restrict__ U3 = a.U3 + ((long long)p * a.NB + nb) * a.S * STAGE_BYTES;
  const int S = a.S;

  // A: lane (row r = lane & 31, half h = lane >> 5) reads 16 consecutive floats per stage; rows past M are clamped (never stored)
  int row = m0 + wid * 32 + (lane & 31);
  const bool row_ok = row < a.Mt;
  if (!row_ok) row = a.Mt - 1;
  const float* Ap = V + (long long)row * (GEN ? a.ldA0 : a.Cin) + 16 * (lane >> 5);
  const float* Ap1 = (GEN && a.A1) ? a.A1 + (long long)row * a.ldA1 + 16 * (lane >> 5) - a.C0 : nullptr;   // channels >= C0 come from the second source
  // B: the stage image is copied linearly, 6 x 16 B per thread (thread t moves bytes 16 t + 4096 j)
  const u32x4* Bg = reinterpret_cast<const u32x4*>(U3) + tid;
  u32x4* Bs = reinterpret_cast<u32x4*>(smem) + tid;

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float4 ra[4];
  u32x4 rb[6];
  auto loadA = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4*>(((GEN && Ap1 && s * WKS >= a.C0) ? Ap1 : Ap) + s * WKS + 4 * j);
  };
  auto loadB = [&](int s) {
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[j] = Bg[(long long)s * (STAGE_BYTES / 16) + j * WNT];
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 6; ++j) Bs[buf * (STAGE_BYTES / 16) + j * WNT] = rb[j];
  };

  loadA(0);
  loadB(0);
  storeB(0);
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    const float4 ca[4] = {ra[0], ra[1], ra[2], ra[3]};
    if (s + 1 < S) { loadA(s + 1); loadB(s + 1); }           // stage s + 1 is in flight under the 48 MFMAs of stage s
    const unsigned char* Bcur = smem + (s & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const Split3 av = split3(ca[2 * kc], ca[2 * kc + 1]);
      bf16x8 b[4][3];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[cb][q] = *reinterpret_cast<const bf16x8*>(Bcur + ((kc * 4 + cb) * 3 + q) * FRAG);
      // smallest terms first (mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi; B plane, A plane); the accumulators interleaved so that consecutive
      // MFMAs never share one
      constexpr int PB[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][PB[t]], av.p[PA[t]], acc[cb], 0, 0, 0);
    }
    if (s + 1 < S) storeB((s + 1) & 1);                        // that buffer was last read in stage s - 1: every wave is past its barrier
    __syncthreads();
  }

  // epilogue: accumulator = C^T tile, lane (row = lane & 31, h = lane >> 5) holds channels 8 g + 4 h + 0..3 of each 32-channel block
  if (EPI && !GNB) {
    constexpr int SP = 68;                                   // floats per staged row (64 columns + 4: conflict-free 16-byte writes down a column)
    float* St = reinterpret_cast<float*>(smem) + wid * (32 * SP);
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    const long long ldc = GEN ? a.ldC : a.Cout;
    float* Mrow = a.M + (long long)p * a.sM + (long long)(m0 + wid * 32) * ldc + nb * WBN;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[2 * hb + cl][4 * g], acc[2 * hb + cl][4 * g + 1], acc[2 * hb + cl][4 * g + 2], acc[2 * hb + cl][4 * g + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (GEN) {                                              // general form: alpha * acc, + bias, + C (same operation order as the direct-store epilogue)
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias_n) bs = *reinterpret_cast<const float4*>(a.bias_n + nb * WBN + hb * 64 + c4);
        float4 pv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {                       // previous values (accumulate) in flight together, rows clamped
          const int r = min(4 * it + rr, a.Mt - 1 - (m0 + wid * 32));
          pv[it] = a.accumulate ? *reinterpret_cast<const float4*>(Mrow + (long long)r * ldc + hb * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r = 4 * it + rr;
          float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
          v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
          if (a.bias_n) { v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w; }
          if (a.accumulate) { v.x += pv[it].x; v.y += pv[it].y; v.z += pv[it].z; v.w += pv[it].w; }
          if (m0 + wid * 32 + r < a.Mt) *reinterpret_cast<float4*>(Mrow + (long long)r * ldc + hb * 64 + c4) = v;
        }
      } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rr;
        const float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
        if (m0 + wid * 32 + r < a.Mt) *reinterpret_cast<float4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4) = v;
      }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if constexpr (GNB) {
    // The accumulator tile goes through the wave-private LDS slab of the EPI epilogue so that a lane owns 16-byte pieces of 256-byte ROW pieces
    // (16 lanes per row): the x / da loads and the dx stores are whole cache lines (in MFMA order a lane pair covers 32 bytes of 32 rows).
    constexpr int SP = 68;
    float* St = reinterpret_cast<float*>(smem) + wid * (32 * SP);
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    const int rbase = m0 + wid * 32;
    const int cpg = a.Cout / a.gG;
    const bool second = a.gxv.p1 != nullptr && nb * WBN >= a.gxv.C0, dsecond = a.gd.p1 != nullptr && nb * WBN >= a.gd.C0;
    const long long ldx = second ? a.gxv.ld1 : a.gxv.ld0, ldo = dsecond ? a.gd.ld1 : a.gd.ld0;
    const bool accd = (dsecond ? a.gd.acc1 : a.gd.acc0) != 0;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int col = nb * WBN + hb * 64 + c4, grp = col / cpg;
      const float* xs = (second ? a.gxv.p1 + (col - a.gxv.C0) : a.gxv.p0 + col);
      float* o = (dsecond ? a.gd.p1 + (col - a.gd.C0) : a.gd.p0 + col);
      const float* das = a.gda + col;
      const float4 gm = *reinterpret_cast<const float4*>(a.ggamma + col), bt = *reinterpret_cast<const float4*>(a.gbeta + col);
      const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[2 * hb + cl][4 * g], acc[2 * hb + cl][4 * g + 1], acc[2 * hb + cl][4 * g + 2], acc[2 * hb + cl][4 * g + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // 16 rows at a time: the loads of the batch first (rows clamped: never stored past Mt), then the arithmetic (registers: 12 float4 + 16 scalars)
#pragma unroll
      for (int bt4 = 0; bt4 < 2; ++bt4) {
        float4 xv[4], dv[4], pv[4];
        float mean[4], rstd[4], m1[4], m2[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int r = min(rbase + 16 * bt4 + 4 * it + rr, a.Mt - 1), bb = r / a.gHW;
          xv[it] = *reinterpret_cast<const float4*>(xs + (long long)r * ldx);
          dv[it] = *reinterpret_cast<const float4*>(das + (long long)r * a.Cout);
          pv[it] = accd ? *reinterpret_cast<const float4*>(o + (long long)r * ldo) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float2 sm = *reinterpret_cast<const float2*>(a.gstats + ((long long)bb * a.gG + grp) * 2), rm = *reinterpret_cast<const float2*>(a.gred + ((long long)bb * a.gG + grp) * 2);
          mean[it] = sm.x; rstd[it] = sm.y; m1[it] = rm.x; m2[it] = rm.y;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = 16 * bt4 + 4 * it + rr;
          const float4 cv = *reinterpret_cast<const float4*>(St + rl * SP + c4);
          const float x4[4] = {xv[it].x, xv[it].y, xv[it].z, xv[it].w}, d4[4] = {dv[it].x, dv[it].y, dv[it].z, dv[it].w};
          const float p4[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w}, c4v[4] = {cv.x, cv.y, cv.z, cv.w};
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {                         // same operation order as gn_bwd_apply_m0_kernel: GroupNorm term, + extra_scale * extra, + previous
            const float xh = (x4[j] - mean[it]) * rstd[it];
            const float z = xh * g4[j] + b4[j];
            const float dxh = d4[j] * (a.gsilu ? dsilu_g(z) : 1.f) * g4[j];
            r[j] = rstd[it] * (dxh - m1[it] - xh * m2[it]);
            r[j] += a.alpha * c4v[j];
            r[j] += p4[j];
          }
          if (rbase + rl < a.Mt) *reinterpret_cast<float4*>(o + (long long)(rbase + rl) * ldo) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if (!row_ok) return;
  float* dst = a.M + (long long)p * a.sM + (long long)row * (GEN ? a.ldC : a.Cout) + nb * WBN + 4 * (lane >> 5);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
      if (GEN) {     // same operation order as the fp32 kernel's epilogue: alpha * acc, + bias, + C
        v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
        if (a.bias_n) { const float4 t = *reinterpret_cast<const float4*>(a.bias_n + nb * WBN + 4 * (lane >> 5) + cb * 32 + 8 * g); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (a.accumulate) { const float4 t = *reinterpret_cast<const float4*>(dst + cb * 32 + 8 * g); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      }
      *reinterpret_cast<float4*>(dst + cb * 32 + 8 * g) = v;
    }
}

// ------------------------------------------------------------------------------------------------ f16x2 form of the batched GEMM
// Arithmetic.  Every fp32 operand is scaled by a power of two (exact) and split into TWO f16 terms by round-to-nearest, x s = hi + lo (11 + 11
// significant bits: |x s - hi - lo| <= 2^-23 |x s|), and the product is accumulated in fp32 as  lo*hi + hi*lo + hi*hi  on v_mfma_f32_32x32x16_f16 (f16 x f16
// products are exact in fp32; the dropped term lo*lo is <= 2^-22 of a product).  Three MFMAs per 16 k instead of six: half the matrix-pipe cycles of bf16x3 and
// 4 instead of 6 operand bytes per weight, for operands good to 2^-22 instead of 2^-24 -- against float64 a whole F(6x6,3x3) convolution moves from 117 dB
// (exact products) to 114 dB, the level of plain fp32 accumulation.  f16 has 5 exponent bits, hence the scales: the weights carry one power of
// two per position (their abs-max -> [2^14, 2^15), applied when packing; the inverse is in the image's tail), V one per UTTERANCE (abs-max collected by the
// input transform with an atomic max, the power of two derived here from its exponent field), so that an utterance's result does not depend on its batch;
// lo is a normal f16 for |x s| >= 2^-3, i.e. 2^-18 of the abs-max; below that the representation error is absolute, <= 2^-25 = 2^-40 of the abs-max.
// Same workgroup / wave tiling, staging, XCD order and epilogue as wgemm_bf16x3_kernel<false, true>; a stage image is 16 KB ([k chunk][column block][plane]
// [lane] x 16 B), 24 MFMAs per wave and barrier.
// LDS-DMA of 16 bytes per lane as inline asm: source = uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset, LDS destination = M0 + 16 * lane
__device__ __forceinline__ void glds16_asm(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
constexpr int STAGE2_BYTES = WBN * WKS * 4;                   // 16 KB: 2 k-chunks x 4 column blocks x 2 planes x 1 KB
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
struct Split2 { f16x8 p[2]; };
// x s = hi + lo by round-to-nearest.  Measured and rejected (round 5, same box A/B): lo as ONE v_fma_mixlo/hi_f16 per element through inline asm (64 instead of
// 116 VALU instructions per 48 MFMAs; bit-identical) -- 1.5 % SLOWER on the 256-channel shapes: the vector ALU is not what the kernel waits for.
__device__ __forceinline__ Split2 split2(const float4 a, const float4 b, float s) {
  const float x[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
  Split2 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)x[i];
    r.p[0][i] = h;
    r.p[1][i] = (_Float16)(x[i] - (float)h);
  }
  return r;
}
// the power of two that takes an abs-max (float bits) into [2^14, 2^15), and its inverse; exponent fields outside [15, 253] are clamped (zero / tiny / huge
// tensors: the scale stays a finite normal number)
__device__ __forceinline__ void pow2_scale(unsigned bits, float& s, float& inv) {
  int e = (int)((bits >> 23) & 0xFF);
  e = e < 15 ? 15 : (e > 253 ? 253 : e);
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 14) << 23);
}

// abs-max of every position's weight matrix -> umax[p] (float bits); grid (chunks, P), umax zeroed before
__global__ __launch_bounds__(256) void wgemm_umax_kernel(const float* __restrict__ U, unsigned* __restrict__ umax, long long per) {
  const float* u = U + (long long)blockIdx.y * per;
  float m = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < per; i += (long long)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(u + i);                      // per = Cout * Cin, a multiple of 4
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(umax + blockIdx.y, __float_as_uint(m));
}
// out[u][VMAX_SUB][VMAX_STRIDE]: partial maxima (bit patterns) of |x| over x[g][u][0 .. seg_len), g < groups; grid (chunks, segments), out zeroed before
__global__ __launch_bounds__(256) void abs_max_bits_kernel(const float* __restrict__ x, int groups, int segments, long long seg_len, unsigned* __restrict__ out) {
  const int u = blockIdx.y;
  float m = 0.f;
  for (int g = 0; g < groups; ++g) {
    const float* s = x + ((long long)g * segments + u) * seg_len;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < seg_len; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(s[i]));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out + ((long long)u * VMAX_SUB + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (VMAX_SUB - 1))) * VMAX_STRIDE, __float_as_uint(m));
}
// U fp32 [P][Cout][Cin] -> stage images [P][Cout/128][Cin/32][2][4][2][64] x 16 B, then P inverse scales (floats); one thread per 16-byte element
__global__ __launch_bounds__(256) void wgemm_pack2_kernel(const float* __restrict__ U, u32x4* __restrict__ out, const unsigned* __restrict__ umax, int P, int Cout,
                                                          int Cin) {
  const long long n16 = (long long)P * Cout * Cin * 4 / 16;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < P) { float s, inv; pow2_scale(umax[i], s, inv); reinterpret_cast<float*>(out + n16)[i] = inv; }
  if (i >= n16) return;
  const int lane = (int)(i & 63);
  long long r = i >> 6;
  const int q = (int)(r & 1); r >>= 1;
  const int cb = (int)(r & 3); r >>= 2;
  const int kc = (int)(r & 1); r >>= 1;
  const int S = Cin / WKS, NB = Cout / WBN;
  const int s = (int)(r % S); r /= S;
  const int nb = (int)(r % NB); r /= NB;
  const int p = (int)r;
  const int n = nb * WBN + cb * 32 + (lane & 31), k = s * WKS + 16 * (lane >> 5) + 8 * kc;
  const float* src = U + ((long long)p * Cout + n) * Cin + k;
  float sc, inv; pow2_scale(umax[p], sc, inv);
  const Split2 sp = split2(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), sc);
  out[i] = (u32x4)sp.p[q];
}

template <bool EPI>
__global__ __launch_bounds__(WNT, 3) void wgemm_f16x2_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * STAGE2_BYTES > 4 * 32 * 68 * 4) ? 2 * STAGE2_BYTES : 4 * 32 * 68 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid, p;                                                 // XCD-aware order: see wgemm_bf16x3_kernel
  if (a.pz > 0) {
    const int orig = blockIdx.x, xcd = orig & 7, k = orig >> 3;
    lid = k % a.gx; p = xcd + 8 * (k / a.gx);
  } else {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    p = blockIdx.z;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * WBM;
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const unsigned char* __restrict__ U2 = a.U3 + ((long long)p * a.NB + nb) * a.S * STAGE2_BYTES;
  const int S = a.S;

  int row = m0 + wid * 32 + (lane & 31);
  const bool row_ok = row < a.Mt;
  if (!row_ok) row = a.Mt - 1;
  // the utterance's abs-max = the maximum of its VMAX_SUB partial words: one word per lane, a wave-wide maximum; a wave's 32 rows touch at most two utterances
  // (tpu >= 32)
  float sv, inv;
  {
    const int r0 = min(m0 + wid * 32, a.Mt - 1), r1 = min(m0 + wid * 32 + 31, a.Mt - 1), b0 = r0 / a.tpu, b1 = r1 / a.tpu;
    // agent-scope atomic loads: the words were written by agent-scope atomics (executed at the memory side); a plain load may hit a stale line of this XCD's L2
    unsigned mb0 = __hip_atomic_load(a.vmax + ((long long)b0 * VMAX_SUB + lane) * VMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) mb0 = max(mb0, (unsigned)__shfl_xor((int)mb0, o));
    unsigned mb1 = mb0;
    if (b1 != b0) {
      mb1 = __hip_atomic_load(a.vmax + ((long long)b1 * VMAX_SUB + lane) * VMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int o = 32; o > 0; o >>= 1) mb1 = max(mb1, (unsigned)__shfl_xor((int)mb1, o));
    }
    pow2_scale(row / a.tpu == b0 ? mb0 : mb1, sv, inv);
  }
  inv *= a.uinv[p];
  // uniform bases + 32-bit per-lane byte offsets: the loads are `global_load_dwordx4 v, v_off, s[base]` -- no 64-bit per-lane address arithmetic per stage
  const unsigned aoff = (unsigned)(((long long)row * a.Cin + 16 * (lane >> 5)) * 4), boff = (unsigned)tid * 16u;
  const char* Vb = reinterpret_cast<const char*>(V);
  const char* Ub = reinterpret_cast<const char*>(U2);

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // Two K-stages of A in flight (the stage is half as long as bf16x3's: one stage of prefetch no longer covers the HBM latency), the loop unrolled by two
  // so that each register set keeps its name.  The weights' stage image goes global -> LDS by LDS-DMA written as inline asm (no staging registers, no
  // ds_write pass; invisible to hipcc's wait-count pass, which would otherwise drain every load in flight before the first ds_read): it is requested BEFORE the
  // A loads of the stage and waited for by ONE counted `s_waitcnt vmcnt(4)` in front of the barrier, which leaves those four A loads in flight.
  float4 ra[2][4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  auto loadA = [&](int s, float4 (&r)[4]) {
    const char* base = Vb + (long long)s * (WKS * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const float4*>(base + aoff + 16 * j);
  };
  auto dmaB = [&](int s) {
    const void* base = uniform_ptr(Ub + (long long)s * STAGE2_BYTES);
    const unsigned l = lds0 + (s & 1) * STAGE2_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_asm(base, boff + j * (WNT * 16), l + j * (WNT * 16));
  };
  // NB / NA (compile time): this stage requests the weights of stage s + 1 / the A rows of stage s + 2 (the last two stages are peeled: behind a run-time `if`
  // the wait-count pass must assume the loads were skipped and waits for everything in flight).
  auto stage = [&](int s, float4 (&r)[4], auto nb_, auto na_) {
    constexpr bool NB = decltype(nb_)::value, NA = decltype(na_)::value;
    // the stage's A rows are split FIRST (their registers are then dead and the reload below lands in place: with a copy kept for later, hipcc renames
    // the reload's destination and moves it back at the loop's back edge -- behind a vmcnt(0))
    const Split2 av[2] = {split2(r[0], r[1], sv), split2(r[2], r[3], sv)};
    if (NB) dmaB(s + 1);
    if (NA) loadA(s + 2, r);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* Bcur = smem + (s & 1) * STAGE2_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      f16x8 b[4][2];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[cb][q] = *reinterpret_cast<const f16x8*>(Bcur + ((kc * 4 + cb) * 2 + q) * FRAG);
      constexpr int PB[3] = {1, 0, 0}, PA[3] = {0, 1, 0};     // smallest terms first: lo*hi, hi*lo, hi*hi (B plane, A plane)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cb][PB[t]], av[kc].p[PA[t]], acc[cb], 0, 0, 0);
    }
    if (NB) { if (NA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
  };

  dmaB(0);
  loadA(0, ra[0]);
  loadA(1, ra[1]);                                             // S is even (wgemm_f16x2_supported)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  int s = 0;
  for (; s + 2 < S; s += 2) {
    stage(s, ra[0], std::true_type{}, std::true_type{});
    stage(s + 1, ra[1], std::true_type{}, std::true_type{});
  }
  stage(s, ra[0], std::true_type{}, std::false_type{});
  stage(s + 1, ra[1], std::false_type{}, std::false_type{});
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] *= inv;               // the accumulator is C^T: a lane holds ONE row, so one (utterance, position) scale

  if (EPI) {
    constexpr int SP = 68;
    float* St = reinterpret_cast<float*>(smem) + wid * (32 * SP);
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    float* Mrow = a.M + (long long)p * a.sM + (long long)(m0 + wid * 32) * a.Cout + nb * WBN;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[2 * hb + cl][4 * g], acc[2 * hb + cl][4 * g + 1], acc[2 * hb + cl][4 * g + 2], acc[2 * hb + cl][4 * g + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rr;
        const float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
        if (m0 + wid * 32 + r < a.Mt) *reinterpret_cast<float4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if (!row_ok) return;
  float* dst = a.M + (long long)p * a.sM + (long long)row * a.Cout + nb * WBN + 4 * (lane >> 5);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + cb * 32 + 8 * g) = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
}

// ------------------------------------------------------------------------------------------------ f16x2 form of the GENERAL GEMM (round 6)
// C (M x N) = alpha * [A0 | A1] (M x K) W^T + bias (+ C), or -- GNB -- the ResBlock skip path's data-gradient with the GroupNorm backward apply as its
// epilogue: the 1x1 convolutions / NIN layers / DFT GEMMs in the arithmetic of the Winograd-domain GEMMs (three f16 MFMA products instead of bf16x3's six).
// The A operand here is an activation nobody has measured: its power-of-two scale is taken PER ROW and found ON THE WAY -- every K-stage takes the abs-max
// of the row's 32 values it is about to split; while that stays below 2^15 under the current scale nothing happens, otherwise the row's accumulators are
// multiplied by the (exact) power of two that takes them to the new scale, which puts the stage's abs-max into [2^12, 2^13).  A row's result depends on
// nothing but the row, no pre-pass re-reads A (the first version's did, from beyond L2: it took away what the halved MFMA count gave), and the rescale is a
// wave-uniform branch taken a few times per row.  Structure = wgemm_f16x2_kernel<true> (32-row waves, LDS-DMA weight stages, two K-stages of A in flight)
// with the two-source A of wgemm_bf16x3_kernel<true, ...> and its epilogues.

// epilogues of wgemm_bf16x3_kernel<true, true, GNB> for one 32-row x 128-column accumulator tile of a wave (rows rbase ..., x the rows' inverse scale)
// through the wave-private LDS slab St, 256-byte row pieces
template <bool GNB>
__device__ __forceinline__ void f16x2_gen_epilogue(const WgemmArgs& a, const f32x16 (&acc)[4], const float inv, const int rbase, const int nb, float* St, const int lane) {
  constexpr int SP = 68;
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
  if constexpr (!GNB) {
    const long long ldc = a.ldC;
    float* Mrow = a.M + (long long)rbase * ldc + nb * WBN;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[2 * hb + cl][4 * g] * inv, acc[2 * hb + cl][4 * g + 1] * inv, acc[2 * hb + cl][4 * g + 2] * inv, acc[2 * hb + cl][4 * g + 3] * inv);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias_n) bs = *reinterpret_cast<const float4*>(a.bias_n + nb * WBN + hb * 64 + c4);
      float4 pv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = min(4 * it + rr, a.Mt - 1 - rbase);
        pv[it] = a.accumulate ? *reinterpret_cast<const float4*>(Mrow + (long long)r * ldc + hb * 64 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rr;
        float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
        v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
        if (a.bias_n) { v.x += bs.x; v.y += bs.y; v.z += bs.z; v.w += bs.w; }
        if (a.accumulate) { v.x += pv[it].x; v.y += pv[it].y; v.z += pv[it].z; v.w += pv[it].w; }
        if (rbase + r < a.Mt) *reinterpret_cast<float4*>(Mrow + (long long)r * ldc + hb * 64 + c4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    const int cpg = a.Cout / a.gG;
    const bool second = a.gxv.p1 != nullptr && nb * WBN >= a.gxv.C0, dsecond = a.gd.p1 != nullptr && nb * WBN >= a.gd.C0;
    const long long ldx = second ? a.gxv.ld1 : a.gxv.ld0, ldo = dsecond ? a.gd.ld1 : a.gd.ld0;
    const bool accd = (dsecond ? a.gd.acc1 : a.gd.acc0) != 0;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int col = nb * WBN + hb * 64 + c4, grp = col / cpg;
      const float* xs = (second ? a.gxv.p1 + (col - a.gxv.C0) : a.gxv.p0 + col);
      float* o = (dsecond ? a.gd.p1 + (col - a.gd.C0) : a.gd.p0 + col);
      const float* das = a.gda + col;
      const float4 gm = *reinterpret_cast<const float4*>(a.ggamma + col), bt = *reinterpret_cast<const float4*>(a.gbeta + col);
      const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[2 * hb + cl][4 * g] * inv, acc[2 * hb + cl][4 * g + 1] * inv, acc[2 * hb + cl][4 * g + 2] * inv, acc[2 * hb + cl][4 * g + 3] * inv);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int bt4 = 0; bt4 < 2; ++bt4) {
        float4 xv[4], dv[4], pv[4];
        float mean[4], rstd[4], m1[4], m2[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int r = min(rbase + 16 * bt4 + 4 * it + rr, a.Mt - 1), bb = r / a.gHW;
          if (a.gnt) { xv[it] = ld4nt_g(xs + (long long)r * ldx); dv[it] = ld4nt_g(das + (long long)r * a.Cout); }
          else { xv[it] = *reinterpret_cast<const float4*>(xs + (long long)r * ldx); dv[it] = *reinterpret_cast<const float4*>(das + (long long)r * a.Cout); }
          pv[it] = accd ? *reinterpret_cast<const float4*>(o + (long long)r * ldo) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float2 sm = *reinterpret_cast<const float2*>(a.gstats + ((long long)bb * a.gG + grp) * 2), rm = *reinterpret_cast<const float2*>(a.gred + ((long long)bb * a.gG + grp) * 2);
          mean[it] = sm.x; rstd[it] = sm.y; m1[it] = rm.x; m2[it] = rm.y;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int rl = 16 * bt4 + 4 * it + rr;
          const float4 cv = *reinterpret_cast<const float4*>(St + rl * SP + c4);
          const float x4[4] = {xv[it].x, xv[it].y, xv[it].z, xv[it].w}, d4[4] = {dv[it].x, dv[it].y, dv[it].z, dv[it].w};
          const float p4[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w}, c4v[4] = {cv.x, cv.y, cv.z, cv.w};
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (x4[j] - mean[it]) * rstd[it];
            const float z = xh * g4[j] + b4[j];
            const float dxh = d4[j] * (a.gsilu ? dsilu_g(z) : 1.f) * g4[j];
            r[j] = rstd[it] * (dxh - m1[it] - xh * m2[it]);
            r[j] += a.alpha * c4v[j];
            r[j] += p4[j];
          }
          if (rbase + rl < a.Mt) *reinterpret_cast<float4*>(o + (long long)(rbase + rl) * ldo) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <bool GNB>
__global__ __launch_bounds__(WNT, 3) void wgemm_f16x2_gen_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * STAGE2_BYTES > 4 * 32 * 68 * 4) ? 2 * STAGE2_BYTES : 4 * 32 * 68 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * WBM;
  const unsigned char* __restrict__ U2 = a.U3 + (long long)nb * a.S * STAGE2_BYTES;
  const int S = a.S;
  int row = m0 + wid * 32 + (lane & 31);
  if (row >= a.Mt) row = a.Mt - 1;
  const float* Ap0 = a.V + (long long)row * a.ldA0 + 16 * (lane >> 5);
  const float* Ap1 = a.A1 ? a.A1 + (long long)row * a.ldA1 + 16 * (lane >> 5) - a.C0 : Ap0;       // channels >= C0 come from the second source
  const int s1 = a.A1 ? a.C0 / WKS : S;                                                            // first K-stage of the second source
  // running per-row power of two (see the header comment): exponent field of the scale's reference magnitude, the scale itself
  int ecur = 15;
  float sv = __uint_as_float((unsigned)(266 - 15) << 23);
  const unsigned boff = (unsigned)tid * 16u;
  const char* Ub = reinterpret_cast<const char*>(U2);

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float4 ra[2][4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  auto loadA = [&](int s, float4 (&r)[4]) {
    const float* q = (s >= s1 ? Ap1 : Ap0) + s * WKS;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const float4*>(q + 4 * j);
  };
  auto dmaB = [&](int s) {
    const void* base = uniform_ptr(Ub + (long long)s * STAGE2_BYTES);
    const unsigned l = lds0 + (s & 1) * STAGE2_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_asm(base, boff + j * (WNT * 16), l + j * (WNT * 16));
  };
  auto stage = [&](int s, float4 (&r)[4], auto nb_, auto na_) {
    constexpr bool NB = decltype(nb_)::value, NA = decltype(na_)::value;
    {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[j].x), fabsf(r[j].y))), fmaxf(fabsf(r[j].z), fabsf(r[j].w)));
      mx = fmaxf(mx, __shfl_xor(mx, 32));                       // the row's other sixteen k of this stage
      const int es = min((int)(__float_as_uint(mx) >> 23), 253);
      const bool grow = es > ecur + 2;                          // the stage would leave [0, 2^15) under the current scale
      if (__any(grow)) {
        const int d = grow ? ecur - es : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[c][q] = ldexpf(acc[c][q], d);
        if (grow) { ecur = es; sv = __uint_as_float((unsigned)(266 - es) << 23); }
      }
    }
    const Split2 av[2] = {split2(r[0], r[1], sv), split2(r[2], r[3], sv)};
    if (NB) dmaB(s + 1);
    if (NA) loadA(s + 2, r);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* Bcur = smem + (s & 1) * STAGE2_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      f16x8 b[4][2];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[cb][q] = *reinterpret_cast<const f16x8*>(Bcur + ((kc * 4 + cb) * 2 + q) * FRAG);
      constexpr int PB[3] = {1, 0, 0}, PA[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cb][PB[t]], av[kc].p[PA[t]], acc[cb], 0, 0, 0);
    }
    if (NB) { if (NA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
  };
  dmaB(0);
  loadA(0, ra[0]);
  loadA(1, ra[1]);                                             // S is even (wgemm_f16x2_supported)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  int s = 0;
  for (; s + 2 < S; s += 2) {
    stage(s, ra[0], std::true_type{}, std::true_type{});
    stage(s + 1, ra[1], std::true_type{}, std::true_type{});
  }
  stage(s, ra[0], std::true_type{}, std::false_type{});
  stage(s + 1, ra[1], std::false_type{}, std::false_type{});
  const float inv = __uint_as_float((unsigned)(ecur - 12) << 23) * a.uinv[0];

  f16x2_gen_epilogue<GNB>(a, acc, inv, m0 + wid * 32, nb, reinterpret_cast<float*>(smem) + wid * (32 * 68), lane);
}

// The same with TWO column blocks per workgroup (128 rows x 256 columns, 32-row waves, 128 accumulators per lane, two workgroups per CU): with N = 256 the
// two workgroups of a row block each read -- and split -- the same A rows; measured in isolation (tools/gen_gemm_one.py with the stores and the matrix work
// taken out) that second read costs like a first one.  Here a row block's A is read and split once for both column blocks; the weight stage is 32 KB.
template <bool GNB>
__global__ __launch_bounds__(WNT, 2) void wgemm_f16x2_gencp_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * STAGE2_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int NP = a.NB >> 1, np = lid % NP, m0 = (lid / NP) * WBM;
  const unsigned char* __restrict__ U2 = a.U3 + (long long)(2 * np) * a.S * STAGE2_BYTES;       // column block 2 np; 2 np + 1 follows S stage images later
  const int S = a.S;
  int row = m0 + wid * 32 + (lane & 31);
  if (row >= a.Mt) row = a.Mt - 1;
  const float* Ap0 = a.V + (long long)row * a.ldA0 + 16 * (lane >> 5);
  const float* Ap1 = a.A1 ? a.A1 + (long long)row * a.ldA1 + 16 * (lane >> 5) - a.C0 : Ap0;
  const int s1 = a.A1 ? a.C0 / WKS : S;
  int ecur = 15;
  float sv = __uint_as_float((unsigned)(266 - 15) << 23);
  const unsigned boff = (unsigned)tid * 16u;
  const char* Ub = reinterpret_cast<const char*>(U2);

  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  float4 ra[2][4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  auto loadA = [&](int s, float4 (&r)[4]) {
    const float* q = (s >= s1 ? Ap1 : Ap0) + s * WKS;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const float4*>(q + 4 * j);
  };
  auto dmaB = [&](int s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const void* base = uniform_ptr(Ub + ((long long)t * S + s) * STAGE2_BYTES);
      const unsigned l = lds0 + ((s & 1) * 2 + t) * STAGE2_BYTES;
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16_asm(base, boff + j * (WNT * 16), l + j * (WNT * 16));
    }
  };
  auto stage = [&](int s, float4 (&r)[4], auto nb_, auto na_) {
    constexpr bool NB = decltype(nb_)::value, NA = decltype(na_)::value;
    {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[j].x), fabsf(r[j].y))), fmaxf(fabsf(r[j].z), fabsf(r[j].w)));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int es = min((int)(__float_as_uint(mx) >> 23), 253);
      const bool grow = es > ecur + 2;
      if (__any(grow)) {
        const int d = grow ? ecur - es : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = ldexpf(acc[t][c][q], d);
        if (grow) { ecur = es; sv = __uint_as_float((unsigned)(266 - es) << 23); }
      }
    }
    const Split2 av[2] = {split2(r[0], r[1], sv), split2(r[2], r[3], sv)};
    if (NB) dmaB(s + 1);
    if (NA) loadA(s + 2, r);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned char* Bcur = smem + ((s & 1) * 2 + t) * STAGE2_BYTES + lane * 16;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        f16x8 b[4][2];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int q = 0; q < 2; ++q) b[cb][q] = *reinterpret_cast<const f16x8*>(Bcur + ((kc * 4 + cb) * 2 + q) * FRAG);
        constexpr int PB[3] = {1, 0, 0}, PA[3] = {0, 1, 0};
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cb][PB[tm]], av[kc].p[PA[tm]], acc[t][cb], 0, 0, 0);
      }
    }
    if (NB) { if (NA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
  };
  dmaB(0);
  loadA(0, ra[0]);
  loadA(1, ra[1]);                                             // S is even (wgemm_f16x2_supported)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  int s = 0;
  for (; s + 2 < S; s += 2) {
    stage(s, ra[0], std::true_type{}, std::true_type{});
    stage(s + 1, ra[1], std::true_type{}, std::true_type{});
  }
  stage(s, ra[0], std::true_type{}, std::false_type{});
  stage(s + 1, ra[1], std::false_type{}, std::false_type{});
  const float inv = __uint_as_float((unsigned)(ecur - 12) << 23) * a.uinv[0];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    f16x2_gen_epilogue<GNB>(a, acc[t], inv, m0 + wid * 32, 2 * np + t, reinterpret_cast<float*>(smem) + wid * (32 * 68), lane);
}

// The same with 64 rows per wave (the tiling of wgemm_f16x2_rt2_kernel below: workgroup = 256 rows x 128 columns, 128 accumulators per lane, two workgroups
// per CU, one K-stage of A in flight): a weight stage is fetched once per 256 rows instead of once per 128.
template <bool GNB>
__global__ __launch_bounds__(WNT, 2) void wgemm_f16x2_gen2_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * STAGE2_BYTES > 4 * 32 * 68 * 4) ? 2 * STAGE2_BYTES : 4 * 32 * 68 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * (2 * WBM);
  const unsigned char* __restrict__ U2 = a.U3 + (long long)nb * a.S * STAGE2_BYTES;
  const int S = a.S;
  const float* Ap0[2];
  const float* Ap1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = min(m0 + wid * 64 + t * 32 + (lane & 31), a.Mt - 1);
    Ap0[t] = a.V + (long long)row * a.ldA0 + 16 * (lane >> 5);
    Ap1[t] = a.A1 ? a.A1 + (long long)row * a.ldA1 + 16 * (lane >> 5) - a.C0 : Ap0[t];
  }
  const int s1 = a.A1 ? a.C0 / WKS : S;
  int ecur[2] = {15, 15};
  float sv[2] = {__uint_as_float((unsigned)(266 - 15) << 23), __uint_as_float((unsigned)(266 - 15) << 23)};
  const unsigned boff = (unsigned)tid * 16u;
  const char* Ub = reinterpret_cast<const char*>(U2);

  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  float4 ra[2][4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  auto loadA = [&](int s) {
    const bool snd = s >= s1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* q = (snd ? Ap1[t] : Ap0[t]) + s * WKS;
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[t][j] = *reinterpret_cast<const float4*>(q + 4 * j);
    }
  };
  auto dmaB = [&](int s) {
    const void* base = uniform_ptr(Ub + (long long)s * STAGE2_BYTES);
    const unsigned l = lds0 + (s & 1) * STAGE2_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_asm(base, boff + j * (WNT * 16), l + j * (WNT * 16));
  };
  auto stage = [&](int s, auto nx_) {
    constexpr bool NX = decltype(nx_)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ra[t][j].x), fabsf(ra[t][j].y))), fmaxf(fabsf(ra[t][j].z), fabsf(ra[t][j].w)));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int es = min((int)(__float_as_uint(mx) >> 23), 253);
      const bool grow = es > ecur[t] + 2;
      if (__any(grow)) {
        const int d = grow ? ecur[t] - es : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[t][c][q] = ldexpf(acc[t][c][q], d);
        if (grow) { ecur[t] = es; sv[t] = __uint_as_float((unsigned)(266 - es) << 23); }
      }
    }
    Split2 av[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) av[t][kc] = split2(ra[t][2 * kc], ra[t][2 * kc + 1], sv[t]);
    if (NX) { dmaB(s + 1); loadA(s + 1); }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* Bcur = smem + (s & 1) * STAGE2_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      f16x8 b[4][2];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[cb][q] = *reinterpret_cast<const f16x8*>(Bcur + ((kc * 4 + cb) * 2 + q) * FRAG);
      constexpr int PB[3] = {1, 0, 0}, PA[3] = {0, 1, 0};
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cb][PB[tm]], av[t][kc].p[PA[tm]], acc[t][cb], 0, 0, 0);
    }
    if (NX) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
  };
  dmaB(0);
  loadA(0);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  int s = 0;
  for (; s + 1 < S; ++s) stage(s, std::true_type{});
  stage(s, std::false_type{});
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float inv = __uint_as_float((unsigned)(ecur[t] - 12) << 23) * a.uinv[0];
    f16x2_gen_epilogue<GNB>(a, acc[t], inv, m0 + wid * 64 + t * 32, nb, reinterpret_cast<float*>(smem) + wid * (32 * 68), lane);
  }
}

// 64 rows per wave (two 32-row tiles; workgroup = 256 rows x 128 columns, 128 accumulators per lane, two workgroups per CU): every weight fragment read from LDS
// feeds two MFMAs per product term instead of one (the 32-row form reads 0.67 fragments per MFMA) and a barrier separates 48 instead of 24 MFMAs per wave.  One K-stage of A in flight (a stage is 48 MFMAs per wave), reloaded in place after the split; the rest as wgemm_f16x2_kernel<true>.
// NT (A/B switch wgemm_nt): bit 0 = the V rows are read with non-temporal loads (V is read exactly once: it should not displace the weight panels from
// this XCD's L2), bit 1 = M leaves with non-temporal stores (its reader is the next launch, 0.4 - 1 GB later).
// Measured and rejected (round 6, profiles/README.md): a PERSISTENT form -- 2 x CUs workgroups walking (position, row block, column block) items with the K-stage
// pipeline running across items, so that an item's stores leave while the next item's first rows and weights are in flight: 422.8 us against 401.7 us per
// launch (29584 x 128 x 128 x 64, rocprofv3), +-1.5 % on the other shapes.  Taken apart in isolation the kernel's time is close to the SUM of its parts (whole
// 0.454 ms; V read + split only 0.176; + stores 0.16; + MFMAs 0.09; + weight DMA 0.07), but what fails to overlap is not one workgroup's phases.
// Also: 8 waves per workgroup (512 rows, one workgroup per CU: a weight stage fetched from L2 once per 512 rows, half the L2 -> LDS traffic): 2-15 % slower on six
// shapes (0.451 vs 0.431, 1.293 vs 1.263, 0.358 vs 0.310, 0.640 vs 0.594, 0.078 vs 0.075, 0.787 vs 0.713 ms).
typedef float f32x4nt __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(WNT, 2) void wgemm_f16x2_rt2_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * STAGE2_BYTES > 4 * 32 * 68 * 4) ? 2 * STAGE2_BYTES : 4 * 32 * 68 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid, p;
  if (a.pz > 0) {
    const int orig = blockIdx.x, xcd = orig & 7, k = orig >> 3;
    lid = k % a.gx; p = xcd + 8 * (k / a.gx);
  } else {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    p = blockIdx.z;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * (2 * WBM);
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const unsigned char* __restrict__ U2 = a.U3 + ((long long)p * a.NB + nb) * a.S * STAGE2_BYTES;
  const int S = a.S;

  int row[2];
  float sv[2], inv[2];
  unsigned aoff[2];
  {
    const int r0 = min(m0 + wid * 64, a.Mt - 1), r1 = min(m0 + wid * 64 + 63, a.Mt - 1), b0 = r0 / a.tpu, b1 = r1 / a.tpu;     // tpu >= 64: at most two utterances
    unsigned mb0 = __hip_atomic_load(a.vmax + ((long long)b0 * VMAX_SUB + lane) * VMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) mb0 = max(mb0, (unsigned)__shfl_xor((int)mb0, o));
    unsigned mb1 = mb0;
    if (b1 != b0) {
      mb1 = __hip_atomic_load(a.vmax + ((long long)b1 * VMAX_SUB + lane) * VMAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int o = 32; o > 0; o >>= 1) mb1 = max(mb1, (unsigned)__shfl_xor((int)mb1, o));
    }
    const float ui = a.uinv[p];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      row[t] = min(m0 + wid * 64 + t * 32 + (lane & 31), a.Mt - 1);
      pow2_scale(row[t] / a.tpu == b0 ? mb0 : mb1, sv[t], inv[t]);
      inv[t] *= ui;
      aoff[t] = (unsigned)(((long long)row[t] * a.Cin + 16 * (lane >> 5)) * 4);
    }
  }
  const unsigned boff = (unsigned)tid * 16u;
  const char* Vb = reinterpret_cast<const char*>(V);
  const char* Ub = reinterpret_cast<const char*>(U2);

  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  float4 ra[2][4];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + wid * 1024);
  auto loadA = [&](int s) {
    const char* base = Vb + (long long)s * (WKS * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (NT & 1) { const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(base + aoff[t] + 16 * j)); ra[t][j] = make_float4(v.x, v.y, v.z, v.w); }
        else ra[t][j] = *reinterpret_cast<const float4*>(base + aoff[t] + 16 * j);
      }
  };
  auto dmaB = [&](int s) {
    const void* base = uniform_ptr(Ub + (long long)s * STAGE2_BYTES);
    const unsigned l = lds0 + (s & 1) * STAGE2_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16_asm(base, boff + j * (WNT * 16), l + j * (WNT * 16));
  };
  auto stage = [&](int s, auto nx_) {
    constexpr bool NX = decltype(nx_)::value;                  // stage s + 1 exists: request its weights and A rows
    Split2 av[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) av[t][kc] = split2(ra[t][2 * kc], ra[t][2 * kc + 1], sv[t]);
    if (NX) { dmaB(s + 1); loadA(s + 1); }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* Bcur = smem + (s & 1) * STAGE2_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      f16x8 b[4][2];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 2; ++q) b[cb][q] = *reinterpret_cast<const f16x8*>(Bcur + ((kc * 4 + cb) * 2 + q) * FRAG);
      constexpr int PB[3] = {1, 0, 0}, PA[3] = {0, 1, 0};
#pragma unroll
      for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cb][PB[tm]], av[t][kc].p[PA[tm]], acc[t][cb], 0, 0, 0);
    }
    if (NX) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // the four weight DMAs have landed, the eight A loads stay in flight
    __syncthreads();
  };
  dmaB(0);
  loadA(0);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  int s = 0;
  for (; s + 1 < S; ++s) stage(s, std::true_type{});
  stage(s, std::false_type{});

  constexpr int SP = 68;
  float* St = reinterpret_cast<float*>(smem) + wid * (32 * SP);
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rb = m0 + wid * 64 + t * 32;
    float* Mrow = a.M + (long long)p * a.sM + (long long)rb * a.Cout + nb * WBN;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + (lane & 31) * SP + cl * 32 + 8 * g + 4 * (lane >> 5)) =
              make_float4(acc[t][2 * hb + cl][4 * g] * inv[t], acc[t][2 * hb + cl][4 * g + 1] * inv[t], acc[t][2 * hb + cl][4 * g + 2] * inv[t],
                          acc[t][2 * hb + cl][4 * g + 3] * inv[t]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rr;
        const float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
        if (rb + r < a.Mt) {
          if (NT & 2) __builtin_nontemporal_store(f32x4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4nt*>(Mrow + (long long)r * a.Cout + hb * 64 + c4));
          else *reinterpret_cast<float4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

bool wgemm_supported(int Cout, int Cin) { return Cout % WBN == 0 && Cin % WKS == 0; }
bool wgemm_f16x2_supported(int Cout, int Cin) { return Cout % WBN == 0 && Cin % (2 * WKS) == 0; }    // an even number of K-stages
size_t wgemm_packed_bytes(int P, int Cout, int Cin) { return (size_t)P * Cout * Cin * 6; }

void wgemm_pack_weights(const float* U_dev, void* U3_dev, int P, int Cout, int Cin, hipStream_t st) {
  const long long n16 = (long long)P * Cout * Cin * 6 / 16;
  hipLaunchKernelGGL(wgemm_pack_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, U_dev, reinterpret_cast<u32x4*>(U3_dev), P, Cout, Cin);
}

// general form: C (M x N, row stride ldC) = alpha * [A0 | A1] (M x K) . W^T + bias_n (+ C), W pre-split by wgemm_pack_weights(W, ., 1, N, K)
bool wgemm_general_supported(int N, int K, int C0, int ldA0, int ldA1, int ldC, const void* A0, const void* A1, const void* C, const void* bias) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return wgemm_supported(N, K) && C0 % WKS == 0 && ldA0 % 4 == 0 && ldA1 % 4 == 0 && ldC % 4 == 0 && al16(A0) && al16(A1) && al16(C) && al16(bias);
}
void launch_wgemm_bf16x3_general(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W3, float* C, int ldC, long long M, int N, int K,
                                 const float* bias_n, float alpha, int accumulate, hipStream_t st) {
  WgemmArgs a{};
  a.V = A0; a.U3 = reinterpret_cast<const unsigned char*>(W3); a.M = C;
  a.Mt = (int)M; a.Cin = K; a.Cout = N; a.S = K / WKS; a.NB = N / WBN; a.sV = 0; a.sM = 0;
  a.A1 = A1; a.C0 = A1 ? C0 : K; a.ldA0 = ldA0; a.ldA1 = ldA1; a.ldC = ldC; a.bias_n = bias_n; a.alpha = alpha; a.accumulate = accumulate;
  a.pz = 0; a.gx = 0;
  const dim3 grid((unsigned)(cdiv((int)M, WBM) * a.NB), 1, 1);
  const bool direct_store = cur_opt().wgemm_gen_epi == 0;      // A/B switch: 32-byte-piece stores
  if (direct_store) hipLaunchKernelGGL((wgemm_bf16x3_kernel<true, false>), grid, dim3(WNT), 0, st, a);
  else hipLaunchKernelGGL((wgemm_bf16x3_kernel<true, true>), grid, dim3(WNT), 0, st, a);
}

// out (two-destination view) = alpha * A (M x K) . W^T (N x K, pre-split) + GroupNorm backward apply of (x, da): see the GNB epilogue
bool wgemm_gnbwd_supported(int N, int K, int ldA, const Src2& x, const Dst2& d, const void* A, const void* da) {
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool split_ok = (x.p1 == nullptr || x.C0 % WBN == 0) && (d.p1 == nullptr || d.C0 % WBN == 0);
  return wgemm_supported(N, K) && ldA % 4 == 0 && split_ok && x.ld0 % 4 == 0 && x.ld1 % 4 == 0 && d.ld0 % 4 == 0 && d.ld1 % 4 == 0 && al16(A) && al16(da) &&
         al16(x.p0) && al16(x.p1) && al16(d.p0) && al16(d.p1);
}
void launch_wgemm_bf16x3_gnbwd(const float* A, int ldA, const void* W3, long long M, int N, int K, float alpha, Src2 x, const float* da, const float* stats,
                               const float* red, const float* gamma, const float* beta, int G, int silu, int HW, Dst2 d, hipStream_t st) {
  WgemmArgs a{};
  a.V = A; a.U3 = reinterpret_cast<const unsigned char*>(W3); a.M = nullptr;
  a.Mt = (int)M; a.Cin = K; a.Cout = N; a.S = K / WKS; a.NB = N / WBN; a.sV = 0; a.sM = 0;
  a.A1 = nullptr; a.C0 = K; a.ldA0 = ldA; a.ldA1 = 0; a.ldC = N; a.bias_n = nullptr; a.alpha = alpha; a.accumulate = 0;
  a.pz = 0; a.gx = 0;
  a.gxv = x; a.gd = d; a.gda = da; a.gstats = stats; a.gred = red; a.ggamma = gamma; a.gbeta = beta; a.gG = G; a.gsilu = silu; a.gHW = HW;
  const dim3 grid((unsigned)(cdiv((int)M, WBM) * a.NB), 1, 1);
  hipLaunchKernelGGL((wgemm_bf16x3_kernel<true, false, true>), grid, dim3(WNT), 0, st, a);
}

// 64-row waves where option gen_rows says so; 0 = by size: from 32768 rows on (below that the halved workgroup count leaves CUs idle: 16384 x 256 x 256
// runs 17 us in the 32-row form, 25 us in the 64-row form; 65536 x 256 x 256 50 / 42 us)
// two column blocks per workgroup where N allows (option gen_cp: 0 never, 2 always, 1 = from 32768 rows on: 262144 x 256 x 512 279 -> 245 us, x 384 226 ->
// 201 us, x 256 166 -> 158 us, 65536 x 256 x 512 71 -> 65 us in isolation; 16384 x 256 x 256 17 -> 23 us)
static bool gen_colpair(long long M, int NB) { const int c = cur_opt().gen_cp; return NB % 2 == 0 && (c == 2 || (c == 1 && M >= 32768)); }
static bool gen_rows64(long long M) { const int r = cur_opt().gen_rows; return r == 64 || (r == 0 && M >= 32768); }
// f16x2 forms of the two general launches: W2 = wgemm_f16x2_pack_weights(W, ., 1, N, K) (one power of two for the whole matrix)
bool wgemm_f16x2_general_supported(int N, int K, int C0, int ldA0, int ldA1, int ldC, const void* A0, const void* A1, const void* C, const void* bias) {
  return wgemm_f16x2_supported(N, K) && wgemm_general_supported(N, K, C0, ldA0, ldA1, ldC, A0, A1, C, bias);
}
void launch_wgemm_f16x2_general(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W2, float* C, int ldC, long long M, int N, int K,
                                const float* bias_n, float alpha, int accumulate, hipStream_t st) {
  WgemmArgs a{};
  a.V = A0; a.U3 = reinterpret_cast<const unsigned char*>(W2); a.M = C;
  a.Mt = (int)M; a.Cin = K; a.Cout = N; a.S = K / WKS; a.NB = N / WBN;
  a.A1 = A1; a.C0 = A1 ? C0 : K; a.ldA0 = ldA0; a.ldA1 = ldA1; a.ldC = ldC; a.bias_n = bias_n; a.alpha = alpha; a.accumulate = accumulate;
  a.uinv = reinterpret_cast<const float*>(a.U3 + (size_t)N * K * 4);
  if (gen_colpair(M, a.NB)) hipLaunchKernelGGL((wgemm_f16x2_gencp_kernel<false>), dim3((unsigned)(cdiv((int)M, WBM) * (a.NB / 2))), dim3(WNT), 0, st, a);
  else if (gen_rows64(M)) hipLaunchKernelGGL((wgemm_f16x2_gen2_kernel<false>), dim3((unsigned)(cdiv((int)M, 2 * WBM) * a.NB)), dim3(WNT), 0, st, a);
  else hipLaunchKernelGGL((wgemm_f16x2_gen_kernel<false>), dim3((unsigned)(cdiv((int)M, WBM) * a.NB)), dim3(WNT), 0, st, a);
}
void launch_wgemm_f16x2_gnbwd(const float* A, int ldA, const void* W2, long long M, int N, int K, float alpha, Src2 x, const float* da, const float* stats,
                              const float* red, const float* gamma, const float* beta, int G, int silu, int HW, Dst2 d, hipStream_t st) {
  WgemmArgs a{};
  a.V = A; a.U3 = reinterpret_cast<const unsigned char*>(W2); a.M = nullptr;
  a.Mt = (int)M; a.Cin = K; a.Cout = N; a.S = K / WKS; a.NB = N / WBN;
  a.A1 = nullptr; a.C0 = K; a.ldA0 = ldA; a.ldA1 = 0; a.ldC = N; a.alpha = alpha;
  a.uinv = reinterpret_cast<const float*>(a.U3 + (size_t)N * K * 4);
  a.gxv = x; a.gd = d; a.gda = da; a.gstats = stats; a.gred = red; a.ggamma = gamma; a.gbeta = beta; a.gG = G; a.gsilu = silu; a.gHW = HW;
  a.gnt = cur_opt().gnb_nt;
  if (gen_colpair(M, a.NB)) hipLaunchKernelGGL((wgemm_f16x2_gencp_kernel<true>), dim3((unsigned)(cdiv((int)M, WBM) * (a.NB / 2))), dim3(WNT), 0, st, a);
  else if (gen_rows64(M)) hipLaunchKernelGGL((wgemm_f16x2_gen2_kernel<true>), dim3((unsigned)(cdiv((int)M, 2 * WBM) * a.NB)), dim3(WNT), 0, st, a);
  else hipLaunchKernelGGL((wgemm_f16x2_gen_kernel<true>), dim3((unsigned)(cdiv((int)M, WBM) * a.NB)), dim3(WNT), 0, st, a);
}

void launch_wgemm_bf16x3(const float* V, const void* U3, float* M, long long Mt, int Cout, int Cin, int P, hipStream_t st) {
  WgemmArgs a{};
  a.A1 = nullptr; a.C0 = Cin; a.ldA0 = Cin; a.ldA1 = 0; a.ldC = Cout; a.bias_n = nullptr; a.alpha = 1.f; a.accumulate = 0;
  a.V = V; a.U3 = reinterpret_cast<const unsigned char*>(U3); a.M = M;
  a.Mt = (int)Mt; a.Cin = Cin; a.Cout = Cout; a.S = Cin / WKS; a.NB = Cout / WBN;
  a.sV = Mt * Cin; a.sM = Mt * Cout;
  const bool by_pos = cur_opt().wgemm_xcdpos != 0;     // A/B switch
  const int gx = cdiv((int)Mt, WBM) * a.NB;
  const bool fold = by_pos && P % 8 == 0 && (long long)gx * P < (1LL << 31);
  a.pz = fold ? P : 0; a.gx = gx;
  const dim3 grid(fold ? (unsigned)(gx * P) : (unsigned)gx, 1, fold ? 1u : (unsigned)P);
  const bool direct_store = cur_opt().wgemm_epi == 0;      // A/B switch: 32-byte-piece stores (+0.3 ... 1.9 % slower)
  if (direct_store) hipLaunchKernelGGL((wgemm_bf16x3_kernel<false, false>), grid, dim3(WNT), 0, st, a);
  else hipLaunchKernelGGL((wgemm_bf16x3_kernel<false, true>), grid, dim3(WNT), 0, st, a);
}

// f16x2 form
// image = P * Cout * Cin * 4 bytes of stage images + 256 bytes (the positions' inverse scales, floats) + 256 bytes (their abs-max bit patterns); P <= 64
size_t wgemm_f16x2_packed_bytes(int P, int Cout, int Cin) { return (size_t)P * Cout * Cin * 4 + 512; }
void wgemm_f16x2_pack_weights(const float* U_dev, void* U2_dev, int P, int Cout, int Cin, hipStream_t st) {
  unsigned* umax_scratch = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(U2_dev) + (size_t)P * Cout * Cin * 4 + 256);
  (void)hipMemsetAsync(umax_scratch, 0, 256, st);
  hipLaunchKernelGGL(wgemm_umax_kernel, dim3(16, (unsigned)P), dim3(256), 0, st, U_dev, umax_scratch, (long long)Cout * Cin);
  const long long n16 = (long long)P * Cout * Cin * 4 / 16;
  hipLaunchKernelGGL(wgemm_pack2_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, U_dev, reinterpret_cast<u32x4*>(U2_dev), umax_scratch, P, Cout, Cin);
}
void launch_abs_max_bits(const float* x, int groups, int segments, long long seg_len, unsigned* out, hipStream_t st) {
  (void)hipMemsetAsync(out, 0, (size_t)segments * VMAX_SUB * VMAX_STRIDE * 4, st);
  const unsigned chunks = (unsigned)std::min<long long>(256, (seg_len + 2047) / 2048);
  hipLaunchKernelGGL(abs_max_bits_kernel, dim3(chunks, (unsigned)segments), dim3(256), 0, st, x, groups, segments, seg_len, out);
}
// vmax: [utterance][VMAX_SUB][VMAX_STRIDE] partial abs-maxima (float bits) of V; an utterance = tiles_per_utt (>= 32) consecutive rows
void launch_wgemm_f16x2(const float* V, const void* U2, float* M, long long Mt, int Cout, int Cin, int P, const unsigned* vmax, int tiles_per_utt, hipStream_t st) {
  WgemmArgs a{};
  a.V = V; a.U3 = reinterpret_cast<const unsigned char*>(U2); a.M = M;
  a.Mt = (int)Mt; a.Cin = Cin; a.Cout = Cout; a.S = Cin / WKS; a.NB = Cout / WBN;
  a.sV = Mt * Cin; a.sM = Mt * Cout;
  a.vmax = vmax; a.tpu = tiles_per_utt; a.uinv = reinterpret_cast<const float*>(a.U3 + (size_t)P * Cout * Cin * 4);
  const bool by_pos = cur_opt().wgemm_xcdpos != 0;
  // 64-row waves (wgemm_f16x2_rt2_kernel): 4-6 % faster on every shape of the shipped network (tools/wgemm_modes_bench.py, profiles/README.md r05i); option
  // wgemm_rt 1 = the 32-row kernel (A/B switch)
  const bool rt2 = tiles_per_utt >= 64 && cur_opt().wgemm_epi != 0 && cur_opt().wgemm_rt != 1;
  const int gx = cdiv((int)Mt, rt2 ? 2 * WBM : WBM) * a.NB;
  const bool fold = by_pos && P % 8 == 0 && (long long)gx * P < (1LL << 31);
  a.pz = fold ? P : 0; a.gx = gx;
  const dim3 grid(fold ? (unsigned)(gx * P) : (unsigned)gx, 1, fold ? 1u : (unsigned)P);
  if (rt2) {
    switch (cur_opt().wgemm_nt) {
      case 1: hipLaunchKernelGGL(wgemm_f16x2_rt2_kernel<1>, grid, dim3(WNT), 0, st, a); break;
      case 2: hipLaunchKernelGGL(wgemm_f16x2_rt2_kernel<2>, grid, dim3(WNT), 0, st, a); break;
      case 3: hipLaunchKernelGGL(wgemm_f16x2_rt2_kernel<3>, grid, dim3(WNT), 0, st, a); break;
      default: hipLaunchKernelGGL(wgemm_f16x2_rt2_kernel<0>, grid, dim3(WNT), 0, st, a);
    }
  }
  else if (cur_opt().wgemm_epi == 0) hipLaunchKernelGGL((wgemm_f16x2_kernel<false>), grid, dim3(WNT), 0, st, a);
  else hipLaunchKernelGGL((wgemm_f16x2_kernel<true>), grid, dim3(WNT), 0, st, a);
}

}  // namespace buddy
