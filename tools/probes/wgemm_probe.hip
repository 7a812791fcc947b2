// Timing probe (tool, not product): the bf16x3 Winograd-domain GEMM kernel of buddy_amd/csrc/wgemm.hip with pieces switched off
// (results are then wrong; only the time is of interest) to see where a launch's time goes.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I buddy_amd/csrc
#include "common.h"
#include <cstdint>
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
  long long sV, sM;                                            // strides between positions (floats)
  // general form (GEN = true; 1x1 convolutions, NIN): A from up to two sources (channel concatenation, split at C0), row strides, C = alpha * A W^T
  // + bias [+ C]
  const float* A1; int C0, ldA0, ldA1, ldC; const float* bias_n; float alpha; int accumulate;
};

// GEN: the general form (two-source A, row strides, alpha / bias / accumulate epilogue).  EPI: the accumulator tile goes through a wave-private LDS
// slab (the weight buffers are free after the last stage) and leaves as 256-byte row pieces instead of 32-byte pieces per lane pair.
// Measured and rejected (profiles/README.md r03a): a second stage of A in flight (190 VGPRs: -3 %), 64 rows per wave (256+ VGPRs: -25 %), two
// instead of three workgroups per CU (-2...4 %).
template <bool GEN, bool EPI, int PROBE>
__global__ __launch_bounds__(WNT, 3) void wgemm_bf16x3_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware order (hardware places block b on XCD b % 8): each XCD gets a contiguous range of logical tiles, the column blocks of one row
  // tile adjacent, so the second column block finds its V rows in the same L2
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nb = lid % a.NB, m0 = (lid / a.NB) * WBM;
  const int p = blockIdx.z;
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
  // PROBE & 64: the same bytes fetched with a coalesced lane pattern (instruction j = rows 8j..8j+7 of the wave, 8 lanes per 128-byte row piece);
  // the lanes then hold the wrong elements (timing only)
  int crow = m0 + wid * 32 + (lane >> 3); if (crow + 24 >= a.Mt) crow = a.Mt - 25;
  const float* Apc = V + (long long)crow * a.Cin + 4 * (lane & 7);
  auto loadA = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (PROBE & 64) ra[j] = *reinterpret_cast<const float4*>(Apc + (long long)(8 * j) * a.Cin + s * WKS);
      else if (PROBE & 256) { typedef float nf4 __attribute__((ext_vector_type(4))); const nf4 vv = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(Ap + s * WKS + 4 * j)); ra[j] = make_float4(vv[0], vv[1], vv[2], vv[3]); }
      else ra[j] = *reinterpret_cast<const float4*>(((GEN && Ap1 && s * WKS >= a.C0) ? Ap1 : Ap) + s * WKS + 4 * j);
    }
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
    if (s + 1 < S) {
      if (PROBE & 512) { loadB(s + 1); loadA(s + 1); }      // B first: the wait before the LDS store of B then leaves the A loads in flight
      else { if (!(PROBE & 1)) loadA(s + 1); if (!(PROBE & 2)) loadB(s + 1); }
    }
    const unsigned char* Bcur = smem + (s & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      Split3 av;
      if (PROBE & 8) { av.p[0] = (bf16x8)(*(const u32x4*)&ca[2 * kc]); av.p[1] = (bf16x8)(*(const u32x4*)&ca[2 * kc + 1]); av.p[2] = av.p[0]; }
      else av = split3(ca[2 * kc], ca[2 * kc + 1]);
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
        for (int cb = 0; cb < 4; ++cb) {
          if (PROBE & 16) { acc[cb][t] += ((const float*)&b[cb][PB[t]])[0] * ((const float*)&av.p[PA[t]])[1]; }
          else acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][PB[t]], av.p[PA[t]], acc[cb], 0, 0, 0);
        }
    }
    if (s + 1 < S && !(PROBE & 2)) storeB((s + 1) & 1);
    if (PROBE & 512) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    else if (!(PROBE & 4)) __syncthreads();
  }

  if (PROBE & 32) { if (acc[0][0] == 123.456f && acc[1][3] == 1.f && acc[2][5] == 2.f && acc[3][7] == 3.f) a.M[0] = 1.f; return; }
  // epilogue: accumulator = C^T tile, lane (row = lane & 31, h = lane >> 5) holds channels 8 g + 4 h + 0..3 of each 32-channel block
  if (EPI && !GEN) {
    constexpr int SP = 68;                                   // floats per staged row (64 columns + 4: conflict-free 16-byte writes down a column)
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
        if (m0 + wid * 32 + r < a.Mt) {
          if (PROBE & 128) { typedef float nf4 __attribute__((ext_vector_type(4))); nf4 vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<nf4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4)); }
          else *reinterpret_cast<float4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4) = v;
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

// ---- persistent variant: a workgroup walks row tiles g, g + G, ... of ONE (position, column block); the K-stage stream runs on across tile
// boundaries (the next tile's first A / B stage is in flight under the last MFMAs of the current one), the finished tile's accumulators leave
// by direct stores while the next tile computes.  WAVES x 32 rows per tile.  AD: A prefetch depth in stages.
template <int WAVES, int AD, int MINB, int BMODE>
__global__ __launch_bounds__(64 * WAVES, MINB) void wgemm_p_kernel(const WgemmArgs a, const int G) {
  constexpr int NT = 64 * WAVES, BM = 32 * WAVES, NPC = STAGE_BYTES / 16 / NT;      // 16-byte pieces of a B stage per thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int nb = lid % a.NB, qq = lid / a.NB, g = qq % G, p = qq / G;
  const int ntile = (a.Mt + BM - 1) / BM;
  if (g >= ntile) return;
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const unsigned char* __restrict__ U3 = a.U3 + ((long long)p * a.NB + nb) * a.S * STAGE_BYTES;
  const int S = a.S;
  const int total = ((ntile - g + G - 1) / G) * S;

  auto aptr = [&](int t) {
    int row = t * BM + wid * 32 + (lane & 31);
    if (row >= a.Mt) row = a.Mt - 1;
    return V + (long long)row * a.Cin + 16 * (lane >> 5);
  };
  const u32x4* Bg = reinterpret_cast<const u32x4*>(U3) + tid;
  u32x4* Bs = reinterpret_cast<u32x4*>(smem) + tid;

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float4 ra[AD][4];
  u32x4 rb[NPC];
  // stream position of the A prefetch (AD stages ahead) 
  int pt = g, ps = 0;
  const float* pA = aptr(pt);
  auto loadA = [&](float4* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const float4*>(pA + ps * WKS + 4 * j);
    if (++ps == S) { ps = 0; pt += G; pA = aptr(pt < ntile ? pt : g); }
  };
  auto loadB = [&](int s, int buf) {
    if (BMODE == 1) {                  // LDS-DMA: global -> LDS without passing through registers (destination = wave-uniform base + lane * 16)
#pragma unroll
      for (int j = 0; j < NPC; ++j)
        __builtin_amdgcn_global_load_lds(Bg + (long long)s * (STAGE_BYTES / 16) + j * NT,
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<u32x4*>(smem) + buf * (STAGE_BYTES / 16) + j * NT + wid * 64), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NPC; ++j) rb[j] = Bg[(long long)s * (STAGE_BYTES / 16) + j * NT];
    }
  };
  auto storeB = [&](int buf) {
    if (BMODE == 1) return;
#pragma unroll
    for (int j = 0; j < NPC; ++j) Bs[buf * (STAGE_BYTES / 16) + j * NT] = rb[j];
  };
#pragma unroll
  for (int d = 0; d < AD; ++d) loadA(ra[d]);
  loadB(0, 0);
  storeB(0);
  __syncthreads();
  int tile = g, s = 0;
  for (int gs = 0; gs < total; ++gs) {
    const float4 ca[4] = {ra[0][0], ra[0][1], ra[0][2], ra[0][3]};
#pragma unroll
    for (int d = 0; d + 1 < AD; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[d][j] = ra[d + 1][j];
    const int sn = (s + 1 == S) ? 0 : s + 1;
    if (BMODE == 2) { if (gs + 1 < total) loadB(sn, (gs + 1) & 1); if (gs + AD < total) loadA(ra[AD - 1]); }
    else { if (gs + AD < total) loadA(ra[AD - 1]); if (gs + 1 < total) loadB(sn, (gs + 1) & 1); }
    const unsigned char* Bcur = smem + (gs & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const Split3 av = split3(ca[2 * kc], ca[2 * kc + 1]);
      bf16x8 b[4][3];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[cb][q] = *reinterpret_cast<const bf16x8*>(Bcur + ((kc * 4 + cb) * 3 + q) * FRAG);
      constexpr int PB[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][PB[t]], av.p[PA[t]], acc[cb], 0, 0, 0);
    }
    if (gs + 1 < total) storeB((gs + 1) & 1);
    if (BMODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    else __syncthreads();
    if (s + 1 == S) {           // tile finished: its accumulators leave while the next tile's first stage is already in flight
      const int row = tile * BM + wid * 32 + (lane & 31);
      if (row < a.Mt) {
        float* dst = a.M + (long long)p * a.sM + (long long)row * a.Cout + nb * WBN + 4 * (lane >> 5);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int gg = 0; gg < 4; ++gg)
            *reinterpret_cast<float4*>(dst + cb * 32 + 8 * gg) = make_float4(acc[cb][4 * gg], acc[cb][4 * gg + 1], acc[cb][4 * gg + 2], acc[cb][4 * gg + 3]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      tile += G;
    }
    s = sn;
  }
}

// ---- N = 256 per workgroup: a wave owns 32 rows x 256 columns (8 accumulator tiles), so a V row is loaded and split ONCE for both column
// blocks (the 128-column kernel does it once per column block).  K advances in 16-k sub-stages (one MFMA k-chunk; 24 KB of B per sub-stage,
// double-buffered, by LDS-DMA); 48 MFMAs per wave per barrier as in the 128-column kernel.  Cout % 256 == 0.
__global__ __launch_bounds__(256, 2) void wgemm_n256_kernel(const WgemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int NB2 = a.NB / 2;
  const int nb2 = lid % NB2, m0 = (lid / NB2) * WBM;
  const int p = blockIdx.z;
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const int S = a.S, U = 2 * S;                                 // 16-k sub-stages
  int row = m0 + wid * 32 + (lane & 31);
  const bool row_ok = row < a.Mt;
  if (!row_ok) row = a.Mt - 1;
  const float* Ap = V + (long long)row * a.Cin + 16 * (lane >> 5);
  // B sub-stage u = (s, kc): 12 KB from each of the two column blocks' stage images
  const unsigned char* U3a = a.U3 + ((long long)p * a.NB + 2 * nb2) * S * STAGE_BYTES;
  const unsigned char* U3b = U3a + (long long)S * STAGE_BYTES;
  auto copyB = [&](int u, int buf) {
    const int s = u >> 1, kc = u & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const u32x4* src = reinterpret_cast<const u32x4*>((half ? U3b : U3a) + (long long)s * STAGE_BYTES + kc * (STAGE_BYTES / 2));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int piece = (j * 4 + wid) * 64;                  // wave-uniform first piece of this instruction
        __builtin_amdgcn_global_load_lds(src + piece + lane,
            (__attribute__((address_space(3))) void*)(reinterpret_cast<u32x4*>(smem) + buf * (STAGE_BYTES / 16) + half * (STAGE_BYTES / 32) + piece), 16, 0, 0);
      }
    }
  };
  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float4 ra[4];
  auto loadA = [&](int s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4*>(Ap + s * WKS + 4 * j);
  };
  loadA(0);
  copyB(0, 0);
  __syncthreads();
  float4 ca[4];
  for (int u = 0; u < U; ++u) {
    const int kc = u & 1;
    if (kc == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) ca[j] = ra[j];
      if (u + 2 < U) loadA((u >> 1) + 1);
    }
    if (u + 1 < U) copyB(u + 1, (u + 1) & 1);
    const unsigned char* Bcur = smem + (u & 1) * STAGE_BYTES + lane * 16;
    const Split3 av = split3(ca[2 * kc], ca[2 * kc + 1]);
    constexpr int PB[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 b[4][3];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[cb][q] = *reinterpret_cast<const bf16x8*>(Bcur + half * (STAGE_BYTES / 2) + (cb * 3 + q) * FRAG);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[4 * half + cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][PB[t]], av.p[PA[t]], acc[4 * half + cb], 0, 0, 0);
    }
    __syncthreads();
  }
  if (!row_ok) return;
  float* dst = a.M + (long long)p * a.sM + (long long)row * a.Cout + nb2 * 256 + 4 * (lane >> 5);
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + cb * 32 + 8 * g) = make_float4(acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]);
}

}  // namespace
}  // namespace buddy
#include <vector>
#include <time.h>
#include <cstdio>
#include <cstring>
using namespace buddy;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static int g_gap_us = 300;
template <typename F> static float gapped(F launch, int reps) {     // each launch alone, after an idle gap: the regime of the real step (GEMMs between memory-bound kernels)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0;
  for (int i = 0; i < reps + 2; ++i) {
    hipDeviceSynchronize();
    if (g_gap_us) { timespec ts{0, g_gap_us * 1000}; nanosleep(&ts, nullptr); }
    hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2) tot += ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return tot / reps;
}
template <int PROBE> static float run(const WgemmArgs& a, dim3 grid, int reps) {
  return gapped([&] { hipLaunchKernelGGL((wgemm_bf16x3_kernel<false, true, PROBE>), grid, dim3(WNT), 0, 0, a); }, reps);
}
template <int WAVES, int AD, int MINB, int BMODE> static float run_p(const WgemmArgs& a, int P, int wg_target, int reps, int* Gout) {
  const int BM = 32 * WAVES, ntile = (a.Mt + BM - 1) / BM;
  int G = wg_target / (P * a.NB); if (G < 1) G = 1; if (G > ntile) G = ntile;
  *Gout = G;
  const dim3 grid((unsigned)(P * a.NB * G));
  return gapped([&] { hipLaunchKernelGGL((wgemm_p_kernel<WAVES, AD, MINB, BMODE>), grid, dim3(64 * WAVES), 0, 0, a, G); }, reps);
}
static bool same(const float* dM, const std::vector<float>& ref) {
  std::vector<float> h(ref.size());
  hipMemcpy(h.data(), dM, ref.size() * 4, hipMemcpyDeviceToHost);
  return memcmp(h.data(), ref.data(), ref.size() * 4) == 0;
}
// one launch at a time with an idle gap before it (host sleep): does the kernel run faster when the chip had time to cool / bank power?
template <int PROBE> static void run_gapped(const WgemmArgs& a, dim3 grid, int gap_us, const char* name, double fl) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0, mn = 1e9f, mx = 0;
  const int R = 12;
  for (int i = 0; i < R + 2; ++i) {
    hipDeviceSynchronize();
    if (gap_us) { timespec ts{0, gap_us * 1000}; nanosleep(&ts, nullptr); }
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wgemm_bf16x3_kernel<false, true, PROBE>), grid, dim3(WNT), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2) { tot += ms; mn = ms < mn ? ms : mn; mx = ms > mx ? ms : mx; }
  }
  printf("  %-30s gap %6d us: avg %8.1f us (min %8.1f max %8.1f)  %7.0f bf16-TF\n", name, gap_us, tot / R * 1e3, mn * 1e3, mx * 1e3, 6 * fl / (tot / R * 1e-3) / 1e12);
}
int main() {
  const int shapes[][4] = {{29584, 256, 256, 64}, {29584, 128, 128, 64}, {29584, 128, 384, 64}, {7568, 256, 512, 64}};
  for (auto& sh : shapes) {
    const int Mt = sh[0], N = sh[1], K = sh[2], P = sh[3];
    float *V, *U, *M; void* U3;
    CK(hipMalloc(&V, (size_t)P * Mt * K * 4)); CK(hipMalloc(&U, (size_t)P * N * K * 4)); CK(hipMalloc(&M, (size_t)P * Mt * N * 4)); CK(hipMalloc(&U3, (size_t)P * N * K * 6));
    std::vector<float> h((size_t)P * Mt * K);
    unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    CK(hipMemcpy(V, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(U, h.data(), (size_t)P * N * K * 4, hipMemcpyHostToDevice));
    const long long n16 = (long long)P * N * K * 6 / 16;
    hipLaunchKernelGGL(wgemm_pack_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, U, reinterpret_cast<u32x4*>(U3), P, N, K);
    WgemmArgs a; memset(&a, 0, sizeof(a));
    a.C0 = K; a.ldA0 = K; a.ldC = N; a.alpha = 1.f; a.V = V; a.U3 = (const unsigned char*)U3; a.M = M; a.Mt = Mt; a.Cin = K; a.Cout = N; a.S = K / WKS; a.NB = N / WBN;
    a.sV = (long long)Mt * K; a.sM = (long long)Mt * N;
    const dim3 grid((unsigned)(((Mt + WBM - 1) / WBM) * a.NB), 1, (unsigned)P);
    const double fl = 2.0 * P * Mt * (double)N * K;
    const int R = 10;
    struct { const char* name; float ms; } r[] = {
      {"full", run<0>(a, grid, R)}, {"B loads first + raw barrier (no vmcnt drain)", run<512>(a, grid, R)}, {"full again", run<0>(a, grid, R)}, {"B first + raw barrier again", run<512>(a, grid, R)}, {"coalesced A pattern", run<64>(a, grid, R)}, {"coalesced A, no B copy", run<66>(a, grid, R)}, {"coalesced A, no epilogue", run<96>(a, grid, R)}, {"no A loads", run<1>(a, grid, R)}, {"no B copy", run<2>(a, grid, R)}, {"no B copy, no barrier", run<6>(a, grid, R)},
      {"no A, no B, no barrier", run<7>(a, grid, R)}, {"no split", run<8>(a, grid, R)}, {"no A/B/barrier/split", run<15>(a, grid, R)},
      {"no MFMA", run<16>(a, grid, R)}, {"no epilogue", run<32>(a, grid, R)}, {"no A/B/bar/split/epi (MFMA+LDS reads)", run<47>(a, grid, R)},
      {"no MFMA no epi", run<48>(a, grid, R)} };
    printf("Mt=%d N=%d K=%d P=%d\n", Mt, N, K, P);
    {
      run<0>(a, grid, 1);
      std::vector<float> ref((size_t)P * Mt * N);
      CK(hipMemcpy(ref.data(), M, ref.size() * 4, hipMemcpyDeviceToHost));
      { CK(hipMemset(M, 0xff, ref.size() * 4)); hipLaunchKernelGGL((wgemm_bf16x3_kernel<false, true, 512>), grid, dim3(WNT), 0, 0, a); hipDeviceSynchronize();
        printf("  B-first + raw barrier variant: %s\n", same(M, ref) ? "bit-exact" : "MISMATCH"); }
      int G = 0;
      struct { const char* name; float ms; bool ok; int G; } pr[12]; int np = 0;
#define PV(W, AD, MB, BM_, TGT, NAME) { CK(hipMemset(M, 0xff, ref.size() * 4)); float t_ = run_p<W, AD, MB, BM_>(a, P, TGT, R, &G); pr[np++] = {NAME, t_, same(M, ref), G}; }
      PV(4, 1, 3, 0, 1 << 30, "4 waves, A depth 1, syncthreads")
      PV(4, 1, 3, 2, 1 << 30, "4 waves, A depth 1, B first + raw barrier")
      PV(4, 2, 2, 2, 1 << 30, "4 waves, A depth 2, B first + raw barrier (2 wg/CU)")
      PV(4, 3, 2, 2, 1 << 30, "4 waves, A depth 3, B first + raw barrier (2 wg/CU)")
      PV(4, 1, 3, 2, 768, "4 waves, A depth 1, raw barrier, persistent")
      PV(4, 2, 2, 2, 512, "4 waves, A depth 2, raw barrier, persistent")
      PV(8, 2, 1, 2, 1 << 30, "8 waves, A depth 2, raw barrier")
      PV(8, 3, 1, 2, 256, "8 waves, A depth 3, raw barrier, persistent")
      for (int i = 0; i < np; ++i) printf("  %-44s %8.1f us  %7.1f TF-eq  %7.0f bf16-TF  G=%d %s\n", pr[i].name, pr[i].ms * 1e3, fl / (pr[i].ms * 1e-3) / 1e12, 6 * fl / (pr[i].ms * 1e-3) / 1e12, pr[i].G, pr[i].ok ? "bit-exact" : "MISMATCH");
    }
    for (auto& x : r) printf("  %-44s %8.1f us  %7.1f TF-eq  %7.0f bf16-TF\n", x.name, x.ms * 1e3, fl / (x.ms * 1e-3) / 1e12, 6 * fl / (x.ms * 1e-3) / 1e12);
    hipFree(V); hipFree(U); hipFree(M); hipFree(U3);
  }
  return 0;
}
