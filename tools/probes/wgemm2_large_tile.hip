// EXPERIMENT, not product (round 5; measured and rejected -- profiles/README.md "Dominant GEMM, round 5"): a from-scratch large-tile form of the batched
// bf16x3 GEMM (256-row workgroups, one wave per SIMD, both operands by inline-asm LDS-DMA, three-deep V ring, the next stage's split inside the MFMA blocks,
// peeled tail).  Bit-identical to csrc/wgemm.hip and within +-5 % of it on every shape (0.58 vs 0.54 ms at K = N = 128, 0.915 vs 0.94 at K = 512): two
// structurally unrelated kernels land on the same throughput, which (with the 49 %-busy pipe at 2.06 GHz against 100 % at 1.57 GHz for a register-only loop)
// says the shape is bounded by the power budget of its instruction mix, not by a schedule.  Kept for the record; to build it, copy it next to wgemm.hip, add it
// to build.sh, declare wgemm2_pays / launch_wgemm2_bf16x3 in common.h and call them from launch_wgemm_bf16x3 (tools/wgemm_v2_check.py compares the two).
// Winograd-domain batched GEMM  M[p] (tiles x Cout) = V[p] (tiles x Cin) . U[p]^T  in bf16x3 arithmetic, LARGE-TILE form (round 5) for the big layers:
// the same products in the same order as wgemm.hip's kernel (bit-identical results: every fp32 operand split exactly into three bf16 terms, six
// v_mfma_f32_32x32x16_bf16 products per 16 k, smallest terms first, fp32 accumulation over k ascending), restructured around what limits that kernel:
//   * workgroup = 4 waves x 64 rows (two 32-row tiles per wave) x 128 columns, ONE wave per SIMD with the whole register file: a weight fragment read from
//     LDS feeds TWO MFMAs and a staged weight stage serves 256 rows (half the LDS reads and half the L2 weight traffic per multiply-add);
//   * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4): the V rows as whole 128-byte lines (the register path read 16-byte pieces of 32 lines
//     per instruction), swizzled on the source address; no staging registers, no ds_write pass; stage s + 1 is in flight under the 96 MFMAs of stage s;
//   * a wave's V rows are private to it: only the weight stage needs the workgroup barrier (one per stage);
//   * the fragment reads and the three-way split of the NEXT k chunk are issued inside the MFMA block of the current one, the order pinned with
//     sched_barrier / sched_group_barrier (hipcc otherwise serialises read -> wait -> MFMA in a one-wave-per-SIMD loop: profiles/README.md round 5).
#include "common.h"
#include <cstdint>
#include <type_traits>

namespace buddy {
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM2 = 256, BN2 = 128, KS2 = 32;
constexpr int B_STAGE = BN2 * KS2 * 6;             // 24 KB: the weight stage image of wgemm.hip (2 k chunks x 4 column blocks x 3 planes x 1 KB)
constexpr int A_WAVE = 64 * KS2 * 4;               // 8 KB: one wave's 64 rows x 32 k fp32
constexpr int A_STAGE = 4 * A_WAVE;                // 32 KB
constexpr int NA = 3, NBUF = 2;                    // LDS ring depths: V rows three stages (a wave reads and splits stage s + 1 while it multiplies stage s),
                                                   // weights two (they need the workgroup barrier anyway)
constexpr int A_BASE = 0, B_BASE = NA * A_STAGE;   // 96 KB of V stages, then 48 KB of weight stages: 144 KB
constexpr int FRAG2 = 1024;

struct Split3 { bf16x8 p[3]; };
__device__ __forceinline__ Split3 split3(const float4 a, const float4 b) {     // identical to wgemm.hip
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
  for (int q = 0; q < 4; ++q) {
    ph[q] = __builtin_amdgcn_perm(h[2 * q + 1], h[2 * q], 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(m[2 * q + 1], m[2 * q], 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(l[2 * q + 1], l[2 * q], 0x07060302u);
  }
  Split3 s;
  s.p[0] = (bf16x8)ph; s.p[1] = (bf16x8)pm; s.p[2] = (bf16x8)pl;
  return s;
}

// LDS-DMA as inline asm (cdna_hip_programming.md section 5.7): with the builtin, hipcc's wait-count pass drains ALL pending LDS-DMA (s_waitcnt vmcnt(0)) in
// front of the first ds_read of the stage being computed and again at every __syncthreads() -- it cannot tell the buffer being filled from the one being
// read -- i.e. no load / MFMA overlap at all (first build of this kernel).  An asm DMA is invisible to that pass: its completion is waited for by the one
// asm `s_waitcnt vmcnt(0)` at the top of the stage that consumes it, followed by the barrier.  M0 (LDS base of the DMA) is saved and restored inside the
// statement; source = uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset.
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
struct AOff { unsigned o[8]; };

struct W2Args {
  const float* V; const unsigned char* U3; float* M;
  int Mt, Cin, Cout, S, NB;
  int pz, gx;                    // as wgemm.hip: pz > 0: positions folded into a 1-D grid, XCD x owns positions x mod 8
  long long sV, sM;
};

__global__ __launch_bounds__(256, 1) void wgemm2_bf16x3_kernel(const W2Args a) {
  __shared__ __attribute__((aligned(16))) char smem[NA * A_STAGE + NBUF * B_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hi = lane >> 5;
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
  const int nb = lid % a.NB, m0 = (lid / a.NB) * BM2;
  const float* __restrict__ V = a.V + (long long)p * a.sV;
  const unsigned char* __restrict__ U3 = a.U3 + ((long long)p * a.NB + nb) * a.S * B_STAGE;
  const int S = a.S;

  // A DMA: this wave's 64 rows, 8 instructions of 8 rows x 128 B; LDS piece q = 64 i + lane: row r = q >> 3, slot q & 7 holds source slot ^ ((r >> 1) & 7)
  AOff aoff;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = 64 * i + lane, r = q >> 3, sg = (q & 7) ^ ((r >> 1) & 7);
    int row = m0 + 64 * wid + r;
    if (row >= a.Mt) row = a.Mt - 1;                          // clamped rows are never stored
    aoff.o[i] = (unsigned)(((long long)row * a.Cin + sg * 4) * 4);
  }
  const unsigned boff = (unsigned)(tid * 16);                 // B DMA: thread t moves bytes 16 t + 4096 j of the linear stage image
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem);
  // DMA issue: the weight stage first, then the V stage -- `s_waitcnt vmcnt(8)` then means "everything but the 8 V DMAs issued last has landed"
  auto stage_b = [&](int s) {
    const void* vb = uniform_ptr(U3 + (long long)s * B_STAGE);
    const unsigned lb = lds0 + B_BASE + (s % NBUF) * B_STAGE + wid * 1024;
#pragma unroll
    for (int j = 0; j < 6; ++j) glds16_asm(vb, boff + j * 4096, lb + j * 4096);
  };
  auto stage_a = [&](int s) {
    const void* va = uniform_ptr(V + s * KS2);
    const unsigned la = lds0 + A_BASE + (s % NA) * A_STAGE + wid * A_WAVE;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16_asm(va, aoff.o[i], la + i * 1024);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  const int fa = (lq >> 1) & 7;
  // fragment reads: A two tiles x two 16-byte slots (8 floats) of k chunk kc of this wave's rows; B 12 fragments (4 column blocks x 3 planes)
  auto readA = [&](int s, int kc, float4 (&raw)[2][2]) {
    const char* base = smem + A_BASE + (s % NA) * A_STAGE + wid * A_WAVE;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) raw[t][j] = *reinterpret_cast<const float4*>(base + (32 * t + lq) * 128 + (((4 * hi + 2 * kc + j) ^ fa) << 4));
  };
  auto readB = [&](int s, int kc, bf16x8 (&b)[4][3]) {
    const char* base = smem + B_BASE + (s % NBUF) * B_STAGE + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) b[cb][q] = *reinterpret_cast<const bf16x8*>(base + ((kc * 4 + cb) * 3 + q) * FRAG2);
  };
  constexpr int PB[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};      // smallest terms first (B plane, A plane), as wgemm.hip
  auto mma = [&](const bf16x8 (&b)[4][3], const Split3 (&av)[2]) {
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cb][PB[t6]], av[t].p[PA[t6]], acc[t][cb], 0, 0, 0);
  };

  // prologue: weights 0, V 0, V 1 in flight; V 0 read and split
  stage_b(0);
  stage_a(0);
  if (S > 1) stage_a(1);
  if (S > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Split3 av0[2], av1[2];                                       // the split V fragments of the stage being multiplied (k chunks 0 and 1)
  {
    float4 r0[2][2], r1[2][2];
    readA(0, 0, r0); readA(0, 1, r1);
    av0[0] = split3(r0[0][0], r0[0][1]); av0[1] = split3(r0[1][0], r0[1][1]);
    av1[0] = split3(r1[0][0], r1[0][1]); av1[1] = split3(r1[1][0], r1[1][1]);
  }
  // one stage; H1 / H2 = "stage s + 1 / s + 2 exists" as COMPILE-TIME flags (the last two stages are peeled): a run-time branch inside the K loop splits
  // it into basic blocks and hipcc's schedule falls apart (wgemm.hip: 65 -> 81 ms/step from one never-taken `if`)
  auto body = [&](int s, auto h1, auto h2) {
    constexpr bool H1 = decltype(h1)::value, H2 = decltype(h2)::value;
    // weights of stage s: issued before the 8 V DMAs of the previous top -> "all but the last 8" covers them
    if constexpr (H1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // ... for every wave; every wave is done reading the weight buffer of stage s - 1
    if constexpr (H1) stage_b(s + 1);
    if constexpr (H2) stage_a(s + 2);                           // V ring slot (s + 2) % 3 held stage s - 1: this wave read it during stage s - 2
    bf16x8 b0[4][3], b1[4][3];
    readB(s, 0, b0);
    __builtin_amdgcn_sched_barrier(0);
    // V of stage s + 1 (this wave's own rows: its DMAs only need this wave's wait, no barrier): landed when at most the 14 / 6 DMAs of this top are out
    float4 r0[2][2], r1[2][2];
    Split3 nv0[2], nv1[2];
    if constexpr (H1) {
      if constexpr (H2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      readA(s + 1, 0, r0); readA(s + 1, 1, r1);
    }
    readB(s, 1, b1);
    __builtin_amdgcn_sched_barrier(0);
    // 48 MFMAs of chunk 0 with the split of the next stage's chunk 0 (88 VALU) inside the fenced region, then chunk 1 with the next chunk 1
    if constexpr (H1) { nv0[0] = split3(r0[0][0], r0[0][1]); nv0[1] = split3(r0[1][0], r0[1][1]); }
    mma(b0, av0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H1) { nv1[0] = split3(r1[0][0], r1[0][1]); nv1[1] = split3(r1[1][0], r1[1][1]); }
    mma(b1, av1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H1) { av0[0] = nv0[0]; av0[1] = nv0[1]; av1[0] = nv1[0]; av1[1] = nv1[1]; }
  };
  int s = 0;
  for (; s + 2 < S; ++s) body(s, std::true_type{}, std::true_type{});
  if (S >= 2) { body(s, std::true_type{}, std::false_type{}); ++s; }
  body(s, std::false_type{}, std::false_type{});

  // epilogue: each 32-row tile through this wave's LDS slab (the stage buffers are free: every wave has passed its last read), 256-byte row pieces out
  __syncthreads();
  constexpr int SP = 68;
  float* St = reinterpret_cast<float*>(smem) + wid * (32 * SP);
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rbase = m0 + 64 * wid + 32 * t;
    float* Mrow = a.M + (long long)p * a.sM + (long long)rbase * a.Cout + nb * BN2;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(St + lq * SP + cl * 32 + 8 * g + 4 * hi) =
              make_float4(acc[t][2 * hb + cl][4 * g], acc[t][2 * hb + cl][4 * g + 1], acc[t][2 * hb + cl][4 * g + 2], acc[t][2 * hb + cl][4 * g + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + rr;
        const float4 v = *reinterpret_cast<const float4*>(St + r * SP + c4);
        if (rbase + r < a.Mt) *reinterpret_cast<float4*>(Mrow + (long long)r * a.Cout + hb * 64 + c4) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}
}  // namespace

// large-tile form: worth it when a position has enough 256-row tiles to fill the chip at one workgroup per CU
bool wgemm2_pays(long long Mt, int Cout, int P) { return ((Mt + BM2 - 1) / BM2) * (Cout / BN2) * P >= 1024; }
void launch_wgemm2_bf16x3(const float* V, const void* U3, float* M, long long Mt, int Cout, int Cin, int P, hipStream_t st) {
  W2Args a{};
  a.V = V; a.U3 = reinterpret_cast<const unsigned char*>(U3); a.M = M;
  a.Mt = (int)Mt; a.Cin = Cin; a.Cout = Cout; a.S = Cin / KS2; a.NB = Cout / BN2;
  a.sV = Mt * Cin; a.sM = Mt * Cout;
  const int gx = cdiv((int)Mt, BM2) * a.NB;
  const bool fold = P % 8 == 0 && (long long)gx * P < (1LL << 31);
  a.pz = fold ? P : 0; a.gx = gx;
  const dim3 grid(fold ? (unsigned)(gx * P) : (unsigned)gx, 1, fold ? 1u : (unsigned)P);
  hipLaunchKernelGGL(wgemm2_bf16x3_kernel, grid, dim3(256), 0, st, a);
}

}  // namespace buddy
