// Implicit-GEMM convolution / GEMM on the CDNA4 matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves every dense contraction of the score network (reference ddpm_conv3x3 layers.py:119-126,
// ddpm_conv1x1 layers.py:100-106, NIN layers.py:548-557, AttnBlockpp einsums layerspp.py:82-86) and the
// DFT-as-GEMM STFT/iSTFT (reference ncsnpp.py:473-496; n_fft = 510 is not a power of two).
//
// Tiling: 256 threads = 4 wave64 in a 2x2 grid, block tile 128(M) x 128(N) x 32(K); each wave owns 64x64 =
// 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  A/B tiles are staged through LDS as [row][k] with a 36-float
// row stride: the ds_read_b128 fragment reads (16-lane groups, rows distinct mod 16) are bank-conflict free and
// the 8-lane ds_write_b128 groups cover one 128-B row each.  The MFMA consumes one f32 per lane for A and B
// (lane l: row l&31, k-slot l>>5); a float4 fragment read per lane therefore feeds FOUR MFMAs (the two
// lane-halves take k = g*8+j and g*8+4+j), so a 32-deep K step costs 16 ds_read_b128 per wave for 64 MFMAs
// (4096 MFMA cycles): the loop is matrix-pipe bound, global loads for the next K step are issued before the
// first quarter of the MFMA block and land under the rest.  K order is channel-chunk outer / tap inner and the tile order is
// XCD-aware, so the nine shifted reads of a 3x3 conv hit the same L2 (HBM FETCH 9.7 GB -> 1.1 GB per 1.07 GB input).
#include "common.h"
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

namespace buddy {

namespace {
constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36, NT = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// MT = 32-row MFMA tiles per wave along M: 2 -> 128-row block tile (default), 1 -> 64-row block tile for GEMMs whose 128-row
// tiling would leave CUs idle (the operator's 4040 x 512 x 1028 DFT GEMMs are 128 tiles on 256 CUs).
// TAG only names an instantiation (the 36-batch GEMM of the F(4x4,3x3) convolutions shows up as its own row in rocprofv3 summaries).
template <int TAPS, bool TA, bool TB, int V = 2, int MT = 2, int TAG = 0>
__global__ __launch_bounds__(NT, 2) void igemm_kernel(const IgemmParams p) {
  static_assert(MT == 2 || !TA, "64-row tiles are only built for row-major A");
  constexpr int NBUF = (V >= 4) ? 2 : 1;
  constexpr int BMt = 64 * MT, AR = BMt / 32;
  constexpr int SM_MAIN = NBUF * (BMt + BN) * LDS_LD, SM_EPI = 64 * (BN + 4);
  __shared__ __attribute__((aligned(16))) float smem[SM_MAIN > SM_EPI ? SM_MAIN : SM_EPI];
  float* As = smem;
  float* Bs = smem + BMt * LDS_LD;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // XCD-aware tile order: hardware places block b on XCD b % 8 (each XCD has its own 4 MiB L2).  Give every XCD a
  // contiguous range of logical tiles, N-tiles of one M-tile adjacent, so the 3x3 halo rows and the A tile shared by the
  // N-tiles are re-read from the same L2 instead of 8 different ones (bijective remap, any grid size).
  const int nNt = (p.N + BN - 1) / BN;
  int lid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, k = orig >> 3;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (lid / nNt) * BMt, n0 = (lid % nNt) * BN;
  const long long bz = blockIdx.z;
  const float* __restrict__ A0 = p.A0 + bz * p.sA;
  const float* __restrict__ A1 = p.A1 ? p.A1 + bz * p.sA : nullptr;
  const float* __restrict__ Bt = p.Bt + bz * p.sB;
  float* __restrict__ C = p.C + bz * p.sC;
  const int M = p.M, N = p.N, Cin = p.Cin, H = p.H, W = p.W;
  const int chunks = (Cin + BK - 1) / BK;
  const int nk = TAPS * chunks;

  // loader coordinates
  const int lr = tid >> 3, lc4 = tid & 7;     // row-major operands: rows lr + 32 i, float4 column lc4
  const int tk = tid >> 5, tm4 = tid & 31;    // k-major operands:   k rows tk + 8 i, float4 (4 rows of the tile) tm4
  int ah[AR], aw[AR];
  bool am_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lr + 32 * i;
    am_ok[i] = m < M;
    if (TAPS == 9) { aw[i] = m % W; ah[i] = (m / W) % H; } else { aw[i] = 0; ah[i] = 0; }
  }

  auto loadA = [&](int kt, float4 (&r)[4]) {
    if (!TA) {
      // K order: channel chunk outer, tap inner -- the 9 shifted reads of one 32-channel slab are back to back (L2 reuse)
      const int tap = (TAPS == 9) ? kt % 9 : 0;
      const int c0 = ((TAPS == 9) ? kt / 9 : kt) * BK;
      const int dy = (TAPS == 9) ? tap / 3 - 1 : 0, dx = (TAPS == 9) ? tap % 3 - 1 : 0;
      int cc = c0 + lc4 * 4;
      const bool k_ok = cc < Cin;
      const float* src = A0; int ld = p.ldA0;
      if (A1 != nullptr && cc >= p.C0) { src = A1; ld = p.ldA1; cc -= p.C0; }
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        bool v = am_ok[i] && k_ok;
        if (TAPS == 9) v = v && (unsigned)(ah[i] + dy) < (unsigned)H && (unsigned)(aw[i] + dx) < (unsigned)W;
        const long long pix = (long long)(m0 + lr + 32 * i) + dy * W + dx;
        r[i] = v ? ld4(src + pix * ld + cc) : zero4();
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kt * BK + tk + 8 * i, m = m0 + tm4 * 4;
        r[i] = (k < Cin && m < M) ? ld4(A0 + (long long)k * p.ldA0 + m) : zero4();
      }
    }
  };
  auto loadB = [&](int kt, float4 (&r)[4]) {
    if (!TB) {
      const int tap = (TAPS == 9) ? kt % 9 : 0;
      const int c0 = ((TAPS == 9) ? kt / 9 : kt) * BK;
      const int cc = c0 + lc4 * 4;
      const bool k_ok = cc < Cin;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + lr + 32 * i;
        r[i] = (n < N && k_ok) ? ld4(Bt + (long long)n * p.ldB + tap * Cin + cc) : zero4();
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kt * BK + tk + 8 * i, n = n0 + tm4 * 4;
        r[i] = (k < Cin && n < N) ? ld4(Bt + (long long)k * p.ldB + n) : zero4();
      }
    }
  };
  auto store_tile = [&](float* S, const float4 (&r)[4], bool trans, int cnt = 4) {
    if (!trans) {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (i < cnt) *reinterpret_cast<float4*>(S + (lr + 32 * i) * LDS_LD + lc4 * 4) = r[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = tk + 8 * i;
        S[(tm4 * 4 + 0) * LDS_LD + k] = r[i].x;
        S[(tm4 * 4 + 1) * LDS_LD + k] = r[i].y;
        S[(tm4 * 4 + 2) * LDS_LD + k] = r[i].z;
        S[(tm4 * 4 + 3) * LDS_LD + k] = r[i].w;
      }
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[4], rb[4];
  loadA(0, ra);
  loadB(0, rb);
  const float* Af = As + (wm * 32 * MT + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
  const float* Bf = Bs + (wn * 64 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;

  auto mfma_group = [&](const float* Afp, const float* Bfp, int g) {
    const float4 a0 = *reinterpret_cast<const float4*>(Afp + g * 8);
    const float4 a1 = (MT == 2) ? *reinterpret_cast<const float4*>(Afp + 32 * LDS_LD + g * 8) : zero4();
    const float4 b0 = *reinterpret_cast<const float4*>(Bfp + g * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(Bfp + 32 * LDS_LD + g * 8);
    const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
    const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // operands swapped (weights first): the accumulator tile is C^T, i.e. a lane holds pixel m = lane&31 and FOUR CONSECUTIVE
      // output channels per register quad -- the epilogue moves 16-byte vectors.  Products and k order are unchanged (bit-identical).
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0[j], av0[j], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1[j], av0[j], acc[0][1], 0, 0, 0);
      if (MT == 2) {
        acc[MT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0[j], av1[j], acc[MT - 1][0], 0, 0, 0);
        acc[MT - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1[j], av1[j], acc[MT - 1][1], 0, 0, 0);
      }
    }
  };

  if (V < 4) {
    for (int kt = 0; kt < nk; ++kt) {
      store_tile(As, ra, TA, AR);
      store_tile(Bs, rb, TB);
      __syncthreads();
      if (V == 0 && kt + 1 < nk) { loadA(kt + 1, ra); loadB(kt + 1, rb); }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (V == 2 && g == 1 && kt + 1 < nk) { loadA(kt + 1, ra); loadB(kt + 1, rb); }
        if (V == 3 && g == 1 && kt + 1 < nk) loadA(kt + 1, ra);
        if (V == 3 && g == 2 && kt + 1 < nk) loadB(kt + 1, rb);
        mfma_group(Af, Bf, g);
      }
      __syncthreads();
    }
  } else {
    // double-buffered LDS: one barrier per K step
    constexpr int BUFSZ = (BMt + BN) * LDS_LD;
    store_tile(As, ra, TA, AR);
    store_tile(Bs, rb, TB);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = (kt & 1) * BUFSZ, nxt = BUFSZ - cur;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g == 1 && kt + 1 < nk) loadA(kt + 1, ra);
        if (g == 2 && kt + 1 < nk) loadB(kt + 1, rb);
        mfma_group(Af + cur, Bf + cur, g);
      }
      if (kt + 1 < nk) {
        store_tile(As + nxt, ra, TA, AR);
        store_tile(Bs + nxt, rb, TB);
      }
      __syncthreads();
    }
  }

  // epilogue.  Accumulator layout (transposed tile): m = lane&31, n = (r&3) + 8*(r>>2) + 4*(lane>>5) within a 32x32 tile.
  const bool need_pix = (p.res_mode == 2) || (p.bias_bn != nullptr);
  if (p.wide_epi) {
    // 16-byte-aligned rows, N a multiple of 4: stage the tile through LDS 64 rows at a time (the A/B buffers are free now) and
    // write whole 512-byte output rows with float4 stores; bias / residual reads are float4 too.
    constexpr int SLD = BN + 4;
    float* S = smem;
#pragma unroll
    for (int h = 0; h < MT; ++h) {
      if (h) __syncthreads();
      if (MT == 1 || wm == h) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(S + ((MT == 2 ? mt * 32 : wm * 32) + (lane & 31)) * SLD + wn * 64 + nt * 32 + 8 * g + 4 * (lane >> 5)) =
                  make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
      }
      __syncthreads();
      const int c4 = (tid & 31) * 4, n = n0 + c4;
      const bool n_ok = n < N;
      float4 bn4 = zero4();
      if (p.bias_n && n_ok) bn4 = ld4(p.bias_n + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (tid >> 5) + 8 * i;
        const int m = m0 + h * 64 + row;
        if (m >= M || !n_ok) continue;
        float4 v = *reinterpret_cast<const float4*>(S + row * SLD + c4);
        int bidx = 0; long long res_pix = m;
        if (need_pix) {
          bidx = m / p.rows_per_batch;
          if (p.res_mode == 2) {
            const int w = m % W, hh = (m / W) % H;
            res_pix = ((long long)bidx * (H >> 1) + (hh >> 1)) * (W >> 1) + (w >> 1);
          }
        }
        // same operation order per element as the scalar path: alpha*acc + bias_m, + bias_n, + bias_bn, + res, * out_scale, + C
        v.x = p.alpha * v.x; v.y = p.alpha * v.y; v.z = p.alpha * v.z; v.w = p.alpha * v.w;
        if (p.bias_m) { const float bm = p.bias_m[m]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
        if (p.bias_n) { v.x += bn4.x; v.y += bn4.y; v.z += bn4.z; v.w += bn4.w; }
        if (p.bias_bn) { const float4 t = ld4(p.bias_bn + (long long)bidx * p.ld_bias_bn + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (p.res_mode) { const float4 t = ld4(p.res + res_pix * p.ldRes + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        v.x *= p.out_scale; v.y *= p.out_scale; v.z *= p.out_scale; v.w *= p.out_scale;
        float* dst = C + (long long)m * p.ldC + n;
        if (p.accumulate) { const float4 t = ld4(dst); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        *reinterpret_cast<float4*>(dst) = v;
      }
    }
    return;
  }
  // generic path (ragged N, unaligned rows): scalar stores
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * 32 * MT + mt * 32 + (lane & 31);
    if (m >= M) continue;
    int bidx = 0; long long res_pix = m;
    if (need_pix) {
      bidx = m / p.rows_per_batch;
      if (p.res_mode == 2) {
        const int w = m % W, h = (m / W) % H;
        res_pix = ((long long)bidx * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1);
      }
    }
    const float bm = p.bias_m ? p.bias_m[m] : 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n >= N) continue;
        float v = p.alpha * acc[mt][nt][r] + bm;
        if (p.bias_n) v += p.bias_n[n];
        if (p.bias_bn) v += p.bias_bn[(long long)bidx * p.ld_bias_bn + n];
        if (p.res_mode) v += p.res[res_pix * p.ldRes + n];
        v *= p.out_scale;
        float* dst = C + (long long)m * p.ldC + n;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}
}  // namespace

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) ----
namespace {
struct ProfRec { hipEvent_t e0, e1; double flops, bytes, exec_flops; int taps; int M, N, K, batch, kind; };
// level 0: off.  1: only the dominant kernel (the 36 batched GEMMs of the three-pass convolutions, launch_wino4) -- every event pair costs a
// dispatch bubble of several microseconds, so the timed region of bench.py brackets nothing else.  2: every class below (attribution pass).
int g_prof_level = 0;
bool g_prof_on = false;          // level 2
std::vector<ProfRec> g_prof;
}
void igemm_prof_enable(int level) { g_prof_level = level < 0 ? 0 : level; g_prof_on = g_prof_level >= 2; }
bool igemm_prof_enabled() { return g_prof_on; }
int igemm_prof_level() { return g_prof_level; }
// sums elapsed time / algorithmic flops / launches per class (class 0: 3x3 convs, class 1: everything else) and clears
int igemm_prof_collect(double ms[2], double flops[2], long long launches[2], double bytes[2], double exec_flops[2]) {
  for (int c = 0; c < 2; ++c) { ms[c] = 0; flops[c] = 0; launches[c] = 0; bytes[c] = 0; exec_flops[c] = 0; }
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.e1) != hipSuccess) return 1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return 1;
    const int c = r.taps == 9 ? 0 : 1;
    static FILE* dump = prof_dump_path() ? fopen(prof_dump_path(), "w") : nullptr;   // per-launch shapes for tools/gemm_shapes.py
    if (dump) setvbuf(dump, nullptr, _IOLBF, 0);
    if (dump) fprintf(dump, "%d %d %d %d %d %d %.6f\n", r.kind, r.taps, r.M, r.N, r.K, r.batch, t);
    ms[c] += t; flops[c] += r.flops; launches[c] += 1; bytes[c] += r.bytes; exec_flops[c] += r.exec_flops;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return 0;
}

// generic bracket for HBM-bound launches (class 2: GroupNorm statistics / apply / backward): algorithmic bytes only
namespace { std::vector<ProfRec> g_prof_hbm; ProfRec g_cur_hbm; }
void prof_hbm_begin(double bytes, hipStream_t st) {
  if (!g_prof_on) return;
  (void)hipEventCreate(&g_cur_hbm.e0); (void)hipEventCreate(&g_cur_hbm.e1);
  g_cur_hbm.bytes = bytes; g_cur_hbm.flops = 0; g_cur_hbm.exec_flops = 0; g_cur_hbm.taps = 0;
  (void)hipEventRecord(g_cur_hbm.e0, st);
}
void prof_hbm_end(hipStream_t st) {
  if (!g_prof_on) return;
  (void)hipEventRecord(g_cur_hbm.e1, st);
  g_prof_hbm.push_back(g_cur_hbm);
}
int prof_hbm_collect(double* ms, double* bytes, long long* launches) {
  *ms = 0; *bytes = 0; *launches = 0;
  for (auto& r : g_prof_hbm) {
    if (hipEventSynchronize(r.e1) != hipSuccess) return 1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return 1;
    *ms += t; *bytes += r.bytes; *launches += 1;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_prof_hbm.clear();
  return 0;
}

// per-pass timing of the three-pass F(4x4,3x3) convolutions (bench.py roofline): [input transform, batched GEMM, output transform]
namespace { struct W4Rec { hipEvent_t e[4]; double gemm_flops, bytes_in, bytes_out, bytes_gemm; }; std::vector<W4Rec> g_prof_w4; }
void prof_w4_push(hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3, double gemm_flops, double bytes_in, double bytes_out, double bytes_gemm) {
  g_prof_w4.push_back(W4Rec{{e0, e1, e2, e3}, gemm_flops, bytes_in, bytes_out, bytes_gemm});
}
int prof_w4_collect(double ms[3], double* gemm_flops, double* bytes_in, double* bytes_out, double* bytes_gemm, long long* launches) {
  ms[0] = ms[1] = ms[2] = 0; *gemm_flops = 0; *bytes_in = 0; *bytes_out = 0; *bytes_gemm = 0; *launches = 0;
  for (auto& r : g_prof_w4) {
    const bool full = r.e[0] != nullptr;                     // level 1 records hold only the GEMM bracket e[1], e[2]
    if (hipEventSynchronize(r.e[full ? 3 : 2]) != hipSuccess) return 1;
    for (int i = full ? 0 : 1; i < (full ? 3 : 2); ++i) {
      float t = 0.f; if (hipEventElapsedTime(&t, r.e[i], r.e[i + 1]) != hipSuccess) return 1; ms[i] += t;
    }
    for (int i = 0; i < 4; ++i) if (r.e[i]) (void)hipEventDestroy(r.e[i]);
    *gemm_flops += r.gemm_flops; *bytes_in += r.bytes_in; *bytes_out += r.bytes_out; *bytes_gemm += r.bytes_gemm; *launches += 1;
  }
  g_prof_w4.clear();
  return 0;
}

static ProfRec g_cur;
void igemm_prof_record(const IgemmParams& p, int taps, int batch, hipStream_t st, bool begin, double exec_ratio) {
  if (!g_prof_on) return;
  if (begin) {
    // called for the Winograd launches: algorithmic flops are those of the direct 3x3 conv, executed MACs are exec_ratio of them (4/9 for F(2x2,3x3), 1/4 for F(4x4,3x3))
    (void)hipEventCreate(&g_cur.e0); (void)hipEventCreate(&g_cur.e1);
    g_cur.flops = 2.0 * (double)p.M * (double)p.N * (double)p.Cin * (double)taps * (double)batch; g_cur.taps = taps;
    g_cur.exec_flops = g_cur.flops * exec_ratio;
    g_cur.M = p.M; g_cur.N = p.N; g_cur.K = p.Cin; g_cur.batch = batch; g_cur.kind = 9;
    g_cur.bytes = 4.0 * (double)batch * ((double)p.M * p.Cin + (double)p.N * p.Cin * taps + (double)p.M * p.N * (p.res_mode ? 2.0 : 1.0));
    (void)hipEventRecord(g_cur.e0, st);
  } else {
    (void)hipEventRecord(g_cur.e1, st);
    g_prof.push_back(g_cur);
  }
}

void launch_igemm(const IgemmParams& p, int taps, bool transA, bool transB, int batch, hipStream_t st) {
  dim3 grid(cdiv(p.N, BN) * cdiv(p.M, BM), 1, batch), block(NT);
  // fewer than two 128-row tiles per CU: halve the tile height so the grid fills the chip (row-major A only)
  const bool small_grid = (taps == 1) && !transA && ((long long)grid.x * batch < 2 * 256);
  if (small_grid) grid.x = cdiv(p.N, BN) * cdiv(p.M, 64);
  IgemmParams pw = p;
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool wide = (p.N % 4 == 0) && (p.ldC % 4 == 0) && al16(p.C) && (p.sC % 4 == 0);    // a float4 of columns is all inside or all outside N
    if (p.bias_n) wide = wide && al16(p.bias_n);
    if (p.bias_bn) wide = wide && al16(p.bias_bn) && (p.ld_bias_bn % 4 == 0);
    if (p.res_mode) wide = wide && al16(p.res) && (p.ldRes % 4 == 0);
    const bool force_scalar = cur_opt().igemm_epi == 0;   // A/B switch
    pw.wide_epi = (wide && !force_scalar) ? 1 : 0;
  }
  ProfRec rec{};
  if (g_prof_on) {
    (void)hipEventCreate(&rec.e0); (void)hipEventCreate(&rec.e1);
    rec.flops = 2.0 * (double)p.M * (double)p.N * (double)p.Cin * (double)taps * (double)batch; rec.taps = taps; rec.exec_flops = rec.flops;
    rec.M = p.M; rec.N = p.N; rec.K = p.Cin; rec.batch = batch; rec.kind = (transA ? 2 : 0) + (transB ? 1 : 0);
    // algorithmic bytes: read A once, read the weights once, write C once (+ residual read once if fused)
    rec.bytes = 4.0 * (double)batch * ((double)p.M * p.Cin + (double)p.N * p.Cin * taps + (double)p.M * p.N * (p.res_mode ? 2.0 : 1.0));
    (void)hipEventRecord(rec.e0, st);
  }
  struct Fin { ProfRec& r; hipStream_t s; ~Fin() { if (g_prof_on) { (void)hipEventRecord(r.e1, s); g_prof.push_back(r); } } } fin{rec, st};
  // V: 0 = next-tile global loads issued before the MFMA block, 2 = after its first quarter (default; +4 % measured,
  // profiles/README.md), 4 = double-buffered LDS, one barrier per K step (slower: 2 blocks/CU).  A/B switch for the 3x3 kernel:
  const int variant = cur_opt().igemm_variant;
  if (taps == 9) {
    if (variant == 0) hipLaunchKernelGGL((igemm_kernel<9, false, false, 0>), grid, block, 0, st, pw);
    else if (variant == 4) hipLaunchKernelGGL((igemm_kernel<9, false, false, 4>), grid, block, 0, st, pw);
    else hipLaunchKernelGGL((igemm_kernel<9, false, false, 2>), grid, block, 0, st, pw);
  } else if (!transA && !transB && p.tag == 36 && !small_grid) {
    hipLaunchKernelGGL((igemm_kernel<1, false, false, 2, 2, 36>), grid, block, 0, st, pw);
  } else if (!transA && !transB) {
    if (small_grid) hipLaunchKernelGGL((igemm_kernel<1, false, false, 2, 1>), grid, block, 0, st, pw);
    else hipLaunchKernelGGL((igemm_kernel<1, false, false>), grid, block, 0, st, pw);
  } else if (!transA && transB) {
    if (small_grid) hipLaunchKernelGGL((igemm_kernel<1, false, true, 2, 1>), grid, block, 0, st, pw);
    else hipLaunchKernelGGL((igemm_kernel<1, false, true>), grid, block, 0, st, pw);
  } else if (transA && !transB) {
    hipLaunchKernelGGL((igemm_kernel<1, true, false>), grid, block, 0, st, pw);
  } else {
    hipLaunchKernelGGL((igemm_kernel<1, true, true>), grid, block, 0, st, pw);
  }
}

}  // namespace buddy
