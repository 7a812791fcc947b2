// Standalone check + timing of the 16-bit attention kernels (buddy_amd/csrc/attn16.hip) without torch:
//   attn16_probe check B T C prec     fp64 host reference of softmax(q k^T C^-1/2) v and its three input gradients (incl. a spiked key that forces the
//                                     deferred-rescale branch); prints abs-max relative errors
//   attn16_probe time  B T C prec n   n timed repetitions of the forward and backward launchers (HIP events), TFLOP/s of 4 / 10 T^2 C B
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I buddy_amd/csrc tools/probes/attn16_probe.hip buddy_amd/csrc/obj/attn16.o -o tools/probes/attn16_probe
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
namespace buddy {
long long flash_attn16_ws_floats(int B, int T, int C);
void launch_flash_attn16_fwd(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, int prec, float* ws, hipStream_t st);
void launch_flash_attn16_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq,
                             float* dk, float* dv, int B, int T, int C, float scale, int prec, float* ws, hipStream_t st);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned long long rs = 88172645463325252ULL;
static double urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
static float nrand() { const double u = urand() + 1e-12, v = urand(); return (float)(std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v)); }
static double relerr(const std::vector<float>& a, const std::vector<double>& r) {
  double e = 0, m = 0;
  for (size_t i = 0; i < a.size(); ++i) { e = std::fmax(e, std::fabs((double)a[i] - r[i])); m = std::fmax(m, std::fabs(r[i])); }
  return e / m;
}
int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: attn16_probe check|time B T C prec [n]\n"); return 2; }
  const bool check = !strcmp(argv[1], "check");
  const int B = atoi(argv[2]), T = atoi(argv[3]), C = atoi(argv[4]), prec = atoi(argv[5]), n = argc > 6 ? atoi(argv[6]) : 5;
  const float scale = 1.f / std::sqrt((float)C);
  const size_t N = (size_t)B * T * C;
  std::vector<float> q(N), k(N), v(N), dO(N);
  for (size_t i = 0; i < N; ++i) { q[i] = 1.5f * nrand(); k[i] = nrand(); v[i] = nrand(); dO[i] = nrand(); }
  if (check && T > 200) {      // a spiked key late in the sequence: the running maximum of query row 5 jumps far past the deferral threshold there
    for (int c = 0; c < C; ++c) k[((size_t)0 * T + (T - 70)) * C + c] = 3.f * q[((size_t)0 * T + 5) * C + c];
  }
  float *dq_, *dk_, *dv_, *dq2, *dk2, *dv2, *dO_, *O_, *L_, *D_, *ws;
  CK(hipMalloc(&dq_, N * 4)); CK(hipMalloc(&dk_, N * 4)); CK(hipMalloc(&dv_, N * 4)); CK(hipMalloc(&dO_, N * 4)); CK(hipMalloc(&O_, N * 4));
  CK(hipMalloc(&dq2, N * 4)); CK(hipMalloc(&dk2, N * 4)); CK(hipMalloc(&dv2, N * 4));
  CK(hipMalloc(&L_, (size_t)B * T * 4)); CK(hipMalloc(&D_, (size_t)B * T * 4));
  CK(hipMalloc(&ws, (size_t)buddy::flash_attn16_ws_floats(B, T, C) * 4));
  CK(hipMemcpy(dq_, q.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dk_, k.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv_, v.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dO_, dO.data(), N * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  buddy::launch_flash_attn16_fwd(dq_, dk_, dv_, O_, L_, B, T, C, scale, prec, ws, st);
  buddy::launch_flash_attn16_bwd(dq_, dk_, dv_, O_, dO_, L_, D_, dq2, dk2, dv2, B, T, C, scale, prec, ws, st);
  CK(hipStreamSynchronize(st)); CK(hipGetLastError());
  if (check) {
    std::vector<float> O(N), gq(N), gk(N), gv(N), L((size_t)B * T);
    CK(hipMemcpy(O.data(), O_, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gq.data(), dq2, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gk.data(), dk2, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gv.data(), dv2, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(L.data(), L_, (size_t)B * T * 4, hipMemcpyDeviceToHost));
    std::vector<double> rO(N, 0.0), rq(N, 0.0), rk(N, 0.0), rv(N, 0.0), rL((size_t)B * T);
    std::vector<double> P((size_t)T), dP((size_t)T);
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < T; ++i) {
        const float* qi = &q[((size_t)b * T + i) * C];
        const float* di = &dO[((size_t)b * T + i) * C];
        double mx = -1e300;
        for (int j = 0; j < T; ++j) {
          const float* kj = &k[((size_t)b * T + j) * C];
          double s = 0; for (int c = 0; c < C; ++c) s += (double)qi[c] * kj[c];
          P[j] = s * scale; mx = std::fmax(mx, P[j]);
        }
        double sum = 0; for (int j = 0; j < T; ++j) { P[j] = std::exp(P[j] - mx); sum += P[j]; }
        rL[(size_t)b * T + i] = mx + std::log(sum);
        double* Oi = &rO[((size_t)b * T + i) * C];
        for (int j = 0; j < T; ++j) {
          P[j] /= sum;
          const float* vj = &v[((size_t)b * T + j) * C];
          double d = 0;
          for (int c = 0; c < C; ++c) { Oi[c] += P[j] * vj[c]; d += (double)di[c] * vj[c]; }
          dP[j] = d;
        }
        double Dd = 0; for (int c = 0; c < C; ++c) Dd += (double)di[c] * Oi[c];
        for (int j = 0; j < T; ++j) {
          const double ds = P[j] * (dP[j] - Dd) * scale;
          const float* kj = &k[((size_t)b * T + j) * C];
          double* gqi = &rq[((size_t)b * T + i) * C]; double* gkj = &rk[((size_t)b * T + j) * C]; double* gvj = &rv[((size_t)b * T + j) * C];
          for (int c = 0; c < C; ++c) { gqi[c] += ds * kj[c]; gkj[c] += ds * qi[c]; gvj[c] += P[j] * di[c]; }
        }
      }
    double eL = 0; for (size_t i = 0; i < L.size(); ++i) eL = std::fmax(eL, std::fabs((double)L[i] - rL[i]));
    printf("check B=%d T=%d C=%d prec=%d: O %.2e lse %.2e dq %.2e dk %.2e dv %.2e\n", B, T, C, prec, relerr(O, rO), eL, relerr(gq, rq), relerr(gk, rk), relerr(gv, rv));
    return 0;
  }
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float tf = 0, tb = 0;
  for (int r = 0; r < n; ++r) {
    CK(hipEventRecord(e0, st));
    buddy::launch_flash_attn16_fwd(dq_, dk_, dv_, O_, L_, B, T, C, scale, prec, ws, st);
    CK(hipEventRecord(e1, st));
    buddy::launch_flash_attn16_bwd(dq_, dk_, dv_, O_, dO_, L_, D_, dq2, dk2, dv2, B, T, C, scale, prec, ws, st);
    CK(hipEventRecord(e2, st));
    CK(hipEventSynchronize(e2));
    float a, b2; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b2, e1, e2));
    tf += a; tb += b2;
  }
  tf /= n; tb /= n;
  const double fl = (double)T * T * C * B;
  printf("time B=%d T=%d C=%d prec=%d: fwd %.3f ms (%.0f TFLOP/s of 4T^2CB)  bwd %.3f ms (%.0f TFLOP/s of 10T^2CB, %.0f of the 14 executed)\n", B, T, C, prec, tf,
         4 * fl / tf * 1e-9, tb, 10 * fl / tb * 1e-9, 14 * fl / tb * 1e-9);
  return 0;
}
