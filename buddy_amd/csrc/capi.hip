// extern "C" boundary (include/buddy_hip.h) over the C++ graph and kernels.
#include "../../include/buddy_hip.h"
#include "common.h"
#include "net.h"

#include <cstring>

using namespace buddy;

namespace buddy {
void launch_axpby_rows(const float* x, const float* y, const float* a, const float* c, float* out, int B, int L, hipStream_t st);
void launch_row_moments(const float* x, double* out, int B, int L, hipStream_t st);
void launch_perturb(const float* x, const float* eps, float scale, float* out, long long n, hipStream_t st);
void launch_dps_update(const float* x_hat, const float* x_den, const float* lh, const float* lh_scale_b, const float* den_scale_b, const float* base, const float* d_prev,
                       float t, float dt, float w_prev, float w_cur, float* out, float* d_out, float* x_den_out, int B, int L, hipStream_t st);
void launch_row_scale(const float* x, float* out, int B, int L, int mode, float p0, float p1, hipStream_t st);
void launch_fill_rows4(float* out, int B, float v0, float v1, float v2, float v3, hipStream_t st);
void launch_mfma_ubench(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, hipStream_t st);
void launch_mfma_ubench_bf16(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, hipStream_t st);
int launch_hbm_ubench(const void* src, void* dst, long long bytes, int mode, int nt, int blocks, hipStream_t st);
void launch_fir(const float* x, const float* h, long long h_stride, float* y, int B, int L, int M, int adjoint, hipStream_t st);
}

static int finish() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return BUDDY_ERR_HIP; }
  return BUDDY_OK;
}

static NetCfg mk_cfg(int nf, const int* ch_mult, int nlev, int nrb, int n_fft, int hop) {
  NetCfg c; std::memset(&c, 0, sizeof(c));
  c.nf = nf; c.nlev = nlev; c.nrb = nrb; c.n_fft = n_fft; c.hop = hop;
  for (int i = 0; i < nlev && i < 8; ++i) c.ch_mult[i] = ch_mult[i];
  return c;
}

extern "C" {

const char* buddy_last_error(void) { return last_error(); }
int buddy_version(void) { return 1; }

int buddy_ncsnpp_param_count(int nf, const int* ch_mult, int n_levels, int num_res_blocks, long long* count) {
  if (!ch_mult || !count || n_levels < 1 || n_levels > 8) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  *count = param_count(mk_cfg(nf, ch_mult, n_levels, num_res_blocks, 510, 128));
  return BUDDY_OK;
}

int buddy_ncsnpp_create(const float* host_params, long long n_params, int nf, const int* ch_mult, int n_levels, int num_res_blocks,
                        int n_fft, int hop, void** handle) {
  if (!host_params || !ch_mult || !handle || n_levels < 1 || n_levels > 8 || nf % 32 != 0) { set_error("bad arguments (nf must be a multiple of 32)"); return BUDDY_ERR_ARG; }
  Net* N = nullptr;
  int rc = net_create(host_params, n_params, mk_cfg(nf, ch_mult, n_levels, num_res_blocks, n_fft, hop), &N);
  if (rc) return rc;
  *handle = N;
  return BUDDY_OK;
}

int buddy_ncsnpp_destroy(void* handle) { net_destroy((Net*)handle); return BUDDY_OK; }

int buddy_ncsnpp_replica(void* handle, void** replica) {
  if (!handle || !replica) { set_error("null argument"); return BUDDY_ERR_ARG; }
  Net* N = nullptr;
  int rc = net_replica((Net*)handle, &N);
  if (rc) return rc;
  *replica = N;
  return BUDDY_OK;
}

int buddy_ncsnpp_weight_bytes(void* handle, long long* params, long long* packed, long long* lazy, int* lazy_forms) {
  if (!handle) { set_error("null handle"); return BUDDY_ERR_ARG; }
  return net_weight_bytes((Net*)handle, params, packed, lazy, lazy_forms);
}

int buddy_conv3_weight_prep(const float* w_oihw, int O, int I, int dgrad, int kind, float* out, void* stream) {
  if (!w_oihw || !out || O < 1 || I < 1 || conv3_weight_floats(O, I, kind) == 0) { set_error("bad arguments (kind must be 0, 2, 4, 6 or 61)"); return BUDDY_ERR_ARG; }
  if (launch_conv3_weight_prep(w_oihw, O, I, dgrad != 0, kind, out, (hipStream_t)stream)) { set_error("F(2x2,3x3) form needs an input-channel count that is a multiple of 8"); return BUDDY_ERR_ARG; }
  return finish();
}

int buddy_ncsnpp_reserve(void* handle, int B, int L, int with_vjp, long long* bytes) {
  if (!handle) { set_error("null handle"); return BUDDY_ERR_ARG; }
  return net_reserve((Net*)handle, B, L, with_vjp, bytes);
}

int buddy_ncsnpp_forward(void* handle, const float* x, const float* cnoise, const float* cin, const float* cskip, const float* cout, float* y,
                         int B, int L, int save_for_vjp, void* stream) {
  if (!handle || !x || !cnoise || !y) { set_error("null argument"); return BUDDY_ERR_ARG; }
  if ((cin == nullptr) != (cskip == nullptr) || (cin == nullptr) != (cout == nullptr)) { set_error("cin/cskip/cout must be given together"); return BUDDY_ERR_ARG; }
  return net_forward((Net*)handle, x, cnoise, cin, cskip, cout, y, B, L, save_for_vjp, (hipStream_t)stream);
}

int buddy_ncsnpp_vjp(void* handle, const float* cot, float* grad_x, void* stream) {
  if (!handle || !cot || !grad_x) { set_error("null argument"); return BUDDY_ERR_ARG; }
  return net_vjp((Net*)handle, cot, grad_x, (hipStream_t)stream);
}

int buddy_ncsnpp_tap(void* handle, int module_idx, const float** ptr, int dims[4]) {
  if (!handle || !ptr || !dims) { set_error("null argument"); return BUDDY_ERR_ARG; }
  return net_get_tap((Net*)handle, module_idx, ptr, dims);
}

int buddy_copy_d2d(void* dst, const void* src, long long bytes, void* stream) {
  if (!dst || !src || bytes < 0) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) { set_error(std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); return BUDDY_ERR_HIP; }
  return BUDDY_OK;
}

int buddy_hbm_ubench(const void* src, void* dst, long long bytes, int mode, int nt, int blocks, void* stream) {
  if ((mode == 3 || mode == 4) && (!src || !dst)) { set_error("hbm_ubench: null buffer"); return BUDDY_ERR_ARG; }
  if ((mode != 2 && !src) || !dst || bytes < 16 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) { set_error("hbm_ubench: 16-byte aligned device buffers of >= 16 bytes"); return BUDDY_ERR_ARG; }
  if (launch_hbm_ubench(src, dst, bytes, mode, nt, blocks, (hipStream_t)stream)) { set_error("hbm_ubench: mode 0 copy | 1 read | 2 write; blocks >= 1 (grid-stride) or -1 | -2 | -4 | -8 (one chunk of that many 16-byte words per thread and workgroup)"); return BUDDY_ERR_ARG; }
  return BUDDY_OK;
}
int buddy_mfma_ubench(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, void* stream) {
  if (!seed || !out || !clk) { set_error("null argument"); return BUDDY_ERR_ARG; }
  launch_mfma_ubench(seed, out, blocks, iters, clk, (hipStream_t)stream);
  return finish();
}

int buddy_mfma_ubench_bf16(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, void* stream) {
  if (!seed || !out || !clk) { set_error("null argument"); return BUDDY_ERR_ARG; }
  launch_mfma_ubench_bf16(seed, out, blocks, iters, clk, (hipStream_t)stream);
  return finish();
}

int buddy_prof_enable(int on) { igemm_prof_enable(on); return BUDDY_OK; }
int buddy_prof_collect(double* ms, double* flops, long long* launches, double* bytes, double* exec_flops) {
  if (!ms || !flops || !launches || !bytes || !exec_flops) { set_error("null argument"); return BUDDY_ERR_ARG; }
  if (igemm_prof_collect(ms, flops, launches, bytes, exec_flops)) { set_error("event timing failed"); return BUDDY_ERR_HIP; }
  return BUDDY_OK;
}

int buddy_prof_collect_hbm(double* ms, double* bytes, long long* launches) {
  if (!ms || !bytes || !launches) { set_error("null argument"); return BUDDY_ERR_ARG; }
  if (prof_hbm_collect(ms, bytes, launches)) { set_error("event timing failed"); return BUDDY_ERR_HIP; }
  return BUDDY_OK;
}

int buddy_prof_collect_wino4(double* ms, double* gemm_flops, double* bytes_in, double* bytes_out, double* bytes_gemm, long long* launches) {
  if (!ms || !gemm_flops || !bytes_in || !bytes_out || !bytes_gemm || !launches) { set_error("null argument"); return BUDDY_ERR_ARG; }
  if (prof_w4_collect(ms, gemm_flops, bytes_in, bytes_out, bytes_gemm, launches)) { set_error("event timing failed"); return BUDDY_ERR_HIP; }
  return BUDDY_OK;
}

static int gemm_impl(const float* A, int ldA, int transA, const float* Bt, int ldB, int transB, float* C, int ldC, int M, int N, int K, float alpha,
                     const float* bias_n, int accumulate, int batch, long long strideA, long long strideB, long long strideC, int tag, void* stream) {
  if (!A || !Bt || !C || K % 4 || (transA && M % 4) || (transB && N % 4)) { set_error("bad gemm arguments (K, and M/N of k-major operands, must be multiples of 4)"); return BUDDY_ERR_ARG; }
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = A; p.ldA0 = ldA; p.Cin = K; p.M = M; p.N = N; p.Bt = Bt; p.ldB = ldB; p.C = C; p.ldC = ldC; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.alpha = alpha; p.out_scale = 1.f; p.bias_n = bias_n; p.accumulate = accumulate; p.H = 1; p.W = 1; p.rows_per_batch = 1;
  p.tag = tag;
  launch_igemm(p, 1, transA != 0, transB != 0, batch, (hipStream_t)stream);
  return finish();
}

int buddy_gemm(const float* A, int ldA, int transA, const float* Bt, int ldB, int transB, float* C, int ldC, int M, int N, int K, float alpha,
               const float* bias_n, int accumulate, int batch, long long strideA, long long strideB, long long strideC, void* stream) {
  return gemm_impl(A, ldA, transA, Bt, ldB, transB, C, ldC, M, N, K, alpha, bias_n, accumulate, batch, strideA, strideB, strideC, 0, stream);
}

int buddy_gemm_winograd_domain(const float* V, const float* U, float* Mo, int tiles, int Cout, int Cin, int positions, void* stream) {
  if (positions < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  return gemm_impl(V, Cin, 0, U, Cin, 0, Mo, Cout, tiles, Cout, Cin, 1.f, nullptr, 0, positions, (long long)tiles * Cin, (long long)Cout * Cin,
                   (long long)tiles * Cout, 36, stream);
}

long long buddy_wgemm_packed_bytes(int positions, int Cout, int Cin) {
  return (positions < 1 || !wgemm_supported(Cout, Cin)) ? 0 : (long long)wgemm_packed_bytes(positions, Cout, Cin);
}

int buddy_wgemm_pack_weights(const float* U, void* U3, int positions, int Cout, int Cin, void* stream) {
  if (!U || !U3 || positions < 1 || !wgemm_supported(Cout, Cin)) { set_error("bad arguments (Cout % 128, Cin % 32)"); return BUDDY_ERR_ARG; }
  wgemm_pack_weights(U, U3, positions, Cout, Cin, (hipStream_t)stream);
  return finish();
}

int buddy_gemm_winograd_domain_bf16x3(const float* V, const void* U3, float* Mo, int tiles, int Cout, int Cin, int positions, void* stream) {
  if (!V || !U3 || !Mo || tiles < 1 || positions < 1 || !wgemm_supported(Cout, Cin)) { set_error("bad arguments (Cout % 128, Cin % 32)"); return BUDDY_ERR_ARG; }
  launch_wgemm_bf16x3(V, U3, Mo, tiles, Cout, Cin, positions, (hipStream_t)stream);
  return finish();
}

long long buddy_wgemm_f16x2_packed_bytes(int positions, int Cout, int Cin) {
  return (positions < 1 || positions > 64 || !wgemm_f16x2_supported(Cout, Cin)) ? 0 : (long long)wgemm_f16x2_packed_bytes(positions, Cout, Cin);
}
int buddy_wgemm_f16x2_pack_weights(const float* U, void* U2, int positions, int Cout, int Cin, void* stream) {
  if (!U || !U2 || positions < 1 || positions > 64 || !wgemm_f16x2_supported(Cout, Cin)) { set_error("bad arguments (positions <= 64, Cout % 128, Cin % 64)"); return BUDDY_ERR_ARG; }
  wgemm_f16x2_pack_weights(U, U2, positions, Cout, Cin, (hipStream_t)stream);
  return finish();
}
int buddy_gemm_winograd_domain_f16x2(const float* V, const void* U2, float* Mo, int tiles, int Cout, int Cin, int positions, const unsigned* vmax,
                                     int tiles_per_utt, void* stream) {
  if (!V || !U2 || !Mo || !vmax || tiles < 1 || tiles_per_utt < 32 || tiles % tiles_per_utt || positions < 1 || positions > 64 || !wgemm_f16x2_supported(Cout, Cin)) {
    set_error("bad arguments (positions <= 64, Cout % 128, Cin % 64, tiles a multiple of tiles_per_utt >= 32)"); return BUDDY_ERR_ARG;
  }
  launch_wgemm_f16x2(V, U2, Mo, tiles, Cout, Cin, positions, vmax, tiles_per_utt, (hipStream_t)stream);
  return finish();
}
int buddy_abs_max_bits(const float* x, int groups, int segments, long long seg_len, unsigned* out, void* stream) {
  if (!x || !out || groups < 1 || segments < 1 || seg_len < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  launch_abs_max_bits(x, groups, segments, seg_len, out, (hipStream_t)stream);
  return finish();
}

int buddy_gemm_bf16x3(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W3, float* Cm, int ldC, long long M, int N, int K,
                      const float* bias_n, float alpha, int accumulate, void* stream) {
  if (!A0 || !W3 || !Cm || M < 1 || !wgemm_general_supported(N, K, A1 ? C0 : 0, ldA0, A1 ? ldA1 : 0, ldC, A0, A1, Cm, bias_n)) {
    set_error("bad arguments (N % 128, K % 32, C0 % 32, 16-byte aligned rows)"); return BUDDY_ERR_ARG;
  }
  launch_wgemm_bf16x3_general(A0, ldA0, A1, ldA1, C0, W3, Cm, ldC, M, N, K, bias_n, alpha, accumulate, (hipStream_t)stream);
  return finish();
}

// the same general form in f16x2 arithmetic (W2 = buddy_wgemm_f16x2_pack_weights(W, ., 1, N, K)); gn_bwd != 0 arguments as buddy_gemm_bf16x3_gn_bwd
int buddy_gemm_f16x2(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W2, float* Cm, int ldC, long long M, int N, int K,
                     const float* bias_n, float alpha, int accumulate, void* stream) {
  if (!A0 || !W2 || !Cm || M < 1 || !wgemm_f16x2_general_supported(N, K, A1 ? C0 : 0, ldA0, A1 ? ldA1 : 0, ldC, A0, A1, Cm, bias_n)) {
    set_error("bad arguments (N % 128, K % 64, C0 % 32, 16-byte aligned rows)"); return BUDDY_ERR_ARG;
  }
  launch_wgemm_f16x2_general(A0, ldA0, A1, ldA1, C0, W2, Cm, ldC, M, N, K, bias_n, alpha, accumulate, (hipStream_t)stream);
  return finish();
}
int buddy_gemm_f16x2_gn_bwd(const float* A, int ldA, const void* W2, const float* x0, const float* x1, int C0, const float* da, const float* stats,
                            const float* gamma, const float* beta, int G, int silu, float alpha, float* dx0, float* dx1, int acc0, int acc1,
                            double* stat_scratch, float* red, int B, int HW, int N, int K, void* stream) {
  if (!A || !W2 || !x0 || !da || !stats || !gamma || !beta || !dx0 || !stat_scratch || !red || B < 1 || HW < 1 || G < 1 || N % 4 || (N / G) % 4 || N > 1024 ||
      (x1 && (C0 % 4 || C0 < 4 || C0 >= N)) || (x1 != nullptr) != (dx1 != nullptr)) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  Src2 x; x.p0 = x0; x.p1 = x1; x.C0 = x1 ? C0 : N; x.ld0 = x1 ? C0 : N; x.ld1 = x1 ? N - C0 : 0;
  Dst2 d; d.p0 = dx0; d.p1 = dx1; d.C0 = x.C0; d.ld0 = x.ld0; d.ld1 = x.ld1; d.acc0 = acc0; d.acc1 = acc1;
  if (!wgemm_gnbwd_supported(N, K, ldA, x, d, A, da) || !wgemm_f16x2_supported(N, K)) { set_error("bad arguments (N % 128, K % 64, C0 % 128, 16-byte aligned rows)"); return BUDDY_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  launch_gn_bwd_sums(x, stats, gamma, beta, da, B, 1, HW, N, G, 0, silu, stat_scratch, red, st);
  launch_wgemm_f16x2_gnbwd(A, ldA, W2, (long long)B * HW, N, K, alpha, x, da, stats, red, gamma, beta, G, silu, HW, d, st);
  return finish();
}

int buddy_gemm_bf16x3_gn_bwd(const float* A, int ldA, const void* W3, const float* x0, const float* x1, int C0, const float* da, const float* stats,
                             const float* gamma, const float* beta, int G, int silu, float alpha, float* dx0, float* dx1, int acc0, int acc1,
                             double* stat_scratch, float* red, int B, int HW, int N, int K, void* stream) {
  if (!A || !W3 || !x0 || !da || !stats || !gamma || !beta || !dx0 || !stat_scratch || !red || B < 1 || HW < 1 || G < 1 || N % 4 || (N / G) % 4 || N > 1024 ||
      (x1 && (C0 % 4 || C0 < 4 || C0 >= N)) || (x1 != nullptr) != (dx1 != nullptr)) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  Src2 x; x.p0 = x0; x.p1 = x1; x.C0 = x1 ? C0 : N; x.ld0 = x1 ? C0 : N; x.ld1 = x1 ? N - C0 : 0;
  Dst2 d; d.p0 = dx0; d.p1 = dx1; d.C0 = x.C0; d.ld0 = x.ld0; d.ld1 = x.ld1; d.acc0 = acc0; d.acc1 = acc1;
  if (!wgemm_gnbwd_supported(N, K, ldA, x, d, A, da)) { set_error("bad arguments (N % 128, K % 32, C0 % 128, 16-byte aligned rows)"); return BUDDY_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  launch_gn_bwd_sums(x, stats, gamma, beta, da, B, 1, HW, N, G, 0, silu, stat_scratch, red, st);
  launch_wgemm_bf16x3_gnbwd(A, ldA, W3, (long long)B * HW, N, K, alpha, x, da, stats, red, gamma, beta, G, silu, HW, d, st);
  return finish();
}

int buddy_conv3x3(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!x || !wt || !y || Cin % 4) { set_error("bad conv arguments"); return BUDDY_ERR_ARG; }
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = x; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.Bt = wt; p.ldB = 9 * Cin; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  launch_igemm(p, 9, false, false, 1, (hipStream_t)stream);
  return finish();
}

int buddy_winograd_transform_weights(const float* wt_host, int Cout, int Cin, float* U_host) {
  if (!wt_host || !U_host || Cin % 16) { set_error("bad arguments (Cin must be a multiple of 16)"); return BUDDY_ERR_ARG; }
  wino_transform_weights(wt_host, Cout, Cin, U_host);
  return BUDDY_OK;
}

int buddy_conv3x3_winograd(const float* x, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!x || !U || !y) { set_error("null argument"); return BUDDY_ERR_ARG; }
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = x; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino_supported(p)) { set_error("shape not supported by the Winograd kernel (H, W even; Cin % 16; Cout % 32)"); return BUDDY_ERR_ARG; }
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, true);
  launch_wino(p, U, (hipStream_t)stream);
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, false);
  return finish();
}

int buddy_winograd4_transform_weights(const float* wt_host, int Cout, int Cin, float* U4_host) {
  if (!wt_host || !U4_host || Cout < 1 || Cin < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  wino4_transform_weights(wt_host, Cout, Cin, U4_host);
  return BUDDY_OK;
}
int buddy_conv3x3_winograd4(const float* x, const float* U4, const float* bias, float* y, float* scratch, int B, int H, int W, int Cin, int Cout,
                            void* stream) {
  if (!x || !U4 || !y || !scratch) { set_error("null argument"); return BUDDY_ERR_ARG; }
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = x; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino4_supported(p)) { set_error("shape not supported by the F(4x4,3x3) path (H, W, Cin, Cout multiples of 4)"); return BUDDY_ERR_ARG; }
  long long vf = 0, mf = 0; wino4_scratch(p, &vf, &mf);
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, true, 0.25);
  launch_wino4(p, U4, scratch, scratch + vf, (hipStream_t)stream);
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, false, 0.25);
  return finish();
}

int buddy_winograd6_transform_weights(const float* wt_host, int Cout, int Cin, float* U6_host) {
  if (!wt_host || !U6_host || Cout < 1 || Cin < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  wino6_transform_weights(wt_host, Cout, Cin, U6_host);
  return BUDDY_OK;
}
int buddy_conv3x3_winograd6(const float* x, const float* U6, const float* bias, float* y, float* scratch, int B, int H, int W, int Cin, int Cout,
                            void* stream) {
  if (!x || !U6 || !y || !scratch) { set_error("null argument"); return BUDDY_ERR_ARG; }
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = x; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino6_supported(p)) { set_error("shape not supported by the F(6x6,3x3) path (H, W >= 6; Cin, Cout multiples of 4)"); return BUDDY_ERR_ARG; }
  long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf);
  const double xr = wino6_exec_ratio(p);
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, true, xr);
  launch_wino6(p, U6, scratch, scratch + vf, (hipStream_t)stream);
  igemm_prof_record(p, 9, 1, (hipStream_t)stream, false, xr);
  return finish();
}

static int gn_conv3x3_winograd(bool f6, const float* x0, const float* x1, int C0, const float* gamma, const float* beta, int G, int silu, const float* U4,
                               const float* bias, float* y, float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W,
                               int Cin, int Cout, void* stream) {
  if (!x0 || !gamma || !beta || !U4 || !y || !scratch || !stats || !stat_scratch || G < 1 || Cin % 4 || (Cin / G) % 4 || Cin > 1024 ||
      (x1 && (C0 % 4 || C0 < 4 || C0 >= Cin))) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!(f6 ? wino6_supported(p) : wino4_supported(p))) { set_error("shape not supported by this Winograd path"); return BUDDY_ERR_ARG; }
  const int sc = csum ? (f6 ? wino6_stat_chunks(p) : wino4_stat_chunks(p)) : 0;
  if (csum && (sc == 0 || (long long)sc * Cout > 256LL * 1024)) { set_error("shape not supported by the statistics epilogue"); return BUDDY_ERR_ARG; }
  W4Gn gn;
  gn.x.p0 = x0; gn.x.p1 = x1; gn.x.C0 = x1 ? C0 : Cin; gn.x.ld0 = x1 ? C0 : Cin; gn.x.ld1 = x1 ? Cin - C0 : 0;
  gn.stats = stats; gn.gamma = gamma; gn.beta = beta; gn.G = G; gn.silu = silu;
  launch_gn_stats(gn.x, B, H * W, Cin, G, 1e-6f, stat_scratch, stats, st);
  long long vf = 0, mf = 0;
  if (f6) { wino6_scratch(p, &vf, &mf); launch_wino6(p, U4, scratch, scratch + vf, st, &gn, csum ? stat_scratch : nullptr); }
  else { wino4_scratch(p, &vf, &mf); launch_wino4(p, U4, scratch, scratch + vf, st, &gn, csum ? stat_scratch : nullptr); }
  if (csum) launch_csum_collapse(stat_scratch, sc, B, Cout, csum, st);
  return finish();
}
int buddy_gn_conv3x3_winograd4(const float* x0, const float* x1, int C0, const float* gamma, const float* beta, int G, int silu, const float* U4,
                               const float* bias, float* y, float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W,
                               int Cin, int Cout, void* stream) {
  return gn_conv3x3_winograd(false, x0, x1, C0, gamma, beta, G, silu, U4, bias, y, scratch, stats, stat_scratch, csum, B, H, W, Cin, Cout, stream);
}
int buddy_gn_conv3x3_winograd6(const float* x0, const float* x1, int C0, const float* gamma, const float* beta, int G, int silu, const float* U6,
                               const float* bias, float* y, float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W,
                               int Cin, int Cout, void* stream) {
  return gn_conv3x3_winograd(true, x0, x1, C0, gamma, beta, G, silu, U6, bias, y, scratch, stats, stat_scratch, csum, B, H, W, Cin, Cout, stream);
}

int buddy_conv3x3_winograd6_gn_bwd_sums(const float* g, const float* U6, float* da, float* scratch, const float* x0, const float* x1, int C0,
                                        const float* stats, const float* gamma, const float* beta, int G, int silu, double* stat_scratch,
                                        double* chsum, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!g || !U6 || !da || !scratch || !x0 || !stats || !gamma || !beta || !stat_scratch || !chsum || G < 1 || Cout % 4 || (Cout / G) % 4 ||
      Cout > 1024 || (x1 && (C0 % 4 || C0 < 4 || C0 >= Cout))) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.A0 = g; p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = da; p.ldC = Cout;
  p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino6_supported(p)) { set_error("shape not supported by the F(6x6,3x3) path (H, W >= 6; Cin, Cout multiples of 4)"); return BUDDY_ERR_ARG; }
  const int sc = wino6_stat_chunks(p);
  if (sc == 0 || (long long)sc * Cout > 256LL * 1024) { set_error("shape not supported by the statistics epilogue"); return BUDDY_ERR_ARG; }
  W4Gn bg;
  bg.x.p0 = x0; bg.x.p1 = x1; bg.x.C0 = x1 ? C0 : Cout; bg.x.ld0 = x1 ? C0 : Cout; bg.x.ld1 = x1 ? Cout - C0 : 0;
  bg.stats = stats; bg.gamma = gamma; bg.beta = beta; bg.G = G; bg.silu = silu;
  long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf);
  launch_wino6(p, U6, scratch, scratch + vf, st, nullptr, stat_scratch, &bg);
  launch_csum_collapse(stat_scratch, sc, B, Cout, chsum, st);
  return finish();
}

int buddy_gnbwd_conv3x3_winograd6(const float* x, const float* gamma, const float* beta, const float* stats, const float* da, int G, int silu,
                                  const float* U6, float* y, float* scratch, double* stat_scratch, float* red, int B, int H, int W, int C, int Cout,
                                  void* stream) {
  if (!x || !gamma || !beta || !stats || !da || !U6 || !y || !scratch || !stat_scratch || !red || G < 1 || C % 4 || (C / G) % 4 || C > 1024) {
    set_error("bad arguments"); return BUDDY_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.ldA0 = C; p.Cin = C; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino6_supported(p)) { set_error("shape not supported by the F(6x6,3x3) path (H, W >= 6; channels multiples of 4)"); return BUDDY_ERR_ARG; }
  W4Gn gn;
  gn.x.p0 = x; gn.x.p1 = nullptr; gn.x.C0 = C; gn.x.ld0 = C; gn.x.ld1 = 0;
  gn.stats = stats; gn.gamma = gamma; gn.beta = beta; gn.G = G; gn.silu = silu; gn.da = da; gn.ldda = C; gn.red = red;
  launch_gn_bwd_sums(gn.x, stats, gamma, beta, da, B, H, W, C, G, 0, silu, stat_scratch, red, st);
  long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf);
  launch_wino6(p, U6, scratch, scratch + vf, st, &gn);
  return finish();
}

int buddy_gn_upconv3x3_winograd6(const float* x, const float* gamma, const float* beta, int G, int silu, const float* U6up, const float* bias, float* y,
                                 float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!x || !gamma || !beta || !U6up || !y || !scratch || !stats || !stat_scratch || G < 1 || Cin % 4 || (Cin / G) % 4 || Cin > 1024 || Cout % 4) {
    set_error("bad arguments"); return BUDDY_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.ldA0 = Cin; p.Cin = Cin; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.bias_n = bias; p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino6_supported(p)) { set_error("shape not supported by the F(6x6,3x3) path (H, W >= 6; Cin, Cout multiples of 4)"); return BUDDY_ERR_ARG; }
  const int sc = csum ? wino6_stat_chunks(p, 1) : 0;
  if (csum && (sc == 0 || (long long)sc * Cout > 256LL * 1024)) { set_error("shape not supported by the statistics epilogue"); return BUDDY_ERR_ARG; }
  W4Gn gn;
  gn.x.p0 = x; gn.x.p1 = nullptr; gn.x.C0 = Cin; gn.x.ld0 = Cin; gn.x.ld1 = 0;
  gn.stats = stats; gn.gamma = gamma; gn.beta = beta; gn.G = G; gn.silu = silu;
  launch_gn_stats(gn.x, B, H * W, Cin, G, 1e-6f, stat_scratch, stats, st);
  long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf, 1);
  launch_wino6(p, U6up, scratch, scratch + vf, st, &gn, csum ? stat_scratch : nullptr, nullptr, nullptr, 1);
  if (csum) launch_csum_collapse(stat_scratch, sc, B, Cout, csum, st);
  return finish();
}

int buddy_gnbwd_upconv3x3_winograd6(const float* h, const float* gamma, const float* beta, const float* stats, const float* da, int G, int silu,
                                    const float* U6upT, float* y, float* scratch, double* stat_scratch, float* red, int B, int H, int W, int C,
                                    int Cout, void* stream) {
  if (!h || !gamma || !beta || !stats || !da || !U6upT || !y || !scratch || !stat_scratch || !red || G < 1 || C % 4 || (C / G) % 4 || C > 1024 ||
      Cout % 4) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  IgemmParams p; std::memset(&p, 0, sizeof(p));
  p.ldA0 = C; p.Cin = C; p.H = H; p.W = W; p.M = B * H * W; p.N = Cout; p.C = y; p.ldC = Cout;
  p.alpha = 1.f; p.out_scale = 1.f; p.rows_per_batch = H * W;
  if (!wino6_supported(p)) { set_error("shape not supported by the F(6x6,3x3) path (H, W >= 6; channels multiples of 4)"); return BUDDY_ERR_ARG; }
  W4Gn gn;
  gn.x.p0 = h; gn.x.p1 = nullptr; gn.x.C0 = C; gn.x.ld0 = C; gn.x.ld1 = 0;
  gn.stats = stats; gn.gamma = gamma; gn.beta = beta; gn.G = G; gn.silu = silu; gn.da = da; gn.ldda = C; gn.red = red;
  launch_gn_bwd_sums(gn.x, stats, gamma, beta, da, B, 2 * H, 2 * W, C, G, 0, silu, stat_scratch, red, st);
  long long vf = 0, mf = 0; wino6_scratch(p, &vf, &mf, 2);
  launch_wino6(p, U6upT, scratch, scratch + vf, st, &gn, nullptr, nullptr, nullptr, 2);
  return finish();
}

int buddy_groupnorm_act(const float* x, const float* gamma, const float* beta, float* y, float* stats, void* scratch, int B, int H, int W, int C,
                        int G, int mode, int silu, void* stream) {
  if (!x || !y || !stats || !scratch || C % 4 || (C / G) % 4 || C > 1024) { set_error("bad groupnorm arguments"); return BUDDY_ERR_ARG; }
  Src2 s; s.p0 = x; s.p1 = nullptr; s.C0 = C; s.ld0 = C; s.ld1 = 0;
  launch_gn_stats(s, B, H * W, C, G, 1e-6f, (double*)scratch, stats, (hipStream_t)stream);
  launch_gn_apply(s, stats, gamma, beta, B, H, W, C, G, mode, silu, y, nullptr, (hipStream_t)stream);
  return finish();
}

int buddy_groupnorm_act_bwd(const float* x, const float* gamma, const float* beta, const float* stats, const float* dy, float* dx, void* scratch,
                            float* red, int B, int H, int W, int C, int G, int mode, int silu, void* stream) {
  if (!x || !dy || !dx || !stats || !scratch || !red || C % 4 || (C / G) % 4 || C > 1024) { set_error("bad groupnorm arguments"); return BUDDY_ERR_ARG; }
  Src2 s; s.p0 = x; s.p1 = nullptr; s.C0 = C; s.ld0 = C; s.ld1 = 0;
  Dst2 d; d.p0 = dx; d.p1 = nullptr; d.C0 = C; d.ld0 = C; d.ld1 = 0; d.acc0 = 0; d.acc1 = 0;
  launch_gn_bwd(s, stats, gamma, beta, dy, B, H, W, C, G, mode, silu, nullptr, 0, 0.f, (double*)scratch, red, d, (hipStream_t)stream);
  return finish();
}

int buddy_options_check(void) { return options_check(); }
int buddy_option_validate(const char* key, int value) { Options o = default_options(); return option_set(o, key, value); }
int buddy_ncsnpp_set_option(void* h, const char* key, int value) { if (!h) { set_error("null handle"); return BUDDY_ERR_ARG; } return net_set_option((Net*)h, key, value); }
int buddy_ncsnpp_get_option(void* h, const char* key, int* value) { if (!h) { set_error("null handle"); return BUDDY_ERR_ARG; } return net_get_option((Net*)h, key, value); }
int buddy_ncsnpp_set_gemm(void* h, int mode) { if (!h) { set_error("null handle"); return BUDDY_ERR_ARG; } return net_set_gemm((Net*)h, mode); }
int buddy_ncsnpp_set_attention(void* h, int mode) { if (!h) { set_error("null handle"); return BUDDY_ERR_ARG; } return net_set_attention((Net*)h, mode); }
int buddy_ncsnpp_set_fir(void* h, int fir) { if (!h) { set_error("null handle"); return BUDDY_ERR_ARG; } return net_set_fir((Net*)h, fir); }
int buddy_fir_resample2(const float* x, float* y, int B, int H, int W, int C, int up, float scale, int accumulate, void* stream) {
  if (!x || !y || B < 1 || H < 1 || W < 1 || C < 1 || (!up && ((H | W) & 1))) { set_error("bad resample arguments"); return BUDDY_ERR_ARG; }
  if (up) launch_fir_up2(x, y, B, H, W, C, scale, accumulate, (hipStream_t)stream);
  else launch_fir_down2(x, y, B, H, W, C, scale, accumulate, (hipStream_t)stream);
  return finish();
}

int buddy_flash_attention_fwd(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int prec,
                              void* stream) {
  if (!q || !k || !v || !O || !lse || B < 1 || T < 1 || !flash_attn_supported(C) || prec != 0) {
    set_error("bad attention arguments (C in {64, 128, 256}; prec 0 -- the 16-bit-operand kernels are buddy_flash_attention16_*)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn_fwd(q, k, v, O, lse, B, T, C, scale, nullptr, 1, (hipStream_t)stream);
  return finish();
}
long long buddy_flash_attention16_workspace(int B, int T, int C) { return (B < 1 || T < 1 || !flash_attn_supported(C)) ? 0 : flash_attn16_ws_floats(B, T, C); }
int buddy_flash_attention16_fwd(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int prec, float* ws,
                                void* stream) {
  if (!q || !k || !v || !O || !lse || !ws || B < 1 || T < 1 || !flash_attn_supported(C) || prec < 1 || prec > 2) {
    set_error("bad attention arguments (C in {64, 128, 256}; prec 1 = bf16, 2 = f16; ws = buddy_flash_attention16_workspace floats)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn16_fwd(q, k, v, O, lse, B, T, C, scale, prec, ws, (hipStream_t)stream);
  return finish();
}
int buddy_flash_attention16_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta, float* dq,
                                float* dk, float* dv, int B, int T, int C, float scale, int prec, float* ws, void* stream) {
  if (!q || !k || !v || !O || !dO || !lse || !delta || !dq || !dk || !dv || !ws || B < 1 || T < 1 || !flash_attn_supported(C) || prec < 1 || prec > 2) {
    set_error("bad attention arguments (C in {64, 128, 256}; prec 1 = bf16, 2 = f16; ws = buddy_flash_attention16_workspace floats)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn16_bwd(q, k, v, O, dO, lse, delta, dq, dk, dv, B, T, C, scale, prec, ws, (hipStream_t)stream);
  return finish();
}
long long buddy_flash_attention_workspace(int B, int T, int C, int splits) {
  if (B < 1 || T < 1 || !flash_attn_supported(C)) return 0;
  return flash_attn_ws_floats(B, T, C, splits > 0 ? splits : flash_attn_splits(B, T));
}
int buddy_flash_attention_splits(int B, int T) { return (B < 1 || T < 1) ? 1 : flash_attn_splits(B, T); }
static bool split_args_ok(int T, int splits, const float* ws) {
  const int nb = (T + 31) / 32;
  if (splits < 1 || splits > nb || (splits > 1 && !ws)) return false;
  return splits == 1 || (long long)((nb + splits - 1) / splits) * (splits - 1) < nb;     // every split non-empty
}
int buddy_flash_attention_fwd_split(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int splits,
                                    float* ws, void* stream) {
  if (!q || !k || !v || !O || !lse || B < 1 || T < 1 || !flash_attn_supported(C) || !split_args_ok(T, splits, ws)) {
    set_error("bad split-attention arguments (C in {64, 128, 256}; every split needs at least one 32-row block)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn_fwd(q, k, v, O, lse, B, T, C, scale, ws, splits, (hipStream_t)stream);
  return finish();
}
int buddy_flash_attention_bwd_split(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta,
                                    float* dq, float* dk, float* dv, int B, int T, int C, float scale, int splits, float* ws, void* stream) {
  if (!q || !k || !v || !O || !dO || !lse || !delta || !dq || !dk || !dv || B < 1 || T < 1 || !flash_attn_supported(C) || !split_args_ok(T, splits, ws)) {
    set_error("bad split-attention arguments (C in {64, 128, 256}; every split needs at least one 32-row block)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn_bwd(q, k, v, O, dO, lse, delta, dq, dk, dv, B, T, C, scale, ws, splits, (hipStream_t)stream);
  return finish();
}
int buddy_flash_attention_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta, float* dq,
                              float* dk, float* dv, int B, int T, int C, float scale, int prec, void* stream) {
  if (prec != 0 || !q || !k || !v || !O || !dO || !lse || !delta || !dq || !dk || !dv || B < 1 || T < 1 || !flash_attn_supported(C)) {
    set_error("bad attention arguments (C in {64, 128, 256}; prec 0 -- the 16-bit-operand kernels are buddy_flash_attention16_*)"); return BUDDY_ERR_ARG;
  }
  launch_flash_attn_bwd(q, k, v, O, dO, lse, delta, dq, dk, dv, B, T, C, scale, nullptr, 1, (hipStream_t)stream);
  return finish();
}

int buddy_axpby_rows(const float* x, const float* y, const float* a, const float* c, float* out, int B, int L, void* stream) {
  if (!x || !a || !out || (y && !c)) { set_error("null argument"); return BUDDY_ERR_ARG; }
  launch_axpby_rows(x, y, a, c, out, B, L, (hipStream_t)stream);
  return finish();
}

int buddy_perturb(const float* x, const float* eps, float scale, float* out, long long n, void* stream) {
  if (!x || !eps || !out) { set_error("null argument"); return BUDDY_ERR_ARG; }
  launch_perturb(x, eps, scale, out, n, (hipStream_t)stream);
  return finish();
}

int buddy_dps_update(const float* x_hat, const float* x_den, const float* lh, const float* lh_scale, const float* den_scale, const float* base,
                     const float* d_prev, float t, float dt, float w_prev, float w_cur, float* out, float* d_out, float* x_den_out, int B, int L, void* stream) {
  if (!x_hat || !x_den || !base || !out || t <= 0.f) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  launch_dps_update(x_hat, x_den, lh, lh_scale, den_scale, base, d_prev, t, dt, w_prev, w_cur, out, d_out, x_den_out, B, L, (hipStream_t)stream);
  return finish();
}
int buddy_row_scale(const float* x, float* out, int B, int L, int mode, float p0, float p1, void* stream) {
  if (!x || !out || B < 1 || L < 2 || (mode != 0 && mode != 1) || (mode == 1 && !(p1 > 0.f))) { set_error("row_scale: mode 0 (p0 / std) or 1 (p0 / (norm / p1 + 1e-8)), L >= 2"); return BUDDY_ERR_ARG; }
  launch_row_scale(x, out, B, L, mode, p0, p1, (hipStream_t)stream);
  return finish();
}
int buddy_fill_rows4(float* out, int B, float v0, float v1, float v2, float v3, void* stream) {
  if (!out || B < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  launch_fill_rows4(out, B, v0, v1, v2, v3, (hipStream_t)stream);
  return finish();
}

int buddy_row_moments(const float* x, double* out, int B, int L, void* stream) {
  if (!x || !out) { set_error("null argument"); return BUDDY_ERR_ARG; }
  launch_row_moments(x, out, B, L, (hipStream_t)stream);
  return finish();
}

int buddy_fir(const float* x, const float* h, long long h_stride, float* y, int B, int L, int M, int adjoint, void* stream) {
  if (!x || !h || !y || M < 1) { set_error("bad fir arguments"); return BUDDY_ERR_ARG; }
  launch_fir(x, h, h_stride, y, B, L, M, adjoint, (hipStream_t)stream);
  return finish();
}


// ---- WPE warm start ----
int buddy_wpe(const double* Y, double* X, double* scratch, int rows, int T, int taps, int delay, int iterations, void* stream) {
  if (!Y || !X || !scratch || rows < 1 || T < 1 || taps < 1 || taps > 56 || delay < 0 || iterations < 0) { set_error("bad wpe arguments (taps <= 56)"); return BUDDY_ERR_ARG; }
  launch_wpe(Y, X, scratch, rows, T, taps, delay, iterations, (hipStream_t)stream);
  return finish();
}

long long buddy_wpe_workspace_bytes(int B, int L) { return (B < 1 || L < 1) ? 0 : (long long)wpe_workspace_bytes(B, L); }

int buddy_wpe_dereverb(const float* y, float* out, void* workspace, int B, int L, int taps, int delay, int iterations, void* stream) {
  if (!y || !out || !workspace || B < 1 || L < 1 || taps < 1 || taps > 56 || delay < 0 || iterations < 0) { set_error("bad wpe arguments (taps <= 56)"); return BUDDY_ERR_ARG; }
  launch_wpe_dereverb(y, out, workspace, B, L, taps, delay, iterations, (hipStream_t)stream);
  return finish();
}

// ---- blind operator ----
int buddy_blindop_create(int U, int L, int Nf, int E, int num_knots, const float* knots, int sample_rate, float comp, float min_decay,
                         float max_decay, float w_lo, float w_hi, int clamp_decay, int long_second, void** handle) {
  if (!knots || !handle || U < 1 || L < 1024 || num_knots > 64) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  BlindOpCfg c; std::memset(&c, 0, sizeof(c));
  c.n_fft = 1024; c.win = 512; c.hop = 128; c.Nf = Nf; c.E = E; c.num_knots = num_knots; c.sample_rate = sample_rate; c.comp = comp;
  for (int i = 0; i < num_knots; ++i) c.knots[i] = knots[i];
  c.min_decay = min_decay; c.max_decay = max_decay; c.w_lo = w_lo; c.w_hi = w_hi; c.clamp_decay = clamp_decay; c.long2nd = long_second;
  BlindOp* o = nullptr;
  int rc = blindop_create(c, U, L, &o);
  if (rc) return rc;
  *handle = o;
  return BUDDY_OK;
}
int buddy_blindop_destroy(void* h) { blindop_destroy((BlindOp*)h); return BUDDY_OK; }
#define BOP_CHECK(h) if (!(h)) { set_error("null handle"); return BUDDY_ERR_ARG; }
int buddy_blindop_set_params(void* h, const float* decay, const float* weights, const float* phases, int reset_adam, void* stream) {
  BOP_CHECK(h); return blindop_set_params((BlindOp*)h, decay, weights, phases, reset_adam, (hipStream_t)stream);
}
int buddy_blindop_get_params(void* h, float* decay, float* weights, float* phases, void* stream) {
  BOP_CHECK(h); return blindop_get_params((BlindOp*)h, decay, weights, phases, (hipStream_t)stream);
}
int buddy_blindop_update_H(void* h, const float* noise, void* stream) { BOP_CHECK(h); return blindop_update_H((BlindOp*)h, noise, (hipStream_t)stream); }
int buddy_blindop_get_H(void* h, float* out, void* stream) { BOP_CHECK(h); if (!out) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_get_H((BlindOp*)h, out, (hipStream_t)stream); }
int buddy_blindop_set_y(void* h, const float* y, void* stream) { BOP_CHECK(h); if (!y) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_set_y((BlindOp*)h, y, (hipStream_t)stream); }
int buddy_blindop_degrade(void* h, const float* x, float* y, void* stream) { BOP_CHECK(h); if (!x || !y) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_degrade((BlindOp*)h, x, y, (hipStream_t)stream); }
int buddy_blindop_time_rir(void* h, float* out, void* stream) { BOP_CHECK(h); if (!out) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_time_rir((BlindOp*)h, out, (hipStream_t)stream); }
int buddy_blindop_design_filter(void* h, float* A, void* stream) { BOP_CHECK(h); if (!A) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_design_filter((BlindOp*)h, A, (hipStream_t)stream); }
int buddy_blindop_apply_stft(void* h, const float* x, float* X, void* stream) { BOP_CHECK(h); if (!x || !X) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_apply_stft((BlindOp*)h, x, X, (hipStream_t)stream); }
int buddy_blindop_degrade_vjp(void* h, const float* x, const float* g_y, float* g_x, float* g_H, void* stream) {
  BOP_CHECK(h);
  if (!g_y || (!g_x && !g_H) || (g_H && !x)) { set_error("degrade_vjp: g_y and at least one output; x is needed for g_H"); return BUDDY_ERR_ARG; }
  return blindop_degrade_vjp((BlindOp*)h, x, g_y, g_x, g_H, (hipStream_t)stream);
}
int buddy_blindop_time_rir_vjp(void* h, const float* g_rir, float* g_H, void* stream) { BOP_CHECK(h); if (!g_rir || !g_H) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_time_rir_vjp((BlindOp*)h, g_rir, g_H, (hipStream_t)stream); }
int buddy_blindop_update_H_vjp(void* h, const float* g_H, float* g_decay, float* g_weights, float* g_phases, void* stream) {
  BOP_CHECK(h); if (!g_H) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_update_H_vjp((BlindOp*)h, g_H, g_decay, g_weights, g_phases, (hipStream_t)stream);
}
int buddy_blindop_stft(void* h, const float* x, int len, float* X, void* stream) { BOP_CHECK(h); if (!x || !X) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_stft_len((BlindOp*)h, x, len, X, (hipStream_t)stream); }
int buddy_blindop_stft_adjoint(void* h, const float* G, int len, float* g_x, void* stream) { BOP_CHECK(h); if (!G || !g_x) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_stft_len_adj((BlindOp*)h, G, len, g_x, (hipStream_t)stream); }
int buddy_blindop_stft_loss(void* h, const float* a, const float* b, int len, float weight, float* loss, float* g_a, float* g_b, void* stream) {
  BOP_CHECK(h); if (!a || !b || !loss) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_stft_loss((BlindOp*)h, a, b, len, weight, loss, g_a, g_b, (hipStream_t)stream);
}
int buddy_blindop_set_compression(void* h, float comp) { BOP_CHECK(h); return blindop_set_compression((BlindOp*)h, comp); }
int buddy_blindop_set_loss_norm(void* h, int mode) { BOP_CHECK(h); return blindop_set_loss_norm((BlindOp*)h, mode); }
int buddy_blindop_lengths(void* h, int* L, int* Lr, int* T, int* Td) { BOP_CHECK(h); return blindop_lengths((BlindOp*)h, L, Lr, T, Td); }
int buddy_blindop_minphase(void* h, const float* hin, float* out, void* stream) { BOP_CHECK(h); if (!hin || !out) { set_error("null"); return BUDDY_ERR_ARG; } return blindop_minphase((BlindOp*)h, hin, out, (hipStream_t)stream); }
int buddy_blindop_project(void* h, void* stream) { BOP_CHECK(h); return blindop_project((BlindOp*)h, (hipStream_t)stream); }
int buddy_blindop_get_adam(void* h, float* m_decay, float* v_decay, float* m_weights, float* v_weights, float* m_phases, float* v_phases, int* step,
                           void* stream) {
  BOP_CHECK(h); return blindop_get_adam((BlindOp*)h, m_decay, v_decay, m_weights, v_weights, m_phases, v_phases, step, (hipStream_t)stream);
}
int buddy_blindop_rec_loss_grad(void* h, const float* x_den, float weight, float* loss, float* g_x, void* stream) {
  BOP_CHECK(h); if (!x_den || !loss) { set_error("null"); return BUDDY_ERR_ARG; }
  return blindop_rec_loss_grad((BlindOp*)h, x_den, weight, loss, g_x, (hipStream_t)stream);
}
int buddy_blindop_fir_loss_grad(void* h, const float* x_den, const float* rir, long long rir_stride, int M, float weight, float* loss, float* g_x,
                                void* stream) {
  BOP_CHECK(h); if (!x_den || !rir || !loss || M < 1) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  return blindop_fir_loss_grad((BlindOp*)h, x_den, rir, rir_stride, M, weight, loss, g_x, (hipStream_t)stream);
}
int buddy_blindop_param_grads(void* h, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, float* g_decay, float* g_weights,
                              float* g_phases, float* losses, void* stream) {
  BOP_CHECK(h); if (!x_den) { set_error("null"); return BUDDY_ERR_ARG; }
  return blindop_param_grads((BlindOp*)h, x_den, noise, t_op, w_rec, w_reg, g_decay, g_weights, g_phases, losses, (hipStream_t)stream);
}
int buddy_blindop_optimize(void* h, const float* x_den, const float* noise, float t_op, int n_iters, float w_rec, float w_reg, float lr, float beta1,
                           float beta2, float weight_decay, void* stream) {
  BOP_CHECK(h); if (!x_den || n_iters < 0) { set_error("bad arguments"); return BUDDY_ERR_ARG; }
  return blindop_optimize((BlindOp*)h, x_den, noise, t_op, n_iters, w_rec, w_reg, lr, beta1, beta2, weight_decay, (hipStream_t)stream);
}

}  // extern "C"
