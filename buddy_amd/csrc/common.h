// Shared declarations for the gfx950 (MI355X, CDNA4) kernels of the BUDDy sampler path.
// Layout convention: activations are fp32 "NHWC" [B][H][W][C] with H = STFT frames (time) and W = frequency
// bins -- i.e. the reference's (B,C,F,T) tensors with the two spatial axes swapped and channels innermost, so
// the STFT GEMM writes the network input directly and every conv tap is a contiguous C-vector.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define BUDDY_OK 0
#define BUDDY_ERR_HIP 1
#define BUDDY_ERR_ARG 2
#define BUDDY_ERR_STATE 3

namespace buddy {

// ---- launcher options (options.hip): per handle, defaults from the validated BUDDY_* environment ----------------------
struct Options {
  int conv;            // 3x3 convolution form: 0 by shape (F(6x6) / F(4x4) three-pass, fused F(2x2), direct), 1 direct, 2 wino2, 3 wino4
  int gemm;            // Winograd-domain GEMM arithmetic: 2 f16x2 (default: two-term f16 splits, 2^-22), 1 bf16x3 (exact split), 0 fp32 MFMA
  int attn;            // attention core 0 .. 4 (net.hip)
  int gn_fuse, gn_fuse_bwdin, gn_fuse_bwd, upconv, c2_fuse, attn_tr;                 // fusions of the network graph (A/B switches, default 1)
  int attn_split, attn_nw;                                                           // fp32 attention: forced loop-split count / forward tile height (0 = by shape)
  int igemm_epi, igemm_variant, wgemm_gen_epi, wgemm_xcdpos, wgemm_epi, wgemm_rt, wgemm_nt, gen_f16x2, gen_rows, gen_cp, gnb_nt;              // GEMM kernels
  int wino_epi, wino_abl, wino_geo, w6_xcd, w6_nt;                                          // Winograd kernels
  int gn_fast, gn_trips, ew_grid, c2in4, c2out_tiled;                                                   // GroupNorm / 2-channel convolutions
  int fir_lds, op_graph;                                                             // blind operator
};
const Options& default_options();           // process defaults (environment)
int options_check();                        // BUDDY_ERR_ARG (+ set_error) when the environment holds an unknown BUDDY_* name or a bad value
const Options& cur_opt();                   // options of the handle whose call runs on this thread (the defaults outside one)
struct OptScope { const Options* prev; explicit OptScope(const Options* o); ~OptScope(); OptScope(const OptScope&) = delete; };
int option_set(Options& o, const char* key, int value);
int option_get(const Options& o, const char* key, int* value);
const char* prof_dump_path();               // BUDDY_PROF_DUMP

// ---- implicit-GEMM (conv3x3 / conv1x1 / plain GEMM) -------------------------------------------------
// C[m][n] = out_scale * ( alpha * sum_k A[m][k] * Bt[n][k] + bias_n[n] + bias_m[m] + bias_bn[b(m)][n] + res[...] )
struct IgemmParams {
  // A operand. TAPS==9: A is an NHWC image, M = B*H*W, K per tap = Cin, optionally split over two sources
  // along channels (channel c < C0 from A0, else from A1) -- the torch.cat([h, skip]) of the U-Net up path.
  const float* A0; const float* A1; int C0; int ldA0; int ldA1;
  int Cin;                 // K per tap (total over both sources); multiple of 4
  int H, W;                // spatial dims (TAPS==9), also used to derive batch index for bias_bn / res_up
  int M, N;                // GEMM M (rows/pixels per batch slice) and N
  const float* Bt; int ldB;  // weights [N][TAPS*Cin] (k contiguous) or, if TRANS_B, [K][N] with ldB
  float* C; int ldC;
  long long sA, sB, sC;    // batch strides (blockIdx.z), floats
  const float* bias_n; const float* bias_m; const float* bias_bn; int ld_bias_bn; int rows_per_batch;
  const float* res; int ldRes; int res_mode;   // 0 none, 1 same pixel, 2 nearest-upsampled source (H/2 x W/2)
  float alpha, out_scale; int accumulate;
  int wide_epi;            // set by launch_igemm: full-width aligned tile -> LDS-staged float4 epilogue
  int tag;                 // 36: the batched GEMM of a F(4x4,3x3) convolution (own kernel instantiation, profiling only)
};
void launch_igemm(const IgemmParams& p, int taps, bool transA, bool transB, int batch, hipStream_t st);
// bf16x3 form of the Winograd-domain batched GEMM (wgemm.hip): weights pre-split into the LDS stage image, V / M dense fp32
bool wgemm_supported(int Cout, int Cin);
size_t wgemm_packed_bytes(int P, int Cout, int Cin);
void wgemm_pack_weights(const float* U_dev, void* U3_dev, int P, int Cout, int Cin, hipStream_t st);
void launch_wgemm_bf16x3(const float* V, const void* U3, float* M, long long Mt, int Cout, int Cin, int P, hipStream_t st);
bool wgemm_f16x2_supported(int Cout, int Cin);
size_t wgemm_f16x2_packed_bytes(int P, int Cout, int Cin);
void wgemm_f16x2_pack_weights(const float* U_dev, void* U2_dev, int P, int Cout, int Cin, hipStream_t st);
// f16x2 GEMM: the abs-max of V per utterance is kept as VMAX_SUB partial maxima (float bit patterns), one per 128-byte line: [utterance][VMAX_SUB][VMAX_STRIDE]
constexpr int VMAX_SUB = 64, VMAX_STRIDE = 32;
void launch_abs_max_bits(const float* x, int groups, int segments, long long seg_len, unsigned* out, hipStream_t st);
void launch_wgemm_f16x2(const float* V, const void* U2, float* M, long long Mt, int Cout, int Cin, int P, const unsigned* vmax, int tiles_per_utt, hipStream_t st);
bool wgemm_general_supported(int N, int K, int C0, int ldA0, int ldA1, int ldC, const void* A0, const void* A1, const void* C, const void* bias);
void launch_wgemm_bf16x3_general(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W3, float* C, int ldC, long long M, int N, int K,
                                 const float* bias_n, float alpha, int accumulate, hipStream_t st);
bool wgemm_f16x2_general_supported(int N, int K, int C0, int ldA0, int ldA1, int ldC, const void* A0, const void* A1, const void* C, const void* bias);
void launch_wgemm_f16x2_general(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W2, float* C, int ldC, long long M, int N, int K,
                                const float* bias_n, float alpha, int accumulate, hipStream_t st);
// Winograd F(2x2,3x3) variant of the 3x3 conv (wino.hip): same IgemmParams, pre-transformed weights U[Cin/16][16][Cout][16]
bool wino_supported(const IgemmParams& p);
void launch_wino(const IgemmParams& p, const float* Uw, hipStream_t st);
void wino_transform_weights(const float* wt_host, int Cout, int Cin, float* U_host);
struct Src2 { const float* p0; const float* p1; int C0; int ld0; int ld1; };   // channel-concatenated input view
struct Dst2 { float* p0; float* p1; int C0; int ld0; int ld1; int acc0; int acc1; };
// the skip path's 1x1 data-gradient GEMM with the GroupNorm_0 backward's apply pass as its epilogue (wgemm.hip, GNB): d (two-destination channel view,
// accumulating where its acc flag says) = alpha * A (M x K) W^T (N x K, bf16x3 image) + rstd * (dxhat - m1 - xhat * m2) of (x, da); stats / red [B][G][2]
bool wgemm_gnbwd_supported(int N, int K, int ldA, const Src2& x, const Dst2& d, const void* A, const void* da);
void launch_wgemm_bf16x3_gnbwd(const float* A, int ldA, const void* W3, long long M, int N, int K, float alpha, Src2 x, const float* da, const float* stats,
                               const float* red, const float* gamma, const float* beta, int G, int silu, int HW, Dst2 d, hipStream_t st);
void launch_wgemm_f16x2_gnbwd(const float* A, int ldA, const void* W2, long long M, int N, int K, float alpha, Src2 x, const float* da, const float* stats,
                              const float* red, const float* gamma, const float* beta, int G, int silu, int HW, Dst2 d, hipStream_t st);
// Winograd F(4x4,3x3) in three passes (wino4.hip): weights U4[36][Cout][Cin], scratch V (36*M/16*Cin floats) and Mb (36*M/16*N floats)
// Fusions with the GroupNorms either side of the convolution (both optional):
//   gn   -- the input is act(GroupNorm(gn->x)) of a same-resolution (concatenated) view, applied inside the input transform (p.A0 unused)
//   stat -- the output transform also writes per-(utterance, channel) partial (sum, sum of squares), wino4_stat_chunks(p) per utterance
// with da != nullptr the input is instead the GroupNorm BACKWARD, d/dx of act(GroupNorm(x)) applied to the incoming gradient da (same layout as
// x's channels, row stride ldda) with the two per-group backward means `red` -- F(6x6,3x3) input transform only
struct W4Gn { Src2 x; const float* stats; const float* gamma; const float* beta; int G; int silu;
              const float* da = nullptr; int ldda = 0; const float* red = nullptr; };
bool wino4_supported(const IgemmParams& p);
void wino4_scratch(const IgemmParams& p, long long* v_floats, long long* m_floats);
int wino4_stat_chunks(const IgemmParams& p);
void launch_wino4(const IgemmParams& p, const float* U4, float* V, float* Mb, hipStream_t st, const W4Gn* gn = nullptr, double* stat = nullptr,
                  const void* U4x = nullptr);   // U4x: the same weights in the bf16x3 stage image (wgemm.hip) -> the GEMM pass runs in bf16x3
void wino4_transform_weights(const float* wt_host, int Cout, int Cin, float* U4_host);
// Winograd F(6x6,3x3) in three passes (wino6.hip): weights U6[64][Cout][Cin], 64 * tiles * (Cin + N) floats of scratch, any H, W >= 6 (tiles
// overhang); same fusions.  wino6_pays: executed work incl. the overhang is at least 10 % below F(4x4,3x3)'s (the large layers).
bool wino6_supported(const IgemmParams& p);
bool wino6_pays(const IgemmParams& p);
void wino6_scratch(const IgemmParams& p, long long* v_floats, long long* m_floats, int up = 0);
int wino6_stat_chunks(const IgemmParams& p, int up = 0);
double wino6_exec_ratio(const IgemmParams& p, int up = 0);
//   bwd_gn (with stat) -- data-gradient convolutions: the output is the gradient w.r.t. act(GroupNorm(bwd_gn->x)); the partials are the two sums
//           of that GroupNorm's backward, (dxhat, dxhat * xhat), instead of (sum, sum of squares)
//   U6x, xform, vmax -- the GEMM pass on the stage image U6x: xform 1 bf16x3, 2 f16x2 (vmax: B zeroed slots for the abs-max of V per utterance)
//   up -- sub-pixel forms of conv3x3(nearest-upsample x2) (1) and of its data-gradient (2); p describes the LOW resolution (wino6.hip)
void launch_wino6(const IgemmParams& p, const float* U6, float* V, float* Mb, hipStream_t st, const W4Gn* gn = nullptr, double* stat = nullptr,
                  const W4Gn* bwd_gn = nullptr, const void* U6x = nullptr, int up = 0, int xform = 1, unsigned* vmax = nullptr);
void wino6_transform_weights(const float* wt_host, int Cout, int Cin, float* U6_host);
// device-side weight preparation (wprep.hip): raw torch OIHW [O][I][3][3] -> the operand form of one kernel variant (kind 0 direct [Co][9][Ci],
// 2 F(2x2) [Ci/8][16][Co][8], 4 F(4x4) [36][Co][Ci], 6 F(6x6) [64][Co][Ci]) for the forward (Co = O, Ci = I) or the data-gradient direction
// (Co = I, Ci = O, taps flipped); the wino*_transform_weights host functions above are its restatement for the unit tests
long long conv3_weight_floats(int O, int I, int kind);
int launch_conv3_weight_prep(const float* w_oihw, int O, int I, bool dgrad, int kind, float* out, hipStream_t st);
void igemm_prof_record(const IgemmParams& p, int taps, int batch, hipStream_t st, bool begin, double exec_ratio = 4.0 / 9.0);
void igemm_prof_enable(int level);   // 0 off, 1 the dominant kernel only (36 batched Winograd-domain GEMMs), 2 every instrumented class
bool igemm_prof_enabled();           // level 2
int igemm_prof_level();
void prof_w4_push(hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, hipEvent_t e3, double gemm_flops, double bytes_in, double bytes_out, double bytes_gemm);
int prof_w4_collect(double ms[3], double* gemm_flops, double* bytes_in, double* bytes_out, double* bytes_gemm, long long* launches);
void prof_hbm_begin(double algorithmic_bytes, hipStream_t st);   // bracket of an HBM-bound launch group (GroupNorm kernels)
void prof_hbm_end(hipStream_t st);
int prof_hbm_collect(double* ms, double* bytes, long long* launches);
int igemm_prof_collect(double ms[2], double flops[2], long long launches[2], double bytes[2], double exec_flops[2]);

// ---- small-channel direct convs -----------------------------------------------------------------------
// Cin == 2 -> Cout (first conv, Combine 1x1, dgrad of the 2-channel pyramid heads)
void launch_conv_c2in(const float* x, const float* w /*[Cout][taps][2]*/, const float* bias, const float* add, int add_ld,
                      float* y, int ldY, int B, int H, int W, int Cout, int taps, int accumulate, hipStream_t st);
// Cin -> 2 (pyramid heads, dgrad of first conv / Combine). up_add: previous pyramid level at (H/2,W/2) added nearest-up.
void launch_conv_c2out(const float* x, int ldX, const float* w /*[taps][Cin][2]*/, const float* bias, const float* up_add,
                       float* y, int B, int H, int W, int Cin, int taps, int accumulate, hipStream_t st);

// ---- GroupNorm (+SiLU, + 2x resample) -------------------------------------------------------------------
int  gn_num_chunks(int HW);
void launch_gn_stats(Src2 x, int B, int HW, int C, int G, float eps, double* partial, float* stats /*[B][G][2]*/, hipStream_t st);
// per-(utterance, channel) sums kept with a tensor (csum[b][c][2] = sum, sum of squares, fp64) so that every GroupNorm reading it -- alone or
// as one half of a skip concatenation -- gets its statistics without another pass over the tensor:
//   launch_chan_sums      one pass over a single-source tensor (for tensors whose producer does not leave partials)
//   launch_csum_collapse  partial[b][chunks][C][2] (a producer's epilogue, wino4 stat) -> csum
//   launch_gn_stats_csum  (mean, rstd) per group from the csums of the one or two sources of a view
void launch_chan_sums(const float* x, int B, int HW, int C, double* partial, double* csum, hipStream_t st);
void launch_csum_collapse(const double* partial, int chunks, int B, int C, double* csum, hipStream_t st);
void launch_gn_stats_partial(const double* partial, int chunks, int B, int HW, int C, int G, float eps, float* stats, hipStream_t st);
void launch_gn_stats_csum(const double* csum0, const double* csum1, int C0, int B, int HW, int C, int G, float eps, float* stats, hipStream_t st);
// mode: 0 same, 1 down (2x2 mean of the activated tensor; pooled_raw gets the 2x2 mean of x itself), 2 up (nearest x2)
void launch_gn_apply(Src2 x, const float* stats, const float* gamma, const float* beta, int B, int H, int W, int C, int G,
                     int mode, int silu, float* out, float* pooled_raw, hipStream_t st);
// backward wrt x. da: gradient of the (resampled) activated output. extra: additional gradient added to dx
// (extra_mode 0 none, 1 same index, 2 quarter of a pooled-resolution tensor), scaled by extra_scale.
void launch_gn_bwd_sums(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C, int G,
                        int mode, int silu, double* partial, float* red /*[B][G][2]*/, hipStream_t st, int ready_chunks = 0);
void launch_gn_bwd_apply(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C, int G,
                         int mode, int silu, const float* extra, int extra_mode, float extra_scale, const float* red, Dst2 dx, hipStream_t st);
// ready_chunks > 0: the backward-sum partials [B][ready_chunks][C][2] are already in `partial` (left by the producing convolution's epilogue):
// no reduction pass
void launch_gn_bwd(Src2 x, const float* stats, const float* gamma, const float* beta, const float* da, int B, int H, int W, int C,
                   int G, int mode, int silu, const float* extra, int extra_mode, float extra_scale, double* partial,
                   float* red /*[B][G][2]*/, Dst2 dx, hipStream_t st, int ready_chunks = 0);

// ---- misc elementwise -------------------------------------------------------------------------------------
// WPE warm start (wpe.hip): rows x T complex128 in/out, scratch rows*T doubles
void launch_wpe(const double* Y, double* X, double* inv_scratch, int rows, int T, int taps, int delay, int iters, hipStream_t st);
size_t wpe_workspace_bytes(int B, int L);
int wpe_frames(int L);
void launch_wpe_dereverb(const float* y, float* out, void* work, int B, int L, int taps, int delay, int iters, hipStream_t st);
void launch_axpy(float* dst, const float* src, float alpha, long long n, int accumulate, hipStream_t st);
// fir=True resampling with the (1,3,3,1) kernel: (H,W)->(2H,2W) / (H,W)->(H/2,W/2); adjoints: up^T = 4 down, down^T = up / 4
void launch_fir_up2(const float* x, float* y, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st);
void launch_fir_down2(const float* x, float* y, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st);
void launch_pool2(const float* src, float* dst, int B, int H, int W, int C, float scale, int accumulate, hipStream_t st); // (H,W)->(H/2,W/2), sum*scale
void launch_up2_acc(const float* src, float* dst, int B, int Hs, int Ws, int C, float scale, int accumulate, hipStream_t st); // (Hs,Ws)->(2Hs,2Ws)
bool flash_attn_supported(int C);
int flash_attn_splits(int B, int T);                                 // loop splits the launchers want for this grid (1 = none)
long long flash_attn_ws_floats(int B, int T, int C, int splits);     // workspace floats of the split forms (0 for splits <= 1)
void launch_flash_attn_fwd(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, float* ws, int splits,
                           hipStream_t st);
void launch_flash_attn_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq,
                           float* dk, float* dv, int B, int T, int C, float scale, float* ws, int splits, hipStream_t st);
// 16-bit-operand attention (attn16.hip; prec 1 = bf16, 2 = f16): ws = flash_attn16_ws_floats floats (operand arrays of the pre-pass)
long long flash_attn16_ws_floats(int B, int T, int C);
void launch_flash_attn16_fwd(const float* q, const float* k, const float* v, float* O, float* Lse, int B, int T, int C, float scale, int prec, float* ws,
                             hipStream_t st);
void launch_flash_attn16_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* Lse, float* D, float* dq,
                             float* dk, float* dv, int B, int T, int C, float scale, int prec, float* ws, hipStream_t st);
void launch_softmax_rows(float* S, int rows, int cols, hipStream_t st);
void launch_softmax_bwd_rows(const float* P, float* dP /*in: dP, out: dS*/, int rows, int cols, hipStream_t st);
void launch_linear(const float* x, const float* W, const float* b, float* y, int B, int K, int N, int silu_in, hipStream_t st);
void launch_fourier(const float* cnoise, const float* Wf, float* out, int B, int nf, hipStream_t st);
void launch_mix2(const float* x, const float* w /*[2][2]*/, const float* b, float* y, long long npix, int transpose, int accumulate, hipStream_t st);

// ---- STFT / iSTFT glue ------------------------------------------------------------------------------------
void launch_reflect_pad(const float* x, float* xp, int B, int L, int pad, int Lp, float scale, const float* scale_b, hipStream_t st);
void launch_ola(const float* frames, int ldF, int Tp, int n_fft, int hop, const float* inv_env, float* y, int B, int L,
                int pad, const float* xin, const float* cskip_b, const float* cout_b, hipStream_t st);
void launch_ola_adj(const float* g, int B, int L, int pad, int Tp, int n_fft, int hop, const float* inv_env, const float* cout_b,
                    float* frames, int ldF, hipStream_t st);
void launch_unpad_adj(const float* dframes, int ldF, int T, int n_fft, int hop, int B, int L, int pad, float scale,
                      const float* scale_b, const float* g_out, const float* cskip_b, float* dx, hipStream_t st);

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace buddy
