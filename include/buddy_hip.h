/* C-ABI of libbuddy_hip.so -- the MI355X (gfx950) drop-in boundary for the reverse-diffusion dereverberation
 * sampler path of sp-uhh/buddy.
 *
 * The reference has no FFI boundary on this path: its plug points are Python classes resolved from Hydra
 * `_target_` strings.  This library sits *under* those classes: the Python side
 * (buddy_amd/) keeps the reference's class/constructor/method surface and calls these entry points with raw
 * device pointers (plain pointers and sizes, no torch types).  Each entry point names the reference interface it
 * replaces.  All functions return 0 on success, non-zero on error (text via buddy_last_error()); `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  All tensors are fp32, device-resident, contiguous.
 */
#ifndef BUDDY_HIP_H
#define BUDDY_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* buddy_last_error(void);
int buddy_version(void);

/* ---- score network: replaces networks/ncsnpp.py NCSNppTime (ctor :458-464, forward :498-506) and the torch.autograd
 * backward through it that EulerHeunSamplerDPS.get_likelihood_score needs (testing/EulerHeunSamplerDPS.py:61-69). ---- */

/* number of fp32 parameters for an architecture = size of the flat blob `buddy_ncsnpp_create` expects: every tensor of
 * NCSNppTime.state_dict() (names all_modules.N.*, output_layer.*; reference networks/ncsnpp.py:157-274) concatenated in
 * module-construction order (GroupNorm_0.{weight,bias}, Conv_0.{weight,bias}, Dense_0.{weight,bias}, GroupNorm_1.*,
 * Conv_1.*, [Conv_2.*]) -- buddy_amd/synth.py:module_specs is the Python mirror of that order. */
int buddy_ncsnpp_param_count(int nf, const int* ch_mult, int n_levels, int num_res_blocks, long long* count);

/* build a network handle from a HOST blob: the parameters are uploaded once; the 3x3 convolutions' operand forms (Winograd-domain weights,
 * bf16x3 stage images) are derived ON THE DEVICE, per layer, for the one kernel variant the first forward / VJP of a workload selects.
 * n_fft / hop: conf/network/ncsnpp.yaml:2-5 (510 / 128).  (The reference builds its module with `instantiate(args.network)` and loads a
 * state dict, test.py:60-75; this is the library side of NCSNppTime.load_state_dict + .to(device).) */
int buddy_ncsnpp_create(const float* host_params, long long n_params, int nf, const int* ch_mult, int n_levels,
                        int num_res_blocks, int n_fft, int hop, void** handle);
int buddy_ncsnpp_destroy(void* handle);
/* a second handle on the SAME prepared weights (reference-counted, read-only): own activation arena, VJP tape and attention / gemm / fir
 * settings (copied from `handle`).  For concurrent sub-batches on several streams; no reference counterpart (testing/tester.py:132-153 samples
 * one utterance at a time on one module). */
int buddy_ncsnpp_replica(void* handle, void** replica);
/* device bytes held by the (shared) weight store of a handle: raw parameters, bf16x3 images of the 1x1 / NIN matrices, lazily prepared 3x3
 * operand forms (and their count). */
int buddy_ncsnpp_weight_bytes(void* handle, long long* params, long long* packed, long long* lazy, int* lazy_forms);
/* the device-side weight preparation on its own: raw torch OIHW [O][I][3][3] (device) -> operand form `kind` (61: the sub-pixel up form, see buddy_gn_upconv3x3_winograd6; 0 direct [Co][9][Ci], 2 Winograd
 * F(2x2,3x3) [Ci/8][16][Co][8], 4 F(4x4,3x3) [36][Co][Ci], 6 F(6x6,3x3) [64][Co][Ci]); dgrad != 0: the data-gradient direction (Co = I, Ci = O,
 * taps flipped).  out: conv weight count x (9 | 16 | 36 | 64) floats.  Weights of ddpm_conv3x3, networks/ncsnpp_utils/layers.py:119-126. */
int buddy_conv3_weight_prep(const float* w_oihw, int O, int I, int dgrad, int kind, float* out, void* stream);

/* size the activation arena for batch B, length L samples (done implicitly by forward; exposed to report bytes). */
int buddy_ncsnpp_reserve(void* handle, int B, int L, int with_vjp, long long* bytes);

/* y = cskip[b]*x + cout[b]*net(cin[b]*x, cnoise[b])  if cin/cskip/cout are non-NULL (EDM denoiser,
 * reference diff_params/shared.py:98-120), else y = net(x, cnoise) (NCSNppTime.forward).
 * x, y: [B][L]; cnoise, cin, cskip, cout: [B].  save_for_vjp != 0 keeps activations for buddy_ncsnpp_vjp. */
int buddy_ncsnpp_forward(void* handle, const float* x, const float* cnoise, const float* cin, const float* cskip,
                         const float* cout, float* y, int B, int L, int save_for_vjp, void* stream);

/* grad_x = (d y / d x)^T cot for the last forward issued with save_for_vjp (includes the EDM scalars if they were given). */
int buddy_ncsnpp_vjp(void* handle, const float* cot, float* grad_x, void* stream);

/* debugging / per-module parity: device pointer + NHWC dims ([B][frames][bins][C]) of the output of all_modules[idx]. */
int buddy_ncsnpp_tap(void* handle, int module_idx, const float** ptr, int dims[4]);

/* device-to-device copy on `stream` (used to read taps without touching another HIP runtime handle) */
int buddy_copy_d2d(void* dst, const void* src, long long bytes, void* stream);

/* ---- measurement: time every matrix-core (igemm) launch with HIP events on its own launch stream.  collect() waits for
 * the recorded events and returns, per class (index 0: 3x3 convolutions, index 1: 1x1 convs / attention / DFT GEMMs),
 * the summed kernel time [ms], the algorithmic FLOPs (2*M*N*K, zero padding counted like torch's flop counter) and
 * the number of launches since the last collect, and the algorithmic bytes (A once + weights once + C once). ---- */
/* level: 0 off; 1 only the dominant kernel (the 36 batched GEMMs of the three-pass convolutions; collect with buddy_prof_collect_wino4, only its
 * GEMM entries are filled) -- every event pair costs a dispatch bubble of several microseconds, so a throughput measurement brackets nothing
 * else; 2 every instrumented class (attribution). */
int buddy_prof_enable(int level);
int buddy_prof_collect(double* ms /*[2]*/, double* flops /*[2]*/, long long* launches /*[2]*/, double* bytes /*[2]*/,
                       double* executed_flops /*[2]: = flops except for Winograd launches (4/9 of the direct-conv flops)*/);
/* same for the HBM-bound GroupNorm launch groups (statistics; apply + SiLU + resample; backward sums + apply): summed time [ms],
 * algorithmic bytes (each pass reads its inputs once and writes its output once) and launch-group count since the last collect. */
int buddy_prof_collect_hbm(double* ms, double* bytes, long long* launches);
/* per-pass totals of the three-pass F(4x4,3x3) convolutions since the last collect: ms[3] = {input transform, 36 batched GEMMs, output
 * transform}, executed FLOPs of the GEMM pass (2 * 36 * tiles * Cin * Cout), algorithmic bytes of the two transform passes and of the GEMM pass. */
int buddy_prof_collect_wino4(double* ms /*[3]*/, double* gemm_flops, double* bytes_in, double* bytes_out, double* bytes_gemm, long long* launches);

/* calibration: `blocks` workgroups x 4 waves issue 4*iters fp32 MFMAs (32x32x2) each on operands from seed[1024] with no
 * memory traffic; out[blocks*256] keeps the result live; clk[0] = shader clocks, clk[1] = 100 MHz wall ticks of block 0.
 * FLOPs = blocks * 4 waves * 4 * iters * 4096. */
int buddy_mfma_ubench(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, void* stream);
/* the same loop on v_mfma_f32_32x32x16_bf16 with random operand bits (12 MFMAs = 393 216 FLOP per wave and iteration): the bf16 matrix rate this box
 * sustains under its power budget, the ceiling of the bf16x3 GEMM */
int buddy_mfma_ubench_bf16(const float* seed, float* out, int blocks, int iters, unsigned long long* clk, void* stream);

/* calibration: one streaming pass over `bytes` (a multiple of 16, 16-byte aligned device buffers): mode 0 copy src -> dst (bytes read + bytes written),
 * 1 read src (summed in registers), 2 write dst; nt != 0 uses the non-temporal loads / stores.  blocks >= 1: that many workgroups of 256 threads stride
 * over the array in contiguous chunks of 8 x 256 sixteen-byte words (eight independent requests per thread); blocks = -1 | -2 | -4 | -8: the one-shot
 * form, one chunk of that many words per thread and workgroup (-1 = the classic one-float4-per-thread copy).  mode 3 / 4: READ in the access pattern of
 * the batched GEMM's A operand (rows of nt = 512 | 1024 | 2048 bytes, 64 bytes per lane and K-stage in four 16-byte loads, 64 rows per wave): all stages'
 * loads in flight (3) or one stage ahead with a stand-in for the matrix work between them (4); `blocks` is ignored.  The HBM rate a plain kernel reaches on THIS box, next to the nominal 8 TB/s (MI355X_MICROARCH.md records 6.29 TB/s for a
 * float4 copy): the second denominator of every HBM-bound roofline in bench.py. */
int buddy_hbm_ubench(const void* src, void* dst, long long bytes, int mode, int nt, int blocks, void* stream);

/* ---- unit-level kernels (the pieces the network is made of; used by the parity tests) ---- */
/* C[b] = alpha * op(A[b]) op(Bt[b])^T (+ bias_n), row-major; transX = operand stored k-major. replaces torch.einsum/bmm
 * in AttnBlockpp (networks/ncsnpp_utils/layerspp.py:82-86) and NIN (layers.py:548-557). */
int buddy_gemm(const float* A, int ldA, int transA, const float* Bt, int ldB, int transB, float* C, int ldC, int M, int N,
               int K, float alpha, const float* bias_n, int accumulate, int batch, long long strideA, long long strideB,
               long long strideC, void* stream);
/* The batched Winograd-domain products of the three-pass 3x3 convolutions on their own (benchmarks, tools/gemm36.py):
 * M[p] (tiles x Cout) = V[p] (tiles x Cin) . U[p]^T (Cout x Cin), p < positions (36 = F(4x4,3x3), 64 = F(6x6,3x3)), dense strides.
 * Same arithmetic as buddy_gemm; this entry point selects the instantiation the convolutions use (own name in profiles). */
int buddy_gemm_winograd_domain(const float* V, const float* U, float* M, int tiles, int Cout, int Cin, int positions, void* stream);
/* The same products in "bf16x3" arithmetic (csrc/wgemm.hip): every fp32 operand split exactly into three bf16 terms, six bf16 MFMA products,
 * fp32 accumulation -- the accuracy of the fp32 kernel at 2.67x fewer matrix-pipe cycles.  The weights are split once into the kernel's LDS
 * stage image: U3 = buddy_wgemm_packed_bytes(...) bytes of device memory filled by buddy_wgemm_pack_weights from U [positions][Cout][Cin]
 * (device).  Cout % 128 == 0, Cin % 32 == 0 (packed_bytes returns 0 otherwise). */
long long buddy_wgemm_packed_bytes(int positions, int Cout, int Cin);
int buddy_wgemm_pack_weights(const float* U, void* U3, int positions, int Cout, int Cin, void* stream);
int buddy_gemm_winograd_domain_bf16x3(const float* V, const void* U3, float* M, int tiles, int Cout, int Cin, int positions, void* stream);
/* The same products in "f16x2" arithmetic (csrc/wgemm.hip; network option gemm = "f16x2"): every fp32 operand scaled by a power of two and split into
 * two f16 terms (22 significant bits), three f16 MFMA products (lo*hi + hi*lo + hi*hi), fp32 accumulation -- half the matrix-pipe cycles of bf16x3 for operands
 * good to 2^-22 instead of 2^-24.  U2 = buddy_wgemm_f16x2_packed_bytes(...) bytes filled by buddy_wgemm_f16x2_pack_weights (one power of two per position;
 * positions <= 64, Cout % 128 == 0, Cin % 64 == 0, else 0).
 * The rows of V are `tiles` = utterances * tiles_per_utt (tiles_per_utt >= 32); vmax holds the abs-max of V per utterance (over all positions) as 64
 * partial maxima, float bit patterns, one per 128-byte line: unsigned [utterances][64][32], word 0 of each line used (buddy_abs_max_bits fills it; the
 * three-pass convolutions collect it inside their input transform with one atomic max per tile): the power of two of V's scale is derived from the exponent
 * field of the largest, so an utterance's result does not depend on the rest of the batch. */
long long buddy_wgemm_f16x2_packed_bytes(int positions, int Cout, int Cin);
int buddy_wgemm_f16x2_pack_weights(const float* U, void* U2, int positions, int Cout, int Cin, void* stream);
int buddy_gemm_winograd_domain_f16x2(const float* V, const void* U2, float* M, int tiles, int Cout, int Cin, int positions, const unsigned* vmax,
                                     int tiles_per_utt, void* stream);
/* out [segments][64][32]: partial maxima (bit patterns, word 0 of each 128-byte line) of |x| over segment u of every one of `groups` equally spaced blocks:
 * x is [groups][segments][seg_len] floats (V: groups = positions, segments = utterances, seg_len = tiles_per_utt * Cin); out is overwritten. */
int buddy_abs_max_bits(const float* x, int groups, int segments, long long seg_len, unsigned* out, void* stream);
/* The 1x1 convolutions / NIN layers (layers.py:100-106, 548-557) on the same kernel: C (M x N, row stride ldC) = alpha * [A0 | A1] W^T + bias_n
 * (+ C if accumulate); the K input channels come from A0 (first C0, row stride ldA0) and, if A1 != NULL, A1 (the rest, ldA1) -- the U-Net's
 * channel concatenation is never materialised.  W3 = buddy_wgemm_pack_weights(W [N][K], ., 1, N, K).  N % 128, K % 32, C0 % 32 == 0. */
int buddy_gemm_bf16x3(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W3, float* C, int ldC, long long M, int N, int K,
                      const float* bias_n, float alpha, int accumulate, void* stream);
/* A ResBlock's input gradient in one GEMM launch (layerspp.py:242-274 backward): dx = alpha * A W^T + the input-gradient of act(GroupNorm(cat[x0, x1]))
 * for the incoming da -- the skip path's 1x1 data-gradient (Conv_2^T; A = the block's output gradient, M = B * HW rows, K channels; W3 =
 * buddy_wgemm_pack_weights(W [N][K], ., 1, N, K)) with the GroupNorm backward's apply pass as the GEMM's epilogue: neither the 1x1 result nor a separate
 * apply pass reaches HBM.  x / dx: channel-concatenated views split at C0 (x1 = dx1 = NULL: one tensor); accK != 0: that destination accumulates.
 * stats: forward (mean, rstd) [B][G][2]; red [B][G][2] out; stat_scratch >= B*256*N*16 bytes.  N % 128, K % 32, C0 % 128 == 0. */
int buddy_gemm_bf16x3_gn_bwd(const float* A, int ldA, const void* W3, const float* x0, const float* x1, int C0, const float* da, const float* stats,
                             const float* gamma, const float* beta, int G, int silu, float alpha, float* dx0, float* dx1, int acc0, int acc1,
                             double* stat_scratch, float* red, int B, int HW, int N, int K, void* stream);
/* The two general forms above in f16x2 arithmetic (round 6; what a handle with gemm = f16x2 runs where option gen_f16x2 says so): W2 =
 * buddy_wgemm_f16x2_pack_weights(W, ., 1, N, K) (one power of two for the matrix); the A operand's power of two is taken PER ROW inside the kernel,
 * found on the way along K (a stage that leaves the current scale's range rescales the row's accumulators by an exact power of two: no pre-pass), so a
 * row's result depends on nothing but the row.  K % 64 == 0; everything else as the bf16x3 entries.  Options gen_f16x2 (1: the network's 1x1 / NIN /
 * skip-path launches use these; 0: the exact bf16x3 split), gen_rows (0 by size | 32 | 64 rows per wave), gen_cp (two column blocks per workgroup). */
int buddy_gemm_f16x2(const float* A0, int ldA0, const float* A1, int ldA1, int C0, const void* W2, float* C, int ldC, long long M, int N, int K,
                     const float* bias_n, float alpha, int accumulate, void* stream);
int buddy_gemm_f16x2_gn_bwd(const float* A, int ldA, const void* W2, const float* x0, const float* x1, int C0, const float* da, const float* stats,
                            const float* gamma, const float* beta, int G, int silu, float alpha, float* dx0, float* dx1, int acc0, int acc1,
                            double* stat_scratch, float* red, int B, int HW, int N, int K, void* stream);
/* NHWC 3x3 stride-1 pad-1 conv, packed weights wt[Cout][9*Cin] (tap-major, channel-minor); replaces ddpm_conv3x3
 * (networks/ncsnpp_utils/layers.py:119-126). */
int buddy_conv3x3(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                  void* stream);
/* the same convolution through the fused Winograd F(2x2,3x3) kernel (4*Cin instead of 9*Cin MACs per output, fp32):
 * transform_weights (host -> host): wt[Cout][9*Cin] -> U[Cin/8][16 positions][Cout][8]; conv takes U on the device. */
int buddy_winograd_transform_weights(const float* wt_host, int Cout, int Cin, float* U_host);
int buddy_conv3x3_winograd(const float* x, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                           void* stream);
/* F(4x4,3x3) three-pass variant (input transform, 36 batched GEMMs, output transform) used for the large layers: U4[36][Cout][Cin];
 * scratch: 36 * (B*H*W/16) * (Cin + Cout) floats. */
int buddy_winograd4_transform_weights(const float* wt_host, int Cout, int Cin, float* U4_host);
int buddy_conv3x3_winograd4(const float* x, const float* U4, const float* bias, float* y, float* scratch, int B, int H, int W, int Cin, int Cout,
                            void* stream);
/* F(6x6,3x3) three-pass variant for the large layers (64 positions, 8x8 input patch per 6x6 output tile: 1.78 instead of 2.25 multiply-adds per
 * output, any H, W >= 6 -- tiles may overhang): U6[64][Cout][Cin]; scratch: 64 * B * ceil(H/6) * ceil(W/6) * (Cin + Cout) floats. */
int buddy_winograd6_transform_weights(const float* wt_host, int Cout, int Cin, float* U6_host);
int buddy_conv3x3_winograd6(const float* x, const float* U6, const float* bias, float* y, float* scratch, int B, int H, int W, int Cin, int Cout,
                            void* stream);
/* act(GroupNorm(cat[x0, x1])) -> conv3x3 as ONE three-pass convolution: the normalisation and SiLU are applied inside the input transform
 * (the activated tensor never reaches HBM) and, with csum != NULL, the output transform leaves the per-(utterance, channel) sum and sum of
 * squares of y (csum[B][Cout][2], float64) -- the statistics the NEXT GroupNorm needs (layerspp.py:243-245, 257-259).  x1 may be NULL (single
 * source; C0 ignored).  stats: [B][G][2] out.  scratch as for buddy_conv3x3_winograd4; stat_scratch: >= B*256*1024*16 bytes. */
int buddy_gn_conv3x3_winograd4(const float* x0, const float* x1, int C0, const float* gamma, const float* beta, int G, int silu, const float* U4,
                               const float* bias, float* y, float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W,
                               int Cin, int Cout, void* stream);
/* the same through the F(6x6,3x3) passes (scratch as for buddy_conv3x3_winograd6) */
int buddy_gn_conv3x3_winograd6(const float* x0, const float* x1, int C0, const float* gamma, const float* beta, int G, int silu, const float* U6,
                               const float* bias, float* y, float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W,
                               int Cin, int Cout, void* stream);
/* The data-gradient side of the same fusion: da = conv3x3(g) (U6 = the transposed / flipped weights in the F(6x6,3x3) domain) is the gradient
 * w.r.t. act(GroupNorm(cat[x0, x1])) (Cout channels); the output transform also leaves that GroupNorm's per-(utterance, channel) backward sums
 * chsum[B][Cout][2] = (sum dxhat, sum dxhat * xhat), dxhat = da * act'(z) * gamma, float64 -- what buddy_groupnorm_act_bwd otherwise obtains
 * with a reduction pass over (x, da).  stats: the forward (mean, rstd) [B][G][2]. */
int buddy_conv3x3_winograd6_gn_bwd_sums(const float* g, const float* U6, float* da, float* scratch, const float* x0, const float* x1, int C0,
                                        const float* stats, const float* gamma, const float* beta, int G, int silu, double* stat_scratch,
                                        double* chsum, int B, int H, int W, int Cin, int Cout, void* stream);
/* ... and the input side of the backward: y = conv3x3(dx), dx = the input-gradient of act(GroupNorm(x)) for the incoming gradient da, with the
 * GroupNorm backward's apply pass evaluated inside the F(6x6,3x3) input transform (dx never reaches HBM).  stats: forward (mean, rstd) [B][G][2];
 * red: [B][G][2] out (the two per-group backward means); stat_scratch: >= B*256*C*16 bytes. */
int buddy_gnbwd_conv3x3_winograd6(const float* x, const float* gamma, const float* beta, const float* stats, const float* da, int G, int silu,
                                  const float* U6, float* y, float* scratch, double* stat_scratch, float* red, int B, int H, int W, int C, int Cout,
                                  void* stream);
/* The up ResBlock's first convolution, Conv_0(naive_upsample_2d(act(GroupNorm_0(x)))) (layerspp.py:243-257, up_or_down_sampling.py:172-176), as ONE
 * three-pass F(6x6,3x3) convolution on the LOW-resolution grid ("sub-pixel" form): x (B, H, W, Cin) -> y (B, 2H, 2W, Cout).  Output pixel
 * (2i + py, 2j + px) reads a 2x2 subset of the upsampled taps, so each of the four phases is a 3x3 convolution of x with summed taps; U6up
 * [64][4 Cout][Cin] = buddy_conv3_weight_prep(kind 61, dgrad 0).  Neither the activated nor the upsampled tensor reaches HBM.  csum (optional):
 * per-(utterance, channel) (sum, sum of squares) of y, float64.  scratch: 64 * B * ceil(H/6) * ceil(W/6) * (Cin + 4 Cout) floats; stats [B][G][2]
 * out; stat_scratch >= B*256*1024*16 bytes. */
int buddy_gn_upconv3x3_winograd6(const float* x, const float* gamma, const float* beta, int G, int silu, const float* U6up, const float* bias, float* y,
                                 float* scratch, float* stats, double* stat_scratch, double* csum, int B, int H, int W, int Cin, int Cout, void* stream);
/* Its data-gradient with the GroupNorm_1 backward in front (the up block's backward between Conv_1 and Conv_0): y (B, H, W, Cout) = the gradient
 * w.r.t. the LOW-resolution input of conv3x3(upsample(.)) for the high-resolution gradient dx, dx = input-gradient of act(GroupNorm(h)) for the
 * incoming da; h, da: (B, 2H, 2W, C), read space-to-depth inside the input transform.  U6upT [64][Cout][4 C] = buddy_conv3_weight_prep(kind 61,
 * dgrad 1) of the OIHW tensor [C][Cout][3][3].  stats: forward (mean, rstd) of h [B][G][2]; red [B][G][2] out; scratch: 64 * tiles * (4 C + Cout)
 * floats; stat_scratch >= B*256*C*16 bytes. */
int buddy_gnbwd_upconv3x3_winograd6(const float* h, const float* gamma, const float* beta, const float* stats, const float* da, int G, int silu,
                                    const float* U6upT, float* y, float* scratch, double* stat_scratch, float* red, int B, int H, int W, int C,
                                    int Cout, void* stream);
/* GroupNorm(G, C, eps=1e-6) [+SiLU] [+2x down(mode 1)/up(mode 2)] forward; replaces nn.GroupNorm + nn.SiLU +
 * naive_{up,down}sample_2d (layerspp.py:243-258). stats: [B][G][2] out; scratch: >= B*256*C*16 bytes. */
int buddy_groupnorm_act(const float* x, const float* gamma, const float* beta, float* y, float* stats, void* scratch, int B,
                        int H, int W, int C, int G, int mode, int silu, void* stream);
/* input-gradient of the above given dy (at the resampled resolution). red: [B][G][2] scratch. */
int buddy_groupnorm_act_bwd(const float* x, const float* gamma, const float* beta, const float* stats, const float* dy, float* dx,
                            void* scratch, float* red, int B, int H, int W, int C, int G, int mode, int silu, void* stream);

/* fir=True resampling (networks/ncsnpp_utils/up_or_down_sampling.py:195-257 upsample_2d / downsample_2d -> upfirdn2d, op/upfirdn2d_kernel.cu)
 * with the (1,3,3,1) kernel, factor 2, NHWC: up != 0: (H,W) -> (2H,2W), else (H,W) -> (H/2,W/2); y = (accumulate ? y : 0) + scale * resample(x).
 * The transposes are the other direction times 4 resp. 1/4.  buddy_ncsnpp_set_fir switches a network handle to this resampling (no parameters). */
int buddy_fir_resample2(const float* x, float* y, int B, int H, int W, int C, int up, float scale, int accumulate, void* stream);
int buddy_ncsnpp_set_fir(void* handle, int fir);
/* Per-handle launcher options -- "no hidden global state" (SURVEY.md 8(b)): every switch a launcher consults (attention core, GEMM arithmetic, the
 * fusion / layout A/B switches) is a field of the handle's option struct; two handles in one process may differ.  Keys (csrc/options.hip): conv, gemm,
 * attention, gn_fuse, gn_fuse_bwdin, gn_fuse_bwd, upconv, c2_fuse, attn_tr, attn_split, attn_nw, igemm_epi, igemm_variant, wgemm_gen_epi, wgemm_xcdpos,
 * wgemm_epi, wgemm_rt, wgemm_nt, gen_f16x2, gen_rows, gen_cp, gnb_nt, wino_epi, wino_abl, wino_geo, w6_xcd, w6_nt, gn_fast, gn_trips, ew_grid, c2in4, c2out_tiled, fir_lds, op_graph.  An unknown key or a value out of range is
 * BUDDY_ERR_ARG.  A handle starts from the process defaults = the BUDDY_<KEY> environment variables, parsed and validated in ONE place at handle
 * creation: a bad value, or an unknown BUDDY_* name within edit distance 2 of a switch (a misspelling), makes buddy_ncsnpp_create fail with a message naming it;
 * BUDDY_* names that resemble no switch are not this library's and are left alone.  Set options before the first forward or between calls: the
 * activation arena is sized again and a saved forward is dropped (buddy_ncsnpp_vjp returns BUDDY_ERR_STATE until the next forward with save = 1). */
int buddy_options_check(void);   /* the environment check alone (no GPU needed): BUDDY_OK, or BUDDY_ERR_ARG with buddy_last_error() naming the variable */
int buddy_option_validate(const char* key, int value);   /* would buddy_ncsnpp_set_option accept (key, value)?  no handle, no GPU, nothing changed */
int buddy_ncsnpp_set_option(void* handle, const char* key, int value);
int buddy_ncsnpp_get_option(void* handle, const char* key, int* value);
/* attention core of a network handle: 4 = auto (default; fp32: the materialised T x T form while T <= 4096, the online-softmax kernels beyond -- a
 * function of T alone); 0 = online-softmax kernels, fp32 operands; 1 / 2 = the same with bf16 / f16 MFMA operands (opt-in fast mode, fp32 accumulate
 * + fp32 softmax); 3 = always the materialised T x T matrix.  Initial value from BUDDY_ATTN = auto | flash | bf16 | f16 | matrix. */
int buddy_ncsnpp_set_attention(void* handle, int mode);
/* Arithmetic of the Winograd-domain GEMMs of the 3x3 convolutions (94 % of the FLOPs):
 *   2 (default) = "f16x2": both fp32 operands scaled by a power of two and split by round-to-nearest into TWO f16 terms (22 significant bits instead of
 *       fp32's 24), three f16 MFMA products, fp32 accumulation; measured against float64 it is as close as the exact forms (one denoiser evaluation +
 *       VJP 117.2 / 113.0 dB), but it is NOT bit-for-bit the fp32 product;
 *   1 = "bf16x3": every fp32 operand split EXACTLY into three bf16 terms, six bf16 MFMA products, fp32 accumulation: the fp32 kernel's accuracy;
 *   0 = v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chains; the reference run).
 * The 1x1 / NIN / DFT GEMMs run in bf16x3 in modes 1 and 2.  Env BUDDY_GEMM=fp32|bf16x3|f16x2 sets the process default. */
int buddy_ncsnpp_set_gemm(void* handle, int mode);

/* single-head attention over T tokens without the T x T matrix (online softmax, fp32 MFMA), token-major q, k, v, O [B][T][C], C in {64,128,256}:
 * O = softmax(scale * q k^T) v, lse [B][T] = row log-sum-exp; replaces the einsum / softmax / einsum of AttnBlockpp.forward
 * (networks/ncsnpp_utils/layerspp.py:82-86).  bwd: gradients of the same three steps given dO (delta [B][T] is scratch).
 * prec must be 0 (fp32 operands, the reference arithmetic); the 16-bit-operand kernels take a workspace and are the next three entries. */
int buddy_flash_attention_fwd(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int prec, void* stream);
int buddy_flash_attention_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta,
                              float* dq, float* dk, float* dv, int B, int T, int C, float scale, int prec, void* stream);
/* The same three steps with 16-bit MFMA operands (prec 1 = bf16, 2 = f16; v_mfma_f32_32x32x16_*), fp32 accumulation and fp32 softmax statistics --
 * the opt-in fast mode of buddy_ncsnpp_set_attention(handle, 1 | 2) (BASELINE configs[4] "fp16 MFMA attention path"; not the reference's arithmetic).
 * A pre-pass converts q (times scale log2 e), k, v, dO once into 16-bit operand arrays in `ws` (buddy_flash_attention16_workspace floats: token-major
 * rows and channel-major transposes, T padded to 128); the kernels stream them through LDS by DMA and keep the softmax in registers (csrc/attn16.hip). */
long long buddy_flash_attention16_workspace(int B, int T, int C);
int buddy_flash_attention16_fwd(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int prec,
                                float* ws, void* stream);
int buddy_flash_attention16_bwd(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta,
                                float* dq, float* dk, float* dv, int B, int T, int C, float scale, int prec, float* ws, void* stream);
/* The same fp32 kernels with their sequential loop (keys in fwd / dq, queries in dk / dv) split over `splits` workgroups per row block and the partial
 * results combined in fixed split order (no atomics): what the network uses when T / 64 workgroups per utterance would leave most of the 256 CUs idle
 * -- ONE utterance at a time is the reference's own shape (testing/tester.py:132-153).  buddy_flash_attention_splits = the count the network picks: a
 * function of T alone (B is ignored), so that a row's result does not depend on the batch it is computed in,
 * buddy_flash_attention_workspace = floats of `ws` for a count (splits <= 0: for the picked one); every split must own at least one 32-row block. */
int buddy_flash_attention_splits(int B, int T);
long long buddy_flash_attention_workspace(int B, int T, int C, int splits);
int buddy_flash_attention_fwd_split(const float* q, const float* k, const float* v, float* O, float* lse, int B, int T, int C, float scale, int splits,
                                    float* ws, void* stream);
int buddy_flash_attention_bwd_split(const float* q, const float* k, const float* v, const float* O, const float* dO, const float* lse, float* delta,
                                    float* dq, float* dk, float* dv, int B, int T, int C, float scale, int splits, float* ws, void* stream);

/* ---- sampler elementwise / reductions (replace the tensor expressions of testing/EulerHeunSampler.py:41-72,
 * testing/EulerHeunSamplerDPS.py:61-69,115-157, diff_params/edm.py:83-96), per-utterance (row) semantics ---- */
/* out[b][i] = a[b]*x[b][i] + c[b]*y[b][i] (y may be NULL) */
int buddy_axpby_rows(const float* x, const float* y, const float* a, const float* c, float* out, int B, int L, void* stream);
/* stochastic churn: out = x + scale * eps (EulerHeunSampler.py:41-45, scale = sqrt(t_hat^2 - t^2)) */
int buddy_perturb(const float* x, const float* eps, float scale, float* out, long long n, void* stream);
/* fused Euler / Heun update of the DPS sampler (EulerHeunSamplerDPS.py:128-157, edm.py:83-96), per-utterance rows:
 *   x_den' = x_den * den_scale[b] (NULL = 1);  d = -t * (x_den' - x_hat) / t^2 + lh_scale[b] (NULL = 1) * lh (NULL = 0);
 *   out = base + dt * (w_prev * d_prev (NULL = 0) + w_cur * d);   d_out / x_den_out optional.
 * lh_scale (round 6): the guidance normaliser zeta / (||grad||_2 / sqrt(audio_len) + 1e-8) per utterance (:66-69, buddy_row_scale mode 1), so the raw
 * likelihood gradient goes in and no scaled copy of it is materialised. */
int buddy_dps_update(const float* x_hat, const float* x_den, const float* lh, const float* lh_scale, const float* den_scale, const float* base,
                     const float* d_prev, float t, float dt, float w_prev, float w_cur, float* out, float* d_out, float* x_den_out, int B, int L, void* stream);
/* one scalar per utterance row of x (B, L), one launch, fp64 accumulation: mode 0: out[b] = p0 / std(x_b) (unbiased, Tensor.std(): the speech-magnitude
 * constraint, :127-129); mode 1: out[b] = p0 / (||x_b||_2 / p1 + 1e-8) (the guidance normaliser with p0 = zeta, p1 = sqrt(audio_len)) */
int buddy_row_scale(const float* x, float* out, int B, int L, int mode, float p0, float p1, void* stream);
/* out[k][b] = v_k, k < 4, b < B: the four EDM preconditioning scalars of one sigma (evaluated on the host in fp32) broadcast over the batch */
int buddy_fill_rows4(float* out, int B, float v0, float v1, float v2, float v3, void* stream);
/* per-row sum and sum of squares in double precision: out[b] = {sum, sumsq} */
int buddy_row_moments(const float* x, double* out, int B, int L, void* stream);
/* time-domain FIR (direct form) y[b][n] = sum_m h_b[m] x[b][n-m], n < L, h_b = h + b*h_stride (h_stride 0 = shared RIR):
 * replaces utils/reverb_utils.py:25-61 fast_apply_RIR (same linear convolution, no FFT); adjoint != 0 computes the
 * correlation (its transpose) for the VJP. */
int buddy_fir(const float* x, const float* h, long long h_stride, float* y, int B, int L, int M, int adjoint, void* stream);

/* ---- WPE warm start (`wpe_scaled`, testing/EulerHeunSamplerDPS.py:32-54 -> nara_wpe.wpe.wpe(Y, taps, delay, iterations,
 * statistics_mode='full'), single channel): Y, X are (rows, T) complex128 as interleaved doubles, one row per (utterance, frequency bin);
 * scratch: rows*T doubles.  Per row and iteration: inverse power, correlation matrix/vector, Cholesky solve, prediction filter. ---- */
int buddy_wpe(const double* Y, double* X, double* scratch, int rows, int T, int taps, int delay, int iterations, void* stream);
/* The whole warm-start estimate of testing/EulerHeunSamplerDPS.py:36-49 in one call: y (B, L) float32 -> out (B, L) float32 =
 * istft(wpe(stft(y)))[..., :L] with nara_wpe.utils.stft / istft conventions (size 512, shift 128, periodic Blackman window, fading,
 * bi-orthogonal synthesis window), every utterance on its own, complex128 inside.  workspace: buddy_wpe_workspace_bytes(B, L) bytes of
 * device memory.  The rescaling to scaling_factor / std (:51) and the noise (:52) stay with the caller. */
long long buddy_wpe_workspace_bytes(int B, int L);
int buddy_wpe_dereverb(const float* y, float* out, void* workspace, int B, int L, int taps, int delay, int iterations, void* stream);

/* ---- blind subband-filtering reverb operator, batched over U utterances; replaces testing/operators/subband_filtering.py
 * (BlindSubbandFiltering :142-351 incl. SubbandFiltering :8-136), utils/reverb_utils.py:3-23, utils/losses.py:59-64 and the
 * torch.optim.Adam loop of EulerHeunSamplerDPS.optimize_op (testing/EulerHeunSamplerDPS.py:71-113) with hand-written
 * forward + analytic backward kernels.  STFT is fixed to NFFT 1024 / win 512 / hop 128 (op_hp of the conf/tester yaml files).
 * Tensor shapes follow the reference: decay/weights (U,E,bands), phases (U,513,Nf), H (U,513,Nf) complex64 interleaved. ---- */
int buddy_blindop_create(int U, int L, int Nf, int E, int num_knots, const float* knots_hz /*host*/, int sample_rate, float compression,
                         float min_decay, float max_decay, float w_lo, float w_hi, int clamp_decay, int long_second, void** handle);
int buddy_blindop_destroy(void* handle);
int buddy_blindop_set_params(void* handle, const float* decay, const float* weights, const float* phases, int reset_adam, void* stream);
int buddy_blindop_get_params(void* handle, float* decay, float* weights, float* phases, void* stream);
/* update_H (:253-285): H = cons(A exp(j phases)); noise != NULL (U, 128*Nf samples) = use_noise=True (phases := angle(H)). */
int buddy_blindop_update_H(void* handle, const float* noise, void* stream);
int buddy_blindop_get_H(void* handle, float* H_out, void* stream);
int buddy_blindop_set_y(void* handle, const float* y /*(U,L)*/, void* stream);            /* caches comp(STFT(y)) for the losses */
int buddy_blindop_degrade(void* handle, const float* x, float* y, void* stream);           /* degradation (:82-101) with the current H */
int buddy_blindop_time_rir(void* handle, float* rir /*(U, 128*Nf+1024)*/, void* stream);   /* get_time_RIR (:103-113) */
/* per-function views (what optimize_op / the likelihood run internally), used by the parity tests against reference fixtures:
 * design_filter (subband_filtering.py:241-251) -> A (U,513,Nf); apply_stft (:41-52) of x (U,L) -> (U,513,T,2), T = 1+(L+512)/128;
 * minimum_phase_version (utils/reverb_utils.py:9-23) of h (U, 128*(Nf+1)); project_params (:298-331) on the current decay/weights;
 * Adam moments (torch.optim.Adam exp_avg / exp_avg_sq; any pointer may be NULL) and the step count. */
int buddy_blindop_design_filter(void* handle, float* A, void* stream);
int buddy_blindop_apply_stft(void* handle, const float* x, float* X, void* stream);
int buddy_blindop_minphase(void* handle, const float* h, float* out, void* stream);
/* ---- the differentiable surface (round 6): vector-Jacobian products of the pieces the REFERENCE's own sampler autograds through -- operator.degradation
 * in get_likelihood_score (testing/EulerHeunSamplerDPS.py:61-69), update_H / degradation / get_time_RIR in optimize_op (:71-113), apply_stft inside
 * get_loss(...)(y, y_hat) (utils/losses.py:26-71).  buddy_amd/testing/operators wraps each as a torch.autograd.Function, so an unmodified torch-loop
 * sampler (oracle/sampler_ref.py = the reference's control flow) runs on the HIP operator.  H-shaped gradients are (U, 513, Nf, 2) = (d/dRe, d/dIm). */
/* degradation^T for the CURRENT H: g_x (U, L) and / or g_H from g_y (U, L); x (U, L) is needed for g_H only; either output may be NULL (not both) */
int buddy_blindop_degrade_vjp(void* handle, const float* x, const float* g_y, float* g_x, float* g_H, void* stream);
int buddy_blindop_time_rir_vjp(void* handle, const float* g_rir /*(U, 128*Nf+1024)*/, float* g_H, void* stream);      /* get_time_RIR^T */
/* update_H^T (H = cons(design_filter(decay, weights) exp(j phases)), subband_filtering.py:253-285): gradients of the three parameter tensors in the
 * reference layouts from g_H, using the state the LAST buddy_blindop_update_H left in the handle; any output may be NULL */
int buddy_blindop_update_H_vjp(void* handle, const float* g_H, float* g_decay, float* g_weights, float* g_phases, void* stream);
/* apply_stft (:41-52) of signals of the bound length L (T = 1 + (L + 512) / 128 frames) or of the time-RIR length 128 * Nf + 1024, X (U, 513, frames, 2),
 * and its adjoint g_x (U, len) from G (U, 513, frames, 2); other lengths are BUDDY_ERR_ARG */
int buddy_blindop_stft(void* handle, const float* x, int len, float* X, void* stream);
int buddy_blindop_stft_adjoint(void* handle, const float* G, int len, float* g_x, void* stream);
/* loss[u] = weight * l2_comp_stft_summean(a_u, b_u) (utils/losses.py:59-64) for two signals of length len (as above) with the gradient w.r.t. either
 * argument (g_a / g_b (U, len), NULL = not wanted) */
int buddy_blindop_stft_loss(void* handle, const float* a, const float* b, int len, float weight, float* loss, float* g_a, float* g_b, void* stream);
/* compression exponent of the spectral losses, (0, 1] (the handle is created with one; the shipped configs use 0.667); call buddy_blindop_set_y again
 * afterwards: the cached compressed observation depends on it */
int buddy_blindop_set_compression(void* handle, float compression_factor);
/* which member of the reference's compressed-spectrum family every loss entry of the handle evaluates (utils/losses.py:46-64):
 * 0 = l2_comp_stft_summean (default, the shipped configs), 1 = l2_comp_stft_sum, 2 = l2_comp_stft_mean */
int buddy_blindop_set_loss_norm(void* handle, int mode);
int buddy_blindop_lengths(void* handle, int* L, int* L_rir, int* frames, int* frames_rir);
int buddy_blindop_project(void* handle, void* stream);
int buddy_blindop_get_adam(void* handle, float* m_decay, float* v_decay, float* m_weights, float* v_weights, float* m_phases, float* v_phases,
                           int* step /*host*/, void* stream);
/* loss[u] = weight * l2_comp_stft_summean(y_u, degrade(x_den_u)); g_x = d sum_u loss / d x_den (NULL to skip) */
int buddy_blindop_rec_loss_grad(void* handle, const float* x_den, float weight, float* loss, float* g_x, void* stream);
/* informed counterpart (RIROperator, testing/operators/reverb.py:33-35 + utils/losses.py:59-64): degradation = time-domain FIR with the
 * known RIR(s) rir[u * rir_stride + m], m < M (fast_apply_RIR semantics: first L output samples); same loss, same cached y (set_y). */
int buddy_blindop_fir_loss_grad(void* handle, const float* x_den, const float* rir, long long rir_stride, int M, float weight, float* loss,
                                float* g_x, void* stream);
/* one gradient evaluation of optimize_op's objective (H rebuilt from the parameters first); losses = [rec (U), reg (U)] */
int buddy_blindop_param_grads(void* handle, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, float* g_decay,
                              float* g_weights, float* g_phases, float* losses, void* stream);
/* n_iters iterations of optimize_op: update_H, rec + RIR-noise losses, backward, Adam on [decay, weights, phases], projection.
 * noise: (n_iters, U, 128*Nf+1024) N(0,1) draws (NULL = no regulariser) */
int buddy_blindop_optimize(void* handle, const float* x_den, const float* noise, float t_op, int n_iters, float w_rec, float w_reg, float lr,
                           float beta1, float beta2, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
