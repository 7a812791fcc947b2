// Internal C++ interface between the C-ABI (capi.hip) and the network graph (net.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace buddy {

struct NetCfg { int nf; int ch_mult[8]; int nlev; int nrb; int n_fft; int hop; };
struct Net;

void set_error(const std::string& s);
const char* last_error();
long long param_count(const NetCfg& c);
int net_create(const float* host_params, long long n, const NetCfg& cfg, Net** out);
void net_destroy(Net* N);
int net_replica(Net* src, Net** out);      // a second handle sharing src's prepared weights (own arena / tape / settings)
int net_weight_bytes(Net* N, long long* params, long long* packed, long long* lazy, int* lazy_forms);   // device bytes of the shared weight store
int net_set_option(Net* N, const char* key, int value);   // per-handle launcher option (keys: csrc/options.hip); unknown key / bad value -> BUDDY_ERR_ARG
int net_get_option(Net* N, const char* key, int* value);
int net_set_gemm(Net* N, int mode);        // Winograd-domain GEMM arithmetic: 2 f16x2 (default), 1 bf16x3 exact split, 0 fp32 MFMA
int net_set_attention(Net* N, int mode);   // 0 flash fp32 (default), 1 bf16 / 2 f16 MFMA operands, 3 materialised T x T
int net_set_fir(Net* N, int fir);     // fir=True resampling (reference up_or_down_sampling.py:195-257), set before the first forward
int net_reserve(Net* N, int B, int L, int with_vjp, long long* bytes);
int net_forward(Net* N, const float* x, const float* cnoise, const float* cin_b, const float* cskip_b, const float* cout_b, float* y, int B, int L,
                int save, hipStream_t st);
int net_vjp(Net* N, const float* cot, float* gx, hipStream_t st);
int net_get_tap(Net* N, int module_idx, const float** p, int dims[4]);


// ---- blind subband-filtering operator (operator.hip) ----
struct BlindOpCfg {
  int n_fft, win, hop, Nf, E, num_knots, sample_rate;
  float knots[64];
  float comp;                       // compression exponent of the spectral loss
  float min_decay, max_decay, w_lo, w_hi;
  int clamp_decay, long2nd;
};
struct BlindOp;
int blindop_create(const BlindOpCfg& cfg, int U, int L, BlindOp** out);
void blindop_destroy(BlindOp* o);
int blindop_set_params(BlindOp* o, const float* decay, const float* wts, const float* phases_ref, int reset_adam, hipStream_t st);
int blindop_get_params(BlindOp* o, float* decay, float* wts, float* phases_ref, hipStream_t st);
int blindop_update_H(BlindOp* o, const float* noise, hipStream_t st);
int blindop_get_H(BlindOp* o, float* out, hipStream_t st);
int blindop_set_y(BlindOp* o, const float* y, hipStream_t st);
int blindop_degrade(BlindOp* o, const float* x, float* y, hipStream_t st);
int blindop_time_rir(BlindOp* o, float* out, hipStream_t st);
int blindop_design_filter(BlindOp* o, float* A_ref, hipStream_t st);
int blindop_apply_stft(BlindOp* o, const float* x, float* X_ref, hipStream_t st);
int blindop_minphase(BlindOp* o, const float* h, float* out, hipStream_t st);
int blindop_project(BlindOp* o, hipStream_t st);
int blindop_get_adam(BlindOp* o, float* m_decay, float* v_decay, float* m_wts, float* v_wts, float* m_phases, float* v_phases, int* step, hipStream_t st);
int blindop_degrade_vjp(BlindOp* o, const float* x, const float* g_y, float* g_x, float* g_H_ref, hipStream_t st);
int blindop_time_rir_vjp(BlindOp* o, const float* g_rir, float* g_H_ref, hipStream_t st);
int blindop_update_H_vjp(BlindOp* o, const float* g_H_ref, float* g_decay, float* g_wts, float* g_phases_ref, hipStream_t st);
int blindop_stft_len(BlindOp* o, const float* x, int len, float* X_ref, hipStream_t st);
int blindop_stft_len_adj(BlindOp* o, const float* G_ref, int len, float* g_x, hipStream_t st);
int blindop_stft_loss(BlindOp* o, const float* a, const float* b, int len, float weight, float* loss, float* g_a, float* g_b, hipStream_t st);
int blindop_set_compression(BlindOp* o, float comp);
int blindop_set_loss_norm(BlindOp* o, int mode);
int blindop_lengths(BlindOp* o, int* L, int* Lr, int* T, int* Td);
int blindop_rec_loss_grad(BlindOp* o, const float* x_den, float weight, float* loss, float* g_x, hipStream_t st);
int blindop_fir_loss_grad(BlindOp* o, const float* x_den, const float* rir, long long rir_stride, int M, float weight, float* loss, float* g_x,
                          hipStream_t st);
int blindop_param_grads(BlindOp* o, const float* x_den, const float* noise, float t_op, float w_rec, float w_reg, float* g_decay, float* g_wts,
                        float* g_phases_ref, float* losses, hipStream_t st);
int blindop_optimize(BlindOp* o, const float* x_den, const float* noise, float t_op, int n_iters, float w_rec, float w_reg, float lr, float b1,
                     float b2, float wd, hipStream_t st);

}  // namespace buddy
