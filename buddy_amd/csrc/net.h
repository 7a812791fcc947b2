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
int net_reserve(Net* N, int B, int L, int with_vjp, long long* bytes);
int net_forward(Net* N, const float* x, const float* cnoise, const float* cin_b, const float* cskip_b, const float* cout_b, float* y, int B, int L,
                int save, hipStream_t st);
int net_vjp(Net* N, const float* cot, float* gx, hipStream_t st);
int net_get_tap(Net* N, int module_idx, const float** p, int dims[4]);

}  // namespace buddy
