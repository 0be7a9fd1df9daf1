// K10: RMSprop step of the PatchGAN parameters as ONE launch (the discriminator step of
// model/nerf_adapt_st_gan.py:129-171 ends with optim_disc.step(), torch.optim.RMSprop with the reference's settings:
// alpha 0.99, eps 1e-8, no momentum, not centred, no weight decay, options/nerf_lm_adapt_gan.yaml:100-104).
// torch's capturable foreach implementation issues ~13 multi-tensor launches plus two per-parameter launches on 0-dim
// tensors when the learning rate lives in device memory (which a hipGraph-captured step needs).  Arithmetic in torch's
// order:  sq = alpha * sq + (1 - alpha) * g * g;   p = p - lr * g / (sqrt(sq) + eps).
#include "tp_common.h"

namespace {
struct Table {
  float* p[TP_RMSPROP_MAX_TENSORS];
  const float* g[TP_RMSPROP_MAX_TENSORS];
  float* sq[TP_RMSPROP_MAX_TENSORS];
  float* step[TP_RMSPROP_MAX_TENSORS];       // optional step counters (torch state "step"): +1 per applied step
  int64_t end[TP_RMSPROP_MAX_TENSORS];       // exclusive prefix end of each tensor in the concatenated index space
  int n;
};

__global__ __launch_bounds__(256) void rmsprop_kernel(Table t, const float* lr_dev, float lr_host, float alpha, float one_minus_alpha,
                                                      float eps, int64_t total, const int* gate, int n_gate) {
  for (int k = 0; k < n_gate; ++k)
    if (gate[k] != 0) return;                       // a flagged step: parameters and statistics stay bit-for-bit as they are
  const float lr = lr_dev != nullptr ? *lr_dev : lr_host;
  if (blockIdx.x == 0 && (int)threadIdx.x < t.n && t.step[threadIdx.x] != nullptr) {       // (RMSprop never READS its step counters)
    bool first = true;
    for (int j = 0; j < (int)threadIdx.x; ++j) first = first && t.step[j] != t.step[threadIdx.x];
    if (first) t.step[threadIdx.x][0] += 1.0f;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int k = 0;
    while (e >= t.end[k]) ++k;
    const int64_t i = e - (k == 0 ? 0 : t.end[k - 1]);
    const float g = t.g[k][i];
    float sq = t.sq[k][i];
    sq = __fadd_rn(__fmul_rn(sq, alpha), __fmul_rn(__fmul_rn(one_minus_alpha, g), g));     // mul_(alpha).addcmul_(g, g, 1 - alpha)
    t.sq[k][i] = sq;
    const float avg = __fadd_rn(sqrtf(sq), eps);
    t.p[k][i] = __fadd_rn(t.p[k][i], __fmul_rn(-lr, __fdiv_rn(g, avg)));                // addcdiv_(g, avg, value = -lr)
  }
}
}  // namespace

extern "C" int tp_rmsprop_step(const tp_rmsprop_tensor* tensors, int n, const float* lr_dev, double lr_host, double alpha, double eps,
                               const int32_t* gate, int n_gate, tp_stream_t stream) {
  TP_REQUIRE(n_gate == 0 || gate != nullptr, "gate words missing");
  TP_REQUIRE(tensors != nullptr && n > 0 && n <= TP_RMSPROP_MAX_TENSORS, "bad tensor table");
  Table t;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    TP_REQUIRE(tensors[k].param && tensors[k].grad && tensors[k].square_avg && tensors[k].numel > 0, "null tensor");
    t.p[k] = tensors[k].param; t.g[k] = tensors[k].grad; t.sq[k] = tensors[k].square_avg; t.step[k] = tensors[k].step;
    total += tensors[k].numel;
    t.end[k] = total;
  }
  for (int k = n; k < TP_RMSPROP_MAX_TENSORS; ++k) { t.p[k] = nullptr; t.g[k] = nullptr; t.sq[k] = nullptr; t.step[k] = nullptr; t.end[k] = total; }
  t.n = n;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, lr_dev, (float)lr_host, (float)alpha,
                     (float)(1.0 - alpha), (float)eps, total, gate, n_gate);
  return tp::check_launch("tp_rmsprop_step");
}
