// Optimizer step over ONE flat fp32 buffer (all parameters of the model are views into it): gradient-norm clipping,
// SGD with Nesterov momentum + per-element weight decay, EMA of the weights - three launches instead of ~600 small ATen
// kernels per step (reference: yolo/engine/trainer.py:462-470 optimizer_step, :633-664 build_optimizer groups,
// yolo/utils/torch_utils.py:335-367 ModelEMA).  The same flat gradient buffer is what the data-parallel all-reduce sends.
#include "common.h"

#define OPT_BLOCK 256
__global__ __launch_bounds__(OPT_BLOCK) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  double acc = 0.0;
  for (long i = blockIdx.x * (long)OPT_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * OPT_BLOCK) { double v = g[i]; acc += v * v; }
  __shared__ double red[OPT_BLOCK];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = OPT_BLOCK / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* partial, int nb, float max_norm, float* out) {
  __shared__ double red[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) t += partial[i];          // fixed order: thread t owns partials t, t+256, ...
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const double s = red[0];
    float norm = (float)sqrt(s);
    out[0] = norm;
    float c = max_norm / (norm + 1e-6f);
    out[1] = c < 1.f ? c : 1.f;
  }
}

extern "C" size_t mgdt_grad_norm_workspace_bytes(void) { return 1024 * sizeof(double); }
extern "C" int mgdt_grad_clip_coef(const float* g, long n, float max_norm, float* out2, void* ws, mgdt_stream s) {
  if (!g || !out2 || !ws || n <= 0) MGDT_FAIL(MGDT_BAD_ARG, "grad_clip_coef: null/empty argument");
  int nb = (int)std::min<long>((n + OPT_BLOCK - 1) / OPT_BLOCK, 1024);
  sumsq_partial_kernel<<<nb, OPT_BLOCK, 0, (hipStream_t)s>>>(g, n, (double*)ws);
  clip_coef_kernel<<<1, 256, 0, (hipStream_t)s>>>((const double*)ws, nb, max_norm, out2);
  MGDT_CHECK_LAUNCH("grad_clip_coef");
  return MGDT_OK;
}

// g' = clip*g + wd[i]*p ; buf = first ? g' : momentum*buf + g' ; step = nesterov ? g' + momentum*buf : buf ; p -= lr*step
// wd[i] < 0 marks the reference's bias group (trainer.py:644): no decay and its own learning rate `lr_bias` (warm-up, trainer.py:323)
__global__ void sgd_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, const float* __restrict__ wd, long n,
                                float lr, float lr_bias, float momentum, int nesterov, int first, const float* __restrict__ clip) {
  const float c = clip ? clip[1] : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float w = wd ? wd[i] : 0.f;
    float gi = c * g[i] + (w > 0.f ? w * p[i] : 0.f);
    float b = first ? gi : momentum * buf[i] + gi;
    buf[i] = b;
    p[i] -= (w < 0.f ? lr_bias : lr) * (nesterov ? gi + momentum * b : b);
  }
}
extern "C" int mgdt_sgd_step(float* p, const float* g, float* buf, const float* wd, long n, float lr, float lr_bias, float momentum, int nesterov,
                             int first, const float* clip2, mgdt_stream s) {
  if (!p || !g || !buf || n <= 0) MGDT_FAIL(MGDT_BAD_ARG, "sgd_step: null/empty argument");
  int nb = (int)std::min<long>((n + 255) / 256, 8192);
  sgd_flat_kernel<<<nb, 256, 0, (hipStream_t)s>>>(p, g, buf, wd, n, lr, lr_bias, momentum, nesterov, first, clip2);
  MGDT_CHECK_LAUNCH("sgd_step");
  return MGDT_OK;
}

__global__ void ema_flat_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float d) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ema[i] = d * ema[i] + (1.f - d) * p[i];   // v *= d; v += (1 - d) * w (torch_utils.py ModelEMA.update); this file is built with -ffp-contract=off
}
extern "C" int mgdt_ema_update(float* ema, const float* p, long n, float decay, mgdt_stream s) {
  if (!ema || !p || n <= 0) MGDT_FAIL(MGDT_BAD_ARG, "ema_update: null/empty argument");
  int nb = (int)std::min<long>((n + 255) / 256, 8192);
  ema_flat_kernel<<<nb, 256, 0, (hipStream_t)s>>>(ema, p, n, decay);
  MGDT_CHECK_LAUNCH("ema_update");
  return MGDT_OK;
}

// SGD + EMA in one launch with the step's scalars read from device memory: hyper = {lr, lr_bias, momentum, ema_decay}.  A training step
// captured in a hipGraph replays with values the host writes between replays (warm-up interpolation trainer.py:317-326, the EMA ramp
// torch_utils.py:342).  Elements [0, n_param) are parameters (SGD, then EMA of the new value); [n_param, n_total) are buffers (EMA only).
__global__ void sgd_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, const float* __restrict__ wd, long n_param,
                                   float* __restrict__ ema, long n_total, const float* __restrict__ hyper, int nesterov, int first,
                                   const float* __restrict__ clip) {
  const float lr = hyper[0], lr_bias = hyper[1], momentum = hyper[2], d = hyper[3];
  const float c = clip ? clip[1] : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_total; i += (long)gridDim.x * blockDim.x) {
    float v = p[i];
    if (i < n_param) {
      const float w = wd ? wd[i] : 0.f;
      const float gi = c * g[i] + (w > 0.f ? w * v : 0.f);
      const float b = first ? gi : momentum * buf[i] + gi;
      buf[i] = b;
      v -= (w < 0.f ? lr_bias : lr) * (nesterov ? gi + momentum * b : b);
      p[i] = v;
    }
    if (ema) ema[i] = d * ema[i] + (1.f - d) * v;
  }
}
extern "C" int mgdt_sgd_ema_step_dev(float* p, const float* g, float* buf, const float* wd, long n_param, float* ema, long n_total, const float* hyper4,
                                     int nesterov, int first, const float* clip2, mgdt_stream s) {
  if (!p || !g || !buf || !hyper4 || n_param <= 0 || n_total < n_param) MGDT_FAIL(MGDT_BAD_ARG, "sgd_ema_step_dev: null/empty argument");
  int nb = (int)std::min<long>((n_total + 255) / 256, 8192);
  sgd_ema_dev_kernel<<<nb, 256, 0, (hipStream_t)s>>>(p, g, buf, wd, n_param, ema, n_total, hyper4, nesterov, first, clip2);
  MGDT_CHECK_LAUNCH("sgd_ema_step_dev");
  return MGDT_OK;
}
