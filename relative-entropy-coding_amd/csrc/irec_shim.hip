// irec_shim.hip -- the elementwise hand-offs between the convolutions of the RVAE host shim and the coder, one launch each.
//
// Reference (file:line): BidirectionalResidualBlock.call, rec/models/resnet_vae.py:372-497 -- what lies BETWEEN its
// convolutions on the compression path: the ELU in front of every convolution (:385,:400,:488), the posterior =
// inference-side + generative-side statistics and the exp of the log-scales (:409-413, :464-469, :148-154), the concat of
// the deterministic features with the coded latent (:479-487), the residual update `input + 0.1 * tensor` (:492-496).
// As stock PyTorch ops these are ~10 launches of 3-5 us per residual block and pass -- on a single 32x32 image as much
// device time as the 96 convolutions (profiles/r02i/single_image_kernel_stats.csv).  Here: three kernels.
// Encoder and decoder run the SAME kernels, so the prior statistics the decoder rebuilds have the encoder's bits.
// Layouts: activations NCHW float32 contiguous (what the convolutions produce); the coder's statistics and the latent NHWC
// (the reference's layout, which fixes the flattening order Coder.split shuffles: coder.py:56-67).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "irec_kernels.h"

namespace irec {

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }   // tf.nn.elu, alpha = 1

// out[k][n][hw][c] (k < n_stats, c < s):
//   k = 0: y[n][c][hw]                                   prior loc                 (:409-411)
//   k = 1: exp(y[n][s + c][hw])                          prior scale               (:412-413)
//   k = 2: y[n][2s + c][hw] + inf[n][c][hw]              posterior loc             (:148-150, :464-466)
//   k = 3: exp(y[n][3s + c][hw] + inf[n][s + c][hw])     posterior scale           (:151-154, :467-469)
// y has Cy channels, inf (the inference pass's heads, nullptr for n_stats = 2) Ci channels.  by / bi (may be null): the
// biases of the convolutions that produced y / inf, added here -- in the order PyTorch adds them, right after the
// convolution -- instead of by a launch of their own per convolution (96 per image).
__global__ __launch_bounds__(256) void shim_stats_kernel(const float *__restrict__ y, const float *__restrict__ inf, float *__restrict__ out,
                                                         int n_stats, int N, int Cy, int Ci, int s, int HW,
                                                         const float *__restrict__ by, const float *__restrict__ bi) {
  const int64_t per_k = (int64_t)N * HW * s, total = per_k * n_stats;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e / per_k);
    const int64_t r = e - (int64_t)k * per_k;
    const int c = (int)(r % s);
    const int64_t nhw = r / s;
    const int hw = (int)(nhw % HW), n = (int)(nhw / HW);
    float v = y[((int64_t)n * Cy + k * s + c) * HW + hw];
    if (by) v = v + by[k * s + c];
    if (k >= 2) {
      float w = inf[((int64_t)n * Ci + (k - 2) * s + c) * HW + hw];
      if (bi) w = w + bi[(k - 2) * s + c];
      v = v + w;
    }
    out[e] = (k & 1) ? expf(v) : v;
  }
}

// out[n][c][hw], c < d + s:  elu(y[n][c_off + c][hw]) for c < d, elu(latent[n][hw][c - d]) beyond (latent NHWC; s = 0: none)
// -- tf.concat([tensor, latent_code], axis=-1) then tf.nn.elu (:479-488), or the ELU of a channel slice (:398-400).
__global__ __launch_bounds__(256) void shim_cat_elu_kernel(const float *__restrict__ y, const float *__restrict__ latent, float *__restrict__ out,
                                                           int N, int Cy, int c_off, int d, int s, int HW, const float *__restrict__ by) {
  const int C = d + s;
  const int64_t total = (int64_t)N * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int hw = (int)(e % HW);
    const int64_t nc = e / HW;
    const int c = (int)(nc % C), n = (int)(nc / C);
    float v;
    if (c < d) { v = y[((int64_t)n * Cy + c_off + c) * HW + hw]; if (by) v = v + by[c_off + c]; }
    else v = latent[((int64_t)n * HW + hw) * s + (c - d)];
    out[e] = elu1(v);
  }
}

// out = inp + alpha * t (:492-496) and out_elu = elu(out): the next block's first op (:385), or last_gen_conv's input.
// (bt: bias of the convolution that produced t, per channel of the [N][C][HW] tensors; null = none)
__global__ __launch_bounds__(256) void shim_residual_elu_kernel(const float *__restrict__ inp, const float *__restrict__ t, float alpha,
                                                                float *__restrict__ out, float *__restrict__ out_elu, int64_t count,
                                                                const float *__restrict__ bt, int C, int HW) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 256) {
    float tv = t[e];
    if (bt) tv = tv + bt[(e / HW) % C];
    const float v = inp[e] + alpha * tv;
    out[e] = v;
    out_elu[e] = elu1(v);
  }
}

static int shim_grid(int64_t count) { const int64_t g = (count + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

hipError_t launch_shim_stats(const float *y, const float *inf, float *out, int n_stats, int N, int Cy, int Ci, int s, int HW,
                             const float *by, const float *bi, hipStream_t st) {
  hipLaunchKernelGGL(shim_stats_kernel, dim3(shim_grid((int64_t)n_stats * N * HW * s)), dim3(256), 0, st, y, inf, out, n_stats, N, Cy, Ci, s, HW, by, bi);
  return hipGetLastError();
}
hipError_t launch_shim_cat_elu(const float *y, const float *latent, float *out, int N, int Cy, int c_off, int d, int s, int HW,
                               const float *by, hipStream_t st) {
  hipLaunchKernelGGL(shim_cat_elu_kernel, dim3(shim_grid((int64_t)N * (d + s) * HW)), dim3(256), 0, st, y, latent, out, N, Cy, c_off, d, s, HW, by);
  return hipGetLastError();
}
hipError_t launch_shim_residual_elu(const float *inp, const float *t, float alpha, float *out, float *out_elu, int64_t count,
                                    const float *bt, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(shim_residual_elu_kernel, dim3(shim_grid(count)), dim3(256), 0, st, inp, t, alpha, out, out_elu, count, bt, C, HW);
  return hipGetLastError();
}

} // namespace irec
