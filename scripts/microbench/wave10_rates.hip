// wave10_rates.hip -- microbenchmark (diagnostics only; nothing here is part of the product): the scoring loop of a
// would-be "one wave per block, ten beams" encoder: lane l owns dims 256 g + 4 l .. + 3 of all four dim
// groups, G of ten beams x 16 dims in registers (160 VGPRs), 8 waves per CU at 256 VGPRs, three quantile-table copies in
// LDS, proposal rows (uint16 with copy bits) streamed from an L2-resident table.  Per sample and dim-group pair: 80
// look-ups, 20 accumulators (10 beams x 2 groups, beams paired for v_pk_fma_f32), one reduce_scatter_20.  Prints look-ups
// per clock per CU, to be set against the team encoder's 11.5-11.9 and the gather roofline's 13.4 at 8 waves per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -I../../relative-entropy-coding_amd/csrc -I../../include -o wave10_rates wave10_rates.hip && ./wave10_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TAB = 10006;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int NWV>
__global__ __launch_bounds__(NWV * 64, 1) void wave10_kernel(const float *__restrict__ tab_g, const uint16_t *__restrict__ rows, int n_rows,
                                                             const uint32_t *__restrict__ beta, float *out, int n_samples) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *t = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int k = tid; k < 3 * TAB; k += NWV * 64) t[k] = tab_g[k % TAB];
  __syncthreads();
  uint32_t bet[10];
#pragma unroll
  for (int b = 0; b < 10; ++b) bet[b] = __builtin_amdgcn_readfirstlane(beta[b]) * 4u;
  // G of 10 beams x 16 dims (pairs of beams share a register pair), H of 16 dims
  f2 G2[5][16];
  float H[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    H[d] = -0.25f - 0.001f * (float)((lane + d) & 7);
#pragma unroll
    for (int k = 0; k < 5; ++k) G2[k][d] = (f2){0.01f * (float)(k + d + (lane & 3)), -0.02f * (float)(k + 2 * d)};
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)rows, (short)0, (int)0x7FFFFFFF, 0x00020000);
  uint32_t roff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) roff[g] = 2u * (uint32_t)(256 * g + 4 * lane);
  auto fetch = [&](int r, uint2 (&q)[4]) {
    const uint32_t rowb = (uint32_t)(r % n_rows) * 2048u;
#pragma unroll
    for (int g = 0; g < 4; ++g) { const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)roff[g], (int)rowb, 0); q[g] = make_uint2(v.x, v.y); }
  };
  float check = 0.f;
  int r0 = (int)(blockIdx.x * 97 + (tid >> 6) * 13);
  uint2 qn[4];
  fetch(r0, qn);
  for (int s = 0; s < n_samples; ++s) {
    uint2 q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = qn[g];
    fetch(r0 + s + 1, qn);
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      irec::rs_f2 acc[10];                                  // [g_local * 5 + k]: dim group 2 gp + g_local, beams 2 k, 2 k + 1
#pragma unroll
      for (int a = 0; a < 10; ++a) acc[a] = (irec::rs_f2){0.f, 0.f};
#pragma unroll
      for (int gl = 0; gl < 2; ++gl) {
        const int g = 2 * gp + gl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t w = (i & 2) ? q[g].y : q[g].x;
          f2 z[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            uint32_t a0, a1;
            if (i & 1) {
              asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a0) : "v"(w), "s"(bet[2 * k]));
              asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(w), "s"(bet[2 * k + 1]));
            } else {
              asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a0) : "v"(w), "s"(bet[2 * k]));
              asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a1) : "v"(w), "s"(bet[2 * k + 1]));
            }
            z[k].x = irec::lds_abs_f32(a0);
            z[k].y = irec::lds_abs_f32(a1);
          }
          const f2 h2 = {H[4 * g + i], H[4 * g + i]};
          f2 in[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) in[k] = __builtin_elementwise_fma(h2, z[k], G2[k][4 * g + i]);
#pragma unroll
          for (int k = 0; k < 5; ++k) acc[gl * 5 + k] = __builtin_elementwise_fma(in[k], z[k], acc[gl * 5 + k]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      check += irec::reduce_scatter_20(acc, lane);
    }
  }
  out[blockIdx.x * blockDim.x + tid] = check;
}

template <int NWV>
static void run(int n_cu, double clk, const float *tab, const uint16_t *rows, int n_rows, const uint32_t *beta, float *out, int n_samples) {
  const size_t lds = (size_t)3 * TAB * 4 + 64;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wave10_kernel<NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(wave10_kernel<NWV>, dim3(n_cu), dim3(NWV * 64), lds, 0, tab, rows, n_rows, beta, out, n_samples / 8);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(wave10_kernel<NWV>, dim3(n_cu), dim3(NWV * 64), lds, 0, tab, rows, n_rows, beta, out, n_samples);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lookups = (double)n_cu * NWV * 64.0 * (double)n_samples * 160.0;
  printf("one wave x ten beams x 16 dims per lane, %2d waves/CU: %8.3f ms  %7.2f G look-ups/s  %6.2f look-ups/clk/CU (at %.2f GHz)\n", NWV, ms,
         lookups / (ms * 1e-3) * 1e-9, lookups / (ms * 1e-3) / (clk * 1e9) / n_cu, clk);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e-6;
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.gcnArchName, n_cu, clk);
  std::vector<float> tab(TAB);
  for (int i = 0; i < TAB; ++i) tab[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
  const int n_rows = 1024;                                        // 2 MB of rows: L2-resident, as a call's proposal tables
  std::vector<uint16_t> rows((size_t)n_rows * 1024);
  uint32_t x = 12345u;
  for (auto &v : rows) { x = x * 1664525u + 1013904223u; v = (uint16_t)((x >> 8) % (2 * TAB)); }   // dlog + 10006 c, random copy bits
  std::vector<uint32_t> beta(10);
  for (auto &v : beta) { x = x * 1664525u + 1013904223u; v = (x >> 8) % TAB; }
  float *d_tab, *d_out; uint16_t *d_rows; uint32_t *d_beta;
  CHECK(hipMalloc(&d_tab, TAB * 4)); CHECK(hipMalloc(&d_out, (size_t)n_cu * 1024 * 4));
  CHECK(hipMalloc(&d_rows, rows.size() * 2)); CHECK(hipMalloc(&d_beta, 40));
  CHECK(hipMemcpy(d_tab, tab.data(), TAB * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_beta, beta.data(), 40, hipMemcpyHostToDevice));
  printf("random copy bits (busiest bank ~3.5 addresses per 32 lanes; the 2-choice assignment of alpha_choice_kernel has 2.15):\n");
  run<4>(n_cu, clk, d_tab, d_rows, n_rows, d_beta, d_out, 4000);
  run<8>(n_cu, clk, d_tab, d_rows, n_rows, d_beta, d_out, 4000);
  // conflict-free rows: dim 4 l + i of a group lands on bank l mod 32 whatever the beam's rotation -- the VALU-side ceiling
  for (size_t r = 0; r < (size_t)n_rows; ++r)
    for (int d = 0; d < 1024; ++d) { x = x * 1664525u + 1013904223u; rows[r * 1024 + d] = (uint16_t)((((x >> 8) % 600u) * 32u) + (uint32_t)((d >> 2) & 31)); }
  CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 2, hipMemcpyHostToDevice));
  printf("conflict-free rows (VALU-side ceiling):\n");
  run<4>(n_cu, clk, d_tab, d_rows, n_rows, d_beta, d_out, 4000);
  run<8>(n_cu, clk, d_tab, d_rows, n_rows, d_beta, d_out, 4000);
  return 0;
}
