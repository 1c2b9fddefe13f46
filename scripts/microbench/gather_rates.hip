// gather_rates.hip -- microbenchmark behind DESIGN.md §4 "gather roofline": how many random 4-byte table look-ups per
// clock one gfx950 CU sustains from (a) LDS with ds_read_b32, (b) the vector L1 / L2 path, (c) both at once, and what
// the address arithmetic in front of each look-up costs.  Diagnostics only: nothing here is part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_rates gather_rates.hip && ./gather_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TAB = 10006;          // entries of the quantile table
constexpr int NBEAM = 20;           // look-ups issued back to back (one per beam)

typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ float lds_abs(uint32_t byte_addr) { return *(lds_cfloat *)(uintptr_t)byte_addr; }

// MODE 0: LDS, 3-op address (add, subtract, unsigned min = mod 10006)      -- what the r01 encoder does
// MODE 1: LDS, 1-op address (add only; table stored twice back to back)
// MODE 2: LDS, 1-op address, conflict-free (lane l always hits bank l % 32)
// MODE 3: global table (L1/L2 path), 1-op address
// MODE 4: 16 look-ups from LDS (1-op) + 4 from the global table per 20
// MODE 5: LDS 3-op, but only the address arithmetic + fma (no look-up): VALU floor
template <int MODE>
__global__ __launch_bounds__(1024) void gather_kernel(const float *__restrict__ tab_g, const uint32_t *__restrict__ alpha,
                                                      const uint32_t *__restrict__ beta, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *t = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x;
  const bool twice = MODE == 1 || MODE == 2 || MODE == 4;
  for (int k = tid; k < (twice ? 2 * TAB : TAB); k += blockDim.x) t[k] = tab_g[k % TAB];
  __syncthreads();
  uint32_t bet[NBEAM];
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) bet[b] = __builtin_amdgcn_readfirstlane(beta[b]) * 4u;
  float acc[NBEAM];
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) acc[b] = 0.f;
  const float H = 0.25f;
  uint32_t a = alpha[(blockIdx.x * blockDim.x + tid) % 65536] * 4u;
  for (int it = 0; it < iters; ++it) {
    // next pseudo-random alpha in [0, 10006): cheap LCG step, amortised over the 20 look-ups
    a = (a * 1664525u + 1013904223u);
    uint32_t al = ((a >> 8) % (uint32_t)TAB) * 4u;
    if (MODE == 2) al = ((al >> 7) << 7) + (uint32_t)(tid & 31) * 4u;   // bank = lane % 32 after adding a multiple of 128 B
    float z[NBEAM];
#pragma unroll
    for (int b = 0; b < NBEAM; ++b) {
      uint32_t ad = al + bet[b];
      if (MODE == 0 || MODE == 5) {
        const uint32_t ad2 = ad - (uint32_t)TAB * 4u;
        ad = ad2 < ad ? ad2 : ad;
      }
      if (MODE == 2) ad = al + (bet[b] & ~127u);
      if (MODE == 5) z[b] = __uint_as_float(ad | 0x3f800000u);
      else if (MODE == 3 || (MODE == 4 && b >= 16)) {
        if (MODE == 4) { const uint32_t ad2 = ad - (uint32_t)TAB * 4u; ad = ad2 < ad ? ad2 : ad; }
        if (MODE == 3) { const uint32_t ad2 = ad - (uint32_t)TAB * 4u; ad = ad2 < ad ? ad2 : ad; }
        z[b] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(tab_g) + ad);
      } else z[b] = lds_abs(ad);
    }
#pragma unroll
    for (int b = 0; b < NBEAM; ++b) acc[b] = fmaf(fmaf(H, z[b], 1.0f), z[b], acc[b]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) s += acc[b];
  out[blockIdx.x * blockDim.x + tid] = s;
}


// MODE 6/7: alpha' = dlog + 10006 * c streamed from a precomputed table (4 bytes per lane per iteration, coalesced), the
// quantile table stored THREE times back to back so that copies c = 0 and c = 1 never wrap.  MODE 6: c = 0 everywhere
// (random banks); MODE 7: c chosen per 32-lane group on the host so that the busiest bank serves as few distinct
// addresses as possible (bank = (alpha + 22 c + beta) mod 32: the choice does not depend on beta).
__global__ __launch_bounds__(1024) void gather3_kernel(const float *__restrict__ tab_g, const uint32_t *__restrict__ atab, int n_rows,
                                                       const uint32_t *__restrict__ beta, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *t = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int k = tid; k < 3 * TAB; k += blockDim.x) t[k] = tab_g[k % TAB];
  __syncthreads();
  uint32_t bet[NBEAM];
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) bet[b] = __builtin_amdgcn_readfirstlane(beta[b]) * 4u;
  float acc[NBEAM];
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) acc[b] = 0.f;
  const float H = 0.25f;
  int row = (blockIdx.x * 7 + (tid >> 6) * 13) % n_rows;
  uint32_t al_next = atab[row * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    const uint32_t al = al_next * 4u;
    row = row + 1 < n_rows ? row + 1 : 0;
    al_next = atab[row * 64 + lane];
    float z[NBEAM];
#pragma unroll
    for (int b = 0; b < NBEAM; ++b) z[b] = lds_abs(al + bet[b]);
#pragma unroll
    for (int b = 0; b < NBEAM; ++b) acc[b] = fmaf(fmaf(H, z[b], 1.0f), z[b], acc[b]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) s += acc[b];
  out[blockIdx.x * blockDim.x + tid] = s;
}

static void run3(const char *name, int nt, int n_cu, const float *tab, const uint32_t *atab, int n_rows, const uint32_t *beta, float *out,
                 int iters, double clk_ghz) {
  const size_t lds = (size_t)3 * TAB * 4 + 64;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gather3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(gather3_kernel, dim3(n_cu), dim3(nt), lds, 0, tab, atab, n_rows, beta, out, iters / 8);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(gather3_kernel, dim3(n_cu), dim3(nt), lds, 0, tab, atab, n_rows, beta, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lookups = (double)n_cu * nt * (double)iters * NBEAM;
  printf("%-44s %4d thr x 1 wg/cu = %2d waves/CU  %8.3f ms  %7.2f G lookups/s  %6.2f lookups/clk/CU (at %.2f GHz)\n", name, nt, nt / 64, ms,
         lookups / (ms * 1e-3) * 1e-9, lookups / (ms * 1e-3) / (clk_ghz * 1e9) / n_cu, clk_ghz);
}

// exact min-max orientation of one 32-lane group (host): lanes are edges (a, a + 22) on two 16-rings of banks
static int assign_group(const uint32_t *alpha, int *c) {
  int a[32];
  for (int j = 0; j < 32; ++j) a[j] = alpha[j] & 31;
  int worst = 0;
  for (int cyc = 0; cyc < 2; ++cyc) {
    int bank[16], n[16], x[16];
    for (int p = 0; p < 16; ++p) { bank[p] = (cyc + 22 * p) & 31; n[p] = 0; }
    for (int p = 0; p < 16; ++p) for (int j = 0; j < 32; ++j) n[p] += a[j] == bank[p];
    bool done = false;
    for (int L = 1; L <= 32 && !done; ++L)
      for (int x0 = 0; x0 <= n[0] && !done; ++x0) {
        x[0] = x0; bool ok = true;
        for (int p = 1; p < 16 && ok; ++p) { const int ub = L - n[p - 1] + x[p - 1]; if (ub < 0) ok = false; else x[p] = n[p] < ub ? n[p] : ub; }
        if (ok && x[0] + n[15] - x[15] <= L) { done = true; if (L > worst) worst = L; }
      }
    for (int p = 0; p < 16; ++p) { int r = 0; for (int j = 0; j < 32; ++j) if (a[j] == bank[p]) { c[j] = r < x[p] ? 0 : 1; ++r; } }
  }
  return worst;
}

template <int MODE>
static void run(const char *name, int nt, int wg_per_cu, int n_cu, const float *tab, const uint32_t *alpha, const uint32_t *beta, float *out,
                int iters, double clk_ghz) {
  const bool twice = MODE == 1 || MODE == 2 || MODE == 4;
  const size_t lds = (size_t)(twice ? 2 : 1) * TAB * 4 + 64;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = wg_per_cu * n_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(nt), lds, 0, tab, alpha, beta, out, iters / 8);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(nt), lds, 0, tab, alpha, beta, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lookups = (double)grid * nt * (double)iters * NBEAM;
  const double per_clk_cu = lookups / (ms * 1e-3) / (clk_ghz * 1e9) / n_cu;
  printf("%-44s %4d thr x %d wg/cu = %2d waves/CU  %8.3f ms  %7.2f G lookups/s  %6.2f lookups/clk/CU (at %.2f GHz)\n", name, nt, wg_per_cu, nt * wg_per_cu / 64, ms,
         lookups / (ms * 1e-3) * 1e-9, per_clk_cu, clk_ghz);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e-6; // kHz -> GHz
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.gcnArchName, n_cu, clk);
  std::vector<float> tab(TAB);
  for (int i = 0; i < TAB; ++i) tab[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
  std::vector<uint32_t> alpha(65536), beta(NBEAM);
  uint32_t x = 12345u;
  for (auto &v : alpha) { x = x * 1664525u + 1013904223u; v = (x >> 8) % TAB; }
  for (auto &v : beta) { x = x * 1664525u + 1013904223u; v = (x >> 8) % TAB; }
  float *d_tab, *d_out; uint32_t *d_alpha, *d_beta;
  CHECK(hipMalloc(&d_tab, TAB * 4)); CHECK(hipMalloc(&d_out, (size_t)4 * n_cu * 1024 * 4));
  CHECK(hipMalloc(&d_alpha, 65536 * 4)); CHECK(hipMalloc(&d_beta, NBEAM * 4));
  CHECK(hipMemcpy(d_tab, tab.data(), TAB * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_alpha, alpha.data(), 65536 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_beta, beta.data(), NBEAM * 4, hipMemcpyHostToDevice));
  const int iters = 20000;
  struct Cfg { int nt, w; };
  for (Cfg c : {Cfg{256, 1}, Cfg{256, 2}, Cfg{512, 1}, Cfg{1024, 1}}) {
    run<0>("LDS, 3-op address (mod 10006)", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters, clk);
    if (c.w == 1) {
      run<1>("LDS, 1-op address (table stored twice)", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters, clk);
      run<2>("LDS, 1-op address, conflict-free", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters, clk);
      run<4>("16 LDS (1-op) + 4 global (3-op) per 20", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters, clk);
    }
    run<3>("global table (L1/L2), 3-op address", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters / 4, clk);
    run<5>("address arithmetic + fma only (no look-up)", c.nt, c.w, n_cu, d_tab, d_alpha, d_beta, d_out, iters, clk);
  }

  {
    const int n_rows = 1024;
    std::vector<uint32_t> a6(n_rows * 64), a7(n_rows * 64);
    double sum_rand = 0, sum_opt = 0;
    for (int r = 0; r < n_rows; ++r)
      for (int h = 0; h < 2; ++h) {
        uint32_t al[32]; int c[32];
        for (int j = 0; j < 32; ++j) { x = x * 1664525u + 1013904223u; al[j] = (x >> 8) % TAB; }
        int cnt[32] = {0}, mx = 0;
        for (int j = 0; j < 32; ++j) { const int bk = al[j] & 31; if (++cnt[bk] > mx) mx = cnt[bk]; }
        sum_rand += mx; sum_opt += assign_group(al, c);
        for (int j = 0; j < 32; ++j) { a6[r * 64 + h * 32 + j] = al[j]; a7[r * 64 + h * 32 + j] = al[j] + (uint32_t)TAB * c[j]; }
      }
    printf("host: mean busiest-bank load per 32-lane group: random %.3f, 2-choice optimum %.3f\n", sum_rand / (2 * n_rows), sum_opt / (2 * n_rows));
    uint32_t *d_a6, *d_a7;
    CHECK(hipMalloc(&d_a6, a6.size() * 4)); CHECK(hipMalloc(&d_a7, a7.size() * 4));
    CHECK(hipMemcpy(d_a6, a6.data(), a6.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_a7, a7.data(), a7.size() * 4, hipMemcpyHostToDevice));
    for (int nt : {256, 512, 768, 1024}) {
      run3("3 copies, streamed alpha, c = 0 (random banks)", nt, n_cu, d_tab, d_a6, n_rows, d_beta, d_out, iters, clk);
      run3("3 copies, streamed alpha, 2-choice optimum", nt, n_cu, d_tab, d_a7, n_rows, d_beta, d_out, iters, clk);
    }
  }
  return 0;
}
