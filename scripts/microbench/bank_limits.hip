// bank_limits.hip -- what bounds the look-up rate of ANY <= 160 KB table family on the gfx950 LDS (round 5; DESIGN.md §4 "The
// gather ceiling").  Diagnostics only: nothing here is part of the product.
//
// The encoder's look-up is z = lut2[alpha' + beta]: alpha' = dlog r + 10006 c per (sample, dim) from the proposal table, beta = dlog hash
// per beam, one ds_read_b32 per (lane, beam).  A 32-lane group costs as many LDS cycles as its busiest bank has distinct addresses, and
// bank(alpha' + beta) = (alpha' + beta) mod 32: the conflicts of a group are a property of the proposal table alone.
//
//  host part   the DISTRIBUTION of the busiest bank per 32-lane group (10^5 random groups) under
//                1 choice            one table copy (random banks)
//                2 choices, rings    what the product does: copy bit c moves a lane by 22 banks (10006 mod 32), exact min-max assignment
//                2 choices, free     a second bank independent of the first (not realisable with identical copies; for comparison)
//                3 choices           c in {0, 1, 2}: banks a, a + 22, a + 12 -- FOUR table copies = 160 096 B, which no kernel can afford
//              and the look-up ceiling each implies (64 lanes / (cycles of group 0 + group 1), two groups per wave instruction).
//  device part the instruction stream of the team encoder's scoring loop (row stream, one add per look-up, packed fma) at 12 waves per CU on
//              address streams whose every group has busiest bank exactly L = 1, 2, 3 and on the product's 2-choice mixture: cycles per
//              wave instruction as a function of L -- is the pipe's cost L cycles per group, and what is lost on top of it? -- plus a
//              double-buffered issue (the product's half-slot pipeline), and a share of the look-ups sent down the vector-memory path
//              (buffer loads of the same table from L1 / L2) beside the LDS pipe.
//
//   hipcc --offload-arch=gfx950 -O3 -o bank_limits bank_limits.hip && ./bank_limits
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TAB = 10006;
constexpr int NBEAM = 20;

typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ float lds_abs(uint32_t byte_addr) { return *(lds_cfloat *)(uintptr_t)byte_addr; }
typedef float f2 __attribute__((ext_vector_type(2)));

// VAR 0: issue the 20 look-ups of a row, wait, consume (scalar fma)          -- gather_rates.hip's form
// VAR 1: the same with packed fma (v_pk_fma_f32: two beams per instruction)  -- the product's arithmetic
// VAR 2: packed fma, double buffered in granules of 10 look-ups              -- the product's half-slot pipeline (<20,3,1>)
// NG: look-ups per row that take the vector-memory path (global copy of the three tables) instead of the LDS
template <int VAR, int NG>
__global__ __launch_bounds__(1024) void stream_kernel(const float *__restrict__ tab_g, const uint32_t *__restrict__ atab, int n_rows,
                                                      const uint32_t *__restrict__ beta, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *t = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  for (int k = tid; k < 3 * TAB; k += blockDim.x) t[k] = tab_g[k];
  __syncthreads();
  uint32_t bet[NBEAM];
#pragma unroll
  for (int b = 0; b < NBEAM; ++b) bet[b] = __builtin_amdgcn_readfirstlane(beta[b]) * 4u;
  f2 acc[NBEAM / 2];
#pragma unroll
  for (int b = 0; b < NBEAM / 2; ++b) acc[b] = (f2){0.f, 0.f};
  const f2 H = {0.25f, 0.25f}, G = {1.0f, 1.0f};
  int row = (blockIdx.x * 7 + (tid >> 6) * 13) % n_rows;
  uint32_t al_next = atab[row * 64 + lane];
  const char *tg = reinterpret_cast<const char *>(tab_g);
  if constexpr (VAR == 2) {
    f2 z[2][5];
    uint32_t al = al_next * 4u;
    row = row + 1 < n_rows ? row + 1 : 0;
    al_next = atab[row * 64 + lane];
#pragma unroll
    for (int k = 0; k < 5; ++k) { z[0][k].x = lds_abs(al + bet[2 * k]); z[0][k].y = lds_abs(al + bet[2 * k + 1]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // issue the next granule, then consume the current one
        if (h == 0) {
#pragma unroll
          for (int k = 0; k < 5; ++k) { z[1][k].x = lds_abs(al + bet[10 + 2 * k]); z[1][k].y = lds_abs(al + bet[10 + 2 * k + 1]); }
        } else {
          al = al_next * 4u;
          row = row + 1 < n_rows ? row + 1 : 0;
          al_next = atab[row * 64 + lane];
#pragma unroll
          for (int k = 0; k < 5; ++k) { z[0][k].x = lds_abs(al + bet[2 * k]); z[0][k].y = lds_abs(al + bet[2 * k + 1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          asm volatile("" : "+v"(z[h][k]));
          acc[5 * h + k] = __builtin_elementwise_fma(__builtin_elementwise_fma(H, z[h][k], G), z[h][k], acc[5 * h + k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      const uint32_t al = al_next * 4u;
      row = row + 1 < n_rows ? row + 1 : 0;
      al_next = atab[row * 64 + lane];
      float z[NBEAM];
#pragma unroll
      for (int b = 0; b < NBEAM; ++b) {
        if (b < NBEAM - NG) z[b] = lds_abs(al + bet[b]);
        else z[b] = *reinterpret_cast<const float *>(tg + (al + bet[b]));
      }
      if constexpr (VAR == 0) {
#pragma unroll
        for (int k = 0; k < NBEAM / 2; ++k) {   // (two scalar fma chains per pair: asm barriers keep them from being packed)
          float a0 = acc[k].x, a1 = acc[k].y;
          a0 = fmaf(fmaf(0.25f, z[2 * k], 1.0f), z[2 * k], a0);
          asm volatile("" : "+v"(a0));
          a1 = fmaf(fmaf(0.25f, z[2 * k + 1], 1.0f), z[2 * k + 1], a1);
          asm volatile("" : "+v"(a1));
          acc[k] = (f2){a0, a1};
        }
      } else {
#pragma unroll
        for (int k = 0; k < NBEAM / 2; ++k) {
          const f2 zz = {z[2 * k], z[2 * k + 1]};
          acc[k] = __builtin_elementwise_fma(__builtin_elementwise_fma(H, zz, G), zz, acc[k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NBEAM / 2; ++b) s += acc[b].x + acc[b].y;
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int VAR, int NG>
static double run(const char *name, int nt, int n_cu, const float *tab3, const uint32_t *atab, int n_rows, const uint32_t *beta, float *out,
                  int iters, double clk_ghz, double mean_cycles) {
  const size_t lds = (size_t)3 * TAB * 4 + 64;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_kernel<VAR, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((stream_kernel<VAR, NG>), dim3(n_cu), dim3(nt), lds, 0, tab3, atab, n_rows, beta, out, iters / 8);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((stream_kernel<VAR, NG>), dim3(n_cu), dim3(nt), lds, 0, tab3, atab, n_rows, beta, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lookups = (double)n_cu * nt * (double)iters * NBEAM;
  const double rate = lookups / (ms * 1e-3) / (clk_ghz * 1e9) / n_cu;
  // cycles the CU spends per LDS wave instruction (64 look-ups) vs what the banks alone would take
  const double cyc = 64.0 * (NBEAM - NG) / NBEAM / rate;
  printf("%-64s %2d waves/CU %8.3f ms %6.2f look-ups/clk/CU  %5.2f clk per LDS wave-instr (banks alone: %4.2f)\n", name, nt / 64, ms, rate, cyc,
         mean_cycles);
  return rate;
}

// ---- exact min-max assignments (host) ----
// rings: lanes are edges (a, a + 22) on two 16-rings of banks; returns the optimum and the copy bits
static int assign_rings(const uint32_t *alpha, int *c) {
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
// general d-choice: smallest L such that the 32 lanes can be matched to banks with capacity L (augmenting paths)
static bool try_match(int j, int L, const int (*ch)[3], int d, int *load, std::vector<int> *occ, bool *seen) {
  for (int k = 0; k < d; ++k) {
    const int b = ch[j][k];
    if (seen[b]) continue;
    seen[b] = true;
    if (load[b] < L) { occ[b].push_back(j); ++load[b]; return true; }
    for (size_t q = 0; q < occ[b].size(); ++q) {
      const int other = occ[b][q];
      if (try_match(other, L, ch, d, load, occ, seen)) { occ[b][q] = j; return true; }
    }
  }
  return false;
}
static int assign_general(const int (*ch)[3], int d) {
  for (int L = 1; L <= 32; ++L) {
    int load[32] = {0};
    std::vector<int> occ[32];
    bool ok = true;
    for (int j = 0; j < 32 && ok; ++j) {
      bool seen[32] = {false};
      ok = try_match(j, L, ch, d, load, occ, seen);
    }
    if (ok) return L;
  }
  return 32;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e-6;
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.gcnArchName, n_cu, clk);
  uint32_t x = 20261004u;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) % TAB; };

  // ---------------- host: distributions ----------------
  const int NGRP = 100000;
  long hist[4][9] = {{0}};
  double sum[4] = {0};
  for (int g = 0; g < NGRP; ++g) {
    uint32_t al[32]; int c[32];
    for (int j = 0; j < 32; ++j) al[j] = rnd();
    int cnt[32] = {0}, mx = 0;
    for (int j = 0; j < 32; ++j) { const int bk = al[j] & 31; if (++cnt[bk] > mx) mx = cnt[bk]; }
    const int L2r = assign_rings(al, c);
    int ch[32][3];
    for (int j = 0; j < 32; ++j) { ch[j][0] = al[j] & 31; ch[j][1] = rnd() & 31; ch[j][2] = 0; }
    const int L2f = assign_general(ch, 2);
    for (int j = 0; j < 32; ++j) { ch[j][0] = al[j] & 31; ch[j][1] = (al[j] + 22) & 31; ch[j][2] = (al[j] + 12) & 31; }
    const int L3 = assign_general(ch, 3);
    const int v[4] = {mx, L2r, L2f, L3};
    for (int k = 0; k < 4; ++k) { hist[k][v[k] < 8 ? v[k] : 8]++; sum[k] += v[k]; }
  }
  const char *nm[4] = {"1 choice (one copy, random banks)", "2 choices, rings of 22 (the product: three copies)", "2 choices, independent second bank (hypothetical)",
                       "3 choices, banks a / a+22 / a+12 (FOUR copies: 160 096 B)"};
  printf("\nbusiest bank per 32-lane group, %d random groups (share of groups at each load; ceiling = 32 / mean look-ups/clk/CU):\n", NGRP);
  printf("%-58s %7s %7s %7s %7s %7s %7s   mean  ceiling\n", "", "L=1", "L=2", "L=3", "L=4", "L=5", "L>=6");
  for (int k = 0; k < 4; ++k) {
    printf("%-58s", nm[k]);
    for (int L = 1; L <= 5; ++L) printf(" %6.2f%%", 100.0 * hist[k][L] / NGRP);
    printf(" %6.2f%%", 100.0 * (hist[k][6] + hist[k][7] + hist[k][8]) / NGRP);
    printf("  %5.3f  %6.2f\n", sum[k] / NGRP, 32.0 / (sum[k] / NGRP));
  }
  {
    double m2 = 0, m2c = 0, at3 = 0;
    for (int L = 1; L <= 8; ++L) { const double pL = (double)hist[1][L] / NGRP; m2 += L * pL; m2c += (L < 2 ? L : 2) * pL; if (L >= 3) at3 += pL; }
    printf("\"never 3\" under the product's assignment: %.2f %% of the groups sit at L >= 3; clamping them to 2 would move the mean %.3f -> %.3f, the "
           "ceiling %.2f -> %.2f look-ups/clk/CU (+%.1f %%)\n", 100 * at3, m2, m2c, 32.0 / m2, 32.0 / m2c, 100.0 * (m2 / m2c - 1.0));
  }

  // ---------------- device: cost per load level ----------------
  std::vector<float> tab3(3 * TAB);
  for (int i = 0; i < 3 * TAB; ++i) tab3[i] = (float)(((i % TAB) * 2654435761u) >> 8) / 16777216.0f - 0.5f;
  std::vector<uint32_t> beta(NBEAM);
  for (auto &v : beta) v = rnd();
  float *d_tab, *d_out; uint32_t *d_beta;
  CHECK(hipMalloc(&d_tab, tab3.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)n_cu * 1024 * 4)); CHECK(hipMalloc(&d_beta, NBEAM * 4));
  CHECK(hipMemcpy(d_tab, tab3.data(), tab3.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_beta, beta.data(), NBEAM * 4, hipMemcpyHostToDevice));
  const int n_rows = 1024;
  // row streams: [0] the product's mixture (random groups, optimum assignment); [L] every group at busiest bank exactly L
  std::vector<uint32_t> rows[4];
  double mean_cyc[4] = {0, 2, 4, 6};
  for (int k = 0; k < 4; ++k) rows[k].resize((size_t)n_rows * 64);
  {
    double s0 = 0;
    for (int r = 0; r < n_rows; ++r)
      for (int h = 0; h < 2; ++h) {
        uint32_t al[32]; int c[32];
        for (int j = 0; j < 32; ++j) al[j] = rnd();
        s0 += assign_rings(al, c);
        for (int j = 0; j < 32; ++j) rows[0][(size_t)r * 64 + h * 32 + j] = al[j] + (uint32_t)TAB * c[j];
        // L = 1: a permutation of the banks; L = 2: sixteen banks twice; L = 3: ten banks three times + two once (distinct addresses throughout)
        int perm[32];
        for (int j = 0; j < 32; ++j) perm[j] = j;
        for (int j = 31; j > 0; --j) { const int q = rnd() % (j + 1); std::swap(perm[j], perm[q]); }
        for (int L = 1; L <= 3; ++L)
          for (int j = 0; j < 32; ++j) {
            const int bank = L == 1 ? perm[j] : L == 2 ? perm[j / 2] : perm[j < 30 ? j / 3 : 10 + (j - 30)];
            const uint32_t base = (rnd() % (uint32_t)((TAB - 64) / 32)) * 32u;     // a multiple of 32: bank = the low five bits
            rows[L][(size_t)r * 64 + h * 32 + j] = base + (uint32_t)bank + (uint32_t)TAB * 0u;
          }
      }
    mean_cyc[0] = s0 / n_rows;   // two groups per row
  }
  uint32_t *d_rows[4];
  for (int k = 0; k < 4; ++k) {
    CHECK(hipMalloc(&d_rows[k], rows[k].size() * 4));
    CHECK(hipMemcpy(d_rows[k], rows[k].data(), rows[k].size() * 4, hipMemcpyHostToDevice));
  }
  const int iters = 20000;
  printf("\ndevice: cycles per LDS wave instruction against the busiest bank of its two groups (12 waves per CU unless stated)\n");
  const char *sn[4] = {"product's mixture (2-choice optimum)", "every group L = 1 (conflict-free)", "every group L = 2", "every group L = 3"};
  for (int k = 0; k < 4; ++k) {
    char nmb[96];
    snprintf(nmb, sizeof nmb, "scalar fma, wait per row          | %s", sn[k]);
    run<0, 0>(nmb, 768, n_cu, d_tab, d_rows[k], n_rows, d_beta, d_out, iters, clk, mean_cyc[k]);
    snprintf(nmb, sizeof nmb, "packed fma, wait per row          | %s", sn[k]);
    run<1, 0>(nmb, 768, n_cu, d_tab, d_rows[k], n_rows, d_beta, d_out, iters, clk, mean_cyc[k]);
    snprintf(nmb, sizeof nmb, "packed fma, double-buffered halves | %s", sn[k]);
    run<2, 0>(nmb, 768, n_cu, d_tab, d_rows[k], n_rows, d_beta, d_out, iters, clk, mean_cyc[k]);
  }
  printf("\noccupancy (product's mixture, packed fma, double-buffered halves):\n");
  for (int nt : {256, 512, 768, 1024}) run<2, 0>("  waves", nt, n_cu, d_tab, d_rows[0], n_rows, d_beta, d_out, iters, clk, mean_cyc[0]);
  printf("\na share of the look-ups down the vector-memory path (global copy of the three tables, L1 / L2), product's mixture, packed fma, wait per row:\n");
  run<1, 0>("  20 LDS +  0 global", 768, n_cu, d_tab, d_rows[0], n_rows, d_beta, d_out, iters, clk, mean_cyc[0]);
  run<1, 1>("  19 LDS +  1 global", 768, n_cu, d_tab, d_rows[0], n_rows, d_beta, d_out, iters / 2, clk, mean_cyc[0]);
  run<1, 2>("  18 LDS +  2 global", 768, n_cu, d_tab, d_rows[0], n_rows, d_beta, d_out, iters / 2, clk, mean_cyc[0]);
  run<1, 4>("  16 LDS +  4 global", 768, n_cu, d_tab, d_rows[0], n_rows, d_beta, d_out, iters / 4, clk, mean_cyc[0]);
  return 0;
}
