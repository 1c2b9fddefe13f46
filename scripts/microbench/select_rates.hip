// Cycles of the top-B selection (select_topB_sync, irec_fast_common.h) in isolation: one 4-wave workgroup alone on a CU, keys in LDS,
// as in a step of the team / one-table encoders.  Diagnostics; nothing here is part of the product.
//   ./select_rates            -> cycles per selection for the (N, B) of the BASELINE configurations
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <functional>
#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"
using namespace irec;

#ifndef V_THRESH
#define V_THRESH 0
#endif
#ifndef V_RANK
#define V_RANK 0
#endif
#ifndef V_SKIP_RANK
#define V_SKIP_RANK 0
#endif
#ifndef V_SKIP_COUNT
#define V_SKIP_COUNT 0
#endif
#ifndef V_RANK32
#define V_RANK32 0
#endif
#ifndef SM_CANDS
#define SM_CANDS 64
#endif
#ifndef SELECT_IMPL
#define SELECT_IMPL select_topB_sync
#endif

__global__ __launch_bounds__(256) void time_select(const uint32_t *keys_g, int N, int Bnew, int Bcur, int reps, int32_t *sel_out,
                                                   unsigned long long *cycles) {
  __shared__ uint32_t key_s[4096];
  __shared__ SmallLdsT<64, 64, SM_CANDS> sm;
  const int tid = threadIdx.x;
  unsigned long long tot = 0ull;
  for (int r = 0; r < reps; ++r) {
    for (int f = tid; f < N; f += 256) key_s[f] = keys_g[(size_t)r * N + f];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#ifdef V_ASSUME_SMALL
    if (N > 1024) __builtin_unreachable();
#endif
    SELECT_IMPL<256>(key_s, N, Bnew, Bcur, &sm, tid, WorkgroupSync());
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    tot += t1 - t0;
    if (tid < Bnew) { sel_out[((size_t)r * 64 + tid) * 2] = sm.sel_s[tid]; sel_out[((size_t)r * 64 + tid) * 2 + 1] = sm.sel_b[tid]; }
    __syncthreads();
  }
  if (tid == 0) *cycles = tot;
}


// ---- instrumented copy of the N <= 1024 path (stages: 0 wait for keys, 1 loads + lane maxima, 2 threshold count, 3 threshold min, 4 compaction, 5 rank, 6 closing barrier)
template <int NT, class SM, class Sync>
__device__ __forceinline__ void select_staged(uint32_t *key, int N, int Bnew, int Bcur, SM *sm, const int tid, Sync &&sync, unsigned long long *st) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto stamp = [&](int k) { if (tid == 0) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); st[k] += t1 - t0; t0 = t1; } };
  sync();
  stamp(0);
  if (tid < 64) {
    __builtin_amdgcn_s_setprio(3);
    const int nslots = (N + 63) >> 6;
    uint32_t k[16];
    uint32_t M = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int f = q * 64 + tid;
      k[q] = (q < nslots && f < N) ? key[f] : 0u;
      M = k[q] > M ? k[q] : M;
    }
    stamp(1);
#if V_THRESH == 1
    // threshold by bisection on the key bits: 32 rounds of ballot + popcount (scalar after one compare)
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t tryT = T | (1u << bit);
      if (__popcll(__ballot(M >= tryT)) >= Bnew) T = tryT;
    }
    stamp(2);
    stamp(3);
#else
    uint32_t cnt_gt = 0u;
#pragma unroll
    for (int l = 0; l < 64; ++l) cnt_gt += (uint32_t)__builtin_amdgcn_readlane((int)M, l) > M ? 1u : 0u;
    stamp(2);
    uint32_t T = cnt_gt < (uint32_t)Bnew ? M : 0xFFFFFFFFu;
    T = 0xFFFFFFFFu - (uint32_t)wave_max_u64((unsigned long long)(0xFFFFFFFFu - T)); // wave min
    stamp(3);
#endif
    uint32_t base = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < nslots) {
        const bool in = k[q] >= T;
        const unsigned long long mask = __ballot(in);
        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (in && pos < (uint32_t)SM::CANDS) sm->cand[pos] = ((unsigned long long)k[q] << 32) | (uint32_t)(q * 64 + tid);
        base += (uint32_t)__popcll(mask);
      }
    }
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    stamp(4);
#if V_RANK == 1
    {
      // rank with ONE 64-bit compare per candidate: (key, ~flat) packed so that larger is better
      const unsigned long long mine = tid < (int)C ? sm->cand[tid] : 0ull;
      const uint32_t mk1 = (uint32_t)(mine >> 32), mf1 = (uint32_t)mine;
      const unsigned long long mp = ((unsigned long long)mk1 << 32) | (unsigned long long)(0xFFFFFFFFu - mf1);
      uint32_t rank = 0u;
      for (uint32_t l = 0; l < C; ++l) {
        const uint32_t ok_ = (uint32_t)__builtin_amdgcn_readlane((int)mk1, (int)l);
        const uint32_t of_ = (uint32_t)__builtin_amdgcn_readlane((int)mf1, (int)l);
        const unsigned long long op = ((unsigned long long)ok_ << 32) | (unsigned long long)(0xFFFFFFFFu - of_);
        rank += op > mp ? 1u : 0u;
      }
      if (tid < (int)C && rank < (uint32_t)Bnew) { sm->sel_s[rank] = (int32_t)(mf1 / (uint32_t)Bcur); sm->sel_b[rank] = (int32_t)(mf1 % (uint32_t)Bcur); }
      sm->misc[7] = C <= 64u ? 1 : 0;
    }
#else
    sm->misc[7] = rank_survivors(sm, C, Bnew, Bcur, tid, NoPost()) ? 1 : 0;
#endif
    __builtin_amdgcn_s_setprio(0);
    stamp(5);
  }
  sync();
  stamp(6);
}
__global__ __launch_bounds__(256) void time_staged(const uint32_t *keys_g, int N, int Bnew, int Bcur, int reps, int32_t *sel_out, unsigned long long *st_out) {
  __shared__ uint32_t key_s[4096];
  __shared__ SmallLdsT<64, 64, SM_CANDS> sm;
  const int tid = threadIdx.x;
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (int f = tid; f < N; f += 256) key_s[f] = keys_g[(size_t)r * N + f];
    __syncthreads();
    select_staged<256>(key_s, N, Bnew, Bcur, &sm, tid, WorkgroupSync(), st);
    if (tid < Bnew) { sel_out[((size_t)r * 64 + tid) * 2] = sm.sel_s[tid]; sel_out[((size_t)r * 64 + tid) * 2 + 1] = sm.sel_b[tid]; }
    __syncthreads();
  }
  if (tid == 0) for (int k = 0; k < 8; ++k) st_out[k] = st[k];
}

// ---- v3: 32-bit wave min for the threshold, compaction by a wave scan of the per-lane survivor counts, rank by constant-lane readlanes
__device__ __forceinline__ uint32_t wave_min_u32_(uint32_t v) {
  uint32_t o;
  o = xor_lane_u32<32>(v); v = o < v ? o : v;
  o = xor_lane_u32<16>(v); v = o < v ? o : v;
  o = xor_lane_u32<8>(v); v = o < v ? o : v;
  o = xor_lane_u32<4>(v); v = o < v ? o : v;
  o = xor_lane_u32<2>(v); v = o < v ? o : v;
  o = xor_lane_u32<1>(v); v = o < v ? o : v;
  return v;
}
// inclusive prefix sum over the 64 lanes (small counts)
__device__ __forceinline__ uint32_t wave_incl_scan_u32_(uint32_t v, uint32_t &total) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1, lanes without a source read 0
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  const uint32_t row = __lane_id() >> 4;
  v += row == 0 ? 0u : row == 1 ? r0 : row == 2 ? r0 + r1 : r0 + r1 + r2;
  total = r0 + r1 + r2 + r3;
  return v;
}
template <int NT, class SM, class Sync>
__device__ __forceinline__ void select_v3(uint32_t *key, int N, int Bnew, int Bcur, SM *sm, const int tid, Sync &&sync, unsigned long long *st) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto stamp = [&](int k) { if (st && tid == 0) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); st[k] += t1 - t0; t0 = t1; } };
  sync();
  stamp(0);
  if (tid < 64) {
    __builtin_amdgcn_s_setprio(3);
    const int nslots = (N + 63) >> 6;
    uint32_t k[16];
    uint32_t M = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int f = q * 64 + tid;
      k[q] = (q < nslots && f < N) ? key[f] : 0u;
      M = k[q] > M ? k[q] : M;
    }
    stamp(1);
#if V_THRESH == 1
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t tryT = T | (1u << bit);
      if (__popcll(__ballot(M >= tryT)) >= Bnew) T = tryT;
    }
#else
    uint32_t cnt_gt = 0u;
#pragma unroll
    for (int l = 0; l < (V_SKIP_COUNT ? 0 : 64); ++l) cnt_gt += (uint32_t)__builtin_amdgcn_readlane((int)M, l) > M ? 1u : 0u;
    stamp(2);
    // the Bnew-th largest lane maximum: the lane(s) with the largest count below Bnew (without ties the counts are a permutation of
    // 0..63 and the first probe hits)
    uint32_t T = 0u;
    for (int c = Bnew - 1; c >= 0; --c) {
      const unsigned long long hit = __ballot(cnt_gt == (uint32_t)c);
      if (hit) { T = (uint32_t)__builtin_amdgcn_readlane((int)M, (int)__builtin_ctzll(hit)); break; }
    }
    if (V_SKIP_COUNT) T = 1u;
#endif
    stamp(3);
    uint32_t base = 0u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q < nslots) {
        const bool in = k[q] >= T;
        const unsigned long long mask = __ballot(in);
        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (in && pos < (uint32_t)SM::CANDS) sm->cand[pos] = ((unsigned long long)k[q] << 32) | (uint32_t)(q * 64 + tid);
        base += (uint32_t)__popcll(mask);
      }
    }
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    stamp(4);
    if (C <= 64u) {
      const unsigned long long mine = tid < (int)C ? sm->cand[tid] : 0ull;
      const uint32_t mk1 = (uint32_t)(mine >> 32), mf1 = (uint32_t)mine;
      const uint32_t nf1 = ~mf1;
      const unsigned long long mp = ((unsigned long long)mk1 << 32) | nf1;     // larger = better: key descending, flat ascending
      uint32_t rank = 0u;
#if V_RANK32
      // by key alone (one broadcast, one 32-bit compare per candidate); equal keys -- rare -- show up as two survivors with one rank
#pragma unroll
      for (int l0 = 0; l0 < 64; l0 += 8) {
        if ((uint32_t)l0 < C) {   // wave-uniform
#pragma unroll
          for (int l = l0; l < l0 + 8; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)mk1, l) > mk1 ? 1u : 0u;
        }
      }
      volatile unsigned long long *vc = sm->cand;   // (another lane may own the slot: no store-to-load forwarding)
      if (tid < (int)C) vc[rank] = (unsigned long long)tid;
      const bool lost = tid < (int)C && vc[rank] != (unsigned long long)tid;
      if (__ballot(lost)) {   // (wave-uniform) a tie: the exact order, key descending then flat ascending
        rank = 0u;
#endif
#pragma unroll
      for (int l0 = 0; l0 < (V_SKIP_RANK ? 0 : 64); l0 += 8) {
        if ((uint32_t)l0 < C) {   // wave-uniform
#pragma unroll
          for (int l = l0; l < l0 + 8; ++l) {
            const uint32_t ok_ = (uint32_t)__builtin_amdgcn_readlane((int)mk1, l);
            const uint32_t of_ = (uint32_t)__builtin_amdgcn_readlane((int)nf1, l);
            const unsigned long long op = ((unsigned long long)ok_ << 32) | of_;
            rank += op > mp ? 1u : 0u;                                          // (null candidates: key 0 never beats a survivor)
          }
        }
      }
#if V_RANK32
      }
#endif
      if (tid < (int)C && rank < (uint32_t)Bnew) { sm->sel_s[rank] = (int32_t)(mf1 / (uint32_t)Bcur); sm->sel_b[rank] = (int32_t)(mf1 % (uint32_t)Bcur); }
      sm->misc[7] = 1;
    } else sm->misc[7] = rank_survivors(sm, C, Bnew, Bcur, tid, NoPost()) ? 1 : 0;
    __builtin_amdgcn_s_setprio(0);
    stamp(5);
  }
  sync();
  stamp(6);
}
__global__ __launch_bounds__(256) void time_v3(const uint32_t *keys_g, int N, int Bnew, int Bcur, int reps, int32_t *sel_out, unsigned long long *st_out) {
  __shared__ uint32_t key_s[4096];
  __shared__ SmallLdsT<64, 64, SM_CANDS> sm;
  const int tid = threadIdx.x;
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (int f = tid; f < N; f += 256) key_s[f] = keys_g[(size_t)r * N + f];
    __syncthreads();
    const unsigned long long a0 = __builtin_amdgcn_s_memtime();
    select_v3<256>(key_s, N, Bnew, Bcur, &sm, tid, WorkgroupSync(), (unsigned long long *)nullptr);
    const unsigned long long a1 = __builtin_amdgcn_s_memtime();
    st[7] += a1 - a0;
    if (tid < Bnew) { sel_out[((size_t)r * 64 + tid) * 2] = sm.sel_s[tid]; sel_out[((size_t)r * 64 + tid) * 2 + 1] = sm.sel_b[tid]; }
    __syncthreads();
  }
  if (tid == 0) for (int k = 0; k < 8; ++k) st_out[k] = st[k];
}

int main() {
  const int cfg[][3] = {{720, 20, 20}, {200, 10, 10}, {36, 20, 1}, {1024, 20, 20}, {360, 10, 10}, {4440, 30, 30}};
  const int reps = 200;
  for (auto &c : cfg) {
    const int N = c[0], Bnew = c[1], Bcur = c[2];
    std::vector<uint32_t> h((size_t)reps * N);
    srand(1234 + N);
    for (auto &v : h) { float f = -20.f + 5.f * ((float)rand() / RAND_MAX + (float)rand() / RAND_MAX); uint32_t u; memcpy(&u, &f, 4); v = (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
    for (int r = 0; r < reps; r += 7) { h[(size_t)r * N + 3] = h[(size_t)r * N + 11]; }   // a few exact ties
    for (int r = 0; r < reps; r += 3) {   // ties among the best: several copies of the row's maximum, and of a value near the B-th best
      uint32_t *k = &h[(size_t)r * N];
      std::vector<uint32_t> srt(k, k + N); std::sort(srt.begin(), srt.end(), std::greater<uint32_t>());
      for (int c = 0; c < 4; ++c) k[rand() % N] = srt[0];
      if (N > Bnew + 4) for (int c = 0; c < 3; ++c) k[rand() % N] = srt[Bnew - 1];
    }
    uint32_t *d_k; int32_t *d_sel; unsigned long long *d_c;
    hipMalloc(&d_k, h.size() * 4); hipMalloc(&d_sel, (size_t)reps * 64 * 2 * 4); hipMalloc(&d_c, 8);
    hipMemcpy(d_k, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_sel, 0, (size_t)reps * 64 * 2 * 4);
    time_select<<<1, 256>>>(d_k, N, Bnew, Bcur, reps, d_sel, d_c);
    time_select<<<1, 256>>>(d_k, N, Bnew, Bcur, reps, d_sel, d_c);
    hipDeviceSynchronize();
    unsigned long long cyc; hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
    std::vector<int32_t> sel((size_t)reps * 64 * 2);
    hipMemcpy(sel.data(), d_sel, sel.size() * 4, hipMemcpyDeviceToHost);
    // check against a host top-B (value descending, ties to the lower flat index)
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
      std::vector<int> idx(N);
      for (int f = 0; f < N; ++f) idx[f] = f;
      const uint32_t *k = &h[(size_t)r * N];
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return k[a] > k[b]; });
      for (int j = 0; j < Bnew; ++j)
        if (sel[((size_t)r * 64 + j) * 2] != idx[j] / Bcur || sel[((size_t)r * 64 + j) * 2 + 1] != idx[j] % Bcur) ++bad;
    }
    printf("N=%5d Bnew=%2d: %8.0f cycles per selection (s_memtime ticks, 100 MHz x clock ratio), wrong entries %d\n", N, Bnew, (double)cyc / reps, bad);
    if (N <= 1024) {
      unsigned long long *d_st; hipMalloc(&d_st, 64); hipMemset(d_st, 0, 64);
      hipMemset(d_sel, 0, (size_t)reps * 64 * 2 * 4);
      time_staged<<<1, 256>>>(d_k, N, Bnew, Bcur, reps, d_sel, d_st);
      hipDeviceSynchronize();
      unsigned long long st[8]; hipMemcpy(st, d_st, 64, hipMemcpyDeviceToHost);
      hipMemcpy(sel.data(), d_sel, sel.size() * 4, hipMemcpyDeviceToHost);
      int bad2 = 0;
      for (int r = 0; r < reps; ++r) {
        std::vector<int> idx(N);
        for (int f = 0; f < N; ++f) idx[f] = f;
        const uint32_t *k = &h[(size_t)r * N];
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return k[a] > k[b]; });
        for (int j = 0; j < Bnew; ++j)
          if (sel[((size_t)r * 64 + j) * 2] != idx[j] / Bcur || sel[((size_t)r * 64 + j) * 2 + 1] != idx[j] % Bcur) ++bad2;
      }
      double tot = 0; for (int k2 = 0; k2 < 7; ++k2) tot += (double)st[k2];
      printf("   staged (V_THRESH=%d V_RANK=%d): total %6.0f | wait %5.0f | loads+max %5.0f | count %5.0f | min %5.0f | compact %5.0f | rank %5.0f | close %5.0f | wrong %d\n",
             V_THRESH, V_RANK, tot / reps, (double)st[0] / reps, (double)st[1] / reps, (double)st[2] / reps, (double)st[3] / reps, (double)st[4] / reps, (double)st[5] / reps, (double)st[6] / reps, bad2);
      hipMemset(d_st, 0, 64); hipMemset(d_sel, 0, (size_t)reps * 64 * 2 * 4);
      time_v3<<<1, 256>>>(d_k, N, Bnew, Bcur, reps, d_sel, d_st);
      hipDeviceSynchronize();
      hipMemcpy(st, d_st, 64, hipMemcpyDeviceToHost);
      hipMemcpy(sel.data(), d_sel, sel.size() * 4, hipMemcpyDeviceToHost);
      bad2 = 0;
      for (int r = 0; r < reps; ++r) {
        std::vector<int> idx(N);
        for (int f = 0; f < N; ++f) idx[f] = f;
        const uint32_t *k = &h[(size_t)r * N];
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return k[a] > k[b]; });
        for (int j = 0; j < Bnew; ++j)
          if (sel[((size_t)r * 64 + j) * 2] != idx[j] / Bcur || sel[((size_t)r * 64 + j) * 2 + 1] != idx[j] % Bcur) ++bad2;
      }
      tot = 0; for (int k2 = 0; k2 < 7; ++k2) tot += (double)st[k2];
      printf("   v3 without stamps: %6.0f cycles per selection\n", (double)st[7] / reps);
      if (0) printf("   v3:                             total %6.0f | wait %5.0f | loads+max %5.0f | count %5.0f | min %5.0f | compact %5.0f | rank %5.0f | close %5.0f | wrong %d\n",
             tot / reps, (double)st[0] / reps, (double)st[1] / reps, (double)st[2] / reps, (double)st[3] / reps, (double)st[4] / reps, (double)st[5] / reps, (double)st[6] / reps, bad2);
      hipFree(d_st);
    }
    hipFree(d_k); hipFree(d_sel); hipFree(d_c);
  }
  return 0;
}
