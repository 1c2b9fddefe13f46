// irec_ten.hip -- the team encoder of calls with at most TEN beams and at most 256 candidates per step (S * 10 <= 256): the settings both
// drivers of the reference ship as their default -- n_beams = 10, kl_per_partition = 3, extra_samples = 1 -> S = 20, block_size = 1000
// (examples/lossless/compression_performance.py:46-58, examples/lossy/compress_with_lossy_model.py:52-64) -- BASELINE configs[3].
//
// Hot path (reference file:line): BeamSearchCoder.encode_block rec/coding/beam_search_coder.py:53-122; the arithmetic specification
//  and therefore every emitted bit are encode_team_kernel's (irec_team.hip), whose tables, slabs, hand-out and scoring
// loop this kernel shares.  What differs is everything BETWEEN two scoring phases.  A 10 x 20 step has a quarter of the headline's
// look-ups under the same fixed chain of barrier -> combine -> barrier -> one-wave selection -> barrier -> update, and in a batch the
// three teams of a CU then spend half their time outside the scoring loop (r06a stamps: 29 k of 61 k cycles per block-step), so the
// gather pipe idles.  Here a step has ONE team barrier:
//   * partials leave the scoring loop as part[candidate][dim group] (one 16-byte read per candidate), double buffered by step parity;
//   * behind the barrier EVERY wave forms all sort keys (four candidates per lane) and runs the SAME top-B selection in its own
//     registers -- quad maxima, their B-th largest as threshold, the dozen survivors compacted to one per lane through the LDS crossbar
//     (ds_permute_b32: no bank is touched), ranked against their broadcasts, sent to the lane of their rank (ten_select_fast) -- so no
//     wave waits for a selecting wave and nothing is broadcast: the four waves of a team sit on four SIMDs;
//   * hash sums and table offsets of the beams live in lanes (lane j = beam j), parents are fetched by v_readlane / ds_bpermute;
//   * all ten parents of a dim quad are loaded before the first new beam is stored, so blocks of more than 512 dims (one wave per
//     dim group) update their beams IN PLACE: half the slab footprint in the L2; the update's arithmetic runs on dim pairs (v_pk_*_f32);
//   * look-up addresses of a beam pair by one v_pk_add_f32 on their bit patterns (see the scoring loop).
// What bounds it: VALU issue and the LDS gather pipe together, both 68-69 % busy; time follows the VALU work of a step.
// Serves 2 <= B <= 10, S * 10 <= 256, D <= 1024, plain calls (no shared rows, no margins: those stay on encode_team_kernel<10,..>).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"
#include "irec_team_common.h"

#ifndef IREC_TEN_ABLATE
#define IREC_TEN_ABLATE 0       // diagnostics (make variant_ten): phases replaced by stubs -- 1 selection, 2 step constants, 4 update arithmetic, 8 keys,
                                // 16 scoring; the outputs are wrong, the time that remains is the point
#endif
namespace irec {

constexpr int TEN_NB = 10;
constexpr int TEN_MAXC = 256;                    // candidates per step, at most (host: S * 10 <= 256)

struct TenLds {
  float part[2][TEN_MAXC][4];                    // by step parity: group partials of candidate (sample * 10 + beam)
  float cpart[2][16][4];                         // by step parity: group partials of C_b
  double gpart[4];                               // KL group sums of the block in hand
  int32_t misc[8];                               // [0] hand-out slot
  uint32_t bar;                                  // the team barrier's counter
  uint32_t pad[3];
};
constexpr size_t TEN_LDS_ONE = (sizeof(TenLds) + 15) & ~(size_t)15;
__host__ __device__ inline size_t ten_lds_total(int teams) { return T3_BYTES + (size_t)teams * TEN_LDS_ONE; }

typedef float f2t __attribute__((ext_vector_type(2)));

// ---- the step's top-B in one wave's registers (beam_search_coder.py:85-89) ----
// k[q]: sort key of candidate f = 64 q + lane (0: none), N <= 256 candidates, 1 <= Bnew <= 10.  Returns in lane r < Bnew the flat index of
// the candidate of rank r (key descending, ties to the lower flat index: tf.argsort(DESCENDING)).  Every lane of the wave calls.
__device__ __forceinline__ uint32_t ten_select(const uint32_t (&k)[4], int N, int Bnew, int lane) {
  uint32_t sel = 0u;
  if (N <= 64) {
    // one candidate per lane: rank them all (step 0 of a block: S candidates)
    uint32_t rank = 0u;
    for (int l = 0; l < N; ++l) {
      const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)k[0], l);
      rank += (o > k[0] || (o == k[0] && l < lane)) ? 1u : 0u;
    }
    for (int l = 0; l < N; ++l) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rank, l);
      if (r < (uint32_t)Bnew) sel = writelane_u32((uint32_t)l, (int)r, sel);
    }
    return sel;
  }
  // threshold: the Bnew-th largest of the 16 quad maxima -- at least Bnew candidates are >= it (a quad holds 16 of them)
  uint32_t M = k[0] > k[1] ? k[0] : k[1];
  { const uint32_t m2 = k[2] > k[3] ? k[2] : k[3]; M = M > m2 ? M : m2; }
  { const uint32_t o = dpp_u32<0xB1, 0xF>(M, M); M = M > o ? M : o; }      // quad_perm [1,0,3,2]
  { const uint32_t o = dpp_u32<0x4E, 0xF>(M, M); M = M > o ? M : o; }      // quad_perm [2,3,0,1]
  uint32_t cnt_gt = 0u;
#pragma unroll
  for (int g = 0; g < 16; ++g) cnt_gt += (uint32_t)__builtin_amdgcn_readlane((int)M, 4 * g) > M ? 1u : 0u;
  uint32_t T = 0u;
  for (int c = Bnew - 1; c >= 0; --c) {
    const unsigned long long hit = __ballot(cnt_gt == (uint32_t)c);
    if (hit) { T = (uint32_t)__builtin_amdgcn_readlane((int)M, (int)__builtin_ctzll(hit)); break; }   // (wave-uniform)
  }
  T = T ? T : 1u;                                              // (fewer than Bnew quads hold a candidate: every candidate survives)
  // survivors: ranked among themselves by (key, ~flat) as one 64-bit number
  unsigned long long P[4];
  unsigned long long mask[4];
  uint32_t rank[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    P[q] = ((unsigned long long)k[q] << 32) | (unsigned long long)(~(uint32_t)(64 * q + lane));
    mask[q] = __ballot(k[q] >= T);
    rank[q] = 0u;
  }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    unsigned long long m = mask[qq];
    while (m) {                                                // (wave-uniform)
      const int l = (int)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint32_t ok_ = (uint32_t)__builtin_amdgcn_readlane((int)k[qq], l);
      const unsigned long long o = ((unsigned long long)ok_ << 32) | (unsigned long long)(~(uint32_t)(64 * qq + l));
#pragma unroll
      for (int q = 0; q < 4; ++q) rank[q] += o > P[q] ? 1u : 0u;
    }
  }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    unsigned long long m = mask[qq];
    while (m) {
      const int l = (int)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rank[qq], l);
      if (r < (uint32_t)Bnew) sel = writelane_u32((uint32_t)(64 * qq + l), (int)r, sel);
    }
  }
  return sel;
}

// The same, in straight-line vector code (r06: a single wave issues the scalar loops above at ~8 cycles per instruction, 5.4 k cycles per
// selection with or without other waves on its SIMD).  Threshold as above (the B-th largest quad maximum: a wave-wide minimum); the
// survivors -- a dozen -- are compacted to one per lane by v_mbcnt prefix counts and ds_permute_b32 (the LDS crossbar: no bank is
// touched, non-survivors send to lane 63), ranked by the key against the C broadcast survivors, and sent to the lane of their rank by one
// more ds_permute_b32.  Exactly equal keys among the survivors (exactly equal float32 scores: rare) and 64 or more survivors (a tie
// storm) go to ten_select above, which ranks by (key, flat).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  { const uint32_t o = xor_lane_u32<1>(v); v = o < v ? o : v; }
  { const uint32_t o = xor_lane_u32<2>(v); v = o < v ? o : v; }
  { const uint32_t o = xor_lane_u32<4>(v); v = o < v ? o : v; }
  { const uint32_t o = xor_lane_u32<8>(v); v = o < v ? o : v; }
  { const uint32_t o = xor_lane_u32<16>(v); v = o < v ? o : v; }
  { const uint32_t o = xor_lane_u32<32>(v); v = o < v ? o : v; }
  return v;
}
template <class Mid>
__device__ __forceinline__ uint32_t ten_select_fast(const uint32_t (&k)[4], int N, int Bnew, int lane, Mid &&mid) {
  uint32_t M = k[0] > k[1] ? k[0] : k[1];
  { const uint32_t m2 = k[2] > k[3] ? k[2] : k[3]; M = M > m2 ? M : m2; }
  { const uint32_t o = dpp_u32<0xB1, 0xF>(M, M); M = M > o ? M : o; }      // quad_perm [1,0,3,2]
  { const uint32_t o = dpp_u32<0x4E, 0xF>(M, M); M = M > o ? M : o; }      // quad_perm [2,3,0,1]
  uint32_t cnt_gt = 0u;
#pragma unroll
  for (int g = 0; g < 16; ++g) cnt_gt += (uint32_t)__builtin_amdgcn_readlane((int)M, 4 * g) > M ? 1u : 0u;
  // the B-th largest quad maximum = the smallest one that fewer than B others exceed (0 when fewer than B quads hold a candidate)
  uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(cnt_gt < (uint32_t)Bnew ? M : 0xFFFFFFFFu));
  T = T ? T : 1u;
  uint32_t pos[4];
  bool in[4];
  uint32_t base = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    in[q] = k[q] >= T;
    const unsigned long long mask = __ballot(in[q]);
    pos[q] = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    base += (uint32_t)__popcll(mask);
  }
  const uint32_t C = base;                                       // (wave-uniform: ballots)
  if (C > 63u) return ten_select(k, N, Bnew, lane);
  uint32_t ck = 0u, cf = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int dst = (int)((in[q] ? pos[q] : 63u) << 2);
    ck |= (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)k[q]);
    cf |= (uint32_t)__builtin_amdgcn_ds_permute(dst, 64 * q + lane);
  }
  ck = (uint32_t)lane < C ? ck : 0u;                             // (lane 63 caught what the non-survivors sent)
  mid();                                                         // (diagnostic builds: a phase stamp)
  uint32_t rank = 0u, eq = 0u;
  for (uint32_t l = 0; l < C; ++l) {                             // (wave-uniform trip count)
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)ck, (int)l);
    rank += o > ck ? 1u : 0u;
    eq += o == ck ? 1u : 0u;
  }
  if (__ballot((uint32_t)lane < C && eq > 1u)) return ten_select(k, N, Bnew, lane);   // equal keys: the order is by flat index
  const int dst = (int)((((uint32_t)lane < C && rank < (uint32_t)Bnew) ? rank : 63u) << 2);
  return (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)cf);
}

// TEAMS 4-wave teams per workgroup: three for batches (a team per block, the other teams' scoring under its serial phases), two at the full
// register budget for calls of at most two blocks per CU.  (Measured and dropped, r06a: FOUR teams at 128 VGPRs -- no spill in the scoring
// loop -- lose 12 %; two 8-WAVE teams for calls of at most two blocks per CU -- a block's samples on two stripes -- lose 20 % at 302 blocks:
// 128 VGPRs, 304 B of scratch in the serial phases, both stripes repeat selection and update.  A CU that holds two blocks steps each of
// them 1.15 times slower in cycles at a clock 4 % lower (35.0 k -> 40.4 k cycles per block-step, all of it in the scoring loop: 18.3 k ->
// 23.2 k, profiles/r06end/stamps_lone_vs_doubled.log); letting the two teams enter their scoring loops only by turns -- a gate of two LDS
// counters, so that one team gathers under the other's serial phases -- made it 4 % worse still (302 blocks 185 -> 192 us,
// score_gate_ab.log): the scoring team also loses VALU issue slots to the other team's serial phases, the two floors of section "What bounds
// it" again.)
template <int TEAMS>
__global__ __launch_bounds__(TEAMS * TEAM_NT, 1) void encode_ten_kernel(EncArgs A) {
  constexpr int NB = TEN_NB, NT = TEAM_NT, NWT = TEAM_NW, NP = NB / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = A.S, B = A.B;
  const int lane = threadIdx.x & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform by construction
  const int team = wave_wg / NWT, wave = wave_wg % NWT;
  const int tid = (int)threadIdx.x - team * NT;                                 // index inside the team
  TenLds *sm = reinterpret_cast<TenLds *>(smem + T3_BYTES + (size_t)team * TEN_LDS_ONE);
  int32_t *misc = sm->misc;
  const uint16_t *dlog_s = A.dlog4r;                                            // [10006] 4*dlog(j+1), global (L2)
  const int rs_p20 = rs20_owner(lane);                                          // accumulator whose total reduce_scatter_20 leaves here
  const int rs_c = rsn_owner<NB>(lane);                                         // same for the ten C_b partials of the update

  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  commit_table_stamps(A);
  // (measured and not taken, profiles/r06end/: a call of one to 1.5 rows per CU dealt by cost as encode_team_kernel<20,2,1> deals its own --
  //  here the workgroups that hold two rows get the cheapest.  One tensor of 302 blocks gains 5 us of 190, 257 blocks 12 us
  //  (placement_ab.log); calls of many tensors LOSE 1.5 - 7 us, 11 us where K differs between tensors (placement_multi_tensor.log): their
  //  layout lists the small residual blocks last, so the two-row workgroups hold those anyway, and the hand-out by XCD below keeps the
  //  blocks of a tensor -- slices of the same cache lines -- on one L2; what is left is the price of the cost rows and of the ranking.)
  {
    // (measured and dropped, profiles/r06end/prefetch_first_block.log: the first block's gather chain -- descriptors -> permutation -> statistics -- walked once here, under the
    //  table loads, so that the prologue finds the lines close by: 302 blocks 185 -> 194 us.  The chain is bound by the vector L1's one line
    //  per clock -- 5 000 random lines per block --, not by where the lines are: walking it twice costs twice.)
    float *l3 = reinterpret_cast<float *>(smem);
    for (int k = (int)threadIdx.x; k < (int)IREC_PM1; k += TEAMS * NT) {
      const float v = A.lut2[k];
      l3[k] = v; l3[k + IREC_PM1] = v; l3[k + 2 * IREC_PM1] = v;
    }
    if (tid == 0) sm->bar = 0u;
  }
  __syncthreads(); // the only workgroup-wide barrier: from here on the teams never wait for each other
  TeamBarrier tsync{&sm->bar, 0u, (uint32_t)NWT};

  // scratch slab of the team (fast_ws_bytes(10, max_K)): bp int32 [max_K][10] | ... | stats [3][1024] | beams [2][10][1024]
  char *slab = A.ws + ((size_t)blockIdx.x * TEAMS + team) * A.ws_per_wg;
  int32_t *bp = reinterpret_cast<int32_t *>(slab);
  float *beams_g = reinterpret_cast<float *>(slab + A.ws_per_wg - (size_t)2 * NB * FAST_MAX_DIM * 4);
  float *stats_g = beams_g - 3 * FAST_MAX_DIM;

  const int64_t n_slots = A.n_blocks;
  const int64_t n_static = (int64_t)TEAMS * (int64_t)gridDim.x < n_slots ? (int64_t)TEAMS * (int64_t)gridDim.x : n_slots;
#ifdef IREC_TEAM_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long st_t0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#define TSTAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define TSTAMP(slot) do { } while (0)
#endif
  bool first_block = true;
  int steal = 0;
  for (;;) {
    tsync();
    if (tid == 0) {
      int64_t r;
      if (first_block) {
        r = (int64_t)team * (int64_t)gridDim.x + (int64_t)blockIdx.x;
        r = r < n_static ? xcd_static_row(r, n_static, (int)gridDim.x) : n_slots;
      } else r = xcd_pull_row(A, n_static, n_slots, steal);   // (pulling the slot during the previous block's last step: 1.5 % slower, r06p)
      misc[0] = (int32_t)r;
    }
    first_block = false;
    tsync();
    const int64_t blk = __builtin_amdgcn_readfirstlane(misc[0]);   // (wave-uniform: the block's descriptors come by scalar loads)
    TSTAMP(0);
    if (blk >= n_slots) break; // every wave of the team reaches this; the other teams drain on their own
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const uint16_t *tab = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (A.tab_dim[q] == D) tab = A.tab[q];
    if (D < 1 || D > FAST_MAX_DIM || tab == nullptr) { // host promised D <= 1024 and listed dims
      if (tid == 0) A.out_K[blk] = -1;
      continue;
    }
    const int Dp = (D + 3) & ~3;            // row stride of the proposal table
    const int NG = (D + 255) >> 8;          // 1..4 dim groups
    const int NSW = NWT / NG;               // sample stripes
    const bool active = wave < NG * NSW;
    const int g = wave % NG, sw = wave / NG;
    const int d0 = g * 256 + lane * 4;
    const bool inplace = NSW == 1;          // one wave per dim group: nobody else reads the beams it overwrites

    // ---- my 4 dims (split == gather through perm) and the block's KL ----
    float c[4];
    bool valid[4];
    float own_dmu[4], own_vq[4], own_vp[4];
    const bool own_stats = active && sw == 0 && TEAMS < 3;   // (three teams: 168 VGPRs, the slab keeps the statistics)
    double klacc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = d0 + i;
      valid[i] = d < D;
      c[i] = 0.f;
      const int64_t ixi = valid[i] ? src_index(A, base, pos, d) : src_index(A, base, pos, 0);
      float st3[3] = {0.f, 1.f, 1.f};
      if (valid[i] && active && sw == 0) { // one wave per dim group does the float64 KL and publishes the statistics
        const float mq_ = A.q_loc[ixi], sq_ = A.q_scale[ixi], mp_ = A.p_loc[ixi], sp_ = A.p_scale[ixi];
        klacc = klacc + kl_dim(mq_, sq_, mp_, sp_);
        st3[0] = mq_ - mp_; st3[1] = sq_ * sq_; st3[2] = sp_ * sp_;
      }
      own_dmu[i] = st3[0]; own_vq[i] = st3[1]; own_vp[i] = st3[2];
      if (active && sw == 0 && (NSW > 1 || TEAMS >= 3)) { // somebody else (or a later step) needs them
        stats_g[d0 + i] = st3[0]; stats_g[FAST_MAX_DIM + d0 + i] = st3[1]; stats_g[2 * FAST_MAX_DIM + d0 + i] = st3[2];
      }
    }
    {
      const double gs = wave_tree_sum_valu(klacc);   // (the lane tree on DPP / permlane swaps: +1.5 % against six ds_bpermute trips, r06p)
      if (sw == 0 && active && lane == 0) sm->gpart[g] = gs;
      tsync();
    }
    int K;
    {                                            // every wave forms K itself: no second barrier
      double tot = sm->gpart[0];
      for (int gg = 1; gg < NG; ++gg) tot = tot + sm->gpart[gg];
      K = __builtin_amdgcn_readfirstlane(num_aux((float)tot, A.omega));
      if (tid == 0) A.out_K[blk] = K;
    }
    if (K > A.max_K || K > A.K_limit) continue;
    if (K > A.K_tab) { // beyond the table window: the fused-Philox pass codes it
      if (tid == 0) atomicAdd(A.defer_count, 1u);
      continue;
    }
    if (K == 0) { // nothing to code: sample = p.loc
      if (active && sw == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid[i]) { const int64_t ixo = src_index(A, base, pos, d0 + i); A.out_sample[ixo] = 0.f + A.p_loc[ixo]; }
      }
      continue;
    }

    float sa[4], cH[4];
    f2t G2[NP][4];                                   // G of the beams 2 k, 2 k + 1: the operands of one v_pk_fma_f32
    auto step_consts = [&](int t_next, float (&m)[4], float (&cA)[4], float (&cBv)[4]) {
      const float rho = A.rho[K - 1 - t_next];
      float dmu_[4], vq_[4], vp_[4];
      if (own_stats) { // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) { dmu_[i] = own_dmu[i]; vq_[i] = own_vq[i]; vp_[i] = own_vp[i]; }
      } else {
        const float4 q0 = *reinterpret_cast<const float4 *>(stats_g + d0);
        const float4 q1 = *reinterpret_cast<const float4 *>(stats_g + FAST_MAX_DIM + d0);
        const float4 q2 = *reinterpret_cast<const float4 *>(stats_g + 2 * FAST_MAX_DIM + d0);
        dmu_[0] = q0.x; dmu_[1] = q0.y; dmu_[2] = q0.z; dmu_[3] = q0.w;
        vq_[0] = q1.x; vq_[1] = q1.y; vq_[2] = q1.z; vq_[3] = q1.w;
        vp_[0] = q2.x; vp_[1] = q2.y; vp_[2] = q2.z; vp_[3] = q2.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const StepConst sc = step_constants(rho, dmu_[i], vq_[i], vp_[i], c[i]);
        sa[i] = valid[i] ? sc.sa : 0.f; cH[i] = valid[i] ? sc.H : 0.f;
        m[i] = valid[i] ? sc.m : 0.f; cA[i] = valid[i] ? sc.A : 0.f; cBv[i] = valid[i] ? sc.Bv : 0.f;
        c[i] = c[i] + sc.a; // cumulative_auxiliary_variance += auxiliary_var (:109)
        __builtin_amdgcn_sched_barrier(0); // one dim at a time: the division sequences are register hungry (letting the four chains
                                           // interleave changes nothing: r06i -- the phase is bound by VALU issue, not by the chain)
      }
    };
    // ---- prologue: step 0 has one (all-zero) beam ----
    {
      float m[4], cA[4], cBv[4];
      step_consts(0, m, cA, cBv);
      float cacc = 0.f;
#pragma unroll
      for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) G2[k][i] = (f2t){0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        G2[0][i].x = beam_G(0.f, m[i], cA[i], cBv[i], sa[i]);
        cacc = beam_C_term(cacc, 0.f, m[i], cA[i], cBv[i]);
      }
      const float cg = wave_tree_sum_valu(cacc);
      if (active && sw == 0 && lane == 0) sm->cpart[0][0][g] = cg;
      // (visibility of the C_b partials: the barrier after scoring)
    }

    TSTAMP(1);
    int cur = 0, Bcur = 1;
    // lane j = beam j: the running int32 sum of simple_hash (beam_search_coder.py:33-35) and 4 * dlog(hash) -- every wave carries its own copy
    int32_t hs_cur = 0;
    uint32_t bv_cur = 0u;                                            // hash of the empty path is 1 = g^0
    for (int t = 0; t < K; ++t) {
#ifdef IREC_TEAM_STAMPS
      st_acc[11] += 1ull;
#endif
      const int par = t & 1;
      float (*part)[4] = sm->part[par];
      const uint16_t *tab_tu = tab + (size_t)t * S * Dp;             // uniform base of this step's rows
      // my quad inside a row; lanes past the padded row end read the row's LAST quad (irec_team.hip: same addresses as the last
      // real lane of their 32-lane group, no extra bank conflict)
      const uint32_t tab_lo = (uint32_t)(d0 < Dp ? d0 : Dp - 4);
      uint32_t bet[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) bet[b] = (uint32_t)__builtin_amdgcn_readlane((int)bv_cur, b < Bcur ? b : 0);   // dead slots: beam 0's (phantoms)
      // Look-up addresses are formed two at a time by ONE v_pk_add_f32 on their BIT PATTERNS: byte addresses below 2^23 read as float32 are
      // denormals (value = bits x 2^-149), the kernel runs with denormals preserved (the arithmetic contract needs them anyway), and
      // the sum of two such numbers is exact and carries the integer sum in its bits -- half the address instructions of a v_add_u32 each
      // (r06: a 10 x 20 step is bound by VALU issue, not by the gather pipe).
      f2t betf[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) betf[k] = (f2t){__uint_as_float(bet[2 * k]), __uint_as_float(bet[2 * k + 1])};
      const int N = S * Bcur;
      const int n_mine = S > sw ? (S - sw + NSW - 1) / NSW : 0;     // my samples: sw, sw + NSW, ...
      // ---------------- scoring: S samples x Bcur candidates (beam_search_coder.py:80-84) ----------------
      if (active && Bcur > 1 && !(IREC_TEN_ABLATE & 16)) {
        // software pipelined by dim slot as encode_team_kernel's steady state (irec_team.hip): the ten gathers of the NEXT slot are in
        // flight under the current slot's fma; a chunk is two samples x ten beam slots = 20 accumulators, one reduce_scatter_20.
        // Beam slots beyond Bcur are PHANTOMS (G = 0, beam 0's offset): finite values no candidate reads.
        const int n_chunks = (n_mine + 1) / 2;
        auto row = [&](int m) {                                     // proposal row of my m-th sample; zero row past the end:
          uint2 r = make_uint2(0u, 0u);                             // entry 0 is a valid address, its results are dropped
          if (m < n_mine) r = *reinterpret_cast<const uint2 *>(tab_tu + ((uint32_t)(m * NSW + sw) * (uint32_t)Dp + tab_lo));
          return r;
        };
        uint2 ap_cur[2], ap_nxt[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { ap_cur[cc] = row(cc); ap_nxt[cc] = row(2 + cc); }
#define TEN_AL(CC, I) ((((I) & 2) ? (((I) & 1) ? (ap_cur[CC].y >> 16) : (ap_cur[CC].y & 0xFFFFu)) : (((I) & 1) ? (ap_cur[CC].x >> 16) : (ap_cur[CC].x & 0xFFFFu))) << 2)
        f2t zz[2][NP];
#define TEN_ISSUE(Z, AD) do { const float adf_ = __uint_as_float(AD); const f2t ad2_ = {adf_, adf_}; \
                              _Pragma("unroll") for (int k = 0; k < NP; ++k) { const f2t a2_ = ad2_ + betf[k]; \
                                Z[k].x = lds_abs_f32(__float_as_uint(a2_.x)); Z[k].y = lds_abs_f32(__float_as_uint(a2_.y)); } \
                              __builtin_amdgcn_sched_barrier(0); } while (0)
#define TEN_CONSUME(Z, I, ACC) do { _Pragma("unroll") for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(Z[k])); \
                                f2t t2_[NP]; \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) { \
                                  const f2t h2 = {cH[I], cH[I]}; \
                                  t2_[k] = __builtin_elementwise_fma(h2, Z[k], G2[k][I]); } \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) \
                                  ACC[k] = __builtin_elementwise_fma(t2_[k], Z[k], ACC[k]); \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(ACC[k])); \
                                __builtin_amdgcn_sched_barrier(0); } while (0)
        TEN_ISSUE(zz[0], TEN_AL(0, 0));
        for (int ch = 0; ch < n_chunks; ++ch) {
          f2t acc2[2][NP];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int k = 0; k < NP; ++k) acc2[cc][k] = (f2t){0.f, 0.f};
          uint2 ap_new[2];                                          // rows of the chunk after next: a whole chunk of lead
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) ap_new[cc] = row((ch + 2) * 2 + cc);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (q + 1 < 8) {
              TEN_ISSUE(zz[(q + 1) & 1], TEN_AL((q + 1) >> 2, (q + 1) & 3));
            } else {
#pragma unroll
              for (int cc = 0; cc < 2; ++cc) { ap_cur[cc] = ap_nxt[cc]; ap_nxt[cc] = ap_new[cc]; }
              TEN_ISSUE(zz[0], TEN_AL(0, 0));
            }
            TEN_CONSUME(zz[q & 1], q & 3, acc2[q >> 2]);
          }
          rs_f2 a20[10];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int k = 0; k < NP; ++k) a20[cc * NP + k] = acc2[cc][k];
          const float tot = reduce_scatter_20(a20, lane);
          const int cc = rs_p20 / NB, b = rs_p20 - cc * NB;        // rs_p20 < 0: unused slot
          const int m = ch * 2 + cc;                                // my m-th sample
          if (rs_p20 >= 0 && (lane & 1) == 0 && m < n_mine && b < Bcur) part[(m * NSW + sw) * NB + b][g] = tot;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(zz[0][k])); // drain the look-ups issued past the last sample
#undef TEN_ISSUE
#undef TEN_CONSUME
#undef TEN_AL
      } else if (active) {
        // First step (beam_search_coder.py:97-106): ONE beam, so a sample is a single candidate; 20 SAMPLES share a reduce-scatter -- a
        // sample sits where a beam slot sits in the steady state, every total comes out of the same lane chain and lane tree.
        const f2t bet0f = {__uint_as_float(bet[0]), __uint_as_float(bet[0])};
        for (int m0 = 0; m0 < n_mine; m0 += 20) {
          f2t acc2[10];
#pragma unroll
          for (int p = 0; p < 10; ++p) acc2[p] = (f2t){0.f, 0.f};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            uint2 ap[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) {
              int m = m0 + h * 10 + k;                              // past my last sample: the last row again, the total is dropped
              m = m < n_mine ? m : n_mine - 1;
              ap[k] = *reinterpret_cast<const uint2 *>(tab_tu + ((uint32_t)(m * NSW + sw) * (uint32_t)Dp + tab_lo));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float z[10];
#pragma unroll
              for (int k = 0; k < 10; k += 2) {
                const uint32_t w0 = (i & 2) ? ap[k].y : ap[k].x, w1 = (i & 2) ? ap[k + 1].y : ap[k + 1].x;
                const f2t a2 = (f2t){__uint_as_float(((i & 1) ? (w0 >> 16) : (w0 & 0xFFFFu)) << 2), __uint_as_float(((i & 1) ? (w1 >> 16) : (w1 & 0xFFFFu)) << 2)} + bet0f;
                z[k] = lds_abs_f32(__float_as_uint(a2.x)); z[k + 1] = lds_abs_f32(__float_as_uint(a2.y));
              }
              const f2t h2 = {cH[i], cH[i]}, g2 = {G2[0][i].x, G2[0][i].x};
#pragma unroll
              for (int k = 0; k < 5; ++k) {
                const f2t z2 = {z[2 * k], z[2 * k + 1]};
                acc2[h * 5 + k] = __builtin_elementwise_fma(__builtin_elementwise_fma(h2, z2, g2), z2, acc2[h * 5 + k]);   // proposal_term, two samples
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          rs_f2 a20[10];
#pragma unroll
          for (int p = 0; p < 10; ++p) a20[p] = acc2[p];
          const float tot = reduce_scatter_20(a20, lane);
          const int m = m0 + rs_p20;                                // rs_p20 < 0: unused slot
          if (rs_p20 >= 0 && (lane & 1) == 0 && m < n_mine) part[(m * NSW + sw) * NB][g] = tot;
        }
      }
      TSTAMP(2);
      tsync();   // the step's ONE barrier: every partial (and the C_b partials of the previous update) is in place
      TSTAMP(3);
      const int Bnew = B < N ? B : N;
      const bool last = (t == K - 1);
      uint32_t bv_new = 0u;
      int32_t hs_new = 0;
      if (active) {
        __builtin_amdgcn_s_setprio(2); // serial phase: ahead of the other teams' scoring waves
        // ---------------- combine dim groups in order, add C_b, sort keys: four candidates per lane ----------------
        uint32_t key[4];
        {
          const uint32_t inv = (65536u + (uint32_t)Bcur - 1u) / (uint32_t)Bcur;   // f / Bcur == (f * inv) >> 16 for f < 256, Bcur <= 10
          float4 pr[4], cb4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int f = q * 64 + lane, fs = f < N ? f : 0;
            const int s = (int)(((uint32_t)fs * inv) >> 16), b = fs - s * Bcur;
            pr[q] = *reinterpret_cast<const float4 *>(part[s * NB + b]);
            cb4[q] = *reinterpret_cast<const float4 *>(sm->cpart[par][b]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float sc = pr[q].x;
            if (NG > 1) sc = sc + pr[q].y;
            if (NG > 2) sc = sc + pr[q].z;
            if (NG > 3) sc = sc + pr[q].w;
            float cb = cb4[q].x;                        // C_b: its dim-group partials in order, as the scores'
            if (NG > 1) cb = cb + cb4[q].y;
            if (NG > 2) cb = cb + cb4[q].z;
            if (NG > 3) cb = cb + cb4[q].w;
            key[q] = (q * 64 + lane) < N ? score_key(sc + cb) : 0u;
          }
        }
        TSTAMP(4);
        // ---------------- top-B (beam_search_coder.py:85-89), in this wave's registers ----------------
#if (IREC_TEN_ABLATE & 1)
        const uint32_t sel = (uint32_t)lane + (key[0] & 1u);
#else
        const uint32_t sel = ten_select_fast(key, N, Bnew, lane, [&]() { TSTAMP(7); });   // lane r < Bnew: flat index of rank r
#endif
        int32_t v_sp, v_bp;
        {
          const uint32_t inv = (65536u + (uint32_t)Bcur - 1u) / (uint32_t)Bcur;
          const uint32_t fl = lane < Bnew ? sel : 0u;
          v_sp = (int32_t)((fl * inv) >> 16);                       // best_ind_aux  (:89)
          v_bp = (int32_t)fl - v_sp * Bcur;                         // best_ind_beam (:88)
          const int32_t hp = __builtin_amdgcn_ds_bpermute(v_bp << 2, hs_cur);   // the parent's hash sum: lane b's (LDS crossbar, no bank)
          hs_new = (int32_t)((uint32_t)hp + (uint32_t)v_sp * (uint32_t)(69 + t));   // (:33-35, int32 wrap-around)
          if (wave == 0 && lane < Bnew) bp[(size_t)t * NB + lane] = (v_sp << 6) | v_bp;
        }
        TSTAMP(5);
        // ---------------- gather the surviving beams (beam_search_coder.py:92-93), prepare the next step ----------------
        // new beams' table offsets: lane j looks up dlog(hash(path_j)) -- a global load, consumed at the end of the update
        bv_new = dlog_s[hash_from_sum(lane < Bnew ? hs_new : 0) - 1u];
        const f2t sa_t2[2] = {{sa[0], sa[1]}, {sa[2], sa[3]}};   // this step's sample scale (dims in pairs: the update runs on v_pk_*_f32)
        const int nxt = inplace ? cur : (cur ^ 1);
        // addresses: a wave-uniform base (scalar arithmetic on the selected sample / parent) plus my quad's byte offset
        const char *rows_u = reinterpret_cast<const char *>(tab_tu);
        const uint32_t row_lane = tab_lo * 2u, beam_lane = (uint32_t)d0 * 4u;
        const char *bold_u = reinterpret_cast<const char *>(beams_g + ((size_t)cur * NB) * FAST_MAX_DIM);
        char *bnew_u = reinterpret_cast<char *>(beams_g + ((size_t)nxt * NB) * FAST_MAX_DIM);
        // the last step keeps ONE beam: beams[0] is all that leaves the block (beam_search_coder.py:118-122)
        const int Bupd = last ? 1 : Bnew;
        // ---- every global read of the update back to back: proposal rows, parents ----
        uint2 apv[NB];
        float4 obv4[NB];
        uint32_t bet_old[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          apv[j] = make_uint2(0u, 0u);
          obv4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          bet_old[j] = 0u;
          if (j < Bupd) { // wave-uniform
            const uint32_t sp_ = (uint32_t)__builtin_amdgcn_readlane(v_sp, j);
            const uint32_t bp_ = (uint32_t)__builtin_amdgcn_readlane(v_bp, j);
            bet_old[j] = (uint32_t)__builtin_amdgcn_readlane((int)bv_cur, (int)bp_);
            apv[j] = *reinterpret_cast<const uint2 *>(rows_u + (size_t)(sp_ * (uint32_t)Dp * 2u) + row_lane);
            if (t) obv4[j] = *reinterpret_cast<const float4 *>(bold_u + (size_t)(bp_ * (uint32_t)(FAST_MAX_DIM * 4)) + beam_lane);
          }
        }
        float m[4], cA[4], cBv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = 0.f; cA[i] = 0.f; cBv[i] = 0.f; }
#if (IREC_TEN_ABLATE & 2)
        if (!last) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { m[i] = sa[i]; cA[i] = cH[i]; cBv[i] = c[i]; }
        }
#else
        if (!last) step_consts(t + 1, m, cA, cBv); // next step's constants, under the loads' latency
#endif
        TSTAMP(9);
        const f2t m2[2] = {{m[0], m[1]}, {m[2], m[3]}}, A2[2] = {{cA[0], cA[1]}, {cA[2], cA[3]}}, Bv2[2] = {{cBv[0], cBv[1]}, {cBv[2], cBv[3]}};
        const f2t AA2[2] = {A2[0] + A2[0], A2[1] + A2[1]};          // (beam_G: A + A)
        float cacc[rsn_room(NB)];
#pragma unroll
        for (int j = 0; j < rsn_room(NB); ++j) cacc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) G2[k][i] = (f2t){0.f, 0.f};
        // look-ups of five beams are issued back to back, then consumed
        constexpr int YB = 5;
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += YB) {
          f2t zy[YB][2];
#pragma unroll
          for (int v = 0; v < YB; ++v) {
            const int j = j0 + v;
            const uint32_t al[4] = {(apv[j].x & 0xFFFFu) << 2, (apv[j].x >> 16) << 2, (apv[j].y & 0xFFFFu) << 2, (apv[j].y >> 16) << 2};
            const f2t bo2 = {__uint_as_float(bet_old[j]), __uint_as_float(bet_old[j])};
#pragma unroll
            for (int h = 0; h < 2; ++h) {                             // entry 0 for beams that do not exist
              const f2t a2 = (f2t){__uint_as_float(al[2 * h]), __uint_as_float(al[2 * h + 1])} + bo2;
              zy[v][h] = (f2t){lds_abs_f32(__float_as_uint(a2.x)), lds_abs_f32(__float_as_uint(a2.y))};
            }
          }
#pragma unroll
          for (int v = 0; v < YB; ++v) {
            const int j = j0 + v;
            if (j < Bupd) { // wave-uniform
              const f2t obv[2] = {{obv4[j].x, obv4[j].y}, {obv4[j].z, obv4[j].w}};
              f2t nb[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f2t y = sa_t2[h] * zy[v][h];                          // dist.quantile(.), :48-49
                nb[h] = obv[h] + y;                                         // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
              }
              if (last) {
                if (sw == 0) {                                              // (j == 0: Bupd == 1)
                  const float nbs[4] = {nb[0].x, nb[0].y, nb[1].x, nb[1].y};
#pragma unroll
                  for (int i = 0; i < 4; ++i)
                    if (valid[i]) { // beams[0] + coding_dist.loc, :122
                      const int64_t ixo = src_index(A, base, pos, d0 + i);
                      A.out_sample[ixo] = nbs[i] + A.p_loc[ixo];
                    }
                }
              } else {
                if (sw == 0) *reinterpret_cast<float4 *>(bnew_u + (size_t)(j * FAST_MAX_DIM * 4) + beam_lane) = make_float4(nb[0].x, nb[0].y, nb[1].x, nb[1].y);
                // beam_G / beam_C_term (irec_device.h) on dim pairs: p = beam - m; G = fma(A + A, p, Bv) * sa; C += fma(fma(A, p, Bv), p, .) in dim order
                f2t u2[2], t2[2], p2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  p2[h] = nb[h] - m2[h];
                  t2[h] = __builtin_elementwise_fma(AA2[h], p2[h], Bv2[h]);
                  u2[h] = __builtin_elementwise_fma(A2[h], p2[h], Bv2[h]);
                }
                const float gv[4] = {t2[0].x * sa[0], t2[0].y * sa[1], t2[1].x * sa[2], t2[1].y * sa[3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) { if (j & 1) G2[j >> 1][i].y = gv[i]; else G2[j >> 1][i].x = gv[i]; }
                float ca = cacc[j];
                ca = fmaf(u2[0].x, p2[0].x, ca); ca = fmaf(u2[0].y, p2[0].y, ca); ca = fmaf(u2[1].x, p2[1].x, ca); ca = fmaf(u2[1].y, p2[1].y, ca);
                cacc[j] = ca;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        TSTAMP(10);
        if (!last) {
          const float ctot = reduce_scatter_n<NB>(cacc, lane);
          if (sw == 0 && (lane & 1) == 0 && rs_c >= 0 && rs_c < Bnew) sm->cpart[par ^ 1][rs_c][g] = ctot;
        }
        TSTAMP(6);
        __builtin_amdgcn_s_setprio(0);
        cur = nxt;
      }
      bv_cur = bv_new;
      hs_cur = hs_new;
      // no barrier here: the new C_b partials are read behind the next step's barrier, the partials and C_b of the two parities
      // never meet, beams are read by the waves that wrote them (or, with sample stripes, from the other buffer)
      Bcur = Bnew;
    }
    // ---- index path of beam 0 (beam_search_coder.py:118-121) ----
    tsync();   // (sample stripes: their last reads of the step's partials; everybody: bp of wave 0 is in place)
    if (tid == 0) {
      int j = 0;
      for (int t = K - 1; t >= 0; --t) {
        const int32_t v = __builtin_nontemporal_load(&bp[(size_t)t * NB + j]);
        A.out_indices[blk * (int64_t)A.max_K + t] = v >> 6;
        j = v & 63;
      }
    }
    TSTAMP(8);
  }
#ifdef IREC_TEAM_STAMPS
  if (A.dbg && lane == 0)
    for (int k = 0; k < 12; ++k) A.dbg[((size_t)blockIdx.x * (TEAMS * NWT) + wave_wg) * 16 + k] = st_acc[k];
  if (A.dbg && lane == 0) {
    A.dbg[((size_t)blockIdx.x * (TEAMS * NWT) + wave_wg) * 16 + 12] = __builtin_amdgcn_s_memtime() - st_t0;
    A.dbg[((size_t)blockIdx.x * (TEAMS * NWT) + wave_wg) * 16 + 13] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
#undef TSTAMP
}

// ---- launcher ----
bool ten_applies(int B, int S) { return B >= 2 && B <= TEN_NB && S >= 1 && (int64_t)S * TEN_NB <= TEN_MAXC; }
size_t ten_lds_for(int teams) { return ten_lds_total(teams); }
template <int TEAMS>
static hipError_t launch_ten_t(const EncArgs &A, int grid, hipStream_t st) {
  const size_t lds = ten_lds_total(TEAMS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_ten_kernel<TEAMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((encode_ten_kernel<TEAMS>), dim3(grid), dim3(TEAMS * TEAM_NT), lds, st, A);
  return hipGetLastError();
}
hipError_t launch_encode_ten(const EncArgs &A, int teams, int grid, hipStream_t st) {
  if (!ten_applies(A.B, A.S) || A.coop_W > 1 || A.out_margin != nullptr) return hipErrorInvalidValue;
  switch (teams) {
    case 3: return launch_ten_t<3>(A, grid, st);
    case 2: return launch_ten_t<2>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}

} // namespace irec
