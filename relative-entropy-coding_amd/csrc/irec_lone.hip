// irec_lone.hip -- one-beam calls (n_beams = 1 of the reference's sweep, examples/lossless/data_aggregation.py:7): ONE WAVE
// per block, no barrier after the table fill.
//
// Hot path (reference file:line): BeamSearchCoder.encode_block rec/coding/beam_search_coder.py:53-122 with n_beams = 1;
// arithmetic specification DESIGN.md §3 -- every emitted bit is the team encoder's (irec_team.hip).
//
// Why its own kernel.  With one beam a step of a block is S * D look-ups (7 000 at S = 7) behind ~110 VALU operations per dim of
// IEEE step constants, a selection that is an arg-max and an update of one beam: the team encoder pays four team barriers,
// a one-wave selection and slab round trips per step for it (~26 k cycles per step at S = 7, three steps in flight per CU).
// Here the whole beam lives in ONE wave's registers -- lane l owns dims 256 g + 4 l .. + 3 of all four dim groups g, exactly
// the lanes the canonical reduction tree gives them -- so a step needs no exchange with any other wave:
//   * 12 waves per CU, each coding its own block (block counter pulls per wave), three quantile-table copies shared in LDS
//     and the team encoder's proposal tables with their bank-spreading copy bits (choice_table_rows, irec_team.hip): a ds_read_b32 still
//     gathers dims 256 g + 4 l + i of one (g, i) over the lanes, the grouping the copy bits were chosen for;
//   * scores of four samples x four dim groups leave one 16-value reduce-scatter (lane distances 32 .. 1, the canonical
//     tree); the four group totals of a sample then sit 4 lanes apart in one row and are added in group order by DPP;
//   * selection = running (key, sample) maximum per lane + one 64-bit wave maximum; ties go to the lower sample index
//     (beam_search_coder.py:85-89 with one beam: flat index = sample index);
//   * the block's statistics are parked in a 12 KB slab per wave (L2), everything else stays in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"

namespace irec {

constexpr int LONE_NWV = 12;                                        // waves per workgroup (one workgroup per CU): 168 VGPRs each
constexpr size_t LONE_T3_BYTES = (((size_t)3 * IREC_PM1) * 4 + 15) & ~(size_t)15;
constexpr size_t LONE_SLAB_BYTES = (size_t)3 * FAST_MAX_DIM * 4;    // per wave: mq - mp, sq^2, sp^2 of its block, [3][1024] f32

#ifdef IREC_LONE_STAMPS   // diagnostic build: per-wave cycle sums per phase (0 prologue, 1 constants, 2 scoring, 3 selection, 4 update, 5 epilogue)
#define LSTAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (A.dbg && (threadIdx.x & 63) == 0) A.dbg[((size_t)blockIdx.x * LONE_NWV + (threadIdx.x >> 6)) * 16 + (slot)] += now_ - *lst_prev; *lst_prev = now_; } while (0)
#else
#define LSTAMP(slot) do { } while (0)
#endif
typedef float lone_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int lone_u2 __attribute__((ext_vector_type(2)));

// value of lane + N of the same row of 16 (N = 4, 8, 12); only lanes 0..3 of a row use the result
template <int N>
__device__ __forceinline__ float row_ahead(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x100 + N, 0xF, 0xF, true));   // row_shl:N
}

// Steps of one block whose K, statistics slab and proposal table are in place.  NGX = dim groups the wave carries (2 or 4:
// pairs of groups share a v_pk_fma_f32, so a block of one or three groups carries a zero-coefficient partner).
template <int NGX>
__device__ __forceinline__ void lone_code_block(const EncArgs &A, const int64_t blk, const int D, const int64_t base, const int32_t pos,
                                                const int K, const uint16_t *tab, float *stats_g, const int lane, unsigned long long *lst_prev) {
  (void)lst_prev;
  constexpr int NP = NGX / 2;
  const int S = A.S;
  const int Dp = (D + 3) & ~3;            // row stride of the proposal table
  const int NG = (D + 255) >> 8;          // 1..4 dim groups
  const bool score_lane = (lane & 12) == 0;            // holds a sample's group-0 total; groups 1..3 are 4, 8, 12 lanes ahead
  const int my_cc = lane >> 4;                         // that sample's place in its chunk of four
  // my quad inside a row, per dim group (bytes).  Lanes past the padded row end (all their dims invalid, zero coefficients)
  // read the row's LAST quad: finite z, and addresses the LDS serves as a broadcast with the last real lane's
  uint32_t roff[NGX];
#pragma unroll
  for (int g = 0; g < NGX; ++g) {
    const int d0 = g * 256 + lane * 4;
    roff[g] = 2u * (uint32_t)(d0 < Dp ? d0 : Dp - 4);
  }
  float c[NGX][4], beam[NGX][4], sa[NGX][4];
  lone_f2 Gp[NP][4], Hp[NP][4];           // pair gp = dim groups 2 gp, 2 gp + 1: two independent chains per v_pk_fma_f32
#pragma unroll
  for (int g = 0; g < NGX; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[g][i] = 0.f; beam[g][i] = 0.f; sa[g][i] = 0.f; }
  uint32_t hsum = 0u, bet = 0u;           // hash of the empty path is 1 = g^0
  const uint64_t tab_u = (uint64_t)(uintptr_t)tab;
  const __amdgpu_buffer_rsrc_t tab_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tab_u >> 32)) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tab_u)),
      (short)0, (int)0x7FFFFFFF, 0x00020000);

  LSTAMP(0);
  for (int t = 0; t < K; ++t) {
    // ---- step constants of my dims (beam_search_coder.py:67-77), G / H / C_b of the one beam ----
    const float rho = A.rho[K - 1 - t];
    float cv[NGX];                          // C_b partials of my lane, per dim group
#pragma unroll
    for (int g = 0; g < NGX; ++g) {
      float cacc = 0.f;
      float Gg[4] = {0.f, 0.f, 0.f, 0.f}, Hg[4] = {0.f, 0.f, 0.f, 0.f};
      if (g < NG) { // wave-uniform
        const int d0 = g * 256 + lane * 4;
        asm volatile("" ::: "memory");   // (the slab was written by this wave; a real reload every step)
        const float4 q0 = *reinterpret_cast<const float4 *>(stats_g + d0);
        const float4 q1 = *reinterpret_cast<const float4 *>(stats_g + FAST_MAX_DIM + d0);
        const float4 q2 = *reinterpret_cast<const float4 *>(stats_g + 2 * FAST_MAX_DIM + d0);
        const float dmu_[4] = {q0.x, q0.y, q0.z, q0.w}, vq_[4] = {q1.x, q1.y, q1.z, q1.w}, vp_[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool valid = d0 + i < D;
          const StepConst sc = step_constants(rho, dmu_[i], vq_[i], vp_[i], c[g][i]);
          const float sa_ = valid ? sc.sa : 0.f, H_ = valid ? sc.H : 0.f;
          const float m_ = valid ? sc.m : 0.f, A_ = valid ? sc.A : 0.f, Bv_ = valid ? sc.Bv : 0.f;
          c[g][i] = c[g][i] + sc.a;                                   // cumulative_auxiliary_variance += auxiliary_var (:109)
          sa[g][i] = sa_;
          Hg[i] = H_;
          Gg[i] = beam_G(beam[g][i], m_, A_, Bv_, sa_);
          cacc = beam_C_term(cacc, beam[g][i], m_, A_, Bv_);
          __builtin_amdgcn_sched_barrier(0); // one dim at a time: the division sequences are register hungry
        }
      }
      cv[g] = cacc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (g & 1) { Gp[g >> 1][i].y = Gg[i]; Hp[g >> 1][i].y = Hg[i]; }
        else { Gp[g >> 1][i].x = Gg[i]; Hp[g >> 1][i].x = Hg[i]; }
      }
    }
    // C_b: every group's 64 lane partials through the canonical tree (one reduce-scatter on the VALU: the group totals land
    // in lanes 0, 16, 32, 48 -- lanes 0, 32 for two groups --), then the dim-group totals in increasing order, as the scores'
    float cb;
    {
      const float mine = reduce_scatter_n<NGX>(cv, lane);
      constexpr int LSTEP = 64 / NGX;
      cb = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mine), 0));
#pragma unroll
      for (int g = 1; g < NGX; ++g)
        if (g < NG) cb = cb + __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mine), g * LSTEP));
    }

    LSTAMP(1);
    // ---- scoring (beam_search_coder.py:80-84): S candidates, four per reduce-scatter ----
    // rows through a buffer descriptor: the wave-uniform row offset travels in soffset, my quad's offset in voffset -- no vector
    // address arithmetic per row (the flat form costs two VALU operations per load here)
    const uint32_t tab_t = (uint32_t)t * (uint32_t)S * (uint32_t)Dp * 2u;         // (tables are bounded by IREC_TABLE_BYTES_HARD = 1 GB)
    auto rows = [&](int s, uint2 (&r)[NGX]) {
      const uint32_t rowb = tab_t + (uint32_t)s * (uint32_t)Dp * 2u;               // wave-uniform
#pragma unroll
      for (int g = 0; g < NGX; ++g) {
        const lone_u2 q = __builtin_amdgcn_raw_buffer_load_b64(tab_rsrc, (int)roff[g], (int)rowb, 0);
        r[g] = make_uint2(q.x, q.y);
      }
    };
    // byte address of dim slot i's entry in copy 0 plus the beam's rotation: ONE v_mad_u32_u16 (16-bit half of the packed
    // quad x 4 + bet; op_sel picks the high half) instead of a bit-field extract and a shift-add
    auto addr = [&](const uint2 &q, int i) -> uint32_t {
      const uint32_t w = (i & 2) ? q.y : q.x;
      uint32_t a;
      if (i & 1) asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "v"(bet));
      else asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a) : "v"(w), "v"(bet));
      return a;
    };
    // Software pipeline by sample: the 4 NGX look-ups of sample s + 1 are issued before the values of sample s are consumed
    // (two look-up buffers), the rows of sample s + 2 are in flight under both.
    auto issue = [&](const uint2 (&r)[NGX], lone_f2 (&z)[4][NP]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int gp = 0; gp < NP; ++gp) { z[i][gp].x = lds_abs_f32(addr(r[2 * gp], i)); z[i][gp].y = lds_abs_f32(addr(r[2 * gp + 1], i)); }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto consume = [&](lone_f2 (&z)[4][NP], lone_f2 (&acc)[NP]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int gp = 0; gp < NP; ++gp) asm volatile("" : "+v"(z[i][gp]));
      lone_f2 in[4][NP];
#pragma unroll
      for (int i = 0; i < 4; ++i)                 // the inner fma of every (slot, pair) first: no dependent back-to-back issue
#pragma unroll
        for (int gp = 0; gp < NP; ++gp) in[i][gp] = __builtin_elementwise_fma(Hp[gp][i], z[i][gp], Gp[gp][i]);
#pragma unroll
      for (int i = 0; i < 4; ++i)                 // proposal_term of two dim groups, dim slots chained in order
#pragma unroll
        for (int gp = 0; gp < NP; ++gp) acc[gp] = __builtin_elementwise_fma(in[i][gp], z[i][gp], acc[gp]);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto clamp_s = [&](int s) { return s < S ? s : S - 1; };   // (past the end: the last row again, its totals are never read)
    uint32_t best_k = 0u, best_s = 0u;
    uint2 ra[NGX], rb[NGX];
    lone_f2 za[4][NP], zb[4][NP];
    rows(0, ra);
    rows(clamp_s(1), rb);
    issue(ra, za);
    for (int s0 = 0; s0 < S; s0 += 4) {
      lone_f2 acc2[4][NP];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int gp = 0; gp < NP; ++gp) acc2[cc][gp] = (lone_f2){0.f, 0.f};
      // sample s0 (values in za; rows of s0 + 1 in rb)
      rows(clamp_s(s0 + 2), ra);
      issue(rb, zb);
      consume(za, acc2[0]);
      rows(clamp_s(s0 + 3), rb);
      issue(ra, za);
      consume(zb, acc2[1]);
      rows(clamp_s(s0 + 4), ra);
      issue(rb, zb);
      consume(za, acc2[2]);
      rows(clamp_s(s0 + 5), rb);
      issue(ra, za);                                  // first sample of the next chunk
      consume(zb, acc2[3]);
      float v[16];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        v[4 * cc] = acc2[cc][0].x; v[4 * cc + 1] = acc2[cc][0].y;
        v[4 * cc + 2] = NP > 1 ? acc2[cc][NP - 1].x : 0.f; v[4 * cc + 3] = NP > 1 ? acc2[cc][NP - 1].y : 0.f;
      }
      const float t0 = reduce_scatter_n<16>(v, lane);             // this lane: sample (lane >> 4), dim group (lane >> 2) & 3
      float sc = t0;                                              // dim groups in increasing order
      const float t1 = row_ahead<4>(t0), t2 = row_ahead<8>(t0), t3 = row_ahead<12>(t0);
      if (NG > 1) sc = sc + t1;
      if (NG > 2) sc = sc + t2;
      if (NG > 3) sc = sc + t3;
      const uint32_t key = score_key(sc + cb);
      const uint32_t s_mine = (uint32_t)(s0 + my_cc);
      if (score_lane && s_mine < (uint32_t)S && key > best_k) { best_k = key; best_s = s_mine; }   // (earlier sample wins a tie)
    }
    LSTAMP(2);
    // ---- top-1 (beam_search_coder.py:85-89): value descending, ties to the lower index ----
    const unsigned long long win = wave_max_u64(cand_pack(best_k, best_s));
    const uint32_t s_star = 0xFFFFFFFFu - (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)win);
    if (lane == 0) A.out_indices[blk * (int64_t)A.max_K + t] = (int32_t)s_star;

    LSTAMP(3);
    // ---- the surviving beam (:92-93): beam += sa * z of the chosen sample ----
    {
      uint2 r[NGX];
      rows((int)s_star, r);
#pragma unroll
      for (int g = 0; g < NGX; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = lds_abs_f32(addr(r[g], i));
          const float y = sa[g][i] * z;                           // dist.quantile(.), :48-49   (sa = 0 where there is no dim)
          beam[g][i] = beam[g][i] + y;                            // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
        }
    }
    hsum += s_star * (uint32_t)(69 + t);                          // simple_hash's running int32 sum (:33-35, :94-95)
    bet = (uint32_t)A.dlog4r[hash_from_sum((int32_t)hsum) - 1u];
    LSTAMP(4);
  }
  // ---- beams[0] + coding_dist.loc (:118-122), merge == scatter through perm: positions, then mu_p, in one batch each ----
  int64_t ixo[NGX][4];
#pragma unroll
  for (int g = 0; g < NGX; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = g * 256 + lane * 4 + i;
      ixo[g][i] = src_index(A, base, pos, d < D ? d : 0);
    }
  float plv[NGX][4];
#pragma unroll
  for (int g = 0; g < NGX; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) plv[g][i] = A.p_loc[ixo[g][i]];
#pragma unroll
  for (int g = 0; g < NGX; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (g * 256 + lane * 4 + i < D) A.out_sample[ixo[g][i]] = beam[g][i] + plv[g][i];
}

__global__ __launch_bounds__(LONE_NWV * 64, 1) void encode_lone_kernel(EncArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  commit_table_stamps(A);
  {
    float *l3 = reinterpret_cast<float *>(smem);
    for (int k = (int)threadIdx.x; k < (int)IREC_PM1; k += LONE_NWV * 64) {
      const float v = A.lut2[k];
      l3[k] = v; l3[k + IREC_PM1] = v; l3[k + 2 * IREC_PM1] = v;
    }
  }
  __syncthreads();   // the only barrier: from here on the waves never wait for each other

  float *stats_g = reinterpret_cast<float *>(A.ws + ((size_t)blockIdx.x * LONE_NWV + wave) * LONE_SLAB_BYTES);
  unsigned long long lst_t = __builtin_amdgcn_s_memtime();
  (void)lst_t;
  // Block hand-out (irec_fast_common.h): the first block of wave k of workgroup w is slot k * gridDim.x + w -- one block per
  // CU before any CU gets a second -- with the rows of a tensor dealt to one XCD; later blocks from the XCD's counter.
  const int64_t n_static = (int64_t)LONE_NWV * (int64_t)gridDim.x < A.n_blocks ? (int64_t)LONE_NWV * (int64_t)gridDim.x : A.n_blocks;
  bool first_block = true;
  int steal = 0;
  for (;;) {
    int64_t blk;
    if (first_block) {
      blk = (int64_t)wave * (int64_t)gridDim.x + (int64_t)blockIdx.x;
      if (blk < n_static) blk = xcd_static_row(blk, n_static, (int)gridDim.x);
    } else {
      int64_t v = 0;
      if (lane == 0) v = xcd_pull_row(A, n_static, A.n_blocks, steal);
      blk = ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
    first_block = false;
    if (blk >= A.n_blocks) break;
    const int D = __builtin_amdgcn_readfirstlane(A.block_dim[blk]);
    const int64_t base = A.block_base[blk];
    const int32_t pos = __builtin_amdgcn_readfirstlane(A.block_pos[blk]);
    const uint16_t *tab = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (A.tab_dim[q] == D) tab = A.tab[q];
    if (D < 1 || D > FAST_MAX_DIM || tab == nullptr) {   // host promised D <= 1024 and listed dims
      if (lane == 0) A.out_K[blk] = -1;
      continue;
    }
    const int NG = (D + 255) >> 8;          // 1..4 dim groups

    // ---- the block's statistics (split == gather through perm) and its KL ----
    // All dim groups' loads in two batches -- the 16 positions through perm, then the 64 statistics -- instead of a dependent
    // pair per group: a wave owns the whole block here, and eight dependent random-access round trips in front of every block
    // were 29 % of a wave's time at S = 7 (profiles/archive/r03j/stamps_lone.log).
    double tot = 0.0;
    {
      int64_t ixs[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = g * 256 + lane * 4 + i;
          ixs[g][i] = src_index(A, base, pos, d < D ? d : 0);
        }
      float mqv[4][4], sqv[4][4], mpv[4][4], spv[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          mqv[g][i] = A.q_loc[ixs[g][i]]; sqv[g][i] = A.q_scale[ixs[g][i]];
          mpv[g][i] = A.p_loc[ixs[g][i]]; spv[g][i] = A.p_scale[ixs[g][i]];
        }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < NG) { // wave-uniform
          const int d0 = g * 256 + lane * 4;
          double klacc = 0.0;
          float st[3][4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            st[0][i] = 0.f; st[1][i] = 1.f; st[2][i] = 1.f;
            if (d0 + i < D) {
              klacc = klacc + kl_dim(mqv[g][i], sqv[g][i], mpv[g][i], spv[g][i]);
              st[0][i] = mqv[g][i] - mpv[g][i]; st[1][i] = sqv[g][i] * sqv[g][i]; st[2][i] = spv[g][i] * spv[g][i];
            }
          }
#pragma unroll
          for (int k = 0; k < 3; ++k)
            *reinterpret_cast<float4 *>(stats_g + k * FAST_MAX_DIM + d0) = make_float4(st[k][0], st[k][1], st[k][2], st[k][3]);
          const double gs = wave_tree_sum(klacc);
          tot = g == 0 ? gs : tot + gs;         // dim-group sums in increasing order
        }
      }
    }
    const int32_t K = __builtin_amdgcn_readfirstlane(num_aux((float)tot, A.omega));
    if (lane == 0) A.out_K[blk] = K;
    if (K > A.max_K || K > A.K_limit) continue;
    if (K > A.K_tab) { // beyond the table window: the fused-Philox pass codes it
      if (lane == 0) atomicAdd(A.defer_count, 1u);
      continue;
    }
    if (K == 0) { // nothing to code: sample = p.loc
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = g * 256 + lane * 4 + i;
          if (d < D) { const int64_t ixo = src_index(A, base, pos, d); A.out_sample[ixo] = 0.f + A.p_loc[ixo]; }
        }
      continue;
    }
    if (NG > 2) lone_code_block<4>(A, blk, D, base, pos, K, tab, stats_g, lane, &lst_t);
    else lone_code_block<2>(A, blk, D, base, pos, K, tab, stats_g, lane, &lst_t);
    { unsigned long long *lst_prev = &lst_t; (void)lst_prev; LSTAMP(5); }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool lone_applies(int B, int shape_override) { return B == 1 && shape_override == 0; }
int lone_waves() { return LONE_NWV; }
size_t lone_lds_bytes() { return LONE_T3_BYTES; }
size_t lone_ws_bytes_per_wg() { return (size_t)LONE_NWV * LONE_SLAB_BYTES; }
const char *lone_kernel_name() { return "encode_lone_kernel"; }

hipError_t launch_encode_lone(const EncArgs &A, int grid, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_lone_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)LONE_T3_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encode_lone_kernel, dim3(grid), dim3(LONE_NWV * 64), LONE_T3_BYTES, st, A);
  return hipGetLastError();
}

} // namespace irec
