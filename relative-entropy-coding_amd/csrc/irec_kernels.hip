// irec_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels of the iREC beam-search coder.
//
// Hot path re-implemented here (reference file:line):
//   BeamSearchCoder.encode_block            rec/coding/beam_search_coder.py:53-122
//     get_pseudo_random_sample              rec/coding/beam_search_coder.py:37-51
//     simple_hash                           rec/coding/beam_search_coder.py:33-35
//     get_auxiliary_coder / _target         rec/coding/coder.py:141-154
//     get_auxiliary_ratio                   rec/coding/coder.py:16,218-220
//   (BeamSearchCoder.decode_block, rec/coding/beam_search_coder.py:124-148: irec_decode.hip)
//   Coder.split / merge (as gather/scatter) rec/coding/coder.py:38-122
//
// All encoders share one arithmetic specification; the default one is encode_team_kernel (irec_team.hip).
// Here:
//   encode_fast_kernel<NB,NW,TABLE>  D <= 1024, B <= 32: one persistent workgroup per block, G of every beam in VGPRs
//                              (lane owns 4 dims), ONE copy of the quantile table in LDS; proposals from the per-call
//                              table (TABLE) or from the Philox stream fused into the kernel.  Serves B > 20, large S*B,
//                              calls without dim hints, and IREC_FLAG_ONE_TABLE / IREC_FLAG_FUSED_PHILOX.
//                              In SPLIT mode (calls of few blocks: A.coop_W workgroups per block, exchanging their sort keys
//                              through L2 every step) the workgroups share the block's beams (A.coop_beams: each owns at most
//                              two beam slots, scores every sample for them and forms only its own new beams in the block's
//                              shared slab) or, in the older form, its samples.
//   encode_generic_kernel      any D, B <= 256: beams in a global scratch slab; correctness fallback.
//
// Compiled with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no implicit fma: see irec_device.h).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"

namespace irec {

#ifndef IREC_SPLIT_CONSTS_EARLY
#define IREC_SPLIT_CONSTS_EARLY 0   // beam-split build: 1 = next step's constants between publishing the keys and sweeping the partners'
                                    // (r04 A/B, profiles/archive/r04d: neutral -- the wait is skew between the partners, not hand-off latency); 0: in the update
#endif
#ifndef IREC_SPLIT_NW
#define IREC_SPLIT_NW 0             // waves per workgroup of the beam-split build; 0 = 8 for the 20-beam build (two sample stripes per dim group:
                                    // 9 blocks 0.132 -> 0.127 ms on the same box, r04ao), 4 for the others (13 ten-beam blocks: 0.105 against 0.108 with 8)
#endif
constexpr int split_beam_nw(int NB) { return IREC_SPLIT_NW ? IREC_SPLIT_NW : (NB == 20 ? 8 : 4); }
#define IREC_STAMP(slot)                                                    \
  do {                                                                      \
    if (A.dbg && tid == 0) {                                                \
      const unsigned long long now_ = stamp_now();                          \
      A.dbg[(size_t)blockIdx.x * 16 + (slot)] += now_ - stamp_prev;          \
      stamp_prev = now_;                                                    \
    }                                                                       \
  } while (0)



// ======================================================================================================
//  KL / K kernel  (beam_search_coder.py:57-59)
// ======================================================================================================
__global__ __launch_bounds__(256) void block_kl_kernel(EncArgs A, float *out_kl) {
  __shared__ double gpart[4];
  __shared__ double total_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform by construction: keep it in an SGPR
  for (int64_t blk = blockIdx.x; blk < A.n_blocks; blk += gridDim.x) {
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const int NG = (D + 255) >> 8;
    for (int g0 = 0; g0 < NG; g0 += 4) {
      const int g = g0 + wave;
      double acc = 0.0;
      if (g < NG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = g * 256 + lane * 4 + i;
          if (d < D) {
            const int64_t ix = src_index(A, base, pos, d);
            acc = acc + kl_dim(A.q_loc[ix], A.q_scale[ix], A.p_loc[ix], A.p_scale[ix]);
          }
        }
      }
      const double gs = wave_tree_sum(acc);
      if (lane == 0) gpart[wave] = gs;
      __syncthreads();
      if (tid == 0) {
        double t = g0 == 0 ? 0.0 : total_s;
        for (int w = 0; w < 4 && g0 + w < NG; ++w) t = (g0 + w == 0) ? gpart[w] : t + gpart[w];
        total_s = t;
      }
      __syncthreads();
    }
    if (tid == 0) {
      const float kl = NG ? (float)total_s : 0.0f;
      if (out_kl) out_kl[blk] = kl;
      A.out_K[blk] = num_aux(kl, A.omega);
    }
    __syncthreads();
  }
}

// ======================================================================================================
//  generic encoder: any D, B <= 256 (round 4; 64 until then: the reference's n_beams is any Python int, beam_search_coder.py:28).
//  Scratch slab per workgroup:
//    float dmu,vq,vp,mp,c,sa,m,A,Bv,H [10][Dpad] | float beams[2][B][Dpad] | float G[B][Dpad] | int32 bp[max_K][B] | uint32 key[S*B]
// ======================================================================================================
constexpr int GEN_NT = 256;
constexpr int GEN_CH = 16;    // beams scored together against one sample's draw (one reduce-scatter per dim group; 32: spills, slower)
constexpr int GEN_NSC = 8192; // score/key entries kept in LDS; larger candidate sets go to the scratch slab
constexpr int GEN_MB = 256;   // beams at most (selected beam = thread index: GEN_NT; back-pointers hold the parent in 8 bits)
using GenLds = SmallLdsT<GEN_MB, 32, 512>;   // (more than 64 beams: the threshold selection always leaves more than 64 survivors and falls
                                             //  to the scan, one barrier per selected beam -- slow and correct, as a fallback may be)
constexpr size_t GEN_SMALL_BYTES = (sizeof(GenLds) + 15) & ~(size_t)15;
static_assert(GEN_MB <= GEN_NT, "new beam j is recorded by thread j");

__global__ __launch_bounds__(GEN_NT) void encode_generic_kernel(EncArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *lut_s = reinterpret_cast<float *>(smem);                              // [10008]
  uint32_t *key_lds = reinterpret_cast<uint32_t *>(smem + 40032);              // [GEN_NSC]
  GenLds *sm = reinterpret_cast<GenLds *>(smem + 40032 + GEN_NSC * 4);
  double *gpart = sm->gpart;                                                   // [4]
  double *total_s = reinterpret_cast<double *>(smem + 40032 + GEN_NSC * 4 + GEN_SMALL_BYTES); // [1]
  int32_t *sel_s = sm->sel_s, *sel_b = sm->sel_b;                              // [GEN_MB]
  int32_t *hsum = &sm->hsum[0][0];                                             // [2][GEN_MB]
  int32_t *misc = sm->misc;
  float *Cb_s = reinterpret_cast<float *>(total_s + 1);                        // [GEN_MB] C_b of the live beams

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform by construction: keep it in an SGPR
  const int S = A.S, B = A.B;
  commit_table_stamps(A);
  if (A.deferred_pass && *A.defer_count == 0u) return;   // (uniform over the grid: the first pass deferred nothing -- leave before the table is staged)
  for (int k = tid; k < (int)IREC_P; k += GEN_NT) lut_s[k] = A.lut[k];

  const int Dpad = A.max_dim_pad;
  char *slab = A.ws + (size_t)blockIdx.x * A.ws_per_wg;
  float *f_dmu = reinterpret_cast<float *>(slab);
  float *f_vq = f_dmu + Dpad, *f_vp = f_vq + Dpad, *f_mp = f_vp + Dpad, *f_c = f_mp + Dpad;
  float *f_sa = f_c + Dpad, *f_m = f_sa + Dpad, *f_A = f_m + Dpad, *f_Bv = f_A + Dpad, *f_H = f_Bv + Dpad;
  float *beams = f_H + Dpad;                                                  // [2][B][Dpad]
  float *G_s = beams + (size_t)2 * B * Dpad;                                  // [B][Dpad] G of the live beams, rebuilt every step
  int32_t *bp = reinterpret_cast<int32_t *>(G_s + (size_t)B * Dpad);          // [max_K][B]
  uint32_t *key_glb = reinterpret_cast<uint32_t *>(bp + (size_t)A.max_K * B); // [S*B]

  for (;;) {
    __syncthreads();
    if (tid == 0) misc[0] = (int32_t)atomicAdd(A.counter, 1u);
    __syncthreads();
    const int64_t blk = misc[0];
    if (blk >= A.n_blocks) break; // every wave of every workgroup reaches this
    if (A.deferred_pass) { // second pass of a windowed-table call of the team encoder (B > 32): only the blocks it left
      if (*A.defer_count == 0u) break;   // (uniform over the grid: nothing was deferred)
      const int32_t k1 = A.out_K[blk];
      if (k1 <= A.K_tab || k1 > A.max_K || k1 > A.K_limit) continue;
    }
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const int NG = (D + 255) >> 8;
    if (D < 1 || D > Dpad) { // not codable with this scratch slab: report and skip
      if (tid == 0) A.out_K[blk] = -1;
      continue;
    }

    // ---- load the block (split == gather through perm), KL in the canonical tree ----
    for (int g0 = 0; g0 < NG; g0 += 4) {
      const int g = g0 + wave;
      double acc = 0.0;
      if (g < NG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = g * 256 + lane * 4 + i;
          float dmu = 0.f, vq = 1.f, vp = 1.f, mp = 0.f;
          if (d < D) {
            const int64_t ix = src_index(A, base, pos, d);
            const float mq = A.q_loc[ix], sq = A.q_scale[ix], sp = A.p_scale[ix];
            mp = A.p_loc[ix];
            acc = acc + kl_dim(mq, sq, mp, sp);
            dmu = mq - mp; vq = sq * sq; vp = sp * sp;
          }
          f_dmu[d] = dmu; f_vq[d] = vq; f_vp[d] = vp; f_mp[d] = mp; f_c[d] = 0.f;
          f_sa[d] = 0.f; f_m[d] = 0.f; f_A[d] = 0.f; f_Bv[d] = 0.f; f_H[d] = 0.f;
        }
      }
      const double gs = wave_tree_sum(acc);
      if (lane == 0) gpart[wave] = gs;
      __syncthreads();
      if (tid == 0) {
        double t = g0 == 0 ? 0.0 : *total_s;
        for (int w = 0; w < 4 && g0 + w < NG; ++w) t = (g0 + w == 0) ? gpart[w] : t + gpart[w];
        *total_s = t;
      }
      __syncthreads();
    }
    if (tid == 0) {
      const int32_t K = num_aux(NG ? (float)*total_s : 0.0f, A.omega);
      misc[1] = K;
      A.out_K[blk] = K;
      hsum[0] = 0;
    }
    __syncthreads();
    const int K = misc[1];
    const bool margins = A.out_margin != nullptr;
    MarginAcc macc;
    if (margins && tid == 0) margin_write(A.out_margin, blk, macc);   // (a block that is not coded keeps "no comparison")
    if (K > A.max_K || K > A.K_limit) continue;

    int cur = 0, Bcur = 1;
    for (int t = 0; t < K; ++t) {
      const StepSeed ss = make_step_seed(A.seed + t);
      const float rho = A.rho[K - 1 - t];
      // phase 1: per-dim constants; c <- c + a  (beam_search_coder.py:67-77,109)
      for (int d = tid; d < D; d += GEN_NT) {
        const StepConst sc = step_constants(rho, f_dmu[d], f_vq[d], f_vp[d], f_c[d]);
        f_sa[d] = sc.sa; f_m[d] = sc.m; f_A[d] = sc.A; f_Bv[d] = sc.Bv; f_H[d] = sc.H;
        f_c[d] = f_c[d] + sc.a;
      }
      __syncthreads();
      const float *bcur = beams + (size_t)cur * B * Dpad;
      // phase 1b: C_b = sum_d (A p + Bv) p of every live beam, canonical tree
      for (int b = wave; b < Bcur; b += GEN_NT / 64) {
        float cb = 0.f;
        for (int g = 0; g < NG; ++g) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int d = g * 256 + lane * 4 + i;
            if (d < D) acc = beam_C_term(acc, t ? bcur[(size_t)b * Dpad + d] : 0.f, f_m[d], f_A[d], f_Bv[d]);
          }
          const float gs = wave_tree_sum(acc);
          cb = g == 0 ? gs : cb + gs;
        }
        if (lane == 0) Cb_s[b] = cb;
      }
      __syncthreads();
      // phase 1c: G of every live beam (it used to be rebuilt for every candidate, four scratch loads and three operations per
      // proposal; r03i: once per step)
      for (int b = wave; b < Bcur; b += GEN_NT / 64)
        for (int d = lane; d < ((D + 3) & ~3); d += 64)   // (the tail of the last quad is read with it: keep it a number)
          G_s[(size_t)b * Dpad + d] = d < D ? beam_G(t ? bcur[(size_t)b * Dpad + d] : 0.f, f_m[d], f_A[d], f_Bv[d], f_sa[d]) : 0.f;
      __syncthreads();
      // phase 2: scores, canonical tree over dims.  A wave takes SAMPLES (s = wave, wave + 4, ...) and scores up to GEN_CH beams at a
      // time against the sample's draw: the Philox block of a dim quad is evaluated once per GEN_CH candidates (r03i; once per
      // candidate until then: one wave per candidate), the group partials leave one reduce-scatter instead of a wave sum each.
      const int N = S * Bcur;
      uint32_t *key = (N <= GEN_NSC) ? key_lds : key_glb;
      const int ownc = rsn_owner<GEN_CH>(lane);
      for (int s = wave; s < S; s += GEN_NT / 64) {
        for (int b0 = 0; b0 < Bcur; b0 += GEN_CH) {
          uint32_t hb[GEN_CH];
#pragma unroll
          for (int j = 0; j < GEN_CH; ++j) hb[j] = hash_from_sum(hsum[cur * GEN_MB + (b0 + j < Bcur ? b0 + j : 0)]);
          float sc = 0.f;
          for (int g = 0; g < NG; ++g) {
            const int d0 = g * 256 + lane * 4;
            float acc[GEN_CH];
#pragma unroll
            for (int j = 0; j < GEN_CH; ++j) acc[j] = 0.f;
            if (d0 < D) {
              uint32_t rm1[4];
              draw_rm1_x4(ss, (uint64_t)s * (uint64_t)D + (uint64_t)d0, rm1);
              float Hd[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) Hd[i] = d0 + i < D ? f_H[d0 + i] : 0.f;
              // the beams' G quads in one batch (one scratch round trip per chunk, not one per beam); Dpad is a
              // multiple of 256 and d0 of 4: the quads are aligned and inside the slab row also where d0 + 3 >= D
              float4 Gq[GEN_CH];
#pragma unroll
              for (int j = 0; j < GEN_CH; ++j)
                Gq[j] = *reinterpret_cast<const float4 *>(G_s + (size_t)(b0 + j < Bcur ? b0 + j : b0) * Dpad + d0);
#pragma unroll
              for (int j = 0; j < GEN_CH; ++j) {
                if (b0 + j < Bcur) { // wave-uniform
                  const float Gj[4] = {Gq[j].x, Gq[j].y, Gq[j].z, Gq[j].w};
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    if (d0 + i < D) {
                      const uint32_t k = ((rm1[i] + 1u) * hb[j]) % IREC_P;   // floormod(r * hash, 10007), :45-47
                      const float z = lut_s[k];                              // ndtri(k / 10007): dist.quantile / scale, :48-49
                      acc[j] = proposal_term(acc[j], z, Hd[i], Gj[i]);
                    }
                  }
                }
              }
            }
            const float gs = reduce_scatter<GEN_CH>(acc, lane);   // lane: total of beam b0 + ownc over this dim group's lanes
            sc = g == 0 ? gs : sc + gs;
          }
          if ((lane & (64 / GEN_CH - 1)) == 0 && b0 + ownc < Bcur) key[s * Bcur + b0 + ownc] = __float_as_uint(sc + Cb_s[b0 + ownc]);
        }
      }
      __syncthreads();
      // phase 3: top-B (beam_search_coder.py:85-89 / :104)
      const int Bnew = B < N ? B : N;
      for (int f = tid; f < N; f += GEN_NT) key[f] = score_key(__uint_as_float(key[f]));
      select_topB_sync<GEN_NT>(key, N, Bnew, Bcur, sm, tid, WorkgroupSync(), nullptr,
                               [&](int j, int32_t, int32_t, uint32_t key_) { if (margins) margin_stash(sm, j, key_); });
      if (margins) {   // (uniform) how close was it?  irec_fast_common.h, "top-B margins"
        if (tid < 64) margin_step(key, N, Bnew, Bcur, sm, lane, t == K - 1, macc);
        __syncthreads();
      }
      // phase 4: gather the surviving beams, extend their index paths (:92-95 / :105-106)
      if (tid < Bnew) {
        const int32_t sp_ = sel_s[tid], bp_ = sel_b[tid];
        hsum[(cur ^ 1) * GEN_MB + tid] = (int32_t)((uint32_t)hsum[cur * GEN_MB + bp_] + (uint32_t)sp_ * (uint32_t)(69 + t));
        bp[(size_t)t * B + tid] = (int32_t)(((uint32_t)sp_ << 8) | (uint32_t)bp_);   // (S <= 2^24, irec_params)
      }
      float *bnext = beams + (size_t)(cur ^ 1) * B * Dpad;
      for (int d0 = tid * 4; d0 < D; d0 += GEN_NT * 4) {   // a thread forms FOUR consecutive dims of every new beam: one Philox block
        float sa4[4];                                       // yields their four draws (r03i: one block per dim until then)
#pragma unroll
        for (int i = 0; i < 4; ++i) sa4[i] = d0 + i < D ? f_sa[d0 + i] : 0.f;
        for (int j = 0; j < Bnew; ++j) {
          const int32_t sp_ = sel_s[j], bp_ = sel_b[j];
          const uint32_t h = hash_from_sum(hsum[cur * GEN_MB + bp_]);
          uint32_t rm1[4];
          draw_rm1_x4(ss, (uint64_t)sp_ * (uint64_t)D + (uint64_t)d0, rm1);   // ((sp_ * D + d0) & 3 is uniform: d0 % 4 == 0)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int d = d0 + i;
            if (d < D) {
              const uint32_t k = ((rm1[i] + 1u) * h) % IREC_P;
              const float y = sa4[i] * lut_s[k];
              const float ob = t ? bcur[(size_t)bp_ * Dpad + d] : 0.f;
              bnext[(size_t)j * Dpad + d] = ob + y; // combined_samples = beams + samples, :81
            }
          }
        }
      }
      __syncthreads();
      cur ^= 1;
      Bcur = Bnew;
    }
    // ---- outputs: beams[0] + p.loc scattered back (merge), index path of beam 0 (:118-122) ----
    const float *bfin = beams + (size_t)cur * B * Dpad;
    for (int d = tid; d < D; d += GEN_NT) {
      const float b0 = K ? bfin[d] : 0.f;
      A.out_sample[src_index(A, base, pos, d)] = b0 + f_mp[d];
    }
    if (tid == 0) {
      int j = 0;
      for (int t = K - 1; t >= 0; --t) {
        const uint32_t v = (uint32_t)bp[(size_t)t * B + j];
        A.out_indices[blk * (int64_t)A.max_K + t] = (int32_t)(v >> 8);
        j = (int)(v & 255u);
      }
      if (margins && K > 0) margin_write(A.out_margin, blk, macc);
    }
  }
}


// LDS carve (bytes): lut2 40032 | [dlog 20016 unless TABLE] | part [4][S][NB] f32 (sort keys overwrite group 0) | SmallLds
// Sample passes.  The per-group partial scores of S_pass samples sit in LDS at a time ([4][S_pass][NB] f32); a step
// scores S in ceil(S / S_pass) passes, each followed by the group combine into the sort keys.  With one pass and at
// most 1024 candidates the keys are written over group 0 of the partials; otherwise they get their own array.
struct FastPlan { int s_pass; bool alias; int nw; size_t off_part, off_key, off_small, bytes; };
// two_tables (the beam-split build): a second copy of the quantile table right behind the first, so that entry offset +
// beam rotation (both < 40 024 bytes) addresses the pair without the wrap -- ONE v_mad_u32_u16 per look-up instead of an
// extract, an add and a conditional subtract; its workgroups have a CU each, the 40 KB are there.
__host__ __device__ inline FastPlan fast_plan(int NB, int S, bool table, bool two_tables = false) {
  FastPlan p;
  const size_t head = (table ? 40032 - 8 : 40032 + 20016) + (two_tables ? (size_t)IREC_PM1 * 4 : 0);
  const size_t row = (size_t)4 * NB * 4;                       // partial-score bytes per sample
  const size_t keys = (((size_t)S * NB * 4) + 15) & ~(size_t)15;
  auto total = [&](int s_pass, bool alias) { return head + (((size_t)s_pass * row + 15) & ~(size_t)15) + (alias ? 0 : keys) + SMALL_LDS_BYTES; };
  const size_t two_per_cu = 80 * 1024 - 256, one_per_cu = 160 * 1024 - 1024;
  p.alias = S * NB <= 1024 && total(S, true) <= (two_tables ? one_per_cu : two_per_cu);   // (two_tables: one workgroup per CU anyway)
  if (p.alias) { p.s_pass = S; p.nw = 4; }
  else {
    // largest pass that still lets two workgroups share a CU, if that leaves passes of >= 16 samples
    long long fit2 = ((long long)two_per_cu - (long long)(head + keys + SMALL_LDS_BYTES + 16)) / (long long)row;
    long long fit1 = ((long long)one_per_cu - (long long)(head + keys + SMALL_LDS_BYTES + 16)) / (long long)row;
    if (fit2 >= 16 || fit2 >= S) { p.s_pass = (int)(fit2 < S ? fit2 : S); p.nw = 4; }
    else { p.s_pass = (int)(fit1 < S ? fit1 : S); p.nw = 8; }   // one 8-wave workgroup per CU
    if (p.s_pass < 1) p.s_pass = 0;                              // does not fit at all -> caller falls back
  }
  p.off_part = head;
  p.off_key = head + (p.alias ? 0 : (((size_t)p.s_pass * row + 15) & ~(size_t)15));
  p.off_small = p.off_key + (p.alias ? (((size_t)p.s_pass * row + 15) & ~(size_t)15) : keys);
  p.bytes = p.off_small + SMALL_LDS_BYTES;
  return p;
}


// SPLIT: 0 = one workgroup per block; 1 = split encoder sharing a block's samples; 2 = split encoder sharing its beams.  Builds
// of their own since r03n: the beam-sharing form holds G of its two beam slots only (201 VGPRs and no scratch at NB = 20,
// against 256 + 512 B when all three forms were one kernel), and the plain form carries no exchange code.
template <int NB, int NW, bool TABLE, int SPLIT = 0>
__global__ __launch_bounds__(NW * 64, 2) void encode_fast_kernel(EncArgs A) {
  using Cfg = FastCfg<NB, TABLE>;
  constexpr int NT = NW * 64;
  constexpr int RW = Cfg::RW, SPC = Cfg::SPC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = A.S, B = A.B;
  const FastPlan plan = fast_plan(NB, S, TABLE, SPLIT == 2);
  const int SP = plan.s_pass;                                                 // samples scored per pass
  const bool keys_alias = plan.alias;
  char *lut2_b = smem;                                                        // float [10006], dlog order
  const uint16_t *dlog_s = TABLE ? A.dlog4r : reinterpret_cast<const uint16_t *>(smem + 40032); // [10006] 4*dlog(j+1)
  float *part_s = reinterpret_cast<float *>(smem + plan.off_part);            // [4][SP][NB] per-group partial scores
  uint32_t *key_s = reinterpret_cast<uint32_t *>(smem + plan.off_key);        // [S*NB] sort keys (over group 0 if aliased)
  SmallLds *sm = reinterpret_cast<SmallLds *>(smem + plan.off_small);
  double *gpart = sm->gpart;
  int32_t *sel_s = sm->sel_s, *sel_b = sm->sel_b;
  int32_t *hsum = &sm->hsum[0][0];                                            // [2][64]
  uint32_t *beta4 = &sm->beta4[0][0];                                         // [2][64] 4*dlog(hash(beam))
  int32_t *misc = sm->misc;
  float *cpart_s = &sm->cpart[0][0];                                          // [4][32] per-group partial C_b
  float *Cb_s = sm->Cb;                                                       // [32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform by construction: keep it in an SGPR
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  commit_table_stamps(A);
  // second pass of a windowed-table call that deferred nothing (the usual case): leave before the tables are staged -- the launch
  // used to cost 11 us of LDS fill per call (r04w kernel trace), now the 3 us of an empty kernel
  if (!TABLE && A.deferred_pass && *A.defer_count == 0u) return;   // (uniform over the grid)
  {
    float *l2 = reinterpret_cast<float *>(lut2_b);
    for (int k = tid; k < (int)IREC_PM1; k += NT) {
      const float v = A.lut2[k];
      l2[k] = v;
      if (SPLIT == 2) l2[k + (int)IREC_PM1] = v;   // (fast_plan: two_tables)
    }
    if (!TABLE) {
      uint16_t *dl = reinterpret_cast<uint16_t *>(smem + 40032);
      for (int k = tid; k < (int)IREC_PM1; k += NT) dl[k] = A.dlog4r[k];
    }
  }
  char *slab = A.ws + (size_t)blockIdx.x * A.ws_per_wg;
  int32_t *bp = reinterpret_cast<int32_t *>(slab);                                            // [max_K][NB]
  float *beams_g = reinterpret_cast<float *>(slab + A.ws_per_wg - (size_t)2 * NB * FAST_MAX_DIM * 4); // [2][NB][1024]
  float *stats_g = beams_g - 3 * FAST_MAX_DIM;  // [3][1024]: mq - mp, sq^2, sp^2 of the block, coalesced

  unsigned long long stamp_prev = A.dbg ? stamp_now() : 0ull;
  // Split mode (A.coop_W > 1, TABLE kernels of 4 waves with aliased keys only): this workgroup serves ONE block together
  // with coop_W - 1 partners and scores sample stripe coop_w of it; no block counter, one trip through the loop.
  const int coop_W = SPLIT != 0 ? (A.coop_W > 1 ? A.coop_W : 2) : 1;   // (the split builds are launched with coop_W >= 2: launch_fast_nw)
  const int coop_w = coop_W > 1 ? (int)(blockIdx.x % (unsigned)coop_W) : 0;
  bool coop_done = false;
  // Beam mode of the split encoder (r02i): the workgroups of a block share its BEAMS instead of its samples -- workgroup w
  // owns the beam slots w, w + W (at most NOWN = 2), scores every sample for the beams it owns, and after the common
  // selection gathers / forms only the NEW beams that land in its slots.  The update -- the longest piece of a split
  // step, and until now repeated in full by every workgroup -- shrinks to two beams; the beams themselves live in ONE
  // slab per block (that of the block's workgroup 0), stored and loaded `sc1` like the exchanged keys and ordered by the
  // same per-step arrival counter (a workgroup reads step t - 1's beams behind step t's counter wait, and nobody can be
  // more than one step ahead, so the double buffer suffices).
  constexpr int NOWN = 2;
  constexpr bool beam_mode = SPLIT == 2;
  // the block's beams: the slab of its first workgroup (beam mode), or my own
  float *beams_blk = reinterpret_cast<float *>(A.ws + (size_t)(blockIdx.x - (unsigned)(beam_mode ? coop_w : 0)) * A.ws_per_wg +
                                               A.ws_per_wg - (size_t)2 * NB * FAST_MAX_DIM * 4);
  if (coop_W > 1 && A.coop_test_orphan && coop_w != 0) return;   // test hook: workgroup 0 of each block is left waiting
  for (;;) {
    __syncthreads();
    if (coop_W > 1) { if (tid == 0) { misc[0] = coop_done ? 0x7FFFFFFF : (int32_t)(blockIdx.x / (unsigned)coop_W); misc[6] = 0; } coop_done = true; }
    else if (tid == 0) misc[0] = (int32_t)atomicAdd(A.counter, 1u);
    __syncthreads();
    const int64_t blk = misc[0];
    if (blk >= A.n_blocks) break; // every wave of every workgroup reaches this
    if (!TABLE && A.deferred_pass) { // second pass of a windowed-table call: only the blocks the table kernels left
      if (*A.defer_count == 0u) break;   // (uniform over the grid: nothing was deferred)
      const int32_t k1 = A.out_K[blk];
      if (k1 <= A.K_tab || k1 > A.max_K || k1 > A.K_limit) continue;
    }
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    // proposal table of this block's dim count (the host listed the distinct dims of the call)
    const uint16_t *tab = nullptr;
    if (TABLE) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (A.tab_dim[q] == D) tab = A.tab[q];
    }
    if (D < 1 || D > FAST_MAX_DIM || (TABLE && tab == nullptr)) { // host promised D <= 1024 and listed dims
      if (tid == 0) A.out_K[blk] = -1;
      continue;
    }
    const int Dp = (D + 3) & ~3;            // row stride of the proposal table
    const int NG = (D + 255) >> 8;          // 1..4 dim groups
    const int NSW = NW / NG;                // sample stripes
    const bool active = wave < NG * NSW;
    const int g = wave % NG, sw = wave / NG;
    const int d0 = g * 256 + lane * 4;

    // ---- my 4 dims (split == gather through perm) and the block's KL ----
    float c[4];
    bool valid[4];
    int64_t ix[4];
    double klacc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = d0 + i;
      valid[i] = d < D;
      c[i] = 0.f;
      ix[i] = valid[i] ? src_index(A, base, pos, d) : src_index(A, base, pos, 0);
      float st3[3] = {0.f, 1.f, 1.f};
      if (valid[i] && active && sw == 0) { // one wave per dim group does the float64 KL and publishes the statistics
        const float mq_ = A.q_loc[ix[i]], sq_ = A.q_scale[ix[i]], mp_ = A.p_loc[ix[i]], sp_ = A.p_scale[ix[i]];
        klacc = klacc + kl_dim(mq_, sq_, mp_, sp_);
        st3[0] = mq_ - mp_; st3[1] = sq_ * sq_; st3[2] = sp_ * sp_;
      }
      if (active && sw == 0) {
        stats_g[d0 + i] = st3[0]; stats_g[FAST_MAX_DIM + d0 + i] = st3[1]; stats_g[2 * FAST_MAX_DIM + d0 + i] = st3[2];
      }
    }
    {
      const double gs = wave_tree_sum(klacc);
      if (sw == 0 && active && lane == 0) gpart[g] = gs;
      __syncthreads();
      if (tid == 0) {
        double tot = gpart[0];
        for (int gg = 1; gg < NG; ++gg) tot = tot + gpart[gg];
        const int32_t K = num_aux((float)tot, A.omega);
        misc[1] = K;
        if (coop_w == 0) A.out_K[blk] = K;
        hsum[0] = 0;
        beta4[0] = 0u; // hash of the empty path is 1 = g^0
      }
      __syncthreads();
    }
    const int K = misc[1];
    if (K > A.max_K || K > A.K_limit) continue;
    if (TABLE && K > A.K_tab) { // beyond the table window: the fused-Philox pass codes it
      if (tid == 0 && coop_w == 0) atomicAdd(A.defer_count, 1u);
      continue;
    }
    if (K == 0) { // nothing to code: sample = p.loc
      if (active && sw == 0 && coop_w == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid[i]) A.out_sample[ix[i]] = 0.f + A.p_loc[ix[i]];
      }
      continue;
    }

    // Per step the lane keeps, for its 4 dims: sa (sample scale), H (z^2 coefficient) and, per live beam, G.
    float sa[4], cH[4];
    float G[NB][4];
    // constants of step `t_next` from the running cumulative variance; returns m, A, Bv for the G / C_b update
    auto step_consts = [&](int t_next, float (&m)[4], float (&cA)[4], float (&cBv)[4]) {
      const float rho = A.rho[K - 1 - t_next];
      // three coalesced 16-byte reads of the slab per step instead of 12 VGPRs held for the whole block
      const float4 q0 = *reinterpret_cast<const float4 *>(stats_g + d0);
      const float4 q1 = *reinterpret_cast<const float4 *>(stats_g + FAST_MAX_DIM + d0);
      const float4 q2 = *reinterpret_cast<const float4 *>(stats_g + 2 * FAST_MAX_DIM + d0);
      const float dmu_[4] = {q0.x, q0.y, q0.z, q0.w}, vq_[4] = {q1.x, q1.y, q1.z, q1.w}, vp_[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const StepConst sc = step_constants(rho, dmu_[i], vq_[i], vp_[i], c[i]);
        sa[i] = valid[i] ? sc.sa : 0.f; cH[i] = valid[i] ? sc.H : 0.f;
        m[i] = valid[i] ? sc.m : 0.f; cA[i] = valid[i] ? sc.A : 0.f; cBv[i] = valid[i] ? sc.Bv : 0.f;
        c[i] = c[i] + sc.a; // cumulative_auxiliary_variance += auxiliary_var (:109)
        __builtin_amdgcn_sched_barrier(0); // one dim at a time: the division sequences are register hungry
      }
    };
    // ---- prologue: step 0 has one (all-zero) beam ----
    {
      float m[4], cA[4], cBv[4];
      step_consts(0, m, cA, cBv);
      float cacc = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) G[b][i] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        G[0][i] = beam_G(0.f, m[i], cA[i], cBv[i], sa[i]);
        cacc = beam_C_term(cacc, 0.f, m[i], cA[i], cBv[i]);
      }
      const float cg = wave_tree_sum(cacc);
      if (active && sw == 0 && lane == 0) cpart_s[g * 32 + 0] = cg;
      __syncthreads();
      if (tid == 0) {
        float cb = cpart_s[0];
        for (int gg = 1; gg < NG; ++gg) cb = cb + cpart_s[gg * 32];
        Cb_s[0] = cb;
      }
      // (visibility of Cb_s: the barrier after scoring)
    }

    IREC_STAMP(0);
    if (A.dbg && tid == 0) {
      A.dbg[(size_t)blockIdx.x * 16 + 4] += 1ull;                                       // blocks coded by this workgroup
      if (A.dbg[(size_t)blockIdx.x * 16 + 5] == 0ull) A.dbg[(size_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memrealtime(); // first block
      A.dbg[(size_t)blockIdx.x * 16 + 6] = __builtin_amdgcn_s_memrealtime();
    }
    int cur = 0, Bcur = 1;
    for (int t = 0; t < K; ++t) {
      const StepSeed ss = make_step_seed(A.seed + t);
      // row s at + s * Dp.  Lanes past the padded row end (d0 >= Dp, all their dims invalid) read the row START instead:
      // their coefficients are zero, but the offsets they gather with must still be table entries (finite z) --
      // reading past the last row would hand them arbitrary bits, and 0 * NaN is not 0.
      const uint16_t *tab_t = TABLE ? tab + (size_t)t * S * Dp + (d0 < Dp ? d0 : 0) : nullptr;
      uint32_t bet[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) bet[b] = __builtin_amdgcn_readfirstlane(beta4[cur * 64 + (b < Bcur ? b : 0)]);

      // my samples: all of them, or stripe coop_w of the split encoder (contiguous: [s_lo, s_hi))
      const int S_w = coop_W > 1 ? (S + coop_W - 1) / coop_W : S;
      const int s_lo = coop_W > 1 ? (coop_w * S_w < S ? coop_w * S_w : S) : 0;
      const int s_hi = coop_W > 1 ? (s_lo + S_w < S ? s_lo + S_w : S) : S;
      const int N = (s_hi - s_lo) * Bcur;                       // candidates scored HERE (all of the step's unless split)
      // beam mode: the slots I own and which of them hold a beam in this step
      int own_b[NOWN];
      uint32_t own_bet[NOWN];
#pragma unroll
      for (int o = 0; o < NOWN; ++o) {
        own_b[o] = coop_w + o * coop_W;
        own_bet[o] = __builtin_amdgcn_readfirstlane(beta4[cur * 64 + (own_b[o] < Bcur ? own_b[o] : 0)]);
      }
      if (beam_mode) {
        // ---------------- scoring: every sample x the beams I own (beam_search_coder.py:80-84) ----------------
        constexpr int SPB = RW / NOWN;                            // samples per reduce-scatter: accumulator sl * NOWN + o
        constexpr int HB = SPB < 8 ? SPB : 8;                     // rows fetched together
        if (active && own_b[0] < Bcur) {
          const int n_mine = S > sw ? (S - sw + NSW - 1) / NSW : 0;   // my samples: sw, sw + NSW, ...
          // Rows are fetched ONE SUB-BATCH AHEAD (round 4): with two beams per workgroup a sub-batch of eight rows is 64 look-ups per
          // lane, far less than the L2 latency of its row loads, which every sub-batch used to wait out (five per step at S = 36).
          uint2 ap_nxt[HB];
          auto load_rows = [&](const int first) {
#pragma unroll
            for (int k = 0; k < HB; ++k) {
              const int m = first + k;
              ap_nxt[k] = make_uint2(0u, 0u);
              if (m < n_mine) ap_nxt[k] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)(m * NSW + sw) * Dp);
            }
          };
          load_rows(0);
          for (int m0 = 0; m0 < n_mine; m0 += SPB) {
            float acc[RW];
#pragma unroll
            for (int p = 0; p < RW; ++p) acc[p] = 0.f;
            // one sub-batch of NH rows: rows m0 + h .. + NH - 1 (past my last sample: entry 0, the total is dropped)
            auto sub_batch = [&](auto nh_tag, const int h, const uint2 (&apb)[HB]) {
              constexpr int NH = decltype(nh_tag)::value;
              uint2 ap[NH];
#pragma unroll
              for (int k = 0; k < NH; ++k) ap[k] = apb[k];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float z[NH][NOWN];
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                  const uint32_t w = (i & 2) ? ap[k].y : ap[k].x;
#pragma unroll
                  for (int o = 0; o < NOWN; ++o) {
                    uint32_t ad;                                   // 16-bit half x 1 + rotation: into the table pair, no wrap
                    if (i & 1) asm("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[1,0,0,0]" : "=v"(ad) : "v"(w), "s"(own_bet[o]));
                    else asm("v_mad_u32_u16 %0, %1, 1, %2" : "=v"(ad) : "v"(w), "s"(own_bet[o]));
                    z[k][o] = lds_abs_f32(ad);
                  }
                }
#pragma unroll
                for (int k = 0; k < NH; ++k)
#pragma unroll
                  for (int o = 0; o < NOWN; ++o)
                    acc[(h + k) * NOWN + o] = proposal_term(acc[(h + k) * NOWN + o], z[k][o], cH[i], G[o][i]);
                __builtin_amdgcn_sched_barrier(0);
              }
            };
            // the last reduce-scatter of a step is rarely full (S = 36: 16 + 16 + 4): sub-batches past my last sample are
            // skipped, one that holds at most half its rows runs at half width -- 36 sample slots instead of 48
#pragma unroll
            for (int h = 0; h < SPB; h += HB) {
              const int left = n_mine - (m0 + h);                  // wave-uniform
              if (left <= 0) continue;
              uint2 ap_cur[HB];
#pragma unroll
              for (int k = 0; k < HB; ++k) ap_cur[k] = ap_nxt[k];
              if (left > HB) load_rows(m0 + h + HB);               // the next sub-batch's rows, under this one's look-ups
              if constexpr (HB >= 8) {
                if (left <= HB / 2) { sub_batch(std::integral_constant<int, HB / 2>{}, h, ap_cur); continue; }
              }
              sub_batch(std::integral_constant<int, HB>{}, h, ap_cur);
            }
            const float tot = reduce_scatter<RW>(acc, lane);
            const int p = RW == 64 ? lane : (lane >> 1);
            const int sl = p / NOWN, o = p - sl * NOWN;
            const int m = m0 + sl, b = coop_w + o * coop_W;
            if (m < n_mine && b < Bcur && (RW == 64 || (lane & 1) == 0)) part_s[((size_t)g * SP + (m * NSW + sw)) * NB + b] = tot;
          }
        }
#if IREC_COOP_GRANULES
        // (granules: a partner that has seen ALL my keys of this step goes on to load the beams I stored in the last update --
        //  every wave's beam stores are drained before the barrier that precedes the first key store)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        IREC_STAMP(1);
        // ---------------- my candidates' keys go straight into the exchange (flat f = s * Bcur + b) ----------------
        {
#if IREC_COOP_GRANULES
          unsigned long long *xg = reinterpret_cast<unsigned long long *>(A.coop_xch) + ((size_t)(t & 1) * COOP_MAX_BLOCKS + (size_t)blk) * COOP_KEYS;
          const unsigned long long tag = (unsigned long long)(uint32_t)(t + 1) << 32;
#else
          uint32_t *xk = A.coop_xch + ((size_t)(t & 1) * COOP_MAX_BLOCKS + (size_t)blk) * COOP_KEYS;
#endif
          for (int idx = tid; idx < S * NOWN; idx += NT) {
            const int s_ = idx / NOWN, o = idx - s_ * NOWN, b = coop_w + o * coop_W;
            if (b < Bcur) {
              float sc = part_s[((size_t)0 * SP + s_) * NB + b];
              for (int gg = 1; gg < NG; ++gg) sc = sc + part_s[((size_t)gg * SP + s_) * NB + b];
#if IREC_COOP_GRANULES
              __hip_atomic_store(&xg[s_ * Bcur + b], tag | (unsigned long long)score_key(sc + Cb_s[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
              __hip_atomic_store(&xk[s_ * Bcur + b], score_key(sc + Cb_s[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            }
          }
        }
      } else
      for (int s_base = s_lo; s_base < s_hi || s_base == s_lo; s_base += SP) {
      const int s_end = s_base + SP < s_hi ? s_base + SP : s_hi;
      // ---------------- scoring: samples [s_base, s_end) x Bcur candidates (beam_search_coder.py:80-84) ----------------
      if (active) {
        const int s_per_stripe = (s_end - s_base + NSW - 1) / NSW;
        const int nchunks = (s_per_stripe + SPC - 1) / SPC;
        // table variant: proposal rows (4 x uint16 byte offsets 4*dlog(r) of my dims) are fetched one chunk ahead
        uint2 alp_next[SPC];
        if (TABLE) {
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc) {
            const int s0 = s_base + cc * NSW + sw;
            alp_next[cc] = make_uint2(0u, 0u);
            if (s0 < s_end) alp_next[cc] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)s0 * Dp);
          }
        }
        for (int ch = 0; ch < nchunks; ++ch) {
          float acc[RW];
#pragma unroll
          for (int p = 0; p < RW; ++p) acc[p] = 0.f;
          uint2 alp[SPC];
          if (TABLE) {
#pragma unroll
            for (int cc = 0; cc < SPC; ++cc) {
              alp[cc] = alp_next[cc];
              const int sn = s_base + ((ch + 1) * SPC + cc) * NSW + sw;
              if (sn < s_end) alp_next[cc] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)sn * Dp);
            }
          }
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc) {
            const int s = s_base + (ch * SPC + cc) * NSW + sw;
            if (s < s_end) { // wave-uniform
              uint32_t al[4];
              if (TABLE) {
                const uint2 ap = alp[cc];
                al[0] = ap.x & 0xFFFFu; al[1] = ap.x >> 16; al[2] = ap.y & 0xFFFFu; al[3] = ap.y >> 16;
              } else {
                uint32_t rm1[4];
                draw_rm1_x4(ss, (uint64_t)s * (uint64_t)D + (uint64_t)d0, rm1);
#pragma unroll
                for (int i = 0; i < 4; ++i) al[i] = dlog_s[rm1[i]];
              }
              if (Bcur == NB) {
                // steady state: all NB beams alive -> branch-free; the NB gathers of one dim are issued back to back
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  float z[NB];
#pragma unroll
                  for (int b = 0; b < NB; ++b) {
                    uint32_t ad = al[i] + bet[b];                       // 4*(dlog r + dlog h)
                    const uint32_t ad2 = ad - IREC_LUT2_BYTES;
                    ad = ad2 < ad ? ad2 : ad;                           // mod 10006 (one conditional subtract)
                    z[b] = lds_abs_f32(ad);
                  }
#pragma unroll
                  for (int b = 0; b < NB; ++b) acc[cc * NB + b] = proposal_term(acc[cc * NB + b], z[b], cH[i], G[b][i]);
                  __builtin_amdgcn_sched_barrier(0); // one dim's NB gathers in flight at a time (VGPR budget)
                }
              } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                  if (b < Bcur) { // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                      uint32_t ad = al[i] + bet[b];
                      const uint32_t ad2 = ad - IREC_LUT2_BYTES;
                      ad = ad2 < ad ? ad2 : ad;
                      const float z = lds_abs_f32(ad);
                      acc[cc * NB + b] = proposal_term(acc[cc * NB + b], z, cH[i], G[b][i]);
                    }
                  }
                }
              }
            }
          }
          const float tot = reduce_scatter<RW>(acc, lane);
          const int p = RW == 64 ? lane : (lane >> 1);
          const int cc = p / NB, b = p - cc * NB;
          const int s = s_base + (ch * SPC + cc) * NSW + sw;
          if (cc < SPC && s < s_end && b < Bcur && (RW == 64 || (lane & 1) == 0))
            part_s[((size_t)g * SP + (s - s_base)) * NB + b] = tot;
        }
      }
      __syncthreads();
      if (s_base == s_lo) IREC_STAMP(1);
      // ---------------- combine dim groups in order, add C_b, build sort keys ----------------
      if (keys_alias) {
        // single pass, keys written over group 0 of the partials: two phases with a barrier in between because
        // key f = s * Bcur + b and partial (s, b) = s * NB + b only coincide when Bcur == NB
        constexpr int MK = (1024 + NT - 1) / NT;
        uint32_t mykey[MK];
#pragma unroll
        for (int q = 0; q < MK; ++q) {
          const int f = q * NT + tid;
          mykey[q] = 0u;
          if (f < N) {
            const int s = f / Bcur, b = f - s * Bcur;
            float sc = part_s[((size_t)0 * SP + s) * NB + b];
            for (int gg = 1; gg < NG; ++gg) sc = sc + part_s[((size_t)gg * SP + s) * NB + b];
            mykey[q] = score_key(sc + Cb_s[b]);
          }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MK; ++q) {
          const int f = q * NT + tid;
          if (f < N) key_s[f] = mykey[q];
        }
      } else {
        for (int f = s_base * Bcur + tid; f < s_end * Bcur; f += NT) {
          const int s = f / Bcur, b = f - s * Bcur;
          float sc = part_s[((size_t)0 * SP + (s - s_base)) * NB + b];
          for (int gg = 1; gg < NG; ++gg) sc = sc + part_s[((size_t)gg * SP + (s - s_base)) * NB + b];
          key_s[f] = score_key(sc + Cb_s[b]);
        }
        if (s_end < S) __syncthreads(); // the next pass overwrites the partials
      }
      if (s_end >= s_hi) break;
      } // sample passes
      const int Ng = S * Bcur;                                  // candidates of the step over all workgroups
      const int Bnew = B < Ng ? B : Ng;
      const bool last = (t == K - 1);
      // Beam-split build (r04): this step's sample scale is copied and the NEXT step's IEEE constants (six divisions and a
      // square root per dim: ~3 k cycles of one wave per SIMD) are formed HERE, between publishing my keys and sweeping the
      // partners' -- they depend on t only, not on the selection -- instead of in the update behind it: the wait for the
      // partners (4.3 k cycles per step, profiles/archive/r04a/stamps.log) absorbs them.
      float sa_now[4] = {sa[0], sa[1], sa[2], sa[3]};
      float m_nx[4] = {0.f, 0.f, 0.f, 0.f}, cA_nx[4] = {0.f, 0.f, 0.f, 0.f}, cBv_nx[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (beam_mode && IREC_SPLIT_CONSTS_EARLY != 0) {
        if (active && !last) step_consts(t + 1, m_nx, cA_nx, cBv_nx);
      }
      if (A.dbg && tid == 0) { const unsigned long long now_ = stamp_now(); A.dbg[(size_t)blockIdx.x * 16 + 11] += now_ - stamp_prev; } // combine
      if (coop_W > 1) {
        // ---- split encoder: every workgroup of the block publishes the sort keys of ITS candidates at their global flat
        // positions, waits for its partners, reads the whole step's keys back and runs the same selection as everybody else.
        // Hand-off without fences (MI355X_MICROARCH.md, inter-workgroup visibility): every handed-off byte is stored and
        // loaded `sc1` (relaxed agent-scope atomics: L2-bypassing, never a stale line), the storing waves drain their stores
        // (s_waitcnt vmcnt(0)) before ONE lane behind the workgroup barrier adds to the block's arrival counter, the polling
        // lane reads that counter `sc1`, and the payload loads sit behind the next workgroup barrier.
#if IREC_COOP_GRANULES
        // Granules (MI355X_MICROARCH.md, handoff-1to1): a key travels as ONE naturally aligned 8-byte {key, step tag} `sc1` store
        // and every partner sweeps the step's granules with `sc1` loads until each carries this step's tag -- no drain, no
        // arrival counter, no second round trip for the payload (r03l: ≈ 3 µs of a 12 µs step were the three serial round trips
        // of the counter form).  Tags: t + 1 >= 1; the preparation kernel zeroed this block's granules, buffer t & 1 last held tag
        // t - 1.  A thread gives up like the counter form did: sticky error flag, COOP_GIVE_UP_TICKS (100 ms).
        unsigned long long *xg = reinterpret_cast<unsigned long long *>(A.coop_xch) + ((size_t)(t & 1) * COOP_MAX_BLOCKS + (size_t)blk) * COOP_KEYS;
        const uint32_t tag = (uint32_t)(t + 1);
        unsigned long long sub_prev = A.dbg ? stamp_now() : 0ull;   // diagnostics: [12] publish, [13] sweep (wait + read back)
        auto sub_stamp = [&](int slot) {
          if (A.dbg && tid == 0) { const unsigned long long now_ = stamp_now(); A.dbg[(size_t)blockIdx.x * 16 + slot] += now_ - sub_prev; sub_prev = now_; }
        };
        __syncthreads();   // (aliased keys: all of them written; beam mode: every partial read before key_s is refilled)
        if (!beam_mode)
          for (int f = tid; f < N; f += NT)
            __hip_atomic_store(&xg[s_lo * Bcur + f], ((unsigned long long)tag << 32) | (unsigned long long)key_s[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sub_stamp(12);
        int32_t bad = 0;
        {
          constexpr int MKX = (1024 + NT - 1) / NT;
          uint32_t kk[MKX];
          uint32_t pending = 0u;
#pragma unroll
          for (int q = 0; q < MKX; ++q) { kk[q] = 0u; if (q * NT + tid < Ng) pending |= 1u << q; }
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
          for (uint32_t turn = 1; pending != 0u; ++turn) {
            unsigned long long gq[MKX];
#pragma unroll
            for (int q = 0; q < MKX; ++q)                                           // all of a thread's loads in flight together
              gq[q] = (pending >> q) & 1u ? __hip_atomic_load(&xg[q * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (int q = 0; q < MKX; ++q)
              if (((pending >> q) & 1u) && (uint32_t)(gq[q] >> 32) == tag) { kk[q] = (uint32_t)gq[q]; pending &= ~(1u << q); }
            if (pending == 0u || (turn & 63u) != 0u) continue;
            if (__hip_atomic_load(A.coop_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > COOP_GIVE_UP_TICKS) { // the partners are not resident -- give up, loudly
              __hip_atomic_store(A.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bad = 1; break;
            }
          }
          // (key_s of the sample mode holds my own keys at [0, N): every thread has read its own before the barrier above and
          //  nobody reads key_s again before the barrier below)
          // (no __syncthreads_or: it brings a static LDS word with it, and the table must stay at LDS address 0 -- lds_abs_f32)
          if (bad) misc[6] = 1;
          __syncthreads();
          if (misc[6]) { // every workgroup of the block sees the flag (it is sticky): nobody waits for anybody any more
            if (tid == 0 && coop_w == 0) A.out_K[blk] = -2;
            break;
          }
#pragma unroll
          for (int q = 0; q < MKX; ++q) {
            const int f = q * NT + tid;
            if (f < Ng) key_s[f] = kk[q];
          }
        }
        sub_stamp(13);
#else
        uint32_t *xk = A.coop_xch + ((size_t)(t & 1) * COOP_MAX_BLOCKS + (size_t)blk) * COOP_KEYS;
        unsigned long long sub_prev = A.dbg ? stamp_now() : 0ull;   // diagnostics: [12] publish, [13] wait for partners, [14] read back
        auto sub_stamp = [&](int slot) {
          if (A.dbg && tid == 0) { const unsigned long long now_ = stamp_now(); A.dbg[(size_t)blockIdx.x * 16 + slot] += now_ - sub_prev; sub_prev = now_; }
        };
        __syncthreads();   // (aliased keys: all of them written; beam mode: every partial read before key_s is refilled)
        if (!beam_mode)
          for (int f = tid; f < N; f += NT) __hip_atomic_store(&xk[s_lo * Bcur + f], key_s[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __hip_atomic_fetch_add(&A.coop_arrive[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sub_stamp(12);
          const uint32_t want = (uint32_t)coop_W * (uint32_t)(t + 1);
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
          int32_t bad = 0;
          // (the counter load's own round trip paces the loop; the error flag -- a second far load -- and the clock are
          //  looked at every 32nd turn only: r02i, the wait is the longest piece of a split step)
          for (uint32_t turn = 1; __hip_atomic_load(&A.coop_arrive[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++turn) {
            if ((turn & 31u) != 0u) continue;
            if (__hip_atomic_load(A.coop_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > COOP_GIVE_UP_TICKS) { // the partners are not resident -- give up, loudly
              __hip_atomic_store(A.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bad = 1; break;
            }
          }
          misc[6] = bad;
          sub_stamp(13);
        }
        __syncthreads();
        if (misc[6]) { // every workgroup of the block sees the flag (it is sticky): nobody waits for anybody any more
          if (tid == 0 && coop_w == 0) A.out_K[blk] = -2;
          break;
        }
        {   // all of a thread's loads in flight before the first is stored (Ng <= 1024 in split mode)
          constexpr int MKX = (1024 + NT - 1) / NT;
          uint32_t kk[MKX];
#pragma unroll
          for (int q = 0; q < MKX; ++q) {
            const int f = q * NT + tid;
            kk[q] = f < Ng ? __hip_atomic_load(&xk[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          }
#pragma unroll
          for (int q = 0; q < MKX; ++q) {
            const int f = q * NT + tid;
            if (f < Ng) key_s[f] = kk[q];
          }
        }
        sub_stamp(14);
#endif
      }
      if constexpr (SPLIT != 0) __builtin_assume(Ng <= 1024);   // (host, split_width(): S * NB <= 1024 -- the selection's other paths fold away)
      select_topB<NT, true>(key_s, Ng, Bnew, Bcur, sm, A.dbg ? A.dbg + (size_t)blockIdx.x * 16 : nullptr); // first barrier inside orders key_s writes
      IREC_STAMP(2);
      // ---------------- new hashes / back-pointers (beam_search_coder.py:94-95) ----------------
      // (the discrete log of the new hash is a load from the global table when the proposals come from tables: it is issued
      //  here and stored into the LDS behind the update, whose own loads it then waits under -- nobody reads the next step's
      //  offsets before the barrier that closes the update)
      uint32_t beta_new = 0u;
      if (tid < Bnew) {
        const int32_t sp_ = sel_s[tid], bp_ = sel_b[tid];
        const int32_t nh = (int32_t)((uint32_t)hsum[cur * 64 + bp_] + (uint32_t)sp_ * (uint32_t)(69 + t));
        hsum[(cur ^ 1) * 64 + tid] = nh;
        beta_new = dlog_s[hash_from_sum(nh) - 1u];
        bp[(size_t)t * NB + tid] = (sp_ << 6) | bp_;
      }
      // ---------------- gather the surviving beams (beam_search_coder.py:92-93), prepare the next step ----------------
      __builtin_amdgcn_s_setprio(2); // serial phase: ahead of the co-resident workgroup's scoring waves
      if (active && beam_mode) {
        // beam mode: only the new beams that land in my slots; parents come from (and new beams go to) the block's shared slab
        const float sa_t[4] = {sa_now[0], sa_now[1], sa_now[2], sa_now[3]};
        const float *bold = beams_blk + ((size_t)cur * NB) * FAST_MAX_DIM + d0;
        float *bnew = beams_blk + ((size_t)(cur ^ 1) * NB) * FAST_MAX_DIM + d0;
        uint2 apv[NOWN];
        float obv[NOWN][4];
        uint32_t bet_old[NOWN];
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
          const int j = coop_w + o * coop_W;
          apv[o] = make_uint2(0u, 0u); bet_old[o] = 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) obv[o][i] = 0.f;
          if (j < Bnew) { // wave-uniform
            const int32_t sp_ = __builtin_amdgcn_readfirstlane(sel_s[j]);
            const int32_t bp_ = __builtin_amdgcn_readfirstlane(sel_b[j]);
            bet_old[o] = __builtin_amdgcn_readfirstlane(beta4[cur * 64 + bp_]);
            apv[o] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)sp_ * Dp);
            if (t) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                obv[o][i] = __hip_atomic_load(bold + (size_t)bp_ * FAST_MAX_DIM + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if constexpr (IREC_SPLIT_CONSTS_EARLY == 0) {
          if (!last) step_consts(t + 1, m_nx, cA_nx, cBv_nx);               // (A/B build: in the update, under its loads' latency)
        }
        const float (&m)[4] = m_nx, (&cA)[4] = cA_nx, (&cBv)[4] = cBv_nx;   // next step's constants: formed under the partners' wait
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
          const int j = coop_w + o * coop_W;
          float cacc = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) G[o][i] = 0.f;
          if (j < Bnew) { // wave-uniform
            const uint32_t al[4] = {apv[o].x & 0xFFFFu, apv[o].x >> 16, apv[o].y & 0xFFFFu, apv[o].y >> 16};
            float nb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t ad = al[i] + bet_old[o];      // (beam-split build: the table pair, no wrap)
              const float y = sa_t[i] * lds_abs_f32(ad); // dist.quantile(.), :48-49
              nb[i] = obv[o][i] + y;                       // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
            }
            if (last) {
              if (j == 0 && sw == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (valid[i]) A.out_sample[ix[i]] = nb[i] + A.p_loc[ix[i]]; // beams[0] + coding_dist.loc, :122
              }
            } else {
              if (sw == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  __hip_atomic_store(bnew + (size_t)j * FAST_MAX_DIM + i, nb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                G[o][i] = beam_G(nb[i], m[i], cA[i], cBv[i], sa[i]);
                cacc = beam_C_term(cacc, nb[i], m[i], cA[i], cBv[i]);
              }
            }
          }
          if (!last) {   // (every wave of the group computes the same bits; one writes)
            const float ctot = wave_tree_sum(cacc);
            if (sw == 0 && lane == 0 && j < Bnew) cpart_s[g * 32 + j] = ctot;
          }
        }
      } else if (active) {
        const float sa_t[4] = {sa[0], sa[1], sa[2], sa[3]};   // this step's sample scale
        const float *bold = beams_g + ((size_t)cur * NB) * FAST_MAX_DIM + d0;
        float *bnew = beams_g + ((size_t)(cur ^ 1) * NB) * FAST_MAX_DIM + d0;
        // G is dead from the end of scoring until it is rebuilt below: every entry is redefined here, so nothing of it
        // has to survive the selection (no spills), and its registers take the in-flight loads of the update.
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) G[j][i] = 0.f;
        float m[4], cA[4], cBv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = 0.f; cA[i] = 0.f; cBv[i] = 0.f; }
        float cacc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) cacc[j] = 0.f;
        constexpr int UB = NB <= 10 ? NB : (NB + 1) / 2;      // beams per load batch
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += UB) {
          // ---- issue the batch's global reads (proposal rows, old beams) back to back ----
          uint2 apv[UB];
          float4 obv4[UB];
          uint32_t bet_old[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int j = j0 + u;
            apv[u] = make_uint2(0u, 0u);
            obv4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bet_old[u] = 0u;
            if (j < NB && j < Bnew) { // wave-uniform
              const int32_t sp_ = __builtin_amdgcn_readfirstlane(sel_s[j]);
              const int32_t bp_ = __builtin_amdgcn_readfirstlane(sel_b[j]);
              bet_old[u] = __builtin_amdgcn_readfirstlane(beta4[cur * 64 + bp_]);
              if (TABLE) {
                apv[u] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)sp_ * Dp);
              } else {
                uint32_t rm1[4];
                draw_rm1_x4(ss, (uint64_t)sp_ * (uint64_t)D + (uint64_t)d0, rm1);
                apv[u] = make_uint2((uint32_t)dlog_s[rm1[0]] | ((uint32_t)dlog_s[rm1[1]] << 16),
                                    (uint32_t)dlog_s[rm1[2]] | ((uint32_t)dlog_s[rm1[3]] << 16));
              }
              if (t) obv4[u] = *reinterpret_cast<const float4 *>(bold + (size_t)bp_ * FAST_MAX_DIM);
            }
          }
          if (j0 == 0 && !last) step_consts(t + 1, m, cA, cBv); // next step's constants, under the loads' latency
          // ---- new beams, their G and C terms ----
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int j = j0 + u;
            if (j < NB && j < Bnew) { // wave-uniform
              const uint32_t al[4] = {apv[u].x & 0xFFFFu, apv[u].x >> 16, apv[u].y & 0xFFFFu, apv[u].y >> 16};
              const float obv[4] = {obv4[u].x, obv4[u].y, obv4[u].z, obv4[u].w};
              float nb[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                uint32_t ad = al[i] + bet_old[u];
                const uint32_t ad2 = ad - IREC_LUT2_BYTES;
                ad = ad2 < ad ? ad2 : ad;
                const float y = sa_t[i] * lds_abs_f32(ad); // dist.quantile(.), :48-49
                nb[i] = obv[i] + y;                          // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
              }
              if (last) {
                if (j == 0 && sw == 0 && coop_w == 0) {
#pragma unroll
                  for (int i = 0; i < 4; ++i)
                    if (valid[i]) A.out_sample[ix[i]] = nb[i] + A.p_loc[ix[i]]; // beams[0] + coding_dist.loc, :122
                }
              } else {
                if (sw == 0) *reinterpret_cast<float4 *>(bnew + (size_t)j * FAST_MAX_DIM) = make_float4(nb[0], nb[1], nb[2], nb[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  G[j][i] = beam_G(nb[i], m[i], cA[i], cBv[i], sa[i]);
                  cacc[j] = beam_C_term(cacc[j], nb[i], m[i], cA[i], cBv[i]);
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) {
          const float ctot = reduce_scatter<32>(cacc, lane);  // lane l holds beam (l >> 1)
          const int j = lane >> 1;
          if (sw == 0 && (lane & 1) == 0 && j < Bnew) cpart_s[g * 32 + j] = ctot;
        }
      }
      if (tid < Bnew) beta4[(cur ^ 1) * 64 + tid] = beta_new;
      __syncthreads();
      __builtin_amdgcn_s_setprio(0);
      IREC_STAMP(3);
      if (!last && tid < Bnew) {
        float cb = cpart_s[tid];
        for (int gg = 1; gg < NG; ++gg) cb = cb + cpart_s[gg * 32 + tid];
        Cb_s[tid] = cb; // read after the next scoring barrier
      }
      cur ^= 1;
      Bcur = Bnew;
    }
    // ---- index path of beam 0 (beam_search_coder.py:118-121) ----
    __syncthreads();
    if (tid == 0 && coop_w == 0 && !(coop_W > 1 && misc[6])) {
      int j = 0;
      for (int t = K - 1; t >= 0; --t) {
        const int32_t v = __builtin_nontemporal_load(&bp[(size_t)t * NB + j]);
        A.out_indices[blk * (int64_t)A.max_K + t] = v >> 6;
        j = v & 63;
      }
    }
  }
}

// ======================================================================================================
//  shared proposal table: the int32 draw of get_pseudo_random_sample depends only on (seed + t, S, D) -- it is the
//  same for every block of D dims in the call (every caller passes one seed: coder.py:444-449) -- so the Philox
//  stream is evaluated ONCE per call, fused with "% 10006" and the discrete-log map, into
//      tab[t][s][d] = 4 * dlog_g(r[s, d])   (uint16, row stride = D rounded up to 4)
//  which the beam-striped encoder streams from L2 (8 bytes per lane per sample).
// ======================================================================================================
__global__ __launch_bounds__(256) void alpha_table_kernel(int64_t seed, int32_t S, int32_t D, int32_t K_tab,
                                                          const uint16_t *__restrict__ dlog4r, uint16_t *__restrict__ tab,
                                                          const uint32_t *__restrict__ keep) {
  if (keep && *keep) return;   // the table in place was built for exactly this key
  plain_table_rows(seed, S, D, K_tab, dlog4r, tab, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// ======================================================================================================
//  test hooks
// ======================================================================================================
__global__ void uniform_int_kernel(int64_t seed, int64_t n, int32_t *out) {
  const StepSeed ss = make_step_seed(seed);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = 1 + (int32_t)draw_rm1(ss, (uint64_t)e);
}

// scores[N] float -> sel[2 * Bnew] = (sample, beam) of the Bnew best candidates in order (tie -> lower flat index)
template <bool QUICK>
__global__ __launch_bounds__(256) void select_test_kernel(const float *scores, int N, int Bnew, int Bcur, uint32_t *keys,
                                                          int32_t *sel) {
  __shared__ SmallLdsT<64, 64, 512> sm;   // (the 60-beam team build's: room to refine up to 512 survivors)
  for (int f = threadIdx.x; f < N; f += 256) keys[f] = score_key(scores[f]);
  select_topB_sync<256, QUICK>(keys, N, Bnew, Bcur, &sm, (int)threadIdx.x, WorkgroupSync());
  if (threadIdx.x < Bnew) { sel[2 * threadIdx.x] = sm.sel_s[threadIdx.x]; sel[2 * threadIdx.x + 1] = sm.sel_b[threadIdx.x]; }
}

// in: [64 lanes][width] floats; out[lane] = canonical-tree total of column (lane >> shift); out has 128 floats
__global__ void reduce_scatter_test_kernel(const float *in, float *out, int width) {
  const int lane = threadIdx.x & 63;
  if (width == 64) {
    float v[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = in[lane * 64 + j];
    out[lane] = reduce_scatter<64>(v, lane);
  } else if (width == 32) {
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = in[lane * 32 + j];
    out[lane] = reduce_scatter<32>(v, lane);
  } else if (width == 20) { // arbitrary-width form: out[lane] = total, out[64 + lane] = column this lane owns (or -1)
    float v[rsn_room(20)];
#pragma unroll
    for (int j = 0; j < 20; ++j) v[j] = in[lane * 20 + j];
    out[lane] = reduce_scatter_n<20>(v, lane);
    out[64 + lane] = (float)rsn_owner<20>(lane);
  } else if (width == 21) { // the scoring loop's form of the 20-value reduce-scatter (register pairs, bank-masked DPP adds)
    rs_f2 a[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) a[j] = (rs_f2){in[lane * 20 + 2 * j], in[lane * 20 + 2 * j + 1]};
    out[lane] = reduce_scatter_20(a, lane);
    out[64 + lane] = (float)rs20_owner(lane);
  } else {
    float v[rsn_room(10)];
#pragma unroll
    for (int j = 0; j < 10; ++j) v[j] = in[lane * 10 + j];
    out[lane] = reduce_scatter_n<10>(v, lane);
    out[64 + lane] = (float)rsn_owner<10>(lane);
  }
}

// ======================================================================================================
//  launchers (called from irec_host.cpp)
// ======================================================================================================
hipError_t launch_block_kl(const EncArgs &A, float *out_kl, int grid, hipStream_t st) {
  hipLaunchKernelGGL(block_kl_kernel, dim3(grid), dim3(256), 0, st, A, out_kl);
  return hipGetLastError();
}

size_t generic_lds_bytes() { return 40032 + (size_t)GEN_NSC * 4 + GEN_SMALL_BYTES + 8 + (size_t)GEN_MB * 4 + 64; }

hipError_t launch_encode_generic(const EncArgs &A, int grid, hipStream_t st) {
  const size_t lds = generic_lds_bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_generic_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encode_generic_kernel, dim3(grid), dim3(GEN_NT), lds, st, A);
  return hipGetLastError();
}

template <int NB, int NW, bool TABLE, int SPLIT = 0>
static hipError_t launch_fast_t(const EncArgs &A, int grid, hipStream_t st) {
  const size_t lds = fast_plan(NB, A.S, TABLE, SPLIT == 2).bytes;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_fast_kernel<NB, NW, TABLE, SPLIT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((encode_fast_kernel<NB, NW, TABLE, SPLIT>), dim3(grid), dim3(NW * 64), lds, st, A);
  return hipGetLastError();
}

int fast_nb_for(int B) { return B <= 10 ? 10 : B <= 20 ? 20 : B <= 32 ? 32 : 0; }

// LDS bytes of the fast encoder for (B, S), or (size_t)-1 when it cannot run (B too large / nothing fits)
size_t fast_lds_for(int B, int S, bool table) {
  const int nb = fast_nb_for(B);
  if (!nb) return (size_t)-1;
  const FastPlan p = fast_plan(nb, S, table);
  return p.s_pass >= 1 ? p.bytes : (size_t)-1;
}
int fast_split_beam_waves(int B) { const int nb = fast_nb_for(B); return nb ? split_beam_nw(nb) : 0; }   // waves per workgroup of the beam-split build
int fast_waves_for(int B, int S, bool table) { const int nb = fast_nb_for(B); return nb ? fast_plan(nb, S, table).nw : 0; }
const char *fast_kernel_name(int B, int S, bool table) {
  static thread_local char buf[64];
  const int nb = fast_nb_for(B);
  snprintf(buf, sizeof buf, "encode_fast_kernel<%d,%d,%s>", nb, nb ? fast_plan(nb, S, table).nw : 0, table ? "true" : "false");
  return buf;
}

size_t fast_ws_for(int B, int max_K) { return fast_ws_bytes(fast_nb_for(B), max_K); }
size_t fast_ws_bytes_nb(int NB, int max_K) { return fast_ws_bytes(NB, max_K); }

template <int NB, bool TABLE>
static hipError_t launch_fast_nw(const EncArgs &A, int grid, hipStream_t st) {
  if constexpr (TABLE)
    if (A.coop_W > 1) {   // split call (host: split_width / split_beam_width; aliased-key plans only, four waves)
      if (fast_plan(NB, A.S, TABLE).nw != 4) return hipErrorInvalidValue;
      return A.coop_beams != 0 ? launch_fast_t<NB, split_beam_nw(NB), true, 2>(A, grid, st) : launch_fast_t<NB, 4, true, 1>(A, grid, st);
    }
  return fast_plan(NB, A.S, TABLE).nw == 8 ? launch_fast_t<NB, 8, TABLE>(A, grid, st) : launch_fast_t<NB, 4, TABLE>(A, grid, st);
}

hipError_t launch_encode_fast(const EncArgs &A, bool table, int grid, hipStream_t st) {
  switch (fast_nb_for(A.B)) {
    case 10: return table ? launch_fast_nw<10, true>(A, grid, st) : launch_fast_nw<10, false>(A, grid, st);
    case 20: return table ? launch_fast_nw<20, true>(A, grid, st) : launch_fast_nw<20, false>(A, grid, st);
    case 32: return table ? launch_fast_nw<32, true>(A, grid, st) : launch_fast_nw<32, false>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_alpha_table(int64_t seed, int32_t S, int32_t D, int32_t K_tab, const uint16_t *dlog4r, uint16_t *tab,
                              const uint32_t *keep, hipStream_t st) {
  const int64_t quads = ((int64_t)S * ((D + 3) & ~3) * K_tab + 3) / 4;
  const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
  hipLaunchKernelGGL(alpha_table_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, seed, S, D, K_tab, dlog4r, tab, keep);
  return hipGetLastError();
}

hipError_t launch_uniform_int(int64_t seed, int64_t n, int32_t *out, hipStream_t st) {
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(uniform_int_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, seed, n, out);
  return hipGetLastError();
}

hipError_t launch_select_test(const float *scores, int N, int Bnew, int Bcur, uint32_t *keys, int32_t *sel, bool quick, hipStream_t st) {
  if (quick) hipLaunchKernelGGL(select_test_kernel<true>, dim3(1), dim3(256), 0, st, scores, N, Bnew, Bcur, keys, sel);
  else hipLaunchKernelGGL(select_test_kernel<false>, dim3(1), dim3(256), 0, st, scores, N, Bnew, Bcur, keys, sel);
  return hipGetLastError();
}

hipError_t launch_reduce_scatter_test(const float *in, float *out, int width, hipStream_t st) {
  hipLaunchKernelGGL(reduce_scatter_test_kernel, dim3(1), dim3(64), 0, st, in, out, width);
  return hipGetLastError();
}

} // namespace irec
