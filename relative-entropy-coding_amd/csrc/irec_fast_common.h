// irec_fast_common.h -- device pieces shared by the register-resident encoders (irec_kernels.hip: one 4/8-wave
// workgroup per block; irec_team.hip: two independent 4-wave teams per workgroup over one shared look-up table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "irec_device.h"
#include "irec_kernels.h"

namespace irec {

// ======================================================================================================
//  small shared pieces
// ======================================================================================================
__device__ __forceinline__ int64_t src_index(const EncArgs &A, int64_t base, int32_t pos, int d) {
  return base + (A.perm ? (int64_t)A.perm[pos + d] : (int64_t)(pos + d));
}

// First thing in every encode kernel: the table keys the call's preparation kernel left pending become the stamps (irec_kernels.h,
// "The call's preparation kernel") -- by now the tables they describe have been built.  Idempotent (the deferred pass repeats it).
__device__ __forceinline__ void commit_table_stamps(const EncArgs &A) {
  if (A.ws_head != nullptr && blockIdx.x == 0 && threadIdx.x < 4 * WS_STAMP_WORDS)
    A.ws_head[WS_STAMP_WORD + threadIdx.x] = A.ws_head[WS_PENDING_WORD + threadIdx.x];
}

// Rows of a plain proposal table tab[t][s][d] = 4 * dlog_g(r[s, d]) (uint16, row stride = D rounded up to 4), four entries per
// thread; workgroup `wg` of `n_wg` (256 threads each).  The int32 draw of get_pseudo_random_sample (beam_search_coder.py:38-43)
// depends only on (seed + t, S, D): it is evaluated once per call.
__device__ __forceinline__ void plain_table_rows(int64_t seed, int32_t S, int32_t D, int32_t K_tab, const uint16_t *__restrict__ dlog4r,
                                                 uint16_t *__restrict__ tab, int64_t wg, int64_t n_wg) {
  const int Dp = (D + 3) & ~3;
  const int64_t per_step = (int64_t)S * Dp;
  const int64_t total = per_step * K_tab;
  for (int64_t q = (wg * 256 + threadIdx.x) * 4; q < total; q += n_wg * 256 * 4) {
    const int t = (int)(q / per_step);
    const int64_t rem = q - (int64_t)t * per_step;
    const int s = (int)(rem / Dp), d0 = (int)(rem - (int64_t)s * Dp);
    const StepSeed ss = make_step_seed(seed + t);
    uint16_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (d0 + i < D) ? dlog4r[draw_rm1(ss, (uint64_t)s * (uint64_t)D + (uint64_t)(d0 + i))] : (uint16_t)0;
    *reinterpret_cast<uint2 *>(tab + q) = make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16));
  }
}

// Diagnostic phase stamps (only when EncArgs.dbg != nullptr; the values never reach an output of the coder).
__device__ __forceinline__ unsigned long long stamp_now() { return __builtin_amdgcn_s_memtime(); }

// Small per-workgroup (or per-team) LDS state shared by the encoders; MB = most beams the kernel serves, CPW = beams the
// C_b partial rows hold, NC = survivors of the threshold selection that can be refined (64: ranked directly; more beams than
// ~32 leave more than 64 candidates above the B-th largest lane maximum, see select_topB_sync).
template <int MB, int CPW = 32, int NC = 64>
struct SmallLdsT {
  static constexpr int CANDS = NC;
  static constexpr int CP = CPW;
  union {                           // the two selection paths never run at the same time
    unsigned long long wb[32];      // per-wave maxima of the scan-based selection, double buffered (<= 16 waves)
    unsigned long long cand[NC];    // compacted (key, flat) survivors of the threshold selection
  };
  int32_t sel_s[MB], sel_b[MB];     // selected (sample, beam) per new beam
  uint32_t sel_bo[MB];              // 4 * dlog(hash) of the selected parent beam (team encoder)
  int32_t hsum[2][MB];              // running int32 sum of simple_hash per beam, double buffered
  uint32_t beta4[2][MB];            // 4 * dlog(hash) per beam, double buffered
  int32_t misc[8];                  // [0] block id, [1] K, [7] selection path flag
  union {                           // KL partials are consumed before the first C_b partial is written
    double gpart[4];                // per dim-group KL partial sums
    float cpart[4][CPW];            // per dim-group partial C_b
  };
  float Cb[CPW];                    // C_b of the live beams
};
using SmallLds = SmallLdsT<64>;
constexpr size_t SMALL_LDS_BYTES = (sizeof(SmallLds) + 15) & ~(size_t)15;

// Block-wide top-Bnew selection over key[0..N) (uint32 sort keys, 0 = taken / empty), NT threads.
// tf.argsort(DESCENDING)[:B] semantics (beam_search_coder.py:85-89): value descending, ties to the lower flat index.
//  - N <= 1024 (every BASELINE config but the S=403 stress case): wave 0 alone, candidates in registers:
//      1. lane maxima; T = Bnew-th largest lane maximum  => at least Bnew candidates are >= T
//      2. compact the candidates >= T (a few dozen) to one per lane through LDS
//      3. rank them by (key desc, flat asc); rank r < Bnew IS new beam r
//    ~600 wave instructions and ONE barrier instead of Bnew barrier rounds.
//  - N > 1024 (B = 30 / S = 148, the S = 403 stress case): the same three stages, wave 0 streaming the keys twice
//    (16-byte reads; LDS, or the L2-resident slab) instead of holding them: lane maxima, then the compaction.  Two barriers
//    instead of Bnew scan rounds with one each.
//  - more than 64 candidates at the threshold (a tie storm): all waves scan the keys, one barrier per selected beam
//    (element f owned by thread f % NT).
// `sync` is the barrier of the NT threads that run the selection together (the whole workgroup, or one team of it) and
// `tid` the thread's index among them.
// `post(j, s, b, key)` is called by the thread that has just recorded new beam j = candidate (sample s, parent beam b)
// with sort key `key`, before the barrier that publishes the selection.
struct NoPost { __device__ __forceinline__ void operator()(int, int32_t, int32_t, uint32_t) const {} };
// The Bnew-th largest of the 64 lane maxima M (every lane calls; wave-uniform result): at least Bnew keys are >= it.  Lane l counts the
// lanes with a strictly larger maximum; the value whose count is the largest one below Bnew is the Bnew-th largest (values below it
// see at least Bnew larger ones) -- found by probing the counts downwards from Bnew - 1: without ties the counts are a permutation of
// 0..63 and the first probe hits.  (Round 4: this replaced a 64-bit wave reduction of six cross-lane steps, r04ah/select_rates.log.)
template <bool QUICK>
__device__ __forceinline__ uint32_t kth_largest_lane_max(uint32_t M, int Bnew) {
  uint32_t cnt_gt = 0u;
#pragma unroll
  for (int l = 0; l < 64; ++l) cnt_gt += (uint32_t)__builtin_amdgcn_readlane((int)M, l) > M ? 1u : 0u;
  if constexpr (!QUICK) {   // the form of rounds 1-3: the minimum over the lanes whose count is below Bnew, by a 64-bit wave reduction
    uint32_t T = cnt_gt < (uint32_t)Bnew ? M : 0xFFFFFFFFu;
    T = 0xFFFFFFFFu - (uint32_t)wave_max_u64((unsigned long long)(0xFFFFFFFFu - T)); // wave min
    return T;
  }
  uint32_t T = 0u;
  for (int c = Bnew <= 64 ? Bnew - 1 : 63; c >= 0; --c) {   // (more than 64 beams, generic kernel: the smallest lane maximum -- at least 64
                                                            //  survivors, not Bnew: rank_survivors sends the shortfall to the scan)
    const unsigned long long hit = __ballot(cnt_gt == (uint32_t)c);
    if (hit) { T = (uint32_t)__builtin_amdgcn_readlane((int)M, (int)__builtin_ctzll(hit)); break; }   // (wave-uniform)
  }
  return T;
}

// Stage 3 of the threshold selection, run by ONE wave (lane = tid < 64): cand[0, C) holds the (key, flat) pairs of every
// candidate >= T, at least Bnew of them.  Ranks them by (key descending, flat ascending) and records rank r < Bnew as new beam
// r (sel_s / sel_b, post()).  More than 64 survivors (many beams: the B-th largest of 64 lane maxima leaves ~2 B candidates
// above it) are first cut down: the exact Bnew-th largest key T' is found bit by bit -- 32 rounds of "how many survivors
// are >= T' | bit" over the <= NC / 64 survivors a lane holds in registers -- and the survivors >= T' are compacted again.
// Returns false when more than 64 candidates remain at the threshold (ties), C exceeded the buffer or FEWER than Bnew survived:
// caller falls back.  (Fewer than Bnew: only with more than 64 beams -- the generic kernel -- where kth_largest_lane_max returns the
// smallest lane maximum and guarantees 64 survivors, not Bnew: S = 65, B >= 65 at step 0 when the 64 largest keys sit in 64
// distinct lanes.  Round 4 recorded ranks 0..63 there and left sel_s / sel_b[64..Bnew) stale.)
template <bool QUICK, class SM, class Post>
__device__ __forceinline__ bool rank_survivors(SM *sm, uint32_t C, int Bnew, int Bcur, int tid, Post &&post) {
  constexpr int NQ = SM::CANDS / 64;
  if (C > (uint32_t)SM::CANDS || C < (uint32_t)Bnew) return false;
  if (C > 64u) {
    if constexpr (NQ > 1) {
      uint32_t mk[NQ], mf[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const uint32_t at = (uint32_t)(q * 64 + tid);
        const unsigned long long v = at < C ? sm->cand[at] : 0ull;
        mk[q] = (uint32_t)(v >> 32); mf[q] = (uint32_t)v;
      }
      uint32_t T2 = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t tryT = T2 | (1u << bit);
        uint32_t cnt = 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) cnt += (uint32_t)__popcll(__ballot(mk[q] >= tryT));
        if (cnt >= (uint32_t)Bnew) T2 = tryT;                // wave-uniform
      }
      uint32_t base = 0u;                                    // compact the survivors >= T2 (all reads above are complete)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool in = mk[q] >= T2 && mk[q] != 0u;
        const unsigned long long mask = __ballot(in);
        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (in && pos < 64u) sm->cand[pos] = ((unsigned long long)mk[q] << 32) | mf[q];
        base += (uint32_t)__popcll(mask);
      }
      C = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (C > 64u) return false;
    } else {
      return false;
    }
  }
  // rank the C <= 64 survivors; lanes >= C hold a null candidate
  const unsigned long long mine = tid < (int)C ? sm->cand[tid] : 0ull;
  const uint32_t mk1 = (uint32_t)(mine >> 32), mf1 = (uint32_t)mine;
  uint32_t rank = 0u;
  if constexpr (!QUICK) {   // the form of rounds 1-3
    for (uint32_t l = 0; l < C; ++l) {
      const uint32_t ok_ = (uint32_t)__builtin_amdgcn_readlane((int)mk1, (int)l);
      const uint32_t of_ = (uint32_t)__builtin_amdgcn_readlane((int)mf1, (int)l);
      rank += (ok_ > mk1 || (ok_ == mk1 && of_ < mf1)) ? 1u : 0u;
    }
  } else {
  // (key, ~flat) as ONE 64-bit number: larger = better (key descending, ties to the lower flat index).  Constant-lane broadcasts in
  // groups of eight, one 64-bit compare per candidate (round 4; a run-time lane index and three compares until then).  The lanes >= C
  // hold key 0, which beats no survivor (T >= 1).
  const uint32_t nf1 = ~mf1;
  const unsigned long long mp = ((unsigned long long)mk1 << 32) | nf1;
  // First by the key alone -- one broadcast and one 32-bit compare per candidate.  Equal keys among the survivors (exactly equal
  // float32 scores: rare) then share a rank: every survivor writes its lane into slot `rank` of cand[] (all lanes have read their
  // candidate: one wave, program order) and reads it back; a lane that finds another's there lost a collision, and the wave ranks
  // again by (key, ~flat).
#pragma unroll
  for (int l0 = 0; l0 < 64; l0 += 8) {
    if ((uint32_t)l0 < C) { // wave-uniform
#pragma unroll
      for (int l = l0; l < l0 + 8; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)mk1, l) > mk1 ? 1u : 0u;
    }
  }
  volatile unsigned long long *slot = sm->cand;   // (another lane may own the slot: no store-to-load forwarding)
  if (tid < (int)C) slot[rank] = (unsigned long long)tid;
  const bool lost = tid < (int)C && slot[rank] != (unsigned long long)tid;
  if (__ballot(lost)) { // wave-uniform
    rank = 0u;
#pragma unroll
    for (int l0 = 0; l0 < 64; l0 += 8) {
      if ((uint32_t)l0 < C) { // wave-uniform
#pragma unroll
        for (int l = l0; l < l0 + 8; ++l) {
          const uint32_t ok_ = (uint32_t)__builtin_amdgcn_readlane((int)mk1, l);
          const uint32_t on_ = (uint32_t)__builtin_amdgcn_readlane((int)nf1, l);
          rank += (((unsigned long long)ok_ << 32) | on_) > mp ? 1u : 0u;
        }
      }
    }
  }
  }
  if (tid < (int)C && rank < (uint32_t)Bnew) {
    const int32_t s_ = (int32_t)(mf1 / (uint32_t)Bcur); // best_ind_aux  (beam_search_coder.py:89)
    const int32_t b_ = (int32_t)(mf1 % (uint32_t)Bcur); // best_ind_beam (beam_search_coder.py:88)
    sm->sel_s[rank] = s_;
    sm->sel_b[rank] = b_;
    post((int)rank, s_, b_, mk1);
  }
  return true;
}

// ---- XCD-aware block hand-out of the persistent batch encoders (encode_team_kernel, encode_lone_kernel) ----
// Workgroups go to the eight XCDs round-robin (workgroup w runs on XCD w mod 8) and each XCD has its own L2.  The rows of one
// latent tensor lie next to each other (eight 1000-dim blocks of an 8192-dim tensor) and gather their statistics out of the
// same 4 x 32 KB through the shuffle, a line of 32 floats feeding all eight blocks.  Handing consecutive rows to consecutive
// workgroups makes every XCD fetch every line of every tensor (a one-beam call of 256 latents: 0.26 ms; XCD-aware: 0.19 ms,
// profiles/archive/r03l/ab_lone_xcd.log).  So rows are dealt in groups of 64 = 8 x 8: the rows 8 x .. 8 x + 7 of a group go to XCD x.
//   * static first round (slot u = k * gridDim.x + w for the k-th block stream of workgroup w, one block per CU before any
//     CU gets a second): each aligned group of 64 slots is transposed as an 8 x 8 when gridDim.x is a multiple of 8 (then
//     u mod 8 == w mod 8); a bijection on every full group below n_static, the tail and other grids keep u -> u;
//   * later blocks: one counter per XCD, value c of XCD x -> row first + 64 (c / 8) + 8 x + c mod 8; an XCD whose share has
//     run out takes from the next one's (rows of a counter only grow, so "past the end" is final; this is also what codes
//     the shares of XCDs that have no workgroup of the launch).  The counters lie 256 bytes apart.
__device__ __forceinline__ int64_t xcd_static_row(int64_t u, int64_t n_static, int grid) {
  const int64_t m = u & ~(int64_t)63;
  if ((grid & 7) != 0 || m + 64 > n_static) return u;
  const int v = (int)(u & 63);
  return m + ((v & 7) << 3) + (v >> 3);
}
// One lane calls this.  `steal` (0 at kernel start, kept by the caller) counts the XCD shares this caller found exhausted.
// Returns n when nothing is left.
__device__ __forceinline__ int64_t xcd_pull_row(const EncArgs &A, int64_t first, int64_t n, int &steal) {
  if (first >= n) return n;                       // the static round dealt every row: no counter is touched
  const uint32_t span = (uint32_t)(n - first);    // (n < 2^31: irec_beam_encode)
  for (; steal < 8; ++steal) {
    const uint32_t x = ((uint32_t)blockIdx.x + (uint32_t)steal) & 7u;
    const uint32_t rem = span & 63u;
    const uint32_t share = (span >> 6) * 8u + (rem > 8u * x ? (rem - 8u * x < 8u ? rem - 8u * x : 8u) : 0u);   // rows of XCD x
    const uint32_t c = atomicAdd(A.xcd_counter + WS_XCD_STRIDE * x, 1u);
    if (c < share) return first + (int64_t)(c >> 3) * 64 + (int64_t)(8u * x + (c & 7u));
  }
  return n;
}

constexpr int SELECT_PAR_MIN = 4096;   // candidates per step from which every wave takes part in the streamed selection
// QUICK (round 4, scripts/microbench/select_rates.hip): the threshold by probing the lane counts and the ranks by constant-lane
// broadcasts -- 6.6 k -> 5.0 k cycles per selection at 720 candidates.  The encoders whose calls are bound by a lone chain's serial
// phases take it (one-table / split encoder, the two-team builds: 9 blocks 0.137 -> 0.131 ms); the three-team 168-VGPR builds keep the
// form they were tuned with -- with QUICK the headline kernel is 1.1 % slower on the same box (r04al/ab.log).
template <int NT, bool QUICK = false, class SM, class Sync, class Post = NoPost>
__device__ __forceinline__ void select_topB_sync(uint32_t *key, int N, int Bnew, int Bcur, SM *sm, const int tid, Sync &&sync,
                                                 unsigned long long *dbg = nullptr, Post &&post = Post()) {
  constexpr int NWV = NT / 64;
  int32_t *sel_s = sm->sel_s, *sel_b = sm->sel_b;
  unsigned long long t0 = dbg ? stamp_now() : 0ull;
  // (streamed selection by every wave, below: its lane-maxima array lies over cand[], which is idle until the compaction)
  const bool par = NWV > 1 && N > SELECT_PAR_MIN;
  uint32_t *lane_max = reinterpret_cast<uint32_t *>(sm->cand);
  if (par) { if (tid < 64) lane_max[tid] = 0u; if (tid == 0) sm->misc[4] = 0; }
  sync(); // keys written by all waves
  if (dbg && tid == 0) { const unsigned long long t1 = stamp_now(); dbg[8] += t1 - t0; t0 = t1; } // wait for keys
  bool done = false;
  if (N <= 1024) {
    if (tid < 64) {
      __builtin_amdgcn_s_setprio(3); // the whole workgroup waits for this wave: win issue arbitration on its SIMD
      const int nslots = (N + 63) >> 6;
      uint32_t k[16];
      uint32_t M = 0u;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int f = q * 64 + tid;
        k[q] = (q < nslots && f < N) ? key[f] : 0u;
        M = k[q] > M ? k[q] : M;
      }
      // 1. threshold: #lanes with a strictly larger maximum
      uint32_t T = kth_largest_lane_max<QUICK>(M, Bnew);
      // 2. compact candidates >= T (T >= 1 because at least Bnew <= N lanes hold a real key)
      uint32_t base = 0u;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q < nslots) { // wave-uniform
          const bool in = k[q] >= T;
          const unsigned long long mask = __ballot(in);
          const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          if (in && pos < (uint32_t)SM::CANDS) sm->cand[pos] = ((unsigned long long)k[q] << 32) | (uint32_t)(q * 64 + tid);
          base += (uint32_t)__popcll(mask);
        }
      }
      const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      sm->misc[7] = rank_survivors<QUICK>(sm, C, Bnew, Bcur, tid, post) ? 1 : 0;   // 0: pathological tie storm -> the scan below
      __builtin_amdgcn_s_setprio(0);
      if (dbg && tid == 0) { const unsigned long long t1 = stamp_now(); dbg[9] += t1 - t0; t0 = t1; } // wave-0 selection
    }
    sync();
    if (dbg && tid == 0) { const unsigned long long t1 = stamp_now(); dbg[10] += t1 - t0; t0 = t1; } // closing barrier
    done = sm->misc[7] != 0;
  }
  else if (par) {
    // N beyond SELECT_PAR_MIN (B = 30 / 50 with hundreds of samples; the keys may lie in the L2-resident slab): EVERY
    // wave streams a share of the keys -- one wave alone is bound by the latency of its ~N / 256 dependent 16-byte reads per
    // pass (r03g stamps: 5-9 % of a step at B = 50).  Same three stages: (1) lane maxima of a share, merged over the waves
    // by LDS max (virtual lane l = the keys the lanes l of all waves read: 64 disjoint sets, so the Bnew-th largest of their
    // maxima still leaves at least Bnew candidates); (2) every wave compacts its survivors, slots from an LDS counter (the
    // ranking does not depend on their order); (3) wave 0 ranks.  Three barriers more, ~NWV times shorter passes.
    const bool al = (reinterpret_cast<uintptr_t>(key) & 15) == 0;
    const int n4 = al ? N >> 2 : 0;
    const uint4 *key4 = reinterpret_cast<const uint4 *>(key);
    const int lane = tid & 63;
    {
      uint32_t M = 0u;
      for (int i = tid; i < n4; i += NT) {
        const uint4 v = key4[i];
        const uint32_t a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
        M = M > a ? M : a;
        M = M > b ? M : b;
      }
      for (int f = (n4 << 2) + tid; f < N; f += NT) M = key[f] > M ? key[f] : M;
      if (M) __hip_atomic_fetch_max(&lane_max[lane], M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    sync();
    uint32_t T;
    {
      const uint32_t M = lane_max[lane];
      T = kth_largest_lane_max<QUICK>(M, Bnew);
      T = T ? T : 1u;                                        // (0 marks a taken / empty key)
    }
    sync(); // every wave has read the maxima: cand[] may be written
    {
      auto put = [&](bool in, uint32_t k, uint32_t flat) {
        const unsigned long long mask = __ballot(in);
        if (mask) {                                          // wave-uniform
          const uint32_t n_in = (uint32_t)__popcll(mask);
          uint32_t base = 0u;
          if (lane == 0) base = (uint32_t)__hip_atomic_fetch_add(&sm->misc[4], (int32_t)n_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          if (in && pos < (uint32_t)SM::CANDS) sm->cand[pos] = ((unsigned long long)k << 32) | flat;
        }
      };
      for (int i0 = tid - lane; i0 < n4; i0 += NT) {         // (wave-uniform trip count)
        const int i = i0 + lane;
        const uint4 v = i < n4 ? key4[i] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
        if (__ballot((a > b ? a : b) >= T)) {                // most 256-key rounds hold no survivor at all
          put(v.x >= T, v.x, (uint32_t)(4 * i));
          put(v.y >= T, v.y, (uint32_t)(4 * i + 1));
          put(v.z >= T, v.z, (uint32_t)(4 * i + 2));
          put(v.w >= T, v.w, (uint32_t)(4 * i + 3));
        }
      }
      for (int f0 = (n4 << 2) + tid - lane; f0 < N; f0 += NT) {
        const int f = f0 + lane;
        const uint32_t k = f < N ? key[f] : 0u;
        put(k >= T, k, (uint32_t)f);
      }
    }
    sync();
    if (tid < 64) {
      __builtin_amdgcn_s_setprio(3);
      const uint32_t C = (uint32_t)sm->misc[4];
      sm->misc[7] = rank_survivors<QUICK>(sm, C, Bnew, Bcur, tid, post) ? 1 : 0;
      __builtin_amdgcn_s_setprio(0);
    }
    sync();
    done = sm->misc[7] != 0;
  }
  else {
    if (tid < 64) {
      __builtin_amdgcn_s_setprio(3);
      const bool al = (reinterpret_cast<uintptr_t>(key) & 15) == 0;
      const int n4 = al ? N >> 2 : 0;                        // quads read as uint4, the rest one by one
      const uint4 *key4 = reinterpret_cast<const uint4 *>(key);
      uint32_t M = 0u;
      for (int i = tid; i < n4; i += 64) {
        const uint4 v = key4[i];
        const uint32_t a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
        M = M > a ? M : a;
        M = M > b ? M : b;
      }
      for (int f = (n4 << 2) + tid; f < N; f += 64) M = key[f] > M ? key[f] : M;
      uint32_t T = kth_largest_lane_max<QUICK>(M, Bnew);
      T = T ? T : 1u;                                        // (0 marks a taken / empty key)
      uint32_t base = 0u;
      auto put = [&](bool in, uint32_t k, uint32_t flat) {
        const unsigned long long mask = __ballot(in);
        if (mask) {                                          // wave-uniform
          const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
          if (in && pos < (uint32_t)SM::CANDS) sm->cand[pos] = ((unsigned long long)k << 32) | flat;
          base += (uint32_t)__popcll(mask);
        }
      };
      for (int i0 = 0; i0 < n4; i0 += 64) {
        const int i = i0 + tid;
        const uint4 v = i < n4 ? key4[i] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
        if (__ballot((a > b ? a : b) >= T)) {                // most 256-key rounds hold no survivor at all
          put(v.x >= T, v.x, (uint32_t)(4 * i));
          put(v.y >= T, v.y, (uint32_t)(4 * i + 1));
          put(v.z >= T, v.z, (uint32_t)(4 * i + 2));
          put(v.w >= T, v.w, (uint32_t)(4 * i + 3));
        }
      }
      for (int f0 = n4 << 2; f0 < N; f0 += 64) {
        const int f = f0 + tid;
        const uint32_t k = f < N ? key[f] : 0u;
        put(k >= T, k, (uint32_t)f);
      }
      const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      sm->misc[7] = rank_survivors<QUICK>(sm, C, Bnew, Bcur, tid, post) ? 1 : 0;
      __builtin_amdgcn_s_setprio(0);
    }
    sync();
    done = sm->misc[7] != 0;
  }
  if (done) return;
  unsigned long long *wb = sm->wb;
  for (int it = 0; it < Bnew; ++it) {
    unsigned long long best = 0ull;
    for (int f = tid; f < N; f += NT) {
      const unsigned long long c = cand_pack(key[f], (uint32_t)f);
      best = c > best ? c : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) wb[(it & 1) * NWV + (tid >> 6)] = best;
    sync();
    unsigned long long g = 0ull;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const unsigned long long o = wb[(it & 1) * NWV + w];
      g = o > g ? o : g;
    }
    const uint32_t fstar = 0xFFFFFFFFu - (uint32_t)g;
    if ((uint32_t)tid == fstar % (uint32_t)NT) key[fstar] = 0u;
    if (tid == 0) {
      const int32_t s_ = (int32_t)(fstar / (uint32_t)Bcur), b_ = (int32_t)(fstar % (uint32_t)Bcur);
      sel_s[it] = s_;
      sel_b[it] = b_;
      post(it, s_, b_, (uint32_t)(g >> 32));
    }
  }
  sync();
}


// ======================================================================================================
//  top-B margins (round 5; irec_beam_encode_ex, IREC_FLAG_MARGINS): how close was the selection?
//
//  The emitted indices depend on the scores only through the top-B SET of every step but the last and through the WINNER of the
//  last step (beams[0], beam_search_coder.py:85-89,118-122).  Another summation order of the same float32 terms (TensorFlow's
//  reduce_sum) can change an index only where those comparisons are closer than the orders disagree.  A margin
//  build reports, per block (EncArgs::out_margin, four floats):
//    [0] min over the steps t < K - 1 that reject a candidate of  score(rank Bnew - 1) - score(rank Bnew)   (+inf: no such step)
//    [1] |score(rank Bnew - 1)| at that step
//    [2] score(rank 0) - score(rank 1) at the last step   (+inf: one candidate)
//    [3] |score(rank 0)| at the last step
//  Scores are the float32 values the selection ranked (the inverse of score_key); the differences are float32 subtractions.
//
//  Mechanics: the selection's post() callback stashes the key of new beam j (margin_stash: the upper half of the first 512 bytes of
//  cand[], idle once the ranks are final and never touched by the scan's wb[]); after the selection's closing barrier ONE wave
//  zeroes the selected keys in key[] (the scan path has done so already), takes the maximum of what is left -- the best rejected
//  candidate -- and the minimum of the stash.  The caller puts a barrier behind it before anything overwrites key[].
// ======================================================================================================
struct MarginAcc { float gap = __builtin_inff(), at = 0.f, top_gap = __builtin_inff(), top_at = 0.f; };
__device__ __forceinline__ float key_score(uint32_t key) {   // inverse of score_key (-0 comes back as +0, NaN as a NaN)
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}
template <class SM>
__device__ __forceinline__ void margin_stash(SM *sm, int j, uint32_t key) { reinterpret_cast<uint32_t *>(sm->cand)[64 + j] = key; }
template <class SM>
__device__ __forceinline__ void margin_step(uint32_t *key, int N, int Bnew, int Bcur, SM *sm, int lane, bool last, MarginAcc &acc) {
  static_assert(sizeof(sm->cand) >= (64 + (sizeof(sm->sel_s) / sizeof(int32_t))) * 4, "the key stash must fit behind the scan's wb[]");
  const uint32_t *stash = reinterpret_cast<const uint32_t *>(sm->cand) + 64;
  uint32_t kmin = 0xFFFFFFFFu;
  for (int j = lane; j < Bnew; j += 64) {
    const uint32_t k = stash[j];
    kmin = k < kmin ? k : kmin;
    key[sm->sel_s[j] * Bcur + sm->sel_b[j]] = 0u;          // taken
  }
  kmin = 0xFFFFFFFFu - (uint32_t)wave_max_u64((unsigned long long)(0xFFFFFFFFu - kmin));   // key of rank Bnew - 1
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (keys in the slab: the zeroes are in place before other lanes read them)
  uint32_t M = 0u;
  for (int f = lane; f < N; f += 64) { const uint32_t k = key[f]; M = k > M ? k : M; }
  M = (uint32_t)wave_max_u64((unsigned long long)M);        // key of rank Bnew: the best rejected candidate (0: none)
  if (!last) {
    if (M != 0u) {
      const float g = key_score(kmin) - key_score(M);
      if (g < acc.gap) { acc.gap = g; acc.at = __builtin_fabsf(key_score(kmin)); }
    }
  } else {
    const uint32_t k0 = stash[0], k1 = Bnew >= 2 ? stash[1] : M;
    if (k1 != 0u) acc.top_gap = key_score(k0) - key_score(k1);
    acc.top_at = __builtin_fabsf(key_score(k0));
  }
}
__device__ __forceinline__ void margin_write(float *out_margin, int64_t blk, const MarginAcc &acc) {
  *reinterpret_cast<float4 *>(out_margin + 4 * blk) = make_float4(acc.gap, acc.at, acc.top_gap, acc.top_at);
}

struct WorkgroupSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
template <int NT, bool QUICK = false>
__device__ __forceinline__ void select_topB(uint32_t *key, int N, int Bnew, int Bcur, SmallLds *sm,
                                            unsigned long long *dbg = nullptr) {
  select_topB_sync<NT, QUICK>(key, N, Bnew, Bcur, sm, (int)threadIdx.x, WorkgroupSync(), dbg);
}

// ======================================================================================================
//  fast encoder: D <= 1024, B <= NB <= 32.
//  wave w -> (dim group g = w % NG, sample stripe sw = w / NG); lane l owns dims 256 g + 4 l .. +3.
// ======================================================================================================

// ---- reduce-scatter over the 64 lanes in the canonical tree order (lane bits 5,4,3,2,1,0) -----------------
__device__ __forceinline__ void swap32(float &a, float &b) { // a[32+i] <-> b[i]
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) { // a[16+i] <-> b[i], a[48+i] <-> b[32+i]
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __uint_as_float(
      (uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(src), CTRL, 0xF, BANK, false));
}
// value of the partner lane (lane ^ DIST) for DIST in {8,4,2,1}
template <int DIST>
__device__ __forceinline__ float partner(float v) {
  if constexpr (DIST == 8) return dpp_f<0x128, 0xF>(v, v);             // row_ror:8
  else if constexpr (DIST == 4) {
    float r = dpp_f<0x104, 0x5>(v, v);                                 // row_shl:4 -> lanes with bit2 = 0 read lane+4
    return dpp_f<0x114, 0xA>(r, v);                                    // row_shr:4 -> lanes with bit2 = 1 read lane-4
  } else if constexpr (DIST == 2) return dpp_f<0x4E, 0xF>(v, v);       // quad_perm [2,3,0,1]
  else return dpp_f<0xB1, 0xF>(v, v);                                  // quad_perm [1,0,3,2]
}

// One stage of the reduce-scatter: lanes at distance DIST exchange halves of their N values and add.
// With N == 1 it degenerates into an all-reduce add (both partners end with the same bits).
template <int DIST, int N>
__device__ __forceinline__ void rs_stage(float *v, int lane) {
  if constexpr (N >= 2) {
    constexpr int H = N / 2;
    if constexpr (DIST == 32) {
#pragma unroll
      for (int j = 0; j < H; ++j) { swap32(v[j], v[j + H]); v[j] = v[j] + v[j + H]; }
    } else if constexpr (DIST == 16) {
#pragma unroll
      for (int j = 0; j < H; ++j) { swap16(v[j], v[j + H]); v[j] = v[j] + v[j + H]; }
    } else {
      const bool hi = (lane & DIST) != 0;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float keep = hi ? v[j + H] : v[j], send = hi ? v[j] : v[j + H];
        v[j] = keep + partner<DIST>(send);
      }
    }
  } else {
    static_assert(DIST <= 8, "all-reduce stages only exist inside a row");
    v[0] = v[0] + partner<DIST>(v[0]);
  }
}
constexpr int rs_half(int n) { return n >= 2 ? n / 2 : 1; }

// v[0..N0) per lane -> every lane l returns the sum over all 64 lanes of v[l * N0 / 64], added in the canonical
// tree (pairs at lane distance 32, 16, 8, 4, 2, 1).  N0 in {64, 32, 16}.
template <int N0>
__device__ __forceinline__ float reduce_scatter(float (&v)[N0], int lane) {
  static_assert(N0 == 64 || N0 == 32 || N0 == 16, "unsupported width");
  constexpr int n4 = rs_half(N0), n3 = rs_half(n4), n2 = rs_half(n3), n1 = rs_half(n2), n0 = rs_half(n1);
  rs_stage<32, N0>(v, lane);
  rs_stage<16, n4>(v, lane);
  rs_stage<8, n3>(v, lane);
  rs_stage<4, n2>(v, lane);
  rs_stage<2, n1>(v, lane);
  rs_stage<1, n0>(v, lane);
  return v[0];
}

// ---- reduce-scatter of an arbitrary number of values (2..64), same canonical tree ------------------------------
// Stage DIST halves the value count to ceil(N/2): the lane whose bit DIST is clear keeps values [0, H), its partner
// keeps [H, N) (and, for odd N, one unused slot).  The pairing of lanes (distance 32, 16, 8, 4, 2, 1) and therefore every
// rounding is the same as in reduce_scatter<64|32|16>; only which lane ends up with which value differs.
// 20 values cost 10+5+3+2+1+1 = 22 exchange+add pairs instead of the 31 of a zero-padded 32-wide reduce-scatter.
template <int DIST>
__device__ __forceinline__ float allreduce_pair(float v) { // v + (value of lane ^ DIST), both lanes get the same bits
  if constexpr (DIST == 32) { float a = v, b = v; swap32(a, b); return a + b; }
  else if constexpr (DIST == 16) { float a = v, b = v; swap16(a, b); return a + b; }
  else return v + partner<DIST>(v);
}
template <int DIST, int N>
__device__ __forceinline__ void rsn_stage(float *v, int lane) { // v has room for 2 * ceil(N / 2) values
  if constexpr (N == 1) v[0] = allreduce_pair<DIST>(v[0]);
  else {
    constexpr int H = (N + 1) / 2;
    if constexpr ((N & 1) != 0) v[N] = 0.f; // the partner of the middle value
    if constexpr (DIST == 32) {
#pragma unroll
      for (int j = 0; j < H; ++j) { swap32(v[j], v[j + H]); v[j] = v[j] + v[j + H]; }
    } else if constexpr (DIST == 16) {
#pragma unroll
      for (int j = 0; j < H; ++j) { swap16(v[j], v[j + H]); v[j] = v[j] + v[j + H]; }
    } else {
      const bool hi = (lane & DIST) != 0;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float keep = hi ? v[j + H] : v[j], send = hi ? v[j] : v[j + H];
        v[j] = keep + partner<DIST>(send);
      }
    }
  }
}
constexpr int rsn_next(int n) { return n >= 2 ? (n + 1) / 2 : 1; }
constexpr int rsn_room(int n) { return 2 * ((n + 1) / 2); }
// v must have room for rsn_room(N0) values; returns the total this lane ends up with (see rsn_owner)
template <int N0>
__device__ __forceinline__ float reduce_scatter_n(float *v, int lane) {
  static_assert(N0 >= 2 && N0 <= 64, "unsupported width");
  constexpr int n4 = rsn_next(N0), n3 = rsn_next(n4), n2 = rsn_next(n3), n1 = rsn_next(n2), n0 = rsn_next(n1);
  rsn_stage<32, N0>(v, lane);
  rsn_stage<16, n4>(v, lane);
  rsn_stage<8, n3>(v, lane);
  rsn_stage<4, n2>(v, lane);
  rsn_stage<2, n1>(v, lane);
  rsn_stage<1, n0>(v, lane);
  return v[0];
}
// index of the value whose total reduce_scatter_n<N0> leaves in this lane, or -1 if the lane holds an unused slot
template <int N0>
__device__ __forceinline__ int rsn_owner(int lane) {
  int idx = 0, cnt = N0, n = N0;
#pragma unroll
  for (int dist = 32; dist >= 1; dist >>= 1) {
    if (n >= 2) {
      const int H = (n + 1) / 2;
      if (lane & dist) { idx += H; cnt -= H; } else cnt = cnt < H ? cnt : H;
      n = H;
    }
  }
  return cnt >= 1 ? idx : -1;
}

// ---- the scoring loop's reduce-scatter: 20 values held as 10 register pairs (r02i) ---------------------------------
// Same lane tree as reduce_scatter_n<20> (pairs at lane distance 32, 16, 8, 4, 2, 1: every total has the bits the
// specification gives it); what differs is the cost and, at distance 16, which value goes with which lane:
//  * distance 32 / 16: the halves arrive by v_permlane{32,16}_swap as before, but the additions run on the register PAIRS
//    the accumulators already live in (v_pk_add_f32: 5 + 2 + 1 instructions instead of 10 + 5).  At distance 16 pair k
//    goes with pair k + 2 (k = 0, 1) and the two halves of pair 4 with each other, so a lane whose bit 4 is clear keeps
//    the values 0, 1, 2, 3, 8 of its ten and its partner 4, 5, 6, 7, 9 (rs20_owner);
//  * distance 8 / 4: ONE v_add_f32 with a DPP operand and a bank mask per kept value and lane half (the lanes whose bit is
//    clear add their partner's copy of the value they keep, the others theirs) instead of two v_cndmask, a DPP move and
//    an add -- 8 instructions instead of ~22.  Inline assembly: the compiler's DPP combiner only folds full-mask moves.
//    The s_nop in front of each group covers the "VALU write -> DPP read: 2 wait states" hazard, which the hazard
//    recogniser cannot see inside an asm statement; no instruction of a group reads a register another one of it wrote.
typedef float rs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float reduce_scatter_20(rs_f2 (&a)[10], int lane) {
  float w[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float px = a[k].x, py = a[k].y, qx = a[k + 5].x, qy = a[k + 5].y;
    swap32(px, qx); swap32(py, qy);
    a[k] = (rs_f2){px, py} + (rs_f2){qx, qy};
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float px = a[k].x, py = a[k].y, qx = a[k + 2].x, qy = a[k + 2].y;
    swap16(px, qx); swap16(py, qy);
    a[k] = (rs_f2){px, py} + (rs_f2){qx, qy};
  }
  float tx = a[4].x, ty = a[4].y;
  swap16(tx, ty);
  w[0] = a[0].x; w[1] = a[0].y; w[2] = a[1].x; w[3] = a[1].y; w[4] = tx + ty;
  float r0, r1, r2, q0, q1;
  // distance 8: 5 -> 3 values; pairs (w0, w3), (w1, w4), (w2, -)
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]));
  // distance 4: 3 -> 2 values; pairs (r0, r2), (r1, -)
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %3, %3 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(q0), "=&v"(q1) : "v"(r0), "v"(r1), "v"(r2));
  // distance 2: both candidates add their partner's copy (a quad permutation has no per-lane mask), one select picks the
  // value the lane keeps; distance 1: the closing all-reduce
  float s0, s1, tot;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(s0), "=&v"(s1) : "v"(q0), "v"(q1));
  const float kept = (lane & 2) ? s1 : s0;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(tot) : "v"(kept));
  return tot;
}
// index (0..19) of the value whose total reduce_scatter_20 leaves in this lane, or -1 (unused slot)
__device__ __forceinline__ int rs20_owner(int lane) {
  int o5 = 0, cnt = 5, n = 5;                      // distances 8, 4, 2, 1 over the five values a lane quarter keeps
#pragma unroll
  for (int dist = 8; dist >= 1; dist >>= 1) {
    if (n >= 2) {
      const int H = (n + 1) / 2;
      if (lane & dist) { o5 += H; cnt -= H; } else cnt = cnt < H ? cnt : H;
      n = H;
    }
  }
  if (cnt < 1) return -1;
  const int hi16 = (lane >> 4) & 1;
  const int v10 = o5 < 4 ? o5 + 4 * hi16 : 8 + hi16;
  return v10 + ((lane & 32) ? 10 : 0);
}

// The fast kernel addresses its LUT by ABSOLUTE LDS byte address (the table is the first thing in the dynamic LDS
// region, which starts at 0 because the kernel has no static __shared__): saves one VALU add per proposal.
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ float lds_abs_f32(uint32_t byte_addr) { return *(lds_cfloat *)(uintptr_t)(byte_addr); }

template <int NB, bool TABLE>
struct FastCfg {
  // accumulators reduced together
  static constexpr int RW = NB <= 10 ? 64 : 32;   // accumulators reduced together (32 keeps the 20/32-beam builds nearly spill-free)
  static constexpr int SPC = RW / NB;             // samples per chunk
  static_assert(SPC >= 1, "NB too large");
};

// Scratch slab of one workgroup (bytes): bp int32 [max_K][NB] | parked float [2][1024] | stats float [3][1024] | beams float [2][NB][1024]
// (parked: cumulative variance and sample scale of the three-team builds, which cannot hold them across the scoring loop)
__host__ __device__ inline size_t fast_ws_bytes(int NB, int max_K) {
  const size_t bp = (((size_t)(max_K > 0 ? max_K : 1) * NB * 4) + 255) & ~(size_t)255;
  return bp + (size_t)5 * FAST_MAX_DIM * 4 + (size_t)2 * NB * FAST_MAX_DIM * 4;
}

} // namespace irec
