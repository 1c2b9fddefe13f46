// irec_team.hip -- the default gfx950 encoder: independent TEAMS of waves inside one workgroup per CU, all of them looking
// up ONE shared set of three quantile-table copies in LDS.
//
// Hot path (reference file:line): BeamSearchCoder.encode_block rec/coding/beam_search_coder.py:53-122, as in
// irec_kernels.hip; the arithmetic specification and therefore every emitted bit are the same.
//
// What is different is how the look-up  z = quantile((r * hash) mod 10007 / 10007) = lut2[(dlog r + dlog hash) mod 10006]
// reaches the LDS (scripts/microbench/gather_rates.hip):
//   * a random 4-byte gather costs a 32-lane group as many LDS cycles as the busiest of the 32 banks has distinct
//     addresses (3.5 on average) -- the 8.9 look-ups/clk/CU ceiling the one-table encoders sit under;
//   * here the table is stored THREE times back to back (120 KB, shared by every team of the CU).  Entry
//     e = alpha' + beta with alpha' = dlog r + 10006 c, c in {0, 1}, beta = dlog hash never leaves the three copies, so
//     the address is ONE add (no "mod 10006"), and bank(e) = (dlog r + 22 c + beta) mod 32: the copy bit c moves a
//     lane by 22 banks whatever the beam.  c is chosen once per call for every (step, sample, 32-lane group, dim
//     slot) by the call's preparation kernel (choice_table_rows) -- an exact min-max assignment on two 16-rings of banks -- and travels inside the
//     proposal table the block kernel streams anyway.  Busiest bank: 2.15 addresses instead of 3.5.
//   * one workgroup owns the CU; its teams code blocks independently (own block counter pulls, own LDS scratch, own
//     scratch slab) and synchronise with team barriers (an LDS counter), never with s_barrier, so one team's serial
//     phases (top-B, beam update) overlap the other teams' scoring.
// Around the steady-state scoring loop (r02i, DESIGN.md §4 "Outside the steady state"): the first step (one beam) reduces RW
// SAMPLES per reduce-scatter instead of one; the last step forms beam 0 only; the loop's 20-value reduce-scatter runs on the
// accumulator pairs with bank-masked DPP adds (reduce_scatter_20, irec_fast_common.h); rows are fetched a chunk ahead.
// Shapes (team_shape() at the end of the file): B <= 10: three 4-wave teams (two if the LDS is short); B <= 20: three 4-wave
// teams at 168 VGPRs -- the BASELINE workload (two at 256 VGPRs for calls of one to two blocks per CU); 20 < B <= 32, or
// B <= 20 with more samples than one scoring pass holds: ONE team whose waves split the beams into stripes (12 waves x 10
// beams, 8 x 16, 8 x 10) and that scores the samples in passes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"
#include "irec_team_common.h"

#ifndef IREC_GANG_ABLATE
#define IREC_GANG_ABLATE 0   // diagnostics (make variant_gang): phases of a gang step removed -- 1 sample loops, 2 update, 4 reduction, 8 selection,
                             // 32 gang barriers; the outputs are wrong, the time that remains is the point (scripts/gang_latency.py --ablate)
#endif
#ifndef IREC_PK_ADDR
#define IREC_PK_ADDR 1          // scoring loop: look-up addresses of a beam pair by one v_pk_add_f32 (0: two v_add_u32; A/B builds)
#endif
#if defined(IREC_TEAM_MARGIN_TU) || defined(IREC_TEAM_GANG_TU)
#define IREC_TEAM_AUX_TU 1   // (irec_team_margin.hip / irec_team_gang.hip: this file once more, for their builds only)
#endif
namespace irec {

// beams a team serves at most (sizes its small LDS arrays): 32 for the builds of up to 32 beams -- the three-team 20-beam
// build has no LDS to spare -- and 64 (with room for 512 selection survivors) for the 60-beam build
__host__ __device__ constexpr int team_mb(int NB) { return NB <= 32 ? 32 : 64; }
template <int NB> using TeamLdsT = SmallLdsT<(NB <= 32 ? 32 : 64), (NB <= 32 ? 32 : 64), (NB <= 32 ? 64 : 512)>;
__host__ __device__ inline size_t team_small_bytes(int NB) {
  return NB <= 32 ? ((sizeof(TeamLdsT<32>) + 15) & ~(size_t)15) : ((sizeof(TeamLdsT<64>) + 15) & ~(size_t)15);
}
// Sample passes.  The per-group partial scores of SP samples sit in LDS at a time ([4][SP][NB] f32); a step scores S in
// ceil(S / SP) passes, each followed by the group combine into the sort keys ([S*NB] u32, all of them resident).  Every
// BASELINE configuration with B <= 20 takes one pass; the 30-beam stress configuration (S = 148) takes four of 37.
// (`ps` below: beam slots per sample in the partial-score rows and the key array -- NB, or 1 when the call has ONE beam
//  (B = 1 of the reference's sweep: a 10-beam layout would spend ten times the LDS per sample and ten times the passes))
__host__ __device__ inline int team_row(int NB, int B) { return (NB == 10 && B == 1) ? 1 : NB; }
__host__ __device__ inline size_t team_key_bytes(int ps, int S) { return (((size_t)S * ps * 4) + 15) & ~(size_t)15; }
// Keys in LDS unless they alone would leave room for fewer than 16 samples of partials (single-team builds only: 12 090
// candidates of B = 30, S = 403 are 48 KB); then they live in the team's scratch slab (L2) and the selection scans them there.
// (`passes`: a multi-team build that scores S in passes -- round 3, B <= 10 with more samples than one pass holds; the
//  other multi-team builds take S in one pass and keep their keys in LDS by construction)
__host__ __device__ inline bool team_keys_in_lds(int NB, int S, int teams, bool passes, int ps = 0) {
  if (ps <= 0) ps = NB;
  const long long avail = (long long)((FAST_LDS_LIMIT - T3_BYTES) / (size_t)teams) - (long long)team_key_bytes(ps, S) -
                          (long long)team_small_bytes(NB) - 32;
  return (teams > 1 && !passes) || avail / (4LL * ps * 4) >= (S < 16 ? S : 16);
}
// Three 20-beam teams only fit the 160 KB next to the table copies if the sort keys are written over group 0 of the partial
// scores (key f = s * Bcur + b lands on partial s * NB + b: the same word when Bcur == NB, which every step but the first
// has; otherwise a barrier separates the partial reads from the key writes).
__host__ __device__ inline bool team_keys_alias(int NB, int teams) { return teams >= 3 && NB == 20; }
__host__ __device__ inline int team_s_pass(int NB, int S, int teams, int cmax, bool passes = false, int ps = 0) {
  if (ps <= 0) ps = NB;
  const long long avail = (long long)((FAST_LDS_LIMIT - T3_BYTES) / (size_t)teams) -
                          ((team_keys_in_lds(NB, S, teams, passes, ps) && !(team_keys_alias(NB, teams) && !passes)) ? (long long)team_key_bytes(ps, S) : 0) -
                          (long long)team_small_bytes(NB) - 32;
  long long fit = avail / (4LL * ps * 4);           // samples whose partials fit
  if (fit > cmax / ps) fit = cmax / ps;             // and whose candidates one combine round covers
  if (fit < 1) return 0;
  if (fit >= S) return S;
  const int n_pass = (int)((S + fit - 1) / fit);
  return (S + n_pass - 1) / n_pass;                 // balanced passes
}
__host__ __device__ inline size_t team_part_bytes(int ps, int SP) { return (((size_t)4 * SP * ps * 4) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t team_lds_one(int NB, int S, int SP, int teams, bool passes = false, int ps = 0) {
  if (ps <= 0) ps = NB;
  return team_part_bytes(ps, SP) + ((team_keys_in_lds(NB, S, teams, passes, ps) && !(team_keys_alias(NB, teams) && !passes)) ? team_key_bytes(ps, S) : 0) +
         team_small_bytes(NB) + 16;
}
__host__ __device__ inline size_t team_lds_total(int NB, int S, int SP, int teams, bool passes = false, int ps = 0) {
  return T3_BYTES + (size_t)teams * team_lds_one(NB, S, SP, teams, passes, ps);
}

// Diagnostic build (-DIREC_TEAM_STAMPS, scripts/gpu_stamps.sh): per-wave cycle sums per phase, written once at exit.
#ifdef IREC_TEAM_STAMPS
#define TSTAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define TSTAMP(slot) do { } while (0)
#endif

// LDS carve (bytes): lut2 x 3 [120080] | team 0: part [4][S][NB] f32 | sort keys [S*NB] | TeamLds | barrier | team 1: same
// BS = beam stripes: with BS == 2 a team has 8 waves, wave w serves dim group / sample stripe (w & 3) and the beams
// [NBW * (w >> 2), + NBW), NBW = NB / BS: half the G registers per lane (128-VGPR budget, 16 waves per CU).
// PASSES: a multi-team build that scores the S samples of a step in several passes and may keep its sort keys in the slab
// (the single-team builds always can): S beyond what a team's share of the LDS holds -- the reference's sweep reaches
// S = int(e^9) = 8103 (examples/lossless/data_aggregation.py:5-7).
// ONE: the build of one-beam calls (B = 1 of the reference's sweep): every step takes the wide path, which is therefore the
// pipelined form (rows a half batch ahead, two samples per v_pk_fma_f32) -- in the other builds only step 0 comes there and
// the extra row buffers cost registers everywhere else (r03e A/B: B = 32 -14 %, B = 30 -2 %, B = 10 -3 % with it).
// SHARE: the build of calls whose rows beyond one per CU are shared between teams (A.coop_W > 1, below) -- its own instantiation:
// the exchange code costs the two-team builds their spill-free register allocation (r04: <20,2,1> 80 -> 400 B of scratch with it).
// MARGIN (round 5): the build behind irec_beam_encode_ex / IREC_FLAG_MARGINS -- after every selection wave 0 also takes the gap between
// the last candidate kept and the best one rejected (margin_step, irec_fast_common.h) and the block leaves four floats in
// A.out_margin; one team barrier more per step.  Instantiated in its own translation unit (irec_team_margin.hip) for the shapes the
// BASELINE configurations run; every other call with margins takes the generic kernel.  Same indices and samples, bit for bit.
template <int NB, int TEAMS, int BS, bool PASSES = false, bool ONE = false, bool SHARE = false, bool MARGIN = false>
__global__ __launch_bounds__(TEAMS * BS * TEAM_NT, 1) void encode_team_kernel(EncArgs A) {
  using TeamLds = TeamLdsT<NB>;
  constexpr int TEAM_MB = team_mb(NB);
  constexpr size_t TEAM_SMALL_BYTES = (sizeof(TeamLds) + 15) & ~(size_t)15;
  constexpr bool MULTI_PASS = TEAMS == 1 || PASSES;
  constexpr bool SHORT = TEAMS * BS >= 3 && NB / BS == 20;   // 12 waves per CU x 20 beams per wave: the 168-VGPR builds
  constexpr int NW = TEAM_NW;                    // waves per beam stripe (dim groups x sample stripes)
  constexpr int NWT = TEAM_NW * BS, NT = 64 * NWT; // waves / threads per team
  constexpr int NBW = NB / BS;                   // beams per wave
  static_assert(NB % BS == 0, "beam stripes must divide the beam count");
  // 20 accumulators are reduced together (one sample x 20 beams, or two samples x 10 beams) by the 20-value
  // reduce-scatter: 22 exchange+add pairs instead of the 31 of a zero-padded 32-wide one, and 12 registers fewer
  static_assert(NBW == 10 || NBW == 16 || NBW == 18 || NBW == 20, "the team encoder is built for 10, 16, 18 or 20 beams per wave");
  static_assert(NB <= TEAM_MB && NB <= 64, "beam indices are lane indices and 6-bit back-pointers");
  constexpr int CMAX = TEAMS == 1 ? 2048 : 1024;    // candidates one combine round covers (host: SP * NB <= CMAX)
  constexpr int SPC = BS >= 2 ? 1 : 20 / NBW;   // samples per chunk (beam-striped builds: one, to fit 128 VGPRs)
  constexpr int RW = NBW * SPC;                  // accumulators reduced together
  constexpr int ACC_ROOM = rsn_room(RW);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = A.S, B = A.B;
  const int lane = threadIdx.x & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform by construction
  const int team = wave_wg / NWT, wave_t = wave_wg % NWT;
  const int wave = wave_t & 3, bs = wave_t >> 2;                                // quad position, beam stripe
  const int b_lo = bs * NBW;                                                    // first beam of my stripe
  const int tid = (int)threadIdx.x - team * NT;                                 // index inside the team
  // samples scored per pass: only the single-team builds take more than one pass (the host checked that the others fit
  // S in one -- then the pass loop below folds away)
  const int PS = NB == 10 ? team_row(NB, B) : NB;                              // beam slots per sample row (1 for a one-beam call)
  const int SP = MULTI_PASS ? team_s_pass(NB, S, TEAMS, CMAX, PASSES, PS) : S;
  const bool keys_lds = !MULTI_PASS || team_keys_in_lds(NB, S, TEAMS, PASSES, PS); // (compile-time true for the one-pass multi-team builds)
  constexpr bool KEYS_ALIAS = TEAMS >= 3 && NB == 20 && !PASSES;               // team_keys_alias()
  const size_t key_lds_bytes = (keys_lds && !KEYS_ALIAS) ? team_key_bytes(PS, S) : 0;
  char *tbase = smem + T3_BYTES + (size_t)team * team_lds_one(NB, S, SP, TEAMS, PASSES, PS);
  float *part_s = reinterpret_cast<float *>(tbase);                             // [4][SP][PS] per-group partial scores
  TeamLds *sm = reinterpret_cast<TeamLds *>(tbase + team_part_bytes(PS, SP) + key_lds_bytes);
  uint32_t *bar_word = reinterpret_cast<uint32_t *>(tbase + team_part_bytes(PS, SP) + key_lds_bytes + TEAM_SMALL_BYTES);
  double *gpart = sm->gpart;
  int32_t *sel_s = sm->sel_s, *sel_b = sm->sel_b;
  int32_t *hsum = &sm->hsum[0][0];
  uint32_t *beta4 = &sm->beta4[0][0];
  int32_t *misc = sm->misc;
  float *cpart_s = &sm->cpart[0][0];
  const uint16_t *dlog_s = A.dlog4r;                                            // [10006] 4*dlog(j+1), global (L2)
  const int rs_p = rsn_owner<RW>(lane);                                         // accumulator whose total reduce_scatter_n<20> leaves here
  const int rs_p20 = RW == 20 ? rs20_owner(lane) : -1;                          // same for the scoring loop's reduce_scatter_20
  const int rs_c = rsn_owner<NBW>(lane);                                        // same for the NBW C_b partials of the update

  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  commit_table_stamps(A);
  constexpr bool CAN_SHARE = SHARE && TEAMS >= 2 && BS == 1 && !PASSES && !ONE;
  const int Wsh = (CAN_SHARE && A.coop_W > 1) ? A.coop_W : 1;
  const int64_t n_whole = Wsh > 1 ? (A.tsplit_first < A.n_blocks ? A.tsplit_first : A.n_blocks) : A.n_blocks;
  const int64_t n_slots = n_whole + (A.n_blocks - n_whole) * Wsh;
  const int64_t n_static = (int64_t)TEAMS * (int64_t)gridDim.x < n_slots ? (int64_t)TEAMS * (int64_t)gridDim.x : n_slots;
  // Cost-ordered hand-out (round 4; A.row_cost: distinct keys (K * dims) << 10 | row from the call's preparation kernel, set by the host only
  // when the static round deals every slot): slot u < lo = min(grid, n_whole) is the row of ascending cost rank u, slot lo + k the row
  // of rank n_blocks - 1 - k / W -- a workgroup's first team gets a cheap row, its other teams (or the teams that share a row) the
  // costliest ones, longest first.  Every workgroup ranks the rows itself, once, in the LDS the table copies are about to fill
  // (one row per thread, n compares each: ~2 us): no second kernel, no grid-wide wait.
  constexpr bool CAN_PLACE = NB == 20 && TEAMS == 2 && BS == 1 && !PASSES && !ONE;   // (team_placeable(): the builds the host asks it of)
  // (n_static == n_slots: the static round deals every slot, which the ranking below assumes -- the host only sets row_cost then
  //  (irec_host.cpp); should a later change break that, the rows are dealt as listed instead of the slots beyond the static round being lost)
  const bool placed = CAN_PLACE && A.row_cost != nullptr && n_static == n_slots;
  const int64_t lo_rank = (int64_t)gridDim.x < n_whole ? (int64_t)gridDim.x : n_whole;
  int32_t placed_row = -1;
  if (placed) {
    uint32_t *cs = reinterpret_cast<uint32_t *>(smem);
    const int n = (int)A.n_blocks;
    for (int k = (int)threadIdx.x; k < n; k += TEAMS * NT) cs[k] = A.row_cost[k];
    if ((int)threadIdx.x < TEAMS) cs[n + (int)threadIdx.x] = 0xFFFFFFFFu;
    int want[TEAMS];
#pragma unroll
    for (int tm = 0; tm < TEAMS; ++tm) {
      const int64_t u = (int64_t)tm * (int64_t)gridDim.x + (int64_t)blockIdx.x;
      want[tm] = u >= n_static ? -1 : u < lo_rank ? (int)u : n - 1 - (int)((u - lo_rank) / (u < n_whole ? 1 : Wsh));
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < n; i += TEAMS * NT) {
      const uint32_t ci = cs[i];
      int rk = 0;
#pragma unroll 8
      for (int j = 0; j < n; ++j) rk += cs[j] < ci ? 1 : 0;
#pragma unroll
      for (int tm = 0; tm < TEAMS; ++tm)
        if (rk == want[tm]) cs[n + tm] = (uint32_t)i;
    }
    __syncthreads();
    placed_row = (int32_t)cs[n + team];
    __syncthreads();   // (the table copies overwrite the keys)
  }
  {
    float *l3 = reinterpret_cast<float *>(smem);
    for (int k = (int)threadIdx.x; k < (int)IREC_PM1; k += TEAMS * NT) {
      const float v = A.lut2[k];
      l3[k] = v; l3[k + IREC_PM1] = v; l3[k + 2 * IREC_PM1] = v;
    }
    if (tid == 0) *bar_word = 0u;
  }
  __syncthreads(); // the only workgroup-wide barrier: from here on the teams never wait for each other
  TeamBarrier tsync{bar_word, 0u, (uint32_t)NWT};

  char *slab = A.ws + ((size_t)blockIdx.x * TEAMS + team) * A.ws_per_wg;
  const size_t key_glb_bytes = keys_lds ? 0 : ((team_key_bytes(PS, S) + 255) & ~(size_t)255);   // sort keys at the slab's end
  uint32_t *key_s = KEYS_ALIAS ? reinterpret_cast<uint32_t *>(tbase)                            // over partial group 0
                    : keys_lds ? reinterpret_cast<uint32_t *>(tbase + team_part_bytes(PS, SP))     // [S*PS] sort keys
                               : reinterpret_cast<uint32_t *>(slab + A.ws_per_wg - key_glb_bytes);
  int32_t *bp = reinterpret_cast<int32_t *>(slab);                                            // [max_K][NB]
  float *beams_g = reinterpret_cast<float *>(slab + A.ws_per_wg - key_glb_bytes - (size_t)2 * NB * FAST_MAX_DIM * 4); // [2][NB][1024]
  float *stats_g = beams_g - 3 * FAST_MAX_DIM;  // [3][1024]: mq - mp, sq^2, sp^2 of the block, coalesced
  float *park_g = stats_g - 2 * FAST_MAX_DIM;   // [2][1024]: c (cumulative variance) and sa of the PARK builds
  // Three-team builds have 168 VGPRs: the cumulative variance and the sample scale (needed only by the update) are parked
  // in the slab across the scoring loop instead of being spilled inside it
  constexpr bool PARK = SHORT;
  constexpr bool LATE_G = false;   // (tried: new beams wait in G's registers and G is formed after the batches -- the
                                   //  register allocator then shuffles eight registers through scratch per beam: 844 B)

#ifdef IREC_TEAM_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long st_t0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();   // shader clock = d(memtime) / d(memrealtime) x 100 MHz
#endif
  // Block hand-out: the first block of team k of workgroup w is k * gridDim.x + w -- a call of fewer blocks than resident
  // teams (a mid-size batch: 38 images x 9 blocks) puts ONE block on every CU before any CU gets a second one, instead of
  // three on the first third of the CUs; later blocks come from the atomic counter, which starts behind the static ones.
  // Both rounds deal the rows of a tensor to one XCD (irec_fast_common.h: xcd_static_row / xcd_pull_row).
  // Shared blocks (round 4; calls of one to two blocks per CU): the rows from A.tsplit_first on are coded by A.coop_W teams EACH --
  // anywhere on the chip -- that split the row's SAMPLES: team q of a row scores the samples of stripe q, publishes their sort
  // keys as tagged granules (the split encoder's exchange, irec_kernels.hip), sweeps its partners' and then runs the same
  // selection and the same update as they do, on its own slab: nothing but keys is shared.  A 342-block call then loads a CU
  // with one whole block and a fraction of another instead of two whole ones.  Hand-out slots: one per whole row, coop_W per
  // shared row; the host sizes coop_W so that the static round deals every slot (all partners resident at once).
  bool first_block = true;
  int steal = 0;
  for (;;) {
    tsync();
    if (tid == 0) {
      int64_t r;
      if (first_block) {
        r = (int64_t)team * (int64_t)gridDim.x + (int64_t)blockIdx.x;
        r = r < n_static ? (placed ? r : xcd_static_row(r, n_static, (int)gridDim.x)) : n_slots;
      } else r = placed ? n_slots : xcd_pull_row(A, n_static, n_slots, steal);
      misc[0] = (int32_t)r;
      misc[6] = 0;                          // (shared rows: a partner gave up)
    }
    first_block = false;
    tsync();
    const int64_t slot = misc[0];
    TSTAMP(0);
    if (slot >= n_slots) break; // every wave of the team reaches this; the other team drains on its own
    int64_t blk = slot < n_whole ? slot : n_whole + (slot - n_whole) / Wsh;
    const int64_t xrow = slot < n_whole ? 0 : (slot - n_whole) / Wsh;        // exchange slot of a shared row
    const int qsh = slot < n_whole ? 0 : (int)((slot - n_whole) % Wsh);      // my sample stripe of a shared row
    const int Wrow = slot < n_whole ? 1 : Wsh;                               // teams that code this row
    if (placed) {
      blk = placed_row;
      if (blk < 0 || blk >= A.n_blocks) __builtin_trap();   // (the cost keys are distinct: their ranks are a permutation of the rows)
    }
    if (Wrow > 1 && A.coop_test_orphan && qsh != 0) continue;                // test hook: team 0 of a shared row is left waiting
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const uint16_t *tab = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (A.tab_dim[q] == D) tab = A.tab[q];
    if (D < 1 || D > FAST_MAX_DIM || tab == nullptr) { // host promised D <= 1024 and listed dims
      if (tid == 0 && qsh == 0) A.out_K[blk] = -1;
      continue;
    }
    const int Dp = (D + 3) & ~3;            // row stride of the proposal table
    const int NG = (D + 255) >> 8;          // 1..4 dim groups
    const int NSW = NW / NG;                // sample stripes
    const bool active = wave < NG * NSW;
    const int g = wave % NG, sw = wave / NG;
    const int NSWe = NSW * Wrow, swe = sw * Wrow + qsh;   // ... of the row over all the teams that code it: my samples are swe, swe + NSWe, ...
    const int d0 = g * 256 + lane * 4;

    // ---- my 4 dims (split == gather through perm) and the block's KL ----
    float c[4];
    bool valid[4];
    // the wave that reads a dim group's statistics keeps them in registers; only the other waves of the group (sample /
    // beam stripes of short blocks and of the striped builds) re-read them from the slab every step
    float own_dmu[4], own_vq[4], own_vp[4];
    const bool own_stats = active && sw == 0 && bs == 0 && TEAMS < 3 && !SHORT;   // (168-VGPR builds: the slab keeps them)
    double klacc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = d0 + i;
      valid[i] = d < D;
      c[i] = 0.f;
      // (the element's index is looked up again when the sample is written: four 64-bit indices per lane are not worth
      // holding, or spilling, across the whole step loop)
      const int64_t ixi = valid[i] ? src_index(A, base, pos, d) : src_index(A, base, pos, 0);
      float st3[3] = {0.f, 1.f, 1.f};
      if (valid[i] && active && sw == 0 && bs == 0) { // one wave per dim group does the float64 KL and publishes the statistics
        const float mq_ = A.q_loc[ixi], sq_ = A.q_scale[ixi], mp_ = A.p_loc[ixi], sp_ = A.p_scale[ixi];
        klacc = klacc + kl_dim(mq_, sq_, mp_, sp_);
        st3[0] = mq_ - mp_; st3[1] = sq_ * sq_; st3[2] = sp_ * sp_;
      }
      own_dmu[i] = st3[0]; own_vq[i] = st3[1]; own_vp[i] = st3[2];
      if (active && sw == 0 && bs == 0 && (NSW > 1 || BS > 1 || TEAMS >= 3)) { // somebody else (or a later step) needs them
        stats_g[d0 + i] = st3[0]; stats_g[FAST_MAX_DIM + d0 + i] = st3[1]; stats_g[2 * FAST_MAX_DIM + d0 + i] = st3[2];
      }
    }
    {
      const double gs = wave_tree_sum(klacc);
      if (sw == 0 && bs == 0 && active && lane == 0) gpart[g] = gs;
      tsync();
      if (tid == 0) {
        double tot = gpart[0];
        for (int gg = 1; gg < NG; ++gg) tot = tot + gpart[gg];
        const int32_t K = num_aux((float)tot, A.omega);
        misc[1] = K;
        if (qsh == 0) A.out_K[blk] = K;
        hsum[0] = 0;
        beta4[0] = 0u; // hash of the empty path is 1 = g^0
        if constexpr (MARGIN) { if (qsh == 0) margin_write(A.out_margin, blk, MarginAcc()); }   // (a block that is not coded here keeps "no comparison")
      }
      tsync();
    }
    const int K = misc[1];
    MarginAcc macc;
    if (K > A.max_K || K > A.K_limit) continue;
    if (K > A.K_tab) { // beyond the table window: the fused-Philox pass codes it
      if (tid == 0 && qsh == 0) atomicAdd(A.defer_count, 1u);
      continue;
    }
    if (K == 0) { // nothing to code: sample = p.loc
      if (active && sw == 0 && bs == 0 && qsh == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (valid[i]) { const int64_t ixo = src_index(A, base, pos, d0 + i); A.out_sample[ixo] = 0.f + A.p_loc[ixo]; }
      }
      continue;
    }

    float sa[4], cH[4];
    float G[NBW][4];
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
        __builtin_amdgcn_sched_barrier(0); // one dim at a time: the division sequences are register hungry
      }
    };
    auto park = [&]() {     // (waves of the same dim group write the same bits)
      if (PARK && active) {
        *reinterpret_cast<float4 *>(park_g + d0) = make_float4(c[0], c[1], c[2], c[3]);
        *reinterpret_cast<float4 *>(park_g + FAST_MAX_DIM + d0) = make_float4(sa[0], sa[1], sa[2], sa[3]);
      }
    };
    auto unpark = [&]() {
      if (PARK && active) {
        asm volatile("" ::: "memory");   // a real reload: the values are dead between park() and here
        const float4 pc = *reinterpret_cast<const float4 *>(park_g + d0);
        const float4 ps = *reinterpret_cast<const float4 *>(park_g + FAST_MAX_DIM + d0);
        c[0] = pc.x; c[1] = pc.y; c[2] = pc.z; c[3] = pc.w;
        sa[0] = ps.x; sa[1] = ps.y; sa[2] = ps.z; sa[3] = ps.w;
      }
    };
    // ---- prologue: step 0 has one (all-zero) beam ----
    {
      float m[4], cA[4], cBv[4];
      step_consts(0, m, cA, cBv);
      park();
      float cacc = 0.f;
#pragma unroll
      for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) G[b][i] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (bs == 0) G[0][i] = beam_G(0.f, m[i], cA[i], cBv[i], sa[i]);
        cacc = beam_C_term(cacc, 0.f, m[i], cA[i], cBv[i]);
      }
      const float cg = wave_tree_sum(cacc);
      if (active && sw == 0 && bs == 0 && lane == 0) cpart_s[g * TEAM_MB + 0] = cg;
      // (visibility of the C_b partials: the barrier after scoring)
    }

    TSTAMP(1);
    bool abandoned = false;
    int cur = 0, Bcur = 1;
    // 4 * dlog(hash) of the live beams, lane j = beam j: every wave carries its own copy from the update into the next step's
    // scoring (r04: the table look-up behind it is a GLOBAL load; it used to sit inside the one-wave selection, a round trip
    // to the L2 on the path every wave of the team waits on)
    uint32_t bv_cur = 0u;                                            // hash of the empty path is 1 = g^0
    for (int t = 0; t < K; ++t) {
#ifdef IREC_TEAM_STAMPS
      st_acc[11] += 1ull;                                            // (diagnostic build: block-steps of this wave)
#endif
      const uint16_t *tab_tu = tab + (size_t)t * S * Dp;             // uniform base of this step's rows
      // my quad inside a row (32-bit offsets: one VGPR).  Lanes past the padded row end (all their dims invalid, zero
      // coefficients) read the row's LAST quad: finite z, and the same addresses as the last real lane of their 32-lane
      // group -- the LDS serves identical addresses as one broadcast, so they add no bank conflict to the look-ups
      const uint32_t tab_lo = (uint32_t)(d0 < Dp ? d0 : Dp - 4);
      const uint16_t *tab_t = tab_tu + tab_lo;
      uint32_t bet[NBW];
      {
#pragma unroll
        for (int b = 0; b < NBW; ++b) bet[b] = (uint32_t)__builtin_amdgcn_readlane((int)bv_cur, b_lo + b);
      }
      const int nlive = Bcur - b_lo < 0 ? 0 : (Bcur - b_lo < NBW ? Bcur - b_lo : NBW);   // live beams of my stripe

      const int N = S * Bcur;
      for (int s_base = 0; s_base < S; s_base += SP) {
      const int s_end = s_base + SP < S ? s_base + SP : S;
      const int Sp = s_end - s_base;                                // samples of this pass
      // ---------------- scoring: samples [s_base, s_end) x Bcur candidates (beam_search_coder.py:80-84) ----------------
#ifdef IREC_ABLATE_SCORING
      if (false) {
#else
      // Steady state: every beam of my stripe alive -- or at least half of them (B = 50 on the 60-beam build: the third stripe
      // holds beams 40..49): the dead ones are scored as PHANTOM beams (G = 0, the address of beam 0: finite values that no
      // candidate ever reads) rather than sending the whole stripe down the beam-by-beam path below.
      if (active && Bcur > 1 && (nlive == NBW || 2 * nlive >= NBW)) {
        // Steady state, software pipelined by dim slot: the NBW gathers of the NEXT slot (of this sample, of the chunk's
        // next sample, or of the next chunk's first sample) are issued before the current slot's values are consumed, so
        // the wave always has look-ups in flight -- also under the fma chain and the reduce-scatter.  A wave can have at
        // most 15 LDS operations outstanding, so by the time the next slot's are issued the current slot's have landed
        // (or the compiler's wait counts see to it).  Empty volatile asm statements pin the order (pure arithmetic would
        // otherwise drift across the scheduling barriers).  A chunk is SPC samples (20 accumulators, one reduce-scatter).
        const int n_mine = Sp > swe ? (Sp - swe + NSWe - 1) / NSWe : 0; // my samples of the pass: s_base + swe, + NSWe, ...
        const int n_chunks = (n_mine + SPC - 1) / SPC;
        auto row = [&](int m) {                                     // proposal row of my m-th sample; zero row past the end:
          uint2 r = make_uint2(0u, 0u);                             // entry 0 is a valid address, its results are dropped
          if (m < n_mine) r = *reinterpret_cast<const uint2 *>(tab_tu + ((uint32_t)(s_base + m * NSWe + swe) * (uint32_t)Dp + tab_lo));
          return r;
        };
        // proposal rows stay packed (4 x uint16 per sample in two registers); the byte address of a dim slot's entry in copy 0
        // is unpacked when its look-ups are issued (one live temporary instead of four registers per sample)
        uint2 ap_cur[SPC], ap_nxt[SPC];
#pragma unroll
        for (int cc = 0; cc < SPC; ++cc) { ap_cur[cc] = row(cc); ap_nxt[cc] = row(SPC + cc); }
#define IREC_AL(CC, I) ((((I) & 2) ? (((I) & 1) ? (ap_cur[CC].y >> 16) : (ap_cur[CC].y & 0xFFFFu)) : (((I) & 1) ? (ap_cur[CC].x >> 16) : (ap_cur[CC].x & 0xFFFFu))) << 2)
        // values travel in register pairs so that the two fma of two beams stay one v_pk_fma_f32 each
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int NP = NBW / 2;
        constexpr int NQ = 4 * SPC;                                 // dim slots per chunk (even: buffers alternate cleanly)
        static_assert(NBW % 2 == 0, "beams are processed in pairs");
        // Pipeline granule: a whole dim slot (NP beam pairs per look-up buffer), or HALF a slot for the three-team
        // 20-beam build (5 pairs per buffer: 20 instead of 40 look-up registers -- three waves per SIMD hide what the
        // shallower pipeline exposes).
        constexpr int NH = SHORT ? 2 : 1;                           // granules per dim slot
        constexpr int HP = NP / NH;                                 // beam pairs per granule
        constexpr int NQH = NQ * NH;                                // granules per chunk
        static_assert(NP % NH == 0 && NQH % 2 == 0, "granules must tile the slots and alternate buffers cleanly");
        f2 zz[2][HP];
#if IREC_PK_ADDR
        // round 6: the two addresses of a beam pair come out of ONE v_pk_add_f32 on their bit patterns -- byte addresses below 2^23 read as
        // float32 are denormals, denormals are preserved (the arithmetic contract needs them), and the sum of two such numbers is exact and
        // carries the integer sum in its bits: half the address instructions (irec_ten.hip has the same)
        f2 betf[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) betf[k] = (f2){__uint_as_float(bet[2 * k]), __uint_as_float(bet[2 * k + 1])};
#define IREC_ISSUE(Z, AD, K0) do { const float adf_ = __uint_as_float(AD); const f2 ad2_ = {adf_, adf_}; \
                               _Pragma("unroll") for (int k = 0; k < HP; ++k) { const f2 a2_ = ad2_ + betf[(K0) + k]; \
                                 Z[k].x = lds_abs_f32(__float_as_uint(a2_.x)); Z[k].y = lds_abs_f32(__float_as_uint(a2_.y)); } \
                               __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IREC_ISSUE(Z, AD, K0) do { _Pragma("unroll") for (int k = 0; k < HP; ++k) { Z[k].x = lds_abs_f32((AD) + bet[2 * ((K0) + k)]); Z[k].y = lds_abs_f32((AD) + bet[2 * ((K0) + k) + 1]); } \
                               __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
#define IREC_CONSUME(Z, I, ACC, K0) do { _Pragma("unroll") for (int k = 0; k < HP; ++k) asm volatile("" : "+v"(Z[k])); \
                                f2 t2_[HP]; \
                                _Pragma("unroll") for (int k = 0; k < HP; ++k) { \
                                  const f2 h2 = {cH[I], cH[I]}, g2 = {G[2 * ((K0) + k)][I], G[2 * ((K0) + k) + 1][I]}; \
                                  t2_[k] = __builtin_elementwise_fma(h2, Z[k], g2); } /* inner fma of every pair first: */ \
                                _Pragma("unroll") for (int k = 0; k < HP; ++k) /* no dependent back-to-back issue */ \
                                  ACC[(K0) + k] = __builtin_elementwise_fma(t2_[k], Z[k], ACC[(K0) + k]); \
                                _Pragma("unroll") for (int k = 0; k < HP; ++k) asm volatile("" : "+v"(ACC[(K0) + k])); \
                                __builtin_amdgcn_sched_barrier(0); } while (0)
        if constexpr (PARK) { // every G of the wave in a register HERE: a spilled one is reloaded once per step, not per sample
#pragma unroll
          for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(G[b][i]));
        }
        IREC_ISSUE(zz[0], IREC_AL(0, 0), 0);
        for (int ch = 0; ch < n_chunks; ++ch) {
          if constexpr (BS >= 2) {
            // One team, BS waves per SIMD (the stripes' waves of one dim group share a SIMD): the SIMD's arbiter favours its
            // oldest wave, which then reaches the pass's barrier a third of the pass early and leaves the SIMD to two waves,
            // then one (r03g stamps: 20-28 % of every wave's time is that wait).  The favoured stripe rotates instead.
            if ((((ch >> 2) + bs) % BS) == 0) __builtin_amdgcn_s_setprio(1);   // (every four chunks)
            else __builtin_amdgcn_s_setprio(0);
          }
          f2 acc2[SPC][NP];
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
            for (int k = 0; k < NP; ++k) acc2[cc][k] = (f2){0.f, 0.f};
          uint2 ap_new[SPC];                                        // rows of the chunk after next: a whole chunk of lead
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc) ap_new[cc] = row((ch + 2) * SPC + cc);
#pragma unroll
          for (int qh = 0; qh < NQH; ++qh) {
            constexpr int dummy_ = 0; (void)dummy_;
            const int q = qh / NH, h = qh % NH;                     // dim slot of the chunk, granule of the slot
            if (qh + 1 < NQH) {
              const int qn = (qh + 1) / NH, hn = (qh + 1) % NH;
              IREC_ISSUE(zz[(qh + 1) & 1], IREC_AL(qn >> 2, qn & 3), hn * HP);
            } else {
              // next chunk's rows
#pragma unroll
              for (int cc = 0; cc < SPC; ++cc) { ap_cur[cc] = ap_nxt[cc]; ap_nxt[cc] = ap_new[cc]; }
              IREC_ISSUE(zz[0], IREC_AL(0, 0), 0);
            }
            IREC_CONSUME(zz[qh & 1], q & 3, acc2[q >> 2], h * HP);
          }
          float tot;
          int own;                                                  // value of the chunk whose total this lane ends up with
          if constexpr (RW == 20) {               // the accumulator pairs go in as they are
            rs_f2 a20[10];
#pragma unroll
            for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
              for (int k = 0; k < NP; ++k) a20[cc * NP + k] = acc2[cc][k];
            tot = reduce_scatter_20(a20, lane);
            own = rs_p20;
          } else {
            float acc[ACC_ROOM];
#pragma unroll
            for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
              for (int k = 0; k < NP; ++k) { acc[cc * NBW + 2 * k] = acc2[cc][k].x; acc[cc * NBW + 2 * k + 1] = acc2[cc][k].y; }
            tot = reduce_scatter_n<RW>(acc, lane);
            own = rs_p;
          }
          const int cc = own / NBW, b = own - cc * NBW;            // own < 0: unused slot
          const int m = ch * SPC + cc;                              // my m-th sample
          if (own >= 0 && (lane & 1) == 0 && m < n_mine && b < nlive) part_s[((size_t)g * SP + m * NSWe + swe) * PS + b_lo + b] = tot;
        }
        if constexpr (BS >= 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int k = 0; k < HP; ++k) asm volatile("" : "+v"(zz[0][k])); // drain the look-ups issued past the last sample
#undef IREC_ISSUE
#undef IREC_CONSUME
#undef IREC_AL
      } else if (active && Bcur == 1 && bs == 0) {
        // First step (beam_search_coder.py:97-106): ONE beam, so a sample is a single candidate.  Scored beam-wise it would
        // pay a whole reduce-scatter (86 instructions) for 4 look-ups; here RW SAMPLES share one -- a sample sits where a
        // beam sits in the steady state, every total still comes out of the same lane chain and lane tree.  36 samples: two
        // rounds instead of 36 (r02i: step 0 cost half a full step's instructions for a twentieth of its look-ups).
        const int n_mine = Sp > swe ? (Sp - swe + NSWe - 1) / NSWe : 0; // my samples of the pass: s_base + swe, + NSWe, ...
        const uint32_t bet0 = bet[0];
        if constexpr (ONE) {   // (the one-beam builds: every step comes here)
        // two half batches of rows (even sizes: samples go through the fma in pairs), 20 registers as the steady state's
        constexpr int HA = ((RW / 2) + 1) & ~1, HBb = RW - HA;
        static_assert(RW % 2 == 0 && HBb >= 2 && HBb % 2 == 0, "half batches of sample pairs");
        // round 3 (one-beam calls live here for EVERY step): the rows of the next half batch are in flight under the current
        // one's look-ups (they used to be fetched and waited for batch by batch), and two samples share each v_pk_fma_f32
        typedef float f2w __attribute__((ext_vector_type(2)));
        // Row of my m-th sample: a wave-uniform base (scalar arithmetic) plus my quad's byte offset -- no vector address
        // arithmetic per row; past my last sample the LAST row is read again (its totals are dropped at the store).
        const uint32_t lane_off = tab_lo * 2u;
        auto rows = [&](int m_first, auto &ap) {
          constexpr int H = (int)(sizeof(ap) / sizeof(ap[0]));
#pragma unroll
          for (int k = 0; k < H; ++k) {
            int m = m_first + k;
            m = m < n_mine ? m : n_mine - 1;
            const char *rowp = reinterpret_cast<const char *>(tab_tu) + (size_t)((uint32_t)(s_base + m * NSWe + swe) * (uint32_t)Dp) * 2u;
            ap[k] = *reinterpret_cast<const uint2 *>(rowp + lane_off);
          }
        };
        auto half = [&](const auto &ap, f2w *acc2) {
          constexpr int H = (int)(sizeof(ap) / sizeof(ap[0]));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float z[H];
#pragma unroll
            for (int k = 0; k < H; ++k) {
              const uint32_t w = (i & 2) ? ap[k].y : ap[k].x;
              const uint32_t e = __builtin_amdgcn_ubfe(w, (i & 1) ? 16u : 0u, 16u);   // v_bfe_u32, then one v_lshl_add_u32 (an
              z[k] = lds_abs_f32((e << 2) + bet0);                  // `& 0xFFFF` is folded into shift + mask + add: three)
            }
            const f2w h2 = {cH[i], cH[i]}, g2 = {G[0][i], G[0][i]};
#pragma unroll
            for (int k = 0; k < H / 2; ++k) {
              const f2w z2 = {z[2 * k], z[2 * k + 1]};
              acc2[k] = __builtin_elementwise_fma(__builtin_elementwise_fma(h2, z2, g2), z2, acc2[k]);   // proposal_term, two samples
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        uint2 apA[HA], apB[HBb];
        rows(0, apA);
        for (int m0 = 0; m0 < n_mine; m0 += RW) {
          f2w acc2[RW / 2];
#pragma unroll
          for (int p = 0; p < RW / 2; ++p) acc2[p] = (f2w){0.f, 0.f};
          rows(m0 + HA, apB);
          half(apA, &acc2[0]);
          rows(m0 + RW, apA);                                       // first half of the next round
          half(apB, &acc2[HA / 2]);
          float tot;
          int own;
          if constexpr (RW == 20) {
            rs_f2 a20[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) a20[k] = acc2[k];
            tot = reduce_scatter_20(a20, lane);
            own = rs_p20;
          } else {
            float acc[ACC_ROOM];
#pragma unroll
            for (int k = 0; k < RW / 2; ++k) { acc[2 * k] = acc2[k].x; acc[2 * k + 1] = acc2[k].y; }
            tot = reduce_scatter_n<RW>(acc, lane);
            own = rs_p;
          }
          const int m = m0 + own;                                   // own < 0: unused slot
          if (own >= 0 && (lane & 1) == 0 && m < n_mine) part_s[((size_t)g * SP + m * NSWe + swe) * PS] = tot;
        }
        } else {   // the round-2 form (builds of more than one beam: only their first step comes here)
          constexpr int HB = RW / 2;                                  // rows fetched together (20 registers, as the steady state's)
          static_assert(RW % 2 == 0, "half batches");
          for (int m0 = 0; m0 < n_mine; m0 += RW) {
            float acc[ACC_ROOM];
#pragma unroll
            for (int p = 0; p < RW; ++p) acc[p] = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              uint2 ap[HB];
#pragma unroll
              for (int k = 0; k < HB; ++k) {
                const int m = m0 + h * HB + k;                        // past my last sample: entry 0, the total is dropped
                ap[k] = make_uint2(0u, 0u);
                if (m < n_mine) ap[k] = *reinterpret_cast<const uint2 *>(tab_tu + ((uint32_t)(s_base + m * NSWe + swe) * (uint32_t)Dp + tab_lo));
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float z[HB];
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                  const uint32_t w = (i & 2) ? ap[k].y : ap[k].x;
                  z[k] = lds_abs_f32((((i & 1) ? (w >> 16) : (w & 0xFFFFu)) << 2) + bet0);
                }
#pragma unroll
                for (int k = 0; k < HB; ++k) acc[h * HB + k] = proposal_term(acc[h * HB + k], z[k], cH[i], G[0][i]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            const float tot = reduce_scatter_n<RW>(acc, lane);
            const int m = m0 + rs_p;                                  // rs_p < 0: unused slot
            if (rs_p >= 0 && (lane & 1) == 0 && m < n_mine) part_s[((size_t)g * SP + m * NSWe + swe) * PS] = tot;
          }
        }
      } else if (active && nlive > 0) {
#endif
        const int s_per_stripe = (Sp + NSWe - 1) / NSWe;
        const int nchunks = (s_per_stripe + SPC - 1) / SPC;
        // proposal rows (4 x uint16: dlog(r) + 10006 c of my dims) are fetched one chunk ahead
        uint2 alp_next[SPC];
#pragma unroll
        for (int cc = 0; cc < SPC; ++cc) {
          const int s0 = cc * NSWe + swe;   // sample index inside the pass
          alp_next[cc] = make_uint2(0u, 0u);
          if (s0 < Sp) alp_next[cc] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)(s_base + s0) * Dp);
        }
        for (int ch = 0; ch < nchunks; ++ch) {
          float acc[ACC_ROOM];
#pragma unroll
          for (int p = 0; p < RW; ++p) acc[p] = 0.f;
          uint2 alp[SPC];
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc) {
            alp[cc] = alp_next[cc];
            const int sn = ((ch + 1) * SPC + cc) * NSWe + swe;
            if (sn < Sp) alp_next[cc] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)(s_base + sn) * Dp);
          }
#pragma unroll
          for (int cc = 0; cc < SPC; ++cc) {
            const int s = (ch * SPC + cc) * NSWe + swe;   // sample index inside the pass
            if (s < Sp) { // wave-uniform
              const uint2 ap = alp[cc];
              // byte address of entry alpha' in copy 0 (the table starts at LDS address 0)
              const uint32_t al[4] = {(ap.x & 0xFFFFu) << 2, (ap.x >> 16) << 2, (ap.y & 0xFFFFu) << 2, (ap.y >> 16) << 2};
              if (nlive == NBW) {
                // steady state: all my beams alive -> branch-free; the NBW gathers of one dim are issued back to back
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  float z[NBW];
#pragma unroll
                  for (int b = 0; b < NBW; ++b) z[b] = lds_abs_f32(al[i] + bet[b]); // 4*(dlog r + 10006 c + dlog h): no wrap
#pragma unroll
                  for (int b = 0; b < NBW; ++b) acc[cc * NBW + b] = proposal_term(acc[cc * NBW + b], z[b], cH[i], G[b][i]);
                  __builtin_amdgcn_sched_barrier(0); // one dim's NBW gathers in flight at a time (VGPR budget)
                }
              } else {
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                  if (b < nlive) { // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                      const float z = lds_abs_f32(al[i] + bet[b]);
                      acc[cc * NBW + b] = proposal_term(acc[cc * NBW + b], z, cH[i], G[b][i]);
                    }
                  }
                }
              }
            }
          }
          const float tot = reduce_scatter_n<RW>(acc, lane);
          const int cc = rs_p / NBW, b = rs_p - cc * NBW;   // rs_p < 0: unused slot
          const int s = (ch * SPC + cc) * NSWe + swe;
          if (rs_p >= 0 && (lane & 1) == 0 && s < Sp && b < nlive) part_s[((size_t)g * SP + s) * PS + b_lo + b] = tot;
        }
      }
      TSTAMP(2);
      tsync();
      TSTAMP(3);
      // ---------------- combine dim groups in order, add C_b, build sort keys ----------------
      {
        constexpr int MK = (CMAX + NT - 1) / NT;
        const int Np = Sp * Bcur, f_base = s_base * Bcur; // this pass's candidates: flat indices f_base + [0, Np)
        {
          // every partial of my candidates (and of their beams' C_b) is fetched before the first is used: one LDS
          // latency instead of 4 per key
          float pr[MK][4], cbv[MK][4];
          const int gstride = SP * PS;
#pragma unroll
          for (int q = 0; q < MK; ++q) {
            const int f = q * NT + tid, fs = f < Np ? f : 0;
            const int s = Bcur == NB ? fs / NB : fs / Bcur, b = fs - s * Bcur;   // s: sample index inside the pass
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) pr[q][gg] = part_s[(gg < NG ? gg : 0) * gstride + s * PS + b];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) cbv[q][gg] = cpart_s[(gg < NG ? gg : 0) * TEAM_MB + b];
          }
          if (KEYS_ALIAS && Bcur != NB) { // key f and partial f belong to different candidates: all reads before any write
#pragma unroll
            for (int q = 0; q < MK; ++q)
#pragma unroll
              for (int gg = 0; gg < 4; ++gg) asm volatile("" : "+v"(pr[q][gg]));
            tsync();
          }
          // shared rows: a candidate is MINE when its sample lies in my stripe; the others' keys come out of the exchange
          unsigned long long *xg = nullptr;
          if (CAN_SHARE && Wrow > 1)
            xg = reinterpret_cast<unsigned long long *>(A.coop_xch) + ((size_t)(t & 1) * COOP_MAX_BLOCKS + (size_t)xrow) * COOP_KEYS;
          const unsigned long long tag64 = (unsigned long long)(uint32_t)(t + 1) << 32;
          uint32_t foreign = 0u;                        // bit q: candidate q * NT + tid is a partner's
#pragma unroll
          for (int q = 0; q < MK; ++q) {
            float sc = pr[q][0];
            if (NG > 1) sc = sc + pr[q][1];
            if (NG > 2) sc = sc + pr[q][2];
            if (NG > 3) sc = sc + pr[q][3];
            float cb = cbv[q][0];                       // C_b: its dim-group partials in order, as the scores'
            if (NG > 1) cb = cb + cbv[q][1];
            if (NG > 2) cb = cb + cbv[q][2];
            if (NG > 3) cb = cb + cbv[q][3];
            const int f = q * NT + tid;
            if (f < Np) {
              bool mine = true;
              if (CAN_SHARE && Wrow > 1) {
                const int s_ = Bcur == NB ? f / NB : f / Bcur;
                mine = ((s_ % NSWe) % Wrow) == qsh;
              }
              if (mine) {
                const uint32_t key = score_key(sc + cb);
                key_s[f_base + f] = key;
                if (CAN_SHARE && Wrow > 1) __hip_atomic_store(&xg[f], tag64 | (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else foreign |= 1u << q;
            }
          }
          if (CAN_SHARE && Wrow > 1) {
            // sweep the partners' granules until each carries this step's tag (tags start at 1; the preparation kernel zeroed the row's
            // granules; buffer t & 1 last held tag t - 1).  Give-up as in the split encoder: sticky error flag, COOP_GIVE_UP_TICKS (100 ms).
            const uint32_t tag = (uint32_t)(t + 1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
            int32_t bad = 0;
            for (uint32_t turn = 1; foreign != 0u; ++turn) {
              unsigned long long gq[MK];
#pragma unroll
              for (int q = 0; q < MK; ++q)
                gq[q] = (foreign >> q) & 1u ? __hip_atomic_load(&xg[q * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
              for (int q = 0; q < MK; ++q)
                if (((foreign >> q) & 1u) && (uint32_t)(gq[q] >> 32) == tag) { key_s[f_base + q * NT + tid] = (uint32_t)gq[q]; foreign &= ~(1u << q); }
              if (foreign == 0u || (turn & 63u) != 0u) continue;
              if (__hip_atomic_load(A.coop_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
              if (__builtin_amdgcn_s_memrealtime() - t0 > COOP_GIVE_UP_TICKS) { // the partners are not resident -- give up, loudly
                __hip_atomic_store(A.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1; break;
              }
            }
            if (bad) misc[6] = 1;
          }
        }
      }
      if (s_end < S) tsync(); // the next pass overwrites the partials
      } // sample passes
      if (CAN_SHARE && Wrow > 1) {   // every team of the row sees the (sticky) flag: nobody waits for anybody any more
        tsync();
        if (misc[6]) { abandoned = true; break; }
      }
      const int Bnew = B < N ? B : N;
      TSTAMP(4);
#ifdef IREC_ABLATE_SELECT
      tsync();
      if (tid < Bnew) { sel_s[tid] = tid % S; sel_b[tid] = tid % Bcur; sm->sel_bo[tid] = beta4[cur * TEAM_MB + tid % Bcur];
                        hsum[(cur ^ 1) * TEAM_MB + tid] = 0; beta4[(cur ^ 1) * TEAM_MB + tid] = 0u; }
      tsync();
#else
      // top-B (beam_search_coder.py:85-89); the thread that records new beam j also extends its hash / back-pointer
      // (:94-95) and notes its parent's table offset, so one barrier publishes everything the update needs
      constexpr bool QUICK_SEL = TEAMS <= 2 && !MULTI_PASS;   // (fast_common.h, select_topB_sync: not the 168-VGPR builds)
      if constexpr (QUICK_SEL) __builtin_assume(N <= 1024);   // (host: one pass holds S * NB <= CMAX = 1024 candidates -- the
                                                                //  selection's other paths fold away: -500 cycles per step, scripts/microbench/select_rates.hip)
      select_topB_sync<NT, QUICK_SEL>(key_s, N, Bnew, Bcur, sm, tid, tsync, nullptr, [&](int j, int32_t sp_, int32_t bp_, uint32_t key_) {
        const int32_t nh = (int32_t)((uint32_t)hsum[cur * TEAM_MB + bp_] + (uint32_t)sp_ * (uint32_t)(69 + t));
        hsum[(cur ^ 1) * TEAM_MB + j] = nh;                    // (its discrete log: looked up by every wave in the update, below)
        sm->sel_bo[j] = beta4[cur * TEAM_MB + bp_];
        bp[(size_t)t * NB + j] = (sp_ << 6) | bp_;
        if constexpr (MARGIN) margin_stash(sm, j, key_);
      });
      if constexpr (MARGIN) {   // how close was it?  (wave 0; the barrier keeps the next step's partials off the keys meanwhile)
        if (tid < 64) margin_step(key_s, N, Bnew, Bcur, sm, lane, t == K - 1, macc);
        tsync();
      }
#endif
      TSTAMP(5);
      // ---------------- gather the surviving beams (beam_search_coder.py:92-93), prepare the next step ----------------
      const bool last = (t == K - 1);
      __builtin_amdgcn_s_setprio(2); // serial phase: ahead of the other team's scoring waves
      // new beams' table offsets: lane j looks up dlog(hash(path_j)) -- a global load, consumed at the end of the update
      uint32_t bv_new;
      {
        const int32_t nh = hsum[(cur ^ 1) * TEAM_MB + (lane < Bnew ? lane : 0)];
        bv_new = dlog_s[hash_from_sum(nh) - 1u];
      }
#ifdef IREC_ABLATE_UPDATE
      if (false) {
#else
      if (active) {
#endif
        unpark();
        const float sa_t[4] = {sa[0], sa[1], sa[2], sa[3]};   // this step's sample scale
        const float *bold = beams_g + ((size_t)cur * NB) * FAST_MAX_DIM + d0;
        float *bnew = beams_g + ((size_t)(cur ^ 1) * NB) * FAST_MAX_DIM + d0;
        // G is dead from the end of scoring until it is rebuilt below: every entry is redefined here
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) G[j][i] = 0.f;
        float m[4], cA[4], cBv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = 0.f; cA[i] = 0.f; cBv[i] = 0.f; }
        float cacc[rsn_room(NBW)];
#pragma unroll
        for (int j = 0; j < rsn_room(NBW); ++j) cacc[j] = 0.f;
        constexpr int UB = (TEAMS >= 3 || BS >= 2) ? 5 : NBW;              // beams per load batch: all at once (G's registers are free) unless registers are short
        // the last step keeps ONE beam: beams[0] is all that leaves the block (beam_search_coder.py:118-122), so the parents,
        // rows and look-ups of the other new beams are not fetched at all
        const int Bupd = last ? 1 : Bnew;
        // lane j fetches the selection of new beam j and the offset of its parent: two LDS round trips for all beams
        const int32_t v_sp = sel_s[lane < Bnew ? lane : 0], v_bp = sel_b[lane < Bnew ? lane : 0];
        const uint32_t v_bo = sm->sel_bo[lane < Bnew ? lane : 0];
#pragma unroll
        for (int j0 = 0; j0 < NBW; j0 += UB) {
          // ---- issue the batch's global reads (proposal rows, old beams) back to back ----
          uint2 apv[UB];
          float4 obv4[UB];
          uint32_t bet_old[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int jj = j0 + u, j = b_lo + jj;  // jj: slot in my stripe, j: new beam
            apv[u] = make_uint2(0u, 0u);
            obv4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bet_old[u] = 0u;
            if (jj < NBW && j < Bupd) { // wave-uniform
              const int32_t sp_ = __builtin_amdgcn_readlane(v_sp, j);
              const int32_t bp_ = __builtin_amdgcn_readlane(v_bp, j);
              bet_old[u] = (uint32_t)__builtin_amdgcn_readlane((int)v_bo, j);
              apv[u] = *reinterpret_cast<const uint2 *>(tab_t + (size_t)sp_ * Dp);
#ifndef IREC_ABLATE_SLAB
              if (t) obv4[u] = *reinterpret_cast<const float4 *>(bold + (size_t)bp_ * FAST_MAX_DIM);
#endif
            }
          }
          if (j0 == 0 && !last) { step_consts(t + 1, m, cA, cBv); park(); } // next step's constants, under the loads' latency
          if (j0 == 0) TSTAMP(9);   // (diagnostic build: selection fetch + first load batch issued + step constants)
          // ---- new beams, their G and C terms ----
          // look-ups of YB beams are issued back to back, then consumed: UB / YB LDS latencies per batch instead of UB
          constexpr int YB = 5;
          float zy[YB][4];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            if (u % YB == 0) {
#pragma unroll
              for (int v = 0; v < YB; ++v) {
                const int uu = u + v < UB ? u + v : UB - 1;
                const uint32_t al[4] = {(apv[uu].x & 0xFFFFu) << 2, (apv[uu].x >> 16) << 2, (apv[uu].y & 0xFFFFu) << 2, (apv[uu].y >> 16) << 2};
#pragma unroll
                for (int i = 0; i < 4; ++i) zy[v][i] = lds_abs_f32(al[i] + bet_old[uu]); // entry 0 for beams that do not exist
              }
            }
            const int jj = j0 + u, j = b_lo + jj;
            if (jj < NBW && j < Bupd) { // wave-uniform
              const float obv[4] = {obv4[u].x, obv4[u].y, obv4[u].z, obv4[u].w};
              float nb[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float y = sa_t[i] * zy[u % YB][i];                    // dist.quantile(.), :48-49
                nb[i] = obv[i] + y;                                         // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
              }
              if (last) {
                if (j == 0 && sw == 0 && qsh == 0) {
#pragma unroll
                  for (int i = 0; i < 4; ++i)
                    if (valid[i]) { // beams[0] + coding_dist.loc, :122
                      const int64_t ixo = src_index(A, base, pos, d0 + i);
                      A.out_sample[ixo] = nb[i] + A.p_loc[ixo];
                    }
                }
              } else {
#ifndef IREC_ABLATE_SLAB
                if (sw == 0) *reinterpret_cast<float4 *>(bnew + (size_t)j * FAST_MAX_DIM) = make_float4(nb[0], nb[1], nb[2], nb[3]);
#endif
                if constexpr (LATE_G) {
                  // 168-VGPR builds: the new beam waits in G's own registers; G and the C terms are formed below, once the
                  // batch temporaries (parent beams, proposal rows, look-ups) are dead -- same operations on the same values
#pragma unroll
                  for (int i = 0; i < 4; ++i) G[jj][i] = nb[i];
                } else {
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    G[jj][i] = beam_G(nb[i], m[i], cA[i], cBv[i], sa[i]);
                    cacc[jj] = beam_C_term(cacc[jj], nb[i], m[i], cA[i], cBv[i]);
                  }
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LATE_G) {
          if (!last) {
#pragma unroll
            for (int jj = 0; jj < NBW; ++jj) {
              if (b_lo + jj < Bnew) { // wave-uniform; beams that do not exist keep G = 0
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float nbv = G[jj][i];
                  G[jj][i] = beam_G(nbv, m[i], cA[i], cBv[i], sa[i]);
                  cacc[jj] = beam_C_term(cacc[jj], nbv, m[i], cA[i], cBv[i]);
                }
              }
              __builtin_amdgcn_sched_barrier(0); // one beam at a time (the scheduler would interleave all twenty)
            }
          }
        }
        TSTAMP(10);                 // (diagnostic build: the load batches -- new beams, G, C)
        if (!last) {
          const float ctot = reduce_scatter_n<NBW>(cacc, lane);
          if (sw == 0 && (lane & 1) == 0 && rs_c >= 0 && b_lo + rs_c < Bnew) cpart_s[g * TEAM_MB + b_lo + rs_c] = ctot;
        }
      }
      TSTAMP(6);
      __builtin_amdgcn_s_setprio(0);
      // (wave 0 keeps the LDS copy for the parent look-up of the next selection, which it runs itself: same wave, program order)
      if (tid < Bnew) beta4[(cur ^ 1) * TEAM_MB + tid] = bv_new;
      bv_cur = bv_new;
      // no barrier here: the new C_b partials, hashes and beams are first read behind the next step's barriers
      cur ^= 1;
      Bcur = Bnew;
    }
    // ---- index path of beam 0 (beam_search_coder.py:118-121) ----
    tsync();
    if (abandoned) {                          // a shared row whose partners were not all resident: not coded (the host codes the call again)
      if (tid == 0 && qsh == 0) A.out_K[blk] = -2;
      continue;
    }
    if (tid == 0 && qsh == 0) {
      int j = 0;
      for (int t = K - 1; t >= 0; --t) {
        const int32_t v = __builtin_nontemporal_load(&bp[(size_t)t * NB + j]);
        A.out_indices[blk * (int64_t)A.max_K + t] = v >> 6;
        j = v & 63;
      }
      if constexpr (MARGIN) margin_write(A.out_margin, blk, macc);
    }
    TSTAMP(8);
  }
#ifdef IREC_TEAM_STAMPS
  if (A.dbg && lane == 0)
    for (int k = 0; k < 12; ++k) A.dbg[((size_t)blockIdx.x * (TEAMS * BS * 4) + wave_wg) * 16 + k] = st_acc[k];
  if (A.dbg && lane == 0) {
    A.dbg[((size_t)blockIdx.x * (TEAMS * BS * 4) + wave_wg) * 16 + 12] = __builtin_amdgcn_s_memtime() - st_t0;
    A.dbg[((size_t)blockIdx.x * (TEAMS * BS * 4) + wave_wg) * 16 + 13] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}

#ifndef IREC_TEAM_MARGIN_TU   // (irec_team_margin.hip compiles the team kernel's margin builds only)
// ======================================================================================================
//  encode_chunk_kernel: blocks of MORE than 1024 dims on the team encoder's tables (round 4).
//
//  Coder.__init__ takes any block_size, None included (rec/coding/coder.py:29-36,415-419: the whole latent tensor as ONE
//  block -- the reference's default), and the register-resident encoders above hold 1024 dims per block.  Until now such
//  blocks went to encode_generic_kernel (5-10 k latents/s).  Here a team walks the block in CHUNKS of 1024 dims (four dim
//  groups, one per wave, exactly the lane ownership of the canonical tree) twice per step:
//    scoring   per chunk: the step's constants from the slab's statistics and cumulative variance (the IEEE chain of
//              coder.py:141-154), G and the C_b terms of the live beams from the beams in the slab, then every sample x beam
//              over the chunk -- three table copies, copy bits of choice_table_rows, as in encode_team_kernel -- into the
//              per-group partials; a combine adds the chunk's group sums to the RUNNING score of every candidate in
//              increasing group order, which is the specification's order ("group sums are added in increasing
//              group order"), so the bits are encode_generic_kernel's;
//    update    per chunk: the selected parents and proposal rows -> new beams into the slab's other buffer.
//  A (chunk, dim group) is read and written by ONE wave in both phases: no cross-wave traffic through global memory, the
//  only shared state is the team's LDS (partials, running scores / keys, selection).  Two teams per CU at 256 VGPRs for up to 20
//  beams; one team (beam passes of 10 / 16) for 30 / 32 beam slots, whose partials take the LDS of two.
//  Slab of a team: stats [3][Dpad] | cvar [2][Dpad] (by step parity) | sa [Dpad] | beams [2][NB][Dpad] | bp [max_K][NB].
//
//  Gangs (GANG builds, round 5): a call of FEWER blocks than team slots -- block_size = None on one image's latents: one block of 8192
//  dims would keep one team of one CU busy for 21 ms while 255 CUs idle.  G = A.coop_W teams, each on a CU of its own where the grid allows,
//  code a block together: G = GC chunk owners x SP sample stripes; member m owns the chunks m % GC, + GC, ... (statistics, step constants,
//  update: nothing of a chunk ever leaves its member but its group sums; the SP stripes of a chunk repeat that work, each in its own slab)
//  and scores the sample-chunks m / GC, + SP, ... of them.  Per step: every member writes the group sums of its chunks and samples to the
//  block's exchange in HBM; gang barrier;
//  member m forms the canonical sums -- all group sums of a candidate in increasing group order: the same float32 chain the
//  one-team form adds chunk by chunk -- of the candidates m, m + G, ..., and publishes their sort keys; gang barrier; every member reads
//  all keys and runs the same selection.  The bits are the one-team form's (and the generic kernel's); the K of the block comes the same
//  way from the group sums of its KL.  Members wait for each other: all n_blocks * G teams must be resident (one static hand-out slot
//  each), a member that waits 100 ms for partners that are not poisons the block's counter and the block is reported not coded (-2).
// ======================================================================================================
constexpr int CHUNK_MAX_DIM = 1 << 22;   // (= the bound of irec_beam_encode's max_block_dim; the host caps the scratch slabs of huge blocks)
__host__ __device__ inline size_t chunk_ws_bytes(int NB, int dpad, int max_K) {
  return (size_t)(6 + 2 * NB) * dpad * 4 + ((((size_t)(max_K > 0 ? max_K : 1) * NB * 4) + 255) & ~(size_t)255);
}
__host__ __device__ inline size_t chunk_lds_one(int NB, int NBP, int S) {   // part [4][S][NBP] (of ONE beam pass) | run / keys [S][NB] | TeamLds | barrier
  return team_part_bytes(NBP, S) + team_key_bytes(NB, S) + team_small_bytes(NB) + 16;
}
__host__ __device__ inline size_t chunk_lds_total(int NB, int NBP, int S, int teams) { return T3_BYTES + (size_t)teams * chunk_lds_one(NB, NBP, S); }

// NB beam slots; NBP beams per scoring PASS (the G of NBP beams is what a wave holds in registers: NB = 30 scores a chunk in three passes
// of 10 beams, NB = 32 in two of 16 -- the chunk's step constants are formed once, its rows are re-read per pass); TEAMS per workgroup.
// Round 5: the steady-state scoring is the team encoder's software pipeline (the look-ups of the next dim slot in flight under the
// current slot's fma, accumulators in register pairs, reduce_scatter_20 where 20 values are reduced together); any D; steps beyond the
// proposal tables draw their rows in the kernel (plain scoring form), so no block of a chunked call is left to a second pass;
// the partials are those of ONE pass (combined into the running scores pass by pass), so three teams of 10-beam passes -- 12 waves
// per CU at 168 VGPRs, the register budget G of 10 beams fits without a spill -- find room next to the table copies; up to 60 beam slots
// (passes of 10, two teams: 32 < B <= 60 of blocks beyond 1024 dims no longer falls to the generic kernel).
template <int NB, int NBP, int TEAMS, bool GANG = false>
__global__ __launch_bounds__(TEAMS * TEAM_NT, 1) void encode_chunk_kernel(EncArgs A) {
  using TeamLds = TeamLdsT<NB>;
  constexpr int TEAM_MB = team_mb(NB);
  constexpr size_t TEAM_SMALL_BYTES = (sizeof(TeamLds) + 15) & ~(size_t)15;
  constexpr int NT = TEAM_NT;
  constexpr int SPC = NBP <= 10 ? 20 / NBP : 1;    // samples per reduce-scatter
  constexpr int RW = NBP * SPC;                    // accumulators reduced together
  static_assert(NB % NBP == 0 && (NBP == 10 || NBP == 16 || NBP == 20) && NB <= 60, "chunked encoder: passes of 10, 16 or 20 beams, at most 60 beam slots (6-bit back-pointers)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = A.S, B = A.B;
  const int lane = threadIdx.x & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int team = wave_wg / TEAM_NW, g = wave_wg % TEAM_NW;                    // wave g of a team owns dim group g of every chunk
  const int tid = (int)threadIdx.x - team * NT;
  char *tbase = smem + T3_BYTES + (size_t)team * chunk_lds_one(NB, NBP, S);
  float *part_s = reinterpret_cast<float *>(tbase);                             // [4][S][NBP] of the beam pass being scored
  float *run_s = reinterpret_cast<float *>(tbase + team_part_bytes(NBP, S));    // [S * Bcur] running scores, then the sort keys
  uint32_t *key_s = reinterpret_cast<uint32_t *>(run_s);
  TeamLds *sm = reinterpret_cast<TeamLds *>(tbase + team_part_bytes(NBP, S) + team_key_bytes(NB, S));
  uint32_t *bar_word = reinterpret_cast<uint32_t *>(tbase + team_part_bytes(NBP, S) + team_key_bytes(NB, S) + TEAM_SMALL_BYTES);
  double *gpart = sm->gpart;
  int32_t *sel_s = sm->sel_s, *sel_b = sm->sel_b;
  int32_t *hsum = &sm->hsum[0][0];
  uint32_t *beta4 = &sm->beta4[0][0];
  int32_t *misc = sm->misc;
  float *cpart_s = &sm->cpart[0][0];
  float *Cb_s = sm->Cb;
  const uint16_t *dlog_s = A.dlog4r;
  const int rs_p = rsn_owner<RW>(lane), rs_c = rsn_owner<NBP>(lane);
  const int rs_p20 = RW == 20 ? rs20_owner(lane) : -1;
  double *kl_tot = reinterpret_cast<double *>(sm->wb);                          // running KL total of the prologue (wb is idle then)

  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  commit_table_stamps(A);
  {
    float *l3 = reinterpret_cast<float *>(smem);
    for (int k = (int)threadIdx.x; k < (int)IREC_PM1; k += TEAMS * NT) {
      const float v = A.lut2[k];
      l3[k] = v; l3[k + IREC_PM1] = v; l3[k + 2 * IREC_PM1] = v;
    }
    if (tid == 0) *bar_word = 0u;
  }
  __syncthreads();
  TeamBarrier tsync{bar_word, 0u, (uint32_t)TEAM_NW};

  const int Dpad = A.max_dim_pad;
  char *slab = A.ws + ((size_t)blockIdx.x * TEAMS + team) * A.ws_per_wg;
  float *stats_g = reinterpret_cast<float *>(slab);                 // [3][Dpad]: mq - mp, sq^2, sp^2
  float *cvar_g = stats_g + (size_t)3 * Dpad;                       // [2][Dpad]: cumulative variance, by step parity
  float *sa_g = cvar_g + (size_t)2 * Dpad;                          // [Dpad]: this step's sample scale
  float *beams_g = sa_g + Dpad;                                     // [2][NB][Dpad]
  int32_t *bp = reinterpret_cast<int32_t *>(beams_g + (size_t)2 * NB * Dpad);   // [max_K][NB]

  const int64_t n_static = (int64_t)TEAMS * (int64_t)gridDim.x < A.n_blocks ? (int64_t)TEAMS * (int64_t)gridDim.x : A.n_blocks;
  bool first_block = true;
  int steal = 0;
  // GANG: the G = A.coop_W teams in hand-out slots [blk * G, blk * G + G) code block blk together -- member gm owns the chunks gm, gm + G, ...
  // of it (see "gangs" above the kernel); a team takes its one slot of the static round and leaves
  // The G members are GC chunk owners x SP sample stripes: member gm owns the chunks gc = gm % GC, gc + GC, ... and scores the samples of
  // sample-chunk sp = gm / GC, sp + SP, ... of them (statistics, step constants, G and the update of a chunk are repeated by its SP stripes,
  // each in its own slab; the group sums of a candidate still come from ONE member each).
  constexpr int ABL = GANG ? IREC_GANG_ABLATE : 0;
  const int G = GANG ? A.coop_W : 1;
  const int GC = GANG ? A.gang_chunks : 1, SP = GANG ? G / GC : 1;
  uint32_t gang_epoch = 0u;
  for (;;) {
    tsync();
    if (tid == 0) {
      int64_t r;
      if constexpr (GANG) {
        const int64_t slot = (int64_t)team * (int64_t)gridDim.x + (int64_t)blockIdx.x;
        r = first_block && slot < A.n_blocks * (int64_t)G ? slot / G : A.n_blocks;
        misc[2] = (int32_t)(slot % G);
      } else if (first_block) {
        r = (int64_t)team * (int64_t)gridDim.x + (int64_t)blockIdx.x;
        r = r < n_static ? xcd_static_row(r, n_static, (int)gridDim.x) : A.n_blocks;
      } else r = xcd_pull_row(A, n_static, A.n_blocks, steal);
      misc[0] = (int32_t)r;
    }
    first_block = false;
    tsync();
    const int64_t blk = misc[0];
    if (blk >= A.n_blocks) break; // every wave of the team reaches this
    const int gm = GANG ? misc[2] : 0;
    const int gc = gm % GC, sp = gm / GC;
    if (GANG && A.coop_test_orphan && gm != 0) break;   // IREC_FLAG_TEST_SPLIT_ORPHAN: member 0 waits alone, gives up, reports -2
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const uint16_t *tab = nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (A.tab_dim[q] == D) tab = A.tab[q];
    if (D < 1 || D > Dpad || tab == nullptr) { // host promised D <= max_block_dim and listed dims
      if (tid == 0) A.out_K[blk] = -1;
      continue;
    }
    const int Dp = (D + 3) & ~3;            // row stride of the proposal table
    const int NC = (D + 1023) >> 10;        // chunks of 1024 dims
    auto groups_of = [&](int c) { const int left = D - (c << 10); return left >= 1024 ? 4 : (left + 255) >> 8; };
    // ---- gang exchange of this block (GANG only; layout: gang_xch_bytes, irec_kernels.h) ----
    const int NGt = (D + 255) >> 8;                                // dim groups of the block: the terms of every canonical sum, in order
    const int NGm = 4 * ((Dpad + 1023) >> 10);                     // row stride of the exchange (groups of the call's largest block)
    const int NCAND = S * NB;
    float *gx_part = nullptr, *gx_cpart = nullptr;
    unsigned long long *gx_kl = nullptr;
    uint32_t *gx_keys = nullptr;
    unsigned int *gang_ctr = nullptr;
    if constexpr (GANG) {
      char *xb = A.gang_xch + (size_t)blk * A.gang_stride;
      gx_part = reinterpret_cast<float *>(xb);                     // [S * NB][NGm]: group sums of every candidate, candidate-major
      gx_cpart = gx_part + (size_t)NCAND * NGm;                    // [NB][NGm]: group sums of the C_b terms
      gx_kl = reinterpret_cast<unsigned long long *>(gx_cpart + (size_t)NB * NGm);   // [NGm] doubles: group sums of the KL
      gx_keys = reinterpret_cast<uint32_t *>(gx_kl + NGm);         // [S * NB] sort keys of the step
      gang_ctr = reinterpret_cast<unsigned int *>(A.coop_xch) + (size_t)blk * (COOP_KEYS * 2);   // first word of the block's exchange granules in the
                                                                                             // workspace head: zeroed by the call's preparation kernel
    }
    // Barrier of the gang: a monotonic arrival counter in HBM.  Everything handed over travels as agent-scope (sc1) stores that have
    // drained before the arrival (s_waitcnt vmcnt(0) in every wave, then the team barrier, then one arrival) and is read back by agent-scope loads.  A member
    // that has waited COOP_GIVE_UP_TICKS for partners that are not resident POISONS the counter (bit 31, by compare-and-swap against an
    // incomplete count, so that either every member passes a barrier or none does) and the block is reported as not coded (out_K = -2).
    auto gsync = [&]() -> bool {
      if constexpr ((ABL & 32) != 0) { tsync(); return true; }
      gang_epoch += (uint32_t)G;
      // every wave drains its stores before the team barrier: the release fence in there is workgroup-scoped and need not wait for
      // vector-memory stores to be acknowledged (the waves of a workgroup share their L1), but the partners of the gang sit on other CUs
      // and must find the sums in place once the arrival below is visible.
      // Why this is enough on gfx950, and why it is asm and not the memory model (round 6, scripts/microbench/litmus.hip, profiles/r06r/):
      //   * everything handed over is written by agent-scope stores (global_store .. sc1: written through to the memory side that all XCDs
      //     share) and read by agent-scope loads (global_load .. sc1: not served from a stale L1 / L2 line);
      //   * vmcnt counts a store down when the memory side has ACKNOWLEDGED it, so after s_waitcnt vmcnt(0) the wave's data is where every
      //     agent-scope load finds it; the team barrier then orders the four waves' drains before thread 0's arrival (LDS counter);
      //   * the arrival itself is a relaxed agent-scope RMW on one word: whoever sees it, sees it after the acknowledgements.
      //   The memory model says the same with an agent-scope release fence in every wave and an acquire fence behind the wait; that form
      //   passes the litmus too (form 2m) and costs 3 x the hand-off (47 against 16 us for 16 KB under light load, 76 against 58 under
      //   heavy): buffer_wbl2 + buffer_inv sc1 write back and invalidate the whole L2 for data that never was in it.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tsync();
      if (tid == 0) {
        int32_t bad = 0;
        __hip_atomic_fetch_add(gang_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (uint32_t turn = 1;; ++turn) {
          const uint32_t v = __hip_atomic_load(gang_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >> 31) { bad = 1; break; }
          if ((int32_t)(v - gang_epoch) >= 0) break;
          __builtin_amdgcn_s_sleep(2);
          if ((turn & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > COOP_GIVE_UP_TICKS) {
            uint32_t expect = v;
            if (__hip_atomic_compare_exchange_strong(gang_ctr, &expect, v | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(A.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bad = 1; break;
            }
          }
        }
        misc[6] = bad;
      }
      tsync();
      return misc[6] == 0;
    };
    auto ld_f32 = [](const float *p_) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    auto st_f32x2 = [](float *p_, float a_, float b_) {   // (8-byte aligned)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(p_), (unsigned long long)__float_as_uint(a_) | ((unsigned long long)__float_as_uint(b_) << 32),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Canonical sums of `nrows` exchange rows (NGt terms each, increasing group order), staged through the partial buffer: all threads
    // fetch a range of groups of up to NT rows, one thread per row adds the range onto its running value in order.
    // src(i): the row's base; sink(i, v): what to do with its sum.
    auto gang_reduce = [&](int nrows, auto src, auto sink) {
      float *stage = part_s;
      const int CAP = 4 * S * NBP;
      const int TI = NT < CAP ? NT : CAP;
      for (int r0 = 0; r0 < nrows; r0 += TI) {
        const int nr = nrows - r0 < TI ? nrows - r0 : TI;
        int GR = CAP / nr;
        if (GR > NGt) GR = NGt;
        float v = 0.f;
        for (int g0 = 0; g0 < NGt; g0 += GR) {
          const int ng = NGt - g0 < GR ? NGt - g0 : GR;
          for (int e = tid; e < nr * ng; e += NT) { const int i = e / ng, gi = e - i * ng; stage[e] = ld_f32(src(r0 + i) + g0 + gi); }
          tsync();
          if (tid < nr) {
            const float *p_ = stage + tid * ng;
            int gi = 0;
            if (g0 == 0) { v = p_[0]; gi = 1; }
            for (; gi < ng; ++gi) v = v + p_[gi];
          }
          tsync();
        }
        if (tid < nr) sink(r0 + tid, v);
      }
      tsync();
    };
    bool gang_lost = false;

    // ---- statistics (split == gather through perm) and the block's KL, groups in increasing order ----
    for (int c = gc; c < NC; c += GC) {
      const int ngc = groups_of(c);
      const int d0 = (c << 10) + g * 256 + lane * 4;
      double klacc = 0.0;
      if (g < ngc) {
        float st3[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          st3[0][i] = 0.f; st3[1][i] = 1.f; st3[2][i] = 1.f;
          if (d0 + i < D) {
            const int64_t ixi = src_index(A, base, pos, d0 + i);
            const float mq_ = A.q_loc[ixi], sq_ = A.q_scale[ixi], mp_ = A.p_loc[ixi], sp_ = A.p_scale[ixi];
            klacc = klacc + kl_dim(mq_, sq_, mp_, sp_);
            st3[0][i] = mq_ - mp_; st3[1][i] = sq_ * sq_; st3[2][i] = sp_ * sp_;
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
          *reinterpret_cast<float4 *>(stats_g + (size_t)k * Dpad + d0) = make_float4(st3[k][0], st3[k][1], st3[k][2], st3[k][3]);
        *reinterpret_cast<float4 *>(cvar_g + d0) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const double gs = wave_tree_sum(klacc);
      if constexpr (GANG) {
        if (g < ngc && lane == 0 && sp == 0) __hip_atomic_store(gx_kl + c * 4 + g, (unsigned long long)__double_as_longlong(gs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      if (g < ngc && lane == 0) gpart[g] = gs;
      tsync();
      if (tid == 0) {
        double tot = c == 0 ? gpart[0] : *kl_tot + gpart[0];
        for (int gg = 1; gg < ngc; ++gg) tot = tot + gpart[gg];
        *kl_tot = tot;
      }
      tsync();
    }
    if constexpr (GANG) {   // the group sums of every member, added in group order by every member
      if (!gsync()) { if (tid == 0) A.out_K[blk] = -2; continue; }
      double *stage = reinterpret_cast<double *>(part_s);
      const int CAPD = 2 * S * NBP;
      double tot = 0.0;
      for (int g0 = 0; g0 < NGt; g0 += CAPD) {
        const int ng = NGt - g0 < CAPD ? NGt - g0 : CAPD;
        for (int e = tid; e < ng; e += NT) stage[e] = __longlong_as_double((long long)__hip_atomic_load(gx_kl + g0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        tsync();
        if (tid == 0) {
          int gi = 0;
          if (g0 == 0) { tot = stage[0]; gi = 1; }
          for (; gi < ng; ++gi) tot = tot + stage[gi];
        }
        tsync();
      }
      if (tid == 0) *kl_tot = tot;
    }
    if (tid == 0) {
      const int32_t K = num_aux((float)*kl_tot, A.omega);
      misc[1] = K;
      if (gm == 0) A.out_K[blk] = K;
      hsum[0] = 0;
      beta4[0] = 0u; // hash of the empty path is 1 = g^0
    }
    tsync();
    const int K = misc[1];
    if (K > A.max_K || K > A.K_limit) continue;
    // (steps beyond the table window -- K grows with the dims: 2 200 partitions for a 301 056-dim block -- draw their rows in the kernel, below)
    if (K == 0) { // nothing to code: sample = p.loc
      for (int c = gc; sp == 0 && c < NC; c += GC) {
        const int d0 = (c << 10) + g * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (g < groups_of(c) && d0 + i < D) { const int64_t ixo = src_index(A, base, pos, d0 + i); A.out_sample[ixo] = 0.f + A.p_loc[ixo]; }
      }
      continue;
    }

    int cur = 0, Bcur = 1;
    uint32_t bv_cur = 0u;                                            // lane j: 4 * dlog(hash(path of beam j))
    for (int t = 0; t < K; ++t) {
      const bool fused = t >= A.K_tab;                               // beyond the proposal tables: the rows are drawn here
      const uint16_t *tab_tu = tab + (size_t)(fused ? 0 : t) * S * Dp; // this step's rows (fused: never read)
      const StepSeed ss = make_step_seed(A.seed + t);
      // Row of sample s_ for the quad at dim q0 (a multiple of 4) of a step beyond the tables: the int32 draw of get_pseudo_random_sample
      // itself (beam_search_coder.py:38-43) mapped to discrete logs, copy bit 0 -- the table's format, random banks (8.9 instead of
      // 13.7 look-ups/clk/CU, and a Philox block per quad and sample on the VALU: the regime of blocks no window can hold).
      auto fused_row = [&](int s_, uint32_t q0) {
        uint32_t rm1[4];
        draw_rm1_x4(ss, (uint64_t)s_ * (uint64_t)D + (uint64_t)q0, rm1);   // ((s_ * D + q0) & 3 is wave-uniform: q0 % 4 == 0)
        uint32_t a_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a_[i] = (uint32_t)dlog_s[rm1[i]] >> 2;
        return make_uint2(a_[0] | (a_[1] << 16), a_[2] | (a_[3] << 16));
      };
      const float rho = A.rho[K - 1 - t];
      const int N = S * Bcur;
      // ---------------- scoring, chunk by chunk (beam_search_coder.py:67-84) ----------------
      for (int c = gc; c < NC; c += GC) {
        const int ngc = groups_of(c);
        const int d0 = (c << 10) + g * 256 + lane * 4;
        const bool mine = g < ngc;                                    // (wave-uniform) this dim group exists in the chunk
        // the step's constants of my four dims (coder.py:141-154), from the statistics and the cumulative variance in the slab
        float sa[4] = {0.f, 0.f, 0.f, 0.f}, cH[4] = {0.f, 0.f, 0.f, 0.f}, m[4] = {0.f, 0.f, 0.f, 0.f}, cA[4] = {0.f, 0.f, 0.f, 0.f}, cBv[4] = {0.f, 0.f, 0.f, 0.f};
        if (mine) {
          float cn[4];
          {
            const float4 q0 = *reinterpret_cast<const float4 *>(stats_g + d0);
            const float4 q1 = *reinterpret_cast<const float4 *>(stats_g + (size_t)Dpad + d0);
            const float4 q2 = *reinterpret_cast<const float4 *>(stats_g + (size_t)2 * Dpad + d0);
            const float4 qc = *reinterpret_cast<const float4 *>(cvar_g + (size_t)(t & 1) * Dpad + d0);
            const float dmu_[4] = {q0.x, q0.y, q0.z, q0.w}, vq_[4] = {q1.x, q1.y, q1.z, q1.w}, vp_[4] = {q2.x, q2.y, q2.z, q2.w};
            const float c_[4] = {qc.x, qc.y, qc.z, qc.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool ok = d0 + i < D;
              const StepConst sc = step_constants(rho, dmu_[i], vq_[i], vp_[i], c_[i]);
              sa[i] = ok ? sc.sa : 0.f; cH[i] = ok ? sc.H : 0.f;
              m[i] = ok ? sc.m : 0.f; cA[i] = ok ? sc.A : 0.f; cBv[i] = ok ? sc.Bv : 0.f;
              cn[i] = c_[i] + sc.a;                                    // cumulative_auxiliary_variance += auxiliary_var (:109)
              __builtin_amdgcn_sched_barrier(0);
            }
            *reinterpret_cast<float4 *>(cvar_g + (size_t)((t + 1) & 1) * Dpad + d0) = make_float4(cn[0], cn[1], cn[2], cn[3]);
            *reinterpret_cast<float4 *>(sa_g + d0) = make_float4(sa[0], sa[1], sa[2], sa[3]);
          }
        }
        const uint32_t tab_lo = (uint32_t)(d0 < Dp ? d0 : Dp - 4);   // lanes past the row's end: its last quad (zero coefficients)
        const uint16_t *tab_t = tab_tu + tab_lo;
#pragma unroll 1
        for (int bp0 = 0; bp0 < NB; bp0 += NBP) {                    // beam passes (one for NB = NBP)
          const int nlive = Bcur - bp0 < NBP ? Bcur - bp0 : NBP;     // live beams of this pass
          if (nlive <= 0) break;                                     // (uniform over the team)
          if (mine) {
            // G and the C_b terms of the pass's live beams (dead slots: G = 0, never read)
            float G[NBP][4];
            {
              float cacc[rsn_room(NBP)];
#pragma unroll
              for (int b = 0; b < rsn_room(NBP); ++b) cacc[b] = 0.f;
              const float *bold = beams_g + ((size_t)cur * NB + bp0) * Dpad + d0;
              float4 bq[NBP];
#pragma unroll
              for (int b = 0; b < NBP; ++b) {
                bq[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t && b < nlive) bq[b] = *reinterpret_cast<const float4 *>(bold + (size_t)b * Dpad);   // (wave-uniform)
              }
#pragma unroll
              for (int b = 0; b < NBP; ++b) {
                const float bv4[4] = {bq[b].x, bq[b].y, bq[b].z, bq[b].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  G[b][i] = b < nlive ? beam_G(bv4[i], m[i], cA[i], cBv[i], sa[i]) : 0.f;
                  if (b < nlive) cacc[b] = beam_C_term(cacc[b], bv4[i], m[i], cA[i], cBv[i]);
                }
              }
              const float ctot = reduce_scatter_n<NBP>(cacc, lane);
              if ((lane & 1) == 0 && rs_c >= 0 && rs_c < nlive) cpart_s[g * TEAM_MB + bp0 + rs_c] = ctot;
            }
            uint32_t bet[NBP];
#pragma unroll
            for (int b = 0; b < NBP; ++b) bet[b] = (uint32_t)__builtin_amdgcn_readlane((int)bv_cur, bp0 + b < Bcur ? bp0 + b : 0);
            if (nlive == NBP && Bcur > 1 && !fused) {
              // ---- steady state, software pipelined by dim slot (as encode_team_kernel's scoring loop): the NBP look-ups of the NEXT slot are
              // issued before the current slot's values are consumed; values in register pairs (v_pk_fma_f32); rows two sample-chunks ahead
              typedef float f2 __attribute__((ext_vector_type(2)));
              constexpr int NP = NBP / 2, NQ = 4 * SPC;
              const int n_sch = (S + SPC - 1) / SPC;
              auto row = [&](int s_) {
                uint2 r = make_uint2(0u, 0u);
                if (s_ < S) r = *reinterpret_cast<const uint2 *>(tab_tu + ((uint32_t)s_ * (uint32_t)Dp + tab_lo));
                return r;
              };
              uint2 ap_cur[SPC], ap_nxt[SPC];
#pragma unroll
              for (int cc = 0; cc < SPC; ++cc) { ap_cur[cc] = row(sp * SPC + cc); ap_nxt[cc] = row((sp + SP) * SPC + cc); }
#define CHUNK_AL(CC, I) ((((I) & 2) ? (((I) & 1) ? (ap_cur[CC].y >> 16) : (ap_cur[CC].y & 0xFFFFu)) : (((I) & 1) ? (ap_cur[CC].x >> 16) : (ap_cur[CC].x & 0xFFFFu))) << 2)
#define CHUNK_ISSUE(Z, AD) do { _Pragma("unroll") for (int k = 0; k < NP; ++k) { Z[k].x = lds_abs_f32((AD) + bet[2 * k]); Z[k].y = lds_abs_f32((AD) + bet[2 * k + 1]); } \
                                __builtin_amdgcn_sched_barrier(0); } while (0)
#define CHUNK_CONSUME(Z, I, ACC) do { _Pragma("unroll") for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(Z[k])); \
                                f2 t2_[NP]; \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) { \
                                  const f2 h2 = {cH[I], cH[I]}, g2 = {G[2 * k][I], G[2 * k + 1][I]}; \
                                  t2_[k] = __builtin_elementwise_fma(h2, Z[k], g2); } \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) ACC[k] = __builtin_elementwise_fma(t2_[k], Z[k], ACC[k]); \
                                _Pragma("unroll") for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(ACC[k])); \
                                __builtin_amdgcn_sched_barrier(0); } while (0)
              f2 zz[2][NP];
              CHUNK_ISSUE(zz[0], CHUNK_AL(0, 0));
              for (int ch = sp; ch < ((ABL & 1) ? 0 : n_sch); ch += SP) {          // (my stripe of the sample-chunks; SP = 1 but in gangs)
                f2 acc2[SPC][NP];
#pragma unroll
                for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
                  for (int k = 0; k < NP; ++k) acc2[cc][k] = (f2){0.f, 0.f};
                uint2 ap_new[SPC];
#pragma unroll
                for (int cc = 0; cc < SPC; ++cc) ap_new[cc] = row((ch + 2 * SP) * SPC + cc);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                  if (q + 1 < NQ) CHUNK_ISSUE(zz[(q + 1) & 1], CHUNK_AL((q + 1) >> 2, (q + 1) & 3));
                  else {
#pragma unroll
                    for (int cc = 0; cc < SPC; ++cc) { ap_cur[cc] = ap_nxt[cc]; ap_nxt[cc] = ap_new[cc]; }
                    CHUNK_ISSUE(zz[0], CHUNK_AL(0, 0));
                  }
                  CHUNK_CONSUME(zz[q & 1], q & 3, acc2[q >> 2]);
                }
                float tot;
                int own;
                if constexpr (RW == 20) {
                  rs_f2 a20[10];
#pragma unroll
                  for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
                    for (int k = 0; k < NP; ++k) a20[cc * NP + k] = acc2[cc][k];
                  tot = reduce_scatter_20(a20, lane);
                  own = rs_p20;
                } else {
                  float acc[rsn_room(RW)];
#pragma unroll
                  for (int cc = 0; cc < SPC; ++cc)
#pragma unroll
                    for (int k = 0; k < NP; ++k) { acc[cc * NBP + 2 * k] = acc2[cc][k].x; acc[cc * NBP + 2 * k + 1] = acc2[cc][k].y; }
                  tot = reduce_scatter_n<RW>(acc, lane);
                  own = rs_p;
                }
                const int cc = own / NBP, b = own - cc * NBP;          // own < 0: unused slot
                const int s_ = ch * SPC + cc;
                if (own >= 0 && (lane & 1) == 0 && s_ < S) part_s[((size_t)g * S + s_) * NBP + b] = tot;
              }
#pragma unroll
              for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(zz[0][k])); // drain the look-ups issued past the last sample
#undef CHUNK_AL
#undef CHUNK_ISSUE
#undef CHUNK_CONSUME
            } else {
              // ---- the first step (one beam) and passes that are not full: a dim slot's look-ups issued together, then consumed
              const int nchunks = (S + SPC - 1) / SPC;
              uint2 alp_next[SPC];
#pragma unroll
              for (int cc = 0; cc < SPC; ++cc) {
                alp_next[cc] = make_uint2(0u, 0u);
                const int s0 = sp * SPC + cc;
                if (s0 < S) alp_next[cc] = fused ? fused_row(s0, tab_lo) : *reinterpret_cast<const uint2 *>(tab_t + (size_t)s0 * Dp);
              }
              for (int ch = sp; ch < ((ABL & 1) ? 0 : nchunks); ch += SP) {
                float acc[rsn_room(RW)];
#pragma unroll
                for (int p_ = 0; p_ < rsn_room(RW); ++p_) acc[p_] = 0.f;
                uint2 alp[SPC];
#pragma unroll
                for (int cc = 0; cc < SPC; ++cc) {
                  alp[cc] = alp_next[cc];
                  const int sn = (ch + SP) * SPC + cc;
                  if (sn < S) alp_next[cc] = fused ? fused_row(sn, tab_lo) : *reinterpret_cast<const uint2 *>(tab_t + (size_t)sn * Dp);
                }
#pragma unroll
                for (int cc = 0; cc < SPC; ++cc) {
                  const int s_ = ch * SPC + cc;
                  if (s_ < S) { // wave-uniform
                    const uint2 ap = alp[cc];
                    const uint32_t al[4] = {(ap.x & 0xFFFFu) << 2, (ap.x >> 16) << 2, (ap.y & 0xFFFFu) << 2, (ap.y >> 16) << 2};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                      float z[NBP];
#pragma unroll
                      for (int b = 0; b < NBP; ++b) z[b] = lds_abs_f32(al[i] + bet[b]);   // 4 * (dlog r + 10006 c + dlog h): no wrap
#pragma unroll
                      for (int b = 0; b < NBP; ++b) acc[cc * NBP + b] = proposal_term(acc[cc * NBP + b], z[b], cH[i], G[b][i]);
                      __builtin_amdgcn_sched_barrier(0);
                    }
                  }
                }
                const float tot = reduce_scatter_n<RW>(acc, lane);
                const int cc = rs_p / NBP, b = rs_p - cc * NBP;         // rs_p < 0: unused slot
                const int s_ = ch * SPC + cc;
                if (rs_p >= 0 && (lane & 1) == 0 && s_ < S && b < nlive) part_s[((size_t)g * S + s_) * NBP + b] = tot;
              }
            }
          }
          tsync();
          if constexpr (GANG) {   // the pass's group sums of this chunk: to the gang's exchange, four groups of a candidate side by side
            for (int f = tid; f < S * nlive; f += NT) {
              const int s_ = f / nlive, bl = f - s_ * nlive;
              if ((s_ / SPC) % SP != sp) continue;                     // (another stripe's sample)
              float *row = gx_part + (size_t)(s_ * Bcur + bp0 + bl) * NGm + c * 4;
              float v4[4];
#pragma unroll
              for (int gg = 0; gg < 4; ++gg) v4[gg] = gg < ngc ? part_s[((size_t)gg * S + s_) * NBP + bl] : 0.f;
              st_f32x2(row, v4[0], v4[1]); st_f32x2(row + 2, v4[2], v4[3]);
            }
            if (tid < nlive && sp == 0) {
              const int b = bp0 + tid;
              float *row = gx_cpart + (size_t)b * NGm + c * 4;
              float v4[4];
#pragma unroll
              for (int gg = 0; gg < 4; ++gg) v4[gg] = gg < ngc ? cpart_s[gg * TEAM_MB + b] : 0.f;
              st_f32x2(row, v4[0], v4[1]); st_f32x2(row + 2, v4[2], v4[3]);
            }
            tsync();   // partials free for the next pass / chunk
            continue;
          }
          // the pass's group sums of this chunk onto the running scores / C_b of its beams, increasing group order
          for (int f = tid; f < S * nlive; f += NT) {
            const int s_ = f / nlive, bl = f - s_ * nlive;
            const int fr = s_ * Bcur + bp0 + bl;                       // flat candidate index of (sample, beam)
            float v = part_s[((size_t)0 * S + s_) * NBP + bl];
            if (c > 0) v = run_s[fr] + v;
            for (int gg = 1; gg < ngc; ++gg) v = v + part_s[((size_t)gg * S + s_) * NBP + bl];
            run_s[fr] = v;
          }
          if (tid < nlive) {
            const int b = bp0 + tid;
            float cb = cpart_s[b];
            if (c > 0) cb = Cb_s[b] + cb;
            for (int gg = 1; gg < ngc; ++gg) cb = cb + cpart_s[gg * TEAM_MB + b];
            Cb_s[b] = cb;
          }
          tsync();   // partials free for the next pass / chunk; running sums and C_b published
        }
      }
      if constexpr (GANG) {
        // every group sum of the step is out: C_b of every beam by every member, the scores of the candidates gm, gm + G, ... by member gm
        // (the terms of each in increasing group order: the canonical sums), their sort keys to the exchange, all keys back
        if (!gsync()) { gang_lost = true; break; }
        const int n_mine = gm < N ? (N - gm + G - 1) / G : 0;
        const bool cb_all = n_mine >= Bcur;                       // (else: only the C_b of my candidates' beams)
        const int n_cb = cb_all ? Bcur : n_mine;
        auto cb_of = [&](int i) { return cb_all ? i : (gm + i * G) % Bcur; };
        if constexpr ((ABL & 4) == 0)
        gang_reduce(n_cb + n_mine,
                    [&](int i) { return i < n_cb ? gx_cpart + (size_t)cb_of(i) * NGm : gx_part + (size_t)(gm + (i - n_cb) * G) * NGm; },
                    [&](int i, float v) { if (i < n_cb) Cb_s[cb_of(i)] = v; else run_s[i - n_cb] = v; });
        for (int i = tid; i < n_mine; i += NT) {
          const int f = gm + i * G;
          __hip_atomic_store(gx_keys + f, score_key(run_s[i] + Cb_s[f % Bcur]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!gsync()) { gang_lost = true; break; }
        for (int f = tid; f < N; f += NT) key_s[f] = __hip_atomic_load(gx_keys + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int f = tid; f < N; f += NT) {
          const int s_ = f / Bcur, b = f - s_ * Bcur;
          key_s[f] = score_key(run_s[f] + Cb_s[b]);
        }
      }
      const int Bnew = B < N ? B : N;
      // top-B (beam_search_coder.py:85-89); the selection's first barrier orders the key writes
      if constexpr ((ABL & 8) != 0) {
        tsync();
        if (tid < Bnew) { sel_s[tid] = tid % S; sel_b[tid] = tid % Bcur; sm->sel_bo[tid] = beta4[cur * TEAM_MB + tid % Bcur];
                          hsum[(cur ^ 1) * TEAM_MB + tid] = 0; bp[(size_t)t * NB + tid] = ((tid % S) << 6) | (tid % Bcur); }
        tsync();
      } else
      select_topB_sync<NT>(key_s, N, Bnew, Bcur, sm, tid, tsync, nullptr, [&](int j, int32_t sp_, int32_t bp_, uint32_t) {
        const int32_t nh = (int32_t)((uint32_t)hsum[cur * TEAM_MB + bp_] + (uint32_t)sp_ * (uint32_t)(69 + t));
        hsum[(cur ^ 1) * TEAM_MB + j] = nh;
        sm->sel_bo[j] = beta4[cur * TEAM_MB + bp_];
        bp[(size_t)t * NB + j] = (sp_ << 6) | bp_;
      });
      // ---------------- new beams, chunk by chunk (beam_search_coder.py:92-93) ----------------
      const bool last = (t == K - 1);
      uint32_t bv_new;
      {
        const int32_t nh = hsum[(cur ^ 1) * TEAM_MB + (lane < Bnew ? lane : 0)];
        bv_new = dlog_s[hash_from_sum(nh) - 1u];
      }
      const int Bupd = last ? 1 : ((ABL & 2) ? 0 : Bnew);     // beams[0] is all that leaves the block (:118-122)
      const int32_t v_sp = sel_s[lane < Bnew ? lane : 0], v_bp = sel_b[lane < Bnew ? lane : 0];
      const uint32_t v_bo = sm->sel_bo[lane < Bnew ? lane : 0];
      for (int c = gc; c < NC; c += GC) {
        const int d0 = (c << 10) + g * 256 + lane * 4;
        if (g >= groups_of(c)) continue;     // wave-uniform
        const uint32_t tab_lo = (uint32_t)(d0 < Dp ? d0 : Dp - 4);
        const uint16_t *tab_t = tab_tu + tab_lo;
        const float4 sq = *reinterpret_cast<const float4 *>(sa_g + d0);
        const float sa_t[4] = {sq.x, sq.y, sq.z, sq.w};
        const float *bold = beams_g + (size_t)cur * NB * Dpad + d0;
        float *bnew = beams_g + (size_t)(cur ^ 1) * NB * Dpad + d0;
        constexpr int UB = 5;                // beams per load batch
#pragma unroll 1
        for (int j0 = 0; j0 < Bupd; j0 += UB) {
          uint2 apv[UB];
          float4 obv4[UB];
          uint32_t bet_old[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int j = j0 + u;
            apv[u] = make_uint2(0u, 0u); obv4[u] = make_float4(0.f, 0.f, 0.f, 0.f); bet_old[u] = 0u;
            if (j < Bupd) { // wave-uniform
              const int32_t sp_ = __builtin_amdgcn_readlane(v_sp, j);
              const int32_t bp_ = __builtin_amdgcn_readlane(v_bp, j);
              bet_old[u] = (uint32_t)__builtin_amdgcn_readlane((int)v_bo, j);
              apv[u] = fused ? fused_row(sp_, tab_lo) : *reinterpret_cast<const uint2 *>(tab_t + (size_t)sp_ * Dp);
              if (t) obv4[u] = *reinterpret_cast<const float4 *>(bold + (size_t)bp_ * Dpad);
            }
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int j = j0 + u;
            if (j < Bupd) { // wave-uniform
              const uint32_t al[4] = {(apv[u].x & 0xFFFFu) << 2, (apv[u].x >> 16) << 2, (apv[u].y & 0xFFFFu) << 2, (apv[u].y >> 16) << 2};
              const float obv[4] = {obv4[u].x, obv4[u].y, obv4[u].z, obv4[u].w};
              float nb[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float y = sa_t[i] * lds_abs_f32(al[i] + bet_old[u]);   // dist.quantile(.), :48-49
                nb[i] = obv[i] + y;                                          // combined_samples[best_ind_aux, best_ind_beam], :81,92-93
              }
              if (last) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (d0 + i < D && sp == 0) { // beams[0] + coding_dist.loc, :122
                    const int64_t ixo = src_index(A, base, pos, d0 + i);
                    A.out_sample[ixo] = nb[i] + A.p_loc[ixo];
                  }
              } else {
                *reinterpret_cast<float4 *>(bnew + (size_t)j * Dpad) = make_float4(nb[0], nb[1], nb[2], nb[3]);
              }
            }
          }
        }
      }
      if (tid < Bnew) beta4[(cur ^ 1) * TEAM_MB + tid] = bv_new;   // (wave 0: read by its own next selection)
      bv_cur = bv_new;
      cur ^= 1;
      Bcur = Bnew;
    }
    // ---- index path of beam 0 (beam_search_coder.py:118-121) ----
    tsync();
    if (gang_lost) {                           // partners not resident: the block is not coded (the caller codes the call again, IREC_FLAG_NO_SPLIT)
      if (tid == 0) A.out_K[blk] = -2;
      continue;
    }
    if (tid == 0 && gm == 0) {
      int j = 0;
      for (int t = K - 1; t >= 0; --t) {
        const int32_t v = __builtin_nontemporal_load(&bp[(size_t)t * NB + j]);
        A.out_indices[blk * (int64_t)A.max_K + t] = v >> 6;
        j = v & 63;
      }
    }
  }
}

#endif   // IREC_TEAM_MARGIN_TU
#ifndef IREC_TEAM_AUX_TU
// ======================================================================================================
//  proposal table with copy bits: tab[t][s][d] = dlog_g(r[s, d]) + 10006 * c   (uint16, row stride = D rounded up to 4)
//
//  The int32 draw of get_pseudo_random_sample (beam_search_coder.py:38-43) depends only on (seed + t, S, D): it is
//  evaluated once per call.  The block kernel's lane l of dim group g reads the quad d = 256 g + 4 l .. +3 of a row and
//  issues, per dim slot i and beam, one ds_read_b32 whose 32-lane groups are the quads [32 m, 32 m + 32) of the row.
//  For every such group and slot the 32 look-ups are spread over the banks by choosing c per lane: lane with
//  a = dlog mod 32 lands on bank a (c = 0) or a + 22 (c = 1), plus the beam's common rotation.  Since gcd(22, 32) = 2
//  the banks form two rings of 16 (p -> p + 1 is bank -> bank + 22) and a lane is an edge between neighbours; the
//  assignment minimising the busiest bank is found exactly: for L = 1, 2, ... and every x_0, push as many edges as
//  node p still takes (x_p = min(n_p, L - n_{p-1} + x_{p-1})) and test the closing node.
//  One half-wave per (t, s, m); choice bits never change any emitted value (all three table copies are identical).
// ======================================================================================================
// Round 4 (second half): ONE launch builds every table of the call (a latent's 1000-dim and residual-dim tables used to be
// two launches of 56 + 21 us at the default 32-step window); the rank of a look-up among those of its bank is the value an
// LDS atomic returns (any order serves: the x lowest ranks stay) instead of 32 ballots per slot; the four draws of a quad come
// from one Philox block (two where S * D is not a multiple of 4); the ring search starts at the average load and keeps one
// running value instead of three 16-entry arrays -- 82 VGPRs instead of 256 + 65 AGPRs, so several workgroups share a CU.
#ifndef IREC_CHOICE_WPE
#define IREC_CHOICE_WPE 2   // waves per SIMD the table-building kernels are compiled for
#endif
// rows with copy bits: workgroup `wg` of `n_wg` (256 threads: 8 half-waves); skip bit q set: table q is in place already
__device__ __forceinline__ void choice_table_rows(int64_t seed, int32_t S, int32_t K_tab, const uint16_t *__restrict__ dlog4r,
                                                  const ChoiceJobs &jobs, uint32_t skip, int64_t wg, int64_t n_wg) {
  __shared__ uint32_t n_s[8][4][32]; // [half-wave][slot][bank] look-ups whose c = 0 bank this is
  __shared__ uint8_t x_s[8][4][32];  // how many of them stay (c = 0)
  const int hwl = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int64_t n_hw = jobs.hw_end[jobs.n - 1];
  for (int64_t hw0 = wg * 8; hw0 < n_hw; hw0 += n_wg * 8) {
    const int64_t hwg = hw0 + hwl;
    int q = 0;
    while (q + 1 < jobs.n && hwg >= jobs.hw_end[q]) ++q;
    const int64_t hw = hwg - (q ? jobs.hw_end[q - 1] : 0);
    // (a table in place that was built for exactly this key is left alone)
    const bool hw_ok = hwg < n_hw && !((skip >> q) & 1u) && !(jobs.keep[q] && *jobs.keep[q]);
    const int D = jobs.D[q];
    const int Dp = (D + 3) & ~3;
    const int NQ = Dp >> 2;             // quads per row
    const int NM = (NQ + 31) >> 5;      // 32-lane groups per row
    const int64_t row = hw_ok ? hw / NM : 0;           // t * S + s
    const int m = hw_ok ? (int)(hw - row * NM) : 0;
    const int t = (int)(row / S), s = (int)(row - (int64_t)t * S);
    const int quad = 32 * m + j;
    const bool q_ok = hw_ok && quad < NQ;
    uint32_t al[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) n_s[hwl][i][j] = 0u;
    if (q_ok) {
      const StepSeed ss = make_step_seed(seed + t);
      uint32_t rm1[4];
      draw_rm1_x4(ss, (uint64_t)s * (uint64_t)D + (uint64_t)(4 * quad), rm1);   // (draws past the row's end are not used)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * quad + i < D) al[i] = (uint32_t)dlog4r[rm1[i]] >> 2;
    }
    __syncthreads();
    // rank of every look-up among those of its group with the same c = 0 bank, and the per-bank counts
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
    if (q_ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rank[i] = atomicAdd(&n_s[hwl][i][al[i] & 31u], 1u);
    }
    __syncthreads();
    if (j < 8 && hw_ok) { // 4 slots x 2 rings per half-wave
      const int slot = j >> 1, ring = j & 1;
      int n[16];
      int tot = 0;
#pragma unroll
      for (int p = 0; p < 16; ++p) { n[p] = (int)n_s[hwl][slot][(ring + 22 * p) & 31]; tot += n[p]; }
      int Lf = 32, x0f = n[0];           // (L = 32 with every look-up at c = 0 is always feasible)
      bool done = false;
      for (int L = tot > 16 ? (tot + 15) >> 4 : 1; L < 32 && !done; ++L)
        for (int x0 = 0; x0 <= n[0] && !done; ++x0) {
          int xp = x0;
          bool ok = true;
#pragma unroll
          for (int p = 1; p < 16; ++p) {
            const int ub = L - n[p - 1] + xp;
            ok = ok && ub >= 0;
            xp = n[p] < ub ? n[p] : (ub < 0 ? 0 : ub);
          }
          if (ok && x0 + n[15] - xp <= L) { done = true; Lf = L; x0f = x0; }
        }
      int xp = x0f;
      x_s[hwl][slot][ring] = (uint8_t)xp;
#pragma unroll
      for (int p = 1; p < 16; ++p) {
        const int ub = Lf - n[p - 1] + xp;
        xp = n[p] < ub ? n[p] : (ub < 0 ? 0 : ub);
        x_s[hwl][slot][(ring + 22 * p) & 31] = (uint8_t)xp;
      }
    }
    __syncthreads();
    if (q_ok) {
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = al[i] + (rank[i] < (uint32_t)x_s[hwl][i][al[i] & 31u] ? 0u : IREC_PM1);
      *reinterpret_cast<uint2 *>(jobs.tab[q] + (row * Dp + 4 * quad)) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    }
    __syncthreads(); // n_s / x_s are reused by the next round
  }
}
__global__ __launch_bounds__(256, IREC_CHOICE_WPE) void alpha_choice_kernel(int64_t seed, int32_t S, int32_t K_tab,
                                                           const uint16_t *__restrict__ dlog4r, ChoiceJobs jobs) {
  choice_table_rows(seed, S, K_tab, dlog4r, jobs, 0u, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// cost key of one row by one 256-thread workgroup: the row's KL summed in any order -- it places the row, it does not code it.
// (Two halves: all gathers of the call's statistics -- four dims per thread -- are issued before the first is used.)
struct CostRow { float v[4][4]; bool ok[4]; bool okD; int D; };
__device__ __forceinline__ void cost_row_issue(const EncArgs &A, int64_t blk, int t, CostRow &c) {
  c.D = A.block_dim[blk];
  const int64_t base = A.block_base[blk];
  const int32_t pos = A.block_pos[blk];
  c.okD = c.D >= 1 && c.D <= FAST_MAX_DIM;
  int64_t ix[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int d = t + 256 * i; c.ok[i] = c.okD && d < c.D; ix[i] = c.ok[i] ? src_index(A, base, pos, d) : 0; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c.v[i][0] = c.v[i][1] = c.v[i][2] = c.v[i][3] = 1.f;
    if (c.ok[i]) { c.v[i][0] = A.q_loc[ix[i]]; c.v[i][1] = A.q_scale[ix[i]]; c.v[i][2] = A.p_loc[ix[i]]; c.v[i][3] = A.p_scale[ix[i]]; }
  }
}
__device__ __forceinline__ void cost_row_finish(const PrepArgs &P, const EncArgs &A, int64_t blk, int t, const CostRow &c) {
  __shared__ double part[4];
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (c.ok[i]) acc = acc + kl_dim(c.v[i][0], c.v[i][1], c.v[i][2], c.v[i][3]);
  const double ws = wave_tree_sum(acc);
  if ((t & 63) == 0) part[t >> 6] = ws;
  __syncthreads();
  if (t == 0) {
    const double tot = ((part[0] + part[1]) + part[2]) + part[3];
    int32_t K = c.okD ? num_aux((float)tot, A.omega) : 0;
    K = K < 0 ? 0 : (K > (1 << 20) ? (1 << 20) : K);
    const uint32_t cst = (uint32_t)K * (uint32_t)(c.okD ? c.D : 0);
    P.cost[blk] = ((cst < (1u << 22) ? cst : (1u << 22) - 1u) << 10) | (uint32_t)blk;   // distinct keys: ties go to the lower row
  }
}

// ======================================================================================================
//  the call's preparation kernel (irec_kernels.h, "The call's preparation kernel"): books, exchange granules, row costs, tables
// ======================================================================================================
__global__ __launch_bounds__(256, IREC_CHOICE_WPE) void prep_kernel(PrepArgs P, EncArgs A) {
  const int t = (int)threadIdx.x;
  uint32_t *p = P.head;
  int wg = (int)blockIdx.x;
  if (wg == 0) { // ---- books
    for (int w = t; w < (int)(WS_COUNTER_BYTES / 4); w += 256) {
      const bool book = (w >= WS_KEEP_WORD && w < WS_KEEP_WORD + 4) || (w >= WS_STAMP_WORD && w < WS_STAMP_WORD + 4 * WS_STAMP_WORDS) ||
                        (w >= WS_PENDING_WORD && w < WS_PENDING_WORD + 4 * WS_STAMP_WORDS);
      if (!book) p[w] = 0u;
    }
    if (t < 4) {   // one thread owns a slot's words
      uint32_t *stamp = p + WS_STAMP_WORD + t * WS_STAMP_WORDS, *pend = p + WS_PENDING_WORD + t * WS_STAMP_WORDS;
      bool same = P.ts.reuse != 0 && P.ts.w[t][0] != 0u;
#pragma unroll
      for (int k = 0; k < WS_STAMP_WORDS; ++k) same = same && stamp[k] == P.ts.w[t][k];
      p[WS_KEEP_WORD + t] = same ? 1u : 0u;
#pragma unroll
      for (int k = 0; k < WS_STAMP_WORDS; ++k) pend[k] = P.ts.w[t][k];
      if (!same) {   // not this call's table: no word of the slot may pass for the key's until the encode kernel commits it
#pragma unroll
        for (int k = 0; k < WS_STAMP_WORDS; ++k) stamp[k] = ~P.ts.w[t][k];
      }
    }
    return;
  }
  wg -= 1;
  if (wg < P.n_granule) { // ---- exchange granules of shared block `wg`, both parities (16 KB): the step tags of the split encoder
                          // start at 1, so no granule of an earlier call on this workspace -- or whatever the memory held -- passes for one of this call's
    uint4 *x = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(p) + WS_COUNTER_BYTES);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      uint4 *xb = x + ((size_t)par * COOP_MAX_BLOCKS + (size_t)wg) * (COOP_KEYS * 8 / 16);
      for (int k = t; k < COOP_KEYS * 8 / 16; k += 256) xb[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  wg -= P.n_granule;
  if (wg < P.n_table_wgs) {
    // ---- proposal tables: which slots are in place?  (read-only; see irec_kernels.h for why the race with workgroup 0 is benign)
    uint32_t skip = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bool same = P.ts.reuse != 0 && P.ts.w[q][0] != 0u;
#pragma unroll
      for (int k = 0; k < WS_STAMP_WORDS; ++k) same = same && __builtin_nontemporal_load(p + WS_STAMP_WORD + q * WS_STAMP_WORDS + k) == P.ts.w[q][k];
      skip |= same ? (1u << q) : 0u;
    }
    if (P.table_kind == 1) choice_table_rows(P.seed, P.S, P.K_tab, P.dlog4r, P.jobs, skip, (int64_t)wg, (int64_t)P.n_table_wgs);
    else if (P.table_kind == 2) {
      // plain rows: the workgroups are dealt to the tables in proportion to their rows (jobs.hw_end counts 1024-entry units)
      int q = 0;
      while (q + 1 < P.jobs.n && (int64_t)wg >= P.jobs.hw_end[q]) ++q;
      const int64_t first = q ? P.jobs.hw_end[q - 1] : 0;
      if (!((skip >> q) & 1u)) plain_table_rows(P.seed, P.S, P.jobs.D[q], P.K_tab, P.dlog4r, P.jobs.tab[q], (int64_t)wg - first, P.jobs.hw_end[q] - first);
    }
    return;
  }
  wg -= P.n_table_wgs;
  // ---- cost key of row `wg`, a workgroup each, BEHIND the table workgroups (profiles/r06end/).  The key is three dependent gathers away
  // (descriptors -> permutation -> statistics, 5 000 random lines per row through the vector L1) and costs the launch 4 us (17 against 13)
  // here; in front of the tables, as in round 4, the cost workgroups held those back as well (18.5 us at 302 rows, 29 at 512:
  // prep_cost_first.log).  Also measured: the row riding on table workgroup `wg` with its gathers issued first (25 us: they did not overlap
  // with the table work, prep_cost_riding.log); a wave per row, four rows per workgroup (30 us, prep_cost_wave_per_row.log).
  if (wg < P.n_cost) {
    CostRow cr;
    cost_row_issue(A, wg, t, cr);
    cost_row_finish(P, A, wg, t, cr);
  }
}

#endif   // IREC_TEAM_AUX_TU
// ======================================================================================================
//  launchers
// ======================================================================================================
#ifndef IREC_TEAM_GANG_TU
// teams per workgroup / beam stripes per team.  Defaults: B <= 20: 3 x 1 where the LDS allows (else 2 x 1, or 1 x 2 with
// sample passes); B <= 30: 1 x 3.  Diagnostic overrides for B <= 20 travel in irec_params.flags (IREC_FLAG_SHAPE_*, no
// environment variable is read on the product path): cfg 20 = exactly two teams (also where three would be the default),
// 3 = three 4-wave teams (168 VGPRs), 12 = one 8-wave beam-striped team.  (Round 6: the one-team and the 2 x 2 shapes, which lost
// at every size, are gone with their builds <20,1,1> and <20,2,2>.)
static int team_cfg(int shape_override) {
  switch (shape_override) {
    case 2: return 20;
    case 3: return 3;
    case 5: return 12;
    default: return 2;
  }
}
// shape of the workgroup that serves B beams: beams per build, teams per workgroup, beam stripes per team
struct TeamShape { int nb, teams, bs; bool passes; bool one = false; };
static TeamShape team_shape(int B, int S, int ovr) {
  const int cfg = team_cfg(ovr);
  // 10 beams: G is 40 registers per lane, three teams fit the register file (168 VGPRs) and, for small S, the LDS: +9 %
  if (B == 1 && cfg == 2) {   // one beam: its own build (pipelined wide path), three teams, packed rows
    return TeamShape{10, 3, 1, team_s_pass(10, S, 3, 1024, false, 1) != S, true};
  }
  if (B <= 10) {
    const int ps = team_row(10, B);
    if (cfg == 3 || (cfg == 2 && team_s_pass(10, S, 3, 1024, false, ps) == S)) return TeamShape{10, 3, 1, false};
    if (cfg != 2 || team_s_pass(10, S, 2, 1024, false, ps) == S) return TeamShape{10, 2, 1, false};
    return TeamShape{10, 3, 1, true};   // more samples than one pass of two teams holds (S > 102): three teams, sample passes
  }
  if (B <= 20) {
    if (cfg == 20) return TeamShape{20, 2, 1, false};
    if (cfg == 12) return TeamShape{20, 1, 2, false};
    // three 4-wave teams (168 VGPRs: half-slot look-up pipeline, statistics / variance / scale parked in the slab, sort keys
    // over the partial scores) wherever their LDS fits next to the table copies -- S <= 38, the BASELINE workload: a third
    // team scores while another is in a serial phase (r02c: 13.2 ms against 13.7 ms of two teams)
    if (team_s_pass(20, S, 3, 1024) == S) return TeamShape{20, 3, 1, false};
    // more samples than two teams can hold in one pass (e.g. Omega = 5: S = 148): one 8-wave team with two beam stripes
    // and sample passes
    if (team_s_pass(20, S, 2, 1024) != S) return TeamShape{20, 1, 2, false};
    return TeamShape{20, 2, 1, false};
  }
  if (B <= 30) return TeamShape{30, 1, 3, false};   // one 12-wave team: three stripes of 10 beams (the B = 30 stress configuration)
  if (B <= 32) return TeamShape{32, 1, 2, false};   // one 8-wave team: two stripes of 16 beams
  // round 3: B = 50 of the reference's sweep (examples/lossless/data_aggregation.py:7).  One 12-wave team, three stripes of
  // 20 beams at 168 VGPRs (the three-team build's register diet: half-slot pipeline, parked state); a stripe that is at least
  // half alive scores its missing beams as phantoms
  // round 4: stripes of 18 beams for 48 < B <= 54 at S >= 128 -- B = 50 of the sweep then scores 54 beam slots instead of 60
  // (its third stripe holds 14 live beams and 4 phantoms instead of 10 and 10): -3 % time from S = 148 on (10.06 against
  // 9.71 look-ups/clk/CU at S = 1808), but its full-slot look-up pipeline at 168 VGPRs (416 B of scratch) loses 10-26 % to the
  // 60-beam build's half-slot diet below S ~ 100; stripes of 16 for B <= 48 lose at every S (B = 40 on the 60-beam build
  // leaves its third stripe idle anyway) and are not built (profiles/archive/r04g/).
  if (cfg != 3 && B > 48 && B <= 54 && S >= 128) return TeamShape{54, 1, 3, false};
  if (B <= 60) return TeamShape{60, 1, 3, false};
  return TeamShape{0, 0, 0, false};
}
#ifndef IREC_TEAM_AUX_TU
int team_count_for(int B, int S, int ovr) { return team_shape(B, S, ovr).teams; }
int team_shareable(int B, int S, int ovr) {
  const TeamShape sh = team_shape(B, S, ovr);
  return (sh.nb && sh.teams >= 2 && sh.bs == 1 && !sh.passes && !sh.one && (int64_t)S * sh.nb <= COOP_KEYS) ? sh.teams : 0;
}
bool team_placeable(int B, int S, int ovr) {   // the builds that can deal their rows by cost (CAN_PLACE in encode_team_kernel)
  const TeamShape sh = team_shape(B, S, ovr);
  return sh.nb == 20 && sh.teams == 2 && sh.bs == 1 && !sh.passes && !sh.one;
}
int team_waves_for(int B, int S, int ovr) { const TeamShape sh = team_shape(B, S, ovr); return sh.teams * sh.bs * TEAM_NW; }
size_t team_ws_extra_for(int B, int S, int ovr) { // scratch-slab bytes on top of fast_ws_for(): the sort keys when they do not fit the LDS
  const TeamShape sh = team_shape(B, S, ovr);
  const int ps = sh.nb ? team_row(sh.nb, B) : 0;
  return (sh.nb && !team_keys_in_lds(sh.nb, S, sh.teams, sh.passes, ps)) ? ((team_key_bytes(ps, S) + 255) & ~(size_t)255) : 0;
}
size_t team_ws_bytes_for(int B, int S, int ovr, int max_K) {
  const TeamShape sh = team_shape(B, S, ovr);
  return sh.nb ? fast_ws_bytes(sh.nb, max_K) + team_ws_extra_for(B, S, ovr) : 0;
}
// teams per workgroup of encode_ten_kernel (irec_ten.hip) when a plain call of this shape takes it, else 0
int team_ten_teams(int B, int S, int ovr) {
  const TeamShape sh = team_shape(B, S, ovr);
  if (!(sh.nb == 10 && sh.bs == 1 && !sh.passes && !sh.one && (sh.teams == 2 || sh.teams == 3) && ten_applies(B, S))) return 0;
  return sh.teams;   // (r06f: FOUR teams at 128 VGPRs hold the scoring loop without a spill and LOSE 12 %: 4096 latents 8.6 -> 9.7 ms)
}
const char *team_kernel_name(int B, int S, int ovr) {
  static thread_local char buf[64];
  const TeamShape sh = team_shape(B, S, ovr);
  snprintf(buf, sizeof buf, "encode_team_kernel<%d,%d,%d%s%s>", sh.nb, sh.teams, sh.bs, sh.passes ? ",passes" : "", sh.one ? ",one" : "");
  return buf;
}

size_t team_lds_for(int B, int S, int ovr) {
  const TeamShape sh = team_shape(B, S, ovr);
  if (!sh.nb) return (size_t)-1;
  if ((int64_t)S * sh.nb >= (1 << 24)) return (size_t)-1;
  const int ps = team_row(sh.nb, B);
  const int sp = team_s_pass(sh.nb, S, sh.teams, sh.teams == 1 ? 2048 : 1024, sh.passes, ps);
  if (sp < 1 || (sp < S && (sp < 16 || (sh.teams > 1 && !sh.passes)))) return (size_t)-1;   // does not fit (one-pass builds: in one
                                                                             // pass), or only in slivers: the one-table encoder takes it
  const size_t b = team_lds_total(sh.nb, S, sp, sh.teams, sh.passes, ps);
  return b <= FAST_LDS_LIMIT ? b : (size_t)-1;
}

#endif   // IREC_TEAM_AUX_TU

template <int NB, int TEAMS, int BS, bool PASSES = false, bool ONE = false, bool SHARE = false, bool MARGIN = false>
static hipError_t launch_team_t(const EncArgs &A, int grid, hipStream_t st) {
  const int ps = team_row(NB, A.B);
  const int sp = (TEAMS == 1 || PASSES) ? team_s_pass(NB, A.S, TEAMS, TEAMS == 1 ? 2048 : 1024, PASSES, ps) : A.S;
  const size_t lds = team_lds_total(NB, A.S, sp, TEAMS, PASSES, ps);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_team_kernel<NB, TEAMS, BS, PASSES, ONE, SHARE, MARGIN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((encode_team_kernel<NB, TEAMS, BS, PASSES, ONE, SHARE, MARGIN>), dim3(grid), dim3(TEAMS * BS * TEAM_NT), lds, st, A);
  return hipGetLastError();
}

#ifdef IREC_TEAM_MARGIN_TU
// ---- margin builds (irec_beam_encode_ex, IREC_FLAG_MARGINS): the shapes the BASELINE configurations run ----
static int team_margin_key(int B, int S, int ovr) {
  const TeamShape sh = team_shape(B, S, ovr);
  if (!sh.nb || sh.passes || sh.one) return 0;
  const int key = sh.nb * 100 + sh.teams * 10 + sh.bs;
  return (key == 2031 || key == 2021 || key == 1031 || key == 1021 || key == 3013) ? key : 0;
}
bool team_margin_build(int B, int S, int shape_override) { return team_margin_key(B, S, shape_override) != 0; }
hipError_t launch_encode_team_margin(const EncArgs &A, int grid, hipStream_t st) {
  if (A.out_margin == nullptr || A.coop_W > 1) return hipErrorInvalidValue;   // (no rows are shared under IREC_FLAG_MARGINS)
  switch (team_margin_key(A.B, A.S, A.shape_override)) {
    case 2031: return launch_team_t<20, 3, 1, false, false, false, true>(A, grid, st);
    case 2021: return launch_team_t<20, 2, 1, false, false, false, true>(A, grid, st);
    case 1031: return launch_team_t<10, 3, 1, false, false, false, true>(A, grid, st);
    case 1021: return launch_team_t<10, 2, 1, false, false, false, true>(A, grid, st);
    case 3013: return launch_team_t<30, 1, 3, false, false, false, true>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}
#else

hipError_t launch_encode_team(const EncArgs &A, int grid, hipStream_t st) {
  const TeamShape sh = team_shape(A.B, A.S, A.shape_override);
  const int key = sh.nb * 100 + sh.teams * 10 + sh.bs + (sh.passes ? 10000 : 0) + (sh.one ? 100000 : 0);
  if (A.coop_W > 1) {   // rows shared between teams (host: team_share_width -> team_shareable builds only)
    switch (key) {
      case 1021: return launch_team_t<10, 2, 1, false, false, true>(A, grid, st);
      case 2021: return launch_team_t<20, 2, 1, false, false, true>(A, grid, st);
      case 2031: return launch_team_t<20, 3, 1, false, false, true>(A, grid, st);
      default: return hipErrorInvalidValue;
    }
  }
  if ((key == 1031 || key == 1021) && !A.no_ten && ten_applies(A.B, A.S)) return launch_encode_ten(A, team_ten_teams(A.B, A.S, A.shape_override), grid, st);   // (irec_ten.hip)
  switch (key) {
    case 101031: return launch_team_t<10, 3, 1, false, true>(A, grid, st);
    case 111031: return launch_team_t<10, 3, 1, true, true>(A, grid, st);
    case 11031: return launch_team_t<10, 3, 1, true>(A, grid, st);
    case 6013: return launch_team_t<60, 1, 3>(A, grid, st);
    case 5413: return launch_team_t<54, 1, 3>(A, grid, st);
    case 1021: return launch_team_t<10, 2, 1>(A, grid, st);
    case 1031: return launch_team_t<10, 3, 1>(A, grid, st);
    case 2021: return launch_team_t<20, 2, 1>(A, grid, st);
    case 2031: return launch_team_t<20, 3, 1>(A, grid, st);
    case 2012: return launch_team_t<20, 1, 2>(A, grid, st);
    case 3013: return launch_team_t<30, 1, 3>(A, grid, st);
    case 3212: return launch_team_t<32, 1, 2>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}

#endif   // IREC_TEAM_MARGIN_TU (else)
#endif   // IREC_TEAM_GANG_TU

#ifndef IREC_TEAM_MARGIN_TU
// ---- chunked encoder (blocks of more than 1024 dims) ----
// The build that serves B beams and S samples: beam slots, beams per scoring pass, teams per workgroup -- the first of the candidates
// whose LDS fits next to the table copies (three teams only with passes of 10 beams: 168 VGPRs hold the G of ten, not of twenty).
struct ChunkShape { int nb, nbp, teams; };
static int chunk_nb(int B) { return B <= 10 ? 10 : B <= 20 ? 20 : B <= 30 ? 30 : B <= 32 ? 32 : B <= 40 ? 40 : B <= 50 ? 50 : B <= 60 ? 60 : 0; }
static ChunkShape chunk_shape(int B, int S) {
  // (round 6: {20, 20, 2} -- passes of twenty beams on two teams -- could never be chosen: wherever its LDS fits (S <= 49), that of {20, 10, 3}
  //  before it in the list does too (S <= 52); the planner enumeration of tests/test_kernel_coverage.py found it, the build is gone)
  static const ChunkShape cand[] = {{10, 10, 3}, {10, 10, 2}, {10, 10, 1}, {20, 10, 3}, {20, 10, 1},
                                    {30, 10, 3}, {30, 10, 2}, {30, 10, 1}, {32, 16, 2}, {32, 16, 1},
                                    {40, 10, 2}, {40, 10, 1}, {50, 10, 2}, {50, 10, 1}, {60, 10, 2}, {60, 10, 1}};
  const int nb = chunk_nb(B);
  if (!nb || (int64_t)S * nb > 4096) return ChunkShape{0, 0, 0};
  for (const ChunkShape &c : cand)
    if (c.nb == nb && chunk_lds_total(c.nb, c.nbp, S, c.teams) <= FAST_LDS_LIMIT) return c;
  return ChunkShape{0, 0, 0};
}
template <int NB, int NBP, int TEAMS, bool GANG = false>
static hipError_t launch_chunk_t(const EncArgs &A, int grid, hipStream_t st) {
  const size_t lds = chunk_lds_total(NB, NBP, A.S, TEAMS);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_chunk_kernel<NB, NBP, TEAMS, GANG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((encode_chunk_kernel<NB, NBP, TEAMS, GANG>), dim3(grid), dim3(TEAMS * TEAM_NT), lds, st, A);
  return hipGetLastError();
}
#ifdef IREC_TEAM_GANG_TU
// Gang builds (A.coop_W > 1, irec_team_gang.hip): the three-team shape of passes of ten beams where its LDS fits (B <= 30), else the one-team
// shape of the beam count -- a gang spreads its members over the CUs, so teams per workgroup only bound how many members a call may have.
static ChunkShape chunk_gang_shape(int B, int S) {
  const ChunkShape c = chunk_shape(B, S);
  if (!c.teams) return c;
  if (c.nb <= 30 && c.nbp == 10 && c.teams == 3) return c;
  return ChunkShape{c.nb, c.nb == 32 ? 16 : 10, 1};            // (the last candidate of every beam count: fits where any does)
}
int chunk_gang_teams(int B, int S) { return chunk_gang_shape(B, S).teams; }
int chunk_gang_nb(int B, int S) { return chunk_gang_shape(B, S).nb; }
size_t chunk_gang_lds_for(int B, int S) { const ChunkShape c = chunk_gang_shape(B, S); return chunk_lds_total(c.nb, c.nbp, S, c.teams); }
const char *chunk_gang_kernel_name(int B, int S) {
  static thread_local char buf[56];
  const ChunkShape c = chunk_gang_shape(B, S);
  snprintf(buf, sizeof buf, "encode_chunk_kernel<%d,%d,%d,gang>", c.nb, c.nbp, c.teams);
  return buf;
}
hipError_t launch_encode_chunk_gang(const EncArgs &A, int grid, hipStream_t st) {
  const ChunkShape c = chunk_gang_shape(A.B, A.S);
  if (!c.teams || A.max_dim_pad <= FAST_MAX_DIM || A.max_dim_pad > CHUNK_MAX_DIM) return hipErrorInvalidValue;
  if (A.coop_W < 2 || A.gang_chunks < 1 || A.coop_W % A.gang_chunks != 0 || !A.gang_xch || A.n_blocks > GANG_MAX_BLOCKS ||
      A.n_blocks * (int64_t)A.coop_W > (int64_t)grid * c.teams)
    return hipErrorInvalidValue;
  switch (c.nb * 1000 + c.nbp * 10 + c.teams) {
    case 10103: return launch_chunk_t<10, 10, 3, true>(A, grid, st);
    case 20103: return launch_chunk_t<20, 10, 3, true>(A, grid, st);
    case 30103: return launch_chunk_t<30, 10, 3, true>(A, grid, st);
    case 10101: return launch_chunk_t<10, 10, 1, true>(A, grid, st);
    case 20101: return launch_chunk_t<20, 10, 1, true>(A, grid, st);
    case 30101: return launch_chunk_t<30, 10, 1, true>(A, grid, st);
    case 32161: return launch_chunk_t<32, 16, 1, true>(A, grid, st);
    case 40101: return launch_chunk_t<40, 10, 1, true>(A, grid, st);
    case 50101: return launch_chunk_t<50, 10, 1, true>(A, grid, st);
    case 60101: return launch_chunk_t<60, 10, 1, true>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}
#else
int chunk_teams(int B, int S) { return chunk_shape(B, S).teams; }
bool chunk_applies(int B, int S, int max_dim) {
  return max_dim > FAST_MAX_DIM && max_dim <= CHUNK_MAX_DIM && chunk_shape(B, S).teams != 0;
}
size_t chunk_lds_for(int B, int S) { const ChunkShape c = chunk_shape(B, S); return chunk_lds_total(c.nb, c.nbp, S, c.teams); }
size_t chunk_ws_for(int B, int dpad, int max_K) { return chunk_ws_bytes(chunk_nb(B) ? chunk_nb(B) : 60, dpad, max_K); }
const char *chunk_kernel_name(int B, int S) {
  static thread_local char buf[48];
  const ChunkShape c = chunk_shape(B, S);
  snprintf(buf, sizeof buf, "encode_chunk_kernel<%d,%d,%d>", c.nb, c.nbp, c.teams);
  return buf;
}
hipError_t launch_encode_chunk(const EncArgs &A, int grid, hipStream_t st) {
  if (!chunk_applies(A.B, A.S, A.max_dim_pad)) return hipErrorInvalidValue;
  if (A.coop_W > 1) return launch_encode_chunk_gang(A, grid, st);
  const ChunkShape c = chunk_shape(A.B, A.S);
  switch (c.nb * 1000 + c.nbp * 10 + c.teams) {
    case 10103: return launch_chunk_t<10, 10, 3>(A, grid, st);
    case 10102: return launch_chunk_t<10, 10, 2>(A, grid, st);
    case 10101: return launch_chunk_t<10, 10, 1>(A, grid, st);
    case 20103: return launch_chunk_t<20, 10, 3>(A, grid, st);
    case 20101: return launch_chunk_t<20, 10, 1>(A, grid, st);
    case 30103: return launch_chunk_t<30, 10, 3>(A, grid, st);
    case 30102: return launch_chunk_t<30, 10, 2>(A, grid, st);
    case 30101: return launch_chunk_t<30, 10, 1>(A, grid, st);
    case 32162: return launch_chunk_t<32, 16, 2>(A, grid, st);
    case 32161: return launch_chunk_t<32, 16, 1>(A, grid, st);
    case 40102: return launch_chunk_t<40, 10, 2>(A, grid, st);
    case 40101: return launch_chunk_t<40, 10, 1>(A, grid, st);
    case 50102: return launch_chunk_t<50, 10, 2>(A, grid, st);
    case 50101: return launch_chunk_t<50, 10, 1>(A, grid, st);
    case 60102: return launch_chunk_t<60, 10, 2>(A, grid, st);
    case 60101: return launch_chunk_t<60, 10, 1>(A, grid, st);
    default: return hipErrorInvalidValue;
  }
}
#endif   // IREC_TEAM_GANG_TU
#endif   // IREC_TEAM_MARGIN_TU

#ifndef IREC_TEAM_AUX_TU
// workgroups that build the call's tables (kind 1: 8 half-waves per workgroup, at most 4096 workgroups, grid-stride; kind 2: per table
// one workgroup per 1024 entries, at most 1024 per table, grid-stride inside the table); fills jobs->D / hw_end / n
int64_t prep_table_wgs(int kind, int32_t S, int32_t K_tab, int n, const int32_t *dims, ChoiceJobs *jobs) {
  int64_t end = 0;
  for (int q = 0; q < n; ++q) {
    if (kind == 1) end += (int64_t)K_tab * S * ((((dims[q] + 3) >> 2) + 31) >> 5);   // half-waves: one per (step, sample, 32-quad group)
    else {
      const int64_t total = (int64_t)K_tab * S * ((dims[q] + 3) & ~3);
      end += std::min<int64_t>((total + 1023) / 1024, 1024);
    }
    jobs->D[q] = dims[q]; jobs->hw_end[q] = end;
  }
  jobs->n = n;
  if (kind == 1) { const int64_t want = (end + 7) / 8; return want < 4096 ? want : 4096; }
  return end;
}
hipError_t launch_prep(const PrepArgs &P, const EncArgs &A, hipStream_t st) {
  const int64_t grid = 1 + (int64_t)P.n_granule + P.n_table_wgs + P.n_cost;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)grid), dim3(256), 0, st, P, A);
  return hipGetLastError();
}
hipError_t launch_alpha_choice_all(int64_t seed, int32_t S, int32_t K_tab, const uint16_t *dlog4r, int n, const int32_t *dims,
                                   uint16_t *const *tabs, const uint32_t *const *keeps, hipStream_t st) {
  if (n < 1 || n > 4) return hipErrorInvalidValue;
  ChoiceJobs jobs{};
  const int64_t grid = prep_table_wgs(1, S, K_tab, n, dims, &jobs);
  for (int q = 0; q < n; ++q) { jobs.tab[q] = tabs[q]; jobs.keep[q] = keeps ? keeps[q] : nullptr; }
  hipLaunchKernelGGL(alpha_choice_kernel, dim3(grid > 0 ? (unsigned)grid : 1u), dim3(256), 0, st, seed, S, K_tab, dlog4r, jobs);
  return hipGetLastError();
}
hipError_t launch_alpha_choice(int64_t seed, int32_t S, int32_t D, int32_t K_tab, const uint16_t *dlog4r, uint16_t *tab,
                               const uint32_t *keep, hipStream_t st) {
  return launch_alpha_choice_all(seed, S, K_tab, dlog4r, 1, &D, &tab, &keep, st);
}

#endif   // IREC_TEAM_AUX_TU

} // namespace irec
