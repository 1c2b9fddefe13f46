// irec_kernels.h -- argument blocks and launchers shared by irec_kernels.hip and irec_host.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define IREC_MAX_PARTITIONS_DEV 65536

namespace irec {

constexpr int FAST_NW = 4;          // waves per workgroup of the fast encoder
constexpr int FAST_MAX_DIM = 1024;  // 4 dim groups x 64 lanes x 4 dims
constexpr size_t FAST_LDS_LIMIT = 160 * 1024;

struct EncArgs {
  // problem
  const int64_t *block_base; const int32_t *block_pos; const int32_t *block_dim; const int32_t *perm;
  const float *q_loc, *q_scale, *p_loc, *p_scale;
  int64_t n_blocks; int64_t seed;
  float omega; int32_t S, B, max_K;
  int32_t K_limit;   // partitions the context's auxiliary ratios cover: IREC_MAX_PARTITIONS_DEV (the power law of coder.py:16), or the length of the
                     // caller's fitted table (irec_create_with) -- a block that needs more is not coded (coder.py:222-229 raises for it)
  // outputs
  int32_t *out_K; int32_t *out_indices; float *out_sample;
  // constant tables of the context
  const float *lut;        // [10007] natural order: lut[k] = quantile(k/10007)
  const float *lut2;       // [10006] lut2[e] = lut[g^e mod 10007]
  const uint16_t *dlog4r;  // [10006] 4 * dlog_g(j + 1)
  const float *rho;        // [IREC_MAX_PARTITIONS_DEV] rho[i] = float32((i+1)^-0.7864636765648174)
  // scratch
  unsigned int *counter; char *ws; size_t ws_per_wg; int32_t max_dim_pad;
  unsigned int *xcd_counter;   // [8] block counters of the XCD-aware hand-out (irec_fast_common.h: xcd_static_row / xcd_pull_row)
  // shared proposal tables (beam-striped encoder): tab[q] serves blocks with block_dim == tab_dim[q]
  const uint16_t *tab[4]; int32_t tab_dim[4];
  // The tables hold the first K_tab steps only (bounded workspace: O(K_tab * S * D), not O(max_K * S * D)).  A table-fed
  // kernel leaves a block with K > K_tab uncoded (out_K[blk] = K is written) and bumps *defer_count; the host then runs
  // the fused-Philox encoder with deferred_pass = 1, which codes exactly the blocks with K_tab < out_K[blk] <= max_K.
  int32_t K_tab; int32_t deferred_pass; unsigned int *defer_count;
  int32_t shape_override;   // team-encoder workgroup shape (diagnostics; 0 = default)
  int32_t no_ten;           // IREC_FLAG_NO_TEN (diagnostics): plain calls of at most ten beams stay on encode_team_kernel<10,..>
  // Split encoder (small calls: fewer blocks than CUs).  coop_W > 1: workgroup blockIdx.x serves block blockIdx.x / coop_W
  // and scores the samples of stripe blockIdx.x % coop_W only; per step the coop_W workgroups of a block publish the sort keys
  // of their candidates in coop_xch [2][COOP_MAX_BLOCKS][COOP_KEYS] behind the arrival counter coop_arrive[block], read all of
  // the step's keys back and run the same selection.  *coop_err != 0: a workgroup gave up waiting (partners not resident):
  // block not coded.
  int32_t coop_W; unsigned int *coop_arrive; uint32_t *coop_xch; unsigned int *coop_err;
  // Shared rows of the team encoder (irec_team.hip, round 4): rows [tsplit_first, n_blocks) are coded by coop_W teams each, which
  // split the row's samples and exchange their sort keys through coop_xch (row r uses exchange slot r - tsplit_first).
  int64_t tsplit_first;
  // Cost-ordered hand-out (round 4): the key (K * dims) << 10 | row of every row, written by the call's preparation kernel (nullptr: rows
  // are dealt as listed).
  // Set only for calls whose hand-out slots all fall into the static round (every team takes at most one): workgroup w's first team
  // then takes the row of ascending cost rank w and its other teams the costliest rows (shared rows: the costliest), so that no CU
  // pairs two long rows while another pairs two short ones.
  const uint32_t *row_cost;
  // head of the workspace (nullptr: no table books, e.g. irec_block_kl): every encode kernel's first workgroup commits the table
  // stamps the call's preparation kernel left pending (commit_table_stamps, irec_fast_common.h)
  uint32_t *ws_head;
  // Gangs of the chunked encoder (irec_team.hip, round 5): calls of so few blocks of more than 1024 dims that most CUs would idle --
  // block_size = None on one image's latents.  coop_W teams code a block together, each the chunks c = member (mod coop_W) of it; per step
  // they hand the group sums of their chunks over through gang_xch (gang_stride bytes per block: gang_xch_bytes) behind an arrival
  // counter (first word of the block's granules in coop_xch), and every canonical sum is formed from all of its group sums in group order.
  char *gang_xch; size_t gang_stride;
  int32_t gang_chunks;        // the coop_W members of a gang = gang_chunks chunk owners x coop_W / gang_chunks sample stripes
  int32_t coop_beams;         // 1: the workgroups of a block share its beams (slots w, w + coop_W) instead of its samples
  int32_t coop_test_orphan;   // IREC_FLAG_TEST_SPLIT_ORPHAN: partners leave at once (exercises the give-up exit)
  // top-B margins (irec_beam_encode_ex, IREC_FLAG_MARGINS): [n_blocks][4] floats, see "top-B margins" in irec_fast_common.h; nullptr in
  // every other call.  Read by the margin builds of the team encoder (irec_team_margin.hip) and by the generic kernel only.
  float *out_margin;
  // diagnostics (IREC_STAMPS=1): per-workgroup cycle sums [grid][8]; nullptr in normal runs
  unsigned long long *dbg;
};

struct DecArgs {
  const int64_t *block_base; const int32_t *block_pos; const int32_t *block_dim; const int32_t *perm;
  const float *p_loc, *p_scale;
  int64_t n_blocks; int64_t seed;
  int32_t max_K; const int32_t *K; const int32_t *indices; float *out_sample;
  int32_t K_limit;   // as EncArgs::K_limit: a row with more partitions than the context's ratios cover is not decodable
  const float *lut; const float *rho;
  // wave-granular decoder (irec_decode.hip): upb = 256-dim units per block (0: the call gave no dim hints -> round-2 kernel),
  // quantile table in discrete-log order, and the per-call proposal tables tab[q][t][s][d] = 4 * dlog_g(r) of the first
  // K_tab steps for blocks of tab_dim[q] dims (K_tab = 0: no tables, the draw is fused into the kernel)
  int32_t upb; int32_t S; int32_t K_tab;
  const float *lut2; const uint16_t *dlog4r;
  const uint16_t *tab[4]; int32_t tab_dim[4];
  // tensor-staged decoder (tn > 0): n_tensors tensors of tn dims back to back, cut into blocks of tbs shuffled positions
  // (tbpt per tensor); K / index row of block j of tensor i: block_row[i * tbpt + j], or i * tbpt + j without a map
  int64_t n_tensors; int32_t tn, tbs, tbpt; const int32_t *block_row;
};

hipError_t launch_block_kl(const EncArgs &A, float *out_kl, int grid, hipStream_t st);
size_t generic_lds_bytes();
hipError_t launch_encode_generic(const EncArgs &A, int grid, hipStream_t st);
int fast_nb_for(int B);
size_t fast_lds_for(int B, int S, bool table);
size_t fast_ws_for(int B, int max_K);
size_t fast_ws_bytes_nb(int NB, int max_K);
hipError_t launch_encode_fast(const EncArgs &A, bool table, int grid, hipStream_t st);
int fast_waves_for(int B, int S, bool table);
int fast_split_beam_waves(int B);                        // waves per workgroup of the split encoder's beam mode
// (`keep`: device word that is 1 when the table in place already is this one -- the kernel
//  then returns at once; nullptr = always build)
hipError_t launch_alpha_table(int64_t seed, int32_t S, int32_t D, int32_t K_tab, const uint16_t *dlog4r, uint16_t *tab,
                              const uint32_t *keep, hipStream_t st);
// two-teams-per-CU encoder over three table copies (irec_team.hip)
// (shape_override: 0 = default shape, 1..4 = the diagnostic shapes of IREC_FLAG_SHAPE_*, see team_cfg() in irec_team.hip)
int team_count_for(int B, int S, int shape_override);   // teams per workgroup (= scratch slabs per workgroup) of the build that serves B beams
int team_shareable(int B, int S, int shape_override);   // teams per workgroup when that build can share rows between teams (several 4-wave teams, one pass, keys within the exchange), else 0
bool team_placeable(int B, int S, int shape_override);  // that build can deal the rows of a mid-size call by cost (EncArgs::row_cost)
int team_waves_for(int B, int S, int shape_override);   // waves per workgroup of that build
size_t team_ws_extra_for(int B, int S, int shape_override); // extra scratch-slab bytes of that build
size_t team_ws_bytes_for(int B, int S, int shape_override, int max_K);   // whole scratch slab of one team of that build
size_t team_lds_for(int B, int S, int shape_override);  // LDS bytes of one workgroup, or (size_t)-1 when the configuration is not served
hipError_t launch_encode_team(const EncArgs &A, int grid, hipStream_t st);
bool team_margin_build(int B, int S, int shape_override);   // a MARGIN build of the team encoder serves this shape (irec_team_margin.hip)
hipError_t launch_encode_team_margin(const EncArgs &A, int grid, hipStream_t st);
const char *team_kernel_name(int B, int S, int shape_override);   // e.g. "encode_team_kernel<20,2,1>"
const char *fast_kernel_name(int B, int S, bool table);
hipError_t launch_alpha_choice(int64_t seed, int32_t S, int32_t D, int32_t K_tab, const uint16_t *dlog4r, uint16_t *tab,
                               const uint32_t *keep, hipStream_t st);
// every proposal table of a call (one per distinct block dim, at most four) in ONE launch
hipError_t launch_alpha_choice_all(int64_t seed, int32_t S, int32_t K_tab, const uint16_t *dlog4r, int n, const int32_t *dims,
                                   uint16_t *const *tabs, const uint32_t *const *keeps, hipStream_t st);
// at most ten beams and S * 10 <= 256 candidates per step (the reference's default settings): one team barrier per step, the selection
// in every wave's registers (irec_ten.hip); plain calls of encode_team_kernel<10,3|2,1>'s shapes take it
bool ten_applies(int B, int S);
size_t ten_lds_for(int teams);
hipError_t launch_encode_ten(const EncArgs &A, int teams, int grid, hipStream_t st);
int team_ten_teams(int B, int S, int shape_override);   // teams per workgroup of encode_ten_kernel when a plain call of this shape takes it, else 0
// blocks of more than 1024 dims: a team walks the block in chunks of 1024 over the team encoder's tables (irec_team.hip, encode_chunk_kernel)
bool chunk_applies(int B, int S, int max_dim);          // B <= 60, max_dim > 1024, a step's partials and running scores fit the LDS next to the tables
int chunk_teams(int B, int S);                           // teams (= scratch slabs) per workgroup of the build that serves the call: 2, 1, or 0 = none
size_t chunk_lds_for(int B, int S);
size_t chunk_ws_for(int B, int dpad, int max_K);         // scratch slab of one team
const char *chunk_kernel_name(int B, int S);
// gang builds of the chunked encoder (irec_team_gang.hip): teams per workgroup (0 = none), beam slots, LDS, name
int chunk_gang_teams(int B, int S);
int chunk_gang_nb(int B, int S);
size_t chunk_gang_lds_for(int B, int S);
const char *chunk_gang_kernel_name(int B, int S);
hipError_t launch_encode_chunk_gang(const EncArgs &A, int grid, hipStream_t st);
hipError_t launch_encode_chunk(const EncArgs &A, int grid, hipStream_t st);
// one-beam calls: one wave per block over the team encoder's tables (irec_lone.hip)
bool lone_applies(int B, int shape_override);           // n_beams == 1 and no diagnostic shape pinned (IREC_FLAG_SHAPE_TEAM pins the team encoder)
int lone_waves();                                        // waves per workgroup (= blocks in flight per CU)
size_t lone_lds_bytes();
size_t lone_ws_bytes_per_wg();                           // scratch of one workgroup: a statistics slab per wave
const char *lone_kernel_name();
hipError_t launch_encode_lone(const EncArgs &A, int grid, hipStream_t st);
// Head of the workspace, 128 uint32: [0..3] block / deferred counters and the split encoder's error flag, [8..11] "keep"
// and behind them the eight per-XCD block counters of the batch encoders' first pass, [128 + 64 x]: 256 bytes apart, so that the
// adds of different XCDs do not queue on one line.
// words of the call's proposal tables, [16..47] the stamps of the four table slots (8 words each: what the table in place
// was built for), [64..127] arrival counters of the split encoder.  The preparation kernel of every call zeroes the counters,
// compares each slot's stamp with the call's key (IREC_FLAG_REUSE_TABLES; keep = 1 on a match, else 0) and stamps the key:
// its table workgroups make the same comparison themselves (`keep` is kept for diagnostics and the tests).
constexpr int WS_KEEP_WORD = 8, WS_STAMP_WORD = 16, WS_STAMP_WORDS = 8, WS_XCD_WORD = 128, WS_XCD_STRIDE = 64;
constexpr int WS_PENDING_WORD = 132;   // [132, 164): the call's table keys until the encode kernel commits them (inside XCD 0's counter line: only its first word counts)
struct TableStamps { uint32_t w[4][WS_STAMP_WORDS]; int32_t reuse; };   // all-zero key = slot unused (never matches)
// The call's preparation kernel (round 4; until then a head kernel and one table kernel per table: three to four launches of
// >= 4.5 us each before the block kernel): ONE launch whose workgroups
//   [0]                 keep the books: zero the counters, compare each table slot's stamp with the call's key (keep word = 1 on a
//                       match under IREC_FLAG_REUSE_TABLES), leave the key PENDING; a slot whose stamp differs is overwritten with the
//                       key's complement -- never equal to the key, whatever mixture of old and new words a table workgroup reads,
//   [1, 1 + n_granule)  zero the exchange granules of the call's shared blocks,
//   [.., + n_table_wgs) build the proposal tables of the slots whose stamp does not match (every table workgroup makes that
//                       comparison itself: read-only, and a mismatch stays a mismatch while workgroup 0 overwrites the stamp),
//   the rest (n_cost)   write the cost key (K * dims) << 10 | row of one row each (EncArgs::row_cost).
// The first workgroup of the encode kernel that follows copies the pending key over the stamp (commit_table_stamps): a table is
// stamped only once it has been built.
struct ChoiceJobs { int32_t D[4]; uint16_t *tab[4]; const uint32_t *keep[4]; int64_t hw_end[4]; int32_t n; };
struct PrepArgs {
  uint32_t *head; TableStamps ts;
  int32_t n_granule, n_cost, n_table_wgs, table_kind;   // table_kind: 0 none, 1 rows with copy bits (team / chunk encoders), 2 plain rows
  int64_t seed; int32_t S, K_tab; const uint16_t *dlog4r; ChoiceJobs jobs; uint32_t *cost;
};
hipError_t launch_prep(const PrepArgs &P, const EncArgs &A, hipStream_t st);
int64_t prep_table_wgs(int kind, int32_t S, int32_t K_tab, int n, const int32_t *dims, ChoiceJobs *jobs);   // fills jobs.hw_end / D / n
constexpr size_t WS_COUNTER_BYTES = 512 + 8 * 256;        // [0,256): counters, keep words, table stamps; [256,512): 64 arrival counters; [512,2560): XCD counters
constexpr int COOP_MAX_BLOCKS = 384, COOP_KEYS = 1024;     // key exchange: blocks per call that are shared (split encoder: <= 64 blocks of a small call; team encoder:
                                                           // the rows beyond one per CU of a mid-size call), sort keys per step (S * NB)
// How long a workgroup / team waits for the sort keys of its partners before it raises the sticky error flag and leaves (out_K = -2;
// s_memrealtime ticks of 10 ns): 100 ms.  A step's hand-off takes microseconds when the partners are resident; when they are not (a
// co-tenant holds CUs, two cooperating calls are in flight) no wait helps until that work is over -- 2 s until round 3, which turned a
// 0.2 ms call into 2 s in that case; now into 0.1 s before the call is coded again without sharing.
constexpr unsigned long long COOP_GIVE_UP_TICKS = 10000000ull;
constexpr int COOP_SPLIT_MAX_BLOCKS = 64;                  // the split encoder takes calls of at most this many blocks
#ifndef IREC_COOP_GRANULES
#define IREC_COOP_GRANULES 1   // split encoder: sort keys travel as 8-byte {key, step tag} granules that the partners sweep directly
                               // (0: 4-byte keys behind an arrival counter, r02-r03l)
#endif
constexpr bool IREC_COOP_GRANULES_ON = IREC_COOP_GRANULES != 0;
constexpr size_t WS_XCH_BYTES = (size_t)2 * COOP_MAX_BLOCKS * COOP_KEYS * 8;   // key exchange of the split encoder, double buffered: {key, tag} granules
// gang exchange of one block: group sums of the candidates [S * nb][NGm] and of the C_b terms [nb][NGm] (float), of the KL [NGm] (double),
// the step's sort keys [S * nb]; NGm = 4 * chunks of the call's largest block (dpad: its dims rounded up to 256)
__host__ __device__ inline size_t gang_xch_bytes(int nb, int S, int dpad) {
  const size_t NGm = (size_t)4 * (((size_t)dpad + 1023) >> 10), NC_ = (size_t)S * nb;
  return (4 * NGm * (NC_ + nb) + 8 * NGm + 4 * NC_ + 255) & ~(size_t)255;
}
#ifndef IREC_GANG_STRIPES
#define IREC_GANG_STRIPES 9   // sample stripes per chunk of a gang, at most (r05s/gang_stripes.log: one block of 8192 dims 5.2 / 3.5 / 2.9 / 2.5 / 2.3 ms with
                              // 1 / 2 / 4 / 6 / 9; flat from there to 15; IREC_FLAG_SPLIT_* bits: a call's own cap)
#endif
constexpr int GANG_MAX_BLOCKS = COOP_MAX_BLOCKS;                 // blocks of a call coded by gangs (their arrival counters: the head's exchange granules)
constexpr size_t GANG_XCH_BYTES_MAX = (size_t)1 << 30;          // ... and no more of them than this much exchange holds
constexpr int COST_MAX_ROWS = 1024;                              // rows of a call whose hand-out is cost-ordered (EncArgs::row_cost)
constexpr size_t WS_COST_BYTES = (size_t)COST_MAX_ROWS * 4;
constexpr size_t WS_HEAD_BYTES = WS_COUNTER_BYTES + WS_XCH_BYTES + WS_COST_BYTES;   // (the row costs lie behind the exchange granules)
hipError_t launch_decode(const DecArgs &A, int n_cu, hipStream_t st);   // irec_decode.hip
int decode_tensor_waves(int n, int bs, bool table, size_t *lds_out);     // waves per workgroup of the tensor-staged decoder, 0 = does not apply
// elementwise hand-offs of the RVAE host shim (irec_shim.hip)
hipError_t launch_shim_stats(const float *y, const float *inf, float *out, int n_stats, int N, int Cy, int Ci, int s, int HW,
                             const float *by, const float *bi, hipStream_t st);
hipError_t launch_shim_cat_elu(const float *y, const float *latent, float *out, int N, int Cy, int c_off, int d, int s, int HW,
                               const float *by, hipStream_t st);
hipError_t launch_shim_residual_elu(const float *inp, const float *t, float alpha, float *out, float *out_elu, int64_t count,
                                    const float *bt, int C, int HW, hipStream_t st);
hipError_t launch_dec_sqrt_test(unsigned long long *out, hipStream_t st);
hipError_t launch_uniform_int(int64_t seed, int64_t n, int32_t *out, hipStream_t st);
hipError_t launch_select_test(const float *scores, int N, int Bnew, int Bcur, uint32_t *keys, int32_t *sel, bool quick, hipStream_t st);
hipError_t launch_reduce_scatter_test(const float *in, float *out, int width, hipStream_t st);

} // namespace irec
