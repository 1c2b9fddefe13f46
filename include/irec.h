/*
 * irec.h -- C ABI of libirec_hip.so: the MI355X-native iREC beam-search coder.
 *
 * This is the drop-in boundary for the hot path of gergely-flamich/relative-entropy-coding:
 *   rec/coding/beam_search_coder.py  BeamSearchCoder.encode_block / decode_block
 *   rec/coding/coder.py              GaussianCoder.encode / decode, Coder.split / merge
 * Plain pointers and sizes only; no torch / HIP types in the signatures (a HIP stream is passed as void*).
 *
 * Conventions
 *   - "device pointer" arguments must point to memory of the HIP device the context was created on.
 *   - Device entry points are ASYNCHRONOUS on the given stream; they never allocate, free or synchronise.
 *   - The library keeps no hidden mutable state besides the read-only constant tables of a context
 *     (the reference mutates TF's process-global RNG on every call: beam_search_coder.py:38, coder.py:62,111).
 *   - Every function returns an irec_status (0 = ok, negative = error); irec_last_error() gives the text
 *     of the calling thread's last error.
 *
 * Blocks.  A "block" is one <= block_size slice of a shuffled latent tensor (coder.py:69-83).  For block k:
 *     element i (0 <= i < block_dim[k]) lives at flat index
 *         block_base[k] + ( perm ? perm[block_pos[k] + i] : block_pos[k] + i )
 *     of q_loc / q_scale / p_loc / p_scale / out_sample, where block_base[k] is the offset of the block's tensor
 *     inside the concatenated arrays, block_pos[k] the block's start inside the shuffled tensor and perm the
 *     tf.random.shuffle permutation of that tensor size (irec_tf_shuffle_perm), shared by all tensors of the call.
 *     Reading through perm IS Coder.split; writing out_sample through it IS Coder.merge.
 */
#ifndef IREC_H_
#define IREC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IREC_BIG_PRIME 10007     /* beam_search_coder.py:30 */
#define IREC_MAX_BEAMS 256       /* hard limit of this build (reference: any python int, beam_search_coder.py:28); B <= 60 takes the */
                                 /* register-resident encoders, anything above the generic kernel                                    */
#define IREC_MAX_PARTITIONS 65536 /* hard limit on K = ceil(KL / kl_per_partition) */

typedef enum {
  IREC_OK = 0,
  IREC_E_INVALID = -1,     /* bad argument (null pointer, non-positive size, B > IREC_MAX_BEAMS ...)            */
  IREC_E_HIP = -2,         /* a HIP runtime call failed; text in irec_last_error()                             */
  IREC_E_NO_DEVICE = -3,   /* no usable gfx950 device: the library has NO CPU fallback                         */
  IREC_E_WORKSPACE = -4    /* workspace smaller than irec_encode_workspace_bytes()                             */
} irec_status;

/* Constructor arguments of BeamSearchCoder (beam_search_coder.py:15-30) after its own preprocessing. */
typedef struct {
  float kl_per_partition; /* Omega, stored as float32 by GaussianCoder.__init__ (coder.py:192)                 */
  int32_t n_samples;      /* S = int(exp(kl_per_partition * extra_samples)) (beam_search_coder.py:29)          */
  int32_t n_beams;        /* B (beam_search_coder.py:28)                                                       */
  int32_t flags;          /* IREC_FLAG_* below; 0 = defaults                                                   */
  int32_t table_dims[4];  /* optional hint: the distinct block_dim values of the call (0-terminated, <= 4).     */
                          /* With it the encoder evaluates the shared-seed Philox draw once per call into a     */
                          /* proposal table (same seed for every block, coder.py:444-449) instead of once per   */
                          /* block; a block whose dim is not listed gets out_K = -1.  All zero = no hint.       */
  int32_t table_steps;    /* partitions the proposal tables cover (0 = IREC_TABLE_STEPS_DEFAULT), clamped to    */
                          /* [1, min(max_K, IREC_TABLE_STEPS_MAX)] and to what IREC_TABLE_BYTES_MAX holds of the */
                          /* call's tables (2 * S * sum of the table dims bytes per step; at least               */
                          /* IREC_TABLE_STEPS_FLOOR steps within IREC_TABLE_BYTES_HARD).  Table scratch is       */
                          /* O(table_steps * S * D): a block with more partitions is coded by the fused-Philox   */
                          /* kernel in a second pass of the same call (blocks of more than 1024 dims: inside the */
                          /* chunked encoder, which draws the rows of those steps itself) -- same outputs.       */
} irec_params;

#define IREC_FLAG_FORCE_GENERIC 1 /* use the generic (any D, any B) kernel even where the fast kernels apply   */
#define IREC_FLAG_NO_SPLIT 16     /* no block is shared: neither the split encoder (several workgroups per block for calls of few     */
                                  /* blocks) nor shared rows between teams (calls of one to 1.5 blocks per CU, B > 10) nor gangs of   */
                                  /* teams (calls of at most 384 blocks of more than 1024 dims: several teams per block)              */
#define IREC_FLAG_TABLES_PRESENT 65536 /* stronger than REUSE_TABLES: the caller vouches that the PREVIOUS irec_beam_encode call on this  */
                                  /* workspace and stream was this call's twin (same seed, S, table_dims, table window, table kind --  */
                                  /* e.g. the next residual block of the same image) and that nothing has run on the workspace since:   */
                                  /* the table kernels are not even launched (two launches fewer per call of a latency-bound pass).     */
#define IREC_FLAG_REUSE_TABLES 64 /* the caller vouches for the workspace: its first 512 bytes were zero when it was allocated  */
                                  /* and nothing but irec_beam_encode has written to it since.  Every call stamps the key of  */
                                  /* each proposal table it builds (seed, S, D, table window, table kind, offset) into the    */
                                  /* workspace head; with this flag a later call on the SAME workspace whose key matches the  */
                                  /* stamp -- checked on the device, so also inside a replayed HIP graph -- keeps the table   */
                                  /* instead of rebuilding it (the 24 residual blocks of an image share seed, S and dims:     */
                                  /* beam_search_coder.py:38-43, resnet_vae.py:822-824).  Same outputs, bit for bit.          */
#define IREC_FLAG_MARGINS 524288   /* the call reports its top-B margins: irec_beam_encode_ex with out_margin (and only that entry point) takes it.  The    */
                                  /* flag is part of the params because it sizes the workspace (irec_encode_workspace_bytes) and picks the kernels     */
                                  /* (irec_encode_plan): a margin build of the team encoder where one exists, the generic kernel otherwise; no block   */
                                  /* is shared between workgroups or teams.  Indices, K and samples are the plain call's, bit for bit.                 */
/* (diagnostic flag bits -- pinned kernel shapes, A/B switches, test hooks -- and the unit-test entry points: csrc/irec_internal.h) */
#define IREC_TABLE_STEPS_DEFAULT 32
#define IREC_TABLE_STEPS_MAX 4096
#define IREC_TABLE_BYTES_MAX (64u << 20)
#define IREC_TABLE_STEPS_FLOOR 8            /* ... but the byte bound never cuts the window below 8 steps (one step of S = 8103, */
#define IREC_TABLE_BYTES_HARD (1u << 30)    /* the reference's largest, is 19 MB for 1000 + 192 dims) unless those exceed 1 GB   */
#define IREC_SLAB_BYTES_MAX (16ull << 30)   /* scratch slabs of a call whose blocks exceed 1024 dims: one per resident team, but no more than this  */
                                            /* holds (a 301 056-dim block needs 55 MB of slab: 290 teams instead of 768 code such a call)         */
#define IREC_TABLE_BYTES_BIG (4ull << 30)   /* the bound of calls whose blocks exceed 1024 dims (block_size = None on a whole tensor:  */
                                            /* K grows with the dims -- an 8192-dim block has ~60 partitions of 590 KB of rows each)   */

/* What irec_beam_encode does for a given call: filled by irec_encode_plan (same decision code as the launch). */
typedef struct {
  char kernel[64];         /* block kernel, e.g. "encode_team_kernel<20,2,1>"                                   */
  char table_kernel[32];   /* who builds the proposal tables: "prep_kernel (copy bits)" / "prep_kernel (plain rows)" -- the call's one */
                           /* preparation launch (books, exchange granules, row costs, tables) -- or "" (Philox fused in the block kernel) */
  int32_t grid;            /* workgroups of the block kernel                                                    */
  int32_t waves_per_wg;
  int32_t teams_per_wg;    /* independent teams inside a workgroup (1 for the one-workgroup-per-block encoders) */
  int32_t lds_bytes;       /* dynamic LDS of one workgroup                                                      */
  int32_t table_steps;     /* partitions the proposal tables cover (0 = no tables)                              */
  int32_t n_tables;
  int32_t split;           /* workgroups that share one block (split encoder of calls of < 64 blocks), or teams that share each */
                           /* row beyond one per CU (team encoder, calls of one to 1.5 blocks per CU), or the teams of a gang  */
                           /* (chunked encoder, calls of <= 384 blocks of more than 1024 dims: chunk owners x sample stripes,  */
                           /* kernel "...,gang>"); 0 = nothing is shared                                                       */
  int32_t n_cu;            /* compute units of the context's device                                             */
  int32_t clock_mhz;       /* its maximum engine clock                                                          */
  int32_t split_beams;     /* split encoder: 1 = the workgroups of a block share its beams (each owns <= 2 beam slots,  */
                           /* scores every sample for them, forms only its own new beams), 0 = they share its samples  */
  int64_t table_bytes;     /* proposal tables inside the workspace                                              */
  int64_t workspace_bytes; /* = irec_encode_workspace_bytes()                                                   */
} irec_plan_info;

typedef struct irec_context irec_context;

/* ---- host-only helpers ---------------------------------------------------------------------------------- */
const char *irec_last_error(void);
const char *irec_version(void);

/* int(np.exp(kl_per_partition * extra_samples))  -- beam_search_coder.py:29 */
int32_t irec_n_samples(double kl_per_partition, double extra_samples);

/* len(indices) * np.log(n_samples)  -- BeamSearchCoder.get_codelength, beam_search_coder.py:150-151 (nats) */
double irec_codelength(int64_t n_indices, int32_t n_samples);

/* lut[k] = Normal(0,1).quantile(float32(k)/10007) for k = 1..10006, lut[0] = 0
 * -- the 10006 values dist.quantile can take in get_pseudo_random_sample, beam_search_coder.py:45-49. */
irec_status irec_build_lut(float *lut10007);

/* perm = tf.random.shuffle(tf.range(n)) after tf.random.set_seed(seed) -- Coder.split/merge, coder.py:62-64,111-113.
 * Host memory. */
irec_status irec_tf_shuffle_perm(int64_t seed, int64_t n, int64_t *perm);

/* ---- importance sampler (config 1 plumbing; the reference runs it on the CPU, and so does this: host memory) ---------
 * encode_gaussian_importance_sample (rec/coding/importance_sampling.py:9-79): standardise the target w.r.t. the coder,
 * draw n_samples = ceil(exp(coding_bits * log 2)) proposals x[s] ~ N(0,1)^n from the stream of tf.random.set_seed(seed);
 * tfd.Normal(0,1).sample(n_samples) (:37,50-53), weight them by w[s] = sum_d [log N(x; t', s') - log N(x; 0, 1)] (:57-58).
 *   alpha = inf : index = argmax_s w[s] (first one on ties, tf.argmax :64)
 *   1 <= alpha  : index = argmax_s alpha * w[s] + g[s], g = stateless_gumbel_sample([n_samples], seed + 1) (:67-71;
 *                 rec/coding/utils.py:9-12 puts a stateless NORMAL draw inside -log(-log(.)), so g is NaN for most s and
 *                 tf.argmax skips those: reproduced as written)
 *   alpha < 1   : error (:33-34)
 * Returns the index and p_scale * x[index] + p_loc (:73-76).  n = number of dims of the (flattened) distributions. */
irec_status irec_importance_encode(const float *t_loc, const float *t_scale, const float *p_loc, const float *p_scale,
                                   int64_t n, double coding_bits, double alpha, int64_t seed, int64_t *out_index,
                                   float *out_sample);
/* decode_gaussian_importance_sample (importance_sampling.py:82-103): sample `index` of the same stream. */
irec_status irec_importance_decode(const float *p_loc, const float *p_scale, int64_t n, int64_t index, int64_t seed,
                                   float *out_sample);
/* n_samples of the call above (float32 arithmetic as in importance_sampling.py:50), or -1 if it does not fit int32. */
int64_t irec_importance_n_samples(double coding_bits);
/* out[e] = element e of tf.random.uniform([n], 1, 10007, seed=seed, dtype=int32) after tf.random.set_seed(seed)
 * -- beam_search_coder.py:38-43.  Host memory; test hook for the in-kernel Philox stream. */
irec_status irec_philox_uniform_int(int64_t seed, int64_t n, int32_t *out);

/* ---- context ---------------------------------------------------------------------------------------------- */
/* Builds the constant tables (quantile LUT in discrete-log order, discrete-log table of Z_10007^*, power-law ratios
 * get_auxiliary_ratio, coder.py:16,218-220) and uploads them to HIP device `device`.
 * Fails with IREC_E_NO_DEVICE when there is no GPU: there is no CPU fallback. */
irec_status irec_create(int device, irec_context **out);
/* irec_create with the path's TF-dependent primitive supplied by the caller.
 *   lut10007  host float [10007], may be NULL (= irec_build_lut): lut10007[k] = the float32 that
 *             tfd.Normal(0, 1).quantile(float32(k) / 10007) returns in the CALLER's TensorFlow, k = 1..10006 (entry 0 is never
 *             indexed) -- the 10006 values `dist.quantile` can take at beam_search_coder.py:48-49 before the scale is applied.
 *             Every kernel of the context (encoders, decoders) reads its quantiles from this table and from nothing else.
 * Why: irec_build_lut restates TFP 0.9's float32 ndtri with a correctly rounded log; TensorFlow evaluates it over Eigen's
 * vectorised log, and a 1-ulp difference in one entry can move an emitted index.  A maintainer with a TF 2.1 machine dumps
 * the table once (scripts/make_tf_vectors.py writes it as `quantile` in tf_primitives.npz) and creates the context with it:
 * the stream then decodes on the TensorFlow side, no recompile.  (The other TF-dependent input, the tf.random.shuffle
 * permutation of Coder.split, is a caller argument already: `perm` of irec_beam_encode / irec_beam_decode.)
 * All entries 1..10006 must be finite (IREC_E_INVALID otherwise). */
irec_status irec_create_ex(int device, const float *lut10007, irec_context **out);
/* The context's caller-supplied tables in one (extensible) struct; zero-initialise it, every member is optional.
 *   lut10007       as irec_create_ex.
 *   aux_ratios     host float [n_aux_ratios]: the FITTED auxiliary-variance ratios of a coder constructed with
 *                  extrapolate_auxiliary_ratios=False -- the values of its `aux_variable_variance_ratios` variable, which
 *                  GaussianCoder.get_auxiliary_ratio(index) returns in place of the power law (coder.py:203-231; the SGD fitter that
 *                  produces them, coder.py:233-410, stays on the caller's side: SURVEY.md §2).  Every entry must lie in (0, 1].
 *                  A block whose K = ceil(KL / Omega) exceeds n_aux_ratios is NOT coded (out_K = K is still written, as for K > max_K):
 *                  the reference raises "KL divergence higher than auxiliary variables can account for" there (coder.py:222-229), and so
 *                  does the Python mirror.  The decoder treats such a row as not decodable.  irec_max_partitions(ctx) tells the bound. */
typedef struct {
  const float *lut10007;
  const float *aux_ratios;
  int32_t n_aux_ratios;
} irec_tables;
irec_status irec_create_with(int device, const irec_tables *tables, irec_context **out);
/* Partitions the context's auxiliary ratios cover: IREC_MAX_PARTITIONS (power law) or n_aux_ratios. */
int32_t irec_max_partitions(const irec_context *ctx);
void irec_destroy(irec_context *ctx);

/* Bytes of device scratch irec_beam_encode needs for blocks of at most max_dim dims and max_K partitions. */
size_t irec_encode_workspace_bytes(const irec_context *ctx, const irec_params *p, int32_t max_dim, int32_t max_K);

/* The same for ONE call of n_blocks blocks (round 6).  Differs from the bound above only for blocks of more than 1024 dims (block_size = None,
 * 2048, ...): their scratch slab is (6 + 2 B) x dims x 4 bytes per team -- 55 MB at 301 056 dims -- and the bound above reserves one per team slot
 * of the device (at most IREC_SLAB_BYTES_MAX = 16 GB, plus up to IREC_TABLE_BYTES_BIG = 4 GB of proposal tables and 1 GB of gang exchange),
 * whereas a call uses one per team it launches: n_blocks x (teams per gang).  irec_beam_encode takes either size, or any size in between (it launches
 * no more teams than the workspace has slabs; same results). */
size_t irec_encode_workspace_bytes_for(const irec_context *ctx, const irec_params *p, int64_t n_blocks, int32_t max_dim, int32_t max_K);

/* The kernels, grid and scratch irec_beam_encode(ctx, p, n_blocks, ..., max_block_dim, ..., max_K, ...) launches.
 * Host only, no device work; bench.py reports the kernel it measured from this instead of a string literal. */
irec_status irec_encode_plan(const irec_context *ctx, const irec_params *p, int64_t n_blocks, int32_t max_block_dim,
                             int32_t max_K, irec_plan_info *out);

/* ---- device entry points ---------------------------------------------------------------------------------- */

/* total_kl and num_aux_variables of every block -- beam_search_coder.py:57-59.
 *   out_kl [n_blocks] float32 (may be NULL), out_K [n_blocks] int32.  Device pointers. */
irec_status irec_block_kl(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                          const int32_t *block_pos, const int32_t *block_dim, const int32_t *perm,
                          const float *q_loc, const float *q_scale, const float *p_loc, const float *p_scale,
                          float *out_kl, int32_t *out_K, void *hip_stream);

/* BeamSearchCoder.encode_block on n_blocks independent blocks -- beam_search_coder.py:53-122, driven the way
 * GaussianCoder.encode drives it (same seed for every block, coder.py:444-449).
 *   max_block_dim                   upper bound of block_dim[] (host knows it: block_size); a block with more dims
 *                                   is not coded and gets out_K = -1
 *   out_K       [n_blocks]          K of each block.  K > max_K means "not coded: retry with a larger max_K"; -1: the block's
 *                                   dim is not covered by max_block_dim / table_dims; -2: the block was shared between workgroups or
 *                                   teams and its partners were not all resident within 100 ms (a co-tenant on the device, two
 *                                   cooperating calls in flight): call again, or with IREC_FLAG_NO_SPLIT.
 *   out_indices [n_blocks, max_K]   idx[t], t < K: the sample index chosen at iteration t (rest untouched)
 *   out_sample  flat, same indexing as the inputs: beams[0] + p.loc, merged
 *   workspace   device scratch of at least irec_encode_workspace_bytes() -- or irec_encode_workspace_bytes_for(n_blocks) -- bytes, 256-byte aligned
 * All pointers except p are device pointers. */
irec_status irec_beam_encode(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                             const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                             const int32_t *perm, const float *q_loc, const float *q_scale, const float *p_loc,
                             const float *p_scale, int64_t seed, int32_t max_K, int32_t *out_K, int32_t *out_indices,
                             float *out_sample, void *workspace, size_t workspace_bytes, void *hip_stream);

/* irec_beam_encode that also says HOW CLOSE every block's selections were (round 5) -- the top-B step of beam_search_coder.py:85-89
 * (`tf.argsort(..., direction='DESCENDING')[:n_beams]`) decides the emitted indices through two comparisons only: which candidates
 * make the top-B SET of every step but the last, and which candidate WINS the last step (beams[0], :118-122).  Another float32
 * summation order of the same terms -- TensorFlow's reduce_sum (:84) -- can move an index only where those
 * comparisons are closer than the two orders disagree, so the margins measure the encoder-side exposure of index parity.
 *   out_margin [n_blocks][4] float32, device pointer; needs IREC_FLAG_MARGINS in p->flags (NULL without it: plain irec_beam_encode):
 *     [0] min over the steps t < K - 1 that reject a candidate of  score(rank Bnew - 1) - score(rank Bnew), Bnew = min(B, S * Bcur):
 *         the gap between the last candidate kept and the best one rejected; +inf when there is no such step (K <= 1)
 *     [1] |score(rank Bnew - 1)| at that step (scale of the sum the gap is a difference of)
 *     [2] score(rank 0) - score(rank 1) at the last step (+inf: a single candidate)        [3] |score(rank 0)| at the last step
 *   Scores are the float32 values the selection ranks, differences are float32 subtractions: the oracle's traced
 *   scores give the same four numbers bit for bit.  A block that is not coded (K = 0, K > max_K, K < 0) reports {+inf, 0, +inf, 0}.
 *   Decoding needs none of this: decode_block never compares scores (beam_search_coder.py:124-148). */
irec_status irec_beam_encode_ex(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                                const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                                const int32_t *perm, const float *q_loc, const float *q_scale, const float *p_loc,
                                const float *p_scale, int64_t seed, int32_t max_K, int32_t *out_K, int32_t *out_indices,
                                float *out_sample, float *out_margin, void *workspace, size_t workspace_bytes, void *hip_stream);

/* BeamSearchCoder.decode_block on n_blocks blocks -- beam_search_coder.py:124-148 (GaussianCoder.decode, coder.py:459-491).
 *   K [n_blocks], indices [n_blocks, max_K] in ENCODER order (idx[t] = choice at iteration t).  A row with K < 0 or
 *   K > max_K (what the encoder leaves for a block it did not code), or one that holds an index outside [0, n_samples) (a
 *   damaged or foreign stream; the reference fails in tf.gather), is not decodable: the block's elements are returned as
 *   p_loc -- by every decode entry point, and no table row outside the call's tables is ever addressed.  So is a block whose
 *   dim exceeds the call's bound (max_block_dim, or the largest listed table dim when there is none). */
irec_status irec_beam_decode(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                             const int32_t *block_pos, const int32_t *block_dim, const int32_t *perm,
                             const float *p_loc, const float *p_scale, int64_t seed, int32_t max_K, const int32_t *K,
                             const int32_t *indices, float *out_sample, void *hip_stream);

/* The same decode with caller-provided scratch (irec_decode_workspace_bytes(), 256-byte aligned; may be NULL / 0): with
 * p->table_dims set the shared-seed draw of the call is evaluated once into per-call proposal tables (every block of one dim
 * count reads its rows from the same S rows per step, beam_search_coder.py:38-43,141-146) instead of once per block -- same
 * outputs, bit for bit.  max_block_dim: upper bound of block_dim[] (>= every listed table dim).  A block whose dim exceeds it
 * is not decoded.  With table_dims set, irec_beam_decode above takes the same kernel without the tables. */
size_t irec_decode_workspace_bytes(const irec_context *ctx, const irec_params *p, int32_t max_K);
irec_status irec_beam_decode_ws(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                                const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                                const int32_t *perm, const float *p_loc, const float *p_scale, int64_t seed, int32_t max_K,
                                const int32_t *K, const int32_t *indices, float *out_sample, void *workspace,
                                size_t workspace_bytes, void *hip_stream);

/* GaussianCoder.decode (coder.py:459-491) on n_tensors whole tensors of tensor_dims dims lying back to back in p_loc /
 * p_scale / out_sample, cut the way Coder.split cuts them (coder.py:69-83): block j of a tensor = shuffled positions
 * [j * block_size, min((j + 1) * block_size, tensor_dims)) through perm [tensor_dims] (NULL: no shuffle).  One workgroup
 * decodes all blocks of a tensor with sigma_p / mu_p / the samples staged in LDS, so split and merge touch global memory
 * in natural order only (the element-wise gathers of irec_beam_decode cost three times the arithmetic).
 *   K [n_tensors * bpt], indices [n_tensors * bpt, max_K], bpt = ceil(tensor_dims / block_size): row of block j of tensor i
 *   is block_row[i * bpt + j] (device int32; NULL: i * bpt + j).  p->table_dims is ignored (derived from the sizes).
 * Same outputs as irec_beam_decode, bit for bit.  Applies when irec_decode_tensors_supported() (tensors that fit the LDS);
 * workspace as for irec_beam_decode_ws (irec_decode_workspace_bytes with table_dims = {block_size, last block's dim}). */
int32_t irec_decode_tensors_supported(const irec_params *p, int32_t tensor_dims, int32_t block_size);
irec_status irec_beam_decode_tensors(irec_context *ctx, const irec_params *p, int64_t n_tensors, int32_t tensor_dims,
                                     int32_t block_size, const int32_t *block_row, const int32_t *perm, const float *p_loc,
                                     const float *p_scale, int64_t seed, int32_t max_K, const int32_t *K,
                                     const int32_t *indices, float *out_sample, void *workspace, size_t workspace_bytes,
                                     void *hip_stream);

/* ---- .rec wire format: entropy coder of the index streams (host memory; the reference's is CPU Cython too) ------------ */
const char *irec_io_last_error(void);
/* ArithmeticCoder(counts, precision).encode(message) -- rec/io/entropy_coding.pyx:51-121.  out_bits: one ASCII '0'/'1'
 * per code bit; *n_bits = code length (also when it exceeds cap, in which case an error is returned). */
irec_status irec_ac_encode(const int64_t *counts, int32_t n_symbols, const int64_t *message, int64_t n_message,
                           int32_t precision, uint8_t *out_bits, int64_t cap, int64_t *n_bits);
/* ArithmeticCoder.decode_fast(code) -- rec/io/entropy_coding.pyx:213-302 (symbol lookup of data_structures.py:186-213
 * by binary search).  Decodes up to and including the terminator symbol 0. */
irec_status irec_ac_decode(const int64_t *counts, int32_t n_symbols, const uint8_t *bits, int64_t n_bits,
                           int32_t precision, int64_t *out_message, int64_t cap, int64_t *n_message);
/* int('1' + code, 2).to_bytes(ceil((len + 1) / 8), 'big') -- rec/io/utils.py:66-72,100-106.  Returns bytes written or -1. */
int64_t irec_rec_pack_bits(const uint8_t *bits, int64_t n_bits, uint8_t *out_bytes, int64_t cap);
/* bin(int.from_bytes(bytes, 'big'))[3:] -- rec/io/utils.py:158-170.  Returns code bits written or -1. */
int64_t irec_rec_unpack_bits(const uint8_t *bytes, int64_t n_bytes, uint8_t *out_bits, int64_t cap);

/* Whole .rec files with the default symbol models, one call each (the per-stream Python glue of the reference, utils.py:55-106 and
 * :150-216, costs more host time per image than the GPU takes to code it).
 * write_compressed_code(file_path, seed, image_shape, block_size, block_indices, max_index) -- rec/io/utils.py:7-106:
 *   blocks_per_res [R]; K [sum blocks_per_res] partitions of every coded block, residual block after residual block;
 *   indices [sum K] their sample indices back to back.  Returns the file's byte count (also when > cap: call again), -1 on error. */
int64_t irec_rec_encode_file(uint32_t seed, uint32_t block_size, uint32_t max_index, uint32_t height, uint32_t width,
                             uint32_t channels, int32_t n_res_blocks, const int32_t *blocks_per_res, const int32_t *K,
                             const int32_t *indices, uint8_t *out, int64_t cap);
/* read_compressed_code -- rec/io/utils.py:109-216.  header_out[9] = seed, block_size, max_index, height, width, channels,
 * uses_count_file, uses_index_file, R; sizes_out[3] = R, coded blocks, indices.  IREC_E_WORKSPACE (sizes filled) when an
 * output array is null or short. */
irec_status irec_rec_decode_file(const uint8_t *bytes, int64_t n_bytes, uint32_t *header_out, int64_t *sizes_out,
                                 int32_t *blocks_per_res, int64_t cap_res, int32_t *K, int64_t cap_blocks, int32_t *indices,
                                 int64_t cap_indices);

/* Many containers at once (host threads; n_threads = 0: one per core, at most 32).  A batched model pass hands over ONE packed
 * read-back for all its images -- K [n_images][R][bpt], idx [n_images][R][bpt][max_K] (first K entries of a row count), R
 * residual blocks of bpt coded blocks each -- instead of nested per-image lists (the per-image loop of
 * compression_performance.py:350-375: write_compressed_code, then read_compressed_code and compare).
 * irec_rec_encode_files: the n_images files back to back in out, offsets [n_images + 1]; every file byte for byte what
 * irec_rec_encode_file gives for that image.  Returns the total byte count (also when > cap: call again), -1 on error.
 * irec_rec_decode_files: the inverse, rows zero-filled past K; headers [n_images][9] as irec_rec_decode_file's header_out. */
int64_t irec_rec_encode_files(uint32_t seed, uint32_t block_size, uint32_t max_index, uint32_t height, uint32_t width,
                              uint32_t channels, int32_t n_images, int32_t n_res_blocks, int32_t blocks_per_res, int32_t max_K,
                              const int32_t *K, const int32_t *idx, uint8_t *out, int64_t cap, int64_t *offsets, int32_t n_threads);
irec_status irec_rec_decode_files(const uint8_t *bytes, const int64_t *offsets, int32_t n_images, int32_t n_res_blocks,
                                  int32_t blocks_per_res, int32_t max_K, uint32_t *headers, int32_t *K, int32_t *idx,
                                  int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* IREC_H_ */
