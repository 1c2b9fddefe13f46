/*
 * irec_internal.h -- what libirec_hip.so exports BESIDE the drop-in boundary of include/irec.h: diagnostic flags of irec_params.flags (A/B
 * builds, pinned kernel shapes, test hooks), unit-test entry points of single device functions, and the elementwise hand-offs of the RVAE
 * model shim.  Nothing a caller of the coder needs: a maintainer of the reference binds include/irec.h alone.  The
 * tests, bench.py's diagnostics and scripts/ use these through irec/_lib.py.
 */
#ifndef IREC_INTERNAL_H_
#define IREC_INTERNAL_H_

#include "irec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostic / test bits of irec_params.flags (results never depend on them) ---------------------------------------------- */
#define IREC_FLAG_FUSED_PHILOX 2  /* keep the Philox draw fused in the block kernel even when table_dims is set */
#define IREC_FLAG_ONE_TABLE 4     /* with table_dims: always the one-table-copy encoder (one workgroup per block)      */
#define IREC_FLAG_TEAM 8          /* with table_dims: always the teams-per-CU encoder over three table copies.         */
                                  /* Neither flag: the team encoder for calls of >= 64 blocks, the one-table encoder   */
                                  /* (cheaper per-call set-up) below; same outputs, bit for bit                        */
#define IREC_FLAG_TEST_SPLIT_ORPHAN 32 /* test hook: the partner workgroups of the split encoder leave at once, so workgroup 0  */
                                  /* of every block must take the give-up exit (100 ms) (out_K = -2) instead of hanging   */
#define IREC_FLAG_LISTED_ORDER 262144 /* team encoder, calls of one to a few rows per CU: deal the rows to the CUs in the order listed.    */
                                  /* Default: by cost -- the call's preparation kernel also computes K * dims of every row and a CU's first   */
                                  /* team takes a cheap row, its other teams (and the teams that share a row) the costliest ones, so   */
                                  /* that the longest rows do not meet on one CU.  Results do not depend on it (diagnostics, A/B).     */
#define IREC_FLAG_NO_TEN 1048576   /* diagnostics (A/B, tests): plain calls of at most ten beams and S * 10 <= 256 stay on encode_team_kernel<10,..> instead */
                                  /* of encode_ten_kernel (irec_ten.hip).  Results do not depend on it.                                  */
#define IREC_FLAG_SPLIT_SHIFT 12  /* bits 12-15: workgroups per block of the split encoder / sample stripes per chunk of a gang, at most; */
                                  /* 0 = chosen by the library (diagnostics) */
#define IREC_FLAG_SPLIT_MASK (0xF << IREC_FLAG_SPLIT_SHIFT)
/* Diagnostic workgroup shapes of the team encoder for B <= 20 (bits 8-11 of flags; 0 = the default shape).  Same outputs.
 * (Values 1 and 4 -- one 4-wave team per CU, two 8-wave striped teams -- lost at every size and were removed in round 6 with their builds.) */
#define IREC_FLAG_SHAPE_SHIFT 8
#define IREC_FLAG_SHAPE_MASK (0xF << IREC_FLAG_SHAPE_SHIFT)
#define IREC_FLAG_SHAPE_2 (2 << IREC_FLAG_SHAPE_SHIFT)   /* exactly two teams (also where three are the default)      */
#define IREC_FLAG_SHAPE_3 (3 << IREC_FLAG_SHAPE_SHIFT)   /* three 4-wave teams (168 VGPRs)                            */
#define IREC_FLAG_SHAPE_1X2 (5 << IREC_FLAG_SHAPE_SHIFT) /* one 8-wave beam-striped team (the default of 64..n_CU blocks) */
#define IREC_FLAG_SHAPE_TEAM (6 << IREC_FLAG_SHAPE_SHIFT) /* the team encoder's default shape also for one-beam calls (which the one-wave-per-block encoder takes otherwise) */


/* ---- test hooks (host memory) -------------------------------------------------------------------------------------------- */
/* The launch of irec_beam_encode(p, n_blocks, max_block_dim, max_K) on a device of n_cu compute units, in numbers -- the same code the
 * launch runs (irec_host.cpp: call_detail), no device touched: the planner's invariants are tested host-only at other CU counts than the
 * one box everything was measured on (MI355X partition modes expose 32 / 64 / 128 CUs).
 *   kind 1 chunked, 2 one-beam, 3 team (encode_team_kernel / encode_ten_kernel), 4 one-table / split, 5 fused-Philox fast, 6 generic */
typedef struct {
  int32_t kind, grid, teams_per_wg;   /* workgroups; teams (= scratch slabs) per workgroup: slab index < grid * teams_per_wg           */
  int32_t coop_width;                 /* teams per shared row / workgroups per block / members per gang; 0 = nothing is shared        */
  int32_t coop_beams, gang_chunks;    /* split encoder shares beams (not samples); chunk owners of a gang                              */
  int32_t placed;                     /* team encoder: rows dealt by cost -- the kernel requires the static round to deal EVERY slot  */
  int32_t split_blocks;               /* blocks whose exchange granules the preparation kernel zeroes                                 */
  int64_t share_first, n_slots;       /* first shared row; hand-out slots (whole rows + coop_width per shared row / block)             */
  int64_t slabs_in_workspace, slab_bytes, fixed_bytes;   /* what irec_encode_workspace_bytes sizes: fixed + slabs * slab_bytes        */
  int32_t exchange_rows, exchange_keys;                  /* key exchange: shared blocks per call, sort keys per step, at most          */
} irec_plan_detail;
irec_status irec_test_plan(int32_t n_cu, int32_t clock_mhz, const irec_params *p, int64_t n_blocks, int32_t max_block_dim, int32_t max_K,
                           irec_plan_info *info, irec_plan_detail *detail);
/* out[e] = element e of tf.random.normal([count]) after tf.random.set_seed(seed) -- the stream behind
 * tfd.Normal.sample.  Host memory; test hook. */
irec_status irec_tf_random_normal(int64_t seed, int64_t count, float *out);
/* out[e] = element e of tf.random.stateless_normal([count], seed=[seed0, seed1]) -- the draw inside
 * stateless_gumbel_sample (rec/coding/utils.py:9-12).  Host memory; test hook. */
irec_status irec_tf_stateless_normal(int64_t seed0, int64_t seed1, int64_t count, float *out);

/* ---- hand-offs of the RVAE model shim (device pointers, asynchronous; rec/models/resnet_vae.py:372-497) -------------------------
 * What lies between the convolutions of BidirectionalResidualBlock.call on the compression path, one launch each instead of
 * ~10 elementwise PyTorch launches per residual block and pass.  Activations NCHW float32 contiguous; statistics / latent NHWC.
 * irec_shim_stats: out [n_stats][n][hw][stochastic] = prior loc, exp(prior log-scale) (:409-413) and, for n_stats = 4, posterior
 *   loc = generative + inference side, exp(posterior log-scale) (:148-154, :464-469) from y [n][channels_y][hw] (channels
 *   0 .. n_stats * stochastic) and the inference pass's heads infer_heads [n][channels_infer][hw] (channels 0 .. 2 * stochastic).
 * irec_shim_cat_elu: out [n][deterministic + stochastic][hw] = elu(concat(y[:, channel_offset : + deterministic], latent NHWC))
 *   (:479-488); stochastic = 0: the ELU of a channel slice (:398-400).
 * irec_shim_residual_elu: out = input + alpha * tensor (:492-496), out_elu = elu(out) (the next block's first op, :385);
 *   tensors [n][channels][hw].
 * bias_* (device, per channel; may be NULL): the bias of the convolution that produced the operand, added first -- the
 *   convolution is then called without one, which saves its separate bias-add launch. */
irec_status irec_shim_stats(irec_context *ctx, const float *y, const float *infer_heads, float *out, int32_t n_stats, int32_t n,
                            int32_t channels_y, int32_t channels_infer, int32_t stochastic, int32_t hw, const float *bias_y,
                            const float *bias_infer, void *hip_stream);
irec_status irec_shim_cat_elu(irec_context *ctx, const float *y, const float *latent, float *out, int32_t n, int32_t channels_y,
                              int32_t channel_offset, int32_t deterministic, int32_t stochastic, int32_t hw, const float *bias_y,
                              void *hip_stream);
irec_status irec_shim_residual_elu(irec_context *ctx, const float *input, const float *tensor, float alpha, float *out, float *out_elu,
                                   int32_t n, int32_t channels, int32_t hw, const float *bias_tensor, void *hip_stream);

/* ---- test hooks (device pointers) ---------------------------------------------------------------------------- */
/* r[s*D + d] of get_pseudo_random_sample's int32 draw, generated by the in-kernel Philox stream.  out: int32 [n]. */
irec_status irec_device_uniform_int(irec_context *ctx, int64_t seed, int64_t n, int32_t *out, void *hip_stream);
/* The decoder's short correctly-rounded square root against sqrtf on every float32 bit pattern it is allowed to see (all
 * but the non-zero values below 2^-96, which take sqrtf itself): out2[0] = mismatches, out2[1] = patterns compared.
 * out2: device uint64 [2]. */
irec_status irec_test_decoder_sqrt(irec_context *ctx, uint64_t *out2, void *hip_stream);
/* in: float [64 lanes][width]; out: float [128].  width in {64, 32}: out[lane] = sum over lanes of
 * in[.][lane*width/64] in the canonical 64-lane reduction tree of the score kernels.  width in {20, 10}
 * (the arbitrary-width reduce-scatter of the team encoder): out[lane] = such a total of column out[64 + lane] (a column
 * index as a float, or -1 if the lane ends up with an unused slot); every column is owned by two lanes. */
irec_status irec_test_reduce_scatter(irec_context *ctx, const float *in, float *out, int32_t width, void *hip_stream);
/* tf.argsort(scores, DESCENDING)[:n_select] split into (index // n_beams_cur, index % n_beams_cur) -- the top-B step of
 * beam_search_coder.py:85-89 in isolation.  scores: float [n]; scratch_keys: uint32 [n]; out_sel: int32 [n_select][2]. */
irec_status irec_test_select(irec_context *ctx, const float *scores, int32_t n, int32_t n_select, int32_t n_beams_cur,
                             uint32_t *scratch_keys, int32_t *out_sel, void *hip_stream);
/* The same step in the form the one-table / split encoders and the two-team builds of the team encoder run since round 4 (threshold by
 * probing the lane counts, ranks by the key alone with a collision check for ties): same outputs. */
irec_status irec_test_select_quick(irec_context *ctx, const float *scores, int32_t n, int32_t n_select, int32_t n_beams_cur,
                                   uint32_t *scratch_keys, int32_t *out_sel, void *hip_stream);
/* The per-call proposal table of the default encoder for blocks of `dim` dims: out_tab uint16 [n_steps][n_samples][dim
 * rounded up to 4] = dlog_g(r) + 10006 * c, r the int32 draw of beam_search_coder.py:38-43 at seed + t, c the copy bit
 * that spreads each 32-lane look-up group over the LDS banks. */
irec_status irec_test_proposal_table(irec_context *ctx, int64_t seed, int32_t n_samples, int32_t dim, int32_t n_steps,
                                     uint16_t *out_tab, void *hip_stream);
/* device addresses of the context's constant tables (lut [10007], lut2 [10006], dlog4r [10006] u16, rho [65536]). */
irec_status irec_device_tables(irec_context *ctx, const float **lut, const float **lut2, const uint16_t **dlog4r,
                               const float **rho);

#ifdef __cplusplus
}
#endif
#endif /* IREC_INTERNAL_H_ */
