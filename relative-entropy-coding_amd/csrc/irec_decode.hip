// irec_decode.hip -- gfx950 decoder of the iREC beam-search code.
//
// Hot path (reference file:line): BeamSearchCoder.decode_block rec/coding/beam_search_coder.py:124-148, driven the way
// GaussianCoder.decode drives it (rec/coding/coder.py:459-491: split the prior through the shuffle, decode every block with
// the SAME seed, merge).  Per step t the reference regenerates all S rows of the draw and keeps row indices[t]
// (:141-146); here only that row is touched.
//
// decode_wave_kernel (round 3; decode_kernel below is the round-2 form, kept for calls that give no dim hints):
//   * work unit = ONE WAVE x 256 dims of a block (lane = 4 consecutive dims).  A 192-dim tail block is one unit of 48
//     lanes, not a 256-thread workgroup with 208 idle lanes; the waves of a workgroup take units independently (no
//     barrier after the table fill).
//   * the index path of a block is loaded ONCE, 64 steps per lane-vector: lane t holds indices[t], the hash of the prefix
//     (simple_hash, :33-35: a wave prefix sum of idx[j] * (69 + j)), its discrete log and the power-law ratio of step t;
//     the step loop reads them with v_readlane -- no dependent global load and no integer division inside the loop.
//   * TABLE: the int32 draw of get_pseudo_random_sample (:38-43) depends only on (seed + t, s, d) -- every block of the
//     call with the same dim count draws from the same S rows per step -- so the call evaluates the Philox stream once
//     into tab[t][s][d] = 4 * dlog_g(r) (alpha_table_kernel, the one-table encoder's proposal table) and a unit streams
//     row indices[t]: 8 coalesced bytes per lane and step instead of a Philox block and four `% 10006`.  The rows of
//     several steps are in flight at once (they depend on nothing but the index path).  A block with more partitions
//     than the table window, and every call without a table, draws in the kernel (Philox fused, as before).
//   * quantile look-up in discrete-log order from ONE 40 KB LDS copy: z = lut2[(dlog r + dlog hash) mod 10006], one add and
//     one unsigned min instead of a multiply and a `% 10007`.
//   * what remains per (dim, step) is the chain the reference runs too (:131-139): a = rho (var_p - c), sa = sqrt(a) (IEEE,
//     correctly rounded: part of the bit-exact contract), sample += sa * z, c += a -- about 20 VALU lane-operations, which
//     is what bounds the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "irec_device.h"
#include "irec_kernels.h"
#include "irec_fast_common.h"

namespace irec {

constexpr int DEC_NW = 8;                 // waves per workgroup
constexpr int DEC_NT = DEC_NW * 64;
constexpr size_t DEC_LUT_BYTES = 40032;   // lut2 [10006] f32, padded to 16
constexpr size_t DEC_DLOG_BYTES = 20016;  // dlog4r [10006] u16 (kernels without a table)

// One unit = one wave x 256 dims of a block: runs the K steps of decode_block for the lane's four dims.
//   idx: the block's index row; var_p: sigma_p^2 of my dims; row_off: my quad inside a row (a real quad also for lanes past
//   the block's end); tab: the block's proposal table or nullptr (draw in the kernel); dlog_f: dlog table of the fused draw.
// Correctly rounded float32 sqrt for the decoder's chain.  hipcc's own expansion of sqrtf (IEEE mode, denormals on) is
//   scale tiny inputs by 2^32 | v_sqrt_f32 (1 ulp) | try the neighbours s - 1ulp / s + 1ulp: keep s - 1ulp if fma(-(s - 1ulp), s, x)
//   <= 0, take s + 1ulp if fma(-(s + 1ulp), s, x) > 0 | unscale | pass +-0 and +inf through (v_cmp_class)
// -- 16 instructions, four of them per dim and step: the largest single item of the kernel.  The middle part alone (9
// instructions) returns the same bits for every input that is not scaled (x >= 2^-96) and also for +-0, +inf and NaN
// (checked for all 2^32 float32 bit patterns against sqrtf on the device: tests/test_gpu_parity.py::test_decoder_fast_sqrt_exhaustive);
// A unit runs on the core and keeps the smallest input it met; if any lane met one the core may not see, the wave runs the
// unit again on sqrtf (decode_unit<TABLE, false>): no test inside the step loop.
#ifndef IREC_DEC_FAST_SQRT
#define IREC_DEC_FAST_SQRT 1
#endif
__device__ __forceinline__ float dec_sqrt_core(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
  float r = rd <= 0.0f ? sd : s;
  r = ru > 0.0f ? su : r;
  return r;
}
__device__ __forceinline__ bool dec_sqrt_core_ok(float x) {   // inputs the core handles as sqrtf does
  return !(x < 0x1.0p-96f) || x == 0.0f;                       // (NaN compares false: allowed; negatives: sqrtf's NaN either way, but keep them on the slow path)
}
// `bad` (wave-uniform on return): an index of the row is not a sample index (>= S: a damaged or foreign stream).  The reference
// would fail in tf.gather (beam_search_coder.py:141-146); here the block is "not decodable" like a row with K out of range --
// and the proposal-table row it would have addressed is never read (the fused draw is memory-safe for any index).
template <bool TABLE, bool FAST>
__device__ __forceinline__ float decode_unit_run(const DecArgs &A, const int32_t *idx, int K, int D, int Dp, const uint16_t *tab,
                                                const uint16_t *dlog_f, uint32_t row_off, int lane, const float (&var_p)[4],
                                                float (&sample)[4], bool &bad) {
  float c[4];
  bad = false;
  float amin = 1.0f;                                      // smallest auxiliary variance seen (FAST: what the short sqrt was fed)
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i] = 0.f; sample[i] = 0.f; }
  uint32_t hs_base = 0u;                                  // int32 wrap-around sum of simple_hash over the steps before t0
  for (int t0 = 0; t0 < K; t0 += 64) {
    const int nt = K - t0 < 64 ? K - t0 : 64;
    // lane j: step t0 + j
    const uint32_t idxv = lane < nt ? (uint32_t)idx[t0 + lane] : 0u;
    if (__ballot(idxv >= (uint32_t)A.S)) { bad = true; return amin; }   // (wave-uniform; before any row of this chunk is addressed)
    uint32_t incl = idxv * (uint32_t)(69 + t0 + lane);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
      if (lane >= d) incl += o;
    }
    const uint32_t hsum = hs_base + incl - idxv * (uint32_t)(69 + t0 + lane);   // hash sum of the path BEFORE step t0 + lane
    const uint32_t b4v = lane < nt ? (uint32_t)A.dlog4r[hash_from_sum((int32_t)hsum) - 1u] : 0u;
    const float rhov = lane < nt ? A.rho[K - 1 - (t0 + lane)] : 0.f;
    hs_base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    auto step = [&](const uint32_t (&al)[4], uint32_t b4, float rho) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t ad = al[i] + b4;                           // 4 * (dlog r + dlog hash)
        const uint32_t ad2 = ad - IREC_LUT2_BYTES;
        ad = ad2 < ad ? ad2 : ad;                           // mod 10006
        const float z = lds_abs_f32(ad);                    // ndtri(((r * hash) mod 10007) / 10007), :45-49
        const float a = rho * (var_p[i] - c[i]);            // auxiliary variance, :131-134
        float sa;
        if constexpr (FAST) { sa = dec_sqrt_core(a); amin = fminf(amin, a == 0.0f ? 1.0f : a); }
        else sa = sqrtf(a);
        sample[i] = sample[i] + sa * z;                     // :146
        c[i] = c[i] + a;                                    // :147
      }
    };
    if (TABLE && tab != nullptr) {
      // rows of DEPTH steps in flight: they depend on the index path only
      constexpr int DEPTH = 4;
      uint2 rows[DEPTH];
      auto fetch = [&](int j) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)idxv, j < nt ? j : nt - 1);
        return *reinterpret_cast<const uint2 *>(tab + (((size_t)(t0 + (j < nt ? j : nt - 1)) * A.S + it) * Dp + row_off));
      };
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) rows[j] = fetch(j);
      for (int j0 = 0; j0 < nt; j0 += DEPTH) {
#pragma unroll
        for (int jj = 0; jj < DEPTH; ++jj) {
          const int j = j0 + jj;
          if (j < nt) {                                     // wave-uniform
            const uint2 r = rows[jj];
            rows[jj] = fetch(j + DEPTH);
            const uint32_t al[4] = {r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16};
            step(al, (uint32_t)__builtin_amdgcn_readlane((int)b4v, j), __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rhov), j)));
          }
        }
      }
    } else {
      for (int j = 0; j < nt; ++j) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)idxv, j);
        const StepSeed ss = make_step_seed(A.seed + t0 + j);
        uint32_t rm1[4];
        draw_rm1_x4(ss, (uint64_t)it * (uint64_t)D + (uint64_t)row_off, rm1);   // (it * D + d0) & 3 is wave-uniform: d0 % 4 == 0
        const uint32_t al[4] = {dlog_f[rm1[0]], dlog_f[rm1[1]], dlog_f[rm1[2]], dlog_f[rm1[3]]};
        step(al, (uint32_t)__builtin_amdgcn_readlane((int)b4v, j), __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rhov), j)));
      }
    }
  }
  return amin;
}
template <bool TABLE>
__device__ __forceinline__ void decode_unit(const DecArgs &A, const int32_t *idx, int K, int D, int Dp, const uint16_t *tab,
                                            const uint16_t *dlog_f, uint32_t row_off, int lane, const float (&var_p)[4],
                                            float (&sample)[4]) {
  bool bad = false;
  auto undecodable = [&]() {          // the block's elements come out as mu_p, as for a row with K out of range
#pragma unroll
    for (int i = 0; i < 4; ++i) sample[i] = 0.f;
  };
  if constexpr (IREC_DEC_FAST_SQRT != 0) {
    const float amin = decode_unit_run<TABLE, true>(A, idx, K, D, Dp, tab, dlog_f, row_off, lane, var_p, sample, bad);
    if (bad) { undecodable(); return; }
    if (__builtin_expect(__ballot(!dec_sqrt_core_ok(amin)) == 0ull, 1)) return;   // wave-uniform
  }
  decode_unit_run<TABLE, false>(A, idx, K, D, Dp, tab, dlog_f, row_off, lane, var_p, sample, bad);
  if (bad) undecodable();
}

template <bool TABLE>
__global__ __launch_bounds__(DEC_NT) __attribute__((amdgpu_waves_per_eu(6, 8))) void decode_wave_kernel(DecArgs A) {   // <= 80 VGPRs: three workgroups per CU
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  {
    float *l2 = reinterpret_cast<float *>(smem);
    for (int k = tid; k < (int)IREC_PM1; k += DEC_NT) l2[k] = A.lut2[k];
    if (!TABLE) {
      uint16_t *dl = reinterpret_cast<uint16_t *>(smem + DEC_LUT_BYTES);
      for (int k = tid; k < (int)IREC_PM1; k += DEC_NT) dl[k] = A.dlog4r[k];
    }
  }
  __syncthreads();   // the only barrier: from here on every wave works on its own
  // rows of the fused draw: dlog from the LDS copy (kernels without a table) or through the L1 (the rare block beyond the
  // table window of a table kernel)
  const uint16_t *dlog_f = TABLE ? A.dlog4r : reinterpret_cast<const uint16_t *>(smem + DEC_LUT_BYTES);
  const int upb = A.upb;
  const int64_t n_units = A.n_blocks * (int64_t)upb;
  const int64_t n_waves = (int64_t)gridDim.x * DEC_NW;
  for (int64_t u = (int64_t)blockIdx.x * DEC_NW + wave; u < n_units; u += n_waves) {
    const int64_t blk = u / upb;
    const int chunk = (int)(u - blk * upb);
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    if (D > upb * 256) {                                    // beyond the host's bound (a call without max_block_dim whose dim hints
      if (chunk == 0)                                       // do not list this block): not decodable -- mu_p, as every such block
        for (int d = lane; d < D; d += 64) {
          const int64_t ixo = base + (A.perm ? A.perm[pos + d] : pos + d);
          A.out_sample[ixo] = 0.f + A.p_loc[ixo];
        }
      continue;
    }
    if (chunk * 256 >= D) continue;                         // (a short block has fewer units than upb)
    const int K = A.K[blk];
    const int32_t *idx = A.indices + blk * (int64_t)A.max_K;
    const int d0 = chunk * 256 + lane * 4;
    const bool live = d0 < D;
    if (K > A.max_K || K > A.K_limit || K < 0) {                               // not decodable: its elements come out as mu_p (as the staged decoder's)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (d0 + i < D) {
          const int e = pos + d0 + i;
          const int64_t ixo = base + (A.perm ? A.perm[e] : e);
          A.out_sample[ixo] = 0.f + A.p_loc[ixo];
        }
      continue;
    }
    // my 4 dims: element offsets inside the tensor (Coder.split: gather through the shuffle, coder.py:62-83)
    int32_t off[4];
    float var_p[4], sample[4];
    bool valid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      valid[i] = d0 + i < D;
      const int e = pos + (valid[i] ? d0 + i : 0);
      off[i] = A.perm ? A.perm[e] : e;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sp = A.p_scale[base + off[i]];
      var_p[i] = sp * sp;
    }
    const uint16_t *tab = nullptr;
    const int Dp = (D + 3) & ~3;
    if (TABLE && K <= A.K_tab) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (A.tab_dim[q] == D) tab = A.tab[q];
    }
    decode_unit<TABLE>(A, idx, K, D, Dp, tab, dlog_f, (uint32_t)(live ? d0 : 0), lane, var_p, sample);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (valid[i]) A.out_sample[base + off[i]] = sample[i] + A.p_loc[base + off[i]];   // + coding_dist.loc (:148); merge = scatter
  }
}

// ======================================================================================================
//  decode_tensor_kernel: GaussianCoder.decode (coder.py:459-491) on whole tensors, split and merge through the LDS.
//
//  Coder.split / merge read and write every tensor through tf.random.shuffle's permutation (coder.py:62-83,111-117): taken
//  element by element from global memory that is one 128-byte line per 4-byte value (measured, r03a: 1.4 of the 1.9 ms of a
//  73 728-block decode; the same arithmetic on unshuffled blocks takes 0.5 ms).  Here ONE workgroup decodes ALL blocks of a
//  tensor: sigma_p is read coalesced into an LDS region, the waves gather their dims' variances from there and run their
//  units (one wave x 256 dims of a block, as above), pulled from an LDS counter, and write their samples IN PLACE over the
//  variances -- split and merge address the same elements, each owned by one unit (r03h; until then: fixed rounds, samples
//  waiting in registers, a zero fill and a scatter pass behind a second barrier) -- and the region leaves coalesced with
//  mu_p -- read coalesced -- added on the way out.  The global reads of a pass are issued a phase ahead into registers
//  (sigma_p a whole tensor ahead), two workgroup barriers per tensor, two workgroups per CU (72 KB of LDS each at 8192 dims).
//  Blocks are Coder.split's: block j = shuffled positions [j * tbs, min((j + 1) * tbs, tn)) of tensor `tensor`; its K /
//  index row is block_row[tensor * bpt + j] (or tensor * bpt + j).
// ======================================================================================================
#ifndef IREC_DEC_ABLATE
#define IREC_DEC_ABLATE 0        // diagnostics (make variant_dec): 1 = no staging copies, 2 = no unit arithmetic, 4 = no perm look-ups
#endif
// Shape (r03a sweep on 8192-dim tensors in 1000-dim blocks, 36 unit slots, 8192 tensors per call): 12 waves x 3 rounds at
// 80 VGPRs -- two workgroups = 24 waves per CU -- 0.746 ms; 9 x 4 at 96 VGPRs 1.04 ms (nine waves load the four SIMDs
// 3-2-2-2); 8 x 5 0.76-0.90 ms; one 12-wave workgroup per CU at 128 VGPRs, no spill at all, 0.82 ms.  More rounds cost
// registers (four per round across the unit loop): the five-round build serves tensors of up to 60 unit slots.
#ifndef IREC_DEC_WMAX
#define IREC_DEC_WMAX 12         // waves per workgroup at most
#endif
#ifndef IREC_DEC_DYN
#define IREC_DEC_DYN 1           // 1: the waves pull a tensor's units from an LDS counter and write their samples IN PLACE over the
                                 //    variances they gathered (every element has one owner); 0: r03a's fixed rounds (A/B builds)
#endif
#ifndef IREC_DEC_WPE
#define IREC_DEC_WPE 6           // waves per SIMD the tensor kernel is compiled for (80 VGPRs)
#endif
constexpr int DEC_RMAX_SMALL = 3, DEC_RMAX_BIG = 5;   // units per wave and tensor of the two builds
template <bool TABLE, int DEC_RMAX>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(IREC_DEC_WPE, 8))) void decode_tensor_kernel(DecArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, NT = (int)blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = NT >> 6;
  (void)wave; (void)NW;
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap(); // see lds_abs_f32
  float *region = reinterpret_cast<float *>(smem + DEC_LUT_BYTES + (TABLE ? 0 : DEC_DLOG_BYTES));   // [tn]
  {
    float *l2 = reinterpret_cast<float *>(smem);
    for (int k = tid; k < (int)IREC_PM1; k += NT) l2[k] = A.lut2[k];
    if (!TABLE) {
      uint16_t *dl = reinterpret_cast<uint16_t *>(smem + DEC_LUT_BYTES);
      for (int k = tid; k < (int)IREC_PM1; k += NT) dl[k] = A.dlog4r[k];
    }
  }
  const uint16_t *dlog_f = TABLE ? A.dlog4r : reinterpret_cast<const uint16_t *>(smem + DEC_LUT_BYTES);
  const int n = A.tn, bs = A.tbs, bpt = A.tbpt;
  uint32_t *unit_next = reinterpret_cast<uint32_t *>(region + (((size_t)n + 3) & ~(size_t)3));   // 16 bytes behind the region
  const int upb = (bs + 255) >> 8;                          // units of a full block
  const int upt = bpt * upb;                                // unit slots per tensor (the short last block leaves some empty)
  // A thread's share of a tensor for the staging passes: PF = 4 * RMAX floats (RMAX 16-byte pieces when the tensor allows).
  // Enough for every tensor the host sends here: its unit slots cover it (upt * 256 >= n) and NW * RMAX >= upt, so
  // NT * 4 * RMAX >= n.
  constexpr int PF = 4 * DEC_RMAX;
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(A.p_scale) | reinterpret_cast<uintptr_t>(A.p_loc) |
                                      reinterpret_cast<uintptr_t>(A.out_sample)) & 15) == 0;
  auto fetch16 = [&](const float *src, float (&v)[PF]) {
    if (vec) {
      const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
      for (int k = 0; k < DEC_RMAX; ++k) {
        const int i = k * NT + tid;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * i < n) q = s4[i];
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PF; ++k) { const int i = k * NT + tid; v[k] = i < n ? src[i] : 0.f; }
    }
  };
  float psn[PF];                                            // sigma_p of the NEXT tensor, fetched a tensor ahead
  if ((int64_t)blockIdx.x < A.n_tensors && !(IREC_DEC_ABLATE & 1)) fetch16(A.p_scale + (int64_t)blockIdx.x * n, psn);
  for (int64_t tensor = blockIdx.x; tensor < A.n_tensors; tensor += gridDim.x) {
    const int64_t base = tensor * (int64_t)n;
    // (fixed rounds: the previous tensor's store pass must be over before the region is refilled.  Pulled units: a thread
    //  refills exactly the elements it has just stored out -- the same i = k * NT + tid in both passes --, no barrier needed;
    //  the table fill is covered by the barrier behind the refill)
    if (IREC_DEC_DYN == 0) __syncthreads();
    if (!(IREC_DEC_ABLATE & 1)) {
      if (vec) {
#pragma unroll
        for (int k = 0; k < DEC_RMAX; ++k) {
          const int i = k * NT + tid;
          if (4 * i < n) reinterpret_cast<float4 *>(region)[i] = make_float4(psn[4 * k], psn[4 * k + 1], psn[4 * k + 2], psn[4 * k + 3]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int i = k * NT + tid; if (i < n) region[i] = psn[k]; }
      }
    }
    if (IREC_DEC_DYN != 0 && tid == 0) *unit_next = 0u;
    __syncthreads();
#if IREC_DEC_DYN
    // Units of this tensor, pulled from the counter (blocks differ in K, a tensor's 33 units do not divide by 12 waves: fixed
    // rounds left every wave waiting for the slowest at the barrier below).  A unit gathers its variances from the region and
    // writes its samples over them: split and merge address the SAME elements (coder.py:62-83,111-117), each owned by one
    // unit, so nobody else reads or writes them between the two barriers -- no second barrier, no zero fill, no samples
    // waiting in registers.  A block that cannot be decoded leaves zeros.  (Handing the blocks out longest first, K and row
    // of a tensor's blocks fetched a tensor ahead into lanes: no further gain, profiles/archive/r03h/ab_dec_lpt.log.)
    for (;;) {
      uint32_t u_ = 0u;
      if (lane == 0) u_ = __hip_atomic_fetch_add(unit_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int u = __builtin_amdgcn_readfirstlane((int)u_);
      if (u >= upt) break;                                  // wave-uniform
      const int j = u / upb, chunk = u - j * upb;
      const int pos = j * bs;
      const int D = n - pos < bs ? n - pos : bs;
      if (chunk * 256 >= D) continue;
      const int64_t row = A.block_row ? (int64_t)A.block_row[tensor * bpt + j] : tensor * bpt + j;
      const int K = A.K[row];
      const int32_t *idx = A.indices + row * (int64_t)A.max_K;
      const int d0 = chunk * 256 + lane * 4;
      int at[4];
      float var_p[4], smp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = pos + (d0 + i < D ? d0 + i : 0);
        at[i] = (A.perm && !(IREC_DEC_ABLATE & 4)) ? A.perm[e] : e;
        // (a padding lane owns no element: position `pos` belongs to the block's chunk-0 unit, which may already have written
        //  its sample there -- a zero variance keeps the lane's chain at zero and out of the slow-sqrt test)
        const float sp = d0 + i < D ? region[at[i]] : 0.f;
        var_p[i] = sp * sp;
      }
      if (K >= 0 && K <= A.max_K && K <= A.K_limit) {                         // (else not decodable: its elements come out as mu_p)
        const uint16_t *tab = nullptr;
        const int Dp = (D + 3) & ~3;
        if (TABLE && K <= A.K_tab) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (A.tab_dim[q] == D) tab = A.tab[q];
        }
        if (!(IREC_DEC_ABLATE & 2)) decode_unit<TABLE>(A, idx, K, D, Dp, tab, dlog_f, (uint32_t)(d0 < D ? d0 : 0), lane, var_p, smp);
        else { smp[0] = var_p[0]; smp[1] = var_p[1]; smp[2] = var_p[2]; smp[3] = var_p[3]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (d0 + i < D) region[at[i]] = smp[i];             // merge (coder.py:111-117)
    }
    float pln[PF];
    if (!(IREC_DEC_ABLATE & 1)) {
      fetch16(A.p_loc + base, pln);
      const int64_t nxt = tensor + (int64_t)gridDim.x < A.n_tensors ? tensor + (int64_t)gridDim.x : tensor;
      fetch16(A.p_scale + nxt * n, psn);
    }
    __syncthreads();                                        // every unit has written its samples
#else
    // my units of this tensor, one after the other (a rolled loop: the unit code exists once); their samples wait in registers
    // for the merge -- RMAX x 4 floats, selected by the (scalar) round number
    float keep[DEC_RMAX][4];
#pragma unroll
    for (int r = 0; r < DEC_RMAX; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) keep[r][i] = 0.f;
    uint32_t done = 0u;                                     // bit r: round r decoded a block's unit here
#pragma clang loop unroll(disable)
    for (int r = 0; r < DEC_RMAX; ++r) {
      const int u = wave + r * NW;
      if (u >= upt) break;                                  // wave-uniform
      const int j = u / upb, chunk = u - j * upb;
      const int pos = j * bs;
      const int D = n - pos < bs ? n - pos : bs;
      if (chunk * 256 >= D) continue;
      const int64_t row = A.block_row ? (int64_t)A.block_row[tensor * bpt + j] : tensor * bpt + j;
      const int K = A.K[row];
      if (K < 0 || K > A.max_K || K > A.K_limit) continue;                   // not decodable: its elements come out as mu_p
      const int32_t *idx = A.indices + row * (int64_t)A.max_K;
      const int d0 = chunk * 256 + lane * 4;
      float var_p[4], smp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = pos + (d0 + i < D ? d0 + i : 0);
        const float sp = region[(A.perm && !(IREC_DEC_ABLATE & 4)) ? A.perm[e] : e];
        var_p[i] = sp * sp;
      }
      const uint16_t *tab = nullptr;
      const int Dp = (D + 3) & ~3;
      if (TABLE && K <= A.K_tab) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (A.tab_dim[q] == D) tab = A.tab[q];
      }
      if (!(IREC_DEC_ABLATE & 2)) decode_unit<TABLE>(A, idx, K, D, Dp, tab, dlog_f, (uint32_t)(d0 < D ? d0 : 0), lane, var_p, smp);
      else { smp[0] = var_p[0]; smp[1] = var_p[1]; smp[2] = var_p[2]; smp[3] = var_p[3]; }
      done |= 1u << r;
#pragma unroll
      for (int rr = 0; rr < DEC_RMAX; ++rr)
        if (rr == r) {                                      // scalar compare
#pragma unroll
          for (int i = 0; i < 4; ++i) keep[rr][i] = smp[i];
        }
    }
    // global reads of the passes to come, issued now that the unit loop has released its registers: mu_p of this tensor (added
    // in natural order when the region leaves) and sigma_p of the next one.  Their latency hides behind the barriers and the
    // merge below -- and behind the other workgroup of the CU.
    float pln[PF];
    if (!(IREC_DEC_ABLATE & 1)) {
      fetch16(A.p_loc + base, pln);
      // (unconditional, so that psn is dead across the unit loop: the last round fetches its own tensor again)
      const int64_t nxt = tensor + (int64_t)gridDim.x < A.n_tensors ? tensor + (int64_t)gridDim.x : tensor;
      fetch16(A.p_scale + nxt * n, psn);
    }
    __syncthreads();                                        // everybody has gathered its variances: the region takes the samples
    for (int i = tid; i < n; i += NT) region[i] = 0.f;      // (elements of a block that was not decodable)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DEC_RMAX; ++r) {
      if (done & (1u << r)) {                               // wave-uniform
        const int u = wave + r * NW;
        const int j = u / upb, chunk = u - j * upb;
        const int pos = j * bs;
        const int D = n - pos < bs ? n - pos : bs;
        const int d0 = chunk * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (d0 + i < D) {
            const int e = pos + d0 + i;
            region[(A.perm && !(IREC_DEC_ABLATE & 4)) ? A.perm[e] : e] = keep[r][i];   // merge (coder.py:111-117): every element has one owner
          }
      }
    }
    __syncthreads();
#endif
    if (!(IREC_DEC_ABLATE & 1)) {                           // sample + coding_dist.loc (:148), leaving in natural order
      if (vec) {
#pragma unroll
        for (int k = 0; k < DEC_RMAX; ++k) {
          const int i = k * NT + tid;
          if (4 * i < n) {
            const float4 sv = reinterpret_cast<const float4 *>(region)[i];
            reinterpret_cast<float4 *>(A.out_sample + base)[i] =
                make_float4(sv.x + pln[4 * k], sv.y + pln[4 * k + 1], sv.z + pln[4 * k + 2], sv.w + pln[4 * k + 3]);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int i = k * NT + tid; if (i < n) A.out_sample[base + i] = region[i] + pln[k]; }
      }
    }
  }
}

// ======================================================================================================
//  round-2 decoder: one 256-thread workgroup per block, Philox fused, natural-order table.  Serves calls without dim hints
//  (the unit count per block is not known) and blocks of more than 1024 dims.
// ======================================================================================================
template <bool LDS_LUT>
__global__ __launch_bounds__(256) void decode_kernel(DecArgs A) {
  const int tid = threadIdx.x;
  __shared__ float lut_s[LDS_LUT ? IREC_P : 1];
  if constexpr (LDS_LUT) {
    for (int k = tid; k < (int)IREC_P; k += 256) lut_s[k] = A.lut[k];
    __syncthreads();
  }
  const float *lut = LDS_LUT ? lut_s : A.lut;
  for (int64_t blk = blockIdx.x; blk < A.n_blocks; blk += gridDim.x) {
    const int D = A.block_dim[blk];
    const int64_t base = A.block_base[blk];
    const int32_t pos = A.block_pos[blk];
    const int K = A.K[blk];
    const int32_t *idx = A.indices + blk * (int64_t)A.max_K;
    bool bad_idx = false;                                     // (every thread reads the same row: uniform over the workgroup)
    if (K >= 0 && K <= A.max_K && K <= A.K_limit)
      for (int t = 0; t < K; ++t) bad_idx |= (uint32_t)idx[t] >= (uint32_t)A.S;
    if (K > A.max_K || K > A.K_limit || K < 0 || bad_idx) {                    // not decodable: its elements come out as mu_p
      for (int d = tid; d < D; d += 256) {
        const int64_t ixo = base + (A.perm ? (int64_t)A.perm[pos + d] : (int64_t)(pos + d));
        A.out_sample[ixo] = 0.f + A.p_loc[ixo];
      }
      continue;
    }
    // a thread decodes FOUR consecutive dims: one Philox block yields their four draws (two when the row start is not a
    // multiple of 4), instead of one block per dim with three of its four words thrown away
    for (int d0 = tid * 4; d0 < D; d0 += 256 * 4) {
      int64_t ix[4];
      float var_p[4], c[4], sample[4];
      bool valid[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        valid[i] = d0 + i < D;
        ix[i] = base + (A.perm ? (int64_t)A.perm[pos + (valid[i] ? d0 + i : d0)] : (int64_t)(pos + (valid[i] ? d0 + i : d0)));
        const float sp = A.p_scale[ix[i]];
        var_p[i] = sp * sp; c[i] = 0.f; sample[i] = 0.f;
      }
      uint32_t hs = 0u;
      for (int t = 0; t < K; ++t) {
        const float rho = A.rho[K - 1 - t];
        const StepSeed ss = make_step_seed(A.seed + t);
        const uint32_t it = (uint32_t)idx[t];
        const uint32_t h = hash_from_sum((int32_t)hs);
        uint32_t rm1[4];
        draw_rm1_x4(ss, (uint64_t)it * (uint64_t)D + (uint64_t)d0, rm1);   // (it * D + d0) & 3 is uniform: d0 % 4 == 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = rho * (var_p[i] - c[i]);
          const float sa = sqrtf(a);
          const uint32_t k = ((rm1[i] + 1u) * h) % IREC_P;
          sample[i] = sample[i] + sa * lut[k];
          c[i] = c[i] + a;
        }
        hs += it * (uint32_t)(69 + t);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (valid[i]) A.out_sample[ix[i]] = sample[i] + A.p_loc[ix[i]];
    }
  }
}

// test hook: dec_sqrt_core against sqrtf on every float32 bit pattern the core is allowed to see; out[0] = mismatches (NaN
// results count as equal to each other), out[1] = patterns tested
__global__ void dec_sqrt_test_kernel(unsigned long long *out) {
  unsigned long long bad = 0ull, seen = 0ull;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)b);
    if (!dec_sqrt_core_ok(x)) continue;
    const float f = dec_sqrt_core(x), g = sqrtf(x);
    ++seen;
    if (__float_as_uint(f) != __float_as_uint(g) && !(f != f && g != g)) ++bad;
  }
  atomicAdd(out, bad);
  atomicAdd(out + 1, seen);
}
hipError_t launch_dec_sqrt_test(unsigned long long *out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, 16, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dec_sqrt_test_kernel, dim3(4096), dim3(256), 0, st, out);
  return hipGetLastError();
}

size_t decode_lds_bytes(bool table) { return DEC_LUT_BYTES + (table ? 0 : DEC_DLOG_BYTES); }

// Shape of the tensor-staged decode: waves per workgroup (0 = the tensors do not fit: block-wise kernel) and its LDS bytes.
int decode_tensor_waves(int n, int bs, bool table, size_t *lds_out) {
  if (n < 1 || bs < 1) return 0;
  const int bpt = (n + bs - 1) / bs, upb = (bs + 255) / 256;
  const int64_t upt = (int64_t)bpt * upb;
  const size_t lds = decode_lds_bytes(table) + (((size_t)n * 4 + 15) & ~(size_t)15) + 16;   // table(s), region, unit counter
  // 96 VGPRs: five waves per SIMD, so two workgroups share a CU when each has at most ten waves
  const int rounds = (int)((upt + IREC_DEC_WMAX - 1) / IREC_DEC_WMAX);
  if (rounds > DEC_RMAX_BIG || lds > FAST_LDS_LIMIT) return 0;
  if (lds_out) *lds_out = lds;
#if IREC_DEC_DYN
  return (int)(upt < IREC_DEC_WMAX ? upt : IREC_DEC_WMAX);   // units are pulled: the wave count need not divide them
#else
  return (int)((upt + rounds - 1) / rounds);
#endif
}

hipError_t launch_decode(const DecArgs &A, int n_cu, hipStream_t st) {
  const bool table = A.K_tab > 0;
  if (A.tn > 0) {
    size_t lds = 0;
    const int nw = decode_tensor_waves(A.tn, A.tbs, table, &lds);
    if (nw < 1) return hipErrorInvalidValue;               // (the host checks first)
    const int per_cu = (int)(FAST_LDS_LIMIT / lds) < 2 ? 1 : 2;
    const int64_t cap = (int64_t)per_cu * n_cu;
    const int grid = (int)(A.n_tensors < cap ? A.n_tensors : cap);
    const int64_t upt = (int64_t)A.tbpt * ((A.tbs + 255) / 256);
    const bool small = (upt + nw - 1) / nw <= DEC_RMAX_SMALL;
    auto go = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid > 0 ? grid : 1), dim3(nw * 64), lds, st, A);
      return hipGetLastError();
    };
    if (table) return small ? go(decode_tensor_kernel<true, DEC_RMAX_SMALL>) : go(decode_tensor_kernel<true, DEC_RMAX_BIG>);
    return small ? go(decode_tensor_kernel<false, DEC_RMAX_SMALL>) : go(decode_tensor_kernel<false, DEC_RMAX_BIG>);
  }
  if (A.upb > 0) {
    const size_t lds = decode_lds_bytes(table);
    const int64_t units = A.n_blocks * (int64_t)A.upb;
    // resident workgroups: the LDS copy allows three (table) / two per CU; a small call gets one wave per unit
    const int64_t want = (units + DEC_NW - 1) / DEC_NW;
    const int64_t cap = (int64_t)(table ? 3 : 2) * n_cu;
    const int grid = (int)(want < cap ? want : cap);
    hipError_t e;
    if (table) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(decode_wave_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(decode_wave_kernel<true>, dim3(grid > 0 ? grid : 1), dim3(DEC_NT), lds, st, A);
    } else {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(decode_wave_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(decode_wave_kernel<false>, dim3(grid > 0 ? grid : 1), dim3(DEC_NT), lds, st, A);
    }
    return hipGetLastError();
  }
  if (A.n_blocks >= 16LL * n_cu) // four resident workgroups per CU (40 KB of LDS each), four blocks or more per workgroup
    hipLaunchKernelGGL(decode_kernel<true>, dim3(4 * n_cu), dim3(256), 0, st, A);
  else
    hipLaunchKernelGGL(decode_kernel<false>, dim3((unsigned)(A.n_blocks < 8LL * n_cu ? A.n_blocks : 8LL * n_cu)), dim3(256), 0, st, A);
  return hipGetLastError();
}

} // namespace irec
