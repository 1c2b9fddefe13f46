// irec_device.h -- device-side building blocks of the iREC beam-search kernels (gfx950 only).
//
// Arithmetic contract: everything below uses only IEEE-754 correctly rounded + - * / sqrt and
// explicit fma, in float32 unless stated; the translation unit is compiled with -ffp-contract=off, so no
// other fused operation exists.  That is what makes the kernels reproducible bit for bit on a CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IREC_P 10007u   // big_prime            (reference: rec/coding/beam_search_coder.py:30)
#define IREC_PM1 10006u // big_prime - 1, order of the multiplicative group, modulus of simple_hash (:35)
#define IREC_LUT2_BYTES (IREC_PM1 * 4u)

namespace irec {

// ---- Philox4x32-10, TensorFlow's stream layout ------------------------------------------
struct StepSeed { uint32_t k0, k1, c2, c3; };

// tf.random.set_seed(seed + t); tf.random.uniform(..., seed=seed + t): both seeds truncated mod 2^31-1,
// (0,0) -> (0, 2^31-1); key = seed1 (64 bit), counter words 2,3 = seed2 (reference: beam_search_coder.py:38-42).
__host__ __device__ inline StepSeed make_step_seed(int64_t seed_plus_t) {
  const int64_t M = 2147483647LL;
  int64_t a = seed_plus_t % M;
  if (a < 0) a += M;
  int64_t b = a;
  if (a == 0) b = M;
  StepSeed s;
  s.k0 = (uint32_t)a; s.k1 = 0u; s.c2 = (uint32_t)b; s.c3 = 0u;
  return s;
}

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// Philox block number `blk` of the step's stream.
__device__ __forceinline__ uint4 philox_block(const StepSeed &ss, uint64_t blk) {
  return philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), ss.c2, ss.c3, ss.k0, ss.k1);
}

// r-1 (in 0..10005) for ONE flat element e of the [S,1,D] int32 draw: 1 + u32 % 10006  (beam_search_coder.py:39-43)
__device__ __forceinline__ uint32_t draw_rm1(const StepSeed &ss, uint64_t e) {
  const uint4 o = philox_block(ss, e >> 2);
  const uint32_t lane = (uint32_t)e & 3u;
  const uint32_t u = lane == 0 ? o.x : lane == 1 ? o.y : lane == 2 ? o.z : o.w;
  return u % IREC_PM1;
}

// r-1 for the FOUR consecutive elements e0..e0+3.  `off = e0 & 3` must be wave-uniform (it is whenever every lane's
// first dim is a multiple of 4: off = (s*D) & 3).
__device__ __forceinline__ void draw_rm1_x4(const StepSeed &ss, uint64_t e0, uint32_t rm1[4]) {
  const uint32_t off = (uint32_t)e0 & 3u;
  const uint4 a = philox_block(ss, e0 >> 2);
  uint32_t u0 = a.x, u1 = a.y, u2 = a.z, u3 = a.w;
  if (off != 0u) { // uniform branch
    const uint4 b = philox_block(ss, (e0 >> 2) + 1);
    if (off == 1u) { u0 = a.y; u1 = a.z; u2 = a.w; u3 = b.x; }
    else if (off == 2u) { u0 = a.z; u1 = a.w; u2 = b.x; u3 = b.y; }
    else { u0 = a.w; u1 = b.x; u2 = b.y; u3 = b.z; }
  }
  rm1[0] = u0 % IREC_PM1; rm1[1] = u1 % IREC_PM1; rm1[2] = u2 % IREC_PM1; rm1[3] = u3 % IREC_PM1;
}

// simple_hash from the running int32 sum  sum_j idx[j]*(69+j)  (beam_search_coder.py:33-35), tf.math.floormod.
__device__ __forceinline__ uint32_t hash_from_sum(int32_t sum) {
  int32_t m = sum % (int32_t)IREC_PM1;
  if (m < 0) m += (int32_t)IREC_PM1;
  return (uint32_t)m + 1u;
}

// ---- deterministic float64 log (same operation sequence as the test oracle's restatement) ------------------
__device__ __forceinline__ double det_log(double x) {
  uint64_t bits = (uint64_t)__double_as_longlong(x);
  int64_t e = (int64_t)((bits >> 52) & 0x7FF);
  if (e == 0) {
    x = x * 18014398509481984.0;
    bits = (uint64_t)__double_as_longlong(x);
    e = (int64_t)((bits >> 52) & 0x7FF) - 54;
  }
  e -= 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m = __longlong_as_double((long long)bits);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  const double s = (m - 1.0) / (m + 1.0);
  const double s2 = s * s;
  double q = 1.0 / 25.0;
  q = q * s2 + 1.0 / 23.0;
  q = q * s2 + 1.0 / 21.0;
  q = q * s2 + 1.0 / 19.0;
  q = q * s2 + 1.0 / 17.0;
  q = q * s2 + 1.0 / 15.0;
  q = q * s2 + 1.0 / 13.0;
  q = q * s2 + 1.0 / 11.0;
  q = q * s2 + 1.0 / 9.0;
  q = q * s2 + 1.0 / 7.0;
  q = q * s2 + 1.0 / 5.0;
  q = q * s2 + 1.0 / 3.0;
  const double lnm = 2.0 * s + (2.0 * s) * (s2 * q);
  return (double)e * 0.6931471805599453 + lnm;
}

// KL(N(mq,sq) || N(mp,sp)) of one dim in float64 (canonical form of tfd.kl_divergence, beam_search_coder.py:57).
__device__ __forceinline__ double kl_dim(float mq, float sq, float mp, float sp) {
  const double t = (double)sq / (double)sp;
  const double r = t * t;
  const double dm = ((double)mq - (double)mp) / (double)sp;
  return 0.5 * (dm * dm) + (0.5 * (r - 1.0) - det_log(t));
}

// num_aux_variables = int32(ceil(total_kl / kl_per_partition))  (beam_search_coder.py:59)
__device__ __forceinline__ int32_t num_aux(float kl, float omega) {
  if (!(kl > 0.0f)) return 0;
  const float k = ceilf(kl / omega);
  if (!(k < 1.0e9f)) return 1000000000;
  return (int32_t)k;
}

// ---- per-dim, per-step constants of the Gaussian partition algebra ------------------------------------------
// reference: beam_search_coder.py:67-77 with coder.py:141-154 (get_auxiliary_coder / get_auxiliary_target).
struct StepConst { float a, sa, m, A, Bv, H; };

__device__ __forceinline__ StepConst step_constants(float rho, float dmu, float var_q, float var_p, float c) {
  StepConst o;
  const float a = rho * (var_p - c);                                              // auxiliary_var
  const float v = a + c;                                                          // + cumulative_auxiliary_variance
  const float m = dmu * v / var_p;                                                // auxiliary_target_mean
  const float var = var_q * (v * v) / (var_p * var_p) + v * (var_p - v) / var_p;  // auxiliary_target_var
  o.a = a;
  o.sa = sqrtf(a);                        // scale of the auxiliary coder N(0, sqrt(a))
  o.m = m;
  o.A = 0.5f * (1.0f / v - 1.0f / var);   // score(x) = const + (A*w + Bv)*w,  w = x - m
  o.Bv = m / v;
  o.H = o.A * (o.sa * o.sa);              // coefficient of z^2 once w = p + sa*z is expanded (p = beam - m)
  return o;
}

// Score of candidate (s, b), up to a constant shared by all candidates of the step (beam_search_coder.py:82-84):
//   sum_d [log N(x; m, s_t) - log N(x; 0, sqrt(v))] = const + sum_d (A w + Bv) w,   w = x - m = p + sa z,  p = beam - m
//                                                   = const + C_b + sum_d (G_bd + H_d z) z
// with   G_bd = ((A+A) p + Bv) sa,   H_d = A sa^2,   C_b = sum_d (A p + Bv) p.
// Only z depends on the sample index s, so the inner loop is two fma per proposal.
__device__ __forceinline__ float beam_G(float beam, float m, float A, float Bv, float sa) {
  const float p = beam - m;
  return fmaf(A + A, p, Bv) * sa;
}
__device__ __forceinline__ float beam_C_term(float acc, float beam, float m, float A, float Bv) {
  const float p = beam - m;
  return fmaf(fmaf(A, p, Bv), p, acc);
}
__device__ __forceinline__ float proposal_term(float acc, float z, float H, float G) {
  return fmaf(fmaf(H, z, G), z, acc);
}

// ---- ordering of candidates: tf.argsort(DESCENDING) == top_k: value desc, ties -> lower flat index ----
__device__ __forceinline__ uint32_t score_key(float s) {
  if (s != s) return 1u;          // NaN sorts after every number
  if (s == 0.0f) return 0x80000000u; // -0 == +0
  const uint32_t u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long cand_pack(uint32_t key, uint32_t flat) {
  return key ? (((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - flat)) : 0ull;
}

// ---- cross-lane moves on the VALU (DPP / permlane swaps): they stay off the LDS pipe, which the LUT gathers own ----
template <int CTRL, int BANK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK, false);
}
// value held by lane (lane ^ DIST), DIST in {32,16,8,4,2,1}
template <int DIST>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t v) {
  if constexpr (DIST == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); // r[0]: [lo,lo]  r[1]: [hi,hi]
    return (__lane_id() & 32) ? r[0] : r[1];
  } else if constexpr (DIST == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); // r[0]: even rows everywhere, r[1]: odd rows
    return (__lane_id() & 16) ? r[0] : r[1];
  } else if constexpr (DIST == 8) return dpp_u32<0x128, 0xF>(v, v);  // row_ror:8
  else if constexpr (DIST == 4) {
    const uint32_t r = dpp_u32<0x104, 0x5>(v, v);                     // row_shl:4 -> lanes with bit2 = 0 read lane+4
    return dpp_u32<0x114, 0xA>(r, v);                                 // row_shr:4 -> lanes with bit2 = 1 read lane-4
  } else if constexpr (DIST == 2) return dpp_u32<0x4E, 0xF>(v, v);   // quad_perm [2,3,0,1]
  else return dpp_u32<0xB1, 0xF>(v, v);                               // quad_perm [1,0,3,2]
}
template <int DIST>
__device__ __forceinline__ unsigned long long max_step_u64(unsigned long long v) {
  const uint32_t lo = xor_lane_u32<DIST>((uint32_t)v), hi = xor_lane_u32<DIST>((uint32_t)(v >> 32));
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = max_step_u64<32>(v); v = max_step_u64<16>(v); v = max_step_u64<8>(v);
  v = max_step_u64<4>(v); v = max_step_u64<2>(v); v = max_step_u64<1>(v);
  return v;
}

// v_writelane_b32: `value` (wave-uniform) into lane `lane_select` (wave-uniform) of `old`; the other lanes keep `old`.  This clang has no
// builtin for it; the intrinsic is reached by its IR name.
extern "C" __device__ int irec_llvm_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t writelane_u32(uint32_t value, int lane_select, uint32_t old) {
  return (uint32_t)irec_llvm_writelane_i32((int)value, lane_select, (int)old);
}

// canonical 64-lane tree: pair lanes at distance 32,16,8,4,2,1 (all lanes end with the same bits).
__device__ __forceinline__ float wave_tree_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_tree_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// The same trees on the VALU (DPP / permlane swaps) instead of ds_bpermute: same partner at every stage (lane ^ 32, 16, 8, 4, 2, 1), same
// additions, same bits -- and no LDS round trip per stage (round 6: six dependent trips through an LDS queue full of look-ups)
__device__ __forceinline__ float wave_tree_sum_valu(float v) {
  v = v + __uint_as_float(xor_lane_u32<32>(__float_as_uint(v)));
  v = v + __uint_as_float(xor_lane_u32<16>(__float_as_uint(v)));
  v = v + __uint_as_float(xor_lane_u32<8>(__float_as_uint(v)));
  v = v + __uint_as_float(xor_lane_u32<4>(__float_as_uint(v)));
  v = v + __uint_as_float(xor_lane_u32<2>(__float_as_uint(v)));
  v = v + __uint_as_float(xor_lane_u32<1>(__float_as_uint(v)));
  return v;
}
template <int DIST>
__device__ __forceinline__ double xor_lane_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = xor_lane_u32<DIST>((uint32_t)b), hi = xor_lane_u32<DIST>((uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_tree_sum_valu(double v) {
  v = v + xor_lane_f64<32>(v); v = v + xor_lane_f64<16>(v); v = v + xor_lane_f64<8>(v);
  v = v + xor_lane_f64<4>(v); v = v + xor_lane_f64<2>(v); v = v + xor_lane_f64<1>(v);
  return v;
}

} // namespace irec
