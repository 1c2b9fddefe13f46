// irec_host.cpp -- host side of libirec_hip.so: C ABI (include/irec.h), constant tables, argument checks, launches.
//
// Nothing here computes the hot path on the CPU: the host builds input-independent constant tables
// (quantile LUT, discrete-log table of Z_10007^*, power-law variance ratios, the tf.random.shuffle permutation)
// and launches the gfx950 kernels of irec_kernels.hip.  Without a HIP device every device entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <exception>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "irec_internal.h"   // include/irec.h (the boundary) + the diagnostic flags and test hooks this library also exports
#include "irec_kernels.h"

namespace {

thread_local std::string g_last_error;

irec_status fail(irec_status code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(IREC_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Entry points select the context's device for their launches and put the caller's current device back on return
// (a multi-GPU process keeps torch's / its own current device).
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev == device) return hipSuccess;
    e = hipSetDevice(device);
    changed = (e == hipSuccess);
    return e;
  }
  ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};
#define IREC_ON_DEVICE(dev) DeviceGuard guard_; HIP_TRY(guard_.enter(dev))

// ---- deterministic log, float64, IEEE basic ops only (same operation sequence as the device code) ----
double det_log(double x) {
  uint64_t bits;
  std::memcpy(&bits, &x, 8);
  int64_t e = (int64_t)((bits >> 52) & 0x7FF);
  if (e == 0) {
    x = x * 18014398509481984.0;
    std::memcpy(&bits, &x, 8);
    e = (int64_t)((bits >> 52) & 0x7FF) - 54;
  }
  e -= 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m;
  std::memcpy(&m, &bits, 8);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
  double q = 1.0 / 25.0;
  for (int n = 23; n >= 3; n -= 2) q = q * s2 + 1.0 / (double)n;
  const double lnm = 2.0 * s + (2.0 * s) * (s2 * q);
  return (double)e * 0.6931471805599453 + lnm;
}
inline float logf_det(float x) { return (float)det_log((double)x); }

// ---- Normal(0,1).quantile in float32: TFP 0.9.0 special_math._ndtri, operation by operation ----
// (called through tfd.Normal.quantile at rec/coding/beam_search_coder.py:49)
template <size_t N>
float poly(float v, const double (&c)[N]) { // highest power first; "mul then add", each rounded to float32
  float acc = (float)c[0];
  for (size_t i = 1; i < N; ++i) acc = acc * v + (float)c[i];
  return acc;
}

float ndtri_f32(float p) {
  static const double p0[] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1,
                              1.39312609387279679503E1, -1.23916583867381258016E0};
  static const double q0[] = {1.0, 1.95448858338141759834E0, 4.67627912898881538453E0, 8.63602421390890590575E1,
                              -2.25462687854119370527E2, 2.00260212380060660359E2, -8.20372256168333339912E1,
                              1.59056225126211695515E1, -1.18331621121330003142E0};
  static const double p1[] = {4.05544892305962419923E0, 3.15251094599893866154E1, 5.71628192246421288162E1,
                              4.40805073893200834700E1, 1.46849561928858024014E1, 2.18663306850790267539E0,
                              -1.40256079171354495875E-1, -3.50424626827848203418E-2, -8.57456785154685413611E-4};
  static const double q1[] = {1.0, 1.57799883256466749731E1, 4.53907635128879210584E1, 4.13172038254672030440E1,
                              1.50425385692907503408E1, 2.50464946208309415979E0, -1.42182922854787788574E-1,
                              -3.80806407691578277194E-2, -9.33259480895457427372E-4};
  static const double p2[] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0,
                              1.33303460815807542389E0, 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                              3.01581553508235416007E-4, 2.65806974686737550832E-6, 6.23974539184983293730E-9};
  static const double q2[] = {1.0, 6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0,
                              2.16236993594496635890E-1, 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                              2.89247864745380683936E-6, 6.79019408009981274425E-9};
  const float hi_cut = (float)0.8646647167633873; // 1 - exp(-2)
  const float lo_cut = (float)0.1353352832366127; // exp(-2)
  if (p <= 0.0f) return -INFINITY;
  if (p >= 1.0f) return INFINITY;
  const float mcp = p > hi_cut ? 1.0f - p : p;
  const float sp = mcp <= 0.0f ? 0.5f : mcp;
  const float w = sp - 0.5f, ww = w * w;
  float x_centre = w + (w * ww) * (poly(ww, p0) / poly(ww, q0));
  x_centre = x_centre * (float)(-2.5066282746310002);
  const float z = std::sqrt(-2.0f * logf_det(sp));
  const float first = z - logf_det(z) / z;
  const float rz = 1.0f / z;
  const float x_far = first - poly(rz, p2) / poly(rz, q2) / z;
  const float x_tail = first - poly(rz, p1) / poly(rz, q1) / z;
  const float x = sp > lo_cut ? x_centre : (z >= 8.0f ? x_far : x_tail);
  return p > hi_cut ? x : -x;
}

// ---- TF seed plumbing (python/framework/random_seed.py) + Philox4x32-10 on the host ----
void tf_seed_pair(int64_t global_seed, int64_t op_seed, uint64_t &s1, uint64_t &s2) {
  const int64_t M = 2147483647LL;
  int64_t a = global_seed % M; if (a < 0) a += M;
  int64_t b = op_seed % M;     if (b < 0) b += M;
  if (a == 0 && b == 0) b = M;
  s1 = (uint64_t)a; s2 = (uint64_t)b;
}

struct Philox {
  uint32_t key[2], ctr[4];
  Philox(uint64_t seed1, uint64_t seed2) {
    key[0] = (uint32_t)seed1; key[1] = (uint32_t)(seed1 >> 32);
    ctr[0] = ctr[1] = 0; ctr[2] = (uint32_t)seed2; ctr[3] = (uint32_t)(seed2 >> 32);
  }
  void block(uint64_t index, uint32_t out[4]) const {
    uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
      const uint64_t a = (uint64_t)0xD2511F53u * c0, b = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(b >> 32) ^ c1 ^ k0, n2 = (uint32_t)(a >> 32) ^ c3 ^ k1;
      c1 = (uint32_t)b; c3 = (uint32_t)a; c0 = n0; c2 = n2;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
  uint32_t element(uint64_t e) const {
    uint32_t o[4];
    block(e >> 2, o);
    return o[e & 3];
  }
};

// ---- CPython's random.Random(seed).randint(0, 2**31-1): the op seed TF hands to tf.random.shuffle ----
struct MT19937 {
  uint32_t mt[624];
  int idx;
  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  void init_by_array(const std::vector<uint32_t> &key) {
    init_genrand(19650218u);
    size_t i = 1, j = 0;
    for (size_t k = std::max<size_t>(624, key.size()); k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
      if (++j >= key.size()) j = 0;
    }
    for (size_t k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
};

int64_t py_first_randint31(int64_t seed) {
  const uint64_t a = seed < 0 ? (uint64_t)(-(seed + 1)) + 1u : (uint64_t)seed;
  std::vector<uint32_t> key{(uint32_t)a};
  if (a >> 32) key.push_back((uint32_t)(a >> 32));
  MT19937 g;
  g.init_by_array(key);
  uint32_t r;
  do r = g.next(); while (r >= 0x80000000u); // _randbelow(2**31) with k = 32 bits
  return (int64_t)r;
}

// ---- tf.random.normal on the host: Box-Muller on consecutive uint32 pairs of the Philox stream ----
// random_distributions.h: Uint32ToFloat keeps 23 mantissa bits, BoxMullerFloat clamps u1 at 1e-7.
inline float u32_to_unit_float(uint32_t x) {
  const uint32_t bits = (127u << 23) | (x & 0x7fffffu);
  float f;
  std::memcpy(&f, &bits, 4);
  return f - 1.0f;
}
inline void box_muller(uint32_t x0, uint32_t x1, float &f0, float &f1) {
  float u1 = u32_to_unit_float(x0);
  if (u1 < 1.0e-7f) u1 = 1.0e-7f;
  const float v1 = (float)(2.0 * 3.14159265358979323846 * (double)u32_to_unit_float(x1));
  const float u2 = std::sqrt(-2.0f * std::log(u1));
  f0 = std::sin(v1) * u2;
  f1 = std::cos(v1) * u2;
}
// stream of tf.random.set_seed(seed); tf.random.normal(shape, seed=None): element e <- output (e & 3) of Philox block e >> 2
struct TfNormalStream {
  Philox gen;
  uint64_t have = ~0ull;
  float f[4];
  explicit TfNormalStream(int64_t seed) : gen(0, 0) {
    uint64_t s1, s2;
    tf_seed_pair(seed, py_first_randint31(seed), s1, s2);
    gen = Philox(s1, s2);
  }
  float element(uint64_t e) {
    if ((e >> 2) != have) {
      uint32_t o[4];
      gen.block(e >> 2, o);
      box_muller(o[0], o[1], f[0], f[1]);
      box_muller(o[2], o[3], f[2], f[3]);
      have = e >> 2;
    }
    return f[e & 3];
  }
};

// stream of tf.random.stateless_normal(shape, seed=[seed0, seed1]) (rec/coding/utils.py:9-12): stateless_random_ops.cc
// GenerateKey scrambles the seed pair with one Philox block under the fixed key (0x3ec8f720, 0x02461e29) and counter
// (seed0 lo, seed0 hi, seed1 lo, seed1 hi); the result's words 0-1 become the key, words 2-3 the upper counter half; the
// fill is the stateful kernel's (Box-Muller on consecutive uint32 pairs, four outputs per Philox block, no skip).
struct TfStatelessNormalStream {
  Philox gen;
  uint64_t have = ~0ull;
  float f[4];
  TfStatelessNormalStream(int64_t seed0, int64_t seed1) : gen(0, 0) {
    const Philox scramble(((uint64_t)0x02461e29u << 32) | 0x3ec8f720u, (uint64_t)seed1);
    uint32_t mix[4];
    scramble.block((uint64_t)seed0, mix);
    gen = Philox(((uint64_t)mix[1] << 32) | mix[0], ((uint64_t)mix[3] << 32) | mix[2]);
  }
  float element(uint64_t e) {
    if ((e >> 2) != have) {
      uint32_t o[4];
      gen.block(e >> 2, o);
      box_muller(o[0], o[1], f[0], f[1]);
      box_muller(o[2], o[3], f[2], f[3]);
      have = e >> 2;
    }
    return f[e & 3];
  }
};

int round_up(int v, int m) { return (v + m - 1) / m * m; }
size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

} // namespace

struct irec_context {
  int device = -1;
  int n_cu = 0;
  int clock_mhz = 0;
  float *d_lut = nullptr;
  float *d_lut2 = nullptr;
  uint16_t *d_dlog4r = nullptr;
  float *d_rho = nullptr;
  int32_t n_rho = IREC_MAX_PARTITIONS;  // partitions the ratios cover: the power law's IREC_MAX_PARTITIONS, or the caller's table (irec_create_with)
  unsigned long long *d_dbg = nullptr; // IREC_STAMPS=1 diagnostics only
};

extern "C" {

const char *irec_last_error(void) { return g_last_error.c_str(); }
const char *irec_version(void) { return "irec-hip 0.1 (gfx950)"; }

int32_t irec_n_samples(double kl_per_partition, double extra_samples) {
  const double s = std::exp(kl_per_partition * extra_samples);   // int(np.exp(.)), beam_search_coder.py:28-29
  if (!(s >= 0.0)) return 0;                                       // NaN
  return s >= 2147483647.0 ? INT32_MAX : (int32_t)s;               // (saturates; irec_params takes at most 2^24 samples anyway)
}

double irec_codelength(int64_t n_indices, int32_t n_samples) { return (double)n_indices * std::log((double)n_samples); }

irec_status irec_build_lut(float *lut) {
  if (!lut) return fail(IREC_E_INVALID, "irec_build_lut: null output");
  lut[0] = 0.0f;
  for (int k = 1; k < IREC_BIG_PRIME; ++k) lut[k] = ndtri_f32((float)k / (float)IREC_BIG_PRIME);
  return IREC_OK;
}

irec_status irec_tf_shuffle_perm(int64_t seed, int64_t n, int64_t *perm) {
  if (n < 0 || (n > 0 && !perm)) return fail(IREC_E_INVALID, "irec_tf_shuffle_perm: bad arguments");
  if (n > 0xFFFFFFFFLL) return fail(IREC_E_INVALID, "irec_tf_shuffle_perm: n too large");
  for (int64_t i = 0; i < n; ++i) perm[i] = i;
  if (n <= 1) return IREC_OK;
  uint64_t s1, s2;
  tf_seed_pair(seed, py_first_randint31(seed), s1, s2);
  const Philox gen(s1, s2);
  uint32_t blk[4];
  for (int64_t i = 0; i < n - 1; ++i) { // random_shuffle_op.cc: forward Fisher-Yates, one uint32 per swap
    if ((i & 3) == 0) gen.block((uint64_t)i >> 2, blk);
    const int64_t j = i + (int64_t)(blk[i & 3] % (uint32_t)(n - i));
    std::swap(perm[i], perm[j]);
  }
  return IREC_OK;
}

irec_status irec_philox_uniform_int(int64_t seed, int64_t n, int32_t *out) {
  if (n < 0 || (n > 0 && !out)) return fail(IREC_E_INVALID, "irec_philox_uniform_int: bad arguments");
  uint64_t s1, s2;
  tf_seed_pair(seed, seed, s1, s2);
  const Philox gen(s1, s2);
  for (int64_t e = 0; e < n; ++e) out[e] = 1 + (int32_t)(gen.element((uint64_t)e) % (uint32_t)(IREC_BIG_PRIME - 1));
  return IREC_OK;
}

int64_t irec_importance_n_samples(double coding_bits) {
  const float n = std::ceil(std::exp((float)coding_bits * std::log(2.0f))); // float32 throughout, importance_sampling.py:50
  if (!(n >= 1.0f) || !(n < 2147483648.0f)) return -1;
  return (int64_t)n;
}

irec_status irec_tf_random_normal(int64_t seed, int64_t count, float *out) {
  if (count < 0 || (count > 0 && !out)) return fail(IREC_E_INVALID, "irec_tf_random_normal: bad arguments");
  TfNormalStream st(seed);
  for (int64_t e = 0; e < count; ++e) out[e] = st.element((uint64_t)e);
  return IREC_OK;
}

irec_status irec_tf_stateless_normal(int64_t seed0, int64_t seed1, int64_t count, float *out) {
  if (count < 0 || (count > 0 && !out)) return fail(IREC_E_INVALID, "irec_tf_stateless_normal: bad arguments");
  TfStatelessNormalStream st(seed0, seed1);
  for (int64_t e = 0; e < count; ++e) out[e] = st.element((uint64_t)e);
  return IREC_OK;
}

irec_status irec_importance_encode(const float *t_loc, const float *t_scale, const float *p_loc, const float *p_scale,
                                   int64_t n, double coding_bits, double alpha, int64_t seed, int64_t *out_index,
                                   float *out_sample) try {
  if (!t_loc || !t_scale || !p_loc || !p_scale || !out_index || !out_sample || n < 1)
    return fail(IREC_E_INVALID, "irec_importance_encode: bad arguments");
  if (!(alpha >= 1.0)) return fail(IREC_E_INVALID, "Alpha must be in the range [1, inf), but %g was given!", alpha); // :33-34
  const int64_t S = irec_importance_n_samples(coding_bits);
  if (S < 1) return fail(IREC_E_INVALID, "irec_importance_encode: coding_bits %g gives no valid sample count", coding_bits);
  // standardise the target w.r.t. the coding distribution (:40-41); per-dim constants of Normal.log_prob (TFP 0.9)
  const float half_log_2pi = (float)(0.5 * std::log(2.0 * 3.14159265358979323846));
  std::vector<float> tl(n), ts(n), ln_t(n);
  for (int64_t d = 0; d < n; ++d) {
    tl[d] = (t_loc[d] - p_loc[d]) / p_scale[d];
    ts[d] = t_scale[d] / p_scale[d];
    ln_t[d] = half_log_2pi + std::log(ts[d]);
  }
  TfNormalStream st(seed);
  // alpha < inf: Gumbel-max over alpha * w + g (:67-71), g = stateless_gumbel_sample([S], seed + 1) =
  // -log(-log(stateless_normal([S], [seed + 1, seed + 2]))) (rec/coding/utils.py:9-12 -- a NORMAL draw inside the double
  // log, as the reference has it: g is NaN wherever the draw is outside (0, 1]; tf.argmax never selects a NaN).
  const bool gumbel = !std::isinf(alpha);
  TfStatelessNormalStream gst(seed + 1, seed + 2);
  float best = -FLT_MAX;   // Eigen's ArgMaxTupleReducer: accumulator starts at (0, lowest()) and moves on a strict ">"
  int64_t best_s = 0;
  for (int64_t s = 0; s < S; ++s) {
    double acc = 0.0; // canonical reduction: float32 terms, summed in float64 in dim order, rounded to float32 once
    for (int64_t d = 0; d < n; ++d) {
      const float x = st.element((uint64_t)(s * n + d));
      const float dt = x / ts[d] - tl[d] / ts[d];
      const float lp_t = -0.5f * (dt * dt) - ln_t[d];
      const float dp = x / 1.0f - 0.0f / 1.0f;
      const float lp_p = -0.5f * (dp * dp) - (half_log_2pi + 0.0f);
      acc += (double)(lp_t - lp_p);
    }
    float w = (float)acc;
    if (gumbel) w = (float)alpha * w + (-std::log(-std::log(gst.element((uint64_t)s))));
    if (w > best) { best = w; best_s = s; } // tf.argmax: first maximum; a NaN never compares greater
  }
  *out_index = best_s;
  for (int64_t d = 0; d < n; ++d) out_sample[d] = p_scale[d] * st.element((uint64_t)(best_s * n + d)) + p_loc[d];
  return IREC_OK;
} catch (const std::exception &e) { return fail(IREC_E_INVALID, "irec_importance_encode: %s", e.what()); }

irec_status irec_importance_decode(const float *p_loc, const float *p_scale, int64_t n, int64_t index, int64_t seed,
                                   float *out_sample) {
  if (!p_loc || !p_scale || !out_sample || n < 1 || index < 0)
    return fail(IREC_E_INVALID, "irec_importance_decode: bad arguments");
  TfNormalStream st(seed);
  for (int64_t d = 0; d < n; ++d) out_sample[d] = p_scale[d] * st.element((uint64_t)(index * n + d)) + p_loc[d];
  return IREC_OK;
}

irec_status irec_create(int device, irec_context **out) { return irec_create_ex(device, nullptr, out); }

irec_status irec_create_ex(int device, const float *lut10007, irec_context **out) {
  irec_tables t{};
  t.lut10007 = lut10007;
  return irec_create_with(device, &t, out);
}

int32_t irec_max_partitions(const irec_context *ctx) { return ctx ? ctx->n_rho : 0; }

irec_status irec_create_with(int device, const irec_tables *tables, irec_context **out) try {
  if (!out) return fail(IREC_E_INVALID, "irec_create: null output");
  *out = nullptr;
  const float *lut10007 = tables ? tables->lut10007 : nullptr;
  const float *aux_ratios = tables ? tables->aux_ratios : nullptr;
  const int32_t n_aux = tables ? tables->n_aux_ratios : 0;
  if (aux_ratios) {   // fitted ratios (coder.py:203-231): positive numbers at most 1, as many as the caller has -- at most IREC_MAX_PARTITIONS
    if (n_aux < 1 || n_aux > IREC_MAX_PARTITIONS) return fail(IREC_E_INVALID, "irec_create_with: n_aux_ratios %d out of range [1, %d]", n_aux, IREC_MAX_PARTITIONS);
    for (int i = 0; i < n_aux; ++i)
      if (!(aux_ratios[i] > 0.0f) || !(aux_ratios[i] <= 1.0f)) return fail(IREC_E_INVALID, "irec_create_with: aux_ratios[%d] is not in (0, 1]", i);
  } else if (n_aux != 0) return fail(IREC_E_INVALID, "irec_create_with: n_aux_ratios without aux_ratios");
  if (lut10007)   // an injected quantile table must hold numbers: one NaN / inf entry would poison every score it touches
    for (int k = 1; k < IREC_BIG_PRIME; ++k)
      if (!std::isfinite(lut10007[k])) return fail(IREC_E_INVALID, "irec_create_ex: lut10007[%d] is not finite", k);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(IREC_E_NO_DEVICE, "irec_create: no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= count) return fail(IREC_E_INVALID, "irec_create: device %d out of range (%d)", device, count);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(IREC_E_NO_DEVICE, "irec_create: device %d is %s; this build targets gfx950 only", device, prop.gcnArchName);
  IREC_ON_DEVICE(device);

  const int P = IREC_BIG_PRIME;
  std::vector<float> lut(P);
  if (lut10007) { std::memcpy(lut.data(), lut10007, (size_t)P * sizeof(float)); lut[0] = 0.0f; }
  else irec_build_lut(lut.data());
  // smallest primitive root of 10007 and the discrete-log tables
  auto mulmod = [&](uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); };
  uint32_t g = 0;
  std::vector<uint32_t> powers(P - 1);
  for (uint32_t cand = 2; cand < (uint32_t)P && !g; ++cand) {
    uint32_t x = 1;
    bool ok = true;
    std::vector<char> seen(P, 0);
    for (int e = 0; e < P - 1; ++e) {
      if (seen[x]) { ok = false; break; }
      seen[x] = 1; powers[e] = x; x = mulmod(x, cand);
    }
    if (ok) g = cand;
  }
  if (!g) return fail(IREC_E_INVALID, "irec_create: no primitive root found");
  std::vector<float> lut2(P - 1);
  std::vector<uint16_t> dlog4r(P - 1);
  for (int e = 0; e < P - 1; ++e) {
    lut2[e] = lut[powers[e]];                 // lut2[e] = quantile(g^e / 10007)
    dlog4r[powers[e] - 1] = (uint16_t)(4 * e); // byte offset of dlog(r) for r - 1 = u32 % 10006
  }
  std::vector<float> rho(IREC_MAX_PARTITIONS);
  for (int i = 0; i < IREC_MAX_PARTITIONS; ++i) // get_auxiliary_ratio, coder.py:16,218-220 (float64 -> float32)
    rho[i] = (float)std::pow((double)i + 1.0, -0.7864636765648174);
  if (aux_ratios)   // ... or self.aux_variable_variance_ratios[index] (coder.py:231); entries past the table are never read (K_limit)
    for (int i = 0; i < IREC_MAX_PARTITIONS; ++i) rho[i] = i < n_aux ? aux_ratios[i] : 1.0f;

  irec_context *ctx = new irec_context();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  ctx->clock_mhz = prop.clockRate / 1000;
  ctx->n_rho = aux_ratios ? n_aux : IREC_MAX_PARTITIONS;
  const irec_status st = [&]() -> irec_status { // any failure below frees what was allocated so far
    HIP_TRY(hipMalloc(&ctx->d_lut, P * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_lut2, (P - 1) * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_dlog4r, (P - 1) * sizeof(uint16_t)));
    HIP_TRY(hipMalloc(&ctx->d_rho, rho.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(ctx->d_lut, lut.data(), P * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_lut2, lut2.data(), (P - 1) * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_dlog4r, dlog4r.data(), (P - 1) * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_rho, rho.data(), rho.size() * sizeof(float), hipMemcpyHostToDevice));
#ifdef IREC_HOST_STAMPS   // the stamps build only (make variants: variants/stamps.so, scripts/gpu_stamps.sh); the product reads no environment
    if (const char *e = std::getenv("IREC_STAMPS"); e && e[0] == '1') {
      HIP_TRY(hipMalloc(&ctx->d_dbg, 4096 * 16 * sizeof(unsigned long long)));
    }
#endif
    return IREC_OK;
  }();
  if (st != IREC_OK) { irec_destroy(ctx); return st; }
  *out = ctx;
  return IREC_OK;
} catch (const std::exception &e) { return fail(IREC_E_INVALID, "irec_create: %s", e.what()); }

void irec_destroy(irec_context *ctx) {
  if (!ctx) return;
  DeviceGuard guard_;
  (void)guard_.enter(ctx->device);
  (void)hipFree(ctx->d_lut); (void)hipFree(ctx->d_lut2); (void)hipFree(ctx->d_dlog4r); (void)hipFree(ctx->d_rho);
  (void)hipFree(ctx->d_dbg);
  delete ctx;
}

} // extern "C"

namespace {

struct Plan {
  bool fast;         // register-resident fast encoder with the Philox draw fused in
  bool table;        // fast encoder fed by per-call proposal tables (Philox hoisted out of the block kernel)
  bool team;         // table && two-teams-per-CU encoder over three table copies (the default where it applies)
  bool team_only;    // B > 32: no one-table / fused fast encoder exists; the team encoder takes every call, the generic kernel its deferred pass
  bool lone;         // team && one beam: the one-wave-per-block encoder (irec_lone.hip) stands in for the team encoder
  bool chunk;        // blocks of more than 1024 dims (block_size = None, 2048, ...): encode_chunk_kernel over the team encoder's tables
  int shape;         // team-encoder workgroup shape override (IREC_FLAG_SHAPE_*; 0 = default)
  int grid_cap;      // scratch slabs = resident workgroups / teams (persistent kernels pull blocks from an atomic counter)
  int one_grid_cap;  // resident workgroups of the one-workgroup-per-block encoder of this plan (small calls of a team plan too)
  int fast_grid_cap; // resident workgroups of the deferred pass of a table plan (fused-Philox encoder), 0 without tables
  size_t ws_per_wg;
  int dpad;
  int K_tab;         // partitions the proposal tables cover
  int n_tab;
  int tab_dim[4];
  size_t tab_off[4]; // byte offsets of the proposal tables inside the workspace (after the 256-byte counter block)
  size_t tab_bytes;  // total
  int gang_blocks;   // chunk plans: blocks of one call that gangs of teams may code (0: no gang build for the shape), and
  size_t gang_stride, gang_bytes;   // the exchange of one block / of all, behind the slabs
};
size_t plan_ws_bytes(const Plan &pl) { return irec::WS_HEAD_BYTES + pl.tab_bytes + (size_t)pl.grid_cap * pl.ws_per_wg + pl.gang_bytes; }

// steps of proposal tables the byte bounds allow at `per_step` bytes per step: what IREC_TABLE_BYTES_MAX holds, but not fewer
// than IREC_TABLE_STEPS_FLOOR while those stay within IREC_TABLE_BYTES_HARD
// (big: a call whose blocks exceed 1024 dims -- block_size = None on a whole tensor: K grows with the dims, and so must the window, or
//  every block falls to the generic kernel's second pass: IREC_TABLE_BYTES_BIG)
size_t table_steps_that_fit(size_t per_step, bool big = false) {
  const size_t soft = (big ? (size_t)IREC_TABLE_BYTES_BIG : (size_t)IREC_TABLE_BYTES_MAX) / per_step;
  const size_t hard = (big ? (size_t)IREC_TABLE_BYTES_BIG : (size_t)IREC_TABLE_BYTES_HARD) / per_step;
  return std::max(soft, std::min<size_t>((size_t)IREC_TABLE_STEPS_FLOOR, hard));
}

irec_status check_params(const irec_params *p) {
  if (!p) return fail(IREC_E_INVALID, "null irec_params");
  if (!(p->kl_per_partition > 0.0f)) return fail(IREC_E_INVALID, "kl_per_partition must be > 0");
  if (p->n_samples < 1 || p->n_samples > (1 << 24)) return fail(IREC_E_INVALID, "n_samples %d out of range", p->n_samples);
  if (p->n_beams < 1 || p->n_beams > IREC_MAX_BEAMS) return fail(IREC_E_INVALID, "n_beams %d out of range [1,%d]", p->n_beams, IREC_MAX_BEAMS);
  if (p->table_steps < 0) return fail(IREC_E_INVALID, "table_steps %d < 0", p->table_steps);
  { const int sh = (p->flags & IREC_FLAG_SHAPE_MASK) >> IREC_FLAG_SHAPE_SHIFT;
    if (sh > 6 || sh == 1 || sh == 4) return fail(IREC_E_INVALID, "unknown IREC_FLAG_SHAPE_* value"); }   // (1 and 4: the one-team and 2 x 2 shapes, removed in round 6)
  return IREC_OK;
}

Plan make_plan(const irec_context *ctx, const irec_params *p, int32_t max_dim, int32_t max_K) {
  Plan pl;
  const int B = p->n_beams, S = p->n_samples;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  pl.shape = (p->flags & IREC_FLAG_SHAPE_MASK) >> IREC_FLAG_SHAPE_SHIFT;
  pl.dpad = round_up(max_dim > 0 ? max_dim : 1, 256);
  pl.fast = !(p->flags & IREC_FLAG_FORCE_GENERIC) && max_dim <= irec::FAST_MAX_DIM && irec::fast_nb_for(B) != 0 &&
            irec::fast_lds_for(B, S, false) <= irec::FAST_LDS_LIMIT && (int64_t)S * B < (1 << 24);
  pl.grid_cap = 2 * n_cu; // measured residency: 2 workgroups per CU for every encoder
  pl.table = false; pl.team = false; pl.lone = false; pl.chunk = false; pl.n_tab = 0; pl.tab_bytes = 0;
  pl.gang_blocks = 0; pl.gang_stride = 0; pl.gang_bytes = 0;
  // table window: the tables cover the first K_tab partitions; blocks with more go to the fused-Philox second pass
  const int want = p->table_steps > 0 ? p->table_steps : IREC_TABLE_STEPS_DEFAULT;
  pl.K_tab = std::max(1, std::min(std::min(want, IREC_TABLE_STEPS_MAX), max_K > 0 ? max_K : 1));
  {
    size_t per_step = 0;   // bytes of one partition step over all tables of the call
    for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) per_step += (size_t)S * round_up(p->table_dims[q], 4) * 2;
    if (per_step > 0) pl.K_tab = std::max(1, (int)std::min<size_t>((size_t)pl.K_tab, table_steps_that_fit(per_step, max_dim > irec::FAST_MAX_DIM)));
  }
  // B > 32 has no one-workgroup-per-block encoder; where the team encoder serves it (32 < B <= 60) it is the only table
  // consumer and its deferred pass is the generic kernel
  pl.team_only = !pl.fast && !(p->flags & IREC_FLAG_FORCE_GENERIC) && max_dim <= irec::FAST_MAX_DIM && (int64_t)S * B < (1 << 24) &&
                 !(p->flags & (IREC_FLAG_FUSED_PHILOX | IREC_FLAG_ONE_TABLE)) && irec::team_lds_for(B, S, pl.shape) != (size_t)-1;
  if ((pl.fast || pl.team_only) && !(p->flags & IREC_FLAG_FUSED_PHILOX) && p->table_dims[0] > 0) {
    pl.table = true;
    for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) {
      if (p->table_dims[q] > irec::FAST_MAX_DIM) { pl.table = false; break; }
      pl.tab_dim[pl.n_tab] = p->table_dims[q];
      pl.tab_off[pl.n_tab] = pl.tab_bytes;
      pl.tab_bytes += round_up_sz((size_t)pl.K_tab * S * round_up(p->table_dims[q], 4) * 2, 256);
      ++pl.n_tab;
    }
    if (!pl.table) { pl.n_tab = 0; pl.tab_bytes = 0; }
    pl.team = pl.table && !(p->flags & IREC_FLAG_ONE_TABLE) && irec::team_lds_for(B, S, pl.shape) != (size_t)-1;
    if (pl.team_only && !pl.team) { pl.table = false; pl.n_tab = 0; pl.tab_bytes = 0; }
    if (pl.team) pl.grid_cap = std::max(irec::team_count_for(B, S, pl.shape), (p->flags & IREC_FLAG_NO_TEN) ? 0 : irec::team_ten_teams(B, S, pl.shape)) * n_cu; // one scratch slab per team
    pl.lone = pl.team && irec::lone_applies(B, pl.shape);
  }
  if (!pl.table) pl.team_only = false;
  // Blocks of more than 1024 dims -- Coder.__init__ takes any block_size, None (the whole tensor as one block) included,
  // coder.py:29-36,415-419 -- are walked in chunks of 1024 by encode_chunk_kernel (irec_team.hip); its second pass and
  // everything it does not serve (B > 20, more samples than one pass holds, no dim hints) is the generic kernel's.
  pl.chunk = !pl.fast && !pl.team_only && !(p->flags & (IREC_FLAG_FORCE_GENERIC | IREC_FLAG_FUSED_PHILOX | IREC_FLAG_ONE_TABLE)) &&
             p->table_dims[0] > 0 && irec::chunk_applies(B, S, max_dim);
  if (pl.chunk) {
    pl.table = true;
    for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) {
      if (p->table_dims[q] > max_dim) { pl.table = false; break; }
      pl.tab_dim[pl.n_tab] = p->table_dims[q];
      pl.tab_off[pl.n_tab] = pl.tab_bytes;
      pl.tab_bytes += round_up_sz((size_t)pl.K_tab * S * round_up(p->table_dims[q], 4) * 2, 256);
      ++pl.n_tab;
    }
    if (!pl.table) { pl.chunk = false; pl.n_tab = 0; pl.tab_bytes = 0; }
  }
  // resident workgroups of the one-workgroup-per-block encoders: two per CU, one for the big-LDS 8-wave configurations
  auto one_cap = [&](bool table) { return (irec::fast_waves_for(B, S, table) == 8 ? 1 : 2) * n_cu; };
  if (pl.fast) {
    pl.one_grid_cap = one_cap(pl.table);
    pl.fast_grid_cap = pl.table ? one_cap(false) : 0;              // deferred pass of a table plan (fused Philox)
    if (!pl.team) pl.grid_cap = pl.one_grid_cap;
    pl.grid_cap = std::max(pl.grid_cap, pl.fast_grid_cap);          // both passes index the same slabs
    pl.ws_per_wg = round_up_sz(irec::fast_ws_for(B, max_K) + (pl.team ? irec::team_ws_extra_for(B, S, pl.shape) : 0), 256);
    if (pl.team) pl.ws_per_wg = std::max(pl.ws_per_wg, round_up_sz(irec::team_ws_bytes_for(B, S, pl.shape, max_K), 256));
    // (the one-wave-per-block encoder lays its own slabs -- one workgroup per CU, a statistics slab per wave -- over the same area)
    if (pl.lone) pl.ws_per_wg = std::max(pl.ws_per_wg, round_up_sz(((size_t)n_cu * irec::lone_ws_bytes_per_wg() + pl.grid_cap - 1) / pl.grid_cap, 256));
  } else {
    const size_t generic_ws = round_up_sz((size_t)10 * pl.dpad * 4 + (size_t)3 * B * pl.dpad * 4 +   // (beams [2][B], G [B])
                                          (size_t)(max_K > 0 ? max_K : 1) * B * 4 + (size_t)S * B * 4, 256);
    if (pl.team_only) {   // slabs serve the team encoder and, for blocks beyond the table window, the generic kernel
      pl.one_grid_cap = 0;
      pl.fast_grid_cap = 2 * n_cu;
      pl.grid_cap = std::max(pl.grid_cap, pl.fast_grid_cap);
      pl.ws_per_wg = std::max(generic_ws, round_up_sz(irec::team_ws_bytes_for(B, S, pl.shape, max_K), 256));
    } else if (pl.chunk) {   // one slab per team of the chunked encoder (it codes every block itself: no second pass)
      pl.one_grid_cap = 0;
      pl.fast_grid_cap = 0;
      pl.ws_per_wg = round_up_sz(irec::chunk_ws_for(B, pl.dpad, max_K), 256);
      // blocks of hundreds of thousands of dims: 55 MB of slab each at 301 056 dims -- no more slabs than IREC_SLAB_BYTES_MAX holds (and at
      // least one workgroup's): such a call has a handful of blocks, each of which holds its team for seconds
      const int teams = irec::chunk_teams(B, S);
      const size_t fit = (size_t)IREC_SLAB_BYTES_MAX / pl.ws_per_wg;
      pl.grid_cap = (int)std::max<size_t>((size_t)teams, std::min<size_t>((size_t)teams * n_cu, fit / teams * teams));
      // gangs (calls of fewer blocks than team slots, gang_width below): the exchange of up to GANG_MAX_BLOCKS blocks behind the slabs
      if (const int gnb = irec::chunk_gang_nb(B, S); gnb && irec::chunk_gang_teams(B, S) > 0 && !(p->flags & (IREC_FLAG_NO_SPLIT | IREC_FLAG_MARGINS))) {
        pl.gang_stride = irec::gang_xch_bytes(gnb, S, pl.dpad);
        pl.gang_blocks = (int)std::min<size_t>((size_t)irec::GANG_MAX_BLOCKS, irec::GANG_XCH_BYTES_MAX / pl.gang_stride);
        pl.gang_bytes = (size_t)pl.gang_blocks * pl.gang_stride;
      }
    } else {
      pl.one_grid_cap = pl.grid_cap; pl.fast_grid_cap = 0;
      pl.ws_per_wg = generic_ws;
    }
  }
  if (p->flags & IREC_FLAG_MARGINS) {
    // top-B margins (irec_beam_encode_ex): a margin build of the team encoder where one exists, the generic kernel for everything else and
    // for the blocks beyond the table window -- the slabs serve both
    const size_t generic_ws = round_up_sz((size_t)10 * pl.dpad * 4 + (size_t)3 * B * pl.dpad * 4 +
                                          (size_t)(max_K > 0 ? max_K : 1) * B * 4 + (size_t)S * B * 4, 256);
    pl.ws_per_wg = std::max(pl.ws_per_wg, generic_ws);
    pl.grid_cap = std::max(pl.grid_cap, 2 * n_cu);
    if (max_dim > irec::FAST_MAX_DIM)   // (huge blocks: no more slabs than IREC_SLAB_BYTES_MAX holds; the generic kernel's grid follows)
      pl.grid_cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)pl.grid_cap, (size_t)IREC_SLAB_BYTES_MAX / pl.ws_per_wg));
  }
  if (!pl.table) pl.K_tab = 0;
  return pl;
}

// (defined below) workgroups of the chunked encoder for a call: one per CU, not more than blocks nor than the plan's slabs
static int chunk_grid(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks);

// Workgroups of a persistent batch encoder (team / one-beam): one per CU, not more than blocks -- rounded up to a multiple of
// 8 where that fits, so that hand-out slot u runs on XCD u mod 8 (irec_fast_common.h: xcd_static_row); the extra workgroups
// find no row and leave.
static int batch_grid(int64_t n_blocks, int cap) {
  const int64_t g = std::min<int64_t>(n_blocks, cap), r = (g + 7) & ~(int64_t)7;
  return (int)(r <= cap ? r : g);
}

static int chunk_grid(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int teams = std::max(1, irec::chunk_teams(p->n_beams, p->n_samples));
  return batch_grid(n_blocks, std::min(n_cu, std::max(1, pl.grid_cap / teams)));
}
// Gangs of the chunked encoder (irec_team.hip, "Gangs"): a call of fewer blocks than team slots -- block_size = None on one image's
// latents -- has G teams code each block together, a chunk of 1024 dims (or several) per member.  All n_blocks * G teams must be resident
// at once (one static hand-out slot each; they wait for each other twice per step).  Returns G (0: every block on one team) and the
// grid that puts the members on CUs of their own as far as the CUs go.
int gang_width(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks, int32_t max_block_dim, int *grid, int *chunk_owners) {
  if (!pl.chunk || pl.gang_blocks < 1 || n_blocks < 1 || n_blocks > pl.gang_blocks || (p->flags & (IREC_FLAG_NO_SPLIT | IREC_FLAG_MARGINS))) return 0;
  if (!irec::IREC_COOP_GRANULES_ON) return 0;   // (the gangs' arrival counters are exchange granules: zeroed by the preparation kernel only in that form)
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int teams = irec::chunk_gang_teams(p->n_beams, p->n_samples);   // (of the gang build: three, or one for the shapes without a three-team build)
  if (teams < 1) return 0;
  const int wgs = std::min(n_cu, std::max(1, pl.grid_cap / teams));
  const int64_t slots = (int64_t)wgs * teams;
  const int64_t chunks = ((int64_t)max_block_dim + 1023) >> 10;
  const int64_t GC = std::min(chunks, slots / n_blocks);               // chunk owners per block
  if (GC < 1) return 0;
  // ... x sample stripes per chunk: the sample-chunks of a step (two samples each) shared between SP teams -- while there are CUs without a
  // member: a stripe repeats its chunk's statistics, step constants and update, which costs more than it saves where members share a CU
  // (r05s/gang_stripes.log: 24 blocks of 8192 dims 6.1 ms with 192 members, 8.9 ms with 576)
  const int want = (p->flags & IREC_FLAG_SPLIT_MASK) >> IREC_FLAG_SPLIT_SHIFT;
  // A stripe takes whole sample-chunks (two samples; one in the 16-beam passes of the 32-slot build), so the step is as long as the
  // stripe with the most of them: of the stripe counts the cap allows, the FEWEST that reach the shortest step (S = 36: 18 sample-chunks,
  // cap 9 -> 9 stripes of 2; cap 8 -> 6 stripes of 3, not 8 of 3 or 2)
  const int64_t n_sch = irec::chunk_gang_nb(p->n_beams, p->n_samples) == 32 ? p->n_samples : (p->n_samples + 1) / 2;
  int64_t SP = std::min<int64_t>(wgs / (n_blocks * GC), want >= 1 ? want : IREC_GANG_STRIPES);
  SP = std::max<int64_t>(1, std::min<int64_t>(SP, n_sch));
  SP = (n_sch + (n_sch + SP - 1) / SP - 1) / ((n_sch + SP - 1) / SP);
  const int64_t G = GC * SP;
  if (G < 2) return 0;
  const int64_t n_slots = n_blocks * G;
  if (grid) *grid = (int)std::min<int64_t>(wgs, n_slots);   // slot u = team u / grid of workgroup u % grid: one member per CU first
  if (chunk_owners) *chunk_owners = (int)GC;
  return (int)G;
}

// Calls of fewer blocks than this take the one-table / split encoders (cheaper set-up: a 6 us plain table and 40 KB of LDS to fill, against
// the bank assignment and 120 KB); from here on the team encoder.  A QUARTER OF THE CUs -- 64 on the 256-CU device the crossover was measured
// on (r02b, r03m) -- within what the split encoder's arrival counters hold (COOP_SPLIT_MAX_BLOCKS) and not below 8.
int small_call_blocks(const irec_context *ctx) {
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  return std::max(8, std::min(irec::COOP_SPLIT_MAX_BLOCKS, n_cu / 4));
}

// Split encoder for calls of so few blocks that most CUs would idle (one image's residual block: 9 blocks): W workgroups
// per block, each scoring a stripe of the samples (irec_kernels.hip).  All n_blocks * W workgroups must be resident at once
// (they wait for each other every step), so the grid stays within HALF the CUs -- room for a second such call on another
// stream -- and W within what the exchange buffers hold.  0 = not split.
int split_width(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  if (!pl.table || pl.chunk || (p->flags & IREC_FLAG_NO_SPLIT)) return 0;
  const int B = p->n_beams, S = p->n_samples, nb = irec::fast_nb_for(B);
  if (!nb || irec::fast_waves_for(B, S, true) != 4 || (int64_t)S * nb > 1024) return 0;   // aliased-key 4-wave builds only
  if (n_blocks < 1 || n_blocks > irec::COOP_SPLIT_MAX_BLOCKS) return 0;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  int64_t W = (n_cu / 2) / n_blocks;
  W = std::min<int64_t>(W, S);                                             // at least one sample per workgroup
  const int want = (p->flags & IREC_FLAG_SPLIT_MASK) >> IREC_FLAG_SPLIT_SHIFT;
  W = std::min<int64_t>(W, want >= 2 ? want : 12);                         // exchange + merge grow with W; scoring is ~S/W
  return W >= 2 ? (int)W : 0;
}
// Beam mode of the split encoder (irec_kernels.hip): the W workgroups of a block own its beam slots w, w + W instead of
// sample stripes -- possible when two slots per workgroup cover the B beams.  Returns the width to launch with (the
// fewest workgroups that still own at most as many slots each: 10, not 12, for B = 20), 0 = stay with sample stripes.
int split_beam_width(const irec_params *p, int W) {
  if (W < 2) return 0;
  const int B = p->n_beams;
  if (W > B) W = B;
  if (W < 2) return 0;
  const int per = (B + W - 1) / W;                                         // slots per workgroup
  if (per > 2) return 0;
  while (W > 2 && (B + (W - 1) - 1) / (W - 1) == per) --W;
  return W;
}

// Shared rows of the team encoder (irec_team.hip) for calls of one to one-and-a-half blocks per CU with more than ten beams (one
// GPU's share of config 3: 342 blocks).  Every CU gets ONE whole block; each of the n_blocks - n_cu rows beyond that is coded by W
// teams in the idle team slots of W CUs, which split its samples and exchange sort keys, so a CU carries one block and a fraction
// instead of two -- the call is as long as its most loaded CU (r04i: 342 blocks 0.59 -> 0.555 ms, 297 blocks 0.575 -> 0.54 ms).
// Not for B <= 10 (302 blocks of a Kodak level: 0.26 ms either way) and not beyond 1.5 blocks per CU (W = 1).
// (Round 6: the diagnostic flag that shared EVERY row of any mid-size call on the three-team build -- slower than the default at every
//  size, r04j -- is gone, and with it the builds only it reached.)
// Returns W (0: no sharing) for the shape the call runs; *first = first shared row, *grid = workgroups the static round needs.
// Calls of 64 blocks up to ~ a block per CU, no shape pinned: EVERY row is shared between the teams of the two-team build (shape 2)
// instead of sitting alone on a CU's one team -- r04x/share_all_probe.log, max_K = 32, as issued: B = 20: 72 blocks 0.296 -> 0.245 ms,
// 126 blocks 0.331 -> 0.287, 162 blocks 0.332 -> 0.312 (three partners or more; with two the per-step wait for the slower partner
// costs more than half a step's scoring saves: 180 blocks 0.332 -> 0.368); B = 10 (whose default build has three 4-wave teams of which
// such a call uses one): 72 blocks 0.159 -> 0.134, 252 blocks 0.183 -> 0.168 -- also with two partners.
bool share_all_auto(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  if (!pl.team || pl.lone || pl.team_only || pl.chunk || !pl.table || (p->flags & IREC_FLAG_NO_SPLIT)) return false;
  if ((p->flags & IREC_FLAG_SHAPE_MASK) != 0 || n_blocks < small_call_blocks(ctx) || n_blocks > irec::COOP_MAX_BLOCKS) return false;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int B = p->n_beams, S = p->n_samples;
  if (B > 20 || irec::team_shareable(B, S, 2) != 2 || irec::team_lds_for(B, S, 2) == (size_t)-1 ||
      irec::team_ws_extra_for(B, S, 2) > irec::team_ws_extra_for(B, S, 0) || irec::team_count_for(B, S, 0) < 2)
    return false;
  const int64_t W = std::min<int64_t>(std::min<int64_t>(8, S), 2LL * n_cu / n_blocks);
  // (round 6: calls the ten-beam encoder serves stay whole -- a lone 10 x 20 chain on encode_ten_kernel is as fast as or faster than the same
  //  row shared by partners on the team encoder: 126 / 180 / 252 blocks 0.17 / 0.18 / 0.18 ms against 0.18 / 0.19 / 0.20, 72 blocks level)
  if (B <= 10 && !(p->flags & IREC_FLAG_NO_TEN) && irec::ten_applies(B, S)) return false;
  return B > 10 ? W >= 3 : (W >= 2 && n_blocks <= n_cu);
}
int team_share_width(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks, int shape, int64_t *first, int *grid) {
  if (!pl.team || pl.lone || pl.team_only || (p->flags & IREC_FLAG_NO_SPLIT)) return 0;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int teams = irec::team_shareable(p->n_beams, p->n_samples, shape);
  if (teams < 2) return 0;
  const int want = (p->flags & IREC_FLAG_SPLIT_MASK) >> IREC_FLAG_SPLIT_SHIFT;
  const int64_t cap = std::min<int64_t>(8, p->n_samples);
  const int64_t slots = (int64_t)teams * n_cu;
  int64_t W = 0, f = 0;
  if (shape == 2 && share_all_auto(ctx, pl, p, n_blocks)) {
    if (n_blocks < small_call_blocks(ctx) || n_blocks > irec::COOP_MAX_BLOCKS || 2 * n_blocks > slots) return 0;
    W = slots / n_blocks;
  } else {
    if (p->n_beams <= 10 || n_blocks <= n_cu || n_blocks >= slots || n_blocks - n_cu > irec::COOP_MAX_BLOCKS) return 0;
    f = n_cu;
    W = (slots - n_cu) / (n_blocks - n_cu);        // every slot in the static round: all partners resident at once
  }
  W = std::min(W, cap);
  if (want >= 2) W = std::min<int64_t>(W, want);
  if (W < 2) return 0;
  const int64_t n_slots = f + (n_blocks - f) * W;
  int64_t g = n_cu;                                  // (whole rows: hand-out slots 0 .. n_cu - 1 = team 0 of every workgroup)
  if (f == 0) { g = (n_slots + teams - 1) / teams; g = std::min<int64_t>(n_cu, (g + 7) & ~(int64_t)7); }
  if (g * teams < n_slots) return 0;
  if (first) *first = f;
  if (grid) *grid = (int)g;
  return (int)W;
}

// teams per workgroup of the kernel a team call of this shape runs: encode_ten_kernel's (irec_ten.hip) for plain calls of at most ten beams
int call_teams(const irec_params *p, int shape, int share_W, bool margins) {
  const int tt = (share_W >= 2 || margins || (p->flags & IREC_FLAG_NO_TEN)) ? 0 : irec::team_ten_teams(p->n_beams, p->n_samples, shape);
  return tt ? tt : irec::team_count_for(p->n_beams, p->n_samples, shape);
}
// small calls (a single image's res-block: 9 blocks) are latency-bound: the one-table encoder's set-up (a 6 us proposal
// table, 40 KB of LDS to fill) beats the team encoder's (38 us per table for the bank assignment, 120 KB); the scratch
// sized for the team plan covers both
bool team_for_call(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  return pl.team && (pl.team_only || (p->flags & IREC_FLAG_TEAM) || n_blocks >= small_call_blocks(ctx));
}
// Workgroup shape of the team encoder for THIS call.  With at most one block per CU a lone 4-wave team is latency-bound
// (one wave per SIMD, ~43 us per step at B = 20, S = 36): the 8-wave beam-striped team (two stripes of 10 beams: half the
// look-ups and half the update per wave) codes 252 blocks in 0.45 ms against 0.57 ms.  From ~1.3 blocks per CU on the
// three-team shape wins again.  Same scratch (fewer slabs, same slab size), same outputs.
int shape_for_call(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int B = p->n_beams, S = p->n_samples;
  if (share_all_auto(ctx, pl, p, n_blocks)) return 2;                             // every row shared between the teams of the two-team build
  if (pl.shape != 0 || !pl.team || pl.lone || n_blocks < small_call_blocks(ctx) || n_blocks > 2 * (int64_t)n_cu || B > 20) return pl.shape;   // (< 64 blocks: only calls
                                                                         // that pin IREC_FLAG_TEAM get here, tests of the default shape among them)
  if (irec::team_count_for(B, S, 0) < 2) return pl.shape;                       // already one striped team
  if (n_blocks <= n_cu && B <= 10) return pl.shape;                             // (no 8-wave build for 10 beams)
  if (n_blocks > n_cu) {
    // One to two blocks per CU (config 3's per-GPU share: 38 images = 342 blocks per call): every block finds a team at
    // once either way, and the two-team build's teams are the faster ones (256 VGPRs: full look-up pipeline, nothing
    // parked or spilled) -- r02i: 342 blocks 0.63 -> 0.595 ms, 513 blocks 0.675 -> 0.64 ms; from 2.7 blocks per CU on the
    // third team wins again (684 blocks: 0.83 against 0.87 ms).  The 10-beam build likewise (the 302 blocks of a Kodak
    // image's first level: B = 10, 306 blocks 0.28 -> 0.26 ms, 513 blocks 0.31 -> 0.29 ms; 630 blocks 0.355 against 0.385 ms).
    if (irec::team_count_for(B, S, 0) < 3 || irec::team_lds_for(B, S, 2) == (size_t)-1 ||
        irec::team_ws_extra_for(B, S, 2) > irec::team_ws_extra_for(B, S, 0))
      return pl.shape;
    return 2;
  }
  if (irec::team_lds_for(B, S, 5) == (size_t)-1 || irec::team_ws_extra_for(B, S, 5) > irec::team_ws_extra_for(B, S, 0))
    return pl.shape;
  return 5;
}

// A call with IREC_FLAG_MARGINS: the team-encoder shape whose MARGIN build (irec_team_margin.hip) serves it, or -1 = the generic kernel.
// No block is shared under the flag (the cooperative forms have no margin builds), so the call's shape is the plain one of its size.
int margin_team_shape(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks) {
  if (!pl.table || !pl.team || pl.lone || pl.chunk || (p->flags & (IREC_FLAG_FORCE_GENERIC | IREC_FLAG_FUSED_PHILOX | IREC_FLAG_ONE_TABLE))) return -1;
  const int shape = shape_for_call(ctx, pl, p, n_blocks);
  if (irec::team_margin_build(p->n_beams, p->n_samples, shape)) return shape;
  if (irec::team_margin_build(p->n_beams, p->n_samples, pl.shape)) return pl.shape;
  return -1;
}

// What ONE call launches, in numbers: the grid, the teams (= scratch slabs) per workgroup, the cooperative width and what the kernels'
// own checks rest on.  irec_beam_encode_ex takes its launch from here; irec_test_plan (csrc/irec_internal.h) returns it for any CU count,
// so that the planner's invariants are tested host-only at 32 ... 304 CUs (round 5's review: every threshold had been measured on one
// 256-CU box).  `pl` is the call's plan with pl.team / pl.shape already settled for the call.
struct CallDetail {
  int kind = 0;             // 1 chunk, 2 lone, 3 team, 4 one-table / split (fast, table), 5 fused fast, 6 generic
  int grid = 0;             // workgroups of the block kernel
  int teams = 1;            // teams (scratch slabs) per workgroup
  int W = 0;                // cooperative width: teams per shared row / workgroups per block / members per gang; 0 = nothing is shared
  int coop_beams = 0;       // split encoder: beams, not samples, are shared
  int gang_chunks = 0;      // chunk owners of a gang
  int64_t share_first = 0;  // team encoder: first shared row
  int64_t n_slots = 0;      // hand-out slots of the call (whole rows + W per shared row)
  int placed = 0;           // team encoder: rows dealt by cost (EncArgs::row_cost)
  int split_blocks = 0;     // blocks whose exchange granules the preparation kernel zeroes
};
CallDetail call_detail(const irec_context *ctx, const Plan &pl, const irec_params *p, int64_t n_blocks, int32_t max_block_dim, bool margins) {
  CallDetail d;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  d.n_slots = n_blocks;
  if (pl.table && !pl.team && split_width(ctx, pl, p, n_blocks) >= 2) d.split_blocks = (int)n_blocks;   // (pl.team: of THIS call)
  int share_grid = 0;
  const int share_W = (pl.table && pl.team) ? team_share_width(ctx, pl, p, n_blocks, pl.shape, &d.share_first, &share_grid) : 0;
  if (share_W >= 2) d.split_blocks = (int)(n_blocks - d.share_first);
  if (pl.chunk && gang_width(ctx, pl, p, n_blocks, max_block_dim, nullptr, nullptr) >= 2) d.split_blocks = (int)n_blocks;   // (their granules' first words: the gangs' arrival counters)
  if (pl.table && pl.chunk) {
    d.kind = 1;
    d.teams = std::max(1, irec::chunk_teams(p->n_beams, p->n_samples));
    d.grid = chunk_grid(ctx, pl, p, n_blocks);
    int ggrid = 0, gchunks = 0;
    if (const int G = gang_width(ctx, pl, p, n_blocks, max_block_dim, &ggrid, &gchunks)) {
      d.W = G; d.gang_chunks = gchunks; d.grid = ggrid; d.n_slots = n_blocks * G;
      d.teams = std::max(1, irec::chunk_gang_teams(p->n_beams, p->n_samples));
    }
  } else if (pl.table && pl.team && pl.lone) {
    d.kind = 2; d.grid = batch_grid(n_blocks, n_cu); d.teams = irec::lone_waves();
  } else if (pl.table && pl.team) {
    d.kind = 3;
    d.teams = call_teams(p, pl.shape, share_W, margins);
    // one workgroup per CU as soon as there is a block for it: team k of workgroup w starts on block k * grid + w
    d.grid = batch_grid(n_blocks, std::min(pl.grid_cap / d.teams, n_cu));
    if (share_W >= 2) {   // rows [share_first, n_blocks) are coded by share_W teams each; the static round deals every slot
      d.W = share_W; d.grid = share_grid; d.n_slots = d.share_first + (n_blocks - d.share_first) * share_W;
    }
    // Cost-ordered hand-out (calls of more rows than workgroups whose slots the static round deals completely -- one to TEAMS rows per
    // CU): the preparation kernel also writes K * dims of every row, and the teams take their rows by cost rank (irec_team.hip).
    // (measured, profiles/archive/r04w: pays on the two-team build with 20-beam steps: 342 blocks 0.467 -> 0.449 ms; not with 10-beam
    //  steps of half the length, nor on the three-team build.
    //  Round 6: encode_ten_kernel<2> with its two-row workgroups on the cheapest rows gains 5 us of 190 on one tensor of 302 blocks and loses
    //  1.5 - 11 us on calls of many tensors, whose layout lists the small blocks last anyway: not taken, profiles/r06end/.)
    if (!pl.lone && !pl.chunk && !(p->flags & IREC_FLAG_LISTED_ORDER) && n_blocks <= irec::COST_MAX_ROWS) {
      const int n_teams = irec::team_count_for(p->n_beams, p->n_samples, pl.shape);
      const int64_t tg = share_W >= 2 ? share_grid : batch_grid(n_blocks, std::min(pl.grid_cap / n_teams, n_cu));
      if (n_blocks > tg && d.n_slots <= tg * n_teams && n_teams == 2 && p->n_beams > 10 && irec::team_placeable(p->n_beams, p->n_samples, pl.shape)) d.placed = 1;
    }
  } else if (pl.table) {
    d.kind = 4;
    d.grid = (int)std::min<int64_t>(n_blocks, pl.one_grid_cap);
    int W = split_width(ctx, pl, p, n_blocks);
    if (const int wb = split_beam_width(p, W)) { W = wb; d.coop_beams = 1; }
    if (W >= 2) { d.W = W; d.grid = (int)(n_blocks * W); d.n_slots = n_blocks * W; }
  } else if (pl.fast) {
    d.kind = 5; d.grid = (int)std::min<int64_t>(n_blocks, pl.one_grid_cap);
  } else {
    d.kind = 6;
    d.grid = margins ? (int)std::min<int64_t>(n_blocks, std::min(pl.grid_cap, 2 * n_cu)) : (int)std::min<int64_t>(n_blocks, pl.grid_cap);
  }
  return d;
}

} // namespace

extern "C" {

size_t irec_encode_workspace_bytes(const irec_context *ctx, const irec_params *p, int32_t max_dim, int32_t max_K) {
  if (!ctx || check_params(p) != IREC_OK || max_dim < 1 || max_K < 0) return 0;
  const Plan pl = make_plan(ctx, p, max_dim, max_K);
  return plan_ws_bytes(pl);
}

// Scratch of ONE call (round 6; review r05: irec_encode_workspace_bytes sizes a call of blocks beyond 1024 dims for the whole device -- a slab
// per team slot, up to IREC_SLAB_BYTES_MAX = 16 GB at 55 MB per slab of 301 056 dims -- whatever the call's block count).  The chunked
// encoder indexes the slabs of the workgroups it launches: their count is what this call needs.  Every other plan: irec_encode_workspace_bytes.
size_t irec_encode_workspace_bytes_for(const irec_context *ctx, const irec_params *p, int64_t n_blocks, int32_t max_dim, int32_t max_K) {
  if (!ctx || check_params(p) != IREC_OK || max_dim < 1 || max_K < 0 || n_blocks < 0) return 0;
  irec_params pm = *p;
  if (pm.flags & IREC_FLAG_MARGINS) pm.flags |= IREC_FLAG_NO_SPLIT;
  Plan pl = make_plan(ctx, &pm, max_dim, max_K);
  if (pl.chunk && !(pm.flags & IREC_FLAG_MARGINS) && n_blocks > 0) {
    const int teams = std::max(1, irec::chunk_teams(pm.n_beams, pm.n_samples));
    int grid = chunk_grid(ctx, pl, &pm, n_blocks), ggrid = 0;
    int gteams = teams;
    if (gang_width(ctx, pl, &pm, n_blocks, max_dim, &ggrid, nullptr) >= 2) { grid = ggrid; gteams = std::max(1, irec::chunk_gang_teams(pm.n_beams, pm.n_samples)); }
    pl.grid_cap = std::min(pl.grid_cap, std::max(teams, grid * std::max(teams, gteams)));
  }
  return plan_ws_bytes(pl);
}

irec_status irec_encode_plan(const irec_context *ctx, const irec_params *p, int64_t n_blocks, int32_t max_block_dim,
                             int32_t max_K, irec_plan_info *out) {
  if (!ctx || !out) return fail(IREC_E_INVALID, "irec_encode_plan: null argument");
  if (irec_status s = check_params(p)) return s;
  if (n_blocks < 0 || max_block_dim < 1 || max_K < 0) return fail(IREC_E_INVALID, "irec_encode_plan: bad sizes");
  irec_params pm = *p;
  if (pm.flags & IREC_FLAG_MARGINS) { pm.flags |= IREC_FLAG_NO_SPLIT; p = &pm; }
  const Plan pl = make_plan(ctx, p, max_block_dim, max_K);
  const bool team = team_for_call(ctx, pl, p, n_blocks);
  std::memset(out, 0, sizeof(*out));
  const int B = p->n_beams, S = p->n_samples;
  if (p->flags & IREC_FLAG_MARGINS) {   // irec_beam_encode_ex: a margin build of the team encoder, or the generic kernel
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const int mshape = margin_team_shape(ctx, pl, p, n_blocks);
    if (mshape >= 0) {
      const int n_teams = irec::team_count_for(B, S, mshape);
      std::string nm = irec::team_kernel_name(B, S, mshape);
      nm.insert(nm.size() - 1, ",margins");
      std::snprintf(out->kernel, sizeof out->kernel, "%s", nm.c_str());
      std::snprintf(out->table_kernel, sizeof out->table_kernel, "prep_kernel (copy bits)");
      out->grid = batch_grid(n_blocks, std::min(pl.grid_cap / n_teams, n_cu));
      out->waves_per_wg = irec::team_waves_for(B, S, mshape);
      out->teams_per_wg = n_teams;
      out->lds_bytes = (int32_t)irec::team_lds_for(B, S, mshape);
      out->table_steps = pl.K_tab; out->n_tables = pl.n_tab; out->table_bytes = (int64_t)pl.tab_bytes;
    } else {
      std::snprintf(out->kernel, sizeof out->kernel, "encode_generic_kernel (margins)");
      out->grid = (int32_t)std::min<int64_t>(n_blocks, std::min(pl.grid_cap, 2 * n_cu));
      out->waves_per_wg = 4; out->teams_per_wg = 1;
      out->lds_bytes = (int32_t)irec::generic_lds_bytes();
    }
    out->n_cu = ctx->n_cu; out->clock_mhz = ctx->clock_mhz;
    out->workspace_bytes = (int64_t)plan_ws_bytes(pl);
    return IREC_OK;
  }
  if (pl.chunk) {
    std::snprintf(out->kernel, sizeof out->kernel, "%s", irec::chunk_kernel_name(B, S));
    std::snprintf(out->table_kernel, sizeof out->table_kernel, "prep_kernel (copy bits)");
    out->grid = chunk_grid(ctx, pl, p, n_blocks);
    {
      int ggrid = 0;
      out->split = gang_width(ctx, pl, p, n_blocks, max_block_dim, &ggrid, nullptr);   // teams that code each block together
      if (out->split >= 2) out->grid = ggrid;
    }
    out->waves_per_wg = irec::chunk_teams(B, S) * 4;
    out->teams_per_wg = irec::chunk_teams(B, S);
    out->lds_bytes = (int32_t)irec::chunk_lds_for(B, S);
    if (out->split >= 2) {
      std::snprintf(out->kernel, sizeof out->kernel, "%s", irec::chunk_gang_kernel_name(B, S));
      out->teams_per_wg = irec::chunk_gang_teams(B, S); out->waves_per_wg = out->teams_per_wg * 4;
      out->lds_bytes = (int32_t)irec::chunk_gang_lds_for(B, S);
    }
  } else if (team && pl.lone) {
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    std::snprintf(out->kernel, sizeof out->kernel, "%s", irec::lone_kernel_name());
    std::snprintf(out->table_kernel, sizeof out->table_kernel, "prep_kernel (copy bits)");
    out->grid = batch_grid(n_blocks, n_cu);
    out->waves_per_wg = irec::lone_waves();
    out->teams_per_wg = irec::lone_waves();       // every wave codes its own block
    out->lds_bytes = (int32_t)irec::lone_lds_bytes();
  } else if (team) {
    const int shape = shape_for_call(ctx, pl, p, n_blocks);
    const int n_teams = irec::team_count_for(B, S, shape);
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    std::snprintf(out->kernel, sizeof out->kernel, "%s", irec::team_kernel_name(B, S, shape));
    std::snprintf(out->table_kernel, sizeof out->table_kernel, "prep_kernel (copy bits)");
    out->grid = batch_grid(n_blocks, std::min(pl.grid_cap / n_teams, n_cu));
    {
      int sgrid = 0;
      out->split = team_share_width(ctx, pl, p, n_blocks, shape, nullptr, &sgrid);   // teams that code each shared row
      if (out->split >= 2) out->grid = sgrid;
    }
    out->waves_per_wg = irec::team_waves_for(B, S, shape);
    out->teams_per_wg = n_teams;
    out->lds_bytes = (int32_t)irec::team_lds_for(B, S, shape);
    if (const int tt = irec::team_ten_teams(B, S, shape); tt && out->split < 2 && !(p->flags & IREC_FLAG_NO_TEN)) {   // plain call of at most ten beams
      std::snprintf(out->kernel, sizeof out->kernel, "encode_ten_kernel<%d>", tt);
      out->lds_bytes = (int32_t)irec::ten_lds_for(tt);
      out->teams_per_wg = tt; out->waves_per_wg = 4 * tt;
      out->grid = batch_grid(n_blocks, std::min(pl.grid_cap / tt, n_cu));
    }
  } else if (pl.fast) {
    std::snprintf(out->kernel, sizeof out->kernel, "%s", irec::fast_kernel_name(B, S, pl.table));
    if (pl.table) std::snprintf(out->table_kernel, sizeof out->table_kernel, "prep_kernel (plain rows)");
    out->grid = (int32_t)std::min<int64_t>(n_blocks, pl.one_grid_cap);
    out->split = split_width(ctx, pl, p, n_blocks);
    if (const int wb = split_beam_width(p, out->split)) { out->split = wb; out->split_beams = 1; }
    if (out->split >= 2) {
      out->grid = (int32_t)(n_blocks * out->split);
      // (the split forms are builds of their own: <NB,4,true,1> shares a block's samples, <NB,4|8,true,2> its beams -- 8 waves for 20 beams)
      std::snprintf(out->kernel, sizeof out->kernel, "encode_fast_kernel<%d,%d,true,%d>", irec::fast_nb_for(B),
                    out->split_beams ? irec::fast_split_beam_waves(B) : 4, out->split_beams ? 2 : 1);
    }
    out->waves_per_wg = (out->split >= 2 && out->split_beams) ? irec::fast_split_beam_waves(B) : irec::fast_waves_for(B, S, pl.table);
    out->teams_per_wg = 1;
    out->lds_bytes = (int32_t)irec::fast_lds_for(B, S, pl.table) + (out->split >= 2 && out->split_beams ? 40024 : 0);   // (beam split: two table copies of 10 006 floats)
  } else {
    std::snprintf(out->kernel, sizeof out->kernel, "encode_generic_kernel");
    out->grid = (int32_t)std::min<int64_t>(n_blocks, pl.grid_cap);
    out->waves_per_wg = 4;
    out->teams_per_wg = 1;
    out->lds_bytes = (int32_t)irec::generic_lds_bytes();
  }
  out->table_steps = pl.K_tab;
  out->n_tables = pl.n_tab;
  out->n_cu = ctx->n_cu;
  out->clock_mhz = ctx->clock_mhz;
  out->table_bytes = (int64_t)pl.tab_bytes;
  out->workspace_bytes = (int64_t)plan_ws_bytes(pl);
  return IREC_OK;
}

// Test hook (csrc/irec_internal.h): the plan and the launch numbers of a call on a device of `n_cu` compute units -- no device is touched.
irec_status irec_test_plan(int32_t n_cu, int32_t clock_mhz, const irec_params *p, int64_t n_blocks, int32_t max_block_dim, int32_t max_K,
                           irec_plan_info *info, irec_plan_detail *detail) {
  if (!info || !detail || n_cu < 1) return fail(IREC_E_INVALID, "irec_test_plan: bad argument");
  irec_context fake;
  fake.n_cu = n_cu; fake.clock_mhz = clock_mhz;
  if (irec_status st = irec_encode_plan(&fake, p, n_blocks, max_block_dim, max_K, info)) return st;
  const bool margins = (p->flags & IREC_FLAG_MARGINS) != 0;
  irec_params pm = *p;
  if (margins) pm.flags |= IREC_FLAG_NO_SPLIT;
  Plan pl = make_plan(&fake, &pm, max_block_dim, max_K);   // ... and then exactly what irec_beam_encode_ex does with it
  const int mshape = margins ? margin_team_shape(&fake, pl, &pm, n_blocks) : -1;
  if (margins && mshape < 0) { pl.table = false; pl.team = false; pl.lone = false; pl.chunk = false; pl.fast = false; pl.team_only = false; pl.n_tab = 0; pl.K_tab = 0; }
  pl.team = mshape >= 0 ? true : team_for_call(&fake, pl, &pm, n_blocks);
  if (mshape >= 0) pl.shape = mshape;
  else if (pl.team) pl.shape = shape_for_call(&fake, pl, &pm, n_blocks);
  const CallDetail cd = call_detail(&fake, pl, &pm, n_blocks, max_block_dim, margins);
  std::memset(detail, 0, sizeof(*detail));
  detail->kind = cd.kind; detail->grid = cd.grid; detail->teams_per_wg = cd.teams; detail->coop_width = cd.W; detail->coop_beams = cd.coop_beams;
  detail->gang_chunks = cd.gang_chunks; detail->placed = cd.placed; detail->split_blocks = cd.split_blocks;
  detail->share_first = cd.share_first; detail->n_slots = cd.n_slots;
  detail->slabs_in_workspace = pl.grid_cap; detail->slab_bytes = (int64_t)pl.ws_per_wg;
  detail->fixed_bytes = (int64_t)(irec::WS_HEAD_BYTES + pl.tab_bytes + pl.gang_bytes);
  detail->exchange_rows = irec::COOP_MAX_BLOCKS; detail->exchange_keys = irec::COOP_KEYS;
  return IREC_OK;
}

irec_status irec_block_kl(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                          const int32_t *block_pos, const int32_t *block_dim, const int32_t *perm, const float *q_loc,
                          const float *q_scale, const float *p_loc, const float *p_scale, float *out_kl, int32_t *out_K,
                          void *hip_stream) {
  if (!ctx) return fail(IREC_E_INVALID, "irec_block_kl: null context");
  if (irec_status s = check_params(p)) return s;
  if (n_blocks < 0) return fail(IREC_E_INVALID, "irec_block_kl: n_blocks < 0");
  if (n_blocks == 0) return IREC_OK;
  if (!block_base || !block_pos || !block_dim || !q_loc || !q_scale || !p_loc || !p_scale || !out_K)
    return fail(IREC_E_INVALID, "irec_block_kl: null pointer argument");
  IREC_ON_DEVICE(ctx->device);
  irec::EncArgs A{};
  A.block_base = block_base; A.block_pos = block_pos; A.block_dim = block_dim; A.perm = perm;
  A.q_loc = q_loc; A.q_scale = q_scale; A.p_loc = p_loc; A.p_scale = p_scale;
  A.n_blocks = n_blocks; A.omega = p->kl_per_partition; A.S = p->n_samples; A.B = p->n_beams; A.out_K = out_K;
  A.K_limit = ctx->n_rho;
  const int grid = (int)std::min<int64_t>(n_blocks, 8LL * ctx->n_cu);
  HIP_TRY(irec::launch_block_kl(A, out_kl, grid, (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_beam_encode(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                             const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                             const int32_t *perm, const float *q_loc, const float *q_scale, const float *p_loc,
                             const float *p_scale, int64_t seed, int32_t max_K, int32_t *out_K, int32_t *out_indices,
                             float *out_sample, void *workspace, size_t workspace_bytes, void *hip_stream) {
  if (p && (p->flags & IREC_FLAG_MARGINS)) return fail(IREC_E_INVALID, "irec_beam_encode: IREC_FLAG_MARGINS needs irec_beam_encode_ex (out_margin)");
  return irec_beam_encode_ex(ctx, p, n_blocks, block_base, block_pos, block_dim, max_block_dim, perm, q_loc, q_scale, p_loc, p_scale, seed,
                             max_K, out_K, out_indices, out_sample, nullptr, workspace, workspace_bytes, hip_stream);
}

irec_status irec_beam_encode_ex(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                                const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                                const int32_t *perm, const float *q_loc, const float *q_scale, const float *p_loc,
                                const float *p_scale, int64_t seed, int32_t max_K, int32_t *out_K, int32_t *out_indices,
                                float *out_sample, float *out_margin, void *workspace, size_t workspace_bytes, void *hip_stream) {
  if (!ctx) return fail(IREC_E_INVALID, "irec_beam_encode: null context");
  if (irec_status s = check_params(p)) return s;
  if (((p->flags & IREC_FLAG_MARGINS) != 0) != (out_margin != nullptr))
    return fail(IREC_E_INVALID, "irec_beam_encode_ex: out_margin and IREC_FLAG_MARGINS go together (the flag sizes the workspace)");
  irec_params pm = *p;
  if (out_margin) { pm.flags |= IREC_FLAG_NO_SPLIT; p = &pm; }   // (no block is shared under the flag: the cooperative forms have no margin builds)
  if (n_blocks < 0) return fail(IREC_E_INVALID, "irec_beam_encode: n_blocks < 0");
  if (n_blocks > 0x7FFF0000ll) return fail(IREC_E_INVALID, "irec_beam_encode: more than 2^31 - 65536 blocks in one call");   // (block rows are int32 in the kernels)
  if (n_blocks == 0) return IREC_OK;
  if (!block_base || !block_pos || !block_dim || !q_loc || !q_scale || !p_loc || !p_scale || !out_K || !out_sample)
    return fail(IREC_E_INVALID, "irec_beam_encode: null pointer argument");
  if (max_K < 0 || max_K > IREC_MAX_PARTITIONS) return fail(IREC_E_INVALID, "irec_beam_encode: max_K %d out of range", max_K);
  if (max_K > 0 && !out_indices) return fail(IREC_E_INVALID, "irec_beam_encode: null out_indices");
  if (max_block_dim < 1 || max_block_dim > (1 << 22)) return fail(IREC_E_INVALID, "irec_beam_encode: max_block_dim %d out of range", max_block_dim);
  Plan pl = make_plan(ctx, p, max_block_dim, max_K);
  size_t need = plan_ws_bytes(pl);
  if (workspace && workspace_bytes < need && pl.chunk && !out_margin) {
    // blocks beyond 1024 dims: a workspace sized for THIS call (irec_encode_workspace_bytes_for) -- or any size in between -- holds fewer
    // slabs than the device has team slots; the call then launches no more teams than it has slabs (smaller gangs, same results)
    const int teams = std::max(std::max(1, irec::chunk_teams(p->n_beams, p->n_samples)), std::max(1, irec::chunk_gang_teams(p->n_beams, p->n_samples)));
    const size_t fixed = irec::WS_HEAD_BYTES + pl.tab_bytes + pl.gang_bytes;
    const size_t fit = workspace_bytes > fixed ? (workspace_bytes - fixed) / pl.ws_per_wg : 0;
    if (fit >= (size_t)teams) { pl.grid_cap = (int)std::min<size_t>((size_t)pl.grid_cap, fit / teams * teams); need = plan_ws_bytes(pl); }
  }
  if (!workspace || workspace_bytes < need)
    return fail(IREC_E_WORKSPACE, "irec_beam_encode: workspace %zu bytes < required %zu", workspace_bytes, need);
  // top-B margins: a margin build of the team encoder where one serves the call's shape (whatever the call's size), else the generic
  // kernel, which draws in the kernel: no tables
  const int mshape = out_margin ? margin_team_shape(ctx, pl, p, n_blocks) : -1;
  if (out_margin && mshape < 0) { pl.table = false; pl.team = false; pl.lone = false; pl.chunk = false; pl.fast = false; pl.team_only = false; pl.n_tab = 0; pl.K_tab = 0; }
  pl.team = mshape >= 0 ? true : team_for_call(ctx, pl, p, n_blocks);
  if (((uintptr_t)workspace & 255) != 0) return fail(IREC_E_WORKSPACE, "irec_beam_encode: workspace must be 256-byte aligned");
  IREC_ON_DEVICE(ctx->device);
  hipStream_t st = (hipStream_t)hip_stream;
  irec::EncArgs A{};
  A.block_base = block_base; A.block_pos = block_pos; A.block_dim = block_dim; A.perm = perm;
  A.q_loc = q_loc; A.q_scale = q_scale; A.p_loc = p_loc; A.p_scale = p_scale;
  A.n_blocks = n_blocks; A.seed = seed;
  A.omega = p->kl_per_partition; A.S = p->n_samples; A.B = p->n_beams; A.max_K = max_K;
  A.K_limit = ctx->n_rho;
  A.out_K = out_K; A.out_indices = out_indices; A.out_sample = out_sample;
  A.lut = ctx->d_lut; A.lut2 = ctx->d_lut2; A.dlog4r = ctx->d_dlog4r; A.rho = ctx->d_rho;
  // workspace head: counter block (WS_COUNTER_BYTES, zeroed per call: [0] block counter of the first pass, [1] deferred-block
  // count, [2] block counter of the deferred pass, [3] split-encoder error flag, [64..127] its per-block arrival counters,
  // [128 + 64 x] the block counter of XCD x), then the candidate exchange of the split encoder
  A.counter = (unsigned int *)workspace;
  A.xcd_counter = (unsigned int *)workspace + irec::WS_XCD_WORD;
  A.defer_count = (unsigned int *)workspace + 1;
  if (mshape >= 0) pl.shape = mshape;
  else if (pl.team) pl.shape = shape_for_call(ctx, pl, p, n_blocks);
  A.K_tab = pl.K_tab; A.deferred_pass = 0; A.shape_override = pl.shape; A.no_ten = (p->flags & IREC_FLAG_NO_TEN) ? 1 : 0;
  A.coop_W = 1; A.coop_err = (unsigned int *)workspace + 3; A.coop_arrive = (unsigned int *)workspace + 64;
  A.coop_xch = (uint32_t *)((char *)workspace + irec::WS_COUNTER_BYTES);
  A.ws = (char *)workspace + irec::WS_HEAD_BYTES + pl.tab_bytes;
  A.ws_per_wg = pl.ws_per_wg;
  A.max_dim_pad = pl.dpad;
  A.out_margin = out_margin;
  // table bookkeeping (IREC_FLAG_REUSE_TABLES): the key of every proposal table this call needs -- what it is a function
  // of (seed, S, D, window), which kernel writes it (the team encoder's rows carry copy bits) and where it lies -- is
  // compared with the slot's stamp ON THE DEVICE by the preparation kernel; a slot the call does not use is stamped with zeros,
  // so a call without tables (whose slabs may lie over the table area) invalidates what was there
  irec::TableStamps stamps{};
  if (pl.table)
    for (int q = 0; q < pl.n_tab; ++q) {
      uint32_t *w = stamps.w[q];
      w[0] = 0x7ab1e000u | ((pl.team || pl.chunk) ? 1u : 2u);   // (rows with copy bits / plain byte offsets)
      w[1] = (uint32_t)(uint64_t)seed; w[2] = (uint32_t)((uint64_t)seed >> 32);
      w[3] = (uint32_t)p->n_samples; w[4] = (uint32_t)pl.tab_dim[q]; w[5] = (uint32_t)pl.K_tab;
      w[6] = (uint32_t)pl.tab_off[q];
      w[7] = ~(w[1] ^ w[2] ^ w[3] ^ w[4] ^ w[5] ^ w[6]);
    }
  stamps.reuse = (p->flags & IREC_FLAG_REUSE_TABLES) ? 1 : 0;
  // the call in numbers (grid, teams per workgroup, cooperative width, cost-ordered hand-out): call_detail, also behind irec_test_plan
  const CallDetail cd = call_detail(ctx, pl, p, n_blocks, max_block_dim, out_margin != nullptr);
  const int split_blocks = cd.split_blocks;   // a cooperative call: the preparation kernel also zeroes the exchange granules of its blocks
  if (cd.placed) A.row_cost = (const uint32_t *)((char *)workspace + irec::WS_COUNTER_BYTES + irec::WS_XCH_BYTES);
  int grid = (int)std::min<int64_t>(n_blocks, pl.one_grid_cap);
  A.dbg = nullptr;
#ifdef IREC_HOST_STAMPS
  A.dbg = ctx->d_dbg;
  if (ctx->d_dbg) HIP_TRY(hipMemsetAsync(ctx->d_dbg, 0, 4096 * 16 * sizeof(unsigned long long), st));
#endif
  // ONE preparation kernel per call: books, exchange granules, row costs and the proposal tables (irec_kernels.h)
  irec::PrepArgs P{};
  P.head = (uint32_t *)workspace; P.ts = stamps;
  P.n_granule = irec::IREC_COOP_GRANULES_ON && split_blocks > 0 ? std::min(split_blocks, irec::COOP_MAX_BLOCKS) : 0;
  P.n_cost = A.row_cost ? (int32_t)n_blocks : 0;
  P.cost = const_cast<uint32_t *>(A.row_cost);
  P.seed = seed; P.S = p->n_samples; P.K_tab = pl.K_tab; P.dlog4r = ctx->d_dlog4r;
  A.ws_head = (uint32_t *)workspace;
  for (int q = 0; q < 4; ++q) { A.tab[q] = nullptr; A.tab_dim[q] = -1; }
  if (pl.table) {
    for (int q = 0; q < pl.n_tab; ++q) {
      uint16_t *tab = (uint16_t *)((char *)workspace + irec::WS_HEAD_BYTES + pl.tab_off[q]);
      A.tab[q] = tab; A.tab_dim[q] = pl.tab_dim[q];
      P.jobs.tab[q] = tab; P.jobs.keep[q] = nullptr;
    }
    if (pl.n_tab > 0 && !(p->flags & IREC_FLAG_TABLES_PRESENT)) {   // (TABLES_PRESENT: the caller's previous call on this workspace built exactly these)
      P.table_kind = (pl.team || pl.chunk) ? 1 : 2;                  // rows with copy bits / plain rows
      P.n_table_wgs = (int32_t)irec::prep_table_wgs(P.table_kind, p->n_samples, pl.K_tab, pl.n_tab, pl.tab_dim, &P.jobs);
    }
  }
  HIP_TRY(irec::launch_prep(P, A, st));
  if (pl.table) {
    // second pass (only when the window is shorter than max_K): the fused-Philox encoder codes the blocks whose K lies
    // beyond the table window; it returns at once when the first pass deferred nothing
    auto deferred_pass = [&]() -> irec_status {
      if (pl.K_tab >= max_K) return IREC_OK;
      irec::EncArgs A2 = A;
      A2.deferred_pass = 1; A2.coop_W = 1;
      A2.counter = (unsigned int *)workspace + 2;
      for (int q = 0; q < 4; ++q) { A2.tab[q] = nullptr; A2.tab_dim[q] = -1; }
      if (out_margin) HIP_TRY(irec::launch_encode_generic(A2, (int)std::min<int64_t>(n_blocks, std::min(pl.grid_cap, 2 * (ctx->n_cu > 0 ? ctx->n_cu : 256))), st));
      else if (pl.team_only || pl.chunk) HIP_TRY(irec::launch_encode_generic(A2, (int)std::min<int64_t>(n_blocks, pl.fast_grid_cap), st));
      else HIP_TRY(irec::launch_encode_fast(A2, false, (int)std::min<int64_t>(n_blocks, pl.fast_grid_cap), st));
      return IREC_OK;
    };
    if (pl.chunk) {   // one workgroup per CU, a block of any dim count per team; steps beyond the table window are drawn in the kernel: no second pass
      const int cgrid = cd.grid;
      if (cd.W >= 2) {
        A.coop_W = cd.W; A.gang_chunks = cd.gang_chunks;
        A.gang_xch = (char *)workspace + irec::WS_HEAD_BYTES + pl.tab_bytes + (size_t)pl.grid_cap * pl.ws_per_wg;
        A.gang_stride = pl.gang_stride;
        A.coop_test_orphan = (p->flags & IREC_FLAG_TEST_SPLIT_ORPHAN) ? 1 : 0;
      }
      HIP_TRY(irec::launch_encode_chunk(A, cgrid, st));
    } else if (pl.team && pl.lone) { // one workgroup per CU, a block per wave
      HIP_TRY(irec::launch_encode_lone(A, cd.grid, st));
      if (irec_status s2 = deferred_pass()) return s2;
#ifdef IREC_HOST_STAMPS
      if (ctx->d_dbg) {   // diagnostic build (-DIREC_LONE_STAMPS): per-wave phase cycles of the one-beam encoder
        const int lgrid = batch_grid(n_blocks, ctx->n_cu > 0 ? ctx->n_cu : 256), nwv = irec::lone_waves();
        std::vector<unsigned long long> h((size_t)lgrid * nwv * 16);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), ctx->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        static const char *nm[6] = {"prologue (fetch, KL)", "step constants", "scoring", "selection", "update", "epilogue"};
        double sum[6] = {0}, tot = 0;
        for (size_t w = 0; w < (size_t)lgrid * nwv; ++w) for (int k = 0; k < 6; ++k) { sum[k] += (double)h[w * 16 + k]; tot += (double)h[w * 16 + k]; }
        fprintf(stderr, "[irec lone stamps] share of wave time:\n");
        for (int k = 0; k < 6; ++k) fprintf(stderr, "  %-22s %5.1f%%\n", nm[k], 100 * sum[k] / tot);
        fprintf(stderr, "  cycles per wave: %.0f\n", tot / (lgrid * nwv));
        return IREC_OK;
      }
#endif
    } else if (pl.team) { // grid_cap counts teams (= scratch slabs): two per workgroup, one workgroup per CU
      const int tgrid = cd.grid;
#ifdef IREC_HOST_STAMPS
      const int n_teams = cd.teams;
#endif
      if (cd.W >= 2) {   // rows [share_first, n_blocks) are coded by W teams each; the static round deals every slot
        A.coop_W = cd.W; A.tsplit_first = cd.share_first;
        A.coop_test_orphan = (p->flags & IREC_FLAG_TEST_SPLIT_ORPHAN) ? 1 : 0;
      }
      if (out_margin) HIP_TRY(irec::launch_encode_team_margin(A, tgrid, st));
      else HIP_TRY(irec::launch_encode_team(A, tgrid, st));
#ifndef IREC_HOST_STAMPS
      if (irec_status s2 = deferred_pass()) return s2;
#else
      if (!ctx->d_dbg) { if (irec_status s2 = deferred_pass()) return s2; }
      if (ctx->d_dbg) { // diagnostic build (-DIREC_TEAM_STAMPS) only: per-wave phase cycles, wave 0 of a team vs the others
        const bool ten_call = !cd.W && !out_margin && !(p->flags & IREC_FLAG_NO_TEN) && irec::team_ten_teams(p->n_beams, p->n_samples, pl.shape) != 0;
        const int nwv = ten_call ? 4 * n_teams : irec::team_waves_for(p->n_beams, p->n_samples, pl.shape);
        std::vector<unsigned long long> h((size_t)tgrid * nwv * 16);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), ctx->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        static const char *nm[12] = {"fetch", "prologue", "scoring", "wait-score", "combine", "select", "update-tail", "wait-update", "epilogue",
                                     "upd:sel+issue+consts", "upd:batches", "-"};
        // (waves of teams that coded no block -- a mid-size call leaves team slots empty -- are left out of the averages)
        const int nwt = nwv / n_teams;   // waves per team
        double w0[12] = {0}, wo[12] = {0}, t0 = 0, to = 0;
        int n0 = 0, no = 0;
        for (int w = 0; w < tgrid * nwv; ++w) {
          if (h[(size_t)w * 16 + 1] == 0ull) continue;   // no prologue: no block
          if ((w % nwt) == 0) ++n0; else ++no;
          for (int k = 0; k < 12; ++k) { const double v = (double)h[(size_t)w * 16 + k]; if ((w % nwt) == 0) { w0[k] += v; t0 += v; } else { wo[k] += v; to += v; } }
        }
        fprintf(stderr, "[irec team stamps] %d of %d teams coded blocks; share of wave time (cycles per wave), wave 0 of a team | the others:\n", n0, tgrid * n_teams);
        t0 -= w0[11]; to -= wo[11];   // (slot 11 counts block-steps, not cycles)
        for (int k = 0; k < 11; ++k) fprintf(stderr, "  %-20s %5.1f%% (%8.0f) | %5.1f%% (%8.0f)\n", nm[k], 100 * w0[k] / t0, w0[k] / (n0 ? n0 : 1), 100 * wo[k] / to, wo[k] / (no ? no : 1));
        fprintf(stderr, "  cycles per wave: %.0f | %.0f;  block-steps per wave: %.2f;  cycles per block-step: %.0f\n", t0 / (n0 ? n0 : 1), to / (no ? no : 1),
                w0[11] / (n0 ? n0 : 1), w0[11] > 0 ? t0 / w0[11] : 0.0);
        {   // idle tail of the persistent grid: wave-time between a wave's exit and the last wave's (slot 12 = a wave's lifetime)
          double life_max = 0, life_sum = 0, life_min = 1e300;
          int nlife = 0;
          for (int w = 0; w < tgrid * nwv; ++w) {
            const double v = (double)h[(size_t)w * 16 + 12];
            if (v <= 0) continue;
            life_max = std::max(life_max, v); life_min = std::min(life_min, v); life_sum += v; ++nlife;
          }
          if (nlife) fprintf(stderr, "  idle tail: %.1f%% of the grid's wave-time lies behind a wave's exit (first exit at %.1f%% of the longest lifetime)\n",
                             100.0 * (1.0 - life_sum / (nlife * life_max)), 100.0 * life_min / life_max);
        }
        if (h[13]) fprintf(stderr, "  shader clock of workgroup 0, wave 0: %.0f MHz over %.3f ms (s_memtime / s_memrealtime)\n",
                           100.0 * (double)h[12] / (double)h[13], (double)h[13] / 1e5);
        return IREC_OK;
      }
#endif
    } else {
      const int W = cd.W;
      A.coop_beams = cd.coop_beams;
      if (W >= 2) {
        A.coop_W = W;
        A.coop_test_orphan = (p->flags & IREC_FLAG_TEST_SPLIT_ORPHAN) ? 1 : 0;
        HIP_TRY(irec::launch_encode_fast(A, true, cd.grid, st));
#ifdef IREC_HOST_STAMPS
        if (ctx->d_dbg) grid = cd.grid;   // (diagnostics below: the stamps of every workgroup)
#endif
      } else HIP_TRY(irec::launch_encode_fast(A, true, grid, st));
      if (irec_status s2 = deferred_pass()) return s2;
    }
  } else if (pl.fast) {
    HIP_TRY(irec::launch_encode_fast(A, false, grid, st));
  } else {
    if (out_margin) grid = (int)std::min<int64_t>(n_blocks, std::min(pl.grid_cap, 2 * (ctx->n_cu > 0 ? ctx->n_cu : 256)));
    HIP_TRY(irec::launch_encode_generic(A, grid, st));
  }
#ifdef IREC_HOST_STAMPS
  if (ctx->d_dbg) { // diagnostic build only: synchronous read-back of the phase stamps
    std::vector<unsigned long long> h((size_t)grid * 16);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(h.data(), ctx->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[16] = {0};
    for (int w = 0; w < grid; ++w) for (int k = 0; k < 16; ++k) sum[k] += (double)h[(size_t)w * 16 + k];
    const double tot = sum[0] + sum[1] + sum[2] + sum[3];
    {
      int zero = 0; unsigned long long t0 = ~0ull, late = 0;
      for (int w = 0; w < grid; ++w) { if (h[(size_t)w * 16 + 4] == 0) ++zero; else if (h[(size_t)w * 16 + 5] < t0) t0 = h[(size_t)w * 16 + 5]; }
      for (int w = 0; w < grid; ++w) if (h[(size_t)w * 16 + 4] && h[(size_t)w * 16 + 5] - t0 > 100000ull) ++late; // > 1 ms at 100 MHz
      fprintf(stderr, "[irec stamps] census: %d of %d workgroups coded no block, %llu started > 1 ms after the first\n", zero, grid, late);
    }
    fprintf(stderr, "[irec stamps] top-B detail cycles/WG: combine %.0f, wait-for-keys barrier %.0f, wave-0 select %.0f, closing barrier %.0f\n",
            sum[11] / grid, sum[8] / grid, sum[9] / grid, sum[10] / grid);
    if (sum[12] + sum[13] + sum[14] > 0)
      fprintf(stderr, "[irec stamps] split exchange cycles/WG: publish %.0f, wait for partners %.0f, read back %.0f\n",
              sum[12] / grid, sum[13] / grid, sum[14] / grid);
    fprintf(stderr, "[irec stamps] waves/workgroup %d, LDS %zu B\n",
            irec::fast_waves_for(p->n_beams, p->n_samples, pl.table), irec::fast_lds_for(p->n_beams, p->n_samples, pl.table));
    fprintf(stderr, "[irec stamps] %s grid=%d cycles/WG: prologue %.0f (%.1f%%) scoring %.0f (%.1f%%) select %.0f (%.1f%%) update %.0f (%.1f%%)\n",
            pl.table ? "table" : pl.fast ? "fused" : "generic", grid, sum[0] / grid, 100 * sum[0] / tot, sum[1] / grid,
            100 * sum[1] / tot, sum[2] / grid, 100 * sum[2] / tot, sum[3] / grid, 100 * sum[3] / tot);
  }
#endif
  return IREC_OK;
}

// Table window of a decode call: the proposal tables pay when the call's blocks share rows (S * K_tab rows per dim count
// against n_blocks * K row reads) and fit the scratch.
struct DecPlan { int upb; int K_tab; int n_tab; int tab_dim[4]; size_t tab_off[4]; size_t bytes; };
static DecPlan make_dec_plan(const irec_params *p, int64_t n_blocks, int32_t max_block_dim, int32_t max_K, bool have_ws) {
  DecPlan d{};
  int maxd = max_block_dim;
  for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) maxd = std::max(maxd, p->table_dims[q]);
  d.upb = maxd > 0 ? (maxd + 255) / 256 : 0;
  if (!have_ws || p->table_dims[0] <= 0 || max_K < 1 || (p->flags & IREC_FLAG_FUSED_PHILOX)) return d;
  if (n_blocks >= 0 && (int64_t)p->n_samples > 2 * n_blocks) return d;   // fewer row reads than rows: draw in the kernel
  size_t per_step = 0;
  for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) per_step += (size_t)p->n_samples * round_up(p->table_dims[q], 4) * 2;
  const int want = p->table_steps > 0 ? p->table_steps : IREC_TABLE_STEPS_DEFAULT;
  int kt = std::min(std::min(want, IREC_TABLE_STEPS_MAX), (int)max_K);
  kt = (int)std::min<size_t>((size_t)kt, table_steps_that_fit(per_step));
  if (kt < 1) return d;
  d.K_tab = kt;
  for (int q = 0; q < 4 && p->table_dims[q] > 0; ++q) {
    d.tab_dim[d.n_tab] = p->table_dims[q];
    d.tab_off[d.n_tab] = d.bytes;
    d.bytes += round_up_sz((size_t)kt * p->n_samples * round_up(p->table_dims[q], 4) * 2, 256);
    ++d.n_tab;
  }
  return d;
}

size_t irec_decode_workspace_bytes(const irec_context *ctx, const irec_params *p, int32_t max_K) {
  if (!ctx || check_params(p) != IREC_OK || max_K < 0) return 0;
  return make_dec_plan(p, -1, 0, max_K, true).bytes;
}

struct DecTensors { int64_t n_tensors; int32_t n, bs; const int32_t *block_row; };
static irec_status beam_decode_impl(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                                    const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                                    const int32_t *perm, const float *p_loc, const float *p_scale, int64_t seed,
                                    int32_t max_K, const int32_t *K, const int32_t *indices, float *out_sample,
                                    void *workspace, size_t workspace_bytes, void *hip_stream, const DecTensors *tens = nullptr) {
  if (!ctx) return fail(IREC_E_INVALID, "irec_beam_decode: null context");
  if (irec_status s = check_params(p)) return s;
  if (n_blocks < 0) return fail(IREC_E_INVALID, "irec_beam_decode: n_blocks < 0");
  if (n_blocks == 0) return IREC_OK;
  if ((!tens && (!block_base || !block_pos || !block_dim)) || !p_loc || !p_scale || !K || !out_sample || (max_K > 0 && !indices))
    return fail(IREC_E_INVALID, "irec_beam_decode: null pointer argument");
  if (max_K < 0 || max_K > IREC_MAX_PARTITIONS) return fail(IREC_E_INVALID, "irec_beam_decode: max_K %d out of range", max_K);
  if (max_block_dim < 0 || max_block_dim > (1 << 22)) return fail(IREC_E_INVALID, "irec_beam_decode: max_block_dim %d out of range", max_block_dim);
  const DecPlan dp = make_dec_plan(p, n_blocks, max_block_dim, max_K, workspace != nullptr);
  if (dp.K_tab > 0) {
    if (workspace_bytes < dp.bytes) return fail(IREC_E_WORKSPACE, "irec_beam_decode_ws: workspace %zu bytes < required %zu", workspace_bytes, dp.bytes);
    if (((uintptr_t)workspace & 255) != 0) return fail(IREC_E_WORKSPACE, "irec_beam_decode_ws: workspace must be 256-byte aligned");
  }
  IREC_ON_DEVICE(ctx->device);
  hipStream_t st = (hipStream_t)hip_stream;
  irec::DecArgs A{};
  A.block_base = block_base; A.block_pos = block_pos; A.block_dim = block_dim; A.perm = perm;
  A.p_loc = p_loc; A.p_scale = p_scale; A.n_blocks = n_blocks; A.seed = seed; A.max_K = max_K; A.K = K;
  A.indices = indices; A.out_sample = out_sample; A.lut = ctx->d_lut; A.rho = ctx->d_rho;
  A.K_limit = ctx->n_rho;
  A.upb = dp.upb; A.S = p->n_samples; A.K_tab = dp.K_tab; A.lut2 = ctx->d_lut2; A.dlog4r = ctx->d_dlog4r;
  for (int q = 0; q < 4; ++q) { A.tab[q] = nullptr; A.tab_dim[q] = -1; }
  for (int q = 0; q < dp.n_tab && dp.K_tab > 0; ++q) {
    uint16_t *tab = (uint16_t *)((char *)workspace + dp.tab_off[q]);
    HIP_TRY(irec::launch_alpha_table(seed, p->n_samples, dp.tab_dim[q], dp.K_tab, ctx->d_dlog4r, tab, nullptr, st));
    A.tab[q] = tab; A.tab_dim[q] = dp.tab_dim[q];
  }
  if (tens) {
    A.n_tensors = tens->n_tensors; A.tn = tens->n; A.tbs = tens->bs; A.tbpt = (tens->n + tens->bs - 1) / tens->bs;
    A.block_row = tens->block_row;
  }
  HIP_TRY(irec::launch_decode(A, ctx->n_cu > 0 ? ctx->n_cu : 256, st));
  return IREC_OK;
}

irec_status irec_beam_decode(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                             const int32_t *block_pos, const int32_t *block_dim, const int32_t *perm,
                             const float *p_loc, const float *p_scale, int64_t seed, int32_t max_K, const int32_t *K,
                             const int32_t *indices, float *out_sample, void *hip_stream) {
  return beam_decode_impl(ctx, p, n_blocks, block_base, block_pos, block_dim, 0, perm, p_loc, p_scale, seed, max_K, K, indices,
                          out_sample, nullptr, 0, hip_stream);
}

irec_status irec_beam_decode_ws(irec_context *ctx, const irec_params *p, int64_t n_blocks, const int64_t *block_base,
                                const int32_t *block_pos, const int32_t *block_dim, int32_t max_block_dim,
                                const int32_t *perm, const float *p_loc, const float *p_scale, int64_t seed, int32_t max_K,
                                const int32_t *K, const int32_t *indices, float *out_sample, void *workspace,
                                size_t workspace_bytes, void *hip_stream) {
  if (max_block_dim < 1) return fail(IREC_E_INVALID, "irec_beam_decode_ws: max_block_dim %d < 1", max_block_dim);
  return beam_decode_impl(ctx, p, n_blocks, block_base, block_pos, block_dim, max_block_dim, perm, p_loc, p_scale, seed, max_K, K,
                          indices, out_sample, workspace, workspace_bytes, hip_stream);
}

int32_t irec_decode_tensors_supported(const irec_params *p, int32_t tensor_dims, int32_t block_size) {
  if (check_params(p) != IREC_OK || tensor_dims < 1 || block_size < 1) return 0;
  block_size = std::min(block_size, tensor_dims);
  return irec::decode_tensor_waves(tensor_dims, block_size, true, nullptr) > 0 && irec::decode_tensor_waves(tensor_dims, block_size, false, nullptr) > 0;
}

irec_status irec_beam_decode_tensors(irec_context *ctx, const irec_params *p, int64_t n_tensors, int32_t tensor_dims,
                                     int32_t block_size, const int32_t *block_row, const int32_t *perm, const float *p_loc,
                                     const float *p_scale, int64_t seed, int32_t max_K, const int32_t *K,
                                     const int32_t *indices, float *out_sample, void *workspace, size_t workspace_bytes,
                                     void *hip_stream) {
  if (n_tensors < 0 || tensor_dims < 1 || block_size < 1) return fail(IREC_E_INVALID, "irec_beam_decode_tensors: bad sizes");
  if (!irec_decode_tensors_supported(p, tensor_dims, block_size))
    return fail(IREC_E_INVALID, "irec_beam_decode_tensors: tensors of %d dims in blocks of %d do not fit the staged decoder "
                                "(irec_decode_tensors_supported); use irec_beam_decode_ws", tensor_dims, block_size);
  block_size = std::min(block_size, tensor_dims);
  const int bpt = (tensor_dims + block_size - 1) / block_size;
  // the proposal tables cover the distinct block dims of Coder.split: block_size and the short last block
  irec_params q = *p;
  const int last = tensor_dims - (bpt - 1) * block_size;
  q.table_dims[0] = block_size;
  q.table_dims[1] = last != block_size ? last : 0;
  q.table_dims[2] = q.table_dims[3] = 0;
  const DecTensors tens{n_tensors, tensor_dims, block_size, block_row};
  return beam_decode_impl(ctx, &q, n_tensors * bpt, nullptr, nullptr, nullptr, block_size, perm, p_loc, p_scale, seed,
                          max_K, K, indices, out_sample, workspace, workspace_bytes, hip_stream, &tens);
}

irec_status irec_device_uniform_int(irec_context *ctx, int64_t seed, int64_t n, int32_t *out, void *hip_stream) {
  if (!ctx || n < 0 || (n > 0 && !out)) return fail(IREC_E_INVALID, "irec_device_uniform_int: bad arguments");
  if (n == 0) return IREC_OK;
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_uniform_int(seed, n, out, (hipStream_t)hip_stream));
  return IREC_OK;
}

// ---- hand-offs of the RVAE host shim (irec_shim.hip): device pointers, asynchronous on the stream ----
irec_status irec_shim_stats(irec_context *ctx, const float *y, const float *infer_heads, float *out, int32_t n_stats, int32_t n,
                            int32_t channels_y, int32_t channels_infer, int32_t stochastic, int32_t hw, const float *bias_y,
                            const float *bias_infer, void *hip_stream) {
  if (!ctx || !y || !out || (n_stats != 2 && n_stats != 4) || (n_stats == 4 && !infer_heads) || n < 1 || stochastic < 1 || hw < 1 ||
      channels_y < n_stats * stochastic || (n_stats == 4 && channels_infer < 2 * stochastic))
    return fail(IREC_E_INVALID, "irec_shim_stats: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_shim_stats(y, infer_heads, out, n_stats, n, channels_y, channels_infer, stochastic, hw, bias_y, bias_infer,
                                  (hipStream_t)hip_stream));
  return IREC_OK;
}
irec_status irec_shim_cat_elu(irec_context *ctx, const float *y, const float *latent, float *out, int32_t n, int32_t channels_y,
                              int32_t channel_offset, int32_t deterministic, int32_t stochastic, int32_t hw, const float *bias_y,
                              void *hip_stream) {
  if (!ctx || !y || !out || n < 1 || deterministic < 0 || stochastic < 0 || deterministic + stochastic < 1 || hw < 1 || channel_offset < 0 ||
      channel_offset + deterministic > channels_y || (stochastic > 0 && !latent))
    return fail(IREC_E_INVALID, "irec_shim_cat_elu: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_shim_cat_elu(y, latent, out, n, channels_y, channel_offset, deterministic, stochastic, hw, bias_y, (hipStream_t)hip_stream));
  return IREC_OK;
}
irec_status irec_shim_residual_elu(irec_context *ctx, const float *input, const float *tensor, float alpha, float *out, float *out_elu,
                                   int32_t n, int32_t channels, int32_t hw, const float *bias_tensor, void *hip_stream) {
  if (!ctx || !input || !tensor || !out || !out_elu || n < 1 || channels < 1 || hw < 1) return fail(IREC_E_INVALID, "irec_shim_residual_elu: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_shim_residual_elu(input, tensor, alpha, out, out_elu, (int64_t)n * channels * hw, bias_tensor, channels, hw,
                                         (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_test_decoder_sqrt(irec_context *ctx, uint64_t *out2, void *hip_stream) {
  if (!ctx || !out2) return fail(IREC_E_INVALID, "irec_test_decoder_sqrt: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_dec_sqrt_test(reinterpret_cast<unsigned long long *>(out2), (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_test_reduce_scatter(irec_context *ctx, const float *in, float *out, int32_t width, void *hip_stream) {
  if (!ctx || !in || !out || (width != 64 && width != 32 && width != 20 && width != 21 && width != 10)) return fail(IREC_E_INVALID, "irec_test_reduce_scatter: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_reduce_scatter_test(in, out, width, (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_test_select(irec_context *ctx, const float *scores, int32_t n, int32_t n_select, int32_t n_beams_cur,
                             uint32_t *scratch_keys, int32_t *out_sel, void *hip_stream) {
  if (!ctx || !scores || !scratch_keys || !out_sel || n < 1 || n_select < 1 || n_select > n || n_select > 64 ||
      n_beams_cur < 1)
    return fail(IREC_E_INVALID, "irec_test_select: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_select_test(scores, n, n_select, n_beams_cur, scratch_keys, out_sel, false, (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_test_select_quick(irec_context *ctx, const float *scores, int32_t n, int32_t n_select, int32_t n_beams_cur,
                                   uint32_t *scratch_keys, int32_t *out_sel, void *hip_stream) {
  if (!ctx || !scores || !scratch_keys || !out_sel || n < 1 || n_select < 1 || n_select > n || n_select > 64 ||
      n_beams_cur < 1)
    return fail(IREC_E_INVALID, "irec_test_select_quick: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_select_test(scores, n, n_select, n_beams_cur, scratch_keys, out_sel, true, (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_test_proposal_table(irec_context *ctx, int64_t seed, int32_t n_samples, int32_t dim, int32_t n_steps,
                                     uint16_t *out_tab, void *hip_stream) {
  if (!ctx || !out_tab || n_samples < 1 || dim < 1 || dim > irec::FAST_MAX_DIM || n_steps < 1)
    return fail(IREC_E_INVALID, "irec_test_proposal_table: bad arguments");
  IREC_ON_DEVICE(ctx->device);
  HIP_TRY(irec::launch_alpha_choice(seed, n_samples, dim, n_steps, ctx->d_dlog4r, out_tab, nullptr, (hipStream_t)hip_stream));
  return IREC_OK;
}

irec_status irec_device_tables(irec_context *ctx, const float **lut, const float **lut2, const uint16_t **dlog4r,
                               const float **rho) {
  if (!ctx) return fail(IREC_E_INVALID, "irec_device_tables: null context");
  if (lut) *lut = ctx->d_lut;
  if (lut2) *lut2 = ctx->d_lut2;
  if (dlog4r) *dlog4r = ctx->d_dlog4r;
  if (rho) *rho = ctx->d_rho;
  return IREC_OK;
}

} // extern "C"
