// div2_check.hip -- diagnostics only, and a negative result kept for the record (profiles/HISTORY.md, round 3): a step's
// constants cost six IEEE divisions and a square root per dim, the floor of every small-S call.  The compiler expands `n / d`
// into v_div_scale_f32 x 2, v_rcp_f32, four v_fma_f32 and a v_mul_f32 of Newton refinement, v_div_fmas_f32, v_div_fixup_f32;
// div2_ieee below is that sequence for two quotients with the five refinement operations packed (v_pk_fma_f32 /
// v_pk_mul_f32) -- 16 instructions instead of 22.  It IS bit-identical (0 mismatches below), but only 1.08x faster
// (profiles/r03l/div2_check.log), i.e. < 3 % of the constants: not adopted.  Checked against the compiler's `/`, bit for bit, over
//   * uniformly random bit patterns (NaN, infinities, denormals, every exponent pair),
//   * operands of moderate size (the range the coder's variances live in), and
//   * every pair out of a list of edge values;
// then the rate of both forms.  Prints the number of mismatches (NaN against NaN counts as equal) -- must be 0.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../relative-entropy-coding_amd/csrc -I../../include -o div2_check div2_check.hip && ./div2_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

namespace irec {
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2v div2_ieee(f2v n, f2v d) {
  bool fx, fy;
  const f2v ds = {__builtin_amdgcn_div_scalef(n.x, d.x, false, &fx), __builtin_amdgcn_div_scalef(n.y, d.y, false, &fy)};
  const f2v ns = {__builtin_amdgcn_div_scalef(n.x, d.x, true, &fx), __builtin_amdgcn_div_scalef(n.y, d.y, true, &fy)};
  const f2v r0 = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  const f2v nd = -ds;
  const f2v e0 = __builtin_elementwise_fma(nd, r0, (f2v){1.0f, 1.0f});
  const f2v r1 = __builtin_elementwise_fma(e0, r0, r0);
  const f2v q0 = ns * r1;
  const f2v e1 = __builtin_elementwise_fma(nd, q0, ns);
  const f2v q1 = __builtin_elementwise_fma(e1, r1, q0);
  const f2v e2 = __builtin_elementwise_fma(nd, q1, ns);
  f2v q;
  q.x = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.x, r1.x, q1.x, fx), d.x, n.x);
  q.y = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.y, r1.y, q1.y, fy), d.y, n.y);
  return q;
}

} // namespace irec

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline bool same(float a, float b) { return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b); }

// mode 0: raw bit patterns; mode 1: moderate magnitudes (sign, exponent 97..157, random mantissa)
__global__ void check_kernel(uint64_t n_per_thread, int mode, uint32_t salt, unsigned long long *bad, float *first_bad) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nb = 0;
  for (uint64_t k = 0; k < n_per_thread; ++k) {
    uint32_t u[4];
    for (int j = 0; j < 4; ++j) u[j] = mix((uint32_t)(tid * 0x9E3779B9u) ^ mix((uint32_t)(k * 4 + j) + salt));
    if (mode == 1)
      for (int j = 0; j < 4; ++j) u[j] = (u[j] & 0x807FFFFFu) | ((97u + (u[j] >> 23) % 61u) << 23);
    const irec::f2v n = {__uint_as_float(u[0]), __uint_as_float(u[1])}, d = {__uint_as_float(u[2]), __uint_as_float(u[3])};
    const irec::f2v q = irec::div2_ieee(n, d);
    const float r0 = n.x / d.x, r1 = n.y / d.y;
    if (!same(q.x, r0) || !same(q.y, r1)) {
      if (nb == 0 && atomicAdd(bad, 0ull) == 0ull) { first_bad[0] = n.x; first_bad[1] = d.x; first_bad[2] = n.y; first_bad[3] = d.y; }
      ++nb;
    }
  }
  if (nb) atomicAdd(bad, nb);
}

__global__ void edge_kernel(const float *v, int nv, unsigned long long *bad) {
  const int i = blockIdx.x, j = threadIdx.x;
  if (i >= nv || j >= nv) return;
  // the pair (v[i] / v[j], v[j] / v[i]) and a pair with a tame partner in either slot
  const irec::f2v q = irec::div2_ieee((irec::f2v){v[i], v[j]}, (irec::f2v){v[j], v[i]});
  const irec::f2v q2 = irec::div2_ieee((irec::f2v){v[i], 3.f}, (irec::f2v){v[j], 7.f});
  const irec::f2v q3 = irec::div2_ieee((irec::f2v){3.f, v[i]}, (irec::f2v){7.f, v[j]});
  unsigned long long nb = 0;
  nb += !same(q.x, v[i] / v[j]); nb += !same(q.y, v[j] / v[i]);
  nb += !same(q2.x, v[i] / v[j]); nb += !same(q2.y, 3.f / 7.f);
  nb += !same(q3.y, v[i] / v[j]); nb += !same(q3.x, 3.f / 7.f);
  if (nb) atomicAdd(bad, nb);
}

template <bool PACKED>
__global__ void rate_kernel(float *out, int iters) {
  irec::f2v n = {1.f + threadIdx.x * 0.001f, 2.f + threadIdx.x * 0.002f}, d = {3.f + blockIdx.x * 0.01f, 5.f};
  irec::f2v acc = {0.f, 0.f};
  for (int k = 0; k < iters; ++k) {
    irec::f2v q;
    if (PACKED) q = irec::div2_ieee(n, d);
    else { q.x = n.x / d.x; q.y = n.y / d.y; }
    acc += q; n += (irec::f2v){0.5f, 0.25f}; d += (irec::f2v){0.125f, 0.375f};
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

int main() {
  unsigned long long *bad; float *fb, *out;
  CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&fb, 16)); CHECK(hipMalloc(&out, 1024 * 256 * 4));
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemset(bad, 0, 8));
    const uint64_t per = 4096;
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, per, mode, 0x1234567u + mode, bad, fb);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0; float hf[4];
    CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hf, fb, 16, hipMemcpyDeviceToHost));
    printf("%s: %llu pairs of divisions, %llu mismatches", mode ? "moderate magnitudes" : "random bit patterns", 4096ull * 256 * per, h);
    if (h) printf("  (first: %a / %a, %a / %a)", hf[0], hf[1], hf[2], hf[3]);
    printf("\n");
  }
  {
    std::vector<float> v;
    const uint32_t bits[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007FFFFFu, 0x00800000u, 0x00800001u, 0x00FFFFFFu, 0x01000000u,
                             0x0C000000u, 0x0B800000u, 0x0C800000u, 0x1F800000u, 0x2F800000u, 0x3F7FFFFFu, 0x3F800000u, 0x3F800001u, 0x40000000u,
                             0x40400000u, 0x4F800000u, 0x5F800000u, 0x6F800000u, 0x7E800000u, 0x7F000000u, 0x7F7FFFFFu, 0x7F800000u, 0xFF800000u,
                             0x7FC00000u, 0x7F800001u, 0xBF800000u, 0xC0400000u, 0x3EAAAAABu, 0x3DCCCCCDu, 0x00400000u, 0x80400000u, 0x7E000000u};
    for (uint32_t b : bits) { union { uint32_t u; float f; } c; c.u = b; v.push_back(c.f); }
    for (int e = -149; e <= 127; e += 3) v.push_back(ldexpf(1.0f, e));
    for (int e = -140; e <= 120; e += 7) v.push_back(ldexpf(1.7320508f, e));
    float *dv; CHECK(hipMalloc(&dv, v.size() * 4)); CHECK(hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(edge_kernel, dim3((unsigned)v.size()), dim3((unsigned)v.size()), 0, 0, dv, (int)v.size(), bad);
    CHECK(hipDeviceSynchronize());
    unsigned long long h = 0; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("edge values: %zu x %zu pairs, %llu mismatches\n", v.size(), v.size(), h);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int iters = 20000;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<false>, dim3(1024), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    hipLaunchKernelGGL(rate_kernel<true>, dim3(1024), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e2)); CHECK(hipEventSynchronize(e2));
    float a, b; CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
    if (rep) printf("rate, 2 divisions per lane and iteration: compiler's / %.3f ms, div2_ieee %.3f ms (%.2fx)\n", a, b, a / b);
  }
  return 0;
}
