// irec_team_common.h -- pieces shared by the team encoders (irec_team.hip, irec_ten.hip): team size, the three quantile-table copies
// at the start of the LDS, the team barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "irec_device.h"
#include "irec_kernels.h"

namespace irec {

constexpr int TEAM_NW = 4;                       // waves per team
constexpr int TEAM_NT = TEAM_NW * 64;            // threads per team
constexpr uint32_t T3_FLOATS = 3u * IREC_PM1;    // three copies of lut2
constexpr size_t T3_BYTES = ((size_t)T3_FLOATS * 4 + 15) & ~(size_t)15;


// Barrier of the 4 waves of one team: a monotonic LDS counter.  LDS operations of one wave execute in program order and
// the LDS serves one instruction at a time, so a wave's earlier writes are in place before its add lands; the fences
// order the global slab traffic (vmcnt) the way __syncthreads would.
struct TeamBarrier {
  uint32_t *cnt;
  uint32_t epoch;
  uint32_t n_waves;
  __device__ __forceinline__ void operator()() {
    epoch += n_waves;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if ((int32_t)(v - epoch) >= 0) break; // every wave of the team has arrived
      __builtin_amdgcn_s_sleep(1);   // (64 clocks between two polls; 0 / 2 / 4: neutral within 0.3 %, r05l)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};

} // namespace irec
