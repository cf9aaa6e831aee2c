// What does hipcc make of the f16x3 pass (armenv_actor.h, as it is) when it has to fit TWO waves per SIMD (256 registers per lane)?
// No GPU needed: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../../drl-on-robot-arm_amd/csrc -c actor_256_probe.hip
// then read vgpr / agpr / scratch of the two kernels (python ../isa.py reads a .so; llvm-readelf --notes on the unbundled object here).
// (round 6, profiles/r06_actor_two_waves_budget.txt)
#include "armenv_engine.h"
using namespace armenv;
template <int WAVES>
__global__ __launch_bounds__(256 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void actor_pass_probe(ActorParams A, ActorParamsH H, int64_t n, const float *states, float *actions) {
  __shared__ float4 w1_lds[ACTOR_W1_LDS_FLOATS_H / 4];
  __shared__ uint4 w2_ring[ACTOR_RING_UINT4];
  actor_stage_w1(A.W1P, w1_lds, A.B2W3, 6);
  actor_stage_w1h(A.W1P, w1_lds, 6);
  actor_ring_init(H, w2_ring, 4);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ic = i < n ? i : n - 1;
  float s[6], a[3];
  static_for<0, 6>([&](auto DI) { constexpr int d = DI; s[d] = states[ic * 6 + d]; });
  actor_forward_wg_f16x3<6>(A, H, w1_lds, w2_ring, 4, s, a);
  actor_ring_drain();
  if (i < n) { actions[3 * i] = a[0]; actions[3 * i + 1] = a[1]; actions[3 * i + 2] = a[2]; }
}
template __global__ void actor_pass_probe<1>(ActorParams, ActorParamsH, int64_t, const float *, float *);
template __global__ void actor_pass_probe<2>(ActorParams, ActorParamsH, int64_t, const float *, float *);
