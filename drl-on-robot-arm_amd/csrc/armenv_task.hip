// armenv_task.hip -- one (task, precision) slice of the engine: compiled six times by the Makefile with
// -DARMENV_TU_TASK={reach,push,pick} -DARMENV_TU_PREC={64,32}; each object carries the reset / step / rollout / FK / IK
// kernels of its task for the three chain paths (KUKA, Diana, generic) and exports one factory function.
#include "armenv_engine.h"

#define ARMENV_CAT_(a, b, c) a##b##_f##c
#define ARMENV_CAT(a, b, c) ARMENV_CAT_(a, b, c)
#define ARMENV_LANE_reach ReachLane
#define ARMENV_LANE_push PushLane
#define ARMENV_LANE_pick PickLane
#define ARMENV_LANE_(t) ARMENV_LANE_##t
#define ARMENV_LANE(t) ARMENV_LANE_(t)
#define ARMENV_REAL_64 double
#define ARMENV_REAL_32 float
#define ARMENV_REAL_(p) ARMENV_REAL_##p
#define ARMENV_REAL(p) ARMENV_REAL_(p)

EngineBase *ARMENV_CAT(armenv_make_engine_, ARMENV_TU_TASK, ARMENV_TU_PREC)(const ArmEnvConfig &cfg) {
  return make_task_engine<ARMENV_LANE(ARMENV_TU_TASK), ARMENV_REAL(ARMENV_TU_PREC)>(cfg);
}

// debug entry points of the instrumented build (tests/tools/exp/run_timeline.py); they address this unit's copies of the stamps
extern "C" {
#ifdef ARMENV_TIMELINE
int armenv_dbg_set_timeline(unsigned long long *buf_dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf_dev, sizeof buf_dev);
}
int armenv_dbg_set_actor_timeline(unsigned long long *buf_dev) {     // [4 waves][64 stamps], workgroup 0 of the fused f16x3 rollout
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_actor_tl), &buf_dev, sizeof buf_dev);
}
#endif
}
