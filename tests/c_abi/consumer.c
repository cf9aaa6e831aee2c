/* A plain-C consumer of include/armenv.h: what a non-Python host (the reference has none) would
 * write against libarmenv.so.  Creates N reach envs, resets them, takes `steps` steps with a fixed action and prints
 * env 0's observation, the reward sum and the engine's counters as one line of numbers.
 * Built and run by tests/test_gpu_parity.py::test_plain_c_consumer (gcc, -lamdhip64; no torch in the process) and
 * link-checked without a GPU by tests/test_host_logic.py. */
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "armenv.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_ENV(x) do { int rc_ = (x); if (rc_ != ARMENV_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, armenv_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 4096;
  const int steps = argc > 2 ? atoi(argv[2]) : 10;
  ArmEnvConfig cfg;
  CHECK_ENV(armenv_default_config(ARMENV_TASK_REACH, &cfg));
  cfg.num_envs = n;
  cfg.seed = 7;
  ArmEnv *env = NULL;
  CHECK_ENV(armenv_create(&cfg, &env));
  if (armenv_obs_dim(env) != 6 || armenv_action_dim(env) != 3 || armenv_num_envs(env) != n) return 4;

  float *obs, *act, *rew;
  uint8_t *done, *succ;
  CHECK_HIP(hipMalloc((void **)&obs, sizeof(float) * 6 * n));
  CHECK_HIP(hipMalloc((void **)&act, sizeof(float) * 3 * n));
  CHECK_HIP(hipMalloc((void **)&rew, sizeof(float) * n));
  CHECK_HIP(hipMalloc((void **)&done, n));
  CHECK_HIP(hipMalloc((void **)&succ, n));
  float *h_act = (float *)malloc(sizeof(float) * 3 * n);
  for (long i = 0; i < n; ++i) { h_act[3 * i] = 0.5f; h_act[3 * i + 1] = -0.25f; h_act[3 * i + 2] = -0.6f; }
  CHECK_HIP(hipMemcpy(act, h_act, sizeof(float) * 3 * n, hipMemcpyHostToDevice));

  CHECK_ENV(armenv_reset(env, NULL, obs, NULL));
  for (int t = 0; t < steps; ++t) CHECK_ENV(armenv_step(env, act, obs, rew, done, succ, NULL, NULL, NULL, NULL));
  CHECK_HIP(hipDeviceSynchronize());

  float h_obs[6];
  float *h_rew = (float *)malloc(sizeof(float) * n);
  CHECK_HIP(hipMemcpy(h_obs, obs, sizeof h_obs, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_rew, rew, sizeof(float) * n, hipMemcpyDeviceToHost));
  double rsum = 0.0;
  for (long i = 0; i < n; ++i) rsum += h_rew[i];
  uint64_t c[16];
  CHECK_ENV(armenv_counters(env, c, NULL));
  printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %llu %llu %s\n", h_obs[0], h_obs[1], h_obs[2], h_obs[3], h_obs[4], h_obs[5], rsum,
         (unsigned long long)c[2], (unsigned long long)c[3], armenv_kernel_name(env));

  /* error path: a NULL output buffer is refused with a message, nothing aborts */
  if (armenv_step(env, act, NULL, rew, done, succ, NULL, NULL, NULL, NULL) != ARMENV_EINVAL || armenv_last_error()[0] == '\0') return 5;
  armenv_destroy(env);
  hipFree(obs); hipFree(act); hipFree(rew); hipFree(done); hipFree(succ);
  free(h_act); free(h_rew);
  return 0;
}
