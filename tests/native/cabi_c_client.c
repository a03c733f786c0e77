/* A plain C (C11) client of include/phovo_hip.h: proves that the boundary is a C ABI -- the header compiles as
 * C, every declared entry point links, and the host-only calls behave.  No GPU needed: device entry points are
 * only checked for failing loudly when no device is present.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "phovo_hip.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, phovo_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
  phovo_config cfg;
  CHECK(phovo_config_default(&cfg) == PHOVO_OK);
  CHECK(cfg.num_levels == 5 && cfg.max_num_iterations[4] == 50 && cfg.min_gradient_norm[0] == 300.0);
  CHECK(strstr(phovo_version(), "gfx950") != NULL);
  CHECK(strcmp(phovo_status_string(PHOVO_E_CONFIG), "configuration error") == 0);

  if (argc > 1) {                                   /* a reference yml file */
    CHECK(phovo_config_read_file(argv[1], &cfg) == PHOVO_OK);
    CHECK(cfg.num_levels == 4 && cfg.max_num_iterations[2] == 20 && cfg.max_num_iterations[3] == 50);
  }
  CHECK(phovo_config_read_file("/nonexistent.yml", &cfg) == PHOVO_E_IO);
  CHECK(strlen(phovo_last_error()) > 0);

  const double s[6] = {0.1, -0.2, 0.3, 0.01, -0.02, 0.03};
  double rt[16];
  CHECK(phovo_eigen_pose(s, rt) == PHOVO_OK);
  CHECK(rt[3] == 0.1 && rt[7] == -0.2 && rt[11] == 0.3 && rt[15] == 1.0);
  CHECK(fabs(rt[0] * rt[0] + rt[4] * rt[4] + rt[8] * rt[8] - 1.0) < 1e-15);

  /* the VisualOdometry app's pose chain and trajectory line (host arithmetic, ...VisualOdometry.cpp:233-243) */
  {
    const double two[12] = {0.1, -0.2, 0.3, 0.01, -0.02, 0.03, 0.1, -0.2, 0.3, 0.01, -0.02, 0.03};
    double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, poses[32], prod[16];
    char line[256];
    CHECK(phovo_trajectory_chain(2, two, pose, poses) == PHOVO_OK);
    /* pose_1 * Rt must be the identity: pose_1 = Rt^-1 */
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        prod[4 * i + j] = 0;
        for (int k = 0; k < 4; k++) prod[4 * i + j] += poses[4 * i + k] * rt[4 * k + j];
        CHECK(fabs(prod[4 * i + j] - (i == j ? 1.0 : 0.0)) < 1e-14);
      }
    CHECK(memcmp(pose, poses + 16, sizeof(pose)) == 0);
    CHECK(phovo_trajectory_format_pose(1305031102.175304, pose, line, sizeof(line)) == PHOVO_OK);
    CHECK(strncmp(line, "1305031102.175304 ", 18) == 0 && strlen(line) > 100);
    CHECK(phovo_trajectory_format_pose(1.0, pose, line, 8) == PHOVO_E_INVALID_ARGUMENT);
    CHECK(phovo_trajectory_chain(1, NULL, pose, NULL) == PHOVO_E_INVALID_ARGUMENT);
  }

  if (phovo_device_count() == 0) {                  /* no GPU: the product path must refuse, not fall back */
    phovo_odometry *o = NULL;
    phovo_engine *e = NULL;
    CHECK(phovo_odometry_create(0, &o) == PHOVO_E_HIP && o == NULL);
    CHECK(phovo_engine_create(0, &e) == PHOVO_E_HIP && e == NULL);
  }
  /* NULL handles are rejected, not dereferenced */
  CHECK(phovo_odometry_optimize(NULL) == PHOVO_E_INVALID_ARGUMENT);
  CHECK(phovo_engine_align_pairs(NULL, 0, NULL, NULL, NULL, NULL, NULL) == PHOVO_E_INVALID_ARGUMENT);
  CHECK(phovo_engine_upload_frames_u16(NULL, 0, 0, 0, NULL, 0, 0, NULL, 0, 0, 1.0) == PHOVO_E_INVALID_ARGUMENT);
  CHECK(phovo_engine_set_level_fusion(NULL, PHOVO_FUSION_OFF) == PHOVO_E_INVALID_ARGUMENT);
  CHECK(phovo_engine_set_slide_policy(NULL, 0) == PHOVO_E_INVALID_ARGUMENT);
  CHECK(phovo_engine_destroy(NULL) == PHOVO_OK && phovo_odometry_destroy(NULL) == PHOVO_OK);
  printf("cabi_c_client ok\n");
  return 0;
}
