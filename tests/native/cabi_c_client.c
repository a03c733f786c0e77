/* A plain C (C11) client of include/phovo_hip.h: proves that the boundary is a C ABI -- the header compiles as
 * C, every declared entry point links, and the host-only calls behave.  Without a GPU the device entry points are
 * only checked for failing loudly.  With a GPU and a problem file as second argument (tests/test_native_clients.py
 * dumps a golden fixture: inputs and the numpy twin's expected result) the program runs ONE alignment through
 * phovo_odometry_* in the reference's call order (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:92-105)
 * and holds the pose to 1e-9.  Exit code 0 = all checks passed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "phovo_hip.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, phovo_last_error()); return 1; } } while (0)

/* Problem file: int32 w, h, levels; double K[9], min_depth, max_depth; per level: int32 max_iter, double min_grad, lambda,
 * grad_scale; u8 gray0[w*h]; double depth0[w*h]; u8 gray1[w*h]; double init[6], expected[6]; int32 expected_iterations[levels]. */
static int align_one_pair_on_the_gpu(const char *path)
{
  FILE *f = fopen(path, "rb");
  CHECK(f != NULL);
  int32_t dims[3];
  double K[9], range[2], init[6], expect[6];
  CHECK(fread(dims, sizeof(int32_t), 3, f) == 3 && fread(K, sizeof(double), 9, f) == 9 && fread(range, sizeof(double), 2, f) == 2);
  const int w = dims[0], h = dims[1], levels = dims[2];
  phovo_config cfg;
  CHECK(phovo_config_default(&cfg) == PHOVO_OK);
  cfg.num_levels = levels;
  for (int l = 0; l < levels; l++) {
    int32_t mi;
    double v[3];
    CHECK(fread(&mi, sizeof(mi), 1, f) == 1 && fread(v, sizeof(double), 3, f) == 3);
    cfg.max_num_iterations[l] = mi;
    cfg.min_gradient_norm[l] = v[0];
    cfg.lambda_optimization_step[l] = v[1];
    cfg.image_gradients_scaling_factor[l] = v[2];
    cfg.blur_filter_size[l] = 0;
  }
  const size_t n = (size_t)w * (size_t)h;
  uint8_t *g0 = malloc(n), *g1 = malloc(n);
  double *d0 = malloc(n * sizeof(double));
  int32_t expect_it[PHOVO_MAX_LEVELS];
  CHECK(g0 && g1 && d0);
  CHECK(fread(g0, 1, n, f) == n && fread(d0, sizeof(double), n, f) == n && fread(g1, 1, n, f) == n);
  CHECK(fread(init, sizeof(double), 6, f) == 6 && fread(expect, sizeof(double), 6, f) == 6);
  CHECK(fread(expect_it, sizeof(int32_t), (size_t)levels, f) == (size_t)levels);
  fclose(f);

  phovo_odometry *o = NULL;
  CHECK(phovo_odometry_create(0, &o) == PHOVO_OK && o != NULL);
  CHECK(phovo_odometry_set_config(o, &cfg) == PHOVO_OK);                                  /* ReadConfigurationFile  :92 */
  CHECK(phovo_odometry_set_min_depth(o, range[0]) == PHOVO_OK && phovo_odometry_set_max_depth(o, range[1]) == PHOVO_OK);
  CHECK(phovo_odometry_set_intrinsic_matrix(o, K) == PHOVO_OK);                           /* SetIntrinsicMatrix     :93 */
  CHECK(phovo_odometry_set_source_frame(o, g0, (size_t)w, d0, (size_t)w * sizeof(double), w, h) == PHOVO_OK);    /* :94 */
  CHECK(phovo_odometry_set_target_frame(o, g1, (size_t)w, NULL, 0, w, h) == PHOVO_OK);                           /* :95 */
  CHECK(phovo_odometry_set_initial_state_vector(o, init) == PHOVO_OK);                    /* SetInitialStateVector  :96 */
  CHECK(phovo_odometry_optimize(o) == PHOVO_OK);                                          /* Optimize               :100 */
  double state[6], rt[16], rt_expect[16];
  phovo_pair_report rep;
  CHECK(phovo_odometry_get_optimal_state_vector(o, state) == PHOVO_OK);
  CHECK(phovo_odometry_get_optimal_rigid_transformation_matrix(o, rt) == PHOVO_OK);       /* :104 */
  CHECK(phovo_odometry_get_report(o, &rep) == PHOVO_OK);
  CHECK(phovo_eigen_pose(expect, rt_expect) == PHOVO_OK);
  double worst = 0.0;
  for (int i = 0; i < 16; i++) worst = fmax(worst, fabs(rt[i] - rt_expect[i]));
  for (int i = 0; i < 6; i++) worst = fmax(worst, fabs(state[i] - expect[i]));
  printf("gpu alignment from C: max |difference| to the expected state / Rt %.3e, iterations", worst);
  for (int l = 0; l < levels; l++) printf(" %d", rep.iterations[l]);
  printf("\n");
  CHECK(worst < 1e-9);
  for (int l = 0; l < levels; l++) CHECK(rep.iterations[l] == expect_it[l]);
  CHECK(rep.flags == 0);
  CHECK(phovo_odometry_destroy(o) == PHOVO_OK);
  free(g0); free(g1); free(d0);
  return 0;
}

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
  } else if (argc > 2) {
    if (align_one_pair_on_the_gpu(argv[2]) != 0) return 1;
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
