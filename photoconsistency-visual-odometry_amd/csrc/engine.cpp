// Host side of the C ABI (include/phovo_hip.h): frame pool in HBM, per-level launches of the
// Gauss-Newton kernel, and the single-pair wrapper that mirrors
// phovo::Analytic::CPhotoconsistencyOdometryAnalytic (CPhotoconsistencyOdometryAnalytic.h:428-607).
// There is no CPU fallback anywhere in this file: without a HIP device every entry point that
// touches data fails with PHOVO_E_HIP.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "phovo_internal.hpp"

namespace phovo_hip {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg) { g_last_error = msg; return status; }

#ifdef PHOVO_TUNING
bool tuning_switch(const char *name) { return std::getenv(name) != nullptr; }
#endif

#define PHOVO_HIP_CHECK(expr)                                                                   \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail(PHOVO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

struct LevelPool {
  int w = 0, h = 0, n = 0;
  bool stored = false;
  unsigned char *planes = nullptr;   // frame f at planes + f*frame_bytes, plane p of it at + plane_off[p]
  size_t frame_bytes = 0;
  size_t plane_off[PLANES_PER_FRAME] = {0, 0, 0, 0};
  size_t rec_off = 0;                  // bilinear sampling: the frame's tap records {I, GX, GY, pad} behind its planes (0: none)
  GNLaunchPlan plan{};
  bool plan_ok = false;
  GNLaunchPlan plan_few{};             // geometry for a handful of pairs (LATENCY_PAIRS or fewer)
  bool plan_few_ok = false;
};

}  // namespace phovo_hip

using namespace phovo_hip;

// Everything ONE enqueue owns: its stream, its pair data, its scratch and its timing events.  The engine keeps
// PHOVO_ENQUEUE_DEPTH of them and uses them in turn, so that enqueue k + 1 can be issued -- and its kernels can start filling
// the CUs that enqueue k's last, long pairs leave idle -- before enqueue k has finished (phovo_hip.h, "Pipelining").
struct AlignSlot {
  hipStream_t stream = nullptr;
  hipEvent_t ev_total_start = nullptr, ev_total_stop = nullptr;
  hipEvent_t ev_start[PHOVO_MAX_LEVELS] = {};
  hipEvent_t ev_stop[PHOVO_MAX_LEVELS] = {};
  bool level_launched[PHOVO_MAX_LEVELS] = {};
  bool have_timing = false;
  int ticket = 0;                              // the enqueue this slot holds (0: none yet)
  int last_pairs = 0;
  // Per-launch pair data in ONE device allocation, laid out for the pairs of the enqueue as
  //   [src int32 | tgt int32 | states fp64 x6 | reports | work-queue heads | hand-over lists]
  // so that an enqueue is one host-to-device copy (src, tgt, initial states, from the pinned mirror h_up) and one
  // memset (reports + heads + lists), and a fetch is one device-to-host copy (states + reports, into the pinned h_down).
  int pair_capacity = 0;
  unsigned char *d_pairs = nullptr;
  unsigned char *h_up = nullptr, *h_down = nullptr;      // pinned
  int *d_src = nullptr, *d_tgt = nullptr;                // views into d_pairs for the enqueue
  double *d_states = nullptr;
  phovo_pair_report *d_reports = nullptr;
  int *d_work_counters = nullptr;              // [2][PHOVO_MAX_LEVELS][QUEUES_PER_LEVEL x QUEUE_HEAD_STRIDE] work-queue heads of the level launches (view into d_pairs)
  int *d_handover = nullptr;                   // [PHOVO_MAX_LEVELS][pairs + 2] hand-over lists: sliding-window kernel -> exact kernel (view into d_pairs)
  int *d_owner = nullptr;                      // owner maps in HBM (levels whose map exceeds LDS; the wide form)
  size_t owner_capacity = 0;
  unsigned long long *d_mask = nullptr;        // in-bounds ballots in HBM (levels whose ballots do not fit LDS next to the rest)
  size_t mask_capacity = 0;
  bool owner_tagged = false;                   // d_owner holds tagged entries of the persistent kernel, not the -1 the wide form expects
  void *d_wide_ws = nullptr;                   // workspace of the wide (many-workgroups-per-pair) level form
  size_t wide_ws_capacity = 0;
  std::vector<int> h_wide_done;
  std::vector<phovo_launch_record> launches;   // what the enqueue launched, in order (phovo_engine_last_launches)
};

struct phovo_engine {
  int device = 0;
  hipStream_t stream = nullptr;                // uploads, pyramid producers, plane access
  hipStream_t copy_stream = nullptr;           // batched uploads: host-to-device copies of chunk i+1 run beside the pyramid kernels of chunk i
  hipEvent_t ev_copied[2] = {}, ev_built[2] = {};     // per staging half
  AlignSlot slots[PHOVO_ENQUEUE_DEPTH];
  int ticket = 0;                              // tickets handed out so far; enqueue t lives in slots[t % PHOVO_ENQUEUE_DEPTH]

  phovo_config cfg{};
  phovo_extensions ext{};                      // plane storage, Huber deltas: all off by default
  double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool have_K = false;
  double min_depth = 0.3, max_depth = 5.0;     // ...Analytic.h:430
  bool build_all = false;

  int n_frames = 0, width = 0, height = 0;
  LevelPool levels[PHOVO_MAX_LEVELS];
  // staging for raw frames: `stage_frames` frames of each kind that has been used so far
  int stage_frames = 0;
  bool stage_has_f64 = false, stage_has_u16 = false;
  uint8_t *d_gray = nullptr;
  double *d_depth = nullptr;
  uint16_t *d_depth16 = nullptr;
  double *d_tmp = nullptr;
  double *d_blur0 = nullptr;                   // blurFilterSize[0] > 0: the blurred level-0 intensity of a staging chunk (see build_pyramids)
  double *d_scratch = nullptr;                 // fp64 planes of one staging chunk (narrow storages, plane get/set)
  double *d_blur_kernel = nullptr;             // [levels][max ksize]
  int blur_kernel_stride = 0;

  int slide_policy = 0;                        // 0 automatic (where the owner map exceeds LDS), -1 never
  int fusion = PHOVO_FUSION_AUTO;              // consecutive levels in one launch (phovo_engine_set_level_fusion)
  int cu_count = 256;
  int wide_policy = 0;                         // 0 auto, 1 always (where possible), -1 never
  bool batch_invariant = false;                // every batch takes the same kernels and geometries (phovo_engine_set_batch_invariant)
  bool latency_forms = false;                  // a handful of pairs may take the forms that finish soonest also where a level has a one-workgroup form with its owner map in LDS (phovo_engine_set_latency_forms)
};

namespace {

void free_pool(phovo_engine *e)
{
  for (auto &lv : e->levels) {
    if (lv.planes) (void)hipFree(lv.planes);
    lv = LevelPool{};
  }
  if (e->d_gray) (void)hipFree(e->d_gray);
  if (e->d_depth) (void)hipFree(e->d_depth);
  if (e->d_depth16) (void)hipFree(e->d_depth16);
  if (e->d_tmp) (void)hipFree(e->d_tmp);
  if (e->d_blur_kernel) (void)hipFree(e->d_blur_kernel);
  if (e->d_scratch) (void)hipFree(e->d_scratch);
  if (e->d_blur0) (void)hipFree(e->d_blur0);
  e->d_scratch = nullptr; e->d_blur0 = nullptr;
  e->d_gray = nullptr; e->d_depth = nullptr; e->d_depth16 = nullptr; e->d_tmp = nullptr;
  e->d_blur_kernel = nullptr;
  e->stage_frames = 0; e->stage_has_f64 = e->stage_has_u16 = false;
  e->n_frames = 0; e->width = 0; e->height = 0;
}

void free_pairs(AlignSlot &s)
{
  if (s.d_pairs) (void)hipFree(s.d_pairs);
  if (s.h_up) (void)hipHostFree(s.h_up);
  if (s.h_down) (void)hipHostFree(s.h_down);
  s.d_pairs = nullptr; s.h_up = s.h_down = nullptr; s.d_work_counters = nullptr; s.d_handover = nullptr;
  s.d_src = s.d_tgt = nullptr; s.d_states = nullptr; s.d_reports = nullptr;
  s.pair_capacity = 0;
}

void free_slot(AlignSlot &s)
{
  free_pairs(s);
  if (s.d_owner) (void)hipFree(s.d_owner);
  if (s.d_mask) (void)hipFree(s.d_mask);
  if (s.d_wide_ws) (void)hipFree(s.d_wide_ws);
  s.d_owner = nullptr; s.owner_capacity = 0; s.d_wide_ws = nullptr; s.wide_ws_capacity = 0;
  s.d_mask = nullptr; s.mask_capacity = 0;
}

// Every enqueue in flight has finished when this returns (host wait).  Called by whatever changes device state that a
// running alignment reads: uploads, plane writes, pool and configuration changes.
hipError_t quiesce(phovo_engine *e)
{
  for (AlignSlot &s : e->slots) {
    if (!s.stream) continue;
    const hipError_t he = hipStreamSynchronize(s.stream);
    if (he != hipSuccess) return he;
  }
  return hipSuccess;
}

int validate_config(const phovo_config *c)
{
  if (c->num_levels < 1 || c->num_levels > PHOVO_MAX_LEVELS)
    return fail(PHOVO_E_CONFIG, "num_levels out of range [1, 16]");
  for (int l = 0; l < c->num_levels; l++) {
    if (c->max_num_iterations[l] < 0) return fail(PHOVO_E_CONFIG, "max_num_iterations < 0");
    const int b = c->blur_filter_size[l];
    if (b < 0 || (b > 0 && (b % 2) == 0))
      return fail(PHOVO_E_CONFIG, "blurFilterSize must be 0 or odd (cv::GaussianBlur requires an odd kernel)");
    if (b > 63) return fail(PHOVO_E_CONFIG, "blurFilterSize > 63");
  }
  return PHOVO_OK;
}

void level_dims(int w, int h, int level, int *lw, int *lh)
{
  const double f = 1.0 / (double)(1 << level);        // factor = factor/2  (...Analytic.h:161)
  *lw = (int)std::rint((double)w * f);                // Size(0,0), fx, fy -> cvRound(cols*fx)
  *lh = (int)std::rint((double)h * f);
}

// A handful of pairs on a large level: cut every pair into many workgroups (gn_wide_kernels.hip) instead of
// giving it one.  Only the reference-exact configuration (fp64 planes, no extension) takes this form.
// The forms sum in different orders, so where a level has a one-workgroup form that is fast enough -- its owner map fits LDS:
// up to ~39 k pixels, every active level of the shipped 4- and 5-level files on 640x480 -- that form runs whatever the batch
// size and a pair has ONE arithmetic (the reference has one, ...Analytic.h:500-563: Optimize() through the class surface and
// the same pair in a batch of thousands agree bit for bit); the caller may trade that for latency
// (phovo_engine_set_latency_forms).  Larger levels would take hundreds of microseconds per iteration in one workgroup:
// there a handful of pairs takes the wide form unless the caller pins the batch forms (phovo_engine_set_batch_invariant).
bool use_wide_level(const phovo_engine *e, int n_pairs, const LevelPool &lv)
{
  if (e->wide_policy < 0) return false;
  if (e->ext.plane_storage != PHOVO_STORAGE_F64 || e->ext.sampling != PHOVO_SAMPLING_NEAREST_SCATTER) return false;
  if (e->wide_policy > 0) return true;
  if (e->batch_invariant) return false;        // the automatic choice looks at the batch size
  if (n_pairs * 8 > 256) return false;
  if (lv.plan_ok && lv.plan.owner_in_lds) return e->latency_forms && lv.n >= 16384;
  return true;
}

constexpr int HEAD_SETS = 2;
// Byte offsets of the per-launch pair data for n pairs (see phovo_engine::d_pairs); every section starts 8-byte aligned.
struct PairLayout {
  size_t src, tgt, states, reports, heads, handover, handover_stride, total;
};
PairLayout pair_layout(int n_pairs)
{
  const size_t n2 = ((size_t)n_pairs + 1) & ~(size_t)1;      // two int32 per 8 bytes
  PairLayout l;
  l.src = 0;
  l.tgt = l.src + sizeof(int) * n2;
  l.states = l.tgt + sizeof(int) * n2;
  l.reports = l.states + sizeof(double) * 6 * (size_t)n_pairs;
  l.heads = l.reports + sizeof(phovo_pair_report) * (size_t)n_pairs;
  // two sets of heads per level: the sliding-window launch of a level and the exact launch behind it drain their own queues
  l.handover = l.heads + sizeof(int) * HEAD_SETS * PHOVO_MAX_LEVELS * QUEUE_HEADS_INTS;
  l.handover_stride = n2 + 2;                   // ints per level: the list of handed-over pairs and, at [n_pairs], its length
  l.total = l.handover + sizeof(int) * l.handover_stride * PHOVO_MAX_LEVELS;
  return l;
}

int ensure_pairs(AlignSlot &s, int n_pairs)
{
  if (n_pairs <= s.pair_capacity) return PHOVO_OK;
  free_pairs(s);
  const PairLayout cap = pair_layout(n_pairs);
  PHOVO_HIP_CHECK(hipMalloc(&s.d_pairs, cap.total));
  PHOVO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s.h_up), cap.reports, hipHostMallocDefault));
  PHOVO_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s.h_down), cap.heads - cap.states, hipHostMallocDefault));
  s.pair_capacity = n_pairs;
  return PHOVO_OK;
}

constexpr int LATENCY_PAIRS = 8;     // up to this many pairs per launch the level kernels use the latency geometry
enum DepthKind { DEPTH_NONE = 0, DEPTH_F64 = 1, DEPTH_U16 = 2 };
constexpr int STAGE_CHUNK = 32;      // frames copied and processed per batch of producer launches

// Makes sure the staging buffers hold `frames` raw frames of the kinds asked for.
int ensure_stage(phovo_engine *e, int frames, bool want_f64, bool want_u16)
{
  const size_t px = (size_t)e->width * (size_t)e->height;
  const bool grow = frames > e->stage_frames;
  if (grow || (want_f64 && !e->stage_has_f64) || (want_u16 && !e->stage_has_u16)) {
    PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
    const int cap = grow ? frames : e->stage_frames;
    want_f64 = want_f64 || e->stage_has_f64;
    want_u16 = want_u16 || e->stage_has_u16;
    if (e->d_gray) (void)hipFree(e->d_gray);
    if (e->d_depth) (void)hipFree(e->d_depth);
    if (e->d_depth16) (void)hipFree(e->d_depth16);
    if (e->d_blur0) (void)hipFree(e->d_blur0);
    e->d_gray = nullptr; e->d_depth = nullptr; e->d_depth16 = nullptr; e->d_blur0 = nullptr;
    e->stage_frames = 0; e->stage_has_f64 = e->stage_has_u16 = false;
    PHOVO_HIP_CHECK(hipMalloc(&e->d_gray, px * (size_t)cap));
    // (the blurred level-0 image of ONE chunk: build_pyramids uses it at offset 0 on the engine's stream, whichever
    // staging half the chunk's raw frames sit in)
    if (e->cfg.blur_filter_size[0] > 0)
      PHOVO_HIP_CHECK(hipMalloc(&e->d_blur0, px * (size_t)(cap < STAGE_CHUNK ? cap : STAGE_CHUNK) * sizeof(double)));
    if (want_f64) PHOVO_HIP_CHECK(hipMalloc(&e->d_depth, px * (size_t)cap * sizeof(double)));
    if (want_u16) PHOVO_HIP_CHECK(hipMalloc(&e->d_depth16, px * (size_t)cap * sizeof(uint16_t)));
    e->stage_frames = cap; e->stage_has_f64 = want_f64; e->stage_has_u16 = want_u16;
  }
  return PHOVO_OK;
}

// Builds the pyramids of `count` consecutive frames whose raw data sits in the staging buffers:
// one launch per (level, producer) for the whole batch.
int build_pyramids(phovo_engine *e, int first_frame, int count, int roles, DepthKind kind, double depth_scale,
                   int stage_offset = 0)
{
  const int w = e->width, h = e->height;
  const size_t px = (size_t)w * (size_t)h;
  // the raw frames of this chunk sit `stage_offset` frames into the staging buffers (double-buffered uploads)
  const uint8_t *s_gray = e->d_gray + px * (size_t)stage_offset;
  const double *s_depth = e->d_depth ? e->d_depth + px * (size_t)stage_offset : nullptr;
  const uint16_t *s_depth16 = e->d_depth16 ? e->d_depth16 + px * (size_t)stage_offset : nullptr;
  const int storage = e->ext.plane_storage;
  // blurFilterSize[0] > 0: the converted level-0 image, blurred twice, is the image every other level is resized from
  // (whether or not level 0 itself is resident)
  const double *blurred0 = nullptr;
  if (e->cfg.blur_filter_size[0] > 0) {
    const int ks0 = e->cfg.blur_filter_size[0];
    PHOVO_HIP_CHECK(pyr_intensity_level(s_gray, px, count, w, h, 0, w, h, e->d_blur0, px, e->stream));   // convertTo  :471,484
    for (int f = 0; f < count; f++) {
      PHOVO_HIP_CHECK(pyr_gaussian_blur(e->d_blur0 + (size_t)f * px, e->d_tmp, w, h, ks0, e->d_blur_kernel, e->stream));
      PHOVO_HIP_CHECK(pyr_gaussian_blur(e->d_blur0 + (size_t)f * px, e->d_tmp, w, h, ks0, e->d_blur_kernel, e->stream));
    }
    blurred0 = e->d_blur0;
  }
  for (int l = 0; l < e->cfg.num_levels; l++) {
    LevelPool &lv = e->levels[l];
    if (!lv.stored) continue;
    // fp64 storage: the producers write straight into the pool.  Narrow storage: they write fp64 planes into
    // the scratch chunk (same [frame][4][n] layout) and a convert pass rounds them into the pool once.
    const size_t fstride = (size_t)PLANES_PER_FRAME * (size_t)lv.n;
    const bool direct = storage == PHOVO_STORAGE_F64 && lv.rec_off == 0;
    double *base = direct ? reinterpret_cast<double *>(lv.planes + (size_t)first_frame * lv.frame_bytes) : e->d_scratch;
    // BuildPyramid(intensity, applyBlur = true)  :474,487
    const int ks = e->cfg.blur_filter_size[l];
    const double *kern = ks > 0 ? e->d_blur_kernel + (size_t)l * e->blur_kernel_stride : nullptr;
    if (blurred0) {
      // Level 0 was blurred IN PLACE (`imgAux = img` is a shallow cv::Mat alias, :136, and GaussianBlur(imgAux, imgAux)
      // writes through it, :146-147), so every cv::resize(img, ...) of the later levels (:132) reads the blurred
      // level 0 (blurred0, made before this loop); level 0 itself is that image, not blurred a second time.
      PHOVO_HIP_CHECK(pyr_depth_level(blurred0, px, count, w, h, l, lv.w, lv.h, base + (size_t)PLANE_I * lv.n, fstride,
                                      e->stream));
    } else {
      PHOVO_HIP_CHECK(pyr_intensity_level(s_gray, px, count, w, h, l, lv.w, lv.h,
                                          base + (size_t)PLANE_I * lv.n, fstride, e->stream));
    }
    if (ks > 0 && !(blurred0 && l == 0)) {                              // GaussianBlur twice  :144-148
      for (int f = 0; f < count; f++) {
        double *pi = base + (size_t)f * fstride + (size_t)PLANE_I * lv.n;
        PHOVO_HIP_CHECK(pyr_gaussian_blur(pi, e->d_tmp, lv.w, lv.h, ks, kern, e->stream));
        PHOVO_HIP_CHECK(pyr_gaussian_blur(pi, e->d_tmp, lv.w, lv.h, ks, kern, e->stream));
      }
    }
    if (roles & PHOVO_ROLE_SOURCE) {                                    // BuildPyramid(depth, false)  :475
      double *pd = base + (size_t)PLANE_D * lv.n;
      if (kind == DEPTH_U16)
        PHOVO_HIP_CHECK(pyr_depth_level_u16(s_depth16, px, depth_scale, count, w, h, l, lv.w, lv.h, pd, fstride, e->stream));
      else
        PHOVO_HIP_CHECK(pyr_depth_level(s_depth, px, count, w, h, l, lv.w, lv.h, pd, fstride, e->stream));
    }
    if (roles & PHOVO_ROLE_TARGET)                                      // BuildDerivativesPyramids  :490
      PHOVO_HIP_CHECK(pyr_scharr(base, fstride, (size_t)PLANE_I * lv.n, (size_t)PLANE_GX * lv.n,
                                 (size_t)PLANE_GY * lv.n, count, lv.w, lv.h,
                                 e->cfg.image_gradients_scaling_factor[l], e->stream));
    if (!direct) {
      unsigned char *dst = lv.planes + (size_t)first_frame * lv.frame_bytes;
      const bool want[PLANES_PER_FRAME] = {true, (roles & PHOVO_ROLE_SOURCE) != 0, (roles & PHOVO_ROLE_TARGET) != 0,
                                           (roles & PHOVO_ROLE_TARGET) != 0};
      for (int p = 0; p < PLANES_PER_FRAME; p++) {
        if (!want[p]) continue;
        PHOVO_HIP_CHECK(pyr_store_plane(base + (size_t)p * lv.n, fstride, count, lv.n, dst + lv.plane_off[p],
                                        lv.frame_bytes, storage, p == PLANE_D, e->stream));
      }
    }
    if (lv.rec_off && (roles & PHOVO_ROLE_TARGET))                      // bilinear sampling: the target's tap records
      PHOVO_HIP_CHECK(pyr_build_tap_records(lv.planes + (size_t)first_frame * lv.frame_bytes, lv.frame_bytes, lv.plane_off,
                                            lv.rec_off, count, lv.n, storage, e->stream));
  }
  return PHOVO_OK;
}

int copy_rows_to_device(void *dst, const void *src, size_t stride, size_t row_bytes, int rows,
                        hipStream_t stream)
{
  if (stride == row_bytes) {
    PHOVO_HIP_CHECK(hipMemcpyAsync(dst, src, row_bytes * (size_t)rows, hipMemcpyHostToDevice, stream));
  } else {
    if (stride < row_bytes) return fail(PHOVO_E_INVALID_ARGUMENT, "stride smaller than a row");
    PHOVO_HIP_CHECK(hipMemcpy2DAsync(dst, row_bytes, src, stride, row_bytes, (size_t)rows,
                                     hipMemcpyHostToDevice, stream));
  }
  return PHOVO_OK;
}

}  // namespace

extern "C" {

const char *phovo_version(void) { return "phovo-hip 0.1.0 (gfx950)"; }

const char *phovo_status_string(int status)
{
  switch (status) {
    case PHOVO_OK: return "ok";
    case PHOVO_E_INVALID_ARGUMENT: return "invalid argument";
    case PHOVO_E_CONFIG: return "configuration error";
    case PHOVO_E_SHAPE: return "shape error";
    case PHOVO_E_HIP: return "HIP runtime error";
    case PHOVO_E_NOT_READY: return "not ready (call order)";
    case PHOVO_E_IO: return "I/O error";
    case PHOVO_E_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}

const char *phovo_last_error(void) { return g_last_error.c_str(); }

int phovo_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int phovo_config_default(phovo_config *cfg)
{
  if (!cfg) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_config_default: null");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->num_levels = 5;                                   // ...Analytic.h:433
  for (int l = 0; l < PHOVO_MAX_LEVELS; l++) {
    cfg->blur_filter_size[l] = 0;                        // :434
    cfg->image_gradients_scaling_factor[l] = 0.0625;     // :435
    cfg->lambda_optimization_step[l] = 1.0;              // :436
    cfg->max_num_iterations[l] = 0;                      // :437
    cfg->min_gradient_norm[l] = 300.0;                   // :441
  }
  cfg->max_num_iterations[2] = 5;                        // :438
  cfg->max_num_iterations[3] = 20;                       // :439
  cfg->max_num_iterations[4] = 50;                       // :440
  cfg->visualize_iterations = 0;                         // :442
  return PHOVO_OK;
}

int phovo_config_read_file(const char *path, phovo_config *cfg) { return read_config_file(path, cfg); }

int phovo_eigen_pose(const double s[6], double rt[16])
{
  if (!s || !rt) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_eigen_pose: null");
  const double x = s[0], y = s[1], z = s[2], yaw = s[3], pitch = s[4], roll = s[5];
  rt[0] = std::cos(yaw) * std::cos(pitch);               // CPhotoconsistencyOdometry.h:52-70
  rt[1] = std::cos(yaw) * std::sin(pitch) * std::sin(roll) - std::sin(yaw) * std::cos(roll);
  rt[2] = std::cos(yaw) * std::sin(pitch) * std::cos(roll) + std::sin(yaw) * std::sin(roll);
  rt[3] = x;
  rt[4] = std::sin(yaw) * std::cos(pitch);
  rt[5] = std::sin(yaw) * std::sin(pitch) * std::sin(roll) + std::cos(yaw) * std::cos(roll);
  rt[6] = std::sin(yaw) * std::sin(pitch) * std::cos(roll) - std::cos(yaw) * std::sin(roll);
  rt[7] = y;
  rt[8] = -std::sin(pitch);
  rt[9] = std::cos(pitch) * std::sin(roll);
  rt[10] = std::cos(pitch) * std::cos(roll);
  rt[11] = z;
  rt[12] = 0; rt[13] = 0; rt[14] = 0; rt[15] = 1;
  return PHOVO_OK;
}

/* ------------------------------------------------------------------ engine ------------------ */

int phovo_engine_create(int device, phovo_engine **out)
{
  if (!out) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_engine_create: null");
  *out = nullptr;
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count <= 0)
    return fail(PHOVO_E_HIP, "no HIP device available: this library has no CPU path");
  if (device < 0 || device >= count) return fail(PHOVO_E_INVALID_ARGUMENT, "device index out of range");
  PHOVO_HIP_CHECK(hipSetDevice(device));
  phovo_engine *e = new (std::nothrow) phovo_engine();
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "out of host memory");
  e->device = device;
  phovo_config_default(&e->cfg);
  phovo_extensions_default(&e->ext);
  hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
  for (int i = 0; i < 2 && he == hipSuccess; i++) {
    he = hipEventCreateWithFlags(&e->ev_copied[i], hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_built[i], hipEventDisableTiming);
  }
  for (AlignSlot &s : e->slots) {
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreate(&s.ev_total_start);
    if (he == hipSuccess) he = hipEventCreate(&s.ev_total_stop);
    for (int l = 0; l < PHOVO_MAX_LEVELS && he == hipSuccess; l++) {
      he = hipEventCreate(&s.ev_start[l]);
      if (he == hipSuccess) he = hipEventCreate(&s.ev_stop[l]);
    }
  }
  if (he == hipSuccess) he = gn_prepare_kernels();
  if (he == hipSuccess) he = gn_prepare_slide_kernels();
  if (he == hipSuccess) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->cu_count = cus;
  }
  if (he != hipSuccess) {
    phovo_engine_destroy(e);
    return fail(PHOVO_E_HIP, std::string("engine setup: ") + hipGetErrorString(he));
  }
  *out = e;
  return PHOVO_OK;
}

int phovo_engine_destroy(phovo_engine *e)
{
  if (!e) return PHOVO_OK;
  (void)hipSetDevice(e->device);
  if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
    (void)quiesce(e);
  free_pool(e);
  for (AlignSlot &s : e->slots) {
    free_slot(s);
    for (int l = 0; l < PHOVO_MAX_LEVELS; l++) {
      if (s.ev_start[l]) (void)hipEventDestroy(s.ev_start[l]);
      if (s.ev_stop[l]) (void)hipEventDestroy(s.ev_stop[l]);
    }
    if (s.ev_total_start) (void)hipEventDestroy(s.ev_total_start);
    if (s.ev_total_stop) (void)hipEventDestroy(s.ev_total_stop);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  for (int i = 0; i < 2; i++) {
    if (e->ev_copied[i]) (void)hipEventDestroy(e->ev_copied[i]);
    if (e->ev_built[i]) (void)hipEventDestroy(e->ev_built[i]);
  }
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return PHOVO_OK;
}

int phovo_engine_set_config(phovo_engine *e, const phovo_config *cfg)
{
  if (!e || !cfg) return fail(PHOVO_E_INVALID_ARGUMENT, "set_config: null");
  const int st = validate_config(cfg);
  if (st != PHOVO_OK) return st;
  // A configuration that changes which levels exist or how their planes are built drops the pool:
  // the reference requires the configuration before Set*Frame (:474 uses m_NumOptimizationLevels).
  // Changing only lambda / max_num_iterations (within the resident levels) / min_gradient_norm keeps it.
  bool keep = e->n_frames > 0 && cfg->num_levels == e->cfg.num_levels;
  for (int l = 0; keep && l < cfg->num_levels; l++) {
    if (cfg->blur_filter_size[l] != e->cfg.blur_filter_size[l]) keep = false;
    if (cfg->image_gradients_scaling_factor[l] != e->cfg.image_gradients_scaling_factor[l]) keep = false;
    if (cfg->max_num_iterations[l] > 0 && !e->levels[l].stored) keep = false;
  }
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
    (void)quiesce(e);
  if (!keep) free_pool(e);
  e->cfg = *cfg;
  return PHOVO_OK;
}

int phovo_extensions_default(phovo_extensions *ext)
{
  if (!ext) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_extensions_default: null");
  std::memset(ext, 0, sizeof(*ext));
  ext->plane_storage = PHOVO_STORAGE_F64;
  return PHOVO_OK;
}

int phovo_extensions_read_file(const char *path, phovo_extensions *ext) { return read_extensions_file(path, ext); }

int phovo_engine_set_extensions(phovo_engine *e, const phovo_extensions *ext)
{
  if (!e || !ext) return fail(PHOVO_E_INVALID_ARGUMENT, "set_extensions: null");
  if (ext->plane_storage != PHOVO_STORAGE_F64 && ext->plane_storage != PHOVO_STORAGE_F32 &&
      ext->plane_storage != PHOVO_STORAGE_F16)
    return fail(PHOVO_E_INVALID_ARGUMENT, "set_extensions: unknown plane_storage");
  if (ext->sampling != PHOVO_SAMPLING_NEAREST_SCATTER && ext->sampling != PHOVO_SAMPLING_BILINEAR)
    return fail(PHOVO_E_INVALID_ARGUMENT, "set_extensions: unknown sampling");
  if (ext->jacobian_corrected != 0 && ext->sampling != PHOVO_SAMPLING_BILINEAR)
    return fail(PHOVO_E_UNSUPPORTED, "set_extensions: jacobian_corrected needs sampling = PHOVO_SAMPLING_BILINEAR "
                                     "(the scatter path is kept reference-exact)");
  for (int l = 0; l < PHOVO_MAX_LEVELS; l++)
    if (!(ext->huber_delta[l] == ext->huber_delta[l])) return fail(PHOVO_E_INVALID_ARGUMENT, "set_extensions: huber_delta is NaN");
  // the pool layout changes with the storage type, and -- fp16 planes under bilinear sampling carry tap records -- with the
  // sampling where that adds or removes the records
  auto has_records = [](const phovo_extensions &x) { return x.sampling == PHOVO_SAMPLING_BILINEAR && x.plane_storage == PHOVO_STORAGE_F16; };
  if (ext->plane_storage != e->ext.plane_storage || has_records(*ext) != has_records(e->ext)) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)quiesce(e);
    free_pool(e);
  }
  e->ext = *ext;
  return PHOVO_OK;
}

int phovo_engine_get_extensions(const phovo_engine *e, phovo_extensions *ext)
{
  if (!e || !ext) return fail(PHOVO_E_INVALID_ARGUMENT, "get_extensions: null");
  *ext = e->ext;
  return PHOVO_OK;
}

int phovo_engine_get_config(const phovo_engine *e, phovo_config *cfg)
{
  if (!e || !cfg) return fail(PHOVO_E_INVALID_ARGUMENT, "get_config: null");
  *cfg = e->cfg;
  return PHOVO_OK;
}

int phovo_engine_set_intrinsic_matrix(phovo_engine *e, const double k[9])
{
  if (!e || !k) return fail(PHOVO_E_INVALID_ARGUMENT, "set_intrinsic_matrix: null");
  std::memcpy(e->K, k, sizeof(e->K));
  e->have_K = true;
  return PHOVO_OK;
}

int phovo_engine_set_depth_range(phovo_engine *e, double min_depth, double max_depth)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_depth_range: null");
  e->min_depth = min_depth;
  e->max_depth = max_depth;
  return PHOVO_OK;
}

int phovo_engine_set_wide_policy(phovo_engine *e, int policy)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_wide_policy: null");
  if (policy < -1 || policy > 1) return fail(PHOVO_E_INVALID_ARGUMENT, "set_wide_policy: policy must be -1, 0 or 1");
  e->wide_policy = policy;
  return PHOVO_OK;
}

int phovo_engine_set_slide_policy(phovo_engine *e, int policy)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_slide_policy: null");
  if (policy < -1 || policy > 0) return fail(PHOVO_E_INVALID_ARGUMENT, "set_slide_policy: policy must be -1 or 0");
  e->slide_policy = policy;
  return PHOVO_OK;
}

int phovo_engine_set_level_fusion(phovo_engine *e, int mode)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_level_fusion: null");
  if (mode != PHOVO_FUSION_AUTO && mode != PHOVO_FUSION_OFF && mode != PHOVO_FUSION_SPLIT)
    return fail(PHOVO_E_INVALID_ARGUMENT, "set_level_fusion: unknown mode");
  e->fusion = mode;
  return PHOVO_OK;
}

int phovo_engine_set_latency_forms(phovo_engine *e, int on)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_latency_forms: null");
  e->latency_forms = on != 0;
  return PHOVO_OK;
}

int phovo_engine_set_batch_invariant(phovo_engine *e, int on)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_batch_invariant: null");
  e->batch_invariant = on != 0;
  return PHOVO_OK;
}

int phovo_engine_level_uses_wide(const phovo_engine *e, int level, int n_pairs)
{
  if (!e || level < 0 || level >= e->cfg.num_levels || e->n_frames == 0) return 0;
  return use_wide_level(e, n_pairs, e->levels[level]) && !(e->ext.huber_delta[level] > 0.0) ? 1 : 0;
}

int phovo_engine_set_build_all_levels(phovo_engine *e, int on)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "set_build_all_levels: null");
  if ((on != 0) != e->build_all) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)quiesce(e);
    free_pool(e);
  }
  e->build_all = on != 0;
  return PHOVO_OK;
}

// The ranges this library has page-locked, by start address.  The contract of the two entry points is the library's,
// not the runtime's (which, depending on its version, accepts a second registration of a range or an unregister of a
// pointer it never saw without saying so): a range that overlaps a registered one is refused, and so is an unregister
// of anything but the start of a registered range.
static std::mutex g_registry_mutex;
static std::map<uintptr_t, size_t> g_registry;

int phovo_host_register(void *ptr, size_t bytes)
{
  if (!ptr || bytes == 0) return fail(PHOVO_E_INVALID_ARGUMENT, "host_register: null or empty");
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  if (a + bytes < a) return fail(PHOVO_E_INVALID_ARGUMENT, "host_register: the range wraps around the address space");
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  auto next = g_registry.lower_bound(a);                    // first registered range starting at or behind `a`
  if (next != g_registry.end() && next->first < a + bytes)
    return fail(PHOVO_E_INVALID_ARGUMENT, "host_register: the range overlaps one that is already registered");
  if (next != g_registry.begin()) {
    auto prev = std::prev(next);
    if (prev->first + prev->second > a)
      return fail(PHOVO_E_INVALID_ARGUMENT, "host_register: the range overlaps one that is already registered");
  }
  PHOVO_HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  g_registry[a] = bytes;
  return PHOVO_OK;
}

int phovo_host_unregister(void *ptr)
{
  if (!ptr) return fail(PHOVO_E_INVALID_ARGUMENT, "host_unregister: null");
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  auto it = g_registry.find(reinterpret_cast<uintptr_t>(ptr));
  if (it == g_registry.end())
    return fail(PHOVO_E_INVALID_ARGUMENT, "host_unregister: not the start of a range registered with phovo_host_register");
  PHOVO_HIP_CHECK(hipHostUnregister(ptr));
  g_registry.erase(it);
  return PHOVO_OK;
}

int phovo_engine_reserve_frames(phovo_engine *e, int n_frames, int width, int height)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "reserve_frames: null");
  if (n_frames < 1 || width < 1 || height < 1) return fail(PHOVO_E_INVALID_ARGUMENT, "reserve_frames: sizes must be positive");
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  PHOVO_HIP_CHECK(quiesce(e));
  free_pool(e);
  size_t max_n = 0;
  int max_ks = 0;
  for (int l = 0; l < e->cfg.num_levels; l++) {
    LevelPool &lv = e->levels[l];
    level_dims(width, height, l, &lv.w, &lv.h);
    if (lv.w < 1 || lv.h < 1) { free_pool(e); return fail(PHOVO_E_SHAPE, "image too small for the number of pyramid levels"); }
    lv.n = lv.w * lv.h;
    lv.stored = e->build_all || e->cfg.max_num_iterations[l] > 0;
    const bool fp64_planes = e->ext.plane_storage == PHOVO_STORAGE_F64;
    lv.plan_ok = gn_plan_level(lv.n, &lv.plan, 0, fp64_planes);
    lv.plan_few_ok = gn_plan_level(lv.n, &lv.plan_few, 1, fp64_planes);
    {   // byte layout of one frame at this level: planes I, D, GX, GY.  fp64: packed [4][n] doubles, which is
        // what the producer kernels write directly; narrow storages: every plane starts 16-byte aligned.
      // Bilinear sampling (extension): behind the planes every frame carries its tap records (pyr_build_tap_records), so
      // the frame is no longer what the producers write in one piece.
      const bool records = e->ext.sampling == PHOVO_SAMPLING_BILINEAR && e->ext.plane_storage == PHOVO_STORAGE_F16;      // (fp16 planes only: gn_bilinear_kernel.hip)
      const bool packed = e->ext.plane_storage == PHOVO_STORAGE_F64 && !records;
      size_t off = 0;
      for (int p = 0; p < PLANES_PER_FRAME; p++) {
        lv.plane_off[p] = off;
        const size_t bytes = storage_elem_size(e->ext.plane_storage, p == PLANE_D) * (size_t)lv.n;
        off += packed ? bytes : ((bytes + 15) & ~(size_t)15);
      }
      lv.rec_off = 0;
      if (records) {
        off = (off + 31) & ~(size_t)31;
        lv.rec_off = off;
        off += 4 * storage_elem_size(e->ext.plane_storage, false) * (size_t)lv.n;
        off = (off + 31) & ~(size_t)31;
      }
      lv.frame_bytes = off;
    }
    if (lv.stored) {
      const size_t bytes = (size_t)n_frames * lv.frame_bytes;
      hipError_t he = hipMalloc(&lv.planes, bytes);
      if (he != hipSuccess) { free_pool(e); return fail(PHOVO_E_HIP, std::string("hipMalloc(frame pool): ") + hipGetErrorString(he)); }
      he = hipMemsetAsync(lv.planes, 0, bytes, e->stream);
      if (he != hipSuccess) { free_pool(e); return fail(PHOVO_E_HIP, std::string("hipMemset(frame pool): ") + hipGetErrorString(he)); }
      if ((size_t)lv.n > max_n) max_n = (size_t)lv.n;
    }
    if (e->cfg.blur_filter_size[l] > max_ks) max_ks = e->cfg.blur_filter_size[l];
  }
  hipError_t he = hipSuccess;        // raw-frame staging is allocated on first upload (ensure_stage)
  {   // fp64 scratch: one staging chunk of planes for the narrow storages, one frame for plane get/set otherwise
    const size_t frames = e->ext.plane_storage == PHOVO_STORAGE_F64 ? 1 : (size_t)STAGE_CHUNK;
    he = hipMalloc(&e->d_scratch, sizeof(double) * frames * PLANES_PER_FRAME * (max_n ? max_n : 1));
  }
  if (he == hipSuccess && max_ks > 0) {
    // (a level-0 blur runs at full resolution even when level 0 is not resident)
    const size_t tmp_n = e->cfg.blur_filter_size[0] > 0 ? (size_t)width * (size_t)height : max_n;
    he = hipMalloc(&e->d_tmp, sizeof(double) * (tmp_n ? tmp_n : 1));
    if (he == hipSuccess) he = hipMalloc(&e->d_blur_kernel, sizeof(double) * (size_t)max_ks * PHOVO_MAX_LEVELS);
    if (he == hipSuccess) {
      // getGaussianKernel(k, sigma = 3, CV_64F): exp(-0.5/sigma^2 * (i-(k-1)/2)^2), normalised.
      e->blur_kernel_stride = max_ks;
      std::vector<double> kern((size_t)max_ks * PHOVO_MAX_LEVELS, 0.0);
      for (int l = 0; l < e->cfg.num_levels; l++) {
        const int ks = e->cfg.blur_filter_size[l];
        if (ks <= 0) continue;
        const double sigma = 3.0, scale2x = -0.5 / (sigma * sigma);
        double sum = 0;
        double *kk = kern.data() + (size_t)l * max_ks;
        for (int i = 0; i < ks; i++) {
          const double x = i - (ks - 1) * 0.5;
          const double t = std::exp(scale2x * x * x);
          kk[i] = t; sum += t;
        }
        sum = 1. / sum;
        for (int i = 0; i < ks; i++) kk[i] *= sum;
      }
      he = hipMemcpy(e->d_blur_kernel, kern.data(), sizeof(double) * kern.size(), hipMemcpyHostToDevice);
    }
  }
  if (he != hipSuccess) { free_pool(e); return fail(PHOVO_E_HIP, std::string("hipMalloc(staging): ") + hipGetErrorString(he)); }
  e->n_frames = n_frames; e->width = width; e->height = height;
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  return PHOVO_OK;
}

int phovo_engine_level_size(const phovo_engine *e, int level, int *width, int *height)
{
  if (!e || !width || !height) return fail(PHOVO_E_INVALID_ARGUMENT, "level_size: null");
  if (level < 0 || level >= e->cfg.num_levels) return fail(PHOVO_E_INVALID_ARGUMENT, "level out of range");
  if (e->n_frames == 0) return fail(PHOVO_E_NOT_READY, "no frame pool reserved");
  *width = e->levels[level].w;
  *height = e->levels[level].h;
  return PHOVO_OK;
}

int phovo_engine_level_is_stored(const phovo_engine *e, int level)
{
  if (!e || level < 0 || level >= e->cfg.num_levels || e->n_frames == 0) return 0;
  return e->levels[level].stored ? 1 : 0;
}

// Copies `count` frames (rows `stride` bytes apart, frames `frame_stride` bytes apart) into a packed
// staging buffer.
static int stage_frames_h2d(phovo_engine *e, void *dst, const void *src, size_t stride, size_t frame_stride,
                            size_t elem, int count, hipStream_t stream)
{
  const size_t row = elem * (size_t)e->width, frame_bytes = row * (size_t)e->height;
  if (stride == row && (frame_stride == frame_bytes || count == 1)) {
    PHOVO_HIP_CHECK(hipMemcpyAsync(dst, src, frame_bytes * (size_t)count, hipMemcpyHostToDevice, stream));
    return PHOVO_OK;
  }
  for (int f = 0; f < count; f++) {
    const int st = copy_rows_to_device(static_cast<char *>(dst) + frame_bytes * (size_t)f,
                                       static_cast<const char *>(src) + frame_stride * (size_t)f, stride, row,
                                       e->height, stream);
    if (st != PHOVO_OK) return st;
  }
  return PHOVO_OK;
}

static int upload_batch(phovo_engine *e, int first_frame, int count, int roles,
                        const uint8_t *intensity, size_t istride, size_t iframe_stride,
                        const void *depth, size_t dstride, size_t dframe_stride, DepthKind kind, double scale)
{
  if (!e || !intensity) return fail(PHOVO_E_INVALID_ARGUMENT, "upload_frames: null");
  if (e->n_frames == 0) return fail(PHOVO_E_NOT_READY, "upload_frames: reserve_frames first");
  if (count < 0 || first_frame < 0 || first_frame + count > e->n_frames)
    return fail(PHOVO_E_INVALID_ARGUMENT, "upload_frames: frame range out of the reserved pool");
  if ((roles & PHOVO_ROLE_BOTH) == 0) return fail(PHOVO_E_INVALID_ARGUMENT, "upload_frames: roles empty");
  if ((roles & PHOVO_ROLE_SOURCE) && !depth) return fail(PHOVO_E_INVALID_ARGUMENT, "upload_frames: a source frame needs depth");
  if (!(roles & PHOVO_ROLE_SOURCE)) kind = DEPTH_NONE;
  if (count == 0) return PHOVO_OK;
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(quiesce(e));                 // (an alignment in flight may be reading the frames about to be replaced)
  const int chunk_cap = count < STAGE_CHUNK ? count : STAGE_CHUNK;
  // Two staging halves: while the pyramid kernels of chunk i read one half on the engine's stream, chunk i + 1 is copied
  // into the other on the copy stream (events order each half: copied -> built -> copied again).
  const bool two_halves = count > chunk_cap;
  int st = ensure_stage(e, two_halves ? 2 * chunk_cap : chunk_cap, kind == DEPTH_F64, kind == DEPTH_U16);
  if (st != PHOVO_OK) return st;
  const size_t px = (size_t)e->width * (size_t)e->height;
  // whatever the engine's stream still does with the staging buffers (a previous upload) is over first
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  // an error half way must not leave a copy from the caller's memory in flight when this function returns
  auto drain = [&](int status) {
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamSynchronize(e->stream);
    (void)quiesce(e);
    return status;
  };
#define PHOVO_UPLOAD_CHECK(expr)                                                                              \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess) return drain(fail(PHOVO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  int chunk = 0;
  for (int done = 0; done < count; done += chunk_cap, chunk++) {
    const int c = count - done < chunk_cap ? count - done : chunk_cap;
    const int half = two_halves ? (chunk & 1) : 0;
    const int off = half * chunk_cap;
    if (chunk >= 2) PHOVO_UPLOAD_CHECK(hipStreamWaitEvent(e->copy_stream, e->ev_built[half], 0));     // the half is free again
    st = stage_frames_h2d(e, e->d_gray + px * (size_t)off, intensity + iframe_stride * (size_t)done, istride, iframe_stride, 1, c,
                          e->copy_stream);
    if (st != PHOVO_OK) return drain(st);
    if (kind == DEPTH_F64)
      st = stage_frames_h2d(e, e->d_depth + px * (size_t)off, static_cast<const char *>(depth) + dframe_stride * (size_t)done,
                            dstride, dframe_stride, sizeof(double), c, e->copy_stream);
    else if (kind == DEPTH_U16)
      st = stage_frames_h2d(e, e->d_depth16 + px * (size_t)off, static_cast<const char *>(depth) + dframe_stride * (size_t)done,
                            dstride, dframe_stride, sizeof(uint16_t), c, e->copy_stream);
    if (st != PHOVO_OK) return drain(st);
    PHOVO_UPLOAD_CHECK(hipEventRecord(e->ev_copied[half], e->copy_stream));
    PHOVO_UPLOAD_CHECK(hipStreamWaitEvent(e->stream, e->ev_copied[half], 0));
    st = build_pyramids(e, first_frame + done, c, roles, kind, scale, off);
    if (st != PHOVO_OK) return drain(st);
    PHOVO_UPLOAD_CHECK(hipEventRecord(e->ev_built[half], e->stream));
  }
#undef PHOVO_UPLOAD_CHECK
  // the caller's buffers may be reused on return, and the staging buffers by the next upload
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->copy_stream));
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  return PHOVO_OK;
}

int phovo_engine_upload_frame(phovo_engine *e, int frame, int roles,
                              const uint8_t *intensity, size_t intensity_stride,
                              const double *depth, size_t depth_stride)
{
  return upload_batch(e, frame, 1, roles, intensity, intensity_stride, 0, depth, depth_stride, 0, DEPTH_F64, 1.0);
}

int phovo_engine_upload_frame_u16(phovo_engine *e, int frame, int roles,
                                  const uint8_t *intensity, size_t intensity_stride,
                                  const uint16_t *depth, size_t depth_stride, double depth_scale)
{
  return upload_batch(e, frame, 1, roles, intensity, intensity_stride, 0, depth, depth_stride, 0, DEPTH_U16, depth_scale);
}

int phovo_engine_upload_frames(phovo_engine *e, int first_frame, int count, int roles,
                               const uint8_t *intensity, size_t intensity_stride, size_t intensity_frame_stride,
                               const double *depth, size_t depth_stride, size_t depth_frame_stride)
{
  return upload_batch(e, first_frame, count, roles, intensity, intensity_stride, intensity_frame_stride,
                      depth, depth_stride, depth_frame_stride, DEPTH_F64, 1.0);
}

int phovo_engine_upload_frames_u16(phovo_engine *e, int first_frame, int count, int roles,
                                   const uint8_t *intensity, size_t intensity_stride, size_t intensity_frame_stride,
                                   const uint16_t *depth, size_t depth_stride, size_t depth_frame_stride,
                                   double depth_scale)
{
  return upload_batch(e, first_frame, count, roles, intensity, intensity_stride, intensity_frame_stride,
                      depth, depth_stride, depth_frame_stride, DEPTH_U16, depth_scale);
}

static int plane_access_check(const phovo_engine *e, int frame, int level)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "level planes: null engine");
  if (e->n_frames == 0) return fail(PHOVO_E_NOT_READY, "level planes: reserve_frames first");
  if (frame < 0 || frame >= e->n_frames) return fail(PHOVO_E_INVALID_ARGUMENT, "level planes: frame index out of range");
  if (level < 0 || level >= e->cfg.num_levels) return fail(PHOVO_E_INVALID_ARGUMENT, "level planes: level out of range");
  if (!e->levels[level].stored) return fail(PHOVO_E_NOT_READY, "level planes: level is not resident (max_num_iterations == 0 and build_all_levels off)");
  return PHOVO_OK;
}

int phovo_engine_set_level_planes(phovo_engine *e, int frame, int level,
                                  const double *intensity, const double *depth,
                                  const double *grad_x, const double *grad_y)
{
  const int st = plane_access_check(e, frame, level);
  if (st != PHOVO_OK) return st;
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(quiesce(e));
  const LevelPool &lv = e->levels[level];
  unsigned char *base = lv.planes + (size_t)frame * lv.frame_bytes;
  const double *srcs[4] = {intensity, depth, grad_x, grad_y};
  for (int p = 0; p < 4; p++) {
    if (!srcs[p]) continue;
    if (e->ext.plane_storage == PHOVO_STORAGE_F64) {
      PHOVO_HIP_CHECK(hipMemcpyAsync(base + lv.plane_off[p], srcs[p], sizeof(double) * (size_t)lv.n,
                                     hipMemcpyHostToDevice, e->stream));
    } else {                                   // round to the storage type on the device, like the producers do
      PHOVO_HIP_CHECK(hipMemcpyAsync(e->d_scratch, srcs[p], sizeof(double) * (size_t)lv.n, hipMemcpyHostToDevice, e->stream));
      PHOVO_HIP_CHECK(pyr_store_plane(e->d_scratch, 0, 1, lv.n, base + lv.plane_off[p], lv.frame_bytes,
                                      e->ext.plane_storage, p == PLANE_D, e->stream));
      PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
  }
  if (lv.rec_off && (intensity || grad_x || grad_y))                    // bilinear sampling: this frame's tap records follow its planes
    PHOVO_HIP_CHECK(pyr_build_tap_records(lv.planes + (size_t)frame * lv.frame_bytes, lv.frame_bytes, lv.plane_off, lv.rec_off,
                                          1, lv.n, e->ext.plane_storage, e->stream));
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  return PHOVO_OK;
}

int phovo_engine_get_level_planes(const phovo_engine *e, int frame, int level,
                                  double *intensity, double *depth, double *grad_x, double *grad_y)
{
  const int st = plane_access_check(e, frame, level);
  if (st != PHOVO_OK) return st;
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  const LevelPool &lv = e->levels[level];
  const unsigned char *base = lv.planes + (size_t)frame * lv.frame_bytes;
  double *dsts[4] = {intensity, depth, grad_x, grad_y};
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  for (int p = 0; p < 4; p++) {
    if (!dsts[p]) continue;
    if (e->ext.plane_storage == PHOVO_STORAGE_F64) {
      PHOVO_HIP_CHECK(hipMemcpy(dsts[p], base + lv.plane_off[p], sizeof(double) * (size_t)lv.n, hipMemcpyDeviceToHost));
    } else {
      PHOVO_HIP_CHECK(pyr_load_plane(base + lv.plane_off[p], lv.n, e->d_scratch, e->ext.plane_storage, p == PLANE_D, e->stream));
      PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
      PHOVO_HIP_CHECK(hipMemcpy(dsts[p], e->d_scratch, sizeof(double) * (size_t)lv.n, hipMemcpyDeviceToHost));
    }
  }
  return PHOVO_OK;
}

int phovo_engine_enqueue_align(phovo_engine *e, int n_pairs, const int *source_frames,
                               const int *target_frames, const double *init_states)
{
  if (!e || !source_frames || !target_frames) return fail(PHOVO_E_INVALID_ARGUMENT, "align: null");
  if (n_pairs < 0) return fail(PHOVO_E_INVALID_ARGUMENT, "align: n_pairs < 0");
  if (e->n_frames == 0) return fail(PHOVO_E_NOT_READY, "align: no frames uploaded");
  if (!e->have_K) return fail(PHOVO_E_NOT_READY, "align: SetIntrinsicMatrix has not been called");
  for (int i = 0; i < n_pairs; i++) {
    if (source_frames[i] < 0 || source_frames[i] >= e->n_frames || target_frames[i] < 0 || target_frames[i] >= e->n_frames)
      return fail(PHOVO_E_INVALID_ARGUMENT, "align: frame index out of range");
  }
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  // Every active level must be launchable BEFORE the enqueue takes a ticket and a slot: a refused enqueue leaves the
  // engine as it was -- in particular the results of the enqueue before the last stay fetchable.
  size_t owner_need = 0, wide_need = 0, mask_need = 0;
  for (int l = 0; l < e->cfg.num_levels && n_pairs > 0; l++) {
    if (e->cfg.max_num_iterations[l] <= 0) continue;
    const LevelPool &lv = e->levels[l];
    if (!lv.stored) return fail(PHOVO_E_NOT_READY, "align: an active level is not resident");
    if (e->ext.sampling == PHOVO_SAMPLING_BILINEAR) continue;       // no owner map, no LDS limit
    const bool wide = use_wide_level(e, n_pairs, lv) && !(e->ext.huber_delta[l] > 0.0);
    if (wide) {
      const size_t ws = gn_wide_workspace_bytes(lv.n, n_pairs);
      if (ws > wide_need) wide_need = ws;
    }
    if (!wide && !lv.plan_ok)
      return fail(PHOVO_E_SHAPE, "align: pyramid level too large for the device path (more than 2 097 151 pixels: the owner map in "
                                 "HBM holds 21-bit source indices); up to 32 pairs at a time take the wide form, which has no such limit");
    if (wide || !lv.plan.owner_in_lds) {
      const size_t need = (size_t)n_pairs * (size_t)lv.n;
      if (need > owner_need) owner_need = need;
    }
    if (!wide && lv.plan.mask_in_hbm) {
      const size_t need = (size_t)n_pairs * (size_t)((lv.n + 63) / 64);
      if (need > mask_need) mask_need = need;
    }
  }
  // The slot this enqueue takes: the one used least recently.  Its previous enqueue (ticket - PHOVO_ENQUEUE_DEPTH) must
  // have finished -- its pinned mirror and pair data are about to be rewritten -- but the enqueue before this one may
  // still be running: it lives in the other slot, on the other stream.
  const int ticket = e->ticket + 1;
  AlignSlot &s = e->slots[ticket % PHOVO_ENQUEUE_DEPTH];
  PHOVO_HIP_CHECK(hipStreamSynchronize(s.stream));
  // From here on the slot's previous content is gone.  It holds NO enqueue until this one has been issued completely: a
  // failure on the way (an allocation, a launch) leaves a slot that every ticket-taking entry point refuses, instead of one
  // that hands out whatever an earlier enqueue left in its buffers.
  s.ticket = 0;
  s.last_pairs = n_pairs;
  s.have_timing = false;
  s.launches.clear();
  for (bool &b : s.level_launched) b = false;
  s.d_states = nullptr;
  if (n_pairs == 0) {                    // nothing to run: a valid, empty enqueue (fetch of 0 pairs succeeds, no device buffer)
    e->ticket = ticket;
    s.ticket = ticket;
    return PHOVO_OK;
  }
  // planes written on the engine's own stream so far (uploads return synchronised; plane setters too): nothing to order
  int st = ensure_pairs(s, n_pairs);
  if (st != PHOVO_OK) return st;

  // Scratch in HBM (owner maps, ballots, the wide form's workspace) belongs to a slot, because two enqueues in flight must
  // not share it -- but a caller that never has two in flight (align_pairs, the apps: one enqueue, one fetch) alternates
  // between the slots and would keep TWO copies of buffers that reach gigabytes on level 0 (8192 pairs x 307 200 pixels:
  // 10 GB of owner map).  So a slot that needs more than it has first looks at the OTHER slot: if that one is idle, its
  // buffer changes hands (nothing of it is in use); only with two enqueues really in flight does a second buffer appear.
  AlignSlot &other = e->slots[(ticket + 1) % PHOVO_ENQUEUE_DEPTH];
  const bool other_idle = other.stream && hipStreamQuery(other.stream) == hipSuccess;
  if (owner_need > s.owner_capacity && other_idle && other.owner_capacity >= owner_need) {
    std::swap(s.d_owner, other.d_owner); std::swap(s.owner_capacity, other.owner_capacity); std::swap(s.owner_tagged, other.owner_tagged);
  }
  if (mask_need > s.mask_capacity && other_idle && other.mask_capacity >= mask_need) {
    std::swap(s.d_mask, other.d_mask); std::swap(s.mask_capacity, other.mask_capacity);
  }
  if (wide_need > s.wide_ws_capacity && other_idle && other.wide_ws_capacity >= wide_need) {
    std::swap(s.d_wide_ws, other.d_wide_ws); std::swap(s.wide_ws_capacity, other.wide_ws_capacity);
  }
  if (owner_need > s.owner_capacity) {
    if (s.d_owner) { (void)hipFree(s.d_owner); s.d_owner = nullptr; s.owner_capacity = 0; }
    if (other_idle && other.d_owner) { (void)hipFree(other.d_owner); other.d_owner = nullptr; other.owner_capacity = 0; }      // (too small for this batch: do not keep it beside the new one)
    PHOVO_HIP_CHECK(hipMalloc(&s.d_owner, sizeof(int) * owner_need));
    s.owner_capacity = owner_need;
    s.owner_tagged = false;
    // -1 everywhere once: the wide form restores -1 after every iteration, the persistent kernel wipes per pair
    PHOVO_HIP_CHECK(fill_i32(s.d_owner, owner_need, -1, s.stream));
  }

  if (mask_need > s.mask_capacity) {
    if (s.d_mask) { (void)hipFree(s.d_mask); s.d_mask = nullptr; s.mask_capacity = 0; }
    if (other_idle && other.d_mask) { (void)hipFree(other.d_mask); other.d_mask = nullptr; other.mask_capacity = 0; }
    PHOVO_HIP_CHECK(hipMalloc(&s.d_mask, sizeof(unsigned long long) * mask_need));
    s.mask_capacity = mask_need;
  }
  if (wide_need > s.wide_ws_capacity) {
    if (s.d_wide_ws) { (void)hipFree(s.d_wide_ws); s.d_wide_ws = nullptr; s.wide_ws_capacity = 0; }
    if (other_idle && other.d_wide_ws) { (void)hipFree(other.d_wide_ws); other.d_wide_ws = nullptr; other.wide_ws_capacity = 0; }
    PHOVO_HIP_CHECK(hipMalloc(&s.d_wide_ws, wide_need));
    s.wide_ws_capacity = wide_need;
  }
  if (wide_need) s.h_wide_done.resize((size_t)n_pairs * 4);
  // The caller may reuse its arrays as soon as this returns and the copies below are asynchronous: the slot's pinned
  // mirror keeps them alive until the slot is used again (synchronised above).
  const PairLayout pl = pair_layout(n_pairs);
  s.d_src = reinterpret_cast<int *>(s.d_pairs + pl.src);
  s.d_tgt = reinterpret_cast<int *>(s.d_pairs + pl.tgt);
  s.d_states = reinterpret_cast<double *>(s.d_pairs + pl.states);
  s.d_reports = reinterpret_cast<phovo_pair_report *>(s.d_pairs + pl.reports);
  s.d_work_counters = reinterpret_cast<int *>(s.d_pairs + pl.heads);
  s.d_handover = reinterpret_cast<int *>(s.d_pairs + pl.handover);
  std::memset(s.h_up, 0, pl.reports);
  std::memcpy(s.h_up + pl.src, source_frames, sizeof(int) * (size_t)n_pairs);
  std::memcpy(s.h_up + pl.tgt, target_frames, sizeof(int) * (size_t)n_pairs);
  if (init_states)                                                                   // SetInitialStateVector  :494
    std::memcpy(s.h_up + pl.states, init_states, sizeof(double) * 6 * (size_t)n_pairs);
  // one copy in (pair list + initial states, zeros without them), one memset (reports + the work-queue heads of all levels)
  PHOVO_HIP_CHECK(hipMemcpyAsync(s.d_pairs, s.h_up, pl.reports, hipMemcpyHostToDevice, s.stream));
  PHOVO_HIP_CHECK(hipMemsetAsync(s.d_pairs + pl.reports, 0, pl.total - pl.reports, s.stream));
  PHOVO_HIP_CHECK(hipEventRecord(s.ev_total_start, s.stream));

  const PairLayout lay = pair_layout(n_pairs);
  auto record = [&](int first, int last, int kind, int threads, int lds, int wgs) {
    phovo_launch_record r{};
    r.level_first = first; r.level_last = last; r.kind = kind; r.threads = threads; r.lds_bytes = lds; r.workgroups = wgs;
    s.launches.push_back(r);
  };
  auto persistent_grid = [&](int wgs_per_cu) { const int slots = e->cu_count * wgs_per_cu; return n_pairs < slots ? n_pairs : slots; };
  auto level_args = [&](int l) {
    const LevelPool &lv = e->levels[l];
    GNLevelArgs a{};
    a.w = lv.w; a.h = lv.h; a.n = lv.n; a.level = l;
    a.max_iter = e->cfg.max_num_iterations[l];
    a.n_chunks = (lv.n + 63) / 64;
    a.lambda = e->cfg.lambda_optimization_step[l];
    a.min_grad_norm = e->cfg.min_gradient_norm[l];
    const double scaleFactor = 1.0 / std::pow(2, l);                                 // :203
    a.fx = e->K[0] * scaleFactor; a.fy = e->K[4] * scaleFactor;                      // :204-207
    a.ox = e->K[2] * scaleFactor; a.oy = e->K[5] * scaleFactor;
    a.ifx = 1.f / a.fx; a.ify = 1.f / a.fy;                                          // :208-209
    a.min_depth = e->min_depth; a.max_depth = e->max_depth;
    a.planes = lv.planes;
    a.frame_bytes = lv.frame_bytes;
    for (int p = 0; p < PLANES_PER_FRAME; p++) a.plane_off[p] = lv.plane_off[p];
    a.huber_delta = e->ext.huber_delta[l];
    a.rec_off = lv.rec_off;
    a.src = s.d_src; a.tgt = s.d_tgt;
    a.states = s.d_states; a.reports = s.d_reports;
    a.g_owner = s.d_owner;
    a.n_pairs = n_pairs;
    a.work_counter = s.d_work_counters + l * QUEUE_HEADS_INTS;
    // one queue per XCD once there are enough pairs to keep every XCD's share of the grid busy
    a.n_queues = (!tuning_switch("PHOVO_QUEUE_SINGLE") && n_pairs >= 8 * 64) ? QUEUES_PER_LEVEL : 1;
    return a;
  };
  // a handful of pairs leaves most CUs empty: such a batch takes the geometry with the shorter iteration
  const bool few_batch = e->latency_forms && !e->batch_invariant && n_pairs <= LATENCY_PAIRS;
  // Can level l be one of several levels of ONE launch (gn_fused_kernel)?  The persistent scatter kernel with its owner map
  // in half a CU's LDS; not the latency geometry of a small batch, not the wide form.
  auto fusable = [&](int l) {
    const LevelPool &lv = e->levels[l];
    if (e->fusion == PHOVO_FUSION_OFF || e->ext.sampling == PHOVO_SAMPLING_BILINEAR || few_batch) return false;
    if (use_wide_level(e, n_pairs, lv) && !(e->ext.huber_delta[l] > 0.0)) return false;
    return gn_level_fusable(lv.n);
  };

  int split_until = -1;        // PHOVO_FUSION_SPLIT: the levels down to this one belong to the run whose first launch has gone out
  for (int l = e->cfg.num_levels - 1; l >= 0; l--) {                                 // coarse to fine  :502-503
    if (e->cfg.max_num_iterations[l] <= 0) continue;                                 // :526 (nothing observable happens)
    const LevelPool &lv = e->levels[l];
    GNLevelArgs a = level_args(l);
    PHOVO_HIP_CHECK(hipEventRecord(s.ev_start[l], s.stream));

    // Data-dependent termination (a gradient threshold on any of them): the run of consecutive fusable levels that starts
    // here goes out as ONE persistent launch in which every pair flows through those levels inside the workgroup that drew
    // it.  Which levels form a run depends on the configuration and the level sizes only, never on the batch (apart from
    // the latency forms of batches of <= 8 pairs, which batch_invariant switches off): see phovo_engine_set_batch_invariant.
    // With thresholds of zero every pair runs max_num_iterations and a level boundary costs nothing: one launch per level,
    // each in its own best geometry.
    if (fusable(l)) {
      int run[GN_MAX_FUSED_LEVELS], n_run = 0;
      bool data_dependent = false;
      for (int m = l; m >= 0 && n_run < GN_MAX_FUSED_LEVELS; m--) {
        if (e->cfg.max_num_iterations[m] <= 0) continue;
        if (!fusable(m)) break;
        run[n_run++] = m;
        if (e->cfg.min_gradient_norm[m] > 0.0) data_dependent = true;
      }
      if (l >= split_until && split_until >= 0) { n_run = 2; data_dependent = true; }        // a later level of a split run
      else if (n_run >= 2 && data_dependent && e->fusion == PHOVO_FUSION_SPLIT) split_until = run[n_run - 1];
      if (n_run >= 2 && data_dependent && e->fusion == PHOVO_FUSION_AUTO) {
        GNFusedArgs f{};
        f.n_levels = n_run; f.n_pairs = n_pairs; f.n_queues = a.n_queues; f.work_counter = a.work_counter;
        for (int i = 0; i < n_run; i++) {
          f.lv[i] = level_args(run[i]);
          if (f.lv[i].n > f.n_max) f.n_max = f.lv[i].n;
        }
        PHOVO_HIP_CHECK(gn_launch_fused(f, e->ext.plane_storage, e->cu_count, s.stream));
        record(l, run[n_run - 1], PHOVO_LAUNCH_FUSED, 512, gn_fused_lds_bytes(f.n_max), persistent_grid(2));
        PHOVO_HIP_CHECK(hipEventRecord(s.ev_stop[l], s.stream));
        s.level_launched[l] = true;
        l = run[n_run - 1];                        // (the loop's l-- moves on to the level below the run)
        continue;
      }
      if (n_run >= 2 && data_dependent && e->fusion == PHOVO_FUSION_SPLIT) {
        // the fused geometry, one launch per level: what a fused run is bit-identical to (tests)
        GNLaunchPlan mid{};
        (void)gn_plan_fused_geometry(lv.n, &mid);
        PHOVO_HIP_CHECK(gn_launch_level(a, mid, e->ext.plane_storage, e->cu_count, s.stream));
        record(l, l, PHOVO_LAUNCH_PERSISTENT, mid.threads, mid.lds_bytes, persistent_grid(mid.wgs_per_cu));
        PHOVO_HIP_CHECK(hipEventRecord(s.ev_stop[l], s.stream));
        s.level_launched[l] = true;
        continue;
      }
    }

    if (e->ext.sampling == PHOVO_SAMPLING_BILINEAR) {
      PHOVO_HIP_CHECK(gn_launch_level_bilinear(a, e->ext.plane_storage, e->ext.jacobian_corrected != 0, e->cu_count, s.stream));
      record(l, l, PHOVO_LAUNCH_BILINEAR, 256, 0, persistent_grid(gn_bilinear_wgs_per_cu(e->ext.plane_storage)));
    } else if (use_wide_level(e, n_pairs, lv) && !(e->ext.huber_delta[l] > 0.0)) {
      if (s.owner_tagged) {            // the wide form starts from -1 everywhere and leaves it so
        PHOVO_HIP_CHECK(fill_i32(s.d_owner, s.owner_capacity, -1, s.stream));
        s.owner_tagged = false;
      }
      PHOVO_HIP_CHECK(gn_run_level_wide(a, n_pairs, s.d_wide_ws, s.h_wide_done.data(), s.stream));
      record(l, l, PHOVO_LAUNCH_WIDE, 256, 0, n_pairs * ((lv.n + 1023) / 1024));
    } else {
      const bool few = few_batch && lv.plan_few_ok && lv.plan_few.owner_in_lds == lv.plan.owner_in_lds;
      const GNLaunchPlan &pl = few ? lv.plan_few : lv.plan;
      a.n_lds = pl.owner_in_lds ? 0 : pl.owner_lds_entries;
      a.g_mask = pl.mask_in_hbm ? s.d_mask : nullptr;
      a.depth_lds_chunks = pl.depth_lds_chunks;
      int *list0 = s.d_handover + (size_t)l * lay.handover_stride;
      int *heads1 = s.d_work_counters + (PHOVO_MAX_LEVELS + l) * QUEUE_HEADS_INTS;
      // (tuning build: the sliding-window kernel also on levels whose owner map fits LDS, from 160x120 up -- an A/B, profiles/r04_runs)
      const bool slide_anyway = tuning_switch("PHOVO_SLIDE_FROM_160x120") && lv.n >= 19200 && !few;
      if ((!pl.owner_in_lds || slide_anyway) && e->slide_policy >= 0) {
        // Owner map too large for LDS: the sliding-window kernel first (owner ring in LDS); pairs whose warp leaves its
        // window are put on the hand-over list and continued, from the iteration they had reached, by the exact kernel
        // right behind it, which draws from that list.
        a.handover_out = list0;
        a.slide_m = gn_slide_reach_bands(lv.w, lv.h);
        PHOVO_HIP_CHECK(gn_launch_level_slide(a, e->ext.plane_storage, e->cu_count, s.stream));
        record(l, l, PHOVO_LAUNCH_SLIDE, gn_slide_threads(), (int)gn_slide_lds_bytes(), persistent_grid(1));
        a.handover_out = nullptr; a.handover_in = list0; a.takeover_flag = PHOVO_PAIR_WINDOW_FALLBACK;
        a.work_counter = heads1; a.n_queues = 1;
        PHOVO_HIP_CHECK(gn_launch_level(a, pl, e->ext.plane_storage, e->cu_count, s.stream));
        record(l, l, PHOVO_LAUNCH_SLIDE_FALLBACK, pl.threads, pl.lds_bytes, persistent_grid(pl.wgs_per_cu));
        s.owner_tagged = true;                              // tagged entries stay behind (the kernel wipes per pair)
      } else {
        PHOVO_HIP_CHECK(gn_launch_level(a, pl, e->ext.plane_storage, e->cu_count, s.stream));
        record(l, l, PHOVO_LAUNCH_PERSISTENT, pl.threads, pl.lds_bytes, persistent_grid(pl.wgs_per_cu));
        if (!pl.owner_in_lds) s.owner_tagged = true;
      }
    }
    PHOVO_HIP_CHECK(hipEventRecord(s.ev_stop[l], s.stream));
    s.level_launched[l] = true;
  }
  PHOVO_HIP_CHECK(hipEventRecord(s.ev_total_stop, s.stream));
  s.have_timing = true;
  e->ticket = ticket;                  // issued completely: the enqueue exists
  s.ticket = ticket;
  return PHOVO_OK;
}

// The slot that holds enqueue `ticket`, or null when that enqueue never happened or its slot has been taken over since.
static AlignSlot *slot_of(phovo_engine *e, int ticket)
{
  if (!e || ticket <= 0 || ticket > e->ticket) return nullptr;
  AlignSlot &s = e->slots[ticket % PHOVO_ENQUEUE_DEPTH];
  return s.ticket == ticket ? &s : nullptr;
}
static const AlignSlot *slot_of(const phovo_engine *e, int ticket) { return slot_of(const_cast<phovo_engine *>(e), ticket); }
static const char *const STALE_TICKET = "no such enqueue in flight (tickets stay valid until PHOVO_ENQUEUE_DEPTH later enqueues)";

int phovo_engine_last_ticket(const phovo_engine *e) { return e ? e->ticket : 0; }

int phovo_engine_last_launches(const phovo_engine *e, phovo_launch_record *out, int capacity, int *count)
{
  if (!e || !count) return fail(PHOVO_E_INVALID_ARGUMENT, "last_launches: null");
  const AlignSlot *s = slot_of(e, e->ticket);
  *count = s ? (int)s->launches.size() : 0;
  if (out && s)
    for (int i = 0; i < capacity && i < *count; i++) out[i] = s->launches[(size_t)i];
  return PHOVO_OK;
}

int phovo_engine_synchronize(phovo_engine *e)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "synchronize: null");
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(hipStreamSynchronize(e->stream));
  PHOVO_HIP_CHECK(quiesce(e));
  return PHOVO_OK;
}

int phovo_engine_wait(phovo_engine *e, int ticket)
{
  AlignSlot *s = slot_of(e, ticket);
  if (!s) return fail(PHOVO_E_INVALID_ARGUMENT, std::string("wait: ") + STALE_TICKET);
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(hipStreamSynchronize(s->stream));
  return PHOVO_OK;
}

int phovo_engine_fetch(phovo_engine *e, int ticket, int n_pairs, double *out_states, phovo_pair_report *reports)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "fetch: null");
  AlignSlot *s = slot_of(e, ticket);
  if (!s) return fail(PHOVO_E_INVALID_ARGUMENT, std::string("fetch: ") + STALE_TICKET);
  if (n_pairs != s->last_pairs) return fail(PHOVO_E_INVALID_ARGUMENT, "fetch: n_pairs differs from that enqueue's");
  if (n_pairs == 0) return PHOVO_OK;
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  const PairLayout pl = pair_layout(n_pairs);
  // one copy out: states (and reports right behind them, when asked for) into pinned memory
  const size_t bytes = (reports ? pl.heads : pl.reports) - pl.states;
  PHOVO_HIP_CHECK(hipMemcpyAsync(s->h_down, s->d_pairs + pl.states, bytes, hipMemcpyDeviceToHost, s->stream));
  PHOVO_HIP_CHECK(hipStreamSynchronize(s->stream));
  if (out_states) std::memcpy(out_states, s->h_down, sizeof(double) * 6 * (size_t)n_pairs);
  if (reports) {
    std::memcpy(reports, s->h_down + (pl.reports - pl.states), sizeof(phovo_pair_report) * (size_t)n_pairs);
    // Levels with max_num_iterations == 0 still run the loop body once in the reference (:510,547-549).
    for (int i = 0; i < n_pairs; i++)
      for (int l = 0; l < e->cfg.num_levels; l++)
        if (e->cfg.max_num_iterations[l] <= 0) reports[i].iterations[l] = 1;
  }
  return PHOVO_OK;
}

int phovo_engine_fetch_results(phovo_engine *e, int n_pairs, double *out_states, phovo_pair_report *reports)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "fetch_results: null");
  if (e->ticket == 0) return n_pairs == 0 ? PHOVO_OK : fail(PHOVO_E_INVALID_ARGUMENT, "fetch_results: nothing has been enqueued");
  return phovo_engine_fetch(e, e->ticket, n_pairs, out_states, reports);
}

int phovo_engine_device_states(phovo_engine *e, int ticket, void **states)
{
  if (!e || !states) return fail(PHOVO_E_INVALID_ARGUMENT, "device_states: null");
  AlignSlot *s = slot_of(e, ticket);
  if (!s) return fail(PHOVO_E_INVALID_ARGUMENT, std::string("device_states: ") + STALE_TICKET);
  *states = s->d_states;
  return PHOVO_OK;
}

int phovo_engine_results_device_ptr(phovo_engine *e, void **states)
{
  if (!e || !states) return fail(PHOVO_E_INVALID_ARGUMENT, "results_device_ptr: null");
  const AlignSlot *s = slot_of(e, e->ticket);
  *states = s ? s->d_states : nullptr;
  return PHOVO_OK;
}

int phovo_engine_align_pairs(phovo_engine *e, int n_pairs, const int *source_frames,
                             const int *target_frames, const double *init_states,
                             double *out_states, phovo_pair_report *reports)
{
  int st = phovo_engine_enqueue_align(e, n_pairs, source_frames, target_frames, init_states);
  if (st != PHOVO_OK) return st;
  return phovo_engine_fetch_results(e, n_pairs, out_states, reports);
}

int phovo_engine_align_ms(const phovo_engine *e, int ticket, double *total_ms, double level_ms[PHOVO_MAX_LEVELS])
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "align_ms: null");
  const AlignSlot *s = slot_of(e, ticket);
  if (!s || !s->have_timing) return fail(PHOVO_E_NOT_READY, "align_ms: nothing has been enqueued under that ticket");
  PHOVO_HIP_CHECK(hipSetDevice(e->device));
  PHOVO_HIP_CHECK(hipStreamSynchronize(s->stream));
  for (int l = 0; l < PHOVO_MAX_LEVELS; l++) {
    double ms = 0;
    if (s->level_launched[l]) {
      float f = 0;
      PHOVO_HIP_CHECK(hipEventElapsedTime(&f, s->ev_start[l], s->ev_stop[l]));
      ms = f;
    }
    if (level_ms) level_ms[l] = ms;
  }
  if (total_ms) {           // first launch to last
    float f = 0;
    PHOVO_HIP_CHECK(hipEventElapsedTime(&f, s->ev_total_start, s->ev_total_stop));
    *total_ms = f;
  }
  return PHOVO_OK;
}

int phovo_engine_last_align_ms(const phovo_engine *e, double *total_ms, double level_ms[PHOVO_MAX_LEVELS])
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "last_align_ms: null");
  if (e->ticket == 0) return fail(PHOVO_E_NOT_READY, "last_align_ms: nothing has been enqueued");
  return phovo_engine_align_ms(e, e->ticket, total_ms, level_ms);
}

int phovo_engine_level_launch_info(const phovo_engine *e, int level, int *threads, int *lds_bytes,
                                   int *owner_in_lds, int *source_in_lds)
{
  if (!e) return fail(PHOVO_E_INVALID_ARGUMENT, "level_launch_info: null");
  if (level < 0 || level >= e->cfg.num_levels) return fail(PHOVO_E_INVALID_ARGUMENT, "level out of range");
  if (e->n_frames == 0) return fail(PHOVO_E_NOT_READY, "no frame pool reserved");
  const LevelPool &lv = e->levels[level];
  if (!lv.plan_ok) return fail(PHOVO_E_SHAPE, "level too large for the device path");
  if (!lv.plan.owner_in_lds && e->slide_policy >= 0) {       // the sliding-window kernel runs first on such a level
    if (threads) *threads = gn_slide_threads();
    if (lds_bytes) *lds_bytes = (int)gn_slide_lds_bytes();
    if (owner_in_lds) *owner_in_lds = 0;                     // a ring of 32768 entries, not the whole map
    if (source_in_lds) *source_in_lds = 0;
    return PHOVO_OK;
  }
  if (threads) *threads = lv.plan.threads;
  if (lds_bytes) *lds_bytes = lv.plan.lds_bytes;
  if (owner_in_lds) *owner_in_lds = lv.plan.owner_in_lds ? 1 : 0;
  if (source_in_lds) *source_in_lds = lv.plan.source_in_lds ? 1 : 0;
  return PHOVO_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------ single pair ------------- */

struct phovo_odometry {
  phovo_engine *engine = nullptr;
  bool have_source = false, have_target = false, optimized = false;
  double init_state[6] = {0, 0, 0, 0, 0, 0};
  double state[6] = {0, 0, 0, 0, 0, 0};          // m_StateVector.setZero()  :432
  phovo_pair_report report{};
};

namespace {

int odometry_set_frame(phovo_odometry *o, int slot, int role, const uint8_t *intensity, size_t istride,
                       const double *depth, size_t dstride, int width, int height)
{
  if (!o || !intensity) return fail(PHOVO_E_INVALID_ARGUMENT, "Set*Frame: null");
  if (width < 1 || height < 1) return fail(PHOVO_E_INVALID_ARGUMENT, "Set*Frame: empty image");
  phovo_engine *e = o->engine;
  if (e->n_frames == 0 || e->width != width || e->height != height) {
    const int st = phovo_engine_reserve_frames(e, 2, width, height);
    if (st != PHOVO_OK) return st;
    o->have_source = o->have_target = false;
  }
  const int st = phovo_engine_upload_frame(e, slot, role, intensity, istride, depth, dstride);
  if (st != PHOVO_OK) return st;
  if (slot == 0) o->have_source = true; else o->have_target = true;
  o->optimized = false;
  return PHOVO_OK;
}

}  // namespace

extern "C" {

int phovo_odometry_create(int device, phovo_odometry **out)
{
  if (!out) return fail(PHOVO_E_INVALID_ARGUMENT, "phovo_odometry_create: null");
  *out = nullptr;
  phovo_engine *e = nullptr;
  const int st = phovo_engine_create(device, &e);
  if (st != PHOVO_OK) return st;
  phovo_odometry *o = new (std::nothrow) phovo_odometry();
  if (!o) { phovo_engine_destroy(e); return fail(PHOVO_E_INVALID_ARGUMENT, "out of host memory"); }
  o->engine = e;
  *out = o;
  return PHOVO_OK;
}

int phovo_odometry_destroy(phovo_odometry *o)
{
  if (!o) return PHOVO_OK;
  phovo_engine_destroy(o->engine);
  delete o;
  return PHOVO_OK;
}

int phovo_odometry_set_config(phovo_odometry *o, const phovo_config *cfg)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "set_config: null");
  const int st = phovo_engine_set_config(o->engine, cfg);
  if (st == PHOVO_OK) o->have_source = o->have_target = o->optimized = false;
  return st;
}

int phovo_odometry_set_extensions(phovo_odometry *o, const phovo_extensions *ext)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "set_extensions: null");
  const int before = o->engine->ext.plane_storage;
  const int st = phovo_engine_set_extensions(o->engine, ext);
  if (st == PHOVO_OK && ext->plane_storage != before) o->have_source = o->have_target = o->optimized = false;
  return st;
}

int phovo_odometry_set_latency_forms(phovo_odometry *o, int on)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "set_latency_forms: null");
  return phovo_engine_set_latency_forms(o->engine, on);
}

int phovo_odometry_read_configuration_file(phovo_odometry *o, const char *path)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "ReadConfigurationFile: null");
  phovo_config c;
  int st = read_config_file(path, &c);
  if (st != PHOVO_OK) return st;
  phovo_extensions x;                        // optional keys; a reference yml has none -> everything stays off
  st = read_extensions_file(path, &x);
  if (st != PHOVO_OK) return st;
  st = phovo_odometry_set_extensions(o, &x);
  if (st != PHOVO_OK) return st;
  return phovo_odometry_set_config(o, &c);
}

int phovo_odometry_set_min_depth(phovo_odometry *o, double v)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "SetMinDepth: null");
  o->engine->min_depth = v;
  return PHOVO_OK;
}

int phovo_odometry_set_max_depth(phovo_odometry *o, double v)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "SetMaxDepth: null");
  o->engine->max_depth = v;
  return PHOVO_OK;
}

int phovo_odometry_set_intrinsic_matrix(phovo_odometry *o, const double k[9])
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "SetIntrinsicMatrix: null");
  return phovo_engine_set_intrinsic_matrix(o->engine, k);
}

int phovo_odometry_set_source_frame(phovo_odometry *o, const uint8_t *intensity, size_t istride,
                                    const double *depth, size_t dstride, int width, int height)
{
  if (!depth) return fail(PHOVO_E_INVALID_ARGUMENT, "SetSourceFrame: depth is required");
  return odometry_set_frame(o, 0, PHOVO_ROLE_SOURCE, intensity, istride, depth, dstride, width, height);
}

int phovo_odometry_set_target_frame(phovo_odometry *o, const uint8_t *intensity, size_t istride,
                                    const double *depth, size_t dstride, int width, int height)
{
  (void)depth; (void)dstride;                      // "Depth image is ignored"  :478
  return odometry_set_frame(o, 1, PHOVO_ROLE_TARGET, intensity, istride, nullptr, 0, width, height);
}

int phovo_odometry_set_initial_state_vector(phovo_odometry *o, const double state[6])
{
  if (!o || !state) return fail(PHOVO_E_INVALID_ARGUMENT, "SetInitialStateVector: null");
  std::memcpy(o->init_state, state, sizeof(o->init_state));
  std::memcpy(o->state, state, sizeof(o->state));   // m_StateVector = initialStateVector  :496
  return PHOVO_OK;
}

// visualizeIterations, headless (...Analytic.h:515-517,359-362,551-557).  The reference fills a warped source image
// while it computes the residuals -- warped(tr, tc) = I0(r, c) for every source pixel that passes the depth gate and lands
// in bounds, last raster writer wins, zero elsewhere -- and after every iteration that does NOT end the level shows
// |I1 - warped| in a window and waits for a key.  Here, when the configuration asks for it AND PHOVO_VISUALIZE_DIR names a
// directory, Optimize() runs one iteration per launch (same kernels, same arithmetic as the fused loop: the state a launch
// ends with is the state the next one starts from) and writes that image as
//   <dir>/optimize_imgDiff_level<L>_iteration<N>.pgm   (8 bit: |diff| * 255, rounded; what imshow does to a [0, 1] image)
// Display code, on the host and outside every timed region; without the directory the key is parsed and ignored as before.
static int write_iteration_image(phovo_engine *e, int level, int iteration, const double pre_state[6], const char *dir)
{
  const LevelPool &lv = e->levels[level];
  std::vector<double> i0((size_t)lv.n), d0((size_t)lv.n), i1((size_t)lv.n), warped((size_t)lv.n, 0.0);
  int st = phovo_engine_get_level_planes(e, 0, level, i0.data(), d0.data(), nullptr, nullptr);
  if (st != PHOVO_OK) return st;
  st = phovo_engine_get_level_planes(e, 1, level, i1.data(), nullptr, nullptr, nullptr);
  if (st != PHOVO_OK) return st;
  double rt[16];
  phovo_eigen_pose(pre_state, rt);
  const double sc = 1.0 / std::pow(2, level);
  const double fx = e->K[0] * sc, fy = e->K[4] * sc, ox = e->K[2] * sc, oy = e->K[5] * sc;
  const double ifx = 1.f / fx, ify = 1.f / fy;
  for (int r = 0; r < lv.h; r++)
    for (int c = 0; c < lv.w; c++) {
      const double pz = d0[(size_t)r * lv.w + c];
      if (!(e->min_depth < pz && pz < e->max_depth)) continue;                     // :280
      const double px = (c - ox) * pz * ifx, py = (r - oy) * pz * ify;            // :282-283
      const double X = ((rt[0] * px + rt[1] * py) + rt[2] * pz) + rt[3];           // :291
      const double Y = ((rt[4] * px + rt[5] * py) + rt[6] * pz) + rt[7];
      const double Z = ((rt[8] * px + rt[9] * py) + rt[10] * pz) + rt[11];
      const double iz = 1.0 / Z;
      const double rr = std::round((Y * fy) * iz + oy), rc = std::round((X * fx) * iz + ox);   // :294-298
      if (rr >= 0 && rr < (double)lv.h && rc >= 0 && rc < (double)lv.w)            // :302-303
        warped[(size_t)rr * lv.w + (size_t)rc] = i0[(size_t)r * lv.w + c];         // :361
    }
  std::vector<unsigned char> img((size_t)lv.n);
  for (int k = 0; k < lv.n; k++) {
    const double v = std::fabs(i1[(size_t)k] - warped[(size_t)k]) * 255.0;         // cv::absdiff  :554
    img[(size_t)k] = (unsigned char)(v >= 255.0 ? 255 : (int)std::lrint(v));
  }
  const std::string path = std::string(dir) + "/optimize_imgDiff_level" + std::to_string(level) + "_iteration" +
                           std::to_string(iteration) + ".pgm";
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) return fail(PHOVO_E_IO, "visualizeIterations: cannot write " + path);
  std::fprintf(f, "P5\n%d %d\n255\n", lv.w, lv.h);
  const bool ok = std::fwrite(img.data(), 1, img.size(), f) == img.size();
  std::fclose(f);
  return ok ? PHOVO_OK : fail(PHOVO_E_IO, "visualizeIterations: short write to " + path);
}

static int optimize_visualized(phovo_odometry *o, const char *dir)
{
  phovo_engine *e = o->engine;
  const phovo_config full = e->cfg;
  const int src = 0, tgt = 1;
  phovo_pair_report total{};
  int st = PHOVO_OK;
  for (int l = full.num_levels - 1; l >= 0 && st == PHOVO_OK; l--) {               // coarse to fine  :502-503
    total.iterations[l] = 1;                                                       // a level without iterations still runs the loop once
    if (full.max_num_iterations[l] <= 0) continue;
    phovo_config one = full;
    for (int m = 0; m < one.num_levels; m++) one.max_num_iterations[m] = m == l ? 1 : 0;
    one.visualize_iterations = 0;
    for (int it = 1;; it++) {
      double pre[6];
      std::memcpy(pre, o->state, sizeof(pre));
      // (levels that stay resident keep the pool: only max_num_iterations changes)
      e->cfg = one;
      phovo_pair_report rep{};
      st = phovo_engine_align_pairs(e, 1, &src, &tgt, o->state, o->state, &rep);
      e->cfg = full;
      if (st != PHOVO_OK) break;
      total.iterations[l] = it;
      total.gradient_norm = rep.gradient_norm;
      total.flags |= rep.flags;
      total.valid_pixels[l] = rep.valid_pixels[l];
      const bool stop = it >= full.max_num_iterations[l] || rep.gradient_norm < full.min_gradient_norm[l] ||
                        (rep.flags & PHOVO_PAIR_NONFINITE);                       // :383,388
      if (stop) break;
      st = write_iteration_image(e, l, it, pre, dir);                              // :551-557: only when the level goes on
      if (st != PHOVO_OK) break;
    }
  }
  e->cfg = full;
  if (st != PHOVO_OK) return st;
  o->report = total;
  o->optimized = true;
  return PHOVO_OK;
}

int phovo_odometry_optimize(phovo_odometry *o)
{
  if (!o) return fail(PHOVO_E_INVALID_ARGUMENT, "Optimize: null");
  if (!o->have_source || !o->have_target)
    return fail(PHOVO_E_NOT_READY, "Optimize: SetSourceFrame and SetTargetFrame must be called first");
  if (o->engine->cfg.visualize_iterations) {
    const char *dir = std::getenv("PHOVO_VISUALIZE_DIR");
    if (dir && *dir) return optimize_visualized(o, dir);
  }
  const int src = 0, tgt = 1;
  // Like the reference, Optimize() starts from the CURRENT state vector (:539 updates m_StateVector in place).
  const int st = phovo_engine_align_pairs(o->engine, 1, &src, &tgt, o->state, o->state, &o->report);
  if (st != PHOVO_OK) return st;
  o->optimized = true;
  return PHOVO_OK;
}

int phovo_odometry_get_optimal_state_vector(const phovo_odometry *o, double state[6])
{
  if (!o || !state) return fail(PHOVO_E_INVALID_ARGUMENT, "GetOptimalStateVector: null");
  std::memcpy(state, o->state, sizeof(o->state));
  return PHOVO_OK;
}

int phovo_odometry_get_optimal_rigid_transformation_matrix(const phovo_odometry *o, double rt[16])
{
  if (!o || !rt) return fail(PHOVO_E_INVALID_ARGUMENT, "GetOptimalRigidTransformationMatrix: null");
  return phovo_eigen_pose(o->state, rt);            // :572-578
}

int phovo_odometry_get_report(const phovo_odometry *o, phovo_pair_report *report)
{
  if (!o || !report) return fail(PHOVO_E_INVALID_ARGUMENT, "get_report: null");
  if (!o->optimized) return fail(PHOVO_E_NOT_READY, "get_report: Optimize has not run");
  *report = o->report;
  return PHOVO_OK;
}

int phovo_odometry_last_optimize_ms(const phovo_odometry *o, double *ms)
{
  if (!o || !ms) return fail(PHOVO_E_INVALID_ARGUMENT, "last_optimize_ms: null");
  if (!o->optimized) return fail(PHOVO_E_NOT_READY, "last_optimize_ms: Optimize has not run");
  return phovo_engine_last_align_ms(o->engine, ms, nullptr);
}

}  // extern "C"
