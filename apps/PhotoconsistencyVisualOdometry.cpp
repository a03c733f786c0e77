// PhotoconsistencyVisualOdometry on the MI355X path: frame-to-frame odometry over a TUM-format RGB-D
// directory, writing a TUM trajectory file.
//
//   ./PhotoconsistencyVisualOdometry <config_file.yml> <rgbd_dataset_directory> <output_trajectory_file> [--batch [--gpus N] [--rccl]]
//
// Behaviour kept from the reference's app (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp):
//   * <dir>/rgb.txt and <dir>/depth.txt are read in lock step -- line n of one is paired with line n of the
//     other, there is NO timestamp association (phovo/include/CMultiSensorDataSource.h:74-91); '#' lines are
//     skipped and image paths are relative to the list file (CCameraRecord.h:74-108);
//   * intrinsics (517.3, 516.5, 318.6, 255.3) and depth scale 1/5000 are hard-coded (:163,170-173);
//   * every pair starts from the zero state (:175,224) -> pairs are independent;
//   * pose *= Rt^-1, quaternion from the rotation block, one line `timestamp tx ty tz qx qy qz qw` with 16
//     significant digits per pair, stamped with the CURRENT rgb timestamp (:233-243).
// Default mode goes pair by pair through the class surface exactly like the reference's loop and prints
// `Time = ... sec.` and `Rt:` for each.  --batch loads the whole sequence, builds every pyramid once on the
// GPU (the reference builds each frame's pyramids twice, :222-223) and aligns all pairs in one batched call;
// the trajectory is identical.  --batch --gpus N cuts the pairs into N contiguous ranges, one engine and one host thread
// per device (single process: the results meet in host memory; with --rccl through ONE RCCL all_gather from the engines'
// device buffers, apps/rccl/rccl_gather.cpp); the trajectory file is the same for every N and either way.  The
// one-process-per-GPU form with the RCCL all_gather is apps/PhotoconsistencyVisualOdometrySharded.py.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "io/png_io.h"
#include "rccl/rccl_gather.h"
#include "rccl/shard_vote.h"
#include "phovo/CPhotoconsistencyOdometryAnalytic.h"

typedef double CoordinateType;
typedef unsigned char PixelType;
typedef phovo::Numeric::Matrix33RowMajor<CoordinateType> Matrix33Type;
typedef phovo::Numeric::Matrix44RowMajor<CoordinateType> Matrix44Type;
typedef phovo::Numeric::VectorCol6<CoordinateType> Vector6Type;
typedef phovo::compat::Mat_<PixelType> IntensityImageType;
typedef phovo::compat::Mat_<CoordinateType> DepthImageType;

struct ListEntry { double timestamp; std::string path; };

static bool fileExists(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

static std::string parentDir(const std::string &p)
{
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

static bool readList(const std::string &listFile, std::vector<ListEntry> &out)
{
  std::ifstream in(listFile.c_str());
  if (!in.is_open()) { std::cerr << "Unable to open camera record file " << listFile << std::endl; return false; }
  const std::string dir = parentDir(listFile);
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream iss(line);
    ListEntry e;
    std::string name;
    if (!(iss >> e.timestamp >> name)) continue;
    e.path = dir + "/" + name;
    out.push_back(e);
  }
  return true;
}

static bool loadGray(const std::string &path, IntensityImageType &img)
{
  phovo_io::Image8 im; std::string err;
  if (!phovo_io::read_gray8(path, &im, &err)) { std::cerr << err << std::endl; return false; }
  img.create(im.height, im.width);
  for (size_t i = 0; i < im.pixels.size(); i++) img.data[i] = im.pixels[i];
  return true;
}

static bool loadDepth16(const std::string &path, phovo_io::Image16 &im)
{
  std::string err;
  if (!phovo_io::read_unchanged16(path, &im, &err)) { std::cerr << err << std::endl; return false; }
  return true;
}

// pose chain and trajectory line live behind the C ABI (phovo_trajectory_*): this loop, the --batch path and the
// multi-rank sequence driver write byte-identical files
static bool writePose(std::ofstream &f, double timestamp, const Matrix44Type &pose)
{
  char line[256];
  if (phovo_trajectory_format_pose(timestamp, pose.data(), line, sizeof(line)) != PHOVO_OK) return false;
  f << line << std::endl;                                                    // :240-243
  return true;
}

static void printHelp()
{
  std::cout << "./PhotoconsistencyVisualOdometry <config_file.yml> <rgbd_dataset_directory> "
               "<output_trajectory_file> [--batch [--gpus N] [--rccl]]" << std::endl;
}

#define PHOVO_OK_OR_FAIL(call)                                                              \
  do { if ((call) != PHOVO_OK) { std::cerr << #call << ": " << phovo_last_error() << std::endl; return EXIT_FAILURE; } } while (0)

int main(int argc, char *argv[])
{
  if (argc < 4) { printHelp(); return EXIT_FAILURE; }
  const std::string configFile(argv[1]), datasetDir(argv[2]), trajectoryPath(argv[3]);
  bool batch = false;
  int nGpus = 1;                                      // --batch --gpus N: the pairs of the sequence sharded over N devices
  bool rccl = false;                                  // ... --rccl: the shards' states meet through ONE RCCL all_gather
  for (int i = 4; i < argc; i++) {
    const std::string a(argv[i]);
    if (a == "--batch") batch = true;
    else if (a == "--rccl") rccl = true;
    else if (a == "--gpus" && i + 1 < argc) nGpus = std::atoi(argv[++i]);
    else { printHelp(); return EXIT_FAILURE; }
  }
  if (nGpus < 1 || (nGpus > 1 && !batch)) { std::cerr << "--gpus N needs --batch and N >= 1" << std::endl; return EXIT_FAILURE; }
  if (rccl && !batch) { std::cerr << "--rccl needs --batch" << std::endl; return EXIT_FAILURE; }
  if (!fileExists(configFile)) { std::cerr << "Input config file " << configFile << " does not exist" << std::endl; return EXIT_FAILURE; }
  if (!fileExists(datasetDir)) { std::cerr << "Input RGBD dataset directory " << datasetDir << " does not exist" << std::endl; return EXIT_FAILURE; }
  const std::string rgbList = datasetDir + "/rgb.txt", depthList = datasetDir + "/depth.txt";
  if (!fileExists(rgbList)) { std::cerr << "Input RGB data file " << rgbList << " does not exist" << std::endl; return EXIT_FAILURE; }
  if (!fileExists(depthList)) { std::cerr << "Input depth data file " << depthList << " does not exist" << std::endl; return EXIT_FAILURE; }
  const std::string outDir = parentDir(trajectoryPath);
  if (!fileExists(outDir) && ::mkdir(outDir.c_str(), 0777) != 0) {
    std::cerr << "Cannot create output directory " << outDir << std::endl;
    return EXIT_FAILURE;
  }

  const CoordinateType depthScalingFactor = 1. / 5000.;                    // :163
  Matrix33Type intrinsicMatrix;                                              // :170-173
  intrinsicMatrix << 517.3, 0., 318.6,
                     0., 516.5, 255.3,
                     0., 0., 1.;

  std::vector<ListEntry> rgb, depth;
  if (!readList(rgbList, rgb) || !readList(depthList, depth)) return EXIT_FAILURE;
  const size_t nFrames = rgb.size() < depth.size() ? rgb.size() : depth.size();   // lock step: stops at the shorter list

  std::ofstream trajectoryFile(trajectoryPath.c_str());
  if (!trajectoryFile.is_open()) { std::cerr << "Cannot open output trajectory file " << trajectoryPath << std::endl; return EXIT_FAILURE; }
  trajectoryFile << "# estimated trajectory" << std::endl;                    // :187-188
  trajectoryFile << "# timestamp tx ty tz qx qy qz qw" << std::endl;
  if (nFrames < 2) return EXIT_SUCCESS;

  Matrix44Type pose = Matrix44Type::Identity();
  try {
    if (!batch) {
      phovo::Analytic::CPhotoconsistencyOdometryAnalytic<PixelType, CoordinateType> odometry;
      odometry.ReadConfigurationFile(configFile);
      odometry.SetIntrinsicMatrix(intrinsicMatrix);
      IntensityImageType prevGray, curGray;
      DepthImageType prevDepth, curDepth;
      phovo_io::Image16 d16;
      if (!loadGray(rgb[0].path, prevGray) || !loadDepth16(depth[0].path, d16)) return EXIT_FAILURE;
      prevDepth.create(d16.height, d16.width);
      for (size_t i = 0; i < d16.pixels.size(); i++) prevDepth.data[i] = (double)d16.pixels[i] * depthScalingFactor;
      for (size_t t = 1; t < nFrames; t++) {
        if (!loadGray(rgb[t].path, curGray) || !loadDepth16(depth[t].path, d16)) return EXIT_FAILURE;
        curDepth.create(d16.height, d16.width);
        for (size_t i = 0; i < d16.pixels.size(); i++) curDepth.data[i] = (double)d16.pixels[i] * depthScalingFactor;

        odometry.SetSourceFrame(prevGray, prevDepth);                        // :222-224
        odometry.SetTargetFrame(curGray, curDepth);
        odometry.SetInitialStateVector(Vector6Type::Zero());
        const auto t0 = std::chrono::steady_clock::now();
        odometry.Optimize();
        const auto t1 = std::chrono::steady_clock::now();
        std::cout << "Time = " << std::chrono::duration<double>(t1 - t0).count() << " sec." << std::endl;

        const Matrix44Type Rt = odometry.GetOptimalRigidTransformationMatrix();
        const Vector6Type state = odometry.GetOptimalStateVector();
        PHOVO_OK_OR_FAIL(phovo_trajectory_chain(1, state.data(), pose.data(), nullptr));   // pose *= Rt^-1  :233-234
        if (!writePose(trajectoryFile, rgb[t].timestamp, pose)) return EXIT_FAILURE;
        std::cout << "Rt:" << std::endl << Rt << std::endl;
        prevGray = curGray.clone();
        prevDepth = curDepth.clone();
      }
    } else {
      phovo_config cfg;
      PHOVO_OK_OR_FAIL(phovo_config_read_file(configFile.c_str(), &cfg));
      const int nDevices = phovo_device_count();
      if (nDevices < 1) { std::cerr << "phovo_engine_create: no HIP device available: this library has no CPU path" << std::endl; return EXIT_FAILURE; }
      // PHOVO_VO_SHARE_DEVICES=1 lets more shards than devices run (a rehearsal of --gpus N on a smaller machine)
      if (nGpus > nDevices && !std::getenv("PHOVO_VO_SHARE_DEVICES")) {
        std::cerr << "--gpus " << nGpus << " but only " << nDevices << " device(s) are visible" << std::endl;
        return EXIT_FAILURE;
      }
      // Decoding is the bulk of this mode's wall time (a 640x480 colour PNG + its 16-bit depth PNG take 6-20 ms on one
      // core, the alignment of the whole sequence a few milliseconds on the GPU): frames are independent, so they
      // are decoded by all host threads at once, each straight into its slot of the two packed arrays.
      IntensityImageType gray;
      phovo_io::Image16 d16;
      std::vector<uint8_t> allGray;
      std::vector<uint16_t> allDepth;
      if (!loadGray(rgb[0].path, gray) || !loadDepth16(depth[0].path, d16)) return EXIT_FAILURE;
      const int W = gray.cols, H = gray.rows;
      if (d16.width != W || d16.height != H) { std::cerr << "frame 0: intensity and depth sizes differ" << std::endl; return EXIT_FAILURE; }
      allGray.resize((size_t)W * H * nFrames);
      allDepth.resize((size_t)W * H * nFrames);
      std::copy(gray.data, gray.data + (size_t)W * H, allGray.begin());
      std::copy(d16.pixels.begin(), d16.pixels.end(), allDepth.begin());
      {
        unsigned nThreads = std::thread::hardware_concurrency();
        if (nThreads == 0) nThreads = 1;
        if (nThreads > 32) nThreads = 32;
        if ((size_t)nThreads > nFrames) nThreads = (unsigned)nFrames;
        std::atomic<size_t> next(1);
        std::atomic<long> failed(-1);
        std::vector<std::string> errors(nThreads);
        auto worker = [&](unsigned id) {
          phovo_io::Image8 g8;
          phovo_io::Image16 g16;
          for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= nFrames || failed.load() >= 0) return;
            std::string err;
            if (!phovo_io::read_gray8(rgb[t].path, &g8, &err) || !phovo_io::read_unchanged16(depth[t].path, &g16, &err)) {
              errors[id] = err; failed.store((long)t); return;
            }
            if (g8.width != W || g8.height != H || g16.width != W || g16.height != H) {
              errors[id] = "frame " + std::to_string(t) + " has a different size"; failed.store((long)t); return;
            }
            std::copy(g8.pixels.begin(), g8.pixels.end(), allGray.begin() + (size_t)W * H * t);
            std::copy(g16.pixels.begin(), g16.pixels.end(), allDepth.begin() + (size_t)W * H * t);
          }
        };
        std::vector<std::thread> pool;
        for (unsigned id = 0; id < nThreads; id++) pool.emplace_back(worker, id);
        for (auto &th : pool) th.join();
        if (failed.load() >= 0) {
          for (const auto &e : errors) if (!e.empty()) std::cerr << e << std::endl;
          return EXIT_FAILURE;
        }
      }
      // Pair t aligns frame t with frame t+1 and depends on nothing else (:175,222-224): the pairs are cut into nGpus
      // contiguous ranges (the first nPairs % nGpus ranges one pair longer), each range gets its own engine on its own
      // device and its own host thread, and needs the frames of its range plus one.  In ONE process the results meet
      // in host memory: no collective (the multi-process form, PhotoconsistencyVisualOdometrySharded.py, is where the
      // RCCL all_gather is).  With phovo_engine_set_batch_invariant a pair's result does not depend on its batch, so the file is the same for every N.
      const int nPairs = (int)nFrames - 1;
      std::vector<double> states((size_t)nPairs * 6);
      std::vector<std::string> shardError(nGpus);
      // --rccl: instead of every shard copying its states to the host itself, the shards' device buffers meet in ONE RCCL
      // all_gather (one communicator and one host thread per device in this one process) and rank 0 copies the lot out --
      // the collective of SURVEY.md section 8e, with the host side staying C++.  Same bytes in the trajectory file.
      phovo_rccl::Group group;
      const int maxShard = nPairs / nGpus + (nPairs % nGpus ? 1 : 0);
      if (rccl) {
        std::vector<int> devices(nGpus);
        for (int g = 0; g < nGpus; g++) devices[g] = g % nDevices;
        std::string err;
        if (!group.create(devices, &err)) { std::cerr << err << std::endl; return EXIT_FAILURE; }
      }
      const auto t0 = std::chrono::steady_clock::now();
      // Every shard thread arrives at the vote exactly once -- also one that failed on its way there, and one whose range
      // is empty -- and the collective is entered only if all of them are ready for it: a rank that stays away from an
      // all_gather leaves the others waiting in it for ever.
      phovo_rccl::ShardVote vote(nGpus);
      // test hook: shard N reports a failure before it touches its device (tests: the run must end with that error, not hang)
      const int injectFailure = std::getenv("PHOVO_VO_INJECT_SHARD_FAILURE") ? std::atoi(std::getenv("PHOVO_VO_INJECT_SHARD_FAILURE")) : -1;
      auto alignShard = [&](int g) {
        const int base = nPairs / nGpus, extra = nPairs % nGpus;
        const int a = g * base + (g < extra ? g : extra), b = a + base + (g < extra ? 1 : 0);
        const int f0 = a, nf = b - a + 1;
        phovo_engine *engine = nullptr;
        void *dStates = nullptr;
        auto work = [&]() -> bool {                       // everything of this shard short of the collective
          auto fail = [&](const char *what) { shardError[g] = std::string(what) + ": " + phovo_last_error(); return false; };
          if (b <= a) return true;                        // more shards than pairs: nothing to align, still a rank of the gather
          if (injectFailure == g) { shardError[g] = "shard " + std::to_string(g) + ": failure injected (PHOVO_VO_INJECT_SHARD_FAILURE)"; return false; }
          if (phovo_engine_create(g % nDevices, &engine) != PHOVO_OK) return fail("phovo_engine_create");
          if (phovo_engine_set_config(engine, &cfg) != PHOVO_OK) return fail("phovo_engine_set_config");
          // a pair's pose must not depend on the size of the shard it falls into (same trajectory file for every N)
          if (phovo_engine_set_batch_invariant(engine, 1) != PHOVO_OK) return fail("phovo_engine_set_batch_invariant");
          if (phovo_engine_set_intrinsic_matrix(engine, intrinsicMatrix.data()) != PHOVO_OK) return fail("phovo_engine_set_intrinsic_matrix");
          if (phovo_engine_reserve_frames(engine, nf, W, H) != PHOVO_OK) return fail("phovo_engine_reserve_frames");
          if (phovo_engine_upload_frames_u16(engine, 0, nf, PHOVO_ROLE_BOTH, allGray.data() + (size_t)W * H * f0, (size_t)W,
                                             (size_t)W * H, allDepth.data() + (size_t)W * H * f0, sizeof(uint16_t) * (size_t)W,
                                             sizeof(uint16_t) * (size_t)W * H, depthScalingFactor) != PHOVO_OK)
            return fail("phovo_engine_upload_frames_u16");
          std::vector<int> src(b - a), tgt(b - a);
          for (int p = 0; p < b - a; p++) { src[p] = p; tgt[p] = p + 1; }
          if (!rccl) {
            if (phovo_engine_align_pairs(engine, b - a, src.data(), tgt.data(), nullptr, states.data() + (size_t)a * 6, nullptr) != PHOVO_OK)
              return fail("phovo_engine_align_pairs");
            return true;
          }
          if (phovo_engine_enqueue_align(engine, b - a, src.data(), tgt.data(), nullptr) != PHOVO_OK) return fail("phovo_engine_enqueue_align");
          if (phovo_engine_synchronize(engine) != PHOVO_OK) return fail("phovo_engine_synchronize");
          if (phovo_engine_results_device_ptr(engine, &dStates) != PHOVO_OK) return fail("phovo_engine_results_device_ptr");
          return true;
        };
        bool ok = work();
        if (rccl) {
          std::string err;
          if (ok && !group.stage(g, dStates, b > a ? b - a : 0, maxShard, &err)) { shardError[g] = err; ok = false; }
          if (vote.arrive(ok)) {                          // every rank is staged: all of them enter the collective
            if (!group.gather(g, &err)) shardError[g] = err;
          } else if (ok) {
            shardError[g] = "";                           // (another shard failed and says so; this one just stays out)
          }
        }
        if (engine) phovo_engine_destroy(engine);
      };
      {
        std::vector<std::thread> shards;
        for (int g = 1; g < nGpus; g++) shards.emplace_back(alignShard, g);
        alignShard(0);
        for (auto &th : shards) th.join();
      }
      for (const auto &e : shardError) if (!e.empty()) { std::cerr << e << std::endl; return EXIT_FAILURE; }
      if (rccl) {                          // rank 0's view of the gather: shard g's block holds its (b - a) x 6 states
        for (int g = 0; g < nGpus; g++) {
          const int base = nPairs / nGpus, extra = nPairs % nGpus;
          const int a = g * base + (g < extra ? g : extra), b = a + base + (g < extra ? 1 : 0);
          std::copy(group.gathered(g), group.gathered(g) + (size_t)(b - a) * 6, states.begin() + (size_t)a * 6);
        }
      }
      const auto t1 = std::chrono::steady_clock::now();
      std::cout << "Time = " << std::chrono::duration<double>(t1 - t0).count() << " sec. (" << nPairs << " pairs on " << nGpus
                << " device(s), upload and pyramids included)" << std::endl;
      std::vector<double> poses((size_t)nPairs * 16);
      PHOVO_OK_OR_FAIL(phovo_trajectory_chain(nPairs, states.data(), pose.data(), poses.data()));
      for (int p = 0; p < nPairs; p++) {
        Matrix44Type P;
        for (int i = 0; i < 16; i++) P(i) = poses[(size_t)p * 16 + i];
        if (!writePose(trajectoryFile, rgb[(size_t)p + 1].timestamp, P)) return EXIT_FAILURE;
      }
    }
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
  trajectoryFile.close();
  return EXIT_SUCCESS;
}
