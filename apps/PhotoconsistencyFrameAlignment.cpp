// PhotoconsistencyFrameAlignment on the MI355X path: aligns one RGB-D pair given as four PNG files.
//
// Same command line, hard-coded intrinsics, depth scale and console output as the reference's app
// (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:49-115):
//   ./PhotoconsistencyFrameAlignment <config_file.yml> <imgRGB0.png> <imgDepth0.png> <imgRGB1.png> <imgDepth1.png> [diff.png]
// Differences: the reference tests `argc<5` but reads argv[5] (:56,79) -- five arguments are required
// here; the final |I1 - warp(I0)| image goes to the optional sixth argument as a PNG instead of an
// imshow window (:107-112), because the target machines are headless.
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>

#include "io/png_io.h"
#include "phovo/CPhotoconsistencyOdometryAnalytic.h"

typedef double CoordinateType;
typedef unsigned char PixelType;
typedef phovo::Numeric::Matrix33RowMajor<CoordinateType> Matrix33Type;
typedef phovo::Numeric::Matrix44RowMajor<CoordinateType> Matrix44Type;
typedef phovo::Numeric::VectorCol6<CoordinateType> Vector6Type;
typedef phovo::compat::Mat_<PixelType> IntensityImageType;
typedef phovo::compat::Mat_<CoordinateType> DepthImageType;

static void printHelp()
{
  std::cout << "./PhotoconsistencyFrameAlignment <config_file.yml> <imgRGB0.png> <imgDepth0.png> "
               "<imgRGB1.png> <imgDepth1.png> [imgDiff.png]" << std::endl;
}

static bool loadGray(const char *path, IntensityImageType &img)
{
  phovo_io::Image8 im; std::string err;
  if (!phovo_io::read_gray8(path, &im, &err)) { std::cerr << err << std::endl; return false; }
  img.create(im.height, im.width);
  for (size_t i = 0; i < im.pixels.size(); i++) img.data[i] = im.pixels[i];
  return true;
}

// imread(-1) into a cv::Mat_<double>, then `* 1. / 1000.` : depth in metres (:75-76,79-80)
static bool loadDepthMetres(const char *path, DepthImageType &img)
{
  phovo_io::Image16 im; std::string err;
  if (!phovo_io::read_unchanged16(path, &im, &err)) { std::cerr << err << std::endl; return false; }
  img.create(im.height, im.width);
  const double scale = 1. / 1000.;
  for (size_t i = 0; i < im.pixels.size(); i++) img.data[i] = (double)im.pixels[i] * scale;
  return true;
}

int main(int argc, char **argv)
{
  if (argc < 6) { printHelp(); return -1; }

  Matrix33Type intrinsicMatrix;                       // :68-71
  intrinsicMatrix << 525., 0., 319.5,
                     0., 525., 239.5,
                     0., 0., 1.;

  IntensityImageType imgGray0, imgGray1;
  DepthImageType imgDepth0, imgDepth1;
  if (!loadGray(argv[2], imgGray0) || !loadDepthMetres(argv[3], imgDepth0) ||
      !loadGray(argv[4], imgGray1) || !loadDepthMetres(argv[5], imgDepth1))
    return EXIT_FAILURE;

  try {
    phovo::Analytic::CPhotoconsistencyOdometryAnalytic<PixelType, CoordinateType> photoconsistencyOdometry;
    Vector6Type stateVector;                          // x,y,z,yaw,pitch,roll = 0
    photoconsistencyOdometry.ReadConfigurationFile(std::string(argv[1]));       // :92
    photoconsistencyOdometry.SetIntrinsicMatrix(intrinsicMatrix);
    photoconsistencyOdometry.SetSourceFrame(imgGray0, imgDepth0);
    photoconsistencyOdometry.SetTargetFrame(imgGray1, imgDepth1);
    photoconsistencyOdometry.SetInitialStateVector(stateVector);

    const auto t0 = std::chrono::steady_clock::now();                           // cv::TickMeter  :99-102
    photoconsistencyOdometry.Optimize();
    const auto t1 = std::chrono::steady_clock::now();
    std::cout << "Time = " << std::chrono::duration<double>(t1 - t0).count() << " sec." << std::endl;

    // The reference prints a progress block per level when built with ENABLE_PRINT_CONSOLE_OPTIMIZATION_PROGRESS
    // (...Analytic.h:40,396-421; off as shipped).  Here the block is a run-time opt-in, printed after the fact from
    // the device's report, coarse to fine as Optimize() visits the levels (:502-503).
    if (std::getenv("PHOVO_PRINT_OPTIMIZATION_PROGRESS")) {
      phovo_config cfg;
      if (phovo_config_read_file(argv[1], &cfg) == PHOVO_OK) {
        const phovo_pair_report rep = photoconsistencyOdometry.GetReport();
        for (int level = cfg.num_levels - 1; level >= 0; level--) {
          std::cout << "----------------------------------------" << std::endl;
          std::cout << "Optimization level: " << level << std::endl;
          std::cout << "Number iterations: " << rep.iterations[level] << std::endl;
          // the report keeps the gradient norm of the LAST active level only (the reference's m_Gradients is likewise
          // overwritten level by level)
          int last_active = 0;
          for (int l = 0; l < cfg.num_levels; l++) if (cfg.max_num_iterations[l] > 0) { last_active = l; break; }
          if (level == last_active) std::cout << "gradient norm: " << rep.gradient_norm << std::endl;
          std::cout << "----------------------------------------" << std::endl;
        }
      }
    }

    Matrix44Type Rt = photoconsistencyOdometry.GetOptimalRigidTransformationMatrix();
    std::cout << "main::Rt eigen:" << std::endl << Rt << std::endl;

    if (argc > 6) {                                                             // :106-112, headless
      IntensityImageType warpedImage;
      phovo::warpImage<PixelType, CoordinateType>(imgGray0, imgDepth0, warpedImage, Rt, intrinsicMatrix);
      IntensityImageType imgDiff(imgGray1.rows, imgGray1.cols);
      for (size_t i = 0; i < (size_t)imgGray1.rows * imgGray1.cols; i++) {
        const int d = (int)imgGray1.data[i] - (int)warpedImage.data[i];        // cv::absdiff
        imgDiff.data[i] = (PixelType)(d < 0 ? -d : d);
      }
      std::string err;
      if (!phovo_io::write_gray8(argv[6], imgDiff.cols, imgDiff.rows, imgDiff.data, &err)) {
        std::cerr << err << std::endl;
        return EXIT_FAILURE;
      }
    }
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
  return 0;
}
