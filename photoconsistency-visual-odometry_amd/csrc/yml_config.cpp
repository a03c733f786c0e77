// Reader for the reference's config_files/*.yml, unchanged.
//
// The reference reads them with cv::FileStorage (CPhotoconsistencyOdometryAnalytic.h:581-607).
// The files are OpenCV's "%YAML:1.0" dialect: a directive line that stock YAML parsers reject,
// then `key: value` lines whose keys contain spaces and parentheses, e.g.
//     max_num_iterations (at each level): [0, 0, 20, 50]
// Values are scalars or flow sequences.  Per-level sequences may be LONGER than
// numOptimizationLevels (config_only_level_0_analytic.yml:2-7); only the first
// numOptimizationLevels entries are used.  A missing key leaves the reference with an empty
// vector and undefined behaviour later (:142); here it is PHOVO_E_CONFIG.

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "phovo_internal.hpp"

namespace phovo_hip {

namespace {

std::string trim(const std::string &s)
{
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) a++;
  while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

bool parse_number(const std::string &tok, double *out)
{
  const std::string t = trim(tok);
  if (t.empty()) return false;
  char *end = nullptr;
  const double v = std::strtod(t.c_str(), &end);
  if (end == t.c_str()) return false;
  while (*end && std::isspace((unsigned char)*end)) end++;
  if (*end != '\0') return false;
  *out = v;
  return true;
}

bool parse_values(const std::string &value, std::vector<double> *out)
{
  out->clear();
  std::string v = trim(value);
  if (v.empty()) return false;
  if (v[0] == '[') {
    const size_t close = v.rfind(']');
    if (close == std::string::npos) return false;
    const std::string body = v.substr(1, close - 1);
    std::stringstream ss(body);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
      if (trim(tok).empty()) continue;
      double d;
      if (!parse_number(tok, &d)) return false;
      out->push_back(d);
    }
    return true;
  }
  double d;
  if (!parse_number(v, &d)) return false;
  out->push_back(d);
  return true;
}

// cv::FileNode -> int conversion rounds reals (cvRound).
int to_int(double v) { return (int)std::lrint(v); }

}  // namespace

int read_config_file(const char *path, phovo_config *cfg)
{
  if (!path || !cfg) return fail(PHOVO_E_INVALID_ARGUMENT, "read_config_file: null argument");
  std::ifstream in(path);
  if (!in.is_open()) return fail(PHOVO_E_IO, std::string("cannot open configuration file ") + path);

  std::map<std::string, std::string> kv;
  std::string line, pending_key, pending_val;
  bool open_seq = false;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (open_seq) {                                  // flow sequence continued on the next line
      pending_val += " " + line;
      if (line.find(']') != std::string::npos) { kv[pending_key] = pending_val; open_seq = false; }
      continue;
    }
    const std::string t = trim(line);
    if (t.empty() || t[0] == '#' || t[0] == '%' || t == "---" || t == "...") continue;
    // the key ends at the FIRST ':' that is followed by whitespace or end of line
    size_t colon = std::string::npos;
    for (size_t i = 0; i < t.size(); i++) {
      if (t[i] == ':' && (i + 1 == t.size() || std::isspace((unsigned char)t[i + 1]))) { colon = i; break; }
    }
    if (colon == std::string::npos) continue;
    const std::string key = trim(t.substr(0, colon));
    const std::string val = trim(t.substr(colon + 1));
    if (!val.empty() && val[0] == '[' && val.find(']') == std::string::npos) {
      open_seq = true; pending_key = key; pending_val = val;
      continue;
    }
    kv[key] = val;
  }
  if (open_seq) return fail(PHOVO_E_CONFIG, "unterminated sequence for key '" + pending_key + "'");

  auto get = [&](const char *key, std::vector<double> *out) -> bool {
    auto it = kv.find(key);
    if (it == kv.end()) { set_last_error(std::string("configuration key missing: '") + key + "'"); return false; }
    if (!parse_values(it->second, out)) {
      set_last_error(std::string("configuration key malformed: '") + key + "'");
      return false;
    }
    return true;
  };

  phovo_config c;
  phovo_config_default(&c);
  std::vector<double> v;
  if (!get("numOptimizationLevels", &v) || v.size() != 1) return PHOVO_E_CONFIG;            // :586
  c.num_levels = to_int(v[0]);
  if (c.num_levels < 1 || c.num_levels > PHOVO_MAX_LEVELS)
    return fail(PHOVO_E_CONFIG, "numOptimizationLevels out of range [1, 16]");

  struct IntKey { const char *key; int *dst; };
  struct DblKey { const char *key; double *dst; };
  const IntKey ints[] = {
      {"blurFilterSize (at each level)", c.blur_filter_size},                                  // :590
      {"max_num_iterations (at each level)", c.max_num_iterations},                            // :600
  };
  const DblKey dbls[] = {
      {"imageGradientsScalingFactor (at each level)", c.image_gradients_scaling_factor},       // :594
      {"lambda_optimization_step (at each level)", c.lambda_optimization_step},                // :597
      {"min_gradient_norm (at each level)", c.min_gradient_norm},                              // :603
  };
  for (const IntKey &k : ints) {
    if (!get(k.key, &v)) return PHOVO_E_CONFIG;
    if ((int)v.size() < c.num_levels)
      return fail(PHOVO_E_CONFIG, std::string("'") + k.key + "' has fewer entries than numOptimizationLevels");
    for (int i = 0; i < PHOVO_MAX_LEVELS && i < (int)v.size(); i++) k.dst[i] = to_int(v[i]);
  }
  for (const DblKey &k : dbls) {
    if (!get(k.key, &v)) return PHOVO_E_CONFIG;
    if ((int)v.size() < c.num_levels)
      return fail(PHOVO_E_CONFIG, std::string("'") + k.key + "' has fewer entries than numOptimizationLevels");
    for (int i = 0; i < PHOVO_MAX_LEVELS && i < (int)v.size(); i++) k.dst[i] = v[i];
  }
  if (!get("visualizeIterations", &v) || v.size() != 1) return PHOVO_E_CONFIG;                 // :606
  c.visualize_iterations = v[0] != 0.0 ? 1 : 0;
  *cfg = c;
  return PHOVO_OK;
}


// Optional extension keys (not in the reference).  Same line format; everything else in the file is ignored.
int read_extensions_file(const char *path, phovo_extensions *ext)
{
  if (!path || !ext) return fail(PHOVO_E_INVALID_ARGUMENT, "read_extensions_file: null argument");
  std::ifstream in(path);
  if (!in.is_open()) return fail(PHOVO_E_IO, std::string("cannot open configuration file ") + path);
  phovo_extensions e;
  phovo_extensions_default(&e);
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const std::string t = trim(line);
    if (t.empty() || t[0] == '#' || t[0] == '%') continue;
    size_t colon = std::string::npos;
    for (size_t i = 0; i < t.size(); i++)
      if (t[i] == ':' && (i + 1 == t.size() || std::isspace((unsigned char)t[i + 1]))) { colon = i; break; }
    if (colon == std::string::npos) continue;
    const std::string key = trim(t.substr(0, colon));
    std::vector<double> v;
    if (key == "huber_delta (at each level)") {
      if (!parse_values(t.substr(colon + 1), &v)) return fail(PHOVO_E_CONFIG, "configuration key malformed: 'huber_delta (at each level)'");
      for (int i = 0; i < PHOVO_MAX_LEVELS && i < (int)v.size(); i++) e.huber_delta[i] = v[i];
    } else if (key == "plane_storage_bits") {
      if (!parse_values(t.substr(colon + 1), &v) || v.size() != 1) return fail(PHOVO_E_CONFIG, "configuration key malformed: 'plane_storage_bits'");
      if (v[0] == 64) e.plane_storage = PHOVO_STORAGE_F64;
      else if (v[0] == 32) e.plane_storage = PHOVO_STORAGE_F32;
      else if (v[0] == 16) e.plane_storage = PHOVO_STORAGE_F16;
      else return fail(PHOVO_E_CONFIG, "plane_storage_bits must be 64, 32 or 16");
    } else if (key == "sampling_bilinear" || key == "jacobian_corrected") {
      if (!parse_values(t.substr(colon + 1), &v) || v.size() != 1 || (v[0] != 0 && v[0] != 1))
        return fail(PHOVO_E_CONFIG, "configuration key malformed: '" + key + "' must be 0 or 1");
      if (key == "sampling_bilinear") e.sampling = v[0] != 0 ? PHOVO_SAMPLING_BILINEAR : PHOVO_SAMPLING_NEAREST_SCATTER;
      else e.jacobian_corrected = v[0] != 0 ? 1 : 0;
    }
  }
  *ext = e;
  return PHOVO_OK;
}

}  // namespace phovo_hip
