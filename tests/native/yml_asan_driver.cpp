// Sanitizer harness (CPU only) for the yml reader: csrc/yml_config.cpp compiled with g++ -fsanitize=address,undefined
// and fed the reference's files plus malformed input.
#include <cstdio>
#include <cstring>
#include <string>

#include "phovo_internal.hpp"

namespace phovo_hip {
static std::string g_err;
void set_last_error(const std::string &m) { g_err = m; }
int fail(int st, const std::string &m) { g_err = m; return st; }
}  // namespace phovo_hip

extern "C" int phovo_config_default(phovo_config *cfg)
{
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->num_levels = 5;
  return 0;
}

extern "C" int phovo_extensions_default(phovo_extensions *ext)
{
  std::memset(ext, 0, sizeof(*ext));
  return 0;
}

int main(int argc, char **argv)
{
  int bad = 0;
  for (int i = 1; i < argc; i++) {
    phovo_config c;
    int st = phovo_hip::read_config_file(argv[i], &c);
    phovo_extensions x;
    const int st2 = phovo_hip::read_extensions_file(argv[i], &x);
    if (st == 0) st = st2;
    std::printf("%s -> %d %s\n", argv[i], st, st ? phovo_hip::g_err.c_str() : "");
    if (st != 0) bad++;
  }
  return bad > 250 ? 250 : bad;
}
