// host_tu.cpp -- CPU sanitizer build of the product's HOST-SIDE code (test infrastructure; SURVEY section 5:
// "-fsanitize=address host tests").  Plain g++, no HIP: the single-state rule helpers and the noise spec of the
// C-ABI (caro_host_*, caro_key_words, caro_action_space, caro_obs_cells) compiled from the very headers the
// kernels are compiled from -- caro_rules.h, caro_variants.h, caro_host.inc, include/caro_noise.h -- with
// -fsanitize=address,undefined.  `make -C oracle asan` builds it into oracle/_build/libcaro_host_asan.so;
// oracle/asan/plugin.py binds it in place of libcaro_hip.so for the host-helper tests (oracle/asan/run.sh).  Entry
// points that need a GPU do not exist in this library.
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/caro_hip.h"
#include "../../include/caro_noise.h"
#include "../../caro_ai_amd/csrc/caro_rules.h"
#include "../../caro_ai_amd/csrc/caro_variants.h"

using namespace caro;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

extern "C" {

const char* caro_last_error(void) { return g_err.c_str(); }
int caro_version(void) { return 100; }

#include "../../caro_ai_amd/csrc/caro_host.inc"

}  // extern "C"
