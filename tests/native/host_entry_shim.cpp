// The host-only C-ABI entry points (catfish_amd/csrc/chunks_host.hpp, loader_host.hpp, split_host.hpp: no device work) compiled on their own,
// for the sanitizer build of tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -fPIC ...
// catfish_hip.hip provides fail() / cf_last_error() to them in the product library; this file stands in for just that.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/catfish_hip.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
extern "C" const char* cf_last_error(void) { return g_err.c_str(); }

#include "../../catfish_amd/csrc/chunks_host.hpp"
#include "../../catfish_amd/csrc/loader_host.hpp"
#include "../../catfish_amd/csrc/split_host.hpp"
