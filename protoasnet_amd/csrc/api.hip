// Library-wide plumbing: version, thread-local error text, launch-status check.
#include "common.h"

namespace pasn {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_error = std::string(what) + ": " + hipGetErrorString(e);
        return PASN_ERR_LAUNCH;
    }
    return PASN_OK;
}

}  // namespace pasn

extern "C" int pasn_version(void) { return PASN_VERSION; }
extern "C" const char* pasn_last_error(void) { return pasn::g_last_error.c_str(); }
