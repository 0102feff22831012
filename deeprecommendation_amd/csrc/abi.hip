// Library-wide state of libncf_hip.so: version, build arch, thread-local error string.
#include "ncf_common.h"

namespace ncf {
thread_local char g_err[512] = "";
}

extern "C" int ncf_version(void) { return NCF_ABI_VERSION; }
extern "C" const char* ncf_last_error(void) { return ncf::g_err; }
extern "C" const char* ncf_build_arch(void) { return "gfx950"; }
