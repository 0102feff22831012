// Library-wide state of libncf_hip.so: version, build arch, thread-local error string, the option table and the
// per-device CU-count cache (everything include/ncf_abi.h lists under "state").
#include "ncf_common.h"
#include <atomic>
#include <string.h>

namespace ncf {
thread_local char g_err[512] = "";

static std::atomic<int> g_options[NCF_OPT_COUNT_];  // zero-initialised: every option starts at 0 = "choose by shape"

int option(int opt) { return g_options[opt].load(std::memory_order_relaxed); }

int num_cus() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cache[kMaxDev];  // 0 = not asked yet; a race stores the same value twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
}  // namespace ncf

using namespace ncf;

static const struct {
    const char* name;
    int id, lo, hi;
} kOptions[] = {
    {"bf16_kernel", NCF_OPT_BF16_KERNEL, 0, 3},
    {"linear_kernel", NCF_OPT_LINEAR_KERNEL, 0, 2},
    {"linear_kslices", NCF_OPT_LINEAR_KSLICES, 0, 8},
    {"attn_grouped_kernel", NCF_OPT_ATTN_GROUPED_KERNEL, 0, 2},
    {"gather_kernel", NCF_OPT_GATHER_KERNEL, 0, 2},
};

extern "C" int ncf_version(void) { return NCF_ABI_VERSION; }
extern "C" const char* ncf_last_error(void) { return ncf::g_err; }
extern "C" const char* ncf_build_arch(void) { return "gfx950"; }
#ifndef NCF_BUILD_ID
#define NCF_BUILD_ID "unstamped"
#endif
// hash of every source, header and compiler flag this library was built from (csrc/build.py source_id()); the marker in front lets
// the build script read it out of the file without loading it
static const char kBuildId[] = "NCF_BUILD_ID=" NCF_BUILD_ID;
extern "C" const char* ncf_build_id(void) { return kBuildId + 13; }

extern "C" int ncf_set_option(const char* name, int value) {
    if (!name) return fail(NCF_EINVAL, "ncf_set_option: null name");
    for (const auto& o : kOptions)
        if (!strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi || (o.id == NCF_OPT_LINEAR_KSLICES && value != 0 && value != 4 && value != 8))
                return fail(NCF_EINVAL, "ncf_set_option: %s = %d is not a value of this option", name, value);
            g_options[o.id].store(value, std::memory_order_relaxed);
            return NCF_OK;
        }
    return fail(NCF_EINVAL, "ncf_set_option: unknown option '%s'", name);
}

extern "C" int ncf_get_option(const char* name, int* value) {
    if (!name || !value) return fail(NCF_EINVAL, "ncf_get_option: null argument");
    for (const auto& o : kOptions)
        if (!strcmp(name, o.name)) {
            *value = option(o.id);
            return NCF_OK;
        }
    return fail(NCF_EINVAL, "ncf_get_option: unknown option '%s'", name);
}
