// csrc/kws_core.hip -- version / error plumbing of the C ABI.
#include "kws_common.h"

namespace kws {
std::string &last_error_slot()
{
    static thread_local std::string s;
    return s;
}
}  // namespace kws

extern "C" {

const char *kws_version(void) { return "kws-amd 0.1.0 (gfx950)"; }

const char *kws_last_error(void) { return kws::last_error_slot().c_str(); }

int kws_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
