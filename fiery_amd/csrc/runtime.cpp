// Error reporting and ABI version of libfiery_hip.so.
#include "common.h"

#include <cstring>

namespace fiery {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace fiery

extern "C" int fiery_abi_version(void) { return FIERY_ABI_VERSION; }
extern "C" const char* fiery_last_error(void) { return fiery::error_buffer(); }
