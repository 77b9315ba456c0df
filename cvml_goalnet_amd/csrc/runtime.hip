// Error reporting and ABI version of libgoalnet_hip.so (int status codes, no exceptions).
#include <stdarg.h>

#include "common.h"

namespace goalnet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace goalnet

extern "C" {
int goalnet_abi_version(void) { return GOALNET_ABI_VERSION; }
const char* goalnet_last_error(void) { return goalnet::g_err; }
}
