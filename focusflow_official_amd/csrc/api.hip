// libfocusflow_hip: error channel + ABI version.
#include "ff_common.h"

namespace ff {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace ff

extern "C" const char* ff_last_error(void) { return ff::err_buf(); }
extern "C" int ff_abi_version(void) { return 1; }
