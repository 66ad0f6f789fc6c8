// thread-local last-error string for the C ABI (cfgpp_last_error).
#include <cstdarg>
#include <cstdio>
#include "common.h"
static thread_local char g_err[1024] = "";
void cfgpp_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* cfgpp_last_error(void) { return g_err; }

// One device per process (the torchrun layout: one rank per GPU).  The K-split workspace, the per-kernel
// "attribute set" flags and the tuning switches of the igemm / attention launchers are process-global, so a second
// engine on a DIFFERENT device would launch with the wrong workspace: refuse it instead.
static int g_claimed_device = -1;
int cfgpp_claim_device(int device_id) {
    if (g_claimed_device < 0) g_claimed_device = device_id;
    if (g_claimed_device != device_id) {
        cfgpp_set_error("this process already drives HIP device %d; libcfgpp_hip.so supports ONE device per process "
                        "(run one rank per GPU) - device %d refused", g_claimed_device, device_id);
        return -1;
    }
    return 0;
}
