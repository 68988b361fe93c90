// Error reporting for the C-ABI (include/denet_hip.h). The reference's GpuOps report failures through
// PyErr_Format + %(fail)s (denet_sparse_op.py:137-142); here every entry point returns an int status and
// leaves a thread-local message behind.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void denet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* denet_last_error(void) { return g_err; }

extern "C" int denet_abi_version(void) { return 1; }

// device properties the host side needs to size grids / report rooflines
extern "C" int denet_device_info(int device, int* cu_count, int* clock_khz, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        denet_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return -(int)e;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return DENET_OK;
}
