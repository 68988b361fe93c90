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

// ---------------------------------------------------------------------------------------------------------
// Host helper for the training-time RoI list editing (denet/layer/denet_sparse.py:184-187): the reference
// trims an over-full RoI list with Python's `random.sample(list, n)`. To stay call-for-call on the stdlib
// generator without a 16k-iteration Python loop per step, this function advances a copy of CPython's MT19937
// state exactly like `random.sample(range(n), k)` does for the "pool" branch (n <= setsize):
//     for i in range(k): j = randbelow(n - i); result[i] = pool[j]; pool[j] = pool[n - i - 1]
//     randbelow(m): k = m.bit_length(); r = getrandbits(k); while r >= m: r = getrandbits(k)
//     getrandbits(k <= 32) = genrand_uint32() >> (32 - k)
// mt: the 624 state words, *pos: the index word of random.getstate()[1]. Pure host code.
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t mt_next(uint32_t* mt, int* pos) {
    const int N = 624, M = 397;
    if (*pos >= N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        for (; kk < N - 1; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        y = (mt[N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        *pos = 0;
    }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

extern "C" int denet_host_py_random_sample(uint32_t* mt, int* pos, int n, int k, int* pool_ws, int* out) {
    DENET_CHECK_ARG(mt && pos && pool_ws && out, "py_random_sample: null pointer");
    DENET_CHECK_ARG(n > 0 && k >= 0 && k <= n, "py_random_sample: need 0 <= k <= n");
    // setsize of random.sample (CPython Lib/random.py): 21, plus 4**ceil(log(3k, 4)) when k > 5
    long setsize = 21;
    if (k > 5) {
        long p = 1;
        while (p < 3L * k) p *= 4;
        setsize += p;
    }
    DENET_CHECK_ARG(n <= setsize, "py_random_sample: the set-based branch of random.sample is not provided (n=%d k=%d)",
                    n, k);
    for (int i = 0; i < n; ++i) pool_ws[i] = i;
    for (int i = 0; i < k; ++i) {
        const uint32_t m = (uint32_t)(n - i);
        int bits = 0;
        while ((m >> bits) != 0) bits++;
        uint32_t r = mt_next(mt, pos) >> (32 - bits);
        while (r >= m) r = mt_next(mt, pos) >> (32 - bits);
        out[i] = pool_ws[r];
        pool_ws[r] = pool_ws[n - i - 1];
    }
    return DENET_OK;
}
