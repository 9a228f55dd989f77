// Error string, version and device check of libpnr.so.
#include <stdarg.h>
#include <string.h>

#include "pnr_common.h"

static thread_local char g_err[512] = "";

void pnr_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pnr_cu_count(void)
{
    static int cached[64];                       // per device ordinal; 0 = not asked yet (benign race: same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

PNR_EXPORT int pnr_version(void) { return 1; }

PNR_EXPORT const char* pnr_last_error(void) { return g_err; }

PNR_EXPORT int pnr_device_check(int dev, char* buf_host, int buflen)
{
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        pnr_set_error("hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
        return PNR_ENODEV;
    }
    if (buf_host && buflen > 0) {
        snprintf(buf_host, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
                 prop.multiProcessorCount);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pnr_set_error("device %d is %s, libpnr.so is built for gfx950 only", dev, prop.gcnArchName);
        return PNR_ENODEV;
    }
    return PNR_OK;
}
