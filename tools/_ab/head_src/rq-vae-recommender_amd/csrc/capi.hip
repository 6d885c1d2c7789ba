// capi.hip -- library-level entry points of librqhip.so (version, error string, device info).
#include <stdarg.h>

#include "rqhip_common.h"

namespace rqhip {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    return dev;
}

int cu_count() {
    static int cached[kMaxDevices] = {};   // per device: a process may drive several GPUs
    const int dev = current_device();
    if (dev < kMaxDevices && cached[dev] > 0) return cached[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    if (dev < kMaxDevices) cached[dev] = n;
    return n;
}


__global__ void fill_words_kernel(uint32_t *dst, uint32_t word, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = word;
}

int fill_words(void *dst, uint32_t word, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 4;
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t *>(dst), word, n);
    RQ_CHECK_LAUNCH("fill_words_kernel");
    return 0;
}

static hipEvent_t *g_ev = nullptr;   // 2 * g_cap events
struct ProfNote { int tag; double flops, bytes; };
static ProfNote *g_note = nullptr;   // g_cap notes
static int g_cap = 0, g_n = 0;
static unsigned g_mask = ~0u;        // bit t set: launches tagged t are recorded
static bool g_open = false;

void profile_begin(hipStream_t s, int tag, double flops, double bytes) {
    g_open = false;
    if (g_cap == 0 || g_n >= g_cap || !((g_mask >> (tag & 31)) & 1u)) return;
    g_note[g_n] = ProfNote{tag, flops, bytes};
    if (hipEventRecord(g_ev[2 * g_n], s) == hipSuccess) g_open = true;
}

void profile_end(hipStream_t s) {
    if (!g_open) return;
    g_open = false;
    if (hipEventRecord(g_ev[2 * g_n + 1], s) == hipSuccess) ++g_n;
}

}  // namespace rqhip

extern "C" int rqhip_version(void) { return RQHIP_VERSION; }

extern "C" const char *rqhip_last_error(void) { return rqhip::g_err; }

extern "C" int rqhip_device_cu_count(int *out) {
    if (!out) {
        rqhip::set_error("rqhip_device_cu_count: null output pointer");
        return RQHIP_EARG;
    }
    int dev = 0, n = 0;
    RQ_RETURN_IF_HIP(hipGetDevice(&dev));
    RQ_RETURN_IF_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    *out = n;
    return RQHIP_OK;
}

extern "C" int rqhip_profile_enable(int max_records) {
    using namespace rqhip;
    for (int i = 0; i < 2 * g_cap; ++i) (void)hipEventDestroy(g_ev[i]);
    delete[] g_ev;
    delete[] g_note;
    g_ev = nullptr;
    g_note = nullptr;
    g_cap = g_n = 0;
    if (max_records <= 0) return RQHIP_OK;
    g_ev = new hipEvent_t[2 * (size_t)max_records];
    g_note = new ProfNote[(size_t)max_records];
    for (int i = 0; i < 2 * max_records; ++i) RQ_RETURN_IF_HIP(hipEventCreate(&g_ev[i]));
    g_cap = max_records;
    return RQHIP_OK;
}

extern "C" int rqhip_profile_select(unsigned tag_mask) {
    rqhip::g_mask = tag_mask;
    return RQHIP_OK;
}

extern "C" int rqhip_profile_read_tagged(rqhip_profile_record *out, int cap, int *n_out) {
    using namespace rqhip;
    if (!n_out || (cap > 0 && !out)) {
        set_error("rqhip_profile_read_tagged: null output pointer");
        return RQHIP_EARG;
    }
    int n = 0;
    for (int i = 0; i < g_n && n < cap; ++i) {
        RQ_RETURN_IF_HIP(hipEventSynchronize(g_ev[2 * i + 1]));
        float ms = 0.f;
        RQ_RETURN_IF_HIP(hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]));
        out[n].tag = g_note[i].tag;
        out[n].ms = ms;
        out[n].flops = g_note[i].flops;
        out[n].bytes = g_note[i].bytes;
        ++n;
    }
    *n_out = n;
    g_n = 0;
    return RQHIP_OK;
}

extern "C" int rqhip_profile_read(float *ms_out, int cap, int *n_out) {
    using namespace rqhip;
    if (!n_out || (cap > 0 && !ms_out)) {
        set_error("rqhip_profile_read: null output pointer");
        return RQHIP_EARG;
    }
    int n = 0;
    for (int i = 0; i < g_n && n < cap; ++i) {
        if (g_note[i].tag != RQHIP_PROF_RQ_FORWARD) continue;
        RQ_RETURN_IF_HIP(hipEventSynchronize(g_ev[2 * i + 1]));
        float ms = 0.f;
        RQ_RETURN_IF_HIP(hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]));
        ms_out[n++] = ms;
    }
    *n_out = n;
    g_n = 0;
    return RQHIP_OK;
}
