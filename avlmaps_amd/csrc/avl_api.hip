// libavlmaps_hip.so -- library / device plumbing entry points (include/avlmaps_hip.h, first block).
#include <cstring>

#include "avl_common.h"

namespace avl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int num_cus() {
    static thread_local int cached_dev = -1;
    static thread_local int cached = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cached;
    if (dev != cached_dev) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cached = p.multiProcessorCount;
        cached_dev = dev;
    }
    return cached;
}

}  // namespace avl

using namespace avl;

extern "C" {

const char* avl_last_error(void) { return g_err; }
int avl_version(void) { return 100; }

int avl_device_count(int* h_count) {
    AVL_REQUIRE(h_count, "avl_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return AVL_ERR_NO_DEVICE;
    }
    *h_count = n;
    return AVL_OK;
}

int avl_set_device(int device) {
    AVL_HIP_CHECK(hipSetDevice(device));
    return AVL_OK;
}

int avl_device_name(int device, char* h_buf, size_t buf_len) {
    AVL_REQUIRE(h_buf && buf_len > 0, "avl_device_name: bad buffer");
    hipDeviceProp_t p;
    AVL_HIP_CHECK(hipGetDeviceProperties(&p, device));
    snprintf(h_buf, buf_len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return AVL_OK;
}

int avl_device_sync(void) {
    AVL_HIP_CHECK(hipDeviceSynchronize());
    return AVL_OK;
}

int avl_stream_create(void** h_stream_out) {
    AVL_REQUIRE(h_stream_out, "avl_stream_create: null output");
    hipStream_t s;
    AVL_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *h_stream_out = s;
    return AVL_OK;
}

int avl_stream_destroy(void* stream) {
    AVL_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    return AVL_OK;
}

int avl_stream_sync(void* stream) {
    AVL_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return AVL_OK;
}

int avl_malloc(void** h_ptr_out, size_t bytes) {
    AVL_REQUIRE(h_ptr_out, "avl_malloc: null output");
    AVL_HIP_CHECK(hipMalloc(h_ptr_out, bytes ? bytes : 1));
    return AVL_OK;
}

int avl_free(void* d_ptr) {
    if (d_ptr) AVL_HIP_CHECK(hipFree(d_ptr));
    return AVL_OK;
}

int avl_memset(void* d_ptr, int value, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemsetAsync(d_ptr, value, bytes, as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    AVL_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return AVL_OK;
}

int avl_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream) {
    AVL_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return AVL_OK;
}

int avl_event_create(void** h_event_out) {
    AVL_REQUIRE(h_event_out, "avl_event_create: null output");
    hipEvent_t e;
    AVL_HIP_CHECK(hipEventCreate(&e));
    *h_event_out = e;
    return AVL_OK;
}

int avl_event_destroy(void* event) {
    AVL_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return AVL_OK;
}

int avl_event_record(void* event, void* stream) {
    AVL_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return AVL_OK;
}

int avl_event_sync(void* event) {
    AVL_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return AVL_OK;
}

int avl_event_elapsed_ms(void* start, void* stop, float* h_ms) {
    AVL_REQUIRE(h_ms, "avl_event_elapsed_ms: null output");
    AVL_HIP_CHECK(hipEventElapsedTime(h_ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return AVL_OK;
}

}  // extern "C"
