// Internal helpers shared by the HIP translation units of libavlmaps_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/avlmaps_hip.h"

namespace avl {

void set_error(const char* fmt, ...);

#define AVL_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            avl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AVL_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define AVL_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            avl::set_error(__VA_ARGS__);  \
            return AVL_ERR_INVALID;       \
        }                                 \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// number of CUs of the current device (cached)
int num_cus();

// Small per-thread, per-device scratch buffer owned by the library (grown on demand, never shrunk): the SYNCHRONOUS entry
// points that return host scalars (avl_argmax_f32, avl_topk_f32) keep their partial results here instead of allocating on
// every call.  Valid until the calling thread's next scratch() call; returns nullptr (and sets the error) on failure.
void* scratch(size_t bytes);
// keep what the device's default stream-ordered pool (hipMallocAsync) has grown to instead of handing it back at every synchronisation
// (HIP's default release threshold is 0: tens of ms per GB to map it again); once per device and thread, cheap afterwards
void keep_mempool_once();
void trim_mempool(uint64_t keep_bytes);   // hipMemPoolTrimTo on the device's default pool

constexpr int kWave = 64;

}  // namespace avl
