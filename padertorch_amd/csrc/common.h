// Shared host/device helpers for libptmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ptmi.h"

namespace ptmi {

#define PTMI_RETURN_IF(cond, code) \
    do {                           \
        if (cond) return (code);   \
    } while (0)

// Launch check: hipGetLastError after the <<<>>> (positive hipError_t on failure).
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PTMI_OK : static_cast<int>(e);
}

// Zeroing of a few device words in front of a kernel that accumulates into them.  A kernel of this library, not
// hipMemsetAsync: the runtime's fill goes through its blit path, and on a queue whose memory pool another queue is
// working in it has been measured to start 60-260 us late (round 2, scripts/step_timeline.py: the gaps in front of every
// __amd_rocclr_fillBufferAligned of the step).
static __global__ void zero_words_kernel(uint32_t* p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

inline hipError_t zero_words_async(void* p, size_t n_words, hipStream_t st) {
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t*>(p), n_words);
    return hipGetLastError();
}

struct Geo {
    int size, shift, L, pad_left, pad_right, pad;
};

inline Geo to_geo(const ptmi_stft_geom* g) {
    return Geo{g->size, g->shift, g->window_length, g->pad_left, g->pad_right, g->pad};
}

// Frames the reference's conv1d produces for a row of n samples (padertorch/ops/_stft.py:137-158).
__host__ __device__ inline long long row_frames_of(const Geo& g, long long n) {
    const long long T = n + g.pad_left + g.pad_right;
    if (g.pad) {
        if (T < g.L) return 1;
        return (T - g.L + g.shift - 1) / g.shift + 1;
    }
    if (T < g.L) return 0;
    return (T - g.L) / g.shift + 1;
}

}  // namespace ptmi
