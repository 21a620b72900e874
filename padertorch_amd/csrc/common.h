// Shared host/device helpers for libptmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
