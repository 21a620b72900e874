// Optimizer step on the flat gradient bucket: global-norm clip + Adam + zero_grad in two passes over HBM.
//
// Replaces, on the step path of padertorch/train/trainer.py:512-532 (`clip_grad` -> `optimizer.step` -> `zero_grad`),
// torch.nn.utils.clip_grad_norm_ (padertorch/train/optimizer.py:35-42), torch.optim.Adam.step (:79-90) and
// Optimizer.zero_grad (:27-29) for the Trainer's flat fp32 gradient bucket (train/optimizer.py::FlatGrads):
//   pass 1  sum of squares of the bucket  -> per-workgroup partials (double) -> one fixed-order fold: the 2-norm
//           (reproducible: no atomics, the same order every run)
//   pass 2  g' = g * min(1, max_norm / (norm + 1e-6));  Adam moments and parameter update with torch's arithmetic
//           (lerp for exp_avg, bias corrections in double, denom = sqrt(v) / sqrt(bc2) + eps);  g = 0
// HBM-bound: 32 B per parameter in pass 2 (g, p, m, v read; p, m, v, g written), 4 B in pass 1.
//
// Parameters are separate tensors (the reference's state_dict layout); the flat bucket's segment table maps a flat
// index to (parameter pointer, offset).  The moments live in two flat buffers with the bucket's indexing.
#include "common.h"

namespace ptmi {

constexpr int kNormBlocks = 1024;        // partials of pass 1 (fixed: the fold order is part of the result)
constexpr int kMaxSegs = 1024;

struct AdamArgs {
    float* grad;             // flat bucket [n]
    float* m;                // exp_avg     [n]
    float* v;                // exp_avg_sq  [n]
    const long long* segs;   // device [nseg][3]: parameter pointer, first flat index, element count (ascending, dense)
    int nseg;
    long long n;
    const float* norm;       // device scalar from pass 1, or NULL (no clipping)
    float max_norm;
    const float* found_inf;  // device scalar: != 0 skips the update (gradients are still zeroed), or NULL
    const float* finite;     // device scalar: a NON-FINITE value skips the update as well (e.g. the sum of the step's losses), or NULL
    float* applied;          // device scalar out: 1 when the update was applied, 0 when it was skipped, or NULL
    const float* step;       // device scalar: optimizer steps taken so far (this one is step + 1)
    double lr, beta1, beta2;  // bias corrections and lr / bc1 are evaluated in double, as torch does on the host
    double eps, weight_decay;
    // device doubles [6] = lr, beta1, beta2, eps, weight_decay, max_norm, or NULL.  When set they REPLACE the by-value arguments: a captured
    // step (hipGraph) freezes kernel arguments, and the reference changes exactly these between iterations (hooks.py:736,1029:
    // BackOffValidationHook / LRAnnealingHook write param_group['lr']) - a replay reads the words its owner rewrote instead
    const double* hyper;
    int zero_grad;
};

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ x, long long n, double* __restrict__ partials) {
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // block-contiguous ranges: a partial covers one fixed slice of the bucket whatever the grid the hardware runs at a time
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = min(n4, lo + per);
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float4 q = x4[i];
        acc[0] = fmaf(q.x, q.x, acc[0]);
        acc[1] = fmaf(q.y, q.y, acc[1]);
        acc[2] = fmaf(q.z, q.z, acc[2]);
        acc[3] = fmaf(q.w, q.w, acc[3]);
    }
    double s = (double)acc[0] + (double)acc[1] + ((double)acc[2] + (double)acc[3]);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < (n & 3)) {          // tail past the last float4
        const float t = x[(n4 << 2) + threadIdx.x];
        s += (double)t * (double)t;
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void norm_fold_kernel(const double* __restrict__ partials, int nblk, float* __restrict__ norm) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm[0] = (float)sqrt(red[0]);
}

typedef __attribute__((address_space(1))) float gfloat;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4v gf32x4;

struct AdamConsts {
    float clip, step_size, bc2_sqrt;
    float beta2f, omb1, omb2, eps, weight_decay;      // float(beta2), float(1 - beta1), float(1 - beta2)
    bool skip;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, const AdamConsts& C) {
    float gg = g * C.clip;
    if (C.weight_decay != 0.f) gg = fmaf(C.weight_decay, p, gg);
    m = m + C.omb1 * (gg - m);                                          // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = C.beta2f * v + C.omb2 * gg * gg;
    const float denom = sqrtf(v) / C.bc2_sqrt + C.eps;
    p -= C.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(const AdamArgs A) {
    __shared__ long long seg_off[kMaxSegs + 1];
    __shared__ AdamConsts consts;
    for (int i = threadIdx.x; i < A.nseg; i += 256) seg_off[i] = A.segs[3 * i + 1];
    if (threadIdx.x == 0) {
        seg_off[A.nseg] = A.n;
        AdamConsts c;
        // a non-finite norm always skips: nothing useful can come from such gradients (torch's fused Adam leaves that
        // decision to the caller's found_inf; the Trainer used to compute it with four small kernels)
        c.skip = (A.found_inf != nullptr && A.found_inf[0] != 0.f) || (A.finite != nullptr && !isfinite(A.finite[0])) ||
                 (A.norm != nullptr && !isfinite(A.norm[0]));
        if (blockIdx.x == 0 && A.applied != nullptr) A.applied[0] = c.skip ? 0.f : 1.f;
        double lr = A.lr, beta1 = A.beta1, beta2 = A.beta2, eps = A.eps, wd = A.weight_decay;
        float max_norm = A.max_norm;
        if (A.hyper != nullptr) {
            lr = A.hyper[0]; beta1 = A.hyper[1]; beta2 = A.hyper[2]; eps = A.hyper[3]; wd = A.hyper[4];
            max_norm = (float)A.hyper[5];
        }
        const double t = (double)A.step[0] + 1.0;
        const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
        c.step_size = (float)(lr / bc1);
        c.bc2_sqrt = (float)sqrt(bc2);
        c.beta2f = (float)beta2;
        c.omb1 = (float)(1.0 - beta1);
        c.omb2 = (float)(1.0 - beta2);
        c.eps = (float)eps;
        c.weight_decay = (float)wd;
        c.clip = 1.f;
        if (A.norm != nullptr) c.clip = fminf(max_norm / (A.norm[0] + 1e-6f), 1.f);
        consts = c;
    }
    __syncthreads();
    const AdamConsts C = consts;
    const long long n4 = (A.n + 3) >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        const long long i0 = q << 2;
        // segment of i0: the last one whose first index is <= i0
        int lo = 0, hi = A.nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_off[mid] <= i0) lo = mid; else hi = mid - 1;
        }
        const bool full = i0 + 4 <= A.n;
        if (full && i0 + 4 <= seg_off[lo + 1]) {
            // the parameter's address comes out of the segment table as an integer: as a generic pointer its accesses are FLAT
            // instructions, behind which the compiler cannot count vector-memory returns (every wait becomes vmcnt(0) lgkmcnt(0)) -
            // parameters live in device memory: global address space
            gfloat* pp = reinterpret_cast<gfloat*>(static_cast<unsigned long long>(A.segs[3 * lo])) + (i0 - seg_off[lo]);
            float4 g = *reinterpret_cast<float4*>(A.grad + i0);
            if (!C.skip) {
                float4 m = *reinterpret_cast<float4*>(A.m + i0), v = *reinterpret_cast<float4*>(A.v + i0);
                const bool al = (reinterpret_cast<unsigned long long>(pp) & 15) == 0;
                float4 p;
                if (al) {
                    const f32x4v q = *reinterpret_cast<gf32x4*>(pp);
                    p = float4{q[0], q[1], q[2], q[3]};
                } else {
                    p = float4{pp[0], pp[1], pp[2], pp[3]};
                }
                adam_one(p.x, g.x, m.x, v.x, C);
                adam_one(p.y, g.y, m.y, v.y, C);
                adam_one(p.z, g.z, m.z, v.z, C);
                adam_one(p.w, g.w, m.w, v.w, C);
                if (al) *reinterpret_cast<gf32x4*>(pp) = f32x4v{p.x, p.y, p.z, p.w};
                else { pp[0] = p.x; pp[1] = p.y; pp[2] = p.z; pp[3] = p.w; }
                *reinterpret_cast<float4*>(A.m + i0) = m;
                *reinterpret_cast<float4*>(A.v + i0) = v;
            }
            if (A.zero_grad) *reinterpret_cast<float4*>(A.grad + i0) = float4{0.f, 0.f, 0.f, 0.f};
        } else {
            int s = lo;
            for (int e = 0; e < 4 && i0 + e < A.n; ++e) {
                const long long i = i0 + e;
                while (i >= seg_off[s + 1]) ++s;
                if (!C.skip) {
                    gfloat* pp = reinterpret_cast<gfloat*>(static_cast<unsigned long long>(A.segs[3 * s])) + (i - seg_off[s]);
                    float p = *pp, g = A.grad[i], m = A.m[i], v = A.v[i];
                    adam_one(p, g, m, v, C);
                    *pp = p;
                    A.m[i] = m;
                    A.v[i] = v;
                }
                if (A.zero_grad) A.grad[i] = 0.f;
            }
        }
    }
}

// The bias gradients of one BLSTM layer: db [ndir][n] (the backward recurrence's in-kernel sums) added into the 2 ndir gradient buffers
// bias_ih / bias_hh of every direction (torch.nn.LSTM keeps two bias vectors per direction; both get the same gradient) - ONE launch
// instead of 2 ndir `add_` launches on the weight-gradient queue.
struct BiasGradArgs {
    const float* db;
    float* out[4];          // [direction][ih | hh]
    int ndir, n;
};

__global__ void bias_grad_add_kernel(const BiasGradArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = blockIdx.y;
    if (i >= A.n) return;
    const float g = A.db[(size_t)d * A.n + i];
    A.out[2 * d][i] += g;
    A.out[2 * d + 1][i] += g;
}

}  // namespace ptmi

extern "C" int ptmi_lstm_bias_grad_add(const float* db, int32_t ndir, int32_t n, float* const* bias_ih_grad, float* const* bias_hh_grad,
                                       ptmi_stream_t stream) {
    PTMI_RETURN_IF(!db || !bias_ih_grad || !bias_hh_grad || (ndir != 1 && ndir != 2) || n < 1, PTMI_E_INVALID);
    ptmi::BiasGradArgs A{db, {nullptr, nullptr, nullptr, nullptr}, ndir, n};
    for (int d = 0; d < ndir; ++d) {
        PTMI_RETURN_IF(!bias_ih_grad[d] || !bias_hh_grad[d], PTMI_E_INVALID);
        A.out[2 * d] = bias_ih_grad[d];
        A.out[2 * d + 1] = bias_hh_grad[d];
    }
    hipLaunchKernelGGL(ptmi::bias_grad_add_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)ndir), dim3(256), 0,
                       static_cast<hipStream_t>(stream), A);
    return ptmi::launch_status();
}

namespace ptmi {

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int64_t ptmi_grad_norm_workspace_elems(void) { return kNormBlocks; }

int ptmi_grad_norm(const float* flat, int64_t n, double* workspace, float* norm_out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(flat == nullptr || workspace == nullptr || norm_out == nullptr || n < 0, PTMI_E_INVALID);
    PTMI_RETURN_IF((reinterpret_cast<uintptr_t>(flat) & 15) != 0, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(kNormBlocks), dim3(256), 0, st, flat, (long long)n, workspace);
    hipLaunchKernelGGL(norm_fold_kernel, dim3(1), dim3(256), 0, st, workspace, kNormBlocks, norm_out);
    return launch_status();
}

int ptmi_adam_flat(float* flat_grad, float* exp_avg, float* exp_avg_sq, const int64_t* segments, int32_t nseg, int64_t n,
                   const float* norm, float max_norm, const float* found_inf, const float* finite, float* applied, const float* step,
                   double lr, double beta1, double beta2, double eps, double weight_decay, const double* hyper, int32_t zero_grad,
                   ptmi_stream_t stream) {
    PTMI_RETURN_IF(flat_grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || segments == nullptr || step == nullptr,
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(nseg < 1 || n < 1, PTMI_E_INVALID);
    PTMI_RETURN_IF(nseg > kMaxSegs, PTMI_E_UNSUPPORTED);
    PTMI_RETURN_IF(((reinterpret_cast<uintptr_t>(flat_grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                     reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) != 0, PTMI_E_INVALID);
    PTMI_RETURN_IF((reinterpret_cast<uintptr_t>(hyper) & 7) != 0, PTMI_E_INVALID);
    AdamArgs A{flat_grad, exp_avg, exp_avg_sq, reinterpret_cast<const long long*>(segments), nseg, (long long)n, norm, max_norm,
               found_inf, finite, applied, step, lr, beta1, beta2, eps, weight_decay, hyper, zero_grad};
    const long long n4 = (n + 3) >> 2;
    const long long blocks = (n4 + 255) / 256;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);                // 32 workgroups per CU, grid-stride beyond
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    return launch_status();
}

}  // extern "C"
