// Shared pieces of the persistent (B)LSTM recurrence kernels (csrc/lstm.hip: exact fp32 MFMA; csrc/lstm_split.hip:
// split 16-bit MFMA products).
#pragma once
#include "common.h"

namespace ptmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, about 1 ulp each): they
// sit on the serial chain of every time step, where the library expf / tanhf / IEEE division cost about
// 0.4 us per step.  Absolute error < 2e-7 (checked against torch's CPU LSTM in tests/test_gpu_lstm.py).
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(x));       // exp(-2|x|)
    return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
}

struct LstmPersistArgs {
    float* gx;
    float* hy;
    float* c;
    const float* w;
    const int32_t* bs;       // device [T]
    const int64_t* offs;     // device [T]
    unsigned* flags;         // device [ndir][row tiles][kSlots] hand-off slots + 8 error words, zeroed per call
    int T, H, KP, ndir;
    unsigned expected;       // producer workgroups per chain (direction x row tile)
    unsigned max_polls;
    int hy_bytes;
    unsigned err_off;        // index of the error words in flags
    int dbg;                 // PTMI_LSTM_DBG timing ablations (16: no poll, 32: no drain, 64: no MFMA, 128: no operand loads)
    int tile0, ntiles;       // first row tile of this launch / row tiles of the whole batch
    const float* c0;         // [ndir, max_batch, H] initial cell state or null
    int max_batch;
    float* hyt;              // tile-major copy of hy for the hand-off: [T][16-row tile][dir][KP / 16][16 rows][16]
    int nt16;                // 16-row tiles of the whole batch
    const unsigned* w_amax;  // split kernels: float bits of max |W_hh| (device) or null
    int KP32;                // split kernels: H rounded up to 32
    unsigned* err_sink = nullptr;   // per-device count of timed-out launches (ptmi_lstm_set_error_sink) or null
    // split kernels, 1-D grid with the workgroups of a chain on `span` neighbouring XCDs (chain_tile below); span 0: the
    // 3-D grid (unit tile, direction, row tile) in dispatch order
    int nx = 0, nt = 0, span = 0;
    int uniform = 0;         // split kernels: all sequences have the same length (bs[t] = max_batch, offs[t] = t max_batch)
    // data-as-flag forward kernel: `fill_n16` 16-byte units from `fill_ptr` on (the hand-off planes of this layer's BACKWARD
    // scratch) get the fill pattern from an otherwise idle wavefront, a slice per workgroup and step
    uint4* fill_ptr = nullptr;
    unsigned long long fill_n16 = 0;
    // ... and the `zero_n16` 16-byte units BEHIND them (the backward scratch's bias sums, maximum word, arrival slots and error words)
    // get zeros the same way (round 6): the backward launch then has nothing to enqueue in front of its recurrence kernel
    unsigned long long zero_n16 = 0;
    // Row-slot batches (data-as-flag kernels; layout = uniform [T][max_batch] rows, max_batch <= 64 slots): several sequences lie END TO
    // END in one row slot, so "row b is alive at step t" and "row b has a predecessor" are no prefix rules any more.  masks[3 t + 0]
    // = rows alive at time index t, + 1 = rows whose sequence STARTS at t (no predecessor in the forward direction), + 2 = rows whose
    // sequence ENDS at t (no predecessor in the reverse direction); bit b = row b.  Null: the PackedSequence rules (bs / offs).
    const unsigned long long* masks = nullptr;
};

// Hand-off flags: every workgroup of a chain owns ONE slot and stores the number of steps it has
// finished (a plain write-through store: no read-modify-write, no two producers on one address); a
// consumer reads all slots of its chain with one or two loads per lane and goes on when every one of them
// has reached the step it needs.  (First form: 8 sharded arrival counters per step; the ~6 atomic adds
// queueing on each shard were part of every step's chain.)
constexpr int kSlots = 128;          // slots per chain = most producer workgroups a chain may have

// `dead`: set once a wait of this workgroup has run out (or another workgroup's has: the error word is sticky and
// re-read every 256 polls); every later wait then returns at once, so a launch whose workgroups are not all
// resident ends after ONE bounded spin per workgroup instead of one per time step.
__device__ __forceinline__ bool wait_arrivals(const unsigned* slots, unsigned producers, unsigned step,
                                              unsigned max_polls, unsigned* err, unsigned* sink = nullptr) {
    const unsigned lane = threadIdx.x & 63;
    for (unsigned it = 0; it < max_polls; ++it) {
        unsigned v = lane < producers ? __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
        if (producers > 64) {
            const unsigned w = lane + 64 < producers
                                   ? __hip_atomic_load(slots + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
            v = min(v, w);
        }
        if (__all(v >= step)) return true;
        if ((it & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sink) atomicAdd(sink, 1u);          // the failure path only: what the host polls instead of every call's own word
    }
    return false;
}

struct LstmPersistBwdArgs {
    const float* gates;
    const float* c;
    const float* dhy;
    const float* wt;
    float* dg;
    const int32_t* bs;
    const int64_t* offs;
    unsigned* flags;
    int T, H, ndir;
    unsigned expected;
    unsigned max_polls;
    int dg_bytes;
    unsigned err_off;
    int tile0, ntiles;   // first 16-row tile of this launch / tiles of the whole batch
    int dbg;             // PTMI_LSTM_DBG timing ablations (as in the forward kernel)
    const float* c0;
    int max_batch;
    int nx, nt, span;    // 1-D grid: unit tiles, row tiles of this launch, XCDs per chain (0: plain order)
    float* dgt;          // tile-major copy of dgates for the hand-off: [T][16-row tile][dir][4H / 16][16 rows][16]
    int nt16;            // 16-row tiles of the whole batch
    float* dbias;        // [ndir][4H] sum of dgates over all rows (zeroed by the host call, accumulated atomically)
    unsigned* dg_amax;   // split kernels: float bits of max |dgates| (zeroed by the host call, atomicMax)
    int G32;             // split kernels: 4H rounded up to 32
    unsigned* err_sink = nullptr;   // as in LstmPersistArgs
    // split kernels: this launch runs the processing steps [s_begin, s_end) of the T steps (a recurrence cut into several
    // launches lets the weight-gradient GEMMs of the finished time range start under the rest); the cell-state
    // gradient crosses the cut through dc_carry [ndir][max_batch][H] (written when s_end < T, read when s_begin > 0)
    int s_begin = 0, s_end = -1;
    float* dc_carry = nullptr;
    int uniform = 0;         // as in LstmPersistArgs
    // split kernels, equal lengths, batch a multiple of 16: dgates^T as bf16 (hi, lo) planes for the weight-gradient GEMMs
    // ([ndir][4H / 16][tp_kb][2][64] chunks of 16 B; null: not written); with them `dg` may be null
    uint4* dgtp = nullptr;
    long long tp_dir_stride = 0;      // chunks per direction
    int tp_kb = 0;                    // k blocks (32 packed rows) per column tile
    long long tp_row0[2] = {0, 0};    // first packed row of this launch's step range per direction: the planes' k index 0
    const unsigned long long* masks = nullptr;      // row-slot batches: as in LstmPersistArgs
    // gradient w.r.t. the FINAL cell state c_n [ndir][max_batch][H] (or null): enters the cell-state gradient of every sequence's
    // last step in the direction's forward sense = the first step this pass processes for it
    const float* dcn = nullptr;
};

// Workgroup L of a 1-D grid runs on XCD L % 8 (round-robin dispatch).  A chain = (direction, row tile)
// exchanges its operand rows among its own workgroups every step; they are written through one XCD's L2
// and read over the fabric by the others, which is what bounds the hand-off at batch >= 16.  With
// `span` = 8 / chains XCDs per chain, a chain's workgroups sit on `span` neighbouring XCDs instead of all
// eight, so 1/span of what a workgroup reads is local.  Returns false for the padding workgroups.
__device__ __forceinline__ bool chain_tile(int nx, int nt, int span, int* x, int* y, int* dir, int nchains = 1 << 30) {
    const int L = blockIdx.x;
    int chain;
    if (span > 0) {
        const int xcd = L & 7;
        chain = xcd / span;
        *x = (L >> 3) * span + (xcd - chain * span);
        if (*x >= nx || chain >= nchains) return false;
    } else {
        *x = L % nx;
        chain = L / nx;
    }
    *y = chain % nt;
    *dir = chain / nt;
    return true;
}

// MTL = 16-row tiles per workgroup (2: a batch of 64 stays ONE launch of 32-row chains; the weights in
// registers serve both tiles, the second tile's operands are requested while the first one multiplies).

// csrc/lstm_split.hip
// data-as-flag kernels (csrc/lstm_split.hip): the caller pre-fills the hand-off planes with daf_prefill
bool fwd_daf_applies(int jt, bool small, bool one_per_cu);
bool bwd_daf_applies();
int daf_prefill(void* p, size_t words, hipStream_t st);
int daf_prefill_and_zero(void* p, size_t words, void* z, size_t zero_words, hipStream_t st);      // + `zero_words` zeroed words at z, one launch
int launch_fwd_split(const LstmPersistArgs& A, int jt, bool small, bool one_per_cu, dim3 grid, hipStream_t st, bool daf);
int launch_bwd_split(const LstmPersistBwdArgs& A, int mtl, unsigned nwg, hipStream_t st);

}  // namespace ptmi
