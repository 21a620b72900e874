// Persistent (B)LSTM recurrence kernels with split 16-bit MFMA products (gfx950).
//
// Same launches, hand-off protocol (write-through tile-major copy, one slot per producer workgroup, bounded polls),
// layouts and results as lstm_{fwd,bwd}_persistent_kernel in lstm.hip (the recurrence of torch.nn.LSTM on a
// PackedSequence: padertorch/contrib/examples/source_separation/pit/model.py:60-66,97, contrib/tcl/dc.py:32-34,61).
// What changes is how the per-step matrix product is evaluated.  fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the
// 16-bit rate; the product h_{t-1} W_hh^T (dgates_{t+1} W_hh in the backward pass) is the longest single item of
// every step's serial chain (60 / 76 MFMAs of 32 cycles per wavefront).  Here
//   forward   h in (-1, 1) is handed on as fp16 (hi, lo) halves of 2^10 h, W_hh lives in registers as fp16 halves of
//             2^(13-e) W (e = exponent of max |W_hh|, device word from ptmi_absmax), and the product is
//             hi*hi + hi*lo + lo*hi in one fp32 accumulator (v_mfma_f32_16x16x32_f16: 9 x 3 MFMAs of ~17 cycles per
//             wavefront): as close to the fp64 product as the exact-fp32 chain (scripts/mb/split_mfma_accuracy.hip);
//   backward  dgates have no bound known before they are computed, so their halves are bf16 (fp32's range, no scale;
//             error <= 7e-7 of sum |a b|, v_mfma_f32_16x16x32_bf16), W_hh^T likewise.
// The tile-major hand-off copy keeps its size: a "tile" is now 16 rows x 32 k of 16-bit values (1 KB, one
// buffer_load_dwordx4 per lane) in MFMA-fragment order (handoff_index below), two planes (hi, lo) per 32-wide k block; a
// producer lane trades one half with its neighbour lane so that it still issues ONE 4-byte write-through store per value.
#include <stdlib.h>

#include <algorithm>

#include "lstm_common.h"

namespace ptmi {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));

// Hand-off planes: per (time, 16-row tile, direction) and 32-wide k block two 1 KB plane tiles (hi, lo) in the MFMA-fragment
// order of csrc/gemm_planes.hip - the 16-byte chunk of (k group g = 0..3, row r) at slot 16 g + r holds the 8 values
// [r][32 kb + 8 g ..] - so that lane l of a consumer wavefront reads its operand registers at l * 16 AND the dense GEMMs that
// follow (next layer's projection, the input gradient) take the copy as their operand planes as it lies.
// Index (in 16-bit values) of column `col` of row `row` in plane `plane` within one (time, tile, direction) block:
__device__ __forceinline__ int handoff_index(int col, int plane, int row) {
    return (((col >> 5) * 2 + plane) * 64 + ((col & 31) >> 3) * 16 + row) * 8 + (col & 7);
}

// Data-as-flag hand-off (DAF kernels below): the planes are pre-filled with 0xFFFF in every 16-bit value - a NaN pattern no
// conversion produces (they give the canonical 0x7E00 / 0x7FC0 forms) -, producers only issue their write-through stores,
// consumers request their operand tiles and check every value they are going to use.
constexpr unsigned kFill = 0xffffffffu;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// running maximum over the unsigned 16-bit halves of a fragment (3 + 1 packed VALU operations)
__device__ __forceinline__ unsigned fold_max16(const uint4 v, unsigned m) {
    us2 x = __builtin_elementwise_max(__builtin_bit_cast(us2, v.x), __builtin_bit_cast(us2, v.y));
    x = __builtin_elementwise_max(x, __builtin_bit_cast(us2, v.z));
    x = __builtin_elementwise_max(x, __builtin_bit_cast(us2, v.w));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, __builtin_bit_cast(us2, m)));
}
__device__ __forceinline__ bool has_fill(unsigned m) { return (m & 0xffffu) == 0xffffu || (m >> 16) == 0xffffu; }

// A step's FIRST request is held back until `hold` ticks of the 100 MHz clock after the workgroup's previous barrier (the one
// point of a step all its wavefronts share; the owners still have their activations and stores to do behind it, the others
// do not): a request that arrives before the slowest producer's stores are visible is wasted, and every wasted 39 / 154 KB
// round of every workgroup loads the fabric the stores travel on - without the hold the protocol was no faster than the
// flags (3.27 vs 3.21 us per step).  `hold` adapts slowly and in step: a first request that still found the pattern adds 3
// ticks at once; every 64th step every wavefront that has not failed since the last one takes 1 tick off (all at the same
// step: with free-running per-wavefront probing some wavefront of the chain is always just trying a shorter wait, and the
// whole chain pays its failed attempt).
// A word of a read-only, launch-constant table (row-slot masks; the PackedSequence batch sizes / row offsets) through the SCALAR cache: the address is wave-uniform, but the
// compiler cannot prove that none of the kernel's stores aliases the table and issues a vector load - whose use then waits with
// s_waitcnt vmcnt(0) for every outstanding vector-memory operation, i.e. for the previous step's write-through hand-off stores (a store
// round trip at the head of every step of the chain).  The tables' entries used to come through `uniform ? A.max_batch : A.bs[t]`, which the
// compiler turns into ONE load from a selected address - kernel-argument segment or table - i.e. a FLAT load, and with a FLAT load
// outstanding it can no longer count vector-memory returns: every wait of the operand passes becomes vmcnt(0).
template <class T>
__device__ __forceinline__ T ld_const(const T* p, int idx) {
    typedef const __attribute__((address_space(4))) T* cptr;
    return reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(p))[idx];
}
__device__ __forceinline__ unsigned long long ld_const64(const unsigned long long* p, int idx) { return ld_const(p, idx); }

struct DafHold {
    unsigned hold, failed;
    unsigned long long ref;
    __device__ __forceinline__ void mark() { ref = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void wait() const {
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() - ref) < hold) __builtin_amdgcn_s_sleep(1);
    }
    __device__ __forceinline__ void update(bool first_clean, int step, bool adapt) {
        if (!adapt) return;
        if (!first_clean) {
            hold = min(hold + 3u, 1000u);
            failed = 1u;
        }
        if ((step & 63) == 63) {
            if (!failed && hold > 0u) --hold;
            failed = 0u;
        }
    }
};

__global__ void fill_words_kernel(uint4* p, size_t n16, unsigned word) {
    const uint4 v = make_uint4(word, word, word, word);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// host: fill `words` 32-bit words (a multiple of 4, 16-byte aligned) with the pattern
int daf_prefill(void* p, size_t words, hipStream_t st) {
    const size_t n16 = words / 4;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 8192);
    hipLaunchKernelGGL(fill_words_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint4*>(p), n16, kFill);
    return launch_status();
}

// The set-up of a persistent launch in ONE kernel: the planes start as the pattern, the words behind them (bias sums, arrival
// slots, error words) as zero.  (Two launches before: ~5 us of queue time each in front of every recurrence of the step.)
__global__ void fill_and_zero_kernel(uint4* p, size_t n16, unsigned word, unsigned* z, size_t nz) {
    const uint4 v = make_uint4(word, word, word, word);
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n16; i += step) p[i] = v;
    for (size_t i = i0; i < nz; i += step) z[i] = 0u;
}

int daf_prefill_and_zero(void* p, size_t words, void* z, size_t zero_words, hipStream_t st) {
    const size_t n16 = words / 4;
    const unsigned grid = (unsigned)std::min<size_t>((std::max(n16, zero_words) + 255) / 256, 8192);
    hipLaunchKernelGGL(fill_and_zero_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint4*>(p), n16, kFill, static_cast<unsigned*>(z), zero_words);
    return launch_status();
}

constexpr float kHScale = 1024.f;       // forward: h is handed on as halves of 2^10 h (lo stays normal down to |h| = 2^-13)

__device__ __forceinline__ float pow2_scale(const unsigned* amax_bits) {
    if (!amax_bits) return 1.f;
    const unsigned e = (*amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 0xffu) return 1.f;
    return __uint_as_float((unsigned)(127 + 13 + 127 - (int)e) << 23);
}

// 16-bit (hi, lo) halves of one value, as bit patterns
template <bool BF16>
__device__ __forceinline__ void split1(float v, unsigned* hi, unsigned* lo) {
    if (BF16) {
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        *hi = __builtin_bit_cast(unsigned short, h);
        *lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        *hi = __builtin_bit_cast(unsigned short, h);
        *lo = __builtin_bit_cast(unsigned short, l);
    }
}

// 8 consecutive values -> the (hi, lo) MFMA operand registers of one lane
template <bool BF16>
__device__ __forceinline__ void split8(const float (&v)[8], uint4* hi, uint4* lo) {
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1<BF16>(v[e], &h[e], &l[e]);
    *hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    *lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

template <bool BF16>
__device__ __forceinline__ f32x4 mma16(const uint4 a, const uint4 b, const f32x4 c) {
    if (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// One value of a producer lane -> its 4-byte store into the (hi, lo) planes.  Lanes 2i and 2i+1 own neighbouring
// columns c, c+1 (c even): the even lane stores {hi_c, hi_c+1} into the hi plane, the odd lane {lo_c, lo_c+1} into the
// lo plane.  Must be executed by both lanes of a pair (the exchange is a wave-level shuffle).  Returns the word and
// sets *plane.
template <bool BF16>
__device__ __forceinline__ unsigned pair_word(float v, int lane, int* plane) {
    unsigned hi, lo;
    split1<BF16>(v, &hi, &lo);
    const bool odd = lane & 1;
    const unsigned send = odd ? hi : lo;                               // what the neighbour stores
    // lane ^ 1 through the data-parallel primitive (quad_perm [1, 0, 3, 2]: one VALU operation; __shfl_xor goes through the LDS crossbar,
    // ds_bpermute_b32 + s_waitcnt lgkmcnt(0), in front of the hand-off store of every step)
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);
    *plane = odd ? 1 : 0;
    return odd ? (recv | (lo << 16)) : (hi | (recv << 16));
}

// ---------------------------------------------------------------------------------------------------------------
// Forward, data-as-flag hand-off (the default wherever fwd_daf_applies; one workgroup per CU).
// The protocol above costs every step two store round trips in series (the write-through drain, then the flag) and two
// load round trips (the poll, then the operands).  Here the scratch planes are pre-filled with a pattern no value can
// have (0xFFFF: a NaN in fp16), producers only issue their write-through stores, and a consumer wavefront requests its
// operand tiles at once and again until none of their 16-bit values is the pattern: one store and one load round trip.
// Every 4-byte word is written exactly once per launch and checked by the lane that uses it, so no ordering between
// stores is assumed.  The partial sums are double-buffered in LDS (one barrier per step is left).
// Timing ablations of the step's terms (scripts/exp_lstm.py with a library built with -DPTMI_LSTM_ABLATE: profiles/r6_lstm_ablations.txt;
// results void).  PTMI_LSTM_DBG bits: 64 no MFMAs, 128 operand values not waited for, 1024 no look-ahead loads of the next step's input
// pre-activations (forward), 2048 no row-major / plane stores, 16384 no hand-off stores, 32768 no clock read behind the barrier, 65536 no
// LDS reduction / barrier, 131072 no transcendentals (forward); 4096 / 8192 exist in every build.  The product
// build compiles none of it (ABL is false: the branches fold away).
#ifdef PTMI_LSTM_ABLATE
constexpr bool ABL = true;
#else
constexpr bool ABL = false;
#endif

template <int JT, int NW, int CB, int MTL, int RED_PAD = 4>
__global__ __launch_bounds__(NW * 64, 2) void lstm_fwd_daf_kernel(const LstmPersistArgs A) {
    constexpr int NC = 4 * JT;
    constexpr int NT = NC / 16;
    constexpr int MR = 16 * MTL;
    constexpr int ACTW = (MR * JT + 63) / 64;
    int bx = blockIdx.x, bz = blockIdx.z, dir = blockIdx.y;
    if (A.span > 0 && !chain_tile(A.nx, A.nt, A.span, &bx, &bz, &dir, A.ndir * A.nt)) return;
    // dense index of this workgroup among those that run (for its share of the backward scratch's fill)
    const unsigned wg_lin = A.span > 0 ? (unsigned)((dir * A.nt + bz) * A.nx + bx)
                                       : (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned wg_cnt = A.span > 0 ? (unsigned)(A.ndir * A.nt * A.nx) : gridDim.x * gridDim.y * gridDim.z;
    const int j0 = bx * JT;
    const int m0 = (A.tile0 + bz) * MR;
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, r = lane & 15;
    // row pitch NC + 4 (a multiple of 16, plus 4): the accumulator rows a wavefront writes at once - 4 g4 + q, 16 lanes each -
    // fall into 64 different banks (pitch NC + 1: SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles)
    __shared__ float red[2][NW][MR][NC + RED_PAD];
    const bool uniform = A.uniform != 0;
    auto bs_at = [&](int t) { return uniform ? A.max_batch : ld_const(A.bs, t); };
    auto offs_at = [&](int t) { return uniform ? (long long)t * A.max_batch : (long long)ld_const(A.offs, t); };
    // row-slot batches: per-step row masks instead of the prefix rules "b < bs[t]" / "b < bs[t -+ 1]" (LstmPersistArgs::masks)
    const bool masked = A.masks != nullptr;
    typedef unsigned long long u64;
    auto alive_at = [&](int t) -> u64 { return masked ? ld_const64(A.masks, 3 * t) : ~0ull; };
    auto bit = [](u64 m, int i) { return ((m >> (i & 63)) & 1ull) != 0ull; };
    const int nblk = A.KP32 >> 5;
    const int base = nblk / NW, extra = nblk - base * NW;
    const int kb0 = __builtin_amdgcn_readfirstlane(wave * base + min(wave, extra));
    const int nbw = __builtin_amdgcn_readfirstlane(base + (wave < extra ? 1 : 0));
    const int kfirst = __builtin_amdgcn_readfirstlane(min(kb0, nblk - 1));
    const int ilast = __builtin_amdgcn_readfirstlane(max(nbw - 1, 0));
    const float ws = pow2_scale(A.w_amax);
    const float inv = 1.f / (ws * kHScale);
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 bh[CB][NT], bl[CB][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cidx = nt * 16 + r;
        const int gate = cidx / JT, uu = cidx - gate * JT;
        const bool bv = j0 + uu < H;
        const float* bp = A.w + ((long long)dir * G + gate * H + (bv ? j0 + uu : 0)) * A.KP;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int k = (kb0 + i) * 32 + g4 * 8;
            const bool in = bv && i < nbw;
            const f32x4 w0 = (in && k + 4 <= A.KP) ? *reinterpret_cast<const f32x4*>(bp + (k + 4 <= A.KP ? k : 0)) : zero;
            const f32x4 w1 = (in && k + 8 <= A.KP) ? *reinterpret_cast<const f32x4*>(bp + (k + 8 <= A.KP ? k + 4 : 0)) : zero;
            const float v[8] = {w0[0] * ws, w0[1] * ws, w0[2] * ws, w0[3] * ws, w1[0] * ws, w1[1] * ws, w1[2] * ws, w1[3] * ws};
            split8<false>(v, &bh[i][nt], &bl[i][nt]);
            // the halves are what the time loop keeps in registers: without this the compiler holds the fp32 weights instead (the same 8
            // registers per fragment) and REPEATS scale + split in every time step - 146 v_fma_mixlo_f16 + 72 v_or_b32_sdwa + 78 shifts
            // of the loop's 785 vector instructions (scripts/loop_instr_count.py; round 6)
            asm volatile("" : "+v"(bh[i][nt].x), "+v"(bh[i][nt].y), "+v"(bh[i][nt].z), "+v"(bh[i][nt].w), "+v"(bl[i][nt].x), "+v"(bl[i][nt].y),
                              "+v"(bl[i][nt].z), "+v"(bl[i][nt].w));
        }
    }
    const size_t tile_elems = (size_t)A.KP32 * 16;
    const int tile16 = (A.tile0 + bz) * MTL;
    unsigned* const err = A.flags + A.err_off;
    const int bl_ = tid / JT, u = tid - bl_ * JT;
    const int b = m0 + bl_;
    bool alive = true;
    float pre_n[4] = {0.f, 0.f, 0.f, 0.f};
    float c_reg = 0.f;
    // hand-off store of this thread's element, in 32-bit words: lanes 2i / 2i+1 store the (hi, lo) pair words of columns c, c + 1
    const long long hs_step = (long long)A.nt16 * A.ndir * (long long)tile_elems;         // (tile_elems: fp32 = 32-bit words per block)
    const long long hs_thr = ((long long)(tile16 + (bl_ >> 4)) * A.ndir + dir) * (long long)tile_elems +
                             handoff_index((j0 + u) & ~1, lane & 1, bl_ & 15) / 2;
    // initial cell state of this thread's element (the same row and unit at every step; zero without hx)
    const float c0_own = (A.c0 && tid < MR * JT && b < A.max_batch && j0 + u < H) ? A.c0[((long long)dir * A.max_batch + b) * H + j0 + u] : 0.f;
    {
        const int t0 = dir == 0 ? 0 : A.T - 1;
        if (tid < MR * JT && b < bs_at(t0) && bit(alive_at(t0), b) && j0 + u < H) {
            const float* np = A.gx + (offs_at(t0) + b) * ld_g + (long long)dir * G + j0 + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) pre_n[q] = np[q * H];
        }
    }
    // this workgroup's slice of the pattern fill, dealt out over the steps (the last wavefront owns no element here)
    const unsigned long long fill_all = A.fill_n16 + A.zero_n16;          // pattern units, then zero units
    const unsigned long long fill_share = A.fill_ptr ? (fill_all + wg_cnt - 1) / wg_cnt : 0ull;
    unsigned long long fill_pos = fill_share * wg_lin;
    const unsigned long long fill_end = A.fill_ptr ? (fill_pos + fill_share < fill_all ? fill_pos + fill_share : fill_all) : 0ull;
    const unsigned long long fill_step = (fill_share + (unsigned)A.T - 1) / (unsigned)A.T;
    auto fill_some = [&](bool all) {
        if (ACTW < NW && wave >= ACTW && fill_pos < fill_end) {        // the wavefronts without elements share the step's portion
            const unsigned long long stop = all ? fill_end : (fill_pos + fill_step < fill_end ? fill_pos + fill_step : fill_end);
            const uint4 v = make_uint4(kFill, kFill, kFill, kFill), z = make_uint4(0u, 0u, 0u, 0u);
            for (unsigned long long i = fill_pos + (unsigned)(wave - ACTW) * 64u + lane; i < stop; i += 64u * (NW - ACTW))
                A.fill_ptr[i] = i < A.fill_n16 ? v : z;
            fill_pos = stop;
        }
    };
    // first-request hold of a launch, in ticks of the 100 MHz clock (PTMI_LSTM_DBG bits 16-23: / 4); one 16-row tile per workgroup: 92 (in
    // the training step, one box: adaptive from 100 / 92 / 88: 6.82 / 6.77 / 6.75 ms, fixed 92: 6.73; two tiles (B = 64): 100 - 22.05 against
    // 22.15 from 92, 22.6 fixed 92)
    DafHold dd{(A.dbg >> 16) & 0xff ? (unsigned)((A.dbg >> 16) & 0xff) * 4u : (MTL == 1 ? 92u : 100u), 0u, 0ull};
    const bool adapt = !(A.dbg & (1 << 24));
    dd.mark();
    for (int s = 0; s < A.T; ++s) {
        const int t = dir == 0 ? s : A.T - 1 - s;
        const int nb = bs_at(t);
        const long long row0 = offs_at(t);
        const int tp = dir == 0 ? t - 1 : t + 1;
        const int nprev = (tp >= 0 && tp < A.T) ? min(bs_at(tp), nb) : 0;
        // rows of this step that continue a sequence (their hidden / cell state of the previous step counts)
        const u64 amask = alive_at(t);
        const u64 pmask = masked ? ((tp >= 0 && tp < A.T) ? amask & ~ld_const64(A.masks, 3 * t + (dir == 0 ? 1 : 2)) : 0ull) : ~0ull;
        auto has_pred = [&](int row) { return row < nprev && bit(pmask, row); };
        const bool has_rec = masked ? ((pmask >> m0) & ((1ull << MR) - 1ull)) != 0ull : nprev > m0;
        const bool act = tid < MR * JT && b < nb && bit(amask, b) && j0 + u < H;
        // (this step's input pre-activations are in pre_n: requested behind the previous step's reduction, below.  The initial cell
        //  state of this thread's element is c0_own, loaded once in front of the loop: as a conditional load into a register at this
        //  point it made the compiler wait with s_waitcnt vmcnt(0) - for the owners' stores of the previous step - in front of the hold.)
        float pre[4];
        float* gp = A.gx + (row0 + b) * ld_g + (long long)dir * G + j0 + u;
        const int t1 = dir == 0 ? s + 1 : A.T - 2 - s;
        const bool more = s + 1 < A.T;
        const int nb1 = more ? bs_at(t1) : 0;
        const long long row1 = more ? offs_at(t1) : 0;
        const u64 amask1 = more ? alive_at(t1) : 0ull;
        auto prefetch = [&]() {
            if (ABL && (A.dbg & 1024)) return;
            if (tid < MR * JT && b < nb1 && bit(amask1, b) && j0 + u < H) {
                const float* np = A.gx + (row1 + b) * ld_g + (long long)dir * G + j0 + u;
#pragma unroll
                for (int q = 0; q < 4; ++q) pre_n[q] = np[q * H];
            }
        };
        if (!has_rec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[q] = pre_n[q];
            prefetch();
            fill_some(s + 1 == A.T);
        }
        if (has_rec) {
            constexpr int NF = MTL * CB * 2;                 // fragments: (row tile mt, k block i, plane p)
            const __amdgpu_buffer_rsrc_t h_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
                A.hyt + (((size_t)tp * A.nt16 + tile16) * A.ndir + dir) * tile_elems, 0, A.KP32 * 64, 0x00020000);
            const __amdgpu_buffer_rsrc_t h_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
                A.hyt + (((size_t)tp * A.nt16 + tile16 + (MTL > 1 ? 1 : 0)) * A.ndir + dir) * tile_elems, 0, A.KP32 * 64, 0x00020000);
            const unsigned vin = (unsigned)(kfirst * 2048 + lane * 16);
            const unsigned vb0 = has_pred(m0 + r) ? vin : 0x80000000u;
            const unsigned vb1 = (MTL > 1 && has_pred(m0 + 16 + r)) ? vin : 0x80000000u;
            uint4 a[NF];
            if (!(A.dbg & 8192)) dd.wait();
            unsigned polls = 0;
            for (;;) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int mt = f / (2 * CB), rem = f - mt * 2 * CB;
                    a[f] = __builtin_bit_cast(uint4, mt == 0 ? __builtin_amdgcn_raw_buffer_load_b128(h_rsrc0, vb0, (min(rem >> 1, ilast) * 2 + (rem & 1)) * 1024, 16)
                                                             : __builtin_amdgcn_raw_buffer_load_b128(h_rsrc1, vb1, (min(rem >> 1, ilast) * 2 + (rem & 1)) * 1024, 16));
                }
                if (ABL && (A.dbg & 128)) {           // timing ablation: the operands' values are not waited for
#pragma unroll
                    for (int f = 0; f < NF; ++f) a[f] = make_uint4(0x3c003c00u + s, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
                }
                unsigned m = 0u;
#pragma unroll
                for (int f = 0; f < NF; ++f) m = fold_max16(a[f], m);
                const bool clean = !__any(has_fill(m)) || !alive || (A.dbg & 8192);      // (8192: timing ablation, no waiting at all)
                if (polls == 0) dd.update(clean, s, adapt);
                if (clean) break;
                if (++polls >= A.max_polls || ((polls & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (polls >= A.max_polls && lane == 0) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (A.err_sink) atomicAdd(A.err_sink, 1u);
                    }
                    alive = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            f32x4 acc[MTL][NT];
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero;
            if (!(ABL && (A.dbg & 64)))
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const uint4 ah = a[(mt * CB + i) * 2], al = a[(mt * CB + i) * 2 + 1];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16<false>(al, bh[i][nt], acc[mt][nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16<false>(ah, bl[i][nt], acc[mt][nt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16<false>(ah, bh[i][nt], acc[mt][nt]);
                }
            if (!(ABL && (A.dbg & 65536))) {
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[s & 1][wave][mt * 16 + g4 * 4 + q][nt * 16 + r] = acc[mt][nt][q];
            __syncthreads();
            }
            if (!(ABL && (A.dbg & 32768))) dd.mark();
            fill_some(s + 1 == A.T);          // (behind the barrier: the wavefronts without elements have nothing else to do there, and in front
                                              //  of it the whole workgroup waited for their stores to be issued)
#pragma unroll
            for (int q = 0; q < 4; ++q) pre[q] = pre_n[q];
            if (tid < MR * JT && !(ABL && (A.dbg & 65536))) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cidx = q * JT + u;
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += red[s & 1][w][bl_][cidx];
                    pre[q] += sum * inv;
                }
            }
            // the NEXT step's input pre-activations, into the registers this step's have just left: a whole step until they are used.
            // Placements measured (us per step alone / ms per training step, one box): here 2.43 / 6.76; behind the activations, in front
            // of the hand-off store 2.45 / 6.80; behind the hand-off store 2.74 / 7.02 (the compiler's wait for the registers' previous
            // loads - long complete - is then a vmcnt(0) that includes the store's round trip)
            // (requested in front of the MFMAs, into registers of their own, the compiler copied them over behind the reduction with
            //  s_waitcnt vmcnt(0) - cold lines of about a microsecond - in front of the activations and the hand-off store)
            prefetch();
        }
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f;
        if (act && ABL && (A.dbg & 131072)) {
            ig = pre[0]; fg = pre[1]; gg = pre[2]; og = pre[3];
            c_reg = fg + ig * gg;
            h = og * c_reg;
        } else if (act) {
            ig = sigmoidf_(pre[0]);
            fg = sigmoidf_(pre[1]);
            gg = tanhf_(pre[2]);
            og = sigmoidf_(pre[3]);
            c_reg = fg * ((has_rec && has_pred(b)) ? c_reg : c0_own) + ig * gg;        // (a sequence's first step: its initial cell state)
            h = og * tanhf_(c_reg);
        }
        {
            int plane;
            const unsigned word = pair_word<false>(h * kHScale, lane, &plane);
            // (row-slot batches: an idle slot step writes ZEROS - nobody in this launch waits for them, but the planes then are a
            //  valid operand of the GEMMs that follow, like those of an equal-length batch)
            if ((act || (masked && tid < MR * JT && b < nb && j0 + u < H)) && !(ABL && (A.dbg & 16384))) {
                // (the thread's slot inside a (time, tile, direction) block is a constant - hs_thr -, the block's offset wave-uniform:
                //  one vector add in front of the store instead of three 64-bit multiplications)
                unsigned* dst = reinterpret_cast<unsigned*>(A.hyt) + ((long long)t * hs_step + hs_thr);
                __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (j0 < H && j0 + JT >= H) {                     // owner of the last unit: the padding columns H .. KP32-1 (every step: they carry the pattern too)
            const int pw2 = (A.KP32 - H) >> 1;
            for (int e = tid; e < MR * pw2 * 2 && tid < ACTW * 64; e += ACTW * 64) {
                const int rl = e / (pw2 * 2), rem = e - rl * pw2 * 2, plane = rem / pw2, ce = H + 2 * (rem - plane * pw2);
                if (m0 + rl < nb && (masked || bit(amask, m0 + rl))) {
                    unsigned* tq = reinterpret_cast<unsigned*>(A.hyt + (((size_t)t * A.nt16 + tile16 + (rl >> 4)) * A.ndir + dir) * tile_elems);
                    __hip_atomic_store(tq + handoff_index(ce, plane, rl & 15) / 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (act && !(ABL && (A.dbg & 2048))) {
            // saved activations / row-major outputs: nobody in this launch reads them again - stored with the non-temporal hint, so that
            // they do not displace the operand panels the co-running weight-gradient GEMMs share through the L2s (with the matching
            // loads in the backward kernel: 7.16 -> 7.12 ms per step, profiles/r4_ab_step.txt)
            __builtin_nontemporal_store(ig, gp);
            __builtin_nontemporal_store(fg, gp + H);
            __builtin_nontemporal_store(gg, gp + 2 * H);
            __builtin_nontemporal_store(og, gp + 3 * H);
            const long long o = (row0 + b) * ld_h + dir * H + j0 + u;
            __builtin_nontemporal_store(c_reg, A.c + o);
            __builtin_nontemporal_store(h, A.hy + o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward-through-time.  8 wavefronts, CB = 32-wide k blocks of K = 4H per wavefront (even split), CAB blocks in
// flight per wavefront (re-requested as soon as their MFMAs have consumed them, as in lstm_bwd_persistent_kernel).
//
// TP (equal-length batches whose size is a multiple of 16): the gate gradients leave the kernel a second time as bf16 (hi, lo)
// planes of dgates^T - the operand [4H gate columns][packed rows] of the weight-gradient GEMMs dW = dgates^T [x | h_prev]
// (ptmi_gemm_planes_bf16), in the layout ptmi_pack_planes_t_bf16 would produce from the row-major fp32 tensor: per direction
// [4H / 16 column tiles][rows / 32 k blocks][hi | lo][64 chunks], chunk (k group g, column r) = 8 consecutive packed rows of one
// gate column.  A workgroup's step produces 16 MTL rows x 64 columns; the owners park their halves in LDS as [plane][8-row
// group][column][row], and BEHIND THE NEXT STEP'S BARRIER (no barrier of its own; double-buffered by step parity) every thread
// stores one 16-byte chunk.  With the planes, the hand-off copy (the LSTM input gradient's operand) and the in-kernel bias sums,
// nobody reads the row-major fp32 gate gradients any more: A.dg may be null, and the 155 MB store + two transposing pack
// passes per layer of the B = 32 step go away.
template <int NW, int CB, int MTL, int CABW, int CP = 16, bool UNI = false, bool DAF = false, bool TP = false, bool MSK = false>
__global__ __launch_bounds__(NW * 64, 2) void lstm_bwd_split_kernel(const LstmPersistBwdArgs A) {
    static_assert(DAF, "the flag-protocol form of this kernel (rounds 1-2) is gone: data-as-flag instantiations only");
    int bx, by, dir;
    if (!chain_tile(A.nx, A.nt, A.span, &bx, &by, &dir)) return;
    const int n0 = bx * 16;
    constexpr int MR = 16 * MTL;
    const int m0 = (A.tile0 + by) * MR;
    const int tile16 = (A.tile0 + by) * MTL;
    const int H = A.H, G = 4 * H;
    const long long ld_g = (long long)A.ndir * G, ld_h = (long long)A.ndir * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (A.dbg & 512) __builtin_amdgcn_s_setprio(3);      // experiment: issue priority over co-resident GEMM wavefronts
    const int g4 = lane >> 4, r = lane & 15;
    __shared__ float red[2][NW][MR][20];       // double-buffered by step parity (one barrier per step); pitch 20: conflict-free accumulator writes
    __shared__ unsigned short tbuf[TP ? 2 : 1][2][TP ? MR / 8 : 1][TP ? 64 : 1][8];      // TP: [step parity][plane][8-row group][gate-major column][row]
    int t_pend = -1;                                   // TP: time index whose chunks lie in tbuf[(its step) & 1], not stored yet
    auto flush_tp = [&](int par, int tt) {
        // 2 planes x MR / 8 row groups x 64 columns = 16 MR chunks of 16 B: one per thread; a wavefront reads 1 KB of LDS lane-linear
        // and writes runs of 16 chunks (256 B: the 16 units of one gate) into the planes
        if (TP && tid < 16 * MR && !(ABL && (A.dbg & 2048))) {
            const int unit = tid & 15, gate = (tid >> 4) & 3, plane = (tid >> 6) & 1, rg = tid >> 7;
            if (n0 + unit < H && m0 + rg * 8 < A.max_batch) {
                const int col = gate * H + n0 + unit;
                const long long prow = (long long)tt * A.max_batch + m0 + rg * 8 - A.tp_row0[dir];
                const long long kb = prow >> 5;
                const int kg = (int)(prow & 31) >> 3;
                uint4* dst = A.dgtp + (long long)dir * A.tp_dir_stride + ((((long long)(col >> 4) * A.tp_kb + kb) * 2 + plane) * 64 + kg * 16 + (col & 15));
                *dst = *reinterpret_cast<const uint4*>(&tbuf[par][plane][rg][gate * 16 + unit][0]);
            }
        }
    };
    // MSK: the row-slot instantiation (host checked: masks present, rows = [T, slots]) - the grid's bookkeeping is arithmetic at compile
    // time like UNI's, only the masks are data
    static_assert(!(UNI && MSK), "UNI: no masks");
    const bool uniform = UNI || MSK || A.uniform != 0;        // equal lengths: bookkeeping by arithmetic (see the forward kernel)
    auto bs_at = [&](int t) { if (UNI || MSK) return A.max_batch; return uniform ? A.max_batch : ld_const(A.bs, t); };
    auto offs_at = [&](int t) { if (UNI || MSK) return (long long)t * A.max_batch; return uniform ? (long long)t * A.max_batch : (long long)ld_const(A.offs, t); };
    // row-slot batches (LstmPersistBwdArgs::masks; not in the UNI instantiations): per-step row masks instead of the prefix rules
    typedef unsigned long long u64;
    const bool masked = MSK || (!UNI && A.masks != nullptr);
    // (round 6: a workgroup keeps only ITS rows' bits of a mask word - bit i = row m0 + i, 32-bit - : nine 64-bit mask values in a kernel that
    //  is short of scalar registers cost 59 more v_readlane per step than the uniform instantiation has)
    typedef unsigned u32;
    auto lbit = [&](u32 m, int row) { return ((m >> ((row - m0) & 31)) & 1u) != 0u; };
    auto local = [&](u64 m) { return (u32)(m >> m0) & (MR >= 32 ? 0xffffffffu : ((1u << (MR & 31)) - 1u)); };

    const int nblk = A.G32 >> 5;
    const int base = nblk / NW, extra = nblk - base * NW;
    const int kb0 = __builtin_amdgcn_readfirstlane(wave * base + min(wave, extra));
    const int nbw = __builtin_amdgcn_readfirstlane(base + (wave < extra ? 1 : 0));        // <= CB (host checked)
    const int kfirst = __builtin_amdgcn_readfirstlane(min(kb0, nblk - 1));
    const int ilast = __builtin_amdgcn_readfirstlane(max(nbw - 1, 0));
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // resident slice of W_hh^T as bf16 halves: lane (hidden unit n0 + r, k group g4)
    uint4 bh[CB], bl[CB];
    {
        const bool bv = n0 + r < H;
        const float* bp = A.wt + ((long long)dir * H + (bv ? n0 + r : 0)) * G;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int k = (kb0 + i) * 32 + g4 * 8;
            const bool in = bv && i < nbw;
            const f32x4 w0 = (in && k + 4 <= G) ? *reinterpret_cast<const f32x4*>(bp + (k + 4 <= G ? k : 0)) : zero;
            const f32x4 w1 = (in && k + 8 <= G) ? *reinterpret_cast<const f32x4*>(bp + (k + 8 <= G ? k + 4 : 0)) : zero;
            const float v[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            split8<true>(v, &bh[i], &bl[i]);
        }
    }
    unsigned* const err = A.flags + A.err_off;
    const int bl_ = (tid >> 4) & (MR - 1), jl = tid & 15;
    const int b = m0 + bl_, j = n0 + jl;
    const size_t tile_elems = (size_t)A.G32 * 16;          // floats per (time, 16-row tile, direction)
    bool alive = true;
    DafHold dd{(A.dbg >> 16) & 0xff ? (unsigned)((A.dbg >> 16) & 0xff) * 4u : (MTL == 1 ? 92u : 100u), 0u, 0ull};      // (see the forward kernel)
    const bool adapt = !(A.dbg & (1 << 24));
    dd.mark();
    float dc_state = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f;
    float amax = 0.f;                                      // max |dgates| this thread has produced
    // (Tried here as well: the forward kernel's L2 warm-up of the owners' cold loads - dhy, gates, c two steps ahead - by the
    //  last wavefront.  4.52 instead of 4.09 us per step at B = 32: that wavefront carries a k slice of the product, and its
    //  cold loads sit in front of its next operand loads in the in-order return queue.)

    // bookkeeping one step ahead (see the forward kernel): this step needs the batch sizes of time index t, of the one processed
    // before (t_n) and of the one processed next (t_p = the forward-sense predecessor, for c_{t-1}); the latter is loaded
    // one iteration early
    auto tindex = [&](int step) { return dir == 0 ? A.T - 1 - step : step; };
    const int s0 = A.s_begin, s1 = A.s_end < 0 ? A.T : A.s_end;        // this launch's range of processing steps
    int nb_c = bs_at(tindex(s0)), nb_n = s0 > 0 ? bs_at(tindex(s0 - 1)) : 0;    // this step's / the previously processed time index
    long long row_c = offs_at(tindex(s0));
    int nb_f = s0 + 1 < A.T ? bs_at(tindex(s0 + 1)) : 0;        // the time index processed next
    long long row_f = s0 + 1 < A.T ? offs_at(tindex(s0 + 1)) : 0;
    // hand-off stores of this thread's element (32-bit words: lanes 2i / 2i+1 store the (hi, lo) pair words of columns c, c + 1 of each gate)
    const long long hs_step = (long long)A.nt16 * A.ndir * (long long)tile_elems;         // (tile_elems: fp32 = 32-bit words per block)
    const long long hs_thr = ((long long)(tile16 + (bl_ >> 4)) * A.ndir + dir) * (long long)tile_elems;
    int hs_gate[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) hs_gate[g] = handoff_index(g * H + (j & ~1), lane & 1, bl_ & 15) / 2;
    const float c0_own = (A.c0 && tid < 16 * MR && b < A.max_batch && j < H) ? A.c0[((long long)dir * A.max_batch + b) * H + j] : 0.f;
    // ... and the gradient w.r.t. its FINAL cell state (a conditional load behind the barrier made the wait for the saved activations a
    // vmcnt(0): it then also waited for the plane chunk the thread had just stored)
    const float dcn_own = (A.dcn && tid < 16 * MR && b < A.max_batch && j < H) ? A.dcn[((size_t)dir * A.max_batch + b) * H + j] : 0.f;
    const bool carries = tid < 16 * MR && b < A.max_batch && j < H && A.dc_carry != nullptr;
    if (s0 > 0 && carries) dc_state = A.dc_carry[((size_t)dir * A.max_batch + b) * H + j];
    // row-slot batches: the mask words of this step's time index (alive rows, rows at a sequence boundary in this direction's sense)
    // and of the previously processed one
    const int msel = dir == 0 ? 1 : 2;
    u32 mk_a = 0u, mk_b = 0u, pv_a = 0u, pv_b = 0u;
    if (masked) {
        mk_a = local(ld_const64(A.masks, 3 * tindex(s0)));
        mk_b = local(ld_const64(A.masks, 3 * tindex(s0) + msel));
        if (s0 > 0) {
            pv_a = local(ld_const64(A.masks, 3 * tindex(s0 - 1)));
            pv_b = local(ld_const64(A.masks, 3 * tindex(s0 - 1) + msel));
        }
    }
    for (int s = s0; s < s1; ++s) {
        const int t = tindex(s);
        const int nb = nb_c;
        const long long row0 = row_c;
        const int tn = dir == 0 ? t + 1 : t - 1;
        const int nnext = s > 0 ? min(nb_n, nb) : 0;
        const bool has_p = s + 1 < A.T;
        const int npv = has_p ? min(nb_f, nb) : 0;
        const long long prow0 = has_p ? row_f : 0;
        // rotate, and request the time index after the next one (first used in the NEXT iteration)
        nb_n = nb;
        nb_c = nb_f;
        row_c = row_f;
        {
            const int t2 = tindex(min(s + 2, A.T - 1));       // clamped, unconditional (see the forward kernel)
            nb_f = bs_at(t2);
            row_f = offs_at(t2);
        }
        // masks of this step: rows alive; rows whose NEXT processed... whose previously processed step (time index tn) belongs to the
        // same sequence (its gate gradients reach this step through W_hh; the cell-state gradient carries over); rows whose cell
        // state c_{t-1} (forward sense of this direction) exists
        // (the mask words arrive one iteration ahead - mk_a / mk_b below - and the previously processed index's are kept: loaded at
        //  the loop head they put a scalar-load round trip into every step of the chain: 4.8 instead of 3.9 us per step)
        u32 amask = ~0u, smask = ~0u, cmask = ~0u;
        if (masked) {
            const u32 cur_a = mk_a, cur_b = mk_b;
            const int t2m = tindex(min(s + 1, A.T - 1));          // clamped, unconditional
            mk_a = local(ld_const64(A.masks, 3 * t2m));
            mk_b = local(ld_const64(A.masks, 3 * t2m + msel));
            amask = cur_a;
            const bool tn_ok = tn >= 0 && tn < A.T;
            smask = tn_ok ? pv_a & ~pv_b : 0u;
            cmask = cur_a & ~cur_b;
            pv_a = cur_a;
            pv_b = cur_b;
        }
        auto has_succ = [&](int row) { return row < nnext && lbit(smask, row); };
        const bool has_rec = masked ? smask != 0u && nnext > m0 : nnext > m0;
        const bool act = tid < 16 * MR && b < nb && lbit(amask, b) && j < H;
        // The saved activations of this thread's element: UNCONDITIONAL loads from clamped addresses in the wavefronts that own
        // elements (a wave-uniform branch).  With `if (act) x = load` the compiler zero-initialises the destination registers at
        // the loop head and, to protect them, waits there for every memory operation of the previous step (s_waitcnt
        // vmcnt(0): the trailing row-major stores) BEFORE it issues these loads - cold HBM lines that need the whole step
        // to arrive.  Values of threads without an element are never used.
        float dh, ig, fg, gg, og, cn, cprev;
        const long long og_ = (row0 + b) * ld_g + (long long)dir * G + j;
        if (__builtin_amdgcn_readfirstlane(wave) < (16 * MR) / 64 && !(A.dbg & 4096)) {        // scalar branch (4096: timing ablation)
            const int bb = min(b, nb - 1), jj = min(j, H - 1);
            const long long ohc = (row0 + bb) * ld_h + dir * H + jj;
            const long long ogc = (row0 + bb) * ld_g + (long long)dir * G + jj;
            dh = __builtin_nontemporal_load(A.dhy + ohc);
            ig = __builtin_nontemporal_load(A.gates + ogc);
            fg = __builtin_nontemporal_load(A.gates + ogc + H);
            gg = __builtin_nontemporal_load(A.gates + ogc + 2 * H);
            og = __builtin_nontemporal_load(A.gates + ogc + 3 * H);
            cn = __builtin_nontemporal_load(A.c + ohc);
            cprev = A.c[(prow0 + min(bb, max(npv - 1, 0))) * ld_h + dir * H + jj];
        } else {        // never used: "defined" without an instruction (a zero store here is hoisted in front of the branch,
                        // where it brings the wait back)
            asm volatile("" : "=v"(dh), "=v"(ig), "=v"(fg), "=v"(gg), "=v"(og), "=v"(cn), "=v"(cprev));
        }
        const bool has_prev_c = b < npv && lbit(cmask, b);
        // (the initial cell state of this thread's element is c0_own, loaded once in front of the loop: as `c0v = 0; if (...) c0v =
        //  A.c0[...]` at this point, the zero-initialisation of a register that a load of the previous iteration may still own made the
        //  compiler wait HERE with s_waitcnt vmcnt(0) - for the cold loads just issued and for the previous step's stores, in front of the
        //  hold and the operand requests of every step)
        if (has_rec) {
            // data-as-flag: no poll, no barrier in front of the operand loads.  The k blocks of this wavefront go through two
            // register sets in passes of CAB blocks: pass p is checked (and requested again until it is free of the fill
            // pattern), pass p + 1 is requested, pass p is multiplied.
            const float* const tbase = A.dgt + (((size_t)tn * A.nt16 + tile16) * A.ndir + dir) * tile_elems;
            const unsigned vin = (unsigned)(kfirst * 2048 + lane * 16);
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tbase), 0, A.G32 * 64, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(tbase + (MTL > 1 ? (size_t)A.ndir * tile_elems : 0)), 0, A.G32 * 64, 0x00020000);
            const unsigned vb0 = has_succ(m0 + r) ? vin : 0x80000000u;
            const unsigned vb1 = (MTL > 1 && has_succ(m0 + 16 + r)) ? vin : 0x80000000u;
            constexpr int NB = MTL * CB;
            constexpr int CAB = CABW < NB ? CABW : NB;
            constexpr int NP = (NB + CAB - 1) / CAB;
            auto fragment = [&](int blk, int p) {           // compile-time constants after unrolling
                const int mt = blk / CB, i = blk - mt * CB;
                return __builtin_bit_cast(uint4, mt == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rs0, vb0, (min(i, ilast) * 2 + p) * 1024, 16)
                                                         : __builtin_amdgcn_raw_buffer_load_b128(rs1, vb1, (min(i, ilast) * 2 + p) * 1024, 16));
            };
            uint4 fh[2][CAB], fl[2][CAB];
            auto request = [&](int pass, int set) {
#pragma unroll
                for (int i = 0; i < CAB; ++i)
                    if (pass * CAB + i < NB) {
                        fh[set][i] = fragment(pass * CAB + i, 0);
                        fl[set][i] = fragment(pass * CAB + i, 1);
                    }
            };
            f32x4 acc3[MTL][3];
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc3[mt][k] = zero;
            if (!(A.dbg & 8192)) dd.wait();
            request(0, 0);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int set = pass & 1;
                unsigned polls = 0;
                for (;;) {
                    unsigned m = 0u;
#pragma unroll
                    for (int i = 0; i < CAB; ++i)
                        if (pass * CAB + i < NB) {
                            m = fold_max16(fh[set][i], m);
                            m = fold_max16(fl[set][i], m);
                        }
                    const bool clean = !__any(has_fill(m)) || !alive || (A.dbg & 8192);      // (8192: timing ablation, no waiting at all)
                    if (pass == 0 && polls == 0) dd.update(clean, s, adapt);
                    if (clean) break;
                    if (++polls >= A.max_polls || ((polls & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (polls >= A.max_polls && lane == 0) {
                            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (A.err_sink) atomicAdd(A.err_sink, 1u);
                        }
                        alive = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    request(pass, set);
                }
                if (pass + 1 < NP) request(pass + 1, set ^ 1);
                if (ABL && (A.dbg & 128)) {           // timing ablation: the operands' values are not waited for
#pragma unroll
                    for (int i = 0; i < CAB; ++i) fh[set][i] = fl[set][i] = make_uint4(0x3c003c00u + s, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
                }
                if (!(ABL && (A.dbg & 64)))
#pragma unroll
                for (int i = 0; i < CAB; ++i) {
                    const int blk = pass * CAB + i;
                    if (blk < NB) {
                        acc3[blk / CB][0] = mma16<true>(fl[set][i], bh[blk % CB], acc3[blk / CB][0]);
                        acc3[blk / CB][1] = mma16<true>(fh[set][i], bl[blk % CB], acc3[blk / CB][1]);
                        acc3[blk / CB][2] = mma16<true>(fh[set][i], bh[blk % CB], acc3[blk / CB][2]);
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[s & 1][wave][mt * 16 + g4 * 4 + q][r] = (acc3[mt][0][q] + acc3[mt][1][q]) + acc3[mt][2][q];
            __syncthreads();
            dd.mark();
            if (tid < 16 * MR && has_succ(b)) {
                float part[NW];              // all partial sums requested, then added in order (one LDS round trip, not NW / 2 dependent ones)
#pragma unroll
                for (int w = 0; w < NW; ++w) part[w] = red[s & 1][w][bl_][jl];
                __builtin_amdgcn_sched_barrier(0);
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += part[w];
                dh += sum;
            }
            // (the previous step's plane chunks - parked in front of this barrier - are stored behind this step's hand-off stores,
            //  below: stored here, the chunk was the youngest vector-memory operation when the gate arithmetic waits for the saved
            //  activations, and the wait became a vmcnt(0) that included the store's round trip)
        }
        if (TP && !has_rec && t_pend >= 0) {       // a step without the chain's barrier (row slots: no row of the tile continues a sequence)
            __syncthreads();
            flush_tp((s - 1) & 1, t_pend);
            t_pend = -1;
        }
        float gi = 0.f, gf = 0.f, gc = 0.f, go = 0.f;
        int t_flush = -1;
        if (act) {
            float dc = has_succ(b) ? dc_state : dcn_own;         // (a sequence's last step: the gradient of its final cell state)
            const float tc = tanhf_(cn);
            const float d_o = dh * tc;
            dc += dh * og * (1.f - tc * tc);
            const float d_i = dc * gg;
            const float d_g = dc * ig;
            const float d_f = dc * (has_prev_c ? cprev : c0_own);                      // (a sequence's first step: its initial cell state)
            dc_state = dc * fg;
            gi = d_i * ig * (1.f - ig);
            gf = d_f * fg * (1.f - fg);
            gc = d_g * (1.f - gg * gg);
            go = d_o * og * (1.f - og);
            sb0 += gi;
            sb1 += gf;
            sb2 += gc;
            sb3 += go;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(gi), fabsf(gf))), fmaxf(fabsf(gc), fabsf(go)));
        }
        // hand-off copy: bf16 (hi, lo) planes, written through; lanes 2i / 2i+1 (columns c, c+1) trade one half
        {
            int plane;
            const unsigned w0 = pair_word<true>(gi, lane, &plane);
            const unsigned w1 = pair_word<true>(gf, lane, &plane);
            const unsigned w2 = pair_word<true>(gc, lane, &plane);
            const unsigned w3 = pair_word<true>(go, lane, &plane);
            t_flush = (TP && has_rec) ? t_pend : -1;      // the previous step's chunks: behind the hand-off stores
            if (TP) t_pend = t;          // (every thread: the flush in a step without the chain's barrier brings its own barrier)
            if ((act || (masked && tid < 16 * MR && b < nb && j < H)) && !(ABL && (A.dbg & 16384))) {       // (row slots: zeros for an idle slot step, see the forward kernel)
                // (block offset wave-uniform, the thread's four slots constants: see the forward kernel)
                unsigned* tq = reinterpret_cast<unsigned*>(A.dgt) + ((long long)t * hs_step + hs_thr);
                const unsigned ws_[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    __hip_atomic_store(tq + hs_gate[g], ws_[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (TP && tid < 16 * MR) {       // this step's halves parked for the planes of dgates^T - behind the hand-off stores, which the chain waits
                                         // for (every row of the tile: rows past the batch / units past H carry zeros; flush_tp skips them)
            const float gv[4] = {gi, gf, gc, go};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned hi, lo;
                split1<true>(gv[g], &hi, &lo);
                tbuf[s & 1][0][bl_ >> 3][g * 16 + jl][bl_ & 7] = (unsigned short)hi;
                tbuf[s & 1][1][bl_ >> 3][g * 16 + jl][bl_ & 7] = (unsigned short)lo;
            }
        }
        if (TP && t_flush >= 0) flush_tp((s - 1) & 1, t_flush);
        if (A.G32 != G && n0 + 16 >= H && n0 < H) {       // owner of the last unit tile: zero the padding columns 4H .. G32-1
            const int pw2 = (A.G32 - G) >> 1;
            for (int e = tid; e < MR * pw2 * 2; e += NW * 64) {
                const int rl = e / (pw2 * 2), rem = e - rl * pw2 * 2, plane = rem / pw2, ce = G + 2 * (rem - plane * pw2);
                if (m0 + rl < nb && (masked || lbit(amask, m0 + rl))) {
                    unsigned* tq = reinterpret_cast<unsigned*>(A.dgt + (((size_t)t * A.nt16 + tile16 + (rl >> 4)) * A.ndir + dir) * tile_elems);
                    __hip_atomic_store(tq + handoff_index(ce, plane, rl & 15) / 2, 0u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (masked && !act && tid < 16 * MR && b < nb && j < H && A.dg) {       // a slot's idle step: its row exists in the buffers - zeros
            float* dgp = A.dg + og_;
            dgp[0] = dgp[H] = dgp[2 * H] = dgp[3 * H] = 0.f;
        }
        if (act && (!TP || A.dg)) {                   // row-major dgates (what the GEMMs read): nobody in this launch waits for them
            float* dgp = A.dg + og_;
            dgp[0] = gi;
            dgp[H] = gf;
            dgp[2 * H] = gc;
            dgp[3 * H] = go;
        }
    }
    // cell-state gradient behind this launch's last step: what the next range continues from, and after the last range the
    // gradient w.r.t. the initial cell state c0 (a sequence's thread keeps it from that sequence's first time step on)
    if (carries) A.dc_carry[((size_t)dir * A.max_batch + b) * H + j] = dc_state;
    // bias gradient = sum of dgates over all rows; max |dgates| for the GEMMs that follow (operand scale)
    float* const fold = &red[0][0][0][0];
    __syncthreads();
    if (TP && t_pend >= 0) flush_tp((s1 - 1) & 1, t_pend);       // the last step's chunks
    if (tid < 16 * MR) {
        fold[(0 * MR + bl_) * 16 + jl] = sb0;
        fold[(1 * MR + bl_) * 16 + jl] = sb1;
        fold[(2 * MR + bl_) * 16 + jl] = sb2;
        fold[(3 * MR + bl_) * 16 + jl] = sb3;
    }
    __syncthreads();
    if (tid < 64 && n0 + jl < H) {
        const int g = tid >> 4;
        float sum = 0.f;
#pragma unroll
        for (int rr = 0; rr < MR; ++rr) sum += fold[(g * MR + rr) * 16 + jl];
        atomicAdd(A.dbias + (size_t)dir * G + g * H + n0 + jl, sum);
    }
    unsigned m = __float_as_uint(amax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0 && m != 0u && A.dg_amax) atomicMax(A.dg_amax, min(m, 0x7f7fffffu));
}

// ---------------------------------------------------------------------------------------------------------------
bool fwd_daf_applies(int jt, bool small, bool one_per_cu) {
    return one_per_cu && ((small && (jt == 8 || jt == 12 || jt == 16)) || (!small && jt == 12));
}

bool bwd_daf_applies() { return true; }

int launch_fwd_split(const LstmPersistArgs& A, int jt, bool small, bool one_per_cu, dim3 grid, hipStream_t st, bool daf) {
    constexpr int NW = 8, CB = 3;
    const dim3 block(NW * 64);
    if (daf) {
        if (jt == 16 && small)
            hipLaunchKernelGGL((lstm_fwd_daf_kernel<16, NW, CB, 1>), grid, block, 0, st, A);
        else if (jt == 12 && small)
            hipLaunchKernelGGL((lstm_fwd_daf_kernel<12, NW, CB, 1>), grid, block, 0, st, A);
        else if (jt == 8 && small)
            hipLaunchKernelGGL((lstm_fwd_daf_kernel<8, NW, CB, 1>), grid, block, 0, st, A);
        else
            hipLaunchKernelGGL((lstm_fwd_daf_kernel<12, NW, CB, 2>), grid, block, 0, st, A);
        return launch_status();
    }
    return PTMI_E_UNSUPPORTED;        // (a tile shape without a data-as-flag instantiation: the caller has checked fwd_daf_applies)
}

int launch_bwd_split(const LstmPersistBwdArgs& A, int mtl, unsigned nwg, hipStream_t st) {
    // equal-length batches: an instantiation without the PackedSequence tables (no loads at the loop head)
    const bool uni = A.uniform != 0 && !A.masks;         // (row-slot batches: the instantiation that reads the per-step masks)
    if (A.dgtp && A.masks && A.uniform) {          // row slots (rows = [T, slots]) + dgates^T planes
        if (mtl == 2)
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 2, 3, 16, false, true, true, true>), dim3(nwg), dim3(512), 0, st, A);
        else
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 1, 3, 16, false, true, true, true>), dim3(nwg), dim3(512), 0, st, A);
        return launch_status();
    }
    if (A.dgtp && A.masks) {
        if (mtl == 2)
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 2, 3, 16, false, true, true>), dim3(nwg), dim3(512), 0, st, A);
        else
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 1, 3, 16, false, true, true>), dim3(nwg), dim3(512), 0, st, A);
        return launch_status();
    }
    if (A.dgtp) {          // + dgates^T as bf16 planes (host checked: equal lengths, batch a multiple of 16)
        if (mtl == 2)
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 2, 3, 16, true, true, true>), dim3(nwg), dim3(512), 0, st, A);
        else
            hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 1, 3, 16, true, true, true>), dim3(nwg), dim3(512), 0, st, A);
        return launch_status();
    }
    if (mtl == 2 && uni)
        hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 2, 3, 16, true, true>), dim3(nwg), dim3(512), 0, st, A);
    else if (mtl == 2)
        hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 2, 3, 16, false, true>), dim3(nwg), dim3(512), 0, st, A);
    else if (uni)
        hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 1, 3, 16, true, true>), dim3(nwg), dim3(512), 0, st, A);
    else
        hipLaunchKernelGGL((lstm_bwd_split_kernel<8, 10, 1, 3, 16, false, true>), dim3(nwg), dim3(512), 0, st, A);
    return launch_status();
}

}  // namespace ptmi
