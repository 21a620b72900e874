// STFT / iSTFT / fused PIT feature front-end for gfx950 (MI355X).
//
// Replaces padertorch/ops/_stft.py:103-174 (STFT.__call__), :176-263 (STFT.inverse) and
// padertorch/contrib/examples/source_separation/pit/data.py:49-77 (pre_batch_transform).
//
// Design:
//  * one 256-thread workgroup owns FPB consecutive frames of one batch row; the samples those
//    frames cover ((FPB-1)*shift + L floats) are staged ONCE into LDS with coalesced dword loads
//    (fading / frame padding = predicated zero fill, bit-exact framing, no padded copy in HBM);
//  * a real FFT of `size` points is a complex FFT of M = size/2 = R1*R2 points on packed
//    even/odd samples, done as two in-register FFTs (R1 then R2 points per lane, LPF = max(R1,R2)
//    lanes per frame, 64/LPF frames per wavefront) with ONE transposition through LDS
//    (pitch R2+1 -> conflict free ds_write_b64/ds_read_b64), window and inter-stage twiddles
//    held in VGPRs;
//  * the hermitian split (and |.| / cos-phase epilogues) reads the natural-order spectrum from
//    LDS and each wavefront streams its frames' rows out as one contiguous run of HBM stores.
//  * the inverse runs the same machinery backwards and overlap-adds from LDS; halo frames are
//    recomputed (ceil(L/shift)-1 per workgroup) so the result is deterministic (no atomics).
// The kernels are HBM-bound (2568 algorithmic bytes per frame at size 512 / shift 128).
#include "common.h"
#include "fft_regs.h"

namespace ptmi {

// ------------------------------------------------------------------------------------------------
template <int R1_, int R2_>
struct Plan {
    static constexpr int R1 = R1_, R2 = R2_;
    static constexpr int M = R1 * R2;      // complex FFT length
    static constexpr int F = M + 1;        // one-sided bins
    static constexpr int SIZE = 2 * M;     // real FFT length
    static constexpr int LPF = R1 > R2 ? R1 : R2;  // lanes per frame
    static constexpr int FPW = 64 / LPF;           // frames per wavefront
    static constexpr int FPB = 4 * FPW;            // frames per 256-thread workgroup
    static constexpr int P = R2 + 1;               // transposition pitch (complex)
    static constexpr int FS0 = (R1 * P > F ? R1 * P : F);
    static constexpr int FS = (FS0 + 1) & ~1;      // per-frame LDS region (complex), even
    static constexpr int NIT = (FPW * F + 63) / 64;  // epilogue items per lane
};

struct FwdArgs {
    const float* x;
    const float* s;  // features kernel: sources [batch, K, row_stride]
    const int32_t* row_samples;
    const float* window;
    const cpx* twiddle;
    float* out;      // stft: spectrum; features: Y_abs
    float* X_abs;
    float* cos_pd;
    long long x_row_stride;
    long long num_samples;
    long long out_frames;
    int nchunks;
    int layout;
    int K;
    float edge_scale;
    Geo g;
};

// W_size^j for 0 <= j < size from the half-circle table (j = 0..M): W^(j) = -W^(j-M) for j > M.
template <int M>
__device__ __forceinline__ cpx tw_full(const cpx* tw, int j) {
    if (j > M) {
        const cpx v = tw[j - M];
        return cpx{-v.x, -v.y};
    }
    return tw[j];
}

// LDS carve-up shared by the forward kernels (all offsets multiples of 16 bytes):
//   buf  [FPB][FS] cpx   per-frame transposition / natural-order spectrum
//   tws  [F+1]     cpx   split twiddles W_size^k, k = 0..M
//   tw1t [R1][LPF] cpx   inter-stage twiddles W_M^(l*k1), lane-contiguous
//   win  [SIZE]    float analysis window, zero beyond window_length
//   sig  [...]     float staged samples of the FPB frames, zero-filled tail up to SIZE
template <class PL>
struct FwdLds {
    cpx* buf;
    cpx* tws;
    cpx* tw1t;
    float* win;
    float* sig;
    __device__ __forceinline__ explicit FwdLds(char* smem) {
        buf = reinterpret_cast<cpx*>(smem);
        tws = buf + PL::FPB * PL::FS;
        tw1t = tws + PL::F + 1;
        win = reinterpret_cast<float*>(tw1t + PL::R1 * PL::LPF);
        sig = win + PL::SIZE;
    }
    static size_t bytes(int shift) {
        const size_t chunk = (size_t)(PL::FPB - 1) * shift + PL::SIZE;
        return sizeof(cpx) * ((size_t)PL::FPB * PL::FS + PL::F + 1 + PL::R1 * PL::LPF) +
               sizeof(float) * (PL::SIZE + chunk + 2 * PL::LPF + 8);
    }
};

// Fill the constant tables (once per workgroup, coalesced; the tables are L2 resident).
template <class PL>
__device__ __forceinline__ void load_tables(const FwdLds<PL>& S, const float* __restrict__ window,
                                            const cpx* __restrict__ twiddle, int L, int tid) {
    for (int i = tid; i <= PL::M; i += 256) S.tws[i] = twiddle[i];
    for (int i = tid; i < PL::SIZE; i += 256) S.win[i] = (i < L) ? window[i] : 0.f;
    for (int i = tid; i < PL::R1 * PL::LPF; i += 256) {
        const int k1 = i / PL::LPF, l = i - k1 * PL::LPF;
        S.tw1t[i] = tw_full<PL::M>(twiddle, (2 * l * k1) % PL::SIZE);   // W_M^(l*k1) = W_size^(2*l*k1)
    }
}

// Two-step complex FFT of one frame.  On entry lane l (< R2) holds a[i1] = in[R2*i1 + l].
// On exit fbuf[k], k < M, holds the transform in natural order.  Contains block-wide barriers.
template <class PL, bool INV>
__device__ __forceinline__ void fft_to_lds(cpx (&a)[PL::R1], cpx* fbuf, int l, const cpx* tw1t) {
    constexpr int R1 = PL::R1, R2 = PL::R2, P = PL::P;
    if (l < R2) {
        fft_dif<R1, INV>(a);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const cpx v = a[bitrev<R1>(k1)];
            const cpx w = tw1t[k1 * PL::LPF + l];
            fbuf[k1 * P + l] = INV ? cmulc(v, w) : cmul(v, w);
        }
    }
    __syncthreads();
    cpx c[R2];
    if (l < R1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = fbuf[l * P + n2];
    }
    __syncthreads();
    if (l < R1) {
        fft_dif<R2, INV>(c);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) fbuf[l + R1 * k2] = c[bitrev<R2>(k2)];
    }
    __syncthreads();
}

// Stage the samples of frames [t0, t0+FPB) of one row into LDS: zero outside [0, n_b) (fading and
// frame padding) and zero up to SIZE past the last frame start (so the unpredicated window loads
// below never see uninitialised LDS).  Row offsets fit in 32 bits (checked on the host).
template <class PL>
__device__ __forceinline__ void stage_signal(float* sig, const float* __restrict__ xrow, int n_b,
                                             int t0, const Geo& g, int tid) {
    const int chunk = (PL::FPB - 1) * g.shift + PL::SIZE;
    const int v0 = t0 * g.shift - g.pad_left;
    for (int i = tid; i < chunk; i += 256) {
        const int xi = v0 + i;
        float v = 0.f;
        if (xi >= 0 && xi < n_b) v = xrow[xi];
        sig[i] = v;
    }
}

// Load the windowed, even/odd packed samples of this lane's frame into registers
// (a[n1] = x_w[2 (R2 n1 + l)] + i x_w[2 (R2 n1 + l) + 1]).  No predicates: win is 0 beyond L.
template <class PL>
__device__ __forceinline__ void load_windowed(cpx (&a)[PL::R1], const float* sf, const float* win, int l) {
#pragma unroll
    for (int n1 = 0; n1 < PL::R1; ++n1) {
        const int k = 2 * (PL::R2 * n1 + l);
        a[n1] = cpx{sf[k] * win[k], sf[k + 1] * win[k + 1]};
    }
}

// Hermitian split: bin k of the size-point real FFT from the natural-order M-point complex FFT.
template <class PL>
__device__ __forceinline__ cpx split_bin(const cpx* zb, const cpx* tws, int k) {
    constexpr int M = PL::M;
    const cpx z1 = zb[k & (M - 1)];
    const cpx z2 = zb[(M - k) & (M - 1)];
    const cpx w = tws[k];
    const float ex = 0.5f * (z1.x + z2.x), ey = 0.5f * (z1.y - z2.y);  // E = (Z[k] + conj Z[M-k]) / 2
    const float dx = 0.5f * (z1.x - z2.x), dy = 0.5f * (z1.y + z2.y);  // D = (Z[k] - conj Z[M-k]) / 2
    // X = E + (-i D) * w,  -i D = (dy, -dx)
    return cpx{ex + dy * w.x + dx * w.y, ey + dy * w.y - dx * w.x};
}

// ------------------------------------------------------------------------------------------------
template <class PL>
__global__ __launch_bounds__(256, 3) void stft_fwd_kernel(const FwdArgs A) {
    constexpr int M = PL::M, F = PL::F, FPW = PL::FPW, FPB = PL::FPB, FS = PL::FS, LPF = PL::LPF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FwdLds<PL> S(smem);

    const int tid = threadIdx.x;
    const int b = blockIdx.x / A.nchunks;
    const int t0 = (blockIdx.x - b * A.nchunks) * FPB;
    const int n_b = A.row_samples ? A.row_samples[b] : (int)A.num_samples;
    const int frames_b = (int)row_frames_of(A.g, n_b);
    const int lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    const int fb = wave * FPW + fl;

    stage_signal<PL>(S.sig, A.x + (long long)b * A.x_row_stride, n_b, t0, A.g, tid);
    load_tables<PL>(S, A.window, A.twiddle, A.g.L, tid);
    __syncthreads();

    cpx a[PL::R1];
    load_windowed<PL>(a, S.sig + fb * A.g.shift, S.win, l);
    fft_to_lds<PL, false>(a, S.buf + fb * FS, l, S.tw1t);

    // epilogue: this wavefront's FPW frames are FPW*F consecutive complex outputs of one row
    const int tw0 = t0 + wave * FPW;
    const int nvalid = min(FPW, (int)A.out_frames - tw0) * F;     // outputs inside [0, out_frames)
    float* __restrict__ orow = A.out + ((long long)b * A.out_frames + tw0) * (2 * F);
    const cpx* zw = S.buf + wave * FPW * FS;
    const float es = A.edge_scale;
    for (int idx = lane; idx < nvalid; idx += 64) {
        const int f = idx / F, k = idx - f * F;
        cpx X = split_bin<PL>(zw + f * FS, S.tws, k);
        if (k == 0 || k == M) X = cpx{X.x * es, es == 1.f ? X.y : 0.f};
        if (tw0 + f >= frames_b) X = cpx{0.f, 0.f};
        if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
            reinterpret_cast<float2*>(orow)[idx] = make_float2(X.x, X.y);
        } else {
            orow[f * 2 * F + k] = X.x;
            orow[f * 2 * F + F + k] = X.y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused PIT front-end: Y_abs [B,T,F], X_abs / cos_phase_difference [B,T,K,F].
// cos(angle(Y) - angle(X)) = Re(Y conj X) / (|Y| |X|), with angle(0) := 0 like np.angle.
template <class PL>
__global__ __launch_bounds__(256, 3) void pit_features_kernel(const FwdArgs A) {
    constexpr int F = PL::F, FPW = PL::FPW, FPB = PL::FPB, FS = PL::FS, LPF = PL::LPF;
    constexpr int NIT = PL::NIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FwdLds<PL> S(smem);

    const int tid = threadIdx.x;
    const int b = blockIdx.x / A.nchunks;
    const int t0 = (blockIdx.x - b * A.nchunks) * FPB;
    const int n_b = A.row_samples ? A.row_samples[b] : (int)A.num_samples;
    const int frames_b = (int)row_frames_of(A.g, n_b);
    const int lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    const int fb = wave * FPW + fl;
    const int tw0 = t0 + wave * FPW;
    const int nvalid = min(FPW, (int)A.out_frames - tw0) * F;
    const cpx* zw = S.buf + wave * FPW * FS;

    load_tables<PL>(S, A.window, A.twiddle, A.g.L, tid);

    float yr[NIT], yi[NIT];  // mixture spectrum of this lane's epilogue items
    const int nsig = A.s ? A.K + 1 : 1;
    for (int q = 0; q < nsig; ++q) {
        const float* row = (q == 0) ? A.x + (long long)b * A.x_row_stride
                                    : A.s + ((long long)b * A.K + (q - 1)) * A.x_row_stride;
        stage_signal<PL>(S.sig, row, n_b, t0, A.g, tid);
        __syncthreads();
        cpx a[PL::R1];
        load_windowed<PL>(a, S.sig + fb * A.g.shift, S.win, l);
        fft_to_lds<PL, false>(a, S.buf + fb * FS, l, S.tw1t);
        // output rows of this wavefront: Y_abs (tw0.., F) contiguous; X_abs/cos (t, q-1, F) rows
        float* __restrict__ yo = A.out + ((long long)b * A.out_frames + tw0) * F;
        const long long xo = (((long long)b * A.out_frames + tw0) * A.K + (q - 1)) * F;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            if (idx < nvalid) {
                const int f = idx / F, k = idx - f * F;
                cpx X = split_bin<PL>(zw + f * FS, S.tws, k);
                if (tw0 + f >= frames_b) X = cpx{0.f, 0.f};
                const float p = X.x * X.x + X.y * X.y;
                const float r = __builtin_amdgcn_rsqf(p);          // 1/|X| (inf at 0, handled below)
                const float mag = p > 0.f ? p * r : 0.f;
                if (q == 0) {
                    yr[it] = X.x;
                    yi[it] = X.y;
                    yo[idx] = mag;
                } else {
                    const float py = yr[it] * yr[it] + yi[it] * yi[it];
                    const float ry = __builtin_amdgcn_rsqf(py);
                    // unit phasors (1, 0) for zero magnitudes
                    const float cy = py > 0.f ? yr[it] * ry : 1.f, sy = py > 0.f ? yi[it] * ry : 0.f;
                    const float cx = p > 0.f ? X.x * r : 1.f, sx = p > 0.f ? X.y * r : 0.f;
                    const int o = f * A.K * F + k;
                    A.X_abs[xo + o] = mag;
                    A.cos_pd[xo + o] = (tw0 + f < frames_b) ? cy * cx + sy * sx : 0.f;
                }
            }
        }
        __syncthreads();  // buf / sig are reused by the next signal
    }
}

// ------------------------------------------------------------------------------------------------
struct InvArgs {
    const float* spec;
    const int32_t* row_frames;
    const float* syn_window;
    const cpx* twiddle;
    float* out;
    long long num_frames;
    long long out_samples;
    long long out_row_stride;
    long long cut_left;
    int nchunks;
    int layout;
    int halo;   // ceil(L / shift) - 1 frames recomputed per workgroup
    float edge_scale;
    Geo g;
};

template <class PL>
__global__ __launch_bounds__(256, 3) void istft_kernel(const InvArgs A) {
    constexpr int M = PL::M, F = PL::F, FPW = PL::FPW, FPB = PL::FPB, FS = PL::FS, LPF = PL::LPF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx* buf = reinterpret_cast<cpx*>(smem);
    cpx* tws = buf + FPB * FS;
    cpx* tw1t = tws + F + 1;
    float* wsyn = reinterpret_cast<float*>(tw1t + PL::R1 * LPF);

    const int tid = threadIdx.x;
    const int b = blockIdx.x / A.nchunks;
    const int c = blockIdx.x - b * A.nchunks;
    const int nout = FPB - A.halo;                    // frames whose hop segment this block owns
    const long long tstart = (long long)c * nout - A.halo;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.num_frames;
    const int lane = tid & 63, wave = tid >> 6;
    const int fl = lane / LPF, l = lane - fl * LPF;
    const int fb = wave * FPW + fl;

    for (int i = tid; i <= M; i += 256) tws[i] = A.twiddle[i];
    for (int i = tid; i < A.g.L; i += 256) wsyn[i] = A.syn_window[i];
    for (int i = tid; i < PL::R1 * LPF; i += 256) {
        const int k1 = i / LPF, ll = i - k1 * LPF;
        tw1t[i] = tw_full<M>(A.twiddle, (2 * ll * k1) % PL::SIZE);
    }

    // (a) raw one-sided spectra of this wavefront's frames -> LDS (coalesced row reads)
    for (int idx = lane; idx < FPW * F; idx += 64) {
        const int f = idx / F, k = idx - f * F;
        const long long t = tstart + wave * FPW + f;
        cpx X = cpx{0.f, 0.f};
        if (t >= 0 && t < T_b) {
            const long long r = (long long)b * A.num_frames + t;
            if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
                const float2 v = *reinterpret_cast<const float2*>(A.spec + (r * F + k) * 2);
                X = cpx{v.x, v.y};
            } else {
                X = cpx{A.spec[r * 2 * F + k], A.spec[r * 2 * F + F + k]};
            }
            // imaginary parts of DC / Nyquist never reach the output (_stft.py:37-40: sin(0)=sin(pi n)=0)
            if (k == 0 || k == M) X = cpx{X.x * A.edge_scale, 0.f};
        }
        buf[(wave * FPW + f) * FS + k] = X;
    }
    __syncthreads();

    // (b) Z'[k] = (X[k] + conj X[M-k]) + i e^{+i pi k / M} (X[k] - conj X[M-k]),  k = R2*i1 + l
    cpx a[PL::R1];
    {
        const cpx* raw = buf + fb * FS;
#pragma unroll
        for (int i1 = 0; i1 < PL::R1; ++i1) {
            const int k = PL::R2 * i1 + l;
            cpx v = cpx{0.f, 0.f};
            if (l < PL::R2) {
                const cpx x1 = raw[k], x2 = raw[M - k], w = tws[k];
                const float sx = x1.x + x2.x, sy = x1.y - x2.y;   // X[k] + conj X[M-k]
                const float dx = x1.x - x2.x, dy = x1.y + x2.y;   // X[k] - conj X[M-k]
                // i * conj(w) * D, conj(w) = (w.x, -w.y):  conj(w) * D = (w.x dx + w.y dy, w.x dy - w.y dx)
                const float px = w.x * dx + w.y * dy, py = w.x * dy - w.y * dx;
                v = cpx{sx - py, sy + px};
            }
            a[i1] = v;
        }
    }
    __syncthreads();  // every lane has consumed raw before the transposition overwrites it
    fft_to_lds<PL, true>(a, buf + fb * FS, l, tw1t);

    // (c) overlap-add from LDS: buf viewed as floats holds frame samples x[0..size)
    const long long o0 = (long long)c * nout * A.g.shift;
    const int seg = nout * A.g.shift;
    const float* __restrict__ fr = reinterpret_cast<const float*>(buf);
    for (int i = tid; i < seg; i += 256) {
        const long long o = o0 + i;
        const long long n = o - A.cut_left;
        if (n < 0 || n >= A.out_samples) continue;
        long long tlo = (o - A.g.L + A.g.shift) / A.g.shift;   // ceil((o - L + 1) / shift) for o-L+1 > 0
        if (o - A.g.L + 1 <= 0) tlo = 0;
        if (tlo < tstart) tlo = tstart;
        long long thi = o / A.g.shift;
        if (thi > tstart + FPB - 1) thi = tstart + FPB - 1;
        if (thi > T_b - 1) thi = T_b - 1;
        float acc = 0.f;
        for (long long t = tlo; t <= thi; ++t) {
            const int j = (int)(o - t * A.g.shift);
            acc += fr[(int)(t - tstart) * (2 * FS) + j] * wsyn[j];
        }
        A.out[(long long)b * A.out_row_stride + n] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic (any even size) fallbacks: direct DFT, O(F*L) per frame.  Correctness path for sizes
// that are not 64..2048 powers of two (e.g. the doctests' STFT(100..200, ...)).
__device__ __forceinline__ cpx tw_any(const cpx* tw, int size, long long j) {
    const int M = size >> 1;
    int r = (int)(j % size);
    if (r <= M) return tw[r];
    const cpx v = tw[size - r];   // W^(size-r) = conj(W^r)
    return cpx{v.x, -v.y};
}

__global__ __launch_bounds__(256) void stft_generic_kernel(const FwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xw = reinterpret_cast<float*>(smem);
    const int F = A.g.size / 2 + 1, M = A.g.size / 2;
    const long long b = blockIdx.x / A.out_frames;
    const long long t = blockIdx.x - b * A.out_frames;
    const long long n_b = A.row_samples ? (long long)A.row_samples[b] : A.num_samples;
    const long long frames_b = row_frames_of(A.g, n_b);
    const float* xrow = A.x + b * A.x_row_stride;
    for (int j = threadIdx.x; j < A.g.L; j += 256) {
        const long long xi = t * A.g.shift - A.g.pad_left + j;
        xw[j] = (xi >= 0 && xi < n_b) ? xrow[xi] * A.window[j] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < F; k += 256) {
        float re = 0.f, im = 0.f;
        if (t < frames_b) {
            for (int j = 0; j < A.g.L; ++j) {
                const cpx w = tw_any(A.twiddle, A.g.size, (long long)k * j);
                re += xw[j] * w.x;
                im += xw[j] * w.y;
            }
            if (k == 0 || k == M) {
                re *= A.edge_scale;
                im = A.edge_scale == 1.f ? im : 0.f;
            }
        }
        float* orow = A.out + (b * A.out_frames + t) * 2 * F;
        if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
            orow[2 * k] = re;
            orow[2 * k + 1] = im;
        } else {
            orow[k] = re;
            orow[F + k] = im;
        }
    }
}

__global__ __launch_bounds__(256) void istft_generic_kernel(const InvArgs A) {
    const int F = A.g.size / 2 + 1, M = A.g.size / 2;
    const long long per_row = (A.out_samples + 255) / 256;
    const long long b = blockIdx.x / per_row;
    const long long n = (blockIdx.x - b * per_row) * 256 + threadIdx.x;
    if (n >= A.out_samples) return;
    const long long T_b = A.row_frames ? (long long)A.row_frames[b] : A.num_frames;
    const long long o = n + A.cut_left;
    long long tlo = (o - A.g.L + 1 <= 0) ? 0 : (o - A.g.L + A.g.shift) / A.g.shift;
    long long thi = o / A.g.shift;
    if (thi > T_b - 1) thi = T_b - 1;
    float acc = 0.f;
    for (long long t = tlo; t <= thi; ++t) {
        const int j = (int)(o - t * A.g.shift);
        const float* row = A.spec + (b * A.num_frames + t) * 2 * F;
        float v = 0.f;
        for (int k = 0; k <= M; ++k) {
            float re, im;
            if (A.layout == PTMI_LAYOUT_INTERLEAVED) {
                re = row[2 * k];
                im = row[2 * k + 1];
            } else {
                re = row[k];
                im = row[F + k];
            }
            const cpx w = tw_any(A.twiddle, A.g.size, (long long)k * j);  // (cos, -sin)(2 pi k j / size)
            if (k == 0 || k == M) {
                v += A.edge_scale * re * w.x;
            } else {
                v += 2.f * (re * w.x + im * w.y);   // 2 Re(X e^{+i theta}) = 2 (re cos - im sin), w.y = -sin
            }
        }
        acc += v * A.syn_window[j];
    }
    A.out[b * A.out_row_stride + n] = acc;
}

// ------------------------------------------------------------------------------------------------
template <class PL>
static size_t fwd_smem_bytes(const Geo& g) {
    return FwdLds<PL>::bytes(g.shift);
}
template <class PL>
static size_t inv_smem_bytes(const Geo& g) {
    return sizeof(cpx) * ((size_t)PL::FPB * PL::FS + PL::F + 1 + PL::R1 * PL::LPF) +
           sizeof(float) * (g.L + 4);
}

constexpr size_t kMaxSmem = 64 * 1024;  // keep >= 2 workgroups per CU (160 KiB LDS)

template <class PL>
static int launch_fwd(FwdArgs& A, long long batch, bool features, hipStream_t st) {
    const size_t smem = fwd_smem_bytes<PL>(A.g);
    if (smem > kMaxSmem) return PTMI_E_UNSUPPORTED;
    A.nchunks = (int)((A.out_frames + PL::FPB - 1) / PL::FPB);
    const long long blocks = batch * A.nchunks;
    if (blocks <= 0) return PTMI_OK;
    if (blocks > 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
    if (features)
        hipLaunchKernelGGL(pit_features_kernel<PL>, dim3((unsigned)blocks), dim3(256), smem, st, A);
    else
        hipLaunchKernelGGL(stft_fwd_kernel<PL>, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

static int dispatch_fwd(FwdArgs& A, long long batch, bool features, hipStream_t st) {
    if (A.g.L > A.g.size) return PTMI_E_INVALID;
    switch (A.g.size) {
        case 64: return launch_fwd<Plan<4, 8>>(A, batch, features, st);
        case 128: return launch_fwd<Plan<8, 8>>(A, batch, features, st);
        case 256: return launch_fwd<Plan<8, 16>>(A, batch, features, st);
        case 512: return launch_fwd<Plan<16, 16>>(A, batch, features, st);
        case 1024: return launch_fwd<Plan<16, 32>>(A, batch, features, st);
        case 2048: return launch_fwd<Plan<32, 32>>(A, batch, features, st);
        default: return PTMI_E_UNSUPPORTED;
    }
}

template <class PL>
static int launch_inv(InvArgs& A, long long batch, hipStream_t st) {
    const size_t smem = inv_smem_bytes<PL>(A.g);
    if (smem > kMaxSmem) return PTMI_E_UNSUPPORTED;
    A.halo = (A.g.L + A.g.shift - 1) / A.g.shift - 1;
    if (A.halo >= PL::FPB) return PTMI_E_UNSUPPORTED;
    const long long seg = (long long)(PL::FPB - A.halo) * A.g.shift;
    A.nchunks = (int)((A.cut_left + A.out_samples + seg - 1) / seg);
    const long long blocks = batch * A.nchunks;
    if (blocks <= 0) return PTMI_OK;
    if (blocks > 0x7fffffffLL) return PTMI_E_UNSUPPORTED;
    hipLaunchKernelGGL(istft_kernel<PL>, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

static int dispatch_inv(InvArgs& A, long long batch, hipStream_t st) {
    if (A.g.L > A.g.size) return PTMI_E_INVALID;
    switch (A.g.size) {
        case 64: return launch_inv<Plan<4, 8>>(A, batch, st);
        case 128: return launch_inv<Plan<8, 8>>(A, batch, st);
        case 256: return launch_inv<Plan<8, 16>>(A, batch, st);
        case 512: return launch_inv<Plan<16, 16>>(A, batch, st);
        case 1024: return launch_inv<Plan<16, 32>>(A, batch, st);
        case 2048: return launch_inv<Plan<32, 32>>(A, batch, st);
        default: return PTMI_E_UNSUPPORTED;
    }
}

static bool geom_ok(const ptmi_stft_geom* g) {
    return g && g->size >= 2 && (g->size % 2 == 0) && g->shift >= 1 && g->window_length >= 1 &&
           g->pad_left >= 0 && g->pad_right >= 0;
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int64_t ptmi_stft_num_frames(const ptmi_stft_geom* g, int64_t num_samples) {
    if (!geom_ok(g)) return PTMI_E_INVALID;
    return row_frames_of(to_geo(g), num_samples);
}

int64_t ptmi_istft_num_samples(const ptmi_stft_geom* g, int64_t num_frames) {
    if (!geom_ok(g)) return PTMI_E_INVALID;
    // (frames-1)*shift + L minus the fading cut: int(pw) left, ceil(pw) right (_stft.py:257-262)
    return (num_frames - 1) * g->shift + g->window_length - g->pad_left - g->pad_right;
}

int ptmi_stft_forward(const float* x, int64_t batch, int64_t x_row_stride, int64_t num_samples,
                      const int32_t* row_samples, const float* window, const float* twiddle,
                      const ptmi_stft_geom* g, int64_t out_frames, int32_t layout, float edge_scale,
                      float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !x || !window || !twiddle || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || out_frames < 0 || (layout != 0 && layout != 1), PTMI_E_INVALID);
    PTMI_RETURN_IF(g->window_length > g->size, PTMI_E_INVALID);
    PTMI_RETURN_IF(num_samples > 0x7ff00000LL || out_frames > 0x7ff00000LL / (g->size + 2), PTMI_E_UNSUPPORTED);
    if (batch == 0 || out_frames == 0) return PTMI_OK;
    FwdArgs A{};
    A.x = x;
    A.row_samples = row_samples;
    A.window = window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = out;
    A.x_row_stride = x_row_stride;
    A.num_samples = num_samples;
    A.out_frames = out_frames;
    A.layout = layout;
    A.edge_scale = edge_scale;
    A.g = to_geo(g);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = dispatch_fwd(A, batch, false, st);
    if (rc != PTMI_E_UNSUPPORTED) return rc;
    const long long blocks = (long long)batch * out_frames;
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    const size_t smem = sizeof(float) * (size_t)g->window_length;
    PTMI_RETURN_IF(smem > kMaxSmem, PTMI_E_UNSUPPORTED);
    hipLaunchKernelGGL(stft_generic_kernel, dim3((unsigned)blocks), dim3(256), smem, st, A);
    return launch_status();
}

int ptmi_istft_forward(const float* spec, int64_t batch, int64_t num_frames, const int32_t* row_frames,
                       const float* syn_window, const float* twiddle, const ptmi_stft_geom* g,
                       int32_t layout, float edge_scale, int64_t cut_left, int64_t out_samples,
                       int64_t out_row_stride, float* out, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !spec || !syn_window || !twiddle || !out, PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || num_frames < 0 || out_samples < 0 || (layout != 0 && layout != 1),
                   PTMI_E_INVALID);
    PTMI_RETURN_IF(g->window_length > g->size, PTMI_E_INVALID);
    if (batch == 0 || out_samples == 0) return PTMI_OK;
    InvArgs A{};
    A.spec = spec;
    A.row_frames = row_frames;
    A.syn_window = syn_window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = out;
    A.num_frames = num_frames;
    A.out_samples = out_samples;
    A.out_row_stride = out_row_stride;
    A.cut_left = cut_left;
    A.layout = layout;
    A.edge_scale = edge_scale;
    A.g = to_geo(g);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = dispatch_inv(A, batch, st);
    if (rc != PTMI_E_UNSUPPORTED) return rc;
    const long long blocks = (long long)batch * ((out_samples + 255) / 256);
    PTMI_RETURN_IF(blocks > 0x7fffffffLL, PTMI_E_UNSUPPORTED);
    hipLaunchKernelGGL(istft_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, A);
    return launch_status();
}

int ptmi_pit_features(const float* y, const float* s, int64_t batch, int32_t K, int64_t row_stride,
                      int64_t num_samples, const int32_t* row_samples, const float* window,
                      const float* twiddle, const ptmi_stft_geom* g, int64_t out_frames, float* Y_abs,
                      float* X_abs, float* cos_pd, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!geom_ok(g) || !y || !window || !twiddle || !Y_abs, PTMI_E_INVALID);
    PTMI_RETURN_IF(s && (!X_abs || !cos_pd || K < 1), PTMI_E_INVALID);
    PTMI_RETURN_IF(batch < 0 || out_frames < 0, PTMI_E_INVALID);
    PTMI_RETURN_IF(num_samples > 0x7ff00000LL || out_frames > 0x7ff00000LL / (g->size + 2), PTMI_E_UNSUPPORTED);
    if (batch == 0 || out_frames == 0) return PTMI_OK;
    FwdArgs A{};
    A.x = y;
    A.s = s;
    A.K = K;
    A.row_samples = row_samples;
    A.window = window;
    A.twiddle = reinterpret_cast<const cpx*>(twiddle);
    A.out = Y_abs;
    A.X_abs = X_abs;
    A.cos_pd = cos_pd;
    A.x_row_stride = row_stride;
    A.num_samples = num_samples;
    A.out_frames = out_frames;
    A.edge_scale = 1.f;
    A.g = to_geo(g);
    return dispatch_fwd(A, batch, true, static_cast<hipStream_t>(stream));
}

}  // extern "C"
