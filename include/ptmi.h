/*
 * ptmi.h - C ABI of libptmi.so: the MI355X (gfx950) kernels behind the padertorch PIT hot path.
 *
 * Plain pointers and sizes only (no torch types).  Every pointer marked "device" is a HIP device
 * pointer owned by the caller; kernels are enqueued on `stream` (a hipStream_t passed as void*)
 * and borrow the buffers until that work has run.  Every entry point returns 0 on success, a
 * negative PTMI_E_* code for argument errors, or a positive hipError_t from the launch.
 *
 * Each function cites the reference interface (file:line below /root/reference) it replaces.
 */
#ifndef PTMI_H_
#define PTMI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_OK 0
#define PTMI_E_INVALID (-1)     /* bad argument (null pointer, odd size, ...)            */
#define PTMI_E_UNSUPPORTED (-2) /* configuration the kernels do not cover                 */

typedef void* ptmi_stream_t; /* hipStream_t */

const char* ptmi_version(void);
const char* ptmi_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * STFT geometry: the framing of padertorch/ops/_stft.py:131-158 (STFT.__call__).
 *   pad_left / pad_right : fading zeros (L-shift each for 'full'; (L-shift)//2, ceil((L-shift)/2)
 *                          for 'half'; 0 for None/False)                       (_stft.py:137-146)
 *   pad                  : 1 = right-pad to a whole frame (ceil), 0 = cut (floor) (_stft.py:148-154)
 * ------------------------------------------------------------------------------------------- */
typedef struct ptmi_stft_geom {
    int32_t size;          /* FFT size (even)                                   */
    int32_t shift;         /* hop                                               */
    int32_t window_length; /* L <= size                                         */
    int32_t pad_left;
    int32_t pad_right;
    int32_t pad;
} ptmi_stft_geom;

/* Frame / sample bookkeeping (integer, bit-exact).
 * Replaces STFT.samples_to_frames / frames_to_samples (_stft.py:265-307 -> paderbox
 * _samples_to_stft_frames / _stft_frames_to_samples) and the conv1d output length of _stft.py:158. */
int64_t ptmi_stft_num_frames(const ptmi_stft_geom* g, int64_t num_samples);
int64_t ptmi_istft_num_samples(const ptmi_stft_geom* g, int64_t num_frames);

/* Output layouts of ptmi_stft_forward / input layouts of ptmi_istft_forward (_stft.py:162-174). */
#define PTMI_LAYOUT_INTERLEAVED 0 /* [..., frames, F, 2] == complex64 == 'stacked' */
#define PTMI_LAYOUT_CONCAT 1      /* [..., frames, 2F]  (re | im)                  */

/* Forward STFT: framing + window + real FFT in one kernel.
 * Replaces STFT.__call__ (_stft.py:103-174: F.pad, F.pad, kernel.to(x), F.conv1d, rearrange, chunk).
 *   x            device [batch, x_row_stride] float32, row b valid for row_samples[b] (or num_samples)
 *   row_samples  device int32[batch] or NULL (all rows num_samples long)
 *   window       device float32[window_length]
 *   twiddle      device float32[(size/2+1)*2]: (cos, -sin)(2*pi*j/size), j = 0..size/2
 *   out          device float32 [batch, out_frames, F, 2] or [batch, out_frames, 2F]; rows past a
 *                row's own frame count are zero-filled
 *   edge_scale   1.0 for the plain transform; 0.5 when used as the adjoint of ptmi_istft_forward
 *                (scales the DC/Nyquist bins and zeroes their imaginary part)
 */
int ptmi_stft_forward(const float* x, int64_t batch, int64_t x_row_stride, int64_t num_samples,
                      const int32_t* row_samples, const float* window, const float* twiddle,
                      const ptmi_stft_geom* g, int64_t out_frames, int32_t layout, float edge_scale,
                      float* out, ptmi_stream_t stream);

/* Inverse STFT: hermitian inverse real FFT + synthesis window + overlap-add + fading cut.
 * Replaces STFT.inverse (_stft.py:176-263: two conv_transpose1d over the hermitian-extended
 * spectrum, sum, cut).
 *   spec         device float32, layout as above, [batch, num_frames, ...]
 *   row_frames   device int32[batch] or NULL (all rows num_frames)
 *   syn_window   device float32[window_length]: biorthogonal window / size   (_stft.py:27-28)
 *   out          device float32 [batch, out_row_stride]; out_samples per row are written
 *   cut_left     samples dropped in front (int(pad_width), _stft.py:257-262)
 *   edge_scale   1.0 for the plain inverse; 2.0 when used as the adjoint of ptmi_stft_forward
 */
int ptmi_istft_forward(const float* spec, int64_t batch, int64_t num_frames, const int32_t* row_frames,
                       const float* syn_window, const float* twiddle, const ptmi_stft_geom* g,
                       int32_t layout, float edge_scale, int64_t cut_left, int64_t out_samples,
                       int64_t out_row_stride, float* out, ptmi_stream_t stream);

/* Fused PIT feature front-end: STFT of the mixture and of K sources + |.| + cos(phase difference).
 * Replaces pre_batch_transform (padertorch/contrib/examples/source_separation/pit/data.py:49-77).
 *   y            device [batch, row_stride]           mixture waveforms
 *   s            device [batch, K, row_stride]        source waveforms (may be NULL: only Y_abs)
 *   row_samples  device int32[batch] or NULL
 *   Y_abs        device [batch, out_frames, F]
 *   X_abs, cos_pd device [batch, out_frames, K, F]    (NULL when s is NULL)
 * Any even g->size (paderbox.stft takes any): powers of two in 64..2048 run the FFT kernels, every other size the direct-DFT kernel.
 */
int ptmi_pit_features(const float* y, const float* s, int64_t batch, int32_t K, int64_t row_stride,
                      int64_t num_samples, const int32_t* row_samples, const float* window,
                      const float* twiddle, const ptmi_stft_geom* g, int64_t out_frames, float* Y_abs,
                      float* X_abs, float* cos_pd, ptmi_stream_t stream);
/* The same launch, also writing the model's first-layer input (pit/model.py:91-94: pack_sequence(Y_abs) -> log1p):
 *   log1p_packed    device [rows, F] fp32, rows = sum of the examples' frame counts: log1p(Y_abs) as PackedSequence data, the row
 *                   of frame t of example b = packed_offsets[t] + b (examples sorted by descending length), or t * batch + b
 *                   when packed_offsets is NULL (all examples have out_frames frames)
 *   log1p_planes    NULL, or the same matrix as fp16 (hi, lo) planes of 2^9 log1p(Y_abs) in the layout of ptmi_pack_planes_n
 *                   (ptmi_planes_elems(rows, F) values, 16-byte aligned, ZEROED by the caller: padding is not written): operand A
 *                   of the first input projection on ptmi_gemm_planes with an operand-scale word of 16.0f (scale 2^13 / 16).
 *                   The fixed scale holds for every finite input: log1p(FLT_MAX) 2^9 < 65504. */
int ptmi_pit_features_packed(const float* y, const float* s, int64_t batch, int32_t K, int64_t row_stride,
                             int64_t num_samples, const int32_t* row_samples, const float* window,
                             const float* twiddle, const ptmi_stft_geom* g, int64_t out_frames, float* Y_abs,
                             float* X_abs, float* cos_pd, float* log1p_packed, uint16_t* log1p_planes,
                             const int64_t* packed_offsets, ptmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Permutation-invariant training loss.
 * Replaces pit_loss (padertorch/ops/losses/source_separation.py:34-124) with loss_fn = mse_loss
 * and the review loop of pit/model.py:117-140.
 * ------------------------------------------------------------------------------------------- */

/* Pairwise sums of squared errors for `batch` ragged examples stored in padded buffers.
 *   est      device, element (b, t, i, f) at b*strides[0] + t*strides[1] + i*F + f: estimates
 *            (or masks when obs != NULL: est = mask * obs); batch- or time-major padded storage
 *   obs      device or NULL, element (b, t, f) at b*strides[2] + t*strides[3] + f
 *   tgt      device, element (b, t, j, f) at b*strides[4] + t*strides[5] + j*F + f
 *   tgt_scale device like tgt or NULL; second target = tgt * tgt_scale (cos phase difference)
 *   strides  HOST int64[6] (element strides, see above)
 *   row_frames device int32[batch] or NULL (all t_len)
 *   sse      device float64 [batch, nvar, K, K]: sse[b, v, i, j] = sum_{t,f} (est_i - tgt^v_j)^2
 *            nvar = 1 (tgt) or 2 (tgt, tgt*tgt_scale)
 *   workspace device float64 [ptmi_pit_workspace_elems(...)]
 */
int64_t ptmi_pit_workspace_elems(int64_t batch, int64_t t_len, int32_t K, int32_t F);
int ptmi_pit_pairwise_sse(const float* est, const float* obs, const float* tgt, const float* tgt_scale,
                          int64_t batch, int64_t t_len, const int64_t* strides, int32_t K, int32_t F,
                          const int32_t* row_frames, double* workspace, double* sse,
                          ptmi_stream_t stream);

/* Permutation search on the pairwise matrix + batch mean (first minimum wins, torch.min).
 *   loss     device float32 [nvar]         mean_b min_perm (1/(T_b K F)) sum_j sse[b,v,perm[j],j]
 *   perm     device int32 [batch, nvar, K] estimate index per target (pit_loss's convention)
 *   ex_loss  device float32 [batch, nvar]  per-example losses (may be NULL)
 */
int ptmi_pit_assign(const double* sse, int64_t batch, int32_t nvar, int32_t K, int32_t F,
                    int64_t t_len, const int32_t* row_frames, float* loss, int32_t* perm,
                    float* ex_loss, ptmi_stream_t stream);

/* Gradient wrt est (or wrt the mask when obs != NULL):
 *   grad[b,t,i,f] = sum_v gscale[v] * 2/(B T_b K F) * (est_i - tgt^v_{j: perm[b,v,j]=i}) [* obs]
 *   gscale   device float32 [nvar] (upstream gradients of the two batch-mean losses)
 *   grad     device, same addressing as est (strides[0], strides[1]); frames t >= T_b get 0
 */
int ptmi_pit_backward(const float* est, const float* obs, const float* tgt, const float* tgt_scale,
                      const int32_t* perm, const float* gscale, int64_t batch, int64_t t_len,
                      const int64_t* strides, int32_t K, int32_t F, int32_t nvar, const int32_t* row_frames,
                      float* grad, ptmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Deep-clustering loss.  Replaces deep_clustering_loss (padertorch/ops/losses/source_separation.py:
 * 13-31) and the per-example review loop + re-layout of contrib/tcl/dc.py:73-84.
 *   x  embeddings, element (b, t, e, f) at b*strides[0] + t*strides[1] + e*strides[2] + f*strides[3]
 *   t  targets,    element (b, t, k, f) at b*strides[4] + t*strides[5] + k*strides[6] + f*strides[7]
 *      rows of example b: n = (t, f), t < T_b (row_frames[b] or T), f < F;  N_b = T_b * F
 *      (plain (N, E) / (N, K) matrices: T = N, F = 1, strides = {0, E, 1, 0, 0, K, 1, 0})
 *   strides HOST int64[8];  E + K <= 32
 *   gram    device float64 [batch, 32, 32]: V'V with V = [x | t] (zero padded)
 *   ex_loss device float32 [batch]: (|X'X|^2 - 2|X'T|^2 + |T'T|^2) / N_b^2;  loss [1]: batch mean
 *   workspace device float32 [ptmi_dc_workspace_elems(batch, T, F)]
 * ------------------------------------------------------------------------------------------- */
int64_t ptmi_dc_workspace_elems(int64_t batch, int64_t T, int32_t F);
int ptmi_dc_loss_forward(const float* x, const float* t, int64_t batch, int64_t T, const int64_t* strides,
                         int32_t E, int32_t K, int32_t F, const int32_t* row_frames, float* workspace,
                         double* gram, float* ex_loss, float* loss, ptmi_stream_t stream);
/* dx = gscale[0] / batch * 4 / N_b^2 * (x (X'X) - t (T'X)), same addressing as x. */
int ptmi_dc_loss_backward(const float* x, const float* t, const double* gram, const float* gscale,
                          int64_t batch, int64_t T, const int64_t* strides, int32_t E, int32_t K, int32_t F,
                          const int32_t* row_frames, float* dx, ptmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Packed-sequence (B)LSTM recurrence (time loop of torch.nn.LSTM on a PackedSequence:
 * pit/model.py:60-66,97, contrib/tcl/dc.py:32-34,61; gate order i,f,g,o; zero initial state).
 * rows = packed time-major rows, row(t, b) = offsets[t] + b, b < batch_sizes[t] (descending).
 * One launch per timestep covers both directions (direction 1 walks time backwards).
 * ------------------------------------------------------------------------------------------- */

/* Forward through time.
 *   gates      device [rows, ndir, 4, H]  in: x W_ih^T + b_ih + b_hh; out: activated i,f,g,o (in place)
 *   hy, c      device [rows, ndir, H]     out: hidden / cell state of every step
 *   c0         device [ndir, max_batch, H] initial cell state of every sequence, or NULL (zero).  The
 *              initial HIDDEN state enters through `gates`: the caller adds h0 W_hh^T to the rows of
 *              each sequence's first processed step (torch.nn.LSTM(x, (h0, c0)) semantics).
 *   w_hh_pad   device [ndir, 4H, KP]      recurrent weights, K zero-padded to KP = roundup(H, 16)
 *   batch_sizes HOST int32 [T], offsets HOST int64 [T] (PackedSequence bookkeeping; per-step row
 *              ranges travel as kernel arguments)
 *   H % 4 == 0 is required (16-byte aligned operand rows), else PTMI_E_UNSUPPORTED.
 */
int ptmi_lstm_forward(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad, const int32_t* batch_sizes,
                      const int64_t* offsets, int32_t T, int32_t max_batch, int32_t H, int32_t KP,
                      int32_t ndir, ptmi_stream_t stream);

/* Backward through time.
 *   gates, c   saved by ptmi_lstm_forward;  dhy device [rows, ndir, H] gradient wrt hy
 *   w_hh_t     device [ndir, H, 4H]  (transposed recurrent weights)
 *   dgates     device [rows, ndir, 4, H]  out: gradient wrt the gate pre-activations
 *   dc_state   device [max_batch, ndir, H] scratch (need not be initialised)
 * dW_ih, dW_hh, db and dx follow from dgates by dense GEMMs on the caller's side.
 */
int ptmi_lstm_backward(const float* gates, const float* c, const float* c0, const float* dhy, const float* w_hh_t, float* dgates,
                       float* dc_state, const int32_t* batch_sizes, const int64_t* offsets, int32_t T,
                       int32_t max_batch, int32_t H, int32_t ndir, ptmi_stream_t stream);

/* Persistent forward recurrence: ONE launch for all T steps; W_hh stays in registers and the steps
 * are chained by write-through stores + per-step arrival counters (see csrc/lstm.hip).  Same
 * results as ptmi_lstm_forward.  batch_sizes_dev / offsets_dev are DEVICE copies here; flags is a
 * device scratch of ptmi_lstm_scratch_elems(T, ndir, max_batch, H, 0) uint32: the tile-major copy of hy the
 * workgroups hand to each other, ptmi_lstm_flags_elems(T, ndir, max_batch) arrival counters (zeroed by
 * the call) and, as the LAST 8 words, the error words (non-zero after the call = a bounded spin ran
 * out).  KP must be H rounded up to 16.  Returns PTMI_E_UNSUPPORTED when the configuration cannot be
 * kept resident (caller falls back to ptmi_lstm_forward).
 * The default kernels (csrc/lstm_split.hip; ptmi_lstm_split_enabled() != 0) evaluate h W_hh^T as three fp16 MFMA
 * products of (hi, lo) operand halves with fp32 accumulation (fp32-equivalent result); w_hh_amax = device word
 * from ptmi_absmax over w_hh_pad (NULL: |W_hh| within fp16's range as it is).  The backward kernel splits into
 * bf16 halves (no scale) and leaves max |dgates| (float bits) in the word right behind the bias gradient.
 * PTMI_LSTM_F32=1 in the environment selects the exact-fp32 MFMA kernels of round 1. */
int64_t ptmi_lstm_flags_elems(int32_t T, int32_t ndir, int32_t max_batch);
/* ptmi_lstm_weight_prep (csrc/lstm_prep.hip): the operand forms of ONE (bi)directional LSTM layer's parameters
 * (torch.nn.LSTM layout, padertorch/contrib/examples/source_separation/pit/model.py:60-66) in one launch:
 *   w_ih / w_hh / b_ih / b_hh  HOST arrays of ndir device pointers: [4H, I], [4H, H], [4H], [4H] per direction
 *   w_ih_cat [ndir * 4H, Ipad]  stacked input weights, columns I .. Ipad-1 zero (Ipad >= I, e.g. I rounded up to 4)
 *   bias     [ndir * 4H]        b_ih + b_hh
 *   w_pad    [ndir, 4H, KP]     recurrent weights, columns H .. KP-1 zero (what ptmi_lstm_forward* take as w_hh_pad)
 *   w_t      [ndir, H, 4H]      transposed recurrent weights (what ptmi_lstm_backward* take)
 *   amax     [2] uint32         float bits of max |w_ih|, max |w_hh| over both directions (the words ptmi_gemm_split /
 *                               ptmi_lstm_forward_persistent take as operand scales) */
int ptmi_lstm_weight_prep(const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                          const float* const* b_hh, int32_t ndir, int32_t H, int32_t I, float* w_ih_cat, int32_t Ipad,
                          float* bias, float* w_pad, int32_t KP, float* w_t, uint32_t* amax, ptmi_stream_t stream);
/* ptmi_lstm_set_error_sink: `word` (device uint32, zeroed by the caller, alive as long as launches may run; NULL to
 * unset) of the CURRENT device is incremented by every persistent launch whose bounded spin runs out (next to that
 * launch's own error word in its scratch): ONE word for the host to watch instead of one per call. */
int ptmi_lstm_set_error_sink(uint32_t* word);
int ptmi_lstm_split_enabled(void);
/* ptmi_lstm_handoff_cols: columns per direction (H resp. 4H rounded up to 32) of the 16-bit hand-off planes the persistent
 * split kernels leave at the START of their scratch - forward (backward = 0): fp16 (hi, lo) halves of 2^10 h, backward: bf16
 * halves of dgates -, as [T][16-row tile][direction][cols / 32][hi | lo][64 chunks of 8 values] in the fragment order of
 * ptmi_gemm_planes: for a batch of equal-length sequences whose size is a multiple of 16 this IS the A operand
 * (rows = packed rows, k = direction-major columns) of the dense GEMM that follows.  0: these kernels do not run for this H
 * (PTMI_LSTM_F32, or a layer too wide for their register budget). */
int32_t ptmi_lstm_handoff_cols(int32_t H, int32_t backward);
int ptmi_lstm_forward_persistent(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad,
                                 const uint32_t* w_hh_amax, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                 uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows, int32_t H, int32_t KP,
                                 int32_t ndir, int32_t prefilled, uint32_t* backward_scratch, ptmi_stream_t stream);

/* Persistent backward-through-time (mirror of ptmi_lstm_forward_persistent; same results as
 * ptmi_lstm_backward; no dc_state scratch: the cell-state gradient stays in registers).  `flags` is a device
 * scratch of ptmi_lstm_scratch_elems(T, ndir, max_batch, H, 1) uint32: a tile-major copy of dgates that the
 * workgroups hand to each other (16 x 16 tiles, one contiguous KB per load), then [ndir][4H] floats that
 * receive the BIAS GRADIENT (sum of dgates over all rows, out), then the ptmi_lstm_flags_elems arrival
 * counters (both zeroed by the call), then the 8 error words (still the LAST 8 words). */
int64_t ptmi_lstm_scratch_elems(int32_t T, int32_t ndir, int32_t max_batch, int32_t H, int32_t backward);
int ptmi_lstm_backward_persistent(const float* gates, const float* c, const float* c0, const float* dhy, const float* w_hh_t,
                                  float* dgates, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                  uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows, int32_t H,
                                  int32_t ndir, int32_t prefilled, ptmi_stream_t stream);
/* The same recurrence cut in time: processes the steps [s_begin, s_end) of the T processing steps (step s handles
 * time index T-1-s in direction 0, s in direction 1).  Ranges must be launched in order on one stream with the same
 * scratch; the first (s_begin == 0) zeroes it, dc_carry (device fp32 [ndir, max_batch, H]) takes the cell-state
 * gradient across a cut.  After the range [0, s) the gate gradients of direction 0 are complete for time indices
 * >= T - s and of direction 1 for time indices < s: their weight-gradient GEMMs can run under the next range.
 * Split kernels only (PTMI_E_UNSUPPORTED otherwise). */
int ptmi_lstm_backward_persistent_range(const float* gates, const float* c, const float* c0, const float* dhy,
                                        const float* w_hh_t, float* dgates, const int32_t* batch_sizes_dev,
                                        const int64_t* offsets_dev, uint32_t* flags, float* dc_carry, int32_t T,
                                        int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin,
                                        int32_t s_end, int32_t prefilled, ptmi_stream_t stream);
/* The same launch with the gate gradients ALSO (or only) leaving the kernel as the operand of the weight-gradient GEMMs
 * dW_ih = dgates^T x, dW_hh = dgates^T h_prev (torch.nn.LSTM backward inside pit/model.py:60-66,97): bf16 (hi, lo) planes of
 * dgates^T per direction, [ndir][4H / 16 column tiles][ceil(rows / 32) k blocks][hi | lo][64 chunks of 8 values] = exactly what
 * ptmi_pack_planes_t_bf16 makes of one direction's [rows, 4H] block, i.e. operand A of ptmi_gemm_planes_bf16(m = 4H, k = rows)
 * at byte offset direction * ptmi_planes_elems(4H, rows) * 2.  No row-major fp32 store, no transposing pack pass.
 *   dgates_t   ndir * ptmi_planes_elems(4H, range_rows) uint16 (16-byte aligned) or NULL, range_rows = (s_end - s_begin) *
 *              max_batch: the rows of THIS launch's step range (all rows for [0, T)) - for the forward direction the time
 *              indices T - s_end .. T - s_begin - 1, for the reverse direction s_begin .. s_end - 1 -; the call writes every value
 *   dgates     may be NULL when dgates_t is given (the hand-off copy in `flags` still serves dx = dgates W_ih)
 * Only for batches of equal-length sequences whose size is a multiple of 16 on the split kernels:
 * ptmi_lstm_backward_planes_ok(...) != 0; PTMI_E_UNSUPPORTED otherwise. */
/* The whole backward recurrence with gradients through BOTH ends of the state (torch.nn.LSTM returns (h_n, c_n) with their graph,
 * padertorch/modules/recurrent.py:42 carries them): dc_n [ndir][max_batch][H] (or NULL) = gradient w.r.t. the final cell state, added
 * to the cell-state gradient at every sequence's last step; dc_0 [ndir][max_batch][H] (or NULL) receives the gradient w.r.t. the
 * initial cell state.  (The final / initial HIDDEN states' gradients travel through dhy resp. dgates W_hh on the caller's side.)
 * Split kernels only. */
int ptmi_lstm_backward_persistent_states(const float* gates, const float* c, const float* c0, const float* dhy, const float* dc_n,
                                         const float* w_hh_t, float* dgates, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, uint32_t* flags, float* dc_0, int32_t T, int32_t max_batch,
                                         int64_t rows, int32_t H, int32_t ndir, int32_t prefilled, ptmi_stream_t stream);
/* Row-slot batches: several sequences lie END TO END in one row slot, so that every one of the (at most 64) row slots works in
 * (nearly) every time step - a recurrence costs its number of steps, whatever the number of rows up to 32 (64) per step, so a ragged
 * batch packed this way takes total frames / slots steps instead of the longest sequence's.  Layout: uniform, row(t, slot) = t *
 * max_batch + slot, every such row exists in all buffers; step_masks [T][3] uint64 (device): rows alive at time index t, rows whose
 * sequence STARTS at t, rows whose sequence ENDS at t (bit b = slot b).  A sequence start resets (h, c) to zero in the forward
 * direction, a sequence end in the reverse direction; idle rows are neither computed nor handed on (hy / c of idle rows are not
 * written: hand in zeroed buffers; their gate gradients come out as zeros, and their rows of the hand-off planes as zeros too, so
 * that for max_batch % 16 == 0 the planes are the GEMM operands they are for equal-length batches: ptmi_lstm_handoff_cols; the
 * backward call takes dgates_t like ptmi_lstm_backward_persistent_planes, dgates may then be NULL).  Same results per sequence as one sequence per row
 * (torch.nn.LSTM on a PackedSequence, pit/model.py:60-66,97).  Data-as-flag split kernels only (PTMI_E_UNSUPPORTED otherwise); no
 * initial states. */
int ptmi_lstm_forward_persistent_slots(float* gates, float* hy, float* c, const float* c0, const float* w_hh_pad,
                                       const uint32_t* w_hh_amax, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                       const uint64_t* step_masks, uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows,
                                       int32_t H, int32_t KP, int32_t ndir, int32_t prefilled, uint32_t* backward_scratch,
                                       ptmi_stream_t stream);
int ptmi_lstm_backward_persistent_slots(const float* gates, const float* c, const float* dhy, const float* w_hh_t, float* dgates,
                                        uint16_t* dgates_t, const int32_t* batch_sizes_dev, const int64_t* offsets_dev,
                                        const uint64_t* step_masks, uint32_t* flags, int32_t T, int32_t max_batch, int64_t rows,
                                        int32_t H, int32_t ndir, int32_t prefilled, ptmi_stream_t stream);
int32_t ptmi_lstm_backward_planes_ok(int32_t T, int32_t ndir, int32_t max_batch, int64_t rows, int32_t H);
int ptmi_lstm_backward_persistent_planes(const float* gates, const float* c, const float* c0, const float* dhy,
                                         const float* w_hh_t, float* dgates, uint16_t* dgates_t, const int32_t* batch_sizes_dev,
                                         const int64_t* offsets_dev, uint32_t* flags, float* dc_carry, int32_t T,
                                         int32_t max_batch, int64_t rows, int32_t H, int32_t ndir, int32_t s_begin, int32_t s_end,
                                         int32_t prefilled, ptmi_stream_t stream);
/* Data-as-flag hand-off (the split kernels' protocol wherever their tile shapes allow it): the planes at the
 * start of the scratch start out as 0xFFFF in every 16-bit value - a pattern no conversion to fp16 / bf16 produces -, producers
 * only store, consumers re-request an operand tile until none of the values they are going to use is the pattern.  The
 * persistent calls fill the planes themselves (prefilled = 0) unless the caller has done it with ptmi_lstm_scratch_prefill
 * on any stream it orders before the launch (prefilled = 1: the fill then runs next to earlier work instead of in front of
 * the recurrence).  ptmi_lstm_scratch_prefill returns 1 when it has filled, 0 when the launch for this configuration does
 * not use the pattern (nothing done; pass prefilled = 0), < 0 on error. */
int ptmi_lstm_scratch_prefill(uint32_t* scratch, int32_t T, int32_t ndir, int32_t max_batch, int32_t H, int32_t backward,
                              ptmi_stream_t stream);
/* backward_scratch of ptmi_lstm_forward_persistent (NULL: none): the scratch the caller will hand to this layer's
 * ptmi_lstm_backward_persistent.  When ptmi_lstm_forward_fills(T, ndir, max_batch, H) != 0 the forward launch writes the
 * pattern into its planes itself - an otherwise idle wavefront of every workgroup, a slice per time step, i.e. for free -
 * and the caller passes the returned value as `prefilled` to the backward launch: 2 = the planes AND the words behind them
 * (bias sums, maximum word, arrival slots, error words: zeroed) are ready, the backward call enqueues nothing in front of its
 * recurrence kernel (round 6; 1, until then: the planes only).  Otherwise the pointer is ignored. */
int ptmi_lstm_forward_fills(int32_t T, int32_t ndir, int32_t max_batch, int32_t H);

/* ---- (log-)mel features ----------------------------------------------------------------------------
 * Replaces MelTransform.forward (padertorch/contrib/je/modules/features.py:297-330: spectrogram @
 * fbanks, log(x + eps)) and, fused with the STFT, the extractor front-end features.py:171-176
 * (power of the stacked STFT -> MelTransform) without materialising the spectrum.
 * The [F, M] filterbank is passed band-compressed in 16-byte aligned groups of 8 bins: filter m covers
 * bins 4 mel_lo[m] .. 4 mel_lo[m] + 8 mel_cnt[m] - 1 with the (zero-padded) weights
 * mel_w[4 mel_off[m] ...] (device arrays; mel_nnz = number of floats in mel_w, a multiple of 8).
 * ptmi_stft_logmel: x [batch, num_samples] -> out [batch, out_frames, mel_M] = f(|STFT|^power),
 * power in {1, 2}; sizes 64..2048 (powers of two).  ptmi_mel_apply: spec [N, F] -> out [N, mel_M]. */
int ptmi_stft_logmel(const float* x, int64_t batch, int64_t x_row_stride, int64_t num_samples,
                     const int32_t* row_samples, const float* window, const float* twiddle,
                     const ptmi_stft_geom* g, int64_t out_frames, const int32_t* mel_lo, const int32_t* mel_cnt,
                     const int32_t* mel_off, const float* mel_w, int32_t mel_M, int32_t mel_nnz, int32_t power,
                     int32_t log_, float eps, float* out, ptmi_stream_t stream);
int ptmi_mel_apply(const float* spec, int64_t N, int32_t F, const int32_t* mel_lo, const int32_t* mel_cnt,
                   const int32_t* mel_off, const float* mel_w, int32_t mel_M, int32_t mel_nnz, int32_t log_,
                   float eps, float* out, ptmi_stream_t stream);

/* ---- Masked normalisation --------------------------------------------------------------------------
 * Replaces padertorch/modules/normalization.py: normalize / _Normalize.forward (:345-372, statistics by
 * mask_and_compute_stats :497-512), the hand-written backward (:374-411) and the running-statistics
 * path Normalization._running_norm (:233-246).
 * The tensor is contiguous, rank <= 5; per axis: its size, its stride in the statistics-group index
 * (0 for an axis the statistics run over) and in the gamma / beta index (0 where those broadcast);
 * batch_dim / seq_dim (or -1) locate the mask t < lengths[b].
 *   ptmi_norm_reduce      mode 0: (sum m x, sum m x^2, sum m) per statistics group;
 *                         mode 1: (sum ghat, sum ghat xc, sum xc), ghat = gy gamma, xc = x - mean (shift);
 *                         mode 2: (sum gy xhat, sum gy, 0) per gamma / beta element (their gradients).
 *                         out [groups, 3] float64; workspace: ptmi_norm_workspace_elems(geom, mode == 2).
 *   ptmi_norm_elementwise backward = 0: y = m ((x - mean) rstd gamma + beta);
 *                         backward = 1: dx = m (gy gamma rstd + c1[g] (x - mean) + c0[g]).
 * mean / rstd / c0 / c1 are float32 per statistics group, gamma / beta float32 per independent index
 * (NULL = absent). */
typedef struct ptmi_norm_geom {
    int32_t rank;
    int64_t size[5];
    int64_t stat_group_stride[5];
    int64_t indep_stride[5];
    int32_t batch_dim, seq_dim;
} ptmi_norm_geom;
int64_t ptmi_norm_workspace_elems(const ptmi_norm_geom* geom, int32_t which);
int ptmi_norm_reduce(int32_t mode, const float* x, const float* gy, const int32_t* lengths, const float* mean,
                     const float* rstd, const float* gamma, const ptmi_norm_geom* geom, int32_t shift,
                     double* workspace, double* out, ptmi_stream_t stream);
int ptmi_norm_elementwise(int32_t backward, const float* x, const float* gy, const int32_t* lengths,
                          const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* c0, const float* c1, const ptmi_norm_geom* geom, int32_t shift,
                          int32_t scale, float* out, ptmi_stream_t stream);

/* ---- Unit-norm embeddings ------------------------------------------------------------------------
 * Replaces torch.nn.functional.normalize(h, dim=-2) of padertorch/contrib/tcl/dc.py:70 (and its autograd
 * backward) on the [N, E, F] embedding (F contiguous; E <= 32: register-tiled, one HBM pass; wider: the E values read twice):
 *   forward : y = x / max(||x[n, :, f]||_2, eps);  inv_norm [N, F] = 1 / max(norm, eps) is kept for
 *   backward: dx = inv_norm (gy - y <gy, y>_E)   (inv_norm gy where the norm was clamped to eps).
 * One HBM pass each. */
int ptmi_unit_norm_forward(const float* x, float* y, float* inv_norm, int64_t N, int32_t E, int32_t F, float eps,
                           ptmi_stream_t stream);
int ptmi_unit_norm_backward(const float* gy, const float* y, const float* inv_norm, float* dx, int64_t N, int32_t E,
                            int32_t F, float eps, ptmi_stream_t stream);

/* ---- Time-domain regression losses under PIT ------------------------------------------------------
 * Replaces padertorch/ops/losses/regression.py:47-378 (mse_loss, log_mse_loss, sdr_loss, si_sdr_loss,
 * log1p_mse_loss, source_aggregated_sdr_loss) evaluated per permutation by pit_loss
 * (ops/losses/source_separation.py:110-119) and the TasNet loss loop
 * (contrib/examples/source_separation/tasnet/model.py:154-176).
 *
 * ptmi_td_pair_stats: one streaming pass over est / tgt [batch, K, T] (time contiguous; strides[4] =
 * {est_b, est_k, tgt_b, tgt_k} in elements, HOST array; lengths[b] <= T valid samples or NULL) ->
 * stats [batch, K*K + 4K] float64 per example:  Set[i][j] = sum e_i t_j | See[i] | Stt[j] | Se[i] | St[j].
 * workspace: ptmi_td_workspace_elems(batch, K, T) float64.  K <= 8, batch <= 65535.
 *
 * ptmi_td_lincomb: out[b,i,t] = a[b,i] x[b,i,t] + sum_j bmat[b,i,j] y[b,j,t] + c[b,i] for t < lengths[b],
 * 0 beyond (the backward pass: d loss / d estimate with x = est, y = tgt; d / d target with the roles
 * swapped).  strides[6] = {x_b, x_k, y_b, y_k, out_b, out_k}; coefficients are device float32. */
int64_t ptmi_td_stats_elems(int32_t K);
int64_t ptmi_td_workspace_elems(int64_t batch, int32_t K, int64_t T);
int ptmi_td_pair_stats(const float* est, const float* tgt, const int32_t* lengths, int64_t batch, int32_t K,
                       int64_t T, const int64_t* strides, double* workspace, double* stats,
                       ptmi_stream_t stream);
int ptmi_td_lincomb(const float* x, const float* y, const int32_t* lengths, const float* coef_a,
                    const float* coef_b, const float* coef_c, int64_t batch, int32_t K, int64_t T,
                    const int64_t* strides, float* out, ptmi_stream_t stream);

/* ---- Dense layers: fp32 GEMM on the 16-bit matrix cores (split operands) --------------------------
 * Replaces the library GEMMs behind torch.nn.LSTM's input projections and torch.nn.Linear in
 * padertorch/contrib/examples/source_separation/pit/model.py:60-66,97-104 and contrib/tcl/dc.py:32-40,61-66
 * (forward, input gradients, weight gradients): every fp32 operand value v is used as v s = hi + lo (16-bit halves, s a power
 * of two per operand tensor) and every product as hi hi + hi lo + lo hi with fp32 accumulation - as close to the exact result as
 * an fp32 MFMA chain at several times its rate (fp32 MFMA runs at 1/16 of the 16-bit rate on gfx950).
 *
 * ptmi_absmax: out_bits[0] = float bits of max |x[r, c]| over a [rows, cols] fp32 matrix with row stride ld
 * (device; zeroed and written on `stream`).  The pack passes derive the operand scale 2^(13 - exponent) from it. */
int ptmi_absmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* out_bits, ptmi_stream_t stream);
/* The same reduction INTO a word that already holds float bits of a non-negative value (0: a pre-zeroed word): the word ends as the
 * maximum of both - a running maximum over several matrices, or ptmi_absmax without its zeroing launch when the caller hands out words
 * of a buffer it has zeroed once. */
int ptmi_absmax_accumulate(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* inout_bits, ptmi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The GEMM itself runs on operands already split into fp16 (hi, lo) planes in MFMA-fragment order (csrc/gemm_planes.hip); a
 * streaming pass per operand makes them from either storage order (weights: once per optimizer step; activations that a
 * recurrence or the feature kernel has already left as planes: not at all).
 *
 * ptmi_pack_planes_t: source x[k][c] fp32, k_rows x cols, row stride ld  ->  planes of the operand whose rows are the
 *   columns c and whose reduction axis is k: [ceil(cols / 16)][ceil(k_rows / 32)][hi | lo][64 chunks of 8 fp16]
 *   (ptmi_planes_elems(cols, k_rows) fp16 values, 16-byte aligned), every value scaled by 2^(13 - exponent(*amax))
 *   (amax: device word from ptmi_absmax or the backward recurrence; NULL: scale 1), zero past the matrix.
 * ptmi_pack_planes_n: the same planes from a source x[r][k] whose reduction axis is contiguous (rows x k, row stride ld):
 *   activations and weights of the forward form x W^T.
 * ptmi_gemm_planes:  C[m, n] (+)= sum_k A[m, k] B[n, k] / (scale_a scale_b) + bias[n]  with A, B as planes of m x k and n x k
 *   operands (same amax words as at packing), fp32 accumulation; products = 3: hi hi + hi lo + lo hi (fp32-equivalent), 1: the
 *   hi planes only (plain 16-bit operands: the reduced-precision mode of BASELINE configs[1]);
 *   split_k > 1: AT MOST that many k ranges (what the workspace of ptmi_gemm_planes_workspace_elems floats holds); how many are used,
 *   and on which workgroup tile (persistent big-tile kernel: work item = (k range, tile)), is a cost model's choice that depends on
 *   the shape alone; the ranges' partial products are summed in range order by a second kernel (reproducible, no atomics).
 *   split_k < 0: exactly -split_k ranges (-1: no split) on the 128 x 128 kernel, whose workgroups fit on a CU next to a workgroup of the
 *   persistent recurrence kernels (callers whose GEMM runs beside a recurrence). */
int64_t ptmi_planes_elems(int64_t rows, int64_t k);
int ptmi_pack_planes_t(const float* x, int64_t k_rows, int64_t cols, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream);
int ptmi_pack_planes_n(const float* x, int64_t rows, int64_t k, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream);
int64_t ptmi_gemm_planes_workspace_elems(int32_t m, int32_t n, int32_t k, int32_t split_k);
int ptmi_gemm_planes(const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b, const float* bias, float* c,
                     int64_t ldc, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t split_k, int32_t products,
                     float* workspace, ptmi_stream_t stream);
/* A dense layer with the ReLU behind it in one call (torch.nn.Linear + torch.nn.ReLU, pit/model.py:98-104, tcl/dc.py:38-40):
 * c = max(a b^T + bias, 0) as ptmi_gemm_planes computes the product (no accumulation), and the float bits of max c into *amax_out (may be
 * NULL; zeroed by the call unless amax_zeroed says the caller has done that) - the operand scale ptmi_pack_planes_n takes when c is the next layer's input, so that neither the activation
 * nor ptmi_absmax is a pass of its own.  ptmi_relu_backward_absmax is the matching backward step: out = g where y > 0 (y = the ReLU's
 * output c), else 0, and the float bits of max |out| into *amax_out - what torch's threshold_backward and ptmi_absmax did in two passes. */
int ptmi_gemm_planes_relu(const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b, const float* bias, float* c,
                          int64_t ldc, int32_t m, int32_t n, int32_t k, int32_t split_k, int32_t products, float* workspace,
                          uint32_t* amax_out, int32_t amax_zeroed, ptmi_stream_t stream);
int ptmi_relu_backward_absmax(const float* g, const float* y, float* out, int64_t rows, int64_t cols, int64_t ld_g, int64_t ld_y,
                              int64_t ld_out, uint32_t* amax_out, int32_t amax_zeroed, ptmi_stream_t stream);
/* The bf16 flavour of the three calls above: the planes hold bf16 (hi, lo) halves, no operand scale (fp32's exponent range).
 * It exists for the LSTM input gradient dx = dgates W_ih (torch.nn.LSTM backward inside pit/model.py:60-66): the persistent
 * backward recurrence hands its gate gradients on as exactly such planes (its scratch, ptmi_lstm_handoff_cols), so the
 * GEMM takes them as operand A as they lie and only W_ih is packed (once per optimizer step). */
int ptmi_pack_planes_t_bf16(const float* x, int64_t k_rows, int64_t cols, int64_t ld, uint16_t* out, ptmi_stream_t stream);
int ptmi_pack_planes_n_bf16(const float* x, int64_t rows, int64_t k, int64_t ld, uint16_t* out, ptmi_stream_t stream);
/* Any of the four pack passes writing INTO a wider operand: `out` are planes with kb_total k blocks (of 32) per row tile; the
 * source's k blocks land at k block kb_offset, kb_count of them (>= ceil(k / 32); the surplus is zero).  x is [rows_or_k][cols]:
 * transposed = 0: rows x k (ptmi_pack_planes_n), 1: k x operand rows (ptmi_pack_planes_t); bf16 = 1: the bf16 flavour (amax NULL).
 * The stacked input weights of a BLSTM layer as the operand of the previous layer's hand-off planes (k = H columns per direction,
 * padded to ptmi_lstm_handoff_cols) are packed direction by direction this way - no padded fp32 copy. */
int ptmi_pack_planes_into(const float* x, int64_t rows_or_k, int64_t cols, int64_t ld, int32_t transposed, int32_t bf16, const uint32_t* amax,
                          uint16_t* out, int64_t kb_total, int64_t kb_offset, int64_t kb_count, ptmi_stream_t stream);
int ptmi_gemm_planes_bf16(const uint16_t* a, const uint16_t* b, const float* bias, float* c, int64_t ldc, int32_t m, int32_t n,
                          int32_t k, int32_t accumulate, int32_t split_k, int32_t products, float* workspace, ptmi_stream_t stream);
/* ptmi_gemm_planes_bf16 with a TWO-PART output: rows m < m_split of the product go to c, rows m >= m_split to c2 (row m - m_split;
 * same row stride ldc; m_split a multiple of 16; c and c2 equally aligned).  Both directions' dW_ih = [dgates_f | dgates_r]^T x of a
 * BLSTM layer (torch.nn.LSTM backward, pit/model.py:60-66: weight_ih_l<k> and weight_ih_l<k>_reverse are separate parameters) as
 * ONE launch over the operand the backward recurrence hands on (both directions' dgates^T planes lie behind each other). */
int ptmi_gemm_planes_bf16_two(const uint16_t* a, const uint16_t* b, float* c, float* c2, int32_t m_split, int64_t ldc, int32_t m,
                              int32_t n, int32_t k, int32_t accumulate, int32_t split_k, int32_t products, float* workspace,
                              ptmi_stream_t stream);
/* Calls without split K run on a persistent big-tile kernel (8 wavefronts, workgroup tile picked per problem by a cost model:
 * 0 = 256 x 320, 1 = 256 x 256, 2 = 256 x 192, 3 = 128 x 320, 4 = 128 x 256) or on the 128 x 128 kernel (5) that also carries
 * every co-resident split-K call; ptmi_gemm_planes_plan reports the choice.  (Tests and sweeps pin a tile through the environment
 * variable PTMI_GEMM_TILE, read at every call; there is no entry point for it.) */
/* What a ptmi_gemm_planes[_bf16] call of this shape runs as: 100 * tile + k ranges (tile as above; measurement / labelling only). */
int32_t ptmi_gemm_planes_plan(int32_t m, int32_t n, int32_t k, int32_t split_k);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer step on the Trainer's flat gradient bucket (csrc/optim.hip): replaces, on the step path of
 * padertorch/train/trainer.py:512-532, torch.nn.utils.clip_grad_norm_ (padertorch/train/optimizer.py:35-42),
 * torch.optim.Adam.step (padertorch/train/optimizer.py:79-90 -> :31-33) and zero_grad (:27-29).
 *
 * ptmi_grad_norm: norm_out[0] = 2-norm of flat[0:n] (device fp32; 16-byte aligned).  Reproducible: fixed
 *   per-workgroup slices summed in double, folded in a fixed order.  workspace = device buffer of
 *   ptmi_grad_norm_workspace_elems() doubles.
 *
 * ptmi_adam_flat: for every element i of the bucket (segment s = the parameter tensor it belongs to):
 *     g = flat_grad[i] * min(1, max_norm / (norm[0] + 1e-6))          (norm == NULL: no clipping)
 *     g += weight_decay * p;  m += (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g
 *     p -= (lr / (1 - beta1^t)) * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps)),   t = step[0] + 1
 *   i.e. torch.optim.Adam's arithmetic (amsgrad / maximize not supported); flat_grad[i] = 0 afterwards when
 *   zero_grad != 0.  The update is SKIPPED (gradients are still zeroed) when found_inf (device fp32 scalar or NULL)
 *   is non-zero - the semantics of torch's fused Adam -, when `finite` (device fp32 scalar or NULL, e.g. the sum of the
 *   step's losses) or norm[0] is not finite; applied (device fp32 scalar or NULL) receives 1 / 0 = applied / skipped.  segments = device int64 [nseg][3]: parameter pointer, first flat
 *   index, element count, ascending and dense over [0, n).  exp_avg / exp_avg_sq: flat fp32 [n].  The caller
 *   advances `step` (a device fp32 scalar) afterwards.
 *   hyper (device doubles [6] = lr, beta1, beta2, eps, weight_decay, max_norm; 8-byte aligned; or NULL): when given, the kernel reads
 *   the hyper-parameters from these words INSTEAD of the by-value arguments - a step replayed from a hipGraph has its kernel
 *   arguments frozen, and padertorch's hooks rewrite param_group['lr'] between iterations (padertorch/train/hooks.py:736,1029). */
/* ptmi_lstm_bias_grad_add: bias_ih_grad[d][i] += db[d][i] and bias_hh_grad[d][i] += db[d][i] for every direction d (HOST arrays of
 * ndir device pointers; db [ndir][n] = the bias gradient the persistent backward recurrence leaves in its scratch): torch.nn.LSTM
 * keeps two bias vectors per direction (pit/model.py:60-66) that receive the same gradient - one launch instead of 2 ndir. */
int ptmi_lstm_bias_grad_add(const float* db, int32_t ndir, int32_t n, float* const* bias_ih_grad, float* const* bias_hh_grad,
                            ptmi_stream_t stream);
int64_t ptmi_grad_norm_workspace_elems(void);
int ptmi_grad_norm(const float* flat, int64_t n, double* workspace, float* norm_out, ptmi_stream_t stream);
int ptmi_adam_flat(float* flat_grad, float* exp_avg, float* exp_avg_sq, const int64_t* segments, int32_t nseg, int64_t n,
                   const float* norm, float max_norm, const float* found_inf, const float* finite, float* applied,
                   const float* step, double lr, double beta1, double beta2, double eps, double weight_decay, const double* hyper,
                   int32_t zero_grad, ptmi_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data parallelism: the gradient exchange of padertorch/train/trainer.py:396-442 (parallel_apply over the devices of ONE
 * process, gradients of the replicas summed at :426-428 - accumulated, NOT averaged) as one process per GPU and one RCCL
 * all_reduce(SUM) over xGMI on the flat fp32 gradient bucket.  librccl is opened on first use (dlopen).
 *
 * ptmi_comm_unique_id : rank 0 draws the 128-byte rendezvous id and hands it to the other ranks (file, socket, MPI, env).
 * ptmi_comm_create    : collective over all ranks; binds the device that is CURRENT in the calling thread.
 * ptmi_allreduce_sum  : buffer[0:n] (device fp32) <- sum over ranks, in place, enqueued on `stream`; no division by the
 *                       world size.  Call it for the same n in the same order on every rank.
 * Errors: PTMI_E_UNSUPPORTED when librccl cannot be opened; RCCL's own result r is returned as -100 - r.
 * ------------------------------------------------------------------------------------------- */
#define PTMI_COMM_ID_BYTES 128
typedef struct ptmi_comm ptmi_comm;
int32_t ptmi_comm_rccl_version(void);                  /* e.g. 22105 for 2.21.5; 0: librccl not available */
int ptmi_comm_unique_id(uint8_t* id_out /* [PTMI_COMM_ID_BYTES] */);
int ptmi_comm_create(ptmi_comm** comm, int32_t world_size, int32_t rank, const uint8_t* id);
int ptmi_allreduce_sum(ptmi_comm* comm, float* buffer, int64_t n, ptmi_stream_t stream);
int ptmi_comm_destroy(ptmi_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H_ */
