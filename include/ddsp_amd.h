/*
 * libddsp_amd - C ABI of the MI355X (gfx950) DDSP harmonic-plus-noise synthesis path.
 *
 * The reference (tarepan/DDSP-SVC-official) has no FFI layer: its callers import Python names
 * (`ddsp.vocoder.{load_model,Sins,CombSub,CombSubFast}`, `ddsp.core.upsample`, `ddsp.loss.RSSLoss`).
 * This header is the boundary a binding for that path attaches to; each entry point names the
 * reference code it replaces (paths relative to the reference root).  The Python mirror under
 * `ddsp-svc-official_amd/ddsp/` binds it with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`;
 *   - tensors are dense, row-major, fp32 unless stated; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     except scratch growth inside ddsp_ctx_reserve / the first call at a larger size;
 *   - return value: DDSP_OK or a negative DDSP_ERR_*; ddsp_last_error() gives the text;
 *   - one ddsp_ctx per stream/thread; a ctx holds scratch + constant tables, no model state.
 */
#ifndef DDSP_AMD_H
#define DDSP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDSP_OK 0
#define DDSP_ERR_ARG (-1)   /* shape / mode / null-pointer contract violated (Python: ValueError)   */
#define DDSP_ERR_HIP (-2)   /* a HIP runtime call or launch failed            (Python: RuntimeError) */
#define DDSP_ERR_OOM (-3)   /* scratch allocation failed                      (Python: RuntimeError) */

typedef struct ddsp_ctx ddsp_ctx;

/* ---- handle ------------------------------------------------------------------------------- */
int ddsp_ctx_create(ddsp_ctx** out, int device);
int ddsp_ctx_destroy(ddsp_ctx* ctx);
const char* ddsp_last_error(const ddsp_ctx* ctx);
/* Pre-size the scratch arena (bytes).  Optional; calls grow it on demand (with a device sync). */
int ddsp_ctx_reserve(ddsp_ctx* ctx, uint64_t bytes);
int ddsp_abi_version(void);
/* Device-side contract violations cannot fail the call that launched them (nothing synchronises): a kernel that meets
 * one - a speaker id outside [1, n_spk], where the reference's nn.Embedding raises - skips the access and sets a flag in
 * host-mapped memory.  The next ddsp_unit2ctrl_* call on the context, or this function (call it after synchronising the
 * stream), returns DDSP_ERR_ARG once and clears the flag. */
int ddsp_ctx_poll_error(ddsp_ctx* ctx);
/* Product arithmetic of the INFERENCE contractions that have both forms (the Linear / conv GEMMs of ddsp_unit2ctrl_fwd,
 * the inverse-DFT GEMMs of ddsp_fir_from_ctrl; ddsp_ltv_fir takes its own `math` argument):
 *   DDSP_MATH_SPLIT_BF16 (default): every fp32 product from three bf16 matrix products (hi*hi + hi*lo + lo*hi), fp32
 *                         accumulation, ~4e-6 relative error per contraction - narrower than the reference's fp32;
 *   DDSP_MATH_FP32:       fp32 matrix products (v_mfma_f32_*_f32), ~3e-7 - the reference's precision class.
 * Training: the weight-gradient and input-gradient GEMMs of ddsp_unit2ctrl_bwd / ddsp_unit2ctrl_bwd_kept FOLLOW this mode
 * too (split-bf16 by default: transposed pre-split weight copies, wgrad_bf16.h); the attention / FIR / filter-synthesis
 * adjoints, the spectral loss and every activation a backward call rebuilds use fp32 products in either mode.  A caller that
 * needs reference-class fp32 gradients sets DDSP_MATH_FP32 around the backward call as well.  The Python mirror runs the
 * training FORWARD in fp32 products (dL/dsignal amplifies a 4e-6 disagreement between forward and backward) and leaves the
 * backward on the context's mode.  GEMMs whose shape keeps them off the LDS-DMA kernel (K not a multiple of 32, unaligned
 * rows) run fp32 products in either mode. */
#define DDSP_MATH_FP32 0
#define DDSP_MATH_SPLIT_BF16 3
int ddsp_ctx_set_math(ddsp_ctx* ctx, int math);
int ddsp_ctx_get_math(const ddsp_ctx* ctx);

/* ---- a1: frame -> sample linear upsampler ------------------------------------------------- */
/* replaces ddsp/core.py:7-21 `upsample(signal(B,Fr,C), factor)`; out (B, Fr*hop, C):
 * out[t] = fma(1-a, x[t/hop], a*x[min(t/hop+1, Fr-1)]), a = (t%hop)/hop. */
int ddsp_upsample(ddsp_ctx* ctx, void* stream, const float* x, int64_t B, int64_t Fr, int64_t C, int hop,
                  float* out);

/* ---- a1-a3: f0 frames -> wrapped rotation, frame phases, combtooth ------------------------- */
#define DDSP_COMB_NONE 0
#define DDSP_COMB_SINC 1        /* ddsp/vocoder.py:539  CombSub                                   */
#define DDSP_COMB_SINC_GATED 2  /* ddsp/vocoder.py:459-460 CombSubFast: comb = 0 where f0 <= 0     */
/* replaces upsample(f0) + fo_to_rot (ddsp/core.py:31-51) + `2*pi*rot[:, ::hop]` (ddsp/vocoder.py:515-517)
 * + torch.sinc(sr*rot/(f0+1e-3)).  `precise` != 0 is the reference's infer=True (fp64 increments);
 * 0 is infer=False (fp32 increments, fp64 running sum rounded to fp32 per sample).
 * f0_frames (B,Fr); initial_phase (B,) radians or NULL; outputs (any may be NULL except phase_frames):
 * rot (B,T) cycles in [-0.5,0.5]; phase (B,T) = 2*pi*rot (Sins, ddsp/vocoder.py:392); comb (B,T);
 * f0_up (B,T) upsampled f0; phase_frames (B,Fr) radians. T = Fr*hop. */
int ddsp_phase_scan(ddsp_ctx* ctx, void* stream, const float* f0_frames, const float* initial_phase, int64_t B,
                    int64_t Fr, int hop, int sr, int precise, int comb_mode, float* rot, float* phase, float* comb,
                    float* f0_up, float* phase_frames);

/* ---- a5-a6: control frames -> linear-phase FIR frames -------------------------------------- */
#define DDSP_FIR_ALLPASS 0  /* mags = exp(j*cumsum(pi*tanh(ctrl))), no window (ddsp/vocoder.py:521,540) */
#define DDSP_FIR_DYNAMIC 1  /* mags = exp(ctrl), raised-cosine of half width 1.5*sr/(f0+1e-3) (:522,541-542; ddsp/core.py:292-303) */
#define DDSP_FIR_STATIC 2   /* mags = exp(ctrl)/128, periodic Hann (:523,546; ddsp/core.py:242-289)  */
/* replaces the ctrl activation + ddsp/core.py:306-328 `_frequency_impulse_response`.
 * ctrl: (B*Fr) rows of `n_mag` values at row stride `ctrl_ld` (so a column block of the fused control
 * matrix can be passed in place); f0_frames (B*Fr) (DYNAMIC only); ir out (B*Fr, n = 2*(n_mag-1)). */
int ddsp_fir_from_ctrl(ddsp_ctx* ctx, void* stream, int mode, const float* ctrl, int64_t ctrl_ld, int n_mag,
                       const float* f0_frames, int64_t rows, int sr, float* ir);

/* ---- a7: frame-varying FIR (50%-overlap triangular cross-fade between frame filters) -------- */
/* replaces ddsp/core.py:185-239 `_fft_convolve` (+ :147-182 crop): out[u] = sum over taps of
 * x[t]*((1-j/hop)*ir[m] + (j/hop)*ir[min(m+1,Fr-1)])[u + n/2 - t], t = m*hop + j.
 * excitation: DDSP_EXC_AUDIO      audio (B,T) is the input signal;
 *             DDSP_EXC_UNIT_NOISE audio (B,T) holds U[0,1) draws and the input is 2*u-1 (the reference's
 *                                 `torch.rand_like(x)*2-1`, ddsp/vocoder.py:545, with the draw injected);
 *             DDSP_EXC_GENERATE   audio is NULL, the input is a counter-based U[-1,1) stream keyed by
 *                                 noise_seed (perf mode: the CPU mt19937 stream cannot be reproduced on a GPU).
 * out (B,T) or NULL receives the filtered signal; out_sum (B,T) or NULL receives filtered + add_in
 * (fuses `harmonic + noise`, ddsp/vocoder.py:548); add_in and out_sum are given together.
 * math: DDSP_FIR_FP32 = products on the fp32 matrix pipe (training forward, tight-tolerance checks);
 *       DDSP_FIR_SPLIT_BF16 = every fp32 product formed from three bf16 matrix products (hi*hi + lo*hi + hi*lo,
 *       fp32 accumulation, ~4e-6 relative error; the inference path).  Filters too long for that kernel's
 *       staging fall back to the fp32 kernel.  (Values 31..58 force one block shape of the split-bf16 kernels or
 *       switch their products off - measurement aids of tools/fir_bf16_check.py, not part of the interface.)
 * Requires hop == 512 and n even, 32 <= n <= 2046. */
#define DDSP_FIR_FP32 0
#define DDSP_FIR_SPLIT_BF16 3
#define DDSP_EXC_AUDIO 0
#define DDSP_EXC_UNIT_NOISE 1
#define DDSP_EXC_GENERATE 2
int ddsp_ltv_fir(ddsp_ctx* ctx, void* stream, const float* audio, int excitation, uint64_t noise_seed,
                 const float* ir, int64_t B, int64_t Fr, int hop, int n, const float* add_in, float* out,
                 float* out_sum, int math);


/* ---- a4: unit -> control network ---------------------------------------------------------- */
/* Device pointers into the model's state dict (reference key names), fp32, dense.
 * Shapes: conv weights (Cout, Cin, k) as torch stores them; Linear weights (out, in); d = 256.
 *   prenet_conv1_w (256, n_unit, 3)  prenet_gn_* (256)  prenet_conv2_w (256, 256, 3)
 *   f0_w/phase_w/volume_w (256, 1) + biases (256)        spk_table (n_spk, 256)
 *   per layer l0..l2: norm_* (256); q/k/v_w (512, 256) + b (512); proj (266, 64); out_w (256, 512) + b (256);
 *                     cm_ln_* (256); cm_pw1_w (1024, 256, 1); cm_dw_w (512, 1, 31); cm_pw2_w (256, 512, 1)
 *   final_ln_* (256); head_g (n_out, 1); head_v (n_out, 256); head_b (n_out)                               */
typedef struct ddsp_u2c_layer {
    const float *norm_w, *norm_b, *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *proj, *out_w, *out_b;
    const float *cm_ln_w, *cm_ln_b, *cm_pw1_w, *cm_pw1_b, *cm_dw_w, *cm_dw_b, *cm_pw2_w, *cm_pw2_b;
} ddsp_u2c_layer;

typedef struct ddsp_u2c_weights {
    const float *prenet_conv1_w, *prenet_conv1_b, *prenet_gn_w, *prenet_gn_b, *prenet_conv2_w, *prenet_conv2_b;
    const float *f0_w, *f0_b, *phase_w, *phase_b, *volume_w, *volume_b, *spk_table;
    int n_spk, n_unit, n_out;
    int causal;   /* 0 = the shipped configs (`c: false`); 1 = causal convolutions + causal linear attention */
    ddsp_u2c_layer layer[3];
    const float *final_ln_w, *final_ln_b, *head_g, *head_v, *head_b;
    /* The caller's change counter of the weight VALUES, or 0.  ddsp_unit2ctrl_fwd prepares the weights for its kernels on every
     * call (weight norm of the head, re-ordered pw1 rows, bf16 hi/lo copies: two launches, ~1.5 % of a 64-clip forward).  With
     * version != 0 it keeps the prepared copies in the context and reuses them while every pointer and integer of this struct
     * AND the version are what they were - so bump it whenever a tensor's contents change (torch: the sum of the parameters'
     * `_version` counters does that).  Ignored while the stream is being captured into a HIP graph (a replay must see the
     * weights of its own time) and by the training entry points. */
    uint64_t version;
} ddsp_u2c_weights;

/* replaces ddsp/unit2control.py:68-101 `Unit2Control.forward` + ddsp/pcmer.py (non-causal, c=False only).
 * units (B,Fr,n_unit); f0_frames, phase_frames, volume (B,Fr); spk_id (n_spk_id) int64, 1-based, n_spk_id in
 * {1, B} (the reference broadcasts a (1,1) id); speaker mixing (`spk_mix_dict`): n_mix > 0 host arrays
 * mix_ids_host (1-based) / mix_w_host replace spk_id (n_mix <= 16).  ctrl out (B,Fr,n_out), the fused
 * control matrix that `split_to_dict` (ddsp/unit2control.py:10-20) views.  The `weights` struct itself is a
 * HOST pointer. */
int ddsp_unit2ctrl_fwd(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* weights_host, const float* units,
                       const float* f0_frames, const float* phase_frames, const float* volume,
                       const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                       const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, float* ctrl);

/* Backward of ddsp_unit2ctrl_fwd for training (reference: autograd through Unit2Control, solver.py:113).  The call
 * re-runs the forward keeping its activations in the scratch arena, then back-propagates d_ctrl (B,Fr,n_out) to
 * every parameter.  `grads_host` has the layout of ddsp_u2c_weights; each pointer receives the gradient of the
 * like-named parameter (written, not accumulated; fast_attention.projection_matrix is a buffer: `proj` is ignored).
 * ctrl_out (B,Fr,n_out) or NULL additionally receives the forward result.  Inputs carry no gradient. */
int ddsp_unit2ctrl_bwd(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* weights_host, const float* units,
                       const float* f0_frames, const float* phase_frames, const float* volume,
                       const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                       const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, const float* d_ctrl,
                       const ddsp_u2c_weights* grads_host, float* ctrl_out);

/* A training step without the second forward: ddsp_unit2ctrl_fwd_keep is ddsp_unit2ctrl_fwd in fp32 products that leaves
 * its activations (~40 KB per frame and layer) in `keep`, a caller-owned device region of ddsp_unit2ctrl_keep_bytes(w, B, Fr)
 * bytes, 256-byte aligned (the autograd node owns it between the two calls: reference `solver.py:111-113`, where PyTorch
 * keeps the activations); ddsp_unit2ctrl_bwd_kept back-propagates from them exactly like ddsp_unit2ctrl_bwd, which re-runs
 * the forward instead.  The weights must not change between the two calls. */
int64_t ddsp_unit2ctrl_keep_bytes(const ddsp_u2c_weights* w, int64_t B, int64_t Fr);
int ddsp_unit2ctrl_fwd_keep(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* w, const float* units,
                            const float* f0_frames, const float* phase_frames, const float* volume,
                            const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                            const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, void* keep, int64_t keep_bytes,
                            float* ctrl);
int ddsp_unit2ctrl_bwd_kept(ddsp_ctx* ctx, void* stream, const ddsp_u2c_weights* w, const float* units,
                            const float* f0_frames, const float* phase_frames, const float* volume,
                            const int64_t* spk_id, int64_t n_spk_id, const int64_t* mix_ids_host,
                            const float* mix_w_host, int n_mix, int64_t B, int64_t Fr, void* keep, int64_t keep_bytes,
                            const float* d_ctrl, const ddsp_u2c_weights* grads_host);

/* ---- backward of a5-a8 (training: reference autograd through frequency_filter, solver.py:113) ------------ */
/* Adjoints of ddsp_ltv_fir for an upstream gradient d_out (B,T): d_audio (B,T) or NULL = gradient w.r.t. the
 * input signal; d_ir (B,Fr,n) or NULL = gradient w.r.t. the filter frames (needs the forward input: audio with
 * its excitation mode, or DDSP_EXC_GENERATE + the same noise_seed). */
int ddsp_ltv_fir_bwd(ddsp_ctx* ctx, void* stream, const float* audio, int excitation, uint64_t noise_seed,
                     const float* ir, const float* d_out, int64_t B, int64_t Fr, int hop, int n, float* d_audio,
                     float* d_ir);
/* Adjoint of ddsp_fir_from_ctrl: d_ir (rows, n) [scaled in place by the dynamic window in DYNAMIC mode] ->
 * d_ctrl, written to `n_mag` columns at row stride d_ctrl_ld (a column block of the control-gradient matrix). */
int ddsp_fir_from_ctrl_bwd(ddsp_ctx* ctx, void* stream, int mode, const float* ctrl, int64_t ctrl_ld, int n_mag,
                           const float* f0_frames, int64_t rows, int sr, float* d_ir, float* d_ctrl,
                           int64_t d_ctrl_ld);

/* ---- a10: additive sinusoid bank ----------------------------------------------------------- */
/* replaces ddsp/vocoder.py:397,402-412 + ddsp/core.py:24-28: amplitudes exp(ctrl)/128 masked by
 * ((k*f0 < sr/2) + 1e-7), upsampled per harmonic, times sin(k*phase), summed over k = 1..n_harmonics.
 * ctrl rows (B*Fr) at stride ctrl_ld; f0_frames (B*Fr); phase (B,T) radians as produced by
 * ddsp_phase_scan; out (B,T).  hop must be a power of two. */
int ddsp_sins_bank(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, int n_harmonics,
                   const float* f0_frames, const float* phase, int64_t B, int64_t Fr, int hop, int sr, float* out);

/* ---- a11: CombSubFast windowed spectral overlap-add ------------------------------------------ */
/* replaces ddsp/vocoder.py:462-490.  ctrl rows (B*Fr) at stride ctrl_ld hold
 * [harmonic_magnitude 513 | harmonic_phase 513 | noise_magnitude 513]; comb (B,T) from ddsp_phase_scan
 * (DDSP_COMB_SINC_GATED); noise (B,T) U[0,1) draws with excitation DDSP_EXC_UNIT_NOISE, or NULL with
 * DDSP_EXC_GENERATE + noise_seed; out (B,T).  hop == 512 (1024-point frames). */
int ddsp_spectral_ola(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, const float* comb,
                      const float* noise, int excitation, uint64_t noise_seed, int64_t B, int64_t Fr, int hop,
                      float* out);

/* Adjoints of a10 / a11 w.r.t. their control blocks (training of Sins / CombSubFast): d_out (B,T) -> d_ctrl written
 * to the same column block layout as the forward reads, at row stride d_ctrl_ld. */
int ddsp_sins_bank_bwd(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, int n_harmonics,
                       const float* f0_frames, const float* phase, const float* d_out, int64_t B, int64_t Fr, int hop,
                       int sr, float* d_ctrl, int64_t d_ctrl_ld);
int ddsp_spectral_ola_bwd(ddsp_ctx* ctx, void* stream, const float* ctrl, int64_t ctrl_ld, const float* comb,
                          const float* noise, int excitation, uint64_t noise_seed, const float* d_out, int64_t B,
                          int64_t Fr, int hop, float* d_ctrl, int64_t d_ctrl_ld);

/* ---- a13: random-scale spectral loss ---------------------------------------------------------- */
/* replaces ddsp/loss.py:7-43 `RSSLoss.forward(x_pred, x_true)` for a given draw of scales: n_ffts_host holds the
 * n_scale values the reference draws with torch.randint(fft_min, fft_max) (ddsp/loss.py:39; the caller draws, so
 * data-parallel ranks can share one draw).  Per scale N: Hann(periodic) window, center=False, hop = hops_host[s] -
 * the caller evaluates the reference's `int(n_fft * (1 - overlap))` (ddsp/loss.py:13) in its own floating point - or
 * hop = N when hops_host is NULL (overlap = 0, what every reference caller uses); magnitude / sqrt(sum w^2) + eps;
 * L_N = mean_b ||S_t-S_p||_F/||S_t+S_p||_F + alpha*mean|ln S_t - ln S_p|; loss = mean_N.
 * x_pred, x_true (B,T) 16-byte aligned, T % 4 == 0; loss: device float[1]; grad_pred (B,T) or NULL receives
 * d loss / d x_pred (the caller's autograd scales it by the upstream gradient). */
int ddsp_rss_loss(ddsp_ctx* ctx, void* stream, const float* x_pred, const float* x_true, int64_t B, int64_t T,
                  const int* n_ffts_host, const int* hops_host, int n_scale, float alpha, float eps, float* loss,
                  float* grad_pred);

/* ---- a14: SOLA splice of the real-time path -------------------------------------------------- */
/* replaces gui.py:405-430: within audio[-block-xfade-search-delay : -delay] find the lag (0..search) that
 * maximises the energy-normalised correlation with sola_buffer (xfade,), cross-fade (sin^2 windows,
 * gui.py:349-351) the head with sola_buffer, emit `block` samples, and keep the next xfade samples in
 * sola_buffer (updated in place).  shift: device int32 (no host sync). */
int ddsp_sola(ddsp_ctx* ctx, void* stream, const float* audio, int64_t n_audio, int block, int xfade, int search,
              int delay, float* sola_buffer, float* emitted, int* shift);

/* ---- a15: volume gate ------------------------------------------------------------------------ */
/* replaces gui.py:14-31 `phase_vocoder(a, b, fade_out, fade_in)` (the optional cross-fade of gui.py:417-423; SURVEY
 * 8(f) rank 3): a = kept tail, b = head of the new block, both (n), fade windows (n), out (n).  n <= 65536. */
int ddsp_phase_vocoder(ddsp_ctx* ctx, void* stream, const float* a, const float* b, const float* fade_out,
                       const float* fade_in, int n, float* out);

/* replaces main.py:111-116,159 / gui.py:108-112,127: signal (B,T) *= upsample(dilate9(volume > threshold)),
 * in place (threshold = 10^(dB/20), linear); volume (B,Fr). */
int ddsp_volume_gate(ddsp_ctx* ctx, void* stream, float* signal, const float* volume, float threshold, int64_t B,
                     int64_t Fr, int hop);

/* ---- the device-side steps immediately before the synthesis path ------------- */
/* replaces ddsp/vocoder.py:116-137 `Volume_Extractor.extract`: audio (B,T) -> volume (B, T/hop + 1), the RMS of
 * non-overlapping hop-sized blocks of the signal reflect-padded by (hop/2, (hop+1)/2) (numpy 'reflect': T must
 * exceed (hop+1)/2). */
int ddsp_volume_extract(ddsp_ctx* ctx, void* stream, const float* audio, int64_t B, int64_t T, int hop,
                        float* volume);
/* replaces the alignment tail of ddsp/vocoder.py:201-211 `Units_Encoder.encode`: out[b][i][:] = units[b][j][:] with
 * j = min(rint(ratio * i), Lu - 1), ratio = (hop/sample_rate) / (encoder_hop/encoder_sample_rate) as fp32, rint =
 * round half to even (torch.round); units (B,Lu,C) -> out (B,n_frames,C). */
int ddsp_align_units(ddsp_ctx* ctx, void* stream, const float* units, int64_t B, int64_t Lu, int64_t C,
                     int64_t n_frames, float ratio, float* out);

/* replaces the host-side f0 re-timing of enhancer.py:56-62 (`f0_np *= real_factor`; `np.interp(time_frame, time_org, f0_np,
 * left=f0_np[0], right=f0_np[-1])`): out[i] = interp(i * step_dst) over the knots x_j = (step_num * j) / div with values
 * fl32(f0[j] * scale), evaluated in fp64 like numpy; f0 (n_src,) -> out (n_dst,).  No host copy of the track. */
int ddsp_retime_f0(ddsp_ctx* ctx, void* stream, const float* f0, int64_t n_src, double step_num, double div, float scale,
                   double step_dst, int64_t n_dst, float* out);

/* ---- sample-rate conversion ------------------------------------------------------ */
/* replaces `torchaudio.transforms.Resample(orig_freq, new_freq, lowpass_filter_width)` as the reference uses it (gui.py:399-404,
 * enhancer.py:50-53,69-73; windowed-sinc polyphase, Hann window, rolloff 0.99 - torchaudio's published algorithm; the package
 * is not in the image, so parity at that boundary is unpinned): x (B,T) -> out (B, ddsp_resample_length(T, orig, new)).
 * The tap table of a rate pair is built on first use and cached in the context. */
int64_t ddsp_resample_length(int64_t T, int orig_freq, int new_freq);
int ddsp_resample(ddsp_ctx* ctx, void* stream, const float* x, int64_t B, int64_t T, int orig_freq, int new_freq,
                  int lowpass_filter_width, float* out);

/* ---- the NSF-HiFiGAN post-net (enhancer.py:24-101, nsf_hifigan/models.py:106-276, nvSTFT.py:65-119) ---- */
/* One utterance per call; activations frame-major (T, C) fp32.
 * ddsp_conv1d: replaces `Conv1d(Cin, Cout, k, dilation=d, padding="same")(leaky_relu(x, in_slope))` (+ residual): the
 *   resblock convolutions (models.py:45-77), conv_pre (:234) and - with weights packed as the host mirror's
 *   `_pack_conv_transpose` does - the ConvTranspose1d upsamplers (:240-243).  x (T,Cin), w_packed (Cout, k*Cin) with column
 *   tap*Cin + ci, bias (Cout) or NULL, residual (T,Cout) or NULL; in_slope = 1 applies no activation.  The result y goes to
 *   out (T,Cout) and / or, as leaky_relu(y, act_slope), to out_act (either may be NULL): every convolution of the generator
 *   reads an activated input, so a producer that emits it lets the consumer run with in_slope = 1 - and only then (with
 *   Cin % 32 == 0) does the convolution run on the LDS-DMA GEMM in the context's product arithmetic; otherwise on the
 *   register-staged fp32 kernel, which activates while loading.  Split layout (split-bf16 arithmetic only): every group of 8
 *   consecutive floats of a row replaced by its 8 bf16 high parts and 8 bf16 remainders (32 bytes, same footprint) - what the
 *   matrix instructions consume; w_split is w_packed in that layout (the host mirror converts once at load), flags say
 *   whether x arrives so and whether out_act is to be written so (Cout % 64 == 0): a chain of convolutions then converts
 *   every activation once, where it is produced, instead of once per tile that reads it.
 * ddsp_nsf_source: replaces `SourceModuleHnNSF.forward(f0, upp)` (:180-216 with `SineGen` :106-177, 9 harmonics): f0 (L) Hz per
 *   frame, rand_ini (9) the harmonics' initial phases in cycles (the reference's torch.rand draw, element 0 = 0), lin_w (9),
 *   lin_b (1) of `l_linear`; out (L*upp) = tanh(linear(sine_amp * sin(2 pi cumsum(f0 h / sr)))).
 * ddsp_nsf_noise_conv: replaces `noise_convs[i]` (:244-249), a Conv1d(1, C, K, stride, padding=pad) on the source signal:
 *   src (T_src), w (C,K), b (C) -> out (T_out, C).
 * ddsp_nsf_post: replaces `tanh(conv_post(leaky_relu(x, slope)))` (:268-270): x (T,C), w (K,C) tap-major, b (1) -> out (T).
 * ddsp_nsf_mean: (a [+ b [+ c]]) / n_terms elementwise - the mean over a stage's residual blocks (:259-266) - to out and / or,
 *   activated, to out_act (as in ddsp_conv1d).
 * ddsp_log_mel: replaces the spectral half of `STFT.get_mel` (nvSTFT.py:100-117): frames (n_frames, n_fft) of the padded
 *   signal, dft_table (2*ldm, n_fft) rows (w cos, -w sin) per bin with ldm = bins rounded up to 4, mel_basis (n_mels, ldm);
 *   out (n_frames, n_mels) = log(max(mel . sqrt(re^2 + im^2 + 1e-9), clip)). */
#define DDSP_CONV_X_SPLIT 1   /* x is in the split layout (written so by a producer with DDSP_CONV_ACT_SPLIT) */
#define DDSP_CONV_ACT_SPLIT 2 /* write out_act in the split layout */
int ddsp_conv1d(ddsp_ctx* ctx, void* stream, const float* x, const float* w_packed, const float* bias, int64_t T, int Cin,
                int Cout, int ktaps, int dil, float in_slope, const float* residual, float* out, float* out_act,
                float act_slope, const float* w_split, int flags);
/* One residual pair of `ResBlock1` (nsf_hifigan/models.py:41-90: xt = c1(leaky_relu(x)); xt = c2(leaky_relu(xt)); x = xt + x) of a
 * narrow stage in one launch: x (T,C) raw, c1 with `dil`, c2 with dilation 1, both `ktaps` taps, weights packed as for
 * ddsp_conv1d; out = the new x, out_act = leaky_relu(out, slope) (either may be null).  Only x is read and the results
 * written: the activations happen on load and c1's output stays in the LDS.  Products follow the context's arithmetic.
 * ddsp_conv1d_pair_supported: 1 when (C, ktaps, dil) is a geometry the fused kernel takes (else use two ddsp_conv1d calls). */
int ddsp_conv1d_pair_supported(ddsp_ctx* ctx, int C, int ktaps, int dil);
int ddsp_conv1d_pair(ddsp_ctx* ctx, void* stream, const float* x, const float* w1, const float* b1, const float* w2,
                     const float* b2, int64_t T, int C, int ktaps, int dil, float slope, float* out, float* out_act);

int ddsp_nsf_source(ddsp_ctx* ctx, void* stream, const float* f0, const float* rand_ini, const float* lin_w,
                    const float* lin_b, int64_t L, int upp, int sr, float sine_amp, float* out);
int ddsp_nsf_noise_conv(ddsp_ctx* ctx, void* stream, const float* src, int64_t T_src, const float* w, const float* b, int C,
                        int K, int stride, int pad, int64_t T_out, float* out);
int ddsp_nsf_post(ddsp_ctx* ctx, void* stream, const float* x, const float* w, const float* b, int64_t T, int C, int K,
                  float slope, float* out);
int ddsp_nsf_mean(ddsp_ctx* ctx, void* stream, const float* a, const float* b, const float* c, int n_terms, int64_t n,
                  float* out, float* out_act, float act_slope, int flags);
int ddsp_log_mel(ddsp_ctx* ctx, void* stream, const float* frames, const float* dft_table, const float* mel_basis,
                 int64_t n_frames, int n_fft, int n_mels, float clip, float* out);

/* ---- a15: optimiser step --------------------------------------------------------------------- */
/* replaces one parameter's update of torch.optim.AdamW (train.py:41, solver.py:114): decoupled weight decay,
 * bias-corrected moments, `step` counted from 1.  All buffers hold n fp32 values. */
int ddsp_adamw_step(ddsp_ctx* ctx, void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step);
/* the same update for n_tensors parameters that share hyper-parameters and step (one optimizer.step() of
 * solver.py:114 over a param group), a handful of launches instead of one per tensor.  params / grads / exp_avg /
 * exp_avg_sq are HOST arrays of n_tensors device pointers, numel a host array of element counts (0 allowed). */
int ddsp_adamw_step_multi(ddsp_ctx* ctx, void* stream, int n_tensors, float* const* params, const float* const* grads,
                          float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step);

/* ---- building block: fp32-in / fp32-accumulate MFMA GEMM ---------------------------------------- */
/* C[m][n] = sum_k A(m,k) B(k,n) (+ bias[n]).  a_k_contig: A(m,k) = A[m*lda+k] else A[k*lda+m];
 * b_k_contig: B(k,n) = B[n*ldb+k] (nn.Linear weight layout) else B[k*ldb+n].  Rows must start 16-byte aligned
 * (lda, ldb multiples of 4).  tile: 0 = auto, 1 = 64x64, 2 = 64x128, 3 = 128x128; variant: schedule A/B switch.
 * Exposed for unit tests of the block every contraction of the path is built on and for tile tuning. */
int ddsp_gemm_f32(ddsp_ctx* ctx, void* stream, const float* A, int64_t lda, int a_k_contig, const float* B,
                  int64_t ldb, int b_k_contig, const float* bias, float* C, int64_t ldc, int M, int N, int K, int tile,
                  int variant);

/* The fused building block of the control network's residual Linear layers (pcmer.py:221-251 `to_out`, :42-63 pw2) at large
 * batches: X = res + A W^T + bias (M x 256; X may be res) and Y = LayerNorm(X) * gamma + beta (eps 1e-5) in one launch.
 * A (M x K) and W (256 x K) are in the pre-split operand layout (per 8 consecutive k: 8 bf16 hi, then 8 bf16 lo); Y is
 * written in that layout too (y_split & 1) or as fp32; y_split & 2: A is given as plain fp32 rows and split in the kernel.  Same bits as ddsp_gemm_f32 on split operands followed by the
 * LayerNorm kernel.  Exposed for tests. */
int ddsp_gemm_res_ln(ddsp_ctx* ctx, void* stream, const float* A_split, const float* W_split, const float* bias,
                     const float* res, const float* gamma, const float* beta, int M, int K, float* X, float* Y, int y_split);

/* ---- building block: the linear attention of one PCmer layer without its Linear layers ------------------------------ */
/* replaces ddsp/pcmer.py:69-77,123-159 (`softmax_kernel` feature maps + `linear_attention`, non-causal): q, k, v
 * (B*Fr, 512) = 8 heads x 64, proj (266, 64) the layer's `projection_matrix`; out (B*Fr, 512) = the merged heads before
 * `to_out`.  math: DDSP_MATH_FP32 (fp32 matrix products) or DDSP_MATH_SPLIT_BF16 (the feature projections from six bf16
 * piece products - they enter an exponential -, the two context products from three).  Exposed for unit tests of the
 * kernels ddsp_unit2ctrl_fwd runs at inference (it picks the split kernels from 32 utterances on).
 * math = DDSP_ATTENTION_CAUSAL: `causal_linear_attention` (pcmer.py:170-188, `c: true` networks) instead - the running sums
 * over the frames up to and including each one, as chunked products with fp32 arithmetic. */
#define DDSP_ATTENTION_CAUSAL 200
int ddsp_performer_attention(ddsp_ctx* ctx, void* stream, const float* q, const float* k, const float* v,
                             const float* proj, int64_t B, int64_t Fr, float* out, int math);

/* ---- measurement: per-kernel-family HIP-event timing on the launch stream --------------------- */
/* ddsp_profile_begin arms the families in `family_mask` (bit i = family i, see the name returned); while armed,
 * each kernel launch of such a family is bracketed by hipEventRecord on the caller's stream.  ddsp_profile_end
 * disarms, waits for the events and returns one aggregated entry per family that launched: launches, summed
 * duration, and the ALGORITHMIC flops / HBM bytes the library attributes to those launches. */
typedef struct ddsp_prof_entry {
    int family;
    char name[36];
    int64_t launches;
    double ms_total;
    double flops_total;
    double bytes_total;
} ddsp_prof_entry;
int ddsp_profile_begin(ddsp_ctx* ctx, uint64_t family_mask);
/* While armed: change the bracketed families (0 = pause) without dropping the records taken so far - for callers that bracket a
 * SAMPLE of their steps (bench.py: every fifth step of the timed region), since an event record costs ~2 us of stream time. */
int ddsp_profile_mask(ddsp_ctx* ctx, uint64_t family_mask);
int ddsp_profile_end(ddsp_ctx* ctx, ddsp_prof_entry* out, int max_entries, int* n_entries);

#ifdef __cplusplus
}
#endif
#endif /* DDSP_AMD_H */
