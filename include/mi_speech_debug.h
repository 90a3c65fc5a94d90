/* mi_speech_debug.h - diagnostics and test scaffolding exported by libmi_speech.so next to the product ABI (include/mi_speech.h).
 *
 * Nothing here replaces a reference interface and no host application needs it: these entry points exist for DESIGN.md's
 * measurements (launch floor, split-factor model) and for tests that have to provoke conditions a healthy run never meets
 * (compute units held by another stream, counted sampler time-outs).  Kept out of mi_speech.h so that the drop-in surface is
 * exactly what a Swift shim binds. */
#ifndef MI_SPEECH_DEBUG_H
#define MI_SPEECH_DEBUG_H
#include "mi_speech.h"
#ifdef __cplusplus
extern "C" {
#endif

/* diagnostics: microseconds per dependent kernel boundary in a replayed hipGraph of n trivial kernels
 * (mode 0: 1 block x 64 threads, 1: 32 x 1024, 2: 1024 x 256).  DESIGN.md quotes it as the launch floor. */
mis_status mis_debug_launch_floor(int device, int n_kernels, int mode, int reps, double* us_per_kernel);
/* diagnostics: the inter-block split-K factor the engines pick for a weight-streaming GEMM with `items` n-tile groups, `k_tiles`
 * 32-wide k-tiles and `waves_per_item` waves per work item ("Split-K factor from a cost model"); no GPU needed
 * (falls back to 256 CUs when no device is visible). */
int32_t mis_debug_choose_split(int32_t items, int32_t k_tiles, int32_t waves_per_item, int32_t s_max);
/* tests: occupy compute units from ANOTHER stream - `blocks` workgroups of `threads` threads, each reserving 128 KB of the CU's 160 KB of LDS (one per CU, and
 * nothing that needs more than 32 KB of LDS fits beside it), that spin (s_sleep) for `seconds` - so that launches on the library's streams find fewer CUs than the device has (the
 * condition under which the one-launch sampler's row barriers time out and the engines recover on the multi-launch path).  Every
 * spinner announces itself on entry; the call returns MIS_OK only once all `blocks` of them are RESIDENT (handshake through a
 * host-visible counter), or MIS_ERR_DEVICE if that does not happen within two seconds - the spinner is then released and the caller
 * should treat the condition as not reproducible on this device (tests skip).  mis_debug_occupy_wait() releases the spinners early
 * (if they are still running), waits for them and frees the stream. */
mis_status mis_debug_occupy_cus(int device, int blocks, int threads, double seconds);
mis_status mis_debug_occupy_wait(void);
int32_t mis_debug_device_cus(int device);            /* compute units of a device (0 when it does not exist) */
/* diagnostics / tests: launches of the one-launch sampler that reported a timed-out row barrier in this process so far */
int32_t mis_debug_sampler_failures(void);

/* csrc/token_engine.hip (round 5): a whole batch-1 request in ONE persistent launch on the compute units of `xcds` (1, 2, 4 or 8) XCDs,
 * streaming the handle's own packed weights; compiled for Soprano-80M's LM widths (other shapes: MIS_ERR_INVALID_INPUT).  The product
 * reaches it through mis_soprano_generate at batch 1; this entry point is for tests and measurements.
 *   sampling == NULL (laboratory form): `n_prompt` prompt positions, then `n_new` arg-max steps; next_tokens[t] = arg-max id after position
 *     t for EVERY t < n_prompt + n_new, logits_out [n_prompt + n_new][vocab], hidden_out [n_prompt + n_new][hidden] (final-norm output).
 *   sampling != NULL (generate form, the semantics of the Soprano loop, Soprano.swift:801-885): a token after the last prompt position and
 *     after every generated one until `stop_id` or n_new ids - arg-max when sampling->temperature == 0, else "mis-sampler-v1" behind the
 *     Soprano repetition penalty (repetition_penalty over the last repetition_context generated ids, seed, row_offset); next_tokens[t] is
 *     set for t >= n_prompt - 1 (the prompt but its last position runs through the launch chain's batched prefill, whose K/V the engine
 *     imports - as in the product); logits_out row k = the logits the k-th token was drawn from; hidden_out row k = position n_prompt - 1 + k.
 * counts (may be NULL): [0] positions processed, [1] ids chosen.  logits_out / hidden_out may be NULL; ms_out = device time of the launch.
 * Host pointers. */
mis_status mis_debug_token_engine(mis_tts* lm, const int32_t* prompt, int n_prompt, int n_new, int xcds, const mis_gen_params* sampling,
                                  int stop_id, int32_t* next_tokens, float* logits_out, float* hidden_out, int32_t* counts, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MI_SPEECH_DEBUG_H */
