/*
 * Additive C entry points of the MI355X engine.  Nothing here changes the drop-in surface of
 * pocket_tts_c.h; these calls expose (1) options the reference can only set from C++ or its CLI,
 * (2) a batch ingress for many independent utterances, and (3) one entry point per ONNX Runtime
 * call site of the reference so each network can be checked in isolation.  All buffers are host
 * memory; sizes are in elements.  Return value 0 = success, -1 = failure with the message available
 * from pocket_tts_get_last_error(), unless stated otherwise.
 */
#ifndef POCKET_TTS_AMD_H
#define POCKET_TTS_AMD_H

#include "pocket_tts_c.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Full option set of the C++ PocketTTSConfig (include/pocket_tts/pocket_tts.hpp:58-68) plus device
 * placement.  Unlike PocketTTSConfig, temperature is taken literally, so 0 selects the deterministic
 * zero start point of src/pocket_tts.cpp:606-613 (the reference reaches that only through its CLI,
 * src/main.cpp:49-50). */
typedef struct {
    const char* models_dir;      /* NULL: "models/onnx" */
    const char* tokenizer_path;  /* NULL: "models/tokenizer.model"; "" or "none": no tokenizer (id-based calls only) */
    const char* precision;       /* NULL: "int8"; "fp32", "bf16" */
    const char* pack_path;       /* NULL: <models_dir>/pocket_tts.ptts */
    float temperature;
    int lsd_steps;               /* <= 0: 10  */
    int max_frames;              /* <= 0: 500 */
    int frames_after_eos;        /* < 0: 3    */
    int device;                  /* HIP device ordinal */
    int max_text_tokens;         /* <= 0: 512 */
    int has_seed;                /* non-zero: seed the noise generator with `seed` */
    unsigned long long seed;
} PocketTTSConfigEx;

POCKET_TTS_API PocketTTSHandle pocket_tts_create_ex(const PocketTTSConfigEx* config);

/* dims[0..7] = latent_dim, d_model, samples_per_frame, vocab, mimi_dim, sample_rate, lsd_steps, max_frames */
POCKET_TTS_API int pocket_tts_dims(PocketTTSHandle handle, int* dims);

/* pocket_tts_encode_voice (/root/reference/include/pocket_tts/pocket_tts_c.h:63, src/pocket_tts_c.cpp:59-82) over n files in one
   call: host threads decode / resample / attenuate / cut the files while the GPU encodes those already prepared.
   voices_out[i] is what pocket_tts_encode_voice(handle, audio_paths[i]) returns (release with pocket_tts_free_voice);
   on failure (-1, pocket_tts_get_last_error) every voices_out[i] is NULL. */
POCKET_TTS_API int pocket_tts_encode_voices(PocketTTSHandle handle, int n, const char* const* audio_paths, VoiceHandle* voices_out);

/* Voice handle <-> raw conditioning rows [frames][d_model]. */
POCKET_TTS_API VoiceHandle pocket_tts_voice_from_embeddings(const float* embeddings, int frames, int d_model);
POCKET_TTS_API int pocket_tts_voice_frames(VoiceHandle voice);
POCKET_TTS_API const float* pocket_tts_voice_data(VoiceHandle voice);

/* n utterances (any n) through the engine's slot table: up to 64 are in flight on the GPU, each with the offline schedule of
 * src/pocket_tts.cpp:548-657, and a slot whose utterance has stopped is refilled from the rest.
 * results[i] is filled like pocket_tts_generate does; release each with pocket_tts_free_audio. */
POCKET_TTS_API int pocket_tts_generate_batch(PocketTTSHandle handle, int n, const char* const* texts,
                                             const VoiceHandle* voices, AudioResult* results);

/* Same, from token ids (the output of the text pre-step, src/tokenizer.cpp:27-60) and raw conditioning rows.
 * noise[i]: NULL, or [max_frames][latent_dim] start points used instead of the internal generator.
 * latents_out[i]: NULL, or room for max_frames*latent_dim floats; eos_out[i]: NULL or max_frames floats.
 * frames_out[i] receives the frame count, eos_count_out[i] (may be NULL) the number of logits written. */
POCKET_TTS_API int pocket_tts_generate_ids(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                           const float* const* voices, const int* voice_frames,
                                           const float* const* noise, AudioResult* results, float* const* latents_out,
                                           float* const* eos_out, int* frames_out, int* eos_count_out);

/* The same through the slot table (queue with slot refill, any n): utterances finish and are replaced independently. */
POCKET_TTS_API int pocket_tts_generate_ids_queued(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                                  const float* const* voices, const int* voice_frames,
                                                  const float* const* noise, AudioResult* results, float* const* latents_out,
                                                  float* const* eos_out, int* frames_out, int* eos_count_out);

/* pocket_tts_generate_ids in three parts, for callers that time or overlap the device-resident middle:
 *   stage : host -> HBM copy of ids, conditioning rows and start points; returns a batch owned by the engine
 *           (valid until the next stage / generate call with the same n on this handle), or NULL;
 *   run   : prefill + frame loop + chunked decode; audio stays in HBM; frames_out[i] = frames of utterance i;
 *   fetch : HBM -> host; fills results[i] (release with pocket_tts_free_audio) and, if given, latents_out[i]. */
POCKET_TTS_API void* pocket_tts_stage_ids(PocketTTSHandle handle, int n, const long long* const* ids, const int* n_ids,
                                          const float* const* voices, const int* voice_frames, const float* const* noise);
POCKET_TTS_API int pocket_tts_run_staged(PocketTTSHandle handle, void* staged, int* frames_out);
POCKET_TTS_API int pocket_tts_fetch_staged(PocketTTSHandle handle, void* staged, const int* frames, AudioResult* results,
                                           float* const* latents_out);

/* Streaming from token ids; same schedule and callback contract as pocket_tts_generate_streaming.
 * Returns total samples or -1. */
POCKET_TTS_API int pocket_tts_streaming_ids(PocketTTSHandle handle, const long long* ids, int n_ids, const float* voice,
                                            int voice_frames, const float* noise, int chunk_size_frames,
                                            AudioChunkCallbackC callback, void* user_data);

/* Text pre-step alone: normalisation + SentencePiece ids (src/tokenizer.cpp:27-60).
 * Returns the id count (may exceed cap; only cap ids are written) or -1. */
POCKET_TTS_API int pocket_tts_tokenize(PocketTTSHandle handle, const char* text, long long* ids, int cap);

/* ---- one entry point per reference graph call ------------------------------------------------- */

/* text_conditioner: ids[T] -> out[T][d_model]                       (src/pocket_tts.cpp:299-320) */
POCKET_TTS_API int ptts_text_conditioner(PocketTTSHandle handle, const long long* ids, int T, float* out);

/* Recurrent state of `batch` lock-step utterances (LM KV caches + decoder state), zero-initialised
 * like the reference's initState (src/pocket_tts.cpp:133-186). */
POCKET_TTS_API void* ptts_state_new(PocketTTSHandle handle, int batch);
POCKET_TTS_API void ptts_state_free(PocketTTSHandle handle, void* state);
POCKET_TTS_API int ptts_state_reset(PocketTTSHandle handle, void* state);

/* flow_lm_main, prefill shape: sequence empty, text_embeddings = emb[rows][d_model] for one utterance
 * (src/pocket_tts.cpp:572,575). */
POCKET_TTS_API int ptts_lm_prefill(PocketTTSHandle handle, void* state, int slot, const float* emb, int rows);

/* flow_lm_main, decode shape: sequence[batch][latent_dim] (NaN row = BOS), empty text
 * -> cond[batch][d_model], eos[batch]                               (src/pocket_tts.cpp:591-593). */
POCKET_TTS_API int ptts_lm_step(PocketTTSHandle handle, void* state, const float* sequence, float* cond, float* eos);

/* flow_lm_flow iterated lsd_steps times with the Euler update, fused on the GPU:
 * cond[rows][d_model], x0[rows][latent_dim] -> x[rows][latent_dim]  (src/pocket_tts.cpp:414-451, 615-622). */
POCKET_TTS_API int ptts_flow_euler(PocketTTSHandle handle, const float* cond, const float* x0, int rows, float* x);

/* ptts_flow_euler for callers that cannot allocate device memory (code under HIP graph capture, serving loops and
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  cond, x0 and x are DEVICE pointers on the
 * handle's GPU (x must not alias x0); the scratch the kernels need (adaLN vectors, partial-sum slabs) is a static
 * device array of ptts_flow_euler_noalloc_capacity() bytes (32 MiB) that each GPU holds until the process ends.  The
 * call allocates and frees nothing, returns when x is complete, and fails with a message that names the limit when
 * `rows` needs more scratch than the array holds (full-size model at lsd_steps 10: 59 rows fit).  Calls on one GPU
 * share that GPU's array: the library serialises its own calls per GPU; do not run two processes' worth of
 * overlapping calls through other means. */
POCKET_TTS_API int ptts_flow_euler_noalloc(PocketTTSHandle handle, const float* cond_dev, const float* x0_dev, int rows, float* x_dev);
POCKET_TTS_API unsigned long long ptts_flow_euler_noalloc_capacity(void);

/* mimi_decoder on one chunk: latents[batch][n][latent_dim], n <= 15 -> audio[batch][n*samples_per_frame]
 * with the state carried in `state`                                 (src/pocket_tts.cpp:463-543). */
POCKET_TTS_API int ptts_mimi_decode(PocketTTSHandle handle, void* state, const float* latents, int n, float* audio);

/* mimi_encoder: audio[samples] at 24 kHz -> out[frames][d_model]; returns the frames the clip gives, or -1.  At most
 * cap_frames rows are written: a return value above cap_frames (>= 0) means the output was cut there
 *                                                                   (src/pocket_tts.cpp:261-291). */
POCKET_TTS_API int ptts_mimi_encode(PocketTTSHandle handle, const float* audio, int samples, float* out, int cap_frames);
/* The same over n_clips clips of `samples` samples each, stacked through the encoder together (the reference encodes one
 * clip per call, src/pocket_tts.cpp:241-297): outs[i] receives clip i's rows; returns the rows per clip (each outs[i] gets at
 * most cap_frames of them, as above). */
POCKET_TTS_API int ptts_mimi_encode_batch(PocketTTSHandle handle, const float* const* clips, int n_clips, int samples, float* const* outs,
                                          int cap_frames);

/* ---- host-side pre/post steps, usable without a GPU ---------------------------------------------- */

/* Text pre-step on its own (src/tokenizer.cpp:14-60).  raw != 0 skips the sentence clean-up. */
POCKET_TTS_API void* ptts_tokenizer_new(const char* model_path);
POCKET_TTS_API void ptts_tokenizer_free(void* tokenizer);
POCKET_TTS_API int ptts_tokenizer_vocab(void* tokenizer);
POCKET_TTS_API int ptts_tokenizer_encode(void* tokenizer, const char* text, int raw, long long* ids, int cap);

/* AudioUtils of the reference (src/audio_utils.cpp:45-303).  Each returns the output length (writing at
 * most cap samples) or -1. */
POCKET_TTS_API int ptts_audio_load_wav(const char* path, int target_rate, float* out, int cap);
POCKET_TTS_API int ptts_audio_save_wav(const char* path, const float* samples, int n, int rate);
POCKET_TTS_API int ptts_audio_resample(const float* in, int n, int in_rate, int out_rate, float* out, int cap);
POCKET_TTS_API int ptts_audio_normalize(const float* in, int n, float* out);

/* Benchmark support: HIP-event timing on the streams the engine launches on.  Both arrays hold 4 entries:
 * [0] LM frame step (kernel chain), [1] flow phase (adaLN GEMMs + fused head), [2] decoder chunk,
 * [3] the flow head's launch chain alone (lsd_steps * (blocks + 1) launches per frame, counted in calls[3]);
 * times_ms = totals, calls = number of launches of each. */
/* The int8 matrix product of the "int8" precision on its own: out[M][N] = A[M][K] . W[N][K]^T with W quantised per
 * output row to int8 and A per row and per block of 64 k (K % 64 == 0, N % 16 == 0).  For checking the kernel's
 * arithmetic exactly; the model paths call the same kernel. */
POCKET_TTS_API int ptts_q8_linear(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, float* out);
/* The same with the matrix given in int8 form, q[N][K] and one scale per output row (what a pack's `<name>.q8` carries). */
POCKET_TTS_API int ptts_q8_linear_q(PocketTTSHandle handle, const float* A, int M, int K, const signed char* q, const float* scale, int N,
                                    float* out);

/* The same product in the int8 engine's second activation scheme, the one ONNX Runtime's dynamically quantised graphs
 * use (DynamicQuantizeLinear + MatMulInteger; the reference's *_int8.onnx models, src/pocket_tts.cpp:232-246): each row
 * of A becomes uint8 with one scale and one zero point over the whole row (range widened to include 0), W is int8 per
 * output row, the integer sum runs over all of K and is scaled once. */
POCKET_TTS_API int ptts_q8_linear_dyn(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, float* out);

/* The bf16-weight matrix product on its own: out[M][N] = act(A . W^T + bias) * colscale + R (bias, colscale, R may be
 * NULL; act: 0 none, 1 GELU, 2 SiLU, 3 ELU; W is rounded to bf16).  flags bit 0: carry A as two bf16 terms; bit 1: keep
 * the A rows in bf16 in device memory; bit 2: keep the C rows in bf16.  For checking the kernels at arbitrary shapes; the
 * model paths go through the same dispatch. */
POCKET_TTS_API int ptts_bf16_linear(PocketTTSHandle handle, const float* A, int M, int K, const float* W, int N, const float* bias,
                                    const float* colscale, const float* R, int act, int flags, float* out);

POCKET_TTS_API int ptts_phase_timing(PocketTTSHandle handle, int enable);
/* The slot table that serves pocket_tts_generate, pocket_tts_generate_streaming, pocket_tts_generate_batch and the _queued calls:
 * stats[0] = activations of the table (idle -> busy), stats[1] = utterances served, stats[2] = most slots occupied at once. */
POCKET_TTS_API int ptts_scheduler_stats(PocketTTSHandle handle, long* stats);
/* ... and stats[3] = frame steps launched, [4] = sum of the launch width over them, [5] = frames delivered to callers,
 * [6] = decoder chunk calls, [7] = prefill passes (8 entries). */
POCKET_TTS_API int ptts_scheduler_stats_ex(PocketTTSHandle handle, long* stats);
/* Engines behind the handle: one per GPU listed in the environment variable POCKET_TTS_DEVICES (e.g. "0,1,2,3") when the
 * handle was created, else one.  served[i] (i < cap, may be NULL) receives the utterances engine i has served.  Single
 * utterances and streams go to the engine with the fewest utterances pending, queued batches are dealt out over all. */
POCKET_TTS_API int ptts_engine_count(PocketTTSHandle handle, long* served, int cap);
POCKET_TTS_API int ptts_phase_times(PocketTTSHandle handle, double* times_ms, long* calls);
/* bytes[0..3] = GEMM weight bytes (compute precision) of the LM, the flow head, the decoder, the encoder */
POCKET_TTS_API int ptts_weight_bytes(PocketTTSHandle handle, unsigned long long* bytes);

/* Weights carried by an ONNX file: graph initializers and Constant-node tensors, read at the protobuf wire level
 * (no onnx / protobuf dependency).  The reference gives its five .onnx graphs to ONNX Runtime
 * (src/pocket_tts.cpp:90-113); a converter (pocket-tts.cpp_amd/tools/onnx_to_ptts.py) uses these calls to build the
 * engine's weight pack from them.  dtype values are onnx.TensorProto.DataType (1 float, 2 uint8, 3 int8, 6 int32,
 * 7 int64, 10 float16, 11 double, 16 bfloat16).  read_f32 converts any of those to float; dequantize returns
 * (q - zero_point) * scale for the <base>_quantized / <base>_scale / <base>_zero_point triplets that ONNX Runtime's
 * quantisation tools write (axis -1: the one axis whose extent equals the number of scales).  Both return the
 * element count or -1.  Host only: usable without a GPU. */
POCKET_TTS_API void* ptts_onnx_open(const char* path);
POCKET_TTS_API void ptts_onnx_close(void* onnx);
POCKET_TTS_API int ptts_onnx_tensor_count(void* onnx);
POCKET_TTS_API int ptts_onnx_tensor_info(void* onnx, int index, const char** name, int* dtype, int* rank, long long* dims8);
POCKET_TTS_API int ptts_onnx_find(void* onnx, const char* name);
POCKET_TTS_API long long ptts_onnx_read_f32(void* onnx, int index, float* out, long long cap);
POCKET_TTS_API long long ptts_onnx_dequantize(void* onnx, const char* base, int axis, float* out, long long cap, int* rank, long long* dims8);
POCKET_TTS_API int ptts_onnx_node_count(void* onnx);
POCKET_TTS_API int ptts_onnx_node_info(void* onnx, int index, char* op_type, char* inputs, char* outputs, int cap);

#ifdef __cplusplus
}
#endif

#endif /* POCKET_TTS_AMD_H */
