/*
 * Pocket TTS C ABI -- MI355X (gfx950) engine.
 *
 * This header is the drop-in boundary: libpocket_tts.so built from this repository exports the same
 * eleven symbols, struct layouts and error conventions as gudrob/pocket-tts.cpp, so existing C,
 * ctypes, P/Invoke and N-API callers bind to it unchanged.  Each declaration below names the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Behavioural contract shared by all entry points:
 *   - nothing throws across this boundary; failures return NULL / -1 and leave a message readable
 *     through pocket_tts_get_last_error() on the calling thread;
 *   - handles are plain heap objects owned by the caller until the matching destroy/free call;
 *   - calls block until the work is done; streaming callbacks run on the calling thread;
 *   - one handle may be used from several threads at once: concurrent pocket_tts_generate and
 *     pocket_tts_generate_streaming calls share the GPU (their utterances step side by side in one slot
 *     table); pocket_tts_cancel_streaming ends the streams that are running when it is called.
 * There is no CPU fallback: creating a handle fails when no HIP device is present.
 */
#ifndef POCKET_TTS_C_H
#define POCKET_TTS_C_H

#if defined(_WIN32)
#  if defined(POCKET_TTS_STATIC)
#    define POCKET_TTS_API
#  elif defined(POCKET_TTS_BUILDING_DLL)
#    define POCKET_TTS_API __declspec(dllexport)
#  else
#    define POCKET_TTS_API __declspec(dllimport)
#  endif
#else
#  define POCKET_TTS_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef void* PocketTTSHandle; /* engine instance            (include/pocket_tts/pocket_tts_c.h:22) */
typedef void* VoiceHandle;     /* encoded voice, reusable    (include/pocket_tts/pocket_tts_c.h:23) */

/* Engine options, 40 bytes on LP64 (include/pocket_tts/pocket_tts_c.h:26-33).
 * A field is applied only when it is non-NULL / greater than zero; otherwise the default holds
 * (src/pocket_tts_c.cpp:33-40).  models_dir must contain the weight pack pocket_tts.ptts. */
typedef struct {
    const char* models_dir;     /* default "models/onnx"            */
    const char* tokenizer_path; /* default "models/tokenizer.model" */
    const char* precision;      /* "int8" (default) or "fp32"; this engine also accepts "bf16" */
    float temperature;          /* default 0.7; values <= 0 select the default */
    int lsd_steps;              /* Euler steps of the flow head, default 10    */
    int max_frames;             /* generation cap in 80 ms frames, default 500 */
} PocketTTSConfig;

/* Generated audio, 16 bytes on LP64 (include/pocket_tts/pocket_tts_c.h:36-40). */
typedef struct {
    float* data;      /* mono float PCM, owned by the caller until pocket_tts_free_audio */
    int sample_count;
    int sample_rate;  /* 24000 */
} AudioResult;

/* Streaming chunk sink (include/pocket_tts/pocket_tts_c.h:115-120).  `samples` is valid only for the
 * duration of the call.  is_final is 1 only on the tail chunk that flushes the frames left over after
 * generation stops (src/pocket_tts.cpp:897-965); when the frame count is a multiple of the chunk size nothing is left
 * over and no chunk carries is_final = 1, as in the reference. */
typedef void (*AudioChunkCallbackC)(const float* samples, int sample_count, int is_final, void* user_data);

/* Streaming options, 16 bytes on LP64 (include/pocket_tts/pocket_tts_c.h:125-128). */
typedef struct {
    int chunk_size_frames; /* decode and deliver every N frames; <= 0 selects 5 (src/pocket_tts_c.cpp:188-191) */
    void* user_data;
} StreamingConfig;

/* Replaces pocket_tts_create (include/pocket_tts/pocket_tts_c.h:48, src/pocket_tts_c.cpp:29-51).
 * NULL config selects every default.  Returns NULL on failure. */
POCKET_TTS_API PocketTTSHandle pocket_tts_create(const PocketTTSConfig* config);

/* Replaces pocket_tts_destroy (pocket_tts_c.h:53, src/pocket_tts_c.cpp:53-57).  NULL is ignored. */
POCKET_TTS_API void pocket_tts_destroy(PocketTTSHandle handle);

/* Replaces pocket_tts_encode_voice (pocket_tts_c.h:63, src/pocket_tts_c.cpp:59-82): WAV file ->
 * resample to 24 kHz -> attenuate to peak 0.85 -> first 5 s -> Mimi encoder on the GPU. */
POCKET_TTS_API VoiceHandle pocket_tts_encode_voice(PocketTTSHandle handle, const char* audio_path);

/* Replaces pocket_tts_encode_voice_from_samples (pocket_tts_c.h:74-79, src/pocket_tts_c.cpp:84-112).
 * Mono float samples at sample_rate.  The buffer is encoded directly (no temporary file). */
POCKET_TTS_API VoiceHandle pocket_tts_encode_voice_from_samples(PocketTTSHandle handle, const float* audio_data,
                                                                int sample_count, int sample_rate);

/* Replaces pocket_tts_free_voice (pocket_tts_c.h:84, src/pocket_tts_c.cpp:114-118). */
POCKET_TTS_API void pocket_tts_free_voice(VoiceHandle voice);

/* Replaces pocket_tts_generate (pocket_tts_c.h:95-100, src/pocket_tts_c.cpp:120-149).
 * Returns 0 on success and fills *result; -1 on failure. */
POCKET_TTS_API int pocket_tts_generate(PocketTTSHandle handle, const char* text, VoiceHandle voice, AudioResult* result);

/* Replaces pocket_tts_free_audio (pocket_tts_c.h:105, src/pocket_tts_c.cpp:151-157): releases
 * result->data, then sets data = NULL and sample_count = 0. */
POCKET_TTS_API void pocket_tts_free_audio(AudioResult* result);

/* Replaces pocket_tts_generate_streaming (pocket_tts_c.h:141-147, src/pocket_tts_c.cpp:163-208).
 * Returns the total number of samples delivered, or -1 on failure.  Cancellation is always armed. */
POCKET_TTS_API int pocket_tts_generate_streaming(PocketTTSHandle handle, const char* text, VoiceHandle voice,
                                                 AudioChunkCallbackC callback, const StreamingConfig* config);

/* Replaces pocket_tts_cancel_streaming (pocket_tts_c.h:155, src/pocket_tts_c.cpp:210-221).
 * Safe to call from another thread while a streaming call is running. */
POCKET_TTS_API void pocket_tts_cancel_streaming(PocketTTSHandle handle);

/* Replaces pocket_tts_get_last_error (pocket_tts_c.h:161, src/pocket_tts_c.cpp:159-161):
 * thread-local message of the most recent failing call on this thread. */
POCKET_TTS_API const char* pocket_tts_get_last_error(void);

/* Replaces pocket_tts_version (pocket_tts_c.h:166, src/pocket_tts_c.cpp:223-225): "1.0.0". */
POCKET_TTS_API const char* pocket_tts_version(void);

#ifdef __cplusplus
}
#endif

#endif /* POCKET_TTS_C_H */
