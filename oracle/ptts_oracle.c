/*
 * ptts_oracle.c -- CPU restatement of the Pocket TTS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the HIP engine is compared against.  Only tests/, the smoke check and
 * the cpu_baseline leg of bench.py may load it; the product library never links or calls it.
 *
 * What it restates:
 *   - orchestration of the reference, line for line in behaviour:
 *       offline generate      /root/reference/src/pocket_tts.cpp:548-657
 *       streaming generate    /root/reference/src/pocket_tts.cpp:686-979
 *       flow time grid        /root/reference/src/pocket_tts.cpp:122-130
 *       Euler update          /root/reference/src/pocket_tts.cpp:615-622
 *       EOS / stop rule       /root/reference/src/pocket_tts.cpp:596-603
 *       15-frame decode chunks/root/reference/src/pocket_tts.cpp:461-463
 *       5 s voice truncation  /root/reference/src/pocket_tts.cpp:256-259
 *   - the five graphs the reference runs through ONNX Runtime (text_conditioner, flow_lm_main,
 *     flow_lm_flow, mimi_decoder, mimi_encoder; call sites src/pocket_tts.cpp:269,308,381,442,518).
 *     The graphs themselves (KevinAHM/pocket-tts-onnx, an export of kyutai/pocket-tts) and ONNX
 *     Runtime 1.17 are external downloads that are absent offline, and the reference holds no
 *     golden vectors for them.  The network arithmetic below therefore follows the published
 *     architecture (causal pre-norm transformer with RoPE; adaLN flow MLP; SEANet/Mimi codec) with
 *     every dimension read from the weight pack.
 *
 * PARITY UNPINNED for the network arithmetic: no reference output exists to pin it against.  It is
 * cross-checked against an independent float64 PyTorch transcription (tests/test_oracle_vs_torch.py).
 * The orchestration contract (tensor names, shapes, loop order, constants) is pinned by the
 * reference source cited above.
 *
 * Plain C99 + OpenMP.  All arithmetic is float32 with explicit 16-lane partial sums in the dot
 * products, so gcc can vectorise without reassociation flags.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PTTS_MAX_TENSORS 512

typedef struct {
    char name[64];
    uint32_t dtype, ndim;
    uint64_t shape[4], offset, nbytes;
} PackEntry;

typedef struct {
    int latent, d_model, lm_layers, lm_heads, lm_ffn, lm_max_period, vocab;
    int flow_dim, flow_blocks, flow_freq;
    int mimi_dim, mimi_layers, mimi_heads, mimi_ffn, mimi_context, mimi_max_period;
    int n_filters, ratios[3], kernel, last_kernel, res_kernel, upsample, sample_rate, q8_act_scheme;
    float lm_eps, flow_eps, mimi_eps, flow_max_period;
    int frame_size;
} Spec;

typedef struct Oracle {
    Spec s;
    int n_tensors;
    PackEntry *entries;
    unsigned char *blob;
    size_t blob_size;
} Oracle;

/* ------------------------------------------------------------------ pack */

static const float *tensor(const Oracle *o, const char *name) {
    for (int i = 0; i < o->n_tensors; ++i)
        if (strncmp(o->entries[i].name, name, 64) == 0)
            return (const float *)(o->blob + o->entries[i].offset);
    fprintf(stderr, "ptts_oracle: missing tensor %s\n", name);
    abort();
}

static const float *tensorf(const Oracle *o, const char *fmt, const char *prefix, int idx, const char *suffix) {
    char buf[96];
    snprintf(buf, sizeof buf, fmt, prefix, idx, suffix);
    return tensor(o, buf);
}

Oracle *ptts_oracle_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *blob = (unsigned char *)malloc((size_t)sz);
    if (!blob || fread(blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(blob); return NULL; }
    fclose(f);
    if (sz < 336 || memcmp(blob, "PTTSPK01", 8) != 0) { free(blob); return NULL; }
    uint32_t ver, n;
    memcpy(&ver, blob + 8, 4);
    memcpy(&n, blob + 12, 4);
    if (ver != 1 || n > PTTS_MAX_TENSORS) { free(blob); return NULL; }
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    o->blob = blob;
    o->blob_size = (size_t)sz;
    o->n_tensors = (int)n;
    o->entries = (PackEntry *)(blob + 336);
    const int32_t *iv = (const int32_t *)(blob + 16);
    const float *fv = (const float *)(blob + 16 + 256);
    Spec *s = &o->s;
    s->latent = iv[0]; s->d_model = iv[1]; s->lm_layers = iv[2]; s->lm_heads = iv[3]; s->lm_ffn = iv[4];
    s->lm_max_period = iv[5]; s->vocab = iv[6]; s->flow_dim = iv[7]; s->flow_blocks = iv[8]; s->flow_freq = iv[9];
    s->mimi_dim = iv[10]; s->mimi_layers = iv[11]; s->mimi_heads = iv[12]; s->mimi_ffn = iv[13];
    s->mimi_context = iv[14]; s->mimi_max_period = iv[15]; s->n_filters = iv[16];
    s->ratios[0] = iv[17]; s->ratios[1] = iv[18]; s->ratios[2] = iv[19];
    s->kernel = iv[20]; s->last_kernel = iv[21]; s->res_kernel = iv[22]; s->upsample = iv[23]; s->sample_rate = iv[24];
    s->q8_act_scheme = iv[25];
    s->lm_eps = fv[0]; s->flow_eps = fv[1]; s->mimi_eps = fv[2]; s->flow_max_period = fv[3];
    s->frame_size = s->upsample * s->ratios[0] * s->ratios[1] * s->ratios[2];
    return o;
}

static void q8_forget(const unsigned char *lo, const unsigned char *hi);

void ptts_oracle_free(Oracle *o) {
    if (!o) return;
    q8_forget(o->blob, o->blob + o->blob_size);
    free(o->blob);
    free(o);
}

/* dims[0..5] = latent, d_model, frame_size, vocab, mimi_dim, sample_rate */
void ptts_oracle_dims(const Oracle *o, int *dims) {
    dims[0] = o->s.latent; dims[1] = o->s.d_model; dims[2] = o->s.frame_size;
    dims[3] = o->s.vocab; dims[4] = o->s.mimi_dim; dims[5] = o->s.sample_rate;
}

/* ------------------------------------------------------------------ math helpers */

static void q8_register(const float *W, int N, int K);

/* Switches the selected families of this oracle's LM-layer and flow-block matrices to the int8 scheme described at
 * linear_q8.  mask bits: 1 qkv, 2 out, 4 ff1, 8 ff2 (LM layers), 16 fc1, 32 fc2 (flow blocks). */
void ptts_oracle_enable_q8_some(Oracle *o, int mask) {
    const Spec *s = &o->s;
    const char *lm[4] = {"qkv.w", "out.w", "ff1.w", "ff2.w"};
    const int lmN[4] = {3 * s->d_model, s->d_model, s->lm_ffn, s->d_model};
    const int lmK[4] = {s->d_model, s->d_model, s->d_model, s->lm_ffn};
    for (int l = 0; l < s->lm_layers; ++l)
        for (int j = 0; j < 4; ++j)
            if (mask & (1 << j)) q8_register(tensorf(o, "%s%d.%s", "lm.l", l, lm[j]), lmN[j], lmK[j]);
    for (int b = 0; b < s->flow_blocks && s->flow_dim % 256 == 0; ++b) {   /* narrower heads keep bf16 blocks */
        if (mask & 16) q8_register(tensorf(o, "%s%d.%s", "flow.b", b, "fc1.w"), s->flow_dim, s->flow_dim);
        if (mask & 32) q8_register(tensorf(o, "%s%d.%s", "flow.b", b, "fc2.w"), s->flow_dim, s->flow_dim);
    }
}

/* Activation scheme the next registered int8 matrices get (each registry entry keeps its own, so oracles of both schemes
 * can be alive at once): 0 = per row and block of 64, symmetric (the default); 1 = ONNX DynamicQuantizeLinear per row,
 * uint8 with a zero point (see linear_q8_dyn). */
static int g_q8_act = 0;
void ptts_oracle_set_q8_act(int scheme) { g_q8_act = scheme; }

/* The int8 engine's model: the LM layers' matrices in int8 and, with flow_blocks (the engine's POCKET_TTS_Q8_FLOW=1; per-block
 * scheme only), the flow blocks' too.  The activation scheme is taken from the pack header. */
void ptts_oracle_enable_q8_ex(Oracle *o, int flow_blocks) {
    const int saved = g_q8_act;
    g_q8_act = o->s.q8_act_scheme;
    ptts_oracle_enable_q8_some(o, (flow_blocks && g_q8_act == 0) ? 63 : 15);
    g_q8_act = saved;
}
void ptts_oracle_enable_q8(Oracle *o) { ptts_oracle_enable_q8_ex(o, 0); }

/* Attribution knob (tools/int8_attribution.py): which side of an int8 product is quantised.  3 = both (the engine's
 * scheme, the default), 1 = weights only (activations stay float32), 2 = activations only (weights stay float32). */
static int g_q8_parts = 3;
void ptts_oracle_set_q8_parts(int parts) { g_q8_parts = parts; }

static float dotf(const float *a, const float *b, int n) {
    float acc[16] = {0};
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j) acc[j] += a[i + j] * b[i + j];
    float tail = 0.f;
    for (; i < n; ++i) tail += a[i] * b[i];
    float s8[8], s4[4];
    for (int j = 0; j < 8; ++j) s8[j] = acc[j] + acc[j + 8];
    for (int j = 0; j < 4; ++j) s4[j] = s8[j] + s8[j + 4];
    return ((s4[0] + s4[2]) + (s4[1] + s4[3])) + tail;
}

/* ------------------------------------------------------------------ int8 mode (the engine's "int8" precision)
 *
 * The reference's int8 graphs (flow_lm_main_int8.onnx, flow_lm_flow_int8.onnx, src/pocket_tts.cpp:90-96) are
 * ONNX Runtime quantised models whose exact scheme cannot be read offline; the scheme restated here is the one
 * the HIP engine implements, so the two can be compared exactly:
 *   weights      per output channel n: sw[n] = max|W[n][:]| / 127, qw = rint(W * 127 / max) in [-127, 127]
 *   activations  per row and per block of 64 consecutive k: s = max|x|, qx = rint(x * (127 / s)), d = s / 127
 *   product      y[n] = (sum over blocks of (float)(sum_k qx*qw) * d) * sw[n] + bias[n], blocks in k order
 * Applied to the LM layers' qkv/out/ff1/ff2 and (when flow_dim is a multiple of 256) the flow head's fc1/fc2
 * matrices; every other matrix keeps its bf16-rounded weights.  A quantised matrix is recognised by its address in the weight blob. */
typedef struct { const float *W; int N, K, act; int8_t *q; float *sw; } Q8Entry;
static Q8Entry g_q8[128];
static int g_nq8 = 0;

static const Q8Entry *q8_find(const float *W) {
    for (int i = 0; i < g_nq8; ++i)
        if (g_q8[i].W == W) return &g_q8[i];
    return NULL;
}

static void q8_register(const float *W, int N, int K) {
    if (q8_find(W) || K % 64) return;
    if (g_nq8 >= 128) {   /* a matrix silently left in float would look like a parity error of the engine */
        fprintf(stderr, "ptts_oracle: more than 128 int8 matrices registered at once\n");
        abort();
    }
    Q8Entry *e = &g_q8[g_nq8];
    e->W = W; e->N = N; e->K = K; e->act = g_q8_act;
    e->q = (int8_t *)malloc((size_t)N * K);
    e->sw = (float *)malloc((size_t)N * sizeof(float));
    for (int n = 0; n < N; ++n) {
        float mx = 0.f;
        for (int k = 0; k < K; ++k) { float a = fabsf(W[(size_t)n * K + k]); if (a > mx) mx = a; }
        const float inv = mx > 0.f ? 127.0f / mx : 0.f;
        e->sw[n] = mx / 127.0f;
        for (int k = 0; k < K; ++k) {
            float r = rintf(W[(size_t)n * K + k] * inv);
            if (r > 127.f) r = 127.f;
            if (r < -127.f) r = -127.f;
            e->q[(size_t)n * K + k] = (int8_t)r;
        }
    }
    ++g_nq8;
}

/* drop the entries that point into a blob about to be freed */
static void q8_forget(const unsigned char *lo, const unsigned char *hi) {
    int w = 0;
    for (int i = 0; i < g_nq8; ++i) {
        const unsigned char *p = (const unsigned char *)g_q8[i].W;
        if (p >= lo && p < hi) { free(g_q8[i].q); free(g_q8[i].sw); continue; }
        g_q8[w++] = g_q8[i];
    }
    g_nq8 = w;
}

static float dotf(const float *a, const float *b, int n);

/* ONNX DynamicQuantizeLinear + MatMulInteger, the activation scheme of a dynamically quantised ONNX graph (the
 * reference's *_int8.onnx files, /root/reference/src/pocket_tts.cpp:90-96; operator definitions: ONNX opset 11).  The
 * reference runs one utterance per call, so a decode step's activation tensor is one row; the rows of a longer pass are
 * quantised one by one here as in the engine (ONNX Runtime would use one scale for the whole [T][K] tensor of a prefill call).
 *   lo = min(0, min x), hi = max(0, max x), s = (hi - lo) / 255, zp = rint(clamp(-lo / s, 0, 255))
 *   q = clamp(rint(x / s) + zp, 0, 255)                         (rint: half to even; s = 0: q = zp = 0)
 *   y[n] = (float)(sum_k (q[k] - zp) * qw[n][k]) * (s * sw[n]) + bias[n]          (integer sum exact in int32)          */
static void linear_q8_dyn(float *Y, int ldy, const float *X, int ldx, const Q8Entry *e, const float *bias, int M) {
    const int N = e->N, K = e->K;
#pragma omp parallel for schedule(static) if ((long)M * N * K > 400000)
    for (int m = 0; m < M; ++m) {
        const float *x = X + (long)m * ldx;
        int *q = (int *)malloc((size_t)K * sizeof(int));
        float lo = 0.f, hi = 0.f;
        for (int k = 0; k < K; ++k) { if (x[k] < lo) lo = x[k]; if (x[k] > hi) hi = x[k]; }
        float s = (hi - lo) / 255.0f;
        int zp = 0;
        if (s > 0.f) {
            float z = -lo / s;
            if (z < 0.f) z = 0.f;
            if (z > 255.f) z = 255.f;
            zp = (int)rintf(z);
            for (int k = 0; k < K; ++k) {
                int v = (int)rintf(x[k] / s) + zp;
                q[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        } else {
            s = 0.f;
            for (int k = 0; k < K; ++k) q[k] = 0;
        }
        if (g_q8_parts == 2) {   /* activations only: the de-quantised row against the float weights */
            float *xd = (float *)malloc((size_t)K * sizeof(float));
            for (int k = 0; k < K; ++k) xd[k] = (float)(q[k] - zp) * s;
            for (int n = 0; n < N; ++n) Y[(long)m * ldy + n] = dotf(xd, e->W + (size_t)n * K, K) + (bias ? bias[n] : 0.f);
            free(xd);
            free(q);
            continue;
        }
        for (int n = 0; n < N; ++n) {
            const int8_t *qw = e->q + (size_t)n * K;
            int isum = 0;
            for (int k = 0; k < K; ++k) isum += (q[k] - zp) * (int)qw[k];
            Y[(long)m * ldy + n] = (float)isum * (s * e->sw[n]) + (bias ? bias[n] : 0.f);
        }
        free(q);
    }
}

static void linear_q8(float *Y, int ldy, const float *X, int ldx, const Q8Entry *e, const float *bias, int M) {
    const int N = e->N, K = e->K, KB = K / 64;
    if (e->act == 1 && g_q8_parts != 1) { linear_q8_dyn(Y, ldy, X, ldx, e, bias, M); return; }
    if (g_q8_parts == 1) {   /* weights only: the float dot product with the de-quantised row */
#pragma omp parallel for schedule(static) if ((long)M * N * K > 400000)
        for (int n = 0; n < N; ++n) {
            float *w = (float *)malloc((size_t)K * sizeof(float));
            for (int k = 0; k < K; ++k) w[k] = (float)e->q[(size_t)n * K + k] * e->sw[n];
            for (int m = 0; m < M; ++m) Y[(long)m * ldy + n] = dotf(X + (long)m * ldx, w, K) + (bias ? bias[n] : 0.f);
            free(w);
        }
        return;
    }
#pragma omp parallel for schedule(static) if ((long)M * N * K > 400000)
    for (int m = 0; m < M; ++m) {
        int8_t *qx = (int8_t *)malloc((size_t)K);
        float *d = (float *)malloc((size_t)KB * sizeof(float));
        const float *x = X + (long)m * ldx;
        for (int kb = 0; kb < KB; ++kb) {
            float s = 0.f;
            for (int k = 0; k < 64; ++k) { float a = fabsf(x[kb * 64 + k]); if (a > s) s = a; }
            const float inv = s > 0.f ? 127.0f / s : 0.f;
            d[kb] = s / 127.0f;
            for (int k = 0; k < 64; ++k) {
                float r = rintf(x[kb * 64 + k] * inv);
                if (r > 127.f) r = 127.f;
                if (r < -127.f) r = -127.f;
                qx[kb * 64 + k] = (int8_t)r;
            }
        }
        for (int n = 0; n < N && g_q8_parts == 2; ++n) {   /* activations only: quantised row against the float32 weights */
            const float *wf = e->W + (size_t)n * K;
            float acc = 0.f;
            for (int kb = 0; kb < KB; ++kb) {
                float fs = 0.f;
                for (int k = 0; k < 64; ++k) fs += (float)qx[kb * 64 + k] * wf[kb * 64 + k];
                acc += fs * d[kb];
            }
            Y[(long)m * ldy + n] = acc + (bias ? bias[n] : 0.f);
        }
        for (int n = 0; n < N && g_q8_parts != 2; ++n) {
            const int8_t *qw = e->q + (size_t)n * K;
            float acc = 0.f;
            for (int kb = 0; kb < KB; ++kb) {
                int isum = 0;
                for (int k = 0; k < 64; ++k) isum += (int)qx[kb * 64 + k] * (int)qw[kb * 64 + k];
                acc += (float)isum * d[kb];
            }
            Y[(long)m * ldy + n] = acc * e->sw[n] + (bias ? bias[n] : 0.f);
        }
        free(qx); free(d);
    }
}

/* Y[m][n] = bias[n] + sum_k X[m*ldx + k] * W[n][k]   (rows of X may overlap: ldx < K is allowed) */
static void linear(float *Y, int ldy, const float *X, int ldx, const float *W, const float *bias,
                   int M, int N, int K) {
    const Q8Entry *qe = g_nq8 ? q8_find(W) : NULL;
    if (qe) { linear_q8(Y, ldy, X, ldx, qe, bias, M); return; }
    long work = (long)M * N * K;
#pragma omp parallel for collapse(2) schedule(static) if (work > 400000)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n)
            Y[(long)m * ldy + n] = dotf(X + (long)m * ldx, W + (long)n * K, K) + (bias ? bias[n] : 0.f);
}

static void layernorm(float *y, const float *x, const float *g, const float *b, int n, float eps) {
    float mean = 0.f;
    for (int i = 0; i < n; ++i) mean += x[i];
    mean /= (float)n;
    float var = 0.f;
    for (int i = 0; i < n; ++i) { float d = x[i] - mean; var += d * d; }
    var /= (float)n;
    float r = 1.0f / sqrtf(var + eps);
    for (int i = 0; i < n; ++i) {
        float v = (x[i] - mean) * r;
        if (g) v = v * g[i] + b[i];
        y[i] = v;
    }
}

static inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static inline float silu_f(float x) { return x / (1.0f + expf(-x)); }
static inline float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

/* RoPE on interleaved pairs (2i, 2i+1) of one head vector; angle = pos * max_period^(-2i/dh). */
static void rope(float *v, int dh, long pos, int max_period) {
    for (int i = 0; i < dh / 2; ++i) {
        double freq = exp(-log((double)max_period) * 2.0 * (double)i / (double)dh);
        double ang = (double)pos * freq;
        float c = (float)cos(ang), s = (float)sin(ang);
        float a = v[2 * i], b = v[2 * i + 1];
        v[2 * i] = a * c - b * s;
        v[2 * i + 1] = a * s + b * c;
    }
}

/* One int8 product on its own, for the operator-level tests: W [N][K] float is quantised per row as the engine does
 * (q8_register), A [M][K] per the scheme (0 block, 1 dynamic uint8).  The registry entry is dropped afterwards. */
void ptts_oracle_q8_linear(const float *A, int M, int K, const float *W, int N, int scheme, float *out) {
    const int saved = g_q8_act;
    g_q8_act = scheme;
    q8_register(W, N, K);
    g_q8_act = saved;
    const Q8Entry *e = q8_find(W);
    if (!e) { memset(out, 0, (size_t)M * N * sizeof(float)); return; }
    linear_q8(out, N, A, K, e, NULL, M);
    q8_forget((const unsigned char *)W, (const unsigned char *)W + 1);
}

/* ------------------------------------------------------------------ transformer (shared by LM and Mimi) */

typedef struct {
    int layers, d, heads, cap, context; /* context <= 0: unlimited causal */
    long pos;                           /* absolute position of the next token */
    float *k, *v;                       /* [layers][cap][d]; ring indexed by pos % cap */
} KvState;

static KvState *kv_new(int layers, int d, int heads, int cap, int context) {
    KvState *st = (KvState *)calloc(1, sizeof(KvState));
    st->layers = layers; st->d = d; st->heads = heads; st->cap = cap; st->context = context;
    st->k = (float *)calloc((size_t)layers * cap * d, sizeof(float));
    st->v = (float *)calloc((size_t)layers * cap * d, sizeof(float));
    return st;
}
static void kv_free(KvState *st) { if (st) { free(st->k); free(st->v); free(st); } }

/* Attribution knob (tools/bf16_attribution.py): with the flag set, the LM's K and V rows are rounded to bfloat16
 * (nearest even) as they enter the cache, which is what the bf16 and int8 engines store.  Off by default. */
static int g_lm_kv_bf16 = 0;
void ptts_oracle_set_lm_kv_bf16(int on) { g_lm_kv_bf16 = on; }
static inline float bf16_round_f(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

/* One pre-norm transformer stack over R new rows X[R][d] (in place).
 * prefix selects "lm" or "dec"/"enc" tensors; layer scale vectors are optional (Mimi only). */
static void transformer_rows(const Oracle *o, const char *prefix, KvState *st, float *X, int R,
                             int ffn, int max_period, float eps, int has_ls) {
    int d = st->d, H = st->heads, dh = d / H;
    float *h = (float *)malloc((size_t)R * d * sizeof(float));
    float *qkv = (float *)malloc((size_t)R * 3 * d * sizeof(float));
    float *att = (float *)malloc((size_t)R * d * sizeof(float));
    float *ff = (float *)malloc((size_t)R * ffn * sizeof(float));
    float *tmp = (float *)malloc((size_t)R * d * sizeof(float));
    float scale = 1.0f / sqrtf((float)dh);
    for (int l = 0; l < st->layers; ++l) {
        const float *g1 = tensorf(o, "%s%d.%s", prefix, l, "ln1.g"), *b1 = tensorf(o, "%s%d.%s", prefix, l, "ln1.b");
        const float *g2 = tensorf(o, "%s%d.%s", prefix, l, "ln2.g"), *b2 = tensorf(o, "%s%d.%s", prefix, l, "ln2.b");
        const float *wqkv = tensorf(o, "%s%d.%s", prefix, l, "qkv.w"), *wo = tensorf(o, "%s%d.%s", prefix, l, "out.w");
        const float *w1 = tensorf(o, "%s%d.%s", prefix, l, "ff1.w"), *w2 = tensorf(o, "%s%d.%s", prefix, l, "ff2.w");
        const float *ls1 = has_ls ? tensorf(o, "%s%d.%s", prefix, l, "ls1") : NULL;
        const float *ls2 = has_ls ? tensorf(o, "%s%d.%s", prefix, l, "ls2") : NULL;
        float *kc = st->k + (size_t)l * st->cap * d, *vc = st->v + (size_t)l * st->cap * d;
        for (int r = 0; r < R; ++r) layernorm(h + (size_t)r * d, X + (size_t)r * d, g1, b1, d, eps);
        linear(qkv, 3 * d, h, d, wqkv, NULL, R, 3 * d, d);
        for (int r = 0; r < R; ++r) {
            long p = st->pos + r;
            float *q = qkv + (size_t)r * 3 * d, *k = q + d, *v = q + 2 * d;
            for (int hh = 0; hh < H; ++hh) { rope(q + hh * dh, dh, p, max_period); rope(k + hh * dh, dh, p, max_period); }
            if (g_lm_kv_bf16 && !has_ls) for (int e = 0; e < d; ++e) { k[e] = bf16_round_f(k[e]); v[e] = bf16_round_f(v[e]); }
            memcpy(kc + (size_t)(p % st->cap) * d, k, (size_t)d * sizeof(float));
            memcpy(vc + (size_t)(p % st->cap) * d, v, (size_t)d * sizeof(float));
        }
#pragma omp parallel for collapse(2) schedule(static) if ((long)R * H > 8)
        for (int r = 0; r < R; ++r) {
            for (int hh = 0; hh < H; ++hh) {
                long p = st->pos + r;
                long j0 = 0;
                if (st->context > 0 && p - st->context + 1 > 0) j0 = p - st->context + 1;
                int nk = (int)(p - j0 + 1);
                float *sc = (float *)malloc((size_t)nk * sizeof(float));
                const float *q = qkv + (size_t)r * 3 * d + hh * dh;
                float mx = -INFINITY;
                for (int j = 0; j < nk; ++j) {
                    sc[j] = dotf(q, kc + (size_t)((j0 + j) % st->cap) * d + hh * dh, dh) * scale;
                    if (sc[j] > mx) mx = sc[j];
                }
                float sum = 0.f;
                for (int j = 0; j < nk; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                float inv = 1.0f / sum;
                float *out = att + (size_t)r * d + hh * dh;
                for (int e = 0; e < dh; ++e) out[e] = 0.f;
                for (int j = 0; j < nk; ++j) {
                    const float *vv = vc + (size_t)((j0 + j) % st->cap) * d + hh * dh;
                    float pj = sc[j] * inv;
                    for (int e = 0; e < dh; ++e) out[e] += pj * vv[e];
                }
                free(sc);
            }
        }
        linear(tmp, d, att, d, wo, NULL, R, d, d);
        for (size_t i = 0; i < (size_t)R * d; ++i) X[i] += (ls1 ? ls1[i % d] : 1.0f) * tmp[i];
        for (int r = 0; r < R; ++r) layernorm(h + (size_t)r * d, X + (size_t)r * d, g2, b2, d, eps);
        linear(ff, ffn, h, d, w1, NULL, R, ffn, d);
        for (size_t i = 0; i < (size_t)R * ffn; ++i) ff[i] = gelu_f(ff[i]);
        linear(tmp, d, ff, ffn, w2, NULL, R, d, ffn);
        for (size_t i = 0; i < (size_t)R * d; ++i) X[i] += (ls2 ? ls2[i % d] : 1.0f) * tmp[i];
    }
    st->pos += R;
    free(h); free(qkv); free(att); free(ff); free(tmp);
}

/* ------------------------------------------------------------------ graph: text_conditioner */

/* token_ids i64[T] -> embeddings f32[T][d_model]   (src/pocket_tts.cpp:299-320) */
void ptts_oracle_text_conditioner(const Oracle *o, const int64_t *ids, int T, float *out) {
    const float *E = tensor(o, "text.embed");
    int D = o->s.d_model;
    for (int t = 0; t < T; ++t) {
        int64_t id = ids[t];
        if (id < 0 || id > o->s.vocab) id = o->s.vocab;
        memcpy(out + (size_t)t * D, E + (size_t)id * D, (size_t)D * sizeof(float));
    }
}

/* ------------------------------------------------------------------ graph: flow_lm_main */

typedef struct { KvState *kv; } LmState;

LmState *ptts_oracle_lm_state_new(const Oracle *o, int max_ctx) {
    LmState *st = (LmState *)calloc(1, sizeof(LmState));
    st->kv = kv_new(o->s.lm_layers, o->s.d_model, o->s.lm_heads, max_ctx, 0);
    return st;
}
void ptts_oracle_lm_state_free(LmState *st) { if (st) { kv_free(st->kv); free(st); } }
long ptts_oracle_lm_state_pos(const LmState *st) { return st->kv->pos; }

/* sequence f32[S][latent] (NaN rows = BOS), text_embeddings f32[T][d_model], recurrent state
 * -> conditioning f32[d_model] and EOS logit of the last sequence row (src/pocket_tts.cpp:323-411).
 * With S == 0 (the two prefill calls, :572,:575) cond/eos are left untouched. */
int ptts_oracle_lm_main(const Oracle *o, LmState *st, const float *seq, int S, const float *text, int T,
                        float *cond, float *eos) {
    const Spec *s = &o->s;
    int D = s->d_model, L = s->latent, R = T + S;
    if (R == 0) return 0;
    if (st->kv->pos + R > st->kv->cap) return -1;
    float *X = (float *)malloc((size_t)R * D * sizeof(float));
    if (T) memcpy(X, text, (size_t)T * D * sizeof(float));
    if (S) {
        const float *bos = tensor(o, "lm.bos_emb"), *win = tensor(o, "lm.in_proj.w");
        float *z = (float *)malloc((size_t)S * L * sizeof(float));
        for (int i = 0; i < S * L; ++i) z[i] = isnan(seq[i]) ? bos[i % L] : seq[i];
        linear(X + (size_t)T * D, D, z, L, win, NULL, S, D, L);
        free(z);
    }
    transformer_rows(o, "lm.l", st->kv, X, R, s->lm_ffn, s->lm_max_period, s->lm_eps, 0);
    if (S) {
        float *last = (float *)malloc((size_t)D * sizeof(float));
        layernorm(last, X + (size_t)(R - 1) * D, tensor(o, "lm.out_norm.g"), tensor(o, "lm.out_norm.b"), D, s->lm_eps);
        memcpy(cond, last, (size_t)D * sizeof(float));
        *eos = dotf(last, tensor(o, "lm.eos.w"), D) + tensor(o, "lm.eos.b")[0];
        free(last);
    }
    free(X);
    return 0;
}

/* ------------------------------------------------------------------ graph: flow_lm_flow */

static void time_embed(const Oracle *o, int which, float tval, float *out /*[C]*/) {
    const Spec *s = &o->s;
    int C = s->flow_dim, Q = s->flow_freq, half = Q / 2;
    float *fe = (float *)malloc((size_t)Q * sizeof(float));
    float *h = (float *)malloc((size_t)C * sizeof(float));
    for (int i = 0; i < half; ++i) {
        double freq = exp(-log((double)s->flow_max_period) * (double)i / (double)half);
        double a = (double)tval * freq;
        fe[i] = (float)cos(a);
        fe[half + i] = (float)sin(a);
    }
    char n1[64], n2[64], n3[64], n4[64];
    snprintf(n1, 64, "flow.time%d.l1.w", which); snprintf(n2, 64, "flow.time%d.l1.b", which);
    snprintf(n3, 64, "flow.time%d.l2.w", which); snprintf(n4, 64, "flow.time%d.l2.b", which);
    linear(h, C, fe, Q, tensor(o, n1), tensor(o, n2), 1, C, Q);
    for (int i = 0; i < C; ++i) h[i] = silu_f(h[i]);
    linear(out, C, h, C, tensor(o, n3), tensor(o, n4), 1, C, C);
    free(fe); free(h);
}

/* c f32[d_model], s, t scalars, x f32[latent] -> flow_dir f32[latent]  (src/pocket_tts.cpp:414-451) */
void ptts_oracle_flow(const Oracle *o, const float *c, float sv, float tv, const float *x, float *v) {
    const Spec *s = &o->s;
    int C = s->flow_dim, L = s->latent, D = s->d_model;
    float *h = (float *)malloc((size_t)C * sizeof(float)), *y = (float *)malloc((size_t)C * sizeof(float));
    float *ts = (float *)malloc((size_t)C * sizeof(float)), *tt = (float *)malloc((size_t)C * sizeof(float));
    float *sy = (float *)malloc((size_t)C * sizeof(float)), *mod = (float *)malloc((size_t)3 * C * sizeof(float));
    float *n = (float *)malloc((size_t)C * sizeof(float)), *a = (float *)malloc((size_t)C * sizeof(float));
    float *b = (float *)malloc((size_t)C * sizeof(float));
    linear(h, C, x, L, tensor(o, "flow.in_proj.w"), tensor(o, "flow.in_proj.b"), 1, C, L);
    time_embed(o, 0, sv, ts);
    time_embed(o, 1, tv, tt);
    linear(y, C, c, D, tensor(o, "flow.cond.w"), tensor(o, "flow.cond.b"), 1, C, D);
    for (int i = 0; i < C; ++i) { y[i] = (ts[i] + tt[i]) * 0.5f + y[i]; sy[i] = silu_f(y[i]); }
    for (int blk = 0; blk < s->flow_blocks; ++blk) {
        linear(mod, 3 * C, sy, C, tensorf(o, "%s%d.%s", "flow.b", blk, "ada.w"), tensorf(o, "%s%d.%s", "flow.b", blk, "ada.b"), 1, 3 * C, C);
        const float *shift = mod, *scale = mod + C, *gate = mod + 2 * C;
        layernorm(n, h, tensorf(o, "%s%d.%s", "flow.b", blk, "ln.g"), tensorf(o, "%s%d.%s", "flow.b", blk, "ln.b"), C, s->flow_eps);
        for (int i = 0; i < C; ++i) n[i] = n[i] * (1.0f + scale[i]) + shift[i];
        linear(a, C, n, C, tensorf(o, "%s%d.%s", "flow.b", blk, "fc1.w"), tensorf(o, "%s%d.%s", "flow.b", blk, "fc1.b"), 1, C, C);
        for (int i = 0; i < C; ++i) a[i] = silu_f(a[i]);
        linear(b, C, a, C, tensorf(o, "%s%d.%s", "flow.b", blk, "fc2.w"), tensorf(o, "%s%d.%s", "flow.b", blk, "fc2.b"), 1, C, C);
        for (int i = 0; i < C; ++i) h[i] += gate[i] * b[i];
    }
    linear(mod, 2 * C, sy, C, tensor(o, "flow.final.ada.w"), tensor(o, "flow.final.ada.b"), 1, 2 * C, C);
    layernorm(n, h, NULL, NULL, C, s->flow_eps);
    for (int i = 0; i < C; ++i) n[i] = n[i] * (1.0f + mod[C + i]) + mod[i];
    linear(v, L, n, C, tensor(o, "flow.final.lin.w"), tensor(o, "flow.final.lin.b"), 1, L, C);
    free(h); free(y); free(ts); free(tt); free(sy); free(mod); free(n); free(a); free(b);
}

/* lsd_steps Euler iterations from x (in place): s_j = j/N, t_j = s_j + 1/N, x += v * dt
 * (src/pocket_tts.cpp:122-130, 582, 615-622). */
void ptts_oracle_flow_euler(const Oracle *o, const float *c, float *x, int lsd_steps) {
    int L = o->s.latent;
    float dt = 1.0f / (float)lsd_steps;
    float *v = (float *)malloc((size_t)L * sizeof(float));
    for (int j = 0; j < lsd_steps; ++j) {
        float sv = (float)j / (float)lsd_steps;
        float tv = sv + dt;
        ptts_oracle_flow(o, c, sv, tv, x, v);
        for (int k = 0; k < L; ++k) x[k] += v[k] * dt;
    }
    free(v);
}

/* ------------------------------------------------------------------ graph: mimi_decoder (streaming) */

typedef struct {
    KvState *kv;
    float *up_prev;       /* [M] previous projected frame for the depthwise transposed conv */
    float *conv0_hist;    /* [(kernel-1)][M] */
    float *tr_partial[3]; /* transposed-conv overlap tails [r][Cout] (without bias) */
    float *res_hist[3];   /* [(res_kernel-1)][Cout] ELU'd inputs of the residual conv */
    float *final_hist;    /* [(last_kernel-1)][n_filters] ELU'd */
} DecState;

DecState *ptts_oracle_dec_state_new(const Oracle *o) {
    const Spec *s = &o->s;
    DecState *st = (DecState *)calloc(1, sizeof(DecState));
    int M = s->mimi_dim;
    int cap = s->mimi_context + 16 * s->upsample + 8;
    st->kv = kv_new(s->mimi_layers, M, s->mimi_heads, cap, s->mimi_context);
    st->up_prev = (float *)calloc((size_t)M, sizeof(float));
    st->conv0_hist = (float *)calloc((size_t)(s->kernel - 1) * M, sizeof(float));
    int c = 8 * s->n_filters;
    for (int i = 0; i < 3; ++i) {
        int co = c / 2;
        st->tr_partial[i] = (float *)calloc((size_t)s->ratios[i] * co, sizeof(float));
        st->res_hist[i] = (float *)calloc((size_t)(s->res_kernel - 1) * co, sizeof(float));
        c = co;
    }
    st->final_hist = (float *)calloc((size_t)(s->last_kernel - 1) * c, sizeof(float));
    return st;
}
void ptts_oracle_dec_state_free(DecState *st) {
    if (!st) return;
    kv_free(st->kv); free(st->up_prev); free(st->conv0_hist);
    for (int i = 0; i < 3; ++i) { free(st->tr_partial[i]); free(st->res_hist[i]); }
    free(st->final_hist); free(st);
}

/* Causal stride-1 conv over channels-last rows with carried history.
 * in[T][Cin]; hist[(k-1)][Cin] is updated to the last k-1 input rows; out[T][Cout]. */
static void conv_stream(float *out, const float *in, float *hist, const float *W, const float *bias,
                        int T, int Cin, int Cout, int k) {
    int H = k - 1;
    float *buf = (float *)malloc((size_t)(H + T) * Cin * sizeof(float));
    memcpy(buf, hist, (size_t)H * Cin * sizeof(float));
    memcpy(buf + (size_t)H * Cin, in, (size_t)T * Cin * sizeof(float));
    linear(out, Cout, buf, Cin, W, bias, T, Cout, k * Cin);
    memcpy(hist, buf + (size_t)T * Cin, (size_t)H * Cin * sizeof(float));
    free(buf);
}

/* Streaming transposed conv, kernel 2r stride r, by overlap-add with a carried tail.
 * Weight layout [r*Cout][2*Cin] as documented in the pack; out[T*r][Cout]. */
static void convtr_stream(float *out, const float *in, float *partial, const float *W, const float *bias,
                          int T, int Cin, int Cout, int r) {
    float *acc = (float *)calloc((size_t)(T + 1) * r * Cout, sizeof(float));
    memcpy(acc, partial, (size_t)r * Cout * sizeof(float));
    float *cur = (float *)malloc((size_t)r * Cout * sizeof(float));
    float *nxt = (float *)malloc((size_t)r * Cout * sizeof(float));
    for (int p = 0; p < T; ++p) {
        const float *x = in + (size_t)p * Cin;
#pragma omp parallel for schedule(static) if ((long)r * Cout * Cin > 200000)
        for (int n = 0; n < r * Cout; ++n) {
            cur[n] = dotf(x, W + (size_t)n * 2 * Cin + Cin, Cin); /* taps 0..r-1  -> outputs p*r + phase       */
            nxt[n] = dotf(x, W + (size_t)n * 2 * Cin, Cin);       /* taps r..2r-1 -> outputs (p+1)*r + phase   */
        }
        float *a0 = acc + (size_t)p * r * Cout, *a1 = acc + (size_t)(p + 1) * r * Cout;
        for (int n = 0; n < r * Cout; ++n) { a0[n] += cur[n]; a1[n] += nxt[n]; }
    }
    for (size_t i = 0; i < (size_t)T * r * Cout; ++i) out[i] = acc[i] + bias[i % Cout];
    memcpy(partial, acc + (size_t)T * r * Cout, (size_t)r * Cout * sizeof(float));
    free(acc); free(cur); free(nxt);
}

/* latent f32[n][latent] + state -> audio f32[n*frame_size]  (src/pocket_tts.cpp:454-546) */
void ptts_oracle_mimi_decode(const Oracle *o, DecState *st, const float *latent, int n, float *audio) {
    const Spec *s = &o->s;
    int L = s->latent, M = s->mimi_dim, U = s->upsample;
    const float *mean = tensor(o, "mimi.emb_mean"), *std = tensor(o, "mimi.emb_std");
    float *z = (float *)malloc((size_t)n * L * sizeof(float));
    for (int i = 0; i < n * L; ++i) z[i] = latent[i] * std[i % L] + mean[i % L];
    float *q = (float *)malloc((size_t)n * M * sizeof(float));
    linear(q, M, z, L, tensor(o, "dec.in_proj.w"), NULL, n, M, L);
    /* depthwise transposed conv: y[p*U + r][c] = q[p][c]*w[c][r] + q[p-1][c]*w[c][r+U] */
    int T = n * U;
    const float *wu = tensor(o, "dec.up.w");
    float *y = (float *)malloc((size_t)T * M * sizeof(float));
    for (int p = 0; p < n; ++p) {
        const float *cur = q + (size_t)p * M, *prev = p ? q + (size_t)(p - 1) * M : st->up_prev;
        for (int r = 0; r < U; ++r)
            for (int c = 0; c < M; ++c)
                y[((size_t)p * U + r) * M + c] = cur[c] * wu[(size_t)c * 2 * U + r] + prev[c] * wu[(size_t)c * 2 * U + r + U];
    }
    memcpy(st->up_prev, q + (size_t)(n - 1) * M, (size_t)M * sizeof(float));
    transformer_rows(o, "dec.t", st->kv, y, T, s->mimi_ffn, s->mimi_max_period, s->mimi_eps, 1);
    /* SEANet decoder */
    int c = 8 * s->n_filters;
    float *x = (float *)malloc((size_t)T * c * sizeof(float));
    conv_stream(x, y, st->conv0_hist, tensor(o, "dec.conv0.w"), tensor(o, "dec.conv0.b"), T, M, c, s->kernel);
    for (int si = 0; si < 3; ++si) {
        int r = s->ratios[si], co = c / 2, T2 = T * r;
        for (size_t i = 0; i < (size_t)T * c; ++i) x[i] = elu_f(x[i]);
        float *u = (float *)malloc((size_t)T2 * co * sizeof(float));
        convtr_stream(u, x, st->tr_partial[si], tensorf(o, "%s%d.%s", "dec.s", si, "up.w"),
                      tensorf(o, "%s%d.%s", "dec.s", si, "up.b"), T, c, co, r);
        float *e = (float *)malloc((size_t)T2 * co * sizeof(float));
        for (size_t i = 0; i < (size_t)T2 * co; ++i) e[i] = elu_f(u[i]);
        float *h1 = (float *)malloc((size_t)T2 * (co / 2) * sizeof(float));
        conv_stream(h1, e, st->res_hist[si], tensorf(o, "%s%d.%s", "dec.s", si, "res.c1.w"),
                    tensorf(o, "%s%d.%s", "dec.s", si, "res.c1.b"), T2, co, co / 2, s->res_kernel);
        for (size_t i = 0; i < (size_t)T2 * (co / 2); ++i) h1[i] = elu_f(h1[i]);
        float *h2 = (float *)malloc((size_t)T2 * co * sizeof(float));
        linear(h2, co, h1, co / 2, tensorf(o, "%s%d.%s", "dec.s", si, "res.c2.w"),
               tensorf(o, "%s%d.%s", "dec.s", si, "res.c2.b"), T2, co, co / 2);
        for (size_t i = 0; i < (size_t)T2 * co; ++i) u[i] += h2[i];
        free(x); free(e); free(h1); free(h2);
        x = u; c = co; T = T2;
    }
    for (size_t i = 0; i < (size_t)T * c; ++i) x[i] = elu_f(x[i]);
    conv_stream(audio, x, st->final_hist, tensor(o, "dec.final.w"), tensor(o, "dec.final.b"), T, c, 1, s->last_kernel);
    free(x); free(y); free(q); free(z);
}

/* ------------------------------------------------------------------ graph: mimi_encoder (whole clip) */

/* Causal strided conv over a whole clip: left pad k-stride zeros, right pad to a stride multiple.
 * in[T][Cin] -> out[ceil(T/stride)][Cout]; returns the output length. */
static int conv_clip(float **out, const float *in, const float *W, const float *bias, int T, int Cin, int Cout,
                     int k, int stride) {
    int To = (T + stride - 1) / stride;
    int padl = k - stride;
    size_t rows = (size_t)To * stride + padl;
    float *buf = (float *)calloc(rows * Cin, sizeof(float));
    memcpy(buf + (size_t)padl * Cin, in, (size_t)T * Cin * sizeof(float));
    *out = (float *)malloc((size_t)To * Cout * sizeof(float));
    linear(*out, Cout, buf, stride * Cin, W, bias, To, Cout, k * Cin);
    free(buf);
    return To;
}

/* Number of conditioning frames the encoder yields for S samples. */
int ptts_oracle_mimi_encode_frames(const Oracle *o, int S) {
    const Spec *s = &o->s;
    int T = S;
    for (int i = 2; i >= 0; --i) T = (T + s->ratios[i] - 1) / s->ratios[i];
    return (T + s->upsample - 1) / s->upsample;
}

/* audio f32[S] -> latents f32[Nv][d_model]  (src/pocket_tts.cpp:261-291); returns Nv. */
int ptts_oracle_mimi_encode(const Oracle *o, const float *audio, int S, float *out) {
    const Spec *s = &o->s;
    int M = s->mimi_dim, D = s->d_model;
    int c = s->n_filters, T = S;
    float *x = NULL;
    T = conv_clip(&x, audio, tensor(o, "enc.conv0.w"), tensor(o, "enc.conv0.b"), T, 1, c, s->kernel, 1);
    for (int si = 0; si < 3; ++si) {
        int r = s->ratios[2 - si];
        float *e = (float *)malloc((size_t)T * c * sizeof(float));
        for (size_t i = 0; i < (size_t)T * c; ++i) e[i] = elu_f(x[i]);
        float *h1 = NULL;
        conv_clip(&h1, e, tensorf(o, "%s%d.%s", "enc.s", si, "res.c1.w"), tensorf(o, "%s%d.%s", "enc.s", si, "res.c1.b"),
                  T, c, c / 2, s->res_kernel, 1);
        for (size_t i = 0; i < (size_t)T * (c / 2); ++i) h1[i] = elu_f(h1[i]);
        float *h2 = (float *)malloc((size_t)T * c * sizeof(float));
        linear(h2, c, h1, c / 2, tensorf(o, "%s%d.%s", "enc.s", si, "res.c2.w"), tensorf(o, "%s%d.%s", "enc.s", si, "res.c2.b"), T, c, c / 2);
        for (size_t i = 0; i < (size_t)T * c; ++i) e[i] = elu_f(x[i] + h2[i]);
        float *d = NULL;
        int T2 = conv_clip(&d, e, tensorf(o, "%s%d.%s", "enc.s", si, "down.w"), tensorf(o, "%s%d.%s", "enc.s", si, "down.b"),
                           T, c, 2 * c, 2 * r, r);
        free(x); free(e); free(h1); free(h2);
        x = d; T = T2; c *= 2;
    }
    for (size_t i = 0; i < (size_t)T * c; ++i) x[i] = elu_f(x[i]);
    float *y = NULL;
    conv_clip(&y, x, tensor(o, "enc.final.w"), tensor(o, "enc.final.b"), T, c, M, s->last_kernel, 1);
    free(x);
    KvState *kv = kv_new(s->mimi_layers, M, s->mimi_heads, T > 1 ? T : 1, s->mimi_context);
    transformer_rows(o, "enc.t", kv, y, T, s->mimi_ffn, s->mimi_max_period, s->mimi_eps, 1);
    kv_free(kv);
    float *d = NULL;
    int Nv = conv_clip(&d, y, tensor(o, "enc.down.w"), NULL, T, M, M, 2 * s->upsample, s->upsample);
    free(y);
    linear(out, D, d, M, tensor(o, "enc.spk_proj.w"), NULL, Nv, D, M);
    free(d);
    return Nv;
}

/* ------------------------------------------------------------------ orchestration */

typedef void (*ptts_oracle_chunk_cb)(const float *samples, int count, int is_final, void *user);

/* Decode all latents in chunks of <= 15 frames with one carried state (src/pocket_tts.cpp:454-546). */
static void decode_all(const Oracle *o, DecState *st, const float *latents, int F, float *audio) {
    int L = o->s.latent, fs = o->s.frame_size;
    for (int i = 0; i < F; i += 15) {
        int n = F - i < 15 ? F - i : 15;
        ptts_oracle_mimi_decode(o, st, latents + (size_t)i * L, n, audio + (size_t)i * fs);
    }
}

/* Shared autoregressive body.  noise: NULL (temperature 0: x starts at zeros, :606-613) or
 * f32[max_frames][latent] start points already scaled by sqrt(temperature).
 * chunk_frames <= 0: offline (decode after the loop, :640).  Otherwise the streaming schedule
 * (:790-888 flush every chunk_frames, :897-965 tail flush with is_final=1).
 * Outputs (each may be NULL): latents[F][latent], eos[F_steps], audio[F*frame_size].
 * Returns the number of frames generated. */
int ptts_oracle_generate(const Oracle *o, const int64_t *ids, int T, const float *voice, int Nv,
                         const float *noise, int lsd_steps, int max_frames, int frames_after_eos,
                         int chunk_frames, ptts_oracle_chunk_cb cb, void *user,
                         float *latents_out, float *eos_out, float *audio_out, int *eos_steps_run) {
    const Spec *s = &o->s;
    int D = s->d_model, L = s->latent, fs = s->frame_size;
    float *text = (float *)malloc((size_t)(T > 0 ? T : 1) * D * sizeof(float));
    ptts_oracle_text_conditioner(o, ids, T, text);                       /* :559 / :706 */
    LmState *lm = ptts_oracle_lm_state_new(o, Nv + T + max_frames + 1);   /* :563 / :710 */
    float dummy_c, dummy_e;
    ptts_oracle_lm_main(o, lm, NULL, 0, voice, Nv, &dummy_c, &dummy_e);   /* voice prefill :572 / :719 */
    ptts_oracle_lm_main(o, lm, NULL, 0, text, T, &dummy_c, &dummy_e);     /* text prefill  :575 / :725 */
    DecState *dec = ptts_oracle_dec_state_new(o);                         /* :458 / :737 */
    float *lat = (float *)malloc((size_t)max_frames * L * sizeof(float));
    float *cur = (float *)malloc((size_t)L * sizeof(float));
    float *cond = (float *)malloc((size_t)D * sizeof(float));
    float *chunk_audio = (float *)malloc((size_t)(chunk_frames > 0 ? chunk_frames : 1) * fs * sizeof(float));
    for (int k = 0; k < L; ++k) cur[k] = NAN;                             /* BOS marker :579 / :729 */
    int eos_step = -1, F = 0, pending = 0, steps = 0;
    for (int step = 0; step < max_frames; ++step) {
        float eos;
        ptts_oracle_lm_main(o, lm, cur, 1, NULL, 0, cond, &eos);          /* :591 / :753 */
        if (eos_out) eos_out[step] = eos;
        ++steps;
        if (eos > -4.0f && eos_step < 0) eos_step = step;                 /* :596 */
        if (eos_step >= 0 && step >= eos_step + frames_after_eos) break;  /* :601 */
        float *x = lat + (size_t)F * L;
        for (int k = 0; k < L; ++k) x[k] = noise ? noise[(size_t)step * L + k] : 0.f;
        ptts_oracle_flow_euler(o, cond, x, lsd_steps);                    /* :615-622 */
        memcpy(cur, x, (size_t)L * sizeof(float));                        /* :624-625 */
        ++F; ++pending;
        if (chunk_frames > 0 && pending >= chunk_frames) {                /* :790 (second clause unreachable) */
            decode_all(o, dec, lat + (size_t)(F - pending) * L, pending, chunk_audio);
            if (audio_out) memcpy(audio_out + (size_t)(F - pending) * fs, chunk_audio, (size_t)pending * fs * sizeof(float));
            if (cb) cb(chunk_audio, pending * fs, 0, user);               /* :879-880 */
            pending = 0;
        }
    }
    if (chunk_frames > 0) {
        if (pending > 0) {                                                /* :897-965 */
            decode_all(o, dec, lat + (size_t)(F - pending) * L, pending, chunk_audio);
            if (audio_out) memcpy(audio_out + (size_t)(F - pending) * fs, chunk_audio, (size_t)pending * fs * sizeof(float));
            if (cb) cb(chunk_audio, pending * fs, 1, user);
        }
    } else if (audio_out && F > 0) {
        decode_all(o, dec, lat, F, audio_out);                            /* :640 */
    }
    if (latents_out) memcpy(latents_out, lat, (size_t)F * L * sizeof(float));
    if (eos_steps_run) *eos_steps_run = steps;
    free(text); free(lat); free(cur); free(cond); free(chunk_audio);
    ptts_oracle_lm_state_free(lm);
    ptts_oracle_dec_state_free(dec);
    return F;
}
