/*
 * include/kws.h -- C ABI of the MI355X-native keyword-spotting hot path.
 *
 * The reference (david8862/tf-keras-speech-commands) has no FFI of its own: its
 * hot path is Python calling sonopy and tf.keras.  This header is the boundary
 * a maintainer would bind from that Python with ctypes;
 * every entry point names the reference interface it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no torch / C++ types; all array arguments are caller-owned
 *     DEVICE pointers (row-major, float32 unless stated); `stream` is a
 *     hipStream_t passed as void* (NULL = default stream);
 *   - every int-returning function returns KWS_OK (0) or a negative kws_status;
 *     kws_last_error() gives the message for the calling thread;
 *   - work is only enqueued on `stream`; nothing synchronises the device;
 *   - there is no CPU fallback: without a HIP device the create/launch calls
 *     fail with KWS_ERR_HIP.
 */
#ifndef KWS_H
#define KWS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kws_status {
    KWS_OK = 0,
    KWS_ERR_INVALID = -1,     /* bad argument / inconsistent params              */
    KWS_ERR_UNSUPPORTED = -2, /* valid in the reference, not built here yet      */
    KWS_ERR_HIP = -3,         /* HIP runtime error (no device, launch failure)   */
    KWS_ERR_NOMEM = -4,
    KWS_ERR_WORKSPACE = -5,   /* caller's workspace too small                    */
    KWS_ERR_COMM = -6         /* RCCL missing or a collective failed             */
} kws_status;

const char *kws_version(void);
/* "<source file>:<sha1 prefix>;..." of every file the library was built from (measurement records in profiles/ are tied to it) */
const char *kws_build_id(void);
const char *kws_last_error(void);
/* number of visible HIP devices (0 when none / no driver); never fails */
int kws_device_count(void);

/* Opt-in per-kernel timing: while enabled every kernel launched through this library is bracketed by HIP
 * events on its launch stream.  kws_prof_enable(1) clears and starts, (0) stops and clears;
 * kws_prof_report synchronises the recorded events and writes a JSON object
 * {"<kernel>": {"count": n, "total_ms": t}, ...} into buf; returns the bytes needed (incl. NUL). */
int kws_prof_enable(int on);
int64_t kws_prof_report(char *buf, size_t buflen);

/* ------------------------------------------------------------------------
 * Audio-pipeline parameters: the numeric fields of ListenerParams
 * (classifier/params.py:49-59) as read from params.json (configs/params.json).
 * ---------------------------------------------------------------------- */
typedef struct kws_params {
    double buffer_t, window_t, hop_t;
    int32_t sample_rate, sample_depth, n_fft, n_filt, n_mfcc, use_delta;
} kws_params;

/* derived properties, classifier/params.py:59-91 */
typedef struct kws_geometry {
    int32_t window_samples, hop_samples, max_samples, buffer_samples, n_features, feature_size;
} kws_geometry;

/* defaults of classifier/params.py:99-103 */
void kws_params_default(kws_params *p);
/* host-only; KWS_ERR_INVALID if the parameters are unusable */
int kws_params_derive(const kws_params *p, kws_geometry *g);

/* ------------------------------------------------------------------------
 * Featurizer: replaces common/data_utils.py:73-86 audio_to_feature (and with
 * it vectorize_raw :61-70 = sonopy.mfcc_spec, and add_deltas :50-58), batched.
 * KWS_BANK_BARK swaps in the filterbank of common/bark_feature.py:92-136
 * (bfcc_spec :156-175).
 * ---------------------------------------------------------------------- */
typedef enum kws_bank_kind { KWS_BANK_MEL = 0, KWS_BANK_BARK = 1 } kws_bank_kind;
typedef enum kws_wav_dtype {
    KWS_WAV_F32 = 0, /* float32 in [-1,1), what librosa.load returns (data_utils.py:93) */
    KWS_WAV_I16 = 1  /* raw little-endian PCM16, scaled by 1/32768 (data_utils.py:21)     */
} kws_wav_dtype;

typedef struct kws_featurizer kws_featurizer;

int kws_featurizer_create(const kws_params *p, int bank_kind, kws_featurizer **out);
void kws_featurizer_destroy(kws_featurizer *f);
/* geometry the featurizer was built for */
int kws_featurizer_geometry(const kws_featurizer *f, kws_geometry *g);
/* host copy of the dense (n_filt x (n_fft/2+1)) bank the sparse tables were built from */
int kws_featurizer_bank(const kws_featurizer *f, float *host_bank, size_t count);

/*
 * wav      : (B, stride) samples of `wav_dtype`; row b holds clip b from its first sample
 * valid_len: NULL (every clip has `stride` samples) or B DEVICE int32 clip lengths
 *            (0 <= len <= stride).  Per clip the reference's contract applies: keep
 *            the FIRST max_samples (data_utils.py:77), LEFT-pad with zeros when
 *            shorter (:79-80).
 * feat     : (B, n_features, feature_size) float32
 */
int kws_featurize(kws_featurizer *f, const void *wav, int wav_dtype, int B, int64_t stride,
                  const int32_t *valid_len, float *feat, void *stream);
/* The same for B clips GATHERED from a device-resident dataset: clip b is row index[b] (device int32) of `wav` (rows x stride) and of
 * `valid_len` (one length per ROW).  This is the minibatch draw of model.fit(shuffle=True) (train.py:81-92) without a copy of the
 * audio: the reference's fit gathers feature rows on the host; here a step's 4096 x 64 KB of samples are read in place. */
int kws_featurize_gather(kws_featurizer *f, const void *wav, int wav_dtype, const int32_t *index, int B, int64_t stride,
                         const int32_t *valid_len, float *feat, void *stream);

/*
 * vectorize_raw (common/data_utils.py:61-70): audio of exactly n_samples per clip, no length
 * clipping, no padding, no deltas.  feat: (B, n_frames, n_mfcc) with
 * n_frames = kws_featurize_raw_frames(f, n_samples) = (n_samples - window)/hop + 1 (0 if shorter).
 */
int kws_featurize_raw_frames(const kws_featurizer *f, int32_t n_samples);

int kws_featurize_raw(kws_featurizer *f, const void *wav, int wav_dtype, int B, int64_t stride, int32_t n_samples,
                      float *feat, void *stream);

/* How much of every compute unit one launch of the tuned (default-geometry) kernel may hold.  2 (default): two persistent blocks of
 * 8 waves per CU (4 waves per SIMD, 2 x 52 KB of LDS), fastest when the featurizer has the chip to itself (inference, dataset
 * featurization); 1: ONE block of 12 waves (75 KB of LDS), so that kernels of OTHER streams still find wave slots and LDS on every
 * CU -- use it when a batch is featurized beside a running train step (kws_train_args.overlap_event).  Same arithmetic, same bits.
 * Other geometries (generic kernels) ignore it. */
int kws_featurizer_set_cu_share(kws_featurizer *f, int blocks_per_cu);

/* Diagnostics: resident blocks (= clips) per compute unit the runtime reports for the float32 featurizer kernel at this
 * featurizer's LDS size, and that LDS size in bytes.  No reference counterpart; used by tools/ and DESIGN.md. */
int kws_featurizer_occupancy(const kws_featurizer *f, int *blocks_per_cu, size_t *lds_bytes);

/* ------------------------------------------------------------------------
 * Model: replaces the tf.keras objects built by classifier/model.py:14-46
 * get_model() (backbones classifier/models/cnn.py, rnn.py) and the work
 * Keras does inside model.fit / model.predict (train.py:75-92).
 *
 * Weights live in two caller-owned flat float32 device buffers:
 *   params : the trainable tensors           (kws_model_param_count floats)
 *   state  : BatchNormalization moving stats (kws_model_state_count floats)
 * kws_model_tensor_info() enumerates the tensors in Keras get_weights() order
 * with their offset into params (trainable) or state (not trainable); offsets
 * are multiples of 4 floats, gaps are zero.  grads / Adam moments mirror params.
 * ---------------------------------------------------------------------- */
typedef enum kws_model_kind {
    KWS_SIMPLE_CNN = 0,      /* classifier/models/cnn.py:11-74  */
    KWS_SIMPLE_CNN_LITE = 1, /* classifier/models/cnn.py:77-141 */
    KWS_SIMPLE_GRU = 2,      /* classifier/models/rnn.py:10-43  */
    KWS_SIMPLE_LSTM = 3      /* classifier/models/rnn.py:46-79  */
} kws_model_kind;

typedef struct kws_model kws_model;

typedef struct kws_tensor_info {
    char name[64];      /* e.g. "conv2d/kernel", "batch_normalization/gamma", "score_predict/bias" */
    int32_t ndim;
    int32_t shape[4];   /* Keras shapes: conv HWIO, dense (in, out) */
    int32_t trainable;  /* 1: offset is into params, 0: into state */
    int64_t offset;     /* in floats */
    int64_t size;       /* in floats */
} kws_tensor_info;

/* host only (no GPU needed).  kind outside the enum -> KWS_ERR_INVALID "Unsupported model type"
 * (classifier/model.py:32). n_features / feature_size: classifier/params.py:66-68,86-91. */
int kws_model_create(int kind, int num_classes, int n_features, int feature_size, kws_model **out);
void kws_model_destroy(kws_model *m);
int64_t kws_model_param_count(const kws_model *m);
int64_t kws_model_state_count(const kws_model *m);
int kws_model_num_tensors(const kws_model *m);
int kws_model_tensor_info(const kws_model *m, int index, kws_tensor_info *out);
/* Creates the model's per-device resources (its side stream and fork / join events) on the CURRENT device now instead of at the first
 * train step.  HIP deals streams to a small number of hardware queues in creation order; when the side stream lands on the queue of
 * the caller's stream the step's fork / join overlap is lost (measured: 0.65 -> 1.2 ms per step when an RCCL communicator, which
 * creates streams of its own, was initialised before the model's first step).  Call it before kws_comm_init. */
int kws_model_bind_device(kws_model *m);
/* bytes of 256-byte aligned device scratch the calls below need for batch B */
int64_t kws_model_workspace_bytes(const kws_model *m, int B, int training);

/* model.predict (inference mode: BN moving statistics, no dropout).
 * feat (B, n_features, feature_size) -> probs (B, C) and/or argmax (B) (either may be NULL). */
int kws_model_forward(kws_model *m, const float *feat, int B, const float *params, const float *state, void *ws,
                      size_t ws_bytes, float *probs, int32_t *argmax, void *stream);

/* Inference with FIXED weights (serving, evaluation): kws_model_prepare_inference derives everything kws_model_forward computes
 * from the weights alone -- the folded BatchNormalization coefficients, the bf16 planes of the split-precision matrix operands,
 * the fp16 weight blob of simple_cnn_lite -- once, into `ws`; every later kws_model_forward with the SAME (B, params, state, ws)
 * and precisions then skips that work (17 of 300 us per call at B = 4096 for simple_cnn).  The caller promises not to change
 * `params` / `state` in between; kws_model_invalidate_prepared (or a train step, or a different forward in the same workspace)
 * drops the prepared state; it is keyed by the model's precisions, so a forward at another precision simply does not match it.  Recurrent models have nothing to prepare (returns KWS_OK). */
int kws_model_prepare_inference(kws_model *m, int B, const float *params, const float *state, void *ws, size_t ws_bytes, void *stream);
int kws_model_invalidate_prepared(kws_model *m);

/* One training forward + backward (what Keras does per batch inside model.fit, train.py:81):
 * batch-statistics BN (moving stats in `state` are updated), dropout from `dropout_seed` (0 = off),
 * loss = classifier/loss.py SparseCategoricalCrossEntropy (class_weights NULL) or
 * WeightedSparseCategoricalCrossEntropy (class_weights: C device floats), reduced by the batch mean.
 * grads <- grad_scale * d(mean loss)/d(params)  (data parallel: grad_scale = 1/world, then sum-all-reduce). */
struct kws_comm;
typedef struct kws_train_args {
    const float *feat;          /* (B, n_features, feature_size)                                   */
    const int32_t *labels;      /* (B) class indices                                               */
    const float *class_weights; /* NULL or (C)                                                     */
    int32_t B;
    int32_t ignore_index;       /* <= 0: none (the reference tests truthiness, loss.py:25,59); else samples
                                   with this label contribute zero loss and zero gradient            */
    const float *params;
    float *state;
    float *grads;
    void *ws;
    size_t ws_bytes;
    uint64_t dropout_seed;
    float grad_scale;
    float *probs;               /* NULL or (B, C)                                                   */
    float *stats;               /* NULL or 2 floats: {sum of per-sample losses, number of top-1 hits} */
    void *bucket_event;         /* NULL or a hipEvent_t recorded (possibly on a library-internal stream) as soon as the
                                   gradients of the LAST kws_model_grad_split() .. param_count floats are final (they are
                                   produced first by the backward pass): a stream that waits on it may all-reduce that
                                   bucket while the rest of the backward pass runs */
    void *forward_event;        /* NULL or a hipEvent_t recorded on `stream` once the forward pass and the loss are
                                   enqueued: work that should share the chip with the backward pass (the next batch's
                                   featurization) can be ordered after it                                       */
    void *overlap_event;        /* NULL or a hipEvent_t recorded on `stream` at the best point of the step to start independent
                                   vector-ALU / memory work on another stream (the next batch's featurization).  simple_cnn: behind
                                   conv3's forward kernel (round 3's sweep of eleven points at B = 4096: 0.548 ms per step there,
                                   0.579 at forward_event; kws_model.hip; kws_model_set_overlap_point moves it);
                                   simple_cnn_lite and the recurrent models record it together with forward_event.          */
    void (*overlap_callback)(void *user);   /* NULL or a host function the call invokes (same thread, once) right after it has
                                   enqueued the work overlap_event marks: enqueueing the next batch's kws_featurize from it
                                   puts that launch at the same place in HOST order, so the overlap does not depend on how far
                                   the host runs ahead of the device (under a tracing profiler it does not run ahead at all) */
    void *overlap_user;
    struct kws_comm *comm;      /* NULL, or a communicator (kws_comm_init): the step then EXCHANGES its gradients itself -- sum over the
                                   ranks, in place: the early bucket grads[kws_model_grad_split(m), P) (conv4 + BN4 + dense + head, 82 % of
                                   the bytes, final first) on the model's side stream right behind conv4's weight gradient, i.e. under the
                                   rest of the backward pass; the late bucket together with `state * comm_state_weight` (BatchNormalization
                                   moving statistics: weight = local clips / global clips gives their batch-weighted mean over the replicas)
                                   on `stream` behind the backward pass.  When the call returns, kws_adam_step can be enqueued on `stream`.
                                   Set grad_scale = local clips / global clips.  bucket_event is not needed (and still honoured).        */
    float comm_state_weight;
    const double *feat_moments; /* NULL or the KWS_FEATURE_MOMENTS doubles kws_feature_moments() wrote for `feat` (same B): simple_cnn
                                   derives the batch statistics of its first BatchNormalization and the closed forms of its first
                                   layer's gradients from them instead of computing them at the head of the step, so an input
                                   pipeline can prepare them on its own stream right behind the featurizer.  Ignored by the
                                   other model kinds and at geometries kws_feature_moments() does not cover.             */
} kws_train_args;
int kws_model_train_fwd_bwd(kws_model *m, const kws_train_args *a, void *stream);

/* Second moments of a feature batch as seen by a 3x3 'same' convolution with one input channel (the first layer of
 * classifier/models/cnn.py:27): Q[t][t'] = sum over clips and pixels of a_t a_t', a_t = the feature at tap t of the pixel's
 * 3x3 patch (zero outside the map) for t < 9 and a_9 = 1; row-major 10 x 10 doubles (Q[t][9] = tap sums, Q[9][9] = B*H*W).
 * They depend on the features only, so they can be computed where the features are produced (kws_train_args.feat_moments).
 * ws: kws_feature_moments_workspace_bytes() bytes of 256-byte aligned device scratch.  Deterministic (fixed summation order).
 * KWS_ERR_UNSUPPORTED outside the geometries of the wave-per-clip kernels (H, W even, (H+2)(W+2) <= 768, H*W/4 <= 160): pass
 * feat_moments = NULL there. */
#define KWS_FEATURE_MOMENTS 100
int64_t kws_feature_moments_workspace_bytes(int B);
int kws_feature_moments(const float *feat, int B, int n_features, int feature_size, double *moments, void *ws, size_t ws_bytes,
                        void *stream);
/* Arithmetic of the GEMM-shaped layers with 32 or more reduced channels (simple_cnn: conv3, conv4, dense).
 *   KWS_MATRIX_BF16X6 (default): every fp32 operand is carried as h + m + l in bf16 (24 bits) and a product is the six
 *                      leading partial products on the bf16 matrix cores with fp32 accumulation: fp32-level error at
 *                      ~2.7x the fp32 matrix rate
 *   KWS_MATRIX_FP32:   the fp32 MFMA everywhere (bit-identical to an fp32 fmaf chain)
 * kws_set_matrix_precision sets the library-wide DEFAULT, which a model follows until kws_model_set_precision gives it its own
 * value; read at the next forward / train call. */
enum { KWS_MATRIX_FP32 = 0, KWS_MATRIX_BF16X6 = 1 };
int kws_set_matrix_precision(int mode);
int kws_get_matrix_precision(void);
/* Precision of simple_cnn_lite INFERENCE (kws_model_forward; BASELINE configs[4] asks fp16).  Library-wide default, per model
 * through kws_model_set_precision.
 *   KWS_INFER_FP32 (default): fp32 activations and products.
 *   KWS_INFER_FP16: the activations between stages and every matrix operand are fp16, all accumulation (depthwise taps,
 *                   matrix products, bias / BatchNorm affine, softmax) fp32; the network behind the second pooling stage
 *                   runs as ONE kernel with its fp16 weights in LDS.  Needs the default geometry family (pooled maps up to
 *                   7 x 5 / 4 x 3) and at most 48 classes, otherwise kws_model_forward returns KWS_ERR_UNSUPPORTED.
 * Other model kinds ignore the switch (classifier/models/cnn.py:77-141 is the topology it applies to). */
enum { KWS_INFER_FP32 = 0, KWS_INFER_FP16 = 1 };
int kws_set_inference_precision(int mode);
int kws_get_inference_precision(void);
/* Per-model precision attributes: `matrix` in {KWS_MATRIX_FP32, KWS_MATRIX_BF16X6}, `infer` in {KWS_INFER_FP32, KWS_INFER_FP16},
 * or -1 = follow the library-wide default above.  Two models with different precisions may live in one process (one host
 * thread per model; a model's calls are not re-entrant).  kws_model_get_precision reports the EFFECTIVE values. */
int kws_model_set_precision(kws_model *m, int matrix, int infer);
int kws_model_get_precision(const kws_model *m, int *matrix, int *infer);
/* Test aid: with on != 0 every weight-gradient reduction of simple_cnn / simple_cnn_lite runs in a fixed order (one block
 * along the reduced axis, a batch-ordered head kernel) instead of per-block float atomics, so two runs of a step give
 * bit-identical gradients.  Much slower at large batches; recurrent models: KWS_ERR_UNSUPPORTED. */
int kws_model_set_deterministic(kws_model *m, int on);

/* Tuning aid: where in the simple_cnn train step kws_train_args.overlap_event is recorded / overlap_callback is called.  -1 (default):
 * the library's choice (10 = behind conv3's forward); 6 behind the last forward convolution's BatchNormalization statistics (its activation now rides in the Dense + head kernel), 0 behind the last forward convolution, 1 behind
 * the loss, 2 behind the head's backward kernel, 3 behind the dense data gradient, 4 behind BatchNorm-4's backward, 5 behind conv4's
 * data gradient, 7 behind the dense forward product, 8 behind layer 1's forward kernel, 9 behind conv2's forward, 10 behind conv3's.
 * Changes scheduling only, never results (tests/test_model_gpu.py). */
int kws_model_set_overlap_point(kws_model *m, int point);

/* offset (in floats) that splits `grads` into {late bucket [0, split), early bucket [split, param_count)} */
int64_t kws_model_grad_split(const kws_model *m);

/* ------------------------------------------------------------------------
 * Data-parallel exchange (RCCL over xGMI).  New: the reference trains in one
 * process (train.py:81-92, model.fit(..., workers=1) at :90-91); this is the
 * collective SURVEY.md section 5 / 8(e) specify around that loop.  One process
 * per GPU, one communicator per process, created on the CURRENT device.
 * RCCL is bound at run time (dlopen of librccl.so.1, sharing the instance the
 * process already holds if any), so single-GPU hosts never load it.
 *
 *   rank 0:  kws_comm_unique_id(id);  ship the 128 bytes to every rank (file, socket, MPI, torch.distributed ...)
 *   all:     kws_comm_init(rank, world, id, &comm)            -- collective
 *   step:    args.comm = comm; args.grad_scale = args.comm_state_weight = local clips / global clips;
 *            kws_model_train_fwd_bwd(m, &args, stream);       -- gradients arrive summed over the ranks
 *            kws_adam_step(...)
 * The communicator owns no stream: every collective is enqueued on a stream the step already uses (a collective on a stream of its
 * own stalled the device by ~1.1 ms per step on MI355X / ROCm 7.2: 0.65 -> 1.79 ms, tools/commbench.py).
 * ---------------------------------------------------------------------- */
typedef struct kws_comm kws_comm;
#define KWS_COMM_ID_BYTES 128
int kws_comm_unique_id(void *id /* KWS_COMM_ID_BYTES host bytes */);
int kws_comm_init(int rank, int world, const void *unique_id, kws_comm **out);
void kws_comm_destroy(kws_comm *c);
/* rccl_version: NCCL-style code of the bound library (e.g. 22707), 0 if unknown; any out pointer may be NULL */
int kws_comm_info(const kws_comm *c, int *rank, int *world, int *rccl_version);

/* The exchange as a call of its own, for steps that did not run kws_model_train_fwd_bwd with args.comm (a rank whose shard of a
 * partial last batch is empty: cleared gradients, weight 0; or a caller with its own backward pass): in-place sum over the ranks of
 * grads[split, n), then -- one RCCL group -- of grads[0, split) and of state[0, n_state) * state_weight, all on `stream`, i.e.
 * the same collectives in the same order as the train step issues, so ranks may mix the two forms.  split in {0, n}: one bucket. */
int kws_allreduce_grads(kws_comm *c, float *grads, int64_t n, int64_t split, float *state, int64_t n_state, float state_weight,
                        void *stream);

/* A plain in-place all-reduce on `stream` (loss / hit counters of a logging step, timing maxima). */
enum { KWS_DT_F32 = 0, KWS_DT_F64 = 1, KWS_DT_I32 = 2, KWS_DT_I64 = 3 };
enum { KWS_OP_SUM = 0, KWS_OP_MAX = 1, KWS_OP_AVG = 2 };
int kws_comm_allreduce(kws_comm *c, void *buf, int64_t n, int dtype, int op, void *stream);
/* In-place broadcast of nbytes device bytes from rank `root` on `stream`: what a data-parallel fit needs once per run (the initial
 * weights / moving statistics, where the reference has a single model object, classifier/model.py:14-46) and once per epoch (the
 * shuffle permutation of train.py:90, shuffle=True), so that no second communicator -- with streams of its own -- is ever built. */
int kws_comm_broadcast(kws_comm *c, void *buf, int64_t nbytes, int root, void *stream);

/* Opt-in timing of the two buckets of the most recent exchange (HIP events around each RCCL launch on the stream it was enqueued
 * on); kws_comm_last_us synchronises those events; -1 = that bucket was not issued / timing off. */
int kws_comm_timing(kws_comm *c, int on);
int kws_comm_last_us(kws_comm *c, float *early_us, float *late_us);

/* Standalone losses of classifier/loss.py: y_pred (B, C) probabilities (or logits when from_logits != 0),
 * labels (B) -> per-sample losses (B), exactly what the two classes' __call__ return. */
int kws_loss_forward(const float *y_pred, const int32_t *labels, const float *class_weights, int from_logits,
                     int ignore_index, int B, int C, float *losses, void *stream);

/* keras.optimizers.Adam update (common/model_utils.py:47) on flat buffers of n floats, step count t >= 1:
 *   lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m/(sqrt(v)+eps)
 * with g = grad_scale * grads. */
int kws_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t t, float grad_scale, void *stream);

/* Evaluation (eval.py:201-256): counts[label * C + pred] += 1 for every sample of the batch (int32, caller zeroes it). */
int kws_confusion_counts(const int32_t *labels, const int32_t *pred, int B, int C, int32_t *counts, void *stream);

/* keras SGD(momentum=0) and RMSprop(rho=0.9, momentum=0, epsilon=1e-7, centered=False), model_utils.py:48-51:
 *   sgd:     p -= lr * g
 *   rmsprop: a = rho a + (1-rho) g^2;  p -= lr * g / (sqrt(a) + eps)            with g = grad_scale * grads */
int kws_sgd_step(float *params, const float *grads, int64_t n, float lr, float grad_scale, void *stream);
int kws_rmsprop_step(float *params, const float *grads, float *accum, int64_t n, float lr, float rho, float eps,
                     float grad_scale, void *stream);

/* ------------------------------------------------------------------------
 * Streaming post-processing: replaces the per-chunk work of listen.py for S
 * concurrent audio streams that advance in lockstep (one chunk each per step):
 *   Listener.update_vectors   listen.py:96-114   sliding feature matrix
 *   ThresholdDecoder          listen.py:452-522  logit-normal cumulative table
 *   TriggerDetector.update    listen.py:525-559  activation counter
 *   the loop around them      listen.py:350-375  argmax / max / decode / update
 * New MFCC rows come from kws_featurize_raw() on the carried + new samples.
 * ------------------------------------------------------------------------ */
typedef struct kws_decoder kws_decoder;

/* ThresholdDecoder.__init__ (listen.py:467-472): mu_stds is n pairs (mu, std) on the host; the cumulative table
 * (resolution * out_range float64 entries) is built once in double precision and kept on the device. */
int kws_decoder_create(const double *mu_stds, int n, double center, int resolution, double min_z, double max_z,
                       kws_decoder **out);
void kws_decoder_destroy(kws_decoder *d);
/* min_out, out_range (= max_out - min_out) and len(cd) */
int kws_decoder_info(const kws_decoder *d, int32_t *min_out, int32_t *out_range, int64_t *table_len);
/* host copy of the table `cd` (count must equal table_len) */
int kws_decoder_table(const kws_decoder *d, double *host_cd, size_t count);
/* ThresholdDecoder.decode (listen.py:496-508) on n device values.  raw_dtype KWS_RAW_F64: Python-float semantics;
 * KWS_RAW_F32: the live loop's semantics, where the float32 network output makes numpy evaluate 1/x - 1 in float32. */
enum { KWS_RAW_F64 = 0, KWS_RAW_F32 = 1 };
int kws_decoder_decode(const kws_decoder *d, const void *raw, int raw_dtype, double *decoded, int64_t n, void *stream);
/* ThresholdDecoder.encode (listen.py:510-517), a host-side scalar helper (searchsorted on the host copy of the table) */
int kws_decoder_encode(const kws_decoder *d, double threshold, double *raw_out);

/* mfccs = concatenate(mfccs[n:], new[-n:]) with n = min(n_rows, F) for every stream (listen.py:107-109):
 * feat (S, F, D) updated in place from rows (S, n_rows, D). */
int kws_stream_push_rows(float *feat, const float *rows, int S, int F, int D, int n_rows, void *stream);

/* TriggerDetector.update (listen.py:538-559) for S streams: state (S, 2) int32 = {activation, record_index}, start it
 * at {0, -1}.  fired[s] = 1 when the prediction activates the stream. */
int kws_trigger_update(const int32_t *index, const double *score, int S, int background_index, double sensitivity,
                       int trigger_level, int chunk_size, int32_t *state, int32_t *fired, void *stream);

/* One step of the loop listen.py:361-375 for S streams, fused: probs (S, C) float32 -> index = argmax, score = max,
 * decoded through `dec` unless the class is background (dec may be NULL: no decoding), then the trigger update above. */
int kws_stream_postprocess(const kws_decoder *dec, const float *probs, int S, int C, int background_index,
                           double sensitivity, int trigger_level, int chunk_size, int32_t *state, int32_t *index,
                           double *score, int32_t *fired, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KWS_H */
