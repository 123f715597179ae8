// csrc/kws_model_types.h -- model descriptor shared by the CNN and RNN translation units.
#pragma once
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kws_common.h"
#include "kws_device.h"

namespace kws {

struct Tensor {
    std::string name;
    std::vector<int> shape;
    bool trainable;
    int64_t offset, size;
};

// Streams and events a model's train step forks onto (weight gradients beside the data-gradient chain).  Owned by the
// model -- two models, or one model per host thread and device, never share an event that one of them re-records.
struct ModelRes {
    hipStream_t side = nullptr;
    hipEvent_t ev[16] = {nullptr};
    // BatchNorm sum accumulators of the finalize-free train step (kws_device.h: acc_add): [pass 0 = forward, 1 = backward][layer 0..3]
    // [parity][kAccDoubles] doubles, zero at creation; a pass uses the sets of its parity and its consumers clear the other one
    double *acc = nullptr;
    unsigned acc_uses[2][4] = {};       // passes that USED the sets of (pass, layer): the parity of the next one (a pass that keeps the
                                        // partial-sum form -- deterministic mode, another geometry -- must not flip it: it clears nothing)
    bool acc_dirty = false;             // an enqueue failed half-way: clear everything before the next pass
    bool pool4_pending = false;         // the forward pass left layer 4's activation to the fused Dense + head kernel of the backward pass
    float *pool4_mm = nullptr, *pool4_mv = nullptr;     // ... and these are BatchNorm-4's moving statistics it will update
};

struct CnnDims { int H0, W0, H1, W1, H2, W2, H3, W3, H4, W4, flat; };

inline int64_t al4(int64_t x) { return (x + 3) & ~(int64_t)3; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline unsigned blocks_for(long n, int per) { return (unsigned)((n + per - 1) / per); }

// bump allocator over the caller's workspace (base may be NULL to only measure)
struct WsCarver {
    unsigned char *base;
    size_t off = 0;
    explicit WsCarver(void *b) : base(static_cast<unsigned char *>(b)) {}
    float *take(size_t nfloats)
    {
        float *p = reinterpret_cast<float *>(base + off);
        off = al256(off + nfloats * sizeof(float));
        return p;
    }
};

}  // namespace kws

struct kws_model {
    int kind, C, n_features, feature_size;
    // -1: follow the library-wide default (kws_set_matrix_precision / kws_set_inference_precision); else the model's own
    int matrix_precision = -1, infer_precision = -1;
    int deterministic = 0;       // 1: weight gradients reduced in a fixed order (kws_model_set_deterministic)
    // kws_model_prepare_inference: the weight-derived tables of an inference forward (bf16 weight planes, folded BatchNorm
    // coefficients, fp16 weight blob) already sit in THIS workspace for THESE buffers, batch and precisions
    int overlap_point = -1;             // kws_model_set_overlap_point: -1 = the library's choice
    struct Prepared { const float *params = nullptr, *state = nullptr; void *ws = nullptr; int B = 0, matrix = -2, infer = -2; } prep;
    bool prepared_for(const float *p, const float *st, void *w, int B, int matrix, int infer) const
    {
        return prep.params == p && prep.state == st && prep.ws == w && prep.B == B && prep.matrix == matrix && prep.infer == infer && p != nullptr;
    }
    std::mutex res_mu;
    std::map<int, kws::ModelRes> res;   // per device, created on first use
    kws::ModelRes *dev_res();           // resources for the CURRENT device (nullptr on failure)
    ~kws_model();
    std::vector<kws::Tensor> tensors;
    int64_t P = 0, S = 0;
    kws::CnnDims d{};
    // simple_cnn offsets into params / state
    int64_t o_k[4], o_g[4], o_b[4], o_dk, o_db, o_mm[4], o_mv[4];
    // simple_cnn_lite: depthwise / pointwise kernels and pointwise bias of the four SeparableConv2D stages
    int64_t o_dwk[4], o_pwk[4], o_pwb[4];
    // simple_gru offsets
    int64_t o_rk, o_ru, o_rb;
    // head (all models): Dense(C) on a K-wide feature vector
    int64_t o_hk, o_hb;
    int head_K = 128;

    int64_t add(const std::string &name, std::vector<int> shape, bool trainable)
    {
        int64_t n = 1;
        for (int s : shape) n *= s;
        int64_t &cur = trainable ? P : S;
        const int64_t off = cur;
        tensors.push_back({name, shape, trainable, off, n});
        cur = kws::al4(cur + n);
        return off;
    }
};

namespace kws {

// head shared by every model (kws_model.hip)
int run_head(const kws_model *m, int B, const float *params, const float *x, float *loss_i, float *correct_i,
             const int32_t *labels, const float *class_w, float *probs, int32_t *argmax, float *dlogits, float grad_scale,
             float *stats, int ignore_index, hipStream_t s);
// dx_colsum / stats: optional extras the MFMA form of the kernel produces on the way (bias gradient of the layer in front of
// the head; {sum loss, sum correct} from loss_i / correct_i).  head_bwd_fuses(m) tells whether that form applies.
bool head_bwd_fuses(const kws_model *m);
int run_head_bwd(const kws_model *m, int B, const float *params, const float *x, const float *dlogits, float *dx,
                 float *grads, bool relu6_gate, hipStream_t s, float *dx_colsum = nullptr, const float *loss_i = nullptr,
                 const float *correct_i = nullptr, float *stats = nullptr, bool deterministic = false, const HeadFwdArgs *fwd = nullptr);

// stats[0 .. 1] = the sums of the per-sample losses / correct flags, in a fixed order (loss_reduce_kernel)
int run_loss_reduce(const float *loss_i, const float *correct_i, int B, float *stats, hipStream_t s);

// simple_gru (kws_rnn.hip)
size_t gru_workspace_bytes(const kws_model *m, int B, bool training);
int gru_forward(kws_model *m, const float *feat, int B, const float *params, void *ws, size_t ws_bytes, float *probs,
                int32_t *argmax, hipStream_t s);
int gru_train_fwd_bwd(kws_model *m, const kws_train_args *a, hipStream_t s);

}  // namespace kws
