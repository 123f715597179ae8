// csrc/kws_rnn.hip -- the recurrent models behind the C ABI: GRU(48, linear) or LSTM(48, tanh) -> Dense(C, softmax)
// (classifier/models/rnn.py:10-43 and 46-79, classifier/model.py:20-29,37).
#include "kws_lstm.h"
#include "kws_model_types.h"

namespace kws {

namespace {

struct GruWs {
    float *h_last, *loss_i, *correct_i, *saved, *dlogits, *dh_last;
    size_t bytes;
};

GruWs carve_gru(const kws_model *m, int B, bool training, void *base)
{
    WsCarver c(base);
    GruWs w{};
    w.h_last = c.take((size_t)B * kGruU);
    w.loss_i = c.take(B);
    w.correct_i = c.take(B);
    if (training) {
        w.saved = c.take((size_t)B * m->n_features * (m->kind == KWS_SIMPLE_LSTM ? kLstmSave : kGruSave) * kGruU);
        w.dlogits = c.take((size_t)B * m->C);
        w.dh_last = c.take((size_t)B * kGruU);
    }
    w.bytes = c.off;
    return w;
}

int check(const kws_model *m, int B, bool training, void *ws, size_t ws_bytes, GruWs &w)
{
    if (!ws) return fail(KWS_ERR_INVALID, "null workspace");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(KWS_ERR_INVALID, "workspace must be 256-byte aligned");
    w = carve_gru(m, B, training, ws);
    if (w.bytes > ws_bytes) return fail(KWS_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return KWS_OK;
}

template <int KX>
int launch_fwd(const kws_model *m, const float *feat, int B, const float *params, GruWs &w, bool save, float rate, uint64_t seed,
               hipStream_t s, float *zero_buf, long zero_n)
{
    const int T = m->n_features, F = m->feature_size;
    const size_t smem = gru_fwd_smem(T, F);
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);
    if (smem > 160 * 1024) return fail(KWS_ERR_UNSUPPORTED, "GRU tile needs %zu B of LDS", smem);
    if (m->kind == KWS_SIMPLE_LSTM) {
        if (save) {
            if (smem > 64 * 1024)
                KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_fwd_kernel<KX, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            KWS_LAUNCH("lstm_fwd_kernel", (lstm_fwd_kernel<KX, true>), dim3(blocks_for(B, 16)), dim3(kLstmFwdThreads), smem, s, feat, params + m->o_rk,
                       params + m->o_ru, params + m->o_rb, w.h_last, w.saved, B, T, F, rate, slo, shi, zero_buf, zero_n);
        } else {
            if (smem > 64 * 1024)
                KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_fwd_kernel<KX, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            KWS_LAUNCH("lstm_fwd_kernel", (lstm_fwd_kernel<KX, false>), dim3(blocks_for(B, 16)), dim3(kLstmFwdThreads), smem, s, feat, params + m->o_rk,
                       params + m->o_ru, params + m->o_rb, w.h_last, w.saved, B, T, F, rate, slo, shi);
        }
        KWS_LAUNCH_CHECK("lstm_fwd_kernel");
        return KWS_OK;
    }
    if (save) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gru_fwd_kernel<KX, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("gru_fwd_kernel", (gru_fwd_kernel<KX, true>), dim3(blocks_for(B, 16)), dim3(kGruFwdThreads), smem, s, feat, params + m->o_rk,
                   params + m->o_ru, params + m->o_rb, w.h_last, w.saved, B, T, F, rate, slo, shi, zero_buf, zero_n);
    } else {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gru_fwd_kernel<KX, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("gru_fwd_kernel", (gru_fwd_kernel<KX, false>), dim3(blocks_for(B, 16)), dim3(kGruFwdThreads), smem, s, feat, params + m->o_rk,
                   params + m->o_ru, params + m->o_rb, w.h_last, w.saved, B, T, F, rate, slo, shi);
    }
    KWS_LAUNCH_CHECK("gru_fwd_kernel");
    return KWS_OK;
}

template <int KX>
int launch_bwd(const kws_model *m, const float *feat, int B, const float *params, float *grads, GruWs &w, float rate, uint64_t seed,
               hipStream_t s)
{
    const int T = m->n_features, F = m->feature_size;
    const size_t smem = gru_bwd_smem(T, F);
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);
    if (smem > 160 * 1024) return fail(KWS_ERR_UNSUPPORTED, "GRU tile needs %zu B of LDS", smem);
    if (m->kind == KWS_SIMPLE_LSTM) {
        if (smem > 64 * 1024)
            KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_kernel<KX>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        KWS_LAUNCH("lstm_bwd_kernel", lstm_bwd_kernel<KX>, dim3(blocks_for(B, 16)), dim3(kLstmBwdThreads), smem, s, feat, params + m->o_ru, w.saved,
                   w.dh_last, grads + m->o_rk, grads + m->o_ru, grads + m->o_rb, B, T, F, rate, slo, shi);
        KWS_LAUNCH_CHECK("lstm_bwd_kernel");
        return KWS_OK;
    }
    if (smem > 64 * 1024)
        KWS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gru_bwd_kernel<KX>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    KWS_LAUNCH("gru_bwd_kernel", gru_bwd_kernel<KX>, dim3(blocks_for(B, 16)), dim3(kGruBwdThreads), smem, s, feat, params + m->o_ru, w.saved,
               w.dh_last, grads + m->o_rk, grads + m->o_ru, grads + m->o_rb, B, T, F, rate, slo, shi);
    KWS_LAUNCH_CHECK("gru_bwd_kernel");
    return KWS_OK;
}

int dispatch_fwd(const kws_model *m, const float *feat, int B, const float *params, GruWs &w, bool save, float rate, uint64_t seed,
                 hipStream_t s, float *zero_buf = nullptr, long zero_n = 0)
{
    const int kx = (m->feature_size + 3) / 4;
    if (kx <= 5) return launch_fwd<5>(m, feat, B, params, w, save, rate, seed, s, zero_buf, zero_n);
    if (kx <= 10) return launch_fwd<10>(m, feat, B, params, w, save, rate, seed, s, zero_buf, zero_n);
    return launch_fwd<16>(m, feat, B, params, w, save, rate, seed, s, zero_buf, zero_n);
}

int dispatch_bwd(const kws_model *m, const float *feat, int B, const float *params, float *grads, GruWs &w, float rate, uint64_t seed,
                 hipStream_t s)
{
    const int kx = (m->feature_size + 3) / 4;
    if (kx <= 5) return launch_bwd<5>(m, feat, B, params, grads, w, rate, seed, s);
    if (kx <= 10) return launch_bwd<10>(m, feat, B, params, grads, w, rate, seed, s);
    return launch_bwd<16>(m, feat, B, params, grads, w, rate, seed, s);
}

}  // namespace

size_t gru_workspace_bytes(const kws_model *m, int B, bool training) { return carve_gru(m, B, training, nullptr).bytes; }

int gru_forward(kws_model *m, const float *feat, int B, const float *params, void *ws, size_t ws_bytes, float *probs,
                int32_t *argmax, hipStream_t s)
{
    GruWs w;
    int rc = check(m, B, false, ws, ws_bytes, w);
    if (rc) return rc;
    rc = dispatch_fwd(m, feat, B, params, w, false, 0.f, 0, s);
    if (rc) return rc;
    return run_head(m, B, params, w.h_last, w.loss_i, w.correct_i, nullptr, nullptr, probs, argmax, nullptr, 0.f, nullptr, 0, s);
}

int gru_train_fwd_bwd(kws_model *m, const kws_train_args *a, hipStream_t s)
{
    GruWs w;
    int rc = check(m, a->B, true, a->ws, a->ws_bytes, w);
    if (rc) return rc;
    const float rate = a->dropout_seed != 0 ? 0.2f : 0.f;          // GRU / LSTM(dropout=0.2): input dropout, rnn.py:34-35,70-71
    if (head_bwd_fuses(m)) {
        // The recurrent kernel clears the gradient buffer in its own grid, and the head's forward pass (logits, softmax, loss, dlogits)
        // runs inside its backward kernel (kws_layers.h: head_bwd_mfma_kernel<.., FWD>), as in the simple_cnn step: between the
        // recurrent forward and backward kernels the chain is ONE launch instead of four (head forward, loss sums, memset, head backward)
        rc = dispatch_fwd(m, a->feat, a->B, a->params, w, true, rate, a->dropout_seed, s, a->grads, (long)m->P);
        if (rc) return rc;
        const HeadFwdArgs hf{a->params + m->o_hb, a->labels, a->class_weights, a->probs, w.loss_i, w.correct_i, a->grad_scale / (float)a->B, a->ignore_index};
        rc = run_head_bwd(m, a->B, a->params, w.h_last, nullptr, w.dh_last, a->grads, false, s, nullptr, nullptr, nullptr, nullptr, false, &hf);
        if (rc) return rc;
        if (a->overlap_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->overlap_event), s));
        if (a->overlap_callback) a->overlap_callback(a->overlap_user);
        if (a->forward_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->forward_event), s));
        rc = dispatch_bwd(m, a->feat, a->B, a->params, a->grads, w, rate, a->dropout_seed, s);
        if (rc) return rc;
        if (a->stats) { rc = run_loss_reduce(w.loss_i, w.correct_i, a->B, a->stats, s); if (rc) return rc; }   // fixed-order sums of the per-sample values
        if (a->bucket_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->bucket_event), s));
        return KWS_OK;
    }
    rc = dispatch_fwd(m, a->feat, a->B, a->params, w, true, rate, a->dropout_seed, s);
    if (rc) return rc;
    rc = run_head(m, a->B, a->params, w.h_last, w.loss_i, w.correct_i, a->labels, a->class_weights, a->probs, nullptr, w.dlogits,
                  a->grad_scale / (float)a->B, a->stats, a->ignore_index, s);
    if (rc) return rc;
    if (a->overlap_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->overlap_event), s));
    if (a->overlap_callback) a->overlap_callback(a->overlap_user);
    if (a->forward_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->forward_event), s));
    KWS_HIP_CHECK(hipMemsetAsync(a->grads, 0, sizeof(float) * (size_t)m->P, s));
    rc = run_head_bwd(m, a->B, a->params, w.h_last, w.dlogits, w.dh_last, a->grads, false, s);
    if (rc) return rc;
    rc = dispatch_bwd(m, a->feat, a->B, a->params, a->grads, w, rate, a->dropout_seed, s);
    if (rc) return rc;
    if (a->bucket_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->bucket_event), s));
    return KWS_OK;
}

}  // namespace kws
