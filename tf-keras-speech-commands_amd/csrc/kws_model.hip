// csrc/kws_model.hip -- model descriptors and the simple_cnn forward / backward behind the C ABI.
//
// Topology follows classifier/models/cnn.py:27-66 (SimpleCNN) and the softmax head of classifier/model.py:37.
// Flat buffers: `params` holds the trainable tensors and `state` the BatchNormalization moving statistics, both
// in Keras get_weights() order (kws_model_tensor_info lists name / shape / offset); `grads` mirrors `params`.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kws_common.h"
#include "kws_model_types.h"
#include "kws_conv.h"
#include "kws_layers.h"
#include "kws_layer1.h"
#include "kws_layer1_moments.h"
#include "kws_layer1_fast.h"
#include "kws_lite.h"
#include "kws_lite_f16.h"
#include "kws_infer_fused.h"
#include "kws_conv_group.h"
#include "kws_dense_head.h"

using namespace kws;

namespace {

inline int same_out(int n, int s) { return (n + s - 1) / s; }
inline int same_pad_before(int n, int k, int s)
{
    const int out = same_out(n, s), total = std::max((out - 1) * s + k - n, 0);
    return total / 2;   // TF 'SAME': the extra element goes to the end
}

// where in the step the caller's overlap event is recorded (and its callback runs): see kws_train_args.overlap_event
struct OverlapHook {
    hipEvent_t ev = nullptr;
    void (*cb)(void *) = nullptr;
    void *user = nullptr;
    int at = 0;
    bool fired = false;
    bool wants(int point) const { return !fired && point == at; }
    // already_bound: the event rides on the completion signal of the last kernel launched on s (kws_common.h: ArmedEvent)
    int fire(int point, hipStream_t s, bool already_bound = false)
    {
        if (fired || point != at) return KWS_OK;
        fired = true;
        if (ev && !already_bound) KWS_HIP_CHECK(hipEventRecord(ev, s));
        if (cb) cb(user);
        return KWS_OK;
    }
};

#define KWS_TRY_NB(...)               \
    do {                              \
        const int rc__ = (__VA_ARGS__); \
        if (rc__ < 0) return rc__;    \
    } while (0)
#define KWS_TRY(...)                  \
    do {                              \
        if (int rc__ = (__VA_ARGS__)) return rc__; \
    } while (0)

constexpr int kCh[5] = {1, 16, 32, 64, 128};
constexpr int kMaxStatBlocks = kStatStride;

// ---- workspace layout (simple_cnn) ---------------------------------------------------------------------------
struct CnnWs {
    float *z[4], *a[4], *d1, *logits_unused, *loss_i, *correct_i, *dlogits, *dd1, *da4, *gz[4], *da[3];
    float *dwo[4], *ddw[4];   // simple_cnn_lite: depthwise outputs and their gradients
    float *coef[4];     // 6*C floats per BN layer
    __bf16 *wsp[3][6];  // simple_cnn: bf16 h/m/l planes of conv3, conv4 and dense weights (original order x3, transposed x3; kws_conv.h)
    __bf16 *dzp[3];     // simple_cnn training: h/m/l planes of dz4 (written by BN4's backward, read by conv4's data / weight gradients)
    double *partial;    // [kMaxStatBlocks][2][256]
    double *moments;    // Q[10][10] of the feature map (kws_layer1_moments.h) when the caller did not supply it
    float *zmax2;       // training: z2 at the routed element of every pool window of layer 2, and
    unsigned char *arg2;   // that element's index (written by the forward activation kernel, kws_layers.h: bn_bwd_reduce_routed_kernel)
    float *zmax4;       // the same for layer 4 (full-size g: bn_bwd_reduce_routed_full_kernel)
    unsigned char *arg4;
    unsigned char *base;
    size_t bytes;
};

CnnWs carve_cnn(const kws_model *m, int B, bool training, unsigned char *base)
{
    CnnWs w{};
    w.base = base;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float *p = reinterpret_cast<float *>(base + off);
        off = al256(off + nfloats * sizeof(float));
        return p;
    };
    const CnnDims &d = m->d;
    const size_t zs[4] = {(size_t)d.H0 * d.W0 * 16, (size_t)d.H1 * d.W1 * 32, (size_t)d.H3 * d.W3 * 64, (size_t)d.H3 * d.W3 * 128};
    const size_t as[4] = {(size_t)d.H1 * d.W1 * 16, (size_t)d.H2 * d.W2 * 32, (size_t)d.H3 * d.W3 * 64, (size_t)d.flat};
    const bool lite = m->kind == KWS_SIMPLE_CNN_LITE;
    const size_t dws[4] = {(size_t)d.H0 * d.W0 * 1, (size_t)d.H1 * d.W1 * 16, (size_t)d.H3 * d.W3 * 32, (size_t)d.H3 * d.W3 * 64};
    for (int i = 0; i < 4; ++i) { w.z[i] = take((i == 0 && !lite) ? 0 : zs[i] * B); w.a[i] = take(as[i] * B); }   // simple_cnn never materialises z1 (kws_layer1.h)
    for (int i = 0; i < 4; ++i) w.dwo[i] = take(lite ? dws[i] * B : 0);
    w.d1 = take((size_t)B * 128);
    w.loss_i = take(B);
    w.correct_i = take(B);
    for (int i = 0; i < 4; ++i) w.coef[i] = take(6 * 128);
    {
        const size_t wn[3] = {(size_t)9 * 32 * 64, (size_t)9 * 64 * 128, (size_t)d.flat * 128};
        for (int t = 0; t < 3; ++t)
            for (int q = 0; q < 6; ++q) w.wsp[t][q] = reinterpret_cast<__bf16 *>(take(lite ? 0 : (wn[t] + 1) / 2));
    }
    w.partial = reinterpret_cast<double *>(take((size_t)kMaxStatBlocks * 9 * 64 * 2));   // [9*64 or 2*C rows][kStatStride] doubles
    w.moments = reinterpret_cast<double *>(take(2 * kMomCount));
    if (training) {
        w.dlogits = take((size_t)B * m->C);
        w.dd1 = take((size_t)B * 128);
        w.da4 = take((size_t)B * d.flat);
        for (int i = 0; i < 4; ++i) w.gz[i] = take((i == 0 && !lite) ? 0 : zs[i] * B);
        for (int p = 0; p < 3; ++p) w.dzp[p] = reinterpret_cast<__bf16 *>(take(lite ? 0 : (zs[3] * B + 1) / 2));
        for (int i = 0; i < 4; ++i) w.ddw[i] = take(lite ? dws[i] * B : 0);
        for (int i = 0; i < 3; ++i) w.da[i] = take(as[i] * B);
        w.zmax2 = take(lite ? 0 : as[1] * B);
        w.arg2 = reinterpret_cast<unsigned char *>(take(lite ? 0 : (as[1] * B + 3) / 4));
        w.zmax4 = take(lite ? 0 : as[3] * B);
        w.arg4 = reinterpret_cast<unsigned char *>(take(lite ? 0 : (as[3] * B + 3) / 4));
    }
    w.bytes = off;
    return w;
}

BnCoef coef_of(float *base, int C) { return BnCoef{base, base + C, base + 2 * C, base + 3 * C, base + 4 * C, base + 5 * C}; }
// A train step that is being CAPTURED into a hipGraph keeps the partial-sum forms: the accumulator sets' parity and the ticket counter are
// host-side / device-side state that a replay would not advance (the second replay would add to sums nobody cleared)
static bool stream_capturing(hipStream_t s) { return stream_is_capturing(s); }
// accumulator set of (pass, layer, parity) -- kws_model_types.h: ModelRes::acc
static double *acc_set(ModelRes *R, int pass, int layer, unsigned parity) { return R->acc + ((size_t)(pass * 4 + layer) * 2 + (parity & 1u)) * kAccDoubles; }
static int acc_make_clean(ModelRes *R, hipStream_t s)
{
    if (R->acc_dirty) {
        KWS_HIP_CHECK(hipMemsetAsync(R->acc, 0, sizeof(double) * 2 * 4 * 2 * kAccDoubles, s));
        R->acc_dirty = false;
    }
    return KWS_OK;
}

// inference: scale / shift of all four BatchNorm layers in one launch
static int infer_coefs(const kws_model *m, const float *params, const float *state, CnnWs &w, hipStream_t s)
{
    BnInferAll a;
    for (int l = 0; l < 4; ++l) {
        a.C[l] = kCh[l + 1];
        a.gamma[l] = params + m->o_g[l]; a.beta[l] = params + m->o_b[l];
        a.mm[l] = state + m->o_mm[l]; a.mv[l] = state + m->o_mv[l];
        a.k[l] = coef_of(w.coef[l], kCh[l + 1]);
    }
    KWS_LAUNCH("bn_infer_coef_all_kernel", bn_infer_coef_all_kernel, dim3(4), dim3(128), 0, s, a);
    return KWS_OK;
}

// rows-per-block and grid for the (M x C) channel reductions
inline void stat_grid(long M, int C, int &nblk, int &rows)
{
    const int R = 256 / C;
    long want = (M + (long)R * 8 - 1) / ((long)R * 8);
    nblk = (int)std::min<long>(kMaxStatBlocks, std::max<long>(1, want));
    rows = (int)((M + nblk - 1) / nblk);
    nblk = (int)((M + rows - 1) / rows);
}

template <int CR, int CO, int MODE, int EPI>
int launch_gemm(const float *src, const float *w, const float *bias, float *dst, const ConvGeom &g, hipStream_t s)
{
    const long M = (long)g.B * g.Ho * g.Wo;
    static const std::string name = std::string("conv_gemm_fwd<") + std::to_string(CR) + "," +
                                    std::to_string(CO) + ">";
    KWS_LAUNCH(name.c_str(), (conv_gemm_kernel<CR, CO, MODE, EPI>), dim3(blocks_for(M, 64)), dim3(256), 0, s, src, w, bias, dst, g);
    return KWS_OK;
}

// CUs of the current device (cached) and resident blocks per CU of a kernel at a given dynamic LDS size
static int cu_count() { return device_cus(); }
template <typename K>
static int resident_blocks(K kernel, int threads, size_t smem)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, smem) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
    return nb;
}

// dW += wgrad(x, dz).  The grid is sized so that all blocks are resident at once (ONE round): with more blocks than slots the
// last round runs on a fraction of the chip (measured on conv4: 990 blocks on 768 slots left the CUs idle 43 % of the time).
template <int CIN, int COUT, int GPB>
int launch_wgrad(const float *x, const float *dz, float *dw, const ConvGeom &g, hipStream_t s, bool deterministic = false)
{
    const float *zp = zero_page();
    if (!zp) return fail(KWS_ERR_NOMEM, "cannot allocate the zero page on this device");
    constexpr int CB = CIN >= 64 ? 64 : CIN;
    constexpr size_t smem = (4 * 64 * 4 + 16 * COUT) * sizeof(float);
    const long M = (long)g.B * g.Ho * g.Wo;
    const int ngroups = g.KH * g.KW * (CIN / CB), gy = (ngroups + GPB - 1) / GPB;
    const long steps = (M + 3) / 4;                                   // 4-pixel MFMA k-steps
    static const int occ = resident_blocks(conv_wgrad_direct_kernel<CIN, COUT, GPB>, 256, smem);
    // deterministic: ONE block along the pixel axis, so every output element receives a single add onto the cleared buffer
    const long slots = deterministic ? 1 : std::max<long>(1, (long)cu_count() * occ / gy); // blocks along x that fit at once
    // >= 8 steps per wave so the end-of-block tile reduction amortises
    const long gx_want = std::max<long>(1, std::min<long>(slots, (steps + 4 * 8 - 1) / (4 * 8)));
    const int spw = (int)((steps + 4 * gx_want - 1) / (4 * gx_want));
    const long gx = (steps + 4L * spw - 1) / (4L * spw);
    static const std::string name = "conv_wgrad<" + std::to_string(CIN) + "," + std::to_string(COUT) + ">";
    KWS_LAUNCH(name.c_str(), (conv_wgrad_direct_kernel<CIN, COUT, GPB>), dim3((unsigned)gx, (unsigned)gy), dim3(256), smem,
               s, x, dz, dw, zp, g, spw);
    return KWS_OK;
}

// dW += wgrad(x, dz) in the split-precision form (conv_wgrad_bf16_kernel): grid = (tap groups) * (pixel ranges), one resident
// round, the taps of a range on one XCD.  TPB = taps per block: 3 (one kernel row, dz split once for three products) pays
// for conv3 (0.047 -> 0.042 ms); for conv4 its 238 registers and 74 KB of LDS cost more than they save (0.111 -> 0.122 ms).
template <int CIN, int COUT, int TPB, bool DPRE = false, bool XBN = false>
int launch_wgrad_bf16(const float *x, const float *dz, float *dw, const ConvGeom &g, hipStream_t s, __bf16 *const *dz_planes = nullptr,
                      bool deterministic = false, const float *xbn = nullptr)
{
    if (TPB != 1 && g.KW != TPB) return fail(KWS_ERR_UNSUPPORTED, "split-precision weight gradient expects a kernel %d taps wide", TPB);
    const float *zp = zero_page();
    if (!zp) return fail(KWS_ERR_NOMEM, "cannot allocate the zero page on this device");
    constexpr size_t stage = (size_t)3 * 32 * 4 * (TPB * tr_row_words(CIN) + tr_row_words(COUT)), tile = (size_t)CIN * COUT * sizeof(float);
    constexpr size_t smem = stage > tile ? stage : tile;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(conv_wgrad_bf16_kernel<CIN, COUT, TPB, DPRE, XBN>), (int)smem)) return rc;
    static const int occ = resident_blocks(conv_wgrad_bf16_kernel<CIN, COUT, TPB, DPRE, XBN>, 256, smem);
    const long M = (long)g.B * g.Ho * g.Wo, nchunk = (M + 31) / 32;
    const int ngroups = g.KH * g.KW / TPB;
    // ranges: a multiple of 8 (one per XCD), at most one resident round, at least 4 chunks per block
    long nranges = std::max<long>(8, ((long)cu_count() * occ / ngroups) / 8 * 8);
    nranges = std::min<long>(nranges, std::max<long>(8, (nchunk / 4 + 7) / 8 * 8));
    int cpb = (int)((nchunk + nranges - 1) / nranges);
    if (deterministic) { nranges = 8; cpb = (int)nchunk; }   // range 0 takes every chunk: a single add per output element
    static const std::string name = "conv_wgrad_bf16<" + std::to_string(CIN) + "," + std::to_string(COUT) + ">";
    const Bf16Planes dpl{{DPRE ? dz_planes[0] : nullptr, DPRE ? dz_planes[1] : nullptr, DPRE ? dz_planes[2] : nullptr}};
    KWS_LAUNCH(name.c_str(), (conv_wgrad_bf16_kernel<CIN, COUT, TPB, DPRE, XBN>), dim3((unsigned)(nranges * ngroups)), dim3(256), smem, s, x, dz, dw,
               zp, g, cpb, (int)nranges, dpl, xbn);
    return KWS_OK;
}

// dx <- dgrad(dz): ONE launch over every stride-parity class of the input pixels (blockIdx.y = class).  MW (16-row tiles
// per wave) is picked so that the waves divide evenly over the SIMDs: every SIMD's matrix pipe then runs
// ceil(waves / SIMDs) * MW tile-times, and the smallest such product wins (ties: the larger MW reuses weights more).
template <int CR, int CO, int MW, int STRIDE>
int launch_dgrad_mw(const float *dz, const float *w, float *dx, const ConvGeom &g, const DgradClasses &cls, int ncls, long max_rows,
                     hipStream_t s)
{
    static const std::string name = "conv_dgrad<" + std::to_string(CR) + "," + std::to_string(CO) + ">";
    const float *zp = zero_page();
    if (!zp) return fail(KWS_ERR_NOMEM, "cannot allocate the zero page on this device");
    KWS_LAUNCH(name.c_str(), (conv_dgrad_direct_kernel<CR, CO, MW, STRIDE>), dim3(blocks_for(max_rows, 64 * MW), (unsigned)ncls), dim3(256), 0, s,
               dz, w, dx, zp, g, cls);
    return KWS_OK;
}

template <int CR, int CO, int STRIDE>
int launch_dgrad(const float *dz, const float *w, float *dx, const ConvGeom &g, hipStream_t s)
{
    if (g.KH > 8 * g.stride || g.KW > 8 * g.stride)
        return fail(KWS_ERR_UNSUPPORTED, "kernel %dx%d too large for the dgrad tap masks (at most %d taps per axis)", g.KH, g.KW, 8 * g.stride);
    DgradClasses cls;
    int ncls = 0;
    long rows[4] = {0, 0, 0, 0}, max_rows = 0;
    for (int cy = 0; cy < g.stride && cy < 2; ++cy)
        for (int cx = 0; cx < g.stride && cx < 2; ++cx) {
            DgradClass c{cy, cx, (g.H - cy + g.stride - 1) / g.stride, (g.W - cx + g.stride - 1) / g.stride};
            if (c.ny <= 0 || c.nx <= 0) continue;
            rows[ncls] = (long)g.B * c.ny * c.nx;
            max_rows = std::max(max_rows, rows[ncls]);
            cls.c[ncls++] = c;
        }
    for (int i = ncls; i < 4; ++i) cls.c[i] = DgradClass{0, 0, 0, 0};
    if (g.stride > 2) return fail(KWS_ERR_UNSUPPORTED, "dgrad is built for strides 1 and 2");
    const long simds = 4L * cu_count();
    int best = 4;
    long best_cost = -1;
    for (int mw = 4; mw >= 1; --mw) {
        long waves = 0;
        for (int i = 0; i < ncls; ++i) waves += ((rows[i] + 15) / 16 + mw - 1) / mw;
        const long cost = ((waves + simds - 1) / simds) * mw;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mw; }
    }
    switch (best) {
    case 1: return launch_dgrad_mw<CR, CO, 1, STRIDE>(dz, w, dx, g, cls, ncls, max_rows, s);
    case 2: return launch_dgrad_mw<CR, CO, 2, STRIDE>(dz, w, dx, g, cls, ncls, max_rows, s);
    case 3: return launch_dgrad_mw<CR, CO, 3, STRIDE>(dz, w, dx, g, cls, ncls, max_rows, s);
    default: return launch_dgrad_mw<CR, CO, 4, STRIDE>(dz, w, dx, g, cls, ncls, max_rows, s);
    }
}

// 1 (default): conv3, conv4 and the dense layer run as split-precision bf16 products (kws_device.h: mfma_bf16x6, three-way
// split, fp32-level error); 0: every product on the fp32 MFMA.  kws_set_matrix_precision() switches it library-wide.
static int g_matrix_precision = 1;
static inline int matrix_prec(const kws_model *m) { return m->matrix_precision >= 0 ? m->matrix_precision : g_matrix_precision; }
// Storage / matrix-operand precision of simple_cnn_lite INFERENCE (kws_set_inference_precision): 0 = fp32, 1 = fp16
static int g_infer_precision = 0;
static inline int infer_prec(const kws_model *m) { return m->infer_precision >= 0 ? m->infer_precision : g_infer_precision; }

// returns the number of blocks that wrote BatchNorm partial sums (0: `partial` was not given or the grid exceeds its stride)
template <int CR, int CO, int MODE, int EPI>
int launch_bf16(const char *what, const float *src, __bf16 *const planes[6], const float *bias, float *dst, const ConvGeom &g_in,
                hipStream_t s, double *partial = nullptr, const float *shift = nullptr, __bf16 *const *src_planes = nullptr,
                const float *abn = nullptr)
{
    ConvGeom g = g_in;
    const long M = MODE == MODE_FWD ? (long)g.B * g.Ho * g.Wo : (long)g.B * g.H * g.W;
    if (M >= (1L << 31)) return fail(KWS_ERR_UNSUPPORTED, "%s: %ld rows exceed the kernel's 32-bit row index", what, M);
    {
        // small maps with 'same' padding: pixel-major rows let a block skip the taps that are padding for its position (kws_conv.h)
        const int P = MODE == MODE_FWD ? g.Ho * g.Wo : g.H * g.W;
        g.pmajor = (g.KH == 3 && g.KW == 3 && P > 1 && P <= 16) ? 1 : 0;
    }
    static const std::string name = std::string(what) + "<" + std::to_string(CR) + "," + std::to_string(CO) + ">";
    const int o = MODE == MODE_FWD ? 3 : 0;      // forward reads the transposed planes, the data gradient the original order
    const Bf16Planes wp{{planes[o], planes[o + 1], planes[o + 2]}};
    // rows per block = 32 * RT: 96 for conv4's forward (8 column tiles: the larger tile halves the LDS reads per MFMA), 64
    // elsewhere (3 resident blocks per CU overlap their staging and MFMA phases better; measured per kernel at B = 4096)
    constexpr int RT = (MODE == MODE_FWD && CO == 128 && CR == 64) ? 3 : 2;
    const unsigned nblk = blocks_for(M, 32 * RT);
    if (abn) {
        // src is the pre-activation tensor of the BatchNormalization -> ReLU6 in front of this layer (conv4's training forward)
        if constexpr (MODE == MODE_FWD && CR == 64 && CO == 128 && EPI == EPI_RELU) {
            if (!partial || (int)nblk > kStatStride || src_planes) return fail(KWS_ERR_UNSUPPORTED, "%s: activation-on-load form needs the fused statistics", what);
            KWS_LAUNCH(name.c_str(), (conv_bf16_kernel<CR, CO, MODE, EPI, RT, true, false, true>), dim3(nblk), dim3(256), 0, s, src, wp, bias, dst, g, partial,
                       kStatStride, nullptr, (Bf16Planes{{nullptr, nullptr, nullptr}}), abn);
            return (int)nblk;
        } else
            return fail(KWS_ERR_UNSUPPORTED, "%s: no activation-on-load form for this layer", what);
    }
    if (partial && (int)nblk <= kStatStride && !src_planes) {
        KWS_LAUNCH(name.c_str(), (conv_bf16_kernel<CR, CO, MODE, EPI, RT, true>), dim3(nblk), dim3(256), 0, s, src, wp, bias, dst, g, partial,
                   kStatStride);
        return (int)nblk;
    }
    if (src_planes) {                            // the A operand is already split (bn_bwd_apply_planes_kernel)
        if constexpr ((32 * RT * 4) % 256 == 0) {
            const Bf16Planes apl{{src_planes[0], src_planes[1], src_planes[2]}};
            if constexpr (EPI == EPI_BNBWD_GATE6) {
                if (!partial || (int)nblk > kStatStride) return fail(KWS_ERR_UNSUPPORTED, "fused BatchNorm-backward sums need at most %d blocks", kStatStride);
                KWS_LAUNCH(name.c_str(), (conv_bf16_kernel<CR, CO, MODE, EPI, RT, true, true>), dim3(nblk), dim3(256), 0, s, src, wp, bias, dst, g, partial,
                           kStatStride, shift, apl);
                return (int)nblk;
            }
            KWS_LAUNCH(name.c_str(), (conv_bf16_kernel<CR, CO, MODE, EPI, RT, false, true>), dim3(nblk), dim3(256), 0, s, src, wp, bias, dst, g, nullptr,
                       0, shift, apl);
            return 0;
        } else {
            return fail(KWS_ERR_UNSUPPORTED, "pre-split operand planes need 64 rows per block");
        }
    }
    KWS_LAUNCH(name.c_str(), (conv_bf16_kernel<CR, CO, MODE, EPI, RT, false>), dim3(nblk), dim3(256), 0, s, src, wp, bias, dst, g, nullptr, 0, shift);
    return 0;
}

// h/m/l bf16 planes of the three GEMM weight tensors, once per step (one launch)
// training at the default geometry: conv3 / conv4 forward run as clip-group kernels (kws_conv_group.h), whose weights are fragment-major
static bool group_fwd_ok(const kws_model *m)
{
    const CnnDims &d = m->d;
    return m->kind == KWS_SIMPLE_CNN && d.H2 == kFuH2 && d.W2 == kFuW2 && d.H3 == kFuH3 && d.W3 == kFuW3;
}
// non-deterministic training in split precision: Dense forward, head forward / backward and the Dense data gradient are one kernel
// (kws_dense_head.h), whose Dense weights are fragment-major in both orders
static bool dense_head_fused_ok(const kws_model *m, int mprec)
{
    return m->kind == KWS_SIMPLE_CNN && mprec == 1 && !m->deterministic && head_bwd_fuses(m) && m->head_K == kDhK && m->d.flat % (16 * kDhWaves) == 0 &&
           m->d.flat <= 1024;
}
static SplitDescs split_descs(const kws_model *m, const float *params, CnnWs &w, bool group_fwd = false, bool dense_fused = false)
{
    SplitDescs all{};
    const float *src[3] = {params + m->o_k[2], params + m->o_k[3], params + m->o_dk};
    const int taps[3] = {9, 9, m->d.H4 * m->d.W4}, ci[3] = {32, 64, 128}, co[3] = {64, 128, 128};
    const int frag[3] = {group_fwd ? 1 : 0, group_fwd ? 2 : 0, dense_fused ? 1 : 0}, ofrag[3] = {group_fwd ? 1 : 0, group_fwd ? 1 : 0, dense_fused ? 1 : 0};
    for (int t = 0; t < 3; ++t)
        all.d[t] = SplitDesc{src[t], {w.wsp[t][0], w.wsp[t][1], w.wsp[t][2]}, {w.wsp[t][3], w.wsp[t][4], w.wsp[t][5]}, taps[t], ci[t], co[t], frag[t],
                             ofrag[t]};
    all.d[3] = all.d[2];
    return all;
}

static int split_weights(const kws_model *m, const float *params, CnnWs &w, hipStream_t s, bool group_fwd = false, bool dense_fused = false)
{
    const SplitDescs all = split_descs(m, params, w, group_fwd, dense_fused);
    KWS_LAUNCH("weight_split_kernel", weight_split_kernel, dim3(64, 3), dim3(256), 0, s, all);
    return KWS_OK;
}
static int launch_group_conv3(const kws_model *m, int B, CnnWs &w, hipStream_t s, bool fuse_pool2, const BnAccFwd *in2 = nullptr, double *acc3 = nullptr)
{
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(conv3_group_fwd_kernel), 4 * kFuA2)) return rc;
    GroupConv3Args a{};
    a.a2 = w.a[1]; a.z3 = w.z[2]; a.partial = w.partial; a.stride = kStatStride; a.B = B;
    if (in2) a.in = *in2;               // BatchNorm-2's coefficients from conv2's accumulator set
    a.acc = acc3;
    if (fuse_pool2) {            // layer 2's BatchNorm -> ReLU6 -> max-pool happens while the tile is staged (bn_act_pool_kernel<true>'s contract)
        const BnCoef k2 = coef_of(w.coef[1], 32);
        a.z2 = w.z[1]; a.sc2 = k2.scale; a.sh2 = k2.shift; a.a2w = w.a[1]; a.zmax2 = w.zmax2; a.arg2 = w.arg2;
    }
    for (int p = 0; p < 3; ++p) a.f3[p] = w.wsp[0][3 + p];
    KWS_LAUNCH("conv_group_fwd<32,64>", conv3_group_fwd_kernel, dim3(blocks_for(B, kFuClips)), dim3(kGrThreads), (size_t)(4 * kFuA2), s, a);
    return (int)blocks_for(B, kFuClips);
}
static int launch_group_dgrad3(const kws_model *m, int B, CnnWs &w, hipStream_t s, double *acc2 = nullptr)
{
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(conv3_group_dgrad_kernel), 6 * kFuA3P)) return rc;
    GroupDgrad3Args a{};
    a.dz3 = w.gz[2]; a.da2 = w.da[1]; a.B = B;
    a.zmax2 = w.zmax2; a.coef2 = coef_of(w.coef[1], 32).scale; a.acc = acc2;      // acc2: BatchNorm-2's backward reduction in the epilogue
    for (int p = 0; p < 3; ++p) a.fw[p] = w.wsp[0][p];
    KWS_LAUNCH("conv_group_dgrad<64,32>", conv3_group_dgrad_kernel, dim3(blocks_for(B, kFuClips)), dim3(kGrThreads), (size_t)(6 * kFuA3P), s, a);
    return KWS_OK;
}
static int launch_group_dgrad4(const kws_model *m, int B, CnnWs &w, hipStream_t s, double *acc3 = nullptr)
{
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(conv4_group_dgrad_kernel), 6 * kGrD4P)) return rc;
    GroupDgrad4Args a{};
    a.z3 = w.z[2]; a.coef = coef_of(w.coef[2], 64).scale; a.g3 = w.gz[2]; a.partial = w.partial; a.stride = kStatStride; a.B = B;
    a.acc = acc3;                       // acc3: the sums go to the accumulator set, no finalize kernel follows
    for (int p = 0; p < 3; ++p) { a.dz[p] = w.dzp[p]; a.fw[p] = w.wsp[1][p]; }
    KWS_LAUNCH("conv_group_dgrad<128,64>", conv4_group_dgrad_kernel, dim3(blocks_for(B, kFuClips)), dim3(kGrThreads), (size_t)(6 * kGrD4P), s, a);
    return (int)blocks_for(B, kFuClips);
}
static int launch_group_conv4(const kws_model *m, int B, CnnWs &w, hipStream_t s, const BnAccFwd *in3 = nullptr, double *acc4 = nullptr)
{
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(conv4_group_fwd_kernel), 6 * kFuA3P)) return rc;
    GroupConv4Args a{};
    if (in3) a.in = *in3;
    a.acc = acc4;
    const BnCoef k3 = coef_of(w.coef[2], 64);
    a.z3 = w.z[2]; a.sc3 = k3.scale; a.sh3 = k3.shift; a.z4 = w.z[3]; a.partial = w.partial; a.stride = kStatStride; a.B = B;
    for (int p = 0; p < 3; ++p) a.f4[p] = w.wsp[1][3 + p];
    KWS_LAUNCH("conv_group_fwd<64,128>", conv4_group_fwd_kernel, dim3(blocks_for(B, kFuClips)), dim3(kGrThreads), (size_t)(6 * kFuA3P), s, a);
    return (int)blocks_for(B, kFuClips);
}

// Inference of simple_cnn in split precision at the default geometry: everything behind the second pooling stage is ONE kernel
// (kws_infer_fused.h).  Its weights are prepared fragment-major; they take the place of the transposed planes (same element counts), the
// head's go to the head of the double partial slab (unused by the inference forward).
static bool fused_tail_ok(const kws_model *m, bool bf16)
{
    const CnnDims &d = m->d;
    return bf16 && m->kind == KWS_SIMPLE_CNN && d.H2 == kFuH2 && d.W2 == kFuW2 && d.H3 == kFuH3 && d.W3 == kFuW3 && d.H4 == kFuH4 && d.W4 == kFuW4 &&
           m->C <= kFuHeadCols;
}
static __bf16 *fused_head_plane(CnnWs &w, int p) { return reinterpret_cast<__bf16 *>(w.partial) + (size_t)p * (kFuD / 32) * (kFuHeadCols / 16) * 512; }
static int split_weights_fused(const kws_model *m, const float *params, CnnWs &w, hipStream_t s)
{
    FragDescs all{};
    all.d[0] = FragDesc{params + m->o_k[2], {w.wsp[0][3], w.wsp[0][4], w.wsp[0][5]}, 9, kFuC2, kFuC3, kFuC3 / 16, 0};
    all.d[1] = FragDesc{params + m->o_k[3], {w.wsp[1][3], w.wsp[1][4], w.wsp[1][5]}, 9, kFuC3, kFuC4, kFuC4 / 16, 1};
    all.d[2] = FragDesc{params + m->o_dk, {w.wsp[2][3], w.wsp[2][4], w.wsp[2][5]}, kFuH4 * kFuW4, kFuC4, kFuD, kFuD / 16, 0};
    all.d[3] = FragDesc{params + m->o_hk, {fused_head_plane(w, 0), fused_head_plane(w, 1), fused_head_plane(w, 2)}, 1, kFuD, m->C, kFuHeadCols / 16, 0};
    KWS_LAUNCH("infer_frag_kernel", infer_frag_kernel, dim3(32, 4), dim3(256), 0, s, all);
    return KWS_OK;
}
static int launch_fused_tail(const kws_model *m, int B, const float *params, CnnWs &w, float *probs, int32_t *argmax, hipStream_t s)
{
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(infer_tail_kernel), kFuLdsBytes)) return rc;
    FusedTailArgs a{};
    a.a2 = w.a[1];
    for (int p = 0; p < 3; ++p) { a.f3[p] = w.wsp[0][3 + p]; a.f4[p] = w.wsp[1][3 + p]; a.fd[p] = w.wsp[2][3 + p]; a.fh[p] = fused_head_plane(w, p); }
    const BnCoef k3 = coef_of(w.coef[2], 64), k4 = coef_of(w.coef[3], 128);
    a.sc3 = k3.scale; a.sh3 = k3.shift; a.sc4 = k4.scale; a.sh4 = k4.shift;
    a.db = params + m->o_db; a.hb = params + m->o_hb;
    a.probs = probs; a.argmax = argmax; a.B = B; a.C = m->C;
    KWS_LAUNCH("infer_tail_kernel", infer_tail_kernel, dim3(blocks_for(B, kFuClips)), dim3(kFuThreads), (size_t)kFuLdsBytes, s, a);
    return KWS_OK;
}

ConvGeom geom3x3(int B, int H, int W, int stride)
{
    ConvGeom g;
    g.B = B; g.H = H; g.W = W; g.stride = stride; g.KH = 3; g.KW = 3;
    g.Ho = same_out(H, stride); g.Wo = same_out(W, stride);
    g.pt = same_pad_before(H, 3, stride); g.pl = same_pad_before(W, 3, stride);
    return g;
}

// Training forward pass in split precision: 48 extra blocks in the grid of the layer-1 activation kernel
// (l1m_act_pool_moments_kernel<true>) split the conv3 / conv4 / dense weights into their bf16 planes and 16 clear the gradient
// buffer.  Both jobs used to run on the side stream behind an event and were joined before conv3: two events on the main chain
// (6-8 us each) for 12 us of work that hides under the activation pass.
// conv2's backward pass in split precision keeps g compact (the routed value per pool window + the element index), when the clip
// fits the kernels' staging; the forward activation kernel of layer 2 then also leaves zmax / arg for the routed backward reduction
static bool cnn_compact_g2(const kws_model *m, bool bf16)
{
    const CnnDims &d = m->d;
    return bf16 && d.H1 * d.W1 * 8 <= 1280 && (size_t)(d.H1 / 2) * (d.W1 / 2) * 32 <= sizeof(float) * (size_t)d.H3 * d.W3 * 64;
}
// training in split precision: conv4 and its weight gradient form a3 from z3 on the fly (kws_conv.h: ABN / XBN); the fp32 mode keeps
// the activation kernel (same-box A/B at B = 4096: 0.6714 -> 0.666 ms per step)
static bool cnn_a3_on_load(const kws_model *m, bool bf16, bool training) { return bf16 && training; }
constexpr int kPrepSplitBlocks = 16, kPrepZeroBlocks = 16, kPrepBlocks = 3 * kPrepSplitBlocks + kPrepZeroBlocks;

// ---- forward ------------------------------------------------------------------------------------------------
// zero_grads (training, split precision): the gradient buffer of the backward pass that follows is cleared on the side
// stream beside the weight split instead of on the main chain; *zeroed tells the caller whether that happened
int cnn_forward(const kws_model *m, const float *feat, int B, const float *params, float *state, CnnWs &w, bool training,
                uint64_t seed, hipStream_t s, float *zero_grads = nullptr, bool *zeroed = nullptr, OverlapHook *hook = nullptr,
                const double *moments = nullptr, float *probs = nullptr, int32_t *argmax = nullptr, bool *head_done = nullptr)
{
    const DisarmOnExit disarm_guard;        // no armed fork event outlives this call, whichever way it returns
    const CnnDims &d = m->d;
    const int Hs[4] = {d.H0, d.H1, d.H2, d.H3}, Ws[4] = {d.W0, d.W1, d.W2, d.W3};   // conv input sizes
    const int Hz[4] = {d.H0, d.H1, d.H3, d.H3}, Wz[4] = {d.W0, d.W1, d.W3, d.W3};   // conv output sizes
    const bool pool[4] = {true, true, false, true};
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);

    const bool bf16 = matrix_prec(m) == 1;
    // inference after kws_model_prepare_inference on the same buffers: the weight planes and BatchNorm coefficients are in place
    const bool prepared = !training && m->prepared_for(params, state, w.base, B, matrix_prec(m), infer_prec(m));
    // inference with a caller that takes the head's outputs here: conv3 .. softmax as one kernel (kws_infer_fused.h)
    const bool fused_tail = !training && head_done && fused_tail_ok(m, bf16);
    // training in split precision at the default geometry: conv3 / conv4 forward as clip-group kernels (needs at most kStatStride blocks)
    const bool group_fwd = training && bf16 && group_fwd_ok(m) && (long)blocks_for(B, kFuClips) <= kStatStride;
    // training: the Dense layer's forward product runs inside the fused Dense + head kernel of the backward pass (kws_dense_head.h)
    const bool dense_fused = training && dense_head_fused_ok(m, matrix_prec(m));
    ModelRes *R = nullptr;       // only the split-on-the-side-stream branch below needs the model's stream / events
    // training in split precision at a geometry the MFMA layer-1 kernels cover: the weight split and the gradient clear ride
    // in the grid of the layer-1 activation kernel -- no side-stream branch, no events
    const bool prep_in_stats = bf16 && training && d.H0 % 2 == 0 && d.W0 % 2 == 0 && (d.H0 + 2) * (d.W0 + 2) <= 64 * kL1Stage &&
                               (d.H0 / 2) * (d.W0 / 2) <= 4 * kL1MaxTiles;
    if (prep_in_stats) {
        if (zero_grads && zeroed) *zeroed = true;
    } else if (bf16 && training) {
        // the h/m/l planes are first needed by conv3: split on the side stream beside layers 1-2, join before conv3
        R = const_cast<kws_model *>(m)->dev_res();
        if (!R) return fail(KWS_ERR_HIP, "cannot create the model's side stream / events on this device");
        hipStream_t s2 = R->side;
        KWS_HIP_CHECK(hipEventRecord(R->ev[10], s));
        KWS_HIP_CHECK(hipStreamWaitEvent(s2, R->ev[10], 0));
        if (zero_grads) {
            KWS_HIP_CHECK(hipMemsetAsync(zero_grads, 0, sizeof(float) * (size_t)m->P, s2));
            if (zeroed) *zeroed = true;
        }
        if (int rc = split_weights(m, params, w, s2, group_fwd, dense_fused)) return rc;
        KWS_HIP_CHECK(hipEventRecord(R->ev[11], s2));
    } else if (bf16) {
        // inference stays on ONE stream: callers capture it into hipGraphs, and a fork to the library's side stream inside
        // several captured graphs made every graph after the first replay 0.2 ms slower
        if (!prepared)
            if (int rc = fused_tail ? split_weights_fused(m, params, w, s) : split_weights(m, params, w, s)) return rc;
    }
    if (!training && !prepared)
        if (int rc = infer_coefs(m, params, state, w, s)) return rc;
    // layer 1: conv1 is recomputed from the feature map wherever z1 is needed (kws_layer1.h)
    {
        const int cpb = std::max(1, (B + kMaxStatBlocks - 1) / kMaxStatBlocks), nb = (B + cpb - 1) / cpb;
        const size_t smem1 = sizeof(float) * (size_t)(d.H0 + 2) * (d.W0 + 2);
        // every pixel in a pool window, a haloed map of at most 768 floats and at most 40 tiles: the MFMA, wave-per-clip forms
        const bool l1m = d.H0 % 2 == 0 && d.W0 % 2 == 0 && (d.H0 + 2) * (d.W0 + 2) <= 64 * kL1Stage &&
                         (d.H0 / 2) * (d.W0 / 2) <= 4 * kL1MaxTiles;
        const int cpw = std::max(1, (B + 4 * kMaxStatBlocks - 1) / (4 * kMaxStatBlocks)), nbm = (B + 4 * cpw - 1) / (4 * cpw);
        // per-wave LDS tiles; the compile-time form of the default map reads two rows past the haloed map (kws_layer1.h: L1Runs)
        const size_t smemm = sizeof(float) * 4 * (d.H0 == 30 && d.W0 == 20 ? (size_t)L1Runs<30, 20>::TILE : (size_t)((((d.H0 + 2) * (d.W0 + 2)) + 3) & ~3));
        const long M1 = (long)B * d.H0 * d.W0;
        BnCoef k1 = coef_of(w.coef[0], 16);
        const float *kern1 = params + m->o_k[0];
        if (training && l1m) {
            // batch statistics of BatchNorm 1 from the second moments Q of the features (kws_layer1_moments.h): no statistics pass
            // over z1, no finalize launch; Q comes from the caller (input pipeline) or is computed here
            const double *q = moments;
            if (!q) {
                KWS_LAUNCH("l1_moments_kernel", l1_moments_kernel, dim3(nbm), dim3(256), smemm, s, feat, B, d.H0, d.W0, cpw, w.partial);
                KWS_LAUNCH("l1_moments_finalize_kernel", l1_moments_finalize_kernel, dim3(kMomCount), dim3(64), 0, s, w.partial, nbm, w.moments);
                q = w.moments;
            }
            L1PrepArgs pa{};
            if (prep_in_stats) {
                pa.all = split_descs(m, params, w, group_fwd, dense_fused); pa.zero_buf = zero_grads; pa.zero_n = (long)m->P;
                pa.nsplit = kPrepSplitBlocks; pa.nzero = kPrepZeroBlocks;
                KWS_LAUNCH("l1m_act_pool_kernel", l1m_act_pool_moments_kernel<true>, dim3(nbm + kPrepBlocks), dim3(256), smemm, s, feat, kern1, q,
                           params + m->o_g[0], params + m->o_b[0], state + m->o_mm[0], state + m->o_mv[0], k1, w.a[0], B, d.H0, d.W0, cpw, nbm, pa);
            } else {
                KWS_LAUNCH("l1m_act_pool_kernel", l1m_act_pool_moments_kernel<false>, dim3(nbm), dim3(256), smemm, s, feat, kern1, q,
                           params + m->o_g[0], params + m->o_b[0], state + m->o_mm[0], state + m->o_mv[0], k1, w.a[0], B, d.H0, d.W0, cpw, nbm, pa);
            }
        } else if (training) {
            KWS_LAUNCH("l1_stats_kernel", l1_stats_kernel<16>, dim3(nb), dim3(256), smem1, s, feat, kern1, B, d.H0, d.W0, cpb, w.partial);
            KWS_LAUNCH(prof_name("bn_finalize_train_kernel", 1), bn_finalize_train_kernel, dim3(16), dim3(64), 0, s, w.partial, nb, M1, 16,
                       params + m->o_g[0], params + m->o_b[0], state + m->o_mm[0], state + m->o_mv[0], k1);
        }
        if (training && l1m) ;
        else if (l1m) KWS_LAUNCH("l1m_act_pool_kernel", l1m_act_pool_kernel, dim3(nbm), dim3(256), smemm, s, feat, kern1, k1.scale, k1.shift, w.a[0], B,
                   d.H0, d.W0, cpw);
        else KWS_LAUNCH("l1_act_pool_kernel", l1_act_pool_kernel<16>, dim3(nb), dim3(256), smem1, s, feat, kern1, k1.scale, k1.shift, w.a[0], B,
                   d.H0, d.W0, cpb);
    }
    if (hook) KWS_TRY(hook->fire(8, s));                 // behind layer 1: the featurizer then shares the chip with the forward convolutions
    bool bound6 = false;
    const bool a3_on_load = cnn_a3_on_load(m, bf16, training);
    const bool routed_bwd2 = cnn_compact_g2(m, bf16);
    const bool fuse_pool2 = group_fwd && routed_bwd2 && d.H1 == kGrH1 && d.W1 == kGrW1;
    // finalize-free batch statistics (kws_device.h: acc_add), non-deterministic training at the default geometry: conv2 .. conv4 add their
    // sums to accumulator sets and the next kernel of the chain derives scale / shift in its prologue -- three launches less
    const bool acc_fwd = fuse_pool2 && !m->deterministic && kCh[4] <= 128 && !stream_capturing(s);
    unsigned fpar[4] = {0, 0, 0, 0};
    if (acc_fwd) {
        if (!R) R = const_cast<kws_model *>(m)->dev_res();
        if (!R) return fail(KWS_ERR_HIP, "cannot create the model's side stream / events on this device");
        for (int l = 1; l < 4; ++l) fpar[l] = R->acc_uses[0][l]++;
        KWS_TRY(acc_make_clean(R, s));
        R->acc_dirty = true;               // until every kernel of this pass is enqueued
    }
    auto acc_in = [&](int l) {             // the consumer's view of layer l's statistics
        return BnAccFwd{acc_set(R, 0, l, fpar[l]), acc_set(R, 0, l, fpar[l] + 1), (long)B * Hz[l] * Wz[l], params + m->o_g[l], params + m->o_b[l],
                        state + m->o_mm[l], state + m->o_mv[l], coef_of(w.coef[l], kCh[l + 1])};
    };
    for (int l = 1; l < 4; ++l) {
        const float *in = w.a[l - 1];
        const float *kern = params + m->o_k[l];
        const long M = (long)B * Hz[l] * Wz[l];
        const int C = kCh[l + 1];
        int fused_stat_blocks = 0;           // > 0: the conv kernel already wrote the BN partial sums
        if (l == 1) {
            // conv2: clip-resident kernel, batch statistics fused into the epilogue when training
            const unsigned nblk = (unsigned)std::min(B, kMaxStatBlocks);
            const size_t sm = std::max(sizeof(float) * (size_t)(Hs[1] + 2) * (Ws[1] + 2) * 20, sizeof(double) * 4 * 2 * 32);
            const size_t smb = std::max((size_t)6 * 16 * (((Hs[1] + 2) * (Ws[1] + 2) + 15) & ~15), sizeof(double) * 4 * 2 * 16);
            if (training && bf16) {
                KWS_LAUNCH("conv_fwd_clip_bf16<16,32>", (conv_fwd_clip_bf16_kernel<true>), dim3(nblk), dim3(256), smb, s, in, kern, w.z[1], B, Hs[1],
                           Ws[1], w.partial, kStatStride, nullptr, nullptr, acc_fwd ? acc_set(R, 0, 1, fpar[1]) : nullptr);
                fused_stat_blocks = (int)nblk;
            } else if (training) {
                KWS_LAUNCH("conv_fwd_clip<16,32>", (conv_fwd_clip_kernel<32, true>), dim3(nblk), dim3(256), sm, s, in, kern, w.z[1], B, Hs[1], Ws[1],
                           w.partial, kStatStride);
                fused_stat_blocks = (int)nblk;
            } else if (bf16) {
                const BnCoef k2 = coef_of(w.coef[1], 32);
                KWS_LAUNCH("conv_fwd_clip_pool_bf16<16,32>", (conv_fwd_clip_bf16_kernel<false, true>), dim3(nblk), dim3(256), smb, s, in, kern, w.a[1], B,
                           Hs[1], Ws[1], w.partial, kStatStride, k2.scale, k2.shift);
                if (fused_tail) {
                    KWS_TRY(launch_fused_tail(m, B, params, w, probs, argmax, s));
                    *head_done = true;
                    KWS_LAUNCH_CHECK("simple_cnn forward");
                    return KWS_OK;
                }
                continue;
            } else {
                // inference: BatchNorm affine + ReLU6 + 2x2 max in the kernel's epilogue, a2 written directly
                const BnCoef k2 = coef_of(w.coef[1], 32);
                KWS_LAUNCH("conv_fwd_clip_pool<16,32>", (conv_fwd_clip_kernel<32, false, true>), dim3(nblk), dim3(256), sm, s, in, kern, w.a[1], B,
                           Hs[1], Ws[1], w.partial, kStatStride, k2.scale, k2.shift);
                continue;
            }
        } else if (l == 2 && bf16 && !training) {
            // inference: BatchNorm affine + ReLU6 in the epilogue, a3 written directly (conv3 has no pool)
            const BnCoef k3 = coef_of(w.coef[2], 64);
            KWS_TRY_NB(launch_bf16<32, 64, MODE_FWD, EPI_BN_RELU6>("conv_bf16_fwd_bn", in, w.wsp[0], k3.scale, w.a[2], geom3x3(B, Hs[2], Ws[2], 2), s, nullptr, k3.shift));
            continue;
        } else if (l == 2) {
            if (bf16 && training && !prep_in_stats) KWS_HIP_CHECK(hipStreamWaitEvent(s, R->ev[11], 0));     // the weight planes are ready
            // the split-precision kernels write the BatchNorm partial sums from their epilogue when training
            if (group_fwd) {
                if (acc_fwd) { const BnAccFwd in2 = acc_in(1); fused_stat_blocks = launch_group_conv3(m, B, w, s, fuse_pool2, &in2, acc_set(R, 0, 2, fpar[2])); }
                else fused_stat_blocks = launch_group_conv3(m, B, w, s, fuse_pool2);
                if (fused_stat_blocks < 0) return fused_stat_blocks;
            } else if (bf16) {
                fused_stat_blocks = launch_bf16<32, 64, MODE_FWD, EPI_NONE>("conv_bf16_fwd", in, w.wsp[0], nullptr, w.z[2], geom3x3(B, Hs[2], Ws[2], 2), s,
                                                                          training ? w.partial : nullptr);
                if (fused_stat_blocks < 0) return fused_stat_blocks;
            } else KWS_TRY(launch_gemm<32, 64, MODE_FWD, EPI_NONE>(in, kern, nullptr, w.z[2], geom3x3(B, Hs[2], Ws[2], 2), s));
        } else {
            // activation='relu', cnn.py:55
            if (group_fwd) {
                // forms a3 = relu6(BN3(z3)) while staging, like the ABN form below
                if (acc_fwd) { const BnAccFwd in3 = acc_in(2); fused_stat_blocks = launch_group_conv4(m, B, w, s, &in3, acc_set(R, 0, 3, fpar[3])); }
                else fused_stat_blocks = launch_group_conv4(m, B, w, s);
                if (fused_stat_blocks < 0) return fused_stat_blocks;
            } else if (bf16) {
                // training: a3 = relu6(BN3(z3)) is never written -- conv4 forms it from z3 while it stages its rows, and so does conv4's
                // weight gradient (conv3 has no pooling, so the activation is a per-element map)
                fused_stat_blocks = launch_bf16<64, 128, MODE_FWD, EPI_RELU>("conv_bf16_fwd", a3_on_load ? w.z[2] : in, w.wsp[1], nullptr, w.z[3],
                                                                           geom3x3(B, Hs[3], Ws[3], 1), s, training ? w.partial : nullptr, nullptr, nullptr,
                                                                           a3_on_load ? coef_of(w.coef[2], 64).scale : nullptr);
                if (fused_stat_blocks < 0) return fused_stat_blocks;
            } else KWS_TRY(launch_gemm<64, 128, MODE_FWD, EPI_RELU>(in, kern, nullptr, w.z[3], geom3x3(B, Hs[3], Ws[3], 1), s));
        }
        if (hook && l < 3) KWS_TRY(hook->fire(8 + l, s));     // 9 behind conv2's forward, 10 behind conv3's
        BnCoef k = coef_of(w.coef[l], C);
        if (training && acc_fwd) ;          // the next kernel of the chain derives the coefficients from the accumulator set
        else if (training) {
            int nblk, rows;
            stat_grid(M, C, nblk, rows);
            if (fused_stat_blocks) nblk = fused_stat_blocks;
            else KWS_LAUNCH(prof_name("channel_stats_kernel", l + 1), channel_stats_kernel, dim3(nblk), dim3(256), 0, s, w.z[l], M, C, rows, w.partial);
            KWS_LAUNCH(prof_name("bn_finalize_train_kernel", l + 1), bn_finalize_train_kernel, dim3(C), dim3(64), 0, s, w.partial, nblk, M, C, params + m->o_g[l],
                               params + m->o_b[l], state + m->o_mm[l], state + m->o_mv[l], k);
        }
        const float rate = (training && l == 3 && seed != 0) ? 0.5f : 0.f;   // Dropout(0.5) after Flatten, cnn.py:63
        // behind the last convolution: from here to BN4's backward the main chain is small kernels (activation, dense, head),
        // the best place for the caller to start the next batch's featurizer (kws_train_args.overlap_event)
        if (l == 3 && hook) KWS_TRY(hook->fire(0, s));
        // overlap point 6 sits right behind this layer's activation kernel: the caller's event rides on that kernel's completion signal
        // instead of a marker packet of its own (kws_common.h: ArmedEvent)
        // layer 4's activation rides in the fused Dense + head kernel of the backward pass (kws_dense_head.h: z4): no kernel here
        const bool pool4_fused = l == 3 && acc_fwd && dense_fused;
        const bool arm6 = l == 3 && hook && hook->wants(6) && hook->ev && !pool4_fused;
        if (arm6) arm_stop_event(hook->ev, s);
        if (pool4_fused) { R->pool4_pending = true; R->pool4_mm = state + m->o_mm[3]; R->pool4_mv = state + m->o_mv[3]; }
        else if (l == 1 && fuse_pool2) ;               // conv3's group kernel forms a2 (and zmax2 / arg2) from z2 while it stages its tile
        else if (pool[l]) {
            const long total = (long)B * (Hz[l] / 2) * (Wz[l] / 2) * C;
            // training: the routed element of every window for the backward reduction (layer 2: compact g; layer 4: full-size g)
            float *zm = nullptr;
            unsigned char *ag = nullptr;
            if (l == 1 && training && routed_bwd2) { zm = w.zmax2; ag = w.arg2; }
            if (l == 3 && training && bf16) { zm = w.zmax4; ag = w.arg4; }
            if (l == 3 && acc_fwd)
                KWS_LAUNCH(prof_name("bn_act_pool_kernel", l + 1), bn_act_pool_acc_kernel, dim3(std::min<unsigned>(1024u, blocks_for(total, 256))), dim3(256), 0, s,
                           w.z[l], acc_in(3), w.a[l], B, Hz[l], Wz[l], C, rate, slo, shi, zm, ag);
            else
                KWS_LAUNCH(prof_name("bn_act_pool_kernel", l + 1), bn_act_pool_kernel<true>, dim3(blocks_for(total, 256)), dim3(256), 0, s, w.z[l], k.scale, k.shift,
                           w.a[l], B, Hz[l], Wz[l], C, rate, slo, shi, zm, ag);
        } else if (!(l == 2 && a3_on_load)) {
            const long total = M * C;
            KWS_LAUNCH(prof_name("bn_act_pool_kernel", l + 1), bn_act_pool_kernel<false>, dim3(blocks_for(total, 256)), dim3(256), 0, s, w.z[l], k.scale, k.shift,
                               w.a[l], B, Hz[l], Wz[l], C, rate, slo, shi);
        }
        if (arm6) bound6 = stop_event_bound(hook->ev);
    }
    // Dense(128, use_bias=True) + ReLU6 as a (H4 x W4) 'valid' convolution over the pooled map (Flatten is h,w,c)
    ConvGeom g;
    g.B = B; g.H = d.H4; g.W = d.W4; g.Ho = 1; g.Wo = 1; g.stride = 1; g.pt = 0; g.pl = 0; g.KH = d.H4; g.KW = d.W4;
    if (hook) KWS_TRY(hook->fire(6, s, bound6));
    if (dense_fused) ;                              // d1 is formed by dense_head_fused_kernel (cnn_backward)
    else if (bf16) KWS_TRY_NB(launch_bf16<128, 128, MODE_FWD, EPI_BIAS_RELU6>("conv_bf16_fwd", w.a[3], w.wsp[2], params + m->o_db, w.d1, g, s));
    else KWS_TRY(launch_gemm<128, 128, MODE_FWD, EPI_BIAS_RELU6>(w.a[3], params + m->o_dk, params + m->o_db, w.d1, g, s));
    if (hook) KWS_TRY(hook->fire(7, s));
    KWS_LAUNCH_CHECK("simple_cnn forward");
    if (acc_fwd) R->acc_dirty = false;
    return KWS_OK;
}

// ---- backward -----------------------------------------------------------------------------------------------
int cnn_backward(const kws_model *m, const float *feat, int B, const float *params, float *grads, CnnWs &w, uint64_t seed,
                 hipEvent_t bucket_event, hipStream_t s, float *stats, bool grads_zeroed = false, const double *moments = nullptr,
                 OverlapHook *hook = nullptr, kws_comm *comm = nullptr, const kws_train_args *fused_head = nullptr)
{
    // fused_head != nullptr: the head's forward pass (logits, softmax, loss, dlogits) has NOT run; the head's backward kernel does it
    // (cnn_head_fwd_fused below decides)
    const DisarmOnExit disarm_guard;        // no armed fork event outlives this call, whichever way it returns
    const CnnDims &d = m->d;
    const int Hs[4] = {d.H0, d.H1, d.H2, d.H3}, Ws[4] = {d.W0, d.W1, d.W2, d.W3};
    const int Hz[4] = {d.H0, d.H1, d.H3, d.H3}, Wz[4] = {d.W0, d.W1, d.W3, d.W3};
    const bool pool[4] = {true, true, false, true};
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);

    if (!grads_zeroed) KWS_HIP_CHECK(hipMemsetAsync(grads, 0, sizeof(float) * (size_t)m->P, s));
    // The weight-gradient GEMM of a layer only READS (x, dz) and adds into grads; the data-gradient GEMM and the next
    // layer's BN backward do not depend on it.  Each of them alone leaves the MFMA pipe more than half idle, so wgrad runs
    // on the library's side stream (fork after dz is final, one join at the end) and shares the chip with the main chain.
    ModelRes *R = const_cast<kws_model *>(m)->dev_res();
    if (!R) return fail(KWS_ERR_HIP, "cannot create the model's side stream / events on this device");
    hipStream_t s2 = R->side;
    const int mprec = matrix_prec(m);
    const bool det = m->deterministic != 0;
    // fork: the side stream waits for everything enqueued on s so far.  arm(ev) in front of the LAST kernel before the fork lets that
    // kernel's own completion signal be the event (kws_common.h: ArmedEvent) instead of a marker packet on the main chain
    constexpr bool no_arm = false;
    hipEvent_t fork0_ev = nullptr;
    bool fork0_bound = false, dense_fused = false;
    auto arm = [&](int ev) { if (!no_arm) arm_stop_event(R->ev[ev], s); };
    auto fork = [&](int ev) -> int {
        if (!stop_event_bound(R->ev[ev])) KWS_HIP_CHECK(hipEventRecord(R->ev[ev], s));
        KWS_HIP_CHECK(hipStreamWaitEvent(s2, R->ev[ev], 0));
        return KWS_OK;
    };
    const bool routed_bwd2 = cnn_compact_g2(m, mprec == 1);      // the forward pass left zmax2 / arg2 (same predicate)
    // the clip-group form of conv4's / conv3's data gradients (the forward pass prepared the weights for it: cnn_forward's group_fwd)
    const bool group_bwd = mprec == 1 && group_fwd_ok(m) && (long)blocks_for(B, kFuClips) <= kStatStride;
    // finalize-free BatchNorm backward (kws_device.h: acc_add), non-deterministic mode.  Layer 2: conv3's data gradient does the reduction
    // in its epilogue and conv2's clip kernels derive k2 / k3 from the accumulator set.  Layer 3: conv4's data gradient adds its sums to the
    // set and the apply kernel derives the coefficients.  Layer 4: the fused Dense + head kernel's epilogue is the reduction, the apply
    // kernel expands the compact gradient.
    const bool acc_ok = !det && !stream_capturing(s);
    const bool acc_bn2 = group_bwd && acc_ok && routed_bwd2 && Hs[1] * Ws[1] <= 160 && Hs[1] * Ws[1] * 8 <= 1280 && Hs[1] * Ws[1] * 4 <= 768;
    const bool acc_bn3 = group_bwd && acc_ok;
    const bool acc_bn4 = fused_head && dense_head_fused_ok(m, mprec) && acc_ok && kCh[4] == kDhK && d.flat == d.H4 * d.W4 * kCh[4];
    // parity of each layer's sets: flipped only by a pass that uses them (its consumer is what clears the other parity)
    const unsigned bpar[4] = {0u, acc_bn2 ? R->acc_uses[1][1]++ : 0u, acc_bn3 ? R->acc_uses[1][2]++ : 0u, acc_bn4 ? R->acc_uses[1][3]++ : 0u};
    KWS_TRY(acc_make_clean(R, s));
    R->acc_dirty = true;               // until every kernel of this pass is enqueued
    // head: dW2, db2, dd1 (gated by dense's ReLU6)
    {
        // the MFMA head kernel also leaves the dense bias gradient (column sums of dd1) and the loss / accuracy sums
        const bool fuse = head_bwd_fuses(m);
        // the head's backward kernel is the last one in front of the dense fork AND of overlap point 2: its completion signal carries the
        // caller's overlap event when there is one (the side stream then waits for that event too), the fork event otherwise
        // (deterministic mode launches two kernels here and keeps the recorded events)
        fork0_ev = (hook && hook->wants(2) && hook->ev) ? hook->ev : R->ev[0];
        if (!det && !no_arm) arm_stop_event(fork0_ev, s);
        if (fused_head && dense_head_fused_ok(m, mprec)) {
            const kws_train_args *a = fused_head;
            DenseHeadArgs da{};
            da.a4 = w.a[3]; da.db = params + m->o_db; da.w2 = params + m->o_hk;
            for (int p = 0; p < 3; ++p) { da.fd[p] = w.wsp[2][3 + p]; da.fo[p] = w.wsp[2][p]; }
            da.d1 = w.d1; da.dd1 = w.dd1; da.da4 = w.da4; da.dw2 = grads + m->o_hk; da.db2 = grads + m->o_hb; da.ddb = grads + m->o_db;
            da.B = B; da.C = m->C; da.flat = d.flat;
            if (acc_bn4) {
                da.zmax4 = w.zmax4; da.coef4 = coef_of(w.coef[3], 128).scale; da.acc4 = acc_set(R, 1, 3, bpar[3]);
                da.drop_rate = seed != 0 ? 0.5f : 0.f; da.seed_lo = slo; da.seed_hi = shi;
            }
            if (R->pool4_pending) {             // the forward pass of this step left layer 4's activation to this kernel
                const unsigned fpar = R->acc_uses[0][3] - 1;
                da.z4 = w.z[3]; da.a4w = w.a[3]; da.zmax4w = w.zmax4; da.arg4w = w.arg4; da.H3 = Hz[3]; da.W3 = Wz[3];
                da.in4 = BnAccFwd{acc_set(R, 0, 3, fpar), acc_set(R, 0, 3, fpar + 1), (long)B * Hz[3] * Wz[3], params + m->o_g[3], params + m->o_b[3],
                                  R->pool4_mm, R->pool4_mv, coef_of(w.coef[3], 128)};
                da.drop_rate = seed != 0 ? 0.5f : 0.f; da.seed_lo = slo; da.seed_hi = shi;
                R->pool4_pending = false;
            }
            da.fw = HeadFwdArgs{params + m->o_hb, a->labels, a->class_weights, a->probs, w.loss_i, w.correct_i, a->grad_scale / (float)B, a->ignore_index};
            const size_t smem = sizeof(float) * (size_t)(16 * (d.flat + 8) + 2 * 16 * kDhKS + 16 * kDhCS + kDhK * kDhCS);
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(dense_head_fused_kernel), (int)smem)) return rc;
            KWS_LAUNCH("dense_head_fused_kernel", dense_head_fused_kernel, dim3(blocks_for(B, 16)), dim3(kDhThreads), smem, s, da);
            dense_fused = true;
        } else if (fused_head) {
            const kws_train_args *a = fused_head;
            const HeadFwdArgs hf{params + m->o_hb, a->labels, a->class_weights, a->probs, w.loss_i, w.correct_i, a->grad_scale / (float)B, a->ignore_index};
            KWS_TRY(run_head_bwd(m, B, params, w.d1, nullptr, w.dd1, grads, true, s, grads + m->o_db, nullptr, nullptr, nullptr, false, &hf));
        } else
            KWS_TRY(run_head_bwd(m, B, params, w.d1, w.dlogits, w.dd1, grads, true, s, fuse && !det ? grads + m->o_db : nullptr, w.loss_i, w.correct_i,
                                 fuse ? stats : nullptr, det));
        fork0_bound = stop_event_bound(fork0_ev);
        if (fused_head) {
            // the forward pass and the loss are enqueued only now
            if (fused_head->forward_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(fused_head->forward_event), s));
            if (hook) KWS_TRY(hook->fire(1, s));
        }
    }
    // dense 256->128: bias grad (column sums), wgrad, dgrad -> da4 (gradient w.r.t. the dropped, flattened map)
    {
        ConvGeom g;
        g.B = B; g.H = d.H4; g.W = d.W4; g.Ho = 1; g.Wo = 1; g.stride = 1; g.pt = 0; g.pl = 0; g.KH = d.H4; g.KW = d.W4;
        const bool hook_ev_here = hook && hook->wants(2) && hook->ev;
        if (hook) KWS_TRY(hook->fire(2, s, fork0_bound && hook_ev_here));
        // its own fork: started later, together with conv4's weight gradient, the step was 2 % slower (same-box A/B)
        if (hook_ev_here) {
            KWS_HIP_CHECK(hipStreamWaitEvent(s2, fork0_ev, 0));       // the overlap event (recorded or bound just above) is the fork event
        } else {
            if (!fork0_bound) KWS_HIP_CHECK(hipEventRecord(R->ev[0], s));
            KWS_HIP_CHECK(hipStreamWaitEvent(s2, R->ev[0], 0));
        }
        // fused head: the sums of the per-sample losses / correct flags, in a fixed order, off the main chain
        if (fused_head && stats) KWS_LAUNCH("loss_reduce_kernel", loss_reduce_kernel, dim3(1), dim3(256), 0, s2, w.loss_i, w.correct_i, B, stats);
        KWS_TRY(launch_wgrad<128, 128, 1>(w.a[3], w.dd1, grads + m->o_dk, g, s2, det));
        int nblk, rows;
        stat_grid(B, 128, nblk, rows);
        if (!head_bwd_fuses(m) || det) {
            KWS_LAUNCH("channel_stats_kernel.dense", channel_stats_kernel, dim3(nblk), dim3(256), 0, s, w.dd1, (long)B, 128, rows, w.partial);
            KWS_LAUNCH("colsum_finalize_kernel", colsum_finalize_kernel, dim3(128), dim3(256), 0, s, w.partial, nblk, 128, grads + m->o_db);
        }
        if (dense_fused) ;                          // da4 came out of dense_head_fused_kernel
        else if (mprec == 1) KWS_TRY_NB(launch_bf16<128, 128, MODE_DGRAD, EPI_NONE>("conv_bf16_dgrad", w.dd1, w.wsp[2], nullptr, w.da4, g, s));
        else KWS_TRY(launch_dgrad<128, 128, 1>(w.dd1, params + m->o_dk, w.da4, g, s));
        if (hook) KWS_TRY(hook->fire(3, s));
    }
    int fused_bn3_blocks = 0;          // > 0: conv4's data gradient already did layer 3's BatchNorm-backward reduction
    for (int l = 3; l >= 1; --l) {
        if (hook && l == 2) KWS_TRY(hook->fire(5, s));
        const int C = kCh[l + 1];
        const long M = (long)B * Hz[l] * Wz[l];
        const float *da = l == 3 ? w.da4 : w.da[l];
        BnCoef k = coef_of(w.coef[l], C);
        int nblk, rows;
        stat_grid(M, C, nblk, rows);
        // the forward applied dropout to a[3]; its mask is re-derived from the seed here
        const float rate = (l == 3 && seed != 0) ? 0.5f : 0.f;
        // conv2 (split precision): g stays compact -- the routed value per pool window in place of da[1], the element index as
        // a byte in the (dead by now) da[2] buffer -- and the clip data gradient rebuilds it while staging: 54 MB less to
        // write here and 54 MB less to read there at B = 4096
        const bool compact_g = l == 1 && cnn_compact_g2(m, mprec == 1);
        if (pool[l]) {
            const long NW = (long)B * (Hz[l] / 2) * (Wz[l] / 2);       // one thread per (pool window, channel)
            stat_grid(NW, C, nblk, rows);
            if (l == 1 && acc_bn2) ;                            // done by conv3_group_dgrad_kernel's epilogue
            else if (l == 3 && acc_bn4) ;                       // done by dense_head_fused_kernel's epilogue
            else if (compact_g && routed_bwd2)
                KWS_LAUNCH(prof_name("bn_bwd_reduce_pool_kernel", l + 1), bn_bwd_reduce_routed_kernel, dim3(nblk), dim3(256), 0, s, w.zmax2, w.da[1], k, NW, C,
                           rows, w.partial);
            else if (compact_g)
                KWS_LAUNCH(prof_name("bn_bwd_reduce_pool_kernel", l + 1), bn_bwd_reduce_pool_kernel<true>, dim3(nblk), dim3(256), 0, s, w.z[l], w.da[1], k,
                           w.gz[l], B, Hz[l], Wz[l], C, rows, w.partial, rate, slo, shi, reinterpret_cast<unsigned char *>(w.da[2]));
            else if (l == 3 && mprec == 1)
                KWS_LAUNCH(prof_name("bn_bwd_reduce_pool_kernel", l + 1), bn_bwd_reduce_routed_full_kernel, dim3(nblk), dim3(256), 0, s, w.zmax4, w.arg4, da, k,
                           w.gz[l], B, Hz[l], Wz[l], C, rows, w.partial, rate, slo, shi);
            else
                KWS_LAUNCH(prof_name("bn_bwd_reduce_pool_kernel", l + 1), bn_bwd_reduce_pool_kernel<false>, dim3(nblk), dim3(256), 0, s, w.z[l],
                           const_cast<float *>(da), k, w.gz[l], B, Hz[l], Wz[l], C, rows, w.partial, rate, slo, shi);
        } else if (l == 2 && fused_bn3_blocks > 0)
            nblk = fused_bn3_blocks;
        else
            KWS_LAUNCH(prof_name("bn_bwd_reduce_kernel", l + 1), bn_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, s, w.z[l], da, k, w.gz[l], B, Hz[l], Wz[l], C,
                       rows, w.partial, rate, slo, shi);
        // conv2's early weight gradient forks right behind this finalize kernel
        const bool wgrad_early_l1 = compact_g && Hs[1] * Ws[1] <= 160 && Hs[1] * Ws[1] * 8 <= 1280 && Hs[1] * Ws[1] * 4 <= 768;
        const bool acc_l = (l == 1 && acc_bn2) || (l == 2 && acc_bn3 && fused_bn3_blocks > 0) || (l == 3 && acc_bn4);
        const BnAccBwd ab{acc_set(R, 1, l, bpar[l]), acc_set(R, 1, l, bpar[l] + 1), M, grads + m->o_g[l], grads + m->o_b[l]};
        if (acc_l) ;                                            // the consumers derive k2 / k3 themselves (layer 2: fork(1) was armed in front of conv3's data gradient)
        else {
            if (l == 1 && wgrad_early_l1) arm(1);
            KWS_LAUNCH(prof_name("bn_bwd_finalize_kernel", l + 1), bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, w.partial, nblk, M, C, params + m->o_g[l],
                       grads + m->o_g[l], grads + m->o_b[l], k);
        }
        if (l != 1) arm(l);                                 // the apply kernel below is the last one in front of fork(l)
        if (l == 1)
            ;                                                   // fused into conv_dgrad_clip's staging below
        else if (l == 3 && acc_bn4)
            KWS_LAUNCH(prof_name("bn_bwd_apply_planes_kernel", l + 1), bn_bwd_apply_routed_planes_kernel<true>, dim3(std::min<unsigned>(1024u, blocks_for(M * C / 4, 256))),
                       dim3(256), 0, s, w.z[l], w.da4, w.arg4, k, params + m->o_g[l], B, Hz[l], Wz[l], C, ab, (Bf16PlanesOut{{w.dzp[0], w.dzp[1], w.dzp[2]}}));
        else if (l == 2 && acc_l)
            KWS_LAUNCH(prof_name("bn_bwd_apply_kernel", l + 1), bn_bwd_apply_acc_kernel<false>, dim3(std::min<unsigned>(1024u, blocks_for(M * C / 4, 256))), dim3(256), 0, s,
                       w.z[l], w.gz[l], k, params + m->o_g[l], M * C / 4, C, ab);
        else if (l == 3 && mprec == 1)
            // split precision: dz4 leaves as bf16 h/m/l planes, which is what both of its consumers stage (no fp32 dz4)
            KWS_LAUNCH(prof_name("bn_bwd_apply_planes_kernel", l + 1), bn_bwd_apply_planes_kernel<true>, dim3(blocks_for(M * C / 4, 256)), dim3(256), 0, s,
                       w.z[l], w.gz[l], k, params + m->o_g[l], M * C / 4, C, (Bf16PlanesOut{{w.dzp[0], w.dzp[1], w.dzp[2]}}));
        else if (l == 3)
            KWS_LAUNCH(prof_name("bn_bwd_apply_kernel", l + 1), bn_bwd_apply_kernel<true>, dim3(blocks_for(M * C, 256)), dim3(256), 0, s, w.z[l], w.gz[l], k,
                       params + m->o_g[l], M * C, C);
        else
            KWS_LAUNCH(prof_name("bn_bwd_apply_kernel", l + 1), bn_bwd_apply_kernel<false>, dim3(blocks_for(M * C, 256)), dim3(256), 0, s, w.z[l], w.gz[l], k,
                       params + m->o_g[l], M * C, C);
        const float *in = w.a[l - 1];
        float *dk = grads + m->o_k[l];
        const float *kern = params + m->o_k[l];
        if (hook && l == 3) KWS_TRY(hook->fire(4, s));
        if (l != 1)
            if (int rc = fork(l)) return rc;                   // dz of layer l is final: wgrad may start on the side stream
        if (l == 3) {
            const ConvGeom g = geom3x3(B, Hs[3], Ws[3], 1);
            if (mprec == 1 && cnn_a3_on_load(m, true, true))
                KWS_TRY((launch_wgrad_bf16<64, 128, 1, true, true>(w.z[2], nullptr, dk, g, s2, w.dzp, det, coef_of(w.coef[2], 64).scale)));
            else if (mprec == 1) KWS_TRY(launch_wgrad_bf16<64, 128, 1, true>(in, nullptr, dk, g, s2, w.dzp, det));
            else KWS_TRY(launch_wgrad<64, 128, 1>(in, w.gz[3], dk, g, s2, det));
            // grads[o_k[3] ..] (conv4, bn4, dense, head = 82 % of the bytes) are final once the side stream gets here: it
            // runs the dense and conv4 weight gradients in order and joined the caller's stream at fork(3), i.e. after
            // head_bwd, the dense bias sums and BN4's backward.  The early bucket's all-reduce goes right here, on the side stream
            // (the caller's stream does not wait for the conv4 wgrad, and the collective runs under the rest of the backward pass).
            if (comm) KWS_TRY(comm_allreduce_early(comm, grads + m->o_k[3], m->P - m->o_k[3], s2));
            if (bucket_event) KWS_HIP_CHECK(hipEventRecord(bucket_event, s2));
            if (group_bwd) {
                fused_bn3_blocks = launch_group_dgrad4(m, B, w, s, acc_bn3 ? acc_set(R, 1, 2, bpar[2]) : nullptr);
                if (fused_bn3_blocks < 0) return fused_bn3_blocks;
            } else if (mprec == 1 && blocks_for((long)B * Hs[3] * Ws[3], 64) <= (unsigned)kStatStride) {
                // the data gradient's epilogue is BatchNorm 3's backward reduction (conv3 has no pooling): it gates by ReLU6(y3), stores
                // g in gz[2] and leaves the partial sums of g and g xhat -- no bn_bwd_reduce pass over (z3, da3)
                const BnCoef k3 = coef_of(w.coef[2], 64);
                fused_bn3_blocks = launch_bf16<128, 64, MODE_DGRAD, EPI_BNBWD_GATE6>("conv_bf16_dgrad", nullptr, w.wsp[1], k3.scale, w.gz[2], g, s, w.partial,
                                                                                   w.z[2], w.dzp);
                if (fused_bn3_blocks < 0) return fused_bn3_blocks;
            } else if (mprec == 1)
                KWS_TRY_NB(launch_bf16<128, 64, MODE_DGRAD, EPI_NONE>("conv_bf16_dgrad", nullptr, w.wsp[1], nullptr, w.da[2], g, s, nullptr, nullptr, w.dzp));
            else KWS_TRY(launch_dgrad<128, 64, 1>(w.gz[3], kern, w.da[2], g, s));
        } else if (l == 2) {
            const ConvGeom g = geom3x3(B, Hs[2], Ws[2], 2);
            if (mprec == 1) KWS_TRY(launch_wgrad_bf16<32, 64, 3>(in, w.gz[2], dk, g, s2, nullptr, det));
            else KWS_TRY(launch_wgrad<32, 64, 3>(in, w.gz[2], dk, g, s2, det));
            if (group_bwd) {
                if (acc_bn2) arm(1);                            // the last kernel in front of conv2's early weight-gradient fork
                KWS_TRY(launch_group_dgrad3(m, B, w, s, acc_bn2 ? acc_set(R, 1, 1, bpar[1]) : nullptr));
            }
            else KWS_TRY(launch_dgrad<64, 32, 2>(w.gz[2], kern, w.da[1], g, s));
        } else {
            // conv2 (16 -> 32, 3x3, stride 1): clip-resident kernels, the clip's tiles are staged in LDS once
            const int H1 = Hs[1], W1 = Ws[1];
            // persistent grids: at most the LDS-limited residency (3 resp. 5 blocks per CU), with an equal share of clips each
            auto even_grid = [&](int max_blocks) { const int cpb = (B + max_blocks - 1) / max_blocks; return (unsigned)((B + cpb - 1) / cpb); };
            const unsigned nblk = even_grid(cu_count() * 3), nblk_d = even_grid(cu_count() * 5);
            const size_t smw = sizeof(float) * ((size_t)(H1 + 2) * (W1 + 2) * 16 + (size_t)((H1 * W1 + 3) / 4) * 4 * stride16(32));
            const size_t smw2 = std::max(smw, sizeof(float) * (size_t)(1024 + 16 * 32));
            // dgrad forms dz2 from (g, z2) while staging and leaves it in gz[1]; wgrad then overlaps with layer 1's kernels
            const size_t smd = sizeof(float) * (size_t)(H1 + 2) * (W1 + 2) * (32 + 4);
            BnBwdArgs bn = {w.z[1], params + m->o_g[1], k.mean, k.inv, k.k2, k.k3};
            if (compact_g) { bn.gw = w.da[1]; bn.arg = routed_bwd2 ? w.arg2 : reinterpret_cast<const unsigned char *>(w.da[2]); }
            BnBwdArgs bn_w = bn;                                // the weight gradient's copy: it only reads the accumulator set
            if (acc_bn2) {
                bn.acc = bn_w.acc = acc_set(R, 1, 1, bpar[1]);
                bn.M = bn_w.M = M;
                bn.acc_clear_set = acc_set(R, 1, 1, bpar[1] + 1);
                bn.dgamma = grads + m->o_g[1]; bn.dbeta = grads + m->o_b[1]; bn.k2w = k.k2; bn.k3w = k.k3;
            }
            const bool wgrad_bf16 = mprec == 1 && H1 * W1 <= 160;
            // compact g and a clip that fits the kernels' register staging: the weight gradient forms dz itself (from the routed g
            // and z2), so it forks BEFORE the data gradient and runs beside it and beside layer 1 on the side stream; the data
            // gradient then does not write dz back
            const bool wgrad_early = compact_g && wgrad_bf16 && H1 * W1 * 8 <= 1280 && H1 * W1 * 4 <= 768;
            const size_t smwb = std::max((size_t)3 * 32 * (H1 + 2) * (W1 + 2) + (size_t)6 * 32 * (H1 * W1 + 1), sizeof(float) * 9 * 16 * 32);
            // two blocks per CU although three fit: the third takes the LDS the layer-1 kernels of the main chain need beside it
            // (same-box A/B: 0.785 ms/step with one or two, 0.800 with three); deterministic: one persistent block walks every clip
            auto wgrad_grid = [&](int occ) { return dim3(det ? 1u : even_grid(cu_count() * std::min(occ, 2))); };
            if (wgrad_early) {
                if (int rc = fork(1)) return rc;
                static const int occ = resident_blocks(conv_wgrad_clip_bf16_kernel<true>, 256, smwb);
                if (!no_arm) arm_stop_event(R->ev[9], s2);       // the last kernel of the side stream: its completion is the join event
                KWS_LAUNCH("conv_wgrad_clip_bf16<16,32>", conv_wgrad_clip_bf16_kernel<true>, wgrad_grid(occ), dim3(256), smwb, s2, in, nullptr, dk, B, H1, W1, bn_w);
            }
            if (mprec == 1) {
                // split-precision form: 2 blocks per CU by registers (the weight fragments of all nine taps stay in them)
                const size_t smdb = (size_t)12 * 16 * (((H1 + 2) * (W1 + 2) + 15) & ~15);
                if (wgrad_early)
                    KWS_LAUNCH("conv_dgrad_clip_bf16<32,16>", (conv_dgrad_clip_bf16_kernel<true, true, false>), dim3(even_grid(cu_count() * 2)), dim3(256), smdb, s,
                               w.gz[1], kern, w.da[0], B, H1, W1, bn);
                else if (compact_g)
                    KWS_LAUNCH("conv_dgrad_clip_bf16<32,16>", (conv_dgrad_clip_bf16_kernel<true, true>), dim3(even_grid(cu_count() * 2)), dim3(256), smdb, s,
                               w.gz[1], kern, w.da[0], B, H1, W1, bn);
                else
                    KWS_LAUNCH("conv_dgrad_clip_bf16<32,16>", (conv_dgrad_clip_bf16_kernel<true, false>), dim3(even_grid(cu_count() * 2)), dim3(256), smdb, s,
                               w.gz[1], kern, w.da[0], B, H1, W1, bn);
            } else
                KWS_LAUNCH("conv_dgrad_clip<32,16>", (conv_dgrad_clip_kernel<32, true>), dim3(nblk_d), dim3(256), smd, s, w.gz[1], kern, w.da[0], B, H1, W1, bn);
            if (wgrad_early) {
                ;
            } else if (wgrad_bf16) {
                if (int rc = fork(1)) return rc;
                static const int occ = resident_blocks(conv_wgrad_clip_bf16_kernel<false>, 256, smwb);
                KWS_LAUNCH("conv_wgrad_clip_bf16<16,32>", conv_wgrad_clip_bf16_kernel<false>, wgrad_grid(occ), dim3(256), smwb, s2, in, w.gz[1], dk, B, H1, W1, bn);
            } else {
                if (int rc = fork(1)) return rc;
                KWS_LAUNCH("conv_wgrad_clip<16,32>", conv_wgrad_clip_kernel<32>, dim3(det ? 1u : nblk), dim3(256), smw2, s2, in, w.gz[1], dk, B, H1, W1);
            }
        }
    }
    // layer 1: da1 -> (dgamma1, dbeta1, dW1) with conv1 recomputed; no z1-sized tensor is read or written
    {
        const int cpb = std::max(1, (B + kMaxStatBlocks - 1) / kMaxStatBlocks), nb = (B + cpb - 1) / cpb;
        const size_t smem1 = sizeof(float) * (size_t)(d.H0 + 2) * (d.W0 + 2);
        // every pixel in a pool window, a haloed map of at most 768 floats and at most 40 tiles: the MFMA, wave-per-clip forms
        const bool l1m = d.H0 % 2 == 0 && d.W0 % 2 == 0 && (d.H0 + 2) * (d.W0 + 2) <= 64 * kL1Stage &&
                         (d.H0 / 2) * (d.W0 / 2) <= 4 * kL1MaxTiles;
        const int cpw = std::max(1, (B + 4 * kMaxStatBlocks - 1) / (4 * kMaxStatBlocks)), nbm = (B + 4 * cpw - 1) / (4 * cpw);
        // per-wave LDS tiles; the compile-time form of the default map reads two rows past the haloed map (kws_layer1.h: L1Runs)
        const size_t smemm = sizeof(float) * 4 * (d.H0 == 30 && d.W0 == 20 ? (size_t)L1Runs<30, 20>::TILE : (size_t)((((d.H0 + 2) * (d.W0 + 2)) + 3) & ~3));
        const long M1 = (long)B * d.H0 * d.W0;
        BnCoef k1 = coef_of(w.coef[0], 16);
        const float *kern1 = params + m->o_k[0];
        if (l1m) {
            // one pass collects G, sum g, sum g z per block (double partials, fixed order); the closed forms of dW1, dgamma, dbeta use Q
            const double *q = moments ? moments : w.moments;
            // the default map (30 frames x 20 coefficients) has a fully unrolled form with its own window walk (kws_layer1_fast.h)
            // non-deterministic mode, default map: the kernel's last block evaluates the closed forms itself (no finalize launch); layer 0's
            // backward set of the model's accumulators holds the sums, its last word the ticket counter (kAccSlots * kL1BwdRows < kAccDoubles)
            const bool fin_in_kernel = acc_ok && d.H0 == 30 && d.W0 == 20;
            if (fin_in_kernel) {
                double *acc0 = acc_set(R, 1, 0, 0);
                const L1FinalizeArgs fin{acc0, reinterpret_cast<unsigned *>(acc0 + kAccDoubles - 1), q, params + m->o_g[0], grads + m->o_k[0],
                                         grads + m->o_g[0], grads + m->o_b[0]};
                KWS_LAUNCH("l1m_bwd_onepass_kernel", (l1f_bwd_onepass_kernel<30, 20>), dim3(nbm), dim3(256), smemm, s, feat, kern1, w.da[0], k1, B, cpw, w.partial, fin);
            } else if (d.H0 == 30 && d.W0 == 20)
                KWS_LAUNCH("l1m_bwd_onepass_kernel", (l1f_bwd_onepass_kernel<30, 20>), dim3(nbm), dim3(256), smemm, s, feat, kern1, w.da[0], k1, B, cpw, w.partial);
            else
                KWS_LAUNCH("l1m_bwd_onepass_kernel", l1m_bwd_onepass_kernel, dim3(nbm), dim3(256), smemm, s, feat, kern1, w.da[0], k1, B, d.H0, d.W0, cpw,
                           w.partial);
            if (!fin_in_kernel)
                KWS_LAUNCH("l1_bwd_finalize_moments_kernel", l1_bwd_finalize_moments_kernel, dim3(144 + 16), dim3(64), 0, s, w.partial, nbm, q, kern1,
                           params + m->o_g[0], k1, grads + m->o_k[0], grads + m->o_g[0], grads + m->o_b[0]);
        } else {
            KWS_LAUNCH("l1_bwd_reduce_kernel", l1_bwd_reduce_kernel<16>, dim3(nb), dim3(256), smem1, s, feat, kern1, w.da[0], k1, B, d.H0, d.W0,
                       cpb, w.partial);
            KWS_LAUNCH(prof_name("bn_bwd_finalize_kernel", 1), bn_bwd_finalize_kernel, dim3(16), dim3(64), 0, s, w.partial, nb, M1, 16,
                       params + m->o_g[0], grads + m->o_g[0], grads + m->o_b[0], k1);
            KWS_LAUNCH("l1_bwd_wgrad_kernel", l1_bwd_wgrad_kernel<16>, dim3(det ? 1 : nb), dim3(256), smem1, s, feat, kern1, w.da[0], k1, params + m->o_g[0],
                       grads + m->o_k[0], B, d.H0, d.W0, det ? B : cpb);
        }
    }
    if (!stop_event_bound(R->ev[9])) KWS_HIP_CHECK(hipEventRecord(R->ev[9], s2));   // join: every wgrad is part of the caller's stream order again
    KWS_HIP_CHECK(hipStreamWaitEvent(s, R->ev[9], 0));
    KWS_LAUNCH_CHECK("simple_cnn backward");
    R->acc_dirty = false;
    return KWS_OK;
}


// ---- simple_cnn_lite (cnn.py:77-141) --------------------------------------------------------------------------------
ConvGeom geom1x1(int B, int H, int W)
{
    ConvGeom g;
    g.B = B; g.H = H; g.W = W; g.Ho = H; g.Wo = W; g.stride = 1; g.pt = 0; g.pl = 0; g.KH = 1; g.KW = 1;
    return g;
}

// Storage / matrix-operand precision of simple_cnn_lite INFERENCE (kws_set_inference_precision): 0 = fp32, 1 = fp16

// fp16 inference (kws_lite_f16.h): front kernel with fp16 output, then ONE kernel from a2 to the probabilities.
// The fp16 weight blob lives at the head of the (otherwise unused in inference) double partial slab.
static int lite_forward_f16(const kws_model *m, const float *feat, int B, const float *params, const float *state, CnnWs &w, float *probs,
                            int32_t *argmax, hipStream_t s)
{
    const CnnDims &d = m->d;
    const bool front_ok = d.H0 % 2 == 0 && d.W0 % 2 == 0 && (d.H0 + 2) * (d.W0 + 2) <= 64 * 12 && d.H2 >= 1 && d.W2 >= 1;
    if (!front_ok || d.H2 * d.W2 > kF16MaxN2 || d.H3 * d.W3 > kF16MaxP3 || d.H4 < 1 || d.W4 < 1 || m->C > kF16HeadCols || d.flat > kWdRow)
        return fail(KWS_ERR_UNSUPPORTED, "fp16 inference of simple_cnn_lite needs a feature map up to about 30 x 20 and at most %d classes "
                                         "(got %d x %d, %d classes): use KWS_INFER_FP32", kF16HeadCols, d.H0, d.W0, m->C);
    const bool prepared = m->prepared_for(params, state, w.base, B, matrix_prec(m), infer_prec(m));
    _Float16 *blob = reinterpret_cast<_Float16 *>(w.partial);
    if (!prepared) {
        if (int rc = infer_coefs(m, params, state, w, s)) return rc;
        KWS_LAUNCH("lite_f16_prepare_kernel", lite_f16_prepare_kernel, dim3(32), dim3(256), 0, s, params + m->o_pwk[2], params + m->o_pwk[3],
                   params + m->o_dk, params + m->o_hk, m->C, d.flat, blob);
    }
    BnCoef k0 = coef_of(w.coef[0], 16), k1 = coef_of(w.coef[1], 32), k2 = coef_of(w.coef[2], 64), k3 = coef_of(w.coef[3], 128);
    const LiteFrontArgs fa = {params + m->o_dwk[0], params + m->o_pwk[0], params + m->o_pwb[0], k0.scale, k0.shift,
                              params + m->o_dwk[1], params + m->o_pwk[1], params + m->o_pwb[1], k1.scale, k1.shift};
    const size_t sm = sizeof(float) * (size_t)lite_front_floats(d.H0, d.W0);
    const int waves = cu_count() * std::max(1, std::min(8, (int)(160 * 1024 / sm)));
    const int cpw = std::max(1, (B + waves - 1) / waves), nblk = (B + cpw - 1) / cpw;
    _Float16 *a2h = reinterpret_cast<_Float16 *>(w.a[1]);
    KWS_LAUNCH("lite_front_infer_kernel", lite_front_infer_kernel, dim3(nblk), dim3(64), sm, s, feat, fa, w.a[1], B, d.H0, d.W0, cpw, a2h);
    LiteF16Args a;
    a.dwk3 = params + m->o_dwk[2]; a.pwb3 = params + m->o_pwb[2]; a.sc3 = k2.scale; a.sh3 = k2.shift;
    a.dwk4 = params + m->o_dwk[3]; a.pwb4 = params + m->o_pwb[3]; a.sc4 = k3.scale; a.sh4 = k3.shift;
    a.db = params + m->o_db; a.hb = params + m->o_hb; a.blob = blob;
    a.C = m->C; a.H2 = d.H2; a.W2 = d.W2; a.H3 = d.H3; a.W3 = d.W3;
    a.pt3 = same_pad_before(d.H2, 3, 2); a.pl3 = same_pad_before(d.W2, 3, 2); a.H4 = d.H4; a.W4 = d.W4;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(lite_back_f16_kernel), kF16LdsBytes)) return rc;
    const int ntile = (B + kF16Clips - 1) / kF16Clips;
    KWS_LAUNCH("lite_back_f16_kernel", lite_back_f16_kernel, dim3(std::min(ntile, cu_count())), dim3(256), (size_t)kF16LdsBytes, s, a2h, a, B, probs, argmax);
    KWS_LAUNCH_CHECK("simple_cnn_lite fp16 forward");
    return KWS_OK;
}

int lite_forward(const kws_model *m, const float *feat, int B, const float *params, float *state, CnnWs &w, bool training,
                 uint64_t seed, hipStream_t s)
{
    const CnnDims &d = m->d;
    const int Hs[4] = {d.H0, d.H1, d.H2, d.H3}, Ws[4] = {d.W0, d.W1, d.W2, d.W3};
    const int Hz[4] = {d.H0, d.H1, d.H3, d.H3}, Wz[4] = {d.W0, d.W1, d.W3, d.W3};
    const int strd[4] = {1, 1, 2, 1};
    const bool pool[4] = {true, true, false, true};
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);
    int l_begin = 0;
    if (!training && !m->prepared_for(params, state, w.base, B, matrix_prec(m), infer_prec(m)))
        if (int rc = infer_coefs(m, params, state, w, s)) return rc;
    if (!training && d.H0 % 2 == 0 && d.W0 % 2 == 0 && (d.H0 + 2) * (d.W0 + 2) <= 64 * 12 && d.H2 >= 1 && d.W2 >= 1) {
        // inference: stages 1 and 2 fused, one wave per clip, features -> a2 without touching HBM in between (kws_lite.h)
        BnCoef k0 = coef_of(w.coef[0], 16), k1 = coef_of(w.coef[1], 32);
        const LiteFrontArgs fa = {params + m->o_dwk[0], params + m->o_pwk[0], params + m->o_pwb[0], k0.scale, k0.shift,
                                  params + m->o_dwk[1], params + m->o_pwk[1], params + m->o_pwb[1], k1.scale, k1.shift};
        const size_t sm = sizeof(float) * (size_t)lite_front_floats(d.H0, d.W0);
        const int waves = cu_count() * std::max(1, std::min(8, (int)(160 * 1024 / sm)));
        const int cpw = std::max(1, (B + waves - 1) / waves), nblk = (B + cpw - 1) / cpw;
        KWS_LAUNCH("lite_front_infer_kernel", lite_front_infer_kernel, dim3(nblk), dim3(64), sm, s, feat, fa, w.a[1], B, d.H0, d.W0, cpw);
        l_begin = 2;
    }
    for (int l = l_begin; l < 4; ++l) {
        const float *in = l == 0 ? feat : w.a[l - 1];
        const int Cin = kCh[l], C = kCh[l + 1];
        const long M = (long)B * Hz[l] * Wz[l];
        const ConvGeom g = geom3x3(B, Hs[l], Ws[l], strd[l]);
        KWS_LAUNCH(prof_name("dwconv_fwd_kernel", l + 1), dwconv_fwd_kernel, dim3(blocks_for(M * Cin, 256)), dim3(256), 0, s, in,
                   params + m->o_dwk[l], w.dwo[l], g, Cin);
        const float *pwk = params + m->o_pwk[l], *pwb = params + m->o_pwb[l];
        const ConvGeom g1 = geom1x1(B, Hz[l], Wz[l]);
        if (l == 0) KWS_LAUNCH("pw1_fwd_kernel", pw1_fwd_kernel<16>, dim3(blocks_for(M * 16, 256)), dim3(256), 0, s, w.dwo[0], pwk, pwb, w.z[0], M);
        else if (l == 1) KWS_TRY(launch_gemm<16, 32, MODE_FWD, EPI_BIAS>(w.dwo[1], pwk, pwb, w.z[1], g1, s));
        else if (l == 2) KWS_TRY(launch_gemm<32, 64, MODE_FWD, EPI_BIAS_RELU>(w.dwo[2], pwk, pwb, w.z[2], g1, s));      // activation='relu', cnn.py:113
        else KWS_TRY(launch_gemm<64, 128, MODE_FWD, EPI_BIAS_RELU>(w.dwo[3], pwk, pwb, w.z[3], g1, s));                  // cnn.py:122
        BnCoef k = coef_of(w.coef[l], C);
        if (training) {
            int nblk, rows;
            stat_grid(M, C, nblk, rows);
            KWS_LAUNCH(prof_name("channel_stats_kernel", l + 1), channel_stats_kernel, dim3(nblk), dim3(256), 0, s, w.z[l], M, C, rows, w.partial);
            KWS_LAUNCH(prof_name("bn_finalize_train_kernel", l + 1), bn_finalize_train_kernel, dim3(C), dim3(64), 0, s, w.partial, nblk, M, C,
                       params + m->o_g[l], params + m->o_b[l], state + m->o_mm[l], state + m->o_mv[l], k);
        }
        const float rate = (training && l == 3 && seed != 0) ? 0.5f : 0.f;
        if (pool[l])
            KWS_LAUNCH(prof_name("bn_act_pool_kernel", l + 1), bn_act_pool_kernel<true>, dim3(blocks_for((long)B * (Hz[l] / 2) * (Wz[l] / 2) * C, 256)),
                       dim3(256), 0, s, w.z[l], k.scale, k.shift, w.a[l], B, Hz[l], Wz[l], C, rate, slo, shi);
        else
            KWS_LAUNCH(prof_name("bn_act_pool_kernel", l + 1), bn_act_pool_kernel<false>, dim3(blocks_for(M * C, 256)), dim3(256), 0, s, w.z[l],
                       k.scale, k.shift, w.a[l], B, Hz[l], Wz[l], C, rate, slo, shi);
    }
    ConvGeom g;
    g.B = B; g.H = d.H4; g.W = d.W4; g.Ho = 1; g.Wo = 1; g.stride = 1; g.pt = 0; g.pl = 0; g.KH = d.H4; g.KW = d.W4;
    KWS_TRY(launch_gemm<128, 128, MODE_FWD, EPI_BIAS_RELU6>(w.a[3], params + m->o_dk, params + m->o_db, w.d1, g, s));
    KWS_LAUNCH_CHECK("simple_cnn_lite forward");
    return KWS_OK;
}

int lite_backward(const kws_model *m, const float *feat, int B, const float *params, float *grads, CnnWs &w, uint64_t seed,
                  hipEvent_t bucket_event, hipStream_t s)
{
    const CnnDims &d = m->d;
    const int Hs[4] = {d.H0, d.H1, d.H2, d.H3}, Ws[4] = {d.W0, d.W1, d.W2, d.W3};
    const int Hz[4] = {d.H0, d.H1, d.H3, d.H3}, Wz[4] = {d.W0, d.W1, d.W3, d.W3};
    const int strd[4] = {1, 1, 2, 1};
    const bool pool[4] = {true, true, false, true};
    const uint32_t slo = (uint32_t)(seed & 0xFFFFFFFFu), shi = (uint32_t)(seed >> 32);
    const bool det = m->deterministic != 0;
    KWS_HIP_CHECK(hipMemsetAsync(grads, 0, sizeof(float) * (size_t)m->P, s));
    KWS_TRY(run_head_bwd(m, B, params, w.d1, w.dlogits, w.dd1, grads, true, s, nullptr, nullptr, nullptr, nullptr, det));
    {
        ConvGeom g;
        g.B = B; g.H = d.H4; g.W = d.W4; g.Ho = 1; g.Wo = 1; g.stride = 1; g.pt = 0; g.pl = 0; g.KH = d.H4; g.KW = d.W4;
        int nblk, rows;
        stat_grid(B, 128, nblk, rows);
        KWS_LAUNCH("channel_stats_kernel.dense", channel_stats_kernel, dim3(nblk), dim3(256), 0, s, w.dd1, (long)B, 128, rows, w.partial);
        KWS_LAUNCH("colsum_finalize_kernel", colsum_finalize_kernel, dim3(128), dim3(256), 0, s, w.partial, nblk, 128, grads + m->o_db);
        KWS_TRY(launch_wgrad<128, 128, 1>(w.a[3], w.dd1, grads + m->o_dk, g, s, det));
        KWS_TRY(launch_dgrad<128, 128, 1>(w.dd1, params + m->o_dk, w.da4, g, s));
    }
    for (int l = 3; l >= 0; --l) {
        const int Cin = kCh[l], C = kCh[l + 1];
        const long M = (long)B * Hz[l] * Wz[l];
        const float *da = l == 3 ? w.da4 : w.da[l];
        BnCoef k = coef_of(w.coef[l], C);
        int nblk, rows;
        stat_grid(M, C, nblk, rows);
        const float rate = (l == 3 && seed != 0) ? 0.5f : 0.f;
        if (pool[l])
            KWS_LAUNCH(prof_name("bn_bwd_reduce_kernel", l + 1), bn_bwd_reduce_kernel<true>, dim3(nblk), dim3(256), 0, s, w.z[l], da, k, w.gz[l], B,
                       Hz[l], Wz[l], C, rows, w.partial, rate, slo, shi);
        else
            KWS_LAUNCH(prof_name("bn_bwd_reduce_kernel", l + 1), bn_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, s, w.z[l], da, k, w.gz[l], B,
                       Hz[l], Wz[l], C, rows, w.partial, rate, slo, shi);
        KWS_LAUNCH(prof_name("bn_bwd_finalize_kernel", l + 1), bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, w.partial, nblk, M, C,
                   params + m->o_g[l], grads + m->o_g[l], grads + m->o_b[l], k);
        if (l >= 2)      // sepconv3 and sepconv4 carry activation='relu' (cnn.py:113,122)
            KWS_LAUNCH(prof_name("bn_bwd_apply_kernel", l + 1), bn_bwd_apply_kernel<true>, dim3(blocks_for(M * C, 256)), dim3(256), 0, s, w.z[l], w.gz[l],
                       k, params + m->o_g[l], M * C, C);
        else
            KWS_LAUNCH(prof_name("bn_bwd_apply_kernel", l + 1), bn_bwd_apply_kernel<false>, dim3(blocks_for(M * C, 256)), dim3(256), 0, s, w.z[l], w.gz[l],
                       k, params + m->o_g[l], M * C, C);
        // pointwise 1x1 + bias
        const float *pwk = params + m->o_pwk[l];
        const ConvGeom g1 = geom1x1(B, Hz[l], Wz[l]);
        if (l == 0) {
            KWS_LAUNCH("pw1_bwd_kernel", pw1_bwd_kernel<16>, dim3(nblk), dim3(256), 0, s, w.dwo[0], pwk, w.gz[0], w.ddw[0], M, rows, w.partial);
            KWS_LAUNCH("partial_finalize_kernel", partial_finalize_kernel, dim3(16), dim3(256), 0, s, w.partial, nblk, 16, 0, grads + m->o_pwk[0]);
            KWS_LAUNCH("partial_finalize_kernel", partial_finalize_kernel, dim3(16), dim3(256), 0, s, w.partial, nblk, 16, 1, grads + m->o_pwb[0]);
        } else {
            KWS_LAUNCH(prof_name("channel_stats_kernel.bias", l + 1), channel_stats_kernel, dim3(nblk), dim3(256), 0, s, w.gz[l], M, C, rows, w.partial);
            KWS_LAUNCH("colsum_finalize_kernel", colsum_finalize_kernel, dim3(C), dim3(256), 0, s, w.partial, nblk, C, grads + m->o_pwb[l]);
            if (l == 3) {
                KWS_TRY(launch_wgrad<64, 128, 1>(w.dwo[3], w.gz[3], grads + m->o_pwk[3], g1, s, det));
                KWS_TRY(launch_dgrad<128, 64, 1>(w.gz[3], pwk, w.ddw[3], g1, s));
            } else if (l == 2) {
                KWS_TRY(launch_wgrad<32, 64, 1>(w.dwo[2], w.gz[2], grads + m->o_pwk[2], g1, s, det));
                KWS_TRY(launch_dgrad<64, 32, 1>(w.gz[2], pwk, w.ddw[2], g1, s));
            } else {
                KWS_TRY(launch_wgrad<16, 32, 1>(w.dwo[1], w.gz[1], grads + m->o_pwk[1], g1, s, det));
                KWS_TRY(launch_dgrad<32, 16, 1>(w.gz[1], pwk, w.ddw[1], g1, s));
            }
        }
        // depthwise 3x3
        const float *in = l == 0 ? feat : w.a[l - 1];
        const ConvGeom g = geom3x3(B, Hs[l], Ws[l], strd[l]);
        int nb2, rows2;
        stat_grid(M, Cin, nb2, rows2);
        KWS_LAUNCH(prof_name("dwconv_wgrad_kernel", l + 1), dwconv_wgrad_kernel, dim3(nb2), dim3(256), 0, s, in, w.ddw[l], g, Cin, rows2, w.partial);
        KWS_LAUNCH("partial_finalize_kernel", partial_finalize_kernel, dim3(9 * Cin), dim3(256), 0, s, w.partial, nb2, 9 * Cin, 0, grads + m->o_dwk[l]);
        if (l > 0)
            KWS_LAUNCH(prof_name("dwconv_dgrad_kernel", l + 1), dwconv_dgrad_kernel, dim3(blocks_for((long)B * Hs[l] * Ws[l] * Cin, 256)), dim3(256), 0, s,
                       w.ddw[l], params + m->o_dwk[l], w.da[l - 1], g, Cin);
        if (l == 3 && bucket_event) KWS_HIP_CHECK(hipEventRecord(bucket_event, s));
    }
    KWS_LAUNCH_CHECK("simple_cnn_lite backward");
    return KWS_OK;
}

int check_ws(const kws_model *m, int B, bool training, void *ws, size_t ws_bytes, CnnWs &w)
{
    if (!ws) return fail(KWS_ERR_INVALID, "null workspace");
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(KWS_ERR_INVALID, "workspace must be 256-byte aligned");
    w = carve_cnn(m, B, training, static_cast<unsigned char *>(ws));
    if (w.bytes > ws_bytes) return fail(KWS_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return KWS_OK;
}

}  // namespace

namespace kws {

int run_head(const kws_model *m, int B, const float *params, const float *x, float *loss_i, float *correct_i,
             const int32_t *labels, const float *class_w, float *probs, int32_t *argmax, float *dlogits, float grad_scale,
             float *stats, int ignore_index, hipStream_t s)
{
    const int K = m->head_K;
    const size_t smem = sizeof(float) * (size_t)(16 * K + 16 * m->C);
    const size_t smem_fast = sizeof(float) * (size_t)(16 * (K + 1) + K * m->C + 16 * m->C);
    if (smem_fast <= 60 * 1024)      // W2 fits in LDS beside the tile (C <= ~100 at K = 128): the 16-lanes-per-sample form
        KWS_LAUNCH("head_fwd_kernel", head_fwd_fast_kernel, dim3(blocks_for(B, 16)), dim3(256), smem_fast, s, x, params + m->o_hk, params + m->o_hb,
                   labels, class_w, probs, argmax, loss_i, correct_i, dlogits, B, K, m->C, grad_scale, ignore_index);
    else
    KWS_LAUNCH("head_fwd_kernel", head_fwd_kernel, dim3(blocks_for(B, 16)), dim3(256), smem, s, x, params + m->o_hk, params + m->o_hb,
               labels, class_w, probs, argmax, loss_i, correct_i, dlogits, B, K, m->C, grad_scale, ignore_index);
    if (labels && stats) KWS_LAUNCH("loss_reduce_kernel", loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_i, correct_i, B, stats);
    KWS_LAUNCH_CHECK("head");
    return KWS_OK;
}

int run_loss_reduce(const float *loss_i, const float *correct_i, int B, float *stats, hipStream_t s)
{
    KWS_LAUNCH("loss_reduce_kernel", loss_reduce_kernel, dim3(1), dim3(256), 0, s, loss_i, correct_i, B, stats);
    KWS_LAUNCH_CHECK("loss sums");
    return KWS_OK;
}

bool head_bwd_fuses(const kws_model *m) { return m->head_K % 16 == 0 && m->head_K <= 128 && m->C <= 48; }

int run_head_bwd(const kws_model *m, int B, const float *params, const float *x, const float *dlogits, float *dx,
                 float *grads, bool relu6_gate, hipStream_t s, float *dx_colsum, const float *loss_i, const float *correct_i,
                 float *stats, bool deterministic, const HeadFwdArgs *fwd)
{
    const int K = m->head_K;
    // deterministic: the kernels below only produce dx (and the loss sums); dW2 / db2 come from a batch-ordered kernel
    float *dw2 = deterministic ? nullptr : grads + m->o_hk, *db2 = deterministic ? nullptr : grads + m->o_hb;
    if (deterministic) {
        dx_colsum = nullptr;
        KWS_LAUNCH("head_wgrad_det_kernel", head_wgrad_det_kernel, dim3(blocks_for((long)K * m->C + m->C, 256)), dim3(256), 0, s, x, dlogits,
                   grads + m->o_hk, grads + m->o_hb, B, K, m->C);
    }
    const size_t smem = sizeof(float) * (size_t)(kHeadBwdRows * K + kHeadBwdRows * m->C);
    if (head_bwd_fuses(m)) {    // the MFMA form: W2, a 16-sample tile and dlogits padded to 48 classes live in LDS
        constexpr int G = 1;                          // 16-sample groups per block (4 measured slower: 64 blocks expose each group's staging latency)
        const size_t smem_fast = sizeof(float) * (size_t)(16 * (K + 2) + (16 + K) * 50);
        if (fwd) {          // the train step's fused form: forward + loss + both backward products (kws_layers.h)
            if (deterministic) return fail(KWS_ERR_INVALID, "the fused head kernel serves the non-deterministic train steps");
            if (relu6_gate)
                KWS_LAUNCH("head_fwd_bwd_kernel", (head_bwd_mfma_kernel<true, 1, true>), dim3(blocks_for(B, 16)), dim3(256), smem_fast, s, x, params + m->o_hk,
                           nullptr, dx, dw2, db2, B, K, m->C, dx_colsum, nullptr, nullptr, nullptr, *fwd);
            else            // the recurrent models: no activation between the last hidden state and the head
                KWS_LAUNCH("head_fwd_bwd_kernel", (head_bwd_mfma_kernel<false, 1, true>), dim3(blocks_for(B, 16)), dim3(256), smem_fast, s, x, params + m->o_hk,
                           nullptr, dx, dw2, db2, B, K, m->C, dx_colsum, nullptr, nullptr, nullptr, *fwd);
            KWS_LAUNCH_CHECK("head forward + backward");
            return KWS_OK;
        }
        if (relu6_gate)
            KWS_LAUNCH("head_bwd_kernel", (head_bwd_mfma_kernel<true, G>), dim3(blocks_for(B, 16 * G)), dim3(256), smem_fast, s, x, params + m->o_hk, dlogits,
                       dx, dw2, db2, B, K, m->C, dx_colsum, loss_i, correct_i, stats);
        else
            KWS_LAUNCH("head_bwd_kernel", (head_bwd_mfma_kernel<false, G>), dim3(blocks_for(B, 16 * G)), dim3(256), smem_fast, s, x, params + m->o_hk, dlogits,
                       dx, dw2, db2, B, K, m->C, dx_colsum, loss_i, correct_i, stats);
        KWS_LAUNCH_CHECK("head backward");
        return KWS_OK;
    }
    if (relu6_gate)
        KWS_LAUNCH("head_bwd_kernel", head_bwd_kernel<true>, dim3(blocks_for(B, kHeadBwdRows)), dim3(256), smem, s, x, params + m->o_hk, dlogits,
                   dx, dw2, db2, B, K, m->C);
    else
        KWS_LAUNCH("head_bwd_kernel", head_bwd_kernel<false>, dim3(blocks_for(B, kHeadBwdRows)), dim3(256), smem, s, x, params + m->o_hk, dlogits,
                   dx, dw2, db2, B, K, m->C);
    KWS_LAUNCH_CHECK("head backward");
    return KWS_OK;
}

}  // namespace kws

kws::ModelRes *kws_model::dev_res()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(res_mu);
    kws::ModelRes &r = res[dev];
    if (!r.side) {
        // lowest priority: the side stream carries the weight-gradient kernels, which must not delay the small kernels of
        // the caller's (critical-path) stream when both have work queued
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
        if (hipStreamCreateWithPriority(&r.side, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); r.side = nullptr; return nullptr; }
        for (auto &e : r.ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        const size_t acc_bytes = sizeof(double) * 2 * 4 * 2 * kws::kAccDoubles;
        if (hipMalloc(reinterpret_cast<void **>(&r.acc), acc_bytes) != hipSuccess || hipMemsetAsync(r.acc, 0, acc_bytes, r.side) != hipSuccess ||
            hipStreamSynchronize(r.side) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return &r;
}

kws_model::~kws_model()
{
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto &kv : res) {
        if (have) (void)hipSetDevice(kv.first);
        for (auto &e : kv.second.ev)
            if (e) (void)hipEventDestroy(e);
        if (kv.second.side) (void)hipStreamDestroy(kv.second.side);
        if (kv.second.acc) (void)hipFree(kv.second.acc);
    }
    if (have && !res.empty()) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

namespace kws {
int default_matrix_precision() { return g_matrix_precision; }
int default_infer_precision() { return g_infer_precision; }
}  // namespace kws

extern "C" {

int64_t kws_feature_moments_workspace_bytes(int B)
{
    (void)B;
    return (int64_t)sizeof(double) * kMomCount * kStatStride;
}

int kws_feature_moments(const float *feat, int B, int n_features, int feature_size, double *moments, void *ws, size_t ws_bytes,
                        void *stream)
{
    if (!feat || !moments || !ws) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 1 || n_features < 1 || feature_size < 1) return fail(KWS_ERR_INVALID, "bad shape");
    const int H = n_features, W = feature_size;
    if (H % 2 || W % 2 || (H + 2) * (W + 2) > 64 * kL1Stage || (H / 2) * (W / 2) > 4 * kL1MaxTiles)
        return fail(KWS_ERR_UNSUPPORTED, "feature moments cover even maps up to (H+2)(W+2) <= %d (got %d x %d)", 64 * kL1Stage, H, W);
    if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(KWS_ERR_INVALID, "workspace must be 256-byte aligned");
    if (ws_bytes < (size_t)kws_feature_moments_workspace_bytes(B)) return fail(KWS_ERR_WORKSPACE, "workspace too small for the moment partial sums");
    const int cpw = std::max(1, (B + 4 * kMaxStatBlocks - 1) / (4 * kMaxStatBlocks)), nbm = (B + 4 * cpw - 1) / (4 * cpw);
    const size_t smemm = sizeof(float) * 4 * (size_t)((((H + 2) * (W + 2)) + 3) & ~3);
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(ws);
    KWS_LAUNCH("l1_moments_kernel", l1_moments_kernel, dim3(nbm), dim3(256), smemm, s, feat, B, H, W, cpw, partial);
    KWS_LAUNCH("l1_moments_finalize_kernel", l1_moments_finalize_kernel, dim3(kMomCount), dim3(64), 0, s, partial, nbm, moments);
    KWS_LAUNCH_CHECK("feature moments");
    return KWS_OK;
}

int kws_model_bind_device(kws_model *m)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    if (!m->dev_res()) return fail(KWS_ERR_HIP, "cannot create the model's side stream / events on this device");
    return KWS_OK;
}

int kws_model_prepare_inference(kws_model *m, int B, const float *params, const float *state, void *ws, size_t ws_bytes, void *stream)
{
    if (!m || !params || !state) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 1) return fail(KWS_ERR_INVALID, "batch must be >= 1");
    m->prep = kws_model::Prepared{};
    if (m->kind == KWS_SIMPLE_GRU || m->kind == KWS_SIMPLE_LSTM) return KWS_OK;      // nothing derived from the weights
    CnnWs w;
    if (int rc = check_ws(m, B, false, ws, ws_bytes, w)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    KWS_TRY(infer_coefs(m, params, state, w, s));
    if (m->kind == KWS_SIMPLE_CNN && matrix_prec(m) == 1) KWS_TRY(fused_tail_ok(m, true) ? split_weights_fused(m, params, w, s) : split_weights(m, params, w, s));
    if (m->kind == KWS_SIMPLE_CNN_LITE && infer_prec(m) == KWS_INFER_FP16)
        KWS_LAUNCH("lite_f16_prepare_kernel", lite_f16_prepare_kernel, dim3(32), dim3(256), 0, s, params + m->o_pwk[2], params + m->o_pwk[3],
                   params + m->o_dk, params + m->o_hk, m->C, m->d.flat, reinterpret_cast<_Float16 *>(w.partial));
    KWS_LAUNCH_CHECK("inference preparation");
    m->prep.params = params; m->prep.state = state; m->prep.ws = ws; m->prep.B = B;
    m->prep.matrix = matrix_prec(m); m->prep.infer = infer_prec(m);
    return KWS_OK;
}

int kws_model_invalidate_prepared(kws_model *m)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    m->prep = kws_model::Prepared{};
    return KWS_OK;
}

int kws_model_set_precision(kws_model *m, int matrix, int infer)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    if (matrix != -1 && matrix != KWS_MATRIX_FP32 && matrix != KWS_MATRIX_BF16X6) return fail(KWS_ERR_INVALID, "unknown matrix precision %d", matrix);
    if (infer != -1 && infer != KWS_INFER_FP32 && infer != KWS_INFER_FP16) return fail(KWS_ERR_INVALID, "unknown inference precision %d", infer);
    m->matrix_precision = matrix;
    m->infer_precision = infer;
    // the prepared state is keyed by the precisions it was derived for (prepared_for): a forward at another precision does not match it,
    // and switching back finds it again -- unless a different forward has meanwhile used the same workspace (kws_model_forward drops it)
    return KWS_OK;
}

int kws_model_get_precision(const kws_model *m, int *matrix, int *infer)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    if (matrix) *matrix = matrix_prec(m);
    if (infer) *infer = infer_prec(m);
    return KWS_OK;
}

int kws_model_set_overlap_point(kws_model *m, int point)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    if (point < -1 || point > 10) return fail(KWS_ERR_INVALID, "overlap point %d outside -1 .. 10", point);
    m->overlap_point = point;
    return KWS_OK;
}

int kws_model_set_deterministic(kws_model *m, int on)
{
    if (!m) return fail(KWS_ERR_INVALID, "null argument");
    if (on && (m->kind == KWS_SIMPLE_GRU || m->kind == KWS_SIMPLE_LSTM))
        return fail(KWS_ERR_UNSUPPORTED, "the deterministic weight-gradient mode covers simple_cnn and simple_cnn_lite");
    m->deterministic = on ? 1 : 0;
    return KWS_OK;
}

int kws_model_create(int kind, int num_classes, int n_features, int feature_size, kws_model **out)
{
    if (!out) return fail(KWS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (kind < KWS_SIMPLE_CNN || kind > KWS_SIMPLE_LSTM) return fail(KWS_ERR_INVALID, "Unsupported model type");   // model.py:32
    if (num_classes < 2 || num_classes > 1024) return fail(KWS_ERR_INVALID, "num_classes must be in 2..1024");
    if (n_features < 1 || feature_size < 1) return fail(KWS_ERR_INVALID, "bad input geometry");
    auto *m = new kws_model();
    m->kind = kind; m->C = num_classes; m->n_features = n_features; m->feature_size = feature_size;
    if (kind == KWS_SIMPLE_GRU || kind == KWS_SIMPLE_LSTM) {
        if (feature_size > 64) {
            delete m;
            return fail(KWS_ERR_UNSUPPORTED, "the recurrent models support feature_size <= 64");
        }
        if (kind == KWS_SIMPLE_GRU) {
            // GRU(48, activation='linear', dropout=0.2) -> Dense(C, softmax)   (rnn.py:34-35, model.py:37)
            m->o_rk = m->add("gru_unit_0/kernel", {feature_size, 144}, true);
            m->o_ru = m->add("gru_unit_0/recurrent_kernel", {48, 144}, true);
            m->o_rb = m->add("gru_unit_0/bias", {2, 144}, true);
        } else {
            // LSTM(48, activation='tanh', dropout=0.2) -> Dense(C, softmax)    (rnn.py:70-71, model.py:37)
            m->o_rk = m->add("lstm_unit_0/kernel", {feature_size, 192}, true);
            m->o_ru = m->add("lstm_unit_0/recurrent_kernel", {48, 192}, true);
            m->o_rb = m->add("lstm_unit_0/bias", {192}, true);
        }
        m->head_K = 48;
        m->o_hk = m->add("score_predict/kernel", {48, num_classes}, true);
        m->o_hb = m->add("score_predict/bias", {num_classes}, true);
        *out = m;
        return KWS_OK;
    }
    CnnDims &d = m->d;
    d.H0 = n_features; d.W0 = feature_size;
    d.H1 = d.H0 / 2; d.W1 = d.W0 / 2;                 // MaxPooling2D(): 2x2, stride 2, 'valid'
    d.H2 = d.H1 / 2; d.W2 = d.W1 / 2;
    d.H3 = same_out(d.H2, 2); d.W3 = same_out(d.W2, 2);
    d.H4 = d.H3 / 2; d.W4 = d.W3 / 2;
    d.flat = d.H4 * d.W4 * 128;
    if (d.H4 < 1 || d.W4 < 1) {
        delete m;
        return fail(KWS_ERR_INVALID, "input %dx%d is too small for simple_cnn", n_features, feature_size);
    }
    const char *cn[4] = {"conv2d", "conv2d_1", "conv2d_2", "conv2d_3"};
    const char *sn[4] = {"separable_conv2d", "separable_conv2d_1", "separable_conv2d_2", "separable_conv2d_3"};
    const char *bn[4] = {"batch_normalization", "batch_normalization_1", "batch_normalization_2", "batch_normalization_3"};
    for (int l = 0; l < 4; ++l) {
        if (kind == KWS_SIMPLE_CNN_LITE) {
            m->o_dwk[l] = m->add(std::string(sn[l]) + "/depthwise_kernel", {3, 3, kCh[l], 1}, true);
            m->o_pwk[l] = m->add(std::string(sn[l]) + "/pointwise_kernel", {1, 1, kCh[l], kCh[l + 1]}, true);
            m->o_pwb[l] = m->add(std::string(sn[l]) + "/bias", {kCh[l + 1]}, true);
            m->o_k[l] = m->o_dwk[l];
        } else {
            m->o_k[l] = m->add(std::string(cn[l]) + "/kernel", {3, 3, kCh[l], kCh[l + 1]}, true);
        }
        m->o_g[l] = m->add(std::string(bn[l]) + "/gamma", {kCh[l + 1]}, true);
        m->o_b[l] = m->add(std::string(bn[l]) + "/beta", {kCh[l + 1]}, true);
        m->o_mm[l] = m->add(std::string(bn[l]) + "/moving_mean", {kCh[l + 1]}, false);
        m->o_mv[l] = m->add(std::string(bn[l]) + "/moving_variance", {kCh[l + 1]}, false);
    }
    m->o_dk = m->add("dense/kernel", {d.flat, 128}, true);
    m->o_db = m->add("dense/bias", {128}, true);
    m->o_hk = m->add("score_predict/kernel", {128, num_classes}, true);
    m->o_hb = m->add("score_predict/bias", {num_classes}, true);
    *out = m;
    return KWS_OK;
}

void kws_model_destroy(kws_model *m) { delete m; }

int64_t kws_model_param_count(const kws_model *m) { return m ? m->P : 0; }
int64_t kws_model_state_count(const kws_model *m) { return m ? m->S : 0; }
int kws_model_num_tensors(const kws_model *m) { return m ? (int)m->tensors.size() : 0; }

int kws_model_tensor_info(const kws_model *m, int index, kws_tensor_info *out)
{
    if (!m || !out || index < 0 || index >= (int)m->tensors.size()) return fail(KWS_ERR_INVALID, "bad tensor index");
    const Tensor &t = m->tensors[index];
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->name, t.name.c_str(), sizeof(out->name) - 1);
    out->ndim = (int32_t)t.shape.size();
    for (size_t i = 0; i < t.shape.size() && i < 4; ++i) out->shape[i] = t.shape[i];
    out->trainable = t.trainable ? 1 : 0;
    out->offset = t.offset;
    out->size = t.size;
    return KWS_OK;
}

int64_t kws_model_workspace_bytes(const kws_model *m, int B, int training)
{
    if (!m || B < 1) return 0;
    if (m->kind == KWS_SIMPLE_GRU || m->kind == KWS_SIMPLE_LSTM) return (int64_t)gru_workspace_bytes(m, B, training != 0);
    return (int64_t)carve_cnn(m, B, training != 0, nullptr).bytes;
}

int kws_model_forward(kws_model *m, const float *feat, int B, const float *params, const float *state, void *ws,
                      size_t ws_bytes, float *probs, int32_t *argmax, void *stream)
{
    if (!m || !feat || !params || !state) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 1) return fail(KWS_ERR_INVALID, "batch must be >= 1");
    if (m->kind == KWS_SIMPLE_GRU || m->kind == KWS_SIMPLE_LSTM)
        return gru_forward(m, feat, B, params, ws, ws_bytes, probs, argmax, static_cast<hipStream_t>(stream));
    CnnWs w;
    int rc = check_ws(m, B, false, ws, ws_bytes, w);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a forward that is not the prepared one re-derives the tables in this workspace: whatever was prepared in it is gone
    if (m->prep.ws == ws && !m->prepared_for(params, state, ws, B, matrix_prec(m), infer_prec(m))) m->prep = kws_model::Prepared{};
    if (m->kind == KWS_SIMPLE_CNN_LITE && infer_prec(m) == KWS_INFER_FP16)
        return lite_forward_f16(m, feat, B, params, state, w, probs, argmax, s);
    bool head_done = false;
    rc = m->kind == KWS_SIMPLE_CNN_LITE ? lite_forward(m, feat, B, params, const_cast<float *>(state), w, false, 0, s)
                                        : cnn_forward(m, feat, B, params, const_cast<float *>(state), w, false, 0, s, nullptr, nullptr, nullptr, nullptr,
                                                      probs, argmax, &head_done);
    if (rc || head_done) return rc;
    return run_head(m, B, params, w.d1, w.loss_i, w.correct_i, nullptr, nullptr, probs, argmax, nullptr, 0.f, nullptr, 0, s);
}

int kws_model_train_fwd_bwd(kws_model *m, const kws_train_args *a, void *stream)
{
    if (!m || !a || !a->feat || !a->labels || !a->params || !a->state || !a->grads) return fail(KWS_ERR_INVALID, "null argument");
    if (a->B < 1) return fail(KWS_ERR_INVALID, "batch must be >= 1");
    if (m->kind == KWS_SIMPLE_GRU || m->kind == KWS_SIMPLE_LSTM) {
        if (int rc = gru_train_fwd_bwd(m, a, static_cast<hipStream_t>(stream))) return rc;
        // recurrent models have no BatchNormalization and one bucket: the whole exchange behind the backward pass
        if (a->comm) return comm_allreduce_late(a->comm, a->grads, m->P, nullptr, 0, 1.f, static_cast<hipStream_t>(stream));
        return KWS_OK;
    }
    CnnWs w;
    int rc = check_ws(m, a->B, true, a->ws, a->ws_bytes, w);
    if (rc) return rc;
    m->prep = kws_model::Prepared{};       // a train step rewrites the BatchNorm coefficients (and is followed by a weight update)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool lite = m->kind == KWS_SIMPLE_CNN_LITE;
    bool grads_zeroed = false;     // cleared beside the weight split (the main chain joins that branch before conv3)
    // point 0 = behind the last forward convolution (simple_cnn); swept again with the persistent featurizer (same box, ms/step):
    // 0: 0.722, behind the loss 0.753, behind the head's backward 0.721, behind the dense data gradient 0.734, behind BN4's
    // backward 0.760, behind conv4's data gradient 0.727
    OverlapHook hook;
    hook.ev = static_cast<hipEvent_t>(a->overlap_event); hook.cb = a->overlap_callback; hook.user = a->overlap_user;
    // simple_cnn: behind BatchNorm-4's activation kernel, in front of the dense forward product.  Same-box sweeps with the round-2 kernels
    // (ms per step at B = 4096, four different boxes): behind conv4's forward 0.679-0.697, HERE 0.681-0.691 (never worse than the former,
    // usually 2-6 us better), behind the dense forward 0.714, behind the forward pass 0.729, behind the dense data gradient 0.704, behind
    // BatchNorm-4's backward 0.721, behind conv4's data gradient 0.699.  Behind the head's backward kernel (point 2) was the best point on two
    // boxes (0.686) and the worst on two others (0.709-0.722, whatever the hardware-queue count): the featurizer then starts together with
    // the side stream's weight-gradient chain, and which of the two gets the chip first decides.  kws_model_set_overlap_point(m, 0 .. 7) re-runs the sweep.
    // Round 3, after the finalize-free BatchNorm chain (fewer, shorter kernels behind conv4): behind conv3's forward (point 10) 0.548, behind
    // conv4's forward 0.552, behind BN4's backward 0.556, point 6 0.561, behind layer 1 0.575, behind the loss 0.579 (two rounds, one box):
    // the featurizer's persistent blocks (62 KB of LDS per CU) then run beside conv4's forward, the activation and the Dense + head kernel,
    // which fit beside them, and are gone when conv4's data gradient needs 147 KB of every CU.
    hook.at = lite ? 1 : (m->overlap_point >= 0 ? m->overlap_point : 10);     // kws_model_set_overlap_point re-runs the sweep
    rc = lite ? lite_forward(m, a->feat, a->B, a->params, a->state, w, true, a->dropout_seed, s)
              : cnn_forward(m, a->feat, a->B, a->params, a->state, w, true, a->dropout_seed, s, a->grads, &grads_zeroed, &hook, a->feat_moments);
    if (rc) return rc;
    if (lite) KWS_TRY(hook.fire(1, s));
    // Keras reduces the per-sample losses with a batch mean (train.py:75-77): d(mean)/d(logits) carries 1/B
    // simple_cnn: the head's backward kernel also sums the per-sample losses (no separate loss_reduce launch)
    const bool fuse_stats = !lite && head_bwd_fuses(m);
    // simple_cnn (non-deterministic mode): the head's forward pass rides in its backward kernel (kws_layers.h: head_bwd_mfma_kernel<.., FWD>):
    // one launch and the dlogits round trip less on the main chain (same-box: -11 us upper bound measured by skipping the launch)
    const bool fuse_head_fwd = fuse_stats && !m->deterministic;
    if (!fuse_head_fwd) {
        rc = run_head(m, a->B, a->params, w.d1, w.loss_i, w.correct_i, a->labels, a->class_weights, a->probs, nullptr, w.dlogits,
                      a->grad_scale / (float)a->B, fuse_stats ? nullptr : a->stats, a->ignore_index, s);
        if (rc) return rc;
        if (a->forward_event) KWS_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(a->forward_event), s));
        if (!lite) KWS_TRY(hook.fire(1, s));
    }
    rc = lite ? lite_backward(m, a->feat, a->B, a->params, a->grads, w, a->dropout_seed, static_cast<hipEvent_t>(a->bucket_event), s)
              : cnn_backward(m, a->feat, a->B, a->params, a->grads, w, a->dropout_seed, static_cast<hipEvent_t>(a->bucket_event), s,
                             fuse_stats ? a->stats : nullptr, grads_zeroed, a->feat_moments, &hook, a->comm, fuse_head_fwd ? a : nullptr);
    if (rc || !a->comm) return rc;
    // simple_cnn reduced its early bucket on the side stream (joined again by now); the rest, and the BatchNormalization moving
    // statistics, go behind the backward pass on the caller's stream.  simple_cnn_lite has no side stream: both buckets here.
    if (lite) KWS_TRY(comm_allreduce_early(a->comm, a->grads + m->o_k[3], m->P - m->o_k[3], s));
    return comm_allreduce_late(a->comm, a->grads, m->o_k[3], a->state, m->S, a->comm_state_weight, s);
}

int kws_set_matrix_precision(int mode)
{
    if (mode != KWS_MATRIX_FP32 && mode != KWS_MATRIX_BF16X6) return fail(KWS_ERR_INVALID, "unknown matrix precision %d", mode);
    g_matrix_precision = mode;
    return KWS_OK;
}

int kws_get_matrix_precision(void) { return g_matrix_precision; }

int kws_set_inference_precision(int mode)
{
    if (mode != KWS_INFER_FP32 && mode != KWS_INFER_FP16) return fail(KWS_ERR_INVALID, "unknown inference precision %d", mode);
    g_infer_precision = mode;
    return KWS_OK;
}
int kws_get_inference_precision(void) { return g_infer_precision; }

int64_t kws_model_grad_split(const kws_model *m)
{
    return !m ? 0 : ((m->kind == KWS_SIMPLE_CNN || m->kind == KWS_SIMPLE_CNN_LITE) ? m->o_k[3] : 0);
}

int kws_loss_forward(const float *y_pred, const int32_t *labels, const float *class_weights, int from_logits,
                     int ignore_index, int B, int C, float *losses, void *stream)
{
    if (!y_pred || !labels || !losses) return fail(KWS_ERR_INVALID, "null argument");
    if (B < 0 || C < 1) return fail(KWS_ERR_INVALID, "bad shape");
    if (B == 0) return KWS_OK;
    KWS_LAUNCH("loss_forward_kernel", loss_forward_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
               y_pred, labels, class_weights, from_logits, ignore_index, B, C, losses);
    KWS_LAUNCH_CHECK("loss_forward_kernel");
    return KWS_OK;
}

int kws_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t t, float grad_scale, void *stream)
{
    if (!params || !grads || !m || !v) return fail(KWS_ERR_INVALID, "null argument");
    if (n < 0 || t < 1) return fail(KWS_ERR_INVALID, "n must be >= 0 and the step count t >= 1");
    if (n == 0) return KWS_OK;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return fail(KWS_ERR_INVALID, "Adam buffers must be 16-byte aligned");
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    KWS_LAUNCH("adam_kernel", adam_kernel, dim3(blocks_for((n + 3) / 4, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), params,
                       grads, m, v, (long)n, (float)lr_t, beta1, beta2, eps, grad_scale);
    KWS_LAUNCH_CHECK("adam_kernel");
    return KWS_OK;
}

int kws_confusion_counts(const int32_t *labels, const int32_t *pred, int B, int C, int32_t *counts, void *stream)
{
    if (!labels || !pred || !counts || B < 0 || C < 1) return fail(KWS_ERR_INVALID, "bad argument");
    if (B == 0) return KWS_OK;
    KWS_LAUNCH("confusion_kernel", confusion_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), labels, pred, B,
               C, counts);
    KWS_LAUNCH_CHECK("confusion_kernel");
    return KWS_OK;
}

int kws_sgd_step(float *params, const float *grads, int64_t n, float lr, float grad_scale, void *stream)
{
    if (!params || !grads || n < 0) return fail(KWS_ERR_INVALID, "bad argument");
    if (n == 0) return KWS_OK;
    KWS_LAUNCH("sgd_kernel", sgd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads,
               (long)n, lr, grad_scale);
    KWS_LAUNCH_CHECK("sgd_kernel");
    return KWS_OK;
}

int kws_rmsprop_step(float *params, const float *grads, float *accum, int64_t n, float lr, float rho, float eps,
                     float grad_scale, void *stream)
{
    if (!params || !grads || !accum || n < 0) return fail(KWS_ERR_INVALID, "bad argument");
    if (n == 0) return KWS_OK;
    KWS_LAUNCH("rmsprop_kernel", rmsprop_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), params,
               grads, accum, (long)n, lr, rho, eps, grad_scale);
    KWS_LAUNCH_CHECK("rmsprop_kernel");
    return KWS_OK;
}

}  // extern "C"
