// csrc/kws_stream.hip -- streaming post-processing for S concurrent audio streams (gfx950 only).
//
// Replaces the per-chunk Python of the reference's listen.py, batched over streams:
//   ThresholdDecoder          listen.py:452-522   -> kws_decoder_* (table built once on the host in double, kept on device)
//   TriggerDetector.update    listen.py:538-559   -> kws_trigger_update
//   update_vectors (rows)     listen.py:107-109   -> kws_stream_push_rows
//   the prediction loop       listen.py:361-375   -> kws_stream_postprocess (argmax + max + decode + trigger, one kernel)
// Everything here is a few bytes per stream and one thread per stream: the kernels are latency-sized, the point of doing
// them on the device is that probabilities, scores and detector state never leave HBM between the forward pass of one
// chunk and the next (the whole step can sit in one hipGraph).  Arithmetic is float64 wherever the reference computes
// with Python floats, so activation decisions are identical, not just close.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "kws_common.h"

struct kws_decoder {
    int32_t min_out, out_range;
    int64_t n;                 // len(cd)
    double center;
    std::vector<double> cd;    // host copy (encode, kws_decoder_table)
    double *d_cd;              // device copy
};

namespace kws {

struct DecDev {
    const double *cd;
    long n;
    int min_out, out_range;
    double center;
    int enabled;
};

// ThresholdDecoder.decode, listen.py:496-508.  `odds` is 1/x - 1 as the caller's precision produced it.
__device__ __forceinline__ double decode_one(double x, double odds, const DecDev &d)
{
    if (x == 1.0 || x == 0.0) return x;
    double cp;
    if (d.out_range == 0) {
        cp = x > (double)d.min_out ? 1.0 : 0.0;
    } else {
        const double logit = (x > 0.0 && x < 1.0) ? -log(odds) : -10.0;            // asigmoid, listen.py:480-485
        double ratio = (logit - (double)d.min_out) / (double)d.out_range;
        ratio = fmin(fmax(ratio, 0.0), 1.0);
        cp = d.cd[(long)(ratio * (double)(d.n - 1) + 0.5)];
    }
    return cp < d.center ? 0.5 * cp / d.center : 0.5 + 0.5 * (cp - d.center) / (1.0 - d.center);
}

__device__ __forceinline__ double decode_f64(double x, const DecDev &d) { return decode_one(x, 1.0 / x - 1.0, d); }
// float32 network output: numpy evaluates `1 / x - 1` in float32 (listen.py:361-367 hands decode a float32 array)
__device__ __forceinline__ double decode_f32(float x, const DecDev &d) { return decode_one((double)x, (double)(1.0f / x - 1.0f), d); }

template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(const T *__restrict__ raw, double *__restrict__ out, long n, DecDev d)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (sizeof(T) == 4) out[i] = decode_f32((float)raw[i], d);
    else out[i] = decode_f64((double)raw[i], d);
}

// TriggerDetector.update, listen.py:538-559.  st = {activation, record_index}
__device__ __forceinline__ int trigger_one(int index, double score, int background, double sensitivity, int level, int refractory,
                                           int32_t *st)
{
    int act = st[0];
    const int rec = st[1];
    if (index != background && index == rec && score > sensitivity) {
        act += 1;
        if (act > level) {
            st[0] = refractory;            // -(8 * 2048) // chunk_size; record_index stays (it equals index)
            return 1;
        }
    } else if (act < 0) {
        act += 1;
    } else if (act > 0) {
        act -= 1;
    }
    st[0] = act;
    st[1] = index;
    return 0;
}

__global__ __launch_bounds__(256) void trigger_kernel(const int32_t *__restrict__ index, const double *__restrict__ score, int S,
                                                       int background, double sensitivity, int level, int refractory,
                                                       int32_t *__restrict__ state, int32_t *__restrict__ fired)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    fired[s] = trigger_one(index[s], score[s], background, sensitivity, level, refractory, state + 2 * (long)s);
}

// one thread per stream: first maximum (np.argmax), decode unless background, detector update
__global__ __launch_bounds__(256) void postprocess_kernel(const float *__restrict__ probs, int S, int C, int background, DecDev d,
                                                           double sensitivity, int level, int refractory,
                                                           int32_t *__restrict__ state, int32_t *__restrict__ index,
                                                           double *__restrict__ score, int32_t *__restrict__ fired)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const float *p = probs + (long)s * C;
    float best = p[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c];
        if (v > best) { best = v; arg = c; }
    }
    double sc = (double)best;
    if (arg != background && d.enabled) sc = decode_f32(best, d);
    index[s] = arg;
    score[s] = sc;
    fired[s] = trigger_one(arg, sc, background, sensitivity, level, refractory, state + 2 * (long)s);
}

// block = one stream: every element is read into a register before the barrier, so the in-place shift has no hazards
__global__ __launch_bounds__(256) void push_rows_kernel(float *__restrict__ feat, const float *__restrict__ rows, int F, int D,
                                                         int n_rows, int n_keep)
{
    const int s = blockIdx.x;
    float *f = feat + (long)s * F * D;
    const float *r = rows + (long)s * n_rows * D + (long)(n_rows - n_keep) * D;     // the last n_keep new rows
    const int total = F * D, shift = n_keep * D;
    constexpr int kPer = 8;                                                          // F*D <= 2048 per pass
    for (int base = 0; base < total; base += 256 * kPer) {
        float v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = base + threadIdx.x + 256 * j;
            v[j] = 0.f;
            if (i < total) v[j] = i + shift < total ? f[i + shift] : r[i + shift - total];
        }
        __syncthreads();     // the reads of this pass precede its writes; later passes only read further right
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = base + threadIdx.x + 256 * j;
            if (i < total) f[i] = v[j];
        }
        __syncthreads();
    }
}

static DecDev dec_dev(const kws_decoder *d)
{
    DecDev v;
    if (!d) {
        v.cd = nullptr; v.n = 0; v.min_out = 0; v.out_range = 0; v.center = 0.5; v.enabled = 0;
    } else {
        v.cd = d->d_cd; v.n = (long)d->n; v.min_out = d->min_out; v.out_range = d->out_range; v.center = d->center; v.enabled = 1;
    }
    return v;
}

// Python floor division of -(8 * 2048) by chunk_size (listen.py:549)
static int refractory_of(int chunk_size)
{
    const int a = -(8 * 2048);
    int q = a / chunk_size;
    if ((a % chunk_size != 0) && ((a < 0) != (chunk_size < 0))) --q;
    return q;
}

}  // namespace kws

using namespace kws;

extern "C" {

int kws_decoder_create(const double *mu_stds, int n, double center, int resolution, double min_z, double max_z, kws_decoder **out)
{
    if (!mu_stds || !out || n < 1) return fail(KWS_ERR_INVALID, "threshold_config needs at least one (mu, std) pair");
    if (resolution < 1) return fail(KWS_ERR_INVALID, "resolution must be positive, got %d", resolution);
    // listen.py:468-470: int() truncates toward zero
    double lo = mu_stds[0] + min_z * mu_stds[1], hi = mu_stds[0] + max_z * mu_stds[1];
    for (int i = 1; i < n; ++i) {
        lo = std::min(lo, mu_stds[2 * i] + min_z * mu_stds[2 * i + 1]);
        hi = std::max(hi, mu_stds[2 * i] + max_z * mu_stds[2 * i + 1]);
    }
    if (!(std::fabs(lo) < 1e6) || !(std::fabs(hi) < 1e6)) return fail(KWS_ERR_INVALID, "threshold_config out of range");
    auto *d = new kws_decoder();
    d->min_out = (int32_t)lo;
    const int32_t max_out = (int32_t)hi;
    d->out_range = max_out - d->min_out;
    d->center = center;
    d->d_cd = nullptr;
    const int64_t npts = (int64_t)resolution * d->out_range;
    if (npts < 0 || npts > (int64_t)1 << 26) { delete d; return fail(KWS_ERR_INVALID, "threshold table of %lld points", (long long)npts); }
    if (npts == 0) {
        d->cd.assign(1, 0.0);          // np.cumsum of the scalar 0.0 (listen.py:519-522 with an empty linspace)
    } else {
        // _calc_pd: np.linspace(min_out, max_out, npts) -> i * step + start, last point exactly max_out
        std::vector<double> dens((size_t)npts, 0.0);
        const double start = (double)d->min_out, stop = (double)max_out;
        const double step = npts > 1 ? (stop - start) / (double)(npts - 1) : 0.0;
        for (int c = 0; c < n; ++c) {
            const double mu = mu_stds[2 * c], sd = mu_stds[2 * c + 1];
            if (sd == 0.0) continue;                                       // pdf() returns 0, listen.py:491-492
            const double norm = 1.0 / (sd * std::sqrt(2 * M_PI)), den = 2 * (sd * sd);
            for (int64_t i = 0; i < npts; ++i) {
                const double x = (npts > 1 && i == npts - 1) ? stop : (double)i * step + start;
                const double a = x - mu;
                dens[(size_t)i] += norm * std::exp(-(a * a) / den);
            }
        }
        d->cd.resize((size_t)npts);
        const double div = (double)((int64_t)resolution * n);
        double run = 0.0;
        for (int64_t i = 0; i < npts; ++i) { run += dens[(size_t)i] / div; d->cd[(size_t)i] = run; }
    }
    d->n = (int64_t)d->cd.size();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        delete d;
        return fail(KWS_ERR_HIP, "no HIP device: the threshold decoder has no CPU fallback");
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d->d_cd), sizeof(double) * d->cd.size());
    if (e == hipSuccess) e = hipMemcpy(d->d_cd, d->cd.data(), sizeof(double) * d->cd.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d->d_cd) (void)hipFree(d->d_cd);
        delete d;
        return fail(KWS_ERR_HIP, "threshold table upload failed: %s", hipGetErrorString(e));
    }
    *out = d;
    return KWS_OK;
}

void kws_decoder_destroy(kws_decoder *d)
{
    if (!d) return;
    if (d->d_cd) (void)hipFree(d->d_cd);
    delete d;
}

int kws_decoder_info(const kws_decoder *d, int32_t *min_out, int32_t *out_range, int64_t *table_len)
{
    if (!d) return fail(KWS_ERR_INVALID, "null decoder");
    if (min_out) *min_out = d->min_out;
    if (out_range) *out_range = d->out_range;
    if (table_len) *table_len = d->n;
    return KWS_OK;
}

int kws_decoder_table(const kws_decoder *d, double *host_cd, size_t count)
{
    if (!d || !host_cd) return fail(KWS_ERR_INVALID, "null argument");
    if (count != d->cd.size()) return fail(KWS_ERR_INVALID, "table has %zu entries, caller asked for %zu", d->cd.size(), count);
    std::memcpy(host_cd, d->cd.data(), count * sizeof(double));
    return KWS_OK;
}

int kws_decoder_decode(const kws_decoder *d, const void *raw, int raw_dtype, double *decoded, int64_t n, void *stream)
{
    if (!d || (!raw && n > 0) || (!decoded && n > 0)) return fail(KWS_ERR_INVALID, "null argument");
    if (n < 0) return fail(KWS_ERR_INVALID, "negative count");
    if (n == 0) return KWS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (raw_dtype == KWS_RAW_F64)
        KWS_LAUNCH("decode_kernel_f64", decode_kernel<double>, grid, block, 0, s, static_cast<const double *>(raw), decoded, (long)n, dec_dev(d));
    else if (raw_dtype == KWS_RAW_F32)
        KWS_LAUNCH("decode_kernel_f32", decode_kernel<float>, grid, block, 0, s, static_cast<const float *>(raw), decoded, (long)n, dec_dev(d));
    else
        return fail(KWS_ERR_INVALID, "unknown raw dtype %d", raw_dtype);
    KWS_LAUNCH_CHECK("decode_kernel");
    return KWS_OK;
}

int kws_decoder_encode(const kws_decoder *d, double threshold, double *raw_out)
{
    if (!d || !raw_out) return fail(KWS_ERR_INVALID, "null argument");
    // listen.py:510-517
    const double t = 0.5 * threshold / d->center;
    const double cp = t < 0.5 ? t * d->center * 2 : (t - 0.5) * 2 * (1 - d->center) + d->center;
    const double ratio = (double)(std::lower_bound(d->cd.begin(), d->cd.end(), cp) - d->cd.begin()) / (double)d->cd.size();
    *raw_out = 1.0 / (1.0 + std::exp(-((double)d->min_out + (double)d->out_range * ratio)));
    return KWS_OK;
}

int kws_stream_push_rows(float *feat, const float *rows, int S, int F, int D, int n_rows, void *stream)
{
    if (S < 0 || F < 1 || D < 1 || n_rows < 0) return fail(KWS_ERR_INVALID, "bad stream geometry S=%d F=%d D=%d n_rows=%d", S, F, D, n_rows);
    if (S == 0 || n_rows == 0) return KWS_OK;
    if (!feat || !rows) return fail(KWS_ERR_INVALID, "null argument");
    KWS_LAUNCH("push_rows_kernel", push_rows_kernel, dim3((unsigned)S), dim3(256), 0, static_cast<hipStream_t>(stream), feat, rows, F, D,
               n_rows, std::min(n_rows, F));
    KWS_LAUNCH_CHECK("push_rows_kernel");
    return KWS_OK;
}

int kws_trigger_update(const int32_t *index, const double *score, int S, int background_index, double sensitivity, int trigger_level,
                       int chunk_size, int32_t *state, int32_t *fired, void *stream)
{
    if (S < 0 || chunk_size == 0) return fail(KWS_ERR_INVALID, "bad S=%d or chunk_size=%d", S, chunk_size);
    if (S == 0) return KWS_OK;
    if (!index || !score || !state || !fired) return fail(KWS_ERR_INVALID, "null argument");
    KWS_LAUNCH("trigger_kernel", trigger_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), index,
               score, S, background_index, sensitivity, trigger_level, refractory_of(chunk_size), state, fired);
    KWS_LAUNCH_CHECK("trigger_kernel");
    return KWS_OK;
}

int kws_stream_postprocess(const kws_decoder *dec, const float *probs, int S, int C, int background_index, double sensitivity,
                           int trigger_level, int chunk_size, int32_t *state, int32_t *index, double *score, int32_t *fired,
                           void *stream)
{
    if (S < 0 || C < 1 || chunk_size == 0) return fail(KWS_ERR_INVALID, "bad S=%d C=%d chunk_size=%d", S, C, chunk_size);
    if (S == 0) return KWS_OK;
    if (!probs || !state || !index || !score || !fired) return fail(KWS_ERR_INVALID, "null argument");
    KWS_LAUNCH("postprocess_kernel", postprocess_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
               probs, S, C, background_index, dec_dev(dec), sensitivity, trigger_level, refractory_of(chunk_size), state, index, score,
               fired);
    KWS_LAUNCH_CHECK("postprocess_kernel");
    return KWS_OK;
}

}  // extern "C"
