/*
 * oracle/kws_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See kws_oracle.h for provenance and pinning.  Plain C99, double precision,
 * written for clarity not speed; every function cites the reference lines
 * (relative to /root/reference) whose behaviour it restates.
 */
#include "kws_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* np.finfo(float).eps -- common/bark_feature.py:77 */
#define ORACLE_EPS 2.220446049250313e-16

/* ---- classifier/params.py:59-91 ---------------------------------------- */
int oracle_window_samples(const oracle_params *p) { return (int)(p->sample_rate * p->window_t + 0.5); } /* :73 */
int oracle_hop_samples(const oracle_params *p) { return (int)(p->sample_rate * p->hop_t + 0.5); }       /* :78 */
int oracle_max_samples(const oracle_params *p) { return (int)(p->buffer_t * p->sample_rate); }          /* :83 */
int oracle_buffer_samples(const oracle_params *p)                                                        /* :60-63 */
{
    int samples = (int)(p->sample_rate * p->buffer_t + 0.5);
    int hop = oracle_hop_samples(p);
    return hop * (samples / hop);
}
int oracle_n_features(const oracle_params *p)                                                            /* :66-68 */
{
    double q = (double)(oracle_buffer_samples(p) - oracle_window_samples(p)) / (double)oracle_hop_samples(p);
    return 1 + (int)floor(q);
}
int oracle_feature_size(const oracle_params *p) { return p->use_delta ? 2 * p->n_mfcc : p->n_mfcc; }    /* :86-91 */

/* ---- mel grid: sonopy.filterbanks / inference/tflite/mfcc.h:134-145,235-248 */
static double hz_to_mel(double f) { return 1127.0 * log(1.0 + f / 700.0); }
static double mel_to_hz(double m) { return 700.0 * (exp(m / 1127.0) - 1.0); }

int oracle_mel_points(int sample_rate, int n_fft, int n_filt, int *pts)
{
    /* span is 0 Hz .. sample_rate (NOT Nyquist): speech_commands.h:304-307 */
    int n_bins = n_fft / 2 + 1, n = n_filt + 2, i;
    double lo = hz_to_mel(0.0), hi = hz_to_mel((double)sample_rate);
    double step = (hi - lo) / (double)(n - 1); /* np.linspace */
    for (i = 0; i < n; i++) {
        double m = (i == n - 1) ? hi : lo + i * step;
        double hz = mel_to_hz(m);
        pts[i] = (int)(hz * n_bins / sample_rate); /* mfcc.h:245, trunc toward 0 */
    }
    for (i = 1; i < n; i++)
        if (pts[i] <= pts[i - 1]) return -1;
    if (pts[n - 1] > n_bins) return -1;
    return 0;
}

int oracle_mel_bank(int sample_rate, int n_fft, int n_filt, double *bank)
{
    int n_bins = n_fft / 2 + 1, i, j;
    int *pts = (int *)malloc(sizeof(int) * (size_t)(n_filt + 2));
    if (!pts) return -2;
    if (oracle_mel_points(sample_rate, n_fft, n_filt, pts)) { free(pts); return -1; }
    memset(bank, 0, sizeof(double) * (size_t)n_filt * (size_t)n_bins);
    for (i = 0; i < n_filt; i++) {
        int l = pts[i], m = pts[i + 1], r = pts[i + 2];
        /* np.linspace(0,1,m-l,endpoint=False) / np.linspace(1,0,r-m,False); mfcc.h:253-258 */
        for (j = l; j < m; j++) bank[i * n_bins + j] = (double)(j - l) / (double)(m - l);
        for (j = m; j < r; j++) bank[i * n_bins + j] = (double)(r - j) / (double)(r - m);
    }
    free(pts);
    return 0;
}

/* ---- bark bank: common/bark_feature.py:28-72,92-136 -------------------- */
static double hz2bark(double f) { return 6.0 * asinh(f / 600.0); }   /* :28-30 */
static double bark2hz(double b) { return 600.0 * sinh(b / 6.0); }    /* :33-35 */
/* :48-57 are called WITHOUT nfft/sample_rate at :112,134, so their defaults
 * (nfft=512, sample_rate=16000) apply whatever the caller's n_fft is. */
static double bark2fft_default(double b) { return (512 + 1) * bark2hz(b) / 16000.0; }
static double fft2bark_default(double j) { return hz2bark((j * 16000.0) / (512 + 1)); }
static double bark_Fm(double fb, double fc)                           /* :59-72 */
{
    if (fc - 2.5 <= fb && fb <= fc - 0.5) return pow(10.0, 2.5 * (fb - fc + 0.5));
    if (fc - 0.5 < fb && fb < fc + 0.5) return 1.0;
    if (fc + 0.5 <= fb && fb <= fc + 1.3) return pow(10.0, -2.5 * (fb - fc - 0.5));
    return 0.0;
}

int oracle_bark_bank(int sample_rate, int n_fft, int n_filt, double *bank)
{
    int n_bins = n_fft / 2 + 1, n = n_filt + 4, i, j;
    double lo = hz2bark(0.0), hi = hz2bark(sample_rate / 2.0); /* :105-110 */
    double step = (hi - lo) / (double)(n - 1);
    double *pt = (double *)malloc(sizeof(double) * (size_t)n);
    int *bins = (int *)malloc(sizeof(int) * (size_t)n);
    if (!pt || !bins) { free(pt); free(bins); return -2; }
    for (i = 0; i < n; i++) {
        pt[i] = (i == n - 1) ? hi : lo + i * step;
        bins[i] = (int)floor(bark2fft_default(pt[i]));               /* :112 */
    }
    memset(bank, 0, sizeof(double) * (size_t)n_filt * (size_t)n_bins);
    for (i = 0; i < n_filt; i++) {
        for (j = bins[i]; j < bins[i + 4]; j++) {                    /* :131 */
            if (j < 0 || j >= n_bins) { free(pt); free(bins); return -1; } /* numpy IndexError */
            bank[i * n_bins + j] = fabs(bark_Fm(fft2bark_default((double)j), pt[i + 2]));
        }
    }
    free(pt); free(bins);
    return 0;
}

/* ---- FFT (np.fft.rfft equivalent) -------------------------------------- */
static void fft_inplace(double *re, double *im, int n)
{
    int i, j, k, len;
    if ((n & (n - 1)) != 0) { /* not a power of two: plain DFT */
        double *tr = (double *)malloc(sizeof(double) * (size_t)n * 2), *ti = tr + n;
        for (k = 0; k < n; k++) {
            double sr = 0, si = 0;
            for (j = 0; j < n; j++) {
                double a = -2.0 * M_PI * (double)(((long long)j * k) % n) / n;
                sr += re[j] * cos(a) - im[j] * sin(a);
                si += re[j] * sin(a) + im[j] * cos(a);
            }
            tr[k] = sr; ti[k] = si;
        }
        memcpy(re, tr, sizeof(double) * (size_t)n);
        memcpy(im, ti, sizeof(double) * (size_t)n);
        free(tr);
        return;
    }
    for (i = 1, j = 0; i < n; i++) { /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (i = 0; i < n; i += len) {
            for (k = 0; k < half; k++) {
                double a = -2.0 * M_PI * k / len, wr = cos(a), wi = sin(a);
                double xr = re[i + k + half] * wr - im[i + k + half] * wi;
                double xi = re[i + k + half] * wi + im[i + k + half] * wr;
                re[i + k + half] = re[i + k] - xr; im[i + k + half] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
        }
    }
}

static int count_frames(int n, int window, int hop)
{
    /* chop_array: range(window, len+1, hop) -- bark_feature.py:80-82 */
    if (n < window) return 0;
    return (n - window) / hop + 1;
}

int oracle_power_spec(const double *audio, int n, int window, int hop, int n_fft, double *powers)
{
    int n_frames = count_frames(n, window, hop), n_bins = n_fft / 2 + 1, f, k;
    double *re, *im;
    if (!powers) return n_frames;
    re = (double *)malloc(sizeof(double) * (size_t)n_fft * 2);
    if (!re) return -2;
    im = re + n_fft;
    for (f = 0; f < n_frames; f++) {
        const double *fr = audio + (size_t)f * hop;
        /* np.fft.rfft(frames, n=fft_size): crop or zero-pad to n_fft -- :87 */
        for (k = 0; k < n_fft; k++) { re[k] = (k < window) ? fr[k] : 0.0; im[k] = 0.0; }
        fft_inplace(re, im, n_fft);
        for (k = 0; k < n_bins; k++)
            powers[(size_t)f * n_bins + k] = (re[k] * re[k] + im[k] * im[k]) / n_fft; /* :88 */
    }
    free(re);
    return n_frames;
}

static double safe_log(double x) { return log(x < ORACLE_EPS ? ORACLE_EPS : x); } /* bark_feature.py:75-77 */

int oracle_mfcc_spec(const double *audio, int n, const oracle_params *p, int bank_kind, double *out)
{
    int window = oracle_window_samples(p), hop = oracle_hop_samples(p);
    int n_bins = p->n_fft / 2 + 1, n_filt = p->n_filt;
    int n_out = p->n_mfcc < n_filt ? p->n_mfcc : n_filt;
    int n_frames = count_frames(n, window, hop), f, i, k, rc;
    double *powers, *bank, *mel;
    if (n_frames == 0) return 0; /* sonopy: empty result */
    powers = (double *)malloc(sizeof(double) * ((size_t)n_frames * n_bins + (size_t)n_filt * n_bins + (size_t)n_filt));
    if (!powers) return -2;
    bank = powers + (size_t)n_frames * n_bins;
    mel = bank + (size_t)n_filt * n_bins;
    rc = bank_kind == 0 ? oracle_mel_bank(p->sample_rate, p->n_fft, n_filt, bank)
                        : oracle_bark_bank(p->sample_rate, p->n_fft, n_filt, bank);
    if (rc) { free(powers); return rc; }
    oracle_power_spec(audio, n, window, hop, p->n_fft, powers);
    for (f = 0; f < n_frames; f++) {
        const double *P = powers + (size_t)f * n_bins;
        double energy = 0.0;
        for (k = 0; k < n_bins; k++) energy += P[k];
        for (i = 0; i < n_filt; i++) { /* np.dot(powers, filters.T) then safe_log: bark_feature.py:168-170 */
            double s = 0.0;
            for (k = 0; k < n_bins; k++) s += P[k] * bank[(size_t)i * n_bins + k];
            mel[i] = safe_log(s);
        }
        for (k = 0; k < n_out; k++) { /* scipy dct type II norm='ortho' -- :172, mfcc.h:55-67 */
            double s = 0.0;
            for (i = 0; i < n_filt; i++) s += mel[i] * cos(M_PI * (i + 0.5) * k / n_filt);
            out[(size_t)f * n_out + k] = s * (k == 0 ? sqrt(1.0 / n_filt) : sqrt(2.0 / n_filt));
        }
        out[(size_t)f * n_out] = safe_log(energy); /* :173, mfcc.h:358-359 */
    }
    free(powers);
    return n_frames;
}

int oracle_audio_to_feature(const double *audio, int n, const oracle_params *p, int bank_kind, double *out)
{
    int max_samples = oracle_max_samples(p), i, k, n_frames;
    int n_out = p->n_mfcc < p->n_filt ? p->n_mfcc : p->n_filt;
    double *buf, *base;
    if (n < 0) return -3;
    if (n > max_samples) n = max_samples;                 /* data_utils.py:77 keeps the HEAD */
    buf = (double *)calloc((size_t)max_samples, sizeof(double));
    if (!buf) return -2;
    memcpy(buf + (max_samples - n), audio, sizeof(double) * (size_t)n); /* :79-80 LEFT zero pad */
    if (!p->use_delta) {
        n_frames = oracle_mfcc_spec(buf, max_samples, p, bank_kind, out);
        free(buf);
        return n_frames;
    }
    base = (double *)malloc(sizeof(double) * (size_t)(count_frames(max_samples, oracle_window_samples(p), oracle_hop_samples(p)) + 1) * n_out);
    if (!base) { free(buf); return -2; }
    n_frames = oracle_mfcc_spec(buf, max_samples, p, bank_kind, base);
    for (i = 0; i < n_frames; i++) {                      /* add_deltas, data_utils.py:50-58 */
        for (k = 0; k < n_out; k++) {
            out[(size_t)i * 2 * n_out + k] = base[(size_t)i * n_out + k];
            out[(size_t)i * 2 * n_out + n_out + k] = i == 0 ? 0.0 : base[(size_t)i * n_out + k] - base[(size_t)(i - 1) * n_out + k];
        }
    }
    free(base); free(buf);
    return n_frames;
}

int oracle_featurize_batch_f32(const float *wav, int B, int stride, const int *valid_len,
                               const oracle_params *p, int bank_kind, float *out)
{
    int nf = oracle_n_features(p), fs = oracle_feature_size(p), b, err = 0;
    /* clips are independent; OMP_NUM_THREADS decides how many host cores are used */
#pragma omp parallel for schedule(dynamic, 4)
    for (b = 0; b < B; b++) {
        int n = valid_len ? valid_len[b] : stride, rc, i;
        double *a = (double *)malloc(sizeof(double) * ((size_t)stride + (size_t)nf * fs + 64));
        double *o;
        if (!a) { err = -2; continue; }
        o = a + stride;
        if (n > stride) n = stride;
        for (i = 0; i < n; i++) a[i] = (double)wav[(size_t)b * stride + i];
        rc = oracle_audio_to_feature(a, n, p, bank_kind, o);
        if (rc != nf) err = rc < 0 ? rc : -4;
        else for (i = 0; i < nf * fs; i++) out[(size_t)b * nf * fs + i] = (float)o[i];
        free(a);
    }
    return err;
}
