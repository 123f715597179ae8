/*
 * oracle/kws_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double precision) of the keyword-spotting
 * featurizer the reference runs on its hot path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call this.
 *
 * Where the arithmetic comes from
 * -------------------------------
 * The reference calls the third-party package `sonopy` (unpinned in
 * requirements.txt:7; latest published release 0.1.2) at
 * common/data_utils.py:69:
 *     sonopy.mfcc_spec(audio, sample_rate, (window_samples, hop_samples),
 *                      num_filt=n_filt, fft_size=n_fft, num_coeffs=n_mfcc)
 * sonopy's source is NOT under /root/reference.  Its published algorithm is
 * restated here and is corroborated line by line by two in-repo restatements:
 *   - common/bark_feature.py:75-89   safe_log / chop_array / power_spec
 *   - common/bark_feature.py:156-175 bfcc_spec (same log/DCT/c0 tail)
 *   - inference/tflite/mfcc.h:214-264,272-290,42-71,345-359 (C++ twin)
 *
 * Pinning: parity is pinned by fixtures under tests/golden/ generated in the
 * build container from (i) the reference's mfcc.h compiled as oracle/_ref and
 * (ii) the reference's own common/bark_feature.py functions imported from
 * /root/reference (tests/golden/make_golden.py).  The reference holds no
 * tests or golden vectors of its own for this path.
 */
#ifndef KWS_ORACLE_H
#define KWS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors classifier/params.py:49-59 (the fields the featurizer reads) */
typedef struct {
    double buffer_t, window_t, hop_t;
    int sample_rate, sample_depth, n_fft, n_filt, n_mfcc, use_delta;
} oracle_params;

/* derived geometry, classifier/params.py:59-91 */
int oracle_window_samples(const oracle_params *p);
int oracle_hop_samples(const oracle_params *p);
int oracle_max_samples(const oracle_params *p);
int oracle_buffer_samples(const oracle_params *p);
int oracle_n_features(const oracle_params *p);
int oracle_feature_size(const oracle_params *p);

/* sonopy.filterbanks grid: n_filt+2 bin indices (mfcc.h:235-248).  Returns 0,
 * or -1 if the grid is not strictly increasing (sonopy's duplicate-point
 * correction is not restated; such configs are refused). */
int oracle_mel_points(int sample_rate, int n_fft, int n_filt, int *pts);

/* dense (n_filt x (n_fft/2+1)) row-major banks */
int oracle_mel_bank(int sample_rate, int n_fft, int n_filt, double *bank);
/* common/bark_feature.py:92-136, scale="constant", with its nfft=512 quirk */
int oracle_bark_bank(int sample_rate, int n_fft, int n_filt, double *bank);

/* power spectrogram of already-framed audio: (n_frames x (n_fft/2+1)).
 * bark_feature.py:80-89 */
int oracle_power_spec(const double *audio, int n, int window, int hop,
                      int n_fft, double *powers /* may be NULL */);

/* sonopy.mfcc_spec on raw audio of any length (vectorize_raw,
 * common/data_utils.py:61-70).  bank_kind 0 = mel, 1 = bark.
 * out: (n_frames x min(n_filt,n_mfcc)).  Returns n_frames or <0. */
int oracle_mfcc_spec(const double *audio, int n, const oracle_params *p,
                     int bank_kind, double *out);

/* audio_to_feature, common/data_utils.py:73-86: keep the first max_samples,
 * left-pad zeros if shorter, mfcc_spec, optional add_deltas (:50-58).
 * out: (n_features x feature_size).  Returns 0 or <0. */
int oracle_audio_to_feature(const double *audio, int n, const oracle_params *p,
                            int bank_kind, double *out);

/* batched convenience used by bench.py's cpu_baseline leg:
 * wav (B x stride) float32, valid_len may be NULL (= stride each). */
int oracle_featurize_batch_f32(const float *wav, int B, int stride,
                               const int *valid_len, const oracle_params *p,
                               int bank_kind, float *out);

#ifdef __cplusplus
}
#endif
#endif
