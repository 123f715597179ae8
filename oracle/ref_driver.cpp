// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" driver around the REFERENCE's own header-only C++ MFCC
// (inference/tflite/mfcc.h), compiled from where it lies under /root/reference
// (never copied into this repo) into oracle/_ref/libmfcc_ref.so by
// oracle/Makefile.  It is used to pin oracle/kws_oracle.c and to generate
// tests/golden fixtures; optionally as the "reference" CPU baseline.
//
// The header uses std::all_of / std::max_element / assert without including
// <algorithm>/<cassert> (mfcc.h:193,196,306), so they are supplied here.
// The call below follows the reference's own call site,
// inference/tflite/speech_commands.h:293-316 (vectorize()).
#include <algorithm>
#include <cassert>
#include <type_traits>
#include <vector>
#include "mfcc.h"   // -I/root/reference/inference/tflite

extern "C" int ref_mfcc_f32(const float *audio, int n, int sample_rate, int length_frame, int stride,
                            int length_fft, int num_coeffs, int num_filters, int low_freq, int high_freq,
                            int use_delta, float *out /* n_frames x feature_size */)
{
    std::vector<float> a(audio, audio + n);
    std::vector<std::vector<float>> fv;
    mfcc::mfcc<float>(fv, a, sample_rate, length_frame, stride, length_fft, num_coeffs, num_filters,
                      low_freq, high_freq, /*use_preprocess=*/false, use_delta != 0, /*use_delta2=*/false);
    size_t k = 0;
    for (auto &row : fv)
        for (float v : row) out[k++] = v;
    return (int)fv.size();
}

extern "C" int ref_mfcc_f64(const double *audio, int n, int sample_rate, int length_frame, int stride,
                            int length_fft, int num_coeffs, int num_filters, int low_freq, int high_freq,
                            int use_delta, double *out)
{
    std::vector<double> a(audio, audio + n);
    std::vector<std::vector<double>> fv;
    mfcc::mfcc<double>(fv, a, sample_rate, length_frame, stride, length_fft, num_coeffs, num_filters,
                       low_freq, high_freq, false, use_delta != 0, false);
    size_t k = 0;
    for (auto &row : fv)
        for (double v : row) out[k++] = v;
    return (int)fv.size();
}
