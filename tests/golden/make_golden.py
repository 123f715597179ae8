#!/usr/bin/env python3
"""Generate the featurizer golden fixtures (run in the BUILD container only).

Inputs  : /root/reference/example/*.wav (the reference's only real-audio
          fixtures; all mono / 16 kHz / 16-bit / 16 000 frames) and seeded
          synthetic clips.
Outputs : tests/golden/featurizer_golden.npz  (data only: PCM + expected
          feature matrices + banks), tests/golden/params_defaults.json.

Expected outputs come from the REFERENCE ITSELF, two ways:
  (A) inference/tflite/mfcc.h compiled unmodified into oracle/_ref (g++), driven
      exactly as inference/tflite/speech_commands.h:293-316 drives it
      (float and double instantiations);
  (B) the reference's own Python functions imported from /root/reference:
      common/bark_feature.py power_spec / safe_log / bark_filterbanks /
      bfcc_spec (its module-level `import librosa` is only used under
      __main__, so an empty module object satisfies the import, as in
      SURVEY.md section 8c), and classifier/params.py for the geometry.
      The mel chain (B-mel) composes bf.power_spec + bf.safe_log with the
      sonopy.filterbanks grid restated from inference/tflite/mfcc.h:230-264 and
      scipy.fftpack.dct(norm='ortho') -- the same composition SURVEY.md 8c
      validated against (A) to 4.7e-7.  The bark chain (B-bark) is
      bf.bfcc_spec end to end, no restatement involved.
Nothing of the reference's source text is stored; only arrays.
"""
import json
import os
import sys
import types
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, ROOT)
from oracle import featurizer_oracle as fo  # noqa: E402  (only for the compiled-reference driver)


def read_wav_i16(path):
    w = wave.open(path)
    assert w.getnchannels() == 1 and w.getframerate() == 16000 and w.getsampwidth() == 2
    return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()


def main():
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    sys.path.insert(0, REF)
    import common.bark_feature as bf
    from classifier.params import pr
    from scipy.fftpack import dct

    fo.build(force=True)
    out = {}
    geom = dict(n_features=pr.n_features, feature_size=pr.feature_size, window_samples=pr.window_samples,
                hop_samples=pr.hop_samples, max_samples=pr.max_samples, buffer_samples=pr.buffer_samples)
    with open(os.path.join(HERE, "params_defaults.json"), "w") as f:
        d = {k: (list(map(list, v)) if k == "threshold_config" else v) for k, v in pr.__dict__.items()}
        json.dump({"params": d, "derived": geom}, f, indent=2)

    mel_bank = fo.numpy_mel_bank(pr.sample_rate, pr.n_fft, pr.n_filt)  # grid of mfcc.h:230-264

    def py_mel_chain(audio):
        powers = bf.power_spec(audio, (pr.window_samples, pr.hop_samples), pr.n_fft)
        mels = bf.safe_log(np.dot(powers, mel_bank.T))
        m = dct(mels, norm="ortho")[:, :pr.n_mfcc]
        m[:, 0] = bf.safe_log(np.sum(powers, 1))
        return m

    def py_bark_chain(audio):
        return bf.bfcc_spec(audio, pr.sample_rate, pr.window_samples, pr.hop_samples, fft_size=pr.n_fft,
                            num_filt=pr.n_filt, num_coeffs=pr.n_mfcc)

    names = ["right_1", "left_1", "up_1", "down_1", "right_2", "left_2", "up_2", "down_2"]   # every clip under example/
    for n in names:
        pcm = read_wav_i16(os.path.join(REF, "example", n + ".wav"))
        assert len(pcm) == 16000
        a32 = pcm.astype(np.float32) / 32768.0          # librosa.load / buffer_to_audio scaling (data_utils.py:21)
        a64 = a32.astype(np.float64)
        out["pcm_" + n] = pcm
        out["refcpp_f32_" + n] = fo.ref_mfcc(a32, np.float32)
        out["refcpp_f64_" + n] = fo.ref_mfcc(a64, np.float64)
        out["refpy_mel_" + n] = py_mel_chain(a64)
        out["refpy_bark_" + n] = py_bark_chain(a64)

    # synthetic: short clip -> left zero pad (data_utils.py:79-80) -> all-zero leading frames hit the log floor
    rng = np.random.default_rng(0)
    short = np.round(np.clip(0.1 * rng.standard_normal(9000), -1, 1 - 2.0 ** -15) * 32768) / 32768
    padded = np.concatenate([np.zeros((pr.max_samples - len(short),)), short])
    out["syn_short_audio"] = short
    out["refpy_mel_syn_short"] = py_mel_chain(padded)
    out["refpy_bark_syn_short"] = py_bark_chain(padded)
    # synthetic: long clip -> head kept (data_utils.py:77)
    longc = np.round(np.clip(0.2 * rng.standard_normal(20000), -1, 1 - 2.0 ** -15) * 32768) / 32768
    out["syn_long_audio"] = longc
    out["refpy_mel_syn_long"] = py_mel_chain(longc[:pr.max_samples])
    out["refcpp_f64_syn_long"] = fo.ref_mfcc(longc[:pr.max_samples], np.float64)
    # pure tone + silence
    t = np.arange(16000) / 16000.0
    tone = np.round(0.5 * np.sin(2 * np.pi * 1000.0 * t) * 32768) / 32768
    out["syn_tone_audio"] = tone
    out["refpy_mel_syn_tone"] = py_mel_chain(tone)
    out["refcpp_f64_syn_tone"] = fo.ref_mfcc(tone, np.float64)
    out["refpy_mel_silence"] = py_mel_chain(np.zeros(16000))

    out["bark_bank_20x513"] = np.asarray(bf.bark_filterbanks(nfilts=20, nfft=1024, sample_rate=16000, low_freq=0,
                                                             high_freq=None, scale="constant"))
    out["power_spec_right_1"] = bf.power_spec(out["pcm_right_1"].astype(np.float64) / 32768.0, (1024, 512), 1024)
    np.savez_compressed(os.path.join(HERE, "featurizer_golden.npz"), **out)
    for k, v in out.items():
        print("%-28s %s %s" % (k, v.shape, v.dtype))
    d = np.abs(out["refcpp_f64_right_1"] - out["refpy_mel_right_1"]).max()
    print("max |refcpp_f64 - refpy_mel| on right_1:", d)


if __name__ == "__main__":
    main()
