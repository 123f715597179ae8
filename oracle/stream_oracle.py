"""CPU restatement of the reference's streaming post-processing (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(kws_amd/stream.py, csrc/kws_stream.hip) never does.

Follows /root/reference/listen.py:
  * update_vectors            listen.py:96-114   (chunk -> window audio -> new MFCC rows -> sliding feature matrix)
  * ThresholdDecoder          listen.py:452-522  (logit-normal cumulative table, decode / encode)
  * TriggerDetector.update    listen.py:525-559  (activation counter with hysteresis and a refractory period)
and the loop around them, listen.py:350-375 / 403-428 (argmax, max, decode only non-background, detector update).

Pinned by tests/golden/stream_golden.npz, which tests/golden/make_golden_stream.py produced by executing the reference's
own two class definitions (tests/test_oracle_stream.py).  All arithmetic is float64, like the reference's Python floats.
"""
import math

import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# ThresholdDecoder (listen.py:467-522)
# ---------------------------------------------------------------------------------------------------------------------
def decoder_table(mu_stds, resolution=200, min_z=-4.0, max_z=4.0):
    """(min_out, out_range, cd): listen.py:467-472 and _calc_pd, listen.py:519-522.

    int() truncates toward zero (listen.py:468-469).  The table has resolution*out_range points spread over
    [min_out, max_out] with BOTH ends included (np.linspace default), each the mean over the (mu, std) components of the
    normal density divided by `resolution`; a component with std == 0 contributes nothing (listen.py:491-492).
    """
    mu_stds = [(float(m), float(s)) for m, s in mu_stds]
    min_out = int(min(m + min_z * s for m, s in mu_stds))
    max_out = int(max(m + max_z * s for m, s in mu_stds))
    out_range = max_out - min_out
    n = resolution * out_range
    if n == 0:
        # np.sum([0], axis=0) / (resolution * len) is the scalar 0.0 and its cumsum a one-element table (never read:
        # decode takes the out_range == 0 branch, listen.py:499-500)
        return min_out, out_range, np.zeros(1)
    pts = np.linspace(min_out, max_out, n)
    dens = np.zeros(n)
    for mu, std in mu_stds:
        if std != 0:
            dens = dens + (1.0 / (std * math.sqrt(2 * math.pi))) * np.exp(-(pts - mu) ** 2 / (2 * std ** 2))
    return min_out, out_range, np.cumsum(dens / (resolution * len(mu_stds)))


def decode(raw, min_out, out_range, cd, center, f32_input=False):
    """listen.py:496-508 for an array of raw network outputs.

    f32_input=True restates what the live loop does (listen.py:361-367): the score is the float32 array
    np.max(output, axis=-1), so `1 / x - 1` inside asigmoid is evaluated by numpy in float32 and only math.log widens it.
    """
    if f32_input:
        raw32 = np.asarray(raw, dtype=np.float32)
    raw = np.asarray(raw32 if f32_input else raw, dtype=np.float64)
    out = np.empty_like(raw)
    flat = raw.ravel()
    res = out.ravel()
    for i, x in enumerate(flat):
        if x == 1.0 or x == 0.0:
            res[i] = x
            continue
        if out_range == 0:
            cp = float(int(x > min_out))
        else:
            if f32_input:
                x32 = raw32.ravel()[i]
                odds = float(np.float32(1.0) / x32 - np.float32(1.0))
            else:
                odds = 1.0 / x - 1.0
            logit = -math.log(odds) if 0.0 < x < 1.0 else -10.0               # asigmoid, listen.py:480-485
            ratio = min(max((logit - min_out) / out_range, 0.0), 1.0)
            cp = cd[int(ratio * (len(cd) - 1) + 0.5)]
        res[i] = 0.5 * cp / center if cp < center else 0.5 + 0.5 * (cp - center) / (1.0 - center)
    return out


def encode(threshold, min_out, out_range, cd, center):
    """listen.py:510-517 (scalar)."""
    t = 0.5 * threshold / center
    cp = t * center * 2 if t < 0.5 else (t - 0.5) * 2 * (1 - center) + center
    ratio = np.searchsorted(cd, cp) / len(cd)
    return 1.0 / (1.0 + math.exp(-(min_out + out_range * ratio)))


# ---------------------------------------------------------------------------------------------------------------------
# TriggerDetector (listen.py:525-559)
# ---------------------------------------------------------------------------------------------------------------------
class TriggerState:
    """One stream's detector state: the activation counter and the previous prediction."""

    def __init__(self):
        self.activation = 0
        self.record_index = -1            # the reference starts with None, which never equals an index

    def update(self, index, score, is_background, sensitivity, trigger_level, chunk_size):
        """listen.py:538-559.  Returns True when this prediction fires."""
        if (not is_background) and index == self.record_index and score > sensitivity:
            self.activation += 1
            if self.activation > trigger_level:
                self.activation = -(8 * 2048) // chunk_size                  # refractory period, listen.py:549
                return True                                                   # record_index is left as it is (== index)
        elif self.activation < 0:
            self.activation += 1
        elif self.activation > 0:
            self.activation -= 1
        self.record_index = index
        return False


# ---------------------------------------------------------------------------------------------------------------------
# update_vectors (listen.py:96-114) and the prediction loop (listen.py:350-375)
# ---------------------------------------------------------------------------------------------------------------------
class StreamState:
    """Sliding feature matrix of one audio stream.  `featurize_raw(audio) -> (n_frames, n_mfcc)` is the framing MFCC of
    common/data_utils.py:61-70 (oracle/featurizer_oracle.py provides it)."""

    def __init__(self, n_features, n_mfcc, window_samples, hop_samples, featurize_raw):
        self.mfccs = np.zeros((n_features, n_mfcc))
        self.window_audio = np.zeros(0)
        self.window_samples, self.hop_samples = window_samples, hop_samples
        self.featurize_raw = featurize_raw

    def push(self, audio):
        """audio: float samples of one chunk (buffer_to_audio already applied).  Returns the (n_features, n_mfcc) matrix."""
        self.window_audio = np.concatenate((self.window_audio, np.asarray(audio, dtype=np.float64)))
        if len(self.window_audio) >= self.window_samples:
            new = self.featurize_raw(self.window_audio)
            self.window_audio = self.window_audio[len(new) * self.hop_samples:]
            if len(new) > len(self.mfccs):
                new = new[-len(self.mfccs):]
            self.mfccs = np.concatenate((self.mfccs[len(new):], new))
        return self.mfccs


def postprocess(probs, background_index, dec, center, trig, sensitivity, trigger_level, chunk_size):
    """One step of the loop listen.py:361-375 for one stream: probs (C,) -> (index, score, fired)."""
    index = int(np.argmax(probs))
    score = float(np.max(probs))
    if index != background_index:
        score = float(decode(np.array([score], dtype=np.float32), dec[0], dec[1], dec[2], center, f32_input=True)[0])
    fired = trig.update(index, score, index == background_index, sensitivity, trigger_level, chunk_size)
    return index, score, fired
