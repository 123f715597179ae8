"""Data-process utility functions (host mirror of the reference's common/data_utils.py).

Same names and argument meaning; the MFCC arithmetic that the reference delegates to
`sonopy.mfcc_spec` (reference :69) runs in the HIP featurizer behind include/kws.h.  There is no CPU
path: without a GPU these functions raise `kws_amd.KwsError`.
"""
import wave

import numpy as np

from classifier.params import pr
from kws_amd import lib as _l
from kws_amd.featurizer import Featurizer, params_key


class InvalidAudio(ValueError):
    """Raised for empty audio (the reference raises an undefined name here, reference :65-66)."""


_featurizers = {}


def get_featurizer(bank="mel"):
    """Featurizer for the CURRENT global params (re-created when inject_params changes them)."""
    key = (params_key(pr), bank)
    f = _featurizers.get(key)
    if f is None:
        f = _featurizers[key] = Featurizer(pr, bank)
    return f


def buffer_to_audio(buffer):
    """raw mono 16-bit little-endian bytes -> float32 in [-1, 1)"""
    assert pr.sample_depth == 2, 'only support 16-bit sample depth.'
    return np.frombuffer(buffer, dtype='<i2').astype(np.float32, order='C') / (np.iinfo(np.int16).max + 1)


def audio_to_buffer(audio):
    """float audio -> raw mono 16-bit little-endian bytes"""
    assert pr.sample_depth == 2, 'only support 16-bit sample depth.'
    return (np.asarray(audio) * (np.iinfo(np.int16).max + 1)).astype('<i2').tobytes()


def save_audio(filename, audio):
    """write float audio as a PCM16 wav with the configured sample rate (scale 32767 as the reference, :46)"""
    assert pr.sample_depth == 2, 'only support 16-bit sample depth.'
    data = (np.asarray(audio) * np.iinfo(np.int16).max).astype('<i2')
    w = wave.open(filename, 'wb')
    try:
        w.setnchannels(1)
        w.setsampwidth(pr.sample_depth)
        w.setframerate(pr.sample_rate)
        w.writeframes(data.tobytes())
    finally:
        w.close()


def add_deltas(features):
    """append the difference between adjacent timesteps on the last axis (first row: zeros)"""
    features = np.asarray(features)
    deltas = np.zeros_like(features)
    deltas[1:] = features[1:] - features[:-1]
    return np.concatenate([features, deltas], -1)


def _to_device_f32(audio):
    import torch
    a = np.ascontiguousarray(np.asarray(audio), dtype=np.float32)
    if a.ndim != 1:
        raise ValueError("expected 1-D audio")
    if not torch.cuda.is_available():
        raise _l.KwsError(-3, "no HIP device: the featurizer has no CPU fallback")
    return torch.from_numpy(a).cuda().unsqueeze(0)


def vectorize_raw(audio):
    """audio of any length -> (n_frames, n_mfcc) feature vectors, without clipping for length"""
    if len(audio) == 0:
        raise InvalidAudio('Cannot vectorize empty audio!')
    return get_featurizer().raw(_to_device_f32(audio))[0].cpu().numpy()


def audio_to_feature(audio_data):
    """audio -> (n_features, feature_size): keep the head, left-pad zeros, MFCC, optional deltas"""
    import torch
    audio_data = np.asarray(audio_data)[:pr.max_samples]
    f = get_featurizer()
    if len(audio_data) == 0:  # all padding
        wav = torch.zeros((1, 1), dtype=torch.float32, device="cuda")
        vl = torch.zeros((1,), dtype=torch.int32, device="cuda")
    else:
        wav = _to_device_f32(audio_data)
        vl = torch.full((1,), len(audio_data), dtype=torch.int32, device="cuda")
    return f(wav, vl)[0].cpu().numpy()


def load_wav(audio_path):
    """wav file -> float32 mono at pr.sample_rate: the contract of librosa.load(path, sr=pr.sample_rate, mono=True)
    (common/data_utils.py:93): channels averaged, 8 / 16 / 32-bit PCM scaled to [-1, 1), and a file at another rate RESAMPLED to
    pr.sample_rate.  librosa resamples with a band-limited sinc kernel (soxr_hq / kaiser_best depending on its version, unpinned in the
    reference); here it is scipy's polyphase FIR with a Kaiser window (resample_poly, beta 14, ~-90 dB stop band): the same length
    (ceil(n * sr / rate)) and the same samples to ~1e-3 of full scale inside the pass band, not bit-equal -- Speech Commands itself is
    16 kHz and never takes this branch."""
    w = wave.open(audio_path, 'rb')
    try:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    finally:
        w.close()
    if width == 2:
        audio = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif width == 1:
        audio = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 4:
        audio = (np.frombuffer(raw, dtype='<i4').astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError('%s: unsupported PCM sample width %d' % (audio_path, width))
    if nch > 1:
        audio = audio.reshape(-1, nch).mean(axis=1).astype(np.float32)
    if rate != pr.sample_rate and len(audio):
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(pr.sample_rate), int(rate))
        want = int(np.ceil(len(audio) * pr.sample_rate / float(rate)))
        audio = resample_poly(audio.astype(np.float64), pr.sample_rate // g, rate // g, window=('kaiser', 14.0))[:want].astype(np.float32)
    return audio


def get_mfcc_feature(audio_path):
    """audio file -> (n_features, feature_size, 1) feature"""
    return np.expand_dims(audio_to_feature(load_wav(audio_path)), axis=-1)
