"""Audio-pipeline parameters (host mirror of the reference's classifier/params.py).

Same surface: the `ListenerParams` fields and derived properties (reference :49-91), the module
global `pr` with the same defaults (:99-103), `inject_params` (:107-115) and `save_params`
(:118-121).  `pr` stays frozen for attribute assignment and, as in the reference, is updated in
place through its instance dict by `inject_params`.
"""
import json
import os
from math import floor

_FIELDS = ("buffer_t", "window_t", "hop_t", "sample_rate", "sample_depth", "n_fft", "n_filt", "n_mfcc",
           "use_delta", "threshold_config", "threshold_center")


class ListenerParams(object):
    """
    buffer_t: input size of audio (s) -- window_t / hop_t: frame length / advance (s)
    sample_rate, sample_depth (bytes) -- n_fft, n_filt, n_mfcc -- use_delta
    threshold_config, threshold_center: output-distribution settings of the streaming decoder
    """

    def __init__(self, buffer_t, window_t, hop_t, sample_rate, sample_depth, n_fft, n_filt, n_mfcc, use_delta,
                 threshold_config, threshold_center):
        values = (buffer_t, window_t, hop_t, sample_rate, sample_depth, n_fft, n_filt, n_mfcc, use_delta,
                  threshold_config, threshold_center)
        self.__dict__.update(dict(zip(_FIELDS, values)))

    def __setattr__(self, name, value):
        raise AttributeError("ListenerParams is frozen; use inject_params() to load a params.json")

    __delattr__ = __setattr__

    def __repr__(self):
        return "ListenerParams(%s)" % ", ".join("%s=%r" % (k, self.__dict__[k]) for k in _FIELDS if k in self.__dict__)

    def __eq__(self, other):
        return isinstance(other, ListenerParams) and self.__dict__ == other.__dict__

    @property
    def buffer_samples(self):
        """buffer_t converted to samples, truncating partial frames"""
        samples = int(self.sample_rate * self.buffer_t + 0.5)
        return self.hop_samples * (samples // self.hop_samples)

    @property
    def n_features(self):
        """number of timesteps in one network input"""
        return 1 + int(floor((self.buffer_samples - self.window_samples) / self.hop_samples))

    @property
    def window_samples(self):
        return int(self.sample_rate * self.window_t + 0.5)

    @property
    def hop_samples(self):
        return int(self.sample_rate * self.hop_t + 0.5)

    @property
    def max_samples(self):
        return int(self.buffer_t * self.sample_rate)

    @property
    def feature_size(self):
        return self.n_mfcc * 2 if self.use_delta else self.n_mfcc


# global listener parameters, reference defaults
pr = ListenerParams(buffer_t=1.0, window_t=0.064, hop_t=0.032, sample_rate=16000, sample_depth=2, n_fft=1024,
                    n_filt=20, n_mfcc=20, use_delta=False, threshold_config=((6, 4),), threshold_center=0.2)


def inject_params(params_file):
    """Overlay a saved params.json on the global `pr` (never raises for a missing or bad file)."""
    try:
        with open(params_file) as f:
            pr.__dict__.update(**json.load(f))
    except (OSError, ValueError, TypeError):
        if os.path.isfile(params_file):
            print('Warning: Failed to load parameters from ' + params_file)
    return pr


def save_params(params_file):
    """Write the current global params as JSON."""
    with open(params_file, 'w') as f:
        json.dump(pr.__dict__, f, indent=2)
