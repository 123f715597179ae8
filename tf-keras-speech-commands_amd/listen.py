#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""Streaming keyword detection on the MI355X path: the `Listener` of the reference's listen.py with the same
constructor keywords and methods (`update_vectors`, `predict`, `run_wav`, `on_prediction`, `on_activation`), built on
kws_amd.stream.StreamBatch -- feature update, forward pass, score decoding and trigger logic all run on the device.

Differences from listen.py: checkpoints are the `.npz` files classifier.model writes (no h5/pb/tflite/onnx/mnn
back ends); there is no PyAudio in this image, so `run_microphone` raises and `run_wav` does not play the audio while it
analyses it; `Listener.batch(n)` gives the many-stream form for serving.
"""
import argparse
import os
import wave
from shutil import get_terminal_size

import numpy as np

from classifier.model import get_model
from classifier.params import inject_params, pr
from common.utils import get_classes
from kws_amd.stream import StreamBatch, ThresholdDecoder, TriggerDetector  # noqa: F401  (re-exported like listen.py:452,525)

default_config = {                                     # listen.py:31-40
    "model_path": '',
    "model_type": 'simple_cnn',
    "classes_path": os.path.join('configs', 'direction_classes.txt'),
    "params_path": None,
    "chunk_size": 1024,
    "sensitivity": 0.5,
    "trigger_level": 3,
    "save_dir": None,
    "input_wav": None,
}


class Listener(object):
    _defaults = default_config

    @classmethod
    def get_defaults(cls, n):
        if n in cls._defaults:
            return cls._defaults[n]
        return "Unrecognized attribute name '" + n + "'"

    def __init__(self, **kwargs):
        self.__dict__.update(self._defaults)
        self.__dict__.update(kwargs)
        self.pr = inject_params(self.params_path) if self.params_path else pr
        self.class_names = get_classes(self.classes_path)
        assert self.class_names[0] == 'background', '1st class should be background.'
        self.model = kwargs.get("model") or get_model(self.model_type, len(self.class_names), weights_path=self.model_path or None)
        self.threshold_decoder = ThresholdDecoder(self.pr.threshold_config, self.pr.threshold_center)
        self._sb = self.batch(1)
        self.detector = self._sb                       # detector state lives in the stream batch (device)
        self.activations = []

    def batch(self, n_streams):
        """A StreamBatch of n lock-stepped streams sharing this listener's model, decoder and settings."""
        return StreamBatch(self.pr, self.model._device(), n_streams, chunk_size=self.chunk_size, class_names=self.class_names,
                           sensitivity=self.sensitivity, trigger_level=self.trigger_level, decoder=self.threshold_decoder)

    def update_vectors(self, chunk):
        """listen.py:96-114: bytes of int16 PCM in, the (n_features, n_mfcc, 1) feature matrix out."""
        feats = self._sb.update_vectors([chunk])
        return np.expand_dims(feats[0].cpu().numpy(), axis=-1)

    def predict(self, data):
        return self.model.predict(np.asarray(data, dtype=np.float32))

    def step(self, chunk):
        """One iteration of the loop listen.py:350-375: (index, score, activated)."""
        index, score, fired = self._sb.push([chunk])
        return int(index[0]), float(score[0]), bool(fired[0])

    def on_prediction(self, index, score):
        width = min(get_terminal_size()[0], 80)
        class_name = self.class_names[index]
        if class_name == 'background':                 # show the inverted score and no label, listen.py:281-283
            score = 1.0 - score
            class_name = ''
        units = int(round(score * width))
        bar = 'X' * units + '-' * (width - units)
        cutoff = round(self.sensitivity * width)
        print(bar[:cutoff] + bar[cutoff:].replace('X', 'x') + class_name)

    def on_activation(self, index, play_activate=False):
        print('command {} detected!'.format(self.class_names[index]))
        self.activations.append(index)

    def run_microphone(self):
        raise RuntimeError("PyAudio is not available in this image; feed chunks with Listener.step() or use run_wav()")

    def run_wav(self, quiet=False):
        wf = wave.open(self.input_wav, 'rb')
        assert wf.getnchannels() == 1, 'input wav channels mismatch'
        assert wf.getframerate() == self.pr.sample_rate, 'input wav sample rate mismatch'
        assert wf.getsampwidth() == self.pr.sample_depth, 'input wav sample depth mismatch'
        assert wf.getnframes() > 0, 'no valid data in input wav'
        results = []
        chunk = wf.readframes(self.chunk_size)
        while len(chunk) > 0:
            index, score, fired = self.step(chunk)
            if not quiet:
                self.on_prediction(index, score)
            if fired:
                self.on_activation(index, play_activate=False)
            results.append((index, score, fired))
            chunk = wf.readframes(self.chunk_size)
        wf.close()
        return results

    def run(self):
        if self.input_wav:
            return self.run_wav()
        return self.run_microphone()


def main():
    parser = argparse.ArgumentParser(description='keyword detection on a wav file (MI355X path)')
    parser.add_argument('--model_path', type=str, required=True, help='.npz weights written by classifier.model')
    parser.add_argument('--model_type', type=str, default=default_config['model_type'])
    parser.add_argument('--classes_path', type=str, default=default_config['classes_path'])
    parser.add_argument('--params_path', type=str, default=None)
    parser.add_argument('--chunk_size', type=int, default=1024)
    parser.add_argument('--sensitivity', type=float, default=0.5)
    parser.add_argument('--trigger_level', type=int, default=3)
    parser.add_argument('--input_wav', type=str, required=True)
    args = parser.parse_args()
    Listener(**vars(args)).run()


if __name__ == '__main__':
    main()
