#!/usr/bin/env python3
"""Evaluate a trained keyword-spotting model on a dataset (mirror of the reference's eval.py for the native runtime).

The reference predicts one sample at a time through five foreign runtimes and builds the confusion matrix with sklearn
(eval.py:201-256).  Here the whole set is scored in GPU batches and the confusion counts are accumulated on the device
(kws_confusion_counts); the printed summary is the reference's: accuracy, then the matrix."""
import argparse
import os
import sys

import numpy as np

sys.path.append(os.path.dirname(os.path.realpath(__file__)))
from classifier.data import get_dataset
from classifier.model import get_model
from classifier.params import inject_params
from common.utils import get_classes
from kws_amd import lib as _l


def evaluate_accuracy(model, x, y, class_names, batch_size=4096):
    """-> (top-1 accuracy, confusion matrix [label, prediction] as int64 numpy)"""
    import torch
    dm = model._device()
    xd, is_audio = model._to_device_inputs(x)
    yd = model._labels(y, xd.shape[0])
    C = len(class_names)
    counts = torch.zeros((C, C), dtype=torch.int32, device=xd.device)
    L = _l.get_lib()
    for i in range(0, xd.shape[0], batch_size):
        _, am = dm.forward(model._features_of(xd[i:i + batch_size], is_audio).contiguous(), False, True)
        yb = yd[i:i + batch_size].contiguous()
        _l.check(L.kws_confusion_counts(yb.data_ptr(), am.data_ptr(), yb.numel(), C, counts.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream))
    cm = counts.cpu().numpy().astype(np.int64)
    total = int(cm.sum())
    return (float(np.trace(cm)) / total if total else 0.0), cm


def print_confusion_matrix(cm, class_names):
    w = max(8, max(len(c) for c in class_names) + 1)
    print(' ' * w + ''.join('%*s' % (w, c[:w - 1]) for c in class_names))
    for i, c in enumerate(class_names):
        print('%*s' % (w, c[:w - 1]) + ''.join('%*d' % (w, v) for v in cm[i]))


def main():
    parser = argparse.ArgumentParser(description='evaluate a trained model (.npz weights) on a dataset')
    parser.add_argument('--model_type', type=str, default='simple_cnn')
    parser.add_argument('--weights_path', type=str, required=True)
    parser.add_argument('--dataset_path', type=str, required=True)
    parser.add_argument('--classes_path', type=str, required=True)
    parser.add_argument('--params_path', type=str, default=None)
    parser.add_argument('--batch_size', type=int, default=4096)
    args = parser.parse_args()
    class_names = get_classes(args.classes_path)
    assert class_names[0] == 'background', '1st class should be background.'
    if args.params_path:
        inject_params(args.params_path)
    x, y, _, _ = get_dataset(args.dataset_path, class_names)
    model = get_model(args.model_type, len(class_names), weights_path=args.weights_path)
    acc, cm = evaluate_accuracy(model, x, y, class_names, args.batch_size)
    print('%d correct out of %d samples, accuracy %.4f' % (int(np.trace(cm)), int(cm.sum()), acc))
    print_confusion_matrix(cm, class_names)


if __name__ == '__main__':
    main()
