"""Dataset front-end (host mirror of the reference's classifier/data.py:15-120).

Layout and cache format are the reference's: <dataset>/sounds/<class>/*.wav is featurized once into
<dataset>/features/<class>/<uuid>.npy (float32, shape (n_features, feature_size, 1)); later runs load the cache.
The featurization itself is batched on the GPU (PCM16 wav decode with the stdlib, one kws_featurize launch per
`batch` files) instead of the reference's serial per-file loop (:39-44)."""
import glob
import os
import uuid
from shutil import rmtree

import numpy as np

from classifier.params import pr
from common.data_utils import get_featurizer, load_wav


def get_sample_list(audio_path, class_names):
    sample_list = []
    for class_name in class_names:
        class_path = os.path.join(audio_path, class_name)
        if not os.path.isdir(class_path):
            raise Exception('audio path for \'' + class_name + '\' not found at ' + class_path + '!')
        for audio_file in sorted(glob.glob(os.path.join(class_path, '*.wav'))):
            sample_list.append({'file': audio_file, 'word': class_name})
    return sample_list


def extract_features(audio_path, class_names, batch=1024):
    """audio files -> list of {'data': (n_features, feature_size, 1) float32, 'label': class name}"""
    import torch
    print('Extracting mfcc feature from audio files')
    sample_list = get_sample_list(audio_path, class_names)
    feat = get_featurizer()
    features = []
    for i in range(0, len(sample_list), batch):
        chunk = sample_list[i:i + batch]
        wav = np.zeros((len(chunk), pr.max_samples), np.float32)
        lens = np.zeros((len(chunk),), np.int32)
        for j, s in enumerate(chunk):
            a = load_wav(s['file'])[:pr.max_samples]          # keep the head (common/data_utils.py:77)
            wav[j, :len(a)] = a
            lens[j] = len(a)
        out = feat(torch.from_numpy(wav).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
        for j, s in enumerate(chunk):
            features.append({'data': out[j][..., None], 'label': s['word']})
    return features


def save_features(features, feature_path):
    if os.path.isdir(feature_path):
        rmtree(feature_path)
    os.makedirs(feature_path, exist_ok=True)
    print('Saving mfcc features as npy files to {}'.format(feature_path))
    for feature in features:
        class_path = os.path.join(feature_path, feature['label'])
        os.makedirs(class_path, exist_ok=True)
        np.save(os.path.join(class_path, uuid.uuid4().hex + '.npy'), feature['data'].astype(np.float32))


def split_data(x, y, val_split):
    """shuffled train/val split (the reference uses sklearn's train_test_split(shuffle=True), unseeded)"""
    x, y = np.asarray(x), np.asarray(y)
    n = len(x)
    n_val = int(math_ceil(n * val_split))
    perm = np.random.permutation(n)
    val, train = perm[:n_val], perm[n_val:]
    return x[train], y[train], x[val], y[val]


def math_ceil(v):
    return -int(-v // 1)


def get_dataset(dataset_path, class_names, val_split=None):
    """-> x_train, y_train, x_val, y_val   (x: (N, n_features, feature_size, 1) float32, y: (N,) int)"""
    audio_path = os.path.join(dataset_path, 'sounds')
    feature_path = os.path.join(dataset_path, 'features')

    if os.path.exists(feature_path):
        print('feature files path {} already exists, ignore feature extraction'.format(feature_path))
    else:
        features = extract_features(audio_path, class_names)
        save_features(features, feature_path)
        del features

    print('Loading mfcc features into memory')
    x, y = [], []
    for feature_file in sorted(glob.glob(os.path.join(feature_path, '*', '*.npy'))):
        feature_data = np.load(feature_file).astype(np.float32)
        _, class_name = os.path.split(os.path.dirname(feature_file))
        x.append(feature_data)
        y.append(class_names.index(class_name.lower()))

    if val_split:
        return split_data(x, y, val_split)
    return np.asarray(x), np.asarray(y), None, None
