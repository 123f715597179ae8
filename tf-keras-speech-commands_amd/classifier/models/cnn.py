"""Topology descriptors of the CNN backbones (what the reference builds with tf.keras layers in
classifier/models/cnn.py:11-141).  Pure host-side metadata: names, output shapes and parameter counts used by
`summary()`; the arithmetic is in csrc/ (kws_model.hip)."""


def _same(n, s):
    return -(-n // s)


def _cnn_layers(input_shape, feature_size, dropout_rate, separable):
    h, w, c = input_shape
    layers = []
    idx = {"conv": 0, "bn": 0, "relu": 0, "pool": 0}

    def name(kind, base):
        i = idx[kind]
        idx[kind] += 1
        return base if i == 0 else "%s_%d" % (base, i)

    cfg = [(16, 1, False, True), (32, 1, False, True), (64, 2, separable, False), (128, 1, True, True)]
    for filters, stride, relu, pool in cfg:
        h, w = _same(h, stride), _same(w, stride)
        if separable:
            params = 9 * c + c * filters + filters            # depthwise 3x3 (multiplier 1) + pointwise 1x1 + bias
            layers.append(dict(name=name("conv", "separable_conv2d"), type="SeparableConv2D", output_shape=(h, w, filters),
                               params=params, activation="relu" if relu else None, strides=stride))
        else:
            layers.append(dict(name=name("conv", "conv2d"), type="Conv2D", output_shape=(h, w, filters),
                               params=9 * c * filters, activation="relu" if relu else None, strides=stride))
        c = filters
        layers.append(dict(name=name("bn", "batch_normalization"), type="BatchNormalization", output_shape=(h, w, c),
                           params=4 * c, non_trainable=2 * c))
        layers.append(dict(name=name("relu", "re_lu"), type="ReLU", output_shape=(h, w, c), params=0))
        if pool:
            h, w = h // 2, w // 2
            layers.append(dict(name=name("pool", "max_pooling2d"), type="MaxPooling2D", output_shape=(h, w, c), params=0))
    flat = h * w * c
    layers.append(dict(name="flatten", type="Flatten", output_shape=(flat,), params=0))
    layers.append(dict(name="dropout", type="Dropout", output_shape=(flat,), params=0, rate=dropout_rate))
    layers.append(dict(name="dense", type="Dense", output_shape=(feature_size,), params=flat * feature_size + feature_size))
    layers.append(dict(name=name("relu", "re_lu"), type="ReLU", output_shape=(feature_size,), params=0))
    return layers


def SimpleCNN(input_shape=(30, 20, 1), feature_size=128, dropout_rate=0.5, **kwargs):
    """Conv3x3x16-BN-ReLU6-pool, Conv3x3x32-BN-ReLU6-pool, Conv3x3x64/2-BN-ReLU6, Conv3x3x128(relu)-BN-ReLU6-pool,
    Flatten, Dropout, Dense(feature_size), ReLU6"""
    return _cnn_layers(tuple(input_shape), feature_size, dropout_rate, separable=False)


def SimpleCNNLite(input_shape=(30, 20, 1), feature_size=128, dropout_rate=0.5, **kwargs):
    """same skeleton with SeparableConv2D(use_bias=True); relu on the 3rd and 4th separable convolutions"""
    return _cnn_layers(tuple(input_shape), feature_size, dropout_rate, separable=True)
