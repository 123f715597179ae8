"""Topology descriptors of the recurrent backbones (reference: classifier/models/rnn.py:10-79)."""


def SimpleGRU(input_shape=(30, 20), recurrent_units=48, num_layers=1, dropout_rate=0.2, **kwargs):
    """num_layers x GRU(units, activation='linear', dropout=rate); Keras v2 GRU: reset_after=True, bias (2, 3u)"""
    t, c = input_shape
    layers = []
    for i in range(num_layers):
        last = i == num_layers - 1
        u = recurrent_units
        layers.append(dict(name="gru_unit_%d" % i, type="GRU", output_shape=(u,) if last else (t, u),
                           params=3 * u * c + 3 * u * u + 2 * 3 * u, activation="linear", dropout=dropout_rate))
        c = u
    return layers


def SimpleLSTM(input_shape=(30, 20), recurrent_units=48, num_layers=1, dropout_rate=0.2, **kwargs):
    """num_layers x LSTM(units, activation='tanh', dropout=rate)"""
    t, c = input_shape
    layers = []
    for i in range(num_layers):
        last = i == num_layers - 1
        u = recurrent_units
        layers.append(dict(name="lstm_unit_%d" % i, type="LSTM", output_shape=(u,) if last else (t, u),
                           params=4 * u * c + 4 * u * u + 4 * u, activation="tanh", dropout=dropout_rate))
        c = u
    return layers
