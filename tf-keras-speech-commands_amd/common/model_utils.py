"""Optimizer / learning-rate schedule factory (host mirror of the reference's common/model_utils.py:17-58).

The returned objects are small descriptors; the arithmetic runs in the fused HIP optimizer kernels
(kws_adam_step / kws_rmsprop_step / kws_sgd_step).  The schedules restate the tf.keras ones the reference picks:
CosineDecay(alpha=0.2), ExponentialDecay(decay_rate=0.9), PolynomialDecay(end = lr/100, power 1),
PiecewiseConstantDecay([500, 0.9*S, S] -> [1e-3, lr, lr/10, lr/100])."""
import math


class LearningRateSchedule(object):
    def __call__(self, step):
        raise NotImplementedError


class CosineDecay(LearningRateSchedule):
    def __init__(self, initial_learning_rate, decay_steps, alpha=0.0):
        self.initial_learning_rate, self.decay_steps, self.alpha = initial_learning_rate, decay_steps, alpha

    def __call__(self, step):
        s = min(step, self.decay_steps)
        cosine = 0.5 * (1.0 + math.cos(math.pi * s / self.decay_steps))
        return self.initial_learning_rate * ((1 - self.alpha) * cosine + self.alpha)


class ExponentialDecay(LearningRateSchedule):
    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.initial_learning_rate, self.decay_steps, self.decay_rate, self.staircase = \
            initial_learning_rate, decay_steps, decay_rate, staircase

    def __call__(self, step):
        p = step / self.decay_steps
        if self.staircase:
            p = math.floor(p)
        return self.initial_learning_rate * self.decay_rate ** p


class PolynomialDecay(LearningRateSchedule):
    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0):
        self.initial_learning_rate, self.decay_steps, self.end_learning_rate, self.power = \
            initial_learning_rate, decay_steps, end_learning_rate, power

    def __call__(self, step):
        s = min(step, self.decay_steps)
        return (self.initial_learning_rate - self.end_learning_rate) * (1 - s / self.decay_steps) ** self.power + \
            self.end_learning_rate


class PiecewiseConstantDecay(LearningRateSchedule):
    def __init__(self, boundaries, values):
        if len(boundaries) != len(values) - 1:
            raise ValueError("The length of boundaries should be 1 less than the length of values")
        self.boundaries, self.values = list(boundaries), list(values)

    def __call__(self, step):
        for b, v in zip(self.boundaries, self.values):
            if step <= b:
                return v
        return self.values[-1]


def get_lr_scheduler(learning_rate, decay_type, decay_steps):
    if decay_type:
        decay_type = decay_type.lower()

    if decay_type is None:
        lr_scheduler = learning_rate
    elif decay_type == 'cosine':
        lr_scheduler = CosineDecay(initial_learning_rate=learning_rate, decay_steps=decay_steps, alpha=0.2)
    elif decay_type == 'exponential':
        lr_scheduler = ExponentialDecay(initial_learning_rate=learning_rate, decay_steps=decay_steps, decay_rate=0.9)
    elif decay_type == 'polynomial':
        lr_scheduler = PolynomialDecay(initial_learning_rate=learning_rate, decay_steps=decay_steps,
                                       end_learning_rate=learning_rate / 100)
    elif decay_type == 'piecewise_constant':
        boundaries = [500, int(decay_steps * 0.9), decay_steps]
        values = [0.001, learning_rate, learning_rate / 10., learning_rate / 100.]
        lr_scheduler = PiecewiseConstantDecay(boundaries=boundaries, values=values)
    else:
        raise ValueError('Unsupported lr decay type')

    return lr_scheduler


class Optimizer(object):
    """Descriptor of a Keras optimizer; `kind` selects the HIP update kernel."""
    kind = None

    def __init__(self, learning_rate):
        self.learning_rate = learning_rate
        self.iterations = 0

    def current_lr(self):
        """learning rate for the NEXT update (schedules are evaluated at the number of updates done so far)"""
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    @property
    def lr(self):
        return self.current_lr()

    def set_lr(self, value):
        if callable(self.learning_rate):
            raise TypeError("the learning rate is a schedule and cannot be set")
        self.learning_rate = float(value)


class Adam(Optimizer):
    kind = 'adam'

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, clipnorm=None,
                 clipvalue=None):
        if amsgrad or clipnorm or clipvalue:
            raise NotImplementedError("amsgrad / clipping are not used by the reference and have no kernel")
        Optimizer.__init__(self, learning_rate)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon


class RMSprop(Optimizer):
    kind = 'rmsprop'

    def __init__(self, learning_rate=0.001, rho=0.9, momentum=0.0, epsilon=1e-7, centered=False, clipnorm=None,
                 clipvalue=None):
        if momentum or centered or clipnorm or clipvalue:
            raise NotImplementedError("momentum / centered / clipping are not used by the reference and have no kernel")
        Optimizer.__init__(self, learning_rate)
        self.rho, self.epsilon = rho, epsilon


class SGD(Optimizer):
    kind = 'sgd'

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, clipnorm=None, clipvalue=None):
        if momentum or nesterov or clipnorm or clipvalue:
            raise NotImplementedError("momentum / nesterov / clipping are not used by the reference and have no kernel")
        Optimizer.__init__(self, learning_rate)


def get_optimizer(optim_type, learning_rate, average_type=None, decay_type='cosine', decay_steps=100000):
    optim_type = optim_type.lower()

    lr_scheduler = get_lr_scheduler(learning_rate, decay_type, decay_steps)

    if optim_type == 'adam':
        optimizer = Adam(learning_rate=lr_scheduler, amsgrad=False, clipnorm=None, clipvalue=None)
    elif optim_type == 'rmsprop':
        optimizer = RMSprop(learning_rate=lr_scheduler, rho=0.9, momentum=0.0, centered=False, clipnorm=None, clipvalue=None)
    elif optim_type == 'sgd':
        optimizer = SGD(learning_rate=lr_scheduler, momentum=0.0, nesterov=False, clipnorm=None, clipvalue=None)
    else:
        raise ValueError('Unsupported optimizer type')

    if average_type:
        # the reference wraps with tensorflow-addons (MovingAverage / SWA / Lookahead), which is out of scope here
        raise ValueError('Unsupported average type')

    return optimizer
