"""Training callbacks (the subset train.py of the reference wires up, train.py:30-43, plus the reference's own
CheckpointCleanCallBack, common/callbacks.py:9-21).  They follow the Keras callback protocol on the host."""
import glob
import json
import math
import os


class Callback(object):
    model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


def _better(mode, cur, best, min_delta=0.0):
    return cur > best + min_delta if mode == 'max' else cur < best - min_delta


class ModelCheckpoint(Callback):
    """save the model when `monitor` improves (save_best_only) or every epoch; filepath may use {epoch:03d} and log keys"""

    def __init__(self, filepath, monitor='val_loss', mode='min', verbose=0, save_best_only=False, save_weights_only=False,
                 period=1):
        self.filepath, self.monitor, self.mode, self.verbose = filepath, monitor, mode, verbose
        self.save_best_only, self.period = save_best_only, period
        self.best = -math.inf if mode == 'max' else math.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if (epoch + 1) % self.period:
            return
        cur = logs.get(self.monitor)
        if self.save_best_only:
            if cur is None or not _better(self.mode, cur, self.best):
                return
            self.best = cur
        path = self.filepath.format(epoch=epoch + 1, **logs)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.model.save(path)
        if self.verbose:
            print('Epoch %05d: saving model to %s' % (epoch + 1, path))


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='min', min_delta=1e-4, cooldown=0, min_lr=0):
        self.monitor, self.factor, self.patience, self.verbose, self.mode = monitor, factor, patience, verbose, mode
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.best = -math.inf if mode == 'max' else math.inf
        self.wait = self.cooldown_counter = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if _better(self.mode, cur, self.best, self.min_delta):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = self.model.optimizer.current_lr()
                if old > self.min_lr:
                    new = max(old * self.factor, self.min_lr)
                    self.model.optimizer.set_lr(new)
                    if self.verbose:
                        print('Epoch %05d: ReduceLROnPlateau reducing learning rate to %g.' % (epoch + 1, new))
                    self.cooldown_counter, self.wait = self.cooldown, 0


class EarlyStopping(Callback):
    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='min'):
        self.monitor, self.min_delta, self.patience, self.verbose, self.mode = monitor, min_delta, patience, verbose, mode
        self.best = -math.inf if mode == 'max' else math.inf
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if _better(self.mode, cur, self.best, self.min_delta):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.model.stop_training = True
                if self.verbose:
                    print('Epoch %05d: early stopping' % (epoch + 1))


class TerminateOnNaN(Callback):
    """checked once per epoch (the loss is accumulated on the device; no per-batch host sync)"""

    def on_epoch_end(self, epoch, logs=None):
        loss = (logs or {}).get('loss')
        if loss is not None and (math.isnan(loss) or math.isinf(loss)):
            print('Epoch %d: Invalid loss, terminating training' % (epoch + 1))
            self.model.stop_training = True


class CheckpointCleanCallBack(Callback):
    """keep only the newest `max_keep` checkpoints in `checkpoint_dir`"""

    def __init__(self, checkpoint_dir, max_keep=5, pattern='ep*'):
        self.checkpoint_dir, self.max_keep, self.pattern = checkpoint_dir, max_keep, pattern

    def on_epoch_end(self, epoch, logs=None):
        files = sorted(glob.glob(os.path.join(self.checkpoint_dir, self.pattern)), key=os.path.getmtime)
        for f in files[:-self.max_keep] if self.max_keep > 0 else files:
            os.remove(f)


class JsonlLogger(Callback):
    """one JSON line per epoch: loss, accuracy, val_*, clips/s (stands in for the TensorBoard scalars of train.py:30)"""

    def __init__(self, path):
        self.path = path

    def on_epoch_end(self, epoch, logs=None):
        with open(self.path, 'a') as f:
            f.write(json.dumps(dict(epoch=epoch + 1, **(logs or {}))) + '\n')
