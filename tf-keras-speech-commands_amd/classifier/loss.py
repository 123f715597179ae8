"""Loss objects (host mirror of the reference's classifier/loss.py).

`model.compile(loss=...)` inspects these objects and runs the matching fused HIP loss/gradient inside the train step;
calling them directly evaluates the per-sample losses on the GPU through kws_loss_forward (include/kws.h)."""
import numpy as np

from kws_amd import lib as _l


def _per_sample_losses(y_true, y_pred, class_weights, from_logits, ignore_index):
    import torch
    if not torch.cuda.is_available():
        raise _l.KwsError(-3, "no HIP device: losses have no CPU fallback")
    as_numpy = not isinstance(y_pred, torch.Tensor)
    p = torch.as_tensor(np.asarray(y_pred, np.float32) if as_numpy else y_pred, dtype=torch.float32).cuda().contiguous()
    y = torch.as_tensor(np.asarray(y_true) if not isinstance(y_true, torch.Tensor) else y_true)
    y = y.reshape(-1).to(torch.int32).cuda().contiguous()            # y_true[..., 0], reference :29,63
    B, C = p.shape
    if y.numel() != B:
        raise ValueError("y_true has %d labels for %d predictions" % (y.numel(), B))
    w = None
    if class_weights is not None:
        if len(class_weights) != C:
            raise ValueError("%d class weights for %d classes" % (len(class_weights), C))
        w = torch.as_tensor(np.asarray(class_weights, np.float32)).cuda()
    out = torch.empty((B,), dtype=torch.float32, device="cuda")
    _l.check(_l.get_lib().kws_loss_forward(p.data_ptr(), y.data_ptr(), w.data_ptr() if w is not None else None,
                                           int(bool(from_logits)), int(ignore_index or 0), B, C, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
    return out.cpu().numpy() if as_numpy else out


class SparseCategoricalCrossEntropy(object):
    """sparse categorical cross entropy on softmax probabilities, with (truthy) ignore_index support"""

    def __init__(self, ignore_index=None, from_logits=False):
        self.ignore_index = ignore_index
        self.from_logits = from_logits
        self.__name__ = 'sparse_categorical_crossentropy'

    def __call__(self, y_true, y_pred):
        return self.sparse_categorical_crossentropy(y_true, y_pred)

    def sparse_categorical_crossentropy(self, y_true, y_pred):
        return _per_sample_losses(y_true, y_pred, None, self.from_logits, self.ignore_index)


class WeightedSparseCategoricalCrossEntropy(object):
    """-log(p[label]) * weights[label] (no clipping), with (truthy) ignore_index support"""

    def __init__(self, weights, ignore_index=None, from_logits=False):
        self.weights = np.array(weights).astype('float32')
        self.ignore_index = ignore_index
        self.from_logits = from_logits
        self.__name__ = 'weighted_sparse_categorical_crossentropy'

    def __call__(self, y_true, y_pred):
        return self.weighted_sparse_categorical_crossentropy(y_true, y_pred)

    def weighted_sparse_categorical_crossentropy(self, y_true, y_pred):
        return _per_sample_losses(y_true, y_pred, self.weights, self.from_logits, self.ignore_index)
