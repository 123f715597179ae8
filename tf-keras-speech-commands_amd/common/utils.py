"""Miscellaneous utility functions (host mirror of the reference's common/utils.py)."""


def optimize_tf_gpu(tf=None, K=None):
    """TensorFlow memory-growth housekeeping in the reference (:10-29); nothing to do on this runtime."""
    return None


def get_classes(classes_path):
    '''loads the classes'''
    with open(classes_path) as f:
        class_names = f.readlines()
    class_names = [c.strip() for c in class_names]
    return class_names
