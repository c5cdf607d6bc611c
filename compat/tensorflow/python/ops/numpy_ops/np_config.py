"""``np_config.enable_numpy_behavior()`` (reference train.py:15, analysis.py:23): the reference needs it so that
TF tensors accept ``.reshape`` / NumPy-style indexing; this package's DeviceArray has the ndarray protocol
already, so the call is a no-op."""


def enable_numpy_behavior(prefer_float32=False):
    return None
