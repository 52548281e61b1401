"""Training-side (autograd) glue.  Filled in by the training milestone; the inference path does
not depend on it."""


def _todo(name):
    raise NotImplementedError(f"{name}: training path not built yet in this revision")


def selector_train(*a, **k):
    _todo("selector_train")


def temporal_train(*a, **k):
    _todo("temporal_train")


def text_features_train(*a, **k):
    _todo("text_features_train")


def anomaly_clip_train_forward(*a, **k):
    _todo("anomaly_clip_train_forward")
