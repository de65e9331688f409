"""Host-side helpers with the reference's names (src/utils.py).  NumPy only; the GPU path never
needs them, they exist so callers written against the reference keep working."""
import os
import numpy as np


def logmeanexp(log_w, axis):
    """src/utils.py:6-8."""
    log_w = np.asarray(log_w)
    m = np.max(log_w, axis=axis)
    return np.log(np.mean(np.exp(log_w - np.expand_dims(m, axis)), axis=axis)) + m


def bernoullisample(x):
    """src/utils.py:26-27 (dynamic binarisation)."""
    return np.random.binomial(1, x, size=x.shape).astype('float32')


class MyMetric():
    """src/utils.py:30-45: list-append mean."""

    def __init__(self):
        self.VALUES = []
        self.N = []

    def update_state(self, losses):
        losses = np.asarray(losses, dtype=np.float32)
        self.VALUES.append(losses.reshape(losses.shape[0], -1) if losses.ndim else losses.reshape(1, 1))
        self.N.append(self.VALUES[-1].shape[0])

    def result(self):
        return np.float32(np.sum(np.concatenate(self.VALUES, axis=0)) / np.float32(np.sum(self.N)))

    def reset_states(self):
        self.VALUES = []
        self.N = []


_MNIST_CANDIDATES = ("IWAE_MNIST_PATH", "~/.keras/datasets/mnist.npz", "./mnist.npz", "./data/mnist.npz")


def find_mnist():
    for c in _MNIST_CANDIDATES:
        p = os.environ.get(c) if c.isupper() else os.path.expanduser(c)
        if p and os.path.exists(p):
            return p
    return None


def load_mnist(path=None):
    """keras.datasets.mnist.load_data() replacement for an offline machine (main.py:59): reads the
    same mnist.npz file Keras caches.  Returns ((Xtrain, ytrain), (Xtest, ytest)) uint8 / None."""
    path = path or find_mnist()
    if path is None:
        return None
    with np.load(path) as f:
        return (f["x_train"], f["y_train"]), (f["x_test"], f["y_test"])


def synthetic_pixel_means(x_dim=784):
    """MNIST-like per-pixel Bernoulli means: smooth centred blob, global mean ~0.13."""
    side = int(round(np.sqrt(x_dim)))
    yy, xx = np.mgrid[0:side, 0:side]
    c = (side - 1) / 2.0
    r2 = ((yy - c) ** 2 + (xx - c) ** 2) / (0.30 * side) ** 2
    return (0.62 * np.exp(-r2)).reshape(-1)[:x_dim]


def synthetic_mnist(n_train=60000, n_test=10000, seed=123, x_dim=784):
    """Grey-level stand-in for MNIST when the real file is absent: per-image intensity-modulated
    blobs in [0,1] (so that dynamic binarisation still has something to sample)."""
    rng = np.random.default_rng(seed)
    p = synthetic_pixel_means(x_dim)[None]

    def make(n):
        scale = rng.uniform(0.5, 1.5, size=(n, 1))
        return np.clip(p * scale, 0.0, 1.0).astype(np.float64)

    return make(n_train), make(n_test)


def bias_from_mean(train_mean):
    """src/utils.py:19-21."""
    return (-np.log(1. / np.clip(np.asarray(train_mean, dtype=np.float64), 0.001, 0.999) - 1.)).astype(np.float32)


def get_bias(Xtrain=None):
    """src/utils.py:11-23: logit of the clipped per-pixel training mean.  The reference downloads
    MNIST here; offline we take the training matrix from the caller or a local mnist.npz."""
    if Xtrain is None:
        data = load_mnist()
        if data is None:
            raise FileNotFoundError("get_bias(): no local mnist.npz (set IWAE_MNIST_PATH) and no Xtrain given")
        Xtrain = data[0][0]
    Xtrain = np.asarray(Xtrain)
    Xtrain = Xtrain.reshape(Xtrain.shape[0], -1)
    if Xtrain.dtype == np.uint8:
        Xtrain = Xtrain / 255
    return bias_from_mean(np.mean(Xtrain, axis=0))
