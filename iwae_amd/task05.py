"""Reference API of tasks/task05.py on the MI355X-native step: the conditional IWAE (CIWAE) whose encoder sees
concat(x, onehot(y)) and whose decoder sees concat(z, onehot(y)), prior N(0, 1) (tasks/task05.py:101-198).

    model = CIWAE(200, 100)
    res = model.train_step(x, y, n_samples, beta, optimizer, objective="iwae_elbo")
    res = model.val_step(x, y, n_samples, beta)          # also model(x, y, n_samples)
    x_sample, probs = model.sample(z, label)              # tasks/task05.py:185-198

Same kernels as iwae1.IWAE with the first encoder layer 794 -> H and the first decoder layer 110 -> H; the C ABI carries the
condition separately (iwae_set_condition).  tasks/task04.py (learned conditional prior on top of this): iwae_amd/task04.py.
"""
import numpy as np

from ._shim import as_tensor
from .iwae1 import IWAE

N_CLASSES = 10      # tasks/task05.py:110: tf.one_hot(..., depth=10)


def one_hot(y, depth=N_CLASSES):
    y = np.asarray(y).astype(np.int64).ravel()
    out = np.zeros((y.size, depth), dtype=np.float32)
    out[np.arange(y.size), y] = 1.0
    return out


class CIWAE(IWAE):
    # tasks/task05.py:156-166: lpxzy / lpz / lqzxy instead of lpxz / lpz / lqzx
    _rename = {"lpxz": "lpxzy", "lqzx": "lqzxy"}

    def __init__(self, n_hidden, n_latent, **kwargs):
        super().__init__(n_hidden, n_latent, cond_dim=N_CLASSES, **kwargs)
        self.decoder = type(self.decoder)(self, 8, 14)

    def _result(self, raw):
        res = super()._result(raw)
        return {self._rename.get(k, k): v for k, v in res.items()}

    def call(self, x, y, n_samples, beta=1.0, outputs=None, eps=None):
        self._net.set_condition(one_hot(y))
        return super().call(x, n_samples, beta, outputs=outputs, eps=eps)

    __call__ = call

    def val_step(self, x, y, n_samples, beta, outputs=None):
        return self.call(x, y, n_samples, beta, outputs=outputs)

    def train_step(self, x, y, n_samples, beta, optimizer, objective="vae_elbo", outputs=None, eps=None):
        self._net.set_condition(one_hot(y))
        return super().train_step(x, n_samples, beta, optimizer, objective=objective, outputs=outputs, eps=eps)

    def eval_llh(self, x, y, n_samples=5000, chunk=0):
        self._net.set_condition(one_hot(y))
        return self._net.eval_llh(np.asarray(x, dtype=np.float32), n_samples, chunk)

    def sample(self, z, y):
        """tasks/task05.py:185-198: one label for all rows of z (tf.repeat(y, z.shape[0]))."""
        z = np.asarray(z, dtype=np.float32)
        self._net.set_condition(one_hot(np.repeat(int(y), z.shape[0])))
        probs = self._net.decode(z)
        x_sample = (np.random.random_sample(probs.shape) < probs).astype(np.float32)
        return as_tensor(x_sample), as_tensor(probs)

    def set_dataset(self, X_gray, y):
        """The labelled training set resident in HBM (tasks/task05.py:296-322 feeds (x, y) batches from tf.data): grey levels as uint8 and one
        class id per image; train_step_dataset then gathers, binarises and one-hot-encodes on the device (iwae_dataset_set_labels)."""
        y = np.asarray(y).astype(np.int64).ravel()
        if y.size != np.asarray(X_gray).shape[0]:
            raise ValueError("set_dataset: one label per image")
        if y.min() < 0 or y.max() >= N_CLASSES:
            raise ValueError("set_dataset: labels must be in [0, %d)" % N_CLASSES)
        super().set_dataset(X_gray)
        self._net.dataset_set_labels(y.astype(np.uint8))

    def train_step_dataset(self, start, batch_size, n_samples, beta, optimizer, objective="vae_elbo"):
        res = super().train_step_dataset(start, batch_size, n_samples, beta, optimizer, objective=objective)
        return {self._rename.get(k, k): v for k, v in res.items()}
