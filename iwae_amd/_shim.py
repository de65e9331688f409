"""Shared machinery of the reference-API shims (iwae1.IWAE, iwae2.IWAE, task02.IWAEDReG)."""
import numpy as np

from .native import NativeModel


class Tensor(np.ndarray):
    """ndarray with the two TF-tensor habits the reference's callers rely on: .numpy() and
    '{:.2f}'.format(scalar_tensor) (main.py:161-162)."""

    def numpy(self):
        return np.asarray(self) if self.ndim else np.float32(self)

    def __format__(self, spec):
        if self.ndim == 0:
            return format(float(self), spec)
        return np.ndarray.__format__(self, spec)


def as_tensor(a):
    return np.asarray(a, dtype=np.float32).view(Tensor)


class _Sub:
    """model.encoder / model.decoder: expose .trainable_weights like the Keras sub-models."""

    def __init__(self, owner, lo, hi):
        self._owner, self._lo, self._hi = owner, lo, hi

    @property
    def trainable_weights(self):
        return self._owner.trainable_weights[self._lo:self._hi]


# cheap entries materialised by default; the big ones ([k,B,D] / [k,B,784]) only on request
_DEFAULT_1L = ("al", "lpxz", "lpz", "lqzx")
_ALL_1L = ("z", "snis_z", "al", "logits", "lpxz", "lpz", "lqzx")
_DEFAULT_2L = ("al", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2")
_ALL_2L = ("z", "z2", "snis_z", "snis_z2", "al", "logits", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2")
# C-ABI tensor name -> reference dict key (src/iwae2.py:154-167)
_RENAME_2L = {"z": "z1", "z2": "z2", "snis_z": "snis_z1", "snis_z2": "snis_z2", "al": "al", "logits": "logits",
              "lpxz": "lpxz1", "lpz": "lpz1z2", "lqzx": "lqz1x", "lpz2": "lpz2", "lqzx2": "lqz2z1"}


class BaseIWAE:
    n_layers = 1
    scalar_keys = ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14")

    def __init__(self, n_hidden, n_latent, x_dim=784, seed=123, device=0, output_bias=None, cond_dim=0, cond_prior=False,
                 precision="bf16", world_size=1, rank=0, **kwargs):
        self._net = NativeModel(self.n_layers, n_hidden, n_latent, x_dim=x_dim, device=device, seed=seed, cond_dim=cond_dim,
                                cond_prior=cond_prior, precision=precision, world_size=world_size, rank=rank)
        self._adam_hyper = (0.9, 0.999, 1e-4)      # the device optimizer's defaults (main.py:93)
        if output_bias is not None:
            self._net.set_output_bias(output_bias)
        self._table = self._net.tensor_table()
        self.outputs = None          # None: cheap default set; "all": everything the reference dict holds

    # ---- weights ----------------------------------------------------------------------
    @property
    def trainable_weights(self):
        flat = self._net.get_params()
        return [as_tensor(flat[off:off + int(np.prod(shape))].reshape(shape)) for _, shape, off in self._table]

    def get_weights(self):
        return [np.asarray(w) for w in self.trainable_weights]

    def set_weights(self, weights):
        flat = self._net.get_params()
        for (_, shape, off), w in zip(self._table, weights):
            w = np.asarray(w, dtype=np.float32)
            if w.shape != tuple(shape):
                raise ValueError("set_weights: shape mismatch %s vs %s" % (w.shape, shape))
            flat[off:off + w.size] = w.ravel()
        self._net.set_params(flat)

    def save_weights(self, path):
        """main.py:165.  Flat .npz (params in Keras trainable_weights order + Adam state), not a TF checkpoint."""
        m, v, t = self._net.get_adam_state()
        np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", params=self._net.get_params(),
                 adam_m=m, adam_v=v, adam_t=np.int64(t), names=np.array([n for n, _, _ in self._table]))

    def load_weights(self, path):
        p = path if str(path).endswith(".npz") else str(path) + ".npz"
        with np.load(p) as f:
            self._net.set_params(f["params"])
            if "adam_m" in f:
                self._net.set_adam_state(f["adam_m"], f["adam_v"], int(f["adam_t"]))

    # ---- result dict ------------------------------------------------------------------
    def _want(self, outputs):
        outputs = self.outputs if outputs is None else outputs
        two = self.n_layers == 2
        if outputs is None:
            return _DEFAULT_2L if two else _DEFAULT_1L
        if outputs == "all":
            return _ALL_2L if two else _ALL_1L
        if two:
            back = {v: k for k, v in _RENAME_2L.items()}
            return tuple(back.get(o, o) for o in outputs)
        return tuple(outputs)

    def _result(self, raw):
        res = {}
        for k in self.scalar_keys:
            res[k] = as_tensor(raw[k])
        for k, v in raw.items():
            if isinstance(v, np.ndarray):
                res[_RENAME_2L[k] if self.n_layers == 2 else k] = as_tensor(v)
        return res

    def call(self, x, n_samples, beta=1.0, outputs=None, eps=None):
        raw = self._net.forward(np.asarray(x, dtype=np.float32), int(n_samples), float(beta), eps=eps, want=self._want(outputs))
        return self._result(raw)

    __call__ = call

    def val_step(self, x, n_samples, beta, outputs=None):
        return self.call(x, n_samples, beta, outputs=outputs)

    def _bind_optimizer(self, optimizer):
        """Hand the Keras-style optimizer's (beta_1, beta_2, epsilon) to the device optimizer when they change."""
        hyper = optimizer.hyper() if hasattr(optimizer, "hyper") else self._adam_hyper
        if hyper != self._adam_hyper:
            self._net.set_adam(*hyper)
            self._adam_hyper = hyper

    def train_step(self, x, n_samples, beta, optimizer, objective="vae_elbo", outputs=None, eps=None):
        if objective not in self.scalar_keys:
            raise KeyError(objective)          # res[objective] in the reference (src/iwae1.py:157)
        self._bind_optimizer(optimizer)
        raw = self._net.train_step(np.asarray(x, dtype=np.float32), int(n_samples), float(beta),
                                   float(optimizer.learning_rate), objective, eps=eps, want=self._want(outputs))
        optimizer.iterations += 1
        return self._result(raw)

    # ---- device-resident data pipeline (main.py:59-65,117-120 on the GPU) -----------------------
    def set_dataset(self, X_gray):
        """Keep the grey-level training set in HBM (uint8; float data in [0,1] is quantised to 1/255 steps)."""
        X_gray = np.asarray(X_gray)
        if X_gray.dtype != np.uint8:
            X_gray = np.clip(np.rint(X_gray.reshape(X_gray.shape[0], -1) * 255.0), 0, 255).astype(np.uint8)
        self._net.dataset_upload(X_gray.reshape(X_gray.shape[0], -1))

    def begin_epoch(self, epoch, order=None):
        """New dynamic binarisation (keyed by the epoch) and visiting order (tf.data shuffle) for this epoch."""
        self._net.dataset_begin_epoch(epoch, order)

    def train_step_dataset(self, start, batch_size, n_samples, beta, optimizer, objective="vae_elbo"):
        if objective not in self.scalar_keys and objective != "dreg":
            raise KeyError(objective)
        self._bind_optimizer(optimizer)
        raw = self._net.train_step_dataset(start, batch_size, int(n_samples), float(beta), float(optimizer.learning_rate), objective)
        optimizer.iterations += 1
        res = {k: as_tensor(raw[k]) for k in self.scalar_keys if k in raw}
        # the means write_to_tensorboard logs (src/iwae1.py:228-232), already reduced on the device
        names = ("lpxz1", "lpz1z2", "lpz2") if self.n_layers == 2 else ("lpxz", "lpz", "lqzx")
        for key, src in zip(names, ("mean_lpxz", "mean_lpz", "mean_lqzx")):
            res[key] = as_tensor(raw[src])
        return res

    def eval_llh(self, x, n_samples=5000, chunk=0):
        """The test-set loop of main.py:170-184 in one call (mean of per-image iwae_elbo at B=1)."""
        return self._net.eval_llh(np.asarray(x, dtype=np.float32), n_samples, chunk)

    @staticmethod
    def write_to_tensorboard(res, step):
        """src/iwae1.py:226-232 logs five scalars through tf.summary; TensorBoard is not in this image,
        so return them (main.py writes them to a CSV)."""
        out = {"step": int(step)}
        for k in ("vae_elbo", "iwae_elbo"):
            if k in res:
                out[k] = float(res[k])
        for k in ("lpxz", "lqzx", "lpz", "lpxz1", "lpz1z2", "lqz1x", "lqz2z1", "lpz2", "lpxzy", "lqzxy", "lpzy"):      # (the last three: tasks/task05.py:248-253, task04.py)
            if k in res:
                out[k] = float(np.mean(res[k]))
        return out
