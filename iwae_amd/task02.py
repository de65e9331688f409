"""Reference API of tasks/task02.py: IWAEDReG (doubly reparameterised gradient estimator).
call() returns iwae_elbo / inference_loss (+ tensors), train_step has no objective argument
(tasks/task02.py:34-101)."""
import numpy as np

from .iwae1 import IWAE
from ._shim import as_tensor


class IWAEDReG(IWAE):
    scalar_keys = ("iwae_elbo", "inference_loss")      # tasks/task02.py:78-79

    def train_step(self, x, n_samples, beta, optimizer, outputs=None, eps=None):
        self._bind_optimizer(optimizer)
        raw = self._net.train_step(np.asarray(x, dtype=np.float32), int(n_samples), float(beta),
                                   float(optimizer.learning_rate), "dreg", eps=eps, want=self._want(outputs))
        optimizer.iterations += 1
        return self._result(raw)

    @staticmethod
    def write_to_tensorboard(res, step):          # tasks/task02.py:103-108
        out = {"step": int(step), "iwae_elbo": float(res["iwae_elbo"])}
        for k in ("lpxz", "lqzx", "lpz"):
            if k in res:
                out[k] = float(np.mean(res[k]))
        return out
