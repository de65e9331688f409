"""Reference API of src/iwae1.py on the MI355X-native step: IWAE(n_hidden, n_latent),
model(x, n_samples, beta), train_step, val_step, sample (src/iwae1.py:88-178)."""
import numpy as np

from ._shim import BaseIWAE, _Sub, as_tensor


class IWAE(BaseIWAE):
    n_layers = 1
    scalar_keys = ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14")   # src/iwae1.py:141-144

    def __init__(self, n_hidden, n_latent, **kwargs):
        super().__init__(int(n_hidden), int(n_latent), **kwargs)
        self.encoder = _Sub(self, 0, 8)     # l1, l2, lmu, lstd (kernel, bias)   src/iwae1.py:54
        self.decoder = _Sub(self, 8, 14)    # three Dense                        src/iwae1.py:70-77

    def sample(self, z):
        """src/iwae1.py:168-178: (x_sample ~ Bernoulli(probs), probs = sigmoid(decoder(z)))."""
        probs = self._net.decode(np.asarray(z, dtype=np.float32))
        x_sample = (np.random.random_sample(probs.shape) < probs).astype(np.float32)
        return as_tensor(x_sample), as_tensor(probs)
