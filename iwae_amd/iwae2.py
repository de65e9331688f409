"""Reference API of src/iwae2.py (two stochastic layers, src/iwae2.py:99-182)."""
import numpy as np

from ._shim import BaseIWAE, _Sub, as_tensor


class IWAE(BaseIWAE):
    n_layers = 2
    scalar_keys = ("vae_elbo", "iwae_elbo", "iwae_eq14")    # src/iwae2.py:154-156 (no vae_elbo_kl: KeyError)

    def __init__(self, n_hidden, n_latent, **kwargs):
        super().__init__([int(v) for v in n_hidden], [int(v) for v in n_latent], **kwargs)
        self.encoder = _Sub(self, 0, 16)     # encode_x_to_z1, encode_z1_to_z2   src/iwae2.py:55-56
        self.decoder = _Sub(self, 16, 30)    # decode_z2_to_z1, decode_z1_to_x   src/iwae2.py:77-87

    def val_step(self, x, n_samples, beta, outputs=None):
        return self.call(x, n_samples, beta, outputs=outputs)

    def sample(self, z2):
        """src/iwae2.py:184-196: z1 ~ p(z1|z2), probs = sigmoid(decoder(z1)), x_sample ~ Bernoulli(probs)."""
        probs = self._net.decode(np.asarray(z2, dtype=np.float32))
        x_sample = (np.random.random_sample(probs.shape) < probs).astype(np.float32)
        return as_tensor(x_sample), as_tensor(probs)
