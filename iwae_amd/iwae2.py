"""Reference API of src/iwae2.py (two stochastic layers, src/iwae2.py:99-182)."""
from ._shim import BaseIWAE, _Sub


class IWAE(BaseIWAE):
    n_layers = 2
    scalar_keys = ("vae_elbo", "iwae_elbo", "iwae_eq14")    # src/iwae2.py:154-156 (no vae_elbo_kl: KeyError)

    def __init__(self, n_hidden, n_latent, **kwargs):
        super().__init__([int(v) for v in n_hidden], [int(v) for v in n_latent], **kwargs)
        self.encoder = _Sub(self, 0, 16)     # encode_x_to_z1, encode_z1_to_z2   src/iwae2.py:55-56
        self.decoder = _Sub(self, 16, 30)    # decode_z2_to_z1, decode_z1_to_x   src/iwae2.py:77-87

    def val_step(self, x, n_samples, beta, outputs=None):
        return self.call(x, n_samples, beta, outputs=outputs)
