"""Reference API of tasks/task04.py on the MI355X-native step: the conditional IWAE with a LEARNED conditional prior --
encoder on concat(x, onehot(y)), decoder on concat(z, onehot(y)) and p(z|y) = N(mu_p(y), sigma_p(y)) from a BasicBlock on
onehot(y) (tasks/task04.py:101-173); sample(z, y) maps z through that prior first (:190-204).

    model = CIWAE(200, 100)
    res = model.train_step(x, y, n_samples, beta, optimizer, objective="iwae_elbo")      # keys lpxzy, lpzy, lqzxy

22 trainable tensors: encoder (8), decoder (6), conditional prior network (8), in that (Keras creation) order.
"""
from . import task05


class CIWAE(task05.CIWAE):
    _rename = {"lpxz": "lpxzy", "lpz": "lpzy", "lqzx": "lqzxy"}       # tasks/task04.py:163-173

    def __init__(self, n_hidden, n_latent, **kwargs):
        super().__init__(n_hidden, n_latent, cond_prior=True, **kwargs)
