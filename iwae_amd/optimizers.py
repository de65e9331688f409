"""The slice of keras.optimizers.Adam that main.py touches (main.py:93-94,128-133):
Adam(lr, epsilon=1e-4), .learning_rate.numpy(), .learning_rate.assign(v).  The moment
buffers and the update itself live on the GPU inside the model handle (adam_kernel)."""
import numpy as np


class _LearningRate:
    def __init__(self, v):
        self._v = float(v)

    def numpy(self):
        return np.float32(self._v)

    def assign(self, v):
        self._v = float(v)
        return self

    def __float__(self):
        return self._v


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        if abs(beta_1 - 0.9) > 1e-12 or abs(beta_2 - 0.999) > 1e-12:
            raise ValueError("iwae_amd Adam: beta_1/beta_2 are fixed to the Keras defaults 0.9/0.999")
        if abs(epsilon - 1e-4) > 1e-12:
            raise ValueError("iwae_amd Adam: epsilon is fixed to 1e-4, the value the reference trains with (main.py:93)")
        self.learning_rate = _LearningRate(learning_rate)
        self.epsilon = epsilon
        self.iterations = 0
