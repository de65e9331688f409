"""The slice of keras.optimizers.Adam that main.py touches (main.py:93-94,128-133):
Adam(lr, epsilon=1e-4), .learning_rate.numpy(), .learning_rate.assign(v).  The moment
buffers and the update itself live on the GPU inside the model handle (adam_element in kernels.hip)."""
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
    """keras.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon) -- Keras defaults; the reference passes epsilon=1e-4
    (main.py:93).  The hyper-parameters travel to the device optimizer (iwae_set_adam) with the first train_step."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kwargs):
        if not (0.0 <= beta_1 < 1.0 and 0.0 <= beta_2 < 1.0 and epsilon > 0.0):
            raise ValueError("Adam: need 0 <= beta_1, beta_2 < 1 and epsilon > 0")
        self.learning_rate = _LearningRate(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0

    def hyper(self):
        return (self.beta_1, self.beta_2, self.epsilon)
