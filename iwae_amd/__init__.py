"""iwae_amd -- MI355X-native (gfx950) drop-in for the train / eval step of nbip/IWAE.

Layout (only what the hot path needs):
  csrc/        hand-written HIP kernels + the C ABI (libiwae_amd.so, include/iwae_amd.h)
  _capi.py     ctypes binding of the C ABI
  native.py    numpy-in / numpy-out wrapper over the ABI
  iwae1.py     reference API of src/iwae1.py   (IWAE, train_step, val_step, sample)
  iwae2.py     reference API of src/iwae2.py   (2 stochastic layers)
  task02.py    reference API of tasks/task02.py (IWAEDReG)
  task04.py    reference API of tasks/task04.py (CIWAE with the learned conditional prior p(z|y))
  task05.py    reference API of tasks/task05.py (CIWAE: encoder on concat(x, y), decoder on concat(z, y))
  optimizers.py  the keras.optimizers.Adam surface main.py uses
  utils.py     logmeanexp / bernoullisample / MyMetric / get_bias (src/utils.py)
  parallel.py  data-parallel step: shard, RCCL all-reduce of the flat gradient, Adam

Importing the package never touches the GPU; constructing a model does, and raises if
libiwae_amd.so or the GPU is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
