"""NativeModel: numpy-in / numpy-out wrapper over the C ABI (include/iwae_amd.h).

This is the layer the reference-API shims (iwae1.py, iwae2.py, task02.py) delegate to.
All compute happens in libiwae_amd.so on the GPU; nothing here falls back to the CPU.
"""
import ctypes as C
import numpy as np

from . import _capi
from ._capi import OBJECTIVES, PRECISIONS, Config, Scalars, Tensors, check

_SCALAR_NAMES = ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14", "inference_loss",
                 "mean_lpxz", "mean_lpz", "mean_lqzx", "mean_kl")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class NativeModel:
    def __init__(self, n_layers, n_hidden, n_latent, x_dim=784, device=0, seed=123, world_size=1, rank=0, cond_dim=0, cond_prior=False, precision="bf16",
                 options=None):
        self.lib = _capi.load()
        cfg = Config()
        cfg.n_layers = int(n_layers)
        nh = list(n_hidden) if isinstance(n_hidden, (list, tuple)) else [n_hidden]
        nl = list(n_latent) if isinstance(n_latent, (list, tuple)) else [n_latent]
        for i in range(2):
            cfg.n_hidden[i] = int(nh[i]) if i < len(nh) else 0
            cfg.n_latent[i] = int(nl[i]) if i < len(nl) else 0
        cfg.x_dim, cfg.device, cfg.seed = int(x_dim), int(device), int(seed)
        cfg.world_size, cfg.rank = int(world_size), int(rank)
        cfg.cond_dim = int(cond_dim)
        cfg.cond_prior = 1 if cond_prior else 0
        cfg.precision = PRECISIONS[precision]
        self.precision = precision
        self.cond_dim = int(cond_dim)
        self.n_layers, self.x_dim = cfg.n_layers, cfg.x_dim
        self.n_hidden, self.n_latent = nh[:cfg.n_layers], nl[:cfg.n_layers]
        h = C.c_void_p()
        check(self.lib.iwae_create(C.byref(cfg), C.byref(h)))
        self.h = h
        n = C.c_size_t()
        check(self.lib.iwae_param_count(self.h, C.byref(n)))
        self.n_params = n.value
        for name, value in (options or {}).items():      # kernel-selection switches (A/B measurements, variant tests): iwae_set_option
            self.set_option(name, value)

    def set_option(self, name, value=1):
        check(self.lib.iwae_set_option(self.h, str(name).encode(), int(value)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.iwae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------
    def tensor_table(self):
        n = C.c_int32()
        check(self.lib.iwae_num_tensors(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.create_string_buffer(64)
            r, c, off = C.c_int32(), C.c_int32(), C.c_size_t()
            check(self.lib.iwae_tensor_info(self.h, i, name, 64, C.byref(r), C.byref(c), C.byref(off)))
            shape = (r.value,) if name.value.endswith(b"bias") else (r.value, c.value)
            out.append((name.value.decode(), shape, off.value))
        return out

    def get_params(self):
        a = np.empty(self.n_params, dtype=np.float32)
        check(self.lib.iwae_get_params(self.h, a.ctypes.data, a.size))
        return a

    def set_params(self, flat):
        a = _f32(flat).ravel()
        check(self.lib.iwae_set_params(self.h, a.ctypes.data, a.size))

    def set_output_bias(self, bias):
        a = _f32(bias).ravel()
        check(self.lib.iwae_set_output_bias(self.h, a.ctypes.data, a.size))

    def get_grads(self):
        a = np.empty(self.n_params, dtype=np.float32)
        check(self.lib.iwae_get_grads(self.h, a.ctypes.data, a.size))
        return a

    def get_adam_state(self):
        m = np.empty(self.n_params, dtype=np.float32)
        v = np.empty(self.n_params, dtype=np.float32)
        t = C.c_int64()
        check(self.lib.iwae_get_adam_state(self.h, m.ctypes.data, v.ctypes.data, m.size, C.byref(t)))
        return m, v, t.value

    def set_adam_state(self, m, v, t):
        m, v = _f32(m).ravel(), _f32(v).ravel()
        check(self.lib.iwae_set_adam_state(self.h, m.ctypes.data, v.ctypes.data, m.size, int(t)))

    def grad_devptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(self.lib.iwae_grad_devptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_stream(self, stream_ptr):
        check(self.lib.iwae_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_condition(self, y):
        """Conditional model (cond_dim > 0): y [n, cond_dim] float32 for the next forward / train step / eval_llh / decode."""
        y = _f32(y).reshape(-1, self.cond_dim)
        check(self.lib.iwae_set_condition(self.h, y.ctypes.data, y.shape[0]))

    # ---- data-parallel training inside the library (RCCL) -------------------------------
    @staticmethod
    def comm_unique_id():
        """Rank 0: the opaque id blob every rank passes to comm_init (ship it with any channel)."""
        lib = _capi.load()
        buf = C.create_string_buffer(1024)
        n = C.c_size_t()
        check(lib.iwae_comm_unique_id(buf, 1024, C.byref(n)))
        return buf.raw[:n.value]

    def comm_init(self, unique_id, world_size, rank):
        """Collective over all ranks: from now on train_step all-reduces the gradient inside the library."""
        blob = bytes(unique_id)
        check(self.lib.iwae_comm_init(self.h, blob, len(blob), int(world_size), int(rank)))
        self.comm = True

    def comm_preflight(self, unique_id, world_size, rank):
        """The non-collective checks of comm_init (arguments, state, RCCL loadable); raises like comm_init would, without the rendezvous."""
        blob = bytes(unique_id)
        check(self.lib.iwae_comm_preflight(self.h, blob, len(blob), int(world_size), int(rank)))

    def comm_info(self):
        """(world size, rank) as RCCL reports them for the handle's communicators; (0, -1) without communicators."""
        w, r = C.c_int32(), C.c_int32()
        check(self.lib.iwae_comm_info(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_destroy(self):
        check(self.lib.iwae_comm_destroy(self.h))
        self.comm = False

    def set_step(self, noise_step, batch_offset=0):
        check(self.lib.iwae_set_step(self.h, int(noise_step), int(batch_offset)))

    def sync(self):
        check(self.lib.iwae_sync(self.h))

    # ---- calls ------------------------------------------------------------------------
    def _eps_arg(self, eps, B, k):
        if eps is None:
            return None, None
        if self.n_layers == 1:
            e = _f32(eps)
            assert e.shape == (k, B, self.n_latent[0]), e.shape
            return e, e.ctypes.data
        e1, e2 = _f32(eps[0]), _f32(eps[1])
        assert e1.shape == (k, B, self.n_latent[0]) and e2.shape == (k, B, self.n_latent[1])
        e = np.concatenate([e1.ravel(), e2.ravel()])
        return e, e.ctypes.data

    def _want(self, want, B, k):
        if not want:
            return None, {}
        t = Tensors()
        D1 = self.n_latent[0]
        D2 = self.n_latent[1] if self.n_layers == 2 else 0
        shapes = {"z": (k, B, D1), "z2": (k, B, D2), "snis_z": (B, D1), "snis_z2": (B, D2), "al": (k, B),
                  "logits": (k, B, self.x_dim), "lpxz": (k, B), "lpz": (k, B), "lqzx": (k, B), "lpz2": (k, B),
                  "lqzx2": (k, B), "log_w": (k, B)}
        bufs = {}
        for name in want:
            if name in ("z2", "snis_z2", "lpz2", "lqzx2") and self.n_layers == 1:
                continue
            bufs[name] = np.empty(shapes[name], dtype=np.float32)
            setattr(t, name, bufs[name].ctypes.data)
        return t, bufs

    @staticmethod
    def _scalars_dict(s):
        return {n: float(getattr(s, n)) for n in _SCALAR_NAMES}

    def forward(self, x, k, beta=1.0, eps=None, want=()):
        x = _f32(x)
        B = x.shape[0]
        keep, ep = self._eps_arg(eps, B, k)
        t, bufs = self._want(want, B, k)
        s = Scalars()
        check(self.lib.iwae_forward(self.h, x.ctypes.data, B, int(k), float(beta), ep, C.byref(s),
                                    C.byref(t) if t is not None else None))
        out = self._scalars_dict(s)
        out.update(bufs)
        return out

    def forward_backward(self, x, k, beta=1.0, objective="iwae_elbo", eps=None, want=()):
        x = _f32(x)
        B = x.shape[0]
        keep, ep = self._eps_arg(eps, B, k)
        t, bufs = self._want(want, B, k)
        s = Scalars()
        check(self.lib.iwae_forward_backward(self.h, x.ctypes.data, B, int(k), float(beta), OBJECTIVES[objective], ep,
                                             C.byref(s), C.byref(t) if t is not None else None))
        out = self._scalars_dict(s)
        out.update(bufs)
        return out

    def set_adam(self, beta_1=0.9, beta_2=0.999, epsilon=1e-4):
        """keras.optimizers.Adam hyper-parameters of the device optimizer (default: the reference's, main.py:93)."""
        check(self.lib.iwae_set_adam(self.h, float(beta_1), float(beta_2), float(epsilon)))

    def set_eval_precision(self, precision):
        """Arithmetic of eval_llh: "fp32" (default, the reference's) or "bf16" (the fast path)."""
        check(self.lib.iwae_set_eval_precision(self.h, PRECISIONS[precision]))

    def adam_step(self, lr, grad_scale=1.0):
        check(self.lib.iwae_adam_step(self.h, float(lr), float(grad_scale)))

    def train_step(self, x, k, beta=1.0, lr=1e-3, objective="iwae_elbo", eps=None, want=(), scalars=True):
        x = _f32(x)
        B = x.shape[0]
        keep, ep = self._eps_arg(eps, B, k)
        t, bufs = self._want(want, B, k)
        s = Scalars()
        check(self.lib.iwae_train_step(self.h, x.ctypes.data, B, int(k), float(beta), float(lr), OBJECTIVES[objective], ep,
                                       C.byref(s) if scalars else None, C.byref(t) if t is not None else None))
        out = self._scalars_dict(s) if scalars else {}
        out.update(bufs)
        return out

    def train_step_devptr(self, x_devptr, B, k, beta, lr, objective_id):
        """Hot loop entry for benchmarks: x already resident in HBM, no host round trip."""
        check(self.lib.iwae_train_step(self.h, C.c_void_p(x_devptr), int(B), int(k), float(beta), float(lr), int(objective_id),
                                       None, None, None))

    def forward_backward_devptr(self, x_devptr, B, k, beta, objective_id):
        check(self.lib.iwae_forward_backward(self.h, C.c_void_p(x_devptr), int(B), int(k), float(beta), int(objective_id),
                                             None, None, None))

    def forward_backward_split_devptr(self, x_devptr, B, k, beta, objective_id):
        """iwae_forward_backward_split: returns (side stream handle, first float of the gradient segment completed on it)."""
        side, off = C.c_void_p(), C.c_size_t()
        check(self.lib.iwae_forward_backward_split(self.h, C.c_void_p(x_devptr), int(B), int(k), float(beta), int(objective_id),
                                                   None, C.byref(side), C.byref(off)))
        return side.value or 0, off.value

    def eval_llh(self, x, k=5000, chunk=0, per_image=False):
        x = _f32(x)
        N = x.shape[0]
        llh = C.c_double()
        pi = np.empty(N, dtype=np.float32) if per_image else None
        check(self.lib.iwae_eval_llh(self.h, x.ctypes.data, N, int(k), int(chunk), C.byref(llh),
                                     pi.ctypes.data if per_image else None))
        return (llh.value, pi) if per_image else llh.value

    def decode(self, z):
        z = _f32(z)
        out = np.empty((z.shape[0], self.x_dim), dtype=np.float32)
        check(self.lib.iwae_decode(self.h, z.ctypes.data, z.shape[0], out.ctypes.data))
        return out

    # ---- resident dataset (device-side shuffle order + dynamic binarisation) -------------------
    def dataset_upload(self, gray_u8):
        g = np.ascontiguousarray(gray_u8, dtype=np.uint8).reshape(-1, self.x_dim)
        check(self.lib.iwae_dataset_upload(self.h, g.ctypes.data, g.shape[0]))
        self.n_data = g.shape[0]

    def dataset_begin_epoch(self, epoch, order=None):
        if order is None:
            check(self.lib.iwae_dataset_begin_epoch(self.h, int(epoch), None, 0))
        else:
            o = np.ascontiguousarray(order, dtype=np.int32)
            check(self.lib.iwae_dataset_begin_epoch(self.h, int(epoch), o.ctypes.data, o.size))

    def dataset_get_batch(self, start, B):
        out = np.empty((B, self.x_dim), dtype=np.float32)
        check(self.lib.iwae_dataset_get_batch(self.h, int(start), int(B), out.ctypes.data))
        return out

    def dataset_set_labels(self, labels):
        """One class id (< cond_dim) per image of the uploaded set: conditional models on the resident dataset (tasks/task05.py:296-322)."""
        y = np.ascontiguousarray(labels, dtype=np.uint8).ravel()
        check(self.lib.iwae_dataset_set_labels(self.h, y.ctypes.data, y.size))

    def dataset_get_labels(self, start, B):
        out = np.empty((B, self.cond_dim), dtype=np.float32)
        check(self.lib.iwae_dataset_get_labels(self.h, int(start), int(B), out.ctypes.data))
        return out

    def train_step_dataset(self, start, B, k, beta=1.0, lr=1e-3, objective="iwae_elbo", scalars=True):
        s = Scalars()
        check(self.lib.iwae_train_step_dataset(self.h, int(start), int(B), int(k), float(beta), float(lr), OBJECTIVES[objective],
                                               C.byref(s) if scalars else None))
        return self._scalars_dict(s) if scalars else {}

    def enable_timing(self, every=1):
        """every = n > 0: bracket the dominant kernels of every n-th step with HIP events; 0 / False: off."""
        check(self.lib.iwae_enable_timing(self.h, int(every)))

    def kernel_time(self, name):
        us, cnt = C.c_double(), C.c_int64()
        check(self.lib.iwae_kernel_time(self.h, name.encode(), C.byref(us), C.byref(cnt)))
        return us.value, cnt.value

    def debug_tensor(self, name):
        r, c = C.c_int32(), C.c_int32()
        check(self.lib.iwae_debug_tensor(self.h, name.encode(), None, 0, C.byref(r), C.byref(c)))
        out = np.empty((r.value, c.value), dtype=np.float32)
        check(self.lib.iwae_debug_tensor(self.h, name.encode(), out.ctypes.data, out.size, C.byref(r), C.byref(c)))
        return out

    def debug_eps(self, B, k, layer=0):
        out = np.empty((k, B, self.n_latent[layer]), dtype=np.float32)
        check(self.lib.iwae_debug_eps(self.h, B, k, layer, out.ctypes.data))
        return out
