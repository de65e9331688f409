"""ctypes binding of libiwae_amd.so (include/iwae_amd.h).  There is no CPU fallback: if the
library is missing or no AMD GPU is present, construction raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiwae_amd.so")

OBJECTIVES = {"vae_elbo": 0, "iwae_elbo": 1, "iwae_eq14": 2, "vae_elbo_kl": 3, "dreg": 4}
PRECISIONS = {"bf16": 0, "fp32": 1}


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_layers", C.c_int32), ("n_hidden", C.c_int32 * 2), ("n_latent", C.c_int32 * 2),
                ("x_dim", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64),
                ("world_size", C.c_int32), ("rank", C.c_int32), ("cond_dim", C.c_int32), ("cond_prior", C.c_int32),
                ("precision", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(Config)       # the ABI guard iwae_create checks


class Scalars(C.Structure):
    _fields_ = [("vae_elbo", C.c_float), ("vae_elbo_kl", C.c_float), ("iwae_elbo", C.c_float),
                ("iwae_eq14", C.c_float), ("inference_loss", C.c_float), ("mean_lpxz", C.c_float),
                ("mean_lpz", C.c_float), ("mean_lqzx", C.c_float), ("mean_kl", C.c_float),
                ("reserved", C.c_float * 7)]


TENSOR_FIELDS = ("z", "z2", "snis_z", "snis_z2", "al", "logits", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2", "log_w")


class Tensors(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TENSOR_FIELDS]


# every symbol include/iwae_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "iwae_last_error": (C.c_char_p, []),
    "iwae_version": (C.c_int, []),
    "iwae_build_id": (C.c_char_p, []),
    "iwae_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "iwae_destroy": (None, [_P]),
    "iwae_set_stream": (C.c_int, [_P, _P]),
    "iwae_sync": (C.c_int, [_P]),
    "iwae_param_count": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "iwae_num_tensors": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "iwae_tensor_info": (C.c_int, [_P, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "iwae_set_params": (C.c_int, [_P, _P, C.c_size_t]),
    "iwae_get_params": (C.c_int, [_P, _P, C.c_size_t]),
    "iwae_set_output_bias": (C.c_int, [_P, _P, C.c_size_t]),
    "iwae_get_grads": (C.c_int, [_P, _P, C.c_size_t]),
    "iwae_get_adam_state": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_int64)]),
    "iwae_set_adam_state": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int64]),
    "iwae_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, _P, C.POINTER(Scalars), C.POINTER(Tensors)]),
    "iwae_train_step": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, _P, C.POINTER(Scalars), C.POINTER(Tensors)]),
    "iwae_forward_backward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, C.POINTER(Scalars), C.POINTER(Tensors)]),
    "iwae_forward_backward_split": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "iwae_grad_devptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "iwae_adam_step": (C.c_int, [_P, C.c_float, C.c_float]),
    "iwae_set_adam": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "iwae_set_eval_precision": (C.c_int, [_P, C.c_int32]),
    "iwae_set_step": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "iwae_comm_unique_id": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "iwae_comm_init": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, C.c_int32]),
    "iwae_comm_preflight": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, C.c_int32]),
    "iwae_comm_destroy": (C.c_int, [_P]),
    "iwae_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "iwae_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "iwae_eval_llh": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), _P]),
    "iwae_decode": (C.c_int, [_P, _P, C.c_int32, _P]),
    "iwae_dataset_upload": (C.c_int, [_P, _P, C.c_int32]),
    "iwae_dataset_begin_epoch": (C.c_int, [_P, C.c_uint32, _P, C.c_int32]),
    "iwae_dataset_get_batch": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "iwae_dataset_set_labels": (C.c_int, [_P, _P, C.c_int32]),
    "iwae_dataset_get_labels": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "iwae_train_step_dataset": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.POINTER(Scalars)]),
    "iwae_set_condition": (C.c_int, [_P, _P, C.c_int32]),
    "iwae_enable_timing": (C.c_int, [_P, C.c_int32]),
    "iwae_kernel_time": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "iwae_debug_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "iwae_debug_eps": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
}

_lib = None

_ID_SOURCES = ("build.sh", "fp32_kernels.hip", "kernels.h", "kernels.hip", "layout.h", "model.hip", os.path.join("..", "..", "include", "iwae_amd.h"))


def source_build_id():
    """What csrc/build.sh would stamp into a library built from THIS tree (same files, order and framing): sha256, 16 hex digits."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for f in _ID_SOURCES:
        h.update(("== %s\n" % os.path.basename(f)).encode())
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_build_id():
    """iwae_build_id() of the loaded library ("unknown": built without csrc/build.sh)."""
    return load().iwae_build_id().decode()


def file_build_id(path=None):
    """The id stamped into the library FILE, read from its bytes (no dlopen: a process that has already loaded an older build of the
    same path would be handed that one again).  None: no file; "unknown": no stamp."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        blob = fh.read()
    i = blob.find(b"IWAE_BUILD_ID=")
    if i < 0:
        return "unknown"
    j = blob.find(b"\0", i)
    return blob[i + 14:j].decode("ascii", "replace")


def load():
    """Load libiwae_amd.so (built in-tree by __graft_entry__.build() / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("iwae_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the IWAE hot path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class IwaeError(RuntimeError):
    pass


def check(rc):
    if rc == 0:
        return
    msg = load().iwae_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -3:
        raise MemoryError(msg)
    raise IwaeError("iwae_amd error %d: %s" % (rc, msg))
