"""ctypes harness over the C ABI (include/metalchat_hip.h).  See package docstring."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

BF16, F32 = 0, 1
FAMILY_LLAMA3, FAMILY_GEMMA3 = 0, 1
WFMT_T, WFMT_I8, WFMT_I4 = 0, 1, 2
QMODE_EXACT, QMODE_FAST = 0, 1


class McError(RuntimeError):
    """status 1 -> std::invalid_argument, 2 -> std::runtime_error, 3 -> alloc_error."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def library_path() -> str:
    return os.path.join(HERE, "lib", "libmetalchat_hip.so")


def hsaco_path() -> str:
    return os.path.join(HERE, "lib", "metalchat.hsaco")


class DecoderConfig(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("family", C.c_int32),
        ("dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("ffn_dim", C.c_int32), ("n_layers", C.c_int32),
        ("vocab", C.c_int32), ("max_seq_len", C.c_int32),
        ("rope_theta", C.c_float), ("rope_sliding_theta", C.c_float),
        ("sliding_stride", C.c_int32), ("norm_eps", C.c_float), ("attn_scale", C.c_float),
        ("sink_pre_len", C.c_int32), ("layer_begin", C.c_int32), ("layer_end", C.c_int32),
        ("weight_format", C.c_int32), ("group_size", C.c_int32), ("qmode", C.c_int32),
        ("use_graph", C.c_int32),
    ]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_char_p), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 8), ("data", C.c_void_p), ("nbytes", C.c_size_t)]


CKPT_META_LLAMA3, CKPT_HF_LLAMA3, CKPT_META_LLAMA3_QLORA, CKPT_HF_GEMMA3 = 0, 1, 2, 3
SAMPLER_GREEDY, SAMPLER_DEFAULT = 0, 1

_lib = None


def capi() -> C.CDLL:
    """Loads libmetalchat_hip.so and declares every prototype of include/metalchat_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise McError(2, f"hip: {path} is missing -- run `python -m metalchat_amd.build` "
                         "(there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, sz, i32, u32, u64, f32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint64, C.c_float
    pvp = C.POINTER(C.c_void_p)
    P = {
        "mc_last_error": (C.c_char_p, []),
        "mc_version": (C.c_char_p, []),
        "mc_trace_ranges_enabled": (i32, []),
        "mc_device_create": (i32, [i32, pvp]),
        "mc_device_release": (None, [vp]),
        "mc_device_count": (i32, []),
        "mc_device_name": (C.c_char_p, [vp]),
        "mc_device_max_buffer_size": (sz, [vp]),
        "mc_device_ordinal": (i32, [vp]),
        "mc_device_compute_units": (i32, [vp]),
        "mc_library_open": (i32, [vp, C.c_char_p, pvp]),
        "mc_library_release": (None, [vp]),
        "mc_library_get_kernel": (i32, [vp, C.c_char_p, pvp]),
        "mc_kernel_release": (None, [vp]),
        "mc_kernel_name": (C.c_char_p, [vp]),
        "mc_kernel_max_threads_per_group": (sz, [vp]),
        "mc_buffer_alloc": (i32, [vp, sz, pvp]),
        "mc_buffer_alloc_copy": (i32, [vp, vp, sz, pvp]),
        "mc_buffer_wrap_nocopy": (i32, [vp, vp, sz, pvp]),
        "mc_buffer_release": (None, [vp]),
        "mc_buffer_contents": (vp, [vp]),
        "mc_buffer_length": (sz, [vp]),
        "mc_buffer_upload": (i32, [vp, sz, vp, sz]),
        "mc_buffer_download": (i32, [vp, sz, vp, sz]),
        "mc_buffer_fill_zero": (i32, [vp, sz, sz]),
        "mc_queue_create": (i32, [vp, vp, pvp]),
        "mc_queue_release": (None, [vp]),
        "mc_queue_stream": (vp, [vp]),
        "mc_encoder_set_kernel": (i32, [vp, vp]),
        "mc_encoder_set_bytes": (i32, [vp, vp, sz]),
        "mc_encoder_set_buffer": (i32, [vp, vp, sz]),
        "mc_encoder_memory_barrier": (i32, [vp, vp]),
        "mc_encoder_dispatch_threads": (i32, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "mc_encoder_dispatch_threads_lds": (i32, [vp, C.POINTER(sz), C.POINTER(sz), sz]),
        "mc_queue_on_completed": (i32, [vp, vp, vp]),
        "mc_queue_commit": (i32, [vp]),
        "mc_queue_wait": (i32, [vp]),
        "mc_queue_timer_begin": (i32, [vp]),
        "mc_queue_timer_end": (i32, [vp]),
        "mc_queue_timer_elapsed_ms": (i32, [vp, C.POINTER(f32)]),
        "mc_decoder_create": (i32, [vp, vp, vp, C.POINTER(DecoderConfig), pvp]),
        "mc_decoder_release": (None, [vp]),
        "mc_decoder_load_linear": (i32, [vp, i32, C.c_char_p, i32, i32, i32, i32, vp, vp]),
        "mc_decoder_load_vector": (i32, [vp, i32, C.c_char_p, i32, vp]),
        "mc_decoder_load_lora": (i32, [vp, i32, C.c_char_p, i32, i32, i32, vp, vp, C.c_float]),
        "mc_decoder_init_synthetic": (i32, [vp, u64]),
        "mc_decoder_step": (i32, [vp, i32, i32, vp, C.POINTER(i32)]),
        "mc_decoder_generate": (i32, [vp, i32, i32, i32, C.POINTER(i32)]),
        "mc_decoder_prefill": (i32, [vp, C.POINTER(i32), i32, i32, i32, C.POINTER(i32)]),
        "mc_decoder_hidden_out": (vp, [vp]),
        "mc_decoder_hidden_in": (vp, [vp]),
        "mc_pipeline_layer_range": (None, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
        "mc_pipeline_unique_id": (i32, [vp]),
        "mc_pipeline_create": (i32, [vp, i32, i32, vp, pvp]),
        "mc_pipeline_create_local": (i32, [pvp, i32, pvp]),
        "mc_pipeline_generate": (i32, [vp, i32, i32, i32, C.POINTER(i32)]),
        "mc_pipeline_allreduce_max": (i32, [vp, C.POINTER(C.c_double)]),
        "mc_pipeline_prefill": (i32, [vp, C.POINTER(i32), i32, i32, i32, C.POINTER(i32)]),
        "mc_decoder_prefill_stage": (i32, [vp, C.POINTER(i32), vp, i32, i32, i32, pvp, C.POINTER(i32)]),
        "mc_pipeline_release": (None, [vp]),
        "mc_pipeline_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "mc_decoder_set_taps": (i32, [vp, i32]),
        "mc_decoder_get_logits": (i32, [vp, vp]),
        "mc_decoder_get_hidden": (i32, [vp, i32, vp]),
        "mc_decoder_export_kv": (i32, [vp, i32, vp, vp, C.POINTER(i32)]),
        "mc_decoder_import_kv": (i32, [vp, i32, vp, vp, i32]),
        "mc_decoder_weight_bytes": (sz, [vp]),
        "mc_decoder_time_gemv": (i32, [vp, C.c_char_p, i32, C.POINTER(f32), C.POINTER(C.c_double),
                                       C.POINTER(i32)]),
        "mc_decoder_gemv_kernel_name": (i32, [vp, C.c_char_p, C.c_char_p, sz]),
        "mc_decoder_handoff_fallbacks": (i32, [vp]),
        "mc_decoder_handoff_rearms": (i32, [vp]),
        "mc_decoder_handoffs_active": (i32, [vp]),
        "mc_decoder_derived_weight_bytes": (C.c_size_t, [vp]),
        "mc_decoder_launch_log": (i32, [vp, i32]),
        "mc_decoder_launch_log_read": (sz, [vp, C.c_char_p, sz]),
        "mc_decoder_weight_ptrs": (i32, [vp, i32, C.c_char_p, pvp, pvp, C.POINTER(i32),
                                         C.POINTER(i32), C.POINTER(i32)]),
        "mc_decoder_get_config": (i32, [vp, C.POINTER(DecoderConfig)]),
        "mc_decoder_set_sampler": (i32, [vp, i32, i32, C.c_float, C.c_float]),
        "mc_decoder_set_seeds": (i32, [vp, C.POINTER(u64), i32]),
        "mc_decoder_get_sampler_taps": (i32, [vp, C.POINTER(f32)]),
        "mc_document_create": (i32, [pvp]),
        "mc_document_open": (i32, [C.c_char_p, pvp]),
        "mc_document_open_sharded": (i32, [C.c_char_p, pvp]),
        "mc_document_release": (None, [vp]),
        "mc_document_size": (i32, [vp]),
        "mc_document_tensor": (i32, [vp, i32, C.POINTER(TensorInfo)]),
        "mc_document_find": (i32, [vp, C.c_char_p, C.POINTER(TensorInfo)]),
        "mc_document_insert": (i32, [vp, C.c_char_p, C.c_char_p, i32, C.POINTER(C.c_int64), vp]),
        "mc_document_link": (i32, [vp, C.c_char_p, C.c_char_p]),
        "mc_document_set_metadata": (i32, [vp, C.c_char_p, C.c_char_p]),
        "mc_document_metadata": (C.c_char_p, [vp, C.c_char_p]),
        "mc_document_adapt": (i32, [vp, i32]),
        "mc_document_save": (i32, [vp, C.c_char_p]),
        "mc_config_from_json": (i32, [C.c_char_p, i32, C.POINTER(DecoderConfig)]),
        "mc_config_from_document": (i32, [vp, C.POINTER(DecoderConfig)]),
        "mc_decoder_load_document": (i32, [vp, vp, i32]),
        "mc_gpt2_encode": (i32, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]),
        "mc_gpt2_decode": (i32, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]),
        "mc_regexp_split": (i32, [C.c_char_p, C.c_char_p, sz, C.POINTER(sz), sz, C.POINTER(sz)]),
        "mc_tokenizer_create": (i32, [C.c_char_p, pvp]),
        "mc_tokenizer_open_tiktoken": (i32, [C.c_char_p, C.c_char_p, pvp]),
        "mc_tokenizer_open_hf": (i32, [C.c_char_p, pvp]),
        "mc_tokenizer_create_sentence_piece": (i32, [pvp]),
        "mc_tokenizer_open_hf_gemma3": (i32, [C.c_char_p, pvp]),
        "mc_tokenizer_release": (None, [vp]),
        "mc_tokenizer_insert": (i32, [vp, C.c_char_p, sz, i32, i32]),
        "mc_tokenizer_insert_back": (i32, [vp, C.c_char_p, sz, i32]),
        "mc_tokenizer_size": (sz, [vp]),
        "mc_tokenizer_encode": (i32, [vp, C.c_char_p, sz, C.POINTER(i32), sz, C.POINTER(sz)]),
        "mc_tokenizer_encode_control": (i32, [vp, i32, C.POINTER(i32)]),
        "mc_tokenizer_decode": (i32, [vp, C.POINTER(i32), sz, C.c_char_p, sz, C.POINTER(sz)]),
        "mc_interpreter_create": (i32, [vp, vp, pvp]),
        "mc_interpreter_release": (None, [vp]),
        "mc_interpreter_set_scanner": (i32, [vp, sz, C.POINTER(i32), sz, i32]),
        "mc_interpreter_declare_variable": (i32, [vp, C.c_char_p, C.c_char_p]),
        "mc_interpreter_write": (i32, [vp, C.c_char_p, C.c_char_p]),
        "mc_interpreter_read": (i32, [vp, i32, C.c_char_p, sz, C.POINTER(sz), C.POINTER(i32), sz, C.POINTER(sz)]),
        "mc_interpreter_start_pos": (sz, [vp]),
        "mc_interpreter_pending": (i32, [vp, C.POINTER(i32), sz, C.POINTER(sz)]),
        "mc_synth_weight": (i32, [u64, u32, u32, u32, i32]),
        "mc_synth_scale": (f32, [u64, u32, u32, u32, i32, i32]),
        "mc_synth_value": (f32, [u64, u32, u32, i32, u32]),
    }
    for name, (res, args) in P.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    lib._prototypes = P
    _lib = lib
    return lib


def _check(status: int):
    if status != 0:
        raise McError(status, capi().mc_last_error().decode())


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def layout(sizes, strides=None, offsets=None) -> np.ndarray:
    """tensor_layout<N>: 3N uint32 words {sizes, strides, offsets} (kernel/tensor.h:10-14)."""
    sizes = list(sizes)
    n = len(sizes)
    if strides is None:
        strides, acc = [0] * n, 1
        for d in range(n - 1, -1, -1):
            strides[d] = acc
            acc *= sizes[d]
    if offsets is None:
        offsets = [0] * n
    return np.array(list(sizes) + list(strides) + list(offsets), dtype=np.uint32)


def make_kernel_grid_2d(num_rows: int, dim_size: int, max_threads: int):
    """src/kernel.cc:13-37 -- returns (grid, thread) as 3-tuples of TOTAL threads / group size."""
    if num_rows * dim_size <= max_threads:
        return (dim_size, num_rows, 1), (dim_size, num_rows, 1)
    if dim_size <= max_threads:
        return (dim_size, num_rows, 1), (dim_size, 1, 1)
    groups = (dim_size + max_threads - 1) // max_threads
    return (max_threads * groups, num_rows, 1), (max_threads, 1, 1)


class Buffer:
    """metal::shared_buffer / hardware_memory_container (include/metalchat/container.h:641-700)."""

    def __init__(self, acc: "HardwareAccelerator", handle, keep=None):
        self.acc, self._h, self._keep = acc, handle, keep

    @property
    def nbytes(self) -> int:
        return capi().mc_buffer_length(self._h)

    @property
    def device_ptr(self) -> int:
        return capi().mc_buffer_contents(self._h)

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        _check(capi().mc_buffer_upload(self._h, offset, _np_ptr(a), a.nbytes))

    def download(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        _check(capi().mc_buffer_download(self._h, offset, _np_ptr(out), out.nbytes))
        return out

    def release(self):
        if self._h:
            capi().mc_buffer_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Kernel:
    """basic_kernel (include/metalchat/kernel.h:59-98)."""

    def __init__(self, acc, handle, name):
        self.acc, self._h, self.name = acc, handle, name

    def max_threads_per_threadgroup(self) -> int:
        return capi().mc_kernel_max_threads_per_group(self._h)


_ST_NP = {"F32": np.float32, "BF16": np.uint16, "F16": np.float16, "I8": np.int8, "U8": np.uint8,
          "I16": np.int16, "U16": np.uint16, "I32": np.int32, "U32": np.uint32, "I64": np.int64,
          "U64": np.uint64, "F64": np.float64, "BOOL": np.bool_}


class Document:
    """safetensor_document (include/metalchat/safetensor.h:534-975) over the C ABI, host only.
    bf16 tensors are exposed as uint16 arrays (numpy has no bfloat16)."""

    def __init__(self, path: str | None = None, sharded: bool = False):
        self._h = C.c_void_p()
        if path is None:
            _check(capi().mc_document_create(C.byref(self._h)))
        elif sharded:
            _check(capi().mc_document_open_sharded(os.fspath(path).encode(), C.byref(self._h)))
        else:
            _check(capi().mc_document_open(os.fspath(path).encode(), C.byref(self._h)))

    def __len__(self):
        return capi().mc_document_size(self._h)

    @staticmethod
    def _unpack(ti: TensorInfo):
        shape = tuple(ti.shape[i] for i in range(ti.ndim))
        dtype = ti.dtype.decode()
        npd = _ST_NP.get(dtype)
        arr = None
        if npd is not None:
            n = ti.nbytes // np.dtype(npd).itemsize
            arr = (np.ctypeslib.as_array(C.cast(ti.data, C.POINTER(C.c_uint8)), (ti.nbytes,)).view(npd)[:n]
                   .reshape(shape) if ti.nbytes else np.zeros(shape, npd))
        return dict(name=ti.name.decode(), dtype=dtype, shape=shape, data=arr, address=ti.data, nbytes=ti.nbytes)

    def tensor(self, index: int) -> dict:
        ti = TensorInfo()
        _check(capi().mc_document_tensor(self._h, index, C.byref(ti)))
        return self._unpack(ti)

    def find(self, name: str) -> dict:
        ti = TensorInfo()
        _check(capi().mc_document_find(self._h, name.encode(), C.byref(ti)))
        return self._unpack(ti)

    def names(self):
        return [self.tensor(i)["name"] for i in range(len(self))]

    def insert(self, name: str, array: np.ndarray, dtype: str | None = None):
        array = np.ascontiguousarray(array)
        if dtype is None:
            dtype = {v: k for k, v in _ST_NP.items() if k != "BF16"}[array.dtype.type]
        shape = (C.c_int64 * max(array.ndim, 1))(*array.shape)
        _check(capi().mc_document_insert(self._h, name.encode(), dtype.encode(), array.ndim, shape, _np_ptr(array)))

    def link(self, name: str, source: str):
        _check(capi().mc_document_link(self._h, name.encode(), source.encode()))

    def set_metadata(self, key: str, value: str):
        _check(capi().mc_document_set_metadata(self._h, key.encode(), value.encode()))

    def metadata(self, key: str):
        v = capi().mc_document_metadata(self._h, key.encode())
        return v.decode() if v is not None else None

    def adapt(self, flavour: int):
        _check(capi().mc_document_adapt(self._h, flavour))

    def save(self, path: str):
        _check(capi().mc_document_save(self._h, os.fspath(path).encode()))

    def release(self):
        if self._h:
            capi().mc_document_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def config_from_json(text: str, flavour: int, cfg: DecoderConfig | None = None) -> DecoderConfig:
    cfg = cfg or DecoderConfig()
    _check(capi().mc_config_from_json(text.encode(), flavour, C.byref(cfg)))
    return cfg


def config_from_document(doc: Document, cfg: DecoderConfig) -> DecoderConfig:
    _check(capi().mc_config_from_document(doc._h, C.byref(cfg)))
    return cfg


class KernelTask:
    """kernel_task (include/metalchat/kernel.h:101-298): positional arguments, then dispatch.
    Arguments: a Buffer (optionally (Buffer, byte_offset)), a numpy uint32 layout array, or a
    numpy scalar / 0-d array passed by value.  Tensor = layout then buffer, exactly as
    hardware_function_encoder::encode (include/metalchat/kernel_thread.h:111-125)."""

    def __init__(self, kernel: Kernel, grid, thread, args=(), lds_bytes: int = 0):
        self.kernel, self.grid, self.thread, self.args, self.lds = kernel, grid, thread, list(args), lds_bytes

    def bind_front(self, *front):
        return KernelTask(self.kernel, self.grid, self.thread, list(front) + self.args, self.lds)

    def bind_back(self, *back):
        return KernelTask(self.kernel, self.grid, self.thread, self.args + list(back), self.lds)

    def __call__(self):
        lib, q = capi(), self.kernel.acc._queue
        _check(lib.mc_encoder_set_kernel(q, self.kernel._h))
        for a in self.args:
            if a is None:
                _check(lib.mc_encoder_set_buffer(q, None, 0))
            elif isinstance(a, Buffer):
                _check(lib.mc_encoder_set_buffer(q, a._h, 0))
                _check(lib.mc_encoder_memory_barrier(q, a._h))
            elif isinstance(a, tuple) and isinstance(a[0], Buffer):
                _check(lib.mc_encoder_set_buffer(q, a[0]._h, a[1]))
                _check(lib.mc_encoder_memory_barrier(q, a[0]._h))
            else:
                arr = np.ascontiguousarray(a)
                _check(lib.mc_encoder_set_bytes(q, _np_ptr(arr), arr.nbytes))
        g = (C.c_size_t * 3)(*self.grid)
        t = (C.c_size_t * 3)(*self.thread)
        _check(lib.mc_encoder_dispatch_threads_lds(q, g, t, self.lds))


class HardwareAccelerator:
    """hardware_accelerator (include/metalchat/accelerator.h:55-219, src/accelerator.cc:26-158):
    device + shader library + kernel cache + the command queue."""

    def __init__(self, path: str | None = None, ordinal: int = -1, stream: int | None = None):
        lib = capi()
        self._dev = C.c_void_p()
        _check(lib.mc_device_create(ordinal, C.byref(self._dev)))
        self._lib = C.c_void_p()
        _check(lib.mc_library_open(self._dev, (path or os.environ.get("MC_HSACO") or hsaco_path()).encode(), C.byref(self._lib)))  # MC_HSACO: tuning builds (tools/)
        self._queue = C.c_void_p()
        _check(lib.mc_queue_create(self._dev, C.c_void_p(stream) if stream else None,
                                   C.byref(self._queue)))
        self._kernels: dict[str, Kernel] = {}

    def name(self) -> str:
        return capi().mc_device_name(self._dev).decode()

    def max_buffer_size(self) -> int:
        return capi().mc_device_max_buffer_size(self._dev)

    def compute_units(self) -> int:
        return capi().mc_device_compute_units(self._dev)

    def load(self, name: str, *types) -> Kernel:
        """load("rmsnorm", "bfloat") -> kernel "rmsnorm_bfloat" (accelerator.h:175-218)."""
        full = "_".join([name, *[str(t) for t in types]])
        if full not in self._kernels:
            h = C.c_void_p()
            _check(capi().mc_library_get_kernel(self._lib, full.encode(), C.byref(h)))
            self._kernels[full] = Kernel(self, h, full)
        return self._kernels[full]

    def alloc(self, nbytes: int) -> Buffer:
        h = C.c_void_p()
        _check(capi().mc_buffer_alloc(self._dev, nbytes, C.byref(h)))
        return Buffer(self, h)

    def to_device(self, a: np.ndarray) -> Buffer:
        a = np.ascontiguousarray(a)
        h = C.c_void_p()
        _check(capi().mc_buffer_alloc_copy(self._dev, _np_ptr(a), a.nbytes, C.byref(h)))
        return Buffer(self, h)

    def wrap(self, device_ptr: int, nbytes: int, keep=None) -> Buffer:
        h = C.c_void_p()
        _check(capi().mc_buffer_wrap_nocopy(self._dev, C.c_void_p(device_ptr), nbytes, C.byref(h)))
        return Buffer(self, h, keep)

    def wait(self):
        _check(capi().mc_queue_wait(self._queue))

    def timer_begin(self):
        _check(capi().mc_queue_timer_begin(self._queue))

    def timer_end_ms(self) -> float:
        _check(capi().mc_queue_timer_end(self._queue))
        ms = C.c_float()
        _check(capi().mc_queue_timer_elapsed_ms(self._queue, C.byref(ms)))
        return ms.value

    def stream(self) -> int:
        return capi().mc_queue_stream(self._queue)


def device_count() -> int:
    return capi().mc_device_count()


def pipeline_unique_id() -> bytes:
    """The 128-byte RCCL id rank 0 creates and hands to every rank (file, socket, ...)."""
    buf = C.create_string_buffer(128)
    _check(capi().mc_pipeline_unique_id(buf))
    return buf.raw


def pipeline_layer_range(rank: int, world: int, n_layers: int):
    lb, le = C.c_int32(), C.c_int32()
    capi().mc_pipeline_layer_range(rank, world, n_layers, C.byref(lb), C.byref(le))
    return lb.value, le.value


class Pipeline:
    """mc_pipeline_*: the layer pipeline behind the C ABI.  Pipeline.local(stages) keeps all stages in this
    process (device-to-device hops); Pipeline.rccl(stage, rank, world, uid) is one process per GPU."""

    def __init__(self, handle, keep):
        self._h, self._keep = handle, keep

    @classmethod
    def local(cls, stages):
        arr = (C.c_void_p * len(stages))(*[d._h for d in stages])
        h = C.c_void_p()
        _check(capi().mc_pipeline_create_local(arr, len(stages), C.byref(h)))
        return cls(h, list(stages))

    @classmethod
    def rccl(cls, stage, rank: int, world: int, uid: bytes):
        assert len(uid) == 128
        h = C.c_void_p()
        _check(capi().mc_pipeline_create(stage._h, rank, world, uid, C.byref(h)))
        return cls(h, [stage])

    def generate(self, first_token: int, start_pos: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.int32)
        _check(capi().mc_pipeline_generate(self._h, first_token, start_pos, n, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def prefill(self, tokens, start_pos: int, sliding_window: int = 0) -> int:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        nxt = C.c_int32(-1)
        _check(capi().mc_pipeline_prefill(self._h, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), start_pos, sliding_window,
                                          C.byref(nxt)))
        return nxt.value

    def allreduce_max(self, value: float = 0.0) -> float:
        """Barrier over the stages + device synchronise; returns max(value) over the ranks."""
        v = C.c_double(value)
        _check(capi().mc_pipeline_allreduce_max(self._h, C.byref(v)))
        return v.value

    def comm_info(self):
        """(ranks, rank) as the transport reports them (ncclCommCount / ncclCommUserRank); (-1, -1) for a local pipeline"""
        n, r = C.c_int32(), C.c_int32()
        _check(capi().mc_pipeline_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def release(self):
        if self._h:
            capi().mc_pipeline_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Decoder:
    """The fused decode pipeline (Part 2 of the C ABI): nn::llama3 / nn::gemma3 per-token step."""

    def __init__(self, acc: HardwareAccelerator, **cfg):
        self.acc = acc
        c = DecoderConfig()
        defaults = dict(family=FAMILY_LLAMA3, rope_sliding_theta=0.0, sliding_stride=0,
                        sink_pre_len=-1, layer_begin=0, layer_end=cfg["n_layers"],
                        weight_format=WFMT_T, group_size=0, qmode=QMODE_EXACT, use_graph=0)
        defaults.update(cfg)
        for k, _ in DecoderConfig._fields_:
            setattr(c, k, defaults[k])
        self.cfg = defaults
        self._c = c
        self._h = C.c_void_p()
        _check(capi().mc_decoder_create(acc._dev, acc._lib, acc._queue, C.byref(c), C.byref(self._h)))
        self.np_T = np.uint16 if defaults["dtype"] == BF16 else np.float32

    @classmethod
    def from_config(cls, acc: "HardwareAccelerator", c: DecoderConfig) -> "Decoder":
        """A decoder from a filled DecoderConfig (mc_config_from_json / mc_config_from_document)."""
        return cls(acc, **{k: getattr(c, k) for k, _ in DecoderConfig._fields_})

    # -- weights ------------------------------------------------------------------------------
    def load_linear(self, layer: int, name: str, fmt: int, weight: np.ndarray, scales=None,
                    group_size: int = 0):
        weight = np.ascontiguousarray(weight)
        out_f, in_f = weight.shape
        sc = np.ascontiguousarray(scales, dtype=np.float32) if scales is not None else None
        _check(capi().mc_decoder_load_linear(self._h, layer, name.encode(), fmt, out_f, in_f,
                                             group_size, _np_ptr(weight),
                                             _np_ptr(sc) if sc is not None else None))

    def load_vector(self, layer: int, name: str, data: np.ndarray):
        data = np.ascontiguousarray(data)
        _check(capi().mc_decoder_load_vector(self._h, layer, name.encode(), data.size, _np_ptr(data)))

    def load_lora(self, layer: int, name: str, a: np.ndarray, b: np.ndarray, scale: float):
        """a: T[rank, in], b: T[out, rank] (quantization::lora_adaptor, lora.h:17-53)."""
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        rank, in_f = a.shape
        out_f = b.shape[0]
        assert b.shape[1] == rank
        _check(capi().mc_decoder_load_lora(self._h, layer, name.encode(), rank, out_f, in_f,
                                           _np_ptr(a), _np_ptr(b), C.c_float(scale)))

    # -- sampler ------------------------------------------------------------------------------
    def set_sampler(self, kind: int, top_k: int = 50, temperature: float = 0.6, top_p: float = 0.9):
        """kind: SAMPLER_GREEDY / SAMPLER_DEFAULT (make_default_sampler, nn/sampling.h:303-313)."""
        _check(capi().mc_decoder_set_sampler(self._h, kind, top_k, temperature, top_p))
        self._top_k = min(top_k, self.cfg["vocab"])

    def set_seeds(self, pairs):
        a = np.ascontiguousarray(np.asarray(pairs, dtype=np.uint64).reshape(-1, 2))
        _check(capi().mc_decoder_set_seeds(self._h, a.ctypes.data_as(C.POINTER(C.c_uint64)), a.shape[0]))

    def sampler_taps(self) -> np.ndarray:
        out = np.zeros((7, self._top_k), np.float32)
        _check(capi().mc_decoder_get_sampler_taps(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def load_document(self, doc: "Document", flavour: int):
        _check(capi().mc_decoder_load_document(self._h, doc._h, flavour))

    def load_model(self, weights: dict):
        """weights: the dict produced by tests/modelgen.py (reference-native formats)."""
        fmt_of = {0: WFMT_T, 1: None, 2: WFMT_I8}
        lb, le = self.cfg["layer_begin"], self.cfg["layer_end"]
        for li in range(lb, le):
            lw = weights["layers"][li]
            for n in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
                spec = lw[n]
                if spec["kind"] == 0:
                    self.load_linear(li, n, WFMT_T, spec["weight"])
                else:
                    self.load_linear(li, n, spec["hbm_format"], spec["weight"], spec["scales"],
                                     spec.get("group_size", 0) if spec["kind"] == 1 else 0)
                if spec.get("lora_a") is not None:
                    self.load_lora(li, n, spec["lora_a"], spec["lora_b"], spec["lora_scale"])
            for n in ("attention_norm", "ffn_norm", "q_norm", "k_norm", "attention_post_norm",
                      "ffn_post_norm"):
                if lw.get(n) is not None:
                    self.load_vector(li, n, lw[n])
        if lb == 0:
            e = weights["embedding"]
            if e["kind"] == 0:
                self.load_linear(-1, "tok_embeddings", WFMT_T, e["weight"])
            else:
                self.load_linear(-1, "tok_embeddings", WFMT_I8, e["weight"], e["scales"])
        if le == self.cfg["n_layers"]:
            o = weights["output"]
            if o["kind"] == 0:
                self.load_linear(-1, "output", WFMT_T, o["weight"])
            else:
                self.load_linear(-1, "output", o["hbm_format"], o["weight"], o["scales"],
                                 o.get("group_size", 0) if o["kind"] == 1 else 0)
            self.load_vector(-1, "norm", weights["final_norm"])

    def init_synthetic(self, seed: int):
        _check(capi().mc_decoder_init_synthetic(self._h, seed))

    # -- stepping -----------------------------------------------------------------------------
    def step(self, token: int, start_pos: int, hidden_in: int | None = None, sync: bool = True):
        nt = C.c_int32(-1)
        _check(capi().mc_decoder_step(self._h, token, start_pos,
                                      C.c_void_p(hidden_in) if hidden_in else None,
                                      C.byref(nt) if sync else None))
        return nt.value

    def prefill(self, tokens, start_pos: int = 0, sliding_window: int = 0) -> int:
        """The prompt pass on len(tokens) rows; returns the sampler's pick for the last row."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        nt = C.c_int32(-1)
        _check(capi().mc_decoder_prefill(self._h, t.ctypes.data_as(C.POINTER(C.c_int32)), t.size, start_pos,
                                         sliding_window, C.byref(nt)))
        return nt.value

    def generate(self, first_token: int, start_pos: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.int32)
        _check(capi().mc_decoder_generate(self._h, first_token, start_pos, n,
                                          out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def hidden_out_ptr(self) -> int:
        return capi().mc_decoder_hidden_out(self._h)

    def hidden_in_ptr(self) -> int:
        return capi().mc_decoder_hidden_in(self._h)

    def set_taps(self, enable: bool):
        _check(capi().mc_decoder_set_taps(self._h, 1 if enable else 0))

    def logits(self) -> np.ndarray:
        out = np.empty(self.cfg["vocab"], dtype=self.np_T)
        _check(capi().mc_decoder_get_logits(self._h, _np_ptr(out)))
        return out

    def hidden(self, layer: int) -> np.ndarray:
        out = np.empty(self.cfg["dim"], dtype=self.np_T)
        _check(capi().mc_decoder_get_hidden(self._h, layer, _np_ptr(out)))
        return out

    def export_kv(self, layer: int):
        c = self.cfg
        shape = (c["max_seq_len"], c["n_kv_heads"], c["head_dim"])
        k = np.zeros(shape, dtype=self.np_T)
        v = np.zeros(shape, dtype=self.np_T)
        n = C.c_int32()
        _check(capi().mc_decoder_export_kv(self._h, layer, _np_ptr(k), _np_ptr(v), C.byref(n)))
        return k[: n.value], v[: n.value]

    def import_kv(self, layer: int, keys: np.ndarray, values: np.ndarray):
        """Test aid: logical rows [n, n_kv_heads, head_dim] of T become positions 0 .. n-1 of `layer`'s cache."""
        k = np.ascontiguousarray(keys, dtype=self.np_T)
        v = np.ascontiguousarray(values, dtype=self.np_T)
        assert k.shape == v.shape and k.shape[1:] == (self.cfg["n_kv_heads"], self.cfg["head_dim"])
        _check(capi().mc_decoder_import_kv(self._h, layer, _np_ptr(k), _np_ptr(v), k.shape[0]))

    def weight_bytes(self) -> int:
        return capi().mc_decoder_weight_bytes(self._h)

    def time_gemv(self, which: str, repeats: int):
        ms, by, ln = C.c_float(), C.c_double(), C.c_int32()
        _check(capi().mc_decoder_time_gemv(self._h, which.encode(), repeats, C.byref(ms),
                                           C.byref(by), C.byref(ln)))
        return ms.value, by.value, ln.value

    def gemv_kernel_name(self, which: str) -> str:
        """host name of the kernel time_gemv(which) -- i.e. a token -- launches for that matrix"""
        buf = C.create_string_buffer(256)
        _check(capi().mc_decoder_gemv_kernel_name(self._h, which.encode(), buf, 256))
        return buf.value.decode()

    def derived_weight_bytes(self) -> int:
        """HBM held by derived copies of the weights (mc_decoder_derived_weight_bytes)."""
        return int(capi().mc_decoder_derived_weight_bytes(self._h))

    def handoff_fallbacks(self) -> int:
        """how often an in-launch hand-off gave up and the decoder fell back to launches that need no co-residency"""
        return int(capi().mc_decoder_handoff_fallbacks(self._h))

    def handoff_rearms(self) -> int:
        """how often the decoder went BACK to the hand-off launches after a fall-back (256 clean tokens, doubling; MC_HANDOFF_REARM)"""
        return int(capi().mc_decoder_handoff_rearms(self._h))

    def handoffs_active(self) -> bool:
        return bool(capi().mc_decoder_handoffs_active(self._h))

    def launch_log(self, enable: bool = True):
        """start (and clear) / stop recording the names of the kernels this decoder launches"""
        _check(capi().mc_decoder_launch_log(self._h, 1 if enable else 0))

    def launched(self) -> list[str]:
        n = capi().mc_decoder_launch_log_read(self._h, None, 0)
        buf = C.create_string_buffer(n)
        capi().mc_decoder_launch_log_read(self._h, buf, n)
        return [x for x in buf.value.decode().split("\n") if x]

    def weight_ptrs(self, layer: int, name: str):
        w, s = C.c_void_p(), C.c_void_p()
        rows, inf, ng = C.c_int32(), C.c_int32(), C.c_int32()
        _check(capi().mc_decoder_weight_ptrs(self._h, layer, name.encode(), C.byref(w), C.byref(s),
                                             C.byref(rows), C.byref(inf), C.byref(ng)))
        return w.value, s.value, rows.value, inf.value, ng.value

    def release(self):
        if self._h:
            capi().mc_decoder_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------ Part 4: text
TOKEN_REGULAR, TOKEN_BEGIN_TEXT, TOKEN_END_TEXT, TOKEN_RESERVED, TOKEN_FINETUNE_RIGHT_PAD = 1, 2, 4, 8, 16
TOKEN_BEGIN_HEADER, TOKEN_END_HEADER, TOKEN_END_MESSAGE, TOKEN_END_TURN, TOKEN_IPYTHON = 32, 64, 128, 256, 512


def _bytes_call(fn, *args) -> bytes:
    """Calls an ABI function whose last three arguments are (out, cap, n): sizes, then fills."""
    n = C.c_size_t()
    _check(fn(*args, None, 0, C.byref(n)))
    buf = C.create_string_buffer(max(n.value, 1))
    _check(fn(*args, buf, n.value, C.byref(n)))
    return buf.raw[:n.value]


def gpt2_encode(data: bytes) -> str:
    """text::gpt2_codec::encode"""
    return _bytes_call(capi().mc_gpt2_encode, data, len(data)).decode("utf-8")


def gpt2_decode(text: str) -> bytes:
    """text::gpt2_codec::decode"""
    raw = text.encode("utf-8")
    return _bytes_call(capi().mc_gpt2_decode, raw, len(raw))


def regexp_split(pattern: bytes | None, subject: bytes) -> list[bytes]:
    """text::regexp::begin .. end over `subject` (None: the llama3 pattern)."""
    n = C.c_size_t()
    _check(capi().mc_regexp_split(pattern, subject, len(subject), None, 0, C.byref(n)))
    off = (C.c_size_t * max(2 * n.value, 1))()
    _check(capi().mc_regexp_split(pattern, subject, len(subject), off, n.value, C.byref(n)))
    return [subject[off[2 * i]:off[2 * i + 1]] for i in range(n.value)]


class Tokenizer:
    """text::byte_pair_encoder<char> (+ the llama3 loaders)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def create(cls, token_regex: bytes | None = None):
        h = C.c_void_p()
        _check(capi().mc_tokenizer_create(token_regex, C.byref(h)))
        return cls(h)

    @classmethod
    def open_tiktoken(cls, path: str, token_regex: bytes | None = None):
        h = C.c_void_p()
        _check(capi().mc_tokenizer_open_tiktoken(path.encode(), token_regex, C.byref(h)))
        return cls(h)

    @classmethod
    def open_hf(cls, path: str):
        h = C.c_void_p()
        _check(capi().mc_tokenizer_open_hf(path.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def create_sentence_piece(cls):
        """text::sentence_piece(): byte-pair merging over code points, spaces as U+2581."""
        h = C.c_void_p()
        _check(capi().mc_tokenizer_create_sentence_piece(C.byref(h)))
        return cls(h)

    @classmethod
    def open_hf_gemma3(cls, path: str):
        """huggingface::gemma3_tokenizer_loader::load (tokenizer.json)."""
        h = C.c_void_p()
        _check(capi().mc_tokenizer_open_hf_gemma3(path.encode(), C.byref(h)))
        return cls(h)

    def release(self):
        if self._h:
            capi().mc_tokenizer_release(self._h)
            self._h = None

    def insert(self, value: bytes, key: int, kind: int = TOKEN_REGULAR):
        _check(capi().mc_tokenizer_insert(self._h, value, len(value), key, kind))

    def insert_back(self, value: bytes, kind: int = TOKEN_REGULAR):
        _check(capi().mc_tokenizer_insert_back(self._h, value, len(value), kind))

    def __len__(self):
        return capi().mc_tokenizer_size(self._h)

    def encode(self, text: bytes | str) -> list[int]:
        raw = text.encode("utf-8") if isinstance(text, str) else text
        n = C.c_size_t()
        _check(capi().mc_tokenizer_encode(self._h, raw, len(raw), None, 0, C.byref(n)))
        ids = (C.c_int32 * max(n.value, 1))()
        _check(capi().mc_tokenizer_encode(self._h, raw, len(raw), ids, n.value, C.byref(n)))
        return list(ids[:n.value])

    def encode_control(self, kind: int) -> int:
        v = C.c_int32()
        _check(capi().mc_tokenizer_encode_control(self._h, kind, C.byref(v)))
        return v.value

    def decode(self, ids) -> bytes:
        ids = [ids] if isinstance(ids, int) else list(ids)
        arr = (C.c_int32 * max(len(ids), 1))(*ids)
        return _bytes_call(capi().mc_tokenizer_decode, self._h, arr, len(ids))


class Interpreter:
    """metalchat::interpreter over a Decoder and a Tokenizer (message framing + the read loop)."""

    def __init__(self, dec: "Decoder | None", tok: Tokenizer):
        self.dec, self.tok = dec, tok
        h = C.c_void_p()
        _check(capi().mc_interpreter_create(dec._h if dec is not None else None, tok._h, C.byref(h)))
        self._h = h

    def release(self):
        if self._h:
            capi().mc_interpreter_release(self._h)
            self._h = None

    def set_token_scanner(self, limit: int = 0, stop_ids=(), op_and: bool = True):
        stop = list(stop_ids)
        arr = (C.c_int32 * max(len(stop), 1))(*stop)
        _check(capi().mc_interpreter_set_scanner(self._h, limit, arr, len(stop), 1 if op_and else 0))

    def declare_variable(self, name: str, value: str):
        _check(capi().mc_interpreter_declare_variable(self._h, name.encode(), value.encode()))

    def write(self, role: str, content: str | bytes):
        raw = content.encode("utf-8") if isinstance(content, str) else content
        _check(capi().mc_interpreter_write(self._h, role.encode(), raw))

    def read(self, sliding_window: int = 0, max_tokens: int = 4096):
        """-> (text bytes, ids)"""
        n, nid = C.c_size_t(), C.c_size_t()
        buf = C.create_string_buffer(max_tokens * 64)
        ids = (C.c_int32 * max_tokens)()
        _check(capi().mc_interpreter_read(self._h, sliding_window, buf, len(buf), C.byref(n), ids, max_tokens,
                                          C.byref(nid)))
        return buf.raw[:min(n.value, len(buf))], list(ids[:min(nid.value, max_tokens)])

    @property
    def start_pos(self) -> int:
        return capi().mc_interpreter_start_pos(self._h)

    def pending(self) -> list[int]:
        n = C.c_size_t()
        _check(capi().mc_interpreter_pending(self._h, None, 0, C.byref(n)))
        ids = (C.c_int32 * max(n.value, 1))()
        _check(capi().mc_interpreter_pending(self._h, ids, n.value, C.byref(n)))
        return list(ids[:n.value])
