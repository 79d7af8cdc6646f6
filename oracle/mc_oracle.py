"""ctypes binding of the CPU oracle (oracle/mc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (metalchat_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmc_oracle.so")

BF16, F32, I32 = 0, 1, 2


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mc_oracle.c")
    hdr = os.path.join(_HERE, "mc_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # MC_ORACLE_SO: bench.py's cpu_baseline leg loads a build made -O3 -march=native on the timing host
        _lib = C.CDLL(os.environ.get("MC_ORACLE_SO") or build())
        _lib.mco_bf16_to_f32.restype = C.c_float
        _lib.mco_bf16_to_f32.argtypes = [C.c_uint16]
        _lib.mco_f32_to_bf16.restype = C.c_uint16
        _lib.mco_f32_to_bf16.argtypes = [C.c_float]
        _lib.mco_model_create.restype = C.c_void_p
        _lib.mco_pcg32_uniform.restype = C.c_float
        _lib.mco_pcg32_uniform.argtypes = [C.c_uint64, C.c_uint64]
        _lib.mco_sample_default.restype = C.c_int32
        _lib.mco_model_step.restype = C.c_int32
        _lib.mco_model_forward.restype = C.c_int32
        _lib.mco_model_step_range.restype = C.c_int32
        _lib.mco_model_get_kv.restype = C.c_int32
        _lib.mco_model_set_kv.restype = C.c_int32
        # small test shapes: a handful of threads beats one OpenMP team per host core
        _lib.mco_set_num_threads(C.c_int(min(8, os.cpu_count() or 1)))
    return _lib


# ---------------------------------------------------------------- bf16 <-> f32 (numpy, RNE)
def to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bits (uint16), round-to-nearest-even, NaN preserved."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    nan = ((u & 0x7F800000) == 0x7F800000) & ((u & 0x007FFFFF) != 0)
    r = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    r = np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)
    return r.astype(np.uint16)


def from_bf16(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def round_bf16(x: np.ndarray) -> np.ndarray:
    return from_bf16(to_bf16(x))


def np_dtype(dt: int):
    return {BF16: np.uint16, F32: np.float32, I32: np.int32}[dt]


def encode(dt: int, x: np.ndarray) -> np.ndarray:
    """float values -> storage array of element type dt."""
    if dt == BF16:
        return to_bf16(x)
    return np.ascontiguousarray(x, dtype=np.float32)


def decode(dt: int, x: np.ndarray) -> np.ndarray:
    if dt == BF16:
        return from_bf16(x)
    return np.asarray(x, dtype=np.float32)


# ---------------------------------------------------------------- layouts
def layout(sizes, strides=None, offsets=None) -> np.ndarray:
    """tensor_layout<N> as 3N uint32 words {sizes, strides, offsets} (kernel/tensor.h:10-14)."""
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


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _l(l):
    return l.ctypes.data_as(C.POINTER(C.c_uint32))


# ---------------------------------------------------------------- kernels
def bmm(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_bmm(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


def hadamard(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_hadamard(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


def hadamard_broadcast(odt, sdt, out_l, out, in1_l, in1, in2_l, in2):
    lib().mco_hadamard_broadcast(odt, sdt, _l(out_l), _p(out), _l(in1_l), _p(in1), _l(in2_l), _p(in2))


def scalar_mul(dt, out_l, out, in_l, inp, mult):
    m = encode(dt, np.array([mult], dtype=np.float32))
    lib().mco_scalar_mul(dt, _l(out_l), _p(out), _l(in_l), _p(inp), _p(m))


def rmsnorm(dt, out_l, out, in_l, inp, w_l, w, eps, mu, max_threads=1024):
    lib().mco_rmsnorm(dt, _l(out_l), _p(out), _l(in_l), _p(inp), _l(w_l), _p(w),
                      C.c_float(eps), C.c_float(mu), C.c_uint32(max_threads))


def rope(dt, out_l, out, in_l, inp, cos_l, fcos, sin_l, fsin, batch_size, n_head, start_pos):
    lib().mco_rope(dt, _l(out_l), _p(out), _l(in_l), _p(inp), _l(cos_l), _p(fcos), _l(sin_l),
                   _p(fsin), C.c_uint32(batch_size), C.c_uint32(n_head), C.c_uint32(start_pos))


def rope_freqs(cos_l, fcos, sin_l, fsin, dim, start_pos, theta):
    lib().mco_rope_freqs(_l(cos_l), _p(fcos), _l(sin_l), _p(fsin), C.c_uint32(dim),
                         C.c_uint32(start_pos), C.c_float(theta))


def softmax(dt, out_l, out, in_l, inp, max_threads=1024):
    lib().mco_softmax(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_uint32(max_threads))


def embedding(dt, out_l, out, in_l, ids, w_l, w):
    lib().mco_embedding(dt, _l(out_l), _p(out), _l(in_l), _p(ids), _l(w_l), _p(w))


def copy(dt, out_l, out, in_l, inp):
    lib().mco_copy(BF16 if dt == BF16 else F32, _l(out_l), _p(out), _l(in_l), _p(inp))


def roll(dt, out_l, out, in_l, inp, shift, size, stride):
    lib().mco_roll(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_uint32(shift),
                   C.c_uint32(size), C.c_uint32(stride))


def add(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_add(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


def add_broadcast(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_add_broadcast(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


# ---- sampler chain (include/metalchat/nn/sampling.h:152-315)
def sub(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_sub(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


def div(dt, out_l, out, a_l, a, b_l, b):
    lib().mco_div(dt, _l(out_l), _p(out), _l(a_l), _p(a), _l(b_l), _p(b))


def row_sum(dt, out_l, out, in_l, inp, max_threads=1024):
    lib().mco_sum(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_uint32(max_threads))


def gt(dt, out_l, out, in_l, inp, value):
    lib().mco_gt(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_float(value))


def le(dt, out_l, out, in_l, inp, value):
    lib().mco_le(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_float(value))


def scatter(dt, out_l, out, mask_l, mask, value):
    lib().mco_scatter(dt, _l(out_l), _p(out), _l(mask_l), _p(mask), C.c_float(value))


def gather(dt, out_l, out, in_l, inp, index_l, index):
    """dt 2 = int32 payload"""
    lib().mco_gather(dt, _l(out_l), _p(out), _l(in_l), _p(inp), _l(index_l), _p(index))


def sort(dt, values_l, values, indices_l, indices, in_l, inp):
    lib().mco_sort(dt, _l(values_l), _p(values), _l(indices_l), _p(indices), _l(in_l), _p(inp))


def cumsum(dt, out_l, out, in_l, inp, max_threads=1024):
    lib().mco_cumsum(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_uint32(max_threads))


def pcg32_uniform(init_state: int, init_seq: int) -> float:
    return float(lib().mco_pcg32_uniform(C.c_uint64(init_state & (2**64 - 1)), C.c_uint64(init_seq & (2**64 - 1))))


def multinomial(dt, out_l, out, in_l, inp, init_state, init_seq):
    lib().mco_multinomial(dt, _l(out_l), _p(out), _l(in_l), _p(inp), C.c_uint64(init_state), C.c_uint64(init_seq))


def topk(dt, logits, k):
    n = logits.size
    k = min(k, n)
    values = np.zeros(k, dtype=logits.dtype)
    indices = np.zeros(k, dtype=np.int32)
    lib().mco_topk(dt, _p(logits), None, C.c_uint32(n), C.c_uint32(k), _p(values), _p(indices))
    return values, indices


def sample_default(dt, logits, top_k=50, temperature=0.6, top_p=0.9, init_state=0, init_seq=0, taps=False):
    """make_default_sampler (nn/sampling.h:303-313) on one row of logits -> vocabulary id
    (and, with taps, the [7, k] float table of intermediates)."""
    k = min(top_k, logits.size)
    t = np.zeros((7, k), np.float32) if taps else None
    tok = lib().mco_sample_default(dt, _p(logits), C.c_uint32(logits.size), C.c_uint32(top_k),
                                   C.c_float(temperature), C.c_float(top_p), C.c_uint64(init_state),
                                   C.c_uint64(init_seq), _p(t))
    return (tok, t) if taps else tok


def silu(dt, out_l, out, in_l, inp):
    lib().mco_silu(dt, _l(out_l), _p(out), _l(in_l), _p(inp))


def gelu(dt, out_l, out, in_l, inp):
    lib().mco_gelu(dt, _l(out_l), _p(out), _l(in_l), _p(inp))


# ---------------------------------------------------------------- model
class Linear(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("out_features", C.c_int32), ("in_features", C.c_int32),
        ("group_size", C.c_int32), ("weight", C.c_void_p), ("scales", C.c_void_p),
        ("lora_rank", C.c_int32), ("lora_a", C.c_void_p), ("lora_b", C.c_void_p),
        ("lora_scale", C.c_float),
    ]


class LayerWeights(C.Structure):
    _fields_ = [
        ("wq", Linear), ("wk", Linear), ("wv", Linear), ("wo", Linear),
        ("w1", Linear), ("w2", Linear), ("w3", Linear),
        ("attention_norm", C.c_void_p), ("ffn_norm", C.c_void_p),
        ("q_norm", C.c_void_p), ("k_norm", C.c_void_p),
        ("attention_post_norm", C.c_void_p), ("ffn_post_norm", C.c_void_p),
        ("rope_table", C.c_int32),
    ]


class ModelOptions(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("family", C.c_int32),
        ("dim", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("ffn_dim", C.c_int32), ("n_layers", C.c_int32),
        ("vocab", C.c_int32), ("max_seq_len", C.c_int32),
        ("rope_theta", C.c_float), ("rope_sliding_theta", C.c_float),
        ("norm_eps", C.c_float), ("attn_scale", C.c_float), ("sink_pre_len", C.c_int32),
    ]


def make_linear(spec) -> Linear:
    """spec: dict(kind, weight(np), scales(np|None), group_size, lora_a, lora_b, lora_scale)."""
    w = spec["weight"]
    L = Linear()
    L.kind = spec["kind"]
    L.out_features, L.in_features = w.shape
    L.group_size = spec.get("group_size", 0) or 0
    L.weight = w.ctypes.data
    sc = spec.get("scales")
    L.scales = sc.ctypes.data if sc is not None else None
    la, lb = spec.get("lora_a"), spec.get("lora_b")
    L.lora_rank = la.shape[0] if la is not None else 0
    L.lora_a = la.ctypes.data if la is not None else None
    L.lora_b = lb.ctypes.data if lb is not None else None
    L.lora_scale = spec.get("lora_scale", 0.0)
    return L


class Model:
    """Wraps mco_model.  `weights` is the dict produced by tests/modelgen.py (reference-native
    formats: T or int8-held weights + f32 scales); the arrays are kept alive here."""

    def __init__(self, cfg: dict, weights: dict):
        self._keep = weights
        o = ModelOptions()
        o.dtype = cfg["dtype"]
        o.family = cfg.get("family", 0)
        for k in ("dim", "n_heads", "n_kv_heads", "head_dim", "ffn_dim", "n_layers", "vocab",
                  "max_seq_len"):
            setattr(o, k, cfg[k])
        o.rope_theta = cfg["rope_theta"]
        o.rope_sliding_theta = cfg.get("rope_sliding_theta", 0.0)
        o.norm_eps = cfg["norm_eps"]
        o.attn_scale = cfg["attn_scale"]
        o.sink_pre_len = cfg.get("sink_pre_len", -1)
        self.cfg = cfg
        layers = (LayerWeights * cfg["n_layers"])()
        for i, lw in enumerate(weights["layers"]):
            for name in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
                setattr(layers[i], name, make_linear(lw[name]))
            for name in ("attention_norm", "ffn_norm", "q_norm", "k_norm",
                         "attention_post_norm", "ffn_post_norm"):
                a = lw.get(name)
                setattr(layers[i], name, a.ctypes.data if a is not None else None)
            layers[i].rope_table = lw.get("rope_table", 0)
        emb = weights["embedding"]
        out = make_linear(weights["output"])
        esc = emb.get("scales")
        self._h = C.c_void_p(lib().mco_model_create(
            C.byref(o), layers, C.c_int32(emb["kind"]), C.c_void_p(emb["weight"].ctypes.data),
            C.c_void_p(esc.ctypes.data if esc is not None else None),
            C.c_void_p(weights["final_norm"].ctypes.data), C.byref(out)))

    def step(self, token: int, start_pos: int, want_logits: bool = True):
        dt = self.cfg["dtype"]
        logits = np.empty(self.cfg["vocab"], dtype=np_dtype(dt)) if want_logits else None
        tok = lib().mco_model_step(self._h, C.c_int32(token), C.c_int32(start_pos), _p(logits))
        return tok, logits

    def forward(self, tokens, start_pos: int = 0, sliding_window: int = 0):
        """The prompt pass on len(tokens) tokens; returns (greedy token, logits of the last row)."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        logits = np.empty(self.cfg["vocab"], dtype=np_dtype(self.cfg["dtype"]))
        tok = lib().mco_model_forward(self._h, _p(t), C.c_int32(t.size), C.c_int32(start_pos),
                                      C.c_int32(sliding_window), _p(logits))
        if tok == -2:  # the reference throws std::invalid_argument here (nn/cache.h:178-183, kernel/copy.h:38-39)
            raise ValueError("sink_cache: the chunk does not fit the cache")
        return tok, logits

    def step_range(self, token: int, start_pos: int, layer_begin: int, layer_end: int,
                   hidden_in=None):
        """One pipeline stage; returns (token or -1, hidden_out or None)."""
        dt = self.cfg["dtype"]
        last = layer_end == self.cfg["n_layers"]
        hout = None if last else np.empty(self.cfg["dim"], dtype=np_dtype(dt))
        tok = lib().mco_model_step_range(self._h, C.c_int32(token), C.c_int32(start_pos),
                                         C.c_int32(layer_begin), C.c_int32(layer_end),
                                         _p(hidden_in), _p(hout), None)
        return tok, hout

    def hidden(self, layer: int) -> np.ndarray:
        out = np.empty(self.cfg["dim"], dtype=np_dtype(self.cfg["dtype"]))
        lib().mco_model_get_hidden(self._h, C.c_int32(layer), _p(out))
        return out

    def kv(self, layer: int):
        c = self.cfg
        shape = (c["max_seq_len"], c["n_kv_heads"], c["head_dim"])
        k = np.zeros(shape, dtype=np_dtype(c["dtype"]))
        v = np.zeros(shape, dtype=np_dtype(c["dtype"]))
        n = lib().mco_model_get_kv(self._h, C.c_int32(layer), _p(k), _p(v))
        return k[:n], v[:n]

    def set_kv(self, layer: int, keys: np.ndarray, values: np.ndarray):
        """Test aid: logical rows [n, n_kv, hd] of T become positions 0 .. n-1 of `layer`'s cache."""
        k = np.ascontiguousarray(keys, dtype=np_dtype(self.cfg["dtype"]))
        v = np.ascontiguousarray(values, dtype=np_dtype(self.cfg["dtype"]))
        n = lib().mco_model_set_kv(self._h, C.c_int32(layer), _p(k), _p(v), C.c_int32(k.shape[0]))
        assert n == k.shape[0]

    def close(self):
        if self._h:
            lib().mco_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_num_threads(n: int):
    lib().mco_set_num_threads(C.c_int(n))
