"""Decoder weight bundle: the drop-in for the reference's `decoder` argument.

Reference: `deepsdf/deep_sdf/workspace.py:203-225` (config_decoder: specs.json -> Decoder -> load_state_dict with
`module.`-prefixed keys) and `deepsdf/networks/deep_sdf_decoder.py:11-72` (the layer table: `dims`, `latent_in`,
`xyz_in_all`, `norm_layers` with / without `weight_norm`, `use_tanh`).  The MI355X build folds weight-norm once at load.
The shipped architecture (9 Linear layers, 8 x 512, skip concat at layer 4, weight norm) goes to `hm_decoder_create`,
which packs the matrices for the specialised MFMA kernels (exact f32, f16x3, f16); every other layer table the reference
class can build goes to `hm_decoder_create_arch` (exact-f32 MFMA kernel of `csrc/hm_decoder_any.hip`)."""
from __future__ import annotations

import ctypes
import json
import os
import threading
from typing import Optional

import numpy as np

from . import _lib

N_LIN = 9            # Linear layers of the shipped architecture
MAX_LIN = 16         # HM_MAX_LIN of include/hortimapping_amd.h
MAX_WIDTH = 512


def fold_state_dict_full(sd) -> tuple:
    """(Ws, bs, ln): folded fp32 numpy arrays from a state dict with keys `[module.]lin{l}.weight_g/_v|weight/bias`
    for l = 0 .. n-1 (n found from the keys) and `ln` = {l: (weight, bias)} of the `bn{l}` LayerNorm modules that
    `Decoder(weight_norm=False, norm_layers=[...])` inserts between Linear and ReLU (deep_sdf_decoder.py:57-62, 96-101)."""
    def get(name):
        for pre in ("", "module."):
            if pre + name in sd:
                v = sd[pre + name]
                return np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        return None
    n = 0
    while get(f"lin{n}.bias") is not None or get(f"lin{n}.weight") is not None or get(f"lin{n}.weight_v") is not None:
        n += 1
    if n == 0:
        raise KeyError("lin0.weight[_v] missing from state dict")
    Ws, bs, ln = [], [], {}
    for l in range(n):
        v = get(f"lin{l}.weight_v")
        if v is not None:
            g = get(f"lin{l}.weight_g").reshape(-1, 1)
            nrm = np.sqrt((v * v).sum(axis=1, keepdims=True, dtype=np.float32))
            w = v * (g / nrm)
        else:
            w = get(f"lin{l}.weight")
            if w is None:
                raise KeyError(f"lin{l}.weight[_v] missing from state dict")
        Ws.append(np.ascontiguousarray(w, dtype=np.float32))
        bs.append(np.ascontiguousarray(get(f"lin{l}.bias"), dtype=np.float32))
        gw = get(f"bn{l}.weight")
        if gw is not None and l < n - 1:        # a bn on the last layer exists in __init__ but forward never applies it (:96)
            ln[l] = (np.ascontiguousarray(gw), np.ascontiguousarray(get(f"bn{l}.bias"), dtype=np.float32))
    return Ws, bs, ln


def fold_state_dict(sd) -> tuple:
    """(Ws, bs) of a checkpoint WITHOUT LayerNorm modules (the shipped models); see `fold_state_dict_full`."""
    norm_keys = [k for k in sd if any(part.startswith("bn") and part[2:].isdigit() for part in k.split("."))]
    if norm_keys:
        raise NotImplementedError("LayerNorm ('bn*') parameters found (%s ...): fold_state_dict returns Linear layers "
                                  "only -- use fold_state_dict_full" % norm_keys[0])
    Ws, bs, _ = fold_state_dict_full(sd)
    return Ws, bs


def layer_table(Ws, latent_dim: int, ln=None, use_tanh: bool = False) -> dict:
    """The layer table of `Decoder.__init__` / `.forward` recovered from the folded matrices: layer l's input is the
    previous output, with [z | xyz] appended where the widths differ by latent_dim + 3 (`latent_in`,
    deep_sdf_decoder.py:41-42, 87-88) or xyz appended where they differ by 3 (`xyz_in_all`, :45-46, 89-90)."""
    D0, n = latent_dim + 3, len(Ws)
    in_dim, out_dim, cat = [], [], []
    for l, w in enumerate(Ws):
        od, idim = int(w.shape[0]), int(w.shape[1])
        if l == 0:
            if idim != D0:
                raise ValueError(f"lin0 takes {idim} inputs, expected latent_dim + 3 = {D0}")
            c = 0
        else:
            extra = idim - out_dim[-1]
            if extra not in (0, 3, D0):
                raise ValueError(f"lin{l}: {idim} inputs after a layer of {out_dim[-1]} outputs is neither a plain, an "
                                 f"xyz_in_all (+3) nor a latent_in (+{D0}) connection")
            c = {0: 0, D0: 1, 3: 2}[extra]
        in_dim.append(idim); out_dim.append(od); cat.append(c)
    if out_dim[-1] != 1:
        raise ValueError("the last Linear layer must have one output")
    lnl = [1 if (ln and l in ln) else 0 for l in range(n)]
    return {"latent_dim": int(latent_dim), "n_lin": n, "use_tanh": bool(use_tanh), "in_dim": in_dim, "out_dim": out_dim,
            "cat": cat, "layer_norm": lnl}


def is_shipped_table(t: dict) -> bool:
    L = t["latent_dim"]
    m = 512 - (L + 3)
    return (t["n_lin"] == N_LIN and not t["use_tanh"] and not any(t["layer_norm"]) and
            t["out_dim"] == [512, 512, 512, m, 512, 512, 512, 512, 1] and t["cat"] == [0, 0, 0, 0, 1, 0, 0, 0, 0])


class HmDecoderArch(ctypes.Structure):
    """ctypes mirror of `hm_decoder_arch` (include/hortimapping_amd.h)."""
    _fields_ = [("latent_dim", ctypes.c_int), ("n_lin", ctypes.c_int), ("use_tanh", ctypes.c_int),
                ("in_dim", ctypes.c_int * MAX_LIN), ("out_dim", ctypes.c_int * MAX_LIN),
                ("cat", ctypes.c_int * MAX_LIN), ("layer_norm", ctypes.c_int * MAX_LIN)]


class DecoderWeights:
    """Owns an `hm_decoder_t` handle (device-resident packed weights)."""

    def __init__(self, Ws, bs, latent_dim: int, ln=None, use_tanh: bool = False, force_generic: bool = False,
                 precision: Optional[str] = None):
        """`precision`: None = the process default (HM_PRECISION, else exact f32); a name = that arithmetic whatever the
        environment says (what `f32_twin` passes: it must not touch process-global state)."""
        self._twin_lock = threading.Lock()
        self.latent_dim = int(latent_dim)
        L = self.latent_dim
        self.Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in Ws]
        self.bs = [np.ascontiguousarray(b, dtype=np.float32).reshape(-1) for b in bs]
        self.ln = {int(l): (np.ascontiguousarray(g, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32))
                   for l, (g, b) in (ln or {}).items()}
        self.use_tanh = bool(use_tanh)
        self.table = layer_table(self.Ws, L, self.ln, self.use_tanh)
        # force_generic: run the SHIPPED table on the any-architecture kernel too (parity / timing A-B against the
        # specialised kernels; tests/test_gpu_arch.py, scripts/time_arch_decoder.py)
        self.generic = force_generic or not is_shipped_table(self.table)
        lib = _lib.lib()
        n = len(self.Ws)
        h = ctypes.c_void_p()
        if not self.generic:
            Wp = (_lib.c_float_p * n)(*[w.ctypes.data_as(_lib.c_float_p) for w in self.Ws])
            bp = (_lib.c_float_p * n)(*[b.ctypes.data_as(_lib.c_float_p) for b in self.bs])
            _lib.check(lib.hm_decoder_create(L, Wp, bp, ctypes.byref(h)), "hm_decoder_create")
        else:
            if n > MAX_LIN:
                raise NotImplementedError(f"{n} Linear layers: at most {MAX_LIN} are supported")
            if max(max(self.table["in_dim"]), max(self.table["out_dim"])) > MAX_WIDTH:
                raise NotImplementedError(f"layer widths above {MAX_WIDTH} are not supported "
                                          f"(in {self.table['in_dim']}, out {self.table['out_dim']})")
            for l, (w, b) in enumerate(zip(self.Ws, self.bs)):
                if b.shape[0] != w.shape[0]:
                    raise ValueError(f"lin{l}: bias of {b.shape[0]} for {w.shape[0]} outputs")
            arch = HmDecoderArch()
            arch.latent_dim, arch.n_lin, arch.use_tanh = L, n, int(self.use_tanh)
            for l in range(n):
                arch.in_dim[l], arch.out_dim[l] = self.table["in_dim"][l], self.table["out_dim"][l]
                arch.cat[l], arch.layer_norm[l] = self.table["cat"][l], self.table["layer_norm"][l]
            null = ctypes.cast(None, _lib.c_float_p)
            Wp = (_lib.c_float_p * MAX_LIN)(*[w.ctypes.data_as(_lib.c_float_p) for w in self.Ws])
            bp = (_lib.c_float_p * MAX_LIN)(*[b.ctypes.data_as(_lib.c_float_p) for b in self.bs])
            gp = (_lib.c_float_p * MAX_LIN)(*[self.ln[l][0].ctypes.data_as(_lib.c_float_p) if l in self.ln else null
                                              for l in range(n)])
            ep = (_lib.c_float_p * MAX_LIN)(*[self.ln[l][1].ctypes.data_as(_lib.c_float_p) if l in self.ln else null
                                              for l in range(n)])
            _lib.check(lib.hm_decoder_create_arch(ctypes.byref(arch), Wp, bp, gp, ep, ctypes.byref(h)),
                       "hm_decoder_create_arch")
        self.handle = h
        if precision is not None:
            self.set_precision(precision)
        else:
            default = os.environ.get("HM_PRECISION", "")
            if default and (not self.generic or default in ("f32", "f16x3")):   # any-architecture handles: f32 and f16x3 only
                self.set_precision(default)

    PRECISIONS = {"f32": 0, "f16x3": 1, "f16x3f_f16b": 2, "f16": 3}

    def set_precision(self, name: str):
        """'f32' (exact fp32 MFMA, default), 'f16x3' (fp16 MFMA, hi/lo split operands, ~2^-22 relative) or the
        mixed 'f16x3f_f16b' (forward as f16x3, backward in one fp16 pass: Jacobians ~1e-3 relative, NOT fp32-class) or
        'f16' (plain fp16 MFMA decoder of BASELINE.json configs[4]: everything ~1e-3 relative).  A decoder of a
        non-shipped layer table (`self.generic`) has 'f32' and 'f16x3' kernels: the other two names are refused by the library."""
        _lib.check(_lib.lib().hm_decoder_set_precision(self.handle, self.PRECISIONS[name]), "hm_decoder_set_precision")
        return self

    @property
    def precision(self) -> str:
        v = _lib.lib().hm_decoder_get_precision(self.handle)
        return {v_: k for k, v_ in self.PRECISIONS.items()}[v]

    def f32_twin(self) -> "DecoderWeights":
        """A second handle on the same weights fixed at exact fp32 (created on first use, ~30 MB of HBM).  The exact-f32
        fallbacks (`optimize_batch(retry_f32=True)`, `MeshExtractor.decode_grids`) run on it instead of switching THIS
        handle's precision, which is a host field read at enqueue time: flipping it would silently change the arithmetic
        of any other thread using the decoder in that window."""
        if self.precision == "f32":
            return self
        with self._twin_lock:                       # two threads asking at once share ONE twin
            tw = getattr(self, "_f32_twin", None)
            if tw is None:
                tw = DecoderWeights(self.Ws, self.bs, self.latent_dim, self.ln, self.use_tanh, self.generic,
                                    precision="f32")   # f32 whatever the process default says
                self._f32_twin = tw
        return tw

    @classmethod
    def from_params(cls, params, use_tanh: bool = False, force_generic: bool = False):
        """`params`: dict of lin{l}.weight_v/weight_g/bias (or lin{l}.weight; + bn{l}.weight/bias) arrays plus 'latent_dim'."""
        Ws, bs, ln = fold_state_dict_full({k: v for k, v in params.items() if k not in ("latent_dim", "hidden", "use_tanh")})
        return cls(Ws, bs, int(params["latent_dim"]), ln, bool(use_tanh or params.get("use_tanh", False)), force_generic)

    @classmethod
    def from_module(cls, module):
        """Accept the reference's `Decoder` nn.Module (or a DataParallel wrapper of it), whatever its layer table."""
        inner = getattr(module, "module", module)
        Ws, bs, ln = fold_state_dict_full(module.state_dict())
        if getattr(inner, "weight_norm", False):
            ln = {}                                  # deep_sdf_decoder.py:57-62: bn modules only without weight_norm
        return cls(Ws, bs, Ws[0].shape[1] - 3, ln, bool(getattr(inner, "use_tanh", False)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().hm_decoder_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def config_decoder(experiment_directory: str, checkpoint: str = "latest") -> DecoderWeights:
    """Mirror of `deepsdf/deep_sdf/workspace.py:203-225`: specs.json + ModelParameters/<ckpt>.pth."""
    import torch
    specs_filename = os.path.join(experiment_directory, "specs.json")
    if not os.path.isfile(specs_filename):
        raise Exception('The experiment directory does not include specifications file "specs.json"')
    specs = json.load(open(specs_filename))
    ns = specs["NetworkSpecs"]
    saved = torch.load(os.path.join(experiment_directory, "ModelParameters", checkpoint + ".pth"),
                       map_location="cpu")
    Ws, bs, ln = fold_state_dict_full(saved["model_state_dict"])
    if ns.get("weight_norm", False):
        ln = {}
    dec = DecoderWeights(Ws, bs, int(specs["CodeLength"]), ln, bool(ns.get("use_tanh", False)))
    # the layer table recovered from the checkpoint must be the one specs.json asks Decoder.__init__ for
    want = specs_layer_table(int(specs["CodeLength"]), ns)
    for key in ("in_dim", "out_dim", "cat", "layer_norm"):
        if dec.table[key] != want[key]:
            raise ValueError(f"checkpoint does not match specs.json: {key} {dec.table[key]} != {want[key]}")
    return dec


def specs_layer_table(latent_size: int, ns: dict) -> dict:
    """`Decoder(latent_size, **NetworkSpecs)`'s layer table, restated from deep_sdf_decoder.py:29-62 and :85-102."""
    dims = [latent_size + 3] + list(ns["dims"]) + [1]
    num_layers = len(dims)
    latent_in = tuple(ns.get("latent_in", ()))
    norm_layers = ns.get("norm_layers", ())
    xyz_in_all = bool(ns.get("xyz_in_all"))
    in_dim, out_dim, cat, lnl = [], [], [], []
    for layer in range(num_layers - 1):
        if layer + 1 in latent_in:
            od = dims[layer + 1] - dims[0]
        else:
            od = dims[layer + 1]
            if xyz_in_all and layer != num_layers - 2:
                od -= 3
        in_dim.append(dims[layer]); out_dim.append(od)
        cat.append(1 if (layer in latent_in) else (2 if (layer != 0 and xyz_in_all) else 0))
        lnl.append(1 if ((not ns.get("weight_norm", False)) and norm_layers is not None and layer in norm_layers
                         and layer < num_layers - 2) else 0)
    if cat[0] != 0:
        raise ValueError("latent_in containing layer 0 does not build a runnable reference Decoder")
    return {"latent_dim": latent_size, "n_lin": num_layers - 1, "use_tanh": bool(ns.get("use_tanh", False)),
            "in_dim": in_dim, "out_dim": out_dim, "cat": cat, "layer_norm": lnl}


def load_latent_vectors(experiment_directory: str, checkpoint: str = "latest"):
    """Mirror of `deepsdf/deep_sdf/workspace.py:82-114`: returns the (n, L) latent matrix (CPU tensor)."""
    import torch
    filename = os.path.join(experiment_directory, "LatentCodes", checkpoint + ".pth")
    if not os.path.isfile(filename):
        raise Exception(f"The experiment directory ({experiment_directory}) does not include a latent code file"
                        f" for checkpoint '{checkpoint}'")
    data = torch.load(filename, map_location="cpu")
    lc = data["latent_codes"]
    if isinstance(lc, torch.Tensor):
        return lc.detach()
    return lc["weight"].detach()
