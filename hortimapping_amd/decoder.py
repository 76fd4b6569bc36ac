"""Decoder weight bundle: the drop-in for the reference's `decoder` argument.

Reference: `deepsdf/deep_sdf/workspace.py:203-225` (config_decoder: specs.json -> Decoder -> load_state_dict with
`module.`-prefixed keys) and `deepsdf/networks/deep_sdf_decoder.py:29-72` (9 Linear layers, weight-norm on
lin0..lin7, skip concat at layer 4).  The MI355X build folds weight-norm once at load and hands the nine fp32
matrices to `hm_decoder_create`, which packs them for the MFMA kernels."""
from __future__ import annotations

import ctypes
import json
import os

import numpy as np

from . import _lib

N_LIN = 9


def fold_state_dict(sd) -> tuple:
    """(Ws, bs): folded fp32 numpy arrays from a state dict with keys `[module.]lin{l}.weight_g/_v|weight/bias`."""
    def get(name):
        for pre in ("", "module."):
            if pre + name in sd:
                v = sd[pre + name]
                return np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        return None
    norm_keys = [k for k in sd if any(part.startswith("bn") and part[2:].isdigit() for part in k.split("."))]
    if norm_keys:
        # Decoder(weight_norm=False, norm_layers=[...]) inserts nn.LayerNorm modules `bn{l}` between Linear and ReLU
        # (deep_sdf_decoder.py:57-62, 96-99): folding only the Linear layers would evaluate a different network.
        raise NotImplementedError("LayerNorm ('bn*') parameters found (%s ...): only weight-normalised or plain "
                                  "Linear stacks are supported" % norm_keys[0])
    Ws, bs = [], []
    for l in range(N_LIN):
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
    return Ws, bs


class DecoderWeights:
    """Owns an `hm_decoder_t` handle (device-resident packed weights)."""

    def __init__(self, Ws, bs, latent_dim: int):
        self.latent_dim = int(latent_dim)
        L = self.latent_dim
        m = 512 - (L + 3)
        shapes = [(512, L + 3), (512, 512), (512, 512), (m, 512)] + [(512, 512)] * 4 + [(1, 512)]
        for l, (w, shp) in enumerate(zip(Ws, shapes)):
            if tuple(w.shape) != shp:
                raise ValueError(f"lin{l}: expected shape {shp}, got {tuple(w.shape)} "
                                 "(only the shipped 8x512, latent_in=[4] architecture is supported)")
        self.Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in Ws]
        self.bs = [np.ascontiguousarray(b, dtype=np.float32) for b in bs]
        lib = _lib.lib()
        Wp = (_lib.c_float_p * N_LIN)(*[w.ctypes.data_as(_lib.c_float_p) for w in self.Ws])
        bp = (_lib.c_float_p * N_LIN)(*[b.ctypes.data_as(_lib.c_float_p) for b in self.bs])
        h = ctypes.c_void_p()
        _lib.check(lib.hm_decoder_create(L, Wp, bp, ctypes.byref(h)), "hm_decoder_create")
        self.handle = h
        default = os.environ.get("HM_PRECISION", "")
        if default:
            self.set_precision(default)

    PRECISIONS = {"f32": 0, "f16x3": 1, "f16x3f_f16b": 2, "f16": 3}

    def set_precision(self, name: str):
        """'f32' (exact fp32 MFMA, default), 'f16x3' (fp16 MFMA, hi/lo split operands, ~2^-22 relative) or the
        mixed 'f16x3f_f16b' (forward as f16x3, backward in one fp16 pass: Jacobians ~1e-3 relative, NOT fp32-class) or
        'f16' (plain fp16 MFMA decoder of BASELINE.json configs[4]: everything ~1e-3 relative)."""
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
        tw = getattr(self, "_f32_twin", None)
        if tw is None:
            env = os.environ.pop("HM_PRECISION", None)          # the twin is f32 whatever the process default says
            try:
                tw = DecoderWeights(self.Ws, self.bs, self.latent_dim)
            finally:
                if env is not None:
                    os.environ["HM_PRECISION"] = env
            self._f32_twin = tw
        return tw

    @classmethod
    def from_params(cls, params):
        """`params`: dict of lin{l}.weight_v/weight_g/bias (+ lin8.weight) arrays plus 'latent_dim'."""
        Ws, bs = fold_state_dict({k: v for k, v in params.items() if k not in ("latent_dim", "hidden")})
        return cls(Ws, bs, int(params["latent_dim"]))

    @classmethod
    def from_module(cls, module):
        """Accept the reference's `Decoder` nn.Module (or a DataParallel wrapper of it)."""
        sd = module.state_dict()
        Ws, bs = fold_state_dict(sd)
        return cls(Ws, bs, Ws[0].shape[1] - 3)

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
    if list(ns["dims"]) != [512] * 8 or list(ns["latent_in"]) != [4] or ns.get("xyz_in_all") or ns.get("use_tanh"):
        raise NotImplementedError("only the shipped 8x512, latent_in=[4] DeepSDF architecture is supported")
    if ns.get("norm_layers") and not ns.get("weight_norm", False):
        raise NotImplementedError("norm_layers without weight_norm inserts LayerNorm modules "
                                  "(deep_sdf_decoder.py:57-62): not supported")
    saved = torch.load(os.path.join(experiment_directory, "ModelParameters", checkpoint + ".pth"),
                       map_location="cpu")
    Ws, bs = fold_state_dict(saved["model_state_dict"])
    return DecoderWeights(Ws, bs, int(specs["CodeLength"]))


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
