"""Import the *actual* reference (read-only mount /root/reference) on CPU.

TEST INFRASTRUCTURE, BUILD-CONTAINER ONLY.  /root/reference does not exist on the GPU
box; nothing under tests -m gpu, smoke() or bench.py may import this module.  It is
used by `tests/golden/make_golden.py` (to capture golden vectors from the reference's own
code) and by `tests/test_oracle_vs_reference.py` (skipped when the mount is absent).

Recipe: the reference imports open3d/plyfile/skimage/addict at module
top (`wild_completion/utils.py:14-20`) and hard-codes `.cuda()` (`loss.py:33,55,...`),
so we register empty stub modules and make `.cuda()` the identity before importing.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "wild_completion"))


def import_reference():
    """Returns a namespace with the reference's hot-path callables (CPU)."""
    if not available():
        raise RuntimeError("reference mount not present")
    import torch
    sys.dont_write_bytecode = True
    for n in ["open3d", "plyfile", "skimage", "skimage.measure", "addict"]:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["addict"].Dict = dict
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    torch.cuda.synchronize = lambda *a, **k: None
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from deepsdf.networks.deep_sdf_decoder import Decoder
    import wild_completion.loss as loss
    import wild_completion.utils as utils
    import wild_completion.optimizer as optimizer
    ns = types.SimpleNamespace(Decoder=Decoder, loss=loss, utils=utils, optimizer=optimizer)
    return ns


def build_reference_decoder(ns, params):
    """Instantiate the reference `Decoder` with the shipped NetworkSpecs
    (`deepsdf/models/sweetpepper_32/specs.json:7-18`) and load a parameter dict made by
    `hortimapping_amd.synthetic.make_synthetic_decoder`."""
    import torch
    L, H = int(params["latent_dim"]), int(params["hidden"])
    dec = ns.Decoder(L, dims=[H] * 8, dropout=list(range(8)), dropout_prob=0.2,
                     norm_layers=list(range(8)), latent_in=[4], xyz_in_all=False,
                     use_tanh=False, latent_dropout=False, weight_norm=True)
    sd = {}
    for k, v in params.items():
        if k in ("latent_dim", "hidden"):
            continue
        sd[k] = torch.from_numpy(v.copy())
    missing = dec.load_state_dict(sd, strict=True)
    dec.eval()
    return dec
