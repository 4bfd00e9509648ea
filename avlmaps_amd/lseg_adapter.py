"""Adapter to the UPSTREAM LSeg model (stays on PyTorch-ROCm; not re-implemented here).

Needs the upstream `avlmaps` package, its dependencies (timm, clip, encoding, ...) and the demo_e200.ckpt checkpoint
importable on the machine.  Returns a callable rgb(H,W,3 uint8) -> (Hf, Wf, D) float32 CUDA tensor, channels-last, kept
on the device (avlmaps_amd.utils.lseg_utils.get_lseg_feat) -- the reference instead copies a (1, D, Hf, Wf) array to the
host every frame (lseg_utils.py:101-102)."""
from __future__ import annotations


def load_upstream_lseg(checkpoint_path=None):
    import torch
    from avlmaps.lseg.modules.models.lseg_net import LSegEncNet          # upstream package (model definition only)
    from .utils.lseg_utils import default_transform, get_lseg_feat        # sliding-window evaluation kept on the device

    device = "cuda"
    model = LSegEncNet("", arch_option=0, block_depth=0, activation="lrelu", crop_size=480)
    if checkpoint_path is None:
        from pathlib import Path
        import avlmaps
        checkpoint_path = Path(avlmaps.__file__).resolve().parent / "lseg" / "checkpoints" / "demo_e200.ckpt"
    sd = torch.load(checkpoint_path, map_location=device)["state_dict"]
    model.load_state_dict({k.lstrip("net."): v for k, v in sd.items()})
    model = model.eval().to(device)

    def extract(rgb):
        # (Hf, Wf, D) float32 CUDA tensor, channels-last; never leaves the GPU
        return get_lseg_feat(model, rgb, ["example"], default_transform, device, 480, 520, [0.5] * 3, [0.5] * 3)

    return extract
