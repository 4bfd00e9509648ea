"""Adapter to the UPSTREAM LSeg model (stays on PyTorch-ROCm; not re-implemented here).

Needs the upstream `avlmaps` package, its dependencies (timm, clip, encoding, ...) and the demo_e200.ckpt checkpoint
importable on the machine.  Returns a callable rgb(H,W,3 uint8) -> (Hf, Wf, D) float32 CUDA tensor, channels-last, kept
on the device -- the reference instead copies a (1, D, Hf, Wf) array to the host every frame (lseg_utils.py:101-102)."""
from __future__ import annotations


def load_upstream_lseg(checkpoint_path=None):
    import torch
    import torchvision.transforms as transforms
    from avlmaps.lseg.modules.models.lseg_net import LSegEncNet          # upstream package
    from avlmaps.utils import lseg_utils                                  # upstream sliding-window evaluation

    device = "cuda"
    model = LSegEncNet("", arch_option=0, block_depth=0, activation="lrelu", crop_size=480)
    if checkpoint_path is None:
        from pathlib import Path
        import avlmaps
        checkpoint_path = Path(avlmaps.__file__).resolve().parent / "lseg" / "checkpoints" / "demo_e200.ckpt"
    sd = torch.load(checkpoint_path, map_location=device)["state_dict"]
    model.load_state_dict({k.lstrip("net."): v for k, v in sd.items()})
    model = model.eval().to(device)
    tfm = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.5] * 3, [0.5] * 3)])

    def extract(rgb):
        f = lseg_utils.get_lseg_feat(model, rgb, ["example"], tfm, device, 480, 520, [0.5] * 3, [0.5] * 3)
        return torch.from_numpy(f[0]).to(device).permute(1, 2, 0).contiguous()

    return extract
