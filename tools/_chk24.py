import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from avlmaps_amd import ops
torch.manual_seed(0)
N, D = 512, 512
feat = torch.randn((N, D), device="cuda") * 3
q = torch.zeros((64, D), device="cuda")
q[torch.arange(64), torch.arange(64)] = 1.0
pm = ops.prepare_map(feat, compact=True)
sc, _, _ = ops.sim_scores(pm, q)
x = feat[:, :64]
rel = ((sc - x).abs() / x.abs().clamp_min(1e-20))
print("median rel", rel.median().item(), "max", rel.max().item())
print(["%d:%.0e" % (i, v) for i, v in enumerate(rel.median(dim=0).values.tolist())][:32])
# magnitude dependence
for lo_, hi_ in ((0, 0.01), (0.01, 0.1), (0.1, 1), (1, 3), (3, 20)):
    m = (x.abs() >= lo_) & (x.abs() < hi_)
    if m.any():
        print(f"|x| in [{lo_},{hi_}): median rel {rel[m].median().item():.2e} max {rel[m].max().item():.2e} n={int(m.sum())}")
buf = pm.feat.cpu().numpy().reshape(N, D // 32, 96)
rs = pm.row_scale.cpu().numpy()
hi = buf[:, :, :64].copy().view(np.float16).reshape(N, D)
u = buf[:, :, 64:].reshape(N, D).astype(np.int32)
eb = (hi.view(np.uint16) >> 10) & 31
lo = np.where(eb > 18, (u - 128) * np.exp2(eb.astype(np.float64) - 33), 0.0)
xs = feat.cpu().numpy().astype(np.float64) / rs[:, None]
rec = hi.astype(np.float64) + lo
r2 = np.abs(rec - xs) / np.maximum(np.abs(xs), 1e-30)
print("host decode of the buffer: median rel", np.median(r2), "p99", np.quantile(r2, 0.99), "max", r2.max())
