"""How much of the tile backward's walk qualifies for its short form?  (GPU box; BASELINE configs[1] scene)

The short form of tiles_backward (raster_tiles.hip, stage_batch<FAST>) needs, for a finalisation batch, (a) every
entry from the batch to the ring's tail to be plain (entry_is_plain) and (b) the batch's first entry at or before
EVERY pixel's last contributor (top_h <= nc_min).  This prints, from the forward's saved state, the distribution
of nc_min / walk end per tile and the fraction of entries that are not plain.
    python tools/fast_form_probe.py [--scene survey]
"""
import argparse, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelsplat_amd.synthetic import make_workload
from pixelsplat_amd.decoder import render_cuda
from pixelsplat_amd.raster import state_views

ap = argparse.ArgumentParser(); ap.add_argument("--scene", default="survey"); a = ap.parse_args()
dev = torch.device("cuda:0")
b, v, hw = 7, 4, (256, 256)
ctx, tgt, g, _ = make_workload(b, hw, v_ctx=2, v_tgt=v, seed=0, scene=a.scene)
V = b * v
t = lambda x: x.to(dev)
img, aux = render_cuda(t(tgt.extrinsics.reshape(V, 4, 4)), t(tgt.intrinsics.reshape(V, 3, 3)), t(tgt.near.reshape(V)),
                       t(tgt.far.reshape(V)), hw, torch.zeros(3, device=dev), t(g.means), t(g.covariances),
                       t(g.harmonics), t(g.opacities), views_per_scene=v, return_aux=True)
sv = state_views(aux["cfg"], aux["state"], aux["layout"])
H, W = hw
nc = sv["n_contrib"].reshape(V, H // 16, 16, W // 16, 16).permute(0, 1, 3, 2, 4).reshape(V, -1, 256).long()
T = sv["final_T"].reshape(V, H // 16, 16, W // 16, 16).permute(0, 1, 3, 2, 4).reshape(V, -1, 256)
cnt = sv["tile_ranges"][..., 1].long(); end = sv["tile_end"].long()
ncmin = nc.min(-1).values; ncmax = nc.max(-1).values
print("tiles", cnt.numel(), "mean list", cnt.float().mean().item(), "mean walk end (tile_end)", end.float().mean().item())
r = (ncmin.float() / end.clamp(min=1).float())
q = torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], device=dev)
print("nc_min / walk end quantiles", torch.quantile(r.flatten(), q).tolist())
print("entries at or before nc_min / entries walked:", (ncmin.sum() / end.sum()).item())
print("tiles with a pixel without contributors:", (ncmin == 0).float().mean().item())
# per-pixel: how many pixels end early (n_contrib below the tile's walk end by more than 10 %)
print("pixels with n_contrib < 0.5 walk end:", (nc.float() < 0.5 * end[..., None].float()).float().mean().item())
print("final_T quantiles", torch.quantile(T.flatten()[::7], q).tolist())
rec = sv["records"]; vis = aux["radii"].reshape(V, -1) > 0
k = 1.4426950408889634
A = -0.5 * k * rec[..., 2]; B = -k * rec[..., 3]; C = -0.5 * k * rec[..., 4]; o = rec[..., 5]
det = 4 * A * C - B * B; tr = A + C
plain = (A < 0) & (C < 0) & (det > 1e-4 * tr * tr) & (o <= 0.98 * 0.99) & (o >= 0)
print("visible pairs", int(vis.sum()), "not plain:", (~plain & vis).sum().item() / vis.sum().item())
print("  of which opacity:", ((o > 0.98 * 0.99) & vis).sum().item() / vis.sum().item(),
      " conditioning:", (~(det > 1e-4 * tr * tr) & vis).sum().item() / vis.sum().item())
