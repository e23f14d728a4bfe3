"""Reader of the per-wave timing experiment of round 4 (profiles/r4_forward_wave_timing.txt).  Needs a VARIANT library whose
tiles_forward_kernel stores, from lane 63 of every half-tile wave, uint4(s_memrealtime at entry (low word), high word | XCC_ID << 8,
s_memrealtime at exit (low word), list length) into the last lane's checkpoint record of its second quadrant -- twelve lines that
lived behind -DPS_WAVE_TIMING for the experiment and are not in the product source.  What bounds the forward: throughput, its
longest task, or the tail?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelsplat_amd.decoder import render_cuda
from pixelsplat_amd.raster import state_views
from pixelsplat_amd.synthetic import make_workload
dev = torch.device("cuda")
b, v, hw = 7, 4, (256, 256)
ctx, tgt, g, _ = make_workload(b, hw, v_ctx=2, v_tgt=v, seed=0, scene=sys.argv[1] if len(sys.argv) > 1 else "survey")
V = b * v
args = (tgt.extrinsics.reshape(V, 4, 4).to(dev), tgt.intrinsics.reshape(V, 3, 3).to(dev), tgt.near.reshape(V).to(dev),
        tgt.far.reshape(V).to(dev), hw, torch.zeros((V, 3), device=dev), g.means.to(dev), g.covariances.to(dev),
        g.harmonics.to(dev), g.opacities.to(dev))
img, aux = render_cuda(*args, views_per_scene=v, return_aux=True)          # exact list size
cap = int(aux["point_list"].numel() * 1.25)
for _ in range(3):                                                        # fixed capacity: one stream, no host sync
    torch.cuda.synchronize()
    img, aux = render_cuda(*args, views_per_scene=v, return_aux=True, list_capacity=cap)
torch.cuda.synchronize()
ck = state_views(aux["cfg"], aux["state"], aux["layout"])["checkpoint"].cpu().numpy().view(np.uint32)   # [V,T,4,64,4]
rec = ck[:, :, [1, 3], 63, :].reshape(-1, 4).astype(np.uint64)       # one record per half-tile wave
t0 = rec[:, 0] | ((rec[:, 1] & 0xFF) << 32)
t1 = rec[:, 2]
dur = ((t1 - (t0 & 0xFFFFFFFF)) & 0xFFFFFFFF).astype(np.float64)
hw_id = (rec[:, 1] >> 8).astype(np.int64)
n = rec[:, 3].astype(np.float64)
xcc = hw_id & 0xF
TICK_US = 0.01                                   # s_memrealtime: 100 MHz
st = (t0 - t0.min()).astype(np.float64) * TICK_US
du = dur * TICK_US
en = st + du
span = en.max()
print(f"waves {len(du)}; kernel span {span:.1f} us; longest wave {du.max():.1f} us = {du.max() / span:.2f} of the span; "
      f"mean wave {du.mean():.1f} us; waves in flight on average {du.sum() / span:.0f} of {256 * 4 * 6} slots")
edges = np.linspace(0, span, 21)
infl = [int(((st < edges[i + 1]) & (en > edges[i])).sum()) for i in range(20)]
print("waves in flight per 5 % slice of the span:".replace("%", "%%") % () , infl)
print("start times (us): p50 %.1f p90 %.1f p99 %.1f last %.1f" % tuple(np.quantile(st, [0.5, 0.9, 0.99, 1.0])))
print("per XCD: last wave ends at (us):", [round(float(en[xcc == x].max()), 1) for x in np.unique(xcc)])
k = np.argsort(-du)[:5]
print("longest waves (us, list length, start us):", [(round(float(du[i]), 1), int(n[i]), round(float(st[i]), 1)) for i in k])
tpe = du / np.maximum(n, 1) * 1e3
print("ns per list entry: p10 %.0f  median %.0f  p90 %.0f" % tuple(np.quantile(tpe, [0.1, 0.5, 0.9])))
q1, q3 = np.quantile(st, [0.25, 0.75])
print("ns per list entry of the first-started quarter %.0f, of the last-started quarter %.0f" %
      (np.median(tpe[st <= q1]), np.median(tpe[st >= q3])))
lastq = en > 0.9 * span
print("waves still running in the last 10 %% of the span: %d; their list lengths p50 %d (all waves: %d)" %
      (int(lastq.sum()), int(np.median(n[lastq])), int(np.median(n))))
