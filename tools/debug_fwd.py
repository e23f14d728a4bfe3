"""Debug: where does the forward differ from the oracle (small scene 3/400/(64,48))?"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import raster_ref as R
from tests.cases import small_scene
from tests.test_raster_gpu import _single_view_hip
seed, n, hw = 3, 400, (64, 48)
sc = small_scene(n, hw, seed=seed, dtype=np.float32)
st = R.forward(dtype=np.float32, **sc)
dL = np.zeros((3,) + hw, np.float32)
img, radii, g = _single_view_hip(sc, torch.device("cuda"), dL)
err = np.abs(img - st.image).max(0)
bad = np.argwhere(err > 2e-5)
print("bad pixels", len(bad), "max err", err.max())
T = st.final_T.reshape(hw); nc = st.n_contrib.reshape(hw)
print("opacity max", sc["opacity"].max())
for y, x in bad[:40]:
    print(f"y {y} x {x} tile ({y//16},{x//16}) quad ({(y%16)//8},{(x%16)//8}) err {err[y,x]:.2e} oracle T {T[y,x]:.3e} n_contrib {nc[y,x]} list len {st.ranges[(y//16)*(hw[1]//16)+x//16][1]-st.ranges[(y//16)*(hw[1]//16)+x//16][0]}")
