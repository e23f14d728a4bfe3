"""CPU checks of two round-4 additions, with the oracle as the only heavy lifter (test infrastructure):

* the scene distributions of pixelsplat_amd/synthetic.py have the statistics bench.py --scene and
  DESIGN.md 13 say they have (measured with oracle/raster_ref.c on one scene, all four target views);
* the algebra behind the tile backward's two tasks per tile: started from the state the forward leaves at a
  list's split point -- the transmittance T_s after the first half and (C_final - C_s) / T_s, the colour
  composited behind it over that transmittance -- the backward recurrence of SURVEY.md Appendix A.4 gives
  the same per-entry gradients as the walk from the end of the list."""
import numpy as np
import pytest

from oracle import raster_ref as R
from pixelsplat_amd.synthetic import SCENES, make_workload
from tests.cases import oracle_view_inputs


@pytest.fixture(scope="module")
def scene_stats():
    out = {}
    for scene in SCENES:
        _, tgt, g, _ = make_workload(1, (256, 256), v_ctx=2, v_tgt=4, seed=0, scene=scene)
        G = g.means.shape[1]
        D = vis = big = 0
        ends = []
        for v in range(4):
            st = R.forward(H=256, W=256, **oracle_view_inputs(g, tgt, 0, v))
            D += st.num_rendered
            vis += int((st.radii > 0).sum())
            big += int((st.tiles_touched > 4).sum())
            cnt = (st.ranges[:, 1] - st.ranges[:, 0]).reshape(16, 16)
            tl = np.repeat(np.repeat(cnt, 16, 0), 16, 1).reshape(-1)
            ends.append(np.median(st.n_contrib.reshape(-1) / np.maximum(tl, 1)))
        out[scene] = dict(visible=vis / (4 * G), D_over_GV=D / (4 * G), large=big / max(vis, 1),
                          walk_end=float(np.median(ends)))
    return out


def test_scene_variants_share_the_cameras_and_random_draws():
    base = make_workload(1, (64, 64), seed=3)
    for scene in SCENES[1:]:
        other = make_workload(1, (64, 64), seed=3, scene=scene)
        assert np.array_equal(base[1].extrinsics.numpy(), other[1].extrinsics.numpy())   # target cameras
        assert np.array_equal(base[3].numpy(), other[3].numpy())                         # target images
        assert np.array_equal(base[2].harmonics.numpy(), other[2].harmonics.numpy())     # colours
    with pytest.raises(ValueError):
        make_workload(1, (64, 64), scene="nope")


def test_survey_scene_is_the_survey_recipe(scene_stats):
    s = scene_stats["survey"]
    assert 0.35 < s["visible"] < 0.5 and 1.1 < s["D_over_GV"] < 1.6 and s["large"] < 0.05
    assert s["walk_end"] > 0.85              # pixels end near the end of their lists (0.99 over the bench batch)


def test_dense_scene_keeps_most_pairs_in_frame(scene_stats):
    s = scene_stats["dense"]
    assert s["visible"] >= 0.8 and s["D_over_GV"] >= 2.0
    assert s["walk_end"] < 0.6                                   # early termination carries weight


def test_opaque_scene_terminates_early(scene_stats):
    s, base = scene_stats["opaque"], scene_stats["survey"]
    assert abs(s["visible"] - base["visible"]) < 1e-9 and abs(s["D_over_GV"] - base["D_over_GV"]) < 1e-9
    assert s["walk_end"] < 0.3


def test_large_scene_takes_the_atomic_path(scene_stats):
    s = scene_stats["large"]
    assert s["large"] > 0.8 and s["D_over_GV"] > 3.5


# ---- the split backward, in numpy ------------------------------------------------------------------------
def _backward_segment(alpha, colour, g, T_start, acc_start, lo, hi):
    """SURVEY.md A.4 for entries lo <= i < hi of one pixel, walked back to front: returns dL/dalpha_i and
    dL/dcolour_i.  T_start = transmittance AFTER entry hi - 1, acc_start = colour composited behind it
    (relative to that transmittance).  (No background term: bg = 0.)"""
    T, acc = T_start, acc_start.copy()
    d_alpha = np.zeros(len(alpha))
    d_col = np.zeros_like(colour)
    for i in range(hi - 1, lo - 1, -1):
        T = T / (1.0 - alpha[i])                    # transmittance in front of entry i
        d_col[i] = alpha[i] * T * g
        d_alpha[i] = T * float(((colour[i] - acc) * g).sum())
        acc = alpha[i] * colour[i] + (1.0 - alpha[i]) * acc
    return d_alpha, d_col


def test_backward_started_from_the_forward_checkpoint_equals_the_walk_from_the_end():
    rng = np.random.default_rng(0)
    n, split = 300, 128
    alpha = rng.uniform(0.0, 0.35, n)
    colour = rng.uniform(0, 1, (n, 3))
    g = rng.normal(size=3)                           # dL/dC of the pixel
    # forward
    T = 1.0
    C = np.zeros(3)
    T_s = C_s = None
    for i in range(n):
        C = C + colour[i] * alpha[i] * T
        T = T * (1.0 - alpha[i])
        if i + 1 == split:
            T_s, C_s = T, C.copy()
    # one walk from the end
    da_full, dc_full = _backward_segment(alpha, colour, g, T, np.zeros(3), 0, n)
    # two tasks: the back half from (T_final, 0), the front half from the checkpoint
    da_b, dc_b = _backward_segment(alpha, colour, g, T, np.zeros(3), split, n)
    da_f, dc_f = _backward_segment(alpha, colour, g, T_s, (C - C_s) / T_s, 0, split)
    assert np.allclose(da_b[split:], da_full[split:], rtol=0, atol=0)          # the same arithmetic
    assert np.allclose(dc_b[split:], dc_full[split:], rtol=0, atol=0)
    assert np.allclose(da_f[:split], da_full[:split], rtol=1e-12, atol=1e-15)
    assert np.allclose(dc_f[:split], dc_full[:split], rtol=1e-12, atol=1e-15)


def test_split_point_is_a_refine_batch_boundary():
    # mirrors raster_common.h::split_point (the forward drains its ring after a whole batch of 64)
    def split_point(n, split_min=256):
        return ((n >> 1) & ~63) if n >= split_min else 0
    for n in (0, 1, 255, 256, 257, 1000, 2013, 3946, 65536):
        s = split_point(n)
        assert s % 64 == 0 and s <= n // 2 and (s == 0) == (n < 256)
        if s:
            assert n - s >= s               # the back task is never the shorter one by more than a batch
