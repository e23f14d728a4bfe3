"""The tile backward's per-pixel recurrence (csrc/raster_tiles.hip, round 6) against the reference's form, in fp64.

The reference backward (restated by oracle/raster_ref_impl.inc) keeps, per pixel, the 3-vector
`accum_rec` (colour composited behind the current entry over the transmittance in front of it) and adds a second
term for the background:

    dL/dalpha_i = T_i sum_ch (c_i - accum_rec_i) g  -  T_final / (1 - alpha_i) (bg . g)

The HIP kernel keeps ONE scalar per pixel, hb = (colour behind, background included) . g:

    hb starts at bg . g behind the pixel's last contributor,  e_i = c_i . g - hb,
    dL/dalpha_i = T_i e_i,   hb <- hb + alpha_i e_i                      (T_i = T_{i+1} / (1 - alpha_i), back to front)

and a task that starts in the middle of a list (the forward's checkpoint: T at the split, colour behind the split
over that T, background NOT included) adds bg T_final / T_split first.  Same algebra; this test holds the two forms
and a central finite difference of the composited colour together, so that the derivation in DESIGN.md 4a is executable.
"""
import numpy as np


def _forward(alpha, col, bg):
    T = 1.0
    C = np.zeros(3)
    for a, c in zip(alpha, col):
        C = C + c * a * T
        T = T * (1.0 - a)
    return C + bg * T, T


def _reference_form(alpha, col, bg, g):
    """Back to front as the reference walks: accum_rec, last_alpha / last_color, the background term apart."""
    _, T_final = _forward(alpha, col, bg)
    T = T_final
    accum = np.zeros(3)
    last_a, last_c = 0.0, np.zeros(3)
    out = np.zeros(len(alpha))
    for i in range(len(alpha) - 1, -1, -1):
        a, c = alpha[i], col[i]
        T = T / (1.0 - a)                       # transmittance in front of entry i
        accum = last_a * last_c + (1.0 - last_a) * accum
        d = float(np.dot(c - accum, g)) * T
        d += (-T_final / (1.0 - a)) * float(np.dot(bg, g))
        out[i] = d
        last_a, last_c = a, c
    return out


def _scalar_form(alpha, col, bg, g, start=None):
    """The kernel's recurrence.  `start` = (first index walked, T behind it, hb behind it): a task that begins at a
    checkpoint instead of at the pixel's last contributor."""
    _, T_final = _forward(alpha, col, bg)
    if start is None:
        hi, T, hb = len(alpha) - 1, T_final, float(np.dot(bg, g))
    else:
        hi, T, hb = start
    out = np.zeros(hi + 1)
    for i in range(hi, -1, -1):
        a, c = alpha[i], col[i]
        Tn = T / (1.0 - a)
        e = float(np.dot(c, g)) - hb
        out[i] = e * Tn
        hb = hb + a * e
        T = Tn
    return out


def test_scalar_state_equals_reference_form_and_finite_differences():
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 40):
        for bg in (np.zeros(3), np.array([0.2, 0.5, 0.7])):
            alpha = rng.uniform(0.004, 0.6, n)
            col = rng.uniform(0.0, 1.0, (n, 3))
            g = rng.normal(size=3)
            ref = _reference_form(alpha, col, bg, g)
            new = _scalar_form(alpha, col, bg, g)
            assert np.allclose(ref, new, rtol=1e-12, atol=1e-14)
            # central finite difference of (composited colour . g) with respect to every alpha
            for i in range(n):
                h = 1e-6
                ap, am = alpha.copy(), alpha.copy()
                ap[i] += h
                am[i] -= h
                fd = (np.dot(_forward(ap, col, bg)[0], g) - np.dot(_forward(am, col, bg)[0], g)) / (2 * h)
                assert abs(fd - new[i]) <= 1e-7 * max(1.0, abs(fd))


def test_task_starting_at_the_forward_checkpoint():
    """A list walked as two tasks: the front task starts from the forward's state at the split point -- T there and
    the colour behind the split over that T, WITHOUT the background -- and adds bg T_final / T_split itself."""
    rng = np.random.default_rng(5)
    n, split = 30, 12                       # entries 0 .. split-1 belong to the front task
    bg = np.array([0.3, 0.1, 0.9])
    alpha = rng.uniform(0.004, 0.4, n)
    col = rng.uniform(0.0, 1.0, (n, 3))
    g = rng.normal(size=3)
    C_total, T_final = _forward(alpha, col, bg)
    C_front, T_split = _forward(alpha[:split], col[:split], np.zeros(3))
    behind = (C_total - bg * T_final - C_front) / T_split      # what the forward leaves in state.checkpoint
    hb0 = float(np.dot(behind + bg * (T_final / T_split), g))
    whole = _scalar_form(alpha, col, bg, g)
    front = _scalar_form(alpha, col, bg, g, start=(split - 1, T_split, hb0))
    assert np.allclose(front, whole[:split], rtol=1e-11, atol=1e-13)
