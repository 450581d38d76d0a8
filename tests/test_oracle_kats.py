"""Known-answer tests pinning the oracle (no GPU): analytically derived facts about the
reference and the committed golden fixtures."""
import numpy as np
import pytest

import oracle_lib as O
from golden_util import CASES, GOLDEN, load_case, sha
from sift_amd.synthetic import synth_frame


def test_tap_tables():
    radii = {1.0: 3, 1.6: 5, 2.2627418: 7, 3.2: 10, 4.5254836: 14, 6.4: 19, 9.050967: 27, 12.8: 38, 18.101934: 54}
    g = np.load(f"{GOLDEN}/kats.npz")
    for i, (sigma, r) in enumerate(radii.items()):
        rr, t = O.gauss_taps(sigma)
        assert rr == r and t.size == 2 * r + 1
        assert (t == t[::-1]).all()                       # exactly symmetric
        assert abs(float(t.astype(np.float64).sum()) - 1) < 1e-6
        assert t.tobytes() == g[f"taps_{i}"].tobytes()
        # independent float32 numpy emulation of Vigra's formula (exp may differ from expf by 1 ulp)
        x = np.arange(-r, r + 1, dtype=np.float32)
        s = np.float32(sigma)
        e = np.exp((x * x * np.float32(-0.5 / float(s) / float(s))).astype(np.float32)).astype(np.float32)
        e = (e * np.float32(1.0 / np.sqrt(2 * np.pi) / float(s))).astype(np.float32)
        acc = np.float32(0)
        for v in e:
            acc = np.float32(acc + v)
        e = (e * np.float32(np.float32(1) / acc)).astype(np.float32)
        assert np.abs(e - t).max() <= 2e-7
    r, t = O.gauss_taps(1.6)  # SURVEY §8(c) provisional values
    assert np.allclose(t[5:], [0.24945803, 0.20519857, 0.11421021, 0.04301196, 0.01096042, 0.00188981], atol=2e-8)
    assert O.gauss_taps(0.0)[0] == 0 and O.gauss_taps(0.0)[1].tolist() == [1.0]
    assert O.gauss_taps(0.05)[0] == 1                     # radius 0 is bumped to 1


def test_resize_index_maps():
    m = O.resize_index_map(1920, 960)
    assert (m[1], m[479], m[480], m[959]) == (2, 958, 961, 1919)
    assert O.resize_index_map(1080, 540)[269:271].tolist() == [538, 541]
    assert (O.resize_index_map(135, 68) == 2 * np.arange(68)).all()       # exact 2i for odd sizes
    assert (O.resize_index_map(1920, 3840) == np.arange(3840) // 2).all()
    g = np.load(f"{GOLDEN}/kats.npz")
    for k in g.files:
        if k.startswith("lut_"):
            a, b = map(int, k.split("_")[1:])
            assert (O.resize_index_map(a, b) == g[k]).all()


def test_convolution_border_and_order():
    # reflect without repeating the edge pixel, ascending-source float summation
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (9, 13)).astype(np.float32)
    r, t = O.gauss_taps(1.6)
    out = O.convolve(img, 1.6)
    h, w = img.shape

    def refl(p, n):
        return -p if p < 0 else (2 * (n - 1) - p if p >= n else p)
    tmp = np.zeros_like(img)
    for y in range(h):
        for x in range(w):
            s = np.float32(0)
            for k in range(2 * r + 1):
                s = np.float32(s + np.float32(t[2 * r - k] * img[y, refl(x - r + k, w)]))
            tmp[y, x] = s
    ref = np.zeros_like(img)
    for y in range(h):
        for x in range(w):
            s = np.float32(0)
            for k in range(2 * r + 1):
                s = np.float32(s + np.float32(t[2 * r - k] * tmp[refl(y - r + k, h), x]))
            ref[y, x] = s
    assert out.tobytes() == ref.tobytes()
    with pytest.raises(O.OracleError, match=r"separableConvolveX\(\): kernel longer than line"):
        O.convolve(np.zeros((20, 5), np.float32), 1.6)
    with pytest.raises(O.OracleError, match=r"separableConvolveY\(\): kernel longer than line"):
        O.convolve(np.zeros((5, 20), np.float32), 1.6)


def test_scale_schedule_and_quirks():
    run = O.OracleRun(synth_frame(160, 120, 1), 3, 2)
    s = np.float32(1.6)
    k = np.float32(np.sqrt(2.0))
    # B-2: g(0,0) and g(0,1) share sigma; exponent restarts at exp-2 per octave
    assert run.scale("gaussian", 0, 0) == run.scale("gaussian", 0, 1) == s
    assert run.scale("dog", 0, 0) == 0.0
    assert run.scale("gaussian", 1, 0) == run.scale("gaussian", 0, 2) == np.float32(float(k) * float(s))
    assert run.scale("gaussian", 1, 1) == run.scale("gaussian", 1, 0)
    # a5: DoG = 128 + (higher - lower)
    d = run.level("dog", 0, 1)
    assert d.tobytes() == (np.float32(128) + (run.level("gaussian", 0, 2) - run.level("gaussian", 0, 1))).tobytes()
    pts, desc = run.points("final")
    assert pts.size > 50
    assert np.all(np.abs(pts["orientation"] - 177.4913) < 1e-3)          # B-9
    d = desc.reshape(-1, 16, 8)
    assert np.all(d[..., 7] == 0)                                         # B-8/B-12: % 7
    sm = d.sum(axis=2)
    assert np.all((np.abs(sm - 1) < 1e-5) | (sm == 0))                    # L1 per cell
    cand, _ = run.points("candidates")
    # B-3: order octave, dog, x outer, y inner
    key = cand["octave"].astype(np.int64) * 10 ** 8 + cand["x"].astype(np.int64) * 10 ** 4 + cand["y"]
    assert (np.diff(key) > 0).all()
    # mutated gradient pyramid: orientation level (0,0) carries multiples of ~177.49 (B-11)
    assert run.level("orientation", 0, 0).max() > 177


def test_extrema_are_2x2x3_nonstrict():
    run = O.OracleRun(synth_frame(96, 80, 3), 3, 1)
    d = [run.level("dog", 0, j) for j in range(3)]
    cand, _ = run.points("candidates")
    got = set(zip(cand["x"].tolist(), cand["y"].tolist()))
    want = set()
    h, w = d[1].shape
    for x in range(1, w - 1):
        for y in range(1, h - 1):
            nb = np.stack([a[y - 1:y + 1, x - 1:x + 1] for a in d]).ravel()
            c = d[1][y, x]
            if not (nb > c).any() or not (nb < c).any():
                want.add((x, y))
    assert got == want and len(got) > 100


def test_constant_image_and_u16_truncation():
    run = O.OracleRun(np.full((64, 80), 50.0, np.float32), 3, 1)
    cand, _ = run.points("candidates")
    assert cand.size == 78 * 62 and cand["filtered"].all()
    assert run.points("final")[0].size == 0
    # B-7: survivor count is truncated to 16 bits by `u16_t size`
    flags = np.zeros(70000, np.uint8)
    flags[::7] = 1
    perm = O.sort_by_filter(flags)
    assert sorted(perm.tolist()) == list(range(70000))
    nz = int((flags == 0).sum())
    assert (flags[perm[:nz]] == 0).all() and (flags[perm[nz:]] == 1).all()


def test_exceptions():
    img = synth_frame(160, 120, 1)
    r = O.OracleRun(img, 3, 4)               # octave 3 is 20x15: radius 14 needs 15 rows... r=19 fails
    assert r.status == 1 and "kernel longer than line" in r.error
    assert O.OracleRun(img, 2, 3).status == 2 and O.OracleRun(img, 3, 0).status == 2
    # B-13: 1080p, subpixel, 4 oct x 5 DoG throws separableConvolveY at octave 3 (480x270, r=307)
    big = np.zeros((1080, 1920), np.float32)
    r = O.OracleRun(big, 5, 4, subpixel=True)
    assert r.status == 1 and "separableConvolveY(): kernel longer than line" in r.error
    assert r.image().shape == (2160, 3840)   # B-16: the caller's image is already replaced


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(name):
    g, img, dogs, octaves, subpixel = load_case(name)
    run = O.OracleRun(img, dogs, octaves, subpixel=subpixel)
    assert run.status == 0
    for k, want in zip(g["level_names"], g["level_sha"]):
        kind = "gaussian" if k[0] == "g" else "dog"
        o, j = map(int, k[1:].split("_"))
        assert sha(run.level(kind, o, j)) == want, k
    fin, desc = run.points("final")
    assert [run.points(s)[0].size for s in ("candidates", "after_sort1", "after_orient", "after_sort2", "final")] == g["counts"].tolist()
    assert (fin["x"] == g["kp_x"]).all() and (fin["y"] == g["kp_y"]).all() and (fin["octave"] == g["kp_octave"]).all()
    assert fin["orientation"].tobytes() == g["kp_orientation"].tobytes()
    assert desc.tobytes() == g["descriptors"].tobytes()


def test_faithful_cost_mode_gives_identical_results():
    img = synth_frame(128, 96, 4)
    a = O.OracleRun(img, 3, 2, faithful=False)
    b = O.OracleRun(img, 3, 2, faithful=True)
    pa, da = a.points("final")
    pb, db = b.points("final")
    assert pa.tobytes() == pb.tobytes() and da.tobytes() == db.tobytes() and pa.size > 20
