"""Input staging: the numpy restatement against hand-computed values (CPU tier; parity unpinned, see
oracle/staging_oracle.py) and the HIP kernel + pinned-ring stager against the restatement (GPU tier)."""
import numpy as np
import pytest
import torch

from oracle import staging_oracle as S


def test_oracle_known_answers():
    sar = np.array([[[1.0, 0.0], [np.nan, -3.0]]], dtype=np.float32)
    got = S.load_sar(sar.copy())
    # 10*log10(1+1e-7) ~ 4.3e-7 -> clipped to 0 ; log10(1e-7) = -70 -> -25 ; NaN -> 0 ; log10(negative) = NaN -> 0
    want = (np.array([[[0.0, -25.0], [0.0, 0.0]]]) - (-7.9447875)) / 2.777256
    assert np.allclose(got, want, atol=1e-6) and got.dtype == np.float32
    rgb = np.zeros((3, 2, 2), dtype=np.uint8); rgb[0] = 81; rgb[1] = 200; rgb[2, 0, 0] = 255
    got = S.load_rgb(rgb.copy())
    assert np.allclose(got[0], (81 - 81.29692) / 39.61512, atol=1e-6)
    assert np.allclose(got[1], (200 - 87.93711) / 35.407978, atol=1e-6)
    assert np.isclose(got[2, 0, 0], (255 - 72.041306) / 35.84708, atol=1e-6)
    dsm = np.array([[[1.0, 3.0], [np.nan, 4.0]]], dtype=np.float32)           # NaN -> 0: values 1,3,0,4: mean 2, var 2.5
    got = S.load_dsm(dsm.copy())
    assert np.allclose(got, (np.array([[[1.0, 3.0], [0.0, 4.0]]]) - 2.0) / np.sqrt(2.5 + 1e-6), atol=1e-6)


def test_oracle_area_resize_is_block_mean_and_uint8_rounds():
    img = np.arange(16, dtype=np.float32).reshape(1, 4, 4)
    assert np.array_equal(S.resize_area(img, 2)[0], np.array([[2.5, 4.5], [10.5, 12.5]]))
    u8 = np.array([[[1, 2, 0, 0], [2, 2, 0, 1], [5, 5, 9, 9], [5, 6, 9, 9]]], dtype=np.uint8)
    # block means 1.75, 0.25, 5.25, 9.0 -> rounded like cv2's uint8 output: 2, 0, 5, 9
    assert np.array_equal(S.resize_area(u8, 2)[0], np.array([[2.0, 0.0], [5.0, 9.0]]))
    assert np.array_equal(S.resize_area(img, 1)[0], img[0])


def test_oracle_area_resize_matches_pillow_box_reduce():
    """A second, independent implementation of box (area) resampling for integer factors: Pillow's Image.reduce on float
    rasters.  It anchors the float block mean only -- cv2's uint8 rounding stays held to the hand-computed case above (cv2 is
    absent here: that part of row f3 is still 'parity unpinned')."""
    Image = pytest.importorskip("PIL.Image")
    g = np.random.default_rng(5)
    for f in (2, 3, 4):
        img = g.normal(0.0, 3.0, size=(2, 12 * f, 8 * f)).astype(np.float32)
        got = S.resize_area(img, f)
        for c in range(2):
            want = np.asarray(Image.fromarray(img[c], mode="F").reduce(f), dtype=np.float32)
            assert want.shape == got[c].shape
            assert np.abs(got[c] - want).max() <= 1e-5 * np.abs(img).max(), f


def _raw(B, f, seed):
    g = np.random.default_rng(seed)
    n = 256 * f
    sar = g.gamma(2.0, 0.1, size=(B, 1, n, n)).astype(np.float32)
    sar[0, 0, :3, :5] = np.nan; sar[-1, 0, 7, 7] = -1.0; sar[0, 0, 9, 9] = 0.0; sar[0, 0, 10, 10] = np.inf
    rgb = g.integers(0, 256, size=(B, 3, n, n), dtype=np.uint8)
    dsm = (g.normal(5.0, 7.0, size=(B, 1, n, n))).astype(np.float32)
    dsm[0, 0, 100:110, 50:60] = np.nan
    return {'s1': sar, 's2': rgb, 'dem': dsm}


@pytest.mark.gpu
@pytest.mark.parametrize("f", [1, 2])
def test_stage_kernels_match_oracle(f):
    from incomplete_multimodal_fusion_amd import staging
    raw = _raw(3, f, 10 + f)
    want = {'s1': np.stack([S.load_sar(t, f) for t in raw['s1']]), 's2': np.stack([S.load_rgb(t, f) for t in raw['s2']]),
            'dem': np.stack([S.load_dsm(t, f) for t in raw['dem']])}
    for d, r in staging.DFC2023.items():
        got = staging.stage_tiles(torch.from_numpy(raw[d]).cuda(), r['kind'], 256, r['mean'], r['std']).cpu().numpy()
        assert got.shape == want[d].shape and got.dtype == np.float32
        err = np.abs(got - want[d]).max()
        assert err <= 2e-5 * max(1.0, np.abs(want[d]).max()), (d, f, err)     # fp32 log10 / summation order
    rgbf = raw['s2'].astype(np.float32)                                        # float RGB: no uint8 re-quantisation
    got = staging.stage_tiles(torch.from_numpy(rgbf).cuda(), staging.AFFINE, 256, S.RGB_MEAN, S.RGB_STD).cpu().numpy()
    want_f = np.stack([S.load_rgb(t, f) for t in rgbf])
    assert np.abs(got - want_f).max() <= 2e-5 * np.abs(want_f).max()


@pytest.mark.gpu
def test_stager_ring_overlaps_and_returns_each_batch():
    from incomplete_multimodal_fusion_amd import staging
    st = staging.TileStager("cuda:0", image_size=256, slots=2)
    batches = [_raw(2, 1, 100 + i) for i in range(5)]
    st.submit(batches[0])
    outs = []
    for i in range(5):
        x = st.get()
        if i + 1 < 5:
            st.submit(batches[i + 1])
        outs.append({d: (v.double().sum().item(), v.clone()) for d, v in x.items()})   # "the step": consumes x on the compute stream
    for i, o in enumerate(outs):
        for d, r in staging.DFC2023.items():
            direct = staging.stage_tiles(torch.from_numpy(batches[i][d]).cuda(), r['kind'], 256, r['mean'], r['std'])
            assert torch.equal(o[d][1], direct), (i, d)
    with pytest.raises(AssertionError):
        st.get()
