"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's input staging arithmetic
(pretraining/utils/multimodal_dfc2023.py).  Only tests/ may import this.

PARTLY PINNED: the reference module imports rasterio and cv2 at the top (both absent in this image).
oracle/ref_loader.load_aux() imports it with never-called placeholder module objects for those two names, which makes the
constants and the functions that need neither -- normalization / normalize_rgb / normalize_sar / normalize_dem (:18-49) --
callable: tests/golden/aux.npz (`staging/*`) pins those here.  The raster-reading load_* functions (:99-139: dB
conversion, clip, nan_to_num, z-score, and their ORDER) and cv2.resize(INTER_AREA) cannot be run and stay UNPINNED:
restated from the cited lines, resize for integer shrink factors as the block mean (factor 1 = identity) with cv2's dtype
rule (uint8 in -> uint8 out, rounded half to even), checked against hand-computed answers only.
"""
import numpy as np

RGB_MEAN = np.array([81.29692, 87.93711, 72.041306])       # multimodal_dfc2023.py:25-26
RGB_STD = np.array([39.61512, 35.407978, 35.84708])
SAR_MEAN = np.array([-7.9447875])                           # :34-35
SAR_STD = np.array([2.777256])
DEM_MEAN = np.array([5.0160093])                            # :43-44 (normalize_dem: defined, not used by load_dsm)
DEM_STD = np.array([7.6128364])


def normalization(data):                                    # :18-24 (min-max)
    rng = np.max(data) - np.min(data)
    return (data - np.min(data)) / rng


def normalize_rgb(imgs):                                    # :28-31 (in place, per channel)
    for i in range(3):
        imgs[i] = (imgs[i] - RGB_MEAN[i]) / RGB_STD[i]
    return imgs


def normalize_sar(imgs):                                    # :37-40
    for i in range(1):
        imgs[i] = (imgs[i] - SAR_MEAN[i]) / SAR_STD[i]
    return imgs


def normalize_dem(imgs):                                    # :46-49
    for i in range(1):
        imgs[i] = (imgs[i] - DEM_MEAN[i]) / DEM_STD[i]
    return imgs


def resize_area(img: np.ndarray, factor: int) -> np.ndarray:
    """resiz_4pl (:10-16) for an integer shrink factor: float64 output buffer, per-channel INTER_AREA."""
    C, Hr, Wr = img.shape
    H, W = Hr // factor, Wr // factor
    out = np.zeros((C, H, W))
    for c in range(C):
        per = img[c]
        if factor > 1:
            blk = per.reshape(H, factor, W, factor).astype(np.float32).mean(axis=(1, 3), dtype=np.float32)
            per = np.rint(blk).astype(np.uint8) if img.dtype == np.uint8 else blk.astype(img.dtype)
        out[c] = per
    return out


def load_sar(sar: np.ndarray, factor: int = 1) -> np.ndarray:       # :127-139 after the raster read
    with np.errstate(divide="ignore", invalid="ignore"):
        sar = 10 * np.log10(sar + 0.0000001)
    sar = np.clip(sar, -25, 0)
    sar = np.nan_to_num(sar)
    sar = resize_area(sar, factor).astype(np.float32)
    return normalize_sar(sar)


def load_rgb(rgb: np.ndarray, factor: int = 1) -> np.ndarray:       # :114-124
    rgb = np.nan_to_num(rgb)
    rgb = resize_area(rgb, factor).astype(np.float32)
    return normalize_rgb(rgb)


def load_dsm(dsm: np.ndarray, factor: int = 1) -> np.ndarray:       # :99-111 (dsm already (1, H, W))
    dsm = np.nan_to_num(dsm)
    dsm = resize_area(dsm, factor).astype(np.float32)
    return (dsm - dsm.mean()) / np.sqrt(dsm.var() + 1e-6)
