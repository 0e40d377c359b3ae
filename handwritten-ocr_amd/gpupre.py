"""Strategy preprocessing + image-processor resize on the device.

The reference applies a chain of named transforms to a page before every read (`preprocess_image`,
ocr_agent/tools.py:633-673), saves the result as a PNG, and the HF image processor re-opens and resizes it
(tools.py:756-762).  Where OpenCV is absent — the only configuration this repository can pin
(tests/golden/preprocess_kats.json) — the chain reduces to three PIL operations (high_contrast -> ImageEnhance.Contrast(2.0),
binarize -> convert("L").point(v > 128), sharpen -> ImageFilter.SHARPEN; deskew / denoise / remove_lines are identities,
tools.py:572, :588, :618).  `StrategyPages` runs them and Pillow's BICUBIC resize on the MI355X with Pillow's own integer
arithmetic (csrc/imagepre.hip): the page is uploaded once for all its strategy reads and the pixels the vision tower sees are
bit-identical to the host path (tests/test_gpupre_gpu.py compares against Pillow itself).

The host path stays the default; this one is chosen by HWOCR_GPU_PREPROCESS=1 in the batch driver and refuses to run where
OpenCV is importable (there the reference takes its cv2 branches, which this module does not restate).
"""
from __future__ import annotations

import functools
import math

import numpy as np
import torch

from . import _lib, preprocess

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c: coefficients of 8-bit images are 22-bit fixed point

_IDENTITY = ("deskew", "denoise", "remove_lines")  # without OpenCV (ocr_agent/tools.py:572, :588, :618)
_KNOWN = _IDENTITY + ("high_contrast", "binarize", "sharpen")


def _bicubic(x: float) -> float:
    """Pillow Resample.c bicubic_filter (a = -0.5)."""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=64)
def bicubic_coeffs(n_in: int, n_out: int) -> tuple[np.ndarray, np.ndarray]:
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for one axis: (bounds int32 [n_out][2] = first tap,
    number of taps; coef int32 [n_out][ksize])."""
    scale = n_in / n_out
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((n_out, 2), np.int32)
    coef = np.zeros((n_out, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(n_out):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        if ww != 0.0:
            k = [v / ww for v in k]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(k):
            coef[xx, x] = int(v * (1 << PRECISION_BITS) - 0.5) if v < 0 else int(v * (1 << PRECISION_BITS) + 0.5)
    return bounds, coef


def supported(strategy) -> bool:
    """True when the chain can run here: OpenCV absent, known names, and nothing after a `binarize` (PIL would run it on a
    mode-L image)."""
    if preprocess._cv2() is not None:
        return False
    names = [strategy] if isinstance(strategy, str) else list(strategy)
    names = [n for n in names if n in _KNOWN]  # unknown names are skipped by the reference with a warning
    if "binarize" in names and names.index("binarize") != len(names) - 1:
        return False
    return True


class StrategyPages:
    def __init__(self, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.HwocrError("GPU preprocessing needs an MI355X (ROCm) device")
        self.dev = torch.device(device)
        self.lib = _lib.hip()
        self._tables: dict = {}
        self._sum = torch.zeros(1, dtype=torch.int64, device=self.dev)

    def _coeffs(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        if key not in self._tables:
            b, c = bicubic_coeffs(n_in, n_out)
            self._tables[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(c).contiguous().to(self.dev), c.shape[1])
        return self._tables[key]

    def resize(self, img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
        """uint8 [H][W][3] on the device -> uint8 [out_h][out_w][3], Pillow's Image.resize(BICUBIC)."""
        H, W = int(img.shape[0]), int(img.shape[1])
        if (H, W) == (out_h, out_w):
            return img
        hb, hc, hk = self._coeffs(W, out_w)
        vb, vc, vk = self._coeffs(H, out_h)
        tmp = torch.empty(H, out_w, 3, dtype=torch.uint8, device=self.dev)
        out = torch.empty(out_h, out_w, 3, dtype=torch.uint8, device=self.dev)
        _lib.check(self.lib.hwocr_img_resize_bicubic(_lib.ptr(img), _lib.ptr(tmp), _lib.ptr(out), H, W, out_h, out_w,
                                                     _lib.ptr(hb), _lib.ptr(hc), hk, _lib.ptr(vb), _lib.ptr(vc), vk,
                                                     _lib.stream_handle()), "hwocr_img_resize_bicubic")
        return out

    def high_contrast(self, img: torch.Tensor) -> torch.Tensor:
        n = int(img.shape[0]) * int(img.shape[1])
        _lib.check(self.lib.hwocr_img_luma_sum(_lib.ptr(img), n, _lib.ptr(self._sum), _lib.stream_handle()), "hwocr_img_luma_sum")
        # ImageEnhance.Contrast: mean = int(ImageStat.Stat(L).mean[0] + 0.5), taken on the device (a .item() here would wait for
        # everything queued on the stream - with two lanes in flight that was 84 waits behind the other lane's kernels per batch)
        out = torch.empty_like(img)
        _lib.check(self.lib.hwocr_img_contrast_dev(_lib.ptr(img), _lib.ptr(out), 3 * n, _lib.ptr(self._sum), n, 2.0,
                                                   _lib.stream_handle()), "hwocr_img_contrast_dev")
        return out

    def binarize(self, img: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(img)
        _lib.check(self.lib.hwocr_img_binarize(_lib.ptr(img), _lib.ptr(out), int(img.shape[0]) * int(img.shape[1]),
                                               _lib.stream_handle()), "hwocr_img_binarize")
        return out

    def sharpen(self, img: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(img)
        _lib.check(self.lib.hwocr_img_sharpen(_lib.ptr(img), _lib.ptr(out), int(img.shape[0]), int(img.shape[1]),
                                              _lib.stream_handle()), "hwocr_img_sharpen")
        return out

    def apply(self, img: torch.Tensor, strategy) -> torch.Tensor:
        """The chain of `preprocess.apply_strategy` on a device image (RGB uint8 [H][W][3])."""
        if not supported(strategy):
            raise _lib.HwocrError(f"strategy {strategy!r} is not covered by the device path (OpenCV present, or a transform "
                                  "after binarize): use the host path")
        names = [strategy] if isinstance(strategy, str) else list(strategy)
        if len(names) == 0 or names == ["original"]:
            return img
        for name in names:
            if name in _IDENTITY or name not in _KNOWN:
                continue
            img = getattr(self, name)(img)
        return img

    def pages(self, page_rgb, strategies: list, target_hw: tuple[int, int]) -> list[torch.Tensor]:
        """page_rgb: RGB uint8 [H][W][3], a host array or a device tensor.  One upload of the original page, every strategy's tower-resolution image resident in HBM.  Shared prefixes of
        the chains (all of the reference's start with high_contrast) are computed once."""
        if isinstance(page_rgb, torch.Tensor):  # already resident in HBM (uint8 [H][W][3])
            if page_rgb.dtype != torch.uint8 or page_rgb.dim() != 3 or page_rgb.shape[2] != 3:
                raise ValueError("a device page must be uint8 [H][W][3]")
            base = page_rgb.to(self.dev).contiguous()
        else:
            base = torch.from_numpy(np.array(page_rgb, dtype=np.uint8, order="C")).to(self.dev)  # (a copy: PIL arrays are read-only)
        cache: dict = {(): base}
        out = []
        for s in strategies:
            names = tuple(n for n in ([s] if isinstance(s, str) else list(s)) if n in _KNOWN and n not in _IDENTITY)
            if not supported(s):
                raise _lib.HwocrError(f"strategy {s!r} is not covered by the device path")
            for k in range(1, len(names) + 1):
                if names[:k] not in cache:
                    cache[names[:k]] = getattr(self, names[k - 1])(cache[names[: k - 1]])
            out.append(self.resize(cache[names], *target_hw))
        return out
