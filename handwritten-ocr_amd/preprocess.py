"""Preprocessing strategies of a page read (mirror of ocr_agent/tools.py:496-673).

`preprocess_image(path, strategy) -> path` keeps the reference's contract (chain of named transforms, "original"
short-circuit, unknown names skipped with a printed warning, result saved to a temp file `ocr_<label>_*` that the
caller never deletes).  `apply_strategy(img, strategy)` is the same chain in memory, used by the batched driver so
that three reads of a page do not each pay a PNG encode + decode.

Each transform uses OpenCV when it is importable and otherwise the reference's PIL fallback (tools.py:513-516,
:530-531, :543-546; deskew / denoise / remove_lines fall back to identity, :572, :588, :618).  OpenCV is not
installed in the build or test images, so only the fallback behaviour is pinned by tests/golden/preprocess_kats.json;
the OpenCV branches are written to the reference's parameters (CLAHE clip 3.0 / 8x8; adaptive Gaussian 21 / C 10;
3x3 sharpen kernel; minAreaRect deskew with cubic warp + replicate border; NL-means h 10, 7, 21; line removal by
(W/4 x 1) opening, 1x3 dilation, Telea inpaint r 3) and are parity-unpinned.
"""
from __future__ import annotations

import tempfile
from pathlib import Path

from PIL import Image, ImageEnhance, ImageFilter


def _cv2():
    try:
        import cv2  # noqa: F401

        return cv2
    except ImportError:
        return None


def _gray(cv2, arr):
    return cv2.cvtColor(arr, cv2.COLOR_RGB2GRAY) if arr.ndim == 3 else arr


def high_contrast(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return ImageEnhance.Contrast(img).enhance(2.0)
    import numpy as np

    return Image.fromarray(cv2.createCLAHE(clipLimit=3.0, tileGridSize=(8, 8)).apply(_gray(cv2, np.array(img))))


def binarize(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return img.convert("L").point(lambda v: 255 if v > 128 else 0)
    import numpy as np

    return Image.fromarray(cv2.adaptiveThreshold(_gray(cv2, np.array(img)), 255, cv2.ADAPTIVE_THRESH_GAUSSIAN_C,
                                                 cv2.THRESH_BINARY, 21, 10))


def sharpen(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return img.filter(ImageFilter.SHARPEN)
    import numpy as np

    k = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], dtype=np.float32)
    return Image.fromarray(cv2.filter2D(np.array(img), -1, k))


def deskew(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return img
    import numpy as np

    arr = np.array(img)
    g = _gray(cv2, arr)
    ink = np.column_stack(np.where(g < 128))
    if len(ink) <= 100:
        return img
    angle = cv2.minAreaRect(ink)[-1]
    angle = -(90 + angle) if angle < -45 else -angle
    h, w = g.shape
    rot = cv2.getRotationMatrix2D((w // 2, h // 2), angle, 1.0)
    return Image.fromarray(cv2.warpAffine(arr, rot, (w, h), flags=cv2.INTER_CUBIC, borderMode=cv2.BORDER_REPLICATE))


def denoise(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return img
    import numpy as np

    arr = np.array(img)
    if arr.ndim == 3:
        return Image.fromarray(cv2.fastNlMeansDenoisingColored(arr, None, 10, 10, 7, 21))
    return Image.fromarray(cv2.fastNlMeansDenoising(arr, None, 10, 7, 21))


def remove_lines(img: Image.Image) -> Image.Image:
    cv2 = _cv2()
    if cv2 is None:
        return img
    import numpy as np

    arr = np.array(img)
    g = _gray(cv2, arr)
    thr = cv2.adaptiveThreshold(cv2.bitwise_not(g), 255, cv2.ADAPTIVE_THRESH_MEAN_C, cv2.THRESH_BINARY, 15, -2)
    mask = cv2.morphologyEx(thr, cv2.MORPH_OPEN, cv2.getStructuringElement(cv2.MORPH_RECT, (g.shape[1] // 4, 1)),
                            iterations=1)
    mask = cv2.dilate(mask, cv2.getStructuringElement(cv2.MORPH_RECT, (1, 3)))
    return Image.fromarray(cv2.inpaint(arr, mask, 3, cv2.INPAINT_TELEA))


TRANSFORMS = {"high_contrast": high_contrast, "binarize": binarize, "sharpen": sharpen, "deskew": deskew,
              "denoise": denoise, "remove_lines": remove_lines}


def steps_of(strategy) -> list[str]:
    return [strategy] if isinstance(strategy, str) else list(strategy)


def label_of(strategy) -> str:
    return "+".join(s for s in steps_of(strategy) if s != "original")


def apply_strategy(img: Image.Image, strategy, quiet: bool = False) -> Image.Image:
    for step in steps_of(strategy):
        if step == "original":
            continue
        fn = TRANSFORMS.get(step)
        if fn is None:
            if not quiet:
                print(f"  [preprocess] Unknown transform '{step}', skipping")
            continue
        img = fn(img)
    return img


LOSSY_SUFFIXES = (".jpg", ".jpeg", ".webp")  # of the reference's IMAGE_EXTENSIONS (transcribe.py:29): saving re-quantises


def through_tempfile_codec(img: Image.Image, suffix: str) -> Image.Image:
    """What the next reader sees after `preprocess_image` saved `img` under the input's suffix (tools.py:668-672) and
    `run_ocr` re-opened it (tools.py:745): for .jpg / .jpeg / .webp that is a lossy re-encode with Pillow's default
    settings, for the other formats the same pixels.  In memory — the batched driver keeps no temp files."""
    suffix = (suffix or ".png").lower()
    if suffix not in LOSSY_SUFFIXES:
        return img
    import io

    fmt = Image.registered_extensions()[suffix]
    buf = io.BytesIO()
    img.save(buf, format=fmt)
    out = Image.open(io.BytesIO(buf.getvalue()))
    out.load()
    return out


def preprocess_image(image_path: str, strategy) -> str:
    steps = steps_of(strategy)
    if not steps or steps == ["original"]:
        return image_path
    label = label_of(strategy)
    print(f"  [preprocess] Applying {label}...")
    out = apply_strategy(Image.open(image_path), steps)
    suffix = Path(image_path).suffix or ".png"
    tmp = tempfile.NamedTemporaryFile(suffix=suffix, delete=False, prefix=f"ocr_{label}_")
    out.save(tmp.name)
    return tmp.name
