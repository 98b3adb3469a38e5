"""Generates tests/golden/opj_part1.npz: Part-1 (MQ-coded) JPEG 2000 codestreams written by a third-party
ENCODER (OpenJPEG 2.5.4 as bundled with Pillow 12.2) together with the pixels they must decode to.

    python tests/golden/make_openjpeg_part1.py

For the reversible streams the expected pixels are the source image (lossless: any conforming decoder must
return exactly these).  For the irreversible ones the expectation is OpenJPEG's own decode of the stream and
the tests allow 1 LSB (float 9/7 implementations differ in rounding).  Only data is stored: streams and pixels.
"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def image(h, w, c, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = (128 + 60 * np.sin(x / 17.0 + seed) + 50 * np.cos(y / 23.0))[..., None] + rng.integers(-9, 10, (h, w, c))
    return np.clip(base, 0, 255).astype(np.uint8)


CASES = {
    # name: (h, w, components, Pillow save options)
    "opj_gray_64":          (64, 64, 1, dict(irreversible=False)),
    "opj_gray_odd":         (77, 131, 1, dict(irreversible=False, num_resolutions=4)),
    "opj_rgb_mct":          (96, 120, 3, dict(irreversible=False, mct=1)),
    "opj_rgb_cb32_layers":  (128, 128, 3, dict(irreversible=False, codeblock_size=(32, 32), quality_mode="rates",
                                                quality_layers=[40, 10, 1], progression="RPCL")),
    "opj_gray_tiles":       (100, 150, 1, dict(irreversible=False, tile_size=(64, 64), num_resolutions=3)),
    "opj_gray_cb16x64":     (90, 90, 1, dict(irreversible=False, codeblock_size=(16, 64))),
    "opj_rgb_97_rate":      (96, 96, 3, dict(irreversible=True, quality_mode="rates", quality_layers=[12])),
    "opj_gray_97_layers":   (80, 112, 1, dict(irreversible=True, quality_mode="rates", quality_layers=[30, 8])),
}


def main():
    out = {}
    for name, (h, w, c, kw) in CASES.items():
        im = image(h, w, c, len(name))
        img = Image.fromarray(im[..., 0] if c == 1 else im)
        bio = io.BytesIO()
        img.save(bio, format="JPEG2000", no_jp2=True, **kw)
        data = bio.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(data)))
        src = im[..., 0] if c == 1 else im
        if not kw["irreversible"]:
            assert np.array_equal(dec, src), name
        out[name + ".j2k"] = np.frombuffer(data, dtype=np.uint8)
        out[name + ".pix"] = dec
        out[name + ".lossless"] = np.array([0 if kw["irreversible"] else 1], dtype=np.uint8)
        print(name, len(data), "bytes")
    np.savez_compressed(os.path.join(HERE, "opj_part1.npz"), **out)


if __name__ == "__main__":
    main()
