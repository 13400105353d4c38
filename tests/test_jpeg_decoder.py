"""CPU tests of the harness's JPEG reader (samples/jpeg_decoder.h; the reference loads its sample images through the vendored
stb_image, dependencies/stbi/stbi_wrapper.cpp:37-44).  JPEG decoders may differ by the rounding of their inverse DCT (+-1 per
sample) and, for subsampled chroma, by the interpolation filter; the decoder is pinned against PIL (libjpeg) within those bounds
on baseline and progressive files, grayscale and colour, with and without restart markers -- and on the reference's
data/images/albert.jpg (progressive grayscale, BASELINE config 3) where the reference tree is present."""
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PIL = pytest.importorskip("PIL.Image")
ALBERT = "/root/reference/data/images/albert.jpg"


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    gxx = shutil.which("g++")
    assert gxx
    exe = tmp_path_factory.mktemp("jpeg") / "jpeg_to_pnm"
    subprocess.check_call([gxx, "-O2", "-std=c++14", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "jpeg_to_pnm.cpp"), "-o", str(exe)])
    return str(exe)


def _decode(tool, path, tmp_path):
    out = tmp_path / "out.pnm"
    r = subprocess.run([tool, str(path), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = out.read_bytes()
    magic, dims, maxval, data = raw.split(b"\n", 3)
    w, h = (int(v) for v in dims.split())
    ch = 1 if magic == b"P5" else 3
    return np.frombuffer(data, dtype=np.uint8).reshape(h, w, ch)


def _test_image(w, h, channels):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 128 + 90 * np.sin(x / 17.0) * np.cos(y / 23.0) + 30 * np.sin((x + 2 * y) / 5.0)
    rs = np.random.RandomState(0)
    img = np.stack([base + 20 * np.sin(x / (7.0 + 3 * c)) + rs.normal(0, 6, (h, w)) for c in range(channels)], axis=2)
    img[h // 3: h // 3 + 9, w // 4: w // 4 + 40] = 255  # a hard edge
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [
    # (channels, size, save options, max abs difference, mean abs difference)
    (1, (97, 61), dict(quality=90), 2, 0.6),
    (1, (97, 61), dict(quality=75, progressive=True), 2, 0.6),
    (1, (256, 200), dict(quality=95, progressive=True, optimize=True), 2, 0.6),
    (3, (120, 83), dict(quality=90, subsampling=0), 3, 0.8),
    (3, (120, 83), dict(quality=85, subsampling=0, progressive=True), 3, 0.8),
    (3, (131, 77), dict(quality=90, subsampling=2), 80, 3.0),                    # 4:2:0: libjpeg interpolates chroma, this reader replicates it
    (3, (131, 77), dict(quality=90, subsampling=1, progressive=True), 80, 3.0),  # 4:2:2
    (1, (150, 90), dict(quality=80, restart_marker_blocks=3), 2, 0.6),
    (3, (150, 90), dict(quality=80, subsampling=0, restart_marker_rows=1), 3, 0.8),
]


@pytest.mark.parametrize("channels,size,opts,max_abs,mean_abs", CASES)
def test_decoder_matches_pil(tool, tmp_path, channels, size, opts, max_abs, mean_abs):
    w, h = size
    src = _test_image(w, h, channels)
    im = PIL.fromarray(src[:, :, 0] if channels == 1 else src, mode="L" if channels == 1 else "RGB")
    path = tmp_path / "in.jpg"
    try:
        im.save(path, format="JPEG", **opts)
    except TypeError:
        pytest.skip("this PIL cannot write the requested JPEG variant")
    want = np.asarray(PIL.open(path).convert("L" if channels == 1 else "RGB"), dtype=np.int32).reshape(h, w, channels)
    got = _decode(tool, path, tmp_path).astype(np.int32)
    assert got.shape == want.shape
    diff = np.abs(got - want)
    assert diff.max() <= max_abs and diff.mean() <= mean_abs, (diff.max(), diff.mean())
    if opts.get("progressive"):
        assert b"\xff\xc2" in path.read_bytes()  # really a progressive file


def test_errors_are_reported(tool, tmp_path):
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"not a jpeg at all")
    r = subprocess.run([tool, str(bad), str(tmp_path / "o.pnm")], capture_output=True, text=True)
    assert r.returncode == 1 and "missing SOI" in r.stderr
    # a truncated file: decodes what is there or reports an error, never crashes
    buf = io.BytesIO()
    PIL.fromarray(_test_image(64, 64, 1)[:, :, 0], mode="L").save(buf, format="JPEG", quality=90)
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(buf.getvalue()[: len(buf.getvalue()) // 2])
    r = subprocess.run([tool, str(cut), str(tmp_path / "o.pnm")], capture_output=True, text=True)
    assert r.returncode in (0, 1)


@pytest.mark.skipif(not os.path.exists(ALBERT), reason="reference tree not present (GPU box)")
def test_reference_sample_image(tool, tmp_path):
    """data/images/albert.jpg, the image BASELINE config 3 trains on: 3250 x 4333, 8-bit grayscale, progressive."""
    want = np.asarray(PIL.open(ALBERT).convert("L"), dtype=np.int32)
    got = _decode(tool, ALBERT, tmp_path).astype(np.int32)[:, :, 0]
    assert got.shape == want.shape == (4333, 3250)
    diff = np.abs(got - want)
    assert diff.max() <= 2 and diff.mean() <= 0.5, (diff.max(), diff.mean())
