"""samples/mlp_learning_an_image: the caller harness -- image lookup kernel, training loop and the image
benchmark protocol of the reference (samples/mlp_learning_an_image.cu, benchmarks/image/bench_ours.cu), run as a program."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "samples", "mlp_learning_an_image")


def bilinear_lookup(img, xy):
    """numpy restatement of k_eval_image (= tex2D<float4> with normalized coordinates, linear filter, clamp addressing as the CUDA
    programming guide specifies it: xB = x W - 0.5, weights with 8 fractional bits).  Parity for this harness kernel is UNPINNED:
    no texture unit is available to run the reference's lookup; the formula is the published one."""
    h, w, _ = img.shape
    x = xy[:, 0].astype(np.float32)
    y = xy[:, 1].astype(np.float32)
    xb = x * np.float32(w) - np.float32(0.5)
    yb = y * np.float32(h) - np.float32(0.5)
    xf, yf = np.floor(xb), np.floor(yb)
    a = np.floor((xb - xf) * np.float32(256.0) + np.float32(0.5)) * np.float32(1 / 256)
    b = np.floor((yb - yf) * np.float32(256.0) + np.float32(0.5)) * np.float32(1 / 256)
    i0 = np.clip(xf.astype(np.int64), 0, w - 1)
    i1 = np.clip(xf.astype(np.int64) + 1, 0, w - 1)
    j0 = np.clip(yf.astype(np.int64), 0, h - 1)
    j1 = np.clip(yf.astype(np.int64) + 1, 0, h - 1)
    one = np.float32(1)
    w00, w10, w01, w11 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
    out = (w00[:, None] * img[j0, i0, :3] + w10[:, None] * img[j0, i1, :3]) + w01[:, None] * img[j1, i0, :3] + w11[:, None] * img[j1, i1, :3]
    return out.astype(np.float32)


def test_sample_source_and_configs_exist():
    assert os.path.exists(os.path.join(ROOT, "samples", "mlp_learning_an_image.hip"))
    for name in ("config_hash.json", "config_oneblob.json"):
        assert os.path.exists(os.path.join(ROOT, "samples", name))


def test_pnm_roundtrip_host_side(tmp_path):
    """The PPM the test writes is what the harness' loader expects (8-bit P6); decoding itself is checked on the GPU test below."""
    img = (np.arange(6 * 5 * 3) % 251).astype(np.uint8).reshape(5, 6, 3)
    p = tmp_path / "t.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n# comment\n6 5\n255\n" + img.tobytes())
    assert os.path.getsize(p) == len(b"P6\n# comment\n6 5\n255\n") + 90


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        pytest.fail("samples/mlp_learning_an_image is not built: run python tiny-cuda-nn_amd/build.py")
    return EXE


@pytest.mark.gpu
def test_image_lookup_matches_restatement(exe, tmp_path):
    rs = np.random.RandomState(0)
    # a PPM with a comment line, odd sizes; coordinates include the borders (clamp addressing) and exact texel centres
    img8 = rs.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    ppm = tmp_path / "img.ppm"
    with open(ppm, "wb") as f:
        f.write(b"P6\n# made by the test\n53 37\n255\n" + img8.tobytes())
    xy = rs.uniform(-0.05, 1.05, size=(4096, 2)).astype(np.float32)
    xy[:64, 0] = (np.arange(64) % 53 + 0.5) / 53
    xy[:64, 1] = (np.arange(64) % 37 + 0.5) / 37
    xy[64:70] = [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [0.999999, 0.000001]]
    coords = tmp_path / "coords.f32"
    xy.tofile(coords)
    out = tmp_path / "out.f32"
    r = subprocess.run([exe, "--sample", str(ppm), str(coords), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.float32).reshape(-1, 3)
    raw = np.fromfile(str(out) + ".image", dtype=np.uint8)
    w, h = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
    assert (w, h) == (53, 37)
    img = np.frombuffer(raw[8:].tobytes(), dtype=np.float32).reshape(h, w, 4)
    # decoding: stbi_loadf semantics, value = (v / 255)^2.2, alpha 1
    want_img = (img8.astype(np.float32) / np.float32(255)) ** np.float32(2.2)
    assert np.allclose(img[:, :, :3], want_img, rtol=2e-6, atol=1e-7) and np.all(img[:, :, 3] == 1)
    want = bilinear_lookup(img, xy)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    # at texel centres the lookup returns the texel
    assert np.allclose(got[:64], img[np.arange(64) % 37, np.arange(64) % 53, :3], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_sample_learns_the_synthetic_image(exe, tmp_path):
    """The reference sample's loop (random batch -> lookup -> training_step, 2^18 samples per step) with the hash-grid config:
    after 300 steps the learned image must be close to the target (PSNR in linear colour)."""
    cfg = os.path.join(ROOT, "samples", "config_hash.json")
    final = tmp_path / "final.ppm"
    r = subprocess.run([exe, "synthetic:512x512", cfg, "300", str(final)], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("PSNR")][-1]
    psnr = float(line.split(":")[1].split()[0])
    assert psnr > 30.0, r.stdout
    head = open(final, "rb").read(15)
    assert head.startswith(b"P6\n512 512\n255\n")
    assert os.path.exists(tmp_path / "reference.ppm")
    losses = [float(l.split("loss=")[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("Step#")]
    assert losses[-1] < 0.1 * losses[0]


@pytest.mark.gpu
def test_bench_protocol_writes_result_json(exe, tmp_path):
    """bench_ours.cu's protocol on two batch sizes: the result file has the reference's shape."""
    cfg = os.path.join(ROOT, "samples", "config_hash.json")
    out = tmp_path / "bench_result_ours.json"
    r = subprocess.run([exe, "--bench", "synthetic:256x256", cfg, str(out), "--batches", "16,14"], capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.load(open(out))
    assert set(res) == {"fully_fused", "cutlass"}
    for method in res:
        assert [row["batch_size"] for row in res[method]] == [1 << 16, 1 << 14]
        for row in res[method]:
            assert row["training_throughput"] > 1e6 and row["inference_throughput"] > row["training_throughput"]
            assert row["psnr"] > 25.0


@pytest.mark.gpu
def test_sample_learns_a_progressive_jpeg(exe, tmp_path):
    """BASELINE config 3 trains on a JPEG (data/images/albert.jpg, progressive, grayscale).  The harness reads JPEGs itself
    (samples/jpeg_decoder.h): a progressive grayscale file written by PIL goes through decode -> gamma 2.2 -> bilinear lookup ->
    training with the hash-grid config, and the learned image approaches the decoded one."""
    Image = pytest.importorskip("PIL.Image")
    h, w = 384, 512
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 128 + 70 * np.sin(x / 23.0) * np.cos(y / 31.0) + 40 * np.sin((x + y) / 9.0) + 20 * (((x // 32 + y // 32) % 2) - 0.5)
    jpg = tmp_path / "target.jpg"
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), mode="L").save(jpg, format="JPEG", quality=92, progressive=True)
    assert b"\xff\xc2" in jpg.read_bytes()
    cfg = os.path.join(ROOT, "samples", "config_hash.json")
    final = tmp_path / "final.ppm"
    r = subprocess.run([exe, str(jpg), cfg, "300", str(final)], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    psnr = float([l for l in r.stdout.splitlines() if l.startswith("PSNR")][-1].split(":")[1].split()[0])
    assert psnr > 30.0, r.stdout
    # reference.ppm is the lookup of the decoded file at the pixel centres: the decoder's output, gamma there and back
    ref = open(tmp_path / "reference.ppm", "rb").read()
    assert ref.startswith(b"P6\n512 384\n255\n")
    got = np.frombuffer(ref[len(b"P6\n512 384\n255\n"):], dtype=np.uint8).reshape(h, w, 3).astype(np.int32)
    want = np.asarray(Image.open(jpg).convert("L"), dtype=np.int32)
    assert np.abs(got[:, :, 0] - want).max() <= 3 and np.array_equal(got[:, :, 0], got[:, :, 1])
