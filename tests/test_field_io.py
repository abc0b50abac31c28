"""Field export: Project::file_write's PNG and the VectorField speed shading."""
import struct
import zlib

import numpy as np
import pytest


def _read_png_gray8(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, W, H = 8, b"", None, None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)     # chunk CRC
        if typ == b"IHDR":
            W, H, depth, colour = struct.unpack(">IIBB", body[:10])
            assert (depth, colour) == (8, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, W + 1)      # checks the Adler-32 too
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:]


def test_write_png_matches_reference_conversion(tmp_path):
    """Project.cpp:58-62: `int a = 255.0f*image[i]; data[i] = a;` -- truncation towards zero, byte wrap-around"""
    from demiurge_amd import capi
    rng = np.random.default_rng(0)
    f = rng.uniform(-0.5, 1.5, (70, 333)).astype(np.float32)          # 70*334 > 65535: more than one stored block
    capi.write_png(tmp_path / "a.png", f, clamp=False)
    want = (np.float32(255.0) * f).astype(np.int32).astype(np.uint8)   # C int conversion truncates; uint8 wraps
    assert np.array_equal(_read_png_gray8(tmp_path / "a.png"), want)
    capi.write_png(tmp_path / "b.png", f, clamp=True)
    want = (np.float32(255.0) * np.clip(f, 0, 1)).astype(np.int32).astype(np.uint8)
    assert np.array_equal(_read_png_gray8(tmp_path / "b.png"), want)


@pytest.mark.gpu
def test_speed_shade(tmp_path):
    from demiurge_amd import capi, synth
    W, H = 256, 128
    ctx = capi.Context(0)
    oc = capi.Ocean(ctx, capi.make_grid(W, H))
    img = synth.terrain(W, H)
    v = synth.velocity(W, H, scale=0.05)
    v[img > 0] = 0
    oc.set_terrain(img)
    oc.set_velocity(v)
    shade = oc.speed_shade(land_value=0.25)
    speed = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]).astype(np.float32))
    want = np.where(img <= 0, np.float32(1) - speed * np.float32(8), np.float32(0.25)).astype(np.float32)
    assert np.array_equal(shade, want)
    capi.write_png(tmp_path / "speed.png", shade)
    assert _read_png_gray8(tmp_path / "speed.png").shape == (H, W)
    ctx.close()
