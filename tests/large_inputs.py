"""Deterministic inputs of the hashed 512x256 goldens (tests/golden/large_512x256_full.npz): shared by the generator
(tests/golden/make_goldens.py large) and by the tests, which regenerate them instead of storing 512 KiB fields."""
import numpy as np

NOISE_MODES = ["DEFAULT", "RIDGED", "BILLOWY", "IQNoise", "SWISS", "Jordan", "Plateaus"]


def large_inputs(W=512, H=256):
    from demiurge_amd import synth
    f32 = np.float32
    img = synth.terrain(W, H)
    rough = (synth.lcg_uniform(W * H, 77).reshape(H, W)).astype(f32)
    return dict(terrain=img, heights=(img * 100 + rough * 8).astype(f32), T=(rough * 60).astype(f32),
                relief=(np.round(img * 2) * 150 + np.where(rough > 0.3, 260, 0)).astype(f32),
                flow=(np.abs(synth.divw(W, H)) * 3).astype(f32), tect=(img * 3).astype(f32))


def check(G, key, a, stride=8):
    """compare a computed field with the stored SHA-256 (and the strided sample, for a readable failure)"""
    import hashlib
    a = np.ascontiguousarray(a, np.float32)
    assert np.array_equal(a[::stride, ::stride], G[key + "_sample"]), (key, int((a[::stride, ::stride] != G[key + "_sample"]).sum()))
    assert hashlib.sha256(a.tobytes()).hexdigest() == str(G[key + "_sha256"]), key
