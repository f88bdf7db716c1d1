"""Host-side native code of libbaybe_hip.so that needs no device: the complete base-sample draw (``bbh_sobol_normal``, the CPU twin of
the device draw ``bbh_sobol_normal_dev``) against torch's engine and torch.erfinv - what BoTorch's ``SobolQMCNormalSampler`` produces
(sampler built through baybe/acquisition/_builder.py:195-334) -, and the content key of the resident candidate matrix."""

import ctypes as C
import math

import numpy as np
import pytest


def _reference_draw(S, q, seed):
    import torch

    u = torch.quasirandom.SobolEngine(dimension=q, scramble=True, seed=seed).draw(S, dtype=torch.float64)
    v = 0.5 + (1 - torch.finfo(torch.float64).eps) * (u - 0.5)
    return (torch.erfinv(2 * v - 1) * math.sqrt(2)).numpy()


@pytest.mark.parametrize("S,q,seed", [(1, 1, 0), (9, 1, 3), (33, 7, 123456), (130, 41, 999), (512, 6, 1234), (2048, 96, 77), (257, 200, 987654),
                                      (64, 3, 2**32 + 5)])
def test_native_base_samples_equal_the_samplers(S, q, seed):
    """MT19937 scrambling bits, Owen scrambling, Gray-code walk, first point in single precision, clamp and inverse error function:
    a wrong bit anywhere gives unrelated points; agreement is to the last bits of erfinv (torch's is the vendor library's)."""
    from baybe_amd import engine

    got = engine.sobol_normal_native(S, q, seed)
    want = _reference_draw(S, q, seed & 0xFFFFFFFF if seed >= 2**32 else seed)
    assert got.shape == want.shape == (S, q)
    assert np.allclose(got, want, rtol=2e-15, atol=0), np.abs(got / want - 1).max()
    assert np.allclose(got, engine.sobol_normal_base_samples(S, q, seed & 0xFFFFFFFF if seed >= 2**32 else seed), rtol=2e-15, atol=0)


def test_native_draw_self_check_passes_here():
    from baybe_amd import engine

    assert engine._native_sobol_usable()


def _key(*bufs):
    from baybe_amd import _lib

    lib = _lib.load_library()
    arrs = [np.frombuffer(b, dtype=np.uint8) for b in bufs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    lens = (C.c_int64 * len(arrs))(*[a.size for a in arrs])
    return int(lib.bbh_content_key(ptrs, lens, len(arrs), 1))


def test_content_key_has_no_zero_collapse():
    """in a bare multiply-fold a word equal to the lane secret zeroes the product, and the next word - and the lane's
    history - no longer matter.  With the protected fold every word of such a buffer still changes the key."""
    secrets = [0x2D358DCCAA6C78A5, 0x8BB84B93962EACC9, 0x4B33A62ED433D4A3, 0x4D5A2DA51DE1AA47]
    base = np.zeros(64, dtype=np.uint64)  # eight 64-byte stripes
    for lane, sec in enumerate(secrets):
        base[8 + 2 * lane] = sec  # stripe 1: the first word of every lane equals its secret
    keys = {_key(base.tobytes())}
    for pos in range(64):  # flip one bit of any word: a new key every time
        b = base.copy()
        b[pos] ^= np.uint64(1) << np.uint64(17)
        keys.add(_key(b.tobytes()))
    assert len(keys) == 65
    # history before the collapsing stripe still matters
    b = base.copy()
    b[1] = 12345
    assert _key(b.tobytes()) != _key(base.tobytes())
    # the same bytes split differently over buffers key differently only through the piece structure - but equal splits agree
    raw = np.random.default_rng(0).integers(0, 255, size=5_000_000, dtype=np.uint8).tobytes()
    assert _key(raw) == _key(raw) and _key(raw[:100], raw[100:]) == _key(raw[:100], raw[100:])
    edited = bytearray(raw)
    edited[4_999_999] ^= 1
    assert _key(bytes(edited)) != _key(raw)
