"""Pins the oracle's building blocks: elementary functions against numpy (fp64), Philox4x32-10 against the
known-answer vectors of Random123 (kat_vectors: philox4x32 10 rounds)."""
import ctypes as C

import numpy as np

from oracle_util import fptr


def test_elementary_functions(oracle):
    d = oracle.dll
    x = np.linspace(-7.5, 7.5, 200001).astype(np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    d.tfo_sincos(fptr(x), fptr(s), fptr(c), len(x))
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 1.5e-7
    y = np.empty_like(x)
    x = np.linspace(-60, 60, 200001).astype(np.float32)
    d.tfo_exp(fptr(x), fptr(y), len(x))
    assert np.abs(y / np.exp(x.astype(np.float64)) - 1).max() < 2e-7
    x = np.linspace(-1, 1, 200001).astype(np.float32)
    d.tfo_asin(fptr(x), fptr(y), len(x))
    assert np.abs(y - np.arcsin(x.astype(np.float64))).max() < 3e-7
    x = np.exp(np.linspace(-30, 30, 200001)).astype(np.float32)
    d.tfo_log(fptr(x), fptr(y), len(x))
    ref = np.log(x.astype(np.float64))
    assert (np.abs(y - ref) < 2e-7 * np.maximum(1.0, np.abs(ref))).all()


def test_newton_reciprocal_and_rsqrt(oracle):
    """The integer-seeded Newton reciprocal / reciprocal square root of the physics-internal scalings: ~1 ulp / ~2 ulp
    over the whole range they are used on (1e-12 ... 1e12)."""
    d = oracle.dll
    x = np.exp(np.linspace(np.log(1e-12), np.log(1e12), 400001)).astype(np.float32)
    y = np.empty_like(x)
    d.tfo_rcp(fptr(x), fptr(y), len(x))
    assert np.abs(y.astype(np.float64) * x.astype(np.float64) - 1).max() < 1.2e-7
    d.tfo_rsqrt(fptr(x), fptr(y), len(x))
    assert np.abs(y.astype(np.float64) * np.sqrt(x.astype(np.float64)) - 1).max() < 2.5e-7


def test_philox_known_answers(oracle):
    def ph(ctr, key):
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        oracle.dll.tfo_philox_raw(c, k, o)
        return list(o)
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
