"""Oracle (test infrastructure): ctypes front-end of torchac_port.c + torchac's float->int16 CDF conversion.

Restates torchac 0.9.3 ``encode_float_cdf`` / ``decode_float_cdf`` (third-party, pinned in enviroment.yaml:32):
    cdf_int = round(cdf_float * (2^16 - (Lp-1))).to(int16) + arange(Lp)       (wraps mod 2^16)
Call sites restated: BinaryArithmeticCoding (models/module_utils.py:8-40), cdf = [0, 1-p, 1].
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'liboracle_ac.so')
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.oracle_ac_encode.restype = ctypes.c_size_t
        lib.oracle_ac_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_size_t]
        lib.oracle_ac_decode.restype = None
        lib.oracle_ac_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def cdf_float_to_int(cdf_float):
    """torchac._convert_to_int_and_normalize(needs_normalization=True) in float32 arithmetic."""
    cdf_float = np.asarray(cdf_float, dtype=np.float32)
    lp = cdf_float.shape[-1]
    scaled = np.rint(cdf_float * np.float32(65536 - (lp - 1)))          # round half to even, like torch.round
    return ((scaled.astype(np.int64) + np.arange(lp, dtype=np.int64)) & 0xFFFF).astype(np.uint16)


def encode_int_cdf(cdf_u16, sym):
    cdf_u16 = np.ascontiguousarray(cdf_u16, dtype=np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.int16)
    n, lp = cdf_u16.shape
    assert sym.shape == (n,) and sym.min(initial=0) >= 0 and sym.max(initial=0) <= lp - 2
    cap = 4 * n + 64
    out = np.empty(cap, dtype=np.uint8)
    need = _lib().oracle_ac_encode(cdf_u16.ctypes.data, sym.ctypes.data, n, lp, out.ctypes.data, cap)
    assert need <= cap
    return out[:need].tobytes()


def decode_int_cdf(cdf_u16, data):
    cdf_u16 = np.ascontiguousarray(cdf_u16, dtype=np.uint16)
    n, lp = cdf_u16.shape
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(n, dtype=np.int16)
    _lib().oracle_ac_decode(cdf_u16.ctypes.data, n, lp, buf.ctypes.data if len(buf) else None, len(buf),
                            out.ctypes.data)
    return out


def encode_float_cdf(cdf_float, sym):
    return encode_int_cdf(cdf_float_to_int(cdf_float), sym)


def decode_float_cdf(cdf_float, data):
    return decode_int_cdf(cdf_float_to_int(cdf_float), data)


def binary_cdf(prob):
    """BinaryArithmeticCoding._get_cdf (module_utils.py:11-16): [0, 1-p, 1] in float32."""
    prob = np.asarray(prob, dtype=np.float32).reshape(-1, 1)
    return np.concatenate([np.zeros_like(prob), np.float32(1) - prob, np.ones_like(prob)], axis=1)


def encode_binary(prob, occupancy):
    return encode_float_cdf(binary_cdf(prob), np.asarray(occupancy).reshape(-1))


def decode_binary(prob, data):
    return decode_float_cdf(binary_cdf(prob), data)
