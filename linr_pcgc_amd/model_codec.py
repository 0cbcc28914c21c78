"""Model (weight) codec: mirror of model_compression/model_size_est.py on the C++ range coder.

quant_uniform2 (model_size_est.py:72-91): global affine 8-bit quantisation of the flat parameter vector;
compress_model (:390-521): Laplace(mu, b) arithmetic coding (mode 2) vs zlib (mode 1) vs raw (mode 0);
decompress_model (:523-579).  The Laplace CDF keeps the reference's quirk (cumsum that does not start at 0 plus a
trailing 0, :470-480) because it defines the byte stream.  Streams follow torchac 0.9.3's published coder (csrc/ac.cpp;
pinned by the reference's one known-answer vector, tests/test_oracle_golden.py::test_model_stream_known_answer).
"""
import time
import zlib

import numpy as np
import torch

from . import _lib


def quant_uniform2(param_lst, bitdepth=8):
    min_n, max_n = param_lst.min(), param_lst.max()
    ten_range = max_n - min_n
    sym_max = float(np.ceil(2 ** bitdepth) - 1)
    new_p = torch.round((param_lst - min_n) / ten_range * sym_max)
    assert new_p.min() >= 0 and new_p.max() <= sym_max
    return new_p, new_p / sym_max * ten_range + min_n


def _laplace_cdf_u16(mu, b, bitdepth):
    """float32 pdf -> normalise -> cumsum -> append 0 -> torchac's int16 conversion (Lp = 2^bitdepth + 1)."""
    x = torch.arange(float(np.ceil(2 ** bitdepth)))
    pdf = torch.exp(-torch.abs(x - mu) / b) / (2 * b)
    pdf = pdf / pdf.sum()
    cdf = torch.cat([torch.cumsum(pdf, dim=-1).to(torch.float32), torch.zeros(1)])
    lp = cdf.numel()
    scaled = torch.round(cdf * float(65536 - (lp - 1))).to(torch.int64) + torch.arange(lp)
    return (scaled & 0xFFFF).numpy().astype(np.uint16)


def _ac_encode(cdf_u16, sym_i16):
    n = sym_i16.size
    cap = 4 * n + 64
    out = np.empty(cap, dtype=np.uint8)
    got = _lib.lib().linr_ac_encode_cdf16(cdf_u16.ctypes.data, len(cdf_u16), 1, sym_i16.ctypes.data, n, out.ctypes.data, cap)
    if got < 0:
        _lib.check(int(got), 'linr_ac_encode_cdf16')
    return out[:got].tobytes()


def _ac_decode(cdf_u16, data, n):
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(n, dtype=np.int16)
    _lib.check(_lib.lib().linr_ac_decode_cdf16(cdf_u16.ctypes.data, len(cdf_u16), 1, n,
                                               buf.ctypes.data if buf.size else None, buf.size, out.ctypes.data),
               'linr_ac_decode_cdf16')
    return out


def compress_params(params, bitdepth=8):
    """compress_model (model_size_est.py:390-521) on a flat float32 parameter vector (any device)."""
    params = params.detach().to('cpu', torch.float32)
    min_param, max_param = params.min(), params.max()
    quant_ret, recon_ret = quant_uniform2(params, bitdepth)
    mu = torch.round(quant_ret.mean())
    b = torch.round((quant_ret - mu).abs().mean())
    like = torch.exp(-torch.abs(quant_ret - mu) / b) / (2 * b)
    bits = float(-torch.sum(torch.log2(like))) + 2 * bitdepth
    n = quant_ret.numel()
    np_type = np.uint8 if bitdepth <= 8 else (np.uint16 if bitdepth <= 16 else np.uint32)
    quant_byte = quant_ret.numpy().astype(np_type).tobytes()
    quant_zlib = zlib.compress(quant_byte)
    bpp_zlib = len(quant_zlib) * 8 / n
    bpp_low = bpp_zlib if bpp_zlib < bitdepth else bitdepth
    enc_mode, side_info_bit = 2, 2 + 2 * 32

    def fallback():
        if bpp_low == bitdepth:
            return 0, quant_byte
        return 1, quant_zlib

    if bits / n > bpp_low or bitdepth > 8:
        enc_mode, final_bytes = fallback()
        bit_real = bpp_low * n + 2 + 2 * 32
        bit_laplace_real = float('inf')
    else:
        cdf = _laplace_cdf_u16(mu, b, bitdepth)
        encoded = _ac_encode(cdf, np.ascontiguousarray(quant_ret.numpy().astype(np.int16)))
        bit_laplace_real = len(encoded) * 8 + 2 * np.ceil(bitdepth) + 2 + 2 * 32
        if bit_laplace_real > bpp_low * n + 2 + 2 * 32:
            enc_mode, final_bytes = fallback()
            bit_real = bpp_low * n + 2 + 2 * 32
        else:
            bit_real, final_bytes = bit_laplace_real, encoded
            side_info_bit = 2 * np.ceil(bitdepth) + 2 + 2 * 32
    return {'bpp_real': bit_real / n, 'bit_real': bit_real, 'side_info_bit': side_info_bit, 'bitdepth': bitdepth,
            'enc_mode': enc_mode, 'laplace_real_bpp': bit_laplace_real / n, 'zlib_bpp': bpp_zlib,
            'min_param': float(min_param), 'max_param': float(max_param), 'mu': float(mu), 'b': float(b),
            'final_bytes': final_bytes, 'recon_ret': recon_ret, 'symbols': quant_ret.numpy().astype(np_type)}


def decompress_params(enc_out, n, with_symbols=False):
    """decompress_model (model_size_est.py:523-579): returns the de-quantised flat float32 vector (CPU)
    (and, with_symbols, the integer codes it was rebuilt from)."""
    mode, data, bitdepth = enc_out['enc_mode'], enc_out['final_bytes'], enc_out['bitdepth']
    # the raw / zlib modes hold the codes in the integer type the encoder chose for the bit depth (model_size_est.py:434-441).  The
    # reference reads them back as uint8 whatever the depth (:546-549), so its own decoder fails above 8 bits; up to 8 - its default,
    # and every shipped stream - the two agree.
    np_type = np.uint8 if bitdepth <= 8 else (np.uint16 if bitdepth <= 16 else np.uint32)
    if mode == 0:
        q = np.frombuffer(data, dtype=np_type)
    elif mode == 1:
        q = np.frombuffer(zlib.decompress(data), dtype=np_type)
    else:
        cdf = _laplace_cdf_u16(torch.tensor(float(enc_out['mu'])), torch.tensor(float(enc_out['b'])), bitdepth)
        q = _ac_decode(cdf, data, n)
    recon = torch.tensor(np.asarray(q, dtype=np.float32))
    sym_max = float(np.ceil(2 ** bitdepth) - 1)
    min_p = torch.tensor(float(enc_out['min_param']), dtype=torch.float32)
    max_p = torch.tensor(float(enc_out['max_param']), dtype=torch.float32)
    out = recon / sym_max * (max_p - min_p) + min_p
    return (out, np.asarray(q)) if with_symbols else out


def esti_model_size(model):
    """model_size_est.py:30-36: the uncompressed size of the model in bits (32 per parameter)."""
    return 32 * sum(p.numel() for p in model.parameters())


class Model_Estimate:
    """The call surface main.py / encoder.py / decoder.py / test_utils.py use (model_size_est.py:40-579)."""

    quant_uniform2 = staticmethod(quant_uniform2)

    @staticmethod
    def _fill(model, recon, symbols=None, min_param=None, max_param=None, bitdepth=8):
        """Writes the de-quantised parameters into `model`.  With 8-bit codes (the default --model_bitdepth) it also hands the model the codes themselves
        (device uint8, parameters() order) + the two range floats: the bf16 inference path (linr_net_forward_bf16) runs
        straight from them."""
        with torch.no_grad():
            model.flat_parameters().copy_(recon.to(model.flat_parameters().device))
        if symbols is not None and bitdepth == 8:          # the kernels de-quantise 8-bit codes (q / 255 * range + min); other depths: fp32 only
            model.set_quantised(torch.as_tensor(np.ascontiguousarray(symbols).astype(np.uint8)), float(min_param), float(max_param))
        return model

    @torch.no_grad()
    def compress_model(self, model, bitdepth=8, derive_new_model=False, model_ori=None):
        out = compress_params(model.flat_parameters(), bitdepth)
        out['new_model'] = None
        if derive_new_model and model_ori is not None:
            out['new_model'] = self._fill(model_ori, out['recon_ret'], out['symbols'], out['min_param'], out['max_param'], bitdepth)
        return out

    @torch.no_grad()
    def decompress_model(self, new_model, enc_out):
        recon, q = decompress_params(enc_out, new_model.flat_parameters().numel(), with_symbols=True)
        return self._fill(new_model, recon, q, enc_out['min_param'], enc_out['max_param'], enc_out['bitdepth']), recon

    @torch.no_grad()
    def estibits(self, model, new_model, bitdepth=8):
        """model_size_est.py:99-179 (main.py:20,293 binds it as esti_compress_model and checks it against compress_test before
        training): the size ESTIMATE of the coded model - Laplace entropy of the codes + 2*bitdepth, against the zlib / raw bound -
        without running the coder, and `new_model` filled with the de-quantised parameters."""
        st1 = time.time()
        flat = model.flat_parameters().detach().to('cpu', torch.float32)
        codes, recon = quant_uniform2(flat, bitdepth)
        self._fill(new_model, recon)
        n = codes.numel()
        mu = torch.round(codes.mean())
        b = torch.round((codes - mu).abs().mean())
        bits_laplace = float(-torch.sum(torch.log2(torch.exp(-torch.abs(codes - mu) / b) / (2 * b))))
        bits = bits_laplace + 2 * bitdepth
        np_type = np.uint8 if bitdepth <= 8 else (np.uint16 if bitdepth <= 16 else np.uint32)
        zlib_bpp = len(zlib.compress(codes.numpy().astype(np_type).tobytes())) * 8 / n
        bound = zlib_bpp if zlib_bpp < bitdepth else bitdepth
        enc_mode, bit_real = 2, bits + 2 + 2 * 32
        if bits / n > bound:
            enc_mode = 0 if bound == bitdepth else 1
            bit_real = bound * n + 2
        dt = time.time() - st1
        return {'new_model': new_model, 'bpp_real': bit_real / n, 'bit_real': bit_real, 'enc_mode': enc_mode,
                'laplace_bpp': bits_laplace / n, 'zlib_bpp': zlib_bpp, 'final_bytes': b'0', 'min_param': flat.min(), 'max_param': flat.max(),
                'mu': mu, 'b': b, 'enc_time': dt, 'dec_time': dt, 'recon_ret': recon}

    @torch.no_grad()
    def compress_test(self, model, new_model, bitdepth=8):
        st1 = time.time()
        out = self.compress_model(model, bitdepth)
        st2 = time.time()
        recon_model, recon = self.decompress_model(new_model, out)
        st3 = time.time()
        assert bool((out['recon_ret'] == recon).all())
        out.update(enc_time=st2 - st1, dec_time=st3 - st2, new_model=recon_model)
        return out
