"""Oracle (test infrastructure): weight quantiser + Laplace model stream of the reference.

Restates model_compression/model_size_est.py: quant_uniform2 :72-91, Laplace parameters :409-410,
the CDF construction with its trailing-zero quirk :466-482 and de-quantisation :566-568.
Pinned by loot/gop_32_62/70/side_info.json and the model stream implied by 70/result.json (35,319 bytes with the CPU's pdf; the
artefact is also consistent with 35,320 bytes from a CUDA pdf that differs in the last bit: tests/test_oracle_golden.py).
"""
import numpy as np
import torch
from . import ac


def quant_uniform2(params, bitdepth=8):
    """float32 torch arithmetic exactly as the reference (min/max/round in fp32)."""
    params = torch.as_tensor(params, dtype=torch.float32)
    min_n, max_n = params.min(), params.max()
    rng = max_n - min_n
    sym_max = float(np.ceil(2 ** bitdepth) - 1)
    q = torch.round((params - min_n) / rng * sym_max)
    recon = q / sym_max * rng + min_n
    return q, recon, min_n, max_n


def laplace_params(q):
    mu = torch.round(q.mean())
    b = torch.round((q - mu).abs().mean())
    return mu, b


def laplace_cdf(mu, b, bitdepth=8):
    """pdf over 0..2^bitdepth-1, normalised; cdf = cat(cumsum(pdf), [0]) - note: does NOT start at 0."""
    x = torch.arange(float(np.ceil(2 ** bitdepth)))
    pdf = torch.exp(-torch.abs(x - mu) / b) / (2 * b)
    pdf = pdf / pdf.sum()
    cdf = torch.cumsum(pdf, dim=-1).to(torch.float32)
    return torch.cat([cdf, torch.zeros(1, dtype=torch.float32)])


def encode_model(params, bitdepth=8):
    q, recon, min_n, max_n = quant_uniform2(params, bitdepth)
    mu, b = laplace_params(q)
    cdf = laplace_cdf(mu, b, bitdepth).numpy()
    cdf_int = np.broadcast_to(ac.cdf_float_to_int(cdf[None, :]), (len(q), len(cdf)))
    data = ac.encode_int_cdf(cdf_int, q.numpy().astype(np.int16))
    return {'bytes': data, 'mu': float(mu), 'b': float(b), 'min_param': float(min_n), 'max_param': float(max_n),
            'symbols': q.numpy().astype(np.uint8), 'recon': recon}


def decode_model(data, n, mu, b, min_param, max_param, bitdepth=8):
    cdf = laplace_cdf(torch.tensor(mu), torch.tensor(b), bitdepth).numpy()
    cdf_int = np.broadcast_to(ac.cdf_float_to_int(cdf[None, :]), (n, len(cdf)))
    sym = ac.decode_int_cdf(cdf_int, data)
    sym_max = float(np.ceil(2 ** bitdepth) - 1)
    q = torch.tensor(sym.astype(np.float32))
    rng = torch.tensor(max_param, dtype=torch.float32) - torch.tensor(min_param, dtype=torch.float32)
    return q / sym_max * rng + torch.tensor(min_param, dtype=torch.float32), sym
