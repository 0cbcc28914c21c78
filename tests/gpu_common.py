"""Helpers shared by the GPU parity tests (tests/test_gpu_*.py); the fixtures `pkg` and `shell` live in conftest.py.

Tolerances (the reference states none): fp32 HIP vs fp32 oracle
  logits  |d| <= 1e-4 + 1e-4 |x|     bits  rel <= 1e-5
  gradients, per tensor against ITS OWN largest entry: the HIP gradient must be as close to the float64 oracle as the fp32
  oracle itself is, err_hip(f64) <= max(3 * err_oracle32(f64), 1e-4 * max|g_tensor|); the direct fp32-vs-fp32 difference
  (two summation orders of sums with heavy cancellation) is only sanity-bounded: 1e-3 * max|g_tensor| at block_layers 1, 3e-3 for the deeper block_in variants
Integer / index / byte work (kernel map, streams, decoded geometry) is bit-exact.
"""
import numpy as np
import torch

from oracle import network as onet


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    return torch.device('cuda:0')


def _close(a, b, rtol, atol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), '%s: max err %.3e (tol %.3e at worst)' % (what, float(err.max()), float(tol.min()))


# ---- whole network --------------------------------------------------------------------------------------------------------
def _model_and_oracle(pkg, scale_num, seed=8807, block_layers=1):
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    torch.manual_seed(seed)
    model = LINR_PCGC_Model({'scale_num': scale_num, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': block_layers,
                             'outstage': 8, 'instage': 1})
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.cuda(), sd


def _grads_close_per_tensor(grads, sdo, rtol=1e-3, floor=1e-9, sd64=None):
    """Every tensor against ITS OWN largest gradient (a tensor whose gradients are orders of magnitude below the model's
    largest one must still be right).  A gradient entry is a sum over all rows with heavy cancellation (bias gradients
    most of all), so two fp32 evaluations in different summation orders (the oracle adds the taps in ascending order, the
    kernels column by column: common.h LINR_TAP) differ by up to ~2e-3 of the tensor's largest entry at block_layers 3
    (measured worst: 2.07e-3 on a bias gradient of 5e-4): the direct fp32-vs-fp32 bound (rtol) is only a sanity check.  The criterion proper needs sd64, the
    same leaves evaluated by the oracle in float64: the HIP gradient must be as accurate as the fp32 oracle is,
        err_hip(f64) <= max(3 * err_oracle32(f64), 1e-4 * max|g_tensor|)."""
    off, worst = 0, (0.0, '')
    for name, v in sdo.items():
        n = v.numel()
        mine = grads[off:off + n].view(v.shape).detach().double().cpu()
        ref = v.grad.detach().double()
        gmax = float(ref.abs().max())
        err = float((mine - ref).abs().max())
        assert err <= rtol * gmax + floor, 'grad %s: max err %.3e vs tolerance %.3e (own max %.3e)' % (name, err, rtol * gmax + floor, gmax)
        if sd64 is not None:
            truth = sd64[name].grad
            e_hip, e_o32 = float((mine - truth).abs().max()), float((ref - truth).abs().max())
            assert e_hip <= max(3.0 * e_o32, 1e-4 * gmax) + floor, \
                'grad %s vs float64: HIP %.3e, fp32 oracle %.3e (own max %.3e)' % (name, e_hip, e_o32, gmax)
        if gmax > 0 and err / gmax > worst[0]:
            worst = (err / gmax, name)
        off += n
    return worst
