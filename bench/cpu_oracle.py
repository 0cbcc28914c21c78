"""The CPU legs of bench.py: the oracle timed on the host cores (`cpu_baseline`, kind "port") and the full-size parity check that re-uses
its gradient."""
import ctypes   # noqa: F401
import json     # noqa: F401
import os
import sys      # noqa: F401
import time     # noqa: F401

import numpy as np   # noqa: F401
import torch

from .common import EPOCHS, host_threads, log

def cpu_baseline(model_sd, gop_info, point_num, sample_rows):
    """The CPU oracle ("port": ME/torchac are not installable, the reference has no CPU path) on the host cores:
    one overfit step (forward + autograd backward + Adam) + one inference forward on frame 0.
    Also returns the oracle's bits and per-tensor gradients of that step for the full-size parity check."""
    from oracle import network as onet
    scales = []
    for s in gop_info['all_input_info']:
        scales.append({'coord': s['coord'].cpu().numpy(), 'occ': s['occ'].cpu().numpy(),
                       'offset_tensor': s['offset_tensor'].cpu().numpy(), 'scale_idx': s['scale_idx']})
    rows = sum(len(s['coord']) for s in scales)
    t0 = time.time()
    tsc = onet.to_torch_scales(scales)          # builds the kernel maps (oracle.octree.neighbour_table: sorted-key searches in numpy)
    t_kmap = time.time() - t0
    sd = {k: v.clone().requires_grad_() for k, v in model_sd.items()}
    flat_p = torch.cat([v.detach().reshape(-1) for v in sd.values()])
    m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    t0 = time.time()
    bits = onet.frame_bits(sd, tsc)
    (bits / point_num).backward()
    g = torch.cat([t.grad.reshape(-1) for t in sd.values()])
    grads = {k: t.grad.detach().clone() for k, t in sd.items()}
    onet.adam_step(flat_p, g.clone(), m, v, 1, 0.01)
    t_step = time.time() - t0
    from oracle import ac as oac
    with torch.no_grad():
        t0 = time.time()
        sdd = {k: v.detach() for k, v in sd.items()}
        outs = [onet.forward_scale(sdd, s) for s in tsc]
        t_fwd = time.time() - t0
        # the arithmetic-coder feed of encode (models/upsample.py:224-237): 8 streams per scale through the oracle's plain-C
        # restatement of torchac's coder, one thread (torchac's own encoder is serial too)
        t0 = time.time()
        ac_bytes = 0
        for s, o in zip(scales, outs):
            for k in range(8):
                ac_bytes += len(oac.encode_binary(o['probs'][k].reshape(-1).numpy(), s['occ'][:, k].astype(np.uint8)))
        t_ac = time.time() - t0
    out = {'value': round(EPOCHS * t_step + t_fwd + t_ac, 3), 'unit': 's/frame', 'cores': torch.get_num_threads(),
           'kind': 'port',
           'sample': '1 overfit step (%.2f s) + 1 forward (%.2f s) + range coding of its %d symbols (%.3f s, 1 thread, %d bytes) of '
                     'frame 0 (%d rows), x%d epochs of the step' % (t_step, t_fwd, 8 * rows, t_ac, ac_bytes, rows, EPOCHS),
           'train_step_s': round(t_step, 3), 'forward_s': round(t_fwd, 3), 'ac_s': round(t_ac, 4),
           'kernel_map_s': round(t_kmap, 3),          # once per frame, outside `value` like the GPU side's staging
           'bits_frame0_init': float(bits.detach())}
    return out, float(bits.detach()), grads


# full-size parity (frame 0, 336 k rows, initial parameters): HIP forward/backward against the oracle step the CPU baseline
# runs anyway.  bits: relative 1e-5; gradients PER TENSOR: max |d| <= 1e-3 * max |g| of that tensor
# + 1e-9 (two fp32 evaluations with 336 k-row sums in different orders and heavy cancellation; measured worst 1.2e-4 -
# the float64-anchored criterion lives in tests/test_gpu_parity.py, where the oracle is cheap enough to run twice).
PARITY_BITS_RTOL = 1e-5


PARITY_GRAD_RTOL = 3e-4


def full_size_parity(model_sd, frame, point_num, oracle_bits, oracle_grads, scale_num):
    from linr_pcgc_amd import engine, overfit
    model = overfit.gen_model(scale_num, 'cuda')
    model.load_state_dict(model_sd)
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    engine.net_forward(frame, model.flat_parameters(), 0, 8, None, bits)
    flat_g = torch.zeros_like(model.flat_parameters())
    engine.net_backward(frame, model.flat_parameters(), flat_g, 1.0 / float(point_num))
    torch.cuda.synchronize()
    got_bits = float(bits)
    worst, worst_name, off = 0.0, '', 0
    flat_g = flat_g.cpu()
    for name, p in model.state_dict().items():
        n = p.numel()
        g_hip = flat_g[off:off + n].view(p.shape)
        g_ref = oracle_grads[name]
        off += n
        gmax = float(g_ref.abs().max())
        err = float((g_hip - g_ref).abs().max())
        rel = err / (gmax + 1e-30) if gmax > 0 else (0.0 if err <= 1e-9 else float('inf'))
        if err > 1e-9 and rel > worst:
            worst, worst_name = rel, name
    bits_rel = abs(got_bits - oracle_bits) / abs(oracle_bits)
    ok = bits_rel <= PARITY_BITS_RTOL and worst <= PARITY_GRAD_RTOL
    return {'ok': bool(ok), 'bits_hip': got_bits, 'bits_oracle': oracle_bits, 'bits_rel_err': bits_rel,
            'grad_worst_rel_err_per_tensor': worst, 'grad_worst_tensor': worst_name, 'tensors': len(oracle_grads),
            'tolerance': {'bits_rel': PARITY_BITS_RTOL, 'grad_rel_to_own_tensor_max': PARITY_GRAD_RTOL}}
