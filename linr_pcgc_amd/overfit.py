"""Per-GOP overfitting on the fast path: mirror of main.overfit_one_gop's hot loop (main.py:297-437).

The reference re-reads a pickle from disk, rebuilds ~60 coordinate maps per scale and synchronises with the host
~20 times per scale-step.  Here every frame of the GOP is resident in HBM (kernel map, features,
occupancy), a step is one stream-ordered launch sequence (forward, backward, fused Adam) and the host reads the loss
once per epoch.
"""
import torch

from .model_core import LINR_PCGC_Model, train_step
from .module_utils import prepare_frame


def gen_model(scale_num, device='cuda', seed=None, block_layers=1, hidden=8):
    """Gen_Model of main.py:97,218 (hidden = --hidden_channel_conv, main.py:520: 8 on the tuned kernels, 16 / 32 channel-blocked)."""
    if seed is not None:
        torch.manual_seed(seed)
    m = LINR_PCGC_Model({'scale_num': scale_num, 'in_channel': 7, 'hidden_channel_conv': hidden, 'block_layers': block_layers,
                         'outstage': 8, 'instage': 1})
    return m.to(device)


class Gop:
    """A group of pictures resident on one GPU: per frame the batched multi-scale Frame + side data."""

    def __init__(self, model, clouds, scale_num=None, min_point_num=64, device='cuda', block_layers=None):
        self.frames, self.point_nums, self.coord_mins, self.low_xyz, self.infos = [], [], [], [], []
        self.scale_num = scale_num
        for pts in clouds:
            fr = prepare_frame(pts, self.scale_num, min_point_num, device=device, with_offsets=False)
            if self.scale_num is None:
                self.scale_num = fr['scale_num']             # frozen from frame 0 (main.py:77-78)
            self.infos.append(fr)
        self.model_scale_num = model.scale_num if model is not None else self.scale_num
        if block_layers is None:
            block_layers = model.block_layers if model is not None else 1
        self.block_layers = int(block_layers)
        max_rows = 0
        from . import engine
        for fr in self.infos:
            f = engine.Frame(fr['all_input_info'], self.model_scale_num, device, validate=True, with_arena=False,
                             block_layers=self.block_layers)
            for i, sinfo in enumerate(fr['all_input_info']):      # the reference's per-scale dicts carry 'offset_tensor'
                sinfo['offset_tensor'] = f.offset_feat[f.scale_slice(i)]
                sinfo['xyzqsc_t'].offset_tensor = sinfo['offset_tensor']
            self.frames.append(f)
            self.point_nums.append(fr['point_num'])
            self.coord_mins.append(fr['coord_data_min'])
            self.low_xyz.append(fr['all_input_info'][-1]['coord'])
            max_rows = max(max_rows, f.rows)
        # one activation arena shared by all frames of the GOP (a step finishes before the next begins)
        from . import _lib
        arena = _lib.scratch(_lib.lib().linr_net_arena_bytes(max_rows, self.block_layers), device)
        for f in self.frames:
            f.arena = arena

    def share_train_bf16_arena(self):
        """One arena of the bf16 training executor for all frames of the GOP (a step finishes before the next begins)."""
        if all(getattr(f, 'arena_train_bf16', None) is not None for f in self.frames):
            return
        from . import _lib
        rows = max(f.rows for f in self.frames)
        nbytes = _lib.lib().linr_net_train_bf16_arena_bytes(rows, self.block_layers)
        if nbytes == 0:
            raise _lib.LinrError('the bf16 training executor supports block_layers=1 only')
        arena = _lib.scratch(nbytes + 64, self.frames[0].device)
        for f in self.frames:
            f.arena_train_bf16 = arena

    def __len__(self):
        return len(self.frames)

    def subset(self, n):
        """A view of the first n frames (shares every buffer): warm-up runs, spot checks."""
        import copy
        g = copy.copy(self)
        g.frames, g.point_nums, g.coord_mins = self.frames[:n], self.point_nums[:n], self.coord_mins[:n]
        g.low_xyz, g.infos = self.low_xyz[:n], self.infos[:n]
        return g


def staging_split(config, frame_ids, device='cuda', scale_num=None, min_point_num=64):
    """Milliseconds per frame of the three staging steps, each timed synchronously on the given synthetic frames: the generator
    (stands in for file input), the octree levels (parents, child occupancy: module_utils.prepare_frame) and the kernel maps
    (engine.Frame: neighbour search, compressed map, tiled copy, 7-neighbour features).  Measurement aid of bench.py / tools."""
    import time
    from . import engine, synthetic
    t = {'generator': 0.0, 'octree': 0.0, 'kernel_map': 0.0}
    n = 0
    for i in frame_ids:
        torch.cuda.synchronize()
        t0 = time.time()
        pts = synthetic.sequence_frame_device(config, i, device)
        torch.cuda.synchronize()
        t1 = time.time()
        fr = prepare_frame(pts, scale_num, min_point_num, device=device, with_offsets=False)
        torch.cuda.synchronize()
        t2 = time.time()
        f = engine.Frame(fr['all_input_info'], fr['scale_num'], device, validate=True, with_arena=False)
        torch.cuda.synchronize()
        t3 = time.time()
        t['generator'] += t1 - t0
        t['octree'] += t2 - t1
        t['kernel_map'] += t3 - t2
        n += 1
        del f, fr, pts
    out = {k: round(v * 1e3 / max(n, 1), 3) for k, v in t.items()}
    out['total_without_generator'] = round(out['octree'] + out['kernel_map'], 3)
    out['frames'] = n
    return out


class BestState:
    """Device-side snapshot of (parameters, Adam moments, counters, lr) of the best epoch so far - what the reference keeps in
    model.pth: it saves the checkpoint only when the epoch's mean loss improves (main.py:413-426,440-451), so the encoder codes
    with, and the GOPs >= 1 warm-start from, the BEST epoch of an overfit, not the last one.  One D2D copy of 3 x n_params floats."""

    def __init__(self, model, opt):
        self.model, self.opt = model, opt
        self.params = torch.empty_like(model.flat_parameters())
        self.exp_avg, self.exp_avg_sq = torch.empty_like(opt.exp_avg), torch.empty_like(opt.exp_avg_sq)
        self.epoch, self.loss, self.meta = -1, float('inf'), None

    def offer(self, epoch, loss):
        """Called at the end of an epoch, BEFORE the lr clamp (the reference saves first, main.py:413-437)."""
        if not loss < self.loss:
            return False
        self.params.copy_(self.model.flat_parameters())
        self.exp_avg.copy_(self.opt.exp_avg)
        self.exp_avg_sq.copy_(self.opt.exp_avg_sq)
        self.epoch, self.loss = epoch, loss
        self.meta = (self.opt.t, self.opt.t_scale.copy(), self.opt.lr, self.opt.sched_steps)
        return True

    def restore(self):
        if self.meta is None:
            return
        self.model.flat_parameters().copy_(self.params)
        self.opt.exp_avg.copy_(self.exp_avg)
        self.opt.exp_avg_sq.copy_(self.exp_avg_sq)
        self.opt.t, self.opt.t_scale, self.opt.lr, self.opt.sched_steps = self.meta[0], self.meta[1].copy(), self.meta[2], self.meta[3]


def overfit_gop(model, opt, gop, epochs, min_lr=4e-4, on_epoch=None, keep='best', info=None):
    """main.py:297-437: frames in fixed order, one optimiser + StepLR step per frame, lr clamp after each epoch.
    keep='best' (the reference's policy, main.py:413-426,440-451): model and optimiser are left in the state at the end of the epoch
    with the lowest mean loss; keep='last': in the state after the last epoch.  Returns the per-epoch mean loss (bits per point),
    like the reference logs; `info` (a dict) receives 'coded_epoch' and 'coded_loss'."""
    if keep not in ('best', 'last'):
        raise ValueError("keep must be 'best' or 'last'")
    if getattr(model, 'train_precision', 'f32') == 'bf16':
        gop.share_train_bf16_arena()
    losses = []
    dev = gop.frames[0].device
    bits = torch.zeros(len(gop), dtype=torch.float64, device=dev)         # one slot per frame: no per-step torch kernels
    pns = torch.tensor([float(pn) for pn in gop.point_nums], dtype=torch.float64, device=dev)
    best = BestState(model, opt) if keep == 'best' else None
    for epoch in range(epochs):
        bits.zero_()
        for j, (f, pn) in enumerate(zip(gop.frames, gop.point_nums)):
            train_step(model, opt, f, pn, out=bits[j:j + 1])
        # the only host sync of the epoch.  Non-finite parameters make the epoch count as diverged (loss = inf): the loss itself
        # would not show them - BCELoss's clamp at -100 (models/model_core.py:76-81) turns a NaN probability into 100 nats
        val = (bits / pns).sum() / len(gop)
        loss_mean = float(torch.where(torch.isfinite(model.flat_parameters()).all(), val, torch.full_like(val, float('inf'))))
        if best is not None:
            best.offer(epoch, loss_mean)
        opt.clamp_lr(min_lr)
        losses.append(loss_mean)
        if on_epoch is not None:
            on_epoch(epoch, loss_mean)
    import math
    if (best is not None and best.meta is None and losses) or (best is None and losses and not math.isfinite(losses[-1])):
        raise FloatingPointError('the overfit diverged: epoch losses %s (no finite epoch to keep; learning rate %g)'
                                 % (['%.4g' % x for x in losses], opt.lr))
    if best is not None:
        best.restore()          # also when the last epoch is the best: the checkpoint holds the lr from before the clamp
    if info is not None:
        info['coded_epoch'] = best.epoch if best is not None else epochs - 1
        info['coded_loss'] = best.loss if best is not None else (losses[-1] if losses else None)
    return losses


def checkpoint(model, opt, epoch, loss, bitdepth=8):
    """main.py:365-374 checkpoint dict (same keys, torch.optim.Adam state format)."""
    return {'model': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, 'epoch': epoch,
            'optimizer_state_dict': opt.state_dict(), 'loss': loss, 'bitdepth': bitdepth}


def warm_start(model, opt, ckpt):
    """main.py:241-248: GOPs >= 1 start from GOP 0's model AND optimiser state."""
    model.load_state_dict(ckpt['model'])
    opt.load_state_dict(ckpt['optimizer_state_dict'])
