"""GOP-level sharding across GPUs: one process per GPU, no data-path collective.

main.py:83-104 splits the sequence into GOPs; GOP 0 trains from scratch and every later GOP warm-starts from GOP 0's
model + optimiser state (main.py:99-104,241-248), so GOP 0 is a serial prefix and GOPs 1..G-1 are independent.
Phase A: rank 0 overfits GOP 0 and writes ``model.pth``; phase B: GOPs 1..G-1 are dealt to the ranks (longest first,
round-robin), each rank overfits + encodes its own GOPs.  The only cross-rank traffic is the checkpoint file and a
barrier / MAX-reduce of wall times for reporting (torch.distributed: RCCL on GPUs, gloo in the CPU tests).
"""
import os
import time

import torch


def split_gops(frame_num, gop_size):
    """main.py:83-88: [[0..gop-1], [gop..2gop-1], ...] (the last GOP may be short)."""
    return [list(range(i, min(i + gop_size, frame_num))) for i in range(0, frame_num, gop_size)]


def gop_name(group):
    return 'gop_%d_%d' % (group[0], group[-1])


def assign_gops(groups, world):
    """Phase-B assignment: GOPs 1..G-1 sorted by length (desc, stable) and dealt round-robin.  Returns a list of
    per-rank lists of GOP indices.  Deterministic, identical on every rank."""
    order = sorted(range(1, len(groups)), key=lambda g: (-len(groups[g]), g))
    per_rank = [[] for _ in range(world)]
    for i, g in enumerate(order):
        per_rank[i % world].append(g)
    return per_rank


def ideal_speedup(groups, world):
    """Whole-sequence speed-up bound with the serial GOP-0 prefix, in units of frames."""
    total = sum(len(g) for g in groups)
    per_rank = assign_gops(groups, world)
    phase_b = max((sum(len(groups[g]) for g in lst) for lst in per_rank), default=0)
    return total / float(len(groups[0]) + phase_b)


def wait_for_file(path, timeout_s=3600.0, poll_s=0.05):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            raise TimeoutError('checkpoint %s did not appear' % path)
        time.sleep(poll_s)


def run_sequence(groups, work_dir, first_fn, other_fn, rank=0, world=1, dist=None):
    """Drives the two phases.  first_fn(group) -> checkpoint object (GOP 0, rank 0 only);
    other_fn(group, checkpoint) -> result for GOPs >= 1.  Returns {gop_index: result} of THIS rank.
    The checkpoint crosses ranks through ``work_dir/<gop_0>/model.pth`` (atomic rename), like the reference."""
    results = {}
    ck_dir = os.path.join(work_dir, gop_name(groups[0]))
    ck_path = os.path.join(ck_dir, 'model.pth')
    if rank == 0:
        os.makedirs(ck_dir, exist_ok=True)
        ckpt = first_fn(groups[0])
        results[0] = ckpt.get('result') if isinstance(ckpt, dict) else None
        tmp = ck_path + '.tmp.%d' % os.getpid()
        torch.save(ckpt, tmp)
        os.replace(tmp, ck_path)
    if dist is not None and world > 1:
        dist.barrier()
    else:
        wait_for_file(ck_path)
    mine = assign_gops(groups, world)[rank]
    if mine:
        ckpt = torch.load(ck_path, map_location='cpu', weights_only=False)
        for g in mine:
            results[g] = other_fn(groups[g], ckpt)
    return results


def max_over_ranks(value, dist=None, device='cpu'):
    """Wall-clock reporting: MAX over ranks (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
