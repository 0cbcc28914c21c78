"""CPU, world_size 2 over gloo: GOP sharding, the GOP-0 checkpoint hand-off and the MAX-over-ranks timing."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linr_pcgc_amd import gop_parallel as gp


def test_split_and_assign():
    groups = gp.split_gops(300, 32)
    assert len(groups) == 10 and len(groups[-1]) == 12 and groups[1][0] == 32
    assert gp.gop_name(groups[1]) == 'gop_32_63'
    per = gp.assign_gops(groups, 8)
    flat = sorted(g for lst in per for g in lst)
    assert flat == list(range(1, 10))                       # every GOP >= 1 exactly once, GOP 0 is phase A
    assert per[0] == [1, 9] and per[1] == [2]               # the short last GOP is dealt last
    assert abs(gp.ideal_speedup(groups, 8) - 300 / (32 + 44)) < 1e-9
    assert abs(gp.ideal_speedup(gp.split_gops(32 + 8 * 32, 32), 8) - (32 + 256) / 64.0) < 1e-9
    assert gp.assign_gops(gp.split_gops(32, 32), 4) == [[], [], [], []]


def _worker(rank, world, init_file, work_dir, out_dir):
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    groups = gp.split_gops(7 * 4, 4)

    def first_fn(group):
        return {'model': {'w': torch.full((3,), 7.0)}, 'epoch': 9, 'result': 'gop0 by %d' % rank, 'frames': group}

    def other_fn(group, ckpt):
        assert float(ckpt['model']['w'][0]) == 7.0 and ckpt['frames'] == [0, 1, 2, 3]     # warm start from GOP 0
        return {'rank': rank, 'frames': group}

    res = gp.run_sequence(groups, work_dir, first_fn, other_fn, rank, world, dist)
    t = gp.max_over_ranks(1.0 + rank, dist)
    torch.save({'res': res, 'tmax': t}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sequence():
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, 'init')
        mp.spawn(_worker, args=(2, init_file, os.path.join(tmp, 'work'), tmp), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp, 'r0.pt'), weights_only=False)
        r1 = torch.load(os.path.join(tmp, 'r1.pt'), weights_only=False)
        assert r0['tmax'] == 2.0 and r1['tmax'] == 2.0
        assert sorted(r0['res']) == [0, 1, 3, 5] and sorted(r1['res']) == [2, 4, 6]
        assert r0['res'][0] == 'gop0 by 0'
        assert all(v['rank'] == 1 for v in r1['res'].values())
        assert r1['res'][2]['frames'] == [8, 9, 10, 11]
        assert os.path.exists(os.path.join(tmp, 'work', 'gop_0_3', 'model.pth'))
