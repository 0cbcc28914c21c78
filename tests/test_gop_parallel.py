"""CPU, world_size 2 and 8 over gloo: GOP sharding, the GOP-0 checkpoint hand-off and the MAX-over-ranks timing."""
import os
import pytest
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


def _pull_worker(rank, world, init_file, work_dir, out_dir):
    """schedule='pull' + prepare_fn: ranks >= 1 stage their first GOP BEFORE the GOP-0 checkpoint exists (phase-A overlap),
    every GOP >= 1 is taken exactly once from the claim-file queue, rank 0 joins the queue after GOP 0."""
    import time
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    groups = gp.split_gops(5 * 4 + 2, 4)                     # 6 GOPs, the last one short
    ck_path = os.path.join(work_dir, 'gop_0_3', 'model.pth')
    staged_before_ckpt = []

    def prepare_fn(group):
        staged_before_ckpt.append((group[0], not os.path.exists(ck_path)))
        return {'staged': group[0]}

    def first_fn(group, prepared):
        assert prepared == {'staged': 0}
        time.sleep(0.5)                                      # rank 1 stages its first GOP meanwhile
        return {'model': {'w': torch.ones(2)}, 'result': 'gop0'}

    def other_fn(group, ckpt, prepared):
        assert prepared == {'staged': group[0]} and float(ckpt['model']['w'][0]) == 1.0
        time.sleep(0.05)
        return {'rank': rank, 'first': group[0]}

    res = gp.run_sequence(groups, work_dir, first_fn, other_fn, rank, world, dist, prepare_fn=prepare_fn, schedule='pull')
    torch.save({'res': res, 'staged': staged_before_ckpt}, os.path.join(out_dir, 'p%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pull_schedule_and_phase_a_overlap():
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, 'init')
        mp.spawn(_pull_worker, args=(2, init_file, os.path.join(tmp, 'work'), tmp), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp, 'p0.pt'), weights_only=False)
        r1 = torch.load(os.path.join(tmp, 'p1.pt'), weights_only=False)
        done = sorted([g for g in r0['res'] if g != 0] + list(r1['res']))
        assert done == [1, 2, 3, 4, 5] and 0 in r0['res']                    # every GOP exactly once
        assert not (set(r0['res']) & set(r1['res']))
        assert r1['staged'][0] == (4, True)                                 # rank 1 staged GOP 1 (frames 4..) before the checkpoint existed
        assert len(r0['res']) >= 2 and len(r1['res']) >= 2                  # both ranks pulled from the queue in phase B
        assert os.listdir(os.path.join(tmp, 'work', 'claims'))


def test_rank0_failure_releases_waiting_ranks(tmp_path):
    """A failing GOP 0 leaves an error marker; a waiting rank raises instead of polling for ever."""
    groups = gp.split_gops(8, 4)
    work = str(tmp_path / 'work')
    import pytest

    def boom(group):
        raise ValueError('gop 0 failed')

    with pytest.raises(ValueError):
        gp.run_sequence(groups, work, boom, lambda g, c: None, rank=0, world=2)
    with pytest.raises(RuntimeError):
        gp.run_sequence(groups, work, boom, lambda g, c: None, rank=1, world=2)


def _pull8_worker(rank, world, init_file, work_dir, out_dir, fail):
    """The real BASELINE config[2] split (300 frames, GOP 32: 10 GOPs, the last one short) on 8 ranks with schedule='pull'."""
    import time
    dist.init_process_group('gloo', init_method='file://' + init_file, rank=rank, world_size=world)
    groups = gp.split_gops(300, 32)
    ck_path = os.path.join(work_dir, gp.gop_name(groups[0]), 'model.pth')
    staged = []

    def prepare_fn(group):
        staged.append((group[0], not os.path.exists(ck_path), time.time()))
        return group[0]

    def first_fn(group, prepared):
        time.sleep(1.0)                                      # the seven other ranks stage their first GOP meanwhile
        if fail:
            raise ValueError('gop 0 failed')
        return {'model': {'w': torch.ones(2)}, 'result': {'t_done': time.time()}}

    def other_fn(group, ckpt, prepared):
        assert prepared == group[0] and float(ckpt['model']['w'][0]) == 1.0
        time.sleep(0.1 * len(group) / 32.0)
        return {'rank': rank, 'first': group[0], 't_start': time.time()}

    err = None
    try:
        res = gp.run_sequence(groups, work_dir, first_fn, other_fn, rank, world, dist, prepare_fn=prepare_fn, schedule='pull')
    except (ValueError, RuntimeError) as e:
        res, err = {}, type(e).__name__
    torch.save({'res': res, 'staged': staged, 'err': err}, os.path.join(out_dir, 'q%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_pull_on_the_config2_split():
    """north_star's 8-GPU case without the hardware: every GOP exactly once, rank 0 joins the queue after GOP 0, all seven idle
    ranks stage a GOP before the checkpoint exists, the two GOPs left over go to whoever is free first."""
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_pull8_worker, args=(8, os.path.join(tmp, 'init'), os.path.join(tmp, 'work'), tmp, False), nprocs=8, join=True)
        r = [torch.load(os.path.join(tmp, 'q%d.pt' % k), weights_only=False) for k in range(8)]
        assert all(x['err'] is None for x in r)
        done = sorted(g for x in r for g in x['res'])
        assert done == list(range(10))                                       # GOP 0 + every GOP >= 1 exactly once
        assert 0 in r[0]['res']
        t_ck = r[0]['res'][0]['t_done']
        for k in range(1, 8):
            assert r[k]['staged'] and r[k]['staged'][0][1], 'rank %d did not stage before the checkpoint existed' % k
            assert len(r[k]['res']) >= 1
            assert all(v['t_start'] >= t_ck for v in r[k]['res'].values())   # nobody trains a GOP >= 1 before the hand-off
        first = sorted(x['staged'][0][0] for x in r[1:])
        assert first == [32 * g for g in range(1, 8)]                        # longest first: GOPs 1..7 claimed in phase A
        later = [g for x in r for g in x['res'] if g in (8, 9)]
        assert sorted(later) == [8, 9]                                       # the rest (incl. the short last GOP) in phase B
        assert abs(gp.ideal_speedup(gp.split_gops(300, 32), 8) - 300 / 76.0) < 1e-9


def test_eight_rank_rank0_failure_releases_seven_waiters():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_pull8_worker, args=(8, os.path.join(tmp, 'init'), os.path.join(tmp, 'work'), tmp, True), nprocs=8, join=True)
        r = [torch.load(os.path.join(tmp, 'q%d.pt' % k), weights_only=False) for k in range(8)]
        assert r[0]['err'] == 'ValueError'
        assert all(x['err'] == 'RuntimeError' and not x['res'] for x in r[1:])
        assert os.path.exists(os.path.join(tmp, 'work', 'rank0_failed'))
        import pytest
        with pytest.raises(RuntimeError):
            gp.check_failures(os.path.join(tmp, 'work'))


def test_stale_failure_markers_of_an_earlier_run_are_cleared():
    """A job that died leaves rank*_failed files in its work directory; running again in the same directory must not trip over
    them (run.py --out is reused in practice)."""
    with tempfile.TemporaryDirectory() as tmp:
        work = os.path.join(tmp, 'work')
        os.makedirs(work)
        for name in ('rank0_failed', 'rank3_failed'):
            open(os.path.join(work, name), 'w').close()
        groups = gp.split_gops(5, 2)
        res = gp.run_sequence(groups, work, lambda g: {'result': ('first', g[0])}, lambda g, ck: ('other', g[0]), schedule='pull')
        assert sorted(res) == [0, 1, 2]
        gp.check_failures(work)          # nothing left to raise about
        # and again in the same directory: the claim files of the first run must not make the second one stop after GOP 0
        res = gp.run_sequence(groups, work, lambda g: {'result': ('first', g[0])}, lambda g, ck: ('other', g[0]), schedule='pull')
        assert sorted(res) == [0, 1, 2]


def test_wait_all_done_rendezvous(tmp_path):
    """a rank that finishes BEFORE another one fails must not walk into the final collective.  wait_all_done returns
    only when every rank left its done marker and raises as soon as a failure marker appears, also one written later."""
    import threading
    work = str(tmp_path / 'work')
    os.makedirs(work)
    for r in range(3):
        open(os.path.join(work, 'rank%d_done' % r), 'w').close()
    gp.wait_all_done(work, 3, timeout_s=5)                       # all there: returns
    with pytest.raises(RuntimeError, match='still running'):
        gp.wait_all_done(work, 4, poll_s=0.01, timeout_s=0.2)    # rank 3 never finishes: the optional timeout says who
    threading.Timer(0.2, lambda: open(os.path.join(work, 'rank3_failed'), 'w').close()).start()
    with pytest.raises(RuntimeError, match='rank3_failed'):
        gp.wait_all_done(work, 4, poll_s=0.01, timeout_s=10)     # ... and fails late: the waiting ranks raise, nobody hangs


def _doomed_rank(work, started):
    """A rank that registers, starts working and is killed by a signal: no done marker, no failed marker."""
    import time
    gp.mark_alive(work, 1)
    started.set()
    time.sleep(60)


def test_wait_all_done_notices_a_rank_killed_by_a_signal(tmp_path):
    """a SIGKILLed rank (GPU-fault abort, out-of-memory kill) runs no `except` block and leaves no marker; the
    survivors must not poll for ever.  wait_all_done probes the process named in rank<N>_pid and raises once it is gone."""
    import signal
    import time
    work = str(tmp_path / 'work')
    os.makedirs(work)
    ctx = mp.get_context('spawn')
    started = ctx.Event()
    p = ctx.Process(target=_doomed_rank, args=(work, started))
    p.start()
    assert started.wait(60)
    gp.mark_alive(work, 0)
    gp.mark_done(work, 0)
    with pytest.raises(RuntimeError, match='still running'):      # alive and working: only the bound ends the wait
        gp.wait_all_done(work, 2, poll_s=0.01, timeout_s=0.3, grace_s=0.1)
    os.kill(p.pid, signal.SIGKILL)
    t0 = time.time()
    with pytest.raises(RuntimeError, match='rank 1 .* is gone'):
        gp.wait_all_done(work, 2, poll_s=0.01, timeout_s=60, grace_s=0.2)     # (the unreaped child is a zombie: counted as gone)
    assert time.time() - t0 < 10
    p.join()
    assert gp.wait_all_done.__defaults__[1] is not None           # the default wait is bounded


def _doomed_rank0(work, started):
    """Rank 0 of a two-rank job (no torch.distributed: a fresh directory) that is killed by a signal in the middle of GOP 0."""
    import time

    def first(group):
        started.set()
        time.sleep(60)
        return {'result': 0}

    gp.run_sequence(gp.split_gops(8, 4), work, first, lambda g, ck: 1, rank=0, world=2)


def test_phase_a_wait_notices_rank0_killed_by_a_signal(tmp_path):
    """rank 0 SIGKILLed during GOP 0 (the longest single phase) writes neither the checkpoint nor rank0_failed; ranks
    >= 1, polling for the checkpoint, probe the process named in rank0_pid and raise instead of waiting out the 24 h timeout."""
    import signal
    import time
    work = str(tmp_path / 'work')
    ctx = mp.get_context('spawn')
    started = ctx.Event()
    p = ctx.Process(target=_doomed_rank0, args=(work, started))
    p.start()
    assert started.wait(60)
    ck = os.path.join(work, gp.gop_name([0, 1, 2, 3]), 'model.pth')
    pidf = os.path.join(work, 'rank0_pid')
    assert os.path.exists(pidf)
    with pytest.raises(TimeoutError):                              # alive and training: only the bound ends the wait
        gp.wait_for_file(ck, timeout_s=0.3, writer_pid_file=pidf, grace_s=0.1)
    os.kill(p.pid, signal.SIGKILL)
    t0 = time.time()
    with pytest.raises(RuntimeError, match='rank 0 .* is gone'):
        gp.run_sequence(gp.split_gops(8, 4), work, None, lambda g, ck: 1, rank=1, world=2)       # the real phase-A wait of rank 1
    assert time.time() - t0 < 20
    p.join()
    # a checkpoint that appeared while rank 0 exited normally is NOT an error: the file wins over the liveness probe
    os.makedirs(os.path.dirname(ck), exist_ok=True)
    open(ck, 'w').close()
    gp.wait_for_file(ck, timeout_s=1, writer_pid_file=pidf, grace_s=0.0)


def test_done_marker_is_the_callers_when_asked(tmp_path):
    """run.py writes rank<N>_done itself, after its device synchronisation: run_sequence(done_marker=False) leaves only the pid file."""
    work = str(tmp_path / 'work')
    groups = gp.split_gops(8, 4)
    gp.run_sequence(groups, work, lambda g: {'result': 0}, lambda g, ck: 1, done_marker=False)
    assert os.path.exists(os.path.join(work, 'rank0_pid')) and not os.path.exists(os.path.join(work, 'rank0_done'))
    gp.mark_done(work, 0)
    gp.wait_all_done(work, 1, timeout_s=5)
