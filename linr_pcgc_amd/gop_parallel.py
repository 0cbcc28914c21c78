"""GOP-level sharding across GPUs: one process per GPU, no data-path collective.

main.py:83-104 splits the sequence into GOPs; GOP 0 trains from scratch and every later GOP warm-starts from GOP 0's
model + optimiser state (main.py:99-104,241-248), so GOP 0 is a serial prefix and GOPs 1..G-1 are independent.
Phase A: rank 0 overfits GOP 0 and writes ``model.pth`` while every other rank already stages its first GOP (octrees,
kernel maps, resident in HBM).  Phase B: GOPs 1..G-1 are handed out longest first - either a fixed round-robin deal
(``schedule='static'``: deterministic, what bench.py uses so that all inputs are resident before the timed region) or
a pull from a shared queue (``schedule='pull'``: a rank claims the next GOP whenever it is free, rank 0 joins after
GOP 0; the queue is a directory of O_EXCL claim files, so it needs no collective either).  The only cross-rank traffic
is the checkpoint file; ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests) is used for a start-up barrier and
the MAX-reduce of wall times.  Waiting for the checkpoint is a host-side file poll, never a pending collective, so a
long first_epoch cannot trip the NCCL watchdog.
"""
import os
import shutil
import time

import torch


def split_gops(frame_num, gop_size):
    """main.py:83-88: [[0..gop-1], [gop..2gop-1], ...] (the last GOP may be short)."""
    return [list(range(i, min(i + gop_size, frame_num))) for i in range(0, frame_num, gop_size)]


def gop_name(group):
    return 'gop_%d_%d' % (group[0], group[-1])


def phase_b_order(groups):
    """GOPs 1..G-1, longest first (stable)."""
    return sorted(range(1, len(groups)), key=lambda g: (-len(groups[g]), g))


def assign_gops(groups, world):
    """Static phase-B assignment: phase_b_order dealt round-robin.  Returns a list of per-rank lists of GOP indices.
    Deterministic, identical on every rank."""
    per_rank = [[] for _ in range(world)]
    for i, g in enumerate(phase_b_order(groups)):
        per_rank[i % world].append(g)
    return per_rank


def ideal_speedup(groups, world):
    """Whole-sequence speed-up bound with the serial GOP-0 prefix, in units of frames."""
    total = sum(len(g) for g in groups)
    per_rank = assign_gops(groups, world)
    phase_b = max((sum(len(groups[g]) for g in lst) for lst in per_rank), default=0)
    return total / float(len(groups[0]) + phase_b)


def wait_for_file(path, timeout_s=24 * 3600.0, poll_s=0.02, alive=None, writer_pid_file=None, grace_s=2.0):
    """Host-side poll for the GOP-0 checkpoint.  `alive`: optional callable returning False when rank 0 is known to have
    failed (its error marker exists) - the wait then raises instead of hanging.  `writer_pid_file`: rank 0's rank0_pid file
    (mark_alive): a rank 0 killed by a signal during GOP 0 (GPU-fault abort, out-of-memory kill) leaves no error marker, so
    when the process it names is gone - and the checkpoint still absent after `grace_s` (the atomic rename may just have
    happened) - the wait raises instead of polling for the whole timeout."""
    t0 = time.time()
    gone_at = None
    while not os.path.exists(path):
        if alive is not None and not alive():
            raise RuntimeError('rank 0 failed before writing %s' % path)
        now = time.time()
        if writer_pid_file is not None:
            pid = _read_pid(writer_pid_file)
            if pid is None or _pid_alive(pid):
                gone_at = None                           # not started yet (no pid file) or still working
            else:
                if gone_at is None:
                    gone_at = now
                if now - gone_at > grace_s and not os.path.exists(path):
                    raise RuntimeError('rank 0 (process %d) is gone without writing %s or a failure marker: killed by a signal?'
                                       % (pid, path))
        if now - t0 > timeout_s:
            raise TimeoutError('checkpoint %s did not appear' % path)
        time.sleep(poll_s)


def _read_pid(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def _claim(claim_dir, g, rank):
    """Atomically take GOP g from the shared queue (True if this rank got it)."""
    try:
        fd = os.open(os.path.join(claim_dir, 'gop_%d' % g), os.O_CREAT | os.O_EXCL | os.O_WRONLY)
    except FileExistsError:
        return False
    os.write(fd, ('%d\n' % rank).encode())
    os.close(fd)
    return True


def mark_alive(work_dir, rank):
    """Leaves rank<N>_pid (this process's id) in work_dir: wait_all_done() tells a rank that was killed by a signal - it writes
    neither a done nor a failed marker - from one that is still working by probing that process."""
    os.makedirs(work_dir, exist_ok=True)
    tmp = os.path.join(work_dir, 'rank%d_pid.tmp.%d' % (rank, os.getpid()))
    with open(tmp, 'w') as f:
        f.write('%d\n' % os.getpid())
    os.replace(tmp, os.path.join(work_dir, 'rank%d_pid' % rank))


def mark_done(work_dir, rank):
    open(os.path.join(work_dir, 'rank%d_done' % rank), 'w').close()


def mark_failed(work_dir, rank):
    os.makedirs(work_dir, exist_ok=True)
    open(os.path.join(work_dir, 'rank%d_failed' % rank), 'w').close()


def run_sequence(groups, work_dir, first_fn, other_fn, rank=0, world=1, dist=None, prepare_fn=None, schedule='static',
                 prepared=None, done_marker=True):
    """Drives the two phases.  first_fn(group[, prepared]) -> checkpoint object (GOP 0, rank 0 only);
    other_fn(group, checkpoint[, prepared]) -> result for GOPs >= 1.  prepare_fn(group) (optional) stages a GOP's
    inputs (octrees / kernel maps in HBM) and its return value is handed to first_fn / other_fn as `prepared`; ranks
    >= 1 call it for their first GOP BEFORE the checkpoint exists (phase-A overlap).  `prepared`: {gop index: object}
    already staged by the caller (bench.py stages everything before its timed region).
    done_marker=False: the caller writes rank<N>_done itself (mark_done) once its own tail - device synchronisation, worker
    shutdown - has succeeded, and rank<N>_failed (mark_failed) if that tail raises.
    Returns {gop_index: result} of THIS rank.  The checkpoint crosses ranks through
    ``work_dir/<gop_0>/model.pth`` (atomic rename), like the reference.  With `dist` (behind a start-up barrier) or with a single rank,
    rank 0 clears what an earlier run left in work_dir; several ranks without `dist` must be given a fresh directory."""
    if schedule not in ('static', 'pull'):
        raise ValueError('schedule must be static or pull')
    prepared = dict(prepared or {})
    results = {}
    ck_dir = os.path.join(work_dir, gop_name(groups[0]))
    ck_path = os.path.join(ck_dir, 'model.pth')
    claim_dir = os.path.join(work_dir, 'claims')
    err_path = os.path.join(work_dir, 'rank0_failed')
    if rank == 0:
        if world == 1 or dist is not None:
            # what an earlier run left in this directory: failure markers (any rank's), the GOP-0 checkpoint the other ranks poll for,
            # and the claim files - with those in place a new run would find every GOP already taken and stop after GOP 0.  (Ranks
            # without a barrier between them cannot clean up safely: such callers pass a fresh directory.)
            if os.path.isdir(work_dir):
                for f in os.listdir(work_dir):
                    if f.endswith('_failed') or f.endswith('_done') or f.endswith('_pid'):
                        os.remove(os.path.join(work_dir, f))
            if os.path.exists(ck_path):
                os.remove(ck_path)
            shutil.rmtree(claim_dir, ignore_errors=True)
        os.makedirs(ck_dir, exist_ok=True)
        os.makedirs(claim_dir, exist_ok=True)
    if dist is not None and world > 1:
        dist.barrier()                                   # start-up only: nothing long-running is pending behind it
    os.makedirs(claim_dir, exist_ok=True)
    mark_alive(work_dir, rank)                           # behind the barrier: rank 0's clean-up has run

    def staged(g):
        if g not in prepared and prepare_fn is not None:
            prepared[g] = prepare_fn(groups[g])
        return prepared.pop(g, None)

    def call(fn, *a):
        p = staged(a[-1])
        args = a[:-1]
        return fn(*args, p) if (prepare_fn is not None or p is not None) else fn(*args)

    order = phase_b_order(groups)
    static_mine = assign_gops(groups, world)[rank]
    first_claim = None
    if rank == 0:
        try:
            ckpt = call(first_fn, groups[0], 0)
            results[0] = ckpt.get('result') if isinstance(ckpt, dict) else None
            tmp = ck_path + '.tmp.%d' % os.getpid()
            torch.save(ckpt, tmp)
            os.replace(tmp, ck_path)
        except BaseException:
            open(err_path, 'w').close()                  # training OR the checkpoint write failed: the waiting ranks fail too
            raise
    else:
        # phase-A overlap: stage this rank's first GOP while rank 0 still trains GOP 0
        if schedule == 'static':
            first_claim = static_mine[0] if static_mine else None
        else:
            first_claim = next((g for g in order if _claim(claim_dir, g, rank)), None)
        if first_claim is not None and prepare_fn is not None and first_claim not in prepared:
            prepared[first_claim] = prepare_fn(groups[first_claim])
        wait_for_file(ck_path, alive=lambda: not os.path.exists(err_path), writer_pid_file=os.path.join(work_dir, 'rank0_pid'))
    ckpt = None

    def load():
        return torch.load(ck_path, map_location='cpu', weights_only=False)

    try:
        _phase_b(schedule, static_mine, first_claim, order, claim_dir, rank, groups, results, load, call, other_fn)
    except BaseException:
        mark_failed(work_dir, rank)                      # wait_all_done() lets the others skip the final reduce
        raise
    if done_marker:
        mark_done(work_dir, rank)
    return results


def check_failures(work_dir):
    """Raises if any rank left a failure marker in work_dir - called before the final reductions so that the surviving ranks of
    a job without kill-on-failure supervision do not wait in a collective for a rank that is gone."""
    bad = sorted(f for f in os.listdir(work_dir) if f.endswith('_failed')) if os.path.isdir(work_dir) else []
    if bad:
        raise RuntimeError('ranks failed: %s' % ', '.join(bad))


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                                 # a killed child that nobody has reaped yet still answers the probe
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[0] != 'Z'
    except (OSError, IndexError):
        return True


def wait_all_done(work_dir, world, poll_s=0.05, timeout_s=12 * 3600.0, grace_s=2.0):
    """File rendezvous in front of the final reductions: returns once every rank of the job has left its rank<N>_done marker,
    raises as soon as ANY rank<N>_failed marker exists.  check_failures() alone only covers 'fail first, check later': a rank
    that finishes before another one fails would see no marker and wait in the collective for a rank that is gone.  A rank killed
    by a signal (GPU fault abort, out-of-memory kill) writes no marker at all: its rank<N>_pid file (mark_alive) names the
    process, and when that process is gone - and still no marker after `grace_s` - the wait raises instead of polling for ever.
    `timeout_s` bounds the wait whatever happens (default: the 12 h of the collectives' own watchdog).  Nothing is pending in a
    collective while this polls."""
    t0 = time.time()
    gone_since = {}
    while True:
        check_failures(work_dir)
        missing = [r for r in range(world) if not os.path.exists(os.path.join(work_dir, 'rank%d_done' % r))]
        if not missing:
            return
        now = time.time()
        for r in missing:
            pid = _read_pid(os.path.join(work_dir, 'rank%d_pid' % r))
            if pid is None:
                continue                                 # not started yet (or an external launcher without pid files)
            if _pid_alive(pid):
                gone_since.pop(r, None)
            elif now - gone_since.setdefault(r, now) > grace_s:
                raise RuntimeError('rank %d (process %d) is gone without a done or failed marker: killed by a signal?' % (r, pid))
        if timeout_s is not None and now - t0 > timeout_s:
            raise RuntimeError('ranks still running after %.0f s: %s' % (timeout_s, ', '.join(str(r) for r in missing)))
        time.sleep(poll_s)


def _phase_b(schedule, static_mine, first_claim, order, claim_dir, rank, groups, results, load, call, other_fn):
    ckpt = None
    if schedule == 'static':
        todo = static_mine
        if todo:
            ckpt = load()
        for g in todo:
            results[g] = call(other_fn, groups[g], ckpt, g)
    else:
        g = first_claim
        while True:
            if g is None:
                g = next((h for h in order if _claim(claim_dir, h, rank)), None)
                if g is None:
                    break
            if ckpt is None:
                ckpt = load()
            results[g] = call(other_fn, groups[g], ckpt, g)
            g = None
    return results


def max_over_ranks(value, dist=None, device='cpu'):
    """Wall-clock reporting: MAX over ranks (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def sum_over_ranks(value, dist=None, device='cpu'):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t)
