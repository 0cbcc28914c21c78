"""Shared by the legs of bench.py: CLI, constants, logging, small timing helpers."""
import argparse
import ctypes   # noqa: F401
import json     # noqa: F401
import os
import sys      # noqa: F401
import time     # noqa: F401

import numpy as np   # noqa: F401
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EPOCHS = 10          # first_epoch / others_epoch of BASELINE config[1]
PROF_EVERY = int(os.environ.get('LINR_BENCH_PROF_EVERY',
    8))          # live kernel timing samples every 8th timed step (every step when --steps <= 32)
TABLE_STEPS = 32     # fully instrumented extra steps behind the overfit (per-kernel table)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=320)
    ap.add_argument('--warmup', type=int, default=32)
    ap.add_argument('--ramp-s', dest='ramp_s', type=float, default=1.0,
                    help='seconds of untimed steps before the warm-up steps (clock ramp of a fresh box); 0 disables')
    ap.add_argument('--config', default='loot10', help='synthetic sequence (linr_pcgc_amd.synthetic.CONFIGS)')
    ap.add_argument('--gop', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sequence', action='store_true', help='headline = the whole BASELINE config[2] sequence (strong scaling)')
    ap.add_argument('--no-sequence', action='store_true', help='skip the config[2] sequence leg after the headline')
    ap.add_argument('--seq-frames', type=int, default=300)
    ap.add_argument('--seq-epochs', type=int, default=EPOCHS)
    ap.add_argument('--seq-decode-frames', type=int, default=1, help='frames per GOP decoded and checked in the sequence leg')
    ap.add_argument('--cpu-sample-rows', type=int, default=0, help='0 = whole frame 0')
    return ap.parse_args()


def _time_launches(go, iters):
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / iters


def _popcount32(t):
    """Set bits per element of an int32 tensor (the 27-bit masks of the compressed kernel map)."""
    v = t.to(torch.int64) & 0xFFFFFFFF
    v = v - ((v >> 1) & 0x55555555)
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333)
    v = (v + (v >> 4)) & 0x0F0F0F0F
    return (v * 0x01010101 >> 24) & 0xFF


def steps_done_so_far(steps, rest, total):
    """True when the run covered exactly one complete overfit (the best-epoch bookkeeping is per overfit)."""
    return steps + rest == total


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print('[bench %7.1fs] %s' % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def host_threads():
    """CPU threads this process may really use (the GPU box gives one GPU a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))
