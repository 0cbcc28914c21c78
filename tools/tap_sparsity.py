"""Live taps per 64-row tile of the 3x3x3 kernel map under different row orders (CPU only, numpy).

A convolution kernel that skips a tap for a whole wave needs the tap absent in ALL 64 rows of the wave's tile.  Per row about half
of the 27 taps are present (K_eff); what a tile can skip depends on which rows share a tile:
  x-major      the storage order of the frame (tiles = 64 consecutive rows)
  morton       rows in Z-curve order, tiles = 64 consecutive rows of it
  win W        rows sorted by their 27-bit mask inside windows of W consecutive x-major rows (VERDICT r3's proposal)
  blk W        rows sorted by their mask inside blocks of W consecutive rows of the Z-curve order: a block is a compact patch of the
               surface (one orientation, few distinct masks) AND small enough to stage in LDS with its one-voxel halo
Also printed for the blk orders: halo rows per block (the neighbours outside the block that must be staged with it).

    python tools/tap_sparsity.py [config ...]      # default: all four stand-ins; output kept in profiles/r04_tap_sparsity.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linr_pcgc_amd import synthetic          # noqa: E402
from oracle import octree                    # noqa: E402  (measurement tool, not product code)

_PC9 = np.array([bin(v).count('1') for v in range(1 << 9)])


def popcount27(m):
    return _PC9[m & 511] + _PC9[(m >> 9) & 511] + _PC9[(m >> 18) & 511]


def spread3(v):
    out = np.zeros_like(v)
    for b in range(12):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def live_per_tile(mask, order, tile=64):
    m = mask[order]
    pad = (-len(m)) % tile
    m = np.concatenate([m, np.zeros(pad, np.int64)]).reshape(-1, tile)
    return popcount27(np.bitwise_or.reduce(m, axis=1))


def sorted_in_chunks(mask, base_order, w):
    order = base_order.copy()
    for s in range(0, len(order), w):
        seg = order[s:s + w]
        order[s:s + w] = seg[np.argsort(mask[seg], kind='stable')]
    return order


def scale_report(coord, name):
    c = coord.astype(np.int64)
    n = len(c)
    nbr = octree.neighbour_table(c)
    pres = nbr >= 0
    mask = (pres.astype(np.int64) << np.arange(27)).sum(1)
    ident = np.arange(n)
    mort = np.argsort((spread3(c[:, 0]) << 2) | (spread3(c[:, 1]) << 1) | spread3(c[:, 2]), kind='stable')
    cols = [('x-major', live_per_tile(mask, ident).mean()), ('morton', live_per_tile(mask, mort).mean())]
    for w in (1024, 4096, 16384, n):
        cols.append(('win %s' % ('all' if w == n else w), live_per_tile(mask, sorted_in_chunks(mask, ident, w)).mean()))
    halo = {}
    for w in (256, 512, 1024):
        o = sorted_in_chunks(mask, mort, w)
        cols.append(('blk %d' % w, live_per_tile(mask, o).mean()))
        hs = []
        for s in range(0, n, max(w, (n // 64 // w) * w or w)):          # a sample of blocks
            own = mort[s:s + w]
            nn = nbr[own].ravel()
            hs.append(len(np.setdiff1d(np.unique(nn[nn >= 0]), own)))
        halo[w] = (float(np.mean(hs)), int(np.max(hs)))
    line = '%-10s rows %8d  K_eff %5.2f  masks %5d | ' % (name, n, pres.sum(1).mean(), len(np.unique(mask)))
    line += '  '.join('%s %5.2f' % kv for kv in cols)
    line += ' | halo rows per block (mean / max of a sample): ' + '  '.join('%d: %.0f / %d' % (w, halo[w][0], halo[w][1]) for w in halo)
    return line, n, {k: v for k, v in cols}


def main():
    configs = sys.argv[1:] or ['sphere8', 'loot10', 'andrew10', 'owlii11']
    for cfg in configs:
        fr = octree.prepare_frame(synthetic.sequence_frame(cfg, 0), None, 64)
        tot, acc = 0, {}
        for s in fr['scales']:
            if len(s['coord']) < 64:
                continue
            line, n, cols = scale_report(s['coord'], '%s/s%d' % (cfg, s['scale_idx']))
            print(line, flush=True)
            tot += n
            for k, v in cols.items():
                acc[k] = acc.get(k, 0.0) + v * n
        print('%-10s rows %8d  row-weighted over the scales: ' % (cfg, tot) + '  '.join('%s %5.2f' % (k, v / tot) for k, v in acc.items()))
        print()


if __name__ == '__main__':
    main()
