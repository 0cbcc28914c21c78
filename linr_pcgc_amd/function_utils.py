"""Bitstream containers of the reference (models/function_utils.py:109-132): uint32 count | uint32 lengths | payload."""
import numpy as np


def pack_bitstream(bitstream_list, dtype='uint32'):
    lens = [len(b) for b in bitstream_list]
    if any(n >= 2 ** 32 - 1 for n in lens):
        raise ValueError('bitstream too long for a uint32 length field')
    head = np.array(len(bitstream_list), dtype=dtype).tobytes() + np.array(lens, dtype=dtype).tobytes()
    return head + b''.join(bytes(b) for b in bitstream_list)


def unpack_bitstream(bitstream_all, dtype='uint32'):
    width = np.dtype(dtype).itemsize
    total = len(bitstream_all)
    if total < width:
        raise ValueError('bitstream container of %d bytes: too short for its count field' % total)
    num = int(np.frombuffer(bitstream_all[:width], dtype=dtype)[0])
    if width * (1 + num) > total:
        raise ValueError('bitstream container announces %d streams but holds %d bytes' % (num, total))
    lens = np.frombuffer(bitstream_all[width:width * (1 + num)], dtype=dtype)
    pos = width * (1 + num)
    if pos + int(lens.astype(np.int64).sum()) > total:
        raise ValueError('bitstream container is truncated: %d bytes announced, %d present' % (pos + int(lens.astype(np.int64).sum()), total))
    out = []
    for n in lens:
        out.append(bitstream_all[pos:pos + int(n)])
        pos += int(n)
    return out
