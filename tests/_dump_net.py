"""Helper of test_gpu_parity.py: runs one forward + backward of the network on the golden shell and dumps the results.

Executed as a child process so that the executor's env switches (LINR_BATCHED, LINR_JOIN_BLOCK_IN, LINR_CONV_MFMA, LINR_FUSED_BWD),
which the library reads once per process, can be compared against each other bit for bit.
usage: python tests/_dump_net.py <golden npz> <out npz>
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(golden, out):
    import linr_pcgc_amd  # noqa: F401
    from linr_pcgc_amd import engine
    from linr_pcgc_amd.model_core import LINR_PCGC_Model
    g = np.load(golden)
    scales = [{'coord': g['s%d_coord' % s], 'occ': g['s%d_occ' % s], 'offset_tensor': g['s%d_offset' % s], 'scale_idx': s}
              for s in range(int(g['scale_num']))]
    torch.manual_seed(8807)
    model = LINR_PCGC_Model({'scale_num': 5, 'in_channel': 7, 'hidden_channel_conv': 8, 'block_layers': 1, 'outstage': 8,
                             'instage': 1}).cuda()
    frame = model.make_frame(scales)
    flat = model.flat_parameters()
    probs = torch.empty((8, frame.rows), dtype=torch.float32, device='cuda')
    bits = torch.zeros(1, dtype=torch.float64, device='cuda')
    engine.net_forward(frame, flat, 0, 8, probs, bits)
    grads = torch.zeros_like(flat)
    engine.net_backward(frame, flat, grads, 1.0 / len(g['ori']))
    torch.cuda.synchronize()
    np.savez(out, probs=probs.cpu().numpy(), bits=bits.cpu().numpy(), grads=grads.cpu().numpy())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
