"""Fixture: the optimiser state inside the checkpoint the reference ships (loot/gop_32_62/model.pth, written by torch 1.13.1's
torch.optim.Adam at epoch 70 of a warm-started GOP) - the object GOPs >= 1 load besides the weights (main.py:241-248).
Data only: both moment vectors in parameters() order, the per-tensor step counters and the param-group scalars.
Run in the build container (needs /root/reference):  python tests/golden/make_optimizer_golden.py"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ck = torch.load('/root/reference/loot/gop_32_62/model.pth', map_location='cpu', weights_only=False)
osd = ck['optimizer_state_dict']
n = len(osd['state'])
group = osd['param_groups'][0]
assert group['params'] == list(range(n)) and len(osd['param_groups']) == 1
np.savez_compressed(os.path.join(HERE, 'loot_optimizer_state.npz'),
                    exp_avg=torch.cat([osd['state'][i]['exp_avg'].reshape(-1) for i in range(n)]).numpy(),
                    exp_avg_sq=torch.cat([osd['state'][i]['exp_avg_sq'].reshape(-1) for i in range(n)]).numpy(),
                    step=np.array([float(osd['state'][i]['step']) for i in range(n)]),
                    sizes=np.array([osd['state'][i]['exp_avg'].numel() for i in range(n)]),
                    lr=group['lr'], initial_lr=group['initial_lr'], beta1=group['betas'][0], beta2=group['betas'][1], eps=group['eps'],
                    weight_decay=group['weight_decay'], amsgrad=group['amsgrad'], epoch=ck['epoch'], loss=ck['loss'])
print('loot_optimizer_state.npz', n, 'tensors, lr', group['lr'], 'step', float(osd['state'][0]['step']))
