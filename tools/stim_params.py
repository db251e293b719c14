"""Carrier offset, phase and gain of the channels of synth_channels_torch (the draws of its generator, replayed):
python tools/stim_params.py <channels> <samples> <seed> <channel> [<channel> ...]   (GPU box: the generator is the device's)"""
import math
import sys

import torch

C, N, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3], 0)
M, S = 4, 8
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(seed)
n_sym = -(-N // S)
nc = min(C, 256)
k = torch.randint(0, M, (nc, n_sym), generator=gen, device=dev)
phi0 = torch.rand((nc, 1), generator=gen, device=dev) * (2 * math.pi / M)
gain = 0.5 + 1.5 * torch.rand((nc, 1), generator=gen, device=dev)
dphi = (2 * torch.rand((nc, 1), generator=gen, device=dev) - 1) * (1e-3 / M)
q = 2 * math.pi / (M * n_sym)
dq = torch.round(dphi / q)
for c in [int(v) for v in sys.argv[4:]]:
    print("channel %d: offset %+d quanta (%.3e rad per symbol, M-th power phase %.3e per symbol), phi0 %.4f, M phi0 %.4f, gain %.3f" % (
        c, int(dq[c, 0]), float(dq[c, 0]) * q, M * float(dq[c, 0]) * q, float(phi0[c, 0]), M * float(phi0[c, 0]), float(gain[c, 0])))
print("offsets of all channels, in quanta:", [int(v) for v in dq[:, 0].tolist()])
