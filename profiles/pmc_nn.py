#!/usr/bin/env python3
"""PMC target for the NN search alone (uniform_ fills: the hash generators are serialised under --pmc): one warm-up + one timed call of
find_nn_indices per loss configuration at 720p, 52 / 75 frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from videoloop3d_amd.utils_vid import find_nn_indices
dev = torch.device("cuda:0")
x = torch.empty((1, 3, 52, 719, 1279), device=dev).uniform_()
y = torch.empty((1, 3, 75, 719, 1279), device=dev).uniform_()
for ps, s, al in ((11, 4, 0.5), (3, 2, None)):
    for _ in range(2):
        find_nn_indices(x, y, ps, 3, s, 1, al)
torch.cuda.synchronize()
