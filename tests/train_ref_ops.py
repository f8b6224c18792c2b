"""TEST INFRASTRUCTURE: plain-torch twins of ppsurf_amd.train_ops (gradients by torch autograd).

The product ops are HIP-only.  The CPU suite patches these in so that the surrounding training graph
(ppsurf_amd/train_graph.py) can be checked against the reference's train-mode fixtures without a GPU; the GPU suite
compares the HIP ops (forward and backward) with these twins."""
import contextlib

import torch


def gather_rows(x, idx):
    return x[idx]


def neighbour_max(x, idx):
    return x[idx.reshape(-1)].view(idx.shape[0], idx.shape[1], -1).max(dim=1)[0]


def neighbour_contract(x, idx, g):
    xg = x[idx.reshape(-1)].view(idx.shape[0], idx.shape[1], -1)             # [m,k,c]
    return torch.einsum('mkc,mkt->mct', xg, g).reshape(idx.shape[0], -1)


@contextlib.contextmanager
def patched():
    from ppsurf_amd import train_ops
    saved = (train_ops.gather_rows, train_ops.neighbour_max, train_ops.neighbour_contract)
    train_ops.gather_rows, train_ops.neighbour_max, train_ops.neighbour_contract = gather_rows, neighbour_max, neighbour_contract
    try:
        yield
    finally:
        train_ops.gather_rows, train_ops.neighbour_max, train_ops.neighbour_contract = saved
