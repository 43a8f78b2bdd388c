"""Multi-GPU plumbing for the party-sharded encrypt (one process per GPU, torch.distributed).

The path shards by party index: rank g owns B-hat rows [party_lo, party_hi) and
computes c1 rows [c1_lo, c1_hi); A-hat is broadcast ONCE at load time (RCCL over xGMI with
backend "nccl", gloo in the CPU tests); r is derived on every rank from the same 32-byte seed,
so there is no collective on the data path.  torch.distributed is plumbing here, not the product.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .api import GlobalPublicKey, PvwCrs, PvwParameters, PvwParametersBuilder, _check


def shard_ranges(n: int, k: int, world: int, rank: int) -> Tuple[int, int, int, int]:
    """Contiguous, balanced blocks: parties [party_lo, party_hi), c1 rows [c1_lo, c1_hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n * rank // world, n * (rank + 1) // world, k * rank // world, k * (rank + 1) // world)


def sharded_builder(n: int, k: int, l: int, moduli, world: int, rank: int, device: int = -1) -> PvwParametersBuilder:
    lo, hi, clo, chi = shard_ranges(n, k, world, rank)
    return (PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(moduli)
            .set_device(device).set_shard(lo, hi, clo, chi))


def broadcast_crs(a_hat: Optional[np.ndarray], shape, src: int = 0, device=None):
    """Broadcast the NTT-domain CRS [k][k][L][l] from `src` to every rank; returns a torch tensor
    (int64 view of the u64 residues) on `device` (None = CPU).  Called once per key set."""
    import torch
    import torch.distributed as dist
    t = torch.empty(tuple(shape), dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        if a_hat is None:
            raise ValueError("source rank must supply the CRS")
        t.copy_(torch.from_numpy(np.ascontiguousarray(a_hat, dtype=np.uint64).view(np.int64)))
    dist.broadcast(t, src=src)
    return t


def load_broadcast_crs(params: PvwParameters, t, repr: int = _ffi.REPR_NTT) -> PvwCrs:
    """Hand a broadcast CRS tensor (device or host) to the context; it keeps its own c1 rows."""
    import torch
    if t.is_cuda:
        stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        params._call("pvw_load_crs_device", C.c_void_p(t.data_ptr()), repr, stream)
        torch.cuda.synchronize(t.device)
    else:
        a = t.numpy().view(np.uint64)
        params._call("pvw_load_crs", a.ctypes.data_as(C.c_void_p), repr)
    return PvwCrs(params)


def shard_dealers(num_dealers: int, world: int, rank: int) -> Tuple[int, int]:
    """Dealer ciphertexts [lo, hi) decrypted by `rank` (batched decrypt_party_shares, config 5)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return num_dealers * rank // world, num_dealers * (rank + 1) // world


def all_gather_decrypted(local_values, num_dealers: int, device=None) -> np.ndarray:
    """The one collective of the sharded decrypt: every rank contributes the u64 results of its
    dealer shard (decoded on its GPU) and receives all `num_dealers` of them.  8 bytes per
    ciphertext -- an RCCL all_gather over xGMI with backend "nccl", gloo in the CPU tests."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    per = max(shard_dealers(num_dealers, world, r)[1] - shard_dealers(num_dealers, world, r)[0] for r in range(world))
    mine = torch.zeros(per, dtype=torch.int64, device=device)
    vals = np.asarray(local_values, dtype=np.uint64).view(np.int64)
    mine[: len(vals)] = torch.from_numpy(vals.copy()).to(mine.device)
    parts = [torch.zeros(per, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.zeros(num_dealers, dtype=np.uint64)
    for r in range(world):
        lo, hi = shard_dealers(num_dealers, world, r)
        out[lo:hi] = parts[r][: hi - lo].cpu().numpy().view(np.uint64)
    return out


def gather_rows(local: np.ndarray, lo: int, hi: int, total_rows: int, dst: int = 0) -> Optional[np.ndarray]:
    """Test/diagnostic helper: gather row shards [lo, hi) of every rank on `dst` (gloo or nccl)."""
    import torch.distributed as dist
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((lo, hi, np.ascontiguousarray(local[lo:hi])), parts, dst=dst)
    if parts is None:
        return None
    out = np.zeros((total_rows,) + local.shape[1:], dtype=local.dtype)
    for plo, phi, rows in parts:
        out[plo:phi] = rows
    return out
