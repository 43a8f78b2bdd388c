"""Worker for the world_size-2 tests (spawned by tests/test_dist_gloo.py).

mode "oracle": CPU only -- exercises the shard plan / CRS broadcast / gather plumbing of
pvw_rs_amd.dist over gloo, with the C restatement standing in for the per-rank compute.
mode "hip": both ranks drive the HIP path on cuda:0 (gloo for the broadcast, because two ranks
cannot share one device under RCCL); rank 0 checks the union against an unsharded context.
mode "nccl": ONE rank, backend "nccl" (= RCCL) on cuda:0 -- the collectives of pvw_rs_amd.dist and of bench.py
(broadcast of the CRS as a device tensor, all_gather of the decoded shares, barrier, all_reduce(MAX)) run through
RCCL on device tensors; more ranks need more GPUs than this pipeline's box has."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    mode, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dev = None
    if mode == "nccl":
        import torch
        assert world == 1 and dist.is_nccl_available()
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import pvw_model as M
    import pvw_oracle as O
    from pvw_rs_amd import dist as D
    import pvw_rs_amd as P

    seed = bytes([0x2A]) * 32
    n, k, l, L = 22, 10, 8, 3
    moduli = M.bench_moduli(L)
    lo, hi, clo, chi = D.shard_ranges(n, k, world, rank)
    orc = O.Oracle(moduli, l)
    scalars = np.array([(i * 1000 + 1) % (1 << 32) for i in range(n)], dtype=np.uint64)
    r = O.sample_cbd(seed, M.DOM_R, 0, k, l, 0.5)
    e1 = O.sample_uniform(seed, M.DOM_E1, 0, k, l, 100)
    e2 = O.sample_uniform(seed, M.DOM_E2, 0, n, l, 200)
    a_src = orc.fill_uniform(seed, M.DOM_CRS, 0, k * k).reshape(k, k, L, l) if rank == 0 else None
    a_t = D.broadcast_crs(a_src, (k, k, L, l), src=0, device=dev)     # once, at load time
    a_hat = a_t.cpu().numpy().view(np.uint64)
    b_full = orc.fill_uniform(seed, M.DOM_PK, 0, n * k).reshape(n, k, L, l)
    if mode == "oracle":
        g_hat = orc.ntt_forward(np.array([[pow(M.Params(n, k, l, moduli).delta, j, q) for j in range(l)]
                                           for q in moduli], dtype=np.uint64)[None])[0]
        # per-rank compute on the shard only (rows of A for c1, rows of B for c2)
        c1_loc = np.zeros((k, L, l), dtype=np.uint64)
        c2_loc = np.zeros((n, L, l), dtype=np.uint64)
        c1_full, c2_part = orc.encrypt(a_hat, b_full[lo:hi], g_hat, scalars[lo:hi], r, e1, e2[lo:hi])
        c1_loc[clo:chi] = c1_full[clo:chi]
        c2_loc[lo:hi] = c2_part
    else:
        p = D.sharded_builder(n, k, l, moduli, world, rank, device=0).build()
        crs = D.load_broadcast_crs(p, a_t)
        gpk = P.GlobalPublicKey.new(crs)
        gpk.fill_uniform(seed)
        assert gpk.is_full()
        ct = P.encrypt(scalars, gpk, seed)                              # same seed on every rank
        c1_loc, c2_loc = ct.c1, ct.c2
        g_hat = p.gadget_polynomial(P.REPR_NTT)
    # config-5 shape: dealers sharded over the ranks, D x u64 results all-gathered
    Dn = 9
    dlo, dhi = D.shard_dealers(Dn, world, rank)
    local_vals = np.arange(dlo, dhi, dtype=np.uint64) * np.uint64(3) + np.uint64(1)
    got_all = D.all_gather_decrypted(local_vals, Dn, device=dev)
    assert got_all.tolist() == [3 * d + 1 for d in range(Dn)], got_all
    if mode == "nccl":
        # the rest of bench.py's collectives on device tensors: barrier, max-reduce of the elapsed time
        import torch
        dist.barrier()
        t = torch.tensor([1.25 + rank], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25 + (world - 1)
    c1 = D.gather_rows(c1_loc, clo, chi, k)
    c2 = D.gather_rows(c2_loc, lo, hi, n)
    if rank == 0:
        c1o, c2o = orc.encrypt(a_hat, b_full, g_hat, scalars, r, e1, e2)
        assert np.array_equal(c1, c1o), "c1 union mismatch"
        assert np.array_equal(c2, c2o), "c2 union mismatch"
        print("DIST_OK", mode)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
