"""Device-pointer C-ABI entry points (torch tensors as device memory; torch is imported FIRST so
both libraries share one HIP runtime).  Spawned by tests/test_gpu_device_api.py."""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401  (must precede pvw_rs_amd in this process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import pvw_model as M  # noqa: E402
import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi  # noqa: E402


def ptr(t):
    return C.c_void_p(t.data_ptr())


def main():
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    lib = _ffi.lib()
    seed = bytes([0x2A]) * 32
    n, k, l, L = 37, 12, 8, 4
    moduli = M.bench_moduli(L)
    p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(moduli).build()
    gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, seed))
    gpk.fill_uniform(seed)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # --- pvw_encrypt_device: 7 back-to-back asynchronous calls, distinct and shared output buffers ---
    calls = 7
    seeds = [P.api._dealer_seed(seed, i) for i in range(calls)]
    scal = [torch.tensor([(i * 77 + j) % (1 << 32) for j in range(n)], dtype=torch.int64, device=dev) for i in range(calls)]
    outs = [(torch.zeros((k, L, l), dtype=torch.int64, device=dev), torch.zeros((n, L, l), dtype=torch.int64, device=dev))
            for _ in range(calls)]
    for i in range(calls):
        rnd = _ffi.pvw_randomness_t()
        rnd.mode = _ffi.RND_SEED
        C.memmove(rnd.seed, seeds[i], 32)
        P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal[i]), n, C.byref(rnd), ptr(outs[i][0]), ptr(outs[i][1]),
                                            P.REPR_NTT if i % 2 == 0 else P.REPR_POWER, stream))
    torch.cuda.synchronize()
    for i in range(calls):
        want = P.encrypt([int(x) for x in scal[i].cpu().numpy()], gpk, seeds[i], repr=P.REPR_NTT if i % 2 == 0 else P.REPR_POWER)
        assert np.array_equal(outs[i][0].cpu().numpy().view(np.uint64), want.c1), f"c1 call {i}"
        assert np.array_equal(outs[i][1].cpu().numpy().view(np.uint64), want.c2), f"c2 call {i}"
    # same output buffers reused every call: the last call must win, bit-exact
    c1, c2 = outs[0]
    for i in range(calls):
        rnd = _ffi.pvw_randomness_t()
        rnd.mode = _ffi.RND_SEED
        C.memmove(rnd.seed, seeds[i], 32)
        P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal[i]), n, C.byref(rnd), ptr(c1), ptr(c2), P.REPR_NTT, stream))
    torch.cuda.synchronize()
    want = P.encrypt([int(x) for x in scal[-1].cpu().numpy()], gpk, seeds[-1])
    assert np.array_equal(c1.cpu().numpy().view(np.uint64), want.c1) and np.array_equal(c2.cpu().numpy().view(np.uint64), want.c2)
    # explicit randomness through device pointers
    r = torch.from_numpy(p.sample_vec_cbd(seed, P.DOM_R, 0, k)).to(dev)
    e1 = torch.from_numpy(p.sample_uniform_coefficients(seed, P.DOM_E1, 0, k, 100)).to(dev)
    e2 = torch.from_numpy(p.sample_uniform_coefficients(seed, P.DOM_E2, 0, n, 200)).to(dev)
    rnd = _ffi.pvw_randomness_t()
    rnd.mode = _ffi.RND_EXPLICIT
    rnd.r, rnd.e1, rnd.e2 = r.data_ptr(), e1.data_ptr(), e2.data_ptr()
    P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal[0]), n, C.byref(rnd), ptr(c1), ptr(c2), P.REPR_NTT, stream))
    torch.cuda.synchronize()
    want = P.encrypt([int(x) for x in scal[0].cpu().numpy()], gpk, seed)
    assert np.array_equal(c1.cpu().numpy().view(np.uint64), want.c1) and np.array_equal(c2.cpu().numpy().view(np.uint64), want.c2)

    # --- pvw_encrypt_multi_device ---
    D = 6
    sc_m = torch.stack(scal[:D]).contiguous()
    c1m = torch.zeros((D, k, L, l), dtype=torch.int64, device=dev)
    c2m = torch.zeros((D, n, L, l), dtype=torch.int64, device=dev)
    sd = np.concatenate([np.frombuffer(s, dtype=np.uint8) for s in seeds[:D]]).copy()
    P.api._check(lib.pvw_encrypt_multi_device(p._h, ptr(sc_m), D, n, sd.ctypes.data_as(C.c_void_p), ptr(c1m), ptr(c2m),
                                              P.REPR_NTT, stream))
    torch.cuda.synchronize()
    for i in range(D):
        want = P.encrypt([int(x) for x in scal[i].cpu().numpy()], gpk, seeds[i])
        assert np.array_equal(c1m[i].cpu().numpy().view(np.uint64), want.c1)
        assert np.array_equal(c2m[i].cpu().numpy().view(np.uint64), want.c2)

    # --- pvw_decrypt_noisy_device + pvw_decode_device vs the host-buffer path ---
    sk = p.sample_vec_cbd(seed, P.DOM_SK, 0, k)
    d_sk = torch.from_numpy(sk).to(dev)
    d_c1s = c1m.clone()
    d_c2col = c2m[:, 3].contiguous()
    d_noisy = torch.zeros((D, L, l), dtype=torch.int64, device=dev)
    d_vals = torch.zeros(D, dtype=torch.int64, device=dev)
    P.api._check(lib.pvw_decrypt_noisy_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy), stream))
    P.api._check(lib.pvw_decode_device(p._h, ptr(d_noisy), D, ptr(d_vals), stream))
    torch.cuda.synchronize()
    cts = [P.PvwCiphertext(c1m[i].cpu().numpy().view(np.uint64), c2m[i].cpu().numpy().view(np.uint64), p, P.REPR_NTT) for i in range(D)]
    vals, noisy = P.api._decrypt_batch(p, cts, P.SecretKey.from_coefficients(p, sk), 3, return_noisy=True)
    assert np.array_equal(d_noisy.cpu().numpy().view(np.uint64), noisy)
    assert [int(v) for v in d_vals.cpu().numpy().view(np.uint64)] == vals
    # the one-call form (single chunk here)
    d_noisy_b = torch.zeros_like(d_noisy)
    d_vals_b = torch.zeros_like(d_vals)
    P.api._check(lib.pvw_decrypt_batch_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy_b), ptr(d_vals_b), stream))
    torch.cuda.synchronize()
    assert torch.equal(d_noisy_b, d_noisy) and torch.equal(d_vals_b, d_vals)
    # ... and left nothing of NTT(sk) behind (a single-pass call has its decode launch clear it; secret_key.rs:20-30)
    nz, scanned = P.api._secret_residue(p)
    assert nz == 0 and scanned > 0, (nz, scanned)
    # ... and with the key resident on the device (pvw_sk_load): the same values, nothing transformed or wiped per call
    with P.SecretKey.from_coefficients(p, sk).load_device() as dkey:
        for _ in range(2):
            d_noisy_c = torch.zeros_like(d_noisy)
            d_vals_c = torch.zeros_like(d_vals)
            P.api._check(lib.pvw_decrypt_batch_device_sk(p._h, dkey._h, ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy_c), ptr(d_vals_c), stream))
            torch.cuda.synchronize()
            assert torch.equal(d_noisy_c, d_noisy) and torch.equal(d_vals_c, d_vals)
    graph_capture(lib, dev)
    config5_full_size(lib, stream, dev)
    print("DEVICE_API_OK")


def graph_capture(lib, dev):
    """pvw_prepare, then pvw_encrypt_device captured into a graph and replayed: a captured call must not allocate or
    synchronise, and replays with new scalars in the same buffer give that many different, correct ciphertexts."""
    seed = bytes([0x51]) * 32
    n, k, l, L = 70, 256, 8, 3
    moduli = M.bench_moduli(L)
    p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(moduli).build()
    gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, seed))
    gpk.fill_uniform(seed)
    s = torch.cuda.Stream(device=dev)
    assert p.prepare(P.PREPARE_PACKED, s.cuda_stream) > 0 and p.packed_active() == 61
    scal = torch.zeros(n, dtype=torch.int64, device=dev)
    c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
    c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
    rnd = _ffi.pvw_randomness_t()
    rnd.mode = _ffi.RND_SEED
    C.memmove(rnd.seed, seed, 32)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal), n, C.byref(rnd), ptr(c1), ptr(c2), P.REPR_NTT,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for rep in range(3):
        vals = [(rep * 1000003 + 17 * j) % (1 << 32) for j in range(n)]
        scal.copy_(torch.tensor(vals, dtype=torch.int64))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = P.encrypt(vals, gpk, seed)
        assert np.array_equal(c1.cpu().numpy().view(np.uint64), want.c1), f"graph replay {rep}: c1"
        assert np.array_equal(c2.cpu().numpy().view(np.uint64), want.c2), f"graph replay {rep}: c2"
    # ... and the eager path on the same stream afterwards
    P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal), n, C.byref(rnd), ptr(c1), ptr(c2), P.REPR_NTT, C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(c2.cpu().numpy().view(np.uint64), want.c2)


def config5_full_size(lib, stream, dev):
    """BASELINE.json configs[4] at full size on ONE GPU: D = 8192 dealer ciphertexts, k = 512, l = 16, 34 limbs:
    18.25 GB of stacked c1, synthetic residues generated on the device.  Sampled dealers' noisy polynomials
    against the C restatement, every decoded value against the model's decode of the device's noisy polynomial
    (decode is checked exhaustively elsewhere), and linearity: decrypt(c2 + delta) - decrypt(c2) = -delta."""
    import pvw_model as M
    import pvw_oracle as O
    D, k, l, L = 8192, 512, 16, 34
    moduli = M.bench_moduli(L)
    p = (P.PvwParametersBuilder().set_parties(4).set_dimension(k).set_l(l).set_moduli(moduli)
         .set_secret_variance(0.5).set_error_bounds(100, 200).build())
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    qmin = min(moduli)
    d_c1s = torch.randint(0, qmin, (D, k, L, l), dtype=torch.int64, device=dev, generator=g)     # residues < every q_i
    d_c2col = torch.randint(0, qmin, (D, L, l), dtype=torch.int64, device=dev, generator=g)
    sk = p.sample_vec_cbd(bytes([9]) * 32, P.DOM_SK, 0, k)
    d_sk = torch.from_numpy(sk).to(dev)
    d_noisy = torch.zeros((D, L, l), dtype=torch.int64, device=dev)
    d_vals = torch.zeros(D, dtype=torch.int64, device=dev)
    P.api._check(lib.pvw_decrypt_noisy_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy), stream))
    P.api._check(lib.pvw_decode_device(p._h, ptr(d_noisy), D, ptr(d_vals), stream))
    torch.cuda.synchronize()
    orc = O.Oracle(moduli, l)
    pick = [0, 1, 4095, 4096, 8190, 8191]
    c1_h = np.stack([d_c1s[d].cpu().numpy().view(np.uint64) for d in pick])
    c2_h = np.stack([d_c2col[d].cpu().numpy().view(np.uint64) for d in pick])
    want = orc.decrypt_noisy(sk, c1_h, c2_h)
    got = np.stack([d_noisy[d].cpu().numpy().view(np.uint64) for d in pick])
    assert np.array_equal(got, want), "config-5 full size: noisy polynomials"
    mp = M.Params(4, k, l, moduli)
    vals = d_vals.cpu().numpy().view(np.uint64)
    for d, z in zip(pick, got):
        lifted = M.from_rns([[int(v) for v in row] for row in z], list(moduli))
        assert int(vals[d]) == M.decode_scalar_pvw(lifted, mp), "config-5 full size: decode"
    # pvw_decrypt_batch_device: eight chunks of 1024 dealers, each decode on the helper stream under the next MAC
    d_noisy_b = torch.zeros_like(d_noisy)
    d_vals_b = torch.zeros_like(d_vals)
    for _ in range(3):
        P.api._check(lib.pvw_decrypt_batch_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy_b), ptr(d_vals_b), stream))
    torch.cuda.synchronize()
    assert torch.equal(d_noisy_b, d_noisy) and torch.equal(d_vals_b, d_vals), "config-5 full size: overlapped batch decrypt"
    # linearity over ALL dealers: delta added to every NTT slot of c2 is the constant polynomial delta, so the
    # noisy polynomial (power basis) loses delta in coefficient 0 and nothing elsewhere, limb-wise
    delta = 12345
    d_c2b = d_c2col + delta
    for i, q in enumerate(moduli):
        d_c2b[:, i] %= q
    d_noisy2 = torch.zeros_like(d_noisy)
    P.api._check(lib.pvw_decrypt_noisy_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2b), D, P.REPR_NTT, ptr(d_noisy2), stream))
    d_noisy1 = torch.zeros_like(d_noisy)
    P.api._check(lib.pvw_decrypt_noisy_device(p._h, ptr(d_sk), ptr(d_c1s), ptr(d_c2col), D, P.REPR_NTT, ptr(d_noisy1), stream))
    torch.cuda.synchronize()
    for i, q in enumerate(moduli):
        diff = (d_noisy1[:, i] - d_noisy2[:, i]) % q
        assert bool((diff[:, 0] == delta).all()) and bool((diff[:, 1:] == 0).all()), "config-5 full size: linearity"


if __name__ == "__main__":
    main()
