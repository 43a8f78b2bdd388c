#!/usr/bin/env python3
"""bench.py -- parties/s of one n-party PVW `encrypt` on MI355X, with the roofline of the
dominant kernel (mac_rows) and the CPU restatement timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c1|c4shard]

A "step" is one full encrypt (sample r/e1/e2 from a seed, NTT, c1 = A r + e1,
c2 = B r + e2 + m g; src/crypto/encryption.rs:105-214) over synthetic A-hat / B-hat that are
already resident in HBM.  N > 1: one rank per GPU -- either under torch.distributed.run, or started
as `python bench.py --gpus N`, which launches torch.distributed.run itself as a child process before
anything touches the GPU.  The parties are sharded over the ranks (weak scaling: every rank holds the
per-GPU party count; `--config c4shard` / `--path decrypt --config c5shard` are BASELINE configs[3] /
[4] as their 8-GPU shards), A-hat is broadcast once over RCCL at load time, and there is no collective
on the data path.  Rank 0 prints ONE JSON line.  The workload definition (geometries, modulus chain,
seeds) is pvw_rs_amd/workloads.py; oracle/ is touched by the cpu_baseline leg only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# Kernel arguments in device memory (a documented ROCm runtime setting, read when the HIP runtime initialises, so it
# must be in the environment before torch is imported; inherited by the ranks of a self-launch).  The encrypt
# prologue's batch descriptor travels as kernel arguments: host-resident, every workgroup's first read of it crosses
# PCIe -- 17.4 -> 14.5 us per launch on a box whose default is host memory (deployment notes).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from pvw_rs_amd import workloads as W          # noqa: E402  (pure Python: geometries, modulus chain, seeds)

CONFIGS = W.ENCRYPT_CONFIGS
DECRYPT_CONFIGS = W.DECRYPT_CONFIGS
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DIST_ON = False                # a process group exists (N > 1, or PVW_BENCH_FORCE_DIST=1 at N = 1)
SEED_A, SEED_B, SEED_ENC = W.SEED_A, W.SEED_B, W.SEED_ENC


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start torch.distributed.run as a CHILD process
    (one rank per GPU, 127.0.0.1 rendezvous) before anything here has touched the GPU, and exit with its
    code.  Under a launcher (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def measured_traffic(config, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/r*_<config>_summary.json, written by tools/summarize_prof.py), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_summary.json"))):
        try:
            for name, kv in json.load(open(f))["kernels"].items():
                if name.startswith(kernel_prefix) and "hbm_traffic_bytes_per_launch" in kv:
                    best = (kv["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS) + sorted(DECRYPT_CONFIGS))
    ap.add_argument("--dealers", type=int, default=0,
                    help="encrypt path only: > 0 batches this many dealers per step through pvw_encrypt_multi "
                         "(encrypt_all_party_shares); value is then party-ciphertexts/s")
    ap.add_argument("--resident-key", action="store_true",
                    help="decrypt path: the secret key stays on the device in NTT form (pvw_sk_load) instead of being handed "
                         "over, transformed and wiped on every call")
    ap.add_argument("--no-worst-case", dest="worst_case", action="store_false",
                    help="decrypt path: skip the second measurement on uniform residues")
    ap.add_argument("--path", default="encrypt", choices=["encrypt", "decrypt", "keygen"],
                    help="encrypt = the headline metric; decrypt = batched decrypt_party_value (BASELINE configs[4] shape)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-probe", action="store_true", help="skip the read-only bandwidth probe (tuning build)")
    ap.add_argument("--tuning-library", action="store_true",
                    help="run on libpvw_hip_tuning.so (the measurement build: honours the PVW_* schedule selectors of DESIGN 7a)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="length of the back-to-back leg reported as `sustained` (0 = skip)")
    args = ap.parse_args()
    self_launch(args)

    import numpy as np
    import torch

    import pvw_rs_amd as P
    from pvw_rs_amd import _ffi
    if args.tuning_library:
        _ffi.select("tuning")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available() or not P.device_available():
        print("bench.py needs a gfx950 GPU: the PVW hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # PVW_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, CRS broadcast, barrier, max-reduce of the time,
    # all-gather of the decrypt) with world_size 1 -- RCCL exercised on a one-GPU box
    global DIST_ON
    DIST_ON = world > 1 or os.environ.get("PVW_BENCH_FORCE_DIST") == "1"
    if DIST_ON:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            sock.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL ("nccl") over xGMI in production; PVW_BENCH_BACKEND=gloo only to rehearse the N>1 code path
        # on a box with fewer GPUs than ranks (see PVW_BENCH_SAME_DEVICE below)
        dist.init_process_group(backend=os.environ.get("PVW_BENCH_BACKEND", "nccl"))
    if os.environ.get("PVW_BENCH_SAME_DEVICE") == "1":
        local_rank = 0                     # rehearsal: every rank on cuda:0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.path == "decrypt":
        return bench_decrypt(args, world, rank, local_rank, dev)
    if args.path == "keygen":
        return bench_keygen(args, world, rank, local_rank, dev)
    n_per, k, l, L, desc = CONFIGS[args.config or "c3"]
    n_total = n_per * world
    moduli = W.config_moduli(args.config or "c3", L)
    from pvw_rs_amd import dist as D
    lo, hi, clo, chi = D.shard_ranges(n_total, k, world, rank)
    params = D.sharded_builder(n_total, k, l, moduli, world, rank, device=local_rank) \
        .set_secret_variance(W.SECRET_VARIANCE).set_error_bounds_u32(W.ERROR_BOUND_1, W.ERROR_BOUND_2).build()
    h = params._h
    lib = _ffi.lib()

    # ---- residency: A-hat (generated on rank 0, broadcast ONCE over RCCL/xGMI), B-hat shard ----
    if DIST_ON:
        a_host = None
        if rank == 0:
            p0 = (P.PvwParametersBuilder().set_parties(n_total).set_dimension(k).set_l(l).set_moduli(moduli)
                  .set_device(local_rank).build())
            a_host = P.PvwCrs.new_deterministic(p0, SEED_A).matrix(P.REPR_NTT)
            del p0
        a_dev = D.broadcast_crs(a_host, (k, k, L, l), src=0, device=dev)
        crs = D.load_broadcast_crs(params, a_dev)
        del a_dev, a_host
    else:
        crs = P.PvwCrs.new_deterministic(params, SEED_A)
    gpk = P.GlobalPublicKey.new(crs)
    gpk.fill_uniform(SEED_B)

    scalars = torch.tensor(W.scalars(n_total), dtype=torch.int64, device=dev)
    c1 = torch.zeros((chi - clo, L, l), dtype=torch.int64, device=dev)
    c2 = torch.zeros((n_per, L, l), dtype=torch.int64, device=dev)
    rnd = _ffi.pvw_randomness_t()
    rnd.mode = _ffi.RND_SEED
    C.memmove(rnd.seed, SEED_ENC, 32)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    Dm = args.dealers
    if Dm > 0:
        scalars_m = scalars.repeat(Dm, 1).contiguous()
        c1m = torch.zeros((Dm, chi - clo, L, l), dtype=torch.int64, device=dev)
        c2m = torch.zeros((Dm, n_per, L, l), dtype=torch.int64, device=dev)
        seeds_m = np.concatenate([np.frombuffer(P.api._dealer_seed(SEED_ENC, d), dtype=np.uint8) for d in range(Dm)]).copy()

    def step():
        if Dm > 0:
            rc = lib.pvw_encrypt_multi_device(h, C.c_void_p(scalars_m.data_ptr()), Dm, n_total,
                                              seeds_m.ctypes.data_as(C.c_void_p), C.c_void_p(c1m.data_ptr()),
                                              C.c_void_p(c2m.data_ptr()), P.REPR_NTT, stream)
        else:
            rc = lib.pvw_encrypt_device(h, C.c_void_p(scalars.data_ptr()), n_total, C.byref(rnd),
                                        C.c_void_p(c1.data_ptr()), C.c_void_p(c2.data_ptr()), P.REPR_NTT, stream)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())

    def barrier():
        torch.cuda.synchronize()
        if DIST_ON:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if DIST_ON:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * max(Dm, 1) * args.steps / elapsed

    # ---- roofline of the dominant kernel: HIP events around every mac_rows launch --------------
    params.set_profiling(True)
    params.reset_profiling()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    kt = {name: params.kernel_time(name) for name in ("mac_rows", "mac_rows_multi", "gemm_digits", "vec_digits", "prologue")}
    params.set_profiling(False)
    gemm_path = Dm > 0 and kt["gemm_digits"][1] > 0
    mac_ms, mac_launches = kt["gemm_digits"] if gemm_path else (kt["mac_rows_multi"] if Dm > 0 else kt["mac_rows"])
    mac_avg_s = mac_ms / max(mac_launches, 1) * 1e-3
    rows_a = chi - clo
    # algorithmic bytes of one mac_rows launch: B-hat + A-hat reads, c2 + c1 writes, r-hat read
    # vectors per launch of the dominant kernel (the matrix-core path puts up to 64 dealers into one launch)
    per_step = max(mac_launches // max(args.steps, 1), 1)
    nv = max(Dm // per_step, 1) if Dm > 0 else 1
    alg_bytes = 8 * L * l * (n_per * k + rows_a * k + nv * (n_per + rows_a + k))
    achieved = alg_bytes / mac_avg_s / 1e9 if mac_avg_s > 0 else 0.0

    # single-dealer launches stream the matrices from a bit-packed copy when the geometry qualifies (40 / 48 / 56 / 61 bits
    # per residue by the widest modulus; the library says which one it used): the algorithmic bytes stay SURVEY 8d's 8
    # bytes per residue, the bytes the kernel actually has to read are reported next to them
    width = params.packed_active() if Dm == 0 else 0
    if args.tuning_library and (os.environ.get("PVW_MAC_PACKED") == "0" or os.environ.get("PVW_MAC_VARIANT", "0") not in ("0", "44")):
        width = 0
    packed = width != 0
    mac_kernel = ("mac_rows_packed61_kernel" if width == 61 else "mac_rows_packedw_kernel") if packed else "mac_rows_kernel"
    streamed_bytes = (8 * L * l * ((n_per * k + rows_a * k) * width // 64 + nv * (n_per + rows_a + k))) if packed else alg_bytes
    tr = measured_traffic(args.config or "c3", mac_kernel)
    out = {
        "metric": "parties/s for n-party encrypt (pvw::crypto::encrypt); achieved HBM GB/s vs peak",
        "value": value, "unit": "parties/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": desc, "parties_per_gpu": n_per, "parties_total": n_total, "k": k, "l": l,
                   "rns_limbs": L, "q_bits": int(params.q_total().bit_length()), "randomness": "seed (ChaCha8), on device", "dealers_per_step": max(Dm, 1),
                   "sharding": f"party-sharded x{world}, A-hat broadcast once, no data-path collective",
                   "world_size_observed": world, "backend": (os.environ.get("PVW_BENCH_BACKEND", "nccl") if DIST_ON else None)},
        "roofline": {"bound": "hbm", "kernel": mac_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": tr[0] if tr else None,
                     "traffic_source": (tr[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled)") if tr else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "streamed_bytes_per_launch": streamed_bytes,
                     "frac_of_streamed_bytes": (streamed_bytes / mac_avg_s / 1e9 / HBM_PEAK_GBS) if mac_avg_s > 0 else 0.0,
                     "packing": (f"matrix residues stored at {width} of 64 bits ({mac_kernel}): `achieved` / `frac` count the algorithmic "
                                 "8 bytes per residue, `frac_of_streamed_bytes` the bytes the kernel reads") if packed else None,
                     "avg_launch_us": mac_avg_s * 1e6,
                     "launches_timed": mac_launches},
        "kernel_ms_per_step": {name: (v[0] / max(args.steps, 1)) for name, v in kt.items()},
    }
    # ---- sustained leg: >= --sustain-seconds of back-to-back steps (clocks and thermals settle; the driver's
    # gpu_busy sampler sees the GPU), then a profiled stretch of the same loop while the chip is still hot ----
    if args.sustain_seconds > 0:
        per_sync = 256
        barrier()
        t_s, n_s = time.perf_counter(), 0
        while True:
            for _ in range(per_sync):
                step()
            torch.cuda.synchronize()
            n_s += per_sync
            if time.perf_counter() - t_s >= args.sustain_seconds:
                break
        el_s = time.perf_counter() - t_s
        params.set_profiling(True)
        params.reset_profiling()
        for _ in range(per_sync):
            step()
        torch.cuda.synchronize()
        hot = params.kernel_time("gemm_digits" if gemm_path else ("mac_rows_multi" if Dm > 0 else "mac_rows"))
        params.set_profiling(False)
        hot_us = hot[0] / max(hot[1], 1) * 1e3
        out["sustained"] = {"seconds": el_s, "steps": n_s, "value": n_total * max(Dm, 1) * n_s / el_s, "unit": out["unit"],
                            "ms_per_step": el_s / n_s * 1e3, "dominant_kernel_avg_us_hot": hot_us,
                            "roofline_frac_hot": (alg_bytes / (hot_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if hot_us > 0 and not gemm_path else None,
                            "note": f"rank 0's own clock, synchronised every {per_sync} steps; not the contract's timed region"}
    if Dm == 0 and rank == 0 and world == 1 and not args.no_probe:
        # what the memory system gives this access pattern with the arithmetic removed (read-only probe over an
        # identical B-hat).  The probe lives in the measurement build only (include/pvw_hip_tuning.h), so it runs on
        # a second context of libpvw_hip_tuning.so holding its own copy of the public key.
        try:
            prev = _ffi.select("tuning")
            try:
                pt = (P.PvwParametersBuilder().set_parties(n_total).set_dimension(k).set_l(l).set_moduli(moduli)
                      .set_device(local_rank).build())
                P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(pt, SEED_A)).fill_uniform(SEED_B)
            finally:
                _ffi.select(prev)
            sec, nbytes = C.c_double(0.0), C.c_uint64(0)
            pt._call("pvw_selftest_read_bandwidth", 20, C.byref(sec), C.byref(nbytes))
            if sec.value > 0:
                gbps = nbytes.value / sec.value / 1e9
                out["roofline"]["read_probe"] = {"GBps": gbps, "bytes_per_pass": nbytes.value,
                                                 "mac_rows_over_probe": (streamed_bytes / mac_avg_s / 1e9) / gbps if mac_avg_s > 0 else None,
                                                 "note": "mac_rows' loads (1-KiB tiles, 16 B/lane, nt, 16 in flight per wave) over the UNPACKED tiled B-hat with an xor instead of the "
                                                         "modular MAC; mac_rows_over_probe compares the bytes mac_rows actually streams per second with it; libpvw_hip_tuning.so"}
            del pt
        except Exception as e:                 # the tuning build is optional equipment
            out["roofline"]["read_probe"] = {"error": str(e)[:200]}
    if Dm > 0:
        out["metric"] = ("party-ciphertexts/s for encrypt_all_party_shares (D dealers x n parties, "
                         + ("batches of 16, up to 128 per launch" if gemm_path else "4") + " dealers per pass over B-hat)")
        out["unit"] = "party-ciphertexts/s"
        if "sustained" in out:
            out["sustained"]["unit"] = out["unit"]
        out["roofline"]["kernel"] = "gemm_digits_kernel (i8 MFMA)" if gemm_path else "mac_rows_multi_kernel"
        mm = L * l * (n_per * k + rows_a * k) * nv / mac_avg_s if mac_avg_s > 0 else 0.0
        out["roofline"]["modular_macs_per_s"] = mm
        out["roofline"]["vectors_per_launch"] = nv
        if gemm_path:
            # the digit GEMM is priced against the matrix cores: 64 i8 products (128 ops) per modular MAC; the
            # HBM figures of the same launch are kept beside it
            out["roofline"].update({"bound": "mfma", "hbm_achieved_GBps": achieved, "achieved": mm * 128 / 1e12,
                                    "peak": 5000.0, "unit": "TOP/s (i8)", "frac": mm * 128 / 1e12 / 5000.0, "traffic": None,
                                    "traffic_source": None,
                                    "kernel": "gemm_digits_kernel (i8 MFMA, 64 byte-products per modular MAC)"})

    # ---- CPU baseline: the C restatement (oracle/) on this box's host cores, rank 0, N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu and Dm == 0:
        # the same encrypt through the HOST-buffer entry point (scalars in, c1/c2 out over PCIe):
        # reported for completeness, never as `value`.  Measured BEFORE the OpenMP baseline (whose idle
        # worker threads spin for a while and disturb a synchronous host call); median of 20 calls.
        sc_host = np.array([(i * 1000 + 1) % (1 << 32) for i in range(n_total)], dtype=np.uint64)
        c1h = np.zeros((k, L, l), dtype=np.uint64)
        c2h = np.zeros((n_total, L, l), dtype=np.uint64)
        t_calls = []
        for it in range(23):
            t_h = time.perf_counter()
            rc = lib.pvw_encrypt(h, sc_host.ctypes.data_as(C.c_void_p), n_total, C.byref(rnd),
                                 c1h.ctypes.data_as(C.c_void_p), c2h.ctypes.data_as(C.c_void_p), P.REPR_NTT)
            if rc != 0:
                raise RuntimeError(_ffi.last_error())
            if it >= 3:
                t_calls.append(time.perf_counter() - t_h)
        t_h = sorted(t_calls)[len(t_calls) // 2]
        out["host_buffer_path"] = {"ms_per_encrypt": t_h * 1e3, "ms_min": min(t_calls) * 1e3, "ms_max": max(t_calls) * 1e3,
                                   "parties_per_s": n_total / t_h,
                                   "note": "pvw_encrypt with pageable host buffers, synchronous, PCIe-inclusive (c1+c2 = "
                                           f"{(k + n_total) * L * l * 8 / 1e6:.1f} MB D2H per call); median of 20 calls",
                                   "bit_exact_vs_device_path": bool(np.array_equal(c2h.view(np.int64), c2.cpu().numpy()))}
        # ... and with the output buffers in pinned, device-visible host memory (pvw_host_alloc): the MAC stores the ciphertexts
        # straight into them while it runs -- no copy after the kernel
        try:
            ptrs = []
            def pinned(shape):
                nbytes = int(np.prod(shape)) * 8
                pp = C.c_void_p()
                if lib.pvw_host_alloc(nbytes, C.byref(pp)) != 0:
                    raise RuntimeError(_ffi.last_error())
                ptrs.append(pp)
                return np.ctypeslib.as_array((C.c_uint64 * (nbytes // 8)).from_address(pp.value)).reshape(shape)
            c1p, c2p = pinned((k, L, l)), pinned((n_total, L, l))
            t_pin = []
            for it in range(23):
                t_h2 = time.perf_counter()
                rc = lib.pvw_encrypt(h, sc_host.ctypes.data_as(C.c_void_p), n_total, C.byref(rnd),
                                     c1p.ctypes.data_as(C.c_void_p), c2p.ctypes.data_as(C.c_void_p), P.REPR_NTT)
                if rc != 0:
                    raise RuntimeError(_ffi.last_error())
                if it >= 3:
                    t_pin.append(time.perf_counter() - t_h2)
            t_p = sorted(t_pin)[len(t_pin) // 2]
            out["host_buffer_path"]["pinned_output"] = {
                "ms_per_encrypt": t_p * 1e3, "ms_min": min(t_pin) * 1e3, "parties_per_s": n_total / t_p,
                "note": "pvw_encrypt with c1 / c2 in pvw_host_alloc memory: the kernel writes them over PCIe as its workgroups finish",
                "bit_exact_vs_device_path": bool(np.array_equal(c2p.view(np.int64), c2.cpu().numpy()) and
                                                 np.array_equal(c1p.view(np.int64), c1.cpu().numpy()))}
            del c1p, c2p
            for pp in ptrs:
                lib.pvw_host_free(pp)
        except Exception as e:
            out["host_buffer_path"]["pinned_output"] = {"error": str(e)[:200]}
        # the ONLY place bench.py touches oracle/: the checker timed as the CPU baseline
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pvw_model as M
        import pvw_oracle as O
        n_cpu = min(n_per, 4096)
        orc = O.Oracle(moduli, l)
        a_hat = crs.matrix(P.REPR_NTT)
        b_hat = gpk.matrix(0, n_cpu, P.REPR_NTT)
        g_hat = params.gadget_polynomial(P.REPR_NTT)
        r = O.sample_cbd(SEED_ENC, M.DOM_R, 0, k, l, 0.5)
        e1 = O.sample_uniform(SEED_ENC, M.DOM_E1, 0, k, l, 100)
        e2 = O.sample_uniform(SEED_ENC, M.DOM_E2, 0, n_cpu, l, 200)
        sc = np.array([(i * 1000 + 1) % (1 << 32) for i in range(n_cpu)], dtype=np.uint64)
        c1o, c2o = orc.encrypt(a_hat, b_hat, g_hat, sc, r, e1, e2, serial_c1=True)   # also warms up
        # the run above doubles as a full-size parity check of this very bench workload
        same = bool(np.array_equal(c1o.view(np.int64), c1.cpu().numpy()) and
                    np.array_equal(c2o.view(np.int64), c2.cpu().numpy()[:n_cpu]))
        reps, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds and reps < 200:
            t1 = time.perf_counter()
            orc.encrypt(a_hat, b_hat, g_hat, sc, r, e1, e2, serial_c1=True)
            t_cpu += time.perf_counter() - t1
            reps += 1
        out["cpu_baseline"] = {
            "value": n_cpu * reps / t_cpu, "unit": "parties/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"{reps} x the same encrypt (n={n_cpu}, k={k}, l={l}, {L} limbs, explicit r/e1/e2) with oracle/pvw_oracle.c, "
                      f"OpenMP over parties, c1 loop serial as in crs.rs:188; {t_cpu:.1f} s of CPU work",
            "bit_exact_vs_gpu": same,
        }
    if rank == 0:
        print(json.dumps(out))
    if DIST_ON:
        import torch.distributed as dist
        dist.destroy_process_group()


def bench_keygen(args, world, rank, local_rank, dev):
    """Batched key generation b_i = s_i*A + e_i (public_key.rs:111-147, crs.rs:138-171) for the parties
    of this rank: batches of 16 parties, up to 128 per launch, against the transposed CRS on the matrix cores (gemm_digits)."""
    import numpy as np
    import torch  # noqa: F401

    import pvw_rs_amd as P
    from pvw_rs_amd import _ffi

    n_per, k, l, L, desc = CONFIGS[args.config or "c3"]
    moduli = W.bench_moduli(L)
    from pvw_rs_amd import dist as D
    lo, hi, _, _ = D.shard_ranges(n_per * world, k, world, rank)
    params = (P.PvwParametersBuilder().set_parties(n_per * world).set_dimension(k).set_l(l).set_moduli(moduli)
              .set_device(local_rank).set_shard(lo, hi, 0, k).build())
    h, lib = params._h, _ffi.lib()
    P.PvwCrs.new_deterministic(params, SEED_A)
    sk = np.zeros((n_per, k, l), dtype=np.int64)
    P.api._check(lib.pvw_sample_secret_keys(h, np.frombuffer(SEED_ENC, dtype=np.uint8).ctypes.data_as(C.c_void_p),
                                            lo, n_per, sk.ctypes.data_as(C.c_void_p)))
    seed = np.frombuffer(SEED_B, dtype=np.uint8).copy()

    def step():
        P.api._check(lib.pvw_keygen(h, lo, hi, sk.ctypes.data_as(C.c_void_p), None, seed.ctypes.data_as(C.c_void_p)))

    steps = max(1, min(args.steps, 5))
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    elapsed = time.perf_counter() - t0
    macs = L * l * n_per * k * k
    out = {
        "metric": "party public keys/s for batched PublicKey::generate (b_i = s_i*A + e_i), host sk in, B-hat resident out",
        "value": n_per * world * steps / elapsed, "unit": "keys/s", "n_gpus": world, "steps": steps, "warmup": 1,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "key generation at " + desc, "parties_per_gpu": n_per, "k": k, "l": l, "rns_limbs": L},
        "roofline": {"bound": "mfma", "kernel": "gemm_digits_kernel (i8 MFMA, 64 byte-products per modular MAC)",
                     "achieved": macs * 64 * 2 / (elapsed / steps) / 1e12, "peak": 5000.0, "unit": "TOP/s (i8)",
                     "frac": macs * 64 * 2 / (elapsed / steps) / 1e12 / 5000.0, "traffic": None,
                     "modular_macs_per_s": macs / (elapsed / steps),
                     "note": "whole pvw_keygen call incl. H2D of the secret keys, prologues and re-tiling of B-hat"},
    }
    if rank == 0:
        print(json.dumps(out))


def bench_decrypt(args, world, rank, local_rank, dev):
    """Batched decrypt_party_value (decryption.rs:249-278) of D dealer ciphertexts per GPU for one
    secret key: dealers are sharded over the ranks; the only exchange is the all-gather of the decoded
    D x u64 shares at the end of every step (RCCL; N > 1 only)."""
    import numpy as np
    import torch

    import pvw_rs_amd as P
    from pvw_rs_amd import _ffi

    D, k, l, L, desc = DECRYPT_CONFIGS[args.config or "c5shard"]
    moduli = W.bench_moduli(L)
    params = (P.PvwParametersBuilder().set_parties(D * world).set_dimension(k).set_l(l).set_moduli(moduli)
              .set_device(local_rank).build())
    h, lib = params._h, _ffi.lib()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    qmin = int(min(moduli))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # Two inputs of the same shape.  "dealt": what decrypt_party_shares is handed in the protocol -- D dealers' ciphertexts
    # of their shares for ONE party, made here by this library's own keygen and multi-dealer encrypt (builder-default
    # noise) -- and the step is timed on those.  "random": uniform residues, which decrypt to garbage; the inner
    # products cost the same, the decode takes its longest path on every ciphertext (its short cuts for noise-sized
    # values never apply), so this is the worst case and is reported beside the headline.
    dealt = torch.tensor([(rank * D + d) * 1000 + 1 for d in range(D)], dtype=torch.int64, device=dev)
    gen = (P.PvwParametersBuilder().set_parties(1).set_dimension(k).set_l(l).set_moduli(moduli)
           .set_secret_variance(W.SECRET_VARIANCE).set_error_bounds(W.ERROR_BOUND_1, W.ERROR_BOUND_2).set_device(local_rank).build())
    gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(gen, SEED_A))
    receiver = P.Party.new(0, gen, SEED_B)
    gpk.generate_all_party_keys([receiver], SEED_B)
    sk = torch.from_numpy(np.ascontiguousarray(receiver.secret_key.coefficients())).to(dev)
    c1s = torch.empty((D, k, L, l), dtype=torch.int64, device=dev)
    c2col = torch.empty((D, L, l), dtype=torch.int64, device=dev)
    for lo in range(0, D, 512):
        hi = min(D, lo + 512)
        seeds = np.concatenate([np.frombuffer(P.api._dealer_seed(SEED_ENC, rank * D + d), dtype=np.uint8) for d in range(lo, hi)]).copy()
        rc = lib.pvw_encrypt_multi_device(gen._h, C.c_void_p(dealt[lo:hi].data_ptr()), hi - lo, 1, seeds.ctypes.data_as(C.c_void_p),
                                          C.c_void_p(c1s[lo:hi].data_ptr()), C.c_void_p(c2col[lo:hi].data_ptr()), P.REPR_NTT, stream)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())
    torch.cuda.synchronize()
    del gpk, receiver, gen
    inputs = {"dealt": (c1s, c2col)}
    if args.worst_case:
        inputs["random"] = (torch.randint(0, qmin, (D, k, L, l), dtype=torch.int64, device=dev, generator=g),
                            torch.randint(0, qmin, (D, L, l), dtype=torch.int64, device=dev, generator=g))
    noisy = torch.zeros((D, L, l), dtype=torch.int64, device=dev)

    vals_dev = torch.zeros(D, dtype=torch.int64, device=dev)
    # config 5's one exchange step: every rank ends up with all D x world decoded shares (8 bytes each)
    gathered = torch.zeros(D * world, dtype=torch.int64, device=dev) if DIST_ON else None

    dkey = P.SecretKey.from_coefficients(params, sk.cpu().numpy()).load_device() if args.resident_key else None

    def step(which="dealt"):
        # inner products, INTT and gadget decode on the device: only D x u64 would leave the GPU
        a, b = inputs[which]
        fn = lib.pvw_decrypt_batch_device_sk if dkey else lib.pvw_decrypt_batch_device
        rc = fn(h, dkey._h if dkey else C.c_void_p(sk.data_ptr()), C.c_void_p(a.data_ptr()),
                                          C.c_void_p(b.data_ptr()), D, P.REPR_NTT,
                                          C.c_void_p(noisy.data_ptr()), C.c_void_p(vals_dev.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError(_ffi.last_error())
        if DIST_ON:
            import torch.distributed as dist
            dist.all_gather(list(gathered.chunk(world)), vals_dev)

    def barrier():
        torch.cuda.synchronize()
        if DIST_ON:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if DIST_ON:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    def kernel_times(which):
        params.set_profiling(True)
        params.reset_profiling()
        for _ in range(args.steps):
            step(which)
        torch.cuda.synchronize()
        out = {name: params.kernel_time(name) for name in ("decrypt_mac", "prep", "intt", "decode")}
        params.set_profiling(False)
        return out

    worst = None
    if args.worst_case:                               # the same step on uniform residues (untimed by the contract's clock)
        for _ in range(args.warmup):
            step("random")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step("random")
        torch.cuda.synchronize()
        t_rand = time.perf_counter() - t1
        kt_r = kernel_times("random")
        worst = {"input": "uniform residues (decrypt to garbage; every decode takes its longest path)",
                 "ms_per_step": t_rand / args.steps * 1e3, "value": D * args.steps / t_rand,
                 "kernel_ms_per_step": {name: v[0] / max(args.steps, 1) for name, v in kt_r.items()}}
        del inputs["random"]
    kt = kernel_times("dealt")
    dealt_ok = bool(torch.equal(vals_dev, dealt))
    mac_ms, launches = kt["decrypt_mac"]
    avg_s = mac_ms / max(launches, 1) * 1e-3
    # c1s + c2col reads, noisy write, s-hat read; a large batch runs as several launches
    # (chunks of dealers), each reading s-hat again: bytes per launch = that chunk's share
    per_step = max(launches // max(args.steps, 1), 1)
    alg_bytes = 8 * L * l * (D * k + D + D + k * per_step) // per_step
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    # host decode of the D noisy polynomials (decryption.rs:10-58), timed separately
    nz = noisy.cpu().numpy().view(np.uint64)
    t1 = time.perf_counter()
    vals = P.decode_scalar_pvw_host(params, nz)
    t_dec = time.perf_counter() - t1
    tr = measured_traffic(args.config or "c5shard", "decrypt_mac")
    out = {
        "metric": "dealer ciphertexts/s for batched decrypt_party_value (<sk,c1> - c2, INTT and gadget decode, all on the device)",
        "value": D * world * args.steps / elapsed, "unit": "ciphertexts/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic: D dealers' ciphertexts for one receiver from this library's keygen + multi-dealer encrypt (seeded)",
        "decrypts_to_the_dealt_values": dealt_ok, "worst_case_random_residues": worst,
        "config": {"workload": desc, "dealers_per_gpu": D, "k": k, "l": l, "rns_limbs": L,
                   "q_bits": int(params.q_total().bit_length()),
                   "sharding": f"dealer-sharded x{world}" + (", all-gather of D x u64 decoded shares per step" if DIST_ON else ""),
                   "world_size_observed": world, "backend": (os.environ.get("PVW_BENCH_BACKEND", "nccl") if DIST_ON else None),
                   "secret_key": "resident on the device in NTT form (pvw_sk_load)" if dkey else "coefficients handed over, transformed and wiped per call"},
        "roofline": {"bound": "hbm", "kernel": "decrypt_mac_fw_kernel" if L * l // 2 >= 128 else "decrypt_mac_grouped_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": tr[0] if tr else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": avg_s * 1e6, "launches_timed": launches,
                     "launches_per_step": per_step},
        "kernel_ms_per_step": {name: v[0] / max(args.steps, 1) for name, v in kt.items()},
        "host_decode_reference": {"seconds": t_dec, "ciphertexts": D,
                                  "note": "the same decode with host big integers (pvw_decode_host), outside the timed region",
                                  "matches_device": bool([int(x) for x in vals] == [int(x) for x in vals_dev.cpu().numpy().view(np.uint64)])},
    }
    if rank == 0:
        print(json.dumps(out))
    if DIST_ON:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
