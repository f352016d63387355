#!/usr/bin/env python3
"""bench.py -- headline benchmark: voxels/s of the per-voxel fingerprint matcher.

Workload (BASELINE.json configs[1], "C2"): 1e5 voxels, 2 fascicles, 782-atom x 200-measurement
multishell dictionary, exhaustive 2-sub-dictionary NNLS.  One "step" = one pass of the hot path
(rotation + exhaustive NNLS + parameter packing) over the whole 1e5-voxel batch, inputs already
resident in HBM.  With N GPUs every rank processes its own 1e5-voxel shard (weak scaling, no
data-path collective); rank 0 builds the dictionary tables and broadcasts them once over RCCL.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_VOXEL = 261.0e6      # 244.6 MFLOP Gram + 16.4 MFLOP vector work (the reference's FP64 count)
BYTES_PER_VOXEL = 1710.0      # y (1600 B) + peaks (48 B) + flags in, 56 B out
PEAK_FP64_MFMA_TFLOPS = 78.6  # AMD public spec, FP64 matrix (the CDNA4 guide lists no FP64 MFMA rate)
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense FP16/BF16 MFMA, MI355X_MICROARCH.md
# The dominant kernel (mfx_fit_k2s_kernel) ranks the atom pairs with a Gram computed from operands split in two
# FP16 halves: 3 MFMA products per 32x32x16 block over the padded problem (800 x 800 atoms x 208 rows).
EXEC_F16_FLOP_PER_VOXEL = 3 * 2.0 * 800 * 800 * 208
# FP32 screening table bytes a voxel pulls from L2: 6 passes over a rotated dictionary (200 rows x 782 atoms x 8 B) + the
# shared last tile's operand read by all 8 waves
L2_TABLE_BYTES_PER_VOXEL = 6 * 200 * 782 * 8.0 + 8 * 32 * 200 * 8.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--voxels", type=int, default=100000, help="voxels per GPU (default: BASELINE config 2)")
    ap.add_argument("--atoms", type=int, default=782)
    ap.add_argument("--cpu-sample", type=int, default=768, help="voxels timed on the host cores for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def build_model(atoms):
    """Host-side, once: synthetic dense scheme + dictionary -> per-shell knot tables."""
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from microstructure_fingerprinting_amd import synth
    sch, dic, rng = synth.make_model("C2", N=atoms)
    ms = mfu.init_PGSE_multishell_interp(dic, sch, np.array([0.0, 0.0, 1.0]))
    return sch, dic, ms


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from microstructure_fingerprinting_amd import _lib as L
    from microstructure_fingerprinting_amd import engine, synth
    from microstructure_fingerprinting_amd import mf_utils as mfu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MFX_BENCH_BACKEND", "nccl")     # "gloo" only for rehearsals on a 1-GPU box
    ndev = torch.cuda.device_count()
    dev_index = (local_rank % ndev) if world > 1 else 0
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- dictionary tables: built on rank 0, broadcast once (RCCL over xGMI), then staged to HBM
    ms = sch = None
    if rank == 0:
        sch, dic, ms = build_model(a.atoms)
    if world > 1:
        from microstructure_fingerprinting_amd import dist as mdist
        ms, sch = mdist.broadcast_interpolator(ms, sch, src=0, device=dev if backend == "nccl" else None)
    ms.device = dev.index or 0
    plan = engine.Plan(ms.device_tables(), scheme=sch)
    M, N = sch.shape[0], ms.num_subs

    # ---- synthetic voxels, generated on the device with the library's own rotation kernel
    V = a.voxels
    rng = np.random.default_rng(1000 + rank)
    peaks_h = np.concatenate([synth.unit_vectors(rng, V), synth.unit_vectors(rng, V)], axis=1)
    atoms_h = rng.integers(0, N, (V, 2)).astype(np.int32)
    nu_h = rng.dirichlet(np.ones(2), V)
    d_peaks = torch.from_numpy(peaks_h).to(dev)
    d_Y = torch.zeros((V, M), dtype=torch.float64, device=dev)
    for k in range(2):
        col = engine.rotate_columns_dev(plan, d_peaks[:, 3 * k:3 * k + 3].contiguous(),
                                        torch.from_numpy(atoms_h[:, k].copy()).to(dev))      # [V, M]
        d_Y += 500.0 * torch.from_numpy(nu_h[:, k:k + 1].copy()).to(dev) * col
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    d_Y += torch.randn((V, M), dtype=torch.float64, device=dev, generator=gen) * (500.0 / 30.0)
    d_out = torch.zeros((V, 7), dtype=torch.float64, device=dev)
    lib = L.lib()
    stream = torch.cuda.current_stream(dev)

    def step():
        L.check(lib.mfx_fit_batch_dev(plan.handle(), d_Y.data_ptr(), d_peaks.data_ptr(), 2, 0, 0, None, None, 0, V,
                                      d_out.data_ptr(), stream.cuda_stream))

    for _ in range(a.warmup):
        step()
    # ---- timed region: barrier + sync on both sides; HIP events on the launch stream for the kernel
    lib.mfx_set_profiling(1)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(a.steps):
        step()
        # the event pair is recorded on the launch stream inside the library; reading it waits
        # for that launch only (the next launch is queued right after)
        kern_ms.append(lib.mfx_last_kernel_ms())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    lib.mfx_set_profiling(0)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3
    value = world * V * a.steps / elapsed

    # sanity: selected atoms are plausible indices
    ids = d_out[:, 3:5]
    assert bool(((ids >= 0) & (ids < N)).all()), "atom ids out of range"

    res = None
    if rank == 0:
        kavg = float(np.mean(kern_ms)) if kern_ms and min(kern_ms) > 0 else None
        roof = None
        if kavg:
            ach = FLOP_PER_VOXEL * V / (kavg * 1e-3) / 1e12
            traffic = None
            tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json" if os.environ.get("MFX_K2_SCREEN", "1") == "0"
                              else "r01_pmc_traffic_k2s.json")
            if os.path.exists(tf) and V == 100000 and a.atoms == 782:
                try:
                    traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            screen = os.environ.get("MFX_K2_SCREEN", "1") != "0"
            if screen:
                # achieved: the reference's algorithmic FP64 flop count per second, priced against the dense MFMA
                # peak of the type the dominant kernel multiplies in (FP16).  exec_*: what the matrix pipe really did.
                roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic,
                        "kernel": "mfx_fit_k2s_kernel<13, false, 3>", "kernel_ms": round(kavg, 3),
                        "flop_per_voxel": FLOP_PER_VOXEL,
                        "exec_f16_mfma_tflops": round(EXEC_F16_FLOP_PER_VOXEL * V / (kavg * 1e-3) / 1e12, 1),
                        "exec_f16_mfma_frac": round(EXEC_F16_FLOP_PER_VOXEL * V / (kavg * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                        "vs_fp64_mfma_peak": round(ach / PEAK_FP64_MFMA_TFLOPS, 3),
                        "note": "pair screening on split-FP16 MFMA + exact FP64 re-evaluation; the kernel is VALU-issue/L2 bound, "
                                "see DESIGN.md 4.1",
                        "hbm_bytes_per_voxel_algorithmic": BYTES_PER_VOXEL,
                        "achieved_hbm_GBps_algorithmic": round(BYTES_PER_VOXEL * V / (kavg * 1e-3) / 1e9, 3),
                        # second roofline: the FP32 screening table is re-read from L2 6.3 times per
                        # voxel (D2 five times, D1 once, the shared last tile's operand eight times) at M rows x N atoms
                        # x 8 B per pass; an XCD's L2 delivers 66-73 GB/s per CU (MI355X_MICROARCH.md), 68 x 256 here
                        "l2_table_bytes_per_voxel": L2_TABLE_BYTES_PER_VOXEL,
                        "l2_table_TBps": round(L2_TABLE_BYTES_PER_VOXEL * V / (kavg * 1e-3) / 1e12, 2),
                        "l2_table_frac_of_17.4TBps": round(L2_TABLE_BYTES_PER_VOXEL * V / (kavg * 1e-3) / 17.4e12, 3)}
            else:
                roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": traffic,
                        "kernel": "mfx_fit_k2_kernel<50,false,true,8,2,2>", "kernel_ms": round(kavg, 3),
                        "flop_per_voxel": FLOP_PER_VOXEL, "hbm_bytes_per_voxel_algorithmic": BYTES_PER_VOXEL,
                        "achieved_hbm_GBps_algorithmic": round(BYTES_PER_VOXEL * V / (kavg * 1e-3) / 1e9, 3)}
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(sch, ms, d_Y, peaks_h, d_out, min(a.cpu_sample, V))
        res = {"metric": "voxels/sec, 2-fascicle exhaustive NNLS, 782-atom x 200-measurement dictionary",
               "value": round(value, 1), "unit": "voxels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms_per_step, 3), "ms_per_voxel": round(ms_per_step / V, 6),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64" if os.environ.get("MFX_K2_SCREEN", "1") == "0" else "f64 (pairs ranked on split-f16 MFMA, short list re-evaluated in f64)",
               "data": "synthetic",
               "config": {"workload": "C2: %d voxels/GPU, 2 fascicles, %d atoms x %d measurements" % (V, N, M),
                          "voxels_per_gpu": V, "global_voxels": world * V, "atoms": N, "measurements": M,
                          "sharding": "voxel shards, no data-path collective; dictionary broadcast once over RCCL"},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


def cpu_baseline(sch, ms, d_Y, peaks_h, d_out, nsample):
    """Time the CPU oracle (compiled restatement of the reference algorithm) on a bounded sample of
    the same voxels, on the host cores of this box, and check the GPU result against it."""
    from oracle import oracle as orc
    nthreads = max(1, min(16, os.cpu_count() or 1, orc.max_threads()))
    T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
    Ys = d_Y[:nsample].cpu().numpy()
    pk = np.ascontiguousarray(peaks_h[:nsample])
    K = np.full(nsample, 2, dtype=np.int32)
    z = np.zeros(nsample, dtype=np.uint8)
    # one-thread leg (reference parallel=False) on a smaller slice, all-threads leg (mp.Pool analogue)
    n1 = max(8, nsample // 16)
    t0 = time.perf_counter()
    orc.fit_batch(T, sch, Ys[:n1], K[:n1], z[:n1], z[:n1], pk[:n1], 2, False, False, None, None, 0, nthreads=1)
    t1 = time.perf_counter()
    ref = orc.fit_batch(T, sch, Ys, K, z, z, pk, 2, False, False, None, None, 0, nthreads=nthreads)
    t2 = time.perf_counter()
    got = d_out[:nsample].cpu().numpy()
    ids_equal = bool(np.array_equal(got[:, 3:5], ref[:, 3:5]))
    relerr = float(np.max(np.abs(got[:, :3] - ref[:, :3]) / np.maximum(np.abs(ref[:, :3]), 1e-300)))
    return {"value": round(nsample / (t2 - t1), 2), "unit": "voxels/s", "cores": nthreads, "kind": "port",
            "sample": "%d voxels of the same workload, %d OpenMP threads (mp.Pool analogue); "
                      "single-thread leg: %d voxels" % (nsample, nthreads, n1),
            "single_thread_value": round(n1 / (t1 - t0), 3),
            "parity_on_sample": {"atom_ids_equal": ids_equal, "max_rel_err_weights": relerr}}


if __name__ == "__main__":
    main()
