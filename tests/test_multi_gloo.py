"""N>1 path on CPU: world_size-2 gloo run of the dictionary broadcast + voxel sharding + gather.
The per-shard compute is the CPU oracle here (no GPU in this container); on the GPU box the same
host logic feeds the HIP library (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from microstructure_fingerprinting_amd import dist as mdist
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = np.load(os.path.join(G, "fit_c2_small.npz"))
    V = c["Y"].shape[0]
    ms = sch = None
    if rank == 0:   # only rank 0 owns the dictionary before the broadcast
        ms = mfu.init_PGSE_multishell_interp(c["dictionary"], c["sch_ms"], np.array([0, 0, 1.0]))
        sch = c["sch_ms"]
    ms, sch = mdist.broadcast_interpolator(ms, sch, src=0)
    lo, hi = mdist.shard_range(V, rank, world)
    T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
    n = hi - lo
    P = orc.fit_batch(T, sch, c["Y"][lo:hi], np.full(n, 2), np.zeros(n, bool), np.zeros(n, bool), c["peaks"][lo:hi],
                      2, False, False, None, None, 0)
    full = mdist.gather_rows(P, V)
    np.save(os.path.join(outdir, "r%d.npy" % rank), full)
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy"); r1 = np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)
    c = np.load(os.path.join(G, "fit_c2_small.npz"))
    # the gathered rows equal the reference's single-process result (maps M0, fractions, MSE, R2)
    assert np.allclose(r0[:, 0], c["map_M0"], rtol=1e-9)
    assert np.allclose(r0[:, 1], c["map_frac_f0"], rtol=1e-9, atol=1e-12)
    assert np.allclose(r0[:, 2], c["map_frac_f1"], rtol=1e-9, atol=1e-12)
    assert np.allclose(r0[:, -2], c["map_MSE"], rtol=1e-9)
    assert np.allclose(r0[:, -1], c["map_R2"], rtol=1e-9)


def _worker_mixed(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from microstructure_fingerprinting_amd import dist as mdist
    from microstructure_fingerprinting_amd import mf_utils as mfu
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = np.load(os.path.join(G, "fit_cases.npz"))
    V = c["Y"].shape[0]
    ms = sch = None
    if rank == 0:
        ms = mfu.init_PGSE_multishell_interp(c["dictionary"], c["sch_ms"], np.array([0, 0, 1.0]))
        sch = c["sch"]
    ms, sch = mdist.broadcast_interpolator(ms, sch, src=0)
    idx = mdist.balanced_shard_indices(c["numfasc"], c["csf"], c["ear"], rank, world)
    T = {"S": ms.S, "N": ms.num_subs, "G_un": ms.Gms_un, "off": ms.off, "x": ms.x_flat, "Y": ms.Y_flat}
    E = int(c["E"])
    b = (orc.GAMMA_H * sch[:, 3] * sch[:, 5]) ** 2 * (sch[:, 4] - sch[:, 5] / 3)
    sig_csf = np.exp(-sch[:, 6] / float(c["T2_csf"])) * np.exp(-b * float(c["DIFF_csf"]))
    sig_ear = np.stack([np.exp(-sch[:, 6] / float(c["T2_ear"])) * np.exp(-b * x) for x in c["DIFF_ear"]], axis=1)
    def fit(sel):
        return orc.fit_batch(T, sch, c["Y"][sel], c["numfasc"][sel], c["csf"][sel].astype(bool), c["ear"][sel].astype(bool),
                             c["peaks"][sel], 2, True, True, sig_csf, sig_ear, E)
    P = fit(idx)
    full = mdist.gather_rows_indexed(P, idx, V)
    if rank == 0:
        np.save(os.path.join(outdir, "single.npy"), fit(np.arange(V)))
        np.save(os.path.join(outdir, "counts.npy"), np.array([idx.size]))
    np.save(os.path.join(outdir, "m%d.npy" % rank), full)
    np.save(os.path.join(outdir, "i%d.npy" % rank), idx)
    dist.destroy_process_group()


def test_two_rank_class_balanced_shards(tmp_path):
    """Mixed voxel classes (numfasc 0..2, CSF and EAR masks): every rank gets the same mix of classes, and the rows
    gathered by ROI index equal the single-process result."""
    world = 2
    mp.spawn(_worker_mixed, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m0 = np.load(tmp_path / "m0.npy"); m1 = np.load(tmp_path / "m1.npy")
    single = np.load(tmp_path / "single.npy")
    assert np.array_equal(m0, m1) and np.array_equal(m0, single)
    i0 = np.load(tmp_path / "i0.npy"); i1 = np.load(tmp_path / "i1.npy")
    c = np.load(os.path.join(G, "fit_cases.npz"))
    assert np.array_equal(np.sort(np.concatenate([i0, i1])), np.arange(c["Y"].shape[0]))
    key = c["numfasc"] * 4 + c["csf"] * 2 + c["ear"]
    for k in np.unique(key):   # per class the two shards differ by at most one voxel
        assert abs(int((key[i0] == k).sum()) - int((key[i1] == k).sum())) <= 1
    assert abs(i0.size - i1.size) <= 1
