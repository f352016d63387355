#!/usr/bin/env python3
"""Time mfx_monte_carlo_average_dev on a dictionary-generation-sized job and the CPU oracle beside it.
   python tools/dev_time_mc.py [n_spin] [n_seq] [n_ref]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from microstructure_fingerprinting_amd import _lib as L  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_spin = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 552
n_ref = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dim = 3
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
planes = torch.randn((dim, n_ref * n_spin), dtype=torch.float64, device=dev, generator=gen) * 2.0
rng = np.random.default_rng(0)
dm = np.sort(rng.integers(0, n_ref, n_seq)).astype(np.int64)
gs = rng.uniform(-1.5, 1.5, (n_seq, dim))
out = np.zeros(n_seq)
lib = L.lib()
st = torch.cuda.current_stream(dev)


def run():
    L.check(lib.mfx_monte_carlo_average_dev(planes.data_ptr(), n_ref * n_spin, 1, n_ref * n_spin, dim, L.lptr(dm), L.dptr(gs),
                                            0.9, n_spin, n_seq, L.dptr(out), st.cuda_stream))


run()
lib.mfx_set_profiling(1)
ks, ws = [], []
for _ in range(5):
    t0 = time.perf_counter(); run(); ws.append(time.perf_counter() - t0); ks.append(lib.mfx_last_kernel_ms())
lib.mfx_set_profiling(0)
kms = float(np.mean(ks))
terms = n_seq * n_spin
# CPU: the oracle on a bounded sample of the same job (all sequences, fewer spins)
ns = min(n_spin, 40000)
ph_h = np.ascontiguousarray(planes[:, :].reshape(dim, n_ref, n_spin)[:, :, :ns].permute(1, 2, 0).reshape(n_ref * ns, dim).cpu().numpy())
nt = max(1, min(16, os.cpu_count() or 1, orc.max_threads()))
t0 = time.perf_counter(); ref = orc.monte_carlo_average(ph_h, dm, gs, 0.9, ns, nthreads=nt); tc = time.perf_counter() - t0
# parity of that sample through the device path (row-major host entry)
got = np.zeros(n_seq)
L.check(lib.mfx_monte_carlo_average(L.dptr(ph_h), ph_h.shape[0], dim, L.lptr(dm), L.dptr(gs), 0.9, ns, n_seq, L.dptr(got), 0))
print(json.dumps({"workload": "%d sequences x %d spins x %d components, %d simulated acquisitions" % (n_seq, n_spin, dim, n_ref),
                  "kernel_ms": round(kms, 3), "wall_ms": round(float(np.mean(ws)) * 1e3, 3),
                  "Gterms_per_s": round(terms / kms / 1e6, 2),
                  "phase_bytes_unique": n_ref * n_spin * dim * 8,
                  "cpu_oracle_Gterms_per_s": round(n_seq * ns / tc / 1e9, 4), "cpu_threads": nt,
                  "max_abs_diff_sample": float(np.max(np.abs(got - ref)))}))
