"""CPU-only checks of the host logic: topology compiler vs tables exported from the reference, C-ABI surface,
traffic-stream definition, multi-rank statistics reduction (gloo)."""
import ctypes
import json
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from common import GOLDEN, golden_tables, holder_for, jocn_modulations, load_traj, traj_requests
from cpython_traffic import cpython_request_stream
from optical_networking_gym import _native as nat
from optical_networking_gym._tables import StaticTables
from optical_networking_gym.topology import bundled_topology_path, get_topology
from optical_networking_gym.utils import rle
from oracle_lib import OracleEnv

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_TOPOLOGIES = "/root/reference/examples/topologies"


def tables_equal(a: StaticTables, b: StaticTables):
    assert (a.n_nodes, a.n_links, a.n_paths, a.k_paths, a.max_hops) == (b.n_nodes, b.n_links, b.n_paths, b.k_paths, b.max_hops)
    for f in ("pair_paths", "path_hops", "path_links", "link_nodes", "link_nspans"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)
    for f in ("path_length", "link_length", "link_span_km", "link_alpha", "link_nf"):
        np.testing.assert_allclose(getattr(a, f), getattr(b, f), rtol=1e-15, err_msg=f)
    assert a.path_nodes == b.path_nodes


@pytest.mark.parametrize("fname,gold", [("nsfnet_chen.txt", "nsfnet"), ("cost239.txt", "cost239"),
                                        ("ring_4.txt", "ring4"), ("nobel-eu.txt", "nobel-eu")])
def test_topology_compiler_matches_reference_tables(fname, gold):
    """own parser + span rule + networkx KSP vs the tables the reference's get_topology produced (golden)."""
    topo = get_topology(bundled_topology_path(fname), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    tables_equal(StaticTables.from_topology(topo), golden_tables(gold))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_TOPOLOGIES), reason="reference data files only exist in the build container")
@pytest.mark.parametrize("fname,gold", [("nobel-eu.xml", "nobel-eu"), ("germany50.xml", "germany50")])
def test_sndlib_reader_matches_reference_tables(fname, gold):
    topo = get_topology(os.path.join(REFERENCE_TOPOLOGIES, fname), None, jocn_modulations(), 80, 0.2, 4.5, 5)
    tables_equal(StaticTables.from_topology(topo), golden_tables(gold))


def test_span_rule_and_constants():
    t = golden_tables("nsfnet")
    e = {(int(a), int(b)): i for i, (a, b) in enumerate(t.link_nodes)}
    i = e[(0, 7)]                                 # 2400 km -> 30 x 80 km
    assert t.link_nspans[i] == 30 and t.link_span_km[i] == 80.0
    i = e[(1, 3)]                                 # 750 km -> 10 x 75 km
    assert t.link_nspans[i] == 10 and t.link_span_km[i] == 75.0
    assert t.link_alpha[0] == pytest.approx(2.3025850929940457e-05, rel=1e-15)
    assert t.link_nf[0] == pytest.approx(2.818382931264454, rel=1e-15)
    h = nat.ConfigHolder(t, modulations=jocn_modulations(), load=300)
    # the reference's -ffast-math build yields 0.0009999999999999994 for 0 dBm (qrmsa.pyx:288): 1 ulp from 10**-3
    assert h.struct.launch_power_w == pytest.approx(0.0009999999999999994, rel=1e-15) and h.reject_action == 9600


def test_unknown_topology_format():
    with pytest.raises(ValueError, match="Supplied topology format is unknown"):
        get_topology("nsfnet.h5")


def test_rle():
    s, v, l = rle(np.array([1, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1]))
    assert s.tolist() == [0, 3, 4, 8, 9] and v.tolist() == [1, 0, 1, 0, 1] and l.tolist() == [3, 1, 4, 1, 3]
    assert rle(np.array([])) == (None, None, None)


# ---- C ABI ---------------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "ongym.h")).read()
    declared = set(re.findall(r"\b(ongym_[a-z_]+)\s*\(", header))
    assert declared == set(nat.EXPORTED_SYMBOLS)
    lib = nat.load_library()            # dlopen + ABI version + struct sizes; no compute, works without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ongym_abi_version() == nat.ABI_VERSION
    assert lib.ongym_sizeof(2) == nat.STEP_DTYPE.itemsize == 56
    assert lib.ongym_sizeof(1) == nat.REQUEST_DTYPE.itemsize == 16


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    lib = nat.load_library()
    h = nat.ConfigHolder(golden_tables("nsfnet"), modulations=jocn_modulations(), load=300)
    out = ctypes.c_void_p()
    bad = nat.OngymConfig.from_buffer_copy(h.struct)
    bad.struct_size = 8
    assert lib.ongym_create(ctypes.byref(bad), ctypes.byref(out)) == -1
    assert b"mismatch" in lib.ongym_last_error(None)
    import torch
    if not torch.cuda.is_available():
        from optical_networking_gym.envs.batched import BatchedQRMSAEnv, OngymError
        with pytest.raises(OngymError, match="no usable HIP device"):
            BatchedQRMSAEnv(tables=golden_tables("nsfnet"), modulations=jocn_modulations(), load=300)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.load_library(str(tmp_path / "nope.so"))


# ---- traffic -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["traj_nsfnet320", "traj_nsfnet320_cont", "traj_nobeleu320"])
def test_reference_request_stream_is_the_documented_draw_order(tag):
    """A twin CPython generator (same seed) reproduces the captured reference requests bit for bit: pins the draw order
    expovariate, expovariate, choices, choices, choices|randint and the float32 rounding points (qrmsa.pyx:1079-1089)."""
    meta, d = load_traj(tag)
    t = golden_tables(meta["topology"])
    twin = cpython_request_stream(meta["seed"], 400, t.n_nodes, meta["load"], meta["mean_holding"],
                                  meta["bit_rates"] if meta["bit_rate_selection"] == "discrete" else None)
    assert twin.tobytes() == traj_requests(d)[:400].tobytes()


def test_device_stream_definition_statistics():
    """own counter-based stream (include/ongym_traffic.h), drawn through the oracle: distributions of A.7."""
    meta, _ = load_traj("traj_nsfnet320")
    h = holder_for(meta)
    o = OracleEnv(h)
    o.seed(123)
    reqs = []
    o.reset()
    last = 0.0
    for _ in range(6000):
        q = o.request()
        reqs.append((q["arrival_time"] - last, q["holding_time"], q["source"], q["destination"], q["bit_rate"]))
        last = float(q["arrival_time"])
        rc, r = o.step(o.reject_action)
    a = np.array(reqs, dtype=np.float64)
    assert a[:, 0].mean() == pytest.approx(10800 / 300, rel=0.05)
    assert a[:, 1].mean() == pytest.approx(10800, rel=0.05)
    assert (a[:, 2] != a[:, 3]).all()
    assert np.bincount(a[:, 2].astype(int), minlength=14).min() > 6000 / 14 * 0.7
    assert np.bincount(a[:, 3].astype(int), minlength=14).min() > 6000 / 14 * 0.7
    assert set(np.unique(a[:, 4])) == {10.0, 40.0, 100.0, 400.0}


def test_det_log_accuracy():
    """ongym_logf_det (exactly rounded float ops only): abs error < 1.5e-7 (+ relative 1e-7 of e*ln2) vs libm."""
    src = r'''
    #include <stdio.h>
    #include "%s/include/ongym_traffic.h"
    int main(void){ double worst=0; for (int i=1;i<=400000;i++){ double x=(double)i/400000.0; double a=ongym_logf_det(x), b=log(x);
      double e=fabs(a-b)/fmax(fabs(b),1.0); if(e>worst) worst=e;} double x=ldexp(1.0,-53); printf("%%.3e %%.9g %%.9g\n", worst, (double)ongym_logf_det(x), log(x)); return 0; }
    ''' % REPO
    exe = os.path.join("/tmp", "ongym_det_log_test")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-x", "c", "-", "-o", exe, "-lm"], input=src.encode(), check=True)
    worst, a, b = subprocess.run([exe], capture_output=True, check=True).stdout.split()
    assert float(worst) < 1.5e-7 and float(a) == pytest.approx(float(b), rel=2e-7)


# ---- multi-rank statistics reduction (the only collective of the N>1 path) ---------------------------------------------
_WORKER = r'''
import os, sys, json
sys.path[:0] = [%r, %r]
import numpy as np, torch, torch.distributed as dist
from optical_networking_gym._dist import init_process_group, shard_bounds, reduce_run_statistics
from common import holder_for, load_traj
from oracle_lib import OracleEnv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
init_process_group("gloo")
meta, _ = load_traj("traj_nsfnet320")
h = holder_for(meta, batch=5)
steps = 0; acc = 0
base, n = shard_bounds(5, rank, 2)        # 5 global replicas over 2 ranks (3 + 2); the oracle stands in for the GPU engine
for r in range(base, base + n):          # global replica index = stream index, as env.seed(seed, replica_base=base) does
    o = OracleEnv(h, replica=r); o.seed(7); o.reset()
    rec = o.run_first_fit(150); steps += len(rec); acc += int(rec["accepted"].sum())
delta, dt_max, k_ms = reduce_run_statistics(np.array([steps, acc], np.float64), 1.0 + rank, 10.0 * (rank + 1), dist, device="cpu")
if rank == 0:
    print(json.dumps(dict(delta=delta.tolist(), dt_max=dt_max, k_ms=k_ms)))
dist.destroy_process_group()
'''


def test_two_rank_statistics_reduction_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % (os.path.join(REPO, "optical-networking-gym_amd"), os.path.join(REPO, "tests")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    # ground truth: the UNSHARDED run of the same 5 replicas in one process — sharded == unsharded, exactly
    from optical_networking_gym._dist import shard_bounds
    assert [shard_bounds(5, k, 2) for k in range(2)] == [(0, 3), (3, 2)]
    assert [shard_bounds(65536, k, 8) for k in (0, 7)] == [(0, 8192), (57344, 8192)]
    meta, _ = load_traj("traj_nsfnet320")
    h = holder_for(meta, batch=5)
    steps = acc = 0
    for r in range(5):
        o = OracleEnv(h, replica=r); o.seed(7); o.reset()
        rec = o.run_first_fit(150); steps += len(rec); acc += int(rec["accepted"].sum())
    assert got["delta"] == [float(steps), float(acc)]
    assert got["dt_max"] == 2.0 and got["k_ms"] == 20.0


_GRAD_WORKER = r'''
import json, os, sys
sys.path[:0] = [%r]
import torch, torch.distributed as dist
from optical_networking_gym._dist import allreduce_mean_gradients, gather_per_rank, init_process_group
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
init_process_group("gloo")
torch.manual_seed(0)                        # same initial weights on every rank, as tools/bench_rl.py --learner
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
head = torch.nn.Linear(3, 1)                # a parameter group without gradient on this step must be skipped, not crash
params = list(net.parameters()) + list(head.parameters())
opt = torch.optim.SGD(list(net.parameters()), lr=0.1)
x = torch.arange(24, dtype=torch.float32).reshape(4, 6) * (rank + 1) / 10      # every rank sees its own shard of the batch
loss = net(x).pow(2).mean()
loss.backward()
own = [p.grad.clone() for p in net.parameters()]
allreduce_mean_gradients(params, dist)
opt.step()
per_rank = gather_per_rank([float(rank), float(sum(g.sum() for g in own))], dist, device="cpu")
if rank == 0:
    print(json.dumps(dict(grads=[p.grad.reshape(-1).tolist() for p in net.parameters()],
                          weights=[p.detach().reshape(-1).tolist() for p in net.parameters()], per_rank=per_rank)))
dist.destroy_process_group()
'''


def test_two_rank_gradient_allreduce_gloo(tmp_path):
    """the data-parallel learner of tools/bench_rl.py --gpus N: one flat bucket, one all-reduce, mean over the ranks, the
    same update everywhere (world_size 2 on gloo; on the GPUs the same call runs over RCCL)."""
    import torch
    script = tmp_path / "grad_worker.py"
    script.write_text(_GRAD_WORKER % os.path.join(REPO, "optical-networking-gym_amd"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    # ground truth in one process: the mean of the two ranks' gradients, one SGD step
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    grads = []
    for rank in range(2):
        net.zero_grad()
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) * (rank + 1) / 10
        net(x).pow(2).mean().backward()
        grads.append([p.grad.clone() for p in net.parameters()])
    for i, p in enumerate(net.parameters()):
        mean = (grads[0][i] + grads[1][i]) / 2
        np.testing.assert_allclose(got["grads"][i], mean.reshape(-1).numpy(), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(got["weights"][i], (p.detach() - 0.1 * mean).reshape(-1).numpy(), rtol=1e-6, atol=1e-8)
    assert [r[0] for r in got["per_rank"]] == [0.0, 1.0]
    assert got["per_rank"][0][1] != got["per_rank"][1][1]       # each rank really had its own gradients before the reduction


def test_bench_line_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py's `issue` / `roofline.traffic` figures come from the tracked counter summary (profiles/pmc_summary.json); every
    entry carries the hash of the kernel sources it was measured on and bench.py marks an entry of other sources stale."""
    sys.path.insert(0, REPO)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    summ = json.load(open(os.path.join(REPO, "profiles", "pmc_summary.json")))
    for key in ("nsfnet320", "cost239_320", "nobeleu768", "nsfnet320_p1", "nsfnet320_p2", "nsfnet320_p10"):
        assert "kernel_source_sha" in summ[key] and summ[key]["valu_per_env_step"] > 0, key
    assert bench.pmc_key("nsfnet320", 0) == "nsfnet320" and bench.pmc_key("nsfnet320", 10) == "nsfnet320_p10"
    fresh = bench.pmc_summary("nsfnet320", 1)
    assert fresh["stale"] == (fresh["kernel_source_sha"] != h)
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)          # any other sources
    assert bench.pmc_summary("nsfnet320", 1)["stale"] is True
    assert bench.pmc_summary("no_such_workload") is None
    # busy fractions: active cycles against the SIMD-clocks one env-step takes at the resident waves per SIMD
    pm = dict(valu_active_cycles_per_env_step=1000.0, wave_cycles_per_env_step=10000.0)
    assert bench.busy_frac(pm, "valu_active_cycles_per_env_step", dict(blocks_per_cu=20)) == pytest.approx(0.5)
    assert bench.busy_frac(pm, "salu_active_cycles_per_env_step", dict(blocks_per_cu=20)) is None


# ---- the reference's own unit tests, restated (tests/test_utils.py, tests/test_rmsa.py of the reference) -----------------
def test_span_link_and_rmsa_plumbing():
    from optical_networking_gym.envs.rmsa import RMSAEnv
    from optical_networking_gym.topology import Link, Span
    s = Span(length=80, attenuation=0.2, noise_figure=4.5)
    assert s.attenuation_normalized != 0 and s.noise_figure_normalized != 0
    before = s.attenuation_normalized
    s.set_attenuation(0.3)
    assert s.attenuation_normalized != before
    before = s.noise_figure_normalized
    s.set_noise_figure(6)
    assert s.noise_figure_normalized != before and str(s) != ""
    spans = tuple(Span(length=10.0 * (i + 1), attenuation=0.2, noise_figure=4.5) for i in range(5))
    assert len(Link(id=0, node1="t", node2="d", length=150.0, spans=spans).spans) == 5
    # BASELINE config 1: get_topology(nsfnet_chen.txt, "NSFNET", mods, 80, 0.2, 4.5, 5) + RMSAEnv(topology, 360)
    topology = get_topology(bundled_topology_path("nsfnet_chen.txt"), "NSFNET", jocn_modulations(), 80, 0.2, 4.5, 5)
    env = RMSAEnv(topology=topology, num_spectrum_resources=360)
    assert env.spectrum_use.shape == (22, 360) and topology.name == "NSFNET"
    assert len(topology.graph["ksp"]["1", "13"]) == 5 and topology.graph["ksp"]["1", "13"][0].hops == 3


def test_fragmentation_helpers_match_the_reference():
    """utils.rle / link_shannon_entropy_ / fragmentation_route_cuts / fragmentation_route_rss (utils.pyx:44-110) against
    values computed by the reference's own functions on random rows (tests/golden/kats_utils.json)."""
    from optical_networking_gym.utils import (fragmentation_route_cuts, fragmentation_route_rss, link_shannon_entropy_, rle)
    cases = json.load(open(os.path.join(GOLDEN, "kats_utils.json")))["cases"]
    assert len(cases) == 40
    for c in cases:
        rows = np.array(c["rows"], np.int32)
        starts, values, lengths = rle(rows[0])
        assert [starts.tolist(), values.tolist(), lengths.tolist()] == c["rle"]
        assert [link_shannon_entropy_(r.tolist()) for r in rows] == c["entropy"]
        assert fragmentation_route_cuts([r.tolist() for r in rows]) == c["cuts"]
        assert fragmentation_route_rss([r.tolist() for r in rows]) == c["rss"]
