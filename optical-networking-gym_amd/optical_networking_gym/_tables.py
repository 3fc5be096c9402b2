"""Static-table compiler: flattens the annotated topology graph into the arrays `ongym_config` (include/ongym.h) takes.

Host-side, init-time.  The reference walks Python objects on every request
(`topology[a][b]["index"]`, `path.links`, `link.spans`, `modulation.minimum_osnr`; envs/qrmsa.pyx:1255-1259,
core/osnr.pyx:48-55); the device kernels read these constant tables instead.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


@dataclass
class StaticTables:
    name: str
    n_nodes: int
    n_links: int
    n_paths: int
    k_paths: int
    max_hops: int
    node_names: list
    pair_paths: np.ndarray    # int32 [N, N, k]  path id or -1
    path_hops: np.ndarray     # int32 [P]
    path_links: np.ndarray    # int32 [P, max_hops]  (link "index"), -1 padded
    path_length: np.ndarray   # float64 [P]
    path_nodes: list          # list of node-index tuples (host only)
    link_nodes: np.ndarray    # int32 [E, 2]
    link_length: np.ndarray   # float64 [E]
    link_nspans: np.ndarray   # int32 [E]
    link_span_km: np.ndarray  # float64 [E]
    link_alpha: np.ndarray    # float64 [E]  1/m
    link_nf: np.ndarray       # float64 [E]  linear

    @staticmethod
    def from_topology(topology) -> "StaticTables":
        nodes = list(topology.graph.get("node_indices") or topology.nodes())
        nidx = {n: i for i, n in enumerate(nodes)}
        E = topology.number_of_edges()
        link_nodes = np.zeros((E, 2), np.int32)
        link_length = np.zeros(E)
        link_nspans = np.zeros(E, np.int32)
        link_span_km = np.zeros(E)
        link_alpha = np.zeros(E)
        link_nf = np.zeros(E)
        seen = set()
        for u, v, d in topology.edges(data=True):
            e = int(d["index"])
            if e in seen or not 0 <= e < E:
                raise ValueError(f"edge index {e} is not a permutation of 0..{E - 1}")
            seen.add(e)
            spans = d["link"].spans
            first = spans[0]
            for s in spans:
                if (s.length, s.attenuation_normalized, s.noise_figure_normalized) != (
                        first.length, first.attenuation_normalized, first.noise_figure_normalized):
                    raise ValueError("device tables need identical spans within a link "
                                     "(get_topology always builds them so, topology.pyx:288-299)")
            link_nodes[e] = (nidx[u], nidx[v])
            link_length[e] = d["length"]
            link_nspans[e] = len(spans)
            link_span_km[e] = first.length
            link_alpha[e] = first.attenuation_normalized
            link_nf[e] = first.noise_figure_normalized
        ksp = topology.graph["ksp"]
        k_paths = int(topology.graph.get("k_paths") or max(len(v) for v in ksp.values()))
        by_id = {}
        for routes in ksp.values():
            for p in routes:
                by_id[int(p.id)] = p
        P = len(by_id)
        if sorted(by_id) != list(range(P)):
            raise ValueError("path ids must be 0..P-1")
        max_hops = max(int(p.hops) for p in by_id.values())
        path_hops = np.zeros(P, np.int32)
        path_links = np.full((P, max_hops), -1, np.int32)
        path_length = np.zeros(P)
        path_nodes = [None] * P
        for pid, p in by_id.items():
            path_hops[pid] = p.hops
            path_links[pid, :p.hops] = [int(l.id) for l in p.links]
            path_length[pid] = float(p.length)
            path_nodes[pid] = tuple(nidx[n] for n in p.node_list)
        N = len(nodes)
        pair_paths = np.full((N, N, k_paths), -1, np.int32)
        for (a, b), routes in ksp.items():
            for k, p in enumerate(routes[:k_paths]):
                pair_paths[nidx[a], nidx[b], k] = int(p.id)
        return StaticTables(
            name=str(topology.graph.get("name", "")), n_nodes=N, n_links=E, n_paths=P, k_paths=k_paths,
            max_hops=max_hops, node_names=nodes, pair_paths=pair_paths, path_hops=path_hops, path_links=path_links,
            path_length=path_length, path_nodes=path_nodes, link_nodes=link_nodes, link_length=link_length,
            link_nspans=link_nspans, link_span_km=link_span_km, link_alpha=link_alpha, link_nf=link_nf)

    def truncated(self, k_paths: int) -> "StaticTables":
        """The same tables restricted to the first `k_paths` routes of every node pair (path ids and per-path arrays are
        kept, so ids stay those of the topology's `Path` objects)."""
        k_paths = int(k_paths)
        if k_paths == self.k_paths:
            return self
        if not 0 < k_paths < self.k_paths:
            raise ValueError(f"k_paths={k_paths} outside 1..{self.k_paths}")
        from dataclasses import replace
        return replace(self, k_paths=k_paths, pair_paths=np.ascontiguousarray(self.pair_paths[:, :, :k_paths]))

    @staticmethod
    def from_golden(tables: dict) -> "StaticTables":
        """Build from a `tests/golden/tables_*.json`-shaped dict (exported from the reference's get_topology)."""
        nodes = list(tables["node_names"])
        edges = sorted(tables["edges"], key=lambda e: e["index"])
        E, N, k = len(edges), len(nodes), int(tables["k_paths"])
        paths = {}
        for key, lst in tables["pairs"].items():
            for p in lst:
                paths[p["id"]] = p
        P = len(paths)
        max_hops = max(p["hops"] for p in paths.values())
        path_hops = np.zeros(P, np.int32)
        path_links = np.full((P, max_hops), -1, np.int32)
        path_length = np.zeros(P)
        path_nodes = [None] * P
        for pid, p in paths.items():
            path_hops[pid] = p["hops"]
            path_links[pid, :p["hops"]] = p["links"]
            path_length[pid] = p["length"]
            path_nodes[pid] = tuple(p["nodes"])
        pair_paths = np.full((N, N, k), -1, np.int32)
        for key, lst in tables["pairs"].items():
            i, j = (int(x) for x in key.split(","))
            for kk, p in enumerate(lst):
                pair_paths[i, j, kk] = p["id"]
                pair_paths[j, i, kk] = p["id"]
        return StaticTables(
            name=tables["name"], n_nodes=N, n_links=E, n_paths=P, k_paths=k, max_hops=max_hops, node_names=nodes,
            pair_paths=pair_paths, path_hops=path_hops, path_links=path_links, path_length=path_length,
            path_nodes=path_nodes,
            link_nodes=np.array([[e["a"], e["b"]] for e in edges], np.int32),
            link_length=np.array([e["length"] for e in edges], float),
            link_nspans=np.array([e["nspans"] for e in edges], np.int32),
            link_span_km=np.array([e["span_km"] for e in edges], float),
            link_alpha=np.array([e["alpha"] for e in edges], float),
            link_nf=np.array([e["nf"] for e in edges], float))


def modulation_arrays(modulations: Sequence) -> tuple:
    se = np.array([int(m.spectral_efficiency) for m in modulations], np.int32)
    thr = np.array([float(m.minimum_osnr) for m in modulations], np.float64)
    return se, thr


def cumulative(probabilities: Optional[Sequence[float]], n: int) -> np.ndarray:
    """itertools.accumulate of the weights, as CPython's random.choices builds cum_weights (Lib/random.py)."""
    w = np.full(n, 1.0 / n) if probabilities is None else np.asarray(probabilities, np.float64)
    if len(w) != n:
        raise ValueError("probability vector has the wrong length")
    out = np.zeros(n)
    acc = 0.0
    for i, x in enumerate(w):
        acc = acc + float(x)
        out[i] = acc
    return out
