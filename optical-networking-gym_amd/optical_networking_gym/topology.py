"""Static network description: value classes, topology-file readers and the k-shortest-path table.

Host-side, init-time only (never on the per-request path).  Public names and field names mirror the reference's
`optical_networking_gym/topology.pyx` so that callers (`examples/JOCN_Benchmark_2024/graph_load.py:12,297-314`) and the
heuristic plugins keep working:

* value classes ``Span`` (topology.pyx:11-34), ``Link`` (:36-51), ``Modulation`` (:53-70), ``Path`` (:72-95)
* ``read_txt_file`` (:215-241), ``read_sndlib_topology`` (:149-212), ``get_topology`` (:244-369)

What the batched device kernels consume is not this graph but the flat tables `_tables.StaticTables` derives from it.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from itertools import islice
from typing import Optional, Sequence, Tuple
from xml.dom import minidom

import networkx as nx
import numpy as np

_DB_PER_NEPER_KM = 2 * 10 * np.log10(np.exp(1)) * 1e3  # dB/km -> 1/m (field attenuation), topology.pyx:21


class Span:
    """One amplified fibre span. ``attenuation_normalized`` is in 1/m, ``noise_figure_normalized`` is linear."""

    __slots__ = ("length", "attenuation_db_km", "attenuation_normalized", "noise_figure_db", "noise_figure_normalized")

    def __init__(self, length: float, attenuation: float, noise_figure: float):
        self.length = float(length)
        self.set_attenuation(attenuation)
        self.set_noise_figure(noise_figure)

    def set_attenuation(self, attenuation: float) -> None:
        self.attenuation_db_km = float(attenuation)
        self.attenuation_normalized = self.attenuation_db_km / _DB_PER_NEPER_KM

    def set_noise_figure(self, noise_figure: float) -> None:
        self.noise_figure_db = float(noise_figure)
        self.noise_figure_normalized = 10 ** (self.noise_figure_db / 10)

    def __repr__(self) -> str:
        return (f"Span(length={self.length:.2f}, attenuation_db_km={self.attenuation_db_km}, "
                f"noise_figure_db={self.noise_figure_db})")


@dataclass
class Link:
    id: int
    node1: str
    node2: str
    length: float
    spans: Tuple[Span, ...]


@dataclass
class Modulation:
    name: str
    maximum_length: float
    spectral_efficiency: int
    minimum_osnr: float = 0.0
    inband_xt: float = 0.0


@dataclass
class Path:
    id: int
    k: int
    node_list: Tuple[str, ...]
    links: Tuple[Link, ...]
    hops: int
    length: float
    best_modulation: Optional[Modulation] = None

    def get_node_list(self) -> Tuple[str, ...]:
        return self.node_list


def get_k_shortest_paths(G: nx.Graph, source: str, target: str, k: int, weight=None):
    """First k loop-free paths in non-decreasing weight (Yen, via networkx); topology.pyx:100-104."""
    return tuple(islice(nx.shortest_simple_paths(G, source, target, weight=weight), k))


def get_path_weight(graph: nx.Graph, path: Sequence[str], weight: str = "length"):
    return np.sum([graph[u][v][weight] for u, v in zip(path[:-1], path[1:])])


def get_best_modulation_format(length: float, modulations: Sequence[Modulation]) -> Modulation:
    """Most spectrally efficient format whose reach covers `length` (topology.pyx:372-384)."""
    for mod in sorted(modulations, key=lambda m: m.spectral_efficiency, reverse=True):
        if length <= mod.maximum_length:
            return mod
    raise ValueError("It was not possible to find a suitable MF for a path with {} km".format(length))


get_best_modulation_format_by_length = get_best_modulation_format


def calculate_geographical_distance(latlong1, latlong2) -> float:
    """Haversine distance in km with r = 6373 km; arguments are (lon, lat) in degrees (topology.pyx:125-146)."""
    lon1, lat1, lon2, lat2 = map(math.radians, (latlong1[0], latlong1[1], latlong2[0], latlong2[1]))
    a = math.sin((lat2 - lat1) / 2) ** 2 + math.cos(lat1) * math.cos(lat2) * math.sin((lon2 - lon1) / 2) ** 2
    return 6373.0 * (2 * math.atan2(math.sqrt(a), math.sqrt(1 - a)))


def read_sndlib_topology(file_name: str) -> nx.Graph:
    """SNDlib native XML: nodes in document order, duplicate links skipped, edge `index` = order of first appearance;
    lengths are haversine km rounded to 3 decimals for geographical coordinates, Euclidean otherwise."""
    doc = minidom.parse(file_name).documentElement
    graph = nx.Graph()
    graph.graph["coordinatesType"] = doc.getElementsByTagName("nodes")[0].getAttribute("coordinatesType")
    text = lambda el, tag: el.getElementsByTagName(tag)[0].childNodes[0].data  # noqa: E731
    for position, node in enumerate(doc.getElementsByTagName("node")):
        graph.add_node(node.getAttribute("id"), pos=(float(text(node, "x")), float(text(node, "y"))), id=position)
    geographic = graph.graph["coordinatesType"] == "geographical"
    next_index = 0
    for link in doc.getElementsByTagName("link"):
        u, v = text(link, "source"), text(link, "target")
        if graph.has_edge(u, v):
            continue
        pu, pv = graph.nodes[u]["pos"], graph.nodes[v]["pos"]
        dist = calculate_geographical_distance(pu, pv) if geographic else math.hypot(pu[0] - pv[0], pu[1] - pv[1])
        graph.add_edge(u, v, id=link.getAttribute("id"), weight=1.0, length=np.around(dist, 3), index=next_index)
        next_index += 1
    return graph


def read_txt_file(file_name: str) -> nx.Graph:
    """Plain text: '#' lines dropped; line 0 = node count N (nodes are "1".."N"); line 1 is skipped unconditionally
    (link count); every further non-empty line is `src dst length_km`.  Edge `index` = order of appearance.
    (A file without the link-count line therefore loses its first edge — the reference's ring_4.txt quirk.)
    Superset of the reference format: a length containing '.' is read as float."""
    graph = nx.Graph()
    with open(file_name, "r", encoding="utf-8") as handle:
        rows = [ln for ln in handle if not ln.startswith("#")]
    for n in range(1, int(rows[0]) + 1):
        graph.add_node(str(n), name=str(n))
    index = 0
    for ln in rows[2:]:
        if len(ln) <= 1:
            continue
        u, v, km = ln.replace("\n", "").split(" ")[:3]
        graph.add_edge(u, v, id=index, index=index, weight=1, length=float(km) if "." in km else int(km))
        index += 1
    return graph


def _make_spans(length: float, max_span_length: float, attenuation: float, noise_figure: float) -> Tuple[Span, ...]:
    # topology.pyx:288-299: floor division, at least one, one more unless the length is an exact multiple
    count = int(length // max_span_length) or 1
    if length % max_span_length != 0:
        count += 1
    return tuple(Span(length / count, attenuation, noise_figure) for _ in range(count))


def get_topology(
    file_path: str,
    topology_name: Optional[str] = None,
    modulations: Optional[Tuple[Modulation, ...]] = None,
    max_span_length: float = 100,
    default_attenuation: float = 0.2,
    default_noise_figure: float = 4.5,
    k_paths: int = 5,
) -> nx.Graph:
    """Read a topology file, split links into equal spans, compute k shortest paths (by length) for every unordered
    node pair and index the nodes.  Returns the annotated ``nx.Graph`` the env constructor takes."""
    if file_path.endswith(".xml"):
        topology = read_sndlib_topology(file_path)
    elif file_path.endswith(".txt"):
        topology = read_txt_file(file_path)
    else:
        raise ValueError("Supplied topology format is unknown")
    if topology_name is None:
        topology_name = os.path.splitext(os.path.basename(file_path))[0]

    topology.graph["has_links_object"] = True
    for u, v, data in topology.edges(data=True):
        data["link"] = Link(id=data.get("index", f"{u}-{v}"), node1=u, node2=v, length=data["length"],
                            spans=_make_spans(data["length"], max_span_length, default_attenuation,
                                              default_noise_figure))

    ksp = {}
    ordered = list(topology.nodes())
    next_path_id = 0
    for i, src in enumerate(ordered):
        for dst in ordered[i + 1:]:
            routes = []
            for rank, nodes in enumerate(get_k_shortest_paths(topology, src, dst, k_paths, weight="length")):
                length = get_path_weight(topology, nodes, weight="length")
                routes.append(Path(
                    id=next_path_id, k=rank, node_list=tuple(nodes),
                    links=tuple(topology[a][b]["link"] for a, b in zip(nodes[:-1], nodes[1:])),
                    hops=len(nodes) - 1, length=length,
                    best_modulation=get_best_modulation_format(length, modulations) if modulations is not None else None))
                next_path_id += 1
            # the SAME list serves both directions: the reverse direction walks node_list/links in forward order
            ksp[src, dst] = routes
            ksp[dst, src] = routes

    topology.graph["name"] = topology_name
    topology.graph["ksp"] = ksp
    if modulations is not None:
        topology.graph["modulations"] = modulations
    topology.graph["k_paths"] = k_paths
    topology.graph["node_indices"] = ordered
    for idx, node in enumerate(ordered):
        topology.nodes[node]["index"] = idx
    return topology


TOPOLOGY_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "topologies")


def bundled_topology_path(name: str) -> str:
    """Path of a topology data file shipped with this package (nsfnet_chen.txt, cost239.txt, nobel-eu.txt, ring_4.txt)."""
    from ._topology_data import GENERATED, materialize
    if name in GENERATED:
        return materialize(name)
    path = os.path.join(TOPOLOGY_DIR, name)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return path
