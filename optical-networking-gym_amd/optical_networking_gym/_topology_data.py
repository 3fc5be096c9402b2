"""Small public topologies kept as Python literals; `materialize()` writes them in the plain-text topology format
(node count, link count, then `src dst length_km` per link; link index = order of appearance) that `read_txt_file` parses.

NSFNET: 14 nodes / 22 links with the link lengths (km) used by DeepRMSA (DOI 10.1109/jlt.2019.2923615).
RING_4_QUIRK: the 4-node ring of the reference's examples, *without* the link-count line — the reader skips line 1
unconditionally, so the first edge is swallowed and only 3 edges remain; kept to test ragged k-paths.
"""
import os
import tempfile

NSFNET = (14, (
    (1, 2, 1050),
    (1, 3, 1500),
    (1, 8, 2400),
    (2, 3, 600),
    (2, 4, 750),
    (3, 6, 1800),
    (4, 5, 600),
    (4, 11, 1950),
    (5, 6, 1200),
    (5, 7, 600),
    (6, 10, 1050),
    (6, 14, 1800),
    (7, 8, 750),
    (7, 10, 1350),
    (8, 9, 750),
    (9, 10, 750),
    (9, 12, 300),
    (9, 13, 300),
    (11, 12, 600),
    (11, 13, 750),
    (12, 14, 300),
    (13, 14, 150),
))

RING_4 = (4, ((1, 2, 250), (2, 3, 250), (3, 4, 250), (4, 1, 250)))


def _text(nodes, edges, with_link_count=True):
    head = [str(nodes)] + ([str(len(edges))] if with_link_count else [])
    return "\n".join(head + [f"{a} {b} {km}" for a, b, km in edges]) + "\n"


GENERATED = {
    "nsfnet_chen.txt": lambda: _text(*NSFNET),
    "ring_4.txt": lambda: _text(*RING_4, with_link_count=False),
}


def materialize(name: str) -> str:
    """Write the generated topology `name` into a per-user cache directory and return its path."""
    cache = os.path.join(tempfile.gettempdir(), f"ongym_topologies_{os.getuid()}")
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, name)
    text = GENERATED[name]()
    if not os.path.exists(path) or open(path).read() != text:
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, path)
    return path
