#!/usr/bin/env python3
"""Edge-list + partition text files (reference formats) -> the engine's binary graph container (cognn_amd/host/graph.h):
    python tools/convert_graph.py <edge list> <partition file> <out.cgb>
bin/gcn-optimize accepts the container in place of the edge-list argument (the partition argument is then ignored)."""
import struct
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cognn_amd.worker import read_edge_list, read_partition  # noqa: E402


def write_binary(path, src, dst, part):
    with open(path, "wb") as f:
        f.write(b"COGNNBG1")
        f.write(struct.pack("<QQ", len(part), len(src)))
        f.write(np.ascontiguousarray(src, dtype="<i8").tobytes())
        f.write(np.ascontiguousarray(dst, dtype="<i8").tobytes())
        f.write(np.ascontiguousarray(part, dtype="<i4").tobytes())


if __name__ == "__main__":
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    s, d = read_edge_list(sys.argv[1])
    write_binary(sys.argv[3], s, d, read_partition(sys.argv[2]))
