"""Reader (and a minimal writer) of faiss `IVF{n},Flat` index files - the `added_IVF*_Flat_*.index` files RVC users have.

The reference builds them in RVCTrainModelNode.train_index (custom_nodes/rvc_nodes.py:500-554: `faiss.index_factory(dim, f"IVF{n_ivf},Flat")`,
train, add, `faiss.write_index`) and opens them with `faiss.read_index` + `index.reconstruct_n(0, index.ntotal)` (pitch_extraction.py:52-73).
faiss (third party, unpinned in requirements.txt) is optional here: all the conversion path needs from the file are the stored vectors in id
order (= big_npy), which the device-resident exact search then owns (lib/feature_index.py).  Layout restated from faiss 1.7.x
`impl/index_write.cpp` / `index_read.cpp` (little-endian, `size_t` = 8 bytes):

    "IwFl"                                              IndexIVFFlat
      header: d i32 | ntotal i64 | dummy i64 | dummy i64 | is_trained u8 | metric_type i32 [| metric_arg f32 if metric_type > 1]
      nlist u64 | nprobe u64
      quantizer: "IxF2" / "IxFI" / "IxFl" + header + count u64 (in floats) + centroids f32[nlist * d]
      direct map: type u8 | array: count u64 + i64[count] [| hashtable: count u64 + (i64, i64)[count] when type == 2]
    "ilar"                                              ArrayInvertedLists
      nlist u64 | code_size u64 (= 4 d) | "full": count u64 + sizes u64[nlist]   or   "sprs": count u64 + (list, size) u64 pairs
      per non-empty list: codes f32[n * d] | ids i64[n]

A flat file ("IxF2" ... alone, `IndexFlatL2`) is read too.  Anything else (PQ / SQ / HNSW ...) raises ValueError naming the fourcc.
No file of the reference's own is reachable offline, so the reader is pinned against this module's writer, which emits the layout above
byte for byte (tests/test_host_logic.py) - a first-run check against a real index is listed in INTEGRATION.md.
"""
import struct

import numpy as np


def _fourcc(s):
    return struct.unpack("<I", s.encode("ascii"))[0]


class _Reader:
    def __init__(self, buf):
        self.b, self.o = memoryview(buf), 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def array(self, dtype, n):
        dt = np.dtype(dtype).newbyteorder("<")
        nbytes = int(n) * dt.itemsize
        if self.o + nbytes > len(self.b):
            raise ValueError("faiss index file is truncated")
        a = np.frombuffer(self.b, dtype=dt, count=int(n), offset=self.o)
        self.o += nbytes
        return a

    def fourcc(self):
        raw = bytes(self.b[self.o:self.o + 4])
        self.o += 4
        return raw.decode("ascii", "replace")


def _read_header(r):
    d = r.take("i")
    ntotal = r.take("q")
    r.take("q"); r.take("q")
    is_trained = r.take("B")
    metric = r.take("i")
    if metric > 1:
        r.take("f")
    if d <= 0 or ntotal < 0:
        raise ValueError(f"implausible faiss index header (d = {d}, ntotal = {ntotal})")
    return d, ntotal, bool(is_trained), metric


def _read_flat(r, tag):
    d, ntotal, _, metric = _read_header(r)
    n = r.take("Q")
    if n != ntotal * d:
        raise ValueError(f"{tag}: {n} stored floats for ntotal * d = {ntotal * d}")
    return r.array(np.float32, n).reshape(ntotal, d), metric


def read_index_vectors(path_or_bytes):
    """-> (vectors float32 [ntotal, d] in id order, info dict).  Accepts IndexIVFFlat ("IwFl") and IndexFlat ("IxF2" / "IxFI" / "IxFl") files."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    r = _Reader(buf)
    tag = r.fourcc()
    if tag in ("IxF2", "IxFI", "IxFl"):
        v, metric = _read_flat(r, tag)
        return np.array(v, dtype=np.float32), {"kind": "flat", "d": int(v.shape[1]), "ntotal": int(v.shape[0]), "metric": int(metric)}
    if tag != "IwFl":
        raise ValueError(f"unsupported faiss index type {tag!r}: only IVF*,Flat (IwFl) and Flat (IxF2) files hold raw vectors")
    d, ntotal, trained, metric = _read_header(r)
    nlist, nprobe = r.take("Q"), r.take("Q")
    qtag = r.fourcc()
    if qtag not in ("IxF2", "IxFI", "IxFl"):
        raise ValueError(f"unsupported coarse quantizer {qtag!r}")
    centroids, _ = _read_flat(r, qtag)
    dm_type = r.take("B")
    r.array(np.int64, r.take("Q"))
    if dm_type == 2:
        r.array(np.int64, 2 * r.take("Q"))
    ltag = r.fourcc()
    if ltag != "ilar":
        raise ValueError(f"unsupported inverted-list storage {ltag!r}")
    nl2, code_size = r.take("Q"), r.take("Q")
    if nl2 != nlist or code_size != 4 * d:
        raise ValueError(f"inverted lists: nlist {nl2} / code size {code_size} do not match an IVF{nlist},Flat index of dimension {d}")
    ftag = r.fourcc()
    sizes = np.zeros(nlist, dtype=np.int64)
    if ftag == "full":
        s = r.array(np.uint64, r.take("Q"))
        if s.shape[0] != nlist:
            raise ValueError("inverted lists: size table does not cover every list")
        sizes[:] = s.astype(np.int64)
    elif ftag == "sprs":
        s = r.array(np.uint64, r.take("Q")).astype(np.int64)
        sizes[s[0::2]] = s[1::2]
    else:
        raise ValueError(f"unknown inverted-list size encoding {ftag!r}")
    if int(sizes.sum()) != ntotal:
        raise ValueError(f"inverted lists hold {int(sizes.sum())} vectors, header says {ntotal}")
    out = np.empty((ntotal, d), dtype=np.float32)
    seen = np.zeros(ntotal, dtype=bool)
    list_of = np.empty(ntotal, dtype=np.int32)
    for li in np.nonzero(sizes)[0]:
        n = int(sizes[li])
        codes = r.array(np.float32, n * d).reshape(n, d)
        ids = r.array(np.int64, n)
        if ids.min() < 0 or ids.max() >= ntotal or seen[ids].any():
            raise ValueError("inverted lists: ids are not a permutation of 0 .. ntotal - 1 (index built with custom ids?)")
        out[ids] = codes
        seen[ids] = True
        list_of[ids] = li
    return out, {"kind": "ivf_flat", "d": int(d), "ntotal": int(ntotal), "nlist": int(nlist), "nprobe": int(nprobe), "metric": int(metric),
                 "trained": trained, "centroids": np.array(centroids, dtype=np.float32), "list_of": list_of}


def write_ivf_flat(path, vectors, nlist, centroids=None, nprobe=1, sparse=None):
    """Writes `vectors` [N, d] as a faiss IVF{nlist},Flat (L2) file with ids 0 .. N - 1: every vector goes to the list of its nearest
    centroid (centroids default to evenly spaced samples of the data - faiss would have trained k-means; the file layout is the same).
    Returns the list assignment.  Test infrastructure for read_index_vectors and a way to hand big_npy to a faiss-based tool."""
    x = np.ascontiguousarray(vectors, dtype="<f4")
    N, d = x.shape
    if centroids is None:
        centroids = x[np.linspace(0, N - 1, nlist).astype(np.int64)]
    c = np.ascontiguousarray(centroids, dtype="<f4")
    assert c.shape == (nlist, d)
    d2 = (x.astype(np.float64) ** 2).sum(1)[:, None] - 2.0 * x.astype(np.float64) @ c.astype(np.float64).T + (c.astype(np.float64) ** 2).sum(1)[None]
    assign = d2.argmin(1)

    def header(nt):
        return struct.pack("<iqqqBi", d, nt, 1 << 20, 1 << 20, 1, 1)
    out = [b"IwFl", header(N), struct.pack("<QQ", nlist, nprobe), b"IxF2", header(nlist), struct.pack("<Q", nlist * d), c.tobytes(),
           struct.pack("<B", 0), struct.pack("<Q", 0), b"ilar", struct.pack("<QQ", nlist, 4 * d)]
    sizes = np.bincount(assign, minlength=nlist).astype("<u8")
    if sparse is None:
        sparse = int((sizes > 0).sum()) <= nlist // 2
    if sparse:
        nz = np.nonzero(sizes)[0]
        pairs = np.empty(2 * nz.shape[0], dtype="<u8"); pairs[0::2] = nz; pairs[1::2] = sizes[nz]
        out += [b"sprs", struct.pack("<Q", pairs.shape[0]), pairs.tobytes()]
    else:
        out += [b"full", struct.pack("<Q", nlist), sizes.tobytes()]
    for li in range(nlist):
        ids = np.nonzero(assign == li)[0].astype("<i8")
        if ids.shape[0]:
            out += [x[ids].tobytes(), ids.tobytes()]
    with open(path, "wb") as f:
        f.write(b"".join(out))
    return assign
