"""Device-resident feature retrieval index.

The reference keeps a faiss IVF-Flat index over the training features `big_npy [N, D]` (built by
custom_nodes/rvc_nodes.py:500-554, loaded by pitch_extraction.py:52-73) and calls `index.search(npy, k=1)` on the HuBERT
frames of every segment (vc_infer_pipeline.py:60-75).  `DeviceIndex` offers that call surface - `search`, `ntotal`,
`reconstruct_n` - over the same `big_npy` on the GPU (rvc_index_*, csrc/index.hip), with the semantics of the object it stands for:

* built from a faiss `IVF*,Flat` file (`ivf=(centroids, list_of, nprobe)`, what lib/faiss_io.py reads out of the file): faiss's
  IndexIVFFlat search - the `nprobe` centroids nearest to the query pick the cells, the nearest vector INSIDE those cells is the answer
  (nprobe 1 as train_index sets it: often not the global nearest neighbour), empty probed cells give label -1 / distance FLT_MAX;
* built from a bare `big_npy` (`.npy` file, the reference's preloaded tuple): no cell structure exists, the search is the exact one.

faiss itself is not available offline: both are pinned against restatements of faiss's published algorithm
(oracle/pipeline.py::index_search_ivf / index_search), not against faiss.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class DeviceIndex:
    def __init__(self, big_npy, device="cuda:0", ivf=None):
        big_npy = np.ascontiguousarray(big_npy, dtype=np.float32)
        assert big_npy.ndim == 2, "big_npy must be [N, D]"
        self.device = torch.device(device)
        self.ntotal, self.d = int(big_npy.shape[0]), int(big_npy.shape[1])
        self._big = big_npy
        self._h = C.c_void_p()
        self.nprobe = 0                                    # 0: exact search
        ctx = _lib.get_ctx(self.device.index or 0)
        with torch.cuda.device(self.device):
            if ivf is None:
                _lib.check(_lib.lib.rvc_index_create(ctx, _lib.ptr(big_npy), self.ntotal, self.d, C.byref(self._h)))
            else:
                centroids, list_of, nprobe = ivf
                centroids = np.ascontiguousarray(centroids, dtype=np.float32)
                list_of = np.ascontiguousarray(list_of, dtype=np.int32)
                assert centroids.ndim == 2 and centroids.shape[1] == self.d and list_of.shape == (self.ntotal,)
                _lib.check(_lib.lib.rvc_index_create_ivf(ctx, _lib.ptr(big_npy), self.ntotal, self.d, _lib.ptr(centroids), int(centroids.shape[0]),
                                                         _lib.ptr(list_of), int(nprobe), C.byref(self._h)))
                self.nprobe = int(_lib.lib.rvc_index_nprobe(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib.rvc_index_destroy(h)
            except Exception:   # noqa: BLE001 - interpreter teardown
                pass
            self._h = C.c_void_p()

    def reconstruct_n(self, i0, n):
        return self._big[i0: i0 + n]

    # ---- device entry points (features channel-major [D][T] on the device)
    def search_device(self, feats_cm, want_score=False):
        T = int(feats_cm.shape[1])
        assert feats_cm.shape[0] == self.d and feats_cm.is_contiguous() and feats_cm.dtype == torch.float32
        idx = torch.empty(T, dtype=torch.int64, device=feats_cm.device)
        score = torch.empty(T, dtype=torch.float32, device=feats_cm.device) if want_score else None
        with torch.cuda.device(feats_cm.device):
            _lib.check(_lib.lib.rvc_index_search(self._h, _lib.current_stream(), _lib.ptr(feats_cm), T, _lib.ptr(idx), _lib.ptr(score)))
        return idx, score

    def blend_device(self, feats_cm, index_rate):
        """index_rate * big_npy[nearest] + (1 - index_rate) * feats  (reference :71-74), channel-major in and out."""
        idx, _ = self.search_device(feats_cm)
        out = torch.empty_like(feats_cm)
        with torch.cuda.device(feats_cm.device):
            _lib.check(_lib.lib.rvc_index_blend(self._h, _lib.current_stream(), _lib.ptr(feats_cm), _lib.ptr(idx), int(feats_cm.shape[1]),
                                                float(index_rate), _lib.ptr(out)))
        return out

    # ---- faiss call surface (host arrays), used by the generic VC.vc path
    def search(self, npy, k=1):
        if k != 1:
            raise NotImplementedError("the reference only ever asks for k = 1 (vc_infer_pipeline.py:65)")
        f = torch.from_numpy(np.ascontiguousarray(npy, dtype=np.float32)).to(self.device).t().contiguous()
        idx, score = self.search_device(f, want_score=True)
        return score.cpu().numpy()[:, None], idx.cpu().numpy()[:, None]
