"""FeatureExtractor: f0 front-end dispatch and coarse-pitch quantisation (mirror of reference pitch_extraction.py:13-303).

The RMVPE front-ends ("rmvpe", "rmvpe+") and the CREPE front-ends ("crepe", "mangio-crepe" and their -tiny forms: the network of
third-party torchcrepe re-implemented as HIP kernels, lib/crepe.py) run on this build's HIP path; pm / harvest / dio are third-party
CPU libraries in the reference (parselmouth, pyworld) and stay out of scope - their dictionary slots exist so that a caller can
plug a replacement in, exactly as with the reference's `f0_method_dict`.
"""
import os
from functools import partial

import numpy as np

from .lib import BASE_MODELS_DIR
from .lib.audio import autotune_f0, hz_to_mel, pad_audio, get_merge_func
from .lib.rmvpe import RMVPE
from .lib.utils import gc_collect


def _unsupported(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"f0 method '{name}' relies on a third-party CPU library that is out of scope here; "
                                  "use 'rmvpe' / 'rmvpe+' or assign a callable to f0_method_dict['%s']" % name)
    return fn


class FeatureExtractor:
    def __init__(self, tgt_sr, config, onnx=False):
        self.x_pad, self.x_query, self.x_center, self.x_max, self.is_half = (
            config.x_pad, config.x_query, config.x_center, config.x_max, config.is_half)
        self.sr = 16000          # hubert / rmvpe input rate
        self.window = 160        # 10 ms hop
        self.f0_bins = 256
        self.t_pad = self.sr * self.x_pad
        self.t_pad_tgt = tgt_sr * self.x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sr * self.x_query
        self.t_center = self.sr * self.x_center
        self.t_max = self.sr * self.x_max
        self.device = config.device
        self.onnx = onnx
        self.f0_method_dict = {
            "pm": _unsupported("pm"), "harvest": _unsupported("harvest"), "dio": _unsupported("dio"),
            "rmvpe": self.get_rmvpe, "rmvpe_onnx": self.get_rmvpe, "rmvpe+": self.get_pitch_dependant_rmvpe,
            "crepe": self.get_f0_official_crepe_computation,
            # (the reference binds model='model' for the -tiny slots, pitch_extraction.py:42,:44 - a name torchcrepe rejects; 'tiny' is what is
            # meant.  Through get_f0 the bound keyword never applies: its parameter dict carries model="full", as upstream, :268-271)
            "crepe-tiny": partial(self.get_f0_official_crepe_computation, model="tiny"),
            "mangio-crepe": self.get_f0_crepe_computation,
            "mangio-crepe-tiny": partial(self.get_f0_crepe_computation, model="tiny"),
        }
        self.model_crepe = {}        # capacity ("full" / "tiny") -> lib.crepe.Crepe; filled lazily from models/torchcrepe/<capacity>.pth

    def __del__(self):
        if hasattr(self, "model_rmvpe"):
            del self.model_rmvpe
            try:
                gc_collect()
            except Exception:   # noqa: BLE001 - interpreter teardown
                pass

    def load_index(self, file_index):
        """(index, big_npy) for feature retrieval (reference :52-73).  Accepts the reference's preloaded tuple, "" (no index),
        a `.npy` file holding big_npy [N, D] (what `train_index` saves next to the faiss file as total_fea.npy), or a faiss
        `.index` file (IVF*,Flat and Flat files are read natively, lib/faiss_io.py; other types need faiss).  The search object is device
        resident (lib/feature_index.py): an IVF file is searched the way faiss searches it - the file's nprobe nearest cells, the nearest vector
        inside them (reference: index built with nprobe 1, custom_nodes/rvc_nodes.py:500-554) - a bare big_npy exactly.
        Errors are printed and turn into "no index", as in the reference."""
        index = big_npy = None
        try:
            if isinstance(file_index, tuple):
                index, big_npy = file_index
                if index is not None and not hasattr(index, "search_device") and big_npy is not None:
                    index = self._device_index(big_npy)          # a faiss object from the caller: keep its vectors, search on the GPU
            elif file_index == "" or file_index is None:
                pass
            elif str(file_index).endswith(".npy"):
                big_npy = np.load(file_index).astype(np.float32)
                index = self._device_index(big_npy)
            else:
                # the `added_IVF*_Flat_*.index` files RVC users have: the stored vectors in id order are all the conversion needs (what
                # faiss.read_index + reconstruct_n(0, ntotal) returns); read natively - faiss is optional (lib/faiss_io.py)
                from .lib.faiss_io import read_index_vectors   # noqa: PLC0415
                ivf = None
                try:
                    big_npy, info = read_index_vectors(file_index)
                    if info.get("kind") == "ivf_flat" and info.get("metric", 1) == 1:
                        nprobe, nlist = max(1, int(info["nprobe"])), int(info["nlist"])
                        if 16 < nprobe < nlist:
                            # the device probe handles up to 16 cells per query; a file saved with more (none of the reference's: train_index sets 1)
                            # is searched exactly instead of being dropped - the exact answer is what faiss converges to as nprobe grows
                            print(f"{file_index}: nprobe {nprobe} of {nlist} cells exceeds the device probe (16): using the exact search")
                        else:
                            # (a frame whose probed cells are all empty gets faiss's label -1 and, through the reference's 1 / score^2 weights, a NaN feature
                            # frame - reference behaviour, kept and documented in INTEGRATION.md; every index train_index writes has nprobe 1, so a message
                            # here would greet every normal user with something they cannot act on)
                            ivf = (info["centroids"], info["list_of"], nprobe)
                except ValueError:
                    import faiss   # noqa: PLC0415 - other index types (PQ ...): only faiss can decode them
                    fidx = faiss.read_index(file_index)
                    big_npy = fidx.reconstruct_n(0, fidx.ntotal)
                index = self._device_index(big_npy, ivf)
        except Exception as e:   # noqa: BLE001 - reference behaviour
            print(f"Could not open Faiss index file for reading. {e}")
            index = big_npy = None
        return index, big_npy

    def _device_index(self, big_npy, ivf=None):
        from .lib.feature_index import DeviceIndex   # noqa: PLC0415
        dev = self.device if str(self.device).startswith("cuda") else "cuda:0"
        return DeviceIndex(big_npy, device=dev, ivf=ivf)

    def _rmvpe(self):
        if not hasattr(self, "model_rmvpe"):
            self.model_rmvpe = RMVPE(os.path.join(BASE_MODELS_DIR, "rmvpe.pt"), is_half=self.is_half, device=self.device, onnx=False)
        return self.model_rmvpe

    def get_rmvpe(self, x, *args, **kwargs):
        return self._rmvpe().infer_from_audio(x, thred=0.03)

    def get_pitch_dependant_rmvpe(self, x, f0_min=0, f0_max=40000, *args, **kwargs):
        return self._rmvpe().infer_from_audio_with_pitch(x, thred=0.03, f0_min=f0_min, f0_max=f0_max)

    def _crepe(self, model):
        if model not in self.model_crepe:
            from .lib.crepe import Crepe   # noqa: PLC0415
            dev = self.device if str(self.device).startswith("cuda") else "cuda:0"
            self.model_crepe[model] = Crepe(None, model, dev)
        return self.model_crepe[model]

    def get_f0_crepe_computation(self, x, f0_min, f0_max, *args, **kwargs):
        """"mangio-crepe" (reference pitch_extraction.py:76-120): quantile-normalised audio, torchcrepe.predict at crepe_hop_length,
        frames below 1 mHz dropped, linear interpolation onto x.shape[0] // hop frames."""
        from .lib import crepe as tc   # noqa: PLC0415
        x = np.asarray(x.cpu() if hasattr(x, "cpu") else x).astype(np.float32)
        x /= np.quantile(np.abs(x), 0.999)
        hop_length = kwargs.get("crepe_hop_length", 160)
        model = kwargs.get("model", "full")
        pitch = tc.predict(x[None], self.sr, hop_length, f0_min, f0_max, model, batch_size=hop_length * 2, device=self.device, pad=True,
                           crepe=self._crepe(model))
        p_len = x.shape[0] // hop_length
        source = np.array(pitch.squeeze(0).cpu().float().numpy())
        source[source < 0.001] = np.nan
        target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
        return np.nan_to_num(target)

    def get_f0_official_crepe_computation(self, x, f0_min, f0_max, *args, **kwargs):
        """"crepe" (reference pitch_extraction.py:122-150): torchcrepe.predict at the 10 ms hop with periodicity, median-3 on the
        periodicity, mean-3 on the pitch, frames with periodicity < 0.1 unvoiced."""
        from .lib import crepe as tc   # noqa: PLC0415
        x = np.asarray(x.cpu() if hasattr(x, "cpu") else x)
        model = kwargs.get("model", "full")
        f0, pd = tc.predict(np.copy(x).astype(np.float32)[None], self.sr, self.window, f0_min, f0_max, model, batch_size=512, device=self.device,
                            return_periodicity=True, crepe=self._crepe(model))
        pd = tc.filter_median(pd[0].numpy(), 3)
        f0 = tc.filter_mean(f0[0].numpy(), 3)
        f0[pd < 0.1] = 0
        return f0

    def get_f0_hybrid_computation(self, methods_list, merge_type, x, f0_min, f0_max, filter_radius, crepe_hop_length, time_step, **kwargs):
        """Median/mean/... merge of several f0 tracks (reference pitch_extraction.py:205-248), run sequentially."""
        params = {"x": x, "f0_min": f0_min, "f0_max": f0_max, "time_step": time_step, "filter_radius": filter_radius,
                  "crepe_hop_length": crepe_hop_length, "model": "full"}
        x = x.astype(np.float32)
        x /= np.quantile(np.abs(x), 0.999)
        stack = []
        for method in methods_list:
            if method not in self.f0_method_dict:
                raise Exception(f"Method {method} not found.")
            stack.append(self.f0_method_dict[method](**params))
        stack = pad_audio(*stack)
        return get_merge_func(merge_type)(stack, axis=0)

    def get_f0(self, x, f0_up_key, f0_method, merge_type="median", filter_radius=3, crepe_hop_length=160, f0_autotune=False,
               rmvpe_onnx=False, inp_f0=None, f0_min=50, f0_max=1100, **kwargs):
        time_step = self.window / self.sr * 1000
        f0_mel_min = hz_to_mel(f0_min)
        f0_mel_max = hz_to_mel(f0_max)
        params = {"x": x, "f0_up_key": f0_up_key, "f0_min": f0_min, "f0_max": f0_max, "time_step": time_step,
                  "filter_radius": filter_radius, "crepe_hop_length": crepe_hop_length, "model": "full", "onnx": rmvpe_onnx}
        if hasattr(f0_method, "pop") and len(f0_method) == 1:
            f0_method = f0_method.pop()
        if isinstance(f0_method, list):
            f0 = self.get_f0_hybrid_computation(f0_method, merge_type, **params)
        else:
            f0 = self.f0_method_dict[f0_method](**params)
        if f0_autotune:
            f0 = autotune_f0(f0)
        f0 *= pow(2, f0_up_key / 12)
        tf0 = self.sr // self.window
        if inp_f0 is not None:   # f0 curve supplied by the user: splice it in after the left pad
            delta_t = np.round((inp_f0[:, 0].max() - inp_f0[:, 0].min()) * tf0 + 1).astype("int16")
            replace_f0 = np.interp(list(range(delta_t)), inp_f0[:, 0] * 100, inp_f0[:, 1])
            shape = f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)].shape[0]
            f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)] = replace_f0[:shape]
        f0_mel = hz_to_mel(f0)
        f0_mel = (f0_mel - f0_mel_min) * (self.f0_bins - 2) / (f0_mel_max - f0_mel_min) + 1
        f0_mel = np.clip(f0_mel, a_min=1, a_max=self.f0_bins - 1)
        f0_coarse = np.rint(f0_mel).astype(np.int16)
        return f0_coarse, f0
