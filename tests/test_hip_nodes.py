"""GPU: the ComfyUI node graph end to end from model FILES (safetensors HuBERT, .pth voice model, rmvpe.pt, big_npy index), i.e. the
drop-in surface of reference custom_nodes/rvc_nodes.py:44-206 as a graph executor would call it."""
import json
import os

import numpy as np
import pytest
import torch

from comfy_rvc_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_node_graph_from_files(tmp_path, monkeypatch):
    from safetensors.torch import save_file
    import comfy_rvc_amd.lib as lib
    import comfy_rvc_amd.pitch_extraction as pe
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N
    from comfy_rvc_amd.vc_infer_pipeline import vc_single
    models = tmp_path / "models"
    (models / "RVC" / ".index").mkdir(parents=True)
    # files in the layouts the reference downloads: content-vec-best.safetensors (+ HubertConfig JSON in the metadata), RVC/*.pth, rmvpe.pt
    as_t = lambda sd: {k: torch.as_tensor(np.ascontiguousarray(v)).clone() for k, v in sd.items()}   # noqa: E731
    save_file(as_t(S.hubert_state_dict(0)), str(models / "content-vec-best.safetensors"),
              metadata={"config": json.dumps(S.HUBERT_CONFIG)})
    cpt = S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0)
    cpt["weight"] = as_t(cpt["weight"])                            # real .pth files hold torch tensors (torch.load weights_only default)
    torch.save(cpt, str(models / "RVC" / "voice.pth"))
    torch.save(as_t(S.rmvpe_state_dict(0)), str(models / "rmvpe.pt"))
    for mod in (lib, pe, N):
        monkeypatch.setattr(mod, "BASE_MODELS_DIR", str(models), raising=False)
    audio = S.synth_audio(1.5, seed=40)
    (params,) = N.LoadPitchExtractionParams().load_params(f0_method="rmvpe", f0_autotune=False, index_rate=0.0, resample_sr=0, rms_mix_rate=0.25,
                                                          protect=0.25, crepe_hop_length=160)
    (hub_thunk,) = N.LoadHubertModel().load_model("content-vec-best.safetensors")
    model_thunk, name = N.LoadRVCModelNode().load_model("RVC/voice.pth")
    assert name == "voice" and callable(hub_thunk) and callable(model_thunk)
    assert "RVC/voice.pth" in N.LoadRVCModelNode.INPUT_TYPES()["required"]["model"][0]
    audio_in = N.to_audio_dict(audio, 16000)                       # AUDIO socket layout [1, N, C]
    torch.manual_seed(7)
    res = N.RVCNode().convert(audio_in, model_thunk, hub_thunk, params, f0_up_key=2, format="flac", use_cache=False)
    vhs, aud = res["result"]
    from comfy_rvc_amd.lib.audio import bytes_to_audio
    blob = vhs()                                                   # VHS_AUDIO: thunk -> encoded WAV bytes (reference rvc_nodes.py:206)
    assert isinstance(blob, bytes) and blob[:4] == b"RIFF"
    wav_f, sr = bytes_to_audio(blob)
    wav = aud["waveform"][0, :, 0].numpy()
    assert np.array_equal(np.round(wav_f * 32768).astype(np.int16), wav)
    assert sr == 40000 and wav.dtype == np.int16 and aud["sample_rate"] == 40000 and tuple(aud["waveform"].shape) == (1, wav.shape[0], 1)
    assert res["ui"]["preview"][0]["filename"].endswith(".flac")
    # the thunks are memoised per (path, mtime): same objects on the second call, and the direct API gives the same audio
    assert hub_thunk() is hub_thunk() and model_thunk() is model_thunk()
    vm = model_thunk()
    torch.manual_seed(7)
    direct = vc_single(hubert_model=hub_thunk(), input_audio=(audio, 16000), f0_up_key=2, **vm, **params)
    assert direct is not None and np.array_equal(direct[0], wav)
    assert np.abs(wav.astype(np.int32)).max() > 1000              # not silence
    # with a retrieval index file (big_npy): listed by the node, preloaded by get_vc, applied with index_rate
    rng = np.random.default_rng(0)
    feats = hub_thunk().extract_features(torch.from_numpy(S.synth_audio(2.0, seed=41))[None], version="v2")[0].cpu().numpy()
    np.save(str(models / "RVC" / ".index" / "voice.npy"), (feats[rng.integers(0, feats.shape[0], 500)] + 0.05 * rng.standard_normal((500, 768))).astype(np.float32))
    assert "RVC/.index/voice.npy" in N.LoadRVCModelNode.INPUT_TYPES()["optional"]["index"][0]
    model_idx, _ = N.LoadRVCModelNode().load_model("RVC/voice.pth", "RVC/.index/voice.npy")
    vmi = model_idx()
    assert isinstance(vmi["file_index"], tuple) and vmi["file_index"][0].ntotal == 500
    params_i = dict(params, index_rate=0.8)
    torch.manual_seed(7)
    res_i = N.RVCNode().convert(audio_in, model_idx, hub_thunk, params_i, f0_up_key=2, use_cache=False)
    wav_i = res_i["result"][1]["waveform"][0, :, 0].numpy()
    assert wav_i.shape == wav.shape and np.abs(wav_i.astype(np.int32) - wav.astype(np.int32)).max() > 100     # the blend changes the audio


def test_node_graph_bytes_in_bytes_out(tmp_path, monkeypatch):
    """VHS_AUDIO on both sockets: a thunk returning WAV bytes goes in (reference lib/audio.py:115-124), a thunk returning WAV bytes
    comes out, and feeding the output thunk into a second RVCNode works (chained conversion, resampled 40 k -> 16 k on the way in)."""
    import comfy_rvc_amd.lib as lib
    import comfy_rvc_amd.pitch_extraction as pe
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N
    from comfy_rvc_amd.lib.audio import audio_to_bytes, bytes_to_audio
    from comfy_rvc_amd.lib.infer_pack.loaders import HubertModelWithFinalProj
    from comfy_rvc_amd.vc_infer_pipeline import get_vc
    models = tmp_path / "models"
    models.mkdir()
    as_t = lambda sd: {k: torch.as_tensor(np.ascontiguousarray(v)).clone() for k, v in sd.items()}   # noqa: E731
    torch.save(as_t(S.rmvpe_state_dict(0)), str(models / "rmvpe.pt"))
    for mod in (lib, pe, N):
        monkeypatch.setattr(mod, "BASE_MODELS_DIR", str(models), raising=False)
    hub = HubertModelWithFinalProj(S.hubert_state_dict(0), S.HUBERT_CONFIG)
    vcd = get_vc(S.synth_checkpoint(S.CONFIG_40K_V2, "v2", 0))
    (params,) = N.LoadPitchExtractionParams().load_params(f0_method="rmvpe", f0_autotune=False, index_rate=0.0, resample_sr=0, rms_mix_rate=0.25,
                                                          protect=0.25, crepe_hop_length=160)
    audio = S.synth_audio(1.2, seed=42)
    blob_in = audio_to_bytes(audio, 16000)                          # float32 WAV (|x| <= 1)
    torch.manual_seed(3)
    a = N.RVCNode().convert(lambda: blob_in, lambda: vcd, lambda: hub, params, f0_up_key=0, use_cache=False)
    torch.manual_seed(3)
    b = N.RVCNode().convert(N.to_audio_dict(audio, 16000), lambda: vcd, lambda: hub, params, f0_up_key=0, use_cache=False)
    blob_a, blob_b = a["result"][0](), b["result"][0]()
    assert isinstance(blob_a, bytes) and blob_a == blob_b           # bytes in == AUDIO dict in
    wav, sr = bytes_to_audio(blob_a)
    assert sr == 40000 and wav.ndim == 1 and np.abs(wav).max() > 0.5
    c = N.RVCNode().convert(a["result"][0], lambda: vcd, lambda: hub, params, f0_up_key=-2, use_cache=False)   # chained through the thunk
    wav2, sr2 = bytes_to_audio(c["result"][0]())
    assert sr2 == 40000 and abs(wav2.shape[0] - wav.shape[0]) <= 1200


def test_package_exports_node_mappings():
    """ComfyUI imports the pack directory and reads NODE_CLASS_MAPPINGS / NODE_DISPLAY_NAME_MAPPINGS off the module (reference __init__.py:12-31)."""
    import comfy_rvc_amd
    from comfy_rvc_amd.custom_nodes import rvc_nodes as N
    assert hasattr(comfy_rvc_amd, "NODE_CLASS_MAPPINGS") and all(comfy_rvc_amd.NODE_CLASS_MAPPINGS[k] is v for k, v in N.NODE_CLASS_MAPPINGS.items())
    assert set(comfy_rvc_amd.NODE_DISPLAY_NAME_MAPPINGS) == {"LoadRVCModelNode", "RVCNode", "LoadHubertModel", "LoadPitchExtractionParams", "UVR5Node"}
