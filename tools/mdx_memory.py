"""Device memory of the MDX23C network (shipped recipe) with 1 and with 3 chunk streams: free-memory difference around a 30 s stereo demix."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from comfy_rvc_amd import synthetic as S
from comfy_rvc_amd.custom_nodes.uvr import MDX23C_CONFIG as cfg
from comfy_rvc_amd.lib.karafan.inference import demix_mdxv3
from comfy_rvc_amd.lib.karafan.tfc_tdf import TFC_TDF_net
free0 = torch.cuda.mem_get_info()[0]
net = TFC_TDF_net(cfg); net.load_state_dict(S.mdx23c_state_dict(cfg, 0))
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
mix = np.stack([S.synth_audio(30.0, seed=100, sr=44100), S.synth_audio(30.0, seed=300, sr=44100)]).astype(np.float32)
out = {}
for k in (1, 3):
    net.set_streams(k)
    demix_mdxv3(mix, net, net.device, cfg, 8); torch.cuda.synchronize(); torch.cuda.empty_cache()
    out[k] = torch.cuda.mem_get_info()[0]
print(f"weights + images {(free0 - free1) / 2**30:.2f} GiB; after a demix with 1 stream {(free1 - out[1]) / 2**30:.2f} GiB more, with 3 streams {(free1 - out[3]) / 2**30:.2f} GiB more "
      f"(two extra chunk streams: {(out[1] - out[3]) / 2**30:.2f} GiB)")
