"""Runtime constants of the inference path (mirror of reference config.py:87-168, constants :124-141).

The reference picks its segmentation constants from the detected device: fp16 GPUs get (x_pad, x_query, x_center,
x_max) = (3, 10, 60, 64), fp32/CPU gets (1, 6, 38, 41), <=4 GB GPUs (1, 5, 30, 32).  This build computes in fp32 and
has 288 GB of HBM per GPU, so the default is the reference's fp32 set - the same constants its CPU path (the parity
baseline) uses; `Config(x_pad=..., ...)` overrides them.
"""


class Config:
    def __init__(self, device="cuda:0", is_half=False, x_pad=1, x_query=6, x_center=38, x_max=41):
        self.device = device
        self.is_half = bool(is_half)
        self.x_pad, self.x_query, self.x_center, self.x_max = x_pad, x_query, x_center, x_max
        self.n_cpu = 0
        self.gpu_name = None
        self.gpu_mem = None

    def device_config(self):
        return self.x_pad, self.x_query, self.x_center, self.x_max


config = Config()
