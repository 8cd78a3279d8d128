"""gc_collect (mirror of reference lib/utils.py:59)."""
import gc

import torch


def gc_collect():
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
