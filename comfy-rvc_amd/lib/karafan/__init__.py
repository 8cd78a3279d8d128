"""Mirror of the part of reference lib/karafan that sits on the UVR chain of BASELINE config C5: the MDX23C network (tfc_tdf.py) and the
chunked overlap-add driver demix_mdxv3 (inference.py:32-74).  The rest of karafan (multi-model ensembling, filters, file IO) is out of scope."""
from . import inference, tfc_tdf   # noqa: F401
