// The text encoder's instantiation of attention_dma_kernel (head dimension 96, windowed relative positions; reference attentions.py:230-267).
// A translation unit of its own: attention_dma.hip is built with -amdgpu-mfma-vgpr-form, whose rewrite pass crashes on this instantiation.
#include "attention_dma_kernel.h"

namespace rvc {

#ifdef RVC_CONV_TIMING
void attention_dma_rel_timing_read(unsigned long long* out8, bool reset) { attd_timing_read_tu(out8, reset); }
#endif
void attention_dma_rel_launch(const AttnDmaArgs& a, int heads, hipStream_t s) { launch_att_dma<96, 4, 1, true>(a, heads, s); }

}  // namespace rvc
