#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/pmc_kernels.sh att attention_dma_kernel att64 2>&1 | tail -40
bash tools/pmc_kernels.sh attrel attention_dma_kernel att96rel 2>&1 | tail -40
bash tools/pmc_kernels.sh cbr16 cbr2_small cbr16 2>&1 | tail -40
bash tools/pmc_kernels.sh cbr32 cbr2_small cbr32 2>&1 | tail -40
