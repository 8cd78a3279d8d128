#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version, round 5: the review's optional 2-MFMA-per-product mode, MEASURED on a variant build of the persistent ResBlock kernel - group lo_w * hi_x left out,
#  comfy-rvc_amd/csrc/variants/librvc_hip_two.so = -DRVC_X3Q_TWO_MFMA - accuracy on the three full-size goldens and time; profiles/r5_two_mfma.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
V=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_two.so
echo "== full-size goldens on the 2-MFMA variant (the gates are expected to FAIL; the recorded maxima are the result)"
rm -f gpurun_out/fullsize_parity.json
RVC_HIP_LIB=$V timeout 600 python -m pytest tests/test_hip_pipeline.py -m gpu -q -k "c3_30s or c4_30s_48k or 45s_clip" 2>&1 | tail -4
python - <<'PY'
import json
p = json.load(open("gpurun_out/fullsize_parity.json"))
for k, v in p.items():
    print(k, {a: v[a] for a in ("max", "p9999", "mean", "within", "f0_within_1e-3", "coarse_equal", "voicing_equal") if a in v})
PY
echo "== the same three tests on the product build"
rm -f gpurun_out/fullsize_parity.json
timeout 600 python -m pytest tests/test_hip_pipeline.py -m gpu -q -k "c3_30s or c4_30s_48k or 45s_clip" 2>&1 | tail -2
python - <<'PY'
import json
p = json.load(open("gpurun_out/fullsize_parity.json"))
for k, v in p.items():
    print(k, {a: v[a] for a in ("max", "p9999", "mean", "within") if a in v})
PY
echo "== time: product, variant, product, variant"
bash tools/ab.sh "A=0" "RVC_HIP_LIB=$V" "A=1" "RVC_HIP_LIB=$V"
