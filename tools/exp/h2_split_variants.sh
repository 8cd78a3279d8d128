for v in s1 s2 s3; do echo "=== split2h variant $v"; RVC_HIP_LIB=$PWD/comfy-rvc_amd/csrc/variants/librvc_hip_$v.so python tools/exp/h2_c1_image.py 2>&1 | grep "^C"; done
