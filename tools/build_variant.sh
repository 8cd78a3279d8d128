#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] (always -DRVC_EXPERIMENTS: env knobs read, rvc_debug_* exported) : builds comfy-rvc_amd/csrc/variants/librvc_hip_NAME.so for A/B kernel timing
set -e
cd "$(dirname "$0")/../comfy-rvc_amd/csrc"
name=$1; shift
mkdir -p variants/obj_$name
for f in conv_mfma conv_x3 conv_x3p conv_x3q conv_rbh conv_rb3 conv_x3s split2d conv_cbr2 attention attention_dma attention_dma_rel ops model_synth model_hubert model_rmvpe model_crepe model_mdx23 index rvc_api; do
  extra=""; [ $f = attention_dma ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DRVC_EXPERIMENTS $extra "$@" -c $f.hip -o variants/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/librvc_hip_$name.so variants/obj_$name/*.o
echo built variants/librvc_hip_$name.so
