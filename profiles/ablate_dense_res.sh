#!/bin/bash
# Runs ON THE GPU BOX: ablation of k_dense3x3_res (diagnostic builds made in the build container into _abl/, WRONG results by
# construction): what do VGG-16's conv1_2 / conv2_1 cost (encode pre-pass + GEMM kernel) without the MFMAs / the fragment reads
# from LDS / the epilogue and its stores / the halo DMAs of the following tiles?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R   # _abl/libslfp_<V>.so: hipcc -DSLFP_RES_<V> on conv_dense.hip, linked with the other objects of cnns_slfp_quantization_amd/build/
cp cnns_slfp_quantization_amd/libslfp_hip.so _abl/libslfp_KEEP.so
for v in BASE NOMFMA NOLDS NOST NODMA NOSTDMA BASE; do
  cp _abl/libslfp_$v.so cnns_slfp_quantization_amd/libslfp_hip.so
  echo "== $v"; DLT_LAYERS=2 DLT_VARIANTS=1 python profiles/dense_layer_time.py 2>/dev/null
done
cp _abl/libslfp_KEEP.so cnns_slfp_quantization_amd/libslfp_hip.so
