#!/bin/bash
# build a variant library: ab_build.sh <name> <file.hip[,file2.hip...]> <flags...>; result deeprecommendation_amd/libncf_hip_<name>.so
set -e
NAME=$1; SRCS=",$2,"; shift 2
cd /root/repo/deeprecommendation_amd/csrc
mkdir -p build/$NAME
for f in abi gather mlp_fused linear spmm attn attn_split attn_cand attn_tail mlp_bf16 mlp_bf16_ws8 backward exchange dense_csr probe; do
  if [[ "$SRCS" == *",$f.hip,"* ]]; then
    EXTRA=""; [[ "$f" == mlp_bf16* ]] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA "$@" -c $f.hip -o build/$NAME/$f.hip.o
    OBJS="$OBJS build/$NAME/$f.hip.o"
  else
    OBJS="$OBJS build/$f.hip.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libncf_hip_$NAME.so $OBJS
echo built ../libncf_hip_$NAME.so
