#!/bin/bash
# A/B builds (no GPU needed): tools/build_ab.sh <name> <kind 0|1|2|all> [extra -D flags]  ->  build/ab/<name>.so
# One env kind, default layout only (seconds to a minute to compile); `all` = the product's flags.  build/ab/ travels with gpurun.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; kind=$2; shift 2
mkdir -p $ROOT/build/ab
ONLY="-DQR_ONLY_KIND=$kind -DQR_ONLY_LAYOUT=0 -Wno-unused-value -Wno-unused-const-variable"; [ "$kind" = all ] && ONLY=""
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$ROOT/include -ffp-contract=fast -fno-slp-vectorize \
  -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $ONLY "$@" \
  -o $ROOT/build/ab/$name.so $ROOT/gym_rotor_amd/csrc/quadrotor_kernels.hip && ls -la $ROOT/build/ab/$name.so
