#!/bin/bash
# builds the micro-benchmarks into tools/micro/bin (gfx950)
set -e
cd "$(dirname "$0")"
mkdir -p bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
for f in fp32_packed_rate fp16_packed_rate fp32_operand_forms fp64_op_rates fp64_outer_product mix_fp64_fp32; do
    [ -f $f.hip ] && /opt/rocm/bin/hipcc $FLAGS $f.hip -o bin/$f
done
# accumulators in VGPRs (no v_accvgpr_read per result register); MFMA_AGPR=1 builds the compiler's default form
MFMA_FORM="-mllvm -amdgpu-mfma-vgpr-form=1"
[ -n "$MFMA_AGPR" ] && MFMA_FORM=""
/opt/rocm/bin/hipcc $FLAGS $MFMA_FORM -fPIC -shared -fvisibility=hidden mfma_filter.hip -o bin/libmfma_filter.so
/opt/rocm/bin/hipcc $FLAGS $MFMA_FORM -fPIC -shared -fvisibility=hidden mfma_bf16_filter.hip -o bin/libmfma_bf16_filter.so
