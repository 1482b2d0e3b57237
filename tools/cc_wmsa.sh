#!/bin/bash
# compile csrc/wmsa_block.hip alone and print its register / spill report (cwd-independent)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/t
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -Rpass-analysis=kernel-resource-usage \
  -c "$R/small-object-detection-transformers_amd/csrc/wmsa_block.hip" -o /tmp/t/w.o 2>&1 | grep -E "error|Function Name|VGPRs Spill|ScratchSize" | tail -${1:-12}
