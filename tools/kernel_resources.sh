#!/bin/bash
# Registers / LDS / scratch / occupancy of every kernel of the library, from the compiler's own report
# (no GPU needed): tools/kernel_resources.sh [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" \
  -o /dev/null $R/nim-snappy_amd/csrc/snappy_hip.hip 2>&1 | grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ \[-Rpass.*$//' |
  awk '/^Function Name/ {if (name) print name ": " acc; name=$3; acc=""; next}
       /^VGPRs:|^SGPRs Spill|^VGPRs Spill|^ScratchSize|^Occupancy|^LDS Size|^TotalSGPRs/ {acc = acc $0 "; "}
       END {print name ": " acc}' | sed -e 's/_ZN10snappy_hip//' | c++filt -n 2>/dev/null
