#!/bin/bash
# usage: scripts/kernel_regs.sh [pattern] ["extra flags"]  - registers, spills and scratch of the strict flavour's kernels whose
# mangled name matches the pattern (default: the two-samples-per-lane kernels), from the compiler's resource remarks
cd "$(dirname "$0")/../raytracer_amd/csrc"
PAT=${1:-pairkernel_strict}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -ffp-contract=off -DRT_FLAVOUR_STRICT $2 \
  -Rpass-analysis=kernel-resource-usage -c rt_kernels.hip -o /tmp/kernel_regs.o 2>&1 | grep -A12 "Function Name: .*$PAT" | grep "Function Name\|VGPRs:\|Spill\|ScratchSize\|Occupancy" | \
  sed 's/\[-Rpass.*//; s/.*remark: *//' | paste - - - - - - | sed 's/Function Name: _Z[0-9]*//; s/EEv8DevScene.*HIP_vector_typeIfLj4EES5_//'
