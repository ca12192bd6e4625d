#!/bin/bash
# Register / scratch use of one kernel of one source file: tools/kres.sh <file.hip> <kernel-name-substring> [flags]
cd "$(dirname "$0")/../graphblast_amd/csrc"
src=$1; k=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. "$@" -c $src -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "Function Name: .*$k" | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
