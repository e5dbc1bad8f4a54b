#!/bin/bash
# Resource usage of every kernel in one .hip file: tools/kres.sh pixo_amd/csrc/jpeg_kernels.hip [extra flags]
f=$1; shift
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c "$(basename "$f")" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name|remark:.* Name:/ {name=$NF} / Name: /{n=$0; sub(/.*Name: /,"",n); sub(/ \[.*/,"",n); name=n}
       /TotalSGPRs:/{s=$0; sub(/.*TotalSGPRs: /,"",s); sub(/ .*/,"",s)} /    VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
       /ScratchSize/{sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)} /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ .*/,"",o)}
       /VGPRs Spill/{sp=$0; sub(/.*: /,"",sp); sub(/ .*/,"",sp)}
       /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); printf "%-70s sgpr %3s vgpr %3s scratch %s occ %s vspill %s lds %s\n", name, s, v, sc, o, sp, l}' | c++filt
