#!/bin/bash
# development aid: ISA + register report of wstream_kernel alone (seconds instead of the 2.5 minutes of the whole gemm.hip)
cd "$(dirname "$0")/../../spider_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=fast -mllvm -pragma-unroll-threshold=200000 -DSPIDER_F16 -DSPIDER_WS_DEV \
  -S --cuda-device-only gemm.hip -o /tmp/ws_dev.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|ScratchSize|SGPRs:" 
