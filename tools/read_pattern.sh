#!/bin/bash
# read-only access-pattern calibration over 1 GiB (beyond the 256 MiB Infinity Cache): contiguous region per workgroup (3) vs the
# workgroups' 4 KiB chunks interleaved (4), per grid size; 2 = grid-stride loop 8 in flight; 1 = one float4 per thread copy
for grid in 256 688 1376 2048 8192; do for v in 3 4; do
ZGML_COPY_VARIANT=$((v + grid * 256)) python3 -c "
import sys; sys.path.insert(0,'.')
from zgml_amd import Backend
be=Backend(0)
cp=be._lib.zgml_hip_copy_bench(be.ctx, 1<<30, 3, 20)
print('grid', $grid, 'variant', $v, 'us', round(cp,2), 'GB/s read', round((1<<30)/cp/1e3,1))
"; done; done
