mkdir -p gpurun_out/r3
for v in 0 0x80102 0x80104 0x80108 0x80008 0x100108 0x200108 0x040108; do echo "variant $v"; ZGML_COPY_VARIANT=$v python -c "
import sys; sys.path.insert(0,'.')
from zgml_amd import Backend
be=Backend(0)
cp=be._lib.zgml_hip_copy_bench(be.ctx, 1<<30, 3, 20)
print(round(2*(1<<30)/cp/1e3,1), 'GB/s read+write')
"; done
python tools/overlap_bench.py
