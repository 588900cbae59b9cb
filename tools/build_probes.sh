#!/bin/bash
# Build the micro-probes (run them on the GPU box through gpurun: ./tools/<name>.bin).
# The binaries are not tracked; they travel with the gpurun snapshot like the .so files.
cd "$(dirname "$0")"
for p in valu_probe pingpong_probe mfma_probe coexec_probe lds_dma_probe place_probe place_probe2 clock_probe f16_probe f16_order_probe gridsync_probe kloop_probe flag_probe launch_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $p.hip -o $p.bin || exit 1
done
