#!/bin/bash
# A-B of tile shapes (fused_tile_geometry's environment overrides) over the layouts of tools/layout_bench.py
L=${LAYOUTS:-cmyk,3x1,1x4,lumasub,3x3,444_12,422_12}
for cfg in "X=1" "MIJPEG_TILE_H=64" "MIJPEG_TILE_W=192" "MIJPEG_TILE_W=192 MIJPEG_TILE_H=32" "MIJPEG_TILE_W=64 MIJPEG_TILE_H=64" "MIJPEG_TILE_W=64 MIJPEG_TILE_H=48" "MIJPEG_TILE_W=96 MIJPEG_TILE_H=96"; do
  echo "== $cfg"
  env $cfg LAYOUTS=$L timeout 200 python tools/layout_bench.py 2>&1 | grep "ms/launch" | cut -c1-20,70-125
done
