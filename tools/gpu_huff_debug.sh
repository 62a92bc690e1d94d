for D in 0 1 2 4 6 8 14; do echo "DEBUG=$D"; MIJPEG_HUFF_DEBUG=$D RI=8,32 LANES="2 8" tools/gpu_huff_sweep.sh; done
