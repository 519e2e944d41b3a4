#!/bin/bash
# Component timing of the LDS-tiled MSDA kernel: RDETR_TILE_DBG masks switch parts of the kernel off (WRONG results, timing
# only): 2 = no window fills, 4 = no passes, 8 = no store, 16 = no location loads, 32 = no MFMA loop.
for m in ${@:-0 2 4 6 32 34 8 30}; do
  echo "dbg=$m: $(RDETR_TILE_DBG=$m python3 tools/ab_msda.py 20 | grep -v max | tr '\n' ' ' | sed 's/GB\/s algorithmic//g')"
done
