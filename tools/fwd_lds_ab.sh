#!/bin/bash
# usage: tools/fwd_lds_ab.sh  -> forward tile kernel per shape with the LDS-staged kernel off (threshold out of reach) and on
for v in 1000000 1; do echo "tile_fwd_lds_min=$v"; python tools/fwd_bench.py 96 332 704 2000 --option tile_fwd_lds_min=$v; done
