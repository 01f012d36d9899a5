#!/bin/bash
export PYTHONPATH=.
for v in 0 1 0 1; do
  echo "gn_cat=$v  sd15: $(SPIDER_GN_CAT=$v python3 scripts/prof_unet.py 40 2>&1 | tail -1)"
  echo "gn_cat=$v  audio: $(SPIDER_GN_CAT=$v python3 scripts/bench_audio.py 40 2>&1 | tail -1 | cut -c1-150)"
done
