#!/bin/bash
export PYTHONPATH=.
for v in 0 1 0 1; do echo "xcd_order=$v $(SPIDER_ATTN_XCD=$v python3 scripts/prof_unet.py 40 2>&1 | tail -1)"; done
