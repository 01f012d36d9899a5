#!/bin/bash
export PYTHONPATH=.
for cfg in "0 -1" "16 -1" "16 0" "32 0" "16 1" "32 1"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset SPIDER_ATTN_NSPLIT; else export SPIDER_ATTN_NSPLIT=$1; fi
  if [ "$2" = "-1" ]; then unset SPIDER_ATTN_WIDE; else export SPIDER_ATTN_WIDE=$2; fi
  echo "nsplit=$1 wide=$2: $(python3 scripts/prof_decode_batch.py 8 40 2>&1 | tail -1)"
done
