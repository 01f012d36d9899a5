#!/bin/bash
# Decode-attention configuration sweep on the Qwen2.5-7B decode loop (split count x inline combine x wide block)
export PYTHONPATH=.
for cfg in "64 0 -1" "64 1 -1" "16 1 1" "8 1 1" "8 1 0" "16 1 0" "32 1 0" "4 1 1" "16 0 1" "8 0 1"; do
  set -- $cfg
  if [ "$3" = "-1" ]; then unset SPIDER_ATTN_WIDE; else export SPIDER_ATTN_WIDE=$3; fi
  echo "nsplit=$1 inline=$2 wide=$3: $(SPIDER_ATTN_NSPLIT=$1 SPIDER_ATTN_INLINE=$2 python3 scripts/prof_decode.py 66 2>&1 | tail -1)"
done
