#!/bin/bash
# A/B of the key-split tail round (pd_attention_tail) at 20 / 24 / 40 samples per call, same box
for t in 0 1; do
  PD_ATTN_TAIL=$t python physdock_amd/build.py attention.hip > /dev/null 2>&1
  PD_ATTN_TAIL=$t python tools/b20_time.py 20 24 40 64 2>&1 | grep "B="
done
