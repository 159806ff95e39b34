#!/bin/bash
# timing ablations of the pipelined attention on the token shape (wrong results by construction): what is a 256-key launch made of?
for v in 0 1 4 8 12 32 64 96 127; do
  PD_PIPE_ABL=$v python physdock_amd/build.py attn_pipe.hip > /dev/null 2>&1
  python tools/attn_pipe_bench.py 2>&1 | grep -E "token DiT|triangle" | sed "s/HPB=default/ABL=$v/"
done
python physdock_amd/build.py attn_pipe.hip > /dev/null 2>&1
