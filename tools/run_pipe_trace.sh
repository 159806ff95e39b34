#!/bin/bash
PD_LAB=1 python physdock_amd/build.py --force > /dev/null 2>&1
python tools/attn_pipe_trace.py 64 16 256
python tools/attn_pipe_trace.py 256 4 256
python physdock_amd/build.py --force > /dev/null 2>&1
